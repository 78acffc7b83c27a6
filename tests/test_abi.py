"""CPU-only: the C-ABI library loads and exports every symbol include/instrain_amd.h declares;
struct layouts used by the ctypes binding match the header; no compute calls."""
import ctypes as C
import os
import re

from instrain_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(REPO, "include", "instrain_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(isx_[a-z_0-9]+)\s*\(", h)))


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(syms) == sorted(_lib.SYMBOLS)


def test_abi_version_and_error_string():
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "instrain_amd.h")).read()
    declared = int(re.search(r"#define\s+ISX_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.isx_abi_version() == declared == _lib.ABI_VERSION == 5
    assert isinstance(lib.isx_last_error(), bytes)


def test_loader_refuses_another_abi_version(tmp_path):
    """a library of another ABI version is never called into (isx_pipe_result / isx_pipe_params changed size and meaning between versions)"""
    import subprocess
    import sys
    src = tmp_path / "old.c"
    src.write_text("int isx_abi_version(void) { return 3; }\nconst char *isx_last_error(void) { return \"\"; }\n")
    so = tmp_path / "libold.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    code = ("import os, sys; os.environ['ISX_LIB'] = %r; sys.path.insert(0, %r)\nfrom instrain_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.IsxError as e:\n    assert 'ABI version 3' in str(e), str(e); print('refused')\n" % (str(so), REPO))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "refused" in r.stdout, r.stderr[-1500:]


def test_struct_sizes_match_header(tmp_path):
    """sizeof() of every ABI struct as the C compiler sees the header == the ctypes / numpy mirror."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "instrain_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(isx_params),sizeof(isx_sizes),sizeof(isx_timings),sizeof(isx_bam_params),sizeof(isx_bam_info),'
                   'sizeof(isx_obs),sizeof(isx_entry),sizeof(isx_snv),sizeof(isx_ld),sizeof(isx_scaffold_level),sizeof(isx_compare_level),sizeof(isx_compare_snp),sizeof(isx_pipe_params),sizeof(isx_pipe_result),'
                   'sizeof(isx_read_planes),sizeof(isx_ref_planes),sizeof(isx_segs));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    c_sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    py_sizes = [C.sizeof(_lib.Params), C.sizeof(_lib.Sizes), C.sizeof(_lib.Timings), C.sizeof(_lib.BamParams),
                C.sizeof(_lib.BamInfo), _lib.OBS_DT.itemsize, _lib.ENTRY_DT.itemsize, _lib.SNV_DT.itemsize,
                _lib.LD_DT.itemsize, _lib.SCAFFOLD_LEVEL_DT.itemsize, _lib.COMPARE_LEVEL_DT.itemsize,
                _lib.COMPARE_SNP_DT.itemsize, C.sizeof(_lib.PipeParams), C.sizeof(_lib.PipeResult),
                C.sizeof(_lib.ReadPlanes), C.sizeof(_lib.RefPlanes), C.sizeof(_lib.Segs)]
    assert c_sizes == py_sizes, (c_sizes, py_sizes)


def test_no_gpu_is_a_loud_error_not_a_fallback():
    """Without a visible MI355X, context creation must fail (there is no CPU path in the product)."""
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.isx_ctx_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"device" in lib.isx_last_error().lower() or b"hip" in lib.isx_last_error().lower()


def test_host_register_arguments_and_no_gpu():
    """isx_host_register (round 6): bad arguments and unknown ranges are errors; without a device nothing can be pinned for its copy engine
    and the call says so (the pipe then stages the planes as before: registration is an optimisation a caller asks for, never a silent no-op)"""
    import numpy as np
    import torch
    lib = _lib.load()
    buf = np.zeros(1 << 16, np.uint8)
    assert lib.isx_host_register(None, 16) != 0 and lib.isx_host_register(buf.ctypes.data, 0) != 0
    assert lib.isx_host_unregister(buf.ctypes.data) != 0 and b"not a registered range" in lib.isx_last_error()
    if not torch.cuda.is_available():
        assert lib.isx_host_register(buf.ctypes.data, buf.nbytes) != 0
        assert lib.isx_host_unregister(buf.ctypes.data) != 0


def test_product_never_imports_oracle():
    import glob
    for f in glob.glob(os.path.join(REPO, "instrain_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src, f
    for f in glob.glob(os.path.join(REPO, "instrain_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "oracle" not in open(f).read(), f


def test_graft_entry_build_passes():
    """the driver's "does it build" check: __graft_entry__.build() (make is up to date here: seconds) with its own assertions --
    every declared symbol present, the ABI version the loader and the header agree on"""
    import __graft_entry__
    __graft_entry__.build()
