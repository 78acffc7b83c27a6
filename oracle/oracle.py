"""
oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ctypes front-end of oracle_core.c (the plain-C restatement of the reference's
per-split hot path) plus the null-model LUT builder
(/root/reference/inStrain/profile/snv_utilities.py:14-38 generate_snp_model).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BASES = "ACTG"      # P2C order, profile_utilities.py:34
CLASSES = ["AmbiguousReference", "DivergentSite", "SNS", "SNV", "con_SNV", "pop_SNV"]

ENTRY_DT = np.dtype([("pos", "<i4"), ("mm", "<i4"), ("cnt", "<i8", (4,)), ("clon", "<f4")], align=True)
SNV_DT = np.dtype([("pos", "<i4"), ("mm", "<i4"), ("cnt", "<i8", (4,)), ("ref_base", "i1"),
                   ("con_base", "i1"), ("var_base", "i1"), ("allele_count", "i1"), ("cls", "i1"),
                   ("cryptic", "i1"), ("position_coverage", "<i8")], align=True)
LD_DT = np.dtype([("pos_a", "<i4"), ("pos_b", "<i4"), ("mm", "<i4"), ("distance", "<i4"),
                  ("total", "<i8"), ("cAB", "<i8"), ("cAb", "<i8"), ("caB", "<i8"), ("cab", "<i8"),
                  ("allele_A", "i1"), ("allele_a", "i1"), ("allele_B", "i1"), ("allele_b", "i1"),
                  ("r2", "<f8"), ("d_prime", "<f8")], align=True)


class _Result(C.Structure):
    _fields_ = [("n_entries", C.c_int64), ("n_snv", C.c_int64), ("n_ld", C.c_int64),
                ("entries", C.c_void_p), ("snv", C.c_void_p), ("ld", C.c_void_p),
                ("n_edges", C.c_int64), ("n_increments", C.c_int64)]


def build():
    """Compile oracle_core.c -> liboracle.so (gcc only; no reference sources involved)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        lib = C.CDLL(so)
        lib.orc_profile_split.restype = C.POINTER(_Result)
        lib.orc_profile_split.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                          C.c_int32, C.c_int64, C.c_double, C.c_int64]
        lib.orc_free.argtypes = [C.POINTER(_Result)]
        lib.orc_free.restype = None
        assert ENTRY_DT.itemsize == 48 and SNV_DT.itemsize == 56 and LD_DT.itemsize == 80
        _LIB = lib
    return _LIB


def _copy(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


def null_model_lut(model, n=10001):
    """dict {coverage: k, -1: fallback} -> (int32 lut[n] with -1 for missing, fallback)."""
    lut = np.full(n, -1, dtype=np.int32)
    for k, v in model.items():
        if 0 <= k < n:
            lut[k] = v
    return lut, int(model[-1])


def generate_snp_model(model_file, fdr=1e-6):
    """snv_utilities.py:14-38: model[cov] = 0-based index of the first column of the
    NullModel.txt row whose probability is < fdr; model[-1] = max of those."""
    model = {}
    with open(model_file) as f:
        for line in f:
            if "coverage" in line:
                continue
            parts = line.split()
            for i, c in enumerate(parts[1:]):
                if float(c) < fdr:
                    model[int(parts[0])] = i
                    break
    model[-1] = max(model.values())
    return model


def profile_split(pos, base, mm, pair, seq, start, lut, fallback, min_cov=5, min_freq=0.05,
                  min_snp=20, convert=True):
    """Run the C oracle on one split's packed observations. Returns dict of structured arrays
    (positions absolute) + n_edges / n_increments."""
    lib = _lib()
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.uint8)
    mm = np.ascontiguousarray(mm, dtype=np.int32)
    pair = np.ascontiguousarray(pair, dtype=np.int32)
    lut = np.ascontiguousarray(lut, dtype=np.int32)
    seqb = seq.encode() if isinstance(seq, str) else bytes(seq)
    r = lib.orc_profile_split(len(pos), pos.ctypes.data, base.ctypes.data, mm.ctypes.data,
                              pair.ctypes.data, seqb, len(seqb), int(start), lut.ctypes.data,
                              len(lut), int(fallback), int(min_cov), float(min_freq), int(min_snp))
    if not convert:             # timing only (bench.py's cpu_baseline): sizes, no table copies
        rc = r.contents
        out = {"n_entries": int(rc.n_entries), "n_snv": int(rc.n_snv), "n_ld": int(rc.n_ld)}
        lib.orc_free(r)
        return out
    try:
        rc = r.contents
        out = {"entries": _copy(rc.entries, rc.n_entries, ENTRY_DT),
               "snv": _copy(rc.snv, rc.n_snv, SNV_DT),
               "ld": _copy(rc.ld, rc.n_ld, LD_DT),
               "n_edges": int(rc.n_edges), "n_increments": int(rc.n_increments)}
    finally:
        lib.orc_free(r)
    return out
