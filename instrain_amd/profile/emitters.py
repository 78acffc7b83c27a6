"""On-disk emitters for what the hot path produces, in the reference's formats:

  SNVprofile._store_special    /root/reference/inStrain/SNVprofile.py:717-733   covT / clonT / clonTR -> one HDF5 file,
                               one gzip-compressed 2 x N dataset "{scaffold}::{mm}" per (scaffold, mm):
                               row 0 = values, row 1 = positions (np.array([arr.values, arr.index]))
  SNVprofile._load_special     SNVprofile.py:690-712                            the reader `inStrain compare` uses
  SNVprofile._store_pandas     SNVprofile.py:46-113 (type 'pandas')             DataFrame -> <name>.csv.gz

h5py is used when it can be imported (it is a dependency of the reference, so inside the reference's environment it
always can); otherwise the same HDF5 C library h5py wraps is driven through ctypes (H5Fcreate / H5Dcreate2 / H5Dwrite
with the deflate filter), so the files are ordinary HDF5 either way.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np
import pandas as pd


def store_pandas(df, fileloc):
    """SNVprofile._store_pandas: csv.gz"""
    if not fileloc.endswith('.csv.gz'):
        fileloc += '.csv.gz'
    df.to_csv(fileloc, compression='gzip')
    return fileloc


def load_pandas(fileloc):
    return pd.read_csv(fileloc, index_col=0)


# ---- HDF5 through the C library ----
class _H5:
    def __init__(self):
        cands = [os.environ.get("ISX_HDF5_LIB"), ctypes.util.find_library("hdf5")]
        for pat in ("/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                    "/opt/conda/lib/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
            cands += sorted(glob.glob(pat))
        self.lib = None
        for c in cands:
            if not c:
                continue
            try:
                self.lib = C.CDLL(c)
                break
            except OSError:
                continue
        if self.lib is None:
            raise RuntimeError("neither h5py nor an HDF5 shared library was found (set ISX_HDF5_LIB)")
        L = self.lib
        hid = C.c_int64
        L.H5open()
        self.hid = hid
        g = lambda n: hid.in_dll(L, n).value
        self.T_I64, self.T_F64 = g("H5T_NATIVE_INT64_g"), g("H5T_NATIVE_DOUBLE_g")
        self.P_DCREATE = g("H5P_CLS_DATASET_CREATE_ID_g")
        for fn, res, args in (("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
                              ("H5Fclose", C.c_int, [hid]), ("H5Screate_simple", hid, [C.c_int, C.c_void_p, C.c_void_p]),
                              ("H5Sclose", C.c_int, [hid]), ("H5Pcreate", hid, [hid]), ("H5Pclose", C.c_int, [hid]),
                              ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.c_void_p]), ("H5Pset_deflate", C.c_int, [hid, C.c_uint]),
                              ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
                              ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                              ("H5Dclose", C.c_int, [hid]), ("H5Dget_space", hid, [hid]), ("H5Dget_type", hid, [hid]),
                              ("H5Tget_class", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]),
                              ("H5Sget_simple_extent_dims", C.c_int, [hid, C.c_void_p, C.c_void_p]),
                              ("H5Literate", C.c_int, [hid, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])):
            f = getattr(L, fn)
            f.restype, f.argtypes = res, args

    def write(self, fileloc, datasets):
        L = self.lib
        f = L.H5Fcreate(fileloc.encode(), 2, 0, 0)          # H5F_ACC_TRUNC
        if f < 0:
            raise IOError("cannot create " + fileloc)
        try:
            for name, arr in datasets:
                arr = np.ascontiguousarray(arr)
                t = self.T_F64 if arr.dtype.kind == 'f' else self.T_I64
                arr = arr.astype(np.float64 if arr.dtype.kind == 'f' else np.int64)
                dims = (C.c_uint64 * 2)(*arr.shape)
                sp = L.H5Screate_simple(2, dims, None)
                pl = L.H5Pcreate(self.P_DCREATE)
                if arr.size:                                # gzip needs a chunked layout (h5py picks chunks itself)
                    chunk = (C.c_uint64 * 2)(arr.shape[0], max(1, min(arr.shape[1], 1 << 16)))
                    L.H5Pset_chunk(pl, 2, chunk)
                    L.H5Pset_deflate(pl, 4)                 # h5py's default gzip level
                d = L.H5Dcreate2(f, name.encode(), t, sp, 0, pl, 0)
                if d < 0:
                    raise IOError("cannot create dataset " + name)
                if arr.size and L.H5Dwrite(d, t, 0, 0, 0, arr.ctypes.data) < 0:
                    raise IOError("cannot write dataset " + name)
                L.H5Dclose(d); L.H5Pclose(pl); L.H5Sclose(sp)
        finally:
            L.H5Fclose(f)

    def read(self, fileloc):
        L = self.lib
        f = L.H5Fopen(fileloc.encode(), 0, 0)               # H5F_ACC_RDONLY
        if f < 0:
            raise IOError("cannot open " + fileloc)
        names = []
        CB = C.CFUNCTYPE(C.c_int, self.hid, C.c_char_p, C.c_void_p, C.c_void_p)

        def visit(g, name, info, data):
            names.append(name.decode())
            return 0
        cb = CB(visit)
        L.H5Literate(f, 0, 0, None, cb, None)               # H5_INDEX_NAME, H5_ITER_INC
        out = {}
        try:
            for name in names:
                d = L.H5Dopen2(f, name.encode(), 0)
                sp = L.H5Dget_space(d)
                dims = (C.c_uint64 * 2)()
                L.H5Sget_simple_extent_dims(sp, dims, None)
                ty = L.H5Dget_type(d)
                is_f = L.H5Tget_class(ty) == 1              # H5T_FLOAT
                arr = np.empty((dims[0], dims[1]), dtype=np.float64 if is_f else np.int64)
                if arr.size:
                    L.H5Dread(d, self.T_F64 if is_f else self.T_I64, 0, 0, 0, arr.ctypes.data)
                out[name] = arr
                L.H5Tclose(ty); L.H5Sclose(sp); L.H5Dclose(d)
        finally:
            L.H5Fclose(f)
        return out


_h5 = None


def _backend():
    global _h5
    try:
        import h5py
        return h5py
    except Exception:
        if _h5 is None:
            _h5 = _H5()
        return _h5


def store_special(obj, fileloc):
    """obj = {scaffold: {mm: pd.Series(values, index=positions)}} (covT / clonT / clonTR of SNVprofile) -> <fileloc>.hd5
    with the reference's layout (SNVprofile.py:717-733)."""
    if not fileloc.endswith('.hd5'):
        fileloc += '.hd5'
    sets = [("{0}::{1}".format(scaff, mm), np.array([arr.values, arr.index])) for scaff, d in obj.items() for mm, arr in d.items()]
    be = _backend()
    if hasattr(be, "File"):
        with be.File(fileloc, "w") as f:
            for name, data in sets:
                f.create_dataset(name, data=data, compression="gzip")
    else:
        be.write(fileloc, sets)
    return fileloc


def load_special(fileloc, kind="coverage", scaffolds=None):
    """-> {scaffold: {mm: pd.Series}} the way SNVprofile._load_special hands covT / clonT to `inStrain compare`
    (SNVprofile.py:690-712, compare_controller.py:520-577): int32 coverage / float32 clonality, int positions."""
    be = _backend()
    if hasattr(be, "File"):
        with be.File(fileloc, "r") as f:
            raw = {k: np.array(f[k]) for k in f.keys()}
    else:
        raw = be.read(fileloc)
    out = {}
    for key, data in raw.items():
        scaff, mm = key.rsplit("::", 1)
        if scaffolds is not None and scaff not in scaffolds:
            continue
        dt = "int32" if kind == "coverage" else "float32"
        out.setdefault(scaff, {})[int(mm)] = pd.Series(data[0].astype(dt), index=np.array(data[1]).astype("int"))
    return out
