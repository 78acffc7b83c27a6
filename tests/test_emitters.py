"""CPU-only: covT / clonT HDF5 emitters (SNVprofile._store_special layout) and the csv.gz table writer round trip."""
import os
import shutil
import subprocess

import numpy as np
import pandas as pd
import pytest

from instrain_amd.profile import emitters


def _have_hdf5():
    try:
        emitters._backend()
        return True
    except RuntimeError:
        return False


@pytest.mark.skipif(not _have_hdf5(), reason="no h5py / libhdf5 on this host")
def test_covT_clonT_hdf5_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    covT, clonT = {}, {}
    for s in ("scaf::odd name", "contig_2", "empty_levels"):
        covT[s], clonT[s] = {}, {}
        for mm in (0, 1, 5):
            pos = np.sort(rng.choice(50_000, 3000 if s != "empty_levels" or mm == 0 else 0, replace=False))
            covT[s][mm] = pd.Series(rng.integers(1, 900, len(pos)).astype("int32"), index=pos)
            clonT[s][mm] = pd.Series(rng.random(len(pos)).astype("float32"), index=pos)
    fc = emitters.store_special(covT, str(tmp_path / "covT"))
    fl = emitters.store_special(clonT, str(tmp_path / "clonT"))
    assert fc.endswith("covT.hd5") and os.path.getsize(fc) > 1000
    assert open(fc, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    gc = emitters.load_special(fc, "coverage")
    gl = emitters.load_special(fl, "clonality", scaffolds={"contig_2", "empty_levels"})
    assert sorted(gc) == sorted(covT) and sorted(gl) == ["contig_2", "empty_levels"]
    for s in covT:
        assert sorted(gc[s]) == [0, 1, 5]
        for mm in covT[s]:
            a, b = gc[s][mm], covT[s][mm]
            assert a.dtype == np.int32 and (a.values == b.values).all() and (a.index.values == b.index.values).all()
    for s in gl:
        for mm in clonT[s]:
            a, b = gl[s][mm], clonT[s][mm]
            assert a.dtype == np.float32 and (a.values == b.values).all() and (a.index.values == b.index.values).all()
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump:                                       # an independent reader agrees on names, shapes, filter
        out = subprocess.run([h5dump, "-H", "-p", fc], capture_output=True, text=True).stdout
        assert 'DATASET "contig_2::5"' in out and "DEFLATE" in out and "( 2, 3000 )" in out


def test_csv_gz_tables(tmp_path):
    df = pd.DataFrame({"scaffold": ["a", "b"], "position": [1, 2], "r2": [0.5, np.nan]})
    f = emitters.store_pandas(df, str(tmp_path / "raw_linkage_table"))
    assert f.endswith(".csv.gz")
    back = emitters.load_pandas(f)
    pd.testing.assert_frame_equal(back, df)
