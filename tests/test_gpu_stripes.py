"""GPU parity of the STRIPE path of k_pileup_dense (round 6): one-mm-bin reference-delta batches of a pipe slot without a count table --
a thread owns eight positions, their coverage comes straight from the difference row and leaves as whole words, only the positions
that reach min_cov (or rarefied_coverage) are compacted into the second pass.  Tables must be byte for byte those of the round-4
epilogue (isx_params.layout = ISX_LAYOUT_NO_STRIPES) and of a slot that keeps the count table (which never takes the stripe path);
tests/test_gpu_reads.py / test_gpu_planes.py pin those against the golden vectors and the oracle.
Reference semantics: profile_utilities.py:288-295 (update_covT), snv_utilities.py:85-104 (what a position below min_cov gets)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NO_STRIPES = 64     # isx_params.layout: ISX_LAYOUT_NO_STRIPES


@pytest.fixture(scope="module")
def ctx():
    from instrain_amd import engine
    from tests import util
    c = engine.Context(0)
    lut, fb = util.load_lut()
    c.set_null_model(lut, fb)
    yield c
    c.close()


def _batches():
    from instrain_amd import synth
    ws = [synth.make_workload(genome_len=400_003, coverage=3, n_sites=300, seed=159, skip_mm=True),     # n_pos not a multiple of 8: the last stripe
          synth.make_workload(genome_len=300_000, coverage=6, n_sites=300, seed=161, skip_mm=True),
          synth.make_workload(genome_len=200_001, coverage=60, n_sites=400, seed=162, skip_mm=True),
          synth.make_workload(genome_len=60_000, coverage=300, n_sites=100, seed=163, skip_mm=True, err=0.01)]
    # a reference with positions that are not A/C/T/G: every observed base there is an exception
    n = ws[1]["ref_codes"].copy()
    rng = np.random.default_rng(5)
    n[rng.integers(0, len(n), 2000)] = 4
    ws[1] = dict(ws[1], ref_codes=n)
    return ws, [synth.segs_from_obs(w["obs"], w["pair"]) for w in ws]


def _run(ctx, ws, segs, **kw):
    from instrain_amd import engine
    cap = dict(max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(s.n_seg for s in segs),
               max_splits=max(len(w["split_bounds"]) for w in ws), depth=2, host_threads=4, pin_threads=False, n_mm_bins=1, min_snp=20)
    cap.update(kw)
    pipe = engine.Pipe(ctx, **cap)
    res = []
    for w, sg in zip(ws, segs):
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], sg)
        r = pipe.collect(t)
        d = {k: r[k].copy() for k in ("cov16", "clon", "snv") if k in r}
        if "ld" in r:
            d["ld"] = r["ld"].copy()
        d["rare"] = r["rare"].copy()
        d["sizes"] = r["sizes"]
        d["kernel_ms"] = r["stats"].get("kernel_ms")
        res.append(d)
        pipe.release(t)
    pipe.close()
    return res


def _same(a, b, what):
    assert a["sizes"] == b["sizes"], what
    for k in ("cov16", "snv", "ld", "rare"):
        if k in a or k in b:
            assert a[k].tobytes() == b[k].tobytes(), (k, what)
    assert a["clon"].view(np.uint32).tobytes() == b["clon"].view(np.uint32).tobytes(), what


@pytest.mark.parametrize("linkage", [False, True])
@pytest.mark.parametrize("lean", [False, True])
def test_stripes_equal_the_per_position_epilogue(ctx, linkage, lean):
    """shallow (4-bit plane in a lean slot), mid, deep and very deep batches; N positions in the reference; n_pos that is no multiple of 8"""
    ws, segs = _batches()
    got = _run(ctx, ws, segs, enable_linkage=linkage, lean_output=lean)
    old = _run(ctx, ws, segs, enable_linkage=linkage, lean_output=lean, layout=NO_STRIPES)
    for i, (a, b) in enumerate(zip(got, old)):
        _same(a, b, ("stripes vs per-position", i, linkage, lean))
    if not lean:
        kept = _run(ctx, ws, segs, enable_linkage=linkage, want_counts=True)       # the count table: never the stripe path
        for i, (a, b) in enumerate(zip(got, kept)):
            _same(a, b, ("stripes vs count table", i, linkage))
    # coverage against the observations themselves
    for w, a in zip(ws, got):
        assert (a["cov16"] == np.minimum(np.bincount(w["obs"]["gpos"], minlength=w["n_pos"]), 65535)).all()


@pytest.mark.parametrize("window", [512, 1024, 2752])
@pytest.mark.parametrize("rarefied,min_cov", [(3, 5), (50, 1), (0, 5), (24, 0)])
def test_stripes_small_windows_and_gates(ctx, window, rarefied, min_cov):
    """windows that leave most lanes without a stripe; rarefied_coverage below min_cov (positions go on for clonTR alone), min_cov 0 / 1
    (every position goes on), rarefied off"""
    ws, segs = _batches()
    ws, segs = ws[:2], segs[:2]
    kw = dict(enable_linkage=True, window=window, rarefied_coverage=rarefied, min_cov=min_cov)
    for lean in (False, True):
        got = _run(ctx, ws, segs, lean_output=lean, **kw)
        old = _run(ctx, ws, segs, lean_output=lean, layout=NO_STRIPES, **kw)
        for i, (a, b) in enumerate(zip(got, old)):
            _same(a, b, (i, window, rarefied, min_cov, lean))
