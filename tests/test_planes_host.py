"""Host side of the bit-plane hand-over (no GPU): isx_read_planes / isx_ref_planes, the XOR stager (isx_encode_planes) against the
byte-compare stager (isx_encode_delta, itself pinned by tests/test_segs_host.py), the BAM front end's plane emission, the read packer.
The planes must stand for exactly the observations of the segments they replace (profile_utilities.py:150-153, 268-286)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from instrain_amd import engine, synth
from instrain_amd._lib import IsxError
from tests import util
from tests.test_segs_host import _mutated_workload, _workload

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _planes_numpy(segs):
    """the definition of the format, slowly: [n, 8] uint64 from the 3-bit codes"""
    cd = engine.unpack_codes(segs.bases).astype(np.uint64)
    live = np.arange(150)[None, :] < segs.len[:, None]
    obs = live & (cd < 4)
    out = np.zeros((segs.n_seg, 8), dtype=np.uint64)
    mark = live & (cd == 5)         # a base that is not A/C/T/G but passed the filter: code 1 at its (not observed) column + the line's flag
    for j in range(150):
        out[:, j // 32] |= np.where(obs[:, j], cd[:, j], np.where(mark[:, j], 1, 0)).astype(np.uint64) << np.uint64(2 * (j % 32))
        out[:, 5 + j // 64] |= (live[:, j] & ~obs[:, j]).astype(np.uint64) << np.uint64(j % 64)
    out[:, 7] |= mark.any(axis=1).astype(np.uint64) << np.uint64(63)
    return out


def _masked(planes, length):
    """planes with everything a consumer must ignore cleared: base bits of columns that are not observed, skip bits from len on"""
    p = np.array(planes, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros_like(p)
    for j in range(150):
        live = j < length
        sk = ((p[:, 5 + j // 64] >> np.uint64(j % 64)) & np.uint64(1)).astype(bool) & live
        out[:, 5 + j // 64] |= sk.astype(np.uint64) << np.uint64(j % 64)
        code = (p[:, j // 32] >> np.uint64(2 * (j % 32))) & np.uint64(3)
        out[:, j // 32] |= np.where(live & ~sk, code, 0).astype(np.uint64) << np.uint64(2 * (j % 32))
    return out


def test_planes_from_segs_and_reference_planes():
    w = _workload(seed=5, G=40_000, cov=10, skip_mm=True)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    # some code-5 columns (a base that is not A/C/T/G) and short segments
    cd = engine.unpack_codes(segs.bases)
    rng = np.random.default_rng(3)
    hit = rng.random(cd.shape) < 0.01
    cd[hit & (cd < 4)] = 5
    segs = engine.SegBatch(segs.gpos, segs.len, engine.pack_codes(cd), None, segs.pair)
    for threads in (1, 3):
        pb = engine.PlaneBatch.from_segs(segs, threads=threads)
        assert pb.planes.ctypes.data % 64 == 0
        assert (pb.planes == _planes_numpy(segs)).all()
    # the reference: 2-bit plane (non-bases as 0) + the plane of non-bases, any length
    for n in (1, 7, 8, 9, 64, 1001, 262144 + 13):
        ref = rng.integers(0, 4, n).astype(np.uint8)
        for frac in (0.0, 0.02):
            r = ref.copy()
            r[rng.random(n) < frac] = rng.integers(4, 255)
            rp = engine.RefPlanes.from_codes(r, threads=2)
            c = np.where(r < 4, r, 0).astype(np.uint8)
            c = np.r_[c, np.zeros((-n) % 4, np.uint8)].reshape(-1, 4)
            assert (rp.plane2 == (c[:, 0] | c[:, 1] << 2 | c[:, 2] << 4 | c[:, 3] << 6)).all()
            if (r > 3).any():
                assert (np.unpackbits(rp.nplane, bitorder="little")[:n] == (r > 3)).all()
            else:
                assert rp.nplane is None


def _both(segs, ref, threads=2, **kw):
    a = engine.encode_delta(segs, ref, threads=threads, **kw)
    b = engine.encode_planes(engine.PlaneBatch.from_segs(segs, threads=threads), engine.RefPlanes.from_codes(ref, threads=threads), threads=threads, **kw)
    return a, b


@pytest.mark.parametrize("variant", [None, 0, 1, 2])
def test_encode_planes_equals_encode_delta(variant):
    """the XOR stager writes the byte-compare stager's records, bit for bit: clean reads (dual records), reads with skipped columns (full
    records), many mismatches (pieces), a reference with non-ACGT positions (N plane), segments at the very end of the flat space, short
    segments, input that is not sorted by start (group rebase), the staging ring -- through every compiled variant of the per-segment
    pass (portable / BMI2 / AVX-512 VBMI2 + GFNI; the choice is made once per process, so the forced ones run in a subprocess)"""
    if variant is not None:
        code = ("import os, sys; os.environ['ISX_PLANES_VARIANT'] = '%d'; sys.path.insert(0, %r); import tests.test_planes_host as t; "
                "t.test_encode_planes_equals_encode_delta(None)" % (variant, REPO))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    w = _workload(seed=21, G=300_000, cov=8, skip_mm=True, p_keep=0.995)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    for threads in (1, 3):
        a, b = _both(segs, w["ref_codes"], threads=threads)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[3] == b[3]
    n_dual = int((a[0][:, 0] >> 31).sum())
    assert 0 < n_dual < len(a[0])
    # pieces + N plane
    w2, ref2 = _mutated_workload(seed=22)
    segs2 = synth.segs_from_obs(w2["obs"], w2["pair"])
    a, b = _both(segs2, ref2)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[3] == b[3] and a[3] > 1
    g, ln, mm, cd, pr, full = engine.decode_delta(b[0], b[1], ref2)
    gg, bb, _, _ = util.segs_to_obs(segs2)
    from tests.test_segs_host import _pieces_to_columns
    assert (_pieces_to_columns(g, ln, cd, len(ref2)) == np.sort(gg * 8 + bb)).all()
    # the ring
    r2 = engine.encode_planes(engine.PlaneBatch.from_segs(segs2), engine.RefPlanes.from_codes(ref2), threads=2, slack_groups=a[3], ring_records=2 * 32768)
    assert (r2[0] == a[0]).all() and (r2[1] == a[1]).all()
    with pytest.raises(IsxError):
        engine.encode_planes(engine.PlaneBatch.from_segs(segs2), engine.RefPlanes.from_codes(ref2), slack_groups=1, retry=False)
    # random segments everywhere incl. the last positions, short ones, unsorted starts, with / without N positions
    rng = np.random.default_rng(11)
    for n_pos, n_seg, n_frac, sort in ((1000, 300, 0.0, True), (777, 500, 0.05, True), (5000, 2000, 0.01, False), (150, 40, 0.0, True), (260, 64, 0.3, False)):
        ref = rng.integers(0, 4, n_pos).astype(np.uint8)
        ref[rng.random(n_pos) < n_frac] = 4
        ln = np.where(rng.random(n_seg) < 0.7, min(150, n_pos), rng.integers(1, min(150, n_pos) + 1, n_seg)).astype(np.uint8)
        gpos = (rng.random(n_seg) * (n_pos - ln.astype(np.int64) + 1)).astype(np.uint32)
        gpos[: n_seg // 8] = (n_pos - ln[: n_seg // 8].astype(np.int64)).astype(np.uint32)       # ... ending exactly at n_pos
        if sort:
            o = np.argsort(gpos, kind="stable")
            gpos, ln = gpos[o], ln[o]
        cd = np.full((n_seg, 150), 4, dtype=np.uint8)
        for i in range(n_seg):
            L = int(ln[i])
            base = np.where(ref[gpos[i]:gpos[i] + L] < 4, ref[gpos[i]:gpos[i] + L], rng.integers(0, 4, L))
            mut = rng.random(L) < rng.choice([0.0, 0.01, 0.2])
            base = np.where(mut, (base + rng.integers(1, 4, L)) & 3, base)
            drop = rng.random(L) < rng.choice([0.0, 0.0, 0.1])
            cd[i, :L] = np.where(drop, rng.integers(4, 6, L), base)
        segs3 = engine.SegBatch(gpos, ln, engine.pack_codes(cd), None, rng.integers(0, 1 << 20, n_seg).astype(np.uint32))
        a, b = _both(segs3, ref, threads=2)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all(), (n_pos, n_seg)
    # no segments at all: one empty group
    e = engine.SegBatch(np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros((0, 15), np.uint32))
    a, b = _both(e, np.zeros(100, np.uint8))
    assert len(b[0]) == 32 and (a[0] == b[0]).all()


def test_encode_planes_ignores_what_the_format_says_it_ignores():
    """base bits of skipped columns and anything from column len on may hold garbage"""
    w = _workload(seed=8, G=50_000, cov=6, skip_mm=True)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    pb = engine.PlaneBatch.from_segs(segs)
    rp = engine.RefPlanes.from_codes(w["ref_codes"])
    good = engine.encode_planes(pb, rp)
    rng = np.random.default_rng(1)
    dirty = pb.planes.copy()
    junk = rng.integers(0, 1 << 63, dirty.shape, dtype=np.uint64) | (rng.integers(0, 2, dirty.shape, dtype=np.uint64) << np.uint64(63))
    cd = engine.unpack_codes(segs.bases)
    for j in range(150):
        dead = (j >= segs.len) | (cd[:, j] >= 4)                   # skipped or beyond the segment: base bits are free
        m = np.uint64(3) << np.uint64(2 * (j % 32))
        dirty[:, j // 32] = np.where(dead, (dirty[:, j // 32] & ~m) | (junk[:, j // 32] & m), dirty[:, j // 32])
        beyond = j >= segs.len                                     # skip bits from len on are free
        m1 = np.uint64(1) << np.uint64(j % 64)
        dirty[:, 5 + j // 64] = np.where(beyond, (dirty[:, 5 + j // 64] & ~m1) | (junk[:, 5 + j // 64] & m1), dirty[:, 5 + j // 64])
    dirty[:, 4] |= junk[:, 4] & (np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(2 * (150 % 32)))         # columns 150 .. 159
    dirty[:, 7] |= junk[:, 7] & (np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(150 - 128))
    got = engine.encode_planes(engine.PlaneBatch(pb.gpos, pb.len, dirty, pb.pair), rp)
    assert (got[0] == good[0]).all() and (got[1] == good[1]).all()


def test_encode_planes_rejects_bad_input():
    ref = engine.RefPlanes.from_codes(np.zeros(1000, np.uint8))
    pl = np.zeros((2, 8), np.uint64)
    with pytest.raises(IsxError, match="beyond n_pos"):
        engine.encode_planes(engine.PlaneBatch([10, 900], [150, 150], pl), ref)
    with pytest.raises(IsxError, match=r"\[1, 150\]"):
        engine.encode_planes(engine.PlaneBatch([10, 20], [150, 0], pl), ref)
    with pytest.raises(IsxError, match=r"\[1, 150\]"):
        engine.encode_planes(engine.PlaneBatch([10, 20], [151, 10], pl), ref)


@pytest.mark.parametrize("avx512", [True, False])
def test_bam_front_end_emits_the_planes_of_its_segments(avx512):
    """isx_bam_copy_read_planes (straight from the records' 4-bit seq + qualities, 64 bases a step) == the planes of the front end's own
    3-bit segments, on the golden BAMs (indels, soft clips, N bases, low qualities, odd query offsets); the scalar emission agrees"""
    if not avx512:
        code = ("import os, sys; os.environ['ISX_NO_AVX512'] = '1'; sys.path.insert(0, %r); import tests.test_planes_host as t; "
                "t.test_bam_front_end_emits_the_planes_of_its_segments(True)" % REPO)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    n_checked = n_marked = 0
    for name in ("sars_cov_2.sorted.bam", "SmallScaffold.fa.sorted.bam", "filter_modes.bam"):
        path = os.path.join(REPO, "tests", "golden", name)
        bam = engine.BamFile(path)
        bam.scan()
        bam.filter(skip_mm=True)
        refs = np.arange(len(bam.refs()), dtype=np.int32)
        segs, _, _ = bam.segment_refs(refs, skip_mm=True)
        planes = bam.read_planes()
        assert planes.shape == (segs.n_seg, 8)
        assert (_masked(planes, segs.len) == _masked(_planes_numpy(segs), segs.len)).all(), name
        n_checked += segs.n_seg
        # mm profiling on (round 6): the non-ACGT bases that pass the filter are MARKED (code 1 at their column, the line's flag), every
        # other column that is not observed has code 0 -- the planes are the definition's, bit for bit
        bam.filter(skip_mm=False)
        segs, _, _ = bam.segment_refs(refs, skip_mm=False)
        planes = bam.read_planes()
        assert (planes == _planes_numpy(segs)).all(), name
        n_marked += int((planes[:, 7] >> np.uint64(63)).sum())
        bam.close()
    assert n_checked > 20000 and n_marked > 0


def test_pack_read_planes_equals_pack_reads():
    rng = np.random.default_rng(4)
    n = 200
    cig, cig_off, seqs, quals, seq_off, starts = [], [0], [], [], [0], []
    for r in range(n):
        ops = []
        q = 0
        for _ in range(rng.integers(1, 5)):
            op = rng.choice([0, 0, 0, 1, 2, 4, 7, 8, 3])
            ln = int(rng.integers(1, 260 if op in (0, 7, 8) else 12))
            ops.append((ln << 4) | int(op))
            if op in (0, 1, 4, 7, 8):
                q += ln
        cig += ops
        cig_off.append(len(cig))
        seqs.append(rng.choice(list(b"ACGTN"), q, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8))
        quals.append(rng.choice([12, 25, 30, 37], q, p=[0.03, 0.07, 0.1, 0.8]).astype(np.uint8))
        seq_off.append(seq_off[-1] + q)
        starts.append(int(rng.integers(-50, 5000)))
    cigars = [np.array(cig[cig_off[r]:cig_off[r + 1]], np.uint32) for r in range(n)]
    args = dict(ref_start=np.array(starts, np.int64), clip_lo=np.zeros(n, np.int64), clip_hi=np.full(n, 5200, np.int64),
                cigars=cigars, seqs=[bytes(x) for x in seqs], quals=quals, pair=np.arange(n, dtype=np.uint32) // 2)
    segs = engine.pack_reads(mm=None, **args)
    pb = engine.pack_read_planes(**args)
    assert pb.n_seg == segs.n_seg and (pb.gpos == segs.gpos).all() and (pb.len == segs.len).all() and (pb.pair == segs.pair).all()
    assert (pb.planes == _planes_numpy(segs)).all()


def _guarded(data):
    """a copy of `data` (uint8) that ENDS at a page boundary followed by an inaccessible page: a read past the end faults"""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    n = len(data)
    size = (n + page - 1) // page * page + page
    m = mmap.mmap(-1, size)
    buf = np.frombuffer(m, dtype=np.uint8)
    start = size - page - n
    buf[start:start + n] = data
    libc = ctypes.CDLL(None, use_errno=True)
    addr = buf.ctypes.data + size - page
    assert libc.mprotect(ctypes.c_void_p(addr), ctypes.c_size_t(page), 0) == 0        # PROT_NONE
    return buf[start:start + n], m


def _guard_page_body():
    rng = np.random.default_rng(9)
    for n_pos in (4096 * 8, 4096 * 8 + 5, 50_003):
        ref = rng.integers(0, 4, n_pos).astype(np.uint8)
        ref[rng.random(n_pos) < 0.05] = 4
        ref[-300:][rng.random(300) < 0.3] = 4
        starts = np.r_[np.arange(max(0, n_pos - 520), n_pos - 1, 1), [0, 10, n_pos - 1]].astype(np.uint32)
        starts.sort()
        ln = np.minimum(150, n_pos - starts.astype(np.int64)).astype(np.uint8)
        codes = np.full((len(starts), 150), 4, np.uint8)
        for i, (s, l) in enumerate(zip(starts.tolist(), ln.tolist())):
            c = ref[s:s + l].copy()
            c[c > 3] = rng.integers(0, 4, int((c > 3).sum()))
            flip = rng.random(l) < 0.03
            c[flip] = (c[flip] + 1) & 3
            c[rng.random(l) < 0.05] = 4
            codes[i, :l] = c
        segs = engine.SegBatch(starts, ln, engine.pack_codes(codes), None, np.arange(len(starts), dtype=np.uint32))
        want = engine.encode_delta(segs, ref, threads=2)
        rp = engine.RefPlanes.from_codes(ref, threads=1)
        p2, keep2 = _guarded(rp.plane2[:(n_pos + 3) // 4])
        pn, keepn = _guarded(rp.nplane[:(n_pos + 7) // 8])
        got = engine.encode_planes(engine.PlaneBatch.from_segs(segs), engine.RefPlanes(p2, pn, n_pos), threads=2)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_planes_stager_never_reads_past_the_caller_planes(variant):
    """isx_encode_planes read up to 7 bytes past the caller's non-ACGT bit plane for segments starting in the last ~250 positions.
    Both reference planes end at an inaccessible page here (a stray read is a segfault: the body runs in a subprocess), segments start at
    every one of the last 520 positions, through every compiled variant of the per-segment pass; records == the byte-compare stager's"""
    code = ("import os, sys; os.environ['ISX_PLANES_VARIANT'] = '%d'; sys.path.insert(0, %r); import tests.test_planes_host as t; t._guard_page_body()" % (variant, REPO))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


@pytest.mark.parametrize("variant", [None, 0, 1, 2])
def test_planes_with_mm_levels_equal_encode_delta(variant):
    """mm profiling on (round 6): the pairs' levels (isx_read_planes.mm) ride in the records' headers and the marked non-ACGT columns become
    exceptions at skipped columns -- the XOR stager writes exactly the records the byte-compare stager makes of the segments (with their mm
    and their code-5 columns), through every compiled variant; with one mm bin the marks are ignored; a level beyond the bins is refused"""
    if variant is not None:
        code = ("import os, sys; os.environ['ISX_PLANES_VARIANT'] = '%d'; sys.path.insert(0, %r); import tests.test_planes_host as t; "
                "t.test_planes_with_mm_levels_equal_encode_delta(None)" % (variant, REPO))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    rng = np.random.default_rng(17)
    w, ref = _mutated_workload(seed=41, G=90_000, cov=9, err=0.012)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    codes = engine.unpack_codes(segs.bases)
    inside = np.arange(150)[None, :] < segs.len[:, None]
    n5 = inside & (rng.random(codes.shape) < 0.003)
    n5[rng.random(segs.n_seg) < 0.8] = False                         # most lines carry no mark (the stager's common path)
    codes = np.where(n5, 5, np.where(inside, codes, 4)).astype(np.uint8)
    lvl = rng.integers(0, 19, segs.n_seg).astype(np.uint8)
    segs = engine.SegBatch(segs.gpos, segs.len, engine.pack_codes(codes), lvl, segs.pair)
    rp = engine.RefPlanes.from_codes(ref, threads=2)
    pb = engine.PlaneBatch.from_segs(segs, threads=2)
    assert pb.mm is not None and ((pb.planes[:, 7] >> np.uint64(63)) == n5.any(axis=1)).all() and 0 < n5.any(axis=1).mean() < 0.5
    for threads in (1, 3):
        a = engine.encode_delta(segs, ref, n_mm_bins=19, threads=threads)
        b = engine.encode_planes(pb, rp, threads=threads, n_mm_bins=19)
        assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[3] == b[3]
    g, ln, mm, cd, pr, full = engine.decode_delta(b[0], b[1], ref)
    assert (cd == 5).sum() == n5.sum() and mm.max() == 18
    # through the staging ring
    r2 = engine.encode_planes(pb, rp, threads=2, n_mm_bins=19, slack_groups=b[3], ring_records=2 * 32768)
    assert (r2[0] == b[0]).all() and (r2[1] == b[1]).all()
    # one mm bin: the marks and the levels are ignored -- the records of the same segments with code 5 read as "not observed", level 0
    plain = engine.SegBatch(segs.gpos, segs.len, engine.pack_codes(np.where(codes == 5, 4, codes)), None, segs.pair)
    a1 = engine.encode_delta(plain, ref, threads=2)
    b1 = engine.encode_planes(engine.PlaneBatch(pb.gpos, pb.len, pb.planes, pb.pair), rp, threads=2)
    assert (a1[0] == b1[0]).all() and (a1[1] == b1[1]).all()
    with pytest.raises(IsxError, match="mm >= n_mm_bins"):
        engine.encode_planes(pb, rp, threads=2, n_mm_bins=18)
