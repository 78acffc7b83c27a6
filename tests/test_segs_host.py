"""Host side of the read-level hand-over (no GPU): observation stream <-> read segments, the staging encoder
(isx_encode_segs) and the read packer (isx_pack_reads).  The segments must stand for exactly the observations the
reference's pileup loop visits (profile_utilities.py:150-153, 268-286)."""
import os

import numpy as np
import pytest

from instrain_amd import engine, synth
from tests import util


def _workload(seed=7, G=60_000, cov=12, skip_mm=False, **kw):
    return synth.make_workload(genome_len=G, coverage=cov, n_sites=200, seed=seed, skip_mm=skip_mm, **kw)


def test_segments_of_a_read_major_stream_round_trip():
    w = _workload()
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    # one segment per read: 2 x 150 bp reads whose kept bases span at most 150 columns
    assert segs.n_seg <= 2 * w["n_pairs"] and segs.n_seg >= 2 * w["n_pairs"] - 4
    g, b, m, p = util.segs_to_obs(segs)
    assert (g == w["obs"]["gpos"]).all() and (b == w["obs"]["base"]).all()
    assert (m == w["obs"]["mm"]).all() and (p == w["pair"]).all()
    assert int(segs.len.max()) <= 150 and int(segs.len.min()) >= 1
    # unused slots hold code 4, the top two bits of every word are clear
    assert (segs.bases >> 30 == 0).all()
    cd = engine.unpack_codes(segs.bases)
    tail = np.arange(150)[None, :] >= segs.len[:, None]
    assert (cd[tail] == 4).all()


def test_segments_of_a_column_major_stream_keep_arrival_order():
    g = util.load_case("synth_selfpairs")
    pos = g["pos"].astype(np.int64) - int(g["start"])
    sel = (pos >= 0) & (pos < len(str(g["seq"])))
    for make in (lambda: util.reassemble_segs(pos[sel], g["base"][sel], g["mm"][sel], g["pair"][sel]),
                 lambda: synth.segs_from_obs(engine.pack_obs(pos[sel].astype(np.uint32), g["base"][sel], g["mm"][sel]), g["pair"][sel].astype(np.uint32))):
        segs = make()
        gg, bb, mm, pp = util.segs_to_obs(segs)
        a = np.lexsort((bb, mm, pp, gg))
        e = np.lexsort((np.minimum(g["base"][sel], 4), g["mm"][sel], g["pair"][sel], pos[sel]))
        assert (gg[a] == pos[sel][e]).all() and (bb[a] == np.minimum(g["base"][sel], 4)[e]).all()
        assert (mm[a] == g["mm"][sel][e]).all() and (pp[a] == g["pair"][sel][e]).all()
        # two observations of one pair at one site: the segment created first holds the one that arrived first
        first_seen = {}
        order_ok = True
        seg_of = np.repeat(np.arange(segs.n_seg), ((engine.unpack_codes(segs.bases) < 4) | (engine.unpack_codes(segs.bases) == 5)).sum(axis=1))
        arrival = {}
        for i, (q, pr, b) in enumerate(zip(pos[sel], g["pair"][sel], g["base"][sel])):
            arrival.setdefault((int(q), int(pr)), []).append(min(int(b), 4))
        got = {}
        for q, pr, b, s in zip(gg, pp, bb, seg_of):
            got.setdefault((int(q), int(pr)), []).append((int(s), int(b)))
        for k, v in got.items():
            assert [b for _, b in sorted(v)] == arrival[k], k


def test_reassembled_segments_are_whole_reads():
    w = _workload(seed=9, G=20_000, cov=8)
    o = w["obs"]
    # shuffle into column-major order (stable within a column = arrival order)
    k = np.argsort(o["gpos"], kind="stable")
    segs = util.reassemble_segs(o["gpos"][k], o["base"][k], o["mm"][k], w["pair"][k])
    assert segs.n_seg <= 2 * w["n_pairs"]
    gg, bb, mm, pp = util.segs_to_obs(segs)
    a, e = np.lexsort((pp, gg)), np.lexsort((w["pair"], o["gpos"]))
    assert (gg[a] == o["gpos"][e]).all() and (bb[a] == o["base"][e]).all() and (pp[a] == w["pair"][e]).all()


@pytest.mark.parametrize("threads", [1, 4])
def test_encode_segs_layout(threads):
    w = _workload(seed=11, G=400_000, cov=6)
    # a jump of > 65535 positions in the middle of the stream: the group there is closed early and padded
    o = w["obs"].copy()
    o["gpos"][o["gpos"] >= 200_000] += 300_000
    segs = synth.segs_from_obs(o, w["pair"])
    n_pos = 400_000 + 300_000
    rec, gbase, pout = engine.encode_segs(segs, n_pos, n_mm_bins=int(o["mm"].max()) + 1, threads=threads)
    assert len(rec) % 16 == 0 and len(gbase) == len(rec) // 16
    g, ln, mm, cd = engine.decode_segs(rec, gbase)
    assert (g == segs.gpos).all() and (ln == segs.len).all() and (mm == segs.mm).all()
    assert (engine.pack_codes(cd) == segs.bases).all()
    hdr = rec[:, 0]
    real = ((hdr >> 16) & 0xFF) > 0
    assert (pout[real] == segs.pair).all() and (pout[~real] == 0).all()
    assert (rec[~real, 1:] == 0x24924924).all() and (hdr[~real] == 0).all()
    # every group's deltas fit 16 bits by construction; the jump costs padding
    assert real.sum() == segs.n_seg and (~real).sum() >= 1
    # the number of groups does not depend on the thread count beyond the per-task rounding
    assert len(rec) <= (segs.n_seg // 16 + segs.n_seg // 4096 + 64) * 16       # reads straddling the jump alternate between its sides


@pytest.mark.parametrize("ring_records", [2 * 4096 + 64, 3 * 4096, 1 << 14])
def test_encode_segs_through_the_staging_ring(ring_records):
    """waves through a ring of two halves give the stream the whole-arena layout gives, whatever the half size"""
    w = _workload(seed=13, G=300_000, cov=10)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    a = engine.encode_segs(segs, w["n_pos"], n_mm_bins=int(w["obs"]["mm"].max()) + 1, threads=3)
    b = engine.encode_segs(segs, w["n_pos"], n_mm_bins=int(w["obs"]["mm"].max()) + 1, threads=3, ring_records=ring_records)
    assert len(a[0]) == len(b[0]) > ring_records // 2
    for x, y in zip(a, b):
        assert (x == y).all()
    from instrain_amd._lib import IsxError
    with pytest.raises(IsxError):                       # a half that cannot hold one task of 4096 segments
        engine.encode_segs(segs, w["n_pos"], n_mm_bins=16, ring_records=2 * 2048)


def test_encode_segs_sparse_stream_is_sized_exactly():
    """low coverage: starts 5 000 positions apart close a group every 14 segments (the 65 535 span), without any jump of >= 32 768
    -- the default capacity comes from the encoder's own counting pass (the old estimate gave ISX_ERR_CAPACITY)"""
    n = 8192
    gpos = (np.arange(n, dtype=np.uint32) * 5000).astype(np.uint32)
    segs = engine.SegBatch(gpos, np.full(n, 150, np.uint8), np.zeros((n, 15), np.uint32))
    n_pos = int(gpos[-1]) + 150
    need = engine._lib.load().isx_seg_records_needed(segs.gpos.ctypes.data, n, 2)
    assert need > (n // 16 + n // 4096 + 1) * 16            # more than the old estimate
    rec, gbase, _ = engine.encode_segs(segs, n_pos)
    assert len(rec) == need
    g, ln, _, _ = engine.decode_segs(rec, gbase)
    assert (g == gpos).all() and (ln == 150).all()
    from instrain_amd._lib import IsxError
    with pytest.raises(IsxError):
        engine.encode_segs(segs, n_pos, cap_rec=need - 16)


def _pieces_to_columns(g, ln, cd, n_pos):
    """pieces -> per-position multiset signature: (position, code) pairs of every observed column, sorted"""
    j = np.arange(150)[None, :]
    ok = (j < ln[:, None]) & (cd < 4)
    pos = (g[:, None].astype(np.int64) + j)[ok]
    return np.sort(pos * 8 + cd[ok])


def _mutated_workload(seed, G=120_000, cov=6, err=0.05, n_frac=0.01):
    """reads with MANY mismatches (5 %: ~7 per read -> pieces) over a reference with non-ACGT stretches"""
    rng = np.random.default_rng(seed)
    w = synth.make_workload(genome_len=G, coverage=cov, n_sites=G // 50, err=err, seed=seed, skip_mm=True, p_keep=0.995)
    ref = w["ref_codes"].copy()
    ref[rng.random(G) < n_frac] = 4
    ref[1000:1400] = 4
    return w, ref


@pytest.mark.parametrize("threads,vbmi", [(1, True), (3, True), (2, False)])
def test_encode_delta_round_trip(threads, vbmi):
    """reference-delta records decode to exactly the observations the segments stand for: plain reads (one piece each), reads
    with more than six differences (several pieces), a reference with non-ACGT positions (every base there is an exception);
    the AVX-512 VBMI and the scalar compare agree (the scalar path runs in a subprocess: the choice is made once per process)"""
    if not vbmi:
        import subprocess
        import sys
        code = ("import os, sys; os.environ['ISX_NO_VBMI'] = '1'; sys.path.insert(0, %r); import tests.test_segs_host as t; "
                "t.test_encode_delta_round_trip(2, True)" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    w = _workload(seed=21, G=300_000, cov=8, skip_mm=True, p_keep=0.995)      # about half of the reads have no base below the quality bar
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    rec, gbase, _, slack = engine.encode_delta(segs, w["ref_codes"], threads=threads)
    assert len(rec) % 32 == 0 and len(gbase) == len(rec) // 32
    g, ln, mm, cd, pr, full = engine.decode_delta(rec, gbase, w["ref_codes"])
    exp = engine.unpack_codes(segs.bases)
    exp = np.where((np.arange(150)[None, :] < segs.len[:, None]) & (exp < 4), exp, 4)
    assert len(g) == segs.n_seg and (g == segs.gpos).all() and (ln == segs.len).all() and (cd == exp).all() and (pr == segs.pair).all()
    # a segment without a skipped column is half of a dual record (16 bytes a read), one with skipped columns a full record
    has_skip = ((exp == 4) & (np.arange(150)[None, :] < segs.len[:, None])).any(axis=1)
    assert (full == has_skip).all() and 0.2 < has_skip.mean() < 0.8
    n_dual = int((rec[:, 0] >> 31).sum())
    assert n_dual <= ((~has_skip).sum() + 1) // 2 + int(has_skip.sum()) + len(gbase)        # halves are filled wherever the order allows
    pad = (((rec[:, 0] >> 16) & 0xFF) == 0) & ((rec[:, 0] >> 31) == 0)
    assert (rec[pad][:, [0, 1, 2, 4, 5, 6, 7]] == 0).all() and (rec[pad][:, 3] == 0x3FFFFFFF).all()
    # many mismatches + non-ACGT reference: pieces
    w, ref = _mutated_workload(seed=22)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    rec, gbase, _, slack = engine.encode_delta(segs, ref, threads=threads)
    g, ln, mm, cd, pr, full = engine.decode_delta(rec, gbase, ref)
    assert len(g) > segs.n_seg * 1.2 and slack > 1                 # pieces, and the first attempt's single spare group was not enough
    gg, bb, _, _ = util.segs_to_obs(segs)
    assert (_pieces_to_columns(g, ln, cd, len(ref)) == np.sort(gg * 8 + bb)).all()
    # a piece never carries more exceptions than its kind holds: six in a dual half, three in a full record
    ref_at = ref[np.minimum(g[:, None].astype(np.int64) + np.arange(150)[None, :], len(ref) - 1)]
    n_exc = ((cd < 4) & (cd != ref_at) & (np.arange(150)[None, :] < ln[:, None])).sum(axis=1)
    assert n_exc[~full].max() == 6 and n_exc[full].max() == 3
    # pieces keep their segment's pair id and the stream's order
    first = np.r_[True, (pr[1:] != pr[:-1]) | (g[1:].astype(np.int64) != g[:-1].astype(np.int64) + ln[:-1])]        # a piece continues where the one before it ends
    starts = g[first]
    assert len(starts) == segs.n_seg and (starts == segs.gpos).all() and (pr[first] == segs.pair).all()
    # through the staging ring: the same stream
    r2 = engine.encode_delta(segs, ref, threads=threads, slack_groups=slack, ring_records=2 * 32768)
    assert (r2[0] == rec).all() and (r2[1] == gbase).all()
    from instrain_amd._lib import IsxError
    with pytest.raises(IsxError):
        engine.encode_delta(segs, ref, slack_groups=1, retry=False)


def test_encode_delta_carries_the_mm_level_and_non_acgt_bases():
    """mm profiling on (round 6): a segment's header carries its pair's mm level (bits 24..30), and a base that is not A/C/T/G but
    passes the quality filter (code 5: it makes its level present, profile_utilities.py:279-285) travels as an exception at a skipped
    column -- the records decode to the segments' codes, 5 included, levels included; with one mm bin code 5 is only a skipped column"""
    rng = np.random.default_rng(5)
    w, ref = _mutated_workload(seed=31, G=60_000, cov=10, err=0.01)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    codes = engine.unpack_codes(segs.bases)
    inside = np.arange(150)[None, :] < segs.len[:, None]
    n5 = inside & (rng.random(codes.shape) < 0.004)                 # sprinkle non-ACGT bases, also over skipped columns and mismatches
    codes = np.where(n5, 5, np.where(inside, codes, 4)).astype(np.uint8)
    lvl = rng.integers(0, 23, segs.n_seg).astype(np.uint8)
    segs = engine.SegBatch(segs.gpos, segs.len, engine.pack_codes(codes), lvl, segs.pair)
    for M, threads in ((23, 1), (23, 3), (1, 2)):
        sg = segs if M > 1 else engine.SegBatch(segs.gpos, segs.len, segs.bases, np.zeros(segs.n_seg, np.uint8), segs.pair)
        rec, gbase, _, slack = engine.encode_delta(sg, ref, n_mm_bins=M, threads=threads)
        g, ln, mm, cd, pr, full = engine.decode_delta(rec, gbase, ref)
        first = np.r_[True, (pr[1:] != pr[:-1]) | (g[1:].astype(np.int64) != g[:-1].astype(np.int64) + ln[:-1])]
        seg_of = np.cumsum(first) - 1
        assert seg_of[-1] + 1 == segs.n_seg and (g[first] == segs.gpos).all()
        assert (mm == (lvl[seg_of] if M > 1 else 0)).all()                                        # every piece carries its segment's level
        # piece columns -> segment columns
        off = (g.astype(np.int64) - segs.gpos[seg_of].astype(np.int64))
        got = np.full(codes.shape, 4, np.uint8)
        j = np.arange(150)[None, :]
        ok = j < ln[:, None]
        rows = np.broadcast_to(seg_of[:, None], ok.shape)[ok]
        cols = (off[:, None] + j)[ok]
        got[rows, cols] = cd[ok]
        want = codes if M > 1 else np.where(codes == 5, 4, codes)
        assert (got == want).all()
        assert (M > 1) == bool((cd == 5).any())
        assert full[(cd == 5).any(axis=1)].all()                                                # a marker needs the skip plane: a full record
    with pytest.raises(engine.IsxError, match="mm >= n_mm_bins"):
        engine.encode_delta(segs, ref, n_mm_bins=22)


def test_encode_segs_rejects_bad_input():
    from instrain_amd._lib import IsxError
    segs = engine.SegBatch([10, 20], [150, 150], np.full((2, 15), 0x24924924, np.uint32), mm=[0, 3])
    with pytest.raises(IsxError, match="beyond n_pos"):
        engine.encode_segs(segs, 100, n_mm_bins=4)
    with pytest.raises(IsxError, match="mm >= n_mm_bins"):
        engine.encode_segs(segs, 1000, n_mm_bins=2)
    bad = engine.SegBatch([10], [0], np.full((1, 15), 0x24924924, np.uint32))
    with pytest.raises(IsxError, match="length"):
        engine.encode_segs(bad, 1000)
    empty = engine.SegBatch(np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros((0, 15), np.uint32))
    rec, gbase, _ = engine.encode_segs(empty, 1000)
    assert len(rec) == 16 and (rec[:, 0] == 0).all()


def _cig(*ops):
    code = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    return np.asarray([(n << 4) | code[o] for n, o in ops], dtype=np.uint32)


def test_pack_reads_follows_the_cigar_and_the_quality_filter():
    rng = np.random.Generator(np.random.PCG64(3))
    L = 400
    seq = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=L))
    qual = rng.choice([12, 25, 30, 37], size=L).astype(np.uint8)
    # 5S 100M 3I 40= 7D 20X 2N 180M 50S: query = 5 + 100 + 3 + 40 + 20 + 180 + 50 = 398
    cig = _cig((5, "S"), (100, "M"), (3, "I"), (40, "="), (7, "D"), (20, "X"), (2, "N"), (180, "M"), (50, "S"))
    seq, qual = seq[:398], qual[:398]
    start = 1000
    segs = engine.pack_reads([start], [0], [10_000], [cig], [seq], [qual], mm=[2], pair=[7])
    # expected observations by a plain walk
    exp = []
    ref, q = start, 0
    for n, op in ((5, "S"), (100, "M"), (3, "I"), (40, "="), (7, "D"), (20, "X"), (2, "N"), (180, "M"), (50, "S")):
        if op in "M=X":
            for j in range(n):
                if qual[q + j] >= 30:
                    exp.append((ref + j, "ACTG".find(seq[q + j]) if seq[q + j] in "ACTG" else 4))
            q += n; ref += n
        elif op in "IS":
            q += n
        elif op in "DN":
            ref += n
    g, b, m, p = util.segs_to_obs(segs)
    assert list(zip(g.tolist(), b.tolist())) == exp
    assert (m == 2).all() and (p == 7).all()
    # runs: 100 | 40 | 20 | 180 -> 150 + 30: five segments, none longer than 150
    assert segs.n_seg == 5 and segs.len.tolist() == [100, 40, 20, 150, 30]
    # truncation to the scaffold: columns outside [clip_lo, clip_hi) are dropped like the reference's truncate=True
    clipped = engine.pack_reads([start], [1050], [1300], [cig], [seq], [qual])
    gc, bc, _, _ = util.segs_to_obs(clipped)
    assert list(zip(gc.tolist(), bc.tolist())) == [e for e in exp if 1050 <= e[0] < 1300]


def _same_stream(segs, obs, pair):
    g, b, m, p = util.segs_to_obs(segs)
    assert len(g) == len(obs)
    assert (g == obs["gpos"]).all() and (b == np.minimum(obs["base"], 4)).all() and (m == obs["mm"]).all() and (p == pair).all()


def test_bam_segments_stand_for_the_bam_observations():
    """the front end's read segments (what a read-level pipe is handed) decode to exactly the observation stream the same
    front end expands -- and that stream is pinned against the reference's stored sars_cov_2 run (test_bam_front)"""
    import os
    path = os.path.join(util.GOLD, "sars_cov_2.sorted.bam")
    bam = engine.BamFile(path)
    obs, pair, bounds, sref = bam.expand()
    segs, b2, s2 = bam.segment_refs([0])
    assert bam.info["n_obs"] >= len(obs) and (b2 == bounds).all() and (s2 == sref).all()
    bam.close()
    _same_stream(segs, obs, pair)
    # reads are 2 x ~100-150 bp with indels: a handful of segments per read, all <= 150 columns, most of them full reads
    assert segs.len.max() <= 150 and segs.n_seg < 2.5 * 2 * 13124


@pytest.mark.parametrize("seed,skip_mm", [(1, False), (2, True), (3, False)])
def test_messy_bam_segments_equal_observations(tmp_path, seed, skip_mm):
    """clips, indels, ref skips, = / X, N bases, overlapping mates (qualities rewritten by the overlap resolution before they
    are packed), reads hanging over scaffold ends, several references, subsets of them"""
    from tests import bamwriter
    refs = [("sA", 4000), ("sB", 1500), ("sC", 9000)]
    reads = bamwriter.random_reads(seed, refs, 900)
    path = str(tmp_path / "m.bam")
    bamwriter.write_bam(path, refs, reads)
    for sel in ([0, 1, 2], [1], [0, 2]):
        bam = engine.BamFile(path, threads=3)
        bam.scan()
        bam.filter(min_read_ani=0.8, skip_mm=skip_mm)
        obs, pair, bounds, sref = bam.expand_refs(sel, skip_mm=skip_mm, min_read_ani=0.8)
        segs, b2, s2 = bam.segment_refs(sel, skip_mm=skip_mm, min_read_ani=0.8)
        assert (b2 == bounds).all() and (s2 == sref).all()
        bam.close()
        _same_stream(segs, obs, pair)
