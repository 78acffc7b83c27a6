"""CPU-only: the pipe's host-side encoder (isx_encode_obs) -- isx_obs -> 2- / 4-byte resident records.
Decoding the stream must give back every observation in arrival order (padding aside), groups never
span the delta range, and the multi-threaded / slack / two-pass layouts all decode to the same input."""
import numpy as np
import pytest

from instrain_amd import engine
from instrain_amd._lib import OBS_DT


def decode(rec, gbase, pair_out, record_bytes):
    G = 512 if record_bytes == 2 else 256
    gb = np.repeat(gbase.astype(np.int64), G)
    if record_bytes == 2:
        real = rec != 0xFFFF
        gpos = gb + (rec & 0x1FFF)
        base = rec >> 13
        mm = np.zeros(len(rec), np.int64)
    else:
        real = rec != 0x0700FFFF
        gpos = gb + (rec & 0xFFFF)
        mm = (rec >> 16) & 0xFF
        base = (rec >> 24) & 7
    out = {"gpos": gpos[real], "base": base[real], "mm": mm[real]}
    if pair_out is not None:
        out["pair"] = pair_out[real]
        assert (pair_out[~real] == 0).all()
    # a group never spans the delta range, and real records sit at the front of their group
    r2 = real.reshape(-1, G)
    n_real = r2.sum(axis=1)
    assert (r2 == (np.arange(G)[None, :] < n_real[:, None])).all()
    return out, n_real


def make_stream(rng, n_reads, read_len, islands, gap, mm_max):
    starts = np.sort(rng.integers(0, 3000, n_reads))
    isl = rng.integers(0, islands, n_reads)
    isl.sort()
    starts = starts + isl * gap
    order = np.argsort(starts, kind="stable")
    starts = starts[order]
    gpos = (starts[:, None] + np.arange(read_len)[None, :]).reshape(-1)
    obs = np.zeros(len(gpos), dtype=OBS_DT)
    obs["gpos"] = gpos
    obs["base"] = rng.integers(0, 6, len(gpos))          # 5 -> clamped to 4
    obs["mm"] = np.repeat(rng.integers(0, mm_max + 1, n_reads), read_len)
    pair = np.repeat(rng.integers(0, 1 << 20, n_reads), read_len).astype(np.uint32)
    return obs, pair


@pytest.mark.parametrize("rb", [2, 4])
@pytest.mark.parametrize("islands,gap", [(1, 0), (5, 9_000), (7, 70_000), (40, 200_000)])
def test_roundtrip(rb, islands, gap):
    rng = np.random.default_rng(7 + islands)
    obs, pair = make_stream(rng, 4000, 100, islands, gap, 0 if rb == 2 else 200)
    span = 8191 if rb == 2 else 65535
    ref = None
    for threads, slack in ((1, 0.0), (4, 0.0), (3, 0.5), (8, 0.05)):
        rec, gbase, pout, passes = engine.encode_obs(obs, pair, record_bytes=rb, threads=threads, slack=slack)
        assert len(rec) % 2048 == 0
        got, n_real = decode(rec, gbase, pout, rb)
        np.testing.assert_array_equal(got["gpos"], obs["gpos"])
        np.testing.assert_array_equal(got["base"], np.minimum(obs["base"], 4))
        if rb == 4:
            np.testing.assert_array_equal(got["mm"], obs["mm"])
        np.testing.assert_array_equal(got["pair"], pair)
        G = 512 if rb == 2 else 256
        d = (rec & (0x1FFF if rb == 2 else 0xFFFF)).reshape(-1, G)
        real = np.arange(G)[None, :] < n_real[:, None]
        assert (np.where(real, d, 0).max(axis=1) < span).all()
        jumps = gap >= span
        if slack == 0.0:                    # no slack: jumps force the second, exact layout; none -> minimal identity layout
            assert passes == (2 if jumps else 1)
            if not jumps:
                assert len(rec) == max(2048, (len(obs) + 2047) // 2048 * 2048)
        if ref is None:
            ref = got


def test_empty_and_tiny():
    rec, gbase, pout, passes = engine.encode_obs(np.zeros(0, dtype=OBS_DT), None, n_pos=10, record_bytes=2)
    assert len(rec) == 2048 and (rec == 0xFFFF).all() and passes == 1
    obs = np.zeros(3, dtype=OBS_DT)
    obs["gpos"] = [5, 6, 7]
    obs["base"] = [0, 3, 9]
    rec, gbase, _, _ = engine.encode_obs(obs, None, record_bytes=2)
    assert list(rec[:3]) == [0 | (0 << 13), 1 | (3 << 13), 2 | (4 << 13)] and gbase[0] == 5 and (rec[3:] == 0xFFFF).all()


def test_errors():
    from instrain_amd._lib import IsxError
    obs = np.zeros(600, dtype=OBS_DT)
    obs["gpos"] = np.arange(600)
    with pytest.raises(IsxError):
        engine.encode_obs(obs, None, n_pos=100, record_bytes=2)           # gpos >= n_pos
    obs["mm"] = 300
    with pytest.raises(IsxError):
        engine.encode_obs(obs, None, n_pos=1000, record_bytes=4)          # mm >= 256
    obs["mm"] = 0
    obs["gpos"] = np.arange(600) * 10_000                                 # every record its own group
    with pytest.raises(IsxError):
        engine.encode_obs(obs, None, n_pos=10_000_000, record_bytes=2, cap_rec=4096)


@pytest.mark.parametrize("rb", [2, 4])
@pytest.mark.parametrize("islands,gap", [(1, 0), (7, 70_000), (40, 200_000)])
def test_ring_mode_matches_whole_arena(rb, islands, gap):
    """the pipe's staging ring (waves of at most half a ring, each copied to its place once complete): the stream that
    arrives is the one the whole-arena encoder writes for the same layout, whatever the ring size / thread count"""
    rng = np.random.default_rng(70 + islands)
    obs, _ = make_stream(rng, 6000, 100, islands, gap, 0 if rb == 2 else 200)
    for threads, slack, ring in ((1, 0.0, 8192), (4, 0.3, 16384), (3, 0.0, 4 * 65536), (8, 0.05, 8192)):
        rec, gbase, _, passes = engine.encode_obs(obs, None, record_bytes=rb, threads=threads, slack=slack, ring_records=ring)
        assert len(rec) % 2048 == 0
        assert not (rec == (0xABAB if rb == 2 else 0xABABABAB)).any()         # nothing stale / unwritten travelled
        got, n_real = decode(rec, gbase, None, rb)
        np.testing.assert_array_equal(got["gpos"], obs["gpos"])
        np.testing.assert_array_equal(got["base"], np.minimum(obs["base"], 4))
        if rb == 4:
            np.testing.assert_array_equal(got["mm"], obs["mm"])


def test_ring_mode_every_record_its_own_group():
    """a stream that jumps at every record expands 512x: regions outgrow a ring half, the encoder re-cuts its tasks"""
    obs = np.zeros(3000, dtype=OBS_DT)
    obs["gpos"] = np.arange(3000) * 10_000
    rec, gbase, _, passes = engine.encode_obs(obs, None, n_pos=30_000_000, record_bytes=2, threads=3,
                                              cap_rec=3000 * 512 + 8192, ring_records=1 << 21)
    got, n_real = decode(rec, gbase, None, 2)
    np.testing.assert_array_equal(got["gpos"], obs["gpos"])
    assert passes >= 2
    rec, gbase, _, _ = engine.encode_obs(np.zeros(0, dtype=OBS_DT), None, n_pos=10, record_bytes=4, ring_records=8192)
    assert len(rec) == 2048 and (rec == 0x0700FFFF).all()
