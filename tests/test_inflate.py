"""BGZF blocks through the library's own deflate decoder (csrc/isx_inflate.hip): pinned against zlib on the host (no GPU), then the
same blocks through the device kernel (-m gpu).  What it stands in for: htslib's bgzf.c + zlib under pysam (filter_reads.py:885-956)."""
import os
import struct
import zlib

import numpy as np
import pytest

from instrain_amd import engine
from instrain_amd._lib import IsxError
from tests import util


def bgzf_block(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(body) + 8
    assert bsize <= 65536
    hdr = struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize - 1)
    return hdr + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def _payloads(seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = [b"", b"A", b"ACGT" * 4000, bytes(rng.integers(0, 256, 60000, dtype=np.uint8)),          # empty, tiny, repetitive, incompressible
           bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 65000, p=[.3, .2, .2, .29, .01])),
           bytes(rng.integers(0, 4, 65280, dtype=np.uint8)), b"\x00" * 65280,
           bytes(np.repeat(rng.integers(0, 256, 300, dtype=np.uint8), rng.integers(1, 300, 300))[:65000])]
    # BAM-like: records with names, flags, packed bases, qualities
    rec = bytearray()
    for i in range(400):
        rec += struct.pack("<iiiBBHHHIiii", 120 + i % 7, i % 3, 1000 + 37 * i, 12, 40, 4681, 1, 99, 150, i % 3, 1300 + 37 * i, 350)
        rec += b"read%07d\x00" % i + bytes(rng.integers(0, 256, 75, dtype=np.uint8)) + bytes(rng.choice([37, 25, 12], 150, p=[.9, .08, .02]).astype(np.uint8))
    out.append(bytes(rec)[:65000])
    return out


def _image():
    blocks, plain = [], []
    for k, p in enumerate(_payloads()):
        for level, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                             (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(p) > 65000:
                p_ = p[:65000]
            else:
                p_ = p
            blocks.append(bgzf_block(p_, level, strat))
            plain.append(p_)
    blocks.append(bgzf_block(b""))                      # the EOF marker block
    plain.append(b"")
    return b"".join(blocks), b"".join(plain), len(blocks)


def test_index_and_host_decoder_against_zlib():
    """stored, fixed and dynamic blocks, matches at every distance class, empty blocks: byte for byte what zlib gives"""
    img, plain, n = _image()
    blocks, total = engine.bgzf_index(img)
    assert len(blocks) == n and total == len(plain)
    assert (blocks["out_off"] == np.cumsum(blocks["out_len"]) - blocks["out_len"]).all()
    out, _ = engine.bgzf_inflate(img)
    assert out.tobytes() == plain
    out, _ = engine.bgzf_inflate(img, fast=True)        # the front end's table-driven decoder (fast_inflate.h), no zlib fallback
    assert out.tobytes() == plain
    # a subset of the blocks, re-based
    sel = blocks[5:17]
    out, _ = engine.bgzf_inflate(img, sel)
    lo, hi = int(sel["out_off"][0]), int(sel["out_off"][-1] + sel["out_len"][-1])
    assert out.tobytes() == plain[lo:hi]


@pytest.mark.parametrize("name", ["sars_cov_2.sorted.bam", "SmallScaffold.fa.sorted.bam", "filter_modes.bam"])
def test_host_decoder_on_the_golden_bams(name):
    raw = open(os.path.join(util.GOLD, name), "rb").read()
    out, _ = engine.bgzf_inflate(raw)
    fast, _ = engine.bgzf_inflate(raw, fast=True)
    assert fast.tobytes() == out.tobytes()
    d = zlib.decompressobj(31)
    exp = b""
    data = raw
    while data:                                         # concatenated gzip members
        exp += d.decompress(data)
        data = d.unused_data
        d = zlib.decompressobj(31)
    assert out.tobytes() == exp and out[:4].tobytes() == b"BAM\x01"


def test_corrupt_blocks_are_refused():
    img, plain, n = _image()
    blocks, total = engine.bgzf_index(img)
    bad = bytearray(img)
    k = int(np.flatnonzero(blocks["in_len"] > 2000)[3])
    bad[int(blocks["in_off"][k]) + 700] ^= 0x55         # inside a compressed stream: some code breaks or the size comes out wrong
    with pytest.raises(IsxError, match="BGZF block"):
        out, _ = engine.bgzf_inflate(bytes(bad))
        assert out.tobytes() != plain                   # (a flipped literal can survive: then the bytes differ and nothing raises)
        raise IsxError(-1, "BGZF block: differs")
    with pytest.raises(IsxError, match="not a BGZF"):
        engine.bgzf_index(b"\x00" * 64)
    with pytest.raises(IsxError, match="corrupt"):
        engine.bgzf_index(img[:-9])


@pytest.mark.gpu
def test_device_inflate_equals_zlib():
    """the same images through k_bgzf_inflate: one lane per block"""
    ctx = engine.Context(0)
    img, plain, n = _image()
    out, ms = engine.bgzf_inflate(img, ctx=ctx)
    assert out.tobytes() == plain and ms > 0
    for name in ("sars_cov_2.sorted.bam", "SmallScaffold.fa.sorted.bam", "filter_modes.bam"):
        raw = open(os.path.join(util.GOLD, name), "rb").read()
        host, _ = engine.bgzf_inflate(raw)
        dev, _ = engine.bgzf_inflate(raw, ctx=ctx)
        assert dev.tobytes() == host.tobytes()
    bad = bytearray(img)
    blocks, _ = engine.bgzf_index(img)
    k = int(np.flatnonzero(blocks["in_len"] > 2000)[3])
    bad[int(blocks["in_off"][k]) + 2] ^= 0xFF
    try:
        dev, _ = engine.bgzf_inflate(bytes(bad), ctx=ctx)
        assert dev.tobytes() != plain
    except IsxError as e:
        assert "BGZF block" in str(e)
    ctx.close()


def test_fast_decoder_long_codes_and_random_streams():
    """skewed symbol statistics give codes of 12-15 bits (the second-level tables), many small blocks exercise every table rebuild;
    truncated or flipped streams must be refused or at least never write beyond their block"""
    rng = np.random.Generator(np.random.PCG64(11))
    blocks, plain = [], []
    for k in range(60):
        n = int(rng.integers(1, 60000))
        p = rng.geometric(0.02 + 0.3 * rng.random(), n).clip(0, 255).astype(np.uint8)       # a long tail of rare byte values
        if k % 3 == 0:
            p[rng.integers(0, n, n // 50)] = rng.integers(0, 256, n // 50, dtype=np.uint8)
        if k % 4 == 1:                                                                      # far matches: distances up to 32 K
            p = np.concatenate([p[:20000], p[:20000], p[:5000]])[:65000]
        blocks.append(bgzf_block(p.tobytes(), int(rng.choice([1, 4, 6, 9]))))
        plain.append(p.tobytes())
    img, plain = b"".join(blocks), b"".join(plain)
    for fast in (False, True):
        out, _ = engine.bgzf_inflate(img, fast=fast)
        assert out.tobytes() == plain, fast
    idx, _ = engine.bgzf_index(img)
    guard = np.full(len(plain) + 64, 0xA5, np.uint8)
    for k in (3, 17, 40):
        bad = bytearray(img)
        bad[int(idx["in_off"][k]) + int(idx["in_len"][k]) // 2] ^= 0x10
        try:
            out, _ = engine.bgzf_inflate(bytes(bad), fast=True)
            assert out.tobytes() != plain
        except IsxError:
            pass


def test_fast_decoder_never_writes_outside_its_block_on_corrupt_input():
    """300 random corruptions (bit flips, truncations, wrong ISIZE) of real blocks: the decoder refuses or decodes something, and the bytes
    before and behind the block's output stay untouched either way"""
    import ctypes as C
    from instrain_amd import _lib
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(23))
    raw = open(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"), "rb").read()
    blocks, _ = engine.bgzf_index(raw)
    blocks = blocks[blocks["out_len"] > 4000]
    assert len(blocks) >= 4
    GUARD = 4096
    n_refused = 0
    for it in range(300):
        b = blocks[int(rng.integers(0, len(blocks)))].copy()
        img = np.frombuffer(raw, dtype=np.uint8).copy()
        lo, n = int(b["in_off"]), int(b["in_len"])
        kind = it % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                img[lo + int(rng.integers(0, n))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            b["in_len"] = int(rng.integers(1, n))                       # truncated stream
        else:
            b["out_len"] = int(rng.integers(1, 65536))                  # a wrong ISIZE
        one = np.zeros(1, dtype=_lib.BGZF_BLOCK_DT)
        one[0] = b
        one["out_off"] = GUARD
        out = np.full(GUARD + int(b["out_len"]) + GUARD, 0xA5, dtype=np.uint8)
        rc = lib.isx_bgzf_inflate_fast(img.ctypes.data, len(img), one.ctypes.data, 1, out.ctypes.data, len(out))
        n_refused += rc != 0
        assert (out[:GUARD] == 0xA5).all() and (out[GUARD + int(b["out_len"]):] == 0xA5).all(), (it, kind)
    assert n_refused > 150
