#!/usr/bin/env python3
"""Throughput of the host BAM front end (isx_bam_open / isx_bam_expand): synthetic 2x150 bp pairs,
written with a vectorised BAM writer.  usage: python tools/bench_front.py [n_pairs] [genome_len]"""
import os, struct, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_simple_bam(path, G, n_pairs, read_len=150, seed=1, n_refs=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    ref = rng.integers(0, 4, G, dtype=np.uint8)
    ins = np.maximum(rng.normal(350, 30, n_pairs), 2 * read_len).astype(np.int64)
    s1 = rng.integers(0, G - int(ins.max()) - 1, n_pairs)
    s2 = s1 + ins - read_len
    starts = np.concatenate([s1, s2]); mate = np.concatenate([s2, s1])
    isz = np.concatenate([ins, -ins]); pid = np.concatenate([np.arange(n_pairs)] * 2)
    first = np.concatenate([np.ones(n_pairs, bool), np.zeros(n_pairs, bool)])
    o = np.argsort(starts, kind="stable")
    code4 = np.array([1, 2, 8, 4], dtype=np.uint8)          # A C T G in the ACTG order of `ref` -> BAM nibbles
    L = G // n_refs                  # n_refs > 1: the genome cut into scaffolds scaf0, scaf1, ... (pairs across a cut are dropped)
    names = ["scaf"] if n_refs == 1 else ["scaf%d" % i for i in range(n_refs)]
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, L if n_refs > 1 else G) for n in names)
    out = [b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names)) +
           b"".join(struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", L if n_refs > 1 else G) for n in names)]
    qual = bytes([37]) * read_len
    cig = struct.pack("<I", (read_len << 4) | 0)
    for k in o:
        st = int(starts[k])
        tid, mt = 0, int(mate[k])
        if n_refs > 1:
            tid = st // L
            if tid >= n_refs or (st + read_len - 1) // L != tid or mt // L != tid or (mt + read_len - 1) // L != tid:
                continue
            mt -= tid * L
        b = ref[st:st + read_len].copy()
        e = rng.random(read_len) < 0.002
        b[e] = rng.integers(0, 4, int(e.sum()))
        nm = int((b != ref[st:st + read_len]).sum())
        nib = code4[b]
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        name = b"p%d\0" % pid[k]
        flag = 0x1 | 0x2 | (0x40 if first[k] else 0x80) | (0x20 if first[k] else 0x10)
        body = struct.pack("<iiBBHHHiiii", tid, st - tid * L if n_refs > 1 else st, len(name), 42, 4680, 1, flag, read_len, tid, mt, int(isz[k])) + \
            name + cig + packed + qual + b"NMC" + struct.pack("<B", nm)
        out.append(struct.pack("<i", len(body)) + body)
    blob = b"".join(out)
    with open(path, "wb") as f:
        for i in range(0, len(blob), 60000):
            data = blob[i:i + 60000]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = co.compress(data) + co.flush()
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return len(blob)


if __name__ == "__main__":
    n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
    G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
    path = "/tmp/bench_front_%d.bam" % n_pairs
    if not os.path.exists(path):
        t0 = time.time(); nb = write_simple_bam(path, G, n_pairs); print("wrote", path, nb, "bytes uncompressed in", round(time.time() - t0, 1), "s")
    from instrain_amd import engine
    for rep in range(3):
        t0 = time.perf_counter()
        bf = engine.BamFile(path)
        t1 = time.perf_counter()
        obs, pair, bounds, sref = bf.expand(skip_mm=False, copy=False)
        n_obs = len(obs)
        del obs, pair
        t2 = time.perf_counter()
        bf.close()
        gbp = n_pairs * 300 / 1e9
        print("open %.3f s  expand %.3f s  -> %.3f Gbp/s  (%d obs, %.1f M reads/s)" %
              (t1 - t0, t2 - t1, gbp / (t2 - t0), n_obs, 2 * n_pairs / (t2 - t0) / 1e6), flush=True)
