"""Minimal BGZF/BAM writer for tests: turns a list of synthetic read dicts into a sorted BAM so the
product's C++ front end and the oracle's Python front end can be compared on inputs far messier
than the reference's fixtures (indels, clips, ref-skips, overlapping mates that disagree, odd flags)."""
import struct
import zlib

import numpy as np

SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
CIG_OP = {c: i for i, c in enumerate("MIDNSHP=X")}


def _bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def write_bam(path, refs, reads, cg_every=0):
    """refs: [(name, length)]; reads: dicts with tid,pos,mapq,flag,isize,name,cigar[(op char,len)],seq(str),qual(array),nm
    -- must already be sorted by (tid, pos).  cg_every = k > 0: every k-th mapped read is written the way a CIGAR of more
    than 65535 operations is (SAM spec 4.2.2): the placeholder <l_seq>S<ref_len>N in the record, the operations in CG:B,I."""
    out = bytearray()
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out += b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    for i_read, r in enumerate(reads):
        name = r["name"].encode() + b"\0"
        cig = b"".join(struct.pack("<I", (n << 4) | CIG_OP[op]) for op, n in r["cigar"])
        n_cig = len(r["cigar"])
        cg = b""
        if cg_every and i_read % cg_every == 0 and r["tid"] >= 0 and n_cig:
            ref_len = sum(n for op, n in r["cigar"] if op in "MDN=X")
            cg = b"CGBI" + struct.pack("<i", n_cig) + cig
            cig = struct.pack("<II", (len(r["seq"]) << 4) | CIG_OP["S"], (ref_len << 4) | CIG_OP["N"])
            n_cig = 2
        seq = r["seq"]
        codes = [SEQ_CODE[c] for c in seq] + ([0] if len(seq) % 2 else [])
        packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
        qual = bytes(int(q) for q in r["qual"])
        tags = b"" if r.get("nm") is None else b"NMC" + struct.pack("<B", r["nm"])
        if r.get("extra_tags"):
            tags = b"XSZ" + b"hello\0" + tags + b"ASi" + struct.pack("<i", -7) + b"ZBBS" + struct.pack("<iHH", 2, 1, 2)
        tags = tags + cg if i_read % 2 else cg + tags
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], 4680, n_cig, r["flag"],
                           len(seq), r.get("mtid", r["tid"]), r.get("mpos", 0), r["isize"]) + name + cig + packed + qual + tags
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for i in range(0, len(out), 60000):
            f.write(_bgzf_block(bytes(out[i:i + 60000])))
        f.write(_bgzf_block(b""))


def random_reads(seed, refs, n_pairs, read_len=60):
    """paired reads with random CIGARs (M/I/D/N/S/H/=/X), overlapping and non-overlapping mates,
    quality mixes around the Q30 cut, a few non-ACGT bases, and the flag zoo."""
    rng = np.random.Generator(np.random.PCG64(seed))
    reads = []
    for p in range(n_pairs):
        tid = int(rng.integers(0, len(refs)))
        L = refs[tid][1]
        s1 = int(rng.integers(0, max(1, L - 3 * read_len)))
        overlap = rng.random() < 0.6
        s2 = s1 + (int(rng.integers(0, read_len)) if overlap else read_len + int(rng.integers(0, 2 * read_len)))
        name = "p%d" % p
        proper = rng.random() < 0.9
        mate_unmapped = rng.random() < 0.03
        ends = []
        for mate, s in enumerate((s1, s2)):
            ops = []
            left = read_len
            if rng.random() < 0.15:
                ops.append(("H", int(rng.integers(1, 5))))
            if rng.random() < 0.25:
                k = int(rng.integers(1, 6)); ops.append(("S", k)); left -= k
            while left > 0:
                k = min(left, int(rng.integers(3, 25)))
                ops.append((rng.choice(["M", "M", "M", "=", "X"]), k)); left -= k
                if left > 4 and rng.random() < 0.35:
                    t = rng.choice(["I", "D", "N"])
                    k2 = int(rng.integers(1, 4))
                    if t == "I":
                        k2 = min(k2, left - 1); left -= k2
                    ops.append((t, k2))
            if rng.random() < 0.2 and ops[-1][0] in "M=X" and ops[-1][1] > 3:
                k = int(rng.integers(1, 3)); ops[-1] = (ops[-1][0], ops[-1][1] - k); ops.append(("S", k))
            qlen = sum(n for op, n in ops if op in "MIS=X")
            seq = "".join(rng.choice(list("ACGT"), qlen))
            if rng.random() < 0.1:
                i = int(rng.integers(0, qlen)); seq = seq[:i] + rng.choice(list("NRY")) + seq[i + 1:]
            qual = rng.choice([2, 12, 25, 29, 30, 31, 37, 40], qlen, p=[.03, .04, .08, .05, .1, .1, .4, .2])
            flag = 0x1 | (0x40 if mate == 0 else 0x80) | (0x2 if proper else 0) | (0x8 if mate_unmapped else 0)
            r = rng.random()
            if r < 0.02: flag |= 0x100
            elif r < 0.04: flag |= 0x400
            elif r < 0.05: flag |= 0x200
            elif r < 0.06: flag |= 0x800
            rlen = sum(n for op, n in ops if op in "MDN=X")
            ends.append(s + rlen)
            reads.append(dict(tid=tid, pos=s, mapq=int(rng.integers(0, 45)), flag=flag, name=name, cigar=ops, seq=seq,
                              qual=qual, nm=int(rng.integers(0, 4)), isize=0, extra_tags=bool(rng.random() < 0.3)))
        span = max(ends) - s1
        wild = rng.random() < 0.1
        reads[-2]["isize"] = span * (5 if wild else 1)
        reads[-1]["isize"] = -span * (5 if wild else 1)
        if rng.random() < 0.03:         # singleton: drop the mate
            reads.pop()
    # a same-name mate pair whose second read starts at the same position (order = file order)
    reads.sort(key=lambda r: (r["tid"], r["pos"]))
    return reads


def cross_scaffold_bam(path, seed, triple):
    """pairs whose mates sit on two scaffolds (a twentieth of them), singletons of one name on two scaffolds, and -- `triple` --
    names on three scaffolds (the reference's KeyError under non_discordant; merged twice under all_reads)"""
    refs = [("s%d" % i, ln) for i, ln in enumerate([9000, 7000, 12000, 800, 6000, 5000, 11000, 4000])]
    reads = random_reads(seed, refs, 9000)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    by_name = {}
    for r in reads:
        by_name.setdefault(r["name"], []).append(r)
    extra = []
    for name, rs in by_name.items():
        u = rng.random()
        if u < 0.05 and len(rs) == 2:               # second mate moves to another scaffold
            t2 = (rs[0]["tid"] + 1 + int(rng.integers(0, len(refs) - 1))) % len(refs)
            rs[1]["tid"] = t2
            rs[1]["pos"] = int(rng.integers(0, refs[t2][1] - 200))
        elif triple and u < 0.06:                   # a copy of the first read on two more scaffolds
            for k in (1, 2):
                c = dict(rs[0])
                c["tid"] = (rs[0]["tid"] + k) % len(refs)
                c["pos"] = int(rng.integers(0, refs[c["tid"]][1] - 200))
                extra.append(c)
    reads = sorted(reads + extra, key=lambda r: (r["tid"], r["pos"]))
    write_bam(path, refs, reads)
    return refs
