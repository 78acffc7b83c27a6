"""
oracle/bam_py.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pure-Python BGZF/BAM decoder + restatement of the htslib-1.9 pileup semantics
that the reference reaches through pysam at
  /root/reference/inStrain/profile/profile_utilities.py:150-153
    samfile.pileup(scaffold, truncate=True, max_depth=100000, stepper='nofilter',
                   compute_baq=True, ignore_orphans=True, ignore_overlaps=True,
                   min_base_quality=30, start=start, stop=end+1)
and of the read-pair filter that produces R2M
  /root/reference/inStrain/filter_reads.py:885-956 (get_paired_reads),
  :388-426 (evaluate_pair), :201-260 (filter_scaff2pair2info),
  :471-532 (paired_read_filter).

Third-party dependency restated here (absent from /root/reference):
  pysam (setup.py:25 `pysam>=0.15`) bundling htslib 1.9.  Published algorithm
  restated: bam_plp overlap resolution (`overlap_push`, `tweak_overlap_quality`,
  `cigar_iref2iseq_set/next` in htslib sam.c), default flag mask
  UNMAP|SECONDARY|QCFAIL|DUP, and pysam's `min_base_quality` test applied
  when listing `PileupColumn.pileups`.  Pinned by the reference's stored golden
  run on the sars_cov_2 fixture (tests/test_oracle_golden.py).
"""
import gzip
import struct
from collections import defaultdict

import numpy as np

SEQ_CODES = "=ACMGRSVTWYHKDBN"
# inStrain base order A,C,T,G (profile_utilities.py:34); everything else -> 4 ("other")
_CODE2IDX = np.full(16, 4, dtype=np.uint8)
_CODE2IDX[1] = 0   # A
_CODE2IDX[2] = 1   # C
_CODE2IDX[8] = 2   # T
_CODE2IDX[4] = 3   # G

FUNMAP, FSECONDARY, FQCFAIL, FDUP = 0x4, 0x100, 0x200, 0x400
FPROPER_PAIR, FMUNMAP = 0x2, 0x8
DEF_MASK = FUNMAP | FSECONDARY | FQCFAIL | FDUP

# cigar ops
CM, CI, CD, CN, CS, CH, CP, CEQ, CX = range(9)


class Read:
    __slots__ = ("name", "tid", "pos", "mapq", "flag", "isize", "cigar", "seq", "qual",
                 "nm", "l_qseq", "dropped")

    def ref_positions(self):
        """pysam get_reference_positions(): ref positions of M/=/X bases."""
        out = []
        r = self.pos
        for op, n in self.cigar:
            if op in (CM, CEQ, CX):
                out.append(np.arange(r, r + n))
                r += n
            elif op in (CD, CN):
                r += n
        if not out:
            return np.zeros(0, dtype=np.int64)
        return np.concatenate(out)

    def infer_query_length(self):
        return sum(n for op, n in self.cigar if op in (CM, CI, CS, CEQ, CX))

    def end_pos(self):
        r = self.pos
        for op, n in self.cigar:
            if op in (CM, CD, CN, CEQ, CX):
                r += n
        return r


def _parse_tags_nm(buf, off, end):
    nm = None
    while off < end:
        tag = buf[off:off + 2]
        typ = chr(buf[off + 2])
        off += 3
        if typ in "AcC":
            val = struct.unpack_from({"A": "c", "c": "b", "C": "B"}[typ], buf, off)[0]
            off += 1
        elif typ in "sS":
            val = struct.unpack_from("<h" if typ == "s" else "<H", buf, off)[0]
            off += 2
        elif typ in "iIf":
            val = struct.unpack_from({"i": "<i", "I": "<I", "f": "<f"}[typ], buf, off)[0]
            off += 4
        elif typ in "ZH":
            e = buf.index(b"\0", off)
            val = None
            off = e + 1
        elif typ == "B":
            sub = chr(buf[off])
            cnt = struct.unpack_from("<i", buf, off + 1)[0]
            sz = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            off += 5 + cnt * sz
            val = None
        else:
            raise ValueError("bad tag type %r" % typ)
        if tag == b"NM":
            nm = int(val)
    return nm


def read_bam(path):
    """Returns (refs [(name, length)], reads [Read ...] in file order)."""
    buf = gzip.open(path, "rb").read()   # BGZF = concatenated gzip members
    assert buf[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", buf, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, off)[0]
    off += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", buf, off)[0]
        name = buf[off + 4: off + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", buf, off + 4 + l_name)[0]
        refs.append((name, l_ref))
        off += 8 + l_name
    reads = []
    n = len(buf)
    while off < n:
        block_size = struct.unpack_from("<i", buf, off)[0]
        (tid, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, _ntid, _npos,
         tlen) = struct.unpack_from("<iiBBHHHiiii", buf, off + 4)
        p = off + 36
        r = Read()
        r.name = buf[p:p + l_read_name - 1].decode()
        p += l_read_name
        cig = np.frombuffer(buf, dtype="<u4", count=n_cigar, offset=p)
        r.cigar = [(int(c & 15), int(c >> 4)) for c in cig]
        p += 4 * n_cigar
        packed = np.frombuffer(buf, dtype=np.uint8, count=(l_seq + 1) // 2, offset=p)
        codes = np.empty(2 * len(packed), dtype=np.uint8)
        codes[0::2] = packed >> 4
        codes[1::2] = packed & 15
        r.seq = codes[:l_seq].copy()
        p += (l_seq + 1) // 2
        r.qual = np.frombuffer(buf, dtype=np.uint8, count=l_seq, offset=p).copy()
        p += l_seq
        r.nm = _parse_tags_nm(buf, p, off + 4 + block_size)
        r.tid, r.pos, r.mapq, r.flag, r.isize, r.l_qseq = tid, pos, mapq, flag, tlen, l_seq
        reads.append(r)
        off += 4 + block_size
    return refs, reads


# --------------------------------------------------------------------------------------
# read-pair filter  (filter_reads.py)
# --------------------------------------------------------------------------------------
def get_paired_reads(reads, tid):
    """filter_reads.py:885-956. reads = all records of the file; samfile.fetch(scaff)
    returns every record placed on the scaffold, whatever its flags."""
    pair2info = {}
    for r in reads:
        if r.tid != tid:
            continue
        if r.flag & FUNMAP:
            # pysam get_reference_positions() == [] for records without an alignment
            continue
        rp = r.ref_positions()
        if len(rp) == 0:
            continue
        if r.name not in pair2info:
            pair2info[r.name] = [r.nm, -1, r.mapq, r.infer_query_length(), 1,
                                 int(rp[0]), int(rp[-1])]
        else:
            i = pair2info[r.name]
            i[0] = int(i[0]) + int(r.nm)
            i[4] += 1
            i[3] += r.infer_query_length()
            i[2] = max(i[2], r.mapq)
            if i[4] == 2:
                if rp[-1] > i[5]:
                    i[1] = int(rp[-1]) - i[5]
                else:
                    i[1] = i[6] - int(rp[0])
            else:
                i[1] = -1
            i[5] = 0
            i[6] = 0
    return pair2info


def filter_pairs(scaff2pair2info, min_read_ani=0.95, min_mapq=-1, max_insert_relative=3,
                 min_insert=50, pairing_filter="paired_only"):
    """paired_read_filter (:471-532, 'paired_only' and 'all_reads'-free subset) +
    filter_scaff2pair2info (:201-260) + evaluate_pair (:388-426).
    Returns (scaff2pair2mm, tallies)."""
    assert pairing_filter == "paired_only"
    kept = {s: {p: i for p, i in p2i.items() if i[4] == 2} for s, p2i in scaff2pair2info.items()}
    inserts = [i[1] for s in kept for i in kept[s].values() if i[4] == 2]
    median_insert = float(np.median(inserts)) if inserts else float("nan")
    max_insert = median_insert * max_insert_relative
    out, tallies = {}, {}
    for s, p2i in kept.items():
        out[s] = {}
        t = defaultdict(int)
        for p, i in p2i.items():
            t["pass_pairing_filter"] += 1
            f = [0, 0, 0, 0]
            pid = 1 - (float(i[0]) / float(i[3]))
            if pid > min_read_ani:
                f[0] = 1
            if i[2] > min_mapq:
                f[3] = 1
            if i[4] == 2 and i[1] != -1:
                if i[1] > min_insert:
                    f[2] = 1
                if i[1] < max_insert:
                    f[1] = 1
            else:
                f[1] = f[2] = 1
            t["pass_min_read_ani"] += f[0]
            t["pass_max_insert"] += f[1]
            t["pass_min_insert"] += f[2]
            t["pass_min_mapq"] += f[3]
            if sum(f) == 4:
                t["filtered_pairs"] += 1
                out[s][p] = int(i[0])
        t["unfiltered_reads"] = sum(i[4] for i in scaff2pair2info[s].values())
        t["unfiltered_pairs"] = sum(1 for i in scaff2pair2info[s].values() if i[4] == 2)
        t["unfiltered_singletons"] = sum(1 for i in scaff2pair2info[s].values() if i[4] == 1)
        t["median_insert"] = median_insert
        tallies[s] = dict(t)
    return out, tallies


# --------------------------------------------------------------------------------------
# htslib-1.9 overlap resolution (sam.c: cigar_iref2iseq_set/next, tweak_overlap_quality,
# overlap_push)
# --------------------------------------------------------------------------------------
class _Cur:
    __slots__ = ("cig", "k", "icig", "iseq", "iref")


def _cur_set(c, pos):
    if pos < 0:
        return -1
    c.k = 0
    c.icig = 0
    c.iseq = 0
    c.iref = 0
    cig = c.cig
    while c.k < len(cig):
        op, n = cig[c.k]
        if op == CS:
            c.k += 1; c.iseq += n; c.icig = 0
        elif op in (CH, CP):
            c.k += 1; c.icig = 0
        elif op in (CM, CEQ, CX):
            pos -= n
            if pos < 0:
                c.icig = n + pos
                c.iseq += c.icig
                c.iref += c.icig
                return 0
            c.k += 1; c.iseq += n; c.icig = 0; c.iref += n
        elif op == CI:
            c.k += 1; c.iseq += n; c.icig = 0
        elif op in (CD, CN):
            pos -= n
            if pos < 0:
                pos = 0
            c.k += 1; c.icig = 0; c.iref += n
        else:
            raise ValueError("cigar op")
    c.iseq = -1
    return -1


def _cur_next(c):
    cig = c.cig
    while c.k < len(cig):
        op, n = cig[c.k]
        if op in (CM, CEQ, CX):
            if c.icig >= n - 1:
                c.icig = 0; c.k += 1
                continue
            c.iseq += 1; c.icig += 1; c.iref += 1
            return 0
        if op in (CD, CN):
            c.k += 1; c.iref += n; c.icig = 0
        elif op in (CI, CS):
            c.k += 1; c.iseq += n; c.icig = 0
        elif op in (CH, CP):
            c.k += 1; c.icig = 0
        else:
            raise ValueError("cigar op")
    c.iseq = -1
    c.iref = -1
    return -1


def tweak_overlap_quality(a, b):
    ca, cb = _Cur(), _Cur()
    ca.cig, cb.cig = a.cigar, b.cigar
    iref = b.pos
    a_ret = _cur_set(ca, iref - a.pos)
    if a_ret < 0:
        return
    b_ret = _cur_set(cb, iref - b.pos)
    if b_ret < 0:
        return
    aq, bq, aseq, bseq = a.qual, b.qual, a.seq, b.seq
    while True:
        while ca.iref >= 0 and ca.iref < iref - a.pos:
            a_ret = _cur_next(ca)
        if a_ret < 0:
            break
        if iref < ca.iref + a.pos:
            iref = ca.iref + a.pos
        while cb.iref >= 0 and cb.iref < iref - b.pos:
            b_ret = _cur_next(cb)
        if b_ret < 0:
            break
        if iref < cb.iref + b.pos:
            iref = cb.iref + b.pos
        iref += 1
        if ca.iref + a.pos != cb.iref + b.pos:
            continue
        qa, qb = int(aq[ca.iseq]), int(bq[cb.iseq])
        if aseq[ca.iseq] == bseq[cb.iseq]:
            q = qa + qb
            aq[ca.iseq] = 200 if q > 200 else q
            bq[cb.iseq] = 0
        elif qa >= qb:
            aq[ca.iseq] = int(0.8 * qa)
            bq[cb.iseq] = 0
        else:
            bq[cb.iseq] = int(0.8 * qb)
            aq[ca.iseq] = 0


def apply_max_depth(reads, tid, max_depth=100000):
    """max_depth of the pileup call (profile_utilities.py:150; polymorpher.py:290), as htslib 1.9 applies it in bam_plp_push
    (sam.c): a read is NOT pushed into the pileup buffer when
        iter->tid == b->core.tid && iter->pos == b->core.pos && iter->mp->cnt > iter->maxcnt
    i.e. when it starts exactly at the column the iterator stands on while the buffer's node pool holds more than max_depth
    nodes (the buffered reads + the list's sentinel tail).  The iterator stands on a read's start only once an EARLIER read with
    the same start has been pushed (bam_plp_auto emits every column before that start first), so the first read of a
    same-start run is always taken; the buffer then holds the accepted reads whose end lies beyond the last emitted column
    (end >= start; bam_plp_next frees a node at column p when end <= p).  Dropped reads never reach overlap_push
    (overlap_remove) nor any column.  Reads the flag mask removes are never pushed and do not count.
    PARITY UNPINNED: no fixture of the reference exercises a column deeper than 100 000 (pysam is not in this image); this follows
    the published htslib-1.9 source.  Marks r.dropped on the reads of `tid`; returns how many were dropped."""
    import heapq
    ends = []                       # min-heap of the end positions of the buffered reads
    n_drop = 0
    prev_pos = None
    for r in reads:
        if r.tid != tid or (r.flag & DEF_MASK):
            continue
        r.dropped = False
        while ends and ends[0] < r.pos:         # columns up to pos - 1 have been emitted: nodes with end <= pos - 1 are gone
            heapq.heappop(ends)
        if prev_pos == r.pos and len(ends) + 1 > max_depth:
            r.dropped = True
            n_drop += 1
            continue
        prev_pos = r.pos
        heapq.heappush(ends, r.end_pos())
    return n_drop


def iterate_splits(s_len, window_len=10000):
    """fasta.py:56-73: 0-based, double-inclusive (start, end) of a scaffold's splits"""
    n_chunks = s_len // window_len + 1
    chunk = int(s_len / n_chunks)
    out, start, end = [], 0, 0
    for i in range(n_chunks):
        if i + 1 == n_chunks:
            out.append((start, s_len - 1))
        else:
            end += chunk
            out.append((start, end - 1))
            start += chunk
    return out


def expand_observations_per_split(reads, tid, r2m, splits, **kw):
    """The reference's own structure, literally (profile_utilities.py:150-153): ONE pileup iterator per split -- fed the reads that
    overlap [start, end] (pos <= end and bam_endpos > start, the region fetch of samfile.pileup(..., start=start, stop=end+1)), with
    its own max_depth buffer (apply_max_depth on that subset alone) and its own overlap resolution on its own copies of the records,
    truncated to the split's columns.  Returns (pos, base, mm, pair) over all splits, pair = ids of the read names in order of first
    appearance; differs from the whole-scaffold replay only where >= 100 000 reads pile up within a read's length of a split bound.
    PARITY UNPINNED like apply_max_depth (no fixture of the reference is that deep, pysam is not in this image)."""
    import copy
    gid = {}
    P, B, M, R = [], [], [], []
    for (s, e) in splits:
        sub = []
        for r in reads:
            if r.tid != tid or (r.flag & DEF_MASK):
                continue
            if r.pos <= e and max(r.end_pos(), r.pos + 1) > s:
                c = copy.copy(r)
                c.qual = r.qual.copy()
                sub.append(c)
        apply_max_depth(sub, tid)
        resolve_overlaps(sub, tid)
        pos, base, mm, pr, name2id = expand_observations(sub, tid, r2m, **kw)
        keep = (pos >= s) & (pos <= e)
        local = np.zeros(max(len(name2id), 1), dtype=np.int64)
        for name, i in name2id.items():
            local[i] = gid.setdefault(name, len(gid))
        P.append(pos[keep]); B.append(base[keep]); M.append(mm[keep]); R.append(local[pr[keep]] if len(pr) else pr)
    if not P:
        z = np.zeros(0, dtype=np.int64)
        return z, z.astype(np.uint8), z, z
    return np.concatenate(P), np.concatenate(B).astype(np.uint8), np.concatenate(M), np.concatenate(R)


def _is_dropped(r):
    try:
        return r.dropped
    except AttributeError:
        return False


def resolve_overlaps(reads, tid):
    """Apply overlap_push in file order to the reads of one reference (mutates qual).
    The hash entry of a read that has left the pileup buffer before its mate arrives is
    dropped (overlap_remove); then the mate is entered as a fresh first-seen read."""
    H = {}
    for r in reads:
        if r.tid != tid or (r.flag & DEF_MASK) or _is_dropped(r):
            continue
        if (r.flag & FMUNMAP) or not (r.flag & FPROPER_PAIR):
            continue
        if abs(r.isize) >= 2 * r.l_qseq:
            continue
        a = H.get(r.name)
        if a is not None and a.end_pos() <= r.pos:
            a = None        # earlier read already left the buffer -> entry was removed
        if a is None:
            H[r.name] = r
        else:
            del H[r.name]
            tweak_overlap_quality(a, r)


def expand_observations(reads, tid, r2m, min_base_quality=30, skip_mm=False, ref_len=None):
    """Packed per-base observations of the reads kept by R2M, in file order then
    query order: (pos, base_idx[A,C,T,G,other], mm, pair_id).  This is exactly the set
    of (column, pileupread) visits on which get_base_counts_mm (profile_utilities.py:268-286)
    touches `table` (ACGT -> +1; other -> level made present only)."""
    name2id = {}
    P, B, M, R = [], [], [], []
    for r in reads:
        if r.tid != tid or (r.flag & DEF_MASK) or _is_dropped(r):
            continue
        if r.name not in r2m:
            continue
        pid = name2id.setdefault(r.name, len(name2id))
        mm = 0 if skip_mm else r2m[r.name]
        q = 0
        ref = r.pos
        for op, n in r.cigar:
            if op in (CM, CEQ, CX):
                qs = r.qual[q:q + n]
                ok = qs >= min_base_quality
                if ref_len is not None:      # pileups are truncated to the scaffold (truncate=True, stop=end+1)
                    rp = ref + np.arange(n)
                    ok = ok & (rp >= 0) & (rp < ref_len)
                keep = np.nonzero(ok)[0]
                if len(keep):
                    P.append(ref + keep)
                    B.append(_CODE2IDX[r.seq[q:q + n][keep]])
                    M.append(np.full(len(keep), mm, dtype=np.int64))
                    R.append(np.full(len(keep), pid, dtype=np.int64))
                q += n
                ref += n
            elif op in (CI, CS):
                q += n
            elif op in (CD, CN):
                ref += n
    if not P:
        z = np.zeros(0, dtype=np.int64)
        return z, z.astype(np.uint8), z, z, name2id
    return (np.concatenate(P).astype(np.int64), np.concatenate(B).astype(np.uint8),
            np.concatenate(M), np.concatenate(R), name2id)
