"""Multi-GPU layer: one process per GPU, scaffolds sharded across ranks, no data-path collective;
a single final gather of the (small) SNV / linkage tables to rank 0 over RCCL (xGMI).

Replaces the reference's process-level data parallelism over splits
(/root/reference/inStrain/profile/profile_controller.py:157-193 queues, :243-271 spawn workers,
:441 per-scaffold cost estimate).  The split is the independent unit (linkage never crosses a
split bound, profile_utilities.py:164,188-189; merge is concatenation :785-792), scaffolds are
kept whole on one rank so per-scaffold summaries stay local.
"""
import os

import numpy as np


def lpt_shards(costs, world):
    """Longest-processing-time bin packing: costs[i] = estimated work of scaffold i (the reference
    uses filtered read pairs, profile_controller.py:460-465). Returns list of index lists."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    shards = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def init_from_env(backend=None):
    """torch.distributed init from the torchrun environment; returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" IS RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_tables(tables, dst=0, device=None):
    """tables: {name: numpy structured array}. Rank `dst` gets {name: concatenation over ranks in rank order}; other
    ranks get None.  The one collective of the path: an all_gather of the byte counts, then every rank sends
    exactly its bytes to `dst` in one grouped round of point-to-point transfers (ncclSend / ncclRecv inside a group on
    RCCL; all tables of a rank travel as ONE packed buffer) -- nothing is padded to the largest rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not os.environ.get("ISX_DIST_FORCE")):
        return tables
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    names = sorted(tables)
    sizes = torch.tensor([tables[n].nbytes for n in names], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    mine = np.concatenate([np.frombuffer(np.ascontiguousarray(tables[n]).tobytes(), dtype=np.uint8) for n in names]) \
        if names else np.zeros(0, np.uint8)
    ops, bufs = [], {}
    if rank == dst:
        for r in range(world):
            if r == dst or all_sizes[r].sum() == 0:
                continue
            bufs[r] = torch.empty(int(all_sizes[r].sum()), dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    elif len(mine):
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(mine.copy()).to(device), dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    out = {}
    raw = {r: (bufs[r].cpu().numpy() if r in bufs else (mine if r == dst else np.zeros(0, np.uint8))) for r in range(world)}
    for j, n in enumerate(names):
        parts = []
        for r in range(world):
            o = int(all_sizes[r, :j].sum())
            parts.append(np.frombuffer(raw[r][o:o + int(all_sizes[r, j])].tobytes(), dtype=tables[n].dtype))
        out[n] = np.concatenate(parts)
    return out


def all_gather_concat(arr, device=None):
    """Every rank gets the concatenation (rank order) of every rank's 1-d array: an all_gather of the lengths, then one
    all_gather of the arrays padded to the longest.  The exchange step of the sharded scan: the read filter's median insert
    is a property of the whole file (filter_reads.py:216-219), each rank has only its share's insert sizes."""
    import torch
    import torch.distributed as dist
    arr = np.ascontiguousarray(arr)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not os.environ.get("ISX_DIST_FORCE")):
        return arr                                  # (ISX_DIST_FORCE: a world of one still goes through the collectives -- tests)
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([len(arr)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    ns = [int(x.item()) for x in ns]
    m = max(max(ns), 1)
    raw = np.zeros(m * arr.dtype.itemsize, dtype=np.uint8)
    raw[:arr.nbytes] = np.frombuffer(arr.tobytes(), dtype=np.uint8)
    mine = torch.from_numpy(raw).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return np.concatenate([np.frombuffer(b.cpu().numpy().tobytes()[:k * arr.dtype.itemsize], dtype=arr.dtype) for b, k in zip(bufs, ns)])


def pack_batches(n_pos, n_obs, max_pos, max_obs):
    """Cut a rank's scaffolds / genomes (in the given order) into batches for one pipe: consecutive items are
    taken while the flat positions stay <= max_pos and the (estimated) observations <= max_obs.  The reference
    groups its profile commands the same way, by estimated seconds (profile_controller.py:397-457).  An item
    larger than a cap gets a batch of its own (the caller sizes the pipe from the largest batch)."""
    out, cur, p, o = [], [], 0, 0
    for i, (a, b) in enumerate(zip(n_pos, n_obs)):
        if cur and (p + a > max_pos or o + b > max_obs):
            out.append(cur)
            cur, p, o = [], 0, 0
        cur.append(i)
        p += int(a)
        o += int(b)
    if cur:
        out.append(cur)
    return out


# ---- one BAM profiled by several ranks (one process per GPU) ----
def shard_scaffolds(filtered_pairs, lengths, world):
    """LPT over the scaffolds that have reads, on the reference's own cost estimate: 0.0061 s x pairs + 0.2 s per
    scaffold (profile_controller.py:460-465).  Every rank computes the same answer from the same scan."""
    idx = [i for i in range(len(lengths)) if lengths[i] > 0]
    cost = [0.0061401594694834305 * float(filtered_pairs[i]) + 0.2 for i in idx]
    return [[idx[j] for j in sh] for sh in lpt_shards(cost, world)]


def scan_share(bf, rank, world, device=None):
    """Every rank scans its share of the BAM (isx_bam_scan_part).  A share scan can fail on one rank alone (no record start
    near its share's beginning, a truncated block): the ranks exchange a status word BEFORE the first data collective, so
    that nobody is left waiting in an all_gather for a rank that raised.  -> True when every share was scanned; False when
    any failed -- the caller then lets every rank scan the whole file with a fresh handle (a handle scans one share only)."""
    from . import engine
    ok, why = 1, ""
    try:
        bf.scan(part=(rank, world))
    except engine.IsxError as e:
        ok, why = 0, str(e)
    if int(all_gather_concat(np.asarray([ok], dtype=np.int32), device).min()) == 0:
        import logging
        logging.warning("share scan failed on a rank (%s): every rank scans the whole file", why or "another rank")
        return False
    return True


CROSS_DT = np.dtype([("h1", "<u8"), ("h2", "<u8"), ("tid", "<i4"), ("rank", "<i4"), ("entry", "<i8"), ("info", "<i8", (4,))])


def resolve_cross_names(bf, rank, world, device=None):
    """non_discordant / all_reads on a file scanned in shares: the reference looks every read name up across ALL scaffolds
    (paired_read_filter, filter_reads.py:497-532).  Every rank hands out the 64-bit hashes of its shares' pair names (8 bytes
    a pair -- an all_gather, not a second scan of the file by every rank), the names held more than once anywhere are found
    from those, and only for them the full records (128-bit key, scaffold, nm / mapq / length / reads) are exchanged; each
    rank then tells its handle how often each of its names occurs in the file and what _merge_info leaves in its entry
    (isx_bam_set_cross_names).  Collective: every rank must call it."""
    h1, h2, tid, info = bf.pair_keys()
    live = np.flatnonzero(info[:, 3] > 0)
    sh = np.sort(all_gather_concat(h1[live], device))
    dup = np.unique(sh[1:][sh[1:] == sh[:-1]])                   # hashes met more than once in the file
    m = live[np.isin(h1[live], dup)]
    rec = np.zeros(len(m), dtype=CROSS_DT)
    rec["h1"], rec["h2"], rec["tid"], rec["rank"], rec["entry"], rec["info"] = h1[m], h2[m], tid[m], rank, m, info[m]
    r = all_gather_concat(rec, device)
    if len(r) == 0:
        bf.set_cross_names([], [], np.zeros((0, 4), np.int64))
        return 0
    # groups of one name (both hashes equal), its scaffolds in header order: a scaffold's table holds a name once
    r = r[np.lexsort((r["tid"], r["h2"], r["h1"]))]
    first = np.r_[True, (r["h1"][1:] != r["h1"][:-1]) | (r["h2"][1:] != r["h2"][:-1])]
    gid = np.cumsum(first) - 1
    start = np.flatnonzero(first)
    occ = np.bincount(gid)[gid]
    # _merge_info in header order: the j-th occurrence ends up with the sum of the first j, the FIRST one with the sum of all
    # (every later scaffold merges into the entry the name was first seen in)
    cs = np.cumsum(r["info"], axis=0)
    pref = cs - (cs[start] - r["info"][start])[gid]
    total = pref[np.r_[start[1:], len(r)] - 1][gid]
    merged = np.where(first[:, None], total, pref)
    mine = (r["rank"] == rank) & (occ >= 2)
    bf.set_cross_names(r["entry"][mine], occ[mine], merged[mine])
    return int(mine.sum())


def profile_bam_sharded(bam, s2s, null_model, rank, world, gather=True, device=None, **kwargs):
    """Scaffolds of ONE sorted BAM over `world` ranks.
    paired_only (the default): every rank scans only its SHARE of the file (isx_bam_scan_part) and owns the scaffolds whose
    reads start there -- shares are equal in compressed bytes, i.e. in reads, which is the reference's own cost measure
    (profile_controller.py:460-465); the one thing that is global, the read filter's median insert, comes from an
    all_gather of the shares' insert sizes (all_gather_concat).  Other pairing filters look read names up across all
    scaffolds: there every rank scans the whole file and the scaffolds are dealt by LPT (shard_scaffolds).
    Each rank profiles its scaffolds through profile_bam; SNV / linkage / per-scaffold summary tables are gathered on rank 0
    (gather_tables).  Returns (splits of this rank, gathered tables on rank 0 | None, this rank's load)."""
    import pandas as pd
    from . import engine
    from ._lib import LD_DT, SCAFFOLD_LEVEL_DT, SNV_DT
    from .profile import profile_utilities as pu
    fkw = dict(min_read_ani=kwargs.get('min_read_ani', 0.95), min_mapq=kwargs.get('min_mapq', -1),
               max_insert_relative=kwargs.get('max_insert_relative', 3), min_insert=kwargs.get('min_insert', 50),
               pairing_filter=kwargs.get('pairing_filter', 'paired_only'))
    sharded_scan = world > 1 and kwargs.get('scan', 'sharded') == 'sharded'
    bf = engine.BamFile(bam, threads=int(kwargs.get('host_threads', 0)))
    try:
        refs = bf.refs()
        usable = [i for i, (n, ln, _) in enumerate(refs) if n in s2s and len(s2s[n]) == ln]
        # the read filter only ever sees the scaffolds of the fasta (filter_reads.py:63-77): every rank restricts the insert
        # sizes it contributes, and its own filter run, to the same set
        filter_refs = usable if len(usable) < len(refs) else []
        bf.set_wanted_refs(filter_refs)
        extra = {}
        if sharded_scan:
            part = (rank, world)
            sharded_scan = scan_share(bf, rank, world, device)      # False: a share failed somewhere, everybody falls back together
            if not sharded_scan:                                    # a handle scans one share only: a fresh one for the whole file
                bf.close()
                bf = engine.BamFile(bam, threads=int(kwargs.get('host_threads', 0)))
                bf.set_wanted_refs(filter_refs)
        if sharded_scan:
            if fkw['pairing_filter'] == 'paired_only':
                ins = all_gather_concat(bf.insert_sizes(), device)
            else:
                # names are looked up across the whole file: exchange them, run the filter's first half to see who goes through
                # paired_read_filter (the median is taken over exactly those), and agree on failure before the next collective
                # (a name on three scaffolds is the reference's KeyError -- on one rank only)
                resolve_cross_names(bf, rank, world, device)
                if kwargs.get('priority_reads'):
                    bf.set_priority_reads(kwargs['priority_reads'])
                ok, why = 1, ""
                try:
                    bf.filter(median_insert=0.0, **fkw)
                except engine.IsxError as e:
                    ok, why = 0, str(e)
                if int(all_gather_concat(np.asarray([ok], dtype=np.int32), device).min()) == 0:
                    raise engine.IsxError(-1, why or "the read filter failed on another rank")
                ins = all_gather_concat(bf.filter_insert_sizes(), device)
            median = float(np.median(ins)) if len(ins) else 0.0
            reads, _ = bf.ref_counts()
            # a scaffold without reads starts in nobody's share: deal those round robin (they still get their empty profile)
            owned = all_gather_concat((reads > 0).astype(np.uint8), device).reshape(world, -1).sum(axis=0)
            mine = [t for t in usable if reads[t] > 0 or (owned[t] == 0 and t % world == rank)]
            extra = dict(scan_part=part, median_insert=median)
        else:
            bf.scan()
            bf.filter(**fkw)
            _, pairs = bf.ref_counts()
            shards = shard_scaffolds([pairs[i] for i in usable], [refs[i][1] for i in usable], world)
            mine = [usable[j] for j in shards[rank]]
        W = int(kwargs.get('window_length', 10000))
        rows = [(refs[t][0], i, s, e) for t in mine for i, (s, e) in enumerate(pu.iterate_splits(refs[t][1], W))]
        fdb = pd.DataFrame(rows, columns=["scaffold", "split_number", "start", "end"])
        tabs = {}
        kw = {k: v for k, v in kwargs.items() if k != 'scan'}
        splits = pu.profile_bam(bam, fdb, None, None, s2s=s2s, null_model=null_model, scaffold_levels=tabs, bamfile=bf,
                                filter_refs=filter_refs, **extra, **kw) if len(rows) else {}
        _, pairs = bf.ref_counts()
        load = float(sum(pairs[t] for t in mine))
    finally:
        bf.close()
    if not gather:
        return splits, None, load
    # packed tables with the scaffold's index in the BAM header as the key
    tid_of = {refs[t][0]: t for t in mine}
    snv_dt = np.dtype([("tid", "<i4")] + [(n, SNV_DT[n]) if SNV_DT[n].shape == () else (n, SNV_DT[n].base, SNV_DT[n].shape) for n in SNV_DT.names])
    ld_dt = np.dtype([("tid", "<i4")] + [(n, LD_DT[n]) for n in LD_DT.names])
    sm_dt = np.dtype([("tid", "<i4")] + [(n, SCAFFOLD_LEVEL_DT[n]) for n in SCAFFOLD_LEVEL_DT.names])
    snv_parts, ld_parts = [], []
    seen = set()
    for S in splits.values():
        tb, i = S.__dict__.get('_src', (None, None))
        if tb is None or (id(tb), i) in seen:
            continue
        seen.add((id(tb), i))
        for src, cut, dt, parts, keys in ((tb.snv, tb.s_cut, snv_dt, snv_parts, ("gpos",)), (tb.ld, tb.l_cut, ld_dt, ld_parts, ("gpos_a", "gpos_b"))):
            rws = src[cut[i]:cut[i + 1]]
            o = np.zeros(len(rws), dtype=dt)
            for n in rws.dtype.names:
                o[n] = rws[n]
            for k in keys:
                o[k] = rws[k] - tb.offset[i]            # scaffold coordinates
            o["tid"] = tid_of[S.scaffold]
            parts.append(o)
    sm_parts = []
    for name, lv in tabs.items():                   # the device's per-(scaffold, mm) aggregates, as they came (no pandas round trip)
        o = np.zeros(len(lv), dtype=sm_dt)
        for n in SCAFFOLD_LEVEL_DT.names:
            o[n] = lv[n]
        o["tid"] = tid_of[name]
        sm_parts.append(o)
    tables = {"snv": np.concatenate(snv_parts) if snv_parts else np.zeros(0, snv_dt),
              "ld": np.concatenate(ld_parts) if ld_parts else np.zeros(0, ld_dt),
              "summary": np.concatenate(sm_parts) if sm_parts else np.zeros(0, sm_dt)}
    return splits, gather_tables(tables, dst=0, device=device), load
