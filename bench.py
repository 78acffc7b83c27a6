#!/usr/bin/env python3
"""bench.py -- headline measurement of the `inStrain profile` hot path on MI355X.

A "step" = one batch of synthetic reads profiled ONCE, the way production does it: the batch is handed over from
host memory as READ SEGMENTS (isx_pipe_submit_reads: 64-byte records staged into pinned memory by the pipe's host
threads, hipMemcpyAsync in), expanded and profiled on the device (k_pileup_dense walks the segments into its LDS
window histograms, SNV call epilogue) and its tables copied back to pinned host memory (isx_pipe_collect).  The K
timed steps stream K batches through the pipe with copy-in / pass / copy-out of consecutive batches overlapping; the
batches are distinct (up to 32 variants of the workload, synth.shifted_variant_segs; cycled beyond that).  At N=1 the
workload is BASELINE.json configs[1] (C2: one 5 Mbp genome, 20x, 2x150 bp, --skip_mm_profiling, linkage off).  For
N>1 every rank streams its own C2 genomes (scaffolds shard embarrassingly; weak scaling; no data-path collective;
one final RCCL gather of the SNV tables after the timed region, reported separately).

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects `roofline` (dominant kernel
vs the HBM roof, the kernel alone over a resident batch), `roofline_lds` (the same kernel vs the LDS atomic rate, its
real bound), `roofline_observation_kernel` (the 2-byte-record kernel of the observation hand-over), `roofline_pcie`,
`resident`, `mm_on`, `linkage` (C3), `c5` (configs[4]: the whole kept database through one GPU at N=1, with
`cpu_baseline` = the C port and `cpu_baseline_python` = the reference-like Python restatement on the same
configuration), `bam_sharded` (one BAM over the ranks), `profile_bam` / `c5_bam` (BAM on disk -> SplitObjects) and
`cpu_baseline` / `cpu_baseline_python` (C2; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
PCIE_PEAK_GBS = 64.0        # PCIe Gen5 x16, one direction (raw; ~57 measured with hipMemcpyAsync on the GPU box)


LDS_ATOMIC_PEAK = 256 * 16 * 2.4e9   # LDS read-modify-write lanes per second: 256 CUs x 16 lanes per clock (a ds_add_u32 wave
                                      # instruction takes 4 LDS cycles, MI355X_MICROARCH.md LDS table: ds_write_b32 class) x 2.4 GHz


def c2_workload(seed, scale=1.0, with_mm=False):
    """C2 of SURVEY 8(d). One generation serves every leg: `segs` = the reads as segments with mm = 0 (the headline
    --skip_mm_profiling run, what the read-level hand-over ships), `obs` = the observation stream they stand for (the
    observation hand-over and the CPU baselines), `segs_mm` / `obs_mm` (with_mm) carry the pairs' mismatch counts."""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=int(5_000_000 * scale), coverage=20, n_sites=int(5000 * scale),
                            seed=seed, skip_mm=not with_mm)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    if with_mm:
        w["obs_mm"], w["segs_mm"], w["n_mm_bins_mm"] = w["obs"], segs, w["n_mm_bins"]
        o = w["obs"].copy()
        o["mm"] = 0
        w["obs"] = o
        w["n_mm_bins"] = 1
        segs = engine.SegBatch(segs.gpos, segs.len, segs.bases, np.zeros(segs.n_seg, np.uint8), segs.pair)
    w["segs"] = segs
    return w


def pileup_algorithmic_bytes(n_obs, n_pos, n_entries, dense, record_bytes=8, n_rec=0, out_bytes_per_pos=20, ref_bytes_per_pos=1.0):
    """Bytes one launch has to move.  In: the resident stream -- record_bytes per observation (2 / 4 for
    the streams the library builds from isx_obs, 8 = isx_obs as is, SURVEY 8(d)'s figure), or for read segments
    (record_bytes 64) 64 B per record + 4 B per 16 records of position bases -- and 1 B/position of reference.  Out, dense
    (M == 1): 16 B counts + 4 B clonality per position (+ 2 B coverage in a pipe slot); mm path: 32 B per present
    (position, mm) entry."""
    b = (n_rec * 64 + n_rec // 16 * 4) if record_bytes == 64 else n_obs * record_bytes
    b += int(n_pos * ref_bytes_per_pos)         # (a pipe slot holds the reference two codes per byte)
    b += n_pos * out_bytes_per_pos if dense else n_entries * 32
    return b


def slot_out_bytes_per_pos(n_obs, n_pos):
    """What k_pileup_dense writes per position in a pipe slot without want_counts (the shrunk hand-back): 16-bit coverage + fp32
    clonality, + the 8-bit coverage a shallow batch (mean depth < 16) sends home instead -- the count table is not written at
    all; the lists (clonalities other than 1.0, saturated coverages, SNV rows) are a few bytes per thousand positions."""
    return 7 if n_obs < 16 * n_pos else 6


def split_obs_ranges(obs_gpos, bounds, chunk=1024):
    """record range that can touch each split (same prefix-max / suffix-min directory as the library)"""
    n = len(obs_gpos)
    nch = (n + chunk - 1) // chunk
    pad = np.full(nch * chunk, obs_gpos[-1] if n else 0, dtype=np.int64)
    pad[:n] = obs_gpos
    m = pad.reshape(nch, chunk)
    pmax = np.maximum.accumulate(m.max(axis=1))
    smin = np.minimum.accumulate(m.min(axis=1)[::-1])[::-1]
    lo = np.searchsorted(pmax, bounds[:-1], side="left") * chunk
    hi = np.searchsorted(smin, bounds[1:], side="left") * chunk
    return np.minimum(lo, n), np.minimum(hi, n)


def cgroup_cpus():
    """cpus this container may use on average (cgroup v2 cpu.max), None when unlimited"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(q) // int(p))
    except Exception:
        return None


def host_cpus():
    return cgroup_cpus() or os.cpu_count() or 1


def bind_to_gpu_numa_node(torch, local):
    """One process per GPU, bound to the cpus of the NUMA node the GPU hangs off: the buffers this rank hands over are
    then first touched next to the GPU and the pipe's encoder threads read local memory (tools/bench_pipe.py: a remote
    hand-over costs up to 2x).  Returns the node (or None when sysfs does not tell)."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _split_jobs(w):
    """per-split inputs of a workload's observation stream, prepared once outside any timing (the reference's workers get
    theirs from the BAM)"""
    obs, pair, bounds = w["obs"], w["pair"], w["split_bounds"]
    gpos = obs["gpos"].astype(np.int64)
    lo, hi = split_obs_ranges(gpos, bounds)
    letters = np.array(list("ACTGN"))
    jobs = []
    for j in range(len(bounds) - 1):
        s, e = int(bounds[j]), int(bounds[j + 1])
        sl = slice(int(lo[j]), int(hi[j]))
        jobs.append((np.ascontiguousarray(gpos[sl], dtype=np.int32), np.ascontiguousarray(obs["base"][sl]),
                     np.ascontiguousarray(obs["mm"][sl], dtype=np.int32), np.ascontiguousarray(pair[sl], dtype=np.int32),
                     "".join(letters[w["ref_codes"][s:e]]), s, e - s, int(((gpos[sl] >= s) & (gpos[sl] < e)).sum())))
    return jobs


def cpu_baseline(w, budget_s=25.0, min_s=10.0):
    """The oracle's C port of the reference per-column loop (oracle/oracle_core.c), split by split exactly
    like profile_split, on the splits of the SAME workload (wrapped around) for >= min_s of wall time:
    first on one core, then on T host threads (the C call releases the GIL; the reference parallelises
    over splits the same way, profile_controller.py:243-271).  `value` / `cores` = the T-thread run."""
    _trace("cpu_baseline")
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from tests import util
    lut, fb = util.load_lut()
    bounds = w["split_bounds"]
    jobs = _split_jobs(w)
    n_splits = len(jobs)

    def one(job):
        oracle.profile_split(job[0], job[1], job[2], job[3], job[4], job[5], lut, fb, min_cov=5, min_freq=0.05, min_snp=20,
                             convert=False)
        return job[6], job[7]

    def run(threads, min_time):
        done_pos = done_obs = n_done = 0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            while True:
                for p, o in ex.map(one, jobs):                 # one sweep over the workload's splits
                    done_pos += p; done_obs += o; n_done += 1
                el = time.perf_counter() - t0
                if el > min_time or el > budget_s:
                    break
        dt = time.perf_counter() - t0
        return w["profiled_bases"] * (done_pos / float(bounds[-1])) / 1e9 / dt, n_done, done_pos, done_obs, dt

    T = max(1, min(32, host_cpus()))
    v1, n1, _, _, dt1 = run(1, min_s / 2)
    vT, nT, posT, obsT, dtT = run(T, min_s)
    return {"value": vT, "unit": "Gbp/s", "cores": T, "kind": "port", "single_core_value": v1,
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": cgroup_cpus(),
            "sample": "%d split profiles on %d threads in %.1f s (the workload's %d splits, wrapped around; %.1f Mbp, "
                      "%d kept observations) after %d on one thread in %.1f s; oracle/oracle_core.c, pileup+SNV call+linkage"
                      % (nT, T, dtT, n_splits, posT / 1e6, obsT, n1, dt1)}


_PY_NM = None


def _py_init(nm):
    global _PY_NM
    _PY_NM = nm
    if REPO not in sys.path:
        sys.path.insert(0, REPO)


def _py_split(job):
    from oracle import py_columns
    py_columns.profile_split(job[0], job[1], job[2], job[3], job[4], job[5], _PY_NM, min_cov=5, min_freq=0.05, min_snp=20)
    return job[6], job[7]


def cpu_baseline_python(w, n_splits=200, budget_s=18.0):
    """The reference-like CPU baseline SURVEY 8(d) asks for: a faithful per-column PYTHON restatement of the reference's
    split worker (oracle/py_columns.py: dict-of-numpy-arrays count tables filled read by read, Python arithmetic per
    (position, mm), per-read SNV lists, dict-of-dicts linkage network) with multiprocessing over splits exactly like the
    reference's worker pool (profile_controller.py:243-271), P = the cpus this process may use, on a subsample of the SAME
    workload's splits (evenly spaced), stopped after budget_s; the rate is extrapolated linearly (cost is per split)."""
    _trace("cpu_baseline_python")
    import multiprocessing as mp
    from tests import util
    lut, fb = util.load_lut()
    nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
    nm[-1] = fb
    jobs = _split_jobs(w)
    pick = np.unique(np.linspace(0, len(jobs) - 1, min(n_splits, len(jobs))).astype(int))
    sample = [jobs[i] for i in pick]
    P = max(1, min(64, host_cpus()))
    done_pos = done_obs = n_done = 0
    # spawned workers (a forked copy of a process that holds a HIP context is not safe); they are up before the clock starts
    with mp.get_context("spawn").Pool(P, initializer=_py_init, initargs=(nm,)) as pool:
        pool.map(abs, range(P))
        t0 = time.perf_counter()
        for p, o in pool.imap_unordered(_py_split, sample):
            done_pos += p; done_obs += o; n_done += 1
            if time.perf_counter() - t0 > budget_s:
                pool.terminate()
                break
    dt = time.perf_counter() - t0
    v = w["profiled_bases"] * (done_pos / float(w["split_bounds"][-1])) / 1e9 / dt
    return {"value": v, "unit": "Gbp/s", "cores": P, "kind": "python-restatement", "cpu_model": cpu_model(),
            "cgroup_cpu_quota": cgroup_cpus(),
            "sample": "%d of the workload's %d splits (evenly spaced; %.2f Mbp, %d kept observations) in %.1f s on %d processes "
                      "(multiprocessing over splits like profile_controller.py:243-271); oracle/py_columns.py, pileup+SNV call+linkage"
                      % (n_done, len(jobs), done_pos / 1e6, done_obs, dt, P)}


INT8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA peak (spec, no sparsity; MI355X_MICROARCH.md: >= 4404 measured)


def _trace(msg):
    """progress on stderr (ISX_BENCH_TRACE=1): which leg a long run is in"""
    if os.environ.get("ISX_BENCH_TRACE"):
        print("[bench %.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _time_batch(b, warm=2, steps=5):
    for _ in range(warm):
        b.run()
    ts, ks = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        b.run()
        ts.append(time.perf_counter() - t0)
        ks.append(b.pileup_ms())
    return float(np.median(ts)), float(np.mean(ks)), float(np.min(ks))


def linkage_leg(ctx, seed=3):
    """Secondary metric: SNV pairs linked / s on BASELINE configs[2] (C3: 5 Mbp, 200x, 50 000 SNV sites): the read-level
    batch (sparse pair-increment path) and the observation batch with the sparse and the dense int8-MFMA path."""
    _trace("linkage_leg")
    from instrain_amd import engine, synth
    glen = int(os.environ.get("ISX_BENCH_C3_BP", 5_000_000))        # configs[2] in full; smaller = a slice of it (debug)
    meta = synth.Metagenome(1, total_read_bp=200.0 * glen, seed=seed, contigs=1, len_lo=glen, len_hi=glen, abundance_sigma=0.0,
                            min_genome_coverage=0.0, site_frac=0.01, af_lo=0.2, af_hi=0.5)
    w = meta.generate([0])
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    out = {"workload": "C3%s: %.1f Mbp genome, 200x, %d SNV sites (1 / 100 bp, two haplotype backgrounds), skip_mm, linkage on"
                       % ("" if glen == 5_000_000 else " slice", glen / 1e6, glen // 100),
           "kept_observations": int(w["n_obs"]), "read_pairs": int(w["n_pairs"]), "read_segments": int(segs.n_seg)}
    for name, src, pr, mode in (("reads", segs, None, 1), ("sparse", w["obs"], w["pair"], 1), ("dense_mfma", w["obs"], w["pair"], 2)):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, pr, n_mm_bins=1, enable_linkage=True, linkage_mode=mode)
        dt, _, _ = _time_batch(b)
        mf = b.timings()["mfma_ms"]
        s, t = b.sizes(), b.timings()
        b.close()
        r = {"snv_pairs_linked_per_s": s["n_edges"] / dt, "edges": s["n_edges"], "ld_rows": s["n_ld"],
             "pair_increments": s["n_increments"], "allele_observations": s["n_allele_obs"], "ms_per_step": dt * 1e3,
             "kernel_ms": {k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
             "gbp_per_s": w["profiled_bases"] / 1e9 / dt, "record_bytes": t["record_bytes"]}
        if mode == 2:
            tops = 2.0 * t["dense_macs"] / (mf * 1e-3) / 1e12 if mf > 0 else 0.0
            r["mfma"] = {"kernel": "k_dense_gemm (v_mfma_i32_32x32x32_i8)", "tiles": t["dense_tiles"],
                         "int8_macs_per_pass": t["dense_macs"], "xt_bytes": t["dense_bytes"], "pass_ms": mf,
                         "achieved_tops": tops, "peak_tops": INT8_MFMA_PEAK_TOPS, "utilisation": tops / INT8_MFMA_PEAK_TOPS}
        out[name] = r
    out["snv_pairs_linked_per_s"] = out["reads"]["snv_pairs_linked_per_s"]
    return out


def _pmc(key):
    try:
        return json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def _roofline(kernel, ab, k_ms, t, traffic=None, **extra):
    r = {"bound": "hbm", "kernel": kernel, "achieved": ab / (k_ms * 1e-3) / 1e9 if k_ms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0, "traffic": traffic, "algorithmic_bytes_per_launch": ab,
         "record_bytes": t["record_bytes"], "kernel_ms_avg": k_ms, "blocks": t["pileup_blocks"], "threads": t["pileup_threads"],
         "lds_bytes": t["pileup_lds_bytes"], "window": t["pileup_window"]}
    r.update(extra)
    return r


def mm_leg(ctx, w, steps=10):
    """C2 again with mm profiling ON (the reference's default): k_pileup_mm, sparse (pos, mm) entries; read-level batch and
    observation batch."""
    _trace("mm_leg")
    from instrain_amd import engine
    out = {"workload": "C2 with mm profiling on (%d mm bins)" % w["n_mm_bins_mm"]}
    for name, src in (("reads", w["segs_mm"]), ("observations", w["obs_mm"])):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=w["n_mm_bins_mm"], enable_linkage=False)
        dt, k, kmin = _time_batch(b, warm=3, steps=steps)
        s, t = b.sizes(), b.timings()
        b.close()
        n_rec = (w["segs_mm"].n_seg + 15) // 16 * 16
        ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], s["n_entries"], dense=False, record_bytes=t["record_bytes"], n_rec=n_rec)
        out[name] = {"gbp_per_s": w["profiled_bases"] / 1e9 / dt, "ms_per_step": dt * 1e3, "entries": s["n_entries"], "snv_rows": s["n_snv"],
                     "roofline": _roofline("k_pileup_mm", ab, k, t, _pmc("c2_mm_reads_bytes_per_launch" if name == "reads" else "c2_mm_pileup_bytes_per_launch"),
                                           kernel_ms_min=kmin)}
    out["gbp_per_s"] = out["reads"]["gbp_per_s"]
    out["roofline"] = out["reads"]["roofline"]
    return out


def resident_leg(ctx, w, window, steps=30):
    """The kernels alone over resident batches (10 blocking runs each, the dispatch's own time stamps = what rocprofv3
    reports): k_pileup_dense on the read segments (the kernel of the timed step) and on the 2-byte observation records.
    Plus the old kernel-only ceiling: two resident read-level batches passed over alternately, nothing handed over or
    fetched -- NOT what production does."""
    _trace("resident_leg")
    from instrain_amd import engine
    out = {}
    for name, src in (("reads", w["segs"]), ("observations", w["obs"])):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=1, enable_linkage=False, window=window)
        for _ in range(4):
            b.run()
        ks = []
        for _ in range(10):
            b.run()
            ks.append(b.pileup_ms())
        t = b.timings()
        out[name] = {"kernel_ms_alone": float(np.mean(ks)), "kernel_ms_min": float(np.min(ks)), "timings": t}
        if name == "reads":
            ring = [b, engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=1, enable_linkage=False, window=window)]
            ring[1].run()
            t0 = time.perf_counter()
            ring[0].launch()
            for i in range(steps):
                if i + 1 < steps:
                    ring[(i + 1) % 2].launch()
                ring[i % 2].wait()
            dt = time.perf_counter() - t0
            out["gbp_per_s"] = w["profiled_bases"] * steps / dt / 1e9
            out["ms_per_step"] = dt / steps * 1e3
            ring[1].close()
        b.close()
    out["note"] = "resident re-runs (no hand-over, no fetch): kernel-only ceiling"
    return out


def _c5_plan(scale, host_threads):
    from instrain_amd import dist as idist
    from instrain_amd import synth
    n_genomes = max(16, int(round(1000 * scale)))            # scale < 1: debug runs only (reported in the workload string)
    meta = synth.Metagenome(n_genomes, total_read_bp=10e9 * n_genomes / 1000.0, seed=5, threads=max(2, host_threads))
    kept = meta.kept_genomes()
    shards = idist.lpt_shards(meta.pairs[kept], 8)
    return meta, kept, shards, n_genomes


def c5_leg(ctx, rank, world, host_threads, barrier, dist_info, depth=4, with_cpu=True, scale=1.0, stage_async=False):
    """BASELINE.json configs[4]: 1000-genome database, 10 Gbp of reads, --database_mode (one mm bin; genomes
    below 1x dropped like fasta.py:110-136 does).  The kept genomes are LPT-sharded 8 ways on the reference's own cost estimate
    (read pairs, profile_controller.py:460-465); rank r streams the shards r, r + N, ... through its read-level pipe in
    batches of a few genomes, every batch handed over, profiled once (pileup + SNV call + linkage), tables copied back.
    N = 1: the WHOLE kept database through one GPU; N = 8: one shard per GPU (strong scaling of the configuration)."""
    _trace("c5_leg")
    from instrain_amd import dist as idist
    from instrain_amd import engine
    meta, kept, shards, n_genomes = _c5_plan(scale, host_threads)
    my_shards = [s for s in range(8) if s % world == rank % world]
    t0 = time.perf_counter()
    ws = []
    for sh in my_shards:
        mine = kept[shards[sh]]
        est = (meta.pairs[mine] * 2).astype(np.int64)          # segments: one per read
        for b in idist.pack_batches(meta.length[mine], est, 40_000_000, 1_000_000):
            ws.append(meta.generate_segs(mine[b]))
    gen_s = time.perf_counter() - t0
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(w["segs"].n_seg for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=depth, host_threads=host_threads,
                       pin_threads=False, n_mm_bins=1, enable_linkage=True, min_snp=20, stage_async=stage_async)
    stream(pipe, ws[:min(len(ws), 2 * depth)], min(len(ws), 2 * depth), depth)      # warm-up: every slot's tables and linkage buffers reach their steady size
    barrier()
    stats = []
    t0 = time.perf_counter()
    last = stream(pipe, ws, len(ws), depth, stats, keep_last=True)
    barrier()
    dt = time.perf_counter() - t0
    pipe.close()
    bases = float(sum(w["profiled_bases"] for w in ws))
    dt_max, bases_all, gather_ms = dist_info(dt, bases, last)
    st = [s for s, _ in stats]
    tot = lambda k: float(np.sum([s[k] for s in st]))
    n_obs = int(sum(w["n_obs"] for w in ws))
    n_pos = int(sum(w["n_pos"] for w in ws))
    n_rec = int(sum((w["segs"].n_seg + 15) // 16 * 16 for w in ws))
    abytes = pileup_algorithmic_bytes(n_obs, n_pos, 0, dense=True, record_bytes=64, n_rec=n_rec, out_bytes_per_pos=slot_out_bytes_per_pos(n_obs, n_pos), ref_bytes_per_pos=0.5)
    k_ms = tot("kernel_ms")
    out = {"workload": "C5%s: the %d kept genomes of the 1000-genome database (%.2f Gbp of positions, %.2f Gbp of reads in all), --database_mode, "
                       "pileup + SNV call + linkage; this rank: shards %s of 8 (%.2f Gbp of reads) streamed as read segments in %d batches%s"
                       % (" whole configuration through ONE GPU" if world == 1 else " over %d GPUs" % world, len(kept),
                          float(meta.length[kept].sum()) / 1e9, float(meta.pairs[kept].sum()) * 2 * meta.read_len / 1e9, my_shards,
                          bases / 1e9, len(ws), "" if n_genomes == 1000 else " [DEBUG SCALE: %d genomes]" % n_genomes),
           "gbp_per_s": bases_all / dt_max / 1e9, "seconds": dt_max, "n_gpus": world, "scaling": "strong",
           "genomes_kept": int(len(kept)), "genomes_total": n_genomes, "positions": n_pos, "kept_observations": n_obs, "read_segments": n_rec,
           "mean_depth": n_obs / max(n_pos, 1), "snv_rows": int(sum(z["n_snv"] for _, z in stats)),
           "linkage": "on (sparse path; the reference links every profile, linkage.py:14-44)",
           "snv_pairs_linked": int(sum(z["n_edges"] for _, z in stats)), "ld_rows": int(sum(z["n_ld"] for _, z in stats)),
           "snv_pairs_linked_per_s": float(sum(z["n_edges"] for _, z in stats)) * world / dt_max,
           "load_imbalance": float(max(meta.pairs[kept[s]].sum() for s in shards) / np.mean([meta.pairs[kept[s]].sum() for s in shards])),
           "generate_s": gen_s,
           "stages_ms_total": {"host_stage": tot("encode_ms"), "copy_in": tot("h2d_ms"), "kernel": k_ms, "copy_out": tot("d2h_ms"),
                               "collect_wait": tot("collect_wait_ms"), "wall": dt * 1e3},
           "roofline": {"bound": "hbm", "kernel": "k_pileup_dense (read segments)", "achieved": abytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0,
                        "algorithmic_bytes": abytes, "bytes_per_position": abytes / max(n_pos, 1), "kernel_ms_total": k_ms, "launches": len(st),
                        "traffic": None},
           "roofline_pcie": {"bound": "pcie", "direction": "device->host", "achieved": tot("d2h_bytes") / dt / 1e9, "peak": PCIE_PEAK_GBS,
                             "unit": "GB/s", "frac": tot("d2h_bytes") / dt / 1e9 / PCIE_PEAK_GBS, "bytes": tot("d2h_bytes"),
                             "bytes_per_position": tot("d2h_bytes") / max(n_pos, 1), "host_to_device_bytes": tot("h2d_bytes")}}
    if gather_ms is not None:
        out["final_gather_ms"] = gather_ms
    if with_cpu:
        # the CPU baselines on the same configuration: the observation stream of one batch of median size
        order = np.argsort([w["n_obs"] for w in ws])
        sel = ws[int(order[len(order) // 2])]["genomes"]
        wo = meta.generate(sel)
        cb = cpu_baseline(wo, budget_s=12.0, min_s=6.0)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_port"] = out["gbp_per_s"] / cb["value"] if cb["value"] else None
        cp = cpu_baseline_python(wo, n_splits=200, budget_s=15.0)
        out["cpu_baseline_python"] = cp
        out["speedup_vs_python_restatement"] = out["gbp_per_s"] / cp["value"] if cp["value"] else None
    return out


def _s2s(info):
    letters = np.array(list("ACTG"))
    sb = info["scaffold_bounds"]
    return {n: "".join(letters[info["ref_codes"][sb[i]:sb[i + 1]]]) for i, n in enumerate(info["names"])}


def _null_model_dict():
    from tests import util
    lut, fb = util.load_lut()
    nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
    nm[-1] = fb
    return nm


def c5_bam_leg(ctx, host_threads, scale=1.0):
    """The north-star shape through the ENTRY POINT: shard 0 of C5 (1/8 of the kept database: ~85 genomes, ~4 200 contigs) written
    as a coordinate-sorted BAM by the generator, profiled by instrain_amd.profile.profile_bam with a fasta_db of every
    contig's splits (what the controller hands over, profile_controller.py:415-433) in --database_mode."""
    _trace("c5_bam_leg")
    import pandas as pd
    import instrain_amd.profile as amd
    from instrain_amd.profile.profile_utilities import iterate_splits
    meta, kept, shards, n_genomes = _c5_plan(scale, host_threads)
    mine = kept[shards[0]]
    path = "/tmp/isx_c5_shard0_%d.bam" % n_genomes
    t0 = time.perf_counter()
    info = meta.write_bam(mine, path)
    write_s = time.perf_counter() - t0
    s2s = _s2s(info)
    lens = np.diff(info["scaffold_bounds"])
    rows = [(n, i, s, e) for n, L in zip(info["names"], lens) for i, (s, e) in enumerate(iterate_splits(int(L), 10000))]
    fdb = pd.DataFrame(rows, columns=["scaffold", "split_number", "start", "end"])
    best, st_best, n_out = None, None, 0
    for rep in range(2):
        st = {}
        ctx.wait_closers()
        t0 = time.perf_counter()
        out = amd.profile_bam(path, fdb, None, None, s2s=s2s, null_model=_null_model_dict(), ctx=ctx, skip_mm_profiling=True,
                              min_snp=20, stats=st, host_threads=host_threads)
        dt = time.perf_counter() - t0
        n_out = len(out)
        if best is None or dt < best:
            best, st_best = dt, st
        del out
    bam_mb = os.path.getsize(path) / 1e6
    os.remove(path)
    return {"workload": "C5 shard 0 of 8 as a sorted BAM on disk (%d genomes, %d contigs, %.2f Gbp of positions, %.2f Gbp of reads, %.0f MB) -> "
                        "profile_bam(bam, fasta_db of %d split rows, --database_mode) -> SplitObjects"
                        % (len(mine), len(info["names"]), info["n_pos"] / 1e9, info["profiled_bases"] / 1e9, bam_mb, len(fdb)),
            "seconds": best, "gbp_per_s": info["profiled_bases"] / 1e9 / best, "split_objects": n_out, "stages_ms": {k: round(v, 1) for k, v in st_best.items()},
            "bam_write_s": write_s}


def profile_bam_leg(ctx, host_threads):
    """The whole seam on a BAM file: instrain_amd.profile.profile_bam (front end scan + filter + read segments packed into the
    pipe's staging, device batch, table hand-back, SplitObjects) on the 0.9 Gbp probe: a synthetic sorted BAM of 3 M pairs
    2 x 150 bp over one 24 Mbp scaffold, --skip_mm_profiling, and with mm profiling on (the reference's default).  Best of three
    runs each with a warm context; the BAM is written once into /tmp by the generator and is not timed."""
    _trace("profile_bam_leg")
    import instrain_amd.profile as amd
    from instrain_amd import synth
    n_pairs, G = int(os.environ.get("ISX_BENCH_BAM_PAIRS", 3_000_000)), int(os.environ.get("ISX_BENCH_BAM_BP", 24_000_000))
    meta = synth.Metagenome(1, total_read_bp=n_pairs * 300.0, seed=21, contigs=1, len_lo=G, len_hi=G, abundance_sigma=0.0,
                            min_genome_coverage=0.0, site_frac=0.001, threads=max(2, host_threads))
    path = "/tmp/isx_bench_probe_%d_%d.bam" % (n_pairs, G)
    t0 = time.perf_counter()
    info = meta.write_bam([0], path)
    write_s = time.perf_counter() - t0
    s2s = _s2s(info)
    nm = _null_model_dict()
    out = {"workload": "profile_bam end to end: sorted BAM on disk (%.2f Gbp of reads, %d read pairs, one %.0f Mbp scaffold, %.0f MB) -> SplitObjects, linkage on"
                       % (info["profiled_bases"] / 1e9, info["n_pairs"], G / 1e6, os.path.getsize(path) / 1e6), "bam_write_s": write_s}
    for name, skip in (("skip_mm", True), ("mm_on", False)):
        best, st_best, n_splits = None, None, 0
        for rep in range(3):
            st = {}
            ctx.wait_closers()                      # the previous call's pipe and BAM handle go on a helper thread: not into this call's time
            t0 = time.perf_counter()
            res = amd.profile_bam(path, None, None, None, s2s=s2s, null_model=nm, ctx=ctx, skip_mm_profiling=skip, stats=st,
                                  host_threads=host_threads)
            dt = time.perf_counter() - t0
            n_splits = len(res)
            del res
            if best is None or dt < best:
                best, st_best = dt, st
        out[name] = {"seconds": best, "gbp_per_s": info["profiled_bases"] / 1e9 / best, "split_objects": n_splits,
                     "stages_ms": {k: round(v, 1) for k, v in st_best.items()}}
    os.remove(path)
    out["gbp_per_s"] = out["skip_mm"]["gbp_per_s"]
    out["seconds"] = out["skip_mm"]["seconds"]
    out["stages_ms"] = out["skip_mm"]["stages_ms"]
    return out


def bam_sharded_leg(ctx, rank, world, local, host_threads, barrier, device):
    """Multi-GPU over ONE BAM (strong scaling): a slice of BASELINE configs[3] (C4: 12 of its 100 genomes, 50 contigs each, 50x
    mean coverage, log-normal abundances) as one sorted BAM, profiled by dist.profile_bam_sharded -- every rank scans its share
    of the file, the shares' insert sizes are all-gathered for the file-wide median, every rank profiles the scaffolds it owns
    and rank 0 gathers the SNV / linkage / summary tables (grouped point-to-point on RCCL)."""
    _trace("bam_sharded_leg")
    import torch.distributed as tdist
    from instrain_amd import dist as idist
    from instrain_amd import synth
    meta = synth.Metagenome(100, mean_coverage=50, seed=4, threads=max(2, host_threads))
    sel = meta.kept_genomes()[:12]
    path = "/tmp/isx_c4_slice_%d.bam" % len(sel)
    if rank == 0:
        info = meta.write_bam(sel, path)
    else:                                               # the other ranks need the FASTA: the layout + reference without the reads
        info = meta.layout(sel)
        info["profiled_bases"] = int(meta.pairs[sel].sum()) * 2 * meta.read_len
    barrier()
    s2s = _s2s(info)
    st = {}
    barrier()
    t0 = time.perf_counter()
    splits, tables, load = idist.profile_bam_sharded(path, s2s, _null_model_dict(), rank, world, gather=True, device=device,
                                                     ctx=ctx, skip_mm_profiling=True, min_snp=20, stats=st, host_threads=host_threads)
    t_mine = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    loads = [load]
    per_rank = [{"rank": rank, "seconds": t_mine, "load_pairs": load, "split_objects": len(splits), **{k: round(v, 1) for k, v in st.items()}}]
    if world > 1:
        objs = [None] * world
        tdist.all_gather_object(objs, per_rank[0])
        per_rank = objs
        loads = [o["load_pairs"] for o in objs]
    barrier()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return {"workload": "C4 slice: 12 of the 100 genomes (%d contigs, %.2f Gbp of positions, %.2f Gbp of reads) as ONE sorted BAM, sharded over %d rank(s) by "
                        "dist.profile_bam_sharded (share scan + all-gathered insert sizes, --database_mode, linkage on), final gather on rank 0"
                        % (len(info["names"]), info["n_pos"] / 1e9, info["profiled_bases"] / 1e9, world),
            "scaling": "strong", "n_gpus": world, "world_size_seen": world, "backend": tdist.get_backend() if world > 1 else None,
            "seconds": dt, "gbp_per_s": info["profiled_bases"] / 1e9 / dt,
            "load_imbalance": float(max(loads) / max(np.mean(loads), 1.0)), "per_rank": per_rank,
            "gathered_rows": {k: int(len(v)) for k, v in tables.items()} if tables is not None else None}


def make_variants(w, n):
    """n distinct batches of w's shape (synth.shifted_variant_segs), built on a few threads"""
    from concurrent.futures import ThreadPoolExecutor
    from instrain_amd import synth
    with ThreadPoolExecutor(max_workers=max(1, min(8, host_cpus()))) as ex:
        return list(ex.map(lambda k: synth.shifted_variant_segs(w, k), range(n)))


def stream(pipe, variants, n_steps, depth, stats=None, keep_last=False):
    """n_steps batches through the pipe, at most `depth` in flight; returns the last collected result"""
    tickets, done, last = [], 0, None
    link = pipe.enable_linkage

    def take():
        nonlocal done, last
        r = pipe.collect(tickets[done], want_ld=link, densify=False)      # the tables as they come (views of the slot)
        if stats is not None:
            stats.append((r["stats"], r["sizes"]))
        if keep_last and done == n_steps - 1:
            last = {"snv": r["snv"].copy()}
        pipe.release(tickets[done])
        done += 1

    for i in range(n_steps):
        if len(tickets) - done == depth:
            take()
        v = variants[i % len(variants)]
        if pipe.read_level:
            tickets.append(pipe.submit_reads(v["ref_codes"], v["split_bounds"], v["segs"]))
        else:
            tickets.append(pipe.submit(v["ref_codes"], v["split_bounds"], v["obs"], v["pair"] if link else None))
    while done < len(tickets):
        take()
    return last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the C2 genome / the C5 database (debug only; reported in config)")
    ap.add_argument("--variants", type=int, default=32, help="distinct batches cycled through the timed steps")
    ap.add_argument("--depth", type=int, default=4, help="pipe slots")
    ap.add_argument("--host-threads", type=int, default=0, help="staging threads of the pipe (0 = the cpus this rank may use)")
    ap.add_argument("--queued-submit", action="store_true", help="submit_reads only queues the batch, the pipe's stager thread encodes it (isx_pipe_params.stage_async = 1; same-box A/B in profiles/r03_stream_ab.md: no gain on a 16-cpu cgroup, the stager competes with the encoder's own threads)")
    ap.add_argument("--pin", action="store_true", help="bind the staging threads to the L3 domains of the GPU's NUMA node")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the process to the GPU's NUMA node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-linkage-leg", action="store_true")
    ap.add_argument("--no-mm-leg", action="store_true")
    ap.add_argument("--no-resident-leg", action="store_true")
    ap.add_argument("--no-c5-leg", action="store_true")
    ap.add_argument("--no-bam-leg", action="store_true", help="skip the BAM end-to-end legs (profile_bam, c5_bam, bam_sharded)")
    ap.add_argument("--only-c5", action="store_true", help="skip the C2 legs' extras (debug)")
    ap.add_argument("--window", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from instrain_amd import dist as idist
    from instrain_amd import engine
    from tests import util        # only for the committed null-model LUT fixture (data, not oracle code)

    # ISX_DIST_BACKEND=gloo + ISX_DEVICE=0 let the N>1 control flow be exercised on a 1-GPU box (tests only)
    rank, local, world = idist.init_from_env(backend=os.environ.get("ISX_DIST_BACKEND"))
    assert world == max(1, args.gpus) or world == 1, (world, args.gpus)
    if "ISX_DEVICE" in os.environ:
        local = int(os.environ["ISX_DEVICE"])
    use_nccl = world > 1 and dist.get_backend() == "nccl"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if (world == 1 or use_nccl) else torch.device("cpu")
    numa_node = None if args.no_bind else bind_to_gpu_numa_node(torch, local)
    ctx = engine.Context(local)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)

    want_mm = world == 1 and not args.no_mm_leg and not args.only_c5
    w = c2_workload(seed=2 + rank, scale=args.scale, with_mm=want_mm)
    n_var = max(1, min(args.variants, args.steps))
    variants = make_variants(w, n_var)
    host_threads = args.host_threads or max(2, min(48, host_cpus() // world))
    pipe = engine.Pipe(ctx, max_pos=max(v["n_pos"] for v in variants), max_obs=0, max_segs=int(w["segs"].n_seg),
                       max_splits=max(len(v["split_bounds"]) for v in variants), depth=args.depth, host_threads=host_threads,
                       pin_threads=args.pin, n_mm_bins=1, enable_linkage=False, window=args.window, stage_async=args.queued_submit)

    def barrier():
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()

    # Every step hands one batch of read segments over from host memory, profiles it once and brings its tables back;
    # consecutive batches overlap in the pipe's three queues.
    _trace("C2 stream")
    stream(pipe, variants, args.warmup, args.depth)
    barrier()
    torch.cuda.synchronize()
    stats = []
    t0 = time.perf_counter()
    last = stream(pipe, variants[args.warmup % n_var:] + variants[:args.warmup % n_var], args.steps, args.depth, stats, keep_last=world > 1)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        u = torch.tensor([float(w["profiled_bases"])], dtype=torch.float64, device=dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units = float(u.item())
    else:
        units = float(w["profiled_bases"])

    # the one collective of the path: final gather of the SNV tables to rank 0 (outside the timed steps)
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        idist.gather_tables({"snv": last["snv"]}, dst=0, device=dev)
        torch.cuda.synchronize()
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
    pipe.close()

    # configs[4] (C5): N = 1 streams the whole kept database through the one GPU, N = 8 one shard per GPU; rank 0 reports
    c5 = None
    if not args.no_c5_leg:
        def dist_info(dt_c5, bases, last_c5):
            g_ms = None
            if world > 1:
                t = torch.tensor([dt_c5], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                u = torch.tensor([bases], dtype=torch.float64, device=dev)
                dist.all_reduce(u, op=dist.ReduceOp.SUM)
                barrier()
                g0 = time.perf_counter()
                idist.gather_tables({"snv": last_c5["snv"]}, dst=0, device=dev)
                torch.cuda.synchronize()
                barrier()
                g_ms = (time.perf_counter() - g0) * 1e3
                return float(t.item()), float(u.item()), g_ms
            return dt_c5, bases, g_ms
        c5 = c5_leg(ctx, rank, world, host_threads, barrier, dist_info, depth=args.depth,
                    with_cpu=(world == 1 and not args.no_cpu_baseline), scale=args.scale, stage_async=args.queued_submit)

    # one BAM sharded over the ranks (every rank takes part; rank 0 reports)
    sharded = None
    if not args.no_bam_leg and not args.only_c5:
        try:
            sharded = bam_sharded_leg(ctx, rank, world, local, host_threads, barrier, dev)
        except Exception as e:                      # never lose the line over an extra leg
            sharded = {"error": repr(e)}
            if world > 1:
                raise

    if rank == 0:
        st = [s for s, _ in stats]
        mean = lambda k: float(np.mean([s[k] for s in st])) if st else 0.0
        k_ms = mean("kernel_ms")                 # dispatch time stamps of every pass of the timed region
        n_pos_v = int(np.mean([v["n_pos"] for v in variants]))
        n_rec = (int(w["segs"].n_seg) + 15) // 16 * 16
        abytes = pileup_algorithmic_bytes(w["n_obs"], n_pos_v, 0, dense=True, record_bytes=64, n_rec=n_rec,
                                          out_bytes_per_pos=slot_out_bytes_per_pos(w["n_obs"], n_pos_v), ref_bytes_per_pos=0.5)
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        ms_step = dt / args.steps * 1e3
        h2d_b, d2h_b = mean("h2d_bytes"), mean("d2h_bytes")
        out = {
            "metric": "Gbp profiled/s", "value": units * args.steps / dt / 1e9, "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "C2 streamed: one 5 Mbp genome (0.1 Gbp of reads) per batch, 20x, 2x150 bp pairs, insert N(350,30), "
                                   "--skip_mm_profiling (1 mm bin), linkage off; every batch handed over from host memory as READ SEGMENTS "
                                   "(64-byte records: start, length, 150 x 3-bit base codes with the quality filter applied -> pinned "
                                   "hipMemcpyAsync), expanded + profiled once on the device (pileup + SNV call), tables copied back; "
                                   "%d distinct batches" % n_var,
                       "genome_bp": int(w["n_pos"]), "kept_observations": int(w["n_obs"]), "read_segments": int(w["segs"].n_seg),
                       "profiled_bases_per_batch": int(w["profiled_bases"]), "splits": int(len(variants[0]["split_bounds"]) - 1),
                       "distinct_batches": n_var, "pipe_depth": args.depth, "host_threads": host_threads,
                       "process_bound_to_numa_node": numa_node,
                       "parallelism": "scaffold-sharded x%d" % world, "scale": args.scale},
            "roofline": {"bound": "hbm", "kernel": "k_pileup_dense (read segments)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc("c2_reads_bytes_per_launch") if args.scale == 1.0 else None,
                         "algorithmic_bytes_per_launch": abytes, "record_bytes": 64,
                         "kernel_ms_avg": k_ms, "launches": len(st),
                         "note": "durations = the dispatches' own time stamps inside the timed (streamed) region; a pipe slot's kernel "
                                 "writes 6 B/pos (16-bit coverage + clonality; no count table)"},
            "roofline_pcie": {"bound": "pcie", "direction": "host->device", "achieved": h2d_b / (ms_step * 1e-3) / 1e9,
                              "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": h2d_b / (ms_step * 1e-3) / 1e9 / PCIE_PEAK_GBS,
                              "bytes_per_step": h2d_b, "bytes_per_profiled_base": h2d_b / float(w["profiled_bases"]), "copy_ms_avg": mean("h2d_ms"),
                              "during_copy_gbs": h2d_b / (mean("h2d_ms") * 1e-3) / 1e9 if mean("h2d_ms") > 0 else 0.0,
                              "device_to_host": {"bytes_per_step": d2h_b, "copy_ms_avg": mean("d2h_ms"),
                                                 "achieved": d2h_b / (ms_step * 1e-3) / 1e9}},
            "stages_ms": {"host_stage": mean("encode_ms"), "copy_in": mean("h2d_ms"), "kernel": k_ms, "copy_out": mean("d2h_ms"),
                          "collect_wait": mean("collect_wait_ms"), "step": ms_step,
                          "host_bytes_read_per_step": int(w["segs"].n_seg) * 65 + n_pos_v},
            "snv_rows": int(stats[-1][1]["n_snv"]) if stats else 0, "snp_sites": int(stats[-1][1]["n_sites"]) if stats else 0,
        }
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if c5 is not None:
            out["c5"] = c5
        if sharded is not None:
            out["bam_sharded"] = sharded
        if args.only_c5:
            args.no_resident_leg = args.no_mm_leg = args.no_linkage_leg = args.no_cpu_baseline = True
            want_mm = False
        if world == 1 and not args.no_resident_leg:
            res = resident_leg(ctx, w, args.window)
            out["resident"] = res
            # The roofline of the dominant kernel is priced on the kernel having the GPU to itself (10 blocking runs over a
            # resident read-level C2 batch, the dispatch's own time stamps = what rocprofv3 reports): in the streamed region a
            # launch is a few % of a PCIe-bound step, the GPU idles between launches and its clocks sag (kernel_ms_in_stream).
            r = out["roofline"]
            k_alone = res["reads"]["kernel_ms_alone"]
            tr = res["reads"]["timings"]
            ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, dense=True, record_bytes=64, n_rec=n_rec, out_bytes_per_pos=20)
            r.update({"kernel_ms_in_stream": r["kernel_ms_avg"], "frac_in_stream": r["frac"], "kernel_ms_avg": k_alone,
                      "kernel_ms_min": res["reads"]["kernel_ms_min"],
                      "algorithmic_bytes_per_launch": ab, "achieved": ab / (k_alone * 1e-3) / 1e9,
                      "frac": ab / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "blocks": tr["pileup_blocks"], "threads": tr["pileup_threads"], "lds_bytes": tr["pileup_lds_bytes"], "window": tr["pileup_window"],
                      "gbs_at_8_bytes_per_observation": pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, True, 8) / (k_alone * 1e-3) / 1e9,
                      "note": "kernel alone over a resident read-level C2 batch (dispatch time stamps, 10 blocking runs): 64 B per read segment + "
                              "4 B per 16 of them + 1 B/pos in, 20 B/pos out; kernel_ms_in_stream = the same kernel inside the PCIe-bound "
                              "streamed region.  With ~0.43 B per base in, the kernel is no longer bound by the stream but by its "
                              "LDS read-modify-writes (one per kept base, see roofline_lds) and by the 20 B/pos it writes"})
            out["roofline_lds"] = {"bound": "lds-atomic", "kernel": r["kernel"], "achieved": w["n_obs"] / (k_alone * 1e-3) / 1e12,
                                   "peak": LDS_ATOMIC_PEAK / 1e12, "unit": "T atomics/s", "frac": w["n_obs"] / (k_alone * 1e-3) / LDS_ATOMIC_PEAK,
                                   "atomics_per_launch": int(w["n_obs"]),
                                   "note": "one ds_add_u32 lane per kept base; peak = 256 CUs x 16 lanes/clk x 2.4 GHz (conflict-free)"}
            ko = res["observations"]["kernel_ms_alone"]
            to = res["observations"]["timings"]
            abo = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, dense=True, record_bytes=to["record_bytes"])
            out["roofline_observation_kernel"] = _roofline("k_pileup_dense (2-byte observation records; isx_pipe_submit / isx_batch_create)", abo, ko, to,
                                                           _pmc("c2_pileup_bytes_per_launch"), kernel_ms_min=res["observations"]["kernel_ms_min"])
        if want_mm:
            out["mm_on"] = mm_leg(ctx, w)
        if world == 1 and not args.no_linkage_leg:
            out["linkage"] = linkage_leg(ctx)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            out["cpu_baseline_python"] = cpu_baseline_python(w)
        if world == 1 and not args.no_bam_leg and not args.only_c5:
            for key, fn in (("profile_bam", lambda: profile_bam_leg(ctx, host_threads)), ("c5_bam", lambda: c5_bam_leg(ctx, host_threads, args.scale))):
                try:
                    out[key] = fn()
                except Exception as e:                  # never lose the line over an extra leg
                    out[key] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
