#!/usr/bin/env python3
"""bench.py -- headline measurement of the `inStrain profile` hot path on MI355X.

The headline is BASELINE.json configs[4], the configuration north_star quotes its target on: the
1000-genome database with 10 Gbp of reads, --database_mode, pileup + SNV call + linkage.  A STEP = one pass over the
rank's share of the kept database (N = 1: all of it, 25 batches of <= 120 Mbp), every batch handed over INSIDE the step from
the caller's own (pageable) arrays -- its reads as bit planes (isx_read_planes: one 64-byte line a read), the reference as
its 2-bit plane -- through isx_pipe_submit_planes: the pipe's host threads copy the reference planes into pinned staging and
make the 32-byte reference-delta wire records by XOR against them, hipMemcpyAsync brings them in, the batch is profiled ONCE
on the device (k_pileup_dense: difference-array pileup in LDS, SNV call epilogue, linkage chain) and its tables are copied
back to pinned host memory (isx_pipe_collect); staging / copy-in / pass / copy-out of consecutive batches overlap in the
pipe.  (Round 4's headline replayed pre-staged pinned images: that is the extra `c5_staged_replay_gbp_per_s` now.)  Before the timed steps ONE untimed pass checks every batch's
tables on the host (C5Run.verify_pass); every timed batch's row counts must equal that pass's.  N > 1: the 8 LPT shards
of the database are dealt over the ranks (strong scaling; no data-path collective; one final RCCL gather of the SNV
tables after the timed region, reported separately).  `python bench.py --gpus N` without a launcher starts its N ranks
itself (torch.distributed.run on 127.0.0.1).

Prints ONE short JSON line on rank 0 (contract in the task statement: metric / value / ... / config / roofline /
cpu_baseline + cpu_baseline_python on the same configuration) and writes the full record -- every leg's detail -- to
bench_detail.json (--detail): `c5` (the headline in full), `c2_stream` (configs[1] streamed, the round-3 headline),
`roofline_c2_resident` / `roofline_lds` / `roofline_observation_kernel` (the kernels alone over resident C2 batches),
`mm_on`, `linkage` (C3), `bam_sharded` (one BAM over the ranks), `profile_bam` / `c5_bam` (BAM on disk -> SplitObjects),
`cpu_baseline_c2` / `cpu_baseline_python_c2`.
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
    b = (n_rec * 64 + n_rec // 16 * 4) if record_bytes == 64 else ((n_rec * 32 + n_rec // 32 * 4) if record_bytes == 32 else n_obs * record_bytes)
    b += int(n_pos * ref_bytes_per_pos)         # (a pipe slot holds the reference two codes per byte)
    b += n_pos * out_bytes_per_pos if dense else n_entries * 32
    return b


# isx_pipe_params.lean_output (a slot's kernel writes only what travels home: 1-2 instead of 6-7 bytes a position) is built and tested,
# but moves no clock: the kernel is bound by its per-window latency chain, not by its stores (C5: 1.24-1.29 ms per launch either
# way, same box) -- so the bench keeps the plain slots whose dense arrays the device summaries can read.
LEAN_SLOTS = bool(int(os.environ.get("ISX_BENCH_LEAN_SLOTS", "1")))
C5_RESERVE_CUS = int(os.environ.get("ISX_BENCH_C5_RESERVE_CUS", "4"))
REGISTER_REF = bool(int(os.environ.get("ISX_BENCH_REGISTER_REF", "1")))    # the workloads' reference planes registered once for the copy engine (isx_host_register); 0 = through staging
C5_LINKAGE = bool(int(os.environ.get("ISX_BENCH_C5_LINKAGE", "1")))      # 0: diagnostic only (what the linkage chain costs the stream); not the workload


def slot_out_bytes_per_pos(n_obs, n_pos, lean=None):
    """What k_pileup_dense writes per position in a pipe slot without want_counts (the shrunk hand-back).  A lean slot
    (isx_pipe_params.lean_output, what bench.py's pipes are): the coverage that travels home -- 1 byte for a shallow batch (mean
    depth < 16), 2 otherwise -- and the sparse lists (clonalities other than 1.0: 8 bytes an entry, a few % of the positions;
    saturated coverages, SNV rows).  A plain slot also writes the dense fp32 clonality and the 16-bit coverage beside the 8-bit
    one: 7 / 6 bytes."""
    shallow = n_obs < 16 * n_pos
    if lean is None:
        lean = LEAN_SLOTS
    if lean and n_obs < 6 * n_pos:
        return 0.5                  # 4-bit coverage plane (the 16-bit rows of the windows beyond 15 not counted)
    return (1 if shallow else 2) if lean else (7 if shallow else 6)


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


def cgroup_throttled_ms():
    """time the cgroup's cpu quota has held this container's threads back so far (cpu.stat throttled_usec), None when not readable"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k == "throttled_usec":
                    return int(v) / 1e3
                if k == "throttled_time":
                    return int(v) / 1e6
        except Exception:
            pass
    return None


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


def cpu_baseline_python(w, n_splits=200, budget_s=22.0, min_s=10.0):
    """The reference-like CPU baseline SURVEY 8(d) asks for: a faithful per-column PYTHON restatement of the reference's
    split worker (oracle/py_columns.py: dict-of-numpy-arrays count tables filled read by read, Python arithmetic per
    (position, mm), per-read SNV lists, dict-of-dicts linkage network) with multiprocessing over splits exactly like the
    reference's worker pool (profile_controller.py:243-271), P = the cpus this process may use, on a subsample of the SAME
    workload's splits (evenly spaced), swept again and again until min_s of wall time have passed (a second of work moved the
    figure by 25 % run to run) and stopped after budget_s; the rate is extrapolated linearly (cost is per split)."""
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
        sweeps, stop = 0, False
        while not stop:
            for p, o in pool.imap_unordered(_py_split, sample):
                done_pos += p; done_obs += o; n_done += 1
                if time.perf_counter() - t0 > budget_s:
                    pool.terminate()
                    stop = True
                    break
            sweeps += 1
            if time.perf_counter() - t0 >= min_s:
                stop = True
    dt = time.perf_counter() - t0
    v = w["profiled_bases"] * (done_pos / float(w["split_bounds"][-1])) / 1e9 / dt
    return {"value": v, "unit": "Gbp/s", "cores": P, "kind": "python-restatement", "cpu_model": cpu_model(),
            "cgroup_cpu_quota": cgroup_cpus(),
            "sample": "%d split profiles (%d of the workload's %d splits, evenly spaced, swept %d times; %.2f Mbp, %d kept observations) in %.1f s on %d processes "
                      "(multiprocessing over splits like profile_controller.py:243-271); oracle/py_columns.py, pileup+SNV call+linkage"
                      % (n_done, len(sample), len(jobs), sweeps, done_pos / 1e6, done_obs, dt, P)}


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


def _register_refs(vs):
    try:
        for v in vs:
            v["ref_planes"].register()
    except Exception as e:                      # (a lease that may not pin: through staging)
        print("bench: reference planes not registered (%s)" % e, file=sys.stderr)
        _unregister_refs(vs)


def _unregister_refs(vs):
    for v in vs:
        if "ref_planes" in v:
            v["ref_planes"].unregister()


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
    # "reads": the read-level batch through the bucket chain (round 6: per-pair chains, a bucket of increments per first site, no sort library);
    # "reads_sorted_chain": the same batch through the device-wide sorts of rounds 2-5 (ISX_LINK_CHAIN=sorted, read at every call) -- the A/B of the line
    for name, src, pr, mode in (("reads", segs, None, 1), ("reads_sorted_chain", segs, None, 1), ("sparse", w["obs"], w["pair"], 1), ("dense_mfma", w["obs"], w["pair"], 2)):
        if name == "reads_sorted_chain":
            if os.environ.get("ISX_LINK_CHAIN"):
                continue                    # (the caller chose a chain for the whole run: nothing to compare)
            os.environ["ISX_LINK_CHAIN"] = "sorted"
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, pr, n_mm_bins=1, enable_linkage=True, linkage_mode=mode)
        try:
            dt, _, _ = _time_batch(b)
        finally:
            if name == "reads_sorted_chain":
                del os.environ["ISX_LINK_CHAIN"]
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


def _kernel_source_sha():
    """identity of the kernels the PMC passes were collected on: sha1 over the pileup kernel source"""
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(REPO, "instrain_amd", "csrc", "isx_pileup.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _pmc_file():
    try:
        return json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}


def _pmc(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/make_profiles.sh) -- None when the kernel source
    changed since the passes were collected (a stale figure is not a measurement)."""
    j = _pmc_file()
    if j.get("kernel_source_sha") != _kernel_source_sha():
        return None
    return j.get(key)


def _pmc_source(key):
    j = _pmc_file()
    if key not in j:
        return None
    stale = j.get("kernel_source_sha") != _kernel_source_sha()
    return "profiles/pmc_traffic.json: rocprofv3 --pmc passes of tag %s at commit %s%s" % (
        j.get("tag"), j.get("commit"), " -- STALE (isx_pileup.hip changed since), traffic withheld" if stale else "")


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
    # reads: 64-byte segment records (the default with mm profiling on); reads_delta_records: 32-byte reference-delta records with the level in
    # the header (ISX_LAYOUT_MM_DELTA_RECORDS, round 6: per-level difference rows + a materialise phase); observations: 4-byte records
    for name, src, layout in (("reads", w["segs_mm"], 0), ("reads_delta_records", w["segs_mm"], 32), ("observations", w["obs_mm"], 0)):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=w["n_mm_bins_mm"], enable_linkage=False, layout=layout)
        dt, k, kmin = _time_batch(b, warm=3, steps=steps)
        s, t = b.sizes(), b.timings()
        b.close()
        n_seg = int(w["segs_mm"].n_seg)
        n_rec = (n_seg + n_seg // 4096 * 32 + 31) // 32 * 32 if t["record_bytes"] == 32 else (n_seg + 15) // 16 * 16
        ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], s["n_entries"], dense=False, record_bytes=t["record_bytes"], n_rec=n_rec)
        out[name] = {"gbp_per_s": w["profiled_bases"] / 1e9 / dt, "ms_per_step": dt * 1e3, "entries": s["n_entries"], "snv_rows": s["n_snv"],
                     "roofline": _roofline("k_pileup_mm", ab, k, t, _pmc({"reads": "c2_mm_reads_bytes_per_launch", "observations": "c2_mm_pileup_bytes_per_launch", "reads_delta_records": "c2_mm_delta_bytes_per_launch"}[name]),
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


# C5 batches: genomes are packed into device batches under these budgets (the reference groups its profile commands by estimated
# cost the same way, profile_controller.py:397-457).  A batch pays ~100 short kernel launches and a handful of host syncs in its
# linkage / hand-back chain whatever its size, so batches are made as large as the pipe's slots comfortably hold.
C5_BATCH_POS = int(os.environ.get("ISX_BENCH_C5_BATCH_POS", 120_000_000))
C5_BATCH_SEGS = int(os.environ.get("ISX_BENCH_C5_BATCH_SEGS", 3_000_000))


def _c5_plan(scale, host_threads):
    from instrain_amd import dist as idist
    from instrain_amd import synth
    n_genomes = max(16, int(round(1000 * scale)))            # scale < 1: debug runs only (reported in the workload string)
    meta = synth.Metagenome(n_genomes, total_read_bp=10e9 * n_genomes / 1000.0, seed=5, threads=max(2, host_threads))
    kept = meta.kept_genomes()
    shards = idist.lpt_shards(meta.pairs[kept], 8)
    return meta, kept, shards, n_genomes


def planes_coverage(w, per_base_at=None, chunk=200_000):
    """exact per-position coverage of a bit-plane workload, on the host (numpy, in chunks): segment starts +1 / ends -1, prefix sum,
    minus the columns a segment does not observe; per_base_at = sorted positions -> also the (A, C, T, G) counts at those positions.
    The check of the tables, independent of the library: only the format's definition (include/instrain_amd.h isx_read_planes)."""
    pb, n_pos = w["planes"], int(w["n_pos"])
    g0 = pb.gpos.astype(np.int64)
    ln = pb.len.astype(np.int64)
    diff = np.bincount(g0, minlength=n_pos + 1).astype(np.int64) - np.bincount(g0 + ln, minlength=n_pos + 1).astype(np.int64)
    cov = np.cumsum(diff[:-1])
    per_base = np.zeros((len(per_base_at), 4), np.int64) if per_base_at is not None else None
    j = np.arange(150, dtype=np.int64)[None, :]
    for c0 in range(0, pb.n_seg, chunk):
        pl = pb.planes[c0:c0 + chunk]
        skip = np.unpackbits(np.ascontiguousarray(pl[:, 5:8]).view(np.uint8), axis=1, bitorder="little")[:, :150].astype(bool)
        inside = j < ln[c0:c0 + chunk, None]
        g = g0[c0:c0 + chunk, None] + j
        cov -= np.bincount(g[inside & skip], minlength=n_pos)
        if per_base is not None:
            ok = inside & ~skip
            two = np.unpackbits(np.ascontiguousarray(pl[:, 0:5]).view(np.uint8), axis=1, bitorder="little")[:, :300].reshape(len(pl), 150, 2)
            code = two[:, :, 0] | (two[:, :, 1] << 1)
            gg, bb = g[ok], code[ok]
            k = np.searchsorted(per_base_at, gg)
            hit = (k < len(per_base_at)) & (per_base_at[np.minimum(k, len(per_base_at) - 1)] == gg)
            np.add.at(per_base, (k[hit], bb[hit]), 1)
    return cov, per_base


class C5Run:
    """BASELINE.json configs[4] (the configuration north_star quotes its target on): 1000-genome database,
    10 Gbp of reads, --database_mode (one mm bin; genomes below 1x dropped like fasta.py:110-136 does).  The kept genomes are
    LPT-sharded 8 ways on the reference's own cost estimate (read pairs, profile_controller.py:460-465); rank r takes the
    shards r, r + N, ... and streams them through its read-level pipe in batches of a few genomes, every batch handed over
    from host memory, profiled once (pileup + SNV call + linkage), tables copied back.  N = 1: the WHOLE kept database
    through one GPU; N = 8: one shard per GPU (strong scaling of the configuration).  One STEP = one pass over all of the
    rank's batches."""

    def __init__(self, ctx, rank, world, host_threads, depth=4, scale=1.0, stage_async=False, min_batches=0):
        from instrain_amd import dist as idist
        from instrain_amd import engine
        self.ctx, self.rank, self.world, self.depth = ctx, rank, world, depth
        self.meta, self.kept, self.shards, self.n_genomes = _c5_plan(scale, host_threads)
        meta, kept = self.meta, self.kept
        self.my_shards = [s for s in range(8) if s % world == rank % world]
        t0 = time.perf_counter()
        self.ws = []
        # batch budget: 120 Mbp / 3 M segments whatever the job's size.  (min_batches > 0 shrinks it so that a rank of an N-GPU job keeps
        # that many batches a pass: measured as a LOSS -- one shard of 8 in 4 batches 7.3 ms a pass, in 11 batches 10.0 ms,
        # tools/c5_rank_probe.py -- a batch's ~60 short launches cost more than a shorter drain saves)
        my_pos = int(sum(meta.length[kept[self.shards[sh]]].sum() for sh in self.my_shards))
        shrink = min(1.0, max(0.25, my_pos / float(min_batches) / C5_BATCH_POS)) if min_batches else 1.0
        batch_pos, batch_segs = int(C5_BATCH_POS * shrink), int(C5_BATCH_SEGS * shrink)
        for sh in self.my_shards:
            mine = kept[self.shards[sh]]
            est = (meta.pairs[mine] * 2).astype(np.int64)          # segments: one per read
            for b in idist.pack_batches(meta.length[mine], est, batch_pos, batch_segs):
                w = meta.generate_segs(mine[b])
                # What the caller holds and hands over (include/instrain_amd.h isx_read_planes / isx_ref_planes): its reads as bit planes
                # -- one 64-byte line a read: 2-bit base codes + the plane of columns that are not observed -- and the database's
                # reference as the 2-bit plane it travels as.  Plain (pageable) memory of this process, written before the timed region,
                # exactly where round 4 generated its isx_segs arrays.
                w["planes"] = engine.PlaneBatch.from_segs(w["segs"], threads=host_threads)
                w["ref_planes"] = engine.RefPlanes.from_codes(w["ref_codes"], threads=host_threads, key=0x1000 + len(self.ws) + 4096 * sh)
                w["n_seg"] = int(w["segs"].n_seg)
                del w["segs"], w["ref_codes"]
                self.ws.append(w)
        self.ws.sort(key=lambda w: -w["n_pos"])                     # largest first: the pass drains behind its smallest batch
        self.gen_s = time.perf_counter() - t0
        # The database's reference planes are what a caller keeps for the whole run (profile_controller.py:415-433 holds the fasta the same
        # way): registered once for the copy engine (isx_host_register), they are copied to the device from where they lie -- they still travel
        # with every batch of every pass, the pipe's threads just do not copy them into staging first.  The reads stay plain pageable memory.
        # ISX_BENCH_REGISTER_REF=0: as before round 6's last part (the A/B of the line).
        self.ref_registered = False
        if REGISTER_REF:
            t_r = time.perf_counter()
            try:
                for w in self.ws:
                    w["ref_planes"].register()
                self.ref_registered = True
            except Exception as e:                                  # (a lease that may not pin that much: the planes go through staging)
                print("bench: reference planes not registered (%s)" % e, file=sys.stderr)
                for w in self.ws:
                    w["ref_planes"].unregister()
            self.register_s = time.perf_counter() - t_r
        ws = self.ws
        self.stage_async = stage_async
        self.pipe = self.make_pipe(host_threads)
        self.bases = float(sum(w["profiled_bases"] for w in ws))
        self.signature = None
        self.wires = None
        self.stage_s = None

    def make_pipe(self, host_threads):
        from instrain_amd import engine
        ws = self.ws
        return engine.Pipe(self.ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(w["n_seg"] for w in ws),
                           max_splits=max(len(w["split_bounds"]) for w in ws), depth=self.depth, host_threads=host_threads,
                           pin_threads=bool(os.environ.get("ISX_BENCH_PIN")), n_mm_bins=1, enable_linkage=C5_LINKAGE, min_snp=20, stage_async=self.stage_async, lean_output=LEAN_SLOTS,
                           layout=int(os.environ.get("ISX_BENCH_LAYOUT", "0")))     # (A/B: 64 = ISX_LAYOUT_NO_STRIPES)

    def stage_all(self):
        """every batch staged once into a pinned image (isx_pipe_stage_planes): what round 4's headline replayed.  An extra of the line
        now -- the headline's steps do this work inside the step"""
        t0 = time.perf_counter()
        self.wires = [self.pipe.stage_planes(w["ref_planes"], w["split_bounds"], w["planes"]) for w in self.ws]
        self.stage_s = time.perf_counter() - t0

    def verify_pass(self):
        """One untimed pass in which every batch's tables are CHECKED on the host (size-independent properties): the coverage
        table sums to the kept observations handed over, SNV rows are strictly ordered by position with counts that sum to
        the coverage at their position and reach min_cov, LD rows' four counts add up to their total.  Returns and remembers
        the per-batch signature (n_snv, n_ld, n_edges) the timed passes are compared with."""
        sig = []
        covered = {}
        exact_i = int(np.argmax([w["n_pos"] for w in self.ws]))

        def check(i, r):
            from instrain_amd import engine
            w = self.ws[i]
            cov = engine.dense_cov(r, w["n_pos"]).astype(np.int64) if "cov4" in r else (r["cov16"] if "cov16" in r else r["cov8"]).astype(np.int64)
            lim = 65535 if ("cov16" in r or "cov4" in r) else 255
            total = int(cov.sum())
            if "saturated" in r:
                s = r["saturated"]
                total += int((s["coverage"].astype(np.int64) - lim).sum())
                cov[s["gpos"]] = s["coverage"]
            if total != w["n_obs"]:
                raise AssertionError("C5 batch %d: coverage table sums to %d, %d observations were handed over" % (i, total, w["n_obs"]))
            covered[i] = int(np.count_nonzero(cov))
            if i == exact_i:
                # the largest batch: EXACT per-position coverage and the SNV rows' counts base by base, recomputed on the host from
                # the planes that were handed over (planes_coverage: numpy on the format's definition, nothing of the library)
                exp, per_base = planes_coverage(w, per_base_at=r["snv"]["gpos"].astype(np.int64))
                if not (cov == exp).all():
                    raise AssertionError("C5 batch %d: coverage differs from the host's at %d positions" % (i, int((cov != exp).sum())))
                if not (per_base == r["snv"]["cnt"]).all():
                    raise AssertionError("C5 batch %d: SNV row counts differ from the host's per-base counts" % i)
                self.exact_checked = {"batch": int(i), "positions": int(len(cov)), "snv_rows": int(len(per_base))}
            snv = r["snv"]
            g = snv["gpos"].astype(np.int64)
            if len(g) != r["sizes"]["n_snv"] or (np.diff(g) <= 0).any():
                raise AssertionError("C5 batch %d: SNV rows not strictly ordered by position" % i)
            c = snv["cnt"].sum(axis=1).astype(np.int64)
            if (c != cov[g]).any() or (c < 5).any():
                raise AssertionError("C5 batch %d: SNV row counts do not sum to the coverage at their position" % i)
            ld = r["ld"]
            if len(ld) and (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] != ld["total"]).any():
                raise AssertionError("C5 batch %d: LD row counts do not add up" % i)
            # (+ the library's checksum over the bytes of the batch's SNV and LD rows: the rows checked above, byte for byte)
            sig.append((int(r["sizes"]["n_snv"]), int(r["sizes"]["n_ld"]), int(r["sizes"]["n_edges"]), int(r["rows_checksum"])))

        stream(self.pipe, self.ws, len(self.ws), self.depth, check=check)
        self.signature = sig
        self.covered = int(sum(covered.values()))           # positions with coverage (E of SURVEY 8(d)'s byte model with one mm bin)
        return sig

    def checked_pass(self, check, staged=False):
        """one pass with check(i, result) called on every collected batch (tests/test_gpu_metagenome.py drives the headline's exact
        configuration through this)"""
        if staged and self.wires is None:
            self.stage_all()
        stream(self.pipe, self.ws, len(self.ws), self.depth, check=check, wires=self.wires if staged else None)

    def run(self, passes, stats=None, keep_last=False, staged=False):
        """passes over the rank's batches.  staged False (the headline): every batch handed over from the caller's arrays --
        isx_pipe_submit_planes: reference planes copied into pinned staging, records made by the XOR pass, DMA, kernels, tables back, all
        inside the step.  staged True: the pinned images of stage_all() replayed (DMA + kernels + tables back)."""
        n = len(self.ws)
        return stream(self.pipe, self.ws, n * passes, self.depth, stats, keep_last=keep_last, wires=self.wires if staged else None)

    def gather_pass(self):
        """one untimed pass whose tables are KEPT: what this rank contributes to the path's one collective (the final gather of the
        profile on rank 0, SURVEY 8e / profile_controller.py:195-233): every batch's SNV rows and LD rows (flat positions made unique
        over the rank by the batch's offset in the rank's own flat space) and a per-scaffold coverage summary (covered positions, summed
        coverage: what the merge step's cumulative tables start from)"""
        from instrain_amd import engine
        snv, ld, summ = [], [], []
        off = np.r_[0, np.cumsum([w["n_pos"] for w in self.ws])].astype(np.int64)
        summ_dt = np.dtype([("genome", "<i4"), ("scaffold", "<i4"), ("length", "<i8"), ("covered", "<i8"), ("sum_cov", "<i8")])

        def keep(i, r):
            w = self.ws[i]
            a = np.zeros(len(r["snv"]), dtype=[("rank_pos", "<i8")] + r["snv"].dtype.descr)
            for k in r["snv"].dtype.names:
                a[k] = r["snv"][k]
            a["rank_pos"] = r["snv"]["gpos"].astype(np.int64) + off[i]
            snv.append(a)
            b = np.zeros(len(r["ld"]), dtype=[("rank_pos_a", "<i8")] + r["ld"].dtype.descr)
            for k in r["ld"].dtype.names:
                b[k] = r["ld"][k]
            b["rank_pos_a"] = r["ld"]["gpos_a"].astype(np.int64) + off[i]
            ld.append(b)
            cov = engine.dense_cov(r, w["n_pos"]).astype(np.int64)
            sb = w["scaffold_bounds"]
            t = np.zeros(len(sb) - 1, dtype=summ_dt)
            t["genome"], t["scaffold"], t["length"] = w["scaffold_genome"], np.arange(len(sb) - 1), np.diff(sb)
            t["sum_cov"] = np.add.reduceat(cov, sb[:-1])
            t["covered"] = np.add.reduceat((cov > 0).astype(np.int64), sb[:-1])
            summ.append(t)

        stream(self.pipe, self.ws, len(self.ws), self.depth, check=keep)
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt)
        return {"snv": cat(snv, snv[0].dtype if snv else np.dtype("u1")), "ld": cat(ld, ld[0].dtype if ld else np.dtype("u1")), "summary": cat(summ, summ_dt)}

    def staged_replay(self, passes=4):
        """round 4's headline as an extra: batches staged once (untimed), a pass = DMA + kernels + tables back"""
        if self.wires is None:
            self.stage_all()
        self.run(1, staged=True)
        stats = []
        t0 = time.perf_counter()
        self.run(passes, stats, staged=True)
        dt = (time.perf_counter() - t0) / passes
        self.check_timed(stats)
        st = [x for x, _ in stats]
        return {"gbp_per_s": self.bases / dt / 1e9, "ms_per_pass": dt * 1e3, "passes": passes, "stage_once_s": self.stage_s,
                "copy_in_ms_per_pass": float(np.sum([x["h2d_ms"] for x in st])) / passes, "kernel_ms_per_pass": float(np.sum([x["kernel_ms"] for x in st])) / passes}

    def resident_reference_passes(self, passes=4):
        """the same passes -- every batch handed over inside the step -- with the database's reference planes RESIDENT on the device by
        key (isx_ref_planes.key: kept after their first trip): what a service that profiles sample after sample against ONE database
        runs -- the reference is the same for every sample, only the reads are new.  An extra of the line, not its value: the
        headline hands every batch over whole"""
        n = len(self.ws)
        stream(self.pipe, self.ws, n, self.depth, keyed=True)           # first trip: the planes travel and stay
        stats = []
        t0 = time.perf_counter()
        stream(self.pipe, self.ws, n * passes, self.depth, stats, keyed=True)
        dt = (time.perf_counter() - t0) / passes
        self.check_timed(stats)
        st = [x for x, _ in stats]
        return {"gbp_per_s": self.bases / dt / 1e9, "ms_per_pass": dt * 1e3, "passes": passes,
                "hand_over": "isx_pipe_submit_planes inside the step, reference resident by key",
                "host_stage_ms_per_pass": float(np.sum([x["encode_ms"] for x in st])) / passes,
                "copy_in_ms_per_pass": float(np.sum([x["h2d_ms"] for x in st])) / passes,
                "h2d_bytes_per_pass": float(np.sum([x["h2d_bytes"] for x in st])) / passes,
                "h2d_bytes_per_base": float(np.sum([x["h2d_bytes"] for x in st])) / passes / max(self.bases, 1.0)}

    def check_timed(self, stats):
        """every timed batch produced the tables of the verified pass: row counts AND the checksum over the bytes of its SNV + LD rows
        (made by the pipe's finisher when the rows land, isx_pipe_result.rows_checksum; the stream is deterministic)"""
        n = len(self.ws)
        for i, (_, z) in enumerate(stats):
            if (int(z["n_snv"]), int(z["n_ld"]), int(z["n_edges"]), int(z["rows_checksum"])) != self.signature[i % n]:
                raise AssertionError("C5 timed batch %d differs from the verified pass: %r vs %r"
                                     % (i, (z["n_snv"], z["n_ld"], z["n_edges"], z["rows_checksum"]), self.signature[i % n]))

    def close(self):
        self.pipe.close()
        if getattr(self, "ref_registered", False):
            for w in self.ws:
                w["ref_planes"].unregister()
            self.ref_registered = False

    def report(self, dt_max, bases_all, stats, passes, gather_ms=None):
        meta, kept, ws, world = self.meta, self.kept, self.ws, self.world
        st = [s for s, _ in stats]
        tot = lambda k: float(np.sum([s[k] for s in st])) / passes          # per pass
        n_obs = int(sum(w["n_obs"] for w in ws))
        n_pos = int(sum(w["n_pos"] for w in ws))
        n_rec = int(sum(w["n_seg"] for w in ws))
        rbytes = int(st[0]["record_bytes"]) if st else 64
        h2d = tot("h2d_bytes")
        # what a launch has to move: the stream as it lies in HBM (= what crossed PCIe: records, group bases, pair ids, packed
        # reference, directory) + what the epilogue writes per position
        abytes = h2d + n_pos * slot_out_bytes_per_pos(n_obs, n_pos)
        k_ms = tot("kernel_ms")
        one = stats[:len(ws)]
        out = {"workload": "C5%s: the %d kept genomes of the 1000-genome database (%.2f Gbp of positions, %.2f Gbp of reads in all), --database_mode, "
                           "pileup + SNV call + linkage; this rank: shards %s of 8 (%.2f Gbp of reads) streamed as read segments in %d batches%s"
                           % (" whole configuration through ONE GPU" if world == 1 else " over %d GPUs" % world, len(kept),
                              float(meta.length[kept].sum()) / 1e9, float(meta.pairs[kept].sum()) * 2 * meta.read_len / 1e9, self.my_shards,
                              self.bases / 1e9, len(ws), "" if self.n_genomes == 1000 else " [DEBUG SCALE: %d genomes]" % self.n_genomes),
               "gbp_per_s": bases_all * passes / dt_max / 1e9, "seconds_per_pass": dt_max / passes, "n_gpus": world, "scaling": "strong",
               "genomes_kept": int(len(kept)), "genomes_total": self.n_genomes, "positions": n_pos, "kept_observations": n_obs, "read_segments": n_rec,
               "batches": len(ws), "mean_depth": n_obs / max(n_pos, 1), "snv_rows": int(sum(z["n_snv"] for _, z in one)),
               "linkage": "on (sparse path; the reference links every profile, linkage.py:14-44)",
               "snv_pairs_linked": int(sum(z["n_edges"] for _, z in one)), "ld_rows": int(sum(z["n_ld"] for _, z in one)),
               "snv_pairs_linked_per_s": float(sum(z["n_edges"] for _, z in one)) * world * passes / dt_max,
               "load_imbalance": float(max(meta.pairs[kept[s]].sum() for s in self.shards) / np.mean([meta.pairs[kept[s]].sum() for s in self.shards])),
               "generate_s": self.gen_s,
               "hand_over": "isx_pipe_submit_planes: every batch handed over from the caller's (pageable) arrays inside the step -- reads as bit planes "
                            "(64 B a read), the reference as its 2-bit plane; the pipe's threads copy the reference planes into pinned staging and make the "
                            "32-byte wire records by XOR against them while the previous batches' DMA and kernels run",
               "record_bytes": int(st[0]["record_bytes"]) if st else None,
               "verified": "every batch checked in an untimed pass through the SAME hand-over (coverage sum == observations handed over, SNV rows ordered "
                           "and consistent with the coverage, LD counts add up; the largest batch: exact per-position coverage and per-base SNV counts "
                           "recomputed on the host); every timed batch's row counts and the checksum of its SNV + LD rows' bytes equal that pass's",
               "exact_check": getattr(self, "exact_checked", None),
               "stages_ms_per_pass": {"host_stage": tot("encode_ms"), "copy_in": tot("h2d_ms"), "kernel": k_ms, "copy_out": tot("d2h_ms"),
                                      "collect_wait": tot("collect_wait_ms"), "wall": dt_max / passes * 1e3},
               "roofline": {"bound": "hbm", "kernel": "k_pileup_dense<linkage, %s> (pipe slot output)" % ("reference-delta records" if rbytes == 32 else "segment records"),
                            "achieved": abytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0,
                            "algorithmic_bytes_per_launch": abytes / max(len(ws), 1), "bytes_per_position": abytes / max(n_pos, 1),
                            "kernel_ms_avg": k_ms / max(len(ws), 1), "kernel_ms_per_pass": k_ms, "launches": len(st),
                            "positions_per_s": n_pos / (k_ms * 1e-3) if k_ms else 0.0,
                            "traffic": _pmc("c5_dense_linkage_bytes_per_launch") if self.n_genomes == 1000 else None,
                            "traffic_source": _pmc_source("c5_dense_linkage_bytes_per_launch"),
                            # the same kernel time priced on SURVEY 8(d)'s byte model of the naive formulation (12 B per kept observation
                            # with linkage + 1 B per position of reference + 28 B per covered position out) instead of this build's format
                            # bytes -- for comparison only: `frac` above is the stricter figure
                            "survey_8d_model": {"bytes_per_pass": n_obs * 12 + n_pos + int(getattr(self, "covered", 0) or n_pos) * 28,
                                                "frac": ((n_obs * 12 + n_pos + int(getattr(self, "covered", 0) or n_pos) * 28) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms else 0.0},
                            "bound_of_the_pass": "host stager %.1f ms + pcie copy-in %.1f ms, overlapped, of %.1f ms" % (tot("encode_ms"), tot("h2d_ms"), dt_max / passes * 1e3)},
               "roofline_pcie": {"bound": "pcie", "direction": "host->device", "achieved": tot("h2d_bytes") / (dt_max / passes) / 1e9, "peak": PCIE_PEAK_GBS,
                                 "unit": "GB/s", "frac": tot("h2d_bytes") / (dt_max / passes) / 1e9 / PCIE_PEAK_GBS, "bytes_per_pass": tot("h2d_bytes"),
                                 "bytes_per_profiled_base": tot("h2d_bytes") / max(self.bases, 1.0),
                                 "device_to_host_bytes_per_pass": tot("d2h_bytes"), "device_to_host_bytes_per_position": tot("d2h_bytes") / max(n_pos, 1)}}
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        return out

    def cpu_baselines(self):
        """the CPU baselines on the same configuration: the observation stream of one batch of median size"""
        order = np.argsort([w["n_obs"] for w in self.ws])
        sel = self.ws[int(order[len(order) // 2])]["genomes"]
        wo = self.meta.generate(sel)
        cb = cpu_baseline(wo, budget_s=14.0, min_s=8.0)
        cp = cpu_baseline_python(wo, n_splits=200, budget_s=22.0, min_s=12.0)
        return cb, cp


def rank_threads_sweep(local, lut, fb, depth, scale=1.0, threads=(2, 4, 8, 16), passes=6):
    """What one rank of an 8-GPU job can do with T stager threads: shard 0 of 8 of the C5 database (the N = 8 job's
    per-rank share) through one GPU with the hand-over inside the step, for T = 2, 4, 8, 16 host threads -- the figure the first hardware
    scaling run is to be read against: on a node whose cpu quota does not grow with its GPUs a rank gets quota / 8 threads, and the
    curve is then the stager's, not the device's.  The pre-staged replay beside it is the device side (no stager)."""
    _trace("rank threads sweep")
    import torch
    from instrain_amd import engine
    out = {"workload": "C5 shard 0 of 8 (one rank's share of the N = 8 job) through one GPU, hand-over inside the step, by stager threads", "by_threads": {}}
    top = max(2, host_cpus())
    ctx = engine.Context(local, reserve_cus=C5_RESERVE_CUS)
    ctx.set_null_model(lut, fb)
    run = C5Run(ctx, 0, 8, min(top, max(threads)), depth=depth, scale=scale)
    run.verify_pass()
    out["gbp_per_pass"], out["batches"] = run.bases / 1e9, len(run.ws)
    for T in threads:
        if T > top:
            continue
        run.pipe.close()
        run.pipe = run.make_pipe(T)
        run.run(2)
        torch.cuda.synchronize()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        stats = []
        run.run(passes, stats)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / passes
        run.check_timed(stats)
        st = [x for x, _ in stats]
        out["by_threads"][str(T)] = {"ms_per_pass": dt * 1e3, "gbp_per_s": run.bases / dt / 1e9, "x8": 8 * run.bases / dt / 1e9,
                                     "cpus_busy": (time.process_time() - cpu0) / (dt * passes),
                                     "host_stage_ms_per_pass": float(np.sum([x["encode_ms"] for x in st])) / passes,
                                     "copy_in_ms_per_pass": float(np.sum([x["h2d_ms"] for x in st])) / passes}
    rep = run.staged_replay(passes=passes)
    out["staged_replay_ms_per_pass"] = rep["ms_per_pass"]
    out["note"] = "x8 = what eight such ranks add up to when every rank has its own PCIe link AND its own T threads"
    run.close()
    ctx.close()
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


def stream(pipe, variants, n_steps, depth, stats=None, keep_last=False, check=None, wires=None, keyed=False):
    """n_steps batches through the pipe, at most `depth` in flight; returns the last collected result.  check(i, result) is
    called on every collected batch (untimed verification passes only).  wires: the batches staged ahead of time
    (Pipe.stage_reads) -- a submit is then DMA + kernels only (isx_pipe_submit_wire)."""
    tickets, done, last = [], 0, None
    link = pipe.enable_linkage

    def take():
        nonlocal done, last
        r = pipe.collect(tickets[done], want_ld=link, densify=False)      # the tables as they come (views of the slot)
        if stats is not None:
            stats.append((r["stats"], dict(r["sizes"], rows_checksum=r["rows_checksum"])))
        if check is not None:
            check(done % len(variants), r)
        if keep_last and done == n_steps - 1:
            last = {"snv": r["snv"].copy()}
        pipe.release(tickets[done])
        done += 1

    for i in range(n_steps):
        if len(tickets) - done == depth:
            take()
        v = variants[i % len(variants)]
        if wires is not None:
            tickets.append(pipe.submit_wire(wires[i % len(variants)]))
        elif "planes" in v:
            tickets.append(pipe.submit_planes(v["ref_planes"], v["split_bounds"], v["planes"], keyed=keyed))
        elif pipe.read_level:
            tickets.append(pipe.submit_reads(v["ref_codes"], v["split_bounds"], v["segs"]))
        else:
            tickets.append(pipe.submit(v["ref_codes"], v["split_bounds"], v["obs"], v["pair"] if link else None))
    while done < len(tickets):
        take()
    return last


def c2_stream_leg(ctx, w, args, host_threads, steps, warmup):
    """The round-3 headline as a leg: BASELINE configs[1] (C2: one 5 Mbp genome, 20x, --skip_mm_profiling, linkage off), one
    batch per step streamed through a read-level pipe (N = 1 only).  Like the headline the leg's value is the INCLUSIVE hand-over:
    every batch goes in from the caller's bit planes (isx_pipe_submit_planes, staged inside the step); beside it the same stream as
    pre-staged pinned images (round 4's way) and with the isx_segs hand-over (round 3's: 3-bit words, byte compare)."""
    _trace("C2 stream")
    from instrain_amd import engine
    n_var = max(1, min(args.variants, steps))
    variants = make_variants(w, n_var)
    pipe = engine.Pipe(ctx, max_pos=max(v["n_pos"] for v in variants), max_obs=0, max_segs=int(w["segs"].n_seg),
                       max_splits=max(len(v["split_bounds"]) for v in variants), depth=args.depth, host_threads=host_threads,
                       pin_threads=args.pin, n_mm_bins=1, enable_linkage=False, window=args.window, stage_async=args.queued_submit, lean_output=LEAN_SLOTS)
    as_planes = [dict(v, planes=engine.PlaneBatch.from_segs(v["segs"], threads=host_threads), ref_planes=engine.RefPlanes.from_codes(v["ref_codes"], threads=host_threads))
                 for v in variants]
    if REGISTER_REF:                            # (the genomes' planes, like the headline's: registered once, copied from where they lie)
        _register_refs(as_planes)
    k0 = warmup % n_var
    rot = lambda xs: xs[k0:] + xs[:k0]
    # (1) the leg's value: planes handed over inside the step
    stream(pipe, as_planes, warmup, args.depth)
    stats = []
    t0 = time.perf_counter()
    stream(pipe, rot(as_planes), steps, args.depth, stats)
    dt = time.perf_counter() - t0
    # (2) pre-staged pinned images replayed: DMA + kernels + tables back
    wires = [pipe.stage_planes(v["ref_planes"], v["split_bounds"], v["planes"]) for v in as_planes]
    stream(pipe, variants, warmup, args.depth, wires=wires)
    st_w = []
    t1 = time.perf_counter()
    stream(pipe, rot(variants), steps, args.depth, st_w, wires=rot(wires))
    dt_w = time.perf_counter() - t1
    # (3) the isx_segs hand-over (isx_pipe_submit_reads)
    st2 = []
    t1 = time.perf_counter()
    stream(pipe, variants, steps, args.depth, st2)
    dt2 = time.perf_counter() - t1
    pipe.close()
    _unregister_refs(as_planes)
    st = [s for s, _ in stats]
    mean = lambda k, xs=None: float(np.mean([s[k] for s in (st if xs is None else xs)])) if (st if xs is None else xs) else 0.0
    k_ms = mean("kernel_ms")
    n_pos_v = int(np.mean([v["n_pos"] for v in variants]))
    h2d_b, d2h_b = mean("h2d_bytes"), mean("d2h_bytes")
    abytes = h2d_b + n_pos_v * slot_out_bytes_per_pos(w["n_obs"], n_pos_v)
    ms_step = dt / steps * 1e3
    stw = [s for s, _ in st_w]
    return {"workload": "C2 streamed: one 5 Mbp genome (0.1 Gbp of reads) per batch, 20x, 2x150 bp, --skip_mm_profiling, linkage off; every batch handed "
                        "over from the caller's bit planes inside the step (isx_pipe_submit_planes), profiled once, tables copied back; %d distinct batches" % n_var,
            "gbp_per_s": float(w["profiled_bases"]) * steps / dt / 1e9, "ms_per_step": ms_step, "steps": steps, "warmup": warmup,
            "record_bytes": int(st[0]["record_bytes"]) if st else None,
            "staged_replay": {"gbp_per_s": float(w["profiled_bases"]) * steps / dt_w / 1e9, "ms_per_step": dt_w / steps * 1e3,
                              "copy_in_ms": mean("h2d_ms", stw), "kernel_ms": mean("kernel_ms", stw)},
            "segs_hand_over": {"gbp_per_s": float(w["profiled_bases"]) * steps / dt2 / 1e9, "ms_per_step": dt2 / steps * 1e3,
                               "host_stage_ms": float(np.mean([x["encode_ms"] for x, _ in st2])) if st2 else 0.0},
            "kept_observations": int(w["n_obs"]), "read_segments": int(w["segs"].n_seg), "pipe_depth": args.depth, "host_threads": host_threads,
            "roofline_in_stream": {"bound": "hbm", "kernel": "k_pileup_dense (wire records, slot output)",
                                   "achieved": abytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0,
                                   "algorithmic_bytes_per_launch": abytes, "kernel_ms_avg": k_ms, "launches": len(st)},
            "roofline_pcie": {"bound": "pcie", "direction": "host->device", "achieved": h2d_b / (ms_step * 1e-3) / 1e9,
                              "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": h2d_b / (ms_step * 1e-3) / 1e9 / PCIE_PEAK_GBS,
                              "bytes_per_step": h2d_b, "bytes_per_profiled_base": h2d_b / float(w["profiled_bases"]), "copy_ms_avg": mean("h2d_ms"),
                              "device_to_host_bytes_per_step": d2h_b},
            "stages_ms": {"host_stage": mean("encode_ms"), "copy_in": mean("h2d_ms"), "kernel": k_ms, "copy_out": mean("d2h_ms"),
                          "collect_wait": mean("collect_wait_ms"), "step": ms_step},
            "snv_rows": int(stats[-1][1]["n_snv"]) if stats else 0}


def c2_mm_stream_leg(ctx, w, args, host_threads, steps, warmup):
    """SURVEY 8(d) C2 "second run with mm on": the reference's DEFAULT mode (mm profiling on, argumentParser.py:131, controller.py:274-281)
    streamed like c2_stream_leg -- every batch handed over inside the step (isx_pipe_submit_reads: the segments carry the pairs' mm), profiled
    once, and what shrink_basewise keeps of the (position, mm) levels (profile_utilities.py:337-350) handed back inside the step too."""
    _trace("C2 mm stream")
    from instrain_amd import engine
    M = int(w["n_mm_bins_mm"])
    n_var = max(1, min(args.variants, steps, 16))
    variants = make_variants(dict(w, segs=w["segs_mm"]), n_var)
    # the hand-over: the reads as bit planes + the pairs' mm levels (isx_read_planes.mm), the batches as 32-byte reference-delta records with
    # the level in the header (ISX_LAYOUT_MM_DELTA_RECORDS, round 6); ISX_BENCH_MM_SEGS=1: isx_segs -> 64-byte segment records (round 3's way)
    as_segs = bool(int(os.environ.get("ISX_BENCH_MM_SEGS", "0")))
    pipe = engine.Pipe(ctx, max_pos=max(v["n_pos"] for v in variants), max_obs=0, max_segs=int(w["segs_mm"].n_seg),
                       max_splits=max(len(v["split_bounds"]) for v in variants), depth=args.depth, host_threads=host_threads,
                       pin_threads=args.pin, n_mm_bins=M, enable_linkage=False, window=args.window, lean_output=LEAN_SLOTS,
                       layout=0 if as_segs else 32)
    if not as_segs:
        variants = [dict(v, planes=engine.PlaneBatch.from_segs(v["segs"], threads=host_threads), ref_planes=engine.RefPlanes.from_codes(v["ref_codes"], threads=host_threads))
                    for v in variants]
        if REGISTER_REF:
            _register_refs(variants)
    out = {"workload": "C2 streamed with mm profiling ON (%d mm bins): one 5 Mbp genome (0.1 Gbp of reads) per batch, 20x, linkage off; every batch handed over "
                       "inside the step (%s), profiled once, the per-level tables handed back inside the step; %d distinct batches"
                       % (M, "isx_pipe_submit_reads: isx_segs" if as_segs else "isx_pipe_submit_planes: bit planes + the pairs' mm levels -> reference-delta records", n_var),
           "mm_bins": M}

    pipe_levels = [False]

    def run(n, stats=None, check=None):
        tickets, done = [], 0
        fetch_s = 0.0
        n_ent = 0

        def take():
            nonlocal done, fetch_s, n_ent
            t0 = time.perf_counter()
            # the level tables as they come home (views of the slot's pinned block: level mask, coverage bytes, lists)
            r = pipe.collect(tickets[done], want_ld=False, densify=False, shrunk_entries=True)
            fetch_s += time.perf_counter() - t0
            n_ent = r["sizes"]["n_entries"]
            if stats is not None:
                stats.append((r["stats"], r["sizes"]))
            if check is not None:
                check(done % n_var, r)
            pipe_levels[0] = "levels" in r
            pipe.release(tickets[done])
            done += 1

        for i in range(n):
            if len(tickets) - done == args.depth:
                take()
            v = variants[i % n_var]
            if "planes" in v:
                tickets.append(pipe.submit_planes(v["ref_planes"], v["split_bounds"], v["planes"], keyed=False))
            else:
                tickets.append(pipe.submit_reads(v["ref_codes"], v["split_bounds"], v["segs"]))
        while done < len(tickets):
            take()
        return fetch_s, n_ent

    # untimed: every distinct batch's level tables expanded on the host -- the levels' coverages must add up to the observations handed
    # over, batch 0 must equal the one-shot batch's entries column by column
    seen = {}

    def check(k, r):
        g, mc, cl, cr = pipe.expand_levels(r) if "levels" in r else r["entries_soa"]
        cov = (mc & 0xFFFFFF).astype(np.int64)
        assert int(cov.sum()) == int(w["n_obs"]), ("mm stream: coverage does not add up", k, int(cov.sum()), int(w["n_obs"]))
        assert (np.diff(g.astype(np.int64) * 256 + (mc >> 24)) > 0).all(), ("mm stream: levels out of order", k)
        seen[k] = (len(g), int(np.isfinite(cl).sum()), int(r["sizes"]["n_snv"]))
        if k == 0:
            b = engine.Batch(ctx, variants[0]["ref_codes"], variants[0]["split_bounds"], variants[0]["segs"], None, n_mm_bins=M, enable_linkage=False)
            b.run()
            e = b.fetch()["entries"]
            b.close()
            assert g.tobytes() == e["gpos"].tobytes() and ((mc >> 24) == e["mm"]).all() and (cov == e["cnt"].sum(axis=1)).all(), "mm stream: levels != one-shot entries"
            assert cl.tobytes() == e["clon"].tobytes() and cr.tobytes() == e["clon_rarefied"].tobytes(), "mm stream: clonalities != one-shot entries"

    run(n_var, check=check)
    run(warmup)
    stats = []
    t0 = time.perf_counter()
    fetch_s, n_ent = run(steps, stats)
    dt = time.perf_counter() - t0
    for i, (_, sz) in enumerate(stats):
        assert (int(sz["n_entries"]), int(sz["n_snv"])) == (seen[i % n_var][0], seen[i % n_var][2]), ("mm stream: a timed batch differs from its verified pass", i)
    out["verified"] = "untimed pass: every batch's levels expanded on the host (coverage adds up to the observations, (position, mm) order), batch 0 == one-shot entries bit by bit; timed batches: level / SNV row counts equal"
    out["hand_back"] = "level-sparse (isx_pipe_result.lev_*)" if pipe_levels[0] else "32-byte entries -> isx_pipe_fetch_entries_shrunk"
    pipe.close()
    _unregister_refs(variants)
    st = [s for s, _ in stats]
    mean = lambda k: float(np.mean([s[k] for s in st])) if st else 0.0
    ms_step = dt / steps * 1e3
    out.update({"gbp_per_s": float(w["profiled_bases"]) * steps / dt / 1e9, "ms_per_step": ms_step, "steps": steps, "warmup": warmup,
                "entries": int(n_ent), "record_bytes": int(st[0]["record_bytes"]) if st else None,
                "h2d_bytes_per_step": mean("h2d_bytes"), "d2h_bytes_per_step": mean("d2h_bytes"),
                "stages_ms": {"host_stage": mean("encode_ms"), "copy_in": mean("h2d_ms"), "kernel": mean("kernel_ms"), "copy_out": mean("d2h_ms"),
                              "collect_wait": mean("collect_wait_ms"), "collect_and_fetch": fetch_s / steps * 1e3, "step": ms_step}})
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) under
    torch.distributed.run on this node and hand their output through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, ISX_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _short(s, n=118):
    return s if len(s) <= n else s[:n - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed steps; a step = one pass over the rank's share of the C5 database")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the C5 database / the C2 genome (debug only; reported in config)")
    ap.add_argument("--variants", type=int, default=32, help="distinct batches cycled through the C2 leg")
    ap.add_argument("--depth", type=int, default=8, help="pipe slots (batches in flight: copy-in, pass, copy-out and up to four finishing)")
    ap.add_argument("--host-threads", type=int, default=0, help="staging threads of the pipe (0 = the cpus this rank may use)")
    ap.add_argument("--queued-submit", action="store_true", help="submit_reads only queues the batch, the pipe's stager thread encodes it (isx_pipe_params.stage_async = 1; same-box A/B in profiles/r03_stream_ab.md: no gain on a 16-cpu cgroup)")
    ap.add_argument("--pin", action="store_true", help="bind the staging threads to the L3 domains of the GPU's NUMA node")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the process to the GPU's NUMA node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-linkage-leg", action="store_true")
    ap.add_argument("--no-mm-leg", action="store_true")
    ap.add_argument("--no-resident-leg", action="store_true")
    ap.add_argument("--no-c2-leg", action="store_true")
    ap.add_argument("--no-rank-sweep", action="store_true", help="skip the threads-per-rank sweep (one shard of 8 with 2 / 4 / 8 / 16 stager threads)")
    ap.add_argument("--no-bam-leg", action="store_true", help="skip the BAM end-to-end legs (profile_bam, c5_bam, bam_sharded)")
    ap.add_argument("--only-mm", action="store_true", help="the mm-on legs alone (debug)")
    ap.add_argument("--only-c5", action="store_true", help="the headline alone: no C2 legs, no BAM legs, no CPU baselines (debug)")
    ap.add_argument("--detail", default=os.path.join(REPO, "bench_detail.json"), help="where the full record goes (the printed line is a digest)")
    ap.add_argument("--window", type=int, default=0)
    args = ap.parse_args()
    if args.only_c5:
        args.no_c2_leg = args.no_resident_leg = args.no_mm_leg = args.no_linkage_leg = args.no_cpu_baseline = args.no_bam_leg = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import instrain_amd           # (sets GPU_MAX_HW_QUEUES before the HIP runtime comes up, see instrain_amd/__init__.py)
    import torch
    import torch.distributed as dist
    from instrain_amd import dist as idist
    from instrain_amd import engine
    from tests import util        # only for the committed null-model LUT fixture (data, not oracle code)

    # More ranks than visible GPUs (a 1-GPU box exercising the N > 1 control flow): gloo, ranks share the devices round robin.
    want_world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("ISX_DIST_BACKEND") or ("gloo" if want_world > max(n_dev, 1) else None)
    rank, local, world = idist.init_from_env(backend=backend)
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    shared = n_dev > 0 and world > n_dev
    if "ISX_DEVICE" in os.environ:
        local = int(os.environ["ISX_DEVICE"])
    elif shared:
        local = local % n_dev
    use_nccl = world > 1 and dist.get_backend() == "nccl"
    if os.environ.get("ISX_BENCH_SCHEDULE"):    # tuning aid: how the runtime's host threads wait (1 spin, 2 yield, 4 blocking sync)
        import ctypes
        ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(int(os.environ["ISX_BENCH_SCHEDULE"]))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if (world == 1 or use_nccl) else torch.device("cpu")
    numa_node = None if args.no_bind else bind_to_gpu_numa_node(torch, local)
    ctx = engine.Context(local)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    host_threads = args.host_threads or max(2, min(48, host_cpus() // world))

    def barrier():
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()

    # ---- the headline: configs[4] (C5), the configuration north_star quotes its target on ----
    _trace("C5 generate")
    # (its own context: 4 CUs of every XCD kept free of pileup kernels for the finishers' linkage chains -- isx_ctx_reserve_cus;
    # the resident-batch legs below keep the whole device)
    ctx5 = engine.Context(local, reserve_cus=C5_RESERVE_CUS)
    ctx5.set_null_model(lut, fb)
    c5 = C5Run(ctx5, rank, world, host_threads, depth=args.depth, scale=args.scale, stage_async=args.queued_submit)
    c5_ref_registered = bool(c5.ref_registered)
    _trace("C5 verify pass")
    c5.verify_pass()                            # untimed: every batch's tables checked on the host; also warms every slot
    # N > 1: the job is the same whole database (strong scaling), so a rank's pass shrinks to a few ms -- a STEP is then PPS passes
    # back to back (the rate is what is reported; the timed region must be long against a barrier's jitter)
    PPS = 1 if world == 1 else int(os.environ.get("ISX_BENCH_PASSES_PER_STEP", 2 * world))
    if args.warmup:
        c5.run(args.warmup * PPS)
    barrier()
    torch.cuda.synchronize()
    stats = []
    cpu0, thr0 = time.process_time(), cgroup_throttled_ms()
    t0 = time.perf_counter()
    c5.run(args.steps * PPS, stats)
    torch.cuda.synchronize()
    dt_mine = time.perf_counter() - t0
    # what the timed passes cost the HOST: cpu time of all threads of this process per wall second (stager threads, finishers, the DMA's
    # submitters, waits that spin) beside the cpus the cgroup grants, and how long the cgroup held the process back meanwhile
    host_cpu = {"cpus_busy": (time.process_time() - cpu0) / dt_mine, "cpus_granted": cgroup_cpus(),
                "throttled_ms_per_pass": None if thr0 is None else (cgroup_throttled_ms() - thr0) / (args.steps * PPS)}
    barrier()
    dt = time.perf_counter() - t0
    c5.check_timed(stats)
    bases_all = c5.bases
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        u = torch.tensor([c5.bases], dtype=torch.float64, device=dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        bases_all = float(u.item())
        mine = torch.tensor([dt_mine, c5.bases, float(len(c5.ws))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        pass_ms = allr[:, 0] / (args.steps * PPS) * 1e3
        per_rank = {"pass_ms": [round(float(x), 3) for x in pass_ms], "pass_ms_min": float(pass_ms.min()), "pass_ms_max": float(pass_ms.max()),
                    "gbp_per_pass": [round(float(x) / 1e9, 4) for x in allr[:, 1]], "batches": [int(x) for x in allr[:, 2]],
                    "time_imbalance": float(pass_ms.max() / pass_ms.mean()), "passes_per_step": PPS}

    # the one collective of the path: the final gather of ONE pass's whole profile to rank 0 -- every batch's SNV rows and LD rows and
    # the per-scaffold summaries -- outside the timed steps, timed on its own
    gather_ms = gathered = None
    if world > 1:
        mine_tables = c5.gather_pass()
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        got = idist.gather_tables(mine_tables, dst=0, device=dev)
        torch.cuda.synchronize()
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            gathered = {"rows": {k: int(len(v)) for k, v in got.items()}, "bytes": int(sum(v.nbytes for v in got.values())),
                        "my_bytes": int(sum(v.nbytes for v in mine_tables.values()))}
        del mine_tables, got
    head = c5.report(dt, bases_all, stats, args.steps * PPS, gather_ms)
    if per_rank is not None:
        head["per_rank"] = per_rank
        head["final_gather"] = gathered
    if (world == 1 and rank == 0 and (not args.only_c5 or os.environ.get("ISX_BENCH_STAGED"))) or world > 1:
        # (N > 1: every rank replays its own staged images -- the device side of the scaling curve, free of the host stager's share of
        # the node's cpus; aggregated below)
        rep = c5.staged_replay(passes=4 * PPS)
        if world > 1:
            barrier()
            u = torch.tensor([rep["ms_per_pass"]], dtype=torch.float64, device=dev)
            dist.all_reduce(u, op=dist.ReduceOp.MAX)
            rep = dict(rep, ms_per_pass=float(u.item()), gbp_per_s=bases_all / (float(u.item()) * 1e-3) / 1e9, note="max over ranks of a rank's own replay time")
        head["staged_replay"] = rep
    if rank == 0 and world == 1 and (not args.only_c5 or os.environ.get("ISX_BENCH_RESIDENT_REF")):
        head["resident_reference"] = c5.resident_reference_passes()         # (last: the wires keep their device copies from here on)
    cb = cp = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, cp = c5.cpu_baselines()
    n_batches = len(c5.ws)
    c5.close()
    del c5
    ctx5.close()

    # one BAM sharded over the ranks (every rank takes part; rank 0 reports)
    sharded = None
    if not args.no_bam_leg:
        try:
            sharded = bam_sharded_leg(ctx, rank, world, local, host_threads, barrier, dev)
        except Exception as e:                      # never lose the line over an extra leg
            sharded = {"error": repr(e)}
            if world > 1:
                raise

    if rank == 0:
        detail = {"c5": head}
        out = {
            "metric": "Gbp profiled/s", "value": head["gbp_per_s"], "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "passes_per_step": PPS,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "world_size_seen": world, "backend": (dist.get_backend() if world > 1 else None),
            "config": {"workload": _short("C5: 1000-genome database, 10 Gbp reads, --database_mode, pileup+SNV call+linkage; step = whole pass%s"
                                          % ("" if args.scale == 1.0 else " [DEBUG scale %g]" % args.scale)),
                       "genomes_kept": head["genomes_kept"], "positions": head["positions"], "read_gbp_per_step": bases_all / 1e9,
                       "batches_per_step": n_batches if world == 1 else None, "hand_over": _short("isx_pipe_submit_planes per batch INSIDE the step: caller's bit planes (pageable) -> XOR stager -> %d-byte wire records in pinned staging -> hipMemcpyAsync -> kernels -> tables back; reference planes %s" % (head.get("record_bytes") or 0, "copied from the caller's registered arrays (isx_host_register, once, untimed) with every batch" if c5_ref_registered else "through staging"), 330),
                       "reference_registered": c5_ref_registered,
                       "pipe_depth": args.depth, "pileup_cus": 256 - 8 * C5_RESERVE_CUS, "lean_slots": LEAN_SLOTS, "host_threads_per_rank": host_threads, "cgroup_cpus": cgroup_cpus(),
                       "numa_node": numa_node, "parallelism": "genome-sharded x%d%s" % (world, " (ranks share %d GPU)" % n_dev if shared else ""),
                       "verified": "per-batch checks in an untimed pass (largest batch: exact coverage + per-base SNV counts from the host); timed batches: row counts + checksum of the SNV / LD bytes equal"},
            "roofline": {k: head["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                            "algorithmic_bytes_per_launch", "kernel_ms_avg", "launches", "positions_per_s", "survey_8d_model", "bound_of_the_pass")},
            "h2d_bytes_per_base": head["roofline_pcie"]["bytes_per_profiled_base"], "pcie_frac": head["roofline_pcie"]["frac"],
            "snv_pairs_linked_per_s": head["snv_pairs_linked_per_s"],
            "stages_ms": {k: round(v, 2) for k, v in head["stages_ms_per_pass"].items()},
            "host_cpu": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in host_cpu.items()},
        }
        if "staged_replay" in head:                 # round 4's headline (pre-staged pinned images replayed): an extra now
            out["c5_staged_replay_gbp_per_s"] = head["staged_replay"]["gbp_per_s"]
            out["c5_stage_once_s"] = round(head["staged_replay"]["stage_once_s"], 3)
        if "resident_reference" in head:
            out["c5_resident_reference_gbp_per_s"] = head["resident_reference"]["gbp_per_s"]
            out["c5_resident_reference_h2d_bytes_per_base"] = head["resident_reference"]["h2d_bytes_per_base"]
        out["roofline"]["kernel"] = _short(out["roofline"]["kernel"], 80)
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
            out["final_gather"] = gathered
        if per_rank is not None:
            out["per_rank"] = per_rank
        if cb is not None:
            detail["cpu_baseline"], detail["cpu_baseline_python"] = cb, cp
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": _short("median C5 batch: " + cb["sample"])}
            out["cpu_baseline_python"] = {"value": cp["value"], "unit": cp["unit"], "cores": cp["cores"], "kind": cp["kind"],
                                          "sample": _short("median C5 batch: " + cp["sample"])}
            out["speedup_vs_cpu_port"] = out["value"] / cb["value"] if cb["value"] else None
            out["speedup_vs_python_restatement"] = out["value"] / cp["value"] if cp["value"] else None
        if sharded is not None:
            detail["bam_sharded"] = sharded
            out["bam_sharded_gbp_per_s"] = sharded.get("gbp_per_s")
        legs = {}
        if world == 1 and not (args.no_c2_leg and args.no_resident_leg and args.no_mm_leg and args.no_linkage_leg):
            want_mm = not args.no_mm_leg
            w = c2_workload(seed=2, scale=args.scale, with_mm=want_mm)
            if not args.no_c2_leg:
                legs["c2_stream"] = c2_stream_leg(ctx, w, args, host_threads, 32, 4)
                out["c2_stream_gbp_per_s"] = legs["c2_stream"]["gbp_per_s"]
                out["c2_staged_replay_gbp_per_s"] = legs["c2_stream"]["staged_replay"]["gbp_per_s"]
            if not args.no_resident_leg:
                res = resident_leg(ctx, w, args.window)
                legs["resident"] = res
                # The dominant kernel with the GPU to itself: 10 blocking runs over a resident read-level C2 batch with its full
                # count table (20 B/pos out), the dispatch's own time stamps = what rocprofv3 reports.
                k_alone = res["reads"]["kernel_ms_alone"]
                tr = res["reads"]["timings"]
                n_seg = int(w["segs"].n_seg)
                n_rec = (n_seg + n_seg // 4096 * 32 + 31) // 32 * 32 if tr["record_bytes"] == 32 else (n_seg + 15) // 16 * 16
                ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, dense=True, record_bytes=tr["record_bytes"], n_rec=n_rec, out_bytes_per_pos=20)
                legs["roofline_c2_resident"] = _roofline("k_pileup_dense (%d-byte wire records, resident C2 batch with count table)" % tr["record_bytes"], ab, k_alone, tr,
                                                         _pmc("c2_reads_bytes_per_launch") if args.scale == 1.0 else None,
                                                         kernel_ms_min=res["reads"]["kernel_ms_min"], traffic_source=_pmc_source("c2_reads_bytes_per_launch"))
                legs["roofline_lds"] = {"bound": "lds-atomic", "achieved": res["reads"]["timings"].get("lds_atomics", w["n_obs"]) / (k_alone * 1e-3) / 1e12,
                                        "peak": LDS_ATOMIC_PEAK / 1e12, "unit": "T atomics/s",
                                        "note": "peak = 256 CUs x 16 lanes/clk x 2.4 GHz (conflict-free)"}
                ko = res["observations"]["kernel_ms_alone"]
                to = res["observations"]["timings"]
                abo = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, dense=True, record_bytes=to["record_bytes"])
                legs["roofline_observation_kernel"] = _roofline("k_pileup_dense (2-byte observation records; isx_pipe_submit / isx_batch_create)", abo, ko, to,
                                                                _pmc("c2_pileup_bytes_per_launch"), kernel_ms_min=res["observations"]["kernel_ms_min"])
                # the yardstick that describes this kernel ("positions per second and VALU issue do"): positions per second, and the
                # share of all SIMD issue slots in which a wave issued a VALU instruction (SQ_ACTIVE_INST_VALU of the committed PMC pass, in
                # quad-cycles, over SIMDs x launch time x 2.4 GHz / 4)
                sq = _pmc("c2_delta_sq") if args.scale == 1.0 else None
                legs["roofline_c2_resident"]["positions_per_s"] = w["n_pos"] / (k_alone * 1e-3)
                if sq and k_alone:
                    simd_quads = 256 * 4 * (k_alone * 1e-3) * 2.4e9 / 4.0
                    legs["roofline_c2_resident"]["valu_issue"] = {"bound": "valu-issue", "achieved": sq["SQ_ACTIVE_INST_VALU"], "peak": simd_quads, "unit": "SIMD quad-cycles per launch",
                                                                  "frac": sq["SQ_ACTIVE_INST_VALU"] / simd_quads, "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"),
                                                                  "valu_insts_per_position": (sq.get("SQ_INSTS_VALU") or 0) * 64.0 / w["n_pos"]}
                out["roofline_c2_resident"] = {k: legs["roofline_c2_resident"][k] for k in ("achieved", "frac", "traffic", "kernel_ms_avg", "algorithmic_bytes_per_launch", "positions_per_s")}
                if "valu_issue" in legs["roofline_c2_resident"]:
                    out["roofline_c2_resident"]["valu_issue_frac"] = legs["roofline_c2_resident"]["valu_issue"]["frac"]
            if want_mm:
                legs["mm_on"] = mm_leg(ctx, w)
                out["mm_on_roofline_frac"] = legs["mm_on"]["roofline"]["frac"]
                legs["c2_mm_stream"] = c2_mm_stream_leg(ctx, w, args, host_threads, 32, 4)
                out["c2_mm_stream_gbp_per_s"] = legs["c2_mm_stream"]["gbp_per_s"]
            if not args.no_linkage_leg:
                legs["linkage"] = linkage_leg(ctx)
                out["c3_snv_pairs_linked_per_s"] = legs["linkage"]["snv_pairs_linked_per_s"]
                if "reads_sorted_chain" in legs["linkage"]:
                    out["c3_snv_pairs_linked_per_s_sorted_chain"] = legs["linkage"]["reads_sorted_chain"]["snv_pairs_linked_per_s"]
            if not args.no_cpu_baseline:
                legs["cpu_baseline_c2"] = cpu_baseline(w)
                legs["cpu_baseline_python_c2"] = cpu_baseline_python(w)
            del w
        if world == 1 and not args.only_c5 and not args.no_rank_sweep:
            try:
                legs["rank_threads_sweep"] = rank_threads_sweep(local, lut, fb, args.depth, args.scale)
                out["rank_of_8_gbp_per_s_by_threads"] = {k: round(v["gbp_per_s"], 1) for k, v in legs["rank_threads_sweep"]["by_threads"].items()}
            except Exception as e:                  # never lose the line over an extra leg
                legs["rank_threads_sweep"] = {"error": repr(e)}
        if world == 1 and not args.no_bam_leg:
            for key, fn in (("profile_bam", lambda: profile_bam_leg(ctx, host_threads)), ("c5_bam", lambda: c5_bam_leg(ctx, host_threads, args.scale))):
                try:
                    legs[key] = fn()
                    out[key + "_gbp_per_s"] = legs[key].get("gbp_per_s")
                except Exception as e:                  # never lose the line over an extra leg
                    legs[key] = {"error": repr(e)}
        detail.update(legs)
        detail["line"] = dict(out)
        try:
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
            out["detail_file"] = os.path.relpath(args.detail, REPO)
        except OSError as e:
            out["detail_file"] = "not written: %r" % (e,)
        if os.environ.get("ISX_BENCH_DETAIL_STDERR"):
            print(json.dumps(detail), file=sys.stderr, flush=True)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
