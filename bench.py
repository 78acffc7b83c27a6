#!/usr/bin/env python3
"""bench.py -- headline measurement of the `inStrain profile` hot path on MI355X.

A "step" = one batch of synthetic observations profiled ONCE, the way production does it: the batch is
handed over from host memory (isx_pipe_submit: 8-byte isx_obs encoded to 2-byte records into pinned
staging by the pipe's host threads, hipMemcpyAsync in), profiled (k_pileup_dense) and its tables copied
back to pinned host memory (isx_pipe_collect).  The K timed steps stream K batches through the pipe with
copy-in / pass / copy-out of consecutive batches overlapping; the batches are distinct (up to 32 variants
of the workload, see instrain_amd.synth.shifted_variant; cycled beyond that).  At N=1 the workload is
BASELINE.json configs[1] (C2: one 5 Mbp genome, 20x, 2x150 bp, --skip_mm_profiling, linkage off).  For
N>1 every rank streams its own C2 genomes (scaffolds shard embarrassingly; weak scaling; no data-path
collective; one final RCCL gather of the SNV tables after the timed region, reported separately).

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects `roofline`
(dominant kernel k_pileup_dense vs the HBM roof, kernel durations from the timed region), `roofline_pcie`
(the hand-over vs the PCIe Gen5 x16 link), `resident` (the same kernel re-run over a resident batch: the
kernel-only ceiling, not what production does) and `cpu_baseline` (the oracle's C restatement of the
reference loop on the host cores, same workload; rank 0, N=1 only).
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


def c2_workload(seed, scale=1.0, with_mm=False):
    """C2 of SURVEY 8(d). One generation serves both runs: `obs` has mm = 0 (the headline
    --skip_mm_profiling run), `obs_mm` (with_mm) carries the pairs' mismatch counts (mm profiling on)."""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=int(5_000_000 * scale), coverage=20, n_sites=int(5000 * scale),
                            seed=seed, skip_mm=not with_mm)
    if with_mm:
        w["obs_mm"] = w["obs"]
        w["n_mm_bins_mm"] = w["n_mm_bins"]
        o = w["obs"].copy()
        o["mm"] = 0
        w["obs"] = o
        w["n_mm_bins"] = 1
    return w


def pileup_algorithmic_bytes(n_obs, n_pos, n_entries, dense, record_bytes=8):
    """Bytes one launch has to move: `record_bytes` per observation in -- 4 for the
    compact resident stream the library builds at upload, 8 for isx_obs as is (SURVEY 8(d)'s figure) --,
    1 B/pos reference in, and out dense (M==1): 16 B counts + 4 B clonality per position; mm path: 32 B
    per present (pos, mm) entry."""
    b = n_obs * record_bytes + n_pos * 1
    b += n_pos * (16 + 4) if dense else n_entries * 32
    return b


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


def cpu_baseline(w, budget_s=25.0, min_s=10.0):
    """The oracle's C port of the reference per-column loop (oracle/oracle_core.c), split by split exactly
    like profile_split, on the splits of the SAME workload (wrapped around) for >= min_s of wall time:
    first on one core, then on T host threads (the C call releases the GIL; the reference parallelises
    over splits the same way, profile_controller.py:243-271).  `value` / `cores` = the T-thread run."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from tests import util
    lut, fb = util.load_lut()
    obs, pair, bounds = w["obs"], w["pair"], w["split_bounds"]
    gpos = obs["gpos"].astype(np.int64)
    lo, hi = split_obs_ranges(gpos, bounds)
    letters = np.array(list("ACTGN"))
    n_splits = len(bounds) - 1
    # per-split inputs prepared once, outside the timing (the reference's workers get theirs from the BAM)
    jobs = []
    for j in range(n_splits):
        s, e = int(bounds[j]), int(bounds[j + 1])
        sl = slice(int(lo[j]), int(hi[j]))
        jobs.append((np.ascontiguousarray(gpos[sl], dtype=np.int32), np.ascontiguousarray(obs["base"][sl]),
                     np.ascontiguousarray(obs["mm"][sl], dtype=np.int32), np.ascontiguousarray(pair[sl], dtype=np.int32),
                     "".join(letters[w["ref_codes"][s:e]]), s, e - s, int(((gpos[sl] >= s) & (gpos[sl] < e)).sum())))

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


INT8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA peak (spec, no sparsity; MI355X_MICROARCH.md: >= 4404 measured)


def linkage_leg(ctx, seed=3):
    """Secondary metric: SNV pairs linked / s on BASELINE configs[2] (C3: 5 Mbp, 200x, 50 000 SNV sites), with
    the sparse pair-increment path (default) and the dense int8-MFMA path (linkage_mode 2)."""
    from instrain_amd import engine, synth
    glen = int(os.environ.get("ISX_BENCH_C3_BP", 5_000_000))        # configs[2] in full; smaller = a slice of it (debug)
    # the multi-threaded generator (same read / site model as synth.make_workload, seconds instead of minutes at 200x)
    meta = synth.Metagenome(1, total_read_bp=200.0 * glen, seed=seed, contigs=1, len_lo=glen, len_hi=glen, abundance_sigma=0.0,
                            min_genome_coverage=0.0, site_frac=0.01, af_lo=0.2, af_hi=0.5)
    w = meta.generate([0])
    out = {"workload": "C3%s: %.1f Mbp genome, 200x, %d SNV sites (1 / 100 bp, two haplotype backgrounds), skip_mm, linkage on"
                       % ("" if glen == 5_000_000 else " slice", glen / 1e6, glen // 100),
           "kept_observations": int(w["n_obs"]), "read_pairs": int(w["n_pairs"])}
    for mode, name in ((1, "sparse"), (2, "dense_mfma")):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=True,
                         linkage_mode=mode)
        for _ in range(2):
            b.run()
        ts, mf = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            b.run()
            ts.append(time.perf_counter() - t0)
            mf.append(b.timings()["mfma_ms"])
        s, t = b.sizes(), b.timings()
        b.close()
        dt = float(np.median(ts))
        r = {"snv_pairs_linked_per_s": s["n_edges"] / dt, "edges": s["n_edges"], "ld_rows": s["n_ld"],
             "pair_increments": s["n_increments"], "allele_observations": s["n_allele_obs"], "ms_per_step": dt * 1e3,
             "kernel_ms": {k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
             "gbp_per_s": w["profiled_bases"] / 1e9 / dt}
        if mode == 2:
            ms = float(np.median(mf))
            tops = 2.0 * t["dense_macs"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            r["mfma"] = {"kernel": "k_dense_gemm (v_mfma_i32_32x32x32_i8)", "tiles": t["dense_tiles"],
                         "int8_macs_per_pass": t["dense_macs"], "xt_bytes": t["dense_bytes"], "pass_ms": ms,
                         "achieved_tops": tops, "peak_tops": INT8_MFMA_PEAK_TOPS, "utilisation": tops / INT8_MFMA_PEAK_TOPS}
        out[name] = r
    out["snv_pairs_linked_per_s"] = out["sparse"]["snv_pairs_linked_per_s"]
    return out


def mm_leg(ctx, w, steps=10):
    """C2 again with mm profiling ON (the reference's default): k_pileup_mm, sparse (pos, mm) entries."""
    from instrain_amd import engine
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs_mm"], None, n_mm_bins=w["n_mm_bins_mm"],
                     enable_linkage=False)
    for _ in range(3):
        b.run()
    ts, ks = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        b.run()
        ts.append(time.perf_counter() - t0)
        ks.append(b.timings()["pileup_ms"])
    s, t = b.sizes(), b.timings()
    b.close()
    dt, k = float(np.median(ts)), float(np.mean(ks))
    ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], s["n_entries"], dense=False, record_bytes=t["record_bytes"])
    traffic = None
    try:
        traffic = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get("c2_mm_pileup_bytes_per_launch")
    except Exception:
        pass
    return {"workload": "C2 with mm profiling on (%d mm bins)" % w["n_mm_bins_mm"], "gbp_per_s": w["profiled_bases"] / 1e9 / dt,
            "ms_per_step": dt * 1e3, "entries": s["n_entries"], "snv_rows": s["n_snv"],
            "roofline": {"bound": "hbm", "kernel": "k_pileup_mm", "achieved": ab / (k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ab / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": ab,
                         "record_bytes": t["record_bytes"], "kernel_ms_avg": k, "blocks": t["pileup_blocks"], "threads": t["pileup_threads"],
                         "lds_bytes": t["pileup_lds_bytes"], "window": t["pileup_window"]}}


def resident_leg(ctx, w, window, steps=30):
    """The kernel-only ceiling: two resident copies of the batch passed over alternately (the next pass is queued
    before the current one is collected).  Nothing is handed over or fetched -- NOT what production does; kept
    because it isolates the pileup kernel (10 blocking runs give the kernel alone, as rocprofv3 sees it)."""
    from instrain_amd import engine
    ring = [engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False, window=window)
            for _ in range(2)]
    for i in range(4):
        ring[i % 2].run()
    k_alone = 0.0
    t0 = time.perf_counter()
    for _ in range(10):
        ring[0].run()
        k_alone += ring[0].pileup_ms()
    sync_ms = (time.perf_counter() - t0) * 1e3 / 10
    k_alone /= 10
    t0 = time.perf_counter()
    ring[0].launch()
    for i in range(steps):
        if i + 1 < steps:
            ring[(i + 1) % 2].launch()
        ring[i % 2].wait()
    dt = time.perf_counter() - t0
    tim = ring[0].timings()
    for b in ring:
        b.close()
    return {"gbp_per_s": w["profiled_bases"] * steps / dt / 1e9, "ms_per_step": dt / steps * 1e3, "sync_ms_per_step": sync_ms,
            "kernel_ms_alone": k_alone, "steps": steps, "window": tim["pileup_window"], "blocks": tim["pileup_blocks"],
            "threads": tim["pileup_threads"], "lds_bytes": tim["pileup_lds_bytes"],
            "note": "resident re-run of one batch (no hand-over, no fetch): kernel-only ceiling"}


def c5_leg(ctx, rank, world, host_threads, barrier, dist_info, depth=4, with_cpu=True, scale=1.0):
    """BASELINE.json configs[4]: 1000-genome database, 10 Gbp of reads, --database_mode (one mm
    bin; genomes below 1x dropped like fasta.py:110-136 does).  The kept genomes are LPT-sharded 8 ways on the
    reference's own cost estimate (read pairs, profile_controller.py:460-465); rank r streams shard r through its
    pipe in batches (one batch = a few genomes), every batch handed over, profiled once, tables copied back.
    N=1: the per-GPU shard (1/8 of C5); N=8: the whole configuration."""
    from instrain_amd import dist as idist
    from instrain_amd import engine, synth
    n_genomes = max(16, int(round(1000 * scale)))            # scale < 1: debug runs only (reported in the workload string)
    meta = synth.Metagenome(n_genomes, total_read_bp=10e9 * n_genomes / 1000.0, seed=5, threads=max(2, host_threads))
    kept = meta.kept_genomes()
    shards = idist.lpt_shards(meta.pairs[kept], 8)
    mine = kept[shards[rank % 8]]
    est = (meta.pairs[mine] * 2 * meta.read_len * 0.92).astype(np.int64)
    batches = idist.pack_batches(meta.length[mine], est, 40_000_000, 150_000_000)
    t0 = time.perf_counter()
    ws = [meta.generate(mine[b]) for b in batches]
    gen_s = time.perf_counter() - t0
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=depth, host_threads=host_threads,
                       pin_threads=False, n_mm_bins=1, enable_linkage=True, min_snp=20, jump_slack=0.5)
    stream(pipe, ws, len(ws), depth)                         # warm-up: one pass over the shard (every slot's tables and linkage buffers reach their steady size; the pipe learns the stream's jump slack)
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
    abytes = pileup_algorithmic_bytes(n_obs, n_pos, 0, dense=True, record_bytes=2)
    k_ms = tot("kernel_ms")
    out = {"workload": "C5 shard %d of 8 per GPU: %d of the %d kept genomes (of the database; %.2f Gbp of positions, %.2f Gbp of reads on this rank), "
                       "--database_mode, pileup + SNV call + linkage, streamed in %d batches%s" % (rank % 8, len(mine), len(kept), n_pos / 1e9, bases / 1e9, len(ws),
                                                                     "" if n_genomes == 1000 else " [DEBUG SCALE: %d genomes]" % n_genomes),
           "gbp_per_s": bases_all / dt_max / 1e9, "seconds": dt_max, "n_gpus": world,
           "genomes_kept": int(len(kept)), "genomes_total": n_genomes, "positions": n_pos, "kept_observations": n_obs,
           "mean_depth": n_obs / max(n_pos, 1), "snv_rows": int(sum(z["n_snv"] for _, z in stats)),
           "linkage": "on (sparse path; the reference links every profile, linkage.py:14-44)",
           "snv_pairs_linked": int(sum(z["n_edges"] for _, z in stats)), "ld_rows": int(sum(z["n_ld"] for _, z in stats)),
           "snv_pairs_linked_per_s": float(sum(z["n_edges"] for _, z in stats)) * world / dt_max,
           "load_imbalance": float(max(meta.pairs[kept[s]].sum() for s in shards) / np.mean([meta.pairs[kept[s]].sum() for s in shards])),
           "generate_s": gen_s,
           "stages_ms_total": {"host_encode": tot("encode_ms"), "copy_in": tot("h2d_ms"), "kernel": k_ms, "copy_out": tot("d2h_ms"),
                               "wall": dt * 1e3},
           "roofline": {"bound": "hbm", "kernel": "k_pileup_dense", "achieved": abytes / (k_ms * 1e-3) / 1e9 if k_ms else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0,
                        "algorithmic_bytes": abytes, "kernel_ms_total": k_ms, "launches": len(st), "traffic": None},
           "roofline_pcie": {"bound": "pcie", "direction": "device->host", "achieved": tot("d2h_bytes") / dt / 1e9, "peak": PCIE_PEAK_GBS,
                             "unit": "GB/s", "frac": tot("d2h_bytes") / dt / 1e9 / PCIE_PEAK_GBS, "bytes": tot("d2h_bytes"),
                             "host_to_device_bytes": tot("h2d_bytes")}}
    if gather_ms is not None:
        out["final_gather_ms"] = gather_ms
    if with_cpu:
        big = max(ws, key=lambda w: w["n_obs"])
        cb = cpu_baseline(big, budget_s=14.0, min_s=7.0)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_port"] = out["gbp_per_s"] / cb["value"] if cb["value"] else None
    return out



def make_variants(w, n):
    """n distinct batches of w's shape (synth.shifted_variant), built on a few threads"""
    from concurrent.futures import ThreadPoolExecutor
    from instrain_amd import synth
    with ThreadPoolExecutor(max_workers=max(1, min(8, host_cpus()))) as ex:
        return list(ex.map(lambda k: synth.shifted_variant(w, k), range(n)))


def profile_bam_leg(ctx):
    """The whole seam on a BAM file: instrain_amd.profile.profile_bam (front end scan + filter + expansion fused into the
    pipe's staging, device batch, table hand-back, SplitObjects) on a synthetic sorted BAM of 0.12 Gbp of reads (2 x 150 bp,
    insert N(350,30)) over one 3 Mbp scaffold, mm profiling on (the reference's default).  Best of three runs; the BAM is
    written once into /tmp by a slow Python writer that is not timed."""
    import instrain_amd.profile as amd
    from tests import util
    from tools.bench_front import write_simple_bam
    n_pairs, G = 400_000, 3_000_000
    path = "/tmp/isx_bench_%d_%d.bam" % (n_pairs, G)
    if not os.path.exists(path):
        write_simple_bam(path, G, n_pairs)
    rng = np.random.Generator(np.random.PCG64(1))                # the reference sequence write_simple_bam draws first
    seq = "".join(np.array(list("ACTG"))[rng.integers(0, 4, G, dtype=np.uint8)])
    lut, fb = util.load_lut()
    nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
    nm[-1] = fb
    best, n_splits = None, 0
    for rep in range(3):
        t0 = time.perf_counter()
        out = amd.profile_bam(path, None, None, None, s2s={"scaf": seq}, null_model=nm, ctx=ctx)
        dt = time.perf_counter() - t0
        n_splits = len(out)
        best = dt if best is None else min(best, dt)
    return {"workload": "profile_bam end to end: sorted BAM on disk (%.2f Gbp of reads, %d read pairs, one %.1f Mbp scaffold) -> SplitObjects, "
                        "mm profiling + linkage on" % (n_pairs * 300 / 1e9, n_pairs, G / 1e6),
            "seconds": best, "gbp_per_s": n_pairs * 300 / 1e9 / best, "split_objects": n_splits,
            "note": "host-bound (BGZF inflate, read filter, per-base expansion on the box's CPUs); a 0.9 Gbp BAM takes 0.8-1.0 s (DESIGN.md section 5)"}


def stream(pipe, variants, n_steps, depth, stats=None, keep_last=False):
    """n_steps batches through the pipe, at most `depth` in flight; returns the last collected result"""
    tickets, done, last = [], 0, None
    link = pipe.enable_linkage
    for i in range(n_steps):
        if len(tickets) - done == depth:
            r = pipe.collect(tickets[done], want_ld=link)
            if stats is not None:
                stats.append((r["stats"], r["sizes"]))
            if keep_last and done == n_steps - 1:
                last = {"snv": r["snv"].copy()}
            pipe.release(tickets[done])
            done += 1
        v = variants[i % len(variants)]
        tickets.append(pipe.submit(v["ref_codes"], v["split_bounds"], v["obs"], v["pair"] if link else None))
    while done < len(tickets):
        r = pipe.collect(tickets[done], want_ld=link)
        if stats is not None:
            stats.append((r["stats"], r["sizes"]))
        if keep_last and done == n_steps - 1:
            last = {"snv": r["snv"].copy()}
        pipe.release(tickets[done])
        done += 1
    return last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the C2 genome (debug only; reported in config)")
    ap.add_argument("--variants", type=int, default=32, help="distinct batches cycled through the timed steps")
    ap.add_argument("--depth", type=int, default=4, help="pipe slots")
    ap.add_argument("--host-threads", type=int, default=0, help="encoder threads of the pipe (0 = the cpus this rank may use)")
    ap.add_argument("--pin", action="store_true", help="bind the encoder threads to the L3 domains of the GPU's NUMA node (pays off when "
                                                       "the caller's buffers live on that node; tools/bench_pipe.py)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the process to the GPU's NUMA node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-linkage-leg", action="store_true")
    ap.add_argument("--no-mm-leg", action="store_true")
    ap.add_argument("--no-resident-leg", action="store_true")
    ap.add_argument("--no-c5-leg", action="store_true")
    ap.add_argument("--no-bam-leg", action="store_true", help="skip the profile_bam end-to-end leg")
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

    want_mm = world == 1 and not args.no_mm_leg
    w = c2_workload(seed=2 + rank, scale=args.scale, with_mm=want_mm)
    n_var = max(1, min(args.variants, args.steps))
    variants = make_variants(w, n_var)
    host_threads = args.host_threads or max(2, min(48, host_cpus() * 3 // 2 // world))    # measured best: 1.5 x the cpu quota
    pipe = engine.Pipe(ctx, max_pos=max(v["n_pos"] for v in variants), max_obs=int(w["n_obs"]),
                       max_splits=max(len(v["split_bounds"]) for v in variants), depth=args.depth, host_threads=host_threads,
                       pin_threads=args.pin, n_mm_bins=1, enable_linkage=False, window=args.window)

    def barrier():
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()

    # Every step hands one batch over from host memory, profiles it once and brings its tables back;
    # consecutive batches overlap in the pipe's three queues.
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

    # configs[4] (C5) sharded over the ranks: every rank streams its shard; rank 0 reports
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
        pipe.close()
        c5 = c5_leg(ctx, rank, world, host_threads, barrier, dist_info, with_cpu=(world == 1 and not args.no_cpu_baseline), scale=args.scale)

    if rank == 0:
        st = [s for s, _ in stats]
        mean = lambda k: float(np.mean([s[k] for s in st])) if st else 0.0
        k_ms = mean("kernel_ms")                 # dispatch time stamps of every pass of the timed region
        n_pos_v = int(np.mean([v["n_pos"] for v in variants]))
        abytes = pileup_algorithmic_bytes(w["n_obs"], n_pos_v, 0, dense=True, record_bytes=st[0]["record_bytes"] if st else 2)
        abytes8 = pileup_algorithmic_bytes(w["n_obs"], n_pos_v, 0, dense=True, record_bytes=8)
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("c2_pileup_bytes_per_launch") if args.scale == 1.0 else None
            except Exception:
                traffic = None
        ms_step = dt / args.steps * 1e3
        h2d_b, d2h_b = mean("h2d_bytes"), mean("d2h_bytes")
        out = {
            "metric": "Gbp profiled/s", "value": units * args.steps / dt / 1e9, "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "C2 streamed: one 5 Mbp genome (0.1 Gbp of reads) per batch, 20x, 2x150 bp pairs, insert N(350,30), "
                                   "--skip_mm_profiling (1 mm bin), linkage off; every batch handed over from host memory "
                                   "(8-byte records -> 2-byte records -> pinned hipMemcpyAsync), profiled once (pileup + SNV call), "
                                   "tables copied back; %d distinct batches" % n_var,
                       "genome_bp": int(w["n_pos"]), "kept_observations": int(w["n_obs"]),
                       "profiled_bases_per_batch": int(w["profiled_bases"]), "splits": int(len(variants[0]["split_bounds"]) - 1),
                       "distinct_batches": n_var, "pipe_depth": args.depth, "host_threads": host_threads,
                       "process_bound_to_numa_node": numa_node,
                       "parallelism": "scaffold-sharded x%d" % world, "scale": args.scale},
            "roofline": {"bound": "hbm", "kernel": "k_pileup_dense", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "record_bytes": st[0]["record_bytes"] if st else 2,
                         "gbs_at_8_bytes_per_observation": abytes8 / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                         "kernel_ms_avg": k_ms, "launches": len(st),
                         "note": "durations = the dispatches' own time stamps inside the timed (streamed) region"},
            "roofline_pcie": {"bound": "pcie", "direction": "host->device", "achieved": h2d_b / (ms_step * 1e-3) / 1e9,
                              "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": h2d_b / (ms_step * 1e-3) / 1e9 / PCIE_PEAK_GBS,
                              "bytes_per_step": h2d_b, "copy_ms_avg": mean("h2d_ms"),
                              "during_copy_gbs": h2d_b / (mean("h2d_ms") * 1e-3) / 1e9 if mean("h2d_ms") > 0 else 0.0,
                              "device_to_host": {"bytes_per_step": d2h_b, "copy_ms_avg": mean("d2h_ms"),
                                                 "achieved": d2h_b / (ms_step * 1e-3) / 1e9}},
            "stages_ms": {"host_encode": mean("encode_ms"), "copy_in": mean("h2d_ms"), "kernel": k_ms, "copy_out": mean("d2h_ms"),
                          "collect_wait": mean("collect_wait_ms"), "step": ms_step,
                          "host_bytes_read_per_step": int(w["n_obs"]) * 8 + n_pos_v},
            "snv_rows": int(stats[-1][1]["n_snv"]) if stats else 0, "snp_sites": int(stats[-1][1]["n_sites"]) if stats else 0,
        }
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if c5 is not None:
            out["c5"] = c5
        if args.only_c5:
            args.no_resident_leg = args.no_mm_leg = args.no_linkage_leg = args.no_cpu_baseline = True
            want_mm = False
        if world == 1 and not args.no_resident_leg:
            out["resident"] = resident_leg(ctx, w, args.window)
            # The roofline of the dominant kernel is priced on the kernel having the GPU to itself (10 blocking runs over a
            # resident batch, the dispatch's own time stamps = what rocprofv3 reports): in the streamed region a launch is 3 %
            # of a PCIe-bound step, the GPU idles between launches and its clocks sag (kernel_ms_in_stream).
            r = out["roofline"]
            k_alone = out["resident"]["kernel_ms_alone"]
            ab = pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, dense=True, record_bytes=r["record_bytes"])
            r.update({"kernel_ms_in_stream": r["kernel_ms_avg"], "frac_in_stream": r["frac"], "kernel_ms_avg": k_alone,
                      "algorithmic_bytes_per_launch": ab, "achieved": ab / (k_alone * 1e-3) / 1e9,
                      "frac": ab / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "gbs_at_8_bytes_per_observation": pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, True, 8) / (k_alone * 1e-3) / 1e9,
                      "note": "kernel alone over a resident C2 batch (dispatch time stamps, 10 blocking runs); kernel_ms_in_stream = "
                              "the same kernel inside the PCIe-bound streamed region (idle GPU between launches)"})
        if want_mm:
            out["mm_on"] = mm_leg(ctx, w)
        if world == 1 and not args.no_linkage_leg:
            out["linkage"] = linkage_leg(ctx)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        if world == 1 and not args.no_bam_leg and not args.only_c5:
            try:
                out["profile_bam"] = profile_bam_leg(ctx)
            except Exception as e:                  # never lose the line over the extra leg
                out["profile_bam"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    pipe.close()
    ctx.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
