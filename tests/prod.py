"""Test helper: run the PRODUCT (libinstrain_amd.so through instrain_amd.engine) and reshape
its tables into the oracle's structured-array layout so tests/util.py can compare them."""
import numpy as np

from instrain_amd import engine
from oracle import oracle as orc


def to_oracle_layout(res, gpos_to_pos):
    """res: Batch.fetch() dict. gpos_to_pos: callable mapping flat gpos array -> absolute position."""
    if "entries" in res:
        e = res["entries"]
    else:
        e = engine.dense_to_entries(res["counts"], res["clon"])
    E = np.zeros(len(e), dtype=orc.ENTRY_DT)
    E["pos"] = gpos_to_pos(e["gpos"]); E["mm"] = e["mm"]; E["cnt"] = e["cnt"]; E["clon"] = e["clon"]
    s = res["snv"]
    S = np.zeros(len(s), dtype=orc.SNV_DT)
    S["pos"] = gpos_to_pos(s["gpos"]); S["mm"] = s["mm"]; S["cnt"] = s["cnt"]
    for k in ("ref_base", "con_base", "var_base", "allele_count", "cls", "cryptic"):
        S[k] = s[k]
    S["position_coverage"] = s["cnt"].sum(axis=1)
    l = res["ld"]
    L = np.zeros(len(l), dtype=orc.LD_DT)
    L["pos_a"] = gpos_to_pos(l["gpos_a"]); L["pos_b"] = gpos_to_pos(l["gpos_b"]); L["mm"] = l["mm"]
    L["distance"] = np.abs(L["pos_b"] - L["pos_a"])
    L["total"] = l["total"]
    L["cAB"], L["cAb"], L["caB"], L["cab"] = l["countAB"], l["countAb"], l["countaB"], l["countab"]
    for k in ("allele_A", "allele_a", "allele_B", "allele_b", "r2", "d_prime"):
        L[k] = l[k]
    return {"entries": E, "snv": S, "ld": L}


def run_split(ctx, pos, base, mm, pair, seq, start, n_mm_bins=None, reads=None, planes=False, **kw):
    """One split through the product. pos absolute; only observations inside the split are sent
    (truncate=True of the pileup call, profile_utilities.py:150).  reads = "stream": the same observations handed over as
    read segments cut from the stream as it comes (synth.segs_from_obs); "reassembled": as segments rebuilt per read pair
    (tests.util.reassemble_segs) -- the read-level hand-over instead of the observation records."""
    pos = np.asarray(pos, dtype=np.int64)
    sel = (pos >= start) & (pos < start + len(seq))
    mm = np.asarray(mm)
    if n_mm_bins is None:
        n_mm_bins = int(mm.max()) + 1 if len(mm) else 1
    obs = engine.pack_obs((pos[sel] - start).astype(np.uint32), np.asarray(base)[sel], mm[sel])
    pr = np.asarray(pair)[sel].astype(np.uint32)
    if reads == "stream":
        from instrain_amd import synth
        obs, pr = synth.segs_from_obs(obs, pr), None
    elif reads == "reassembled":
        from tests import util
        obs, pr = util.reassemble_segs(obs["gpos"], obs["base"], obs["mm"], pr), None
    if planes:
        # the same segments as bit planes through a one-slot pipe (isx_pipe_submit_planes; one mm bin): full tables back (want_counts)
        assert reads in ("stream", "reassembled") and n_mm_bins == 1
        ref = engine.encode_seq(seq)
        pipe = engine.Pipe(ctx, max_pos=len(ref), max_obs=0, max_segs=max(1, obs.n_seg), max_splits=1, depth=1, host_threads=2, pin_threads=False,
                           n_mm_bins=1, enable_linkage=True, want_counts=True, **kw)
        t = pipe.submit_planes(engine.RefPlanes.from_codes(ref), [0, len(ref)], engine.PlaneBatch.from_segs(obs))
        r = pipe.collect(t)
        res = {k: r[k].copy() for k in ("counts", "clon", "clon_r", "snv", "ld")}
        sizes = r["sizes"]
        pipe.release(t)
        pipe.close()
        out = to_oracle_layout(res, lambda g: g.astype(np.int64) + start)
        out["n_edges"] = sizes["n_edges"]
        out["sizes"] = sizes
        return out
    b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], obs, pr,
                     n_mm_bins=n_mm_bins, **kw)
    b.run()
    res = b.fetch()
    sizes = b.sizes()
    b.close()
    out = to_oracle_layout(res, lambda g: g.astype(np.int64) + start)
    out["n_edges"] = sizes["n_edges"]
    out["sizes"] = sizes
    return out
