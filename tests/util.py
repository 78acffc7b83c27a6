"""Shared comparison helpers: canonical (sorted) table forms of a split result.

The reference's own tests compare tables after a canonical sort because its row order is
queue-arrival dependent (test/tests/test_profile.py:893-894,903-904); so do we.
"""
import os

import numpy as np

BASES = np.array(list("ACTG"))
CLASSES = np.array(["AmbiguousReference", "DivergentSite", "SNS", "SNV", "con_SNV", "pop_SNV"])
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def load_lut():
    z = np.load(os.path.join(GOLD, "null_model_fdr1e-6.npz"))
    return z["lut"].astype(np.int32), int(z["fallback"])


def canon_from_struct(res):
    """structured arrays {entries, snv, ld} (oracle layout or product layout with the same
    field names) -> dict of canonical flat arrays comparable with the golden .npz."""
    e, s, l = res["entries"], res["snv"], res["ld"]
    out = {}
    lvl = e["cnt"].sum(axis=1)
    k = lvl > 0                                           # shrink_basewise drops zeros
    o = np.lexsort((e["pos"][k], e["mm"][k]))
    out["cov_pos"], out["cov_mm"], out["cov_val"] = e["pos"][k][o], e["mm"][k][o], lvl[k][o]
    k = ~np.isnan(e["clon"])
    o = np.lexsort((e["pos"][k], e["mm"][k]))
    out["clon_pos"], out["clon_mm"], out["clon_val"] = e["pos"][k][o], e["mm"][k][o], e["clon"][k][o]
    o = np.lexsort((s["mm"], s["pos"]))
    s = s[o]
    out["snv_position"], out["snv_mm"] = s["pos"], s["mm"]
    for i, b in enumerate("ACTG"):
        out["snv_" + b] = s["cnt"][:, i]
    out["snv_allele_count"] = s["allele_count"]
    out["snv_position_coverage"] = s["position_coverage"]
    refc = np.where(s["ref_base"] < 4, BASES[np.minimum(s["ref_base"], 3)], "N")
    out["snv_ref_is_acgt"] = s["ref_base"] < 4
    out["snv_ref_base"] = refc
    out["snv_con_base"] = BASES[s["con_base"]]
    out["snv_var_base"] = BASES[s["var_base"]]
    out["snv_class"] = CLASSES[s["cls"]]
    out["snv_cryptic"] = s["cryptic"].astype(bool)
    o = np.lexsort((l["mm"], l["pos_b"], l["pos_a"]))
    l = l[o]
    out["ld_position_A"], out["ld_position_B"], out["ld_mm"] = l["pos_a"], l["pos_b"], l["mm"]
    out["ld_distance"], out["ld_total"] = l["distance"], l["total"]
    out["ld_countAB"], out["ld_countAb"], out["ld_countaB"], out["ld_countab"] = l["cAB"], l["cAb"], l["caB"], l["cab"]
    for k2 in ["allele_A", "allele_a", "allele_B", "allele_b"]:
        out["ld_" + k2] = BASES[l[k2]]
    out["ld_r2"], out["ld_d_prime"] = l["r2"], l["d_prime"]
    return out


def canon_from_golden(g):
    """golden .npz (written by tests/golden/make_golden.py) -> same canonical dict."""
    out = {}
    o = np.lexsort((g["cov_pos"], g["cov_mm"]))
    out["cov_pos"], out["cov_mm"], out["cov_val"] = g["cov_pos"][o], g["cov_mm"][o], g["cov_val"][o]
    o = np.lexsort((g["clon_pos"], g["clon_mm"]))
    out["clon_pos"], out["clon_mm"], out["clon_val"] = g["clon_pos"][o], g["clon_mm"][o], g["clon_val"][o]
    for k in g:
        if k.startswith("snv_") or k.startswith("ld_"):
            out[k] = g[k]
    return out


INT_KEYS = ["cov_pos", "cov_mm", "cov_val", "clon_pos", "clon_mm",
            "snv_position", "snv_mm", "snv_A", "snv_C", "snv_T", "snv_G", "snv_allele_count",
            "snv_position_coverage", "snv_con_base", "snv_var_base", "snv_class", "snv_cryptic",
            "ld_position_A", "ld_position_B", "ld_mm", "ld_distance", "ld_total", "ld_countAB",
            "ld_countAb", "ld_countaB", "ld_countab", "ld_allele_A", "ld_allele_a", "ld_allele_B",
            "ld_allele_b"]


def assert_same(got, exp, float_tol=1e-6, what=""):
    """Integer / categorical columns bit-exact; clonality float32-exact; r2 / d_prime within
    float_tol (north_star: 1e-6), NaN pattern identical."""
    for k in INT_KEYS:
        a, b = np.asarray(got[k]), np.asarray(exp[k])
        assert a.shape == b.shape, (what, k, a.shape, b.shape)
        assert (a == b).all(), (what, k, np.nonzero(a != b)[0][:5])
    # reference character of a non-ACGT reference base is whatever the FASTA held; we carry 'N'
    if "snv_ref_is_acgt" in got:
        m = got["snv_ref_is_acgt"]
        assert (np.asarray(got["snv_ref_base"])[m] == np.asarray(exp["snv_ref_base"])[m]).all(), (what, "ref_base")
    else:
        assert (np.asarray(got["snv_ref_base"]) == np.asarray(exp["snv_ref_base"])).all()
    a, b = np.asarray(got["clon_val"], dtype=np.float32), np.asarray(exp["clon_val"], dtype=np.float32)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), (what, "clon_val")
    for k in ["ld_r2", "ld_d_prime"]:
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(exp[k], dtype=np.float64)
        assert a.shape == b.shape, (what, k)
        assert (np.isnan(a) == np.isnan(b)).all(), (what, k, "nan pattern")
        m = ~np.isnan(a)
        if m.any():
            assert np.max(np.abs(a[m] - b[m])) <= float_tol, (what, k, np.max(np.abs(a[m] - b[m])))


def reassemble_segs(gpos, base, mm, pair):
    """Any observation stream (e.g. the goldens' column-major one) -> read segments (engine.SegBatch) the way reads would
    carry them: an observation extends the oldest open segment of its pair with the same mm whose last column lies before
    it and whose start is less than 150 columns back, otherwise it opens a new segment; segments keep the order of their
    first observation, so two observations of one pair at one site stay in arrival order (linkage's self pairs)."""
    from instrain_amd import engine
    open_of = {}
    starts, mms, pairs, lasts, codes = [], [], [], [], []
    for i in range(len(gpos)):
        g, m, p, b = int(gpos[i]), int(mm[i]), int(pair[i]), int(base[i])
        hit = None
        for si in open_of.setdefault(p, []):
            if mms[si] == m and g > lasts[si] and g - starts[si] < 150:
                hit = si
                break
        if hit is None:
            hit = len(starts)
            starts.append(g); mms.append(m); pairs.append(p); lasts.append(g)
            codes.append(np.full(150, 4, dtype=np.uint8))
            open_of[p].append(hit)
        codes[hit][g - starts[hit]] = b if b < 4 else 5
        lasts[hit] = g
    n = len(starts)
    ln = np.asarray([lasts[i] - starts[i] + 1 for i in range(n)], dtype=np.uint8)
    cd = np.stack(codes) if n else np.zeros((0, 150), np.uint8)
    return engine.SegBatch(np.asarray(starts, np.uint32), ln, engine.pack_codes(cd), np.asarray(mms, np.uint8), np.asarray(pairs, np.uint32))


def segs_to_obs(segs):
    """engine.SegBatch -> (gpos, base, mm, pair) of the observations it stands for, segment after segment (base 4 = a
    non-ACGT base that passes the filter)"""
    from instrain_amd import engine
    cd = engine.unpack_codes(segs.bases)
    si, off = np.nonzero((cd < 4) | (cd == 5))
    g = segs.gpos[si].astype(np.int64) + off
    b = np.where(cd[si, off] < 4, cd[si, off], 4).astype(np.uint8)
    m = segs.mm[si] if segs.mm is not None else np.zeros(len(si), np.uint8)
    p = segs.pair[si] if segs.pair is not None else np.zeros(len(si), np.uint32)
    return g, b, m, p


def run_group(cmd, env=None, timeout=300):
    """subprocess.run(cmd, capture_output=True, text=True) in a process group of its own; a command that outlives `timeout` is killed WITH
    its children (a launcher killed alone leaves its ranks behind, holding the GPU for every test after it)."""
    import os
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        raise AssertionError("timed out after %d s: %s\n%s\n%s" % (timeout, " ".join(map(str, cmd)), (out or "")[-1500:], (err or "")[-3000:]))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)
