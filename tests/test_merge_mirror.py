"""The merge step on this package's SplitObjects.  tests/golden/make_merge_golden.py pushed the same SplitObjects
through the REFERENCE's ScaffoldSplitObject.merge / make_cumulative_tables and stored the result; here the mirror
(instrain_amd.profile.profile_utilities.scaffold_profile) must reproduce those tables -- from the oracle's tables on
the CPU, and (GPU) from the device's tables and the device summaries."""
import importlib.util
import os

import numpy as np
import pandas as pd
import pytest

from tests import util

NAMES = ["single", "triple", "double"]
RANDOM = ["nucl_diversity_rarefied", "nucl_diversity_rarefied_median", "breadth_rarefied"]    # clonTR is random in the reference


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_merge_golden", os.path.join(util.GOLD, "make_merge_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _inputs():
    z = np.load(os.path.join(util.GOLD, "merge_inputs.npz"))
    return z, [str(s) for s in z["seqs"]], z["pos"].astype(np.int64), z["base"], z["mm"].astype(np.int64), z["pair"].astype(np.int64)


def check_profile(P, name, z, skip_random=True):
    g = pd.read_csv(os.path.join(util.GOLD, "merge_%s_cumulative_scaffold_table.csv" % name))
    t = P.cumulative_scaffold_table
    assert list(t.columns) == list(g.columns) and len(t) == len(g)
    for c in g.columns:
        if skip_random and c in RANDOM:
            continue
        if c == "scaffold":
            assert (t[c].values == g[c].values).all()
        else:
            a, b = t[c].values.astype(float), g[c].values.astype(float)
            assert (np.isnan(a) == np.isnan(b)).all(), c
            assert np.nanmax(np.abs(a - b)) <= 1e-9 if (~np.isnan(a)).any() else True, (name, c, a, b)
    gs = pd.read_csv(os.path.join(util.GOLD, "merge_%s_cumulative_snv_table.csv" % name))
    s = P.cumulative_snv_table.sort_values(["position", "mm"]).reset_index(drop=True)
    gs = gs.sort_values(["position", "mm"]).reset_index(drop=True)
    assert sorted(s.columns) == sorted(gs.columns) and len(s) == len(gs)
    for c in gs.columns:
        if c in ("var_freq", "con_freq", "ref_freq"):
            a, b = s[c].values.astype(float), gs[c].values.astype(float)
            assert (np.isnan(a) == np.isnan(b)).all() and np.nanmax(np.abs(a - b)) <= 1e-12, c
        else:
            assert (s[c].astype(str).values == gs[c].astype(str).values).all(), c
    for att in ("covT", "clonT"):
        d = getattr(P, att)
        mm = np.concatenate([np.full(len(d[m]), m) for m in sorted(d)]) if d else np.zeros(0, int)
        pos = np.concatenate([d[m].index.values for m in sorted(d)]) if d else np.zeros(0, int)
        val = np.concatenate([d[m].values for m in sorted(d)]) if d else np.zeros(0)
        o = np.lexsort((pos, mm))
        go = np.lexsort((z["%s_%s_pos" % (name, att)], z["%s_%s_mm" % (name, att)]))
        assert (mm[o] == z["%s_%s_mm" % (name, att)][go]).all() and (pos[o] == z["%s_%s_pos" % (name, att)][go]).all()
        assert (val[o].astype(np.float32) == z["%s_%s_val" % (name, att)][go].astype(np.float32)).all(), att


def test_mirror_merge_of_oracle_tables_equals_reference_merge():
    from instrain_amd.profile import profile_utilities as ours
    m = _golden_module()
    lut, fb = util.load_lut()
    z, seqs, pos, base, mm, pair = _inputs()
    res, bounds, s_scaff, s_num, s_off, s_len = m.oracle_batch_tables(seqs, pos, base, mm, pair, lut, fb)
    splits = ours.tables_to_splits(res, bounds, s_scaff, s_num, s_off, s_len, 0.05, "x.bam")
    for name in NAMES:
        mine = [S for S in splits if S.scaffold == name]
        assert len(mine) == {"single": 1, "triple": 3, "double": 2}[name]
        check_profile(ours.scaffold_profile.from_splits(mine), name, z)
    # the single-split scaffold through the reference-shaped entry point
    class Holder:
        null_model = None
    P = [S for S in splits if S.scaffold == "single"][0].merge_single_profile(Holder())
    check_profile(P, "single", z)


def test_split_objects_realise_their_fields_on_demand():
    """tables_to_splits makes one cheap object per split (a 1000-genome database is tens of thousands a batch): the plain fields and
    the WorkerLog lines (logUtils.py:939-975 format) appear on first access, pickling carries the split's own tables"""
    import pickle
    from instrain_amd.profile import profile_utilities as ours
    m = _golden_module()
    lut, fb = util.load_lut()
    z, seqs, pos, base, mm, pair = _inputs()
    res, bounds, s_scaff, s_num, s_off, s_len = m.oracle_batch_tables(seqs, pos, base, mm, pair, lut, fb)
    splits = ours.tables_to_splits(res, bounds, s_scaff, s_num, s_off, s_len, 0.05, "x.bam", started=123.5)
    assert len(splits) == len(s_scaff) and all(list(vars(S)) == ["_src"] for S in splits)          # nothing realised yet
    for i, S in enumerate(splits):
        assert hasattr(S, "log") and not hasattr(S, "no_such_field")
        assert (S.scaffold, S.split_number, S.bam, S.length, S.min_freq) == (s_scaff[i], int(s_num[i]), "x.bam", int(s_len[i]), 0.05)
        lines = S.log.split("\n")
        assert lines[0] == "" and len(lines) == 3
        for ln, status in zip(lines[1:], ("start", "end")):
            w = ln.split()
            assert w[:4] == ["WorkerLog", "SplitProfile", "%s.%d" % (s_scaff[i], int(s_num[i])), status] and len(w) == 7
        assert float(lines[1].split()[5]) == 123.5
    T = pickle.loads(pickle.dumps(splits[1]))
    assert "_src" not in vars(T) and T.scaffold == splits[1].scaffold and T.log == splits[1].log
    assert T.raw_snp_table.equals(splits[1].raw_snp_table) and sorted(T.covT) == sorted(splits[1].covT)


@pytest.mark.gpu
def test_device_tables_through_the_mirror_merge_equal_reference_merge():
    from instrain_amd import engine
    from instrain_amd.profile import profile_utilities as ours
    from instrain_amd.synth import iterate_splits
    lut, fb = util.load_lut()
    z, seqs, pos, base, mm, pair = _inputs()
    ctx = engine.Context(0)
    ctx.set_null_model(lut, fb)
    names = [str(n) for n in z["names"]]
    lens = [int(x) for x in z["lengths"]]
    bounds, s_scaff, s_num, s_off, s_len = [], [], [], [], []
    off = 0
    for name, L in zip(names, lens):
        for i, (s, e) in enumerate(iterate_splits(L, 10000)):
            bounds.append(off + s); s_scaff.append(name); s_num.append(i); s_off.append(off); s_len.append(e - s + 1)
        off += L
    bounds.append(off)
    order = np.argsort(pos, kind="stable")                  # the device wants a position-clustered stream
    obs = engine.pack_obs(pos[order].astype(np.uint32), base[order], mm[order])
    b = engine.Batch(ctx, np.concatenate([engine.encode_seq(s) for s in seqs]), bounds, obs, pair[order].astype(np.uint32),
                     n_mm_bins=int(mm.max()) + 1, min_cov=5, min_freq=0.05, min_snp=10, rarefied_coverage=50)
    b.run()
    res = b.fetch()
    levels, _ = b.summarize(np.r_[0, np.cumsum(lens)])
    b.close()
    ctx.close()
    splits = ours.tables_to_splits(res, np.asarray(bounds), s_scaff, s_num, s_off, s_len, 0.05, "x.bam")
    for j, name in enumerate(names):
        mine = [S for S in splits if S.scaffold == name]
        P = ours.scaffold_profile.from_splits(mine)
        check_profile(P, name, z)
        # the device summaries give the same rows without any per-position table on the host
        t = ours.make_coverage_table(levels[j], lens[j], name, P.raw_snp_table)
        g = pd.read_csv(os.path.join(util.GOLD, "merge_%s_cumulative_scaffold_table.csv" % name))
        for c in g.columns:
            if c in RANDOM or c == "scaffold":
                continue
            a, bb = t[c].values.astype(float), g[c].values.astype(float)
            assert (np.isnan(a) == np.isnan(bb)).all() and (np.nanmax(np.abs(a - bb)) <= 1e-9 if (~np.isnan(a)).any() else True), (name, c)
