#!/usr/bin/env python3
"""tests/golden/make_merge_golden.py -- proves that the SplitObjects this package hands back go through the
REFERENCE's own merge step, and stores what that step makes of them (build container only).

Three scaffolds (one split, three splits, two splits at window_length 10000) with mm levels are profiled split by
split with the C oracle; the tables are turned into SplitObjects by instrain_amd.profile.tables_to_splits exactly as
profile_bam does; then the reference's ScaffoldSplitObject (inStrain/profile/profile_utilities.py:719-814, imported
under the stub importer of make_golden.py) takes them: update_splits + merge() -> scaffold_profile with
make_cumulative_tables() (:870-881: _make_snp_table, _parse_Sdb, make_coverage_table).  The single-split scaffold
goes through SplitObject.merge_single_profile (:831-858) of OUR class, called by the reference.
Stored (data only): the packed inputs, and per scaffold the reference's cumulative_scaffold_table,
cumulative_snv_table and merged covT / clonT."""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

SCAFFOLDS = [("single", 4000, 61), ("triple", 25000, 62), ("double", 12500, 63)]


def build_inputs():
    import make_golden as mg
    P, B, M, R, seqs = [], [], [], [], []
    off = 0
    npairs = 0
    for name, L, seed in SCAFFOLDS:
        seq, pos, base, mm, pair = mg.synth_case(seed=seed, mLen=L, depth=22, mm_levels=3, n_sites=L // 120, p_other=0.01,
                                                  ref_ambig=3, self_pairs=0.1)
        seqs.append(seq)
        P.append(pos + off); B.append(base); M.append(mm); R.append(pair + npairs)
        npairs += int(pair.max()) + 1
        if name == "single":        # a read pair at mm level 7 whose kept bases are all non-ACGT: the level exists, counts nothing
            P.append(np.arange(700, 730) + off); B.append(np.full(30, 4, dtype=base.dtype)); M.append(np.full(30, 7, dtype=mm.dtype))
            R.append(np.full(30, npairs, dtype=pair.dtype))
            npairs += 1
        off += L
    return seqs, np.concatenate(P), np.concatenate(B), np.concatenate(M), np.concatenate(R)


def oracle_batch_tables(seqs, pos, base, mm, pair, lut, fb, min_snp=10):
    """every split through the C oracle -> one product-layout result dict (what Batch.fetch() returns)"""
    from instrain_amd._lib import ENTRY_DT, LD_DT, SNV_DT
    from instrain_amd.synth import iterate_splits
    from oracle import oracle
    ent, snv, ld = [], [], []
    bounds, s_scaff, s_num, s_off, s_len = [], [], [], [], []
    off = 0
    for (name, L, _), seq in zip(SCAFFOLDS, seqs):
        for i, (s, e) in enumerate(iterate_splits(L, 10000)):
            r = oracle.profile_split(pos - off, base, mm, pair, seq[s:e + 1], s, lut, fb, min_cov=5, min_freq=0.05, min_snp=min_snp)
            E = np.zeros(len(r["entries"]), dtype=ENTRY_DT)
            E["gpos"] = r["entries"]["pos"] + off; E["mm"] = r["entries"]["mm"]; E["cnt"] = r["entries"]["cnt"]
            E["clon"] = r["entries"]["clon"]; E["clon_rarefied"] = np.nan
            S = np.zeros(len(r["snv"]), dtype=SNV_DT)
            S["gpos"] = r["snv"]["pos"] + off
            for k in ("mm", "con_base", "var_base", "allele_count", "cls", "cryptic", "ref_base", "cnt"):
                S[k] = r["snv"][k]
            Ld = np.zeros(len(r["ld"]), dtype=LD_DT)
            Ld["gpos_a"] = r["ld"]["pos_a"] + off; Ld["gpos_b"] = r["ld"]["pos_b"] + off
            for a, b in (("mm", "mm"), ("total", "total"), ("countAB", "cAB"), ("countAb", "cAb"), ("countaB", "caB"), ("countab", "cab"),
                         ("allele_A", "allele_A"), ("allele_a", "allele_a"), ("allele_B", "allele_B"), ("allele_b", "allele_b"),
                         ("r2", "r2"), ("d_prime", "d_prime")):
                Ld[a] = r["ld"][b]
            Ld["r2_normalized"] = np.nan; Ld["d_prime_normalized"] = np.nan
            for arr, lst, key in ((E, ent, ("gpos", "mm")), (S, snv, ("gpos", "mm")), (Ld, ld, ("gpos_a", "gpos_b", "mm"))):
                lst.append(arr[np.lexsort(tuple(arr[k] for k in reversed(key)))])
            bounds.append(off + s); s_scaff.append(name); s_num.append(i); s_off.append(off); s_len.append(e - s + 1)
        off += L
    bounds.append(off)
    res = {"entries": np.concatenate(ent), "snv": np.concatenate(snv), "ld": np.concatenate(ld)}
    res["clon_r"] = res["entries"]["clon_rarefied"]
    return res, np.asarray(bounds), s_scaff, s_num, s_off, s_len


def main():
    import make_golden as mg
    mods = mg.import_reference()
    pu, su, lk, fa = mods
    from instrain_amd.profile import profile_utilities as ours
    from tests import util
    ours.SCAFFOLD_PROFILE_CLASS = pu.scaffold_profile       # what an integration does once
    lut, fb = util.load_lut()
    nm = su.generate_snp_model(mg.REF + "/inStrain/helper_files/NullModel.txt", fdr=1e-6)
    seqs, pos, base, mm, pair = build_inputs()
    res, bounds, s_scaff, s_num, s_off, s_len = oracle_batch_tables(seqs, pos, base, mm, pair, lut, fb)
    splits = ours.tables_to_splits(res, bounds, s_scaff, s_num, s_off, s_len, 0.05, "x.bam")
    out = {"pos": pos.astype(np.int32), "base": base.astype(np.uint8), "mm": mm.astype(np.int32), "pair": pair.astype(np.int32),
           "seqs": np.array(seqs), "names": np.array([s[0] for s in SCAFFOLDS]), "lengths": np.array([s[1] for s in SCAFFOLDS])}
    for name, L, _ in SCAFFOLDS:
        mine = [S for S in splits if S.scaffold == name]
        Sp = pu.ScaffoldSplitObject(len(mine))              # the reference's own class
        Sp.scaffold = name
        Sp.null_model = nm
        for S in mine:
            Sp = Sp.update_splits(S.split_number, S)
        assert Sp.ready()
        prof = Sp.merge()
        assert prof is not None, "the reference's merge failed on our SplitObjects"
        assert type(prof).__module__.startswith("inStrain"), type(prof)
        assert prof.length == L
        cst = prof.cumulative_scaffold_table
        snv = prof.cumulative_snv_table
        print(name, "splits", len(mine), "coverage rows", len(cst), "snv rows", len(snv), "mm keys", sorted(prof.covT))
        cst.to_csv(os.path.join(HERE, "merge_%s_cumulative_scaffold_table.csv" % name), index=False)
        snv.to_csv(os.path.join(HERE, "merge_%s_cumulative_snv_table.csv" % name), index=False)
        for att in ("covT", "clonT"):
            d = getattr(prof, att)
            out["%s_%s_mm" % (name, att)] = np.concatenate([np.full(len(d[m]), m) for m in sorted(d)]) if d else np.zeros(0, int)
            out["%s_%s_pos" % (name, att)] = np.concatenate([d[m].index.values for m in sorted(d)]) if d else np.zeros(0, int)
            out["%s_%s_val" % (name, att)] = np.concatenate([d[m].values for m in sorted(d)]) if d else np.zeros(0)
    np.savez_compressed(os.path.join(HERE, "merge_inputs.npz"), **out)


if __name__ == "__main__":
    main()
