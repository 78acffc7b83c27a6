#!/usr/bin/env python3
"""tests/golden/make_filter_golden.py -- golden vectors for the read-pair filter modes (build container only).

Writes filter_modes.bam (tests/bamwriter.py: messy paired reads on three scaffolds, some pairs split over two
scaffolds, some read names on three records) and runs the REFERENCE's own paired_read_filter +
filter_scaff2pair2info (inStrain/filter_reads.py:471-532, 201-260; imported under the stub importer of
make_golden.py) on the scaff2pair2info dictionaries that oracle/bam_py.get_paired_reads builds from it, for
pairing_filter in {paired_only, non_discordant, all_reads}, with and without priority reads.  Stores only data:
the BAM, and per mode the resulting scaffold -> {pair: mm} and the read-report tallies (filter_modes.json)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

REFS = [("scafA", 5000), ("scafB", 1800), ("scafC", 9000)]


def build_reads():
    from tests import bamwriter
    reads = bamwriter.random_reads(77, REFS, 1800)
    rng = np.random.Generator(np.random.PCG64(78))
    by_name = {}
    for r in reads:
        by_name.setdefault(r["name"], []).append(r)
    names = sorted(by_name, key=lambda n: int(n[1:]))
    # discordant pairs: the second read moves to another scaffold
    for n in rng.choice(names, 150, replace=False):
        rs = by_name[n]
        if len(rs) == 2:
            t = (rs[1]["tid"] + 1 + int(rng.integers(0, 2))) % 3
            rs[1]["tid"] = t
            rs[1]["pos"] = int(rng.integers(0, REFS[t][1] - 300))
    reads.sort(key=lambda r: (r["tid"], r["pos"]))
    return reads


def main():
    import make_golden as mg
    mods = mg.import_reference()
    import inStrain.filter_reads as fr
    from oracle import bam_py
    from tests import bamwriter
    reads = build_reads()
    path = os.path.join(HERE, "filter_modes.bam")
    bamwriter.write_bam(path, REFS, reads)
    rrefs, rr = bam_py.read_bam(path)
    s2p2i = {}
    for t, (name, ln) in enumerate(rrefs):
        p2i = bam_py.get_paired_reads(rr, t)
        s2p2i[name] = {p: np.array(i, dtype="int64") for p, i in p2i.items()}
    allnames = sorted({p for d in s2p2i.values() for p in d}, key=lambda n: int(n[1:]))
    priority = set(allnames[5::37])
    out = {"refs": REFS, "priority": sorted(priority), "cases": []}
    for mode in ("paired_only", "non_discordant", "all_reads"):
        for pr in (set(), priority):
            kw = dict(pairing_filter=mode, min_read_ani=0.93, min_mapq=1, max_insert_relative=3, min_insert=50)
            # the reference mutates the dictionaries (all_reads): hand it fresh copies
            fresh = {s: {p: i.copy() for p, i in d.items()} for s, d in s2p2i.items()}
            tallys = {}
            try:
                f = fr.paired_read_filter(fresh, priority_reads_set=pr, tallys=tallys, **kw)
                s2p2mm, Rdb = fr.filter_scaff2pair2info(f, tallys, priority_reads_set=pr, **kw)
            except KeyError as e:
                out["cases"].append({"mode": mode, "priority": bool(pr), "keyerror": True})
                print(mode, bool(pr), "reference raised KeyError", e)
                continue
            row = Rdb[Rdb["scaffold"] == "all_scaffolds"].iloc[0]
            case = {"mode": mode, "priority": bool(pr), "params": {k: v for k, v in kw.items() if k != "pairing_filter"},
                    "r2m": {s: {p: int(m) for p, m in d.items()} for s, d in s2p2mm.items()},
                    "tallies": {k: int(row[k]) for k in ("unfiltered_reads", "unfiltered_pairs", "unfiltered_singletons",
                                                         "filtered_pairs", "filtered_singletons")},
                    "median_insert": float(np.median([v[1] for d in f.values() for v in d.values() if v[4] == 2]))}
            out["cases"].append(case)
            print(mode, bool(pr), case["tallies"], "median", case["median_insert"])
    json.dump(out, open(os.path.join(HERE, "filter_modes.json"), "w"))


if __name__ == "__main__":
    main()
