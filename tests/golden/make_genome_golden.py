#!/usr/bin/env python3
"""tests/golden/make_genome_golden.py -- golden vectors of the genome-level coverage roll-up.

Runs ONLY in the build container (needs /root/reference): imports the reference's own
inStrain.genomeUtilities.genomeLevel_coverage_info (genomeUtilities.py:297-365) under the stub importer of
make_golden.py (Bio / lmfit / pysam / h5py / seaborn are not installed; iRep is therefore not pinned and not stored),
feeds it a synthetic covT (scaffold -> mm -> sparse coverage Series) for three genomes -- long and short scaffolds, one
shorter than the 2 x 100 masked edge positions, one without any coverage, one absent from covT -- and stores inputs +
the reference's coverage_median / coverage_SEM / coverage_std columns.  Only data is stored.

usage: python tests/golden/make_genome_golden.py
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                        # noqa: E402  (the stub importer)

mg.import_reference()
import inStrain.genomeUtilities as gu                           # noqa: E402

rng = np.random.Generator(np.random.PCG64(2024))
s2l = {"g1_a": 5000, "g1_b": 1200, "g1_short": 150, "g2_a": 3000, "g2_empty": 900, "g2_absent": 700, "g3_tiny": 199, "g3_b": 201}
genome2scaffolds = {"g1": {"g1_a", "g1_b", "g1_short"}, "g2": {"g2_a", "g2_empty", "g2_absent"}, "g3": {"g3_tiny", "g3_b"},
                    "g4_not_relevant": {"g1_a"}}
relevant = {"g1", "g2", "g3"}
mms = [0, 1, 2, 5]
covT, flat = {}, []
for sc, ln in s2l.items():
    if sc == "g2_absent":
        continue
    covT[sc] = {}
    for mm in (0, 1, 3):
        if sc == "g2_empty":
            covT[sc][mm] = pd.Series(np.zeros(0, dtype="int32"), index=np.zeros(0, dtype=np.int64))
            continue
        k = np.sort(rng.choice(ln, size=int(ln * (0.7, 0.3, 0.1)[(0, 1, 3).index(mm)]), replace=False))
        v = rng.integers(1, (40, 9, 4)[(0, 1, 3).index(mm)], size=len(k)).astype("int32")
        covT[sc][mm] = pd.Series(v, index=k.astype(np.int64))
        flat.append(np.c_[np.full(len(k), list(s2l).index(sc)), np.full(len(k), mm), k, v])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = gu.genomeLevel_coverage_info(covT, genome2scaffolds, relevant, s2l, None, mms)
ref = ref[["mm", "genome", "coverage_median", "coverage_SEM", "coverage_std"]]
ref.to_csv(os.path.join(HERE, "genome_coverage.csv"), index=False)
flat = np.concatenate(flat).astype(np.int64)
np.savez_compressed(os.path.join(HERE, "genome_coverage_inputs.npz"), scaffolds=np.array(list(s2l)), lengths=np.array(list(s2l.values())),
                    cov=flat, mms=np.array(mms),
                    genome_of=np.array([[g, s] for g, ss in genome2scaffolds.items() if g in relevant for s in sorted(ss)]))
print(ref)
