"""Mirror of the coverage half of inStrain.genomeUtilities.genomeLevel_coverage_info
(/root/reference/inStrain/genomeUtilities.py:297-365): per genome and mm level the median / SEM / std of the coverage of the
genome's scaffolds laid end to end with masked scaffold edges (generate_genome_coverage_array :932-981), from the device's
per-genome aggregates (engine.Batch.summarize_genomes).  iRep (the rest of that function) is not part of the hot path: its
columns are NaN here."""
import numpy as np
import pandas as pd


def genome_level_rows(levels, genomes, mms=None):
    """levels: [n_genomes, n_mm_bins] GENOME_LEVEL_DT from the device; genomes: names in the same order; mms: the levels to
    report (default: all bins; a level beyond the last bin repeats the last one -- coverage is cumulative over levels <= mm)"""
    n_bins = levels.shape[1]
    mms = list(range(n_bins)) if mms is None else [int(m) for m in mms]
    table = {"mm": [], "genome": [], "coverage_median": [], "coverage_SEM": [], "coverage_std": []}
    for g, genome in enumerate(genomes):
        for mm in mms:
            r = levels[g, min(mm, n_bins - 1)]
            n = int(r["n"])
            table["mm"].append(mm)
            table["genome"].append(genome)
            if n == 0:                              # the reference: covs = pd.Series([0])
                med, sem, std = 0, np.nan, 0.0
            else:
                s, q = int(r["sum_cov"]), int(r["sumsq_cov"])
                ss = (n * q - s * s) / n            # sum of squared deviations, from exact integers
                med = int(r["median_cov"])
                std = float(np.sqrt(ss / n))
                sem = float(np.sqrt(ss / (n - 1)) / np.sqrt(n)) if n > 1 else np.nan
            table["coverage_median"].append(med)
            table["coverage_SEM"].append(sem)
            table["coverage_std"].append(std)
    db = pd.DataFrame(table)
    db["iRep"] = np.nan
    db["iRep_GC_corrected"] = np.nan
    return db
