"""Mirror of inStrain/profile/snv_utilities.py pieces that stay on the host."""
import numpy as np

P2C = {'A': 0, 'C': 1, 'T': 2, 'G': 3}      # base -> position (profile_utilities.py:34)
C2P = {0: 'A', 1: 'C', 2: 'T', 3: 'G'}
CLASSES = ["AmbiguousReference", "DivergentSite", "SNS", "SNV", "con_SNV", "pop_SNV"]


def generate_snp_model(model_file, fdr=1e-6):
    """Same contract as the reference (snv_utilities.py:14-38): model[coverage] = 0-based index of the
    first column of that NullModel.txt row whose probability is < fdr; model[-1] = the maximum,
    used for coverages missing from the table."""
    model = {}
    with open(model_file) as f:
        for line in f:
            if 'coverage' in line:
                continue
            fields = line.split()
            for i, count in enumerate(fields[1:]):
                if float(count) < fdr:
                    model[int(fields[0])] = i
                    break
    model[-1] = max(model.values())
    return model


def null_model_lut(model, n=None):
    """dict model -> (int32 lut with -1 for missing coverages, fallback) for isx_set_null_model."""
    if n is None:
        n = max(k for k in model if k >= 0) + 1
    lut = np.full(n + 1, -1, dtype=np.int32)
    for k, v in model.items():
        if 0 <= k <= n:
            lut[k] = v
    return lut, int(model[-1])
