"""Mirror of inStrain/profile/profile_utilities.py for the hot path: profile_split's result
carrier (SplitObject, profile_utilities.py:823-858 and the fields set at :195-211) and the batch
driver that replaces the split worker pool (profile_controller.py:157-271)."""
import logging
import time
import traceback

import numpy as np
import pandas as pd

from .. import engine
from .snv_utilities import C2P, CLASSES, null_model_lut

BASES = np.array(["A", "C", "T", "G", "N"])
LD_COLUMNS = ['r2', 'd_prime', 'r2_normalized', 'd_prime_normalized', 'total', 'countAB', 'countAb', 'countaB',
              'countab', 'allele_A', 'allele_a', 'allele_B', 'allele_b', 'distance', 'position_A', 'position_B',
              'mm', 'scaffold']          # linkage.py:230-249 + calculate_ld :67-71
SNP_COLUMNS = ['scaffold', 'position', 'ref_base', 'A', 'C', 'T', 'G', 'con_base', 'var_base', 'mm',
               'allele_count', 'class', 'cryptic', 'position_coverage']   # snv_utilities.py:118-127, 274-290


class SplitObject():
    '''Holds the profile of an individual split (same attributes as the reference's SplitObject).'''

    def __init__(self):
        pass


def _series_by_mm(pos, mm, val, dtype):
    """shrink_basewise (profile_utilities.py:337-350): dict mm -> sparse pd.Series indexed by position"""
    out = {}
    if len(pos) == 0:
        return out
    order = np.lexsort((pos, mm))
    pos, mm, val = pos[order], mm[order], val[order]
    cuts = np.flatnonzero(np.diff(mm)) + 1
    for s, e in zip(np.r_[0, cuts], np.r_[cuts, len(mm)]):
        out[int(mm[s])] = pd.Series(val[s:e].astype(dtype), index=pos[s:e].astype(np.int64))
    return out


def tables_to_splits(res, split_bounds, split_scaffold, split_number, scaffold_offset, split_seq_len, min_freq,
                     bam_name=None):
    """Batch result (engine.Batch.fetch()) -> list of SplitObject, one per split, in split order."""
    if "entries" in res:
        e = res["entries"]
        clon_r = res["clon_r"]
    else:
        e = engine.dense_to_entries(res["counts"], res["clon"])
        clon_r = res["clon_r"][e["gpos"]]
    snv, ld = res["snv"], res["ld"]
    n = len(split_bounds) - 1
    e_cut = np.searchsorted(e["gpos"], split_bounds)
    s_cut = np.searchsorted(snv["gpos"], split_bounds)
    l_cut = np.searchsorted(ld["gpos_a"], split_bounds)
    # the two row tables are built once for the whole batch (vectorised) and cut per split
    split_of_snv = np.searchsorted(split_bounds, snv["gpos"], side="right") - 1
    split_of_ld = np.searchsorted(split_bounds, ld["gpos_a"], side="right") - 1
    scaff_arr = np.asarray(split_scaffold, dtype=object)
    off_arr = np.asarray(scaffold_offset, dtype=np.int64)
    big_snv = pd.DataFrame({
        'scaffold': scaff_arr[split_of_snv], 'position': snv["gpos"].astype(np.int64) - off_arr[split_of_snv],
        'ref_base': BASES[snv["ref_base"]],
        'A': snv["cnt"][:, 0].astype(np.int64), 'C': snv["cnt"][:, 1].astype(np.int64),
        'T': snv["cnt"][:, 2].astype(np.int64), 'G': snv["cnt"][:, 3].astype(np.int64),
        'con_base': BASES[snv["con_base"]], 'var_base': BASES[snv["var_base"]], 'mm': snv["mm"].astype(np.int64),
        'allele_count': snv["allele_count"].astype(np.int64), 'class': np.array(CLASSES)[snv["cls"]],
        'cryptic': snv["cryptic"].astype(bool), 'position_coverage': snv["cnt"].sum(axis=1).astype(np.int64),
    }, columns=SNP_COLUMNS) if len(snv) else None
    if len(ld):
        pa = ld["gpos_a"].astype(np.int64) - off_arr[split_of_ld]
        pb = ld["gpos_b"].astype(np.int64) - off_arr[split_of_ld]
        big_ld = pd.DataFrame({
            'r2': ld["r2"], 'd_prime': ld["d_prime"], 'r2_normalized': ld["r2_normalized"],
            'd_prime_normalized': ld["d_prime_normalized"],
            'total': ld["total"].astype(np.int64), 'countAB': ld["countAB"].astype(np.int64),
            'countAb': ld["countAb"].astype(np.int64), 'countaB': ld["countaB"].astype(np.int64),
            'countab': ld["countab"].astype(np.int64), 'allele_A': BASES[ld["allele_A"]],
            'allele_a': BASES[ld["allele_a"]], 'allele_B': BASES[ld["allele_B"]], 'allele_b': BASES[ld["allele_b"]],
            'distance': np.abs(pb - pa), 'position_A': pa, 'position_B': pb, 'mm': ld["mm"].astype(np.int64),
            'scaffold': scaff_arr[split_of_ld]}, columns=LD_COLUMNS)
    else:
        big_ld = None
    out = []
    for i in range(n):
        scaff = split_scaffold[i]
        off = scaffold_offset[i]
        S = SplitObject()
        S.scaffold = scaff
        S.split_number = int(split_number[i])
        S.bam = bam_name
        S.length = int(split_seq_len[i])
        S.min_freq = min_freq
        ee = e[e_cut[i]:e_cut[i + 1]]
        pos = ee["gpos"].astype(np.int64) - off
        lvl = ee["cnt"].sum(axis=1)
        k = lvl > 0
        S.covT = _series_by_mm(pos[k], ee["mm"][k], lvl[k], "int32")
        k = ~np.isnan(ee["clon"])
        S.clonT = _series_by_mm(pos[k], ee["mm"][k], ee["clon"][k], "float32")
        rr = clon_r[e_cut[i]:e_cut[i + 1]]      # rarefied clonality: random in the reference, Philox-seeded here
        k = ~np.isnan(rr)
        S.clonTR = _series_by_mm(pos[k], ee["mm"][k], rr[k], "float32")
        if s_cut[i + 1] > s_cut[i]:
            S.raw_snp_table = big_snv.iloc[s_cut[i]:s_cut[i + 1]].reset_index(drop=True)
        else:
            S.raw_snp_table = pd.DataFrame()
        if l_cut[i + 1] > l_cut[i]:
            S.raw_linkage_table = big_ld.iloc[l_cut[i]:l_cut[i + 1]].reset_index(drop=True)
        else:
            S.raw_linkage_table = pd.DataFrame()
        S.log = ""
        out.append(S)
    return out


def estimate_breadth(coverage):
    """profile_utilities.py:548-555"""
    return (-1) * np.exp(-1 * ((0.883) * coverage)) + 1


def calc_snps(Odb, mm):
    """snv_utilities.py:249-272 on a raw_snp_table DataFrame"""
    if len(Odb) == 0:
        return [0, 0, 0, 0, 0]
    db = Odb[Odb['mm'] <= mm].sort_values('mm').drop_duplicates(subset=['position'], keep='last')
    return [len(db[(db['allele_count'] == 1)]), len(db[db['allele_count'] > 1]), len(db),
            len(db[db['class'].isin(['SNS', 'con_SNV', 'pop_SNV'])]), len(db[db['class'].isin(['SNS', 'pop_SNV'])])]


def make_coverage_table(levels, lengt, scaff, SNPTable):
    """Mirror of make_coverage_table (profile_utilities.py:425-506): `levels` = this scaffold's row of
    Batch.summarize() (device aggregates per mm); the SNV-table columns are computed here."""
    table = {k: [] for k in ['scaffold', 'length', 'breadth', 'coverage', 'coverage_median', 'coverage_std',
                             'coverage_SEM', 'nucl_diversity', 'nucl_diversity_median', 'nucl_diversity_rarefied',
                             'nucl_diversity_rarefied_median', 'breadth_minCov', 'breadth_rarefied', 'breadth_expected',
                             'divergent_site_count', 'SNS_count', 'SNV_count', 'consensus_divergent_sites',
                             'population_divergent_sites', 'conANI_reference', 'popANI_reference', 'mm']}
    n = float(lengt)
    for r in levels:
        if not r['present']:
            continue                                    # not a key of covT for this scaffold
        mm = int(r['mm'])
        s1, s2 = float(r['sum_cov']), float(r['sumsq_cov'])
        mean = s1 / n
        var = max(s2 / n - mean * mean, 0.0)
        counted, rare = int(r['counted']), int(r['counted_rarefied'])
        SNS_count, SNV_count, div_site_count, con_snps, pop_snps = calc_snps(SNPTable, mm)
        table['scaffold'].append(scaff)
        table['length'].append(lengt)
        table['breadth'].append(int(r['nonzero']) / lengt)
        table['coverage'].append(mean)
        table['coverage_median'].append(int(r['median_cov']))
        table['coverage_std'].append(np.sqrt(var))
        table['coverage_SEM'].append(np.sqrt(var * n / (n - 1)) / np.sqrt(n) if lengt > 1 else np.nan)
        table['nucl_diversity'].append(1 - r['sum_clon'] / counted if counted else np.nan)
        table['nucl_diversity_median'].append(1 - r['median_clon'] if counted else np.nan)
        table['nucl_diversity_rarefied'].append(1 - r['sum_clon_rarefied'] / rare if rare else np.nan)
        table['nucl_diversity_rarefied_median'].append(1 - r['median_clon_rarefied'] if rare else np.nan)
        table['breadth_minCov'].append(counted / lengt)
        table['breadth_rarefied'].append(rare / lengt)
        table['breadth_expected'].append(estimate_breadth(mean))
        table['divergent_site_count'].append(div_site_count)
        table['SNS_count'].append(SNS_count)
        table['SNV_count'].append(SNV_count)
        table['consensus_divergent_sites'].append(con_snps)
        table['population_divergent_sites'].append(pop_snps)
        table['conANI_reference'].append((counted - con_snps) / counted if counted else 0)
        table['popANI_reference'].append((counted - pop_snps) / counted if counted else 0)
        table['mm'].append(mm)
    return pd.DataFrame(table)


def profile_splits(ctx, scaffolds, sequences, obs, pair, null_model, n_mm_bins, window_length=10000, bam_name=None,
                   **kwargs):
    """Profile every split of `scaffolds` in ONE device batch.

    scaffolds / sequences: names and upper-cased sequences laid end to end in the flat space (in the
    order the observations' gpos assume); obs / pair: packed observations (engine.OBS_DT) and pair ids.
    kwargs as the reference's profile_split: min_cov, min_freq, min_snp, rarefied_coverage.
    Returns {"{scaffold}.{split}": SplitObject} like Sprofile_dict (profile_utilities.py:85).
    """
    from ..synth import iterate_splits
    min_cov = int(kwargs.get('min_cov', 5))
    min_freq = float(kwargs.get('min_freq', .05))
    min_snp = int(kwargs.get('min_snp', 10))
    lut, fb = null_model_lut(null_model)
    ctx.set_null_model(lut, fb)
    bounds, s_scaff, s_num, s_off, s_len = [], [], [], [], []
    off = 0
    for name, seq in zip(scaffolds, sequences):
        for i, (s, e) in enumerate(iterate_splits(len(seq), window_length)):
            bounds.append(off + s)
            s_scaff.append(name); s_num.append(i); s_off.append(off); s_len.append(e - s + 1)
        off += len(seq)
    bounds.append(off)
    ref = np.concatenate([engine.encode_seq(s) for s in sequences])
    b = engine.Batch(ctx, ref, bounds, obs, pair, min_cov=min_cov, min_freq=min_freq, min_snp=min_snp,
                     rarefied_coverage=int(kwargs.get('rarefied_coverage', 5)), n_mm_bins=n_mm_bins,
                     enable_linkage=True, seed=int(kwargs.get('seed', 0)))
    try:
        b.run()
        res = b.fetch()
        levels = None
        if kwargs.get('scaffold_tables') is not None:   # only the merge step's cumulative tables need the device summaries
            scaff_bounds = np.r_[0, np.cumsum([len(q) for q in sequences])]
            levels, _ = b.summarize(scaff_bounds)
    finally:
        b.close()
    splits = tables_to_splits(res, np.asarray(bounds), s_scaff, s_num, s_off, s_len, min_freq, bam_name)
    out = {"{0}.{1}".format(S.scaffold, S.split_number): S for S in splits}
    if kwargs.get('scaffold_tables') is not None:       # cumulative_scaffold_table per scaffold (merge step)
        for i, name in enumerate(scaffolds):
            snp = [S.raw_snp_table for S in splits if S.scaffold == name and len(S.raw_snp_table)]
            snp = pd.concat(snp) if snp else pd.DataFrame()
            kwargs['scaffold_tables'][name] = make_coverage_table(levels[i], len(sequences[i]), name, snp)
    return out


def profile_bam(bam, fasta_db=None, sR2M=None, ISP_loc=None, **kwargs):
    """Mirror of inStrain.profile.profile_bam (profile/__init__.py:7-18).

    bam: path of a sorted BAM; kwargs: the CLI's flags (min_cov, min_freq, min_snp, min_read_ani,
    min_mapq, max_insert_relative, min_insert, skip_mm_profiling, window_length) plus `s2s`
    (scaffold -> upper-cased sequence, controller.py:337) and `null_model` (dict, snv_utilities.py:14-38).
    fasta_db / sR2M are accepted for signature compatibility: split geometry and the read-pair filter
    are recomputed by the C++ front end with the reference's rules (filter_reads.py:885-956, 201-260).
    Returns {"scaffold.split": SplitObject}; a failing batch follows the reference's convention
    (profile_utilities.py:104-111): the exception is logged and the splits are dropped."""
    s2s = kwargs['s2s']
    null_model = kwargs['null_model']
    device = int(kwargs.get('device', 0))
    t = time.strftime('%m-%d %H:%M')
    try:
        ctx = kwargs.get('ctx') or engine.Context(device)
        bf = engine.BamFile(bam)
        obs, pair, bounds, sref = bf.expand(min_read_ani=kwargs.get('min_read_ani', 0.95),
                                            min_mapq=kwargs.get('min_mapq', -1),
                                            max_insert_relative=kwargs.get('max_insert_relative', 3),
                                            min_insert=kwargs.get('min_insert', 50),
                                            skip_mm=bool(kwargs.get('skip_mm_profiling', False)),
                                            window_length=int(kwargs.get('window_length', 10000)), copy=False)
        refs = bf.refs()
        names = [r[0] for r in refs]
        for n, ln, _ in refs:
            if n not in s2s or len(s2s[n]) != ln:
                raise ValueError("scaffold {0} is not in the .fasta / length differs from the .bam header".format(n))
        n_mm = bf.info["max_mm"] + 1
        kw = {k: v for k, v in kwargs.items() if k in ('min_cov', 'min_freq', 'min_snp', 'rarefied_coverage', 'scaffold_tables', 'seed')}
        try:                # obs / pair are views of the front end's arrays: keep it open until the batch is built
            return profile_splits(ctx, names, [str(s2s[n]).upper() for n in names], obs, pair, null_model, n_mm,
                                  window_length=int(kwargs.get('window_length', 10000)), bam_name=bam, **kw)
        finally:
            bf.close()
    except Exception as e:
        print(e)
        traceback.print_exc()
        logging.error("\n{1} DEBUG FAILURE SplitException {0} batch\n".format(bam, t))
        return {}
