/*
 * oracle/oracle_core.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C, single-threaded restatement of the reference's per-split hot path,
 * column by column, in the reference's own order of evaluation:
 *
 *   process_bam_sites        /root/reference/inStrain/profile/profile_utilities.py:218-266
 *   get_base_counts_mm       profile_utilities.py:268-286
 *   update_covT              profile_utilities.py:288-295
 *   mm_counts_to_counts      profile_utilities.py:297-312
 *   update_snp_table         /root/reference/inStrain/profile/snv_utilities.py:40-145
 *   call_snv_site            snv_utilities.py:147-196
 *   calc_snp_class           snv_utilities.py:198-223  (+ readComparer.py:307-316 is_present)
 *   calculate_clonality      snv_utilities.py:225-231
 *   generate_snp_table       snv_utilities.py:274-290  (cryptic, position_coverage)
 *   update_linked_reads      /root/reference/inStrain/profile/linkage.py:254-283
 *   calc_mm_SNV_linkage_network  linkage.py:14-44
 *   calculate_ld / _iterator_ld_sites / major_minor_allele / _calc_ld_single
 *                            linkage.py:46-75, 78-131, 133-136, 138-240
 *
 * Input is the packed observation stream of ONE split: for every (pileup column,
 * pileup read) visit on which the reference touches its `table` (read is in R2M, not
 * del/refskip, base quality >= 30 after htslib overlap resolution) one record
 * (pos, base, mm, pair) in BAM arrival order; base 0..3 = A,C,T,G (P2C order,
 * profile_utilities.py:34), 4 = anything else (creates the mm level but counts nothing,
 * profile_utilities.py:279-285).
 *
 * The two unseeded-random outputs of the reference (clonTR, *_normalized;
 * snv_utilities.py:233-247, linkage.py:200-228) are not restated: the reference's own
 * tests delete them before comparing (test/tests/test_profile.py:896-900).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile). Floating point is
 * plain IEEE fp64 in source order, exactly as CPython evaluates the reference's expressions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t pos;        /* absolute position on the scaffold */
    int32_t mm;
    int64_t cnt[4];     /* counts of THIS mm level (A,C,T,G) */
    float clon;         /* clonT[mm][pos] (float32 store, snv_utilities.py:94-96); NaN if unset */
} orc_entry;

typedef struct {
    int32_t pos;
    int32_t mm;
    int64_t cnt[4];     /* cumulative counts over levels <= mm */
    int8_t ref_base;    /* 0..3, 4 = non-ACGT reference */
    int8_t con_base;
    int8_t var_base;
    int8_t allele_count;
    int8_t cls;         /* 0 AmbiguousReference 1 DivergentSite 2 SNS 3 SNV 4 con_SNV 5 pop_SNV */
    int8_t cryptic;
    int64_t position_coverage;
} orc_snv;

typedef struct {
    int32_t pos_a, pos_b, mm;
    int32_t distance;
    int64_t total, cAB, cAb, caB, cab;
    int8_t allele_A, allele_a, allele_B, allele_b;
    double r2, d_prime;
} orc_ld;

typedef struct {
    int64_t n_entries, n_snv, n_ld;
    orc_entry *entries;
    orc_snv *snv;
    orc_ld *ld;
    int64_t n_edges;        /* graph edges (distinct (p1,p2) incl. self pairs) */
    int64_t n_increments;   /* pair increments performed while building the graph */
} orc_result;

#define VEC(T) struct { T *d; int64_t n, cap; }
#define VPUSH(v, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? (v).cap * 2 : 256; \
    (v).d = realloc((v).d, (size_t)(v).cap * sizeof(*(v).d)); } (v).d[(v).n++] = (x); } while (0)

static int base_of_char(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'T': return 2; case 'G': return 3; }
    return 4;
}

/* snv_utilities.py:174-177 / readComparer.py:311-314 */
static int lut_min_bases(const int32_t *lut, int64_t lut_n, int32_t fallback, int64_t total)
{
    if (total >= 0 && total < lut_n && lut[total] >= 0) return lut[total];
    return fallback;
}

static int argmax4(const int64_t *c)           /* np.argmax: first maximum */
{
    int b = 0;
    for (int k = 1; k < 4; k++) if (c[k] > c[b]) b = k;
    return b;
}

/* snv_utilities.py:147-196.  returns: -2 = None (uncounted), -1 = not a SNP, 0..3 = base */
static int call_snv_site(const int64_t *counts, int ref_base, const int32_t *lut, int64_t lut_n,
                         int32_t fallback, int64_t min_cov, double min_freq, int *morphia)
{
    int64_t total = counts[0] + counts[1] + counts[2] + counts[3];
    *morphia = 0;
    if (total < min_cov) return -2;
    int i = 0;
    int min_bases = lut_min_bases(lut, lut_n, fallback, total);
    for (int k = 0; k < 4; k++)
        if (counts[k] >= min_bases && (double)counts[k] / (double)total >= min_freq) i++;
    *morphia = i;
    int am = argmax4(counts);
    if (i > 1) return am;
    if (i == 1 && am != ref_base) return am;
    if (i == 0) return am;
    return -1;
}

/* snv_utilities.py:225-231 */
static double clonality(const int64_t *c)
{
    int64_t s = c[0] + c[1] + c[2] + c[3];
    double ds = (double)s;
    double prob = ((double)c[0] / ds) * ((double)c[0] / ds) + ((double)c[1] / ds) * ((double)c[1] / ds)
                + ((double)c[2] / ds) * ((double)c[2] / ds) + ((double)c[3] / ds) * ((double)c[3] / ds);
    return prob;
}

/* snv_utilities.py:198-223 */
static int snp_class(int con, int ref, int var, const int64_t *counts, int allele_count,
                     const int32_t *lut, int64_t lut_n, int32_t fallback, double min_freq)
{
    if (ref > 3) return 0;
    if (allele_count == 0) return 1;
    if (allele_count == 1) return 2;
    if (ref == con) return 3;
    if (ref == var) return 4;
    int64_t total = counts[0] + counts[1] + counts[2] + counts[3];
    int min_bases = lut_min_bases(lut, lut_n, fallback, total);
    if (counts[ref] >= min_bases && ((double)counts[ref] / (double)total) >= min_freq) return 4;
    return 5;
}

typedef struct { int32_t pair, mm, pos; int8_t base; } allele_obs;
typedef struct { int32_t p1, p2, mm; int8_t b1, b2; } incr;
typedef struct { int32_t pos; int32_t n_levels; int64_t first; } snp_site;   /* index into site_levels */
typedef struct { int32_t mm; int64_t cnt[4]; } site_level;

static int cmp_incr(const void *x, const void *y)
{
    const incr *a = x, *b = y;
    if (a->p1 != b->p1) return a->p1 < b->p1 ? -1 : 1;
    if (a->p2 != b->p2) return a->p2 < b->p2 ? -1 : 1;
    if (a->mm != b->mm) return a->mm < b->mm ? -1 : 1;
    return 0;
}

static const snp_site *find_site(const snp_site *s, int64_t n, int32_t pos)
{
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        if (s[mid].pos == pos) return &s[mid];
        if (s[mid].pos < pos) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

/* mm_counts_to_counts(s, mm) on a snv2mm2counts entry (profile_utilities.py:297-312) */
static void site_cum(const snp_site *s, const site_level *lv, int32_t mm, int64_t *out, int *has_mm)
{
    out[0] = out[1] = out[2] = out[3] = 0;
    *has_mm = 0;
    for (int i = 0; i < s->n_levels; i++) {
        const site_level *l = &lv[s->first + i];
        if (l->mm == mm) *has_mm = 1;
        if (l->mm <= mm) for (int k = 0; k < 4; k++) out[k] += l->cnt[k];
    }
}

/* linkage.py:133-136: sorted(d, key=d.get, reverse=True) over keys A,C,T,G -- stable */
static void major_minor(const int64_t *c, int *maj, int *min_)
{
    int order[4] = {0, 1, 2, 3};
    for (int i = 1; i < 4; i++) {           /* insertion sort, descending, stable */
        int v = order[i], j = i - 1;
        while (j >= 0 && c[order[j]] < c[v]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    *maj = order[0]; *min_ = order[1];
}

void orc_free(orc_result *r)
{
    if (!r) return;
    free(r->entries); free(r->snv); free(r->ld);
    free(r);
}

/*
 * profile_split (profile_utilities.py:115-216) minus the pileup iterator itself.
 *   seq/mLen/start : split sequence (upper-cased), its length, absolute start
 *   lut[0..lut_n)  : null model; lut[c] < 0 means "coverage c missing" -> fallback (model[-1])
 */
orc_result *orc_profile_split(int64_t n_obs, const int32_t *pos, const uint8_t *base, const int32_t *mm,
                              const int32_t *pair, const char *seq, int32_t mLen, int32_t start,
                              const int32_t *lut, int64_t lut_n, int32_t fallback,
                              int64_t min_cov, double min_freq, int64_t min_snp)
{
    orc_result *R = calloc(1, sizeof(*R));
    VEC(orc_entry) E = {0};
    VEC(orc_snv) S = {0};
    VEC(orc_ld) L = {0};
    VEC(allele_obs) AO = {0};
    VEC(snp_site) sites = {0};
    VEC(site_level) slev = {0};

    /* group observations by column, preserving arrival order (counting sort on rel. position) */
    int64_t *colstart = calloc((size_t)mLen + 1, sizeof(int64_t));
    int64_t n_in = 0;
    for (int64_t i = 0; i < n_obs; i++) {
        int64_t rp = (int64_t)pos[i] - start;
        if (rp < 0 || rp >= mLen) continue;        /* truncate=True */
        colstart[rp + 1]++; n_in++;
    }
    for (int32_t p = 0; p < mLen; p++) colstart[p + 1] += colstart[p];
    int64_t *order = malloc((size_t)(n_in ? n_in : 1) * sizeof(int64_t));
    {
        int64_t *fill = malloc((size_t)mLen * sizeof(int64_t));
        memcpy(fill, colstart, (size_t)mLen * sizeof(int64_t));
        for (int64_t i = 0; i < n_obs; i++) {
            int64_t rp = (int64_t)pos[i] - start;
            if (rp < 0 || rp >= mLen) continue;
            order[fill[rp]++] = i;
        }
        free(fill);
    }

    int32_t max_mm = 0;
    for (int64_t i = 0; i < n_obs; i++) if (mm[i] > max_mm) max_mm = mm[i];
    int64_t (*lev)[5] = calloc((size_t)max_mm + 1, sizeof(*lev));   /* table[mm][A,C,T,G,present] */

    /* ---- process_bam_sites: one pileup column at a time ---- */
    for (int32_t rp = 0; rp < mLen; rp++) {
        int64_t c0 = colstart[rp], c1 = colstart[rp + 1];
        if (c0 == c1) continue;                     /* no column yielded */
        int32_t lo = max_mm, hi = 0;
        for (int64_t j = c0; j < c1; j++) {         /* get_base_counts_mm */
            int64_t i = order[j];
            int32_t m = mm[i];
            lev[m][4] = 1;                          /* level exists even for a non-ACGT base */
            if (base[i] < 4) lev[m][base[i]]++;
            if (m < lo) lo = m;
            if (m > hi) hi = m;
        }
        int ref_base = base_of_char(seq[rp]);

        /* update_covT + update_snp_table */
        int anySNP = 0, cryptic = 0;
        unsigned bases_mask = 0;
        int64_t cum[4] = {0, 0, 0, 0};
        int64_t first_snv_row = S.n;
        for (int32_t m = lo; m <= hi; m++) {        /* sorted(MMcounts.keys()) */
            if (!lev[m][4]) continue;
            for (int k = 0; k < 4; k++) cum[k] += lev[m][k];     /* mm_counts_to_counts(MMcounts, mm) */
            int morphia;
            int snp = call_snv_site(cum, ref_base, lut, lut_n, fallback, min_cov, min_freq, &morphia);
            orc_entry e;
            e.pos = rp + start; e.mm = m;
            for (int k = 0; k < 4; k++) e.cnt[k] = lev[m][k];
            e.clon = NAN;
            if (cum[0] + cum[1] + cum[2] + cum[3] >= min_cov) e.clon = (float)clonality(cum);
            VPUSH(E, e);
            if (snp == -2) continue;
            if (snp != -1) {
                int64_t tmp[4] = {cum[0], cum[1], cum[2], cum[3]};
                tmp[snp] = 0;
                int var = argmax4(tmp);             /* list.index(max) == first maximum */
                orc_snv s;
                s.pos = rp + start; s.mm = m;
                for (int k = 0; k < 4; k++) s.cnt[k] = cum[k];
                s.ref_base = (int8_t)ref_base; s.con_base = (int8_t)snp; s.var_base = (int8_t)var;
                s.allele_count = (int8_t)morphia;
                s.cls = (int8_t)snp_class(snp, ref_base, var, cum, morphia, lut, lut_n, fallback, min_freq);
                s.cryptic = 0;
                s.position_coverage = cum[0] + cum[1] + cum[2] + cum[3];
                VPUSH(S, s);
                if (morphia >= 2) { anySNP = 1; bases_mask |= 1u << snp; bases_mask |= 1u << var; }
                else if (morphia == 1 && anySNP) cryptic = 1;
            } else if (anySNP) {
                cryptic = 1;
            }
        }
        if (cryptic) for (int64_t r = first_snv_row; r < S.n; r++) S.d[r].cryptic = 1;   /* p2c map */

        if (anySNP) {                               /* update_linked_reads + snv2mm2counts */
            for (int64_t j = c0; j < c1; j++) {
                int64_t i = order[j];
                if (base[i] < 4 && (bases_mask >> base[i]) & 1u) {
                    allele_obs a = { pair[i], mm[i], rp, (int8_t)base[i] };
                    VPUSH(AO, a);
                }
            }
            snp_site st = { rp, 0, slev.n };
            for (int32_t m = lo; m <= hi; m++) {
                if (!lev[m][4]) continue;
                site_level sl; sl.mm = m;
                for (int k = 0; k < 4; k++) sl.cnt[k] = lev[m][k];
                VPUSH(slev, sl); st.n_levels++;
            }
            VPUSH(sites, st);
        }
        for (int32_t m = lo; m <= hi; m++) memset(lev[m], 0, sizeof(lev[m]));
    }

    /* ---- calc_mm_SNV_linkage_network: per (mm, read name) list, all i<j combinations ---- */
    VEC(incr) INC = {0};
    if (AO.n) {
        int32_t max_pair = 0;
        for (int64_t i = 0; i < AO.n; i++) if (AO.d[i].pair > max_pair) max_pair = AO.d[i].pair;
        int64_t *pstart = calloc((size_t)max_pair + 2, sizeof(int64_t));
        for (int64_t i = 0; i < AO.n; i++) pstart[AO.d[i].pair + 1]++;
        for (int32_t p = 0; p <= max_pair; p++) pstart[p + 1] += pstart[p];
        allele_obs *byp = malloc((size_t)AO.n * sizeof(allele_obs));
        int64_t *fill = malloc(((size_t)max_pair + 1) * sizeof(int64_t));
        memcpy(fill, pstart, ((size_t)max_pair + 1) * sizeof(int64_t));
        for (int64_t i = 0; i < AO.n; i++) byp[fill[AO.d[i].pair]++] = AO.d[i];   /* stable */
        for (int32_t p = 0; p <= max_pair; p++) {
            for (int64_t i = pstart[p]; i < pstart[p + 1]; i++)
                for (int64_t j = i + 1; j < pstart[p + 1]; j++) {
                    incr x = { byp[i].pos, byp[j].pos, byp[i].mm, byp[i].base, byp[j].base };
                    VPUSH(INC, x);
                }
        }
        free(pstart); free(byp); free(fill);
    }
    R->n_increments = INC.n;
    if (INC.n) qsort(INC.d, (size_t)INC.n, sizeof(incr), cmp_incr);

    /* ---- calculate_ld over edges ---- */
    for (int64_t i = 0; i < INC.n;) {
        int64_t j = i;
        while (j < INC.n && INC.d[j].p1 == INC.d[i].p1 && INC.d[j].p2 == INC.d[i].p2) j++;
        R->n_edges++;
        int32_t p1 = INC.d[i].p1, p2 = INC.d[i].p2;
        const snp_site *s1 = find_site(sites.d, sites.n, p1);
        const snp_site *s2 = find_site(sites.d, sites.n, p2);
        int64_t combo[4][4];
        memset(combo, 0, sizeof(combo));
        for (int64_t k = i; k < j;) {               /* _iterator_ld_sites: ascending mm on the edge */
            int32_t m = INC.d[k].mm;
            while (k < j && INC.d[k].mm == m) { combo[INC.d[k].b1][INC.d[k].b2]++; k++; }
            int64_t cA[4], cB[4];
            int h1, h2;
            site_cum(s1, slev.d, m, cA, &h1);
            site_cum(s2, slev.d, m, cB, &h2);
            if (!h1 || !h2) continue;               /* mm not in updateMMs */
            int64_t ssum = cA[0] + cA[1] + cA[2] + cA[3] + cB[0] + cB[1] + cB[2] + cB[3];
            if (ssum < min_snp) continue;
            int A, a, B, b;
            major_minor(cA, &A, &a);
            major_minor(cB, &B, &b);
            if (cA[A] == 0 || cA[a] == 0 || cB[B] == 0 || cB[b] == 0) continue;
            int64_t AB = combo[A][B], Ab = combo[A][b], aB = combo[a][B], ab = combo[a][b];
            int64_t total = AB + Ab + aB + ab;
            if (!(total > min_snp)) continue;       /* _calc_ld_single: strict */
            double t = (double)total;
            double fAB = (double)AB / t, fAb = (double)Ab / t, faB = (double)aB / t, fab = (double)ab / t;
            double fA = fAB + fAb, fa = fab + faB, fB = fAB + faB, fb = fab + fAb;
            double linkD = fAB - fA * fB;
            double r2;
            if (fa == 0 || fA == 0 || fB == 0 || fb == 0) r2 = NAN;
            else r2 = linkD * linkD / (fA * fa * fB * fb);
            double linkd = fab - fa * fb;
            double dp = NAN;
            if (linkd < 0) {
                double d1 = (-fA * fB), d2 = (-fa * fb);
                dp = linkd / (d1 > d2 ? d1 : d2);   /* max([..]) returns the first maximum */
            } else if (linkD > 0) {
                double d1 = (fA * fb), d2 = (fa * fB);
                dp = linkd / (d2 < d1 ? d2 : d1);   /* min([..]) returns the first minimum */
            }
            orc_ld row;
            row.pos_a = p1 + start; row.pos_b = p2 + start; row.mm = m;
            row.distance = p2 > p1 ? p2 - p1 : p1 - p2;
            row.total = total; row.cAB = AB; row.cAb = Ab; row.caB = aB; row.cab = ab;
            row.allele_A = (int8_t)A; row.allele_a = (int8_t)a; row.allele_B = (int8_t)B; row.allele_b = (int8_t)b;
            row.r2 = r2; row.d_prime = dp;
            VPUSH(L, row);
        }
        i = j;
    }

    free(colstart); free(order); free(lev);
    free(AO.d); free(sites.d); free(slev.d); free(INC.d);
    R->n_entries = E.n; R->entries = E.d;
    R->n_snv = S.n; R->snv = S.d;
    R->n_ld = L.n; R->ld = L.d;
    return R;
}
