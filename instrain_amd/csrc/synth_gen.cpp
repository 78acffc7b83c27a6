// synth_gen.cpp -- fast generator of the synthetic metagenome workloads of SURVEY.md section 8(d) (C4 / C5) and of
// single-genome workloads of the C2 / C3 shape: the POST-pileup packed observation stream a BAM of such
// reads would expand to (what isx_bam_expand emits), produced directly, multi-threaded and deterministic
// in (seed, genome) whatever the thread count.  Bench / test infrastructure (libisx_synth.so), not part of
// the product library; there is no reference counterpart (the reference has no workload generator).
//
// Model (same as instrain_amd/synth.py make_workload, per contig): uniform random reference; biallelic
// sites at density `site_frac` with allele frequency U(af_lo, af_hi) on two haplotype backgrounds (a
// pair carries the alternative allele with probability af * 1.6 / 0.4 depending on its haplotype, so
// linkage is non-trivial); read pairs 2 x read_len, insert N(mean, sd) clipped at 2 x read_len (mates
// never overlap), start uniform in the contig; per-base error `err` (uniform substitute); a base is kept
// (quality >= 30) with probability p_keep; mm of a pair = its mismatches to the reference (capped), 0 when
// with_mm == 0 (--skip_mm_profiling / --database_mode).  Genome sizes U(len_lo, len_hi), cut into
// `contigs` scaffolds; abundances log-normal(sigma); coverage_g * len_g sums to total_read_bp.  Genomes
// below min_genome_coverage are dropped like the reference's filter_fasta does in --database_mode
// (/root/reference/inStrain/profile/fasta.py:110-136, controller.py:211-214).
#include <stdint.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {        // xoshiro256++ seeded by splitmix64
    uint64_t s[4];
    static uint64_t sm(uint64_t &x) { uint64_t z = (x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = sm(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
    double normal() { double u1 = uni(), u2 = uni(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

uint64_t mix(uint64_t a, uint64_t b, uint64_t c) { uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull ^ (c + 0x1CE4E5B9ull) * 0x94D049BB133111EBull; return Rng::sm(x); }

}  // namespace

extern "C" {

typedef struct {
    int32_t n_genomes, contigs;
    int64_t len_lo, len_hi;
    double total_read_bp;           // sum over genomes of coverage * length (nominal read bases)
    double abundance_sigma;         // 0 = equal coverage
    double min_genome_coverage;     // genomes below are dropped (database_mode: 1); 0 keeps all
    double site_frac, af_lo, af_hi, err, p_keep;
    int32_t read_len;
    double insert_mean, insert_sd;
    int32_t with_mm, max_mm, threads, pad;
    uint64_t seed;
} isx_synth_params;

typedef struct { uint32_t gpos; uint16_t mm; uint8_t base, flags; } synth_obs;

// per-genome plan: length, coverage, kept flag, read pairs it will get (estimate = coverage * len / (2 read_len))
int isx_synth_plan(const isx_synth_params *p, int64_t *length, double *coverage, int32_t *kept, int64_t *pairs)
{
    if (!p || p->n_genomes <= 0 || p->contigs <= 0 || p->len_lo <= 0 || p->len_hi < p->len_lo) return -1;
    std::vector<double> w((size_t)p->n_genomes);
    double wl = 0;
    for (int g = 0; g < p->n_genomes; g++) {
        Rng r(mix(p->seed, (uint64_t)g, 1));
        length[g] = p->len_lo + (int64_t)r.below((uint64_t)(p->len_hi - p->len_lo + 1));
        w[(size_t)g] = p->abundance_sigma > 0 ? std::exp(p->abundance_sigma * r.normal()) : 1.0;
        wl += w[(size_t)g] * (double)length[g];
    }
    for (int g = 0; g < p->n_genomes; g++) {
        coverage[g] = p->total_read_bp * w[(size_t)g] / wl;
        kept[g] = coverage[g] >= p->min_genome_coverage ? 1 : 0;
        pairs[g] = (int64_t)(coverage[g] * (double)length[g] / (2.0 * p->read_len));
    }
    return 0;
}

typedef struct {
    int64_t n_pos, n_obs, n_pairs, n_scaffolds;
    int64_t profiled_bases;         // pairs * 2 * read_len
    int64_t n_sites;
    uint8_t *ref;                   // [n_pos] base codes
    synth_obs *obs;                 // [n_obs], BAM order (scaffold, read start)
    uint32_t *pair;                 // [n_obs]
    int64_t *scaffold_bounds;       // [n_scaffolds + 1]
    int32_t *scaffold_genome;       // [n_scaffolds] index into the caller's genome list
} isx_synth_out;

void isx_synth_free(isx_synth_out *o)
{
    if (!o) return;
    free(o->ref); free(o->obs); free(o->pair); free(o->scaffold_bounds); free(o->scaffold_genome);
    memset(o, 0, sizeof *o);
}

}  // extern "C" (the generator core below is C++)

namespace {

struct Contig { int64_t off, len, n_pairs, obs_off, obs_cap, obs_n, pair0; int genome, contig, plan_genome; uint64_t seed; };

// the contigs of the genomes genome_sel[0..n_sel), laid end to end in that order
int64_t plan_contigs(const isx_synth_params *p, const int32_t *genome_sel, int32_t n_sel, const int64_t *length, const double *coverage,
                     std::vector<Contig> &cs, int64_t &obs_cap, int64_t &pair0)
{
    const int RL = p->read_len, C = p->contigs;
    cs.assign((size_t)n_sel * C, Contig{});
    int64_t off = 0;
    obs_cap = 0; pair0 = 0;
    const int64_t min_ins = 2 * (int64_t)RL;
    for (int i = 0; i < n_sel; i++) {
        const int g = genome_sel[i];
        Rng r(mix(p->seed, (uint64_t)g, 2));
        // contig lengths: exponential weights, every contig >= 1000 positions
        std::vector<double> w((size_t)C);
        double ws = 0;
        for (auto &x : w) { x = -std::log(1.0 - r.uni()) + 0.05; ws += x; }
        const int64_t L = length[g], spare = std::max<int64_t>(0, L - 1000 * (int64_t)C);
        int64_t used = 0;
        for (int c = 0; c < C; c++) {
            int64_t len = 1000 + (int64_t)((double)spare * w[(size_t)c] / ws);
            if (c == C - 1) len = std::max<int64_t>(1, L - used);
            used += len;
            Contig &k = cs[(size_t)i * C + c];
            k.off = off; k.len = len; k.genome = i; k.contig = c; k.plan_genome = g; k.seed = mix(p->seed, (uint64_t)g, 100 + (uint64_t)c);
            const double want = coverage[g] * (double)len / (2.0 * RL);
            k.n_pairs = len >= min_ins + 8 ? (int64_t)(want + r.uni()) : 0;
            k.obs_off = obs_cap; k.obs_cap = k.n_pairs * 2 * RL; k.pair0 = pair0;
            obs_cap += k.obs_cap; pair0 += k.n_pairs; off += len;
        }
    }
    return off;
}

// One contig's reference and reads, deterministic in its seed.  emit(read id i (mates 2j, 2j + 1 of pair j), start, bases[RL],
// keep[RL] (1 = quality >= 30), mismatches of the pair, mismatches of this read) is called read by read in BAM order.
struct ContigGen {
    std::vector<int32_t> site_of;
    std::vector<uint8_t> alt, bases, hap, keep;
    std::vector<float> af;
    std::vector<int64_t> rstart;
    std::vector<uint32_t> order;
    std::vector<uint16_t> mmv, mmr;
    int64_t n_sites = 0;

    template <class Emit>
    void run(const isx_synth_params *p, const Contig &k, uint8_t *ref, Emit &&emit)
    {
        const int RL = p->read_len;
        const int64_t min_ins = 2 * (int64_t)RL;
        const uint32_t keep_thr = (uint32_t)std::min(65535.0, std::max(0.0, p->p_keep * 65536.0));
        Rng r(k.seed);
        for (int64_t i = 0; i < k.len; i += 32) {                  // 2 bits per base
            uint64_t x = r.next();
            const int64_t e = std::min<int64_t>(k.len, i + 32);
            for (int64_t j = i; j < e; j++, x >>= 2) ref[j] = (uint8_t)(x & 3);
        }
        n_sites = 0;
        if (!k.n_pairs) return;
        // variable sites
        site_of.assign((size_t)k.len, -1);
        const int64_t ns = (int64_t)((double)k.len * p->site_frac + r.uni());
        alt.clear(); af.clear();
        for (int64_t s = 0; s < ns; s++) {
            const int64_t pos = (int64_t)r.below((uint64_t)k.len);
            if (site_of[(size_t)pos] >= 0) continue;
            site_of[(size_t)pos] = (int32_t)alt.size();
            alt.push_back((uint8_t)((ref[pos] + 1 + r.below(3)) & 3));
            af.push_back((float)(p->af_lo + (p->af_hi - p->af_lo) * r.uni()));
        }
        n_sites = (int64_t)alt.size();
        // read starts (pair j: mates 2j, 2j+1), BAM order = by start
        const int64_t np = k.n_pairs, nr = 2 * np;
        rstart.resize((size_t)nr); hap.resize((size_t)np); mmv.assign((size_t)np, 0); mmr.assign((size_t)nr, 0);
        for (int64_t j = 0; j < np; j++) {
            int64_t ins = (int64_t)(p->insert_mean + p->insert_sd * r.normal());
            ins = std::min<int64_t>(std::max<int64_t>(ins, min_ins), k.len - 1);
            const int64_t s1 = (int64_t)r.below((uint64_t)(k.len - ins));
            rstart[(size_t)(2 * j)] = s1; rstart[(size_t)(2 * j + 1)] = s1 + ins - RL;
            hap[(size_t)j] = (uint8_t)(r.next() & 1);
        }
        order.resize((size_t)nr);
        for (int64_t i = 0; i < nr; i++) order[(size_t)i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rstart[a] < rstart[b]; });
        // read bases (in read-id order so mm is known before anything is emitted)
        bases.resize((size_t)nr * RL);
        int64_t next_err = p->err > 0 ? (int64_t)(std::log(1.0 - r.uni()) / std::log(1.0 - p->err)) : (int64_t)1 << 62;
        for (int64_t i = 0; i < nr; i++) {
            const int64_t st = rstart[(size_t)i];
            uint8_t *b = bases.data() + (size_t)i * RL;
            int mm = 0;
            for (int q = 0; q < RL; q++) {
                uint8_t v = ref[st + q];
                const int32_t si = site_of[(size_t)(st + q)];
                if (si >= 0 && r.uni() < (double)af[(size_t)si] * (hap[(size_t)(i >> 1)] ? 1.6 : 0.4)) v = alt[(size_t)si];
                if (next_err-- == 0) {
                    v = (uint8_t)r.below(4);
                    next_err = (int64_t)(std::log(1.0 - r.uni()) / std::log(1.0 - p->err));
                }
                mm += v != ref[st + q];
                b[q] = v;
            }
            mmr[(size_t)i] = (uint16_t)std::min(65535, mm);
            mmv[(size_t)(i >> 1)] = (uint16_t)std::min<int>(65535, mmv[(size_t)(i >> 1)] + mm);
        }
        // BAM order; a base is kept with probability p_keep (16 random bits each)
        keep.resize((size_t)RL);
        for (int64_t oi = 0; oi < nr; oi++) {
            const uint32_t i = order[(size_t)oi];
            uint64_t bits = 0;
            for (int q = 0; q < RL; q++) {
                if ((q & 3) == 0) bits = r.next();
                keep[(size_t)q] = (uint32_t)(bits & 0xFFFF) < keep_thr;
                bits >>= 16;
            }
            emit(i, rstart[i], bases.data() + (size_t)i * RL, keep.data(), mmv[i >> 1], mmr[i]);
        }
    }
};

// ---- BGZF / BAM output of the same reads (what the front end of the product then decodes) ----
void bgzf_append(std::string &out, const uint8_t *data, size_t n)
{
    for (size_t a = 0; a < n || (n == 0 && a == 0); a += 0xFF00) {
        const size_t len = n ? std::min<size_t>(0xFF00, n - a) : 0;
        uint8_t comp[0x10000 + 64];
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<Bytef *>(data + a); zs.avail_in = (uInt)len;
        zs.next_out = comp; zs.avail_out = sizeof comp;
        deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data + a, (uInt)len);
        const uint16_t bsize = (uint16_t)(clen + 25);
        const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xFF), (uint8_t)(bsize >> 8)};
        out.append(reinterpret_cast<const char *>(hdr), 18);
        out.append(reinterpret_cast<const char *>(comp), clen);
        const uint32_t isize = (uint32_t)len;
        out.append(reinterpret_cast<const char *>(&crc), 4);
        out.append(reinterpret_cast<const char *>(&isize), 4);
        if (n == 0) break;
    }
}

inline int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

template <class T> inline void put(std::string &s, T v) { s.append(reinterpret_cast<const char *>(&v), sizeof v); }

}  // namespace

extern "C" {

// generate the genomes genome_sel[0..n_sel) (indices of the plan), laid end to end in that order
int isx_synth_generate(const isx_synth_params *p, const int32_t *genome_sel, int32_t n_sel, const int64_t *length,
                       const double *coverage, isx_synth_out *out)
{
    if (!p || !genome_sel || n_sel <= 0 || !out) return -1;
    memset(out, 0, sizeof *out);
    const int RL = p->read_len, C = p->contigs;
    const int64_t n_sc = (int64_t)n_sel * C;
    std::vector<Contig> cs;
    int64_t obs_cap = 0, pair0 = 0;
    const int64_t off = plan_contigs(p, genome_sel, n_sel, length, coverage, cs, obs_cap, pair0);
    if (off >= (int64_t)0xFFFF0000ll || pair0 >= (int64_t)0xFFFFFFFFll) return -2;
    out->n_pos = off; out->n_pairs = pair0; out->n_scaffolds = n_sc;
    out->profiled_bases = pair0 * 2 * RL;
    out->ref = (uint8_t *)malloc((size_t)std::max<int64_t>(off, 1));
    out->obs = (synth_obs *)malloc((size_t)std::max<int64_t>(obs_cap, 1) * sizeof(synth_obs));
    out->pair = (uint32_t *)malloc((size_t)std::max<int64_t>(obs_cap, 1) * sizeof(uint32_t));
    out->scaffold_bounds = (int64_t *)malloc((size_t)(n_sc + 1) * sizeof(int64_t));
    out->scaffold_genome = (int32_t *)malloc((size_t)n_sc * sizeof(int32_t));
    if (!out->ref || !out->obs || !out->pair || !out->scaffold_bounds || !out->scaffold_genome) { isx_synth_free(out); return -3; }
    for (int64_t i = 0; i < n_sc; i++) { out->scaffold_bounds[i] = cs[(size_t)i].off; out->scaffold_genome[i] = cs[(size_t)i].genome; }
    out->scaffold_bounds[n_sc] = off;

    std::atomic<int64_t> next{0}, n_sites{0};
    auto work = [&]() {
        ContigGen G;
        for (;;) {
            const int64_t ci = next.fetch_add(1);
            if (ci >= n_sc) break;
            Contig &k = cs[(size_t)ci];
            synth_obs *o = out->obs + k.obs_off;
            uint32_t *pr = out->pair + k.obs_off;
            int64_t n = 0;
            G.run(p, k, out->ref + k.off, [&](uint32_t i, int64_t st, const uint8_t *b, const uint8_t *keep, uint16_t mm_pair, uint16_t) {
                const uint16_t mm = p->with_mm ? (uint16_t)std::min<int>(p->max_mm, mm_pair) : (uint16_t)0;
                const uint32_t pid = (uint32_t)(k.pair0 + (i >> 1));
                for (int q = 0; q < RL; q++) {
                    if (!keep[q]) continue;
                    o[n].gpos = (uint32_t)(k.off + st + q); o[n].mm = mm; o[n].base = b[q]; o[n].flags = 0;
                    pr[n] = pid;
                    n++;
                }
            });
            k.obs_n = n;
            n_sites.fetch_add(G.n_sites);
        }
    };
    const int nt = std::max(1, p->threads);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    // close the gaps between the contigs' blocks
    int64_t w = 0;
    for (auto &k : cs) {
        if (k.obs_n && w != k.obs_off) {
            memmove(out->obs + w, out->obs + k.obs_off, (size_t)k.obs_n * sizeof(synth_obs));
            memmove(out->pair + w, out->pair + k.obs_off, (size_t)k.obs_n * sizeof(uint32_t));
        }
        w += k.obs_n;
    }
    out->n_obs = w;
    out->n_sites = n_sites.load();
    return 0;
}

// The same genomes as a coordinate-sorted BAM: one @SQ per contig ("g<genome>_c<contig>", genome = index of the plan), the
// reads of isx_synth_generate record for record (2 x read_len, <read_len>M, proper pairs, quality 37 where the generator keeps a
// base and 12 where it drops it, NM = the read's mismatches), contig after contig.  Returns the number of reads written, < 0 on error.
// ref_out (may be NULL): [sum of contig lengths] base codes, the FASTA of the file.
int64_t isx_synth_write_bam(const isx_synth_params *p, const int32_t *genome_sel, int32_t n_sel, const int64_t *length,
                            const double *coverage, const char *path, uint8_t *ref_out, int64_t *n_pos_out, int64_t *n_pairs_out)
{
    if (!p || !genome_sel || n_sel <= 0 || !path) return -1;
    const int RL = p->read_len, C = p->contigs;
    const int64_t n_sc = (int64_t)n_sel * C;
    std::vector<Contig> cs;
    int64_t obs_cap = 0, pair0 = 0;
    const int64_t off = plan_contigs(p, genome_sel, n_sel, length, coverage, cs, obs_cap, pair0);
    if (n_pos_out) *n_pos_out = off;
    if (n_pairs_out) *n_pairs_out = pair0;
    std::vector<uint8_t> own_ref;
    if (!ref_out) { own_ref.resize((size_t)std::max<int64_t>(off, 1)); ref_out = own_ref.data(); }
    // header
    std::string head, text = "@HD\tVN:1.6\tSO:coordinate\n";
    char name[64];
    for (auto &k : cs) { snprintf(name, sizeof name, "g%05d_c%03d", k.plan_genome, k.contig); text += std::string("@SQ\tSN:") + name + "\tLN:" + std::to_string(k.len) + "\n"; }
    head.append("BAM\1", 4);
    put<int32_t>(head, (int32_t)text.size());
    head += text;
    put<int32_t>(head, (int32_t)n_sc);
    for (auto &k : cs) {
        snprintf(name, sizeof name, "g%05d_c%03d", k.plan_genome, k.contig);
        put<int32_t>(head, (int32_t)strlen(name) + 1);
        head.append(name, strlen(name) + 1);
        put<int32_t>(head, (int32_t)k.len);
    }
    std::vector<std::string> part((size_t)n_sc);
    std::atomic<int64_t> next{0}, n_reads{0};
    static const uint8_t NIB[4] = {1, 2, 8, 4};         // A C T G (P2C order) -> BAM 4-bit codes
    auto work = [&]() {
        ContigGen G;
        std::string raw;
        for (;;) {
            const int64_t ci = next.fetch_add(1);
            if (ci >= n_sc) break;
            const Contig &k = cs[(size_t)ci];
            raw.clear();
            raw.reserve((size_t)k.n_pairs * 2 * (size_t)(60 + RL + RL / 2));
            int64_t nr = 0;
            G.run(p, k, ref_out + k.off, [&](uint32_t i, int64_t st, const uint8_t *b, const uint8_t *keep, uint16_t, uint16_t mm_read) {
                const bool first = (i & 1) == 0;
                const int64_t mate = G.rstart[i ^ 1];
                const int64_t lo = std::min(st, mate), hi = std::max(st, mate) + RL;
                const int32_t tlen = (int32_t)(st <= mate ? hi - lo : -(hi - lo));
                char rn[48];
                const int l_name = snprintf(rn, sizeof rn, "g%dc%dp%lld", k.plan_genome, k.contig, (long long)(i >> 1)) + 1;
                const int32_t block = 32 + l_name + 4 + (RL + 1) / 2 + RL + 4;
                put<int32_t>(raw, block);
                put<int32_t>(raw, (int32_t)ci);
                put<int32_t>(raw, (int32_t)st);
                put<uint8_t>(raw, (uint8_t)l_name);
                put<uint8_t>(raw, 42);
                put<uint16_t>(raw, (uint16_t)reg2bin(st, st + RL));
                put<uint16_t>(raw, 1);
                put<uint16_t>(raw, (uint16_t)(0x1 | 0x2 | (first ? 0x40 | 0x20 : 0x80 | 0x10)));
                put<int32_t>(raw, RL);
                put<int32_t>(raw, (int32_t)ci);
                put<int32_t>(raw, (int32_t)mate);
                put<int32_t>(raw, tlen);
                raw.append(rn, (size_t)l_name);
                put<uint32_t>(raw, ((uint32_t)RL << 4) | 0u);
                for (int q = 0; q < RL; q += 2) put<uint8_t>(raw, (uint8_t)((NIB[b[q]] << 4) | (q + 1 < RL ? NIB[b[q + 1]] : 0)));
                for (int q = 0; q < RL; q++) put<uint8_t>(raw, keep[q] ? 37 : 12);
                raw.append("NMC", 3);
                put<uint8_t>(raw, (uint8_t)std::min<int>(255, mm_read));
                nr++;
            });
            part[(size_t)ci].swap(raw);            // uncompressed for now: the blocks of ALL contigs are compressed in parallel below
            n_reads.fetch_add(nr);
        }
    };
    const int nt = std::max(1, p->threads);
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    // BGZF: pieces of <= 8 blocks of one contig, compressed by all threads, written in file order
    struct Piece { size_t contig, off, len; };
    std::vector<Piece> pieces;
    const size_t PIECE = 8 * 0xFF00;
    for (size_t ci = 0; ci < part.size(); ci++)
        for (size_t a = 0; a < part[ci].size(); a += PIECE) pieces.push_back({ci, a, std::min(PIECE, part[ci].size() - a)});
    std::vector<std::string> comp(pieces.size());
    std::atomic<size_t> nextp{0};
    auto squeeze = [&]() {
        for (;;) {
            const size_t k = nextp.fetch_add(1);
            if (k >= pieces.size()) break;
            bgzf_append(comp[k], reinterpret_cast<const uint8_t *>(part[pieces[k].contig].data()) + pieces[k].off, pieces[k].len);
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(squeeze);
        squeeze();
        for (auto &t : th) t.join();
    }
    FILE *f = fopen(path, "wb");
    if (!f) return -4;
    std::string hz;
    bgzf_append(hz, reinterpret_cast<const uint8_t *>(head.data()), head.size());
    bool ok = fwrite(hz.data(), 1, hz.size(), f) == hz.size();
    for (auto &s : comp) if (ok && !s.empty()) ok = fwrite(s.data(), 1, s.size(), f) == s.size();
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_block, 1, 28, f) == 28;
    ok = fclose(f) == 0 && ok;
    return ok ? n_reads.load() : -5;
}

// The same genomes as READ SEGMENTS (include/instrain_amd.h isx_segs): one segment per read (read_len <= 150 columns, code 4
// where the generator drops a base), what the read-level hand-over ships -- no observation stream in between.
// Two calls: with seg_gpos == NULL only the sizes (out->n_obs = kept bases, return value = segments), then the arrays.
int64_t isx_synth_generate_segs(const isx_synth_params *p, const int32_t *genome_sel, int32_t n_sel, const int64_t *length,
                                const double *coverage, isx_synth_out *out, uint32_t *seg_gpos, uint8_t *seg_len, uint8_t *seg_mm,
                                uint32_t *seg_pair, uint32_t *seg_bases)
{
    if (!p || !genome_sel || n_sel <= 0 || !out || p->read_len > 150) return -1;
    memset(out, 0, sizeof *out);
    const int RL = p->read_len, C = p->contigs;
    const int64_t n_sc = (int64_t)n_sel * C;
    std::vector<Contig> cs;
    int64_t obs_cap = 0, pair0 = 0;
    const int64_t off = plan_contigs(p, genome_sel, n_sel, length, coverage, cs, obs_cap, pair0);
    if (off >= (int64_t)0xFFFF0000ll || pair0 >= (int64_t)0xFFFFFFFFll) return -2;
    out->n_pos = off; out->n_pairs = pair0; out->n_scaffolds = n_sc;
    out->profiled_bases = pair0 * 2 * RL;
    if (!seg_gpos) return 2 * pair0;
    out->ref = (uint8_t *)malloc((size_t)std::max<int64_t>(off, 1));
    out->scaffold_bounds = (int64_t *)malloc((size_t)(n_sc + 1) * sizeof(int64_t));
    out->scaffold_genome = (int32_t *)malloc((size_t)n_sc * sizeof(int32_t));
    if (!out->ref || !out->scaffold_bounds || !out->scaffold_genome) { isx_synth_free(out); return -3; }
    for (int64_t i = 0; i < n_sc; i++) { out->scaffold_bounds[i] = cs[(size_t)i].off; out->scaffold_genome[i] = cs[(size_t)i].genome; }
    out->scaffold_bounds[n_sc] = off;
    std::atomic<int64_t> next{0}, n_sites{0}, n_kept{0};
    auto work = [&]() {
        ContigGen G;
        for (;;) {
            const int64_t ci = next.fetch_add(1);
            if (ci >= n_sc) break;
            const Contig &k = cs[(size_t)ci];
            int64_t at = 2 * k.pair0, kept = 0;             // a contig's reads are consecutive segments
            G.run(p, k, out->ref + k.off, [&](uint32_t i, int64_t st, const uint8_t *b, const uint8_t *keep, uint16_t mm_pair, uint16_t) {
                uint32_t *w = seg_bases + (size_t)at * 15;
                for (int q = 0; q < 15; q++) w[q] = 0x24924924u;
                for (int q = 0; q < RL; q++) {
                    if (!keep[q]) continue;
                    w[q / 10] = (w[q / 10] & ~(7u << (3 * (q % 10)))) | ((uint32_t)b[q] << (3 * (q % 10)));
                    kept++;
                }
                seg_gpos[at] = (uint32_t)(k.off + st); seg_len[at] = (uint8_t)RL;
                if (seg_mm) seg_mm[at] = p->with_mm ? (uint8_t)std::min<int>(std::min(255, p->max_mm), mm_pair) : (uint8_t)0;
                if (seg_pair) seg_pair[at] = (uint32_t)(k.pair0 + (i >> 1));
                at++;
            });
            n_sites.fetch_add(G.n_sites); n_kept.fetch_add(kept);
        }
    };
    const int nt = std::max(1, p->threads);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    out->n_obs = n_kept.load();
    out->n_sites = n_sites.load();
    return 2 * pair0;
}

// a distinct batch of the same shape from a batch of segments: every start moves up by `shift`, every A/C/T/G code is
// rotated by `rot` (reference and reads alike) -- what synth.shifted_variant does to observation records
void isx_synth_shift_segs(const uint32_t *gpos_in, const uint32_t *bases_in, int64_t n_seg, uint32_t shift, int32_t rot, uint32_t *gpos_out,
                          uint32_t *bases_out, int32_t threads)
{
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t a = next.fetch_add(8192);
            if (a >= n_seg) break;
            const int64_t e = std::min<int64_t>(n_seg, a + 8192);
            for (int64_t i = a; i < e; i++) gpos_out[i] = gpos_in[i] + shift;
            for (int64_t i = a * 15; i < e * 15; i++) {
                const uint32_t w = bases_in[i];
                // codes < 4 (bit 2 clear): (c + rot) & 3 on the low two bits of every 3-bit field, no carry into bit 2
                const uint32_t acgt = ~w & 0x24924924u;                     // bit 2 of every field set where the code is A/C/T/G
                const uint32_t sel = (acgt >> 2) * 3u;                      // 0b011 in those fields
                const uint32_t lo = w & 0x1B6DB6DBu;                        // low two bits of every field
                const uint32_t sum = (lo + (uint32_t)rot * 0x09249249u) & 0x1B6DB6DBu;   // per-field add; a carry lands in bit 2 and is masked off
                bases_out[i] = (w & ~sel) | (sum & sel);
            }
        }
    };
    const int nt = std::max(1, threads);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// ---- observation stream -> read segments (include/instrain_amd.h isx_segs) ----
// What the read-level hand-over ships for the same workload: consecutive observations of one read pair / mm level at
// ascending positions less than 150 columns from the first become one segment (missing columns = code 4).  Any stream is
// legal input (a segment only ever grows by the observation that follows it, so the segments keep the arrival order); a
// read-major stream -- what the generator above and the BAM front end emit -- gives one segment per read.
// Two calls: n_seg first (seg_gpos == NULL), then the arrays.  base >= 4 (a non-ACGT base) becomes code 5.
int64_t isx_synth_obs_to_segs(const synth_obs *obs, const uint32_t *pair, int64_t n_obs, uint32_t *seg_gpos, uint8_t *seg_len,
                              uint8_t *seg_mm, uint32_t *seg_pair, uint32_t *seg_bases, int32_t threads)
{
    auto starts_new = [&](int64_t i, uint32_t seg_start) {
        if (i == 0) return true;
        if (pair && pair[i] != pair[i - 1]) return true;
        if (obs[i].mm != obs[i - 1].mm) return true;
        if (obs[i].gpos <= obs[i - 1].gpos) return true;
        return obs[i].gpos - seg_start >= 150u;
    };
    // pass 1 (sequential: a segment's start decides where the next one begins): first observation of every segment
    std::vector<int64_t> first;
    first.reserve((size_t)(n_obs / 100 + 16));
    uint32_t st = 0;
    for (int64_t i = 0; i < n_obs; i++)
        if (starts_new(i, st)) { st = obs[i].gpos; first.push_back(i); }
    const int64_t n_seg = (int64_t)first.size();
    if (!seg_gpos) return n_seg;
    first.push_back(n_obs);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t s0 = next.fetch_add(4096);
            if (s0 >= n_seg) break;
            const int64_t s1 = std::min<int64_t>(n_seg, s0 + 4096);
            for (int64_t sgi = s0; sgi < s1; sgi++) {
                const int64_t a = first[(size_t)sgi], e = first[(size_t)sgi + 1];
                const uint32_t g0 = obs[a].gpos;
                uint32_t *w = seg_bases + (size_t)sgi * 15;
                for (int k = 0; k < 15; k++) w[k] = 0x24924924u;
                for (int64_t i = a; i < e; i++) {
                    const uint32_t j = obs[i].gpos - g0, code = obs[i].base < 4 ? obs[i].base : 5u;
                    w[j / 10] = (w[j / 10] & ~(7u << (3 * (j % 10)))) | (code << (3 * (j % 10)));
                }
                seg_gpos[sgi] = g0;
                seg_len[sgi] = (uint8_t)(obs[e - 1].gpos - g0 + 1);
                if (seg_mm) seg_mm[sgi] = (uint8_t)std::min<uint32_t>(255u, obs[a].mm);
                if (seg_pair) seg_pair[sgi] = pair ? pair[a] : 0u;
            }
        }
    };
    const int nt = std::max(1, threads);
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return n_seg;
}

}  // extern "C"
