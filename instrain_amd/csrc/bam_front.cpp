// bam_front.cpp -- host side of the hot path: BGZF/BAM decode, read-pair filter and the
// htslib-1.9 pileup rules, producing the packed observation stream the kernels consume.
//
// Replaces (host side, C++; the reference reaches all of this through pysam -> htslib, which
// is not vendored under /root/reference):
//   pysam.AlignmentFile(bam) / samfile.fetch      /root/reference/inStrain/profile/profile_utilities.py:56
//   get_paired_reads                              /root/reference/inStrain/filter_reads.py:885-956
//   paired_read_filter ('paired_only')            filter_reads.py:471-532
//   filter_scaff2pair2info / evaluate_pair        filter_reads.py:201-260, 388-426
//   samfile.pileup(..., stepper='nofilter', ignore_overlaps=True, min_base_quality=30, ...)
//                                                 profile_utilities.py:150-153
//       = htslib 1.9 bam_plp: default flag mask UNMAP|SECONDARY|QCFAIL|DUP, overlap_push /
//         tweak_overlap_quality / cigar_iref2iseq_set/next (sam.c), and pysam's
//         `qual >= min_base_quality` test when listing PileupColumn.pileups
//   iterate_splits                                /root/reference/inStrain/profile/fasta.py:56-73
//
// No device code here; it is linked into libinstrain_amd.so so that the whole path sits behind
// one C ABI.  Inflate of the BGZF blocks is multi-threaded (blocks are independent).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <chrono>
#include <cstdio>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../../include/instrain_amd.h"

void isx_set_error(const std::string &msg);

namespace {

enum { CM = 0, CI, CD, CN, CS, CH, CP, CEQ, CX };
constexpr uint16_t FPROPER = 0x2, FUNMAP = 0x4, FMUNMAP = 0x8, FSECONDARY = 0x100, FQCFAIL = 0x200, FDUP = 0x400;
constexpr uint16_t DEF_MASK = FUNMAP | FSECONDARY | FQCFAIL | FDUP;

// 4-bit BAM code -> inStrain base index (A,C,T,G = 0..3; everything else 4)
const uint8_t CODE2IDX[16] = {4, 0, 1, 4, 3, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4};

struct Read {
    int32_t tid, pos, isize, l_seq, nm;
    uint16_t flag, n_cigar;
    uint8_t mapq;
    bool has_nm;
    uint32_t name_off, name_len;
    uint64_t cigar_off, seq_off, qual_off;
};

struct Cursor {     // htslib sam.c cigar_iref2iseq_* state
    const uint32_t *cig; int n; int k, icig, iseq, iref;
};

int cur_set(Cursor &c, int pos)
{
    if (pos < 0) return -1;
    c.k = 0; c.icig = 0; c.iseq = 0; c.iref = 0;
    while (c.k < c.n) {
        const int op = c.cig[c.k] & 15, n = (int)(c.cig[c.k] >> 4);
        if (op == CS) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CH || op == CP) { c.k++; c.icig = 0; continue; }
        if (op == CM || op == CEQ || op == CX) {
            pos -= n;
            if (pos < 0) { c.icig = n + pos; c.iseq += c.icig; c.iref += c.icig; return 0; }
            c.k++; c.iseq += n; c.icig = 0; c.iref += n;
            continue;
        }
        if (op == CI) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CD || op == CN) {
            pos -= n;
            if (pos < 0) pos = 0;
            c.k++; c.icig = 0; c.iref += n;
            continue;
        }
        return -2;
    }
    c.iseq = -1;
    return -1;
}

int cur_next(Cursor &c)
{
    while (c.k < c.n) {
        const int op = c.cig[c.k] & 15, n = (int)(c.cig[c.k] >> 4);
        if (op == CM || op == CEQ || op == CX) {
            if (c.icig >= n - 1) { c.icig = 0; c.k++; continue; }
            c.iseq++; c.icig++; c.iref++;
            return 0;
        }
        if (op == CD || op == CN) { c.k++; c.iref += n; c.icig = 0; continue; }
        if (op == CI || op == CS) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CH || op == CP) { c.k++; c.icig = 0; continue; }
        return -2;
    }
    c.iseq = -1; c.iref = -1;
    return -1;
}

struct PairInfo {       // filter_reads.py i2o order
    int64_t nm, insert, mapq, length, reads, start, stop;
    bool pass;
    uint32_t pair_id;
};

}  // namespace

template <class T>
struct RawBuf {                 // sized once, written once: no value initialisation (vector::resize would memset)
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void resize(size_t m) { p.reset(new T[std::max<size_t>(m, 1)]); n = m; }
    T *data() { return p.get(); }
    const T *data() const { return p.get(); }
    const T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};

struct isx_bam {
    std::vector<std::string> ref_name;
    std::vector<int64_t> ref_len, ref_off;
    std::vector<Read> reads;
    RawBuf<char> names;
    RawBuf<uint32_t> cigars;
    RawBuf<uint8_t> seqs;           // one code per base (unpacked)
    RawBuf<uint8_t> quals;          // mutated by overlap resolution
    // results of expand (plain arrays: no zero fill of what is written once)
    std::unique_ptr<isx_obs[]> obs;
    std::unique_ptr<uint32_t[]> pair;
    size_t n_obs = 0;
    std::vector<int64_t> split_bounds;
    std::vector<int32_t> split_ref;
    bool expanded = false;
};

namespace {

bool inflate_block(const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst)
{
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = (uInt)n_src;
    zs.next_out = dst; zs.avail_out = (uInt)n_dst;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && zs.total_out == n_dst;
    inflateEnd(&zs);
    return ok;
}

int load_file(const char *path, RawBuf<uint8_t> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) { isx_set_error(std::string("cannot open ") + path); return ISX_ERR_IO; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    RawBuf<uint8_t> raw;
    raw.resize((size_t)n);
    if (n && fread(raw.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); isx_set_error("short read"); return ISX_ERR_IO; }
    fclose(f);
    // index the BGZF blocks
    struct Blk { size_t src, n_src, dst, n_dst; };
    std::vector<Blk> blks;
    size_t off = 0, total = 0;
    while (off + 18 <= raw.size()) {
        const uint8_t *h = raw.data() + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { isx_set_error("not a BGZF file"); return ISX_ERR_IO; }
        const size_t xlen = h[10] | (h[11] << 8);
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + xlen;) {
            const size_t slen = h[x + 2] | (h[x + 3] << 8);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || off + bsize > raw.size()) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        const uint8_t *t = h + bsize - 4;
        const size_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((size_t)t[3] << 24);
        blks.push_back({off + 12 + xlen, bsize - 12 - xlen - 8, total, isize});
        total += isize;
        off += bsize;
    }
    out.resize(total);
    const unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    std::vector<int> ok(nt, 1);
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
            for (size_t i = t; i < blks.size(); i += nt)
                if (blks[i].n_dst && !inflate_block(raw.data() + blks[i].src, blks[i].n_src, out.data() + blks[i].dst, blks[i].n_dst))
                    ok[t] = 0;
        });
    for (auto &x : th) x.join();
    for (int v : ok) if (!v) { isx_set_error("BGZF inflate failed"); return ISX_ERR_IO; }
    return ISX_OK;
}

inline int32_t rd32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }

int parse_nm(const uint8_t *p, const uint8_t *end, bool &has, int32_t &nm)
{
    has = false;
    while (p + 3 <= end) {
        const bool is_nm = (p[0] == 'N' && p[1] == 'M');
        const char t = (char)p[2];
        p += 3;
        int64_t v = 0;
        bool num = true;
        switch (t) {
        case 'A': v = *p; p += 1; num = false; break;
        case 'c': v = (int8_t)*p; p += 1; break;
        case 'C': v = *p; p += 1; break;
        case 's': { int16_t x; memcpy(&x, p, 2); v = x; p += 2; break; }
        case 'S': { uint16_t x; memcpy(&x, p, 2); v = x; p += 2; break; }
        case 'i': { int32_t x; memcpy(&x, p, 4); v = x; p += 4; break; }
        case 'I': { uint32_t x; memcpy(&x, p, 4); v = x; p += 4; break; }
        case 'f': p += 4; num = false; break;
        case 'Z': case 'H': while (p < end && *p) p++; p++; num = false; break;
        case 'B': {
            const char sub = (char)p[0];
            const int32_t cnt = rd32(p + 1);
            const int sz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            p += 5 + (size_t)cnt * sz;
            num = false;
            break;
        }
        default: return -1;
        }
        if (is_nm && num) { has = true; nm = (int32_t)v; }
    }
    return 0;
}

int parse_bam(const RawBuf<uint8_t> &buf, isx_bam &B)
{
    if (buf.size() < 12 || memcmp(buf.data(), "BAM\1", 4) != 0) { isx_set_error("not a BAM file"); return ISX_ERR_IO; }
    size_t off = 8 + (size_t)rd32(buf.data() + 4);
    const int32_t n_ref = rd32(buf.data() + off);
    off += 4;
    int64_t flat = 0;
    for (int i = 0; i < n_ref; i++) {
        const int32_t l_name = rd32(buf.data() + off);
        B.ref_name.emplace_back(reinterpret_cast<const char *>(buf.data() + off + 4), (size_t)l_name - 1);
        const int32_t l_ref = rd32(buf.data() + off + 4 + l_name);
        B.ref_len.push_back(l_ref);
        B.ref_off.push_back(flat);
        flat += l_ref;
        off += 8 + (size_t)l_name;
    }
    // pass 1 (sequential, one walk, a few words per record): record boundaries and where each record's
    // name / CIGAR / bases go in the side arrays
    const bool ptiming = getenv("ISX_BAM_TIMING") != nullptr;
    const auto pt0 = std::chrono::steady_clock::now();
    std::vector<size_t> rec_off, name_at, cig_at, seq_at;
    {
        const size_t guess = (buf.size() - off) / 160 + 16;        // a 2 x 150 bp record is ~ 240 bytes
        rec_off.reserve(guess); name_at.reserve(guess); cig_at.reserve(guess); seq_at.reserve(guess);
    }
    size_t n_names = 0, n_cig = 0, n_seq = 0;
    while (off + 36 <= buf.size()) {
        const uint8_t *p = buf.data() + off;
        const int32_t block = rd32(p);
        if (block < 32 || off + 4 + (size_t)block > buf.size()) { isx_set_error("truncated BAM record"); return ISX_ERR_IO; }
        uint16_t ncig;
        memcpy(&ncig, p + 16, 2);
        const int32_t l_seq = rd32(p + 20);
        const size_t need = 32 + (size_t)p[12] + (size_t)ncig * 4 + ((size_t)std::max(l_seq, 0) + 1) / 2 + (size_t)std::max(l_seq, 0);
        if (l_seq < 0 || p[12] == 0 || need > (size_t)block) { isx_set_error("corrupt BAM record"); return ISX_ERR_IO; }
        rec_off.push_back(off); name_at.push_back(n_names); cig_at.push_back(n_cig); seq_at.push_back(n_seq);
        n_names += (size_t)p[12] - 1; n_cig += ncig; n_seq += (size_t)l_seq;
        off += 4 + (size_t)block;
    }
    const size_t n = rec_off.size();
    B.reads.resize(n);
    const auto pt1 = std::chrono::steady_clock::now();
    B.names.resize(n_names); B.cigars.resize(n_cig); B.seqs.resize(n_seq); B.quals.resize(n_seq);
    const auto pt2 = std::chrono::steady_clock::now();
    // pass 2 (threads over record ranges): field extraction, nibble unpack, aux walk for NM
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(64u, std::max(1u, std::thread::hardware_concurrency())), n / 4096 + 1));
    std::vector<int> bad(nt, 0);
    auto work = [&](unsigned t) {
        const size_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        for (size_t i = i0; i < i1; i++) {
            const uint8_t *p = buf.data() + rec_off[i];
            const int32_t block = rd32(p);
            Read r{};
            r.tid = rd32(p + 4); r.pos = rd32(p + 8);
            const uint8_t l_name = p[12];
            r.mapq = p[13];
            uint16_t ncig, flag;
            memcpy(&ncig, p + 16, 2); memcpy(&flag, p + 18, 2);
            r.n_cigar = ncig; r.flag = flag;
            r.l_seq = rd32(p + 20);
            r.isize = rd32(p + 32);
            const uint8_t *q = p + 36;
            r.name_off = (uint32_t)name_at[i]; r.name_len = (uint32_t)l_name - 1;
            memcpy(B.names.data() + name_at[i], q, (size_t)l_name - 1);
            q += l_name;
            r.cigar_off = cig_at[i];
            memcpy(B.cigars.data() + cig_at[i], q, (size_t)ncig * 4);
            q += (size_t)ncig * 4;
            r.seq_off = seq_at[i];
            uint8_t *sq = B.seqs.data() + seq_at[i];
            for (int32_t k = 0; k < r.l_seq; k++) {
                const uint8_t byte = q[k >> 1];
                sq[k] = (k & 1) ? (byte & 15) : (byte >> 4);
            }
            q += ((size_t)r.l_seq + 1) / 2;
            r.qual_off = seq_at[i];
            memcpy(B.quals.data() + seq_at[i], q, (size_t)r.l_seq);
            q += r.l_seq;
            if (parse_nm(q, p + 4 + block, r.has_nm, r.nm) != 0) bad[t] = 1;
            B.reads[i] = r;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    if (ptiming) fprintf(stderr, "[parse_bam] sizes %.1f ms, alloc %.1f ms, fill %.1f ms (%u threads)\n", std::chrono::duration<double, std::milli>(pt1 - pt0).count(), std::chrono::duration<double, std::milli>(pt2 - pt1).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pt2).count(), nt);
    for (int v : bad) if (v) { isx_set_error("bad aux field"); return ISX_ERR_IO; }
    if (n_names >= 0xFFFFFFFFull) { isx_set_error("read names exceed 4 GiB"); return ISX_ERR_IO; }
    return ISX_OK;
}

// htslib sam.c tweak_overlap_quality
void tweak_overlap(isx_bam &B, const Read &a, const Read &b)
{
    Cursor ca{B.cigars.data() + a.cigar_off, a.n_cigar, 0, 0, 0, 0};
    Cursor cb{B.cigars.data() + b.cigar_off, b.n_cigar, 0, 0, 0, 0};
    uint8_t *aq = B.quals.data() + a.qual_off, *bq = B.quals.data() + b.qual_off;
    const uint8_t *as = B.seqs.data() + a.seq_off, *bs = B.seqs.data() + b.seq_off;
    int iref = b.pos;
    int a_ret = cur_set(ca, iref - a.pos);
    if (a_ret < 0) return;
    int b_ret = cur_set(cb, iref - b.pos);
    if (b_ret < 0) return;
    for (;;) {
        while (ca.iref >= 0 && ca.iref < iref - a.pos) a_ret = cur_next(ca);
        if (a_ret < 0) break;
        if (iref < ca.iref + a.pos) iref = ca.iref + a.pos;
        while (cb.iref >= 0 && cb.iref < iref - b.pos) b_ret = cur_next(cb);
        if (b_ret < 0) break;
        if (iref < cb.iref + b.pos) iref = cb.iref + b.pos;
        iref++;
        if (ca.iref + a.pos != cb.iref + b.pos) continue;
        const int qa = aq[ca.iseq], qb = bq[cb.iseq];
        if (as[ca.iseq] == bs[cb.iseq]) {
            const int q = qa + qb;
            aq[ca.iseq] = (uint8_t)(q > 200 ? 200 : q);
            bq[cb.iseq] = 0;
        } else if (qa >= qb) {
            aq[ca.iseq] = (uint8_t)(0.8 * qa);
            bq[cb.iseq] = 0;
        } else {
            bq[cb.iseq] = (uint8_t)(0.8 * qb);
            aq[ca.iseq] = 0;
        }
    }
}

struct RefSpan { int64_t first, last, end; int64_t qlen; bool any; };

RefSpan span_of(const isx_bam &B, const Read &r)
{
    RefSpan s{0, 0, r.pos, 0, false};
    int64_t ref = r.pos;
    for (int k = 0; k < r.n_cigar; k++) {
        const uint32_t c = B.cigars[r.cigar_off + k];
        const int op = c & 15;
        const int64_t n = c >> 4;
        if (op == CM || op == CEQ || op == CX) {
            if (n > 0) {
                if (!s.any) { s.first = ref; s.any = true; }
                s.last = ref + n - 1;
            }
            ref += n;
        } else if (op == CD || op == CN) ref += n;
        if (op == CM || op == CI || op == CS || op == CEQ || op == CX) s.qlen += n;
    }
    s.end = ref;
    return s;
}

}  // namespace

extern "C" {

int isx_bam_open(const char *path, isx_bam **out)
{
    if (!path || !out) { isx_set_error("isx_bam_open: bad argument"); return ISX_ERR_ARG; }
    *out = nullptr;
    RawBuf<uint8_t> buf;
    const bool timing = getenv("ISX_BAM_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = load_file(path, buf);
    if (rc != ISX_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    isx_bam *B = new isx_bam();
    rc = parse_bam(buf, *B);
    if (timing) fprintf(stderr, "[isx_bam_open] read + inflate %.1f ms, record extraction %.1f ms\n",
                        std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    if (rc != ISX_OK) { delete B; return rc; }
    *out = B;
    return ISX_OK;
}

void isx_bam_close(isx_bam *bam) { delete bam; }

int isx_bam_ref(const isx_bam *bam, int32_t i, const char **name, int64_t *length, int64_t *flat_offset)
{
    if (!bam || i < 0 || (size_t)i >= bam->ref_name.size()) { isx_set_error("isx_bam_ref: bad index"); return ISX_ERR_ARG; }
    if (name) *name = bam->ref_name[(size_t)i].c_str();
    if (length) *length = bam->ref_len[(size_t)i];
    if (flat_offset) *flat_offset = bam->ref_off[(size_t)i];
    return ISX_OK;
}

int isx_bam_expand(isx_bam *bam, const isx_bam_params *p, isx_bam_info *info)
{
    if (!bam || !p || !info) { isx_set_error("isx_bam_expand: bad argument"); return ISX_ERR_ARG; }
    if (bam->expanded) { isx_set_error("isx_bam_expand: already expanded (qualities were rewritten)"); return ISX_ERR_STATE; }
    isx_bam &B = *bam;
    memset(info, 0, sizeof(*info));
    const size_t n_ref = B.ref_name.size();
    const bool timing = getenv("ISX_BAM_TIMING") != nullptr;       // tuning aid: stage times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[isx_bam_expand] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    auto name_of = [&](const Read &r) { return std::string_view(B.names.data() + r.name_off, r.name_len); };

    // ---- get_paired_reads per scaffold (filter_reads.py:885-956) ----
    // The (scaffold, read name) -> pair table is hash-partitioned over a few threads: every thread walks all
    // reads in file order (so a pair's reads meet in file order) but owns only the names that hash to it.
    struct NameKey {
        int32_t tid; std::string_view name;
        bool operator==(const NameKey &o) const { return tid == o.tid && name == o.name; }
    };
    struct NameKeyHash {
        size_t operator()(const NameKey &k) const { return std::hash<std::string_view>()(k.name) * 1000003u ^ (size_t)(uint32_t)k.tid; }
    };
    const size_t n_reads_all = B.reads.size();
    const unsigned NT = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), n_reads_all / 8192 + 1));
    std::vector<std::unordered_map<NameKey, uint32_t, NameKeyHash>> maps(NT);      // value: index local to the partition
    std::vector<std::vector<PairInfo>> part_info(NT);
    std::vector<int32_t> read_pi(n_reads_all, -1);         // read -> index of its (scaffold, name) entry
    std::vector<RefSpan> spans(n_reads_all);
    std::vector<uint8_t> part_of(n_reads_all, 0xFF);       // partition of the read's name (0xFF: not in the table)
    std::vector<std::string> no_nm(NT);
    auto run_nt = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < NT; t++) th.emplace_back(fn, t);
        fn(0u);
        for (auto &x : th) x.join();
    };
    run_nt([&](unsigned t) {                                // spans + partition of every read (read ranges)
        for (size_t ri = n_reads_all * t / NT; ri < n_reads_all * (t + 1) / NT; ri++) {
            const Read &r = B.reads[ri];
            if (r.tid < 0 || (size_t)r.tid >= n_ref) continue;
            spans[ri] = span_of(B, r);
            if ((r.flag & FUNMAP) || !spans[ri].any) continue;          // get_reference_positions() == []
            part_of[ri] = (uint8_t)(NameKeyHash()(NameKey{r.tid, name_of(r)}) % NT);
        }
    });
    run_nt([&](unsigned t) {                                // the tables (name partitions)
        auto &m = maps[t];
        auto &pi = part_info[t];
        m.reserve(n_reads_all / (2 * NT) + 16);
        pi.reserve(n_reads_all / (2 * NT) + 16);
        for (size_t ri = 0; ri < n_reads_all; ri++) {
            if (part_of[ri] != t) continue;
            const Read &r = B.reads[ri];
            const RefSpan &s = spans[ri];
            if (!r.has_nm) { if (no_nm[t].empty()) no_nm[t] = std::string(name_of(r)); continue; }
            const NameKey key{r.tid, name_of(r)};
            auto it = m.find(key);
            if (it == m.end()) {
                m.emplace(key, (uint32_t)pi.size());
                read_pi[ri] = (int32_t)pi.size();
                pi.push_back(PairInfo{r.nm, -1, r.mapq, s.qlen, 1, s.first, s.last, false, 0});
            } else {
                PairInfo &i = pi[it->second];
                read_pi[ri] = (int32_t)it->second;
                i.nm += r.nm;
                i.reads += 1;
                i.length += s.qlen;
                i.mapq = std::max<int64_t>(i.mapq, r.mapq);
                if (i.reads == 2) {
                    if (s.last > i.start) i.insert = s.last - i.start;
                    else i.insert = i.stop - s.first;
                } else {
                    i.insert = -1;
                }
                i.start = 0; i.stop = 0;
            }
        }
    });
    for (unsigned t = 0; t < NT; t++)
        if (!no_nm[t].empty()) { isx_set_error("read without NM tag: " + no_nm[t]); return ISX_ERR_IO; }
    // one table: partition t's entries start at part_base[t]
    std::vector<uint32_t> part_base(NT + 1, 0);
    for (unsigned t = 0; t < NT; t++) part_base[t + 1] = part_base[t] + (uint32_t)part_info[t].size();
    std::vector<PairInfo> pinfo;
    pinfo.reserve(part_base[NT] + 16);
    for (unsigned t = 0; t < NT; t++) { pinfo.insert(pinfo.end(), part_info[t].begin(), part_info[t].end()); std::vector<PairInfo>().swap(part_info[t]); }
    run_nt([&](unsigned t) {
        for (size_t ri = n_reads_all * t / NT; ri < n_reads_all * (t + 1) / NT; ri++)
            if (read_pi[ri] >= 0) read_pi[ri] += (int32_t)part_base[part_of[ri]];
    });
    stage("pair table (by name)");
    // ---- paired_only + filter_scaff2pair2info (filter_reads.py:201-260, 471-532) ----
    std::vector<int64_t> ins;
    for (const PairInfo &i : pinfo) if (i.reads == 2) { ins.push_back(i.insert); info->unfiltered_pairs++; }
    double median = NAN;
    if (!ins.empty()) {                             // np.median (selection, not a full sort)
        const size_t n = ins.size();
        std::nth_element(ins.begin(), ins.begin() + n / 2, ins.end());
        const double hi = (double)ins[n / 2];
        median = (n & 1) ? hi : ((double)*std::max_element(ins.begin(), ins.begin() + n / 2) + hi) / 2.0;
    }
    info->median_insert = median;
    const double max_insert = median * p->max_insert_relative;
    int32_t max_mm = 0;
    for (PairInfo &i : pinfo) {
        if (i.reads != 2) continue;                 // pairing_filter == paired_only
        const double pid = 1 - ((double)i.nm / (double)i.length);       // evaluate_pair :406
        bool ok = pid > p->min_read_ani;
        ok = ok && (i.mapq > p->min_mapq);
        if (i.insert != -1) ok = ok && ((double)i.insert > (double)p->min_insert) && ((double)i.insert < max_insert);
        i.pass = ok;
        if (ok) {
            info->filtered_pairs++;
            info->filtered_bases += i.length;
            if (i.nm > max_mm) max_mm = (int32_t)i.nm;
        }
    }
    info->max_mm = p->skip_mm ? 0 : max_mm;
    if (max_mm > 65535) { isx_set_error("mm level > 65535"); return ISX_ERR_ARG; }

    stage("pair filter");
    // ---- overlap_push in file order (htslib-1.9 rule |isize| < 2*l_qseq) ----
    {
        std::vector<int64_t> pending(pinfo.size(), -1);     // per (scaffold, name): read waiting for its mate
        for (size_t ri = 0; ri < B.reads.size(); ri++) {
            const Read &r = B.reads[ri];
            if (r.tid < 0 || (size_t)r.tid >= n_ref || (r.flag & DEF_MASK)) continue;
            if ((r.flag & FMUNMAP) || !(r.flag & FPROPER)) continue;
            if (std::abs((int64_t)r.isize) >= 2 * (int64_t)r.l_seq) continue;
            int32_t pi = read_pi[ri];
            if (pi < 0) {                                   // aligned-base-free read: still hashed by name in htslib
                const NameKey key{r.tid, name_of(r)};
                const unsigned t = (unsigned)(NameKeyHash()(key) % NT);
                auto &m = maps[t];
                auto it = m.find(key);
                if (it == m.end()) {                        // new entries live behind the partitions: global index as value
                    m.emplace(key, 0x80000000u | (uint32_t)pinfo.size());
                    pi = (int32_t)pinfo.size();
                    pinfo.push_back(PairInfo{0, -1, 0, 0, 0, 0, 0, false, 0}); pending.push_back(-1);
                } else pi = (it->second & 0x80000000u) ? (int32_t)(it->second & 0x7FFFFFFFu) : (int32_t)(part_base[t] + it->second);
            }
            int64_t &slot = pending[(size_t)pi];
            if (slot >= 0 && spans[(size_t)slot].end <= r.pos) slot = -1;    // earlier read already left the buffer
            if (slot < 0) slot = (int64_t)ri;
            else {
                const size_t ai = (size_t)slot;
                slot = -1;
                tweak_overlap(B, B.reads[ai], r);
            }
        }
    }

    stage("overlap resolution");
    // ---- expansion: the visits on which get_base_counts_mm touches `table` ----
    // pair ids in order of first appearance (serial, one word per read); then per read the number of
    // visits it contributes (threads), a prefix sum, and the writes (threads) -- file order is kept
    uint32_t next_pair = 0;
    for (PairInfo &i : pinfo) i.pair_id = 0xFFFFFFFFu;
    const size_t n_reads = B.reads.size();
    std::vector<uint8_t> emit(n_reads, 0);
    for (size_t ri = 0; ri < n_reads; ri++) {
        const Read &r = B.reads[ri];
        if (r.tid < 0 || (size_t)r.tid >= n_ref || (r.flag & DEF_MASK)) continue;
        // R2M membership is by NAME on this scaffold (get_base_counts_mm looks up query_name)
        const int32_t pidx = read_pi[ri];
        if (pidx < 0) continue;
        PairInfo &pi = pinfo[(size_t)pidx];
        if (!pi.pass) continue;
        if (pi.pair_id == 0xFFFFFFFFu) pi.pair_id = next_pair++;
        emit[ri] = 1;
    }
    const uint8_t minq = (uint8_t)std::min(255, std::max(0, p->min_base_quality));
    std::vector<uint64_t> out_at(n_reads + 1, 0);
    // one walk of a read's CIGAR; WRITE = false only counts
    auto walk = [&](size_t ri, bool write, isx_obs *po, uint32_t *pp) -> uint64_t {
        const Read &r = B.reads[ri];
        const PairInfo &pi = pinfo[(size_t)read_pi[ri]];
        const uint16_t mm = p->skip_mm ? 0 : (uint16_t)pi.nm;
        const int64_t base_off = B.ref_off[(size_t)r.tid];
        const int64_t ref_len = B.ref_len[(size_t)r.tid];
        const uint8_t *ql = B.quals.data() + r.qual_off, *sq = B.seqs.data() + r.seq_off;
        int64_t ref = r.pos, q = 0;
        uint64_t n_out = 0;
        for (int k = 0; k < r.n_cigar; k++) {
            const uint32_t c = B.cigars[r.cigar_off + k];
            const int op = c & 15;
            const int64_t n = c >> 4;
            if (op == CM || op == CEQ || op == CX) {
                // the reference's pileups are truncated to [0, scaffold length) (profile_utilities.py:150-153)
                const int64_t j0 = std::max<int64_t>(0, -ref), j1 = std::min<int64_t>(n, ref_len - ref);
                const uint32_t g0 = (uint32_t)(base_off + ref);
                for (int64_t j = j0; j < j1; j++) {
                    if (ql[q + j] >= minq) {
                        if (write) {
                            isx_obs &o = po[n_out];
                            o.gpos = g0 + (uint32_t)j; o.mm = mm; o.base = CODE2IDX[sq[q + j]]; o.flags = 0;
                            pp[n_out] = pi.pair_id;
                        }
                        n_out++;
                    }
                }
                q += n; ref += n;
            } else if (op == CI || op == CS) q += n;
            else if (op == CD || op == CN) ref += n;
        }
        return n_out;
    };
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(64u, std::max(1u, std::thread::hardware_concurrency())), n_reads / 4096 + 1));
    auto run_threads = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(fn, t);
        fn(0u);
        for (auto &x : th) x.join();
    };
    run_threads([&](unsigned t) {
        for (size_t ri = n_reads * t / nt; ri < n_reads * (t + 1) / nt; ri++)
            if (emit[ri]) out_at[ri + 1] = walk(ri, false, nullptr, nullptr);
    });
    for (size_t ri = 0; ri < n_reads; ri++) out_at[ri + 1] += out_at[ri];
    const size_t n_out = (size_t)out_at[n_reads];
    B.obs.reset(new isx_obs[std::max<size_t>(n_out, 1)]);
    B.pair.reset(new uint32_t[std::max<size_t>(n_out, 1)]);
    B.n_obs = n_out;
    run_threads([&](unsigned t) {
        for (size_t ri = n_reads * t / nt; ri < n_reads * (t + 1) / nt; ri++)
            if (emit[ri]) walk(ri, true, B.obs.get() + out_at[ri], B.pair.get() + out_at[ri]);
    });

    stage("expansion");
    // ---- iterate_splits (fasta.py:56-73) on the flat space ----
    B.split_bounds.clear(); B.split_ref.clear();
    const int64_t W = p->window_length > 0 ? p->window_length : 10000;
    for (size_t t = 0; t < n_ref; t++) {
        const int64_t sLen = B.ref_len[t];
        if (sLen <= 0) continue;
        const int64_t n_chunks = sLen / W + 1;
        const int64_t chunk = (int64_t)((double)sLen / (double)n_chunks);
        int64_t start = 0;
        for (int64_t i = 0; i < n_chunks; i++) {
            B.split_bounds.push_back(B.ref_off[t] + start);
            B.split_ref.push_back((int32_t)t);
            if (i + 1 < n_chunks) start += chunk;
        }
    }
    int64_t n_pos = 0;
    for (int64_t l : B.ref_len) n_pos += l;
    B.split_bounds.push_back(n_pos);
    B.expanded = true;

    info->n_refs = (int32_t)n_ref;
    info->n_splits = (int32_t)B.split_ref.size();
    info->n_reads = (int64_t)B.reads.size();
    info->n_pos = n_pos;
    info->n_obs = (int64_t)B.n_obs;
    info->n_pairs = next_pair;
    return ISX_OK;
}

int isx_bam_copy(const isx_bam *bam, isx_obs *obs, uint32_t *pair, int64_t *split_bounds, int32_t *split_ref)
{
    if (!bam || !bam->expanded) { isx_set_error("isx_bam_copy: expand first"); return ISX_ERR_STATE; }
    if (obs && bam->n_obs) memcpy(obs, bam->obs.get(), bam->n_obs * sizeof(isx_obs));
    if (pair && bam->n_obs) memcpy(pair, bam->pair.get(), bam->n_obs * sizeof(uint32_t));
    if (split_bounds) memcpy(split_bounds, bam->split_bounds.data(), bam->split_bounds.size() * sizeof(int64_t));
    if (split_ref && !bam->split_ref.empty()) memcpy(split_ref, bam->split_ref.data(), bam->split_ref.size() * sizeof(int32_t));
    return ISX_OK;
}

/* zero-copy access to what isx_bam_copy copies; valid until isx_bam_close */
int isx_bam_view(const isx_bam *bam, const isx_obs **obs, const uint32_t **pair)
{
    if (!bam || !bam->expanded) { isx_set_error("isx_bam_view: expand first"); return ISX_ERR_STATE; }
    if (obs) *obs = bam->obs.get();
    if (pair) *pair = bam->pair.get();
    return ISX_OK;
}

}  // extern "C"
