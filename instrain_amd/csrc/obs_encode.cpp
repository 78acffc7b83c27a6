// obs_encode.cpp -- see obs_encode.h.  Host C++ only (no device code); AVX-512 where the host has it.
#include "obs_encode.h"

#include <immintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

namespace isxenc {

namespace {

constexpr uint32_t G16 = 512, G32 = 256, CHUNK = 1024, PADREC = 2048;
constexpr uint32_t PAD32 = 0x0700FFFFu;

bool have_avx512()
{
    static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    return v;
}

// ---- CPU topology (sysfs): allowed cpus of a NUMA node grouped by the L3 they share ----
std::vector<int> parse_cpulist(const std::string &s)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        char *end = nullptr;
        const long a = strtol(s.c_str() + i, &end, 10);
        if (end == s.c_str() + i) break;
        long b = a;
        i = (size_t)(end - s.c_str());
        if (i < s.size() && s[i] == '-') { b = strtol(s.c_str() + i + 1, &end, 10); i = (size_t)(end - s.c_str()); }
        for (long c = a; c <= b; c++) out.push_back((int)c);
        while (i < s.size() && (s[i] == ',' || s[i] == '\n' || s[i] == ' ')) i++;
    }
    return out;
}

std::string slurp(const std::string &path)
{
    std::string out;
    if (FILE *f = fopen(path.c_str(), "r")) {
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
        fclose(f);
    }
    return out;
}

std::vector<std::vector<int>> l3_domains(int numa_node)
{
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return {};
    std::vector<int> cpus;
    if (numa_node >= 0) cpus = parse_cpulist(slurp("/sys/devices/system/node/node" + std::to_string(numa_node) + "/cpulist"));
    if (cpus.empty()) for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
    std::map<std::string, std::vector<int>> by_l3;
    for (int c : cpus) {
        if (!CPU_ISSET(c, &allowed)) continue;
        std::string key = slurp("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/cache/index3/shared_cpu_list");
        if (key.empty()) key = "all";
        by_l3[key].push_back(c);
    }
    std::vector<std::vector<int>> out;
    for (auto &kv : by_l3) out.push_back(kv.second);
    return out;
}

}  // namespace

HostPool::HostPool(int n_threads, int numa_node, bool pin)
{
    n_threads = std::max(1, n_threads);
    std::vector<std::vector<int>> dom;
    if (pin) dom = l3_domains(numa_node);
    cpus_.resize((size_t)n_threads);
    if (!dom.empty())
        for (int i = 0; i < n_threads; i++) cpus_[(size_t)i] = dom[(size_t)i % dom.size()];
    for (int i = 1; i < n_threads; i++) th_.emplace_back(&HostPool::worker, this, i);
}

HostPool::~HostPool()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
}

void HostPool::worker(int idx)
{
    if (!cpus_[(size_t)idx].empty()) {
        cpu_set_t s;
        CPU_ZERO(&s);
        for (int c : cpus_[(size_t)idx]) CPU_SET(c, &s);
        (void)sched_setaffinity(0, sizeof s, &s);
    }
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        while (next_ < n_tasks_) {
            const int t = next_++;
            const auto *fn = fn_;
            lk.unlock();
            (*fn)(t);
            lk.lock();
        }
        if (--running_ == 0) cv_done_.notify_all();
    }
}

void HostPool::run(int n_tasks, const std::function<void(int)> &fn)
{
    if (n_tasks <= 0) return;
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &fn; n_tasks_ = n_tasks; next_ = 0;
    running_ = (int)th_.size() + 1;
    ++gen_;
    lk.unlock();
    cv_.notify_all();
    lk.lock();
    while (next_ < n_tasks_) {
        const int t = next_++;
        lk.unlock();
        fn(t);
        lk.lock();
    }
    --running_;
    cv_done_.wait(lk, [&] { return running_ == 0; });
    fn_ = nullptr;
}

namespace {

// ---- per-run statistics: lowest / highest position, highest mm level ----
void stat_scalar(const isx_obs *s, size_t n, uint32_t &lo, uint32_t &hi, uint32_t &mm)
{
    uint32_t l = 0xFFFFFFFFu, h = 0, m = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t g = s[i].gpos;
        l = g < l ? g : l; h = g > h ? g : h; m = s[i].mm > m ? s[i].mm : m;
    }
    lo = l; hi = h; mm = m;
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void stat_avx512(const isx_obs *s, size_t n, uint32_t &lo, uint32_t &hi, uint32_t &mm)
{
    __m512i vlo = _mm512_set1_epi32(-1), vhi = _mm512_setzero_si512(), vmm = _mm512_setzero_si512();
    const __m512i m16 = _mm512_set1_epi32(0xFFFF);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m512i v = _mm512_loadu_si512(reinterpret_cast<const void *>(s + i));
        vlo = _mm512_mask_min_epu32(vlo, 0x5555, vlo, v);
        vhi = _mm512_mask_max_epu32(vhi, 0x5555, vhi, v);
        vmm = _mm512_mask_max_epu32(vmm, 0xAAAA, vmm, _mm512_and_si512(v, m16));
    }
    uint32_t l = _mm512_mask_reduce_min_epu32(0x5555, vlo), h = _mm512_mask_reduce_max_epu32(0x5555, vhi);
    uint32_t m = _mm512_mask_reduce_max_epu32(0xAAAA, vmm);
    for (; i < n; i++) {
        const uint32_t g = s[i].gpos;
        l = g < l ? g : l; h = g > h ? g : h; m = s[i].mm > m ? s[i].mm : m;
    }
    lo = l; hi = h; mm = m;
}

// ---- record encoders: n records of one run, positions relative to lo ----
void enc16_scalar(const isx_obs *s, size_t n, uint32_t lo, uint16_t *d)
{
    for (size_t i = 0; i < n; i++) {
        const uint32_t bc = s[i].base > 4 ? 4u : (uint32_t)s[i].base;
        d[i] = (uint16_t)((s[i].gpos - lo) | (bc << 13));
    }
}

void enc32_scalar(const isx_obs *s, size_t n, uint32_t lo, uint32_t *d)
{
    for (size_t i = 0; i < n; i++) {
        const uint32_t bc = s[i].base > 4 ? 4u : (uint32_t)s[i].base;
        d[i] = (s[i].gpos - lo) | ((uint32_t)s[i].mm << 16) | (bc << 24);
    }
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void enc16_avx512(const isx_obs *s, size_t n, uint32_t lo, uint16_t *d)
{
    const __m512i ie = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i io = _mm512_setr_epi32(1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31);
    const __m512i vlo = _mm512_set1_epi32((int)lo), four = _mm512_set1_epi32(4), m8 = _mm512_set1_epi32(0xFF);
    const bool aligned = (reinterpret_cast<uintptr_t>(d) & 31u) == 0;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i a = _mm512_loadu_si512(reinterpret_cast<const void *>(s + i));
        const __m512i b = _mm512_loadu_si512(reinterpret_cast<const void *>(s + i + 8));
        const __m512i g = _mm512_permutex2var_epi32(a, ie, b), o = _mm512_permutex2var_epi32(a, io, b);
        const __m512i bs = _mm512_min_epu32(_mm512_and_si512(_mm512_srli_epi32(o, 16), m8), four);
        const __m512i r = _mm512_or_si512(_mm512_sub_epi32(g, vlo), _mm512_slli_epi32(bs, 13));
        const __m256i out = _mm512_cvtepi32_epi16(r);
        if (aligned) _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i), out);
        else _mm256_storeu_si256(reinterpret_cast<__m256i *>(d + i), out);
    }
    enc16_scalar(s + i, n - i, lo, d + i);
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void enc32_avx512(const isx_obs *s, size_t n, uint32_t lo, uint32_t *d)
{
    const __m512i ie = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i io = _mm512_setr_epi32(1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31);
    const __m512i vlo = _mm512_set1_epi32((int)lo), four = _mm512_set1_epi32(4), m8 = _mm512_set1_epi32(0xFF);
    const bool aligned = (reinterpret_cast<uintptr_t>(d) & 63u) == 0;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i a = _mm512_loadu_si512(reinterpret_cast<const void *>(s + i));
        const __m512i b = _mm512_loadu_si512(reinterpret_cast<const void *>(s + i + 8));
        const __m512i g = _mm512_permutex2var_epi32(a, ie, b), o = _mm512_permutex2var_epi32(a, io, b);
        const __m512i bs = _mm512_min_epu32(_mm512_and_si512(_mm512_srli_epi32(o, 16), m8), four);
        const __m512i mm = _mm512_slli_epi32(_mm512_and_si512(o, m8), 16);
        const __m512i r = _mm512_or_si512(_mm512_or_si512(_mm512_sub_epi32(g, vlo), mm), _mm512_slli_epi32(bs, 24));
        if (aligned) _mm512_stream_si512(reinterpret_cast<__m512i *>(d + i), r);
        else _mm512_storeu_si512(reinterpret_cast<void *>(d + i), r);
    }
    enc32_scalar(s + i, n - i, lo, d + i);
}

// run starts of a group's pair ids: 16 ids per compare against their predecessors; a read's ~100 records share one id, so
// almost every compare finds nothing
__attribute__((target("avx512f,avx512bw,avx512vl")))
void pair_runs_avx512(const uint32_t *p, uint32_t n, uint32_t r0, std::vector<PairRun> &runs, uint32_t &maxp)
{
    if (!n) return;
    uint32_t last = runs.empty() ? ~p[0] : runs.back().pair;
    uint32_t m = maxp;
    uint32_t i = 0;
    if (p[0] != last || runs.empty()) { runs.push_back(PairRun{r0, p[0]}); m = p[0] > m ? p[0] : m; }
    i = 1;
    for (; i + 16 <= n; i += 16) {
        const __m512i v = _mm512_loadu_si512(reinterpret_cast<const void *>(p + i));
        const __m512i q = _mm512_loadu_si512(reinterpret_cast<const void *>(p + i - 1));
        __mmask16 k = _mm512_cmpneq_epu32_mask(v, q);
        while (k) {
            const int b = __builtin_ctz((unsigned)k);
            k = (__mmask16)(k & (k - 1));
            const uint32_t x = p[i + (uint32_t)b];
            runs.push_back(PairRun{r0 + i + (uint32_t)b, x});
            m = x > m ? x : m;
        }
    }
    for (; i < n; i++)
        if (p[i] != p[i - 1]) { runs.push_back(PairRun{r0 + i, p[i]}); m = p[i] > m ? p[i] : m; }
    maxp = m;
}

struct Task {
    int64_t in_a = 0, in_b = 0;         // input groups
    int64_t out_a = 0, out_b = 0;       // device groups of the task's region
    int64_t need = 0;                   // device groups the task's input needs
    uint32_t max_pair = 0;
    int err = 0;
    int ran = -1;                       // the pass that last ran this task
    std::vector<PairRun> runs;          // pair-id runs of the task's region, ascending device records
};

}  // namespace

const char *encode_isa() { return have_avx512() ? "avx512" : "scalar"; }

int encode_obs(HostPool &pool, EncodeJob &J)
{
    const bool w16 = J.record_bytes == 2;
    const uint32_t G = w16 ? G16 : G32, SPAN = w16 ? 8191u : 65535u;
    const int64_t gpc = CHUNK / G, gpp = PADREC / G;        // groups per directory chunk / per padding unit
    const int64_t n_in = (J.n_obs + G - 1) / G;
    J.n_groups_in = n_in;
    const bool fast = have_avx512();
    uint16_t *rec16 = static_cast<uint16_t *>(J.rec);
    uint32_t *rec32 = static_cast<uint32_t *>(J.rec);

    const bool ring = J.ring_groups > 0;
    const int64_t H = J.ring_groups;                        // device groups per ring half
    if (ring && (H % gpp != 0 || H < 2 * gpp)) return ENC_CAPACITY;
    int n_tasks = 0;
    std::vector<Task> tasks;
    auto build_tasks = [&](int64_t per_max) {
        n_tasks = (int)std::max<int64_t>(1, std::min<int64_t>(n_in / 64, (int64_t)8 * pool.size()));
        int64_t per = ((n_in + n_tasks - 1) / n_tasks + gpc - 1) / gpc * gpc;
        if (per_max > 0) per = std::max<int64_t>(gpc, std::min<int64_t>(per, per_max / gpc * gpc));
        n_tasks = per ? (int)((n_in + per - 1) / per) : 0;
        if (ring && n_tasks == 0) n_tasks = 1;              // an empty batch still pads its first wave
        tasks.assign((size_t)n_tasks, Task());
        for (int t = 0; t < n_tasks; t++) { tasks[(size_t)t].in_a = std::min<int64_t>(n_in, per * t); tasks[(size_t)t].in_b = std::min<int64_t>(n_in, per * (t + 1)); }
    };
    // ring mode: a wave should hold a few tasks per thread, and a task's region must fit a half with room to spare
    build_tasks(ring ? std::max<int64_t>(gpc, H / (2 * (int64_t)pool.size())) : 0);
    int64_t rec_shift = 0;                                  // ring mode: device group g of the running wave lives at group g + rec_shift of `rec`

    std::atomic<int> overflow{0};
    auto pad_groups = [&](int64_t g0, int64_t g1) {             // whole padding groups [g0, g1)
        if (g1 <= g0) return;
        if (w16) memset(rec16 + (g0 + rec_shift) * G, 0xFF, (size_t)(g1 - g0) * G * 2);
        else std::fill(rec32 + (g0 + rec_shift) * G, rec32 + (g1 + rec_shift) * G, PAD32);
        std::fill(J.gbase + g0, J.gbase + g1, 0u);
        if (J.pair_out) memset(J.pair_out + g0 * G, 0, (size_t)(g1 - g0) * G * 4);
    };
    const bool have_pairs = J.pair != nullptr || (J.obs == nullptr && J.want_pairs);
    auto emit = [&](int64_t og, const isx_obs *src, const uint32_t *psrc, uint32_t n, uint32_t lo, uint32_t hi, uint32_t &maxp,
                    std::vector<PairRun> &runs) {
        if (w16) {
            uint16_t *d = rec16 + (og + rec_shift) * G;
            if (fast) enc16_avx512(src, n, lo, d); else enc16_scalar(src, n, lo, d);
            if (n < G) memset(d + n, 0xFF, (size_t)(G - n) * 2);
        } else {
            uint32_t *d = rec32 + (og + rec_shift) * G;
            if (fast) enc32_avx512(src, n, lo, d); else enc32_scalar(src, n, lo, d);
            for (uint32_t i = n; i < G; i++) d[i] = PAD32;
        }
        J.gbase[og] = lo;
        if (J.pair_out) {
            uint32_t *pd = J.pair_out + og * G;
            uint32_t m = maxp;
            for (uint32_t i = 0; i < n; i++) { const uint32_t p = psrc[i]; pd[i] = p; m = p > m ? p : m; }
            for (uint32_t i = n; i < G; i++) pd[i] = 0;
            maxp = m;
        }
        if (J.runs_out && have_pairs) {
            const uint32_t r0 = (uint32_t)(og * G);
            if (fast) pair_runs_avx512(psrc, n, r0, runs, maxp);
            else {
                uint32_t last = runs.empty() ? 0xFFFFFFFFu : runs.back().pair;
                uint32_t m = maxp;
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t p = psrc[i];
                    if (p != last || runs.empty()) { runs.push_back(PairRun{r0 + i, p}); last = p; m = p > m ? p : m; }
                }
                maxp = m;
            }
        }
        const int64_t ch = og / gpc;
        J.cmin[ch] = std::min(J.cmin[ch], lo); J.cmax[ch] = std::max(J.cmax[ch], hi); J.cany[ch] = 1;
    };
    auto work = [&](int ti) {
        Task &T = tasks[(size_t)ti];
        T.need = 0; T.max_pair = 0; T.err = 0; T.runs.clear(); T.ran = J.passes;
        std::vector<isx_obs> tmp_obs;           // producer mode: one input group at a time
        std::vector<uint32_t> tmp_pair;
        if (!J.obs) { tmp_obs.resize(G); if (have_pairs) tmp_pair.resize(G); }
        bool writing = !overflow.load(std::memory_order_relaxed);
        if (writing)
            for (int64_t ch = T.out_a / gpc; ch < T.out_b / gpc; ch++) { J.cmin[ch] = 0xFFFFFFFFu; J.cmax[ch] = 0; J.cany[ch] = 0; }
        int64_t out = T.out_a;
        const uint32_t *pair_src = nullptr;     // pair ids of the current input group (array or producer buffer)
        int64_t pair_src0 = 0;
        auto put = [&](const isx_obs *src, int64_t first, uint32_t n, uint32_t lo, uint32_t hi) {
            T.need++;
            if (!writing) return;
            if (out == T.out_b || overflow.load(std::memory_order_relaxed)) { overflow.store(1); writing = false; return; }
            emit(out++, src, pair_src ? pair_src + (first - pair_src0) : nullptr, n, lo, hi, T.max_pair, T.runs);
        };
        for (int64_t ig = T.in_a; ig < T.in_b; ig++) {
            const int64_t s0 = ig * (int64_t)G;
            const uint32_t n = (uint32_t)std::min<int64_t>(G, J.n_obs - s0);
            const isx_obs *src;
            if (J.obs) { src = J.obs + s0; pair_src = J.pair; pair_src0 = 0; }
            else {
                J.produce(s0, n, tmp_obs.data(), have_pairs ? tmp_pair.data() : nullptr);
                src = tmp_obs.data(); pair_src = have_pairs ? tmp_pair.data() : nullptr; pair_src0 = s0;
            }
            uint32_t lo, hi, mm;
            if (fast) stat_avx512(src, n, lo, hi, mm); else stat_scalar(src, n, lo, hi, mm);
            if (!w16 && mm >= 256u) { T.err = ENC_MM_RANGE; return; }
            if ((int64_t)hi >= J.n_pos) { T.err = ENC_BAD_POS; return; }
            if (hi - lo < SPAN) { put(src, s0, n, lo, hi); continue; }
            // the stream jumps inside this group: greedy runs in arrival order
            uint32_t r0 = 0, rl = 0xFFFFFFFFu, rh = 0;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t p = src[i].gpos;
                const uint32_t nl = p < rl ? p : rl, nh = p > rh ? p : rh;
                if (i > r0 && nh - nl >= SPAN) { put(src + r0, s0 + r0, i - r0, rl, rh); r0 = i; rl = rh = p; }
                else { rl = nl; rh = nh; }
            }
            put(src + r0, s0 + r0, n - r0, rl, rh);
        }
        if (writing) pad_groups(out, T.out_b);
        _mm_sfence();               // streaming stores visible before the DMA engine reads the staging buffer
    };

    auto layout = [&](bool exact) -> int64_t {              // regions from the slack estimate / from the exact needs
        int64_t g = 0;
        for (auto &T : tasks) {
            const int64_t nin = T.in_b - T.in_a;
            int64_t want = exact ? T.need : nin + (J.slack > 0 ? (int64_t)(nin * J.slack) + gpc : 0);
            want = (want + gpc - 1) / gpc * gpc;
            T.out_a = g; T.out_b = g + want;
            g += want;
        }
        return g;
    };
    int64_t total = layout(false);
    J.passes = 0;
    bool retasked = false;
    for (int pass = 0; pass < 3; pass++) {
        const int64_t n_rec = std::max<int64_t>(PADREC, (total * G + PADREC - 1) / PADREC * PADREC);
        bool fits = n_rec <= J.cap_rec;
        if (ring && fits) {
            tasks.back().out_b = n_rec / G;                 // the last task pads up to n_rec: padding travels with the last wave
            for (auto &T : tasks) if (T.out_b - T.out_a > H) fits = false;
            if (!fits && pass > 0 && !retasked) {           // a region larger than a ring half (a stream that jumps all the time):
                retasked = true;                            // tasks so small that even one record per device group fits
                build_tasks(std::max<int64_t>(gpc, H / (int64_t)G));
                total = layout(false);
                overflow.store(1);
                J.passes++;
                pool.run(n_tasks, work);                    // count only
                for (auto &T : tasks) if (T.err) return T.err;
                overflow.store(0);
                total = layout(true);
                continue;
            }
        }
        if (!fits) {
            if (pass == 0) overflow.store(1);               // the estimate does not fit: count only, then decide
            else return ENC_CAPACITY;
        }
        J.passes++;
        if (!ring || overflow.load()) pool.run(n_tasks, work);
        else {
            int wave = 0;
            for (int t0 = 0; t0 < n_tasks && !overflow.load();) {
                int t1 = t0 + 1;
                while (t1 < n_tasks && tasks[(size_t)t1].out_b - tasks[(size_t)t0].out_a <= H) t1++;
                const int half = wave & 1;
                const int64_t g0 = tasks[(size_t)t0].out_a, g1 = tasks[(size_t)t1 - 1].out_b;
                J.wave_begin(half);
                rec_shift = (int64_t)half * H - g0;
                pool.run(t1 - t0, [&](int i) { work(t0 + i); });
                bool bad = overflow.load() != 0;
                for (int t = t0; t < t1; t++) if (tasks[(size_t)t].err) bad = true;
                if (!bad) J.wave_flush(half, g0, g1);
                else if (!overflow.load()) break;           // a task refused its input: reported below
                t0 = t1; wave++;
            }
            rec_shift = 0;
            if (overflow.load()) {                          // the waves after the overflow still have to be counted
                std::vector<int> rest;
                for (int t = 0; t < n_tasks; t++) if (tasks[(size_t)t].ran != J.passes) rest.push_back(t);
                pool.run((int)rest.size(), [&](int i) { work(rest[(size_t)i]); });
            }
        }
        for (auto &T : tasks) if (T.err) return T.err;
        if (!overflow.load()) {
            J.n_rec = n_rec;
            if (!ring) {
                for (int64_t ch = total / gpc; ch < n_rec / CHUNK; ch++) { J.cmin[ch] = 0xFFFFFFFFu; J.cmax[ch] = 0; J.cany[ch] = 0; }
                pad_groups(total, n_rec / G);
                _mm_sfence();
            }
            int64_t real = 0;
            uint32_t mp = 0;
            for (auto &T : tasks) { real += T.need; mp = std::max(mp, T.max_pair); }
            if (J.runs_out) {                               // tasks own ascending regions: concatenation stays sorted
                std::vector<size_t> at(tasks.size() + 1, 0);
                for (size_t t = 0; t < tasks.size(); t++) at[t + 1] = at[t] + tasks[t].runs.size();
                J.n_runs = at.back();
                if (J.n_runs == 0) { J.n_runs = 1; if (J.cap_runs) J.runs_out[0] = PairRun{0u, 0u}; at.back() = 1; }
                else if (J.n_runs <= J.cap_runs)
                    pool.run((int)tasks.size(), [&](int t) {
                        if (!tasks[(size_t)t].runs.empty())
                            memcpy(J.runs_out + at[(size_t)t], tasks[(size_t)t].runs.data(), tasks[(size_t)t].runs.size() * sizeof(PairRun));
                    });
                if (J.n_runs <= J.cap_runs && J.run_index_out) {
                    const int64_t n_ch = n_rec / CHUNK;
                    const int pieces = (int)std::max<int64_t>(1, std::min<int64_t>(n_ch / 4096, (int64_t)4 * pool.size()));
                    const PairRun *R = J.runs_out;
                    const size_t nR = J.n_runs;
                    pool.run(pieces, [&](int pc) {
                        const int64_t c0 = n_ch * pc / pieces, c1 = n_ch * (pc + 1) / pieces;
                        // last run with first <= c0 * CHUNK
                        size_t lo = 0, hi = nR;
                        const uint32_t f0 = (uint32_t)(c0 * CHUNK);
                        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (R[mid].first <= f0) lo = mid; else hi = mid; }
                        size_t r = lo;
                        for (int64_t ch = c0; ch < c1; ch++) {
                            const uint32_t first = (uint32_t)(ch * CHUNK);
                            while (r + 1 < nR && R[r + 1].first <= first) r++;
                            J.run_index_out[ch] = (uint32_t)r;
                        }
                    });
                }
            }
            J.n_groups_real = real; J.max_pair = mp;
            (void)gpp;
            return ENC_OK;
        }
        overflow.store(0);
        total = layout(true);
    }
    return ENC_CAPACITY;
}

}  // namespace isxenc
