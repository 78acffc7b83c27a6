// lev_expand.cpp -- host side of the level-sparse hand-back of mm profiling (isx_pipe_result.lev_*, include/instrain_amd.h): the
// four columns shrink_basewise's inputs are cut from (profile_utilities.py:337-350: covT = a level's own coverage, :288-295;
// clonT / clonTR = clonality of the counts up to the level, snv_utilities.py:85-104), rebuilt from
//   the per-position level mask | one coverage element per present level | the windows' first level indices | the lists of
//   saturated coverages, clonalities other than 1.0 and rarefied clonalities
// No GPU call in here.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/instrain_amd.h"

void isx_set_error(const std::string &msg);

namespace {

inline uint32_t mask_at(const void *m, int bytes, size_t p)
{
    if (bytes == 1) return static_cast<const uint8_t *>(m)[p];
    if (bytes == 2) return static_cast<const uint16_t *>(m)[p];
    return static_cast<const uint32_t *>(m)[p];
}

template <class F>
void run_threads(int n_threads, int n_tasks, const F &fn)
{
    n_threads = std::max(1, std::min(n_threads, n_tasks));
    if (n_threads == 1) { for (int t = 0; t < n_tasks; t++) fn(t); return; }
    std::vector<std::thread> th;
    for (int k = 1; k < n_threads; k++)
        th.emplace_back([&, k] { for (int t = k; t < n_tasks; t += n_threads) fn(t); });
    for (int t = 0; t < n_tasks; t += n_threads) fn(t);
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" int isx_levels_expand(const isx_pipe_result *r, int32_t host_threads, uint32_t *gpos, uint32_t *mm_cov, float *clon, float *clon_rarefied)
{
    // (clon_rarefied may be NULL when the result holds no rarefied clonality at all, n_lev_rare == 0: a column of NaNs less to write)
    if (!r || !gpos || !mm_cov || !clon || (!clon_rarefied && r->n_lev_rare)) { isx_set_error("isx_levels_expand: bad argument"); return ISX_ERR_ARG; }
    if (!r->lev_mask || r->lev_window <= 0 || (r->n_lev && (!r->lev_cov || !r->lev_win_off))) {
        isx_set_error("isx_levels_expand: the result holds no level-sparse tables (n_mm_bins in 2..32, read-level pipe without want_counts)");
        return ISX_ERR_STATE;
    }
    const size_t n_pos = (size_t)r->n_pos, W = (size_t)r->lev_window, n_lev = (size_t)r->n_lev;
    const size_t n_win = (size_t)r->n_lev_windows;
    if (n_win != (n_pos + W - 1) / W) { isx_set_error("isx_levels_expand: inconsistent window count"); return ISX_ERR_STATE; }
    const int mb = r->lev_mask_bytes, cb = r->lev_cov_bytes;
    const uint32_t sat_thr = cb == 1 ? 255u : 65535u;
    // levels per window -> where the window's levels go in (gpos, mm) order
    std::vector<uint64_t> out_off(n_win + 1, 0);
    const int T = std::max(1, (int)host_threads);
    const int n_tasks = (int)std::min<size_t>(n_win, (size_t)T * 8);
    auto task_range = [&](int t, size_t &a, size_t &e) { a = n_win * (size_t)t / (size_t)n_tasks; e = n_win * (size_t)(t + 1) / (size_t)n_tasks; };
    run_threads(T, n_tasks, [&](int t) {
        size_t a, e;
        task_range(t, a, e);
        for (size_t w = a; w < e; w++) {
            const size_t p0 = w * W, p1 = std::min(n_pos, p0 + W);
            uint64_t n = 0;
            if (mb == 1) { const uint8_t *m = static_cast<const uint8_t *>(r->lev_mask); for (size_t p = p0; p < p1; p++) n += (unsigned)__builtin_popcount(m[p]); }
            else for (size_t p = p0; p < p1; p++) n += (unsigned)__builtin_popcount(mask_at(r->lev_mask, mb, p));
            out_off[w + 1] = n;
        }
    });
    for (size_t w = 0; w < n_win; w++) out_off[w + 1] += out_off[w];
    if (out_off[n_win] != n_lev) { isx_set_error("isx_levels_expand: the level masks do not add up to n_lev"); return ISX_ERR_STATE; }
    for (size_t w = 0; w < n_win; w++) {
        const uint64_t n = out_off[w + 1] - out_off[w];
        if (n && (uint64_t)r->lev_win_off[w] + n > n_lev) { isx_set_error("isx_levels_expand: a window's level range lies outside the coverage stream"); return ISX_ERR_STATE; }
    }
    // exact coverage of the saturated levels, by device index
    std::vector<isx_sat> sat(r->lev_sat, r->lev_sat + (r->lev_sat ? r->n_lev_sat : 0));
    std::sort(sat.begin(), sat.end(), [](const isx_sat &x, const isx_sat &y) { return x.gpos < y.gpos; });
    auto sat_cov = [&](uint32_t idx, uint32_t dflt) -> uint32_t {
        auto it = std::lower_bound(sat.begin(), sat.end(), idx, [](const isx_sat &x, uint32_t v) { return x.gpos < v; });
        return it != sat.end() && it->gpos == idx ? it->coverage : dflt;
    };
    const int64_t min_cov = r->lev_min_cov;
    const float nanf_ = nanf("");
    std::atomic<int> too_deep{0};
    run_threads(T, n_tasks, [&](int t) {
        size_t a, e;
        task_range(t, a, e);
        for (size_t w = a; w < e; w++) {
            const size_t p0 = w * W, p1 = std::min(n_pos, p0 + W);
            uint64_t o = out_off[w];
            if (o == out_off[w + 1]) continue;
            size_t d = r->lev_win_off[w];
            for (size_t p = p0; p < p1; p++) {
                uint32_t m = mask_at(r->lev_mask, mb, p);
                uint64_t cum = 0;
                while (m) {
                    const int lvl = __builtin_ctz(m);
                    m &= m - 1;
                    uint32_t cov = cb == 1 ? static_cast<const uint8_t *>(r->lev_cov)[d] : static_cast<const uint16_t *>(r->lev_cov)[d];
                    if (cov == sat_thr) cov = sat_cov((uint32_t)d, cov);
                    if (cov >= (1u << 24)) too_deep.store(1, std::memory_order_relaxed);
                    cum += cov;
                    gpos[o] = (uint32_t)p;
                    mm_cov[o] = ((uint32_t)lvl << 24) | (cov & 0xFFFFFFu);
                    clon[o] = (int64_t)cum >= min_cov ? 1.0f : nanf_;
                    if (clon_rarefied) clon_rarefied[o] = nanf_;
                    o++; d++;
                }
            }
        }
    });
    if (too_deep.load()) { isx_set_error("a (position, mm) level with coverage >= 2^24: fetch the full entries (want_counts)"); return ISX_ERR_CAPACITY; }
    // the lists carry device indices: window by first index -> output index
    if (r->n_lev_clon || r->n_lev_rare) {
        std::vector<uint32_t> order;
        order.reserve(n_win);
        for (size_t w = 0; w < n_win; w++) if (out_off[w + 1] > out_off[w]) order.push_back((uint32_t)w);
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return r->lev_win_off[x] < r->lev_win_off[y]; });
        int bad = 0;
        auto to_out = [&](uint32_t idx) -> int64_t {
            size_t lo = 0, hi = order.size();
            while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (r->lev_win_off[order[mid]] <= idx) lo = mid + 1; else hi = mid; }
            if (lo == 0) return -1;
            const uint32_t w = order[lo - 1];
            const uint64_t k = (uint64_t)idx - r->lev_win_off[w];
            if (k >= out_off[w + 1] - out_off[w]) return -1;
            return (int64_t)(out_off[w] + k);
        };
        for (int64_t i = 0; i < r->n_lev_clon; i++) { const int64_t o = to_out(r->lev_clon[i].gpos); if (o < 0) bad = 1; else clon[o] = r->lev_clon[i].clon_rarefied; }
        for (int64_t i = 0; i < r->n_lev_rare; i++) { const int64_t o = to_out(r->lev_rare[i].gpos); if (o < 0) bad = 1; else clon_rarefied[o] = r->lev_rare[i].clon_rarefied; }
        if (bad) { isx_set_error("isx_levels_expand: a list entry points outside every window's level range"); return ISX_ERR_STATE; }
    }
    return ISX_OK;
}
