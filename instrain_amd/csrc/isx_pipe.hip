// isx_pipe.hip -- streaming hand-over: a stream of batches, each profiled exactly once (isx_pipe_* of the C ABI).
//
// Reference analogue: the command queue / result queue pair around the split workers
// (/root/reference/inStrain/profile/profile_controller.py:157-193 make_profile_queues, :243-271
// spawn_profile_workers, :273-314 recieve_profile_results): splits go in, SplitObjects come back, every split
// is profiled once.  Here the unit is a batch of splits and the three stages run on different engines:
//
//   host threads   isx_obs (8 B) -> resident records (2 / 4 B) straight into the slot's pinned staging
//   copy-in queue  split bounds | window directory | reference codes | group bases | run index | records -> the slot's
//                  device arena of the same layout (hipMemcpyAsync); a slot whose records exceed 512 MiB stages
//                  them through a pinned ring of two 128 MiB halves instead (waves: one half is copied while the
//                  threads fill the other), because pinning gigabytes costs more than profiling them
//   pass queue     k_pileup_dense / k_pileup_mm (+ the cursor publication) once the copy-in event has fired
//   copy-out queue counts | clonality | rarefied clonality | first SNV rows -> the slot's pinned result block
//
// A slot is a complete isx_batch whose device tables are sized once for the pipe's largest batch, so a
// submit allocates nothing and everything isx_batch_* offers (linkage stages, entry fetch, summaries,
// compare) works on a collected slot unchanged.
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "bam_front.h"
#include "isx_batch.h"
#include "obs_encode.h"
#include "seg_encode.h"
#include "isx_summary.h"
#include <unordered_map>

namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// host staging that is pinned for a streaming pipe and plain for a single-slot one (nothing to overlap the copy with, and
// pinning + unpinning costs several times the slower copy)
int host_block_alloc(void **p, size_t bytes, bool pinned)
{
    *p = nullptr;
    if (pinned) { HIP_TRY(isx_pin_malloc(p, std::max<size_t>(bytes, 1))); return ISX_OK; }
    if (posix_memalign(p, 4096, std::max<size_t>(bytes, 4096)) != 0) { *p = nullptr; isx_set_error("out of host memory"); return ISX_ERR_ARG; }
    return ISX_OK;
}

void host_block_free(void *p, bool pinned)
{
    if (!p) return;
    if (pinned) isx_pin_free(p); else free(p);
}

#define H2D(p, s) ((s).h2d ? (s).h2d : (p)->s_h2d)
struct Slot {
    isx_batch *b = nullptr;
    hipStream_t h2d = nullptr;              // this slot's copy-in queue when the pipe has two (else the pipe's)
    uint8_t *h_in = nullptr, *d_in = nullptr;
    size_t in_bytes = 0;                    // of the device arena; the pinned one ends after the ring when the pipe stages through one
    size_t off_bounds = 0, off_win = 0, off_ref = 0, off_gbase = 0, off_ridx = 0, off_rec = 0;   // records last
    size_t off_pairs = 0;                   // read-level pipe + linkage: one pair id per record (between the group bases and the records)
    // pair-id runs (linkage): their own pinned / device blocks, grown when a batch has more runs than any before it
    isxenc::PairRun *h_runs = nullptr;
    bool runs_pinned = true;
    uint2 *d_runs = nullptr;
    size_t cap_runs = 0;
    hipEvent_t ev_ring[2] = {nullptr, nullptr};     // ring mode: the copy that last read each half
    bool ring_busy[2] = {false, false};
    uint8_t *h_out = nullptr;
    uint8_t *h_small = nullptr;             // always pinned: the first SNV rows and clonTR entries (o_snv, o_rare are offsets into it)
    size_t small_bytes = 0;
    bool out_pinned = true;                 // false: the dense arrays go to plain memory through the pipe's bounce buffers (see slot_batch_create)
    size_t out_bytes = 0, o_counts = 0, o_clon = 0, o_clonr = 0, o_snv = 0, o_cov16 = 0, o_rare = 0;
    bool rare_dense = false;                // the clonTR table of the last batch went back as the dense array
    bool cov8 = false, clon_sparse = false; // how the last (shallow) batch's coverage / clonality went back
    bool cov4 = false;                      // ... coverage as the 4-bit plane + 16-bit rows (lean slots): rows at h_out + o_cov16 + cov_rows_off
    size_t cov_rows_off = 0;
    int cov_window = 0;
    std::vector<uint32_t> cov_row_win;
    std::vector<isx_sat> sat_rows;          // exact coverage of its saturated positions
    bool sat_complete = true;
    void *sort_temp = nullptr;              // device scratch of the clonality list's sort
    size_t sort_temp_bytes = 0;
    std::vector<float> clonr_big;           // that array when the pipe has no pinned room for it (no want_counts)
    std::vector<isx_rare> rare_big;         // more clonTR entries than the pinned block holds / the device list overflowed
    std::vector<uint32_t> cmin, cmax, dir_pmax, dir_smin;     // (dir_*: scratch of the window directory)
    std::vector<uint8_t> cany;
    std::vector<uint2> win;
    // mm path, level-sparse hand-back (isx_batch::lev_sparse): offsets into h_out and how much of each region the last batch used
    size_t o_lmask = 0, o_lwin = 0, o_lcov = 0, o_lclon = 0, o_lrare = 0;
    size_t lcov_room = 0, lclon_room = 0, lrare_room = 0;          // bytes
    std::vector<uint8_t> lcov_big;          // a batch whose coverage stream / lists outgrow the pinned rooms (a deep sample)
    std::vector<isx_rare> lclon_big, lrare_big;
    std::vector<isx_snv> snv_big;           // more SNV rows than the pinned block holds (rare)
    std::vector<isx_ld> ld_rows;            // the batch's LD rows (linkage), fetched by the finisher
    uint64_t rows_checksum = 0;             // over the SNV + LD rows' bytes (isx_pipe_result.rows_checksum)
    hipEvent_t ev_h2d0 = nullptr, ev_h2d1 = nullptr, ev_pass = nullptr, ev_d2h0 = nullptr, ev_d2h1 = nullptr;
    hipEvent_t ev_h2da = nullptr, ev_h2db = nullptr;   // a copy-in in two parts (the reference leaves before the records exist): end of the first, start of the second
    bool h2d_split = false;
    int64_t ticket = -1;
    BamBatch *dead_batch = nullptr;         // isx_pipe_submit_bam: the front end's batch, freed by the finisher after the slot's work
    bool ref_has_n = false;                 // the batch's reference holds positions that are not A/C/T/G: their bit plane travels too
    int state = 0;                          // 0 free, 1 submitted, 2 finished (tables on the host, linkage done)
    int rc = 0;                             // of the finishing step
    std::string err;
    float finish_wait_ms = 0.f;
    float encode_ms = 0.f;
    int encode_passes = 0;
    int64_t h2d_bytes = 0, d2h_bytes = 0;
    uint16_t *d_gpos16 = nullptr;           // compact stream + linkage: positions alone for the allele pass
};

// order-sensitive 64-bit checksum of a table's bytes (four multiply-add lanes over its 64-bit words: ~20 GB/s on one core)
static uint64_t bytes_checksum(const void *p, size_t bytes, uint64_t seed)
{
    const uint8_t *b = static_cast<const uint8_t *>(p);
    uint64_t a[4] = {seed ^ 0x9E3779B97F4A7C15ull, seed + 0xC2B2AE3D27D4EB4Full, seed ^ 0x165667B19E3779F9ull, seed + 0x27D4EB2F165667C5ull};
    const size_t n8 = bytes / 8;
    size_t i = 0;
    for (; i + 4 <= n8; i += 4) {
        uint64_t w[4];
        memcpy(w, b + 8 * i, 32);
        for (int k = 0; k < 4; k++) a[k] = a[k] * 0x100000001B3ull + w[k];
    }
    uint64_t h = (uint64_t)bytes;
    for (; i < n8; i++) { uint64_t w; memcpy(&w, b + 8 * i, 8); h = h * 0x100000001B3ull + w; }
    for (size_t j = n8 * 8; j < bytes; j++) h = h * 0x100000001B3ull + b[j];
    for (int k = 0; k < 4; k++) { h ^= a[k]; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; }
    return h;
}

// The reference codes of a batch as they travel and lie in a slot: a 2-bit plane (A C T G; anything else as 0), four positions a
// byte, and -- only when the batch holds a position that is not A/C/T/G -- a bit plane marking those (PileupArgs::ref_packed == 2,
// ref_n).  0.25 (0.375) bytes a position over PCIe instead of 0.5 (round 3's nibbles) or 1.  plane2 holds (n_pos + 3) / 4 bytes,
// nplane (n_pos + 7) / 8; returns whether the N plane is needed (it is always written).
static inline size_t ref2_bytes(int64_t n_pos) { return (((size_t)n_pos + 3) / 4 + 15) & ~(size_t)15; }
static inline size_t refn_bytes(int64_t n_pos) { return (((size_t)n_pos + 7) / 8 + 15) & ~(size_t)15; }
static bool pack_ref2(isxenc::HostPool &pool, const uint8_t *ref, int64_t n_pos, uint8_t *plane2, uint8_t *nplane)
{
    return isxenc::pack_ref_planes(pool, ref, n_pos, plane2, nplane);
}

// the caller's own planes into staging (isx_ref_planes): a copy on the pool's threads; returns whether the N plane marks a position
static bool copy_ref_planes(isxenc::HostPool &pool, const isx_ref_planes *rp, int64_t n_pos, uint8_t *plane2, uint8_t *nplane)
{
    const size_t b2 = ((size_t)n_pos + 3) / 4, bn = ((size_t)n_pos + 7) / 8;
    const size_t piece = (size_t)1 << 20;
    const int t2 = (int)((b2 + piece - 1) / piece), tn = rp->nplane ? (int)((bn + piece - 1) / piece) : 0;
    std::atomic<int> any{0};
    pool.run(t2 + tn, [&](int t) {
        if (t < t2) { const size_t a = (size_t)t * piece; memcpy(plane2 + a, rp->plane2 + a, std::min(piece, b2 - a)); return; }
        const size_t a = (size_t)(t - t2) * piece, n = std::min(piece, bn - a);
        const uint8_t *src = rp->nplane + a;
        uint8_t acc = 0;
        for (size_t i = 0; i < n; i++) acc |= src[i];
        memcpy(nplane + a, src, n);
        if (acc) any.store(1, std::memory_order_relaxed);
    });
    // (bits of the last bytes beyond n_pos are the caller's padding: the kernels never look at positions >= n_pos)
    return any.load() != 0;
}

// does a bit plane hold a set bit (the non-ACGT plane of a registered reference: the kernels take the plain path when it is empty)
static bool plane_any(isxenc::HostPool &pool, const uint8_t *pl, size_t bytes)
{
    const size_t piece = (size_t)1 << 20;
    const int nt = (int)((bytes + piece - 1) / piece);
    std::atomic<int> any{0};
    pool.run(nt, [&](int t) {
        const size_t a = (size_t)t * piece, n = std::min(piece, bytes - a);
        uint64_t acc = 0;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, pl + a + i, 8); acc |= v; }
        for (; i < n; i++) acc |= pl[a + i];
        if (acc) any.store(1, std::memory_order_relaxed);
    });
    return any.load() != 0;
}

// a cheap content check of a reference's 2-bit plane (a resident reference is found by the caller's KEY: a key reused for other planes would
// otherwise compare on the host against planes the device does not hold): 64-bit words sampled at 4096 spread places + the length
static uint64_t ref_plane_checksum(const uint8_t *plane2, int64_t n_pos)
{
    const size_t words = (((size_t)n_pos + 3) / 4) / 8;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n_pos;
    if (!words) { for (size_t i = 0; i < ((size_t)n_pos + 3) / 4; i++) h = (h ^ plane2[i]) * 0x100000001B3ull; return h; }
    const size_t n_s = std::min<size_t>(words, 4096), step = words / n_s;
    for (size_t k = 0; k < n_s; k++) {
        uint64_t w;
        memcpy(&w, plane2 + 8 * (k * step), 8);
        h = (h ^ w) * 0x100000001B3ull;
        h ^= h >> 29;
    }
    return h;
}

int cgroup_cpus()
{
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long period = 0;
        const int n = fscanf(f, "%63s %ld", q, &period);
        fclose(f);
        if (n == 2 && period > 0 && strcmp(q, "max") != 0) return (int)std::max<long>(1, atol(q) / period);
    }
    return 0;
}

int gpu_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return -1;
    std::string id(bus);
    std::transform(id.begin(), id.end(), id.begin(), [](unsigned char ch) { return (char)std::tolower(ch); });
    int node = -1;
    if (FILE *f = fopen(("/sys/bus/pci/devices/" + id + "/numa_node").c_str(), "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    return node;
}

}  // namespace

struct isx_pipe {
    isx_ctx *ctx = nullptr;
    isx_params prm{};
    isx_pipe_params pp{};
    std::unique_ptr<isxenc::HostPool> pool;
    std::vector<Slot> slots;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipStream_t s_h2d2 = nullptr;           // ISX_PIPE_H2D_STREAMS=2: the odd slots' copy-in queue (a second copy engine)
    hipStream_t s_fin = nullptr;            // the finishers' own queues: what they fetch must not wait behind the copy-outs of later batches
    hipStream_t s_fin2 = nullptr;           // (one per finisher thread: a sync of one must not wait for the other's 30 MB table)
    int64_t next_ticket = 0;
    int64_t cap_rec = 0;
    int64_t ring_half = 0;                  // records per half of a slot's staging ring; 0 = the pinned arena holds the whole stream
    int rb = 2;                             // record bytes
    bool segs = false;                      // a read-level pipe: 64-byte read-segment records (isx_pipe_params.max_segs > 0) ...
    bool drec = false;                      // ... or, with one mm bin, 32-byte reference-delta records in groups of 32
    int64_t dslack = 1;                     // learned: spare groups per encoder task (pieces of segments that differ from the reference a lot)
    uint32_t G = ISX_GROUP16;
    size_t snv_prefix = 0;                  // SNV rows copied out with the dense tables
    size_t rare_prefix = 0, cap_rare = 0;   // clonTR entries copied out with them / the device list's capacity
    double slack = 0.0;                     // learned: extra device groups per input group of the last jumping batch
    // the finisher: a thread that takes every submitted batch as soon as its copy-out has landed -- sizes, table growth +
    // repeated pass, linkage stages, row sorting -- so that this work overlaps the caller encoding the next batch
    std::thread finisher;
    std::thread finisher2;                  // depth >= 2: two batches are finished side by side (the work is host latency: syncs, small sorts)
    std::vector<std::thread> more_finishers;    // depth >= 4: further ones, each with a queue of its own (extra_fin)
    std::vector<hipStream_t> extra_fin;
    std::mutex bounce_mu;                   // the bounce buffers and fin_pool belong to one finisher at a time
    std::mutex mu;                          // slot states + the work queue
    std::condition_variable cv_work, cv_done;
    std::deque<int64_t> work;
    bool stop = false;
    std::mutex launch_mu;                   // pass-queue launches (submit vs. the finisher repeating a pass)
    // position-sized result arrays of slots whose result block is plain memory travel through these (finisher thread only)
    uint8_t *bounce[2] = {nullptr, nullptr};
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    size_t bounce_bytes = 0;
    std::unique_ptr<isxenc::HostPool> fin_pool;      // the finisher's own few threads (the encoder pool belongs to the caller's thread)
    // the stager (isx_pipe_params.stage_async): isx_pipe_submit_reads only queues the batch; this thread encodes it into the
    // slot's staging and enqueues its copies and kernels, so the caller's own per-batch work overlaps the encoding of the
    // batch before.  Jobs are staged in ticket order; next_ticket counts the staged ones, next_promise the handed-out ones.
    struct StageJob {
        int64_t ticket, n_pos;
        const uint8_t *ref;
        std::vector<int64_t> bounds;
        isx_segs segs;
        bool planes = false;            // isx_pipe_submit_planes: reads / rp instead of segs / ref
        isx_read_planes reads{};
        isx_ref_planes rp{};
    };
    // resident references (isx_ref_planes.key): device copies of a batch's reference planes, kept after their first trip
    struct RefEntry { uint8_t *d = nullptr; size_t bytes = 0; int64_t n_pos = 0; bool has_n = false; hipEvent_t ready = nullptr; uint64_t sum = 0; };
    std::unordered_map<uint64_t, RefEntry> ref_cache;
    size_t ref_cache_bytes = 0;
    std::atomic<size_t> ref_cache_budget{(size_t)4096 << 20};      // (isx_pipe_set_reference_budget may be called while the stager thread submits)
    std::thread stager;
    std::deque<StageJob> stage_q;
    std::condition_variable cv_stage;
    int64_t next_promise = 0;
    bool stage_stop = false;
};

// device -> plain host memory at link speed: pieces through the two pinned bounce buffers, emptied by the finisher's threads
static int bounce_d2h(isx_pipe *p, void *hdst, const void *dsrc, size_t bytes, hipStream_t st)
{
    if (!bytes) return ISX_OK;
    std::lock_guard<std::mutex> bounce_lk(p->bounce_mu);
    const size_t piece = p->bounce_bytes;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    auto issue = [&](size_t k) -> hipError_t {
        const size_t off = k * piece, len = std::min(piece, bytes - off);
        hipError_t e = isx_copy_to_host(p->bounce[k & 1], static_cast<const uint8_t *>(dsrc) + off, len, st);
        return e == hipSuccess ? hipEventRecord(p->bounce_ev[k & 1], st) : e;
    };
    HIP_TRY(issue(0));
    for (size_t k = 0; k < n_pieces; k++) {
        if (k + 1 < n_pieces) HIP_TRY(issue(k + 1));            // its buffer was emptied one step ago
        HIP_TRY(isx_wait_event(p->bounce_ev[k & 1]));
        const size_t off = k * piece, len = std::min(piece, bytes - off);
        const size_t sub = (size_t)1 << 20;
        const uint8_t *src = p->bounce[k & 1];
        uint8_t *dst = static_cast<uint8_t *>(hdst) + off;
        p->fin_pool->run((int)((len + sub - 1) / sub), [&](int t) {
            const size_t a = (size_t)t * sub;
            memcpy(dst + a, src + a, std::min(sub, len - a));
        });
    }
    return ISX_OK;
}

static void pipe_free(isx_pipe *p)
{
    if (!p) return;
    if (p->stager.joinable()) {                 // it stages what was queued, then leaves
        { std::lock_guard<std::mutex> lk(p->mu); p->stage_stop = true; }
        p->cv_stage.notify_all();
        p->stager.join();
    }
    if (p->finisher.joinable()) {               // it finishes what was submitted, then leaves
        { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
        p->cv_work.notify_all();
        p->finisher.join();
        if (p->finisher2.joinable()) p->finisher2.join();
        for (auto &t : p->more_finishers) if (t.joinable()) t.join();
    }
    (void)hipSetDevice(p->ctx->device);
    if (p->s_h2d) (void)isx_wait_stream(p->s_h2d);
    if (p->s_h2d2) (void)isx_wait_stream(p->s_h2d2);
    if (p->s_d2h) (void)isx_wait_stream(p->s_d2h);
    if (p->s_fin) (void)isx_wait_stream(p->s_fin);
    if (p->s_fin2) (void)isx_wait_stream(p->s_fin2);
    for (hipStream_t st : p->extra_fin) (void)isx_wait_stream(st);
    for (int i = 0; i < 2; i++) if (p->ctx->pstream[i]) (void)isx_wait_stream(p->ctx->pstream[i]);
    const double t_f0 = now_ms();
    double t_batch = 0, t_dev = 0, t_pin = 0;
    for (Slot &s : p->slots) {
        double t_x = now_ms();
        if (s.b) {
            isx_batch *b = s.b;             // the input arrays belong to the arena, not to the batch
            b->d_seg = nullptr; b->d_drec = nullptr;
            b->d_rec = nullptr; b->d_rec32 = nullptr; b->d_rec16 = nullptr; b->d_gbase = nullptr; b->d_pair = nullptr;
            b->d_pair_runs = nullptr; b->d_run_index = nullptr;
            b->d_gpos = nullptr; b->d_gpos16 = nullptr; b->d_cbase = nullptr; b->d_ref = nullptr; b->d_win = nullptr;
            b->d_bounds = nullptr;
            isx_batch_destroy(b);
        }
        t_batch += now_ms() - t_x; t_x = now_ms();
        if (s.sort_temp) isx_dev_free(s.sort_temp);
        if (s.d_gpos16) isx_dev_free(s.d_gpos16);
        if (s.d_in) isx_dev_free(s.d_in);
        if (s.d_runs) isx_dev_free(s.d_runs);
        t_dev += now_ms() - t_x; t_x = now_ms();
        if (s.h_in) isx_pin_free(s.h_in);
        host_block_free(s.h_runs, s.runs_pinned);
        for (hipEvent_t e : s.ev_ring) if (e) (void)hipEventDestroy(e);
        host_block_free(s.h_out, s.out_pinned);
        if (s.h_small) isx_pin_free(s.h_small);
        t_pin += now_ms() - t_x;
        for (hipEvent_t e : {s.ev_h2d0, s.ev_h2d1, s.ev_pass, s.ev_d2h0, s.ev_d2h1, s.ev_h2da, s.ev_h2db}) if (e) (void)hipEventDestroy(e);
    }
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe_destroy] device tables %.1f ms, device arena %.1f ms, pinned staging %.1f ms, total %.1f ms\n", t_batch, t_dev, t_pin, now_ms() - t_f0);
    for (auto &kv : p->ref_cache) { if (kv.second.d) isx_dev_free(kv.second.d); if (kv.second.ready) (void)hipEventDestroy(kv.second.ready); }
    for (int i = 0; i < 2; i++) {
        if (p->bounce[i]) isx_pin_free(p->bounce[i]);
        if (p->bounce_ev[i]) (void)hipEventDestroy(p->bounce_ev[i]);
    }
    if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
    if (p->s_h2d2) (void)hipStreamDestroy(p->s_h2d2);
    if (p->s_d2h) (void)hipStreamDestroy(p->s_d2h);
    if (p->s_fin) (void)hipStreamDestroy(p->s_fin);
    if (p->s_fin2) (void)hipStreamDestroy(p->s_fin2);
    for (hipStream_t st : p->extra_fin) (void)hipStreamDestroy(st);
    delete p;
}

// device tables of a slot, sized for the pipe's capacities (the allocation half of isx_batch_create)
static int slot_batch_create(isx_pipe *p, Slot &s, int index)
{
    isx_ctx *c = p->ctx;
    const isx_params *prm = &p->prm;
    const double t_s0 = now_ms();
    isx_batch *b = new isx_batch();
    s.b = b;
    b->ctx = c; b->prm = *prm; b->M = prm->n_mm_bins;
    b->ps = getenv("ISX_ONE_PASS_QUEUE") ? 0 : (index & 1);
    b->cap_pos = p->pp.max_pos; b->cap_obs = p->pp.max_obs; b->arena = true; b->ref_packed = 2;
    b->n_pos = p->pp.max_pos; b->n_obs = p->pp.max_obs;
    b->segs = p->segs; b->drec = p->drec;
    b->lean = p->pp.lean_output != 0 && !p->pp.want_counts;
    // mm profiling on in a read-level pipe: the levels go home sparse (mask + coverage bytes + lists) instead of as 32-byte entries;
    // a lean slot does not even write the entries
    b->lev_sparse = b->M > 1 && b->M <= 32 && p->segs && !p->pp.want_counts && !(prm->layout & ISX_LAYOUT_MM_ENTRIES);
    // four finishers polling for 2 ms each would take the whole CPU share of a rank that has 2-4 host threads to itself
    b->spin_us = (p->pp.host_threads > 0 && p->pp.host_threads < 8) ? 200 : 2000;
    const bool dense = b->M == 1;
    batch_pick_block(b);
    const int64_t cap_pos = p->pp.max_pos;
    for (auto &e : b->ev) HIP_TRY(hipEventCreate(&e));
    for (auto &e : b->ev_sum) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_cursors), (CUR_N + 4) * sizeof(uint32_t)));
    b->d_flags = b->d_cursors + CUR_N;
    HIP_TRY(hipHostMalloc(&b->h_state, (CUR_N + 8) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(b->h_state, 0, (CUR_N + 8) * sizeof(uint32_t));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_host_state), b->h_state, 0));
    HIP_TRY(hipMemset(b->d_cursors, 0, (CUR_N + 4) * sizeof(uint32_t)));
    {
        std::vector<uint16_t> thr = build_thresholds(c->h_lut, c->fallback, prm->min_freq);
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_thr), thr.size() * sizeof(uint16_t)));
        HIP_TRY(hipMemcpy(b->d_thr, thr.data(), thr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    const uint64_t npm = (uint64_t)cap_pos * b->M;
    const uint64_t cap_obs = (uint64_t)std::max<int64_t>(p->pp.max_obs, 1);
    if (dense) {
        // the per-base count table is only kept when the caller wants it back (--store_everything): otherwise 16 B/pos less to write
        if (p->pp.want_counts) HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_counts), (size_t)cap_pos * sizeof(uint4)));
        else {
            // shallow batches go back as 1-byte coverage + sparse clonality list (see submit: sparse_out)
            b->cap_clon = (size_t)cap_pos / 2 + 65536;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_cov8), (size_t)cap_pos));
            if (p->pp.lean_output) {        // 4-bit coverage plane of shallow batches: the window of every 16-bit row (finish_slot)
                b->cap_cov_row_win = (size_t)cap_pos / 64 + 2;
                HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_cov_row_win), b->cap_cov_row_win * sizeof(uint32_t)));
            }
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_clon_list), b->cap_clon * sizeof(uint2)));
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_clon_sorted), b->cap_clon * sizeof(uint2)));
        }
        b->cap_sat = (size_t)cap_pos / 16 + 65536;
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_sat), b->cap_sat * sizeof(uint2)));
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_clon), (size_t)cap_pos * sizeof(float)));
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_clon_r), (size_t)cap_pos * sizeof(float)));
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_cov16), (size_t)cap_pos * sizeof(uint16_t)));
        if (prm->rarefied_coverage > 0) {
            b->cap_rare = p->cap_rare;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_rare), b->cap_rare * sizeof(uint2)));
        }
    } else if (b->lev_sparse) {
        b->lev_mask_bytes = b->M <= 8 ? 1 : (b->M <= 16 ? 2 : 4);
        b->cap_lev = (size_t)std::min<uint64_t>(std::min<uint64_t>(npm, 0xFFFFFFF0ull), std::max<uint64_t>((uint64_t)cap_pos * (uint64_t)std::min(b->M, 4), 1u << 20));
        HIP_TRY(isx_dev_malloc(&b->d_lev_mask, (size_t)cap_pos * b->lev_mask_bytes + 64));
        HIP_TRY(isx_dev_malloc(&b->d_lev_cov, b->cap_lev * 2 + 64));
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_lev_win_off), ((size_t)cap_pos / 64 + 2) * sizeof(uint32_t)));
        if (!b->lean) {                     // a plain slot keeps the entries too (flat, indexed by level): device summaries, isx_batch_fetch_entries
            b->cap_entries = b->cap_lev;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_entries), b->cap_entries * sizeof(isx_entry)));
        }
        b->cap_clon = (size_t)cap_pos / 4 + 65536;
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_clon_list), b->cap_clon * sizeof(uint2)));
        b->cap_sat = (size_t)cap_pos / 16 + 65536;
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_sat), b->cap_sat * sizeof(uint2)));
        if (prm->rarefied_coverage > 0) {
            b->cap_rare = p->cap_rare;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_rare), b->cap_rare * sizeof(uint2)));
        }
    } else {
        b->slab_region = (size_t)(cap_pos + 2 * b->block) * (size_t)std::min(b->M, 4);
        const size_t ovf = b->M <= 4 ? 16 : (size_t)std::max<uint64_t>(1u << 20, std::min<uint64_t>(cap_obs, npm) / 4);
        if (b->slab_region + ovf >= 0xFFFFFFFFull) { isx_set_error("mm path: more than 2^32 entry slots in one batch"); return ISX_ERR_ARG; }
        b->cap_entries = b->slab_region + ovf;
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_entries), b->cap_entries * sizeof(isx_entry)));
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_win_nent), ((size_t)cap_pos / 64 + 2) * sizeof(uint32_t)));
    }
    b->cap_snv = (size_t)std::min<uint64_t>(npm, std::max<uint64_t>((uint64_t)cap_pos / 2, 1u << 20));
    b->cap_sites = (size_t)std::min<uint64_t>((uint64_t)cap_pos, std::max<uint64_t>((uint64_t)cap_pos / 4, 1u << 20));
    b->cap_ao = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(cap_obs, std::max<uint64_t>(cap_obs / 4, 1u << 20)));
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_snv), b->cap_snv * sizeof(isx_snv)));
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_sites), b->cap_sites * sizeof(isx_site)));
    if (prm->enable_linkage) HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_ao), b->cap_ao * sizeof(isx_ao)));
    if (!dense) {
        b->cap_slev = b->cap_sites * (size_t)std::min(b->M, 8);
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_slev), b->cap_slev * sizeof(isx_slev)));
    }
    // input arena: one layout for the pinned staging block and its device twin
    size_t o = 0;
    s.off_bounds = o; o = up(o + (size_t)(p->pp.max_splits + 1) * sizeof(int64_t));
    s.off_win = o; o = up(o + ((size_t)cap_pos / 64 + 2) * sizeof(uint2));
    s.off_ref = o; o = up(o + ref2_bytes(cap_pos) + refn_bytes(cap_pos));        // 2-bit plane | non-ACGT bit plane
    s.off_gbase = o; o = up(o + ((size_t)(p->cap_rec / p->G) + ISX_TAIL_GROUPS) * sizeof(uint32_t));
    s.off_ridx = o; if (prm->enable_linkage && !p->segs) o = up(o + ((size_t)p->cap_rec / ISX_CHUNK + 2) * sizeof(uint32_t));
    s.off_pairs = o; if (prm->enable_linkage && p->segs && !p->drec) o = up(o + (size_t)p->cap_rec * sizeof(uint32_t));     // (reference-delta records carry the ids themselves)
    s.off_rec = o;
    s.in_bytes = up(o + (size_t)p->cap_rec * p->rb + ISX_TAIL_BYTES);
    const size_t host_bytes = p->ring_half ? up(o + 2 * (size_t)p->ring_half * p->rb) : s.in_bytes;
    const double t_a0 = now_ms();
    HIP_TRY(isx_pin_malloc(reinterpret_cast<void **>(&s.h_in), host_bytes));
    const double t_a1 = now_ms();
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&s.d_in), s.in_bytes));
    if (getenv("ISX_PIPE_TIMING"))
        fprintf(stderr, "[isx_pipe_create] slot %d: device tables %.1f ms, pinned input %.1f MB %.1f ms, device arena %.1f MB %.1f ms\n", index,
                t_a0 - t_s0, host_bytes / 1e6, t_a1 - t_a0, s.in_bytes / 1e6, now_ms() - t_a1);
    if (prm->enable_linkage && !p->segs) {
        // a read pair's records are consecutive: runs of tens to hundreds of records
        s.cap_runs = (size_t)p->cap_rec / 64 + 4096;
        s.runs_pinned = p->pp.depth > 1;
        { const int hrc = host_block_alloc(reinterpret_cast<void **>(&s.h_runs), s.cap_runs * sizeof(isxenc::PairRun), s.runs_pinned); if (hrc != ISX_OK) return hrc; }
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&s.d_runs), s.cap_runs * sizeof(uint2)));
    }
    if (p->ring_half) for (hipEvent_t &e : s.ev_ring) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (prm->enable_linkage && p->rb == 4) HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&s.d_gpos16), (size_t)p->cap_rec * sizeof(uint16_t)));
    // result blocks: a small pinned one (first SNV rows, first clonTR entries) and the position-sized arrays
    o = 0;
    s.o_snv = o; o = up(o + p->snv_prefix * sizeof(isx_snv));
    s.o_rare = o; if (dense && prm->rarefied_coverage > 0) o = up(o + p->rare_prefix * sizeof(isx_rare));
    s.small_bytes = o;
    HIP_TRY(isx_pin_malloc(reinterpret_cast<void **>(&s.h_small), std::max<size_t>(s.small_bytes, 1)));
    o = 0;
    if (dense) {
        s.o_cov16 = o; o = up(o + (size_t)cap_pos * 2);
        s.o_clon = o; o = up(o + (size_t)cap_pos * 4);
        if (p->pp.want_counts) {
            s.o_counts = o; o = up(o + (size_t)cap_pos * 16);
            s.o_clonr = o; if (prm->rarefied_coverage > 0) o = up(o + (size_t)cap_pos * 4);
        }
    }
    if (b->lev_sparse) {
        s.o_lmask = o; o = up(o + (size_t)cap_pos * b->lev_mask_bytes);
        s.o_lwin = o; o = up(o + ((size_t)cap_pos / 64 + 2) * sizeof(uint32_t));
        s.lcov_room = (size_t)cap_pos * (size_t)std::min(b->M, 4);
        s.o_lcov = o; o = up(o + s.lcov_room);
        s.lclon_room = (size_t)cap_pos;                                 // (cap_pos / 8 list entries)
        s.o_lclon = o; o = up(o + s.lclon_room);
        s.lrare_room = prm->rarefied_coverage > 0 ? (size_t)cap_pos / 2 : 0;
        s.o_lrare = o; o = up(o + s.lrare_room);
    }
    s.out_bytes = o;
    const double t_o0 = now_ms();
    // Pinning (then unpinning) hundreds of MB costs more than moving them: 144 MB pinned = ~35 ms + ~30 ms to free.  A large
    // block lives in plain memory and the finisher moves the arrays there in pieces through two small pinned bounce buffers
    // (a hipMemcpyAsync straight into pageable memory crawls at < 2 GB/s and holds up every hipMalloc issued meanwhile).
    // A deep pipe (a long stream of batches) amortises the pinning and keeps its copy-out fully asynchronous.
    s.out_pinned = s.out_bytes <= ((size_t)64 << 20) || p->pp.depth > 2;
    { const int hrc = host_block_alloc(reinterpret_cast<void **>(&s.h_out), s.out_bytes, s.out_pinned); if (hrc != ISX_OK) return hrc; }
    if (getenv("ISX_PIPE_TIMING")) fprintf(stderr, "[isx_pipe_create] slot %d: %s results %.1f MB %.1f ms\n", index, s.out_pinned ? "pinned" : "pageable", s.out_bytes / 1e6, now_ms() - t_o0);
    for (hipEvent_t *e : {&s.ev_h2d0, &s.ev_h2d1, &s.ev_pass, &s.ev_d2h0, &s.ev_d2h1, &s.ev_h2da, &s.ev_h2db}) HIP_TRY(hipEventCreate(e));
    const size_t n_chunks = (size_t)(p->cap_rec / (p->segs ? (int64_t)p->G : (int64_t)ISX_CHUNK)) + 2;
    s.cmin.resize(n_chunks); s.cmax.resize(n_chunks); s.cany.resize(n_chunks);
    return ISX_OK;
}

// everything between "the batch's copy-out has landed" and "its tables can be handed to the caller"
static int finish_slot(isx_pipe *p, Slot &s, hipStream_t sfin)
{
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    const bool dense = b->M == 1;
    const double t0 = now_ms();
    HIP_TRY(isx_wait_event(s.ev_d2h1));
    s.finish_wait_ms = (float)(now_ms() - t0);
    const double t_c0 = now_ms();
    double t_fin = 0, t_rare = 0;
    hipStream_t ps = c->pstream[b->ps];
    bool redo = false;
    for (int attempt = 0;; attempt++) {
        uint32_t cf = 0;
        int rc = finish_pass_sizes(b, &cf, sfin);     // sizes from the published cursors (the linkage stages follow the fetches below: one wait for both)
        if (rc != ISX_OK) return rc;
        // a batch taken for shallow that has more positions beyond 255 than the exact-coverage list holds: again with 16 bits
        const bool cov8_overflow = !cf && dense && b->sparse_out && b->cov8_out && !b->nib_pass && (size_t)b->n_sat > b->cap_sat;
        // a 4-bit plane whose 16-bit rows would not fit the slot's coverage block (most windows beyond 15: not a shallow batch after all)
        const bool nib_overflow = !cf && dense && b->nib_pass &&
                                  ((size_t)b->n_pos / 2 + 128 + (size_t)b->n_cov_rows * (size_t)b->W * 2 > (size_t)b->n_pos * 2 ||
                                   ((size_t)b->n_cov_rows + 1) * (size_t)b->W > (size_t)(b->cap_pos ? b->cap_pos : b->n_pos));
        // a lean slot whose clonality list cannot be used (too many entries for the list / for a list to pay): again, with the dense array
        const bool clon_overflow = !cf && dense && b->sparse_out && b->lean && !b->clon_dense &&
                                   ((size_t)b->n_clon > b->cap_clon || (size_t)b->n_clon * 2 > (size_t)b->n_pos);
        // a lean slot's batch taken for shallow whose clonTR list cannot be used (deep after all): again, with the dense array
        const bool rare_overflow = !cf && dense && p->prm.rarefied_coverage > 0 && !b->rare_dense &&
                                   ((size_t)b->n_rare > p->cap_rare || (size_t)b->n_rare * 8 > (size_t)b->n_pos);
        // level-sparse mm slot: a list that outgrew its table (the kernel wrote what fitted; the cursors say what there was)
        const bool lev_overflow = !cf && b->lev_sparse && ((size_t)b->n_clon > b->cap_clon || (size_t)b->n_sat > b->cap_sat ||
                                                           (p->prm.rarefied_coverage > 0 && (size_t)b->n_rare > b->cap_rare));
        if (!cf && !cov8_overflow && !clon_overflow && !nib_overflow && !rare_overflow && !lev_overflow) break;
        if (attempt == 7) { isx_set_error("output tables still too small after 8 growth steps"); return ISX_ERR_CAPACITY; }
        // a table was too small for this batch: grow it and repeat the pass (the slot still holds its input)
        if (cov8_overflow) b->cov8_out = false;
        if (nib_overflow) b->nib_out = false;
        if (clon_overflow) b->clon_dense = true;
        if (rare_overflow) b->rare_dense = true;
        if (cf && (rc = batch_grow_tables(b, cf)) != ISX_OK) return rc;
        if (lev_overflow) {
            auto regrow_list = [&](uint2 **lp, size_t *cap, size_t need) -> int {
                if (need <= *cap) return ISX_OK;
                HIP_TRY(isx_wait_stream(ps));
                if (*lp) isx_dev_free(*lp);
                *lp = nullptr;
                const size_t want = need + need / 4 + 65536;
                HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(lp), want * sizeof(uint2)));
                *cap = want;
                return ISX_OK;
            };
            if ((rc = regrow_list(&b->d_clon_list, &b->cap_clon, b->n_clon)) != ISX_OK) return rc;
            if ((rc = regrow_list(&b->d_sat, &b->cap_sat, b->n_sat)) != ISX_OK) return rc;
            if (p->prm.rarefied_coverage > 0 && (rc = regrow_list(&b->d_rare, &b->cap_rare, b->n_rare)) != ISX_OK) return rc;
        }
        if (!dense && !b->lev_sparse) b->cap_ovf = b->cap_entries - (size_t)b->n_win * b->slab;
        std::lock_guard<std::mutex> lk(p->launch_mu);
        if (dense && p->prm.rarefied_coverage > 0 && b->rare_dense)
            HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->d_clon_r), 0x7FC00000, (size_t)b->n_pos, ps));
        if ((rc = launch_pass(b)) != ISX_OK) return rc;
        {   // published here, under the launch lock like a submit's pass: finish_pass must not touch the stream's publication state
            PileupArgs pa{};
            pa.cursors = b->d_cursors; pa.host_state = b->d_host_state;
            launch_publish_state(pa, b->epoch, ps);
            b->publish_enqueued = true;
            if (c->unpublished[b->ps] == b) c->unpublished[b->ps] = nullptr;
        }
        redo = true;
    }
    s.d2h_bytes += (int64_t)(std::min((size_t)b->sizes.n_snv, p->snv_prefix) * sizeof(isx_snv));
    if (dense && p->prm.rarefied_coverage > 0) s.d2h_bytes += (int64_t)(std::min((size_t)b->n_rare, p->rare_prefix) * sizeof(isx_rare));
    // device -> the slot's result block: through the bounce buffers into a plain block, a blocking copy into a pinned one
    // small tables into any host memory, waited for: through the calling thread's pinned scratch (isx_read_back), not the DMA queue
    auto pull = [&](void *hdst, const void *dsrc, size_t bytes) -> int {
        if (!bytes) return ISX_OK;
        HIP_TRY(isx_read_back(hdst, dsrc, bytes, sfin));
        HIP_TRY(isx_read_sync(sfin));
        return ISX_OK;
    };
    // the small read-backs from here on (window lists, exact-coverage rows, the linkage stages' state words + rows) leave in ONE copy launch, made
    // by the wait that delivers them: nothing writes their sources in between
    // (an exit with read-backs still pending -- an error between a read-back and its wait -- forgets them: their destinations may not outlive this call)
    struct ReadBatch { ReadBatch() { isx_read_batch(true); } ~ReadBatch() { isx_read_drop(); } } read_batch;
    auto fetch = [&](void *hdst, const void *dsrc, size_t bytes) -> int {
        if (!bytes) return ISX_OK;
        if (!s.out_pinned) return bounce_d2h(p, hdst, dsrc, bytes, sfin);
        HIP_TRY(isx_copy_to_host(hdst, dsrc, bytes, sfin));
        return ISX_OK;
    };
    if (dense) {
        int rc = ISX_OK;
        if (redo) HIP_TRY(isx_wait_stream(ps));
        s.cov8 = false; s.cov4 = false; s.clon_sparse = false; s.sat_complete = (size_t)b->n_sat <= b->cap_sat;
        if (b->sparse_out) {
            // coverage in 2 bytes, or 1 for a shallow batch (exact values of the few positions at 255 / 65535 or beyond in the
            // list below); clonality: a position that reaches min_cov has exactly 1.0 unless more than one base was observed there
            // -- only those exceptions travel, as a (position, value) list sorted by position on the device
            s.cov8 = b->cov8_out && !b->nib_pass;
            if (b->nib_pass) {
                // min(coverage, 15) of two positions a byte + the windows that hold anything beyond 15 as whole 16-bit rows
                const size_t nib_bytes = ((size_t)b->n_pos + 1) / 2, rows_bytes = (size_t)b->n_cov_rows * (size_t)b->W * 2;
                s.cov4 = true; s.cov_window = b->W;
                s.cov_rows_off = (nib_bytes + 63) & ~(size_t)63;
                rc = fetch(s.h_out + s.o_cov16, b->d_cov8, nib_bytes);
                if (rc == ISX_OK) rc = fetch(s.h_out + s.o_cov16 + s.cov_rows_off, b->d_cov16, rows_bytes);
                s.cov_row_win.resize(b->n_cov_rows);
                if (rc == ISX_OK && b->n_cov_rows && isx_read_back(s.cov_row_win.data(), b->d_cov_row_win, (size_t)b->n_cov_rows * sizeof(uint32_t), sfin) != hipSuccess) {
                    isx_set_error("isx_pipe finisher: read-back of the coverage rows' windows"); isx_read_drop(); rc = ISX_ERR_HIP;
                }
                s.d2h_bytes += (int64_t)(nib_bytes + rows_bytes + (size_t)b->n_cov_rows * 4);
            } else if (s.cov8) rc = fetch(s.h_out + s.o_cov16, b->d_cov8, (size_t)b->n_pos);
            else rc = fetch(s.h_out + s.o_cov16, b->d_cov16, (size_t)b->n_pos * 2);
            if (rc != ISX_OK) return rc;
            if (!s.cov4) s.d2h_bytes += (int64_t)b->n_pos * (s.cov8 ? 1 : 2);
            const size_t n_clon = b->n_clon;
            if (n_clon <= b->cap_clon && n_clon * 2 <= (size_t)b->n_pos) {
                // (b->ordered: k_win_gather already wrote the list in position order behind the pileup kernel)
                if (!b->ordered && (rc = sort_pairs_by_position(sfin, b->d_clon_list, b->d_clon_sorted, n_clon, &s.sort_temp, &s.sort_temp_bytes)) != ISX_OK) return rc;
                if ((rc = fetch(s.h_out + s.o_clon, b->d_clon_sorted, n_clon * sizeof(isx_rare))) != ISX_OK) return rc;
                s.clon_sparse = true;
                s.d2h_bytes += (int64_t)(n_clon * sizeof(isx_rare));
            } else {
                if ((rc = fetch(s.h_out + s.o_clon, b->d_clon, (size_t)b->n_pos * 4)) != ISX_OK) return rc;
                s.d2h_bytes += (int64_t)b->n_pos * 4;
            }
        } else if (!s.out_pinned || redo) {     // a plain result block, or tables that predate the repeated pass
            if ((rc = fetch(s.h_out + s.o_cov16, b->d_cov16, (size_t)b->n_pos * 2)) != ISX_OK) return rc;
            if ((rc = fetch(s.h_out + s.o_clon, b->d_clon, (size_t)b->n_pos * 4)) != ISX_OK) return rc;
            if (p->pp.want_counts) {
                if ((rc = fetch(s.h_out + s.o_counts, b->d_counts, (size_t)b->n_pos * 16)) != ISX_OK) return rc;
                if (p->prm.rarefied_coverage > 0 && (rc = fetch(s.h_out + s.o_clonr, b->d_clon_r, (size_t)b->n_pos * 4)) != ISX_OK) return rc;
            }
        }
        // exact coverage of the saturated positions (a handful; none at all for most batches)
        s.sat_rows.resize(s.sat_complete ? (size_t)b->n_sat : 0);
        if (!s.sat_rows.empty()) HIP_TRY(isx_read_back(s.sat_rows.data(), b->d_sat, s.sat_rows.size() * sizeof(isx_sat), sfin));
        // (no wait here: the linkage stages below end with the one wait that also covers these copies)
    }
    if (b->lev_sparse) {
        // mm profiling on: the level mask, the windows' first level indices, one coverage byte (or two) per present level, the lists
        if (redo) HIP_TRY(isx_wait_stream(ps));
        int rc = ISX_OK;
        const size_t n_lev = (size_t)b->sizes.n_entries, cov_bytes = n_lev * (size_t)b->lev_cov_bytes;
        const size_t clon_bytes = (size_t)b->n_clon * sizeof(isx_rare), rare_bytes = p->prm.rarefied_coverage > 0 ? (size_t)b->n_rare * sizeof(isx_rare) : 0;
        // a small batch's tables (a C2 batch: 5 + 12 MB) leave by copy kernel, not by DMA: see isx_copy_to_host_route
        static const size_t lev_kernel_max = [] { const char *e = getenv("ISX_LEV_KERNEL_MAX"); return e ? (size_t)atoll(e) : (size_t)32 << 20; }();
        const bool lev_by_kernel = s.out_pinned && (size_t)b->n_pos * b->lev_mask_bytes + cov_bytes + clon_bytes + rare_bytes <= lev_kernel_max;
        std::vector<isx_copy_job> jobs;                             // (lev_by_kernel: all of the batch's tables in one launch)
        auto fetch_lev = [&](void *hdst, const void *dsrc, size_t bytes) -> int {
            if (!lev_by_kernel || ((reinterpret_cast<uintptr_t>(hdst) | reinterpret_cast<uintptr_t>(dsrc)) & 15)) return fetch(hdst, dsrc, bytes);
            jobs.push_back(isx_copy_job{hdst, dsrc, bytes});
            return ISX_OK;
        };
        if ((rc = fetch_lev(s.h_out + s.o_lmask, b->d_lev_mask, (size_t)b->n_pos * b->lev_mask_bytes)) != ISX_OK) return rc;
        if ((rc = fetch_lev(s.h_out + s.o_lwin, b->d_lev_win_off, (size_t)b->n_win * sizeof(uint32_t))) != ISX_OK) return rc;
        // what outgrows its pinned room (a deep sample's clonTR list, a coverage stream of more than four levels a position): plain vectors
        auto fetch_or_big = [&](size_t off, size_t room, auto &big, const void *dsrc, size_t bytes) -> int {
            big.clear();
            if (bytes <= room) return fetch_lev(s.h_out + off, dsrc, bytes);
            big.resize((bytes + sizeof(big[0]) - 1) / sizeof(big[0]));
            if (p->bounce[0]) return bounce_d2h(p, big.data(), dsrc, bytes, sfin);
            HIP_TRY(hipMemcpy(big.data(), dsrc, bytes, hipMemcpyDeviceToHost));
            return ISX_OK;
        };
        if ((rc = fetch_or_big(s.o_lcov, s.lcov_room, s.lcov_big, b->d_lev_cov, cov_bytes)) != ISX_OK) return rc;
        if ((rc = fetch_or_big(s.o_lclon, s.lclon_room, s.lclon_big, b->d_clon_list, clon_bytes)) != ISX_OK) return rc;
        if (rare_bytes && (rc = fetch_or_big(s.o_lrare, s.lrare_room, s.lrare_big, b->d_rare, rare_bytes)) != ISX_OK) return rc;
        if (!jobs.empty()) HIP_TRY(isx_copy_multi_to_host(jobs.data(), (int)jobs.size(), sfin));
        s.sat_rows.resize((size_t)b->n_sat);
        if (!s.sat_rows.empty()) HIP_TRY(isx_read_back(s.sat_rows.data(), b->d_sat, s.sat_rows.size() * sizeof(isx_sat), sfin));
        // (no wait here: the linkage stages below end with the one wait that also covers these copies)
        s.d2h_bytes += (int64_t)((size_t)b->n_pos * b->lev_mask_bytes + (size_t)b->n_win * 4 + cov_bytes + clon_bytes + rare_bytes + s.sat_rows.size() * sizeof(isx_sat));
    }
    // the linkage stages, on this finisher's queue behind the copies above; the bucket chain's one read-back is the wait for all of it
    {
        const int lrc = finish_pass_link(b, sfin);
        if (lrc != ISX_OK) return lrc;
        if (!(p->prm.enable_linkage && b->link_chain == 3)) HIP_TRY(isx_read_sync(sfin));
    }
    t_fin = now_ms();
    if (dense && p->prm.rarefied_coverage > 0) {        // the sparse clonTR table, ascending positions
        const size_t n_rare = b->n_rare;
        isx_rare *rr = reinterpret_cast<isx_rare *>(s.h_small + s.o_rare);
        s.rare_big.clear();
        s.rare_dense = false;
        if (n_rare > p->cap_rare || n_rare * 8 > (size_t)b->n_pos) {
            // a deep sample: most positions reach the rarefied coverage, so the 8-byte list is no smaller than
            // the 4-byte dense array (and would need sorting) -- or the device list overflowed: hand back the array
            // (without want_counts there is no pinned room for it: deep samples are the exception, pinning 4 more bytes per
            // position for every pipe is not worth it)
            if (!p->pp.want_counts) {
                if (s.clonr_big.size() < (size_t)b->n_pos) s.clonr_big.resize((size_t)b->n_pos);
                if (p->bounce[0]) { const int rc = bounce_d2h(p, s.clonr_big.data(), b->d_clon_r, (size_t)b->n_pos * 4, sfin); if (rc != ISX_OK) return rc; }
                else HIP_TRY(hipMemcpy(s.clonr_big.data(), b->d_clon_r, (size_t)b->n_pos * 4, hipMemcpyDeviceToHost));
            } else if (redo && s.out_pinned)
                HIP_TRY(hipMemcpy(s.h_out + s.o_clonr, b->d_clon_r, (size_t)b->n_pos * 4, hipMemcpyDeviceToHost));
            s.rare_dense = true;
        } else {
            if (n_rare > p->rare_prefix || redo) {
                if (n_rare > p->rare_prefix) { s.rare_big.resize(n_rare); rr = s.rare_big.data(); }
                { const int prc = pull(rr, b->d_rare, n_rare * sizeof(isx_rare)); if (prc != ISX_OK) return prc; }
            }
            if (!b->ordered) std::sort(rr, rr + n_rare, [](const isx_rare &x, const isx_rare &y) { return x.gpos < y.gpos; });
        }
    }
    t_rare = now_ms();
    const size_t n_snv = (size_t)b->sizes.n_snv;
    isx_snv *rows = reinterpret_cast<isx_snv *>(s.h_small + s.o_snv);
    if (n_snv > p->snv_prefix || redo) {
        if (n_snv > p->snv_prefix) { s.snv_big.resize(n_snv); rows = s.snv_big.data(); }
        { const int prc = pull(rows, b->d_snv, n_snv * sizeof(isx_snv)); if (prc != ISX_OK) return prc; }
    }
    if (!b->ordered) std::sort(rows, rows + n_snv, [](const isx_snv &x, const isx_snv &y) { return x.gpos != y.gpos ? x.gpos < y.gpos : x.mm < y.mm; });
    if (p->prm.enable_linkage) {
        // the LD rows too: a blocking copy issued by the caller would queue behind the next batches' large transfers
        s.ld_rows.resize((size_t)b->sizes.n_ld);
        if (!s.ld_rows.empty() && b->ld_host && s.ld_rows.size() <= b->n_ld_host) memcpy(s.ld_rows.data(), b->ld_host, s.ld_rows.size() * sizeof(isx_ld));   // came home with the chain's state words
        else if (!s.ld_rows.empty()) { const int rc = pull(s.ld_rows.data(), b->L.ld.p, s.ld_rows.size() * sizeof(isx_ld)); if (rc != ISX_OK) return rc; }
    }
    s.rows_checksum = bytes_checksum(s.ld_rows.data(), p->prm.enable_linkage ? s.ld_rows.size() * sizeof(isx_ld) : 0, bytes_checksum(rows, n_snv * sizeof(isx_snv), 0));
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe finisher] wait %.2f ms, finish (sizes, linkage) %.2f ms [device: sites %.2f allele %.2f group %.2f incr %.2f ld %.2f; %lld ao, %lld incr, %lld ld], clonTR list (%u) %.2f ms, snv rows (%zu) %.2f ms\n",
                s.finish_wait_ms, t_fin - t_c0, b->tim.sites_ms, b->tim.allele_ms, b->tim.group_ms, b->tim.incr_ms, b->tim.ld_ms,
                (long long)b->sizes.n_allele_obs, (long long)b->sizes.n_increments, (long long)b->sizes.n_ld, b->n_rare, t_rare - t_fin, n_snv, now_ms() - t_rare);
    return ISX_OK;
}

static void finisher_main(isx_pipe *p, hipStream_t sfin)
{
    (void)hipSetDevice(p->ctx->device);
    for (;;) {
        int64_t ticket;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || !p->work.empty(); });
            if (p->work.empty()) return;        // stop, nothing left
            ticket = p->work.front();
            p->work.pop_front();
        }
        Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
        const int rc = finish_slot(p, s, sfin);
        std::string err = rc == ISX_OK ? std::string() : std::string(isx_last_error());
        BamBatch *dead = nullptr;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            s.rc = rc; s.err.swap(err);
            dead = s.dead_batch; s.dead_batch = nullptr;
            s.state = 2;
        }
        p->cv_done.notify_all();
        // the front end's batch goes back to its handle outside the pipe's lock (retiring may free older batches: gigabytes);
        // the handle itself waits for its batches before it goes (isx_bam_close), so the caller may close it right after collect
        if (dead) bam_batch_retire(dead);
    }
}

static int submit_segs_common(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                              isxenc::SegJob &J, int64_t *ticket, const isx_ref_planes *rp = nullptr);

static void stager_main(isx_pipe *p)
{
    for (;;) {
        isx_pipe::StageJob job;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_stage.wait(lk, [&] { return p->stage_stop || !p->stage_q.empty(); });
            if (p->stage_q.empty()) return;
            job = std::move(p->stage_q.front());        // it stays "queued" (next_ticket == its ticket) until it is staged
            p->stage_q.pop_front();
        }
        isxenc::SegJob J;
        if (job.planes) { J.in2 = job.reads; J.n_seg = job.reads.n_seg; }
        else { J.in = job.segs; J.n_seg = job.segs.n_seg; }
        int64_t got = -1;
        const int rc = submit_segs_common(p, job.n_pos, job.ref, (int32_t)job.bounds.size() - 1, job.bounds.data(), J, &got, job.planes ? &job.rp : nullptr);
        if (rc != ISX_OK) {                              // the batch's ticket carries the error to isx_pipe_collect
            Slot &s = p->slots[(size_t)(job.ticket % (int64_t)p->slots.size())];
            std::string err(isx_last_error());
            std::lock_guard<std::mutex> lk(p->mu);
            s.ticket = job.ticket; s.rc = rc; s.err.swap(err); s.state = 2;
            p->next_ticket = job.ticket + 1;
        }
        p->cv_done.notify_all();
    }
}

// a synchronous submit on a pipe with a stager: what was queued before goes first
static void drain_stager(isx_pipe *p)
{
    if (!p->stager.joinable()) return;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->next_ticket == p->next_promise; });
}

extern "C" {

int isx_pipe_create(isx_ctx *c, const isx_params *prm, const isx_pipe_params *pp, isx_pipe **out)
{
    if (!c || !prm || !pp || !out) { isx_set_error("isx_pipe_create: bad argument"); return ISX_ERR_ARG; }
    *out = nullptr;
    if (!c->d_lut) { isx_set_error("isx_pipe_create: call isx_set_null_model first"); return ISX_ERR_STATE; }
    if (prm->n_mm_bins < 1 || prm->n_mm_bins > 128) { isx_set_error("n_mm_bins must be in [1, 128]"); return ISX_ERR_ARG; }
    if (pp->max_pos <= 0 || pp->max_obs < 0 || pp->max_splits <= 0 || pp->depth < 1 || pp->depth > 64 || pp->max_segs < 0) {
        isx_set_error("isx_pipe_create: max_pos > 0, max_obs >= 0, max_segs >= 0, max_splits > 0, 1 <= depth <= 64");
        return ISX_ERR_ARG;
    }
    if (pp->max_segs > 0 && prm->linkage_mode == 2) { isx_set_error("isx_pipe_create: the dense MFMA linkage path takes observation batches only"); return ISX_ERR_ARG; }
    if (pp->max_pos >= (int64_t)0xFFFF0000ll) { isx_set_error("flat position space must be < 2^32 - 65536"); return ISX_ERR_ARG; }
    if (prm->window && (prm->window < 64 || (prm->window & 63) || prm->window > 8192)) { isx_set_error("window must be a multiple of 64 in [64, 8192]"); return ISX_ERR_ARG; }
    if (prm->layout & ISX_LAYOUT_WIDE_RECORDS) { isx_set_error("a pipe streams 2- / 4-byte records only"); return ISX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    isx_pipe *p = new isx_pipe();
    p->ctx = c; p->prm = *prm; p->pp = *pp;
    p->segs = pp->max_segs > 0;
    p->drec = p->segs && !(prm->layout & ISX_LAYOUT_SEG64_RECORDS) && (prm->n_mm_bins == 1 || (prm->layout & ISX_LAYOUT_MM_DELTA_RECORDS));     // (mm profiling on: opt-in)
    p->rb = p->segs ? (p->drec ? 32 : 64) : ((prm->n_mm_bins == 1 && !(prm->layout & ISX_LAYOUT_NO_SHORT_RECORDS)) ? 2 : 4);
    p->G = p->segs ? (p->drec ? ISX_DREC_GROUP : ISX_SEG_GROUP) : (p->rb == 2 ? ISX_GROUP16 : ISX_GROUP);
    const double js = pp->jump_slack > 0 ? pp->jump_slack : 0.25;
    if (p->segs) {
        // groups of 16 records; a group is closed early where the stream jumps >= 65536 positions (at most once per 64 Ki
        // positions and per scaffold in a position-sorted stream) and at the end of every encoder task
        if (p->pp.max_obs == 0) p->pp.max_obs = pp->max_segs * ISX_SEG_BASES;
        // (reference-delta records: + spare groups per encoder task for the pieces of segments with many differences -- two per
        // task of 4096 up front, js more of the stream when the data asks for it; a batch beyond that is ISX_ERR_CAPACITY)
        const uint64_t G = p->G;
        const uint64_t groups = (uint64_t)(pp->max_segs + G - 1) / G + (uint64_t)pp->max_segs / 4096 * (p->drec ? 3 : 1) + (uint64_t)pp->max_pos / 65536 +
                                (uint64_t)pp->max_splits + 64 + (uint64_t)((double)pp->max_segs / G * js * (p->drec ? 1.0 : 0.25));
        p->cap_rec = (int64_t)(groups * G);
        if ((uint64_t)p->cap_rec >= (1ull << 26)) { delete p; isx_set_error("more than 2^26 segment records in one batch (4 GiB of records)"); return ISX_ERR_ARG; }
        // staging: the whole stream pinned while that is cheap (a C2 batch is 43 MB), otherwise waves through a ring of two
        // halves -- pinning costs ~0.2 s per GB and as much again to unpin, more than encoding and copying the records
        const size_t rec_bytes = (size_t)p->cap_rec * (size_t)p->rb;
        size_t ring = 0;
        if (pp->ring_kib > 0) ring = (size_t)pp->ring_kib << 10;
        else if (pp->ring_kib == 0 && rec_bytes > ((size_t)(pp->depth == 1 ? 96 : 256) << 20)) ring = (size_t)64 << 20;
        if (ring && ring < rec_bytes) {
            p->ring_half = (int64_t)(ring / 2 / (size_t)p->rb) / (int64_t)p->G * (int64_t)p->G;
            if (p->ring_half * p->rb < (4096 + 16 * ISX_SEG_GROUP) * 64) { delete p; isx_set_error("isx_pipe_create: ring_kib too small for a read-level pipe (at least 1024)"); return ISX_ERR_ARG; }
        }
    } else {
        const uint64_t want = (uint64_t)((double)pp->max_obs * (1.0 + js)) + 4 * ISX_PAD;
        p->cap_rec = (int64_t)((want + ISX_PAD - 1) / ISX_PAD * ISX_PAD);
        if ((uint64_t)p->cap_rec >= 0xFFFFFFFFull) { delete p; isx_set_error("more than 2^32 records in one batch"); return ISX_ERR_ARG; }
    }
    if (!p->segs) {   // staging: the whole stream in pinned memory, or -- pinning gigabytes costs about a second per 4 GB, more than
        // profiling them -- waves through a ring of two halves that the copy engine drains while the threads fill
        const size_t rec_bytes = (size_t)p->cap_rec * p->rb;
        size_t ring = 0;
        if (pp->ring_kib > 0) ring = (size_t)pp->ring_kib << 10;
        else if (pp->ring_kib == 0 && pp->depth == 1 && rec_bytes > ((size_t)128 << 20)) ring = (size_t)64 << 20;   // no other batch to overlap with
        else if (pp->ring_kib == 0 && rec_bytes > ((size_t)512 << 20)) ring = (size_t)256 << 20;
        if (ring && ring < rec_bytes) {
            p->ring_half = (int64_t)(ring / 2 / p->rb) / ISX_PAD * ISX_PAD;
            if (p->ring_half < 2 * ISX_PAD) { delete p; isx_set_error("isx_pipe_create: ring_kib too small (at least 32)"); return ISX_ERR_ARG; }
        }
    }
    p->snv_prefix = (size_t)std::min<int64_t>(std::max<int64_t>(pp->max_pos / 64, 1 << 16), 1 << 22);
    // clonTR list: the device list can hold every position; a sixteenth of that travels with every batch, the
    // rest only when a batch really has that many (deep samples)
    p->cap_rare = (size_t)std::max<int64_t>(pp->max_pos, 1 << 16);
    p->rare_prefix = (size_t)std::max<int64_t>(pp->max_pos / 16, 1 << 16);
    int nt = pp->host_threads;
    if (nt <= 0) {
        const int q = cgroup_cpus();
        nt = q > 0 ? q : (int)std::thread::hardware_concurrency();
        nt = std::max(1, std::min(nt, 32));
    }
    c->n_pipes++;
    const double t_c0 = now_ms();
    p->pool.reset(new isxenc::HostPool(nt, gpu_numa_node(c->device), pp->pin_threads != 0));
    const double t_c1 = now_ms();
    int rc = ISX_OK;
    hipError_t e;
    if ((e = isx_side_stream_create(c, &p->s_h2d)) != hipSuccess ||
        (e = isx_side_stream_create(c, &p->s_fin)) != hipSuccess ||
        (e = isx_side_stream_create(c, &p->s_fin2)) != hipSuccess ||
        (e = isx_side_stream_create(c, &p->s_d2h)) != hipSuccess) {
        isx_set_error(std::string("isx_pipe_create: ") + hipGetErrorString(e));
        pipe_free(p);
        return ISX_ERR_HIP;
    }
    p->slots.resize((size_t)pp->depth);
    {   // ISX_PIPE_H2D_STREAMS=2 (tuning aid): the odd slots copy in through a queue of their own
        const char *e2 = getenv("ISX_PIPE_H2D_STREAMS");
        if (e2 && atoi(e2) >= 2 && pp->depth >= 2 && isx_side_stream_create(c, &p->s_h2d2) != hipSuccess) { p->s_h2d2 = nullptr; (void)hipGetLastError(); }
    }
    for (int i = 0; i < pp->depth && rc == ISX_OK; i++) {
        rc = slot_batch_create(p, p->slots[(size_t)i], i);
        if ((i & 1) && p->s_h2d2) p->slots[(size_t)i].h2d = p->s_h2d2;
    }
    if (rc != ISX_OK) { pipe_free(p); return rc; }
    if (!p->slots[0].out_pinned) {
        p->bounce_bytes = (size_t)16 << 20;
        for (int i = 0; i < 2 && rc == ISX_OK; i++) {
            if (isx_pin_malloc(reinterpret_cast<void **>(&p->bounce[i]), p->bounce_bytes) != hipSuccess ||
                hipEventCreateWithFlags(&p->bounce_ev[i], hipEventDisableTiming) != hipSuccess) rc = ISX_ERR_HIP;
        }
        if (rc != ISX_OK) { isx_set_error("isx_pipe_create: bounce buffers"); pipe_free(p); return rc; }
        p->fin_pool.reset(new isxenc::HostPool(std::max(1, std::min(nt, 6)), -1, false));
    }
    p->finisher = std::thread(finisher_main, p, p->s_fin);
    if (pp->depth >= 2) p->finisher2 = std::thread(finisher_main, p, p->s_fin2);
    {
        // a deep pipe whose copy-in no longer bounds the stream (reference-delta records halved it) is paced by how many batches
        // are being finished at a time -- sizes, linkage chain, small sorts: chains of short kernels and host syncs
        int want = pp->depth >= 4 ? 4 : 2;
        if (const char *e = getenv("ISX_PIPE_FINISHERS")) want = std::max(1, atoi(e));
        want = std::min(want, pp->depth);
        for (int i = 2; i < want; i++) {
            hipStream_t st = nullptr;
            if (isx_side_stream_create(c, &st) != hipSuccess) break;
            p->extra_fin.push_back(st);
            p->more_finishers.emplace_back(finisher_main, p, st);
        }
    }
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe_create] thread pool %.1f ms, %d slot(s) %.1f ms\n", t_c1 - t_c0, pp->depth, now_ms() - t_c1);
    if (pp->stage_async && p->segs) p->stager = std::thread(stager_main, p);
    *out = p;
    return ISX_OK;
}

void isx_pipe_destroy(isx_pipe *p) { pipe_free(p); }

// the pass queue and the copy-out queue of a slot whose copy-in has been enqueued (s.ev_h2d1 recorded), then the
// hand-over to the finisher
static int enqueue_pass_impl(isx_pipe *p, Slot &s, int64_t n_pos, int64_t *ticket);

// the copy-in queue already holds this batch: when the pass cannot be enqueued behind it the slot stays free, so nothing of
// the batch may still be in flight on its arena when the error is returned
static int enqueue_pass(isx_pipe *p, Slot &s, int64_t n_pos, int64_t *ticket)
{
    const int rc = enqueue_pass_impl(p, s, n_pos, ticket);
    if (rc != ISX_OK) {
        const std::string why(isx_last_error());
        (void)isx_wait_stream(H2D(p, s));
        (void)isx_wait_stream(p->ctx->pstream[s.b->ps]);
        (void)isx_wait_stream(p->s_d2h);
        isx_set_error(why);
    }
    return rc;
}

static int enqueue_pass_impl(isx_pipe *p, Slot &s, int64_t n_pos, int64_t *ticket)
{
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    const bool dense = b->M == 1, linkage = p->prm.enable_linkage != 0;
    hipStream_t ps = c->pstream[b->ps];
    int rc = ISX_OK;
    // ---- pass queue ----
    std::unique_lock<std::mutex> launch_lk(p->launch_mu);
    HIP_TRY(hipStreamWaitEvent(ps, s.ev_h2d1, 0));
    if (dense && p->prm.rarefied_coverage > 0 && b->rare_dense)     // (a lean slot's shallow batch hands back the clonTR list alone)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->d_clon_r), 0x7FC00000, (size_t)n_pos, ps));
    if (linkage && p->rb == 4)
        launch_extract_gpos(nullptr, b->d_rec32, b->d_gbase, nullptr, b->d_gpos16, b->d_gbase, ISX_GROUP, b->n_rec, ps);
    if ((rc = launch_pass(b)) != ISX_OK) return rc;
    {   // cursors -> mapped host state right behind the kernel: the copy-out below needs no host round trip
        PileupArgs pa{};
        pa.cursors = b->d_cursors; pa.host_state = b->d_host_state;
        launch_publish_state(pa, b->epoch, ps);
        b->publish_enqueued = true;
        if (c->unpublished[b->ps] == b) c->unpublished[b->ps] = nullptr;
    }
    HIP_TRY(hipEventRecord(s.ev_pass, ps));

    // ---- copy-out queue ----
    HIP_TRY(hipStreamWaitEvent(p->s_d2h, s.ev_pass, 0));
    HIP_TRY(hipEventRecord(s.ev_d2h0, p->s_d2h));
    const size_t snv_rows = std::min(p->snv_prefix, b->cap_snv);
    // the SNV rows (and clonTR entries) this pass will have produced: counted on the device, no fixed-size prefix
    HIP_TRY(isx_copy_rows_to_host(s.h_small + s.o_snv, b->d_snv, b->d_cursors + CUR_SNV, b->base[CUR_SNV], (uint32_t)sizeof(isx_snv), snv_rows, p->s_d2h));
    s.d2h_bytes = 0;                        // (finish_slot adds what really travelled)
    if (dense) {
        if (p->prm.rarefied_coverage > 0) {
            const size_t n = std::min(p->rare_prefix, std::min(p->cap_rare, (size_t)n_pos));
            HIP_TRY(isx_copy_rows_to_host(s.h_small + s.o_rare, b->d_rare, b->d_cursors + CUR_RARE, b->base[CUR_RARE], (uint32_t)sizeof(isx_rare), n, p->s_d2h));
        }
        if (!b->sparse_out) s.d2h_bytes += (int64_t)n_pos * 6;
    }
    if (dense && s.out_pinned && !b->sparse_out) {     // (a plain result block / a shallow batch's tables are brought in by the finisher)
        HIP_TRY(isx_copy_to_host(s.h_out + s.o_cov16, b->d_cov16, (size_t)n_pos * 2, p->s_d2h));
        HIP_TRY(isx_copy_to_host(s.h_out + s.o_clon, b->d_clon, (size_t)n_pos * 4, p->s_d2h));
        if (p->pp.want_counts) {
            HIP_TRY(isx_copy_to_host(s.h_out + s.o_counts, b->d_counts, (size_t)n_pos * 16, p->s_d2h));
            s.d2h_bytes += (int64_t)n_pos * 16;
            if (p->prm.rarefied_coverage > 0) {
                HIP_TRY(isx_copy_to_host(s.h_out + s.o_clonr, b->d_clon_r, (size_t)n_pos * 4, p->s_d2h));
                s.d2h_bytes += (int64_t)n_pos * 4;
            }
        }
    }
    HIP_TRY(hipEventRecord(s.ev_d2h1, p->s_d2h));
    launch_lk.unlock();
    {
        std::lock_guard<std::mutex> lk(p->mu);
        s.ticket = p->next_ticket++;
        if (p->next_promise < p->next_ticket) p->next_promise = p->next_ticket;
        s.state = 1; s.rc = ISX_OK;
        *ticket = s.ticket;
        p->work.push_back(s.ticket);
    }
    p->cv_work.notify_one();
    return ISX_OK;
}

// the common part of a submit: `J` arrives with its input side set (arrays, or a producer), everything else happens here
static int submit_common(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                         int64_t n_obs, isxenc::EncodeJob &J, int64_t *ticket)
{
    if (n_pos > p->pp.max_pos || n_obs > p->pp.max_obs || n_splits > p->pp.max_splits) {
        isx_set_error("isx_pipe_submit: batch larger than the pipe was created for");
        return ISX_ERR_CAPACITY;
    }
    if (split_bounds[0] != 0 || split_bounds[n_splits] != n_pos) { isx_set_error("split_bounds must span [0, n_pos]"); return ISX_ERR_ARG; }
    for (int i = 0; i < n_splits; i++)
        if (split_bounds[i + 1] <= split_bounds[i]) { isx_set_error("split_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    Slot &s = p->slots[(size_t)(p->next_ticket % (int64_t)p->slots.size())];
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (s.state != 0) { isx_set_error("isx_pipe_submit: every slot is in use (collect + release the oldest batch first)"); return ISX_ERR_STATE; }
    }
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    HIP_TRY(hipSetDevice(c->device));
    const bool dense = b->M == 1, linkage = p->prm.enable_linkage != 0;

    // ---- host threads: records + group bases (+ pair-id runs) into pinned staging, reference codes, bounds ----
    const double t0 = now_ms();
    const bool ring = p->ring_half > 0;
    uint8_t *d_rec = s.d_in + s.off_rec;
    hipError_t ring_err = hipSuccess;
    size_t ring_bytes = 0;
    J.n_obs = n_obs; J.n_pos = n_pos; J.record_bytes = p->rb;
    J.rec = s.h_in + s.off_rec; J.gbase = reinterpret_cast<uint32_t *>(s.h_in + s.off_gbase);
    J.pair_out = nullptr;
    J.cmin = s.cmin.data(); J.cmax = s.cmax.data(); J.cany = s.cany.data();
    J.cap_rec = p->cap_rec;
    if (ring) {
        // the copy-in queue starts here: every finished wave leaves for its place in the device arena while the next
        // one is being written into the other half
        HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
        const size_t half_bytes = (size_t)p->ring_half * p->rb, grp_bytes = (size_t)p->G * p->rb;
        J.ring_groups = p->ring_half / p->G;
        J.wave_begin = [&](int h) {
            if (s.ring_busy[h]) { const hipError_t e = isx_wait_event(s.ev_ring[h]); if (e != hipSuccess && ring_err == hipSuccess) ring_err = e; s.ring_busy[h] = false; }
        };
        J.wave_flush = [&](int h, int64_t g0, int64_t g1) {
            const size_t n = (size_t)(g1 - g0) * grp_bytes;
            if (g0 == 0) ring_bytes = 0;                    // a layout that overflowed is written again from the start
            hipError_t e = hipMemcpyAsync(d_rec + (size_t)g0 * grp_bytes, s.h_in + s.off_rec + (size_t)h * half_bytes, n, hipMemcpyHostToDevice, H2D(p, s));
            if (e == hipSuccess) e = hipEventRecord(s.ev_ring[h], H2D(p, s));
            if (e != hipSuccess && ring_err == hipSuccess) ring_err = e;
            s.ring_busy[h] = true;
            ring_bytes += n;
        };
    }
    int erc = 0;
    for (int attempt = 0;; attempt++) {
        if (linkage) {
            J.runs_out = s.h_runs; J.cap_runs = s.cap_runs;
            J.run_index_out = reinterpret_cast<uint32_t *>(s.h_in + s.off_ridx);
        }
        J.slack = p->slack;
        ring_bytes = 0;
        erc = isxenc::encode_obs(*p->pool, J);
        if (erc != isxenc::ENC_OK || !linkage || J.n_runs <= s.cap_runs || attempt == 1) break;
        // more pair-id runs than any batch of this slot had (short fragments): larger blocks, encode again
        HIP_TRY(isx_wait_stream(H2D(p, s)));
        host_block_free(s.h_runs, s.runs_pinned); s.h_runs = nullptr;
        isx_dev_free(s.d_runs); s.d_runs = nullptr;
        s.cap_runs = J.n_runs + J.n_runs / 4 + 4096;
        { const int hrc = host_block_alloc(reinterpret_cast<void **>(&s.h_runs), s.cap_runs * sizeof(isxenc::PairRun), s.runs_pinned); if (hrc != ISX_OK) return hrc; }
        HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&s.d_runs), s.cap_runs * sizeof(uint2)));
    }
    const double t_enc = now_ms();
    if (ring_err != hipSuccess) { isx_set_error(std::string("isx_pipe_submit: staging ring: ") + hipGetErrorString(ring_err)); return ISX_ERR_HIP; }
    if (erc == isxenc::ENC_CAPACITY) { isx_set_error("isx_pipe_submit: the stream jumps too often for the pipe's record capacity (raise jump_slack)"); return ISX_ERR_CAPACITY; }
    if (erc == isxenc::ENC_MM_RANGE) { isx_set_error("an observation has mm >= 256"); return ISX_ERR_MM_RANGE; }
    if (erc == isxenc::ENC_BAD_POS) { isx_set_error("observation gpos >= n_pos"); return ISX_ERR_ARG; }
    if (linkage) {          // pair-id runs + run index were written into pinned staging by the encoder's threads
        if (J.n_runs > s.cap_runs) { isx_set_error("isx_pipe_submit: internal: pair-id run table did not fit after growing it"); return ISX_ERR_STATE; }
        b->n_runs = (uint32_t)J.n_runs;
    }
    if (J.passes > 1 && J.n_groups_in > 0)            // remember how jumpy this stream is: the next batch gets its slack up front
        p->slack = std::max(p->slack, 1.25 * ((double)J.n_groups_real / (double)J.n_groups_in - 1.0) + 0.01);
    s.ref_has_n = pack_ref2(*p->pool, ref, n_pos, s.h_in + s.off_ref, s.h_in + s.off_ref + ref2_bytes(n_pos));
    memcpy(s.h_in + s.off_bounds, split_bounds, (size_t)(n_splits + 1) * sizeof(int64_t));

    // ---- this batch's geometry ----
    b->n_pos = n_pos; b->n_obs = n_obs; b->n_splits = n_splits; b->n_rec = (uint64_t)J.n_rec;
    // without a count table to hand back, the position-sized tables travel shrunk (see finish_slot): clonality as the list of
    // values other than 1.0, coverage in one byte for a shallow batch
    b->sparse_out = dense && b->d_clon_list != nullptr;
    b->cov8_out = b->sparse_out && (double)b->n_obs < 16.0 * (double)n_pos;
    b->nib_out = b->cov8_out && b->lean && (double)b->n_obs < 6.0 * (double)n_pos;      // (mean depth below 6: most windows stay within 4 bits)
    b->clon_dense = false;
    b->rare_dense = !(b->lean && b->sparse_out) || (double)b->n_obs * 4.0 >= (double)p->prm.rarefied_coverage * (double)b->n_pos;
    if (b->lev_sparse) b->lev_cov_bytes = (double)b->n_obs < 64.0 * (double)b->n_pos ? 1 : 2;      // (a level of a batch this shallow rarely reaches 255: the exact values of those that do travel in a list)
    b->n_pairs = (uint64_t)J.max_pair + 1;
    const uint64_t n_chunks = b->n_rec / ISX_CHUNK;
    b->packed = 0;
    int W = batch_window_for(b, n_pos, false);
    if (!dense) {
        const int Wp = batch_window_for(b, n_pos, true);
        if (!(p->prm.layout & ISX_LAYOUT_NO_PACKED_COUNTERS) &&
            build_window_directory(s.cmin.data(), s.cmax.data(), s.cany.data(), n_chunks, Wp, n_pos, s.win) < 65536) { b->packed = 1; W = Wp; }
    }
    if (!b->packed) build_window_directory(s.cmin.data(), s.cmax.data(), s.cany.data(), n_chunks, W, n_pos, s.win);
    b->W = W;
    b->n_win = (int)s.win.size();
    if (s.win.size() > (size_t)p->pp.max_pos / 64 + 2) { isx_set_error("internal: window directory larger than the arena"); return ISX_ERR_STATE; }
    int rc = batch_set_geometry(b);
    if (rc != ISX_OK) return rc;
    if (!dense && !b->lev_sparse) {
        const size_t used = (size_t)b->n_win * b->slab;
        if (used > b->slab_region) { isx_set_error("internal: entry slabs larger than the slot's region"); return ISX_ERR_STATE; }
        b->cap_ovf = b->cap_entries - used;
    }
    memcpy(s.h_in + s.off_win, s.win.data(), s.win.size() * sizeof(uint2));
    b->d_bounds = reinterpret_cast<int64_t *>(s.d_in + s.off_bounds);
    b->d_win = reinterpret_cast<uint2 *>(s.d_in + s.off_win);
    b->d_ref = s.d_in + s.off_ref;
    b->d_ref_n = s.ref_has_n ? s.d_in + s.off_ref + ref2_bytes(b->n_pos) : nullptr;
    b->d_gbase = reinterpret_cast<uint32_t *>(s.d_in + s.off_gbase);
    b->d_rec16 = p->rb == 2 ? reinterpret_cast<uint16_t *>(s.d_in + s.off_rec) : nullptr;
    b->d_rec32 = p->rb == 4 ? reinterpret_cast<uint32_t *>(s.d_in + s.off_rec) : nullptr;
    b->d_pair = nullptr;
    b->d_pair_runs = linkage ? s.d_runs : nullptr;
    b->d_run_index = linkage ? reinterpret_cast<uint32_t *>(s.d_in + s.off_ridx) : nullptr;
    b->d_gpos16 = s.d_gpos16; b->gpos16_shift = 5;
    s.encode_ms = (float)(now_ms() - t0);
    s.encode_passes = J.passes;
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe_submit] records %.2f ms (%d pass), reference + bounds + windows %.2f ms; %lld obs, %lld runs\n",
                t_enc - t0, J.passes, now_ms() - t_enc, (long long)n_obs, (long long)J.n_runs);

    // ---- copy-in queue ----
    if (!ring) HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
    const size_t ref_bytes = ref2_bytes(n_pos) + (s.ref_has_n ? refn_bytes(n_pos) : 0);   // 2-bit plane (+ the non-ACGT bit plane)
    const size_t head = (size_t)(n_splits + 1) * sizeof(int64_t) + s.win.size() * sizeof(uint2) + ref_bytes;
    // the stream is followed by a tail of padding records / zero bases (see ISX_TAIL_BYTES): the slot's arena still
    // holds the previous batch there
    memset(s.h_in + s.off_gbase + (size_t)(b->n_rec / p->G) * sizeof(uint32_t), 0, ISX_TAIL_GROUPS * sizeof(uint32_t));
    const size_t gb_bytes = ((size_t)(b->n_rec / p->G) + ISX_TAIL_GROUPS) * sizeof(uint32_t), rec_bytes = (size_t)b->n_rec * p->rb + ISX_TAIL_BYTES;
    // bounds | windows | reference planes: what is used of each region, not the regions (a slot is sized for the largest batch)
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_bounds, s.h_in + s.off_bounds, (size_t)(n_splits + 1) * sizeof(int64_t), hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_win, s.h_in + s.off_win, s.win.size() * sizeof(uint2), hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref, s.h_in + s.off_ref, ref_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_gbase, s.h_in + s.off_gbase, gb_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    if (ring) {             // the records left wave by wave; the tail is written on the device
        if (p->rb == 2) HIP_TRY(hipMemsetD8Async(reinterpret_cast<hipDeviceptr_t>(d_rec + (size_t)b->n_rec * 2), 0xFF, ISX_TAIL_BYTES, H2D(p, s)));
        else HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_rec + (size_t)b->n_rec * 4), (int)ISX_PAD32, ISX_TAIL_BYTES / 4, H2D(p, s)));
        if (ring_bytes != (size_t)b->n_rec * p->rb) { isx_set_error("internal: the staging ring did not carry the whole stream"); return ISX_ERR_STATE; }
    } else {
        if (p->rb == 2) memset(s.h_in + s.off_rec + (size_t)b->n_rec * 2, 0xFF, ISX_TAIL_BYTES);
        else std::fill_n(reinterpret_cast<uint32_t *>(s.h_in + s.off_rec + (size_t)b->n_rec * 4), ISX_TAIL_BYTES / 4, (uint32_t)ISX_PAD32);
        HIP_TRY(hipMemcpyAsync(d_rec, s.h_in + s.off_rec, rec_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    }
    s.h2d_bytes = (int64_t)(head + gb_bytes + rec_bytes);
    if (linkage) {
        const size_t rb = (size_t)b->n_runs * sizeof(isxenc::PairRun), ib = (size_t)(b->n_rec / ISX_CHUNK) * sizeof(uint32_t);
        HIP_TRY(hipMemcpyAsync(s.d_runs, s.h_runs, rb, hipMemcpyHostToDevice, H2D(p, s)));
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ridx, s.h_in + s.off_ridx, ib, hipMemcpyHostToDevice, H2D(p, s)));
        s.h2d_bytes += (int64_t)(rb + ib);
    }
    HIP_TRY(hipEventRecord(s.ev_h2d1, H2D(p, s)));
    s.h2d_split = false;

    return enqueue_pass(p, s, n_pos, ticket);
}

// a read-level batch: `J` arrives with its input side set (isx_segs arrays, or a producer + the segment starts)
// (bit-plane input -- J.in2 / J.produce_planes -- goes through encode_planes; its reference arrives as planes (`rp`) or as codes)
static int submit_segs_common(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                              isxenc::SegJob &J, int64_t *ticket, const isx_ref_planes *rp)
{
    const bool planes_in = J.in2.planes != nullptr || (bool)J.produce_planes || (J.n_seg == 0 && rp != nullptr);
    if (n_pos > p->pp.max_pos || J.n_seg > p->pp.max_segs || n_splits > p->pp.max_splits) {
        isx_set_error("isx_pipe_submit_reads: batch larger than the pipe was created for");
        return ISX_ERR_CAPACITY;
    }
    if (split_bounds[0] != 0 || split_bounds[n_splits] != n_pos) { isx_set_error("split_bounds must span [0, n_pos]"); return ISX_ERR_ARG; }
    for (int i = 0; i < n_splits; i++)
        if (split_bounds[i + 1] <= split_bounds[i]) { isx_set_error("split_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    Slot &s = p->slots[(size_t)(p->next_ticket % (int64_t)p->slots.size())];
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (s.state != 0) { isx_set_error("isx_pipe_submit_reads: every slot is in use (collect + release the oldest batch first)"); return ISX_ERR_STATE; }
    }
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    HIP_TRY(hipSetDevice(c->device));
    const bool dense = b->M == 1, linkage = p->prm.enable_linkage != 0;

    // ---- host threads: records + group bases (+ pair ids) into pinned staging, reference codes, bounds ----
    const double t0 = now_ms();
    J.n_pos = n_pos; J.n_mm_bins = b->M;
    J.rec = reinterpret_cast<uint32_t *>(s.h_in + s.off_rec);
    J.gbase = reinterpret_cast<uint32_t *>(s.h_in + s.off_gbase);
    J.pair_out = linkage && !p->drec ? reinterpret_cast<uint32_t *>(s.h_in + s.off_pairs) : nullptr;
    J.cmin = s.cmin.data(); J.cmax = s.cmax.data(); J.cany = s.cany.data();
    J.cap_rec = p->cap_rec;
    const bool ring = p->ring_half > 0;
    hipError_t ring_err = hipSuccess;
    size_t ring_bytes = 0;
    if (ring) {
        // the copy-in queue starts here: every finished wave leaves for its place in the device arena while the next one is
        // being written into the other half
        HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
        const size_t half_bytes = (size_t)p->ring_half * (size_t)p->rb;
        constexpr size_t grp_bytes = (size_t)ISX_SEG_GROUP * 64;           // (= ISX_DREC_GROUP * 32: a group is 1 KiB in both formats)
        uint8_t *d_rec = s.d_in + s.off_rec;
        J.ring_groups = p->ring_half / (int64_t)p->G;
        J.wave_begin = [&s, &ring_err](int h) {
            if (s.ring_busy[h]) { const hipError_t e = isx_wait_event(s.ev_ring[h]); if (e != hipSuccess && ring_err == hipSuccess) ring_err = e; s.ring_busy[h] = false; }
        };
        J.wave_flush = [&s, p, &ring_err, &ring_bytes, half_bytes, d_rec](int h, int64_t g0, int64_t g1) {
            const size_t nb = (size_t)(g1 - g0) * grp_bytes;
            hipError_t e = hipMemcpyAsync(d_rec + (size_t)g0 * grp_bytes, s.h_in + s.off_rec + (size_t)h * half_bytes, nb, hipMemcpyHostToDevice, H2D(p, s));
            if (e == hipSuccess) e = hipEventRecord(s.ev_ring[h], H2D(p, s));
            if (e != hipSuccess && ring_err == hipSuccess) ring_err = e;
            s.ring_busy[h] = true;
            ring_bytes += nb;
        };
    }
    int erc;
    double t_ref0 = 0.0;
    bool early_ref = false, early_rec = false;
    uint8_t *resident_ref = nullptr;            // the batch's reference planes already on the device (isx_ref_planes.key)
    if (planes_in) {
        // the reference planes first, into staging: the record pass compares against that copy
        if (!p->drec) { isx_set_error("bit-plane reads need a pipe whose batches travel as reference-delta records (one mm bin, or ISX_LAYOUT_MM_DELTA_RECORDS)"); return ISX_ERR_STATE; }
        const double t_r = now_ms();
        uint8_t *h2 = s.h_in + s.off_ref, *hn = h2 + ref2_bytes(n_pos);
        if (rp && rp->key) {
            auto it = p->ref_cache.find(rp->key);
            if (it != p->ref_cache.end()) {
                // the device already holds this reference: nothing is staged, nothing travels; the record pass compares against the
                // caller's own planes
                const isx_pipe::RefEntry &e = it->second;
                if (e.n_pos != n_pos || e.has_n != (rp->nplane != nullptr && e.has_n) || e.sum != ref_plane_checksum(rp->plane2, n_pos)) {
                    isx_set_error("isx_pipe_submit_planes: the reference key stands for other planes (positions / non-ACGT plane / content differ)");
                    return ISX_ERR_ARG;
                }
                s.ref_has_n = e.has_n;
                J.ref2 = rp->plane2; J.refn = e.has_n ? rp->nplane : nullptr;
                resident_ref = e.d;
                HIP_TRY(hipStreamWaitEvent(H2D(p, s), e.ready, 0));         // (the entry's own copy, enqueued by an earlier submit)
                if (!ring) HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
                HIP_TRY(hipEventRecord(s.ev_h2da, H2D(p, s)));
                early_ref = true;
                t_ref0 = now_ms() - t_r;
                goto ref_staged;
            }
        }
        if (rp && !(rp->key && !p->ref_cache.count(rp->key) && p->ref_cache_budget.load(std::memory_order_relaxed) > 0) &&
            isx_host_is_registered(rp->plane2, ((size_t)n_pos + 3) / 4) &&
            (!rp->nplane || isx_host_is_registered(rp->nplane, ((size_t)n_pos + 7) / 8))) {
            // the caller's planes are registered for the copy engine (isx_host_register): they leave from where they lie, nothing is
            // staged, the record pass compares against them (a key's FIRST trip still goes through staging: its device copy is made of that)
            bool any_n = false;
            if (rp->nplane) any_n = plane_any(*p->pool, rp->nplane, ((size_t)n_pos + 7) / 8);
            s.ref_has_n = any_n;
            J.ref2 = rp->plane2; J.refn = any_n ? rp->nplane : nullptr;
            t_ref0 = now_ms() - t_r;
            if (!ring) HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
            HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref, rp->plane2, ((size_t)n_pos + 3) / 4, hipMemcpyHostToDevice, H2D(p, s)));
            if (any_n) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref + ref2_bytes(n_pos), rp->nplane, ((size_t)n_pos + 7) / 8, hipMemcpyHostToDevice, H2D(p, s)));
            if (!ring) HIP_TRY(hipEventRecord(s.ev_h2da, H2D(p, s)));
            early_ref = true;
            goto ref_staged;
        }
        s.ref_has_n = rp ? copy_ref_planes(*p->pool, rp, n_pos, h2, hn) : pack_ref2(*p->pool, ref, n_pos, h2, hn);
        J.ref2 = h2; J.refn = s.ref_has_n ? hn : nullptr;
        t_ref0 = now_ms() - t_r;
        // ... and leave at once: the DMA engine brings the reference in while the threads make the records
        // (ISX_PIPE_LATE_DMA=1: every copy after the host pass, as before round 5 -- same-box A/B)
        static const bool late_dma = getenv("ISX_PIPE_LATE_DMA") != nullptr;
        if (late_dma) goto ref_staged;
        if (!ring) HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref, h2, ref2_bytes(n_pos) + (s.ref_has_n ? refn_bytes(n_pos) : 0), hipMemcpyHostToDevice, H2D(p, s)));
        if (!ring) HIP_TRY(hipEventRecord(s.ev_h2da, H2D(p, s)));
        early_ref = true;
        if (rp && rp->key && !resident_ref) {
            // first trip of this key: a device-side copy of what just arrived stays with the pipe (while its budget lasts)
            const size_t rb_all = ref2_bytes(n_pos) + (s.ref_has_n ? refn_bytes(n_pos) : 0);
            if (p->ref_cache_bytes + rb_all <= p->ref_cache_budget.load(std::memory_order_relaxed)) {
                isx_pipe::RefEntry e;
                bool ok = isx_dev_malloc(reinterpret_cast<void **>(&e.d), rb_all + 64) == hipSuccess && hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) == hipSuccess;
                ok = ok && hipMemcpyAsync(e.d, s.d_in + s.off_ref, rb_all, hipMemcpyDeviceToDevice, H2D(p, s)) == hipSuccess && hipEventRecord(e.ready, H2D(p, s)) == hipSuccess;
                if (ok) {
                    e.bytes = rb_all; e.n_pos = n_pos; e.has_n = s.ref_has_n; e.sum = ref_plane_checksum(rp->plane2, n_pos);
                    p->ref_cache_bytes += rb_all;
                    p->ref_cache.emplace(rp->key, e);
                } else {                            // (no entry: the batch goes on with the planes that just travelled; nothing leaks)
                    if (e.d) isx_dev_free(e.d);
                    if (e.ready) (void)hipEventDestroy(e.ready);
                    (void)hipGetLastError();
                }
            }
        }
    ref_staged:;
    }
    if (p->drec) {
        // reference-delta records: the segments are compared with the reference here; pieces of segments with more than six
        // differences take spare groups of their task's region -- a batch that needs more than the pipe has learned so far is
        // encoded a second time (ring mode: its waves simply travel again)
        J.ref = ref;
        std::vector<int64_t> exact;
        s.encode_passes = 1;
        for (int attempt = 0;; attempt++) {
            J.slack_groups = p->dslack;
            J.task_groups = exact.empty() ? nullptr : exact.data();
            ring_bytes = 0;
            erc = planes_in ? isxenc::encode_planes(*p->pool, J) : isxenc::encode_delta(*p->pool, J);
            if (erc == isxenc::SEG_CAPACITY && J.need_slack > p->dslack && attempt == 0) {
                // the second attempt gives every task exactly what the first one found it needs; the pipe remembers the AVERAGE
                // surplus (data that differs from the reference everywhere then fits at once; one task over a stretch where the
                // reference is not A/C/T/G does not inflate the others)
                exact = J.task_need;
                int64_t tot = 0;
                for (int64_t v : exact) tot += v;
                const int64_t n_t = (int64_t)exact.size(), base_tot = J.n_rec / ISX_DREC_GROUP - n_t * p->dslack;
                p->dslack = std::max<int64_t>(p->dslack, 1 + (tot - base_tot + n_t - 1) / std::max<int64_t>(n_t, 1));
                s.encode_passes = 2;
                continue;
            }
            break;
        }
    } else erc = isxenc::encode_segs(*p->pool, J);
    const double t_enc = now_ms();
    if (ring_err != hipSuccess) { isx_set_error(std::string("isx_pipe_submit_reads: staging ring: ") + hipGetErrorString(ring_err)); return ISX_ERR_HIP; }
    if (erc == isxenc::SEG_CAPACITY) { isx_set_error("isx_pipe_submit_reads: the stream jumps too often for the pipe's record capacity (raise jump_slack)"); return ISX_ERR_CAPACITY; }
    if (erc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
    if (erc == isxenc::SEG_BAD_POS) { isx_set_error("a segment reaches beyond n_pos"); return ISX_ERR_ARG; }
    if (erc == isxenc::SEG_BAD_LEN) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
    if (J.n_bases > p->pp.max_obs) { isx_set_error("isx_pipe_submit_reads: more bases than the pipe's max_obs"); return ISX_ERR_CAPACITY; }
    if (early_ref && !ring) {
        // the records leave as soon as they exist; the window directory below is made while they travel
        const size_t gb_bytes0 = (size_t)(J.n_rec / (int64_t)p->G) * sizeof(uint32_t), rec_bytes0 = (size_t)J.n_rec * (size_t)p->rb;
        HIP_TRY(hipEventRecord(s.ev_h2db, H2D(p, s)));
        if (s.off_pairs == s.off_rec && s.off_ridx == s.off_rec && s.off_rec - s.off_gbase < ((size_t)2 << 20)) {
            // group bases | records lie side by side in both arenas (nothing between them in a pipe of reference-delta records): ONE copy --
            // a small copy of its own is a blit kernel and a queue entry; the unused tail of the group bases' region (< 2 MB) travels along
            HIP_TRY(hipMemcpyAsync(s.d_in + s.off_gbase, s.h_in + s.off_gbase, (s.off_rec - s.off_gbase) + rec_bytes0, hipMemcpyHostToDevice, H2D(p, s)));
        } else {
            HIP_TRY(hipMemcpyAsync(s.d_in + s.off_gbase, s.h_in + s.off_gbase, gb_bytes0, hipMemcpyHostToDevice, H2D(p, s)));
            HIP_TRY(hipMemcpyAsync(s.d_in + s.off_rec, s.h_in + s.off_rec, rec_bytes0, hipMemcpyHostToDevice, H2D(p, s)));
        }
        early_rec = true;
    }
    if (!planes_in) s.ref_has_n = pack_ref2(*p->pool, ref, n_pos, s.h_in + s.off_ref, s.h_in + s.off_ref + ref2_bytes(n_pos));
    const double t_ref = now_ms();
    memcpy(s.h_in + s.off_bounds, split_bounds, (size_t)(n_splits + 1) * sizeof(int64_t));

    // ---- this batch's geometry ----
    b->n_pos = n_pos; b->n_obs = J.n_bases; b->n_splits = n_splits; b->n_rec = (uint64_t)J.n_rec;
    // without a count table to hand back, the position-sized tables travel shrunk (see finish_slot): clonality as the list of
    // values other than 1.0, coverage in one byte for a shallow batch
    b->sparse_out = dense && b->d_clon_list != nullptr;
    b->cov8_out = b->sparse_out && (double)b->n_obs < 16.0 * (double)n_pos;
    b->nib_out = b->cov8_out && b->lean && (double)b->n_obs < 6.0 * (double)n_pos;      // (mean depth below 6: most windows stay within 4 bits)
    b->clon_dense = false;
    b->rare_dense = !(b->lean && b->sparse_out) || (double)b->n_obs * 4.0 >= (double)p->prm.rarefied_coverage * (double)b->n_pos;
    if (b->lev_sparse) b->lev_cov_bytes = (double)b->n_obs < 64.0 * (double)b->n_pos ? 1 : 2;      // (a level of a batch this shallow rarely reaches 255: the exact values of those that do travel in a list)
    b->n_pairs = (uint64_t)J.max_pair + 1;
    const uint64_t n_chunks = b->n_rec / p->G;
    b->packed = 0;
    int W = batch_window_for(b, n_pos, false);
    if (!dense || p->drec) {
        const int Wp = batch_window_for(b, n_pos, true);
        if (!(p->prm.layout & ISX_LAYOUT_NO_PACKED_COUNTERS) &&
            build_window_directory_mt(*p->pool, s.cmin.data(), s.cmax.data(), s.cany.data(), n_chunks, Wp, n_pos, s.win, p->G, s.dir_pmax, s.dir_smin) < (p->drec ? 32768u : 65536u)) { b->packed = 1; W = Wp; }
    }
    if (!b->packed) build_window_directory_mt(*p->pool, s.cmin.data(), s.cmax.data(), s.cany.data(), n_chunks, W, n_pos, s.win, p->G, s.dir_pmax, s.dir_smin);
    b->W = W;
    b->n_win = (int)s.win.size();
    if (s.win.size() > (size_t)p->pp.max_pos / 64 + 2) { isx_set_error("internal: window directory larger than the arena"); return ISX_ERR_STATE; }
    int rc = batch_set_geometry(b);
    if (rc != ISX_OK) return rc;
    if (!dense && !b->lev_sparse) {
        const size_t used = (size_t)b->n_win * b->slab;
        if (used > b->slab_region) { isx_set_error("internal: entry slabs larger than the slot's region"); return ISX_ERR_STATE; }
        b->cap_ovf = b->cap_entries - used;
    }
    memcpy(s.h_in + s.off_win, s.win.data(), s.win.size() * sizeof(uint2));
    b->d_bounds = reinterpret_cast<int64_t *>(s.d_in + s.off_bounds);
    b->d_win = reinterpret_cast<uint2 *>(s.d_in + s.off_win);
    b->d_ref = resident_ref ? resident_ref : s.d_in + s.off_ref;
    b->d_ref_n = s.ref_has_n ? b->d_ref + ref2_bytes(b->n_pos) : nullptr;
    b->d_gbase = reinterpret_cast<uint32_t *>(s.d_in + s.off_gbase);
    b->d_seg = p->drec ? nullptr : reinterpret_cast<uint4 *>(s.d_in + s.off_rec);
    b->d_drec = p->drec ? reinterpret_cast<uint4 *>(s.d_in + s.off_rec) : nullptr;
    b->d_rec16 = nullptr; b->d_rec32 = nullptr;
    b->d_pair = linkage && !p->drec ? reinterpret_cast<uint32_t *>(s.d_in + s.off_pairs) : nullptr;
    b->d_pair_runs = nullptr; b->d_run_index = nullptr; b->n_runs = 0;
    s.encode_ms = (float)(now_ms() - t0);
    if (!p->drec) s.encode_passes = 1;
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe_submit_reads] records %.2f ms, reference %.2f ms, bounds + windows %.2f ms; %lld segments, %lld records%s\n",
                t_enc - t0 - t_ref0, t_ref - t_enc + t_ref0, now_ms() - t_ref, (long long)J.n_seg, (long long)J.n_rec, planes_in ? " (bit planes)" : "");
    const double t_q0 = now_ms();

    // ---- copy-in queue: bounds | windows | reference codes, then group bases (| pair ids) | records ----
    if (!ring && !early_ref) HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
    const size_t ref_bytes = ref2_bytes(n_pos) + (s.ref_has_n ? refn_bytes(n_pos) : 0);   // 2-bit plane (+ the non-ACGT bit plane)
    const size_t head = (size_t)(n_splits + 1) * sizeof(int64_t) + s.win.size() * sizeof(uint2) + ref_bytes;
    const size_t gb_bytes = (size_t)(b->n_rec / p->G) * sizeof(uint32_t), rec_bytes = (size_t)b->n_rec * (size_t)p->rb;
    // bounds | windows | reference planes: what is used of each region, not the regions (a slot is sized for the largest batch)
    if (s.off_win - s.off_bounds < ((size_t)1 << 20)) {          // bounds | windows in one copy (the bounds' region is a slot's max_splits: small)
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_bounds, s.h_in + s.off_bounds, (s.off_win - s.off_bounds) + s.win.size() * sizeof(uint2), hipMemcpyHostToDevice, H2D(p, s)));
    } else {
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_bounds, s.h_in + s.off_bounds, (size_t)(n_splits + 1) * sizeof(int64_t), hipMemcpyHostToDevice, H2D(p, s)));
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_win, s.h_in + s.off_win, s.win.size() * sizeof(uint2), hipMemcpyHostToDevice, H2D(p, s)));
    }
    if (!early_ref) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref, s.h_in + s.off_ref, ref_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    if (!early_rec) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_gbase, s.h_in + s.off_gbase, gb_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    if (ring) { if (ring_bytes != rec_bytes) { isx_set_error("internal: the staging ring did not carry the whole stream"); return ISX_ERR_STATE; } }
    else if (!early_rec) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_rec, s.h_in + s.off_rec, rec_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    s.h2d_bytes = (int64_t)(head + gb_bytes + rec_bytes) - (resident_ref ? (int64_t)ref_bytes : 0);
    if (linkage && !p->drec) {
        HIP_TRY(hipMemcpyAsync(s.d_in + s.off_pairs, s.h_in + s.off_pairs, (size_t)b->n_rec * sizeof(uint32_t), hipMemcpyHostToDevice, H2D(p, s)));
        s.h2d_bytes += (int64_t)b->n_rec * 4;
    }
    HIP_TRY(hipEventRecord(s.ev_h2d1, H2D(p, s)));
    s.h2d_split = early_rec;
    const double t_q1 = now_ms();
    rc = enqueue_pass(p, s, n_pos, ticket);
    if (getenv("ISX_PIPE_TIMING")) fprintf(stderr, "[isx_pipe_submit_reads] copy-in queue %.2f ms, pass + copy-out queue %.2f ms\n", t_q1 - t_q0, now_ms() - t_q1);
    return rc;
}

// ---- staged batches (isx_wire): the host work of a read-level submit done ONCE, ahead of time, into a pinned image of its own ----
struct isx_wire {
    isx_pipe *pipe = nullptr;           // staged for this pipe: its record format, capacities and window choice
    uint8_t *h = nullptr;               // pinned: bounds | window directory | reference codes | group bases | pair ids | records
    size_t bytes = 0;
    size_t o_bounds = 0, o_win = 0, o_ref = 0, o_gbase = 0, o_pairs = 0, o_rec = 0;
    size_t bounds_bytes = 0, win_bytes = 0, ref_bytes = 0, gbase_bytes = 0, pairs_bytes = 0, rec_bytes = 0;
    int64_t n_pos = 0, n_bases = 0, n_rec = 0, n_seg = 0;
    int32_t n_splits = 0;
    uint64_t n_pairs = 0;
    int W = 0, packed = 0;
    bool ref_has_n = false;
    float stage_ms = 0.f;
    int encode_passes = 1;
    uint8_t *d_ref = nullptr;           // isx_wire_keep_reference: the reference planes stay on the device between submits
    int device = 0;
};

// isx_pipe_stage_reads / isx_pipe_stage_planes: `segs` + `ref`, or `reads` + `rp`
static int stage_common(isx_pipe *p, int64_t n_pos, const uint8_t *ref, const isx_ref_planes *rp, int32_t n_splits, const int64_t *split_bounds,
                        const isx_segs *segs, const isx_read_planes *reads, isx_wire **out)
{
    isx_segs as_segs{};
    if (reads) { as_segs.n_seg = reads->n_seg; as_segs.gpos = reads->gpos; as_segs.len = reads->len; as_segs.pair = reads->pair; segs = &as_segs; }
    *out = nullptr;
    if (!p->segs) { isx_set_error("isx_pipe_stage_reads: not a read-level pipe (isx_pipe_params.max_segs == 0)"); return ISX_ERR_STATE; }
    const bool linkage = p->prm.enable_linkage != 0;
    if (linkage && segs->n_seg && !segs->pair) { isx_set_error("linkage needs the pair array"); return ISX_ERR_ARG; }
    if (n_pos > p->pp.max_pos || segs->n_seg > p->pp.max_segs || n_splits > p->pp.max_splits) {
        isx_set_error("isx_pipe_stage_reads: batch larger than the pipe was created for");
        return ISX_ERR_CAPACITY;
    }
    if (split_bounds[0] != 0 || split_bounds[n_splits] != n_pos) { isx_set_error("split_bounds must span [0, n_pos]"); return ISX_ERR_ARG; }
    for (int i = 0; i < n_splits; i++)
        if (split_bounds[i + 1] <= split_bounds[i]) { isx_set_error("split_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    const double t0 = now_ms();
    std::unique_ptr<isx_wire, void (*)(isx_wire *)> w(new isx_wire(), isx_wire_free);
    w->pipe = p; w->n_pos = n_pos; w->n_splits = n_splits; w->n_seg = segs->n_seg;
    const isx_batch *b0 = p->slots[0].b;            // (every slot shares the parameters the window choice depends on)
    const int M = b0->M;
    const size_t G = p->G, rb = (size_t)p->rb;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    isxenc::SegJob J;
    std::vector<uint32_t> cmin, cmax;
    std::vector<uint8_t> cany;
    std::vector<int64_t> exact;                 // second attempt: every encoder task's region is what the first one found it needs
    for (int attempt = 0;; attempt++) {
        // exact record capacity of this batch (the encoder's own counting pass), at most the pipe's
        int64_t cap = p->drec ? isxenc::delta_groups_needed(*p->pool, segs->gpos, segs->n_seg, p->dslack) * ISX_DREC_GROUP
                              : isxenc::seg_groups_needed(*p->pool, segs->gpos, segs->n_seg) * ISX_SEG_GROUP;
        if (!exact.empty()) { cap = 0; for (int64_t v : exact) cap += std::max<int64_t>(v, 1) * ISX_DREC_GROUP; }
        if (cap > p->cap_rec) { isx_set_error("isx_pipe_stage_reads: the stream needs more records than the pipe's capacity (raise jump_slack)"); return ISX_ERR_CAPACITY; }
        const size_t n_groups = (size_t)cap / G;
        w->bounds_bytes = (size_t)(n_splits + 1) * sizeof(int64_t);
        w->ref_bytes = ref2_bytes(n_pos) + refn_bytes(n_pos);       // (the non-ACGT plane only travels when the batch has such a position)
        const size_t win_cap = ((size_t)n_pos / 64 + 2) * sizeof(uint2);       // (the smallest window is 64 positions)
        size_t o = 0;
        w->o_bounds = o; o = up(o + w->bounds_bytes);
        w->o_win = o; o = up(o + win_cap);
        w->o_ref = o; o = up(o + w->ref_bytes);
        w->o_gbase = o; o = up(o + n_groups * sizeof(uint32_t));
        w->o_pairs = o; if (linkage && !p->drec) o = up(o + (size_t)cap * sizeof(uint32_t));
        w->o_rec = o; o = up(o + (size_t)cap * rb);
        if (w->h) { isx_pin_free(w->h); w->h = nullptr; }
        HIP_TRY(isx_pin_malloc(reinterpret_cast<void **>(&w->h), o));
        w->bytes = o;
        cmin.assign(n_groups + 2, 0xFFFFFFFFu); cmax.assign(n_groups + 2, 0u); cany.assign(n_groups + 2, 0);
        J = isxenc::SegJob();
        J.n_seg = segs->n_seg; J.n_pos = n_pos; J.n_mm_bins = M;
        if (reads) { J.in2 = *reads; if (!linkage) J.in2.pair = nullptr; }
        else { J.in = *segs; if (!linkage) J.in.pair = nullptr; }
        J.rec = reinterpret_cast<uint32_t *>(w->h + w->o_rec);
        J.gbase = reinterpret_cast<uint32_t *>(w->h + w->o_gbase);
        J.pair_out = linkage && !p->drec ? reinterpret_cast<uint32_t *>(w->h + w->o_pairs) : nullptr;
        J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data();
        J.cap_rec = cap;
        int erc;
        if (reads) {                     // the reference planes first (the record pass compares against the image's copy)
            uint8_t *h2 = w->h + w->o_ref, *hn = h2 + ref2_bytes(n_pos);
            w->ref_has_n = rp ? copy_ref_planes(*p->pool, rp, n_pos, h2, hn) : pack_ref2(*p->pool, ref, n_pos, h2, hn);
            J.ref2 = h2; J.refn = w->ref_has_n ? hn : nullptr;
        }
        if (p->drec) {
            J.ref = ref; J.slack_groups = p->dslack;
            J.task_groups = exact.empty() ? nullptr : exact.data();
            erc = reads ? isxenc::encode_planes(*p->pool, J) : isxenc::encode_delta(*p->pool, J);
            if (erc == isxenc::SEG_CAPACITY && J.need_slack > p->dslack && attempt == 0) { exact = J.task_need; w->encode_passes = 2; continue; }
        } else erc = isxenc::encode_segs(*p->pool, J);
        if (erc == isxenc::SEG_CAPACITY) { isx_set_error("isx_pipe_stage_reads: the stream does not fit the record capacity"); return ISX_ERR_CAPACITY; }
        if (erc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
        if (erc == isxenc::SEG_BAD_POS) { isx_set_error("a segment reaches beyond n_pos"); return ISX_ERR_ARG; }
        if (erc == isxenc::SEG_BAD_LEN) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
        break;
    }
    if (J.n_bases > p->pp.max_obs) { isx_set_error("isx_pipe_stage_reads: more bases than the pipe's max_obs"); return ISX_ERR_CAPACITY; }
    w->n_rec = J.n_rec; w->n_bases = J.n_bases; w->n_pairs = (uint64_t)J.max_pair + 1;
    w->gbase_bytes = (size_t)(J.n_rec / (int64_t)G) * sizeof(uint32_t);
    w->rec_bytes = (size_t)J.n_rec * rb;
    w->pairs_bytes = linkage && !p->drec ? (size_t)J.n_rec * sizeof(uint32_t) : 0;
    if (!reads) w->ref_has_n = pack_ref2(*p->pool, ref, n_pos, w->h + w->o_ref, w->h + w->o_ref + ref2_bytes(n_pos));
    if (!w->ref_has_n) w->ref_bytes = ref2_bytes(n_pos);
    memcpy(w->h + w->o_bounds, split_bounds, w->bounds_bytes);
    {   // the window directory, for the window this pipe's kernels will use on a batch of n_pos positions
        const uint64_t n_chunks = (uint64_t)J.n_rec / G;
        std::vector<uint2> win;
        std::vector<uint32_t> dir_pmax, dir_smin;
        w->packed = 0;
        int W = batch_window_for(b0, n_pos, false);
        if (M > 1 || p->drec) {
            const int Wp = batch_window_for(b0, n_pos, true);
            if (!(p->prm.layout & ISX_LAYOUT_NO_PACKED_COUNTERS) &&
                build_window_directory_mt(*p->pool, cmin.data(), cmax.data(), cany.data(), n_chunks, Wp, n_pos, win, (uint32_t)G, dir_pmax, dir_smin) < (p->drec ? 32768u : 65536u)) { w->packed = 1; W = Wp; }
        }
        if (!w->packed) build_window_directory_mt(*p->pool, cmin.data(), cmax.data(), cany.data(), n_chunks, W, n_pos, win, (uint32_t)G, dir_pmax, dir_smin);
        w->W = W;
        w->win_bytes = win.size() * sizeof(uint2);
        if (w->win_bytes > ((size_t)n_pos / 64 + 2) * sizeof(uint2) || win.size() > (size_t)p->pp.max_pos / 64 + 2) { isx_set_error("internal: window directory larger than its region"); return ISX_ERR_STATE; }
        memcpy(w->h + w->o_win, win.data(), w->win_bytes);
    }
    w->stage_ms = (float)(now_ms() - t0);
    *out = w.release();
    return ISX_OK;
}

int isx_pipe_stage_reads(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                         const isx_segs *segs, isx_wire **out)
{
    if (!p || !ref || !split_bounds || !out || n_pos <= 0 || n_splits <= 0 || !segs || segs->n_seg < 0 ||
        (segs->n_seg && (!segs->gpos || !segs->len || !segs->bases))) {
        isx_set_error("isx_pipe_stage_reads: bad argument");
        return ISX_ERR_ARG;
    }
    return stage_common(p, n_pos, ref, nullptr, n_splits, split_bounds, segs, nullptr, out);
}

int isx_pipe_stage_planes(isx_pipe *p, int64_t n_pos, const isx_ref_planes *ref, int32_t n_splits, const int64_t *split_bounds,
                          const isx_read_planes *reads, isx_wire **out)
{
    if (!p || !ref || !ref->plane2 || !split_bounds || !out || n_pos <= 0 || n_splits <= 0 || !reads || reads->n_seg < 0 ||
        (reads->n_seg && (!reads->gpos || !reads->len || !reads->planes))) {
        isx_set_error("isx_pipe_stage_planes: bad argument");
        return ISX_ERR_ARG;
    }
    if (!p->drec) { isx_set_error("isx_pipe_stage_planes: bit-plane reads need a read-level pipe with one mm bin or ISX_LAYOUT_MM_DELTA_RECORDS"); return ISX_ERR_STATE; }
    return stage_common(p, n_pos, nullptr, ref, n_splits, split_bounds, nullptr, reads, out);
}

void isx_wire_free(isx_wire *w)
{
    if (!w) return;
    if (w->h) isx_pin_free(w->h);
    if (w->d_ref) { (void)hipSetDevice(w->device); isx_dev_free(w->d_ref); }
    delete w;
}

// The reference planes of a staged batch stay in HBM from now on: later submits of the wire copy everything BUT them.  The reference of a
// database is the same for every sample profiled against it (the reference program holds its fasta in host memory for the whole run,
// profile_controller.py:415-433); a quarter of a shallow metagenome batch's copy-in is its 2-bit plane.
int isx_wire_keep_reference(isx_pipe *p, isx_wire *w)
{
    if (!p || !w || w->pipe != p) { isx_set_error("isx_wire_keep_reference: bad argument"); return ISX_ERR_ARG; }
    if (w->d_ref || !w->ref_bytes) return ISX_OK;
    isx_ctx *c = p->ctx;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t *d = nullptr;
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&d), w->ref_bytes + 64));
    const hipError_t e = hipMemcpy(d, w->h + w->o_ref, w->ref_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { isx_dev_free(d); isx_set_error(std::string("isx_wire_keep_reference: ") + hipGetErrorString(e)); return ISX_ERR_HIP; }
    w->d_ref = d; w->device = c->device;
    return ISX_OK;
}

int64_t isx_wire_bytes(const isx_wire *w) { return w ? (int64_t)(w->bounds_bytes + w->win_bytes + (w->d_ref ? 0 : w->ref_bytes) + w->gbase_bytes + w->pairs_bytes + w->rec_bytes) : 0; }

// a staged batch into the next free slot: copies straight from the image, no host work
int isx_pipe_submit_wire(isx_pipe *p, const isx_wire *w, int64_t *ticket)
{
    if (!p || !w || !ticket) { isx_set_error("isx_pipe_submit_wire: bad argument"); return ISX_ERR_ARG; }
    if (w->pipe != p) { isx_set_error("isx_pipe_submit_wire: the batch was staged for another pipe"); return ISX_ERR_ARG; }
    drain_stager(p);
    Slot &s = p->slots[(size_t)(p->next_ticket % (int64_t)p->slots.size())];
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (s.state != 0) { isx_set_error("isx_pipe_submit_wire: every slot is in use (collect + release the oldest batch first)"); return ISX_ERR_STATE; }
    }
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    HIP_TRY(hipSetDevice(c->device));
    const double t0 = now_ms();
    const bool dense = b->M == 1, linkage = p->prm.enable_linkage != 0;
    b->n_pos = w->n_pos; b->n_obs = w->n_bases; b->n_splits = w->n_splits; b->n_rec = (uint64_t)w->n_rec;
    s.ref_has_n = w->ref_has_n;
    b->sparse_out = dense && b->d_clon_list != nullptr;
    b->cov8_out = b->sparse_out && (double)b->n_obs < 16.0 * (double)w->n_pos;
    b->nib_out = b->cov8_out && b->lean && (double)b->n_obs < 6.0 * (double)w->n_pos;
    b->clon_dense = false;
    b->rare_dense = !(b->lean && b->sparse_out) || (double)b->n_obs * 4.0 >= (double)p->prm.rarefied_coverage * (double)b->n_pos;
    if (b->lev_sparse) b->lev_cov_bytes = (double)b->n_obs < 64.0 * (double)b->n_pos ? 1 : 2;      // (a level of a batch this shallow rarely reaches 255: the exact values of those that do travel in a list)
    b->n_pairs = w->n_pairs;
    b->packed = w->packed; b->W = w->W;
    b->n_win = (int)(w->win_bytes / sizeof(uint2));
    int rc = batch_set_geometry(b);
    if (rc != ISX_OK) return rc;
    if (!dense && !b->lev_sparse) {
        const size_t used = (size_t)b->n_win * b->slab;
        if (used > b->slab_region) { isx_set_error("internal: entry slabs larger than the slot's region"); return ISX_ERR_STATE; }
        b->cap_ovf = b->cap_entries - used;
    }
    b->d_bounds = reinterpret_cast<int64_t *>(s.d_in + s.off_bounds);
    b->d_win = reinterpret_cast<uint2 *>(s.d_in + s.off_win);
    uint8_t *const dref = w->d_ref ? w->d_ref : s.d_in + s.off_ref;         // (a kept reference: the wire's own device copy, nothing to bring in)
    b->d_ref = dref;
    b->d_ref_n = s.ref_has_n ? dref + ref2_bytes(b->n_pos) : nullptr;
    b->d_gbase = reinterpret_cast<uint32_t *>(s.d_in + s.off_gbase);
    b->d_seg = p->drec ? nullptr : reinterpret_cast<uint4 *>(s.d_in + s.off_rec);
    b->d_drec = p->drec ? reinterpret_cast<uint4 *>(s.d_in + s.off_rec) : nullptr;
    b->d_rec16 = nullptr; b->d_rec32 = nullptr;
    b->d_pair = linkage && !p->drec ? reinterpret_cast<uint32_t *>(s.d_in + s.off_pairs) : nullptr;
    b->d_pair_runs = nullptr; b->d_run_index = nullptr; b->n_runs = 0;
    HIP_TRY(hipEventRecord(s.ev_h2d0, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_bounds, w->h + w->o_bounds, w->bounds_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_win, w->h + w->o_win, w->win_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    if (!w->d_ref) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_ref, w->h + w->o_ref, w->ref_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_gbase, w->h + w->o_gbase, w->gbase_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    if (w->pairs_bytes) HIP_TRY(hipMemcpyAsync(s.d_in + s.off_pairs, w->h + w->o_pairs, w->pairs_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipMemcpyAsync(s.d_in + s.off_rec, w->h + w->o_rec, w->rec_bytes, hipMemcpyHostToDevice, H2D(p, s)));
    HIP_TRY(hipEventRecord(s.ev_h2d1, H2D(p, s)));
    s.h2d_split = false;
    s.h2d_bytes = isx_wire_bytes(w);
    s.encode_ms = (float)(now_ms() - t0);       // (what this submit itself spent on the host: enqueueing)
    s.encode_passes = 0;                        // staged ahead of time
    return enqueue_pass(p, s, w->n_pos, ticket);
}

int isx_pipe_submit_reads(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                          const isx_segs *segs, int64_t *ticket)
{
    if (!p || !ref || !split_bounds || !ticket || n_pos <= 0 || n_splits <= 0 || !segs || segs->n_seg < 0 ||
        (segs->n_seg && (!segs->gpos || !segs->len || !segs->bases))) {
        isx_set_error("isx_pipe_submit_reads: bad argument");
        return ISX_ERR_ARG;
    }
    if (!p->segs) { isx_set_error("isx_pipe_submit_reads: not a read-level pipe (isx_pipe_params.max_segs == 0)"); return ISX_ERR_STATE; }
    if (p->prm.enable_linkage && segs->n_seg && !segs->pair) { isx_set_error("linkage needs the pair array"); return ISX_ERR_ARG; }
    if (p->stager.joinable()) {
        // queued for the stager: only what can be said without touching the segments is checked here, the rest comes back
        // through isx_pipe_collect.  The caller's arrays stay its own and unchanged until that call (or isx_pipe_release).
        if (n_pos > p->pp.max_pos || segs->n_seg > p->pp.max_segs || n_splits > p->pp.max_splits) {
            isx_set_error("isx_pipe_submit_reads: batch larger than the pipe was created for");
            return ISX_ERR_CAPACITY;
        }
        isx_pipe::StageJob job;
        job.n_pos = n_pos; job.ref = ref; job.segs = *segs;
        if (!p->prm.enable_linkage) job.segs.pair = nullptr;
        job.bounds.assign(split_bounds, split_bounds + n_splits + 1);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            const int64_t t = p->next_promise, n = (int64_t)p->slots.size();
            const Slot &s = p->slots[(size_t)(t % n)];
            // the slot is free when the batch that had it last (ticket t - n) was staged, finished and released
            if (t - n >= p->next_ticket || s.state != 0) { isx_set_error("isx_pipe_submit_reads: every slot is in use (collect + release the oldest batch first)"); return ISX_ERR_STATE; }
            job.ticket = t;
            p->next_promise = t + 1;
            *ticket = t;
            p->stage_q.push_back(std::move(job));
        }
        p->cv_stage.notify_one();
        return ISX_OK;
    }
    isxenc::SegJob J;
    J.in = *segs; J.n_seg = segs->n_seg;
    if (!p->prm.enable_linkage) J.in.pair = nullptr;
    return submit_segs_common(p, n_pos, ref, n_splits, split_bounds, J, ticket);
}

int isx_pipe_submit_planes(isx_pipe *p, int64_t n_pos, const isx_ref_planes *ref, int32_t n_splits, const int64_t *split_bounds,
                           const isx_read_planes *reads, int64_t *ticket)
{
    if (!p || !ref || !ref->plane2 || !split_bounds || !ticket || n_pos <= 0 || n_splits <= 0 || !reads || reads->n_seg < 0 ||
        (reads->n_seg && (!reads->gpos || !reads->len || !reads->planes))) {
        isx_set_error("isx_pipe_submit_planes: bad argument");
        return ISX_ERR_ARG;
    }
    if (!p->segs || !p->drec) { isx_set_error("isx_pipe_submit_planes: bit-plane reads need a read-level pipe (max_segs > 0) with one mm bin (n_mm_bins == 1) or ISX_LAYOUT_MM_DELTA_RECORDS"); return ISX_ERR_STATE; }
    if (p->prm.enable_linkage && reads->n_seg && !reads->pair) { isx_set_error("linkage needs the pair array"); return ISX_ERR_ARG; }
    if (p->stager.joinable()) {         // queued for the stager (see isx_pipe_submit_reads): the caller's arrays stay valid and unchanged until collect / release
        if (n_pos > p->pp.max_pos || reads->n_seg > p->pp.max_segs || n_splits > p->pp.max_splits) {
            isx_set_error("isx_pipe_submit_planes: batch larger than the pipe was created for");
            return ISX_ERR_CAPACITY;
        }
        isx_pipe::StageJob job;
        job.n_pos = n_pos; job.ref = nullptr; job.planes = true; job.reads = *reads; job.rp = *ref;
        if (!p->prm.enable_linkage) job.reads.pair = nullptr;
        job.bounds.assign(split_bounds, split_bounds + n_splits + 1);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            const int64_t t = p->next_promise, n = (int64_t)p->slots.size();
            const Slot &s = p->slots[(size_t)(t % n)];
            if (t - n >= p->next_ticket || s.state != 0) { isx_set_error("isx_pipe_submit_planes: every slot is in use (collect + release the oldest batch first)"); return ISX_ERR_STATE; }
            job.ticket = t;
            p->next_promise = t + 1;
            *ticket = t;
            p->stage_q.push_back(std::move(job));
        }
        p->cv_stage.notify_one();
        return ISX_OK;
    }
    isxenc::SegJob J;
    J.in2 = *reads; J.n_seg = reads->n_seg;
    if (!p->prm.enable_linkage) J.in2.pair = nullptr;
    return submit_segs_common(p, n_pos, nullptr, n_splits, split_bounds, J, ticket, ref);
}

int isx_pipe_set_reference_budget(isx_pipe *p, int64_t mib)
{
    if (!p) { isx_set_error("isx_pipe_set_reference_budget: bad argument"); return ISX_ERR_ARG; }
    p->ref_cache_budget.store(mib < 0 ? 0 : (size_t)(mib ? mib : 4096) << 20, std::memory_order_relaxed);
    return ISX_OK;
}

int isx_pipe_submit(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                    int64_t n_obs, const isx_obs *obs, const uint32_t *pair, int64_t *ticket)
{
    if (!p || !ref || !split_bounds || !ticket || n_pos <= 0 || n_splits <= 0 || n_obs < 0 || (n_obs && !obs)) {
        isx_set_error("isx_pipe_submit: bad argument");
        return ISX_ERR_ARG;
    }
    if (p->prm.enable_linkage && n_obs && !pair) { isx_set_error("linkage needs the pair array"); return ISX_ERR_ARG; }
    if (p->segs) { isx_set_error("isx_pipe_submit: a read-level pipe takes isx_pipe_submit_reads / isx_pipe_submit_bam"); return ISX_ERR_STATE; }
    isxenc::EncodeJob J;
    J.obs = obs; J.pair = p->prm.enable_linkage ? pair : nullptr;
    static const isx_obs none{};
    if (!n_obs) J.obs = &none;
    return submit_common(p, n_pos, ref, n_splits, split_bounds, n_obs, J, ticket);
}

// A batch straight from the BAM front end: the references `refs` (ascending) of a scanned + filtered file are
// loaded, overlap-resolved and expanded INTO the slot's pinned staging -- the encoder pulls the observation stream
// group by group from the front end, so neither the 8-byte records nor the per-record pair ids of the batch are ever
// materialised (profile_utilities.py:150-153 + 268-286 feeding the device directly).
int isx_pipe_submit_bam(isx_pipe *p, isx_bam *bam, const struct isx_bam_params_s *bp, const int32_t *refs, int32_t n_refs,
                        const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds, struct isx_bam_info_s *info, int64_t *ticket)
{
    if (!p || !bam || !bp || !refs || n_refs <= 0 || !ref || !ticket) { isx_set_error("isx_pipe_submit_bam: bad argument"); return ISX_ERR_ARG; }
    BamBatch *q = nullptr;
    const double t_in = now_ms();
    drain_stager(p);
    int rc = bam_batch_prepare(bam, bp, refs, n_refs, &q, 0, -1, p->segs);
    if (rc != ISX_OK) return rc;
    const double t_prep = now_ms();
    std::unique_ptr<BamBatch, void (*)(BamBatch *)> Q(q, bam_batch_free);
    const int64_t n_pos = bam_batch_n_pos(q), n_obs = p->segs ? bam_batch_seg_bases(q) : bam_batch_n_obs(q);
    const std::vector<int64_t> &own = bam_batch_bounds(q);
    if (!split_bounds) { split_bounds = own.data(); n_splits = (int32_t)own.size() - 1; }     // iterate_splits of the front end
    if (n_splits <= 0) { isx_set_error("isx_pipe_submit_bam: no splits"); return ISX_ERR_ARG; }
    if (info) bam_batch_info(q, n_refs, info);
    if (p->segs) {
        // read-level pipe: the reads' segments are packed (3-bit codes, quality filter applied) straight into the encoder's
        // per-task scratch and from there into pinned staging -- no per-base records on the host at all
        isxenc::SegJob J;
        J.n_seg = bam_batch_n_segs(q);
        J.gpos_all = bam_batch_seg_gpos(q);
        J.want_pairs = p->prm.enable_linkage != 0;
        // (one mm bin: as bit planes -- the 4-bit seq maps straight onto the 2-bit plane and the stager XORs instead of unpacking)
        if (p->drec && !getenv("ISX_BAM_SEG_WORDS")) J.produce_planes = [q](int64_t first, int64_t count, uint32_t *g, uint8_t *l, uint8_t *m, uint32_t *pr, uint64_t *pl) { bam_batch_emit_planes(q, first, count, g, l, m, pr, pl); };
        else J.produce = [q](int64_t first, int64_t count, uint32_t *g, uint8_t *l, uint8_t *m, uint32_t *pr, uint32_t *b) { bam_batch_emit_segs(q, first, count, g, l, m, pr, b); };
        rc = submit_segs_common(p, n_pos, ref, n_splits, split_bounds, J, ticket);
    } else {
        isxenc::EncodeJob J;
        J.obs = nullptr;
        J.want_pairs = p->prm.enable_linkage != 0;
        J.produce = [q](int64_t first, uint32_t count, isx_obs *o, uint32_t *pr) { bam_batch_emit(q, first, count, o, pr); };
        rc = submit_common(p, n_pos, ref, n_splits, split_bounds, n_obs, J, ticket);
    }
    const double t_sub = now_ms();
    // giving a gigabyte-sized batch back to the system takes as long as encoding it: not on the caller's time -- and not while
    // the finisher works on this batch either (unmapping holds the process' address-space lock: every HIP call that maps or
    // allocates waits for it).  The finisher drops it once the batch's tables are home.
    if (rc == ISX_OK) {
        std::lock_guard<std::mutex> lk(p->mu);
        Slot &s = p->slots[(size_t)(*ticket % (int64_t)p->slots.size())];
        if (s.ticket == *ticket && s.state == 1) s.dead_batch = Q.release();
    }
    if (Q) bam_batch_retire(Q.release());
    if (getenv("ISX_PIPE_TIMING"))      // tuning aid (stderr only)
        fprintf(stderr, "[isx_pipe_submit_bam] prepare %.1f ms, encode + enqueue %.1f ms, free %.1f ms\n", t_prep - t_in, t_sub - t_prep, now_ms() - t_sub);
    return rc;
}

int isx_pipe_collect(isx_pipe *p, int64_t ticket, isx_pipe_result *out)
{
    if (!p || !out || ticket < 0) { isx_set_error("isx_pipe_collect: bad argument"); return ISX_ERR_ARG; }
    Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    isx_ctx *c = p->ctx;
    isx_batch *b = s.b;
    const bool dense = b->M == 1;
    *out = isx_pipe_result{};
    {
        const double t0 = now_ms();
        std::unique_lock<std::mutex> lk(p->mu);
        if (ticket < p->next_promise) p->cv_done.wait(lk, [&] { return p->next_ticket > ticket; });     // queued for the stager: staged first
        if (s.ticket != ticket || s.state == 0) { isx_set_error("isx_pipe_collect: unknown or already released ticket"); return ISX_ERR_STATE; }
        p->cv_done.wait(lk, [&] { return s.state == 2; });
        out->collect_wait_ms = (float)(now_ms() - t0);
        if (s.rc != ISX_OK) { isx_set_error(s.err); return s.rc; }
    }
    HIP_TRY(hipSetDevice(c->device));
    out->ticket = ticket;
    out->n_pos = b->n_pos; out->n_obs = b->n_obs;
    out->sizes = b->sizes;
    out->snv = (size_t)b->sizes.n_snv > p->snv_prefix ? s.snv_big.data() : reinterpret_cast<const isx_snv *>(s.h_small + s.o_snv);
    if (dense) {
        if (s.cov4) {
            out->coverage4 = s.h_out + s.o_cov16;
            out->cov_rows = reinterpret_cast<const uint16_t *>(s.h_out + s.o_cov16 + s.cov_rows_off);
            out->cov_row_window = s.cov_row_win.data();
            out->n_cov_rows = (int64_t)s.cov_row_win.size();
            out->cov_window = s.cov_window;
        } else if (s.cov8) out->coverage8 = s.h_out + s.o_cov16;
        else out->coverage16 = reinterpret_cast<const uint16_t *>(s.h_out + s.o_cov16);
        if (s.clon_sparse) { out->clon_sparse = reinterpret_cast<const isx_rare *>(s.h_out + s.o_clon); out->n_clon = (int64_t)b->n_clon; }
        else out->clon = reinterpret_cast<const float *>(s.h_out + s.o_clon);
        out->n_saturated = b->n_sat;
        out->saturated = s.sat_complete ? s.sat_rows.data() : nullptr;
        if (p->prm.rarefied_coverage > 0) {
            out->n_rare = (int64_t)b->n_rare;
            if (!s.rare_dense) out->rare = s.rare_big.empty() ? reinterpret_cast<const isx_rare *>(s.h_small + s.o_rare) : s.rare_big.data();
            if (p->pp.want_counts) out->clon_rarefied = reinterpret_cast<const float *>(s.h_out + s.o_clonr);
            else if (s.rare_dense) out->clon_rarefied = s.clonr_big.data();
        }
        if (p->pp.want_counts) out->counts = reinterpret_cast<const uint32_t *>(s.h_out + s.o_counts);
    }
    if (b->lev_sparse) {
        out->lev_mask = s.h_out + s.o_lmask;
        out->lev_win_off = reinterpret_cast<const uint32_t *>(s.h_out + s.o_lwin);
        out->lev_cov = s.lcov_big.empty() ? s.h_out + s.o_lcov : s.lcov_big.data();
        out->lev_clon = s.lclon_big.empty() ? reinterpret_cast<const isx_rare *>(s.h_out + s.o_lclon) : s.lclon_big.data();
        out->lev_rare = s.lrare_big.empty() ? reinterpret_cast<const isx_rare *>(s.h_out + s.o_lrare) : s.lrare_big.data();
        out->lev_sat = s.sat_rows.data();
        out->n_lev = b->sizes.n_entries; out->n_lev_clon = (int64_t)b->n_clon; out->n_lev_sat = (int64_t)b->n_sat;
        out->n_lev_rare = p->prm.rarefied_coverage > 0 ? (int64_t)b->n_rare : 0;
        out->lev_mask_bytes = b->lev_mask_bytes; out->lev_cov_bytes = b->lev_cov_bytes;
        out->lev_window = b->W; out->n_lev_windows = b->n_win; out->lev_min_cov = p->prm.min_cov;
    }
    out->batch = b;
    out->encode_ms = s.encode_ms;
    out->encode_passes = s.encode_passes;
    out->record_bytes = p->rb;
    out->h2d_bytes = s.h2d_bytes; out->d2h_bytes = s.d2h_bytes;
    out->ld = p->prm.enable_linkage ? s.ld_rows.data() : nullptr;
    out->rows_checksum = s.rows_checksum;
    float ms = 0.f;
    if (s.h2d_split) {                          // the time the DMA engine worked for this batch, not the stager's time between its two parts
        float m1 = 0.f, m2 = 0.f;
        if (hipEventElapsedTime(&m1, s.ev_h2d0, s.ev_h2da) == hipSuccess && hipEventElapsedTime(&m2, s.ev_h2db, s.ev_h2d1) == hipSuccess) out->h2d_ms = m1 + m2;
    } else if (hipEventElapsedTime(&ms, s.ev_h2d0, s.ev_h2d1) == hipSuccess) out->h2d_ms = ms;
    if (hipEventElapsedTime(&ms, s.ev_d2h0, s.ev_d2h1) == hipSuccess) out->d2h_ms = ms;
    isx_timings t{};
    if (isx_batch_timings(b, &t) == ISX_OK) out->kernel_ms = t.pileup_ms;
    return ISX_OK;
}

// The entry table of a collected mm batch (isx_batch_fetch_entries on the slot, but the 32 bytes per entry go through the
// slot's idle pinned input staging in two alternating pieces and are moved into `out` by the pipe's host threads: a blocking
// copy into pageable memory runs at a fifth of the link).
static int pipe_fetch_entries(isx_pipe *p, int64_t ticket, isx_entry *out, const EntrySoa *soa);

int isx_pipe_fetch_entries(isx_pipe *p, int64_t ticket, isx_entry *out)
{
    if (!out) { isx_set_error("isx_pipe_fetch_entries: bad argument"); return ISX_ERR_ARG; }
    return pipe_fetch_entries(p, ticket, out, nullptr);
}

int isx_pipe_fetch_entries_shrunk(isx_pipe *p, int64_t ticket, uint32_t *gpos, uint32_t *mm_cov, float *clon, float *clon_rarefied)
{
    if (!gpos || !mm_cov || !clon || !clon_rarefied) { isx_set_error("isx_pipe_fetch_entries_shrunk: bad argument"); return ISX_ERR_ARG; }
    const EntrySoa soa{gpos, mm_cov, clon, clon_rarefied};
    return pipe_fetch_entries(p, ticket, nullptr, &soa);
}

static int pipe_fetch_entries(isx_pipe *p, int64_t ticket, isx_entry *out, const EntrySoa *soa)
{
    if (!p || ticket < 0) { isx_set_error("isx_pipe_fetch_entries: bad argument"); return ISX_ERR_ARG; }
    Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (s.ticket != ticket || s.state != 2 || s.rc != ISX_OK) { isx_set_error("isx_pipe_fetch_entries: collect the batch first"); return ISX_ERR_STATE; }
    }
    isx_batch *b = s.b;
    if (b->M == 1) { isx_set_error("n_mm_bins == 1: the dense tables come with isx_pipe_collect"); return ISX_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    const size_t n = (size_t)b->sizes.n_entries;
    if (!n) return ISX_OK;
    if (b->lev_sparse && soa) {             // the four columns are made on the host from what already came home with the batch
        isx_pipe_result r;
        const int rc = isx_pipe_collect(p, ticket, &r);
        if (rc != ISX_OK) return rc;
        return isx_levels_expand(&r, std::max(1, std::min(p->pool ? p->pool->size() : 1, 16)), soa->gpos, soa->mm_cov, soa->clon, soa->clon_rarefied);
    }
    if (b->lev_sparse && !b->d_entries) {
        isx_set_error("isx_pipe_fetch_entries: a lean slot keeps no 32-byte entries (the levels come with isx_pipe_collect: isx_pipe_result.lev_*, isx_levels_expand)");
        return ISX_ERR_STATE;
    }
    const size_t region = p->ring_half ? 2 * (size_t)p->ring_half * p->rb : (size_t)p->cap_rec * p->rb;
    const size_t piece = std::min<size_t>((size_t)32 << 20, region / 2 / 4096 * 4096);
    if (piece < ((size_t)1 << 20))              // tiny pipe: the staging detour is not worth it
        return fetch_entries_sorted(p->ctx->stream, b->d_entries, b->d_win_nent, (uint32_t)b->slab, entry_wins(b), b->n_ovf, n, out, nullptr, soa);
    uint8_t *bounce[2] = {s.h_in + s.off_rec, s.h_in + s.off_rec + piece};
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (hipEvent_t &e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const EntryCopier copier = [&](const void *dsrc, void *hdst, size_t bytes, hipStream_t st) -> int {
        const size_t n_pieces = (bytes + piece - 1) / piece;
        auto issue = [&](size_t k) -> hipError_t {
            const size_t off = k * piece, len = std::min(piece, bytes - off);
            hipError_t e = isx_copy_to_host(bounce[k & 1], static_cast<const uint8_t *>(dsrc) + off, len, st);
            return e == hipSuccess ? hipEventRecord(ev[k & 1], st) : e;
        };
        HIP_TRY(issue(0));
        for (size_t k = 0; k < n_pieces; k++) {
            if (k + 1 < n_pieces) HIP_TRY(issue(k + 1));        // its half was emptied by the threads one step ago
            HIP_TRY(isx_wait_event(ev[k & 1]));
            const size_t off = k * piece, len = std::min(piece, bytes - off);
            const size_t sub = (size_t)1 << 20;
            const uint8_t *src = bounce[k & 1];
            uint8_t *dst = static_cast<uint8_t *>(hdst) + off;
            p->pool->run((int)((len + sub - 1) / sub), [&](int t) {
                const size_t a = (size_t)t * sub;
                memcpy(dst + a, src + a, std::min(sub, len - a));
            });
        }
        return ISX_OK;
    };
    const int rc = fetch_entries_sorted(p->ctx->stream, b->d_entries, b->d_win_nent, (uint32_t)b->slab, entry_wins(b), b->n_ovf, n, out, &copier, soa);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    return rc;
}

int isx_pipe_release(isx_pipe *p, int64_t ticket)
{
    if (!p || ticket < 0) { isx_set_error("isx_pipe_release: bad argument"); return ISX_ERR_ARG; }
    Slot &s = p->slots[(size_t)(ticket % (int64_t)p->slots.size())];
    std::unique_lock<std::mutex> lk(p->mu);
    if (ticket < p->next_promise) p->cv_done.wait(lk, [&] { return p->next_ticket > ticket; });
    if (s.ticket != ticket || s.state == 0) { isx_set_error("isx_pipe_release: unknown or already released ticket"); return ISX_ERR_STATE; }
    p->cv_done.wait(lk, [&] { return s.state == 2; });     // never collected: its queued work drains before the slot is reused
    s.state = 0;
    return ISX_OK;
}

int isx_encode_obs_ring(const isx_obs *obs, const uint32_t *pair, int64_t n_obs, int64_t n_pos, int32_t record_bytes,
                        int32_t host_threads, double slack, int64_t cap_rec, int64_t ring_records, void *rec, uint32_t *gbase,
                        uint32_t *pair_out, int64_t *n_rec, int32_t *passes)
{
    if ((n_obs && !obs) || n_obs < 0 || n_pos <= 0 || (record_bytes != 2 && record_bytes != 4) || !rec || !gbase || !n_rec ||
        cap_rec < ISX_PAD || (cap_rec % ISX_PAD) || (pair && !pair_out) || ring_records < 0 || (ring_records % (2 * ISX_PAD)) ||
        (ring_records && pair)) {
        isx_set_error("isx_encode_obs: bad argument");
        return ISX_ERR_ARG;
    }
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    const size_t n_chunks = (size_t)(cap_rec / ISX_CHUNK) + 2;
    std::vector<uint32_t> cmin(n_chunks), cmax(n_chunks);
    std::vector<uint8_t> cany(n_chunks);
    isxenc::EncodeJob J;
    J.obs = obs; J.pair = pair; J.n_obs = n_obs; J.n_pos = n_pos; J.record_bytes = record_bytes;
    J.rec = rec; J.gbase = gbase; J.pair_out = pair ? pair_out : nullptr;
    J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data();
    J.cap_rec = cap_rec; J.slack = slack;
    std::vector<uint8_t> ring;
    if (ring_records) {         // the pipe's ring mode with a memcpy standing in for the DMA engine
        const int64_t G = record_bytes == 2 ? ISX_GROUP16 : ISX_GROUP;
        const int64_t half = ring_records / 2;
        ring.assign((size_t)ring_records * record_bytes, 0xAB);
        J.rec = ring.data();
        J.ring_groups = half / G;
        J.wave_begin = [](int) {};
        J.wave_flush = [&](int h, int64_t g0, int64_t g1) {
            memcpy(static_cast<uint8_t *>(rec) + (size_t)g0 * G * record_bytes, ring.data() + (size_t)h * half * record_bytes,
                   (size_t)(g1 - g0) * G * record_bytes);
            memset(ring.data() + (size_t)h * half * record_bytes, 0xAB, (size_t)half * record_bytes);    // stale data must never travel
        };
    }
    const int erc = isxenc::encode_obs(pool, J);
    if (erc == isxenc::ENC_CAPACITY) { isx_set_error("isx_encode_obs: cap_rec too small for this stream"); return ISX_ERR_CAPACITY; }
    if (erc == isxenc::ENC_MM_RANGE) { isx_set_error("an observation has mm >= 256"); return ISX_ERR_MM_RANGE; }
    if (erc == isxenc::ENC_BAD_POS) { isx_set_error("observation gpos >= n_pos"); return ISX_ERR_ARG; }
    *n_rec = J.n_rec;
    if (passes) *passes = J.passes;
    return ISX_OK;
}

int isx_encode_obs(const isx_obs *obs, const uint32_t *pair, int64_t n_obs, int64_t n_pos, int32_t record_bytes,
                   int32_t host_threads, double slack, int64_t cap_rec, void *rec, uint32_t *gbase, uint32_t *pair_out,
                   int64_t *n_rec, int32_t *passes)
{
    return isx_encode_obs_ring(obs, pair, n_obs, n_pos, record_bytes, host_threads, slack, cap_rec, 0, rec, gbase, pair_out, n_rec, passes);
}

}  // extern "C"
