// isx_api.hip -- C ABI of libinstrain_amd.so: context, resident batches, run, fetch.
// (see include/instrain_amd.h for what each entry point replaces in the reference)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <thread>
#include <type_traits>
#include <vector>
#include <unistd.h>
#include <sys/prctl.h>
#include <time.h>

#include "isx_batch.h"
#include "seg_encode.h"

static thread_local std::string g_err;
void isx_set_error(const std::string &msg) { g_err = msg; }

// ---- caching allocators (see isx_internal.h): device memory, and pinned host memory ----
namespace {
struct BlockCache {
    struct Key {
        int dev; size_t cls;
        bool operator<(const Key &o) const { return dev != o.dev ? dev < o.dev : cls < o.cls; }
    };
    std::mutex mu;
    std::multimap<Key, void *> free_blocks;             // (device, class size) -> block: a block only goes back to a context on ITS device
    std::unordered_map<void *, Key> live;               // block -> where it came from
    std::map<int, size_t> cached;                       // bytes kept per device
    const bool per_device;                              // device memory: keyed by the current device; pinned memory: one pool
    size_t (*const limit_of)(int dev);
    hipError_t (*const raw_alloc)(void **, size_t);
    hipError_t (*const raw_free)(void *);
    BlockCache(bool per_dev, size_t (*lim)(int), hipError_t (*a)(void **, size_t), hipError_t (*f)(void *))
        : per_device(per_dev), limit_of(lim), raw_alloc(a), raw_free(f) {}
    static size_t class_of(size_t bytes)
    {
        size_t v = std::max<size_t>(bytes, 4096);
        int e = 0;
        while ((v >> e) > 15) e++;                      // mantissa of 4 bits: classes 12.5 % apart at most
        const size_t m = (v + (((size_t)1 << e) - 1)) >> e;
        return m << e;
    }
    int current() const
    {
        int d = -1;
        if (per_device && hipGetDevice(&d) != hipSuccess) d = -1;
        return d;
    }
    // blocks of `dev` only (dev < -1: every device)
    void trim(int dev = -2)
    {
        std::vector<std::pair<int, void *>> dead;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto it = free_blocks.begin(); it != free_blocks.end();) {
                if (dev < -1 || it->first.dev == dev) {
                    dead.emplace_back(it->first.dev, it->second);
                    cached[it->first.dev] -= it->first.cls;
                    it = free_blocks.erase(it);
                } else ++it;
            }
        }
        int keep = -1;
        const bool sw = per_device && hipGetDevice(&keep) == hipSuccess;
        for (auto &d : dead) {
            if (sw && d.first >= 0) (void)hipSetDevice(d.first);
            (void)raw_free(d.second);
        }
        if (sw && keep >= 0) (void)hipSetDevice(keep);
    }
    hipError_t get(void **p, size_t bytes)
    {
        const Key k{current(), class_of(bytes)};
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_blocks.find(k);
            if (it != free_blocks.end()) {
                *p = it->second;
                free_blocks.erase(it);
                cached[k.dev] -= k.cls;
                live[*p] = k;
                return hipSuccess;
            }
        }
        hipError_t e = raw_alloc(p, k.cls);
        if (e != hipSuccess) {              // give the cache back and try once more
            (void)hipGetLastError();
            trim();
            e = raw_alloc(p, k.cls);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> lk(mu);
        live[*p] = k;
        return hipSuccess;
    }
    void put(void *p)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it != live.end()) {
                const Key k = it->second;
                live.erase(it);
                if (cached[k.dev] + k.cls <= limit_of(k.dev)) { free_blocks.emplace(k, p); cached[k.dev] += k.cls; return; }
            }
        }
        (void)raw_free(p);
    }
};
hipError_t raw_dev_alloc(void **p, size_t n) { return hipMalloc(p, n); }
hipError_t raw_dev_free(void *p) { return hipFree(p); }
hipError_t raw_pin_alloc(void **p, size_t n) { return hipHostMalloc(p, n, hipHostMallocDefault); }
hipError_t raw_pin_free(void *p) { return hipHostFree(p); }
// What may stay cached: a sixth of the device's memory (48 GiB of a 288 GB MI355X; less on a smaller part), an eighth of the
// host's RAM up to 8 GiB pinned.  Asked once per device.
size_t dev_limit(int dev)
{
    static std::mutex mu;
    static std::map<int, size_t> lim;
    std::lock_guard<std::mutex> lk(mu);
    auto it = lim.find(dev);
    if (it != lim.end()) return it->second;
    size_t v = (size_t)8 << 30;
    hipDeviceProp_t pr;
    if (dev >= 0 && hipGetDeviceProperties(&pr, dev) == hipSuccess) v = pr.totalGlobalMem / 6;
    else (void)hipGetLastError();
    lim[dev] = v;
    return v;
}
size_t pin_limit(int)
{
    static const size_t v = [] {
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
        const size_t ram = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)64 << 30;
        return std::min<size_t>((size_t)8 << 30, ram / 8);
    }();
    return v;
}
BlockCache &dev_cache() { static BlockCache c(true, dev_limit, raw_dev_alloc, raw_dev_free); return c; }
BlockCache &pin_cache() { static BlockCache c(false, pin_limit, raw_pin_alloc, raw_pin_free); return c; }
}  // namespace

hipError_t isx_dev_malloc(void **p, size_t bytes) { return dev_cache().get(p, bytes); }
void isx_dev_free(void *p) { dev_cache().put(p); }
hipError_t isx_pin_malloc(void **p, size_t bytes) { return pin_cache().get(p, bytes); }
void isx_pin_free(void *p) { pin_cache().put(p); }
void isx_dev_trim() { dev_cache().trim(); pin_cache().trim(); }
// An allocation that does not go through the cache (one-shot batches, tables that only grow): when the device is full because
// of what the cache keeps, give that back and try once more.
hipError_t isx_raw_dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dev_cache().trim();
        e = hipMalloc(p, bytes);
    }
    return e;
}

// ---- host ranges registered for the copy engine (isx_host_register) ----
namespace {
struct HostRange { uintptr_t a; size_t n; };
std::mutex g_hreg_mu;
std::vector<HostRange> g_hreg;
}  // namespace

bool isx_host_is_registered(const void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    std::lock_guard<std::mutex> lk(g_hreg_mu);
    for (const HostRange &r : g_hreg) if (a >= r.a && a + bytes <= r.a + r.n) return true;
    return false;
}

extern "C" int isx_host_register(const void *ptr, int64_t bytes)
{
    if (!ptr || bytes <= 0) { isx_set_error("isx_host_register: bad argument"); return ISX_ERR_ARG; }
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    // pinning works on pages: two registrations must not share one (the runtime's answer to that is not an error code one can rely on)
    const uintptr_t pa = a & ~(uintptr_t)4095, pe = (a + (size_t)bytes + 4095) & ~(uintptr_t)4095;
    std::lock_guard<std::mutex> lk(g_hreg_mu);
    for (const HostRange &r : g_hreg) {
        const uintptr_t ra = r.a & ~(uintptr_t)4095, re = (r.a + r.n + 4095) & ~(uintptr_t)4095;
        if (pa < re && ra < pe) { isx_set_error("isx_host_register: the range shares a page with a registered one"); return ISX_ERR_ARG; }
    }
    HIP_TRY(hipHostRegister(const_cast<void *>(ptr), (size_t)bytes, hipHostRegisterDefault));
    g_hreg.push_back(HostRange{a, (size_t)bytes});
    return ISX_OK;
}

extern "C" int isx_host_unregister(const void *ptr)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    std::lock_guard<std::mutex> lk(g_hreg_mu);
    for (size_t i = 0; i < g_hreg.size(); i++)
        if (g_hreg[i].a == a) {
            g_hreg.erase(g_hreg.begin() + (long)i);
            HIP_TRY(hipHostUnregister(const_cast<void *>(ptr)));
            return ISX_OK;
        }
    isx_set_error("isx_host_unregister: not a registered range");
    return ISX_ERR_ARG;
}

namespace {
__global__ void __launch_bounds__(256) k_copy_out(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src) + i);
        reinterpret_cast<u32x4 *>(dst)[i] = v;
    }
}
}  // namespace

namespace {
template <class T>
__global__ void __launch_bounds__(256) k_copy_small(T *__restrict__ dst, const T *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
}  // namespace

hipError_t isx_copy_to_host(void *hdst, const void *dsrc, size_t bytes, hipStream_t stream)
{
    // position-sized tables: the DMA engine.  A copy KERNEL that streams megabytes into host memory keeps the memory system's queues
    // full of PCIe writes: every other kernel's HBM traffic waits behind them and the copy-in DMA loses a fifth of its rate
    // (whole-database stream: 82 -> 55 ms a pass without the coverage rows' copy kernel, rate of the copy-in 43 -> 54 GB/s; a 16 MiB
    // threshold, tried in round 6: headline 156-161 -> 153 Gbp/s, collect_wait 0.4 -> 4 ms)
    static const size_t dma_min = [] { const char *e = getenv("ISX_D2H_DMA_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 20; }();
    return isx_copy_to_host_route(hdst, dsrc, bytes, stream, bytes < dma_min);
}

// by_kernel: the copy kernels whatever the size.  What a SMALL batch hands back (the level tables of a C2 batch with mm profiling on: 5 + 12 MB)
// leaves by kernel although its pieces are megabytes: hipMemcpyAsync serves both directions of this stack's copies from one queue, and those
// tables going home by DMA halved the rate of the next batches' copy-in (0.94 -> 0.69 ms a batch; C2 mm-on stream 93.5 -> 124 Gbp/s, round 6)
hipError_t isx_copy_to_host_route(void *hdst, const void *dsrc, size_t bytes, hipStream_t stream, bool by_kernel)
{
    if (!bytes) return hipSuccess;
    if (!by_kernel) return hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, stream);
    const uintptr_t both = reinterpret_cast<uintptr_t>(hdst) | reinterpret_cast<uintptr_t>(dsrc);
    size_t done = 0;
    if ((both & 15) == 0 && bytes >= 4096) {
        const size_t n16 = bytes / 16;
        const int blocks = (int)std::min<size_t>(128, (n16 + 1023) / 1024);
        hipLaunchKernelGGL(k_copy_out, dim3(blocks), dim3(256), 0, stream, static_cast<uint4 *>(hdst), static_cast<const uint4 *>(dsrc), n16);
        done = n16 * 16;
    }
    if (done < bytes) {                     // small pieces and tails: words when they line up, bytes otherwise -- never the DMA engine
        const size_t rest = bytes - done;
        uint8_t *d = static_cast<uint8_t *>(hdst) + done;
        const uint8_t *s = static_cast<const uint8_t *>(dsrc) + done;
        const bool words = (((both | rest) & 3) == 0);
        const size_t n = words ? rest / 4 : rest;
        const int blocks = (int)std::min<size_t>(64, (n + 255) / 256);
        if (words) hipLaunchKernelGGL(k_copy_small<uint32_t>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(d), reinterpret_cast<const uint32_t *>(s), n);
        else hipLaunchKernelGGL(k_copy_small<uint8_t>, dim3(blocks), dim3(256), 0, stream, d, s, n);
    }
    return hipGetLastError();
}

namespace {
__global__ void __launch_bounds__(256) k_copy_counted(uint4 *__restrict__ dst, const uint4 *__restrict__ src, const uint32_t *cursor,
                                                      uint32_t base, uint32_t row_bytes, uint64_t cap_rows)
{
    const uint64_t rows = min((uint64_t)(*cursor - base), cap_rows);
    const uint64_t n16 = (rows * row_bytes + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

hipError_t isx_copy_rows_to_host(void *hdst, const void *dsrc, const uint32_t *d_cursor, uint32_t base, uint32_t row_bytes,
                                 size_t cap_rows, hipStream_t stream)
{
    if (!cap_rows) return hipSuccess;
    if (((reinterpret_cast<uintptr_t>(hdst) | reinterpret_cast<uintptr_t>(dsrc)) & 15) != 0) return hipErrorInvalidValue;
    const size_t n16 = (cap_rows * row_bytes + 15) / 16;
    const int blocks = (int)std::min<size_t>(64, (n16 + 1023) / 1024);
    hipLaunchKernelGGL(k_copy_counted, dim3(std::max(blocks, 1)), dim3(256), 0, stream, static_cast<uint4 *>(hdst), static_cast<const uint4 *>(dsrc),
                       d_cursor, base, row_bytes, (uint64_t)cap_rows);
    return hipGetLastError();
}

// several tables of one batch to pinned host memory in ONE launch (a small batch's level tables: mask, window offsets, coverage bytes, lists)
namespace {
struct CopyJobs { uint4 *dst[ISX_COPY_JOBS]; const uint4 *src[ISX_COPY_JOBS]; uint32_t n16[ISX_COPY_JOBS]; uint32_t first[ISX_COPY_JOBS + 1]; int n; };
__global__ void __launch_bounds__(256) k_copy_multi(const CopyJobs J)
{
    int j = 0;
    while (j + 1 < J.n && blockIdx.x >= J.first[j + 1]) j++;
    const uint32_t nb = J.first[j + 1] - J.first[j], lb = blockIdx.x - J.first[j];
    for (uint32_t i = lb * 256u + threadIdx.x; i < J.n16[j]; i += nb * 256u) J.dst[j][i] = J.src[j][i];
}
}  // namespace

hipError_t isx_copy_multi_to_host(const isx_copy_job *jobs, int n, hipStream_t stream)
{
    CopyJobs J{};
    uint32_t blocks = 0;
    for (int i = 0; i < n && J.n < ISX_COPY_JOBS; i++) {
        if (!jobs[i].bytes) continue;
        if (((reinterpret_cast<uintptr_t>(jobs[i].dst) | reinterpret_cast<uintptr_t>(jobs[i].src)) & 15) != 0 || jobs[i].bytes > ((size_t)1 << 35)) return hipErrorInvalidValue;
        const uint32_t n16 = (uint32_t)((jobs[i].bytes + 15) / 16);         // (whole 16-byte pieces: both ends have the room)
        J.dst[J.n] = static_cast<uint4 *>(jobs[i].dst); J.src[J.n] = static_cast<const uint4 *>(jobs[i].src); J.n16[J.n] = n16;
        J.first[J.n] = blocks;
        blocks += std::max<uint32_t>(1, std::min<uint32_t>(64, (n16 + 1023) / 1024));
        J.n++;
    }
    if (!J.n) return hipSuccess;
    J.first[J.n] = blocks;
    hipLaunchKernelGGL(k_copy_multi, dim3(blocks), dim3(256), 0, stream, J);
    return hipGetLastError();
}

namespace {
struct ReadBack {
    uint8_t *pin = nullptr;
    size_t used = 0;
    struct Item { void *dst; size_t off, bytes; };
    std::vector<Item> items;
    // batching (isx_read_batch): the copies of the pending items are launched together by isx_read_sync -- one kernel instead of one each
    bool batching = false;
    std::vector<isx_copy_job> jobs;
    static constexpr size_t CAP = (size_t)1 << 20;
    ~ReadBack() { if (pin) isx_pin_free(pin); }
};
thread_local ReadBack g_rb;
}  // namespace

void isx_read_batch(bool on) { g_rb.batching = on; }

hipError_t isx_read_back(void *host_dst, const void *dsrc, size_t bytes, hipStream_t stream)
{
    if (!bytes) return hipSuccess;
    ReadBack &rb = g_rb;
    if (!rb.pin && isx_pin_malloc(reinterpret_cast<void **>(&rb.pin), ReadBack::CAP) != hipSuccess) rb.pin = nullptr;
    const size_t off = (rb.used + 15) & ~(size_t)15;
    if (!rb.pin || off + ((bytes + 15) & ~(size_t)15) > ReadBack::CAP) {
        const hipError_t e = hipMemcpyAsync(host_dst, dsrc, bytes, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) { rb.items.clear(); rb.jobs.clear(); rb.used = 0; }             // nothing may stay pending on a failing exit
        return e;
    }
    if (rb.batching && (reinterpret_cast<uintptr_t>(dsrc) & 15) == 0 && rb.jobs.size() < ISX_COPY_JOBS) {
        // (the source must not change before isx_read_sync: the caller's promise when it turns batching on)
        rb.jobs.push_back(isx_copy_job{rb.pin + off, dsrc, bytes});
    } else {
        const hipError_t e = isx_copy_to_host(rb.pin + off, dsrc, bytes, stream);
        if (e != hipSuccess) { rb.items.clear(); rb.jobs.clear(); rb.used = 0; return e; }      // (the caller returns without a sync: nothing may stay pending)
    }
    rb.items.push_back({host_dst, off, bytes});
    rb.used = off + bytes;
    return hipSuccess;
}

void isx_read_drop()
{
    g_rb.items.clear();
    g_rb.jobs.clear();
    g_rb.batching = false;
    g_rb.used = 0;
}

// Host threads that wait for the device (a pipe's finishers between the stages of a linkage chain and in front of the copy-out, the
// bounce copies) SLEEP between polls instead of calling the runtime's own waits: those spin by default, which on a cpu-limited lease (a
// cgroup quota) takes the cpus the pipe's stager threads need -- measured on a 16-cpu lease: ~1.5 cpus of spinning, the cgroup throttling the
// whole process 25-35 ms of every 64 ms pass.  (hipDeviceScheduleBlockingSync would do the same through the completion interrupt, but it is a
// device-wide setting of the host process and a three-rank job sharing one GPU hung under it.)  A few polls back to back for the short waits,
// then naps that grow to 200 us; ISX_ACTIVE_WAIT=1 calls the runtime's waits as before.
static bool isx_active_wait()
{
    static const bool a = getenv("ISX_ACTIVE_WAIT") != nullptr;
    return a;
}
template <class Ready>
static hipError_t isx_nap_until(Ready ready)
{
    // The naps must end on time (default timer slack: 50 us).  The slack is a property of the calling THREAD, and that may be the embedding
    // application's (the Python main thread in collect / fetch): it is lowered for the duration of this wait only and put back.
    long old_slack = -1;
    for (int i = 0;; i++) {
        const hipError_t r = ready();
        if (r != hipErrorNotReady) {
            if (old_slack > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack, 0, 0, 0);
            return r;
        }
        if (i < 4) continue;
        if (old_slack < 0) {
            old_slack = (long)prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
            if (old_slack > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); else old_slack = 0;
        }
        const long us = i < 40 ? 20 : (i < 100 ? 50 : 200);
        struct timespec ts = {0, us * 1000L};
        nanosleep(&ts, nullptr);
    }
}
hipError_t isx_wait_event(hipEvent_t e)
{
    if (isx_active_wait()) return hipEventSynchronize(e);
    return isx_nap_until([&] { return hipEventQuery(e); });
}
// hipStreamSynchronize waits for the work queued BEFORE the call; polling hipStreamQuery would wait until the stream has drained, also of
// what other threads enqueue meanwhile (a pass queue shared by the slots of a deep pipe: more latency, starvation in principle -- ADVICE r5).
// So: an event recorded at the call, polled.  The events are the calling thread's own (created on first use, one per device).
hipError_t isx_wait_stream(hipStream_t s)
{
    if (isx_active_wait()) return hipStreamSynchronize(s);
    struct Ev { hipEvent_t e = nullptr; int dev = -1; ~Ev() { if (e) (void)hipEventDestroy(e); } };
    static thread_local Ev ev;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (ev.e && ev.dev != dev) { (void)hipEventDestroy(ev.e); ev.e = nullptr; }
    if (!ev.e) {
        if (hipEventCreateWithFlags(&ev.e, hipEventDisableTiming) != hipSuccess) { ev.e = nullptr; (void)hipGetLastError(); return isx_nap_until([&] { return hipStreamQuery(s); }); }
        ev.dev = dev;
    }
    const hipError_t r = hipEventRecord(ev.e, s);
    if (r != hipSuccess) return r;
    return isx_nap_until([&] { return hipEventQuery(ev.e); });
}

hipError_t isx_read_sync(hipStream_t stream)
{
    ReadBack &rb = g_rb;
    hipError_t e = hipSuccess;
    if (!rb.jobs.empty()) { e = isx_copy_multi_to_host(rb.jobs.data(), (int)rb.jobs.size(), stream); rb.jobs.clear(); }
    if (e == hipSuccess) e = isx_wait_stream(stream);
    if (e == hipSuccess) for (const auto &it : rb.items) memcpy(it.dst, rb.pin + it.off, it.bytes);
    rb.items.clear();
    rb.used = 0;
    return e;
}

// host -> device through the two pinned staging buffers (hipMemcpyAsync, double-buffered);
// `fill(dst, first, count)` writes `count` elements starting at element `first` into dst.
template <class T, class F>
static int staged_upload(isx_ctx *c, T *d_dst, uint64_t n, F fill)
{
    const uint64_t per = c->pin_bytes / sizeof(T);
    int k = 0;
    for (uint64_t off = 0; off < n; off += per, k ^= 1) {
        const uint64_t cnt = std::min<uint64_t>(per, n - off);
        HIP_TRY(isx_wait_event(c->pin_ev[k]));
        fill(reinterpret_cast<T *>(c->pin[k]), off, cnt);
        HIP_TRY(hipMemcpyAsync(d_dst + off, c->pin[k], cnt * sizeof(T), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pin_ev[k], c->stream));
    }
    return ISX_OK;
}

// call_snv_site's per-base test (snv_utilities.py:179) is
//     c >= null_model[total]  and  float(c) / total >= min_freq
// For every coverage below lut_n both parts are monotone in c, so they fold into one exact
// integer threshold: thr[total] = max(null_model[total], min{c : (double)c / (double)total >= min_freq}),
// found here with the SAME IEEE fp64 division the reference performs (no rounding shortcuts).
std::vector<uint16_t> build_thresholds(const std::vector<int32_t> &lut, int32_t fallback, double min_freq)
{
    std::vector<uint16_t> thr(lut.size(), 65535);
    for (size_t t = 1; t < lut.size(); t++) {
        const int64_t mb = lut[t] >= 0 ? lut[t] : fallback;
        int64_t c = (int64_t)std::floor(min_freq * (double)t) - 2;
        if (c < 0) c = 0;
        while (c <= (int64_t)t && !((double)c / (double)t >= min_freq)) c++;
        while (c > 0 && ((double)(c - 1) / (double)t >= min_freq)) c--;
        const int64_t v = std::max<int64_t>(mb, c);        // c == t + 1: no count can reach min_freq
        thr[t] = (uint16_t)std::min<int64_t>(v, 65535);
    }
    return thr;
}

template <class T>
static int regrow(T **p, size_t *cap, size_t hard_max, size_t elem_pad = 0)
{
    if (*cap >= hard_max) { isx_set_error("output table is at its hard bound and still too small"); return ISX_ERR_CAPACITY; }
    const size_t want = std::min(hard_max, std::max<size_t>(*cap * 4, 1024));
    if (*p) isx_dev_free(*p);
    *p = nullptr;
    HIP_TRY(isx_raw_dev_malloc(p, (want + elem_pad) * sizeof(T)));
    *cap = want;
    return ISX_OK;
}

// The resident observation stream of a batch: built from the caller's isx_obs while they are staged for
// the upload (pinned, double buffered), together with the per-1024-record min/max directory the window
// ranges come from.
//   short stream    2 bytes per record (delta:13 | base:3, groups of 512)      n_mm_bins == 1
//   compact stream  4 bytes per record (delta:16 | mm:8 | base:3, groups of 256) mm profiling on
//   wide stream     isx_obs as is (8 bytes)   only for an mm level >= 256 (legal for no n_mm_bins) or when forced
// A group must span less than the delta range.  Where the stream jumps further (an uncovered stretch, the
// next genome of a database) the group is closed early and padded, so device record i is no longer input
// record i: og_start / og_count map every device group to its run of input records (built only when a
// jump exists; the pair ids follow the same map).
struct ObsStream {
    isx_ctx *c;
    isx_batch *b;
    const isx_obs *obs;
    const uint32_t *pair;
    const int64_t n_obs, n_pos;
    uint64_t n_chunks = 0;                          // directory: per ISX_CHUNK device records
    std::vector<uint32_t> cmin, cmax;
    std::vector<uint8_t> cany;
    std::vector<uint64_t> og_start;                 // empty = identity (device group g starts at input record G g)
    std::vector<uint16_t> og_count;
    std::vector<uint32_t> gbase;
    uint64_t G = ISX_GROUP;                         // records per position base
    uint32_t SPAN = 65535u;                         // a group must span less than this
    bool want16 = false, bad_pos = false;
    std::atomic<int> too_wide{0}, has_jump{0};

    ObsStream(isx_ctx *c_, isx_batch *b_, const isx_obs *obs_, const uint32_t *pair_, int64_t n_obs_, int64_t n_pos_)
        : c(c_), b(b_), obs(obs_), pair(pair_), n_obs(n_obs_), n_pos(n_pos_)
    {
        static_assert(sizeof(isx_obs) == sizeof(uint2), "isx_obs must be the 8-byte device record");
        want16 = b->M == 1 && !(b->prm.layout & (ISX_LAYOUT_NO_SHORT_RECORDS | ISX_LAYOUT_WIDE_RECORDS));
        G = want16 ? ISX_GROUP16 : ISX_GROUP;
        SPAN = want16 ? 8191u : 65535u;
        too_wide.store((b->prm.layout & ISX_LAYOUT_WIDE_RECORDS) ? 1 : 0);   // forced wide stream (tests / A-B)
        reset_directory();
    }

    void reset_directory()
    {
        n_chunks = b->n_rec / ISX_CHUNK;
        cmin.assign(n_chunks, 0xFFFFFFFFu); cmax.assign(n_chunks, 0u); cany.assign(n_chunks, 0);
        bad_pos = false;
    }

    // a few host threads fill the pinned buffer (a single core copies at ~10 GB/s, PCIe Gen5 takes 63);
    // work(a0, a1, bad): element range of the piece, multiples of ISX_CHUNK
    template <class W>
    void fill_threads(uint64_t cnt, W &&work)
    {
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(16, cnt / (64 * ISX_CHUNK)));
        const uint64_t per_t = ((cnt + nt - 1) / nt + ISX_CHUNK - 1) / ISX_CHUNK * ISX_CHUNK;
        std::vector<std::thread> th;
        std::vector<int> bad(nt, 0);
        auto run = [&](unsigned t) {
            const uint64_t a0 = std::min<uint64_t>(cnt, (uint64_t)t * per_t), a1 = std::min<uint64_t>(cnt, a0 + per_t);
            work(a0, a1, bad[t]);
        };
        for (unsigned t = 1; t < nt; t++) th.emplace_back(run, t);
        run(0);
        for (auto &x : th) x.join();
        for (int v : bad) if (v) bad_pos = true;
    }

    uint64_t input_run(uint64_t dev_group, const isx_obs *&src) const      // records of a device group
    {
        if (og_start.empty()) {
            const uint64_t g0 = dev_group * G;
            src = obs + g0;
            return g0 < (uint64_t)n_obs ? std::min<uint64_t>(G, (uint64_t)n_obs - g0) : 0;
        }
        if (dev_group >= og_start.size()) { src = obs; return 0; }
        src = obs + og_start[dev_group];
        return og_count[dev_group];
    }

    // one group: directory + base + encoded records (T = uint32_t compact / uint16_t short)
    template <class T>
    bool encode_group(T *dst, uint64_t dev_first, int &bad)
    {
        const uint64_t dg = dev_first / G;
        const isx_obs *src;
        const uint64_t n_real = input_run(dg, src);
        uint32_t lo = 0xFFFFFFFFu, hi = 0, mmax = 0;
        for (uint64_t i = 0; i < n_real; i++) {
            const uint32_t g = src[i].gpos;
            lo = g < lo ? g : lo; hi = g > hi ? g : hi; mmax = src[i].mm > mmax ? src[i].mm : mmax;
        }
        if (n_real) {
            if (mmax >= 256u) { too_wide.store(1); return false; }
            if (hi - lo >= SPAN) { has_jump.store(1); return false; }              // identity layout only: the map has none
            if ((int64_t)hi >= n_pos) bad = 1;
            const uint64_t ch = dev_first / ISX_CHUNK;             // the groups of a chunk belong to one thread
            cmin[ch] = std::min(cmin[ch], lo); cmax[ch] = std::max(cmax[ch], hi); cany[ch] = 1;
            gbase[dg] = lo;
        }
        for (uint64_t i = 0; i < n_real; i++) {
            const uint32_t bc = src[i].base > 4 ? 4u : (uint32_t)src[i].base;
            if (sizeof(T) == 2) dst[i] = (T)((src[i].gpos - lo) | (bc << 13));
            else dst[i] = (T)((src[i].gpos - lo) | ((uint32_t)src[i].mm << 16) | (bc << 24));
        }
        for (uint64_t i = n_real; i < G; i++) dst[i] = sizeof(T) == 2 ? (T)0xFFFFu : (T)ISX_PAD32;
        return true;
    }

    template <class T>
    int upload_encoded(T **d_dst)
    {
        HIP_TRY(isx_raw_dev_malloc(d_dst, b->n_rec * sizeof(T) + ISX_TAIL_BYTES));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(reinterpret_cast<uint8_t *>(*d_dst) + b->n_rec * sizeof(T)),
                                  (int)(sizeof(T) == 2 ? 0xFFFFFFFFu : ISX_PAD32), ISX_TAIL_BYTES / 4, c->stream));
        return staged_upload(c, *d_dst, b->n_rec, [&](T *dst, uint64_t first, uint64_t cnt) {
            if (too_wide.load() || has_jump.load()) return;
            fill_threads(cnt, [&](uint64_t a0, uint64_t a1, int &bad) {
                for (uint64_t i0 = a0; i0 < a1; i0 += G)                   // `first`, a0, a1 are multiples of ISX_CHUNK
                    if (!encode_group(dst + i0, first + i0, bad)) return;
            });
        });
    }

    int upload_compact()                            // ISX_OK, or 1 = start over (jump found / too wide)
    {
        const uint64_t n_groups = b->n_rec / G;
        gbase.assign(n_groups, 0u);
        const int rc = want16 ? upload_encoded(&b->d_rec16) : upload_encoded(&b->d_rec32);
        if (rc != ISX_OK) return rc;
        if (too_wide.load() || has_jump.load()) {
            HIP_TRY(isx_wait_stream(c->stream));
            if (b->d_rec32) isx_dev_free(b->d_rec32);
            if (b->d_rec16) isx_dev_free(b->d_rec16);
            b->d_rec32 = nullptr; b->d_rec16 = nullptr;
            return 1;
        }
        HIP_TRY(isx_raw_dev_malloc(&b->d_gbase, (n_groups + ISX_TAIL_GROUPS) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(b->d_gbase + n_groups, 0, ISX_TAIL_GROUPS * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemcpy(b->d_gbase, gbase.data(), n_groups * sizeof(uint32_t), hipMemcpyHostToDevice));
        return ISX_OK;
    }

    // the stream jumps: cut the input into runs that fit a group (greedy, arrival order), per input group in
    // parallel, then lay the runs out one device group each
    int cut_at_jumps()
    {
        const uint64_t n_in = ((uint64_t)n_obs + G - 1) / G;
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(16, n_in / 4096 + 1));
        std::vector<std::vector<std::pair<uint64_t, uint16_t>>> parts(nt);
        std::vector<std::thread> th;
        auto cut = [&](unsigned t) {
            auto &out = parts[t];
            for (uint64_t g = n_in * t / nt; g < n_in * (t + 1) / nt; g++) {
                const uint64_t s0 = g * G, s1 = std::min<uint64_t>((uint64_t)n_obs, s0 + G);
                uint64_t run0 = s0;
                uint32_t lo = 0xFFFFFFFFu, hi = 0;
                for (uint64_t i = s0; i < s1; i++) {
                    const uint32_t p = obs[i].gpos;
                    const uint32_t nlo = p < lo ? p : lo, nhi = p > hi ? p : hi;
                    if (i > run0 && nhi - nlo >= SPAN) {
                        out.emplace_back(run0, (uint16_t)(i - run0));
                        run0 = i; lo = hi = p;
                    } else { lo = nlo; hi = nhi; }
                }
                if (s1 > run0) out.emplace_back(run0, (uint16_t)(s1 - run0));
            }
        };
        for (unsigned t = 1; t < nt; t++) th.emplace_back(cut, t);
        cut(0);
        for (auto &x : th) x.join();
        size_t total = 0;
        for (auto &v : parts) total += v.size();
        og_start.reserve(total); og_count.reserve(total);
        for (auto &v : parts) for (auto &r : v) { og_start.push_back(r.first); og_count.push_back(r.second); }
        const uint64_t want = (uint64_t)og_start.size() * G;
        b->n_rec = std::max<uint64_t>(ISX_PAD, (want + ISX_PAD - 1) / ISX_PAD * ISX_PAD);
        if (b->n_rec >= 0xFFFFFFFFull) { isx_set_error("more than 2^32 records in one batch"); return ISX_ERR_ARG; }
        return ISX_OK;
    }

    // wide stream: isx_obs already has the device record layout (gpos | mm, base << 16, flags << 24): plain
    // copy into the pinned buffer, then one vectorisable sweep per chunk for the min/max directory
    int upload_wide()
    {
        HIP_TRY(isx_raw_dev_malloc(&b->d_rec, b->n_rec * sizeof(uint2)));
        return staged_upload(c, b->d_rec, b->n_rec, [&](uint2 *dst, uint64_t first, uint64_t cnt) {
            fill_threads(cnt, [&](uint64_t a0, uint64_t a1, int &bad) {
                for (uint64_t i0 = a0; i0 < a1; i0 += ISX_CHUNK) {
                    const uint64_t i1 = std::min<uint64_t>(a1, i0 + ISX_CHUNK);
                    const uint64_t g0 = first + i0;
                    const uint64_t n_real = g0 < (uint64_t)n_obs ? std::min<uint64_t>(i1 - i0, (uint64_t)n_obs - g0) : 0;
                    if (n_real) memcpy(dst + i0, obs + g0, n_real * sizeof(uint2));
                    for (uint64_t i = i0 + n_real; i < i1; i++) dst[i] = make_uint2(ISX_SENTINEL, 0);
                    if (!n_real) continue;
                    uint32_t lo = 0xFFFFFFFFu, hi = 0;
                    for (uint64_t i = i0; i < i0 + n_real; i++) { const uint32_t g = dst[i].x; lo = g < lo ? g : lo; hi = g > hi ? g : hi; }
                    const uint64_t ch = g0 / ISX_CHUNK;
                    cmin[ch] = lo; cmax[ch] = hi; cany[ch] = 1;
                    if ((int64_t)hi >= n_pos) bad = 1;
                }
            });
        });
    }

    int upload_records()
    {
        int rc;
        if (!too_wide.load()) {
            rc = upload_compact();
            if (rc < 0) return rc;
            if (rc == 1 && has_jump.load() && !too_wide.load()) {
                if ((rc = cut_at_jumps()) != ISX_OK) return rc;
                reset_directory();
                has_jump.store(0);
                rc = upload_compact();
                if (rc < 0) return rc;
                if (rc == 1 && !too_wide.load()) { isx_set_error("internal: a cut run still spans too many positions"); return ISX_ERR_STATE; }
            }
            if (!b->d_rec32 && !b->d_rec16) {       // mm >= 256 somewhere: back to the plain layout
                og_start.clear(); og_count.clear();
                b->n_rec = std::max<uint64_t>(ISX_PAD, ((uint64_t)n_obs + ISX_PAD - 1) / ISX_PAD * ISX_PAD);
                reset_directory();
            }
        }
        if (!b->d_rec32 && !b->d_rec16 && (rc = upload_wide()) != ISX_OK) return rc;
        if (bad_pos) { isx_set_error("observation gpos >= n_pos"); return ISX_ERR_ARG; }
        return ISX_OK;
    }

    // linkage: pair ids in device record order; positions alone for the allele pass -- 2-byte deltas to the
    // group base with the compact stream (always possible), to the chunk's lowest position with the wide one
    // when every chunk spans < 65535 positions, else the 4-byte positions; the short stream is read itself
    int upload_linkage_arrays()
    {
        HIP_TRY(isx_raw_dev_malloc(&b->d_pair, b->n_rec * sizeof(uint32_t)));
        if (!b->d_rec16) {
            bool narrow = true;
            if (!b->d_rec32) for (uint64_t i = 0; i < n_chunks; i++) if (cany[i] && cmax[i] - cmin[i] >= 65535u) { narrow = false; break; }
            std::vector<uint32_t> cb;
            if (narrow) {
                HIP_TRY(isx_raw_dev_malloc(&b->d_gpos16, b->n_rec * sizeof(uint16_t)));
                if (b->d_rec32) { b->gpos16_shift = 5; }                 // base per ISX_GROUP = 32 loads of 8 records
                else {
                    cb.assign(cmin.begin(), cmin.end());
                    for (uint64_t i = 0; i < n_chunks; i++) if (!cany[i]) cb[i] = 0;
                    HIP_TRY(isx_raw_dev_malloc(&b->d_cbase, std::max<uint64_t>(n_chunks, 1) * sizeof(uint32_t)));
                    HIP_TRY(hipMemcpyAsync(b->d_cbase, cb.data(), n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                    b->gpos16_shift = 7;                                 // base per ISX_CHUNK = 128 loads
                }
            } else {
                HIP_TRY(isx_raw_dev_malloc(&b->d_gpos, b->n_rec * sizeof(uint32_t)));
            }
            launch_extract_gpos(b->d_rec, b->d_rec32, b->d_gbase, b->d_gpos, b->d_gpos16, b->d_rec32 ? b->d_gbase : b->d_cbase,
                                b->d_rec32 ? ISX_GROUP : ISX_CHUNK, b->n_rec, c->stream);
            HIP_TRY(isx_wait_stream(c->stream));                     // cb is a local
        }
        uint32_t maxp = 0;
        const int rc = staged_upload(c, b->d_pair, b->n_rec, [&](uint32_t *dst, uint64_t first, uint64_t cnt) {
            if (og_start.empty()) {
                for (uint64_t i = 0; i < cnt; i++) {
                    const uint64_t g = first + i;
                    dst[i] = g < (uint64_t)n_obs ? pair[g] : 0u;
                    if (g < (uint64_t)n_obs) maxp = std::max(maxp, pair[g]);
                }
            } else {                                                      // device group -> run of input records
                for (uint64_t i0 = 0; i0 < cnt; i0 += G) {
                    const uint64_t dg = (first + i0) / G;
                    const uint64_t n_real = dg < og_start.size() ? og_count[dg] : 0;
                    for (uint64_t i = 0; i < n_real; i++) { dst[i0 + i] = pair[og_start[dg] + i]; maxp = std::max(maxp, dst[i0 + i]); }
                    for (uint64_t i = n_real; i < G; i++) dst[i0 + i] = 0u;
                }
            }
        });
        b->n_pairs = (uint64_t)maxp + 1;
        return rc;
    }
};


// window size: explicit, or (dense) the multiple of 64 that wastes least on MI355X for batches that fill the
// chip (tools/tune_pileup.py), smaller for small batches so every CU still owns >= 2 windows; (mm) the
// largest multiple of 64 whose counters fit half of the 160 KiB LDS (two resident workgroups per CU),
// at most 2 positions per lane
// mm kernel: two 512-lane workgroups per CU (78 KB of LDS each; their phases can overlap) while that leaves a window of at
// least 1024 positions; with more mm bins one 1024-lane workgroup with the whole CU's LDS -- a window twice as wide halves the
// over-scan and the per-window fixed costs (16 bins: W 384 -> 1024, 0.390 -> 0.343 ms on C2; 32 bins: 0.757 -> 0.622 ms)
void batch_pick_block(isx_batch *b)
{
    if (b->M == 1) { b->block = 1024; return; }
    b->block = 512;
    if (b->prm.window > 0) { if (b->prm.window > 1024) b->block = 1024; return; }
    if (batch_window_for(b, b->n_pos, true) < 1024) b->block = 1024;
}

int batch_window_for(const isx_batch *b, int64_t n_pos, bool packed)
{
    const isx_params *prm = &b->prm;
    if (prm->window > 0) return prm->window;
    if (b->M == 1) {
        // small batches: every CU should still own >= 2 windows
        const int64_t wsmall = (n_pos / 1024 + 63) / 64 * 64;
        if (wsmall < 2048) return (int)std::max<int64_t>(512, wsmall);
        // otherwise the multiple of 64 in [2048, 3264] (two 1024-lane workgroups per CU fit their LDS)
        // that wastes least: whole rounds of the 512 persistent workgroups x stream over-scan of a
        // window (~ one read length + one directory chunk on each side)
        // (with linkage a position also carries slabc + maskl: 25 bytes, so two workgroups fit up to 3136)
        int best = 2560;
        double best_eff = 0.0;
        // (reference-delta records carry two more rows -- skipped columns aside, the coverage differences: 24 / 29 bytes a position)
        const int wtop = (b->drec && packed) ? (prm->enable_linkage ? 4352 : 6080)        // 16-bit counters: 12 (+ 6 with linkage) bytes a position
                       : b->drec ? (prm->enable_linkage ? 2688 : 3264)
                                 : (prm->enable_linkage ? 3136 : ISX_PK16_MAX_W);      // 3264: the packed decode of 2-byte records needs 16-bit byte offsets
        for (int w = 2048; w <= wtop; w += 64) {
            const double n_win = std::ceil((double)n_pos / w);
            const double rounds = n_win / (2.0 * b->ctx->pass_cus);
            const double eff = rounds / std::ceil(rounds) * (w / (w + 200.0));
            if (eff > best_eff + 1e-9) { best_eff = eff; best = w; }
        }
        return best;
    }
    // (reference-delta records: per level also the skipped-columns / coverage-difference row(s), 4 / 8 bytes)
    const int bytes_per_pos = b->M * ((packed ? 8 : 16) + (b->drec ? (packed ? 4 : 8) : 0)) + ((b->M + 31) / 32) * 4 + 5;
    const int budget = (b->block >= 1024 ? 156 : 78) * 1024;       // one 1024-lane workgroup per CU, or two of 512
    const int wmax = ((budget - 8 * b->qcap - 8192 - 2048 - 256) / bytes_per_pos) / 64 * 64;
    return std::min(std::max(wmax, 64), 2 * b->block);
}

uint64_t build_window_directory(const uint32_t *cmin, const uint32_t *cmax, const uint8_t *cany, uint64_t n_chunks, int W,
                                int64_t n_pos, std::vector<uint2> &win, uint32_t chunk)
{
    std::vector<uint32_t> pmax(n_chunks), smin(n_chunks);
    uint32_t run = 0;
    for (uint64_t i = 0; i < n_chunks; i++) { if (cany[i]) run = std::max(run, cmax[i]); pmax[i] = run; }
    run = 0xFFFFFFFFu;
    for (uint64_t i = n_chunks; i-- > 0;) { if (cany[i]) run = std::min(run, cmin[i]); smin[i] = run; }
    const int n_win = (int)((n_pos + W - 1) / W);
    win.assign((size_t)n_win, make_uint2(0, 0));
    uint64_t lo = 0, hi = 0, longest = 0;
    for (int w = 0; w < n_win; w++) {
        const uint64_t w0 = (uint64_t)w * W, w1 = w0 + W;
        while (lo < n_chunks && (uint64_t)pmax[lo] < w0) lo++;       // chunks before lo: every gpos < w0
        if (hi < lo) hi = lo;
        while (hi < n_chunks && (uint64_t)smin[hi] < w1) hi++;       // chunks from hi on: every gpos >= w1
        win[(size_t)w] = make_uint2((uint32_t)(lo * chunk), (uint32_t)(hi * chunk));
        longest = std::max(longest, (hi - lo) * chunk);
    }
    return longest;
}

// The same directory on a pool's threads (a pipe's submit runs it between a batch's record pass and the batch's copy-in: 0.25 ms on the
// calling thread while the stager's threads wait, once per batch).  Two steps: running max / min per block of chunks with the blocks'
// carries applied on access, then every thread its run of windows (binary search for the first window, two pointers after).
uint64_t build_window_directory_mt(isxenc::HostPool &pool, const uint32_t *cmin, const uint32_t *cmax, const uint8_t *cany, uint64_t n_chunks, int W,
                                   int64_t n_pos, std::vector<uint2> &win, uint32_t chunk, std::vector<uint32_t> &pmax, std::vector<uint32_t> &smin)
{
    const int n_win = (int)((n_pos + W - 1) / W);
    const int T = std::min(pool.size(), 16);
    if (T < 2 || n_chunks < 4096 || n_win < 1024) return build_window_directory(cmin, cmax, cany, n_chunks, W, n_pos, win, chunk);
    if (pmax.size() < n_chunks) { pmax.resize(n_chunks); smin.resize(n_chunks); }
    const uint64_t B = (n_chunks + (uint64_t)T - 1) / (uint64_t)T;         // chunks a block
    uint32_t bmax[16], bmin[16], cmax_in[16], cmin_in[16];
    pool.run(T, [&](int t) {
        const uint64_t a = std::min(n_chunks, B * (uint64_t)t), e = std::min(n_chunks, a + B);
        uint32_t run = 0;
        for (uint64_t i = a; i < e; i++) { if (cany[i]) run = std::max(run, cmax[i]); pmax[i] = run; }
        bmax[t] = run;
        run = 0xFFFFFFFFu;
        for (uint64_t i = e; i-- > a;) { if (cany[i]) run = std::min(run, cmin[i]); smin[i] = run; }
        bmin[t] = run;
    });
    { uint32_t run = 0; for (int t = 0; t < T; t++) { cmax_in[t] = run; run = std::max(run, bmax[t]); } }          // what lies before / behind a block
    { uint32_t run = 0xFFFFFFFFu; for (int t = T; t-- > 0;) { cmin_in[t] = run; run = std::min(run, bmin[t]); } }
    auto PM = [&](uint64_t i) { return std::max(pmax[i], cmax_in[i / B]); };
    auto SM = [&](uint64_t i) { return std::min(smin[i], cmin_in[i / B]); };
    win.resize((size_t)n_win);
    uint64_t longest_t[16];
    pool.run(T, [&](int t) {
        const int wa = (int)((int64_t)n_win * t / T), we = (int)((int64_t)n_win * (t + 1) / T);
        uint64_t longest = 0, lo = 0, hi = 0;
        if (wa < we) {
            const uint64_t w0 = (uint64_t)wa * (uint64_t)W;
            uint64_t l = 0, r = n_chunks;                    // first chunk with PM >= w0
            while (l < r) { const uint64_t m = (l + r) >> 1; if ((uint64_t)PM(m) < w0) l = m + 1; else r = m; }
            lo = hi = l;
        }
        for (int w = wa; w < we; w++) {
            const uint64_t w0 = (uint64_t)w * (uint64_t)W, w1 = w0 + (uint64_t)W;
            while (lo < n_chunks && (uint64_t)PM(lo) < w0) lo++;
            if (hi < lo) hi = lo;
            while (hi < n_chunks && (uint64_t)SM(hi) < w1) hi++;
            win[(size_t)w] = make_uint2((uint32_t)(lo * chunk), (uint32_t)(hi * chunk));
            longest = std::max(longest, (hi - lo) * chunk);
        }
        longest_t[t] = longest;
    });
    uint64_t longest = 0;
    for (int t = 0; t < T; t++) longest = std::max(longest, longest_t[t]);
    if (getenv("ISX_DIR_CHECK")) {          // tests: the threads' directory against the plain one
        std::vector<uint2> ref;
        const uint64_t l2 = build_window_directory(cmin, cmax, cany, n_chunks, W, n_pos, ref, chunk);
        if (l2 != longest || ref.size() != win.size() || memcmp(ref.data(), win.data(), ref.size() * sizeof(uint2)) != 0) {
            fprintf(stderr, "build_window_directory_mt differs from build_window_directory (%llu chunks, %d windows)\n", (unsigned long long)n_chunks, n_win);
            abort();
        }
    }
    return longest;
}

int batch_set_geometry(isx_batch *b)
{
    const bool dense = b->M == 1;
    if (!dense && b->W > 2 * b->block) { isx_set_error("mm path: window must be <= 2 x block"); return ISX_ERR_ARG; }
    if (b->drec && dense && b->W > (b->packed ? 8 : 4) * b->block) { isx_set_error("reference-delta records: window must be <= 4 x block (8 x with 16-bit counters)"); return ISX_ERR_ARG; }
    b->rqcap = dense ? 0 : std::min(b->W, 512);     // positions with SNV rows per window (overflow: per-position atomics)
    b->lds = pileup_lds_bytes(b->W, b->M, b->qcap, b->rqcap, b->prm.enable_linkage, b->packed, b->block, b->segs ? (b->drec ? 32 : 64) : 0, &b->stage_off, &b->dlt_off);
    if (b->lds > 160 * 1024) { isx_set_error("window * n_mm_bins does not fit the 160 KiB LDS"); return ISX_ERR_ARG; }
    {   // persistent kernels: as many workgroups as stay resident on the 256 CUs
        const int per_cu = std::max(1, std::min((int)(160 * 1024 / b->lds), 2048 / b->block));
        int g = b->ctx->pass_cus * per_cu;
#ifdef ISX_TUNING
        if (const char *e = getenv("ISX_GRID")) g = std::max(8, atoi(e));       // tuning builds only
#endif
        g = std::min(g, b->n_win);
        b->grid = std::max(8, (g + 7) / 8 * 8);
    }
    if (!dense) {
        // entry slabs: min(M, 4) levels per position fit without overflow; the rest spills
        b->slab = (uint32_t)b->W * (uint32_t)std::min(b->M, 4);
        if (!b->cap_ovf) {
            const uint64_t npm = (uint64_t)b->n_pos * b->M;
            b->cap_ovf = b->M <= 4 ? 16 : (size_t)std::max<uint64_t>(1u << 20, std::min<uint64_t>((uint64_t)b->n_obs, npm) / 4);
        }
    }
    return ISX_OK;
}


extern "C" {

const char *isx_last_error(void) { return g_err.c_str(); }
int isx_abi_version(void) { return ISX_ABI_VERSION; }

// The two pass queues of a context.  r = 0: plain queues at the highest stream priority -- a pileup kernel is a persistent grid sized for the
// whole device; when copy kernels and the linkage chains of other batches hold some of the wave slots, its late workgroups delay the whole
// pass.  r > 0: queues masked off r CUs of every XCD instead (hipExtStreamCreateWithCUMask: a masked queue carries no priority), and every
// side queue of the context (copy-in, finishers, copy-out: isx_side_stream_create) masked ONTO them.  A persistent pileup grid fills every CU
// it may use (LDS and wave slots) for its whole run; without the reserve each of the ~60 short launches of a finisher's chain (linkage sorts,
// gathers, copy kernels) waits for a pileup kernel to END.
static int make_pass_queues(isx_ctx *c, int r, int n_cu)
{
    r = std::max(0, std::min(r, 8));
    if (n_cu != 256) r = 0;
    for (int i = 0; i < 2; i++) if (c->pstream[i]) { HIP_TRY(isx_wait_stream(c->pstream[i])); HIP_TRY(hipStreamDestroy(c->pstream[i])); c->pstream[i] = nullptr; }
    for (int k = 0; k < 8; k++) c->side_mask[k] = 0;
    c->pass_cus = 256;
    if (r > 0) {
        // bit layouts differ between driver versions (the CUs of an XCD contiguous, or interleaved across the 8 XCDs): clear bits that are
        // r per XCD under BOTH readings -- in block k of 32 bits the bits whose index is k (beyond 4 of them: k + 1) modulo 8
        uint32_t mask[8];
        for (int k = 0; k < 8; k++) {
            mask[k] = 0xFFFFFFFFu;
            for (int j = 0; j < r; j++) mask[k] &= ~(1u << (((k + (j >> 2)) & 7) + 8 * (j & 3)));
        }
        bool ok = true;
        for (int i = 0; i < 2 && ok; i++) ok = hipExtStreamCreateWithCUMask(&c->pstream[i], 8, mask) == hipSuccess;
        if (ok) {
            c->pass_cus = 256 - 8 * r;
            if (!getenv("ISX_SIDE_UNMASKED")) for (int k = 0; k < 8; k++) c->side_mask[k] = ~mask[k];
        } else {            // a stack that refuses masked queues: go on without a reserve (the request is a performance hint, not a contract)
            (void)hipGetLastError();
            for (int i = 0; i < 2; i++) if (c->pstream[i]) { (void)hipStreamDestroy(c->pstream[i]); c->pstream[i] = nullptr; }
            r = 0;
        }
    }
    if (r == 0) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
        const int prio = getenv("ISX_NO_STREAM_PRIORITY") ? least : greatest;
        for (int i = 0; i < 2; i++) HIP_TRY(hipStreamCreateWithPriority(&c->pstream[i], hipStreamNonBlocking, prio));
    }
    return ISX_OK;
}

int isx_ctx_create(int device_id, isx_ctx **out)
{
    if (!out) { isx_set_error("isx_ctx_create: out is NULL"); return ISX_ERR_ARG; }
    *out = nullptr;
    // (a pipe drives a dozen streams at once; the runtime's default of 4 hardware queues serialises them -- see instrain_amd/__init__.py.
    //  Honoured only when this is the process' first HIP call and the user has set nothing.)
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        isx_set_error("no HIP device visible: libinstrain_amd has no CPU fallback");
        return ISX_ERR_HIP;
    }
    if (device_id < 0 || device_id >= n) { isx_set_error("bad device id"); return ISX_ERR_ARG; }
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        isx_set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return ISX_ERR_HIP;
    }
    isx_ctx *c = new isx_ctx();
    c->device = device_id;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        const char *e = getenv("ISX_PASS_CU_RESERVE");       // default of every context of this process (isx_ctx_reserve_cus sets one context's)
        const int rc = make_pass_queues(c, e ? atoi(e) : 0, prop.multiProcessorCount);
        if (rc != ISX_OK) return rc;
    }
    c->pin_bytes = (size_t)64 << 20;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipHostMalloc(&c->pin[i], c->pin_bytes, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    }
    *out = c;
    return ISX_OK;
}

extern "C++" hipError_t isx_side_stream_create(isx_ctx *c, hipStream_t *s)
{
    bool any = false;
    for (int k = 0; k < 8; k++) any = any || c->side_mask[k] != 0;
    if (any) {
        if (hipExtStreamCreateWithCUMask(s, 8, c->side_mask) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();            // (as above: without the mask rather than not at all)
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

int isx_ctx_reserve_cus(isx_ctx *c, int cus_per_xcd)
{
    if (!c) { isx_set_error("isx_ctx_reserve_cus: null context"); return ISX_ERR_ARG; }
    if (c->n_created || c->n_pipes) { isx_set_error("isx_ctx_reserve_cus: call it before the context's first batch or pipe (their launch shapes and queues follow it)"); return ISX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    return make_pass_queues(c, cus_per_xcd, prop.multiProcessorCount);
}

void isx_ctx_destroy(isx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_lut) isx_dev_free(c->d_lut);
    for (int i = 0; i < 2; i++) {
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int i = 0; i < 2; i++) if (c->pstream[i]) (void)hipStreamDestroy(c->pstream[i]);
    isx_dev_trim();
    delete c;
}

int isx_set_null_model(isx_ctx *c, const int32_t *lut, int64_t n, int32_t fallback)
{
    if (!c || !lut || n <= 0) { isx_set_error("isx_set_null_model: bad argument"); return ISX_ERR_ARG; }
    if (n > 65535) { isx_set_error("null model longer than 65535 coverages"); return ISX_ERR_ARG; }
    if (fallback < 0 || fallback >= 255) { isx_set_error("null model fallback out of range"); return ISX_ERR_ARG; }
    std::vector<uint8_t> h((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (lut[i] >= 255) { isx_set_error("null model value >= 255"); return ISX_ERR_ARG; }
        h[(size_t)i] = lut[i] < 0 ? 255 : (uint8_t)lut[i];
    }
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_lut) { isx_dev_free(c->d_lut); c->d_lut = nullptr; }
    HIP_TRY(isx_raw_dev_malloc(&c->d_lut, (size_t)n));
    HIP_TRY(hipMemcpy(c->d_lut, h.data(), (size_t)n, hipMemcpyHostToDevice));
    c->lut_n = (int32_t)n;
    c->fallback = fallback;
    c->h_lut.assign(lut, lut + n);
    return ISX_OK;
}

void isx_batch_destroy(isx_batch *b)
{
    if (!b) return;
    for (int i = 0; i < 2; i++) {
        if (b->ctx->unpublished[i] == b) b->ctx->unpublished[i] = nullptr;
        if (b->ctx->pstream[i]) (void)isx_wait_stream(b->ctx->pstream[i]);
    }
    (void)hipSetDevice(b->ctx->device);
    (void)isx_wait_stream(b->ctx->stream);
    void *ps[] = {b->d_cov_row_win, b->d_cov8, b->d_sat, b->d_clon_list, b->d_clon_sorted, b->d_seg, b->d_drec, b->d_rec, b->d_rec32, b->d_rec16, b->d_gbase, b->d_pair, b->d_gpos, b->d_gpos16, b->d_cbase, b->d_ref, b->d_win, b->d_thr, b->d_bounds, b->d_counts, b->d_clon, b->d_clon_r, b->d_cov16, b->d_rare, b->d_entries, b->d_win_nent, b->d_win_site_base, b->d_win_site_cnt, b->d_lev_mask, b->d_lev_cov, b->d_lev_win_off, b->d_slev,
                  b->d_snv, b->d_sites, b->d_ao, b->d_cursors, b->d_snv_raw, b->d_sites_raw, b->d_rare_raw, b->d_win_rec, b->d_win_out};
    if (b->h_state) (void)hipHostFree(b->h_state);
    for (void *p : ps) if (p) isx_dev_free(p);          // (falls through to hipFree for blocks that did not come from the cache)
    b->L.release();
    b->S.release();
    b->C.release();
    for (auto &e : b->ev_sum) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    delete b;
}

// isx_batch_create (observations) and isx_batch_create_reads (read segments: segs != NULL, n_obs = an upper bound of
// the observations they stand for) share everything but the upload of the stream
static int batch_create_impl(isx_ctx *c, const isx_params *prm, int64_t n_pos, const uint8_t *ref, int32_t n_splits,
                             const int64_t *split_bounds, int64_t n_obs, const isx_obs *obs, const uint32_t *pair,
                             const isx_segs *segs, isx_batch **out)
{
    *out = nullptr;
    if (!c->d_lut) { isx_set_error("isx_batch_create: call isx_set_null_model first"); return ISX_ERR_STATE; }
    if (prm->enable_linkage && n_obs && !(segs ? segs->pair : pair)) { isx_set_error("linkage needs the pair array"); return ISX_ERR_ARG; }
    if (prm->n_mm_bins < 1 || prm->n_mm_bins > 128) { isx_set_error("n_mm_bins must be in [1, 128]"); return ISX_ERR_ARG; }
    if (prm->enable_linkage && prm->linkage_mode == 2 && prm->n_mm_bins != 1) { isx_set_error("the dense MFMA linkage path needs n_mm_bins == 1"); return ISX_ERR_ARG; }
    if (n_pos >= (int64_t)0xFFFF0000ll) { isx_set_error("flat position space must be < 2^32 - 65536"); return ISX_ERR_ARG; }
    if (split_bounds[0] != 0 || split_bounds[n_splits] != n_pos) { isx_set_error("split_bounds must span [0, n_pos]"); return ISX_ERR_ARG; }
    for (int i = 0; i < n_splits; i++)
        if (split_bounds[i + 1] <= split_bounds[i]) { isx_set_error("split_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    HIP_TRY(hipSetDevice(c->device));

    isx_batch *b = new isx_batch();
    b->ps = (int)(c->n_created++ & 1u);
    b->ctx = c; b->prm = *prm; b->n_pos = n_pos; b->n_obs = n_obs; b->n_splits = n_splits;
    b->M = prm->n_mm_bins;
    b->segs = segs != nullptr;
    // (round 6: reference-delta records can carry the pair's mm level in the header -- opt-in with mm profiling on, see ISX_LAYOUT_MM_DELTA_RECORDS)
    b->drec = b->segs && !(prm->layout & ISX_LAYOUT_SEG64_RECORDS) && (b->M == 1 || (prm->layout & ISX_LAYOUT_MM_DELTA_RECORDS));
    const bool dense = b->M == 1;
    batch_pick_block(b);
#ifdef ISX_TUNING
    if (const char *e = getenv("ISX_BLOCK")) b->block = atoi(e);       // tuning builds only (make tuning)
#endif
    if (b->block < 64 || b->block > 1024 || (b->block & 63)) { delete b; isx_set_error("ISX_BLOCK must be a multiple of 64 in [64, 1024]"); return ISX_ERR_ARG; }
    if (prm->window && (prm->window < 64 || (prm->window & 63) || prm->window > 8192)) { delete b; isx_set_error("window must be a multiple of 64 in [64, 8192]"); return ISX_ERR_ARG; }
    b->n_rec = std::max<uint64_t>(ISX_PAD, ((uint64_t)n_obs + ISX_PAD - 1) / ISX_PAD * ISX_PAD);
    if (b->n_rec >= 0xFFFFFFFFull) { delete b; isx_set_error("more than 2^32 observations in one batch"); return ISX_ERR_ARG; }

#define BT(expr) do { int _rc = (expr); if (_rc != ISX_OK) { isx_batch_destroy(b); return _rc; } } while (0)
#define BH(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { isx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); isx_batch_destroy(b); return ISX_ERR_HIP; } } while (0)
    for (auto &e : b->ev) BH(hipEventCreate(&e));
    for (auto &e : b->ev_sum) BH(hipEventCreate(&e));
    BH(isx_raw_dev_malloc(&b->d_ref, (size_t)n_pos));
    BH(isx_raw_dev_malloc(&b->d_bounds, (size_t)(n_splits + 1) * sizeof(int64_t)));
    BH(isx_raw_dev_malloc(&b->d_cursors, (CUR_N + 4) * sizeof(uint32_t)));
    b->d_flags = b->d_cursors + CUR_N;
    BH(hipHostMalloc(&b->h_state, (CUR_N + 8) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(b->h_state, 0, (CUR_N + 8) * sizeof(uint32_t));
    BH(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_host_state), b->h_state, 0));
    BH(hipMemsetAsync(b->d_cursors, 0, (CUR_N + 4) * sizeof(uint32_t), c->stream));
    {   // folded presence threshold per coverage (see build_thresholds)
        std::vector<uint16_t> thr = build_thresholds(c->h_lut, c->fallback, prm->min_freq);
        BH(isx_raw_dev_malloc(&b->d_thr, thr.size() * sizeof(uint16_t)));
        BH(hipMemcpy(b->d_thr, thr.data(), thr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    const uint64_t npm = (uint64_t)n_pos * b->M;
    if (b->M == 1) {
        BH(isx_raw_dev_malloc(&b->d_counts, (size_t)n_pos * sizeof(uint4)));
        BH(isx_raw_dev_malloc(&b->d_clon, (size_t)n_pos * sizeof(float)));
    } else {
        // entries / site-level tables are sized once the window geometry is known (below)
    }
    if (b->M == 1) {   // rarefied clonality: NaN where not produced (the positions are the same every run)
        BH(isx_raw_dev_malloc(&b->d_clon_r, (size_t)n_pos * sizeof(float)));
        BH(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->d_clon_r), 0x7FC00000, (size_t)n_pos, c->stream));
    }
    b->cap_snv = (size_t)std::min<uint64_t>(npm, std::max<uint64_t>((uint64_t)n_pos / 2, 1u << 20));
    b->cap_sites = (size_t)std::min<uint64_t>((uint64_t)n_pos, std::max<uint64_t>((uint64_t)n_pos / 4, 1u << 20));
    b->cap_ao = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_obs, std::max<uint64_t>((uint64_t)n_obs / 4, 1u << 20)));
    BH(isx_raw_dev_malloc(&b->d_snv, b->cap_snv * sizeof(isx_snv)));
    BH(isx_raw_dev_malloc(&b->d_sites, b->cap_sites * sizeof(isx_site)));
    if (prm->enable_linkage) BH(isx_raw_dev_malloc(&b->d_ao, b->cap_ao * sizeof(isx_ao)));

    // ---- observation stream (+ pair ids / allele-pass positions with linkage): see ObsStream; or the read segments ----
    ObsStream st(c, b, obs, pair, segs ? 0 : n_obs, n_pos);
    uint32_t dir_chunk = ISX_CHUNK;
    if (segs && b->drec) {
        // read segments as reference-delta records: compared with the reference and encoded on the host (the pipe does the same into
        // pinned staging, seg_encode.cpp), one upload
        dir_chunk = ISX_DREC_GROUP;
        isxenc::HostPool pool((int)std::max<int64_t>(1, std::min<int64_t>(16, segs->n_seg / 65536 + 1)), -1, false);
        std::vector<uint32_t> h_rec, h_gbase, h_pair;
        isxenc::SegJob J;
        int64_t slack = 1;
        std::vector<int64_t> exact;             // second attempt: every task's region is what the first one found it needs
        for (int attempt = 0;; attempt++) {     // a second time when segments differ from the reference so often that their pieces outgrow a task's spare groups
            int64_t cap_rec = isxenc::delta_groups_needed(pool, segs->gpos, segs->n_seg, slack) * ISX_DREC_GROUP;
            if (!exact.empty()) { cap_rec = 0; for (int64_t v : exact) cap_rec += std::max<int64_t>(v, 1) * ISX_DREC_GROUP; }
            h_rec.resize((size_t)cap_rec * ISX_DREC_WORDS); h_gbase.resize((size_t)(cap_rec / ISX_DREC_GROUP));
            st.cmin.assign(h_gbase.size(), 0xFFFFFFFFu); st.cmax.assign(h_gbase.size(), 0u); st.cany.assign(h_gbase.size(), 0);
            J = isxenc::SegJob();
            J.in = *segs; J.n_seg = segs->n_seg; J.n_pos = n_pos; J.n_mm_bins = b->M; J.ref = ref; J.slack_groups = slack;
            J.task_groups = exact.empty() ? nullptr : exact.data();
            if (!prm->enable_linkage) J.in.pair = nullptr;
            J.rec = h_rec.data(); J.gbase = h_gbase.data(); J.pair_out = nullptr;
            J.cmin = st.cmin.data(); J.cmax = st.cmax.data(); J.cany = st.cany.data(); J.cap_rec = cap_rec;
            const int erc = isxenc::encode_delta(pool, J);
            if (erc == isxenc::SEG_CAPACITY && J.need_slack > slack && attempt == 0) { exact = J.task_need; continue; }
            if (erc != isxenc::SEG_OK) {
                isx_batch_destroy(b);
                if (erc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
                isx_set_error(erc == isxenc::SEG_BAD_POS ? "a segment reaches beyond n_pos" : erc == isxenc::SEG_BAD_LEN ? "a segment's length is not in [1, 150]"
                                                                                           : "internal: segment stream larger than estimated");
                return erc == isxenc::SEG_CAPACITY ? ISX_ERR_STATE : ISX_ERR_ARG;
            }
            break;
        }
        b->n_rec = (uint64_t)J.n_rec;
        b->n_pairs = (uint64_t)J.max_pair + 1;
        st.n_chunks = b->n_rec / ISX_DREC_GROUP;
        BH(isx_raw_dev_malloc(&b->d_drec, (size_t)b->n_rec * 32 + ISX_TAIL_BYTES));
        BH(hipMemcpyAsync(b->d_drec, h_rec.data(), (size_t)b->n_rec * 32, hipMemcpyHostToDevice, c->stream));
        BH(hipMemsetAsync(reinterpret_cast<uint8_t *>(b->d_drec) + (size_t)b->n_rec * 32, 0, ISX_TAIL_BYTES, c->stream));     // len 0: nothing to count
        BH(isx_raw_dev_malloc(&b->d_gbase, (st.n_chunks + ISX_TAIL_GROUPS) * sizeof(uint32_t)));
        BH(hipMemsetAsync(b->d_gbase + st.n_chunks, 0, ISX_TAIL_GROUPS * sizeof(uint32_t), c->stream));
        BH(hipMemcpyAsync(b->d_gbase, h_gbase.data(), st.n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        // (no pair table: the records carry the read-pair ids)
        BH(isx_wait_stream(c->stream));         // the host vectors are locals
    } else if (segs) {
        // read segments: encoded on the host (the pipe does the same into pinned staging, seg_encode.cpp), one upload
        dir_chunk = ISX_SEG_GROUP;
        isxenc::HostPool pool((int)std::max<int64_t>(1, std::min<int64_t>(16, segs->n_seg / 65536 + 1)), -1, false);
        // exactly the groups the encoder will cut (the same counting pass: a group closes after 16 segments or where its starts
        // would span more than 65 535 positions -- every few segments on a sparse stream, not only at jumps)
        const int64_t cap_rec = isxenc::seg_groups_needed(pool, segs->gpos, segs->n_seg) * ISX_SEG_GROUP;
        std::vector<uint32_t> h_rec((size_t)cap_rec * ISX_SEG_REC_WORDS), h_gbase((size_t)(cap_rec / ISX_SEG_GROUP)), h_pair;
        if (prm->enable_linkage) h_pair.resize((size_t)cap_rec);
        st.cmin.assign(h_gbase.size(), 0xFFFFFFFFu); st.cmax.assign(h_gbase.size(), 0u); st.cany.assign(h_gbase.size(), 0);
        isxenc::SegJob J;
        J.in = *segs; J.n_seg = segs->n_seg; J.n_pos = n_pos; J.n_mm_bins = b->M;
        if (!prm->enable_linkage) J.in.pair = nullptr;
        J.rec = h_rec.data(); J.gbase = h_gbase.data(); J.pair_out = prm->enable_linkage ? h_pair.data() : nullptr;
        J.cmin = st.cmin.data(); J.cmax = st.cmax.data(); J.cany = st.cany.data(); J.cap_rec = cap_rec;
        const int erc = isxenc::encode_segs(pool, J);
        if (erc != isxenc::SEG_OK) {
            isx_batch_destroy(b);
            if (erc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
            isx_set_error(erc == isxenc::SEG_BAD_POS ? "a segment reaches beyond n_pos" : erc == isxenc::SEG_BAD_LEN ? "a segment's length is not in [1, 150]"
                                                                                       : "internal: segment stream larger than estimated");
            return erc == isxenc::SEG_CAPACITY ? ISX_ERR_STATE : ISX_ERR_ARG;
        }
        b->n_rec = (uint64_t)J.n_rec;
        b->n_pairs = (uint64_t)J.max_pair + 1;
        st.n_chunks = b->n_rec / ISX_SEG_GROUP;
        BH(isx_raw_dev_malloc(&b->d_seg, (size_t)b->n_rec * 64 + ISX_TAIL_BYTES));
        BH(hipMemcpyAsync(b->d_seg, h_rec.data(), (size_t)b->n_rec * 64, hipMemcpyHostToDevice, c->stream));
        BH(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(reinterpret_cast<uint8_t *>(b->d_seg) + (size_t)b->n_rec * 64), (int)ISX_SEG_SKIP_WORD, ISX_TAIL_BYTES / 4, c->stream));
        BH(isx_raw_dev_malloc(&b->d_gbase, (st.n_chunks + ISX_TAIL_GROUPS) * sizeof(uint32_t)));
        BH(hipMemsetAsync(b->d_gbase + st.n_chunks, 0, ISX_TAIL_GROUPS * sizeof(uint32_t), c->stream));
        BH(hipMemcpyAsync(b->d_gbase, h_gbase.data(), st.n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (prm->enable_linkage) {
            BH(isx_raw_dev_malloc(&b->d_pair, (size_t)b->n_rec * sizeof(uint32_t)));
            BH(hipMemcpyAsync(b->d_pair, h_pair.data(), (size_t)b->n_rec * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        }
        BH(isx_wait_stream(c->stream));         // the host vectors are locals
    } else {
        int rc = st.upload_records();
        if (rc == ISX_OK && prm->enable_linkage) rc = st.upload_linkage_arrays();
        if (rc != ISX_OK) { isx_batch_destroy(b); return rc; }
    }
    const uint64_t n_chunks = st.n_chunks;
    const std::vector<uint32_t> &cmin = st.cmin, &cmax = st.cmax;
    const std::vector<uint8_t> &cany = st.cany;
    BT(staged_upload(c, b->d_ref, (uint64_t)n_pos, [&](uint8_t *dst, uint64_t first, uint64_t cnt) {
        memcpy(dst, ref + first, cnt);
    }));

    // ---- window -> record range: prefix-max / suffix-min over the chunk directory ----
    {
        std::vector<uint2> win;
        b->packed = 0;
        int W = batch_window_for(b, n_pos, false);
        if (!dense || b->drec) {
            // u16-packed counters are legal when no window streams >= 65536 records (reference-delta records: 32768 -- a
            // coverage difference is signed)
            const int Wp = batch_window_for(b, n_pos, true);
            if (!(prm->layout & ISX_LAYOUT_NO_PACKED_COUNTERS) &&
                build_window_directory(cmin.data(), cmax.data(), cany.data(), n_chunks, Wp, n_pos, win, dir_chunk) < (b->drec ? 32768u : 65536u)) { b->packed = 1; W = Wp; }
        }
        if (!b->packed) build_window_directory(cmin.data(), cmax.data(), cany.data(), n_chunks, W, n_pos, win, dir_chunk);
        b->W = W;
        b->n_win = (int)win.size();
        BT(batch_set_geometry(b));
        if (!dense) {
            const uint64_t tot = (uint64_t)b->n_win * b->slab + b->cap_ovf;
            if (tot >= 0xFFFFFFFFull) { isx_batch_destroy(b); isx_set_error("mm path: more than 2^32 entry slots in one batch"); return ISX_ERR_ARG; }
            b->cap_entries = (size_t)tot;
            BH(isx_raw_dev_malloc(&b->d_entries, b->cap_entries * sizeof(isx_entry)));
            BH(isx_raw_dev_malloc(&b->d_win_nent, (size_t)b->n_win * sizeof(uint32_t)));
            b->cap_slev = b->cap_sites * (size_t)std::min(b->M, 8);
            BH(isx_raw_dev_malloc(&b->d_slev, b->cap_slev * sizeof(isx_slev)));
        }
        BH(isx_raw_dev_malloc(&b->d_win, win.size() * sizeof(uint2)));
        BH(hipMemcpyAsync(b->d_win, win.data(), win.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        BH(hipMemcpyAsync(b->d_bounds, split_bounds, (size_t)(n_splits + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        BH(isx_wait_stream(c->stream));
    }
#undef BT
#undef BH
    *out = b;
    return ISX_OK;
}

int isx_batch_create(isx_ctx *c, const isx_params *prm, int64_t n_pos, const uint8_t *ref, int32_t n_splits,
                     const int64_t *split_bounds, int64_t n_obs, const isx_obs *obs, const uint32_t *pair,
                     isx_batch **out)
{
    if (!c || !prm || !out || !ref || !split_bounds || n_pos <= 0 || n_splits <= 0 || n_obs < 0 || (n_obs && !obs)) {
        isx_set_error("isx_batch_create: bad argument");
        return ISX_ERR_ARG;
    }
    return batch_create_impl(c, prm, n_pos, ref, n_splits, split_bounds, n_obs, obs, pair, nullptr, out);
}

int isx_batch_create_reads(isx_ctx *c, const isx_params *prm, int64_t n_pos, const uint8_t *ref, int32_t n_splits,
                           const int64_t *split_bounds, const isx_segs *segs, isx_batch **out)
{
    if (!c || !prm || !out || !ref || !split_bounds || n_pos <= 0 || n_splits <= 0 || !segs || segs->n_seg < 0 ||
        (segs->n_seg && (!segs->gpos || !segs->len || !segs->bases))) {
        isx_set_error("isx_batch_create_reads: bad argument");
        return ISX_ERR_ARG;
    }
    if (prm->layout & (ISX_LAYOUT_WIDE_RECORDS | ISX_LAYOUT_NO_SHORT_RECORDS)) { isx_set_error("isx_batch_create_reads: the record layouts apply to observation batches only"); return ISX_ERR_ARG; }
    if (prm->linkage_mode == 2) { isx_set_error("isx_batch_create_reads: the dense MFMA linkage path takes observation batches only"); return ISX_ERR_ARG; }
    int64_t n_bases = 0;                            // upper bound of the observations: sizes the linkage tables
    for (int64_t i = 0; i < segs->n_seg; i++) n_bases += segs->len[i];
    return batch_create_impl(c, prm, n_pos, ref, n_splits, split_bounds, n_bases, nullptr, nullptr, segs, out);
}

static float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

// enqueue one pass (pileup kernel + state publication) on the context's stream; no host wait
int launch_pass(isx_batch *b)
{
    isx_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->pstream[b->ps];
    b->ran = false;
    // no per-run memset / copy: the cursors run on (slots are relative to `base`), and the last
    // one-wave kernel k_publish_state copies them to mapped pinned memory right behind the pileup kernel

    PileupArgs a{};
    a.seg = b->d_seg; a.drec = b->d_drec; a.dlt_off = b->dlt_off;
    a.rec = b->d_rec; a.rec32 = b->d_rec32; a.rec16 = b->d_rec16; a.gbase = b->d_gbase; a.win_range = b->d_win; a.ref = b->d_ref; a.ref_packed = b->ref_packed; a.ref_n = b->d_ref_n;
    a.pair = b->d_pair; a.pair_runs = b->d_pair_runs; a.run_index = b->d_run_index; a.n_runs = b->n_runs; a.gpos = b->d_gpos; a.gpos16 = b->d_gpos16; a.chunk_base = b->d_rec32 ? b->d_gbase : b->d_cbase; a.gpos16_shift = b->gpos16_shift; a.thr = b->d_thr; a.lut_n = c->lut_n; a.fallback = c->fallback; a.qcap = b->qcap; a.rqcap = b->rqcap; a.stage_off = b->stage_off;
    a.n_pos = (uint32_t)b->n_pos; a.W = b->W; a.logW = b->logW; a.M = b->M; a.n_win = b->n_win;
    a.min_cov = b->prm.min_cov; a.min_freq = b->prm.min_freq;
#ifdef ISX_TUNING
    if (const char *e = getenv("ISX_DEBUG_MODE")) a.debug_mode = atoi(e);     // ablation, tuning builds only: skips parts of the kernel
#endif
    a.counts = b->d_counts; a.clon = b->d_clon; a.clon_r = b->d_clon_r;
    a.min_cov_r = b->prm.rarefied_coverage;
    a.cov16 = b->d_cov16; a.rare = b->d_rare; a.cap_rare = (uint32_t)std::min<size_t>(b->cap_rare, 0xFFFFFFFFu);
    a.cov8 = b->sparse_out && b->cov8_out ? b->d_cov8 : nullptr;
    a.sat = b->d_sat; a.cap_sat = (uint32_t)std::min<size_t>(b->cap_sat, 0xFFFFFFFFu); a.sat_thr = a.cov8 ? 255u : 65535u;
    b->nib_pass = false;
    if (b->sparse_out && b->nib_out && b->lean && b->drec && b->packed && b->M == 1 && b->d_cov_row_win && b->d_cov8 &&
        (size_t)b->n_win <= b->cap_cov_row_win) {      // 4-bit plane + rows: the two coverage buffers change their roles
        b->nib_pass = true;
        a.cov4 = b->d_cov8; a.cov_rows = b->d_cov16; a.cov_row_win = b->d_cov_row_win;
        a.cap_cov_rows = (uint32_t)std::min<uint64_t>((uint64_t)(b->cap_pos ? b->cap_pos : b->n_pos), 0xFFFFFFFFu);
        a.cov8 = nullptr; a.cov16 = nullptr; a.sat_thr = 65535u;
    }
    a.clon_list = b->sparse_out ? b->d_clon_list : nullptr; a.cap_clon = (uint32_t)std::min<size_t>(b->cap_clon, 0xFFFFFFFFu);
    if (b->lean && b->sparse_out) {             // a lean slot writes only what travels home
        if (!b->clon_dense) a.clon = nullptr;
        if (!b->rare_dense && a.rare) a.clon_r = nullptr;
        if (a.cov8) a.cov16 = nullptr;
    }
#ifdef ISX_TUNING
    if (getenv("ISX_NO_COUNTS")) { a.counts = nullptr; if (atoi(getenv("ISX_NO_COUNTS")) > 1) a.clon = nullptr; }      // tuning builds only (tools/timeline.py): a resident batch run like a pipe slot
#endif
    a.stripe = (b->drec && b->packed && b->M == 1 && !a.counts && !(b->prm.layout & ISX_LAYOUT_NO_STRIPES)) ? 1 : 0;
    a.seed_lo = (uint32_t)b->prm.seed; a.seed_hi = (uint32_t)(b->prm.seed >> 32);
    a.entries = b->d_entries; a.slab = b->slab; a.cap_ovf = (uint32_t)b->cap_ovf;
    a.ovf0 = (uint64_t)b->n_win * b->slab; a.win_nent = b->d_win_nent;
    if (b->lev_sparse) {                        // mm path of a pipe slot: the levels go home as mask + coverage bytes + lists (PileupArgs::lev_*)
        a.lev_mask = b->d_lev_mask; a.lev_cov = b->d_lev_cov; a.lev_win_off = b->d_lev_win_off;
        a.lev_mask_bytes = b->lev_mask_bytes; a.lev_cov_bytes = b->lev_cov_bytes;
        a.cap_lev = (uint32_t)std::min<size_t>(b->cap_lev, 0xFFFFFFFEu);
        a.sat_thr = b->lev_cov_bytes == 1 ? 255u : 65535u;
        a.clon_list = b->d_clon_list; a.cap_clon = (uint32_t)std::min<size_t>(b->cap_clon, 0xFFFFFFFFu);
        a.cov16 = nullptr; a.cov8 = nullptr; a.clon = nullptr; a.clon_r = nullptr;
    }
    a.slev = b->d_slev; a.cap_slev = (uint32_t)std::min<size_t>(b->cap_slev, 0xFFFFFFFFu);
    a.snv = b->d_snv; a.cap_snv = (uint32_t)std::min<size_t>(b->cap_snv, 0xFFFFFFFFu);
    a.sites = b->d_sites; a.cap_sites = (uint32_t)std::min<size_t>(b->cap_sites, 0xFFFFFFFFu);
    a.ao = b->d_ao; a.cap_ao = (uint32_t)std::min<size_t>(b->cap_ao, 0xFFFFFFFFu); a.enable_linkage = b->prm.enable_linkage;
    a.cursors = b->d_cursors; a.flags = b->d_flags; a.host_state = b->d_host_state;
    memcpy(a.base, b->base, sizeof(a.base));

    // Publication of the cursors (k_publish_state) is deferred: if another pass follows on the stream, its
    // kernel publishes this one's state as it starts (one kernel per step, no extra boundary); otherwise
    // isx_batch_wait enqueues the one-wave kernel itself.
    if (c->unpublished[b->ps] && c->unpublished[b->ps] != b) {
        isx_batch *u = c->unpublished[b->ps];
        a.pub_cursors = u->d_cursors; a.pub_host_state = u->d_host_state; a.pub_epoch = u->epoch;
        u->publish_enqueued = true;
        c->unpublished[b->ps] = nullptr;
    }
    b->ordered = false;
    if (b->M == 1) {
        // position order by gather (isx_pileup.hip k_win_gather): the kernel writes scratch tables + one (slot, count) record per window
        auto grow = [&](auto **p, size_t *cap, size_t want, size_t elem) -> hipError_t {
            if (*cap >= want && *p) return hipSuccess;
            if (*p) isx_dev_free(*p);
            *p = nullptr; *cap = 0;
            const hipError_t e = isx_dev_malloc(reinterpret_cast<void **>(p), std::max<size_t>(want, 1) * elem);
            if (e == hipSuccess) *cap = want;
            return e;
        };
        HIP_TRY(grow(&b->d_snv_raw, &b->cap_snv_raw, b->cap_snv, sizeof(isx_snv)));
        HIP_TRY(grow(&b->d_sites_raw, &b->cap_sites_raw, b->cap_sites, sizeof(isx_site)));
        if (b->d_rare) HIP_TRY(grow(&b->d_rare_raw, &b->cap_rare_raw, b->cap_rare, sizeof(uint2)));
        if ((size_t)b->n_win > b->cap_win || !b->d_win_rec) {
            if (b->d_win_rec) isx_dev_free(b->d_win_rec);
            if (b->d_win_out) isx_dev_free(b->d_win_out);
            b->d_win_rec = b->d_win_out = nullptr;
            const size_t want = (size_t)b->n_win + (size_t)b->n_win / 4 + 64;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_win_rec), want * 8 * sizeof(uint32_t)));
            // (+ k_win_scan's chunk states behind the window offsets: 8 words per 1024 windows, zero = "no launch has written here")
            const size_t n_state = (want / 1024 + 2) * 8;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_win_out), (want * 4 + n_state) * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(b->d_win_out + want * 4, 0, n_state * sizeof(uint32_t), s));
            b->cap_win = want;
        }
        a.snv = b->d_snv_raw; a.sites = b->d_sites_raw;
        if (b->d_rare) a.rare = b->d_rare_raw;
        a.win_rec = b->d_win_rec;
    }
    if (b->M > 1 && b->prm.enable_linkage && b->prm.linkage_mode != 2) {
        // mm profiling on: a window's SNP sites lie side by side in the site table; (first, count) per window lets the linkage stages put
        // the table in position order window by window instead of sorting it (k_site_order)
        if ((size_t)b->n_win > b->cap_win_sites || !b->d_win_site_cnt) {
            if (b->d_win_site_base) isx_dev_free(b->d_win_site_base);
            if (b->d_win_site_cnt) isx_dev_free(b->d_win_site_cnt);
            b->d_win_site_base = b->d_win_site_cnt = nullptr; b->cap_win_sites = 0;
            const size_t want = (size_t)b->n_win + (size_t)b->n_win / 4 + 64;
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_win_site_base), want * sizeof(uint32_t)));
            HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b->d_win_site_cnt), want * sizeof(uint32_t)));
            b->cap_win_sites = want;
        }
        a.win_site_base = b->d_win_site_base; a.win_site_cnt = b->d_win_site_cnt;
    }
    launch_pileup(a, b->block, b->lds, b->grid, b->packed, s, b->ev[0], b->ev[1]);
    if (b->M == 1) {
        launch_win_order(b->d_win_rec, b->d_win_out, b->n_win, b->W, b->d_snv_raw, b->d_snv, b->d_sites_raw, b->d_sites,
                         a.clon_list, b->d_clon_sorted, b->d_rare ? b->d_rare_raw : nullptr, b->d_rare, b->d_win_out + b->cap_win * 4, b->epoch + 1, s);
        b->ordered = true;
    }
    HIP_TRY(hipGetLastError());
    ++b->epoch;
    b->publish_enqueued = false;
    c->unpublished[b->ps] = b;
    b->in_flight = true;
    return ISX_OK;
}

// wait for the pass enqueued by launch_pass and collect it (linkage stages run here: they need the
// table sizes on the host); *cap_flags receives the ISX_FLAG_CAP_* bits of tables that were too small
int finish_pass(isx_batch *b, uint32_t *cap_flags, hipStream_t link_stream)
{
    const int rc = finish_pass_sizes(b, cap_flags, link_stream);
    if (rc != ISX_OK || *cap_flags) return rc;
    return finish_pass_link(b, link_stream);
}

// the first half: the pass has run, its table sizes are on the host (no linkage stages yet; b->ran stays false until finish_pass_link)
int finish_pass_sizes(isx_batch *b, uint32_t *cap_flags, hipStream_t link_stream)
{
    *cap_flags = 0;
    isx_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = link_stream ? link_stream : c->stream;      // (a pipe's finishers each bring their own: two batches are finished side by side)
    b->in_flight = false;
    if (!b->publish_enqueued) {
        PileupArgs pa{};
        pa.cursors = b->d_cursors; pa.host_state = b->d_host_state;
        launch_publish_state(pa, b->epoch, c->pstream[b->ps]);
        b->publish_enqueued = true;
        if (c->unpublished[b->ps] == b) c->unpublished[b->ps] = nullptr;
    }
    {   // spin on the epoch word for a while (no interrupt latency), then fall back to a stream wait
        volatile uint32_t *ep = b->h_state + CUR_N + 4;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (;;) {
            if (__atomic_load_n(ep, __ATOMIC_ACQUIRE) == b->epoch) { seen = true; break; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(b->spin_us)) break;
        }
        if (!seen) HIP_TRY(isx_wait_stream(c->pstream[b->ps]));
    }
    uint32_t cur[CUR_N];
    for (int i = 0; i < CUR_N; i++) { cur[i] = b->h_state[i] - b->base[i]; b->base[i] = b->h_state[i]; }
    const uint32_t flags = b->h_state[CUR_N];
    if (flags) {        // error path: start the next run from a clean slate
        HIP_TRY(hipMemsetAsync(b->d_cursors, 0, (CUR_N + 4) * sizeof(uint32_t), s));
        HIP_TRY(isx_wait_stream(s));
        memset(b->base, 0, sizeof(b->base));
    }
    b->sites_loose = (flags & ISX_FLAG_SITES_LOOSE) != 0;
    if (flags & ISX_FLAG_MM_RANGE) { isx_set_error("an observation has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
    if (flags & (ISX_FLAG_CAP_ENTRIES | ISX_FLAG_CAP_SNV | ISX_FLAG_CAP_SITES | ISX_FLAG_CAP_AO)) {
        *cap_flags = flags & (ISX_FLAG_CAP_ENTRIES | ISX_FLAG_CAP_SNV | ISX_FLAG_CAP_SITES | ISX_FLAG_CAP_AO);
        return ISX_OK;
    }
    b->sizes = isx_sizes{};
    b->ld_host = nullptr; b->n_ld_host = 0; b->link_chain = 0;
    b->sizes.n_entries = cur[CUR_ENT_TOTAL];
    b->n_ovf = b->lev_sparse ? cur[CUR_ENT_TOTAL] : cur[CUR_ENTRIES];        // (a flat table = no slabs, every entry an "overflow" entry)
    b->sizes.n_snv = cur[CUR_SNV];
    b->sizes.n_sites = cur[CUR_SITES];
    b->n_rare = cur[CUR_RARE]; b->n_sat = cur[CUR_SAT]; b->n_clon = cur[CUR_CLON]; b->n_cov_rows = cur[CUR_COVX];
    b->tim = isx_timings{};
    b->tim_pending = true;                  // the kernel's own time stamps are read when somebody asks (isx_batch_timings)
    b->tim.pileup_blocks = b->grid;
    b->tim.pileup_threads = b->block;
    b->tim.pileup_lds_bytes = (int32_t)b->lds;
    b->tim.pileup_window = b->W;
    b->tim.record_bytes = b->d_seg ? 64 : (b->d_drec ? 32 : (b->d_rec16 ? 2 : (b->d_rec32 ? 4 : 8)));
    b->n_ao_pass = cur[CUR_AO];
    return ISX_OK;
}

// the second half: the linkage stages (they end with one wait for `link_stream`, which also delivers what the caller has queued with
// isx_read_back on that stream meanwhile -- a pipe's finisher enqueues its fetches first and waits once)
int finish_pass_link(isx_batch *b, hipStream_t link_stream)
{
    isx_ctx *c = b->ctx;
    hipStream_t s = link_stream ? link_stream : c->stream;
    if (b->prm.enable_linkage) {
        LinkageIn in{};
        in.stream = s; in.ev = &b->ev[2]; in.ev_mfma = &b->ev[8];
        in.mode = b->prm.linkage_mode == 2 ? 2 : 1;
        in.philox = Philox{(uint32_t)b->prm.seed, (uint32_t)(b->prm.seed >> 32)};
        in.n_pairs = b->n_pairs; in.ao = b->d_ao; in.n_ao = b->n_ao_pass;
        in.sites = b->d_sites; in.n_sites = (uint32_t)b->sizes.n_sites; in.sites_ordered = b->ordered;
        in.slev = b->d_slev; in.snv = b->d_snv;
        if (b->M > 1 && b->d_win_site_cnt && !b->sites_loose) { in.win_site_base = b->d_win_site_base; in.win_site_cnt = b->d_win_site_cnt; in.n_win = b->n_win; }
        in.split_bounds = b->d_bounds; in.n_splits = b->n_splits; in.M = b->M; in.min_snp = b->prm.min_snp;
        LinkageOut lo;
        int rc = run_linkage(in, b->L, lo);
        if (rc != ISX_OK) return rc;
        if (lo.chain != 3) HIP_TRY(isx_wait_stream(s));     // (the bucket chain ends with its one read-back: waited for already)
        b->ld_host = lo.ld_host; b->n_ld_host = (size_t)lo.n_ld_host; b->link_chain = lo.chain;
        b->sizes.n_allele_obs = (int64_t)lo.n_ao;
        b->sizes.n_increments = (int64_t)lo.n_increments;
        b->sizes.n_edges = (int64_t)lo.n_edges;
        b->sizes.n_ld = (int64_t)lo.n_ld;
        b->tim.sites_ms = ev_ms(b->ev[2], b->ev[3]);
        b->tim.allele_ms = ev_ms(b->ev[3], b->ev[4]);
        b->tim.group_ms = ev_ms(b->ev[4], b->ev[5]);
        b->tim.incr_ms = ev_ms(b->ev[5], b->ev[6]);
        b->tim.ld_ms = ev_ms(b->ev[6], b->ev[7]);
        b->tim.total_ms = ev_ms(b->ev[0], b->ev[7]);
        b->tim.mfma_ms = ev_ms(b->ev[8], b->ev[9]);
        b->tim.dense_tiles = (int32_t)lo.dense_tiles; b->tim.dense_macs = (int64_t)lo.dense_macs;
        b->tim.dense_bytes = (int64_t)lo.dense_bytes;
    } else {
        b->tim.total_ms = 0.f;
    }
    b->ran = true;
    return ISX_OK;
}

int batch_grow_tables(isx_batch *b, uint32_t cap_flags)
{
    // hard bounds come from the positions / observations the device buffers were sized for (a pipe slot
    // outlives its batches: cap_pos / cap_obs; a plain batch: its own n_pos / n_obs)
    const uint64_t pos_bound = (uint64_t)(b->cap_pos ? b->cap_pos : b->n_pos);
    const uint64_t obs_bound = (uint64_t)std::max<int64_t>(b->cap_obs ? b->cap_obs : b->n_obs, 1);
    const uint64_t npm = pos_bound * b->M;
    int rc = ISX_OK;
    HIP_TRY(hipSetDevice(b->ctx->device));
    if (cap_flags & ISX_FLAG_CAP_SNV) { if ((rc = regrow(&b->d_snv, &b->cap_snv, (size_t)npm))) return rc; }
    if (cap_flags & ISX_FLAG_CAP_SITES) {
        // the flag covers the site table and (mm path) the per-site level rows, first sized for 8 levels a
        // site: with the site table already at its bound only the level rows are short
        const bool slev_short = b->M > 1 && b->cap_slev < b->cap_sites * (size_t)b->M;
        if (b->cap_sites < (size_t)pos_bound || !slev_short) {
            if ((rc = regrow(&b->d_sites, &b->cap_sites, (size_t)pos_bound))) return rc;
        }
        if (b->M > 1) {
            if (b->d_slev) isx_dev_free(b->d_slev);
            b->d_slev = nullptr;
            b->cap_slev = b->cap_sites * (size_t)b->M;
            HIP_TRY(isx_raw_dev_malloc(&b->d_slev, b->cap_slev * sizeof(isx_slev)));
        }
    }
    if (cap_flags & ISX_FLAG_CAP_AO) { if ((rc = regrow(&b->d_ao, &b->cap_ao, (size_t)obs_bound))) return rc; }
    if ((cap_flags & ISX_FLAG_CAP_ENTRIES) && b->lev_sparse) {
        // level-sparse slot: the coverage stream (and the flat entry table, when kept) x4 up to one level per (position, mm bin)
        const size_t bound = (size_t)std::min<uint64_t>(npm, 0xFFFFFFF0ull);
        if (b->cap_lev >= bound) { isx_set_error("output table is at its hard bound and still too small"); return ISX_ERR_CAPACITY; }
        const size_t cap = std::min(bound, std::max<size_t>(b->cap_lev * 4, 1024));
        if (b->d_lev_cov) isx_dev_free(b->d_lev_cov);
        b->d_lev_cov = nullptr;
        HIP_TRY(isx_raw_dev_malloc(reinterpret_cast<uint8_t **>(&b->d_lev_cov), cap * 2 + 64));
        if (b->d_entries) {
            isx_dev_free(b->d_entries);
            b->d_entries = nullptr;
            HIP_TRY(isx_raw_dev_malloc(&b->d_entries, cap * sizeof(isx_entry)));
            b->cap_entries = cap;
        }
        b->cap_lev = cap;
    } else if (cap_flags & ISX_FLAG_CAP_ENTRIES) {
        const size_t used_slabs = (size_t)b->n_win * b->slab;
        const size_t slabs = b->slab_region ? b->slab_region : used_slabs;      // entries set aside for the window slabs
        size_t cap = b->cap_entries - slabs;
        isx_entry *dummy = nullptr;
        if ((rc = regrow(&dummy, &cap, (size_t)npm))) return rc;      // size check only
        isx_dev_free(dummy);
        if (slabs + cap >= 0xFFFFFFFFull) { isx_set_error("mm path: more than 2^32 entry slots in one batch"); return ISX_ERR_CAPACITY; }
        if (b->d_entries) isx_dev_free(b->d_entries);
        b->d_entries = nullptr;
        HIP_TRY(isx_raw_dev_malloc(&b->d_entries, (slabs + cap) * sizeof(isx_entry)));
        b->cap_entries = slabs + cap;
        b->cap_ovf = b->cap_entries - used_slabs;
    }
    return ISX_OK;
}

int isx_batch_launch(isx_batch *b)
{
    if (!b) { isx_set_error("isx_batch_launch: NULL batch"); return ISX_ERR_ARG; }
    if (b->in_flight) { isx_set_error("isx_batch_launch: the batch already has a pass in flight (isx_batch_wait first)"); return ISX_ERR_STATE; }
    return launch_pass(b);
}

int isx_batch_wait(isx_batch *b)
{
    if (!b) { isx_set_error("isx_batch_wait: NULL batch"); return ISX_ERR_ARG; }
    if (!b->in_flight) { isx_set_error("isx_batch_wait: no pass in flight"); return ISX_ERR_STATE; }
    // Output tables start from generous estimates; a table that turns out too small (e.g. SNS rows at
    // every position of a divergent reference) is grown x4 up to its hard bound and the pass repeated.
    for (int attempt = 0; attempt < 8; attempt++) {
        uint32_t cf = 0;
        int rc = ISX_OK;
        if (attempt > 0 && (rc = launch_pass(b)) != ISX_OK) return rc;
        rc = finish_pass(b, &cf);
        if (rc != ISX_OK) return rc;
        if (!cf) return ISX_OK;
        if ((rc = batch_grow_tables(b, cf)) != ISX_OK) return rc;
    }
    isx_set_error("output tables still too small after 8 growth steps");
    return ISX_ERR_CAPACITY;
}

int isx_batch_run(isx_batch *b)
{
    const int rc = isx_batch_launch(b);
    return rc != ISX_OK ? rc : isx_batch_wait(b);
}

int isx_batch_sizes(const isx_batch *b, isx_sizes *out)
{
    if (!b || !out) { isx_set_error("isx_batch_sizes: bad argument"); return ISX_ERR_ARG; }
    if (!b->ran) { isx_set_error("isx_batch_sizes: run the batch first"); return ISX_ERR_STATE; }
    *out = b->sizes;
    return ISX_OK;
}

int isx_batch_timings(const isx_batch *b, isx_timings *out)
{
    if (!b || !out) { isx_set_error("isx_batch_timings: bad argument"); return ISX_ERR_ARG; }
    if (!b->ran) { isx_set_error("isx_batch_timings: run the batch first"); return ISX_ERR_STATE; }
    if (b->tim_pending) {
        isx_batch *m = const_cast<isx_batch *>(b);
        HIP_TRY(isx_wait_event(b->ev[1]));
        m->tim.pileup_ms = ev_ms(b->ev[0], b->ev[1]);
        if (!b->prm.enable_linkage) m->tim.total_ms = m->tim.pileup_ms;
        m->tim_pending = false;
    }
    *out = b->tim;
    return ISX_OK;
}

#define NEED_RUN(b, out)                                                                       \
    if (!(b) || ((out) == nullptr)) { isx_set_error("fetch: bad argument"); return ISX_ERR_ARG; } \
    if (!(b)->ran) { isx_set_error("fetch: run the batch first"); return ISX_ERR_STATE; }        \
    HIP_TRY(hipSetDevice((b)->ctx->device));

int isx_batch_fetch_entries(isx_batch *b, isx_entry *out)
{
    NEED_RUN(b, out);
    if (b->M == 1) { isx_set_error("n_mm_bins == 1: use isx_batch_fetch_dense"); return ISX_ERR_STATE; }
    const size_t n = (size_t)b->sizes.n_entries;
    if (!n) return ISX_OK;
    // the device table is one slab per window (used prefix = win_nent[w]) + the overflow region
    if (!b->d_entries) { isx_set_error("this batch lives in a lean pipe slot (isx_pipe_params.lean_output): its 32-byte entries were not written -- the levels come with isx_pipe_collect (isx_pipe_result.lev_*)"); return ISX_ERR_STATE; }
    return fetch_entries_sorted(b->ctx->stream, b->d_entries, b->d_win_nent, (uint32_t)b->slab, entry_wins(b), b->n_ovf, n, out);
}

int isx_batch_fetch_dense(isx_batch *b, uint32_t *counts, float *clon, float *clon_rarefied)
{
    NEED_RUN(b, counts);
    if (b->M != 1) { isx_set_error("n_mm_bins > 1: use isx_batch_fetch_entries"); return ISX_ERR_STATE; }
    if (!b->d_counts) { isx_set_error("this pipe slot keeps no per-base count table (create the pipe with want_counts)"); return ISX_ERR_STATE; }
    HIP_TRY(hipMemcpy(counts, b->d_counts, (size_t)b->n_pos * sizeof(uint4), hipMemcpyDeviceToHost));
    if (clon) HIP_TRY(hipMemcpy(clon, b->d_clon, (size_t)b->n_pos * sizeof(float), hipMemcpyDeviceToHost));
    if (clon_rarefied) HIP_TRY(hipMemcpy(clon_rarefied, b->d_clon_r, (size_t)b->n_pos * sizeof(float), hipMemcpyDeviceToHost));
    return ISX_OK;
}

int isx_batch_summarize(isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, isx_scaffold_level *out,
                        float *device_ms)
{
    NEED_RUN(b, out);
    if (n_scaffolds <= 0 || !scaffold_bounds || scaffold_bounds[0] != 0 || scaffold_bounds[n_scaffolds] != b->n_pos) {
        isx_set_error("isx_batch_summarize: scaffold_bounds must span [0, n_pos]");
        return ISX_ERR_ARG;
    }
    for (int i = 0; i < n_scaffolds; i++)
        if (scaffold_bounds[i + 1] <= scaffold_bounds[i]) { isx_set_error("scaffold_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    SummaryIn in{};
    in.stream = b->ctx->stream; in.ev = b->ev_sum;
    in.n_pos = (uint32_t)b->n_pos; in.n_scaffolds = n_scaffolds; in.M = b->M; in.scaffold_bounds = scaffold_bounds;
    if (b->lean && !b->d_counts && !b->d_entries) { isx_set_error("this batch lives in a lean pipe slot (isx_pipe_params.lean_output): its dense coverage / clonality arrays were not written"); return ISX_ERR_STATE; }
    in.counts = b->d_counts; in.clon = b->d_clon; in.clon_r = b->d_clon_r;
    in.cov16 = b->d_cov16; in.sat = b->d_sat; in.n_sat = (uint32_t)std::min<size_t>(b->n_sat, b->cap_sat);
    in.entries = b->d_entries; in.win_nent = b->d_win_nent; in.slab = b->slab; in.n_win = entry_wins(b);
    in.n_ovf = b->n_ovf; in.ovf0 = (uint64_t)entry_wins(b) * b->slab;
    return run_summary(in, b->S, out, device_ms);
}

int isx_batch_summarize_genomes(isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t n_genomes,
                                const int32_t *genome_first_scaffold, int32_t mask_edges, isx_genome_level *out, float *device_ms)
{
    NEED_RUN(b, out);
    if (n_scaffolds <= 0 || !scaffold_bounds || scaffold_bounds[0] != 0 || scaffold_bounds[n_scaffolds] != b->n_pos) {
        isx_set_error("isx_batch_summarize_genomes: scaffold_bounds must span [0, n_pos]");
        return ISX_ERR_ARG;
    }
    for (int i = 0; i < n_scaffolds; i++)
        if (scaffold_bounds[i + 1] <= scaffold_bounds[i]) { isx_set_error("scaffold_bounds must be strictly ascending"); return ISX_ERR_ARG; }
    if (n_genomes <= 0 || !genome_first_scaffold || genome_first_scaffold[0] != 0 || genome_first_scaffold[n_genomes] != n_scaffolds || mask_edges < 0) {
        isx_set_error("isx_batch_summarize_genomes: genome_first_scaffold must span [0, n_scaffolds], mask_edges >= 0");
        return ISX_ERR_ARG;
    }
    for (int g = 0; g < n_genomes; g++)
        if (genome_first_scaffold[g + 1] <= genome_first_scaffold[g]) { isx_set_error("genome_first_scaffold must be strictly ascending (a genome = consecutive scaffolds of the batch)"); return ISX_ERR_ARG; }
    SummaryIn in{};
    in.stream = b->ctx->stream; in.ev = b->ev_sum;
    in.n_pos = (uint32_t)b->n_pos; in.n_scaffolds = n_scaffolds; in.M = b->M; in.scaffold_bounds = scaffold_bounds;
    if (b->lean && !b->d_counts && !b->d_entries) { isx_set_error("this batch lives in a lean pipe slot (isx_pipe_params.lean_output): its dense coverage / clonality arrays were not written"); return ISX_ERR_STATE; }
    in.counts = b->d_counts; in.clon = b->d_clon; in.clon_r = b->d_clon_r;
    in.cov16 = b->d_cov16; in.sat = b->d_sat; in.n_sat = (uint32_t)std::min<size_t>(b->n_sat, b->cap_sat);
    in.entries = b->d_entries; in.win_nent = b->d_win_nent; in.slab = b->slab; in.n_win = entry_wins(b);
    in.n_ovf = b->n_ovf; in.ovf0 = (uint64_t)entry_wins(b) * b->slab;
    return run_genome_summary(in, b->S, n_genomes, genome_first_scaffold, mask_edges, out, device_ms);
}

static void fill_summary_in(isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, SummaryIn &in)
{
    in.stream = b->ctx->stream; in.ev = b->ev_sum;
    in.n_pos = (uint32_t)b->n_pos; in.n_scaffolds = n_scaffolds; in.M = b->M; in.scaffold_bounds = scaffold_bounds;
    in.counts = b->d_counts; in.clon = b->d_clon; in.clon_r = b->d_clon_r;
    in.cov16 = b->d_cov16; in.sat = b->d_sat; in.n_sat = (uint32_t)std::min<size_t>(b->n_sat, b->cap_sat);
    in.entries = b->d_entries; in.win_nent = b->d_win_nent; in.slab = b->slab; in.n_win = entry_wins(b);
    in.n_ovf = b->n_ovf; in.ovf0 = (uint64_t)entry_wins(b) * b->slab;
}

int isx_compare_coverage(isx_batch *a, isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t min_cov,
                         isx_compare_level *out, float *device_ms)
{
    NEED_RUN(a, out);
    if (!b || !b->ran) { isx_set_error("isx_compare_coverage: run both batches first"); return ISX_ERR_STATE; }
    if (a->ctx != b->ctx || a->n_pos != b->n_pos) { isx_set_error("isx_compare_coverage: batches must share ctx and flat space"); return ISX_ERR_ARG; }
    if (n_scaffolds <= 0 || !scaffold_bounds || scaffold_bounds[0] != 0 || scaffold_bounds[n_scaffolds] != a->n_pos) {
        isx_set_error("isx_compare_coverage: scaffold_bounds must span [0, n_pos]");
        return ISX_ERR_ARG;
    }
    SummaryIn ia{}, ib{};
    if ((a->lean && !a->d_counts) || (b->lean && !b->d_counts)) { isx_set_error("a batch of a lean pipe slot (isx_pipe_params.lean_output) keeps no dense coverage / clonality arrays"); return ISX_ERR_STATE; }
    fill_summary_in(a, n_scaffolds, scaffold_bounds, ia);
    fill_summary_in(b, n_scaffolds, scaffold_bounds, ib);
    return run_compare(ia, ib, (uint32_t)std::max(min_cov, 0), CompareSnpIn(), a->C, out, device_ms);
}

int isx_compare_scaffolds(isx_batch *a, isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t min_cov,
                          double min_freq, isx_compare_level *out, int64_t *n_snp_rows, float *device_ms)
{
    NEED_RUN(a, out);
    if (!b || !b->ran) { isx_set_error("isx_compare_scaffolds: run both batches first"); return ISX_ERR_STATE; }
    if (a->ctx != b->ctx || a->n_pos != b->n_pos) { isx_set_error("isx_compare_scaffolds: batches must share ctx and flat space"); return ISX_ERR_ARG; }
    if (n_scaffolds <= 0 || !scaffold_bounds || scaffold_bounds[0] != 0 || scaffold_bounds[n_scaffolds] != a->n_pos) {
        isx_set_error("isx_compare_scaffolds: scaffold_bounds must span [0, n_pos]");
        return ISX_ERR_ARG;
    }
    if (!a->ctx->d_lut) { isx_set_error("isx_compare_scaffolds: set the null model first"); return ISX_ERR_STATE; }
    SummaryIn ia{}, ib{};
    if ((a->lean && !a->d_counts) || (b->lean && !b->d_counts)) { isx_set_error("a batch of a lean pipe slot (isx_pipe_params.lean_output) keeps no dense coverage / clonality arrays"); return ISX_ERR_STATE; }
    fill_summary_in(a, n_scaffolds, scaffold_bounds, ia);
    fill_summary_in(b, n_scaffolds, scaffold_bounds, ib);
    CompareSnpIn sn;
    sn.lut = a->ctx->d_lut; sn.lut_n = a->ctx->lut_n; sn.fallback = a->ctx->fallback; sn.min_freq = min_freq;
    sn.snv_a = a->d_snv; sn.n_a = (uint32_t)a->sizes.n_snv;
    sn.snv_b = b->d_snv; sn.n_b = (uint32_t)b->sizes.n_snv;
    const int rc = run_compare(ia, ib, (uint32_t)std::max(min_cov, 0), sn, a->C, out, device_ms);
    if (n_snp_rows) *n_snp_rows = rc == ISX_OK ? (int64_t)a->C.n_snp_rows : 0;
    return rc;
}

int isx_compare_fetch_snps(isx_batch *a, isx_compare_snp *out)
{
    NEED_RUN(a, out);
    const size_t n = a->C.n_snp_rows;
    if (!n) return ISX_OK;
    HIP_TRY(hipMemcpy(out, a->C.snp_rows, n * sizeof(isx_compare_snp), hipMemcpyDeviceToHost));
    std::sort(out, out + n, [](const isx_compare_snp &x, const isx_compare_snp &y) {
        return x.mm != y.mm ? x.mm < y.mm : x.gpos < y.gpos;
    });
    return ISX_OK;
}

int isx_batch_fetch_snv(isx_batch *b, isx_snv *out)
{
    NEED_RUN(b, out);
    const size_t n = (size_t)b->sizes.n_snv;
    if (!n) return ISX_OK;
    HIP_TRY(hipMemcpy(out, b->d_snv, n * sizeof(isx_snv), hipMemcpyDeviceToHost));
    std::sort(out, out + n, [](const isx_snv &x, const isx_snv &y) {
        return x.gpos != y.gpos ? x.gpos < y.gpos : x.mm < y.mm;
    });
    return ISX_OK;
}

// update_linked_reads' appends (linkage.py:254-283) as the device holds them after the linkage stages: one row per (read pair,
// SNP site, base in the site's `bases` set) -- what read_to_snvs / mm_to_position_graph of --store_everything are made of
int isx_batch_fetch_allele_obs(isx_batch *b, isx_allele_obs *out)
{
    NEED_RUN(b, out);
    const size_t n = (size_t)b->sizes.n_allele_obs, n_sites = (size_t)b->sizes.n_sites;
    if (!n) return ISX_OK;
    if (!b->prm.enable_linkage || !b->d_ao || !b->L.site_gpos.p) { isx_set_error("isx_batch_fetch_allele_obs: the batch was profiled without linkage"); return ISX_ERR_STATE; }
    static_assert(sizeof(isx_allele_obs) == sizeof(isx_ao), "isx_allele_obs mirrors isx_ao");
    std::vector<uint32_t> gpos(n_sites);
    HIP_TRY(hipMemcpy(out, b->d_ao, n * sizeof(isx_ao), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gpos.data(), b->L.site_gpos.p, n_sites * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {                // the linkage stages left the site's RANK in the row (k_ao_rank): back to its position
        if (out[i].gpos >= n_sites) { isx_set_error("internal: allele observation with a site rank beyond the site table"); return ISX_ERR_STATE; }
        out[i].gpos = gpos[out[i].gpos];
    }
    return ISX_OK;
}

int isx_batch_fetch_ld(isx_batch *b, isx_ld *out)
{
    NEED_RUN(b, out);
    const size_t n = (size_t)b->sizes.n_ld;
    if (!n) return ISX_OK;
    // rows are produced in (site1, site2, mm) order == (gpos_a, gpos_b, mm) order
    HIP_TRY(hipMemcpy(out, b->L.ld.p, n * sizeof(isx_ld), hipMemcpyDeviceToHost));
    return ISX_OK;
}

}  // extern "C"
