// obs_encode.h -- host side of the streaming hand-over: a persistent thread pool and the encoder that turns
// the caller's 8-byte isx_obs records into the resident record stream of the kernels (2-byte records
// delta:13 | base:3 in groups of 512 when there is one mm bin, 4-byte records delta:16 | mm:8 | base:3 in
// groups of 256 otherwise) directly inside pinned staging memory, one pass over the input.
//
// Reference analogue: the pysam objects a worker receives per pileup column
// (/root/reference/inStrain/profile/profile_utilities.py:268-286) -- here the same visits arrive as packed
// records and this is the only host touch of them between the producer and the DMA engine.
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/instrain_amd.h"

namespace isxenc {

// Persistent worker threads (the reference's worker pool, profile_controller.py:243-271, feeds splits to
// processes; here the pool only encodes / copies).  Threads are spread over the L3 domains (CCDs) of the
// NUMA node the GPU hangs off when `numa_node` >= 0: a CCD's link to memory, not the core, bounds a
// streaming copy, so 16 threads on 16 CCDs move several times what 16 threads on 2 CCDs do.
class HostPool {
public:
    HostPool(int n_threads, int numa_node, bool pin);
    ~HostPool();
    int size() const { return (int)th_.size() + 1; }
    // fn(task) for every task in [0, n_tasks): dynamic hand-out, the caller works too; returns when all are done
    void run(int n_tasks, const std::function<void(int)> &fn);

private:
    void worker(int idx);
    std::vector<std::thread> th_;
    std::vector<std::vector<int>> cpus_;        // per worker: cpu list it may run on (empty = anywhere)
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_tasks_ = 0, next_ = 0, running_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct PairRun { uint32_t first, pair; };    // device records [first, next run's first) belong to read pair `pair`

struct EncodeJob {
    // input (BAM arrival order): arrays, or a producer that materialises any range of the input on demand
    // (the BAM front end expands reads straight into the encoder: the 8-byte records never exist as a whole)
    const isx_obs *obs = nullptr;
    const uint32_t *pair = nullptr;     // may be NULL
    std::function<void(int64_t first, uint32_t count, isx_obs *obs_out, uint32_t *pair_out)> produce;   // when obs == NULL
    bool want_pairs = false;            // producer mode: ask the producer for pair ids too
    int64_t n_obs = 0, n_pos = 0;
    int record_bytes = 2;               // 2: short stream (one mm bin), 4: compact stream
    // output memory (pinned staging in the library; any host memory in tests)
    void *rec = nullptr;                // cap_rec records of record_bytes
    uint32_t *gbase = nullptr;          // cap_rec / group
    uint32_t *pair_out = nullptr;       // cap_rec pair ids per device record, or NULL ...
    // ... and / or the same as runs of equal ids (what the pipe uploads: ~0.06 B per record instead of 4; padding records
    // continue the run before them): runs_out[cap_runs] ascending by first record, run_index_out[n_rec / 1024] = the run that
    // holds the first record of every 1024-record chunk.  Both are written by the pool's threads.
    PairRun *runs_out = nullptr;
    size_t cap_runs = 0;
    uint32_t *run_index_out = nullptr;
    size_t n_runs = 0;                  // result (> cap_runs: the table did not fit, nothing was written)
    uint32_t *cmin = nullptr, *cmax = nullptr;   // per ISX_CHUNK (1024) device records: position range ...
    uint8_t *cany = nullptr;            // ... and whether the chunk holds a real record
    int64_t cap_rec = 0;
    // ring mode (ring_groups > 0): `rec` is not the whole stream but two halves of ring_groups device groups each.  The tasks
    // run in waves whose regions fit one half; wave_begin(half) is called before a wave writes (the half's previous copy must
    // have left the host), wave_flush(half, g0, g1) after it: device groups [g0, g1) sit at the start of that half.  The
    // padding up to n_rec is part of the last wave.  Both are called from the thread that called encode_obs.
    int64_t ring_groups = 0;
    std::function<void(int half)> wave_begin;
    std::function<void(int half, int64_t g0, int64_t g1)> wave_flush;
    double slack = 0.0;                 // expected extra device groups per input group (0 = none: a stream without jumps)
    // results
    int64_t n_rec = 0;                  // device records (multiple of 2048)
    int64_t n_groups_real = 0;          // device groups that hold records
    int64_t n_groups_in = 0;            // ceil(n_obs / group)
    uint32_t max_pair = 0;
    int passes = 0;                     // 1, or 2 when the first layout overflowed (jumps beyond the slack)
};

enum { ENC_OK = 0, ENC_CAPACITY = 1, ENC_MM_RANGE = 2, ENC_BAD_POS = 3 };

// one pass (two when the stream jumps more often than `slack` allows) over the input, parallel over `pool`
int encode_obs(HostPool &pool, EncodeJob &job);

// which instruction set the encoder picked on this host ("avx512", "avx2", "scalar")
const char *encode_isa();

}  // namespace isxenc
