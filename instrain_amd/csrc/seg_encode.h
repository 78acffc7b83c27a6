// seg_encode.h -- host side of the READ-LEVEL hand-over: read segments (include/instrain_amd.h isx_segs) -> the device
// record stream the k_pileup_* kernels walk (64-byte records in groups of 16 with one position base per group), written
// by the pool's threads straight into pinned staging, together with the per-group position directory the window ranges
// come from.
//
// Reference analogue: the pysam pileup columns a worker iterates (profile_utilities.py:150-153, 268-286) -- here the reads
// themselves travel and the columns are formed in LDS on the device.
#pragma once
#include <stdint.h>

#include <functional>

#include "../../include/instrain_amd.h"
#include "obs_encode.h"

#define ISX_SEG_GROUP 16                // records per position base = one wave-wide 16-byte load (64 lanes x 16 B = 16 records)
#define ISX_SEG_REC_WORDS 16            // header + ISX_SEG_WORDS payload words

namespace isxenc {

struct SegJob {
    // input: arrays (isx_segs), or a producer that writes any range of the segment stream on demand (the BAM front end
    // emits segments straight into the encoder's per-task scratch: the batch's segments never exist as a whole)
    isx_segs in{};
    std::function<void(int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases)> produce;
    const uint32_t *gpos_all = nullptr;         // producer mode: the segment starts alone (the layout pass needs them up front)
    int64_t n_seg = 0, n_pos = 0;
    int n_mm_bins = 1;
    bool want_pairs = false;
    // output memory
    uint32_t *rec = nullptr;                    // cap_rec records of 16 words
    uint32_t *gbase = nullptr;                  // cap_rec / 16
    uint32_t *pair_out = nullptr;               // cap_rec ids (NULL without linkage)
    uint32_t *cmin = nullptr, *cmax = nullptr;  // per group: lowest start / highest last position of its real records
    uint8_t *cany = nullptr;
    int64_t cap_rec = 0;
    // ring mode (ring_groups > 0): `rec` is not the whole stream but two halves of ring_groups groups each.  The tasks run in
    // waves whose groups fit one half; wave_begin(half) is called before a wave writes (the half's previous copy must have left
    // the host), wave_flush(half, g0, g1) after it: device groups [g0, g1) sit at the start of that half.  Both are called from
    // the thread that called encode_segs.  gbase / pair_out / the directory are never ringed (4 + 4 bytes per 64-byte record).
    int64_t ring_groups = 0;
    std::function<void(int half)> wave_begin;
    std::function<void(int half, int64_t g0, int64_t g1)> wave_flush;
    // results
    int64_t n_rec = 0;                          // device records, a multiple of ISX_SEG_GROUP
    int64_t n_bases = 0;                        // sum of the segment lengths (an upper bound of the observations)
    uint32_t max_pair = 0;
};

enum { SEG_OK = 0, SEG_CAPACITY = 1, SEG_MM_RANGE = 2, SEG_BAD_POS = 3, SEG_BAD_LEN = 4 };

int encode_segs(HostPool &pool, SegJob &job);
int64_t seg_groups_needed(HostPool &pool, const uint32_t *gpos, int64_t n);

}  // namespace isxenc
