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
#include <vector>

#include "../../include/instrain_amd.h"
#include "obs_encode.h"

#define ISX_SEG_GROUP 16                // records per position base = one wave-wide 16-byte load (64 lanes x 16 B = 16 records)
#define ISX_SEG_REC_WORDS 16            // header + ISX_SEG_WORDS payload words
// reference-delta records (include/instrain_amd.h, ISX_DREC_*): 32 bytes, 32 per group = one wave-wide 16-byte load
#define ISX_DREC_GROUP 32

namespace isxenc {

struct SegJob {
    // input: arrays (isx_segs), or a producer that writes any range of the segment stream on demand (the BAM front end
    // emits segments straight into the encoder's per-task scratch: the batch's segments never exist as a whole)
    isx_segs in{};
    std::function<void(int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases)> produce;
    const uint32_t *gpos_all = nullptr;         // producer mode: the segment starts alone (the layout pass needs them up front)
    // bit-plane input (include/instrain_amd.h isx_read_planes; encode_planes): arrays, or a producer that writes any range of the
    // stream -- gpos / len / pair [count] and planes [count][ISX_PLANE_WORDS] -- into the task's scratch
    isx_read_planes in2{};
    std::function<void(int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint64_t *planes)> produce_planes;   // (mm: written when n_mm_bins > 1)
    // ... compared with the reference as it travels: the 2-bit plane (four positions a byte, anything that is not A/C/T/G as 0)
    // and the bit plane of those positions (NULL = the batch has none)
    const uint8_t *ref2 = nullptr, *refn = nullptr;
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
    // reference-delta records (encode_delta): the reference codes of the batch's flat space (1 byte per position, 0..3 = A C T G,
    // anything else = not a base) and the spare groups every task's region gets for the extra records of segments that differ
    // from the reference at more than ISX_DREC_EXC columns (a segment is then cut into pieces); need_slack = what the worst task
    // would have needed (> slack_groups: nothing usable was written, encode again with at least that)
    const uint8_t *ref = nullptr;
    int64_t slack_groups = 1;
    int64_t need_slack = 0;
    // every run reports the groups each task of 4096 segments really needs (task_need); a second run may be given exactly those
    // (task_groups, one entry per task) instead of "starts + slack_groups": one task over a stretch where the reference is not
    // A/C/T/G (every base there is an exception) then does not inflate the regions of all the others
    std::vector<int64_t> task_need;
    const int64_t *task_groups = nullptr;
    int64_t n_pieces = 0;                       // delta records written (>= n_seg)
    // results
    int64_t n_rec = 0;                          // device records, a multiple of ISX_SEG_GROUP
    int64_t n_bases = 0;                        // sum of the segment lengths (an upper bound of the observations)
    uint32_t max_pair = 0;
};

enum { SEG_OK = 0, SEG_CAPACITY = 1, SEG_MM_RANGE = 2, SEG_BAD_POS = 3, SEG_BAD_LEN = 4 };

int encode_segs(HostPool &pool, SegJob &job);
// the same stream as 32-byte reference-delta records (groups of ISX_DREC_GROUP); SEG_CAPACITY with need_slack > slack_groups
// means "encode again with more slack", otherwise the stream does not fit cap_rec
int encode_delta(HostPool &pool, SegJob &job);
// the same records from bit-plane input (SegJob::in2 / produce_planes, ref2 / refn): exceptions are found by XOR against the funnel-
// shifted 2-bit reference plane, 32 columns a step, the skip plane is copied.  Same layout rules, same result codes, and the same
// bytes as encode_delta gives for the segments the planes stand for.
int encode_planes(HostPool &pool, SegJob &job);
int64_t delta_groups_needed(HostPool &pool, const uint32_t *gpos, int64_t n, int64_t slack_groups);
int64_t seg_groups_needed(HostPool &pool, const uint32_t *gpos, int64_t n);
// reference codes (1 byte a position) -> the planes that travel (see include/instrain_amd.h isx_ref_planes); returns whether any
// position is not A/C/T/G
bool pack_ref_planes(HostPool &pool, const uint8_t *ref, int64_t n_pos, uint8_t *plane2, uint8_t *nplane);
// one segment's fifteen words of 3-bit codes -> its line of planes
void planes_from_words(const uint32_t *words, uint32_t len, uint64_t *planes);

}  // namespace isxenc
