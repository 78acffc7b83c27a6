// bam_front.h -- what isx_pipe.hip needs from the BAM front end: a prepared batch of references whose observation
// stream can be pulled range by range (see bam_front.cpp BamBatch).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/instrain_amd.h"

struct BamBatch;
// reg_hi < 0: whole references; else n_refs == 1 and only positions [reg_lo, reg_hi) of it
// as_segments: the batch will be pulled as read segments (bam_batch_emit_segs) -- no per-base counting pass, bam_batch_n_obs is 0
int bam_batch_prepare(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, BamBatch **out, int64_t reg_lo = 0,
                      int64_t reg_hi = -1, bool as_segments = false);
int64_t bam_batch_n_segs(const BamBatch *q);
int64_t bam_batch_seg_bases(const BamBatch *q);          // columns the segments cover (an upper bound of the observations)
const uint32_t *bam_batch_seg_gpos(const BamBatch *q);  // [n_segs] flat start of every segment
void bam_batch_emit_segs(const BamBatch *q, int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair,
                         uint32_t *bases);               // thread safe
// the same segments as bit planes (isx_read_planes): gpos / len / pair [count], planes [count][ISX_PLANE_WORDS]; thread safe
void bam_batch_emit_planes(const BamBatch *q, int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint64_t *planes);  // mm: NULL with one mm bin
void bam_batch_free(BamBatch *q);
// done with, but not freed now: the batch goes with its handle (isx_bam_close), see isx_bam::retired
void bam_batch_retire(BamBatch *q);
int64_t bam_batch_n_obs(const BamBatch *q);
int64_t bam_batch_n_pos(const BamBatch *q);
void bam_batch_emit(const BamBatch *q, int64_t first, uint32_t count, isx_obs *obs, uint32_t *pair);     // thread safe
void bam_batch_info(const BamBatch *q, int32_t n_refs, isx_bam_info *info);
const std::vector<int64_t> &bam_batch_bounds(const BamBatch *q);
