"""Thin object layer over the C ABI: Context (one per GPU), Batch (resident set of splits),
BamFile (host BAM front end).  All compute happens in libinstrain_amd.so."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (ENTRY_DT, LD_DT, OBS_DT, SNV_DT, BamInfo, BamParams, IsxError, Params, Sizes,
                   Timings, check)


def pack_obs(gpos, base, mm):
    """SoA -> packed isx_obs records."""
    o = np.empty(len(gpos), dtype=OBS_DT)
    o["gpos"] = gpos
    o["mm"] = mm
    o["base"] = base
    o["flags"] = 0
    return o


SEQ_LUT = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACTG"):
    SEQ_LUT[ord(_c)] = _i


def encode_seq(seq):
    """upper-cased sequence string -> base codes (A,C,T,G = 0..3, else 4)"""
    return SEQ_LUT[np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)]


class Context:
    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.isx_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = device

    def set_null_model(self, lut, fallback):
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        check(self.lib.isx_set_null_model(self.h, lut.ctypes.data, len(lut), int(fallback)))

    def close(self):
        if self.h:
            self.lib.isx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A set of splits resident on the device (isx_batch_*)."""

    def __init__(self, ctx, ref_codes, split_bounds, obs, pair=None, min_cov=5, min_freq=0.05, min_snp=20,
                 rarefied_coverage=50, n_mm_bins=1, enable_linkage=True, linkage_mode=0, window=0, seed=0):
        self.ctx = ctx
        self.lib = ctx.lib
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        obs = np.ascontiguousarray(obs, dtype=OBS_DT)
        if pair is not None:
            pair = np.ascontiguousarray(pair, dtype=np.uint32)
            assert len(pair) == len(obs)
        self.n_pos = len(ref_codes)
        self.n_obs = len(obs)
        self.n_mm_bins = int(n_mm_bins)
        p = Params(int(min_cov), int(min_snp), float(min_freq), int(rarefied_coverage), int(n_mm_bins),
                   1 if enable_linkage else 0, int(linkage_mode), int(window), int(seed))
        h = C.c_void_p()
        check(self.lib.isx_batch_create(ctx.h, C.byref(p), self.n_pos, ref_codes.ctypes.data,
                                        len(split_bounds) - 1, split_bounds.ctypes.data, self.n_obs,
                                        obs.ctypes.data if self.n_obs else None,
                                        pair.ctypes.data if pair is not None and self.n_obs else None, C.byref(h)))
        self.h = h

    def run(self):
        check(self.lib.isx_batch_run(self.h))

    def launch(self):
        """enqueue one pass without waiting (isx_batch_launch); pair with wait()"""
        check(self.lib.isx_batch_launch(self.h))

    def wait(self):
        check(self.lib.isx_batch_wait(self.h))

    def sizes(self):
        s = Sizes()
        check(self.lib.isx_batch_sizes(self.h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in Sizes._fields_}

    def timings(self):
        t = Timings()
        check(self.lib.isx_batch_timings(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def pileup_ms(self):
        """device time of the pileup kernel of the last run (cheap accessor for timing loops)"""
        if not hasattr(self, "_tim"):
            self._tim = Timings()
        check(self.lib.isx_batch_timings(self.h, C.byref(self._tim)))
        return self._tim.pileup_ms

    def fetch(self):
        """-> dict(entries | (counts, clon), snv, ld) as numpy structured arrays (canonical order)."""
        s = self.sizes()
        out = {}
        if self.n_mm_bins == 1:
            counts = np.empty((self.n_pos, 4), dtype=np.uint32)
            clon = np.empty(self.n_pos, dtype=np.float32)
            clon_r = np.empty(self.n_pos, dtype=np.float32)
            check(self.lib.isx_batch_fetch_dense(self.h, counts.ctypes.data, clon.ctypes.data, clon_r.ctypes.data))
            out["counts"], out["clon"], out["clon_r"] = counts, clon, clon_r
        else:
            e = np.empty(max(1, s["n_entries"]), dtype=ENTRY_DT)
            check(self.lib.isx_batch_fetch_entries(self.h, e.ctypes.data))
            out["entries"] = e[:s["n_entries"]]
            out["clon_r"] = out["entries"]["clon_rarefied"]
        v = np.empty(max(1, s["n_snv"]), dtype=SNV_DT)
        check(self.lib.isx_batch_fetch_snv(self.h, v.ctypes.data))
        out["snv"] = v[:s["n_snv"]]
        l = np.empty(max(1, s["n_ld"]), dtype=LD_DT)
        check(self.lib.isx_batch_fetch_ld(self.h, l.ctypes.data))
        out["ld"] = l[:s["n_ld"]]
        return out

    def summarize(self, scaffold_bounds):
        """per-(scaffold, mm) aggregates of make_coverage_table -> (structured array [n_scaffolds, n_mm_bins], device ms)"""
        sb = np.ascontiguousarray(scaffold_bounds, dtype=np.int64)
        out = np.zeros((len(sb) - 1, self.n_mm_bins), dtype=_lib.SCAFFOLD_LEVEL_DT)
        ms = C.c_float(0)
        check(self.lib.isx_batch_summarize(self.h, len(sb) - 1, sb.ctypes.data, out.ctypes.data, C.byref(ms)))
        return out, ms.value

    def close(self):
        if self.h:
            self.lib.isx_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dense_to_entries(counts, clon):
    """dense (n_mm_bins == 1) result -> the same entry table the mm path returns (mm = 0)."""
    tot = counts.sum(axis=1)
    k = np.nonzero(tot > 0)[0]
    e = np.zeros(len(k), dtype=ENTRY_DT)
    e["gpos"] = k
    e["cnt"] = counts[k]
    e["clon"] = clon[k]
    e["clon_rarefied"] = np.nan
    return e


class BamFile:
    """Host BAM front end (isx_bam_*): BGZF/BAM decode + read-pair filter + htslib-1.9 pileup rules."""

    def __init__(self, path):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.isx_bam_open(path.encode(), C.byref(h)))
        self.h = h
        self.info = None

    def expand(self, min_read_ani=0.95, min_mapq=-1, max_insert_relative=3, min_insert=50,
               min_base_quality=30, skip_mm=False, window_length=10000, copy=True):
        """-> (obs, pair, split_bounds, split_ref).  copy=False: obs / pair are views of the handle's own
        arrays (isx_bam_view), valid until close() -- enough to build a Batch from them."""
        p = BamParams(float(min_read_ani), int(min_mapq), float(max_insert_relative), int(min_insert),
                      int(min_base_quality), 1 if skip_mm else 0, int(window_length), 0)
        info = BamInfo()
        check(self.lib.isx_bam_expand(self.h, C.byref(p), C.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in BamInfo._fields_ if n != "pad"}
        bounds = np.empty(info.n_splits + 1, dtype=np.int64)
        sref = np.empty(info.n_splits, dtype=np.int32)
        if copy or info.n_obs == 0:
            obs = np.empty(info.n_obs, dtype=OBS_DT)
            pair = np.empty(info.n_obs, dtype=np.uint32)
            check(self.lib.isx_bam_copy(self.h, obs.ctypes.data, pair.ctypes.data, bounds.ctypes.data, sref.ctypes.data))
        else:
            check(self.lib.isx_bam_copy(self.h, None, None, bounds.ctypes.data, sref.ctypes.data))
            po, pp = C.c_void_p(), C.c_void_p()
            check(self.lib.isx_bam_view(self.h, C.byref(po), C.byref(pp)))
            obs = np.frombuffer((C.c_uint8 * (info.n_obs * OBS_DT.itemsize)).from_address(po.value), dtype=OBS_DT)
            pair = np.frombuffer((C.c_uint32 * info.n_obs).from_address(pp.value), dtype=np.uint32)
        return obs, pair, bounds, sref

    def refs(self):
        out = []
        i = 0
        while True:
            name, ln, off = C.c_char_p(), C.c_int64(), C.c_int64()
            if self.lib.isx_bam_ref(self.h, i, C.byref(name), C.byref(ln), C.byref(off)) != 0:
                break
            out.append((name.value.decode(), ln.value, off.value))
            i += 1
        return out

    def close(self):
        if self.h:
            self.lib.isx_bam_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
