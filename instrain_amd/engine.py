"""Thin object layer over the C ABI: Context (one per GPU), Batch (resident set of splits),
BamFile (host BAM front end).  All compute happens in libinstrain_amd.so."""
import ctypes as C
import os

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


class SegBatch:
    """Read segments of a batch (isx_segs): gpos u32 [n], len u8 [n], mm u8 [n] | None, pair u32 [n] | None,
    bases u32 [n, 15] (ten 3-bit codes per word).  Keeps the arrays alive for the ctypes view."""

    def __init__(self, gpos, length, bases, mm=None, pair=None):
        self.gpos = np.ascontiguousarray(gpos, dtype=np.uint32)
        self.len = np.ascontiguousarray(length, dtype=np.uint8)
        self.bases = np.ascontiguousarray(bases, dtype=np.uint32).reshape(-1, _lib.SEG_WORDS)
        self.mm = None if mm is None else np.ascontiguousarray(mm, dtype=np.uint8)
        self.pair = None if pair is None else np.ascontiguousarray(pair, dtype=np.uint32)
        n = len(self.gpos)
        assert len(self.len) == n and len(self.bases) == n
        assert self.mm is None or len(self.mm) == n
        assert self.pair is None or len(self.pair) == n
        self.n_seg = n

    def c(self, with_pair=True):
        ptr = lambda a: a.ctypes.data if a is not None and len(a) else None
        return _lib.Segs(self.n_seg, ptr(self.gpos), ptr(self.len), ptr(self.mm), ptr(self.pair) if with_pair else None,
                         ptr(self.bases))

    @property
    def n_bases(self):
        return int(self.len.sum(dtype=np.int64))


def _aligned(shape, dtype, align=64):
    """uninitialised C-contiguous array whose data pointer is `align`-byte aligned"""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.empty(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def _own_pages(n_bytes):
    """uninitialised uint8 array of n_bytes that starts on a page and shares its last page with nothing"""
    raw = np.empty(((int(n_bytes) + 4095) // 4096 + 1) * 4096, dtype=np.uint8)
    off = (-raw.ctypes.data) % 4096
    return raw[off:off + int(n_bytes)]


class PlaneBatch:
    """Read segments of a batch as bit planes (isx_read_planes): gpos u32 [n], len u8 [n], pair u32 [n] | None, planes u64 [n, 8] --
    words 0-4 the 2-bit base codes of the columns (A C T G = 0 1 2 3), words 5-7 the columns that are not observed.  One 64-byte line a
    segment; one mm bin only.  from_segs(SegBatch) converts (isx_planes_from_segs)."""

    def __init__(self, gpos, length, planes, pair=None, mm=None):
        self.gpos = np.ascontiguousarray(gpos, dtype=np.uint32)
        self.len = np.ascontiguousarray(length, dtype=np.uint8)
        self.mm = None if mm is None else np.ascontiguousarray(mm, dtype=np.uint8)      # the pairs' mm levels (mm profiling on), see isx_read_planes.mm
        planes = np.asarray(planes, dtype=np.uint64).reshape(-1, _lib.PLANE_WORDS)
        if not planes.flags.c_contiguous or planes.ctypes.data % 64:
            a = _aligned(planes.shape, np.uint64)
            a[...] = planes
            planes = a
        self.planes = planes
        self.pair = None if pair is None else np.ascontiguousarray(pair, dtype=np.uint32)
        n = len(self.gpos)
        assert len(self.len) == n and len(self.planes) == n and (self.pair is None or len(self.pair) == n) and (self.mm is None or len(self.mm) == n)
        self.n_seg = n

    @classmethod
    def from_segs(cls, segs, threads=1):
        planes = _aligned((segs.n_seg, _lib.PLANE_WORDS), np.uint64)
        cs = segs.c()
        check(_lib.load().isx_planes_from_segs(C.byref(cs), int(threads), planes.ctypes.data if segs.n_seg else None))
        return cls(segs.gpos, segs.len, planes, segs.pair, segs.mm)

    def c(self, with_pair=True):
        ptr = lambda a: a.ctypes.data if a is not None and len(a) else None
        return _lib.ReadPlanes(self.n_seg, ptr(self.gpos), ptr(self.len), ptr(self.pair) if with_pair else None, ptr(self.planes), ptr(self.mm))

    @property
    def n_bases(self):
        return int(self.len.sum(dtype=np.int64))


class RefPlanes:
    """The reference of a batch as it travels (isx_ref_planes): plane2 u8 [(n_pos + 3) // 4] (2 bits a position, anything that is not
    A/C/T/G as 0) and nplane u8 [(n_pos + 7) // 8] | None (the positions that are not A/C/T/G).  from_codes packs reference codes."""

    def __init__(self, plane2, nplane, n_pos, key=0):
        self.plane2 = np.ascontiguousarray(plane2, dtype=np.uint8)
        self.nplane = None if nplane is None else np.ascontiguousarray(nplane, dtype=np.uint8)
        self.n_pos = int(n_pos)
        self.key = int(key)         # != 0: the caller's name for this content -- a pipe keeps the planes of a key on the device after their first trip
        assert len(self.plane2) >= (self.n_pos + 3) // 4 and (self.nplane is None or len(self.nplane) >= (self.n_pos + 7) // 8)

    @classmethod
    def from_codes(cls, ref_codes, threads=1, key=0):
        ref = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        n = len(ref)
        # (whole pages of their own: the planes can be registered for the copy engine -- register() -- without sharing a page with anything else)
        p2, pn = _own_pages((n + 3) // 4), _own_pages((n + 7) // 8)
        has = C.c_int32(0)
        check(_lib.load().isx_pack_ref_planes(ref.ctypes.data, n, int(threads), p2.ctypes.data, pn.ctypes.data, C.byref(has)))
        return cls(p2, pn if has.value else None, n, key)

    def c(self, keyed=True):
        return _lib.RefPlanes(self.plane2.ctypes.data, None if self.nplane is None else self.nplane.ctypes.data, self.key if keyed else 0)

    def register(self):
        """pin the planes for the copy engine (isx_host_register): a pipe then copies them to the device from where they lie instead of
        through its staging.  Once per reference, like loading the fasta; unregister() before the arrays go."""
        if getattr(self, "_registered", False):
            return self
        lib = _lib.load()
        check(lib.isx_host_register(C.c_void_p(self.plane2.ctypes.data), C.c_int64(self.plane2.nbytes)))
        if self.nplane is not None:
            try:
                check(lib.isx_host_register(C.c_void_p(self.nplane.ctypes.data), C.c_int64(self.nplane.nbytes)))
            except Exception:
                lib.isx_host_unregister(C.c_void_p(self.plane2.ctypes.data))
                raise
        self._registered = True
        return self

    def unregister(self):
        if not getattr(self, "_registered", False):
            return
        lib = _lib.load()
        lib.isx_host_unregister(C.c_void_p(self.plane2.ctypes.data))
        if self.nplane is not None:
            lib.isx_host_unregister(C.c_void_p(self.nplane.ctypes.data))
        self._registered = False


def pack_codes(codes):
    """[n, 150] uint8 base codes (0..3 A C T G, 4 skip, 5 non-ACGT) -> [n, 15] uint32, ten codes per word"""
    c = np.ascontiguousarray(codes, dtype=np.uint32).reshape(-1, _lib.SEG_WORDS, 10)
    sh = (3 * np.arange(10, dtype=np.uint32))[None, None, :]
    return (c << sh).sum(axis=2, dtype=np.uint32)


def unpack_codes(bases):
    """inverse of pack_codes: [n, 15] uint32 -> [n, 150] uint8"""
    b = np.ascontiguousarray(bases, dtype=np.uint32).reshape(-1, _lib.SEG_WORDS, 1)
    sh = (3 * np.arange(10, dtype=np.uint32))[None, None, :]
    return ((b >> sh) & 7).astype(np.uint8).reshape(-1, _lib.SEG_BASES)


SEQ_LUT = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACTG"):
    SEQ_LUT[ord(_c)] = _i


def encode_seq(seq):
    """upper-cased sequence string -> base codes (A,C,T,G = 0..3, else 4)"""
    return SEQ_LUT[np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)]


class Context:
    def __init__(self, device=0, reserve_cus=None):
        """reserve_cus: CUs of each XCD kept free of pileup kernels for the pipes' side queues (isx_ctx_reserve_cus; None = the
        library's default) -- for contexts that stream batches with linkage through a Pipe"""
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.isx_ctx_create(int(device), C.byref(h)))
        self.h = h
        if reserve_cus is not None:
            check(self.lib.isx_ctx_reserve_cus(self.h, int(reserve_cus)))
        self.device = device
        self._children = []         # weak references to live Batch / Pipe objects: closed before the context
        self._closers = []          # threads closing objects handed to close_later

    def _adopt(self, obj):
        import weakref
        self._children.append(weakref.ref(obj))

    def close_later(self, *objs):
        """close() the objects (pipes, BAM handles), in order, on a helper thread; the context waits for it before it goes"""
        import threading

        def run():
            for o in objs:
                try:
                    if isinstance(o, BamFile):
                        o.close(wait=True)          # (wait_closers() then really waits for the memory)
                    else:
                        o.close()
                except Exception:
                    pass
        self._closers = [t for t in self._closers if t.is_alive()]
        t = threading.Thread(target=run, daemon=True)
        t.start()
        self._closers.append(t)

    def set_null_model(self, lut, fallback):
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        check(self.lib.isx_set_null_model(self.h, lut.ctypes.data, len(lut), int(fallback)))

    def wait_closers(self):
        """block until what close_later was handed is gone (a benchmark that times call after call does this in between)"""
        for t in self._closers:
            t.join()
        self._closers = []

    def close(self):
        self.wait_closers()
        if self.h:
            for r in self._children:
                o = r()
                if o is not None:
                    o.close()
            self._children = []
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
                 rarefied_coverage=50, n_mm_bins=1, enable_linkage=True, linkage_mode=0, window=0, seed=0, layout=0):
        self.ctx = ctx
        self.lib = ctx.lib
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        if isinstance(obs, SegBatch):               # read-level batch (isx_batch_create_reads)
            segs = obs
            self.n_pos, self.n_obs, self.n_mm_bins = len(ref_codes), segs.n_bases, int(n_mm_bins)
            p = Params(int(min_cov), int(min_snp), float(min_freq), int(rarefied_coverage), int(n_mm_bins),
                       1 if enable_linkage else 0, int(linkage_mode), int(window), int(seed), int(layout), 0)
            cs = segs.c(with_pair=bool(enable_linkage))
            h = C.c_void_p()
            check(self.lib.isx_batch_create_reads(ctx.h, C.byref(p), self.n_pos, ref_codes.ctypes.data, len(split_bounds) - 1,
                                                  split_bounds.ctypes.data, C.byref(cs), C.byref(h)))
            self.h = h
            ctx._adopt(self)
            return
        obs = np.ascontiguousarray(obs, dtype=OBS_DT)
        if pair is not None:
            pair = np.ascontiguousarray(pair, dtype=np.uint32)
            assert len(pair) == len(obs)
        self.n_pos = len(ref_codes)
        self.n_obs = len(obs)
        self.n_mm_bins = int(n_mm_bins)
        p = Params(int(min_cov), int(min_snp), float(min_freq), int(rarefied_coverage), int(n_mm_bins),
                   1 if enable_linkage else 0, int(linkage_mode), int(window), int(seed), int(layout), 0)
        h = C.c_void_p()
        check(self.lib.isx_batch_create(ctx.h, C.byref(p), self.n_pos, ref_codes.ctypes.data,
                                        len(split_bounds) - 1, split_bounds.ctypes.data, self.n_obs,
                                        obs.ctypes.data if self.n_obs else None,
                                        pair.ctypes.data if pair is not None and self.n_obs else None, C.byref(h)))
        self.h = h
        ctx._adopt(self)

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

    def fetch_allele_obs(self):
        """update_linked_reads' appends (isx_batch_fetch_allele_obs) -> structured array (pair, gpos, order, mm, base)"""
        n = int(self.sizes()["n_allele_obs"])
        out = np.empty(max(1, n), dtype=_lib.AO_DT)
        check(self.lib.isx_batch_fetch_allele_obs(self.h, out.ctypes.data))
        return out[:n]

    def summarize(self, scaffold_bounds):
        """per-(scaffold, mm) aggregates of make_coverage_table -> (structured array [n_scaffolds, n_mm_bins], device ms)"""
        sb = np.ascontiguousarray(scaffold_bounds, dtype=np.int64)
        out = np.zeros((len(sb) - 1, self.n_mm_bins), dtype=_lib.SCAFFOLD_LEVEL_DT)
        ms = C.c_float(0)
        check(self.lib.isx_batch_summarize(self.h, len(sb) - 1, sb.ctypes.data, out.ctypes.data, C.byref(ms)))
        return out, ms.value

    def summarize_genomes(self, scaffold_bounds, genome_first_scaffold, mask_edges=100):
        """genome-level coverage roll-up (genomeUtilities.genomeLevel_coverage_info without iRep): a genome = consecutive
        scaffolds of the batch -> (structured array [n_genomes, n_mm_bins], device ms)"""
        sb = np.ascontiguousarray(scaffold_bounds, dtype=np.int64)
        gf = np.ascontiguousarray(genome_first_scaffold, dtype=np.int32)
        out = np.zeros((len(gf) - 1, self.n_mm_bins), dtype=_lib.GENOME_LEVEL_DT)
        ms = C.c_float(0)
        check(self.lib.isx_batch_summarize_genomes(self.h, len(sb) - 1, sb.ctypes.data, len(gf) - 1, gf.ctypes.data, int(mask_edges),
                                                   out.ctypes.data, C.byref(ms)))
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


class _SlotBatch(Batch):
    """A collected pipe slot seen as a Batch (fetch of entries / ld rows, summaries); owned by the pipe."""

    def __init__(self, ctx, handle, n_pos, n_obs, n_mm_bins):
        self.ctx, self.lib, self.h = ctx, ctx.lib, C.c_void_p(handle)
        self.n_pos, self.n_obs, self.n_mm_bins = int(n_pos), int(n_obs), int(n_mm_bins)

    def close(self):
        self.h = None


class Pipe:
    """Streaming hand-over (isx_pipe_*): submit batches, collect each one's tables once.

    submit() encodes the batch into pinned staging on the pipe's host threads and enqueues copy-in, the
    pass and copy-out; collect() blocks until that batch's tables are on the host and returns them as
    numpy views of the slot's pinned result block (valid until release())."""

    def __init__(self, ctx, max_pos, max_obs, max_splits, depth=4, host_threads=0, pin_threads=True, jump_slack=0.0,
                 min_cov=5, min_freq=0.05, min_snp=20, rarefied_coverage=50, n_mm_bins=1, enable_linkage=False,
                 linkage_mode=0, window=0, seed=0, layout=0, want_counts=False, ring_kib=0, max_segs=0, stage_async=False, lean_output=False):
        self.ctx, self.lib = ctx, ctx.lib
        self._held = {}                             # stage_async: what a queued batch still reads, by ticket
        self._wires = []                            # staged batches (freed with the pipe at the latest)
        self.stage_async = bool(stage_async) and max_segs > 0
        self.n_mm_bins = int(n_mm_bins)
        self.min_cov = int(min_cov)
        self.want_counts = bool(want_counts)
        self.enable_linkage = bool(enable_linkage)
        p = Params(int(min_cov), int(min_snp), float(min_freq), int(rarefied_coverage), int(n_mm_bins),
                   1 if enable_linkage else 0, int(linkage_mode), int(window), int(seed), int(layout), 0)
        pp = _lib.PipeParams(int(max_pos), int(max_obs), int(max_splits), int(depth), int(host_threads),
                             1 if pin_threads else 0, float(jump_slack), 1 if want_counts else 0, int(ring_kib),
                             1 if self.stage_async else 0, 1 if lean_output else 0, int(max_segs))
        self.read_level = max_segs > 0
        h = C.c_void_p()
        check(self.lib.isx_pipe_create(ctx.h, C.byref(p), C.byref(pp), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def submit(self, ref_codes, split_bounds, obs, pair=None):
        """-> ticket.  Arrays must be C-contiguous uint8 / int64 / OBS_DT / uint32 (no copies are made here)."""
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        obs = np.ascontiguousarray(obs, dtype=OBS_DT)
        if pair is not None:
            pair = np.ascontiguousarray(pair, dtype=np.uint32)
        t = C.c_int64(-1)
        check(self.lib.isx_pipe_submit(self.h, len(ref_codes), ref_codes.ctypes.data, len(split_bounds) - 1,
                                       split_bounds.ctypes.data, len(obs), obs.ctypes.data if len(obs) else None,
                                       pair.ctypes.data if pair is not None and len(obs) else None, C.byref(t)))
        return t.value

    def submit_reads(self, ref_codes, split_bounds, segs):
        """-> ticket.  A read-level batch (SegBatch) through a pipe created with max_segs > 0 (isx_pipe_submit_reads)."""
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        cs = segs.c(with_pair=self.enable_linkage)
        t = C.c_int64(-1)
        check(self.lib.isx_pipe_submit_reads(self.h, len(ref_codes), ref_codes.ctypes.data, len(split_bounds) - 1,
                                             split_bounds.ctypes.data, C.byref(cs), C.byref(t)))
        if self.stage_async:                        # the stager reads these until the batch is collected / released
            self._held[t.value] = (ref_codes, segs, cs)
        return t.value

    def stage_reads(self, ref_codes, split_bounds, segs):
        """-> Wire: the batch staged once into a pinned image of its own (isx_pipe_stage_reads); submit_wire(wire) then costs no
        host work.  The wire stays valid until wire.close() / the pipe's close()."""
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        cs = segs.c(with_pair=self.enable_linkage)
        h = C.c_void_p()
        check(self.lib.isx_pipe_stage_reads(self.h, len(ref_codes), ref_codes.ctypes.data, len(split_bounds) - 1, split_bounds.ctypes.data,
                                            C.byref(cs), C.byref(h)))
        w = Wire(self, h)
        self._wires.append(w)
        return w

    def submit_planes(self, ref_planes, split_bounds, reads, keyed=True):
        """-> ticket.  A read-level batch as bit planes (PlaneBatch) against the reference planes (RefPlanes): isx_pipe_submit_planes.
        keyed=False ignores ref_planes.key (the reference travels like any other)"""
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        cr, cf = reads.c(with_pair=self.enable_linkage), ref_planes.c(keyed)
        t = C.c_int64(-1)
        check(self.lib.isx_pipe_submit_planes(self.h, ref_planes.n_pos, C.byref(cf), len(split_bounds) - 1, split_bounds.ctypes.data,
                                              C.byref(cr), C.byref(t)))
        if self.stage_async:                        # the stager reads these until the batch is collected / released
            self._held[t.value] = (ref_planes, reads, cr, cf)
        return t.value

    def set_reference_budget(self, mib):
        """device memory this pipe may spend on resident references (RefPlanes.key), MiB"""
        check(self.lib.isx_pipe_set_reference_budget(self.h, int(mib)))

    def stage_planes(self, ref_planes, split_bounds, reads):
        """-> Wire, like stage_reads, from bit planes (isx_pipe_stage_planes)"""
        split_bounds = np.ascontiguousarray(split_bounds, dtype=np.int64)
        cr, cf = reads.c(with_pair=self.enable_linkage), ref_planes.c()
        h = C.c_void_p()
        check(self.lib.isx_pipe_stage_planes(self.h, ref_planes.n_pos, C.byref(cf), len(split_bounds) - 1, split_bounds.ctypes.data,
                                             C.byref(cr), C.byref(h)))
        w = Wire(self, h)
        self._wires.append(w)
        return w

    def submit_wire(self, wire):
        """-> ticket.  A staged batch (stage_reads) into the next free slot: DMA copies + pass + copy-out, nothing else"""
        t = C.c_int64(-1)
        check(self.lib.isx_pipe_submit_wire(self.h, wire.h, C.byref(t)))
        return t.value

    def submit_bam(self, bamfile, refs, ref_codes, split_bounds=None, **kw):
        """-> ticket.  The references `refs` (ascending indices) of a scanned + filtered BamFile go straight from the
        front end into the slot's staging (isx_pipe_submit_bam); kw = the expansion flags of BamFile.expand_refs.
        bamfile.info holds the batch's counts afterwards."""
        refs = np.ascontiguousarray(refs, dtype=np.int32)
        ref_codes = np.ascontiguousarray(ref_codes, dtype=np.uint8)
        sb = None if split_bounds is None else np.ascontiguousarray(split_bounds, dtype=np.int64)
        p = BamFile._params(**kw)
        info = BamInfo()
        t = C.c_int64(-1)
        rc = self.lib.isx_pipe_submit_bam(self.h, bamfile.h, C.byref(p), refs.ctypes.data, len(refs), ref_codes.ctypes.data,
                                          0 if sb is None else len(sb) - 1, None if sb is None else sb.ctypes.data,
                                          C.byref(info), C.byref(t))
        bamfile._info(info)                         # the batch's real size is known even when it did not fit the pipe
        check(rc)
        return t.value

    def collect(self, ticket, want_ld=True, rare_list=True, densify=True, shrunk_entries=False):
        """-> dict like Batch.fetch() (+ 'sizes', 'stats'); the dense arrays / snv rows are views of pinned memory.
        One mm bin without want_counts: the tables come back shrunk -- 'cov16' or (a shallow batch) 'cov8', 'saturated' =
        exact coverage of the positions beyond that range, and 'clon_sparse' = the clonalities other than 1.0 (every other
        position with coverage >= min_cov has exactly 1.0); densify rebuilds the 'cov16' / 'clon' arrays from them
        (False: the caller reads the shrunk forms, see dense_clon)."""
        r = _lib.PipeResult()
        check(self.lib.isx_pipe_collect(self.h, int(ticket), C.byref(r)))
        sz = {n: getattr(r.sizes, n) for n, _ in Sizes._fields_}
        out = {"sizes": sz, "ticket": r.ticket, "rows_checksum": int(r.rows_checksum),
               "stats": {k: getattr(r, k) for k in ("encode_ms", "h2d_ms", "kernel_ms", "d2h_ms", "collect_wait_ms",
                                                    "record_bytes", "encode_passes", "h2d_bytes", "d2h_bytes")}}
        n_pos = int(r.n_pos)

        def view(addr, dtype, n):
            if not n:
                return np.empty(0, dtype=dtype)
            return np.frombuffer((C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype)

        slot = _SlotBatch(self.ctx, r.batch, n_pos, r.n_obs, self.n_mm_bins)
        if self.n_mm_bins == 1:
            # the shrunk tables (what shrink_basewise keeps): coverage, clonality, sparse rarefied clonality
            # (a shallow batch comes back smaller still: 1-byte coverage, clonality as a sorted (position, value) list)
            if r.coverage16:
                out["cov16"] = view(r.coverage16, np.uint16, n_pos)
            elif r.coverage4:                       # a lean slot's shallow batch: 4-bit plane + 16-bit rows of the windows beyond 15 (dense_cov)
                out["cov4"] = view(r.coverage4, np.uint8, (n_pos + 1) // 2)
                out["cov_window"] = int(r.cov_window)
                out["cov_row_win"] = view(r.cov_row_window, np.uint32, int(r.n_cov_rows)).copy()
                out["cov_rows"] = view(r.cov_rows, np.uint16, int(r.n_cov_rows) * int(r.cov_window)).reshape(-1, int(r.cov_window))
            else:
                out["cov8"] = view(r.coverage8, np.uint8, n_pos)
            if r.clon:
                out["clon"] = view(r.clon, np.float32, n_pos)
            else:
                out["clon_sparse"] = view(r.clon_sparse, _lib.CLON_DT, int(r.n_clon))
            if r.n_saturated and r.saturated:
                out["saturated"] = view(r.saturated, _lib.SAT_DT, int(r.n_saturated)).copy()
            if densify and ("cov8" in out or "cov4" in out):
                out["cov16"] = dense_cov(out, n_pos)
                for k in ("cov8", "cov4", "cov_rows", "cov_row_win", "cov_window"):
                    out.pop(k, None)
            if densify and "clon_sparse" in out:
                out["clon"] = dense_clon(out["cov16"], out.pop("clon_sparse"), self.min_cov)
            if r.clon_rarefied:                     # want_counts, or a deep sample (the list would not be sparse)
                out["clon_r"] = view(r.clon_rarefied, np.float32, n_pos)
            if r.rare:
                out["rare"] = view(r.rare, _lib.RARE_DT, int(r.n_rare))
            elif r.clon_rarefied and not rare_list:
                pass                                # the caller cuts the dense array itself (profile_bam: per split, on demand)
            elif r.clon_rarefied:                   # the library handed the dense array instead of the list
                k = np.flatnonzero(~np.isnan(out["clon_r"]))
                out["rare"] = np.empty(len(k), dtype=_lib.RARE_DT)
                out["rare"]["gpos"] = k
                out["rare"]["clon_rarefied"] = out["clon_r"][k]
            else:
                out["rare"] = np.empty(0, dtype=_lib.RARE_DT)
            out["n_saturated"] = int(r.n_saturated)
            if r.counts:                            # want_counts: the full tables as Batch.fetch() returns them
                out["counts"] = view(r.counts, np.uint32, n_pos * 4).reshape(n_pos, 4)
                if "clon_r" not in out:
                    out["clon_r"] = np.full(n_pos, np.nan, np.float32)
        else:
            n_e = sz["n_entries"]
            got = False
            if r.lev_mask:
                # level-sparse hand-back (isx_pipe_result.lev_*): the level mask per position, one coverage element per present level,
                # the lists; views of the slot's result block like the dense tables
                out["levels"] = {"mask": view(r.lev_mask, {1: np.uint8, 2: np.uint16, 4: np.uint32}[int(r.lev_mask_bytes)], n_pos),
                                 "cov": view(r.lev_cov, np.uint8 if r.lev_cov_bytes == 1 else np.uint16, int(r.n_lev)),
                                 "win_off": view(r.lev_win_off, np.uint32, int(r.n_lev_windows)), "window": int(r.lev_window),
                                 "clon": view(r.lev_clon, _lib.RARE_DT, int(r.n_lev_clon)), "rare": view(r.lev_rare, _lib.RARE_DT, int(r.n_lev_rare)),
                                 "sat": view(r.lev_sat, _lib.SAT_DT, int(r.n_lev_sat)), "n": int(r.n_lev)}
                if shrunk_entries and densify:      # the four columns shrink_basewise's inputs are cut from, made on the host (isx_levels_expand)
                    out["entries_soa"] = self.expand_levels(r)
                    got = True
                elif shrunk_entries:
                    got = True                      # the caller reads the level tables as they came (expand_levels(result) later)
                    out["_result"] = r
            if shrunk_entries and not got:
                # what shrink_basewise keeps of a (position, mm) level is its coverage, not its four counts: four 4-byte columns
                # (isx_pipe_fetch_entries_shrunk) instead of 32-byte entries; a level deeper than 2^24 -> the full entries
                cols = (np.empty(max(1, n_e), np.uint32), np.empty(max(1, n_e), np.uint32), np.empty(max(1, n_e), np.float32), np.empty(max(1, n_e), np.float32))
                rc = self.lib.isx_pipe_fetch_entries_shrunk(self.h, int(ticket), *(c.ctypes.data for c in cols))
                if rc == 0:
                    out["entries_soa"] = tuple(c[:n_e] for c in cols)
                    got = True
                elif rc != _lib.ERR_CAPACITY:
                    check(rc)
            if not got:
                e = np.empty(max(1, n_e), dtype=ENTRY_DT)
                check(self.lib.isx_pipe_fetch_entries(self.h, int(ticket), e.ctypes.data))
                out["entries"] = e[:n_e]
                out["clon_r"] = out["entries"]["clon_rarefied"]
        out["snv"] = view(r.snv, SNV_DT, sz["n_snv"])
        if want_ld and self.enable_linkage:
            out["ld"] = view(r.ld, LD_DT, sz["n_ld"]) if r.ld else np.empty(0, dtype=LD_DT)
        else:
            out["ld"] = np.empty(0, dtype=LD_DT)
        out["slot"] = slot
        return out

    def expand_levels(self, r, threads=0):
        """The level-sparse tables of a collected batch (the isx_pipe_result `r`, or collect()'s dict with densify=False) as the four
        columns gpos | mm << 24 | coverage | clon | clon_rarefied in (gpos, mm) order: isx_levels_expand, host work only."""
        if isinstance(r, dict):
            r = r["_result"]
        n = max(1, int(r.n_lev))
        cols = [np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.float32), np.empty(n, np.float32) if r.n_lev_rare else None]
        check(self.lib.isx_levels_expand(C.byref(r), int(threads) or min(16, len(os.sched_getaffinity(0))), *(c.ctypes.data if c is not None else None for c in cols)))
        if cols[3] is None:                     # no level reaches the rarefied coverage: one NaN, broadcast (read-only; nobody writes the columns)
            cols[3] = np.broadcast_to(np.float32(np.nan), (n,))
        return tuple(c[:int(r.n_lev)] for c in cols)

    def levels_copy(self, r):
        """The level-sparse tables of a collected batch (collect()'s dict with densify=False) copied out of the slot's result block:
        a LevelTables that outlives release() and makes the four columns when somebody asks (LevelTables.columns())."""
        return LevelTables(self.lib, r["_result"] if isinstance(r, dict) else r)

    def release(self, ticket):
        try:
            check(self.lib.isx_pipe_release(self.h, int(ticket)))
        finally:
            self._held.pop(int(ticket), None)

    def close(self):
        h, self.h = self.h, None                    # (a second close, from another thread, finds nothing to do)
        if h:
            self.lib.isx_pipe_destroy(h)            # (stages and finishes what is still queued)
            for w in self._wires:
                w.close()
        self._wires = []
        self._held.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_COPY_POOL = None


def _par_copy(a):
    """a.copy() on a few threads for the large ones (numpy's copy releases the GIL; one thread moves ~6 GB/s out of pinned memory)"""
    global _COPY_POOL
    if a.nbytes < (8 << 20):
        return a.copy()
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(6)
    out = np.empty_like(a)
    n = len(a)
    step = -(-n // 12)
    list(_COPY_POOL.map(lambda k: np.copyto(out[k:k + step], a[k:k + step]), range(0, n, step)))
    return out


class LevelTables:
    """Own copies of a batch's level-sparse tables (isx_pipe_result.lev_*: mask per position, one coverage element per present level, the
    windows' first level indices, the lists) -- 1-3 bytes a level.  columns() expands them to gpos | mm << 24 | coverage | clon |
    clon_rarefied in (gpos, mm) order (isx_levels_expand, host work) the first time it is called and keeps the result."""

    def __init__(self, lib, r):
        self.lib = lib
        n_pos, n_lev, n_win = int(r.n_pos), int(r.n_lev), int(r.n_lev_windows)

        def own(addr, dtype, n):
            if not n or not addr:
                return np.empty(0, dtype=dtype)
            return _par_copy(np.frombuffer((C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype))

        self.mask = own(r.lev_mask, {1: np.uint8, 2: np.uint16, 4: np.uint32}[int(r.lev_mask_bytes)], n_pos)
        self.cov = own(r.lev_cov, np.uint8 if r.lev_cov_bytes == 1 else np.uint16, n_lev)
        self.win_off = own(r.lev_win_off, np.uint32, n_win)
        self.clon = own(r.lev_clon, _lib.RARE_DT, int(r.n_lev_clon))
        self.rare = own(r.lev_rare, _lib.RARE_DT, int(r.n_lev_rare))
        self.sat = own(r.lev_sat, _lib.SAT_DT, int(r.n_lev_sat))
        q = _lib.PipeResult()
        q.n_pos, q.n_lev, q.n_lev_clon, q.n_lev_rare, q.n_lev_sat = n_pos, n_lev, len(self.clon), len(self.rare), len(self.sat)
        q.lev_mask_bytes, q.lev_cov_bytes, q.lev_window, q.n_lev_windows, q.lev_min_cov = r.lev_mask_bytes, r.lev_cov_bytes, r.lev_window, r.n_lev_windows, r.lev_min_cov
        ptr = lambda a: a.ctypes.data if len(a) else None
        q.lev_mask, q.lev_cov, q.lev_win_off = ptr(self.mask), ptr(self.cov) or ptr(self.win_off), ptr(self.win_off)
        q.lev_clon, q.lev_rare, q.lev_sat = ptr(self.clon), ptr(self.rare), ptr(self.sat)
        self._r = q
        self._cols = None
        self.n = n_lev

    def columns(self, threads=0):
        if self._cols is None:
            n = max(1, self.n)
            cols = [np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.float32), np.empty(n, np.float32) if len(self.rare) else None]
            check(self.lib.isx_levels_expand(C.byref(self._r), int(threads) or min(16, len(os.sched_getaffinity(0))), *(c.ctypes.data if c is not None else None for c in cols)))
            if cols[3] is None:
                cols[3] = np.broadcast_to(np.float32(np.nan), (n,))
            self._cols = tuple(c[:self.n] for c in cols)
        return self._cols


class Wire:
    """A staged batch (isx_wire): pinned image + geometry, made by Pipe.stage_reads"""

    def __init__(self, pipe, h):
        self.lib, self.h, self.pipe_h = pipe.lib, h, pipe.h
        self.bytes = int(self.lib.isx_wire_bytes(h))

    def keep_reference(self):
        """isx_wire_keep_reference: the batch's reference planes stay on the device; later submits copy only the sample's part"""
        check(self.lib.isx_wire_keep_reference(self.pipe_h, self.h))
        self.bytes = int(self.lib.isx_wire_bytes(self.h))

    def close(self):
        h, self.h = self.h, None
        if h:
            self.lib.isx_wire_free(h)

    def __del__(self):
        pass                                        # (freed by its pipe's close(): a wire must outlive the batches submitted from it)


def encode_obs(obs, pair=None, n_pos=None, record_bytes=2, threads=1, slack=0.0, cap_rec=None, ring_records=0):
    """isx_encode_obs / isx_encode_obs_ring (host only): -> (rec, gbase, pair_out | None, passes)"""
    lib = _lib.load()
    obs = np.ascontiguousarray(obs, dtype=OBS_DT)
    G = 512 if record_bytes == 2 else 256
    if cap_rec is None:
        cap_rec = (int(len(obs) * 1.5) + 4 * 2048 + 2047) // 2048 * 2048
    if n_pos is None:
        n_pos = int(obs["gpos"].max()) + 1 if len(obs) else 1
    rec = np.empty(cap_rec, dtype=np.uint16 if record_bytes == 2 else np.uint32)
    gbase = np.empty(cap_rec // G, dtype=np.uint32)
    pout = np.empty(cap_rec, dtype=np.uint32) if pair is not None else None
    if pair is not None:
        pair = np.ascontiguousarray(pair, dtype=np.uint32)
    n_rec, passes = C.c_int64(0), C.c_int32(0)
    check(lib.isx_encode_obs_ring(obs.ctypes.data if len(obs) else None, pair.ctypes.data if pair is not None else None, len(obs),
                                  int(n_pos), int(record_bytes), int(threads), float(slack), int(cap_rec), int(ring_records),
                                  rec.ctypes.data, gbase.ctypes.data, pout.ctypes.data if pout is not None else None,
                                  C.byref(n_rec), C.byref(passes)))
    n = n_rec.value
    return rec[:n], gbase[:n // G], (pout[:n] if pout is not None else None), passes.value


def dense_clon(cov, clon_sparse, min_cov):
    """clonT of a one-mm-bin batch from its shrunk hand-back: 1.0 where the coverage reaches min_cov, the listed values at the
    listed positions, NaN elsewhere (a saturated 8- / 16-bit coverage entry is >= any sensible min_cov)"""
    out = np.where(np.asarray(cov) >= min_cov, np.float32(1.0), np.float32(np.nan)).astype(np.float32)
    out[clon_sparse["gpos"]] = clon_sparse["clon"]
    return out


def encode_segs(segs, n_pos, n_mm_bins=1, threads=1, cap_rec=None, ring_records=0):
    """isx_encode_segs / isx_encode_segs_ring (host only): SegBatch -> (rec [n_rec, 16] uint32, gbase [n_rec / 16], pair_out | None)"""
    lib = _lib.load()
    if cap_rec is None:                     # exactly what the encoder will cut (a sparse stream closes a group every few segments)
        cap_rec = int(lib.isx_seg_records_needed(segs.gpos.ctypes.data if segs.n_seg else None, segs.n_seg, int(threads)))
    rec = np.empty((cap_rec, 16), dtype=np.uint32)
    gbase = np.empty(cap_rec // 16, dtype=np.uint32)
    pout = np.empty(cap_rec, dtype=np.uint32) if segs.pair is not None else None
    n_rec = C.c_int64(0)
    cs = segs.c()
    check(lib.isx_encode_segs_ring(C.byref(cs), int(n_pos), int(n_mm_bins), int(threads), int(cap_rec), int(ring_records), rec.ctypes.data,
                                   gbase.ctypes.data, pout.ctypes.data if pout is not None else None, C.byref(n_rec)))
    n = n_rec.value
    return rec[:n], gbase[:n // 16], (pout[:n] if pout is not None else None)


def encode_delta(segs, ref_codes, n_mm_bins=1, threads=1, slack_groups=1, cap_rec=None, ring_records=0, retry=True):
    """isx_encode_delta (host only): SegBatch + reference codes -> (rec [n_rec, 8] uint32, gbase [n_rec / 32], None, slack_groups used);
    the read-pair ids travel inside the records (decode_delta returns them).  retry: encode again with the slack the first attempt
    asked for"""
    lib = _lib.load()
    ref = np.ascontiguousarray(ref_codes, dtype=np.uint8)
    while True:
        cap = cap_rec if cap_rec is not None else int(lib.isx_delta_records_needed(segs.gpos.ctypes.data if segs.n_seg else None, segs.n_seg, int(threads), int(slack_groups)))
        rec = np.empty((cap, 8), dtype=np.uint32)
        gbase = np.empty(cap // 32, dtype=np.uint32)
        n_rec, need = C.c_int64(0), C.c_int64(0)
        cs = segs.c()
        rc = lib.isx_encode_delta(C.byref(cs), ref.ctypes.data, len(ref), int(n_mm_bins), int(threads), int(slack_groups), cap, int(ring_records),
                                  rec.ctypes.data, gbase.ctypes.data, None, C.byref(n_rec), C.byref(need))
        if rc == _lib.ERR_CAPACITY and retry and need.value > slack_groups and cap_rec is None:
            slack_groups = int(need.value)
            continue
        check(rc)
        n = n_rec.value
        return rec[:n], gbase[:n // 32], None, slack_groups


def encode_planes(reads, ref_planes, threads=1, slack_groups=1, cap_rec=None, ring_records=0, retry=True, n_mm_bins=1):
    """isx_encode_planes (host only): PlaneBatch + RefPlanes -> (rec [n_rec, 8] uint32, gbase [n_rec / 32], None, slack_groups used) --
    the same records encode_delta gives for the segments the planes stand for"""
    lib = _lib.load()
    while True:
        cap = cap_rec if cap_rec is not None else int(lib.isx_delta_records_needed(reads.gpos.ctypes.data if reads.n_seg else None, reads.n_seg, int(threads), int(slack_groups)))
        rec = _aligned((cap, 8), np.uint32)
        gbase = np.empty(cap // 32, dtype=np.uint32)
        n_rec, need = C.c_int64(0), C.c_int64(0)
        cr, cf = reads.c(), ref_planes.c()
        rc = lib.isx_encode_planes_mm(C.byref(cr), C.byref(cf), ref_planes.n_pos, int(n_mm_bins), int(threads), int(slack_groups), cap, int(ring_records),
                                      rec.ctypes.data, gbase.ctypes.data, C.byref(n_rec), C.byref(need))
        if rc == _lib.ERR_CAPACITY and retry and need.value > slack_groups and cap_rec is None:
            slack_groups = int(need.value)
            continue
        check(rc)
        n = n_rec.value
        return rec[:n], gbase[:n // 32], None, slack_groups


def dense_cov(res, n_pos=None):
    """coverage per position (uint16, capped at 65535) of a collected batch in whichever shrunk form it came home: 'cov16', 'cov8'
    (+ 'saturated'), or the lean slots' 'cov4' plane + 'cov_rows' / 'cov_row_win' / 'cov_window' (isx_pipe_result.coverage4)"""
    if "cov16" in res:
        return res["cov16"]
    if "cov8" in res:
        cov = res["cov8"].astype(np.uint16)
    else:
        nib = res["cov4"]
        n_pos = 2 * len(nib) if n_pos is None else int(n_pos)
        cov = np.empty(2 * len(nib), dtype=np.uint16)
        cov[0::2] = nib & 15
        cov[1::2] = nib >> 4
        cov = cov[:n_pos]
        W = int(res["cov_window"])
        for k, w in enumerate(res["cov_row_win"].tolist()):
            lo = w * W
            hi = min(lo + W, n_pos)
            cov[lo:hi] = res["cov_rows"][k, :hi - lo]
    if "saturated" in res:
        cov[res["saturated"]["gpos"]] = np.minimum(res["saturated"]["coverage"], 65535)
    return cov


DREC_DUAL = 0x80000000
DREC_NO_EXC = 0x3FFFFFFF


def decode_delta(rec, gbase, ref_codes):
    """reference-delta record stream -> (gpos, len, mm, codes [n, 150], pair, full) of its pieces in stream order (a dual record's first
    half before its second) -- tests; code 4 where a column is skipped or beyond the piece's length; full: the piece is a full record"""
    rec = np.asarray(rec, dtype=np.uint32).reshape(-1, 8)
    ref = np.asarray(ref_codes, dtype=np.uint8)
    n = len(rec)
    dual = (rec[:, 0] >> 31) == 1
    assert ((rec[dual, 4] >> 31) == 1).all(), "a dual record's second half carries the flag too"
    zero = np.zeros(n, dtype=np.uint32)
    none = np.full(n, DREC_NO_EXC, dtype=np.uint32)
    # per half: header, pair id, two exception words, five skip words
    hdr = np.stack([rec[:, 0], np.where(dual, rec[:, 4], zero)], axis=1)
    pair = np.stack([np.where(dual, rec[:, 1], rec[:, 7]), np.where(dual, rec[:, 5], zero)], axis=1)
    e0 = np.stack([np.where(dual, rec[:, 2], rec[:, 3]), np.where(dual, rec[:, 6], none)], axis=1)
    e1 = np.stack([np.where(dual, rec[:, 3], none), np.where(dual, rec[:, 7], none)], axis=1)
    skw = [np.stack([np.where(dual, zero, rec[:, k]), zero], axis=1) for k in (1, 2, 4, 5, 6)]
    gb = np.repeat(np.asarray(gbase, dtype=np.uint32), 32)[:n]
    start = (gb[:, None] + (hdr & 0xFFFF)).astype(np.int64)
    ln = ((hdr >> 16) & 0xFF).astype(np.int64)
    lvl = ((hdr >> 24) & 0x7F).astype(np.uint8)            # the pair's mm level (0 with one mm bin)
    full = np.stack([~dual, np.zeros(n, dtype=bool)], axis=1)
    real = (ln > 0).reshape(-1)
    flat = lambda x: x.reshape(-1)[real]
    st, ln, pr, e0, e1, fl, lvl = flat(start), flat(ln), flat(pair), flat(e0), flat(e1), flat(full), flat(lvl)
    words = np.stack([flat(w) for w in skw], axis=1)
    m = len(st)
    j = np.arange(160, dtype=np.int64)[None, :]
    skip = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(m, 160).astype(bool)
    assert not (skip & (j >= ln[:, None])).any(), "skip bits beyond a piece's length"
    assert skip[fl].any(axis=1).all() or True
    inside = j < ln[:, None]
    pos = np.minimum(st[:, None] + j, len(ref) - 1)
    codes = np.where(inside & ~skip, ref[pos], 4).astype(np.uint8)
    for e in (e0, e1):
        for k in range(3):
            f = (e >> (10 * k)) & 0x3FF
            has = f != 0x3FF
            off, base = (f & 0xFF).astype(np.int64), ((f >> 8) & 3).astype(np.uint8)
            rows = np.flatnonzero(has)
            assert (off[rows] < ln[rows]).all()
            # an exception at a SKIPPED column (mm profiling on, full records only): a base that is not A/C/T/G -- code 5
            mark = skip[rows, off[rows]]
            assert fl[rows[mark]].all(), "a non-ACGT marker in a dual half"
            codes[rows[mark], off[rows[mark]]] = 5
            rows = rows[~mark]
            assert (base[rows] != ref[st[rows] + off[rows]]).all(), "an exception that equals the reference"
            codes[rows, off[rows]] = base[rows]
        assert ((e >> 30) == 0).all()
    return st.astype(np.uint32), ln.astype(np.uint8), lvl, codes[:, :150], pr, fl


def decode_segs(rec, gbase):
    """device record stream -> (gpos, len, mm, codes [n, 150]) of its real records, in stream order (tests)"""
    rec = np.asarray(rec, dtype=np.uint32).reshape(-1, 16)
    hdr = rec[:, 0]
    ln = (hdr >> 16) & 0xFF
    real = ln > 0
    start = np.repeat(np.asarray(gbase, dtype=np.uint32), 16)[:len(rec)] + (hdr & 0xFFFF)
    return start[real], ln[real].astype(np.uint8), (hdr[real] >> 24).astype(np.uint8), unpack_codes(rec[real, 1:])


def pack_reads(ref_start, clip_lo, clip_hi, cigars, seqs, quals, mm=None, pair=None, min_base_quality=30):
    """isx_pack_reads (host only): per read its flat reference start, its scaffold's [clip_lo, clip_hi) in flat space, its CIGAR
    (uint32 array, BAM encoding), bases (str / bytes) and qualities (uint8 array) -> SegBatch"""
    lib = _lib.load()
    n = len(ref_start)
    rs = np.ascontiguousarray(ref_start, dtype=np.int64)
    lo = np.ascontiguousarray(clip_lo, dtype=np.int64)
    hi = np.ascontiguousarray(clip_hi, dtype=np.int64)
    cig_off = np.zeros(n + 1, dtype=np.int64)
    seq_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        cig_off[1:] = np.cumsum([len(c) for c in cigars])
        seq_off[1:] = np.cumsum([len(q) for q in quals])
    cig = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars]) if n else np.zeros(0, np.uint32))
    seq = b"".join(s.encode() if isinstance(s, str) else bytes(s) for s in seqs)
    qual = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in quals]) if n else np.zeros(0, np.uint8))
    assert len(seq) == len(qual)
    mmv = None if mm is None else np.ascontiguousarray(mm, dtype=np.uint8)
    pv = None if pair is None else np.ascontiguousarray(pair, dtype=np.uint32)
    ns = C.c_int64(0)
    check(lib.isx_count_read_segs(n, cig.ctypes.data, cig_off.ctypes.data, rs.ctypes.data, lo.ctypes.data, hi.ctypes.data, C.byref(ns)))
    cap = max(1, ns.value)
    g = np.empty(cap, np.uint32); ln = np.empty(cap, np.uint8); sm = np.zeros(cap, np.uint8); sp = np.zeros(cap, np.uint32)
    bs = np.empty((cap, _lib.SEG_WORDS), np.uint32)
    check(lib.isx_pack_reads(n, rs.ctypes.data, lo.ctypes.data, hi.ctypes.data, cig.ctypes.data, cig_off.ctypes.data, seq,
                             qual.ctypes.data, seq_off.ctypes.data, mmv.ctypes.data if mmv is not None else None,
                             pv.ctypes.data if pv is not None else None, int(min_base_quality), cap, g.ctypes.data, ln.ctypes.data,
                             sm.ctypes.data, sp.ctypes.data, bs.ctypes.data, C.byref(ns)))
    k = ns.value
    return SegBatch(g[:k], ln[:k], bs[:k], sm[:k] if mm is not None else None, sp[:k] if pair is not None else None)


def pack_read_planes(ref_start, clip_lo, clip_hi, cigars, seqs, quals, pair=None, min_base_quality=30):
    """isx_pack_read_planes (host only): pack_reads with bit planes as output -> PlaneBatch"""
    lib = _lib.load()
    n = len(ref_start)
    rs = np.ascontiguousarray(ref_start, dtype=np.int64)
    lo = np.ascontiguousarray(clip_lo, dtype=np.int64)
    hi = np.ascontiguousarray(clip_hi, dtype=np.int64)
    cig_off = np.zeros(n + 1, dtype=np.int64)
    seq_off = np.zeros(n + 1, dtype=np.int64)
    if n:
        cig_off[1:] = np.cumsum([len(c) for c in cigars])
        seq_off[1:] = np.cumsum([len(q) for q in quals])
    cig = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars]) if n else np.zeros(0, np.uint32))
    seq = b"".join(s.encode() if isinstance(s, str) else bytes(s) for s in seqs)
    qual = np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.uint8) for q in quals]) if n else np.zeros(0, np.uint8))
    assert len(seq) == len(qual)
    pv = None if pair is None else np.ascontiguousarray(pair, dtype=np.uint32)
    ns = C.c_int64(0)
    check(lib.isx_count_read_segs(n, cig.ctypes.data, cig_off.ctypes.data, rs.ctypes.data, lo.ctypes.data, hi.ctypes.data, C.byref(ns)))
    cap = max(1, ns.value)
    g = np.empty(cap, np.uint32); ln = np.empty(cap, np.uint8); sp = np.zeros(cap, np.uint32)
    pl = _aligned((cap, _lib.PLANE_WORDS), np.uint64)
    check(lib.isx_pack_read_planes(n, rs.ctypes.data, lo.ctypes.data, hi.ctypes.data, cig.ctypes.data, cig_off.ctypes.data, seq,
                                   qual.ctypes.data, seq_off.ctypes.data, pv.ctypes.data if pv is not None else None, int(min_base_quality), cap,
                                   g.ctypes.data, ln.ctypes.data, sp.ctypes.data, pl.ctypes.data, C.byref(ns)))
    k = ns.value
    return PlaneBatch(g[:k], ln[:k], pl[:k], sp[:k] if pair is not None else None)


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


def _name_blob(names):
    """list of str -> (bytes blob, int64 offsets[n + 1])"""
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    offs = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        offs[1:] = np.cumsum([len(e) for e in enc])
    return b"".join(enc), offs


class BamFile:
    """Host BAM front end (isx_bam_*): BGZF/BAM decode + read-pair filter + htslib-1.9 pileup rules.

    scan() / filter() / expand_refs() are the three passes (see include/instrain_amd.h); expand() runs all of
    them over the whole file."""

    def __init__(self, path, threads=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.isx_bam_open(path.encode(), C.byref(h)))
        self.h = h
        self.info = None
        self._refs = None
        if threads:
            check(self.lib.isx_bam_set_threads(self.h, int(threads)))

    def set_threads(self, threads):
        """host threads of the handle's passes from now on (0 = automatic: 2 x the container's cpu quota)"""
        check(self.lib.isx_bam_set_threads(self.h, int(threads)))

    @staticmethod
    def _params(min_read_ani=0.95, min_mapq=-1, max_insert_relative=3, min_insert=50, min_base_quality=30,
                skip_mm=False, window_length=10000, pairing_filter="paired_only"):
        pf = _lib.PAIRING_FILTERS[pairing_filter] if isinstance(pairing_filter, str) else int(pairing_filter)
        return BamParams(float(min_read_ani), int(min_mapq), float(max_insert_relative), int(min_insert),
                         int(min_base_quality), 1 if skip_mm else 0, int(window_length), pf)

    def _info(self, info):
        self.info = {n: getattr(info, n) for n, _ in BamInfo._fields_ if n != "pad"}
        return self.info

    def scan(self, part=None):
        """pass 1: pair tables of every reference (get_paired_reads); part = (i, n): only share i of n of the file -- the
        handle then owns the references whose first read lies in that share (ref_counts() is zero for the others)"""
        info = BamInfo()
        if part is None:
            check(self.lib.isx_bam_scan(self.h, C.byref(info)))
        else:
            check(self.lib.isx_bam_scan_part(self.h, int(part[0]), int(part[1]), C.byref(info)))
        return self._info(info)

    def set_wanted_refs(self, refs=None):
        """the scaffolds that exist for the read filter (the reference loads only the scaffolds of the fasta,
        filter_reads.py:63-77): reference indices, None / empty = every reference of the file"""
        r = np.ascontiguousarray(refs if refs is not None else [], dtype=np.int32)
        check(self.lib.isx_bam_set_wanted_refs(self.h, r.ctypes.data if len(r) else None, len(r)))

    def insert_sizes(self):
        n = C.c_int64(0)
        check(self.lib.isx_bam_insert_sizes(self.h, None, 0, C.byref(n)))
        out = np.empty(max(1, n.value), dtype=np.int64)
        check(self.lib.isx_bam_insert_sizes(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def filter_insert_sizes(self):
        """inserts of the pairs that went through paired_read_filter in the last filter() (what the median is taken over)"""
        n = C.c_int64(0)
        check(self.lib.isx_bam_filter_insert_sizes(self.h, None, 0, C.byref(n)))
        out = np.empty(max(1, n.value), dtype=np.int64)
        check(self.lib.isx_bam_filter_insert_sizes(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def pair_keys(self):
        """-> (h1 u64 [n], h2 u64 [n], tid i32 [n], info i64 [n, 4] = nm, mapq, length, reads) per pair entry of this handle
        (isx_bam_pair_keys): the names as 128-bit keys, for the cross-scaffold look-ups of a file scanned in shares"""
        n = C.c_int64(0)
        check(self.lib.isx_bam_pair_keys(self.h, None, None, None, None, 0, C.byref(n)))
        k = n.value
        h1, h2 = np.empty(k, np.uint64), np.empty(k, np.uint64)
        tid, info = np.empty(k, np.int32), np.empty((k, 4), np.int64)
        if k:
            check(self.lib.isx_bam_pair_keys(self.h, h1.ctypes.data, h2.ctypes.data, tid.ctypes.data, info.ctypes.data, k, C.byref(n)))
        return h1, h2, tid, info

    def set_cross_names(self, entry, occurrences, info):
        entry = np.ascontiguousarray(entry, dtype=np.int64)
        occurrences = np.ascontiguousarray(occurrences, dtype=np.int64)
        info = np.ascontiguousarray(info, dtype=np.int64).reshape(-1, 4)
        assert len(entry) == len(occurrences) == len(info)
        check(self.lib.isx_bam_set_cross_names(self.h, len(entry), entry.ctypes.data, occurrences.ctypes.data, info.ctypes.data))

    def set_priority_reads(self, names):
        blob, offs = _name_blob(list(names))
        check(self.lib.isx_bam_set_priority_reads(self.h, len(offs) - 1, blob, offs.ctypes.data))

    def filter(self, median_insert=None, **kw):
        """paired_read_filter + filter_scaff2pair2info; kw as _params"""
        info = BamInfo()
        p = self._params(**kw)
        check(self.lib.isx_bam_filter(self.h, C.byref(p), float("nan") if median_insert is None else float(median_insert),
                                      C.byref(info)))
        return self._info(info)

    def set_r2m(self, ref, names, mm=None):
        """the controller's own R2M for reference index `ref`: exactly these read pairs, with these mm values"""
        blob, offs = _name_blob(list(names))
        mmv = None if mm is None else np.ascontiguousarray(mm, dtype=np.int32)
        check(self.lib.isx_bam_set_r2m(self.h, int(ref), len(offs) - 1, blob, offs.ctypes.data,
                                       mmv.ctypes.data if mmv is not None else None))

    def r2m(self, ref):
        """the filter's decision for reference index `ref` as the reference stores it: {pair name: mm}"""
        n, nb = C.c_int64(0), C.c_int64(0)
        check(self.lib.isx_bam_r2m(self.h, int(ref), C.byref(n), C.byref(nb), None, None, None))
        names = C.create_string_buffer(max(1, nb.value))
        offs = np.zeros(n.value + 1, dtype=np.int64)
        mm = np.zeros(max(1, n.value), dtype=np.int32)
        check(self.lib.isx_bam_r2m(self.h, int(ref), C.byref(n), C.byref(nb), names, offs.ctypes.data, mm.ctypes.data))
        raw = names.raw
        return {raw[offs[i]:offs[i + 1]].decode(): int(mm[i]) for i in range(n.value)}

    def mm_levels(self):
        """the distinct mm values of the kept read pairs, ascending (isx_bam_mm_levels)"""
        n = C.c_int32(0)
        check(self.lib.isx_bam_mm_levels(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        check(self.lib.isx_bam_mm_levels(self.h, out.ctypes.data, int(n.value), C.byref(n)))
        return out[:n.value]

    def set_mm_levels(self, levels):
        """pairs travel with the rank of their mm in `levels` (ascending values; isx_bam_set_mm_levels); empty: with the mm itself"""
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        check(self.lib.isx_bam_set_mm_levels(self.h, lv.ctypes.data if len(lv) else None, int(len(lv))))

    def set_mm_cap(self, cap):
        """pairs with more than `cap` mismatches are piled up at level `cap` (isx_bam_set_mm_cap)"""
        check(self.lib.isx_bam_set_mm_cap(self.h, int(cap)))

    def batch_pair_names(self):
        """names of the read pairs of the batch prepared last (expand_refs / segment_refs / Pipe.submit_bam), indexed by the dense
        pair id the device's tables carry (isx_bam_batch_pair_names); needs the names (no drop_names() before)"""
        n, nb = C.c_int64(0), C.c_int64(0)
        check(self.lib.isx_bam_batch_pair_names(self.h, C.byref(n), C.byref(nb), None, None))
        names = C.create_string_buffer(max(1, nb.value))
        offs = np.zeros(n.value + 1, dtype=np.int64)
        check(self.lib.isx_bam_batch_pair_names(self.h, C.byref(n), C.byref(nb), names, offs.ctypes.data))
        raw = names.raw
        return [raw[offs[i]:offs[i + 1]].decode() for i in range(n.value)]

    def drop_names(self):
        check(self.lib.isx_bam_drop_names(self.h))

    def ref_counts(self):
        """-> (records per reference, filtered pairs per reference)"""
        n = len(self.refs())
        reads, pairs = np.zeros(n, np.int64), np.zeros(n, np.int64)
        check(self.lib.isx_bam_ref_counts(self.h, reads.ctypes.data, pairs.ctypes.data))
        return reads, pairs

    def _results(self, info, copy):
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

    def expand_refs(self, refs, copy=True, **kw):
        """pass 2 for the reference indices `refs` (laid end to end in that order): -> (obs, pair, split_bounds,
        split_ref).  copy=False: obs / pair are views of the handle's arrays, valid until the next expand."""
        refs = np.ascontiguousarray(refs, dtype=np.int32)
        info = BamInfo()
        p = self._params(**kw)
        check(self.lib.isx_bam_expand_refs(self.h, C.byref(p), refs.ctypes.data, len(refs), C.byref(info)))
        self._info(info)
        return self._results(info, copy)

    def segment_refs(self, refs, **kw):
        """pass 2 for the reference indices `refs` as READ SEGMENTS (what a read-level pipe is handed): -> (SegBatch,
        split_bounds, split_ref); info["n_obs"] = the columns the segments cover"""
        refs = np.ascontiguousarray(refs, dtype=np.int32)
        info = BamInfo()
        p = self._params(**kw)
        n = C.c_int64(0)
        check(self.lib.isx_bam_segment_refs(self.h, C.byref(p), refs.ctypes.data, len(refs), C.byref(info), C.byref(n)))
        self._info(info)
        k = n.value
        g, ln, mm, pr = np.empty(k, np.uint32), np.empty(k, np.uint8), np.empty(k, np.uint8), np.empty(k, np.uint32)
        bs = np.empty((k, _lib.SEG_WORDS), np.uint32)
        bounds = np.empty(info.n_splits + 1, dtype=np.int64)
        sref = np.empty(info.n_splits, dtype=np.int32)
        check(self.lib.isx_bam_copy_segs(self.h, g.ctypes.data, ln.ctypes.data, mm.ctypes.data, pr.ctypes.data, bs.ctypes.data,
                                         bounds.ctypes.data, sref.ctypes.data))
        self._n_segs = k
        return SegBatch(g, ln, bs, mm, pr), bounds, sref

    def read_planes(self):
        """the segments of the last segment_refs() as bit planes [n_seg, 8] uint64 (isx_bam_copy_read_planes): what submit_bam hands a
        one-mm-bin pipe's stager"""
        pl = _aligned((self._n_segs, _lib.PLANE_WORDS), np.uint64)
        check(self.lib.isx_bam_copy_read_planes(self.h, pl.ctypes.data if self._n_segs else None))
        return pl

    def expand_region(self, ref, start, stop, copy=True, **kw):
        """pass 2 for the columns [start, stop) of reference index `ref` only (the re-pileup of SNV pooling)"""
        info = BamInfo()
        p = self._params(**kw)
        check(self.lib.isx_bam_expand_region(self.h, C.byref(p), int(ref), int(start), int(stop), C.byref(info)))
        self._info(info)
        return self._results(info, copy)

    def expand(self, copy=True, **kw):
        """scan + filter + expansion of every reference -> (obs, pair, split_bounds, split_ref)"""
        info = BamInfo()
        p = self._params(**kw)
        check(self.lib.isx_bam_expand(self.h, C.byref(p), C.byref(info)))
        self._info(info)
        return self._results(info, copy)

    def refs(self):
        if self._refs is None:
            out = []
            i = 0
            while True:
                name, ln, off = C.c_char_p(), C.c_int64(), C.c_int64()
                if self.lib.isx_bam_ref(self.h, i, C.byref(name), C.byref(ln), C.byref(off)) != 0:
                    break
                out.append((name.value.decode(), ln.value, off.value))
                i += 1
            self._refs = out
        return self._refs

    def close(self, wait=False):
        """wait: the handle's memory is back with the system when this returns (close_later: the caller's helper thread is the
        place for that); otherwise a large handle is freed on a thread of the library's"""
        h, self.h = self.h, None
        if h:
            (self.lib.isx_bam_close_wait if wait else self.lib.isx_bam_close)(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bgzf_index(image):
    """isx_bgzf_index: the blocks of a BGZF image (bytes / uint8 array) -> (blocks [n] of _lib.BGZF_BLOCK_DT, inflated bytes)"""
    lib = _lib.load()
    img = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else np.ascontiguousarray(image, dtype=np.uint8)
    n, tot = C.c_int64(0), C.c_int64(0)
    check(lib.isx_bgzf_index(img.ctypes.data if len(img) else None, len(img), 0, None, C.byref(n), C.byref(tot)) if len(img) else 0)
    blocks = np.zeros(n.value, dtype=_lib.BGZF_BLOCK_DT)
    if n.value:
        check(lib.isx_bgzf_index(img.ctypes.data, len(img), n.value, blocks.ctypes.data, C.byref(n), C.byref(tot)))
    return blocks, int(tot.value)


def bgzf_inflate(image, blocks=None, ctx=None, fast=False):
    """The inflated bytes of a BGZF image's blocks (all of them, or the given rows of bgzf_index re-based to their own output) --
    on the device of `ctx` (isx_bgzf_inflate_device; returns (bytes array, kernel ms)) or, ctx None, by the same decoder on the host;
    fast=True: the BAM front end's table-driven host decoder instead (isx_bgzf_inflate_fast)"""
    lib = _lib.load()
    img = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else np.ascontiguousarray(image, dtype=np.uint8)
    if blocks is None:
        blocks, _ = bgzf_index(img)
    blocks = np.ascontiguousarray(blocks, dtype=_lib.BGZF_BLOCK_DT).copy()
    blocks["out_off"] = np.cumsum(blocks["out_len"], dtype=np.int64) - blocks["out_len"]
    total = int(blocks["out_len"].sum())
    out = np.empty(total, dtype=np.uint8)
    if ctx is None:
        fn = lib.isx_bgzf_inflate_fast if fast else lib.isx_bgzf_inflate_host
        check(fn(img.ctypes.data, len(img), blocks.ctypes.data, len(blocks), out.ctypes.data, total))
        return out, None
    ms = C.c_float(0)
    check(lib.isx_bgzf_inflate_device(ctx.h, img.ctypes.data, len(img), blocks.ctypes.data, len(blocks), out.ctypes.data, total, C.byref(ms)))
    return out, float(ms.value)
