"""danbing-tk_amd — MI355X-native `danbing-tk align` hot path.

Import with importlib.import_module("danbing-tk_amd") (the hyphen follows the
reference's name).  `abi` mirrors include/dbtk.h; `Dbtk` binds the C-ABI of
libdbtk_hip.so (hand-written HIP for gfx950, built by csrc/Makefile) and raises
if that library is missing — there is no CPU execution path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdbtk_hip.so")

u64p, u32p, u8p = abi.u64p, abi.u32p, abi.u8p


class DbtkError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"dbtk status {status}: {msg}")
        self.status = status


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def bind_common(L):
    """Prototypes of the host-side entry points (shared with the test emulator)."""
    L.dbtk_last_error.restype = C.c_char_p
    L.dbtk_abi_version.restype = C.c_uint32
    L.dbtk_rpgg_load.restype = C.c_int
    L.dbtk_rpgg_load.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.dbtk_rpgg_load_tr.restype = C.c_int
    L.dbtk_rpgg_load_tr.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.dbtk_rpgg_from_arrays.restype = C.c_int
    L.dbtk_rpgg_from_arrays.argtypes = [C.POINTER(abi.RpggArrays), C.POINTER(C.c_void_p)]
    L.dbtk_rpgg_free.argtypes = [C.c_void_p]
    L.dbtk_rpgg_serialize.restype = C.c_int
    L.dbtk_rpgg_serialize.argtypes = [C.c_char_p]
    for f in ("dbtk_rpgg_nloci", "dbtk_rpgg_ntrkmers", "dbtk_rpgg_nkeys"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.dbtk_rpgg_view.restype = C.c_int
    L.dbtk_rpgg_view.argtypes = [C.c_void_p, C.POINTER(abi.RpggArrays)]
    L.dbtk_rpgg_output_order.restype = C.c_int
    L.dbtk_rpgg_output_order.argtypes = [C.c_void_p, u64p]
    L.dbtk_params_default.argtypes = [C.POINTER(abi.Params)]
    L.dbtk_write_outputs.restype = C.c_int
    L.dbtk_write_outputs.argtypes = [C.c_void_p, u64p, u64p, u32p, C.c_char_p, C.c_int]
    L.dbtk_aln_format.restype = C.c_size_t
    L.dbtk_aln_format.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]


class Rpgg:
    """Immutable RPGG handle (dbtk_rpgg_t)."""

    def __init__(self, lib, handle):
        self._lib, self.h = lib, handle
        L = lib.L
        self.nloci = L.dbtk_rpgg_nloci(handle)
        self.ntrkmers = L.dbtk_rpgg_ntrkmers(handle)
        self.nkeys = L.dbtk_rpgg_nkeys(handle)

    def output_order(self):
        v = abi.RpggArrays()
        self._lib._chk(self._lib.L.dbtk_rpgg_view(self.h, C.byref(v)))
        n = int(sum(v.tr_cnt[i] for i in range(self.nloci)))
        out = np.zeros(n, np.uint64)
        self._lib._chk(self._lib.L.dbtk_rpgg_output_order(self.h, _ptr(out, u64p)))
        return out

    def set_index_cache(self, path, mode=1):
        """The sidecar of the GPU-layout index images: 0 none, 1 load when present, 2 load or write (dbtk_rpgg_set_index_cache)."""
        self._lib._chk(self._lib.L.dbtk_rpgg_set_index_cache(self.h, path.encode() if path else None, mode))

    def view(self):
        v = abi.RpggArrays()
        self._lib._chk(self._lib.L.dbtk_rpgg_view(self.h, C.byref(v)))
        return v

    def write_outputs(self, counts, kmc, nmapread, out_prefix, with_names=False):
        self._lib._chk(self._lib.L.dbtk_write_outputs(self.h, _ptr(counts, u64p), _ptr(kmc, u64p), _ptr(nmapread, u32p),
                                                      out_prefix.encode(), int(with_names)))

    def close(self):
        if self.h:
            self._lib.L.dbtk_rpgg_free(self.h)
            self.h = None


class _HostSide:
    def _chk(self, st):
        if st != abi.OK:
            raise DbtkError(st, self.L.dbtk_last_error().decode())

    def load(self, prefix, k=21, qc_file=None, bait_file=None, flags=0, tr_file=None) -> Rpgg:
        """tr_file: the TR k-mer file when it is not PREF.tr.kmers (`-t N`: PREF.tr.trimN.kmers)."""
        h = C.c_void_p()
        self._chk(self.L.dbtk_rpgg_load_tr(prefix.encode(), tr_file.encode() if tr_file else None, k, qc_file.encode() if qc_file else None,
                                           bait_file.encode() if bait_file else None, flags, C.byref(h)))
        return Rpgg(self, h)

    def serialize(self, prefix):
        """`ktools serialize PREF`: text k-mer files -> PREF.kmers.dbi / .fl.kdb / .tre.kdb."""
        self._chk(self.L.dbtk_rpgg_serialize(prefix.encode()))

    def from_arrays(self, k, keys, vals, vv, fl_cnt, fl_ks, tr_cnt, tr_ks, tre_cnt=None, tre_ks=None, qc=None, gr_cnt=None, gr_ks=None,
                    gr_ms=None) -> Rpgg:
        keep = [np.ascontiguousarray(x, t) if x is not None else None for x, t in
                ((keys, np.uint64), (vals, np.uint32), (vv, np.uint32), (fl_cnt, np.uint64), (fl_ks, np.uint64),
                 (tre_cnt, np.uint64), (tre_ks, np.uint64), (tr_cnt, np.uint64), (tr_ks, np.uint64), (qc, np.uint8),
                 (gr_cnt, np.uint64), (gr_ks, np.uint64), (gr_ms, np.uint8))]
        a = abi.RpggArrays(ksize=k, nloci=len(keep[7]), nkeys=len(keep[0]), keys=_ptr(keep[0], u64p), vals=_ptr(keep[1], u32p),
                           nvv=len(keep[2]), vv=_ptr(keep[2], u32p), fl_cnt=_ptr(keep[3], u64p), fl_ks=_ptr(keep[4], u64p),
                           tre_cnt=_ptr(keep[5], u64p), tre_ks=_ptr(keep[6], u64p), tr_cnt=_ptr(keep[7], u64p),
                           tr_ks=_ptr(keep[8], u64p), qc=_ptr(keep[9], u8p), gr_cnt=_ptr(keep[10], u64p),
                           gr_ks=_ptr(keep[11], u64p), gr_ms=_ptr(keep[12], u8p))
        h = C.c_void_p()
        self._chk(self.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
        return Rpgg(self, h)


class Context:
    """Per-GPU context (dbtk_ctx_t)."""

    def __init__(self, lib, rpgg: Rpgg, params: abi.Params, device=0):
        self._lib, self.rpgg, self.params = lib, rpgg, params
        h = C.c_void_p()
        lib._chk(lib.L.dbtk_ctx_create(rpgg.h, C.byref(params), device, C.byref(h)))
        self.h = h

    def align(self, seq, off, qual=None, rec_cap=None):
        """One batch of pairs from host buffers.  Returns the batch's records
        (ctypes array, pair order) — counts accumulate inside the context."""
        L = self._lib.L
        off = np.ascontiguousarray(off, np.uint64)
        seq = np.ascontiguousarray(seq, np.uint8)
        npairs = (len(off) - 1) // 2
        p = self.params
        want = bool(p.trace or p.okam or p.extract)
        cap = npairs if rec_cap is None else rec_cap
        recs = (abi.PairRec * max(cap, 1))() if want and cap else None
        nrec = C.c_uint64(0)
        seqp = _ptr(seq if seq.size else np.zeros(1, np.uint8), u8p)
        self._lib._chk(L.dbtk_align_batch(self.h, seqp, _ptr(off, u64p), _ptr(qual, u8p), npairs, recs, cap if recs else 0,
                                          C.byref(nrec)))
        return recs, int(nrec.value)

    def thread(self, seq, off, loci):
        """dbtk_thread_batch: read r walked through graphDB[loci[r]]; returns a ctypes array of abi.ThreadRec."""
        off = np.ascontiguousarray(off, np.uint64)
        seq = np.ascontiguousarray(seq, np.uint8)
        loci = np.ascontiguousarray(loci, np.uint32)
        n = len(off) - 1
        recs = (abi.ThreadRec * max(n, 1))()
        self._lib._chk(self._lib.L.dbtk_thread_batch(self.h, _ptr(seq if seq.size else np.zeros(1, np.uint8), u8p), _ptr(off, u64p),
                                                     _ptr(loci, u32p), n, recs))
        return recs

    def walk_results(self, cap, with_recs=False):
        """dbtk_ctx_walk_results of the last align(): (WalkRes array, ThreadRec array or None, n)."""
        res = (abi.WalkRes * max(cap, 1))()
        trecs = (abi.ThreadRec * max(2 * cap, 1))() if with_recs else None
        n = C.c_uint64(0)
        self._lib._chk(self._lib.L.dbtk_ctx_walk_results(self.h, res, trecs, cap, C.byref(n)))
        return res, trecs, int(n.value)

    def aln_records(self):
        """dbtk_ctx_aln_records of the last align(): list of (AlnHdr copy, "cigar2\\tannot2\\tcigar1\\tannot1")."""
        L = self._lib.L
        n, st, cap = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        rc = L.dbtk_ctx_aln_records(self.h, None, 0, C.byref(n), C.byref(st), C.byref(cap))
        if rc not in (abi.OK, abi.ERR_OVERFLOW):
            self._lib._chk(rc)
        if not n.value:
            return []
        buf = (C.c_uint8 * (n.value * st.value))()
        self._lib._chk(L.dbtk_ctx_aln_records(self.h, buf, len(buf), C.byref(n), C.byref(st), C.byref(cap)))
        out = []
        txt = C.create_string_buffer(8192)
        for i in range(n.value):
            rec = C.byref(buf, i * st.value)
            hdr = abi.AlnHdr.from_buffer_copy(buf, i * st.value)
            L.dbtk_aln_format(rec, cap.value, txt, 8192)
            out.append((hdr, txt.value.decode()))
        return out

    def aln_text(self, npairs):
        """dbtk_ctx_aln_text of the last align() (params.aln | abi.ALN_TEXT): list of (pair, dst, "cigar2\\tannot2\\tcigar1\\tannot1")."""
        L = self._lib.L
        L.dbtk_ctx_aln_text.restype = C.c_int
        L.dbtk_ctx_aln_text.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        idx = np.zeros(max(npairs, 1), np.uint32)
        used = C.c_uint64(0)
        rc = L.dbtk_ctx_aln_text(self.h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), npairs, None, 0, C.byref(used))
        if rc not in (abi.OK, abi.ERR_OVERFLOW):
            self._lib._chk(rc)
        arena = np.zeros(max(int(used.value), 4), np.uint8)
        self._lib._chk(L.dbtk_ctx_aln_text(self.h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), npairs, arena.ctypes.data_as(C.c_void_p), len(arena), C.byref(used)))
        out = []
        for p_ in range(npairs):
            o = int(idx[p_])
            if o == 0xFFFFFFFF:
                continue
            dst, ln = int(arena[o:o + 4].view(np.uint32)[0]), int(arena[o + 4:o + 8].view(np.uint32)[0])
            out.append((p_, dst, arena[o + 8:o + 8 + ln].tobytes().decode()))
        return out

    def align_device(self, d_seq_ptr, d_off_ptr, npairs, max_read_len):
        self._lib._chk(self._lib.L.dbtk_align_batch_device(self.h, C.c_void_p(d_seq_ptr), C.c_void_p(d_off_ptr), npairs,
                                                           max_read_len))

    def write_bait_hits(self, out_prefix):
        self._lib._chk(self._lib.L.dbtk_ctx_write_bait_hits(self.h, out_prefix.encode()))

    def write_bubbles(self, out_prefix):
        self._lib._chk(self._lib.L.dbtk_ctx_write_bubbles(self.h, out_prefix.encode()))

    def synchronize(self):
        self._lib._chk(self._lib.L.dbtk_ctx_synchronize(self.h))

    def counts(self):
        g = self.rpgg
        counts = np.zeros(g.ntrkmers, np.uint64)
        kmc = np.zeros(g.nloci, np.uint64)
        nmap = np.zeros(g.nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        self._lib._chk(self._lib.L.dbtk_ctx_counts(self.h, _ptr(counts, u64p), _ptr(kmc, u64p), _ptr(nmap, u32p), _ptr(ctr, u64p)))
        return dict(counts=counts, kmc=kmc, nmapread=nmap, counters=ctr)

    def counters(self):
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        self._lib._chk(self._lib.L.dbtk_ctx_counts(self.h, None, None, None, _ptr(ctr, u64p)))
        return ctr

    def accum_buffer(self):
        base = C.c_void_p()
        n = C.c_uint64()
        self._lib._chk(self._lib.L.dbtk_ctx_accum_buffer(self.h, C.byref(base), C.byref(n)))
        return base.value, int(n.value)

    def reset(self):
        self._lib._chk(self._lib.L.dbtk_ctx_reset(self.h))

    def kernel_times(self):
        """{kernel: (total_ms, launches)} since timers_reset()."""
        names = (C.c_char_p * 8)()
        ms = (C.c_double * 8)()
        cnt = (C.c_uint64 * 8)()
        n = self._lib.L.dbtk_ctx_kernel_times(self.h, names, ms, cnt, 8)
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def table_bytes(self):
        """{table: HBM bytes} of the context's RPGG tables ("index_images:from_cache": 1 when the images came from the sidecar)."""
        names = (C.c_char_p * 24)()
        b = (C.c_uint64 * 24)()
        n = self._lib.L.dbtk_ctx_table_bytes(self.h, names, b, 24)
        return {names[i].decode(): int(b[i]) for i in range(n)}

    def path_stats(self):
        """Which kernels took how many pairs since creation / reset (dbtk.h: DBTK_PS_*): a dict."""
        v = (C.c_uint64 * 24)()
        n = self._lib.L.dbtk_ctx_path_stats(self.h, v, 24)
        v = [int(x) for x in v[:n]] + [0] * (24 - n)
        return {"probe_items": v[0:3], "probe_pairs": v[3:6], "probe_rest": v[6], "walk_items": v[7:10], "walk_pairs": v[10:13],
                "walk_rest": v[13], "fused_done": v[14], "fused_redone": v[15], "fused_cls": v[16], "fused_inc": v[17], "walk_locus_ec": v[18], "fused_shared": v[19], "lean_done": v[20], "vote_vv": v[21], "pair_vv": v[22]}

    def timers_reset(self):
        self._lib.L.dbtk_ctx_timers_reset(self.h)

    def timers_enable(self, on=True):
        self._lib.L.dbtk_ctx_timers_enable(self.h, int(on))

    def close(self):
        if self.h:
            self._lib.L.dbtk_ctx_free(self.h)
            self.h = None


class Dbtk(_HostSide):
    """The product: C-ABI of libdbtk_hip.so.  Fails loudly when the HIP
    extension has not been built — nothing falls back to the CPU."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: build it with `make -C danbing-tk_amd/csrc` "
                                    "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = self.L = C.CDLL(path)
        bind_common(L)
        L.dbtk_ctx_create.restype = C.c_int
        L.dbtk_ctx_create.argtypes = [C.c_void_p, C.POINTER(abi.Params), C.c_int, C.POINTER(C.c_void_p)]
        L.dbtk_ctx_free.argtypes = [C.c_void_p]
        L.dbtk_align_batch.restype = C.c_int
        L.dbtk_align_batch.argtypes = [C.c_void_p, u8p, u64p, u8p, C.c_uint64, C.POINTER(abi.PairRec), C.c_uint64, u64p]
        L.dbtk_thread_batch.restype = C.c_int
        L.dbtk_thread_batch.argtypes = [C.c_void_p, u8p, u64p, u32p, C.c_uint64, C.POINTER(abi.ThreadRec)]
        L.dbtk_ctx_walk_results.restype = C.c_int
        L.dbtk_ctx_walk_results.argtypes = [C.c_void_p, C.POINTER(abi.WalkRes), C.POINTER(abi.ThreadRec), C.c_uint64, u64p]
        L.dbtk_ctx_aln_records.restype = C.c_int
        L.dbtk_ctx_aln_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u64p, u32p, u32p]
        L.dbtk_align_batch_device.restype = C.c_int
        L.dbtk_align_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        L.dbtk_ctx_synchronize.restype = C.c_int
        L.dbtk_ctx_synchronize.argtypes = [C.c_void_p]
        L.dbtk_ctx_counts.restype = C.c_int
        L.dbtk_ctx_counts.argtypes = [C.c_void_p, u64p, u64p, u32p, u64p]
        L.dbtk_ctx_accum_buffer.restype = C.c_int
        L.dbtk_ctx_accum_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), u64p]
        L.dbtk_ctx_reset.restype = C.c_int
        L.dbtk_ctx_reset.argtypes = [C.c_void_p]
        L.dbtk_ctx_kernel_times.restype = C.c_int
        L.dbtk_ctx_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), u64p, C.c_int]
        L.dbtk_ctx_path_stats.restype = C.c_int
        L.dbtk_ctx_path_stats.argtypes = [C.c_void_p, u64p, C.c_int]
        L.dbtk_ctx_table_bytes.restype = C.c_int
        L.dbtk_ctx_table_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), u64p, C.c_int]
        L.dbtk_ctx_timers_reset.argtypes = [C.c_void_p]
        L.dbtk_ctx_timers_enable.argtypes = [C.c_void_p, C.c_int]
        L.dbtk_ctx_write_bubbles.restype = C.c_int
        L.dbtk_ctx_write_bubbles.argtypes = [C.c_void_p, C.c_char_p]
        L.dbtk_ctx_merge_bubbles.restype = C.c_int
        L.dbtk_ctx_merge_bubbles.argtypes = [C.c_void_p, C.c_void_p]
        L.dbtk_ctx_write_bait_hits.restype = C.c_int
        L.dbtk_ctx_write_bait_hits.argtypes = [C.c_void_p, C.c_char_p]
        L.dbtk_ctx_merge_bait_hits.restype = C.c_int
        L.dbtk_ctx_merge_bait_hits.argtypes = [C.c_void_p, C.c_void_p]
        L.dbtk_allreduce.restype = C.c_int
        L.dbtk_allreduce.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        if L.dbtk_abi_version() != abi.ABI_VERSION:
            raise RuntimeError("libdbtk_hip.so ABI version mismatch")

    def context(self, rpgg: Rpgg, params: abi.Params, device=0) -> Context:
        return Context(self, rpgg, params, device)

    def reserve_host(self, device, chunk_bytes, nslots, lines_bytes=0):
        """dbtk_ingest_reserve_host: pin an ingest's host buffers ahead of its creation (ABI v8)."""
        self.L.dbtk_ingest_reserve_host.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint64]
        self._chk(self.L.dbtk_ingest_reserve_host(int(device), int(chunk_bytes), int(nslots), int(lines_bytes)))

    def allreduce(self, ctxs):
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        self._chk(self.L.dbtk_allreduce(arr, len(ctxs)))


class Ingest:
    """dbtk_ingest_*: raw FASTA / FASTQ bytes in, parsed and paired on the device (include/dbtk.h)."""

    def __init__(self, ctx, fastq, min_read_size, chunk_bytes, nslots=3, with_spans=True):
        self.ctx, self.chunk, self.nslots, self.n = ctx, int(chunk_bytes), int(nslots), 0
        L = self.L = ctx._lib.L
        L.dbtk_ingest_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.dbtk_ingest_free.argtypes = [C.c_void_p]
        L.dbtk_ingest_free.restype = None
        L.dbtk_ingest_chunk_buffer.argtypes = [C.c_void_p, C.c_uint32]
        L.dbtk_ingest_chunk_buffer.restype = C.c_void_p
        L.dbtk_ingest_block.argtypes = [C.c_void_p, C.c_uint32]
        L.dbtk_ingest_block.restype = C.c_void_p
        L.dbtk_ingest_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_int]
        L.dbtk_ingest_wait.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.IngestInfo)]
        L.dbtk_ingest_align.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(abi.PairRec), C.c_uint64, u64p]
        L.dbtk_ingest_spans.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.IngestSpan), C.c_uint64]
        L.dbtk_ingest_align_merged.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_int]
        self.h = C.c_void_p()
        ctx._lib._chk(L.dbtk_ingest_create(ctx.h, int(bool(fastq)), int(min_read_size), self.chunk, self.nslots, int(bool(with_spans)), C.byref(self.h)))

    def submit(self, data: bytes, last: bool):
        slot = self.n % self.nslots
        assert len(data) <= self.chunk
        C.memmove(self.L.dbtk_ingest_chunk_buffer(self.h, slot), data, len(data))
        self.ctx._lib._chk(self.L.dbtk_ingest_submit(self.h, slot, len(data), int(last)))
        self.n += 1
        return slot

    def wait(self, slot):
        info = abi.IngestInfo()
        self.ctx._lib._chk(self.L.dbtk_ingest_wait(self.h, slot, C.byref(info)))
        return info

    def align(self, slot, info, sync=True, ctx=None):
        p = self.ctx.params
        want = sync and bool(p.trace or p.okam or p.extract) and info.nkept
        recs = (abi.PairRec * info.nkept)() if want else None
        nrec = C.c_uint64(0)
        self.ctx._lib._chk(self.L.dbtk_ingest_align(self.h, slot, ctx.h if ctx else None, int(sync), recs, info.nkept if want else 0, C.byref(nrec)))
        return recs, int(nrec.value)

    def align_merged(self, slot, min_pairs, flush=False, ctx=None):
        """dbtk_ingest_align_merged: the block appended to the context's merged batch (slot None: flush only)."""
        self.ctx._lib._chk(self.L.dbtk_ingest_align_merged(self.h, 0xFFFFFFFF if slot is None else slot, ctx.h if ctx else None, int(min_pairs), int(flush)))

    def spans(self, slot, info):
        """[(title, read 2q, read 2q + 1, qual 2q, qual 2q + 1)] of the block's kept pairs, as bytes."""
        sp = (abi.IngestSpan * max(info.nkept, 1))()
        self.ctx._lib._chk(self.L.dbtk_ingest_spans(self.h, slot, sp, info.nkept))
        base = self.L.dbtk_ingest_block(self.h, slot)
        g = lambda o, n: C.string_at(base + o, n)
        return [(g(s.title, s.title_len), g(s.seq[0], s.seq_len[0]), g(s.seq[1], s.seq_len[1]), g(s.qual[0], s.qual_len[0]), g(s.qual[1], s.qual_len[1]))
                for s in sp[:info.nkept]]

    def aln_lines(self, slot, gz=False, ctx=None):
        """(bytes, lines, text bytes): the -a / -ae lines of the block aligned last, as text or as gzip members."""
        L = self.L
        L.dbtk_ingest_aln_lines.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p, u64p, u64p]
        data, nb, nl, tb = C.c_void_p(), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.ctx._lib._chk(L.dbtk_ingest_aln_lines(self.h, slot, ctx.h if ctx else None, int(gz), C.byref(data), C.byref(nb), C.byref(nl), C.byref(tb)))
        return (C.string_at(data.value, nb.value) if nb.value else b""), int(nl.value), int(tb.value)

    def close(self):
        if self.h:
            self.L.dbtk_ingest_free(self.h)
            self.h = None


# every symbol include/dbtk.h declares (checked by the CPU test-suite)
EXPORTS = [
    "dbtk_rpgg_load", "dbtk_rpgg_load_tr", "dbtk_rpgg_uid", "dbtk_rpgg_from_arrays", "dbtk_rpgg_free", "dbtk_rpgg_nloci", "dbtk_rpgg_ntrkmers", "dbtk_rpgg_nkeys",
    "dbtk_rpgg_view", "dbtk_rpgg_output_order", "dbtk_params_default", "dbtk_device_warmup", "dbtk_ctx_create", "dbtk_ctx_free", "dbtk_align_batch",
    "dbtk_align_batch_device", "dbtk_ctx_synchronize", "dbtk_ctx_counts", "dbtk_ctx_accum_buffer", "dbtk_ctx_reset",
    "dbtk_allreduce", "dbtk_rpgg_set_index_cache", "dbtk_ctx_table_bytes", "dbtk_ctx_path_stats", "dbtk_ctx_kernel_times", "dbtk_ctx_timers_reset", "dbtk_ctx_timers_enable", "dbtk_ctx_aln_text", "dbtk_ctx_write_bubbles", "dbtk_ctx_merge_bubbles", "dbtk_ctx_write_bait_hits", "dbtk_ctx_merge_bait_hits", "dbtk_write_outputs", "dbtk_rpgg_serialize", "dbtk_last_error", "dbtk_abi_version",
    "dbtk_thread_batch", "dbtk_ctx_walk_results", "dbtk_ctx_aln_records", "dbtk_aln_format",
    "dbtk_ingest_reserve_host", "dbtk_ingest_create", "dbtk_ingest_free", "dbtk_ingest_chunk_buffer", "dbtk_ingest_block", "dbtk_ingest_submit", "dbtk_ingest_wait",
    "dbtk_ingest_align", "dbtk_ingest_align_merged", "dbtk_ingest_spans", "dbtk_ingest_aln_lines",
]

# include/dbtk_pred.h (the danbing-tk-pred step)
EXPORTS_PRED = [
    "dbtk_pred_create", "dbtk_pred_free", "dbtk_pred_create_from_file", "dbtk_pred_nk", "dbtk_pred_ntr", "dbtk_pred_load_samples",
    "dbtk_pred_correct", "dbtk_pred_matrix", "dbtk_pred_bias", "dbtk_pred_times",
]


class Pred:
    """include/dbtk_pred.h through ctypes: the cohort matrix in HBM, normalisation and bias correction on the GPU."""

    def __init__(self, lib, ns, nk_cum, nik_cum, iki, ikmc, nk=None, device=0):
        self._lib = lib
        L = lib.L
        L.dbtk_pred_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint64, u32p, u8p, C.POINTER(C.c_void_p)]
        L.dbtk_pred_free.argtypes = [C.c_void_p]
        L.dbtk_pred_free.restype = None
        L.dbtk_pred_load_samples.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, u64p, C.POINTER(C.c_float)]
        L.dbtk_pred_correct.argtypes = [C.c_void_p]
        L.dbtk_pred_matrix.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.dbtk_pred_bias.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.dbtk_pred_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self.nk_cum = np.ascontiguousarray(nk_cum, np.uint32)
        self.nik_cum = np.ascontiguousarray(nik_cum, np.uint32)
        self.iki = np.ascontiguousarray(iki, np.uint32)
        self.ikmc = np.ascontiguousarray(ikmc, np.uint8)
        self.ns, self.ntr = int(ns), len(self.nk_cum)
        self.nk = int(nk if nk is not None else (self.nk_cum[-1] if self.ntr else 0))
        self.h = C.c_void_p()
        lib._chk(L.dbtk_pred_create(device, self.ns, self.nk, self.ntr, _ptr(self.nk_cum, u32p), _ptr(self.nik_cum, u32p), len(self.iki),
                                    _ptr(self.iki, u32p), _ptr(self.ikmc, u8p), C.byref(self.h)))

    def load(self, first, counts, depths):
        counts = np.ascontiguousarray(counts, np.uint64)
        depths = np.ascontiguousarray(depths, np.float32)
        self._lib._chk(self._lib.L.dbtk_pred_load_samples(self.h, first, counts.shape[0], _ptr(counts, u64p), depths.ctypes.data_as(C.POINTER(C.c_float))))

    def correct(self):
        self._lib._chk(self._lib.L.dbtk_pred_correct(self.h))

    def matrix(self):
        out = np.empty((self.nk, self.ns), np.float32)
        self._lib._chk(self._lib.L.dbtk_pred_matrix(self.h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def bias(self):
        out = np.empty((self.ntr, self.ns), np.float32)
        self._lib._chk(self._lib.L.dbtk_pred_bias(self.h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def times(self):
        ms = (C.c_float * 3)()
        self._lib._chk(self._lib.L.dbtk_pred_times(self.h, ms))
        return list(ms)

    def close(self):
        if self.h:
            self._lib.L.dbtk_pred_free(self.h)
            self.h = None


class Synth:
    """Seeded release-scale workload generator (csrc/dbtk_synth.cpp): a flat
    RPGG + 150 bp read pairs, for bench.py and the scale tests."""

    def __init__(self, nloci=80000, k=21, flank=700, seed=20250808, nthreads=0, path=os.path.join(_HERE, "libdbtk_synth.so")):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: run `make -C danbing-tk_amd/csrc`")
        L = self.L = C.CDLL(path)
        L.dbtk_synth_create.restype = C.c_void_p
        L.dbtk_synth_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
        L.dbtk_synth_free.argtypes = [C.c_void_p]
        L.dbtk_synth_arrays.argtypes = [C.c_void_p, C.POINTER(abi.RpggArrays)]
        L.dbtk_synth_nbases.restype = C.c_uint64
        L.dbtk_synth_nbases.argtypes = [C.c_void_p]
        L.dbtk_synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_uint64, u8p, C.c_uint32]
        L.dbtk_synth_reads_loci.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_double, C.c_uint64, u8p, C.c_uint32]
        L.dbtk_synth_graph.argtypes = [C.c_void_p, C.c_uint32]
        L.dbtk_synth_write_files.restype = C.c_int
        L.dbtk_synth_write_files.argtypes = [C.c_void_p, C.c_char_p]
        L.dbtk_synth_write_fasta.restype = C.c_int
        L.dbtk_synth_write_fasta.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p]
        self.h = L.dbtk_synth_create(nloci, k, flank, seed, nthreads)
        self.k, self.nloci = k, nloci

    def arrays(self) -> abi.RpggArrays:
        a = abi.RpggArrays()
        self.L.dbtk_synth_arrays(self.h, C.byref(a))
        return a

    def graph(self, nthreads=0):
        """Build graphDB (both strands of every haplotype); arrays() then carries gr_cnt / gr_ks / gr_ms."""
        self.L.dbtk_synth_graph(self.h, nthreads)

    def write_files(self, prefix):
        """The RPGG as the HEAD files the reference binary loads (PREF.tr.kmers, .kmers.dbi, .fl.kdb, .tre.kdb)."""
        if self.L.dbtk_synth_write_files(self.h, prefix.encode()):
            raise IOError(f"could not write {prefix}.*")

    def write_fasta(self, seq, npairs, fn, rlen=150, first_pair=0):
        if self.L.dbtk_synth_write_fasta(_ptr(seq, u8p), npairs, rlen, first_pair, fn.encode()):
            raise IOError(f"could not write {fn}")

    def reads(self, npairs, rlen=150, hit_frac=1.0, seed=1, first_pair=0, out=None, nthreads=0):
        """(seq bytes, offsets) — read r is seq[r*rlen:(r+1)*rlen]."""
        if out is None:
            out = np.empty(npairs * 2 * rlen, np.uint8)
        self.L.dbtk_synth_reads(self.h, npairs, first_pair, rlen, float(hit_frac), seed, _ptr(out, u8p), nthreads)
        off = np.arange(2 * npairs + 1, dtype=np.uint64) * np.uint64(rlen)
        return out, off

    def reads_loci(self, npairs, loci, rlen=150, odd_frac=0.1, seed=1, first_pair=0, nthreads=0):
        """(seq bytes, offsets): pair p drawn from loci[p % len(loci)] — a DENSE slice (many pairs per locus: the regime of the
        locus-resident kernels); odd_frac of the pairs chimeric (mate 2 from any locus) or foreign (both from any locus)."""
        loci = np.ascontiguousarray(loci, np.uint32)
        out = np.empty(npairs * 2 * rlen, np.uint8)
        self.L.dbtk_synth_reads_loci(self.h, npairs, first_pair, rlen, _ptr(loci, C.POINTER(C.c_uint32)), len(loci), float(odd_frac), seed, _ptr(out, u8p), nthreads)
        off = np.arange(2 * npairs + 1, dtype=np.uint64) * np.uint64(rlen)
        return out, off

    def close(self):
        if self.h:
            self.L.dbtk_synth_free(self.h)
            self.h = None
