"""ctypes bindings used by tests/, bench.py and __graft_entry__.smoke():

  Oracle     oracle/liboracle.so                 the plain-C restatement (checker)
  RefHarness oracle/_ref/libdbtk_refharness.so   the real reference, function level

The product binding lives in the package (danbing-tk_amd/); struct layouts
come from there so that all three sides share include/dbtk.h's definitions.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class BubEvent(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("mate", C.c_uint32), ("pos", C.c_uint32), ("locus", C.c_uint32), ("edge", C.c_uint64)]


def events_array(ev, n):
    return np.frombuffer(ev, dtype=np.dtype([("pair", "<u4"), ("mate", "<u4"), ("pos", "<u4"), ("locus", "<u4"), ("edge", "<u8")]),
                         count=n).copy()


class WalkOut(C.Structure):
    _fields_ = [("res", C.POINTER(abi.WalkRes)), ("trecs", C.POINTER(abi.ThreadRec)), ("cap", C.c_uint64), ("n", C.c_uint64)]


def aln_line(O, res, trecs, i, title: str, s1: bytes, s2: bytes, nloci):
    """The -a / -ae record of walk result i (writeAlignments, AQ.cpp:1742-1759) from its two thread records."""
    buf = C.create_string_buffer(16384)
    O.L.orc_write_alignment(-1, res[i].dst, title.encode(), s1, len(s1), s2, len(s2), C.byref(trecs[2 * i]), C.byref(trecs[2 * i + 1]), buf,
                            16384)
    return buf.value.decode()


class Oracle:
    def __init__(self):
        path = os.environ.get("DBTK_ORACLE_LIB", os.path.join(ROOT, "oracle", "liboracle.so"))  # (make asan: the sanitized build)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C oracle oracle`")
        L = self.L = C.CDLL(path)
        L.orc_rpgg_load.restype = C.c_void_p
        L.orc_rpgg_load.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        L.orc_rpgg_from_arrays.restype = C.c_void_p
        L.orc_rpgg_from_arrays.argtypes = [C.POINTER(abi.RpggArrays)]
        L.orc_rpgg_free.argtypes = [C.c_void_p]
        L.orc_rpgg_nloci.restype = C.c_uint64
        L.orc_rpgg_nloci.argtypes = [C.c_void_p]
        L.orc_rpgg_ntrkmers.restype = C.c_uint64
        L.orc_rpgg_ntrkmers.argtypes = [C.c_void_p]
        L.orc_vote_vv.restype = C.c_uint64
        L.orc_vote_vv.argtypes = []
        L.orc_vote_vv_reset.restype = None
        L.orc_vote_vv_reset.argtypes = []
        L.orc_align.restype = C.c_int
        L.orc_align.argtypes = [C.c_void_p, C.POINTER(abi.Params), u8p, u64p, C.c_uint64, u64p, u64p, u32p, u64p,
                                C.POINTER(abi.PairRec)]
        L.orc_align_ex.restype = C.c_int
        L.orc_align_ex.argtypes = [C.c_void_p, C.POINTER(abi.Params), u8p, u64p, u8p, C.c_uint64, u64p, u64p, u32p, u64p,
                                   C.POINTER(abi.PairRec), C.POINTER(BubEvent), C.c_uint64, u64p]
        L.orc_rpgg_load_bait.restype = C.c_int
        L.orc_rpgg_load_bait.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_qstring2qmask.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_nurc.restype = C.c_uint64
        L.orc_nurc.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_read2kmers_edges.restype = C.c_uint64
        L.orc_read2kmers_edges.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, u64p, u64p]
        L.orc_sort_index.argtypes = [u64p, C.c_uint64, u64p]
        L.orc_umap_order.restype = C.c_int
        L.orc_umap_order.argtypes = [u64p, C.c_uint64, u64p]
        L.orc_rpgg_load_graph.restype = C.c_int
        L.orc_rpgg_load_graph.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_thread.restype = C.c_int
        L.orc_thread.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                 C.POINTER(abi.ThreadRec), u64p]
        L.orc_write_cigar.restype = C.c_size_t
        L.orc_write_cigar.argtypes = [C.POINTER(abi.ThreadRec), C.c_char_p, C.c_size_t]
        L.orc_write_annot.restype = C.c_size_t
        L.orc_write_annot.argtypes = [C.POINTER(abi.ThreadRec), C.c_char_p, C.c_size_t]
        L.orc_write_alignment.restype = C.c_size_t
        L.orc_write_alignment.argtypes = [C.c_int64, C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64,
                                          C.POINTER(abi.ThreadRec), C.POINTER(abi.ThreadRec), C.c_char_p, C.c_size_t]
        L.orc_align_walk.restype = C.c_int
        L.orc_align_walk.argtypes = [C.c_void_p, C.POINTER(abi.Params), u8p, u64p, u8p, C.c_uint64, u64p, u64p, u32p, u64p,
                                     C.POINTER(abi.PairRec), C.POINTER(BubEvent), C.c_uint64, u64p, C.POINTER(WalkOut)]

    def load(self, prefix, k, qc_file=None):
        h = self.L.orc_rpgg_load(prefix.encode(), k, qc_file.encode() if qc_file else None)
        if not h:
            raise IOError(f"oracle could not load {prefix}")
        return h

    def from_arrays(self, arrs: "abi.RpggArrays"):
        return self.L.orc_rpgg_from_arrays(C.byref(arrs))

    def load_graph(self, h, graph_file):
        if self.L.orc_rpgg_load_graph(h, graph_file.encode()):
            raise IOError(f"oracle could not load {graph_file}")

    def graph_dump(self, h, locus, cap=1 << 20):
        """graphDB[locus] as the oracle's loaders left it: (nodes ascending, masks)."""
        ks, ms = np.zeros(cap, np.uint64), np.zeros(cap, np.uint8)
        self.L.orc_rpgg_graph_dump.restype = C.c_uint64
        self.L.orc_rpgg_graph_dump.argtypes = [C.c_void_p, C.c_uint32, u64p, u8p, C.c_uint64]
        n = int(self.L.orc_rpgg_graph_dump(h, locus, _p(ks, u64p), _p(ms, u8p), cap))
        assert n <= cap
        return ks[:n].copy(), ms[:n].copy()

    def thread(self, h, locus, read: bytes, params, rec=None):
        """orc_thread: (ret, ThreadRec)."""
        rec = rec if rec is not None else abi.ThreadRec()
        r = self.L.orc_thread(h, locus, read, len(read), params.ksize, params.thread_cth, int(params.correction), params.maxncorrection,
                              C.byref(rec), None)
        return r, rec

    def cigar_annot(self, rec):
        b1, b2 = C.create_string_buffer(4096), C.create_string_buffer(4096)
        self.L.orc_write_cigar(C.byref(rec), b1, 4096)
        self.L.orc_write_annot(C.byref(rec), b2, 4096)
        return b1.value.decode(), b2.value.decode()

    def free(self, h):
        self.L.orc_rpgg_free(h)

    def align(self, h, params, seq, off, trace=True):
        npairs = (len(off) - 1) // 2
        nloci = self.L.orc_rpgg_nloci(h)
        ntr = self.L.orc_rpgg_ntrkmers(h)
        counts = np.zeros(ntr, np.uint64)
        kmc = np.zeros(nloci, np.uint64)
        nmap = np.zeros(nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * npairs)() if trace else None
        seq = np.ascontiguousarray(seq, np.uint8)
        if seq.size == 0:
            seq = np.zeros(1, np.uint8)
        self.L.orc_vote_vv_reset()
        rc = self.L.orc_align(h, C.byref(params), _p(seq, u8p), _p(off, u64p), npairs, _p(counts, u64p), _p(kmc, u64p),
                              _p(nmap, u32p), _p(ctr, u64p), recs)
        if rc:
            raise RuntimeError(f"orc_align -> {rc}")
        # vote_vv: the vv words the reference's votes read for this batch (it votes on every pair); the product's own figure, the path
        # statistic "vote_vv", can only be smaller
        return dict(counts_file=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs, vote_vv=int(self.L.orc_vote_vv()))

    def load_bait(self, h, bait_file):
        if self.L.orc_rpgg_load_bait(h, bait_file.encode()):
            raise IOError(f"oracle could not load {bait_file}")

    def align_walk(self, h, params, seq, off, trace=False, with_recs=True):
        """The hot loop with threading = 2: orc_align_walk.  Adds walk results (pair order) + their thread records."""
        npairs = (len(off) - 1) // 2
        nloci = self.L.orc_rpgg_nloci(h)
        ntr = self.L.orc_rpgg_ntrkmers(h)
        counts = np.zeros(ntr, np.uint64)
        kmc = np.zeros(nloci, np.uint64)
        nmap = np.zeros(nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * npairs)() if trace else None
        res = (abi.WalkRes * max(npairs, 1))()
        trecs = (abi.ThreadRec * max(2 * npairs, 1))() if with_recs else None
        wo = WalkOut(res, trecs, npairs, 0)
        nev = C.c_uint64(0)
        seq = np.ascontiguousarray(seq, np.uint8)
        self.L.orc_vote_vv_reset()
        rc = self.L.orc_align_walk(h, C.byref(params), _p(seq, u8p), _p(off, u64p), None, npairs, _p(counts, u64p), _p(kmc, u64p),
                                   _p(nmap, u32p), _p(ctr, u64p), recs, None, 0, C.byref(nev), C.byref(wo))
        if rc:
            raise RuntimeError(f"orc_align_walk -> {rc}")
        return dict(counts_file=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs, res=res, trecs=trecs, nres=int(wo.n),
                    vote_vv=int(self.L.orc_vote_vv()))

    def qmask(self, qual: bytes, qth, k):
        m = np.zeros(max(len(qual), 1), np.uint8)
        self.L.orc_qstring2qmask(qual, len(qual), qth, k, _p(m, u8p))
        return m[:max(len(qual) - k + 1, 0)]

    def align_ex(self, h, params, seq, off, qual=None, trace=True, evcap=1 << 20):
        npairs = (len(off) - 1) // 2
        nloci = self.L.orc_rpgg_nloci(h)
        ntr = self.L.orc_rpgg_ntrkmers(h)
        counts = np.zeros(ntr, np.uint64)
        kmc = np.zeros(nloci, np.uint64)
        nmap = np.zeros(nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * npairs)() if trace else None
        ev = (BubEvent * evcap)() if params.bubbles else None
        nev = C.c_uint64(0)
        seq = np.ascontiguousarray(seq, np.uint8)
        self.L.orc_vote_vv_reset()
        rc = self.L.orc_align_ex(h, C.byref(params), _p(seq, u8p), _p(off, u64p), _p(qual, u8p), npairs, _p(counts, u64p),
                                 _p(kmc, u64p), _p(nmap, u32p), _p(ctr, u64p), recs, ev, evcap if ev else 0, C.byref(nev))
        if rc:
            raise RuntimeError(f"orc_align_ex -> {rc}")
        assert nev.value <= evcap
        return dict(counts_file=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs,
                    events=events_array(ev, nev.value) if ev else None, vote_vv=int(self.L.orc_vote_vv()))

    def read2kmers_edges(self, read: bytes, k):
        n = max(len(read), 1)
        ks = np.zeros(n, np.uint64)
        es = np.zeros(n, np.uint64)
        nk = self.L.orc_read2kmers_edges(read, len(read), k, _p(ks, u64p), _p(es, u64p))
        return ks[:nk].copy(), es[:max(nk - 1, 0)].copy() if nk else es[:0].copy()

    def sort_index(self, data):
        data = np.ascontiguousarray(data, np.uint64)
        idx = np.zeros(len(data), np.uint64)
        self.L.orc_sort_index(_p(data, u64p), len(data), _p(idx, u64p))
        return idx

    def umap_order(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        order = np.zeros(len(keys), np.uint64)
        rc = self.L.orc_umap_order(_p(keys, u64p), len(keys), _p(order, u64p))
        if rc:
            raise RuntimeError("orc_umap_order: libstdc++ prime table not found")
        return order


class RefHarness:
    """The real reference's functions (needs oracle/_ref/libdbtk_refharness.so)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "_ref", "libdbtk_refharness.so")
        L = self.L = C.CDLL(path)
        L.ref_set_params.argtypes = [C.c_uint64] * 5
        L.ref_nurc.restype = C.c_uint64
        L.ref_nurc.argtypes = [C.c_uint64, C.c_uint64]
        L.ref_read2kmers_edges.restype = C.c_uint64
        L.ref_read2kmers_edges.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, u64p, u64p]
        L.ref_sort_index.argtypes = [u64p, C.c_uint64, u64p]
        L.ref_umap_order.argtypes = [u64p, C.c_uint64, u64p]
        L.ref_db_load.restype = C.c_void_p
        L.ref_db_load.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_db_free.argtypes = [C.c_void_p]
        L.ref_db_nloci.restype = C.c_uint64
        L.ref_db_nloci.argtypes = [C.c_void_p]
        L.ref_db_ntr.restype = C.c_uint64
        L.ref_db_ntr.argtypes = [C.c_void_p]
        L.ref_align.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, u64p, u64p, u32p,
                                u64p, C.POINTER(abi.PairRec)]
        L.ref_align_ex.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_uint32,
                                   C.c_int, C.c_int, u64p, u64p, u32p, u64p, C.POINTER(abi.PairRec), C.POINTER(BubEvent),
                                   C.c_uint64, u64p]
        L.ref_db_load_bait.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_qstring2qmask.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, u8p]
        L.ref_db_load_graph.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_set_thread_params.argtypes = [C.c_uint64, C.c_int]
        L.ref_thread.restype = C.c_int
        L.ref_thread.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(abi.ThreadRec),
                                 u64p, C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_int)]

    def set_params(self, p):
        self.L.ref_set_params(p.ksize, p.n_filter, p.nm_filter, p.max_nt, p.nm_tr)
        self.L.ref_set_thread_params(p.maxncorrection, 0)

    def align_v13(self, h, params, seq, off, titles, tc=False):
        """ref_align_v13: the hot loop with the reference's commented-out v1.3 threading call sites executed."""
        self.set_params(params)
        L = self.L
        L.ref_align_v13.restype = C.c_int64
        L.ref_align_v13.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                    C.c_int, C.c_int, u64p, u64p, C.POINTER(abi.WalkRes), C.POINTER(abi.ThreadRec), C.c_uint64, u64p,
                                    C.c_char_p, C.c_uint64]
        npairs = (len(off) - 1) // 2
        ntr = L.ref_db_ntr(h)
        counts = np.zeros(ntr, np.uint64)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        res = (abi.WalkRes * max(npairs, 1))()
        trecs = (abi.ThreadRec * max(2 * npairs, 1))()
        nres = C.c_uint64(0)
        cap = 64 + npairs * 2048
        text = C.create_string_buffer(cap)
        seq = np.ascontiguousarray(seq, np.uint8)
        rc = L.ref_align_v13(h, seq.tobytes(), _p(off, u64p), "\n".join(titles).encode(), npairs, params.cthreshold, int(params.qc),
                             params.thread_cth, int(params.correction), int(tc), int(params.aln != 0), int(params.aln == 2),
                             _p(counts, u64p), _p(ctr, u64p), res, trecs, npairs, C.byref(nres), text, cap)
        if rc < 0:
            raise RuntimeError("the reference asserted")
        return dict(counts_file=counts, counters=ctr, res=res, trecs=trecs, nres=int(nres.value), aln=text.raw[:rc].decode())

    def load_graph(self, h, graph_kmers_file):
        self.L.ref_db_load_graph(h, graph_kmers_file.encode())

    def graph_dump(self, h, locus, cap=1 << 20):
        """graphDB[locus] of the reference after its own readGraphKmers: (nodes ascending, masks)."""
        ks, ms = np.zeros(cap, np.uint64), np.zeros(cap, np.uint8)
        self.L.ref_graph_dump.restype = C.c_uint64
        self.L.ref_graph_dump.argtypes = [C.c_void_p, C.c_uint32, u64p, u8p, C.c_uint64]
        n = int(self.L.ref_graph_dump(h, locus, _p(ks, u64p), _p(ms, u8p), cap))
        assert n <= cap
        return ks[:n].copy(), ms[:n].copy()

    def thread(self, h, locus, read: bytes, params, tc=False, rec=None):
        """ref_thread: (ret, ThreadRec, cigar, annot, flagged by threadCheck)."""
        rec = rec if rec is not None else abi.ThreadRec()
        cig, ann, fl = C.create_string_buffer(4096), C.create_string_buffer(4096), C.c_int(0)
        r = self.L.ref_thread(h, locus, read, len(read), params.thread_cth, int(params.correction), int(tc), C.byref(rec), None, cig, ann,
                              4096, C.byref(fl))
        return r, rec, cig.value.decode(), ann.value.decode(), fl.value

    def read2kmers_edges(self, read: bytes, k):
        n = max(len(read), 1)
        ks = np.zeros(n, np.uint64)
        es = np.zeros(n, np.uint64)
        nk = self.L.ref_read2kmers_edges(read, len(read), k, _p(ks, u64p), _p(es, u64p))
        return ks[:nk].copy(), es[:max(nk - 1, 0)].copy() if nk else es[:0].copy()

    def sort_index(self, data):
        data = np.ascontiguousarray(data, np.uint64)
        idx = np.zeros(len(data), np.uint64)
        self.L.ref_sort_index(_p(data, u64p), len(data), _p(idx, u64p))
        return idx

    def umap_order(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        order = np.zeros(len(keys), np.uint64)
        self.L.ref_umap_order(_p(keys, u64p), len(keys), _p(order, u64p))
        return order

    def load(self, prefix, qc_file=None):
        return self.L.ref_db_load(prefix.encode(), qc_file.encode() if qc_file else None)

    def free(self, h):
        self.L.ref_db_free(h)

    def load_bait(self, h, prefix):
        self.L.ref_db_load_bait(h, prefix.encode())

    def qmask(self, qual: bytes, qth, k):
        m = np.zeros(max(len(qual), 1), np.uint8)
        self.L.ref_qstring2qmask(qual, len(qual), qth, k, _p(m, u8p))
        return m[:max(len(qual) - k + 1, 0)]

    def align_ex(self, h, params, seq, off, qual=None, evcap=1 << 20):
        self.set_params(params)
        npairs = (len(off) - 1) // 2
        nloci = self.L.ref_db_nloci(h)
        ntr = self.L.ref_db_ntr(h)
        counts = np.zeros(ntr, np.uint64)
        kmc = np.zeros(nloci, np.uint64)
        nmap = np.zeros(nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * npairs)()
        ev = (BubEvent * evcap)()
        nev = C.c_uint64(0)
        seq = np.ascontiguousarray(seq, np.uint8)
        self.L.ref_align_ex(h, seq.tobytes(), _p(off, u64p), qual.tobytes() if qual is not None else None, npairs,
                            params.cthreshold, int(params.okam), int(params.qc), params.qth, int(params.bait), int(params.bubbles),
                            _p(counts, u64p), _p(kmc, u64p), _p(nmap, u32p), _p(ctr, u64p), recs, ev, evcap, C.byref(nev))
        return dict(counts_file=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs, events=events_array(ev, nev.value))

    def align(self, h, params, seq, off, trace=True):
        self.set_params(params)
        npairs = (len(off) - 1) // 2
        nloci = self.L.ref_db_nloci(h)
        ntr = self.L.ref_db_ntr(h)
        counts = np.zeros(ntr, np.uint64)
        kmc = np.zeros(nloci, np.uint64)
        nmap = np.zeros(nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * npairs)() if trace else None
        seq = np.ascontiguousarray(seq, np.uint8)
        self.L.ref_align(h, seq.tobytes(), _p(off, u64p), npairs, params.cthreshold, int(params.okam), int(params.qc),
                         _p(counts, u64p), _p(kmc, u64p), _p(nmap, u32p), _p(ctr, u64p), recs)
        return dict(counts_file=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs)


def walk_res_equal(mine, theirs, n, nloci, every_mate):
    """dbtk_ctx_walk_results against the oracle's: pair and destLocus always; the return codes exactly when every mate's
    alignment was asked for (-a / -ae, trace) — otherwise a mate whose partner threads (cleanly, or after error correction) may be
    reported as DBTK_WALK_NOT_EVALUATED (-2, dbtk.h): then its partner's code is the oracle's, not 0, and the pair is kept.
    Returns how many were skipped, or -1 on a mismatch."""
    if every_mate:
        return 0 if bytes(mine)[:8 * n] == bytes(theirs)[:8 * n] else -1
    skipped = 0
    for i in range(n):
        a, b = mine[i], theirs[i]
        if (a.pair, a.dst) != (b.pair, b.dst):
            return -1
        for m, t, other in ((a.ret1, b.ret1, a.ret2), (a.ret2, b.ret2, a.ret1)):
            if m == -2:
                if other in (0, -2) or b.dst == nloci:
                    return -1
                skipped += 1
            elif m != t:
                return -1
    return skipped


def recs_equal(a, b, n):
    """Byte-compare two PairRec arrays; returns index of first difference or -1."""
    sz = C.sizeof(abi.PairRec)
    ba = np.frombuffer(a, dtype=np.uint8, count=sz * n).reshape(n, sz)
    bb = np.frombuffer(b, dtype=np.uint8, count=sz * n).reshape(n, sz)
    d = np.nonzero((ba != bb).any(axis=1))[0]
    return int(d[0]) if len(d) else -1


def rec_str(r):
    def m(x):
        return (f"si={x.si} ei={x.ei} si_={x.si_} ei_={x.ei_} nt={x.nt} bs={x.bs} ti={x.ti} "
                f"kf={x.kf} hf={x.hf} af={x.af} rm={x.rm} nk={x.nk}")
    return f"pair={r.pair} stage={r.stage} dst={r.dst} dst0={r.dst0} nm=({r.nm1},{r.nm2}) r1[{m(r.r1)}] r2[{m(r.r2)}]"


class Emu(pkg._HostSide):
    """TEST-ONLY: the kernel bodies of dbtk_kernels.h run on coroutine lanes
    (tests/emu/emu.cpp).  Lets the CPU suite check the device logic; it is not
    a backend of the product."""

    def __init__(self):
        path = os.environ.get("DBTK_EMU_LIB", os.path.join(ROOT, "tests", "emu", "libdbtk_emu.so"))
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C tests/emu`")
        L = self.L = C.CDLL(path)
        pkg.bind_common(L)
        L.emu_tables_create.restype = C.c_void_p
        L.emu_tables_create.argtypes = [C.c_void_p]
        L.emu_tables_free.argtypes = [C.c_void_p]
        L.emu_tables_consistent.restype = C.c_uint32
        L.emu_tables_consistent.argtypes = [C.c_void_p]
        L.emu_tables_set_consistent.argtypes = [C.c_void_p, C.c_uint32]
        L.emu_selftest_assign.restype = C.c_uint64
        L.emu_selftest_assign.argtypes = [C.c_uint64, C.c_uint64]
        L.emu_selftest_wavesort.restype = C.c_uint64
        L.emu_selftest_wavesort.argtypes = [C.c_uint64, C.c_uint64]
        L.emu_selftest_vote.restype = C.c_uint64
        L.emu_selftest_vote.argtypes = [C.c_uint64, C.c_uint64]
        L.emu_selftest_sort.restype = C.c_uint64
        L.emu_selftest_sort.argtypes = [C.c_uint64, C.c_uint64]
        L.emu_align_ex.restype = C.c_int
        L.emu_align_ex.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.Params), u8p, u64p, u8p, C.c_uint64, u64p, u64p, u32p, u64p,
                                   C.POINTER(abi.PairRec), C.c_uint64, u64p, C.c_uint32, C.c_uint32, C.POINTER(BubEvent), C.c_uint64,
                                   u64p]
        L.emu_align.restype = C.c_int
        L.emu_align.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.Params), u8p, u64p, C.c_uint64, u64p, u64p, u32p, u64p,
                                C.POINTER(abi.PairRec), C.c_uint64, u64p, C.c_uint32, C.c_uint32]
        L.emu_thread_batch.restype = C.c_int
        L.emu_thread_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.Params), u8p, u64p, u32p, C.c_uint64,
                                       C.POINTER(abi.ThreadRec), C.c_uint32]
        L.emu_set_walk_trecs.argtypes = [C.POINTER(abi.ThreadRec)]
        L.emu_walk_results.restype = C.c_uint64
        L.emu_walk_results.argtypes = [C.POINTER(abi.WalkRes), u32p, C.c_uint64]

    def ingest(self, data: bytes, fastq, min_read, chunk, head=256, line_cap=None, grid=3):
        """dbtk_ingest.h's kernel bodies over `data` cut into chunks: (headers, [(title, read 2q, read 2q + 1, qual 2q, qual 2q + 1)],
        input bytes consumed).  Headers are dicts; the list holds the kept pairs of every block that was not flagged dirty."""
        class Hdr(C.Structure):
            _fields_ = [("base", C.c_uint32), ("nlines", C.c_uint32), ("npairs", C.c_uint32), ("nkept", C.c_uint32), ("flags", C.c_uint32),
                        ("cut", C.c_uint32), ("carry", C.c_uint32), ("pad", C.c_uint32), ("flat_bytes", C.c_uint64), ("maxlen", C.c_uint64),
                        ("pad2", C.c_uint64 * 3)]
        L = self.L
        L.emu_ingest.restype = C.c_int
        L.emu_ingest.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Hdr), C.c_uint32,
                                 u8p, u8p, C.c_uint64, u32p, C.c_uint64, u8p, C.c_uint64, u64p]
        if line_cap is None:
            line_cap = (head + chunk) // 8 + 64
        maxb = len(data) // max(chunk, 1) + 3
        hdrs = (Hdr * maxb)()
        cap = len(data) + 64
        flat, qual, titles = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        lens = np.zeros(cap // 2 + 2, np.uint32)
        tot = np.zeros(4, np.uint64)
        nb = L.emu_ingest(data, len(data), int(bool(fastq)), min_read, chunk, head, line_cap, grid, hdrs, maxb, _p(flat, u8p), _p(qual, u8p), cap,
                          _p(lens, u32p), len(lens), _p(titles, u8p), cap, _p(tot, u64p))
        assert nb >= 0, f"emu_ingest failed: {nb}"
        H = [{f[0]: getattr(hdrs[i], f[0]) for f in Hdr._fields_[:10]} for i in range(nb)]
        nflat, nlens, ntit = int(tot[0]), int(tot[1]), int(tot[2])
        tl = bytes(titles[:ntit]).split(b"\n")[:-1] if ntit else []
        o, pairs = 0, []
        fb, qb = bytes(flat[:nflat]), bytes(qual[:nflat])
        for q in range(nlens // 2):
            l0, l1 = int(lens[2 * q]), int(lens[2 * q + 1])
            pairs.append((tl[q], fb[o:o + l0], fb[o + l0:o + l0 + l1], qb[o:o + l0] if fastq else b"", qb[o + l0:o + l0 + l1] if fastq else b""))
            o += l0 + l1
        return H, pairs, int(tot[3])

    def gz(self, text: bytes, grid=2):
        """dbtk_gz.h's member / scan / pack bodies: the gzip stream of `text`."""
        self.L.emu_gz.restype = C.c_int64
        self.L.emu_gz.argtypes = [C.c_char_p, C.c_uint64, u8p, C.c_uint64, C.c_uint32]
        cap = len(text) + len(text) // 7 + (len(text) // 65536 + 2) * 2048
        out = np.zeros(cap, np.uint8)
        n = self.L.emu_gz(text, len(text), _p(out, u8p), cap, grid)
        assert n >= 0
        return bytes(out[:n])

    def probe_stats(self):
        """(general-probe-body runs, lean-probe-body runs since the last call, keys the last tables' level 1 turned away)"""
        out = np.zeros(3, np.uint64)
        self.L.emu_probe_stats.argtypes = [u64p]
        self.L.emu_probe_stats(out.ctypes.data_as(u64p))
        return int(out[0]), int(out[1]), int(out[2])

    def walk_locus_stats(self):
        """(pairs the lean walk body took in its locus-resident form, pairs left to its plain form) since the last call"""
        out = np.zeros(2, np.uint64)
        self.L.emu_walk_locus_stats.argtypes = [u64p]
        self.L.emu_walk_locus_stats(out.ctypes.data_as(u64p))
        return int(out[0]), int(out[1])

    def locus_stats(self):
        """(pairs the locus-resident probe body took in each of its three classes of workgroup, pairs left to the lean body) since
        the last call; keys the last tables' images left out"""
        out = np.zeros(5, np.uint64)
        self.L.emu_locus_stats.argtypes = [u64p]
        self.L.emu_locus_stats(out.ctypes.data_as(u64p))
        return [int(out[0]), int(out[1]), int(out[2])], int(out[3]), int(out[4])

    def path_stats(self):
        """as Context.path_stats, over the align calls since the last call"""
        v = (C.c_uint64 * 24)()
        self.L.emu_path_stats(v)
        v = [int(x) for x in v]
        return {"probe_items": v[0:3], "probe_pairs": v[3:6], "probe_rest": v[6], "walk_items": v[7:10], "walk_pairs": v[10:13],
                "walk_rest": v[13], "fused_done": v[14], "fused_redone": v[15], "walk_locus_ec": v[18], "fused_shared": v[19], "lean_done": v[20], "vote_vv": v[21]}

    def aln_text(self, npairs):
        """as Context.aln_text: list of (pair, dst, text)"""
        self.L.emu_aln_text.restype = C.c_uint64
        self.L.emu_aln_text.argtypes = [u32p, C.c_uint64, u8p, C.c_uint64]
        idx = np.zeros(max(npairs, 1), np.uint32)
        used = int(self.L.emu_aln_text(_p(idx, u32p), npairs, None, 0))
        arena = np.zeros(max(used, 4), np.uint8)
        self.L.emu_aln_text(_p(idx, u32p), npairs, _p(arena, u8p), len(arena))
        out = []
        for p_ in range(npairs):
            o = int(idx[p_])
            if o != 0xFFFFFFFF:
                dst, ln = int(arena[o:o + 4].view(np.uint32)[0]), int(arena[o + 4:o + 8].view(np.uint32)[0])
                out.append((p_, dst, arena[o + 8:o + 8 + ln].tobytes().decode()))
        return out

    def walk_results(self, cap):
        res = (abi.WalkRes * max(cap, 1))()
        n = self.L.emu_walk_results(res, None, cap)
        return res, int(n)

    def aln_records(self):
        """[(AlnHdr, "cigar2\\tannot2\\tcigar1\\tannot1")] of the last align() with params.aln (as Context.aln_records)."""
        L = self.L
        L.emu_aln_records.restype = C.c_uint64
        L.emu_aln_records.argtypes = [C.c_void_p, C.c_uint64, u32p, u32p]
        st, cap = C.c_uint32(0), C.c_uint32(0)
        n = L.emu_aln_records(None, 0, C.byref(st), C.byref(cap))
        if not n:
            return []
        buf = (C.c_uint8 * (n * st.value))()
        L.emu_aln_records(buf, len(buf), C.byref(st), C.byref(cap))
        out, txt = [], C.create_string_buffer(8192)
        for i in range(n):
            L.dbtk_aln_format(C.byref(buf, i * st.value), cap.value, txt, 8192)
            out.append((abi.AlnHdr.from_buffer_copy(buf, i * st.value), txt.value.decode()))
        return out

    def thread(self, rpgg, tables, params, seq, off, loci, grid=5):
        off = np.ascontiguousarray(off, np.uint64)
        seq = np.ascontiguousarray(seq, np.uint8)
        loci = np.ascontiguousarray(loci, np.uint32)
        n = len(off) - 1
        recs = (abi.ThreadRec * max(n, 1))()
        rc = self.L.emu_thread_batch(rpgg.h, tables, C.byref(params), _p(seq, u8p), _p(off, u64p), _p(loci, u32p), n, recs, grid)
        if rc:
            raise RuntimeError(f"emu_thread_batch -> {rc}")
        return recs

    def tables(self, rpgg):
        return self.L.emu_tables_create(rpgg.h)

    def align_ex(self, rpgg, tables, params, seq, off, qual=None, evcap=1 << 20):
        npairs = (len(off) - 1) // 2
        counts = np.zeros(rpgg.ntrkmers, np.uint64)
        kmc = np.zeros(rpgg.nloci, np.uint64)
        nmap = np.zeros(rpgg.nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        recs = (abi.PairRec * max(npairs, 1))()
        ev = (BubEvent * evcap)()
        nrec, nev = C.c_uint64(0), C.c_uint64(0)
        seq = np.ascontiguousarray(seq, np.uint8)
        rc = self.L.emu_align_ex(rpgg.h, tables, C.byref(params), _p(seq, u8p), _p(off, u64p), _p(qual, u8p), npairs,
                                 _p(counts, u64p), _p(kmc, u64p), _p(nmap, u32p), _p(ctr, u64p), recs, npairs, C.byref(nrec), 2, 3,
                                 ev, evcap, C.byref(nev))
        if rc:
            raise RuntimeError(f"emu_align_ex -> {rc}")
        e = events_array(ev, min(nev.value, evcap))
        e = e[np.lexsort((e["pos"], e["mate"], e["pair"]))]
        return dict(counts=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs, events=e)

    def consistent(self, tables):
        return int(self.L.emu_tables_consistent(tables))

    def set_consistent(self, tables, v):
        self.L.emu_tables_set_consistent(tables, int(v))

    def selftest_fmt(self, seed, iters):
        self.L.emu_selftest_fmt.restype = C.c_uint64
        self.L.emu_selftest_fmt.argtypes = [C.c_uint64, C.c_uint64]
        return int(self.L.emu_selftest_fmt(seed, iters))

    def selftest_assign(self, seed, iters):
        return int(self.L.emu_selftest_assign(seed, iters))

    def selftest_wavesort(self, seed, iters):
        return int(self.L.emu_selftest_wavesort(seed, iters))

    def selftest_vote(self, seed, iters):
        return int(self.L.emu_selftest_vote(seed, iters))

    def selftest_sort(self, seed, iters):
        return int(self.L.emu_selftest_sort(seed, iters))

    def align(self, rpgg, tables, params, seq, off, grid_k1=3, grid_pair=5):
        npairs = (len(off) - 1) // 2
        counts = np.zeros(rpgg.ntrkmers, np.uint64)
        kmc = np.zeros(rpgg.nloci, np.uint64)
        nmap = np.zeros(rpgg.nloci, np.uint32)
        ctr = np.zeros(abi.C_COUNT, np.uint64)
        want = bool(params.trace or params.okam or params.extract)
        recs = (abi.PairRec * max(npairs, 1))() if want else None
        nrec = C.c_uint64(0)
        seq = np.ascontiguousarray(seq, np.uint8)
        if seq.size == 0:
            seq = np.zeros(1, np.uint8)
        rc = self.L.emu_align(rpgg.h, tables, C.byref(params), _p(seq, u8p), _p(off, u64p), npairs, _p(counts, u64p),
                              _p(kmc, u64p), _p(nmap, u32p), _p(ctr, u64p), recs, npairs if recs else 0, C.byref(nrec),
                              grid_k1, grid_pair)
        if rc:
            raise RuntimeError(f"emu_align -> {rc}")
        return dict(counts=counts, kmc=kmc, nmapread=nmap, counters=ctr, recs=recs, nrec=int(nrec.value))


def locus_image_lg(arrs, k):
    """lgnb of every locus' index image as dbtk_locus.h sizes it (loc_lgnb_for: the smallest table of 2^lg 4-slot buckets at a load of at
    most 0.8 over the keys whose index value names the locus), from the flat arrays; and the key counts."""
    nk, nloci = arrs.nkeys, arrs.nloci
    vals = np.ctypeslib.as_array(arrs.vals, (nk,))
    vv = np.ctypeslib.as_array(arrs.vv, (arrs.nvv,))
    even = (vals & 1) == 0
    cnt = np.bincount(vals[even] >> 1, minlength=nloci).astype(np.int64)
    offs = (vals[~even] >> 1).astype(np.int64)
    ns = vv[offs].astype(np.int64)
    idx = np.repeat(offs + 1, ns) + (np.arange(int(ns.sum())) - np.repeat(np.cumsum(ns) - ns, ns))
    cnt += np.bincount(vv[idx], minlength=nloci)[:nloci]
    lgmin = max(5, 2 * k - 40)
    lg = np.full(nloci, lgmin)
    for l in range(lgmin, 16):
        lg[(16 << l) < 5 * cnt] = l + 1
    return lg, cnt


def dense_loci(arrs, k, per_lg=650, seed=1):
    """Loci for a DENSE slice (Synth.reads_loci: many pairs per locus, the regime of the locus-resident kernels) covering every class of
    image the kernels have (up to 512, 1024, 2048 buckets): up to per_lg loci of every image size, every locus of the largest class.
    Returns (loci, classes present)."""
    lg, _ = locus_image_lg(arrs, k)
    rng = np.random.default_rng(seed)
    cls_of = lambda l: 0 if l <= 9 else 1 if l == 10 else 2 if l == 11 else -1
    loci = []
    for l in np.unique(lg):
        if cls_of(int(l)) < 0:
            continue
        pool = np.flatnonzero(lg == l)
        loci.append(pool if cls_of(int(l)) == 2 and len(pool) <= 400 else rng.choice(pool, min(len(pool), per_lg), replace=False))
    loci = rng.permutation(np.concatenate(loci))
    return loci, sorted({cls_of(int(lg[l])) for l in loci})
