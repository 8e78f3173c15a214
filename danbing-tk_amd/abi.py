"""ctypes mirror of include/dbtk.h (struct layouts and constants only)."""
import ctypes as C

MAX_READ_LEN = 256
NAN64 = 0xFFFFFFFFFFFFFFFF
NAN32 = 0xFFFFFFFF
ABI_VERSION = 8
ALN_TEXT = 4  # params.aln | ALN_TEXT: alignment records in text form (dbtk_ctx_aln_text)
THREAD_CAP = 384
ING_DIRTY, ING_LINES, ING_CARRY, ING_TAIL = 1, 2, 4, 8  # dbtk_ingest_info.flags
THREADING_HEAD, THREADING_V13 = 1, 2

(OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_NO_DEVICE, ERR_HIP, ERR_READ_TOO_LONG, ERR_NOMEM, ERR_UNSUPPORTED,
 ERR_OVERFLOW) = range(10)

(STAGE_SHORT, STAGE_SUBFILTER, STAGE_KFILTER, STAGE_LOCUS, STAGE_QC, STAGE_BAIT, STAGE_ASGN, STAGE_COUNTED,
 STAGE_EXTRACT, STAGE_THREADING) = range(10)

(C_NREADS, C_SUBFILTERED, C_KMERFILTERED, C_BAITFILTERED, C_QUALFILTERED, C_LOCUSFILTERED, C_QCFILTERED,
 C_THREADING, C_FEASIBLE, C_ASGN, C_NSHORT, C_NHASH0, C_NHASH1, C_ALGO_PROBES, C_ALGO_VV, C_ALGO_CLS, C_ALGO_INC,
 C_SURVIVORS, C_BASES) = range(19)
C_COUNT = 24

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
u8p = C.POINTER(C.c_uint8)


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("ksize", "n_filter", "nm_filter", "cthreshold", "nm_tr", "max_nt", "qth", "okam", "qc", "bait",
                 "bubbles", "extract", "trace", "threading", "simmode", "thread_cth", "maxncorrection", "correction", "aln", "trackbait", "diag")]


class RpggArrays(C.Structure):
    _fields_ = [
        ("ksize", C.c_uint32), ("nloci", C.c_uint64),
        ("nkeys", C.c_uint64), ("keys", u64p), ("vals", u32p),
        ("nvv", C.c_uint64), ("vv", u32p),
        ("fl_cnt", u64p), ("fl_ks", u64p),
        ("tre_cnt", u64p), ("tre_ks", u64p),
        ("tr_cnt", u64p), ("tr_ks", u64p),
        ("qc", u8p),
        ("bt_cnt", u64p), ("bt_ks", u64p), ("bt_vs", u16p),
        ("gr_cnt", u64p), ("gr_ks", u64p), ("gr_ms", u8p),
    ]


class MateRec(C.Structure):
    _fields_ = [(n, C.c_int16) for n in ("si", "ei", "si_", "ei_", "nt", "bs", "ti")] + \
               [(n, C.c_uint8) for n in ("kf", "hf", "bf", "qf", "af", "rm")] + \
               [("nk", C.c_uint16), ("as2", C.c_uint8 * (MAX_READ_LEN // 4))]

    def annot(self):
        """The `as` vector (0 '*', 1 '.', 2 '=')."""
        return [(self.as2[i >> 2] >> (2 * (i & 3))) & 3 for i in range(self.nk)]


class PairRec(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("stage", C.c_uint32), ("dst", C.c_uint32), ("dst0", C.c_uint32),
                ("nm1", C.c_int32), ("nm2", C.c_int32), ("r1", MateRec), ("r2", MateRec)]


class ThreadRec(C.Structure):
    """What isThreadFeasible leaves behind for one read (include/dbtk.h: dbtk_thread_rec_t)."""
    _fields_ = [("ret", C.c_int32), ("ni", C.c_int32), ("nkm", C.c_uint32), ("nes", C.c_uint32), ("ntr", C.c_uint32),
                ("flags", C.c_uint32),
                ("es_t", C.c_uint8 * THREAD_CAP), ("es_r", C.c_uint8 * THREAD_CAP), ("es_g", C.c_uint8 * THREAD_CAP),
                ("tr", C.c_uint8 * THREAD_CAP), ("kmers", C.c_uint64 * THREAD_CAP)]


class WalkRes(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("dst", C.c_uint32), ("ret1", C.c_int8), ("ret2", C.c_int8), ("pad", C.c_uint8 * 2)]


class AlnHdr(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("dst", C.c_uint32), ("ret1", C.c_int8), ("ret2", C.c_int8), ("pad", C.c_uint8 * 2),
                ("nes1", C.c_uint16), ("ntr1", C.c_uint16), ("nes2", C.c_uint16), ("ntr2", C.c_uint16), ("pad2", C.c_uint32)]


LOAD_INDEX_ONLY, LOAD_GRAPH = 1, 2


def default_params(**kw) -> Params:
    """Defaults of the reference: src/aQueryFasta_thread.cpp:26-34, 2336-2339."""
    p = Params(ksize=21, n_filter=4, nm_filter=1, cthreshold=10, nm_tr=40, max_nt=2, qth=20, okam=1, thread_cth=100,
               maxncorrection=4)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class IngestInfo(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("npairs", C.c_uint32), ("nkept", C.c_uint32), ("max_read_len", C.c_uint32),
                ("first_byte", C.c_uint64), ("cut_byte", C.c_uint64), ("seq_bytes", C.c_uint64)]


class IngestSpan(C.Structure):
    _fields_ = [("title", C.c_uint32), ("title_len", C.c_uint32), ("seq", C.c_uint32 * 2), ("seq_len", C.c_uint32 * 2),
                ("qual", C.c_uint32 * 2), ("qual_len", C.c_uint32 * 2)]
