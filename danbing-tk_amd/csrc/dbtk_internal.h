// dbtk_internal.h — host-side structures shared by dbtk_rpgg.cpp and dbtk_hip.hip.
#ifndef DBTK_INTERNAL_H_
#define DBTK_INTERNAL_H_

#include <stdint.h>

#include <atomic>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/dbtk.h"

// The RPGG exactly as the reference's loaders see it (flat, file-equivalent),
// plus the output order derived once at load time.
inline uint64_t dbtk_next_uid() { static std::atomic<uint64_t> n{0}; return ++n; }
struct dbtk_rpgg {
    // Process-unique, never reused: what the per-device caches of the tables built from this handle are keyed by (a freed
    // handle's ADDRESS can come back with the next `new`; its id cannot, so a context leaked with its tables can never lend
    // them to another RPGG).
    const uint64_t uid = dbtk_next_uid();
    uint32_t ksize = 0;
    uint64_t nloci = 0;
    std::vector<uint64_t> keys;   // PREF.kmers.dbi
    std::vector<uint32_t> vals;
    std::vector<uint32_t> vv;
    std::vector<uint64_t> fl_cnt, fl_ks;    // PREF.fl.kdb
    std::vector<uint64_t> tre_cnt, tre_ks;  // PREF.tre.kdb (may be empty)
    std::vector<uint64_t> tr_cnt, tr_ks;    // PREF.tr.kmers, file order
    std::vector<uint8_t> qc;                // empty or nloci
    std::vector<uint64_t> bt_cnt, bt_ks;    // PREF.bt.kmdb (may be empty)
    std::vector<uint16_t> bt_vs;
    std::vector<uint64_t> gr_cnt, gr_ks;    // PREF.graph.kmers / PREF.graph.umap (empty: no graph loaded)
    std::vector<uint8_t> gr_ms;
    // derived
    bool order_done = false;          // out_slot / out_beg / out_kmer are made (finish_order: the loader runs it beside the index file's read)
    std::vector<uint64_t> out_slot;   // file index -> position in OUT.trkmc.ar
    std::vector<uint64_t> out_kmer;   // position -> k-mer
    std::vector<uint64_t> out_beg;    // nloci+1: first position of each locus
    // sidecar of the GPU-layout per-locus index images (dbtk_locus.h; dbtk_rpgg_set_index_cache): its file and what to do with it
    std::string idx_cache;
    int idx_cache_mode = 0;           // 0: none; 1: load it when present and valid; 2: that, and write it after a build
};

namespace dbtk {
void set_error(const std::string& msg);
dbtk_status_t finish_rpgg(dbtk_rpgg* g);  // validation + output order
// No exception crosses the C-ABI: an entry point that parses files or allocates host memory runs its body through this
// (a count field of a damaged file can ask for terabytes: std::bad_alloc / std::length_error become status codes).
template <class F> dbtk_status_t guarded(F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { set_error("out of host memory (a damaged file, or a batch too large for this host)"); return DBTK_ERR_NOMEM; }
    catch (const std::exception& e) { set_error(std::string("internal error: ") + e.what()); return DBTK_ERR_FORMAT; }
    catch (...) { set_error("internal error"); return DBTK_ERR_FORMAT; }
}
}  // namespace dbtk

#endif
