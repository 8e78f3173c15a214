// dbtk_rpgg.cpp — host side of the RPGG handle: the reference's on-disk formats
// in (loaders of src/aQueryFasta_thread.cpp:2459-2500) and out (dumps of
// src/aQueryFasta_thread.cpp:2631-2641).  No device code here.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <unordered_set>
#include <chrono>
#include <thread>
#include <unordered_map>

#include "dbtk_internal.h"

namespace dbtk {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace dbtk
using dbtk::set_error;

extern "C" const char* dbtk_last_error(void) { return dbtk::g_err.c_str(); }
extern "C" uint32_t dbtk_abi_version(void) { return DBTK_ABI_VERSION; }

namespace {

struct File {
    FILE* f = nullptr;
    explicit File(const std::string& fn, const char* mode) { f = fopen(fn.c_str(), mode); }
    ~File() { if (f) fclose(f); }
    template <class T> bool read(T* p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }
    template <class T> bool write(const T* p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }
};

// PREF.*.kdb: u64 nloci | u64 cnt[nloci] | u64 nk | u64 ks[nk]
// (serializeKsetDB, src/binaryKmerIO.hpp:128-139; reader src/aQueryFasta_thread.h:675-698)
dbtk_status_t read_kdb(const std::string& fn, uint64_t nloci, std::vector<uint64_t>& cnt, std::vector<uint64_t>& ks) {
    File f(fn, "rb");
    if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
    uint64_t nl = 0, nk = 0;
    if (!f.read(&nl, 1)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    if (nloci != ~0ull && nl != nloci) { set_error(fn + ": locus count differs from tr.kmers"); return DBTK_ERR_FORMAT; }
    {   // (before anything is sized from the header: the file must hold what it announces)
        const long at = ftell(f.f);
        fseek(f.f, 0, SEEK_END);
        const uint64_t fsz = (uint64_t)ftell(f.f);
        fseek(f.f, at, SEEK_SET);
        if (nl > fsz / 8) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    }
    cnt.resize(nl);
    if (!f.read(cnt.data(), nl) || !f.read(&nk, 1)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    uint64_t sum = 0;
    for (uint64_t c : cnt) sum += c;
    if (sum != nk) { set_error(fn + ": per-locus counts do not add up"); return DBTK_ERR_FORMAT; }
    ks.resize(nk);
    if (!f.read(ks.data(), nk)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    return DBTK_OK;
}

// PREF.tr.kmers: ">locus" lines and "KMER[\tVALUE]" lines; only the first field
// is used (countLoci src/kmerIO.hpp:33-45, readKmersWithZeroCount
// src/aQueryFasta_thread.h:469-480).
// The text is cut at line ends into one piece per thread (484 MB of decimal k-mers at release scale: a second of one core); a piece's
// k-mer lines before its first '>' line belong to the last locus of the piece before.
dbtk_status_t read_tr_kmers(const std::string& fn, std::vector<uint64_t>& cnt, std::vector<uint64_t>& ks) {
    File f(fn, "rb");
    if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
    fseek(f.f, 0, SEEK_END);
    const long sz = ftell(f.f);
    fseek(f.f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz + 1);
    if (sz && fread(buf.data(), 1, (size_t)sz, f.f) != (size_t)sz) { set_error("short read on " + fn); return DBTK_ERR_IO; }
    buf[(size_t)sz] = '\n';
    const char* const base = buf.data();
    const char* const end = base + sz;
    const unsigned nth = (size_t)sz < (8u << 20) ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<const char*> cut(nth + 1, end);
    cut[0] = base;
    for (unsigned t = 1; t < nth; ++t) {  // the first line start at or after the t-th share of the bytes
        const char* p = base + (size_t)sz / nth * t;
        if (p <= cut[t - 1]) { cut[t] = cut[t - 1]; continue; }
        const char* nl = (const char*)memchr(p - 1, '\n', (size_t)(end - (p - 1)) + 1);
        cut[t] = nl + 1 < end ? nl + 1 : end;
    }
    struct Piece { std::vector<uint64_t> cnt, ks; uint64_t head = 0; int err = 0; };  // head: k-mer lines before the piece's first '>' line
    std::vector<Piece> pc(nth);
    auto parse_piece = [&](unsigned t) {
        Piece& P = pc[t];
        const char* p = cut[t];
        const char* const pe = cut[t + 1];
        bool seen = false;
        while (p < pe) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p) + 1);
            if (*p == '>') {
                P.cnt.push_back(0);
                seen = true;
            } else if (nl > p) {
                uint64_t v = 0;
                const char* q = p;
                while (q < nl && (*q == ' ' || *q == '\t')) ++q;
                if (q == nl || *q < '0' || *q > '9') { P.err = 2; return; }
                while (q < nl && *q >= '0' && *q <= '9') v = v * 10 + (uint64_t)(*q++ - '0');
                P.ks.push_back(v);
                if (seen) P.cnt.back()++; else P.head++;
            }
            p = nl + 1;
        }
    };
    // (no exception may leave a thread: a bad_alloc from a piece's push_back becomes the piece's error — ADVICE r5)
    auto parse = [&](unsigned t) {
        try { parse_piece(t); } catch (const std::bad_alloc&) { pc[t].err = 3; } catch (...) { pc[t].err = 4; }
    };
    if (nth == 1) parse(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; ++t) th.emplace_back(parse, t);
        for (auto& x : th) x.join();
    }
    size_t nl = 0, nk = 0;
    for (auto& P : pc) { nl += P.cnt.size(); nk += P.ks.size(); }
    cnt.reserve(cnt.size() + nl);
    ks.reserve(ks.size() + nk);
    for (auto& P : pc) {  // (in file order: the first error of the file is the one reported)
        if (P.err == 3) { set_error(fn + ": out of memory"); return DBTK_ERR_NOMEM; }
        if (P.err == 4) { set_error(fn + ": internal error while parsing"); return DBTK_ERR_FORMAT; }
        // a line before the file's first '>' is "a k-mer before the first '>' line" whether it parses as a k-mer or not (the message the
        // one-threaded reader gave: a piece that stopped at a malformed line before any '>' counts it as head)
        if ((P.head || (P.err && P.cnt.empty())) && cnt.empty()) { set_error(fn + ": k-mer before the first '>' line"); return DBTK_ERR_FORMAT; }
        if (P.head) cnt.back() += P.head;
        if (P.err) { set_error(fn + ": not a k-mer line"); return DBTK_ERR_FORMAT; }
        cnt.insert(cnt.end(), P.cnt.begin(), P.cnt.end());
        ks.insert(ks.end(), P.ks.begin(), P.ks.end());
    }
    return DBTK_OK;
}

// graphDB of the v1.3 threading path.  PREF.graph.kmers: text, ">locus" lines then "NODE\tMASK" lines
// (readGraphKmers, src/aQueryFasta_thread.h:550-575: `kmerDB[idx][kmer] |= mask`, so repeated nodes OR — kept as
// repeated entries here, the device table ORs them).  PREF.graph.umap: the v1.3 binary
// `u64 nloci | per locus: u64 n | n x (u64 node, u8 mask)` (fixture test/QC/input/pan.graph.umap).
dbtk_status_t read_graph_text(const std::string& fn, uint64_t nloci, std::vector<uint64_t>& cnt, std::vector<uint64_t>& ks,
                              std::vector<uint8_t>& ms) {
    File f(fn, "rb");
    if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
    cnt.assign(nloci, 0);
    std::vector<char> buf(1 << 24);
    std::string carry;
    int64_t locus = -1;
    auto line = [&](const char* p, const char* e) -> bool {
        if (p == e) return true;
        if (*p == '>') { ++locus; return (uint64_t)locus < nloci; }
        if (locus < 0) return false;
        uint64_t v = 0, m = 0;
        const char* q = p;
        if (*q < '0' || *q > '9') return false;
        while (q < e && *q >= '0' && *q <= '9') v = v * 10 + (uint64_t)(*q++ - '0');
        if (q < e && *q == '\t') { ++q; while (q < e && *q >= '0' && *q <= '9') m = m * 10 + (uint64_t)(*q++ - '0'); }
        ks.push_back(v);
        ms.push_back((uint8_t)m);
        cnt[(size_t)locus]++;
        return true;
    };
    for (;;) {
        const size_t n = fread(buf.data(), 1, buf.size(), f.f);
        if (!n) break;
        const char* p = buf.data();
        const char* end = p + n;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!nl) { carry.append(p, end); break; }
            bool ok;
            if (!carry.empty()) { carry.append(p, nl); ok = line(carry.data(), carry.data() + carry.size()); carry.clear(); }
            else ok = line(p, nl);
            if (!ok) { set_error(fn + ": not a graph k-mer file (or more loci than PREF.tr.kmers)"); return DBTK_ERR_FORMAT; }
            p = nl + 1;
        }
    }
    if (!carry.empty() && !line(carry.data(), carry.data() + carry.size())) { set_error(fn + ": bad last line"); return DBTK_ERR_FORMAT; }
    return DBTK_OK;
}
dbtk_status_t read_graph_umap(const std::string& fn, uint64_t nloci, std::vector<uint64_t>& cnt, std::vector<uint64_t>& ks,
                              std::vector<uint8_t>& ms) {
    File f(fn, "rb");
    if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
    uint64_t nl = 0;
    if (!f.read(&nl, 1)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    if (nl != nloci) { set_error(fn + ": locus count differs from tr.kmers"); return DBTK_ERR_FORMAT; }
    cnt.assign(nloci, 0);
    std::vector<uint8_t> rec;
    for (uint64_t l = 0; l < nl; ++l) {
        uint64_t n = 0;
        if (!f.read(&n, 1) || n > (1ull << 40)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
        cnt[l] = n;
        rec.resize(n * 9);
        if (!f.read(rec.data(), n * 9)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
        const size_t at = ks.size();
        ks.resize(at + n);
        ms.resize(at + n);
        for (uint64_t i = 0; i < n; ++i) { memcpy(&ks[at + i], &rec[9 * i], 8); ms[at + i] = rec[9 * i + 8]; }
    }
    return DBTK_OK;
}

}  // namespace

namespace dbtk {

// Output order + sanity checks.  The reference writes per-locus counts in the
// iteration order of std::unordered_map<size_t, atomic<size_t>> filled in file
// order (src/aQueryFasta_thread.h:469-480 -> src/binaryKmerIO.hpp:36-46,
// src/aQueryFasta_thread.h:929-936); libstdc++'s own container is the order
// oracle here, instantiated the same way (identity hash, operator[] inserts).
// The OUT.trkmc.ar order (see finish_rpgg): needs the TR k-mers alone, so the loader runs it on the thread that parsed them while
// the index file is still coming in (0.19 s of a 0.54-s load at release scale)
dbtk_status_t finish_order(dbtk_rpgg* g) {
    if (g->order_done) return DBTK_OK;
    const uint64_t nloci = g->tr_cnt.size();
    if (nloci >= 0x7FFFFFFFull) { set_error("too many loci"); return DBTK_ERR_FORMAT; }
    const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const auto tclk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<uint64_t> beg(nloci + 1, 0);
    for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->tr_cnt[l];
    const uint64_t ntr = beg[nloci];
    if (g->tr_ks.size() != ntr) { set_error("tr k-mer array has the wrong length"); return DBTK_ERR_FORMAT; }
    g->out_slot.assign(ntr, 0);
    g->out_beg.assign(nloci + 1, 0);
    std::vector<uint64_t> uniq(nloci, 0);
    // pass 1 (parallel over loci): local order within each locus
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nth; ++t) {
        th.emplace_back([&, t]() {
            for (uint64_t l = t; l < nloci; l += nth) {
                std::unordered_map<size_t, uint64_t> m;  // value: first file index of the key
                for (uint64_t i = beg[l]; i < beg[l + 1]; ++i) {
                    const size_t before = m.size();
                    uint64_t& v = m[(size_t)g->tr_ks[i]];  // operator[], like `kmerDB[idx][stoul(line)] = 0`
                    if (m.size() != before) v = i;
                }
                uint64_t pos = 0;
                for (auto& p : m) g->out_slot[p.second] = pos++;  // local position, made global below
                // duplicate lines of a k-mer share the node of its first occurrence
                for (uint64_t i = beg[l]; i < beg[l + 1]; ++i) {
                    const uint64_t first_i = m.find((size_t)g->tr_ks[i])->second;
                    if (first_i != i) g->out_slot[i] = g->out_slot[first_i];
                }
                uniq[l] = pos;
            }
        });
    }
    for (auto& x : th) x.join();
    for (uint64_t l = 0; l < nloci; ++l) g->out_beg[l + 1] = g->out_beg[l] + uniq[l];
    g->out_kmer.assign(g->out_beg[nloci], 0);
    th.clear();
    for (unsigned t = 0; t < nth; ++t)  // (a locus' slots are its own range of out_kmer)
        th.emplace_back([&, t]() {
            for (uint64_t l = nloci / nth * t, e = t + 1 == nth ? nloci : nloci / nth * (t + 1); l < e; ++l)
                for (uint64_t i = beg[l]; i < beg[l + 1]; ++i) {
                    g->out_slot[i] += g->out_beg[l];
                    g->out_kmer[g->out_slot[i]] = g->tr_ks[i];
                }
        });
    for (auto& x : th) x.join();
    if (g->out_beg[nloci] >= 0xFFFFFFF0ull) { set_error("too many TR k-mers for 32-bit slots"); return DBTK_ERR_FORMAT; }
    g->order_done = true;
    (void)tclk;
    return DBTK_OK;
}

dbtk_status_t finish_rpgg(dbtk_rpgg* g) {
    const uint64_t nloci = g->nloci;
    const auto tclk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tf0 = tclk();
    if (g->ksize < 2 || g->ksize > 31) { set_error("k must be in 2..31"); return DBTK_ERR_ARG; }
    if (g->tr_cnt.size() != nloci || g->fl_cnt.size() != nloci) { set_error("per-locus count arrays have the wrong length"); return DBTK_ERR_FORMAT; }
    if (nloci >= 0x7FFFFFFFull) { set_error("too many loci"); return DBTK_ERR_FORMAT; }
    const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    // every locus id reachable from the index must exist (the reference would index hits1[] out of bounds)
    {
        std::vector<int> bad(nth, 0);
        std::vector<std::thread> tv;
        const size_t nv = g->vals.size();
        for (unsigned t = 0; t < nth; ++t)
            tv.emplace_back([&, t]() {
                for (size_t i = nv / nth * t, e = t + 1 == nth ? nv : nv / nth * (t + 1); i < e; ++i) {
                    const uint32_t v = g->vals[i];
                    if (v & 1) {
                        const uint64_t o = v >> 1;
                        if (o >= g->vv.size() || o + 1 + g->vv[o] > g->vv.size()) { bad[t] = 1; return; }
                    } else if ((v >> 1) >= nloci) { bad[t] = 2; return; }
                }
            });
        for (auto& x : tv) x.join();
        for (int b : bad) {
            if (b == 1) { set_error("kmers.dbi: value points outside vv"); return DBTK_ERR_FORMAT; }
            if (b == 2) { set_error("kmers.dbi: locus id out of range"); return DBTK_ERR_FORMAT; }
        }
    }
    for (size_t o = 0; o < g->vv.size();) {
        const uint64_t n = g->vv[o];
        if (o + 1 + n > g->vv.size()) break;  // tail not addressed by any value: ignored
        for (uint64_t j = 0; j < n; ++j)
            if (g->vv[o + 1 + j] >= nloci) { set_error("kmers.dbi: vv locus id out of range"); return DBTK_ERR_FORMAT; }
        o += 1 + n;
    }
    if (!g->bt_cnt.empty()) {  // the bait DB's per-locus counts index bt_ks / bt_vs (trackbait loop, build_kl_table)
        uint64_t s = 0;
        for (uint64_t c : g->bt_cnt) s += c;
        if (g->bt_cnt.size() != nloci || s != g->bt_ks.size() || s != g->bt_vs.size()) { set_error("bait DB: per-locus counts do not add up"); return DBTK_ERR_FORMAT; }
    }
    if (!g->gr_cnt.empty()) {
        uint64_t s = 0;
        for (uint64_t c : g->gr_cnt) s += c;
        if (g->gr_cnt.size() != nloci || s != g->gr_ks.size() || s != g->gr_ms.size()) { set_error("graph: per-locus counts do not add up"); return DBTK_ERR_FORMAT; }
    }
    const double tf1 = tclk();
    const dbtk_status_t so = finish_order(g);
    if (so) return so;
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "rpgg finish: checks %.2f s, output order + slots %.2f s (0: made beside the index file's read)\n", tf1 - tf0, tclk() - tf1);
    return DBTK_OK;
}

}  // namespace dbtk

namespace {
bool file_exists(const std::string& fn) { FILE* f = fopen(fn.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }
}  // namespace

extern "C" {

static dbtk_status_t dbtk_rpgg_load_impl(const char* prefix, const char* tr_kmers_file, uint32_t ksize, const char* qc_file, const char* bait_file,
                             uint32_t flags, dbtk_rpgg_t** out) {
    if (!prefix || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    std::unique_ptr<dbtk_rpgg> g(new dbtk_rpgg);
    g->ksize = ksize;
    const std::string pref(prefix);
    // (the TR k-mer file may be named explicitly: -t N reads PREF.tr.trimN.kmers in place of PREF.tr.kmers, AQ.cpp:2389, 2459, 2493)
    const bool head_index = file_exists(pref + ".kmers.dbi");
    const bool legacy = !head_index && file_exists(pref + ".kmerDBi.umap");
    // The HEAD files are read side by side: the text of PREF.tr.kmers is parsed (then PREF.fl.kdb / PREF.tre.kdb read, which are checked
    // against its locus count) on one thread while PREF.kmers.dbi, the largest, comes in on this one.  (Errors are thread-local
    // strings: the helper's is handed over with its status.)
    dbtk_status_t st = DBTK_OK, st_side = DBTK_OK;
    std::string err_side;
    const bool side_kdb = !legacy && !(flags & DBTK_LOAD_INDEX_ONLY);
    std::thread side([&] {
        st_side = dbtk::guarded([&]() -> dbtk_status_t {  // (no exception may leave the thread: a damaged count field asks for terabytes)
            // (PREF.fl.kdb — 0.9 GB at release scale — and PREF.tre.kdb come in on a thread of their own while the text of PREF.tr.kmers
            // is parsed; their locus counts are checked against its once both are there)
            dbtk_status_t s2 = DBTK_OK;
            std::string err2;
            std::thread kdb([&] {
                if (!side_kdb) return;
                s2 = dbtk::guarded([&]() -> dbtk_status_t {
                    dbtk_status_t r = read_kdb(pref + ".fl.kdb", ~0ull, g->fl_cnt, g->fl_ks);
                    if (!r) r = read_kdb(pref + ".tre.kdb", ~0ull, g->tre_cnt, g->tre_ks);
                    return r;
                });
                if (s2) err2 = dbtk::g_err;
            });
            dbtk_status_t s1 = dbtk::guarded([&]() -> dbtk_status_t {
                const dbtk_status_t r = read_tr_kmers(tr_kmers_file ? std::string(tr_kmers_file) : pref + ".tr.kmers", g->tr_cnt, g->tr_ks);
                return r ? r : dbtk::finish_order(g.get());  // (the output order needs nothing else: made here, beside the index file's read)
            });
            const std::string err1 = s1 ? dbtk::g_err : std::string();
            kdb.join();
            if (s1) { set_error(err1); return s1; }
            if (s2) { set_error(err2); return s2; }
            if (side_kdb) {
                const uint64_t nl = g->tr_cnt.size();
                if (g->fl_cnt.size() != nl) { set_error(pref + ".fl.kdb: locus count differs from tr.kmers"); return DBTK_ERR_FORMAT; }
                if (g->tre_cnt.size() != nl) { set_error(pref + ".tre.kdb: locus count differs from tr.kmers"); return DBTK_ERR_FORMAT; }
            }
            return DBTK_OK;
        });
        if (st_side) err_side = dbtk::g_err;
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{side};
    auto read_index = [&]() -> dbtk_status_t {
    if (!legacy) {
        // PREF.kmers.dbi: u64 nk | u64 keys[nk] | u32 vals[nk] | u64 nvv | u32 vv[nvv]
        // (src/kmertools.cpp:271-280; reader src/aQueryFasta_thread.h:654-673)
        const std::string fn = pref + ".kmers.dbi";
        File f(fn, "rb");
        if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
        uint64_t nk = 0, nvv = 0;
        if (!f.read(&nk, 1)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
        fseek(f.f, 0, SEEK_END);
        const uint64_t fsz = (uint64_t)ftell(f.f);
        if (nk > fsz / 12 || 8 + 12 * nk + 8 > fsz) { set_error("truncated " + fn); return DBTK_ERR_IO; }  // (before anything is sized from the header)
        g->keys.resize(nk);
        g->vals.resize(nk);
        {   // 1.7 GB at release scale: the keys and the values come in as eight pieces side by side (one thread's fread: 0.45 s of the start-up)
            const int fd = fileno(f.f);
            const unsigned np = nk < (1u << 20) ? 1u : 8u;
            std::vector<std::thread> rd;
            std::vector<int> ok(2 * np, 1);
            auto piece = [&](char* dst, uint64_t off, uint64_t n, int* good) {
                while (n) {
                    const ssize_t r = pread(fd, dst, (size_t)std::min<uint64_t>(n, 1ull << 30), (off_t)off);
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { *good = 0; return; }
                    dst += r; off += (uint64_t)r; n -= (uint64_t)r;
                }
            };
            for (unsigned t = 0; t < np; ++t) {
                const uint64_t a0 = nk / np * t, a1 = t + 1 == np ? nk : nk / np * (t + 1);
                rd.emplace_back(piece, (char*)(g->keys.data() + a0), 8 + 8 * a0, 8 * (a1 - a0), &ok[2 * t]);
                rd.emplace_back(piece, (char*)(g->vals.data() + a0), 8 + 8 * nk + 4 * a0, 4 * (a1 - a0), &ok[2 * t + 1]);
            }
            for (auto& x : rd) x.join();
            for (int o : ok) if (!o) { set_error("truncated " + fn); return DBTK_ERR_IO; }
        }
        fseek(f.f, (long)(8 + 12 * nk), SEEK_SET);
        if (!f.read(&nvv, 1) || nvv > fsz / 4) { set_error("truncated " + fn); return DBTK_ERR_IO; }
        g->vv.resize(nvv);
        if (!f.read(g->vv.data(), nvv)) { set_error("truncated " + fn); return DBTK_ERR_IO; }
    } else {
        // v1.3 RPGG (what README / pipeline name; fixtures test/QC/input/pan.*), accepted when the HEAD files
        // are absent:  PREF.kmerDBi.umap = u64 n | n x (u64 key, u64 val), val even -> locus = val >> 1, odd -> row val >> 1
        // of PREF.kmerDBi.vv = u64 n_outer | per row u64 len | u32 loci[len].  Converted to the HEAD encoding (odd val ->
        // offset into one flat u32 array whose entry is the row length followed by the loci).
        const std::string fu = pref + ".kmerDBi.umap", fv = pref + ".kmerDBi.vv";
        File f(fu, "rb");
        if (!f.f) { set_error("cannot open " + fu); return DBTK_ERR_IO; }
        uint64_t nk = 0;
        if (!f.read(&nk, 1)) { set_error("truncated " + fu); return DBTK_ERR_IO; }
        std::vector<uint64_t> kv(2 * nk);
        if (!f.read(kv.data(), 2 * nk)) { set_error("truncated " + fu); return DBTK_ERR_IO; }
        std::vector<uint64_t> rowoff;  // flat offset of every .vv row
        {
            File h(fv, "rb");
            uint64_t nout = 0;
            if (h.f && h.read(&nout, 1)) {
                for (uint64_t r = 0; r < nout; ++r) {
                    uint64_t len = 0;
                    if (!h.read(&len, 1) || len >= 0xFFFFFFFFull) { set_error("truncated " + fv); return DBTK_ERR_IO; }
                    rowoff.push_back(g->vv.size());
                    g->vv.push_back((uint32_t)len);
                    const size_t at = g->vv.size();
                    g->vv.resize(at + len);
                    if (!h.read(g->vv.data() + at, len)) { set_error("truncated " + fv); return DBTK_ERR_IO; }
                }
            }
        }
        g->keys.resize(nk);
        g->vals.resize(nk);
        for (uint64_t i = 0; i < nk; ++i) {
            const uint64_t v = kv[2 * i + 1];
            g->keys[i] = kv[2 * i];
            if (v & 1) {
                if ((v >> 1) >= rowoff.size() || rowoff[v >> 1] >= 0x7FFFFFFFull) { set_error(fu + ": value points outside " + fv); return DBTK_ERR_FORMAT; }
                g->vals[i] = (uint32_t)((rowoff[v >> 1] << 1) | 1);
            } else {
                if ((v >> 1) >= 0x7FFFFFFFull) { set_error(fu + ": locus out of range"); return DBTK_ERR_FORMAT; }
                g->vals[i] = (uint32_t)v;
            }
        }
    }
    return DBTK_OK;
    };
    const auto tclk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tl0 = tclk();
    st = read_index();
    const double tl1 = tclk();
    const std::string err_main = st ? dbtk::g_err : std::string();
    side.join();
    const double tl2 = tclk();
    if (st_side) { set_error(err_side); return st_side; }  // (PREF.tr.kmers is what the reference opens first: its error comes first)
    if (st) { set_error(err_main); return st; }
    g->nloci = g->tr_cnt.size();
    if (flags & DBTK_LOAD_INDEX_ONLY) {
        g->fl_cnt.assign(g->nloci, 0);
        g->tre_cnt.assign(g->nloci, 0);
    } else if (legacy) {
        // flank k-mers: PREF.ntr.kmers (text, the v1.3 name of .fl.kmers); no edge DB in v1.3 (so no -bu)
        if ((st = read_tr_kmers(pref + ".ntr.kmers", g->fl_cnt, g->fl_ks))) return st;
        if (g->fl_cnt.size() != g->nloci) { set_error(pref + ".ntr.kmers: locus count differs from .tr.kmers"); return DBTK_ERR_FORMAT; }
        g->tre_cnt.assign(g->nloci, 0);
    }  // (HEAD: PREF.fl.kdb and PREF.tre.kdb were read by the helper thread)
    if (qc_file && (flags & DBTK_LOAD_INDEX_ONLY)) {
        // extract mode never calls readQCFile (AQ.cpp:2484-2488): the mask stays as constructed, all zero
        // (AQ.cpp:2471), so with -qc every assigned pair fails the QC gate.  Reproduced as is.
        g->qc.assign(g->nloci, 0);
    } else if (qc_file) {  // readQCFile, src/kmerIO.hpp:111-120
        File f(qc_file, "rb");
        if (!f.f) { set_error(std::string("cannot open ") + qc_file); return DBTK_ERR_IO; }
        g->qc.resize(g->nloci);
        if (!f.read(g->qc.data(), g->nloci)) { set_error(std::string("truncated ") + qc_file); return DBTK_ERR_IO; }
        for (auto& b : g->qc) b = (uint8_t)(b - 48);
    }
    if (bait_file) {
        // PREF.bt.kmdb: u64 nloci | u64 cnt[nloci] | u64 nk | u64 sizeofval | u64 ks[nk] | u16 vs[nk]
        // (serializeKmapDB src/binaryKmerIO.hpp:53-68; reader :70-98)
        File f(bait_file, "rb");
        if (!f.f) { set_error(std::string("cannot open ") + bait_file); return DBTK_ERR_IO; }
        uint64_t nl = 0, nk = 0, szv = 0;
        if (!f.read(&nl, 1) || nl != g->nloci) { set_error("bait DB: locus count differs"); return DBTK_ERR_FORMAT; }
        g->bt_cnt.resize(nl);
        if (!f.read(g->bt_cnt.data(), nl) || !f.read(&nk, 1) || !f.read(&szv, 1) || szv != 2) { set_error("bait DB: bad header"); return DBTK_ERR_FORMAT; }
        {   // sanity before sizing anything from the header: the file must hold what it announces
            fseek(f.f, 0, SEEK_END);
            const uint64_t fsz = (uint64_t)ftell(f.f);
            if (nk > fsz / 10) { set_error("bait DB: k-mer count exceeds the file size"); return DBTK_ERR_FORMAT; }
            fseek(f.f, (long)(8 * (3 + nl)), SEEK_SET);
        }
        g->bt_ks.resize(nk);
        g->bt_vs.resize(nk);
        if (!f.read(g->bt_ks.data(), nk) || !f.read(g->bt_vs.data(), nk)) { set_error("bait DB truncated"); return DBTK_ERR_IO; }
    }
    if (flags & DBTK_LOAD_GRAPH) {
        if (file_exists(pref + ".graph.kmers")) st = read_graph_text(pref + ".graph.kmers", g->nloci, g->gr_cnt, g->gr_ks, g->gr_ms);
        else st = read_graph_umap(pref + ".graph.umap", g->nloci, g->gr_cnt, g->gr_ks, g->gr_ms);
        if (st) return st;
    }
    if ((st = dbtk::finish_rpgg(g.get()))) return st;
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "rpgg load: index file %.2f s, + TR / flank / edge files (side thread) %.2f s, checks + output order %.2f s\n", tl1 - tl0, tl2 - tl1, tclk() - tl2);
    *out = g.release();
    return DBTK_OK;
}

static dbtk_status_t dbtk_rpgg_from_arrays_impl(const dbtk_rpgg_arrays_t* a, dbtk_rpgg_t** out) {
    if (!a || !out || !a->tr_cnt || !a->fl_cnt) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    std::unique_ptr<dbtk_rpgg> g(new dbtk_rpgg);
    g->ksize = a->ksize;
    g->nloci = a->nloci;
    auto sum = [&](const uint64_t* c) { uint64_t s = 0; for (uint64_t l = 0; l < a->nloci; ++l) s += c[l]; return s; };
    g->keys.assign(a->keys, a->keys + a->nkeys);
    g->vals.assign(a->vals, a->vals + a->nkeys);
    if (a->nvv) g->vv.assign(a->vv, a->vv + a->nvv);
    g->fl_cnt.assign(a->fl_cnt, a->fl_cnt + a->nloci);
    { const uint64_t n = sum(a->fl_cnt); if (n) g->fl_ks.assign(a->fl_ks, a->fl_ks + n); }
    g->tr_cnt.assign(a->tr_cnt, a->tr_cnt + a->nloci);
    { const uint64_t n = sum(a->tr_cnt); if (n) g->tr_ks.assign(a->tr_ks, a->tr_ks + n); }
    if (a->tre_cnt) {
        g->tre_cnt.assign(a->tre_cnt, a->tre_cnt + a->nloci);
        const uint64_t n = sum(a->tre_cnt);
        if (n) g->tre_ks.assign(a->tre_ks, a->tre_ks + n);
    }
    if (a->qc) g->qc.assign(a->qc, a->qc + a->nloci);
    if (a->bt_cnt) {
        g->bt_cnt.assign(a->bt_cnt, a->bt_cnt + a->nloci);
        const uint64_t n = sum(a->bt_cnt);
        if (n) { g->bt_ks.assign(a->bt_ks, a->bt_ks + n); g->bt_vs.assign(a->bt_vs, a->bt_vs + n); }
    }
    if (a->gr_cnt) {
        g->gr_cnt.assign(a->gr_cnt, a->gr_cnt + a->nloci);
        const uint64_t n = sum(a->gr_cnt);
        if (n) { g->gr_ks.assign(a->gr_ks, a->gr_ks + n); g->gr_ms.assign(a->gr_ms, a->gr_ms + n); }
    }
    const dbtk_status_t st = dbtk::finish_rpgg(g.get());
    if (st) return st;
    *out = g.release();
    return DBTK_OK;
}

// writeCigar / writeAnnot (src/aQueryFasta_thread.cpp:1683-1740) on a compact alignment record (dbtk.h: dbtk_aln_hdr_t
// + es1 tr1 es2 tr2), printed in writeAlignments' order (AQ.cpp:1751-1757): cigar2 annot2 cigar1 annot1, tab-separated.
size_t dbtk_aln_format(const void* rec, uint32_t cap, char* out, size_t out_cap) {
    const dbtk_aln_hdr_t* h = (const dbtk_aln_hdr_t*)rec;
    const uint8_t* base = (const uint8_t*)rec + sizeof(dbtk_aln_hdr_t);
    size_t n = 0;
    auto putc_ = [&](char c) { if (n + 1 < out_cap) out[n] = c; ++n; };
    auto puti = [&](int v) { char b[16]; const int l = snprintf(b, sizeof b, "%d", v); for (int i = 0; i < l; ++i) putc_(b[i]); };
    static const char TT[8] = {'*', '=', 'X', 'D', 'I', '?', '?', '?'};
    static const char GG[8] = {0, 'A', 'C', 'G', 'T', '*', '?', '?'};
    auto cigar = [&](const uint8_t* es, int sz) {  // writeCigar
        if (!sz) { putc_('*'); return; }
        int ct = 1;
        char t0 = TT[es[0] & 7], g0 = GG[(es[0] >> 3) & 7], t1 = 0, g1 = 0;
        for (int i = 1; i < sz; ++i) {
            t1 = TT[es[i] & 7]; g1 = GG[(es[i] >> 3) & 7];
            if (t0 == '=' || t0 == '.' || t0 == '*') {
                while (t1 == t0) {
                    ++ct; ++i;
                    if (i == sz) break;
                    t1 = TT[es[i] & 7]; g1 = GG[(es[i] >> 3) & 7];
                }
                puti(ct); putc_(t0);
            } else if (t0 == 'X') { putc_('X'); putc_(g0); }
            else if (t0 == 'D') {
                if (t1 == 'I') { putc_('X'); putc_(g0); ++i; }  // ins + del printed as a mismatch
                else { putc_('D'); putc_(g0); }
            } else if (t0 == 'I') {
                if (t1 == 'D') { putc_('X'); putc_(g1); ++i; }
                else putc_('I');
            } else putc_(t0);
            if (i == sz) return;
            ct = 1;
            t0 = TT[es[i] & 7]; g0 = GG[(es[i] >> 3) & 7];
        }
        puti(ct); putc_(t0);
    };
    auto annot = [&](const uint8_t* tr, int sz) {  // writeAnnot
        if (!sz) { putc_('*'); return; }
        int ct = 1;
        uint8_t c0 = tr[0];
        for (int i = 1; i < sz; ++i) {
            if (c0 == '=' || c0 == '.' || c0 == '*') {
                while (tr[i] == c0) { ++ct; ++i; if (i == sz) break; }
                puti(ct); putc_((char)c0);
            } else putc_((char)c0);
            if (i == sz) return;
            ct = 1;
            c0 = tr[i];
        }
        puti(ct); putc_((char)c0);
    };
    auto clamp = [&](uint16_t v) { return (int)(v < cap ? v : cap); };
    cigar(base + 2 * (size_t)cap, clamp(h->nes2)); putc_('\t');
    annot(base + 3 * (size_t)cap, clamp(h->ntr2)); putc_('\t');
    cigar(base, clamp(h->nes1)); putc_('\t');
    annot(base + (size_t)cap, clamp(h->ntr1));
    if (out_cap) out[n < out_cap ? n : out_cap - 1] = 0;
    return n;
}

void dbtk_rpgg_free(dbtk_rpgg_t* h) { delete h; }
uint64_t dbtk_rpgg_uid(const dbtk_rpgg_t* h) { return h ? h->uid : 0; }
dbtk_status_t dbtk_rpgg_set_index_cache(dbtk_rpgg_t* h, const char* path, int mode) {
    if (!h || mode < 0 || mode > 2 || (mode && (!path || !*path))) { dbtk::set_error("dbtk_rpgg_set_index_cache: handle, path, mode 0 .. 2"); return DBTK_ERR_ARG; }
    return dbtk::guarded([&] { h->idx_cache = mode ? path : ""; h->idx_cache_mode = mode; return DBTK_OK; });
}
uint64_t dbtk_rpgg_nloci(const dbtk_rpgg_t* h) { return h ? h->nloci : 0; }
uint64_t dbtk_rpgg_ntrkmers(const dbtk_rpgg_t* h) { return h ? h->out_kmer.size() : 0; }
uint64_t dbtk_rpgg_nkeys(const dbtk_rpgg_t* h) { return h ? h->keys.size() : 0; }

dbtk_status_t dbtk_rpgg_view(const dbtk_rpgg_t* h, dbtk_rpgg_arrays_t* o) {
    if (!h || !o) { set_error("null argument"); return DBTK_ERR_ARG; }
    memset(o, 0, sizeof(*o));
    o->ksize = h->ksize; o->nloci = h->nloci;
    o->nkeys = h->keys.size(); o->keys = h->keys.data(); o->vals = h->vals.data();
    o->nvv = h->vv.size(); o->vv = h->vv.data();
    o->fl_cnt = h->fl_cnt.data(); o->fl_ks = h->fl_ks.data();
    if (!h->tre_cnt.empty()) { o->tre_cnt = h->tre_cnt.data(); o->tre_ks = h->tre_ks.data(); }
    o->tr_cnt = h->tr_cnt.data(); o->tr_ks = h->tr_ks.data();
    if (!h->qc.empty()) o->qc = h->qc.data();
    if (!h->bt_cnt.empty()) { o->bt_cnt = h->bt_cnt.data(); o->bt_ks = h->bt_ks.data(); o->bt_vs = h->bt_vs.data(); }
    if (!h->gr_cnt.empty()) { o->gr_cnt = h->gr_cnt.data(); o->gr_ks = h->gr_ks.data(); o->gr_ms = h->gr_ms.data(); }
    return DBTK_OK;
}

dbtk_status_t dbtk_rpgg_output_order(const dbtk_rpgg_t* h, uint64_t* out_slot) {
    if (!h || !out_slot) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (!h->out_slot.empty()) memcpy(out_slot, h->out_slot.data(), h->out_slot.size() * sizeof(uint64_t));
    return DBTK_OK;
}

void dbtk_params_default(dbtk_params_t* p) {  // src/aQueryFasta_thread.cpp:26-34, 2336-2339
    memset(p, 0, sizeof(*p));
    p->ksize = 21; p->n_filter = 4; p->nm_filter = 1; p->cthreshold = 10; p->nm_tr = 40; p->max_nt = 2; p->qth = 20;
    p->okam = 1;
}

// `ktools serialize` (src/kmertools.cpp:221-345).  The byte layout of the three outputs depends on the iteration order of
// the reference's std::unordered_map<size_t, size_t> / std::unordered_set<uint64_t>; the same containers filled in the
// same order iterate in the same order, so they are used as such.
static dbtk_status_t dbtk_rpgg_serialize_impl(const char* prefix) {
    if (!prefix) { set_error("null argument"); return DBTK_ERR_ARG; }
    const std::string pref(prefix);
    // text k-mer file -> (locus index, first field) per line, in file order
    auto for_each_kmer = [&](const std::string& fn, const std::function<void(uint64_t, uint64_t)>& fnc, uint64_t* nloci) -> dbtk_status_t {
        File f(fn, "rb");
        if (!f.f) { set_error("cannot open " + fn); return DBTK_ERR_IO; }
        fseek(f.f, 0, SEEK_END);
        const long sz = ftell(f.f);
        fseek(f.f, 0, SEEK_SET);
        std::vector<char> buf((size_t)sz + 1);
        if (sz && fread(buf.data(), 1, (size_t)sz, f.f) != (size_t)sz) { set_error("read failed: " + fn); return DBTK_ERR_IO; }
        buf[sz] = 0;
        uint64_t idx = ~0ull;  // ++ on every '>' line (the reference starts at -1)
        const char* p = buf.data();
        const char* end = p + sz;
        while (p < end) {
            const char* nl = (const char*)memchr(p, '\n', end - p);
            const char* e = nl ? nl : end;
            if (e > p) {
                if (*p == '>') ++idx;
                else fnc(idx, strtoull(p, nullptr, 10));  // stoul / stoull: the first field
            }
            p = nl ? nl + 1 : end;
        }
        if (nloci) *nloci = idx + 1;
        return DBTK_OK;
    };
    // ---- PREF.kmers.dbi (readKmerIndex on .tr.kmers then .fl.kmers, then the flattening of kmertools.cpp:240-280)
    std::unordered_map<size_t, size_t> kmerDBi;
    std::vector<std::vector<uint32_t>> vec;
    uint64_t nloci = 0;
    for (const char* ext : {".tr.kmers", ".fl.kmers"}) {
        uint32_t vsize = (uint32_t)vec.size();
        dbtk_status_t st = for_each_kmer(pref + ext, [&](uint64_t idx64, uint64_t kmer) {
            const uint32_t idx = (uint32_t)idx64;
            auto it = kmerDBi.find(kmer);
            if (it != kmerDBi.end()) {
                const uint32_t vi = (uint32_t)it->second;
                if (vi % 2) {
                    bool good = true;
                    for (uint32_t x : vec[vi >> 1]) if (x == idx) { good = false; break; }
                    if (good) vec[vi >> 1].push_back(idx);
                } else if ((vi >> 1) != idx) {
                    vec.push_back(std::vector<uint32_t>{vi >> 1, idx});
                    it->second = ((vsize++) << 1) + 1;
                }
            } else {
                kmerDBi[kmer] = (idx << 1);
            }
        }, ext[1] == 't' ? &nloci : nullptr);
        if (st) return st;
    }
    {
        std::vector<uint32_t> vv, vvi;
        for (auto& v : vec) { vvi.push_back((uint32_t)vv.size()); vv.push_back((uint32_t)v.size()); vv.insert(vv.end(), v.begin(), v.end()); }
        for (auto& p : kmerDBi) if (p.second % 2) p.second = ((size_t)vvi[p.second >> 1] << 1) + 1;
        const uint64_t nk = kmerDBi.size(), nvv = vv.size();
        std::vector<uint64_t> keys(nk);
        std::vector<uint32_t> vals(nk);
        uint64_t ki = 0;
        for (auto& p : kmerDBi) { keys[ki] = p.first; vals[ki] = (uint32_t)p.second; ++ki; }
        File f(pref + ".kmers.dbi", "wb");
        if (!f.f || !f.write(&nk, 1) || !f.write(keys.data(), nk) || !f.write(vals.data(), nk) || !f.write(&nvv, 1) || !f.write(vv.data(), nvv)) {
            set_error("cannot write " + pref + ".kmers.dbi"); return DBTK_ERR_IO;
        }
    }
    // ---- PREF.fl.kdb, PREF.tre.kdb: per locus an unordered_set, flattened in iteration order
    for (const char* tp : {"fl", "tre"}) {
        std::vector<std::unordered_set<uint64_t>> db(nloci);
        dbtk_status_t st = for_each_kmer(pref + "." + tp + ".kmers", [&](uint64_t idx, uint64_t kmer) { if (idx < nloci) db[idx].insert(kmer); }, nullptr);
        if (st) return st;
        std::vector<uint64_t> index(nloci), ks;
        for (uint64_t l = 0; l < nloci; ++l) { index[l] = db[l].size(); for (uint64_t km : db[l]) ks.push_back(km); }
        const uint64_t nk = ks.size();
        File f(pref + "." + tp + ".kdb", "wb");
        if (!f.f || !f.write(&nloci, 1) || !f.write(index.data(), nloci) || !f.write(&nk, 1) || !f.write(ks.data(), nk)) {
            set_error("cannot write " + pref + "." + tp + ".kdb"); return DBTK_ERR_IO;
        }
    }
    return DBTK_OK;
}

static dbtk_status_t dbtk_write_outputs_impl(const dbtk_rpgg_t* h, const uint64_t* counts, const uint64_t* kmc,
                                 const uint32_t* nmapread, const char* out_prefix, int with_names) {
    if (!h || !counts || !out_prefix) { set_error("null argument"); return DBTK_ERR_ARG; }
    const std::string pref(out_prefix);
    const uint64_t nk = h->out_kmer.size();
    if (with_names) {  // writeKmersWithName, src/aQueryFasta_thread.h:926-937
        File f(pref + ".tr.kmers", "wb");
        if (!f.f) { set_error("cannot create " + pref + ".tr.kmers"); return DBTK_ERR_IO; }
        for (uint64_t l = 0; l < h->nloci; ++l) {
            fprintf(f.f, ">%llu\n", (unsigned long long)l);
            for (uint64_t i = h->out_beg[l]; i < h->out_beg[l + 1]; ++i)
                fprintf(f.f, "%llu\t%llu\n", (unsigned long long)h->out_kmer[i], (unsigned long long)counts[i]);
        }
        return DBTK_OK;
    }
    {  // dumpTRKmers -> serializeKarray, src/aQueryFasta_thread.h:968-973, src/binaryKmerIO.hpp:179-188
        File f(pref + ".trkmc.ar", "wb");
        if (!f.f) { set_error("cannot create " + pref + ".trkmc.ar"); return DBTK_ERR_IO; }
        if (!f.write(&nk, 1) || !f.write(counts, nk)) { set_error("write failed"); return DBTK_ERR_IO; }
    }
    if (kmc && nmapread) {  // writeTRKmerSummary, src/aQueryFasta_thread.h:976-983
        File f(pref + ".tr.summary.txt", "wb");
        if (!f.f) { set_error("cannot create " + pref + ".tr.summary.txt"); return DBTK_ERR_IO; }
        for (uint64_t l = 0; l < h->nloci; ++l)
            fprintf(f.f, "%u\t%llu\n", nmapread[l], (unsigned long long)kmc[l]);
    }
    return DBTK_OK;
}

// ---- the entry points above that parse files or allocate host memory, behind the exception barrier (dbtk_internal.h: guarded)
// the sidecar of a handle loaded from files: PREF.dbtk.idx (PREF.<name of the TR file>.dbtk.idx for -t N: another output order, other images)
static dbtk_status_t with_cache(dbtk_status_t st, const char* prefix, const char* tr_kmers_file, dbtk_rpgg_t** out) {
    if (st || !out || !*out || !prefix) return st;
    int mode = 1;
    if (const char* e = getenv("DBTK_IDX_CACHE")) mode = atoi(e);
    if (mode < 0 || mode > 2) mode = 1;
    std::string path = std::string(prefix);
    if (tr_kmers_file) { std::string t(tr_kmers_file); const size_t sl = t.find_last_of('/'); path += "." + (sl == std::string::npos ? t : t.substr(sl + 1)); }
    (*out)->idx_cache = mode ? path + ".dbtk.idx" : "";
    (*out)->idx_cache_mode = mode;
    return st;
}
dbtk_status_t dbtk_rpgg_load(const char* prefix, uint32_t ksize, const char* qc_file, const char* bait_file,
                             uint32_t flags, dbtk_rpgg_t** out) {
    return dbtk::guarded([&] { return with_cache(dbtk_rpgg_load_impl(prefix, nullptr, ksize, qc_file, bait_file, flags, out), prefix, nullptr, out); });
}
dbtk_status_t dbtk_rpgg_load_tr(const char* prefix, const char* tr_kmers_file, uint32_t ksize, const char* qc_file, const char* bait_file,
                                uint32_t flags, dbtk_rpgg_t** out) {
    return dbtk::guarded([&] { return with_cache(dbtk_rpgg_load_impl(prefix, tr_kmers_file, ksize, qc_file, bait_file, flags, out), prefix, tr_kmers_file, out); });
}
dbtk_status_t dbtk_rpgg_from_arrays(const dbtk_rpgg_arrays_t* a, dbtk_rpgg_t** out) {
    return dbtk::guarded([&] { return dbtk_rpgg_from_arrays_impl(a, out); });
}
dbtk_status_t dbtk_rpgg_serialize(const char* prefix) {
    return dbtk::guarded([&] { return dbtk_rpgg_serialize_impl(prefix); });
}
dbtk_status_t dbtk_write_outputs(const dbtk_rpgg_t* h, const uint64_t* counts, const uint64_t* kmc,
                                 const uint32_t* nmapread, const char* out_prefix, int with_names) {
    return dbtk::guarded([&] { return dbtk_write_outputs_impl(h, counts, kmc, nmapread, out_prefix, with_names); });
}

}  // extern "C"
