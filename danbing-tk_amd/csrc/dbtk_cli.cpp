// dbtk_cli.cpp — `danbing-tk`-compatible command line over the C-ABI (include/dbtk.h).
//
// Drop-in for the process boundary of the reference's aligner
// (src/aQueryFasta_thread.cpp:2286-2660, flag table in SURVEY.md Appendix C):
// same flags with the same order sensitivity around -qs, the same RPGG files in,
// the same output files and stdout records out.  What runs between the reader
// and the writers is the HIP hot path (libdbtk_hip.so); this file is host glue:
//   * argv loop                                   AQ.cpp:2344-2429
//   * reader + on-the-fly mate pairing            AQ.cpp:1918-1976 (critical section A)
//   * kam / extracted-read writers                AQ.cpp:1618-1681 (critical section B)
//   * totals, dumps                               AQ.cpp:2617-2656
// Ingest: the three stages overlap — one thread parses and pairs (no per-line strings: spans of
// the read buffer go straight into the batch's flat arrays), one thread per GPU aligns, the main thread formats and
// writes the records in batch order.
// New flags live under their own namespace: --gpus N (GPUs to use, default 1).
// stderr is informational (the reference's also carries timings); stdout and the
// output files are byte-compatible.
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <malloc.h>
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/dbtk.h"

namespace {

[[noreturn]] void die_assert(const std::string& what) {
    // the reference `assert`s on unusable files (abort, exit status 134)
    fprintf(stderr, "danbing-tk: %s\n", what.c_str());
    abort();
}

struct Opts {
    bool bait = false, aug = false, threading = false, tc = false, aln = false, aln_minimal = false, okam = true, g2pan = false;
    bool writeKmerName = false, outputBubbles = false, invkmer = false, isFastq = false, trackBait = false, qc = false;
    int simmode = 0, extractFastX = 0, verbosity = 0, ngpus = 1, gzLevel = 1, emitThreads = 0, ingestShards = 0, alnAligners = 0;
    bool correction = true;
    bool hostIngest = false; // --host-ingest: parse and pair on the host even where the device reader applies
    bool writeIdxCache = false;  // --write-idx-cache: leave PREF.dbtk.idx (the GPU-layout index images) next to the RPGG for the next run
    bool parseOnly = false;  // --parse-only: run the ingest (reader, splitters, pairing) and report what it handed on; no GPU
    bool v13 = false;       // --v13-threading: -g/-gc/-gcc run the graph walk of the v1.3 contract instead of HEAD's dead path
    std::string alnGz;      // --aln-gz FILE: the -a / -ae records gzip-compressed into FILE (instead of plain on stdout)
    uint64_t trim = 0, thread_cth = 100, Cthreshold = 10, nproc = 1, ksize = 21, qth = 20, N_FILTER = 4, NM_FILTER = 1, NM_TR = 40,
             MAX_NT = 2, maxncorrection = 4;
    float readsPerBatchFactor = 1;
    std::string trPrefix, trFname, fastxFname, outPrefix, qcFn, baitFname;
};

bool readable(const std::string& fn) {
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

void usage() {
    fprintf(stderr,
            "\nUsage: danbing-tk [-bu] [-ka] [-qc] [-k] [-kf] [-cth] [-qth] [-b] [-c] [-r] [-p] <-fa|-fq> -qs <-o|-on>\n"
            "  (MI355X build: the alignment hot path runs on the GPU through libdbtk_hip.so)\n"
            "Input:\n"
            "  -fa <STR> | -fq <STR>  paired reads as FASTA | FASTQ (e.g. samtools fasta/fastq -n); mates are paired on the fly\n"
            "  -qs <STR>              prefix of the RPGG files (PREF.tr.kmers, PREF.kmers.dbi, PREF.fl.kdb, PREF.tre.kdb)\n"
            "Output:\n"
            "  -o <STR> | -on <STR>   output prefix (OUT.trkmc.ar + OUT.tr.summary.txt | OUT.tr.kmers with names)\n"
            "  -ka                    no k-mer assignment (kam) records on stdout\n"
            "  -bu                    write read (k+1)-mers absent from the graph\n"
            "Algorithm:\n"
            "  -k <INT> [21]  -kf <N> <M> [4 1]  -cth <INT> [10]  -c <INT> [40]  -qth <INT> [20]  -qc <FILE>  -b [FILE]\n"
            "Execution:\n"
            "  -p <INT>  -r <FLOAT>   accepted for compatibility (threads / batch factor: batch = 300000*r reads)\n"
            "  --gpus <INT>           GPUs to spread batches over [1]\n"
            "  --v13-threading        -g/-gc/-gcc <INT> [INT] walk both mates through PREF.graph.kmers (or PREF.graph.umap) with error\n"
            "                         correction as danbing-tk v1.3 did (at this HEAD the reference leaves those flags dead: all-zero output);\n"
            "                         TR k-mers are then counted in \"exact\" mode and -a / -ae print alignment records on stdout\n"
            "  --aln-gz <FILE>        write the -a / -ae records gzip-compressed to FILE (deflated on --emit-threads host threads\n"
            "                         while the GPU works on the next batches) instead of plain text on stdout\n"
            "  --ingest-shards <INT>  cut a seekable input file into this many byte ranges, each read, split and paired by its own\n"
            "                         pipeline with its own context (tables are shared per GPU) [the number of GPUs]\n"
            "  --host-ingest          parse and pair the reads on the host even where the device reader applies (a regular file without\n"
            "                         -s / -tb / -a / -ae: there the host only copies bytes and kernels find the records and pair the mates)\n"
            "  --emit-threads <INT>   host threads formatting / compressing records [hardware threads / 4, at most the usable CPUs]\n"
            "  --aln-aligners <INT>   with -a / -ae: aligner threads (each with its own context) per GPU, so that fetching and formatting one\n"
            "                         batch's records overlaps the next batches' kernels [4]\n"
            "  --gz-level <INT>       zlib level of --aln-gz [1: measured 40x less deflate time than gzip's default 6 for 16 %% more bytes]\n"
            "Developer:\n"
            "  -s <1|2>  -e <1|2>  -v <INT>  -g|-gc|-gcc <INT> [INT]  -a  -ae  -tb  -ik  -t <INT>  -m <FILE>  -au\n\n");
}

// ---- reader: AQ.cpp:1918-1976 -------------------------------------------------------------------
struct Reader {
    FILE* f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    bool fill() {
        if (eof) return false;
        if (pos < end) memmove(buf.data(), buf.data() + pos, end - pos);
        end -= pos; pos = 0;
        if (buf.size() - end < (1u << 20)) buf.resize(buf.size() * 2);
        const size_t n = fread(buf.data() + end, 1, buf.size() - end, f);
        if (n == 0) { eof = true; return false; }
        end += n;
        return true;
    }
    // std::getline semantics: false only when no characters are left
    bool getline(std::string& out) {
        for (;;) {
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) { out.assign(buf.data() + pos, nl - (buf.data() + pos)); pos = nl - buf.data() + 1; return true; }
            if (!fill()) {
                if (pos == end) { out.clear(); return false; }
                out.assign(buf.data() + pos, end - pos); pos = end;
                return true;
            }
        }
    }
    // the same line as a span of the buffer (valid until the next call)
    bool getspan(const char** p, size_t* n) {
        for (;;) {
            const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) { *p = buf.data() + pos; *n = nl - (buf.data() + pos); pos = nl - buf.data() + 1; return true; }
            if (!fill()) {
                *p = buf.data() + pos; *n = end - pos; pos = end;
                return *n != 0;
            }
        }
    }
    bool at_eof() { return pos == end && !fill(); }  // in->peek() == EOF
};

// newlines in [p, p + n): the one full pass over the input on a single thread (the cutter), so it gets the wide registers where the CPU has them
__attribute__((target("avx2"))) static size_t count_nl_avx2(const char* p, size_t n) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t c = 0, i = 0;
    for (; i + 128 <= n; i += 128) {
        const uint32_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nl));
        const uint32_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 32)), nl));
        const uint32_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 64)), nl));
        const uint32_t m3 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 96)), nl));
        c += (size_t)__builtin_popcountll(((uint64_t)m1 << 32) | m0) + (size_t)__builtin_popcountll(((uint64_t)m3 << 32) | m2);
    }
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
static size_t count_nl(const char* p, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? count_nl_avx2(p, n) : (size_t)std::count(p, p + n, '\n');
}

inline void prunePEinfo(std::string& title) {  // AQ.cpp:455-462
    const size_t len = title.size();
    if (len >= 2 && title[len - 2] == '/' && (title[len - 1] == '1' || title[len - 1] == '2')) title.resize(len - 2);
}

struct Batch {  // one batch on its way through the stages; read r = flat[off[r], off[r+1]); reads 2p, 2p+1 = pair p
    uint64_t index = 0, nreads = 0, nReads_so_far = 0;
    size_t nparked = 0;
    std::vector<uint8_t> flat;  std::vector<uint64_t> off;   // sequences
    std::vector<char> qar;      std::vector<uint64_t> qoff;  // qualities as read (FASTQ)
    std::vector<char> tar;      std::vector<uint64_t> toff;  // pruned titles, one per pair
    std::vector<uint64_t> src;                               // simmode: source locus per pair
    std::vector<dbtk_pair_rec_t> recs;
    uint64_t nrec = 0;
    struct RawBuf {  // bytes without the zero fill of vector::resize (a batch's records: >100 MB)
        std::unique_ptr<uint8_t[]> p; size_t cap = 0;
        uint8_t* data() const { return p.get(); }
        size_t size() const { return cap; }
        void grow(size_t n) { if (n > cap) { p.reset(new uint8_t[n]); cap = n; } }
    } aln;                      // -a / -ae: the batch's alignment records in text form (dbtk_ctx_aln_text): the arena ...
    std::vector<uint32_t> aln_idx;  // ... and where pair p's record starts in it (DBTK_NAN32: none)
    uint64_t naln = 0;
    std::vector<std::string> aln_chunks;  // -a / -ae: the batch's alignment lines, formatted (and deflated) by the emit pool
    std::vector<uint32_t> aln_em;         //            the pairs that have a record, in pair order
    long gpu_sec = 0;
    // a batch the device reader made (dbtk_ingest_*): titles, reads and qualities stay where they were read, in the slot's pinned
    // block, and the per-pair spans say where
    const char* blk = nullptr;
    std::vector<dbtk_ingest_span_t> spans;
    const char* aln_dev = nullptr; uint64_t aln_dev_bytes = 0;  // -a / -ae lines the device made (text or gzip members), in the slot's pinned buffer
    typedef std::pair<const char*, size_t> Span;
    Span title_s(uint64_t p) const { return blk ? Span(blk + spans[p].title, spans[p].title_len) : Span(tar.data() + toff[p], toff[p + 1] - toff[p]); }
    Span seq_s(uint64_t r) const { return blk ? Span(blk + spans[r >> 1].seq[r & 1], spans[r >> 1].seq_len[r & 1]) : Span((const char*)flat.data() + off[r], off[r + 1] - off[r]); }
    Span qual_s(uint64_t r) const { return blk ? Span(blk + spans[r >> 1].qual[r & 1], spans[r >> 1].qual_len[r & 1]) : Span(qar.data() + qoff[r], qoff[r + 1] - qoff[r]); }
    std::string title(uint64_t p) const { const Span t = title_s(p); return std::string(t.first, t.second); }
    void add_read(const char* sp, size_t sn, const char* qp, size_t qn, bool fq) {
        flat.insert(flat.end(), sp, sp + sn); off.push_back(flat.size());
        if (fq) { qar.insert(qar.end(), qp, qp + qn); qoff.push_back(qar.size()); }
    }
};

// bounded hand-over between two stages
template <class T>
struct Chan {
    std::mutex m; std::condition_variable cv;
    std::deque<T> q; size_t cap = 4; bool closed = false;
    void push(T v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cv.notify_all(); }
    bool pop(T& v) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front()); q.pop_front(); cv.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv.notify_all(); }
};

uint64_t parse_src(const std::string& title, int simmode, uint64_t nloci) {
    if (simmode == 1) {  // >LOCUS.xxx   (AQ.cpp:478-489)
        const size_t first = title.find('.');
        return strtoull(title.substr(1, first).c_str(), nullptr, 10);
    }
    // simmode 2: >CHR:START-END:LOCUS  (AQ.cpp:492-506)
    const size_t p1 = title.find(':'), p2 = title.find(':', p1 + 1);
    const std::string val = title.substr(p2 + 1);
    if (!val.empty() && val[0] == '.') return nloci;
    return strtoull(val.c_str(), nullptr, 10);
}

// ---- writers: AQ.cpp:1618-1681 --------------------------------------------------------------------
std::string annot2str(const dbtk_mate_rec_t& m) {  // km_asgn_t::annot2str_, AQ.cpp:121-138
    static const char chs[3] = {'*', '.', '='};
    if (m.nk == 0) return "*";
    auto st = [&](int i) { return (m.as2[i >> 2] >> (2 * (i & 3))) & 3; };
    std::string s;
    int ct = 1, a0 = st(0);
    for (int i = 1; i < m.nk; ++i) {
        const int a1 = st(i);
        if (a0 != a1) { s += std::to_string(ct) + chs[a0]; ct = 1; }
        else ++ct;
        a0 = a1;
    }
    s += std::to_string(ct) + chs[a0];
    return s;
}
std::string na(int v) { return v == -1 ? std::string(".") : std::to_string(v); }
void mate_fields(std::string& o, const dbtk_mate_rec_t& r) {
    o += std::to_string(r.kf) + ':' + std::to_string(r.hf) + ':' + std::to_string(r.bf) + ':' + std::to_string(r.qf) + ':' +
         std::to_string(r.af) + ':' + std::to_string(r.rm) + ":0:0:" + na(r.si) + ':' + na(r.nt) + ':' + na(r.bs) + ':' + na(r.ti);
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 2) { usage(); return 0; }
    double t_main; { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); t_main = t.tv_sec + 1e-9 * t.tv_nsec; }
    // big blocks come from the heap and stay there: batch buffers of tens of megabytes are grown and recycled all the time, and a
    // map / unmap per growth serialises every thread of the process on the kernel's memory-map lock
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);
    std::vector<std::string> args(argv, argv + argc);
    Opts o;
    auto need = [&](size_t i) -> const std::string& {
        if (i >= args.size()) die_assert("missing value after " + args[i - 1]);  // the reference reads past argv here
        return args[i];
    };
    for (size_t argi = 1; argi < args.size(); ++argi) {  // order-sensitive like AQ.cpp:2344-2429
        const std::string& a = args[argi];
        if (a == "-b") {
            o.bait = true;
            if (need(argi + 1)[0] != '-') o.baitFname = args[++argi];
        }
        else if (a == "-v") o.verbosity = atoi(need(++argi).c_str());
        else if (a == "-e") o.extractFastX = atoi(need(++argi).c_str());
        else if (a == "-bu") o.outputBubbles = true;
        else if (a == "-t") o.trim = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "-s") o.simmode = atoi(need(++argi).c_str());
        else if (a == "-m") { o.g2pan = true; if (!readable(need(++argi))) die_assert("cannot open " + args[argi]); }
        else if (a == "-au") o.aug = true;
        else if (a == "-g" || a == "-gc" || a == "-gcc") {
            o.threading = true;
            o.correction = a != "-g";  // usage text (AQ.cpp:2325-2326): -g walks without error correction, -gc / -gcc with
            if (a == "-gcc") o.tc = true;
            o.thread_cth = strtoull(need(++argi).c_str(), nullptr, 10);
            if (need(argi + 1)[0] != '-') o.maxncorrection = strtoull(args[++argi].c_str(), nullptr, 10);
        }
        else if (a == "-a") o.aln = true;
        else if (a == "-ae") { o.aln = true; o.aln_minimal = true; }
        else if (a == "-ka") o.okam = false;
        else if (a == "-kf") { o.N_FILTER = strtoull(need(++argi).c_str(), nullptr, 10); o.NM_FILTER = strtoull(need(++argi).c_str(), nullptr, 10); }
        else if (a == "-r") o.readsPerBatchFactor = strtof(need(++argi).c_str(), nullptr);
        else if (a == "-c") o.NM_TR = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "-ik") o.invkmer = true;
        else if (a == "-k") o.ksize = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "-tb") o.trackBait = true;
        else if (a == "-qc") { o.qc = true; o.qcFn = need(++argi); if (!readable(o.qcFn)) die_assert("cannot open " + o.qcFn); }
        else if (a == "-qs") {
            o.trPrefix = need(++argi);
            o.trFname = o.trim ? o.trPrefix + ".tr.trim" + std::to_string(o.trim) + ".kmers" : o.trPrefix + ".tr.kmers";
            if (!readable(o.trFname)) die_assert("cannot open " + o.trFname);
            if (o.aug && !readable(o.trPrefix + ".tr.aug.kmers")) die_assert("cannot open " + o.trPrefix + ".tr.aug.kmers");
            if (o.bait) {
                if (o.baitFname.empty()) o.baitFname = o.trPrefix + (o.qc ? ".qc.bt.kmdb" : ".bt.kmdb");
                if (!readable(o.baitFname)) die_assert("cannot open " + o.baitFname);
            }
        }
        else if (a == "-fa" || a == "-fq") {
            o.isFastq = a == "-fq";
            o.fastxFname = need(++argi);
            if (!readable(o.fastxFname)) die_assert("cannot open " + o.fastxFname);
        }
        else if (a == "-o" || a == "-on") {
            o.writeKmerName = a == "-on";
            o.outPrefix = need(++argi);
            FILE* f = fopen((o.outPrefix + ".trkmc.ar").c_str(), "wb");  // truncated at parse time, AQ.cpp:2417
            if (!f) die_assert("cannot create " + o.outPrefix + ".trkmc.ar");
            fclose(f);
        }
        else if (a == "-p") o.nproc = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "-cth") o.Cthreshold = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "-qth") o.qth = strtoull(need(++argi).c_str(), nullptr, 10);
        else if (a == "--gpus") o.ngpus = atoi(need(++argi).c_str());
        else if (a == "--v13-threading") o.v13 = true;
        else if (a == "--parse-only") o.parseOnly = true;
        else if (a == "--host-ingest") o.hostIngest = true;
        else if (a == "--write-idx-cache") o.writeIdxCache = true;
        else if (a == "--aln-gz") o.alnGz = need(++argi);
        else if (a == "--emit-threads") o.emitThreads = atoi(need(++argi).c_str());
        else if (a == "--ingest-shards") o.ingestShards = atoi(need(++argi).c_str());
        else if (a == "--gz-level") o.gzLevel = atoi(need(++argi).c_str());
        else if (a == "--aln-aligners") o.alnAligners = atoi(need(++argi).c_str());
        else {
            fprintf(stderr, "invalid option: %s\n", a.c_str());
            abort();  // the reference does `throw;` with no active exception -> std::terminate
        }
    }
    // DBTK_V13_THREADING=1 in the environment = --v13-threading: the README's command line for the v1.3 contract
    // (`danbing-tk -gc 85 3 -ae ...`, README.md:38-39) then runs unchanged
    if (const char* e = getenv("DBTK_V13_THREADING")) if (atoi(e) != 0) o.v13 = true;

    fprintf(stderr,
            "use baitDB: %d\nextract fastX: %d\noutput bubbles: %d\nis Fastq: %d\nsim mode: %d\ngraph threading mode: %d\n"
            "output kmer assignment (kam): %d\nk: %llu\n# of subsampled kmers in pre-filtering: %llu\n"
            "minimal # of matches in pre-filtering: %llu\nCthreshold: %llu\nmin # of kmer matches for TR spanning read: %llu\n"
            "fastx: %s\nquery: %s.(tr/ntr).kmers\nGPUs: %d\n\n",
            o.bait, o.extractFastX, o.outputBubbles, o.isFastq, o.simmode, o.threading, o.okam, (unsigned long long)o.ksize,
            (unsigned long long)o.N_FILTER, (unsigned long long)o.NM_FILTER, (unsigned long long)o.Cthreshold,
            (unsigned long long)o.NM_TR, o.fastxFname.c_str(), o.trPrefix.c_str(), o.ngpus);

    // ---- load (AQ.cpp:2459-2504)
    time_t time1 = time(nullptr);
    auto wall = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    const double tl0 = wall();
    // (the HIP runtime's start — a few tenths of a second — beside the parsing of the RPGG files)
    // (every device the run will use — logical GPU i on device map[i % n], the same rule as dev_of below — each warmed once; a failure is
    // reported when the thread is joined, before the first context is created: ADVICE r5)
    // host threads worth starting: the hardware threads, capped by what the container may actually use (its CPU affinity and its
    // cgroup CPU quota: a 256-thread host with cpu.max = 16 cores runs 64 deflate threads four times slower each)
    auto usable_cpus = []() -> unsigned {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[64]; unsigned long long per = 0;
            if (fscanf(f, "%63s %llu", q, &per) == 2 && per && strcmp(q, "max") != 0) n = std::min<unsigned>(n, (unsigned)std::max(1ull, (strtoull(q, nullptr, 10) + per - 1) / per));
            fclose(f);
        } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            long long quota = -1, per = 100000;
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &per) != 1) per = 100000; fclose(f2); }
            if (quota > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)((quota + per - 1) / per));
        }
        return n;
    };
    const unsigned cpus = usable_cpus();
    // the shape of a device reader's ring (run_device_ingest below; also what the warm-up thread pins ahead of it)
    auto ingest_chunk = [] { size_t CH = 32u << 20; if (const char* e = getenv("DBTK_INGEST_CHUNK")) { const long v = atol(e); if (v >= 4096) CH = (size_t)v; } return CH; };  // (tests: many small blocks)
    auto ingest_readers = [&](bool piped, int npipes_) {
        // reader threads: pread is a memcpy out of the page cache into a pinned buffer — 1.3 to 4 GB/s per thread, against 30 GB/s and more
        // that the GPU parses — so three quarters of the CPUs this process may use read (round 5: at most 8, which capped a first pass
        // over a 10-GB file at 15 GB/s), never fewer than one nor more than 24
        int nio_want = piped ? 1 : (int)std::max(1u, std::min(24u, cpus * 3 / 4 / (unsigned)std::max(1, npipes_)));
        if (const char* e = getenv("DBTK_INGEST_READERS")) { const int v = atoi(e); if (v > 0 && !piped) nio_want = v; }
        return nio_want;
    };
    auto ingest_slots = [&](int nio_want) {
        uint32_t NS = (uint32_t)std::max(12, nio_want + 6);  // chunks on their way at once (being read, copied, parsed, or waiting for their records
                                                             // to be written): the readers run this many chunks ahead of the oldest one not yet done with
        if (const char* e = getenv("DBTK_INGEST_SLOTS")) { const int v = atoi(e); if (v >= 2 && v <= 64) NS = (uint32_t)v; }
        return NS;
    };
    std::vector<int> warm_devs;
    {
        std::vector<int> wmap;
        if (const char* e = getenv("DBTK_DEVICE_MAP")) { std::string v(e); size_t at = 0; while (at < v.size()) { wmap.push_back(atoi(v.c_str() + at)); at = v.find(',', at); if (at == std::string::npos) break; ++at; } }
        for (int i = 0; i < (o.ngpus < 1 ? 1 : o.ngpus); ++i) {
            const int d = wmap.empty() ? i : wmap[(size_t)i % wmap.size()];
            if (std::find(warm_devs.begin(), warm_devs.end(), d) == warm_devs.end()) warm_devs.push_back(d);
        }
    }
    std::string warm_err;
    double pin_s = 0;
    uint32_t pin_n = 0;
    std::thread warm([&] {
        if (o.parseOnly) return;
        for (int d : warm_devs)
            if (dbtk_device_warmup(d) != DBTK_OK && warm_err.empty()) warm_err = std::string("device ") + std::to_string(d) + ": " + dbtk_last_error();
        if (!warm_err.empty()) return;
        // the device reader's pinned buffers, beside the parsing of the RPGG files: pinned inside the batch loop (18 buffers of 33 MB, by a dozen
        // reader threads at once) they took the loop's first 0.1 s, and every other allocation of those first blocks waited behind them.
        // As many as the input has chunks, at most the ring (a pipe's length is not known: its one reader pins as it goes)
        struct stat sb;
        if (o.hostIngest || o.simmode || (o.trackBait && o.bait) || getenv("DBTK_NO_PIN_AHEAD") || stat(o.fastxFname.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return;
        if (const char* e = getenv("DBTK_DEVICE_INGEST")) if (atoi(e) == 0) return;
        const bool aln = o.v13 && o.threading && !o.extractFastX && o.aln;
        if (aln && o.gzLevel != 1) return;
        const size_t CH = ingest_chunk();
        uint64_t min_size = 64u << 20;
        if (const char* e = getenv("DBTK_SHARD_MIN")) min_size = strtoull(e, nullptr, 10);
        const int want = std::max(o.ngpus < 1 ? 1 : o.ngpus, o.ingestShards);
        const int np = (want > 1 && (uint64_t)sb.st_size > min_size) ? want : 1;
        const uint32_t NS = ingest_slots(ingest_readers(false, np));
        const uint64_t per = ((uint64_t)sb.st_size / (uint64_t)np + CH - 1) / CH + 1;
        const uint32_t n = (uint32_t)std::min<uint64_t>(64, (uint64_t)np * std::min<uint64_t>(NS, per));
        // (a block's lines: its bytes and a record per pair as text, a third of that as gzip members)
        const uint64_t lines = !aln ? 0 : !o.alnGz.empty() ? CH / 2 : CH + CH / 4 + (1u << 20);
        const double t0 = wall();
        if (dbtk_ingest_reserve_host(warm_devs[0], CH, n, lines) != DBTK_OK) { fprintf(stderr, "pinning ahead failed (%s): the reader pins as it goes\n", dbtk_last_error()); return; }
        pin_s = wall() - t0; pin_n = n;
    });
    struct WarmJoin { std::thread& t; ~WarmJoin() { if (t.joinable()) t.join(); } } warm_join{warm};
    dbtk_rpgg_t* rpgg = nullptr;
    const bool use_bait = o.bait && !o.extractFastX && !o.threading;  // baitDB is only read and used on that path
    const bool walk = o.v13 && o.threading && !o.extractFastX;          // the graph walk of the v1.3 contract (AQ.cpp:2072-2088)
    const bool emit_aln = walk && o.aln;                                 // -a / -ae records (AQ.cpp:2232-2248)
    if (dbtk_rpgg_load_tr(o.trPrefix.c_str(), o.trim ? o.trFname.c_str() : nullptr, (uint32_t)o.ksize, o.qc ? o.qcFn.c_str() : nullptr,
                       use_bait ? o.baitFname.c_str() : nullptr,
                       (o.extractFastX ? DBTK_LOAD_INDEX_ONLY : 0) | (walk ? DBTK_LOAD_GRAPH : 0), &rpgg))
        die_assert(dbtk_last_error());
    const uint64_t nloci = dbtk_rpgg_nloci(rpgg);
    if (o.writeIdxCache) {  // (the file name follows dbtk_rpgg_load's: PREF[.name of the -t file].dbtk.idx)
        std::string cp = o.trPrefix;
        if (o.trim) { const size_t sl = o.trFname.find_last_of('/'); cp += "." + (sl == std::string::npos ? o.trFname : o.trFname.substr(sl + 1)); }
        if (dbtk_rpgg_set_index_cache(rpgg, (cp + ".dbtk.idx").c_str(), 2)) die_assert(dbtk_last_error());
    }
    fprintf(stderr, "total number of loci in %s: %llu\n", o.trFname.c_str(), (unsigned long long)nloci);
    fprintf(stderr, "deserialized graph/index and read tr.kmers in %ld sec.\n# unique kmers in kmerDBi: %llu\n", (long)(time(nullptr) - time1),
            (unsigned long long)dbtk_rpgg_nkeys(rpgg));

    dbtk_params_t P;
    dbtk_params_default(&P);
    P.ksize = (uint32_t)o.ksize; P.n_filter = (uint32_t)o.N_FILTER; P.nm_filter = (uint32_t)o.NM_FILTER;
    P.cthreshold = (uint32_t)(uint16_t)o.Cthreshold;  // uint16_t in the reference (AQ.cpp:1765)
    P.nm_tr = (uint32_t)o.NM_TR; P.max_nt = (uint32_t)o.MAX_NT; P.qth = (uint32_t)o.qth;
    P.okam = o.okam; P.qc = o.qc; P.extract = (uint32_t)o.extractFastX; P.simmode = (uint32_t)o.simmode;
    P.threading = walk ? DBTK_THREADING_V13 : (o.threading ? DBTK_THREADING_HEAD : 0);
    P.thread_cth = (uint32_t)o.thread_cth; P.maxncorrection = (uint32_t)o.maxncorrection;
    P.correction = o.correction;
    // CIGAR / annotation strings are written by the GPU: a few tens of bytes per pair come back.  In simulation mode (-s) -ae keeps a
    // walked pair when its source OR its destination is a locus (AQ.cpp:2241-2247): the device then makes a record for every walked
    // pair (as for -a; dst = nloci where threading removed the pair) and the host applies that rule
    P.aln = emit_aln ? (((o.aln_minimal && !o.simmode) ? 2u : 1u) | DBTK_ALN_TEXT) : 0;
    P.trackbait = (o.trackBait && use_bait) ? 1 : 0;  // -tb only does something inside the bait filter (AQ.cpp:2111-2119)
    P.bait = use_bait;
    P.bubbles = o.outputBubbles && !o.extractFastX && !o.threading;  // countNovelEdges only runs on the assignment path
    if (o.ngpus < 1) o.ngpus = 1;
    // DBTK_DEVICE_MAP=a,b,...: logical GPU i runs on device map[i % n] (tests: `--gpus 2` on a one-GPU box, DBTK_DEVICE_MAP=0,0 — two
    // contexts, two sharded readers, the cross-range pairing and the host merge all execute)
    std::vector<int> devmap;
    if (const char* e = getenv("DBTK_DEVICE_MAP")) { std::string v(e); size_t at = 0; while (at < v.size()) { devmap.push_back(atoi(v.c_str() + at)); at = v.find(',', at); if (at == std::string::npos) break; ++at; } }
    auto dev_of = [&](int i) { return devmap.empty() ? i : devmap[(size_t)i % devmap.size()]; };
    std::vector<dbtk_ctx_t*> ctx(o.ngpus, nullptr);
    if (warm.joinable()) warm.join();
    if (!warm_err.empty()) die_assert("GPU warm-up failed: " + warm_err);
    const double tl1 = wall();
    // -a / -ae with one range: several aligner threads per GPU, each with a context of its own (they share the GPU's tables; their
    // accumulators are summed on the host at the end): while one fetches its batch's records and has them formatted and
    // deflated, the others keep the GPU busy — the emit then costs the batch loop next to nothing.  Created on a thread of their own
    // beside the first context of their GPU: dbtk_ctx_create allocates what a context owns before it waits for the shared tables.
    // (Whether the input is cut into ranges — in which case they are not used — is known later: then they are freed unused.)
    const int aln_aligners = emit_aln ? (o.alnAligners > 0 ? o.alnAligners : 4) : 1;
    std::vector<dbtk_ctx_t*> extra_ctx;
    std::string extra_err;
    std::thread extra;
    if (!o.parseOnly && emit_aln && o.ngpus == 1 && o.ingestShards <= 1 && aln_aligners > 1)
        extra = std::thread([&] {
            for (int i = o.ngpus; i < aln_aligners * o.ngpus; ++i) {
                dbtk_ctx_t* c = nullptr;
                if (dbtk_ctx_create(rpgg, &P, dev_of(i % o.ngpus), &c)) { extra_err = dbtk_last_error(); return; }
                extra_ctx.push_back(c);
            }
        });
    struct ExtraJoin { std::thread& t; ~ExtraJoin() { if (t.joinable()) t.join(); } } extra_join{extra};
    if (!o.parseOnly)
        for (int d = 0; d < o.ngpus; ++d)
            if (dbtk_ctx_create(rpgg, &P, dev_of(d), &ctx[d])) die_assert(dbtk_last_error());
    fprintf(stderr, "load: RPGG files %.2f s, tables in HBM %.2f s\n", tl1 - tl0, wall() - tl1);
    if (pin_n && getenv("DBTK_VERBOSE")) fprintf(stderr, "pinned ahead: %u chunk buffers in %.3f s (beside the RPGG files)\n", pin_n, pin_s);
    if (!o.parseOnly && ctx[0]) {  // what the RPGG occupies on a GPU, table by table
        const char* nm[16]; uint64_t tb[16];
        const int nt = dbtk_ctx_table_bytes(ctx[0], nm, tb, 16);
        std::string line = "tables:";
        char buf[96];
        for (int i = 0; i < nt; ++i) {
            if (!strcmp(nm[i], "index_images:from_cache")) { if (tb[i]) line += " (images from the sidecar)"; continue; }
            if (!tb[i]) continue;
            snprintf(buf, sizeof buf, " %s %.1f MB", nm[i], tb[i] / 1e6);
            line += buf;
        }
        fprintf(stderr, "%s\n", line.c_str());
    }
    // --parse-only: what the pairing stage handed to the aligner stage: pairs, bases, and an order-independent digest
    // of (title, seq1, seq2[, qual1, qual2]) per pair (FNV-1a per pair, summed), for the ingest tests (no GPU needed)
    std::atomic<uint64_t> po_pairs{0}, po_bases{0}, po_digest{0};

    // ---- the batch loop (AQ.cpp:1869-2282) as three overlapped stages: parse + pair | align (one thread per GPU) | write
    const uint64_t readsPerBatch = (uint64_t)(300000 * o.readsPerBatchFactor);
    const uint64_t minReadSize = (uint16_t)o.Cthreshold + o.ksize - 1;
    // (with -g / -gc no pair record is ever produced — at HEAD nothing happens behind the threading gate, AQ.cpp:2070-2090; under the v1.3
    // contract the walk counts exactly and prints alignments, not kam lines — so no record buffer travels and the blocks take the record-free path)
    const bool want_recs = !o.threading && (o.okam || o.extractFastX);
    const bool fq = o.isFastq;
    time1 = time(nullptr);
    fprintf(stderr, "threads created\n");
    typedef std::unique_ptr<Batch> BatchP;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
    const double loop_t0 = now();
    const double t_loop = wall();
    const unsigned hw = std::max(4u, std::min(std::thread::hardware_concurrency(), 8 * cpus));  // (what the thread counts below are derived from)
    FILE* gzout = nullptr;
    if (emit_aln && !o.alnGz.empty()) { gzout = fopen(o.alnGz.c_str(), "wb"); if (!gzout) die_assert("cannot create " + o.alnGz); }
    const int emit_threads = o.emitThreads > 0 ? o.emitThreads : (int)std::max(1u, std::min(hw / 4, cpus));  // (never more than the CPUs the
                                                                                                              // container may use: deflate is pure CPU work)
    uint64_t aln_bytes = 0;
    std::mutex out_m, tot_m;  // stdout / the gzip file (one batch at a time); the totals below
    std::atomic<uint64_t> rec_us{0}, fmt_us{0}, gz_us{0};  // -a / -ae: record read-back (wall), formatting and deflate (summed over the emit threads)
    uint64_t nReads = 0, dev_ingest_reads = 0;
    std::atomic<uint64_t> dev_aln_text{0};  // bytes of -a / -ae text the device assembled (and compressed)
    std::vector<dbtk_ingest_t*> spent_ingests;
    double read_busy = 0, cut_busy = 0, pair_busy = 0, gpu_busy = 0, write_busy = 0;  // seconds each stage spent working (not waiting), summed over the shards
    int nsplit_used = 0;
    // -a / -ae: writeAlignments (AQ.cpp:1742-1759), `src dst title seq2 seq1 cigar2 annot2 cigar1 annot1`, formatted (and,
    // with --aln-gz, deflated into independent gzip members) in chunks by a pool of host threads while the GPU threads
    // are already on the next batches; the chunks leave in record order.
    // The emit pool: persistent helper threads (their text / deflate buffers and deflate states live as long as they do — fresh ones
    // per chunk had a hundred threads queueing on the kernel's memory-map lock, and the ingest's threads behind them), fed chunk
    // by chunk by the aligner threads, each of which waits for its own batch's chunks.
    struct EmitTask { Batch* b; uint64_t c; std::atomic<uint64_t>* left; };
    std::mutex ep_m;
    std::condition_variable ep_cv, ep_done;
    std::deque<EmitTask> ep_q;
    bool ep_stop = false;
    const uint64_t CH = 512;  // records per chunk (= per gzip member): ~230 KB of text
    auto emit_worker = [&] {
        std::string t, gzbuf;
        t.reserve(CH * 700);
        z_stream z;
        memset(&z, 0, sizeof z);
        bool zinit = false;
        for (;;) {
            EmitTask k;
            {
                std::unique_lock<std::mutex> l(ep_m);
                ep_cv.wait(l, [&] { return !ep_q.empty() || ep_stop; });
                if (ep_q.empty()) break;
                k = ep_q.front(); ep_q.pop_front();
            }
            Batch& b = *k.b;
            const uint64_t n = b.aln_em.size();
            t.clear();
            const double tf0 = now();
            for (uint64_t i = k.c * CH; i < std::min(n, (k.c + 1) * CH); ++i) {
                const uint64_t p = b.aln_em[i];
                const uint8_t* rec = b.aln.data() + b.aln_idx[p];
                uint32_t dst, len;
                memcpy(&dst, rec, 4); memcpy(&len, rec + 4, 4);
                if (o.simmode) { t += std::to_string((unsigned long long)b.src[p]); t += '\t'; }  // sams[i].src (writeAlignments, AQ.cpp:1744-1745)
                else t += ".\t";  // srcLocus is -1 outside simulation mode
                t += std::to_string((int)dst); t += '\t';
                const Batch::Span ti = b.title_s(p), s1 = b.seq_s(2 * p + 1), s0 = b.seq_s(2 * p);
                t.append(ti.first, ti.second); t += '\t';
                t.append(s1.first, s1.second); t += '\t';
                t.append(s0.first, s0.second); t += '\t';
                t.append((const char*)rec + 8, len); t += '\n';
            }
            const double tf1 = now();
            fmt_us += (uint64_t)((tf1 - tf0) * 1e6);
            if (gzout) {  // one gzip member per chunk: `zcat FILE` is the concatenation
                if (!zinit) { if (deflateInit2(&z, o.gzLevel, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die_assert("deflateInit2 failed"); zinit = true; }
                else if (deflateReset(&z) != Z_OK) die_assert("deflateReset failed");
                const size_t bound = deflateBound(&z, t.size()) + 64;
                if (gzbuf.size() < bound) gzbuf.resize(bound);
                z.next_in = (Bytef*)t.data(); z.avail_in = (uInt)t.size();
                z.next_out = (Bytef*)&gzbuf[0]; z.avail_out = (uInt)gzbuf.size();
                if (deflate(&z, Z_FINISH) != Z_STREAM_END) die_assert("deflate failed");
                b.aln_chunks[k.c].assign(gzbuf.data(), gzbuf.size() - z.avail_out);
                gz_us += (uint64_t)((now() - tf1) * 1e6);
            } else b.aln_chunks[k.c] = t;
            if (k.left->fetch_sub(1) == 1) { std::lock_guard<std::mutex> l(ep_m); ep_done.notify_all(); }
        }
        if (zinit) deflateEnd(&z);
    };
    std::vector<std::thread> emit_pool;
    if (emit_aln && !o.parseOnly) for (int i = 0; i < emit_threads; ++i) emit_pool.emplace_back(emit_worker);
    // Called by the batch's ALIGNER thread right after it has fetched the records (its sibling aligner threads keep the GPU busy
    // meanwhile: --aln-aligners contexts per GPU); the ordered writer then only writes the finished chunks.
    auto prepare_alignments = [&](Batch& b) {
        b.aln_chunks.clear();
        b.aln_em.clear();
        const uint64_t npairs_b = b.nreads / 2;
        for (uint64_t p = 0; p < npairs_b; ++p)
            if (b.aln_idx[p] != DBTK_NAN32) {
                if (o.simmode && o.aln_minimal) {  // -s -ae: srcLocus != nloci or destLocus != nloci (AQ.cpp:2242)
                    uint32_t dst;
                    memcpy(&dst, b.aln.data() + b.aln_idx[p], 4);
                    if (b.src[p] == nloci && dst == nloci) continue;
                }
                b.aln_em.push_back((uint32_t)p);
            }
        const uint64_t n = b.naln = b.aln_em.size();
        if (!n) return;
        const uint64_t nch = (n + CH - 1) / CH;
        b.aln_chunks.resize(nch);
        std::atomic<uint64_t> left{nch};
        {
            std::lock_guard<std::mutex> l(ep_m);
            for (uint64_t c = 0; c < nch; ++c) ep_q.push_back(EmitTask{&b, c, &left});
        }
        ep_cv.notify_all();
        std::unique_lock<std::mutex> l(ep_m);
        ep_done.wait(l, [&] { return left.load() == 0; });
    };
    auto write_alignments = [&](const Batch& b) {
        std::lock_guard<std::mutex> lk(out_m);
        FILE* f = gzout ? gzout : stdout;
        for (auto& c : b.aln_chunks) {
            if (fwrite(c.data(), 1, c.size(), f) != c.size()) die_assert(gzout ? "write to the --aln-gz file failed" : "write to stdout failed");
            aln_bytes += c.size();
        }
        if (b.aln_dev_bytes) {
            if (fwrite(b.aln_dev, 1, b.aln_dev_bytes, f) != b.aln_dev_bytes) die_assert(gzout ? "write to the --aln-gz file failed" : "write to stdout failed");
            aln_bytes += b.aln_dev_bytes;
        }
    };
    auto emit = [&](const Batch& b) {
        std::string out;
        if (emit_aln) write_alignments(b);
        auto seq = [&](uint64_t r) { const Batch::Span x = b.seq_s(r); return std::string(x.first, x.second); };
        auto qual = [&](uint64_t r) { const Batch::Span x = b.qual_s(r); return std::string(x.first, x.second); };
        for (uint64_t i = 0; i < b.nrec; ++i) {
            const dbtk_pair_rec_t& r = b.recs[i];
            const uint64_t p = r.pair;
            if (o.extractFastX) {  // writeExtractedReads, AQ.cpp:1618-1644: mate 2p+1 first, then 2p
                for (int which = 1; which >= 0; --which) {
                    out += b.title(p);
                    if (o.extractFastX != 1) { out += ':'; out += std::to_string(r.dst); }
                    out += '\n'; out += seq(2 * p + which); out += '\n';
                    if (fq) { out += "+\n"; out += qual(2 * p + which); out += '\n'; }
                }
                continue;
            }
            const uint64_t src = o.simmode ? b.src[p] : ~0ull;
            const bool src_ok = src != nloci && src != ~0ull;
            if (!(src_ok || r.dst != nloci)) continue;  // AQ.cpp:2169
            out += (src == ~0ull ? std::string(".") : std::to_string((int)src)); out += '\t';
            out += std::to_string(r.dst); out += '\t';
            out += std::to_string(r.dst != r.dst0 ? (int)r.dst0 : -1); out += '\t';
            out += std::to_string(r.r2.ei - r.r2.si); out += '\t';
            out += std::to_string(r.r1.ei - r.r1.si); out += '\t';
            out += "kf:hf:bf:qf:af:rm:qn:qm:si:nt:bs:ti\t";
            mate_fields(out, r.r2); out += '\t';
            mate_fields(out, r.r1); out += '\t';
            out += annot2str(r.r2); out += '\t';
            out += annot2str(r.r1); out += '\t';
            out += b.title(p).substr(1); out += '\t';
            out += seq(2 * p + 1); out += '\t';
            out += fq ? qual(2 * p + 1) : std::string("."); out += '\t';
            out += seq(2 * p); out += '\t';
            out += fq ? qual(2 * p) : std::string("."); out += '\n';
        }
        if (!out.empty()) { std::lock_guard<std::mutex> lk(out_m); fwrite(out.data(), 1, out.size(), stdout); }
        fprintf(stderr, "Batch query in %ld sec. %llu pairs, %llu records\n", b.gpu_sec, (unsigned long long)(b.nreads / 2), (unsigned long long)b.nrec);
    };
    // --parse-only: what the pairing stage handed on (pairs, bases, order-independent digest)
    auto digest_batch = [&](const Batch& b) {
        const uint64_t npairs = b.nreads / 2;
        uint64_t dg = 0;
        auto fnv = [](uint64_t h, const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; for (size_t i = 0; i < n; ++i) { h ^= q[i]; h *= 0x100000001B3ull; } h ^= 0xFF; h *= 0x100000001B3ull; return h; };
        for (uint64_t pr = 0; pr < npairs; ++pr) {
            uint64_t hsh = 0xCBF29CE484222325ull;
            hsh = fnv(hsh, b.tar.data() + b.toff[pr], b.toff[pr + 1] - b.toff[pr]);
            for (int m = 0; m < 2; ++m) hsh = fnv(hsh, b.flat.data() + b.off[2 * pr + m], b.off[2 * pr + m + 1] - b.off[2 * pr + m]);
            if (fq) for (int m = 0; m < 2; ++m) hsh = fnv(hsh, b.qar.data() + b.qoff[2 * pr + m], b.qoff[2 * pr + m + 1] - b.qoff[2 * pr + m]);
            dg += hsh;
        }
        po_pairs += npairs; po_bases += b.flat.size(); po_digest += dg;
    };
    // Multi-GPU ingest: with --gpus N and a seekable input the file is cut into N byte ranges at record boundaries and every
    // GPU gets its own reader / splitters / pairing / aligner / writer pipeline over its range (one pipeline feeding N GPUs
    // left them idle: the kernels take ~100 x less time than the parse).  Mates are paired INSIDE a range; a range's records
    // that found no partner (mates on either side of a cut, singletons) are collected and paired across ranges at the end,
    // in range order — for titles that occur at most twice this is exactly the single reader's outcome.
    struct Left { std::string title, seq, qual; };
    int npipes = 1;  // ingest pipelines running side by side (set once the ranges are known)
    int nshards_now = 1;
    FILE* handover_stream = nullptr;  // a pipe the device reader has read from: what it took but did not pair, then the rest of the pipe
    auto run_shard = [&](const int shard, const int gpu0, const int ngpu_here, const uint64_t lo, const uint64_t hi, std::vector<Left>& leftovers) {
    Reader in;
    in.f = handover_stream ? handover_stream : fopen(o.fastxFname.c_str(), "rb");
    if (!in.f) die_assert("cannot open " + o.fastxFname);
    if (!handover_stream && lo && fseeko(in.f, (off_t)lo, SEEK_SET)) die_assert("cannot seek in " + o.fastxFname);
    uint64_t remaining = hi - lo;  // bytes of this range still to read
    uint64_t nReads = 0;
    double read_busy = 0, cut_busy = 0, pair_busy = 0, gpu_busy = 0, write_busy = 0;
    Chan<BatchP> parsed, aligned;
    // written batches go back to the pairing stage with their arrays' capacity (as the input blocks do, below)
    std::mutex bpool_m;
    std::vector<BatchP> bpool;
    auto recycle_batch = [&](BatchP b) {
        if (!b) return;
        b->flat.clear(); b->off.clear(); b->qar.clear(); b->qoff.clear(); b->tar.clear(); b->toff.clear(); b->src.clear();
        b->index = 0; b->nreads = 0; b->nReads_so_far = 0; b->nparked = 0; b->nrec = 0; b->gpu_sec = 0; b->naln = 0; b->aln_chunks.clear();
        std::lock_guard<std::mutex> l(bpool_m);
        if (bpool.size() < 16) bpool.push_back(std::move(b));
    };
    auto fresh_batch = [&]() -> BatchP {
        {
            std::lock_guard<std::mutex> l(bpool_m);
            if (!bpool.empty()) { BatchP b = std::move(bpool.back()); bpool.pop_back(); return b; }
        }
        return BatchP(new Batch);
    };

    // Stage A — reader, line splitters, on-the-fly mate pairing (AQ.cpp:1918-1976), all overlapped:
    //   A0 (1 thread)  reads the file in large blocks cut at record boundaries (2 lines per FASTA record, 4 per FASTQ: the
    //                  newline count of a block tells where its last whole record ends);
    //   A1 (n threads) split a block into records: spans (offset, length) of title / sequence / quality, nothing copied;
    //   A2 (1 thread)  takes the blocks back in file order and pairs.  The reference parks every record under its title in a
    //                  map until a record with the same title arrives; here the most recently parked record is held outside
    //                  the map, so that interleaved input (mates adjacent) never touches the map.  A record matches the held
    //                  one first, then the map — the held one is by construction the latest record parked under its title,
    //                  so the outcome is the reference's.  Paired reads are copied once, into the batch's flat arrays.
    //                  Interleaved input is paired in A1 already: a block whose records 2j, 2j+1 all carry equal titles is
    //                  laid out there as the batch arrays it will become ("the record that completed the pair" first), and
    //                  A2 — when nothing is parked, which is then always — appends it with a few block copies.  Any other
    //                  block, or any parked record, takes the record-by-record path; both give the reference's outcome.
    struct Rec { uint32_t t, tn, s, sn, q, qn; };
    struct RawBuf {  // a block's bytes: grown without zero-filling (the reader is a single thread on the critical path)
        std::unique_ptr<char[]> p; size_t n = 0;
        char* data() { return p.get(); }
        const char* data() const { return p.get(); }
        size_t size() const { return n; }
        void resize(size_t m) {
            if (m <= n) return;
            std::unique_ptr<char[]> q(new char[m]);
            if (n) memcpy(q.get(), p.get(), n);
            p = std::move(q); n = m;
        }
    };
    struct Block {
        uint64_t index = 0; RawBuf data; size_t base = 0, len = 0; std::vector<Rec> recs;  // the block's bytes: data[base, base + len)
        bool clean = false;  // every record pairs with its neighbour: `mini` holds the kept pairs in batch layout
        Batch mini;
        uint64_t mpos = 0;   // pairs of `mini` already handed on
    };
    typedef std::unique_ptr<Block> BlockP;
    Chan<BlockP> raw, split;
    // used blocks go back to the reader: fresh 32 MB buffers cost a page fault per 4 KB, on the one thread that limits the ingest
    std::mutex pool_m;
    std::vector<BlockP> pool;
    auto recycle = [&](BlockP b) {
        if (!b) return;
        b->recs.clear(); b->clean = false; b->mpos = 0; b->len = 0; b->base = 0;
        Batch& m = b->mini;
        m.flat.clear(); m.off.clear(); m.qar.clear(); m.qoff.clear(); m.tar.clear(); m.toff.clear(); m.nreads = 0;
        std::lock_guard<std::mutex> l(pool_m);
        if (pool.size() < 64) pool.push_back(std::move(b));
    };
    auto fresh_block = [&]() -> BlockP {
        {
            std::lock_guard<std::mutex> l(pool_m);
            if (!pool.empty()) { BlockP b = std::move(pool.back()); pool.pop_back(); return b; }
        }
        return BlockP(new Block);
    };
    const int nsplit = (int)std::min(16u, std::max(2u, hw / 8 / (unsigned)std::max(1, npipes)));  // (the pipelines share the host's threads)
    raw.cap = split.cap = 2 * (size_t)nsplit;
    const size_t L = fq ? 4 : 2;
    // A0 as two threads, so that the copy out of the page cache and the newline count overlap: `io` freads fixed-size
    // chunks behind some headroom; `reader` (the cutter) puts the unfinished record of the previous chunk in front (into the
    // headroom), counts the newlines and cuts at the last whole record.
    size_t BLK = 32u << 20;
    if (const char* e = getenv("DBTK_INGEST_BLOCK")) { const long v = atol(e); if (v >= 64) BLK = (size_t)v; }  // (tests: many small blocks)
    const size_t HEAD = std::min<size_t>(1u << 20, BLK);  // room for the carried-over partial record in front of a chunk
    Chan<BlockP> chunks;
    chunks.cap = 4;
    std::thread io([&] {
        for (;;) {
            const double tr = now();
            BlockP b = fresh_block();
            b->data.resize(HEAD + BLK);
            const size_t want = (size_t)std::min<uint64_t>(BLK, remaining);
            const size_t n = want ? fread(b->data.data() + HEAD, 1, want, in.f) : 0;
            if (n == 0) break;
            remaining -= n;
            b->base = HEAD; b->len = n;
            read_busy += now() - tr;
            chunks.push(std::move(b));
        }
        chunks.close();
    });
    std::thread reader([&] {
        std::vector<char> carry;
        uint64_t index = 0;
        BlockP b;
        auto emit = [&](BlockP blk) {
            if (blk->len == 0) return;
            if (blk->len >= 0xFFFFFFFFull) die_assert("input block too large");
            blk->index = index++;
            raw.push(std::move(blk));
        };
        while (chunks.pop(b)) {
            const double tr = now();
            if (carry.size() <= b->base) {  // the usual case: the partial record goes into the headroom
                b->base -= carry.size();
                if (!carry.empty()) memcpy(b->data.data() + b->base, carry.data(), carry.size());
                b->len += carry.size();
            } else {  // a partial record longer than the headroom (a record longer than a chunk): rebuild the block
                BlockP nb = fresh_block();
                nb->data.resize(carry.size() + b->len);
                memcpy(nb->data.data(), carry.data(), carry.size());
                memcpy(nb->data.data() + carry.size(), b->data.data() + b->base, b->len);
                nb->base = 0; nb->len = carry.size() + b->len;
                recycle(std::move(b));
                b = std::move(nb);
            }
            carry.clear();
            const char* d = b->data.data() + b->base;
            const size_t end = b->len;
            // whole records in [0, end): the first floor(newlines / L) * L lines
            const size_t nl = count_nl(d, end);
            size_t cut = 0;
            if (nl >= L) {
                size_t drop = nl % L;  // trailing whole lines that start an unfinished record
                const char* p = (const char*)memrchr(d, '\n', end);
                while (drop--) p = (const char*)memrchr(d, '\n', p - d);
                cut = p - d + 1;
            }
            carry.assign(d + cut, d + end);
            b->len = cut;
            cut_busy += now() - tr;
            if (cut) emit(std::move(b)); else recycle(std::move(b));
        }
        if (!carry.empty()) {  // end of file: what is left is the last block (std::getline semantics for a missing last newline)
            BlockP nb = fresh_block();
            nb->data.resize(carry.size());
            memcpy(nb->data.data(), carry.data(), carry.size());
            nb->base = 0; nb->len = carry.size();
            emit(std::move(nb));
        }
        raw.close();
    });
    std::vector<std::thread> splitters;
    std::mutex split_m;
    int splitters_left = nsplit;
    for (int w = 0; w < nsplit; ++w)
        splitters.emplace_back([&] {
            BlockP b;
            while (raw.pop(b)) {
                const char* d0 = b->data.data() + b->base;
                const size_t len = b->len;
                size_t pos = 0;
                b->recs.reserve(len / 300 + 16);
                auto line = [&](uint32_t* o, uint32_t* n) {  // std::getline: the rest of the data when no newline is left
                    if (pos >= len) { *o = (uint32_t)len; *n = 0; return; }
                    const char* nlp = (const char*)memchr(d0 + pos, '\n', len - pos);
                    const size_t e = nlp ? (size_t)(nlp - d0) : len;
                    *o = (uint32_t)pos; *n = (uint32_t)(e - pos);
                    pos = nlp ? e + 1 : len;
                };
                while (pos < len) {
                    Rec r{0, 0, 0, 0, 0, 0};
                    uint32_t xo, xn;
                    line(&r.t, &r.tn);
                    line(&r.s, &r.sn);
                    if (fq) { line(&xo, &xn); line(&r.q, &r.qn); }
                    // prunePEinfo, AQ.cpp:455-462
                    if (r.tn >= 2 && d0[r.t + r.tn - 2] == '/' && (d0[r.t + r.tn - 1] == '1' || d0[r.t + r.tn - 1] == '2')) r.tn -= 2;
                    b->recs.push_back(r);
                }
                // interleaved input: pair here, in parallel
                const size_t nr = b->recs.size();
                bool clean = nr > 0 && nr % 2 == 0 && !o.simmode;
                for (size_t j = 0; clean && j < nr; j += 2) {
                    const Rec &x = b->recs[j], &y = b->recs[j + 1];
                    clean = x.tn == y.tn && memcmp(d0 + x.t, d0 + y.t, x.tn) == 0;
                }
                if (clean) {
                    Batch& m = b->mini;
                    m.off.reserve(nr + 1); m.toff.reserve(nr / 2 + 1);
                    m.off.push_back(0); m.qoff.push_back(0); m.toff.push_back(0);
                    m.flat.reserve(len); m.tar.reserve(len / 8);
                    if (fq) { m.qar.reserve(len / 2); m.qoff.reserve(nr + 1); }
                    for (size_t j = 0; j < nr; j += 2) {
                        const Rec &x = b->recs[j], &y = b->recs[j + 1];  // x is parked, y completes the pair
                        if (y.sn < minReadSize || x.sn < minReadSize) continue;  // AQ.cpp:1940-1943: the pair is dropped
                        m.tar.insert(m.tar.end(), d0 + y.t, d0 + y.t + y.tn); m.toff.push_back(m.tar.size());
                        m.add_read(d0 + y.s, y.sn, d0 + y.q, y.qn, fq);
                        m.add_read(d0 + x.s, x.sn, d0 + x.q, x.qn, fq);
                        m.nreads += 2;
                    }
                    b->clean = true;
                }
                split.push(std::move(b));
            }
            std::lock_guard<std::mutex> l(split_m);
            if (--splitters_left == 0) split.close();
        });
    std::thread parser([&] {
        std::unordered_map<std::string, std::pair<std::string, std::string>> parked;  // readDB / fqDB
        std::string held_title, held_seq, held_qual, key;
        bool held = false;
        uint64_t index = 0, next_block = 0;
        std::map<uint64_t, BlockP> waiting;
        BlockP blk;          // the block being consumed
        size_t ri = 0;       // next record of it
        bool drained = false;
        auto have_record = [&]() -> bool {  // !in.at_eof()
            for (;;) {
                if (blk && ri < blk->recs.size()) return true;
                recycle(std::move(blk));
                auto it = waiting.find(next_block);
                while (it == waiting.end() && !drained) {
                    BlockP got;
                    if (!split.pop(got)) { drained = true; break; }
                    waiting[got->index] = std::move(got);
                    it = waiting.find(next_block);
                }
                if (it == waiting.end()) return false;
                blk = std::move(it->second);
                waiting.erase(it);
                ++next_block;
                ri = 0;
            }
        };
        for (;;) {
            if (!have_record()) break;
            BatchP b = fresh_batch();
            b->off.push_back(0); b->qoff.push_back(0); b->toff.push_back(0);
            double tb = now();
            while (b->nreads < readsPerBatch && have_record()) {
                if (blk->clean && !held && parked.empty() && (ri == 0 || blk->mpos)) {
                    // a pre-paired block and nothing parked: append as many of its pairs as the batch still takes
                    Batch& m = blk->mini;
                    const uint64_t avail = m.nreads / 2 - blk->mpos, room = (readsPerBatch - b->nreads + 1) / 2;
                    const uint64_t n = avail < room ? avail : room, p0 = blk->mpos;
                    auto append = [](auto& dst, auto& doff, const auto& src, const auto& soff, uint64_t i0, uint64_t i1) {
                        const uint64_t base = dst.size(), s0 = soff[i0];
                        dst.insert(dst.end(), src.begin() + s0, src.begin() + soff[i1]);
                        for (uint64_t i = i0 + 1; i <= i1; ++i) doff.push_back(base + (soff[i] - s0));
                    };
                    append(b->flat, b->off, m.flat, m.off, 2 * p0, 2 * (p0 + n));
                    if (fq) append(b->qar, b->qoff, m.qar, m.qoff, 2 * p0, 2 * (p0 + n));
                    append(b->tar, b->toff, m.tar, m.toff, p0, p0 + n);
                    b->nreads += 2 * n;
                    blk->mpos += n;
                    if (blk->mpos == m.nreads / 2) ri = blk->recs.size();  // the block is used up
                    else ri = 1;                                            // (not 0: the block stays on this path)
                    continue;
                }
                const Rec& r = blk->recs[ri++];
                const char* d0 = blk->data.data() + blk->base;
                const char *tp = d0 + r.t, *sp = d0 + r.s, *qp = d0 + r.q;
                const size_t tn = r.tn, sn = r.sn, qn = r.qn;
                const char *s2p = nullptr, *q2p = nullptr;
                size_t s2n = 0, q2n = 0;
                bool matched = false;
                std::pair<std::string, std::string> from_map;
                if (held && held_title.size() == tn && memcmp(held_title.data(), tp, tn) == 0) {
                    s2p = held_seq.data(); s2n = held_seq.size(); q2p = held_qual.data(); q2n = held_qual.size();
                    held = false;
                    matched = true;
                } else if (!parked.empty()) {
                    key.assign(tp, tn);
                    auto it = parked.find(key);
                    if (it != parked.end()) {
                        from_map = std::move(it->second);
                        parked.erase(it);
                        s2p = from_map.first.data(); s2n = from_map.first.size(); q2p = from_map.second.data(); q2n = from_map.second.size();
                        matched = true;
                    }
                }
                if (!matched) {  // park this record; the previously held one moves into the map
                    if (held) parked[held_title] = std::make_pair(held_seq, held_qual);
                    held_title.assign(tp, tn); held_seq.assign(sp, sn); held_qual.assign(qp, qn);
                    held = true;
                    if (!have_record()) break;
                    continue;
                }
                if (sn < minReadSize || s2n < minReadSize) continue;  // AQ.cpp:1940-1943: the pair is dropped
                if (o.simmode) b->src.push_back(parse_src(std::string(tp, tn), o.simmode, nloci));
                b->tar.insert(b->tar.end(), tp, tp + tn); b->toff.push_back(b->tar.size());
                b->add_read(sp, sn, qp, qn, fq);      // seqs[2p]: the record that completed the pair
                b->add_read(s2p, s2n, q2p, q2n, fq);  // seqs[2p+1]: the parked one
                b->nreads += 2;
            }
            pair_busy += now() - tb;  // (includes waiting for the splitters when they are the slower stage)
            nReads += b->nreads;
            b->nReads_so_far = nReads;
            b->nparked = parked.size() + (held ? 1 : 0);
            fprintf(stderr, "Buffered reading %llu\t%llu\t%zu\n", (unsigned long long)b->nreads, (unsigned long long)nReads, b->nparked);
            if (b->nreads == 0) continue;
            b->index = index++;
            parsed.push(std::move(b));
        }
        // what found no partner inside this range goes on to the cross-range pairing
        if (held) leftovers.push_back(Left{held_title, held_seq, held_qual});
        for (auto& kv : parked) leftovers.push_back(Left{kv.first, kv.second.first, kv.second.second});
        parsed.close();
    });

    // Stage B — one thread per GPU: the batch through the hot path (replaces AQ.cpp:1988-2249)
    std::vector<std::thread> workers;
    std::mutex done_m;
    int workers_left = ngpu_here;
    for (int d = gpu0; d < gpu0 + ngpu_here; ++d)
        workers.emplace_back([&, d] {
            BatchP b;
            std::vector<uint8_t> flatq;
            while (parsed.pop(b)) {
                const time_t t2 = time(nullptr);
                const double tg = now();
                const uint64_t npairs = b->nreads / 2;
                if (o.parseOnly) {
                    digest_batch(*b);
                    aligned.push(std::move(b));
                    continue;
                }
                const bool send_qual = use_bait && fq;
                if (send_qual) {  // qualities feed qString2qMask (AQ.cpp:2104-2107); a quality string is as long as its read
                    flatq.assign(b->flat.size() + 1, (uint8_t)'!');
                    for (uint64_t r = 0; r < b->nreads; ++r)
                        memcpy(flatq.data() + b->off[r], b->qar.data() + b->qoff[r], std::min(b->qoff[r + 1] - b->qoff[r], b->off[r + 1] - b->off[r]));
                }
                if (want_recs) b->recs.resize(npairs);
                b->flat.push_back(0);
                const dbtk_status_t st = dbtk_align_batch(ctx[d], b->flat.data(), b->off.data(), send_qual ? flatq.data() : nullptr, npairs,
                                                          want_recs ? b->recs.data() : nullptr, want_recs ? npairs : 0, &b->nrec);
                if (st) die_assert(std::string("align: ") + dbtk_last_error());
                if (emit_aln) {
                    uint64_t used = 0;
                    const double tr0 = now();
                    b->aln_idx.resize(npairs);
                    // (a recycled batch keeps its buffer: usually large enough, and its pages are mapped)
                    dbtk_status_t sa = dbtk_ctx_aln_text(ctx[d], b->aln_idx.data(), npairs, b->aln.data(), b->aln.size(), &used);
                    if (sa == DBTK_ERR_OVERFLOW && used > b->aln.size()) {  // (nothing was copied: `used` = the bytes the buffer must hold)
                        b->aln.grow((size_t)used + used / 4);
                        sa = dbtk_ctx_aln_text(ctx[d], b->aln_idx.data(), npairs, b->aln.data(), b->aln.size(), &used);
                    }
                    if (sa) die_assert(std::string("alignment records: ") + dbtk_last_error());
                    rec_us += (uint64_t)((now() - tr0) * 1e6);
                    prepare_alignments(*b);
                }
                b->gpu_sec = (long)(time(nullptr) - t2);
                { std::lock_guard<std::mutex> l(done_m); gpu_busy += now() - tg; }
                aligned.push(std::move(b));
            }
            std::lock_guard<std::mutex> l(done_m);
            if (--workers_left == 0) aligned.close();
        });

    // Stage C — critical section B (AQ.cpp:2253-2279): stdout, in batch order
    {
        std::map<uint64_t, BatchP> waiting;
        uint64_t next = 0;
        std::string out;
        BatchP got;
        while (aligned.pop(got)) {
            waiting[got->index] = std::move(got);
            for (auto it = waiting.find(next); it != waiting.end(); it = waiting.find(next)) {
                const double tw = now();
                emit(*it->second);
                write_busy += now() - tw;
                recycle_batch(std::move(it->second));
                waiting.erase(it);
                ++next;
            }
        }
    }
    io.join();
    reader.join();
    for (auto& w : splitters) w.join();
    parser.join();
    for (auto& w : workers) w.join();
    fclose(in.f);
    (void)shard;
    return std::make_tuple(nReads, read_busy, cut_busy, pair_busy, gpu_busy, write_busy, nsplit);
    };  // run_shard
    // The reader on the device (include/dbtk.h: dbtk_ingest_*; kernels in dbtk_ingest.h).  For interleaved input the host only moves
    // bytes: a few threads pread fixed-size chunks of [lo, hi) straight into the pinned buffers of the ingest's slots, this thread
    // submits them in file order (host-to-device copy + the parse kernels: newline scan, record table, title comparison of
    // neighbouring records, minimal read size, the batch arrays) and runs every parsed block through the hot path.  The first
    // block that is not a run of adjacent mates ends it: *resume = the input offset the host reader (run_shard: splitters +
    // park-by-title pairing) continues from — up to there every record was paired, so nothing is parked there, exactly as in
    // the reference's reader at that point.  Records (kam lines, -e) are written from the spans the device made, in file order.
    // A pipe (`samtools fasta ... | danbing-tk ... -fa /dev/stdin`, the README's command line) is read the same way by ONE thread, chunk
    // after chunk (a chunk is submitted once it is known whether another follows); when the host reader has to take over, the bytes
    // already taken from the pipe but not paired go to *replay and the open descriptor to *pipe_fd: the host reader continues there.
    auto run_device_ingest = [&](dbtk_ctx_t* cx, const uint64_t lo, const uint64_t hi, uint64_t* resume, const bool piped, std::string* replay, int* pipe_fd) {
        const size_t CH = ingest_chunk();
        const bool want_out = want_recs || emit_aln;  // records need titles and reads on the host: the slot's bytes stay until they are written
                                                      // (-a / -ae lines are made on the device, from the spans)
        const int nio_want = ingest_readers(piped, npipes);
        const uint32_t NS = ingest_slots(nio_want);
        dbtk_ingest_t* ing = nullptr;
        const double ts0 = now();
        if (dbtk_ingest_create(cx, fq, (uint32_t)minReadSize, CH, NS, want_out, &ing)) die_assert(dbtk_last_error());
        const double setup_s = now() - ts0;
        if (getenv("DBTK_VERBOSE")) fprintf(stderr, "device reader: created %.3f s into the batch loop's time\n", ts0 - loop_t0);
        double first_s = 0, wait_s = 0;
        const int fd = open(o.fastxFname.c_str(), O_RDONLY);
        if (fd < 0) die_assert("cannot open " + o.fastxFname);
#ifdef F_SETPIPE_SZ
        if (piped) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);  // (a larger pipe buffer: fewer wake-ups of the producer; refused for a non-pipe, which is fine)
#endif
        const uint64_t total = piped ? 0 : hi - lo;
        std::atomic<uint64_t> nchunks{piped ? ~0ull : std::max<uint64_t>(1, (total + CH - 1) / CH)};  // (a pipe: known once its end has been read; the batch loop reads it without the lock)
        std::mutex m;
        std::condition_variable cv;
        std::vector<char> filled(piped ? 0 : nchunks.load(), 0);
        std::vector<uint64_t> chunk_n(piped ? 0 : nchunks.load(), 0);  // bytes of each chunk
        if (!piped) for (uint64_t j = 0; j < nchunks; ++j) chunk_n[j] = std::min<uint64_t>(CH, total - j * CH);
        uint64_t next_read = 0, nreleased = 0;
        bool stop = false;
        double rb = 0, gb = 0, wb = 0;
        uint64_t pipe_cur = 0;  // (pipe) the chunk being read when the reader was told to stop
        auto io_pipe = [&] {  // sequential: chunk j is handed on once chunk j + 1 has its first byte, or the pipe has ended
            uint64_t j = 0;
            size_t got = 0;
            for (;;) {
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || j < nreleased + NS; });
                    if (filled.size() <= j) { filled.resize(j + 1, 0); chunk_n.resize(j + 1, 0); }
                    if (stop) { chunk_n[j] = got; pipe_cur = j; return; }
                }
                const double t0 = now();
                char* dst = (char*)dbtk_ingest_chunk_buffer(ing, (uint32_t)(j % NS));
                got = 0;
                bool eof = false;
                while (got < CH) {
                    const ssize_t r = read(fd, dst + got, CH - got);
                    if (r < 0 && errno == EINTR) continue;
                    if (r < 0) die_assert("read error on " + o.fastxFname);
                    if (r == 0) { eof = true; break; }
                    got += (size_t)r;
                    if (got == (size_t)r) {  // the first bytes of this chunk: the chunk before it is not the last one
                        std::lock_guard<std::mutex> l(m);
                        if (j > 0 && filled[j - 1] == 2) { filled[j - 1] = 1; cv.notify_all(); }
                    }
                    { std::lock_guard<std::mutex> l(m); if (stop) { chunk_n[j] = got; pipe_cur = j; return; } }
                }
                {
                    std::lock_guard<std::mutex> l(m);
                    rb += now() - t0;
                    chunk_n[j] = got;
                    if (eof) {
                        if (got == 0 && j > 0) { nchunks = j; if (filled[j - 1] == 2) filled[j - 1] = 1; }  // the chunk before was the last
                        else { nchunks = j + 1; if (j > 0 && filled[j - 1] == 2) filled[j - 1] = 1; filled[j] = 1; }
                        pipe_cur = j;
                        cv.notify_all();
                        return;
                    }
                    filled[j] = 2;  // full: whether it is the last one shows with the next read
                }
                ++j; got = 0;
            }
        };
        auto io = [&] {
            for (;;) {
                uint64_t j;
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || next_read >= nchunks || next_read < nreleased + NS; });  // (chunk j - NS has left slot j % NS)
                    if (stop || next_read >= nchunks) return;
                    j = next_read++;
                }
                const double t0 = now();
                char* dst = (char*)dbtk_ingest_chunk_buffer(ing, (uint32_t)(j % NS));
                const size_t want = (size_t)std::min<uint64_t>(CH, total - j * CH);
                size_t got = 0;
                while (got < want) {
                    const ssize_t r = pread(fd, dst + got, want - got, (off_t)(lo + j * CH + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) die_assert("read error on " + o.fastxFname);
                    got += (size_t)r;
                }
                {
                    std::lock_guard<std::mutex> l(m);
                    filled[j] = 1;
                    rb += now() - t0;
                }
                cv.notify_all();
            }
        };
        std::vector<std::thread> ios;
        const int nio = piped ? 1 : (int)std::min<uint64_t>(nchunks.load(), (uint64_t)nio_want);  // (the pipelines share the host's threads)
        if (piped) ios.emplace_back(io_pipe);
        else for (int i = 0; i < nio; ++i) ios.emplace_back(io);
        auto release = [&] { { std::lock_guard<std::mutex> l(m); ++nreleased; } cv.notify_all(); };
        Chan<std::unique_ptr<Batch>> outq;
        outq.cap = NS;
        std::thread writer;
        if (want_out)
            writer = std::thread([&] {  // in block order (the workers below finish in any order)
                std::map<uint64_t, std::unique_ptr<Batch>> waiting;
                uint64_t next = 0;
                std::unique_ptr<Batch> got;
                while (outq.pop(got)) {
                    waiting[got->index] = std::move(got);
                    for (auto it = waiting.find(next); it != waiting.end(); it = waiting.find(next)) {
                        const double t0 = now();
                        emit(*it->second);
                        wb += now() - t0;
                        waiting.erase(it);
                        ++next;
                        release();
                    }
                }
            });
        // a parsed block through the hot path, with what the writer needs: the records, the -a / -ae lines (made and compressed on the
        // device), the spans of titles and reads.  With -a / -ae several worker threads do this side by side, each with a context of
        // its own (the contexts of a GPU share its tables): one's wait for its kernels and copies overlaps the others' kernels.
        struct Work { uint64_t index; uint32_t slot; dbtk_ingest_info_t info; };
        const bool verbose_blocks = getenv("DBTK_VERBOSE") && atoi(getenv("DBTK_VERBOSE")) >= 2;
        std::mutex gb_m;
        auto process = [&](dbtk_ctx_t* wc, const Work& w) {
            const double t0 = now();
            std::unique_ptr<Batch> b(new Batch);
            b->index = w.index; b->nreads = 2 * (uint64_t)w.info.nkept;
            if (want_recs) b->recs.resize(w.info.nkept);
            const time_t t2 = time(nullptr);
            if (dbtk_ingest_align(ing, w.slot, wc, 1, want_recs ? b->recs.data() : nullptr, want_recs ? w.info.nkept : 0, &b->nrec)) die_assert(std::string("align: ") + dbtk_last_error());
            b->gpu_sec = (long)(time(nullptr) - t2);
            const double t_al = now();
            if (emit_aln) {  // the block's lines, text or gzip members, straight from the device
                const double tr0 = now();
                uint64_t nb = 0, nl = 0, tb = 0;
                const void* data = nullptr;
                if (dbtk_ingest_aln_lines(ing, w.slot, wc, gzout ? 1 : 0, &data, &nb, &nl, &tb)) die_assert(std::string("alignment lines: ") + dbtk_last_error());
                b->aln_dev = (const char*)data; b->aln_dev_bytes = nb;  // (in the slot's pinned buffer until the slot is released)
                b->naln = nl;
                rec_us += (uint64_t)((now() - tr0) * 1e6);
                dev_aln_text += tb;
            }
            if (want_recs) {
                b->spans.resize(w.info.nkept);
                if (dbtk_ingest_spans(ing, w.slot, b->spans.data(), w.info.nkept)) die_assert(std::string("ingest: ") + dbtk_last_error());
            }
            b->blk = (const char*)dbtk_ingest_block(ing, w.slot);
            if (verbose_blocks) fprintf(stderr, "block %llu ctx %p: start %.4f align %.4f lines %.4f\n", (unsigned long long)w.index, (void*)wc, t0 - ts0, t_al - t0, now() - t_al);
            { std::lock_guard<std::mutex> l(gb_m); gb += now() - t0; }
            outq.push(std::move(b));
        };
        std::vector<dbtk_ctx_t*> wctx{cx};  // the contexts the blocks may be aligned with: this pipeline's, and with -a / -ae the further ones of its GPU
        if (emit_aln && nshards_now == 1)
            for (size_t i = 1; i < ctx.size() && (int)wctx.size() < aln_aligners; ++i) if ((int)(i % (size_t)o.ngpus) == 0 && ctx[i] != cx) wctx.push_back(ctx[i]);
        Chan<Work> wq;
        wq.cap = NS;
        std::vector<std::thread> workers;
        if (want_out && wctx.size() > 1)
            for (dbtk_ctx_t* wc : wctx) workers.emplace_back([&, wc] { Work w; while (wq.pop(w)) process(wc, w); });
        uint64_t jsub = 0, jaln = 0, nR = 0;
        *resume = hi;
        bool handed = false;   // the host reader takes over
        uint32_t hslot = 0;
        const bool sync = want_out || P.bubbles;  // (-bu replays every batch's novel edges on the host)
        // (merging pays when the input is long — a pipe, or a file of at least 2 GB: below that the loop is bound by reading the file, every
        // block's kernels hide under the next block's bytes, and a merged batch's kernels would only start late.  DBTK_MERGE_PAIRS forces it.)
        const bool long_input = piped || total >= (2ull << 30) || getenv("DBTK_MERGE_PAIRS");
        const bool merged = !sync && !P.trace && !(P.bait && fq) && long_input && !getenv("DBTK_NO_MERGE");
        double align_call_s = 0, align_call_max = 0, flush_s = 0;  // (time inside the align calls themselves: enqueueing, and whatever a batch has to allocate)
        uint64_t merge_pairs = std::max<uint64_t>(1ull << 20, 24 * nloci);  // (LOC_MIN_PAIRS = 16 pairs per locus and half as many again)
        if (const char* e = getenv("DBTK_MERGE_PAIRS")) { const long long v = atoll(e); if (v > 0) merge_pairs = (uint64_t)v; }  // (tests: several merged batches)
        while (jaln < nchunks) {
            for (;;) {  // submit what has been read, up to NS - 1 blocks ahead of the one about to be aligned — and never block i before block
                        // i + 1 - NS has been released: parsing block i puts its carried-over bytes in front of the NEXT slot's device block,
                        // which until then still holds the first record of the block that is being written from it
                uint64_t nb = 0;
                bool last_one = false;
                {
                    std::unique_lock<std::mutex> l(m);
                    auto can = [&] { return jsub < nchunks && jsub < jaln + NS && jsub < filled.size() && filled[jsub] == 1 && jsub + 1 < nreleased + NS; };
                    if (jsub == jaln) cv.wait(l, [&] { return can() || jaln >= nchunks; });
                    if (!can()) break;
                    nb = chunk_n[jsub]; last_one = jsub + 1 == nchunks;
                }
                if (dbtk_ingest_submit(ing, (uint32_t)(jsub % NS), nb, last_one)) die_assert(std::string("ingest: ") + dbtk_last_error());
                ++jsub;
            }
            if (jsub <= jaln) break;  // (a pipe that ended on a chunk boundary: nothing more was submitted)
            const double t0 = now();
            const uint32_t slot = (uint32_t)(jaln % NS);
            dbtk_ingest_info_t info;
            if (dbtk_ingest_wait(ing, slot, &info)) die_assert(std::string("ingest: ") + dbtk_last_error());
            wait_s += now() - t0;
            if (jaln == 0) first_s = now() - ts0;
            if (piped && info.flags) {
                // the host reader takes over behind this block and replays what was taken from the pipe: the pipe's reader must stop BEFORE the
                // block is released (by the writer, or below) — a released block lets it read chunk jaln + NS into this very slot, over the
                // bytes the replay starts with
                { std::lock_guard<std::mutex> l(m); stop = true; }
                cv.notify_all();
            }
            if (info.flags & (DBTK_ING_DIRTY | DBTK_ING_LINES)) { *resume = lo + info.first_byte; handed = true; hslot = slot; std::lock_guard<std::mutex> l(gb_m); gb += now() - t0; break; }
            if (want_out) {
                { std::lock_guard<std::mutex> l(gb_m); gb += now() - t0; }
                const Work w{jaln, slot, info};
                if (workers.empty()) process(cx, w); else wq.push(w);
            } else {
                // no records: the parsed blocks (~100 000 pairs each) are merged on the device into batches of merge_pairs pairs — the kernels
                // that keep a locus' k-mers in LDS want many pairs per locus in a batch (how pairs are cut into batches changes no result)
                const double ta0 = now();
                if (merged ? dbtk_ingest_align_merged(ing, slot, nullptr, merge_pairs, 0) : dbtk_ingest_align(ing, slot, nullptr, sync ? 1 : 0, nullptr, 0, nullptr))
                    die_assert(std::string("align: ") + dbtk_last_error());
                { const double dta = now() - ta0; align_call_s += dta; if (dta > align_call_max) align_call_max = dta; }
                { std::lock_guard<std::mutex> l(gb_m); gb += now() - t0; }
                release();
            }
            nR += 2 * (uint64_t)info.nkept;
            fprintf(stderr, "Buffered reading %llu\t%llu\t%d\n", 2 * (unsigned long long)info.nkept, (unsigned long long)nR, 0);
            if (info.flags) { *resume = lo + info.cut_byte; handed = true; hslot = slot; break; }
            ++jaln;
        }
        const double tl_loop = now() - ts0;
        wq.close();
        for (auto& t : workers) t.join();
        const double tl_workers = now() - ts0;
        if (merged) {  // what is left of the merged batch
            const double t0 = now();
            if (dbtk_ingest_align_merged(ing, ~0u, nullptr, 0, 1)) die_assert(std::string("align: ") + dbtk_last_error());
            flush_s = now() - t0;
            std::lock_guard<std::mutex> l(gb_m); gb += now() - t0;
        }
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto& t : ios) t.join();
        if (piped && handed) {
            // what was taken from the pipe from byte *resume on: the rest of the flagged block (its chunk, and in front of it the bytes
            // carried over from the block before), then the chunks read behind it, in order; the pipe itself goes on from there
            const uint64_t c0 = jaln * (uint64_t)CH;  // input offset of the flagged block's chunk (every chunk before it was full)
            const char* cp = (const char*)dbtk_ingest_chunk_buffer(ing, hslot);
            const uint64_t from = *resume - lo;
            const int64_t rel = (int64_t)from - (int64_t)c0;  // (negative: inside the carried-over bytes in front of the chunk)
            replay->assign(cp + rel, cp + (jaln < chunk_n.size() ? chunk_n[jaln] : 0));
            for (uint64_t j = jaln + 1; j < chunk_n.size() && j <= pipe_cur; ++j)
                if (chunk_n[j]) replay->append((const char*)dbtk_ingest_chunk_buffer(ing, (uint32_t)(j % NS)), chunk_n[j]);
            *pipe_fd = fd;
        } else if (piped && !handed) *resume = hi;
        outq.close();
        if (writer.joinable()) writer.join();
        const double tf0 = now();
        if (verbose_blocks) fprintf(stderr, "batch loop: submitted everything at %.4f, workers done at %.4f, writer done at %.4f\n", tl_loop, tl_workers, tf0 - ts0);
        for (dbtk_ctx_t* wc : wctx) if (dbtk_ctx_synchronize(wc)) die_assert(dbtk_last_error());  // the kernels still in flight
        { std::lock_guard<std::mutex> lk(tot_m); spent_ingests.push_back(ing); }  // (its pinned and device buffers are freed at exit, not inside the batch loop)
        if (!(piped && handed)) close(fd);
        fprintf(stderr, "device reader: %llu blocks of %zu MB on %d reader threads; setup %.3f s, first block parsed after %.3f s, waiting for parsed blocks %.3f s, drain %.3f s; "
                        "align calls %.3f s (longest %.3f s), last merged batch %.3f s%s\n",
                (unsigned long long)jaln, CH >> 20, nio, setup_s, first_s, wait_s, now() - tf0, align_call_s, align_call_max, flush_s, merged ? "" : " (blocks one by one)");
        std::lock_guard<std::mutex> lk(tot_m);
        nReads += nR; read_busy += rb; gpu_busy += gb; write_busy += wb;
        dev_ingest_reads += nR;
    };
    // the ranges
    std::vector<uint64_t> cuts{0, ~0ull};
    {
        struct stat sb;
        uint64_t min_size = 64u << 20;  // below this one pipeline is as good
        if (const char* e = getenv("DBTK_SHARD_MIN")) min_size = strtoull(e, nullptr, 10);  // (tests)
        const int want = std::max(o.ngpus, o.ingestShards);  // ranges asked for: one per GPU, or more (--ingest-shards)
        if (want > 1 && !o.simmode && stat(o.fastxFname.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size > min_size) {
            const uint64_t size = (uint64_t)sb.st_size;
            FILE* f = fopen(o.fastxFname.c_str(), "rb");
            std::vector<uint64_t> c{0};
            std::vector<char> buf(1 << 20);
            for (int i = 1; f && i < want; ++i) {
                // the first record start at or after size * i / N: a line that begins with '>' (2-line FASTA: sequence lines never do),
                // or — FASTQ, where a quality line may begin with '@' — a line beginning with '@' whose second next line begins with '+'
                uint64_t at = size / want * i;
                if (fseeko(f, (off_t)at, SEEK_SET)) break;
                const size_t n = fread(buf.data(), 1, buf.size(), f);
                size_t p = 0, found = n;
                while (p < n && buf[p] != '\n') ++p;  // skip to the end of the line the cut fell into
                ++p;
                while (p < n) {
                    const char* l1 = (const char*)memchr(buf.data() + p, '\n', n - p);
                    if (!l1) break;
                    if (!fq) { if (buf[p] == '>') { found = p; break; } }
                    else if (buf[p] == '@') {
                        const char* l2 = (const char*)memchr(l1 + 1, '\n', n - (l1 + 1 - buf.data()));
                        if (l2 && (size_t)(l2 + 1 - buf.data()) < n && l2[1] == '+') { found = p; break; }
                    }
                    p = l1 + 1 - buf.data();
                }
                if (found == n) { c.clear(); break; }  // (no record start in sight: lines longer than the window) -> one range
                {   // on a PAIR boundary where the file is interleaved: if the record found here carries the title of the record before
                    // it... which is not in the window; equivalently: if records 0 and 1 from here have different titles but 1 and 2 the
                    // same, record 0 is the second mate of the pair the cut fell into -> start one record later.  (The device reader needs
                    // its range to start with a pair; the host reader saves a round through the cross-range pairing.)
                    const size_t Lr = fq ? 4 : 2;
                    size_t tb[3], tl[3], q = found;
                    int have = 0;
                    for (; have < 3 && q < n; ++have) {
                        const char* e = (const char*)memchr(buf.data() + q, '\n', n - q);
                        if (!e) break;
                        tb[have] = q; tl[have] = (size_t)(e - (buf.data() + q));
                        if (tl[have] >= 2 && buf[q + tl[have] - 2] == '/' && (buf[q + tl[have] - 1] == '1' || buf[q + tl[have] - 1] == '2')) tl[have] -= 2;  // prunePEinfo
                        size_t r = q;
                        bool whole = true;
                        for (size_t l = 0; l < Lr; ++l) {
                            const char* e2 = (const char*)memchr(buf.data() + r, '\n', n - r);
                            if (!e2) { whole = false; break; }
                            r = (size_t)(e2 - buf.data()) + 1;
                        }
                        if (!whole) { ++have; break; }
                        q = r;
                    }
                    auto same = [&](int i, int j) { return tl[i] == tl[j] && memcmp(buf.data() + tb[i], buf.data() + tb[j], tl[i]) == 0; };
                    if (have == 3 && !same(0, 1) && same(1, 2)) found = tb[1];
                }
                c.push_back(at + found);
            }
            if (f) fclose(f);
            if ((int)c.size() == want) { c.push_back(size); cuts = c; }
        }
    }
    const int nshards = (int)cuts.size() - 1;
    npipes = nshards;
    nshards_now = nshards;
    // more ranges than GPUs: range i gets a context of its own on GPU i % ngpus (the contexts of a GPU share its tables;
    // an aligner thread and its context belong together: dbtk_align_batch is re-entrant per context, not within one)
    if (!o.parseOnly)
        for (int i = (int)ctx.size(); i < nshards; ++i) {
            ctx.push_back(nullptr);
            if (dbtk_ctx_create(rpgg, &P, dev_of(i % o.ngpus), &ctx[i])) die_assert(dbtk_last_error());
        }
    // the further aligner contexts of -a / -ae (made beside the first one, above)
    if (extra.joinable()) extra.join();
    if (!extra_err.empty()) die_assert(extra_err);
    if (nshards == 1) for (dbtk_ctx_t* c : extra_ctx) ctx.push_back(c);
    else for (dbtk_ctx_t* c : extra_ctx) dbtk_ctx_free(c);
    if (!o.parseOnly && emit_aln && nshards == 1)
        for (int i = (int)ctx.size(); i < aln_aligners * o.ngpus; ++i) {  // (none were made ahead: several GPUs or --ingest-shards, and the input gave one range)
            ctx.push_back(nullptr);
            if (dbtk_ctx_create(rpgg, &P, dev_of(i % o.ngpus), &ctx[i])) die_assert(dbtk_last_error());
        }
    std::vector<std::vector<Left>> lefts(nshards);
    {
        std::vector<std::thread> shards;
        // the device reader first, where it applies: a regular file (pread at offsets), no per-read work the host must do (-s parses
        // titles, -tb replays batches from host copies of the reads, -a / -ae has its own several-contexts-per-GPU emit path)
        struct stat sb;
        const bool is_file = stat(o.fastxFname.c_str(), &sb) == 0 && S_ISREG(sb.st_mode);
        // (-a / -ae: the lines are assembled and gzip-compressed on the device too — a Huffman-only deflate, about zlib's level 1 in size; an
        // explicit --gz-level other than 1 keeps the host's zlib, and with it the host reader)
        bool dev_ingest = !o.parseOnly && !o.hostIngest && !o.simmode && !P.trackbait && (!emit_aln || o.gzLevel == 1) && (is_file || nshards == 1);
        if (const char* e = getenv("DBTK_DEVICE_INGEST")) if (atoi(e) == 0) dev_ingest = false;
        auto one = [&](int i) {
            uint64_t lo = nshards == 1 ? 0 : cuts[i];
            const uint64_t hi = nshards == 1 ? (is_file ? (uint64_t)sb.st_size : ~0ull) : cuts[i + 1];
            if (dev_ingest) {
                uint64_t resume = lo;
                std::string replay;
                int pfd = -1;
                run_device_ingest(ctx[i], lo, hi, &resume, !is_file, &replay, &pfd);
                if (resume >= hi) return;
                fprintf(stderr, "device reader: input is not interleaved at byte %llu; the host reader takes over\n", (unsigned long long)resume);
                lo = resume;
                if (!is_file) {  // the pipe cannot be read again: a stream of what was taken but not paired, then the descriptor
                    struct Cookie { std::string pre; size_t at; int fd; };
                    Cookie* ck = new Cookie{std::move(replay), 0, pfd};
                    cookie_io_functions_t fn;
                    memset(&fn, 0, sizeof(fn));
                    fn.read = [](void* c, char* buf, size_t n) -> ssize_t {
                        Cookie* k = (Cookie*)c;
                        if (k->at < k->pre.size()) { const size_t m2 = std::min(n, k->pre.size() - k->at); memcpy(buf, k->pre.data() + k->at, m2); k->at += m2; return (ssize_t)m2; }
                        for (;;) { const ssize_t r = read(k->fd, buf, n); if (r < 0 && errno == EINTR) continue; return r; }
                    };
                    fn.close = [](void* c) -> int { Cookie* k = (Cookie*)c; close(k->fd); delete k; return 0; };
                    handover_stream = fopencookie(ck, "rb", fn);
                    if (!handover_stream) die_assert("fopencookie failed");
                }
            }
            const auto r = nshards == 1 ? run_shard(0, 0, o.parseOnly ? o.ngpus : (int)ctx.size(), lo, ~0ull, lefts[0]) : run_shard(i, i, 1, lo, hi, lefts[i]);
            std::lock_guard<std::mutex> lk(tot_m);
            nReads += std::get<0>(r); read_busy += std::get<1>(r); cut_busy += std::get<2>(r); pair_busy += std::get<3>(r);
            gpu_busy += std::get<4>(r); write_busy += std::get<5>(r); nsplit_used += std::get<6>(r);
        };
        for (int i = 1; i < nshards; ++i) shards.emplace_back(one, i);
        one(0);
        for (auto& t : shards) t.join();
    }
    if (nshards > 1) {
        // cross-range pairing of what the ranges left over, in range order (the reader's rule: the first record of a title is
        // parked, the next one with that title completes the pair as seq1), then one last batch on GPU 0
        std::unordered_map<std::string, std::pair<std::string, std::string>> parked;
        Batch b;
        uint64_t nleft = 0;
        auto reset_left = [&] { b.flat.clear(); b.off.assign(1, 0); b.qar.clear(); b.qoff.assign(1, 0); b.tar.clear(); b.toff.assign(1, 0); b.src.clear(); b.nreads = 0; b.nrec = 0; };
        // one batch of what the ranges left over through GPU 0 (in chunks of a batch's size: mates far apart in the file — an R1 block
        // followed by an R2 block — leave nearly the whole file here, and one batch of that would neither fit the host nor a 32-bit
        // pair index)
        auto flush_left = [&] {
            nReads += b.nreads; nleft += b.nreads;
            if (b.nreads && !o.parseOnly) {
                const uint64_t npairs = b.nreads / 2;
                std::vector<uint8_t> flatq;
                const bool send_qual = use_bait && fq;
                if (send_qual) {
                    flatq.assign(b.flat.size() + 1, (uint8_t)'!');
                    for (uint64_t r = 0; r < b.nreads; ++r)
                        memcpy(flatq.data() + b.off[r], b.qar.data() + b.qoff[r], std::min(b.qoff[r + 1] - b.qoff[r], b.off[r + 1] - b.off[r]));
                }
                if (want_recs) b.recs.resize(npairs);
                b.flat.push_back(0);
                if (dbtk_align_batch(ctx[0], b.flat.data(), b.off.data(), send_qual ? flatq.data() : nullptr, npairs, want_recs ? b.recs.data() : nullptr,
                                     want_recs ? npairs : 0, &b.nrec)) die_assert(std::string("align: ") + dbtk_last_error());
                b.flat.pop_back();
                if (emit_aln) {
                    uint64_t used = 0;
                    b.aln_idx.resize(npairs);
                    dbtk_status_t sa = dbtk_ctx_aln_text(ctx[0], b.aln_idx.data(), npairs, nullptr, 0, &used);
                    if (sa == DBTK_ERR_OVERFLOW) { b.aln.grow((size_t)used + 16); sa = dbtk_ctx_aln_text(ctx[0], b.aln_idx.data(), npairs, b.aln.data(), b.aln.size(), &used); }
                    if (sa) die_assert(std::string("alignment records: ") + dbtk_last_error());
                    prepare_alignments(b);
                }
                emit(b);
            } else if (b.nreads && o.parseOnly) digest_batch(b);
            reset_left();
        };
        reset_left();
        for (auto& lv : lefts)
            for (auto& r : lv) {
                auto it = parked.find(r.title);
                if (it == parked.end()) { parked[r.title] = std::make_pair(r.seq, r.qual); continue; }
                const std::string s2 = it->second.first, q2 = it->second.second;
                parked.erase(it);
                if (r.seq.size() < minReadSize || s2.size() < minReadSize) continue;
                if (o.simmode) b.src.push_back(parse_src(r.title, o.simmode, nloci));
                b.tar.insert(b.tar.end(), r.title.begin(), r.title.end()); b.toff.push_back(b.tar.size());
                b.add_read(r.seq.data(), r.seq.size(), r.qual.data(), r.qual.size(), fq);
                b.add_read(s2.data(), s2.size(), q2.data(), q2.size(), fq);
                b.nreads += 2;
                if (b.nreads >= readsPerBatch) flush_left();
            }
        flush_left();
        fprintf(stderr, "cross-range pairing: %llu reads\n", (unsigned long long)nleft);
    }
    fflush(stdout);
    const double t_tail0 = now();
    { std::lock_guard<std::mutex> l(ep_m); ep_stop = true; }
    ep_cv.notify_all();
    for (auto& t : emit_pool) t.join();
    if (fflush(stdout) != 0) die_assert("write to stdout failed");
    if (gzout && fclose(gzout) != 0) die_assert("closing the --aln-gz file failed");
    const int nsplit = nsplit_used;
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "batch loop's tail (emit threads joined, outputs closed): %.3f s, from %.3f s\n", now() - t_tail0, t_tail0 - loop_t0);
    if (emit_aln) fprintf(stderr, "emit: record read-back %.2f s; formatting %.2f thread-s, deflate %.2f thread-s on %d emit threads (%u usable CPUs); %llu bytes out; "
                          "%llu bytes of text assembled%s on the device\n",
                          rec_us.load() / 1e6, fmt_us.load() / 1e6, gz_us.load() / 1e6, emit_threads, cpus, (unsigned long long)aln_bytes,
                          (unsigned long long)dev_aln_text.load(), gzout ? " and compressed" : "");
    fprintf(stderr, "ingest: %.2f s for %llu reads (%.2f M reads/s); busy: reading %.2f s, cutting %.2f s, pairing %.2f s, align %.2f s over %d GPU thread(s), write %.2f s; %d splitter threads; device reader: %llu reads\n",
            now() - loop_t0, (unsigned long long)nReads, nReads / (now() - loop_t0) / 1e6, read_busy, cut_busy, pair_busy, gpu_busy, o.ngpus, write_busy, nsplit,
            (unsigned long long)dev_ingest_reads);

    if (o.parseOnly) {
        printf("parse-only\t%llu\t%llu\t%llu\n", (unsigned long long)po_pairs.load(), (unsigned long long)po_bases.load(), (unsigned long long)po_digest.load());
        dbtk_rpgg_free(rpgg);
        return 0;
    }
    // ---- totals + dumps (AQ.cpp:2611-2656)
    const double t_out = wall();
    const int nctx = (int)ctx.size();
    std::vector<uint64_t> counts(dbtk_rpgg_ntrkmers(rpgg)), kmc(nloci), counters(DBTK_C_COUNT);
    std::vector<uint32_t> nmapread(nloci);
    if (nctx == o.ngpus) {  // one context per GPU: the sum is RCCL's (AQ.cpp:2146-2158 across devices)
        if (o.ngpus > 1 && dbtk_allreduce(ctx.data(), o.ngpus)) die_assert(dbtk_last_error());
        if (dbtk_ctx_counts(ctx[0], counts.data(), kmc.data(), nmapread.data(), counters.data())) die_assert(dbtk_last_error());
    } else {  // several contexts per GPU (--ingest-shards): their accumulators are summed on the host
        std::vector<uint64_t> c1(counts.size()), k1(nloci), r1(DBTK_C_COUNT);
        std::vector<uint32_t> n1(nloci);
        for (int d = 0; d < nctx; ++d) {
            if (dbtk_ctx_counts(ctx[d], c1.data(), k1.data(), n1.data(), r1.data())) die_assert(dbtk_last_error());
            for (size_t i = 0; i < counts.size(); ++i) counts[i] += c1[i];
            for (uint64_t l = 0; l < nloci; ++l) { kmc[l] += k1[l]; nmapread[l] += n1[l]; }
            for (int i = 0; i < DBTK_C_COUNT; ++i) counters[i] += r1[i];
        }
    }
    {   // which kernels took the pairs (a diagnostic of this implementation, free-form like the rest of stderr)
        uint64_t ps[DBTK_PATH_STATS] = {0}, one[DBTK_PATH_STATS];
        for (int d = 0; d < nctx; ++d) { const int n = dbtk_ctx_path_stats(ctx[d], one, (int)DBTK_PATH_STATS); for (int i = 0; i < n; ++i) ps[i] += one[i]; }
        fprintf(stderr, "kernel paths: locus-resident probe %llu pairs in %llu items (%llu resolved there, %llu taken back), lean probe %llu pairs; locus-resident walk %llu pairs in %llu items, global walk %llu pairs\n",
                (unsigned long long)(ps[3] + ps[4] + ps[5]), (unsigned long long)(ps[0] + ps[1] + ps[2]), (unsigned long long)ps[14], (unsigned long long)ps[15], (unsigned long long)ps[6],
                (unsigned long long)(ps[10] + ps[11] + ps[12]), (unsigned long long)(ps[7] + ps[8] + ps[9]), (unsigned long long)ps[13]);
    }
    fprintf(stderr,
            "%llu reads processed in total.\n%llu reads removed by subsampled kmer-filter.\n%llu reads removed by kmer-filter.\n"
            "%llu reads removed by bait locus.\n%llu reads removed by qual filter.\n%llu reads removed during locus assignment.\n"
            "%llu reads removed by QC filter.\n%llu reads entered threading step.\n%llu reads passsed threading.\n"
            "%llu reads assigned to TR region.\nparallel query completed in %ld sec.\n",
            (unsigned long long)nReads, (unsigned long long)counters[DBTK_C_SUBFILTERED], (unsigned long long)counters[DBTK_C_KMERFILTERED],
            (unsigned long long)counters[DBTK_C_BAITFILTERED], (unsigned long long)counters[DBTK_C_QUALFILTERED],
            (unsigned long long)counters[DBTK_C_LOCUSFILTERED], (unsigned long long)counters[DBTK_C_QCFILTERED],
            (unsigned long long)counters[DBTK_C_THREADING], (unsigned long long)counters[DBTK_C_FEASIBLE],
            (unsigned long long)counters[DBTK_C_ASGN], (long)(time(nullptr) - time1));
    if (!o.extractFastX) {
        fprintf(stderr, "writing kmers...\n");
        if (dbtk_write_outputs(rpgg, counts.data(), kmc.data(), nmapread.data(), o.outPrefix.c_str(), o.writeKmerName))
            die_assert(dbtk_last_error());
        if (P.trackbait) {  // dumpBaitKmerHits, AQ.cpp:2652-2655
            fprintf(stderr, "writing bait kmer hit statistics...\n");
            for (int d = 1; d < nctx; ++d) if (dbtk_ctx_merge_bait_hits(ctx[0], ctx[d])) die_assert(dbtk_last_error());
            if (dbtk_ctx_write_bait_hits(ctx[0], o.outPrefix.c_str())) die_assert(dbtk_last_error());
        } else if (o.trackBait) {
            // -tb where the bait filter never runs (no -b, or -g): the reference still dumps its tracker — sized nloci when -b
            // read a bait DB (AQ.cpp:2496-2499), empty otherwise
            fprintf(stderr, "writing bait kmer hit statistics...\n");
            const uint64_t nl = o.bait ? nloci : 0, zero = 0, szv = 8;
            FILE* f = fopen((o.outPrefix + ".btk.kmdb").c_str(), "wb");
            if (!f) die_assert("cannot create " + o.outPrefix + ".btk.kmdb");
            fwrite(&nl, 8, 1, f);
            for (uint64_t l = 0; l < nl; ++l) fwrite(&zero, 8, 1, f);
            fwrite(&zero, 8, 1, f); fwrite(&szv, 8, 1, f);
            fclose(f);
        }
        if (o.outputBubbles) {  // dumpBubbles, AQ.cpp:2648-2651
            fprintf(stderr, "writing bubbles...\n");
            if (P.bubbles) {
                for (int d = 1; d < nctx; ++d) if (dbtk_ctx_merge_bubbles(ctx[0], ctx[d])) die_assert(dbtk_last_error());
                if (dbtk_ctx_write_bubbles(ctx[0], o.outPrefix.c_str())) die_assert(dbtk_last_error());
            }
        }
    }
    // Every output is written and closed.  Handing 28 - 47 GB of HBM tables, the pinned buffers and the 3 GB of the handle back piece by
    // piece takes 0.2 - 0.3 s of a run whose batch loop takes as long: the process ends here and the driver reclaims them at once
    // (DBTK_TIDY_EXIT=1: free everything first — leak checkers, make asan).  A process that runs under a tool which flushes its data in
    // a library finalizer or an atexit handler (rocprofv3, gcov, anything preloaded) takes the tidy way by itself: _exit would skip those.
    // (LD_PRELOAD by itself says nothing — the GPU boxes of this pool preload a guard library into every process — only a profiler's name in it does)
    const char* const pre = getenv("LD_PRELOAD");
    const bool pre_tool = pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "roctx") || strstr(pre, "gcov"));
    const bool tidy = getenv("DBTK_TIDY_EXIT") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || pre_tool ||
                      getenv("HSA_TOOLS_LIB") || getenv("GCOV_PREFIX");
    if (tidy) {
        for (auto g : spent_ingests) dbtk_ingest_free(g);
        for (auto c : ctx) dbtk_ctx_free(c);
        dbtk_rpgg_free(rpgg);
    }
    if (getenv("DBTK_VERBOSE")) {  // where the process' wall time went (seconds since main(); the process itself started ~0.03 s earlier)
        timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "timeline: load from %.2f, tables from %.2f, batch loop from %.2f, counts + outputs from %.2f, done at %.2f\n", tl0 - t_main, tl1 - t_main,
                t_loop - t_main, t_out - t_main, t.tv_sec + 1e-9 * t.tv_nsec - t_main);
    }
    fprintf(stderr, "all done!\n");
    const bool out_ok = fflush(stdout) == 0 && !ferror(stdout);  // (records go to stdout: a full disk or a closed pipe must not look like success)
    fflush(stderr);
    if (!out_ok) { fprintf(stderr, "danbing-tk: writing to stdout failed\n"); if (!tidy) _exit(1); return 1; }
    if (!tidy) _exit(0);
    return 0;
}
