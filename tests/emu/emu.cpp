// tests/emu/emu.cpp — TEST-ONLY SPMD emulator for the kernel bodies.
//
// This is not a CPU backend and is not part of the product: libdbtk_hip.so
// neither contains nor can reach it.  It exists so that the exact kernel
// bodies of danbing-tk_amd/csrc/dbtk_kernels.h (the code hipcc compiles for
// gfx950) can be run in this GPU-less container against the oracle before GPU
// minutes are spent: each GPU thread becomes a coroutine (a hand-rolled x86-64 stack switch:
// glibc's swapcontext makes a sigprocmask system call per switch), a block is
// scheduled round-robin from barrier to barrier, wave64 collectives (ballot,
// scans, broadcasts) go through a scratch array, atomics are plain operations.
// Blocks run one after another, which is one of the schedules the GPU may pick.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if !defined(__x86_64__)
#error "the test emulator's coroutine switch is written for x86-64"
#endif

#include <algorithm>
#include <functional>
#include <vector>

#include "../../danbing-tk_amd/csrc/dbtk_internal.h"
#include "../../danbing-tk_amd/csrc/dbtk_kernels.h"
#include "../../danbing-tk_amd/csrc/dbtk_ingest.h"
#include "../../danbing-tk_amd/csrc/dbtk_gz.h"

using namespace dbtk;

// emu_switch(&save_sp, load_sp): push the callee-saved registers, park this stack's pointer in *save_sp,
// continue on the stack whose parked pointer is load_sp.
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace {

struct EmuBlock;
struct EmuX {
    EmuBlock* b;
    int t;
    int tid() const { return t; }
    int nthreads() const;
    uint32_t bid() const;
    uint32_t nblocks() const;
    int lane() const { return t & 63; }
    void sync() const;
    void bsync() const;  // workgroup barrier (the waves of a block make different numbers of yields between two of them)
    uint64_t ballot(bool p) const;
    uint32_t wave_sum(uint32_t v) const;
    uint32_t wave_min(uint32_t v) const;
    uint32_t half_sum(uint32_t v) const;
    uint32_t half_max(uint32_t v) const;
    uint32_t wave_scan_max(uint32_t v) const;
    uint32_t half_scan_max(uint32_t v) const;
    uint32_t wave_excl_scan(uint32_t v) const;
    uint32_t bcast(uint32_t v, int src) const;
    uint32_t wave_scan_lastnz(uint32_t v) const;
    uint32_t shfl_up1(uint32_t v) const;
    uint32_t shfl_down1(uint32_t v) const;
    template <int P0, int P1, int P2, int P3> uint32_t quad_perm(uint32_t v) const;
    template <int E> void shfl_xor64(const uint64_t (&in)[E], uint64_t (&out)[E], int mask) const;
    void atomic_add(uint64_t* p, uint64_t v) const { *p += v; }
    uint32_t atomic_add(uint32_t* p, uint32_t v) const { uint32_t o = *p; *p += v; return o; }
    uint64_t atomic_cas(uint64_t* p, uint64_t e, uint64_t d) const { uint64_t o = *p; if (o == e) *p = d; return o; }
    uint32_t atomic_cas32(uint32_t* p, uint32_t e, uint32_t d) const { uint32_t o = *p; if (o == e) *p = d; return o; }
    void atomic_max(uint64_t* p, uint64_t v) const { if (v > *p) *p = v; }
    void atomic_or(uint64_t* p, uint64_t v) const { *p |= v; }
    uint32_t atomic_or32(uint32_t* p, uint32_t v) const { uint32_t o = *p; *p |= v; return o; }
    uint64_t clock() const { return 0; }
    uint32_t uni(uint32_t v) const { return v; }
    uint32_t lds_add(uint32_t* p, uint32_t v) const { uint32_t o = *p; *p += v; return o; }
    void lds_or(uint32_t* p, uint32_t v) const { *p |= v; }
    void lds_max(uint32_t* p, uint32_t v) const { if (v > *p) *p = v; }
    void lds_min(uint32_t* p, uint32_t v) const { if (v < *p) *p = v; }
    template <class T> T* smem() const;
};

struct EmuBlock {
    int nt = 0;
    uint32_t bid = 0, nblocks = 0;
    std::vector<uint64_t> smem;  // 8-byte aligned backing store
    void* mainsp = nullptr;
    std::vector<void*> sp;  // parked stack pointer per lane
    std::vector<std::vector<char>> stacks;
    std::vector<char> done;
    std::vector<uint64_t> scratch, vscratch;
    int cur = 0;
    int bar_count = 0, ndone = 0;
    uint64_t bar_gen = 0;
    std::function<void(EmuX&)> body;

    void yield() { emu_switch(&sp[cur], mainsp); }
};

thread_local EmuBlock* g_blk = nullptr;

void lane_entry() {
    EmuBlock* b = g_blk;
    const int t = b->cur;
    EmuX x{b, t};
    b->body(x);
    b->done[t] = 1;
    ++b->ndone;
    b->scratch[t] = 0;
    emu_switch(&b->sp[t], b->mainsp);  // never resumed
    __builtin_trap();
}

int EmuX::nthreads() const { return b->nt; }
uint32_t EmuX::bid() const { return b->bid; }
uint32_t EmuX::nblocks() const { return b->nblocks; }
void EmuX::sync() const { b->yield(); }
void EmuX::bsync() const {
    const uint64_t gen = b->bar_gen;
    if (++b->bar_count >= b->nt - b->ndone) { b->bar_count = 0; ++b->bar_gen; }
    else while (b->bar_gen == gen) b->yield();
    b->yield();
}
template <class T> T* EmuX::smem() const { return reinterpret_cast<T*>(b->smem.data()); }
uint64_t EmuX::ballot(bool p) const {
    b->scratch[t] = p;
    b->yield();
    uint64_t m = 0;
    const int w0 = t & ~63;
    for (int l = 0; l < 64 && w0 + l < b->nt; ++l)
        if (!b->done[w0 + l] && b->scratch[w0 + l]) m |= 1ull << l;
    b->yield();
    return m;
}
uint32_t EmuX::wave_sum(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t s = 0;
    const int w0 = t & ~63;
    for (int l = 0; l < 64 && w0 + l < b->nt; ++l) s += (uint32_t)b->scratch[w0 + l];
    b->yield();
    return s;
}
uint32_t EmuX::wave_min(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t s = 0xFFFFFFFFu;
    const int w0 = t & ~63;
    for (int l = 0; l < 64 && w0 + l < b->nt; ++l) if (!b->done[w0 + l] && (uint32_t)b->scratch[w0 + l] < s) s = (uint32_t)b->scratch[w0 + l];
    b->yield();
    return s;
}
uint32_t EmuX::half_sum(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t s = 0;
    const int h0 = t & ~31;
    for (int l = 0; l < 32 && h0 + l < b->nt; ++l) s += (uint32_t)b->scratch[h0 + l];
    b->yield();
    return s;
}
uint32_t EmuX::half_max(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t s = 0;
    const int h0 = t & ~31;
    for (int l = 0; l < 32 && h0 + l < b->nt; ++l) if ((uint32_t)b->scratch[h0 + l] > s) s = (uint32_t)b->scratch[h0 + l];
    b->yield();
    return s;
}
uint32_t EmuX::wave_scan_max(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t r = 0;
    const int w0 = t & ~63;
    for (int l = 0; l <= (t & 63); ++l) if ((uint32_t)b->scratch[w0 + l] > r) r = (uint32_t)b->scratch[w0 + l];
    b->yield();
    return r;
}
uint32_t EmuX::half_scan_max(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t r = 0;
    const int h0 = t & ~31;
    for (int l = 0; l <= (t & 31); ++l) if ((uint32_t)b->scratch[h0 + l] > r) r = (uint32_t)b->scratch[h0 + l];
    b->yield();
    return r;
}
uint32_t EmuX::wave_excl_scan(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t s = 0;
    const int w0 = t & ~63;
    for (int l = 0; l < (t & 63); ++l) s += (uint32_t)b->scratch[w0 + l];
    b->yield();
    return s;
}
template <int E> void EmuX::shfl_xor64(const uint64_t (&in)[E], uint64_t (&out)[E], int mask) const {
    for (int j = 0; j < E; ++j) b->vscratch[(size_t)t * 8 + j] = in[j];
    b->yield();
    for (int j = 0; j < E; ++j) out[j] = b->vscratch[(size_t)(t ^ mask) * 8 + j];
    b->yield();
}
uint32_t EmuX::wave_scan_lastnz(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    uint32_t r = 0;
    const int w0 = t & ~63;
    for (int l = 0; l <= (t & 63); ++l) if (b->scratch[w0 + l]) r = (uint32_t)b->scratch[w0 + l];
    b->yield();
    return r;
}
uint32_t EmuX::shfl_up1(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    const uint32_t r = (t & 63) ? (uint32_t)b->scratch[t - 1] : v;
    b->yield();
    return r;
}
uint32_t EmuX::shfl_down1(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    const uint32_t r = ((t & 63) != 63 && t + 1 < b->nt) ? (uint32_t)b->scratch[t + 1] : v;
    b->yield();
    return r;
}
template <int P0, int P1, int P2, int P3> uint32_t EmuX::quad_perm(uint32_t v) const {
    b->scratch[t] = v;
    b->yield();
    const int p[4] = {P0, P1, P2, P3};
    const uint32_t r = (uint32_t)b->scratch[(t & ~3) + p[t & 3]];
    b->yield();
    return r;
}
uint32_t EmuX::bcast(uint32_t v, int src) const {
    b->scratch[t] = v;
    b->yield();
    const uint32_t r = (uint32_t)b->scratch[(t & ~63) + src];
    b->yield();
    return r;
}

void run_grid(uint32_t nblocks, int nt, size_t smem_bytes, std::function<void(EmuX&)> body) {
    const size_t STK = 256 * 1024;
    EmuBlock b;
    b.nt = nt;
    b.nblocks = nblocks;
    b.body = body;
    b.sp.resize(nt);
    static thread_local std::vector<std::vector<char>> stack_pool;  // kept across calls: the selftests launch thousands of grids
    if ((int)stack_pool.size() < nt) stack_pool.resize(nt);
    for (int t = 0; t < nt; ++t) if (stack_pool[t].size() < STK) stack_pool[t].resize(STK);
    b.stacks.clear();
    b.done.assign(nt, 0);
    b.scratch.assign(nt, 0);
    b.vscratch.assign((size_t)nt * 8, 0);
    g_blk = &b;
    for (uint32_t bid = 0; bid < nblocks; ++bid) {
        b.bid = bid;
        b.smem.assign(smem_bytes / 8 + 2, 0xA5A5A5A5A5A5A5A5ull);  // LDS is NOT zeroed on the GPU either
        b.bar_count = 0; b.ndone = 0;
        for (int t = 0; t < nt; ++t) {
            b.done[t] = 0;
            b.scratch[t] = 0;
            // fresh stack: six zeroed callee-saved registers, then lane_entry as the "return address" (16-byte aligned slot,
            // so that lane_entry starts with the stack alignment of a called function)
            uintptr_t top = ((uintptr_t)stack_pool[t].data() + STK) & ~(uintptr_t)15;
            void** slot = (void**)(top - 16);
            slot[0] = (void*)lane_entry;
            for (int r = 1; r <= 6; ++r) slot[-r] = nullptr;
            b.sp[t] = (void*)(slot - 6);
        }
        bool alive = true;
        while (alive) {
            alive = false;
            for (int t = 0; t < nt; ++t) {
                if (b.done[t]) continue;
                b.cur = t;
                emu_switch(&b.mainsp, b.sp[t]);
                if (!b.done[t]) alive = true;
            }
        }
    }
    g_blk = nullptr;
}

struct EmuTables {
    std::vector<IdxBucket> idx;
    std::vector<ClsSlot> cls;
    std::vector<uint32_t> vv;
    std::vector<uint16_t> perm;
    std::vector<uint32_t> trbeg;
    std::vector<uint64_t> flt;
    std::vector<ClsSlot> tre, bait;
    std::vector<GrSlot> gr;
    std::vector<MzBucket> mz, grmz;
    std::vector<MzSlot> ovf;
    std::vector<LocusDir> ldir, gldir;
    std::vector<uint64_t> limg, glimg;  // (8-byte words: the images are read 16 bytes at a time)
    uint64_t stats[3] = {0, 0, 0};
    DevTables T;
};

}  // namespace

static uint64_t g_probe_runs[2] = {0, 0}, g_mz_turned = 0, g_walk_fast_runs = 0, g_walk_slow_pairs = 0;
static uint64_t g_wfl_pairs[2] = {0, 0};  // pairs the lean walk kernel took in its locus-resident form / left to its plain form
static uint64_t g_loc_left = 0;  // keys the last tables' images left out
static uint64_t g_pstats[DBTK_PATH_STATS] = {0};  // as dbtk_ctx_path_stats, over the emu_align calls since the last emu_path_stats
static uint64_t g_loc_pairs[4] = {0, 0, 0, 0};  // pairs the locus-resident kernel took in its three classes of workgroup, pairs left to the lean kernel
extern "C" {

void* emu_tables_create(const dbtk_rpgg_t* g) {
    EmuTables* e = new EmuTables;
    auto pow2 = [](uint64_t n) { uint64_t c = 1024; while (c < n) c <<= 1; return c; };
    auto lg = [](uint64_t c) { return 63u - (uint32_t)__builtin_clzll(c); };
    const uint64_t nloci = g->nloci;
    const uint64_t icap = pow2(2 * g->keys.size() + 8), nbkt = icap / 4;
    e->idx.assign(nbkt, IdxBucket{{NAN64, NAN64, NAN64, NAN64}, {0, 0, 0, 0}});
    {
        IdxBuildArgs a{e->idx.data(), nbkt - 1, 64 - lg(nbkt), g->keys.data(), g->vals.data(), g->keys.size()};
        run_grid(3, 64, 0, [&](EmuX& x) { body_idx_insert(x, a); });
        run_grid(3, 64, 0, [&](EmuX& x) { body_idx_finalize(x, e->idx.data(), icap); });
        // presence filter (deliberately small here: many false positives exercise the "maybe" path)
        e->flt.assign(pow2((2 * g->keys.size() + 63) / 64) / 1024 ? pow2((2 * g->keys.size() + 63) / 64) : 1024, 0);
        FltBuildArgs fa{e->flt.data(), (uint32_t)(63 - __builtin_clzll(e->flt.size())), g->ksize, g->keys.data(), g->keys.size()};
        run_grid(3, 64, 0, [&](EmuX& x) { body_flt_insert(x, fa); });
    }
    e->vv = g->vv;
    e->vv.push_back(0);
    const uint64_t ccap = pow2(2 * (g->tr_ks.size() + g->fl_ks.size()) + 2);
    e->cls.assign(ccap, ClsSlot{NAN64, ~0ull});
    {
        std::vector<uint64_t> beg(nloci + 1, 0);
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->tr_cnt[l];
        ClsBuildArgs a{e->cls.data(), ccap - 1, 64 - lg(ccap), g->tr_ks.data(), beg.data(), (uint32_t)nloci, g->out_slot.data(), g->tr_ks.size(), &e->stats[2]};
        run_grid(3, 64, 0, [&](EmuX& x) { body_cls_insert(x, a); });
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->fl_cnt[l];
        ClsBuildArgs f{e->cls.data(), ccap - 1, 64 - lg(ccap), g->fl_ks.data(), beg.data(), (uint32_t)nloci, nullptr, g->fl_ks.size(), &e->stats[2]};
        run_grid(3, 64, 0, [&](EmuX& x) { body_cls_insert(x, f); });
    }
    e->perm.resize((size_t)NHMAX * (NHMAX + 1) / 2 + 1);
    {
        std::vector<uint32_t> key(NHMAX, 1);
        int stack[3 * 40];
        for (int n = 1; n <= NHMAX; ++n) gcc_sort_index(e->perm.data() + (size_t)n * (n - 1) / 2, n, key.data(), stack);
    }
    e->trbeg.assign(nloci + 1, 0);
    for (uint64_t l = 0; l <= nloci; ++l) e->trbeg[l] = (uint32_t)g->out_beg[l];
    DevTables& T = e->T;
    memset(&T, 0, sizeof(T));  // optional tables (tre, bait, gr, mz) stay null unless built
    T.trbeg = e->trbeg.data();
    T.flt = e->flt.data(); T.flt_logw = (uint32_t)(63 - __builtin_clzll(e->flt.size()));
    T.idx = e->idx.data(); T.idx_mask = nbkt - 1; T.idx_shift = 64 - lg(nbkt);
    T.vv = e->vv.data();
    T.cls = e->cls.data(); T.cls_mask = ccap - 1; T.cls_shift = 64 - lg(ccap);
    T.qc = g->qc.empty() ? nullptr : g->qc.data();
    T.permtab = e->perm.data();
    T.nloci = (uint32_t)nloci;
    T.ksize = g->ksize;
    T.consistent = 0;
    auto kl_table = [&](const std::vector<uint64_t>& cnt, const std::vector<uint64_t>& ks, const std::vector<uint16_t>* vals,
                        std::vector<ClsSlot>& tab, const ClsSlot** out, uint64_t* mask, uint32_t* shift) {
        const uint64_t cap = pow2(2 * ks.size() + 2);
        tab.assign(cap, ClsSlot{NAN64, ~0ull});
        std::vector<uint64_t> beg(nloci + 1, 0), v64(ks.size(), 0);
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + cnt[l];
        if (vals) for (size_t i = 0; i < ks.size(); ++i) v64[i] = (*vals)[i];
        uint64_t nent = 0;
        ClsBuildArgs b{tab.data(), cap - 1, 64 - lg(cap), ks.data(), beg.data(), (uint32_t)nloci, v64.data(), ks.size(), &nent};
        if (!ks.empty()) run_grid(3, 64, 0, [&](EmuX& x) { body_cls_insert(x, b); });
        *out = tab.data(); *mask = cap - 1; *shift = 64 - lg(cap);
    };
    if (!g->tre_cnt.empty()) kl_table(g->tre_cnt, g->tre_ks, nullptr, e->tre, &T.tre, &T.tre_mask, &T.tre_shift);
    if (!g->bt_cnt.empty()) kl_table(g->bt_cnt, g->bt_ks, &g->bt_vs, e->bait, &T.bait, &T.bait_mask, &T.bait_shift);
    {
        IdxAuxArgs a{e->idx.data(), icap, T, e->stats};
        run_grid(3, 64, 0, [&](EmuX& x) { body_idx_aux(x, a); });
        T.consistent = (e->stats[1] == 0 && e->stats[0] == e->stats[2]) ? 1u : 0u;
    }
    // the probe body's minimizer-grouped copy of the index (small: many turned-away keys exercise level 2), built as
    // build_tables does on the device: level 1 + count, then the overflow table sized from the count
    if (mz_m_for_k(g->ksize) && (!getenv("DBTK_MZ") || atoi(getenv("DBTK_MZ")))) {
        const uint64_t nb = pow2((g->keys.size() * 2) / 8 + 8) / 1024 ? pow2((g->keys.size() * 2) / 8 + 8) : 1024;
        MzBucket empty;
        for (int j = 0; j < 8; ++j) { empty.key[j] = MZ_EMPTY; empty.pl[j].val = empty.pl[j].aux = 0; }
        e->mz.assign(nb, empty);
        uint64_t nturned = 0;
        MzBuildArgs a{e->idx.data(), icap, e->mz.data(), (uint32_t)(nb - 1), nullptr, 0, g->ksize, mz_m_for_k(g->ksize), 0, &nturned};
        run_grid(3, 64, 0, [&](EmuX& x) { body_mz_insert(x, a); });
        g_mz_turned = nturned;
        const uint64_t ocap = pow2(2 * nturned + 8);  // (fuller than on the device: longer probe sequences get exercised)
        e->ovf.assign(ocap, MzSlot{MZ_EMPTY, 0, 0});
        a.ovf = e->ovf.data(); a.ovf_mask = (uint32_t)(ocap - 1); a.pass = 1;
        run_grid(3, 64, 0, [&](EmuX& x) { body_mz_insert(x, a); });
        T.mz = e->mz.data(); T.mz_mask = nb - 1; T.mz_m = a.m;
        T.ovf = e->ovf.data(); T.ovf_mask = ocap - 1;
    }
    // per-locus images of the index (dbtk_locus.h), as build_locus_images makes them on the device; every third locus goes without
    // one here, so that every batch also exercises the hand-over to the global-table kernel
    if (T.mz && T.consistent && nloci && loc_lg_min(g->ksize) <= LOC_LG_MAX && (!getenv("DBTK_LOCUS") || atoi(getenv("DBTK_LOCUS")))) {
        std::vector<uint32_t> cnt(nloci, 0), bad(nloci, 0);
        LocBuildArgs a;
        memset(&a, 0, sizeof(a));
        a.idx = e->idx.data(); a.nslots = icap; a.vv = e->vv.data(); a.trbeg = e->trbeg.data(); a.nloci = (uint32_t)nloci; a.ksize = g->ksize;
        a.cls = T.cls; a.cls_mask = T.cls_mask; a.cls_shift = T.cls_shift;
        a.cnt = cnt.data(); a.bad = bad.data();
        run_grid(3, 64, 0, [&](EmuX& x) { body_loc_count(x, a); });
        e->ldir.assign(nloci, LocusDir{0, 0, 0, 0});
        std::vector<uint64_t> ebeg(nloci + 1, 0);
        uint64_t at = 0;
        for (uint64_t l = 0; l < nloci; ++l) {
            const uint32_t lg = loc_lgnb_for(cnt[l], g->ksize);
            e->ldir[l] = LocusDir{(uint32_t)(at / 16), 0u, lg, e->trbeg[l]};
            ebeg[l + 1] = ebeg[l];
            if (!cnt[l] || lg > LOC_LG_MAX || cnt[l] > 0xFFF0u || l % 3 == 2) continue;
            e->ldir[l].bytes = loc_image_bytes(lg);
            at += e->ldir[l].bytes;
            ebeg[l + 1] += cnt[l];
        }
        e->limg.assign(at / 8 + 2, 0);
        const uint64_t nent = ebeg[nloci];
        const uint32_t gstride = 2 * (1u << LOC_LG_MAX) + 2;
        std::vector<uint64_t> ekey(nent + 1), skey(nent + 1);
        std::vector<uint32_t> epay(nent + 1), spay(nent + 1), ecur(nloci, 0);
        std::vector<uint16_t> gscr(nloci * (size_t)gstride, 0);
        uint64_t nleft = 0;
        a.dir = e->ldir.data(); a.arena = reinterpret_cast<uint8_t*>(e->limg.data());
        a.ebeg = ebeg.data(); a.ecur = ecur.data(); a.ekey = ekey.data(); a.epay = epay.data(); a.skey = skey.data(); a.spay = spay.data();
        a.gscr = gscr.data(); a.gstride = gstride; a.nleft = &nleft;
        run_grid(3, 64, 0, [&](EmuX& x) { body_loc_scatter(x, a); });
        if (getenv("EMU_LOC_PLACE_THREAD")) run_grid(2, 64, 0, [&](EmuX& x) { body_loc_place(x, a); });  // (the thread-per-locus form)
        else run_grid(3, 64, sizeof(LocPlaceSmemT<LOC_LG_MAX>), [&](EmuX& x) { body_loc_place_wave<LOC_LG_MAX, -1>(x, a); });
        g_loc_left = nleft;
        for (uint64_t l = 0; l < nloci; ++l) if (bad[l]) e->ldir[l].bytes = 0;
        T.ldir = e->ldir.data(); T.limg = reinterpret_cast<const uint8_t*>(e->limg.data());
    }
    if (!g->gr_cnt.empty()) {  // graph table: graph pass, then TR pass (as build_graph_table does on the device)
        const uint64_t ngr = g->gr_ks.size(), ntrf = g->tr_ks.size();
        const uint64_t cap = pow2(ngr + ngr / 2 + 2 * ntrf + 2);
        e->gr.assign(cap, GrSlot{NAN64, ~0ull});
        std::vector<uint64_t> beg(nloci + 1, 0);
        uint64_t nent = 0;
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->gr_cnt[l];
        GrBuildArgs a{e->gr.data(), cap - 1, 64 - lg(cap), g->ksize, g->gr_ks.data(), g->gr_ms.data(), beg.data(), (uint32_t)nloci, nullptr, e->trbeg.data(), ngr, &nent};
        if (ngr) run_grid(3, 64, 0, [&](EmuX& x) { body_gr_insert(x, a); });
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->tr_cnt[l];
        a.ks = g->tr_ks.data(); a.ms = nullptr; a.outslot = g->out_slot.data(); a.n = ntrf;
        if (ntrf) run_grid(3, 64, 0, [&](EmuX& x) { body_gr_insert(x, a); });
        T.gr = e->gr.data(); T.gr_mask = cap - 1; T.gr_shift = 64 - lg(cap);
        // its minimizer-grouped copy (small: many turned-away entries exercise the single look-ups), as build_graph_table makes it
        if (mz_m_for_k(g->ksize) && (!getenv("DBTK_MZ") || atoi(getenv("DBTK_MZ")))) {
            const uint64_t nb = pow2((ngr + ntrf) / 8 + 8) / 64 ? pow2((ngr + ntrf) / 8 + 8) : 64;
            MzBucket empty;
            for (int j = 0; j < 8; ++j) { empty.key[j] = MZ_EMPTY; empty.pl[j].val = empty.pl[j].aux = 0; }
            e->grmz.assign(nb, empty);
            GrMzBuildArgs ga{e->gr.data(), cap, e->grmz.data(), (uint32_t)(nb - 1), g->ksize, mz_m_for_k(g->ksize)};
            run_grid(3, 64, 0, [&](EmuX& x) { body_grmz_insert(x, ga); });
            T.grmz = e->grmz.data(); T.grmz_mask = nb - 1;
        }
        // the graph images (as build_graph_images on the device); every third locus without one here, as for the index images
        if (T.grmz && loc_lg_min(g->ksize) <= LOC_LG_MAX && (!getenv("DBTK_LOCUS") || atoi(getenv("DBTK_LOCUS")))) {
            std::vector<uint32_t> cnt(nloci, 0), bad(nloci, 0);
            LocBuildArgs a;
            memset(&a, 0, sizeof(a));
            a.gr = e->gr.data(); a.gr_nslots = cap; a.trbeg = e->trbeg.data(); a.nloci = (uint32_t)nloci; a.ksize = g->ksize;
            a.cnt = cnt.data(); a.bad = bad.data();
            run_grid(3, 64, 0, [&](EmuX& x) { body_gloc_count(x, a); });
            e->gldir.assign(nloci, LocusDir{0, 0, 0, 0});
            std::vector<uint64_t> ebeg(nloci + 1, 0);
            uint64_t at = 0;
            for (uint64_t l = 0; l < nloci; ++l) {
                const uint32_t lg = loc_lgnb_for(cnt[l], g->ksize);
                e->gldir[l] = LocusDir{(uint32_t)(at / 16), 0u, lg, e->trbeg[l]};
                ebeg[l + 1] = ebeg[l];
                if (!cnt[l] || lg > LOC_LG_MAX || cnt[l] > 0xFFF0u || l % 3 == 1 || e->trbeg[l + 1] - e->trbeg[l] >= GLOC_SLOT_MAX) continue;
                e->gldir[l].bytes = loc_image_bytes(lg);
                at += e->gldir[l].bytes;
                ebeg[l + 1] += cnt[l];
            }
            e->glimg.assign(at / 8 + 2, 0);
            const uint64_t nent = ebeg[nloci];
            const uint32_t gstride = 2 * (1u << LOC_LG_MAX) + 2;
            std::vector<uint64_t> ekey(nent + 1), skey(nent + 1);
            std::vector<uint32_t> epay(nent + 1), spay(nent + 1), ecur(nloci, 0);
            std::vector<uint16_t> gscr(nloci * (size_t)gstride, 0);
            uint64_t nleft = 0;
            a.dir = e->gldir.data(); a.arena = reinterpret_cast<uint8_t*>(e->glimg.data());
            a.ebeg = ebeg.data(); a.ecur = ecur.data(); a.ekey = ekey.data(); a.epay = epay.data(); a.skey = skey.data(); a.spay = spay.data();
            a.gscr = gscr.data(); a.gstride = gstride; a.nleft = &nleft;
            run_grid(3, 64, 0, [&](EmuX& x) { body_gloc_scatter(x, a); });
            if (getenv("EMU_LOC_PLACE_THREAD")) run_grid(2, 64, 0, [&](EmuX& x) { body_loc_place(x, a); });  // (the thread-per-locus form)
        else run_grid(3, 64, sizeof(LocPlaceSmemT<LOC_LG_MAX>), [&](EmuX& x) { body_loc_place_wave<LOC_LG_MAX, -1>(x, a); });
            for (uint64_t l = 0; l < nloci; ++l) if (bad[l] || nleft) e->gldir[l].bytes = 0;
            T.gldir = e->gldir.data(); T.glimg = reinterpret_cast<const uint8_t*>(e->glimg.data());
        }
    }
    return e;
}

// dbtk_thread_batch on the emulated lanes (body_walk_reads, dbtk_walk.h)
int emu_thread_batch(const dbtk_rpgg_t* g, void* tables, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
                     const uint32_t* loci, uint64_t nreads, dbtk_thread_rec_t* recs, uint32_t grid) {
    EmuTables* e = (EmuTables*)tables;
    if (!e->T.gr) return -1;
    const uint64_t nbytes = off[nreads];
    std::vector<uint64_t> seqbuf(nbytes / 8 + 8, 0);
    memcpy(seqbuf.data(), seq, nbytes);
    uint32_t err = 0;
    WalkArgs w;
    memset(&w, 0, sizeof(w));
    w.T = e->T; w.P = *p; w.seq = (const uint8_t*)seqbuf.data(); w.off = off;
    w.read_locus = loci; w.nreads = (uint32_t)nreads; w.trecs = recs; w.errflag = &err;
    run_grid(grid ? grid : 1, 64, sizeof(WalkReadSmem), [&](EmuX& x) { body_walk_reads(x, w); });
    return (int)err;
}
void emu_tables_free(void* e) { delete (EmuTables*)e; }
uint32_t emu_tables_consistent(void* e) { return ((EmuTables*)e)->T.consistent; }
void emu_tables_set_consistent(void* e, uint32_t v) { ((EmuTables*)e)->T.consistent = v; }

// gcc_sort (index form and packed form, the code the kernels run) against the host's real
// std::sort with getSortedIndex's comparator, on tie-heavy, sorted, reversed and
// median-of-3-killer key sequences.  Returns the number of mismatching runs.
uint64_t emu_selftest_sort(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int stack[3 * 40];
    for (uint64_t it = 0; it < iters; ++it) {
        const int n = (int)(rnd() % 513);
        std::vector<uint32_t> key(n ? n : 1);
        const int mode = (int)(rnd() % 6);
        const uint32_t hi = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : mode == 3 ? 50 : 100000;
        for (int i = 0; i < n; ++i) key[i] = 1 + (uint32_t)(rnd() % hi);
        if (mode == 4) std::sort(key.begin(), key.begin() + n);
        if (mode == 5 && n >= 4) {  // median-of-3 killer: drives introsort into its heapsort fallback
            const int k2 = n / 2;
            std::fill(key.begin(), key.end(), 0u);
            for (int i = 1; i <= k2; ++i) {
                if (i % 2) { key[i - 1] = (uint32_t)i; key[i] = (uint32_t)(k2 + i); }
                key[k2 + i - 1] = (uint32_t)(2 * i);
            }
        }
        std::vector<uint64_t> ref(n);
        for (int i = 0; i < n; ++i) ref[i] = (uint64_t)i;
        std::sort(ref.begin(), ref.end(), [&](uint64_t a, uint64_t b) { return key[a] < key[b]; });
        std::vector<uint16_t> a(n ? n : 1);
        gcc_sort_index(a.data(), n, key.data(), stack);
        std::vector<uint32_t> pk(n ? n : 1);
        for (int i = 0; i < n; ++i) pk[i] = (key[i] << 9) | (uint32_t)i;
        gcc_sort(pk.data(), n, PackedLt{}, stack);
        bool ok = true;
        for (int i = 0; i < n; ++i) ok &= a[i] == ref[i] && (pk[i] & 0x1FF) == ref[i];
        bad += !ok;
    }
    return bad;
}

// wave_gcc_sort_packed (64 emulated lanes) against the host's real std::sort, same key families.
uint64_t emu_selftest_wavesort(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint64_t it = 0; it < iters; ++it) {
        const int n = (int)(rnd() % 385);
        std::vector<uint32_t> key(n ? n : 1);
        const int mode = (int)(rnd() % 7);
        const uint32_t hi = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : mode == 3 ? 50 : 100000;
        for (int i = 0; i < n; ++i) key[i] = 1 + (uint32_t)(rnd() % hi);
        if (mode == 4) std::sort(key.begin(), key.begin() + n);
        if (mode == 6) { std::sort(key.begin(), key.begin() + n); std::reverse(key.begin(), key.begin() + n); }
        if (mode == 5 && n >= 4) {  // median-of-3 killer: drives introsort into its heapsort fallback
            const int k2 = n / 2;
            std::fill(key.begin(), key.end(), 0u);
            for (int i = 1; i <= k2; ++i) {
                if (i % 2) { key[i - 1] = (uint32_t)i; key[i] = (uint32_t)(k2 + i); }
                key[k2 + i - 1] = (uint32_t)(2 * i);
            }
        }
        std::vector<uint64_t> ref(n);
        for (int i = 0; i < n; ++i) ref[i] = (uint64_t)i;
        std::sort(ref.begin(), ref.end(), [&](uint64_t a, uint64_t b) { return key[a] < key[b]; });
        std::vector<uint32_t> pk(n + 1), out(n + 1, 0);
        std::vector<uint16_t> ap(n + 1), bp(n + 1);
        std::vector<int> stack(3 * 40);
        for (int i = 0; i < n; ++i) pk[i] = (key[i] << 9) | (uint32_t)i;
        run_grid(1, 64, 0, [&](EmuX& x) { wave_gcc_sort_packed(x, pk.data(), n, ap.data(), bp.data(), out.data(), stack.data()); });
        bool ok = true;
        for (int i = 0; i < n; ++i) ok &= (out[i] & 0x1FF) == ref[i];
        bad += !ok;
    }
    return bad;
}

// vote_parallel (64 emulated lanes) against the literal find_matching_locus loop vote(), on random tie-heavy inputs:
// few loci, small dups, multi-locus k-mers, random vote order and threshold.  Returns the number of mismatches.
uint64_t emu_selftest_vote(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    constexpr int NH = 384, LC = 512;
    for (uint64_t it = 0; it < iters; ++it) {
        const int nu = 1 + (int)(rnd() % (it % 5 == 0 ? 300 : 60));
        const int D = 1 + (int)(rnd() % 6);
        const uint32_t cth = (uint32_t)(rnd() % 60);
        std::vector<uint16_t> ord(NH), poff(NH, 0xFFFF);
        std::vector<uint32_t> uval(NH, 0), dd(NH, 0), nml(NH, 1), pool(NH, 0), vv;
        for (int i = 0; i < nu; ++i) ord[i] = (uint16_t)i;
        for (int i = nu - 1; i > 0; --i) std::swap(ord[i], ord[rnd() % (i + 1)]);
        uint32_t at = 0;
        for (int u = 0; u < nu; ++u) {
            const uint32_t d1 = (uint32_t)(rnd() % 4), d2 = (uint32_t)(rnd() % 4);
            dd[u] = (d1 | (d2 << 16)) ? (d1 | (d2 << 16)) : 1u;
            dd[u] |= (uint32_t)(rnd() % 2) << 8 << 0;  // junk above the uint8_t count of mate 0: must be masked away
            dd[u] &= 0x01FF01FFu;
            const int n = (rnd() % 3 == 0 && at + 4 < (uint32_t)NH) ? 2 + (int)(rnd() % 3) : 1;
            if (n == 1) { uval[u] = (uint32_t)(rnd() % D) << 1; nml[u] = 1; }
            else {
                uval[u] = ((uint32_t)vv.size() << 1) | 1u;
                vv.push_back((uint32_t)n);
                nml[u] = (uint32_t)n;
                poff[u] = (uint16_t)at;
                for (int q = 0; q < n; ++q) { const uint32_t l = (uint32_t)(rnd() % D); vv.push_back(l); pool[at++] = l; }
            }
        }
        vv.push_back(0);
        DevTables T;
        memset(&T, 0, sizeof(T));
        T.vv = vv.data();
        // serial reference
        std::vector<uint32_t> lkey(LC, NAN32), lhit(LC, 0), dd2 = dd;
        for (int u = 0; u < nu; ++u) dd2[u] &= 0x00FF00FFu;
        std::vector<uint64_t> g(8, 0);
        HitMap hm{lkey.data(), lhit.data(), 0, g.data(), 1, false, (uint32_t)LC, 23u, (uint32_t)(LC * 3 / 4)};
        Asgn t0;
        uint64_t v0 = 0;
        vote(T, ord.data(), uval.data(), dd2.data(), nu, cth, hm, t0, v0, nml.data(), pool.data(), poff.data());
        // the wave
        std::vector<uint32_t> el(LC, 0), eh(LC, 0), ed(LC, 0);
        Asgn t1{NAN32, 0, 0};
        uint64_t v1 = 0;
        bool ran = false;
        run_grid(1, 64, 0, [&](EmuX& x) {
            Asgn tt{NAN32, 0, 0};
            uint64_t vvw = 0;
            const bool ok = vote_parallel<NH / 64, LC>(x, ord.data(), uval.data(), dd.data(), nml.data(), poff.data(), pool.data(), el.data(),
                                                       eh.data(), ed.data(), (uint32_t)nu, cth, tt, vvw);
            if (x.lane() == 17) { t1 = tt; v1 = vvw; ran = ok; }
        });
        uint32_t nev = 0;
        for (int u = 0; u < nu; ++u) nev += nml[u];
        const bool fits = nev <= (uint32_t)LC;  // otherwise it must decline (the kernel then runs the literal loop)
        if (ran != fits || (ran && (t0.idx != t1.idx || t0.fc != t1.fc || t0.rc != t1.rc || v0 != v1))) {
            if (getenv("EMU_DEBUG")) fprintf(stderr, "it=%llu nu=%d D=%d cth=%u ran=%d serial idx=%llu fc=%llu rc=%llu vv=%llu | wave idx=%llu fc=%llu rc=%llu vv=%llu\n", (unsigned long long)it, nu, D, cth, (int)ran, (unsigned long long)t0.idx, (unsigned long long)t0.fc, (unsigned long long)t0.rc, (unsigned long long)v0, (unsigned long long)t1.idx, (unsigned long long)t1.fc, (unsigned long long)t1.rc, (unsigned long long)v1);
            ++bad;
        }
    }
    return bad;
}

// assign_bits (the mask form the kernels use) against assign_scan (the literal
// restatement of AQ.cpp:1470-1555) on `iters` random state vectors.  Returns the
// number of mismatches.
uint64_t emu_selftest_assign(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint64_t it = 0; it < iters; ++it) {
        const int nk = 1 + (int)(rnd() % 256);
        uint8_t as[256];
        const int mode = (int)(rnd() % 6);
        // runs of equal states (realistic) or per-position noise
        int i = 0;
        while (i < nk) {
            const int st = (int)(rnd() % 3);
            int len = mode < 4 ? 1 + (int)(rnd() % (mode == 0 ? 3 : mode == 1 ? 20 : 90)) : 1;
            for (; len > 0 && i < nk; --len) as[i++] = (uint8_t)st;
        }
        dbtk_params_t P;
        memset(&P, 0, sizeof(P));
        P.max_nt = (rnd() % 8 == 0) ? (uint32_t)(rnd() % 4) : 2;
        P.nm_tr = (uint32_t)(rnd() % 80);
        uint32_t ntr = 0;
        Bits256 K{{0, 0, 0, 0}}, R{{0, 0, 0, 0}};
        for (int j = 0; j < nk; ++j) {
            if (as[j]) K.w[j >> 6] |= 1ull << (j & 63);
            if (as[j] == 2) { R.w[j >> 6] |= 1ull << (j & 63); ++ntr; }
        }
        ntr &= 0xFF;
        {   // the word-wise carry-propagation form the kernels use == the log-step fill-forward definition
            const Bits256 Tdef = transitions_from_masks(K, R);
            uint32_t carry = 0;
            for (int wi = 0; wi < 4; ++wi)
                if (transitions_word(K.w[wi], R.w[wi], carry) != Tdef.w[wi]) { ++bad; break; }
        }
        for (int rm = 0; rm < 2; ++rm) {
            MateState a{-1, -1, 0, 0, -1, -1, -1, 0, rm}, b = a;
            assign_scan(as, nk, ntr, P, a);
            assign_bits(K, R, nk, ntr, P, b);
            if (memcmp(&a, &b, sizeof(a)) != 0) ++bad;
        }
    }
    return bad;
}

// Same outputs as dbtk_align_batch + dbtk_ctx_counts (counts in OUT.trkmc.ar order).
int emu_align_ex(const dbtk_rpgg_t* g, void* tables, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
                 const uint8_t* qual, uint64_t npairs, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* counters,
                 dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec, uint32_t grid_k1, uint32_t grid_pair, BubEvent* ev,
                 uint64_t evcap, uint64_t* nev);

// threading = 2: results of the last emu_align_ex (what dbtk_ctx_walk_results returns), thread records when a buffer was lent
static std::vector<dbtk_walk_res_t> g_walk_res;
static std::vector<uint32_t> g_walk_t;
static dbtk_thread_rec_t* g_walk_trecs = nullptr;  // 2 per survivor index
void emu_set_walk_trecs(dbtk_thread_rec_t* buf) { g_walk_trecs = buf; }
static std::vector<uint8_t> g_aln;  // compact alignment records of the last emu_align_ex (-a / -ae)
static std::vector<uint8_t> g_txt;       // text records of the last emu_align_ex (params.aln | DBTK_ALN_TEXT): the arena in use
static std::vector<uint32_t> g_txt_idx;  //   ... and its per-pair index
static uint32_t g_aln_stride = 0, g_aln_cap = 0;
uint64_t emu_aln_records(uint8_t* buf, uint64_t bytes, uint32_t* stride, uint32_t* cap) {
    *stride = g_aln_stride; *cap = g_aln_cap;
    if (buf && bytes >= g_aln.size() && !g_aln.empty()) memcpy(buf, g_aln.data(), g_aln.size());
    return g_aln_stride ? g_aln.size() / g_aln_stride : 0;
}
// as dbtk_ctx_aln_text
uint64_t emu_aln_text(uint32_t* idx, uint64_t idx_cap, uint8_t* arena, uint64_t arena_cap) {
    for (uint64_t i = 0; idx && i < idx_cap; ++i) idx[i] = i < g_txt_idx.size() ? g_txt_idx[i] : NAN32;
    if (arena && arena_cap >= g_txt.size() && !g_txt.empty()) memcpy(arena, g_txt.data(), g_txt.size());
    return g_txt.size();
}
uint64_t emu_walk_results(dbtk_walk_res_t* res, uint32_t* tidx, uint64_t cap) {
    std::vector<size_t> ord(g_walk_res.size());
    for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return g_walk_res[a].pair < g_walk_res[b].pair; });
    for (size_t i = 0; i < ord.size() && i < cap; ++i) { res[i] = g_walk_res[ord[i]]; res[i].pad[0] = res[i].pad[1] = 0; if (tidx) tidx[i] = g_walk_t[ord[i]]; }
    return g_walk_res.size();
}


// which probe body the last calls of emu_align_ex dispatched: [0] general (body_probe), [1] lean (body_probe2); and the
// keys level 1 of the last emu_tables_create turned away (= entries of its overflow table)
void emu_path_stats(uint64_t* out) { for (uint32_t i = 0; i < DBTK_PATH_STATS; ++i) { out[i] = g_pstats[i]; g_pstats[i] = 0; } }
void emu_locus_stats(uint64_t* out) { for (int i = 0; i < 4; ++i) { out[i] = g_loc_pairs[i]; g_loc_pairs[i] = 0; } out[4] = g_loc_left; }
void emu_probe_stats(uint64_t* out) { out[0] = g_probe_runs[0]; out[1] = g_probe_runs[1]; out[2] = g_mz_turned; g_probe_runs[0] = g_probe_runs[1] = 0; }
// pair-mode walks since the last call: runs of the lean first kernel, pairs it passed on to the second
void emu_walk_stats(uint64_t* out) { out[0] = g_walk_fast_runs; out[1] = g_walk_slow_pairs; g_walk_fast_runs = g_walk_slow_pairs = 0; }
void emu_walk_locus_stats(uint64_t* out) { out[0] = g_wfl_pairs[0]; out[1] = g_wfl_pairs[1]; g_wfl_pairs[0] = g_wfl_pairs[1] = 0; }
int emu_align(const dbtk_rpgg_t* g, void* tables, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
              uint64_t npairs, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* counters,
              dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec, uint32_t grid_k1, uint32_t grid_pair) {
    uint64_t nev = 0;
    return emu_align_ex(g, tables, p, seq, off, nullptr, npairs, counts, kmc, nmapread, counters, recs, rec_cap, nrec, grid_k1, grid_pair,
                        nullptr, 0, &nev);
}

int emu_align_ex(const dbtk_rpgg_t* g, void* tables, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
                 const uint8_t* qual, uint64_t npairs, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* counters,
                 dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec, uint32_t grid_k1, uint32_t grid_pair, BubEvent* ev,
                 uint64_t evcap, uint64_t* nev) {
    EmuTables* e = (EmuTables*)tables;
    const uint64_t nloci = g->nloci, ntr = g->out_kmer.size();
    std::vector<uint64_t> accum(ntr + 2 * nloci + DBTK_C_COUNT, 0);
    std::vector<uint32_t> surv(npairs + 1), small(4, 0), epoch(grid_pair, 0);
    std::vector<uint64_t> vote((size_t)grid_pair * (nloci + 1), 0);
    // device-like sequence buffer: 16-byte aligned copy
    const uint64_t nbytes = off[2 * npairs];
    std::vector<uint64_t> seqbuf(nbytes / 8 + 8, 0);
    memcpy(seqbuf.data(), seq, nbytes);
    BatchArgs a;
    memset(&a, 0, sizeof(a));
    a.T = e->T; a.P = *p;
    a.k1_xcd = 1;  // (the encode stage's tiles in one contiguous range per XCD, as on the device: workgroup b stands for XCD b % 8)
    a.seq = (const uint8_t*)seqbuf.data(); a.off = off; a.seq_len = nbytes; a.npairs = npairs;
    a.surv = surv.data(); a.nsurv = &small[0]; a.nrec = &small[2]; a.errflag = &small[3];
    a.counts = accum.data(); a.kmc = a.counts + ntr; a.nmapread = a.kmc + nloci; a.counters = a.nmapread + nloci;
    a.recs = recs; a.rec_cap = (uint32_t)rec_cap;
    a.pstats = g_pstats;
    // (two rows for any number of workgroups: the pool is taken from and given back per pair)
    std::vector<uint64_t> vbusy(2, 0);
    a.vote_scratch = vote.data(); a.vote_epoch = epoch.data(); a.vote_busy = vbusy.data(); a.vote_rows = grid_pair < 2 ? grid_pair : 2;
    std::vector<uint32_t> walk(2 * npairs + 2, NAN32);
    std::fill(walk.begin() + npairs, walk.end(), 0u);  // (walk_ret is zeroed per batch: no pair is marked WALK_PENDING)
    a.walk_dst = walk.data();
    uint32_t maxlen = 1;
    for (uint64_t r = 0; r < 2 * npairs; ++r) maxlen = std::max<uint32_t>(maxlen, (uint32_t)(off[r + 1] - off[r]));
    const uint32_t nkmax = maxlen >= g->ksize ? maxlen - g->ksize + 1 : 1;
    a.nkp = 64 * ((nkmax + 63) / 64);
    a.pair_base = 0;
    // survivor chunks: small hit buffers force several K2 -> K3 iterations, as on the device
    const uint32_t tcap = npairs > 7 ? (uint32_t)(npairs / 3 + 1) : (uint32_t)(npairs ? npairs : 1);
    std::vector<uint64_t> hitva((size_t)tcap * 2 * a.nkp + 1, 0);  // aux words of every row, then val words
    // (poisoned: a row the probe stage did not write — a pair the fused locus-resident body resolved itself — must never be read as one
    // that was: an offset from such a row is a wild address, as on the device, where the buffers hold whatever was there)
    std::vector<uint32_t> hitnk((size_t)tcap * 2 + 1, 0xA5A5A5A5u), gen(tcap + 1, 0);
    std::vector<uint64_t> hitoff((size_t)tcap * 4 + 1, 0x7FFFDEAD00000000ull);
    a.hitaux = reinterpret_cast<uint32_t*>(hitva.data()); a.hitval = a.hitaux + (size_t)tcap * 2 * a.nkp; a.hitnk = hitnk.data(); a.hitoff = hitoff.data(); a.hithdr = hitoff.data() + (size_t)tcap * 2;
    a.tcap = tcap;
    const bool usual = e->T.consistent && !p->trace && !p->bait && !p->bubbles;  // as the device launcher decides
    const int fuse_bits = getenv("EMU_FUSE") ? atoi(getenv("EMU_FUSE")) : (getenv("EMU_NO_FUSE") ? 0 : 3);  // (as DBTK_FUSE: bit 0 locus-resident, bit 1 lean)
    const bool fuse = usual && !recs && (fuse_bits & 1);                         // ... and the fused form of the locus-resident probe body
    bool fuse_lean = usual && !recs && (fuse_bits & 2);                          // ... and of the lean one
    std::vector<uint64_t> edgebuf, qmaskbuf, qualbuf;
    uint32_t nevents = 0;
    if (p->bubbles) {
        edgebuf.assign((size_t)tcap * 2 * a.nkp + 1, 0);
        a.edgebuf = edgebuf.data(); a.events = ev; a.nevents = &nevents; a.events_cap = (uint32_t)evcap;
    }
    if (p->bait && qual) {
        qualbuf.assign(nbytes / 8 + 8, 0);
        memcpy(qualbuf.data(), qual, nbytes);
        qmaskbuf.assign((size_t)tcap * 2 * 4 + 1, 0);
        a.qual = (const uint8_t*)qualbuf.data(); a.qmaskbuf = qmaskbuf.data();
    }
    std::vector<uint32_t> sorted(npairs + 1), skey(npairs + 1, 0xDEADBEEFu), shist(nloci + 2 + SCAN_BLOCKS, 0);
    // (the encode stage writes every survivor's sort key itself in its one-sample-per-turn form, as launch_batch has it do)
    const bool k1_keys = p->n_filter && p->nm_filter == 1 && e->T.flt && !getenv("EMU_NO_K1_KEYS");
    a.skey = k1_keys ? skey.data() : nullptr;
    // (EMU_K1_LAZY=1: the encode stage's form for a batch that hits, body_encode_subfilter<true>)
    if (getenv("EMU_K1_LAZY") && atoi(getenv("EMU_K1_LAZY"))) run_grid(grid_k1, K1_NT, sizeof(K1Smem), [&](EmuX& x) { body_encode_subfilter<true>(x, a); });
    else run_grid(grid_k1, K1_NT, sizeof(K1Smem), [&](EmuX& x) { body_encode_subfilter(x, a); });
    {   // the survivor list in locus order, as launch_batch does
        SurvSortArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.T = e->T; sa.P = *p; sa.seq = a.seq; sa.off = off; sa.surv = surv.data(); sa.nsurv = &small[0];
        sa.sorted = sorted.data(); sa.key = skey.data(); sa.hist = shist.data(); sa.flag = &small[1]; sa.sort_min = SORT_MIN_PER_LOCUS;
        std::vector<uint32_t> srank(npairs + 1, 0), sstarts(nloci + 2, 0);
        sa.rank = srank.data(); sa.starts = sstarts.data();
        if (k1_keys) {  // ... and they must be the keys body_surv_key looks up (sort_min 0: the keys of a small batch too)
            std::vector<uint32_t> kref(npairs + 1), href(nloci + 2 + SCAN_BLOCKS, 0);
            uint32_t fref = 0;
            SurvSortArgs sr = sa;
            std::vector<uint32_t> rref(npairs + 1, 0);
            sr.key = kref.data(); sr.hist = href.data(); sr.rank = rref.data(); sr.flag = &fref; sr.sort_min = 0; sr.have_keys = 0;
            run_grid(3, 64, 0, [&](EmuX& x) { body_surv_key(x, sr); });
            for (uint32_t t = 0; t < small[0]; ++t) if (kref[t] != skey[t]) return -78;
            sa.have_keys = 1;
        }
        run_grid(3, 64, 0, [&](EmuX& x) { body_surv_key(x, sa); });
        run_grid(SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_surv_scan(x, sa, 0); });
        run_grid(SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_surv_scan(x, sa, 1); });
        run_grid(3, 64, 0, [&](EmuX& x) { body_surv_scatter(x, sa); });
        std::vector<uint32_t> a1(surv.begin(), surv.begin() + small[0]), a2(sorted.begin(), sorted.begin() + small[0]);
        std::sort(a1.begin(), a1.end()); std::sort(a2.begin(), a2.end());
        if (a1 != a2) return -77;  // the sorted list must be a permutation of the encode stage's
        a.surv = sorted.data();
    }
    for (uint32_t t0 = 0; t0 < (uint32_t)(npairs ? npairs : 1); t0 += tcap) {
        a.t0 = t0;
        uint32_t ngen = 0;
        a.gen_list = usual ? gen.data() : nullptr;
        a.ngen = usual ? &ngen : nullptr;
        // same dispatch as the device launcher (launch_batch): the lean probe body where its conditions hold
        const uint32_t wn = a.T.mz ? g->ksize - a.T.mz_m + 1 : 0;
        const int npl = !a.T.mz || p->bubbles || (p->bait && qual) ? 0 : (maxlen <= 32 * 3 + a.T.mz_m - 1 ? 3 : maxlen <= 32 * 5 + a.T.mz_m - 1 ? 5 : 0);
        g_probe_runs[npl ? 1 : 0] += 1;
        const uint32_t grid_p2 = (grid_pair & 1) ? 8 : grid_pair + 2;  // (a multiple of 8: the lean body's per-XCD split of the list)
        // the locus-resident kernel first, as launch_batch does (workgroups of 4 waves here; the small class takes the
        // images of up to 512 buckets)
        std::vector<uint4> items[3];
        std::vector<uint32_t> rest(tcap + 64, 0);
        uint32_t nit[4] = {0, 0, 0, 0};
        a.sel = nullptr; a.nsel = nullptr;
        if (npl && a.T.ldir && !getenv("EMU_NO_LOCUS")) {  // (EMU_NO_LOCUS: every pair through the lean body, as in a batch with few survivors per locus)
            // (three classes of workgroup by image size as on the device: up to 512, 1024, 2048 buckets; 4 waves each here)
            constexpr int EMU_IMGB_XS = LOC_HDR + (32 << 9) + (1 << 9), EMU_IMGB_S = LOC_HDR + (32 << 10) + (1 << 10), EMU_IMGB_L = LOC_HDR + (32 << LOC_LG_MAX) + (1 << LOC_LG_MAX);
            const uint32_t item_cap = tcap / LOC_CH + (uint32_t)nloci + 2;
            for (int q = 0; q < 3; ++q) items[q].assign(item_cap, uint4{0, 0, 0, 0});
            LocItemArgs ia;
            memset(&ia, 0, sizeof(ia));
            ia.hist = shist.data(); ia.nsurv = &small[0]; ia.flag = &small[1]; ia.dir = a.T.ldir; ia.nloci = (uint32_t)nloci; ia.t0 = t0; ia.tcap = tcap;
            ia.cap_bytes[0] = EMU_IMGB_XS; ia.cap_bytes[1] = EMU_IMGB_S; ia.cap_bytes[2] = EMU_IMGB_L;
            for (int q = 0; q < 3; ++q) ia.items[q] = items[q].data();
            ia.nitems = nit; ia.item_cap = item_cap; ia.rest = rest.data();
            run_grid(2, 64, 0, [&](EmuX& x) { body_loc_items(x, ia); });
            run_grid(2, 128, 0, [&](EmuX& x) { body_loc_rest(x, ia); });
            // the workgroups' ranges of the item lists (grids of 2, 3 and 2 workgroups below)
            std::vector<uint32_t> starts[3];
            LocSplitArgs sp;
            memset(&sp, 0, sizeof(sp));
            const uint32_t nblk_[3] = {2, 3, 2};
            for (int q = 0; q < 3; ++q) { starts[q].assign(nblk_[q] + 1, 0xDEADBEEFu); sp.items[q] = items[q].data(); sp.starts[q] = starts[q].data(); sp.nblk[q] = nblk_[q]; sp.wfix[q] = 4u << q; }
            sp.nitems = nit; sp.item_cap = item_cap;
            run_grid(3, 256, sizeof(LocSplitSmem), [&](EmuX& x) { body_loc_split(x, sp); });
            for (int q = 0; q < 3; ++q) {  // the ranges tile the list
                if (starts[q][0] != 0 || starts[q][nblk_[q]] != std::min(nit[q], item_cap)) { fprintf(stderr, "emu: item ranges of class %d do not cover its list\n", q); abort(); }
                for (uint32_t j = 0; j < nblk_[q]; ++j) if (starts[q][j] > starts[q][j + 1]) { fprintf(stderr, "emu: item ranges of class %d out of order\n", q); abort(); }
            }
            LocRunArgs r0{a.T.ldir, a.T.limg, items[0].data(), &nit[0], rest.data(), &nit[3], starts[0].data()}, r1{a.T.ldir, a.T.limg, items[1].data(), &nit[1], rest.data(), &nit[3], starts[1].data()},
                r2{a.T.ldir, a.T.limg, items[2].data(), &nit[2], rest.data(), &nit[3], starts[2].data()};
            if (fuse && npl == 3) {
                run_grid(2, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_XS, true>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_XS, true>(x, a, r0); });
                run_grid(3, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_S, true>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_S, true>(x, a, r1); });
                run_grid(2, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_L, true>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_L, true>(x, a, r2); });
            } else if (fuse) {
                run_grid(2, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_XS, true>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_XS, true>(x, a, r0); });
                run_grid(3, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_S, true>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_S, true>(x, a, r1); });
                run_grid(2, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_L, true>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_L, true>(x, a, r2); });
            } else if (npl == 3) {
                run_grid(2, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_XS, false>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_XS, false>(x, a, r0); });
                run_grid(3, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_S, false>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_S, false>(x, a, r1); });
                run_grid(2, 4 * 64, sizeof(LocSmemT<3, 4, EMU_IMGB_L, false>), [&](EmuX& x) { body_probe_locus<3, 4, EMU_IMGB_L, false>(x, a, r2); });
            } else {
                run_grid(2, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_XS, false>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_XS, false>(x, a, r0); });
                run_grid(3, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_S, false>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_S, false>(x, a, r1); });
                run_grid(2, 4 * 64, sizeof(LocSmemT<5, 4, EMU_IMGB_L, false>), [&](EmuX& x) { body_probe_locus<5, 4, EMU_IMGB_L, false>(x, a, r2); });
            }
            for (int c = 0; c < 3; ++c) for (uint32_t q = 0; q < nit[c]; ++q) g_loc_pairs[c] += items[c][q].z - items[c][q].y;
            g_loc_pairs[3] += nit[3];
            a.sel = rest.data(); a.nsel = &nit[3];
        }
        if (a.sel && !fuse) fuse_lean = false;  // (pairs an unfused locus-resident body took wait in their rows for the usual-pair body)
        if (npl == 3 && wn == 7) run_grid(grid_p2, 64, sizeof(Probe2SmemT<3>), [&](EmuX& x) { if (a.sel && fuse_lean) body_probe2<3, 7, true, true>(x, a); else if (a.sel) body_probe2<3, 7, true, false>(x, a); else if (fuse_lean) body_probe2<3, 7, false, true>(x, a); else body_probe2<3, 7, false, false>(x, a); });
        else if (npl == 3 && wn == 11) run_grid(grid_p2, 64, sizeof(Probe2SmemT<3>), [&](EmuX& x) { if (a.sel && fuse_lean) body_probe2<3, 11, true, true>(x, a); else if (a.sel) body_probe2<3, 11, true, false>(x, a); else if (fuse_lean) body_probe2<3, 11, false, true>(x, a); else body_probe2<3, 11, false, false>(x, a); });
        else if (npl == 5 && wn == 7) run_grid(grid_p2, 64, sizeof(Probe2SmemT<5>), [&](EmuX& x) { if (a.sel && fuse_lean) body_probe2<5, 7, true, true>(x, a); else if (a.sel) body_probe2<5, 7, true, false>(x, a); else if (fuse_lean) body_probe2<5, 7, false, true>(x, a); else body_probe2<5, 7, false, false>(x, a); });
        else if (npl == 5 && wn == 11) run_grid(grid_p2, 64, sizeof(Probe2SmemT<5>), [&](EmuX& x) { if (a.sel && fuse_lean) body_probe2<5, 11, true, true>(x, a); else if (a.sel) body_probe2<5, 11, true, false>(x, a); else if (fuse_lean) body_probe2<5, 11, false, true>(x, a); else body_probe2<5, 11, false, false>(x, a); });
        switch (a.nkp / 64) {
            case 1: case 2:
                if (!npl) run_grid(grid_pair + 2, 64, sizeof(ProbeSmem), [&](EmuX& x) { body_probe<2>(x, a); });
                if (usual && !(fuse_lean && npl)) run_grid(grid_pair + 1, 64, sizeof(UsualSmem), [&](EmuX& x) { if (a.recs) body_pair_usual<2, true, false>(x, a); else if (fuse && a.sel) body_pair_usual<2, false, true>(x, a); else body_pair_usual<2, false, false>(x, a); });
                run_grid(grid_pair, 64, sizeof(PairSmemT<2>), [&](EmuX& x) { if (a.recs) body_pair<2, true>(x, a); else body_pair<2, false>(x, a); });
                break;
            case 3:
                if (!npl) run_grid(grid_pair + 2, 64, sizeof(ProbeSmem), [&](EmuX& x) { body_probe<3>(x, a); });
                if (usual && !(fuse_lean && npl)) run_grid(grid_pair + 1, 64, sizeof(UsualSmem), [&](EmuX& x) { if (a.recs) body_pair_usual<3, true, false>(x, a); else if (fuse && a.sel) body_pair_usual<3, false, true>(x, a); else body_pair_usual<3, false, false>(x, a); });
                run_grid(grid_pair, 64, sizeof(PairSmemT<3>), [&](EmuX& x) { if (a.recs) body_pair<3, true>(x, a); else body_pair<3, false>(x, a); });
                break;
            default:
                if (!npl) run_grid(grid_pair + 2, 64, sizeof(ProbeSmem), [&](EmuX& x) { body_probe<4>(x, a); });
                if (usual && !(fuse_lean && npl)) run_grid(grid_pair + 1, 64, sizeof(UsualSmem), [&](EmuX& x) { if (a.recs) body_pair_usual<4, true, false>(x, a); else if (fuse && a.sel) body_pair_usual<4, false, true>(x, a); else body_pair_usual<4, false, false>(x, a); });
                run_grid(grid_pair, 64, sizeof(PairSmemT<4>), [&](EmuX& x) { if (a.recs) body_pair<4, true>(x, a); else body_pair<4, false>(x, a); });
        }
    }
    if (p->threading == DBTK_THREADING_V13) {
        WalkArgs w;
        memset(&w, 0, sizeof(w));
        w.T = e->T; w.P = *p; w.P.aln &= 3u; w.seq = a.seq; w.off = off;
        w.surv = sorted.data(); w.nsurv = &small[0];
        w.walk_dst = walk.data(); w.walk_ret = walk.data() + npairs;
        w.counts = a.counts; w.counters = a.counters;
        w.trecs = g_walk_trecs; w.errflag = &small[3]; w.pstats = g_pstats;
        std::vector<uint8_t> alnraw;
        uint32_t naln = 0;
        g_aln.clear();
        const bool txtmode = (p->aln & 3u) && (p->aln & DBTK_ALN_TEXT);
        uint32_t ntxt = 0;
        g_txt.clear(); g_txt_idx.clear();
        if (txtmode) {
            g_aln_cap = std::min<uint32_t>(DBTK_THREAD_CAP, (maxlen + maxlen / 4 + 8 + 7) & ~7u);
            g_txt.assign(npairs * (size_t)(8 + 8 * g_aln_cap + 8) + (size_t)TXT_CHUNK * (2 * grid_pair + 2), 0);
            g_txt_idx.assign(npairs + 1, NAN32);
            w.txt = g_txt.data(); w.txt_idx = g_txt_idx.data(); w.ntxt = &ntxt; w.txt_cap = (uint32_t)g_txt.size(); w.aln_cap = g_aln_cap;
        }
        if ((p->aln & 3u) && !txtmode) {
            g_aln_cap = std::min<uint32_t>(DBTK_THREAD_CAP, (maxlen + maxlen / 4 + 8 + 7) & ~7u);
            g_aln_stride = (uint32_t)sizeof(dbtk_aln_hdr_t) + 4 * g_aln_cap;
            const uint64_t amax = npairs + (uint64_t)ALN_CHUNK * grid_pair;
            alnraw.assign(amax * g_aln_stride, 0);
            w.aln = alnraw.data(); w.aln_stride = g_aln_stride; w.aln_cap = g_aln_cap; w.aln_max = (uint32_t)amax; w.naln = &naln;
        }
        // the two-kernel form where nothing needs every mate's alignment, as launch_batch decides
        // (the passed-on list: the fast kernels' waves reserve WF_R places at a time; rows of graph info for HALF of its entries only,
        // so that both of the other kernel's ways of getting a pair's graph nodes run in every test)
        const size_t slow_cap = npairs + (size_t)WF_R * 64 * 16 + 8;
        std::vector<uint32_t> slow(slow_cap, 0xABABABABu);
        const size_t info_rows = std::min<size_t>(slow_cap, std::max<size_t>(npairs / 2, 4));
        std::vector<uint32_t> slow_info(info_rows * 2 * 160, 0xCDCDCDCDu);
        uint32_t nslow = 0;
        const uint32_t kk = g->ksize;
        const int wnpl = (((p->aln & 3u) && !txtmode) || g_walk_trecs) ? 0 : walkfast_npl(maxlen, kk, w.T.grmz != nullptr);
        std::vector<uint4> witems[3];
        std::vector<uint32_t> wrest(npairs + 64, 0);
        uint32_t wnit[4] = {0, 0, 0, 0};
        if (wnpl) {
            w.slow_list = slow.data(); w.nslow = &nslow;
            w.slow_info = slow_info.data(); w.info_cap = (uint32_t)info_rows; w.info_stride = 32u * (uint32_t)wnpl;
            if (w.T.gldir) {  // the locus-resident form first, as launch_batch does (4 waves per workgroup here)
                constexpr int EMU_IMGB_XS = LOC_HDR + (32 << 9) + (1 << 9), EMU_IMGB_S = LOC_HDR + (32 << 10) + (1 << 10), EMU_IMGB_L = LOC_HDR + (32 << LOC_LG_MAX) + (1 << LOC_LG_MAX);
                const uint32_t item_cap = (uint32_t)(npairs / LOC_CH + nloci + 2);
                for (int q = 0; q < 3; ++q) witems[q].assign(item_cap, uint4{0, 0, 0, 0});
                LocItemArgs ia;
                memset(&ia, 0, sizeof(ia));
                ia.hist = shist.data(); ia.nsurv = &small[0]; ia.flag = &small[1]; ia.dir = w.T.gldir; ia.nloci = (uint32_t)nloci; ia.t0 = 0; ia.tcap = (uint32_t)(npairs ? npairs : 1);
                ia.cap_bytes[0] = EMU_IMGB_XS; ia.cap_bytes[1] = EMU_IMGB_S; ia.cap_bytes[2] = EMU_IMGB_L;
                for (int q = 0; q < 3; ++q) ia.items[q] = witems[q].data();
                ia.nitems = wnit; ia.item_cap = item_cap; ia.rest = wrest.data();
                run_grid(2, 64, 0, [&](EmuX& x) { body_loc_items(x, ia); });
                run_grid(2, 128, 0, [&](EmuX& x) { body_loc_rest(x, ia); });
                std::vector<uint32_t> wstarts[3];
                LocSplitArgs sp;
                memset(&sp, 0, sizeof(sp));
                const uint32_t nblk_[3] = {2, 3, 2};
                for (int q = 0; q < 3; ++q) { wstarts[q].assign(nblk_[q] + 1, 0xDEADBEEFu); sp.items[q] = witems[q].data(); sp.starts[q] = wstarts[q].data(); sp.nblk[q] = nblk_[q]; sp.wfix[q] = 4u << q; }
                sp.nitems = wnit; sp.item_cap = item_cap;
                run_grid(3, 256, sizeof(LocSplitSmem), [&](EmuX& x) { body_loc_split(x, sp); });
                for (int q = 0; q < 3; ++q)
                    if (wstarts[q][0] != 0 || wstarts[q][nblk_[q]] != std::min(wnit[q], item_cap)) { fprintf(stderr, "emu: walk item ranges of class %d do not cover its list\n", q); abort(); }
                LocRunArgs r0{w.T.gldir, w.T.glimg, witems[0].data(), &wnit[0], nullptr, nullptr, wstarts[0].data()}, r1{w.T.gldir, w.T.glimg, witems[1].data(), &wnit[1], nullptr, nullptr, wstarts[1].data()},
                    r2{w.T.gldir, w.T.glimg, witems[2].data(), &wnit[2], nullptr, nullptr, wstarts[2].data()};
                const bool pend_locus = getenv("DBTK_WALK_LOCUS_EC") && atoi(getenv("DBTK_WALK_LOCUS_EC")) != 0;  // (as launch_batch: opt-in)
                w.pend_locus = pend_locus ? 1u : 0u;
                if (wnpl == 3) {
                    run_grid(2, 4 * 64, sizeof(WalkFastLocSmemT<3, 4, EMU_IMGB_XS>), [&](EmuX& x) { body_walk_fast_locus<3, 4, EMU_IMGB_XS>(x, w, r0); });
                    run_grid(3, 4 * 64, sizeof(WalkFastLocSmemT<3, 4, EMU_IMGB_S>), [&](EmuX& x) { body_walk_fast_locus<3, 4, EMU_IMGB_S>(x, w, r1); });
                    run_grid(2, 4 * 64, sizeof(WalkFastLocSmemT<3, 4, EMU_IMGB_L>), [&](EmuX& x) { body_walk_fast_locus<3, 4, EMU_IMGB_L>(x, w, r2); });
                } else {
                    run_grid(2, 4 * 64, sizeof(WalkFastLocSmemT<5, 4, EMU_IMGB_XS>), [&](EmuX& x) { body_walk_fast_locus<5, 4, EMU_IMGB_XS>(x, w, r0); });
                    run_grid(3, 4 * 64, sizeof(WalkFastLocSmemT<5, 4, EMU_IMGB_S>), [&](EmuX& x) { body_walk_fast_locus<5, 4, EMU_IMGB_S>(x, w, r1); });
                    run_grid(2, 4 * 64, sizeof(WalkFastLocSmemT<5, 4, EMU_IMGB_L>), [&](EmuX& x) { body_walk_fast_locus<5, 4, EMU_IMGB_L>(x, w, r2); });
                }
                if (pend_locus) {  // the error-correcting walk over the same items, as launch_batch does (two waves per workgroup)
                    run_grid(2, 2 * 64, sizeof(WalkPairsLocSmemT<2, EMU_IMGB_XS>), [&](EmuX& x) { body_walk_pairs_locus<2, EMU_IMGB_XS>(x, w, r0); });
                    run_grid(3, 2 * 64, sizeof(WalkPairsLocSmemT<2, EMU_IMGB_S>), [&](EmuX& x) { body_walk_pairs_locus<2, EMU_IMGB_S>(x, w, r1); });
                    run_grid(2, 2 * 64, sizeof(WalkPairsLocSmemT<2, EMU_IMGB_L>), [&](EmuX& x) { body_walk_pairs_locus<2, EMU_IMGB_L>(x, w, r2); });
                    for (uint32_t t = 0; t < small[0]; ++t) if (walk[npairs + t] == WALK_PENDING) { fprintf(stderr, "emu: a pair left to body_walk_pairs_locus was not walked\n"); abort(); }
                }
                w.pend_locus = 0;
                for (int c = 0; c < 3; ++c) for (uint32_t q = 0; q < wnit[c]; ++q) g_wfl_pairs[0] += witems[c][q].z - witems[c][q].y;
                g_wfl_pairs[1] += wnit[3];
                w.sel = wrest.data(); w.nsel = &wnit[3];
            }
            const bool w11 = w.T.grmz && kk - mz_m_for_k(kk) + 1 == 11;
            if (wnpl == 3) {
                if (w11) run_grid(grid_pair + 1, 64, sizeof(WalkFastSmemT<3>), [&](EmuX& x) { body_walk_fast<3, 11>(x, w); });
                else run_grid(grid_pair + 1, 64, sizeof(WalkFastSmemT<3>), [&](EmuX& x) { body_walk_fast<3, 7>(x, w); });
            } else {
                if (w11) run_grid(grid_pair + 1, 64, sizeof(WalkFastSmemT<5>), [&](EmuX& x) { body_walk_fast<5, 11>(x, w); });
                else run_grid(grid_pair + 1, 64, sizeof(WalkFastSmemT<5>), [&](EmuX& x) { body_walk_fast<5, 7>(x, w); });
            }
            if (nslow > slow_cap) { fprintf(stderr, "emu: the passed-on list outgrew its reservation bound\n"); abort(); }
            uint32_t nreal = 0;
            for (uint32_t q = 0; q < nslow; ++q) {
                if (slow[q] == 0xABABABABu) { fprintf(stderr, "emu: a reserved place of the passed-on list was left unwritten\n"); abort(); }
                nreal += slow[q] != WALK_NO_ENTRY;
            }
            g_walk_fast_runs += 1; g_walk_slow_pairs += nreal;
        }
        run_grid(grid_pair, 64, sizeof(WalkPairSmem), [&](EmuX& x) { body_walk_pairs(x, w); });
        if (txtmode) g_txt.resize(ntxt <= g_txt.size() ? ntxt : g_txt.size());
        if ((p->aln & 3u) && !txtmode) {  // as dbtk_ctx_aln_records: drop the invalid slots, pair order
            std::vector<std::pair<uint32_t, uint32_t>> order;
            for (uint32_t i = 0; i < naln; ++i) {
                const dbtk_aln_hdr_t* h = reinterpret_cast<const dbtk_aln_hdr_t*>(alnraw.data() + (size_t)i * g_aln_stride);
                if (h->pair != NAN32) order.emplace_back(h->pair, i);
            }
            std::sort(order.begin(), order.end());
            for (auto& o : order) g_aln.insert(g_aln.end(), alnraw.begin() + (size_t)o.second * g_aln_stride, alnraw.begin() + (size_t)(o.second + 1) * g_aln_stride);
        }
        g_walk_res.clear();
        for (uint32_t t = 0; t < small[0]; ++t)
            if (walk[t] != NAN32) g_walk_res.push_back({sorted[t], walk[t], (int8_t)(walk[npairs + t] & 0xFF), (int8_t)((walk[npairs + t] >> 8) & 0xFF), {(uint8_t)(t & 0xFF), (uint8_t)0}});
        g_walk_t.clear();
        for (uint32_t t = 0; t < small[0]; ++t) if (walk[t] != NAN32) g_walk_t.push_back(t);
    }
    if (small[3]) return (int)small[3];
    memcpy(counts, accum.data(), ntr * 8);
    memcpy(kmc, accum.data() + ntr, nloci * 8);
    for (uint64_t l = 0; l < nloci; ++l) nmapread[l] = (uint32_t)accum[ntr + nloci + l];
    memcpy(counters, accum.data() + ntr + 2 * nloci, DBTK_C_COUNT * 8);
    if (nrec) *nrec = p->trace ? npairs : small[2];
    *nev = nevents;
    return 0;
}

// The reader on the device (dbtk_ingest.h) on the emulated lanes: the input cut into chunks, every block through the
// kernel bodies in the order dbtk_ingest_submit launches them, the carry-over handed from block to block.  Stops after the
// first flagged block, as the library does.  Out: per block its header; of the kept pairs, concatenated over the blocks,
// the reads back to back (`flat`), their lengths, the qualities the gather wrote ('!'-padded) and the pruned titles
// ('\n'-separated).  Returns the number of blocks run (< 0: an output buffer was too small).
int emu_ingest(const uint8_t* data, uint64_t n, uint32_t fastq, uint32_t min_read, uint32_t chunk, uint32_t head, uint32_t line_cap, uint32_t grid,
               IngestHdr* hdrs, uint32_t max_blocks, uint8_t* flat, uint8_t* qual, uint64_t flat_cap, uint32_t* lens, uint64_t lens_cap,
               uint8_t* titles, uint64_t titles_cap, uint64_t* totals) {
    const uint32_t L = fastq ? 4 : 2, pair_cap = line_cap / (2 * L) + 1;
    const uint64_t raw_bytes = ((uint64_t)head + chunk + 1 + 63) & ~63ull;
    std::vector<uint64_t> raw_a(raw_bytes / 8 + 2, 0x4141414141414141ull), raw_b(raw_bytes / 8 + 2, 0x4141414141414141ull);  // (8-byte aligned: the bodies load 8 bytes at a time)
    std::vector<uint32_t> tile(raw_bytes / ING_TILE + 2 + ING_SCAN_BLOCKS), nlpos(line_cap), pk(pair_cap + 2 * ING_SCAN_BLOCKS), kept(pair_cap);
    std::vector<uint64_t> off(2 * (uint64_t)pair_cap + 1);
    std::vector<uint8_t> bflat(raw_bytes + 64), bqual(raw_bytes + 64);
    std::vector<dbtk_ingest_span_t> spans(pair_cap);
    uint32_t base_w[2] = {head, head};
    uint64_t pos = 0, nflat = 0, nlens = 0, ntit = 0;
    uint32_t blk = 0;
    int last_byte = '\n';
    for (;; ++blk) {
        if (blk >= max_blocks) return -1;
        uint8_t* cur = (uint8_t*)((blk & 1) ? raw_b.data() : raw_a.data());
        uint8_t* nxt = (uint8_t*)((blk & 1) ? raw_a.data() : raw_b.data());
        uint64_t m = std::min<uint64_t>(chunk, n - pos);
        const bool last = pos + m >= n;
        memcpy(cur + head, data + pos, m);
        if (m) last_byte = cur[head + m - 1];
        pos += m;
        if (last && last_byte != '\n') { cur[head + m++] = '\n'; last_byte = '\n'; }
        IngestHdr& h = hdrs[blk];
        memset(&h, 0, sizeof(h));
        IngestArgs a;
        memset(&a, 0, sizeof(a));
        a.raw = cur; a.base_in = &base_w[blk & 1]; a.end = head + (uint32_t)m; a.L = L; a.min_read = min_read; a.last = last;
        a.tile_cnt = tile.data(); a.nlpos = nlpos.data(); a.line_cap = line_cap; a.pk = pk.data(); a.kept = kept.data(); a.off = off.data();
        a.flat = bflat.data(); a.qual = fastq ? bqual.data() : nullptr; a.spans = spans.data(); a.hdr = &h;
        a.next_raw = nxt; a.base_out = &base_w[(blk + 1) & 1]; a.head = head;
        const uint32_t g = grid ? grid : 3;
        run_grid(g, 64, 0, [&](EmuX& x) { body_ing_count(x, a); });
        run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_ing_scan(x, a, 0); });
        run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_ing_scan(x, a, 1); });
        run_grid(g, 64, 0, [&](EmuX& x) { body_ing_lines(x, a); });
        run_grid(g, 64, 0, [&](EmuX& x) { body_ing_pairs(x, a); });
        run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_ing_place(x, a, 0); });
        run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_ing_place(x, a, 1); });
        run_grid(g, 64, 0, [&](EmuX& x) { body_ing_gather(x, a); });
        run_grid(1, 64, 0, [&](EmuX& x) { body_ing_carry(x, a); });
        if (!(h.flags & (ING_F_DIRTY | ING_F_LINES_OVF))) {
            if (nflat + h.flat_bytes > flat_cap || nlens + 2ull * h.nkept > lens_cap) return -2;
            memcpy(flat + nflat, bflat.data(), h.flat_bytes);
            if (fastq && qual) memcpy(qual + nflat, bqual.data(), h.flat_bytes);
            nflat += h.flat_bytes;
            for (uint64_t r = 0; r < 2ull * h.nkept; ++r) lens[nlens++] = (uint32_t)(off[r + 1] - off[r]);
            for (uint32_t q = 0; q < h.nkept; ++q) {
                const dbtk_ingest_span_t& S = spans[q];
                if (ntit + S.title_len + 1 > titles_cap) return -3;
                memcpy(titles + ntit, cur + S.title, S.title_len); ntit += S.title_len; titles[ntit++] = '\n';
                // the spans point at what the gather copied
                if (S.seq_len[0] != lens[nlens - 2ull * h.nkept + 2 * q] || memcmp(cur + S.seq[0], bflat.data() + off[2 * q], S.seq_len[0])) return -4;
                if (S.seq_len[1] != lens[nlens - 2ull * h.nkept + 2 * q + 1] || memcmp(cur + S.seq[1], bflat.data() + off[2 * q + 1], S.seq_len[1])) return -4;
            }
        }
        if (h.flags || last) { ++blk; break; }
    }
    totals[0] = nflat; totals[1] = nlens; totals[2] = ntit; totals[3] = pos;
    return (int)blk;
}
// body_gz_member / body_gz_scan / body_gz_pack (dbtk_gz.h) on the emulated lanes: `text` -> gzip members back to back in `packed`.
// Returns the number of bytes (< 0: packed_cap too small).
int64_t emu_gz(const uint8_t* text, uint64_t n, uint8_t* packed, uint64_t packed_cap, uint32_t grid) {
    const uint32_t nmem = (uint32_t)((n + GZ_MEMBER - 1) / GZ_MEMBER);
    std::vector<uint32_t> out((size_t)(nmem + 1) * GZ_STRIDE / 4 + 16, 0);
    std::vector<uint32_t> out_len(nmem + 2 + ING_SCAN_BLOCKS, 0), tab(288);
    std::vector<uint8_t> tx(n + 8);
    if (n) memcpy(tx.data(), text, n);
    gz_tables(tab.data());
    uint64_t total[2] = {n, 0}, ptotal = 0;
    std::vector<uint8_t> pk((size_t)(nmem + 1) * GZ_STRIDE);
    GzArgs a{tx.data(), total, (uint8_t*)out.data(), out_len.data(), tab.data(), pk.data(), &ptotal, getenv("EMU_GZ_LZ") && atoi(getenv("EMU_GZ_LZ")) == 0 ? 0u : 1u};
    run_grid(grid ? grid : 2, 64, sizeof(GzSmem), [&](EmuX& x) { body_gz_member(x, a); });
    run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_gz_scan(x, a, 0); });
    run_grid(ING_SCAN_BLOCKS, 64, 0, [&](EmuX& x) { body_gz_scan(x, a, 1); });
    run_grid(grid ? grid : 2, 64, 0, [&](EmuX& x) { body_gz_pack(x, a); });
    if (ptotal > packed_cap) return -1;
    memcpy(packed, pk.data(), ptotal);
    return (int64_t)ptotal;
}
// wave_fmt_cigar / wave_fmt_annot (dbtk_walk.h: the -a / -ae strings by the whole wave) against the one-lane scans w_fmt_cigar /
// w_fmt_annot (the restatement of writeCigar / writeAnnot that the reference's own strings pin, tests/test_walk.py) on random edit
// scripts: long runs, D / I stretches of every parity, tokens starting at the last entry, lengths across the 64-entry chunks.
// Returns the number of mismatching strings.
uint64_t emu_selftest_fmt(uint64_t seed, uint64_t iters) {
    uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 7;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (uint64_t it = 0; it < iters; ++it) {
        const int sz = (int)(rnd() % 5 == 0 ? rnd() % 4 : rnd() % (WCAP - 2));
        std::vector<uint8_t> t(WCAP + 8, 0), g(WCAP + 8, 0), tr(WCAP + 8, 0), o1(WTXT + 8), o2(WTXT + 8), o3(WTXT + 8), o4(WTXT + 8);
        const char types[] = "==*XDI=DI?";
        const char bases[] = "ACGT*";
        const char ann[] = "==..**=.x";
        int i = 0;
        const uint64_t style = rnd() % 3;
        while (i < sz) {
            const uint8_t c = style == 0 ? (uint8_t)'=' : (uint8_t)types[rnd() % 10];
            int run = (c == '=' || c == '*') ? (int)(rnd() % (style == 2 ? 4 : 130)) + 1 : (int)(rnd() % 3) + 1;
            for (; run > 0 && i < sz; --run, ++i) {
                t[i] = (c == '?') ? (uint8_t)'q' : c;
                if ((c == 'D' || c == 'I') && rnd() % 2) t[i] = c == 'D' ? 'I' : 'D';  // alternating stretches of both parities
                g[i] = rnd() % 7 == 0 ? 0 : (uint8_t)bases[rnd() % 5];
            }
            if (style == 0 && rnd() % 3 == 0 && i < sz) { t[i] = (uint8_t)types[3 + rnd() % 3]; g[i] = (uint8_t)bases[rnd() % 4]; ++i; }
        }
        i = 0;
        while (i < sz) {
            const uint8_t c = (uint8_t)ann[rnd() % 9];
            int run = (int)(rnd() % 140) + 1;
            for (; run > 0 && i < sz; --run, ++i) tr[i] = c;
        }
        const uint32_t n1 = w_fmt_cigar(t.data(), g.data(), sz, o1.data()), n2 = w_fmt_annot(tr.data(), sz, o2.data());
        uint32_t n3 = 0, n4 = 0;
        run_grid(1, 64, 0, [&](EmuX& x) {
            const uint32_t a = wave_fmt_cigar(x, t.data(), g.data(), sz, o3.data()), b = wave_fmt_annot(x, tr.data(), sz, o4.data());
            if (x.lane() == 0) { n3 = a; n4 = b; }
        });
        if (n1 != n3 || memcmp(o1.data(), o3.data(), n1)) ++bad;
        if (n2 != n4 || memcmp(o2.data(), o4.data(), n2)) ++bad;
    }
    return bad;
}
}  // extern "C"
