#!/usr/bin/env python3
"""bench.py — `danbing-tk align` hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[1], named in config.workload): the release-scale
synthetic RPGG of SURVEY.md 8(d) (80 000 loci, ~1.4e8 index keys; the real
release RPGG is not available offline), replicated per GPU, and 10 M synthetic
150 bp PE reads per GPU (WGS-like mix: `--hit-frac` of the pairs tiled from the
loci with 0.1-0.5 % substitutions, the rest uniform random), aligned with
`-k 21 -kf 4 1 -cth 45 -ka`.  A "step" is one pass of the hot path (encode ->
subfilter -> kfilter probe -> vote -> assign -> count) over the rank's resident
read set; reads are in HBM before the timed region.  N > 1: reads shard across
ranks (weak scaling, no data-path collective); the accumulators are summed once
at the end with one RCCL all-reduce, inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: `roofline` (dominant kernel, algorithmic bytes / HIP-event time) and
`cpu_baseline` (the oracle, a plain-C port of the reference path, timed on one
host core over a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PMC_SUMMARY = "r01i_final_pmc.csv"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nloci", type=int, default=80000, help="loci of the synthetic RPGG (80000 = release scale)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (pairs = reads/2)")
    ap.add_argument("--hit-frac", type=float, default=0.02, help="fraction of pairs drawn from the loci (WGS-like: 0.02)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--parity-pairs", type=int, default=100000, help="pairs of the CPU-baseline sample whose oracle result is compared with the HIP path (0 = skip; needs --cpu-seconds > 0)")
    ap.add_argument("--lanes", type=int, default=1, choices=(1, 2, 3),
                    help="streams the context alternates successive batches on.  2 (the library's default) overlaps one batch's encode kernel "
                         "with the other's probe kernel (+9 %% reads/s) but then a kernel's launch duration includes its neighbour's work, "
                         "so the roofline measurement runs on one lane")
    ap.add_argument("--timer-every", type=int, default=4, help="record the per-kernel HIP events on every n-th step (each step's 8 records cost ~30 us)")
    ap.add_argument("--extra-lanes", action="store_true", help="after the timed region, also run the same steps on two overlapped lanes (the library's default) and report them as `two_lanes`")
    ap.add_argument("--lib", default=None, help="diagnostic: another build of libdbtk_hip.so (tuning variants)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    os.environ["DBTK_LANES"] = str(args.lanes)
    pkg = importlib.import_module("danbing-tk_amd")
    par = importlib.import_module("danbing-tk_amd.parallel")
    abi = pkg.abi
    dbtk = pkg.Dbtk(args.lib) if args.lib else pkg.Dbtk()  # raises if the HIP extension is missing: no CPU fallback
    log = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    # ---- workload: RPGG replica per GPU, read shard per rank
    ncpu = os.cpu_count() or 8
    nth = max(2, ncpu // world)
    t0 = time.time()
    syn = pkg.Synth(nloci=args.nloci, k=21, flank=700, seed=20250808, nthreads=nth)
    arrs = syn.arrays()
    log(f"synthetic RPGG: {args.nloci} loci, {arrs.nkeys} index keys in {time.time() - t0:.1f}s ({nth} threads)")
    t0 = time.time()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = pkg.Rpgg(dbtk, h)
    params = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0)
    ctx = dbtk.context(g, params, device=local_rank)
    log(f"handle + HBM tables: {g.ntrkmers} TR k-mers, {time.time() - t0:.1f}s")
    npairs = args.reads // 2
    rlen = 150
    t0 = time.time()
    seq, off = syn.reads(npairs, rlen=rlen, hit_frac=args.hit_frac, seed=1, first_pair=rank * npairs, nthreads=nth)
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    log(f"reads: {npairs} pairs/GPU resident in HBM, {time.time() - t0:.1f}s")

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    acc_ptr, acc_n = ctx.accum_buffer()
    acc_t, acc_inplace = None, False
    if world > 1:
        try:  # a tensor over the context's accumulator itself: the reduce then needs no staging copies
            class _Acc:
                __cuda_array_interface__ = {"shape": (acc_n,), "typestr": "<i8", "data": (acc_ptr, False), "version": 2}
            acc_t = torch.as_tensor(_Acc(), device=dev)
            acc_inplace = acc_t.data_ptr() == acc_ptr
        except Exception:
            acc_inplace = False
        if not acc_inplace:
            acc_t = torch.empty(acc_n, dtype=torch.int64, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, rlen)

    def reduce_counts():
        """The one exchange of the path: sum of the per-GPU accumulators over xGMI."""
        if world == 1:
            return
        ctx.synchronize()
        if not acc_inplace:
            assert hip.hipMemcpy(acc_t.data_ptr(), acc_ptr, acc_n * 8, 3) == 0
        par.allreduce_accum(acc_t)  # RCCL sum; int64 adds wrap exactly like the reference's uint64 atomics
        if not acc_inplace:
            assert hip.hipMemcpy(acc_ptr, acc_t.data_ptr(), acc_n * 8, 3) == 0

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    reduce_counts()  # untimed: RCCL communicator / kernel warm-up (the reduced values are discarded by the reset)
    ctx.reset()
    ctx.timers_reset()
    ctx.timers_enable(args.timer_every)  # HIP-event pairs around the kernels of every n-th step of the timed region
    if os.environ.get("DBTK_NO_TIMERS"):  # diagnostic: cost of the per-kernel event records themselves
        ctx.timers_enable(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    if world > 1:  # per-rank work counters, before the reduce sums them over ranks (96 B device->host)
        local_ctr = ctx.counters()
    reduce_counts()
    barrier()
    dt = time.perf_counter() - t0
    ktimes = ctx.kernel_times()      # HIP events recorded inside the timed region, read after it
    if world == 1:
        local_ctr = ctx.counters()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_reads = 2 * npairs * args.steps * world
    value = total_reads / dt

    # ---- roofline of the dominant kernel (this rank): algorithmic bytes of SURVEY.md 8(d) per launch
    ctr = local_ctr.astype(np.float64)  # totals over the timed steps
    alg = {
        # every read byte once + 12 B per subfilter probe
        "k_encode_subfilter": ctr[abi.C_BASES] + 12.0 * (ctr[abi.C_ALGO_PROBES] - ctr[abi.C_NHASH1]),
        # the probe kernel: 12 B (8 B key + 4 B value) per kfilter look-up the reference performs
        "k_probe": 12.0 * ctr[abi.C_NHASH1],
        # resolve: 4 B per vv word + 8 B per classified k-mer + 16 B per count increment
        "k_pair": 4.0 * ctr[abi.C_ALGO_VV] + 8.0 * ctr[abi.C_ALGO_CLS] + 16.0 * ctr[abi.C_ALGO_INC],
    }
    per_kernel = {}
    if "k_pair_usual" in ktimes:  # the resolve stage is two kernels (usual pairs, then the rest): priced together
        u, g_ = ktimes.pop("k_pair_usual"), ktimes.get("k_pair", (0.0, 0))
        ktimes["k_pair"] = (u[0] + g_[0], max(u[1], g_[1]))
        ktimes["k_pair: usual-pair part"] = u
    # the encode stage in its binned form is three launches (encode + sort the filter queries by partition, filter + exact
    # look-ups, candidates); "k_encode_subfilter" is timed around all three and priced as one, the parts are listed beside it
    for part in ("k_encode_bin", "k_filter_bins", "k_subfilter_cand"):
        if part in ktimes:
            t = ktimes.pop(part)
            if t[1]:
                ktimes[f"k_encode_subfilter: {part}"] = t
    # the event pairs sit around the kernels of every `timer_every`-th step: n timed launches stand for n * steps / timed steps launches
    timed_steps = (args.steps + args.timer_every - 1) // max(args.timer_every, 1) if args.timer_every > 1 else args.steps
    for name, (ms, n) in ktimes.items():  # large steps run as several sub-batch launches: price per launch
        avg = ms / max(n, 1)
        launches = n * args.steps / max(timed_steps, 1)
        per_launch = alg.get(name, 0.0) / max(launches, 1)
        per_kernel[name] = dict(avg_ms=avg, launches=launches, timed_launches=n, algorithmic_bytes=per_launch,
                                gbs=(per_launch / (avg * 1e-3) / 1e9) if avg > 0 else 0.0)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_ms"] * per_kernel[k]["launches"])
    # HBM bytes per launch of the dominant kernel from the committed PMC summary of this same command (a separate
    # rocprofv3 --pmc FETCH_SIZE pass, profiles/README.md), corrected as MI355X_MICROARCH.md prescribes for gfx950: a wide
    # coalesced stream is reported at half its bytes, so half of the read bytes (the only such stream of K1) is added back.
    traffic, traffic_src = None, None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_SUMMARY)
    if dom == "k_encode_subfilter" and args.reads == 10_000_000 and abs(args.hit_frac - 0.02) < 1e-9 and os.path.exists(pmc):
        for line in open(pmc):
            f = line.strip().split(",")
            if f[0] == dom and f[1] == "FETCH_SIZE":
                traffic = float(f[3]) * 1024.0 + 0.5 * ctr[abi.C_BASES] / max(per_kernel[dom]["launches"], 1)
                traffic_src = f"profiles/{PMC_SUMMARY}: FETCH_SIZE mean per launch (KB) x 1024 + half of the coalesced read stream"
    # the same summary's TCC_MISS_sum: L2-miss requests per launch — the resource K1 actually saturates (tools/lat.hip: the chip
    # serves about 50-58 G random requests/s whatever is behind them)
    requests = None
    if traffic is not None:
        for line in open(pmc):
            f = line.strip().split(",")
            if f[0] == dom and f[1] == "TCC_MISS_sum":
                requests = dict(per_launch=float(f[3]), rate_per_s=float(f[3]) / (per_kernel[dom]["avg_ms"] * 1e-3),
                                ceiling_per_s="5.0e10-5.8e10 (tools/lat.hip)", source=f"profiles/{PMC_SUMMARY}: TCC_MISS_sum")
    roof = dict(bound="hbm", kernel=dom, achieved=per_kernel[dom]["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                frac=per_kernel[dom]["gbs"] / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src, l2_miss_requests=requests,
                algorithmic_bytes_per_launch=per_kernel[dom]["algorithmic_bytes"], avg_ms=per_kernel[dom]["avg_ms"],
                kernels=per_kernel)

    # ---- untimed extra (--extra-lanes, N=1 only; off by default so that a profile of the default command holds the timed launches
    # and nothing else): the same steps on the library's default of two overlapped lanes (DBTK_LANES=2; the timed
    # region above runs on one lane so that a kernel's launch duration is its own) — reported beside `value`, never as it
    two_lanes = None
    if world == 1 and args.lanes == 1 and args.extra_lanes:
        os.environ["DBTK_LANES"] = "2"
        ctx2 = dbtk.context(g, params, device=local_rank)
        os.environ["DBTK_LANES"] = "1"
        ctx2.timers_enable(False)
        for _ in range(args.warmup):
            ctx2.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, rlen)
        ctx2.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            ctx2.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, rlen)
        ctx2.synchronize()
        dt2 = time.perf_counter() - t2
        two_lanes = dict(value=2 * npairs * args.steps / dt2, unit="reads/s", ms_per_step=dt2 / args.steps * 1e3,
                         note="untimed extra: same steps, context with DBTK_LANES=2 (successive batches alternate between two streams)")
        ctx2.close()

    out = None
    if rank == 0:
        cpu = None
        parity = None
        if world == 1 and args.cpu_seconds > 0:
            # ---- the cpu_baseline leg: the oracle (oracle/dbtk_oracle.c, the checker) timed on a bounded sample of the same
            # workload; its result on the first chunk doubles as this run's parity check of the HIP path.
            import bind
            try:
                orc = bind.Oracle()
            except OSError:  # the checker is not built (oracle/liboracle.so: __graft_entry__.build() makes it): build it now
                import subprocess
                subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"), "oracle"], check=False,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                orc = bind.Oracle()
            t0 = time.time()
            go = orc.from_arrays(arrs)
            log(f"oracle tables: {time.time() - t0:.1f}s")
            chunk, done, t_cpu = 250_000, 0, 0.0
            first = min(args.parity_pairs, npairs) if args.parity_pairs > 0 else 0
            while done < npairs and t_cpu < args.cpu_seconds:
                n = first if (done == 0 and first) else min(chunk, npairs - done)
                t1 = time.perf_counter()
                o = orc.align(go, params, seq[2 * done * rlen:2 * (done + n) * rlen], off[:2 * n + 1], trace=False)
                t_cpu += time.perf_counter() - t1
                if done == 0 and first:
                    ctx.reset()
                    ctx.align(seq[:2 * n * rlen], off[:2 * n + 1])
                    r = ctx.counts()
                    co = np.zeros(g.ntrkmers, np.uint64)
                    np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
                    ok = bool((co == r["counts"]).all() and (o["kmc"] == r["kmc"]).all() and (o["nmapread"] == r["nmapread"]).all()
                              and (o["counters"] == r["counters"]).all())
                    parity = dict(pairs=n, bit_exact=ok)
                    log(f"parity on {n} pairs: {'bit-exact' if ok else 'MISMATCH'}")
                    if not ok:
                        raise SystemExit("GPU result differs from the oracle")
                done += n
            cpu = dict(value=2 * done / t_cpu, unit="reads/s", cores=1, kind="port",
                       sample=f"first {2 * done} reads of the same read set and RPGG, oracle/dbtk_oracle.c on 1 host core, {t_cpu:.1f} s",
                       checked=parity)
            log(f"cpu baseline: {cpu['value']:.0f} reads/s on 1 core")
            orc.free(go)
        out = {
            "metric": "paired reads/sec aligned to RPGG (k=21)", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"release-scale synthetic RPGG ({args.nloci} loci, {arrs.nkeys} index keys, {g.ntrkmers} TR k-mers) "
                                   f"replicated per GPU; {args.reads} x 150bp PE reads per GPU per step, {args.hit_frac:.0%} of pairs from loci; "
                                   f"-k 21 -kf 4 1 -cth 45 -ka; RCCL all-reduce of counts at the end",
                       "lanes": args.lanes, "reads_per_gpu": args.reads, "read_len": rlen, "hit_frac": args.hit_frac, "k": 21, "cth": 45},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "two_lanes": two_lanes,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
