#!/usr/bin/env python3
"""Diagnostic: the probe kernel on the bench's release-scale RPGG — all-hit reads (what k_probe's roofline figure is quoted
on) and the WGS-like headline mix — for one or several builds of the library, each in its own child process.
    python tools/probe_bench.py [--nloci 80000] [--mix-reads 4000000] [--reads 10000000] [--libs a.so b.so ...] [--env K=V ...]
Every variant's counts are compared with the first one's (same reads): a tuning variant that changes results is flagged.
With a -DDBTK_STAMPS build the per-phase cycles per read are printed too."""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    import numpy as np
    import torch  # (before the library: both bring a HIP runtime, torch's has to come first)
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("danbing-tk_amd")
    abi = pkg.abi
    lib = pkg.Dbtk(a.lib) if a.lib else pkg.Dbtk()
    syn = pkg.Synth(nloci=a.nloci, k=a.k)
    arr = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(arr), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    p = abi.default_params(ksize=a.k, cthreshold=45, okam=0)
    p.diag = int(os.environ.get("DBTK_DIAG", a.diag))
    ctx = lib.context(g, p)
    out = dict(lib=a.lib or "default", env={k: v for k, v in os.environ.items() if k.startswith("DBTK_")})
    for name, nreads, hit in (("all_hit", a.mix_reads, 1.0), ("headline", a.reads, 0.02)):
        if not nreads:
            continue
        npairs = nreads // 2
        seq, off = syn.reads(npairs, hit_frac=hit, seed=2 if hit == 1.0 else 1)
        d_seq = torch.from_numpy(seq).cuda()
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for _ in range(2):
            ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
        ctx.synchronize()
        rounds = []
        for _ in range(a.rounds):  # (several timed rounds: a box can change its speed under the run)
            ctx.reset(); ctx.timers_reset()
            for _ in range(a.steps):
                ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
            ctx.synchronize()
            kt = ctx.kernel_times()
            rounds.append({k: round(v[0] / max(v[1], 1), 3) for k, v in kt.items() if v[1]})
        r = ctx.counts()
        dig = hashlib.sha256(r["counts"].tobytes() + r["kmc"].tobytes() + r["nmapread"].tobytes() + r["counters"].tobytes()).hexdigest()[:16]
        ctr = r["counters"]
        res = dict(kernels={k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}, k_probe_rounds=[r.get("k_probe") for r in rounds], digest=dig,
                   lookups_per_step=int(ctr[abi.C_NHASH1]) // a.steps, survivors_per_step=int(ctr[abi.C_SURVIVORS]) // a.steps)
        pr = res["kernels"].get("k_probe", 0.0)
        if pr:
            res["k_probe_gbs"] = round(12.0 * res["lookups_per_step"] / (pr * 1e-3) / 1e9, 1)
            res["k_probe_frac"] = round(res["k_probe_gbs"] / 8000.0, 4)
        if a.lib and "stamps" in a.lib:
            st = np.zeros(48, np.uint64)
            lib.L.dbtk_debug_stamps.argtypes = [C.c_void_p, abi.u64p]
            lib.L.dbtk_debug_stamps(ctx.h, st.ctypes.data_as(abi.u64p))
            nrows = 2.0 * float(ctr[abi.C_SURVIVORS]) * (a.steps * a.rounds + 2) / a.steps  # (the stamps are not reset with the counters)
            # (lean kernel | locus-resident kernel: the two share the stamp words)
            k2 = {43: "loop | barriers+image", 40: "fetch+pack(+windows+hashes)", 16: "minimizers | k-mers+image look-ups", 18: "level 1 | row stats+stores+queue", 41: "level 2 | queue against the index", 42: "results | item headers + commits", 1: "- | (fused) assignTRkmc + counts", 44: "(fused) verdict + assignTRkmc | -", 45: "(fused) counter window + atomics issued | -"}
            res["stamps_cycles_per_read"] = {n: round(float(st[i]) / max(nrows, 1.0)) for i, n in k2.items()}
        out[name] = res
        del d_seq, d_off
    ctx.close()
    print("PROBE_BENCH " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nloci", type=int, default=80000)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--mix-reads", type=int, default=4_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--diag", type=int, default=0)
    ap.add_argument("--libs", nargs="*", default=[None])
    ap.add_argument("--env", nargs="*", default=[], help="K=V[,K=V...] per extra variant of the FIRST lib")
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.child:
        return child(a)
    variants = [(l, {}) for l in a.libs] + [(a.libs[0], dict(kv.split("=", 1) for kv in e.split(","))) for e in a.env]
    first = None
    for lib, env in variants:
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--nloci", str(a.nloci), "--k", str(a.k), "--mix-reads", str(a.mix_reads),
               "--reads", str(a.reads), "--steps", str(a.steps), "--rounds", str(a.rounds), "--diag", str(a.diag)] + (["--lib", lib] if lib else [])
        r = subprocess.run(cmd, env={**os.environ, "DBTK_LANES": os.environ.get("DBTK_LANES", "1"), **env}, capture_output=True, text=True)  # (one lane: a kernel's duration must not include its neighbour's)
        line = [l for l in r.stdout.splitlines() if l.startswith("PROBE_BENCH ")]
        if not line:
            print(f"variant {lib} {env}: FAILED rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}", flush=True)
            continue
        d = json.loads(line[0][12:])
        d["env"] = env
        if first is None:
            first = d
        for mix in ("all_hit", "headline"):
            if mix in d and mix in first:
                d[mix]["same_counts_as_first"] = d[mix]["digest"] == first[mix]["digest"]
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
