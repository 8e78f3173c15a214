#!/usr/bin/env python3
"""bench.py — `danbing-tk align` hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[1], named in config.workload): the release-scale synthetic RPGG of SURVEY.md 8(d)
(80 000 loci, ~1.4e8 index keys; the real release RPGG is not available offline), replicated per GPU, and 10 M
synthetic 150 bp PE reads per GPU (WGS-like mix: `--hit-frac` of the pairs tiled from the loci with 0.1-0.5 %
substitutions, the rest uniform random), aligned with `-k 21 -kf 4 1 -cth 45 -ka`.  A "step" is one pass of the hot
path (encode -> subfilter -> kfilter probe -> vote -> assign -> count) over the rank's resident read set; reads are in
HBM before the timed region.  N > 1: reads shard across ranks (weak scaling, no data-path collective); the accumulators
are summed once at the end with one RCCL all-reduce, inside the timed region.

`python bench.py --gpus N` without a launcher spawns the N ranks itself (torch.distributed.run as a child process,
before anything here touches a GPU) and relays rank 0's line.

ONE JSON line on rank 0 (contract in the task statement), with these extra objects:
  roofline      the dominant kernel of the timed region: algorithmic bytes / HIP-event time inside the region
  probe_roofline  the same figures for k_probe, the kernel the north star's 40 % target names
  mixes         N = 1: the all-hit mix (every pair from a locus: k_probe dominates) and the graph-walk mix
                (threading = 2, `-gc 85 3`: config 4's path on the k = 21 RPGG), each with its own ms/step and kernel table
  end_to_end    N = 1: the same workload through host buffers (PCIe) and through this repo's CLI (parse -> counts)
  cpu_baseline  the REFERENCE's own pthread binary (oracle/_ref/danbing-tk, compiled from /root/reference) on the same
                RPGG written out as the files it loads and a FASTA sample of the same reads, at -p 1 / 8 / all host
                threads; `port` beside it = the plain-C oracle on one core (which also parity-checks this run)
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = ("dbtk_kernels.h", "dbtk_probe2.h", "dbtk_locus.h", "dbtk_walkfast.h", "dbtk_ingest.h", "dbtk_gz.h", "dbtk_walk.h", "dbtk_tables.h", "dbtk_sort.h", "dbtk_assign.h", "dbtk_devx.h", "dbtk_hip.hip")


def kernel_source_hash():
    """Identifies the kernels a PMC summary was collected for: sha256 over the device sources."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "danbing-tk_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nloci", type=int, default=80000, help="loci of the synthetic RPGG (80000 = release scale)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (pairs = reads/2)")
    ap.add_argument("--hit-frac", type=float, default=0.02, help="fraction of pairs drawn from the loci (WGS-like: 0.02)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the one-core oracle leg (0 = skip it and the parity check)")
    ap.add_argument("--parity-pairs", type=int, default=100000, help="pairs whose oracle result is compared with the HIP path")
    ap.add_argument("--ref-reads", type=int, default=8_000_000, help="reads of the FASTA sample the reference binary is timed on (0 = skip the reference leg)")
    ap.add_argument("--ref-threads", type=int, nargs="*", default=None, help="-p values for the reference binary [1 8 <all host threads>]")
    ap.add_argument("--mix-reads", type=int, default=10_000_000, help="reads per step of the extra mixes (all-hit, walk: config 2's batch size); 0 = skip them")
    ap.add_argument("--mix-steps", type=int, default=5)
    ap.add_argument("--only-mix", choices=("all_hit", "walk", "genome", "k25"), default=None,
                    help="run this one of the extra mixes only (tools/profile_round.sh: one profiled command per mix)")
    ap.add_argument("--no-walk", action="store_true", help="skip the graph-walk mixes (threading = 2)")
    ap.add_argument("--k25-reads", type=int, default=10_000_000, help="reads per step of the k = 25 graph-walk mix (BASELINE config 4: pipeline/k25.json, -gc 85 3); 0 = skip it")
    ap.add_argument("--k25-parity-pairs", type=int, default=20000, help="pairs of the k = 25 mix whose oracle result (counts, counters, walk results) is compared")
    ap.add_argument("--ingest-reads", type=int, default=64_000_000, help="reads of the FASTA the command line's batch loop is timed on (end_to_end.cli_ingest); 0 = skip")
    ap.add_argument("--walk-long-reads", type=int, default=40_000_000, help="reads of the long all-hit FASTA the command line's walk is timed on with and without merged batches (end_to_end.cli_walk_emit.walk_long); 0 = skip")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="the timed step repeated for at least this long (`sustained`); 0 = skip")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (host buffers, CLI)")
    ap.add_argument("--lanes", type=int, default=1, choices=(1, 2, 3),
                    help="streams the context alternates successive batches on.  2 (the library's default) overlaps one batch's encode kernel "
                         "with the other's probe kernel but then a kernel's launch duration includes its neighbour's work, so the roofline "
                         "measurement runs on one lane")
    ap.add_argument("--timer-every", type=int, default=4, help="record the per-kernel HIP events on every n-th step")
    ap.add_argument("--extra-lanes", action="store_true", default=True, help="also run the timed steps on two overlapped lanes (the library's default outside this bench) and report them as `two_lanes` [on]")
    ap.add_argument("--no-extra-lanes", dest="extra_lanes", action="store_false")
    ap.add_argument("--lib", default=None, help="diagnostic: another build of libdbtk_hip.so (tuning variants)")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as children (torch.distributed.run, rendezvous on
    127.0.0.1) BEFORE this process has touched a GPU, wait for them, relay rank 0's JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def kernel_table(ktimes, alg, steps, timed_steps):
    """{kernel: avg ms, launches, algorithmic bytes per launch, GB/s}; the resolve stage's two kernels are priced together."""
    kt = dict(ktimes)
    if "k_pair_usual" in kt:
        u, g_ = kt.pop("k_pair_usual"), kt.get("k_pair", (0.0, 0))
        kt["k_pair"] = (u[0] + g_[0], max(u[1], g_[1]))
        kt["k_pair: usual-pair part"] = u
    if "k_probe" in kt:
        kt["k_probe: look-ups only"] = kt["k_probe"]  # (the same stage priced with its index look-ups alone: the round-4 figure's definition)
    out = {}
    for name, (ms, n) in kt.items():
        if not n:
            continue
        avg = ms / n
        launches = n * steps / max(timed_steps, 1)
        per_launch = alg.get(name, 0.0) / max(launches, 1)
        out[name] = dict(avg_ms=avg, launches=launches, timed_launches=n, algorithmic_bytes=per_launch,
                         gbs=(per_launch / (avg * 1e-3) / 1e9) if avg > 0 else 0.0)
    return out


def algorithmic_bytes(abi, ctr, walk_probes=0.0, ps=None, walk=False):
    """B = 2L + 12 P + 4 V + 8 A + 16 I, split over the kernels that do the work.  `ps` (dbtk_ctx_path_stats): the part
    of A and I the fused locus-resident probe kernel did itself (it resolves the usual pairs of its items: dbtk_locus.h) is priced with
    the probe stage, where its time is."""
    fa = float(ps["fused_cls"]) if ps else 0.0
    fi = float(ps["fused_inc"]) if ps else 0.0
    # v1.3 threading: k_pair hands a pair to the walk kernels BEFORE assignTRkmc / accumulate (dbtk_kernels.h, DBTK_STAGE_THREADING), and
    # every count increment of the step is the walk kernels' (exact counting, AQ.cpp:2189-2194): 16 I is priced there, once
    inc_pair = 0.0 if walk else ctr[abi.C_ALGO_INC] - fi
    # V: fillstats' vv words (DBTK_C_ALGO_VV) are the reference's on every path; the kernel that actually reads vv is k_pair — for the pairs it
    # handles (path statistic pair_vv) plus the words of the votes it holds (vote_vv).  The fused probe kernels read none: unpriced.
    v_pair = (float(ps.get("pair_vv", 0)) + float(ps.get("vote_vv", 0))) if ps else ctr[abi.C_ALGO_VV]
    return {
        "k_encode_subfilter": ctr[abi.C_BASES] + 12.0 * (ctr[abi.C_ALGO_PROBES] - ctr[abi.C_NHASH1]),
        "k_probe": 12.0 * ctr[abi.C_NHASH1] + 8.0 * fa + 16.0 * fi,
        "k_probe: look-ups only": 12.0 * ctr[abi.C_NHASH1],
        "k_pair": 4.0 * v_pair + 8.0 * (ctr[abi.C_ALGO_CLS] - fa) + 16.0 * inc_pair,
        # the walk: one graph look-up (8 B node + 1 B edge mask, the PREF.graph.umap entry) and one TR-set look-up (8 B) per
        # k-mer of both mates of every walked pair, + 16 B per count increment
        "k_walk_pairs": (walk_probes * 17.0 + 16.0 * ctr[abi.C_ALGO_INC]) if walk else 0.0,
    }


def roofline_of(name, table):
    k = table[name]
    return dict(bound="hbm", kernel=name, achieved=k["gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=k["gbs"] / HBM_PEAK_GBS,
                algorithmic_bytes_per_launch=k["algorithmic_bytes"], avg_ms=k["avg_ms"])


def pmc_summary():
    """The committed PMC summary (profiles/*_pmc.json, tools/profile_round.sh) that was collected for THESE kernel sources."""
    import glob
    h = kernel_source_hash()
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            d = json.load(open(fn))
        except ValueError:
            continue
        if d.get("kernel_source_hash") == h:
            return d, os.path.basename(fn)
    return None, None


def pmc_traffic(kernel, table, ctr_bases_per_launch):
    """HBM bytes per launch from the committed PMC summary — only if it was collected for THESE kernel sources."""
    d, name = pmc_summary()
    if d is None:
        return None, None, True
    k = d.get("kernels", {}).get(kernel)
    if not k or "FETCH_SIZE_KB" not in k:
        return None, None, True
    # guide's gfx950 correction: a wide coalesced stream is reported at half its bytes; K1's read stream is the only one
    add = 0.5 * ctr_bases_per_launch if kernel == "k_encode_subfilter" else 0.0
    return k["FETCH_SIZE_KB"] * 1024.0 + add, f"profiles/{name} ({d.get('command', '')})", False


# the kernels of a stage, by the names rocprofv3 reports: the probe stage is the locus-resident kernel's three classes of workgroup, the
# lean kernel that takes the rest, and the two kernels that make the work lists; the walk is its lean kernel (both forms) and the
# error-correcting one
STAGE_KERNELS = {"k_probe": ("k_probe", "k_loc_items", "k_loc_rest", "k_loc_split"), "k_walk_pairs": ("k_walk_fast", "k_walk_pairs")}


def pmc_mix(mix, stage):
    """Per STEP of the mix `mix`, summed over the kernels of `stage`, from the committed summary (None if stale / absent): HBM bytes
    fetched (FETCH_SIZE), L1 -> L2 read requests (TCP_TCC_READ_REQ), wave instructions, and the stage's duration by rocprofv3's own
    kernel trace (to be read next to the HIP-event figure of this run)."""
    d, name = pmc_summary()
    m = d and d.get("mixes", {}).get(mix)
    if not m or not m.get("steps"):
        return None
    pre = STAGE_KERNELS.get(stage, (stage,))
    ks = {k: v for k, v in m["kernels"].items() if k.startswith(pre)}
    if not ks:
        return None
    st = float(m["steps"])
    tot = lambda key: sum(v.get(key, 0.0) for v in ks.values())
    out = dict(source=f"profiles/{name}", steps=m["steps"], kernels=sorted(ks),
               traffic=tot("FETCH_SIZE_KB:sum") * 1024.0 / st if any("FETCH_SIZE_KB:sum" in v for v in ks.values()) else None,
               write_traffic=tot("WRITE_SIZE_KB:sum") * 1024.0 / st if any("WRITE_SIZE_KB:sum" in v for v in ks.values()) else None,
               insts_salu=tot("SQ_INSTS_SALU:sum") / st if any("SQ_INSTS_SALU:sum" in v for v in ks.values()) else None,
               l1_to_l2_read_requests=tot("TCP_TCC_READ_REQ_sum:sum") / st if any("TCP_TCC_READ_REQ_sum:sum" in v for v in ks.values()) else None,
               insts_valu=tot("SQ_INSTS_VALU:sum") / st if any("SQ_INSTS_VALU:sum" in v for v in ks.values()) else None,
               rocprof_ms=sum(v["calls"] * v["avg_ns"] for v in ks.values() if "avg_ns" in v) / st * 1e-6 if any("avg_ns" in v for v in ks.values()) else None)
    return out


MAX_LINE = 6000  # the driver reads the last 8 001 characters of stdout: the one JSON line must fit with room


def _r(v, sig=5):
    """Floats to `sig` significant digits (the line is read by people and a parser, not used for arithmetic)."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, (float, np.floating)):
        v = float(v)
        return float(f"{v:.{sig}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, np.integer):
        return int(v)
    return v


def _roof(r, formula=None):
    if not r:
        return None
    o = {k: _r(r.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "avg_ms") if k in r}
    if "traffic" in r:
        o["traffic"] = _r(r.get("traffic"))
    if r.get("traffic_source"):
        o["traffic_source"] = str(r["traffic_source"]).split(" ")[0]
    if "traffic_stale" in r:
        o["traffic_stale"] = r["traffic_stale"]
    if formula:
        o["formula"] = formula
    return o


def compact_line(out):
    """The ONE line the driver parses (< MAX_LINE characters).  Everything else — per-kernel tables of every mix, path statistics, the
    command line's legs with their stderr lines, every reference run — goes to bench_detail.json."""
    c = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    c["per_rank_ms_per_step"] = [_r(v, 6) for v in (out.get("per_rank_ms_per_step") or [])]
    c["config"] = out["config"]
    c["roofline"] = _roof(out.get("roofline"))
    if out.get("probe_roofline"):
        c["probe_roofline"] = _roof(out["probe_roofline"], "12*P of the k_probe launches of the timed steps / their HIP-event time (SURVEY 8d: look-ups only)")
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "usable_cpus", "kind", "host") if k in cb}
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:330]
        if cb.get("runs"):
            c["cpu_baseline"]["runs"] = [{"p": r["threads"], "reads_s": _r(r["value"], 4)} for r in cb["runs"]]
        if cb.get("port"):
            c["cpu_baseline"]["port_1core_reads_s"] = _r(cb["port"]["value"], 4)
    else:
        c["cpu_baseline"] = None
    c["parity"] = out.get("parity")
    if out.get("reduce_check") is not None:
        c["reduce_check"] = out["reduce_check"]
    if out.get("sustained"):
        c["sustained_reads_s"] = _r(out["sustained"]["value"])
    if out.get("two_lanes"):
        c["two_lanes_reads_s"] = _r(out["two_lanes"]["value"])
    if out.get("hbm_bytes_tables"):
        c["hbm_bytes_tables_total"] = out["hbm_bytes_tables"].get("total")
    mx = out.get("mixes") or {}
    if mx:
        cm = {}
        for name, m in mx.items():
            r = m.get("roofline") or {}
            e = {"reads_s": _r(m.get("value")), "ms_per_step": _r(m.get("ms_per_step")), "dominant": r.get("kernel"), "dominant_ms": _r(r.get("avg_ms")),
                 "frac": _r(r.get("frac"), 4)}
            pr = m.get("probe_lookups_roofline")
            if pr:
                e["probe_stage_frac_lookups_only"] = _r(pr.get("frac"), 4)
                e["probe_stage_ms"] = _r(pr.get("avg_ms"))
            kp = (r.get("kernels") or {}).get("k_pair")
            if kp:
                e["k_pair_ms"] = _r(kp.get("avg_ms"))
            if isinstance(m.get("parity"), dict):
                e["parity_bit_exact"] = m["parity"].get("bit_exact")
                e["parity_pairs"] = m["parity"].get("pairs")
            cm[name] = e
        c["mixes"] = cm
    e2 = out.get("end_to_end") or {}
    if e2:
        ce = {}
        hb = e2.get("host_buffers")
        if hb:
            ce["host_buffers_reads_s"] = _r(hb.get("value"))
        ci = e2.get("cli_ingest") or {}
        if ci:
            ce["cli_reads"] = ci.get("reads")
            ce["cli_first_pass_reads_s"] = _r((ci.get("device_reader") or {}).get("value"))
            ce["cli_later_pass_reads_s"] = _r((ci.get("device_reader_again") or {}).get("value"))
            ce["cli_same_outputs"] = ci.get("same_outputs")
        cl = e2.get("cli") or {}
        if cl:
            ce["cli_8M_reads_wall_s"] = _r(cl.get("wall_s"), 4)
            ce["cli_8M_reads_main_s"] = _r(cl.get("main_s"), 4)
        cw = e2.get("cli_walk_emit") or {}
        if cw:
            rate = lambda leg: _r(_loop_rate((cw.get(leg) or {}).get("batch_loop")))
            ce["cli_walk_reads_s"] = rate("walk")
            ce["cli_walk_ae_gz_reads_s"] = rate("walk_ae_gz")
            ce["aln_gz_bytes"] = cw.get("aln_gz_bytes")
            ce["aln_gz_zlib1_bytes"] = cw.get("aln_gz_zlib1_bytes")
            if cw.get("walk_k25"):
                ce["cli_walk_k25_reads_s"] = rate("walk_k25")
            wl = cw.get("walk_long") or {}
            if wl:
                ce["cli_walk_long_reads"] = wl.get("reads")
                ce["cli_walk_long_first_pass_reads_s"] = _r((wl.get("merged") or {}).get("value"))
                ce["cli_walk_long_later_pass_reads_s"] = _r((wl.get("merged_again") or {}).get("value"))
        rl = out["config"].get("read_len", 150)
        # a FASTA record of a 150-bp read with a short title is ~164 B; one GPU's PCIe 5.0 x16 link moves ~55 GB/s
        ce["pcie_ceiling_reads_s"] = _r(55e9 / (rl + 14))
        ce["note"] = ("`value` times reads already resident in HBM (contract); a FASTA feed over PCIe caps one GPU at pcie_ceiling_reads_s, "
                      "and the drop-in command line reaches cli_first_pass / cli_later_pass (file in /dev/shm, batch loop only)")
        c["end_to_end"] = ce
    c["bench_wall_s"] = _r(out.get("bench_wall_s"), 4)
    c["detail"] = "bench_detail.json"
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= MAX_LINE:  # never lose the contract's fields to a long note
        for k in ("end_to_end", "mixes"):
            if len(line) < MAX_LINE:
                break
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
    return line


def _loop_rate(batch_loop):
    """reads/s of the command line's batch loop from its 'ingest: … (12.34 M reads/s)' line."""
    m = re.search(r"\(([0-9.]+) M reads/s\)", batch_loop or "")
    return float(m.group(1)) * 1e6 if m else None


def usable_cpus():
    """CPUs this process may actually use: hardware threads, capped by the affinity mask and the cgroup CPU quota (the GPU boxes of
    this pool show 256 hardware threads and a quota of 16 CPUs: the reference's `-p 256` run is a 16-core run)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


def dense_gpu_half(dbtk, mix, syn, arrs, g, params, k, per_lg=120, pairs_per_locus=64):
    """A dense slice of the RPGG through the HIP path: `pairs_per_locus` pairs on each of up to per_lg loci of every image size (every
    class of the locus-resident kernels), 6 % of them chimeric / foreign — in a context created under DBTK_LOCUS_ALWAYS=1, so that the
    batch (tens of thousands of pairs on a release-scale RPGG) is sorted and offered to the locus path like a 10 M-read batch.
    Returns what dense_oracle_half compares."""
    import bind
    pkg = bind.pkg
    loci, classes = bind.dense_loci(arrs, k, per_lg=per_lg, seed=77 + k)
    n = pairs_per_locus * len(loci)
    seq, off = syn.reads_loci(n, loci, odd_frac=0.06, seed=300 + k)
    old = os.environ.get("DBTK_LOCUS_ALWAYS")
    os.environ["DBTK_LOCUS_ALWAYS"] = "1"
    try:
        ctx = dbtk.context(g, params, device=int(os.environ.get("LOCAL_RANK", "0")))
    finally:
        if old is None:
            del os.environ["DBTK_LOCUS_ALWAYS"]
        else:
            os.environ["DBTK_LOCUS_ALWAYS"] = old
    ctx.align(seq, off)
    r = ctx.counts()
    walking = params.threading == pkg.abi.THREADING_V13
    res, nres = (None, 0)
    if walking:
        res, _, nres = ctx.walk_results(n)
    ps = ctx.path_stats()
    ctx.close()
    return dict(mix=mix, k=k, params=params, seq=seq, off=off, n=n, counts=r, res=res, nres=nres, path=ps, classes=classes, nloci_slice=len(loci),
                ntr=g.ntrkmers, order=g.output_order().astype(np.int64), nloci=g.nloci, walking=walking)


def dense_oracle_half(orc, go, chk, log):
    import bind
    abi = bind.abi
    t0 = time.time()
    r = chk["counts"]
    if chk["walking"]:
        o = orc.align_walk(go, chk["params"], chk["seq"], chk["off"], with_recs=False)
    else:
        o = orc.align(go, chk["params"], chk["seq"], chk["off"], trace=False)
    co = np.zeros(chk["ntr"], np.uint64)
    np.add.at(co, chk["order"], o["counts_file"])
    ok = bool((co == r["counts"]).all() and (o["counters"] == r["counters"]).all())
    if chk["walking"]:
        ok = ok and chk["nres"] == o["nres"] and bind.walk_res_equal(chk["res"], o["res"], chk["nres"], chk["nloci"], every_mate=False) >= 0
    else:
        ok = ok and bool((o["kmc"] == r["kmc"]).all() and (o["nmapread"] == r["nmapread"]).all())
    ps = chk["path"]
    surv = int(r["counters"][abi.C_SURVIVORS])
    on_locus_path = (surv - ps["probe_rest"]) / max(surv, 1)
    out = dict(pairs=chk["n"], loci=chk["nloci_slice"], image_classes=chk["classes"], bit_exact=ok, pairs_on_locus_path=on_locus_path,
               probe_items=ps["probe_items"], fused_done=ps["fused_done"], fused_redone=ps["fused_redone"],
               walk_items=ps["walk_items"] if chk["walking"] else None, walked=int(o["nres"]) if chk["walking"] else None,
               note="dense slice (64 pairs per locus, 6 % chimeric / foreign) against oracle/: counts, kmc, nmapread, all counters"
                    + (", walk results" if chk["walking"] else ""))
    log(f"{chk['mix']}: dense-slice parity on {chk['n']} pairs over {chk['nloci_slice']} loci: {'bit-exact' if ok else 'MISMATCH'}; "
        f"{100 * on_locus_path:.1f} % of the pairs on the locus path, {ps['fused_done']} resolved there ({time.time() - t0:.0f}s)")
    if not ok:
        raise SystemExit(f"GPU result of the {chk['mix']} dense slice differs from the oracle")
    return out


def time_steps(ctx, fn, steps, warmup):
    for _ in range(warmup):
        fn()
        ctx.synchronize()  # (per step: which probe path a batch takes is decided from the batch BEFORE it, once that one has run)
    ctx.reset()
    ctx.timers_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.synchronize()
    return time.perf_counter() - t0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test-only (a one-GPU box): every rank on device 0, the reduce over gloo through host staging (RCCL refuses two ranks on one
    # device) — the N > 1 code of this file (sharded read sets, barriers, MAX over ranks, the accumulators' all-reduce) then runs
    # with two real ranks; no scaling number comes out of it
    one_device = os.environ.get("DBTK_BENCH_ALL_ON_DEVICE0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    # the collective path (RCCL process group, barrier, all-reduce of the accumulators): with more than one rank — or, to
    # exercise exactly that code on a one-GPU box, under a launcher with DBTK_BENCH_FORCE_DIST=1 (a world of one)
    use_dist = world > 1 or (os.environ.get("DBTK_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ranks_seen = dist.get_world_size()

    os.environ["DBTK_LANES"] = str(args.lanes)
    pkg = importlib.import_module("danbing-tk_amd")
    par = importlib.import_module("danbing-tk_amd.parallel")
    abi = pkg.abi
    dbtk = pkg.Dbtk(args.lib) if args.lib else pkg.Dbtk()  # raises if the HIP extension is missing: no CPU fallback
    log = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    t_start = time.time()
    solo = world == 1
    do_mixes = solo and args.mix_reads > 0
    do_walk = do_mixes and not args.no_walk
    do_ref = solo and rank == 0 and args.ref_reads > 0 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "danbing-tk"))

    # ---- workload: RPGG replica per GPU, read shard per rank
    ncpu = os.cpu_count() or 8
    nth = max(2, ncpu // world)
    t0 = time.time()
    syn = pkg.Synth(nloci=args.nloci, k=21, flank=700, seed=20250808, nthreads=nth)
    if do_walk:
        syn.graph(nth)
    arrs = syn.arrays()
    log(f"synthetic RPGG: {args.nloci} loci, {arrs.nkeys} index keys{', graph' if do_walk else ''} in {time.time() - t0:.1f}s ({nth} threads)")
    npairs = args.reads // 2
    rlen = 150
    t0 = time.time()
    seq, off = syn.reads(npairs, rlen=rlen, hit_frac=args.hit_frac, seed=1, first_pair=rank * npairs, nthreads=nth)
    log(f"reads: {npairs} pairs/GPU generated, {time.time() - t0:.1f}s")

    # ---- the reference binary needs the RPGG as files and a FASTA sample (written now; its runs come after the GPU legs)
    ref_dir, ref_procs, ref_results = None, [], []
    if do_ref:
        import ref_baseline
        ref_dir = tempfile.mkdtemp(prefix="dbtk_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        t0 = time.time()
        syn.write_files(os.path.join(ref_dir, "pan"))
        nref = min(args.ref_reads // 2, npairs)
        syn.write_fasta(seq, nref, os.path.join(ref_dir, "reads.fa"), rlen=rlen)
        syn.write_fasta(seq, max(nref // 4, 1), os.path.join(ref_dir, "reads_small.fa"), rlen=rlen)
        log(f"reference inputs: RPGG files + {2 * nref}-read FASTA in {ref_dir}, {time.time() - t0:.1f}s")
        pvals = args.ref_threads if args.ref_threads else [1, 8, ncpu]
        import threading
        def bg(pv, fa):
            ref_results.append(dict(ref_baseline.run_reference(os.path.join(ROOT, "oracle", "_ref", "danbing-tk"), ref_dir, fa, pv), fasta=fa))
        def start_small_ref_legs():
            """-p 1 and -p 8 (9 host threads): started only after every GPU leg and CLI leg is done, next to the one-core oracle leg;
            the all-threads run comes last, alone"""
            for pv in [p for p in pvals if p <= 8]:
                th = threading.Thread(target=bg, args=(pv, "reads_small.fa" if pv == 1 else "reads.fa"))
                th.start()
                ref_procs.append(th)

    t0 = time.time()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = pkg.Rpgg(dbtk, h)
    params = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0)
    ctx = dbtk.context(g, params, device=local_rank)
    hbm_tables = ctx.table_bytes()
    log(f"handle + HBM tables: {g.ntrkmers} TR k-mers, {time.time() - t0:.1f}s; {hbm_tables.get('total', 0) / 1e9:.1f} GB of tables")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    acc_ptr, acc_n = ctx.accum_buffer()
    acc_t, acc_inplace = None, False
    if use_dist:
        try:  # a tensor over the context's accumulator itself: the reduce then needs no staging copies
            class _Acc:
                __cuda_array_interface__ = {"shape": (acc_n,), "typestr": "<i8", "data": (acc_ptr, False), "version": 2}
            acc_t = torch.as_tensor(_Acc(), device=dev)
            acc_inplace = acc_t.data_ptr() == acc_ptr
        except Exception:
            acc_inplace = False
        if one_device:
            acc_t, acc_inplace = torch.empty(acc_n, dtype=torch.int64), False  # (host staging for gloo)
        elif not acc_inplace:
            acc_t = torch.empty(acc_n, dtype=torch.int64, device=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, rlen)

    def reduce_counts():
        """The one exchange of the path: sum of the per-GPU accumulators over xGMI."""
        if not use_dist:
            return
        ctx.synchronize()
        if not acc_inplace:
            assert hip.hipMemcpy(acc_t.data_ptr(), acc_ptr, acc_n * 8, 4) == 0  # (hipMemcpyDefault: device or host staging)
        par.allreduce_accum(acc_t)  # RCCL sum; int64 adds wrap exactly like the reference's uint64 atomics
        if not acc_inplace:
            assert hip.hipMemcpy(acc_ptr, acc_t.data_ptr(), acc_n * 8, 4) == 0

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    reduce_counts()  # untimed: RCCL communicator / kernel warm-up (the reduced values are discarded by the reset)
    ctx.reset()
    ctx.timers_reset()
    ctx.timers_enable(args.timer_every)  # HIP-event pairs around the kernels of every n-th step of the timed region
    if os.environ.get("DBTK_NO_TIMERS"):
        ctx.timers_enable(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    if world > 1:  # per-rank work counters, before the reduce sums them over ranks
        local_ctr = ctx.counters()
    reduce_counts()
    barrier()
    dt = time.perf_counter() - t0
    ktimes = ctx.kernel_times()      # HIP events recorded inside the timed region, read after it
    if world == 1:
        local_ctr = ctx.counters()
    per_rank_ms = [dt / args.steps * 1e3]
    if use_dist:
        # every rank's own time for the same K steps (barrier to barrier), so that a first multi-GPU run is diagnosable from its one
        # line: `ms_per_step` is their maximum
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_device else dev)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    reduce_check = None
    if use_dist and os.environ.get("DBTK_BENCH_REDUCE_CHECK") == "1":
        # (tests) the reduced accumulators against the oracle on EVERY rank's reads: sum over ranks x steps
        red = ctx.counts()
        if rank == 0:
            import bind
            orc = bind.Oracle()
            go = orc.from_arrays(arrs)
            co = np.zeros(g.ntrkmers, np.uint64)
            kmc = np.zeros(g.nloci, np.uint64); nmr = np.zeros(g.nloci, np.uint64); ctrs = None
            for r_ in range(world):
                s_, o_ = syn.reads(npairs, rlen=rlen, hit_frac=args.hit_frac, seed=1, first_pair=r_ * npairs, nthreads=nth)
                oo = orc.align(go, params, s_, o_, trace=False)
                np.add.at(co, g.output_order().astype(np.int64), oo["counts_file"])
                kmc += oo["kmc"].astype(np.uint64); nmr += oo["nmapread"].astype(np.uint64)
                ctrs = oo["counters"].astype(np.uint64) if ctrs is None else ctrs + oo["counters"].astype(np.uint64)
            st = np.uint64(args.steps)
            ok = bool((co * st == red["counts"]).all() and (kmc * st == red["kmc"]).all() and ((nmr * st).astype(np.uint32) == red["nmapread"]).all()
                      and (ctrs * st == red["counters"]).all())
            reduce_check = dict(ranks=world, steps=args.steps, pairs_per_rank=npairs, bit_exact=ok)
            log(f"reduce check over {world} ranks x {args.steps} steps: {'bit-exact' if ok else 'MISMATCH'}")
            orc.free(go)
    total_reads = 2 * npairs * args.steps * world
    value = total_reads / dt
    log(f"timed region: {args.steps} steps, {dt / args.steps * 1e3:.3f} ms/step, {value / 1e9:.2f} G reads/s on {world} GPU(s)")

    # ---- roofline (this rank): algorithmic bytes of SURVEY.md 8(d) per launch / HIP-event time
    ctr = local_ctr.astype(np.float64)
    timed_steps = (args.steps + args.timer_every - 1) // max(args.timer_every, 1) if args.timer_every > 1 else args.steps
    ps0 = ctx.path_stats()  # (reset with the accumulators after the warm-up: the timed steps' pairs only)
    table = kernel_table(ktimes, algorithmic_bytes(abi, ctr, ps=ps0), args.steps, timed_steps)
    dom = max((k for k in table if ":" not in k), key=lambda k: table[k]["avg_ms"] * table[k]["launches"])
    roof = roofline_of(dom, table)
    traffic, traffic_src, stale = pmc_traffic(dom, table, ctr[abi.C_BASES] / max(table[dom]["launches"], 1))
    roof.update(traffic=traffic, traffic_source=traffic_src, traffic_stale=stale, kernels=table)
    # the contract's probe figure: 12 B per index look-up of kfilter, nothing else, over the probe stage's time; the same stage
    # priced with the classify / count work the fused kernel also does is roofline.kernels["k_probe"] in the detail file
    probe_roof = roofline_of("k_probe: look-ups only", table) if "k_probe" in table else None
    if probe_roof:
        probe_roof["kernel"] = "k_probe"

    sustained = None
    if solo and args.sustain_seconds > 0:
        ctx.timers_enable(False)
        nsus = max(args.steps, int(np.ceil(args.sustain_seconds / max(dt / args.steps, 1e-6))))
        dts = time_steps(ctx, step, nsus, 0)
        sustained = dict(value=2 * npairs * nsus / dts, unit="reads/s", steps=nsus, wall_s=dts, ms_per_step=dts / nsus * 1e3,
                         note="the same step back to back for >= --sustain-seconds of wall time, no per-kernel events")
        log(f"sustained: {nsus} steps in {dts:.2f}s = {sustained['value'] / 1e9:.2f} G reads/s")
    two_lanes = None
    if solo and args.lanes == 1 and args.extra_lanes:
        os.environ["DBTK_LANES"] = "2"
        ctx2 = dbtk.context(g, params, device=local_rank)
        os.environ["DBTK_LANES"] = "1"
        ctx2.timers_enable(False)
        dt2 = time_steps(ctx2, lambda: ctx2.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, rlen), args.steps, args.warmup)
        two_lanes = dict(value=2 * npairs * args.steps / dt2, unit="reads/s", ms_per_step=dt2 / args.steps * 1e3,
                         note="extra, not `value`: the same steps on a context with the library's default DBTK_LANES=2 (successive batches alternate between two streams, one batch's encode kernel overlaps the other's probe and resolve kernels); `value` is the one-lane figure, whose per-kernel times add up to the step")
        ctx2.close()

    # ---- N = 1 extras: further read mixes, end to end, CPU baselines
    mixes, e2e, cpu, parity, k25_check = None, None, None, None, None
    k25_files = 0  # pairs of the k = 25 all-hit FASTA written for the command line's config-4 leg
    dense_checks = []  # per mix: the GPU half of its dense-slice parity check (the oracle half runs with the CPU legs)
    nhit = 0  # pairs of the all-hit FASTA the CLI walk legs read
    if solo and rank == 0:
        mixes = {}
        only = args.only_mix
        if do_mixes:
            mp = args.mix_reads // 2
            # all-hit: every pair tiled from a locus (SURVEY 8d mix 1): the probe and resolve kernels carry the step
            ah_seq, ah_off = syn.reads(mp, rlen=rlen, hit_frac=1.0, seed=2, nthreads=nth)
            if ref_dir and do_walk:
                nhit = mp
                syn.write_fasta(ah_seq, nhit, os.path.join(ref_dir, "reads_hit.fa"), rlen=rlen)
            d_ah = torch.from_numpy(ah_seq).to(dev)
            d_aho = torch.from_numpy(ah_off.view(np.int64)).to(dev)
            torch.cuda.synchronize()
            if only in (None, "all_hit"):
                # (its own context, as every other mix: a context that has just run the headline batch would take the global-table
                # probe path for its first all-hit batch — the path is chosen from the batch before)
                ctxa = dbtk.context(g, params, device=local_rank)
                ctxa.timers_enable(1)
                dta = time_steps(ctxa, lambda: ctxa.align_device(d_ah.data_ptr(), d_aho.data_ptr(), mp, rlen), args.mix_steps, 2)
                psa = ctxa.path_stats()
                ta = kernel_table(ctxa.kernel_times(), algorithmic_bytes(abi, ctxa.counters().astype(np.float64), ps=psa), args.mix_steps, args.mix_steps)
                if args.cpu_seconds > 0:
                    dense_checks.append(dense_gpu_half(dbtk, "all_hit", syn, arrs, g, params, 21))
                ctxa.close()
                doma = max((k for k in ta if ":" not in k), key=lambda k: ta[k]["avg_ms"] * ta[k]["launches"])
                mixes["all_hit"] = dict(workload=f"{2 * mp} reads per step, 100 % of pairs from loci, -k 21 -kf 4 1 -cth 45 -ka",
                                        value=2 * mp * args.mix_steps / dta, unit="reads/s", ms_per_step=dta / args.mix_steps * 1e3,
                                        steps=args.mix_steps, roofline=dict(roofline_of(doma, ta), kernels=ta),
                                        path=dict(psa, note="pairs per step x steps + warm-up: which kernels took them (dbtk_ctx_path_stats); fused_done = pairs the "
                                                            "locus-resident probe kernel resolved itself (no hit rows, no second kernel)"),
                                        probe_lookups_roofline=dict(roofline_of("k_probe: look-ups only", ta), kernel="k_probe",
                                                                    formula="12 P / stage time (SURVEY 8d: the contract's probe figure)") if "k_probe" in ta else None,
                                        probe_roofline=dict(roofline_of("k_probe", ta), profiled=pmc_mix("all_hit", "k_probe"),
                                                            traffic=(pmc_mix("all_hit", "k_probe") or {}).get("traffic"))
                                        if "k_probe" in ta else None)
                log(f"all-hit mix: {dta / args.mix_steps * 1e3:.3f} ms/step, {mixes['all_hit']['value'] / 1e9:.2f} G reads/s, dominant {doma} "
                    f"{ta[doma]['avg_ms']:.3f} ms = {ta[doma]['gbs']:.0f} GB/s algorithmic")
            if do_walk and only in (None, "walk"):
                # config 4's path: every assigned pair walked through its locus' graph with error correction (-gc 85 3), exact counting
                pw = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0, threading=abi.THREADING_V13, thread_cth=85,
                                        correction=1, maxncorrection=3)
                t0 = time.time()
                ctxw = dbtk.context(g, pw, device=local_rank)
                tbw = ctxw.table_bytes()
                log(f"walk context (graph table in HBM): {time.time() - t0:.1f}s; tables of a walking context: {tbw.get('total', 0) / 1e9:.1f} GB")
                ctxw.timers_enable(1)
                dtw = time_steps(ctxw, lambda: ctxw.align_device(d_ah.data_ptr(), d_aho.data_ptr(), mp, rlen), args.mix_steps, 2)
                cw = ctxw.counters().astype(np.float64)
                walked_kmers = cw[abi.C_THREADING] * (rlen - 21 + 1)  # k-mers of the reads that entered threading
                psw = ctxw.path_stats()
                tw = kernel_table(ctxw.kernel_times(), algorithmic_bytes(abi, cw, walked_kmers, ps=psw, walk=True), args.mix_steps, args.mix_steps)
                domw = max((k for k in tw if ":" not in k), key=lambda k: tw[k]["avg_ms"] * tw[k]["launches"])
                mixes["walk_gc85_3"] = dict(workload=f"{2 * mp} reads per step, 100 % of pairs from loci, --v13-threading -gc 85 3 -k 21 -kf 4 1 -cth 45 -ka",
                                            value=2 * mp * args.mix_steps / dtw, unit="reads/s", ms_per_step=dtw / args.mix_steps * 1e3,
                                            steps=args.mix_steps, reads_walked_per_step=cw[abi.C_THREADING] / args.mix_steps,
                                            reads_feasible_per_step=cw[abi.C_FEASIBLE] / args.mix_steps, hbm_bytes_tables=tbw, path=psw,
                                            roofline=dict(roofline_of(domw, tw), kernels=tw, profiled=pmc_mix("walk", "k_walk_pairs"),
                                                          traffic=(pmc_mix("walk", "k_walk_pairs") or {}).get("traffic")))
                log(f"walk mix: {dtw / args.mix_steps * 1e3:.3f} ms/step, {mixes['walk_gc85_3']['value'] / 1e6:.1f} M reads/s, dominant {domw} {tw[domw]['avg_ms']:.3f} ms")
                if args.cpu_seconds > 0:
                    dense_checks.append(dense_gpu_half(dbtk, "walk_gc85_3", syn, arrs, g, pw, 21))  # (before the walking context goes: they share the graph tables)
                ctxw.close()
            # genome-like background: WGS reads are not uniform random — repeat families shared with the flanks let far more pairs
            # through subfilter than the 2 % that come from a locus.  Here 15 % of the background pairs carry a 64-base stretch of
            # some locus (cut from an all-hit pair) over one of the sampled windows of each mate: they pass subfilter, reach the probe
            # kernel and die in kfilter.  Counts are not oracle-checked here (the parity run above uses the headline batch).
            if only in (None, "genome"):
                rng = np.random.default_rng(7)
                gseq = seq[:2 * mp * rlen].copy()
                pick = np.nonzero(rng.random(mp) < 0.15)[0]
                src = rng.integers(0, mp, len(pick))
                at = rng.choice(np.array([0, (rlen - 21 + 1) // 3, 2 * ((rlen - 21 + 1) // 3), rlen - 64]), len(pick))
                at = np.minimum(at, rlen - 64)
                at2 = np.minimum(rng.choice(np.array([0, (rlen - 21 + 1) // 3, 2 * ((rlen - 21 + 1) // 3), rlen - 64]), len(pick)), rlen - 64)
                for q in range(64):  # (subfilter wants a hit in BOTH mates, AQ.cpp:172-188: a stretch in each)
                    gseq[2 * pick * rlen + at + q] = ah_seq[2 * src * rlen + 40 + q]
                    gseq[(2 * pick + 1) * rlen + at2 + q] = ah_seq[(2 * src + 1) * rlen + 40 + q]
                d_g = torch.from_numpy(gseq).to(dev)
                d_go = torch.from_numpy(ah_off.view(np.int64)).to(dev)
                ctx.timers_enable(1)
                dtg = time_steps(ctx, lambda: ctx.align_device(d_g.data_ptr(), d_go.data_ptr(), mp, rlen), args.mix_steps, 2)
                cg = ctx.counters().astype(np.float64)
                psg = ctx.path_stats()
                tg = kernel_table(ctx.kernel_times(), algorithmic_bytes(abi, cg, ps=psg), args.mix_steps, args.mix_steps)
                domg = max((k for k in tg if ":" not in k), key=lambda k: tg[k]["avg_ms"] * tg[k]["launches"])
                mixes["genome_like"] = dict(workload=f"{2 * mp} reads per step: the headline mix ({args.hit_frac:.0%} of pairs from loci) with 15 % of the background "
                                                     f"pairs carrying, in each mate, a 64-base repeat shared with a locus over a sampled window, -k 21 -kf 4 1 -cth 45 -ka",
                                            value=2 * mp * args.mix_steps / dtg, unit="reads/s", ms_per_step=dtg / args.mix_steps * 1e3, steps=args.mix_steps,
                                            pairs_past_subfilter=cg[abi.C_SURVIVORS] / args.mix_steps / mp, path=psg,
                                            counters={n: int(cg[getattr(abi, "C_" + n)]) for n in ("SURVIVORS", "KMERFILTERED", "LOCUSFILTERED", "ASGN", "NHASH1")},
                                            roofline=dict(roofline_of(domg, tg), kernels=tg))
                log(f"genome-like mix: {dtg / args.mix_steps * 1e3:.3f} ms/step, {mixes['genome_like']['value'] / 1e9:.2f} G reads/s, "
                    f"{100 * mixes['genome_like']['pairs_past_subfilter']:.1f} % of pairs past subfilter, dominant {domg}")
                del d_g, d_go
            del d_ah, d_aho
            if do_walk and args.k25_reads > 0 and only in (None, "k25"):
                # BASELINE config 4 as stated: k = 25 (pipeline/k25.json:5), -gc 85 3, every assigned pair walked, exact counting
                t0 = time.time()
                syn25 = pkg.Synth(nloci=args.nloci, k=25, flank=700, seed=20250808, nthreads=nth)
                syn25.graph(nth)
                arrs25 = syn25.arrays()
                h25 = C.c_void_p()
                dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs25), C.byref(h25)))
                g25 = pkg.Rpgg(dbtk, h25)
                p25 = abi.default_params(ksize=25, n_filter=4, nm_filter=1, cthreshold=45, okam=0, threading=abi.THREADING_V13, thread_cth=85,
                                         correction=1, maxncorrection=3)
                ctx25 = dbtk.context(g25, p25, device=local_rank)
                kp = args.k25_reads // 2
                s25, o25 = syn25.reads(kp, rlen=rlen, hit_frac=1.0, seed=3, nthreads=nth)
                d_s25 = torch.from_numpy(s25).to(dev)
                d_o25 = torch.from_numpy(o25.view(np.int64)).to(dev)
                torch.cuda.synchronize()
                log(f"k = 25 RPGG ({arrs25.nkeys} index keys) + graph + HBM tables + {2 * kp} all-hit reads: {time.time() - t0:.1f}s")
                ctx25.timers_enable(1)
                dt25 = time_steps(ctx25, lambda: ctx25.align_device(d_s25.data_ptr(), d_o25.data_ptr(), kp, rlen), args.mix_steps, 2)
                c25 = ctx25.counters().astype(np.float64)
                ps25 = ctx25.path_stats()
                t25 = kernel_table(ctx25.kernel_times(), algorithmic_bytes(abi, c25, c25[abi.C_THREADING] * (rlen - 25 + 1), ps=ps25, walk=True), args.mix_steps, args.mix_steps)
                dom25 = max((k for k in t25 if ":" not in k), key=lambda k: t25[k]["avg_ms"] * t25[k]["launches"])
                mixes["walk_k25_gc85_3"] = dict(workload=f"synthetic release-scale RPGG at k = 25 ({args.nloci} loci, {arrs25.nkeys} index keys), {2 * kp} reads per step, "
                                                         f"100 % of pairs from loci, --v13-threading -gc 85 3 -k 25 -kf 4 1 -cth 45 -ka",
                                                value=2 * kp * args.mix_steps / dt25, unit="reads/s", ms_per_step=dt25 / args.mix_steps * 1e3, steps=args.mix_steps,
                                                reads_walked_per_step=c25[abi.C_THREADING] / args.mix_steps, reads_feasible_per_step=c25[abi.C_FEASIBLE] / args.mix_steps, path=ps25,
                                                roofline=dict(roofline_of(dom25, t25), kernels=t25, profiled=pmc_mix("k25", "k_walk_pairs"),
                                                              traffic=(pmc_mix("k25", "k_walk_pairs") or {}).get("traffic")), parity=None)
                log(f"k = 25 walk mix: {dt25 / args.mix_steps * 1e3:.3f} ms/step, {mixes['walk_k25_gc85_3']['value'] / 1e6:.1f} M reads/s, dominant {dom25} {t25[dom25]['avg_ms']:.3f} ms")
                if ref_dir and not args.no_e2e:
                    # config 4 through the drop-in command line too: the k = 25 RPGG as the files the CLI loads, its reads as FASTA
                    t0 = time.time()
                    syn25.write_files(os.path.join(ref_dir, "pan25"))
                    syn25.write_fasta(s25, kp, os.path.join(ref_dir, "reads_hit25.fa"), rlen=rlen)
                    k25_files = kp
                    log(f"k = 25 RPGG files + {2 * kp} reads as FASTA: {time.time() - t0:.1f}s")
                del d_s25, d_o25
                k25_check = None
                if args.cpu_seconds > 0 and args.k25_parity_pairs > 0:
                    # the GPU half of the parity slice now, the oracle half with the CPU legs below: the k = 25 tables (80 GB with the walk's)
                    # do not stay in HBM while the command-line legs bring their own
                    k25_check = (syn25, arrs25, dense_gpu_half(dbtk, "walk_k25_gc85_3", syn25, arrs25, g25, p25, 25))
                    ctx25.close()
                else:
                    syn25.close()
                    ctx25.close()
                g25.close()
        if not args.no_e2e:
            # the same batch handed over as HOST buffers: validation + PCIe copies + kernels, one batch after the other
            e2e = {}
            ctx.timers_enable(0)
            ctx.reset()
            ctx.align(seq, off)
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.align(seq, off)
            th = (time.perf_counter() - t0) / 3
            e2e["host_buffers"] = dict(value=2 * npairs / th, unit="reads/s", ms_per_batch=th * 1e3,
                                       note="dbtk_align_batch: pageable host buffers -> H2D -> kernels, per 10 M-read batch")
            log(f"host-buffer batches: {th * 1e3:.1f} ms per batch, {2 * npairs / th / 1e6:.0f} M reads/s")
            cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
            if ref_dir and os.path.exists(cli):
                t0 = time.perf_counter()
                r = subprocess.run([cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "-ka", "-fa", "reads.fa", "-qs", "pan", "-o", "cli"], cwd=ref_dir,
                                   capture_output=True, text=True, env=dict(os.environ, DBTK_VERBOSE="1"))
                tcli = time.perf_counter() - t0
                ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
                tline = [l for l in r.stderr.splitlines() if l.startswith("timeline:")]
                e2e["cli"] = dict(wall_s=tcli, returncode=r.returncode, reads=2 * nref, batch_loop=ing[0] if ing else None,
                                  timeline=tline[0] if tline else None,
                                  main_s=float(tline[0].rsplit("done at", 1)[1]) if tline else None,
                                  load=" | ".join(l for l in r.stderr.splitlines() if l.startswith("load:") or re.match(r"tables: index [0-9]", l)) or None,
                                  note="this repo's danbing-tk on the reference leg's files: RPGG load + HBM tables + parse + pair + align + dump; wall_s = the whole process by the caller's clock, main_s = main() until every output is written and closed (timeline)")
                log(f"CLI end to end: {tcli:.2f}s wall ({e2e['cli']['main_s']} s from main() to the outputs written: the rest is the dynamic loader and the driver reclaiming 28 GB at exit); {e2e['cli']['load']}; {ing[0] if ing else ''}")
                # the batch loop on a file large enough for its steady state: the reader on the device (default for a regular file: the
                # host only copies bytes, kernels find the records and pair the mates) and on the host (--host-ingest), same binary
                need_bytes = args.ingest_reads * (rlen + 24)
                if args.ingest_reads > 0 and shutil.disk_usage(ref_dir).free > 3 * need_bytes:  # (never fill the box's memory-backed scratch)
                    big = os.path.join(ref_dir, "reads_big.fa")
                    t0 = time.perf_counter()
                    with open(big, "wb") as out:
                        part = 4_000_000
                        for p0 in range(0, args.ingest_reads // 2, part):
                            n = min(part, args.ingest_reads // 2 - p0)
                            sq, _ = syn.reads(n, rlen=rlen, hit_frac=args.hit_frac, seed=1, first_pair=p0, nthreads=nth)
                            syn.write_fasta(sq, n, os.path.join(ref_dir, "part.fa"), rlen=rlen, first_pair=p0)
                            with open(os.path.join(ref_dir, "part.fa"), "rb") as f:
                                shutil.copyfileobj(f, out, 64 << 20)
                            del sq
                    os.unlink(os.path.join(ref_dir, "part.fa"))
                    tgen = time.perf_counter() - t0
                    legs = {}
                    for name, extra in (("device_reader", []), ("host_reader", ["--host-ingest"]), ("device_reader_again", [])):
                        r = subprocess.run([cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "-ka", "-fa", "reads_big.fa", "-qs", "pan", "-o", "big_" + name] + extra,
                                           cwd=ref_dir, capture_output=True, text=True)
                        ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
                        rate = float(ing[0].split("(")[1].split()[0]) * 1e6 if ing and r.returncode == 0 else None
                        legs[name] = dict(returncode=r.returncode, value=rate, unit="reads/s", batch_loop=ing[0] if ing else None,
                                          detail=([l for l in r.stderr.splitlines() if l.startswith("device reader:")] or [None])[0])
                    try:
                        same = all(open(os.path.join(ref_dir, "big_device_reader" + e), "rb").read() == open(os.path.join(ref_dir, "big_host_reader" + e), "rb").read()
                                   for e in (".trkmc.ar", ".tr.summary.txt"))
                    except OSError:  # (a leg failed: its returncode says so)
                        same = False
                    e2e["cli_ingest"] = dict(legs, reads=args.ingest_reads, fasta_bytes=os.path.getsize(big), same_outputs=same,
                                             note="batch loop of this repo's danbing-tk (-ka) on an interleaved FASTA in /dev/shm, first byte read to last kernel done: "
                                                  "reader on the device vs reader on the host; the host reader is bound by the container's 16-CPU quota")
                    os.unlink(big)
                    log(f"CLI batch loop on {args.ingest_reads / 1e6:.0f} M reads ({e2e['cli_ingest']['fasta_bytes'] / 1e9:.1f} GB, written in {tgen:.0f}s): device reader "
                        f"{(legs['device_reader']['value'] or 0) / 1e6:.0f} / {(legs['device_reader_again']['value'] or 0) / 1e6:.0f} M reads/s, host reader "
                        f"{(legs['host_reader']['value'] or 0) / 1e6:.0f} M reads/s, same outputs: {same}")
                if do_walk and nhit and os.path.exists(os.path.join(ref_dir, "pan.graph.umap")):
                    # config 5: the walk with and without the -ae emit (records formatted + deflated on host threads while the GPU runs
                    # on), on an all-hit read file so that every pair is walked and printed
                    legs = {}
                    base = [cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "cliw"]
                    # ("walk_again": a second pass over the same file — the first pass over a file another process has just written into
                    # tmpfs is bound by the host's first touch of its pages)
                    for name, extra in (("walk", []), ("walk_again", []), ("walk_ae_gz", ["-ae", "--aln-gz", "cliw.aln.gz"]),
                                        ("walk_ae_gz_host_zlib1", ["-ae", "--aln-gz", "cliwh.aln.gz", "--host-ingest"]),
                                        ("walk_ae_gz_level6", ["-ae", "--aln-gz", "cliw6.aln.gz", "--gz-level", "6"])):
                        t0 = time.perf_counter()
                        r = subprocess.run(base + extra, cwd=ref_dir, capture_output=True, text=True)
                        tw = time.perf_counter() - t0
                        ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
                        em = [l for l in r.stderr.splitlines() if l.startswith("emit:")]
                        legs[name] = dict(wall_s=tw, returncode=r.returncode, batch_loop=ing[0] if ing else None, emit=em[0] if em else None)
                    if k25_files and os.path.exists(os.path.join(ref_dir, "pan25.graph.umap")):
                        # BASELINE config 4 through the command line: k = 25, -gc 85 3, every assigned pair walked
                        b25 = [cli, "-k", "25", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit25.fa", "-qs", "pan25", "-o", "cliw25"]
                        for name in ("walk_k25", "walk_k25_again"):
                            t0 = time.perf_counter()
                            r = subprocess.run(b25, cwd=ref_dir, capture_output=True, text=True)
                            tw = time.perf_counter() - t0
                            ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
                            legs[name] = dict(wall_s=tw, returncode=r.returncode, batch_loop=ing[0] if ing else None, reads=2 * k25_files,
                                              load=([l for l in r.stderr.splitlines() if l.startswith("load:")] or [None])[0])
                        log(f"CLI walk at k = 25 (config 4 through the command line): {legs['walk_k25']['batch_loop']}")
                        for fn in ("reads_hit25.fa",):
                            os.unlink(os.path.join(ref_dir, fn))
                    gz = os.path.join(ref_dir, "cliw.aln.gz")
                    legs["aln_gz_bytes"] = os.path.getsize(gz) if os.path.exists(gz) else None
                    legs["reads"] = 2 * nhit
                    gzh = os.path.join(ref_dir, "cliwh.aln.gz")
                    legs["aln_gz_zlib1_bytes"] = os.path.getsize(gzh) if os.path.exists(gzh) else None
                    gz6 = os.path.join(ref_dir, "cliw6.aln.gz")
                    legs["aln_gz_level6_bytes"] = os.path.getsize(gz6) if os.path.exists(gz6) else None
                    legs["note"] = ("this repo's danbing-tk --v13-threading -gc 85 3 on an all-hit FASTA, without and with -ae --aln-gz: wall seconds "
                                    "including the RPGG + graph load, and the batch loop's own line.  walk_ae_gz: the lines are assembled and gzip-compressed "
                                    "on the GPU (Huffman-only deflate, dbtk_gz.h); walk_ae_gz_host_zlib1 / _level6: zlib on the host's emit pool (level 1, "
                                    "level 6 = gzip's default), bound by deflate on a 16-CPU container.  With -ae every mate of every pair is aligned (the "
                                    "record holds both alignments); without it a pair is decided by its first cleanly threading mate")
                    # a LONG all-hit input (4 x the read set above, > 2 GB: the command line then merges its parsed 32-MB blocks on the device
                    # into batches of ~2 M pairs, which is what lets the locus-resident kernels take them: VERDICT r4 item 5) against the
                    # same run with the blocks aligned one by one
                    need = args.walk_long_reads * (rlen + 24)
                    if args.walk_long_reads > 0 and shutil.disk_usage(ref_dir).free > 3 * need:
                        longfa = os.path.join(ref_dir, "reads_hit_long.fa")
                        with open(longfa, "wb") as out:
                            part = 4_000_000
                            for p0 in range(0, args.walk_long_reads // 2, part):
                                n = min(part, args.walk_long_reads // 2 - p0)
                                sq, _ = syn.reads(n, rlen=rlen, hit_frac=1.0, seed=2, first_pair=p0, nthreads=nth)
                                syn.write_fasta(sq, n, os.path.join(ref_dir, "part.fa"), rlen=rlen, first_pair=p0)
                                with open(os.path.join(ref_dir, "part.fa"), "rb") as f:
                                    shutil.copyfileobj(f, out, 64 << 20)
                                del sq
                        os.unlink(os.path.join(ref_dir, "part.fa"))
                        lbase = [a_ if a_ != "reads_hit.fa" else "reads_hit_long.fa" for a_ in base]
                        long_legs = {}
                        for name, env in (("merged", {}), ("blocks_one_by_one", {"DBTK_NO_MERGE": "1"}), ("merged_again", {})):
                            r = subprocess.run(lbase[:-1] + ["cliwl_" + name], cwd=ref_dir, capture_output=True, text=True, env=dict(os.environ, **env))
                            ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
                            rate = float(ing[0].split("(")[1].split()[0]) * 1e6 if ing and r.returncode == 0 else None
                            long_legs[name] = dict(returncode=r.returncode, value=rate, unit="reads/s", batch_loop=ing[0] if ing else None,
                                                   detail=([l for l in r.stderr.splitlines() if l.startswith("device reader:")] or [None])[0])
                        try:
                            same = all(open(os.path.join(ref_dir, "cliwl_merged" + e), "rb").read() == open(os.path.join(ref_dir, "cliwl_blocks_one_by_one" + e), "rb").read()
                                       for e in (".trkmc.ar", ".tr.summary.txt"))
                        except OSError:
                            same = False
                        legs["walk_long"] = dict(long_legs, reads=args.walk_long_reads, same_outputs=same,
                                                 note="--v13-threading -gc 85 3 (no -ae) on an all-hit FASTA of this many reads in /dev/shm: the batch loop with the "
                                                      "parsed blocks merged on the device into ~2 M-pair batches (the default for inputs of 2 GB and more) and with every "
                                                      "32-MB block its own batch")
                        os.unlink(longfa)
                        log(f"CLI walk on {args.walk_long_reads / 1e6:.0f} M all-hit reads: merged batches {(long_legs['merged']['value'] or 0) / 1e6:.0f} / "
                            f"{(long_legs['merged_again']['value'] or 0) / 1e6:.0f} M reads/s, blocks one by one {(long_legs['blocks_one_by_one']['value'] or 0) / 1e6:.0f} M reads/s, same outputs: {same}")
                    e2e["cli_walk_emit"] = legs
                    log(f"CLI walk: {legs['walk']['wall_s']:.1f}s; with -ae --aln-gz: {legs['walk_ae_gz']['wall_s']:.1f}s "
                        f"({legs['aln_gz_bytes']} bytes of .aln.gz for {2 * nhit} reads)")

        # ---- cpu baselines
        port = None
        if do_ref:
            start_small_ref_legs()
        if args.cpu_seconds > 0:
            import bind
            try:
                orc = bind.Oracle()
            except OSError:
                subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
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
            port = dict(value=2 * done / t_cpu, unit="reads/s", cores=1, kind="port",
                        sample=f"first {2 * done} reads of the same read set and RPGG, oracle/dbtk_oracle.c on 1 host core, {t_cpu:.1f} s", checked=parity)
            log(f"oracle (port) on 1 core: {port['value']:.0f} reads/s")
            # every mix against the oracle on a DENSE slice (64 pairs on each of a few hundred loci of every image class, 6 % of them
            # chimeric / foreign): the regime in which the locus-resident kernels — the ones the mixes time — take the pairs
            for chk in dense_checks:
                mixes[chk["mix"]]["parity"] = dense_oracle_half(orc, go, chk, log)
            orc.free(go)
            if k25_check:
                syn25, arrs25, chk = k25_check
                t0 = time.time()
                go25 = orc.from_arrays(arrs25)
                mixes["walk_k25_gc85_3"]["parity"] = dense_oracle_half(orc, go25, chk, log)
                log(f"(k = 25 oracle tables + slice: {time.time() - t0:.0f}s)")
                orc.free(go25)
                syn25.close()
        if do_ref:
            for th in ref_procs:
                th.join()
            pvals = args.ref_threads if args.ref_threads else [1, 8, ncpu]
            for pv in [p for p in pvals if p > 8]:
                ref_results.append(dict(ref_baseline.run_reference(os.path.join(ROOT, "oracle", "_ref", "danbing-tk"), ref_dir, "reads.fa", pv), fasta="reads.fa"))
            runs = []
            for r in sorted(ref_results, key=lambda r: r["p"]):
                if r.get("query_s") and r.get("reads"):
                    runs.append(dict(threads=r["p"], reads=r["reads"], query_s=r["query_s"], load_s=r["load_s"], value=r["reads"] / r["query_s"], unit="reads/s"))
                    log(f"reference binary -p {r['p']}: {r['reads']} reads in {r['query_s']:.1f}s = {r['reads'] / r['query_s'] / 1e3:.0f} k reads/s (load {r['load_s']:.0f}s)")
            if runs:
                best = max(runs, key=lambda r: r["value"])
                model = ""
                try:
                    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
                except Exception:
                    pass
                cpu = dict(value=best["value"], unit="reads/s", cores=best["threads"], usable_cpus=usable_cpus(), kind="reference",
                           sample=f"oracle/_ref/danbing-tk (the reference compiled from /root/reference) -k 21 -kf 4 1 -cth 45 -ka -p {best['threads']} on the same "
                                  f"RPGG written as its HEAD files and the first {best['reads']} reads of the same read set as 2-line FASTA; timed from its "
                                  f"'threads created' line to 'parallel query completed' ({best['query_s']:.1f} s); best of the -p values in `runs`",
                           host=f"{ncpu} hardware threads ({usable_cpus()} usable: CPU quota of the container), {model}", runs=runs, port=port)
        if cpu is None and port is not None:
            cpu = port
        if ref_dir:
            shutil.rmtree(ref_dir, ignore_errors=True)

    if rank == 0:
        out = {
            "metric": "paired reads/sec aligned to RPGG (k=21)", "value": value, "unit": "reads/s", "n_gpus": world, "ranks_seen": ranks_seen, "per_rank_ms_per_step": per_rank_ms,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"release-scale synthetic RPGG ({args.nloci} loci, {arrs.nkeys} index keys, {g.ntrkmers} TR k-mers) "
                                   f"replicated per GPU; {args.reads} x 150bp PE reads per GPU per step, {args.hit_frac:.0%} of pairs from loci; "
                                   f"-k 21 -kf 4 1 -cth 45 -ka; RCCL all-reduce of counts at the end",
                       "lanes": args.lanes, "reads_per_gpu": args.reads, "read_len": rlen, "hit_frac": args.hit_frac, "k": 21, "cth": 45,
                       "kernel_source_hash": kernel_source_hash()},
            "roofline": roof, "probe_roofline": probe_roof, "hbm_bytes_tables": hbm_tables, "reduce_check": reduce_check, "sustained": sustained, "mixes": mixes, "end_to_end": e2e, "cpu_baseline": cpu, "parity": parity,
            "two_lanes": two_lanes, "bench_wall_s": time.time() - t_start,
        }
        line = compact_line(out)
        detail = os.environ.get("DBTK_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail, "w") as fh:
                json.dump(out, fh)
            log(f"detail (mixes, per-kernel tables, end-to-end legs, reference runs): {detail}")
        except OSError as e:
            log(f"could not write {detail}: {e}")
        assert len(line) < MAX_LINE, f"bench line is {len(line)} characters: the driver reads the last 8 000 of stdout"
        print(line, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
