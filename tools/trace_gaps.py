#!/usr/bin/env python3
"""Gaps between consecutive kernels of one stream in a rocprofv3 --kernel-trace CSV:  tools/trace_gaps.py DIR [name-filter]
(what a stage's HIP-event bracket holds beyond its kernels' own durations: launch gaps, scratch set-up, memsets)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_probe_locus"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.append((r["Kernel_Name"][:56], (e - s) / 1e3, gap, r.get("Scratch_Size", r.get("Private_Segment_Size", "?")), r.get("Queue_Id", "?")))
    prev_end = max(prev_end or 0, e)
idx = [i for i, o in enumerate(out) if flt in o[0]]
lo, hi = (max(0, idx[-9] - 6), min(len(out), idx[-1] + 8)) if len(idx) >= 9 else (0, min(len(out), 60))
for o in out[lo:hi]:
    print(f"{o[0]:56s} dur {o[1]:9.1f} us  gap-before {o[2]:8.1f} us  scratch {o[3]}  q {o[4]}")
