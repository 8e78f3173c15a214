#!/usr/bin/env python3
"""Diagnostic: print ms/step and per-kernel ms/step from a bench.py JSON line on stdin."""
import json
import sys
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"] / 1e9, 3), "G reads/s", round(d["ms_per_step"], 3), "ms/step",
              {k: round(v["avg_ms"] * v["launches"] / d["steps"], 3) for k, v in d["roofline"]["kernels"].items()}, d.get("parity"))
