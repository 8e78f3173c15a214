"""The one JSON line bench.py prints must fit the driver's 8 000-character tail of stdout (the round-5 line had grown to
21 KB and was not parsed).  CPU test: the compaction on a committed detail record of a full run."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("dbtk_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_fits_and_keeps_the_contract():
    b = _bench()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))
    assert files
    for fn in files:
        full = json.load(open(fn))
        if "mixes" not in full or isinstance(full.get("detail"), str):
            continue  # (a compact line kept as a profile: nothing to compact)
        line = b.compact_line(full)
        assert len(line) < b.MAX_LINE < 8000 and "\n" not in line
        d = json.loads(line)
        for key in ("metric", "value", "unit", "n_gpus", "ranks_seen", "per_rank_ms_per_step", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
            assert key in d, (fn, key)
        assert d["config"]["workload"] and "model" not in d["config"]
        for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in d["roofline"], (fn, key)
        assert abs(d["value"] / full["value"] - 1) < 1e-3
        assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
        if full.get("cpu_baseline"):
            for key in ("value", "unit", "cores", "kind", "sample"):
                assert key in d["cpu_baseline"], (fn, key)


def test_compact_line_drops_extras_before_contract_fields():
    b = _bench()
    cands = [json.load(open(fn)) for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))]
    cands = [c for c in cands if isinstance(c.get("mixes"), dict) and c["mixes"]]
    assert cands
    full = cands[-1]
    one = next(iter(full["mixes"].values()))
    full["mixes"] = {f"mix{i}": dict(one) for i in range(80)}  # far too many to fit
    line = b.compact_line(full)
    assert len(line) < b.MAX_LINE
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and "mixes" not in d
