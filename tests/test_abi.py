"""CPU-side checks of the product library: it loads, exports every symbol
include/dbtk.h declares, its host half (RPGG loaders, output order, dumps)
matches the reference's files, and it refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bind
import cases
from test_oracle import GOLD, golden_inputs

abi = bind.abi
pkg = bind.pkg


@pytest.fixture(scope="module")
def lib():
    return pkg.Dbtk()


def test_exports_match_header(lib):
    hdr = open(os.path.join(bind.ROOT, "include", "dbtk.h")).read()
    declared = {s for s in re.findall(r"\b(dbtk_[a-z_0-9]+)\s*\(", hdr) if not s.endswith("_t")}
    assert declared == set(pkg.EXPORTS), declared ^ set(pkg.EXPORTS)
    for s in declared:
        assert hasattr(lib.L, s), s
    assert lib.L.dbtk_abi_version() == abi.ABI_VERSION


def test_struct_layouts():
    assert C.sizeof(abi.Params) == 80
    assert C.sizeof(abi.MateRec) == 14 + 6 + 2 + 64
    assert C.sizeof(abi.PairRec) == 24 + 2 * C.sizeof(abi.MateRec)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_loader_output_order_and_dumps(name, lib, tmp_path):
    """dbtk_rpgg_load + dbtk_rpgg_output_order + dbtk_write_outputs against the
    reference's files: feeding the reference's own counts back through the dump
    must reproduce its .trkmc.ar / summary / -on bytes."""
    d, p, reads, qc = golden_inputs(name)
    g = lib.load(os.path.join(d, "pan"), p.ksize, qc)
    ar = np.fromfile(os.path.join(d, "ref.trkmc.ar"), np.uint64)
    assert g.ntrkmers == ar[0]
    summ = np.loadtxt(os.path.join(d, "ref.tr.summary.txt"), dtype=np.uint64, ndmin=2)
    out = str(tmp_path / "o")
    g.write_outputs(ar[1:].copy(), summ[:, 1].copy(), summ[:, 0].astype(np.uint32), out, with_names=False)
    assert open(out + ".trkmc.ar", "rb").read() == open(os.path.join(d, "ref.trkmc.ar"), "rb").read()
    assert open(out + ".tr.summary.txt", "rb").read() == open(os.path.join(d, "ref.tr.summary.txt"), "rb").read()
    on = [l.split() for l in open(os.path.join(d, "refon.tr.kmers")) if l[0] != ">"]
    cnt = np.array([int(x[1]) for x in on], np.uint64)
    g.write_outputs(cnt, None, None, out, with_names=True)
    assert open(out + ".tr.kmers", "rb").read() == open(os.path.join(d, "refon.tr.kmers"), "rb").read()
    g.close()


def test_load_errors(lib, tmp_path):
    with pytest.raises(pkg.DbtkError) as e:
        lib.load(str(tmp_path / "nope"), 21)
    assert e.value.status == abi.ERR_IO
    d, p, _, _ = golden_inputs("g1_k21")
    bad = tmp_path / "pan"
    for ext in ("tr.kmers", "kmers.dbi", "fl.kdb", "tre.kdb"):
        data = open(os.path.join(d, "pan." + ext), "rb").read()
        open(f"{bad}.{ext}", "wb").write(data[:len(data) // 2] if ext == "kmers.dbi" else data)
    with pytest.raises(pkg.DbtkError) as e:
        lib.load(str(bad), 21)
    assert e.value.status == abi.ERR_IO


def test_no_cpu_fallback(lib):
    """Without a HIP device the product refuses to create a context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d, p, _, _ = golden_inputs("g1_k21")
    g = lib.load(os.path.join(d, "pan"), 21)
    with pytest.raises(pkg.DbtkError) as e:
        lib.context(g, p)
    assert e.value.status == abi.ERR_NO_DEVICE
    g.close()
