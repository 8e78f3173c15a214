"""The kernel bodies (danbing-tk_amd/csrc/dbtk_kernels.h — the code hipcc
compiles for gfx950) run on the test-only SPMD emulator and compared with the
oracle.  This is how the device logic is exercised where no GPU exists; the GPU
suite (test_gpu_parity.py) repeats the same cases on hardware."""
import numpy as np
import pytest

import bind
import cases
from test_oracle import GOLD, check_against_golden, golden_inputs

abi = bind.abi


@pytest.fixture(scope="module")
def O():
    return bind.Oracle()


@pytest.fixture(scope="module")
def E():
    return bind.Emu()


@pytest.mark.parametrize("case", sorted(cases.CASES))
def test_kernel_bodies_match_oracle(case, O, E, tmp_path):
    c = cases.make_case(case, str(tmp_path))
    go = O.load(c.prefix, c.k, c.qc_file)
    g = E.load(c.prefix, c.k, c.qc_file)
    T = E.tables(g)
    order = g.output_order().astype(np.int64)
    seq, off = c.reads.packed()
    # the load-time check: index memberships == flank/TR sets  <=>  the class rides in the index slot
    assert E.consistent(T) == (0 if case == "inconsistent" else 1)
    for i, kw in enumerate(c.param_sets):
        p = abi.default_params(ksize=c.k, trace=1, **kw)
        a = O.align(go, p, seq, off)
        co = np.zeros(g.ntrkmers, np.uint64)
        np.add.at(co, order, a["counts_file"])
        for mode in ((1, 0) if E.consistent(T) else (0,)):  # fused-aux path and the general class-table path
            E.set_consistent(T, mode)
            b = E.align(g, T, p, seq, off, grid_k1=1 + i, grid_pair=2 + 3 * i)
            assert (co == b["counts"]).all()
            assert (a["kmc"] == b["kmc"]).all() and (a["nmapread"] == b["nmapread"]).all()
            assert (a["counters"] == b["counters"]).all(), (a["counters"], b["counters"])
            d = bind.recs_equal(a["recs"], b["recs"], c.reads.npairs)
            assert d < 0, f"{bind.rec_str(a['recs'][d])}\n{bind.rec_str(b['recs'][d])}"
        E.set_consistent(T, 1 if case != "inconsistent" else 0)
    E.L.emu_tables_free(T)
    O.free(go)
    g.close()


def test_device_sort_is_gcc_std_sort(E):
    """dbtk_sort.h (index form and packed form) against this host's std::sort, incl. the heapsort fallback."""
    assert E.selftest_sort(3, 20000) == 0


def test_assign_bits_equals_literal_scan(E):
    """The mask form of assignTRkmc used by the kernels vs the literal restatement of AQ.cpp:1470-1555."""
    assert E.selftest_assign(7, 400000) == 0


@pytest.mark.parametrize("name", sorted(GOLD))
def test_kernel_bodies_reproduce_reference_binary(name, E):
    d, p, reads, qc = golden_inputs(name)
    import os
    g = E.load(os.path.join(d, "pan"), p.ksize, qc)
    T = E.tables(g)
    seq, off = reads.packed()
    p.trace = 1
    b = E.align(g, T, p, seq, off)
    check_against_golden(d, b["counts"], b["kmc"], b["nmapread"], b["counters"], b["recs"], reads)
    E.L.emu_tables_free(T)
    g.close()
