"""Diagnosis of one seeded fuzz case (tests/test_gpu_fuzz.py) on the GPU box: how the
difference between the engine and the CPU oracle develops as rows are added one block at a
time -- a jump points at a defect, a smooth rise at the amplification of the reference's
pedestal recurrence (see assert_spectrum).  Test infrastructure: imports the oracle.

    python tests/diag_fuzz.py <seed> [level]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import oracle                          # noqa: E402
from pylbl_amd.engine import Engine                # noqa: E402
from tests import golden_io                        # noqa: E402
from tests.test_gpu_fuzz import make_case          # noqa: E402

seed = int(sys.argv[1])
case = make_case(seed)
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.port_library()
engine = Engine(0)
for name, value in case["options"].items():
    engine.set_option(name, value)
v0, vn, npv, cut = case["v0"], case["vn"], case["npv"], case["cut"]
t, p, x, source = case["t"][level], case["p"][level], case["x"][level], case["source"]
print({k: v for k, v in case.items() if k not in ("table", "source")})


def compare(rows, remove_pedestal=True):
    part = source.subset(np.arange(source.num_lines) < rows)
    k_ref, _ = oracle.absorption_port(part, t, p, x, v0, vn, npv, cut_off=cut,
                                      remove_pedestal=remove_pedestal)
    k_plain, _ = oracle.absorption_port(part, t, p, x, v0, vn, npv, cut_off=cut)
    handle = engine.load(part)
    got = engine.compute(handle, t, p, x, v0, vn, npv, cut_off=cut,
                         remove_pedestal=remove_pedestal)[0]
    engine.free(handle)
    tol = np.maximum(golden_io.pedestal_tolerance(k_ref, npv, cut, 1e-6), 1e-6*np.abs(k_plain))
    error = np.abs(got - k_ref)/(tol + 1e-300)
    growth = np.max(np.abs(k_ref))/max(np.max(np.abs(k_plain)), 1e-300)
    return float(np.max(error)), growth


print("without pedestal, all rows: worst error / (1e-6 k):", compare(source.num_lines, False)[0])
steps = sorted(set(int(r) for r in np.linspace(1, source.num_lines, 25)))
for rows in steps:
    worst, growth = compare(rows)
    print(f"rows {rows:5d}: worst error {worst:10.3e} x tolerance   |k| growth {growth:10.3e}")
