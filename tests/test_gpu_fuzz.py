"""Seeded random sweep of the whole argument space against the CPU oracle: grids, cut-offs,
pressures from near-vacuum to 50 atm, row orders, tile sizes, pedestal on/off, far-field
on/off, both line-scalar preparations."""
import os

import numpy as np
import pytest

from pylbl_amd import synthetic
from tests import golden_io
from tests.test_gpu_parity import assert_spectrum, oracle_conditioning

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from pylbl_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def make_case(seed):
    """The seeded random case as a dictionary (also used by tests/diag_fuzz.py)."""
    rng = np.random.default_rng(1000 + seed)
    v0 = int(rng.integers(1, 3000))
    span = int(rng.integers(1, 60))
    vn = v0 + span
    npv = int(rng.choice([1, 2, 7, 10, 33, 64, 100, 250, 1000]))
    if span*npv > 40000:
        span = max(40000//npv, 1)
        vn = v0 + span
    cut = int(rng.choice([25, 25, 25, 5, 60]))
    lo, hi = max(v0 - cut - 1., 0.05), vn + cut + 1.
    n_lines = int(rng.integers(1, 3000))
    table = synthetic.line_table(str(rng.choice(["H2O", "CO2", "O3", "CH4"])), lo, hi,
                                 num_lines=n_lines, seed=int(rng.integers(1 << 30)),
                                 tips_range=(100, 900))
    # Some lines exactly on grid points / integers, some duplicates, exaggerated shifts.
    k = max(n_lines//10, 1)
    table.nu[rng.choice(n_lines, k)] = np.round(table.nu[rng.choice(n_lines, k)])
    table.delta_air *= float(rng.choice([0., 1., 5.]))
    table.nu = np.clip(table.nu, lo, hi)
    order = np.argsort(table.nu, kind="stable")
    if rng.random() < 0.25:
        order = rng.permutation(n_lines)            # rows out of order (range rule matters)
    table = table.subset(order)
    levels = int(rng.integers(1, 4))
    case = dict(v0=v0, vn=vn, npv=npv, cut=cut, n_lines=n_lines, table=table, levels=levels,
                t=rng.uniform(150., 800., levels), p=10.**rng.uniform(-2., 6.7, levels),
                x=10.**rng.uniform(-7., -0.5, levels), ped=bool(rng.integers(0, 2)),
                policy="skip" if rng.random() < 0.3 else "reference")
    case["options"] = dict(farfield=int(rng.integers(0, 2)), prep=int(rng.random() < 0.2),
                           points_per_lane=int(rng.choice([0, 0, 1, 2, 4, 8])),
                           aligned_tiles=int(rng.random() < 0.3),
                           scan_chain=int(rng.random() < 0.8))
    source = table
    if case["policy"] == "skip":
        source = table.subset((table.nu >= v0 - (cut + 1)) & (table.nu <= vn + cut + 1))
    case["source"] = source         # what the reference sees under the same policy
    return case


# 64 cases in the regular run; PYLBL_FUZZ_CASES=2000 for a soak.
@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_CASES", "64"))))
def test_random_case(engine, oracle, seed):
    c = make_case(seed)
    v0, vn, npv, cut, ped, policy = c["v0"], c["vn"], c["npv"], c["cut"], c["ped"], c["policy"]
    t, p, x, table, source = c["t"], c["p"], c["x"], c["table"], c["source"]
    for name, value in c["options"].items():
        engine.set_option(name, value)
    molecule = engine.load(table)
    try:
        got = engine.compute(molecule, t, p, x, v0, vn, npv, cut_off=cut, remove_pedestal=ped,
                             range_policy=policy)
        for level in range(c["levels"]):
            k_ref, _ = oracle.absorption_port(source, t[level], p[level], x[level], v0, vn, npv,
                                              cut_off=cut, remove_pedestal=ped)
            k_plain, _ = oracle.absorption_port(source, t[level], p[level], x[level], v0, vn, npv,
                                                cut_off=cut)
            case = golden_io.Case("fuzz", seed, 0, 0, 0, v0, vn, npv, cut, ped, None, 0)
            assert_spectrum(got[level], k_ref, case,
                            f"seed {seed} level {level}: v0={v0} vn={vn} npv={npv} cut={cut} "
                            f"lines={c['n_lines']} ped={ped} policy={policy} p={p[level]:.3g}",
                            k_plain,
                            conditioning=lambda: oracle_conditioning(
                                oracle, source, t[level], p[level], x[level], v0, vn, npv, cut,
                                k_ref))
    finally:
        engine.free(molecule)
        for name, value in (("farfield", 0), ("prep", 0), ("points_per_lane", 0),
                            ("aligned_tiles", 0), ("scan_chain", 1)):
            engine.set_option(name, value)
