"""Seeded random cases at sizes where the work-item plan splits tiles, several levels share a
launch and the pedestal chain is long: 10^4..2x10^5 grid points, 10^3..10^5 lines (uniform or
banded), against the CPU oracle.  Two cases in the regular run (a few seconds of oracle time
each); PYLBL_FUZZ_LARGE=40 for a soak."""
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


@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_LARGE", "2"))))
def test_random_large_case(engine, oracle, seed):
    rng = np.random.default_rng(50_000 + seed)
    npv = int(rng.choice([100, 250, 500, 1000, 2000]))
    span = int(rng.integers(10, max(11, min(200, 200_000//npv))))
    v0 = int(rng.integers(1, 4000))
    vn = v0 + span
    cut = int(rng.choice([25, 25, 10]))
    n_lines = int(10**rng.uniform(3., 5.))
    # Keep the oracle's work (evals ~ lines in range x window points) near 2e9 at most.
    n_lines = int(min(n_lines, 2e9/((2*cut + 1)*npv)))
    formula = str(rng.choice(["H2O", "CO2", "O3", "CH4"]))
    lo, hi = max(v0 - cut - 1., 0.05), vn + cut + 1.
    if rng.random() < 0.5 and hi - lo > 220.:
        table = synthetic.banded_line_table(formula, lo, hi, num_lines=n_lines,
                                            bands=int(rng.integers(1, 6)),
                                            seed=int(rng.integers(1 << 30)))
    else:
        table = synthetic.line_table(formula, lo, hi, num_lines=n_lines,
                                     seed=int(rng.integers(1 << 30)))
    levels = int(rng.integers(1, 4))
    t = rng.uniform(180., 320., levels)
    p = 10.**rng.uniform(0., 5.1, levels)
    x = 10.**rng.uniform(-6., -1.5, levels)
    ped = bool(rng.integers(0, 2))
    engine.set_option("farfield", int(rng.integers(0, 2)))
    engine.set_option("points_per_lane", int(rng.choice([0, 0, 0, 2, 4, 8])))
    engine.set_option("scan_chain", int(rng.random() < 0.8))
    molecule = engine.load(table)
    try:
        got = engine.compute(molecule, t, p, x, v0, vn, npv, cut_off=cut, remove_pedestal=ped)
        for level in range(levels):
            k_ref, _ = oracle.absorption_port(table, t[level], p[level], x[level], v0, vn, npv,
                                              cut_off=cut, remove_pedestal=ped)
            k_plain = k_ref
            if ped:
                k_plain, _ = oracle.absorption_port(table, t[level], p[level], x[level], v0, vn,
                                                    npv, cut_off=cut)
            case = golden_io.Case("large", seed, 0, 0, 0, v0, vn, npv, cut, ped, None, 0)
            assert_spectrum(got[level], k_ref, case,
                            f"seed {seed} level {level}: v0={v0} span={span} npv={npv} cut={cut} "
                            f"lines={table.num_lines} ped={ped} p={p[level]:.3g}", k_plain,
                            conditioning=(lambda: oracle_conditioning(
                                oracle, table, t[level], p[level], x[level], v0, vn, npv, cut,
                                k_ref)) if ped else None)
    finally:
        engine.free(molecule)
        for name, value in (("farfield", 0), ("points_per_lane", 0), ("scan_chain", 1)):
            engine.set_option(name, value)
