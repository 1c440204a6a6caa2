"""Seeded fuzz with HITRAN-shaped values.

Every other table in this repository comes from one parameter box (pylbl_amd/synthetic.py:
gamma_self 0.05-0.5, n_air 0.4-0.85, elower 0-5000, sw 1e-30...1e-19, four isotopologues,
partition sums on 100-900 K).  The tables the reference's ingest produces
(pyLBL/database.py:80-127, tests/test_database.py:20-25) are wider: half-widths of exactly zero,
n_air <= 0, elower = -1 (HITRAN's "unknown"), line positions down to 1.3e-4 cm-1, strengths far
below 1e-30, duplicated positions, twelve isotopologues with HITRAN's id 0 for the tenth
(spectral_database.c:173-177), partition sums from 1 K to 5000 K.  The host's bound that switches
the inner-region pass off (engine.hip: inner_possible), the products inside lorentz_eight and the
core index ranges are the code such values stress.  Every case is compared with the CPU oracle
at the 1e-6 bar (spectra.c:17-62, voigt.c:17-53), far-field series on and off, pedestal on and
off, both line-scalar preparations."""
import os

import numpy as np
import pytest

from pylbl_amd.errors import EngineError
from tests import golden_io
from tests.hitran_shapes import hitran_shaped_table
from tests.test_gpu_parity import assert_spectrum, oracle_conditioning

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from pylbl_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def make_case(seed):
    rng = np.random.default_rng(77000 + seed)
    bottom = rng.random() < 0.35
    v0 = int(rng.choice([0, 1, 1])) if bottom else int(rng.integers(1, 4000))
    span = int(rng.integers(2, 50))
    npv = int(rng.choice([1, 4, 10, 100, 250, 1000]))
    if span*npv > 30000:
        span = max(30000//npv, 2)
    vn = v0 + span
    cut = int(rng.choice([25, 25, 25, 3, 40]))
    lo, hi = max(v0 - cut - 1., 1.e-4), vn + cut + 1.
    n_lines = int(rng.integers(1, 2500))
    table = hitran_shaped_table(rng, lo, hi, n_lines)
    levels = int(rng.integers(1, 4))
    # temperatures at both ends of the partition sums' table, and in between
    t = rng.choice([1., 1.5, 2.25, 70., 4999., 4999.9], levels)
    ordinary = rng.random(levels) < 0.6
    t[ordinary] = rng.uniform(150., 350., int(ordinary.sum()))
    return dict(v0=v0, vn=vn, npv=npv, cut=cut, table=table, levels=levels, t=t,
                p=10.**rng.uniform(-2., 6., levels), x=10.**rng.uniform(-7., 0., levels),
                ped=bool(rng.integers(0, 2)),
                options=dict(farfield=int(rng.integers(0, 2)), prep=int(rng.random() < 0.3),
                             points_per_lane=int(rng.choice([0, 0, 1, 2, 4, 8]))))


@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_HITRAN", "64"))))
def test_hitran_shaped_case(engine, oracle, seed):
    c = make_case(seed)
    v0, vn, npv, cut, ped, table = c["v0"], c["vn"], c["npv"], c["cut"], c["ped"], c["table"]
    t, p, x = c["t"], c["p"], c["x"]
    for name, value in c["options"].items():
        engine.set_option(name, value)
    molecule = engine.load(table)
    try:
        got = engine.compute(molecule, t, p, x, v0, vn, npv, cut_off=cut, remove_pedestal=ped,
                             range_policy="skip")
        source = table.subset((table.nu >= v0 - (cut + 1)) & (table.nu <= vn + cut + 1))
        for level in range(c["levels"]):
            k_ref, _ = oracle.absorption_port(source, t[level], p[level], x[level], v0, vn, npv,
                                              cut_off=cut, remove_pedestal=ped)
            k_plain, _ = oracle.absorption_port(source, t[level], p[level], x[level], v0, vn,
                                                npv, cut_off=cut)
            assert np.all(np.isfinite(k_ref)), "the reference itself is not finite here"
            case = golden_io.Case("hitran", seed, 0, 0, 0, v0, vn, npv, cut, ped, None, 0)
            assert_spectrum(got[level], k_ref, case,
                            f"seed {seed} level {level}: v0={v0} vn={vn} npv={npv} cut={cut} "
                            f"lines={table.num_lines} ped={ped} T={t[level]:g} p={p[level]:.3g} "
                            f"{c['options']}", k_plain,
                            conditioning=lambda: oracle_conditioning(
                                oracle, source, t[level], p[level], x[level], v0, vn, npv, cut,
                                k_ref))
    finally:
        engine.free(molecule)
        for name in c["options"]:
            engine.set_option(name, 0)


def test_one_step_outside_the_partition_sums(engine):
    """The reference reads past its table there (spectral_database.c:97-104 has no bounds
    check); here it is the documented error (LBL_OUT_OF_RANGE), at either end."""
    rng = np.random.default_rng(5)
    table = hitran_shaped_table(rng, 600., 700., 200)
    molecule = engine.load(table)
    try:
        for bad in (0.5, 5000., 5000.5, 7000.):
            with pytest.raises(EngineError, match="partition-function table"):
                engine.compute(molecule, bad, 5.e4, 4.e-4, 620, 680, 10)
        for fine in (1., 4999.99):
            assert np.all(np.isfinite(engine.compute(molecule, fine, 5.e4, 4.e-4, 620, 680, 10)))
    finally:
        engine.free(molecule)
