"""Properties that hold at any size, checked at BASELINE.json's full sizes (5 M points, the
benchmark's own H2O and CO2 tables) where the oracle alone needs its whole farm of processes
(tests/test_gpu_baseline_configs.py does that comparison): what the reference's loops imply
without running them.

* spectra.c:45 multiplies every line strength into its profile and absorption.c:76-86 only adds:
  doubling every sw doubles k -- a power of two, so bit for bit, pedestal recurrence
  (spectra.c:66-78: min, subtraction) included;
* without the pedestal the row loop is a plain sum over rows: the spectrum of a table is the sum
  of the spectra of any two parts of it;
* spectroscopy.py:181-191 scales k by the number density: LBL_SCALE_DENSITY is k times one
  number per level;
* a call has no state: the same call gives the same bits whatever ran before it, on whichever
  stream, with the work items in either order.
"""
import numpy as np
import pytest

from pylbl_amd import synthetic

pytestmark = pytest.mark.gpu

V0, VN, NPV = 1, 5001, 1000       # the target configuration's grid: 5 M points


@pytest.fixture(scope="module")
def engine():
    from pylbl_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def level():
    atmos = synthetic.surface_level(["H2O", "CO2"])
    return atmos


def spectrum(engine, table, level, **options):
    molecule = engine.load(table)
    try:
        return engine.compute(molecule, level.t, level.p, level.vmr[table.formula], V0, VN, NPV,
                              **options)[0].copy()
    finally:
        engine.free(molecule)


@pytest.mark.parametrize("farfield", [False, True])
@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_doubling_every_strength_doubles_the_spectrum(engine, level, remove_pedestal, farfield):
    table = synthetic.line_table("H2O", 1., 5000.)
    k = spectrum(engine, table, level, remove_pedestal=remove_pedestal, farfield=farfield)
    doubled = table.subset(np.ones(table.num_lines, bool))
    doubled.sw = table.sw*2.
    k2 = spectrum(engine, doubled, level, remove_pedestal=remove_pedestal, farfield=farfield)
    assert k.max() > 0. and np.isfinite(k).all()
    assert np.array_equal(k2, 2.*k)


def test_spectrum_of_a_table_is_the_sum_over_its_parts(engine, level):
    table = synthetic.line_table("CO2", 1., 5000.)
    rng = np.random.default_rng(2024)
    part = rng.random(table.num_lines) < 0.5
    whole = spectrum(engine, table, level)
    parts = spectrum(engine, table.subset(part), level) + spectrum(engine, table.subset(~part), level)
    # The same additions in another order: a few units in the last place of the sum.
    assert np.max(np.abs(parts - whole)/whole.max()) < 1e-13
    nonzero = whole > 0.
    assert np.array_equal(parts > 0., nonzero)
    assert np.max(np.abs(parts[nonzero] - whole[nonzero])/whole[nonzero]) < 1e-12


@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_density_scaling_is_one_factor_per_level(engine, remove_pedestal):
    from pylbl_amd.spectroscopy import number_density
    table = synthetic.line_table("H2O", 1., 5000.)
    atmos = synthetic.fixture_atmosphere()
    molecule = engine.load(table)
    k = engine.compute(molecule, atmos.t, atmos.p, atmos.vmr["H2O"], V0, VN, NPV,
                       remove_pedestal=remove_pedestal).copy()
    nk = engine.compute(molecule, atmos.t, atmos.p, atmos.vmr["H2O"], V0, VN, NPV,
                        remove_pedestal=remove_pedestal, scale_density=True).copy()
    engine.free(molecule)
    density = number_density(atmos.t, atmos.p, atmos.vmr["H2O"])
    # (one multiplication per point by the level's n = P x / (kb T), formed once on the host)
    np.testing.assert_allclose(nk, k*density[:, None], rtol=5e-16, atol=0.)


def test_a_call_has_no_memory(engine, level):
    """Blocking and asynchronous calls (lane 0 / the two lanes plain calls take turns on), a
    different call in between: one set of bits."""
    from pylbl_amd.engine import DeviceSpectra
    h2o = synthetic.line_table("H2O", 1., 5000.)
    co2 = synthetic.line_table("CO2", 1., 5000.)
    first = spectrum(engine, co2, level, remove_pedestal=True)
    spectrum(engine, h2o, level, remove_pedestal=True, farfield=True)
    molecule = engine.load(co2)
    blocks = [DeviceSpectra(engine, 1, (VN - V0)*NPV) for _ in range(3)]
    for block in blocks:
        engine.compute(molecule, level.t, level.p, level.vmr["CO2"], V0, VN, NPV,
                       remove_pedestal=True, out=block, asynchronous=True)
    engine.synchronize()
    for block in blocks:
        assert np.array_equal(block.to_host()[0], first)
        block.free()
    engine.free(molecule)
