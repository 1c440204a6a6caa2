"""MT-CKD continua (mechanism slot 1) on a real GPU against the numpy oracle and the
reference's own known answers; everything goes through the C ABI (lbl_continuum_*)."""
import numpy as np
import pytest

from pylbl_amd import mt_ckd, synthetic
from pylbl_amd.engine import DeviceSpectra, EngineError, default_engine
from tests.test_continuum_oracle import KNOWN_ANSWERS, last_level

pytestmark = pytest.mark.gpu

OWNERS = ("H2OForeign", "H2OSelf", "CO2", "N2", "O2", "O3")
RTOL = 1.e-6        # the parity bar; observed differences are ~1e-15 (printed by the tests)


@pytest.fixture(scope="module")
def continua():
    return {owner: mt_ckd.CONTINUA[owner]() for owner in OWNERS}


def level_dictionaries(atmos):
    return [{name: values[i] for name, values in atmos.vmr.items()}
            for i in range(atmos.t.size)]


def assert_close(got, expect, label):
    assert got.shape == expect.shape, label
    assert np.array_equal(np.isnan(got), np.isnan(expect)), label
    scale = np.nanmax(np.abs(expect)) if np.isfinite(expect).any() else 0.
    finite = np.isfinite(expect)
    error = np.abs(got[finite] - expect[finite])
    bound = RTOL*np.abs(expect[finite]) + 1e-300
    assert np.all(error <= bound), f"{label}: max rel {np.max(error/(np.abs(expect[finite]) + 1e-300)):.3e}"
    return float(np.max(error)/scale) if scale > 0. else 0.


def test_band_known_answers(continua, continuum_oracle):
    """The reference's test (tests/test_mt_ckd.py:29-46) run against the GPU classes: same
    loop, same pytest.approx bar, plus point-by-point agreement with the oracle."""
    temperature, pressure, vmr = last_level()
    for owner, answers in KNOWN_ANSWERS.items():
        bands = continua[owner].bands
        expect = continuum_oracle.continuum(owner).band_spectra(temperature, pressure, vmr)
        assert len(bands) == len(answers)
        for band, answer, reference in zip(bands, answers, expect):
            spectrum = band.spectra(temperature, pressure, vmr)
            assert answer == pytest.approx(np.sum(spectrum))
            assert np.array_equal(band.grid(), continuum_oracle.coarse_grid(
                band.lower_bound, band.resolution, band.size))
            assert_close(spectrum, reference, f"{owner} band")


GRIDS = {
    "reference test grid": lambda: np.arange(1., 3250., 0.1),
    "coarse": lambda: np.arange(1., 3000., 1.),
    "on the knots": lambda: np.arange(-30., 20100., 10.),
    "near infrared": lambda: np.arange(7000., 14000., 0.25),
    "visible and ultraviolet": lambda: np.arange(14000., 101000., 1.7),
    "scattered": lambda: np.sort(np.random.default_rng(5).uniform(-100., 101000., 150_000)),
    "single point": lambda: np.asarray([2400.]),
}


@pytest.mark.parametrize("name", list(GRIDS))
def test_spectra_on_grids(continua, continuum_oracle, name):
    """BandedContinuum.spectra for the four fixture levels, one level per call like the
    reference's loop (spectroscopy.py:193-197)."""
    grid = GRIDS[name]()
    atmos = synthetic.fixture_atmosphere()
    worst = 0.
    for owner in OWNERS:
        for level, vmr in enumerate(level_dictionaries(atmos)):
            got = continua[owner].spectra(atmos.t[level], atmos.p[level], vmr, grid)
            expect = continuum_oracle.continuum(owner).spectra(atmos.t[level], atmos.p[level],
                                                               vmr, grid)
            worst = max(worst, assert_close(got, expect, f"{owner} level {level} on {name}"))
    print(f"{name}: worst difference {worst:.2e} of the spectrum maximum")


def test_knots_and_band_edges(continua, continuum_oracle):
    """Points on, one ulp below and one ulp above every coarse knot near the band ends: the
    interval search must agree with numpy.interp's, and the result is 0 outside the band."""
    temperature, pressure, vmr = last_level()
    for owner in OWNERS:
        points = []
        for band in continua[owner].bands:
            knots = band.grid()
            picked = np.concatenate([knots[:6], knots[-6:], knots[knots.size//2:knots.size//2 + 3]])
            points += [picked, np.nextafter(picked, -np.inf), np.nextafter(picked, np.inf)]
        grid = np.unique(np.concatenate(points))
        got = continua[owner].spectra(temperature, pressure, vmr, grid)
        expect = continuum_oracle.continuum(owner).spectra(temperature, pressure, vmr, grid)
        assert_close(got, expect, f"{owner} knots")
        first = min(b.lower_bound for b in continua[owner].bands)
        assert got[grid < first].size > 0 and not got[grid < first].any()


def test_batched_levels_device_output_and_accumulate(continua, continuum_oracle):
    atmos = synthetic.standard_atmosphere(24)
    grid = np.arange(1., 5000., 0.05)
    engine = default_engine(0)
    dictionaries = level_dictionaries(atmos)
    for owner in ("H2OSelf", "O2", "N2"):
        continuum = continua[owner]
        expect = np.stack([continuum_oracle.continuum(owner).spectra(
            atmos.t[i], atmos.p[i], dictionaries[i], grid) for i in range(atmos.t.size)])
        got = continuum.spectra_levels(atmos.t, atmos.p, atmos.vmr, grid)
        assert_close(got, expect, f"{owner} batched")
        # Rows longer than the grid (the lines path pads to whole wavenumbers), left in HBM,
        # written and then added to.
        padded = grid.size + 17
        block = DeviceSpectra(engine, atmos.t.size, padded)
        continuum.spectra_levels(atmos.t, atmos.p, atmos.vmr, grid, out=block, asynchronous=True)
        continuum.spectra_levels(atmos.t, atmos.p, atmos.vmr, grid, out=block, accumulate=True,
                                 asynchronous=True)
        engine.synchronize()
        assert_close(block.to_host()[:, :grid.size], 2.*expect, f"{owner} accumulated in HBM")
        block.free()
        host = np.full((atmos.t.size, grid.size), 1.5)
        continuum.spectra_levels(atmos.t, atmos.p, atmos.vmr, grid, out=host, accumulate=True)
        assert_close(host, expect + 1.5, f"{owner} accumulated on the host")


def test_full_size_grid(continua, continuum_oracle):
    """The benchmark's grid (5 M points, 0.001 cm-1): whole spectrum against the oracle."""
    grid = np.arange(1., 5000., 0.001)
    temperature, pressure, vmr = last_level()
    for owner in ("H2OForeign", "H2OSelf", "CO2"):
        got = continua[owner].spectra(temperature, pressure, vmr, grid)
        expect = continuum_oracle.continuum(owner).spectra(temperature, pressure, vmr, grid)
        assert_close(got, expect, f"{owner} 5 M points")
        assert got.min() >= 0. and got.max() > 0.


def test_grids_are_uploaded_once(continua):
    engine = default_engine(0)
    grid = np.arange(100., 200., 0.5)
    temperature, pressure, vmr = last_level()
    continua["CO2"].spectra(temperature, pressure, vmr, grid)
    resident = [entry for entry in engine._resident_grids if entry[0]() is grid]
    continua["O3"].spectra(temperature, pressure, vmr, grid)
    assert len(resident) == 1
    assert [entry for entry in engine._resident_grids if entry[0]() is grid] == resident


def test_error_paths(continua):
    temperature, pressure, vmr = last_level()
    grid = np.arange(1., 10., 1.)
    with pytest.raises(KeyError):       # the reference's formulas index the dictionary
        continua["N2"].spectra(temperature, pressure, {"N2": 0.78, "H2O": 1e-3}, grid)
    engine = default_engine(0)
    with pytest.raises(ValueError):
        engine.load_continuum([(0, 0., 1., [np.ones(4), np.ones(4)])]*9)
    with pytest.raises(EngineError):    # formula 3 reads four columns
        engine.load_continuum([(3, 0., 1., [np.ones(4)])])
    with pytest.raises(EngineError):
        engine.load_continuum([(99, 0., 1., [np.ones(4)])])
    with pytest.raises(EngineError):
        engine.continuum_compute(12345, 0, 9, [250.], [1e4], np.zeros((1, 5)))
    handle = engine.load_grid(grid)
    engine.free_grid(handle)
    with pytest.raises(EngineError):
        engine.free_grid(handle)


def test_spectroscopy_continuum_slot(continuum_oracle, tmp_path):
    """Slot 1 of compute_absorption (spectroscopy.py:193-197): every continuum of a gas added
    up, no number-density factor; "gas" and "total" sums formed on the device."""
    from pylbl_amd import Spectroscopy
    from pylbl_amd.database import Database, write_database
    # Lines for H2O and CO2 only; the other gases are known to the database but have no
    # partition functions, which the reference answers with zeros (absorption.c:53-59).
    tables = [synthetic.line_table(formula, 1., 130., num_lines=300 if formula in ("H2O", "CO2")
                                   else 5, seed=51 + i, tips_range=(150, 400))
              for i, formula in enumerate(synthetic.fixture_atmosphere().vmr)]
    path = tmp_path / "lines.db"
    write_database(path, tables, with_tips={"H2O", "CO2"})
    atmos = synthetic.fixture_atmosphere()
    grid = np.arange(1., 120., 0.1)
    spec = Spectroscopy(atmos, grid, Database(str(path)))
    out = spec.compute_absorption(output_format="all")
    dictionaries = level_dictionaries(atmos)
    owners = {"H2O": ("H2OForeign", "H2OSelf"), "CO2": ("CO2",), "O3": ("O3",), "O2": ("O2",),
              "N2": ("N2",), "N2O": (), "CH4": (), "CO": ()}
    expect = {}
    for formula, names in owners.items():
        beta = np.asarray(out[f"{formula}_absorption"])
        assert beta.shape == (4, 3, grid.size) and not beta[:, 2].any()
        expect[formula] = np.zeros((4, grid.size))
        for level in range(4):
            for owner in names:
                expect[formula][level] += continuum_oracle.continuum(owner).spectra(
                    atmos.t[level], atmos.p[level], dictionaries[level], grid)
        assert_close(beta[:, 1], expect[formula], f"{formula} continuum slot")
        assert beta[:, 0].any() == (formula in ("H2O", "CO2"))
    per_gas = spec.compute_absorption(output_format="gas")
    total = spec.compute_absorption(output_format="total")
    summed = np.zeros((4, grid.size))
    for formula in owners:
        both = np.asarray(out[f"{formula}_absorption"]).sum(axis=1)
        gas = np.asarray(per_gas[f"{formula}_absorption"])
        assert np.max(np.abs(gas - both)) <= 1e-12*np.max(np.abs(both)) + 1e-300
        summed += both
    assert np.max(np.abs(np.asarray(total["absorption"]) - summed)) <= 1e-12*np.max(summed)
    # Mechanism switched off: slot 1 stays zero and the pedestal stays in by default.
    lines_only = Spectroscopy(atmos, grid, Database(str(path)), continua_backend=None)
    plain = lines_only.compute_absorption(output_format="all")
    assert not np.asarray(plain["H2O_absorption"])[:, 1:].any()
    with pytest.raises(KeyError):
        Spectroscopy(atmos, grid, None, continua_backend="not-a-model")


def test_compute_absorption_is_reproducible_call_to_call(tmp_path):
    """Results travel on a copy stream into recycled page-locked arrays beside the next gas's
    kernels: every call must return the same bits (PYLBL_SOAK_ROUNDS=200 for a soak)."""
    import gc
    import os
    from pylbl_amd import Spectroscopy
    from pylbl_amd.database import Database, write_database
    atmos = synthetic.fixture_atmosphere()
    tables = [synthetic.line_table(formula, 1., 230., num_lines=2000 if formula in ("H2O", "CO2", "O3")
                                   else 5, seed=91 + i, tips_range=(150, 400))
              for i, formula in enumerate(atmos.vmr)]
    path = tmp_path / "lines.db"
    write_database(path, tables, with_tips={"H2O", "CO2", "O3"})
    grid = np.arange(1., 200., 0.01)
    spec = Spectroscopy(atmos, grid, Database(str(path)))
    first = {}
    for round_ in range(int(os.environ.get("PYLBL_SOAK_ROUNDS", "3"))):
        for output_format in ("all", "gas", "total"):
            out = spec.compute_absorption(output_format=output_format)
            for key, value in out.items():
                if key in ("wavenumber", "mechanism"):
                    continue
                value = np.asarray(value)
                if round_ == 0:
                    first[(output_format, key)] = value.copy()
                else:
                    assert np.array_equal(value, first[(output_format, key)]), (round_, key)
            del out
            gc.collect()


@pytest.mark.parametrize("grid_name", ["reference test grid", "scattered", "near infrared",
                                        "on the knots"])
@pytest.mark.parametrize("owners", [("H2OForeign", "H2OSelf"), ("H2OForeign", "H2OSelf", "CO2"),
                                    OWNERS, ("O2",)])
def test_several_continua_in_one_pass_give_the_same_bits(continua, continuum_oracle, owners,
                                                         grid_name):
    """lbl_continuum_compute_many (what Spectroscopy uses for the continua of a gas, and of all
    gases in its "total" format, spectroscopy.py:193-197,225-234): bit for bit what one
    lbl_continuum_compute per continuum leaves in the block -- written by the first, added to by
    the others -- for 1, 5 and 24 levels, writing and adding onto a block that holds something,
    on arithmetic grids (numpy.arange: the wavenumber is formed in registers) and on a scattered
    one (read from HBM); and each within the bar of the oracle's sum."""
    grid = GRIDS[grid_name]()
    engine = default_engine(0)
    for levels in (1, 5, 24):
        atmos = synthetic.standard_atmosphere(max(levels, 2))
        t, p = atmos.t[:levels], atmos.p[:levels]
        vmr = {name: values[:levels] for name, values in atmos.vmr.items()}
        members = [continua[owner] for owner in owners]
        padded = grid.size + 9
        for start in (None, 0.75):
            one_by_one = DeviceSpectra(engine, levels, padded)
            together = DeviceSpectra(engine, levels, padded)
            if start is not None:
                # something in the block already: a first continuum of another gas, say
                for block in (one_by_one, together):
                    continua["O3"].spectra_levels(t, p, vmr, grid, out=block)
            for i, continuum in enumerate(members):
                continuum.spectra_levels(t, p, vmr, grid, out=one_by_one,
                                         accumulate=(start is not None) or i > 0,
                                         asynchronous=True)
            mt_ckd.spectra_levels_many(members, t, p, vmr, grid, together,
                                       accumulate=start is not None, asynchronous=True)
            engine.synchronize()
            a = one_by_one.to_host()[:, :grid.size]
            b = together.to_host()[:, :grid.size]
            assert np.array_equal(a, b, equal_nan=True), (owners, grid_name, levels, start)
            if start is None:
                dictionaries = level_dictionaries(synthetic.Atmos(p=p, t=t, vmr=vmr))
                expect = sum(np.stack([continuum_oracle.continuum(owner).spectra(
                    t[i], p[i], dictionaries[i], grid) for i in range(levels)])
                    for owner in owners)
                scale = np.max(np.abs(expect))
                assert np.all(np.abs(b - expect) <= RTOL*np.abs(expect) + 1e-12*scale)
            one_by_one.free()
            together.free()


def test_arithmetic_grids_are_recognised_exactly(continua, continuum_oracle):
    """A grid is taken as first + i*step only where every element says so: numpy.arange grids
    are, numpy.linspace's and an arange grid with ONE element moved by an ulp are not -- and all
    three give what the oracle gives for the array as it is (the moved point included)."""
    temperature, pressure, vmr = last_level()
    exact = np.arange(600., 2600., 0.01)
    moved = exact.copy()
    moved[123_457] = np.nextafter(moved[123_457], np.inf)
    spaced = np.linspace(600., 2600., 200_001)
    for name, grid in (("arange", exact), ("one ulp off", moved), ("linspace", spaced)):
        for owner in ("H2OSelf", "CO2"):
            got = continua[owner].spectra(temperature, pressure, vmr, grid)
            expect = continuum_oracle.continuum(owner).spectra(temperature, pressure, vmr, grid)
            assert_close(got, expect, f"{owner} on {name}")
