"""ARTS-crossfit cross-sections (mechanism slot 2) on a real GPU against the oracle (the
reference's fit model restated + the scipy interp1d the reference calls), through the C ABI
(lbl_xsec_*)."""
import numpy as np
import pytest

from pylbl_amd import arts_crossfit, synthetic
from pylbl_amd.engine import DeviceSpectra, EngineError, default_engine
from tests.test_xsec_oracle import golden_cases

pytestmark = pytest.mark.gpu

RTOL = 1.e-6


def assert_close(got, expect, label):
    """1e-6 of the value, plus 1e-6 of the band maximum times machine epsilon-ish slack where
    the clipping rule puts exact zeros next to values that are zero only up to rounding."""
    assert got.shape == expect.shape, label
    scale = np.max(np.abs(expect))
    error = np.abs(got - expect)
    assert np.all(error <= RTOL*np.abs(expect) + 1e-12*scale), \
        f"{label}: max error {np.max(error):.3e} of scale {scale:.3e}"
    return float(np.max(error)/scale) if scale > 0. else 0.


@pytest.fixture(scope="module")
def molecule(tmp_path_factory):
    bands = synthetic.cross_section_bands(seed=3)
    path = tmp_path_factory.mktemp("xsec") / "CFC11.npz"
    arts_crossfit.write_npz(path, bands)
    return arts_crossfit.CrossSection("CFC11", str(path)), bands


def test_fit_model_against_reference_vectors():
    """calculate_xsec_fullmodel: the reference's own outputs (tests/golden/xsec_model.npz)."""
    engine = default_engine(0)
    worst = 0.
    for case, coeffs, temperature, pressure, expect in golden_cases():
        frequency = np.linspace(1e13, 2e13, coeffs.shape[1])
        handle = engine.load_xsec([(frequency, coeffs)])
        got = engine.xsec_bands(handle, [coeffs.shape[1]], temperature, pressure)[0]
        worst = max(worst, assert_close(got, expect, f"case {case}"))
        assert np.array_equal(got == 0., expect == 0.), case      # same points clipped
        engine.free_xsec(handle)
    print(f"fit model: worst difference {worst:.2e} of the band maximum")


GRIDS = {
    "reference test grid": lambda: np.arange(1., 3250., 0.1),
    "fine over the bands": lambda: np.arange(550., 1300., 0.001),
    "coarse": lambda: np.arange(1., 3000., 1.),
    "scattered, not ascending": lambda: np.random.default_rng(8).uniform(500., 1400., 100_000),
    "single point": lambda: np.asarray([700.]),
}


@pytest.mark.parametrize("name", list(GRIDS))
def test_absorption_coefficient_on_grids(molecule, name):
    """CrossSection.absorption_coefficient for the four fixture levels, one per call like
    the reference's loop (spectroscopy.py:199-203)."""
    from oracle import xsec_oracle
    cross, bands = molecule
    grid = GRIDS[name]()
    atmos = synthetic.fixture_atmosphere()
    worst = 0.
    for level in range(atmos.t.size):
        got = cross.absorption_coefficient(grid, atmos.t[level], atmos.p[level])
        expect = xsec_oracle.absorption_coefficient(bands, grid, atmos.t[level], atmos.p[level])
        worst = max(worst, assert_close(got, expect, f"level {level} on {name}"))
    print(f"{name}: worst difference {worst:.2e} of the spectrum maximum")


def test_band_knots_and_edges(molecule):
    from oracle import xsec_oracle
    cross, bands = molecule
    points = []
    for frequency, _ in bands:
        w = frequency/(299792458.0*100)
        picked = np.concatenate([w[:5], w[-5:], w[w.size//2:w.size//2 + 3]])
        points += [picked, np.nextafter(picked, -np.inf), np.nextafter(picked, np.inf)]
    grid = np.unique(np.concatenate(points))
    got = cross.absorption_coefficient(grid, 260., 5e4)
    expect = xsec_oracle.absorption_coefficient(bands, grid, 260., 5e4)
    assert_close(got, expect, "knots")
    assert got[0] == 0. or grid[0]*299792458.0*100 >= bands[0][0][0]


def test_batched_density_scaling_and_accumulate(molecule):
    from oracle import xsec_oracle
    from pylbl_amd import number_density
    cross, bands = molecule
    atmos = synthetic.standard_atmosphere(19)
    vmr = np.linspace(1e-10, 3e-10, atmos.t.size)
    grid = np.arange(500., 1400., 0.01)
    engine = default_engine(0)
    k = np.stack([xsec_oracle.absorption_coefficient(bands, grid, atmos.t[i], atmos.p[i])
                  for i in range(atmos.t.size)])
    scaled = number_density(atmos.t, atmos.p, vmr)[:, None]*k
    assert_close(cross.absorption_coefficients(grid, atmos.t, atmos.p), k, "batched")
    assert_close(cross.absorption_coefficients(grid, atmos.t, atmos.p, volume_mixing_ratio=vmr),
                 scaled, "n k")
    block = DeviceSpectra(engine, atmos.t.size, grid.size + 9)
    cross.absorption_coefficients(grid, atmos.t, atmos.p, volume_mixing_ratio=vmr, out=block,
                                  asynchronous=True)
    cross.absorption_coefficients(grid, atmos.t, atmos.p, volume_mixing_ratio=vmr, out=block,
                                  accumulate=True, asynchronous=True)
    engine.synchronize()
    assert_close(block.to_host()[:, :grid.size], 2.*scaled, "accumulated in HBM")
    block.free()
    host = np.full((atmos.t.size, grid.size), 1e-25)
    cross.absorption_coefficients(grid, atmos.t, atmos.p, out=host, accumulate=True)
    assert_close(host, k + 1e-25, "accumulated on the host")


def test_error_paths():
    engine = default_engine(0)
    f = np.linspace(1e13, 2e13, 8)
    with pytest.raises(ValueError):
        engine.load_xsec([(f, np.zeros((3, 8)))])
    with pytest.raises(EngineError):                    # not ascending
        engine.load_xsec([(f[::-1].copy(), np.zeros((4, 8)))])
    with pytest.raises(ValueError):
        engine.load_xsec([(f, np.zeros((4, 8)))]*17)
    with pytest.raises(EngineError):
        engine.xsec_compute(4242, 0, 3, [250.], [1e4])
    handle = engine.load_xsec([(f, np.zeros((4, 8)))])
    engine.free_xsec(handle)
    with pytest.raises(EngineError):
        engine.free_xsec(handle)


def test_spectroscopy_cross_section_slot(tmp_path):
    """Slot 2 of compute_absorption = n k (spectroscopy.py:199-203) for the gases whose
    database entry lists a coefficient file; sums over mechanisms and gases on the device."""
    from oracle import xsec_oracle
    from pylbl_amd import Spectroscopy, number_density
    from pylbl_amd.database import Database, write_database
    bands = synthetic.cross_section_bands(seed=5, ranges=((20., 60.), (80., 110.)), spacing=0.02)
    coefficients = tmp_path / "N2O.npz"
    arts_crossfit.write_npz(coefficients, bands)
    atmos = synthetic.fixture_atmosphere()
    tables = [synthetic.line_table(formula, 1., 130., num_lines=200 if formula == "CO2" else 5,
                                   seed=61 + i, tips_range=(150, 400))
              for i, formula in enumerate(atmos.vmr)]
    path = tmp_path / "lines.db"
    write_database(path, tables, with_tips={"CO2"}, cross_sections={"N2O": str(coefficients)})
    grid = np.arange(1., 120., 0.1)
    spec = Spectroscopy(atmos, grid, Database(str(path)))
    out = spec.compute_absorption(output_format="all")
    expect = np.stack([number_density(atmos.t[i], atmos.p[i], atmos.vmr["N2O"][i]) *
                       xsec_oracle.absorption_coefficient(bands, grid, atmos.t[i], atmos.p[i])
                       for i in range(4)])
    beta = np.asarray(out["N2O_absorption"])
    assert_close(beta[:, 2], expect, "N2O cross-section slot")
    assert not beta[:, :2].any()                # no TIPS rows, no MT-CKD continuum for N2O
    assert not np.asarray(out["CO2_absorption"])[:, 2].any()
    per_gas = spec.compute_absorption(output_format="gas")
    total = spec.compute_absorption(output_format="total")
    summed = np.zeros((4, grid.size))
    for formula in atmos.vmr:
        both = np.asarray(out[f"{formula}_absorption"]).sum(axis=1)
        gas = np.asarray(per_gas[f"{formula}_absorption"])
        assert np.max(np.abs(gas - both)) <= 1e-12*np.max(np.abs(both)) + 1e-300
        summed += both
    assert np.max(np.abs(np.asarray(total["absorption"]) - summed)) <= 1e-12*np.max(summed)
    off = Spectroscopy(atmos, grid, Database(str(path)), cross_sections_backend=None)
    assert not np.asarray(off.compute_absorption()["N2O_absorption"])[:, 2].any()
    with pytest.raises(KeyError):
        Spectroscopy(atmos, grid, None, cross_sections_backend="not-a-model")


def test_total_with_continuum_and_cross_section_of_two_gases(tmp_path, monkeypatch):
    """Output "total" queues every continuum of every gas in one pass, then the cross-sections,
    then the lines (Spectroscopy.compute_absorption): c(g1), c(g2), x(g1), x(g2), lines -- where the
    reference adds gas by gas, mechanism by mechanism (spectroscopy.py:225-234).  Two gases with
    lines, an MT-CKD continuum AND a cross-section each: the total equals the sum of the "all"
    output taken in the reference's order to rounding (1e-12 of the largest value)."""
    import os
    from pylbl_amd import Spectroscopy
    from pylbl_amd.database import Database, write_database
    fixture = os.path.join(os.path.dirname(__file__), "golden", "mt_ckd_bands.npz")
    monkeypatch.setenv("PYLBL_MT_CKD", fixture)
    atmos = synthetic.fixture_atmosphere()
    files = {}
    for i, formula in enumerate(("CO2", "H2O")):
        bands = synthetic.cross_section_bands(seed=21 + i, ranges=((30. + 20*i, 70. + 20*i),),
                                              spacing=0.02)
        files[formula] = str(tmp_path / f"{formula}.npz")
        arts_crossfit.write_npz(files[formula], bands)
    # (every gas of the atmosphere is in the database, as in the reference's fixture; only CO2 and
    # H2O have partition sums, i.e. lines)
    tables = [synthetic.line_table(formula, 1., 130., num_lines=300 if formula in files else 5,
                                   seed=71 + i, tips_range=(150, 400))
              for i, formula in enumerate(atmos.vmr)]
    path = tmp_path / "lines.db"
    write_database(path, tables, with_tips=set(files), cross_sections=files)
    level = atmos
    grid = np.arange(1., 120., 0.05)
    spec = Spectroscopy(level, grid, Database(str(path)))
    everything = spec.compute_absorption(output_format="all")
    total = np.asarray(spec.compute_absorption(output_format="total")["absorption"])
    in_reference_order = np.zeros_like(total)
    for formula in level.vmr:                   # (H2O, O2 and N2 ride along with their continua)
        if f"{formula}_absorption" not in everything:
            continue
        beta = np.asarray(everything[f"{formula}_absorption"])
        if formula in ("CO2", "H2O"):
            assert beta[:, 0].any() and beta[:, 1].any() and beta[:, 2].any(), formula
        for mechanism in range(3):
            in_reference_order += beta[:, mechanism]
    assert np.max(np.abs(total - in_reference_order)) <= 1e-12*np.max(in_reference_order)


def test_coefficient_file_in_the_reference_layout():
    """The netCDF-4 file of tests/golden (layout of cross_section.py:29-41) through
    CrossSection on the GPU against what the reference's own class returned for it
    (tests/golden/make_xsec_layout.py)."""
    from tests.test_xsec_oracle import _layout_paths
    try:
        from pylbl_amd import hdf5_reader
        hdf5_reader.library()
    except OSError as error:
        pytest.skip(str(error))
    path, expected = _layout_paths()
    cross = arts_crossfit.CrossSection("CFC11", str(path))
    assert cross.sizes == [241, 97, 4, 161]
    with np.load(expected) as data:
        for name in ("fine", "coarse", "knots"):
            grid = data[f"grid_{name}"]
            for i, (temperature, pressure) in enumerate(data["states"]):
                got = cross.absorption_coefficient(grid, temperature, pressure)
                assert_close(got, data[f"xsec_{name}_{i}"], f"{name} state {i}")
