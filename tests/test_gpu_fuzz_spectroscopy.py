"""Seeded random atmospheres through Spectroscopy.compute_absorption(): random levels, gases,
grids, line tables, cross-section files and output formats, every mechanism slot checked
against the composition of the three oracles (64 cases; PYLBL_FUZZ_SPECTROSCOPY=1000 to soak)."""
import os

import numpy as np
import pytest

from pylbl_amd import Spectroscopy, arts_crossfit, number_density, synthetic
from pylbl_amd.database import Database, write_database
from tests import golden_io

pytestmark = pytest.mark.gpu

GASES = ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2")
CONTINUA = {"H2O": ("H2OForeign", "H2OSelf"), "CO2": ("CO2",), "O3": ("O3",), "O2": ("O2",),
            "N2": ("N2",)}


@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_SPECTROSCOPY", "64"))))
def test_random_atmosphere(tmp_path, oracle, continuum_oracle, seed):
    from oracle import xsec_oracle
    rng = np.random.default_rng(31_000 + seed)
    npv = int(rng.choice([1, 4, 10, 25, 100]))
    v0 = int(rng.integers(1, 2500))
    span = int(rng.integers(5, max(6, min(150, 20_000//npv))))
    grid = v0 + np.arange(span*npv - int(rng.integers(0, npv)))/npv      # may stop short of v0+span
    grid_v0, grid_vn, grid_npv = synthetic.grid_arguments(grid)
    levels = int(rng.integers(1, 6))
    shape = (levels,) if rng.random() < 0.7 or levels % 2 else (levels//2, 2)
    t = rng.uniform(190., 320., levels)
    p = 10.**rng.uniform(1., 5.05, levels)
    # H2O always present (every continuum needs its mole fraction), the others at random.
    gases = ["H2O"] + [g for g in GASES[1:] if rng.random() < 0.6]
    if ("O2" in gases) != ("N2" in gases):
        # The O2 and N2 continua read each other's mole fraction (oxygen.py:44, nitrogen.py:25):
        # an atmosphere with only one of them is a KeyError in the reference too.
        gases = [g for g in GASES if g in gases or g in ("O2", "N2")]
    vmr = {g: 10.**rng.uniform(-8., -1., levels) for g in gases}
    with_lines = {g for g in gases if rng.random() < 0.7}
    with_xsec = {g for g in gases if rng.random() < 0.3}
    lo, hi = max(grid_v0 - 26., 0.05), grid_vn + 26.
    tables = [synthetic.line_table(g, lo, hi, num_lines=int(rng.integers(1, 400)),
                                   seed=int(rng.integers(1 << 30)), tips_range=(150, 400))
              for g in gases]
    bands = {}
    files = {}
    for g in with_xsec:
        a = grid_v0 + span*rng.uniform(-0.2, 0.6)
        ranges = ((a, a + span*0.3), (a + span*0.35, a + span*0.5))
        # At least a handful of frequencies per band (scipy's interp1d, and the engine, refuse
        # fewer than two).
        bands[g] = synthetic.cross_section_bands(
            seed=int(rng.integers(1 << 20)), ranges=ranges,
            spacing=float(min(rng.uniform(0.02, 0.5), span*0.15/8.)))
        files[g] = tmp_path / f"{g}.npz"
        arts_crossfit.write_npz(files[g], bands[g])
    path = tmp_path / "lines.db"
    write_database(path, tables, with_tips=with_lines, cross_sections={g: str(f) for g, f in files.items()})
    atmosphere = synthetic.Atmos(p=p.reshape(shape), t=t.reshape(shape),
                                 vmr={g: x.reshape(shape) for g, x in vmr.items()})
    ped = bool(rng.integers(0, 2))
    spec = Spectroscopy(atmosphere, grid, Database(str(path)))
    if rng.random() < 0.3:
        spec.device_output_limit = 0                        # the too-large-for-HBM route
    out = spec.compute_absorption(output_format="all", remove_pedestal=ped)
    per_gas = spec.compute_absorption(output_format="gas", remove_pedestal=ped)
    total = spec.compute_absorption(output_format="total", remove_pedestal=ped)
    by_table = {table.formula: table for table in tables}
    summed = np.zeros((levels, grid.size))
    for g in gases:
        beta = np.asarray(out[f"{g}_absorption"]).reshape(levels, 3, grid.size)
        for level in range(levels):
            label = f"seed {seed} {g} level {level} npv={npv} ped={ped}"
            state = {name: vmr[name][level] for name in gases}
            expect = np.zeros(grid.size)
            if g in with_lines:
                k, _ = oracle.absorption_port(by_table[g], t[level], p[level], vmr[g][level],
                                              grid_v0, grid_vn, grid_npv, remove_pedestal=ped)
                plain, _ = oracle.absorption_port(by_table[g], t[level], p[level], vmr[g][level],
                                                  grid_v0, grid_vn, grid_npv)
                density = number_density(t[level], p[level], vmr[g][level])
                expect = density*k[:grid.size]
                tol = 1e-6*np.abs(expect) + 1e-300
                if ped:
                    tol = density*np.maximum(
                        golden_io.pedestal_tolerance(k, grid_npv, 25, 1e-6),
                        np.maximum(1e-6*np.abs(plain),
                                   golden_io.pedestal_tolerance(plain, grid_npv, 25, 1e-13))
                    )[:grid.size] + 1e-300
                assert np.all(np.abs(beta[level, 0] - expect) <= tol), label + " lines"
            else:
                assert not beta[level, 0].any(), label + " lines of a gas without TIPS rows"
            continuum = np.zeros(grid.size)
            for owner in CONTINUA.get(g, ()):
                continuum = continuum + continuum_oracle.continuum(owner).spectra(
                    t[level], p[level], state, grid)
            scale = max(np.max(np.abs(continuum)), 1e-300)
            assert np.all(np.abs(beta[level, 1] - continuum) <=
                          1e-6*np.abs(continuum) + 1e-13*scale), label + " continuum"
            cross = np.zeros(grid.size)
            if g in with_xsec:
                cross = number_density(t[level], p[level], vmr[g][level]) * \
                    xsec_oracle.absorption_coefficient(bands[g], grid, t[level], p[level])
            scale = max(np.max(np.abs(cross)), 1e-300)
            assert np.all(np.abs(beta[level, 2] - cross) <= 1e-6*np.abs(cross) + 1e-12*scale), \
                label + " cross-section"
        both = beta.sum(axis=1)
        gas_sum = np.asarray(per_gas[f"{g}_absorption"]).reshape(levels, grid.size)
        assert np.all(np.abs(gas_sum - both) <= 1e-12*np.max(np.abs(both)) + 1e-300), g
        summed += both
    got_total = np.asarray(total["absorption"]).reshape(levels, grid.size)
    assert np.all(np.abs(got_total - summed) <= 1e-11*np.max(np.abs(summed)) + 1e-300)
