"""Seeded random sweeps of the continuum and cross-section slots against their oracles:
random levels (50 K .. 1000 K, 0.01 Pa .. 5e6 Pa, mole fractions down to zero), random grids
(any range between -100 and 101000 cm-1, ascending or scattered, 1 .. 60000 points), batches,
accumulation, random cross-section bands."""
import os

import numpy as np
import pytest

from pylbl_amd import mt_ckd
from pylbl_amd.engine import default_engine

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("PYLBL_FUZZ_CASES", "24"))
OWNERS = ("H2OForeign", "H2OSelf", "CO2", "N2", "O2", "O3")


def close(got, expect, label, floor=1e-13):
    """1e-6 of the value; `floor` of the largest value absorbs the residue where bands of
    very different size add up or the clipping rule leaves exact zeros."""
    assert got.shape == expect.shape, label
    assert np.array_equal(np.isnan(got), np.isnan(expect)), label
    finite = np.isfinite(expect)
    if not finite.any():
        return
    scale = np.max(np.abs(expect[finite]))
    error = np.abs(got[finite] - expect[finite])
    assert np.all(error <= 1e-6*np.abs(expect[finite]) + floor*scale + 1e-300), \
        f"{label}: max error {np.max(error):.3e}, scale {scale:.3e}"


def random_grid(rng):
    size = int(10**rng.uniform(0., 4.8))
    lower = rng.uniform(-100., 90000.)
    upper = lower + 10**rng.uniform(-1., 4.5)
    kind = rng.integers(0, 3)
    if kind == 0:
        return np.linspace(lower, upper, size)
    if kind == 1:
        return np.sort(rng.uniform(lower, upper, size))
    return rng.uniform(lower, upper, size)          # not ascending


@pytest.fixture(scope="module")
def continua():
    return {owner: mt_ckd.CONTINUA[owner]() for owner in OWNERS}


@pytest.mark.parametrize("seed", range(CASES))
def test_random_continuum(continua, continuum_oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    owner = OWNERS[int(rng.integers(0, len(OWNERS)))]
    grid = random_grid(rng)
    levels = int(rng.integers(1, 9))
    t = rng.uniform(50., 1000., levels)
    p = 10.**rng.uniform(-2., 6.7, levels)
    vmr = {name: 10.**rng.uniform(-9., -0.3, levels) for name in
           ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2")}
    if rng.random() < 0.2:
        vmr["H2O"] = np.zeros(levels)                   # dry air
    expect = np.stack([continuum_oracle.continuum(owner).spectra(
        t[i], p[i], {k: v[i] for k, v in vmr.items()}, grid) for i in range(levels)])
    got = continua[owner].spectra_levels(t, p, vmr, grid)
    close(got, expect, f"seed {seed} {owner} {levels} level(s) {grid.size} points")
    if rng.random() < 0.5:
        again = np.array(got)
        continua[owner].spectra_levels(t, p, vmr, grid, out=again, accumulate=True)
        close(again, 2.*expect, f"seed {seed} {owner} accumulated")


@pytest.mark.parametrize("seed", range(CASES))
def test_random_cross_section(seed):
    from oracle import xsec_oracle
    rng = np.random.default_rng(9000 + seed)
    engine = default_engine(0)
    bands = []
    for _ in range(int(rng.integers(1, 6))):
        size = int(10**rng.uniform(0.4, 4.2))
        lower = rng.uniform(100., 3000.)
        wavenumber = lower + np.cumsum(rng.uniform(1e-3, 1., size))
        coefficients = np.zeros((4, size))
        shape = 1e-22*np.exp(-((np.arange(size)/size - rng.uniform(0.2, 0.8))/0.2)**2)
        coefficients[0] = shape + rng.uniform(-1., 1.)*1e-23 + 2e-24*rng.standard_normal(size)
        coefficients[1] = 1e-26*rng.standard_normal(size)
        coefficients[2] = 1e-29*rng.standard_normal(size)
        coefficients[3] = 1e-29*rng.standard_normal(size)
        bands.append((wavenumber*299792458.0*100, coefficients))
    handle = engine.load_xsec(bands)
    try:
        grid = random_grid(rng)
        grid = grid*(4000./max(np.max(np.abs(grid)), 1.))     # into the bands' neighbourhood
        levels = int(rng.integers(1, 7))
        t = rng.uniform(150., 350., levels)
        p = 10.**rng.uniform(0., 5.2, levels)
        grid_handle = engine.load_grid(grid)
        got = engine.xsec_compute(handle, grid_handle, grid.size, t, p)
        engine.free_grid(grid_handle)
        expect = np.stack([xsec_oracle.absorption_coefficient(bands, grid, t[i], p[i])
                           for i in range(levels)])
        # The clipping rule rescales by a ratio of two band sums: agreement is to 1e-6 of the
        # value or 1e-12 of the band maximum where the fit hovers around zero.
        close(got, expect, f"seed {seed}: {len(bands)} band(s), {levels} level(s)", floor=1e-12)
    finally:
        engine.free_xsec(handle)


@pytest.mark.parametrize("seed", range(CASES))
def test_random_groups_of_continua(continua, continuum_oracle, seed):
    """lbl_continuum_compute_many on random groups (1-6 continua in random order, one of them
    possibly twice), random levels and grids -- numpy.arange ones too, whose wavenumbers the
    kernels form in registers --, writing or adding onto a block: bit for bit what one call per
    continuum leaves, and within the bar of the sum of the oracle's spectra."""
    from pylbl_amd.engine import DeviceSpectra
    rng = np.random.default_rng(11000 + seed)
    engine = default_engine(0)
    count = int(rng.integers(1, 7))
    owners = [OWNERS[i] for i in rng.permutation(len(OWNERS))[:count]]
    if rng.random() < 0.2:
        owners.append(owners[0])
    if rng.random() < 0.4:
        lower = float(np.round(rng.uniform(-100., 30000.), 2))
        step = float(rng.choice([0.001, 0.01, 0.05, 0.25, 1., 7.3]))
        grid = np.arange(lower, lower + step*int(10**rng.uniform(0., 4.8)), step)
        if grid.size < 2:
            grid = np.arange(lower, lower + 3*step, step)
    else:
        grid = random_grid(rng)
    levels = int(rng.integers(1, 10))
    t = rng.uniform(50., 1000., levels)
    p = 10.**rng.uniform(-2., 6.7, levels)
    vmr = {name: 10.**rng.uniform(-9., -0.3, levels) for name in
           ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2")}
    members = [continua[owner] for owner in owners]
    adding = rng.random() < 0.5
    padded = grid.size + int(rng.integers(0, 40))
    one_by_one, together = DeviceSpectra(engine, levels, padded), DeviceSpectra(engine, levels, padded)
    if adding:
        for block in (one_by_one, together):
            continua["O3"].spectra_levels(t, p, vmr, grid, out=block)
    for i, continuum in enumerate(members):
        continuum.spectra_levels(t, p, vmr, grid, out=one_by_one, accumulate=adding or i > 0,
                                 asynchronous=True)
    mt_ckd.spectra_levels_many(members, t, p, vmr, grid, together, accumulate=adding,
                               asynchronous=True)
    engine.synchronize()
    a, b = one_by_one.to_host()[:, :grid.size], together.to_host()[:, :grid.size]
    one_by_one.free()
    together.free()
    assert np.array_equal(a, b, equal_nan=True), f"seed {seed}: {owners}, {levels} level(s)"
    expect = sum(np.stack([continuum_oracle.continuum(owner).spectra(
        t[i], p[i], {k: v[i] for k, v in vmr.items()}, grid) for i in range(levels)])
        for owner in owners + (["O3"] if adding else []))
    close(b, expect, f"seed {seed}: group {owners} adding={adding}", floor=1e-12)
