"""Deterministic synthetic line tables, TIPS tables and atmospheres.

No HITRAN database exists offline (the reference downloads it with an API key,
pyLBL/database.py:148-210), so benchmarks and parity tests run on tables with
HITRAN-like statistics, generated here from fixed seeds.
A real ``*.db`` drops in through ``pylbl_amd.database.Database`` unchanged.
"""
from collections import namedtuple

import numpy as np

from .database import LineTable


# HITRAN molecule ids of the eight README molecules (README.rst:33-48).
MOLECULE_IDS = {"H2O": 1, "CO2": 2, "O3": 3, "N2O": 4, "CO": 5, "CH4": 6, "O2": 7, "N2": 22}

# Approximate HITRAN2020 line counts inside 1-5000 cm-1.
LINES_1_5000 = {"H2O": 110_000, "CO2": 400_000, "O3": 400_000, "N2O": 160_000, "CO": 6_000,
                "CH4": 300_000, "O2": 15_000, "N2": 1_000}

# Masses [g mol-1] of the four most abundant isotopologues.
MASSES = {
    "H2O": (18.010565, 20.014811, 19.01478, 19.01674),
    "CO2": (43.98983, 44.993185, 45.994076, 44.994045),
    "O3": (47.984745, 49.988991, 49.988991, 48.98896),
    "N2O": (44.001062, 44.998096, 44.998096, 46.005308),
    "CO": (27.994915, 28.99827, 29.999161, 28.99913),
    "CH4": (16.0313, 17.034655, 17.037475, 18.04083),
    "O2": (31.98983, 33.994076, 32.994045, 35.998322),
    "N2": (28.006148, 29.003182, 30.000216, 29.003182),
}

# Q(296 K) scale per isotopologue; only ratios Q(296)/Q(T) matter to the lines path.
_Q296 = {"H2O": 174.6, "CO2": 286.1, "O3": 3483.7, "N2O": 4984.9, "CO": 107.4, "CH4": 590.5,
         "O2": 215.7, "N2": 467.1}

Atmos = namedtuple("Atmos", ["p", "t", "vmr"])


def tips_table(formula, num_iso=4, t_max=5000, t_min=1):
    """Q_iso(T) = q0_iso (T/296)^1.5 rounded to float32 (the reference's TIPS parser
    stores float32, pyLBL/webapi/tips_api.py:86-87), T = t_min..t_max K in 1-K steps."""
    temperature = np.arange(float(t_min), t_max + 1., 1.)
    q0 = _Q296.get(formula, 300.)*np.asarray([1., 1.07, 2.01, 0.53, 1.3, 0.9, 1.6, 0.7][:num_iso])
    data = q0[:, None]*(temperature[None, :]/296.)**1.5
    return temperature, data.astype(np.float32).astype(np.float64)


def line_table(formula, v_lo=1., v_hi=5000., num_lines=None, scale=1., seed=None,
               num_iso=4, tips_range=(1, 5000)):
    """Synthetic transitions for one molecule, ascending in wavenumber.

    Args:
        formula: Chemical formula (keys of MOLECULE_IDS, or anything else with seed given).
        v_lo, v_hi: Line centres are drawn uniformly from [v_lo, v_hi).
        num_lines: Number of transitions; default scales LINES_1_5000 by the range.
        scale: Extra factor on the default count.
        seed: RNG seed; default 1000 + HITRAN molecule id.
        tips_range: (t_min, t_max) of the 1-K partition-function table.
    """
    id = MOLECULE_IDS.get(formula, 99)
    rng = np.random.default_rng(1000 + id if seed is None else seed)
    if num_lines is None:
        num_lines = LINES_1_5000.get(formula, 10_000)*(v_hi - v_lo)/4999.*scale
    n = max(int(round(num_lines)), 1)
    nu = np.sort(rng.uniform(v_lo, v_hi, n))
    temperature, data = tips_table(formula, num_iso, tips_range[1], tips_range[0])
    masses = np.asarray(MASSES.get(formula, (30., 31., 32., 33.))[:num_iso])
    return LineTable(
        formula=formula, molecule_id=id, nu=nu,
        sw=10.**rng.uniform(-30., -19., n),
        gamma_air=rng.uniform(0.03, 0.12, n),
        gamma_self=rng.uniform(0.05, 0.50, n),
        n_air=rng.uniform(0.40, 0.85, n),
        elower=rng.uniform(0., 5000., n),
        delta_air=rng.uniform(-0.010, 0.002, n),
        local_iso_id=rng.choice(np.arange(1, 5), size=n,
                                p=(0.90, 0.05, 0.03, 0.02)).astype(np.int32),
        isoid=np.arange(1, num_iso + 1), mass=masses,
        tips_temperature=temperature, tips_data=data)


def banded_line_table(formula, v_lo=1., v_hi=5000., num_lines=100_000, bands=6, seed=None,
                      inside=False):
    """Like line_table but with line centres clustered in Gaussian bands, the way real
    vibration-rotation bands cluster; exercises load balance of the tile schedule.

    Lines that fall outside [v_lo, v_hi) are moved onto its ends (thousands of them on one
    wavenumber where a band sits near an end: a stress of its own for the pedestal chain).
    inside=True keeps the band centres 600 cm-1 away from the ends and drops such lines instead:
    dense the way a band centre is dense, and nothing else."""
    table = line_table(formula, v_lo, v_hi, num_lines, seed=seed)
    rng = np.random.default_rng((seed or 0) + 77)
    margin = 600. if inside else 100.
    centres = rng.uniform(v_lo + margin, v_hi - margin, bands)
    widths = rng.uniform(15., 80., bands)
    which = rng.integers(0, bands, table.num_lines)
    nu = rng.normal(centres[which], widths[which])
    if inside:
        nu = nu[(nu > v_lo) & (nu < v_hi)]
        table = table.subset(np.arange(table.num_lines) < nu.size)
    else:
        nu = np.clip(nu, v_lo, np.nextafter(v_hi, 0.))
    table.nu = np.sort(nu)
    return table


def surface_level(formulae=None):
    """The last level of the reference's 4-level test atmosphere (tests/conftest.py:61-77,
    103-112): T, P and volume mixing ratios for the eight README molecules."""
    atmos = fixture_atmosphere()
    vmr = {k: v[-1:] for k, v in atmos.vmr.items() if formulae is None or k in formulae}
    return Atmos(p=atmos.p[-1:], t=atmos.t[-1:], vmr=vmr)


def fixture_atmosphere():
    """The reference's 4-level test atmosphere (tests/conftest.py:53-78)."""
    pressure = np.asarray([117., 1032., 11419., 98388.])
    temperature = np.asarray([269.01, 227.74, 203.37, 288.99])
    vmr = {
        "H2O": np.asarray([5.244536e-06, 4.763972e-06, 3.039952e-06, 6.637074e-03]),
        "CO2": np.asarray([0.00036, 0.00036, 0.00036, 0.00035999]),
        "O3": np.asarray([2.936688e-06, 7.415223e-06, 2.609510e-07, 6.859128e-08]),
        "N2O": np.asarray([1.050928e-08, 1.319584e-07, 2.895416e-07, 3.199949e-07]),
        "CH4": np.asarray([2.947482e-07, 8.817705e-07, 1.588336e-06, 1.700002e-06]),
        "CO": np.asarray([3.621464e-08, 1.761450e-08, 3.315927e-08, 1.482969e-07]),
        "O2": np.asarray([0.209, 0.209, 0.2090003, 0.208996]),
        "N2": np.asarray([0.78, 0.78, 0.78, 0.78]),
    }
    return Atmos(p=pressure, t=temperature, vmr=vmr)


def standard_atmosphere(num_levels):
    """Build-owned "standard atmosphere" (not in the reference): pressure log-spaced
    101325 -> 10 Pa, temperature from US-Std-1976 lapse segments, fixed formulae for
    the eight gases; no RNG."""
    p = np.logspace(np.log10(101325.), np.log10(10.), num_levels)
    z = -7.0*np.log(p/101325.)  # Scale-height altitude [km].
    t = np.where(z < 11., 288.15 - 6.5*z,
        np.where(z < 20., 216.65,
        np.where(z < 32., 216.65 + 1.0*(z - 20.),
        np.where(z < 47., 228.65 + 2.8*(z - 32.),
        np.where(z < 51., 270.65,
        np.where(z < 71., 270.65 - 2.8*(z - 51.), 214.65 - 2.0*(z - 71.)))))))
    t = np.maximum(t, 180.)
    vmr = {
        "H2O": np.maximum(6.6e-3*np.exp(-z/2.0), 3.e-6),
        "CO2": np.full(num_levels, 4.e-4),
        "O3": 8.e-6*np.exp(-0.5*((z - 32.)/8.)**2) + 3.e-8,
        "N2O": np.full(num_levels, 3.2e-7),
        "CO": np.full(num_levels, 1.5e-7),
        "CH4": np.full(num_levels, 1.7e-6),
        "O2": np.full(num_levels, 0.209),
        "N2": np.full(num_levels, 0.781),
    }
    return Atmos(p=p, t=t, vmr=vmr)


def grid_arguments(grid):
    """(v0, vn, n_per_v) from a wavenumber grid, as pyLBL/c_lib/gas_optics.py:61-63."""
    v0 = int(round(grid[0]))
    vn = int(round(grid[-1]) + 1)
    n_per_v = int(round(1./(grid[1] - grid[0])))
    return v0, vn, n_per_v


def cross_section_bands(seed=0, ranges=((600., 900.), (1050., 1250.)), spacing=0.05):
    """Synthetic ARTS-crossfit-like bands: [(frequency [Hz], coefficients [4, nfreq]), ...].

    No coefficient file exists offline (the reference downloads them,
    pyLBL/arts_crossfit/webapi.py).  Frequencies are unevenly spaced; the fit p00 + p10 T +
    p01 P + p20 T^2 has a smooth two-peak shape of ~1e-22 m2 with small T/P terms and noise
    that drives it negative in the wings, so the clipping rule
    (xsec_aux_functions.py:104-119) is exercised.
    """
    rng = np.random.default_rng(4000 + seed)
    bands = []
    for lower, upper in ranges:
        size = int((upper - lower)/spacing)
        steps = rng.uniform(0.5, 1.5, size)
        wavenumber = lower + (upper - lower)*np.cumsum(steps)/np.sum(steps)
        f = (wavenumber - lower)/(upper - lower)
        shape = 1e-22*(np.exp(-((f - 0.4)/0.1)**2) + 0.5*np.exp(-((f - 0.7)/0.05)**2))
        coefficients = np.zeros((4, size))
        coefficients[0] = shape + 2e-24*rng.standard_normal(size)
        coefficients[1] = 1e-26*rng.standard_normal(size)
        coefficients[2] = 1e-29*rng.standard_normal(size)
        coefficients[3] = 1e-29*rng.standard_normal(size)
        bands.append((wavenumber*299792458.0*100, coefficients))
    return bands
