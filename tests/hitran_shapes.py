"""Line tables with HITRAN-shaped values (see tests/test_gpu_fuzz_hitran.py for what that means and
which lines of the reference's ingest it follows): shared by the seeded fuzz and by the whole-grid
comparison in tests/test_gpu_baseline_configs.py, whose oracle workers rebuild the table from a
recipe (tests/oracle_farm.py)."""
import numpy as np

from pylbl_amd.database import LineTable

MASSES = np.asarray([43.98983, 44.993185, 45.994076, 44.994045, 46.997431, 45.97787,
                     47.998322, 46.998291, 45.998262, 49.001675, 48.001646, 47.001618])


def hitran_shaped_table(rng, lo, hi, n):
    """Twelve isotopologues (local ids 1-9, 0 for the tenth, 11, 12), partition sums 1-5000 K,
    and columns drawn from the corners real tables have."""
    # Q_iso(T) = q0_iso (T/296)^1.5 rounded to float32, 1 ... 5000 K (pyLBL/webapi/tips_api.py:86-87)
    temperature = np.arange(1., 5001., 1.)
    q0 = 286.1*np.asarray([1., 1.07, 2.01, 0.53, 1.3, 0.9, 1.6, 0.7, 2.2, 0.45, 3.1, 1.9])
    data = (q0[:, None]*(temperature[None, :]/296.)**1.5).astype(np.float32).astype(np.float64)
    nu = np.sort(rng.uniform(lo, hi, n))
    # duplicated positions (HITRAN lists blended lines at the same wavenumber)
    twins = rng.choice(n, max(n//8, 1))
    nu[twins] = nu[rng.choice(n, twins.size)]
    # a few at the very bottom of the table (pure-rotation lines: 1.3e-4 cm-1 in H2O)
    if lo < 1.:
        low = rng.choice(n, max(n//20, 1))
        nu[low] = 10.**rng.uniform(-4., 0., low.size)
    nu = np.sort(nu)
    choice = rng.choice
    gamma_air = choice([0., 0., 0.005, 0.05, 0.1], n)*rng.uniform(0.5, 1.5, n)
    gamma_self = choice([0., 0., 0., 0.1, 0.4, 1.2], n)*rng.uniform(0.5, 1.5, n)
    both_zero = rng.random(n) < 0.15
    gamma_air[both_zero] = 0.
    gamma_self[both_zero] = 0.
    local = choice([1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 11, 12], n).astype(np.int32)
    return LineTable(
        formula="CO2", molecule_id=2, nu=nu,
        sw=10.**rng.uniform(-45., -16., n),
        gamma_air=gamma_air, gamma_self=gamma_self,
        n_air=choice([0., -0.5, -0.25, 0.5, 0.75], n)*rng.uniform(0.9, 1.1, n),
        elower=choice([-1., 0., 0., 100., 3000., 12000.], n),
        delta_air=choice([0., 0., -0.05, 0.05, -0.003], n),
        local_iso_id=local, isoid=np.asarray([1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 11, 12]),
        mass=MASSES, tips_temperature=temperature, tips_data=data)
