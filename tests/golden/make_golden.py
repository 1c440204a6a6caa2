"""Generates the golden vectors in this directory from the REFERENCE's own C.

Run from the repo root, in the build container only (needs /root/reference):

    make -C oracle            # compiles the reference's four C files into oracle/_ref/
    python tests/golden/make_golden.py

For every case a synthetic SQLite database with the reference's schema is written to a
temporary directory, the compiled reference ``absorption()`` (pyLBL/c_lib/absorption.c:19)
is called through ctypes exactly as pyLBL/c_lib/gas_optics.py:61-91 does, and inputs +
outputs are stored as ``*.npz``.  ``voigt()`` (pyLBL/c_lib/voigt.c:4) is driven directly
for the profile-only vectors.  Only data is stored: no reference source or binary.

The case list follows SURVEY.md section 8c ("Golden-vector set to generate").
"""
from pathlib import Path
import sys
import tempfile

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import oracle  # noqa: E402
from pylbl_amd import synthetic  # noqa: E402
from pylbl_amd.database import LINE_COLUMNS, LineTable, write_database  # noqa: E402

OUT = Path(__file__).resolve().parent
TIPS = (150, 400)  # compact partition-function table keeps the fixtures small
ARG_NAMES = ("temperature", "pressure", "vmr", "v0", "vn", "n_per_v", "cut_off",
             "remove_pedestal")


def table_arrays(table, prefix="table_"):
    out = {prefix + x: getattr(table, x) for x in LINE_COLUMNS}
    for x in ("local_iso_id", "isoid", "mass", "tips_temperature", "tips_data"):
        out[prefix + x] = getattr(table, x)
    out[prefix + "molecule_id"] = np.asarray(table.molecule_id)
    return out


def run_cases(name, table, cases, tmp, with_tips=None, formula=None):
    """cases: list of (T, P, x, v0, vn, n_per_v, cut_off, remove_pedestal)."""
    db = write_database(Path(tmp) / f"{name}.db", [table], with_tips=with_tips)
    payload = table_arrays(table)
    payload["formula"] = np.asarray(table.formula)
    payload["query_formula"] = np.asarray(formula or table.formula)
    payload["num_cases"] = np.asarray(len(cases))
    for i, c in enumerate(cases):
        t, p, x, v0, vn, npv, cut, ped = c
        rc, k = oracle.absorption_reference(db, formula or table.formula, t, p, x, v0, vn,
                                            npv, cut_off=cut, remove_pedestal=ped)
        payload[f"case{i}_args"] = np.asarray([t, p, x, v0, vn, npv, cut, int(ped)],
                                              dtype=np.float64)
        payload[f"case{i}_k"] = k
        payload[f"case{i}_rc"] = np.asarray(rc)
        print(f"  {name}[{i}] rc={rc} n={k.size} max={k.max() if k.size else 0:.6e} "
              f"nonzero={np.count_nonzero(k)}")
    np.savez_compressed(OUT / f"{name}.npz", **payload)


def levels(formula):
    atmos = synthetic.fixture_atmosphere()
    return [(atmos.t[i], atmos.p[i], atmos.vmr[formula][i]) for i in range(atmos.t.size)]


def main():
    if not oracle.have_reference():
        raise SystemExit("oracle/_ref/libabsorption_ref.so missing: run `make -C oracle`.")
    with tempfile.TemporaryDirectory() as tmp:
        # (2) 40-line table, pedestal off/on, n_per_v 1/10/100, all four fixture levels.
        t40 = synthetic.line_table("H2O", 1., 127., num_lines=40, seed=11, tips_range=TIPS)
        cases = []
        for npv in (1, 10, 100):
            for ped in (False, True):
                for (t, p, x) in levels("H2O"):
                    cases.append((t, p, x, 1, 101, npv, 25, ped))
        run_cases("h2o40", t40, cases, tmp)

        # (2) same lines, two rows swapped: the cumulative pedestal depends on row order.
        order = np.arange(40)
        order[[7, 23]] = order[[23, 7]]
        t40s = t40.subset(order)
        run_cases("h2o40_swapped", t40s,
                  [(288.99, 98388., 6.637074e-3, 1, 101, 10, 25, ped) for ped in (False, True)],
                  tmp)

        # (3) clipping: first / last cm-1, window right of the grid, shift across an integer,
        #     window wholly left of the grid, exact-integer centre.
        t = synthetic.line_table("CO2", 1., 2., num_lines=8, seed=5, tips_range=TIPS)
        t.nu = np.asarray([1.3, 12.0, 50.0005, 99.5, 100.999, 124.2, 125.9, 126.5])
        t.delta_air = np.asarray([0., 0., -0.01, 0.002, 0.002, -0.009, 0.002, -0.01])
        cases = [(288.99, 98388., 3.6e-4, 1, 101, npv, 25, ped)
                 for npv in (1, 10, 100) for ped in (False, True)]
        cases += [(250., 101325., 3.6e-4, 1, 101, 10, 25, False),
                  (296., 101325., 3.6e-4, 1, 101, 10, 25, False),   # (6) integer-valued T
                  (288.99, 98388., 3.6e-4, 1, 101, 10, 10, False),  # other cut-off
                  (288.99, 98388., 3.6e-4, 1, 101, 10, 10, True),
                  (288.99, 98388., 3.6e-4, 40, 60, 10, 25, False),  # first row < v0-26: zeros
                  (288.99, 98388., 3.6e-4, 1, 21, 10, 25, False),   # windows wider than grid
                  (288.99, 98388., 3.6e-4, 1, 21, 10, 25, True)]
        run_cases("clipping", t, cases, tmp)

        # (4) range "break": first row below v0-26 -> all zeros; out-of-range row in the
        #     middle -> later rows dropped (absorption.c:80-83).
        t = synthetic.line_table("CO2", 660., 700., num_lines=12, seed=6, tips_range=TIPS)
        t.nu[0] = 600.5
        run_cases("break_first", t, [(288.99, 98388., 3.6e-4, 650, 700, 10, 25, False)], tmp)
        t = synthetic.line_table("CO2", 660., 700., num_lines=12, seed=6, tips_range=TIPS)
        t.nu[6] = 900.
        run_cases("break_middle", t,
                  [(288.99, 98388., 3.6e-4, 650, 700, 10, 25, ped) for ped in (False, True)], tmp)

        # (5) isotopologue id 0 -> 10, ten TIPS rows, TIPS table starting at 100 K.
        t = synthetic.line_table("O3", 1., 60., num_lines=30, seed=7, tips_range=TIPS)
        temperature = np.arange(100., 501., 1.)
        q0 = 3000.*(1. + 0.1*np.arange(10))
        t.tips_temperature = temperature
        t.tips_data = (q0[:, None]*(temperature[None, :]/296.)**1.5
                       ).astype(np.float32).astype(np.float64)
        t.isoid = np.asarray([1, 2, 3, 4, 5, 6, 7, 8, 9, 0])
        t.mass = 47.98 + 0.5*np.arange(10)
        t.local_iso_id = (np.arange(30) % 10).astype(np.int32)  # includes 0
        run_cases("iso_ten", t, [(288.99, 98388., 6.9e-8, 1, 61, 10, 25, False),
                                 (203.37, 11419., 2.6e-7, 1, 61, 10, 25, True)], tmp)

        # (7) denser CO2-like band incl. tiny-y (P = 0.5 Pa) and Lorentz-only (y >= 70.55).
        t = synthetic.line_table("CO2", 500., 587., num_lines=1500, seed=8, tips_range=TIPS)
        cases = []
        for ped in (False, True):
            for (tt, p, x) in levels("CO2") + [(220., 0.5, 3.6e-4), (300., 5.e6, 3.6e-4)]:
                cases.append((tt, p, x, 500, 561, 100, 25, ped))
        run_cases("co2_band", t, cases, tmp)

        # (7) H2O-like at higher wavenumber: Doppler-dominated cores at low pressure.
        t = synthetic.line_table("H2O", 3600., 3687., num_lines=300, seed=9, tips_range=TIPS)
        cases = [(tt, p, x, 3600, 3661, 100, 25, ped) for ped in (False, True)
                 for (tt, p, x) in levels("H2O")]
        run_cases("h2o_nir", t, cases, tmp)

        # (8) molecule without TIPS rows -> rc 0 and zeros; unknown alias -> rc 1.
        t = synthetic.line_table("N2O", 1., 60., num_lines=10, seed=10, tips_range=TIPS)
        run_cases("no_tips", t, [(288.99, 98388., 3.2e-7, 1, 61, 10, 25, False)], tmp,
                  with_tips=set())
        run_cases("unknown_alias", t, [(288.99, 98388., 3.2e-7, 1, 61, 10, 25, False)], tmp,
                  formula="XYZ")

    # (1) voigt() alone: every region boundary +- a few ulp and a log sweep, many y.
    alpha = 1.
    repwid = np.sqrt(np.log(2.))/alpha
    ys = [1.e-8, 1.e-6, 1.0000001e-6, 1.e-3, 0.1, 1., 5., 8.424999, 8.425, 8.4250001, 20.,
          70.549999, 70.55, 70.5500001, 200.]
    payload = {"num_cases": np.asarray(len(ys))}
    for i, y in enumerate(ys):
        gamma = y/repwid
        yy = repwid*gamma
        lims = [np.sqrt(15100. + yy*(40. - yy*3.6)) if yy < 70.55 else 0.,
                np.sqrt(max(164. - yy*(4.3 + yy*1.8), 0.)), 6.8 - yy, 2.4*yy, 18.1*yy + 1.65]
        xs = [0.]
        for lim in lims:
            if lim > 0.:
                x = lim
                around = [x]
                lo = hi = x
                for _ in range(4):
                    lo = np.nextafter(lo, -np.inf)
                    hi = np.nextafter(hi, np.inf)
                    around += [lo, hi]
                xs += around + [-a for a in around]
        xs += list(np.logspace(-6, 5, 400)) + list(-np.logspace(-6, 5, 400))
        xs = np.sort(np.asarray(xs))
        centre = 1000.
        grid = centre + xs/repwid
        k = oracle.voigt_reference(grid, 0, grid.size - 1, centre, alpha, gamma, 1.e-20)
        payload[f"case{i}_grid"] = grid
        payload[f"case{i}_args"] = np.asarray([centre, alpha, gamma, 1.e-20])
        payload[f"case{i}_k"] = k
        print(f"  voigt[{i}] y={yy:.9g} n={grid.size} max={k.max():.6e}")
    np.savez_compressed(OUT / "voigt_profile.npz", **payload)


if __name__ == "__main__":
    main()
