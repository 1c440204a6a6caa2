"""Loads the golden vectors of tests/golden/ (made by tests/golden/make_golden.py from the
reference's own compiled C)."""
from collections import namedtuple
from pathlib import Path

import numpy as np

from pylbl_amd.database import LINE_COLUMNS, LineTable

GOLDEN = Path(__file__).resolve().parent / "golden"
ABSORPTION_GROUPS = ("h2o40", "h2o40_swapped", "clipping", "break_first", "break_middle",
                     "iso_ten", "co2_band", "h2o_nir")

Case = namedtuple("Case", ["group", "index", "temperature", "pressure", "vmr", "v0", "vn",
                           "n_per_v", "cut_off", "remove_pedestal", "k", "rc"])


def load_group(name):
    """Returns (LineTable, [Case...]) for one fixture file."""
    data = np.load(GOLDEN / f"{name}.npz")
    table = LineTable(
        formula=str(data["formula"]), molecule_id=int(data["table_molecule_id"]),
        local_iso_id=data["table_local_iso_id"].astype(np.int32),
        isoid=data["table_isoid"], mass=data["table_mass"],
        tips_temperature=data["table_tips_temperature"], tips_data=data["table_tips_data"],
        **{x: data["table_" + x] for x in LINE_COLUMNS})
    cases = []
    for i in range(int(data["num_cases"])):
        a = data[f"case{i}_args"]
        cases.append(Case(name, i, float(a[0]), float(a[1]), float(a[2]), int(a[3]), int(a[4]),
                          int(a[5]), int(a[6]), bool(a[7]), data[f"case{i}_k"],
                          int(data[f"case{i}_rc"])))
    return table, cases


def all_absorption_cases():
    for group in ABSORPTION_GROUPS:
        table, cases = load_group(group)
        for case in cases:
            yield table, case


def load_voigt():
    data = np.load(GOLDEN / "voigt_profile.npz")
    out = []
    for i in range(int(data["num_cases"])):
        centre, alpha, gamma, strength = data[f"case{i}_args"]
        out.append((data[f"case{i}_grid"], centre, alpha, gamma, strength, data[f"case{i}_k"]))
    return out


def pedestal_tolerance(k_ref, n_per_v, cut_off, rel=1.e-6):
    """Per-point tolerance for spectra with the pedestal removed: values near window edges
    are differences of near-equal numbers, so the bound is rel x the largest |k_ref| within
    one line window of the point (SURVEY.md section 8c "Parity metric")."""
    # Every line window that holds point i lies inside [i - half, i + half]: the exact sliding
    # maximum over that range (round 6; rounds 1-5 took block maxima, up to twice as wide).
    from scipy.ndimage import maximum_filter1d
    half = (2*cut_off + 1)*n_per_v
    return rel*maximum_filter1d(np.abs(k_ref), size=2*half + 1, mode="constant", cval=0.)
