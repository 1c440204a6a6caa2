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
    half = (2*cut_off + 1)*n_per_v
    n = k_ref.size
    a = np.abs(k_ref)
    # Sliding maximum over [i-half, i+half] by block maxima (exact enough and O(n)).
    block = max(half, 1)
    nb = -(-n//block)
    padded = np.zeros(nb*block)
    padded[:n] = a
    bmax = padded.reshape(nb, block).max(axis=1)
    ext = np.concatenate([[0.], bmax, [0.]])
    local = np.maximum(np.maximum(ext[:-2], ext[1:-1]), ext[2:])
    return rel*np.repeat(local, block)[:n]
