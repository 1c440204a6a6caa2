"""ctypes front-ends for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; nothing under pylbl_amd/ does.

Two checkers:

* ``port``  -- oracle/liblbl_oracle.so, our own C restatement (lbl_oracle.c).
  Takes the line table as arrays.
* ``ref``   -- oracle/_ref/libabsorption_ref.so, the reference's own C compiled
  by oracle/Makefile from /root/reference (binary only, never committed).  It is
  driven exactly as pyLBL/c_lib/gas_optics.py:61-91 drives it (same argtypes) and
  reads a SQLite file itself.
"""
from ctypes import CDLL, POINTER, c_char_p, c_double, c_int, c_long, c_longlong, c_void_p
from pathlib import Path
import subprocess

import numpy as np
from numpy.ctypeslib import ndpointer

HERE = Path(__file__).resolve().parent
PORT_LIB = HERE / "liblbl_oracle.so"
REF_LIB = HERE / "_ref" / "libabsorption_ref.so"
DERIVED_COLUMNS = ("centre", "alpha", "gamma", "strength", "first", "last", "status",
                   "pedestal")

_f64 = ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32 = ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(quiet=True):
    """Runs oracle/Makefile (port always; _ref only where /root/reference exists)."""
    subprocess.run(["make", "-C", str(HERE)], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_port = None
_ref = None


def port_library():
    global _port
    if _port is None:
        if not PORT_LIB.exists():
            build()
        lib = CDLL(str(PORT_LIB))
        lib.lbl_oracle_absorption.restype = c_longlong
        lib.lbl_oracle_absorption.argtypes = (
            [c_double]*3 + [c_int]*3 + [c_long] + [_f64]*7 + [_i32, _f64] +
            [c_int, _f64, _f64] + [c_int, c_int] + [_f64, c_void_p, c_void_p])
        lib.lbl_oracle_voigt.restype = None
        lib.lbl_oracle_voigt.argtypes = [_f64, c_int, c_int] + [c_double]*4 + [_f64, c_void_p]
        lib.lbl_oracle_tips.restype = c_double
        lib.lbl_oracle_tips.argtypes = [_f64, _f64, c_int, c_double, c_int]
        _port = lib
    return _port


def have_reference():
    return REF_LIB.exists()


def ref_library():
    global _ref
    if _ref is None:
        lib = CDLL(str(REF_LIB))
        # pyLBL/c_lib/gas_optics.py:68-73
        lib.absorption.restype = c_int
        lib.absorption.argtypes = [c_double]*3 + [c_int]*3 + [_f64] + [c_char_p]*2 + [c_int]*2
        # pyLBL/c_lib/voigt.h
        lib.voigt.restype = None
        lib.voigt.argtypes = [_f64, c_int, c_int] + [c_double]*4 + [_f64]
        _ref = lib
    return _ref


def mass_slots(isoids, masses):
    """mass[isoid-1] with isoid 0 stored at slot 9 (spectral_database.c:113-129)."""
    out = np.zeros(32, dtype=np.float64)
    for isoid, mass in zip(isoids, masses):
        slot = 10 if int(isoid) == 0 else int(isoid)
        out[slot - 1] = mass
    return out


def absorption_port(table, temperature, pressure, vmr, v0, vn, n_per_v, cut_off=25,
                    remove_pedestal=False, want_derived=False, want_regions=False):
    """Runs the C restatement on a line table (see pylbl_amd.synthetic.LineTable).

    Returns k (float64[(vn-v0)*n_per_v]) and a dict of extras.
    """
    lib = port_library()
    n = max((vn - v0)*n_per_v, 0)
    k = np.zeros(max(n, 1), dtype=np.float64)
    nl = table.nu.size
    derived = np.zeros((max(nl, 1), len(DERIVED_COLUMNS))) if want_derived else None
    regions = np.zeros(7, dtype=np.int64) if want_regions else None
    tips_t = np.ascontiguousarray(np.broadcast_to(table.tips_temperature,
                                                  table.tips_data.shape), dtype=np.float64)
    evals = lib.lbl_oracle_absorption(
        float(pressure), float(temperature), float(vmr), int(v0), int(vn), int(n_per_v),
        nl, *[np.ascontiguousarray(getattr(table, x), dtype=np.float64) for x in
              ("nu", "sw", "gamma_air", "gamma_self", "n_air", "elower", "delta_air")],
        np.ascontiguousarray(table.local_iso_id, dtype=np.int32),
        mass_slots(table.isoid, table.mass),
        int(table.tips_data.shape[1]), tips_t,
        np.ascontiguousarray(table.tips_data, dtype=np.float64),
        int(cut_off), 1 if remove_pedestal else 0, k,
        derived.ctypes.data if derived is not None else None,
        regions.ctypes.data if regions is not None else None)
    extras = {"evals": int(evals)}
    if derived is not None:
        extras["derived"] = derived[:nl]
    if regions is not None:
        extras["regions"] = regions
    return k[:n], extras


def absorption_reference(db_path, formula, temperature, pressure, vmr, v0, vn, n_per_v,
                         cut_off=25, remove_pedestal=False):
    """Calls the compiled reference exactly like pyLBL/c_lib/gas_optics.py:61-91."""
    lib = ref_library()
    k = np.zeros((vn - v0)*n_per_v, dtype=np.float64)
    rc = lib.absorption(float(pressure), float(temperature), float(vmr), int(v0), int(vn),
                        int(n_per_v), k, bytes(str(db_path), encoding="utf-8"),
                        bytes(formula, encoding="utf-8"), int(cut_off),
                        1 if remove_pedestal else 0)
    return rc, k


def voigt_port(grid, first, last, centre, alpha, gamma, strength, k=None, want_regions=False):
    lib = port_library()
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    k = np.zeros_like(grid) if k is None else k
    regions = np.zeros(7, dtype=np.int64) if want_regions else None
    lib.lbl_oracle_voigt(grid, int(first), int(last), float(centre), float(alpha),
                         float(gamma), float(strength), k,
                         regions.ctypes.data if regions is not None else None)
    return (k, regions) if want_regions else k


def voigt_reference(grid, first, last, centre, alpha, gamma, strength, k=None):
    lib = ref_library()
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    k = np.zeros_like(grid) if k is None else k
    lib.voigt(grid, int(first), int(last), float(centre), float(alpha), float(gamma),
              float(strength), k)
    return k


def tips_port(tips_temperature, tips_data, temperature, iso_row):
    lib = port_library()
    t = np.ascontiguousarray(np.broadcast_to(tips_temperature, tips_data.shape), np.float64)
    return lib.lbl_oracle_tips(t, np.ascontiguousarray(tips_data, np.float64),
                               int(tips_data.shape[1]), float(temperature), int(iso_row))
