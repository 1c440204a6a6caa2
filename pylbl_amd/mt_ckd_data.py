"""Tabulated MT-CKD continuum coefficients: where they come from and how they are read.

The reference ships them as ``pyLBL/mt_ckd/mt-ckd.nc`` (netCDF-4, i.e. HDF5) and reads one
variable per ``Spectrum`` object together with its ``wavenumber_lower_bound / upper_bound /
resolution`` attributes (pyLBL/mt_ckd/utils.py:116-143).  netCDF4-python is not part of this
image, so the file is read through the HDF5 C library with ctypes; a ``.npz`` file holding the
same variables (``<name>`` and ``<name>__grid`` = [lower, upper, resolution]) is accepted as
well and is what ``tests/golden/mt_ckd_bands.npz`` is.

Search order for the default table: ``$PYLBL_MT_CKD``, then the data file of an installed
``pyLBL`` package (located without importing it).
"""
import importlib.util
import os

import numpy as np

from . import hdf5_reader

# Every variable the 16 bands read (water_vapor.py:20-21,53-54; carbon_dioxide.py:21-31;
# nitrogen.py:16-17,37-38,61; oxygen.py:20,37,88,103,136; ozone.py:21,43-44,66).
VARIABLES = ("bs296", "bs260", "bfh2o", "xfac_rhu", "bfco2", "tdep_bandhead", "x_factor_co2",
             "ct_296", "sf_296", "ct_220", "sf_220", "xn2_272", "xn2_228", "a_h2o", "xn2",
             "o2_f", "o2_t", "o2_inf1", "o2_inf3", "o2_invis", "o2_infuv",
             "x_o3", "y_o3", "z_o3", "o3_hh0", "o3_hh1", "o3_hh2", "o3_huv")


class CoefficientTable(object):
    """One variable of the data set on its own uniform wavenumber grid."""
    def __init__(self, data, lower_bound, upper_bound, resolution):
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.lower_bound = float(lower_bound)
        self.upper_bound = float(upper_bound)
        self.resolution = float(resolution)

    def wavenumbers(self):
        """lower + i*resolution, the arithmetic of utils.py:136-143."""
        return self.lower_bound + np.arange(self.data.size)*self.resolution


def read_hdf5(path, names=VARIABLES):
    """Reads the named 1-D variables and their grid attributes from a netCDF-4 file."""
    tables = {}
    with hdf5_reader.File(path) as source:
        for name in names:
            bounds = [source.attribute(name, f"wavenumber_{key}")
                      for key in ("lower_bound", "upper_bound", "resolution")]
            tables[name] = CoefficientTable(source.array(name).ravel(), *bounds)
    return tables


def read_npz(path):
    tables = {}
    with np.load(path) as archive:
        for name in archive.files:
            if name.endswith("__grid"):
                continue
            lower, upper, resolution = archive[name + "__grid"]
            tables[name] = CoefficientTable(archive[name], lower, upper, resolution)
    return tables


def write_npz(path, tables):
    arrays = {}
    for name, table in tables.items():
        arrays[name] = table.data
        arrays[name + "__grid"] = np.asarray([table.lower_bound, table.upper_bound,
                                              table.resolution])
    np.savez_compressed(path, **arrays)


def default_path():
    path = os.environ.get("PYLBL_MT_CKD")
    if path:
        return path
    try:
        spec = importlib.util.find_spec("pyLBL")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        candidate = os.path.join(list(spec.submodule_search_locations)[0], "mt_ckd", "mt-ckd.nc")
        if os.path.isfile(candidate):
            return candidate
    raise FileNotFoundError("MT-CKD coefficients not found: set $PYLBL_MT_CKD to mt-ckd.nc (or "
                            "its .npz conversion) or install pyLBL, which ships the file.")


_cache = {}


def load(path=None):
    """Dictionary variable name -> CoefficientTable, cached per path."""
    path = os.path.abspath(path if path is not None else default_path())
    if path not in _cache:
        _cache[path] = read_npz(path) if path.endswith(".npz") else read_hdf5(path)
    return _cache[path]
