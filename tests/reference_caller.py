"""Test infrastructure: the reference's side of the drop-in boundary, without the reference.

pyLBL cannot be imported where the tests run (its C extension is not built, xarray is not
installed, and /root/reference does not exist on the GPU box), so two stand-ins reproduce
what a back end sees when the *reference* drives it:

* ``ReferenceDatabase`` -- an object with exactly the attributes a
  ``pyLBL.database.Database`` offers a back end (``.path``, ``.gas()``, ``.tips()``,
  ``.molecules()``, ``.arts_crossfit()``; pyLBL/database.py:146,340-415) and nothing else,
  serving the values the reference's own class returned for tests/golden/refdb.db
  (tests/golden/refdb.npz, made by tests/golden/make_refdb.py) and raising exception classes
  that, like the reference's, are *not* this package's;
* ``replay_compute_absorption`` -- the sequence of calls ``Spectroscopy.compute_absorption``
  makes on its back ends (pyLBL/spectroscopy.py:53-69 and :163-205), with numpy arrays in the
  place of xarray objects.
"""
import os
from pathlib import Path
from types import SimpleNamespace

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
FIELDS = ("nu", "sw", "gamma_air", "gamma_self", "n_air", "delta_air", "elower",
          "local_iso_id", "molecule_id", "global_iso_id")
kb = 1.38064852e-23     # pyLBL/spectroscopy.py:15


def _foreign_error(name):
    """An exception class shaped like pyLBL/database.py:489-506: BaseException subclass,
    defined in a module called ``...database``, unrelated to pylbl_amd.errors."""
    return type(name, (BaseException,), {"__module__": "pyLBL.database"})


AliasNotFoundError = _foreign_error("AliasNotFoundError")
TipsDataNotFoundError = _foreign_error("TipsDataNotFoundError")
IsotopologuesNotFoundError = _foreign_error("IsotopologuesNotFoundError")
TransitionsNotFoundError = _foreign_error("TransitionsNotFoundError")
CrossSectionNotFoundError = _foreign_error("CrossSectionNotFoundError")
_BY_NAME = {x.__name__: x for x in (AliasNotFoundError, TipsDataNotFoundError,
                                    IsotopologuesNotFoundError, TransitionsNotFoundError,
                                    CrossSectionNotFoundError)}


class _PartitionFunction(object):
    """Attributes of pyLBL/tips.py:9-24."""
    def __init__(self, molecule, temperature, data):
        self.molecule, self.temperature, self.data = molecule, temperature, data


class ReferenceDatabase(object):
    """Only ``.path / .gas / .tips / .molecules / .arts_crossfit`` (see module docstring).

    Args:
        path: what ``.path`` reports: the fixture file (a back end may read it), or a name
              that is not a file, which leaves a back end the query helpers alone.
        cross_sections: formula -> coefficient file, overriding the recorded paths.
    """
    __slots__ = ("path", "_data", "_raised", "_cross_sections", "calls")

    def __init__(self, path=None, cross_sections=None):
        self.path = str(GOLDEN / "refdb.db") if path is None else path
        with np.load(GOLDEN / "refdb.npz") as archive:
            self._data = {k: archive[k] for k in archive.files}
        self._raised = dict(zip(self._data["raised_keys"].tolist(),
                                self._data["raised_values"].tolist()))
        self._cross_sections = dict(cross_sections or {})
        self.calls = []

    def _fail(self, method, name):
        known = set(x.lower() for x in self._data["molecules"].tolist()) | \
            set(self._data["molecules"].tolist())
        if name not in known:
            raise AliasNotFoundError(f"{name} not found in database.")
        error = self._raised.get(f"{method}:{self._formula(name)}")
        if error is not None:
            raise _BY_NAME[error](f"{method}({name})")

    def _formula(self, name):
        for formula in self._data["molecules"].tolist():
            if name in (formula, formula.lower()):
                return formula
        return name

    def molecules(self):
        self.calls.append(("molecules",))
        return self._data["molecules"].tolist()

    def gas(self, name):
        self.calls.append(("gas", name))
        self._fail("gas", name)
        f = self._formula(name)
        columns = {x: self._data[f"{f}_{x}"].tolist() for x in FIELDS}
        rows = [SimpleNamespace(**{x: columns[x][i] for x in FIELDS})
                for i in range(len(columns["nu"]))]
        partition = _PartitionFunction(name, self._data[f"{f}_q_temperature"],
                                       self._data[f"{f}_q_data"])
        return str(self._data[f"{f}_formula"]), self._data[f"{f}_mass"].tolist(), rows, partition

    def tips(self, name):
        self.calls.append(("tips", name))
        self._fail("tips", name)
        f = self._formula(name)
        return self._data[f"{f}_tips_temperature"], self._data[f"{f}_tips_data"]

    def arts_crossfit(self, name):
        self.calls.append(("arts_crossfit", name))
        f = self._formula(name)
        if f in self._cross_sections:
            return self._cross_sections[f]
        self._fail("arts_crossfit", name)
        return str(self._data[f"{f}_arts_crossfit"])

    def recorded(self, formula, field):
        return self._data[f"{formula}_{field}"]


def number_density(temperature, pressure, volume_mixing_ratio):
    return pressure*volume_mixing_ratio/(kb*temperature)


def replay_molecule_cache(name, lines_database, lines_engine, continua_engine,
                          cross_sections_engine):
    """pyLBL/spectroscopy.py:53-69: positional constructor calls, the reference's own exception
    classes in the handlers (here: the stand-ins above)."""
    try:
        gas = lines_engine(lines_database, name)
    except (AliasNotFoundError, IsotopologuesNotFoundError, TipsDataNotFoundError,
            TransitionsNotFoundError):
        gas = None
    names = [name + "Foreign", name + "Self"] if name == "H2O" else [name]
    try:
        gas_continua = [continua_engine[x]() for x in names]
    except KeyError:
        gas_continua = None
    try:
        cross_section = cross_sections_engine(name, lines_database.arts_crossfit(name))
    except (AliasNotFoundError, CrossSectionNotFoundError):
        cross_section = None
    return SimpleNamespace(gas=gas, gas_continua=gas_continua, cross_section=cross_section)


def replay_compute_absorption(atmosphere, grid, lines_database, lines_engine, continua_engine,
                              cross_sections_engine, remove_pedestal=None,
                              continua_backend="mt_ckd"):
    """pyLBL/spectroscopy.py:163-205 for `atmosphere` = (p, t, {formula: vmr}) arrays of one
    shape: {"<formula>_absorption": float64[*t.shape, 3, grid.size]} (output format "all")."""
    pressure, temperature, gases = atmosphere
    pressure, temperature = np.asarray(pressure, float), np.asarray(temperature, float)
    if remove_pedestal is None:
        remove_pedestal = continua_backend == "mt_ckd"
    beta, cache = {}, {}
    for name, mole_fraction in gases.items():
        varname = "{}_absorption".format(name)
        beta[varname] = np.zeros(list(temperature.shape) + [3, grid.size])
        data = cache.get(name)
        if data is None:
            data = cache[name] = replay_molecule_cache(
                name, lines_database, lines_engine, continua_engine, cross_sections_engine)
        for i in range(temperature.size):
            vmr = {x: np.asarray(y).flat[i] for x, y in gases.items()}
            t, p, x = temperature.flat[i], pressure.flat[i], np.asarray(mole_fraction).flat[i]
            n = number_density(t, p, x)
            j = list(np.unravel_index(i, temperature.shape))
            if data.gas is not None:
                k = data.gas.absorption_coefficient(t, p, x, grid,
                                                    remove_pedestal=remove_pedestal)
                beta[varname][tuple(j + [0, slice(None)])] = n*k[:grid.size]
            if data.gas_continua is not None:
                for continuum in data.gas_continua:
                    k = continuum.spectra(t, p, vmr, grid)
                    beta[varname][tuple(j + [1, slice(None)])] += k[:]
            if data.cross_section is not None:
                k = data.cross_section.absorption_coefficient(grid, t, p)
                beta[varname][tuple(j + [2, slice(None)])] = n*k[:]
    return beta, cache


def fixture_available():
    return os.path.exists(GOLDEN / "refdb.db") and os.path.exists(GOLDEN / "refdb.npz")
