"""Exceptions raised by the spectral-database read side.

Same names and the same base class as the reference (pyLBL/database.py:489-506: all derive
from BaseException, which is what lets Spectroscopy's MoleculeCache skip a molecule silently,
pyLBL/spectroscopy.py:53-57,66-69).

When the reference itself is importable, each class here additionally derives from the
reference's class of the same name: the reference's ``MoleculeCache`` catches only its own
classes, so an object of this package used under the reference's ``Spectroscopy`` (a
``pylbl_amd.Database`` as its ``database`` argument, ``pylbl_amd.Gas`` or ``CrossSection`` as
its back ends) must raise something those handlers recognise.  ``PYLBL_AMD_STANDALONE=1``
skips the lookup (pyLBL's import pulls in xarray, SQLAlchemy and its own C library).
"""
import os
import sys

NAMES = ("AliasNotFoundError", "TipsDataNotFoundError", "IsotopologuesNotFoundError",
         "TransitionsNotFoundError", "CrossSectionNotFoundError")


def _reference_classes():
    """The reference's five exception classes, or {} when pyLBL cannot be imported."""
    if os.environ.get("PYLBL_AMD_STANDALONE", "") not in ("", "0"):
        return {}
    module = sys.modules.get("pyLBL.database")
    if module is None:
        try:
            from importlib import import_module
            from importlib.util import find_spec
            if find_spec("pyLBL") is None:
                return {}
            module = import_module("pyLBL.database")
        except Exception:      # a pyLBL without its extension built, without xarray, ...
            return {}
    found = {}
    for name in NAMES:
        cls = getattr(module, name, None)
        if isinstance(cls, type) and issubclass(cls, BaseException):
            found[name] = cls
    return found


_REFERENCE = _reference_classes()


def _define(name):
    base = _REFERENCE.get(name, BaseException)
    return type(name, (base,), {"__module__": __name__, "__doc__":
                                f"Same meaning as pyLBL.database.{name} (pyLBL/database.py:489-506)."})


AliasNotFoundError = _define("AliasNotFoundError")
TipsDataNotFoundError = _define("TipsDataNotFoundError")
IsotopologuesNotFoundError = _define("IsotopologuesNotFoundError")
TransitionsNotFoundError = _define("TransitionsNotFoundError")
CrossSectionNotFoundError = _define("CrossSectionNotFoundError")


def kind(error):
    """Which of the five database conditions `error` stands for -- this package's class or the
    reference's class of the same name (raised by a pyLBL.database.Database handed to one of
    this package's back ends) -- or None."""
    for cls in type(error).__mro__:
        if cls.__name__ in NAMES and (cls.__module__ == __name__
                                      or cls.__module__.split(".")[-1] == "database"):
            return cls.__name__
    return None


class EngineError(RuntimeError):
    """A non-zero status came back over the C-ABI (message from lbl_last_error)."""
