"""Exceptions raised by the spectral-database read side.

Same names and the same base class as the reference (pyLBL/database.py:489-506:
all derive from BaseException, which is what lets Spectroscopy's MoleculeCache
skip a molecule silently, pyLBL/spectroscopy.py:53-57).
"""


class AliasNotFoundError(BaseException):
    pass


class TipsDataNotFoundError(BaseException):
    pass


class IsotopologuesNotFoundError(BaseException):
    pass


class TransitionsNotFoundError(BaseException):
    pass


class CrossSectionNotFoundError(BaseException):
    pass


class EngineError(RuntimeError):
    """A non-zero status came back over the C-ABI (message from lbl_last_error)."""
