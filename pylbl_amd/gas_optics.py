"""The lines backend class: same constructor and method as the reference's ``Gas``
(pyLBL/c_lib/gas_optics.py:29-92), computing on an MI355X through pylbl_amd.engine.

Differences that do not change results: the line table is read from the database once, in
the constructor, and stays resident in HBM (the reference re-opens SQLite and re-reads every
transition on every call, absorption.c:44-86); ``absorption_coefficients`` (plural) is the
batched form over many levels that Spectroscopy uses.
"""
import warnings

import numpy as np

from . import errors
from .database import LineTable, line_table_of
from .engine import default_engine
from .synthetic import grid_arguments


class Gas(object):
    """One molecule's line-by-line absorption on an MI355X: the lines backend that
    ``molecular_lines["mi355x"]`` names (same shape as pyLBL/c_lib/gas_optics.py:29-44).

    Attributes:
        database: path of the SQLite file the lines came from (None for an in-memory table).
        formula: the molecule, e.g. "CO2".
        engine: pylbl_amd.engine.Engine whose HBM holds the line table.
        molecule: the engine's handle of that table (None: no lines or no partition sums).
    """
    def __init__(self, lines_database, formula, device=0, engine=None):
        """Reads the molecule's transitions, masses and partition sums once and uploads them.

        Args:
            lines_database: what the reference passes its back ends (pyLBL/spectroscopy.py:54):
                            a pyLBL.database.Database -- read through its ``.path``, else
                            through ``.gas()``/``.tips()`` (database.line_table_of) --, this
                            package's Database / MemoryDatabase, or a LineTable itself.
            formula: the molecule, e.g. "CO2" (any alias the database knows).
            device: GPU index.
            engine: an Engine to share; by default the process-wide one of `device`.
        """
        self.formula = formula
        self.engine = engine if engine is not None else default_engine(device)
        self.molecule = None
        self._deferred_error = None
        if isinstance(lines_database, LineTable):
            self.database = None
            table = lines_database
        else:
            self.database = getattr(lines_database, "path", None)
            try:
                table = line_table_of(lines_database, formula)
            except BaseException as error:
                # This package's classes, or the reference's own when `lines_database` is a
                # pyLBL.database.Database that had to be read through .gas()/.tips().
                condition = errors.kind(error)
                if condition in ("AliasNotFoundError", "IsotopologuesNotFoundError"):
                    # Reference: the constructor succeeds, the C call returns 1
                    # (spectral_database.c:126-129,152-156) -> ValueError at compute time.
                    self._deferred_error = ValueError("Error inside c functions.")
                elif condition in ("TipsDataNotFoundError", "TransitionsNotFoundError"):
                    pass        # absorption.c:53-59: rc 0 and a zero spectrum
                else:
                    raise
                table = None
        self.num_lines = table.num_lines if table is not None else 0
        self._first_nu = float(table.nu[0]) if self.num_lines else None
        if table is not None:
            self.molecule = self.engine.load(table)

    def absorption_coefficient(self, temperature, pressure, volume_mixing_ratio, grid,
                               remove_pedestal=False, cut_off=25, range_policy="reference"):
        """Absorption cross-section spectrum of one level (signature of
        pyLBL/c_lib/gas_optics.py:46-47; one level of ``absorption_coefficients``).

        Args:
            temperature [K], pressure [Pa], volume_mixing_ratio [mol mol-1]: the level.
            grid: wavenumbers [cm-1]; must start on an integer with spacing 1/integer.
            remove_pedestal: subtract the running pedestal of spectra.c:66-78.
            cut_off: half-width [cm-1] of every line's window.
            range_policy: "reference" stops at the first row outside the grid +- (cut_off+1)
                          like absorption.c:80-83; "skip" ignores such rows.

        Returns:
            float64 array [m2 molecule-1] of (vn - v0)*n_per_v >= grid.size points, as the
            reference returns it; callers slice [:grid.size].
        """
        return self.absorption_coefficients([temperature], [pressure], [volume_mixing_ratio],
                                            grid, remove_pedestal, cut_off, range_policy)[0]

    def absorption_coefficients(self, temperature, pressure, volume_mixing_ratio, grid,
                                remove_pedestal=False, cut_off=25, range_policy="reference",
                                out=None, scale_density=False, accumulate=False,
                                asynchronous=False, farfield=False, deliver=None, pieces=4,
                                defer_finish=False):
        """Batched form: one spectrum per level, float64[levels, (vn-v0)*n_per_v].

        farfield: sum the lines far from each tile of the grid through one power series per tile
        (engine flag LBL_FARFIELD; truncation <= ~1.5e-11 relative, several times faster on fine
        grids).
        deliver: with a device `out`, a page-locked float64 [levels, columns] view that receives
        the result while the call still computes, in `pieces` runs of tiles (Engine.compute)."""
        if self._deferred_error is not None:
            raise self._deferred_error
        v0, vn, n_per_v = grid_arguments(grid)
        if range_policy == "reference" and self._first_nu is not None and \
                self._first_nu < v0 - (cut_off + 1):
            # pyLBL/c_lib/absorption.c:80-83 leaves its row loop at the first transition outside
            # [v0-(cut_off+1), vn+cut_off+1]: with the first row below the grid that is every row.
            warnings.warn(
                f"{self.formula}: the first transition ({self._first_nu:g} cm-1) lies below "
                f"v0-(cut_off+1) = {v0 - (cut_off + 1)} cm-1; like the reference this yields an "
                "all-zero spectrum.  Pass range_policy='skip' to ignore out-of-range rows instead.",
                RuntimeWarning, stacklevel=3)
        levels = np.atleast_1d(np.asarray(temperature, dtype=np.float64)).size
        if self.molecule is None:
            # No partition-function rows or no transitions: the reference returns the zeroed
            # spectrum (absorption.c:41, :53-59), so a caller's buffer must read zero as well
            # (nothing is added to one that is being accumulated into).
            if out is not None:
                if not accumulate:
                    self.engine.fill_zero(out, asynchronous)
                if deliver is not None:
                    # The caller was promised the block's first columns in `deliver`: zeros, or
                    # (accumulating) whatever the block holds once the calls queued so far have run.
                    if not hasattr(out, "to_host_into"):
                        raise ValueError("deliver needs a device `out`.")
                    if accumulate:
                        out.to_host_into(deliver, deliver.shape[1], asynchronous=asynchronous)
                    else:
                        deliver[...] = 0.
                return out
            return np.zeros((levels, (vn - v0)*n_per_v))
        return self.engine.compute(self.molecule, temperature, pressure, volume_mixing_ratio,
                                   v0, vn, n_per_v, cut_off=cut_off,
                                   remove_pedestal=remove_pedestal, range_policy=range_policy,
                                   out=out, scale_density=scale_density, accumulate=accumulate,
                                   asynchronous=asynchronous, farfield=farfield,
                                   deliver=deliver, pieces=pieces, defer_finish=defer_finish)

    def __del__(self):
        try:
            if self.molecule is not None and self.engine.handle:
                self.engine.free(self.molecule)
        except Exception:
            pass
