"""ARTS-crossfit absorption cross-sections on the MI355X: host side of mechanism slot 2.

Mirrors ``pyLBL.arts_crossfit.CrossSection`` (pyLBL/arts_crossfit/cross_section.py:9-48):
``CrossSection(formula, path)`` and ``absorption_coefficient(grid, temperature, pressure)``
-> cross section [m2] on the grid.  The reference re-opens the coefficient file and redoes
the fit on the CPU for every level; here the bands are uploaded once and every call is two
kernel launches (csrc/xsec.h).  There is no CPU path.

Coefficient files: one netCDF-4 file per molecule with the variables ``bands`` (band
numbers), ``band<m>_fgrid`` (frequency [Hz]) and ``band<m>_coeffs`` (the four fit coefficients
per frequency), read the way cross_section.py:29-40 reads them; a ``.npz`` with the same
variable names is accepted too.  The files themselves are a download
(pyLBL/arts_crossfit/webapi.py) and not part of this repository.
"""
import numpy as np

from . import hdf5_reader
from .engine import default_engine
from .mt_ckd import resident_grid


def _as_matrix(coefficients, size):
    """[4, nfreq] from what the file stores.  The reference transposes what xarray hands it and
    then indexes ``coeffs[i, :]`` for the four fit terms (cross_section.py:40,
    xsec_aux_functions.py:50-52): its files hold ``band<m>_coeffs`` as [nfreq, 4], and that
    reading comes first (a band of exactly four frequencies is the case where the shape alone
    does not tell).  The other orientation, which the reference could not read, is accepted for
    conversions made elsewhere."""
    c = np.asarray(coefficients, dtype=np.float64)
    if c.shape == (size, 4):
        return np.ascontiguousarray(c.T)
    if c.shape == (4, size):
        return c
    raise ValueError(f"coefficients of shape {c.shape} do not match {size} frequencies.")


def read_bands(path):
    """[(frequency [Hz] ascending, coefficients [4, nfreq]), ...] in the file's band order."""
    path = str(path)
    if path.endswith(".npz"):
        with np.load(path) as archive:
            numbers = [int(m) for m in np.atleast_1d(archive["bands"])]
            raw = [(archive[f"band{m}_fgrid"], archive[f"band{m}_coeffs"]) for m in numbers]
    else:
        with hdf5_reader.File(path) as source:
            numbers = [int(m) for m in source.array("bands").ravel()]
            raw = [(source.array(f"band{m}_fgrid"), source.array(f"band{m}_coeffs"))
                   for m in numbers]
    bands = []
    for frequency, coefficients in raw:
        frequency = np.asarray(frequency, dtype=np.float64).ravel()
        coefficients = _as_matrix(coefficients, frequency.size)
        # scipy's interp1d sorts its abscissa first (assume_sorted=False).
        order = np.argsort(frequency, kind="mergesort")
        bands.append((np.ascontiguousarray(frequency[order]),
                      np.ascontiguousarray(coefficients[:, order])))
    return bands


def write_npz(path, bands):
    """Writes bands in the layout read_bands accepts (fixtures, conversions)."""
    arrays = {"bands": np.arange(len(bands))}
    for m, (frequency, coefficients) in enumerate(bands):
        arrays[f"band{m}_fgrid"] = np.asarray(frequency, dtype=np.float64)
        # stored like the reference's files: [nfreq, 4]
        arrays[f"band{m}_coeffs"] = np.ascontiguousarray(np.asarray(coefficients,
                                                                   dtype=np.float64).T)
    np.savez_compressed(path, **arrays)


class CrossSection(object):
    """Absorption cross-sections of one molecule, resident on the GPU.

    Attributes:
        formula: String chemical formula.
        path: Path to the coefficient file.
    """
    def __init__(self, formula, path, device=0, engine=None):
        """Reads the coefficient file and uploads its bands.

        The reference's constructor only stores its arguments and opens the file on every call
        (cross_section.py:10-19,29), and Spectroscopy's MoleculeCache lets nothing but the two
        database errors pass (pyLBL/spectroscopy.py:66-69): a file that cannot be read is
        therefore reported by the first ``absorption_coefficient`` call, as there.
        """
        self.formula = formula
        self.path = path
        self.engine = engine if engine is not None else default_engine(device)
        self.handle = None
        self._deferred_error = None
        try:
            bands = read_bands(path)
        except (OSError, KeyError, ValueError, TypeError) as error:
            self._deferred_error = error
            self.sizes, self.frequency = [], []
            return
        self.sizes = [f.size for f, _ in bands]
        self.frequency = [f for f, _ in bands]
        self.handle = self.engine.load_xsec(bands)

    def __del__(self):
        try:
            if self.handle is not None:
                self.engine.free_xsec(self.handle)
        except Exception:
            pass

    def absorption_coefficient(self, grid, temperature, pressure):
        """Cross section [m2] on `grid` [cm-1] for one level; pressure in Pa."""
        return self.absorption_coefficients(grid, [temperature], [pressure])[0]

    def absorption_coefficients(self, grid, temperature, pressure, volume_mixing_ratio=None,
                                out=None, accumulate=False, asynchronous=False):
        """All levels in one call: float64[levels, grid.size] (or fills `out`, a host array
        or DeviceSpectra).  With volume_mixing_ratio the result is n k [m-1], the slot
        Spectroscopy stores (spectroscopy.py:199-203)."""
        if self._deferred_error is not None:
            raise self._deferred_error
        grid = grid if isinstance(grid, np.ndarray) and grid.dtype == np.float64 and \
            grid.flags["C_CONTIGUOUS"] else np.ascontiguousarray(grid, dtype=np.float64)
        return self.engine.xsec_compute(
            self.handle, resident_grid(self.engine, grid), grid.size, temperature, pressure,
            vmr=volume_mixing_ratio, out=out, accumulate=accumulate, asynchronous=asynchronous)

    def band_values(self, temperature, pressure):
        """The clipped fit on the bands' own frequency grids (calculate_xsec_fullmodel per
        band, xsec_aux_functions.py:80-121)."""
        if self._deferred_error is not None:
            raise self._deferred_error
        return self.engine.xsec_bands(self.handle, self.sizes, temperature, pressure)
