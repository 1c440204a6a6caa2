"""MT-CKD continua on the MI355X: host side of mechanism slot 1.

Mirrors the reference's plug-in surface for continua: the six classes registered under the
``mt_ckd`` entry-point group (setup.py:47-54) with a ``bands`` list and
``spectra(temperature, pressure, vmr, grid) -> extinction [m-1]``
(``BandedContinuum.spectra``, pyLBL/mt_ckd/utils.py:157-174); every band offers
``spectra(temperature, pressure_mb, vmr)`` and ``grid()`` (``Continuum``, utils.py:83-113),
which is what the reference's own test drives (tests/test_mt_ckd.py:29-46).

What the host does: read the coefficient tables (mt_ckd_data), do the constructors' one-off
table preparation (scale factors, analytic band shapes), upload the result once.  Everything
per level -- the 16 band formulas and the interpolation to the user's grid -- runs in the HIP
kernels of csrc/continuum.h; there is no CPU path.
"""
import weakref

import numpy as np

from . import mt_ckd_data
from .engine import VMR_COUNT, VMR_H2O, VMR_N2, VMR_O2, VMR_SELF, VMR_TOTAL, default_engine

# LBL_BAND_* of include/lbl_amd.h.
(H2O_SELF, H2O_FOREIGN, CO2, N2_ROTATION, N2_FUNDAMENTAL, N2_OVERTONE, O2_FUNDAMENTAL, O2_NIR,
 O2_NIR2, O2_NIR3, O2_VISIBLE, O2_HERZBERG, O2_UV, O3_CHAPPUIS, O3_HARTLEY, O3_UV) = range(16)


def _inside(outer, inner):
    """First and last index of table `inner` within `outer` (subgrid_bounds, utils.py:62-80)."""
    if outer.resolution != inner.resolution:
        raise ValueError("grid and subgrid have different resolutions.")
    if outer.lower_bound > inner.lower_bound or outer.upper_bound < inner.upper_bound:
        raise ValueError("subgrid not contained in grid.")
    return (int((inner.lower_bound - outer.lower_bound)/outer.resolution),
            int((inner.upper_bound - outer.lower_bound)/outer.resolution))


def _tabulated(kind, *names):
    """Band whose columns are variables of the data set, unchanged."""
    def prepare(tables):
        first = tables[names[0]]
        return kind, first.lower_bound, first.resolution, [tables[n].data for n in names]
    return prepare


def _h2o_foreign(tables):
    """bfh2o with its scale factor: the tabulated correction up to 600 cm-1, then the
    closed form of water_vapor.py:55-69."""
    base, factor = tables["bfh2o"], tables["xfac_rhu"]
    first, last = _inside(base, factor)
    scale = np.zeros(base.data.size)
    scale[first + 1:last + 1] = factor.data[1:]
    scale[first] = scale[first + 1]
    w = base.wavenumbers()[last + 1:]
    near = 57600./((w - 255.67)**2 + 57600. + np.power((w - 255.67)/57.83, 8))
    mirror = 57600./((w + 255.67)**2 + 57600. + np.power((w + 255.67)/57.83, 8))
    scale[last + 1:] = 1. + (0.06 - 0.42*(near + mirror))/(1. + 0.3*np.power(w/630., 8))
    return H2O_FOREIGN, base.lower_bound, base.resolution, [base.data, scale]


def _co2(tables):
    """bfco2 with the chi factor and the band-head temperature exponent spread over its grid
    (1 elsewhere; carbon_dioxide.py:21-31)."""
    base = tables["bfco2"]
    columns = []
    for name in ("x_factor_co2", "tdep_bandhead"):
        column = np.ones(base.data.size)
        first, last = _inside(base, tables[name])
        column[first:last + 1] = tables[name].data
        columns.append(column)
    return CO2, base.lower_bound, base.resolution, [base.data] + columns


def _o2_nir2(tables):
    """Two damped Lorentzians on 9100..11000 step 2, divided by wavenumber (oxygen.py:56-67)."""
    w = np.arange(9100., 11002., 2.)
    shape = np.zeros(w.size)
    for centre, width, strength in ((9375., 58.96, 1.166e-04), (9439., 45.04, 3.086e-05)):
        d = w - centre
        damping = np.where(d < 0., np.exp(d/176.1), 1.)
        shape = shape + (strength*damping/width)/(1. + (d/width)*(d/width))
    return O2_NIR2, 9100., 2., [0.31831*shape*1.054/w]


def _o2_herzberg(tables):
    """Closed-form Herzberg shape on 36000..100000 step 10 (oxygen.py:113-124)."""
    w = np.arange(36000., 100010., 10.)
    ratio = w/48811.0
    shape = 6.884e-4*ratio*np.exp(-69.738*np.power(np.log(ratio), 2))
    shape = shape - np.where(w <= 40000., ((40000. - w)/4000.)*7.917e-7, 0.)
    shape[w <= 36000.] = 0.
    return O2_HERZBERG, 36000., 10., [shape]


def _grid_fingerprint(grid):
    """Cheap signature of an array's current contents: size, end points and 4096 evenly spaced
    samples.  (A full checksum of a 5 M-point grid costs as much as the kernels it feeds.)"""
    step = max(grid.size//4096, 1)
    return (grid.size, float(grid[0]), float(grid[-1]), hash(grid[::step].tobytes()))


def resident_grid(engine, grid):
    """Handle of `grid` in the engine's HBM: uploaded once per array object and shared by all
    continua (the reference is handed the same array for every gas and level,
    spectroscopy.py:195).  An array whose contents changed since the upload (edited or
    refilled in place) is uploaded again; copies whose array has been garbage-collected are
    freed here."""
    cache = engine.__dict__.setdefault("_resident_grids", [])
    found = None
    # Entries leave the list by identity: comparing two of them compares their weak references,
    # i.e. the arrays behind them, element by element.
    for entry in list(cache):
        target = entry[0]()
        if target is None:
            engine.free_grid(entry[1])
            cache[:] = [other for other in cache if other is not entry]
        elif target is grid:
            found = entry
    fingerprint = _grid_fingerprint(grid)
    if found is not None and found[2] != fingerprint:
        engine.synchronize()        # queued kernels may still read the old copy
        engine.free_grid(found[1])
        cache[:] = [other for other in cache if other is not found]
        found = None
    if found is None:
        handle = engine.load_grid(grid)
        try:
            found = (weakref.ref(grid), handle, fingerprint)
        except TypeError:
            found = (lambda: None, handle, fingerprint)     # not weak-referenceable: freed next time
        cache.append(found)
    return found[1]


class Band(object):
    """One band of a continuum: the ``Continuum`` interface of utils.py:83-113."""
    def __init__(self, owner, index, lower_bound, resolution, size):
        self._owner, self._index = owner, index
        self.lower_bound, self.resolution, self.size = lower_bound, resolution, size

    def grid(self):
        return self.lower_bound + np.arange(self.size)*self.resolution

    def spectra(self, temperature, pressure, vmr):
        """Coarse spectrum [cm-1]; pressure in mb, vmr a dictionary of mole fractions."""
        return self._owner._band_spectra(temperature, pressure, vmr)[self._index]


class BandedContinuum(object):
    """All bands of one continuum, resident on the GPU."""
    gas = None          # key of the mole-fraction dictionary this continuum belongs to
    needs = ()          # other keys its formulas read
    recipe = ()         # one preparation function per band

    def __init__(self, path=None, device=0, engine=None):
        tables = mt_ckd_data.load(path)
        self.engine = engine if engine is not None else default_engine(device)
        prepared = [prepare(tables) for prepare in self.recipe]
        self.bands = [Band(self, i, lower, resolution, columns[0].size)
                      for i, (_, lower, resolution, columns) in enumerate(prepared)]
        self.handle = self.engine.load_continuum(prepared)

    def __del__(self):
        try:
            self.engine.free_continuum(self.handle)
        except Exception:
            pass

    # -- mole fractions ----------------------------------------------------------------
    def pack_vmr(self, vmr, levels=None):
        """[levels, VMR_COUNT] from a dictionary formula -> scalar or array."""
        for key in (self.gas,) + tuple(self.needs):
            if key not in vmr:
                raise KeyError(key)
        size = 1 if levels is None else levels
        packed = np.zeros((size, VMR_COUNT))
        for column, key in ((VMR_SELF, self.gas), (VMR_H2O, "H2O"), (VMR_O2, "O2"),
                            (VMR_N2, "N2")):
            if key in vmr:
                packed[:, column] = np.asarray(vmr[key], dtype=np.float64).ravel()
        # air_number_density adds up every entry of the dictionary (utils.py:16-28).
        total = 0.
        for value in vmr.values():
            total = total + np.asarray(value, dtype=np.float64).ravel()
        packed[:, VMR_TOTAL] = total
        return packed

    # -- grids -------------------------------------------------------------------------
    def grid_handle(self, grid):
        return resident_grid(self.engine, grid)

    # -- the two reference entry points ------------------------------------------------
    def _band_spectra(self, temperature, pressure_mb, vmr):
        return self.engine.continuum_bands(self.handle, [b.size for b in self.bands],
                                           temperature, pressure_mb, self.pack_vmr(vmr))

    def spectra(self, temperature, pressure, vmr, grid):
        """Continuum extinction [m-1] on `grid` for one level; pressure in Pa."""
        grid = np.ascontiguousarray(grid, dtype=np.float64)
        return self.spectra_levels([temperature], [pressure], vmr, grid)[0]

    def spectra_levels(self, temperature, pressure, vmr, grid, out=None, accumulate=False,
                       asynchronous=False):
        """All levels in one call: float64[levels, grid.size] (or fills `out`, a host array
        or DeviceSpectra); vmr maps formula -> array over levels."""
        t = np.atleast_1d(np.asarray(temperature, dtype=np.float64))
        grid = grid if isinstance(grid, np.ndarray) and grid.dtype == np.float64 and \
            grid.flags["C_CONTIGUOUS"] else np.ascontiguousarray(grid, dtype=np.float64)
        return self.engine.continuum_compute(
            self.handle, self.grid_handle(grid), grid.size, t, pressure,
            self.pack_vmr(vmr, t.size), out=out, accumulate=accumulate,
            asynchronous=asynchronous)


class WaterVaporSelfContinuum(BandedContinuum):
    gas = "H2O"
    recipe = (_tabulated(H2O_SELF, "bs296", "bs260"),)


class WaterVaporForeignContinuum(BandedContinuum):
    gas = "H2O"
    recipe = (_h2o_foreign,)


class CarbonDioxideContinuum(BandedContinuum):
    gas, needs = "CO2", ("H2O",)
    recipe = (_co2,)


class NitrogenContinuum(BandedContinuum):
    gas, needs = "N2", ("H2O", "O2")
    recipe = (_tabulated(N2_ROTATION, "ct_296", "ct_220", "sf_296", "sf_220"),
              _tabulated(N2_FUNDAMENTAL, "xn2_272", "xn2_228", "a_h2o"),
              _tabulated(N2_OVERTONE, "xn2"))


class OxygenContinuum(BandedContinuum):
    gas, needs = "O2", ("H2O", "N2")
    recipe = (_tabulated(O2_FUNDAMENTAL, "o2_f", "o2_t"), _tabulated(O2_NIR, "o2_inf1"),
              _o2_nir2, _tabulated(O2_NIR3, "o2_inf3"), _tabulated(O2_VISIBLE, "o2_invis"),
              _o2_herzberg, _tabulated(O2_UV, "o2_infuv"))


class OzoneContinuum(BandedContinuum):
    gas, needs = "O3", ("H2O",)
    recipe = (_tabulated(O3_CHAPPUIS, "x_o3", "y_o3", "z_o3"),
              _tabulated(O3_HARTLEY, "o3_hh0", "o3_hh1", "o3_hh2"), _tabulated(O3_UV, "o3_huv"))


def spectra_levels_many(continua, temperature, pressure, vmr, grid, out, accumulate=False,
                        asynchronous=False):
    """Every continuum of `continua` (BandedContinuum objects of one engine) for all levels in
    ONE pass over the grid, summed in that order into the DeviceSpectra `out`
    (lbl_continuum_compute_many): what compute_absorption's loops over a gas's continua
    (pyLBL/spectroscopy.py:193-197) and, in its "gas" / "total" formats, over the gases
    (:225-234) amount to -- the same bits as one spectra_levels(..., accumulate=True) per
    continuum."""
    continua = list(continua)
    if not continua:
        return out
    engine = continua[0].engine
    if len(continua) == 1 or any(c.engine is not engine for c in continua) or \
            sum(len(c.bands) for c in continua) > 64:
        for i, continuum in enumerate(continua):
            continuum.spectra_levels(temperature, pressure, vmr, grid, out=out,
                                     accumulate=accumulate or i > 0, asynchronous=asynchronous)
        return out
    t = np.atleast_1d(np.asarray(temperature, dtype=np.float64))
    grid = grid if isinstance(grid, np.ndarray) and grid.dtype == np.float64 and \
        grid.flags["C_CONTIGUOUS"] else np.ascontiguousarray(grid, dtype=np.float64)
    packed = np.stack([c.pack_vmr(vmr, t.size) for c in continua])
    return engine.continuum_compute_many(
        [c.handle for c in continua], resident_grid(engine, grid), grid.size, t, pressure,
        packed, out, accumulate=accumulate, asynchronous=asynchronous)


# The dictionary pyLBL.plugins builds for the group "mt_ckd" (plugins.py:24-34): the part of
# the entry-point name in front of "Continuum" -> class.
CONTINUA = {"CO2": CarbonDioxideContinuum, "H2OForeign": WaterVaporForeignContinuum,
            "H2OSelf": WaterVaporSelfContinuum, "N2": NitrogenContinuum,
            "O2": OxygenContinuum, "O3": OzoneContinuum}
