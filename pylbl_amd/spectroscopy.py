"""Host orchestration: the counterpart of ``Spectroscopy.compute_absorption``
(pyLBL/spectroscopy.py:144-206) for all three mechanism slots: 0 "lines", 1 "continuum"
(MT-CKD), 2 "cross_section" (ARTS-crossfit).

What is kept from the reference: constructor keywords and the KeyError for an unknown
backend name (:88-118); molecule-outer / level-inner semantics with no state carried between
units (:166-191); beta[level, 0, :] = n * k[:grid.size] with n = P x /(kb T) (:18-29,
:181-191); ``remove_pedestal`` defaulting to ``continua_backend == "mt_ckd"`` (:163-164);
the three output formats (:208-235).

What is different: all levels of a molecule go to the GPU in one batched call and the
number-density scaling happens in the kernel epilogue; xarray is optional (not installed in
this image) -- without it the atmosphere is a plain (p, t, vmr) tuple and the result a dict
of numpy arrays with the same variable names.
"""
from collections import namedtuple
import contextlib
import os

import numpy as np

from . import errors
from .plugins import continua, cross_sections, molecular_lines
from .synthetic import grid_arguments

kb = 1.38064852e-23  # Boltzmann constant [J K-1] (pyLBL/spectroscopy.py:15).

MECHANISMS = ["lines", "continuum", "cross_section"]


def number_density(temperature, pressure, volume_mixing_ratio):
    """Ideal-gas number density [m-3] (pyLBL/spectroscopy.py:18-29)."""
    return pressure*volume_mixing_ratio/(kb*temperature)


_STANDARD_NAME_PREFIX = "mole_fraction_of_"
_FORMULAE = {"water_vapor": "H2O", "carbon_dioxide": "CO2", "ozone": "O3",
             "nitrous_oxide": "N2O", "carbon_monoxide": "CO", "methane": "CH4",
             "oxygen": "O2", "nitrogen": "N2"}


_XARRAY = []


def _optional_xarray():
    """The xarray module, or None where it is not installed -- looked for ONCE: a failed import
    walks every entry of sys.path again (60 us per compute_absorption call, behind its last wait:
    profiles/r05_perf_api_profile.txt)."""
    if not _XARRAY:
        try:
            import xarray
            _XARRAY.append(xarray)
        except ImportError:
            _XARRAY.append(None)
    return _XARRAY[0]


class Atmosphere(object):
    """Pressure, temperature and gas mole fractions as numpy arrays.

    Accepts a (p, t, vmr) tuple/namedtuple or dict with those keys -- vmr maps a chemical
    formula to an array shaped like t -- or, when xarray is installed, a Dataset read by CF
    standard_name the way pyLBL/atmosphere.py:21-47 does.
    """
    def __init__(self, atmosphere, mapping=None):
        if hasattr(atmosphere, "data_vars"):
            self._from_dataset(atmosphere, mapping)
            return
        if isinstance(atmosphere, dict):
            p, t, vmr = atmosphere["p"], atmosphere["t"], atmosphere["vmr"]
        else:
            p, t, vmr = atmosphere.p, atmosphere.t, atmosphere.vmr
        self.pressure = np.asarray(p, dtype=np.float64)
        self.temperature = np.asarray(t, dtype=np.float64)
        self.gases = {k: np.asarray(v, dtype=np.float64) for k, v in vmr.items()}
        for name, value in self.gases.items():
            if value.shape != self.temperature.shape:
                raise ValueError(f"mole fraction of {name} is not shaped like temperature.")
        self.dims = [f"dim_{i}" for i in range(self.temperature.ndim)]

    def _from_dataset(self, dataset, mapping):
        def find(standard_name):
            for name, var in dataset.data_vars.items():
                if var.attrs.get("standard_name") == standard_name:
                    return var
            raise ValueError(f"standard name {standard_name} not found in dataset.")
        if mapping is None:
            pressure, temperature = find("air_pressure"), find("air_temperature")
            gases = {}
            for var in dataset.data_vars.values():
                name = var.attrs.get("standard_name", "")
                if name.startswith(_STANDARD_NAME_PREFIX) and name.endswith("_in_air"):
                    key = name[len(_STANDARD_NAME_PREFIX):-len("_in_air")]
                    gases[_FORMULAE.get(key, key)] = var
        else:
            pressure, temperature = dataset[mapping["play"]], dataset[mapping["tlay"]]
            gases = {k: dataset[v] for k, v in mapping["mole_fraction"].items()}
        self.pressure = np.asarray(pressure.data, dtype=np.float64)
        self.temperature = np.asarray(temperature.data, dtype=np.float64)
        self.gases = {k: np.asarray(v.data, dtype=np.float64) for k, v in gases.items()}
        self.dims = list(temperature.dims)


class MoleculeCache(object):
    """Caches the per-molecule backend objects (pyLBL/spectroscopy.py:32-69): building them
    uploads the molecule's line table and continuum coefficients to HBM once."""
    def __init__(self, name, lines_database, lines_engine, continua_engine,
                 cross_sections_engine, device):
        # `lines_database` may be the reference's own Database object, whose error classes are
        # not this package's: the conditions are told apart by name (errors.kind).
        try:
            self.gas = lines_engine(lines_database, name, device=device)
        except BaseException as error:
            if errors.kind(error) not in ("AliasNotFoundError", "IsotopologuesNotFoundError",
                                          "TipsDataNotFoundError", "TransitionsNotFoundError"):
                raise
            self.gas = None
        # Water vapour has two continua, every other gas at most one (spectroscopy.py:58-65).
        names = [name + "Foreign", name + "Self"] if name == "H2O" else [name]
        self.gas_continua = None
        if continua_engine is not None:
            try:
                self.gas_continua = [continua_engine[x](device=device) for x in names]
            except KeyError:
                self.gas_continua = None
        self.cross_section = None
        if cross_sections_engine is not None and hasattr(lines_database, "arts_crossfit"):
            try:
                self.cross_section = cross_sections_engine(
                    name, lines_database.arts_crossfit(name), device=device)
            except BaseException as error:
                if errors.kind(error) not in ("AliasNotFoundError", "CrossSectionNotFoundError"):
                    raise
                self.cross_section = None


class _Sum(object):
    """A [levels, n] block in HBM that kernels write first and add into afterwards."""
    def __init__(self, engine, levels, n):
        self.engine = engine
        self.buffer = engine.blocks.take(levels, n)     # recycled (engine.DevicePool)
        self.written = False

    def take(self):
        """True if the next kernel must add to what is there."""
        written, self.written = self.written, True
        return written

    def into(self, target):
        """Queues the copy of the first target.shape[1] columns of every level straight into
        `target` (a [levels, columns] view with contiguous rows, page-locked); it runs behind
        the kernels queued so far and beside those queued later."""
        self.buffer.to_host_into(target, target.shape[1], asynchronous=True)
        return self


def _zero_in_background(views):
    """Zeroes float64 [rows, columns] views with contiguous rows on a helper thread (ctypes
    releases the interpreter lock during memset); returns the started thread, or None."""
    if not views:
        return None
    import ctypes
    import threading

    def fill():
        for view in views:
            row_bytes = view.shape[1]*8
            for row in range(view.shape[0]):
                ctypes.memset(view[row].ctypes.data, 0, row_bytes)
    thread = threading.Thread(target=fill)
    thread.start()
    return thread


class Spectroscopy(object):
    """Line-by-line gas optics (lines, MT-CKD continua, ARTS-crossfit cross-sections) on an
    MI355X.

    Attributes mirror pyLBL/spectroscopy.py:72-86.
    """
    def __init__(self, atmosphere, grid, database, mapping=None, lines_backend="mi355x",
                 continua_backend="mt_ckd", cross_sections_backend="arts_crossfit", device=0,
                 group=None, gather_to=0, farfield=True):
        """Args beyond the reference's (pyLBL/spectroscopy.py:88-118):
            device: GPU index of this process.
            farfield: True (default): lines far from a tile of the grid enter through one power
                   series per tile (pylbl_amd/csrc/farfield.h; truncation <= ~1.5e-11 relative,
                   asserted by tests/test_gpu_api.py at 100, 1000 and 2000 points per cm-1;
                   3-4x faster on fine grids).  False: every line at every point of its window,
                   like the reference's loop.
            group: None: this process computes every level.  True (the default process group)
                   or a torch.distributed ProcessGroup: one process per GPU, each computes a
                   contiguous block of levels (all gases and mechanisms of a level on the same
                   GPU, pylbl_amd.distributed.level_shard) and compute_absorption collects the
                   result on rank `gather_to` (None: on every rank); the other ranks get None.
        """
        self.atmosphere = Atmosphere(atmosphere, mapping=mapping)
        # A private copy: the grid's device copy is cached per array object, and the lines
        # path reads (v0, vn, n_per_v) off it on every call -- a caller editing its own array
        # afterwards must not move one mechanism's grid and leave the others behind.
        self.grid = np.array(grid, dtype=np.float64, order="C", copy=True)
        self.lines_database = database
        self.lines_backend = lines_backend
        self.lines_engine = molecular_lines[lines_backend]      # KeyError if unknown
        self.continua_backend = continua_backend
        # None switches the mechanism off; an unknown name is a KeyError (spectroscopy.py:118).
        self.continua_engine = None if continua_backend is None else continua[continua_backend]
        self.cross_sections_backend = cross_sections_backend
        self.cross_sections_engine = None if cross_sections_backend is None \
            else cross_sections[cross_sections_backend]
        self.cache = {}
        self.device = device
        self.group = group
        self.gather_to = gather_to
        self.farfield = bool(farfield)
        self.device_output_limit = 8 << 30     # bytes of spectra kept in HBM per block
        self.delivery_pieces = 4               # runs of tiles of the call that delivers its result
        # "total": in which order the gases add into the one block (see _compute_levels).
        self.total_order = "heavy_last"
        # "gas": "each" -- every gas's lines call delivers its block piece by piece; "last" -- only
        # the last gas does, the others' blocks travel in one copy each.
        self.gas_delivery = os.environ.get("PYLBL_AMD_GAS_DELIVERY", "each")
        Output = namedtuple("Output", ["dims", "dim_sizes", "mechanisms", "units"])
        dims = list(self.atmosphere.dims) + ["mechanism", "wavenumber"]
        dim_sizes = list(self.atmosphere.temperature.shape) + [len(MECHANISMS), self.grid.size]
        self.output = Output(dims=dims, dim_sizes=dim_sizes, mechanisms=MECHANISMS,
                             units={"units": "m-1"})

    def list_molecules(self):
        return self.lines_database.molecules()

    def _molecule(self, name):
        data = self.cache.get(name)
        if data is None:
            data = MoleculeCache(name, self.lines_database, self.lines_engine,
                                 self.continua_engine, self.cross_sections_engine, self.device)
            self.cache[name] = data
        return data

    def compute_absorption(self, output_format="all", remove_pedestal=None,
                           range_policy="reference"):
        """Computes the absorption coefficient [m-1] on the grid for every level and gas of
        the atmosphere: spectral lines (slot 0), MT-CKD continua (slot 1) and ARTS-crossfit
        cross-sections (slot 2).

        Args:
            output_format: "all" (per gas, per mechanism), "gas" (per gas, mechanisms summed)
                           or anything else for the total over gases (spectroscopy.py:208-235).
            remove_pedestal: Subtract the MT-CKD "pedestal" (default: True when the continuum
                             backend is "mt_ckd", as spectroscopy.py:163-164).

        Returns:
            xarray Dataset when xarray is installed, else a dict of numpy arrays with the
            same variable names ("wavenumber", "mechanism", "<formula>_absorption" /
            "absorption").
        """
        temperature = self.atmosphere.temperature.ravel()
        pressure = self.atmosphere.pressure.ravel()
        shape = list(self.atmosphere.temperature.shape)
        if remove_pedestal is None:
            remove_pedestal = self.continua_backend == "mt_ckd"
        # Every gas at every level, the dictionary the continua read (spectroscopy.py:173).
        mole_fractions = {name: x.ravel() for name, x in self.atmosphere.gases.items()}
        mode = output_format if output_format in ("all", "gas") else "total"
        columns = self.grid.size
        if self.group is None:
            flat = self._compute_levels(temperature, pressure, mole_fractions, mode,
                                        remove_pedestal, range_policy)
        else:
            # One process per GPU: this rank's block of levels, then one collection per array.
            from . import distributed
            group = None if self.group is True else self.group
            rank, world, _ = distributed._group_info(group)
            mine = distributed.level_shard(temperature.size, rank, world)
            local = self._compute_levels(
                temperature[mine], pressure[mine], {k: v[mine] for k, v in mole_fractions.items()},
                mode, remove_pedestal, range_policy)
            flat = {name: distributed.gather_arrays(values, temperature.size, self.gather_to,
                                                    group, device=self.device)
                    for name, values in local.items()}
            if any(values is None for values in flat.values()):
                return None
        tail = [len(MECHANISMS), columns] if mode == "all" else [columns]
        return self._create_output_dataset(
            {name: values.reshape(shape + tail) for name, values in flat.items()}, output_format)

    def _compute_levels(self, temperature, pressure, mole_fractions, mode, remove_pedestal,
                        range_policy):
        """The three mechanism slots for a flat list of levels: {variable name: array with the
        levels as leading dimension} ("total" under mode "total")."""
        levels = temperature.size
        v0, vn, n_per_v = grid_arguments(self.grid)
        n = (vn - v0)*n_per_v
        columns = self.grid.size
        if levels == 0:
            # A rank without levels (fewer levels than GPUs): empty blocks of the right shape.
            if mode == "total":
                return {"total": np.zeros((0, columns))}
            tail = (len(MECHANISMS), columns) if mode == "all" else (columns,)
            return {"{}_absorption".format(name): np.zeros((0,) + tail)
                    for name in self.atmosphere.gases}
        in_hbm = levels*n*8 <= self.device_output_limit
        engine = None

        # Queue every kernel before waiting: one batched call per (molecule, mechanism) for
        # all levels, n*k applied in the kernel epilogue, spectra left in HBM until the end;
        # the sums over mechanisms ("gas") and over gases ("total") happen on the device.
        # Within a block the short continuum and cross-section kernels go first and the lines
        # last: the lines call that completes the LAST block of the whole call hands its result
        # to the host itself, piece by piece while it computes (lbl_compute_streamed), so no
        # copy is left standing behind the last kernel.
        blocks = {}             # (gas, mechanism) -> host array (only when too large for HBM)
        zero_fills = []         # row views of results that no mechanism writes
        results = {}            # gas -> its finished array, being filled by queued copies
        in_flight = []          # blocks in HBM to release once everything has arrived
        total = None
        present = []
        for name in self.atmosphere.gases:
            data = self._molecule(name)
            gas = data.gas
            if gas is not None and gas.molecule is None:
                # Deferred errors (unknown alias) surface here like in the reference.
                gas.absorption_coefficients(temperature[:1], pressure[:1],
                                            mole_fractions[name][:1], self.grid)
                gas = None
            continua_here = data.gas_continua or []
            cross = data.cross_section
            if gas is None and not continua_here and cross is None:
                continue
            if engine is None:
                engine = gas.engine if gas is not None else \
                    (continua_here[0].engine if continua_here else cross.engine)
            present.append((name, gas, continua_here, cross))
        # "total" (one block for everything): the gas with the most transitions is queued FIRST and
        # finished LAST.  Its lines call is the longest (with the pedestal removed it ends in a
        # serial chain), so everything it does in buffers of its own -- prologue, far-field series,
        # accumulate, pedestal pre-pass -- starts at once and runs beside the other gases' calls,
        # while the kernels that touch its block, and the copies that hand that block to the host
        # piece by piece, are kept back (LBL_DEFER_FINISH) until the others have been queued: it
        # stays the last to add into a shared block, and its copies queue up behind the other gases'
        # copies, not in front of them.  (Units are independent, spectroscopy.py:166,179; results
        # are reported in the atmosphere's order.) Per-gas blocks ("gas", "all") are the other way
        # round: the link to the host is the bottleneck there (one block per gas to copy), so the
        # lightest gas goes first -- its block is complete early and travels beside the kernels of
        # the others -- and the heaviest last, delivering its block piece by piece while it computes
        # (profiles/r03_ab_api.txt).
        present.sort(key=lambda entry: entry[1].num_lines if entry[1] is not None else -1)
        heavy = present[-1] if present and present[-1][1] is not None else None
        if heavy is not None and mode == "total":
            present = [heavy] + present[:-1]

        # ("all" is bound by the link -- four 40 MB blocks per level for H2O + CO2 -- and its
        # copies are queued back to back as they are: cutting the last one into pieces only
        # puts gaps into that queue, 3.6 -> 4.1 ms per call.)
        pieces = 1 if mode == "all" else self.delivery_pieces

        def lines_into(name, gas, block, deliver=None, defer=False):
            gas.absorption_coefficients(
                temperature, pressure, mole_fractions[name], self.grid,
                remove_pedestal=remove_pedestal, range_policy=range_policy,
                scale_density=True, out=block.buffer, accumulate=block.take(),
                asynchronous=True, farfield=self.farfield, deliver=deliver,
                pieces=pieces, defer_finish=defer)

        def continua_into(continua_list, continuum_sum):
            # All of them in one pass over the grid where they are this package's (one launch that
            # writes the block once instead of a read-modify-write pass per continuum; the same
            # bits: csrc/continuum.h, group kernels); anything else one by one.
            if not continua_list:
                return
            from .mt_ckd import BandedContinuum, spectra_levels_many
            if all(isinstance(c, BandedContinuum) for c in continua_list):
                spectra_levels_many(continua_list, temperature, pressure, mole_fractions,
                                    self.grid, continuum_sum.buffer,
                                    accumulate=continuum_sum.take(), asynchronous=True)
                return
            for continuum in continua_list:
                continuum.spectra_levels(temperature, pressure, mole_fractions, self.grid,
                                         out=continuum_sum.buffer,
                                         accumulate=continuum_sum.take(), asynchronous=True)

        def slots_into(name, continua_here, cross, continuum_sum, cross_sum):
            continua_into(continua_here, continuum_sum)
            if cross is not None:
                cross.absorption_coefficients(self.grid, temperature, pressure,
                                              volume_mixing_ratio=mole_fractions[name],
                                              out=cross_sum.buffer, accumulate=cross_sum.take(),
                                              asynchronous=True)

        # Everything from the first queued call to the final wait is one pipeline on the engine:
        # calls add into shared blocks in a fixed order and one of them may be kept back, so
        # another thread's calls must not come in between (Engine.pipeline; single calls from
        # other threads -- Gas.absorption_coefficient -- only wait for their turn).  If anything
        # fails on the way, what the engine still holds for this call is dropped and waited for
        # BEFORE the blocks and page-locked arrays go back to their pools: a call kept back
        # (LBL_DEFER_FINISH) would otherwise apply itself, and copy, into recycled memory the next
        # time the engine is synchronized.
        with (engine.pipeline if engine is not None else contextlib.nullcontext()):
            try:
                if not in_hbm:
                    for name, gas, continua_here, cross in present:
                        # Too large to keep: one host block per mechanism, summed by numpy below.
                        if gas is not None:
                            blocks[(name, 0)] = gas.absorption_coefficients(
                                temperature, pressure, mole_fractions[name], self.grid,
                                remove_pedestal=remove_pedestal, range_policy=range_policy,
                                scale_density=True, farfield=self.farfield)[:, :columns]
                        for continuum in continua_here:
                            values = continuum.spectra_levels(temperature, pressure, mole_fractions,
                                                              self.grid)
                            blocks[(name, 1)] = blocks[(name, 1)] + values if (name, 1) in blocks \
                                else values
                        if cross is not None:
                            blocks[(name, 2)] = cross.absorption_coefficients(
                                self.grid, temperature, pressure,
                                volume_mixing_ratio=mole_fractions[name])
                elif mode == "total" and present:
                    # Every gas adds into one block.  The heavy gas's slot kernels go first (the
                    # first of them writes the block -- or the engine clears it), then its lines
                    # call, kept back; the other gases' lines with their short continuum and cross-
                    # section kernels behind them; then the heavy gas's last kernels and the
                    # delivery of the finished block.
                    total = _Sum(engine, levels, n)
                    results["total"] = engine.host_array((levels, columns))
                    kept_back = False
                    if heavy is not None and self.total_order == "heavy_last":
                        # The short continuum and cross-section kernels of every gas first, the
                        # lighter gases' lines behind them, the heaviest gas last: each run of tiles
                        # it finishes completes that part of the block, which goes to the host while
                        # the next run computes (its pedestal pass is short since round 4, so the
                        # first copy starts a third of the way into the call instead of behind
                        # everything).  (Lines first and the slot kernels behind them was tried: the
                        # slot kernels then wait for the first gas's pedestal to be applied and the
                        # heaviest gas is queued later, 1.58 -> 1.70 ms.)
                        # (every continuum of every gas in ONE pass -- the block is written once --
                        # then the cross-sections.  The additions into the block therefore run
                        # c(g1), c(g2), ..., x(g1), x(g2), ..., lines -- not the reference's
                        # gas-by-gas order, spectroscopy.py:225-234: the continua are bit-identical
                        # to the one-by-one sum among themselves, the total may differ from the
                        # reference's order of additions in its last bits, within the parity bar:
                        # tests/test_gpu_api.py::test_total_with_continuum_and_cross_section_of_two_gases)
                        continua_into([c for _, _, continua_here, _ in present
                                       for c in continua_here], total)
                        for name, gas, continua_here, cross in present:
                            slots_into(name, [], cross, total, total)
                        if not total.written:
                            engine.fill_zero(total.buffer, asynchronous=True)
                            total.take()
                        for name, gas, continua_here, cross in present[1:]:
                            if gas is not None:
                                lines_into(name, gas, total)
                        lines_into(heavy[0], heavy[1], total, deliver=results["total"])
                        in_flight.append(total)
                        present = []
                    for index, (name, gas, continua_here, cross) in enumerate(present):
                        if heavy is not None and index == 0:
                            slots_into(name, continua_here, cross, total, total)
                            if not total.written:
                                engine.fill_zero(total.buffer, asynchronous=True)
                                total.take()
                            lines_into(name, gas, total, deliver=results["total"], defer=True)
                            kept_back = engine.deferred()
                            continue
                        if gas is not None:
                            lines_into(name, gas, total)
                        slots_into(name, continua_here, cross, total, total)
                    if heavy is not None and kept_back:
                        engine.finish_deferred()
                        in_flight.append(total)
                    elif self.total_order == "heavy_last" and heavy is not None:
                        pass
                    else:
                        # (No gas with lines -- or a call the engine could not keep back, e.g.
                        # without a pedestal pass: it added at once and delivered a block that was
                        # not complete; this copy, queued behind everything, is the one that
                        # counts.)
                        in_flight.append(total.into(results["total"]))
                else:
                    for index, (name, gas, continua_here, cross) in enumerate(present):
                        last = index + 1 == len(present)
                        if mode == "gas":
                            block = _Sum(engine, levels, n)
                            results[name] = engine.host_array((levels, columns))
                            if gas is not None and (last or self.gas_delivery == "each"):
                                slots_into(name, continua_here, cross, block, block)
                                lines_into(name, gas, block, deliver=results[name])
                                in_flight.append(block)
                            else:
                                if gas is not None:
                                    lines_into(name, gas, block)
                                slots_into(name, continua_here, cross, block, block)
                                # This gas's block goes home while the next gas computes: one copy,
                                # from HBM straight into its place in a page-locked result.
                                in_flight.append(block.into(results[name]))
                            continue
                        values = engine.host_array([levels, len(MECHANISMS), columns])
                        results[name] = values
                        continuum_sum = _Sum(engine, levels, n) if continua_here else None
                        cross_sum = _Sum(engine, levels, n) if cross is not None else None
                        if continua_here or cross is not None:
                            slots_into(name, continua_here, cross, continuum_sum, cross_sum)
                        for slot, block in ((1, continuum_sum), (2, cross_sum)):
                            if block is None:
                                # An empty mechanism slot reads zero (40 MB per level at 5 M
                                # points): filled by a helper thread beside the queueing and the
                                # kernels.
                                zero_fills.append(values[:, slot, :])
                            else:
                                in_flight.append(block.into(values[:, slot, :]))
                        if gas is None:
                            zero_fills.append(values[:, 0, :])
                        else:
                            lines_sum = _Sum(engine, levels, n)
                            if last:
                                lines_into(name, gas, lines_sum, deliver=values[:, 0, :])
                                in_flight.append(lines_sum)
                            else:
                                lines_into(name, gas, lines_sum)
                                in_flight.append(lines_sum.into(values[:, 0, :]))
                filler = _zero_in_background(zero_fills)
                if engine is not None:
                    engine.synchronize()
                if filler is not None:
                    filler.join()
                for block in in_flight:
                    engine.blocks.give(block.buffer)
            except BaseException:
                if engine is not None:
                    try:
                        engine.cancel_deferred()
                        engine.synchronize()
                    except Exception:       # the first error is the one to report
                        pass
                raise

        if mode == "total":
            values = results.get("total")
            if values is None:
                values = np.zeros((levels, columns))
            for block in blocks.values():           # host blocks of the too-large case
                values += block
            return {"total": values}
        beta = {}
        for name in self.atmosphere.gases:
            varname = "{}_absorption".format(name)
            values = results.get(name)
            if mode == "all":
                if values is None:
                    values = np.zeros([levels, len(MECHANISMS), columns])
                    for slot in range(len(MECHANISMS)):
                        if (name, slot) in blocks:
                            values[:, slot, :] = blocks[(name, slot)]
                beta[varname] = values
            else:
                if values is None:
                    values = np.zeros((levels, columns))
                    for slot in range(len(MECHANISMS)):
                        if (name, slot) in blocks:
                            values += blocks[(name, slot)]
                beta[varname] = values
        return beta

    def _create_output_dataset(self, absorption, output_format):
        dims = list(self.output.dims)
        if output_format == "all":
            variables = dict(absorption)
            extra = {"mechanism": np.asarray(self.output.mechanisms)}
        elif output_format == "gas":
            dims.pop(-2)
            variables = dict(absorption)
            extra = {}
        else:
            dims.pop(-2)
            parts = list(absorption.values())
            variables = {"absorption": parts[0] if len(parts) == 1 else sum(parts)} \
                if parts else {}
            extra = {}
        xarray = _optional_xarray()
        if xarray is None:
            out = {"wavenumber": self.grid}
            out.update(extra)
            out.update(variables)
            return out
        DataArray, Dataset = xarray.DataArray, xarray.Dataset
        data_vars = {"wavenumber": DataArray(self.grid, dims=("wavenumber",),
                                             attrs={"units": "cm-1"})}
        for key, value in extra.items():
            data_vars[key] = DataArray(value, dims=("mechanism",))
        for key, value in variables.items():
            data_vars[key] = DataArray(value, dims=dims, attrs=self.output.units)
        return Dataset(data_vars=data_vars)
