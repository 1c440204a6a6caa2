"""ctypes binding of the C ABI in include/lbl_amd.h (pylbl_amd/liblbl_amd.so).

This is the only Python<->native boundary of the package, the counterpart of
pyLBL/c_lib/gas_optics.py:11-26,68-91 in the reference.  There is no CPU fallback: if the
library is missing or no MI355X is visible, creating an Engine raises.
"""
from ctypes import CDLL, POINTER, Structure, byref, c_char_p, c_double, c_int32, c_int64, \
                   c_void_p
import os
from pathlib import Path
import threading
import weakref

import numpy as np

from .errors import EngineError

LIBRARY_PATH = Path(__file__).resolve().parent / "liblbl_amd.so"

# Mirrors of the #defines in include/lbl_amd.h.
LBL_OK = 0
RANGE_REFERENCE, RANGE_SKIP = 0, 1
PREP_DEVICE, PREP_HOST = 0, 1
OUT_DEVICE, ASYNC, SCALE_DENSITY, ACCUMULATE, FARFIELD, DEFER_FINISH = 1, 2, 4, 8, 16, 32
RANGE_POLICIES = {"reference": RANGE_REFERENCE, "skip": RANGE_SKIP}

EXPORTED_SYMBOLS = (
    "lbl_engine_create", "lbl_engine_destroy", "lbl_last_error", "lbl_molecule_load",
    "lbl_molecule_free", "lbl_compute", "lbl_compute_streamed", "lbl_finish_deferred",
    "lbl_deferred", "lbl_cancel_deferred", "lbl_synchronize", "lbl_set_option", "lbl_timing",
    "lbl_timing_busy",
    "lbl_stream", "lbl_order_stream_after_engine", "lbl_order_engine_after_stream",
    "lbl_device_alloc", "lbl_device_free", "lbl_copy_to_host",
    "lbl_copy_rows_to_host", "lbl_host_alloc", "lbl_host_free",
    "lbl_line_scalars", "lbl_absorption", "absorption", "lbl_compat_state", "lbl_fill_zero",
    "lbl_version", "lbl_table_read", "lbl_table_shape", "lbl_table_copy", "lbl_table_free",
    "lbl_molecule_load_sqlite",
    "lbl_continuum_load", "lbl_continuum_free", "lbl_grid_load", "lbl_grid_free",
    "lbl_continuum_compute", "lbl_continuum_compute_many", "lbl_continuum_bands",
    "lbl_xsec_load", "lbl_xsec_free", "lbl_xsec_compute", "lbl_xsec_bands",
)

VMR_SELF, VMR_H2O, VMR_O2, VMR_N2, VMR_TOTAL, VMR_COUNT = 0, 1, 2, 3, 4, 5
MAX_BANDS = 8
MAX_XSEC_BANDS = 16


class BandDescriptor(Structure):
    """struct lbl_band of include/lbl_amd.h."""
    _fields_ = [("kind", c_int32), ("size", c_int32), ("lower_bound", c_double),
                ("resolution", c_double), ("column", c_int64*4)]

_library = None


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels ship their own
    libamdhip64.so.7 (same SONAME as /opt/rocm's): when torch is imported first, this library
    binds to torch's copy and all is well; the other way round torch finds the system runtime
    already resident beside its own HSA libraries and sees no GPU.  So when a ROCm torch is
    installed but not imported yet, its runtime is loaded here first (no torch import: only the
    shared object), which makes the order irrelevant."""
    import importlib.util
    import os
    import sys
    from ctypes import RTLD_GLOBAL
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for location in spec.submodule_search_locations:
        candidate = os.path.join(location, "lib", "libamdhip64.so")
        if os.path.exists(candidate):
            try:
                CDLL(candidate, mode=RTLD_GLOBAL)
            except OSError:
                pass
            return


def library():
    """Loads liblbl_amd.so (once) and declares the argument types of every entry point."""
    global _library
    if _library is not None:
        return _library
    path = LIBRARY_PATH
    if os.environ.get("PYLBL_AMD_LIBRARY"):
        # Another build of the same engine (sanitizer / diagnostics builds, A/B of two libraries):
        # the shipped file is never overwritten to try one.
        path = Path(os.environ["PYLBL_AMD_LIBRARY"]).resolve()
        if not path.exists():
            raise EngineError(f"$PYLBL_AMD_LIBRARY names {path}, which does not exist.")
    elif not LIBRARY_PATH.exists():
        # A fresh checkout: compile in-tree (hipcc cross-compiles without a GPU).
        try:
            from . import build
            build.build()
        except Exception as error:
            raise EngineError(
                f"{LIBRARY_PATH} is missing and could not be built ({error}); build it with "
                "`python -m pylbl_amd.build` (there is no CPU fallback).")
    _preload_hip_runtime()
    lib = CDLL(str(path))
    f64p, i32p, i64p = POINTER(c_double), POINTER(c_int32), POINTER(c_int64)
    lib.lbl_engine_create.argtypes = [c_int32, POINTER(c_void_p)]
    lib.lbl_engine_destroy.argtypes = [c_void_p]
    lib.lbl_last_error.argtypes = [c_void_p]
    lib.lbl_last_error.restype = c_char_p
    lib.lbl_molecule_load.argtypes = [c_void_p, c_int64] + [c_void_p]*7 + [c_void_p, c_void_p,
                                     c_int32, c_int32, c_void_p, c_void_p, i32p]
    lib.lbl_molecule_free.argtypes = [c_void_p, c_int32]
    lib.lbl_compute.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p] + \
                               [c_int32]*7 + [c_void_p, c_int64, i64p]
    lib.lbl_compute_streamed.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                         c_void_p] + [c_int32]*7 + [c_void_p, c_int64, c_void_p,
                                                                    c_int64, c_int64, c_int32]
    lib.lbl_finish_deferred.argtypes = [c_void_p]
    lib.lbl_deferred.argtypes = [c_void_p]
    lib.lbl_cancel_deferred.argtypes = [c_void_p]
    lib.lbl_synchronize.argtypes = [c_void_p]
    lib.lbl_set_option.argtypes = [c_void_p, c_char_p, c_int64]
    lib.lbl_timing.argtypes = [c_void_p, f64p, i64p, c_int32]
    lib.lbl_timing_busy.argtypes = [c_void_p, f64p]
    lib.lbl_stream.argtypes = [c_void_p]
    lib.lbl_stream.restype = c_void_p
    lib.lbl_order_stream_after_engine.argtypes = [c_void_p, c_void_p]
    lib.lbl_order_engine_after_stream.argtypes = [c_void_p, c_void_p]
    lib.lbl_device_alloc.argtypes = [c_void_p, c_int64, POINTER(c_void_p)]
    lib.lbl_device_free.argtypes = [c_void_p, c_void_p]
    lib.lbl_copy_to_host.argtypes = [c_void_p, c_void_p, c_void_p, c_int64]
    lib.lbl_copy_rows_to_host.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                          c_int64, c_int64, c_int32]
    lib.lbl_host_alloc.argtypes = [c_void_p, c_int64, POINTER(c_void_p)]
    lib.lbl_host_free.argtypes = [c_void_p, c_void_p]
    lib.lbl_line_scalars.argtypes = [c_void_p, c_int32] + [c_double]*3 + [c_int32]*6 + [c_void_p]
    lib.lbl_absorption.argtypes = [c_double]*3 + [c_int32]*3 + [c_void_p, c_char_p, c_char_p,
                                  c_int32, c_int32]
    lib.absorption.argtypes = lib.lbl_absorption.argtypes
    lib.lbl_compat_state.argtypes = [i32p, i32p]
    lib.lbl_table_read.argtypes = [c_char_p, c_char_p, POINTER(c_void_p)]
    lib.lbl_table_shape.argtypes = [c_void_p, i64p, i32p, i32p, i32p, i32p, c_char_p, c_int32]
    lib.lbl_table_copy.argtypes = [c_void_p] + [c_void_p]*6
    lib.lbl_table_free.argtypes = [c_void_p]
    lib.lbl_molecule_load_sqlite.argtypes = [c_void_p, c_char_p, c_char_p, i32p]
    lib.lbl_fill_zero.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32]
    lib.lbl_version.restype = c_char_p
    lib.lbl_continuum_load.argtypes = [c_void_p, c_int32, POINTER(BandDescriptor), c_void_p,
                                       c_int64, i32p]
    lib.lbl_continuum_free.argtypes = [c_void_p, c_int32]
    lib.lbl_grid_load.argtypes = [c_void_p, c_int64, c_void_p, i32p]
    lib.lbl_grid_free.argtypes = [c_void_p, c_int32]
    lib.lbl_continuum_compute.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_void_p, c_int64]
    lib.lbl_continuum_compute_many.argtypes = [c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                               c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                               c_int64]
    lib.lbl_continuum_bands.argtypes = [c_void_p, c_int32, c_double, c_double, c_void_p,
                                        c_void_p]
    lib.lbl_xsec_load.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, i32p]
    lib.lbl_xsec_free.argtypes = [c_void_p, c_int32]
    lib.lbl_xsec_compute.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                     c_void_p, c_int32, c_void_p, c_int64]
    lib.lbl_xsec_bands.argtypes = [c_void_p, c_int32, c_double, c_double, c_void_p]
    for name in EXPORTED_SYMBOLS:
        if name not in ("lbl_last_error", "lbl_stream", "lbl_version"):
            getattr(lib, name).restype = c_int32
    _library = lib
    return lib


def _f64(array):
    return np.ascontiguousarray(array, dtype=np.float64)


# Status codes of lbl_table_read (include/lbl_amd.h).
TABLE_OPEN_FAILED, TABLE_NO_ALIAS, TABLE_NO_TIPS, TABLE_NOT_RECTANGULAR, TABLE_NO_ISOTOPOLOGUES, \
    TABLE_NO_TRANSITIONS = 10, 11, 12, 13, 14, 15


def read_line_table(path, name):
    """One molecule's rows out of an SQLite file in pyLBL's schema through the engine's own C
    reader (lbl_table_read: the reference C reader's SELECTs, absorption.c:69-70,
    spectral_database.c:55, :113, :143) -- no GPU involved.  Returns (status, message, fields):
    status LBL_OK and a dict of arrays, or a TABLE_* status and the reader's message."""
    from ctypes import create_string_buffer
    lib = library()
    table = c_void_p()
    status = lib.lbl_table_read(os.fsencode(str(path)), str(name).encode(), byref(table))
    if status != LBL_OK:
        return status, lib.lbl_last_error(None).decode(), None
    try:
        n_lines, molecule_id = c_int64(), c_int32()
        rows, num_iso, num_t = c_int32(), c_int32(), c_int32()
        formula = create_string_buffer(256)
        lib.lbl_table_shape(table, byref(n_lines), byref(molecule_id), byref(rows), byref(num_iso),
                            byref(num_t), formula, 256)
        columns = np.empty((7, n_lines.value))
        local_iso_id = np.empty(n_lines.value, dtype=np.int32)
        isoid = np.empty(rows.value, dtype=np.int64)
        mass = np.empty(rows.value)
        tips_temperature = np.empty(num_t.value)
        tips_data = np.empty((num_iso.value, num_t.value))
        lib.lbl_table_copy(table, columns.ctypes.data, local_iso_id.ctypes.data, isoid.ctypes.data,
                           mass.ctypes.data, tips_temperature.ctypes.data, tips_data.ctypes.data)
    finally:
        lib.lbl_table_free(table)
    return LBL_OK, "", {"formula": formula.value.decode(), "molecule_id": molecule_id.value,
                        "columns": columns, "local_iso_id": local_iso_id, "isoid": isoid,
                        "mass": mass, "tips_temperature": tips_temperature,
                        "tips_data": tips_data}


def _levels(values):
    """Per-level input as a contiguous 1-d float64 array (no copy, and none of numpy's dispatch,
    when it already is one: a call on a small grid costs the host ~35 us all told)."""
    if type(values) is np.ndarray and values.dtype == np.float64 and values.ndim == 1 and \
            values.flags.c_contiguous:
        return values
    return _f64(np.atleast_1d(values))


class DeviceSpectra(object):
    """Spectra left in HBM: [levels, n] float64 on the engine's GPU."""
    def __init__(self, engine, levels, n):
        self.engine = engine
        self.shape = (int(levels), int(n))
        self.pointer = c_void_p()
        engine._check(engine.lib.lbl_device_alloc(engine.handle, self.shape[0]*self.shape[1]*8,
                                                  byref(self.pointer)))

    def to_host(self):
        out = np.empty(self.shape, dtype=np.float64)
        self.engine._check(self.engine.lib.lbl_copy_to_host(
            self.engine.handle, out.ctypes.data, self.pointer, out.nbytes))
        return out

    def to_host_into(self, target, columns=None, asynchronous=False):
        """Copies the first `columns` values of every row straight into `target`, a float64
        array view [rows, columns] whose rows are contiguous (any row stride), e.g.
        beta[:, mechanism, :].  asynchronous: queue the copy behind everything queued so far
        and return; Engine.synchronize() waits for it (use page-locked targets,
        Engine.host_array, or the copy blocks anyway)."""
        columns = self.shape[1] if columns is None else int(columns)
        if target.dtype != np.float64 or target.shape != (self.shape[0], columns) or \
                columns > self.shape[1] or (columns > 1 and target.strides[1] != 8) or \
                (self.shape[0] > 1 and target.strides[0] < columns*8):
            raise ValueError("target must be float64[rows, columns] with contiguous rows.")
        pitch = target.strides[0] if self.shape[0] > 1 else columns*8
        self.engine._check(self.engine.lib.lbl_copy_rows_to_host(
            self.engine.handle, target.ctypes.data, pitch, self.pointer, self.shape[1]*8,
            columns*8, self.shape[0], ASYNC if asynchronous else 0))
        return target

    def free(self):
        if self.pointer:
            self.engine.lib.lbl_device_free(self.engine.handle, self.pointer)
            self.pointer = c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevicePool(object):
    """[levels, n] blocks in HBM handed out and taken back (hipMalloc / hipFree cost more than
    the continuum kernels that fill such a block, and hipFree stops the device)."""
    def __init__(self, engine, limit=32 << 30):
        self.engine = weakref.ref(engine)
        self.limit = limit
        self.idle = {}          # shape -> [DeviceSpectra]
        self.idle_bytes = 0
        self.lock = threading.RLock()       # blocks are taken and given by any thread

    def take(self, levels, n):
        shape = (int(levels), int(n))
        with self.lock:
            blocks = self.idle.get(shape)
            if blocks:
                self.idle_bytes -= shape[0]*shape[1]*8
                return blocks.pop()
        return DeviceSpectra(self.engine(), *shape)

    def give(self, block):
        size = block.shape[0]*block.shape[1]*8
        with self.lock:
            if block.pointer and self.idle_bytes + size <= self.limit:
                self.idle.setdefault(tuple(block.shape), []).append(block)
                self.idle_bytes += size
                return
        block.free()

    def clear(self):
        with self.lock:
            idle, self.idle, self.idle_bytes = self.idle, {}, 0
        for blocks in idle.values():
            for block in blocks:
                block.free()


class PinnedPool(object):
    """Page-locked host arrays for results.  Pinning memory is slow, so buffers are recycled:
    when the last view of an array handed out here is garbage-collected its buffer goes back
    to the pool (up to `limit` bytes of idle buffers are kept)."""
    def __init__(self, engine, limit=8 << 30):
        self.engine = weakref.ref(engine)
        self.limit = limit
        self.idle = []          # (capacity, pointer)
        self.idle_bytes = 0
        # Arrays are handed out to any thread and come back from whichever thread drops the last
        # view (a finalizer: it may run inside array() on the same thread, hence re-entrant).
        self.lock = threading.RLock()

    def array(self, shape):
        shape = tuple(int(x) for x in shape)
        count = int(np.prod(shape)) if shape else 1
        nbytes = max(count*8, 8)
        engine = self.engine()
        with self.lock:
            best = None
            for i, (capacity, _) in enumerate(self.idle):
                if nbytes <= capacity <= 2*nbytes + (1 << 20) and \
                        (best is None or capacity < self.idle[best][0]):
                    best = i
            if best is not None:
                capacity, pointer = self.idle.pop(best)
                self.idle_bytes -= capacity
        if best is None:
            capacity, handle = nbytes, c_void_p()
            if engine.lib.lbl_host_alloc(engine.handle, capacity, byref(handle)) != LBL_OK:
                # No more page-locked memory to be had (results held by the caller count):
                # ordinary memory still works, copies into it are only slower.
                self.clear()
                return np.empty(shape, dtype=np.float64)
            pointer = handle.value
        from ctypes import c_char
        buffer = (c_char*capacity).from_address(pointer)
        weakref.finalize(buffer, PinnedPool._release, weakref.ref(self), capacity, pointer)
        return np.frombuffer(buffer, dtype=np.float64, count=count).reshape(shape)

    @staticmethod
    def _release(pool, capacity, pointer):
        pool = pool()
        engine = pool.engine() if pool is not None else None
        if engine is None or not engine.handle:
            return                      # engine gone: the runtime reclaims the pages at exit
        with pool.lock:
            if pool.idle_bytes + capacity <= pool.limit:
                pool.idle.append((capacity, pointer))
                pool.idle_bytes += capacity
                return
        engine.lib.lbl_host_free(engine.handle, c_void_p(pointer))

    def clear(self):
        engine = self.engine()
        with self.lock:
            idle, self.idle, self.idle_bytes = self.idle, [], 0
        for _, pointer in idle:
            if engine is not None and engine.handle:
                engine.lib.lbl_host_free(engine.handle, c_void_p(pointer))


class Engine(object):
    """One GPU's engine: resident line tables plus the batched compute call."""
    def __init__(self, device=0):
        self.lib = library()
        self.handle = c_void_p()
        status = self.lib.lbl_engine_create(int(device), byref(self.handle))
        if status != LBL_OK:
            message = self.lib.lbl_last_error(None).decode()
            self.handle = c_void_p()
            raise EngineError(f"lbl_engine_create failed ({status}): {message}")
        self.device = int(device)
        self.pinned = PinnedPool(self)
        self.blocks = DevicePool(self)
        # Held by callers whose result takes SEVERAL calls on this engine that must not be
        # interleaved with another thread's (Spectroscopy.compute_absorption: calls that add into
        # one block in a fixed order, one deferred call per engine).  Single calls need no lock:
        # the C ABI serialises them per handle (include/lbl_amd.h, "Threads").
        self.pipeline = threading.RLock()
        # Options for every engine of the process, for experiments: PYLBL_AMD_OPTIONS="name=value,..."
        # (kept in `environment_options` so that whoever reports numbers can say so: bench.py
        # echoes them).  "ablate" leaves work out -- results are wrong -- and is refused here: a
        # left-over environment variable must not change what every Gas object returns.
        self.environment_options = {}
        for pair in filter(None, os.environ.get("PYLBL_AMD_OPTIONS", "").split(",")):
            name, _, value = pair.partition("=")
            name = name.strip()
            if name == "ablate":
                self.close()
                raise EngineError("PYLBL_AMD_OPTIONS must not set 'ablate' (results would be "
                                  "wrong for every engine of the process); use "
                                  "Engine.set_option('ablate', ...) on the one engine being timed.")
            self.set_option(name, float(value))
            self.environment_options[name] = float(value)

    def host_array(self, shape):
        """float64 array of the given shape in page-locked host memory (recycled, see
        PinnedPool): the place to receive spectra from HBM."""
        return self.pinned.array(shape)

    def _check(self, status):
        if status != LBL_OK:
            raise EngineError(f"status {status}: {self.lib.lbl_last_error(self.handle).decode()}")

    def close(self):
        if self.handle:
            self.pinned.clear()
            self.blocks.clear()
            self.lib.lbl_engine_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        self._check(self.lib.lbl_set_option(self.handle, name.encode(), int(value)))

    def load(self, table):
        """Uploads a pylbl_amd.database.LineTable; returns the molecule handle."""
        columns = [_f64(getattr(table, x)) for x in
                   ("nu", "sw", "gamma_air", "gamma_self", "n_air", "elower", "delta_air")]
        iso = np.ascontiguousarray(table.local_iso_id, dtype=np.int32)
        mass = _f64(table.mass_by_slot())
        tips_t = _f64(table.tips_temperature)
        tips_q = _f64(table.tips_data)
        if tips_q.ndim != 2 or tips_q.shape[1] != tips_t.size:
            raise ValueError("tips_data must be [num_iso, num_t] with num_t temperatures.")
        handle = c_int32(-1)
        self._check(self.lib.lbl_molecule_load(
            self.handle, columns[0].size, *[x.ctypes.data for x in columns], iso.ctypes.data,
            mass.ctypes.data, tips_q.shape[0], tips_q.shape[1], tips_t.ctypes.data,
            tips_q.ctypes.data, byref(handle)))
        return handle.value

    def load_sqlite(self, path, name):
        """File -> HBM in one call (lbl_molecule_load_sqlite): molecule `name` (any alias) of the
        SQLite file at `path` in pyLBL's schema, read by the C reader and uploaded; returns the
        molecule handle.  Raises EngineError with the reader's message when the file lacks
        something (status LBL_TABLE_*, include/lbl_amd.h)."""
        handle = c_int32(-1)
        self._check(self.lib.lbl_molecule_load_sqlite(self.handle, os.fsencode(str(path)),
                                                      str(name).encode(), byref(handle)))
        return handle.value

    def free(self, molecule):
        self._check(self.lib.lbl_molecule_free(self.handle, int(molecule)))

    def compute(self, molecule, temperature, pressure, vmr, v0, vn, n_per_v, cut_off=25,
                remove_pedestal=False, range_policy="reference", out=None, scale_density=False,
                accumulate=False, asynchronous=False, want_evals=False, farfield=False,
                deliver=None, pieces=4, defer_finish=False):
        """Cross sections [m2] for every level: returns float64[levels, (vn-v0)*n_per_v]
        (or fills `out`: a host array or a DeviceSpectra).

        deliver: (until engine.synchronize() the delivered part of `out` must not be written again:
        calls are ordered by the memory they write, not by what a copy still reads.)
        With a device `out`, a float64 [levels, columns] host view with contiguous rows
        (page-locked: Engine.host_array) that receives the first `columns` points of every level
        while the call computes, in `pieces` runs of tiles (lbl_compute_streamed).
        defer_finish: keep the call's last kernels (the ones that touch `out`) back until
        finish_deferred() / synchronize(): LBL_DEFER_FINISH."""
        t, p, x = _levels(temperature), _levels(pressure), _levels(vmr)
        if not (t.shape == p.shape == x.shape and t.ndim == 1):
            raise ValueError("temperature, pressure and vmr must be 1-d and equally long.")
        n = (int(vn) - int(v0))*int(n_per_v)
        flags = (SCALE_DENSITY if scale_density else 0) | (ACCUMULATE if accumulate else 0) | \
                (ASYNC if asynchronous else 0) | (FARFIELD if farfield else 0) | \
                (DEFER_FINISH if defer_finish else 0)
        if hasattr(out, "pointer"):
            # Device memory: a DeviceSpectra or anything exposing .pointer and .shape.
            if tuple(out.shape) != (t.size, n):
                raise ValueError(f"out has shape {out.shape}, need {(t.size, n)}.")
            pointer, flags = out.pointer, flags | OUT_DEVICE
        else:
            if out is None:
                # Page-locked and recycled: the copy back runs at the rate of the host link.
                out = self.host_array((t.size, max(n, 0)))
                if accumulate:
                    out[...] = 0.
            if out.shape != (t.size, n) or out.dtype != np.float64 or \
                    not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be C-contiguous float64[levels, n].")
            pointer = c_void_p(out.ctypes.data)
        if deliver is not None:
            if not (flags & OUT_DEVICE) or want_evals:
                raise ValueError("deliver needs a device `out` (and no eval count).")
            if deliver.ndim != 2 or deliver.shape[0] != t.size or deliver.dtype != np.float64 \
                    or deliver.strides[1] != 8 or deliver.shape[1] > n:
                raise ValueError("deliver must be float64[levels, columns <= n], rows contiguous.")
            self._check(self.lib.lbl_compute_streamed(
                self.handle, int(molecule), t.size, t.ctypes.data, p.ctypes.data, x.ctypes.data,
                int(v0), int(vn), int(n_per_v), int(cut_off), 1 if remove_pedestal else 0,
                RANGE_POLICIES[range_policy], flags, pointer, 0, c_void_p(deliver.ctypes.data),
                int(deliver.strides[0]) if t.size > 1 else max(int(deliver.strides[0]),
                                                               8*deliver.shape[1]),
                int(deliver.shape[1]), int(pieces)))
            return out
        evals = c_int64(0)
        self._check(self.lib.lbl_compute(
            self.handle, int(molecule), t.size, t.ctypes.data, p.ctypes.data, x.ctypes.data,
            int(v0), int(vn), int(n_per_v), int(cut_off), 1 if remove_pedestal else 0,
            RANGE_POLICIES[range_policy], flags, pointer, 0,
            byref(evals) if want_evals else None))
        return (out, evals.value) if want_evals else out

    def fill_zero(self, out, asynchronous=False):
        """Zeroes a [levels, n] output (host array or DeviceSpectra): the spectrum the
        reference returns for a molecule it has nothing to compute for (absorption.c:41)."""
        if hasattr(out, "pointer"):
            self._check(self.lib.lbl_fill_zero(
                self.handle, out.pointer, int(out.shape[0]), int(out.shape[1]), 0,
                OUT_DEVICE | (ASYNC if asynchronous else 0)))
        else:
            out[...] = 0.
        return out

    def line_scalars(self, molecule, num_lines, temperature, pressure, vmr, v0, vn, n_per_v,
                     cut_off=25, range_policy="reference"):
        """Per-line derived scalars in reference row order (see lbl_line_scalars)."""
        derived = np.zeros((max(int(num_lines), 1), 8), dtype=np.float64)
        self._check(self.lib.lbl_line_scalars(
            self.handle, int(molecule), float(temperature), float(pressure), float(vmr),
            int(v0), int(vn), int(n_per_v), int(cut_off), 0, RANGE_POLICIES[range_policy],
            derived.ctypes.data))
        return derived[:int(num_lines)]

    def finish_deferred(self):
        """Queues what a call with defer_finish=True kept back."""
        self._check(self.lib.lbl_finish_deferred(self.handle))

    def deferred(self):
        """True while a call is kept back (False right after a call whose defer_finish could
        not be honoured: no pedestal pass, several level passes, host output)."""
        return bool(self.lib.lbl_deferred(self.handle))

    def cancel_deferred(self):
        """Drops what a call with defer_finish=True kept back: its `out` block and `deliver`
        array are never written by it (for callers that fail before finish_deferred())."""
        self._check(self.lib.lbl_cancel_deferred(self.handle))

    def synchronize(self):
        self._check(self.lib.lbl_synchronize(self.handle))

    # -- continua (slot 1) -----------------------------------------------------------------
    def load_continuum(self, bands):
        """Uploads a list of bands, each (kind, lower_bound, resolution, [columns]); returns
        the continuum handle."""
        if not 1 <= len(bands) <= MAX_BANDS:
            raise ValueError(f"a continuum has 1 to {MAX_BANDS} bands.")
        descriptors = (BandDescriptor*len(bands))()
        pieces, offset = [], 0
        for d, (kind, lower, resolution, columns) in zip(descriptors, bands):
            columns = [_f64(c) for c in columns]
            size = columns[0].size
            if any(c.shape != (size,) for c in columns) or len(columns) > 4:
                raise ValueError("the columns of a band must be 1-d and equally long (<= 4).")
            d.kind, d.size, d.lower_bound, d.resolution = int(kind), size, float(lower), \
                float(resolution)
            for i in range(4):
                d.column[i] = -1
            for i, c in enumerate(columns):
                d.column[i] = offset
                offset += size
                pieces.append(c)
        table = _f64(np.concatenate(pieces))
        handle = c_int32(-1)
        self._check(self.lib.lbl_continuum_load(self.handle, len(bands), descriptors,
                                                table.ctypes.data, table.size, byref(handle)))
        return handle.value

    def free_continuum(self, continuum):
        self._check(self.lib.lbl_continuum_free(self.handle, int(continuum)))

    def load_grid(self, wavenumber):
        """Uploads a spectral grid [cm-1]; returns the grid handle."""
        grid = _f64(wavenumber)
        if grid.ndim != 1 or grid.size < 1:
            raise ValueError("the grid must be a non-empty 1-d array.")
        handle = c_int32(-1)
        self._check(self.lib.lbl_grid_load(self.handle, grid.size, grid.ctypes.data,
                                           byref(handle)))
        return handle.value

    def free_grid(self, grid):
        self._check(self.lib.lbl_grid_free(self.handle, int(grid)))

    def continuum_compute(self, continuum, grid, n, temperature, pressure, vmr, out=None,
                          accumulate=False, asynchronous=False):
        """Continuum extinction [m-1]: float64[levels, n] (or fills `out`, host array or
        DeviceSpectra).  vmr is [levels, VMR_COUNT]; pressure in Pa."""
        t, p = _f64(np.atleast_1d(temperature)), _f64(np.atleast_1d(pressure))
        x = _f64(vmr).reshape(-1, VMR_COUNT)
        if not (t.ndim == 1 and t.shape == p.shape and x.shape[0] == t.size):
            raise ValueError("temperature, pressure [levels] and vmr [levels, 5] disagree.")
        flags = (ACCUMULATE if accumulate else 0) | (ASYNC if asynchronous else 0)
        out, pointer, flags, stride = self._output(out, t.size, n, flags)
        self._check(self.lib.lbl_continuum_compute(
            self.handle, int(continuum), int(grid), t.size, t.ctypes.data, p.ctypes.data,
            x.ctypes.data, flags, pointer, stride))
        return out

    def continuum_compute_many(self, continua, grid, n, temperature, pressure, vmr, out,
                               accumulate=False, asynchronous=False):
        """Several continua in one pass over the grid (lbl_continuum_compute_many): summed, in
        the order given, into the DeviceSpectra `out` [levels, >= n] -- the same bits as one
        continuum_compute per handle, at a third of the HBM traffic for three of them.
        vmr: [len(continua), levels, VMR_COUNT]."""
        t, p = _f64(np.atleast_1d(temperature)), _f64(np.atleast_1d(pressure))
        handles = np.ascontiguousarray(continua, dtype=np.int32)
        x = _f64(vmr).reshape(handles.size, -1, VMR_COUNT)
        if not (t.ndim == 1 and t.shape == p.shape and x.shape[1] == t.size):
            raise ValueError("temperature, pressure [levels] and vmr [continua, levels, 5] disagree.")
        if not hasattr(out, "pointer"):
            raise ValueError("continuum_compute_many writes a block in HBM (DeviceSpectra).")
        flags = (ACCUMULATE if accumulate else 0) | (ASYNC if asynchronous else 0)
        out, pointer, flags, stride = self._output(out, t.size, n, flags)
        self._check(self.lib.lbl_continuum_compute_many(
            self.handle, handles.size, handles.ctypes.data, int(grid), t.size, t.ctypes.data,
            p.ctypes.data, x.ctypes.data, flags, pointer, stride))
        return out

    def _output(self, out, levels, n, flags):
        """(array or DeviceSpectra, pointer, flags, row stride) for a [levels, >= n] block."""
        if out is None:
            out = self.host_array((levels, n))
            if flags & ACCUMULATE:
                out[...] = 0.
        # Rows may be longer than the grid (the lines path pads them to whole wavenumbers).
        if len(out.shape) != 2 or out.shape[0] != levels or out.shape[1] < n:
            raise ValueError(f"out has shape {out.shape}, need ({levels}, >= {n}).")
        if hasattr(out, "pointer"):
            return out, out.pointer, flags | OUT_DEVICE, int(out.shape[1])
        if out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be C-contiguous float64[levels, >= n].")
        return out, c_void_p(out.ctypes.data), flags, int(out.shape[1])

    # -- cross-sections (slot 2) -----------------------------------------------------------
    def load_xsec(self, bands):
        """Uploads [(frequency [Hz], coefficients [4, nfreq]), ...]; returns the handle."""
        if not 1 <= len(bands) <= MAX_XSEC_BANDS:
            raise ValueError(f"a molecule has 1 to {MAX_XSEC_BANDS} cross-section bands.")
        sizes = np.zeros(len(bands), dtype=np.int32)
        frequency, coefficients = [], []
        for i, (f, c) in enumerate(bands):
            f, c = _f64(f), _f64(c)
            if f.ndim != 1 or c.shape != (4, f.size):
                raise ValueError("a band is (frequency[nfreq], coefficients[4, nfreq]).")
            sizes[i] = f.size
            frequency.append(f)
            coefficients.append(c.ravel())
        frequency, coefficients = _f64(np.concatenate(frequency)), _f64(np.concatenate(coefficients))
        handle = c_int32(-1)
        self._check(self.lib.lbl_xsec_load(self.handle, len(bands), sizes.ctypes.data,
                                           frequency.ctypes.data, coefficients.ctypes.data,
                                           byref(handle)))
        return handle.value

    def free_xsec(self, xsec):
        self._check(self.lib.lbl_xsec_free(self.handle, int(xsec)))

    def xsec_compute(self, xsec, grid, n, temperature, pressure, vmr=None, out=None,
                     accumulate=False, asynchronous=False):
        """Cross sections [m2] (or, with vmr, n k [m-1]): float64[levels, n] or fills `out`."""
        t, p = _f64(np.atleast_1d(temperature)), _f64(np.atleast_1d(pressure))
        if not (t.ndim == 1 and t.shape == p.shape):
            raise ValueError("temperature and pressure must be 1-d and equally long.")
        flags = (ACCUMULATE if accumulate else 0) | (ASYNC if asynchronous else 0)
        x = None
        if vmr is not None:
            x = _f64(np.atleast_1d(vmr))
            if x.shape != t.shape:
                raise ValueError("vmr must be shaped like temperature.")
            flags |= SCALE_DENSITY
        out, pointer, flags, stride = self._output(out, t.size, n, flags)
        self._check(self.lib.lbl_xsec_compute(
            self.handle, int(xsec), int(grid), t.size, t.ctypes.data, p.ctypes.data,
            x.ctypes.data if x is not None else None, flags, pointer, stride))
        return out

    def xsec_bands(self, xsec, sizes, temperature, pressure):
        """The clipped fit on the bands' own grids for one level (list of arrays)."""
        values = np.zeros(int(sum(sizes)), dtype=np.float64)
        self._check(self.lib.lbl_xsec_bands(self.handle, int(xsec), float(temperature),
                                            float(pressure), values.ctypes.data))
        return np.split(values, np.cumsum(sizes)[:-1])

    def continuum_bands(self, continuum, sizes, temperature, pressure_mb, vmr):
        """Coarse spectra [cm-1] of each band for one level (list of arrays)."""
        x = _f64(vmr).reshape(VMR_COUNT)
        spectra = np.zeros(int(sum(sizes)), dtype=np.float64)
        self._check(self.lib.lbl_continuum_bands(self.handle, int(continuum), float(temperature),
                                                 float(pressure_mb), x.ctypes.data,
                                                 spectra.ctypes.data))
        return np.split(spectra, np.cumsum(sizes)[:-1])

    def timing(self, reset=False):
        """(milliseconds[8], launches[8]) for prepare (prologue), far-field series (and the tile
        schedule of host-side prep), accumulate, pedestal, continuum band spectra, continuum
        interpolation, cross-section fit, cross-section interpolation."""
        ms = (c_double*8)()
        launches = (c_int64*8)()
        self._check(self.lib.lbl_timing(self.handle, ms, launches, 1 if reset else 0))
        return list(ms), list(launches)

    def timing_busy(self):
        """milliseconds[8] (indices as timing()) during which at least one timed launch of the kind
        was running since the last reset: launches of calls on different lanes overlap, and what
        timing() sums twice this counts once.  Read it before timing(reset=True)."""
        ms = (c_double*8)()
        self._check(self.lib.lbl_timing_busy(self.handle, ms))
        return list(ms)

    @property
    def stream(self):
        return self.lib.lbl_stream(self.handle)

    def order_stream_after(self, stream):
        """Work queued on `stream` (a raw hipStream_t, e.g. torch.cuda.Stream.cuda_stream) from
        now on runs after everything queued on the engine so far; the host does not wait."""
        self._check(self.lib.lbl_order_stream_after_engine(self.handle, c_void_p(stream or None)))

    def order_after_stream(self, stream):
        """Everything the engine queues from now on runs after what `stream` holds now."""
        self._check(self.lib.lbl_order_engine_after_stream(self.handle, c_void_p(stream or None)))


_default_engines = {}
_default_engines_lock = threading.Lock()


def default_engine(device=0):
    """Process-wide engine per device (what Gas objects share, from any thread)."""
    with _default_engines_lock:
        if device not in _default_engines:
            _default_engines[device] = Engine(device)
        return _default_engines[device]
