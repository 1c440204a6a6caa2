"""Minimal reader for netCDF-4 (HDF5) files through the HDF5 C library and ctypes.

The reference reads its coefficient files with netCDF4-python / xarray
(pyLBL/mt_ckd/utils.py:127-134, pyLBL/arts_crossfit/cross_section.py:29); neither is part of
this image, libhdf5 is.  Only what those readers need: numeric variables as float64 arrays in
their stored shape, scalar numeric attributes, and the names of the root group.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_library = None


def library():
    global _library
    if _library is not None:
        return _library
    lib = None
    for name in (ctypes.util.find_library("hdf5"), "libhdf5.so", "/opt/conda/lib/libhdf5.so"):
        if not name:
            continue
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if lib is None:
        raise OSError("the HDF5 C library (libhdf5.so) was not found; convert the data set to "
                      ".npz on a machine that has it.")
    lib.H5open()
    lib.H5Eset_auto2.argtypes = [_hid, ctypes.c_void_p, ctypes.c_void_p]
    lib.H5Eset_auto2(0, None, None)         # failures are reported through return values
    for name, restype, argtypes in (
            ("H5Fopen", _hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
            ("H5Fclose", ctypes.c_int, [_hid]),
            ("H5Dopen2", _hid, [_hid, ctypes.c_char_p, _hid]),
            ("H5Dclose", ctypes.c_int, [_hid]),
            ("H5Dget_space", _hid, [_hid]),
            ("H5Sclose", ctypes.c_int, [_hid]),
            ("H5Sget_simple_extent_ndims", ctypes.c_int, [_hid]),
            ("H5Sget_simple_extent_dims", ctypes.c_int,
             [_hid, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
            ("H5Dread", ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            ("H5Aopen", _hid, [_hid, ctypes.c_char_p, _hid]),
            ("H5Aread", ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
            ("H5Aclose", ctypes.c_int, [_hid])):
        function = getattr(lib, name)
        function.restype, function.argtypes = restype, argtypes
    lib.native_double = _hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    _library = lib
    return lib


class File(object):
    """Read-only HDF5 file; use as a context manager."""
    def __init__(self, path):
        self.lib = library()
        self.path = str(path)
        self.handle = self.lib.H5Fopen(os.fsencode(self.path), 0, 0)
        if self.handle < 0:
            raise OSError(f"cannot open {self.path} as HDF5.")

    def __enter__(self):
        return self

    def __exit__(self, *unused):
        self.close()

    def close(self):
        if self.handle >= 0:
            self.lib.H5Fclose(self.handle)
            self.handle = -1

    def _dataset(self, name):
        dataset = self.lib.H5Dopen2(self.handle, name.encode(), 0)
        if dataset < 0:
            raise KeyError(f"variable {name} not found in {self.path}.")
        return dataset

    def has(self, name):
        dataset = self.lib.H5Dopen2(self.handle, name.encode(), 0)
        if dataset < 0:
            return False
        self.lib.H5Dclose(dataset)
        return True

    def array(self, name):
        """The variable as float64 in its stored shape."""
        dataset = self._dataset(name)
        try:
            space = self.lib.H5Dget_space(dataset)
            rank = self.lib.H5Sget_simple_extent_ndims(space)
            dims = (ctypes.c_uint64*max(rank, 1))()
            if rank > 0:
                self.lib.H5Sget_simple_extent_dims(space, dims, None)
            self.lib.H5Sclose(space)
            shape = tuple(int(dims[i]) for i in range(rank))
            data = np.zeros(shape, dtype=np.float64)
            if data.size and self.lib.H5Dread(dataset, self.lib.native_double, 0, 0, 0,
                                              data.ctypes.data) < 0:
                raise OSError(f"cannot read variable {name} of {self.path}.")
            return data
        finally:
            self.lib.H5Dclose(dataset)

    def attribute(self, name, attribute):
        """A scalar numeric attribute of a variable as float."""
        dataset = self._dataset(name)
        try:
            handle = self.lib.H5Aopen(dataset, attribute.encode(), 0)
            if handle < 0:
                raise KeyError(f"variable {name} has no attribute {attribute}.")
            value = ctypes.c_double()
            self.lib.H5Aread(handle, self.lib.native_double, ctypes.byref(value))
            self.lib.H5Aclose(handle)
            return value.value
        finally:
            self.lib.H5Dclose(dataset)
