"""Test helper: writes a few numeric arrays into a new HDF5 file through libhdf5 + ctypes
(h5py and netCDF4 are not in the image), so that pylbl_amd.hdf5_reader can be exercised on a
real file with 1-D, 2-D and integer variables."""
import ctypes
import os

import numpy as np

from pylbl_amd import hdf5_reader

_hid = ctypes.c_int64


def write(path, arrays):
    """arrays: name -> ndarray (float64 or int32; stored with that type and shape)."""
    lib = hdf5_reader.library()
    lib.H5Fcreate.restype = _hid
    lib.H5Fcreate.argtypes = [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]
    lib.H5Screate_simple.restype = _hid
    lib.H5Screate_simple.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_uint64)]
    lib.H5Dcreate2.restype = _hid
    lib.H5Dcreate2.argtypes = [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]
    lib.H5Dwrite.restype = ctypes.c_int
    lib.H5Dwrite.argtypes = [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]
    native = {np.dtype(np.float64): _hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value,
              np.dtype(np.int32): _hid.in_dll(lib, "H5T_NATIVE_INT_g").value}
    handle = lib.H5Fcreate(os.fsencode(str(path)), 2, 0, 0)       # H5F_ACC_TRUNC
    if handle < 0:
        raise OSError(f"cannot create {path}.")
    try:
        for name, values in arrays.items():
            values = np.ascontiguousarray(values)
            dims = (ctypes.c_uint64*values.ndim)(*values.shape)
            space = lib.H5Screate_simple(values.ndim, dims, None)
            dataset = lib.H5Dcreate2(handle, name.encode(), native[values.dtype], space, 0, 0, 0)
            if dataset < 0 or lib.H5Dwrite(dataset, native[values.dtype], 0, 0, 0,
                                           values.ctypes.data) < 0:
                raise OSError(f"cannot write {name}.")
            lib.H5Dclose(dataset)
            lib.H5Sclose(space)
    finally:
        lib.H5Fclose(handle)
