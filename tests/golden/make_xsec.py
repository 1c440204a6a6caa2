"""Golden vectors for the ARTS-crossfit fit model: outputs of the reference's own
calculate_xsec_fullmodel (pyLBL/arts_crossfit/xsec_aux_functions.py:80-121, pure numpy, loaded
from /root/reference by file path) for seeded synthetic coefficient sets.

The reference's coefficient files are a download (arts_crossfit/webapi.py) and its
CrossSection class needs xarray, so the interpolation stage is not run here; it is
scipy.interpolate.interp1d, which the oracle calls directly.
Run here:  python tests/golden/make_xsec.py
"""
import importlib.util
import os

import numpy as np

SOURCE = "/root/reference/pyLBL/arts_crossfit/xsec_aux_functions.py"
TARGET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xsec_model.npz")


def coefficient_set(seed, size, offset):
    """A band whose fit goes negative in places: p00 carries the shape, the T, P and T^2
    terms are small corrections, `offset` shifts the whole band up or down."""
    rng = np.random.default_rng(seed)
    f = np.linspace(0., 1., size)
    shape = 1e-22*(np.exp(-((f - 0.4)/0.1)**2) + 0.5*np.exp(-((f - 0.7)/0.05)**2))
    coeffs = np.zeros((4, size))
    coeffs[0] = shape + offset + 2e-24*rng.standard_normal(size)
    coeffs[1] = 1e-26*rng.standard_normal(size)
    coeffs[2] = 1e-29*rng.standard_normal(size)
    coeffs[3] = 1e-29*rng.standard_normal(size)
    return coeffs


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("xsec_aux_functions", SOURCE)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    arrays = {}
    case = 0
    states = [(288.99, 98388.), (203.37, 11419.), (269.01, 117.), (320., 101325.)]
    for seed, size, offset in ((1, 257, 0.), (2, 1000, 5e-24), (3, 64, -3e-23), (4, 513, -2e-22),
                               (5, 2048, 1e-23), (6, 128, 5e-23)):
        coeffs = coefficient_set(seed, size, offset)
        arrays[f"set{seed}_coeffs"] = coeffs
        for temperature, pressure in states:
            out = module.calculate_xsec_fullmodel(temperature, pressure, coeffs.copy())
            raw = module.calculate_xsec(temperature, pressure, coeffs.copy())
            arrays[f"case{case}_set"] = np.asarray(seed)
            arrays[f"case{case}_state"] = np.asarray([temperature, pressure])
            arrays[f"case{case}_xsec"] = out
            print(f"case {case}: size {size} negatives {int((raw < 0).sum())} sum raw {raw.sum():.3e} "
                  f"sum out {out.sum():.3e}")
            case += 1
    arrays["cases"] = np.asarray(case)
    np.savez_compressed(TARGET, **arrays)
    print(TARGET, os.path.getsize(TARGET), "bytes")
