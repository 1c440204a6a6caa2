"""Constants and small helpers shared by bench.py's legs."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz
SIMDS = 1024                # 256 CUs x 4
BOOST_CLOCK_GHZ = 2.4
BYTES_PER_EVAL = 24         # load dwno[i], load k[i], store k[i] (voigt.c:76,188)
FLOPS_PER_EVAL = 7          # 5 common + 2 for the far-wing branch (>99 % of evals)

EIGHT = ["H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2"]
CONFIGS = {
    # name: (molecules, v_lo, v_hi, dv, levels of the BASELINE config)
    "0": (["CO2"], 500., 800., 0.1, 1),
    "1": (["H2O", "CO2"], 1., 5000., 0.01, 1),
    "target": (["H2O", "CO2"], 1., 5000., 0.001, 1),
    "2": (EIGHT, 1., 5000., 0.001, 1),
    "3": (["H2O", "CO2", "O3"], 1., 3000., 0.001, 64),
    "4": (EIGHT, 1., 5000., 0.0005, 256),
}


def atmosphere_for(levels_total, profile):
    """profile "surface": every level is the reference's surface fixture level
    (tests/conftest.py:61-77), so each GPU of a weak-scaling run gets exactly the same work;
    "standard": level 0 is that level, the rest a standard atmosphere (lower pressures are
    10-20 % slower per level: more evaluations fall in the inner Voigt regions)."""
    from pylbl_amd import synthetic
    surface = synthetic.surface_level()
    if levels_total == 1:
        return surface
    if profile == "surface":
        return synthetic.Atmos(p=np.repeat(surface.p, levels_total),
                               t=np.repeat(surface.t, levels_total),
                               vmr={k: np.repeat(v, levels_total) for k, v in surface.vmr.items()})
    standard = synthetic.standard_atmosphere(levels_total)
    t = standard.t.copy()
    p = standard.p.copy()
    vmr = {k: v.copy() for k, v in standard.vmr.items()}
    t[0], p[0] = surface.t[0], surface.p[0]
    for k in vmr:
        vmr[k][0] = surface.vmr[k][0]
    return synthetic.Atmos(p=p, t=t, vmr=vmr)


def cpu_quota():
    """CPUs' worth of time the process may use according to its cgroup (cpu.max, v2; cfs quota,
    v1), or None when unlimited / not readable: a pool may show every hardware thread of the host
    in the affinity mask and still be allotted a fraction of them."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as handle:
            quota, period = handle.read().split()[:2]
        return None if quota == "max" else float(quota)/float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as handle:
            quota = float(handle.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as handle:
            period = float(handle.read())
        return None if quota <= 0 else quota/period
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as handle:
            for line in handle:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"

