"""Spectroscopy.compute_absorption() as a user calls it, and the continuum and cross-section
slots by themselves."""
import os
import time

import numpy as np

from .common import HBM_PEAK_GBS, ROOT
from .profiled import profiled_traffic


def api_leg(engine, tables, atmos, v_lo, v_hi, dv, device_step_ms, repeats=9):
    """Wall clock of the call users make: Spectroscopy.compute_absorption() -- lines with the
    pedestal removed + MT-CKD continua of the same gases, results delivered as host arrays (the
    reference's contract) -- per output format, and the page-locked D2H rate it is bound by."""
    from pylbl_amd import MemoryDatabase, Spectroscopy, synthetic
    from pylbl_amd.engine import DeviceSpectra
    grid = np.arange(v_lo, v_hi, dv)
    formulas = [t.formula for t in tables]
    level = synthetic.Atmos(p=atmos.p[:1], t=atmos.t[:1],
                            vmr={f: atmos.vmr[f][:1] for f in formulas})
    try:
        spec = Spectroscopy(level, grid, MemoryDatabase(tables), device=engine.device)
        spec.compute_absorption(output_format="total")
    except FileNotFoundError:       # no MT-CKD coefficient tables anywhere: lines only
        spec = Spectroscopy(level, grid, MemoryDatabase(tables), continua_backend=None,
                            device=engine.device)
    # What the host link delivers into page-locked memory: one 40 MB-class copy, timed alone.
    block = DeviceSpectra(engine, 1, grid.size)
    target = engine.host_array((1, grid.size))
    block.to_host_into(target)
    engine.synchronize()
    start = time.perf_counter()
    for _ in range(8):          # queued back to back, one wait: the link's own rate
        block.to_host_into(target, asynchronous=True)
    engine.synchronize()
    link_gbs = grid.size*8*8/(time.perf_counter() - start)/1e9
    block.free()
    out = {"workload": f"Spectroscopy.compute_absorption(): 1 level, {'+'.join(formulas)}, "
                       f"{grid.size} points, lines (remove_pedestal as the reference defaults) + "
                       f"continua, host arrays returned", "formats": {},
           "d2h_pinned_gbs_measured": link_gbs}
    # Arrays that cross the link per format; in "all" a mechanism slot no back end fills is zeroed
    # on the host (spectroscopy._zero_in_background) and never travels.
    filled = 0
    for f in formulas:
        data = spec._molecule(f)
        filled += (data.gas is not None) + bool(data.gas_continua) + (data.cross_section is not None)
    for fmt, arrays, over_link in (("total", 1, 1), ("gas", len(formulas), len(formulas)),
                                   ("all", 3*len(formulas), filled)):
        # Warm-up the way the timed loop runs: every engine lane and pooled block used once, and
        # the previous result still alive while the next call computes -- two generations of
        # page-locked result arrays, or the second timed call pays for pinning one (4.6 / 7.5 /
        # 19 ms instead of 2.1 / 2.6 / 3.8: a fifth of which was in every mean before round 3's end).
        result = None
        for _ in range(4):
            result = spec.compute_absorption(output_format=fmt)
        # Every call timed by itself (it returns host arrays: nothing of it is left in flight); the
        # figure is the median, the mean and the extremes ride along -- one call in a dozen comes
        # out a millisecond late on some boxes, and a mean of five then says more about that call
        # than about the other four.
        times = []
        for _ in range(repeats):
            start = time.perf_counter()
            result = spec.compute_absorption(output_format=fmt)
            times.append(time.perf_counter() - start)
        seconds = float(np.median(times))
        del result
        delivered = arrays*grid.size*8
        linked = over_link*grid.size*8
        out["formats"][fmt] = {
            "ms_per_call": seconds*1e3, "spectra_per_s": 1./seconds,
            "ms_per_call_mean": float(np.mean(times))*1e3, "ms_per_call_min": min(times)*1e3,
            "ms_per_call_max": max(times)*1e3, "calls_timed": repeats,
            "bytes_delivered": delivered, "bytes_over_link": linked,
            "bytes_zero_filled_on_host": delivered - linked,
            "roofline": {"bound": "pcie_d2h", "achieved": linked/seconds/1e9,
                         "peak": link_gbs, "unit": "GB/s",
                         "frac": linked/seconds/1e9/link_gbs,
                         "note": "bytes that cross the host link / wall time of the call, against "
                                 "the rate of back-to-back copies into page-locked memory"}}
    # The "total" call against its parts run one after the other: the lines kernels it queues
    # (far-field series + pedestal, the Spectroscopy defaults) and one copy of the total over the
    # host link.  Below 1 since round 4: the heaviest gas delivers its runs of tiles while it
    # computes, so most of the copy hides behind the kernels.
    copy_ms = grid.size*8/link_gbs*1e-6
    out["device_resident_lines_step_ms"] = device_step_ms
    out["d2h_of_total_ms"] = copy_ms
    out["total_vs_lines_plus_copy"] = out["formats"]["total"]["ms_per_call"]/(device_step_ms + copy_ms)
    return out


def continuum_leg(engine, molecules, atmos, mine, v_lo, v_hi, dv, steps, with_cpu):
    """Times the continuum kernels (pylbl_amd/csrc/continuum.h) for the gases of the workload
    that have an MT-CKD continuum.  Needs the coefficient tables ($PYLBL_MT_CKD, an installed
    pyLBL, or the fixture under tests/golden); returns None without them."""
    from pylbl_amd import mt_ckd, mt_ckd_data
    from pylbl_amd.engine import DeviceSpectra
    try:
        path = mt_ckd_data.default_path()
    except FileNotFoundError:
        path = os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz")
        if not os.path.isfile(path):
            return None
    owners = []
    for formula in molecules:
        owners += ["H2OForeign", "H2OSelf"] if formula == "H2O" else \
            [formula] if formula in mt_ckd.CONTINUA else []
    if not owners:
        return None
    grid = np.arange(v_lo, v_hi, dv)
    continua = [mt_ckd.CONTINUA[owner](path=path, engine=engine) for owner in owners]
    t, p = atmos.t[mine], atmos.p[mine]
    vmr = {formula: values[mine] for formula, values in atmos.vmr.items()}
    block = DeviceSpectra(engine, t.size, grid.size)

    def step_one_by_one():
        for i, continuum in enumerate(continua):
            continuum.spectra_levels(t, p, vmr, grid, out=block, accumulate=i > 0,
                                     asynchronous=True)

    def step():
        # every continuum in ONE pass over the grid (lbl_continuum_compute_many): what
        # Spectroscopy queues for the continua of a gas / of all gases
        mt_ckd.spectra_levels_many(continua, t, p, vmr, grid, block, asynchronous=True)

    def timed(run):
        for _ in range(2):
            run()
        engine.synchronize()
        engine.set_option("timing", 1)
        engine.timing(reset=True)
        start = time.perf_counter()
        for _ in range(steps):
            run()
        engine.synchronize()
        seconds = time.perf_counter() - start
        ms, counts = engine.timing(reset=True)
        engine.set_option("timing", 0)
        return seconds, ms, counts
    separate_s, separate_ms, _ = timed(step_one_by_one)
    elapsed, kernel_ms, launches = timed(step)
    block.free()
    cpu = None
    if with_cpu:
        # The numpy restatement of the reference's path (oracle/, "port"; the reference itself
        # needs netCDF4/xarray) for the first level, one thread.
        from oracle import mt_ckd_oracle
        tables = mt_ckd_oracle.load_tables(path)
        first = {formula: values[0] for formula, values in vmr.items()}
        checkers = [mt_ckd_oracle.Continuum(owner, tables) for owner in owners]
        begin = time.perf_counter()
        for checker in checkers:
            checker.spectra(t[0], p[0], first, grid)
        seconds = time.perf_counter() - begin
        cpu = {"value": len(owners)*grid.size/seconds, "unit": "continuum x grid points/s",
               "cores": 1, "kind": "port",
               "sample": f"{'+'.join(owners)} for one level on the same {grid.size} points "
                         f"({seconds:.2f} s)"}
    # One pass: the wavenumber in, the extinction out (what the reference's numpy.interp reads and
    # writes per continuum, utils.py:171-173) -- 16 algorithmic bytes per point and level, once.
    bytes_per_step = grid.size*t.size*16
    interp_seconds = kernel_ms[5]*1e-3/steps
    achieved = bytes_per_step/interp_seconds/1e9
    adding = len(continua) - 1
    return {
        "workload": f"MT-CKD continua {'+'.join(owners)} summed into one [levels, points] block "
                    f"in HBM in ONE pass over the grid, {t.size} level(s), {grid.size} points",
        "ms_per_step": elapsed/steps*1e3,
        "spectra_per_s": t.size*steps/elapsed,
        "value": len(owners)*grid.size*t.size*steps/elapsed, "unit": "continuum x grid points/s",
        "cpu_baseline": cpu,
        "kernel_ms_per_step": {"band_spectra": kernel_ms[4]/steps, "interpolate": kernel_ms[5]/steps},
        "one_launch_per_continuum": {
            "ms_per_step": separate_s/steps*1e3,
            "kernel_ms_per_step": {"band_spectra": separate_ms[4]/steps,
                                   "interpolate": separate_ms[5]/steps},
            "algorithmic_bytes_per_step": grid.size*t.size*(16*len(continua) + 8*adding),
            "note": "the same sum as round 4 formed it: the first continuum writes the block, "
                    "every other one is a read-modify-write pass (bit-identical results)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBS, "traffic": None,
                     "kernel": "lbl::group_interp_kernel",
                     "avg_launch_ms": kernel_ms[5]/max(launches[5], 1),
                     "note": "16 algorithmic bytes per point and level (wavenumber in, extinction "
                             "out), once for all continua of the group; the kernel itself forms "
                             "the wavenumber of an arithmetic grid (numpy.arange) in registers and "
                             "moves 8; HIP events on the engine's stream"},
    }


def cross_section_leg(engine, atmos, mine, v_lo, v_hi, dv, steps, with_cpu):
    """Times the cross-section kernels (pylbl_amd/csrc/xsec.h) for one halocarbon-like
    molecule with synthetic coefficient bands (the reference's files are a download) on the
    workload's grid and levels."""
    from pylbl_amd import synthetic
    from pylbl_amd.engine import DeviceSpectra
    from pylbl_amd.mt_ckd import resident_grid
    grid = np.arange(v_lo, v_hi, dv)
    span = v_hi - v_lo
    ranges = ((v_lo + 0.12*span, v_lo + 0.18*span), (v_lo + 0.21*span, v_lo + 0.25*span))
    bands = synthetic.cross_section_bands(seed=11, ranges=ranges, spacing=0.03)
    handle = engine.load_xsec(bands)
    grid_handle = resident_grid(engine, grid)
    t, p = atmos.t[mine], atmos.p[mine]
    vmr = np.full(t.size, 2.3e-10)
    block = DeviceSpectra(engine, t.size, grid.size)

    def step():
        engine.xsec_compute(handle, grid_handle, grid.size, t, p, vmr=vmr, out=block,
                            asynchronous=True)
    for _ in range(2):
        step()
    engine.synchronize()
    engine.set_option("timing", 1)
    engine.timing(reset=True)
    start = time.perf_counter()
    for _ in range(steps):
        step()
    engine.synchronize()
    elapsed = time.perf_counter() - start
    kernel_ms, launches = engine.timing(reset=True)
    engine.set_option("timing", 0)
    block.free()
    engine.free_xsec(handle)
    cpu = None
    if with_cpu:
        # The reference's fit restated + the scipy interp1d it calls (oracle/, "port").
        from oracle import xsec_oracle
        begin = time.perf_counter()
        xsec_oracle.absorption_coefficient(bands, grid, t[0], p[0])
        seconds = time.perf_counter() - begin
        cpu = {"value": grid.size/seconds, "unit": "grid points/s", "cores": 1, "kind": "port",
               "sample": f"one level on the same {grid.size} points ({seconds:.2f} s)"}
    achieved = 16.*grid.size*t.size/(kernel_ms[7]*1e-3/steps)/1e9
    return {
        "workload": f"ARTS-crossfit-like molecule, {len(bands)} bands of "
                    f"{'+'.join(str(f.size) for f, _ in bands)} frequencies (synthetic), "
                    f"{t.size} level(s), {grid.size} points, n k written to HBM",
        "ms_per_step": elapsed/steps*1e3,
        "spectra_per_s": t.size*steps/elapsed,
        "value": grid.size*t.size*steps/elapsed, "unit": "grid points/s",
        "cpu_baseline": cpu,
        "kernel_ms_per_step": {"fit": kernel_ms[6]/steps, "interpolate": kernel_ms[7]/steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBS, "traffic": None,
                     "kernel": "lbl::xsec_interp_kernel",
                     "avg_launch_ms": kernel_ms[7]/max(launches[7], 1),
                     "note": "16 algorithmic bytes per point and level (wavenumber in, n k out); "
                             "HIP events on the engine's stream"},
    }


def api_and_slot_legs(job, line, leg):
    """Spectroscopy.compute_absorption() as a user calls it, and the continuum and
    cross-section slots by themselves."""
    args, engine, tables, molecules, atmos = job.args, job.engine, job.tables, job.molecules, job.atmos
    v_lo, v_hi, dv, workload, levels_local = job.v_lo, job.v_hi, job.dv, job.workload, job.levels_local
    if leg("api"):
        # What the call queues on the device: Spectroscopy sums distant lines through the
        # far-field series by default and removes the pedestal (continua on).
        device = line.get("farfield_option", {}).get("remove_pedestal") or \
            line.get("pedestal_option", line)
        line["api_call"] = api_leg(engine, tables, atmos, v_lo, v_hi, dv,
                                   device["ms_per_step"])
    if leg("continuum"):
        mine = slice(0, levels_local)
        extra = continuum_leg(engine, molecules, atmos, mine, v_lo, v_hi, dv, args.steps,
                              not args.no_cpu_baseline)
        if extra is not None:
            line["continuum_slot"] = extra
            traffic, source = profiled_traffic(workload, "group_interp_kernel")
            if traffic is not None:
                extra["roofline"]["traffic"] = traffic
                extra["roofline"]["traffic_source"] = f"profiles/{source}"
        extra = cross_section_leg(engine, atmos, mine, v_lo, v_hi, dv, args.steps,
                                  not args.no_cpu_baseline)
        line["cross_section_slot"] = extra
        traffic, source = profiled_traffic(workload, "xsec_interp_kernel")
        if traffic is not None:
            extra["roofline"]["traffic"] = traffic
            extra["roofline"]["traffic_source"] = f"profiles/{source}"

