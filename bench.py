"""Benchmark of the molecular-lines hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch: for every level of this rank's shard and
every molecule of the workload, line-scalar prep + tile schedule + Voigt accumulate (+ the
pedestal pre-pass when --pedestal) with the line tables already resident in HBM and the
spectra left in HBM; with N > 1 the step ends with the RCCL gather of the level shards to
rank 0.  Metric: line x gridpoint Voigt evaluations per second (BASELINE.json), counted in
closed form as the reference's inner-loop iterations (sum of last-first+1, spectra.c:48-62).

Workload at N = 1 (default): the configuration BASELINE.json quotes its target on -- 1
level, H2O + CO2, grid 1-5000 cm-1 at 0.001 cm-1 (5 M points), synthetic HITRAN-like line
tables (no HITRAN database exists offline).  --config selects the other BASELINE configs.
Weak scaling: every rank gets --levels-per-gpu levels (default 1) of a standard atmosphere.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6
BYTES_PER_EVAL = 24         # load dwno[i], load k[i], store k[i] (voigt.c:76,188)
FLOPS_PER_EVAL = 7          # 5 common + 2 for the far-wing branch (>99 % of evals)

CONFIGS = {
    # name: (molecules, v_lo, v_hi, dv, levels_total (None = per-gpu levels))
    "0": (["CO2"], 500., 800., 0.1, 1),
    "1": (["H2O", "CO2"], 1., 5000., 0.01, 1),
    "target": (["H2O", "CO2"], 1., 5000., 0.001, 1),
    "2": (["H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2"], 1., 5000., 0.001, 1),
    "3": (["H2O", "CO2", "O3"], 1., 3000., 0.001, 64),
    "4": (["H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2"], 1., 5000., 0.0005, 256),
}


def parse():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--config", default="target", choices=sorted(CONFIGS))
    parser.add_argument("--levels-per-gpu", type=int, default=1)
    parser.add_argument("--profile", default="surface", choices=["surface", "standard"],
                        help="atmosphere of multi-level runs (see atmosphere_for)")
    parser.add_argument("--pedestal", action="store_true",
                        help="remove_pedestal=True (the default through compute_absorption)")
    parser.add_argument("--line-scale", type=float, default=1.,
                        help="multiplies the HITRAN-like line counts")
    parser.add_argument("--points-per-lane", type=int, default=0)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-extras", action="store_true",
                        help="only the timed steps: no CPU baseline, no far-field extra pass "
                             "(what scripts/profile_bench.sh runs under rocprofv3)")
    parser.add_argument("--extras", default="all", choices=["all", "none", "farfield", "continuum"],
                        help="which untimed extra legs run after the timed steps "
                             "(continuum = the continuum and cross-section slots)")
    parser.add_argument("--farfield", action="store_true",
                        help="engine option farfield=1: distant lines through their power "
                             "series (an algorithmic shortcut; not the default)")
    parser.add_argument("--host-output", action="store_true",
                        help="copy every spectrum back to host memory inside the step "
                             "(PCIe-inclusive rate; never the headline value)")
    parser.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                        help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo "
                             "only to rehearse the multi-rank flow on fewer GPUs than ranks)")
    parser.add_argument("--ablate", type=int, default=0,
                        help="diagnostics: 1 skips the general ranges, 2 the fast ranges "
                             "(results are wrong; the line is marked invalid)")
    parser.add_argument("--cpu-sample-cm", type=float, default=3000.,
                        help="width [cm-1] of the grid sample the CPU baseline is timed on")
    return parser.parse_args()


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


def cpu_baseline(tables, atmos, v0, n_per_v, sample_cm, remove_pedestal):
    """Times the CPU path on a bounded sample of the same workload (level 0, the first
    `sample_cm` cm-1 of the grid, lines within reach of it): the reference's own compiled C
    reading SQLite when oracle/_ref is present ("reference"), else our C restatement
    ("port").  One thread, like the reference."""
    import tempfile
    from oracle import oracle
    from pylbl_amd.database import write_database
    vn = v0 + int(sample_cm)
    sample = [t.subset(t.nu <= vn + 26.) for t in tables]
    evals = 0
    kind = "reference" if oracle.have_reference() else "port"
    seconds = 0.
    port_seconds = 0.
    with tempfile.TemporaryDirectory() as tmp:
        db = None
        if kind == "reference":
            db = write_database(os.path.join(tmp, "sample.db"), sample)
        for t in sample:
            args = (atmos.t[0], atmos.p[0], atmos.vmr[t.formula][0], v0, vn, n_per_v)
            start = time.perf_counter()
            _, extras = oracle.absorption_port(t, *args, remove_pedestal=remove_pedestal)
            port_seconds += time.perf_counter() - start
            evals += extras["evals"]
            if kind == "reference":
                start = time.perf_counter()
                rc, _ = oracle.absorption_reference(db, t.formula, *args,
                                                    remove_pedestal=remove_pedestal)
                seconds += time.perf_counter() - start
                if rc != 0:
                    raise RuntimeError("reference absorption() failed")
    if kind == "port":
        seconds = port_seconds
    return {
        "value": evals/seconds, "unit": "evals/s", "cores": 1, "kind": kind,
        "sample": f"level 0, {'+'.join(t.formula for t in sample)}, grid {v0}-{vn} cm-1 at "
                  f"{1./n_per_v:g} cm-1, {sum(t.num_lines for t in sample)} lines, "
                  f"{evals:.4g} evals in {seconds:.2f} s (SQLite read per call included, as "
                  f"the reference does); C restatement on arrays: {evals/port_seconds:.4g} "
                  f"evals/s",
        "cpu": cpu_model(), "host_cores": os.cpu_count(),
    }


def profiled_traffic(workload, kernel="accumulate_kernel"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 counter summary
    (profiles/*_summary.json, made by scripts/profile_bench.sh + summarize_profile.py: separate
    FETCH_SIZE / WRITE_SIZE passes, KiB units, reads doubled per the gfx950 correction) -- only
    if that profile ran this same workload; the counters cannot be read from inside bench.py."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")), reverse=True):
        try:
            with open(path) as handle:
                summary = json.load(handle)
            if summary["bench_line"]["config"]["workload"] != workload:
                continue
            for name, counters in summary["counters"].items():
                if kernel in name and "hbm_bytes_per_launch" in counters:
                    return counters["hbm_bytes_per_launch"], os.path.basename(path)
        except (OSError, KeyError, TypeError, ValueError):
            continue
    return None, None


def _cpu_chunk(job):
    """Worker of cpu_baseline_parallel: the C restatement on one sub-grid of the sample."""
    from oracle import oracle
    table, t, p, x, v0, vn, n_per_v, remove_pedestal = job
    _, extras = oracle.absorption_port(table, t, p, x, v0, vn, n_per_v,
                                       remove_pedestal=remove_pedestal)
    return extras["evals"]


def cpu_baseline_parallel(tables, atmos, v0, n_per_v, sample_cm, workers):
    """What a user could do with multiprocessing around the reference's Gas: independent
    (molecule, sub-grid) units of the same sample farmed out over `workers` processes
    (our C restatement on arrays; pedestal off, the units would not be independent with it)."""
    import multiprocessing
    vn = v0 + int(sample_cm)
    edges = np.linspace(v0, vn, workers + 1).astype(int)
    jobs = []
    for t in tables:
        for lo, hi in zip(edges[:-1], edges[1:]):
            if hi > lo:
                near = t.subset((t.nu >= lo - 26.) & (t.nu <= hi + 26.))
                jobs.append((near, atmos.t[0], atmos.p[0], atmos.vmr[t.formula][0], int(lo),
                             int(hi), n_per_v, False))
    context = multiprocessing.get_context("spawn")
    with context.Pool(workers) as pool:
        pool.map(_cpu_chunk, jobs[:workers])           # start-up and library load, untimed
        start = time.perf_counter()
        evals = sum(pool.map(_cpu_chunk, jobs, chunksize=1))
        seconds = time.perf_counter() - start
    return {"value": evals/seconds, "unit": "evals/s", "cores": workers, "kind": "port",
            "sample": f"same sample as cpu_baseline cut into {len(jobs)} (molecule, sub-grid) "
                      f"units, {evals:.4g} evals in {seconds:.2f} s"}


def continuum_leg(engine, molecules, atmos, mine, v_lo, v_hi, dv, steps, with_cpu):
    """Times the continuum kernels (pylbl_amd/csrc/continuum.h) for the gases of the workload
    that have an MT-CKD continuum.  Needs the coefficient tables ($PYLBL_MT_CKD, an installed
    pyLBL, or the fixture under tests/golden); returns None without them."""
    from pylbl_amd import mt_ckd, mt_ckd_data
    from pylbl_amd.engine import DeviceSpectra
    try:
        path = mt_ckd_data.default_path()
    except FileNotFoundError:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden",
                            "mt_ckd_bands.npz")
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

    def step():
        for i, continuum in enumerate(continua):
            continuum.spectra_levels(t, p, vmr, grid, out=block, accumulate=i > 0,
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
    # Algorithmic bytes per launch: wavenumber in + extinction out per point and level, plus
    # the extinction read back by the launches that add to it.
    # cpu_baseline of this leg: the numpy restatement of the reference's path (oracle/, "port";
    # the reference itself needs netCDF4/xarray) for the first level, one thread.
    cpu = None
    if with_cpu:
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
    adding = len(continua) - 1
    bytes_per_step = grid.size*t.size*(16*len(continua) + 8*adding)
    interp_seconds = kernel_ms[5]*1e-3/steps
    achieved = bytes_per_step/interp_seconds/1e9
    return {
        "workload": f"MT-CKD continua {'+'.join(owners)} summed into one [levels, points] block "
                    f"in HBM, {t.size} level(s), {grid.size} points",
        "ms_per_step": elapsed/steps*1e3,
        "spectra_per_s": t.size*steps/elapsed,
        "value": len(owners)*grid.size*t.size*steps/elapsed, "unit": "continuum x grid points/s",
        "cpu_baseline": cpu,
        "kernel_ms_per_step": {"band_spectra": kernel_ms[4]/steps, "interpolate": kernel_ms[5]/steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved/HBM_PEAK_GBS, "traffic": None,
                     "kernel": "lbl::continuum_interp_kernel",
                     "avg_launch_ms": kernel_ms[5]/max(launches[5], 1),
                     "note": "16 algorithmic bytes per point and level (wavenumber in, extinction "
                             "out; +8 when adding into the block); HIP events on the engine's stream"},
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as handle:
            for line in handle:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    if args.no_extras:
        args.extras = "none"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for N > 1")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback).")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")

    from pylbl_amd import synthetic
    from pylbl_amd.engine import Engine

    molecules, v_lo, v_hi, dv, config_levels = CONFIGS[args.config]
    grid_v0, grid_vn, n_per_v = synthetic.grid_arguments(np.asarray([v_lo, v_lo + dv, v_hi - dv]))
    n = (grid_vn - grid_v0)*n_per_v
    levels_local = args.levels_per_gpu
    levels_total = levels_local*world
    atmos = atmosphere_for(levels_total, args.profile)
    mine = slice(rank*levels_local, (rank + 1)*levels_local)

    tables = [synthetic.line_table(f, v_lo, v_hi, scale=args.line_scale) for f in molecules]
    engine = Engine(device_index)
    if args.points_per_lane:
        engine.set_option("points_per_lane", args.points_per_lane)
    if args.farfield:
        engine.set_option("farfield", 1)
    if args.ablate:
        engine.set_option("ablate", args.ablate)
    handles = [engine.load(t) for t in tables]

    # Spectra stay in HBM: [molecule, level, n] per rank (torch only owns the memory).  Two
    # buffers alternate so that the gather of one step (N > 1) runs beside the next step.
    spectra = [torch.empty((len(molecules), levels_local, n), dtype=torch.float64, device="cuda")
               for _ in range(2 if world > 1 else 1)]
    on_host = world > 1 and args.backend == "gloo"
    gathered = [None, None]
    if world > 1 and rank == 0:
        gathered = [[torch.empty_like(spectra[0], device="cpu" if on_host else "cuda")
                     for _ in range(world)] for _ in range(2)]
    pending = [None, None]
    counter = [0]
    use_all_gather = [False]

    class Slot(object):
        def __init__(self, tensor):
            self.pointer = tensor.data_ptr()
            self.shape = tuple(tensor.shape)

    # --host-output: page-locked host memory, what Gas/Spectroscopy hand their callers.
    host_spectra = engine.host_array((len(molecules), levels_local, n)) if args.host_output \
        else None

    def settle(which):
        """Waits for the gather that last used buffer `which`."""
        if pending[which] is not None:
            pending[which].wait()
            if not on_host:
                torch.cuda.current_stream().synchronize()
            pending[which] = None

    def step(count_evals=False):
        which = counter[0] % len(spectra)
        counter[0] += 1
        settle(which)
        total = 0
        for m, handle in enumerate(handles):
            formula = molecules[m]
            result = engine.compute(handle, atmos.t[mine], atmos.p[mine], atmos.vmr[formula][mine],
                                    grid_v0, grid_vn, n_per_v, remove_pedestal=args.pedestal,
                                    out=host_spectra[m] if args.host_output
                                    else Slot(spectra[which][m]),
                                    asynchronous=not args.host_output,
                                    want_evals=count_evals)
            if count_evals:
                total += result[1]
        if world > 1:
            # The engine runs on its own streams: finish the spectra, then start the gather;
            # it completes while the next step computes into the other buffer.
            engine.synchronize()
            source = spectra[which].cpu() if on_host else spectra[which]
            if not use_all_gather[0]:
                try:
                    pending[which] = dist.gather(source, gathered[which], dst=0, async_op=True)
                except (RuntimeError, NotImplementedError, ValueError):
                    use_all_gather[0] = True    # backend without gather: every rank collects
            if use_all_gather[0]:
                if gathered[which] is None:
                    gathered[which] = [torch.empty_like(source) for _ in range(world)]
                pending[which] = dist.all_gather(gathered[which], source, async_op=True)
        return total

    def fence():
        settle(0)
        settle(1)
        engine.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    evals_per_step_local = step(count_evals=True)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    fence()
    engine.set_option("timing", 1)
    engine.timing(reset=True)
    fence()
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - start
    kernel_ms, launches = engine.timing(reset=True)
    engine.set_option("timing", 0)

    # Not part of `value`: the same steps with the optional far-field series switched on.
    farfield_extra = None
    if world == 1 and not args.farfield and not args.ablate and not args.host_output and \
            args.extras in ("all", "farfield"):
        engine.set_option("farfield", 1)
        for _ in range(2):
            step()
        fence()
        start = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        far_elapsed = time.perf_counter() - start
        engine.set_option("farfield", 0)
        farfield_extra = {
            "value": evals_per_step_local*args.steps/far_elapsed, "unit": "evals/s",
            "ms_per_step": far_elapsed/args.steps*1e3,
            "note": "engine option farfield=1 (pylbl_amd/csrc/farfield.h): lines at least 4 tile "
                    "half-widths away are summed as one power series per tile (truncation "
                    "<= ~1.5e-11 relative); same closed-form eval count; opt-in, parity-tested "
                    "at the same 1e-6 bar",
        }

    # Not part of `value` either: mechanism slot 1 (MT-CKD continua) for the same gases, levels
    # and grid, written into the same kind of HBM block.
    continuum_extra = None
    if world == 1 and not args.ablate and not args.host_output and \
            args.extras in ("all", "continuum"):
        continuum_extra = continuum_leg(engine, molecules, atmos, mine, v_lo, v_hi, dv, args.steps,
                                        not args.no_cpu_baseline)

    cross_section_extra = None
    if world == 1 and not args.ablate and not args.host_output and \
            args.extras in ("all", "continuum"):
        cross_section_extra = cross_section_leg(engine, atmos, mine, v_lo, v_hi, dv, args.steps,
                                                not args.no_cpu_baseline)

    stats = torch.tensor([elapsed, float(evals_per_step_local)], dtype=torch.float64,
                         device="cpu" if on_host else "cuda")
    if world > 1:
        worst = stats.clone()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        total = stats.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        elapsed = float(worst[0])
        evals_per_step = float(total[1])
    else:
        evals_per_step = float(evals_per_step_local)

    if rank == 0:
        ms_per_step = elapsed/args.steps*1e3
        value = evals_per_step*args.steps/elapsed
        accumulate_ms = kernel_ms[2]/max(launches[2], 1)
        evals_per_launch = evals_per_step_local/max(launches[2]/args.steps, 1)
        achieved = evals_per_launch*BYTES_PER_EVAL/(accumulate_ms*1e-3)/1e9
        line = {
            "metric": "line\u00d7gridpoint Voigt evals/sec (whole job; per GPU: evals_per_s_per_gpu; spectra/sec: spectra_per_s)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"BASELINE config '{args.config}': {levels_local} level(s) per GPU, "
                            f"{'+'.join(molecules)}, grid {v_lo:g}-{v_hi:g} cm-1 at {dv:g} cm-1 "
                            f"({n} points), cut_off 25, remove_pedestal={args.pedestal}"
                            + (", far-field series on" if args.farfield else ""),
                "lines": {t.formula: t.num_lines for t in tables},
                "levels_total": levels_total, "atmosphere": args.profile, "parallelism": f"levels sharded over {world} GPU(s)"
                + (f", {args.backend} gather to rank 0 every step (overlapping the next step)"
                   if world > 1 else ""),
            },
            "evals_per_step": evals_per_step,
            "evals_per_s_per_gpu": value/world,
            "spectra_per_s": levels_total*args.steps/elapsed,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved/HBM_PEAK_GBS, "traffic": None,
                "kernel": "lbl::accumulate_kernel", "avg_launch_ms": accumulate_ms,
                "launches_timed": launches[2],
                "note": "achieved = 24 algorithmic bytes per eval (the reference's load v[i], "
                        "load+store k[i]) x evals per launch / launch time; the kernel keeps "
                        "partial sums in registers, so real HBM traffic is ~8 B per grid point "
                        "and the binding resource is the fp64 vector ALU (see roofline_fp64)",
            },
            "roofline_fp64": {
                "bound": "fp64 vector ALU", "unit": "TFLOP/s", "peak": FP64_VECTOR_PEAK_TFLOPS,
                "achieved": evals_per_launch*FLOPS_PER_EVAL/(accumulate_ms*1e-3)/1e12,
                "frac": evals_per_launch*FLOPS_PER_EVAL/(accumulate_ms*1e-3)/1e12 /
                        FP64_VECTOR_PEAK_TFLOPS,
                "flops_per_eval": FLOPS_PER_EVAL,
            },
            "kernel_ms_per_step": {
                "prepare": kernel_ms[0]/args.steps, "schedule": kernel_ms[1]/args.steps,
                "accumulate": kernel_ms[2]/args.steps, "pedestal": kernel_ms[3]/args.steps},
        }
        if args.pedestal:
            line["roofline"]["note"] += ("; remove_pedestal=True: calls alternate between engine "
                                         "lanes and their accumulate kernels overlap in time, so "
                                         "avg_launch_ms is not the duration of a kernel running "
                                         "alone (see the plain run for that)")
        if farfield_extra is not None:
            line["farfield_option"] = farfield_extra
        if cross_section_extra is not None:
            line["cross_section_slot"] = cross_section_extra
            traffic, source = profiled_traffic(line["config"]["workload"], "xsec_interp_kernel")
            if traffic is not None:
                cross_section_extra["roofline"]["traffic"] = traffic
                cross_section_extra["roofline"]["traffic_source"] = f"profiles/{source}"
        if continuum_extra is not None:
            line["continuum_slot"] = continuum_extra
            traffic, source = profiled_traffic(line["config"]["workload"], "continuum_interp_kernel")
            if traffic is not None:
                continuum_extra["roofline"]["traffic"] = traffic
                continuum_extra["roofline"]["traffic_source"] = f"profiles/{source}"
        if args.host_output:
            line["INVALID"] = "host output: PCIe copies inside the step (reported for DESIGN.md)"
        traffic, source = profiled_traffic(line["config"]["workload"])
        if traffic is not None:
            line["roofline"]["traffic"] = traffic
            line["roofline"]["traffic_source"] = f"profiles/{source}"
        if args.ablate:
            line["INVALID"] = f"ablation {args.ablate}: part of the work was skipped"
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            line["cpu_baseline"] = cpu_baseline(tables, atmos, grid_v0, n_per_v,
                                                args.cpu_sample_cm, args.pedestal)
            workers = min(16, os.cpu_count() or 1)
            if workers > 1:
                line["cpu_baseline_parallel"] = cpu_baseline_parallel(
                    tables, atmos, grid_v0, n_per_v, args.cpu_sample_cm, workers)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()
