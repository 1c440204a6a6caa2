"""Benchmark of the molecular-lines hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nproc-per-node N ...
     bench.py --gpus N ... -- or bare: without WORLD_SIZE in the environment bench.py starts its
     N ranks itself as child processes, before anything touches the GPU: launch_ranks)

A step = one pass of the hot path over one batch: for every (level, molecule) unit of this
rank (pylbl_amd.distributed.partition), line-scalar prep + tile schedule + Voigt accumulate
(+ the pedestal pre-pass with --pedestal), line tables already resident in HBM, spectra left
in HBM; with N > 1 the step ends by starting the collection of the spectra on rank 0 (one
grouped send/recv over RCCL, pylbl_amd.distributed.ShardedLines), which runs beside the next
step.  Metric: line x gridpoint Voigt evaluations per second (BASELINE.json), counted in closed
form as the reference's inner-loop iterations (sum of last-first+1, spectra.c:48-62).

Workload at N = 1 (default): the configuration BASELINE.json quotes its target on -- 1 level,
H2O + CO2, grid 1-5000 cm-1 at 0.001 cm-1 (5 M points), synthetic HITRAN-like line tables (no
HITRAN database exists offline).  --config selects the other BASELINE configs.  Weak scaling:
every rank gets --levels-per-gpu levels (default 1).

Besides the contract's keys the line carries (N = 1): `roofline` (the resource that binds the
dominant kernel: the fp64 vector ALU), `roofline_hbm_algorithmic` (SURVEY 8d's 24 B/eval figure,
for the record), `cpu_baseline` (+ `_parallel`), and untimed legs that measure what users run:
`sustained`, `single_lane_option`, `pedestal_option`, `standard_atmosphere_option`,
`banded_table_option`, `dense_table_option`,
`small_grid_options`, `farfield_option`, `api_call`, `continuum_slot`, `cross_section_slot`,
the other BASELINE configs at one GPU's size -- `config2_option` (8 molecules, 5 M points),
`config3_share_option` (one rank's 8 of the 64 standard-atmosphere levels, through
ShardedLines.for_engine) and `config4_share_option` (4 of a rank's 32 levels x 8 molecules at
10 M points, summed over the gases on the device), each with a roofline fraction from launches
timed alone -- and `ingest`: what it costs to get a molecule's line table from the reference's
SQLite file into HBM by each route (load time reported separately), with
`cpu_baseline` split into the reference's per-call read and its Voigt loop.
`environment` echoes every PYLBL_AMD_* variable that was in effect.
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
    parser.add_argument("--output", default="gas", choices=["gas", "total"],
                        help="what a step leaves / collects: one spectrum per molecule, or "
                             "n k summed over the molecules on the device")
    parser.add_argument("--line-scale", type=float, default=1.,
                        help="multiplies the HITRAN-like line counts")
    parser.add_argument("--banded", action="store_true",
                        help="banded line tables (same counts) instead of uniform ones")
    parser.add_argument("--points-per-lane", type=int, default=0)
    parser.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                        help="lbl_set_option before anything is computed (experiments)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-extras", action="store_true",
                        help="only the timed steps (what scripts/profile_bench.sh profiles)")
    parser.add_argument("--extras", default="all",
                        help="comma list of untimed legs: all, none, or any of sustained, "
                             "overlap, pedestal, atmosphere, banded, dense, small, farfield, api, continuum")
    parser.add_argument("--farfield", action="store_true",
                        help="engine option farfield=1: distant lines through their power "
                             "series (an algorithmic shortcut; never the headline value)")
    parser.add_argument("--host-output", action="store_true",
                        help="copy every spectrum back to host memory inside the step "
                             "(PCIe-inclusive rate; never the headline value)")
    parser.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                        help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo "
                             "only to rehearse the multi-rank flow on fewer GPUs than ranks)")
    parser.add_argument("--exchange-timeout", type=float, default=180.,
                        help="N > 1: seconds any collection of spectra, barrier or reduction may "
                             "take before the rank reports what it was waiting for and exits 3")
    parser.add_argument("--launch-timeout", type=float, default=1500.,
                        help="bare `bench.py --gpus N` (no launcher, N > 1): seconds the N child "
                             "ranks may take before they are ended and the run reported failed")
    parser.add_argument("--launch-grace", type=float, default=5.,
                        help="bare launch: seconds the other ranks get to report after one has failed")
    parser.add_argument("--ablate", type=int, default=0,
                        help="diagnostics: 1 skips the general ranges, 2 the fast ranges "
                             "(results are wrong; the line is marked invalid)")
    parser.add_argument("--cpu-sample-cm", type=float, default=0.,
                        help="width [cm-1] of the grid sample the CPU baseline is timed on "
                             "(0 = the whole grid, ~22 s for the default workload)")
    parser.add_argument("--cpu-workers", type=int, default=16,
                        help="processes of cpu_baseline_parallel (16 = one GPU's share of the "
                             "host on this pool; pass the host's core count to use them all)")
    parser.add_argument("--cpu-all-cores", type=int, default=-1,
                        help="processes of cpu_baseline_all_cores: -1 = every hardware thread this "
                             "process may run on (os.sched_getaffinity), 0 = skip the leg")
    parser.add_argument("--cpu-pool-timeout", type=float, default=45.,
                        help="seconds the all-cores pool may take to start and run before the leg "
                             "is given up (and says so)")
    parser.add_argument("--force-group", action="store_true",
                        help="initialise the torch.distributed process group also at N = 1 (with "
                             "--backend nccl: RCCL loads and a communicator is made on one GPU) and "
                             "run the step through the collection path of ShardedLines")
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


# ---------------------------------------------------------------------------------------------
# CPU baselines (the only place bench.py touches oracle/)
# ---------------------------------------------------------------------------------------------
def cpu_baseline(tables, atmos, v0, vn_full, n_per_v, sample_cm, remove_pedestal, db=None):
    """Times the CPU path on level 0 of the same workload (the whole grid unless --cpu-sample-cm
    bounds it): the reference's own compiled C reading SQLite when oracle/_ref is present
    ("reference"), else our C restatement ("port").  One thread, like the reference.

    The reference pays for its database on EVERY call (absorption.c:44-86: open, id / TIPS / mass
    look-ups, a full scan of the molecule's rows, five transcendental calls per row).  `split`
    separates that from the Voigt loop: the same call on the same file at ONE point per cm-1
    (the loop shrinks to 52 evaluations per line, ~0.1 % of the fine grid's) is the per-call
    cost that does not depend on the resolution; the rest is the loop (voigt.c:21-25,74-189)."""
    import tempfile
    from oracle import oracle
    from pylbl_amd.database import write_database
    vn = vn_full if sample_cm <= 0 else min(vn_full, v0 + int(sample_cm))
    sample = [t.subset(t.nu <= vn + 26.) for t in tables]
    evals = 0
    kind = "reference" if oracle.have_reference() else "port"
    seconds = 0.
    port_seconds = 0.
    read_seconds = 0.
    coarse_evals = 0
    with tempfile.TemporaryDirectory() as tmp:
        if kind == "reference" and (db is None or vn != vn_full):
            db = write_database(os.path.join(tmp, "sample.db"), sample)
        for t in sample:
            args = (atmos.t[0], atmos.p[0], atmos.vmr[t.formula][0], v0, vn, n_per_v)
            if kind == "reference":
                start = time.perf_counter()
                rc, _ = oracle.absorption_reference(db, t.formula, *args,
                                                    remove_pedestal=remove_pedestal)
                seconds += time.perf_counter() - start
                if rc != 0:
                    raise RuntimeError("reference absorption() failed")
                # The reference does not report its iteration count; it is closed form
                # (window lengths, spectra.c:48-62), checked against the restatement's own
                # counter by tests/test_host_logic.py.
                evals += closed_form_evals(t, atmos.p[0], v0, vn, n_per_v)
                start = time.perf_counter()
                rc, _ = oracle.absorption_reference(db, t.formula, *args[:5], 1,
                                                    remove_pedestal=remove_pedestal)
                read_seconds += time.perf_counter() - start
                coarse_evals += closed_form_evals(t, atmos.p[0], v0, vn, 1)
            else:
                start = time.perf_counter()
                _, extras = oracle.absorption_port(t, *args, remove_pedestal=remove_pedestal)
                port_seconds += time.perf_counter() - start
                evals += extras["evals"]
    if kind == "port":
        seconds = port_seconds
    whole = vn == vn_full
    out = {
        "value": evals/seconds, "unit": "evals/s", "cores": 1, "kind": kind,
        "sample": f"level 0, {'+'.join(t.formula for t in sample)}, "
                  f"{'the whole grid' if whole else 'grid sample'} {v0}-{vn} cm-1 at "
                  f"{1./n_per_v:g} cm-1, {sum(t.num_lines for t in sample)} lines, "
                  f"{evals:.4g} evals in {seconds:.2f} s"
                  + (" (SQLite read per call included, as the reference does)"
                     if kind == "reference" else ""),
        "cpu": cpu_model(), "host_cores": os.cpu_count(),
    }
    if kind == "reference":
        loop = max(seconds - read_seconds, 1e-9)
        out["split"] = {
            "total_s": seconds, "read_and_line_scalars_s": read_seconds, "voigt_loop_s": loop,
            "read_fraction": read_seconds/seconds,
            "voigt_loop_evals_per_s": (evals - coarse_evals)/loop,
            "read_s_per_molecule": read_seconds/max(len(sample), 1),
            "note": "read_and_line_scalars_s = the same reference call on the same file at 1 "
                    "point per cm-1 (database open, look-ups, every row stepped and prepared: "
                    "absorption.c:44-86, spectra.c:17-45; its 52 evaluations per line are "
                    f"{coarse_evals:.3g} of the {evals:.3g}); voigt_loop_s = total - that; the "
                    "reference pays the read on every (level, molecule) call, this engine once "
                    "per molecule (`ingest`)"}
    return out


def closed_form_evals(table, pressure, v0, vn, n_per_v, cut_off=25):
    """Sum over accepted lines of last-first+1 exactly as spectra.c:48-62 forms the window."""
    n = (vn - v0)*n_per_v
    accepted = np.ones(table.num_lines, bool)
    outside = (table.nu > vn + cut_off + 1) | (table.nu < v0 - (cut_off + 1))
    if outside.any():
        accepted[np.argmax(outside):] = False           # absorption.c:80-83
    centre = table.nu + (pressure*9.86923e-6)*table.delta_air
    fl = np.floor(centre)
    first = ((fl - cut_off - v0)*n_per_v).astype(np.int64)
    last = ((fl + cut_off + 1 - v0)*n_per_v).astype(np.int64)
    keep = accepted & (first < n)
    first = np.maximum(first, 0)
    last = np.minimum(last, n - 1)
    length = np.where(keep & (last >= first), last - first + 1, 0)
    return int(length.sum())


_WORKER_TABLES = {}


def _recipe_table(recipe):
    """A bench table rebuilt inside a worker from what make_tables() was given (deterministic
    seeds): nothing but a few numbers travels to the worker."""
    from pylbl_amd import synthetic
    table = _WORKER_TABLES.get(recipe)
    if table is None:
        formula, v_lo, v_hi, scale, banded, index = recipe
        table = synthetic.line_table(formula, v_lo, v_hi, scale=scale)
        if banded:
            table = synthetic.banded_line_table(formula, v_lo, v_hi, num_lines=table.num_lines,
                                                bands=8, seed=41 + index)
        _WORKER_TABLES[recipe] = table
    return table


def _warm_worker(recipes):
    """Pool initializer: the oracle library loaded and the tables built before anything is timed."""
    from oracle import oracle
    oracle.port_library()
    for recipe in recipes:
        _recipe_table(recipe)


def _cpu_chunk(job):
    """Worker of cpu_baseline_parallel: the C restatement on one sub-grid of the sample."""
    from oracle import oracle
    table, t, p, x, v0, vn, n_per_v, remove_pedestal = job
    if isinstance(table, tuple):
        table = _recipe_table(table)
        table = table.subset((table.nu >= v0 - 26.) & (table.nu <= vn + 26.))
    _, extras = oracle.absorption_port(table, t, p, x, v0, vn, n_per_v,
                                       remove_pedestal=remove_pedestal)
    return extras["evals"]


def cpu_baseline_parallel(tables, atmos, v0, vn_full, n_per_v, sample_cm, workers, timeout=None,
                          why=None, recipes=None):
    """What a user could do with multiprocessing around the reference's Gas: independent
    (molecule, sub-grid) units of the same grid farmed out over `workers` processes (our C
    restatement on arrays; pedestal off, the units would not be independent with it).
    timeout: seconds the pool may take (start-up included) before the leg is given up.
    recipes: {formula: what make_tables() built the table from}: the workers rebuild the
    (deterministic) tables themselves instead of receiving a slice with every unit -- with
    hundreds of workers the parent's pickling of the slices is otherwise what is timed."""
    import multiprocessing
    vn = vn_full if sample_cm <= 0 else min(vn_full, v0 + int(sample_cm))
    pieces = max(4*workers, 1)
    edges = np.unique(np.linspace(v0, vn, pieces + 1).astype(int))
    jobs = []
    weights = []
    for t in tables:
        for lo, hi in zip(edges[:-1], edges[1:]):
            inside = (t.nu >= lo - 26.) & (t.nu <= hi + 26.)
            near = recipes[t.formula] if recipes else t.subset(inside)
            jobs.append((near, atmos.t[0], atmos.p[0], atmos.vmr[t.formula][0], int(lo),
                         int(hi), n_per_v, False))
            weights.append(int(np.count_nonzero(inside))*(int(hi) - int(lo)))
    jobs = [jobs[i] for i in np.argsort(-np.asarray(weights), kind="stable")]
    context = multiprocessing.get_context("spawn")
    began = time.perf_counter()
    pool = context.Pool(workers, initializer=_warm_worker,
                        initargs=(tuple(recipes.values()) if recipes else (),))
    try:
        # start-up and library load, untimed
        left = None if timeout is None else timeout
        pool.map_async(_cpu_chunk, jobs[-workers:]).get(left)
        ready = time.perf_counter()
        start = time.perf_counter()
        left = None if timeout is None else max(timeout - (start - began), 1.)
        evals = sum(pool.map_async(_cpu_chunk, jobs, chunksize=1).get(left))
        seconds = time.perf_counter() - start
    except multiprocessing.TimeoutError:
        pool.terminate()
        pool.join()
        return {"value": None, "unit": "evals/s", "cores": workers, "kind": "port",
                "host_cores": os.cpu_count(),
                "sample": f"given up: {workers} processes not through after {timeout:g} s "
                          f"(--cpu-pool-timeout)"}
    pool.close()
    pool.join()
    return {"value": evals/seconds, "unit": "evals/s", "cores": workers, "kind": "port",
            "host_cores": os.cpu_count(),
            "usable_hardware_threads": len(os.sched_getaffinity(0)),
            "cgroup_cpu_quota_cores": cpu_quota(),
            "pool_start_s": ready - began,
            "sample": f"the grid {v0}-{vn} cm-1 cut into {len(jobs)} (molecule, sub-grid) "
                      f"units over {workers} processes ({why or '--cpu-workers'}; the host has "
                      f"{os.cpu_count()} hardware threads, "
                      f"{len(os.sched_getaffinity(0))} in this process's affinity mask, cgroup CPU "
                      f"quota {cpu_quota() or 'none'} cores), {evals:.4g} evals "
                      f"in {seconds:.2f} s (+ {ready - began:.1f} s to start the pool, untimed)"}


# ---------------------------------------------------------------------------------------------
# profiles/ look-ups (the PMC counters cannot be read from inside bench.py)
# ---------------------------------------------------------------------------------------------
def profiled_traffic(workload, kernel="accumulate_kernel"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 counter summary
    (profiles/*_summary.json, made by scripts/profile_bench.sh + summarize_profile.py: separate
    FETCH_SIZE / WRITE_SIZE passes, KiB units, reads doubled per the gfx950 correction) -- only
    if that profile ran this same workload."""
    import glob
    # (newest = highest round tag in the name, r04e > r04a > r03e: a fresh checkout gives every
    # file the same modification time)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")),
                   key=os.path.basename, reverse=True)
    for path in paths:
        try:
            with open(path) as handle:
                summary = json.load(handle)
            if summary["bench_line"]["config"]["workload"] != workload:
                continue
            for name, counters in summary["counters"].items():
                if kernel in name and "hbm_bytes_per_launch" in counters:
                    PROFILED_RAW[kernel] = counters.get("hbm_bytes_per_launch_uncorrected")
                    return counters["hbm_bytes_per_launch"], os.path.basename(path)
        except (OSError, KeyError, TypeError, ValueError):
            continue
    return None, None


PROFILED_RAW = {}       # kernel -> FETCH_SIZE + WRITE_SIZE as counted (no gfx950 read correction)


def profiled_issue(workload, kernel="accumulate_kernel"):
    """fp64 VALU wave-instructions per launch of `kernel` (and busy cycles, when collected) from
    the newest profiles/*_valu_counters.json of this workload (scripts/profile_counters.sh)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_counters.json")),
                   key=os.path.basename, reverse=True)
    for path in paths:
        try:
            with open(path) as handle:
                summary = json.load(handle)
            if summary.get("workload") != workload:
                continue
            for name, entry in summary["kernels"].items():
                if kernel in name:
                    c = entry["mean_per_launch"]
                    fp64 = sum(c[x] for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
                                              "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
                    return {"fp64_wave_instructions_per_launch": fp64,
                            "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
                            "salu_wave_instructions_per_launch": c.get("SQ_INSTS_SALU"),
                            "evals_per_launch": summary.get("evals_per_accumulate_launch"),
                            "gui_active_cycles_per_xcd": entry.get("gui_active_cycles_per_xcd"),
                            "sclk_ghz_measured": entry.get("sclk_ghz_from_gui_active"),
                            "kernel": name,
                            "source": f"profiles/{os.path.basename(path)}"}
        except (OSError, KeyError, TypeError, ValueError):
            continue
    return None


def issue_slot_fraction(issue, launch_ms):
    """Fraction of the chip's fp64 issue slots a launch of `launch_ms` filled: a SIMD issues one
    fp64 wave-instruction per 4 cycles (16 lanes per cycle), so the ceiling is SIMDS x clock / 4
    wave-instructions per second -- at the datasheet's 2.4 GHz, and at the clock the profiled
    launch really ran at (GRBM_GUI_ACTIVE) when that was collected."""
    if not issue or not launch_ms:
        return None
    rate = issue["fp64_wave_instructions_per_launch"]/(launch_ms*1e-3)
    out = {"fp64_wave_instructions_per_launch": issue["fp64_wave_instructions_per_launch"],
           "frac_of_issue_slots_at_2.4GHz": rate/(SIMDS*BOOST_CLOCK_GHZ*1e9/4.),
           "kernel": issue.get("kernel"), "source": issue.get("source")}
    if issue.get("sclk_ghz_measured"):
        out["sclk_ghz_measured"] = issue["sclk_ghz_measured"]
        out["frac_of_issue_slots_at_measured_clock"] = \
            rate/(SIMDS*issue["sclk_ghz_measured"]*1e9/4.)
    return out


# ---------------------------------------------------------------------------------------------
# Device legs
# ---------------------------------------------------------------------------------------------
def lines_leg(engine, handles, tables, t, p, vmr, grid_args, steps, remove_pedestal=False,
              warmup=2, min_seconds=0., label="", ring=1):
    """`steps` passes (at least min_seconds) of prep + schedule + accumulate (+ pedestal) for
    every molecule over the given levels, spectra left in HBM; wall clock around a drained
    engine.  Returns evals/s, ms per step, spectra (levels) per second."""
    from pylbl_amd.engine import DeviceSpectra
    v0, vn, n_per_v = grid_args
    n = (vn - v0)*n_per_v
    levels = len(t)
    # `ring` sets of output blocks: successive steps write different memory, so the engine may
    # keep several calls in flight (it orders calls that write the same block).
    outs = [DeviceSpectra(engine, levels, n) for _ in range(ring) for _ in handles]
    evals = 0
    turn = [0]

    def step(count=False):
        total = 0
        first = (turn[0] % ring)*len(handles)
        turn[0] += 1
        for handle, table, out in zip(handles, tables, outs[first:first + len(handles)]):
            result = engine.compute(handle, t, p, vmr[table.formula], v0, vn, n_per_v,
                                    remove_pedestal=remove_pedestal, out=out, asynchronous=True,
                                    want_evals=count)
            if count:
                total += result[1]
        return total
    evals = step(count=True)
    # Asynchronous calls with a pedestal rotate over the engine's lanes (up to eight, option lanes), each with its own
    # workspace allocated at first use: warm all of them up, not only the first few.
    if remove_pedestal or ring > 1:
        warmup = max(warmup, -(-8//len(handles)) + 1)
    for _ in range(max(warmup - 1, 0)):
        step()
    engine.synchronize()
    done, elapsed = 0, 0.
    start = time.perf_counter()
    while True:
        for _ in range(steps):
            step()
        engine.synchronize()
        done += steps
        elapsed = time.perf_counter() - start
        if elapsed >= min_seconds:
            break
    for out in outs:
        out.free()
    return {"workload": label, "value": evals*done/elapsed, "unit": "evals/s",
            "ms_per_step": elapsed/done*1e3, "spectra_per_s": levels*done/elapsed,
            "steps": done, "evals_per_step": evals, "remove_pedestal": bool(remove_pedestal)}


def alone_roofline(engine, calls, evals_per_step, repeats=2):
    """The accumulate launches of one step run ALONE -- blocking calls, one lane, nothing beside
    them -- timed by HIP events on the stream they are launched on (engine option timing = 2):
    the roofline of a leg whose calls overlap on lanes inside its timed region.
    calls: [(handle, t, p, x, grid_args, keywords of Engine.compute)].
    achieved = SURVEY 8(d)'s 7 algorithmic flops per eval x the step's evals / the summed
    duration of the step's accumulate launches."""
    from pylbl_amd.engine import DeviceSpectra
    engine.synchronize()
    scratch = {}
    for handle, t, p, x, grid_args, keywords in calls:
        shape = (len(t), (grid_args[1] - grid_args[0])*grid_args[2])
        if shape not in scratch:
            scratch[shape] = DeviceSpectra(engine, *shape)
    engine.set_option("timing", 2)
    engine.timing(reset=True)
    for _ in range(repeats):
        for handle, t, p, x, grid_args, keywords in calls:
            shape = (len(t), (grid_args[1] - grid_args[0])*grid_args[2])
            engine.compute(handle, t, p, x, *grid_args, out=scratch[shape], **keywords)
    ms, launches = engine.timing(reset=True)
    engine.set_option("timing", 0)
    for block in scratch.values():
        block.free()
    per_step_ms = ms[2]/repeats
    tflops = evals_per_step*FLOPS_PER_EVAL/(per_step_ms*1e-3)/1e12
    return {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tflops/FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
            "kernel": "lbl::accumulate_kernel", "accumulate_ms_per_step_alone": per_step_ms,
            "avg_launch_ms": ms[2]/max(launches[2], 1), "launches_timed": launches[2],
            "farfield_series_ms_per_step_alone": ms[1]/repeats,
            "flops_per_eval": FLOPS_PER_EVAL,
            "note": "launches timed alone (blocking calls on one lane, HIP events on the "
                    "engine's stream), after the leg's timed region"}


def share_leg(engine, name, tables, handles, level_indices, levels_total, grid_args, output,
              steps, label, remove_pedestal=True, farfield=False):
    """One GPU's share of a multi-GPU BASELINE config, through the code the N-GPU job runs
    (pylbl_amd.distributed.ShardedLines.for_engine): the given levels of the build-owned standard
    atmosphere x every molecule of the config, spectra (output "gas") or their n k sum over the
    gases (output "total") left in torch-owned HBM.  Wall clock around `steps` asynchronous runs;
    roofline from the same launches timed alone."""
    from pylbl_amd import distributed, synthetic
    atmos = synthetic.standard_atmosphere(levels_total)
    t, p = atmos.t[level_indices], atmos.p[level_indices]
    vmr = {table.formula: atmos.vmr[table.formula][level_indices] for table in tables}
    by_formula = {table.formula: handles[i] for i, table in enumerate(tables)}
    sharded = distributed.ShardedLines.for_engine(
        engine, by_formula, grid_args, remove_pedestal=remove_pedestal,
        scale_density=(output == "total"), weights=[table.num_lines for table in tables],
        farfield=farfield)
    from pylbl_amd.engine import DeviceSpectra
    n = (grid_args[1] - grid_args[0])*grid_args[2]
    evals = 0
    scratch = DeviceSpectra(engine, len(t), n)
    for table in tables:
        _, count = engine.compute(by_formula[table.formula], t, p, vmr[table.formula], *grid_args,
                                  remove_pedestal=remove_pedestal, want_evals=True,
                                  farfield=farfield, out=scratch)
        evals += count
    scratch.free()
    pending = []
    for _ in range(2):          # every lane's workspace and both sets of blocks used once
        pending.append(sharded.run(t, p, vmr, output=output, async_op=True))
    for item in pending:
        item.wait()
    engine.synchronize()
    start = time.perf_counter()
    pending = [sharded.run(t, p, vmr, output=output, async_op=True) for _ in range(steps)]
    for item in pending:
        item.wait()
    engine.synchronize()
    elapsed = time.perf_counter() - start
    calls = [(by_formula[table.formula], t, p, vmr[table.formula], grid_args,
              {"remove_pedestal": remove_pedestal, "farfield": farfield}) for table in tables]
    roofline = alone_roofline(engine, calls, evals, repeats=1)
    if farfield:
        roofline["frac"] = None
    del sharded
    return {"workload": label, "value": evals*steps/elapsed, "unit": "evals/s",
            "ms_per_step": elapsed/steps*1e3, "spectra_per_s": len(level_indices)*steps/elapsed,
            "steps": steps, "evals_per_step": evals, "remove_pedestal": bool(remove_pedestal),
            "levels": [int(x) for x in level_indices], "levels_of_the_config": levels_total,
            "molecules": [table.formula for table in tables], "points": n, "output": output,
            "hbm_output_bytes": (1 if output == "total" else len(tables))*len(level_indices)*n*8,
            "through": "pylbl_amd.distributed.ShardedLines.for_engine (world 1)",
            "roofline": roofline}


def ingest_leg(engine, tables, db_path, atmos, grid_args):
    """What it costs to get a molecule's line table from the reference's SQLite file into HBM
    (line-table load time reported separately; the reference pays its read
    on every call, absorption.c:44-86): the three routes of pylbl_amd.database.line_table_of,
    the upload (lbl_molecule_load: sort by wavenumber, eleven arrays to HBM), and the
    same-signature C entry's first call on a file (SQLite read in C + upload + compute) against
    its second (line table found resident)."""
    from ctypes import c_char_p, c_double, c_int32
    from pylbl_amd import database
    out = {"database": f"SQLite file in the reference's schema, "
                       f"{'+'.join(f'{t.formula} {t.num_lines}' for t in tables)} transitions",
           "routes": {}, "per_molecule": {}}

    class PathOnly(object):             # what pyLBL.database.Database looks like from outside
        def __init__(self, path):
            self.path = path

    class QueriesOnly(object):          # a database object that cannot be opened as a file
        def __init__(self, inner):
            self.gas, self.tips = inner.gas, inner.tips
    file_backed = database.Database(db_path)
    routes = (("line_table", "an object with line_table(name) (this package's Database)",
               file_backed),
              ("path", "an object with .path only (pyLBL.database.Database as the reference hands "
                       "it over, spectroscopy.py:54): the C engine's own four SELECTs", PathOnly(db_path)),
              ("gas_tips", "an object with .gas(name) / .tips(name) only (record arrays; the "
                           "reference's ORM rows would add their own object construction)",
               QueriesOnly(file_backed)))
    loaded = {}
    for key, what, source in routes:
        seconds = {}
        for table in tables:
            start = time.perf_counter()
            loaded[table.formula] = database.line_table_of(source, table.formula)
            seconds[table.formula] = time.perf_counter() - start
        out["routes"][key] = {"what": what, "seconds": seconds, "total_s": sum(seconds.values())}
    upload = {}
    for table in tables:
        start = time.perf_counter()
        handle = engine.load(loaded[table.formula])
        upload[table.formula] = time.perf_counter() - start
        engine.free(handle)
    out["upload_s"] = upload
    # The drop-in C entry (absorption.c:19-30's signature): first call reads the file itself.
    lib = engine.lib
    v0, vn, n_per_v = grid_args
    k = np.zeros((vn - v0)*n_per_v)
    first, second = {}, {}
    for table in tables:
        args = (c_double(atmos.p[0]), c_double(atmos.t[0]), c_double(atmos.vmr[table.formula][0]),
                c_int32(v0), c_int32(vn), c_int32(n_per_v), k.ctypes.data,
                c_char_p(str(db_path).encode()), c_char_p(table.formula.encode()), c_int32(25),
                c_int32(0))
        for book in (first, second):
            start = time.perf_counter()
            status = lib.lbl_absorption(*args)
            book[table.formula] = time.perf_counter() - start
            if status != 0:
                raise RuntimeError("lbl_absorption failed")
    out["c_entry_first_call_s"] = first
    out["c_entry_second_call_s"] = second
    for table in tables:
        f = table.formula
        out["per_molecule"][f] = {
            "lines": int(table.num_lines),
            "read_s": out["routes"]["path"]["seconds"][f], "upload_s": upload[f],
            "c_entry_ingest_s": first[f] - second[f]}
    out["note"] = ("paid once per molecule and process (resident line tables; the C entry keys "
                   "them by path + mtime + inode); compare cpu_baseline.split."
                   "read_s_per_molecule, which the reference pays on every (level, molecule) call. "
                   "c_entry_*: host array in and out, so both calls include the 40 MB-class "
                   "copy back; their difference is the ingest")
    return out


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


def rccl_libraries():
    """File names of the RCCL libraries mapped into this process (empty: RCCL never loaded)."""
    try:
        with open("/proc/self/maps") as handle:
            return sorted({text.split("/")[-1].strip() for text in handle if "rccl" in text})
    except OSError:
        return []


def device_identity(torch, index):
    """What tells two GPUs apart: name, UUID and PCI address of HIP device `index` (each only
    where this torch exposes it)."""
    out = {"device_index": index}
    try:
        props = torch.cuda.get_device_properties(index)
    except Exception as error:          # diagnostics must not stop the run
        return dict(out, error=str(error))
    out["name"] = props.name
    for key in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id", "gcnArchName",
                "multi_processor_count"):
        value = getattr(props, key, None)
        if value is not None:
            out[key] = str(value) if key == "uuid" else value
    out["hip_visible_devices"] = os.environ.get("HIP_VISIBLE_DEVICES")
    return out


def launch_ranks(args, command=None):
    """`python bench.py --gpus N` with no launcher around it: starts the N ranks as CHILD
    processes (what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1` would start: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, one process per GPU), relays rank 0's JSON line on stdout and the other
    ranks' output on stderr, and returns the worst exit code.  The first rank that fails ends
    the others (a rank blocked in a collective cannot be woken), and so do --launch-timeout
    and a signal to this process.  Called before torch is imported or HIP touched: a process
    that holds a GPU must never start or become another program."""
    import signal
    import socket
    import subprocess
    import threading

    n = args.gpus
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(port), "GROUP_RANK": "0", "ROLE_RANK": "0", "NODE_RANK": "0",
                 "PYLBL_BENCH_LAUNCHER": "bench.py"})
    base.setdefault("OMP_NUM_THREADS", "1")             # as torch.distributed.run does
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs here
    if command is None:             # (tests pass a stand-in for the ranks' program)
        command = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    children, relays = [], []

    def relay(stream, target, prefix):
        for text in stream:
            target.write(prefix + text)
            target.flush()

    for rank in range(n):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        child = subprocess.Popen(command, env=env, stdout=subprocess.PIPE, text=True,
                                 start_new_session=True, cwd=os.getcwd())
        children.append(child)
        target, prefix = (sys.stdout, "") if rank == 0 else (sys.stderr, f"[rank {rank}] ")
        thread = threading.Thread(target=relay, args=(child.stdout, target, prefix), daemon=True)
        thread.start()
        relays.append(thread)

    def stop(sig):
        for child in children:
            if child.poll() is None:
                try:
                    os.killpg(child.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(number, frame):
        stop(signal.SIGTERM)
        time.sleep(2.)
        stop(signal.SIGKILL)
        os._exit(128 + number)
    for number in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(number, on_signal)

    deadline = time.monotonic() + args.launch_timeout
    reason = None
    own = []            # exit codes of the ranks that left by themselves
    while True:
        codes = [child.poll() for child in children]
        if all(code is not None for code in codes):
            break
        failed = [r for r, code in enumerate(codes) if code not in (None, 0)]
        if failed:
            reason = f"rank {failed[0]} left with code {codes[failed[0]]}"
        elif time.monotonic() > deadline:
            reason = f"no result after --launch-timeout {args.launch_timeout:g} s"
        if reason:
            # Let the others print what they were doing: until they have all left, or the grace
            # period is over.
            patience = time.monotonic() + args.launch_grace
            while time.monotonic() < patience and any(c.poll() is None for c in children):
                time.sleep(0.05)
            own = [code for code in (child.poll() for child in children) if code is not None]
            stop(signal.SIGTERM)
            patience = time.monotonic() + min(3., args.launch_grace)
            while time.monotonic() < patience and any(c.poll() is None for c in children):
                time.sleep(0.05)
            stop(signal.SIGKILL)
            for child in children:
                child.wait()
            break
        time.sleep(0.05)
    for thread in relays:
        thread.join(timeout=5.)
    codes = [child.returncode for child in children]
    if not reason:
        own = codes
        failed = [r for r, code in enumerate(codes) if code != 0]
        if failed:          # (every rank had left between two looks at them)
            reason = f"rank {failed[0]} left with code {codes[failed[0]]}"
    # The worst code among the ranks that left by themselves (the ones this launcher ended do not
    # count); 124, like timeout(1), when time ran out with none of them having failed.
    worst = max([(128 - code if code < 0 else code) for code in own] or [0])
    if reason:
        print(json.dumps({"bench_failed": True, "launcher": True, "reason": reason,
                          "exit_codes": codes}), file=sys.stderr, flush=True)
        worst = worst or (124 if "launch-timeout" in reason else 1)
    return worst


def main():
    """Runs the benchmark; a rank that fails says which rank it is and what it was doing, then
    leaves with a non-zero code so that the launcher stops the others (a rank blocked in a
    collective cannot be woken, and a process that holds a GPU is never re-executed)."""
    rank = os.environ.get("RANK", "0")
    try:
        run()
    except SystemExit:
        raise
    except BaseException as error:
        import traceback
        from pylbl_amd.distributed import ExchangeTimeout
        late = isinstance(error, ExchangeTimeout)
        print(json.dumps({"bench_failed": True, "rank": int(rank),
                          "world_size": int(os.environ.get("WORLD_SIZE", "1")),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "kind": type(error).__name__, "message": str(error),
                          "traceback": traceback.format_exc().splitlines()[-12:]}),
              file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3 if late else 1)


def first_level(job):
    """Handles in the workload's order and the first level's T, P and mixing ratios: what the
    one-level legs compute with."""
    handle_list = [job.handles[f] for f in job.molecules]
    vmr1 = {f: job.atmos.vmr[f][:1] for f in job.molecules}
    return handle_list, job.atmos.t[:1], job.atmos.p[:1], vmr1


def lines_option_legs(job, line, leg):
    """The timed step again under the conditions users meet: for >= 2 s, on one lane, with the
    pedestal removed, eight levels per call, banded and very dense line tables."""
    args, engine, tables, molecules = job.args, job.engine, job.tables, job.molecules
    atmos, grid_args, v_lo, v_hi, workload = job.atmos, job.grid_args, job.v_lo, job.v_hi, job.workload
    handle_list, t1, p1, vmr1 = first_level(job)
    from pylbl_amd import synthetic
    if leg("sustained"):
        line["sustained"] = lines_leg(
            engine, handle_list, tables, t1, p1, vmr1, grid_args, args.steps,
            remove_pedestal=args.pedestal, min_seconds=2.,
            label="the timed step repeated for >= 2 s (clocks at their sustained level)")
    if leg("overlap") and not args.pedestal and args.config == "target":
        # The timed step's plain calls take turns on two lanes (the next call's prologue and the
        # head of its accumulate grid beside the tail of this one's).  The same step with the
        # calls back to back on one stream (engine option overlap_plain = 0: every launch has
        # the chip to itself, as the launches `roofline` divides by), for the record.
        engine.set_option("overlap_plain", 0)
        try:
            line["single_lane_option"] = lines_leg(
                engine, handle_list, tables, t1, p1, vmr1, grid_args, args.steps,
                remove_pedestal=False, ring=2,
                label="the timed step with its calls back to back on one lane "
                      "(engine option overlap_plain = 0), two sets of output blocks")
        finally:
            engine.set_option("overlap_plain", 1)
    if leg("pedestal") and not args.pedestal:
        line["pedestal_option"] = lines_leg(
            engine, handle_list, tables, t1, p1, vmr1, grid_args, args.steps,
            remove_pedestal=True, ring=2,
            label="same workload with remove_pedestal=True (the default through "
                  "compute_absorption, spectroscopy.py:163-164), two sets of output blocks "
                  "used in turn like the timed step's")
    if leg("atmosphere"):
        standard = synthetic.standard_atmosphere(8)
        line["standard_atmosphere_option"] = lines_leg(
            engine, handle_list, tables, standard.t, standard.p,
            {f: standard.vmr[f] for f in molecules}, grid_args, max(args.steps//4, 2),
            remove_pedestal=True,
            label="8 standard-atmosphere levels (1013 hPa ... 0.1 hPa) in one batched call "
                  "per molecule, remove_pedestal=True")
    if leg("banded") and not args.banded:
        banded_tables = job.make_tables(True)
        banded_handles = [engine.load(t) for t in banded_tables]
        line["banded_table_option"] = lines_leg(
            engine, banded_handles, banded_tables, t1, p1, vmr1, grid_args, args.steps,
            remove_pedestal=True,
            label="same line counts clustered in 8 Gaussian bands per molecule "
                  "(synthetic.banded_line_table), remove_pedestal=True")
        for h in banded_handles:
            engine.free(h)
    if leg("dense") and args.config == "target":
        # A table several times denser than the workload's (dozens of pressure-shifted lines
        # alternate between two windows at every integer wavenumber): the step with and
        # without the pedestal, whose chain such tables used to send to its serial form.
        dense = [synthetic.banded_line_table("CO2", v_lo, v_hi, num_lines=1_600_000, bands=8,
                                             seed=5, inside=True)]
        dense_handles = [engine.load(t) for t in dense]
        dense_vmr = {"CO2": atmos.vmr["CO2"][:1]} if "CO2" in atmos.vmr else \
            {"CO2": np.asarray([3.6e-4])}
        dense_plain = lines_leg(engine, dense_handles, dense, t1, p1, dense_vmr, grid_args,
                          max(args.steps//2, 2), remove_pedestal=False, ring=2,
                          label="one molecule, 1.6 M lines in 8 Gaussian bands inside the grid "
                                "(synthetic.banded_line_table(inside=True))")
        with_pedestal = lines_leg(engine, dense_handles, dense, t1, p1, dense_vmr, grid_args,
                                  max(args.steps//2, 2), remove_pedestal=True, ring=2,
                                  label="the same with remove_pedestal=True")
        line["dense_table_option"] = {"plain": dense_plain, "remove_pedestal": with_pedestal,
                                      "lines": int(dense[0].num_lines)}
        for h in dense_handles:
            engine.free(h)


def small_grid_legs(job, line):
    """BASELINE configs[0] and [1] as rings of asynchronous calls, and configs[0] as a replayed
    HIP graph against plain launches."""
    engine, atmos = job.engine, job.atmos
    handle_list, t1, p1, vmr1 = first_level(job)
    from pylbl_amd import synthetic
    small = {}
    for name in ("0", "1"):
        mols, lo, hi, step_cm, _ = CONFIGS[name]
        ga = synthetic.grid_arguments(np.asarray([lo, lo + step_cm, hi - step_cm]))
        small_tables = [synthetic.line_table(f, lo, hi) for f in mols]
        small_handles = [engine.load(t) for t in small_tables]
        small[f"config{name}"] = lines_leg(
            engine, small_handles, small_tables, t1, p1,
            {f: atmos.vmr[f][:1] for f in mols}, ga, 50, min_seconds=0.3, ring=4,
            label=f"BASELINE configs[{name}]: {'+'.join(mols)}, {lo:g}-{hi:g} cm-1 at "
                  f"{step_cm:g} cm-1; throughput of asynchronous calls into a ring of 4 "
                  f"output blocks")
        if name == "0":
            # The three-kernel call as a replayed HIP graph (engine option graphs): the
            # ring of asynchronous calls again, and the blocking call that returns a host
            # array -- what the reference's caller sees (gas_optics.py:61-91) -- timed call
            # by call, with the option off and on.
            entry = small["config0"]
            x0 = atmos.vmr[mols[0]][:1]
            entry["graph_replay_option"] = {}
            for graphs in (0, 1):
                engine.set_option("graphs", graphs)
                for _ in range(50):
                    engine.compute(small_handles[0], t1, p1, x0, *ga)
                times = []
                for _ in range(400):
                    begin = time.perf_counter()
                    engine.compute(small_handles[0], t1, p1, x0, *ga)
                    times.append(time.perf_counter() - begin)
                ring = lines_leg(engine, small_handles, small_tables, t1, p1,
                                 {f: atmos.vmr[f][:1] for f in mols}, ga, 50,
                                 min_seconds=0.3, ring=4) if graphs else entry
                entry["graph_replay_option"]["on" if graphs else "off"] = {
                    "ring_evals_per_s": ring["value"],
                    "ring_us_per_call": ring["ms_per_step"]*1e3/len(small_handles),
                    "blocking_call_us_median": float(np.median(times))*1e6,
                    "blocking_call_us_min": min(times)*1e6}
            engine.set_option("graphs", 0)
            entry["graph_replay_option"]["shipped"] = "off (engine option graphs = 0)"
        for h in small_handles:
            engine.free(h)
    line["small_grid_options"] = small


def farfield_legs(job, line):
    """The step with the far-field series (what Spectroscopy runs by default), with and without
    the pedestal, each with the issue-slot roofline of its accumulate launches run alone."""
    args, engine = job.args, job.engine
    tables, grid_args, workload = job.tables, job.grid_args, job.workload
    handle_list, t1, p1, vmr1 = first_level(job)
    engine.set_option("farfield", 1)
    far = {}
    for ped in (False, True):
        key = "remove_pedestal" if ped else "plain"
        far[key] = lines_leg(
            engine, handle_list, tables, t1, p1, vmr1, grid_args, args.steps,
            remove_pedestal=ped)
        # What the series leaves to be executed point by point is no longer "7 flops x the
        # closed-form evals": the fraction is the share of the chip's fp64 ISSUE SLOTS the
        # launch filled -- executed fp64 wave-instructions (PMC pass of this same
        # workload, profiles/) over the launch's duration here, timed alone.
        calls = [(h, t1, p1, vmr1[tb.formula], grid_args, {"remove_pedestal": ped})
                 for h, tb in zip(handle_list, tables)]
        alone = alone_roofline(engine, calls, far[key]["evals_per_step"], repeats=3)
        far_workload = workload.replace(
            "remove_pedestal=False", f"remove_pedestal={ped}") + ", far-field series on"
        roof = {"bound": "valu_fp64_issue", "unit": "fraction of fp64 issue slots",
                "kernel": "lbl::accumulate_kernel<8>",
                "avg_launch_ms": alone["avg_launch_ms"],
                "accumulate_ms_per_step_alone": alone["accumulate_ms_per_step_alone"],
                "farfield_series_ms_per_step_alone":
                    alone["farfield_series_ms_per_step_alone"],
                "launches_timed": alone["launches_timed"], "frac": None, "traffic": None}
        issue = issue_slot_fraction(profiled_issue(far_workload), alone["avg_launch_ms"])
        if issue is not None:
            roof["issue"] = issue
            roof["frac"] = issue.get("frac_of_issue_slots_at_measured_clock",
                                     issue["frac_of_issue_slots_at_2.4GHz"])
            roof["achieved"], roof["peak"] = roof["frac"], 1.0
        for kernel in ("farfield_kernel", "farfield_group_kernel"):
            counted, source = profiled_traffic(far_workload, kernel)
            if counted is not None:
                roof.setdefault("series_kernels", {})[kernel] = {
                    "hbm_bytes_per_launch": counted, "source": f"profiles/{source}"}
        if "series_kernels" in roof and alone["farfield_series_ms_per_step_alone"] > 0.:
            moved = sum(v["hbm_bytes_per_launch"] for v in roof["series_kernels"].values())
            # (one launch of each per molecule call)
            seconds = alone["farfield_series_ms_per_step_alone"]*1e-3/len(handle_list)
            roof["series_kernels"]["hbm"] = {
                "bound": "hbm", "achieved": moved/seconds/1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": moved/seconds/1e9/HBM_PEAK_GBS,
                "note": "HBM bytes of the two series kernels (PMC) / their duration here"}
        roof["note"] = (
            "far-field series on: frac = executed fp64 wave-instructions per "
            "accumulate_kernel<8> launch (rocprofv3 --pmc pass of this workload, "
            "profiles/) x 4 cycles / (1024 SIMDs x shader clock x launch duration, HIP "
            "events, launches run alone); None until a counter summary of this exact "
            "workload is committed")
        far[key]["roofline"] = roof
    engine.set_option("farfield", 0)
    far["note"] = ("engine option farfield=1 (pylbl_amd/csrc/farfield.h): lines at least 4 "
                   "tile half-widths away are summed as one power series per tile "
                   "(truncation <= ~1.5e-11 relative); same closed-form eval count; "
                   "parity-tested at the same 1e-6 bar; what Spectroscopy(farfield=True) "
                   "runs -- remove_pedestal is what a user of compute_absorption() gets "
                   "by default (spectroscopy.py:163-164); never the headline value")
    line["farfield_option"] = far


def other_config_legs(job, line, leg):
    """BASELINE configs[2], [3] and [4] at one GPU's size, on the driver-run line."""
    args, engine, tables = job.args, job.engine, job.tables
    handles, molecules, grid_args = job.handles, job.molecules, job.grid_args
    handle_list, t1, p1, vmr1 = first_level(job)
    from pylbl_amd import distributed, synthetic
    if args.config == "target" and (leg("config2") or leg("config4")):
        # The other BASELINE configs at one GPU's size (the eight README molecules on
        # 1-5000 cm-1 serve configs[2] and configs[4]).
        eight = [t for t in tables if t.formula in EIGHT]
        have = {t.formula for t in eight}
        eight += [synthetic.line_table(f, 1., 5000., scale=args.line_scale)
                  for f in EIGHT if f not in have]
        eight.sort(key=lambda t: EIGHT.index(t.formula))
        eight_handles = [handles[t.formula] if t.formula in handles else engine.load(t)
                         for t in eight]
        if leg("config2"):
            surface = synthetic.surface_level()
            vmr8 = {f: surface.vmr[f][:1] for f in EIGHT}
            entry = lines_leg(
                engine, eight_handles, eight, t1, p1, vmr8, grid_args, max(args.steps//2, 3),
                remove_pedestal=False,
                label="BASELINE configs[2]: 1 level, all 8 README molecules "
                      f"({'+'.join(EIGHT)}), 1-5000 cm-1 at 0.001 cm-1 (5 M points), "
                      "remove_pedestal=False like the headline")
            calls = [(h, t1, p1, vmr8[tb.formula], grid_args, {"remove_pedestal": False})
                     for h, tb in zip(eight_handles, eight)]
            entry["roofline"] = alone_roofline(engine, calls, entry["evals_per_step"])
            entry["lines"] = {t.formula: int(t.num_lines) for t in eight}
            line["config2_option"] = entry
        if leg("config4"):
            ga4 = synthetic.grid_arguments(np.asarray([1., 1.0005, 5000. - 0.0005]))
            rank3 = distributed.level_shard(256, 3, 8)
            picked = list(range(rank3.start, rank3.stop, 8))        # 96, 104, 112, 120
            line["config4_share_option"] = share_leg(
                engine, "4", eight, eight_handles, picked, 256, ga4, "total",
                max(args.steps//6, 3),
                label="BASELINE configs[4] (256 levels x 8 molecules, 1-5000 cm-1 at 0.0005 "
                      "cm-1 = 10 M points, over 8 GPUs): 4 of rank 3's 32 levels "
                      f"(levels {picked} of the 256-level standard atmosphere) x 8 molecules, "
                      "remove_pedestal=True, n k summed over the gases on the device "
                      "(output 'total')")
        for t, h in zip(eight, eight_handles):
            if t.formula not in handles:
                engine.free(h)
    if args.config == "target" and leg("config3"):
        mols3, lo3, hi3, dv3, levels3 = CONFIGS["3"]
        ga3 = synthetic.grid_arguments(np.asarray([lo3, lo3 + dv3, hi3 - dv3]))
        tables3 = [synthetic.line_table(f, lo3, hi3, scale=args.line_scale) for f in mols3]
        handles3 = [engine.load(t) for t in tables3]
        shares = {}
        for share_rank in (0, 7):
            block = distributed.level_shard(levels3, share_rank, 8)
            picked = list(range(block.start, block.stop))
            shares[share_rank] = share_leg(
                engine, "3", tables3, handles3, picked, levels3, ga3, "gas",
                max(args.steps//6, 3),
                label=f"BASELINE configs[3] (64-level standard atmosphere, {'+'.join(mols3)}, "
                      f"1-3000 cm-1 at 0.001 cm-1 = 3 M points, levels sharded over 8 GPUs): "
                      f"rank {share_rank}'s share, levels {picked[0]}-{picked[-1]} "
                      f"({'1013-330 hPa' if share_rank == 0 else '0.3-0.1 hPa: the slowest share, it bounds the job'}), "
                      "remove_pedestal=True, one spectrum per gas left in HBM")
        # The share that bounds the 8-GPU job is the record's entry; rank 0's rides along.
        line["config3_share_option"] = dict(shares[7], rank0_share=shares[0])
        for h in handles3:
            engine.free(h)


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


def cpu_legs(job, line, shared_db):
    """The CPU baselines timed on this box's host cores: the reference's own C on one thread,
    the C restatement on 16 processes, and on every core the process may use."""
    args, tables, atmos, molecules = job.args, job.tables, job.atmos, job.molecules
    grid_v0, grid_vn, n_per_v = job.grid_args
    v_lo, v_hi = job.v_lo, job.v_hi
    db = shared_db
    line["cpu_baseline"] = cpu_baseline(tables, atmos, grid_v0, grid_vn, n_per_v,
                                        args.cpu_sample_cm, args.pedestal, db=db)
    workers = max(1, min(args.cpu_workers, os.cpu_count() or 1))
    if workers > 1:
        line["cpu_baseline_parallel"] = cpu_baseline_parallel(
            tables, atmos, grid_v0, grid_vn, n_per_v, args.cpu_sample_cm, workers)
    # "All host cores" = what this process may use: the affinity mask, cut down to the
    # cgroup's CPU quota where there is one (this pool shows a one-GPU job all 256 hardware
    # threads of the host and allots it 16 cores' worth of time: 256 processes then share
    # those, 8.3e9 evals/s against 1.7e10 for 16 -- profiles/bench_r05b.json).
    usable = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    if quota is not None:
        usable = max(1, min(usable, int(round(quota))))
    every = usable if args.cpu_all_cores < 0 else args.cpu_all_cores
    if 0 < every <= workers and "cpu_baseline_parallel" in line:
        line["cpu_baseline_all_cores"] = dict(
            line["cpu_baseline_parallel"],
            note=f"every core this process may use: affinity mask "
                 f"{len(os.sched_getaffinity(0))} hardware threads, cgroup CPU quota "
                 f"{quota} cores -> {usable}; cpu_baseline_parallel's {workers} processes "
                 f"already use them (the figure is the same run); --cpu-all-cores N forces "
                 f"a pool of N")
    if every > workers:
        line["cpu_baseline_all_cores"] = cpu_baseline_parallel(
            tables, atmos, grid_v0, grid_vn, n_per_v, args.cpu_sample_cm, every,
            timeout=args.cpu_pool_timeout,
            why="--cpu-all-cores: every hardware thread this process may run on",
            recipes={f: (f, v_lo, v_hi, args.line_scale, bool(args.banded), i)
                     for i, f in enumerate(molecules)})


def headline(job, m):
    """Rank 0's JSON line from the timed region: value, the contract's keys, who ran where, and
    the roofline of the accumulate kernel -- from the same launches run alone after the region
    (asynchronous calls take turns on two lanes, so inside it a launch is never alone), with the
    PMC passes committed under profiles/ for traffic and issue slots.

    m: what the timed region measured (elapsed = max over ranks, evals_per_step = sum over ranks,
    this rank's kernel_ms / launches from the engine's events, per_rank records)."""
    import torch.distributed as dist
    args, engine, tables, handles, molecules = job.args, job.engine, job.tables, job.handles, job.molecules
    atmos, grid_args, v_lo, v_hi, dv = job.atmos, job.grid_args, job.v_lo, job.v_hi, job.dv
    levels_local, levels_total, n = job.levels_local, job.levels_total, job.n
    rank, world, plan, sharded, vmr = job.rank, job.world, job.plan, job.sharded, job.vmr
    elapsed, evals_per_step, evals_per_step_local = m.elapsed, m.evals_per_step, m.evals_per_step_local
    kernel_ms, launches, per_rank, busy_ms = m.kernel_ms, m.launches, m.per_rank, m.busy_ms
    grouped, everyone, shared = m.grouped, m.everyone, m.shared
    ms_per_step = elapsed/args.steps*1e3
    value = evals_per_step*args.steps/elapsed
    accumulate_ms = kernel_ms[2]/max(launches[2], 1)
    evals_per_launch = evals_per_step_local/max(launches[2]/args.steps, 1)
    tflops = evals_per_launch*FLOPS_PER_EVAL/(accumulate_ms*1e-3)/1e12
    algorithmic = evals_per_launch*BYTES_PER_EVAL/(accumulate_ms*1e-3)/1e9
    workload = (f"BASELINE config '{args.config}': {levels_local} level(s) per GPU, "
                f"{'+'.join(molecules)}, grid {v_lo:g}-{v_hi:g} cm-1 at {dv:g} cm-1 "
                f"({n} points), cut_off 25, remove_pedestal={args.pedestal}"
                + (", far-field series on" if args.farfield else "")
                + (", banded tables" if args.banded else ""))
    line = {
        "metric": "line×gridpoint Voigt evals/sec (whole job; per GPU: evals_per_s_per_gpu; spectra/sec: spectra_per_s)",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": workload,
            "lines": {t.formula: t.num_lines for t in tables},
            "levels_total": levels_total, "atmosphere": args.profile,
            "output": args.output,
            "parallelism": f"(level, molecule) units over {world} GPU(s): "
                           f"{plan.mode} sharded"
            + (f", one grouped {args.backend} send/recv to rank 0 per step, overlapping the "
               f"next step" if world > 1 else ""),
        },
        "distributed": None if not grouped else {
            "world_size": dist.get_world_size(), "backend": dist.get_backend(),
            "launcher": os.environ.get("PYLBL_BENCH_LAUNCHER") or (
                "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                else "environment"),
            "distinct_devices": len({(r.get("host"), r.get("uuid") or r.get("pci_bus_id"),
                                      r.get("device_index")) for r in everyone}),
            "ranks_sharing_a_device": shared,
            "kernels_to_exchange_ordering": "device (events between the engine's streams and "
                                            "the exchange's, no host wait)"
            if (args.backend == "nccl" or sharded.order_on_device) else "host (synchronize)",
            "exchange_timeout_s": args.exchange_timeout,
            "rccl_mapped": rccl_libraries(),
            "bytes_to_rank0_per_step": (per_rank or [{}])[0].get("bytes_received_per_step"),
            "exchange_alone_ms_max": max((r["unoverlapped_exchange_ms"]
                                          for r in per_rank), default=None)
            if per_rank else None,
            "note": "per rank: the device it ran on, its own wall time for the timed steps, "
                    "bytes it sent/received per step, host time it spent waiting for an "
                    "exchange inside the timed steps (exchange_wait_ms_per_step; 0 = fully "
                    "hidden behind the next step's kernels) and one un-overlapped step "
                    "(kernels, then the collection alone) measured after the timed region",
            "ranks": per_rank if per_rank else everyone,
        },
        "evals_per_step": evals_per_step,
        "evals_per_s_per_gpu": value/world,
        "spectra_per_s": levels_total*args.steps/elapsed,
        "roofline": {
            "bound": "valu_fp64", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tflops/FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
            "kernel": "lbl::accumulate_kernel", "avg_launch_ms": accumulate_ms,
            "launches_timed": launches[2], "flops_per_eval": FLOPS_PER_EVAL,
            "evals_per_launch": evals_per_launch,
            "note": "the kernel keeps partial sums in registers and writes k once, so HBM "
                    "carries ~8 B per grid point (traffic, from the PMC counters) and the "
                    "binding resource is the fp64 vector ALU: achieved = SURVEY 8(d)'s 7 "
                    "algorithmic flops per eval (5 common + 2 far-wing incl. the divide) x "
                    "evals per launch / mean launch time (HIP events on the engine's stream); "
                    "peak = datasheet fp64 vector rate at 2.4 GHz",
        },
        "roofline_hbm_algorithmic": {
            "bound": "hbm", "achieved": algorithmic, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algorithmic/HBM_PEAK_GBS,
            "note": "SURVEY 8(d) as written: 24 B per eval (the reference's load v[i], "
                    "load+store k[i]) x evals / launch time.  These bytes never move here "
                    "(register accumulation), so the 'fraction' exceeds 1 and is not a "
                    "bandwidth; north_star's '>= 40 % of the HBM roofline' is 1.33e11 evals/s",
        },
        "kernel_ms_per_step": {
            "prepare": kernel_ms[0]/args.steps, "schedule": kernel_ms[1]/args.steps,
            "accumulate": kernel_ms[2]/args.steps, "pedestal": kernel_ms[3]/args.steps},
    }
    traffic, source = profiled_traffic(workload)
    if traffic is not None:
        line["roofline"]["traffic"] = traffic
        line["roofline"]["traffic_source"] = f"profiles/{source}"
        line["roofline"]["traffic_uncorrected"] = PROFILED_RAW.get("accumulate_kernel")
        line["roofline"]["traffic_note"] = (
            "traffic = WRITE_SIZE + 2 x FETCH_SIZE (the gfx950 correction for wide coalesced "
            "reads); this kernel reads its line records through scalar loads, for which the "
            "uncorrected count (traffic_uncorrected = WRITE_SIZE + FETCH_SIZE) may be the truer "
            "one -- either way 0.3-0.7 TB/s, a tenth of the HBM roofline")
    issue = profiled_issue(workload)
    if issue is not None and issue.get("evals_per_launch"):
        per_eval = issue["fp64_wave_instructions_per_launch"]*64./issue["evals_per_launch"]
        # One fp64 wave-instruction occupies a SIMD's issue port for 4 cycles (16 lanes/cycle).
        ceiling = SIMDS*BOOST_CLOCK_GHZ*1e9/4.*64./per_eval
        issue.update({
            "fp64_wave_instructions_per_64_evals": per_eval,
            "issue_ceiling_evals_per_s_at_2.4GHz": ceiling,
            "frac_of_issue_ceiling_at_2.4GHz": evals_per_launch/(accumulate_ms*1e-3)/ceiling})
        if issue.get("gui_active_cycles_per_xcd"):
            # Busy cycles of the profiled launch (GRBM_GUI_ACTIVE / 8 XCDs): the fraction of
            # a SIMD's 4-cycle issue slots that fp64 instructions occupied at the clock the
            # chip actually ran.
            per_simd = issue["fp64_wave_instructions_per_launch"]/SIMDS*4.
            issue["frac_of_issue_slots_at_measured_clock"] = \
                per_simd/issue["gui_active_cycles_per_xcd"]
        line["roofline"]["issue"] = issue
    if not args.host_output and launches[2] > 0:
        # Asynchronous calls that leave their spectra in HBM take turns on the engine's lanes:
        # the tail of one accumulate grid and the head of the next overlap in time, and an
        # event-timed launch is stretched by its neighbour.  The fraction is therefore taken
        # from the same launches run alone (blocking calls, one lane), outside the timed
        # region; what the events read inside it is kept beside it, and so is the fraction
        # that follows from the step time alone (every kernel of the step in the denominator).
        from pylbl_amd.engine import DeviceSpectra
        engine.set_option("timing", 2)
        engine.timing(reset=True)
        # (3 to 10 launches per molecule: about 50 ms of them, so that the mean does not hang
        # on one launch's clock)
        alone_repeats = int(min(10, max(3, 50./max(ms_per_step, 1e-3))))
        for m, levels in plan.by_molecule(rank).items():
            scratch = DeviceSpectra(engine, len(levels), n)
            for _ in range(alone_repeats):
                engine.compute(handles[molecules[m]], atmos.t[levels], atmos.p[levels],
                               vmr[molecules[m]][levels], *grid_args,
                               remove_pedestal=args.pedestal, out=scratch)
            scratch.free()
        alone_ms, alone_launches = engine.timing(reset=True)
        engine.set_option("timing", 0)
        alone = alone_ms[2]/max(alone_launches[2], 1)
        alone_tflops = evals_per_launch*FLOPS_PER_EVAL/(alone*1e-3)/1e12
        line["roofline"].update({
            "achieved": alone_tflops, "frac": alone_tflops/FP64_VECTOR_PEAK_TFLOPS,
            "avg_launch_ms": alone, "launches_timed": alone_launches[2],
            "avg_launch_ms_overlapped_in_step": accumulate_ms,
            "frac_from_overlapped_launches": tflops/FP64_VECTOR_PEAK_TFLOPS,
            "frac_from_step_time": (evals_per_step_local*FLOPS_PER_EVAL/(ms_per_step*1e-3)
                                    / 1e12/FP64_VECTOR_PEAK_TFLOPS),
            # HIP events over the timed region itself: the time during which at least one
            # accumulate launch was running (the union of the launches' intervals on the device's
            # clock, lbl_timing_busy), and the kernel's rate over exactly that time.
            "accumulate_busy_ms_per_step_in_region": busy_ms[2]/args.steps,
            "frac_while_running_in_region": (evals_per_step_local*FLOPS_PER_EVAL*args.steps
                                             / max(busy_ms[2]*1e-3, 1e-12)/1e12
                                             / FP64_VECTOR_PEAK_TFLOPS)})
        line["roofline"]["note"] += (
            "; the calls of the timed region take turns on two (with a pedestal pass: four) "
            "engine lanes, so that the tail of one accumulate grid runs beside the next call's "
            "prologue and the head of its grid: ms_per_step is SHORTER than the sum of the "
            "launches run alone.  achieved / frac / avg_launch_ms come from the same launches "
            "run alone after the timed region (blocking calls on one lane: what rocprofv3 shows "
            "for a launch that has the chip to itself); avg_launch_ms_overlapped_in_step is "
            "what the events read inside the region (two grids side by side), and "
            "frac_from_step_time = 7 flops x evals_per_step / ms_per_step / peak, which needs "
            "no launch taken alone; frac_while_running_in_region = 7 flops x the region's evals / "
            "the time at least one accumulate launch was running inside the timed region (union "
            "of the launches' event intervals) / peak: the kernel's rate measured over the timed "
            "region itself, overlap counted once")
    if args.farfield:
        # The series replaces most evaluations by one polynomial per point: "7 flops per eval x
        # evals" is not what the kernel executes, and the quotient is not a fraction of a peak.
        line["roofline"]["frac"] = None
        line["roofline"]["note"] += ("; far-field series on: most of the evaluations counted in "
                                     "`value` are not executed one by one, so `achieved` is not "
                                     "a rate of executed flops and no fraction is given")
    if args.host_output:
        line["INVALID"] = "host output: PCIe copies inside the step (reported for DESIGN.md)"
    if args.ablate:
        line["INVALID"] = f"ablation {args.ablate}: part of the work was skipped"
    return line


def run():
    args = parse()
    if args.no_extras:
        args.extras = "none"
    wanted = set(args.extras.split(","))
    every = "all" in wanted

    def leg(name):
        return every or name in wanted

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.
        # Nothing here has touched HIP or imported torch yet, and it never will -- the ranks
        # are children, this process only relays their output and exit code.
        raise SystemExit(launch_ranks(args))
    if world != args.gpus:
        args.gpus = world

    # stdout carries the one JSON line and nothing else: whatever libraries write to file
    # descriptor 1 from here on (gloo announces its connections there) goes to stderr.
    sys.stdout.flush()
    result_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # (dmabuf IPC is what RCCL needs on this pool; the launchers export it, a bare environment may not)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback).")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    identity = dict(device_identity(torch, device_index), rank=rank, local_rank=local_rank,
                    host=os.uname().nodename, pid=os.getpid())
    # --force-group: the process group (and with nccl: RCCL, a communicator, its streams) also
    # for ONE rank -- what a one-GPU box can rehearse of the N-GPU run before the first lease
    # of a whole node.
    grouped = world > 1 or args.force_group
    if grouped:
        from datetime import timedelta
        if world == 1:
            import socket
            with socket.socket() as probe:
                probe.bind(("127.0.0.1", 0))
                free_port = probe.getsockname()[1]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("PYLBL_AMD_EXCHANGE_TIMEOUT", str(args.exchange_timeout))
        limit = timedelta(seconds=max(args.exchange_timeout, 30.))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index),
                                    timeout=limit)
        else:
            dist.init_process_group("gloo", timeout=limit)
        # Who is here: every rank's device, before anything is computed.  N ranks must sit on N
        # different GPUs for the N-GPU figure to mean anything (RCCL refuses two ranks on one
        # device; gloo does not).
        everyone = [None]*world
        dist.all_gather_object(everyone, identity)
        seen = {}
        for other in everyone:
            key = (other.get("host"), other.get("uuid") or other.get("pci_bus_id"),
                   other.get("device_index"))
            seen.setdefault(key, []).append(other["rank"])
        shared = {str(k): v for k, v in seen.items() if len(v) > 1}
        if shared and args.backend == "nccl":
            raise RuntimeError(f"ranks share a GPU: {shared}")
        if dist.get_world_size() != world:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, the "
                               f"launcher announced {world}")
    else:
        everyone, shared = [identity], {}

    from pylbl_amd import distributed, synthetic
    from pylbl_amd.engine import Engine
    fixture = os.path.join(ROOT, "tests", "golden", "mt_ckd_bands.npz")
    if os.path.isfile(fixture):
        os.environ.setdefault("PYLBL_MT_CKD", fixture)      # continuum coefficients (api leg)

    molecules, v_lo, v_hi, dv, _ = CONFIGS[args.config]
    grid_v0, grid_vn, n_per_v = synthetic.grid_arguments(np.asarray([v_lo, v_lo + dv, v_hi - dv]))
    grid_args = (grid_v0, grid_vn, n_per_v)
    n = (grid_vn - grid_v0)*n_per_v
    levels_local = args.levels_per_gpu
    levels_total = levels_local*world
    atmos = atmosphere_for(levels_total, args.profile)

    def make_tables(banded):
        out = []
        for i, f in enumerate(molecules):
            uniform = synthetic.line_table(f, v_lo, v_hi, scale=args.line_scale)
            if banded:
                out.append(synthetic.banded_line_table(f, v_lo, v_hi, num_lines=uniform.num_lines,
                                                       bands=8, seed=41 + i))
            else:
                out.append(uniform)
        return out
    tables = make_tables(args.banded)
    # The process-wide engine of the device: the one Spectroscopy / Gas objects use too (api leg),
    # so that the process has one set of streams.
    from pylbl_amd.engine import default_engine
    engine = default_engine(device_index)
    if args.points_per_lane:
        engine.set_option("points_per_lane", args.points_per_lane)
    for pair in args.engine_option:
        name, value = pair.split("=")
        engine.set_option(name, int(value))
    if args.farfield:
        engine.set_option("farfield", 1)
    if args.ablate:
        engine.set_option("ablate", args.ablate)
    handles = {t.formula: engine.load(t) for t in tables}

    # The product path for any N: (level, molecule) units partitioned over the ranks, spectra
    # written by the engine into torch-owned HBM, one grouped send/recv to rank 0 per step.
    sharded = distributed.ShardedLines.for_engine(
        engine, handles, grid_args, remove_pedestal=args.pedestal,
        scale_density=(args.output == "total"), weights=[t.num_lines for t in tables],
        always_exchange=args.force_group)
    vmr = {f: atmos.vmr[f] for f in molecules}
    plan = distributed.partition(levels_total, [t.num_lines for t in tables], world)
    host_spectra = engine.host_array((len(molecules), levels_local, n)) if args.host_output \
        else None
    pending = [None, None]
    counter = [0]

    def count_evals():
        """Closed-form evals of this rank's units (the engine's own count, one blocking pass)."""
        from pylbl_amd.engine import DeviceSpectra
        total = 0
        for m, levels in plan.by_molecule(rank).items():
            formula = molecules[m]
            # (spectra into a scratch block in HBM: no 40 MB-class copy to the host for a count)
            scratch = DeviceSpectra(engine, len(levels), n)
            _, evals = engine.compute(handles[formula], atmos.t[levels], atmos.p[levels],
                                      vmr[formula][levels], *grid_args,
                                      remove_pedestal=args.pedestal, want_evals=True, out=scratch)
            scratch.free()
            total += evals
        return total

    def step():
        which = counter[0] % 2
        counter[0] += 1
        if grouped and pending[which] is not None:
            # The exchange that last used this pair of buffers: settled here so that its bytes
            # and seconds are booked (ShardedLines itself orders a buffer's next writer behind
            # the exchange that last used it).  (One rank, no group: the engine's own streams
            # order successive writes to a buffer, nothing to wait for.)
            settle(which)
        if args.host_output:
            for m, formula in enumerate(molecules):
                engine.compute(handles[formula], atmos.t, atmos.p, vmr[formula], *grid_args,
                               remove_pedestal=args.pedestal, out=host_spectra[m])
            return
        pending[which] = sharded.run(atmos.t, atmos.p, vmr, dst=0, output=args.output,
                                     async_op=True)

    exchange = {"count": 0, "wait_s": 0., "in_flight_s": 0., "sent": 0, "received": 0}

    def settle(which):
        """Waits for the exchange that owns buffer pair `which` and books what it moved."""
        item = pending[which]
        pending[which] = None
        begin = time.perf_counter()
        item.wait(timeout=args.exchange_timeout)
        exchange["wait_s"] += time.perf_counter() - begin
        exchange["in_flight_s"] += item.seconds or 0.
        exchange["count"] += 1
        exchange["sent"] += item.bytes_sent
        exchange["received"] += item.bytes_received

    def fence():
        for which in (0, 1):
            if pending[which] is not None:
                settle(which)
        engine.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    evals_per_step_local = count_evals()
    # (With --pedestal the calls rotate over the engine's lanes (up to eight, option lanes): their workspaces are
    # allocated at first use, so the warm-up has to reach all of them.)
    warm = max(args.warmup, 1)
    if args.pedestal and not args.host_output:
        warm = max(warm, -(-8//max(len(plan.by_molecule(rank)), 1)) + 1)
    for _ in range(warm):
        step()
    fence()
    # HIP events around the accumulate launches only (option value 2): events between all five
    # kernels of a call keep them from running back to back.
    engine.set_option("timing", 2)
    engine.timing(reset=True)
    fence()
    for key in exchange:
        exchange[key] = 0
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - start
    busy_ms = engine.timing_busy()
    kernel_ms, launches = engine.timing(reset=True)
    engine.set_option("timing", 0)

    per_rank = None
    if grouped and not args.host_output:
        # The exchange by itself, once, outside the timed region: kernels first (host waits),
        # then the collection alone -- what a step would pay if nothing overlapped it.
        fence()
        begin = time.perf_counter()
        alone = sharded.run(atmos.t, atmos.p, vmr, dst=0, output=args.output, async_op=True)
        engine.synchronize()
        computed = time.perf_counter()
        alone.wait(timeout=args.exchange_timeout)
        finished = time.perf_counter()
        mine = dict(identity, seconds=elapsed, ms_per_step=elapsed/args.steps*1e3,
                    evals_per_step=int(evals_per_step_local),
                    units=len(plan.units[rank]), levels=len(plan.levels_of(rank)),
                    exchanges=exchange["count"],
                    bytes_sent_per_step=exchange["sent"]/max(exchange["count"], 1),
                    bytes_received_per_step=exchange["received"]/max(exchange["count"], 1),
                    exchange_wait_ms_per_step=exchange["wait_s"]/args.steps*1e3,
                    exchange_in_flight_ms=exchange["in_flight_s"]/max(exchange["count"], 1)*1e3,
                    unoverlapped_compute_ms=(computed - begin)*1e3,
                    unoverlapped_exchange_ms=(finished - computed)*1e3,
                    accumulate_ms_per_step=kernel_ms[2]/args.steps)
        per_rank = [None]*world
        dist.all_gather_object(per_rank, mine)
        fence()

    # (One rank: nothing to reduce, and no torch kernel or copy is put on the GPU for it -- the
    # first one a process launches makes every later call of the engine ~0.5 ms slower)
    stats = torch.tensor([elapsed, float(evals_per_step_local)], dtype=torch.float64,
                         device="cpu" if (not grouped or args.backend == "gloo") else "cuda")
    if grouped:
        worst = stats.clone()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        total = stats.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        elapsed = float(worst[0])
        evals_per_step = float(total[1])
    else:
        evals_per_step = float(evals_per_step_local)

    line = None
    from types import SimpleNamespace
    job = SimpleNamespace(args=args, engine=engine, tables=tables, handles=handles,
                          molecules=molecules, atmos=atmos, grid_args=grid_args, v_lo=v_lo,
                          v_hi=v_hi, dv=dv, n=n, levels_local=levels_local,
                          levels_total=levels_total, make_tables=make_tables, rank=rank,
                          world=world, plan=plan, sharded=sharded, vmr=vmr, workload=None)
    if rank == 0:
        line = headline(job, SimpleNamespace(
            elapsed=elapsed, evals_per_step=evals_per_step,
            evals_per_step_local=evals_per_step_local, kernel_ms=kernel_ms, launches=launches,
            busy_ms=busy_ms,
            per_rank=per_rank, grouped=grouped, everyone=everyone, shared=shared))
        job.workload = line["config"]["workload"]

    # ---- untimed legs, one GPU only -----------------------------------------------------------
    plain = world == 1 and not args.force_group and not args.ablate and not args.host_output
    shared_db = None
    if plain and rank == 0 and args.extras != "none":
        lines_option_legs(job, line, leg)
        if leg("small") and args.config == "target":
            small_grid_legs(job, line)
        if leg("farfield") and not args.farfield:
            farfield_legs(job, line)
        if (leg("ingest") or (not args.no_cpu_baseline)) and args.config == "target":
            # One SQLite file in the reference's schema for the ingest leg and the CPU baseline.
            import tempfile
            from pylbl_amd.database import write_database
            shared_tmp = tempfile.TemporaryDirectory()
            begin = time.perf_counter()
            shared_db = write_database(os.path.join(shared_tmp.name, "lines.db"), tables)
            db_written_s = time.perf_counter() - begin
        if leg("ingest") and shared_db is not None:
            line["ingest"] = ingest_leg(engine, tables, shared_db, atmos, grid_args)
            line["ingest"]["fixture_written_in_s"] = db_written_s
        other_config_legs(job, line, leg)
        api_and_slot_legs(job, line, leg)
    if rank == 0:
        if plain and not args.no_cpu_baseline and args.extras != "none":
            cpu_legs(job, line, shared_db)
        line["environment"] = {
            "variables": {k: v for k, v in sorted(os.environ.items())
                          if k.startswith("PYLBL_AMD_") or k in ("LBL_DEVICE", "LBL_COMPAT_CACHE",
                                                                 "PYLBL_MT_CKD", "PYLBL_FUZZ_CASES")},
            "engine_options_from_environment": dict(engine.environment_options),
            "engine_options_from_command_line": list(args.engine_option),
            "note": "every PYLBL_AMD_* variable in effect: PYLBL_AMD_OPTIONS changes engine "
                    "options for every engine of the process ('ablate' is refused there)"}
        if engine.environment_options:
            line["non_default_engine_options"] = dict(engine.environment_options)
        print(json.dumps(line), file=result_stream, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()
