"""CPU baselines of bench.py: the only place the benchmark touches oracle/ (as the thing timed
beside the GPU path, never as the product)."""
import os
import time

import numpy as np

from .common import cpu_model, cpu_quota


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

