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

from benchlegs.common import CONFIGS, atmosphere_for                     # noqa: E402
from benchlegs.compact import compact, write_full_record                                # noqa: E402
from benchlegs.cpu import closed_form_evals, cpu_legs                    # noqa: E402,F401
from benchlegs.headline import headline                                  # noqa: E402
from benchlegs.launcher import device_identity, launch_ranks            # noqa: E402,F401
from benchlegs.lines import (farfield_legs, lines_option_legs, other_config_legs,     # noqa: E402
                             small_grid_legs)
from benchlegs.ingest import ingest_leg                                  # noqa: E402
from benchlegs.profiled import profiled_issue, profiled_traffic          # noqa: E402,F401
from benchlegs.slots import api_and_slot_legs                            # noqa: E402


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
    parser.add_argument("--full-record", default=os.path.join(ROOT, "bench_full.json"),
                        help="where the whole record goes (notes, per-rank identities, splits); "
                             "stdout carries only the short line")
    parser.add_argument("--force-group", action="store_true",
                        help="initialise the torch.distributed process group also at N = 1 (with "
                             "--backend nccl: RCCL loads and a communicator is made on one GPU) and "
                             "run the step through the collection path of ShardedLines")
    return parser.parse_args()


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
                    spectra_per_s=len(plan.levels_of(rank))*args.steps/elapsed,
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
            "engine_options_from_command_line": list(args.engine_option)}
        # Any engine option that differs from the shipped defaults marks the record, whichever
        # way it came in (PYLBL_AMD_OPTIONS, --engine-option, --points-per-lane).
        changed = dict(engine.environment_options)
        changed.update(pair.split("=") for pair in args.engine_option)
        if args.points_per_lane:
            changed["points_per_lane"] = args.points_per_lane
        if changed:
            line["non_default_engine_options"] = {k: int(v) for k, v in changed.items()}
        # stdout: ONE short line (benchlegs/compact.py); the whole record goes beside bench.py.
        full_path = write_full_record(line, args.full_record)
        print(json.dumps(compact(line, full_record=full_path)), file=result_stream, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()

