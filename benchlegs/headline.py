"""Rank 0's full record of the timed region."""
import os

from .common import (BOOST_CLOCK_GHZ, BYTES_PER_EVAL, FLOPS_PER_EVAL, FP64_VECTOR_PEAK_TFLOPS,
                     HBM_PEAK_GBS, SIMDS)
from .launcher import rccl_libraries
from .profiled import PROFILED_RAW, profiled_issue, profiled_traffic


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
    traffic, source = profiled_traffic(workload)
    launch_ms = line["roofline"]["avg_launch_ms"]       # (run alone, where that was measured)
    if traffic is not None:
        raw = PROFILED_RAW.get("accumulate_kernel")
        line["roofline"]["traffic"] = traffic
        line["roofline"]["traffic_source"] = f"profiles/{source}"
        line["roofline"]["traffic_uncorrected"] = raw
        line["roofline"]["traffic_tb_per_s"] = traffic/(launch_ms*1e-3)/1e12
        line["roofline"]["traffic_note"] = (
            "traffic = WRITE_SIZE + 2 x FETCH_SIZE (the gfx950 correction for wide coalesced "
            "reads); this kernel reads its line records through scalar loads, for which the "
            "uncorrected count (traffic_uncorrected = WRITE_SIZE + FETCH_SIZE) may be the truer "
            f"one -- either way {(raw or traffic)/(launch_ms*1e-3)/1e12:.2f}-"
            f"{traffic/(launch_ms*1e-3)/1e12:.2f} TB/s over this launch, "
            f"{traffic/(launch_ms*1e-3)/1e9/HBM_PEAK_GBS:.3f} of the HBM roofline")
    issue = profiled_issue(workload)
    if issue is not None and issue.get("evals_per_launch"):
        # The counters were collected in another process on the same workload: they describe this
        # launch only if both launched the same evaluations.
        if abs(issue["evals_per_launch"] - evals_per_launch) > 1e-3*evals_per_launch:
            line["roofline"]["issue_rejected"] = {
                "source": issue.get("source"), "evals_per_launch": issue["evals_per_launch"],
                "why": "the counter summary's evals per launch differ from this run's"}
            issue = None
    if issue is not None and issue.get("evals_per_launch"):
        per_eval = issue["fp64_wave_instructions_per_launch"]*64./issue["evals_per_launch"]
        # One fp64 wave-instruction occupies a SIMD's issue port for 4 cycles (16 lanes/cycle).
        ceiling = SIMDS*BOOST_CLOCK_GHZ*1e9/4.*64./per_eval
        issue.update({
            "fp64_wave_instructions_per_64_evals": per_eval,
            "issue_ceiling_evals_per_s_at_2.4GHz": ceiling,
            "frac_of_issue_ceiling_at_2.4GHz": evals_per_launch/(launch_ms*1e-3)/ceiling})
        if issue.get("gui_active_cycles_per_xcd"):
            # Busy cycles of the profiled launch (GRBM_GUI_ACTIVE / 8 XCDs): the fraction of
            # a SIMD's 4-cycle issue slots that fp64 instructions occupied at the clock the
            # chip actually ran.
            per_simd = issue["fp64_wave_instructions_per_launch"]/SIMDS*4.
            issue["frac_of_issue_slots_at_measured_clock"] = \
                per_simd/issue["gui_active_cycles_per_xcd"]
        line["roofline"]["issue"] = issue
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

