"""Device legs of the lines path: the timed step again under the conditions users meet, the other
BASELINE configs at one GPU's size, and the roofline of launches run alone."""
import time

import numpy as np

from .common import CONFIGS, EIGHT, FLOPS_PER_EVAL, FP64_VECTOR_PEAK_TFLOPS, HBM_PEAK_GBS
from .profiled import issue_slot_fraction, profiled_issue, profiled_traffic


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
    """BASELINE configs[0] and [1] as rings of asynchronous calls."""
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
        small[f"config{name}"]["us_per_call"] = \
            small[f"config{name}"]["ms_per_step"]*1e3/len(small_handles)
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
        for kernel in ("farfield_series_kernel",):
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

