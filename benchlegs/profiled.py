"""Look-ups in profiles/: the PMC counters cannot be read from inside bench.py, so the record
quotes the newest committed counter summary of the same workload."""
import json
import os

from .common import BOOST_CLOCK_GHZ, ROOT, SIMDS


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
        # (GRBM_GUI_ACTIVE over the launch's own duration; for a short launch that ran beside
        # another the quotient can come out above what the chip can clock: never more than boost)
        clock = min(issue["sclk_ghz_measured"], BOOST_CLOCK_GHZ)
        out["sclk_ghz_measured"] = clock
        out["frac_of_issue_slots_at_measured_clock"] = rate/(SIMDS*clock*1e9/4.)
    return out

