"""The ONE line bench.py prints: the contract's keys, `roofline`, `cpu_baseline` and one short
record per untimed leg -- numbers only, no prose -- small enough for a driver that keeps the last
few KB of stdout.  Everything else (notes, per-rank device identities, splits, sources) is in the
full record written beside bench.py (`full_record`)."""
import json
import os

from .common import ROOT

LIMIT_BYTES = 8192          # what the driver is known to keep of stdout
TARGET_BYTES = 6000         # what the line is trimmed to (margin for a longer workload string)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
            "launches_timed", "evals_per_launch", "flops_per_eval", "frac_from_step_time",
            "frac_while_running_in_region")
ISSUE = ("evals_per_launch", "fp64_wave_instructions_per_launch",
         "fp64_wave_instructions_per_64_evals", "frac_of_issue_ceiling_at_2.4GHz",
         "frac_of_issue_slots_at_measured_clock", "sclk_ghz_measured", "source")
CPU = ("value", "unit", "cores", "kind", "cpu", "host_cores")


def rounded(x, digits=5):
    """Floats to `digits` significant digits (the full record keeps every bit)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: rounded(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rounded(v, digits) for v in x]
    return x


def pick(record, keys):
    return {k: record[k] for k in keys if k in record}


def leg(record):
    """value / ms_per_step / frac of one leg's record (frac from its roofline where it has one)."""
    if not isinstance(record, dict):
        return None
    out = pick(record, ("value", "ms_per_step"))
    roofline = record.get("roofline")
    if isinstance(roofline, dict) and roofline.get("frac") is not None:
        out["frac"] = roofline["frac"]
    return out


def legs_of(line):
    """One compact record per untimed leg present in the full record."""
    out = {}
    for name, key in (("sustained", "sustained"), ("single_lane", "single_lane_option"),
                      ("pedestal", "pedestal_option"),
                      ("standard_atmosphere_8_levels", "standard_atmosphere_option"),
                      ("banded_table", "banded_table_option"), ("config2", "config2_option"),
                      ("config3_share", "config3_share_option"),
                      ("config4_share", "config4_share_option"),
                      ("continuum", "continuum_slot"), ("xsec", "cross_section_slot")):
        if key in line:
            out[name] = leg(line[key])
    dense = line.get("dense_table_option")
    if dense:
        out["dense_table"] = leg(dense["plain"])
        out["dense_table_pedestal"] = leg(dense["remove_pedestal"])
    small = line.get("small_grid_options", {})
    for name in ("config0", "config1"):
        if name in small:
            out[name] = leg(small[name])
            if "us_per_call" in small[name]:
                out[name]["us_per_call"] = small[name]["us_per_call"]
    far = line.get("farfield_option", {})
    for name, key in (("farfield_plain", "plain"), ("farfield_pedestal", "remove_pedestal")):
        if key in far:
            out[name] = leg(far[key])
    api = line.get("api_call")
    if api:
        out["api_call"] = {f"{fmt}_ms": record["ms_per_call"]
                           for fmt, record in api["formats"].items()}
        out["api_call"]["total_frac_of_link"] = api["formats"]["total"]["roofline"]["frac"]
        out["api_call"]["d2h_gbs"] = api["d2h_pinned_gbs_measured"]
    ingest = line.get("ingest")
    if ingest:
        out["ingest_s"] = {route: record["total_s"] for route, record in ingest["routes"].items()}
        out["ingest_s"]["c_entry"] = sum(m["c_entry_ingest_s"]
                                         for m in ingest["per_molecule"].values())
        out["ingest_s"]["upload"] = sum(ingest["upload_s"].values())
    return out


def compact(line, full_record=None):
    """The short line of a full record."""
    out = pick(line, CONTRACT)
    config = line["config"]
    out["config"] = pick(config, ("workload", "lines", "levels_total", "atmosphere", "output"))
    out["config"]["parallelism"] = config.get("parallelism", "").split(":")[-1].strip()[:80]
    out.update(pick(line, ("evals_per_step", "evals_per_s_per_gpu", "spectra_per_s", "INVALID",
                           "non_default_engine_options", "kernel_ms_per_step")))
    roofline = pick(line["roofline"], ROOFLINE)
    if isinstance(line["roofline"].get("issue"), dict):
        roofline["issue"] = pick(line["roofline"]["issue"], ISSUE)
    out["roofline"] = roofline
    if "cpu_baseline" in line:
        base = line["cpu_baseline"]
        out["cpu_baseline"] = pick(base, CPU)
        out["cpu_baseline"]["sample"] = base.get("sample_short") or base.get("sample", "")[:160]
        if "split" in base:
            out["cpu_baseline"]["voigt_loop_evals_per_s"] = base["split"]["voigt_loop_evals_per_s"]
            out["cpu_baseline"]["read_s_per_molecule"] = base["split"]["read_s_per_molecule"]
    for key in ("cpu_baseline_parallel", "cpu_baseline_all_cores"):
        if key in line:
            out[key] = pick(line[key], ("value", "cores"))
    report = line.get("distributed")
    if report:
        out["distributed"] = pick(report, ("world_size", "backend", "launcher", "distinct_devices",
                                           "bytes_to_rank0_per_step", "exchange_alone_ms_max"))
        out["distributed"]["ordering"] = report["kernels_to_exchange_ordering"].split(" ")[0]
        out["distributed"]["ranks"] = [
            pick(r, ("rank", "device_index", "ms_per_step", "spectra_per_s",
                     "exchange_wait_ms_per_step")) for r in report.get("ranks", [])]
    legs = legs_of(line)
    if legs:
        out["legs"] = legs
    if full_record:
        out["full_record"] = full_record
    # (the contract's own numbers keep every digit; everything else five significant ones)
    out = {k: (v if k in CONTRACT or k in ("evals_per_step",) else rounded(v))
           for k, v in out.items()}
    # A guard, not a plan: should the line still be too long (hundreds of ranks), drop the bulk.
    for key in ("legs", "distributed"):
        if len(json.dumps(out)) <= TARGET_BYTES:
            break
        if key == "distributed" and "distributed" in out:
            out["distributed"].pop("ranks", None)
        else:
            out.pop(key, None)
    return out


def write_full_record(line, path):
    """The whole record as indented JSON at `path` (a temporary file if that cannot be written);
    returns where it went, relative to the repo root where possible."""
    import tempfile
    text = json.dumps(line, indent=1)
    try:
        with open(path, "w") as handle:
            handle.write(text)
    except OSError:
        with tempfile.NamedTemporaryFile("w", suffix=".json", prefix="bench_full_",
                                         delete=False) as handle:
            handle.write(text)
            path = handle.name
    path = os.path.abspath(path)
    return os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
