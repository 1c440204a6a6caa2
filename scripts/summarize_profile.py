"""Condenses gpurun_out/prof_<tag>/ (scripts/profile_bench.sh) into profiles/<tag>_*:
the rocprofv3 kernel stats table, per-kernel HBM counter means, and the bench line that was
running under the profiler."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
stats = glob.glob(f"{src}/stats/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 "
           "--warmup 2 --no-cpu-baseline --extras " + os.environ.get("EXTRAS", "none") + " " +
           " ".join(sys.argv[2:]), "kernels": {}, "counters": {}}
for row in csv.DictReader(open(stats)):
    summary["kernels"][row["Name"].split("(")[0]] = {
        "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"])/1e6,
        "min_ms": float(row["MinNs"])/1e6, "max_ms": float(row["MaxNs"])/1e6,
        "percent": float(row["Percentage"])}
# Asynchronous calls take turns on two lanes, so most accumulate launches of the timed region run
# beside the tail or the head of another one; bench.py's `roofline` divides by launches run alone
# (blocking calls after the region).  The trace tells the two kinds apart: mean duration of the
# accumulate launches that overlap no other accumulate launch in time, and of those that do.
traces = glob.glob(f"{src}/stats/*/*_kernel_trace.csv")
if traces:
    spans = collections.defaultdict(list)
    for row in csv.DictReader(open(traces[0])):
        name = row["Kernel_Name"].split("(")[0]
        if "accumulate_kernel" in name:
            spans[name].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    everything = sorted(s for v in spans.values() for s in v)
    for name, mine in spans.items():
        alone, beside = [], []
        for b, e in mine:
            others = sum(1 for (b2, e2) in everything if b2 < e and b < e2) - 1
            (beside if others > 0 else alone).append((e - b)/1e6)
        summary["kernels"].setdefault(name, {}).update({
            "launches_alone": len(alone),
            "avg_ms_alone": sum(alone)/len(alone) if alone else None,
            "launches_beside_another": len(beside),
            "avg_ms_beside_another": sum(beside)/len(beside) if beside else None})
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(f"{src}/{kind}/*/*_counter_collection.csv")
    if not files:
        continue
    values = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            values[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for name, v in values.items():
        # rocprofv3 reports these two counters in KiB.
        summary["counters"].setdefault(name, {})[counter + "_KiB_per_launch_mean"] = sum(v)/len(v)
        summary["counters"][name][counter + "_launches"] = len(v)
for name, c in summary["counters"].items():
    if "FETCH_SIZE_KiB_per_launch_mean" in c and "WRITE_SIZE_KiB_per_launch_mean" in c:
        fetch = c["FETCH_SIZE_KiB_per_launch_mean"]*1024
        write = c["WRITE_SIZE_KiB_per_launch_mean"]*1024
        # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so
        # reads are doubled before comparing with a byte count; WRITE_SIZE is exact.
        c["hbm_bytes_per_launch"] = 2*fetch + write
        c["hbm_bytes_per_launch_uncorrected"] = fetch + write
for name, c in summary["counters"].items():
    if "hbm_bytes_per_launch" in c and name in summary["kernels"]:
        # Achieved HBM rate of the kernel against the 8 TB/s roofline (MI355X_MICROARCH.md).
        c["hbm_gb_per_s"] = c["hbm_bytes_per_launch"]/(summary["kernels"][name]["avg_ms"]*1e-3)/1e9
        c["hbm_fraction_of_8TBps"] = c["hbm_gb_per_s"]/8000.
try:
    summary["bench_line"] = json.loads(open(f"{src}/bench_stats.json").read().strip().splitlines()[-1])
except Exception as error:  # noqa
    summary["bench_line"] = str(error)
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
