"""Condenses gpurun_out/pmc_<tag>/ (scripts/profile_counters.sh) into
profiles/<tag>_valu_counters.json: mean counter values per launch for every kernel, the
workload that ran and the closed-form evaluations per accumulate launch (what bench.py's
`roofline.issue` reads)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
src = f"gpurun_out/pmc_{tag}"
values = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(f"{src}/pass*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        values[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
line = json.loads(open(f"{src}/bench1.json").read().strip().splitlines()[-1])
# Evaluations per accumulate launch: what the profiled run itself reports (roofline.evals_per_launch
# = this rank's evals per step / accumulate launches per step INSIDE the timed region; never
# roofline.launches_timed, which counts the launches re-run alone after the region).
evals_per_launch = line["roofline"]["evals_per_launch"]
summary = {
    "tag": tag,
    "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --extras $EXTRAS (one pass per group, scripts/profile_counters.sh)",
    "workload": line["config"]["workload"],
    "evals_per_accumulate_launch": evals_per_launch,
    "kernels": {},
}
for name, counters in values.items():
    summary["kernels"][name] = {
        "mean_per_launch": {k: sum(v)/len(v) for k, v in counters.items()},
        "launches_sampled": {k: len(v) for k, v in counters.items()},
    }
# Shader clock under this load: GRBM_GUI_ACTIVE (busy cycles, reported as the sum over the 8 XCDs)
# over the kernel's duration in the same pass.
XCDS = 8
clock = collections.defaultdict(list)
for path in sorted(glob.glob(f"{src}/pass*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if ns > 0:
                clock[row["Kernel_Name"].split("(")[0]].append(
                    float(row["Counter_Value"])/XCDS/ns)
for name, ghz in clock.items():
    summary["kernels"][name]["sclk_ghz_from_gui_active"] = sum(ghz)/len(ghz)
    summary["kernels"][name]["gui_active_cycles_per_xcd"] = \
        summary["kernels"][name]["mean_per_launch"]["GRBM_GUI_ACTIVE"]/XCDS
json.dump(summary, open(f"profiles/{tag}_valu_counters.json", "w"), indent=1)
for name, entry in summary["kernels"].items():
    if "accumulate_kernel" in name:
        c = entry["mean_per_launch"]
        fp64 = sum(c.get(x, 0.) for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
                                          "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
        per = fp64*64./summary["evals_per_accumulate_launch"]
        print(f"{name}: {per:.3f} fp64 wave-instructions per 64 evals; shader clock "
              f"{entry.get('sclk_ghz_from_gui_active', 0.):.3f} GHz; issue slots used "
              f"{fp64/1024*4/entry.get('gui_active_cycles_per_xcd', 1.):.3f}")
