"""Prints the results table of DESIGN.md section 8 from profiles/bench_<tag>*_full.json (the whole
records scripts/bench_round.sh writes beside the short lines).  Build container:
python scripts/design_numbers.py r06d"""
import json
import sys

tag = sys.argv[1]


def read(name=""):
    with open(f"profiles/bench_{tag}{'_' + name if name else ''}_full.json") as handle:
        return json.load(handle)


d = read()
f = d["farfield_option"]
rows = [
    ("**default (target): 1 level, H2O+CO2, 5 M points, `remove_pedestal=False`**", d["value"],
     d["ms_per_step"], d["spectra_per_s"],
     "frac **%.2f** (%.2f ms per launch)" % (d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])),
    ("same for ≥ 2 s (`sustained`) / calls back to back on one lane (`single_lane_option`)",
     d["sustained"]["value"], d["sustained"]["ms_per_step"], d["sustained"]["spectra_per_s"],
     "%.3g on one lane; frac from the step time %.2f" % (
         d["single_lane_option"]["value"], d["roofline"].get("frac_from_step_time", float("nan")))),
    ("same, `remove_pedestal=True` — what `compute_absorption` defaults to (`pedestal_option`)",
     d["pedestal_option"]["value"], d["pedestal_option"]["ms_per_step"],
     d["pedestal_option"]["spectra_per_s"], ""),
    ("8 standard-atmosphere levels in one call, pedestal", d["standard_atmosphere_option"]["value"],
     d["standard_atmosphere_option"]["ms_per_step"], d["standard_atmosphere_option"]["spectra_per_s"], ""),
    ("8 Gaussian bands per molecule, pedestal; one molecule of 1.6 M lines plain / pedestal",
     d["banded_table_option"]["value"], d["banded_table_option"]["ms_per_step"],
     d["banded_table_option"]["spectra_per_s"],
     "dense: %.1f / %.1f ms" % (d["dense_table_option"]["plain"]["ms_per_step"],
                                d["dense_table_option"]["remove_pedestal"]["ms_per_step"])),
    ("**far-field series with the pedestal (the `Spectroscopy` default)**",
     f["remove_pedestal"]["value"], f["remove_pedestal"]["ms_per_step"],
     f["remove_pedestal"]["spectra_per_s"],
     "plain %.3f ms (%.3g); round 5: 0.705 plain / 0.778 with the pedestal; fp64 issue slots "
     "filled %.2f / %.2f" % (f["plain"]["ms_per_step"], f["plain"]["value"],
                             f["plain"]["roofline"]["frac"] or 0., f["remove_pedestal"]["roofline"]["frac"] or 0.)),
    ("configs[0]: CO2, 500–800 @ 0.1 (4 calls in flight)", d["small_grid_options"]["config0"]["value"],
     d["small_grid_options"]["config0"]["ms_per_step"], d["small_grid_options"]["config0"]["spectra_per_s"],
     "timed line %.3g, frac %.3f (12 workgroups: launch-bound); %.1f µs per call in the ring" % (
         read("config0")["value"], read("config0")["roofline"]["frac"],
         d["small_grid_options"]["config0"]["us_per_call"])),
    ("configs[1]: H2O+CO2, 1–5000 @ 0.01 (4 calls in flight)", d["small_grid_options"]["config1"]["value"],
     d["small_grid_options"]["config1"]["ms_per_step"], d["small_grid_options"]["config1"]["spectra_per_s"],
     "timed line %.3g, frac %.2f" % (read("config1")["value"], read("config1")["roofline"]["frac"])),
    ("configs[2]: 8 molecules, 5 M points (`config2_option`; `--config 2`)", d["config2_option"]["value"],
     d["config2_option"]["ms_per_step"], d["config2_option"]["spectra_per_s"],
     "frac %.2f; `--config 2` line %.3g, frac %.2f" % (d["config2_option"]["roofline"]["frac"],
                                                     read("config2")["value"], read("config2")["roofline"]["frac"])),
    ("configs[3], rank 7's 8 of 64 levels (0.3–0.1 hPa: the slowest share), pedestal, through "
     "`for_engine` (`config3_share_option`)", d["config3_share_option"]["value"],
     d["config3_share_option"]["ms_per_step"], d["config3_share_option"]["spectra_per_s"],
     "frac %.2f; rank 0's share %.3g, frac %.2f" % (
         d["config3_share_option"]["roofline"]["frac"], d["config3_share_option"]["rank0_share"]["value"],
         d["config3_share_option"]["rank0_share"]["roofline"]["frac"])),
    ("configs[4], 4 of rank 3's 32 levels × 8 molecules × 10 M points, pedestal, `\"total\"` "
     "(`config4_share_option`)", d["config4_share_option"]["value"], d["config4_share_option"]["ms_per_step"],
     d["config4_share_option"]["spectra_per_s"],
     "frac %.2f; the whole share of 32 levels: %.3g (%.0f ms), per gas %.3g" % (
         d["config4_share_option"]["roofline"]["frac"], read("config4_share32_total")["value"],
         read("config4_share32_total")["ms_per_step"], read("config4_share32")["value"])),
]
out = ["| workload | evals/s | ms/step | spectra/s | notes |", "|---|---|---|---|---|"]
for name, value, ms, spectra, note in rows:
    out.append("| %s | %.3g | %.4g | %.4g | %s |" % (name, value, ms, spectra, note))
c, x = d["continuum_slot"], d["cross_section_slot"]
out.append("| slot 1: H2O foreign + self + CO2 continua in one pass, 5 M points | %.3g cont×pts/s | %.3f | "
           "%.0f | interp %.1f µs: %.2f of HBM for 16 algorithmic B/pt; it moves %.1f MB (the block, once) "
           "and is issue-bound; one launch per continuum: %.3f ms |" % (
               c["value"], c["ms_per_step"], c["spectra_per_s"], c["kernel_ms_per_step"]["interpolate"]*1e3,
               c["roofline"]["frac"], (c["roofline"].get("traffic") or 0)/1e6,
               c["one_launch_per_continuum"]["ms_per_step"]))
out.append("| slot 2: one two-band cross-section molecule | %.3g pts/s | %.3f | %.0f | interp %.1f µs: "
           "%.2f of HBM (searches in LDS, one wavefront per band window) |" % (
               x["value"], x["ms_per_step"], x["spectra_per_s"], x["kernel_ms_per_step"]["interpolate"]*1e3,
               x["roofline"]["frac"]))
a = d["api_call"]["formats"]
out.append("| `Spectroscopy.compute_absorption()`, host arrays: `\"total\"` / `\"gas\"` / `\"all\"` | — | "
           "**%.2f / %.2f / %.2f** | %.0f / %.0f / %.0f | round 5: 1.48 / 2.07 / 3.69; link fractions %.2f / "
           "%.2f / %.2f of %.0f GB/s |" % (
               a["total"]["ms_per_call"], a["gas"]["ms_per_call"], a["all"]["ms_per_call"],
               a["total"]["spectra_per_s"], a["gas"]["spectra_per_s"], a["all"]["spectra_per_s"],
               a["total"]["roofline"]["frac"], a["gas"]["roofline"]["frac"], a["all"]["roofline"]["frac"],
               d["api_call"]["d2h_pinned_gbs_measured"]))
g2, gb = read("gloo2"), read("gloo2_bare")
out.append("| 2 ranks over gloo on one GPU, launcher / bare with device-side ordering | %.3g / %.3g | "
           "%.1f / %.1f | — | rehearsal of the N > 1 record, not a scaling figure |" % (
               g2["value"], gb["value"], g2["ms_per_step"], gb["ms_per_step"]))
ho = read("host_output")
out.append("| spectra copied to page-locked host memory inside the step (`--host-output`: PCIe-inclusive, "
           "never `value`) | %.3g | %.3g | %.0f | |" % (ho["value"], ho["ms_per_step"], ho["spectra_per_s"]))
cb = d["cpu_baseline"]
sp = cb["split"]
out.append("| CPU: reference C, 1 thread (%s), whole grid | %.3g | %.1f s | — | per-call read + line scalars "
           "%.2f s (%.1f %%), Voigt loop %.1f s |" % (cb["cpu"], cb["value"], sp["total_s"],
                                                     sp["read_and_line_scalars_s"], 100*sp["read_fraction"],
                                                     sp["voigt_loop_s"]))
cp, ca = d["cpu_baseline_parallel"], d["cpu_baseline_all_cores"]
if ca["cores"] == cp["cores"]:
    out.append("| CPU: C restatement, %d processes = every core the job may use (`cpu_baseline_parallel` = "
               "`cpu_baseline_all_cores`) | %.3g | — | — | affinity mask %s hardware threads, cgroup CPU quota "
               "%s cores; forced to 256 processes (`--cpu-all-cores 256`, `profiles/bench_r05b.json`): 8.3e9 — "
               "they share the 16 cores' time |" % (cp["cores"], cp["value"], cp.get("usable_hardware_threads"),
                                                   cp.get("cgroup_cpu_quota_cores")))
else:
    out.append("| CPU: C restatement, 16 processes / %d processes | %.3g / %.3g | — | — | cgroup CPU quota %s "
               "cores (pool start %.0f s, untimed) |" % (ca["cores"], cp["value"], ca["value"] or 0.,
                                                         ca.get("cgroup_cpu_quota_cores"), ca.get("pool_start_s", 0.)))
ing = d["ingest"]
out.append("| ingest per molecule (H2O 110 k / CO2 400 k lines): SQLite → arrays (any of the three routes) / "
           "upload / C entry's first call − second | — | %.2f / %.2f s; %.1f / %.1f ms; %.0f / %.0f ms | — | "
           "once per molecule; the reference pays %.2f s per call for the pair |" % (
               ing["routes"]["path"]["seconds"]["H2O"], ing["routes"]["path"]["seconds"]["CO2"],
               ing["upload_s"]["H2O"]*1e3, ing["upload_s"]["CO2"]*1e3,
               ing["per_molecule"]["H2O"]["c_entry_ingest_s"]*1e3,
               ing["per_molecule"]["CO2"]["c_entry_ingest_s"]*1e3, sp["read_and_line_scalars_s"]))
print("\n".join(out))
