"""bench.py's contract with the driver, checked on the GPU: ONE short JSON line (<= 8 KB: the
driver keeps only the tail of stdout) with the agreed keys, a roofline fraction in (0, 1], a CPU
baseline, the whole record in the file the line names, and the same from the two-rank launch
(gloo, both ranks on GPU 0: the flow the driver starts with torch.distributed.run over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
        "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


LIMIT = 8192            # bytes of stdout the driver is known to keep


def run(command, env=None, tmp_path=None):
    """Runs a bench command; returns the short line (stdout) and the full record (the file the
    line names)."""
    if tmp_path is not None:
        command = command + ["--full-record", str(tmp_path / "bench_full.json")]
    result = subprocess.run(command, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert result.returncode == 0, result.stderr[-3000:]
    # stdout is the one JSON line and nothing else (library chatter goes to stderr)
    lines = result.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), result.stdout[-2000:]
    assert len(result.stdout.encode()) <= LIMIT, len(result.stdout)
    line = json.loads(lines[0])
    path = line["full_record"]
    with open(path if os.path.isabs(path) else os.path.join(ROOT, path)) as handle:
        full = json.load(handle)
    # the short line is an extract of the full record, not a second measurement
    assert abs(line["value"] - full["value"]) <= 1e-4*full["value"]
    assert line["config"]["workload"] == full["config"]["workload"]
    return line, full


def check(line, n_gpus, steps, warmup):
    assert KEYS <= set(line)
    assert line["n_gpus"] == n_gpus and line["steps"] == steps and line["warmup"] == warmup
    assert line["unit"] == "evals/s" and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] > 1.e9*n_gpus               # north_star's floor, per GPU
    assert abs(line["value"] - line["evals_per_step"]*steps/(line["ms_per_step"]*1e-3*steps)) \
        <= 1e-3*line["value"]               # (the short line carries five significant digits)
    roofline = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roofline)
    assert 0. < roofline["frac"] <= 1.
    assert abs(roofline["frac"] - roofline["achieved"]/roofline["peak"]) < 1e-4
    assert not any(isinstance(v, str) and len(v) > 40 for v in roofline.values())   # no prose


def check_issue(roofline):
    """roofline.issue (from the committed PMC pass of this workload) describes THIS launch: the
    same evaluations per launch, and a fraction of the issue ceiling that is a fraction."""
    issue = roofline.get("issue")
    assert issue is not None, roofline
    assert abs(issue["evals_per_launch"] - roofline["evals_per_launch"]) \
        <= 1e-3*roofline["evals_per_launch"]
    assert 0. < issue["frac_of_issue_ceiling_at_2.4GHz"] <= 1.
    assert 0. < issue["frac_of_issue_slots_at_measured_clock"] <= 1.
    assert 3. < issue["fp64_wave_instructions_per_64_evals"] < 8.


def test_single_gpu_line(tmp_path):
    line, full = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--extras",
                      "pedestal", "--cpu-sample-cm", "200"], tmp_path=tmp_path)
    check(line, 1, 3, 1)
    assert line["config"]["workload"].startswith("BASELINE config 'target'")
    baseline = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(baseline)
    assert baseline["cores"] == 1 and baseline["kind"] in ("reference", "port")
    assert line["legs"]["pedestal"]["value"] > 1.e9
    assert full["pedestal_option"]["value"] > 1.e9
    check_issue(line["roofline"])


def test_the_drivers_own_command_prints_a_line_it_can_keep():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` -- what the driver runs at round end,
    every leg on (only the CPU legs' sample is cut short here): the last line of stdout is the
    whole of stdout, fits the driver's tail, and carries roofline + cpu_baseline + every leg."""
    line, full = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5",
                      "--cpu-sample-cm", "100"])
    check(line, 1, 20, 5)
    check_issue(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample", "cpu", "host_cores"} <= set(line["cpu_baseline"])
    assert {"value", "cores"} <= set(line["cpu_baseline_parallel"])
    legs = line["legs"]
    assert {"pedestal", "config0", "config1", "config2", "config3_share", "config4_share",
            "farfield_plain", "farfield_pedestal", "api_call", "continuum", "xsec",
            "ingest_s"} <= set(legs)
    for name in ("pedestal", "config2", "config3_share", "config4_share", "farfield_pedestal"):
        assert legs[name]["value"] > 1.e9 and legs[name]["ms_per_step"] > 0.
    for name in ("config2", "config3_share", "config4_share", "continuum", "xsec"):
        assert 0. < legs[name]["frac"] <= 1.
    assert full["roofline"]["note"] and full["environment"] is not None


@pytest.mark.parametrize("launcher,ordering", [("torch.distributed.run", "host"),
                                               ("bench.py", "host"), ("bench.py", "device")])
def test_two_rank_line(launcher, ordering, tmp_path):
    """Under the launcher the driver's contract names, and bare (`python bench.py --gpus 2`):
    bench.py then starts its two ranks itself, as children, before anything touches the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    arguments = ["bench.py", "--gpus", "2", "--backend", "gloo", "--config", "1", "--steps", "2",
                 "--warmup", "1"]
    # ordering "device": kernels and exchange ordered by events on the GPU as on the RCCL path
    # (PYLBL_AMD_ORDER_ON_DEVICE, distributed.ShardedLines), gloo carrying the blocks.
    environment = dict(os.environ, PYLBL_AMD_ORDER_ON_DEVICE="1" if ordering == "device" else "0")
    if launcher == "bench.py":
        for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            environment.pop(name, None)
        line, full = run([sys.executable] + arguments, env=environment, tmp_path=tmp_path)
    else:
        line, full = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(port)] + arguments, env=environment, tmp_path=tmp_path)
    check(line, 2, 2, 1)
    assert line["distributed"]["launcher"] == launcher
    assert "cpu_baseline" not in line           # rank 0 at N = 1 only
    assert line["config"]["levels_total"] == 2
    # per rank on the short line: its own spectra/s and what it waited for the exchange
    assert [r["rank"] for r in line["distributed"]["ranks"]] == [0, 1]
    for r in line["distributed"]["ranks"]:
        assert r["spectra_per_s"] > 0. and r["exchange_wait_ms_per_step"] >= 0.
    line = full
    # What makes the first RCCL run on the driver's node diagnosable: who ran where, each rank's
    # own time, what the collection moved and what it cost.
    report = line["distributed"]
    assert report["world_size"] == 2 and report["backend"] == "gloo"
    assert report["kernels_to_exchange_ordering"].startswith(ordering)
    assert [r["rank"] for r in report["ranks"]] == [0, 1]
    assert report["distinct_devices"] == 1 and report["ranks_sharing_a_device"]   # both on GPU 0
    n = 500000
    for r in report["ranks"]:
        assert {"device_index", "name", "ms_per_step", "evals_per_step", "bytes_sent_per_step",
                "bytes_received_per_step", "exchange_wait_ms_per_step",
                "unoverlapped_exchange_ms"} <= set(r)
        assert r["ms_per_step"] > 0. and r["exchanges"] >= 2
    assert report["ranks"][1]["bytes_sent_per_step"] == 2*n*8          # H2O + CO2 of its level
    assert report["ranks"][0]["bytes_received_per_step"] == 2*n*8
    assert report["bytes_to_rank0_per_step"] == 2*n*8


def test_one_rank_line_with_the_process_group_over_rccl(tmp_path):
    """`bench.py --gpus 1 --backend nccl --force-group`: the N-GPU flow with N = 1 -- RCCL
    loaded, a communicator on the device, the step's blocks collected through the grouped
    send/recv, the barrier and the reductions of the record over it -- so that the first lease of
    a whole node does not die on library load, HSA_ENABLE_IPC_MODE_LEGACY or stream semantics."""
    environment = dict(os.environ)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        environment.pop(name, None)
    line, full = run([sys.executable, "bench.py", "--gpus", "1", "--backend", "nccl",
                      "--force-group", "--steps", "3", "--warmup", "1"], env=environment,
                     tmp_path=tmp_path)
    check(line, 1, 3, 1)
    assert line["distributed"]["ranks"][0]["spectra_per_s"] > 0.
    line = full
    report = line["distributed"]
    assert report["world_size"] == 1 and report["backend"] == "nccl"
    assert report["kernels_to_exchange_ordering"].startswith("device")
    assert report["rccl_mapped"], report
    n = 5000000
    rank0 = report["ranks"][0]
    assert rank0["exchanges"] >= 3
    assert rank0["bytes_sent_per_step"] == rank0["bytes_received_per_step"] == 2*n*8
    assert "cpu_baseline" not in line and "farfield_option" not in line
