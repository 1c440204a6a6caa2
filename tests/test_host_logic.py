"""CPU-only checks of the host side: database read side and fixtures, grid handling, the
registry contract, and that the C-ABI library loads and exports every declared symbol."""
import re
from pathlib import Path

import numpy as np
import pytest

from pylbl_amd import database, synthetic
from pylbl_amd.errors import AliasNotFoundError, TipsDataNotFoundError

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    """Every function declared in include/lbl_amd.h is exported by liblbl_amd.so (built by
    __graft_entry__.build()); no compute call is made."""
    import __graft_entry__
    __graft_entry__.build()
    from pylbl_amd import engine
    header = (ROOT / "include" / "lbl_amd.h").read_text()
    declared = set(re.findall(r"\b(lbl_[a-z_]+)\s*\(", header))
    # The reference's own symbol (absorption.c:19-30) is part of the contract too.
    assert re.search(r"^int absorption\(double pressure,", header, re.M)
    declared.add("absorption")
    assert declared == set(engine.EXPORTED_SYMBOLS)
    lib = engine.library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.lbl_version()


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pylbl_amd import engine
    from pylbl_amd.errors import EngineError
    with pytest.raises(EngineError, match="no CPU fallback"):
        engine.Engine(0)


def test_database_round_trip(tmp_path):
    """write_database uses the reference's DDL; the read side returns the rows in order
    (pyLBL/database.py:350-395, absorption.c:69-70)."""
    tables = [synthetic.line_table("H2O", 1., 200., num_lines=300, seed=1, tips_range=(150, 350)),
              synthetic.line_table("CO2", 1., 200., num_lines=200, seed=2, tips_range=(150, 350))]
    path = database.write_database(tmp_path / "lines.db", tables, aliases={"H2O": ["water"]})
    db = database.Database(path)
    assert db.path == path
    assert db.molecules() == ["H2O", "CO2"]
    for name, table in (("H2O", tables[0]), ("water", tables[0]), ("CO2", tables[1])):
        got = db.line_table(name)
        for column in database.LINE_COLUMNS:
            assert np.array_equal(getattr(got, column), getattr(table, column))
        assert np.array_equal(got.local_iso_id, table.local_iso_id)
        assert np.array_equal(got.mass, table.mass)
        assert np.array_equal(got.tips_data, table.tips_data)
        assert np.array_equal(got.tips_temperature, table.tips_temperature)
    formula, mass, transitions, tips = db.gas("CO2")
    assert formula == "CO2" and mass == list(tables[1].mass)
    assert transitions[5].nu == tables[1].nu[5] and transitions.sw.shape == (200,)
    temperature, data = db.tips("H2O")
    assert data.shape == (4, temperature.size)
    # pyLBL/tips.py:26-39 twin.
    q = tips.total_partition_function(279.54, 1)
    j = int(279.54) - 150
    expect = data_co2 = db.tips("CO2")[1][0]
    assert q == pytest.approx(expect[j] + (expect[j+1] - expect[j])*0.54, rel=1e-12)
    with pytest.raises(AliasNotFoundError):
        db.line_table("XYZ")


def test_database_without_tips_raises(tmp_path):
    table = synthetic.line_table("N2O", 1., 50., num_lines=10, tips_range=(150, 350))
    path = database.write_database(tmp_path / "no_tips.db", [table], with_tips=set())
    with pytest.raises(TipsDataNotFoundError):
        database.Database(path).line_table("N2O")


def test_c_reader_and_sqlite3_reader_return_the_same_table(tmp_path):
    """Database.line_table goes through the engine library's C reader (lbl_table_read: the
    reference C reader's SELECTs, absorption.c:69-70, spectral_database.c:55,113,143); the
    standard-library route is what is left when the library cannot be loaded.  Same arrays, same
    types, same exceptions -- on a file this package wrote and on the file the reference's ORM
    wrote (tests/golden/refdb.db)."""
    from pylbl_amd import engine
    from pylbl_amd.errors import IsotopologuesNotFoundError, TransitionsNotFoundError
    assert engine.read_line_table(str(tmp_path / "missing.db"), "CO2")[0] == engine.TABLE_OPEN_FAILED
    tables = [synthetic.line_table("H2O", 1., 200., num_lines=300, seed=1, tips_range=(150, 350)),
              synthetic.line_table("O3", 1., 200., num_lines=50, seed=3, tips_range=(150, 350))]
    tables[1].local_iso_id[::7] = 0                 # HITRAN's tenth isotopologue, stored raw
    own = database.write_database(tmp_path / "lines.db", tables, aliases={"H2O": ["water"]})
    files = [(own, ["H2O", "water", "O3"])]
    reference_file = ROOT / "tests" / "golden" / "refdb.db"
    if reference_file.exists():
        files.append((str(reference_file), database.Database(str(reference_file)).molecules()))
    for path, names in files:
        db = database.Database(path)
        for name in names:
            try:
                slow = db._line_table_sqlite3(name)
            except BaseException as error:          # (the reference's classes are BaseExceptions)
                with pytest.raises(type(error)):
                    db.line_table(name)
                continue
            fast = db.line_table(name)
            assert fast.formula == slow.formula and fast.molecule_id == slow.molecule_id
            for column in database.LINE_COLUMNS + ("local_iso_id", "isoid", "mass",
                                                   "tips_temperature", "tips_data"):
                a, b = getattr(fast, column), getattr(slow, column)
                assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), column
        for route in (db.line_table, db._line_table_sqlite3):
            with pytest.raises(AliasNotFoundError):
                route("no such molecule")
    # What is missing is reported in the same order by both routes.
    import sqlite3
    for drop, error in (("isotopologue", IsotopologuesNotFoundError),
                        ("transition", TransitionsNotFoundError), ("tips", TipsDataNotFoundError)):
        path = database.write_database(tmp_path / f"without_{drop}.db", tables[:1])
        connection = sqlite3.connect(path)
        connection.execute(f"delete from {drop}")
        connection.commit()
        connection.close()
        for route in (database.Database(path).line_table,
                      database.Database(path)._line_table_sqlite3):
            with pytest.raises(error):
                route("H2O")


def test_mass_slots_follow_hitran_counting():
    """isoid 0 is the tenth isotopologue (spectral_database.c:119-123)."""
    table = synthetic.line_table("O3", 1., 50., num_lines=10)
    table.isoid = np.asarray([1, 2, 0, 4])
    slots = table.mass_by_slot()
    assert slots[0] == table.mass[0] and slots[9] == table.mass[2] and slots[2] == 0.


def test_grid_arguments_match_reference_rounding():
    """pyLBL/c_lib/gas_optics.py:61-63 on the reference's own test grids
    (tests/conftest.py:43-50)."""
    assert synthetic.grid_arguments(np.arange(1., 3250., 0.1)) == (1, 3251, 10)
    assert synthetic.grid_arguments(np.arange(1., 3000., 1.)) == (1, 3000, 1)
    assert synthetic.grid_arguments(np.arange(500., 800., 0.1)) == (500, 801, 10)
    assert synthetic.grid_arguments(np.arange(1., 5000., 0.001)) == (1, 5001, 1000)


def test_registry_contract():
    """Same shape and failure mode as pyLBL/plugins.py / spectroscopy.py:118
    (reference test: tests/test_spectroscopy.py:28-32)."""
    from pylbl_amd import Gas, Spectroscopy, molecular_lines
    assert molecular_lines["mi355x"] is Gas
    atmos = synthetic.fixture_atmosphere()
    grid = np.arange(1., 10., 1.)
    with pytest.raises(KeyError):
        Spectroscopy(atmos, grid, None, lines_backend="not-a-backend")
    spec = Spectroscopy(atmos, grid, None)
    assert spec.output.dim_sizes == [4, 3, 9]
    target = {}
    from pylbl_amd import register
    assert register("mi355x", into=target)["mi355x"] is Gas


def test_number_density():
    from pylbl_amd import number_density
    assert number_density(288.99, 98388., 1.) == pytest.approx(98388./(1.38064852e-23*288.99))


def test_synthetic_tables_are_deterministic_and_sorted():
    a = synthetic.line_table("CO2", 1., 100., num_lines=500)
    b = synthetic.line_table("CO2", 1., 100., num_lines=500)
    assert np.array_equal(a.nu, b.nu) and np.all(np.diff(a.nu) >= 0)
    assert a.tips_data.dtype == np.float64
    assert np.array_equal(a.tips_data, a.tips_data.astype(np.float32).astype(np.float64))
    atmos = synthetic.standard_atmosphere(64)
    assert atmos.p[0] == pytest.approx(101325.) and atmos.p[-1] == pytest.approx(10.)
    assert np.all(atmos.t >= 180.) and set(atmos.vmr) == set(synthetic.MOLECULE_IDS)


def test_oracle_under_sanitizers(tmp_path):
    """The C restatement is clean under ASan/UBSan on a clipped, pedestal-on case
    (sanitizers on the CPU build only)."""
    import subprocess, sys, textwrap
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], check=True,
                   stdout=subprocess.DEVNULL)
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        from oracle import oracle
        oracle.PORT_LIB = oracle.HERE / "liblbl_oracle_asan.so"
        from tests import golden_io
        import numpy as np
        table, cases = golden_io.load_group("clipping")
        for case in cases:
            k, _ = oracle.absorption_port(table, case.temperature, case.pressure, case.vmr,
                                          case.v0, case.vn, case.n_per_v, cut_off=case.cut_off,
                                          remove_pedestal=case.remove_pedestal)
            assert np.array_equal(k, case.k)
        print("clean")
    """)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True,
                             text=True).stdout.strip()
    result = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                            env={"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0",
                                 "PATH": "/usr/bin:/bin"})
    assert "clean" in result.stdout, result.stderr[-2000:]
    assert "runtime error" not in result.stderr and "AddressSanitizer" not in result.stderr


class _Variable(object):
    """The slice of an xarray DataArray that Atmosphere touches (xarray is not installed)."""
    def __init__(self, data, standard_name):
        self.data = np.asarray(data)
        self.attrs = {"units": "1", "standard_name": standard_name}
        self.dims = ("layer",)


class _Dataset(object):
    def __init__(self, data_vars):
        self.data_vars = data_vars

    def __getitem__(self, name):
        return self.data_vars[name]


def _dataset():
    atmos = synthetic.fixture_atmosphere()
    names = {"H2O": "water_vapor", "CO2": "carbon_dioxide", "O3": "ozone", "N2O": "nitrous_oxide",
             "CO": "carbon_monoxide", "CH4": "methane", "O2": "oxygen", "N2": "nitrogen"}
    data_vars = {"pressure": _Variable(atmos.p, "air_pressure"),
                 "temperature": _Variable(atmos.t, "air_temperature")}
    for formula, name in names.items():
        data_vars[name] = _Variable(atmos.vmr[formula], f"mole_fraction_of_{name}_in_air")
    return atmos, names, data_vars


def test_atmosphere_from_dataset_like():
    """Counterpart of the reference's tests/test_atmosphere.py: variables found by CF standard
    name, or through a user mapping; a missing name is a ValueError."""
    from pylbl_amd import Atmosphere
    atmos, names, data_vars = _dataset()
    for mapping in (None, {"play": "pressure", "tlay": "temperature", "mole_fraction": names}):
        atm = Atmosphere(_Dataset(data_vars), mapping=mapping)
        assert np.array_equal(atm.pressure, atmos.p) and np.array_equal(atm.temperature, atmos.t)
        assert set(atm.gases) == set(names)
        for formula in names:
            assert np.array_equal(atm.gases[formula], atmos.vmr[formula])
        assert atm.dims == ["layer"]
    del data_vars["pressure"]
    with pytest.raises(ValueError):
        Atmosphere(_Dataset(data_vars))


def test_atmosphere_from_plain_arrays():
    from pylbl_amd import Atmosphere
    atmos = synthetic.fixture_atmosphere()
    for given in (atmos, {"p": atmos.p, "t": atmos.t, "vmr": atmos.vmr}):
        atm = Atmosphere(given)
        assert np.array_equal(atm.temperature, atmos.t) and set(atm.gases) == set(atmos.vmr)
    with pytest.raises(ValueError):
        Atmosphere({"p": atmos.p, "t": atmos.t, "vmr": {"H2O": atmos.vmr["H2O"][:2]}})


def test_bench_closed_form_eval_count_matches_oracle(oracle):
    """bench.py counts the reference's inner-loop iterations in closed form (the compiled
    reference does not report them); same number as the restatement's own counter, with
    clipped windows, lines left and right of the grid and the range `break`."""
    import bench
    for seed, (lo, hi, v0, vn, npv) in enumerate([(1., 200., 1, 150, 7), (30., 400., 60, 90, 100),
                                                  (1., 90., 40, 80, 10), (100., 130., 100, 131, 3)]):
        table = synthetic.line_table("CO2", lo, hi, num_lines=400, seed=seed, tips_range=(150, 400))
        _, extras = oracle.absorption_port(table, 250., 5.e4, 3.6e-4, v0, vn, npv)
        assert bench.closed_form_evals(table, 5.e4, v0, vn, npv) == extras["evals"]


def test_memory_database_mirrors_the_file_backed_one(tmp_path):
    """MemoryDatabase: the read surface Spectroscopy and Gas use, over tables held in memory."""
    tables = [synthetic.line_table("H2O", 1., 100., num_lines=50, seed=1, tips_range=(150, 350)),
              synthetic.line_table("CO2", 1., 100., num_lines=40, seed=2, tips_range=(150, 350))]
    memory = database.MemoryDatabase(tables, aliases={"H2O": ["water"]})
    on_disk = database.Database(database.write_database(tmp_path / "lines.db", tables,
                                                        aliases={"H2O": ["water"]}))
    assert memory.path is None and memory.molecules() == on_disk.molecules()
    for name in ("H2O", "water", "CO2"):
        a, b = memory.line_table(name), on_disk.line_table(name)
        for column in database.LINE_COLUMNS:
            assert np.array_equal(getattr(a, column), getattr(b, column))
        assert np.array_equal(memory.tips(name)[1], on_disk.tips(name)[1])
        assert memory.gas(name).mass == on_disk.gas(name).mass
    with pytest.raises(AliasNotFoundError):
        memory.line_table("XYZ")


def test_partition_function_object_checks_its_shape():
    with pytest.raises(ValueError):
        database.TotalPartitionFunction("H2O", np.arange(3.), np.zeros((2, 4)))
    tips = database.TotalPartitionFunction("H2O", np.asarray([100., 101., 102.]),
                                           np.asarray([[1., 3., 7.]]))
    assert tips.isotopologue == [0]
    assert tips.total_partition_function(101.25, 1) == pytest.approx(4.)
    # An exact table temperature takes the interval below it (the reference's left-sided search).
    assert tips.total_partition_function(101., 1) == pytest.approx(3.)


def test_level_arrays_pass_through_without_a_copy():
    """Engine.compute takes per-level values as they come: a contiguous 1-d float64 array is used
    as it is (no numpy dispatch on the 20 us path of a small call), anything else is converted."""
    from pylbl_amd.engine import _levels
    ready = np.asarray([250., 260.])
    assert _levels(ready) is ready
    for other in (250., [250., 260.], np.asarray([250, 260]), np.asarray([1., 2., 3., 4.])[::2],
                  np.float32(250.)):
        made = _levels(other)
        assert made.dtype == np.float64 and made.ndim == 1 and made.flags.c_contiguous
        assert np.array_equal(made, np.atleast_1d(np.asarray(other, dtype=np.float64)))


def test_bench_starts_its_own_ranks_and_reports_the_worst_exit_code():
    """`python bench.py --gpus 2` with no launcher around it starts two child ranks (before it
    imports torch or touches HIP), relays what they print and leaves with the worst of their exit
    codes plus one JSON diagnostic.  Without a GPU both ranks refuse to run: that refusal, the
    diagnostic and the exit code are what this checks (the GPU twin is
    tests/test_gpu_bench_contract.py::test_two_rank_line[bench.py])."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the ranks would run the benchmark")
    environment = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    result = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo",
                             "--steps", "1", "--warmup", "0", "--launch-timeout", "120",
                             # (the first rank to refuse must not have the other ended before it
                             # has imported torch and refused as well: a loaded machine takes
                             # longer over that import than the default five seconds)
                             "--launch-grace", "100"],
                            capture_output=True, text=True, cwd=ROOT, env=environment, timeout=300)
    assert result.returncode == 1
    assert result.stdout.strip() == ""                 # no result line from a run that failed
    refusals = [x for x in result.stderr.splitlines() if "needs an MI355X" in x]
    assert len(refusals) == 2
    report = json.loads([x for x in result.stderr.splitlines() if x.startswith("{")][-1])
    assert report["bench_failed"] and report["launcher"] and report["exit_codes"] == [1, 1]


def _launch(tmp_path, body, gpus=3, timeout=30., grace=0.3):
    """bench.launch_ranks over a stand-in for the ranks' program (no GPU needed): returns
    (exit code, what the launcher relayed on stdout, on stderr)."""
    import subprocess
    import sys
    import textwrap
    child = tmp_path / "rank.py"
    child.write_text(textwrap.dedent(body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import argparse, sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        args = argparse.Namespace(gpus={gpus}, launch_timeout={timeout}, launch_grace={grace})
        sys.exit(bench.launch_ranks(args, command=[sys.executable, {str(child)!r}]))
        """))
    environment = {k: v for k, v in __import__("os").environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    result = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True,
                            env=environment, timeout=120)
    return result.returncode, result.stdout, result.stderr


def test_bench_launcher_relays_rank_zero_and_hands_every_rank_its_place(tmp_path):
    """What `python bench.py --gpus N` does without a launcher around it: N children with RANK,
    LOCAL_RANK, WORLD_SIZE and one MASTER_ADDR / MASTER_PORT, rank 0's stdout relayed as it is,
    the others' on stderr with their rank, exit code 0 when all leave with 0."""
    code, out, err = _launch(tmp_path, """
        import json, os
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_WORLD_SIZE"] == str(world)
        print(json.dumps({"rank": rank, "n_gpus": world, "port": os.environ["MASTER_PORT"]}))
        """)
    import json
    assert code == 0, err
    lines = [json.loads(x) for x in out.strip().splitlines()]
    assert lines == [{"rank": 0, "n_gpus": 3, "port": lines[0]["port"]}]
    others = sorted(x for x in err.splitlines() if x.startswith("[rank "))
    assert [x.split("]")[0] for x in others] == ["[rank 1", "[rank 2"]
    assert all(json.loads(x.split("] ", 1)[1])["port"] == lines[0]["port"] for x in others)


def test_bench_launcher_ends_the_others_when_a_rank_fails_or_time_runs_out(tmp_path):
    """A rank that leaves with 3 (bench.py's code for an exchange that timed out) ends ranks that
    would wait for ever, and 3 is what the launcher returns; --launch-timeout ends all of them."""
    import json
    import time
    start = time.perf_counter()
    code, out, err = _launch(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            time.sleep(0.5)
            sys.exit(3)
        time.sleep(600)
        """)
    assert code == 3 and time.perf_counter() - start < 60.
    report = json.loads([x for x in err.splitlines() if x.startswith("{")][-1])
    assert report["launcher"] and report["reason"] == "rank 1 left with code 3"
    assert report["exit_codes"][1] == 3 and all(c < 0 for i, c in enumerate(report["exit_codes"])
                                                 if i != 1)          # ended by a signal
    code, out, err = _launch(tmp_path, "import time; time.sleep(600)", gpus=2, timeout=1.)
    report = json.loads([x for x in err.splitlines() if x.startswith("{")][-1])
    assert code != 0 and "launch-timeout" in report["reason"]


def test_bench_reads_the_newest_profile_by_its_round_tag():
    """roofline.traffic / roofline.issue come from the committed rocprofv3 summaries of the same
    workload; "newest" is the highest round tag in the file name (a fresh checkout gives every file
    the same modification time)."""
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    import bench
    lines = sorted((ROOT / "profiles").glob("bench_r*.json"), key=lambda p: p.name, reverse=True)
    workload = next(json.loads(p.read_text())["config"]["workload"] for p in lines
                    if p.name.count("_") == 1)       # a default line: bench_<tag>.json
    traffic, source = bench.profiled_traffic(workload)
    summaries = sorted(p.name for p in (ROOT / "profiles").glob("r??[a-z]_summary.json"))
    assert source == summaries[-1] and traffic > 0
    issue = bench.profiled_issue(workload)
    counters = sorted(p.name for p in (ROOT / "profiles").glob("r??[a-z]_valu_counters.json"))
    assert issue["source"] == "profiles/" + counters[-1]


def test_bench_short_line_fits_the_drivers_tail_also_for_eight_ranks():
    """benchlegs.compact: the ONE line bench.py prints is an extract of the whole record -- the
    contract's keys untouched, roofline / cpu_baseline without prose, one short record per leg --
    and stays far below the 8 KB of stdout the driver keeps, also when eight ranks report."""
    import copy
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    from benchlegs import compact
    records = sorted((ROOT / "profiles").glob("bench_r??[a-z]_full.json"), key=lambda p: p.name)
    if not records:
        pytest.skip("no whole bench record under profiles/")
    full = json.loads(records[-1].read_text())
    line = compact.compact(full, full_record="bench_full.json")
    text = json.dumps(line)
    assert len(text) <= compact.TARGET_BYTES < compact.LIMIT_BYTES
    for key in compact.CONTRACT:
        assert line[key] == full[key], key
    assert line["config"]["workload"] == full["config"]["workload"] and "model" not in line["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
            "launches_timed", "evals_per_launch"} <= set(line["roofline"])
    assert not any(isinstance(v, str) and len(v) > 40 for v in line["roofline"].values())
    assert {"value", "unit", "cores", "kind", "sample", "cpu", "host_cores"} <= set(line["cpu_baseline"])
    assert {"pedestal", "config0", "config1", "config2", "config3_share", "config4_share",
            "farfield_plain", "farfield_pedestal", "api_call", "continuum", "xsec",
            "ingest_s"} <= set(line["legs"])
    issue = line["roofline"]["issue"]
    assert abs(issue["evals_per_launch"] - line["roofline"]["evals_per_launch"]) \
        <= 1e-3*line["roofline"]["evals_per_launch"]
    assert 0. < issue["frac_of_issue_ceiling_at_2.4GHz"] <= 1.
    # Eight ranks (what the driver's scaling run prints): per-rank spectra/s and exchange waits on
    # the short line, device identities only in the whole record.
    eight = copy.deepcopy(full)
    eight["n_gpus"] = 8
    eight["distributed"] = {
        "world_size": 8, "backend": "nccl", "launcher": "torch.distributed.run",
        "distinct_devices": 8, "ranks_sharing_a_device": {},
        "kernels_to_exchange_ordering": "device (events between the engine's streams and the "
                                        "exchange's, no host wait)",
        "bytes_to_rank0_per_step": 7*80000000, "exchange_alone_ms_max": 3.2,
        "ranks": [{"rank": r, "device_index": r, "name": "AMD Instinct MI355X", "uuid": "x"*36,
                   "ms_per_step": 4.2 + 0.01*r, "spectra_per_s": 238.1, "evals_per_step": 25947343421,
                   "exchange_wait_ms_per_step": 0.02*r, "bytes_sent_per_step": 80000000,
                   "unoverlapped_exchange_ms": 3.1} for r in range(8)]}
    line8 = compact.compact(eight, full_record="bench_full.json")
    assert len(json.dumps(line8)) <= compact.TARGET_BYTES
    assert [r["rank"] for r in line8["distributed"]["ranks"]] == list(range(8))
    assert all({"spectra_per_s", "exchange_wait_ms_per_step"} <= set(r)
               for r in line8["distributed"]["ranks"])
    assert "uuid" not in line8["distributed"]["ranks"][0]


def test_banded_tables_inside_the_grid_drop_what_falls_outside():
    a = synthetic.banded_line_table("CO2", 1., 5000., num_lines=20000, bands=8, seed=5)
    b = synthetic.banded_line_table("CO2", 1., 5000., num_lines=20000, bands=8, seed=5, inside=True)
    assert a.num_lines == 20000 and np.all(np.diff(a.nu) >= 0)
    assert b.num_lines <= 20000 and np.all(np.diff(b.nu) >= 0)
    assert b.nu[0] > 1. and b.nu[-1] < 5000.
    # nothing piled onto the ends of the range
    assert np.count_nonzero(b.nu == b.nu[-1]) == 1 and np.count_nonzero(b.nu == b.nu[0]) == 1
    for column in ("sw", "gamma_air", "delta_air", "local_iso_id"):
        assert getattr(b, column).shape == (b.num_lines,)
