"""The drop-in surface on a real GPU: Gas over a database file, the Spectroscopy lines slot,
the same-signature C entry, device-resident outputs, and size-independent properties at the
benchmark's full grid (5 M points)."""
from ctypes import c_char_p, c_double, c_int
import numpy as np
import pytest

from pylbl_amd import synthetic
from pylbl_amd.database import Database, write_database
from tests import golden_io

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_database(tmp_path_factory):
    tables = [synthetic.line_table("H2O", 1., 130., num_lines=400, seed=41, tips_range=(150, 400)),
              synthetic.line_table("CO2", 1., 130., num_lines=600, seed=42, tips_range=(150, 400)),
              synthetic.line_table("N2O", 1., 130., num_lines=50, seed=43, tips_range=(150, 400))]
    path = tmp_path_factory.mktemp("db") / "lines.db"
    write_database(path, tables, with_tips={"H2O", "CO2"})
    return Database(str(path)), {t.formula: t for t in tables}


def check(k, k_ref, n_per_v, remove_pedestal, label):
    case = golden_io.Case("api", 0, 0, 0, 0, 0, 0, n_per_v, 25, remove_pedestal, None, 0)
    if remove_pedestal:
        tol = golden_io.pedestal_tolerance(k_ref, n_per_v, 25, 1.e-6) + 1e-300
        assert np.max(np.abs(k - k_ref)/tol) <= 1., label
    else:
        np.testing.assert_allclose(k, k_ref, rtol=1.e-6, atol=0., err_msg=label)


def test_gas_over_database_file(small_database, oracle):
    """Same constructor and call as the reference's test (tests/test_gas_optics.py:7-16)."""
    from pylbl_amd import Gas
    db, tables = small_database
    grid = np.arange(1., 100., 0.05)
    v0, vn, npv = synthetic.grid_arguments(grid)
    gas = Gas(db, "H2O")
    assert gas.database == db.path and gas.formula == "H2O"
    for ped in (False, True):
        k = gas.absorption_coefficient(temperature=288.99, pressure=98388.,
                                       volume_mixing_ratio=6.637074e-3, grid=grid,
                                       remove_pedestal=ped)
        assert k.shape == ((vn - v0)*npv,) and k.size >= grid.size
        k_ref, _ = oracle.absorption_port(tables["H2O"], 288.99, 98388., 6.637074e-3, v0, vn,
                                          npv, remove_pedestal=ped)
        check(k, k_ref, npv, ped, f"Gas H2O ped={ped}")
    # The reference's range rule: first row below v0-26 -> zeros (absorption.c:80-83); the
    # drop-in reproduces it and says so, range_policy="skip" is the way out.
    high = np.arange(60., 90., 0.05)
    with pytest.warns(RuntimeWarning, match="all-zero"):
        assert not gas.absorption_coefficient(288.99, 98388., 6.6e-3, high).any()
    assert gas.absorption_coefficient(288.99, 98388., 6.6e-3, high, range_policy="skip").any()
    # No TIPS rows: rc 0 and zeros in the reference (absorption.c:53-59).
    k = Gas(db, "N2O").absorption_coefficient(288.99, 98388., 3.2e-7, grid)
    assert k.shape == ((vn - v0)*npv,) and not k.any()
    # Unknown alias: the reference raises ValueError from the return-code hook
    # (gas_optics.py:15-26, spectral_database.c:152-156).
    with pytest.raises(ValueError):
        Gas(db, "XYZ").absorption_coefficient(288.99, 98388., 1e-6, grid)


def test_spectroscopy_lines_slot(small_database, oracle):
    """beta[level, 0, :] = n k[:grid.size] for every gas and level
    (pyLBL/spectroscopy.py:166-191), three output formats (:208-235)."""
    from pylbl_amd import Spectroscopy, number_density
    db, tables = small_database
    full = synthetic.fixture_atmosphere()
    atmos = synthetic.Atmos(p=full.p, t=full.t, vmr={k: full.vmr[k] for k in ("H2O", "CO2", "N2O")})
    grid = np.arange(1., 90., 0.1)
    v0, vn, npv = synthetic.grid_arguments(grid)
    spec = Spectroscopy(atmos, grid, db)
    assert spec.list_molecules() == ["H2O", "CO2", "N2O"]
    for ped in (True, False):
        out = spec.compute_absorption(output_format="all", remove_pedestal=None if ped else False)
        assert list(out["mechanism"]) == ["lines", "continuum", "cross_section"]
        assert np.array_equal(out["wavenumber"], grid)
        for formula in ("H2O", "CO2"):
            beta = np.asarray(out[f"{formula}_absorption"])
            assert beta.shape == (4, 3, grid.size)
            assert beta[:, 1, :].any() and not beta[:, 2, :].any()   # slot 1: test_gpu_continuum
            for level in range(4):
                t, p, x = atmos.t[level], atmos.p[level], atmos.vmr[formula][level]
                k_ref, _ = oracle.absorption_port(tables[formula], t, p, x, v0, vn, npv,
                                                  remove_pedestal=ped)
                ref = number_density(t, p, x)*k_ref[:grid.size]
                check(beta[level, 0], ref, npv, ped, f"{formula} level {level} ped={ped}")
        assert not np.asarray(out["N2O_absorption"]).any()
    per_gas = spec.compute_absorption(output_format="gas", remove_pedestal=False)
    total = spec.compute_absorption(output_format="total", remove_pedestal=False)
    assert np.asarray(per_gas["H2O_absorption"]).shape == (4, grid.size)
    np.testing.assert_allclose(np.asarray(total["absorption"]),
                               np.asarray(per_gas["H2O_absorption"]) +
                               np.asarray(per_gas["CO2_absorption"]), rtol=1e-13)
    # Same with the pedestal removed (the default): the on-device sum goes through the
    # un-pedestalled scratch buffer + pedestal_apply_kernel's accumulate form.
    per_gas = spec.compute_absorption(output_format="gas")
    total = spec.compute_absorption(output_format="total")
    expect = np.asarray(per_gas["H2O_absorption"]) + np.asarray(per_gas["CO2_absorption"])
    assert np.max(np.abs(np.asarray(total["absorption"]) - expect)) <= 1e-13*np.max(expect)


def test_same_signature_c_entry(small_database, oracle):
    """lbl_absorption / absorption: the reference's 11-argument call
    (absorption.c:19-30) driven exactly like gas_optics.py:61-91."""
    from numpy.ctypeslib import ndpointer
    from pylbl_amd import engine
    db, tables = small_database
    lib = engine.library()
    for name in ("lbl_absorption", "absorption"):
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = 3*[c_double] + 3*[c_int] + [ndpointer(c_double, flags="C_CONTIGUOUS")] + \
            2*[c_char_p] + 2*[c_int]
        v0, vn, npv = 1, 101, 10
        for ped in (0, 1):
            k = np.full((vn - v0)*npv, 7.)
            rc = fn(98388., 288.99, 3.6e-4, v0, vn, npv, k, db.path.encode(), b"CO2", 25, ped)
            assert rc == 0
            k_ref, _ = oracle.absorption_port(tables["CO2"], 288.99, 98388., 3.6e-4, v0, vn, npv,
                                              remove_pedestal=bool(ped))
            check(k, k_ref, npv, bool(ped), f"{name} ped={ped}")
        k = np.full((vn - v0)*npv, 7.)
        assert fn(98388., 288.99, 3.2e-7, v0, vn, npv, k, db.path.encode(), b"N2O", 25, 0) == 0
        assert not k.any()                                  # no TIPS rows: zeros, rc 0
        assert fn(98388., 288.99, 3.2e-7, v0, vn, npv, k, db.path.encode(), b"XYZ", 25, 0) == 1


def test_file_to_hbm_in_one_call(small_database):
    """lbl_molecule_load_sqlite (Engine.load_sqlite): the C reader of the same-signature entry on
    an engine the host owns -- same bits as reading the table in Python and uploading it, and the
    reader's statuses for what a file lacks."""
    from pylbl_amd import engine as engine_module
    from pylbl_amd.engine import default_engine
    from pylbl_amd.errors import EngineError
    engine = default_engine(0)
    db, tables = small_database
    t, p, x = np.asarray([288.99, 230.]), np.asarray([98388., 2.e4]), np.asarray([3.6e-4, 3.e-4])
    direct = engine.load_sqlite(db.path, "CO2")
    uploaded = engine.load(db.line_table("CO2"))
    try:
        for ped in (False, True):
            a = engine.compute(direct, t, p, x, 1, 101, 20, remove_pedestal=ped).copy()
            b = engine.compute(uploaded, t, p, x, 1, 101, 20, remove_pedestal=ped)
            assert a.any() and np.array_equal(a, b)
    finally:
        engine.free(direct)
        engine.free(uploaded)
    for name, status in (("XYZ", engine_module.TABLE_NO_ALIAS), ("N2O", engine_module.TABLE_NO_TIPS)):
        with pytest.raises(EngineError, match=f"status {status}"):
            engine.load_sqlite(db.path, name)
    with pytest.raises(EngineError, match=f"status {engine_module.TABLE_OPEN_FAILED}"):
        engine.load_sqlite(db.path + ".missing", "CO2")


def test_device_output_scale_and_accumulate(oracle):
    from pylbl_amd.engine import DeviceSpectra, Engine
    from pylbl_amd import number_density
    e = Engine(0)
    a = synthetic.line_table("H2O", 1., 90., num_lines=150, seed=51, tips_range=(150, 400))
    b = synthetic.line_table("CO2", 1., 90., num_lines=250, seed=52, tips_range=(150, 400))
    ha, hb = e.load(a), e.load(b)
    t, p = np.asarray([288.99, 220.]), np.asarray([98388., 5000.])
    xa, xb = np.asarray([6e-3, 4e-6]), np.asarray([3.6e-4, 3.6e-4])
    v0, vn, npv = 1, 61, 25
    out = DeviceSpectra(e, 2, (vn - v0)*npv)
    e.compute(ha, t, p, xa, v0, vn, npv, out=out, scale_density=True)
    e.compute(hb, t, p, xb, v0, vn, npv, out=out, scale_density=True, accumulate=True)
    total = out.to_host()
    for level in range(2):
        ka, _ = oracle.absorption_port(a, t[level], p[level], xa[level], v0, vn, npv)
        kb, _ = oracle.absorption_port(b, t[level], p[level], xb[level], v0, vn, npv)
        ref = number_density(t[level], p[level], xa[level])*ka + \
            number_density(t[level], p[level], xb[level])*kb
        np.testing.assert_allclose(total[level], ref, rtol=1e-6)
    out.free()
    e.close()


def test_asynchronous_pedestal_calls_share_the_gpu(oracle):
    """Asynchronous calls with a pedestal rotate over the engine's lanes (own streams and
    workspaces); results must equal the blocking calls bit for bit."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    tables = [synthetic.line_table(f, 1., 120., num_lines=400 + 100*i, seed=60 + i,
                                   tips_range=(150, 400))
              for i, f in enumerate(("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "N2", "H2O",
                                     "CO2"))]
    handles = [e.load(t) for t in tables]
    atmos = synthetic.fixture_atmosphere()
    v0, vn, npv = 1, 91, 40
    outs = [DeviceSpectra(e, 4, (vn - v0)*npv) for _ in tables]
    import os
    # The second round reuses every lane's workspace; PYLBL_SOAK_ROUNDS=500 for a soak.
    for _ in range(int(os.environ.get("PYLBL_SOAK_ROUNDS", "2"))):
        for h, t, out in zip(handles, tables, outs):
            e.compute(h, atmos.t, atmos.p, atmos.vmr[t.formula], v0, vn, npv,
                      remove_pedestal=True, out=out, asynchronous=True)
    e.synchronize()
    for h, t, out in zip(handles, tables, outs):
        blocking = e.compute(h, atmos.t, atmos.p, atmos.vmr[t.formula], v0, vn, npv,
                             remove_pedestal=True)
        assert np.array_equal(out.to_host(), blocking)
        k_ref, _ = oracle.absorption_port(t, atmos.t[3], atmos.p[3], atmos.vmr[t.formula][3],
                                          v0, vn, npv, remove_pedestal=True)
        check(blocking[3], k_ref, npv, True, t.formula)
        out.free()
    e.close()


@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_level_chunking_and_strides(remove_pedestal):
    """A tiny workspace budget forces several passes over the levels; results must not depend
    on the chunking, on host vs device output, or on a padded level stride."""
    from ctypes import c_int64, c_void_p
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CH4", 1200., 1400., num_lines=20000, seed=81,
                                 tips_range=(150, 400))
    h = e.load(table)
    atmos = synthetic.standard_atmosphere(7)
    v0, vn, npv = 1250, 1330, 200
    n = (vn - v0)*npv
    whole = e.compute(h, atmos.t, atmos.p, atmos.vmr["CH4"], v0, vn, npv,
                      remove_pedestal=remove_pedestal)
    e.set_option("workspace_bytes", 1 << 20)           # ~3 levels' worth -> several passes
    chunked = e.compute(h, atmos.t, atmos.p, atmos.vmr["CH4"], v0, vn, npv,
                        remove_pedestal=remove_pedestal)
    assert np.array_equal(chunked, whole)
    out = DeviceSpectra(e, 7, n)
    e.compute(h, atmos.t, atmos.p, atmos.vmr["CH4"], v0, vn, npv,
              remove_pedestal=remove_pedestal, out=out)
    assert np.array_equal(out.to_host(), whole)
    out.free()
    # Padded host rows through the raw C ABI (level_stride > n).
    stride = n + 37
    padded = np.full((7, stride), -1.)
    t, p, x = (np.ascontiguousarray(a, dtype=np.float64)
               for a in (atmos.t, atmos.p, atmos.vmr["CH4"]))
    rc = e.lib.lbl_compute(e.handle, h, 7, t.ctypes.data, p.ctypes.data, x.ctypes.data, v0, vn,
                           npv, 25, int(remove_pedestal), 0, 0, c_void_p(padded.ctypes.data),
                           c_int64(stride), None)
    assert rc == 0
    assert np.array_equal(padded[:, :n], whole) and np.all(padded[:, n:] == -1.)
    e.close()


def test_gases_added_into_one_block_in_several_passes():
    """Four gases add n k into one block, queued back to back, each call cut into several passes
    over the levels by a tiny workspace budget: every pass -- not only the last -- has to wait for
    the earlier gases' kernels that add into the same rows (they run on other lanes)."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    atmos = synthetic.standard_atmosphere(12)
    v0, vn, npv = 1250, 1330, 500
    n = (vn - v0)*npv
    formulas = ["CH4", "H2O", "CO2", "N2O"]
    tables = [synthetic.line_table(f, 1200., 1400., num_lines=20000 + 5000*i, seed=90 + i,
                                   tips_range=(150, 400)) for i, f in enumerate(formulas)]
    handles = [e.load(t) for t in tables]
    apart = [e.compute(h, atmos.t, atmos.p, atmos.vmr[f], v0, vn, npv, remove_pedestal=True,
                       scale_density=True).copy() for h, f in zip(handles, formulas)]
    expected = np.zeros_like(apart[0])
    for part in apart:
        expected += part                        # the order the block receives them in
    e.set_option("workspace_bytes", 1 << 20)    # a few levels per pass
    total = DeviceSpectra(e, 12, n)
    for repeat in range(5):
        e.fill_zero(total, asynchronous=True)
        for h, f in zip(handles, formulas):
            e.compute(h, atmos.t, atmos.p, atmos.vmr[f], v0, vn, npv, remove_pedestal=True,
                      scale_density=True, accumulate=True, out=total, asynchronous=True)
        e.synchronize()
        assert np.array_equal(total.to_host(), expected), f"repeat {repeat}"
    total.free()
    e.close()


def test_busy_time_counts_overlapping_launches_once():
    """lbl_timing sums launch durations; lbl_timing_busy gives the time at least one launch of the
    kind ran.  Blocking calls: the same.  Queued calls on two lanes: the launches overlap, the sum
    exceeds the busy time, and the busy time cannot exceed the wall time of the batch."""
    import time
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CO2", 1., 3000., num_lines=200_000, seed=5)
    h = e.load(table)
    level = synthetic.surface_level(["CO2"])
    v0, vn, npv = 1, 3001, 1000
    blocks = [DeviceSpectra(e, 1, (vn - v0)*npv) for _ in range(2)]
    args = (h, level.t, level.p, level.vmr["CO2"], v0, vn, npv)
    for block in blocks:
        e.compute(*args, out=block)                     # plans and workspaces of lane 0
    for i in range(4):
        e.compute(*args, out=blocks[i % 2], asynchronous=True)   # ... and of the second lane
    e.synchronize()
    e.set_option("timing", 2)
    e.timing(reset=True)
    for i in range(4):
        e.compute(*args, out=blocks[i % 2])
    busy = e.timing_busy()[2]
    summed, launches = e.timing(reset=True)
    assert launches[2] == 4
    assert busy == pytest.approx(summed[2], rel=1e-3)
    begin = time.perf_counter()
    for i in range(8):
        e.compute(*args, out=blocks[i % 2], asynchronous=True)
    e.synchronize()
    wall_ms = (time.perf_counter() - begin)*1e3
    busy = e.timing_busy()[2]
    summed, launches = e.timing(reset=True)
    assert launches[2] == 8
    assert busy < 0.95*summed[2]            # launches of successive calls run side by side
    assert busy <= wall_ms
    assert e.timing_busy()[2] == 0.         # cleared with the sums
    e.set_option("timing", 0)
    for block in blocks:
        block.free()
    e.close()


def test_error_paths():
    from pylbl_amd.engine import Engine
    from pylbl_amd.errors import EngineError
    e = Engine(0)
    table = synthetic.line_table("H2O", 1., 90., num_lines=20, seed=1, tips_range=(200, 350))
    h = e.load(table)
    with pytest.raises(EngineError, match="partition-function"):
        e.compute(h, 150., 1000., 1e-3, 1, 50, 10)          # T below the TIPS table
    with pytest.raises(EngineError):
        e.compute(h, 250., 1000., 1e-3, 50, 50, 10)         # empty grid
    with pytest.raises(EngineError):
        e.compute(99, 250., 1000., 1e-3, 1, 50, 10)         # unknown handle
    # A row whose isotopologue has no mass / partition-function row is an error only when a
    # call reaches it: the reference never reads rows behind its range `break`
    # (absorption.c:80-83), so a database it can process must not be refused at load.
    bad = synthetic.line_table("H2O", 1., 90., num_lines=20, seed=1, tips_range=(200, 350))
    row = int(np.searchsorted(bad.nu, 60.))
    bad.local_iso_id[row] = 7                                 # no mass / TIPS row for iso 7
    hb = e.load(bad)
    with pytest.raises(EngineError, match="local_iso_id"):
        e.compute(hb, 250., 1000., 1e-3, 1, 50, 10)          # row within v0-26 .. vn+26
    reached = e.compute(hb, 250., 1000., 1e-3, 1, 20, 10)    # break before the bad row
    good = bad.subset(np.arange(bad.num_lines) != row)
    hg = e.load(good)
    assert np.array_equal(reached, e.compute(hg, 250., 1000., 1e-3, 1, 20, 10))
    with pytest.raises(EngineError, match="local_iso_id"):
        e.compute(hb, 250., 1000., 1e-3, 40, 80, 10, range_policy="skip")
    k = e.compute(h, 250., 1000., 1e-3, 1, 50, 10)           # still usable afterwards
    assert k.shape == (1, 490) and k.any()
    e.close()


@pytest.fixture(scope="module")
def full_size():
    """BASELINE target workload: H2O + CO2, grid 1-5000 cm-1 at 0.001 cm-1."""
    from pylbl_amd.engine import Engine
    e = Engine(0)
    tables = {f: synthetic.line_table(f, 1., 5000.) for f in ("H2O", "CO2")}
    handles = {f: e.load(t) for f, t in tables.items()}
    yield e, tables, handles
    e.close()


def test_full_size_windows_against_oracle(full_size, oracle):
    """Grid points only see lines within cut_off+1 cm-1, so narrow sub-grids computed by the
    oracle must reproduce the matching slices of the 5 M-point spectrum."""
    e, tables, handles = full_size
    t, p, x = 288.99, 98388., {"H2O": 6.637074e-3, "CO2": 3.5999e-4}
    v0, vn, npv = 1, 5001, 1000
    for formula in ("H2O", "CO2"):
        k, evals = e.compute(handles[formula], t, p, x[formula], v0, vn, npv, want_evals=True)
        k = k[0]
        table = tables[formula]
        centre = table.nu + p*9.86923e-6*table.delta_air
        first = np.clip((np.floor(centre) - 25 - v0)*npv, 0, None)
        last = np.clip((np.floor(centre) + 26 - v0)*npv, None, (vn - v0)*npv - 1)
        keep = first < (vn - v0)*npv
        assert evals == int(np.sum((last - first + 1)[keep]))
        for lo in (1, 667, 2349, 4998):
            hi = lo + 2
            # Oracle on [lo-2, hi+2] (clipped to the grid) with exactly the rows its range
            # rule accepts; only the interior [lo, hi) is compared, so rows a hair outside
            # the sub-grid's range cannot matter.
            g0, g1 = max(lo - 2, v0), min(hi + 2, vn)
            near = table.subset((table.nu >= g0 - 26.) & (table.nu <= g1 + 26.))
            k_ref, _ = oracle.absorption_port(near, t, p, x[formula], g0, g1, npv)
            k_ref = k_ref[(lo - g0)*npv:(hi - g0)*npv]
            piece = k[(lo - v0)*npv:(hi - v0)*npv]
            np.testing.assert_allclose(piece, k_ref, rtol=1e-6, err_msg=f"{formula} {lo}")


def test_full_size_linearity_additivity_determinism(full_size):
    e, tables, handles = full_size
    t, p, x = 250., 20000., 1e-3
    v0, vn, npv = 1, 5001, 1000
    table = tables["H2O"]
    k = e.compute(handles["H2O"], t, p, x, v0, vn, npv)[0]
    assert np.array_equal(k, e.compute(handles["H2O"], t, p, x, v0, vn, npv)[0])  # bitwise
    assert np.all(k > 0.) and np.all(np.isfinite(k))
    # Doubling every line strength doubles the spectrum (a power of two: exactly).
    doubled = table.subset(np.ones(table.num_lines, bool))
    doubled.sw = 2.*table.sw
    h2 = e.load(doubled)
    assert np.array_equal(e.compute(h2, t, p, x, v0, vn, npv)[0], 2.*k)
    e.free(h2)
    # Splitting the table in two and adding the spectra gives the same sum.
    odd = np.arange(table.num_lines) % 2 == 1
    ha, hb = e.load(table.subset(odd)), e.load(table.subset(~odd))
    parts = e.compute(ha, t, p, x, v0, vn, npv)[0] + e.compute(hb, t, p, x, v0, vn, npv)[0]
    np.testing.assert_allclose(parts, k, rtol=1e-12)
    e.free(ha)
    e.free(hb)


def test_full_size_farfield_option(full_size):
    """At the benchmark size the opt-in far-field series must reproduce the direct kernel to
    its truncation level (<= ~1.5e-11 by construction; 1e-9 asserted), at 1 atm and 10 Pa."""
    e, tables, handles = full_size
    v0, vn, npv = 1, 5001, 1000
    try:
        for formula, x in (("H2O", 6.6e-3), ("CO2", 3.6e-4)):
            for t, p in ((288.99, 98388.), (232.7, 10.)):
                e.set_option("farfield", 0)
                direct = e.compute(handles[formula], t, p, x, v0, vn, npv)[0]
                e.set_option("farfield", 1)
                series = e.compute(handles[formula], t, p, x, v0, vn, npv)[0]
                assert np.max(np.abs(series - direct)/direct) < 1.e-9, (formula, p)
    finally:
        e.set_option("farfield", 0)


def test_full_size_pedestal_properties(full_size):
    """With the pedestal removed the spectrum stays >= 0 up to rounding, is <= the plain
    one, and levels in a batch do not influence each other."""
    e, tables, handles = full_size
    atmos = synthetic.fixture_atmosphere()
    v0, vn, npv = 1, 5001, 1000
    plain = e.compute(handles["CO2"], atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv)
    ped = e.compute(handles["CO2"], atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv,
                    remove_pedestal=True)
    assert np.all(ped <= plain*(1. + 1e-12))
    assert np.all(ped >= -1e-9*plain.max(axis=1, keepdims=True))
    single = e.compute(handles["CO2"], atmos.t[2], atmos.p[2], atmos.vmr["CO2"][2], v0, vn, npv,
                       remove_pedestal=True)[0]
    assert np.array_equal(single, ped[2])


def test_reference_known_answers_if_real_database_present():
    """The reference's own known-answer test (tests/test_gas_optics.py:6-19) needs its
    HITRAN-derived database ``pyLBL-2-7-23.db`` (anonymous FTP + secret directory), which does
    not exist offline.  If a copy is pointed to by $PYLBL_DATABASE the same two numbers are
    checked here with the reference's own tolerance."""
    import os
    path = os.environ.get("PYLBL_DATABASE")
    if not path or not os.path.isfile(path):
        pytest.skip("no real pyLBL database available ($PYLBL_DATABASE)")
    from pylbl_amd import Gas
    atmos = synthetic.fixture_atmosphere()
    grid = np.arange(1., 3250., 0.1)
    gas = Gas(Database(path), "H2O")
    k = gas.absorption_coefficient(temperature=atmos.t[-1], pressure=atmos.p[-1],
                                   volume_mixing_ratio=atmos.vmr["H2O"][-1], grid=grid)
    k = k[:grid.size]
    assert np.log(np.max(k)) == pytest.approx(-48.159224953962244)
    assert np.log(np.sum(k)*(grid[1] - grid[0])) == pytest.approx(-46.496121930910135)


def test_reference_end_to_end_known_answer_if_real_database_present():
    """The reference's end-to-end test (tests/test_spectroscopy.py:15-25): total absorption
    of the single-layer atmosphere (8 gases; lines + continua + cross-sections) on
    arange(1, 3000, 1).  Needs the real database and the cross-section files it points to."""
    import os
    path = os.environ.get("PYLBL_DATABASE")
    if not path or not os.path.isfile(path):
        pytest.skip("no real pyLBL database available ($PYLBL_DATABASE)")
    from pylbl_amd import Spectroscopy
    grid = np.arange(1., 3000., 1.)
    spec = Spectroscopy(synthetic.surface_level(), grid, Database(path))
    total = np.asarray(spec.compute_absorption(output_format="total")["absorption"])
    assert np.max(total) == pytest.approx(154.77712952851365)
    assert np.log(np.sum(total)) == pytest.approx(7.212513759327571)


def test_plan_cache_is_bounded_and_results_do_not_depend_on_it(oracle):
    """The per-molecule work-item plans are kept for the 16 most recent grids; cycling through
    more grids than that (asynchronously, so evictions meet queued kernels) changes nothing."""
    from pylbl_amd.engine import DeviceSpectra, default_engine
    engine = default_engine(0)
    table = synthetic.line_table("CO2", 1., 400., num_lines=3000, seed=77)
    handle = engine.load(table)
    first = {}
    for sweep in range(2):
        for k in range(20):
            v0, vn, npv = 10 + 5*k, 200 + 5*k, 10
            out = DeviceSpectra(engine, 1, (vn - v0)*npv)
            engine.compute(handle, 250., 5e4, 3.6e-4, v0, vn, npv, out=out, asynchronous=True)
            engine.synchronize()
            values = out.to_host()[0]
            out.free()
            if sweep == 0:
                first[k] = values
            else:
                assert np.array_equal(values, first[k])
    k_ref, _ = oracle.absorption_port(table, 250., 5e4, 3.6e-4, 10, 200, 10)
    np.testing.assert_allclose(first[0], k_ref, rtol=1e-6)
    engine.free(handle)


@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_lane_count_changes_nothing(remove_pedestal):
    """Asynchronous calls rotate over 2 lanes on tiny grids and 4 with a pedestal pass (option
    lanes = 0), or over what the option says: a burst of calls into a ring of two blocks -- every
    block is overwritten many times, by calls on different lanes -- ends with the same bits as the
    blocking call, whatever the number of lanes."""
    from pylbl_amd.engine import DeviceSpectra, default_engine
    engine = default_engine(0)
    tables = [synthetic.line_table("CO2", 600., 700., num_lines=4000, seed=s) for s in (3, 4, 5)]
    handles = [engine.load(t) for t in tables]
    v0, vn, npv = 620, 680, 20
    n = (vn - v0)*npv
    expected = [engine.compute(h, 240., 3e4, 4e-4, v0, vn, npv, remove_pedestal=remove_pedestal)[0]
                for h in handles]
    ring = [DeviceSpectra(engine, 1, n) for _ in range(2)]
    try:
        for lanes in (2, 3, 8, 0):
            engine.set_option("lanes", lanes)
            for turn in range(24):
                engine.compute(handles[turn % 3], 240., 3e4, 4e-4, v0, vn, npv, out=ring[turn % 2],
                               remove_pedestal=remove_pedestal, asynchronous=True)
            engine.synchronize()
            # turn 22 wrote ring[0] last (table 22 % 3 = 1), turn 23 ring[1] (table 2)
            assert np.array_equal(ring[0].to_host()[0], expected[1]), lanes
            assert np.array_equal(ring[1].to_host()[0], expected[2]), lanes
    finally:
        engine.set_option("lanes", 0)
        for block in ring:
            block.free()
        for h in handles:
            engine.free(h)


def test_row_copies_and_pinned_results():
    """lbl_copy_rows_to_host places device rows straight into a strided destination
    (beta[level, mechanism, :]); page-locked result arrays are recycled once dropped."""
    import gc
    from pylbl_amd.engine import DeviceSpectra, default_engine
    engine = default_engine(0)
    table = synthetic.line_table("CO2", 1., 60., num_lines=300, seed=5)
    handle = engine.load(table)
    levels, v0, vn, npv = 3, 1, 41, 10
    n, columns = (vn - v0)*npv, 391
    block = DeviceSpectra(engine, levels, n)
    t, p, x = [250., 260., 270.], [5e4, 6e4, 7e4], [3e-4, 3e-4, 3e-4]
    engine.compute(handle, t, p, x, v0, vn, npv, out=block)
    dense = block.to_host()
    beta = engine.host_array([levels, 3, columns])
    beta[...] = -1.
    block.to_host_into(beta[:, 1, :], columns, asynchronous=True)
    engine.synchronize()
    assert np.array_equal(beta[:, 1, :], dense[:, :columns])
    assert np.all(beta[:, 0, :] == -1.) and np.all(beta[:, 2, :] == -1.)
    pageable = np.zeros((levels, columns))
    block.to_host_into(pageable, columns)
    assert np.array_equal(pageable, dense[:, :columns])
    with pytest.raises(ValueError):
        block.to_host_into(np.zeros((levels, columns + 1))[:, ::2], (columns + 1)//2)
    # Recycling: the buffer of a dropped array is handed out again.
    address = beta.ctypes.data
    del beta
    gc.collect()
    assert address in [pointer for _, pointer in engine.pinned.idle]
    idle = len(engine.pinned.idle)
    again = engine.host_array([levels, 3, columns])
    assert len(engine.pinned.idle) == idle - 1 and again.shape == (levels, 3, columns)
    block.free()
    engine.free(handle)


def test_output_of_molecule_without_data_is_zeroed(small_database):
    """No TIPS rows -> the reference returns the zeroed spectrum (absorption.c:41, :53-59): a
    caller-supplied buffer must read zero too, on the host and in HBM, unless it accumulates."""
    from pylbl_amd import Gas
    from pylbl_amd.engine import DeviceSpectra
    db, _ = small_database
    gas = Gas(db, "N2O")
    assert gas.molecule is None
    grid = np.arange(1., 40., 0.1)
    v0, vn, npv = synthetic.grid_arguments(grid)
    n = (vn - v0)*npv
    host = np.full((2, n), 5.)
    gas.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-7, 3e-7], grid, out=host)
    assert not host.any()
    block = DeviceSpectra(gas.engine, 2, n)
    gas.engine.lib.lbl_fill_zero(gas.engine.handle, block.pointer, 2, n, 0, 1)
    filled = block.to_host()
    assert not filled.any()
    from pylbl_amd import synthetic as syn
    other = Gas(syn.line_table("CO2", 1., 60., num_lines=100, seed=9, tips_range=(150, 400)), "CO2",
                engine=gas.engine)
    other.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-4, 3e-4], grid, out=block)
    kept = block.to_host()
    assert kept.any()
    gas.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-7, 3e-7], grid, out=block,
                                accumulate=True)
    assert np.array_equal(block.to_host(), kept)            # nothing to add
    gas.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-7, 3e-7], grid, out=block,
                                asynchronous=True)
    gas.engine.synchronize()
    assert not block.to_host().any()
    # deliver=: the page-locked view the caller was promised reads what the block reads -- zeros,
    # or (accumulating) what the earlier calls left there -- not its stale contents.
    view = gas.engine.host_array((2, grid.size))
    view[...] = -7.
    gas.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-7, 3e-7], grid, out=block,
                                asynchronous=True, deliver=view)
    gas.engine.synchronize()
    assert not view.any()
    other.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-4, 3e-4], grid, out=block)
    view[...] = -7.
    gas.absorption_coefficients([250., 260.], [5e4, 6e4], [3e-7, 3e-7], grid, out=block,
                                accumulate=True, asynchronous=True, deliver=view)
    gas.engine.synchronize()
    assert np.array_equal(view, kept[:, :grid.size])
    block.free()


def test_delivery_does_not_depend_on_the_lane(oracle):
    """A call that delivers its result while it computes skips the lanes whose stream shares the
    copy stream's hardware queue (engine option skip_delivery_lanes, probed when the engine is
    created): a matter of speed only -- with the option on or off, and whichever lane the call is
    dealt, block and delivered array are those of the plain call."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CO2", 1., 130., num_lines=6000, seed=18, tips_range=(150, 400))
    handle = e.load(table)
    atmos = synthetic.standard_atmosphere(2)
    v0, vn, npv = 1, 101, 500
    n = (vn - v0)*npv
    plain = e.compute(handle, atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True)
    block = DeviceSpectra(e, 2, n)
    for skip in (1, 0):
        e.set_option("skip_delivery_lanes", skip)
        for turn in range(6):            # every lane of the rotation, some of them skipped
            target = e.host_array((2, n - 100))
            target[...] = -1.
            e.compute(handle, atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True,
                      out=block, asynchronous=True, deliver=target, pieces=4)
            e.synchronize()
            assert np.array_equal(block.to_host(), plain), (skip, turn)
            assert np.array_equal(target, plain[:, :n - 100]), (skip, turn)
    block.free()
    e.close()


def test_asynchronous_pedestal_calls_into_one_buffer_are_ordered():
    """Two asynchronous pedestal calls run on different lanes; when they write the same block
    the later one must win, and to_host() must wait for whichever lane wrote last."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    a = e.load(synthetic.line_table("H2O", 1., 400., num_lines=30000, seed=71, tips_range=(150, 400)))
    b = e.load(synthetic.line_table("CO2", 1., 400., num_lines=500, seed=72, tips_range=(150, 400)))
    v0, vn, npv = 1, 361, 500
    expect = e.compute(b, 250., 3e4, 3.6e-4, v0, vn, npv, remove_pedestal=True)
    out = DeviceSpectra(e, 1, (vn - v0)*npv)
    for _ in range(4):
        e.compute(a, 250., 3e4, 5e-3, v0, vn, npv, remove_pedestal=True, out=out,
                  asynchronous=True)        # long
        e.compute(b, 250., 3e4, 3.6e-4, v0, vn, npv, remove_pedestal=True, out=out,
                  asynchronous=True)        # short: would finish first if unordered
        assert np.array_equal(out.to_host(), expect)
    out.free()
    e.close()


def test_molecule_without_lines_adds_nothing_and_stays_ordered():
    """A line table without rows (lbl_molecule_load accepts it) in the path that sums gases on
    the device: asynchronous, pedestal removed, adding into the block another lane is still
    writing.  It has no pedestal pass, so its accumulate kernel does the read-modify-write
    itself and must run behind the block's previous writer (it used to rotate to a lane of its
    own and could write back a stale block)."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    full = synthetic.line_table("H2O", 1., 400., num_lines=30000, seed=71, tips_range=(150, 400))
    a = e.load(full)
    empty = e.load(full.subset(np.zeros(full.num_lines, dtype=bool)))
    v0, vn, npv = 1, 361, 500
    expect = e.compute(a, 250., 3e4, 5e-3, v0, vn, npv, remove_pedestal=True, scale_density=True)
    out = DeviceSpectra(e, 1, (vn - v0)*npv)
    for _ in range(4):
        e.compute(a, 250., 3e4, 5e-3, v0, vn, npv, remove_pedestal=True, out=out,
                  scale_density=True, asynchronous=True)                     # writes, on a lane
        e.compute(empty, 250., 3e4, 1e-3, v0, vn, npv, remove_pedestal=True, out=out,
                  scale_density=True, accumulate=True, asynchronous=True)    # adds zero
        assert np.array_equal(out.to_host(), expect)
    out.free()
    e.close()


@pytest.mark.parametrize("remove_pedestal", [False, True])
@pytest.mark.parametrize("farfield", [False, True])
def test_streamed_call_delivers_what_the_plain_call_computes(remove_pedestal, farfield):
    """lbl_compute_streamed: the grid in 1..8 runs of tiles, each run's columns copied to
    page-locked host memory beside the next run's kernels.  Same kernels, same order of
    additions: the block in HBM and the delivered array equal the plain call bit for bit --
    several levels, level passes forced by a small workspace, fewer columns than points (what
    Spectroscopy asks for), a pitched target, and adding into a block that holds something."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CO2", 1., 260., num_lines=40000, seed=5, tips_range=(150, 400))
    banded = synthetic.banded_line_table("H2O", 1., 260., num_lines=30000, bands=3, seed=6)
    atmos = synthetic.standard_atmosphere(5)
    v0, vn, npv = 1, 241, 1000
    n = (vn - v0)*npv
    columns = n - 1000
    for t, formula in ((table, "CO2"), (banded, "H2O")):
        h = e.load(t)
        x = atmos.vmr[formula]
        plain = DeviceSpectra(e, 5, n)
        e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal,
                  out=plain, scale_density=True, farfield=farfield)
        expect = plain.to_host()
        for pieces in (1, 3, 4, 8):
            out = DeviceSpectra(e, 5, n)
            holder = e.host_array((5, 3, columns))
            holder[...] = -1.
            target = holder[:, 1, :]                        # pitched: rows 3*columns apart
            e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal,
                      out=out, scale_density=True, farfield=farfield, asynchronous=True,
                      deliver=target, pieces=pieces)
            e.synchronize()
            assert np.array_equal(target, expect[:, :columns]), (formula, pieces)
            assert np.all(holder[:, 0, :] == -1.) and np.all(holder[:, 2, :] == -1.)
            assert np.array_equal(out.to_host(), expect), (formula, pieces)
            out.free()
        # level passes (workspace for ~2 levels) and adding into a block that holds something
        e.set_option("workspace_bytes", 96 << 20)
        out = DeviceSpectra(e, 5, n)
        e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal, out=out,
                  scale_density=True, farfield=farfield)
        target = e.host_array((5, columns))
        e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal, out=out,
                  scale_density=True, farfield=farfield, accumulate=True, asynchronous=True,
                  deliver=target, pieces=4)
        e.synchronize()
        e.set_option("workspace_bytes", 4 << 30)
        assert np.array_equal(out.to_host()[:, :columns], target)
        assert np.max(np.abs(target - 2.*expect[:, :columns])) <= 1e-15*np.max(expect)
        out.free()
        plain.free()
        e.free(h)
    with pytest.raises(ValueError):
        e.compute(1, 250., 1e4, 1e-3, 1, 50, 10, deliver=np.zeros((1, 490)))   # host `out`
    e.close()


@pytest.mark.parametrize("farfield", [False, True])
def test_deferred_finish_queued_first_adds_last(farfield):
    """LBL_DEFER_FINISH: a long call is queued first -- prologue, accumulate, pedestal chain run at
    once, in the lane's own buffers -- but the kernels that add into the caller's block (and the
    piecewise delivery to the host) wait for lbl_finish_deferred, behind calls queued later.  The
    block and the delivered array equal the plain sequence; where the engine cannot keep a call
    back (no pedestal pass) it says so and finishes at once."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    heavy_table = synthetic.line_table("CO2", 1., 260., num_lines=60000, seed=15, tips_range=(150, 400))
    light_table = synthetic.line_table("H2O", 1., 260., num_lines=8000, seed=16, tips_range=(150, 400))
    heavy, light = e.load(heavy_table), e.load(light_table)
    atmos = synthetic.standard_atmosphere(3)
    v0, vn, npv = 1, 241, 1000
    n = (vn - v0)*npv
    args = (atmos.t, atmos.p)
    # the plain sequence: light writes, heavy adds
    expect = DeviceSpectra(e, 3, n)
    e.compute(light, *args, atmos.vmr["H2O"], v0, vn, npv, remove_pedestal=True, out=expect,
              scale_density=True, farfield=farfield)
    e.compute(heavy, *args, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True, out=expect,
              scale_density=True, accumulate=True, farfield=farfield)
    want = expect.to_host()
    for pieces in (1, 4):
        block = DeviceSpectra(e, 3, n)
        target = e.host_array((3, n - 500))
        target[...] = -1.
        e.fill_zero(block, asynchronous=True)
        e.compute(heavy, *args, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True, out=block,
                  scale_density=True, accumulate=True, asynchronous=True, farfield=farfield,
                  deliver=target, pieces=pieces, defer_finish=True)
        assert e.deferred()
        e.compute(light, *args, atmos.vmr["H2O"], v0, vn, npv, remove_pedestal=True, out=block,
                  scale_density=True, accumulate=True, asynchronous=True, farfield=farfield)
        assert e.deferred()                 # a call on another lane leaves it kept back
        e.finish_deferred()
        assert not e.deferred()
        e.synchronize()
        got = block.to_host()
        # same kernels, same values per call; only the order of the two additions differs
        assert np.max(np.abs(got - want)) <= 4e-16*np.max(np.abs(want))
        assert np.array_equal(target, got[:, :n - 500])
        block.free()
    # synchronize() finishes what is kept back
    block = DeviceSpectra(e, 3, n)
    e.fill_zero(block, asynchronous=True)
    e.compute(heavy, *args, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True, out=block,
              scale_density=True, accumulate=True, asynchronous=True, farfield=farfield,
              defer_finish=True)
    assert e.deferred()
    e.synchronize()
    assert not e.deferred() and block.to_host().any()
    # without a pedestal pass there is nothing to keep back: the call adds at once
    e.fill_zero(block, asynchronous=True)
    e.compute(heavy, *args, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=False, out=block,
              scale_density=True, accumulate=True, asynchronous=True, farfield=farfield,
              defer_finish=True)
    assert not e.deferred()
    e.synchronize()
    plain = e.compute(heavy, *args, atmos.vmr["CO2"], v0, vn, npv, scale_density=True,
                      farfield=farfield)
    assert np.array_equal(block.to_host(), plain)
    block.free()
    expect.free()
    e.close()


def test_frees_of_other_memory_leave_a_deferred_call_alone():
    """lbl_host_free / lbl_device_free finish a call kept back (LBL_DEFER_FINISH) only
    when the memory released is memory that call still has to write -- its block, the host range
    of its delivery.  A page-locked array handed back by a finalizer thread, or another block, while
    a pipeline sits between its deferred call and lbl_finish_deferred must not make the heavy gas
    add (and stream its copies) before the other gases have added."""
    from ctypes import byref, c_void_p
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CO2", 1., 260., num_lines=20000, seed=15, tips_range=(150, 400))
    heavy = e.load(table)
    atmos = synthetic.standard_atmosphere(2)
    v0, vn, npv = 1, 241, 500
    n = (vn - v0)*npv

    def keep_back(block, target):
        e.fill_zero(block, asynchronous=True)
        e.compute(heavy, atmos.t, atmos.p, atmos.vmr["CO2"], v0, vn, npv, remove_pedestal=True,
                  out=block, scale_density=True, accumulate=True, asynchronous=True,
                  deliver=target, pieces=2, defer_finish=True)
        assert e.deferred()

    block, other = DeviceSpectra(e, 2, n), DeviceSpectra(e, 2, n)
    # page-locked memory straight from the C ABI (host_array()'s pool would keep it)
    mine, unrelated = c_void_p(), c_void_p()
    assert e.lib.lbl_host_alloc(e.handle, 2*n*8, byref(mine)) == 0
    assert e.lib.lbl_host_alloc(e.handle, 1 << 20, byref(unrelated)) == 0
    from ctypes import c_double
    target = np.frombuffer((c_double*(2*n)).from_address(mine.value), dtype=np.float64).reshape(2, n)
    keep_back(block, target)
    assert e.lib.lbl_host_free(e.handle, unrelated) == 0
    assert e.deferred()                     # somebody else's array: the deferral stays
    other.free()
    assert e.deferred()                     # ... and somebody else's block
    e.finish_deferred()
    e.synchronize()
    want = block.to_host()
    assert want.any() and np.array_equal(target, want)
    # the call's own delivery array: finished (and waited for) before the pages go
    keep_back(block, target)
    del target
    assert e.lib.lbl_host_free(e.handle, mine) == 0
    assert not e.deferred()
    e.synchronize()
    assert np.array_equal(block.to_host(), want)
    # the call's own block
    again = e.host_array((2, n))
    keep_back(block, again)
    block.free()
    assert not e.deferred()
    e.synchronize()
    assert np.array_equal(again, want)
    e.close()



def test_compat_entry_device_and_cache(tmp_path):
    """The same-signature entry picks its GPU from LBL_DEVICE / the launcher's local rank,
    re-reads a database file that changed under the same path (the reference re-reads it on
    every call, absorption.c:44-73) and keeps a bounded number of molecules in HBM."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, sys, time
        import numpy as np
        from ctypes import byref, c_int32
        sys.path.insert(0, os.environ["LBL_TEST_ROOT"])
        from pylbl_amd import engine, synthetic
        from pylbl_amd.database import write_database
        lib = engine.library()
        device, resident = c_int32(-2), c_int32(-2)
        lib.lbl_compat_state(byref(device), byref(resident))
        assert (device.value, resident.value) == (-1, 0)
        path = os.path.join(os.environ["LBL_TEST_TMP"], "lines.db")
        formulas = ("H2O", "CO2", "O3")
        def write(seed):
            tables = [synthetic.line_table(f, 1., 90., num_lines=60, seed=seed + i,
                                           tips_range=(150, 400)) for i, f in enumerate(formulas)]
            write_database(path, tables)
            return tables
        def call(formula):
            k = np.full(600, 3.)
            rc = lib.absorption(9e4, 280., 1e-3, 1, 61, 10, k.ctypes.data, path.encode(),
                                formula.encode(), 25, 0)
            assert rc == 0
            return k
        write(100)
        first = call("H2O")
        lib.lbl_compat_state(byref(device), byref(resident))
        assert device.value == int(os.environ["EXPECT_DEVICE"]), device.value
        assert resident.value == 1
        assert np.array_equal(call("H2O"), first)
        call("CO2"); call("O3")
        lib.lbl_compat_state(byref(device), byref(resident))
        assert resident.value == 2, resident.value          # LBL_COMPAT_CACHE=2
        assert np.array_equal(call("H2O"), first)          # evicted, read again: same answer
        os.remove(path)
        time.sleep(0.01)
        write(200)                                          # same path, other lines
        changed = call("H2O")
        assert changed.any() and not np.array_equal(changed, first)
        print("compat ok")
    """)
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    visible = max(torch.cuda.device_count(), 1)         # counts devices without touching them
    for extra, expect in (({"LBL_DEVICE": "0"}, 0), ({"LOCAL_RANK": "5"}, 5 % visible), ({}, 0)):
        env = {k: v for k, v in os.environ.items() if k not in ("LBL_DEVICE", "LOCAL_RANK")}
        env.update(extra)
        env.update({"LBL_COMPAT_CACHE": "2", "EXPECT_DEVICE": str(expect), "LBL_TEST_ROOT": root,
                    "LBL_TEST_TMP": str(tmp_path)})
        result = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                                env=env, timeout=300)
        assert "compat ok" in result.stdout, result.stdout[-2000:] + result.stderr[-4000:]
    # An index beyond the visible devices fails loudly (rc 1), it does not fall back.
    env = dict(os.environ, LBL_DEVICE="99", LBL_TEST_ROOT=root, LBL_TEST_TMP=str(tmp_path))
    bad = textwrap.dedent("""
        import os, sys
        import numpy as np
        sys.path.insert(0, os.environ["LBL_TEST_ROOT"])
        from pylbl_amd import engine
        lib = engine.library()
        k = np.zeros(600)
        path = os.path.join(os.environ["LBL_TEST_TMP"], "lines.db")
        print("rc", lib.absorption(9e4, 280., 1e-3, 1, 61, 10, k.ctypes.data, path.encode(),
                                   b"H2O", 25, 0))
    """)
    result = subprocess.run([sys.executable, "-c", bad], capture_output=True, text=True, env=env,
                            timeout=300)
    assert "rc 1" in result.stdout, result.stdout + result.stderr[-2000:]


def test_grid_edited_in_place_is_uploaded_again(continuum_oracle):
    """The device copy of the spectral grid is cached per array object; editing that array in
    place between two calls must not leave the continuum on the old wavenumbers."""
    from pylbl_amd import mt_ckd
    continuum = mt_ckd.CarbonDioxideContinuum(device=0)
    vmr = {"H2O": 6.6e-3, "CO2": 3.6e-4, "O2": 0.209, "N2": 0.78}
    grid = np.arange(500., 900., 0.01)
    first = np.array(continuum.spectra(288.99, 98388., vmr, grid))
    grid += 700.                                            # same object, new contents
    second = continuum.spectra(288.99, 98388., vmr, grid)
    expect = continuum_oracle.continuum("CO2").spectra(288.99, 98388., vmr, grid)
    np.testing.assert_allclose(second, expect, rtol=1e-6, atol=1e-300)
    assert not np.array_equal(first, second)


@pytest.mark.parametrize("npv,span", [(100, 400), (1000, 60), (2000, 40)])
def test_farfield_truncation_bound_per_resolution(oracle, npv, span):
    """Spectroscopy sums distant lines through the far-field series by default (per-call flag
    LBL_FARFIELD).  Its design bound -- |u|/|a| <= 1/4, 21 terms: truncation <= ~1.5e-11
    relative -- is asserted here against the direct kernel at the resolutions of BASELINE
    configs[1], the target and configs[4], at 1 atm and 10 Pa, and the result meets the 1e-6 bar
    against the oracle."""
    from pylbl_amd.engine import Engine
    e = Engine(0)
    v0 = 2000
    vn = v0 + span
    table = synthetic.line_table("CO2", v0 - 25., vn + 25., num_lines=80*(span + 50), seed=17)
    h = e.load(table)
    for t, p in ((288.99, 98388.), (232.7, 10.)):
        direct = e.compute(h, t, p, 3.6e-4, v0, vn, npv)[0]
        series = e.compute(h, t, p, 3.6e-4, v0, vn, npv, farfield=True)[0]
        assert not np.array_equal(series, direct)           # the flag does change the path
        assert np.max(np.abs(series - direct)/direct) < 1.e-10, (npv, p)
    k_ref, _ = oracle.absorption_port(table, 232.7, 10., 3.6e-4, v0, vn, npv)
    np.testing.assert_allclose(series, k_ref, rtol=1.e-6)
    e.close()


def test_spectroscopy_farfield_flag_and_block_pool(small_database):
    """Spectroscopy(farfield=True), the default, differs from farfield=False only at the series'
    truncation level; the [levels, n] blocks in HBM are recycled between calls."""
    from pylbl_amd import MemoryDatabase, Spectroscopy
    tables = [synthetic.line_table("H2O", 580., 700., num_lines=4000, seed=5),
              synthetic.line_table("CO2", 580., 700., num_lines=9000, seed=6)]
    full = synthetic.fixture_atmosphere()
    atmos = synthetic.Atmos(p=full.p, t=full.t, vmr={k: full.vmr[k] for k in ("H2O", "CO2")})
    grid = np.arange(606., 670., 0.001)
    results = {}
    for farfield in (True, False):
        spec = Spectroscopy(atmos, grid, MemoryDatabase(tables), continua_backend=None,
                            cross_sections_backend=None, farfield=farfield)
        assert spec.farfield is farfield
        for fmt in ("all", "gas", "total"):
            results[(farfield, fmt)] = spec.compute_absorption(fmt, remove_pedestal=True)
    for fmt, names in (("all", ("H2O_absorption", "CO2_absorption")),
                       ("gas", ("H2O_absorption", "CO2_absorption")), ("total", ("absorption",))):
        for name in names:
            series, direct = results[(True, fmt)][name], results[(False, fmt)][name]
            assert series.shape == direct.shape and not np.array_equal(series, direct)
            flat = direct.reshape(direct.shape[0], -1)          # per level, every slot
            scale = np.max(np.abs(flat), axis=1, keepdims=True)
            assert np.max(np.abs(series.reshape(flat.shape) - flat)/scale) < 1.e-9, (fmt, name)
    engine = spec._molecule("CO2").gas.engine
    idle = sum(len(blocks) for blocks in engine.blocks.idle.values())
    assert idle >= 1
    before = engine.blocks.idle_bytes
    spec.compute_absorption("total")
    assert engine.blocks.idle_bytes == before          # taken from the pool, handed back


@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_spectroscopy_slots_stay_ordered_when_lines_calls_take_turns(remove_pedestal):
    """Lines calls with the far-field series rotate over two lanes (with a pedestal pass: four)
    while the continuum kernels of the same gas add into the same block from another stream: the
    sums are the same whether the series is on or not (to its truncation), and the same bits call
    after call."""
    import os
    from pylbl_amd import MemoryDatabase, Spectroscopy
    os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(__file__), "golden",
                                                       "mt_ckd_bands.npz"))
    tables = [synthetic.line_table("H2O", 580., 700., num_lines=4000, seed=15),
              synthetic.line_table("CO2", 580., 700., num_lines=9000, seed=16),
              synthetic.line_table("N2", 580., 700., num_lines=200, seed=17),
              synthetic.line_table("O2", 580., 700., num_lines=300, seed=18)]
    full = synthetic.fixture_atmosphere()
    atmos = synthetic.Atmos(p=full.p, t=full.t,
                            vmr={k: full.vmr[k] for k in ("H2O", "CO2", "N2", "O2")})
    grid = np.arange(606., 670., 0.001)
    results = {}
    for farfield in (True, False):
        spec = Spectroscopy(atmos, grid, MemoryDatabase(tables), cross_sections_backend=None,
                            farfield=farfield)
        for fmt in ("total", "gas", "all"):
            first = spec.compute_absorption(fmt, remove_pedestal=remove_pedestal)
            for _ in range(3):
                again = spec.compute_absorption(fmt, remove_pedestal=remove_pedestal)
                for name in first:
                    if name.endswith("absorption"):
                        assert np.array_equal(np.asarray(first[name]), np.asarray(again[name])), \
                            (fmt, name)
            results[(farfield, fmt)] = first
    for fmt in ("total", "gas", "all"):
        for name in results[(True, fmt)]:
            if not name.endswith("absorption"):
                continue
            series = np.asarray(results[(True, fmt)][name])
            direct = np.asarray(results[(False, fmt)][name])
            flat = direct.reshape(direct.shape[0], -1)
            scale = np.max(np.abs(flat), axis=1, keepdims=True)
            assert np.max(np.abs(series.reshape(flat.shape) - flat)/scale) < 1.e-9, (fmt, name)


def test_gas_output_does_not_depend_on_who_delivers():
    """ "gas": every gas's lines call delivers its own block piece by piece (default), or only the
    last gas does and the others' blocks travel in one copy each: the same sums (the continuum is
    added before or after the lines: rounding only), and "gas" equals "all" summed over the
    mechanisms either way (spectroscopy.py:213-221)."""
    import os
    from pylbl_amd import MemoryDatabase, Spectroscopy
    os.environ.setdefault("PYLBL_MT_CKD", os.path.join(os.path.dirname(__file__), "golden",
                                                       "mt_ckd_bands.npz"))
    tables = [synthetic.line_table("H2O", 580., 700., num_lines=4000, seed=15),
              synthetic.line_table("CO2", 580., 700., num_lines=9000, seed=16),
              synthetic.line_table("N2", 580., 700., num_lines=200, seed=17),
              synthetic.line_table("O2", 580., 700., num_lines=300, seed=18)]
    full = synthetic.standard_atmosphere(5)
    atmos = synthetic.Atmos(p=full.p, t=full.t,
                            vmr={k: full.vmr[k] for k in ("H2O", "CO2", "N2", "O2")})
    grid = np.arange(606., 670., 0.001)
    spec = Spectroscopy(atmos, grid, MemoryDatabase(tables), cross_sections_backend=None)
    assert spec.gas_delivery == "each"
    each = {k: np.array(v) for k, v in spec.compute_absorption("gas").items()
            if k.endswith("absorption")}
    spec.gas_delivery = "last"
    last = {k: np.array(v) for k, v in spec.compute_absorption("gas").items()
            if k.endswith("absorption")}
    parts = {k: np.asarray(v).sum(axis=-2) for k, v in spec.compute_absorption("all").items()
             if k.endswith("absorption")}
    assert set(each) == set(last) == set(parts) and len(each) == 4
    for name in each:
        scale = np.max(np.abs(parts[name]), axis=-1, keepdims=True)
        assert np.max(np.abs(each[name] - last[name])/scale) < 1.e-13, name
        assert np.max(np.abs(each[name] - parts[name])/scale) < 1.e-13, name


@pytest.mark.parametrize("remove_pedestal", [False, True])
def test_streamed_call_with_empty_runs_of_tiles(remove_pedestal):
    """Four tiles, every line of the table beyond the end of the grid (within the cut-off of its
    last tile only): the runs of tiles come out as [0, 3), [3, 4) and two EMPTY ones -- which used
    to move the end of the run before them to zero, so that with a pedestal its columns were
    neither given it nor delivered (found by the Spectroscopy fuzz once every gas delivered its
    block run by run; the bounds of run i are entries i and i+1 of one array)."""
    from pylbl_amd.engine import DeviceSpectra, Engine
    e = Engine(0)
    table = synthetic.line_table("CH4", 975., 984., num_lines=46, seed=3, tips_range=(150, 400))
    atmos = synthetic.standard_atmosphere(3)
    v0, vn, npv = 909, 959, 10
    n = (vn - v0)*npv
    h = e.load(table)
    x = atmos.vmr["CH4"]
    for farfield in (True, False):
        expect = e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal,
                           scale_density=True, farfield=farfield)
        assert expect[:, 400:].any()                    # something to lose in the last tile
        for pieces in (2, 4, 8):
            out = DeviceSpectra(e, 3, n)
            target = e.host_array((3, n - 3))
            target[...] = -1.
            e.compute(h, atmos.t, atmos.p, x, v0, vn, npv, remove_pedestal=remove_pedestal,
                      out=out, scale_density=True, farfield=farfield, asynchronous=True,
                      deliver=target, pieces=pieces)
            e.synchronize()
            assert np.array_equal(target, expect[:, :n - 3]), (farfield, pieces)
            assert np.array_equal(out.to_host(), expect), (farfield, pieces)
            out.free()
    e.free(h)
    e.close()
