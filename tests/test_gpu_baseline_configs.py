"""Every BASELINE.json workload at its full size, every grid point, against the CPU oracle.

The oracle (oracle/lbl_oracle.c, pinned bit-for-bit to the reference's compiled C by
tests/test_oracle_golden.py) does ~1.2e9 evaluations per second and core, so whole 5 M- and
10 M-point spectra are affordable when the (molecule, level, mode) calls are farmed out over
the host cores (tests/oracle_farm.py).  All jobs are queued when the first test of this file
starts and run beside the GPU work of the others.

What is compared (labels contain "baseline"; none of them may need the conditioning
allowance for the pedestal, see test_gpu_zz_tolerance_report.py):

  configs[0]  CO2, 500-800 @ 0.1: the bench's in-range table AND a 1-5000 table whose first
              row lies below 474 cm-1, so the reference's range `break` (absorption.c:80-83)
              discards everything; both range policies; pedestal off/on.
  configs[1]  H2O + CO2, 1-5000 @ 0.01: all 500 k points, pedestal off/on.
  target      H2O + CO2, 1-5000 @ 0.001 (what bench.py times): all 5 M points, pedestal
              off AND on (the default through compute_absorption, spectroscopy.py:163-164).
  configs[2]  the other six README molecules at 5 M points: all points without pedestal,
              O3 / CO / O2 / N2 also with.
  configs[3]  shape: 8 standard-atmosphere levels, H2O + CO2 + O3, 1-3000 @ 0.001, one batched
              call; the surface and the 10 Pa level compared at all 3 M points with pedestal on
              and off, the levels between through windows.  And as the 8-GPU job cuts it: rank 0's
              and rank 7's 8-level blocks of the 64-level atmosphere through
              ShardedLines.for_engine, levels 0 and 63 at all points, pedestal off and on.
  configs[4]  shape: 1-5000 @ 0.0005 (n_per_v = 2000, 10 M points), 4 levels of the 256-level
              atmosphere; CO / O2 / N2 at all points with pedestal on and off at two levels; all
              eight molecules through windows at all four levels; H2O, CO2, O3, N2O and CH4 at
              all 10 M points, pedestal on and off, at one stratospheric level (level 170,
              ~1 hPa); and ONE GPU'S SHARE of the 8-GPU job -- 32 levels x 8 molecules summed on
              the device through ShardedLines.for_engine -- through windows against the
              composition of the oracle's spectra; and another rank's share WITH the pedestal
              removed (the default through compute_absorption), one level at all 10 M points
              against the sum of the oracle's eight spectra.
  banded      a banded CO2 table (split tiles, long pedestal chain) at 5 M points, pedestal on.
  HITRAN-shaped  a 150 000-line table with the corner values real tables have (zero half-widths,
              n_air <= 0, elower = -1, positions down to 1e-4 cm-1, twelve isotopologues:
              tests/hitran_shapes.py) on 0-5000 @ 0.001, all 5 M points, pedestal off and on,
              direct and with the far-field series.
  far-field   configs[1] and the target again with the far-field series (what Spectroscopy runs
              by default), every point, pedestal on and off, same 1e-6 bar.
  end to end  Spectroscopy.compute_absorption() with default arguments at 5 M points, one
              level, H2O + CO2 + a cross-section-only gas: the three mechanism slots against the
              three oracles.
  default path at the big shapes (round 5)  what Spectroscopy runs by default -- far-field
              series AND pedestal removed -- at every point of configs[4]'s grid (n_per_v = 2000,
              10 M points) for H2O / CO2 / O3 at levels 170 (~1 hPa) and 255 (0.1 hPa); through
              ShardedLines.for_engine(farfield=True) for rank 0's and rank 7's blocks of
              configs[3] (levels 0 and 63 at all 3 M points) and rank 5's share of configs[4]
              (32 levels x 8 molecules summed on the device, level 170 at all 10 M points); and
              the CO2 table at 5 M points with cut_off 5 and 40 (spectra.c:48-62 with another
              window, voigt.c:17-27,79-83), pedestal off and on, direct and far-field.
"""
import warnings

import numpy as np
import pytest

from pylbl_amd import synthetic
from tests import golden_io
from tests.oracle_farm import OracleFarm, table_from_recipe
from tests.test_gpu_parity import assert_spectrum

pytestmark = pytest.mark.gpu

SURFACE = synthetic.surface_level()
OTHERS = ("O3", "N2O", "CO", "CH4", "O2", "N2")
PEDESTAL_TOO = ("O3", "CO", "O2", "N2")
STANDARD8 = synthetic.standard_atmosphere(8)
STANDARD256 = synthetic.standard_atmosphere(256)
LEVELS256 = (0, 85, 170, 255)
SMALL = ("CO", "O2", "N2")
LARGE = ("H2O", "CO2", "O3", "N2O", "CH4")
FULL_LEVEL = 170
BANDED = ("banded", "CO2", 1., 5000., 300_000, 8, 41)
HITRAN_SHAPED = ("hitran", 1.e-4, 5026., 150_000, 8101)
HITRAN_LEVEL = (250., 5.e4, 4.e-4)
OTHER_CUTS = (5, 40)


def uniform(formula, v_lo=1., v_hi=5000.):
    return ("uniform", formula, v_lo, v_hi)


def level_of(atmos, formula, level):
    return float(atmos.t[level]), float(atmos.p[level]), float(atmos.vmr[formula][level])


@pytest.fixture(scope="module")
def farm():
    f = OracleFarm()
    # configs[0]
    t, p, x = level_of(SURFACE, "CO2", 0)
    for ped in (False, True):
        f.submit(("c0", "inrange", ped), uniform("CO2", 500., 800.), t, p, x, 500, 801, 10, ped)
        f.submit(("c0", "wide", ped), uniform("CO2"), t, p, x, 500, 801, 10, ped)
    # configs[1] and the target
    for formula in ("H2O", "CO2"):
        t, p, x = level_of(SURFACE, formula, 0)
        for ped in (False, True):
            f.submit(("c1", formula, ped), uniform(formula), t, p, x, 1, 5001, 100, ped)
            f.submit(("target", formula, ped), uniform(formula), t, p, x, 1, 5001, 1000, ped)
    # configs[2]: the other six molecules
    for formula in OTHERS:
        t, p, x = level_of(SURFACE, formula, 0)
        f.submit(("c2", formula, False), uniform(formula), t, p, x, 1, 5001, 1000, False)
        if formula in PEDESTAL_TOO:
            f.submit(("c2", formula, True), uniform(formula), t, p, x, 1, 5001, 1000, True)
    # configs[3] shape: first and last of 8 standard-atmosphere levels
    for formula in ("H2O", "CO2", "O3"):
        for level in (0, 7):
            t, p, x = level_of(STANDARD8, formula, level)
            for ped in (False, True):
                f.submit(("c3", formula, level, ped), uniform(formula, 1., 3000.), t, p, x,
                         1, 3001, 1000, ped)
    # configs[4] shape: 10 M points
    for formula in SMALL:
        for level in (LEVELS256[0], LEVELS256[-1]):
            t, p, x = level_of(STANDARD256, formula, level)
            for ped in (False, True):
                f.submit(("c4", formula, level, ped), uniform(formula), t, p, x, 1, 5001, 2000,
                         ped)
    # (all eight at FULL_LEVEL: the composition one GPU's share is compared with, pedestal removed)
    for formula in LARGE + SMALL:
        t, p, x = level_of(STANDARD256, formula, FULL_LEVEL)
        for ped in (False, True):
            f.submit(("c4", formula, FULL_LEVEL, ped), uniform(formula), t, p, x, 1, 5001, 2000,
                     ped)
    # the default user path at configs[4]'s top level too (H2O / CO2 / O3; CO / O2 / N2 above)
    for formula in ("H2O", "CO2", "O3"):
        t, p, x = level_of(STANDARD256, formula, LEVELS256[-1])
        for ped in (False, True):
            f.submit(("c4", formula, LEVELS256[-1], ped), uniform(formula), t, p, x, 1, 5001,
                     2000, ped)
    # other cut-offs than the 25 every caller of the reference gets (gas_optics.py:46-47)
    t, p, x = level_of(SURFACE, "CO2", 0)
    for cut in OTHER_CUTS:
        for ped in (False, True):
            f.submit(("cut", cut, ped), uniform("CO2"), t, p, x, 1, 5001, 1000, ped, cut=cut)
    # banded table, pedestal on (+ the plain spectrum its tolerance refers to)
    t, p, x = level_of(SURFACE, "CO2", 0)
    for ped in (False, True):
        f.submit(("banded", ped), BANDED, t, p, x, 1, 5001, 1000, ped)
    # HITRAN-shaped values at the target's size
    for ped in (False, True):
        f.submit(("hitran", ped), HITRAN_SHAPED, *HITRAN_LEVEL, 0, 5000, 1000, ped)
    f.start()
    yield f
    f.close()


@pytest.fixture(scope="module")
def engine():
    from pylbl_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def case_for(v0, vn, npv, ped, index=0, cut=25):
    return golden_io.Case("baseline", index, 0, 0, 0, v0, vn, npv, cut, ped, None, 0)


def check_full(farm, key_plain, key, k, v0, vn, npv, ped, label, evals=None, cut=25):
    k_ref, evals_ref = farm.result(key)
    if evals is not None:
        assert evals == evals_ref, f"{label}: eval count {evals} != oracle {evals_ref}"
    k_plain = farm.result(key_plain)[0] if ped else None
    assert_spectrum(k, k_ref, case_for(v0, vn, npv, ped, cut=cut), f"baseline {label}", k_plain)


def check_windows(oracle, table, k, t, p, x, v0, vn, npv, starts, label, width=2):
    """Pedestal-free windows: a point only sees lines within cut_off + 1 cm-1, so the oracle on
    a narrow sub-grid reproduces the matching slice of the full spectrum."""
    for lo in starts:
        hi = lo + width
        g0, g1 = max(lo - 2, v0), min(hi + 2, vn)
        near = table.subset((table.nu >= g0 - 26.) & (table.nu <= g1 + 26.))
        k_ref, _ = oracle.absorption_port(near, t, p, x, g0, g1, npv)
        k_ref = k_ref[(lo - g0)*npv:(hi - g0)*npv]
        piece = k[(lo - v0)*npv:(hi - v0)*npv]
        np.testing.assert_allclose(piece, k_ref, rtol=1e-6, err_msg=f"baseline {label} @{lo}")


def test_config0_co2_500_800(farm, engine):
    """configs[0] through the reference-shaped Gas object, incl. the range `break` quirk."""
    from pylbl_amd import Gas
    grid = np.arange(500., 800., 0.1)
    assert synthetic.grid_arguments(grid) == (500, 801, 10)
    t, p, x = level_of(SURFACE, "CO2", 0)
    inrange = table_from_recipe(uniform("CO2", 500., 800.))
    wide = table_from_recipe(uniform("CO2"))
    assert wide.nu[0] < 474.            # absorption.c:80-83 fires on the very first row
    for ped in (False, True):
        gas = Gas(inrange, "CO2", engine=engine)
        k = gas.absorption_coefficient(t, p, x, grid, remove_pedestal=ped)
        assert k.size == 3010 and k[:grid.size].any()
        check_full(farm, ("c0", "inrange", False), ("c0", "inrange", ped), k, 500, 801, 10, ped,
                   f"config0 in-range table ped={ped}")
        gas = Gas(wide, "CO2", engine=engine)
        with pytest.warns(RuntimeWarning, match="first transition"):
            k = gas.absorption_coefficient(t, p, x, grid, remove_pedestal=ped)
        k_ref, _ = farm.result(("c0", "wide", ped))
        assert not k_ref.any() and not k.any()     # the reference computes nothing here
        # "skip" policy = the reference on the rows inside its own range.
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            k = gas.absorption_coefficient(t, p, x, grid, remove_pedestal=ped,
                                           range_policy="skip")
        inside = wide.subset((wide.nu >= 500 - 26.) & (wide.nu <= 801 + 26.))
        from oracle import oracle
        k_ref, _ = oracle.absorption_port(inside, t, p, x, 500, 801, 10, remove_pedestal=ped)
        k_plain, _ = oracle.absorption_port(inside, t, p, x, 500, 801, 10)
        assert_spectrum(k, k_ref, case_for(500, 801, 10, ped), f"baseline config0 skip ped={ped}",
                        k_plain)


@pytest.mark.parametrize("name,npv", [("c1", 100), ("target", 1000)])
def test_h2o_co2_whole_grid(farm, engine, name, npv):
    """configs[1] (0.01 cm-1) and the target (0.001 cm-1): every point, both modes."""
    for formula in ("H2O", "CO2"):
        t, p, x = level_of(SURFACE, formula, 0)
        handle = engine.load(table_from_recipe(uniform(formula)))
        for ped in (False, True):
            k, evals = engine.compute(handle, t, p, x, 1, 5001, npv, remove_pedestal=ped,
                                      want_evals=True)
            check_full(farm, (name, formula, False), (name, formula, ped), k[0], 1, 5001, npv,
                       ped, f"{name} {formula} ped={ped}", evals)
        engine.free(handle)


def test_config2_other_molecules(farm, engine):
    for formula in OTHERS:
        t, p, x = level_of(SURFACE, formula, 0)
        handle = engine.load(table_from_recipe(uniform(formula)))
        for ped in (False, True):
            if ped and formula not in PEDESTAL_TOO:
                continue
            k, evals = engine.compute(handle, t, p, x, 1, 5001, 1000, remove_pedestal=ped,
                                      want_evals=True)
            check_full(farm, ("c2", formula, False), ("c2", formula, ped), k[0], 1, 5001, 1000,
                       ped, f"config2 {formula} ped={ped}", evals)
        engine.free(handle)


def test_config3_shape_standard_atmosphere(farm, engine, oracle):
    """8 levels from 1013 hPa to 0.1 hPa in ONE batched call per molecule."""
    for formula in ("H2O", "CO2", "O3"):
        table = table_from_recipe(uniform(formula, 1., 3000.))
        handle = engine.load(table)
        spectra = {}
        for ped in (False, True):
            spectra[ped] = engine.compute(handle, STANDARD8.t, STANDARD8.p,
                                          STANDARD8.vmr[formula], 1, 3001, 1000,
                                          remove_pedestal=ped)
            for level in (0, 7):
                check_full(farm, ("c3", formula, level, False), ("c3", formula, level, ped),
                           spectra[ped][level], 1, 3001, 1000, ped,
                           f"config3 {formula} level {level} ped={ped}")
        for level in range(1, 7):
            t, p, x = level_of(STANDARD8, formula, level)
            check_windows(oracle, table, spectra[False][level], t, p, x, 1, 3001, 1000,
                          (1, 700 + 37*level, 2998), f"config3 {formula} level {level}", width=1)
        engine.free(handle)


def test_config4_shape_ten_million_points(farm, engine, oracle):
    """n_per_v = 2000: windows of 102 001 points, tiles of 512.  All eight molecules."""
    t4, p4 = STANDARD256.t[list(LEVELS256)], STANDARD256.p[list(LEVELS256)]
    for formula in SMALL + LARGE:
        table = table_from_recipe(uniform(formula))
        x4 = STANDARD256.vmr[formula][list(LEVELS256)]
        handle = engine.load(table)
        modes = (False, True) if formula in SMALL else (False,)
        for ped in modes:
            k = engine.compute(handle, t4, p4, x4, 1, 5001, 2000, remove_pedestal=ped)
            assert k.shape == (4, 10_000_000)
            if formula in SMALL:
                for row, level in ((0, LEVELS256[0]), (3, LEVELS256[-1])):
                    check_full(farm, ("c4", formula, level, False), ("c4", formula, level, ped),
                               k[row], 1, 5001, 2000, ped,
                               f"config4 {formula} level {level} ped={ped}")
            if not ped:
                for row in range(4):
                    check_windows(oracle, table, k[row], t4[row], p4[row], x4[row], 1, 5001,
                                  2000, (1, 1200 + 301*row, 4999),
                                  f"config4 {formula} level {LEVELS256[row]}", width=1)
        engine.free(handle)


@pytest.mark.parametrize("formula", LARGE)
def test_config4_large_molecules_every_point(farm, engine, formula):
    """The five molecules with 1e5 - 4e5 lines at n_per_v = 2000, all 10 M points, pedestal off
    and on, at a ~1 hPa level (narrow cores: the CPF12 rows of the general path)."""
    t, p, x = level_of(STANDARD256, formula, FULL_LEVEL)
    handle = engine.load(table_from_recipe(uniform(formula)))
    for ped in (False, True):
        k, evals = engine.compute(handle, t, p, x, 1, 5001, 2000, remove_pedestal=ped,
                                  want_evals=True)
        check_full(farm, ("c4", formula, FULL_LEVEL, False), ("c4", formula, FULL_LEVEL, ped),
                   k[0], 1, 5001, 2000, ped, f"config4 {formula} level {FULL_LEVEL} ped={ped}",
                   evals)
    engine.free(handle)


def test_config4_one_gpus_share_through_the_sharded_path(engine, oracle):
    """configs[4] on 8 GPUs gives each 32 levels x 8 molecules x 10 M points; with
    output="total" the sum over gases is formed on the device (2.56 GB instead of 20.5 GB per
    GPU).  Exactly that call -- pylbl_amd.distributed.ShardedLines.for_engine, what
    bench.py --gpus 8 --config 4 --levels-per-gpu 32 --output total runs on every rank -- on
    rank 3's block of levels (96..127), windows of several levels against the oracle's
    n x k summed over the eight molecules."""
    from pylbl_amd import distributed, number_density
    formulas = SMALL + LARGE
    mine = distributed.level_shard(256, 3, 8)
    assert (mine.start, mine.stop) == (96, 128)
    t, p = STANDARD256.t[mine], STANDARD256.p[mine]
    vmr = {f: STANDARD256.vmr[f][mine] for f in formulas}
    tables = {f: table_from_recipe(uniform(f)) for f in formulas}
    handles = {f: engine.load(tables[f]) for f in formulas}
    sharded = distributed.ShardedLines.for_engine(
        engine, handles, (1, 5001, 2000), remove_pedestal=False, scale_density=True,
        weights=[tables[f].num_lines for f in formulas])
    total = sharded.run(t, p, vmr, output="total")
    assert tuple(total.shape) == (32, 10_000_000)
    npv = 2000
    for row in (0, 13, 31):
        for lo in (1, 667 + 97*row, 2349, 4999):
            g0, g1 = max(lo - 2, 1), min(lo + 3, 5001)
            expect = np.zeros(npv)
            for f in formulas:
                near = tables[f].subset((tables[f].nu >= g0 - 26.) & (tables[f].nu <= g1 + 26.))
                k_ref, _ = oracle.absorption_port(near, t[row], p[row], vmr[f][row], g0, g1, npv)
                expect += number_density(t[row], p[row], vmr[f][row]) * \
                    k_ref[(lo - g0)*npv:(lo + 1 - g0)*npv]
            got = total[row, (lo - 1)*npv:lo*npv].cpu().numpy()
            np.testing.assert_allclose(got, expect, rtol=1e-6,
                                       err_msg=f"baseline config4 share level {96 + row} @{lo}")
    for handle in handles.values():
        engine.free(handle)


@pytest.mark.parametrize("farfield", [False, True])
def test_config4_one_gpus_share_with_the_pedestal_removed(farm, engine, farfield):
    """(farfield=True: with the far-field series too, Spectroscopy's other default.)
    The same call the way Spectroscopy.compute_absorption makes it by default
    (spectroscopy.py:163-164: the pedestal is removed when the continuum is MT-CKD): rank 5's
    block of the 8-GPU job (levels 160..191 x 8 molecules x 10 M points), n k summed over the
    gases on the device with remove_pedestal=True -- every call but the first goes through the
    un-pedestalled scratch block and pedestal_apply_kernel's adding form, on rotating lanes.  The
    row of level 170 against the sum of the oracle's eight whole-grid spectra with the pedestal
    removed, all 10 M points; the tolerance is the sum of the per-gas tolerances of
    assert_spectrum."""
    from pylbl_amd import distributed, number_density
    formulas = SMALL + LARGE
    plan = distributed.partition(256, [1.]*len(formulas), 8)
    mine = distributed.level_shard(256, 5, 8)
    assert plan.mode == "levels" and plan.levels_of(5) == list(range(mine.start, mine.stop))
    assert mine.start <= FULL_LEVEL < mine.stop
    t, p = STANDARD256.t[mine], STANDARD256.p[mine]
    vmr = {f: STANDARD256.vmr[f][mine] for f in formulas}
    tables = {f: table_from_recipe(uniform(f)) for f in formulas}
    handles = {f: engine.load(tables[f]) for f in formulas}
    sharded = distributed.ShardedLines.for_engine(
        engine, handles, (1, 5001, 2000), remove_pedestal=True, scale_density=True,
        weights=[tables[f].num_lines for f in formulas], farfield=farfield)
    total = sharded.run(t, p, vmr, output="total")
    assert tuple(total.shape) == (32, 10_000_000)
    row = FULL_LEVEL - mine.start
    got = total[row].cpu().numpy()
    expect = np.zeros(got.size)
    tolerance = np.zeros(got.size)
    for f in formulas:
        tt, pp, x = level_of(STANDARD256, f, FULL_LEVEL)
        density = number_density(tt, pp, x)
        k_ref = farm.result(("c4", f, FULL_LEVEL, True))[0]
        k_plain = farm.result(("c4", f, FULL_LEVEL, False))[0]
        expect += density*k_ref
        tol = np.maximum(golden_io.pedestal_tolerance(k_ref, 2000, 25, 1.e-6), 1.e-6*np.abs(k_plain))
        tol = np.maximum(tol, golden_io.pedestal_tolerance(k_plain, 2000, 25, 1.e-13))
        tolerance += density*tol
    worst = float(np.max(np.abs(got - expect)/(tolerance + 1e-300)))
    assert worst <= 1., (f"baseline config4 share with pedestal, farfield={farfield}, level "
                         f"{FULL_LEVEL}: {worst:.3g} x")
    # The other rows: finite, and the neighbours of level 170 differ from it (every level its own).
    for other in (row - 1, row + 1, 0, 31):
        values = total[other, ::1000].cpu().numpy()
        assert np.all(np.isfinite(values)) and not np.array_equal(values, got[::1000])
    for handle in handles.values():
        engine.free(handle)


@pytest.mark.parametrize("farfield", [False, True])
def test_config3_blocks_of_the_64_level_job(farm, engine, farfield):
    """(farfield=True: the series on, as Spectroscopy runs it by default.)
    configs[3] as the 8-GPU job cuts it: 64 standard-atmosphere levels, H2O + CO2 + O3,
    1-3000 @ 0.001, 8 levels per rank.  Rank 0's and rank 7's blocks (1013 hPa ... and ... 0.1 hPa,
    where the inner Voigt regions carry most) through ShardedLines.for_engine, per gas, pedestal
    off and on; level 0 and level 63 -- which are the first and last of the 8-level atmosphere the
    farm has whole-grid spectra for -- at all 3 M points."""
    from pylbl_amd import distributed
    formulas = ("H2O", "CO2", "O3")
    standard64 = synthetic.standard_atmosphere(64)
    for f in formulas:
        for of64, of8 in ((0, 0), (63, 7)):
            assert level_of(standard64, f, of64) == level_of(STANDARD8, f, of8)
    tables = {f: table_from_recipe(uniform(f, 1., 3000.)) for f in formulas}
    handles = {f: engine.load(tables[f]) for f in formulas}
    weights = [tables[f].num_lines for f in formulas]
    plan = distributed.partition(64, weights, 8)
    for ped in (False, True):
        sharded = distributed.ShardedLines.for_engine(engine, handles, (1, 3001, 1000),
                                                      remove_pedestal=ped, weights=weights,
                                                      farfield=farfield)
        for rank, row, of8 in ((0, 0, 0), (7, 7, 7)):
            mine = distributed.level_shard(64, rank, 8)
            assert plan.levels_of(rank) == list(range(mine.start, mine.stop)) and \
                mine.stop - mine.start == 8
            vmr = {f: standard64.vmr[f][mine] for f in formulas}
            result = sharded.run(standard64.t[mine], standard64.p[mine], vmr, output="gas")
            for f in formulas:
                assert tuple(result[f].shape) == (8, 3_000_000)
                k = result[f][row].cpu().numpy()
                check_full(farm, ("c3", f, of8, False), ("c3", f, of8, ped), k, 1, 3001, 1000,
                           ped, f"config3 rank {rank} of 8, {f} level {mine.start + row} "
                                f"ped={ped} farfield={farfield}")
    for handle in handles.values():
        engine.free(handle)


@pytest.mark.parametrize("name,npv", [("c1", 100), ("target", 1000)])
def test_h2o_co2_whole_grid_farfield_series(farm, engine, name, npv):
    """What Spectroscopy runs unless told otherwise (farfield=True): lines far from a tile enter
    through the tile's power series.  Every point of configs[1] and of the target, pedestal off
    and on, against the oracle at the same bar as the direct kernel."""
    for formula in ("H2O", "CO2"):
        t, p, x = level_of(SURFACE, formula, 0)
        handle = engine.load(table_from_recipe(uniform(formula)))
        for ped in (False, True):
            k, evals = engine.compute(handle, t, p, x, 1, 5001, npv, remove_pedestal=ped,
                                      want_evals=True, farfield=True)
            check_full(farm, (name, formula, False), (name, formula, ped), k[0], 1, 5001, npv,
                       ped, f"{name} {formula} far-field ped={ped}", evals)
        engine.free(handle)


def test_spectroscopy_default_call_end_to_end(farm, engine, continuum_oracle, tmp_path):
    """Spectroscopy(atmosphere, grid, database).compute_absorption() -- every argument at its
    default: output "all", pedestal removed because the continuum is MT-CKD, far-field series
    on -- for the target workload (1 level, H2O + CO2, 1-5000 @ 0.001) plus O2 and N2 (the
    continua need their mole fractions, and a gas of the atmosphere that the database does not
    know is an error in the reference) and a gas that only has a cross-section.  Slot 0 against the farm's whole-grid oracle spectra, slot 1 against the
    MT-CKD oracle, slot 2 against the cross-section oracle, all 4 999 000 points."""
    from oracle import xsec_oracle
    from pylbl_amd import MemoryDatabase, Spectroscopy, arts_crossfit, number_density
    bands = synthetic.cross_section_bands(seed=5)
    arts_crossfit.write_npz(tmp_path / "CFC11.npz", bands)
    with_lines = {"H2O": "target", "CO2": "target", "O2": "c2", "N2": "c2"}
    tables = [table_from_recipe(uniform(f)) for f in with_lines]
    database = MemoryDatabase(tables, cross_sections={"CFC11": str(tmp_path / "CFC11.npz")})
    gases = {"H2O": SURFACE.vmr["H2O"], "CO2": SURFACE.vmr["CO2"], "O2": SURFACE.vmr["O2"],
             "N2": SURFACE.vmr["N2"], "CFC11": np.asarray([2.3e-10])}
    atmosphere = synthetic.Atmos(p=SURFACE.p, t=SURFACE.t, vmr=gases)
    grid = np.arange(1., 5000., 0.001)
    spec = Spectroscopy(atmosphere, grid, database)
    assert spec.farfield and spec.continua_backend == "mt_ckd"
    out = spec.compute_absorption()
    assert list(out["mechanism"]) == ["lines", "continuum", "cross_section"]
    t, p = float(SURFACE.t[0]), float(SURFACE.p[0])
    vmr = {g: float(v[0]) for g, v in gases.items()}
    for formula in gases:
        beta = np.asarray(out[f"{formula}_absorption"])
        assert beta.shape == (1, 3, grid.size)
        n = number_density(t, p, vmr[formula])
        # slot 0
        if formula in with_lines:
            k_ref, _ = farm.result((with_lines[formula], formula, True))
            k_plain, _ = farm.result((with_lines[formula], formula, False))
            assert_spectrum(beta[0, 0], n*k_ref[:grid.size], case_for(1, 5001, 1000, True),
                            f"baseline end-to-end {formula} lines", n*k_plain[:grid.size])
        else:
            assert not beta[0, 0].any()
        # slot 1
        owners = {"H2O": ["H2OForeign", "H2OSelf"], "CFC11": []}.get(formula, [formula])
        expect = np.zeros(grid.size)
        for owner in owners:
            expect = expect + continuum_oracle.continuum(owner).spectra(t, p, vmr, grid)
        scale = np.max(np.abs(expect))
        assert np.all(np.abs(beta[0, 1] - expect) <= 1e-6*np.abs(expect) + 1e-12*scale), formula
        # slot 2
        if formula == "CFC11":
            expect = n*xsec_oracle.absorption_coefficient(bands, grid, t, p)
            scale = np.max(np.abs(expect))
            assert scale > 0.
            assert np.all(np.abs(beta[0, 2] - expect) <= 1e-6*np.abs(expect) + 1e-12*scale)
        else:
            assert not beta[0, 2].any()
    # "gas" and "total" of the same call: sums of the slots above, formed on the device.
    per_gas = spec.compute_absorption("gas")
    total = spec.compute_absorption("total")
    summed = np.zeros((1, grid.size))
    for formula in gases:
        mine = np.asarray(out[f"{formula}_absorption"]).sum(axis=1)
        got = np.asarray(per_gas[f"{formula}_absorption"])
        assert np.all(np.abs(got - mine) <= 1e-12*np.max(np.abs(mine)) + 1e-300), formula
        summed += mine
    assert np.all(np.abs(np.asarray(total["absorption"]) - summed) <= 1e-12*np.max(summed))


def test_banded_table_whole_grid_pedestal(farm, engine):
    """Dense bands: tiles split into many work items, thousands of windows per pedestal run."""
    t, p, x = level_of(SURFACE, "CO2", 0)
    handle = engine.load(table_from_recipe(BANDED))
    for ped in (False, True):
        k, evals = engine.compute(handle, t, p, x, 1, 5001, 1000, remove_pedestal=ped,
                                  want_evals=True)
        check_full(farm, ("banded", False), ("banded", ped), k[0], 1, 5001, 1000, ped,
                   f"banded CO2 ped={ped}", evals)
    engine.free(handle)


@pytest.mark.parametrize("farfield", [False, True])
def test_hitran_shaped_table_whole_grid(farm, engine, farfield):
    """What tests/test_gpu_fuzz_hitran.py checks on small grids, at the size bench.py times: every
    one of 5 M points against the oracle (spectra.c:17-62, voigt.c:17-53), pedestal off and on."""
    table = table_from_recipe(HITRAN_SHAPED)
    molecule = engine.load(table)
    t, p, x = HITRAN_LEVEL
    try:
        for ped in (False, True):
            k = engine.compute(molecule, t, p, x, 0, 5000, 1000, remove_pedestal=ped,
                               farfield=farfield)[0]
            check_full(farm, ("hitran", False), ("hitran", ped), k, 0, 5000, 1000, ped,
                       f"hitran-shaped farfield={farfield} ped={ped}")
    finally:
        engine.free(molecule)


@pytest.mark.parametrize("formula", ("H2O", "CO2", "O3"))
def test_default_user_path_at_ten_million_points(farm, engine, formula):
    """engine.compute(..., farfield=True, remove_pedestal=True) -- the two defaults of
    Spectroscopy.compute_absorption together -- on configs[4]'s grid (n_per_v = 2000: 1000-point
    cell-aligned tiles, windows of 102 001 points), every one of 10 M points, at ~1 hPa and at
    0.1 hPa (narrow cores everywhere: the inner-region pass inside the far-field kernel)."""
    handle = engine.load(table_from_recipe(uniform(formula)))
    for level in (FULL_LEVEL, LEVELS256[-1]):
        t, p, x = level_of(STANDARD256, formula, level)
        k, evals = engine.compute(handle, t, p, x, 1, 5001, 2000, remove_pedestal=True,
                                  want_evals=True, farfield=True)
        check_full(farm, ("c4", formula, level, False), ("c4", formula, level, True), k[0],
                   1, 5001, 2000, True, f"default path {formula} level {level} n_per_v=2000",
                   evals)
    engine.free(handle)


@pytest.mark.parametrize("cut", OTHER_CUTS)
def test_other_cut_offs_whole_grid(farm, engine, cut):
    """cut_off 5 and 40 (the reference's callers always pass 25, gas_optics.py:46-47, but the
    argument is part of absorption()'s signature): the 400 k-line CO2 table on all 5 M points,
    pedestal off and on, direct and with the far-field series.  Windows of 11 001 / 81 001
    points; with 5 cm-1 a tile's far-field range is a few lines wide, with 40 the pedestal's
    slots reach 81 cells back."""
    t, p, x = level_of(SURFACE, "CO2", 0)
    handle = engine.load(table_from_recipe(uniform("CO2")))
    for farfield in (False, True):
        for ped in (False, True):
            k, evals = engine.compute(handle, t, p, x, 1, 5001, 1000, cut_off=cut,
                                      remove_pedestal=ped, want_evals=True, farfield=farfield)
            check_full(farm, ("cut", cut, False), ("cut", cut, ped), k[0], 1, 5001, 1000, ped,
                       f"CO2 cut_off={cut} ped={ped} farfield={farfield}", evals, cut=cut)
    engine.free(handle)
