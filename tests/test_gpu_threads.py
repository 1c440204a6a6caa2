"""Several host threads on ONE engine handle.

The reference's absorption() keeps no state (pyLBL/c_lib/absorption.c:19-99: no globals, no
statics) and ctypes releases the interpreter lock around it (pyLBL/c_lib/gas_optics.py:79-91),
so any number of threads may call ``Gas.absorption_coefficient`` at once.  Here every ``Gas`` of
a process shares the engine of its device; the handle serialises the host side of its calls
(include/lbl_amd.h, "Threads").  Checked: every result a thread gets equals, bit for bit, what
the same call returns alone, and lies within the 1e-6 bar of the CPU oracle."""
import threading

import numpy as np
import pytest

from pylbl_amd import synthetic
from tests import golden_io

pytestmark = pytest.mark.gpu

THREADS = 4
ROUNDS = 3


def jobs_for(tables):
    """(formula, T, P, x, grid, remove_pedestal): different molecules and the same one, pedestal
    on and off, grids of different resolutions (so that the threads also build and look up
    work-item plans of the same molecule at once)."""
    surface = synthetic.surface_level()
    high = synthetic.standard_atmosphere(8)
    grids = [np.arange(1., 120., 0.01), np.arange(20., 90., 0.001), np.arange(1., 150., 0.1)]
    out = []
    for index, formula in enumerate(tables):
        for which, grid in enumerate(grids):
            for pedestal in (False, True):
                if (index + which) % 2:
                    t, p, x = surface.t[0], surface.p[0], surface.vmr[formula][0]
                else:
                    level = 2 + (index + which) % 5
                    t, p, x = high.t[level], high.p[level], high.vmr[formula][level]
                out.append((formula, float(t), float(p), float(x), grid, pedestal))
    return out


def within_bar(k, k_ref, n_per_v, pedestal, label):
    if pedestal:
        tol = golden_io.pedestal_tolerance(k_ref, n_per_v, 25, 1.e-6) + 1e-300
        assert np.max(np.abs(k - k_ref)/tol) <= 1., label
    else:
        np.testing.assert_allclose(k, k_ref, rtol=1.e-6, atol=0., err_msg=label)


def test_threads_share_the_default_engine(oracle):
    from pylbl_amd import Gas
    from pylbl_amd.engine import default_engine
    tables = {f: synthetic.line_table(f, 1., 150., num_lines=count, seed=70 + i,
                                      tips_range=(150, 400))
              for i, (f, count) in enumerate((("H2O", 2500), ("CO2", 4000), ("O3", 1500)))}
    gases = {f: Gas(t, f) for f, t in tables.items()}
    assert len({id(g.engine) for g in gases.values()}) == 1
    assert next(iter(gases.values())).engine is default_engine(0)
    jobs = jobs_for(tables)

    def call(job):
        formula, t, p, x, grid, pedestal = job
        return gases[formula].absorption_coefficient(t, p, x, grid, remove_pedestal=pedestal)

    alone = [np.array(call(job)) for job in jobs]
    for job, k in zip(jobs, alone):
        formula, t, p, x, grid, pedestal = job
        v0, vn, npv = synthetic.grid_arguments(grid)
        k_ref, _ = oracle.absorption_port(tables[formula], t, p, x, v0, vn, npv,
                                          remove_pedestal=pedestal)
        within_bar(k, k_ref, npv, pedestal, f"{formula} alone, pedestal={pedestal}")

    start = threading.Barrier(THREADS)
    failures = []

    def worker(index):
        try:
            order = np.random.default_rng(index).permutation(len(jobs))
            start.wait()
            for _ in range(ROUNDS):
                for j in order:
                    got = call(jobs[j])
                    if not np.array_equal(got, alone[j]):
                        worst = float(np.max(np.abs(got - alone[j])))
                        failures.append(f"thread {index}, job {j} ({jobs[j][0]}, pedestal="
                                        f"{jobs[j][5]}): differs from the call alone by {worst:g}")
        except BaseException as error:          # a thread must not die silently
            failures.append(f"thread {index}: {type(error).__name__}: {error}")
            start.abort()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(THREADS)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join(timeout=600)
    assert not any(thread.is_alive() for thread in threads), "a thread is stuck"
    assert not failures, "\n".join(failures[:10])


def test_threads_mix_the_entry_points(oracle, tmp_path):
    """One thread in Spectroscopy.compute_absorption (a pipeline of asynchronous calls that add
    into shared blocks, one of them kept back: Engine.pipeline), one in the reference's own
    11-argument C entry (which keeps an engine of its own on the same GPU), two in Gas calls on
    the engine Spectroscopy uses.  Everybody gets, bit for bit, what the call returns alone;
    error messages stay with the thread that failed."""
    from ctypes import c_char_p, c_double, c_int
    from numpy.ctypeslib import ndpointer
    from pylbl_amd import Gas, Spectroscopy, engine as binding
    from pylbl_amd.database import Database, write_database
    from pylbl_amd.errors import EngineError
    tables = [synthetic.line_table("H2O", 1., 130., num_lines=1500, seed=81, tips_range=(150, 400)),
              synthetic.line_table("CO2", 1., 130., num_lines=2500, seed=82, tips_range=(150, 400))]
    path = tmp_path / "lines.db"
    write_database(path, tables)
    database = Database(str(path))
    full = synthetic.fixture_atmosphere()
    atmos = synthetic.Atmos(p=full.p, t=full.t, vmr={k: full.vmr[k] for k in ("H2O", "CO2")})
    grid = np.arange(1., 100., 0.01)
    v0, vn, npv = synthetic.grid_arguments(grid)
    spectroscopy = Spectroscopy(atmos, grid, database)
    gas = Gas(database, "CO2")
    entry = binding.library().absorption
    entry.restype = c_int
    entry.argtypes = 3*[c_double] + 3*[c_int] + [ndpointer(c_double, flags="C_CONTIGUOUS")] + \
        2*[c_char_p] + 2*[c_int]

    def through_spectroscopy(fmt):
        out = spectroscopy.compute_absorption(output_format=fmt)
        return {key: np.array(value) for key, value in out.items() if key.endswith("absorption")}

    def through_gas(pedestal):
        return np.array(gas.absorption_coefficient(250., 3.e4, 4.e-4, grid,
                                                   remove_pedestal=pedestal))

    def through_c_entry(pedestal):
        k = np.full((vn - v0)*npv, 7.)
        assert entry(98388., 288.99, 6.6e-3, v0, vn, npv, k, str(path).encode(), b"H2O", 25,
                     int(pedestal)) == 0
        return k

    alone = {"total": through_spectroscopy("total"), "gas": through_spectroscopy("gas"),
             "k0": through_gas(False), "k1": through_gas(True),
             "c0": through_c_entry(False), "c1": through_c_entry(True)}
    k_ref, _ = oracle.absorption_port(tables[0], 288.99, 98388., 6.6e-3, v0, vn, npv)
    within_bar(alone["c0"], k_ref, npv, False, "C entry alone")
    start = threading.Barrier(4)
    failures = []

    def same(got, want, label):
        if isinstance(want, dict):
            for key in want:
                same(got[key], want[key], f"{label}[{key}]")
        elif not np.array_equal(got, want):
            failures.append(f"{label}: differs by {float(np.max(np.abs(got - want))):g} "
                            f"(max {float(np.max(np.abs(want))):g})")

    def guarded(body, label):
        def run():
            try:
                start.wait()
                for round_ in range(ROUNDS):
                    body(round_)
            except BaseException as error:
                failures.append(f"{label}: {type(error).__name__}: {error}")
                start.abort()
        return threading.Thread(target=run)

    def spectroscopy_body(round_):
        fmt = ("total", "gas")[round_ % 2]
        same(through_spectroscopy(fmt), alone[fmt], f"Spectroscopy {fmt}")

    def gas_body(round_):
        for pedestal in (False, True):
            same(through_gas(pedestal), alone[f"k{int(pedestal)}"], f"Gas pedestal={pedestal}")

    def c_body(round_):
        for pedestal in (False, True):
            same(through_c_entry(pedestal), alone[f"c{int(pedestal)}"], f"absorption() {pedestal}")

    def failing_body(round_):
        # A failure of this thread (temperature outside the partition sums) must come back with
        # its own message while the others keep computing.
        gas_body(round_)
        with pytest.raises(EngineError, match="partition-function table"):
            gas.absorption_coefficient(5000., 3.e4, 4.e-4, grid)

    threads = [guarded(spectroscopy_body, "spectroscopy"), guarded(gas_body, "gas"),
               guarded(c_body, "c entry"), guarded(failing_body, "failing")]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join(timeout=600)
    assert not any(thread.is_alive() for thread in threads), "a thread is stuck"
    assert not failures, "\n".join(failures[:10])


def test_a_failure_between_a_deferred_call_and_its_finish_leaves_nothing_behind():
    """Spectroscopy in "total" mode keeps the heaviest gas's last kernels back; an
    exception raised before finish_deferred() used to leave them on the engine, to run later
    into a block and a page-locked array that had gone back to their pools.  Now the pipeline
    drops them (lbl_cancel_deferred) and waits before anything is released."""
    from pylbl_amd import Spectroscopy
    from pylbl_amd.database import MemoryDatabase
    tables = [synthetic.line_table("H2O", 1., 130., num_lines=800, seed=91, tips_range=(150, 400)),
              synthetic.line_table("CO2", 1., 130., num_lines=4000, seed=92, tips_range=(150, 400))]
    database = MemoryDatabase(tables)
    full = synthetic.fixture_atmosphere()
    atmos = synthetic.Atmos(p=full.p, t=full.t, vmr={k: full.vmr[k] for k in ("H2O", "CO2")})
    grid = np.arange(1., 100., 0.01)
    spectroscopy = Spectroscopy(atmos, grid, database)
    spectroscopy.total_order = "deferred"       # the heaviest gas queued first, finished last
    want = np.array(spectroscopy.compute_absorption(output_format="total")["absorption"])
    engine = spectroscopy._molecule("CO2").gas.engine

    class Boom(Exception):
        pass

    light = spectroscopy._molecule("H2O").gas
    original = light.absorption_coefficients

    def failing(*args, **kwargs):
        assert engine.deferred()            # CO2 (the heavier table) is being kept back
        raise Boom("raised between the deferred call and finish_deferred()")
    light.absorption_coefficients = failing
    raised = False
    try:
        spectroscopy.compute_absorption(output_format="total")
    except Boom:
        raised = True           # (no traceback kept: the call's arrays are released here)
    finally:
        light.absorption_coefficients = original
    assert raised and not engine.deferred()
    import gc
    gc.collect()
    # The pools hand the same block and array out again; nothing may write them any more.
    v0, vn, npv = synthetic.grid_arguments(grid)
    block = engine.blocks.take(4, (vn - v0)*npv)
    engine.fill_zero(block)
    marker = engine.host_array((4, grid.size))
    marker[...] = -3.
    engine.synchronize()
    assert not block.to_host().any() and np.all(marker == -3.)
    engine.blocks.give(block)
    got = np.array(spectroscopy.compute_absorption(output_format="total")["absorption"])
    assert np.array_equal(got, want)
