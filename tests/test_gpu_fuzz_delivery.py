"""Seeded fuzz of the call that delivers its result while it computes (lbl_compute_streamed through
Engine.compute(deliver=...)): small grids of a few tiles, 1-8 runs of tiles, tables with a handful of
lines anywhere within reach of the grid (runs of tiles then come out uneven or empty), several
levels, level passes forced by a small workspace, the pedestal removed or not, the far-field
series, fewer columns than points, adding into a block that holds something.  The block in HBM
and the delivered array must be what the plain call computes, bit for bit (adding: to the
rounding of one addition).  64 cases in the regular run; PYLBL_FUZZ_DELIVERY=2000 to soak."""
import os

import numpy as np
import pytest

from pylbl_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from pylbl_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_DELIVERY", "64"))))
def test_delivery(engine, seed):
    from pylbl_amd.engine import DeviceSpectra
    rng = np.random.default_rng(52_000 + seed)
    npv = int(rng.choice([1, 4, 10, 25, 100, 1000]))
    v0 = int(rng.integers(1, 3000))
    span = int(rng.integers(2, max(3, min(80, 40_000//npv))))
    vn = v0 + span
    n = span*npv
    # lines anywhere within reach of the grid -- or only beside one end of it
    lo, hi = max(v0 - 26., 0.05), vn + 26.
    where = rng.random()
    if where < 0.25:
        lo = max(hi - rng.uniform(1., 12.), lo)
    elif where < 0.5:
        hi = min(lo + rng.uniform(1., 12.), hi)
    table = synthetic.line_table(str(rng.choice(["CO2", "CH4", "H2O"])), lo, hi,
                                 num_lines=int(rng.integers(1, 600)), seed=int(rng.integers(1 << 30)),
                                 tips_range=(150, 400))
    levels = int(rng.integers(1, 6))
    t = rng.uniform(190., 320., levels)
    p = 10.**rng.uniform(1., 5.05, levels)
    x = 10.**rng.uniform(-7., -1., levels)
    ped = bool(rng.integers(0, 2))
    far = bool(rng.integers(0, 2))
    pieces = int(rng.integers(1, 9))
    columns = n if rng.random() < 0.4 else int(rng.integers(1, n + 1))
    adding = rng.random() < 0.35
    passes = rng.random() < 0.25
    label = (f"seed {seed}: npv={npv} v0={v0} span={span} lines={table.num_lines} levels={levels} "
             f"ped={ped} farfield={far} pieces={pieces} columns={columns}/{n} adding={adding} "
             f"passes={passes}")
    handle = engine.load(table)
    try:
        plain = engine.compute(handle, t, p, x, v0, vn, npv, remove_pedestal=ped, farfield=far,
                               scale_density=True, range_policy="skip")
        out = DeviceSpectra(engine, levels, n)
        before = np.zeros((levels, n))
        if adding:
            other = engine.compute(handle, t[::-1].copy(), p, x, v0, vn, npv, scale_density=True,
                                   range_policy="skip")
            engine.compute(handle, t[::-1].copy(), p, x, v0, vn, npv, scale_density=True,
                           range_policy="skip", out=out)
            before = other
        holder = engine.host_array((levels, 2, columns))
        holder[...] = -1.
        target = holder[:, 0, :]
        if passes:
            engine.set_option("workspace_bytes", 1 << 20)
        engine.compute(handle, t, p, x, v0, vn, npv, remove_pedestal=ped, farfield=far,
                       scale_density=True, range_policy="skip", out=out, accumulate=adding,
                       asynchronous=True, deliver=target, pieces=pieces)
        engine.synchronize()
        block = out.to_host()
        out.free()
        expect = before + plain if adding else plain
        if adding:
            scale = np.max(np.abs(expect), axis=1, keepdims=True) + 1e-300
            assert np.max(np.abs(block - expect)/scale) <= 4e-16, label
        else:
            assert np.array_equal(block, expect), label
        assert np.array_equal(target, block[:, :columns]), label
        assert np.all(holder[:, 1, :] == -1.), label
    finally:
        engine.set_option("workspace_bytes", 4 << 30)
        engine.free(handle)
