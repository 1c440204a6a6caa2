"""Seeded fuzz of what orders calls against each other: random sequences of asynchronous calls --
several molecules, the pedestal removed or not, the far-field series or not, writing a block or
adding into it, one of them kept back (LBL_DEFER_FINISH) and finished later, some delivering their
block, a zero-fill in between -- on two blocks in HBM.  Calls rotate over the engine's lanes and
run side by side; the only thing that orders them is the memory they write.  Each block must end
up as the same calls made one after the other would leave it (additions in the order the engine
promises: queue order, a kept-back call last).  48 cases in the regular run;
PYLBL_FUZZ_PIPELINE=1000 to soak."""
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


@pytest.mark.parametrize("seed", range(int(os.environ.get("PYLBL_FUZZ_PIPELINE", "48"))))
def test_pipeline(engine, seed):
    from pylbl_amd.engine import DeviceSpectra
    rng = np.random.default_rng(64_000 + seed)
    npv = int(rng.choice([10, 100, 1000]))
    v0 = int(rng.integers(1, 3000))
    span = int(rng.integers(3, max(4, min(60, 600_000//npv))))
    if rng.random() < 0.15:
        # more than 2^20 points: plain calls take turns on two lanes (engine option overlap_plain;
        # three up to 2^20 points, small_points)
        npv, span = 1000, int(rng.integers(1060, 1300))
    vn = v0 + span
    n = span*npv
    lo, hi = max(v0 - 26., 0.05), vn + 26.
    formulas = ["H2O", "CO2", "CH4", "O3"][:int(rng.integers(2, 5))]
    tables = [synthetic.line_table(f, lo, hi, num_lines=int(rng.integers(1, 3000)),
                                   seed=int(rng.integers(1 << 30)), tips_range=(150, 400))
              for f in formulas]
    handles = [engine.load(t) for t in tables]
    levels = int(rng.integers(1, 4))
    t = rng.uniform(190., 320., levels)
    p = 10.**rng.uniform(1., 5.05, levels)
    x = 10.**rng.uniform(-6., -1., levels)
    blocks = [DeviceSpectra(engine, levels, n) for _ in range(2)]
    state = [None, None]            # what each block holds (None: nothing defined yet)
    plain_cache = {}

    def plain(m, ped, far):
        key = (m, ped, far)
        if key not in plain_cache:
            plain_cache[key] = engine.compute(handles[m], t, p, x, v0, vn, npv, remove_pedestal=ped,
                                              farfield=far, scale_density=True,
                                              range_policy="skip").copy()
        return plain_cache[key]

    log = []
    kept = None                     # (block, values) of the call kept back
    deliveries = []                 # (host view, expected values)
    being_read = [False, False]     # a delivery of the block is under way (include/lbl_amd.h: it
                                    # must not be written again before lbl_synchronize)
    try:
        for step in range(int(rng.integers(3, 9))):
            b = int(rng.integers(0, 2))
            kind = rng.random()
            if being_read[b]:
                if kept is not None:            # (synchronize finishes a kept-back call)
                    state[kept[0]] = state[kept[0]] + kept[1]
                    kept = None
                engine.synchronize()
                being_read = [False, False]
                log.append("sync")
            if kind < 0.1 and kept is None:
                engine.fill_zero(blocks[b], asynchronous=True)
                state[b] = np.zeros((levels, n))
                log.append(f"zero {b}")
                continue
            m = int(rng.integers(0, len(handles)))
            ped = bool(rng.integers(0, 2))
            far = bool(rng.integers(0, 2))
            values = plain(m, ped, far)
            adding = state[b] is not None and (rng.random() < 0.7 or
                                               (kept is not None and kept[0] == b))
            defer = kept is None and adding and ped and rng.random() < 0.3
            deliver = None
            # (a kept-back call on this block will add into it when it is finished -- a write like
            # any other, which a delivery of the block must not have under way)
            if rng.random() < 0.3 and not defer and not (kept is not None and kept[0] == b):
                deliver = engine.host_array((levels, n))
                deliver[...] = -1.
            engine.compute(handles[m], t, p, x, v0, vn, npv, remove_pedestal=ped, farfield=far,
                           scale_density=True, range_policy="skip", out=blocks[b],
                           accumulate=adding, asynchronous=True, deliver=deliver,
                           pieces=int(rng.integers(1, 6)), defer_finish=defer)
            log.append(f"{formulas[m]} -> {b} ped={ped} far={far} add={adding} defer={defer} "
                       f"deliver={deliver is not None}")
            if defer and engine.deferred():
                kept = (b, values)
                continue
            state[b] = state[b] + values if adding else values.copy()
            if deliver is not None:
                # (a call kept back on this block adds after this one: the delivered values are
                # the block's as of this call)
                deliveries.append((deliver, state[b].copy()))
                being_read[b] = True
            if kept is not None and rng.random() < 0.4:
                engine.finish_deferred()
                state[kept[0]] = state[kept[0]] + kept[1]
                log.append("finish")
                kept = None
        if kept is not None:
            state[kept[0]] = state[kept[0]] + kept[1]       # lbl_synchronize finishes it
        engine.synchronize()
        label = f"seed {seed}: npv={npv} span={span} levels={levels}: " + "; ".join(log)
        for b in range(2):
            if state[b] is None:
                continue
            got = blocks[b].to_host()
            scale = np.max(np.abs(state[b]), axis=1, keepdims=True) + 1e-300
            assert np.max(np.abs(got - state[b])/scale) <= 1e-15, f"block {b}: " + label
        for index, (view, expected) in enumerate(deliveries):
            scale = np.max(np.abs(expected), axis=1, keepdims=True) + 1e-300
            wrong = np.abs(view - expected)/scale > 1e-15
            where = np.where(wrong.any(axis=0))[0]
            assert not wrong.any(), (f"delivery {index}: columns {where[0]}..{where[-1]} "
                                     f"({where.size} of {n}), view {view[0, where[0]]:.6e} "
                                     f"expected {expected[0, where[0]]:.6e}: " + label)
    finally:
        engine.synchronize()
        for block in blocks:
            block.free()
        for h in handles:
            engine.free(h)
