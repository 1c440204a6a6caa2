"""One rank of tests/test_gpu_distributed.py::test_config3_shape_over_the_ranks_the_card_allows:
BASELINE configs[3]'s shape -- the 64-level standard atmosphere, H2O + CO2 + O3, levels in
contiguous blocks per rank, every molecule of a level on one rank -- on a 1-101 cm-1
grid at 0.001 cm-1, then ONE level cut into (level, molecule) units with the cross-rank reduce
(configs[2]'s mode), every rank on GPU 0, gloo carrying the blocks, kernels and exchange ordered
on the device as on the RCCL path.  Rank 0 checks two levels of EVERY rank's block against the
CPU oracle.  Prints "rank R ok" or raises."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    from oracle import oracle
    from pylbl_amd import distributed, synthetic
    from pylbl_amd.engine import Engine
    from tests import golden_io
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    levels_total = int(sys.argv[1])
    torch.cuda.set_device(0)
    os.environ.setdefault("PYLBL_AMD_EXCHANGE_TIMEOUT", "240")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=300))
    formulas = ("H2O", "CO2", "O3")
    # The bench's tables of configs[3] (1-3000 cm-1), cut to what can reach the grid.
    tables = {}
    for f in formulas:
        whole = synthetic.line_table(f, 1., 3000.)
        tables[f] = whole.subset(whole.nu <= 101. + 27.)
    v0, vn, npv = 1, 101, 1000
    n = (vn - v0)*npv
    atmos = synthetic.standard_atmosphere(levels_total)
    vmr = {f: atmos.vmr[f] for f in formulas}
    engine = Engine(0)
    handles = {f: engine.load(tables[f]) for f in formulas}
    weights = [tables[f].num_lines for f in formulas]

    def expected(f, level, t, p, x):
        return oracle.absorption_port(tables[f], t[level], p[level], x[f][level], v0, vn, npv,
                                      remove_pedestal=True)[0]

    def close(got, want, what):
        tolerance = golden_io.pedestal_tolerance(want, npv, 25, 1.e-6) + 1e-300
        assert np.max(np.abs(got - want)/tolerance) <= 1., what

    # ---- levels >= ranks: blocks of levels, one spectrum per gas collected on rank 0 ----------
    plan = distributed.partition(levels_total, weights, world)
    assert plan.mode == "levels"
    mine = distributed.level_shard(levels_total, rank, world)
    assert sorted({level for level, _ in plan.units[rank]}) == list(range(mine.start, mine.stop))
    sharded = distributed.ShardedLines.for_engine(engine, handles, (v0, vn, npv),
                                                  remove_pedestal=True, weights=weights)
    first = sharded.run(atmos.t, atmos.p, vmr, dst=0, output="gas", async_op=True)
    second = sharded.run(atmos.t, atmos.p, vmr, dst=0, output="gas", async_op=True)
    out = first.wait()
    again = second.wait()
    if rank == 0:
        for f in formulas:
            assert out[f].shape == (levels_total, n)
            assert torch.equal(out[f], again[f])            # queued back to back: same bits
        for other in range(world):
            block = distributed.level_shard(levels_total, other, world)
            for level in {block.start, block.stop - 1}:
                for f in formulas:
                    close(out[f][level].cpu().numpy(), expected(f, level, atmos.t, atmos.p, vmr),
                          f"{f} level {level} (rank {other}'s block)")
    else:
        assert out is None or all(v is None for v in out.values())
    del out, again

    # ---- one level, more ranks than levels: (level, molecule) units and the cross-rank sum ----
    one = distributed.partition(1, weights, world)
    assert one.mode == "units" and sum(len(u) for u in one.units) == len(formulas)
    t1, p1 = atmos.t[:1], atmos.p[:1]
    vmr1 = {f: atmos.vmr[f][:1] for f in formulas}
    total = distributed.ShardedLines.for_engine(engine, handles, (v0, vn, npv),
                                                remove_pedestal=True, scale_density=True,
                                                weights=weights)
    got = total.run(t1, p1, vmr1, dst=0, output="total")
    if rank == 0:
        kb = 1.38064852e-23
        want = sum(expected(f, 0, t1, p1, vmr1)*(p1[0]*vmr1[f][0]/(kb*t1[0])) for f in formulas)
        got = got.cpu().numpy()[0]
        assert np.max(np.abs(got - want)) <= 1e-6*np.max(np.abs(want)), "total of one level"
    dist.barrier()
    dist.destroy_process_group()
    engine.close()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
