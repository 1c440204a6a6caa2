"""One rank of tests/test_gpu_distributed.py: started as a fresh interpreter (env RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT); DIST_BACKEND=gloo: every rank on GPU 0 (RCCL cannot put
two ranks on one device), DIST_BACKEND=nccl: one GPU per rank, RCCL carries the exchange.
Prints "rank R ok" or raises."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from pylbl_amd import distributed, synthetic
    from pylbl_amd.engine import Engine
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    n_levels, output = int(sys.argv[1]), sys.argv[2]
    # DIST_BACKEND=nccl (RCCL): one GPU per rank; default gloo with every rank on GPU 0.
    backend = os.environ.get("DIST_BACKEND", "gloo")
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    from datetime import timedelta
    os.environ.setdefault("PYLBL_AMD_EXCHANGE_TIMEOUT", "120")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", device),
                                timeout=timedelta(seconds=180))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=timedelta(seconds=180))
    formulas = ("H2O", "CO2", "O3")
    tables = {f: synthetic.line_table(f, 600., 700., num_lines=500 + 300*i, seed=31 + i)
              for i, f in enumerate(formulas)}
    atmos = synthetic.standard_atmosphere(max(n_levels, 2))
    t, p = atmos.t[:n_levels], atmos.p[:n_levels]
    vmr = {f: atmos.vmr[f][:n_levels] for f in formulas}
    v0, vn, npv = 610, 650, 200
    engine = Engine(device)
    handles = {f: engine.load(tables[f]) for f in formulas}
    sharded = distributed.ShardedLines.for_engine(
        engine, handles, (v0, vn, npv), remove_pedestal=True, scale_density=(output == "total"),
        weights=[tables[f].num_lines for f in formulas])
    for async_op in (False, True):
        out = sharded.run(t, p, vmr, dst=0, output=output, async_op=async_op)
        if async_op:
            out = out.wait()
        if rank != 0:
            assert out is None or all(v is None for v in out.values())
            continue
        kb = 1.38064852e-23
        expect = {}
        for f in formulas:
            rows = [oracle.absorption_port(tables[f], t[i], p[i], vmr[f][i], v0, vn, npv,
                                           remove_pedestal=True)[0] for i in range(n_levels)]
            expect[f] = np.asarray(rows)
        if output == "total":
            total = sum(expect[f]*(p*vmr[f]/(kb*t))[:, None] for f in formulas)
            got = out.cpu().numpy()
            assert np.max(np.abs(got - total)) <= 1e-6*np.max(np.abs(total)), "total"
        else:
            for f in formulas:
                got = out[f].cpu().numpy()
                scale = np.max(np.abs(expect[f]), axis=1, keepdims=True)
                assert np.max(np.abs(got - expect[f])/scale) <= 1e-6, f
    # The host-level hook: Spectroscopy(group=True) shards the levels, every mechanism of a
    # level stays on one rank, rank 0 collects; bit-identical to the unsharded call.
    import tempfile
    from pylbl_amd import Spectroscopy
    from pylbl_amd.database import Database, write_database
    with tempfile.TemporaryDirectory() as tmp:
        db = Database(write_database(os.path.join(tmp, f"lines{rank}.db"),
                                     [tables[f] for f in formulas]))
        grid = np.arange(float(v0), float(vn) - 1., 1./npv)
        atm = synthetic.Atmos(p=p.reshape(1, -1), t=t.reshape(1, -1),
                              vmr={f: x.reshape(1, -1) for f, x in vmr.items()})
        for fmt in ("all", "gas", "total"):
            sharded_out = Spectroscopy(atm, grid, db, group=True, device=device).compute_absorption(fmt)
            if rank != 0:
                assert sharded_out is None
                continue
            whole = Spectroscopy(atm, grid, db, device=device).compute_absorption(fmt)
            assert set(whole) == set(sharded_out)
            for name in whole:
                if name.endswith("absorption"):
                    assert whole[name].shape[:2] == (1, n_levels)
                assert np.array_equal(np.asarray(whole[name]), np.asarray(sharded_out[name])), name
    dist.barrier()
    dist.destroy_process_group()
    engine.close()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
