"""ONE rank over RCCL on the GPU a one-GPU box has (tests/test_gpu_distributed.py starts it as a
fresh interpreter): what of the N-GPU path can run before a whole node is leased -- librccl
loading, a communicator on the device, the engine-to-RCCL stream ordering, the grouped send/recv
into strided views of the collected array (addressed to itself), all_reduce / reduce on memory
the engine wrote, and Pending.wait's device branch.  Prints "rccl one rank ok" or raises."""
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
    torch.cuda.set_device(0)
    os.environ.setdefault("PYLBL_AMD_EXCHANGE_TIMEOUT", "120")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0),
                            timeout=timedelta(seconds=120))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    formulas = ("H2O", "CO2", "O3")
    tables = {f: synthetic.line_table(f, 600., 700., num_lines=500 + 300*i, seed=31 + i)
              for i, f in enumerate(formulas)}
    n_levels = 3
    v0, vn, npv = 610, 650, 200
    engine = Engine(0)
    handles = {f: engine.load(tables[f]) for f in formulas}
    kb = 1.38064852e-23

    def atmosphere(shift):
        atmos = synthetic.standard_atmosphere(n_levels + 2)
        t, p = atmos.t[shift:shift + n_levels], atmos.p[shift:shift + n_levels]
        return t, p, {f: atmos.vmr[f][shift:shift + n_levels] for f in formulas}

    def expect(t, p, vmr, output):
        rows = {f: np.asarray([oracle.absorption_port(tables[f], t[i], p[i], vmr[f][i], v0, vn, npv,
                                                      remove_pedestal=True)[0]
                               for i in range(n_levels)]) for f in formulas}
        if output == "total":
            return sum(rows[f]*(p*vmr[f]/(kb*t))[:, None] for f in formulas)
        return rows

    def close(got, want):
        scale = np.max(np.abs(want), axis=-1, keepdims=True)
        return bool(np.max(np.abs(got - want)/scale) <= 1e-6)

    for output in ("gas", "total"):
        for collect_limit in (96 << 30, 0):         # the collected array kept twice / once
            sharded = distributed.ShardedLines.for_engine(
                engine, handles, (v0, vn, npv), remove_pedestal=True,
                scale_density=(output == "total"),
                weights=[tables[f].num_lines for f in formulas], always_exchange=True)
            sharded.collect_limit = collect_limit
            # Two calls queued back to back, neither waited for: the second's kernels run
            # beside the first's exchange, its exchange behind it (the buffers'
            # next writer is ordered behind the exchange that last used them).
            first, second = atmosphere(0), atmosphere(1)
            pending = [sharded.run(*first, dst=0, output=output, async_op=True),
                       sharded.run(*second, dst=0, output=output, async_op=True)]
            assert pending[0].device is not None        # Pending.wait's device branch
            results = [item.wait(timeout=60.) for item in pending]
            assert pending[1].bytes_received > 0 and pending[1].bytes_sent > 0   # sent to itself
            checks = ((second, results[1]),) if collect_limit == 0 else \
                ((first, results[0]), (second, results[1]))
            for inputs, out in checks:
                want = expect(*inputs, output)
                if output == "total":
                    assert close(out.cpu().numpy(), want), (output, collect_limit)
                else:
                    for f in formulas:
                        assert close(out[f].cpu().numpy(), want[f]), (output, collect_limit, f)
            # A third and fourth call: the per-rank blocks and the collected array come round
            # again, written by the engine / by RCCL behind the exchanges that last used them.
            third = sharded.run(*first, dst=None, output=output, async_op=True)
            fourth = sharded.run(*second, dst=None, output=output, async_op=False)
            want = expect(*second, output)
            got = fourth if output == "total" else fourth["CO2"]
            assert close(got.cpu().numpy(), want if output == "total" else want["CO2"])
            third.wait(timeout=60.)

    # Collectives on memory the engine has just written, ordered on the device:
    # lbl_order_stream_after_engine makes torch's stream (which RCCL's waits for) wait for the
    # engine's kernels; no host synchronisation in between.
    t, p, vmr = atmosphere(0)
    block = torch.empty((n_levels, (vn - v0)*npv), dtype=torch.float64, device="cuda:0")

    class Slot(object):
        pointer, shape = block.data_ptr(), tuple(block.shape)
    engine.compute(handles["CO2"], t, p, vmr["CO2"], v0, vn, npv, remove_pedestal=True,
                   out=Slot, asynchronous=True)
    engine.order_stream_after(torch.cuda.current_stream().cuda_stream)
    work = dist.all_reduce(block, op=dist.ReduceOp.SUM, async_op=True)
    work.wait()
    dist.reduce(block, dst=0, op=dist.ReduceOp.SUM)
    mirror = torch.zeros_like(block)
    requests = dist.batch_isend_irecv([dist.P2POp(dist.irecv, mirror[1:], 0),
                                       dist.P2POp(dist.isend, block[1:], 0)])
    for request in requests:
        request.wait()
    torch.cuda.current_stream().synchronize()
    want = expect(t, p, vmr, "gas")["CO2"]
    assert close(block.cpu().numpy(), want), "all_reduce / reduce on an engine-written block"
    assert torch.equal(mirror[1:], block[1:]) and float(mirror[0].abs().max()) == 0.
    # And back: the engine's next write of the block waits for what torch's stream holds.
    engine.order_after_stream(torch.cuda.current_stream().cuda_stream)
    engine.compute(handles["CO2"], t, p, vmr["CO2"], v0, vn, npv, remove_pedestal=True,
                   out=Slot, asynchronous=True)
    engine.synchronize()
    assert close(block.cpu().numpy(), want)

    with open("/proc/self/maps") as handle:
        mapped = sorted({line.split("/")[-1].strip() for line in handle if "rccl" in line})
    assert mapped, "librccl is not mapped into a process that ran RCCL collectives?"
    dist.barrier()
    dist.destroy_process_group()
    engine.close()
    print("rccl one rank ok: " + ", ".join(mapped))


if __name__ == "__main__":
    main()
