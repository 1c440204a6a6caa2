"""Multi-GPU form of the lines path: one process per GPU, atmospheric levels sharded.

The reference is serial (no threads, no MPI): ``Spectroscopy.compute_absorption`` loops
molecule-outer / level-inner and carries no state between iterations
(pyLBL/spectroscopy.py:166-191; every ``absorption()`` call starts from a memset,
pyLBL/c_lib/absorption.c:41).  (level, molecule) units are therefore independent and the
only exchange the path ever needs is the final gather of the level shards.

Layout: contiguous blocks of levels per rank, all molecules of a level on the same rank (so
per-gas / total reductions stay local), line tables replicated on every GPU.  The gather is
one ``torch.distributed`` collective -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the
CPU tests.
"""
import numpy as np


def level_shard(n_levels, rank, world):
    """Contiguous block of levels owned by `rank`: the first n_levels % world ranks get one
    extra level.  Returns a slice."""
    base, extra = divmod(int(n_levels), int(world))
    start = rank*base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def shard_sizes(n_levels, world):
    return [level_shard(n_levels, r, world).stop - level_shard(n_levels, r, world).start
            for r in range(world)]


def gather_levels(local, n_levels, dst=0, group=None):
    """Gathers per-rank blocks of levels (leading dimension) onto rank `dst`.

    Args:
        local: torch tensor [levels_local, ...] (CUDA with the nccl backend, CPU with gloo).
        n_levels: Total number of levels over all ranks.
        dst: Destination rank, or None for an all-gather (every rank gets the result).

    Returns:
        Tensor [n_levels, ...] on `dst` (on every rank when dst is None), else None.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_levels, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} levels, expected {sizes[rank]}.")
    if world == 1:
        return local
    widest = max(sizes)
    # Collectives want equal shapes: pad the short shards (at most one level each).
    padded = local
    if local.shape[0] != widest:
        padded = torch.zeros((widest,) + tuple(local.shape[1:]), dtype=local.dtype,
                             device=local.device)
        padded[:local.shape[0]] = local
    padded = padded.contiguous()
    if dst is None:
        pieces = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(pieces, padded, group=group)
    else:
        pieces = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        dist.gather(padded, pieces, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([pieces[r][:sizes[r]] for r in range(world)], dim=0)


class ShardedLines(object):
    """Lines spectra of a whole atmosphere over the ranks of a process group.

    Args:
        compute: Callable (formula, temperature[L], pressure[L], vmr[L]) -> tensor [L, n] on
                 this rank's device: the per-rank engine call.  Supplied by the caller so the
                 sharding / gather logic is independent of the device (see ``for_engine``).
    """
    def __init__(self, compute, group=None):
        self.compute = compute
        self.group = group

    @classmethod
    def for_engine(cls, engine, handles, grid_args, remove_pedestal=False, scale_density=False,
                   group=None):
        """Per-rank compute on an MI355X: spectra are written by the engine straight into a
        torch CUDA tensor (torch only owns the memory and runs the collective)."""
        import torch
        v0, vn, n_per_v = grid_args
        n = (vn - v0)*n_per_v

        class Slot(object):
            def __init__(self, tensor):
                self.pointer, self.shape = tensor.data_ptr(), tuple(tensor.shape)

        def compute(formula, temperature, pressure, vmr):
            out = torch.empty((len(temperature), n), dtype=torch.float64,
                              device=torch.device("cuda", engine.device))
            if len(temperature):
                engine.compute(handles[formula], temperature, pressure, vmr, v0, vn, n_per_v,
                               remove_pedestal=remove_pedestal, scale_density=scale_density,
                               out=Slot(out))
            return out
        return cls(compute, group=group)

    def run(self, temperature, pressure, vmr, dst=0):
        """Args: temperature[L], pressure[L]; vmr: dict formula -> [L].  Every rank passes the
        full atmosphere and computes only its own block of levels.

        Returns dict formula -> tensor [L, n] on rank `dst` (every rank if dst is None)."""
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        temperature = np.asarray(temperature, dtype=np.float64)
        pressure = np.asarray(pressure, dtype=np.float64)
        mine = level_shard(temperature.size, rank, world)
        out = {}
        for formula, x in vmr.items():
            local = self.compute(formula, temperature[mine], pressure[mine],
                                 np.asarray(x, dtype=np.float64)[mine])
            if world == 1:
                out[formula] = local
            else:
                out[formula] = gather_levels(local, temperature.size, dst=dst, group=self.group)
        return out
