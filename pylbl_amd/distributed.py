"""Multi-GPU form of the lines path: one process per GPU, (level, molecule) units sharded.

The reference is serial (no threads, no MPI): ``Spectroscopy.compute_absorption`` loops
molecule-outer / level-inner and carries no state between iterations
(pyLBL/spectroscopy.py:166,179; every ``absorption()`` call starts from a memset,
pyLBL/c_lib/absorption.c:41).  (level, molecule) units are therefore independent, and the only
exchange the path ever needs is the final collection of the spectra on one rank -- plus, when
the molecules of ONE level are spread over several GPUs and the caller wants the sum over gases
(spectroscopy.py:225-234), one sum-reduction.

Partition (``partition``):
  * levels >= ranks: contiguous blocks of whole levels per rank, all molecules of a level on the
    same rank, so the per-gas / total sums stay on the device that computed them (BASELINE
    configs 4 and 5: 64 and 256 levels over 8 GPUs);
  * levels < ranks: the level-major list of (level, molecule) units is cut into contiguous runs
    of near-equal weight (weight = transitions of the molecule), so that e.g. one level x eight
    molecules (BASELINE config 3) occupies eight GPUs instead of one.  A single unit is never
    split: one spectrum's grid stays on one GPU.
Line tables are replicated on every GPU (<= 1.4 M transitions, ~90 MB).

Exchange: one grouped point-to-point gather per call (every rank sends its blocks straight
into their final place on the destination: `torch.distributed.batch_isend_irecv`, i.e. one
ncclGroup of send/recv over xGMI, no padding and no concatenation copy), or one `reduce` for
the cross-rank total.  Backend "nccl" is RCCL on ROCm (tensors stay in HBM); with "gloo" (CPU
tests, or rehearsing several ranks on one GPU) the blocks travel through host memory.
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


class Partition(object):
    """Which (level, molecule) units each rank computes.

    Attributes:
        mode: "levels" (whole levels per rank) or "units" (levels < ranks).
        units: units[rank] = list of (level, molecule index), level-major.
    """
    def __init__(self, mode, units, n_levels, n_molecules):
        self.mode, self.units = mode, units
        self.n_levels, self.n_molecules = n_levels, n_molecules

    def levels_of(self, rank):
        """Sorted levels rank touches."""
        return sorted({level for level, _ in self.units[rank]})

    def by_molecule(self, rank):
        """{molecule index: [levels]} of the rank's units, levels ascending."""
        out = {}
        for level, molecule in self.units[rank]:
            out.setdefault(molecule, []).append(level)
        return out


def partition(n_levels, weights, world):
    """Splits n_levels x len(weights) units over `world` ranks (see the module docstring).

    Args:
        weights: Relative cost of one level of each molecule (e.g. its number of transitions).
    """
    n_levels, world = int(n_levels), int(world)
    weights = [max(float(w), 1e-9) for w in weights]
    n_molecules = len(weights)
    if n_levels >= world or n_molecules <= 1:
        units = []
        for rank in range(world):
            mine = level_shard(n_levels, rank, world)
            units.append([(level, m) for level in range(mine.start, mine.stop)
                          for m in range(n_molecules)])
        return Partition("levels", units, n_levels, n_molecules)
    # Fewer levels than ranks: contiguous runs of the level-major unit list, cut where the
    # running weight (taken at each unit's midpoint) crosses a multiple of total/world.
    flat = [(level, m) for level in range(n_levels) for m in range(n_molecules)]
    total = n_levels*sum(weights)
    units = [[] for _ in range(world)]
    running = 0.
    for level, m in flat:
        middle = running + 0.5*weights[m]
        rank = min(int(middle*world/total), world - 1)
        units[rank].append((level, m))
        running += weights[m]
    return Partition("units", units, n_levels, n_molecules)


def _group_info(group):
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1, None
    return dist.get_rank(group), dist.get_world_size(group), dist.get_backend(group)


def gather_levels(local, n_levels, dst=0, group=None):
    """Gathers per-rank blocks of levels (leading dimension) onto rank `dst` with one padded
    gather / all-gather collective.

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
        dist.gather(padded, pieces, dst=_global_rank(group, dst), group=group)
        if rank != dst:
            return None
    return torch.cat([pieces[r][:sizes[r]] for r in range(world)], dim=0)


class ExchangeTimeout(RuntimeError):
    """An exchange did not complete in time: names this rank and the peers it was waiting for.
    A hung collective cannot be cancelled; callers print what they know and leave the process
    (bench.py exits non-zero with ``os._exit``; never re-exec a process that holds a GPU)."""


def exchange_timeout():
    """Seconds a Pending waits by default: $PYLBL_AMD_EXCHANGE_TIMEOUT, else 300."""
    import os
    try:
        return float(os.environ.get("PYLBL_AMD_EXCHANGE_TIMEOUT", "300"))
    except ValueError:
        return 300.


def _global_rank(group, rank):
    """torch's point-to-point and rooted collectives address *global* ranks; inside a
    sub-group the partition's ranks are group-local."""
    import torch.distributed as dist
    if group is None:
        return rank
    return dist.get_global_rank(group, rank)


class Pending(object):
    """An exchange in flight: wait() returns what run() would have returned.

    Attributes:
        seconds: after wait(): host seconds from the call that started the exchange until its
                 results were usable (transfer + whatever it overlapped).
        bytes_sent, bytes_received: what this rank moves in this exchange.
    """
    def __init__(self, requests, finish, device=None, flush=None, describe="", rank=0,
                 peers=(), bytes_sent=0, bytes_received=0):
        import time
        self.requests, self.finish, self.device = requests, finish, device
        self.flush = flush
        self.describe, self.rank, self.peers = describe, rank, tuple(peers)
        self.bytes_sent, self.bytes_received = int(bytes_sent), int(bytes_received)
        self.started = time.perf_counter()
        self.seconds = None
        self.result = None
        self.done = False
        # Device tensors: an event behind everything the stream that issued the exchange has been
        # given so far (staging copies, the rank's own blocks placed in the collected array).  A
        # later writer of the buffers waits for it too: the work objects only stand for the
        # transfer, and the caller may by then be under another torch stream.
        self.issued = None
        if device is not None:
            import torch
            self.issued = torch.cuda.Event()
            self.issued.record(torch.cuda.current_stream(device))

    def _late(self, timeout):
        return ExchangeTimeout(
            f"rank {self.rank}: {self.describe or 'exchange'} with rank(s) "
            f"{list(self.peers)} not complete after {timeout:g} s "
            f"({self.bytes_sent} B to send, {self.bytes_received} B to receive)")

    def wait(self, timeout=None):
        """Blocks until the exchange has finished and its result may be read.

        Args:
            timeout: seconds (default ``exchange_timeout()``); ExchangeTimeout when exceeded.
        """
        import time
        if self.done:
            return self.result
        timeout = exchange_timeout() if timeout is None else float(timeout)
        deadline = time.perf_counter() + timeout
        if self.flush is not None:
            self.flush()
        if self.device is None:
            # Host tensors (gloo): the requests complete on the host.
            # (gloo's point-to-point work objects complete inside wait() only: no polling.)
            from datetime import timedelta
            for request in self.requests:
                left = deadline - time.perf_counter()
                try:
                    done = left > 0. and request.wait(timeout=timedelta(seconds=left))
                except RuntimeError as error:
                    raise self._late(timeout) from error
                if done is False:
                    raise self._late(timeout)
        else:
            import torch
            # An nccl work object only orders the *current torch stream of its device* behind
            # the transfer: name the device (the caller's current device may be another one)
            # and settle that stream with an event that can be polled against the deadline.
            # (Also with nothing to send or receive: the rank's own blocks were copied into the
            # collected array on that stream, and the engine overwrites them two calls later.)
            stream = torch.cuda.current_stream(self.device)
            with torch.cuda.device(self.device):
                if self.issued is not None:
                    stream.wait_event(self.issued)
                for request in self.requests:
                    request.wait()
                marker = torch.cuda.Event()
                marker.record(stream)
            spins = 0
            while not marker.query():
                spins += 1
                if spins > 2000:
                    if time.perf_counter() > deadline:
                        raise self._late(timeout)
                    time.sleep(1e-4)
        self.result = self.finish()
        self.seconds = time.perf_counter() - self.started
        self.done = True
        return self.result


class ShardedLines(object):
    """Lines spectra of a whole atmosphere over the ranks of a process group.

    Args:
        compute: Callable (formula, temperature[L'], pressure[L'], vmr[L'], out, accumulate):
                 fills (or, with accumulate, adds to) the tensor ``out`` [L', n] on this rank's
                 device with the spectra of those levels.  Supplied by the caller so that the
                 sharding / exchange logic is independent of the device (see ``for_engine``).
        molecules: Formulae, in the order results are reported.
        n: Points per spectrum.
        weights: Relative cost per molecule (default: equal); only used when levels < ranks.
        device: torch device of the per-rank blocks ("cpu" in the gloo tests).
        flush: Callable that returns once everything `compute` queued has finished (the
               engine's synchronize); None if `compute` is synchronous.
        order: Callable that orders the exchange library's stream on `device` behind everything
               `compute` has queued, without stopping the host (for_engine: HIP events between
               the engine's streams and torch's); None: `flush` is used instead.
        zero: Callable (tensor) that zero-fills a block in the order `compute` works in
              (for_engine: the engine's own fill); None: torch fills and the host waits.
        order_on_device: None (default; $PYLBL_AMD_ORDER_ON_DEVICE=1 turns it on): `order` is
              used when the exchange stays on the device (nccl), `flush` otherwise.  True: `order`
              whenever the blocks are on a device.
        order_compute: Callable that orders everything `compute` queues from now on behind what
              the exchange library's stream on `device` holds now (for_engine: the engine's
              streams wait for torch's current stream); None: nothing `compute` queues runs on a
              stream of its own.
        collect_limit: bytes; a collected array of at most half of this is kept twice (used in
              turn, like the per-rank blocks), a larger one once.
        always_exchange: go through the collection step also with ONE rank in the group (the
              rank's blocks are copied into the collected array on the exchange library's stream,
              ordered behind the kernels as for N ranks): what a one-GPU box can run of the RCCL
              path's ordering.

    Memory on the receiving rank: the collected array ([L, n] for "total", [molecules, L, n] for
    "gas") is allocated once and kept -- two of it (used in turn, a result stays valid until the
    call after next) while both fit `collect_limit` (96 GB), else one: BASELINE config 5 in "gas"
    mode needs 164 GB + 2 x 20.5 GB (the rank's own blocks) of its 288 GB, "total" mode
    2 x 20.5 GB + 2 x 2.6 GB.

    Ordering between calls: every kept buffer remembers the exchange that last read or wrote it,
    and whoever writes it next is ordered behind that exchange first -- on the device (the
    writer's stream waits; RCCL) or by waiting for it (host tensors; gloo).  Calls may therefore
    be queued back to back with async_op without waiting for their Pendings; what is lost by not
    waiting is only a result that a later call has overwritten.
    """
    def __init__(self, compute, molecules, n, weights=None, group=None, device="cpu",
                 flush=None, order=None, zero=None, order_on_device=None, order_compute=None,
                 collect_limit=96 << 30, always_exchange=False):
        self.compute = compute
        self.molecules = list(molecules)
        self.n = int(n)
        self.weights = list(weights) if weights is not None else [1.]*len(self.molecules)
        self.group = group
        self.device = device
        self.flush = flush
        self.order = order
        self.zero = zero
        # None: the kernels and the exchange are ordered on the device (no host wait) exactly
        # when the exchange stays in HBM (nccl).  True: also when the blocks travel through host
        # memory (gloo on a GPU) -- the staging copies then wait, on torch's stream, for the
        # engine's events: a rehearsal of the RCCL path's ordering on fewer GPUs than ranks.
        if order_on_device is None:
            import os
            order_on_device = True if os.environ.get("PYLBL_AMD_ORDER_ON_DEVICE") == "1" else None
        self.order_on_device = order_on_device
        self.order_compute = order_compute
        self.collect_limit = int(collect_limit)
        self.always_exchange = bool(always_exchange)
        self._buffers = {}
        self._users = {}                # buffer key -> the Pending that last read or wrote it
        self._turn = 0
        self.last_exchange = None       # the Pending of the latest call with world > 1

    @classmethod
    def for_engine(cls, engine, handles, grid_args, remove_pedestal=False, scale_density=False,
                   range_policy="reference", weights=None, group=None, order_on_device=None,
                   farfield=False, cut_off=25, always_exchange=False):
        """Per-rank compute on an MI355X: the engine writes spectra straight into torch CUDA
        tensors (torch only owns the memory and runs the exchange).

        Args:
            handles: {formula: engine molecule handle} (insertion order = reporting order).
            grid_args: (v0, vn, n_per_v) as pyLBL/c_lib/gas_optics.py:61-63 derives them.
            farfield: distant lines through their power series (LBL_FARFIELD; what
                      Spectroscopy runs by default).
            cut_off: as Gas.absorption_coefficient's (pyLBL/c_lib/gas_optics.py:46-47).
        """
        import torch
        v0, vn, n_per_v = grid_args
        n = (vn - v0)*n_per_v

        class Slot(object):
            def __init__(self, tensor):
                self.pointer, self.shape = tensor.data_ptr(), tuple(tensor.shape)

        def compute(formula, temperature, pressure, vmr, out, accumulate):
            if len(temperature):
                engine.compute(handles[formula], temperature, pressure, vmr, v0, vn, n_per_v,
                               cut_off=cut_off, remove_pedestal=remove_pedestal,
                               scale_density=scale_density, range_policy=range_policy,
                               out=Slot(out), accumulate=accumulate, asynchronous=True,
                               farfield=farfield)
        device = torch.device("cuda", engine.device)

        def order():
            # torch's current stream of the engine's device waits (on the device) for the
            # engine's streams; the exchange torch launches next is ordered behind that stream.
            engine.order_stream_after(torch.cuda.current_stream(device).cuda_stream)

        def order_compute():
            # the other way round: what the engine queues next waits for torch's current stream
            # (which an exchange's work object has just been made to wait for).
            engine.order_after_stream(torch.cuda.current_stream(device).cuda_stream)

        def zero(tensor):
            engine.fill_zero(Slot(tensor), asynchronous=True)
        return cls(compute, list(handles), n, weights=weights, group=group, device=device,
                   flush=engine.synchronize, order=order, zero=zero,
                   order_on_device=order_on_device, order_compute=order_compute,
                   always_exchange=always_exchange)

    # -- buffers -----------------------------------------------------------------------------
    def _buffer(self, name, shape, zero=False, sets=2, writer="compute"):
        """Per-rank blocks are kept between calls, two of each so that an exchange still in
        flight (async_op) is not overwritten by the next call (sets=1: one, for a collected
        array too large to keep twice).  Whoever is about to write the buffer -- `compute`
        (the engine's streams) or the exchange library's stream ("exchange") -- is first ordered
        behind the exchange that last used it (_settle)."""
        import torch
        key = (name, self._turn % sets, tuple(shape))
        tensor = self._buffers.get(key)
        if tensor is None:
            tensor = torch.empty(shape, dtype=torch.float64, device=self.device)
            self._buffers[key] = tensor
        self._settle(key, writer)
        self._last_key = key
        if zero:
            self._zero(tensor)
        return tensor

    def _settle(self, key, writer):
        """Orders the next writer of buffer `key` behind the exchange that last read or wrote it
        (if that is still in flight).  Host tensors: the host waits for the exchange (gloo's
        requests complete inside wait() only: this may block for up to PYLBL_AMD_EXCHANGE_TIMEOUT
        seconds and raise ExchangeTimeout).  Device tensors: the exchange library's current
        stream waits for the exchange's work objects -- no host wait -- and, when the writer is
        `compute`, the engine's streams are then ordered behind that stream."""
        pending = self._users.pop(key, None)
        if pending is None or pending.done:
            return
        if pending.device is None:
            pending.wait()
            return
        import torch
        with torch.cuda.device(pending.device):
            if pending.issued is not None:
                # (what the issuing stream did around the exchange, should the caller have changed
                # torch streams since)
                torch.cuda.current_stream(pending.device).wait_event(pending.issued)
            for request in pending.requests:
                request.wait()              # nccl: orders torch's current stream, not the host
        if writer == "compute":
            if self.order_compute is not None:
                self.order_compute()
            else:
                torch.cuda.current_stream(pending.device).synchronize()

    def _used_by(self, pending, keys):
        for key in keys:
            self._users[key] = pending

    def _zero(self, tensor):
        """With the engine's own fill the zeroes are ordered like a compute call.  Otherwise
        torch queues the fill on its own stream; the engine computes on others, so the fill has
        to have happened before `compute` is handed the block."""
        if self.zero is not None and tensor.is_cuda and tensor.is_contiguous() \
                and tensor.dim() == 2:
            self.zero(tensor)
            return
        tensor.zero_()
        if tensor.is_cuda:
            import torch
            torch.cuda.current_stream(tensor.device).synchronize()

    # -- the call ----------------------------------------------------------------------------
    def run(self, temperature, pressure, vmr, dst=0, output="gas", async_op=False):
        """Every rank passes the full atmosphere and computes only its own units.

        Args:
            temperature, pressure: [L]; vmr: dict formula -> [L].
            dst: Destination rank, or None: every rank gets the result.
            output: "gas": dict formula -> tensor [L, n]; "total": tensor [L, n], the sum over
                    the molecules (meaningful with scale_density: n k summed over gases,
                    pyLBL/spectroscopy.py:225-234), reduced on the device before it travels.
            async_op: Return a Pending; the exchange runs beside whatever is queued next.

        Returns:
            On rank `dst` (every rank if dst is None) the result, elsewhere None (dict of None
            for "gas"); or a Pending that yields it.  The tensors returned are buffers this
            object keeps and writes again: with one rank the rank's own blocks, with several the
            collected array -- two sets of either, used in turn, so contents are valid until the
            call after next (a collected array above collect_limit/2 is kept ONCE: valid until
            the next call's exchange -- wait for its Pending and use or copy the result before
            calling run() again).  Not waiting is safe for everything but the overwritten
            result: a buffer's next writer is ordered behind the exchange that last used it
            (see the class docstring).
        """
        import torch
        import torch.distributed as dist
        rank, world, backend = _group_info(self.group)
        temperature = np.ascontiguousarray(temperature, dtype=np.float64)
        pressure = np.ascontiguousarray(pressure, dtype=np.float64)
        x = {f: np.ascontiguousarray(vmr[f], dtype=np.float64) for f in self.molecules}
        n_levels, n, m_count = temperature.size, self.n, len(self.molecules)
        plan = partition(n_levels, self.weights, world)
        mine = plan.by_molecule(rank)
        my_levels = plan.levels_of(rank)
        row = {level: i for i, level in enumerate(my_levels)}
        self._turn += 1
        on_device = str(self.device) != "cpu"
        through_host = on_device and world > 1 and backend != "nccl"

        # 1. compute this rank's units into its blocks
        used = []               # keys of the kept buffers this call's exchange reads or writes
        if output == "total":
            # One [levels touched, n] block; every molecule adds to the rows of its levels.
            block = self._buffer("total", (len(my_levels), n))
            used.append(self._last_key)
            first = True
            for m, levels in sorted(mine.items()):
                rows = [row[level] for level in levels]
                lo, hi = rows[0], rows[-1] + 1      # contiguous: units are level-major runs
                assert rows == list(range(lo, hi))
                if lo > 0 or hi < len(my_levels):
                    # This molecule covers only part of the block (unit mode): make sure the
                    # rows it skips are defined before anything adds to them.
                    if first:
                        self._zero(block)
                    self.compute(self.molecules[m], temperature[levels], pressure[levels],
                                 x[self.molecules[m]][levels], block[lo:hi], True)
                else:
                    self.compute(self.molecules[m], temperature[levels], pressure[levels],
                                 x[self.molecules[m]][levels], block, not first)
                first = False
            if first and len(my_levels):
                self._zero(block)
            blocks = {None: block}
        else:
            # One contiguous [levels of this molecule, n] block per molecule.
            blocks = {}
            for m, levels in sorted(mine.items()):
                blocks[m] = self._buffer(("gas", m), (len(levels), n))
                used.append(self._last_key)
                self.compute(self.molecules[m], temperature[levels], pressure[levels],
                             x[self.molecules[m]][levels], blocks[m], False)
        # 2. one rank: done (with async_op the kernels stay queued until wait())
        if world == 1 and not (self.always_exchange and backend is not None):
            if output == "total":
                result = blocks[None]
            else:
                result = {f: blocks[m] for m, f in enumerate(self.molecules)}
            if async_op:
                return Pending([], lambda: result, flush=self.flush)
            if self.flush is not None:
                self.flush()
            return result
        if on_device and self.order is not None and \
                (not through_host or self.order_on_device):
            # RCCL: the exchange is queued behind the kernels on the device; the host goes on
            # (to the next call's kernels: compute k+1 runs beside exchange k).  (Rehearsed with
            # gloo under order_on_device: the staging copies below wait on torch's stream.)
            self.order()
        elif self.flush is not None:
            self.flush()

        if through_host:
            blocks = {key: value.cpu() for key, value in blocks.items()}
            used = []           # the exchange reads host copies made just now, not the kept blocks
        where = "cpu" if (through_host or not on_device) else self.device
        wait_on = None if where == "cpu" else self.device
        sizes = {key: value.numel()*8 for key, value in blocks.items()}

        # 3. exchange
        receivers = range(world) if dst is None else (dst,)
        i_receive = rank in receivers
        if output == "total" and plan.mode == "units":
            # The molecules of one level sit on several ranks: a real sum over ranks.
            if where != "cpu":
                partial = self._buffer("reduce", (n_levels, n), writer="exchange",
                                       sets=2 if 16*n_levels*n <= self.collect_limit else 1)
                used.append(self._last_key)
            else:
                partial = torch.empty((n_levels, n), dtype=torch.float64)
            partial.zero_()
            for i, level in enumerate(my_levels):
                partial[level] = blocks[None][i]
            if dst is None:
                work = dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
            else:
                work = dist.reduce(partial, dst=_global_rank(self.group, dst),
                                   op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            finish = (lambda: partial) if i_receive else (lambda: None)
            pending = Pending([work], finish, device=wait_on, rank=rank,
                              describe="sum over ranks of the per-level totals "
                                       f"({'all_reduce' if dst is None else 'reduce'})",
                              peers=[r for r in range(world) if r != rank],
                              bytes_sent=partial.numel()*8,
                              bytes_received=partial.numel()*8 if i_receive else 0)
            self._used_by(pending, used)
            self.last_exchange = pending
            return pending if async_op else pending.wait()

        # Grouped point-to-point gather: every block goes straight into its final place.
        # The collected array is kept between calls -- two of it, used in turn, unless it is too
        # large for that: "gas" output of BASELINE config 5 is 164 GB on the receiving rank, two
        # would not fit its 288 GB (then ONE: its next writer is ordered behind the exchange still
        # using it, _settle; see run()'s Returns).
        def collected(shape):
            if not i_receive:
                return None
            if where == "cpu":
                return torch.empty(shape, dtype=torch.float64)
            nbytes = 8*int(np.prod(shape))
            tensor = self._buffer("final", shape, writer="exchange",
                                  sets=2 if 2*nbytes <= self.collect_limit else 1)
            used.append(self._last_key)
            return tensor
        if output == "total":
            final = collected((n_levels, n))
            pieces = lambda r: [(None, plan.levels_of(r))]                      # noqa: E731
            place = lambda key, levels: final[levels[0]:levels[-1] + 1]          # noqa: E731
        else:
            final = collected((m_count, n_levels, n))
            pieces = lambda r: sorted(plan.by_molecule(r).items())              # noqa: E731
            place = lambda key, levels: final[key, levels[0]:levels[-1] + 1]     # noqa: E731
        ops, peers, sent, received = [], set(), 0, 0
        for receiver in receivers:
            if receiver == rank:
                for sender in range(world):
                    for key, levels in pieces(sender):
                        if not levels:
                            continue
                        if sender == rank and not (self.always_exchange and backend == "nccl"):
                            place(key, levels).copy_(blocks[key])
                        else:
                            if sender == rank:
                                # (always_exchange over RCCL: this rank's own blocks travel by
                                # a send to itself inside the same group of operations -- the
                                # code and the library a block from another GPU goes through)
                                ops.append(dist.P2POp(dist.isend, blocks[key],
                                                      _global_rank(self.group, rank), self.group))
                                sent += sizes[key]
                            target = place(key, levels)
                            ops.append(dist.P2POp(dist.irecv, target,
                                                  _global_rank(self.group, sender), self.group))
                            peers.add(sender)
                            received += target.numel()*8
            else:
                for key, levels in pieces(rank):
                    if levels:
                        ops.append(dist.P2POp(dist.isend, blocks[key],
                                              _global_rank(self.group, receiver), self.group))
                        peers.add(receiver)
                        sent += sizes[key]
        requests = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            if not i_receive:
                return None if output == "total" else {f: None for f in self.molecules}
            if output == "total":
                return final
            return {f: final[m] for m, f in enumerate(self.molecules)}
        pending = Pending(requests, finish, device=wait_on, rank=rank,
                          describe=f"grouped send/recv of the {output!r} blocks to "
                                   f"{'every rank' if dst is None else f'rank {dst}'}",
                          peers=sorted(peers), bytes_sent=sent, bytes_received=received)
        self._used_by(pending, used)
        self.last_exchange = pending
        return pending if async_op else pending.wait()


def gather_arrays(local, n_levels, dst=0, group=None, device=None):
    """Collects host arrays [levels_local, ...] of a level-sharded call (``level_shard``) on
    rank `dst` (every rank if dst is None): what ``Spectroscopy(group=...)`` uses for its
    result arrays, which live in host memory by the reference's contract.  With the nccl
    backend the blocks are staged through the HBM of GPU `device` (the engine's; torch's current
    device if None -- which is every rank's GPU 0 unless the caller set it), with gloo they
    travel as they are."""
    import torch
    rank, world, backend = _group_info(group)
    if world == 1:
        return local
    tensor = torch.from_numpy(np.ascontiguousarray(local))
    if backend == "nccl":
        tensor = tensor.to(torch.device("cuda", torch.cuda.current_device() if device is None
                                        else int(device)))
    out = gather_levels(tensor, n_levels, dst=dst, group=group)
    return None if out is None else out.cpu().numpy()
