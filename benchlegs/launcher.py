"""Starting the N ranks of `python bench.py --gpus N` as child processes, and what tells the
ranks' devices apart."""
import json
import os
import sys
import time


def rccl_libraries():
    """File names of the RCCL libraries mapped into this process (empty: RCCL never loaded)."""
    try:
        with open("/proc/self/maps") as handle:
            return sorted({text.split("/")[-1].strip() for text in handle if "rccl" in text})
    except OSError:
        return []


def device_identity(torch, index):
    """What tells two GPUs apart: name, UUID and PCI address of HIP device `index` (each only
    where this torch exposes it)."""
    out = {"device_index": index}
    try:
        props = torch.cuda.get_device_properties(index)
    except Exception as error:          # diagnostics must not stop the run
        return dict(out, error=str(error))
    out["name"] = props.name
    for key in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id", "gcnArchName",
                "multi_processor_count"):
        value = getattr(props, key, None)
        if value is not None:
            out[key] = str(value) if key == "uuid" else value
    out["hip_visible_devices"] = os.environ.get("HIP_VISIBLE_DEVICES")
    return out


def launch_ranks(args, command=None):
    """`python bench.py --gpus N` with no launcher around it: starts the N ranks as CHILD
    processes (what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1` would start: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, one process per GPU), relays rank 0's JSON line on stdout and the other
    ranks' output on stderr, and returns the worst exit code.  The first rank that fails ends
    the others (a rank blocked in a collective cannot be woken), and so do --launch-timeout
    and a signal to this process.  Called before torch is imported or HIP touched: a process
    that holds a GPU must never start or become another program."""
    import signal
    import socket
    import subprocess
    import threading

    n = args.gpus
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(port), "GROUP_RANK": "0", "ROLE_RANK": "0", "NODE_RANK": "0",
                 "PYLBL_BENCH_LAUNCHER": "bench.py"})
    base.setdefault("OMP_NUM_THREADS", "1")             # as torch.distributed.run does
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs here
    if command is None:             # (tests pass a stand-in for the ranks' program)
        command = [sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:]
    children, relays = [], []

    def relay(stream, target, prefix):
        for text in stream:
            target.write(prefix + text)
            target.flush()

    for rank in range(n):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        child = subprocess.Popen(command, env=env, stdout=subprocess.PIPE, text=True,
                                 start_new_session=True, cwd=os.getcwd())
        children.append(child)
        target, prefix = (sys.stdout, "") if rank == 0 else (sys.stderr, f"[rank {rank}] ")
        thread = threading.Thread(target=relay, args=(child.stdout, target, prefix), daemon=True)
        thread.start()
        relays.append(thread)

    def stop(sig):
        for child in children:
            if child.poll() is None:
                try:
                    os.killpg(child.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(number, frame):
        stop(signal.SIGTERM)
        time.sleep(2.)
        stop(signal.SIGKILL)
        os._exit(128 + number)
    for number in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(number, on_signal)

    deadline = time.monotonic() + args.launch_timeout
    reason = None
    own = []            # exit codes of the ranks that left by themselves
    while True:
        codes = [child.poll() for child in children]
        if all(code is not None for code in codes):
            break
        failed = [r for r, code in enumerate(codes) if code not in (None, 0)]
        if failed:
            reason = f"rank {failed[0]} left with code {codes[failed[0]]}"
        elif time.monotonic() > deadline:
            reason = f"no result after --launch-timeout {args.launch_timeout:g} s"
        if reason:
            # Let the others print what they were doing: until they have all left, or the grace
            # period is over.
            patience = time.monotonic() + args.launch_grace
            while time.monotonic() < patience and any(c.poll() is None for c in children):
                time.sleep(0.05)
            own = [code for code in (child.poll() for child in children) if code is not None]
            stop(signal.SIGTERM)
            patience = time.monotonic() + min(3., args.launch_grace)
            while time.monotonic() < patience and any(c.poll() is None for c in children):
                time.sleep(0.05)
            stop(signal.SIGKILL)
            for child in children:
                child.wait()
            break
        time.sleep(0.05)
    for thread in relays:
        thread.join(timeout=5.)
    codes = [child.returncode for child in children]
    if not reason:
        own = codes
        failed = [r for r, code in enumerate(codes) if code != 0]
        if failed:          # (every rank had left between two looks at them)
            reason = f"rank {failed[0]} left with code {codes[failed[0]]}"
    # The worst code among the ranks that left by themselves (the ones this launcher ended do not
    # count); 124, like timeout(1), when time ran out with none of them having failed.
    worst = max([(128 - code if code < 0 else code) for code in own] or [0])
    if reason:
        print(json.dumps({"bench_failed": True, "launcher": True, "reason": reason,
                          "exit_codes": codes}), file=sys.stderr, flush=True)
        worst = worst or (124 if "launch-timeout" in reason else 1)
    return worst

