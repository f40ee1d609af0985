"""One process per GPU, started from a plain `python script.py --gpus N` (reference: vmc_fluids/mpi_wrapper.py:9-11, where the ranks
come from `mpirun` and every process reads `MPI.COMM_WORLD` at import).

Here a rank is a torch.distributed process (backend "nccl" = RCCL over xGMI).  `run_ranks` is the part of `mpirun` this needs: it
starts N FRESH interpreter processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays what rank 0 prints
on stdout, and makes a failure of any rank the failure of the whole job (the survivors would otherwise wait in a collective for
ever).  The launching process itself never touches the GPU -- it only imports this file's standard-library dependencies -- and
nothing is exec'ed from a process that has: every rank is a child started before any HIP call.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time

RANK_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")


def under_launcher(env=None) -> bool:
    """True inside a rank process (ours or torch.distributed.run's): RANK and WORLD_SIZE are set."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def free_port() -> int:
    """A TCP port nobody listens on right now, on the loopback interface the rendezvous uses."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_environment(rank: int, world: int, port: int, base=None, local_rank=None) -> dict:
    """Environment of rank `rank` of `world` on one node: the variables torch.distributed's env:// rendezvous reads, loopback
    rendezvous address (the container hostname may not resolve), dmabuf IPC for RCCL on this pool."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank if local_rank is None else local_rank), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _stop(procs, grace=10.0):
    """End the ranks still running: SIGTERM to each PID we started, SIGKILL after `grace` seconds."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.send_signal(signal.SIGTERM)
        except ProcessLookupError:
            pass
    t_end = time.monotonic() + grace
    for p in live:
        try:
            p.wait(max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except ProcessLookupError:
                pass
            p.wait()


def run_ranks(argv, world: int, *, port=None, timeout=None, base_env=None, poll=0.1, relay=None, log=None):
    """Start `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "8"]) once per rank and wait.

    Returns (returncode, rank0_stdout): 0 only when EVERY rank exited with 0; otherwise the first non-zero exit code seen (124 for a
    timeout), after the other ranks have been stopped.  Rank 0's stdout is collected (and passed line by line to `relay` if
    given); the other ranks' stdout goes to this process's stderr so that rank 0's record stays the only thing on stdout."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    port = free_port() if port is None else int(port)
    log = (lambda msg: print(msg, file=sys.stderr, flush=True)) if log is None else log
    procs, lines = [], []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_environment(r, world, port, base_env),
                                          stdout=subprocess.PIPE if r == 0 else 2, stderr=None, text=True if r == 0 else None))
    except Exception:
        _stop(procs)
        raise

    def pump():
        for line in procs[0].stdout:
            lines.append(line)
            if relay is not None:
                relay(line)

    reader = threading.Thread(target=pump, daemon=True)
    reader.start()
    t_end = None if timeout is None else time.monotonic() + float(timeout)
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]          # (a rank ended by signal s: 128 + s, as a shell reports it)
            log(f"launch: rank {bad[0][0]} exited with {bad[0][1]}; stopping the other ranks")
            _stop(procs)
            break
        if all(c == 0 for c in codes):
            break
        if t_end is not None and time.monotonic() > t_end:
            rc = 124
            log(f"launch: {world} ranks still running after {timeout:.0f} s; stopping them")
            _stop(procs)
            break
        time.sleep(poll)
    reader.join(5.0)
    return rc, "".join(lines)
