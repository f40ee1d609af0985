"""One process per GPU, started from a plain `python script.py --gpus N` (reference: vmc_fluids/mpi_wrapper.py:9-11, where the ranks
come from `mpirun` and every process reads `MPI.COMM_WORLD` at import).

Here a rank is a torch.distributed process (backend "nccl" = RCCL over xGMI).  `run_ranks` is the part of `mpirun` this needs: it
starts N FRESH interpreter processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays what rank 0 prints
on stdout, and makes a failure of any rank the failure of the whole job (the survivors would otherwise wait in a collective for
ever) -- and so does the end of the launcher itself: a signal, an exception or a kill of the launching process stops every rank.  The
launching process itself never touches the GPU -- it only imports this file's standard-library dependencies -- and
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


def _signal_rank(p, sig):
    """`sig` to rank `p` and whatever it started: every rank is the leader of a session of its own (start_new_session), so its process
    group is exactly its descendants -- never this process, never a sibling."""
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError):
        try:
            p.send_signal(sig)
        except ProcessLookupError:
            pass


def _peek(p):
    """Exit status of rank `p` WITHOUT reaping it (None while it runs): a reaped leader frees its pid, and with it the id of its process group,
    for re-use -- a later SIGKILL to that group could then hit an unrelated one.  Ranks stay zombies until `_reap_all` has signalled their
    groups; only then are they waited for."""
    if p.returncode is not None:
        return p.returncode
    try:
        info = os.waitid(os.P_PID, p.pid, os.WEXITED | os.WNOHANG | os.WNOWAIT)
    except ChildProcessError:                         # (reaped elsewhere: Popen knows the code)
        return p.poll()
    if info is None:
        return None
    return info.si_status if info.si_code == os.CLD_EXITED else -info.si_status


def _reap_all(procs):
    """SIGKILL to every rank's process group while its leader is still unreaped (the group id cannot have been recycled), then reap."""
    for p in procs:
        if p.returncode is None:
            _signal_rank(p, signal.SIGKILL)
    for p in procs:
        p.wait()


def _stop(procs, grace=10.0):
    """End the ranks still running: SIGTERM to each rank's process group, SIGKILL after `grace` seconds; the groups of ranks that have
    already exited get the SIGKILL too (a helper process a rank left behind must not keep a GPU) -- all of it before any rank is reaped."""
    live = [p for p in procs if _peek(p) is None]
    for p in live:
        _signal_rank(p, signal.SIGTERM)
    t_end = time.monotonic() + grace
    while any(_peek(p) is None for p in live) and time.monotonic() < t_end:
        time.sleep(0.02)
    _reap_all(procs)


class LauncherSignalled(BaseException):
    """SIGTERM / SIGINT / SIGHUP reached the launching process while its ranks ran (raised from the handler `run_ranks` installs, so that
    the wait loop's `finally` stops the ranks before the launcher goes)."""
    def __init__(self, signum):
        super().__init__(signum)
        self.signum = signum


def _die_with_parent():
    """In the child, before exec: SIGKILL when the launcher dies without having stopped its ranks (a SIGKILL to the launcher cannot be
    handled; an orphaned rank would wait in a collective and hold its GPU).  Linux prctl(PR_SET_PDEATHSIG); best effort."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)
    except Exception:
        pass


def run_ranks(argv, world: int, *, port=None, timeout=None, base_env=None, poll=0.1, relay=None, log=None, stop_grace=10.0):
    """Start `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "8"]) once per rank and wait.

    Returns (returncode, rank0_stdout): 0 only when EVERY rank exited with 0; otherwise the first non-zero exit code seen (124 for a
    timeout, 125 when rank 0's record could not be relayed), after the other ranks have been stopped.  Rank 0's stdout is collected (and
    passed line by line to `relay` if given); the other ranks' stdout goes to this process's stderr so that rank 0's record stays the only
    thing on stdout.

    No rank outlives this call, however it ends: the wait loop runs under try/finally; SIGTERM, SIGINT and SIGHUP to the launcher are
    turned into `LauncherSignalled` for its duration (main thread only; previous handlers restored), so a `timeout -k` around the job or
    a Ctrl-C stops the ranks first (5 s grace, inside timeout's usual 10 s) and then re-raises; every rank is a session of its own and
    is signalled as a process group; and a launcher that is SIGKILLed takes its ranks with it (PR_SET_PDEATHSIG)."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    port = free_port() if port is None else int(port)
    log = (lambda msg: print(msg, file=sys.stderr, flush=True)) if log is None else log
    procs, lines = [], []
    relay_failed = []

    def on_signal(signum, frame):
        raise LauncherSignalled(signum)

    previous = {}
    if threading.current_thread() is threading.main_thread():
        for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            previous[s] = signal.signal(s, on_signal)

    def pump():
        # (never stops reading: a full 64 KB pipe would block rank 0 in write() and the other ranks in their next collective)
        for line in procs[0].stdout:
            lines.append(line)
            if relay is not None and not relay_failed:
                try:
                    relay(line)
                except BaseException as ex:                      # e.g. BrokenPipeError of `bench.py --gpus 2 | head -1`
                    relay_failed.append(ex)

    reader = None
    rc = 0
    clean = False
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_environment(r, world, port, base_env), start_new_session=True,
                                          preexec_fn=_die_with_parent,
                                          stdout=subprocess.PIPE if r == 0 else 2, stderr=None, text=True if r == 0 else None))
        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
        t_end = None if timeout is None else time.monotonic() + float(timeout)
        while True:
            codes = [_peek(p) for p in procs]                                  # (no rank is reaped before its group has been signalled)
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]          # (a rank ended by signal s: 128 + s, as a shell reports it)
                log(f"launch: rank {bad[0][0]} exited with {bad[0][1]}; stopping the other ranks")
                break
            if all(c == 0 for c in codes):
                clean = True
                break
            if relay_failed:
                rc = 125
                log(f"launch: rank 0's output can no longer be relayed ({relay_failed[0]!r}); stopping the ranks")
                break
            if t_end is not None and time.monotonic() > t_end:
                rc = 124
                log(f"launch: {world} ranks still running after {timeout:.0f} s; stopping them")
                break
            time.sleep(poll)
    except LauncherSignalled as ex:
        log(f"launch: signal {ex.signum} received; stopping {sum(_peek(p) is None for p in procs)} running ranks")
        stop_grace = min(stop_grace, 5.0)
        raise
    finally:
        try:
            if not clean:
                _stop(procs, stop_grace)
            else:
                _reap_all(procs)          # (every rank exited 0: what it may have left behind in its group goes, then the ranks are reaped)
            if reader is not None:
                # all ranks have exited, so the pipe is normally at end of file and the whole record is read before we return; a helper that
                # inherited rank 0's stdout and left its session (setsid) can still hold the pipe open -- do not wait for it for ever
                reader.join(5.0)
                if reader.is_alive():
                    log("launch: rank 0's stdout is still held open by a process outside its group; returning what was read")
                    try:
                        procs[0].stdout.close()
                    except Exception:
                        pass
        finally:
            for s, h in previous.items():
                signal.signal(s, h)
    if clean and relay_failed:
        rc = 125
    return rc, "".join(lines)
