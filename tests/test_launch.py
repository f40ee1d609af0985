"""CPU tests of the rank launcher behind `python bench.py --gpus N` (vmc_pde_amd/launch.py; reference: mpi_wrapper.py:9-11, where the
ranks come from mpirun): environment of a rank, free rendezvous port, relay of rank 0's record, and that one failing rank fails the
job instead of leaving the others waiting in a collective."""
import os
import socket
import sys
import time

import pytest

from vmc_pde_amd import launch


def test_rank_environment_and_port():
    port = launch.free_port()
    with socket.socket() as s:                       # free right now: we can bind it ourselves
        s.bind(("127.0.0.1", port))
    env = launch.rank_environment(3, 8, port, base={"PATH": "/bin", "RANK": "9"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == str(port)     # loopback: the container hostname may not resolve
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert launch.rank_environment(0, 2, port, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # the user's choice stands
    assert launch.rank_environment(1, 2, port, base={}, local_rank=0)["LOCAL_RANK"] == "0"
    with pytest.raises(ValueError):
        launch.rank_environment(2, 2, port)
    assert launch.under_launcher({"RANK": "0", "WORLD_SIZE": "2"}) and not launch.under_launcher({"WORLD_SIZE": "2"}) and not launch.under_launcher({})


def test_all_ranks_start_and_rank0_is_relayed():
    code = ("import os, sys; r = int(os.environ['RANK']); "
            "print('rank', r, os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']); "
            "print('{\"value\": 1}') if r == 0 else None")
    seen = []
    rc, out = launch.run_ranks([sys.executable, "-c", code], 3, relay=seen.append, log=lambda m: None)
    assert rc == 0
    lines = out.splitlines()
    assert lines[0].startswith("rank 0 3 0 127.0.0.1 ") and lines[1] == '{"value": 1}' and len(lines) == 2      # rank 0 only: the record stays alone on stdout
    assert "".join(seen) == out


def test_a_failing_rank_fails_the_job_and_stops_the_others():
    # rank 1 exits with 3 at once; rank 0 and 2 would "wait in a collective" for a minute
    code = "import os, sys, time; r = int(os.environ['RANK']); sys.exit(3) if r == 1 else time.sleep(60)"
    msgs = []
    t0 = time.monotonic()
    rc, out = launch.run_ranks([sys.executable, "-c", code], 3, log=msgs.append)
    assert rc == 3 and time.monotonic() - t0 < 30
    assert any("rank 1 exited with 3" in m for m in msgs)


def test_a_rank_killed_by_a_signal_is_reported_like_a_shell_does():
    code = "import os, signal, time; r = int(os.environ['RANK']); os.kill(os.getpid(), signal.SIGKILL) if r == 0 else time.sleep(60)"
    rc, _ = launch.run_ranks([sys.executable, "-c", code], 2, log=lambda m: None)
    assert rc == 128 + 9


def test_timeout_stops_every_rank():
    t0 = time.monotonic()
    rc, _ = launch.run_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, timeout=1.0, log=lambda m: None)
    assert rc == 124 and time.monotonic() - t0 < 30


def test_bench_becomes_the_launcher_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus 2` outside a launcher: the parent's whole job is run_ranks on its own command line (no torch.cuda call,
    no distributed initialisation in that process), and its exit code is the job's."""
    import bench
    calls = {}

    def fake_run(argv, world, **kw):
        calls.update(argv=argv, world=world, kw=kw)
        kw["relay"]('{"metric": "x"}\n')
        return 0, '{"metric": "x"}\n'

    def no_gpu(*a, **k):
        raise AssertionError("the launching process touched the GPU")

    monkeypatch.setattr(launch, "run_ranks", fake_run)
    monkeypatch.setattr(bench.torch.cuda, "set_device", no_gpu)
    monkeypatch.setattr(bench.torch.cuda, "device_count", no_gpu)
    monkeypatch.setattr(bench.dist, "init_process_group", no_gpu)
    for k in launch.RANK_VARS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--workload", "C4"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    assert calls["world"] == 2 and calls["argv"][0] == sys.executable and calls["argv"][1].endswith("bench.py")
    assert calls["argv"][2:] == ["--gpus", "2", "--steps", "3", "--workload", "C4"]
    # a failing rank is the parent's exit code; a clean exit without a record is a failure too
    monkeypatch.setattr(launch, "run_ranks", lambda argv, world, **kw: (7, ""))
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    monkeypatch.setattr(launch, "run_ranks", lambda argv, world, **kw: (0, "no record here\n"))
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 1


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # (a zombie still answers kill 0: it is gone once its state says so)
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except FileNotFoundError:
        return False


LAUNCHER = ("import sys, time; sys.path.insert(0, {root!r}); from vmc_pde_amd import launch\n"
            "code = 'import os, sys, time, subprocess; "
            "h = subprocess.Popen([sys.executable, \"-c\", \"import time; time.sleep(120)\"]); "
            "print(os.getpid(), h.pid, flush=True); time.sleep(120)'\n"
            "try:\n"
            "    rc, _ = launch.run_ranks([sys.executable, '-c', code], 2, relay=lambda l: print(l, end='', flush=True), base_env=None)\n"
            "except launch.LauncherSignalled as ex:\n"
            "    sys.exit(128 + ex.signum)\n"
            "sys.exit(rc)\n")


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGINT", "SIGKILL"])
def test_no_rank_survives_the_launcher(sig, tmp_path):
    """a SIGTERM (`timeout -k` around the job, the driver's time limit), a Ctrl-C or a SIGKILL to the launching process must
    not leave ranks behind holding their GPUs -- nor helper processes the ranks started (each rank is signalled as a process group)."""
    import signal
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # ranks 0 and 1 both print "pid helper_pid"; rank 0's line is relayed on stdout, rank 1's goes to stderr
    p = subprocess.Popen([sys.executable, "-c", LAUNCHER.format(root=root)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line0 = p.stdout.readline().split()
    line1 = p.stderr.readline().split()
    pids = [int(x) for x in line0 + line1]
    assert len(pids) == 4 and all(_alive(q) for q in pids)
    t0 = time.monotonic()
    p.send_signal(getattr(signal, sig))
    rc = p.wait(30)
    assert rc == (-9 if sig == "SIGKILL" else 128 + int(getattr(signal, sig)))
    ranks = [pids[0], pids[2]]
    t_end = time.monotonic() + 10
    while time.monotonic() < t_end and any(_alive(q) for q in (ranks if sig == "SIGKILL" else pids)):
        time.sleep(0.05)
    assert not any(_alive(q) for q in ranks), "a rank outlived its launcher"
    if sig != "SIGKILL":                               # (a killed launcher cannot reap the ranks' own children: the ranks die by PDEATHSIG)
        assert not any(_alive(q) for q in pids), "a rank's helper process outlived the job"
    else:
        for q in pids:
            try:
                os.kill(q, signal.SIGKILL)
            except ProcessLookupError:
                pass
    assert time.monotonic() - t0 < 25


def test_a_relay_that_breaks_stops_the_job_instead_of_blocking_rank0():
    """`bench.py --gpus 2 | head -1` -- the relay raises BrokenPipeError; rank 0 must not block on a full pipe (the pump keeps
    draining) and the job ends with a non-zero code instead of hanging until the timeout."""
    code = ("import os, sys, time\n"
            "if os.environ['RANK'] == '0':\n"
            "    for i in range(20000): print('x' * 100)\n"          # 2 MB: far more than a pipe buffer
            "    sys.stdout.flush()\n"
            "time.sleep(60)\n")
    n = []

    def relay(line):
        n.append(1)
        if len(n) > 3:
            raise BrokenPipeError("stdout closed")

    msgs = []
    t0 = time.monotonic()
    rc, out = launch.run_ranks([sys.executable, "-c", code], 2, relay=relay, log=msgs.append, stop_grace=2.0)
    assert rc == 125 and time.monotonic() - t0 < 30
    assert any("can no longer be relayed" in m for m in msgs)


def test_the_whole_record_is_returned_even_when_it_is_long():
    code = "import os\nif os.environ['RANK'] == '0':\n    for i in range(30000): print(i)\n"
    rc, out = launch.run_ranks([sys.executable, "-c", code], 2, log=lambda m: None)
    assert rc == 0 and out.splitlines()[-1] == "29999" and len(out.splitlines()) == 30000
