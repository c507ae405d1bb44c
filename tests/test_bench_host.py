"""bench.py on a box without a GPU: the contract's one JSON line with an "error" field, a non-zero exit code and no
traceback - for the plain N = 1 command, for `--gpus N` without a launcher (bench.py starts its own ranks) and for a
launcher whose WORLD_SIZE disagrees with --gpus (round 4: an AssertionError before any line was printed)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ROCODER_BENCH_REHEARSAL"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def no_gpu_here():
    import torch

    return torch.cuda.device_count() == 0


@pytest.mark.parametrize("argv", [[], ["--gpus", "2", "--steps", "2", "--warmup", "1"], ["--gpus", "4"]])
def test_bench_without_enough_gpus_prints_one_error_line(argv):
    if not no_gpu_here():
        pytest.skip("needs a box without a GPU")
    rc, lines, err = run_bench(argv)
    assert rc != 0
    assert len(lines) == 1, lines
    obj = json.loads(lines[0])
    assert obj["value"] is None and "error" in obj and obj["higher_is_better"] is True
    assert obj["n_gpus"] == (int(argv[1]) if argv else 1)
    assert "Traceback" not in err, err


def test_bench_under_a_launcher_with_another_world_size_reports_it():
    rc, lines, err = run_bench(["--gpus", "2"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                 "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc != 0 and len(lines) == 1
    obj = json.loads(lines[0])
    assert "WORLD_SIZE=1" in obj["error"]
    assert "Traceback" not in err and "AssertionError" not in err


def test_bench_rejects_nonsense_counts():
    rc, lines, _ = run_bench(["--gpus", "0"])
    assert rc != 0 and "error" in json.loads(lines[0])


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGINT", "SIGKILL"])
def test_launcher_stopped_by_a_signal_takes_its_ranks_with_it(tmp_path, sig):
    """the self-started ranks live in their own sessions, so a signal that reaches only the launcher (an
    outer `timeout`, Ctrl-C) used to orphan them. Now SIGTERM / SIGINT / SIGHUP are passed on to every rank's process
    group and the launcher prints its one error line; a SIGKILLed launcher runs no handler, so every rank asks the
    kernel for PR_SET_PDEATHSIG."""
    import signal
    import time

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(ROCODER_BENCH_REHEARSAL="1", ROCODER_BENCH_TEST_RANK_PIDDIR=str(tmp_path))
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        t_end = time.monotonic() + 60
        while len(os.listdir(tmp_path)) < 3 and time.monotonic() < t_end:
            time.sleep(0.05)
        time.sleep(0.2)  # (a pid file exists before its text is flushed)
        pids = [int(open(tmp_path / f).read()) for f in sorted(os.listdir(tmp_path))]
        assert len(pids) == 3
        p.send_signal(getattr(signal, sig))
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    if sig != "SIGKILL":
        lines = [ln for ln in out.splitlines() if ln.strip()]
        assert len(lines) == 1 and "stopped by signal" in json.loads(lines[0])["error"], (lines, err)
        assert p.returncode == 128 + int(getattr(signal, sig))
    t_end = time.monotonic() + 20
    alive = pids
    while alive and time.monotonic() < t_end:
        alive = [q for q in alive if os.path.exists(f"/proc/{q}") and
                 open(f"/proc/{q}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"]
        time.sleep(0.1)
    assert not alive, f"ranks {alive} outlived the launcher"
