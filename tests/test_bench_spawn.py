"""`python bench.py --gpus N` must start its own N ranks when no launcher did (it used to run ONE rank and print
n_gpus 1).  The launcher is exercised here without a GPU: --spawn-dry-run makes every self-spawned rank join a gloo group on CPU instead
of running the engine; rank 0's single JSON line is the command's whole stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_n_spawns_n_ranks():
    for n in (2, 3):
        r = _run("--gpus", str(n), "--spawn-dry-run")
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout  # ONE line, rank 0's
        out = json.loads(lines[0])
        assert out["dry_run"] is True and out["n_gpus"] == n and out["ranks"] == n and out["rank_sum"] == n * (n + 1) // 2


def test_too_few_devices_is_loud():
    """The real path: more ranks asked for than devices visible => nothing is started, exit code 2, the reason on stderr."""
    r = _run("--gpus", "64")
    assert r.returncode == 2 and "device(s) visible" in r.stderr and r.stdout.strip() == ""


def test_a_launcher_environment_is_respected():
    """Under torch.distributed.run (WORLD_SIZE set) the process is a rank and must not spawn; a mismatch with --gpus is an error."""
    r = _run("--gpus", "2", "--spawn-dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
