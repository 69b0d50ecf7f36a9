"""CPU: bench.py's own launcher (`python3 bench.py --gpus N` without torchrun around it).
The parent must not import torch or the package (nothing that could touch a GPU): it prints /
starts N child environments.  `--launch-dry-run` shows them without starting anything."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _dry(*argv, env=None):
    e = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=e, timeout=60)
    return p


def test_dry_run_prints_four_child_environments():
    p = _dry("--gpus", "4", "--steps", "20", "--warmup", "5", "--launch-dry-run")
    assert p.returncode == 0, p.stderr.decode()
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1                      # ONE JSON line
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 4 and len(doc["ranks"]) == 4
    ports = set()
    for r, env in enumerate(doc["ranks"]):
        assert env["RANK"] == env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1
    # the children get the caller's arguments, without the dry-run switch
    assert doc["cmd"][1] == BENCH and doc["cmd"][2:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]


def test_launcher_decides_before_anything_heavy_is_imported():
    """The launcher path runs with torch made unimportable: the parent never needs it."""
    blocker = os.path.join(ROOT, "tests", "_no_torch")
    os.makedirs(blocker, exist_ok=True)
    with open(os.path.join(blocker, "torch.py"), "w") as f:
        f.write("raise ImportError('the launcher must not import torch')\n")
    try:
        p = _dry("--gpus", "8", "--global-batch", "512", "--config", "C5", "--launch-dry-run",
                 env={"PYTHONPATH": blocker})
        assert p.returncode == 0, p.stderr.decode()
        doc = json.loads(p.stdout.decode())
        assert len(doc["ranks"]) == 8 and "--global-batch" in doc["cmd"]
        # one rank + --force-dist goes through the same spawn path (the one-GPU rehearsal)
        p = _dry("--gpus", "1", "--force-dist", "--launch-dry-run", env={"PYTHONPATH": blocker})
        assert p.returncode == 0 and len(json.loads(p.stdout.decode())["ranks"]) == 1
    finally:
        os.remove(os.path.join(blocker, "torch.py"))
        os.rmdir(blocker)


def test_a_rank_environment_is_not_relaunched():
    """Under torch.distributed.run (WORLD_SIZE set) the process IS a rank: no launcher.  A
    mismatching --gpus is refused before any GPU work."""
    p = _dry("--gpus", "2", "--launch-dry-run",
             env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and b"WORLD_SIZE=4" in p.stderr
