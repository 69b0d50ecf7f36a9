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


def _launch(n, script):
    """bench.launcher with stand-in ranks (a python -c script each): what the parent forwards."""
    import importlib.util
    import io
    import contextlib
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)          # (imported, not run: the launcher check needs __main__)
    args = bench.parse_args(["--gpus", str(n)])
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = bench.launcher(args, ["--gpus", str(n)], cmd=[sys.executable, "-c", script])
    return rc, out.getvalue()


def test_launcher_forwards_rank_zero_and_the_first_failure():
    """Three stand-in ranks: the parent prints rank 0's LAST non-empty stdout line (the JSON
    line; other ranks' stdout goes to stderr), hands every child its own rank environment, and
    returns the first non-zero exit code of any rank."""
    ok = ("import os, json, sys; r = os.environ['RANK']; print('noise from rank', r); "
          "print(json.dumps({'rank': r, 'world': os.environ['WORLD_SIZE'], "
          "'addr': os.environ['MASTER_ADDR'], 'local': os.environ['LOCAL_RANK']}))")
    rc, text = _launch(3, ok)
    assert rc == 0
    lines = [l for l in text.splitlines() if l.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"rank": "0", "world": "3", "addr": "127.0.0.1", "local": "0"}
    bad = ok + "; sys.exit(7 if r == '2' else 0)"
    rc, text = _launch(3, bad)
    assert rc == 7 and text.strip() == ""       # a failed job prints no line
    silent = "import sys; sys.exit(0)"
    rc, text = _launch(2, silent)
    assert rc == 1 and text.strip() == ""       # rank 0 printed no JSON line: an error, not a success
