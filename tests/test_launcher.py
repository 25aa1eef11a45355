"""`python3 bench.py --gpus N` starts its own ranks: the launcher's environment, line relay and exit
code, on CPU with a gloo stand-in rank program; bench.py's strict WORLD_SIZE check."""
import io
import json
import os
import subprocess
import sys

from agplace_amd import launcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "helpers", "rank_script.py")


def test_launcher_relays_rank0_line_and_returns_zero():
    out = io.StringIO()
    rc = launcher.launch_ranks(SCRIPT, ["--gpus", "2"], 2, out=out)
    assert rc == 0
    # (gloo announces its connections on stdout; RCCL does not)
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, lines                       # only rank 0's stdout is relayed
    rec = json.loads(lines[0])
    assert rec == {"world": 2, "allreduce_ones": 2.0, "local": 0}


def test_launcher_returns_worst_child_exit_code():
    out = io.StringIO()
    rc = launcher.launch_ranks(SCRIPT, ["--gpus", "2", "--fail-rank", "1", "--rc", "7"], 2, out=out)
    assert rc == 7
    assert json.loads([ln for ln in out.getvalue().splitlines() if ln.startswith("{")][0])["world"] == 2   # rank 0's line got out


def test_launcher_terminates_survivors_of_a_dead_rank():
    # rank 1 dies before the rendezvous (bad argument): rank 0 would wait for it forever; the launcher ends it
    out = io.StringIO()
    env = dict(os.environ)
    rc = launcher.launch_ranks(SCRIPT, ["--gpus", "3"], 2, env=env, out=out, grace_s=2.0)   # world 2 != --gpus 3: assertion
    assert rc != 0


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "WORLD_SIZE=3" in p.stderr and p.stdout.strip() == ""


def test_bench_parent_of_a_multi_rank_run_never_initialises_the_gpu():
    """No GPU here: the child ranks fail their `needs a GPU` assertion; the parent must come back with their failure
    code instead of hanging, and must not print a line of its own."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AGP_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode not in (0, 2), (p.returncode, p.stderr[-800:])
        assert "needs a GPU" in p.stderr
        assert not any(ln.startswith("{") for ln in p.stdout.splitlines())
