"""-m gpu: the collective library itself (RCCL behind torch.distributed's "nccl" backend) on the one GPU of the test box, and
the self-launching multi-rank bench.  Two ranks cannot share one GPU under RCCL, so: (i) a ONE-rank
RCCL group runs the collectives and GradBuckets' stream ordering against RCCL's own stream; (ii) `python3 bench.py --gpus 2`
runs its two self-started ranks on the one GPU over gloo (control flow, rc and line relay; never a measurement)."""
import json
import os
import subprocess
import sys

import pytest

from agplace_amd import launcher

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_collectives_and_gradbuckets_stream_ordering(dev):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_single_rank.py"), str(launcher.free_port())],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["allreduce_identity"] and rec["allgather_identity"]
    assert rec["nbuckets"] >= 3 and rec["gradbuckets_values_after_async_allreduce"] and rec["handles_waited"]
    # the bench's real training step under the exchange: learned bucket order, no bucket held back, no zeros shipped
    ge = rec["train_grad_exchange"]
    print("GRAD_EXCHANGE", ge, rec["train_ms_per_step"])
    assert ge["forced_last"] == 0 and ge["buckets"] >= 2 and ge["launched_before_finish"] >= ge["buckets"] - 1
    assert ge["zero_bytes"] == 0 and ge["excluded_bytes"] > 25e6 and ge["bytes"] < 36e6


def test_bench_gpus2_self_launch_on_one_gpu_over_gloo(dev):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(AGP_DIST_BACKEND="gloo", AGP_LOCAL_DEVICE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-knn", "--train-steps", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8
    assert rec["rccl"]["world"] == 2 and rec["rccl"]["allreduce_ones"] == 2.0 and rec["rccl"]["backend"] == "gloo"
    assert rec["config"]["replay_equals_eager"] is True
    assert rec["config"]["steps_in_flight"] == 2 and rec["config"]["ms_per_step_one_in_flight"] > 0      # the default: two steps in flight


def test_sync_batchnorm_two_ranks_equal_one_rank_whole_batch(dev, capfd):
    """parallel.enable_sync_batchnorm(): two ranks (on the one GPU, over gloo), half a batch each, against ONE rank on the whole
    batch -- embeddings, rank-averaged gradients of every parameter, running statistics; 2-D trunks and the sparse branch (whose
    row counts differ per rank).  Without it the same split differs by 8e-2 (test_per_rank_batchnorm_differs_...)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    import io
    buf = io.StringIO()
    rc = launcher.launch_ranks(os.path.join(ROOT, "tests", "helpers", "syncbn_ranks.py"), [], 2, env=env, out=buf, timeout_s=900)
    assert rc == 0, buf.getvalue()[-2000:]
    rec = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    print("SYNCBN", rec)
    assert rec["world"] == 2
    assert rec["outputs"] < 2e-5 and rec["grads_worst_over_bound"] < 1.0 and rec["grads_checked"] > 40
    assert rec["running_mean"] < 1e-5 and rec["running_var"] < 1e-5
    assert rec["sparse_outputs"] < 2e-5 and rec["sparse_running_mean"] < 1e-5
