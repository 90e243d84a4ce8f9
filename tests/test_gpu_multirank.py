"""The multi-rank path of bench.py on ONE GPU: `python bench.py --gpus 2` starts its two ranks itself
(`self_launch`, child processes before the parent touches the GPU); with DSG_BENCH_BACKEND=gloo both ranks share cuda:0, so the
rendezvous, the per-rank sharding of the sampling leg, the gradient all-reduce of the training leg and the self-proving fields of the
JSON line all execute -- the only executable evidence of that code this pool can give until an 8-GPU node exists (RCCL itself needs
one device per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, DSG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2048", "--train-batch", "1024", "--steps", "4",
           "--warmup", "1", "--repeats", "3", "--train-steps", "4", "--no-cpu-baseline", "--no-f32-exact", "--no-other-configs"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["ranks_seen"] == 2 and len(d["per_rank_steps_per_s"]) == 2
    assert d["repeats"]["n"] == 3 and d["repeats"]["ms_per_step_min"] <= d["ms_per_step"] <= d["repeats"]["ms_per_step_max"]
    # whole-job value = 2 ranks x K steps / the slowest rank's median call
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    t = d["train"]
    assert t["ranks_seen"] == 2 and len(t["per_rank_samples_per_s"]) == 2 and t["global_batch"] == 2048
    assert t["bucket_checksum_equal"] is True
    assert "gloo" in t["collective"]
    assert d["box"]["before"]["mfma_tflops"] > 100 and d["box"]["before"]["copy_gbs"] > 100
