"""The N>1 launch contract of bench.py (one process per GPU, barrier, MAX over ranks, one
all-gather of timings) exercised on CPU with the gloo backend, world_size 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "5", "--warmup", "1", "--cpu-dry-run"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    return json.loads(lines[0])


def test_two_ranks_gloo():
    out = _run(2, 29641)
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    # per-step sleep is 2 ms on rank 0 and 4 ms on rank 1: MAX over ranks must see the slow one
    assert out["ms_per_step"] >= 3.9
    # whole-job aggregate: both ranks' units over the max time
    assert abs(out["value"] - 2 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    assert out["per_rank_ms"] == [[1.0, 2.0, 3.0], [2.0, 3.0, 4.0]]  # the single timing all-gather
    # the scaling curve's own denominator, measured in the same run (rank 0 alone between two barriers), and the fields that make a
    # slow box visible in the line
    assert abs(out["per_gpu_value"] - out["value"] / 2) < 1e-9 * out["value"]
    assert out["c5_shape_per_gpu"]["value"] > 0 and out["c5_shape_per_gpu"]["steps"] == 3
    assert abs(out["efficiency_vs_c5_shape_per_gpu"] - out["per_gpu_value"] / out["c5_shape_per_gpu"]["value"]) < 1e-9
    # rank 0 alone sleeps 2 ms per step, the pair is paced by rank 1's 4 ms: the dry run's "efficiency" is about one half
    assert 0.3 < out["efficiency_vs_c5_shape_per_gpu"] < 0.7
    for key in ("ms_per_factorize", "ms_per_factorize_min", "ms_per_factorize_median"):
        assert key in out
    assert set(out["clocks"]) == {"before_timed_region", "after_timed_region"}


def test_single_rank_no_process_group():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-dry-run", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["metric"].startswith("IP iterations/sec")


def test_self_spawn_without_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own two ranks
    (the command the driver would issue must not exit with "launch with torch.distributed.run")
    and defaults to BASELINE config C5's 16 instances per GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-dry-run", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["batch_per_gpu"] == 16
    assert out["per_rank_ms"] == [[1.0, 2.0, 3.0], [2.0, 3.0, 4.0]]


def test_batch_defaults():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        assert bench.parse_args().batch == 1
        sys.argv = ["bench.py", "--gpus", "8"]
        assert bench.parse_args().batch == 16
        sys.argv = ["bench.py", "--gpus", "8", "--batch", "4"]
        assert bench.parse_args().batch == 4
    finally:
        sys.argv = old
