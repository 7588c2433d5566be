"""`python bench.py --gpus N` must start its own ranks (round-1 finding: it asserted on WORLD_SIZE instead).  The
launcher path runs here without a GPU: `--launch-check` makes the ranks rendezvous over gloo and report the shards
the timed run would build from."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_shards_the_default_multi_gpu_workload():
    out = _run(["--gpus", "2", "--launch-check", "--points", "300000"])
    assert out["n_gpus"] == 2 and out["workload"] == "S3" and out["mode"] == "owner"
    (b0, n0, j0), (b1, n1, j1) = out["shards"]
    assert j0 == j1 == 300000                     # point 0 is the origin and is not binned (receiver.cpp:145, 150)
    assert b0 == 0 and b1 == n0 and n0 + n1 == 300000   # contiguous index ranges that tile the binned points
    # the N > 1 line's extra keys, and the cloud rank 0 builds alone for the single-GPU anchor: the very cloud the ranks share
    assert out["multi_gpu_keys"] == ["single_gpu_anchor", "speedup_vs_single_gpu", "gather_ms", "modes", "exchange", "comm_selftest"]
    assert out["anchor_points"] == 300000 and out["anchor_is_the_ranks_cloud"] is True


def test_self_launch_global_mode_without_anchor():
    out = _run(["--gpus", "2", "--launch-check", "--points", "100000", "--mode", "global", "--no-anchor"])
    assert out["mode"] == "global" and out["multi_gpu_keys"] == ["exchange", "comm_selftest"] and out["anchor_points"] is None


def test_self_launch_replicas_mode_gives_each_rank_its_own_cloud():
    out = _run(["--gpus", "2", "--launch-check", "--points", "200000", "--mode", "replicas", "--workload", "S2"])
    assert out["mode"] == "replicas"
    assert [s[1] for s in out["shards"]] == [200000, 200000] and all(s[0] == 0 for s in out["shards"])
