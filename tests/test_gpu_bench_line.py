"""GPU tier: the N > 1 line of bench.py, produced on the one-GPU box by a torchrun-style rank of a world of ONE (RCCL communicator of
one rank inside libgndt) with --force-multi-extras: owner-partitioned steps, the gather to rank 0, the three --mode global steps
and the single-GPU anchor of the same cloud all run, and the anchor must be the very map the ranks built."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multi_gpu_line_carries_anchor_gather_and_modes():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "owner", "--workload", "S3", "--points", "3000000",
                        "--steps", "4", "--warmup", "2", "--force-multi-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["config"]["multi_gpu_mode"] == "owner" and d["retries_in_timed_region"] == 0 and "valid" not in d
    a = d["single_gpu_anchor"]
    assert a["same_map_as_the_ranks"] and a["points_total"] == d["config"]["points_total"] == 3000000 and a["nodes"] == d["config"]["nodes"]
    assert a["retries_in_timed_region"] == 0 and d["speedup_vs_single_gpu"] > 0
    assert d["gather_ms"]["value"] > 0 and d["gather_ms"]["assembled_map_matches_totals"] and d["gather_ms"]["rows"] == d["config"]["nodes"]
    assert d["modes"]["global"]["nodes"] == d["config"]["nodes"] and d["modes"]["global"]["ms_per_step"] > 0
    assert d["comm_selftest"]["ok_mask"] == 15 and d["comm_selftest"]["ranks"] == 1      # (RCCL itself, one rank: every primitive verified)


@pytest.mark.parametrize("fail,ends_on", [("owner", "global"), ("owner,global", "shards_without_exchange")])
def test_multi_gpu_line_steps_down_when_the_exchange_reports_an_error(fail, ends_on):
    """An error of the library's exchange on the first, untimed build (injected here) is agreed on by all ranks, named in the line, and
    the measurement goes on in the next mode instead of ending without a line."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", GNDT_BENCH_FAIL_MODES=fail)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "owner", "--workload", "S3", "--points", "2000000",
                        "--steps", "3", "--warmup", "2", "--force-multi-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["multi_gpu_mode"] == ends_on and [f["mode"] for f in d["fallback"]] == fail.split(",")
    assert all("injected failure" in f["error"] for f in d["fallback"])
    assert d["config"]["points_total"] == 2000000 and d["config"]["nodes"] > 0 and d["value"] > 0 and d["scaling"] == "strong"
    assert "gather_ms" not in d and "modes" not in d
    # a fallback that changes the computation is not a measurement of the configuration: the line says so
    assert (d.get("valid") is False and "shards_without_exchange" in d["invalid_because"]) == (ends_on == "shards_without_exchange")


def test_single_gpu_line_carries_every_config_and_the_host_path():
    """The N = 1 line (what the driver records): S2 as `value`, and — untimed by it — the other BASELINE.json configurations
    (`configs`: S1 campus / bridge_ground / depth frame, S2z, S3, S5, S4 eager and replayed, first builds), the host side of the seam
    stage by stage (`host_path`, eager and lazy), the host-buffer build (`host_build`) and the cost flood on three grids (`cost_flood`)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["retries_in_timed_region"] == 0 and d["config"]["points_total"] == 10_000_000
    c = d["configs"]
    for k in ("S1_campus_200k", "S1_bridge_ground_360k", "S1_depth_frame_215k", "S2z_10M_z01", "S3_terrain_8M", "S5_site_5M"):
        assert c[k]["ms_per_build"] > 0 and c[k]["retries"] == 0 and 0 < c[k]["path_frac"] < 1 and c[k]["first_build_ms"] > 0, (k, c[k])
    assert c["S2_first_build"]["first_build_ms"] > 0
    assert c["S4_stream_131k_frames"]["deferred_emit"]["back_to_back_ms_per_frame"] < c["S4_stream_131k_frames"]["eager"]["back_to_back_ms_per_frame"]
    assert c["S4_stream_131k_frames"]["deferred_emit"]["nodes_at_the_end"] == c["S4_stream_131k_frames"]["eager"]["nodes_at_the_end"]
    for k in ("eager", "hip_graph_replay"):
        s4 = c["S4_stream_131k_frames"][k]
        assert 0 < s4["p50_ms"] <= s4["p99_ms"] < 100.0 and s4["back_to_back_ms_per_frame"] > 0
    hp = d["host_path"]
    for k in ("S1_campus_200k", "S1_bridge_ground_360k", "drivable_site_400k"):
        assert "error" not in hp[k], hp[k]
        for mode in ("eager", "lazy"):
            assert hp[k][mode]["total_ms"] > 0 and hp[k][mode]["build_sync_ms"] > 0
        assert hp[k]["lazy"]["containers_ms"] < hp[k]["eager"]["containers_ms"]
        assert hp[k]["oracle_as_shipped_ms"]["division"] > 0
    site = hp["drivable_site_400k"]
    assert site["eager"]["route_found"] and site["lazy"]["route_found"] and site["eager"]["route_steps"] == site["lazy"]["route_steps"] > 2
    assert d["host_build"]["pageable_ms"] > 0 and d["host_build"]["pinned_ms"] > 0
    cf = d["cost_flood"]                      # the grid's immediate consumer (computeCost), each flood checked against the oracle's
    for k in ("drivable_site_400k", "bridge_ground_360k_own_parameters", "terrain_2M"):
        assert cf[k]["h_bit_exact"] and cf[k]["state_exact"] and cf[k]["layers"] > 10 and cf[k]["gpu_ms"] > 0, (k, cf[k])
    assert cf["bridge_ground_360k_own_parameters"]["ring_depth"] == 2 and cf["drivable_site_400k"]["ring_depth"] == 0
