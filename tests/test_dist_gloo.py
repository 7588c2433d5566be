"""CPU tier, world_size 2 over gloo: the sharded-cloud exchange (grid_ndt_amd/dist.py merge_stats)
must reproduce the single-process statistics, and the map finalised from them must match the oracle
of the whole cloud.  Per-rank statistics come from the kernels' own arithmetic header through the host
shim (tests/host_emulation.py); on the GPU box the same function runs on RCCL with libgndt's exports."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import host_emulation as he
from tests import parity, scenes

P = scenes.CAMPUS_PARAMS


def _worker(rank, world, port, cut, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grid_ndt_amd.dist import merge_stats
    cloud = scenes.campus_frame(50000)
    body, origin = cloud[1:], cloud[0]
    lo, hi = (0, cut) if rank == 0 else (cut, body.shape[0])
    uk, cnt, first, sums, _ = he.accumulate(body[lo:hi], origin, P["grid_len"], P["z_len"], first_base=lo)
    key = torch.from_numpy(uk.view(np.int64).copy())
    out = merge_stats(key, torch.from_numpy(sums), torch.from_numpy(cnt.astype(np.int32)),
                      torch.from_numpy(first.astype(np.int32)))
    q.put((rank, [t.numpy() for t in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_matches_single_process_and_oracle():
    world, port = 2, 29617
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 21111, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # identical on both ranks
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    key, sums, cnt, first = res[0]
    cloud = scenes.campus_frame(50000)
    uk, c1, f1, s1, cen = he.accumulate(cloud[1:], cloud[0], P["grid_len"], P["z_len"])
    order = np.argsort(uk.view(np.int64), kind="stable")          # merge_stats orders keys as signed int64
    assert np.array_equal(key, uk.view(np.int64)[order])
    assert np.array_equal(cnt, c1[order].astype(np.int32)) and np.array_equal(first, f1[order].astype(np.int32))
    assert np.allclose(sums, s1[order], rtol=1e-13, atol=1e-13)
    # finalise the merged statistics and compare with the oracle of the whole cloud
    out = he.finalize(key.view(np.uint64), cnt.astype(np.uint32), first.astype(np.uint32), sums, cen[order],
                      P["slope_interval"], P["demand"])
    parity.assert_parity(out, parity.ref_from_cloud(cloud, P))
