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
from grid_ndt_amd import scenes
from tests import parity

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


# ---------------------------------------------------------------------------------------------------------------------
# owner-partitioned build (include/gndt.h: gndt_build_owned_device): the protocol on two gloo ranks.  The device steps are
# played by the host emulation (same arithmetic header), the transport is grid_ndt_amd/dist.py's (exchange_records,
# gather_column_pairs) and the owner of a column is libgndt's own hash (gndt_owner_of_columns).
# ---------------------------------------------------------------------------------------------------------------------
def _column_pairs(local):
    """(first-seen index << 32 | node count) per column of a map in reference order + the first row of every column."""
    col = (local["sx"].astype(np.int64) << 32) ^ (local["sy"].astype(np.int64) & 0xFFFFFFFF)
    head = np.flatnonzero(np.concatenate([[True], col[1:] != col[:-1]])) if col.size else np.zeros(0, np.int64)
    ncol = np.diff(np.concatenate([head, [col.size]]))
    return (local["first_idx"][head].astype(np.int64) << 32) | ncol.astype(np.int64), head, ncol


def _global_rows(all_pairs, head, ncol, local):
    """What k_pairs_note + the word-weight prefix + k_global_rows compute: columns ordered by first-seen index."""
    p = all_pairs[all_pairs != -1]
    cf, nc = p >> 32, p & 0xFFFFFFFF
    order = np.argsort(cf, kind="stable")
    base = np.concatenate([[0], np.cumsum(nc[order])])
    pos = {int(c): int(b) for c, b in zip(cf[order], base[:-1])}
    rows = np.zeros(local["sx"].shape[0], np.int64)
    for h, k in zip(head, ncol):
        rows[h:h + k] = pos[int(local["first_idx"][h])] + np.arange(k)
    return rows, int(base[-1]), int(p.size)


_ROW_FIELDS = (("sx", 1), ("sy", 1), ("sz", 1), ("count", 1), ("first_idx", 1), ("mean", 3), ("cov", 6), ("rough", 1), ("normal", 3), ("flags", 1))


def _pack_rows(local, rows):
    """What k_rows_pack writes: the 19 words of a result row, the column index (node count on a column's first row) and the global row."""
    n = local["sx"].shape[0]
    out = np.zeros((n, 21), np.int32)
    w = 0
    for name, k in _ROW_FIELDS:
        out[:, w:w + k] = np.ascontiguousarray(local[name]).reshape(n, k).view(np.int32)
        w += k
    _, head, ncol = _column_pairs(local)
    out[head, 19] = ncol.astype(np.int32)
    out[:, 20] = rows.astype(np.int32)
    return out


def _adopt_rows(packed, total):
    """What k_rows_adopt does with everybody's records."""
    keep = packed[packed[:, 20] != -1]
    assert keep.shape[0] == total and np.array_equal(np.sort(keep[:, 20]), np.arange(total))
    order = np.argsort(keep[:, 20], kind="stable")
    keep = keep[order]
    out, w = {}, 0
    for name, k in _ROW_FIELDS:
        col = np.ascontiguousarray(keep[:, w:w + k])
        dt = np.float32 if name in ("mean", "cov", "rough", "normal") else (np.uint32 if name in ("count", "first_idx", "flags") else np.int32)
        out[name] = col.view(dt).reshape((total, k) if k > 1 else (total,))
        w += k
    out["row_ncol"] = keep[:, 19].copy()
    return out


def _owner_worker(rank, world, port, cut, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grid_ndt_amd.dist import exchange_records, gather_column_pairs, owner_of_columns
    cloud = scenes.campus_frame(50000)
    body, origin = cloud[1:], cloud[0]
    total = body.shape[0]
    lo, hi = (0, cut) if rank == 0 else (cut, total)
    shard = body[lo:hi]
    # step 1 (gndt_owner_split_device): records {x, y, z, global index} grouped by the owner of their column
    sx, sy = he.point_columns(shard, origin, P["grid_len"], P["z_len"])
    owner = owner_of_columns(sx, sy, world)
    order = np.argsort(owner, kind="stable")
    recs = np.zeros((shard.shape[0], 4), np.float32)
    recs[:, :3] = shard[order, :3]
    recs[:, 3] = (np.arange(lo, hi, dtype=np.uint32)[order]).view(np.float32)
    off = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=world))])
    own = exchange_records([torch.from_numpy(recs[off[r]:off[r + 1]]) for r in range(world)]).numpy()
    idx = np.ascontiguousarray(own[:, 3]).view(np.uint32).astype(np.int64)
    # every point of a column this rank owns is here, and nothing else
    sx2, sy2 = he.point_columns(own, origin, P["grid_len"], P["z_len"])
    assert np.all(owner_of_columns(sx2, sy2, world) == rank)
    # step 2 (gndt_build_records_device): the ordinary build on the records, index words taken as they are
    uk, cnt, first, sums, cen = he.accumulate(own, origin, P["grid_len"], P["z_len"], idx=idx)
    local = he.finalize(uk, cnt, first, sums, cen, P["slope_interval"], P["demand"])
    # steps 3 + 4 (gndt_owned_columns_device, gndt_owned_global_rows_device)
    pairs, head, ncol = _column_pairs(local)
    allp = gather_column_pairs(torch.from_numpy(pairs)).numpy()
    rows, n_glob, k_glob = _global_rows(allp, head, ncol, local)
    # step 5 (gndt_owned_pack_rows_device -> transport -> gndt_adopt_rows_device): the rows travel as packed records and are
    # scattered by their place in the map of the whole cloud; to every rank, and to rank 1 alone
    from grid_ndt_amd.dist import gather_packed_rows
    packed = _pack_rows(local, rows)
    everywhere = _adopt_rows(gather_packed_rows(torch.from_numpy(packed), root=-1).numpy(), n_glob)
    at_one = gather_packed_rows(torch.from_numpy(packed), root=1)
    assert (at_one is None) == (rank != 1)
    if rank == 1:
        again = _adopt_rows(at_one.numpy(), n_glob)
        assert all(np.array_equal(everywhere[k], again[k]) for k in everywhere)
    q.put((rank, local, rows, n_glob, k_glob, int(own.shape[0]), everywhere))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_owner_partitioned_build_assembles_the_oracle_map():
    world, port = 2, 29619
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, 17777, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cloud = scenes.campus_frame(50000)
    ref = parity.ref_from_cloud(cloud, P)
    N, K = res[0][2], res[0][3]
    assert (N, K) == (res[1][2], res[1][3]) == (ref["num_nodes"], ref["num_columns"])
    assert res[0][4] + res[1][4] == cloud.shape[0] - 1 and min(res[0][4], res[1][4]) > 0.3 * cloud.shape[0]   # a real split
    glob = {k: np.zeros((N,) + v.shape[1:], v.dtype) for k, v in res[0][0].items() if isinstance(v, np.ndarray)}
    seen = np.zeros(N, bool)
    for r in range(world):
        local, rows = res[r][0], res[r][1]
        assert not seen[rows].any()
        seen[rows] = True
        for k in glob:
            glob[k][rows] = local[k]
    assert seen.all()
    glob.update(num_nodes=N, num_columns=K, num_slopes=sum(res[r][0]["num_slopes"] for r in range(world)))
    parity.assert_parity(glob, ref)
    # the map every rank assembled from the packed rows is that same map, and its column index marks every column's first row
    for r in range(world):
        got = res[r][5]
        for k in ("sx", "sy", "sz", "count", "first_idx", "flags", "mean", "cov", "rough", "normal"):
            assert np.array_equal(got[k], glob[k].reshape(got[k].shape)), (r, k)
        assert int(np.count_nonzero(got["row_ncol"])) == K and int(got["row_ncol"].sum()) == N
