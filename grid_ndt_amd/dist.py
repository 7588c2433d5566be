"""One global map from a cloud sharded over several GPUs (BASELINE.json configs[2]): the exchange step.

Each rank accumulates its contiguous shard into per-node sufficient statistics (count, Sum v, Sum v v^T in
cell-local coordinates, first-seen index).  Those are additive over any partition of the points, and the
first-seen index combines with min, so ONE collective round builds the global map:

    all_gather(keys)  ->  every rank sorts/uniques the union = identical canonical node order
    all_reduce(sum)   on the [C, 9] fp64 sums and the counts scattered into that order
    all_reduce(min)   on the first-seen indices
    stats_merge + finalize on every rank (each rank ends with the full map)

torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests); the reduction operands are only the OCCUPIED nodes, never the points.
"""
import torch
import torch.distributed as dist

INT32_MAX = 2**31 - 1


def merge_stats(key, sums, count, first_idx, group=None):
    """key [n] int64 (packed node keys, unique per rank), sums [n, 9] float64, count [n] int32,
    first_idx [n] int32 (global point indices, < 2^31).  Returns the globally reduced
    (key, sums, count, first_idx) in canonical (sorted-key) order, identical on every rank."""
    world = dist.get_world_size(group)
    dev = key.device
    n_local = torch.tensor([key.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(max(sizes), 1)
    padded = torch.full((n_max,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    padded[:key.shape[0]] = key
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    union = torch.unique(torch.cat([g[:s] for g, s in zip(gathered, sizes)]))     # sorted: the canonical order
    c = union.shape[0]
    pos = torch.searchsorted(union, key)
    g_sums = torch.zeros((c, 9), dtype=torch.float64, device=dev)
    g_count = torch.zeros((c,), dtype=torch.int32, device=dev)
    g_first = torch.full((c,), INT32_MAX, dtype=torch.int32, device=dev)
    g_sums[pos] = sums
    g_count[pos] = count
    g_first[pos] = first_idx
    dist.all_reduce(g_sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(g_count, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(g_first, op=dist.ReduceOp.MIN, group=group)
    return union, g_sums, g_count, g_first


def build_global_map(m, demand, shard, first_idx_base, stream=None, group=None, total_points=None, timings=None):
    """`m`: a TwoDmap whose origin is the GLOBAL cloud's point 0 on every rank; `shard`: this rank's
    device-resident points; `first_idx_base`: global index of shard[0].  On return `m` holds the map of
    the whole cloud (same on every rank).  Per rank: shard -> statistics through the counting-partition pipeline
    (no node table), one exchange, then the merged statistics (already sorted by key: the canonical order) ->
    labels, order and rows in one pass."""
    import time
    t0 = time.perf_counter()
    st = m.shard_stats(demand, shard, first_idx_base, stream)
    if stream is not None and hasattr(stream, "synchronize"):
        stream.synchronize()
    else:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    key, sums, count, first = merge_stats(st["key"].clone(), st["sums"].clone(), st["count"].clone(),
                                          st["first_idx"].clone(), group)
    if total_points is None:
        t = torch.tensor([int(shard.shape[0])], dtype=torch.int64, device=key.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        total_points = int(t.item())
    if timings is not None and key.is_cuda:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    m.finalize_stats(key.contiguous(), sums.contiguous(), count.contiguous(), first.contiguous(), total_points, stream)
    if timings is not None:
        if key.is_cuda:
            torch.cuda.synchronize()
        t3 = time.perf_counter()
        for k, v in (("shard_ms", t1 - t0), ("exchange_ms", t2 - t1), ("finalize_ms", t3 - t2)):
            timings[k] = timings.get(k, 0.0) + v * 1e3
    return key.shape[0]
