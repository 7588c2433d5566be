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
import ctypes as C

import torch
import torch.distributed as dist

INT32_MAX = 2**31 - 1


class Communicator:
    """The RCCL communicator libgndt's own exchange runs on (gndt_comm_*, include/gndt.h).  The 128-byte unique id is made
    by rank 0 inside libgndt and handed to the other ranks through the process group that is already up (any backend):
    that is the only use of torch.distributed on this path — the data never passes through it."""

    def __init__(self, device, group=None):
        from . import _lib
        self._L = _lib.lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        if rank == 0:
            rc = self._L.gndt_comm_unique_id(buf)
            if rc:
                raise _lib.GndtError(rc, (self._L.gndt_comm_last_error() or b"").decode())
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=f"cuda:{device}" if on_gpu else "cpu")
        dist.broadcast(t, src=0, group=group)
        ident = bytes(t.cpu().tolist())
        h = C.c_void_p()
        rc = self._L.gndt_comm_create(ident, rank, world, int(device), C.byref(h))
        if rc:
            raise _lib.GndtError(rc, (self._L.gndt_comm_last_error() or b"").decode())
        self.handle, self.rank, self.world = h, rank, world

    @classmethod
    def single(cls, device=0):
        """A communicator of ONE rank, without torch.distributed (tests, single-GPU hosts that use the sharded entry points)."""
        from . import _lib
        self = cls.__new__(cls)
        self._L = _lib.lib()
        buf = C.create_string_buffer(128)
        rc = self._L.gndt_comm_unique_id(buf)
        if rc:
            raise _lib.GndtError(rc, (self._L.gndt_comm_last_error() or b"").decode())
        h = C.c_void_p()
        rc = self._L.gndt_comm_create(buf.raw, 0, 1, int(device), C.byref(h))
        if rc:
            raise _lib.GndtError(rc, (self._L.gndt_comm_last_error() or b"").decode())
        self.handle, self.rank, self.world = h, 0, 1
        return self

    def close(self):
        if self.handle:
            self._L.gndt_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_stats(key, sums, count, first_idx, group=None):
    """The same exchange spelled with torch.distributed collectives, for backends libgndt's RCCL path cannot use (the
    world_size-2 gloo tests on CPU): all-gather of the keys -> sorted unique union -> ONE packed sum all-reduce over
    [C, 10] fp64 (9 sums + count) and one min all-reduce (first-seen).
    key [n] int64 (packed node keys, unique per rank), sums [n, 9] float64, count [n] int32,
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
    packed = torch.zeros((c, 10), dtype=torch.float64, device=dev)                 # 9 sums + the count (exact in fp64)
    g_first = torch.full((c,), INT32_MAX, dtype=torch.int32, device=dev)
    packed[pos, :9] = sums
    packed[pos, 9] = count.to(torch.float64)
    g_first[pos] = first_idx
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(g_first, op=dist.ReduceOp.MIN, group=group)
    g_sums = packed[:, :9].contiguous()
    g_count = packed[:, 9].round().to(torch.int32)
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
