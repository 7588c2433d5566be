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

    @classmethod
    def threads(cls, world, device=0):
        """`world` communicators whose ranks are threads of THIS process (gndt_comm_create_threads): one per calling thread."""
        from . import _lib
        L = _lib.lib()
        arr = (C.c_void_p * int(world))()
        rc = L.gndt_comm_create_threads(int(world), int(device), arr)
        if rc:
            raise _lib.GndtError(rc, (L.gndt_comm_last_error() or b"").decode())
        out = []
        for r in range(int(world)):
            self = cls.__new__(cls)
            self._L, self.handle, self.rank, self.world = L, C.c_void_p(arr[r]), r, int(world)
            out.append(self)
        return out

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


# ---------------------------------------------------------------------------------------------------------------------
# Owner-partitioned build (include/gndt.h: gndt_build_owned_device) spelled with torch.distributed point-to-point calls:
# the transport for backends libgndt's own RCCL path cannot use (the world_size-2 gloo tests on CPU) and the model for
# hosts that bring their own (MPI, a ROS bridge): the four exported steps with the hand-over in between.
# ---------------------------------------------------------------------------------------------------------------------
def owner_of_columns(sx, sy, world):
    """Owner rank of every column (numpy int32 arrays): libgndt's own hash (host helper, no GPU)."""
    import numpy as np
    from . import _lib
    sx = np.ascontiguousarray(sx, np.int32)
    sy = np.ascontiguousarray(sy, np.int32)
    out = np.zeros(sx.shape[0], np.uint32)
    rc = _lib.lib().gndt_owner_of_columns(sx.ctypes.data, sy.ctypes.data, sx.shape[0], int(world), out.ctypes.data)
    if rc:
        raise _lib.GndtError(rc, "gndt_owner_of_columns")
    return out


def exchange_records(runs, group=None):
    """runs[r]: the [c_r, 4] records (any 4-byte dtype) this rank holds for rank r  ->  the records this rank owns: the runs
    every rank holds for it, in rank order.  Sizes first (all-gather), then one send / receive per peer."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = runs[0].device
    mine = torch.tensor([int(r.shape[0]) for r in runs], dtype=torch.int64, device=dev)
    table = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    incoming = [int(table[r][rank]) for r in range(world)]                 # what rank r holds for this rank
    bits = [r.view(torch.int32) for r in runs]                             # index words are bit patterns, never numbers
    got = [torch.empty((incoming[r], 4), dtype=torch.int32, device=dev) for r in range(world)]
    got[rank].copy_(bits[rank])
    ops = []
    for r in range(world):
        if r == rank:
            continue
        if bits[r].shape[0]:
            ops.append(dist.P2POp(dist.isend, bits[r].contiguous(), r, group))
        if incoming[r]:
            ops.append(dist.P2POp(dist.irecv, got[r], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return torch.cat(got, 0).contiguous().view(runs[0].dtype)


def gather_column_pairs(pairs, group=None):
    """Every rank's (first-seen index << 32 | node count) column pairs -> all of them on every rank (int64; the padding of
    the all-gather is -1, which gndt_owned_global_rows_device skips)."""
    world = dist.get_world_size(group)
    n = torch.tensor([pairs.shape[0]], dtype=torch.int64, device=pairs.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = max(1, max(int(x) for x in sizes))
    padded = torch.full((m,), -1, dtype=torch.int64, device=pairs.device)
    padded[:pairs.shape[0]] = pairs
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat(out)


def build_owned_map(m, demand, shard, first_idx_base, total_points, group=None, stream=None):
    """The owner-partitioned build with torch.distributed as the transport: this rank's contiguous range of the cloud in;
    afterwards `m` holds the columns this rank owns.  Returns (global row of every local row, nodes, columns of the whole map)."""
    world = dist.get_world_size(group)
    if 1 < world <= 16:                 # locality-aware ownership: everybody's samples -> the same block table on every rank
        msg = m.owner_sample(demand, shard, stream).clone()
        msgs = [torch.empty_like(msg) for _ in range(world)]
        dist.all_gather(msgs, msg, group=group)
        m.owner_map(torch.cat(msgs).contiguous(), world, stream)
    runs = m.owner_split(demand, shard, first_idx_base, total_points, world, stream)
    own = exchange_records(runs, group)
    m.build_records(demand, own, total_points, stream)
    allp = gather_column_pairs(m.owned_columns(stream).clone(), group)
    return m.owned_global_rows(allp, total_points, stream)


def gather_packed_rows(rows, group=None, root=-1):
    """Packed result rows ([n, 21] int32: gndt_owned_pack_rows_device, last word = the row's place in the map of the whole cloud)
    of every rank -> all of them on rank `root` (root < 0: on every rank); other ranks get None.  Padding records of the
    fixed-size all-gather carry -1 in their last word, which gndt_adopt_rows_device skips."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x) for x in sizes]
    if root < 0:
        m = max(1, max(sizes))
        padded = torch.full((m, rows.shape[1]), -1, dtype=torch.int32, device=rows.device)
        padded[:rows.shape[0]] = rows
        out = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(out, padded, group=group)
        return torch.cat(out)
    if rank == root:
        got = [torch.empty((sizes[r], rows.shape[1]), dtype=torch.int32, device=rows.device) for r in range(world)]
        got[rank].copy_(rows)
        ops = [dist.P2POp(dist.irecv, got[r], r, group) for r in range(world) if r != rank and sizes[r]]
    else:
        ops = [dist.P2POp(dist.isend, rows.contiguous(), root, group)] if rows.shape[0] else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return torch.cat(got) if rank == root else None


def gather_owned_map(m, totals, group=None, root=-1, stream=None):
    """build_owned_map's result assembled: `totals` = (nodes, columns, slopes) of the whole map.  On `root` (or every rank) `m`
    then holds the map of the whole cloud."""
    rows = gather_packed_rows(m.owned_pack_rows(stream).clone(), group, root)
    if rows is not None:
        m.adopt_rows(rows, totals[0], totals[1], totals[2], stream)
