"""Host-side mirror of the reference's map interface for the grid-build path, over libgndt's C ABI.

`TwoDmap` keeps the reference's method names and argument meaning (include/map2D.h:485-507,
592-668, 950-976); the loop `for i in 1..n-1: uniformDivision(points[i])` + `create2DMap(demand)`
(src/receiver.cpp:150-160) becomes one call, `create2DMap(demand, cloud)`.  PyTorch appears only as
the owner of device memory and streams.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Cells, CostStats, ExchangeTimes, GndtError, OwnedInfo, Params, Pcd, PointLayout, Robot, Stats

DEMANDS = {"slope": 0, "true": 1}
FLAG_HAS_STATS, FLAG_SLOPE, FLAG_DOWN = 1, 2, 4


_HIP_STREAM_LEGACY = 1      # hipStreamLegacy ((hipStream_t)1): the null stream by its explicit name


def _stream_ptr(stream):
    """libgndt takes NULL as "the handle's own (non-blocking) stream".  torch's default stream IS the null stream
    (cuda_stream == 0), and work the caller enqueues there — filling the input buffer, say — is not ordered with a
    non-blocking stream: the null stream is therefore passed by its explicit name, hipStreamLegacy."""
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream().cuda_stream or _HIP_STREAM_LEGACY)
        except ImportError:
            pass
        return C.c_void_p(0)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream or _HIP_STREAM_LEGACY)
    return C.c_void_p(int(stream))


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can view libgndt's device buffers zero-copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class TwoDmap:
    """daysun::TwoDmap for the build path (include/map2D.h:190-194, 485-507)."""

    def __init__(self, res, zres, device=0, max_nodes_hint=0, max_points_hint=0, strategy=0, min_points=3):
        self._L = _lib.lib()
        self.gridLen, self.zLen = float(res), float(zres)   # TwoDmap(res, zres), map2D.h:486
        self.slope_interval = 0.0
        self.device, self.strategy, self.min_points = int(device), int(strategy), int(min_points)
        self.max_nodes_hint, self.max_points_hint = int(max_nodes_hint), int(max_points_hint)
        self.cloudFirst = None
        self._h = None
        self._demand = None

    # ---- setters / getters with the reference's names (map2D.h:487-502) ----
    def getGridLen(self):
        return self.gridLen

    def getZLen(self):
        return self.zLen

    def setCloudFirst(self, p):
        self.cloudFirst = tuple(float(v) for v in p[:3])
        if self._h is not None:
            self._destroy()

    def setLen(self, length):
        self.gridLen = float(length)
        self._destroy()

    def setZLen(self, length):
        self.zLen = float(length)
        self._destroy()

    def setInterval(self, interval):
        self.slope_interval = float(interval)
        self._destroy()

    def getInterval(self):
        return self.slope_interval

    # ---- transMortonXYZ (map2D.h:950-976): host codec, same strings as the reference's map keys ----
    def transMortonXYZ(self, position):
        o = (C.c_float * 3)(*self.cloudFirst)
        p = (C.c_float * 3)(*[float(v) for v in position[:3]])
        q = C.create_string_buffer(2)
        key = C.create_string_buffer(16)
        nx, ny, sz = C.c_int32(), C.c_int32(), C.c_int32()
        rc = self._L.gndt_trans_morton_xyz(o, self.gridLen, self.zLen, p, q, C.byref(nx), C.byref(ny), C.byref(sz), key)
        if rc:
            raise GndtError(rc, "position outside the key range")
        return key.value.decode(), sz.value

    # ---- handle management ----
    def _destroy(self):
        if self._h is not None:
            self._L.gndt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            msg = self._L.gndt_last_error(self._h)
            raise GndtError(rc, msg.decode() if msg else "")

    def _ensure(self, demand, need_origin=True):
        d = DEMANDS[demand] if isinstance(demand, str) else int(demand)
        if self._h is not None and self._demand == d:
            return
        self._destroy()
        if self.cloudFirst is None and need_origin:
            raise GndtError(1, "setCloudFirst must be called before building (receiver.cpp:145)")
        P = Params(self.gridLen, self.zLen, self.slope_interval, d, self.min_points, self.device, self.strategy,
                   self.max_points_hint, self.max_nodes_hint)
        h = C.c_void_p()
        rc = self._L.gndt_create(C.byref(P), C.byref(h))
        if rc:
            msg = self._L.gndt_last_error(None)
            raise GndtError(rc, msg.decode() if msg else "")
        self._h, self._demand = h, d
        if self.cloudFirst is not None:
            o = (C.c_float * 3)(*self.cloudFirst)
            self._check(self._L.gndt_set_origin(self._h, o))

    @staticmethod
    def _as_input(points):
        """-> (pointer, n, stride_bytes, is_device, keepalive)"""
        try:
            import torch
            if isinstance(points, torch.Tensor):
                assert points.dtype == torch.float32 and points.dim() == 2 and points.shape[1] in (3, 4)
                t = points if points.is_contiguous() else points.contiguous()
                return t.data_ptr(), t.shape[0], 4 * t.shape[1], t.is_cuda, t
        except ImportError:
            pass
        a = np.ascontiguousarray(points, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] in (3, 4)
        return a.ctypes.data, a.shape[0], 4 * a.shape[1], False, a

    # ---- the build: receiver.cpp:150-154 + map2D.h:592 ----
    def create2DMap(self, demand, points, stream=None):
        """`points` = the cloud WITHOUT point 0 ([N,3] or [N,4] float32; torch CUDA tensor, torch CPU
        tensor or numpy).  Device input is enqueued on `stream` (default: torch's current stream)."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if on_dev:
            self._check(self._L.gndt_build_device(self._h, C.c_void_p(ptr), n, stride, _stream_ptr(stream)))
        else:
            self._check(self._L.gndt_build(self._h, C.c_void_p(ptr), n, stride))
        self._keep = keep
        return True

    def change2DMap(self, demand, points, stream=None):
        """Incremental add (intent of map2D.h:672-822 as defined in SURVEY Appendix A.7)."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if on_dev:
            self._check(self._L.gndt_update_device(self._h, C.c_void_p(ptr), n, stride, _stream_ptr(stream)))
        else:
            self._check(self._L.gndt_update(self._h, C.c_void_p(ptr), n, stride))
        self._keep = keep
        return True

    def del2DMap(self, demand, points, stream=None):
        """Incremental delete (intent of map2D.h:826-915 as defined in include/gndt.h): `points` were added before and leave
        their nodes; empty nodes are deleted, surviving nodes keep their place in the order."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if on_dev:
            self._check(self._L.gndt_remove_device(self._h, C.c_void_p(ptr), n, stride, _stream_ptr(stream)))
        else:
            self._check(self._L.gndt_remove(self._h, C.c_void_p(ptr), n, stride))
        self._keep = keep
        return True

    # ---- input side: raw records (PointCloud2 / .pcd payload) ----
    def pack_points(self, raw, point_step, offsets=(0, 4, 8), demand="slope", stream=None):
        """Raw point records on the device (torch uint8/float32 tensor of n * point_step bytes) -> packed xyz torch
        tensor [n_valid, 3]; rows with a NaN/Inf coordinate dropped, order kept (receiver.cpp:140-143, publisher.cpp:24-26)."""
        import torch
        self._ensure(demand, need_origin=False)
        nbytes = raw.numel() * raw.element_size()
        n = nbytes // int(point_step)
        out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=raw.device)
        lay = PointLayout(int(point_step), int(offsets[0]), int(offsets[1]), int(offsets[2]))
        nv = C.c_uint64()
        self._check(self._L.gndt_pack_points_device(self._h, C.c_void_p(raw.data_ptr()), n, C.byref(lay), C.c_void_p(out.data_ptr()),
                                                    C.byref(nv), _stream_ptr(stream)))
        return out[:int(nv.value)]

    def build_cloud(self, demand, raw, point_step, offsets=(0, 4, 8)):
        """chatterCallback in one call (receiver.cpp:137-160): raw host records (numpy, n * point_step bytes) -> NaN
        strip on the device, origin := first valid point, build of the rest."""
        self._ensure(demand, need_origin=False)
        raw = np.ascontiguousarray(raw)
        n = raw.nbytes // int(point_step)
        lay = PointLayout(int(point_step), int(offsets[0]), int(offsets[1]), int(offsets[2]))
        self._check(self._L.gndt_build_cloud(self._h, C.c_void_p(raw.ctypes.data), n, C.byref(lay)))
        return True

    # ---- split form for sharded clouds ----
    def reset(self, demand="slope", stream=None):
        self._ensure(demand)
        self._check(self._L.gndt_reset(self._h, _stream_ptr(stream)))

    def accumulate(self, demand, points, first_idx_base=0, stream=None):
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "accumulate takes device memory")
        self._check(self._L.gndt_accumulate_device(self._h, C.c_void_p(ptr), n, stride, int(first_idx_base), _stream_ptr(stream)))
        self._keep = keep

    def finalize(self, stream=None):
        self._check(self._L.gndt_finalize_device(self._h, _stream_ptr(stream)))

    def stats_export(self, stream=None):
        """Device-resident sufficient statistics as torch tensors (views of libgndt memory)."""
        import torch
        st = Stats()
        self._check(self._L.gndt_stats_export_device(self._h, C.byref(st), _stream_ptr(stream)))
        n = int(st.num_nodes)
        dev = f"cuda:{self.device}"
        if n == 0:
            return {"key": torch.zeros(0, dtype=torch.int64, device=dev), "sums": torch.zeros(0, 9, dtype=torch.float64, device=dev),
                    "count": torch.zeros(0, dtype=torch.int32, device=dev), "first_idx": torch.zeros(0, dtype=torch.int32, device=dev)}
        mk = lambda p, shape, ts: torch.as_tensor(_DevArray(p, shape, ts, self), device=dev)
        return {"key": mk(st.key, (n,), "<i8"), "sums": mk(st.sums, (n, 9), "<f8"),
                "count": mk(st.count, (n,), "<i4"), "first_idx": mk(st.first_idx, (n,), "<i4")}

    def _stats_tensors(self, st):
        import torch
        n = int(st.num_nodes)
        dev = f"cuda:{self.device}"
        if n == 0:
            return {"key": torch.zeros(0, dtype=torch.int64, device=dev), "sums": torch.zeros(0, 9, dtype=torch.float64, device=dev),
                    "count": torch.zeros(0, dtype=torch.int32, device=dev), "first_idx": torch.zeros(0, dtype=torch.int32, device=dev)}
        mk = lambda p, shape, ts: torch.as_tensor(_DevArray(p, shape, ts, self), device=dev)
        return {"key": mk(st.key, (n,), "<i8"), "sums": mk(st.sums, (n, 9), "<f8"),
                "count": mk(st.count, (n,), "<i4"), "first_idx": mk(st.first_idx, (n,), "<i4")}

    def shard_stats(self, demand, points, first_idx_base=0, stream=None):
        """This rank's shard of a global cloud -> the statistics of its occupied nodes (torch views of libgndt
        memory, valid until the next call): the counting-partition pipeline, no node table."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "shard_stats takes device memory")
        st = Stats()
        self._check(self._L.gndt_shard_stats_device(self._h, C.c_void_p(ptr), n, stride, int(first_idx_base), C.byref(st),
                                                    _stream_ptr(stream)))
        self._keep = keep
        return self._stats_tensors(st)

    def build_global(self, comm, demand, points, first_idx_base, total_points, stream=None, timed=False):
        """This rank's contiguous range of a sharded cloud -> the map of the WHOLE cloud on every rank: shard statistics,
        RCCL exchange (called from C++ inside libgndt) and finalisation in one call.  `comm`: grid_ndt_amd.dist.Communicator."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "build_global takes device memory")
        t = ExchangeTimes()
        self._check(self._L.gndt_build_global_device(self._h, comm.handle, C.c_void_p(ptr), n, stride, int(first_idx_base), int(total_points),
                                                     C.byref(t) if timed else None, _stream_ptr(stream)))
        self._keep = keep
        if timed:
            return {"shard_ms": t.shard_ms, "exchange_ms": t.exchange_ms, "finalize_ms": t.finalize_ms, "ranks": int(t.ranks),
                    "local_nodes": int(t.local_nodes), "global_nodes": int(t.global_nodes), "bytes_reduced": int(t.bytes_reduced)}
        return None

    # ---- owner-partitioned build of a sharded cloud (include/gndt.h): the points travel, the map stays sharded by owner ----
    def _dev_view(self, ptr, nbytes, dtype, shape):
        """torch view of device memory libgndt owns (valid until the next call on the handle)."""
        import torch
        dev = f"cuda:{self.device}"
        if nbytes == 0 or not ptr:
            return torch.empty(shape, dtype=dtype, device=dev)
        ts = {torch.float32: "<f4", torch.int64: "<i8", torch.int32: "<i4"}[dtype]
        return torch.as_tensor(_DevArray(ptr, shape, ts, self), device=dev)

    def owner_sample(self, demand, points, stream=None):
        """Evenly spaced sample of this rank's shard as the fixed-size message every rank publishes (int32 device view)."""
        import torch
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "owner_sample takes device memory")
        p, w = C.c_void_p(), C.c_uint64()
        self._check(self._L.gndt_owner_sample_device(self._h, C.c_void_p(ptr), n, stride, C.byref(p), C.byref(w), _stream_ptr(stream)))
        return self._dev_view(p.value, int(w.value) * 4, torch.int32, (int(w.value),))

    def owner_map(self, all_msgs, world, stream=None):
        """The sample messages of all ranks (int32 device tensor, rank order) -> the block ownership later owner_split calls use."""
        self._check(self._L.gndt_owner_map_device(self._h, C.c_void_p(all_msgs.data_ptr()), int(world), _stream_ptr(stream)))

    def owner_split(self, demand, points, first_idx_base, total_points, world, stream=None):
        """This rank's contiguous range of the cloud -> its points as 16-B records grouped by owner rank.
        Returns the list of runs, one [count, 4] float32 device view per owner rank."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "owner_split takes device memory")
        recs = C.c_void_p()
        counts = (C.c_uint64 * int(world))()
        offsets = (C.c_uint64 * int(world))()
        self._check(self._L.gndt_owner_split_device(self._h, C.c_void_p(ptr), n, stride, int(first_idx_base), int(total_points), int(world),
                                                    C.byref(recs), counts, offsets, _stream_ptr(stream)))
        self._keep = keep
        import torch
        base = recs.value or 0
        return [self._dev_view(base + 16 * int(o), int(c) * 16, torch.float32, (int(c), 4)) for c, o in zip(counts, offsets)]

    def build_records(self, demand, records, total_points, stream=None):
        """The records this rank owns ([m, 4] float32 on the device: x, y, z, index word) -> its part of the map."""
        self._ensure(demand)
        if records.dim() != 2 or records.shape[1] != 4 or not records.is_contiguous() or not records.is_cuda:
            raise GndtError(1, "records must be a contiguous [m, 4] float32 device tensor")
        self._check(self._L.gndt_build_records_device(self._h, C.c_void_p(records.data_ptr()), int(records.shape[0]), int(total_points),
                                                      _stream_ptr(stream)))
        self._keep = records

    def build_records2(self, demand, first, room_and_second, total_points, stream=None):
        """build_records from two segments: `first` [a, 4] stays where it is; `room_and_second` [a + b, 4] holds the second
        segment in its last b rows (the first a rows are room the small-build path fills)."""
        self._ensure(demand)
        a = int(first.shape[0])
        b = int(room_and_second.shape[0]) - a
        assert b >= 0 and first.is_contiguous() and room_and_second.is_contiguous()
        self._check(self._L.gndt_build_records2_device(self._h, C.c_void_p(first.data_ptr()), a, C.c_void_p(room_and_second.data_ptr() + 16 * a), b,
                                                       int(total_points), _stream_ptr(stream)))
        self._keep = (first, room_and_second)

    def owned_columns(self, stream=None):
        """(first-seen index << 32 | node count) of every column of the local map: int64 device view."""
        import torch
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.gndt_owned_columns_device(self._h, C.byref(p), C.byref(n), _stream_ptr(stream)))
        return self._dev_view(p.value or 0, int(n.value) * 8, torch.int64, (int(n.value),))

    def owned_global_rows(self, all_pairs, total_points, stream=None):
        """Everybody's column pairs (int64 device tensor) -> (global row of every local row: int32 device view, nodes and
        columns of the whole map)."""
        import torch
        p, gn, gc = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._check(self._L.gndt_owned_global_rows_device(self._h, C.c_void_p(all_pairs.data_ptr()), int(all_pairs.shape[0]), int(total_points),
                                                          C.byref(p), C.byref(gn), C.byref(gc), _stream_ptr(stream)))
        n = self.sync()[0]
        return self._dev_view(p.value or 0, n * 4, torch.int32, (n,)), int(gn.value), int(gc.value)

    def second_pass_buckets(self):
        """Buckets of the last resolved PARTITION build that went through the bucket kernel's second pass (1024-slot tables)."""
        r = C.c_uint64()
        self._check(self._L.gndt_debug_second_pass_buckets(self._h, C.byref(r)))
        return int(r.value)

    def debug_fail_next_alloc(self, site, demand="slope"):
        """Tests: the next allocation at `site` of the sharded builds fails once on this handle (include/gndt.h)."""
        self._ensure(demand)             # (the handle is created lazily, with the demand)
        self._check(self._L.gndt_debug_fail_next_alloc(self._h, int(site)))

    def build_owned(self, comm, demand, points, first_idx_base, total_points, stream=None):
        """Owner-partitioned build over RCCL (called from C++ inside libgndt): this rank's range of the cloud in, the columns
        this rank owns out (export()), plus the global row of each of its rows.  Returns (global_row view, info dict)."""
        import torch
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "build_owned takes device memory")
        info, p = OwnedInfo(), C.c_void_p()
        self._check(self._L.gndt_build_owned_device(self._h, comm.handle, C.c_void_p(ptr), n, stride, int(first_idx_base), int(total_points),
                                                    C.byref(p), C.byref(info), _stream_ptr(stream)))
        self._keep = keep
        d = {k: (float(getattr(info, k)) if k.endswith("_ms") else int(getattr(info, k))) for k, _ in OwnedInfo._fields_}
        return self._dev_view(p.value or 0, d["local_nodes"] * 4, torch.int32, (d["local_nodes"],)), d

    def gather_owned(self, comm, root=-1, stream=None):
        """After build_owned: the rows of all ranks gathered on rank `root` (root < 0: on every rank) and scattered by their
        global row — this map then IS the map of the whole cloud (export(), computeCost(), ... as after a single-GPU build).
        Every rank of the communicator calls it; ranks other than `root` keep the columns they own."""
        self._check(self._L.gndt_gather_owned_map_device(self._h, comm.handle, int(root), _stream_ptr(stream)))

    def comm_selftest(self, comm, stream=None, demand="slope"):
        """gndt_comm_selftest: one verified round of every collective the sharded builds use (all ranks call it together)."""
        from ._lib import CommSelftest
        self._ensure(demand, need_origin=False)
        rep = CommSelftest()
        self._check(self._L.gndt_comm_selftest(self._h, comm.handle, C.byref(rep), _stream_ptr(stream)))
        return {"ok_mask": int(rep.ok_mask), "ranks": int(rep.ranks), "all_gather_ms": round(float(rep.all_gather_ms), 3),
                "exchange_ms": round(float(rep.exchange_ms), 3), "reduce_scatter_ms": round(float(rep.reduce_scatter_ms), 3),
                "all_reduce_ms": round(float(rep.all_reduce_ms), 3)}

    def owned_pack_rows(self, stream=None):
        """This rank's rows as packed records [n, 21] int32 (device view, valid until the next call): for hosts with their own transport."""
        import torch
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.gndt_owned_pack_rows_device(self._h, C.byref(p), C.byref(n), _stream_ptr(stream)))
        return self._dev_view(p.value or 0, n.value * 21 * 4, torch.int32, (n.value, 21))

    def adopt_rows(self, rows, total_nodes, total_columns, total_slopes, stream=None):
        """Packed records of any number of ranks ([m, 21] int32 on the device; padding records skipped) -> this map's rows."""
        rows = rows.contiguous()
        self._check(self._L.gndt_adopt_rows_device(self._h, C.c_void_p(rows.data_ptr()), int(rows.shape[0]), int(total_nodes), int(total_columns),
                                                   int(total_slopes), _stream_ptr(stream)))

    def finalize_stats(self, key, sums, count, first_idx, total_points, stream=None):
        """Merged statistics of the whole cloud (unique nodes sorted by key) -> the map."""
        st = Stats(int(key.shape[0]), key.data_ptr(), sums.data_ptr(), count.data_ptr(), first_idx.data_ptr())
        self._check(self._L.gndt_finalize_stats_device(self._h, C.byref(st), int(total_points), _stream_ptr(stream)))
        self._keep = (key, sums, count, first_idx)

    def stats_merge(self, key, sums, count, first_idx, stream=None):
        st = Stats(int(key.shape[0]), key.data_ptr(), sums.data_ptr(), count.data_ptr(), first_idx.data_ptr())
        self._check(self._L.gndt_stats_merge_device(self._h, C.byref(st), _stream_ptr(stream)))
        self._keep = (key, sums, count, first_idx)

    # ---- phase timing ----
    PHASES = {1: ("clear", "accumulate", "columns", "rows", "bitmap_scan", "rank", "column_scan", "dest", "emit"),
              5: ("clear", "accumulate", "columns", "rows", "bitmap_scan", "rank", "column_scan", "dest", "emit"),
              2: ("clear", "level1", "layout", "level2", "bucket_build", "bitmap_scan", "rank", "column_scan", "dest", "emit"),
              3: ("clear", "hist", "offsets", "scatter", "bucket_build", "bitmap_scan", "rank", "column_scan", "dest", "emit"),
              6: ("clear", "level1", "layout", "ranges", "bucket_build", "bitmap_scan", "rank", "column_scan", "dest", "emit"),
              7: ("clear", "level1", "layout", "level2", "bucket_build", "bitmap_scan", "rank", "column_scan", "dest", "emit")}
    STRATEGY_NAMES = {1: "atomic", 2: "partition", 3: "partition_exact", 5: "tile", 6: "partition_one_level", 7: "partition_blocked"}
    # phase -> the kernel that fills it, and what each phase's kernel moves algorithmically (bench.py's roofline line):
    # kernels that stream the cloud 12 B/point, the bucket kernel 12 B/point + 76 B/node, node kernels 76 B/node
    KERNEL_OF_PHASE = {"accumulate": "k_accumulate", "hist": "k_part_hist", "scatter": "k_part_scatter",
                       "level1": "k_part2_level1", "level2": "k_part2_level2",
                       "bucket_build": "k_bucket_direct",
                       "columns": "k_tab_columns", "rows": "k_tab_rows", "emit": "k_emit_rows"}
    KERNEL_OF_PHASE_BLOCKED = dict(KERNEL_OF_PHASE, bucket_build="k_bucket_blocked")      # strategy 7 (gndt_blocked.hpp)
    POINT_PHASES = ("accumulate", "hist", "scatter", "level1", "level2")
    POINT_AND_NODE_PHASES = ("bucket_build",)

    def set_profiling(self, on=True, demand="slope"):
        self._ensure(demand)
        self._check(self._L.gndt_set_profiling(self._h, int(on)))

    def locality_sample(self, points, tiles=64, demand="slope", stream=None):
        """Points per partial that strategy TILE would flush, measured on `tiles` tiles of the device-resident cloud."""
        self._ensure(demand)
        ptr, n, stride, on_dev, keep = self._as_input(points)
        if not on_dev:
            raise GndtError(1, "locality_sample takes device memory")
        r = C.c_double()
        self._check(self._L.gndt_locality_sample(self._h, C.c_void_p(ptr), n, stride, int(tiles), C.byref(r), _stream_ptr(stream)))
        return r.value

    def last_strategy(self):
        """1 = ATOMIC, 2 = PARTITION (two-level), 3 = PARTITION_EXACT: what the last build actually ran."""
        return int(self._L.gndt_last_strategy(self._h))

    def phase_times_ms(self):
        arr = (C.c_double * 10)()
        self._check(self._L.gndt_get_phase_times(self._h, arr))
        names = self.PHASES[self.last_strategy()]
        return {k: arr[i] for i, k in enumerate(names)}

    def retry_count(self):
        r = C.c_uint64()
        self._check(self._L.gndt_debug_retry_count(self._h, C.byref(r)))
        return int(r.value)

    def enable_stamps(self, on=True):
        self._L.gndt_debug_enable_stamps(int(bool(on)))

    def reserve(self, max_points, max_nodes=0, demand="slope"):
        """gndt_reserve: every buffer a build of this size can need, allocated now (capture a build on a fresh handle afterwards)."""
        self._ensure(demand, need_origin=False)
        self._check(self._L.gndt_reserve(self._h, int(max_points), int(max_nodes)))

    def warmup(self, expected_points=0, demand="slope"):
        """gndt_warmup: the kernels' code loaded (a temporary handle runs every strategy family once on a synthetic cloud of
        `expected_points` points) and, with a size known, every buffer reserved — so that the FIRST build costs what the next one does."""
        self._ensure(demand, need_origin=False)
        self._check(self._L.gndt_warmup(self._h, int(expected_points)))

    def set_deferred_emit(self, on=True, demand="slope"):
        """gndt_set_deferred_emit: updates relabel the touched columns only; the dense rows are produced when the map is read."""
        self._ensure(demand)
        self._check(self._L.gndt_set_deferred_emit(self._h, int(bool(on))))

    DEBUG_VERBOSE, DEBUG_TILE_RATIO, DEBUG_COST_ONE_WORKGROUP = 1, 2, 3

    @staticmethod
    def set_debug_option(option, value):
        """gndt_debug_set_option: process-wide diagnostics and thresholds (the library reads no environment variable)."""
        rc = _lib.lib().gndt_debug_set_option(int(option), float(value))
        if rc:
            raise _lib.GndtError(rc, f"gndt_debug_set_option({option}, {value})")

    def set_fp_bits(self, bits):
        """Narrow the bucket kernel's index fingerprint (process-wide; tests: forces its exact second pass)."""
        self._L.gndt_debug_set_fp_bits(int(bits))

    def fp_clashes(self):
        r = C.c_uint64()
        self._check(self._L.gndt_debug_fp_clashes(self._h, C.byref(r)))
        return int(r.value)

    def debug_bucket_phases(self):
        """Mean shader cycles per bucket of k_bucket_build's phases (after enable_stamps())."""
        arr = (C.c_double * 10)()
        nb = C.c_uint32()
        self._check(self._L.gndt_debug_bucket_phases(self._h, arr, C.byref(nb)))
        names = ("clear", "accumulate", "columns", "rows", "-", "-", "acc:0", "acc:1", "acc:2", "acc:3")
        return dict(zip(names, list(arr))), nb.value

    # ---- cost map (TwoDmap::computeCost, map2D.h:1285-1397) ----
    def computeCost(self, goal, robot=None, stream=None):
        """Flood the finished grid from the slope under `goal` (xyz).  `robot`: dict with radius,
        reachable_height, max_rough, max_angle_deg (defaults: RobotSphere(0.25), robot.h:38-46).
        Returns the statistics dict; h / state per result row come from cost_export()."""
        g = (C.c_float * 3)(*[float(v) for v in goal])
        rb = None
        if robot is not None:
            d = dict(radius=0.25, reachable_height=0.15, max_rough=100.0, max_angle_deg=30.0)
            d.update(robot)
            rb = C.byref(Robot(d["radius"], d["reachable_height"], d["max_rough"], d["max_angle_deg"]))
        self._check(self._L.gndt_compute_cost(self._h, g, rb, _stream_ptr(stream)))
        st = CostStats()
        self._check(self._L.gndt_cost_export(self._h, None, None, C.byref(st)))
        return self._cost_stats(st)

    @staticmethod
    def _cost_stats(st):
        return {"rc": int(st.goal_status), "ring": int(st.ring), "levels": int(st.levels), "traversable": int(st.traversable),
                "closed": int(st.closed), "check_pushes": int(st.check_pushes), "ring_store": int(st.ring_store)}

    def cost_export(self):
        """Host copy of Slope::h (fp32, FLT_MAX = unreached) and the flood state per result row."""
        n, _, _ = self.sync()
        h = np.zeros(n, np.float32)
        state = np.zeros(n, np.uint32)
        st = CostStats()
        self._check(self._L.gndt_cost_export(self._h, h.ctypes.data, state.ctypes.data, C.byref(st)))
        out = self._cost_stats(st)
        out.update(h=h, state=state.astype(np.uint8))
        return out

    # ---- results ----
    def sync(self):
        n, k, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._L.gndt_sync(self._h, C.byref(n), C.byref(k), C.byref(s)))
        return n.value, k.value, s.value

    def export(self):
        """Host copy of the SoA result (numpy), nodes in the reference's order."""
        n, k, s = self.sync()
        arrs = {"sx": np.zeros(n, np.int32), "sy": np.zeros(n, np.int32), "sz": np.zeros(n, np.int32),
                "count": np.zeros(n, np.uint32), "first_idx": np.zeros(n, np.uint32),
                "mean": np.zeros((n, 3), np.float32), "cov": np.zeros((n, 6), np.float32),
                "rough": np.zeros(n, np.float32), "normal": np.zeros((n, 3), np.float32), "flags": np.zeros(n, np.uint32)}
        c = Cells()
        for name, a in arrs.items():
            setattr(c, name, a.ctypes.data)
        self._check(self._L.gndt_export(self._h, C.byref(c)))
        arrs.update(num_nodes=int(c.num_nodes), num_columns=int(c.num_columns), num_slopes=int(c.num_slopes))
        return arrs

    def export_device(self):
        """Zero-copy torch views of the device-resident SoA (valid until the next build)."""
        import torch
        c = Cells()
        self._check(self._L.gndt_export_device(self._h, C.byref(c)))
        n = int(c.num_nodes)
        dev = f"cuda:{self.device}"
        out = {"num_nodes": n, "num_columns": int(c.num_columns), "num_slopes": int(c.num_slopes)}
        if n == 0:
            return out
        mk = lambda p, shape, ts: torch.as_tensor(_DevArray(p, shape, ts, self), device=dev)
        for name, shape, ts in (("sx", (n,), "<i4"), ("sy", (n,), "<i4"), ("sz", (n,), "<i4"), ("count", (n,), "<i4"),
                                ("first_idx", (n,), "<i4"), ("mean", (n, 3), "<f4"), ("cov", (n, 6), "<f4"),
                                ("rough", (n,), "<f4"), ("normal", (n, 3), "<f4"), ("flags", (n,), "<i4")):
            out[name] = mk(getattr(c, name), shape, ts)
        return out


def count_morton(a, b):
    """countMorton (Stopwatch.h:116-147)."""
    buf = C.create_string_buffer(16)
    rc = _lib.lib().gndt_count_morton(int(a), int(b), buf)
    if rc:
        raise GndtError(rc, "count_morton")
    return buf.value.decode()


def morton_to_xy(m):
    """mortonToXY (Stopwatch.h:171-189)."""
    a, b = C.c_int32(), C.c_int32()
    rc = _lib.lib().gndt_morton_to_xy(int(m), C.byref(a), C.byref(b))
    if rc:
        raise GndtError(rc, "morton_to_xy")
    return a.value, b.value


def trans_morton_xyz(origin, grid_len, z_len, p):
    o = (C.c_float * 3)(*[float(v) for v in origin])
    q = (C.c_float * 3)(*[float(v) for v in p])
    quad = C.create_string_buffer(2)
    key = C.create_string_buffer(16)
    nx, ny, sz = C.c_int32(), C.c_int32(), C.c_int32()
    rc = _lib.lib().gndt_trans_morton_xyz(o, float(grid_len), float(z_len), q, quad, C.byref(nx), C.byref(ny), C.byref(sz), key)
    return rc, key.value.decode(), nx.value, ny.value, sz.value


def device_info(device=0):
    name = C.create_string_buffer(128)
    cu, mem = C.c_int32(), C.c_uint64()
    rc = _lib.lib().gndt_device_info(int(device), name, C.byref(cu), C.byref(mem))
    if rc:
        raise GndtError(rc, "no device")
    return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}


def read_pcd(path):
    """pcl::io::loadPCDFile's part of src/publisher.cpp:19: header + payload of a .pcd file (`DATA ascii` or `binary`).
    Returns (raw, point_step, (off_x, off_y, off_z)): raw = numpy uint8 array of num_points * point_step bytes."""
    L = _lib.lib()
    p = Pcd()
    err = C.create_string_buffer(256)
    rc = L.gndt_pcd_read(str(path).encode(), C.byref(p), err)
    if rc:
        raise GndtError(rc, err.value.decode())
    try:
        nbytes = int(p.num_points) * int(p.layout.point_step)
        raw = np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(p.data))[:nbytes].copy() if nbytes else np.zeros(0, np.uint8)
    finally:
        L.gndt_pcd_free(C.byref(p))
    return raw, int(p.layout.point_step), (int(p.layout.offset_x), int(p.layout.offset_y), int(p.layout.offset_z))


import contextlib as _contextlib


@_contextlib.contextmanager
def graph_capture(graph, stream=None):
    """`with torch.cuda.graph(graph, stream=stream)` with Python's cycle collector paused for the duration of the capture.
    A collection that starts in the middle of a capture can run the destructor of ANOTHER object — a handle of an earlier test
    (gndt_destroy: hipStreamSynchronize, hipFree), a torch tensor, an abandoned CUDAGraph — and in torch's (global) capture mode any
    such call invalidates the capture; torch then aborts the process inside capture_end (seen once in the GPU tier, in a test that
    had just had a capture refused on purpose).  Reference-counted destruction is not affected: do not drop handles inside the block."""
    import gc
    import torch
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        if stream is None:
            with torch.cuda.graph(graph):
                yield
        else:
            with torch.cuda.graph(graph, stream=stream):
                yield
    finally:
        if was_enabled:
            gc.enable()
