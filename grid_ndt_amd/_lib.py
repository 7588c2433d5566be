"""ctypes binding of libgndt.so (the C ABI declared in include/gndt.h).

The shared library is built in-tree by `build_native()` (hipcc, gfx950).  There is no Python or CPU
implementation of the path behind it: if the library is missing, loading raises.
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_CSRC, "libgndt.so")
SOURCES = ["gndt_api_core.hip", "gndt_api_table.hip", "gndt_api_build.hip", "gndt_api_dist.hip", "gndt_api_cost.hip",
           "gndt_api_io.hip", "gndt_codec.cpp", "gndt_io.cpp"]
HEADERS = ["gndt_handle.hpp", "gndt_kernels.hpp", "gndt_table.hpp", "gndt_cost.hpp", "gndt_pack.hpp", "gndt_partition.hpp", "gndt_bucket3.hpp", "gndt_blocked.hpp", "gndt_tile.hpp", "gndt_exchange.hpp", "gndt_math.hpp", os.path.join(_ROOT, "include", "gndt.h")]

GNDT_OK = 0
ERR_NAMES = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "KEY_RANGE", 5: "CAPACITY", 6: "NOMEM", 7: "PEER"}


class GndtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gndt error {code} ({ERR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("grid_len", C.c_float), ("z_len", C.c_float), ("slope_interval", C.c_float),
                ("demand", C.c_int32), ("min_points", C.c_int32), ("device_id", C.c_int32),
                ("strategy", C.c_int32), ("max_points_hint", C.c_uint64), ("max_nodes_hint", C.c_uint64)]


class Cells(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("num_columns", C.c_uint64), ("num_slopes", C.c_uint64),
                ("sx", C.c_void_p), ("sy", C.c_void_p), ("sz", C.c_void_p), ("count", C.c_void_p),
                ("first_idx", C.c_void_p), ("mean", C.c_void_p), ("cov", C.c_void_p), ("rough", C.c_void_p),
                ("normal", C.c_void_p), ("flags", C.c_void_p)]


class Robot(C.Structure):
    _fields_ = [("radius", C.c_float), ("reachable_height", C.c_float), ("max_rough", C.c_float), ("max_angle_deg", C.c_float)]


class CostStats(C.Structure):
    _fields_ = [("goal_status", C.c_int32), ("ring", C.c_uint32), ("levels", C.c_uint32), ("ring_store", C.c_uint32),
                ("traversable", C.c_uint64), ("closed", C.c_uint64), ("check_pushes", C.c_uint64)]


class PointLayout(C.Structure):
    _fields_ = [("point_step", C.c_uint32), ("offset_x", C.c_uint32), ("offset_y", C.c_uint32), ("offset_z", C.c_uint32)]


class Pcd(C.Structure):
    _fields_ = [("num_points", C.c_uint64), ("layout", PointLayout), ("data_kind", C.c_int32), ("reserved", C.c_int32),
                ("data", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("num_nodes", C.c_uint64), ("key", C.c_void_p), ("sums", C.c_void_p), ("count", C.c_void_p),
                ("first_idx", C.c_void_p)]


class ExchangeTimes(C.Structure):
    _fields_ = [("shard_ms", C.c_float), ("exchange_ms", C.c_float), ("finalize_ms", C.c_float), ("ranks", C.c_uint32),
                ("local_nodes", C.c_uint64), ("global_nodes", C.c_uint64), ("bytes_reduced", C.c_uint64)]


class CommSelftest(C.Structure):
    _fields_ = [("all_gather_ms", C.c_float), ("exchange_ms", C.c_float), ("reduce_scatter_ms", C.c_float), ("all_reduce_ms", C.c_float),
                ("ok_mask", C.c_uint32), ("ranks", C.c_uint32)]


class OwnedInfo(C.Structure):
    _fields_ = [("owned_points", C.c_uint64), ("local_nodes", C.c_uint64), ("local_columns", C.c_uint64),
                ("global_nodes", C.c_uint64), ("global_columns", C.c_uint64), ("global_slopes", C.c_uint64),
                ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64),
                ("split_ms", C.c_float), ("exchange_ms", C.c_float), ("build_ms", C.c_float), ("order_ms", C.c_float),
                ("ranks", C.c_uint32)]


def _hip_runtime_dir():
    """Directory of the HIP runtime libgndt must share with its host process.  PyTorch wheels bundle
    their own libamdhip64.so / libhsa-runtime64.so; linking libgndt against /opt/rocm's copy would put
    a second HIP runtime in the process (foreign streams, events and allocations).  GNDT_HIP_LIBDIR
    overrides (e.g. /opt/rocm/lib for a C++/ROS host without torch)."""
    env = os.environ.get("GNDT_HIP_LIBDIR")
    if env:
        return env
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = os.path.join(os.path.dirname(spec.origin), "lib")
            if os.path.exists(os.path.join(d, "libamdhip64.so")):
                return d
    except Exception:
        pass
    return "/opt/rocm/lib"


def source_hash():
    """sha256 over the kernel / host sources libgndt is built from: profiles/ store it next to the PMC figures so that
    bench.py can tell whether a committed HBM-traffic measurement belongs to the code it is timing."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(set(SOURCES + [x for x in HEADERS if not os.path.isabs(x)]))
    for nm in names + [os.path.join(_ROOT, "include", "gndt.h")]:
        with open(nm if os.path.isabs(nm) else os.path.join(_CSRC, nm), "rb") as f:
            h.update(os.path.basename(nm).encode() + b"\0" + f.read())
    return h.hexdigest()


def build_native(force=False, verbose=False):
    """Compile the HIP sources into grid_ndt_amd/csrc/libgndt.so for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(_CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in srcs:                       # one hipcc per translation unit, side by side (8 small jobs)
        obj = os.path.splitext(src)[0] + ".o"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c",
               "-Wno-unused-command-line-argument", "-Wno-unused-function", "-Wno-pass-failed", "-fno-slp-vectorize",
               # the code objects drop their local symbols (rocPRIM's ~500 instantiations in gndt_api_dist carried 1 MB of them).
               # NOT --strip-all: the HIP runtime segfaults on a code object without .symtab (measured on the MI355X box)
               "-Xoffload-linker", "--discard-all",
               "-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-o", obj, src] + os.environ.get("GNDT_EXTRA_CXXFLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    libdir = _hip_runtime_dir()
    link = ["g++", "-shared", "-o", LIB_PATH] + objs + ["-L", libdir, "-l:libamdhip64.so", "-Wl,-rpath," + libdir,
                                                         "-Wl,--no-undefined", "-Wl,--strip-all", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  grid_ndt_amd has no fallback implementation.")
    # One HIP runtime per process: load the copy the host process uses (torch's bundled one when torch
    # is installed) before libgndt, so libgndt's DT_NEEDED libamdhip64.so.7 binds to it by soname.
    rt = os.path.join(_hip_runtime_dir(), "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)
    L = C.CDLL(LIB_PATH)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    H = C.c_void_p
    L.gndt_create.argtypes = [C.POINTER(Params), C.POINTER(H)]
    L.gndt_destroy.argtypes = [H]
    L.gndt_destroy.restype = None
    L.gndt_last_error.argtypes = [H]
    L.gndt_last_error.restype = C.c_char_p
    L.gndt_set_origin.argtypes = [H, C.POINTER(C.c_float)]
    L.gndt_build.argtypes = [H, vp, sz, sz]
    L.gndt_build_device.argtypes = [H, vp, sz, sz, vp]
    L.gndt_update.argtypes = [H, vp, sz, sz]
    L.gndt_update_device.argtypes = [H, vp, sz, sz, vp]
    L.gndt_remove.argtypes = [H, vp, sz, sz]
    L.gndt_remove_device.argtypes = [H, vp, sz, sz, vp]
    L.gndt_reset.argtypes = [H, vp]
    L.gndt_accumulate_device.argtypes = [H, vp, sz, sz, u64, vp]
    L.gndt_finalize_device.argtypes = [H, vp]
    L.gndt_sync.argtypes = [H, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.gndt_export_device.argtypes = [H, C.POINTER(Cells)]
    L.gndt_export.argtypes = [H, C.POINTER(Cells)]
    L.gndt_export_host.argtypes = [H, C.POINTER(Cells)]
    L.gndt_reserve.argtypes = [H, u64, u64]
    L.gndt_set_deferred_emit.argtypes = [H, C.c_int]
    L.gndt_set_deferred_emit.restype = C.c_int
    L.gndt_reserve.restype = C.c_int
    L.gndt_export_host.restype = C.c_int
    L.gndt_stats_export_device.argtypes = [H, C.POINTER(Stats), vp]
    L.gndt_stats_merge_device.argtypes = [H, C.POINTER(Stats), vp]
    L.gndt_shard_stats_device.argtypes = [H, vp, C.c_size_t, C.c_size_t, u64, C.POINTER(Stats), vp]
    L.gndt_finalize_stats_device.argtypes = [H, C.POINTER(Stats), u64, vp]
    L.gndt_get_origin.argtypes = [H, C.POINTER(C.c_float)]
    L.gndt_pcd_read.argtypes = [C.c_char_p, C.POINTER(Pcd), C.c_char_p]
    L.gndt_pcd_free.argtypes = [C.POINTER(Pcd)]
    L.gndt_pcd_free.restype = None
    L.gndt_pack_points_device.argtypes = [H, vp, C.c_size_t, C.POINTER(PointLayout), vp, C.POINTER(u64), vp]
    L.gndt_build_cloud.argtypes = [H, vp, C.c_size_t, C.POINTER(PointLayout)]
    L.gndt_compute_cost.argtypes = [H, C.POINTER(C.c_float), C.POINTER(Robot), vp]
    L.gndt_cost_export_device.argtypes = [H, C.POINTER(vp), C.POINTER(vp), C.POINTER(CostStats)]
    L.gndt_cost_export.argtypes = [H, vp, vp, C.POINTER(CostStats)]
    L.gndt_trans_morton_xyz.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float), C.c_char_p,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p]
    L.gndt_count_morton.argtypes = [C.c_int32, C.c_int32, C.c_char_p]
    L.gndt_morton_to_xy.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gndt_pack_key.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.gndt_pack_key.restype = u64
    L.gndt_unpack_key.argtypes = [u64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gndt_unpack_key.restype = None
    L.gndt_set_profiling.argtypes = [H, C.c_int]
    L.gndt_get_phase_times.argtypes = [H, C.POINTER(C.c_double)]
    L.gndt_debug_bucket_phases.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.gndt_debug_bucket_phases.restype = C.c_int
    L.gndt_debug_retry_count.argtypes = [H, C.POINTER(u64)]
    L.gndt_debug_retry_count.restype = C.c_int
    L.gndt_debug_enable_stamps.argtypes = [C.c_int]
    L.gndt_debug_enable_stamps.restype = C.c_int
    L.gndt_warmup.argtypes = [C.c_void_p, C.c_uint64]
    L.gndt_warmup.restype = C.c_int
    L.gndt_debug_set_option.argtypes = [C.c_int, C.c_double]
    L.gndt_debug_set_option.restype = C.c_int
    L.gndt_debug_set_fp_bits.argtypes = [C.c_int]
    L.gndt_debug_set_fp_bits.restype = C.c_int
    L.gndt_debug_fp_clashes.argtypes = [H, C.POINTER(u64)]
    L.gndt_debug_fp_clashes.restype = C.c_int
    L.gndt_debug_second_pass_buckets.argtypes = [H, C.POINTER(u64)]
    L.gndt_debug_second_pass_buckets.restype = C.c_int
    L.gndt_debug_fail_next_alloc.argtypes = [H, C.c_int]
    L.gndt_debug_fail_next_alloc.restype = C.c_int
    L.gndt_comm_unique_id.argtypes = [C.c_char_p]
    L.gndt_comm_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.gndt_comm_destroy.argtypes = [vp]
    L.gndt_comm_create_threads.argtypes = [C.c_int32, C.c_int32, C.POINTER(vp)]
    L.gndt_comm_create_threads.restype = C.c_int
    L.gndt_comm_destroy.restype = None
    L.gndt_comm_last_error.restype = C.c_char_p
    L.gndt_comm_selftest.argtypes = [H, vp, C.POINTER(CommSelftest), vp]
    L.gndt_comm_selftest.restype = C.c_int
    L.gndt_build_global_device.argtypes = [H, vp, vp, C.c_size_t, C.c_size_t, u64, u64, C.POINTER(ExchangeTimes), vp]
    L.gndt_owner_of_columns.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp]
    L.gndt_owner_of_columns.restype = C.c_int
    L.gndt_owner_sample_device.argtypes = [H, vp, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(u64), vp]
    L.gndt_owner_map_device.argtypes = [H, vp, C.c_uint32, vp]
    L.gndt_owner_sample_device.restype = L.gndt_owner_map_device.restype = C.c_int
    L.gndt_owner_split_device.argtypes = [H, vp, C.c_size_t, C.c_size_t, u64, u64, C.c_uint32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), vp]
    L.gndt_build_records_device.argtypes = [H, vp, C.c_size_t, u64, vp]
    L.gndt_build_records2_device.argtypes = [H, vp, C.c_size_t, vp, C.c_size_t, u64, vp]
    L.gndt_build_records2_device.restype = C.c_int
    L.gndt_owned_columns_device.argtypes = [H, C.POINTER(vp), C.POINTER(u64), vp]
    L.gndt_owned_global_rows_device.argtypes = [H, vp, u64, u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), vp]
    L.gndt_build_owned_device.argtypes = [H, vp, vp, C.c_size_t, C.c_size_t, u64, u64, C.POINTER(vp), C.POINTER(OwnedInfo), vp]
    L.gndt_gather_owned_map_device.argtypes = [H, vp, C.c_int32, vp]
    L.gndt_owned_pack_rows_device.argtypes = [H, C.POINTER(vp), C.POINTER(u64), vp]
    L.gndt_adopt_rows_device.argtypes = [H, vp, u64, u64, u64, u64, vp]
    for name in ("gndt_gather_owned_map_device", "gndt_owned_pack_rows_device", "gndt_adopt_rows_device"):
        getattr(L, name).restype = C.c_int
    for name in ("gndt_comm_unique_id", "gndt_comm_create", "gndt_build_global_device", "gndt_owner_split_device",
                 "gndt_build_records_device", "gndt_owned_columns_device", "gndt_owned_global_rows_device", "gndt_build_owned_device"):
        getattr(L, name).restype = C.c_int
    L.gndt_locality_sample.argtypes = [H, vp, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(C.c_double), vp]
    L.gndt_locality_sample.restype = C.c_int
    L.gndt_last_strategy.argtypes = [H]
    L.gndt_last_strategy.restype = C.c_int
    L.gndt_device_info.argtypes = [C.c_int32, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(u64)]
    for name in ("gndt_create", "gndt_set_origin", "gndt_build", "gndt_build_device", "gndt_update", "gndt_update_device",
                 "gndt_remove", "gndt_remove_device",
                 "gndt_reset", "gndt_accumulate_device", "gndt_finalize_device", "gndt_sync", "gndt_export_device",
                 "gndt_export", "gndt_stats_export_device", "gndt_stats_merge_device", "gndt_trans_morton_xyz",
                 "gndt_compute_cost", "gndt_cost_export_device", "gndt_cost_export",
                 "gndt_shard_stats_device", "gndt_finalize_stats_device",
                 "gndt_pcd_read", "gndt_pack_points_device", "gndt_build_cloud", "gndt_get_origin",
                 "gndt_count_morton", "gndt_morton_to_xy", "gndt_device_info", "gndt_set_profiling", "gndt_get_phase_times"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L
