"""CPU tier: the C-ABI library loads, exports every symbol include/gndt.h declares, and refuses to
compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gndt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gndt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("gndt_create", "gndt_destroy", "gndt_set_origin", "gndt_build", "gndt_build_device", "gndt_update",
                 "gndt_export", "gndt_export_device", "gndt_trans_morton_xyz", "gndt_last_error", "gndt_sync",
                 "gndt_stats_export_device", "gndt_stats_merge_device", "gndt_accumulate_device", "gndt_finalize_device"):
        assert must in syms


def test_library_exports_every_declared_symbol(native_lib):
    raw = C.CDLL(native_lib._name)
    for s in declared_symbols():
        assert hasattr(raw, s), f"libgndt.so does not export {s}"


def test_no_cpu_fallback(native_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import grid_ndt_amd as g
    m = g.TwoDmap(0.5, 0.1)
    m.setCloudFirst((0, 0, 0))
    import numpy as np
    with pytest.raises(g.GndtError) as e:
        m.create2DMap("slope", np.zeros((10, 3), np.float32))
    assert e.value.code == 2   # GNDT_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    bad = []
    for base in ("grid_ndt_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                    t = open(os.path.join(dp, f)).read()
                    if re.search(r"(from|import)\s+oracle|liboracle|ref_cpu", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_library_reads_no_environment_variable_and_options_are_calls(native_lib):
    """VERDICT r5 housekeeping: the last getenv reads of the product library are gndt_debug_set_option fields."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "grid_ndt_amd", "csrc")):
        for f in fs:
            if f.endswith((".hip", ".cpp", ".hpp", ".h")) and re.search(r"\bgetenv\s*\(", open(os.path.join(dp, f)).read()):
                bad.append(f)
    assert not bad, bad
    import grid_ndt_amd as g
    T = g.TwoDmap
    for opt, val in ((T.DEBUG_VERBOSE, 1), (T.DEBUG_VERBOSE, 0), (T.DEBUG_TILE_RATIO, 32.0), (T.DEBUG_TILE_RATIO, 48.0),
                     (T.DEBUG_COST_ONE_WORKGROUP, 0), (T.DEBUG_COST_ONE_WORKGROUP, 1)):
        T.set_debug_option(opt, val)
    for opt, val in ((0, 1), (99, 1), (T.DEBUG_TILE_RATIO, 0.5), (T.DEBUG_TILE_RATIO, float("nan"))):
        with pytest.raises(g.GndtError) as e:
            T.set_debug_option(opt, val)
        assert e.value.code == 1   # GNDT_ERR_INVALID


def test_graph_capture_pauses_the_cycle_collector_and_gives_it_back(monkeypatch):
    """grid_ndt_amd.graph_capture = torch.cuda.graph with Python's cycle collector paused: a collection that starts inside a capture
    can run another object's destructor (a handle's gndt_destroy, a tensor's free), which invalidates the capture and makes torch
    abort the process.  No GPU here: torch.cuda.graph is replaced by a recorder."""
    import contextlib
    import gc
    import torch
    import grid_ndt_amd as g
    seen = {}

    @contextlib.contextmanager
    def fake_graph(graph, stream=None):
        seen["enter"] = (graph, stream, gc.isenabled())
        yield
        seen["exit"] = gc.isenabled()

    monkeypatch.setattr(torch.cuda, "graph", fake_graph)
    assert gc.isenabled()
    with g.graph_capture("G", "S"):
        assert not gc.isenabled()
    assert seen["enter"] == ("G", "S", False) and seen["exit"] is False and gc.isenabled()
    with pytest.raises(RuntimeError):
        with g.graph_capture("G"):
            raise RuntimeError("a refused capture")
    assert seen["enter"] == ("G", None, False) and gc.isenabled()
    gc.disable()
    try:
        with g.graph_capture("G"):
            pass
        assert not gc.isenabled()          # (a caller that had it off keeps it off)
    finally:
        gc.enable()
