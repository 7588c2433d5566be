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
