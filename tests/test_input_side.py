"""The input side: .pcd reader (pcl::io::loadPCDFile, publisher.cpp:19), PointCloud2-style
record unpacking and NaN stripping on the device (receiver.cpp:140-143, publisher.cpp:24-26).
The reference holds no data files (its README points at external datasets), so the fixtures are written here from the
published PCD v0.7 layout; numpy is the checker."""
import os

import numpy as np
import pytest

from grid_ndt_amd import scenes
from tests import parity


def _write_pcd(path, fields, sizes, types, counts, arr_bytes, n, kind, ascii_rows=None, width=None, height=1, with_points=True):
    with open(path, "wb") as f:
        hdr = ["# .PCD v0.7 - Point Cloud Data file format", "VERSION 0.7", "FIELDS " + " ".join(fields),
               "SIZE " + " ".join(str(s) for s in sizes), "TYPE " + " ".join(types), "COUNT " + " ".join(str(c) for c in counts),
               f"WIDTH {width if width is not None else n}", f"HEIGHT {height}", "VIEWPOINT 0 0 0 1 0 0 0"]
        if with_points:
            hdr.append(f"POINTS {n}")
        hdr.append(f"DATA {kind}")
        f.write(("\n".join(hdr) + "\n").encode())
        if kind.startswith("binary"):
            f.write(arr_bytes)
        else:
            f.write(("\n".join(ascii_rows) + "\n").encode())


def _lzf_compress(data):
    """A small LZF encoder for the fixtures (greedy, hash of 3 bytes): literal runs of <= 32 bytes and back references of
    3..264 bytes at distances <= 8192 — every construct the decoder has to handle."""
    data = bytes(data)
    n = len(data)
    out = bytearray()
    lit = bytearray()
    table = {}
    i = 0

    def flush():
        k = 0
        while k < len(lit):
            run = lit[k:k + 32]
            out.append(len(run) - 1)
            out.extend(run)
            k += 32
        lit.clear()

    while i < n:
        key = data[i:i + 3]
        j = table.get(key, -1) if i + 3 <= n else -1
        if i + 3 <= n:
            table[key] = i
        if j >= 0 and i - j <= 8192:
            m = 3
            while i + m < n and m < 264 and data[j + m] == data[i + m]:
                m += 1
            flush()
            dist = i - j - 1
            ln = m - 2
            if ln < 7:
                out.append((ln << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(ln - 7)
            out.append(dist & 0xFF)
            i += m
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def _cloud_with_nans(n, seed=3):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * 40 - 20).astype(np.float32)
    bad = rng.random(n) < 0.07
    xyz[bad, rng.integers(0, 3, bad.sum())] = np.nan
    xyz[rng.random(n) < 0.01, 2] = np.inf
    return xyz


def test_pcd_reader_binary_and_ascii(tmp_path, native_lib):
    import grid_ndt_amd as g
    xyz = _cloud_with_nans(5000)
    n = xyz.shape[0]
    # binary, pcl::PointXYZ as PCL writes it (x y z, 12-byte records)
    p = str(tmp_path / "xyz.pcd")
    _write_pcd(p, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], xyz.tobytes(), n, "binary")
    raw, step, off = g.read_pcd(p)
    assert (step, off) == (12, (0, 4, 8)) and raw.nbytes == n * 12
    assert np.array_equal(raw.view(np.float32).reshape(n, 3), xyz, equal_nan=True)
    # binary with other fields around and between: intensity(f4) x y rgb(u4) z normal(3 x f4) ring(u2 x 2)
    rec = np.dtype([("i", "<f4"), ("x", "<f4"), ("y", "<f4"), ("rgb", "<u4"), ("z", "<f4"), ("nrm", "<f4", 3), ("ring", "<u2", 2)])
    a = np.zeros(n, rec)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    a["i"] = 7.5
    a["rgb"] = 0x00FF00FF
    p = str(tmp_path / "wide.pcd")
    _write_pcd(p, ["intensity", "x", "y", "rgb", "z", "normal", "ring"], [4, 4, 4, 4, 4, 4, 2], ["F", "F", "F", "U", "F", "F", "U"],
               [1, 1, 1, 1, 1, 3, 2], a.tobytes(), n, "binary", width=n // 4, height=4, with_points=False)
    raw, step, off = g.read_pcd(p)
    assert step == rec.itemsize == 36 and off == (4, 8, 16)
    got = np.stack([raw.view(np.uint8).reshape(n, step)[:, o:o + 4].copy().view(np.float32)[:, 0] for o in off], 1)
    assert np.array_equal(got, xyz, equal_nan=True)
    # ascii, with a column before x and "nan" tokens
    rows = [f"{k} {r[0]:.9g} {r[1]:.9g} {r[2]:.9g}".replace("inf", "inf") for k, r in enumerate(xyz)]
    p = str(tmp_path / "ascii.pcd")
    _write_pcd(p, ["idx", "x", "y", "z"], [4, 4, 4, 4], ["U", "F", "F", "F"], [1, 1, 1, 1], None, n, "ascii", ascii_rows=rows)
    raw, step, off = g.read_pcd(p)
    assert (step, off) == (12, (0, 4, 8))
    assert np.array_equal(raw.view(np.float32).reshape(n, 3), xyz, equal_nan=True)    # %.9g round-trips fp32
    # binary_compressed: LZF over the fields stored one after the other (all x, all y, ...), here with runs that compress
    q = xyz.copy()
    q[100:900, 2] = 1.25                                    # a constant stretch -> long back references
    q[2000:2600] = q[1000:1600]                             # a repeated block
    a = np.zeros(n, rec)
    a["x"], a["y"], a["z"], a["i"], a["rgb"] = q[:, 0], q[:, 1], q[:, 2], 3.0, 0x00112233
    soa = b"".join(np.ascontiguousarray(a[f]).tobytes() for f in ("i", "x", "y", "rgb", "z", "nrm", "ring"))
    comp = _lzf_compress(soa)
    assert len(comp) < len(soa)
    payload = np.array([len(comp), len(soa)], "<u4").tobytes() + comp
    p = str(tmp_path / "lzf.pcd")
    _write_pcd(p, ["intensity", "x", "y", "rgb", "z", "normal", "ring"], [4, 4, 4, 4, 4, 4, 2], ["F", "F", "F", "U", "F", "F", "U"],
               [1, 1, 1, 1, 1, 3, 2], payload, n, "binary_compressed")
    raw, step, off = g.read_pcd(p)
    assert step == 36 and off == (4, 8, 16)
    assert raw.tobytes() == a.tobytes()                     # records rebuilt byte for byte
    # errors: missing file, corrupt compressed payload, truncated payload, no xyz
    with pytest.raises(g.GndtError):
        g.read_pcd(str(tmp_path / "missing.pcd"))
    p = str(tmp_path / "badlzf.pcd")
    bad = np.array([len(comp), len(soa)], "<u4").tobytes() + comp[:len(comp) // 2] + b"\xff\xff\xff" + comp[len(comp) // 2 + 3:]
    _write_pcd(p, ["intensity", "x", "y", "rgb", "z", "normal", "ring"], [4, 4, 4, 4, 4, 4, 2], ["F", "F", "F", "U", "F", "F", "U"],
               [1, 1, 1, 1, 1, 3, 2], bad[:-40], n, "binary_compressed")
    with pytest.raises(g.GndtError):
        g.read_pcd(p)
    p = str(tmp_path / "short.pcd")
    _write_pcd(p, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], xyz.tobytes()[:-8], n, "binary")
    with pytest.raises(g.GndtError):
        g.read_pcd(p)
    p = str(tmp_path / "noz.pcd")
    _write_pcd(p, ["x", "y", "w"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], xyz.tobytes(), n, "binary")
    with pytest.raises(g.GndtError):
        g.read_pcd(p)


def test_pcd_reader_rejects_headers_whose_sizes_wrap_or_outrun_the_file(tmp_path, native_lib):
    """A crafted header must not wrap POINTS * record size or make the reader write past its allocation."""
    import grid_ndt_amd as g
    xyz = np.arange(12, dtype=np.float32).reshape(4, 3)
    cases = {
        # POINTS * 12 == 2^64 + 8: malloc(8) then 12-byte writes (the round-1 finding)
        "wrap_ascii": dict(n=1537228672809129302, kind="ascii", rows=["1 2 3"]),
        "wrap_binary": dict(n=1537228672809129302, kind="binary"),
        "beyond_u32": dict(n=0xFFFFFFFF, kind="binary"),
        "more_points_than_bytes": dict(n=1000, kind="binary"),
        "more_points_than_lines": dict(n=1000, kind="ascii", rows=["1 2 3", "4 5 6"]),
        "zero_points": dict(n=0, kind="binary"),
    }
    for name, c in cases.items():
        p = str(tmp_path / (name + ".pcd"))
        _write_pcd(p, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], xyz.tobytes(), c["n"], c["kind"], ascii_rows=c.get("rows"))
        with pytest.raises(g.GndtError):
            g.read_pcd(p)
    # WIDTH x HEIGHT wrapping without a POINTS line, and SIZE x COUNT wrapping the record size
    p = str(tmp_path / "wh.pcd")
    _write_pcd(p, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], xyz.tobytes(), 4, "binary", width=1 << 40, height=1 << 40, with_points=False)
    with pytest.raises(g.GndtError):
        g.read_pcd(p)
    p = str(tmp_path / "step.pcd")
    _write_pcd(p, ["x", "y", "z", "blob"], [4, 4, 4, 4], ["F", "F", "F", "U"], [1, 1, 1, 0x40000000], xyz.tobytes(), 4, "binary")
    with pytest.raises(g.GndtError):
        g.read_pcd(p)
    # compressed: an uncompressed-size word that does not match, a compressed size beyond the file
    for name, words in (("usize", [8, 999]), ("csize", [1 << 30, 48])):
        p = str(tmp_path / (name + ".pcd"))
        _write_pcd(p, ["x", "y", "z"], [4, 4, 4], ["F", "F", "F"], [1, 1, 1], np.array(words, "<u4").tobytes() + b"\x00" * 8, 4, "binary_compressed")
        with pytest.raises(g.GndtError):
            g.read_pcd(p)


@pytest.mark.gpu
def test_pack_points_strips_nan_rows_and_keeps_order():
    import torch
    import grid_ndt_amd as g
    m = g.TwoDmap(0.5, 0.1)
    for n in (1, 63, 64, 65, 4097, 250_000):
        xyz = _cloud_with_nans(n, seed=n)
        keep = np.isfinite(xyz).all(1)
        # pcl::PointXYZ (16-byte records) and a 36-byte record with the fields out of order
        pxyz = np.zeros((n, 4), np.float32)
        pxyz[:, :3] = xyz
        pxyz[:, 3] = 1.0
        out = m.pack_points(torch.from_numpy(pxyz).cuda(), 16)
        assert np.array_equal(out.cpu().numpy(), xyz[keep])
        rec = np.dtype([("i", "<f4"), ("x", "<f4"), ("y", "<f4"), ("rgb", "<u4"), ("z", "<f4"), ("pad", "<f4", 4)])
        a = np.zeros(n, rec)
        a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        a["i"] = np.nan                                    # a NaN in another field must not drop the row
        out = m.pack_points(torch.from_numpy(a.view(np.uint8)).cuda(), rec.itemsize, (4, 8, 16))
        assert np.array_equal(out.cpu().numpy(), xyz[keep])
    with pytest.raises(g.GndtError):
        m.pack_points(torch.zeros(64, dtype=torch.uint8).cuda(), 10)      # misaligned layout


@pytest.mark.gpu
def test_pcd_file_to_grid_equals_the_oracle(tmp_path):
    """publisher + chatterCallback end to end: .pcd with NaN rows -> reader -> device NaN strip -> origin = first
    valid point -> grid; against the oracle run on the cleaned cloud."""
    import grid_ndt_amd as g
    cloud = scenes.campus_frame(120000)
    dirty = np.insert(cloud, [0, 0, 5, 777, 50000], np.float32([np.nan, 1.0, 2.0]), axis=0)     # NaN rows, two before the origin
    rec = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
    a = np.zeros(dirty.shape[0], rec)
    a["x"], a["y"], a["z"] = dirty[:, 0], dirty[:, 1], dirty[:, 2]
    p = str(tmp_path / "frame.pcd")
    _write_pcd(p, ["x", "y", "z", "rgb"], [4, 4, 4, 4], ["F", "F", "F", "U"], [1, 1, 1, 1], a.tobytes(), a.shape[0], "binary")
    raw, step, off = g.read_pcd(p)
    P = scenes.CAMPUS_PARAMS
    m = g.TwoDmap(P["grid_len"], P["z_len"])
    m.setInterval(P["slope_interval"])
    m.build_cloud(P["demand"], raw, step, off)
    parity.assert_parity(m.export(), parity.ref_from_cloud(cloud, P))
