#!/usr/bin/env python3
"""What this box's HBM sustains for the access patterns the pipeline has, without libgndt (torch kernels): a device-to-device copy
(read + write streams, like the partition levels), a read-only reduction, a write-only fill, and a 96-byte-row gather by a random
permutation (like k_emit_rows).  TB/s of bytes moved (copy = 2 x the buffer).  JSON to stdout; GPU box."""
import json
import time

import torch


def rate(fn, bytes_moved, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round(bytes_moved * reps / (time.perf_counter() - t0) / 1e12, 2)


def main():
    n = 160_000_000          # bytes: the 10 M-point scene's record array
    a = torch.empty(n // 4, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    out = {"copy_160MB_TBps": rate(lambda: b.copy_(a), 2 * n), "read_sum_160MB_TBps": rate(lambda: a.sum(), n),
           "fill_160MB_TBps": rate(lambda: b.fill_(1.0), n)}
    big = torch.empty(1_000_000_000 // 4, dtype=torch.float32, device="cuda").normal_()
    big2 = torch.empty_like(big)
    out["copy_1GB_TBps"] = rate(lambda: big2.copy_(big), 2 * big.numel() * 4, reps=10)
    rows = torch.empty(800_000, 24, dtype=torch.float32, device="cuda").normal_()          # 96-byte rows
    perm = torch.randperm(800_000, device="cuda")
    dst = torch.empty_like(rows)
    out["gather_96B_rows_TBps"] = rate(lambda: torch.index_select(rows, 0, perm, out=dst), 2 * rows.numel() * 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
