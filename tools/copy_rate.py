#!/usr/bin/env python3
"""What a plain device-to-device copy reaches on this GPU (read + write bytes per second): the practical ceiling next to which
the streaming stages (pyramid, blur) are to be read.  Sizes straddle the Infinity Cache (256 MB).  usage: python tools/copy_rate.py"""
import json
import torch

dev = torch.device("cuda", 0)
out = {}
for mb in (64, 128, 370, 740, 1480):
    n = mb * 1000 * 1000
    a = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    out[f"{mb}MB"] = {"ms": round(ms, 4), "read_plus_write_TBps": round(2 * n / ms / 1e9, 3)}
    # read-only (sum) and write-only (fill) for comparison
    e0.record()
    for _ in range(R):
        b.fill_(7)
    e1.record()
    torch.cuda.synchronize()
    out[f"{mb}MB"]["fill_TBps"] = round(n / (e0.elapsed_time(e1) / R) / 1e9, 3)
print(json.dumps(out))
