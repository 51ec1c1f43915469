#!/usr/bin/env python3
"""How many in-order chains of small kernels does the GPU run side by side?  L streams x N dependent one-block spin kernels (torch.cuda._sleep)
of ~40 us each; concurrency = L * N * t_kernel / wall.  Run with GPU_MAX_HW_QUEUES set to taste (the runtime maps streams onto that many
hardware queues).   python tools/queue_concurrency.py [cycles]"""
import json, os, sys, time
import torch
cyc = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
torch.cuda._sleep(cyc); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    torch.cuda._sleep(cyc)
torch.cuda.synchronize()
t1 = (time.perf_counter() - t0) / 50
out = {"hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"), "kernel_us": t1 * 1e6, "lanes": {}}
N = 40
for L in (1, 2, 4, 8, 16, 32, 64):
    ss = [torch.cuda.Stream() for _ in range(L)]
    for s in ss:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for s in ss:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["lanes"][L] = {"wall_ms": dt * 1e3, "concurrency": L * N * t1 / dt}
print(json.dumps(out))
