#!/usr/bin/env python3
"""Probe: how fast do pinned host -> device copies run beside the extractor kernels, and do they overlap?  (diagnostic for bench.py's
streamed pass)   python tools/stream_probe.py [pairs]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
import torch
pkg = load_package(); api, synth = pkg.api, pkg.synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 376, 1241
dev = torch.device("cuda", 0)
fr = synth.stereo_batch(8)
imgs = np.concatenate([np.tile(fr[:, 0], (P // 8, 1, 1)), np.tile(fr[:, 1], (P // 8, 1, 1))], 0)
h = torch.from_numpy(imgs).pin_memory()
d_a = torch.from_numpy(imgs).to(dev); d_b = torch.empty_like(d_a)
main = torch.cuda.current_stream(); sC = torch.cuda.Stream()
ext = api.ORBextractor(2000, stream=main.cuda_stream); cap = ext.max_keypoints()
kps = torch.zeros(2 * P * cap * 28, dtype=torch.uint8, device=dev); desc = torch.zeros(2 * P * cap * 32, dtype=torch.uint8, device=dev)
cnt = torch.zeros(2 * P, dtype=torch.int32, device=dev); stat = torch.zeros(2 * P, dtype=torch.int32, device=dev)
def compute():
    ext.detect_and_compute_batch(d_a.data_ptr(), 2 * P, H, W, W, H * W, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), stat.data_ptr(), cap)
def copy():
    with torch.cuda.stream(sC):
        d_b.copy_(h, non_blocking=True)
for _ in range(3): compute(); copy()
torch.cuda.synchronize()
N = 10
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
nb = h.numel()
tc = t(lambda: [compute() for _ in range(N)])
tm = t(lambda: [copy() for _ in range(N)])
def both():
    for _ in range(N): compute(); copy()
tb = t(both)
def both_host_timed():
    ts = []
    for _ in range(N):
        a = time.perf_counter(); compute(); b = time.perf_counter(); copy(); c = time.perf_counter(); ts.append((b - a, c - b))
    return ts
torch.cuda.synchronize(); ts = both_host_timed(); torch.cuda.synchronize()
print(f"pairs {P}: compute alone {tc:.2f} ms, copy alone {tm:.2f} ms = {nb / tm / 1e6:.1f} GB/s, both {tb:.2f} ms per iteration "
      f"(sum {tc + tm:.2f}, max {max(tc, tm):.2f}); host time in enqueue: compute {np.mean([x[0] for x in ts]) * 1e3:.2f} ms, copy {np.mean([x[1] for x in ts]) * 1e3:.2f} ms")
# the same with a raw hipMemcpyAsync through ctypes on a non-torch stream? torch is the plumbing bench.py uses: report only

# ---- the streamed pass's structure: two device buffers, copy(k) -> compute(k); copy(k+2) waits for compute(k) ----
d_in = [torch.empty_like(d_a) for _ in range(2)]
ev_ready = [torch.cuda.Event() for _ in range(2)]; ev_read = [torch.cuda.Event() for _ in range(2)]
def compute_from(buf):
    ext.detect_and_compute_batch(buf.data_ptr(), 2 * P, H, W, W, H * W, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), stat.data_ptr(), cap)
def streamed(n, k0=0):
    for k in range(k0, k0 + n):
        b = k % 2
        sC.wait_event(ev_read[b])
        with torch.cuda.stream(sC):
            d_in[b].copy_(h, non_blocking=True)
        ev_ready[b].record(sC)
        main.wait_event(ev_ready[b])
        compute_from(d_in[b])
        ev_read[b].record(main)
streamed(4); torch.cuda.synchronize()
ts = t(lambda: streamed(N, 4))
print(f"double-buffered with event dependencies, extractor on the default stream: {ts:.2f} ms per step")
s1 = torch.cuda.Stream()
ext2 = api.ORBextractor(2000, stream=s1.cuda_stream)
def streamed2(n, k0=0):
    for k in range(k0, k0 + n):
        b = k % 2
        sC.wait_event(ev_read[b])
        with torch.cuda.stream(sC):
            d_in[b].copy_(h, non_blocking=True)
        ev_ready[b].record(sC)
        s1.wait_event(ev_ready[b])
        ext2.detect_and_compute_batch(d_in[b].data_ptr(), 2 * P, H, W, W, H * W, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), stat.data_ptr(), cap)
        ev_read[b].record(s1)
streamed2(4); torch.cuda.synchronize()
ts2 = t(lambda: streamed2(N, 4))
print(f"the same with the extractor on a torch side stream: {ts2:.2f} ms per step")
