"""GPU micro-timing of myslam_hamming_match_batch at 1 / 2 / 4 / 512 pairs of 2000 x 2000 descriptors (us per launch).   python tools/hamming_time.py"""
import json, sys, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.api
cap = 2000; out = {}
s = torch.cuda.current_stream().cuda_stream
for B in (1, 2, 4, 512):
    q = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device="cuda"); t = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.full((B,), cap, dtype=torch.int32, device="cuda")
    di = torch.zeros(B, cap, dtype=torch.int32, device="cuda"); dd = torch.zeros_like(di)
    run = lambda: api.hamming_match_batch(q.data_ptr(), n.data_ptr(), t.data_ptr(), n.data_ptr(), B, cap, di.data_ptr(), dd.data_ptr(), s)
    for _ in range(5): run()
    torch.cuda.synchronize()
    r = []
    for _ in range(3):
        N = 200 if B < 64 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): run()
        e1.record(); torch.cuda.synchronize()
        r.append(round(e0.elapsed_time(e1) * 1000 / N, 2))
    out[f"pairs_{B}"] = r
print(json.dumps(out))
