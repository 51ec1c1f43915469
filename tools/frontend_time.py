"""Throughput of the frontend operators on device-resident batches (GPU box): LK left -> right tracking of 2000 points per frame
(Frontend::FindFeaturesInRight) and the pose-only optimisation of 500 matches per frame (Frontend::EstimateCurrentPose)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from __graft_entry__ import load_package
pkg = load_package(); api, synth = pkg.api, pkg.synth
dev = torch.device("cuda:0")
H, W = synth.IMG_H, synth.IMG_W
B = int(os.environ.get("FRAMES", "256")); NP = 2000
frames = synth.stereo_batch(8)                       # (8, 2, H, W)
L = torch.from_numpy(np.ascontiguousarray(frames[:, 0])).to(dev).repeat(B // 8, 1, 1).contiguous()
R = torch.from_numpy(np.ascontiguousarray(frames[:, 1])).to(dev).repeat(B // 8, 1, 1).contiguous()
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(20, W - 20, (B, NP)), rng.uniform(20, H - 20, (B, NP))], -1).astype(np.float32)
d_p0 = torch.from_numpy(pts).to(dev); d_p1 = d_p0.clone()
d_cnt = torch.full((B,), NP, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, NP, dtype=torch.uint8, device=dev); d_err = torch.zeros(B, NP, device=dev)
lk = api.LKTracker(stream=torch.cuda.current_stream().cuda_stream)
def run_lk():
    d_p1.copy_(d_p0)
    lk.track_batch(L.data_ptr(), R.data_ptr(), B, H, W, W, H * W, d_p0.data_ptr(), d_p1.data_ptr(), d_cnt.data_ptr(), NP, d_st.data_ptr(), d_err.data_ptr())
run_lk(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): run_lk()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"LK: {B} frame pairs x {NP} points in {dt*1e3:.2f} ms = {B/dt:.0f} frame pairs/s, {B*NP/dt/1e6:.1f} M points/s, tracked {float(d_st.float().mean()):.2f}", flush=True)

NM = 500; FB = int(os.environ.get("POSE_FRAMES", "1024"))
Kt = (synth.KITTI00["fx"], synth.KITTI00["fy"], synth.KITTI00["cx"], synth.KITTI00["cy"])
P3 = np.stack([rng.uniform(-10, 10, (FB, NM)), rng.uniform(-3, 3, (FB, NM)), rng.uniform(5, 40, (FB, NM))], -1)
uv = np.stack([Kt[0] * P3[..., 0] / P3[..., 2] + Kt[2], Kt[1] * P3[..., 1] / P3[..., 2] + Kt[3]], -1) + rng.normal(0, 0.5, (FB, NM, 2))
uv[:, ::10] += 25.0
T0 = np.tile(np.array([0, 0, 0, 1, 0.05, -0.02, 0.1]), (FB, 1))
d_T0 = torch.from_numpy(T0).to(dev); d_T = d_T0.clone()
d_P3 = torch.from_numpy(P3).to(dev); d_uv = torch.from_numpy(uv).to(dev)
d_n = torch.full((FB,), NM, dtype=torch.int32, device=dev)
d_out = torch.zeros(FB, NM, dtype=torch.uint8, device=dev); d_inl = torch.zeros(FB, dtype=torch.int32, device=dev); d_s = torch.zeros(FB, dtype=torch.int32, device=dev)
def run_po():
    d_T.copy_(d_T0)
    api.pose_only_optimize_batch(d_T.data_ptr(), d_P3.data_ptr(), d_uv.data_ptr(), d_n.data_ptr(), FB, NM, Kt, 5.991, 4, 10, d_out.data_ptr(), d_inl.data_ptr(), d_s.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
run_po(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): run_po()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"pose-only: {FB} frames x {NM} matches in {dt*1e3:.2f} ms = {FB/dt:.0f} frames/s, inliers {float(d_inl.float().mean()):.0f}", flush=True)
