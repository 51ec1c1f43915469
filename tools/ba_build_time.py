"""GPU micro-timing of myslam_ba_build_batch: the pose blocks by edge lists (default) against the earlier ds_add_f64 form
(MYSLAM_BA_OPT_BUILD_POSE_ATOMICS), at 1 window (1024 threads: a live stream's key-frame) and at 512 windows (256 threads each: the bench step).
Alternating runs in one process; the window is synth.ba_problem()'s 10 key-frames x 300 landmarks.   python tools/ba_build_time.py"""
import json, sys, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); api, synth = pkg.api, pkg.synth
dev = "cuda"
poses, pts, ep, el, obs, fixed, Kt = synth.ba_problem()
maxP, maxL, maxE = len(poses), len(pts), len(ep)
s = torch.cuda.current_stream().cuda_stream
out = {"window": [maxP, maxL, maxE], "unit": "us per launch"}
for W in (1, 8, 512):
    rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (W,) + a.shape))).to(dev)
    b = [rep(poses), rep(pts), rep(ep), rep(el), rep(obs), rep(fixed), torch.tensor([[maxP, maxL, maxE]] * W, dtype=torch.int32, device=dev)]
    o = [torch.zeros(W, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
    def run():
        api.ba_build_batch(*[t.data_ptr() for t in b], W, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in o], s)
    res = {"lists": [], "atomics": []}
    for rnd in range(3):
        for name, opt in (("lists", 0), ("atomics", 1)):
            api.ba_set_option(api.BA_OPT_BUILD_POSE_ATOMICS, opt)
            for _ in range(5): run()
            torch.cuda.synchronize()
            n = 200 if W < 64 else 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): run()
            e1.record(); torch.cuda.synchronize()
            res[name].append(round(e0.elapsed_time(e1) * 1000.0 / n, 2))
    api.ba_set_option(api.BA_OPT_BUILD_POSE_ATOMICS, 0)
    out[f"windows_{W}"] = res
    print(W, res, flush=True)
print(json.dumps(out))
