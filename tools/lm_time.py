"""GPU micro-timing of myslam_ba_optimize_batch (256 identical 10x300 windows)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); api, synth = pkg.api, pkg.synth
dev = "cuda"; P = 256
poses, pts, ep, el, obs, fixed, Kt = synth.ba_problem()
maxP, maxL, maxE = len(poses), len(pts), len(ep)
rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (P,) + a.shape))).to(dev)
b = [rep(poses), rep(pts), rep(ep), rep(el), rep(obs), rep(fixed), torch.tensor([[maxP, maxL, maxE]] * P, dtype=torch.int32, device=dev)]
scr = torch.zeros(P * maxE * 18, dtype=torch.float64, device=dev)
chi = torch.zeros(P, dtype=torch.float64, device=dev); it = torch.zeros(P, dtype=torch.int32, device=dev); st = torch.zeros(P, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for iters in (1, 2, 5, 10):
    sp, sx = b[0].clone(), b[1].clone()
    def run():
        sp.copy_(b[0]); sx.copy_(b[1])
        api.ba_optimize_batch(sp.data_ptr(), sx.data_ptr(), *[t.data_ptr() for t in b[2:]], P, maxP, maxL, maxE, Kt, 5.991, iters,
                              scr.data_ptr(), chi.data_ptr(), it.data_ptr(), st.data_ptr(), s)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    print(f"iters={iters}: {(time.perf_counter()-t0)*100:.3f} ms / 256 windows; it={int(it[0])} chi={float(chi[0]):.3f} E={maxE}")
tk = scr[maxE * 18 - 10: maxE * 18].cpu().numpy()
names = ["setup", "build", "G+init", "schur_stage", "schur_acc", "schur_flush", "chol", "solve", "xl+chi2+accept", "writeback"]
if tk.sum() > 0:
    print({n: round(float(v) / 100.0, 1) for n, v in zip(names, tk)}, "us (100 MHz ticks), last run =", "10 iters")
