"""GPU micro-timing of the oct-tree kernel for a level-0-only extractor (budget 434 = level 0 of the 2000-feature, 8-level plan)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); api, synth = pkg.api, pkg.synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 376, 1241
imgs = synth.stereo_batch(P).reshape(2 * P, H, W)
d = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
s = torch.cuda.current_stream().cuda_stream
for nf, nl in ((434, 1), (2000, 8)):
    ext = api.ORBextractor(nf, nlevels=nl, stream=s)
    cap = ext.max_keypoints()
    kps = torch.zeros(2 * P * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(2 * P * cap * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2 * P, dtype=torch.int32, device="cuda"); st = torch.zeros(2 * P, dtype=torch.int32, device="cuda")
    for it in range(3):
        if it == 1: api.prof_reset(); api.prof_enable(True)
        ext.detect_and_compute_batch(d.data_ptr(), 2 * P, H, W, W, H * W, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), st.data_ptr(), cap)
    torch.cuda.synchronize()
    api.prof_enable(False); pr = api.prof_read()
    print(nf, nl, "kp/img", float(cnt.float().mean()), {k: round(v[0] / 2, 3) for k, v in pr.items() if v[0] > 0})
