"""Two FREE-RUNNING chains (<pkg>/chain.py through the HIP back end and through the oracle back end) side by side.  LK's initial flow is a
re-projection with a float pose (frontend.cpp:136-147): the two poses differ in the last bits, one ulp in the f32 start point moves a
converged track by up to ~5e-3 px (the iteration stops at 0.01 px) and now and then sends an ill-conditioned window to another local
solution a pixel away (measured, round 4) — whose feature then gets another outlier flag, after which the two feature lists differ in
length.  Two free runs therefore agree in what they estimate (key-frames, poses, landmarks), not entry by entry; the entry-by-entry check
on identical inputs is tests/oracle_backend.CheckedBackend.  This module walks the two logs as far as they are aligned and reports."""
import numpy as np


def walk_logs(a_log, b_log):
    rep = {"aligned_entries": 0, "entries": (len(a_log), len(b_log)), "diverged_at": None, "tracks": 0, "tracks_bit_identical": 0,
           "tracks_within_0.01px": 0, "tracks_within_0.03px": 0, "track_max_dev_px": 0.0, "init_flow_max_dev_px": 0.0, "pose_max_dev": 0.0,
           "flag_flips": 0}
    for k, ((ta, xa), (tb, xb)) in enumerate(zip(a_log, b_log)):
        if ta != tb or len(xa) != len(xb) or any(u.shape != v.shape for u, v in zip(xa, xb)):
            rep["diverged_at"] = (k, ta, tb)
            break
        rep["aligned_entries"] += 1
        if ta in ("lk_track", "lk_right"):
            st = xa[1].astype(bool) & xb[1].astype(bool)
            rep["flag_flips"] += int(np.count_nonzero(xa[1] != xb[1]))
            d = np.abs(xa[0][st] - xb[0][st]).max(axis=1) if st.any() else np.zeros(0)
            rep["tracks"] += len(d); rep["tracks_bit_identical"] += int(np.count_nonzero(d == 0))
            rep["tracks_within_0.01px"] += int(np.count_nonzero(d <= 0.01)); rep["tracks_within_0.03px"] += int(np.count_nonzero(d <= 0.03))
            if len(d):
                rep["track_max_dev_px"] = max(rep["track_max_dev_px"], float(d.max()))
            if len(xa) > 2 and xa[2].size:
                rep["init_flow_max_dev_px"] = max(rep["init_flow_max_dev_px"], float(np.abs(xa[2] - xb[2]).max()))
        elif ta in ("pose_only", "loop_pose"):
            rep["pose_max_dev"] = max(rep["pose_max_dev"], float(np.abs(xa[0] - xb[0]).max()))
            rep["flag_flips"] += int(np.count_nonzero(xa[1] != xb[1]))
    return rep


def compare_runs(a, b):
    """a, b: two finished chains over the same frames -> walk_logs' report + the trajectory-level deviations.  Nothing is asserted about the
    poses: the reference's local BA fixes no key-frame (backend.cpp:139-177), so the rigid gauge of every solve is free and follows the
    last bits of its input — two free runs drift apart by centimetres while each stays equally close to the truth."""
    rep = walk_logs(a.log, b.log)
    n = min(len(a.poses), len(b.poses))
    rep["frame_pose_max_dev"] = max(float(np.abs(np.asarray(a.poses[i]) - np.asarray(b.poses[i])).max()) for i in range(n)) if n else 0.0
    rep["map_points"] = (len(a.all_mps), len(b.all_mps))
    rep["same_key_frames"] = a.kf_frames == b.kf_frames
    rep["same_loops"] = [(x.id, y.id) for x, y in a.loops] == [(x.id, y.id) for x, y in b.loops]
    return rep
