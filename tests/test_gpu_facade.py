"""Builds the C++ facade test (reference class names over the C ABI) and runs it on the GPU."""
import os
import subprocess

import pytest

from conftest import PKG_DIR, ROOT

pytestmark = pytest.mark.gpu


def test_cpp_facade_matches_oracle(api, oracle, tmp_path):
    exe = str(tmp_path / "facade_test")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-o", exe,
           "-L" + PKG_DIR, "-lmyslam_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FACADE TEST OK" in r.stdout, r.stdout + r.stderr


def test_c_abi_gated_handles(tmp_path):
    """INTEGRATION.md's two-handle pipeline through the plain C ABI + HIP runtime API (no Python in the loop)."""
    exe = str(tmp_path / "pipeline_test")
    cmd = ["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "pipeline_test.cpp"), "-o", exe, "-L" + PKG_DIR, "-lmyslam_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG_DIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PIPELINE TEST OK" in r.stdout, r.stdout + r.stderr


def test_cpp_sharded_db_over_rccl(synth, tmp_path):
    """The multi-GPU loop-database exchange as a compiled program (app/sharded_db_rccl.cpp -> bin/sharded_db_rccl, built by build.py): C++ + librccl + the C ABI, no Python in
    the loop — ncclCommInitRank, ncclAllGather of the queries, myslam_lcddb_query_batch_sharded, ncclAllGather of the 16-byte records,
    myslam_lcd_merge_candidates_device, checked against one scan of the whole database.  One rank per visible GPU (1 on this pool's boxes)."""
    import numpy as np
    import torch
    exe = os.path.join(PKG_DIR, "bin", "sharded_db_rccl")          # built by build.py (g++ + librccl + the library), as bin/run_kitti_stereo is
    if not os.path.exists(exe) and not (os.path.exists("/opt/rocm/include/rccl/rccl.h") and os.path.exists("/opt/rocm/lib/librccl.so")):
        pytest.skip("no RCCL on this box: build.py does not build the multi-GPU host (round 6: only this program needs librccl)")
    assert os.path.exists(exe), "build.py did not produce bin/sharded_db_rccl"
    n_db, nq = 3000, 64
    db = synth.lcd_database(n_db)
    rng = np.random.default_rng(11)
    q = db[rng.integers(0, n_db, nq)] * 0.98 + 0.02 * synth.lcd_database(nq, seed=5)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    cur = np.full(nq, n_db + 20, np.uint64); cur[::3] = rng.integers(30, n_db, len(cur[::3]))       # a third of the scans stop at the `cur - id < 20` cut-off
    (tmp_path / "db.f32").write_bytes(db.astype(np.float32).tobytes()); (tmp_path / "q.f32").write_bytes(q.tobytes()); (tmp_path / "cur.u64").write_bytes(cur.tobytes())
    world = max(1, torch.cuda.device_count())
    while nq % world:
        world -= 1
    env = dict(os.environ, WORLD_SIZE=str(world), MYSLAM_NCCL_ID_FILE=str(tmp_path / "nccl_id"), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    args = [exe, str(tmp_path / "db.f32"), str(tmp_path / "q.f32"), str(tmp_path / "cur.u64"), str(n_db), str(nq)]
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[1][-1500:]) for p, o in zip(procs, outs)]
    assert f"SHARDED DB RCCL OK ranks={world}" in outs[0][0] and "mismatches=0" in outs[0][0], outs[0][0]
    # every rank reports the per-stage times of the repeated exchange (the names bench.py --gpus N uses in its line)
    for r, o in enumerate(outs):
        line = [l for l in o[0].splitlines() if l.startswith(f"rank {r}:")]
        assert line and all(k in line[0] for k in ("allgather_queries_ms", "shard_scan_ms", "allgather_candidates_ms", "merge_ms")), o[0]
    print(outs[0][0].strip())


@pytest.mark.parametrize("local_shards", [1, 4])
def test_cpp_growing_sharded_db(synth, tmp_path, local_shards):
    """The GROWING database of round 6 in the compiled host (app/sharded_db_rccl.cpp, last argument = steps): per step the ranks all-gather their new
    key-frames, every shard answers with 32-byte owned records (myslam_lcddb_query_batch_owned), the records are all-gathered and merged on the device, the
    key-frames are appended by arrival order over WORLD_SIZE x MYSLAM_LOCAL_SHARDS shards; rank 0 checks every answer against ONE map, bit for bit, and the
    shards' row counts stay within one of each other.  One rank per visible GPU; 4 local shards interleave ownership for real on a one-GPU box."""
    import numpy as np
    import torch
    exe = os.path.join(PKG_DIR, "bin", "sharded_db_rccl")
    if not os.path.exists(exe) and not (os.path.exists("/opt/rocm/include/rccl/rccl.h") and os.path.exists("/opt/rocm/lib/librccl.so")):
        pytest.skip("no RCCL on this box")
    n_db, nq = 400, 8
    db = synth.lcd_database(n_db); q = db[:nq].copy(); cur = np.full(nq, n_db + 20, np.uint64)
    (tmp_path / "db.f32").write_bytes(db.astype(np.float32).tobytes()); (tmp_path / "q.f32").write_bytes(q.tobytes()); (tmp_path / "cur.u64").write_bytes(cur.tobytes())
    world = max(1, torch.cuda.device_count())
    while nq % world:
        world -= 1
    env = dict(os.environ, WORLD_SIZE=str(world), MYSLAM_NCCL_ID_FILE=str(tmp_path / "nccl_id"), MYSLAM_LOCAL_SHARDS=str(local_shards),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    args = [exe, str(tmp_path / "db.f32"), str(tmp_path / "q.f32"), str(tmp_path / "cur.u64"), str(n_db), str(nq), "0", "200"]
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[0][-500:], o[1][-1500:]) for p, o in zip(procs, outs)]
    line = [l for l in outs[0][0].splitlines() if l.startswith("GROWING SHARDED DB")]
    assert line and line[0].startswith(f"GROWING SHARDED DB OK shards={world * local_shards} ") and "steps=200" in line[0] and "mismatches=0" in line[0], outs[0][0]
    assert "accepted_loops=0 " not in line[0]                        # the file's rows repeat after 97 key-frames: equal rows in different shards are found as loops
    print(line[0])
