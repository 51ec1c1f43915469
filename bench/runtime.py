"""bench/runtime.py — process-level helpers of bench.py: CU-masked streams (an experiment knob), the self-launch of N ranks, the host's core count."""
import os
import socket
import subprocess
import sys

from .config import ROOT

BENCH_PY = os.path.join(ROOT, "bench.py")

_masked = []


def masked_stream(n_cus, first=0):
    """A HIP stream whose kernels may only run on `n_cus` compute units (hipExtStreamCreateWithCUMask; bits first .. first + n_cus - 1 of the
    device's CU mask) as a torch stream — an experiment: does confining the latency-bound side chains to a few CUs keep their long-lived blocks
    out of FAST's way?  (--side-cus / --match-cus; DESIGN_APPENDIX.md section 8 has the result.)"""
    import ctypes
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    words = 8                                              # 256 CUs
    mask = (ctypes.c_uint32 * words)()
    for b in range(first, first + n_cus):
        mask[(b // 32) % words] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0 and st.value, f"hipExtStreamCreateWithCUMask failed: {rc}"
    _masked.append(st)
    return torch.cuda.ExternalStream(st.value)


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU) ourselves and pass rank 0's output on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def physical_cores():
    """(hardware threads this process may run on, physical cores among them — SMT siblings counted once —, CPU quota of the container's
    cgroup in CPUs or None)"""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    n_cores = max(1, len(cores))
    # a container's CPU-time quota (cgroup) can be far below its CPU affinity: threads beyond it only time-slice
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            quota = parse(open(path).read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return len(cpus), n_cores, quota
