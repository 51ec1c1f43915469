"""Builds libmyslam_hip.so (gfx950) in-tree with hipcc, and with g++ the two compiled host programs: bin/run_kitti_stereo (BASELINE configs[0],
app/run_kitti_stereo.cpp: plain C++ over the C ABI) and bin/sharded_db_rccl (BASELINE configs[4]'s loop-database exchange,
app/sharded_db_rccl.cpp: C++ + librccl + the C ABI, one process per GPU).  No torch, no cmake: plain compiler invocations.

    python build.py            # incremental
    python build.py --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmyslam_hip.so")
OBJDIR = os.path.join(HERE, "build")
APP_SRC = os.path.join(HERE, "app", "run_kitti_stereo.cpp")
APP_OUT = os.path.join(HERE, "bin", "run_kitti_stereo")
RCCL_SRC = os.path.join(HERE, "app", "sharded_db_rccl.cpp")
RCCL_OUT = os.path.join(HERE, "bin", "sharded_db_rccl")
CXX = os.environ.get("CXX", "g++")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# float-derived integers (BRIEF coordinates, fastAtan2, resize tables) must not see FMA contraction
EXACT = ["-ffp-contract=off"]
UNITS = {
    "orb_kernels.hip": EXACT + ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "orb_engine.hip": EXACT,
    "match_tri.hip": EXACT + ["-mllvm", "-amdgpu-mfma-vgpr-form"],     # MFMA accumulators in VGPRs: the arg-max reads them directly
    "calc.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "lcddb.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "ba.hip": [],
    "lk.hip": EXACT,
    "pgo.hip": [],
    "pnp.hip": EXACT,
    "prof.hip": [],
    "io.hip": [],
    "graph.hip": [],
}


def _deps():
    host = os.path.join(HERE, "host")          # header-only host code (formats, the Caffe readers) that csrc/io.hip and csrc/calc.hip include
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
           [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "myslam_hip.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_id():
    """Digest of every source the library is built from (csrc/, host/, include/myslam_hip.h, this file's flags): myslam_hip_version() carries it"""
    import hashlib
    h = hashlib.sha256()
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))] + [d for d in _deps() if not d.startswith(CSRC)])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    h.update(repr((COMMON, sorted(UNITS.items()))).encode())
    return h.hexdigest()[:12]


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    deps = _deps()
    jobs = []
    bid = build_id()
    bid_file = os.path.join(OBJDIR, "build_id.txt")
    bid_changed = not os.path.exists(bid_file) or open(bid_file).read().strip() != bid
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if src == "prof.hip":                   # carries the digest: recompiled whenever any source changed
            extra = extra + ['-DMYSLAM_BUILD_ID="%s"' % bid]
        if force or _stale(o, [s] + deps) or (src == "prof.hip" and bid_changed):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compiler failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    with open(bid_file, "w") as f:
        f.write(bid + "\n")
    objs = [os.path.join(OBJDIR, src.replace(".hip", ".o")) for src in UNITS]
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
    build_app(force or _stale(APP_OUT, [APP_SRC, OUT] + deps), verbose)
    build_rccl_host(force or _stale(RCCL_OUT, [RCCL_SRC, OUT] + deps), verbose)
    return OUT


def build_app(force=False, verbose=False):
    """bin/run_kitti_stereo: host C++ only (g++), links the library next to it"""
    if force or not os.path.exists(APP_OUT):
        os.makedirs(os.path.dirname(APP_OUT), exist_ok=True)
        cmd = [CXX, "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", APP_SRC, "-o", APP_OUT, "-L" + HERE, "-lmyslam_hip", "-pthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compiler failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return APP_OUT


def rccl_available():
    return os.path.exists("/opt/rocm/include/rccl/rccl.h") and any(os.path.exists(os.path.join("/opt/rocm/lib", n)) for n in ("librccl.so", "librccl.so.1"))


def build_rccl_host(force=False, verbose=False):
    """bin/sharded_db_rccl: the multi-GPU loop-database host (g++ + the HIP runtime + librccl + the library next to it).  Only this program needs RCCL:
    on a box without its header or library it is skipped with a warning (the library, run_kitti_stereo and every single-GPU path build and run
    without it; tests/test_gpu_facade.py skips the program's test when the binary is absent)."""
    if not rccl_available():
        if os.path.exists(RCCL_OUT):
            os.remove(RCCL_OUT)                    # never leave a stale binary behind a library it was not linked against
        sys.stderr.write("[build.py] rccl.h / librccl not found under /opt/rocm: bin/sharded_db_rccl (the multi-GPU loop-database host) is not built\n")
        return None
    if force or not os.path.exists(RCCL_OUT):
        os.makedirs(os.path.dirname(RCCL_OUT), exist_ok=True)
        cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(HERE, "..", "include"), RCCL_SRC, "-o", RCCL_OUT,
               "-L" + HERE, "-lmyslam_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compiler failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return RCCL_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
