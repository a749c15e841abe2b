"""Build libdmh_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m depthmodelhardening_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree.
"""
import fcntl
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(REPO, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdmh_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = ["runtime.hip", "photo_loss.hip", "warp_view.hip", "smooth_loss.hip", "eot_paste.hip", "attack_ops.hip", "decoder_glue.hip", "roi_glue.hip", "roi_encoder.hip", "encoder_glue.hip", "wino_conv.hip", "wino32_conv.hip", "wino_wrw.hip", "small_conv.hip", "small_wrw.hip", "stem_conv_bwd.hip", "stem_conv_fwd.hip", "down_conv.hip", "down_wrw.hip", "stem_wrw.hip", "head_conv.hip", "layers_ops.hip"]
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent fp32 ops into v_pk_*_f32 and pays for it with register-pair
# shuffles (v_mov): on the fused loss kernel that was +30 % VALU instructions and +50 VGPRs (2 -> 4 waves/SIMD without it)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         "-I" + INCLUDE]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile what is stale and link.  Safe to call from several processes at once (torchrun ranks on a fresh
    checkout): an exclusive file lock serialises the whole build, every object and the library are written to a
    temporary name and moved into place atomically, so a reader never sees a half-written file."""
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(INCLUDE, "dmh_hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append(([hipcc] + FLAGS + ["-c", s, "-o"], o))

    def run(job):
        cmd, target = job
        tmp = "%s.tmp.%d" % (target, os.getpid())
        if verbose:
            print(" ".join(cmd + [target]), flush=True)
        r = subprocess.run(cmd + [tmp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        os.replace(tmp, target)
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        run(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o"], LIB_PATH))
    return LIB_PATH


ASAN_LIB_PATH = os.path.join(LIB_DIR, "libdmh_hip_asan.so")


def build_asan(verbose=False):
    """The HOST side of every source under AddressSanitizer: ``hipcc --offload-host-only -fsanitize=address`` (no device code:
    GPU ASan is not available on this pool, and the host side -- argument checks, workspace sizing, error formatting -- is
    what runs before every launch).  Links lib/libdmh_hip_asan.so and the C driver tests/asan/host_checks.c against it;
    returns the driver's path.  tests/test_asan_host.py runs it in the CPU suite."""
    hipcc = _hipcc()
    odir = os.path.join(OBJ_DIR, "asan")
    os.makedirs(odir, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    # --offload-new-driver: the host-only object then carries no reference to a device fat binary (the classic driver leaves
    # an undefined __hip_fatbin_<hash> behind), so the library links and loads without any device code
    flags = ["--offload-arch=gfx950", "--offload-host-only", "--offload-new-driver", "-O1", "-g", "-fsanitize=address",
             "-fno-omit-frame-pointer",
             "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I" + INCLUDE]
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(INCLUDE, "dmh_hip.h")]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("ASan host build failed:\n" + " ".join(cmd) + "\n" + r.stdout)
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            objs, jobs = [], []
            for src in SOURCES:
                s_, o = os.path.join(CSRC, src), os.path.join(odir, src.replace(".hip", ".o"))
                objs.append(o)
                if _stale(o, [s_] + headers):
                    jobs.append([hipcc] + flags + ["-c", s_, "-o", o])
            with ThreadPoolExecutor(max_workers=4) as ex:
                list(ex.map(run, jobs))
            clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang")
            clang = clang if os.path.exists(clang) else "/opt/rocm/lib/llvm/bin/clang"
            rocm_lib = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib")
            rocm_lib = rocm_lib if os.path.exists(os.path.join(rocm_lib, "libamdhip64.so")) else "/opt/rocm/lib"
            if jobs or _stale(ASAN_LIB_PATH, objs):
                run([clang + "++", "-fsanitize=address", "-shared", "-fPIC"] + objs
                    + ["-L" + rocm_lib, "-lamdhip64", "-Wl,-rpath," + rocm_lib, "-o", ASAN_LIB_PATH])
            driver_src = os.path.join(REPO, "tests", "asan", "host_checks.c")
            driver = os.path.join(odir, "host_checks")
            if _stale(driver, [driver_src, ASAN_LIB_PATH] + headers):
                run([clang, "-std=c11", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-Wall", "-I" + INCLUDE,
                     driver_src, "-L" + LIB_DIR, "-ldmh_hip_asan", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-o", driver])
            return driver
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    else:
        print(build(force="--force" in sys.argv))
