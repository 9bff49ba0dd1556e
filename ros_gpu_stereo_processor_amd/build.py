"""Builds libgpuimageproc_hip.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m ros_gpu_stereo_processor_amd.build [--force]

hipcc cross-compiles without a GPU; the .so lands next to this file under lib/ (git-ignored, but it travels to the
GPU box with the gpurun snapshot).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libgpuimageproc_hip.so")
SOURCES = ["k_stereobm_c4_ra.hip", "k_stereobm_c4_rb.hip", "k_stereobm_c4_rc.hip", "k_stereobm_c4_rd.hip", "k_stereobm_c4_re.hip", "k_stereobm_c4_rf.hip",
           "k_stereobm_c4_rg.hip", "k_image.hip", "k_stereobm.hip", "k_stereobm_c4.hip",
           "k_stereobm_wide.hip", "k_post.hip", "gip_capi.hip", "gip_rowsplit.hip", "gip_streams.hip", "gip_model.cpp", "gip_streamer.cpp", "gip_rosmsg.cpp"]
HEADERS = [os.path.join(CSRC, "gip_internal.h"), os.path.join(CSRC, "gip_model.h"), os.path.join(CSRC, "bm_common.h"), os.path.join(CSRC, "gip_handle_view.h"), os.path.join(CSRC, "k_stereobm_c4_impl.h"),
           os.path.join(ROOT, "include", "gpuimageproc_hip.h")]
ARCH = os.environ.get("GIP_OFFLOAD_ARCH", "gfx950")
# -ffp-contract=off: the float/double paths (disparity->f32, reprojection, colour ramp, remap tables) must round
# exactly like the CPU oracle, so no mul+add fusion on either host or device.
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and not _stale(obj, [path] + HEADERS):
        return obj
    cmd = [hipcc(), "-c", "--offload-arch=" + ARCH] + CXXFLAGS
    if src.endswith(".cpp"):
        cmd += ["-x", "hip"]
    cmd += [path, "-o", obj]
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "--offload-arch=" + ARCH, "-o", LIB] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
