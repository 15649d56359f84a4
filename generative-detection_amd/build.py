"""Builds libodvae_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C ABI."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(HERE, "libodvae_hip.so")
HIP_SOURCES = ["gemm_f32.hip", "conv3x3_f32.hip", "conv3x3_wino_f32.hip", "conv3x3_wino4_f32.hip", "conv3x3_wgrad_f32.hip", "conv3x3_wgrad_wino_f32.hip", "groupnorm.hip", "elementwise.hip",
               "gan_f32.hip", "lpips_f32.hip", "patch_u8.hip", "pose_f32.hip", "linear_f32.hip",
               "conv_bf16.hip", "conv_wgrad_bf16.hip", "flash_attn_bf16.hip", "bf16_ops.hip"]
CXX_SOURCES = ["runtime.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", CSRC]
FLAGS += os.environ.get("ODVAE_EXTRA_HIPCC_FLAGS", "").split()   # diagnostic builds (-DODVAE_STAMPS, ...); use with force=True
# Per-file code-generation switches (none in use).  Tried on flash_attn_bf16.hip: `-mllvm -amdgpu-mfma-vgpr-form=1` removes the
# 256 v_accvgpr_read/write per 32 MFMAs that hipcc's default register split puts at the loop back-edge of the attention kernels
# (592 vs 565 TFLOP/s forward), but the D >= 256 forward kernels then return wrong results (tests/test_bf16_gpu.py), so it stays off.
# In use: -fno-slp-vectorize for the attention file.  At -O3 hipcc packs adjacent f32 multiplies / adds of the softmax sections into
# v_pk_mul_f32 / v_pk_add_f32; beside MFMAs a packed f32 op costs more issue time than the two plain ones it replaces
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
PER_FILE_FLAGS = {"flash_attn_bf16.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    deps = [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "bf16_common.h"), os.path.join(CSRC, "gn_finalize.h"), os.path.abspath(__file__)]
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def build_library(force=False, verbose=False):
    """Compile every kernel source for gfx950 and link the shared library.  Returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + CXX_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", LIB_PATH, file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
