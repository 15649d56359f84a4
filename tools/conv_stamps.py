"""Diagnostic: phase timing of conv3x3_kernel_v2 from s_memtime stamps (separate -DODVAE_STAMPS build of the kernel).
Prints, per wave, cycles spent in prologue / main loop / barrier waits / epilogue against the 64-cycle MFMA budget."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "generative-detection_amd", "csrc")
SO = os.path.join(ROOT, "tools", "bin", "libodvae_stamps.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-DODVAE_STAMPS",
                           "-I", CSRC, os.path.join(CSRC, "conv3x3_f32.hip"), os.path.join(CSRC, "runtime.cpp"), "-o", SO])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        return
    L = ctypes.CDLL(SO)
    B, cin, cout, h = 32, 128, 128, 256
    dev = torch.device("cuda:0")
    x = torch.randn(B, h, h, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty(B, h, h, cout, device=dev)
    L.odvae_conv3x3_pack_floats.restype = ctypes.c_size_t
    pack = torch.empty(L.odvae_conv3x3_pack_floats(cin, cout), device=dev)
    P = ctypes.c_void_p
    L.odvae_conv3x3_pack_f32(P(w.data_ptr()), cout, cin, P(pack.data_ptr()), None, None)
    for _ in range(3):
        rc = L.odvae_conv3x3_f32(0, P(x.data_ptr()), B, h, h, cin, P(pack.data_ptr()), cout, P(b.data_ptr()), None, P(y.data_ptr()), h, h, 0, None)
        assert rc == 0
    torch.cuda.synchronize()
    n = 4096 * 4 * 4
    host = np.zeros(n, dtype=np.uint64)
    assert L.odvae_debug_read_stamps(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n)) == 0
    st = host.reshape(4096, 4, 4).astype(np.float64)
    nm = ["prologue", "main loop", "barrier+halo-store (inside main loop)", "epilogue"]
    mfma = 2304 * 64.0
    print("MFMA-only budget per wave: %.0f cycles (x2 when two waves share the SIMD)" % mfma)
    for i, name in enumerate(nm):
        v = st[:, :, i].ravel()
        print("%-40s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f cycles" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
    tot = st[:, :, 0] + st[:, :, 1] + st[:, :, 3]
    print("block lifetime per wave mean %.0f cycles; main loop / (2 x MFMA budget) = %.3f" % (tot.mean(), st[:, :, 1].mean() / (2 * mfma)))


if __name__ == "__main__":
    main()
