"""Where a period of the dK/dV wave-pair kernel spends its cycles: needs a library built with -DODVAE_FLASH_STAMPS
(python tools/ab_build.py stamps flash_attn_bf16.hip -DODVAE_FLASH_STAMPS=1 -fno-slp-vectorize), selected with ODVAE_PROBE_LIB.
usage: ODVAE_PROBE_LIB=tools/bin/libodvae_stamps.so python tools/flash_stamps.py [N] [H]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
_lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
L = _lib.load()
N, H = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (8, 128)))
C = 256
dev = torch.device("cuda:0")
qkv = torch.randn(N, H, H, 3 * C, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
do = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    o = ops.attention_qkv(qkv)
    raw.odvae_flash_debug_stamps(None, 1)
    o.backward(do)
    qkv.grad = None
raw.odvae_flash_debug_stamps(buf, 0)
periods = H * H // 32
names = ["fetch issue", "ring fill (+ B: dS arithmetic)", "second product (16 MFMAs)", "first product (16 MFMAs)", "A: probabilities",
         "vmcnt wait", "barrier"]
for r, role in enumerate("AB"):
    tot = sum(buf[8 * r + k] for k in range(7))
    print("role %s: %.0f cycles per period" % (role, tot / periods))
    for k in range(7):
        print("   %-34s %7.0f  %5.1f %%" % (names[k], buf[8 * r + k] / periods, 100.0 * buf[8 * r + k] / max(tot, 1)))
