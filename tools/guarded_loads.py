"""Static scan of the gfx950 assembly of csrc/*.hip for chains of GUARDED loads: a vector-memory load inside an exec-masked branch
(`cond ? p[i] : 0`, `if (ptr) ... = ptr[i]`) is waited for on its own (s_waitcnt vmcnt(0) before the branch rejoins), so N of them in a
row cost N memory round trips in front of whatever needs them.  Found that way in round 4: 32 + 16 in front of the first MFMA of every
conv_bf16 block, 32 in the seeds of the softmax-backward GEMM.  usage: python tools/guarded_loads.py [min_count]   (needs hipcc)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    floor = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for src in sorted(glob.glob(os.path.join(ROOT, "generative-detection_amd", "csrc", "*.hip"))):
        out = "/tmp/gl_%s.s" % os.path.basename(src)[:-4]
        extra = ["-fno-slp-vectorize"] if "flash_attn" in src else []
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-S", src, "-o", out, "--cuda-device-only"] + extra,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        txt = open(out).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
            name, lines = m.group(1), m.group(2).split("\n")
            guarded = 0
            for i, l in enumerate(lines):
                if "s_and_saveexec" in l and any(("global_load" in x or "buffer_load" in x) for x in lines[i + 1:i + 8]):
                    guarded += 1
            if guarded >= floor:
                demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
                print("%-28s %-100s guarded loads %3d | vmcnt(0) waits %3d | MFMAs %4d" % (
                    os.path.basename(src), demangled.split("(")[0][:100], guarded, sum("vmcnt(0)" in l for l in lines), sum("v_mfma" in l for l in lines)))


if __name__ == "__main__":
    main()
