"""A/B build of the library: recompile ONE kernel source with extra -D flags and link it with the objects of the default build
into tools/bin/libodvae_<name>.so (select it with ODVAE_PROBE_LIB in the tools/*_time.py / *_probe.py scripts).
usage: python tools/ab_build.py <name> <source.hip> [-DFLAG=1 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "generative-detection_amd")
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, PKG)
import build as B
B.build_library()
os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
obj = os.path.join(ROOT, "tools", "bin", "%s_%s.o" % (name, src))
subprocess.run([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
objs = [obj if f == src + ".o" else os.path.join(B.OBJ_DIR, f) for f in sorted(os.listdir(B.OBJ_DIR))]
out = os.path.join(ROOT, "tools", "bin", "libodvae_%s.so" % name)
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
os.remove(obj)
print(out)
