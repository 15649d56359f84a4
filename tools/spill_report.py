"""Register / scratch report of every __global__ kernel in csrc/ (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the
flags of generative-detection_amd/build.py): name, VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, LDS, occupancy.
usage: python tools/spill_report.py [--all]   (default: only kernels that spill or use scratch)"""
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-detection_amd"))
import build as B  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [o.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for o in out]


def one(src):
    path = os.path.join(B.CSRC, src)
    cmd = [B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[bytes/lane\]|\[bytes/block\])?):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"file": src, "name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def report():
    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(one, B.HIP_SOURCES) for r in rs]
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = d
    return rows


if __name__ == "__main__":
    rows = report()
    show_all = "--all" in sys.argv
    print("%-22s %-60s %5s %5s %6s %6s %8s %7s" % ("file", "kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "LDS"))
    for r in rows:
        vs, sc = int(r.get("VGPRs Spill", 0)), int(r.get("ScratchSize [bytes/lane]", 0))
        if show_all or vs or sc:
            print("%-22s %-60s %5s %5s %6s %6s %8s %7s" % (r["file"], r["kernel"][:60], r.get("VGPRs"), r.get("AGPRs"), vs,
                                                         r.get("SGPRs Spill"), sc, r.get("LDS Size [bytes/block]")))
