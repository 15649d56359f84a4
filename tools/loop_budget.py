"""Instruction-slot budget of a kernel's innermost loop from its gfx950 assembly ("write the instruction-slot budget of the
main loop so the ceiling is a number, not a feeling").

    python tools/loop_budget.py csrc-file.hip 'kernel name regex' [--depth 2]

Compiles the file to assembly with the library's flags (device only), finds the kernel, splits the loop of the given depth into basic blocks
and counts, per block, the instructions by issue class.  Issue prices are the measured ones of MI355X_MICROARCH.md ("Per-instruction cycle
constants", "LDS") and of this repository's own probes (profiles/r02_wino8_loop.md: on gfx950 the f32 MFMA and the vector ALU share their
issue cycles -- every v_* instruction beside a v_mfma_f32_32x32x2_f32 costs its own ~4 cycles, nothing executes under it):
    v_mfma_f32_32x32x2_f32 64 | v_mfma_f32_16x16x4_f32 32 | v_mfma_*_bf16 32x32x16 32 / 16x16x32 16
    plain VALU 4 | packed f32 (v_pk_*) 8 | transcendental (v_exp/rcp/rsq/sqrt/log/sin/cos) 8
    ds_read_b32/b64 2-cycle array slots, b128 4; ds_write_b32 4, b64 6, b128 13 (address + data transfer)
    vector memory (buffer_/global_ loads, stores, LDS-DMA) 4 issue cycles of the wave (latency is not an issue cost)
Scalar instructions, waits, barriers and branches are listed but not priced: the scalar unit issues beside the vector pipe.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-detection_amd"))
import build as B  # noqa: E402

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "valu"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def price(op, cls):
    if cls == "mfma":
        if "32x32x2_f32" in op or "32x32x2f32" in op:
            return 64
        if "16x16x4_f32" in op or "16x16x4f32" in op:
            return 32
        if "32x32x16" in op:
            return 32
        if "16x16x32" in op:
            return 16
        return 32
    if cls == "valu":
        return 4
    if cls in ("valu_pk", "valu_trans"):
        return 8
    if cls == "lds":
        if op.startswith("ds_write_b128") or op.startswith("ds_write2_b64"):
            return 13
        if op.startswith("ds_write_b64") or op.startswith("ds_write2_b32"):
            return 6
        if op.startswith("ds_write"):
            return 4
        if op.startswith("ds_read_b128") or op.startswith("ds_read2"):
            return 4
        return 2
    if cls == "vmem":
        return 4
    return 0


def assembly(src):
    path = src if os.path.isabs(src) else os.path.join(B.CSRC, os.path.basename(src))
    cmd = [B.HIPCC] + B.FLAGS + B.PER_FILE_FLAGS.get(os.path.basename(path), []) + ["-S", "--cuda-device-only", path, "-o", "-"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    return r.stdout.splitlines()


def kernels(lines):
    out, cur, name = {}, None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
        elif cur is not None:
            cur.append(ln)
            if ln.strip() == "s_endpgm":
                cur = None
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {d.replace("(anonymous namespace)::", "").replace("void ", ""): out[n] for n, d in zip(names, dem)}


def blocks_of_loop(body, depth):
    """[(label, in_loop_of_depth, [instructions])]: a block belongs to the loop when its label comment says `Depth=<depth>` (header or member)."""
    blocks, cur, label, inl = [], [], "entry", False
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            blocks.append((label, inl, cur))
            label, cur = m.group(1), []
            c = m.group(2) or ""
            inl = ("Depth=%d" % depth) in c
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            if "Depth=%d" % depth in s and "Loop" in s and not cur:      # a continuation comment line of the label
                inl = True
            continue
        cur.append(s.split(";")[0].strip())
    blocks.append((label, inl, cur))
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("kernel")
    ap.add_argument("--depth", type=int, default=2)
    a = ap.parse_args()
    ks = kernels(assembly(a.src))
    hits = [k for k in ks if re.search(a.kernel, k)]
    if not hits:
        raise SystemExit("no kernel matches %r; have:\n  %s" % (a.kernel, "\n  ".join(ks)))
    for k in hits:
        print("== %s" % k.split("(")[0])
        tot = collections.Counter()
        print("%-10s %5s %5s %5s %6s %4s %5s %5s %5s %5s | %6s  LDS detail" % ("block", "mfma", "valu", "pk", "trans", "lds", "vmem", "salu", "wait", "bar", "cycles"))
        for label, inl, ins in blocks_of_loop(ks[k], a.depth):
            if not inl or not ins:
                continue
            c, cyc, lds = collections.Counter(), collections.Counter(), collections.Counter()
            for i in ins:
                op = i.split()[0]
                cls = classify(op)
                c[cls] += 1
                cyc[cls] += price(op, cls)
                if cls == "lds":
                    lds[op] += 1
            tot.update(c)
            vec = cyc["valu"] + cyc["valu_pk"] + cyc["valu_trans"]
            print("%-10s %5d %5d %5d %6d %4d %5d %5d %5d %5d | mfma %5d vector %5d lds %4d vmem %3d  %s"
                  % (label, c["mfma"], c["valu"], c["valu_pk"], c["valu_trans"], c["lds"], c["vmem"], c["salu"], c["wait"], c["barrier"],
                     cyc["mfma"], vec, cyc["lds"], cyc["vmem"], " ".join("%s:%d" % kv for kv in sorted(lds.items()))))
        print("all blocks of the loop (every path once): %s" % dict(tot))


if __name__ == "__main__":
    main()
