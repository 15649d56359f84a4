"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py` into per-kernel totals for the LAST training step
(steps are delimited by the Adam launches).  Usage: python tools/summarize_trace.py <kernel_trace.csv> > profiles/xx.md"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
    groups = []
    for i in adam:
        if groups and i - groups[-1][-1] < 20:
            groups[-1].append(i)
        else:
            groups.append([i])
    seg = ev[groups[-2][-1] + 1:groups[-1][-1] + 1]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(b - a for a, b, _ in seg) / 1e6
    by, cnt = collections.Counter(), collections.Counter()
    for a, b, n in seg:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0] if not k.startswith("at::") else k[:70]
        by[k] += (b - a) / 1e6
        cnt[k] += 1
    print("# rocprofv3 --kernel-trace, last training step of `bench.py` (%s)\n" % (sys.argv[2] if len(sys.argv) > 2 else "B=32, 256x256, fp32, rec+KL only"))
    print("step wall %.1f ms | kernel busy %.1f ms | idle %.1f ms | %d launches\n" % (wall, busy, wall - busy, len(seg)))
    print("| kernel | launches | total ms | avg us | % of busy |\n|---|---:|---:|---:|---:|")
    for k, v in by.most_common(40):
        print("| `%s` | %d | %.2f | %.1f | %.1f |" % (k, cnt[k], v, 1e3 * v / cnt[k], 100 * v / busy))
    # where the idle time sits: the largest gaps between the end of one kernel and the start of the next
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, short(seg[i][2]), short(seg[i + 1][2])) for i in range(len(seg) - 1))
    big = [g for g in gaps if g[0] > 20.0]
    print("\ngaps between consecutive kernels: %d over 20 us, %.2f ms in all; the ten largest:\n" % (len(big), sum(g[0] for g in big) / 1e3))
    print("| gap us | after | before |\n|---:|---|---|")
    for g in gaps[-10:][::-1]:
        print("| %.0f | `%s` | `%s` |" % g)


if __name__ == "__main__":
    main(sys.argv[1])
