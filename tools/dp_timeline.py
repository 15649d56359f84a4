"""Where the gradient buckets' collectives sit inside the backward pass: from a rocprofv3 --kernel-trace CSV of
`bench.py --force-dist` (RCCL world size 1 on one GPU: the product's bucketed GradReducer runs exactly as with N ranks, the
collective kernels are real, only their peers are missing).  Prints, for the LAST training step: the step's span, the first and last
backward kernel, and per RCCL kernel its start relative to the step start / to the end of the backward, its duration, and what compute
kernel runs at that moment -- the timeline DESIGN.md 6 states for the N-rank run.
usage: python tools/dp_timeline.py <kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
    adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
    groups = []
    for i in adam:
        if groups and i - groups[-1][-1] < 20:
            groups[-1].append(i)
        else:
            groups.append([i])
    seg = ev[groups[-2][-1] + 1:groups[-1][-1] + 1]
    t0 = seg[0][0]
    is_coll = lambda n: ("nccl" in n.lower()) or ("rccl" in n.lower())
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    comp = [e for e in seg if not is_coll(e[2])]
    coll = [e for e in seg if is_coll(e[2])]
    # the backward starts with the first weight-gradient / data-gradient kernel after the loss kernels: take the first wgrad launch
    bwd0 = next((e for e in comp if "wgrad" in e[2]), comp[0])
    clip = next((e for e in comp if "grad_norm" in e[2] or "sumsq" in e[2]), comp[-1])
    print("step span %.2f ms, %d compute launches, %d collective launches; backward from %.2f ms (first weight-gradient kernel) to %.2f ms (clip / Adam)"
          % ((seg[-1][1] - t0) / 1e6, len(comp), len(coll), (bwd0[0] - t0) / 1e6, (clip[0] - t0) / 1e6))
    print("| # | collective kernel | queue | start ms | before the backward's end ms | duration us | compute kernel running at that moment |")
    print("|---|---|---|---:|---:|---:|---|")
    for i, (a, b, n, q) in enumerate(coll):
        under = next((short(c[2]) for c in comp if c[0] <= a < c[1]), "(none: the compute queue is idle)")
        print("| %d | `%s` | %s | %.2f | %.2f | %.1f | `%s` |" % (i, short(n), q, (a - t0) / 1e6, (clip[0] - a) / 1e6, (b - a) / 1e3, under))
    if coll:
        busy = sum(b - a for a, b, _, _ in coll) / 1e6
        tail = max(0.0, (coll[-1][1] - clip[0]) / 1e6)
        print("\ncollective kernels busy %.2f ms in all; %.2f ms of the last one lies past the start of the clip / Adam kernels (exposed)" % (busy, tail))


if __name__ == "__main__":
    main(sys.argv[1])
