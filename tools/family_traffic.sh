#!/bin/bash
# HBM bytes per launch (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of several kernels of the fp32 step in one go, plus their
# durations from the same trace: tools/family_traffic.sh <out.json> <kernel name> [<kernel name> ...]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_fam_$ctr; rm -rf $d
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o p -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs > /dev/null 2> /tmp/pmc_fam_$ctr.err
done
python3 - "$out" "$@" <<'PY'
import csv, glob, json, sys
out, kernels = sys.argv[1], sys.argv[2:]
def table(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    return rows
fetch, write = table("/tmp/pmc_fam_FETCH_SIZE", "FETCH_SIZE"), table("/tmp/pmc_fam_WRITE_SIZE", "WRITE_SIZE")
tr = glob.glob("/tmp/pmc_fam_FETCH_SIZE/**/*kernel_trace.csv", recursive=True)
dur = {}
if tr:
    for r in csv.DictReader(open(tr[0])):
        dur.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k in kernels:
    fv = [float(r["Counter_Value"]) for r in fetch if k in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    wv = [float(r["Counter_Value"]) for r in write if k in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    dv = [x for name, v in dur.items() if k in name for x in v]
    if not fv: continue
    fb, wb = 2.0 * 1024.0 * sum(fv), 1024.0 * sum(wv)     # gfx950: FETCH_SIZE reports half of wide coalesced reads; KiB -> bytes
    res[k] = {"launches": len(fv), "fetch_bytes_total_corrected": fb, "write_bytes_total": wb,
              "hbm_bytes_per_launch": (fb + wb) / len(fv),
              "duration_ms_total_under_counters": sum(dv) / 1e6 if dv else None,
              "hbm_TB_per_s": (fb + wb) / (sum(dv) * 1e-9) / 1e12 if dv else None}
steps = sum(len(v) for name, v in dur.items() if "adam_kernel" in name)      # one Adam launch per training step of this configuration
json.dump({"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 "
                     "--no-kernel-events (warm-up + timed + the three host-enqueue steps: every step of the process is counted, "
                     "steps_counted = launches of adam_kernel); FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; durations from the FETCH_SIZE pass",
           "steps_counted": steps, "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf /tmp/pmc_fam_FETCH_SIZE /tmp/pmc_fam_WRITE_SIZE
