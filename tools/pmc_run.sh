#!/bin/bash
# rocprofv3 --pmc passes of one probe command on the GPU box; prints the last dispatch's counters per kernel.
# usage: tools/pmc_run.sh <tag> "<counters pass 1>" "<counters pass 2>" -- python3 tools/... args
set -e
tag=$1; p1=$2; p2=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in 1 2; do
  ctr=$p1; [ $pass = 2 ] && ctr=$p2
  [ -z "$ctr" ] && continue
  out=/tmp/pmc_${tag}_$pass; rm -rf $out
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- "$@" > /dev/null 2> /tmp/pmc_${tag}_$pass.err || { tail -5 /tmp/pmc_${tag}_$pass.err; exit 1; }
  python3 - $out <<'PY' >> $root/gpurun_out/pmc_$tag.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
last = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    if k.startswith("at::") or "elementwise" in k: continue
    k = k.split("(")[0]
    last.setdefault(k, {})[r["Counter_Name"]] = (r["Dispatch_Id"], float(r["Counter_Value"]))
for k, v in last.items():
    print(k, {c: "%.4g" % x[1] for c, x in v.items()})
PY
  rm -rf $out
done
