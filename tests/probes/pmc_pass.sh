#!/bin/bash
# usage: pmc_pass.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- [bench args]
# Runs bench.py under rocprofv3 --pmc once per counter group (own run each, with --kernel-trace only) and prints per-launch
# averages of every counter for the solver kernel.  gpurun_out/<tag>/pmc_<i>/
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
CGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do CGS+=("$1"); shift; done
[ $# -gt 0 ] && shift
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline $*"
i=0
for g in "${CGS[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/bench_$i.json 2> $OUT/err_$i.log || tail -5 $OUT/err_$i.log
  i=$((i+1))
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]; res = {}
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "k_newton_solve" not in r["Kernel_Name"] and "k_mpc_loop" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k, v in acc.items(): res[k] = v / max(1, len(disp[k]))
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
PY
