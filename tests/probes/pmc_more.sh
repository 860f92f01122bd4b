#!/bin/bash
# extra SQ counters for the fused kernel at C2 (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm$i -o p -- $BENCH > /tmp/pm$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("/tmp/pm*/**/*counter_collection.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        if "k_newton_solve" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    for c in set(r2 for r2 in acc): pass
    n[f]=len(seen)
    for r in csv.DictReader(open(f)):
        pass
per=45056.0
# every pass has the same number of dispatches (4): normalise by it
d=list(n.values())[0]
for k in sorted(acc): print("%-24s %12.0f per game-iteration" % (k, acc[k]/d/per))
PY
