#!/bin/bash
# per-kernel dynamic instruction counts of the stepwise API (run on the GPU box): tests/probes/phase_pmc.sh C2 4096
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python $R/tests/probes/phase_times.py $1 $2 > /tmp/pt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d /tmp/pp -o p -- python $R/tests/probes/phase_times.py $1 $2 > /tmp/pp.log 2>&1
python - <<PY
import csv, glob, collections
f=[x for x in glob.glob("/tmp/pt/**/*.csv",recursive=True) if "kernel_stats" in x][0]
print("kernel avg us:")
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("void k_") : print("  %-40s calls %s avg %.1f us" % (r["Name"][5:45], r["Calls"], float(r["AverageNs"])/1e3))
f=[x for x in glob.glob("/tmp/pp/**/*.csv",recursive=True) if "counter_collection" in x][0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(k,r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k]+=1
for k in acc:
    if not k.startswith("void k_"): continue
    w=acc[k]["SQ_WAVES"]/cnt[k]
    print(k[5:45], "launches", cnt[k], "waves/launch %.0f"%w)
    print("   per wave:", {c: round(v/cnt[k]/max(w,1)) for c,v in acc[k].items() if c!="SQ_WAVES"})
PY
