#!/bin/bash
# Effective shader clock and power while the C2 solve runs (is a cycle saving returned as a lower clock?): rocm-smi samples during a long
# bench run, and GRBM_GUI_ACTIVE / kernel time under rocprofv3, for the default library and the variants given.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_clock; mkdir -p $O; cd $R
for v in default "$@"; do
  if [ $v = default ]; then unset ALGAMES_HIP_LIB; else export ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-pmc --steps 1500 --warmup 5 > $O/long_$v.json 2>/dev/null &
  BP=$!
  sleep 4
  for s in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo; sleep 0.4; done > $O/smi_$v.txt
  wait $BP
  python -c "import json; d=json.load(open('$O/long_$v.json')); print('$v', d['value'], d['roofline']['kernel_ms_avg'])"
  cat $O/smi_$v.txt
  (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ck_$v; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/ck_$v -o p -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 5 --warmup 2 > /dev/null 2>&1
   python - <<PY
import csv,glob,collections
d="/tmp/ck_$v"
kt=[x for x in glob.glob(d+"/**/*kernel_trace.csv",recursive=True)][0]
dur=collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    if "k_newton_solve" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
cc=[x for x in glob.glob(d+"/**/*counter_collection.csv",recursive=True)][0]
acc=collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "k_newton_solve" in r["Kernel_Name"]: acc[r["Dispatch_Id"]][r["Counter_Name"]]=acc[r["Dispatch_Id"]].get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k,v in acc.items():
    if k in dur: print("$v dispatch",k,"ns",dur[k],{c:x for c,x in v.items()},"GUI_ACTIVE/ns = %.3f GHz"%(v.get("GRBM_GUI_ACTIVE",0)/dur[k]))
PY
  ) 2>&1 | tail -8
done
