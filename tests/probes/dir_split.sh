#!/bin/bash
# time k_record / k_direction (full, stop after backward, stop after forward) at a config and batch size
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" tests/probes/lib_stop1.so tests/probes/lib_stop2.so; do
  rm -rf /tmp/ds
  if [ -n "$lib" ]; then export ALGAMES_HIP_LIB=$R/$lib; else unset ALGAMES_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ds -o t -- python $R/tests/probes/phase_times.py $1 $2 > /tmp/ds.log 2>&1
  python - <<PY
import csv,glob
f=[x for x in glob.glob("/tmp/ds/**/*kernel_stats.csv",recursive=True)][0]
out=[]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("void k_") and ("k_direction" in r["Name"] or "k_record" in r["Name"] or "k_line" in r["Name"] or "k_update" in r["Name"]): out.append("%s %.1f" % (r["Name"][5:16], float(r["AverageNs"])/1e3))
print("lib=$lib", "$1 $2:", "; ".join(out))
PY
done
