#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" $R/tests/probes/lib_nogj.so; do
  rm -rf /tmp/ds
  if [ -n "$lib" ]; then export ALGAMES_HIP_LIB=$lib; else unset ALGAMES_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ds -o t -- python $R/tests/probes/phase_times.py $1 $2 > /tmp/ds.log 2>&1
  python - <<PY
import csv,glob
f=[x for x in glob.glob("/tmp/ds/**/*kernel_stats.csv",recursive=True)][0]
for r in csv.DictReader(open(f)):
    if "k_direction" in r["Name"] or "k_newton_solve" in r["Name"]: print("lib=$lib", "$1 $2", r["Name"][5:20], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
