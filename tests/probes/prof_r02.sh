#!/bin/bash
# usage: prof_r02.sh <tag> [bench args...]  -- on the GPU box (gpurun): kernel trace + separate PMC passes of the same bench command
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ST=${PROF_STEPS:-5}; WU=${PROF_WARMUP:-2}
BENCH="python $R/bench.py --steps $ST --warmup $WU --no-cpu-baseline --no-pmc $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/bench_pmc_$i.json 2> $OUT/pmc_$i.err
  i=$((i+1))
done
cd $R
python bench.py --steps $ST --warmup $WU --no-cpu-baseline --no-pmc $* > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json
