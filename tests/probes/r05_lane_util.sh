#!/bin/bash
# Lane utilisation of the step-wise kernels at a BASELINE config, run on the GPU box:
#   bash tests/probes/r05_lane_util.sh C2 4096 OUTDIR [lib.so]
# SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = mean active lanes (exec mask) per VALU instruction, per kernel; for the phases of
# k_direction the same pass is repeated on libraries built with -DALG_DIR_STOP=n (tests/probes/r05_dirstop_build.sh) and the
# consecutive differences are reported by tests/probes/r05_lane_util.py.
R=$GRAFT_REPO_ROOT; CFG=$1; G=$2; O=$3; LIB=$4
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
[ -n "$LIB" ] && export ALGAMES_HIP_LIB=$LIB
TAG=$(basename ${LIB:-default} .so)
rm -rf /tmp/lu_$TAG
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d /tmp/lu_$TAG -o p -- python $R/tests/probes/phase_times.py $CFG $G > $O/lane_util_${CFG}_${G}_$TAG.log 2>&1
python $R/tests/probes/r05_lane_util.py /tmp/lu_$TAG > $O/lane_util_${CFG}_${G}_$TAG.txt 2>&1
cat $O/lane_util_${CFG}_${G}_$TAG.txt
