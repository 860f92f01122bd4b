#!/bin/bash
# Counter passes over the fused C2 solve (bench.py, 3 launches): what the waves wait for.  usage (GPU box): r05_pmc.sh OUTDIR [bench args]
R=$GRAFT_REPO_ROOT; O=$R/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES" \
 "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU" \
 "SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" \
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL GRBM_GUI_ACTIVE" ; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm_$i -o p -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 2 --warmup 1 "$@" > $O/pmc_$i.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
out = open(sys.argv[1] + "/pmc_summary.txt", "w")
for i in range(1, 7):
    fs = [x for x in glob.glob("/tmp/pm_%d/**/*counter_collection.csv" % i, recursive=True)]
    if not fs: print("pass", i, "no output", file=out); continue
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(fs[0])):
        if "k_newton_solve" in r["Kernel_Name"] or "k_mpc_loop" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    print("pass %d (%d launches), per launch:" % (i, n), " ".join("%s=%.5g" % (k, v / n) for k, v in sorted(acc.items())), file=out)
out.close(); print(open(sys.argv[1] + "/pmc_summary.txt").read())
PY
