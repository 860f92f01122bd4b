#!/bin/bash
# Round-4 evidence run on the GPU box (gpurun): kernel traces + PMC passes of the four BASELINE shapes (tests/probes/prof_r02.sh), the
# single-GPU numbers of the strong-scaling split's per-GPU shares, and the default bench line.  Summaries: tests/probes/mk_profile_r02.py.
R=$GRAFT_REPO_ROOT
bash $R/tests/probes/prof_r02.sh r04_c2
bash $R/tests/probes/prof_r02.sh r04_c4 --config C4
bash $R/tests/probes/prof_r02.sh r04_c3 --config C3
bash $R/tests/probes/prof_r02.sh r04_c5mpc --config C5 --mpc-steps 200
cd $R
for g in 4096 2048 1024 512; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --games-per-gpu $g > gpurun_out/r04_share_$g.json 2>/dev/null
done
python bench.py > gpurun_out/bench_r04_default.json 2> gpurun_out/bench_r04_default.err
tail -c 600 gpurun_out/bench_r04_default.json
# keep the merge-back small: the raw per-dispatch counter dumps are large
find gpurun_out/r04_* -name "*agent_info*" -delete 2>/dev/null
du -sh gpurun_out/r04_* | tail -5
