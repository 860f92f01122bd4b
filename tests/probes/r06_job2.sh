#!/bin/bash
# round 6, job 2: batched chunk staging / phase-A load hoisting / dual update inside the record pass: bitwise + same-box A/B, then the GPU suite on the new default
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job2; mkdir -p $O
for v in r6_off r6_stage r6_stage_pa; do echo "bitwise default vs $v:"; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | tail -6; done > $O/bitwise.txt 2>&1; cat $O/bitwise.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_off r6_stage r6_stage_pa r6_sb4 r6_sb8 > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
bash tests/probes/ab.sh "--steps 10 --warmup 4 --games-per-gpu 16384" r6_off r6_stage_pa > $O/ab_c2_16k.txt 2>&1; cat $O/ab_c2_16k.txt
bash tests/probes/ab.sh "--config C5 --games-per-gpu 4096 --steps 10 --warmup 4" r6_off r6_stage r6_stage_pa > $O/ab_c5_4096.txt 2>&1; cat $O/ab_c5_4096.txt
bash tests/probes/ab.sh "--config C3 --games-per-gpu 4096 --steps 10 --warmup 4" r6_off r6_stage r6_stage_pa > $O/ab_c3_4096.txt 2>&1; cat $O/ab_c3_4096.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/gputest.txt; tail -3 $O/gputest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
