#!/bin/bash
# GPU box: same-box A/B of the speculative-reciprocal / max-tree pivot (default) against -DALG_GJSPEC=0, bitwise check, phase profiles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/nogjspec.so > $O/r04_gjspec_bitwise.txt 2>&1
bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" nogjspec > $O/r04_ab_gjspec_c3.txt 2>&1
bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" nogjspec > $O/r04_ab_gjspec_c2.txt 2>&1
bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" nogjspec > $O/r04_ab_gjspec_c5.txt 2>&1
bash tests/probes/phase_prof.sh run C3 1024 2 > $O/r04_phase_c3_team2.txt 2>&1
bash tests/probes/phase_prof.sh run C5 64 4 > $O/r04_phase_c5_team4.txt 2>&1
bash tests/probes/phase_prof.sh run C2 4096 1 > $O/r04_phase_c2.txt 2>&1
cat $O/r04_gjspec_bitwise.txt $O/r04_ab_gjspec_c3.txt $O/r04_ab_gjspec_c2.txt $O/r04_ab_gjspec_c5.txt
tail -14 $O/r04_phase_c3_team2.txt; tail -14 $O/r04_phase_c5_team4.txt
