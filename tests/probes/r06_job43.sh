#!/bin/bash
# round 6, job 43: the final sources (unicycle lane roles behind their switch, off; group pass of the line search restructured into row lambdas): bit-identity against
# the build before the restructure (variant nolru), the whole GPU suite, smoke, the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job43; O=gpurun_out/r06_job43
python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/nolru.so 2>&1 | tail -6 | tee $O/bitwise_final_vs_nolru.txt
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega\|^seed " | tail -12 > $O/gputest_final.txt; tail -3 $O/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -c 400 | tee $O/bench_nopmc.json
