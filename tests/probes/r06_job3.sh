#!/bin/bash
# round 6, job 3: hand-off tests + budget sweep, line-search group test, bitwise of the pinned pair roundings (default vs all-off variant), suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_handoff.py tests/test_gpu_line_search_batch.py -q -x 2>&1 | tail -15 > $O/handoff_test.txt; cat $O/handoff_test.txt
timeout 600 python tests/probes/r06_handoff.py C2 4096 0.3 > $O/handoff_c2.txt 2>&1; cat $O/handoff_c2.txt
timeout 600 python tests/probes/r06_handoff.py C5 4096 0.3 > $O/handoff_c5.txt 2>&1; cat $O/handoff_c5.txt
for v in r6_off; do echo "bitwise default vs $v:"; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | tail -6; done > $O/bitwise.txt 2>&1; cat $O/bitwise.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/gputest.txt; tail -3 $O/gputest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
