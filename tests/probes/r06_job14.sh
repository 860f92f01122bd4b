#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job14; mkdir -p $O
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 4096 > $O/phase_cycles_c2_4096.txt 2>&1; cat $O/phase_cycles_c2_4096.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
