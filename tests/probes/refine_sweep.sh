# cost of the refinement gate and of the correction solves: interleaved A/B runs of bench.py on one box
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.3f M/s" % (d["value"]/1e6), d["config"]["direction_refinement"]["correction_solves_rank0"])'
for rep in 1 2 3; do
for cfg in "--config C2" "--config C3" "--config C5 --mpc-steps 100 --steps 4"; do
  for rt in "--refine-steps 0" "--refine-steps 2 --refine-tol 1e300" "--refine-steps 2"; do
    $B $cfg $rt 2>/dev/null | python -c "$P" "$cfg $rt"
  done
done
done
