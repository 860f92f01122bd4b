# same-box comparison of three source trees (round-3 final, mid round 4, current): bench.py of each tree, alternating
R=$GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.4g" % d["value"], "%.3f ms" % d["roofline"]["kernel_ms_avg"])'
for round in 1 2; do
for cfg in "--config C3 --steps 20 --warmup 3" "--config C5 --mpc-steps 200 --steps 3 --warmup 1" "--steps 20 --warmup 3"; do
  for t in tests/probes/_r03_tree tests/probes/_mid_tree .; do
    extra=""; [ "$t" != "tests/probes/_r03_tree" ] && extra="--refine-steps 0"
    (cd $R/$t && python bench.py $cfg $extra --no-cpu-baseline --no-pmc 2>/dev/null | python -c "$P" "$t $cfg")
  done
done
done
