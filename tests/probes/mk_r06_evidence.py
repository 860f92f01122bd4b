"""gpurun_out/ of tests/probes/r06_measure.sh -> profiles/r06_* (kernel stats, counter summaries, strong-scaling shares, other shapes, heterogeneous
batches, default bench lines), then the table of DESIGN.md section 7 (tests/probes/mk_r04_tables.py r06).   usage: python tests/probes/mk_r06_evidence.py"""
import json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
here = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ, PROF_STEPS="10", PROF_WARMUP="8")
for tag, args in (("r06_c2", []), ("r06_c4", ["--config", "C4"]), ("r06_c3", ["--config", "C3"])):
    subprocess.check_call([sys.executable, os.path.join(here, "mk_profile_r02.py"), tag] + args, env=env)
subprocess.check_call([sys.executable, os.path.join(here, "mk_profile_r02.py"), "r06_c5mpc", "--config", "C5", "--mpc-steps", "200"], env=dict(os.environ, PROF_STEPS="3", PROF_WARMUP="1"))
shares = []
for g in (4096, 2048, 1024, 512):
    d = json.loads(open(os.path.join(root, "gpurun_out", "r06_share_%d.json" % g)).read().strip().split("\n")[-1])
    shares.append({"games_per_gpu": g, "value": d["value"], "ms_per_step": d["ms_per_step"], "wavefronts_per_game": d["config"]["wavefronts_per_game"]})
json.dump({"note": "single-GPU rates of the per-GPU shares of a 4096-game strong-scaling job (bench.py --steps 20 --warmup 8 --games-per-gpu G); no multi-GPU hardware run exists", "shares": shares},
          open(os.path.join(root, "profiles", "r06_strong_shares.json"), "w"), indent=1)
for src, dst in (("bench_r06_default.json", "r06_bench_default.json"), ("bench_r06_steps20_warmup5.json", "r06_bench_steps20_warmup5.json"), ("r06_other_shapes.txt", "r06_other_shapes.txt"),
                 ("r06_hetero.txt", "r06_hetero.txt"), (os.path.join("r06_pmc_final", "pmc_summary.txt"), "r06_pmc_c2_final_summary.txt")):
    shutil.copy(os.path.join(root, "gpurun_out", src), os.path.join(root, "profiles", dst))
subprocess.check_call([sys.executable, os.path.join(here, "mk_r04_tables.py"), "r06"])
