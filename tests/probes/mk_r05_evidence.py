"""gpurun_out/ of tests/probes/r05_measure.sh -> profiles/r05_* (kernel stats, counter summaries, strong-scaling shares, other shapes, default
bench line), then the tables of DESIGN.md section 6 (tests/probes/mk_r04_tables.py r05).   usage: python tests/probes/mk_r05_evidence.py"""
import json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
here = os.path.dirname(os.path.abspath(__file__))
env = dict(os.environ, PROF_STEPS="10", PROF_WARMUP="8")
for tag, args in (("r05_c2", []), ("r05_c4", ["--config", "C4"]), ("r05_c3", ["--config", "C3"])):
    subprocess.check_call([sys.executable, os.path.join(here, "mk_profile_r02.py"), tag] + args, env=env)
subprocess.check_call([sys.executable, os.path.join(here, "mk_profile_r02.py"), "r05_c5mpc", "--config", "C5", "--mpc-steps", "200"], env=dict(os.environ, PROF_STEPS="3", PROF_WARMUP="1"))
shares = []
for g in (4096, 2048, 1024, 512):
    d = json.loads(open(os.path.join(root, "gpurun_out", "r05_share_%d.json" % g)).read().strip().split("\n")[-1])
    shares.append({"games_per_gpu": g, "value": d["value"], "ms_per_step": d["ms_per_step"], "wavefronts_per_game": d["config"]["wavefronts_per_game"]})
json.dump({"note": "single-GPU rates of the per-GPU shares of a 4096-game strong-scaling job (bench.py --steps 20 --warmup 8 --games-per-gpu G); no multi-GPU hardware run exists", "shares": shares},
          open(os.path.join(root, "profiles", "r05_strong_shares.json"), "w"), indent=1)
shutil.copy(os.path.join(root, "gpurun_out", "bench_r05_default.json"), os.path.join(root, "profiles", "r05_bench_default.json"))
shutil.copy(os.path.join(root, "gpurun_out", "r05_other_shapes.txt"), os.path.join(root, "profiles", "r05_other_shapes.txt"))
shutil.copy(os.path.join(root, "gpurun_out", "r05_pmc_final", "pmc_summary.txt"), os.path.join(root, "profiles", "r05_pmc_c2_final_summary.txt"))
subprocess.check_call([sys.executable, os.path.join(here, "mk_r04_tables.py"), "r05"])
