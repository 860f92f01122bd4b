"""gpurun_out/ of tests/probes/r04_measure.sh -> profiles/r04_* (kernel stats, counter summaries, strong-scaling shares, default bench line),
then the tables of DESIGN.md section 6 (tests/probes/mk_r04_tables.py).   usage: python tests/probes/mk_r04_evidence.py"""
import json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
here = os.path.dirname(os.path.abspath(__file__))
for tag, args in (("r04_c2", []), ("r04_c4", ["--config", "C4"]), ("r04_c3", ["--config", "C3"]), ("r04_c5mpc", ["--config", "C5", "--mpc-steps", "200"])):
    subprocess.check_call([sys.executable, os.path.join(here, "mk_profile_r02.py"), tag] + args)
f = os.path.join(root, "profiles", "r04_strong_shares.json")
old = json.load(open(f))
shares = []
for g in (4096, 2048, 1024, 512):
    d = json.loads(open(os.path.join(root, "gpurun_out", "r04_share_%d.json" % g)).read().strip().split("\n")[-1])
    shares.append({"games_per_gpu": g, "value": d["value"], "ms_per_step": d["ms_per_step"], "wavefronts_per_game": d["config"]["wavefronts_per_game"]})
json.dump({"note": old["note"], "shares": shares}, open(f, "w"), indent=1)
shutil.copy(os.path.join(root, "gpurun_out", "bench_r04_default.json"), os.path.join(root, "profiles", "r04_bench_default.json"))
subprocess.check_call([sys.executable, os.path.join(here, "mk_r04_tables.py")])
