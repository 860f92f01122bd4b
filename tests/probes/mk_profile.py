"""Turn a gpurun_out/<tag>/ directory produced by tests/probes/prof_run.sh into the committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and profiles/<tag>_pmc_traffic.json."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
kname = {"C2": "k_newton_solve<Cfg<DI,3,2,0>>", "C3": "k_newton_solve<Cfg<UNI,4,2,0>>", "C5": "k_newton_solve<Cfg<UNI,3,2,0>>"}[cfg]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", tag)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
def counters(sub):
    f = glob.glob(os.path.join(src, sub, "*counter_collection.csv"))[0]
    acc = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if "k_newton_solve" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return {k: v / len(disp) for k, v in acc.items()}, len(disp)
fetch, nf = counters("pmc_fetch"); write, nw = counters("pmc_write"); sq, ns = counters("pmc_sq")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1])
out = {
    "command": "python bench.py --steps 5 --warmup 2 --no-cpu-baseline" + ("" if cfg == "C2" else " --config %s --games-per-gpu %d" % (cfg, bench["config"]["games_per_gpu"])), "config": cfg, "games_per_gpu": bench["config"]["games_per_gpu"],
    "kernel": kname, "launches_averaged": nf,
    "fetch_size_kb": fetch["FETCH_SIZE"], "write_size_kb": write["WRITE_SIZE"],
    "hbm_bytes_per_launch": 1024.0 * (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]),
    "note": "separate --pmc passes (tests/probes/prof_run.sh); KB units; FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE taken as is: calibrated on this box with tests/probes/pmc_calib.hip (2 GiB streams: FETCH_SIZE = 0.500 x bytes for 8 B/lane and 16 B/lane loads, WRITE_SIZE = 1.000 x bytes for 8 B/lane and 16 B/lane stores); these are L2-fabric-side bytes, Infinity-Cache hits included",
    "sq_per_launch": sq, "bench": bench,
}
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_traffic.json"), "w"), indent=1)
it = bench["config"]["newton_iters_per_solve_total"]
print("hbm bytes/launch %.3e = %.1f KB per game-iteration" % (out["hbm_bytes_per_launch"], out["hbm_bytes_per_launch"] / it / 1024))
print({k: round(v / it) for k, v in sq.items()})
