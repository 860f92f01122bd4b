"""gpurun_out/<tag>/ (tests/probes/prof_r02.sh) -> profiles/<tag>_kernel_stats.csv + profiles/<tag>_pmc.json (read by bench.py)."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", tag)
shutil.copy(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1])
kern = bench["roofline"]["kernel"]
cnt = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k, v in acc.items(): cnt[k] = v / len(disp[k])
kname, kavg_ns = None, None
for r in csv.DictReader(open(os.path.join(root, "profiles", tag + "_kernel_stats.csv"))):
    if kern in r["Name"]:
        kname, kavg_ns, kcalls = r["Name"], float(r["AverageNs"]), int(r["Calls"])
it = bench["roofline"]["game_iters_per_launch"]; own = bench["roofline"]["bytes_per_game_iter"]
hbm = 1024.0 * (2.0 * cnt["FETCH_SIZE"] + cnt["WRITE_SIZE"])
simd_quads = cnt["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0          # 8 XCDs report their cycles; 1024 SIMDs; quad-cycle = 4 clocks
out = {
    "command": "python bench.py --steps %s --warmup %s --no-cpu-baseline " % (os.environ.get("PROF_STEPS", "5"), os.environ.get("PROF_WARMUP", "2")) + " ".join(sys.argv[2:]),
    "config": bench["config"]["name"], "games_per_gpu": bench["config"]["games_per_gpu"], "mpc_steps": bench["config"]["mpc_steps"],
    "kernel": kname, "kernel_avg_ms_rocprof": kavg_ns * 1e-6, "kernel_calls": kcalls, "kernel_ms_bench_hip_events": bench["roofline"]["kernel_ms_avg"],
    "game_iters_per_launch": it,
    "hbm_bytes_per_launch": hbm, "hbm_bytes_per_game_iter": hbm / it, "traffic_over_model": hbm / (own * it),
    "valu_issue_frac": cnt["SQ_ACTIVE_INST_VALU"] / simd_quads,
    "wave_issue_frac": cnt["SQ_ACTIVE_INST_ANY"] / cnt["SQ_WAVE_CYCLES"],
    "wave_wait_frac": cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"],
    "mfma_frac": cnt["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / (kavg_ns * 1e-9) / 78.6e12,
    "own_hbm_frac": own * it / (kavg_ns * 1e-9) / 8.0e12,
    "insts_per_game_iter": {k[9:].lower(): cnt[k] / it for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA")},
    "note": "separate rocprofv3 --pmc passes (tests/probes/prof_r02.sh), per-launch averages for the named kernel; FETCH_SIZE doubled (gfx950 correction, "
            "calibrated with tests/probes/pmc_calib.hip), WRITE_SIZE as is, KB units; valu_issue_frac = SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 XCDs / 4 x 1024 SIMDs): "
            "share of all SIMD issue quad-cycles of the launch that issued a VALU / MFMA instruction; mfma_frac = f64 MFMA flops / time / 78.6 TF; "
            "own_hbm_frac = bench.py structured_bytes() x game-iterations / rocprof kernel time / 8 TB/s",
    "counters_per_launch": cnt, "bench": bench,
}
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
for k in ("kernel", "kernel_avg_ms_rocprof", "kernel_ms_bench_hip_events", "hbm_bytes_per_game_iter", "traffic_over_model", "valu_issue_frac", "wave_issue_frac", "wave_wait_frac", "mfma_frac", "own_hbm_frac", "insts_per_game_iter"):
    print(k, out[k])
