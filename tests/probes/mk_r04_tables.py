"""Markdown tables of DESIGN.md section 6 / profiles/README.md from the committed round-4 evidence (profiles/r04_*_pmc.json,
profiles/r04_strong_shares.json).   usage: python tests/probes/mk_r04_tables.py"""
import json, os, glob, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = [(RND + "_c2", "C2, 4096 games, one wavefront per game"), (RND + "_c4", "C4 shard, 8192 games, one wavefront per game"),
        (RND + "_c3", "C3, 1024 games, team of two"), (RND + "_c5mpc", "C5 loop, 64 seeds x 200 steps, team of four")]
print("| workload (kernel shape = the automatic choice) | game-iterations/s | kernel ms (rocprof / HIP events) | own-bytes frac of 8 TB/s | fabric traffic / model | VALU issue | f64 MFMA of 78.6 TF | wave issuing / waiting | VALU / SALU / LDS / VMEM per game-iteration | corrections per launch |")
print("|---|---|---|---|---|---|---|---|---|---|")
for tag, name in rows:
    f = os.path.join(root, "profiles", tag + "_pmc.json")
    if not os.path.exists(f): continue
    p = json.load(open(f)); b = p["bench"]; ins = p["insts_per_game_iter"]
    own = b["roofline"]["bytes_per_game_iter"]
    print("| %s | %.2f M | %.2f / %.2f | %.3f | %.0f KB / %.0f KB = %.2f | %.3f | %.3f | %.2f / %.2f | %.1f K / %.1f K / %.1f K / %.1f K | %d |" % (
        name, b["value"] / 1e6, p["kernel_avg_ms_rocprof"], p["kernel_ms_bench_hip_events"], p["own_hbm_frac"],
        p["hbm_bytes_per_game_iter"] / 1e3, own / 1e3, p["traffic_over_model"], p["valu_issue_frac"], p["mfma_frac"], p["wave_issue_frac"], p["wave_wait_frac"],
        ins["valu"] / 1e3, ins["salu"] / 1e3, ins["lds"] / 1e3, ins["vmem"] / 1e3, b["config"]["direction_refinement"]["correction_solves_rank0"]))
f = os.path.join(root, "profiles", RND + "_strong_shares.json")
if os.path.exists(f):
    s = json.load(open(f))
    print("\n| per-GPU share of a 4096-game job | GPUs | game-iterations/s of ONE GPU at that share | wavefronts per game | predicted job rate without any scaling loss |")
    print("|---|---|---|---|---|")
    for r in s["shares"]:
        print("| %d | %d | %.2f M | %d | %.1f M |" % (r["games_per_gpu"], 4096 // r["games_per_gpu"], r["value"] / 1e6, r["wavefronts_per_game"], r["value"] * (4096 // r["games_per_gpu"]) / 1e6))
