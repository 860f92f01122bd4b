"""Generates tests/golden/oracle_solutions.npz: final trajectories / multipliers / statistics of the CPU oracle
(oracle/, pinned to the reference's known-answer values by tests/test_oracle_kat.py) on a few seeded scenarios.

    python tests/golden/make_golden.py

The reference itself is Julia and cannot run in the build container (no Julia toolchain), so these vectors are oracle
outputs, not reference outputs: they guard the oracle against drift (tests/test_oracle_kat.py::test_golden_solutions_oracle)
and give the GPU tests a committed target that does not depend on rebuilding the oracle
(tests/test_gpu_parity.py::test_golden_solutions_gpu)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {                      # name -> (scenario set, global scenario ids, make_problem kwargs)
    "c2_n12": ("C2", np.arange(40, 48), dict(N=12)),
    "c2_n40": ("C2", np.arange(4090, 4094), {}),
    "c5": ("C5", np.arange(300, 304), {}),
    "c3_n20": ("C3", np.arange(8, 10), dict(N=20)),
    "q2": ("Q", np.arange(16, 20), dict(p=2)),              # quadrotors (scenarios.quadrotor_crossing): dense Newton direction
    "q3_n10": ("Q", np.arange(5, 7), dict(p=3, N=10)),
}


def build(name, alg, backend):
    if name == "intro":
        from test_oracle_kat import _intro_problem
        return _intro_problem(alg, backend)
    cfg, ids, kw = CASES[name]
    if cfg == "C3" and "N" in kw:       # scenarios.make_problem fixes N for C3/C5: call the builder directly
        from algames_jl_amd import scenarios as sc, host
        model, N, dt, x0, obj, con, opts = sc.c3_unicycle(ids, N=kw["N"], p=4)
        return host.GameProblem(N, dt, x0, model, opts, obj, con, backend=backend, game_id0=int(ids[0]))
    return alg.scenarios.make_problem(cfg, ids, backend=backend, **kw)


def solve(name, alg, backend):
    prob = build(name, alg, backend)
    alg.newton_solve(prob)
    lam, mu = prob.batch.get_con_duals()
    s = prob.stats.summary
    return dict(z=prob.batch.get_traj(), lam=lam, mu=mu, newton_iters=s["newton_iters"], outer_iters=s["outer_iters"],
                status=s["status"], converged=s["converged"], res=s["last"]["res"], opt_vio=s["last"]["opt_vio"])


if __name__ == "__main__":
    import algames_jl_amd as alg
    import oracle as orc
    orc.build()
    out = {}
    for name in list(CASES) + ["intro"]:
        for k, v in solve(name, alg, orc.lib()).items():
            out[f"{name}.{k}"] = v
    path = os.path.join(HERE, "oracle_solutions.npz")
    if os.path.exists(path):        # committed vectors stay as they are (a rebuilt oracle agrees with them to rounding); new cases are added
        old = np.load(path)
        out = {**out, **{k: old[k] for k in old.files}}
    np.savez_compressed(path, **out)
    print("wrote", os.path.join(HERE, "oracle_solutions.npz"), {k: v.shape for k, v in out.items() if k.endswith(".z")})
