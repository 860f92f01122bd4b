import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import algames_jl_amd as alg
ids=np.arange(40,56)
pg = alg.scenarios.make_problem("C2", ids, N=12)
lib=alg.hip_lib().dll
lib.alg_debug_check_guards.argtypes=[ctypes.c_void_p]
for rep in range(1):
    alg.newton_solve(pg)
    print('guards bad:', lib.alg_debug_check_guards(pg.batch.h), pg.stats.summary['outer_iters'][:5])
