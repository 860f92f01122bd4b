import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import algames_jl_amd
from algames_jl_amd import _abi
if os.environ.get("DROP_MPC"): _abi.SIGNATURES.pop("mpc_solve", None)
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "intro_time.py")).read())
