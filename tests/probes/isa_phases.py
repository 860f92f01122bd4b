"""Static per-phase instruction counts of the Newton direction (tile path): compiles k_direction with -DALG_ISA_MARK and counts
the instructions between the ALGMARK comments.  usage: python tests/probes/isa_phases.py [model p d ext] [extra flags]"""
import re, subprocess, sys, os, collections
cfg = sys.argv[1:5] if len(sys.argv) > 4 else ["ALG_MODEL_DOUBLE_INTEGRATOR", "3", "2", "0"]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs("/tmp/isa", exist_ok=True)
src = "/tmp/isa/ph.hip"
open(src, "w").write('#include "%s/algames.jl_amd/csrc/algames_kernels.hpp"\ntemplate __global__ void k_direction<Cfg<%s>>(Params, double, int*);\n' % (root, ", ".join(cfg)))
out = "/tmp/isa/ph.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-invalid-offsetof", "-DALG_ISA_MARK",
                       "-mllvm", "-disable-machine-licm", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-o", out, src] + sys.argv[5:])
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem"
    return "other"
cur = "pre"; acc = collections.OrderedDict()
for l in open(out):
    m = re.search(r"; ALGMARK (\d+)", l)
    if m: cur = "->" + m.group(1); continue
    m = re.match(r"\t([a-z_0-9]+)", l)
    if m and not l.startswith("\t."):
        acc.setdefault(cur, collections.Counter())[cat(m.group(1))] += 1
names = {"->11": "setup(end)->0: value recursion", "->0": "Q-add", "->1": "V,y,A' table", "->2": "g", "->3": "column build", "->4": "GJ",
         "->5": "closed loop", "->9": "coef/rec copy", "->10": "gains out (+loop)", "->6": "(after bwd step mark 6) fwd ...", "->7": "costate", "->8": "tail"}
for k, c in acc.items():
    print("%-8s %-36s %5d  %s" % (k, names.get(k, ""), sum(c.values()), dict(c)))
