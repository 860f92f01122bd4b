"""Static instruction counts per source line of algames_device.hpp for one kernel instantiation (needs -gline-tables-only):
python tests/probes/line_profile.py [kernel] [model p d ext] -> instructions attributed to each source line inside the hottest loops."""
import re, subprocess, sys, os, collections
kern = sys.argv[1] if len(sys.argv) > 1 else "k_direction"
cfg = sys.argv[2:6] if len(sys.argv) > 5 else ["ALG_MODEL_DOUBLE_INTEGRATOR", "3", "2", "0"]
sigs = {"k_newton_solve": "(Params, int, uint64_t)", "k_direction": "(Params, double, int*)", "k_line_search": "(Params, double, const double*, double*, int*)", "k_record": "(Params, alg_record*)"}
hdr = os.environ.get("LP_HEADER", "algames_direction.hpp")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = "/tmp/isa/lp_%s.hip" % kern
os.makedirs("/tmp/isa", exist_ok=True)
open(src, "w").write('#include "%s/algames.jl_amd/csrc/algames_kernels.hpp"\ntemplate __global__ void %s<Cfg<%s>>%s;\n' % (root, kern, ", ".join(cfg), sigs[kern]))
out = src.replace(".hip", ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-invalid-offsetof", "-gline-tables-only",
                       "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-mllvm", "-disable-machine-licm", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-o", out, src])
files = {}; cur = None; cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    return "vmem"
for l in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)); continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(m.group(1), "?"), int(m.group(2))); continue
    m = re.match(r"\t([a-z_0-9]+)", l)
    if m and cur and cur[0].endswith(hdr):
        cnt[cur[1]] += 1; kinds[cur[1]][cat(m.group(1))] += 1
lo, hi = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 10 ** 9)
lines = open(os.path.join(root, "algames.jl_amd/csrc/" + hdr)).read().split("\n")
tot = 0
for ln in sorted(cnt):
    if lo <= ln <= hi and cnt[ln] >= 3:
        tot += cnt[ln]
        print("%5d %4d %-40s | %s" % (ln, cnt[ln], dict(kinds[ln]), lines[ln - 1].strip()[:90]))
print("total in range", sum(v for k, v in cnt.items() if lo <= k <= hi))
