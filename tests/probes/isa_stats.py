"""Static ISA accounting of one kernel instantiation (default: k_newton_solve of C2).
usage: python tests/probes/isa_stats.py [kernel] [model p d ext]  -> registers / scratch and per-loop instruction mix."""
import re, subprocess, sys, os, collections
kern = sys.argv[1] if len(sys.argv) > 1 else "k_newton_solve"
cfg = sys.argv[2:6] if len(sys.argv) > 5 else ["ALG_MODEL_DOUBLE_INTEGRATOR", "3", "2", "0"]
sigs = {
 "k_newton_solve": "(Params, int, uint64_t)", "k_direction": "(Params, double, int*)",
 "k_record": "(Params, alg_record*)", "k_line_search": "(Params, double, const double*, double*, int*)",
 "k_newton_step": "(Params, int, int, alg_step_info*)",
 "k_ibr": "(Params, int, int, int, uint64_t, int, IbrOrder, double)",
 "k_mpc_loop": "(Params, int, uint64_t, double*)",
}
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs("/tmp/isa", exist_ok=True)
src = "/tmp/isa/one_%s.hip" % kern
open(src, "w").write('#include "%s/algames.jl_amd/csrc/algames_kernels.hpp"\ntemplate __global__ void %s<Cfg<%s>>%s;\n' % (root, kern, ", ".join(cfg), sigs[kern]))
out = src.replace(".hip", ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-invalid-offsetof",
                       "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-o", out, src] + [a for a in sys.argv[6:]])
txt = open(out).read().split("\n")
meta = {k: None for k in (".vgpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size")}
for l in txt:
    for k in meta:
        if l.strip().startswith(k + ":"): meta[k] = l.split(":")[1].strip()
print(" ".join("%s=%s" % (k[1:], v) for k, v in meta.items()))
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "scratch" if op.startswith("scratch_") else "vmem"
    return "other"
# loops: top-level loop headers (Depth=1) and their extents (until next Depth=1 header or function end)
heads = [i for i, l in enumerate(txt) if "Loop Header: Depth=1" in l]
end = next(i for i, l in enumerate(txt) if l.strip().startswith("s_endpgm"))
tot = collections.Counter()
for i, l in enumerate(txt[:end]):
    m = re.match(r"\t([a-z_0-9]+)", l)
    if m: tot[cat(m.group(1))] += 1
print("whole kernel:", sum(tot.values()), dict(tot))
for a, b in zip(heads, heads[1:] + [end]):
    c = collections.Counter()
    for l in txt[a:b]:
        m = re.match(r"\t([a-z_0-9]+)", l)
        if m: c[cat(m.group(1))] += 1
    n = sum(c.values())
    if n > 150: print("loop@%d: %d %s" % (a + 1, n, dict(c)))
