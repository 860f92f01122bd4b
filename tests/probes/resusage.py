"""Summarise -Rpass-analysis=kernel-resource-usage logs: python tests/probes/resusage.py LOG... [regex]"""
import re, subprocess, sys
pat = r"k_newton_solve|k_ibr|k_direction"
files = [a for a in sys.argv[1:] if a.endswith(".log")]
rest = [a for a in sys.argv[1:] if not a.endswith(".log")]
if rest: pat = rest[0]
for f in files:
    txt = open(f).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not re.search(pat, dem): continue
        def g(k):
            m = re.search(k + r": (\d+)", b); return m.group(1) if m else "?"
        short = re.sub(r"void |alg::|\(.*", "", dem)
        print("%-58s vgpr %4s agpr %3s scratch %5s occ %s lds %s" % (short, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
