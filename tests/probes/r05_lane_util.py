"""Per-kernel lane utilisation from a rocprofv3 --pmc pass (tests/probes/r05_lane_util.sh): mean active lanes per VALU instruction =
SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (both count quad-cycles of VALU work; the first multiplied by the exec-mask popcount).
usage: python r05_lane_util.py DIR"""
import csv, glob, sys, collections
f = [x for x in glob.glob(sys.argv[1] + "/**/*.csv", recursive=True) if "counter_collection" in x][0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
print("%-44s %8s %9s %12s %12s %10s %8s" % ("kernel", "launches", "waves", "VALU/wave", "MFMA/wave", "lanes/VALU", "of 64"))
for k in sorted(acc):
    if not k.startswith("void k_"):
        continue
    a = acc[k]; n = cnt[k]; w = a["SQ_WAVES"] / n
    lanes = a["SQ_THREAD_CYCLES_VALU"] / max(a["SQ_ACTIVE_INST_VALU"], 1.0)
    print("%-44s %8d %9.0f %12.0f %12.0f %10.2f %7.1f%%" % (k[5:49], n, w, a["SQ_INSTS_VALU"] / n / max(w, 1), a["SQ_INSTS_VALU_MFMA_F64"] / n / max(w, 1), lanes, 100 * lanes / 64))
    print("     raw per launch: " + " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(a.items())))
