"""Summarise gpurun_out/pm{1,2,3} counter passes per kernel (averages per launch)."""
import csv, glob, collections, sys
want = sys.argv[1:] or ["render_bwd", "render_fwd"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pm1", "pm2", "pm3"):
    for f in glob.glob(f"/root/repo/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for w in want:
                if w in k:
                    acc[w][r["Counter_Name"]].append(float(r["Counter_Value"]))
for w in want:
    print("==", w)
    c = {k: sum(v) / len(v) for k, v in acc[w].items()}
    for k in sorted(c):
        print("  %-26s %14.0f" % (k, c[k]))
    if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
        tot = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0) + c.get("SQ_INSTS_SMEM", 0) + c.get("SQ_INSTS_BRANCH", 0)
        print("  instr/wave: VALU %.0f SALU %.0f LDS %.0f BRANCH %.0f total %.0f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"],
              c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"], c.get("SQ_INSTS_BRANCH", 0) / c["SQ_WAVES"], tot / c["SQ_WAVES"]))
