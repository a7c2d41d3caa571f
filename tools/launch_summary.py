"""Aggregate an SO_PROF_DUMP per-launch CSV (key,M,N,K,nclass,splitk,us,tflops) by problem shape.

    SO_PROF_DUMP=launches.csv python bench.py ...; python tools/launch_summary.py launches.csv STEPS [mode] [top]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else ""
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
names = {0: "fprop", 1: "dgrad", 2: "wgrad", 3: "gemm"}
agg = collections.OrderedDict()
for r in rows:
    if mode and names[int(r["key"]) // 8] != mode:
        continue
    k = (r["key"], r["M"], r["N"], r["K"], r["nclass"], r["splitk"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["us"])
    a[2] = float(r["tflops"])
print("total ms/step %.3f" % (sum(a[1] for a in agg.values()) / steps / 1000))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    key = int(k[0])
    print(f"{names[key // 8]}/{key % 8} M={k[1]:>7} N={k[2]:>5} K={k[3]:>6} cls={k[4]} sk={k[5]:>3} n/step={a[0] / steps:5.1f} "
          f"us={a[1] / a[0]:8.1f} ms/step={a[1] / steps / 1000:6.3f} tf={a[2]:6.1f}")
