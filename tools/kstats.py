import csv, glob, sys, collections
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].split("(")[0][-40:]
        agg[n][0] += 1; agg[n][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:42s} calls {v[0]:6d}  total {v[1]:9.3f} ms  per-frame {v[1]/frames:8.3f} ms  {100*v[1]/tot:5.1f}%")
