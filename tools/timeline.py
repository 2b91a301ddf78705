"""Kernel timeline of a rocprofv3 --kernel-trace csv: start/end (ms, relative) of the main kernels, to see what overlaps."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("k_spcbpt", "k_light_trace", "k_film_merge", "k_cmf", "k_lvc_compact")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void spc::", "")[:24], r.get("Stream_Id", "")))
rows.sort()
t0 = rows[0][0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for s, e, n, st in rows[-last:]:
    print(f"{(s - t0) / 1e6:10.3f} -> {(e - t0) / 1e6:10.3f}  ({(e - s) / 1e6:7.3f} ms)  {n}  stream {st}")
