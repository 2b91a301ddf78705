import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            print(row["Kernel_Name"][:48], (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6, "ms")
