"""Sum rocprofv3 --pmc counter CSVs per kernel name (diagnostic helper)."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "epw_kernel"
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:28s} per-launch {acc[k][0] / max(acc[k][1], 1):16.1f}   (n={acc[k][1]})")
