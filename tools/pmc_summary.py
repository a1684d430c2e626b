"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k: continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v)//2:]  # drop warm-up half
        print(f"   {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
