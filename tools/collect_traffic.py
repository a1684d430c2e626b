"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/gaussian_pmc.json (mean counter value per Gaussian launch).
usage: python tools/collect_traffic.py <pmc_dir> <frames_per_gpu> <out.json>"""
import csv, glob, json, sys
root, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sepconv" in row["Kernel_Name"] and row["Counter_Name"] in vals:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
rec = {"frames_per_gpu": frames, "kernel": "sepconv_march_kernel<3,2>",
       "FETCH_SIZE_KiB": sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"])),
       "WRITE_SIZE_KiB": sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"])),
       "launches": [len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])],
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline"}
json.dump(rec, open(out, "w"), indent=1)
print(rec)
