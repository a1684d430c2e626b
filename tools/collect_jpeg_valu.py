"""rocprofv3 --pmc SQ_INSTS_VALU pass of tools/jpeg_once.py -> profiles/jpeg_valu_pmc.json: wave64 VALU instructions
of the JPEG writer per pixel (all jpeg_* kernels of one encode call).  usage: collect_jpeg_valu.py <pmc_dir> <out.json>"""
import csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
CALLS, PX = 3, 16 * 2160 * 3840
per_kernel = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "SQ_INSTS_VALU" and "jpeg_" in row["Kernel_Name"]:
            k = row["Kernel_Name"].split("(")[0].replace("imgxf::", "")
            per_kernel[k] = per_kernel.get(k, 0.0) + float(row["Counter_Value"])
total = sum(per_kernel.values())
rec = {"valu_insts_per_px": round(total / CALLS / PX, 4), "frames": 16, "calls": CALLS,
       "per_kernel_per_px": {k: round(v / CALLS / PX, 4) for k, v in sorted(per_kernel.items())},
       "how": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- python3 tools/jpeg_once.py; wave64 instructions summed over "
              "the jpeg_* kernels of one imgxf_jpeg_encode_u8 call on 16 uniform-noise 4K frames, per pixel"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
