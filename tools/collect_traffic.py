"""Turn rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/traffic_pmc.json (mean counter value
per launch of each timed kernel of the 4K step; with a second directory, of the 1080p step as well).
usage: python tools/collect_traffic.py <pmc_dir_4k> <frames_per_gpu> <out.json> [<pmc_dir_1080p>]"""
import csv, glob, json, sys
root, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
root_hd = sys.argv[4] if len(sys.argv) > 4 else None
KERNELS = {"sepconv": "sepconv_march_kernel", "affine_bilinear": "affine_bilinear_wq_kernel"}


def collect(root, px):
    acc = {k: {"FETCH_SIZE": [], "WRITE_SIZE": []} for k in KERNELS}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            for key, sub in KERNELS.items():
                if sub in row["Kernel_Name"] and row["Counter_Name"] in acc[key]:
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for key, v in acc.items():
        if v["FETCH_SIZE"] and v["WRITE_SIZE"]:
            # bench.py also launches these kernels on smaller batches (the checksum check): keep the launches of the timed
            # batch, i.e. those within 10 % of the largest value
            full = lambda xs: [x for x in xs if x >= 0.9 * max(xs)]
            v = {k: full(x) for k, x in v.items()}
            fe, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            res[key] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "launches": [len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"])],
                        "bytes_per_px_corrected": round((2 * fe + wr) * 1024 / px, 4)}
    return res


rec = {"frames_per_gpu": frames, "kernels": collect(root, frames * 2160 * 3840),
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
              "--steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-1080p (kernels_1080p: --only-1080p, 4 x as many 1920x1080 "
              "frames); traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB "
              "(gfx950: FETCH_SIZE counts half the bytes of a wide 16-B-per-lane stream, MI355X_MICROARCH.md)"}
if root_hd:
    rec["kernels_1080p"] = collect(root_hd, 4 * frames * 1080 * 1920)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
