#!/usr/bin/env python3
"""profiles/r01_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc.sh.
usage: make_traffic_json.py gpurun_out/pmc_<tag> <slabs per profiled launch> > profiles/r01_traffic.json"""
import collections, csv, glob, json, os, re, sys
root, nslab = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            m = re.search(r"xrft::(fastp2_\w+?)_kernel", k)
            if m and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": f"{root} (scripts/gpu_pmc.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, bench.py --nt {nslab} --steps 1)",
       "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced streaming reads, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported; both x1024 B",
       "slabs_per_profiled_launch": nslab, "kernels": {}}
tot = 0.0
for k, c in sorted(agg.items()):
    fs = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    ws = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    b = (2 * fs + ws) * 1024 / nslab
    out["kernels"][k] = {"FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1), "hbm_bytes_per_slab": int(b), "launches": len(c["FETCH_SIZE"])}
    tot += b
out["path_hbm_bytes_per_slab"] = int(tot)
out["algorithmic_bytes_per_slab"] = 4096 * 4096 * 8
print(json.dumps(out, indent=1))
