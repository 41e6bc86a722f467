#!/usr/bin/env python3
"""profiles/r02_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc_yf.sh (bench.py --nt <n> --steps 1).
usage: make_traffic_json.py gpurun_out/pmc_<tag> <slabs per profiled launch> [ubench result file] > profiles/r02_traffic.json

Correction: FETCH_SIZE x 2 (gfx950 tallies the L2's 128-byte fabric read requests at 64 B, MI355X_MICROARCH.md 'HBM'; checked
here on kernels with known byte counts: plain copy, and the pass-1 / pass-2 skeletons of scripts/ubench/yfirst.hip, see
profiles/r02_pmc_ubench_calibration.txt), WRITE_SIZE as reported (exact on the same skeletons); both x 1024 B."""
import collections, csv, glob, json, os, re, sys
root, nslab = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            m = re.search(r"xrft::(fast[py2]*_\w+?)_kernel", k)
            if m and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": f"{root} (scripts/gpu_pmc_yf.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, bench.py --nt {nslab} --steps 1 --warmup 1)",
       "note": "HBM-side bytes per step = measured bytes per slab (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, all kernels of the path) x slabs per step",
       "correction": "FETCH_SIZE doubled (gfx950 tallies 128-byte fabric reads at 64 B; calibrated on the skeleton kernels of scripts/ubench/yfirst.hip), WRITE_SIZE as reported; both x1024 B",
       "slabs_per_profiled_launch": nslab, "kernels": {}}
tot = 0.0
for k, c in sorted(agg.items()):
    fs = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    ws = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    b = (2 * fs + ws) * 1024 / nslab
    out["kernels"][k] = {"FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1), "read_bytes_per_slab": int(2 * fs * 1024 / nslab),
                         "write_bytes_per_slab": int(ws * 1024 / nslab), "hbm_bytes_per_slab": int(b), "launches": len(c["FETCH_SIZE"])}
    tot += b
out["path_hbm_bytes_per_slab"] = int(tot)
out["algorithmic_bytes_per_slab"] = 4096 * 4096 * 8
out["two_pass_minimum_bytes_per_slab"] = 4096 * 4096 * 4 * 2 + 2 * 2052 * 4096 * 8  # in + out + the half-spectrum intermediate written and read once
if len(sys.argv) > 3:  # measured two-pass ceiling of the memory system: the no-arithmetic skeletons of the two passes
    txt = open(sys.argv[3]).read()
    m = re.search(r"pass1\(16, xcd\) \+ pass2:\s+([\d.]+) us / slab", txt)
    c = re.search(r"plain copy in->out:\s+([\d.]+) us / slab", txt)
    if m:
        us = float(m.group(1))
        out["two_pass_ceiling"] = {"us_per_slab": us, "GFFT_per_s": round(4096 * 4096 / us / 1e3, 1), "frac_of_8TBps_on_algorithmic_bytes": round(4096 * 4096 * 8 / (us * 1e-6) / 8e12, 3),
                                   "plain_copy_us_per_slab": float(c.group(1)) if c else None,
                                   "source": "scripts/ubench/yfirst.hip (profiles/r02_ubench_yfirst.txt): the two passes' memory access patterns with no arithmetic, 32 slabs cycled; "
                                             "a 4096^2 complex64 half spectrum (67 MB) fits neither LDS + registers (168 MB chip-wide, no cross-CU exchange) nor an XCD's 4 MB L2, "
                                             "so every 2-D FFT of this size makes two trips through the fabric"}
print(json.dumps(out, indent=1))
