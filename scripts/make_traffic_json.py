#!/usr/bin/env python3
"""profiles/r0N_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc_yf.sh (bench.py --nt <n> --steps 1).
usage: make_traffic_json.py gpurun_out/pmc_<tag> <slabs per profiled launch> [fused-skeleton result file] > profiles/r03_traffic.json
The file is stamped with the SHA-1 of xrft_amd/csrc (bench.csrc_sha1): bench.py reports `traffic` only when the stamp matches
the sources it runs.

Correction: FETCH_SIZE x 2 (gfx950 tallies the L2's 128-byte fabric read requests at 64 B, MI355X_MICROARCH.md 'HBM'; checked
here on kernels with known byte counts: plain copy, and the pass-1 / pass-2 skeletons of scripts/ubench/yfirst.hip, see
profiles/r02_pmc_ubench_calibration.txt), WRITE_SIZE as reported (exact on the same skeletons); both x 1024 B."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
root, nslab = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            m = re.search(r"xrft::(fast[pyr2]*_\w+?)_kernel", k) or re.search(r"xrft::(fastr)_kernel", k)
            if m and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"csrc_sha1": bench.csrc_sha1(),
       "source": f"{root} (scripts/gpu_pmc_yf.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, bench.py --nt {nslab} --steps 1 --warmup 1)",
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
if os.environ.get("TRAFFIC_WORKLOAD", "ps") == "c2":  # (1024, 65536) float32 -> complex64: a "slab" is one row
    out["algorithmic_bytes_per_slab"] = 65536 * 12
    out["hbm_bytes_per_point"] = round(tot / 65536, 3)
    out["algorithmic_bytes_per_point"] = 12
else:
    out["algorithmic_bytes_per_slab"] = 4096 * 4096 * 8
    out["two_pass_minimum_bytes_per_slab"] = 4096 * 4096 * 4 * 2 + 2 * 2052 * 4096 * 8  # in + out + the half-spectrum intermediate written and read once
if len(sys.argv) > 3:  # what the memory system allows the two passes' access patterns with no arithmetic (scripts/ubench/fused.hip)
    txt = open(sys.argv[3]).read()
    best = None
    for m in re.finditer(r"two launches\s+wp=(\d) inp=(\d) w2l=(\d) work=\s*0:\s+([\d.]+) us / slab \(cols\s+([\d.]+) rows\s+([\d.]+)\)", txt):
        us = float(m.group(4))
        if best is None or us < best[0]:
            best = (us, float(m.group(5)), float(m.group(6)), m.group(1))
    if best:
        import hashlib
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "fused.hip"), "rb") as fh:
            out["ubench_sha1"] = hashlib.sha1(fh.read()).hexdigest()
        out["claimed_floor"] = {"us_per_slab": best[0], "cols_us": best[1], "rows_us": best[2], "GFFT_per_s": round(4096 * 4096 / best[0] / 1e3, 1),
                                 "frac_of_8TBps_on_algorithmic_bytes": round(4096 * 4096 * 8 / (best[0] * 1e-6) / 8e12, 3),
                                 "source": "scripts/ubench/fused.hip, 'two launches' (profiles/r03_ubench_fused.txt): the real kernels' workgroup shape, LDS footprint and "
                                           "access patterns (32-byte row segments in, 16-byte pieces of 128-byte lines out; one contiguous 128-KB block in, eight output rows "
                                           "out) with no arithmetic, 32 slabs per launch.  The copy floor of this memory system (profiles/r03_ubench_membw.txt): "
                                           "5.2-5.6 TB/s read + written, reads alone 6.3-6.7, writes alone 5.3-5.7; a persistent fused cols -> rows pipeline with the "
                                           "intermediate in a ring of 3 slabs does not beat the two launches (r03_ubench_fused.txt: 51-54 us at best), and the Infinity "
                                           "Cache serves hits and HBM traffic through one ~7 TB/s path (r03_ubench_mall.txt (3))"}
print(json.dumps(out, indent=1))
