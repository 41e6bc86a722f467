#!/usr/bin/env python3
"""Round 4: per-kernel means of the rocprofv3 --pmc passes of scripts/gpu_floor_r04.sh (one directory per pass), with the kernel
durations from the same runs' kernel traces, and a few derived figures per 4096^2 slab (64 slabs per launch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
nslab = float(os.environ.get("NT", "64"))
for d in sorted(glob.glob(os.path.join(root, "*", ""))):
    name = os.path.basename(os.path.dirname(d))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "xrft::" not in k:
                    continue
                k = k.split("(")[0].replace("xrft::", "").split("<")[0]
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "xrft::" not in k:
                    continue
                k = k.split("(")[0].replace("xrft::", "").split("<")[0]
                dur[k].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
    if not agg:
        continue
    print(f"== pass {name}")
    for k in sorted(agg):
        # the last dispatches are the steady state (the first call builds tables / warms up)
        parts = []
        for c in sorted(agg[k]):
            v = agg[k][c]
            v = v[len(v) // 2:] if len(v) > 1 else v
            parts.append(f"{c}={sum(v) / len(v):.4g}")
        t = dur.get(k, [])
        t = t[len(t) // 2:] if len(t) > 1 else t
        ts = f" dur_us={sum(t) / len(t):.1f} ({sum(t) / len(t) / nslab:.2f}/slab)" if t else ""
        print(f"  {k:22s}{ts}  " + "  ".join(parts))
