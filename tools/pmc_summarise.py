#!/usr/bin/env python
"""Sum a PMC counter over the gemm_kernel dispatches of a rocprofv3 counter_collection CSV."""
import csv, glob, json, os, sys
out = {}
for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    tot, n, name = 0.0, 0, None
    for f in files:
        for row in csv.DictReader(open(f)):
            if "gemm_kernel" in row.get("Kernel_Name", ""):
                tot += float(row["Counter_Value"]); n += 1; name = row["Counter_Name"]
    out[name or os.path.basename(d)] = {"sum": tot, "dispatch_rows": n}
print(json.dumps(out))
json.dump(out, open("gpurun_out/pmc_summary.json", "w"))
