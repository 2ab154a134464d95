#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV: start (us, relative to the
step's first kernel), duration, gap to the previous kernel's end, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_bam_filter")]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  dur %6.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
    prev_end = max(prev_end, e)
print("step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
