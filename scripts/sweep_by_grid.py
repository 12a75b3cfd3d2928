"""Mean duration of the sweep's launches by grid size (= by panel index within an update) from a rocprofv3 kernel trace CSV:
python scripts/sweep_by_grid.py <kernel_trace.csv> [kernel substring, default k_chol_step]"""
import csv
import sys
from collections import defaultdict

path, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_chol_step")
by = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        if needle in r["Kernel_Name"]:
            g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            w = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256)
            by[g // max(w, 1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
print(f"{needle}: {sum(len(v) for v in by.values())} launches, {tot / 1e3:.2f} ms")
for g in sorted(by, reverse=True):
    v = by[g]
    print(f"workgroups {g:5d}  launches {len(v):5d}  mean {sum(v) / len(v):7.2f} us  min {min(v):7.2f}  share {100 * sum(v) / tot:5.1f} %")
