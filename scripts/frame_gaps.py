"""One steady-state frame of a rocprofv3 --kernel-trace run as a timeline: every launch with its start (relative to the frame's
first kernel), duration and the idle gap before it; totals per frame.  usage: frame_gaps.py <kernel_trace.csv> [frame index]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void |ekf::|<.*|\(.*", "", r["Kernel_Name"])))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_predict_prepare"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = rows[a][0]
busy = gap_sum = 0
prev_end = t0
print(f"frame {k}: {b - a} launches")
for s, e, name in rows[a:b]:
    gap = s - prev_end
    gap_sum += max(gap, 0)
    busy += e - s
    if name != "k_chol_step" or gap > 3000:
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  {name}")
    prev_end = max(prev_end, e)
print(f"wall {(rows[b][0] - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {gap_sum / 1e3:.1f} us (gap before the next frame {(rows[b][0] - prev_end) / 1e3:.1f})")
# steady state over all frames in the second half
tot = {}
for i in range(len(starts) // 2, len(starts) - 1):
    pe = rows[starts[i]][0]
    for s, e, name in rows[starts[i]:starts[i + 1]]:
        d = tot.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        d[2] += max(s - pe, 0) / 1e3
        pe = max(pe, e)
nf = len(starts) - 1 - len(starts) // 2
print(f"\nper frame over {nf} frames: kernel, launches, busy us, gap-before us")
for name, (c, bu, g) in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{name:24s} {c / nf:6.1f} {bu / nf:8.1f} {g / nf:8.1f}")
print(f"{'total':24s} {sum(v[0] for v in tot.values()) / nf:6.1f} {sum(v[1] for v in tot.values()) / nf:8.1f} {sum(v[2] for v in tot.values()) / nf:8.1f}")
