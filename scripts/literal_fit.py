"""The reference's LITERAL update algorithm (oracle LITERAL variant, Update.cpp:92-109,214-218) timed on this host at
N = 200, 350, 500 (whole frames) -> profiles/r02_literal_cpu_fit.json, the committed points bench.py's
cpu_baseline extrapolates from.  ~2-3 minutes of one core.  usage: literal_fit.py [out.json]"""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_literal_cpu_fit.json")
points = []
for N, frames, budget in ((200, 4, 20.0), (350, 1, 1.0), (500, 1, 1.0)):
    p = bench.time_literal_frames(N, 640, 480, frames, budget)
    print(p, flush=True)
    points.append(p)
cpu = ""
try:
    with open("/proc/cpuinfo") as f:
        cpu = [l.split(":")[1].strip() for l in f if l.startswith("model name")][0]
except Exception:  # noqa: BLE001
    pass
# cubic law: seconds per update ~ c n^3 at m ~ n/6
json.dump({"host_cpu": cpu, "host_cores": os.cpu_count(), "threads_used": 1, "machine": platform.machine(),
           "what": "oracle LITERAL variant, whole frames of the synthetic sequence (two updates per frame)", "points": points},
          open(out, "w"), indent=1)
print("wrote", out)
