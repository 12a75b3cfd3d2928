"""Per-launch timeline of the Cholesky sweep (GPU box): for one steady-state frame, when each role of a k_chol_step launch
ends relative to the launch's first workgroup.  usage: sweep_trace.py [N] [warm frames]
Needs a DEBUG build of the engine (the trace and its role ablations are not in the product library):
    EKF_EXTRA_FLAGS=-DEKF_SWEEP_TRACE python -c "from openekfmonoslam_amd import build; build.build_engine(force=True)"
(scripts/build_variant.sh-style: build it into variants/ and point EKF_ENGINE_LIB at it)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 15
seq = SyntheticSequence(N, F + 1)
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=int(os.environ.get('PRECISION', '1' if N >= 1000 else '0')))
e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
e.set_update_path(int(os.environ.get('UPDATE_PATH', '0')))
e.set_sweep_mode(int(os.environ.get('SWEEP_MODE', '2')))
L = engine.load_library()
if not hasattr(L, "ekf_debug_sweep_trace"):
    sys.exit("this libekf_engine.so was built without -DEKF_SWEEP_TRACE (see the docstring)")
fn = L.ekf_debug_sweep_trace
fn.restype = C.c_int
fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
for t in range(F):
    e.step(*seq.frames[t])
fn(1 | (int(os.environ.get('ABL', '0')) << 8), None, None)
try:
    info = e.step(*seq.frames[F])
except Exception as ex:  # an ablated role leaves a wrong factor: the timeline is still valid
    print('step failed (expected with ABL):', ex)
    info = None
SL = 16
buf = np.zeros(SL * 4096, dtype=np.uint64)
cnt = C.c_int(0)
fn(0, buf.ctypes.data_as(C.c_void_p), C.byref(cnt))
rows = buf[: SL * cnt.value].reshape(-1, SL)
print(f"frame: {(info.n_matches, info.n_inliers, info.n_rescued) if info else None}; {cnt.value} sweep launches")
us = lambda x, t0: (int(x) - t0) / 100.0 if int(x) else float("nan")
if int(os.environ.get("SWEEP_MODE", "1")) == 1:
    print("   k0     m   lookahead   B-role   tiles     rhs   (us after the launch's first workgroup started)  last tile start  last rhs start")
    for r in rows:
        t0 = int(r[0])
        d = [us(x, t0) for x in r[1:5]]
        st = [us(x, t0) for x in r[6:8]]
        print(f"{int(r[5]) >> 32:5d} {int(r[5]) & 0xffffffff:5d}   {d[0]:8.2f} {d[1]:8.2f} {d[2]:8.2f} {d[3]:8.2f}   {st[0]:8.2f} {st[1]:8.2f}")
else:  # two panels per launch (chol_pair.h): role ends, and the look-ahead workgroup's milestones
    print("   k0     m launch | look-ahead: loaded  updated  factor P  Q ready  factor Q      end | B sums    B end    tiles      rhs   (us after the launch's first workgroup started; single launches: look-ahead end, B end, tiles, rhs only)")
    for r in rows:
        t0 = int(r[0])
        la = [us(r[i], t0) for i in (8, 9, 10, 11, 12, 1)]
        ro = [us(r[i], t0) for i in (14, 2, 3, 4)]
        pair = "pair  " if int(r[5]) & 0x80000000 else "single"
        print(f"{int(r[5]) >> 32:5d} {int(r[5]) & 0x7fffffff:5d} {pair} | " + " ".join(f"{x:8.2f}" for x in la) + " | " + " ".join(f"{x:8.2f}" for x in ro))
