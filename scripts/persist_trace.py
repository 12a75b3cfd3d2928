"""Timeline of ONE persistent Cholesky sweep (chol_persist.h) on the GPU box: per panel, when the chain workgroup started,
finished the factor-and-invert, published inv(L_kk) and finished its own tile; when the two tiles it needs next were published;
the first / last B worker's stages; the latest tile workers.  usage: persist_trace.py [N] [warm frames] [update: 0 LI, 1 HI]
Needs the debug build: scripts/build_trace_variant.sh, then EKF_ENGINE_LIB=variants/libekf_engine_trace.so python scripts/persist_trace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 12
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
seq = SyntheticSequence(N, F + 1)
prec = int(os.environ.get("PRECISION", "2" if N >= 1000 else "0"))
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=prec)
e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
L = engine.load_library()
if not hasattr(L, "ekf_debug_persist_trace"):
    sys.exit("this libekf_engine.so was built without -DEKF_SWEEP_TRACE")
fn = L.ekf_debug_persist_trace
fn.restype = C.c_int
fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
for t in range(F):
    e.step(*seq.frames[t])
# stage calls of one more frame, tracing the chosen update only
kps, desc = seq.frames[F]
e.predict()
e.predict_measurements()
mt = e.match(kps, desc)
mask, _ = e.ransac(mt)
if which == 0:
    fn(1, None, None)
e.update(mt[mask])
if which == 0:
    fn(0, None, None)
else:
    outl = mt[~mask]
    p2, _, _ = e.predict_measurements(feat_idx=outl["featureIndex"])
    outl = outl[np.isin(outl["featureIndex"], p2["featureIndex"])]
    rm = e.rescue(outl)
    fn(1, None, None)
    e.update(outl[rm])
    fn(0, None, None)
buf = np.zeros(4096, dtype=np.uint64)
m = C.c_int(0)
fn(0, buf.ctypes.data_as(C.c_void_p), C.byref(m))
nbk = (m.value + 31) // 32
ch = buf[:1024].reshape(-1, 8)
b0 = buf[1024:2048].reshape(-1, 8)
b1 = np.zeros((128, 8), dtype=np.uint64)
tlog = buf[2048:3072]
tl = buf[3072:4096].reshape(-1, 8)
t0 = int(ch[0][0])
us = lambda x: (int(x) - t0) / 100.0 if int(x) else float("nan")  # noqa: E731
print(f"N={N} precision {prec}: update of m={m.value} rows, {nbk} panels; us after the chain's first panel started")
print("  k | chain: start [A half  I half  band of two]  factored published  own tile | next tiles ready | first B: start  sumA  L(k,k-1)  inv  [summed  product  cut]  end | last B: start   end | tiles: col k+1  L(i,k)   all")
for k in range(nbk):
    c, x, y, z = ch[k], b0[k], b1[k], tl[k]
    print(f"{k:3d} | {us(c[0]):8.2f} [{us(c[5]):7.2f} {us(c[6]):7.2f} {us(c[7]):7.2f}] {us(c[1]):8.2f} {us(c[2]):8.2f} {us(c[3]):8.2f} | {us(c[4]):8.2f} | "
          f"{us(x[0]):8.2f} {us(x[1]):8.2f} {us(x[2]):8.2f} {us(x[3]):8.2f} [{us(x[5]):7.2f} {us(x[6]):7.2f} {us(x[7]):7.2f}] {us(x[4]):8.2f} | {us(y[0]):8.2f} {us(y[4]):8.2f} | "
          f"{us(z[0]):8.2f} {us(z[1]):8.2f} {us(z[2]):8.2f}")

nlog = int(tlog[0])
print(f"task log of tile worker n_t/3: {nlog} tasks (kind 0 S-form update, 1 panel-block step, 2 L-form batch)")
prev = None
for n in range(min(nlog, 250)):
    t_s, t_e, info = int(tlog[4 + 4 * n]), int(tlog[4 + 4 * n + 1]), int(tlog[4 + 4 * n + 2])
    kind, k, nb, i, j = info & 255, (info >> 8) & 255, (info >> 16) & 255, (info >> 24) & 255, (info >> 32) & 255
    gap = (t_s - prev) / 100.0 if prev else 0.0
    print(f"  {n:3d}: tile ({i:2d},{j:2d}) kind {kind} panel {k:2d} x{nb}  picked at {us(t_s):8.2f} (idle/scan {gap:5.2f})  took {(t_e - t_s) / 100.0:5.2f} us")
    prev = t_e
