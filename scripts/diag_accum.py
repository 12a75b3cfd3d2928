"""Which rounding limits the cross-covariance blocks of the fp32 downdate P -= B'B?  CPU only (oracle = the fp64 checker).
Rebuilds one low-innovation update of the N-feature synthetic scene in numpy fp64 (G = H P rows of the RANSAC inliers,
S = G H' + R, L = chol(S), B = inv(L) G) and evaluates (B'B)[i][j] on sampled column pairs with:
  E_round   B rounded to fp32, products and sums exact (fp64)           -> what fp32 STORAGE of B costs
  E_bcomp   B from an fp32 blocked forward substitution (fp32 L, fp32 partial sums like the sweep's B role), sums in fp64
  E_seq     B rounded to fp32, k-sequential fp32 fmaf chain (what v_mfma_f32_32x32x2 does)
  E_chunk16 / E_chunk32   the same chain restarted every 16 / 32 rows, chunks added in fp64
  E_i8xS    fp64 B cut per COLUMN into S balanced int8 digits (scale = power of two >= the column's max |B|), digit
            products of level s + t < S only, integer sums exact (what v_mfma_i32_32x32x32_i8 does), levels combined in fp64
Errors are absolute, against the fp64 value; also shown relative to rms|(B'B)[i][j]| of the sampled pairs."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.linalg as sla

import oracle_lib as ol
from openekfmonoslam_amd.synth import SyntheticSequence

SRC = r"""
#include <math.h>
void dots(const float *B, int m, int ld, const int *pi, const int *pj, int np, int chunk, double *out)
{
    for (int p = 0; p < np; ++p) {
        const float *a = B + pi[p], *b = B + pj[p];
        double tot = 0.0; float acc = 0.f;
        for (int k = 0; k < m; ++k) {
            acc = fmaf(a[(long)k * ld], b[(long)k * ld], acc);
            if (chunk > 0 && (k + 1) % chunk == 0) { tot += (double)acc; acc = 0.f; }
        }
        out[p] = tot + (double)acc;
    }
}
"""


def helper():
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "h.c"), "w") as f:
        f.write(SRC)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(d, "h.so"), os.path.join(d, "h.c"), "-lm"])
    L = C.CDLL(os.path.join(d, "h.so"))
    L.dots.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ol.build()
seq = SyntheticSequence(N, t0 + 1)
o = ol.Oracle(seq.cam, seq.par, N + 8)
o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
for t in range(t0):
    o.step(*seq.frames[t], ol.ALGORITHMIC)
o.predict()
preds, Hs, Hf, HP = o.predict_measurements(want_HP=True)
matches = o.match(preds, *seq.frames[t0])
pa, Hsa, Hfa = ol.align_to_matches(preds, Hs, Hf, matches)
mask, _ = o.ransac(pa, Hsa, Hfa, matches)
lut = {int(p["featureIndex"]): k for k, p in enumerate(preds)}
sel = [lut[int(mt["featureIndex"])] for mt, keep in zip(matches, mask) if keep]
M = len(sel)
n = o.n
G = HP[sel].reshape(2 * M, n)                      # m x n
covpos = o.feature_covpos()
S = np.zeros((2 * M, 2 * M))
for b, k in enumerate(sel):
    f = int(preds[k]["featureIndex"])
    pos = covpos[f]
    S[:, 2 * b:2 * b + 2] = G[:, 0:13] @ Hs[k].T + G[:, pos:pos + 6] @ Hf[k].T
S = 0.5 * (S + S.T) + seq.cam.pixelErrorX * np.eye(2 * M)
Lc = np.linalg.cholesky(S)
B = sla.solve_triangular(Lc, G, lower=True)        # fp64
m = 2 * M
print(f"N={N} frame {t0}: low-innovation update with M={M} inliers (m={m}), n={n}; max|B| {np.abs(B).max():.3f} rms {np.sqrt((B**2).mean()):.2e}")
B32 = B.astype(np.float32)
# fp32 blocked forward substitution, the shape of the sweep's B role: fp32 L' and fp32 B_j, partial sums in fp32 (numpy sgemm),
# the last 32x32 solve in fp64, result rounded to fp32
L32 = Lc.astype(np.float32)
Bc = np.zeros_like(B32)
nb = (m + 31) // 32
for k in range(nb):
    r0, r1 = 32 * k, min(m, 32 * k + 32)
    acc = np.zeros((r1 - r0, n), dtype=np.float32)
    if k > 0:
        acc = L32[r0:r1, :r0] @ Bc[:r0]
    rhs = G[r0:r1].astype(np.float32).astype(np.float64) - acc.astype(np.float64)
    Bc[r0:r1] = sla.solve_triangular(Lc[r0:r1, r0:r1], rhs, lower=True).astype(np.float32)
print(f"B from the fp32 forward substitution vs correctly rounded B: max abs {np.abs(Bc.astype(np.float64) - B).max():.2e}, "
      f"rel to max|B| {np.abs(Bc.astype(np.float64) - B).max() / np.abs(B).max():.2e}; rounding alone {np.abs(B32.astype(np.float64) - B).max():.2e}")
rng = np.random.default_rng(1)
# sampled pairs: cross-feature pairs (random), and the whole column of the feature with the smallest |rho|
fp = o.feature_pos()
fsm = int(np.argmin(np.abs(fp[:, 5])))
col = covpos[fsm] + 5
pi = np.concatenate([rng.integers(13, n, 4000), np.arange(13, n, 7)]).astype(np.int32)
pj = np.concatenate([rng.integers(13, n, 4000), np.full(len(np.arange(13, n, 7)), col)]).astype(np.int32)
keep = (pi - 13) // 6 != (pj - 13) // 6
pi, pj = np.ascontiguousarray(pi[keep]), np.ascontiguousarray(pj[keep])
exact = np.einsum("kp,kp->p", B[:, pi], B[:, pj])
H = helper()


def dots(Bm, chunk):
    Bm = np.ascontiguousarray(Bm, dtype=np.float32)
    out = np.zeros(len(pi))
    H.dots(Bm.ctypes.data, m, Bm.shape[1], pi.ctypes.data, pj.ctypes.data, len(pi), chunk, out.ctypes.data)
    return out


res = {
    "E_round   (B fp32-rounded, exact sums)": np.einsum("kp,kp->p", B32[:, pi].astype(np.float64), B32[:, pj].astype(np.float64)),
    "E_bcomp   (B by fp32 forward subst., exact sums)": np.einsum("kp,kp->p", Bc[:, pi].astype(np.float64), Bc[:, pj].astype(np.float64)),
    "E_seq     (rounded B, fp32 fmaf chain)": dots(B32, 0),
    "E_chunk32 (chain restarted every 32 rows, fp64 combine)": dots(B32, 32),
    "E_chunk16": dots(B32, 16),
    "E_engine~ (B by fp32 subst. + fp32 fmaf chain)": dots(Bc, 0),
}


def slices(Bm, S, cmax=None):
    """balanced base-256 digits of every column of Bm relative to a power-of-two column scale: Bm ~ scale * sum_s d_s 256^-(s+1)"""
    if cmax is None:
        cmax = np.abs(Bm).max(axis=0)
    scale = np.where(cmax > 0, 2.0 ** np.ceil(np.log2(np.where(cmax > 0, cmax, 1.0)) + 1e-12), 1.0)  # |Bm / scale| <= 1
    # integer image of the column first, digits from the least significant end: every digit in [-128, 127], no digit overflow
    X = np.rint(Bm / scale * 0.5 * 256.0 ** S).astype(np.int64)
    X = np.clip(X, -(2 ** (8 * S - 1)) + 1, 2 ** (8 * S - 1) - 1 - 128 * sum(256 ** i for i in range(S - 1)) if S > 1 else 127)
    digs = [None] * S
    for s_ in range(S - 1, -1, -1):
        d = ((X + 128) % 256) - 128
        X = (X - d) // 256
        digs[s_] = d
    assert np.all(X == 0) and all(np.abs(d).max() <= 128 and d.max() <= 127 for d in digs)
    return scale * 2.0, digs                             # value = scale2 * sum_s d_s 256^-(s+1)


def dots_i8(Bm, S, cmax=None):
    scale, digs = slices(Bm, S, cmax)
    out = np.zeros(len(pi))
    for lvl in range(S):
        acc = np.zeros(len(pi), dtype=np.int64)
        for s_ in range(lvl + 1):
            acc += np.einsum("kp,kp->p", digs[s_][:, pi], digs[lvl - s_][:, pj])
        out += acc.astype(np.float64) * 256.0 ** -(lvl + 2)
    return out * scale[pi] * scale[pj]


for S_ in (3, 4, 5):
    res[f"E_i8x{S_}    (fp64 B in {S_} int8 digits per column, levels < {S_}, exact sums)"] = dots_i8(B, S_)
# a-priori column scale: sum_k B_kj^2 = P_jj - P_jj(new) <= P_jj, so |B_kj| <= sqrt(P_jj) is known BEFORE B exists
Pdiag = np.diag(o.P()).copy()
apri = np.sqrt(Pdiag) * (1.0 + 1e-6)
for S_ in (5, 6):
    res[f"E_i8x{S_}ap  ({S_} digits, column scale from sqrt(P_jj) instead of the column's max)"] = dots_i8(B, S_, apri)
rms = np.sqrt((exact ** 2).mean())
sm = pj == col
print(f"sampled cross-feature pairs: {len(pi)}; rms|(B'B)_ij| {rms:.2e}; pairs with the smallest-|rho| feature's rho column ({fsm}, rho={fp[fsm, 5]:.2e}): {sm.sum()}")
for name, v in res.items():
    e = np.abs(v - exact)
    print(f"  {name:58s}: max {e.max():.2e}  rms {np.sqrt((e**2).mean()):.2e}  (rms / rms|B'B| {np.sqrt((e**2).mean()) / rms:.1e});  small-rho column: max {e[sm].max():.2e} rms {np.sqrt((e[sm]**2).mean()):.2e}")
