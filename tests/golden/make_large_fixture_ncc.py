"""Mints tests/golden/oracle_n{N}_f{F}_ncc_summary.npz: per-frame SUMMARIES of the fp64 oracle over F image-in steps (matcher
mode B: 3-level pyramid + NCC templates, orc_step_image) of the seeded synthetic sequence -- BASELINE configs[4] as written,
1920x1080 / N = 5000, whose full covariance is too large to commit or to recompute at test time (Matching.h:66, EKF.h:48 in
their image-taking form).  Same summary as make_large_fixture.py: x13, the feature parameters, the camera block, trace, Frobenius
norm, max|P|, diagonal and a 64 x 64 sample of P after every frame, plus the decision counters.  Templates are cut from the
rendered frame 0 at the true pixel positions; the steps run on rendered frames 1 .. F.  ORACLE-minted, ALGORITHMIC variant.

    python tests/golden/make_large_fixture_ncc.py 5000 2
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from make_large_fixture import SIZES, summary  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    W, H = SIZES.get(N, (640, 480))
    seq = SyntheticSequence(N, F, width=W, height=H)
    o = ol.Oracle(seq.cam, seq.par, N + 8)
    o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    img0 = seq.render_image(0)
    uv0 = seq.pixel_positions(0).astype(np.float64)
    o.set_image(img0)
    o.capture_templates(np.arange(N), uv0)
    out = {"n_features": np.int32(N), "frames": np.int32(F), "width": np.int32(W), "height": np.int32(H),
           "input_P0_trace": np.float64(np.trace(seq.P0)), "input_img0_sum": np.int64(img0.astype(np.int64).sum())}
    infos = []
    for t in range(1, F + 1):
        img = seq.render_image(t)
        out[f"input_img{t}_sum"] = np.int64(img.astype(np.int64).sum())
        t0 = time.time()
        i = o.step_image(img, ol.ALGORITHMIC)
        infos.append([i.n_predicted, i.n_matches, i.n_hypotheses, i.n_inliers, i.n_outliers, i.n_rescued, i.status])
        print(f"image step {t}: {infos[-1]}  {time.time() - t0:.1f} s", flush=True)
        out[f"x13_t{t}"] = o.x13()
        out[f"feature_pos_t{t}"] = o.feature_pos()
        for k, v in summary(o.P()).items():
            out[k if k == "sample_idx" else f"{k}_t{t}"] = v
    out["info"] = np.array(infos, dtype=np.int32)
    path = os.path.join(ROOT, "tests", "golden", f"oracle_n{N}_f{F}_ncc_summary.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
