"""EKF updates/s when every frame's keypoints and descriptors come from host buffers (ekf_step: K x 40 B over PCIe per
frame) instead of a sequence staged in HBM (ekf_step_frame, what bench.py reports).  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
N, W, K = 1000, 10, 60
seq = SyntheticSequence(N, W + K)
e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=2 * N + 64, precision=1)
for mode in ("host buffers (ekf_step)", "staged in HBM (ekf_step_frame)"):
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    if mode.startswith("staged"):
        e.upload_frames(seq.frames)
    step = (lambda t: e.step(*seq.frames[t])) if mode.startswith("host") else (lambda t: e.step_frame(t))
    for t in range(W):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        step(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode}: {K / dt:.1f} updates/s ({1e3 * dt / K:.3f} ms/frame)")
