#!/usr/bin/env python3
"""Two exact engines stepped from two host threads, many times: are their results bit for bit those of a run alone, and if not, by how
much and after how many sweep retries?  (tests/test_gpu_exact_edge_cases.py::test_concurrent_exact_engines_give_the_sequential_result)
usage: concurrency_probe.py [rounds=40] [N=300] [frames=4] [sweep_mode=2]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openekfmonoslam_amd import engine as eng_mod  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
seq = SyntheticSequence(N, frames)


def run(out, k):
    e = eng_mod.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=2)
    e.set_sweep_mode(mode)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    for t in range(frames):
        e.step(*seq.frames[t])
    out[k] = (e.get_state(), e.sweep_retries)
    e.close()


ref = {}
run(ref, 0)
bad = 0
for r in range(rounds):
    got = {}
    th = [threading.Thread(target=run, args=(got, k)) for k in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in (1, 2):
        diffs = []
        for name, a, b in zip(("x13", "features", "P"), ref[0][0], got[k][0]):
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            if not np.array_equal(a, b):
                diffs.append((name, float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300)), int(np.count_nonzero(a != b))))
        if diffs:
            bad += 1
            print(f"round {r} engine {k}: differs {diffs}; sweep retries {got[k][1]}", flush=True)
print(f"N={N} frames={frames} sweep_mode={mode}: {bad} of {2 * rounds} concurrent runs differ from the run alone")
