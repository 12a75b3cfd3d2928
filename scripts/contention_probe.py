"""How much slower does the latency-bound Cholesky sweep get while another stream keeps the matrix pipes busy?  (Feasibility of
overlapping the covariance downdate of finished row blocks of B with the rest of the sweep)  Filter A (N = 1000,
fp32) steps frames and reports its sweep time per panel; meanwhile filter B (N = 2000, its frame is 60 % downdate at three
workgroups per CU) steps frames from another host thread on its own stream.  The CU-mask rows need the debug build (the mask hook
is not in the product library): scripts/build_trace_variant.sh, EKF_ENGINE_LIB=variants/libekf_engine_trace.so.
usage: contention_probe.py [frames]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openekfmonoslam_amd import engine  # noqa: E402
from openekfmonoslam_amd.synth import SyntheticSequence  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def make(N, frames, mask=None, **kw):
    if mask:
        os.environ["EKF_PROBE_CU_MASK"] = mask
    else:
        os.environ.pop("EKF_PROBE_CU_MASK", None)
    seq = SyntheticSequence(N, frames, **kw)
    e = engine.EkfEngine(seq.cam, seq.par, N, max_keypoints=len(seq.frames[0][0]) + 64, precision=1)
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, seq.P0)
    e.upload_frames(seq.frames)
    e.set_async_errors(True)
    return e


def run_a(e, label):
    for t in range(10):
        e.step_frame(t)
    e.synchronize()
    e.timing(True)
    e.timing_reset()
    import time
    t0 = time.perf_counter()
    for t in range(10, 10 + F):
        e.step_frame(t)
    e.synchronize()
    dt = (time.perf_counter() - t0) / F
    sw = e.sweep_timing()
    st = e.stage_timing() if hasattr(e, "stage_timing") else None
    print(f"{label}: {dt * 1e3:.3f} ms per frame, sweep {1e3 * sw['ms'] / max(sw['panels'], 1):.2f} us per panel ({sw['ms'] / F:.3f} ms per frame)", flush=True)
    e.timing(False)


a = make(1000, 10 + F)
run_a(a, "alone    ")
a.close()
for ma, mb in ((None, None), ("0,96", "96,160"), ("0,128", "128,128"), ("0,64", "64,192")):
    a = make(1000, 10 + F, mask=ma)
    run_a(a, f"alone, CU mask {ma}")
    a.close()
    a = make(1000, 10 + F, mask=ma)
    b = make(2000, 12, mask=mb, width=1280, height=720)
    stop = False

    def load():
        t = 0
        while not stop:
            b.step_frame(t % 12)
            t += 1

    th = threading.Thread(target=load)
    th.start()
    run_a(a, f"contended, CU masks {ma} | {mb}")
    stop = True
    th.join()
    a.close()
    b.close()
sys.exit(0)
a = make(1000, 10 + F)
b = make(2000, 12, width=1280, height=720)
stop = False


def load():
    t = 0
    while not stop:
        b.step_frame(t % 12)
        t += 1


th = threading.Thread(target=load)
th.start()
run_a(a, "contended")
stop = True
th.join()
