"""Run by tests/test_gpu_sharded.py in a process of its own (the engine resolves its RCCL table once per process):
`world` ranks of ONE sharded filter as threads on one GPU whose exchange is the engine's IN-STREAM transport -- exchange_rows'
RCCL branch: grouped ncclSend / ncclRecv between every pair of ranks on the engine's stream (csrc/engine.cpp) -- with
tests/cpp/mock_rccl.cpp standing in for librccl.so.1 (EKF_RCCL_LIBRARY).  Prints one JSON line: per-frame decisions of every
rank, the difference to the same filter run through the host-callback transport (LocalShardGroup), and the mock's per-rank
send / receive statistics."""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    N, world, precision, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    assert os.environ.get("EKF_RCCL_LIBRARY"), "EKF_RCCL_LIBRARY must point at the mock"
    from openekfmonoslam_amd import engine
    from openekfmonoslam_amd.shard import LocalShardGroup
    from openekfmonoslam_amd.synth import SyntheticSequence

    seq = SyntheticSequence(N, frames)
    P0 = 0.5 * (seq.P0 + seq.P0.T)
    kw = dict(max_keypoints=4 * N + 64, precision=precision)
    uid = engine.comm_unique_id()
    engs = [engine.EkfEngine(seq.cam, seq.par, N, shard=(r, world), **kw) for r in range(world)]
    infos, errs = [None] * world, [None] * world

    def work(r):
        try:
            e = engs[r]
            e.comm_init(uid)  # no callback installed: the in-engine transport carries every exchange
            e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
            infos[r] = [e.step(*seq.frames[t]) for t in range(frames)]
            e.synchronize()
        except BaseException as ex:  # noqa: BLE001
            errs[r] = repr(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    if any(errs) or any(t.is_alive() for t in th):
        print(json.dumps({"error": errs, "hung": [t.is_alive() for t in th]}))
        os._exit(2)
    n = engs[0].n
    P = np.full((n, n), np.nan)
    for e in engs:
        x, fp, _ = e.get_state(P_out=P)
    # the same filter through the host-callback transport
    grp = LocalShardGroup(seq.cam, seq.par, N, world, **kw)
    grp.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    ref_infos = grp.run(lambda r, e: [e.step(*seq.frames[t]) for t in range(frames)])
    xr, fpr, Pr = grp.get_state()
    fields = ("n_predicted", "n_matches", "n_hypotheses", "n_inliers", "n_outliers", "n_rescued", "status")
    mock = C.CDLL(os.environ["EKF_RCCL_LIBRARY"])
    stats = []
    for r in range(world):
        v = [C.c_longlong(0) for _ in range(5)]
        mism = mock.mock_rccl_stats(r, *[C.byref(a) for a in v])
        stats.append({"sent": v[0].value, "received": v[1].value, "sends": v[2].value, "recvs": v[3].value, "groups": v[4].value,
                      "mismatches": mism})
    out = {
        "decisions": [[[int(getattr(i, f)) for f in fields] for i in infos[r]] for r in range(world)],
        "decisions_callback": [[int(getattr(i, f)) for f in fields] for i in ref_infos[0]],
        "nan_rows": bool(np.isnan(P).any()),
        "P_bitwise_equal_to_callback_transport": bool(np.array_equal(P, Pr)),
        "P_symmetric": bool(np.array_equal(P, P.T)),
        "state_bitwise_equal": bool(np.array_equal(x, xr) and np.array_equal(fp, fpr)),
        "callback_bytes_rank0": int(grp.bytes_exchanged),
        "stats": stats,
    }
    print(json.dumps(out))
    for e in engs:
        e.close()
    grp.close()


if __name__ == "__main__":
    main()
