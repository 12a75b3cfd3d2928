"""The one-process-per-rank form of the row-sharded filter on the 1-GPU box: two processes share cuda:0, rendezvous
through torch.distributed (gloo here, because RCCL refuses two ranks on one device; the exchange code path --
engine callback -> shard.DistributedExchange -> dist.broadcast on tensors aliasing the engine buffers -- is the one
`bench.py --mode sharded` uses with backend nccl).  Rank 0 also runs the unsharded engine and compares."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from openekfmonoslam_amd import engine
    from openekfmonoslam_amd.shard import DistributedExchange
    from openekfmonoslam_amd.synth import SyntheticSequence

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    frames, nfeat, precision = 3, 60, int(os.environ["EKF_PRECISION"])
    seq = SyntheticSequence(nfeat, frames)
    P0 = 0.5 * (seq.P0 + seq.P0.T)
    e = engine.EkfEngine(seq.cam, seq.par, nfeat, max_keypoints=4 * nfeat + 64, precision=precision, device=0,
                         shard=(rank, world))
    e.set_exchange(DistributedExchange(dist, dev))
    e.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
    infos = [e.step(*seq.frames[t]) for t in range(frames)]
    n = e.n
    P = np.zeros((n, n))
    x, fp, _ = e.get_state(P_out=P)               # fills the rows this rank holds
    r, w, lo, hi = e.shard_info()
    own = np.zeros(n, dtype=bool); own[max(lo, 13):hi] = True
    Pt = torch.from_numpy(np.where(own[:, None], P, 0.0))
    dist.all_reduce(Pt)                            # owned rows are disjoint: the sum assembles the matrix
    Pfull = Pt.numpy()
    Pfull[:13] = P[:13]                            # camera rows: replicated
    xt = torch.from_numpy(x.copy()); dist.broadcast(xt, src=0)
    assert np.array_equal(xt.numpy(), x), "replicated state differs between ranks"
    if rank == 0:
        ref = engine.EkfEngine(seq.cam, seq.par, nfeat, max_keypoints=4 * nfeat + 64, precision=precision, device=0)
        ref.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
        rinfos = [ref.step(*seq.frames[t]) for t in range(frames)]
        for a, b in zip(infos, rinfos):
            assert (a.n_matches, a.n_inliers, a.n_rescued, a.status) == (b.n_matches, b.n_inliers, b.n_rescued, b.status)
        xr, fpr, Pr = ref.get_state()
        # (precision 2: each rank cuts ITS columns of B into digit planes, the unsharded engine all of them in one sweep: the
        # same quantisation, different summation order of the fp64 rows of B -> 1e-6 does not always hold, 1e-5 does)
        tol = 1e-12 if precision == 0 else (1e-6 if precision == 1 else 1e-5)
        assert np.abs(x - xr).max() <= tol * max(1.0, np.abs(xr).max())
        assert np.abs(Pfull - Pr).max() / np.abs(Pr).max() <= tol
        assert np.array_equal(Pfull, Pfull.T)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
    """
)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_two_processes_one_filter(tmp_path, precision):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), EKF_PRECISION=str(precision), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert f"rank {rank} ok" in o


def test_bench_sharded_mode_two_ranks_one_gpu(tmp_path):
    """bench.py --mode sharded end to end with two ranks on cuda:0 (gloo process group, so the host-callback transport:
    RCCL refuses two ranks on one device): one JSON line from rank 0, scaling "strong", the same decisions as the
    unsharded run, and the one-GPU reference of the same workload"""
    import json

    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), EKF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "sharded",
                                       "--workload", "n200_f64", "--steps", "4", "--warmup", "2", "--no-roofline-pass"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "row-sharded" in d["config"]["parallelism"] and "host callback" in d["config"]["parallelism"]
    assert d["single_gpu_same_workload"]["value"] > 0
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # only rank 0 prints
    ref = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "n200_f64", "--steps", "4",
                          "--warmup", "2", "--no-roofline-pass", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    r = json.loads([ln for ln in ref.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("mean_matches", "mean_li_inliers", "mean_rescued"):
        assert d["config"][k] == r["config"][k]
