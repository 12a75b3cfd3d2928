"""N > 1 plumbing on CPU: world_size-2 gloo process group.  Covers what bench.py --gpus N relies on (rendezvous,
barrier, max-over-ranks timing, aggregate throughput) and the exchange step of the row-sharded filter
(shard.exchange_rows, the function the RCCL callback runs, on uneven row blocks) plus the storage rule of the sharded
downdate: camera rows replicated, feature rows owned."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from openekfmonoslam_amd import dist, engine

    r = dist.Ranks(backend="gloo")
    assert r.world == 2
    # timing contract: barrier, K "steps", barrier, max over ranks
    r.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (r.rank + 1))
    el = time.perf_counter() - t0
    r.barrier()
    mx = r.max_over_ranks(el)
    assert mx >= 0.1 - 1e-3 and mx >= el
    assert r.sum_over_ranks(1.0) == 2.0
    # row partition: python helper == C ABI
    N = 7
    lo, hi = dist.shard_rows(N, r.world, r.rank)
    assert (lo, hi) == engine.shard_rows(N, r.world, r.rank)
    # the exchange the sharded engine asks for (shard.exchange_rows is the function the RCCL callback runs): every
    # rank has written the H.P rows of the features it owns (2 rows per feature, ld elements each) and ends up with
    # the whole table
    import torch
    import torch.distributed as tdist
    from openekfmonoslam_amd import shard
    rng = np.random.default_rng(5)
    n, ld = 13 + 6 * N, 64
    HP_true = rng.standard_normal((2 * N, ld))
    per, extra = divmod(N, r.world)
    fb = [q * per + min(q, extra) for q in range(r.world + 1)]
    row_begin = [2 * f for f in fb]
    table = np.full((2 * N, ld), np.nan)
    table[row_begin[r.rank]:row_begin[r.rank + 1]] = HP_true[row_begin[r.rank]:row_begin[r.rank + 1]]
    buf = torch.from_numpy(table).view(torch.uint8).reshape(-1)
    shard.exchange_rows(tdist, buf, ld * 8, row_begin, r.rank)
    np.testing.assert_array_equal(table, HP_true)
    # sharded downdate P <- P - B'B with the engine's storage: every rank keeps the 13 camera rows and the rows of its
    # own features, downdates those with the replicated B, and the assembled result equals the unsharded downdate
    m = 10
    A = rng.standard_normal((n, n)); P = A @ A.T
    B = rng.standard_normal((m, n))
    own0, own1 = max(lo, 13), hi
    cam = P[0:13] - B[:, 0:13].T @ B
    mine = P[own0:own1] - B[:, own0:own1].T @ B
    blocks = r.all_gather_rows(np.ascontiguousarray(mine), n - 13)
    full = np.concatenate([cam, blocks])
    np.testing.assert_allclose(full, P - B.T @ B, rtol=1e-12, atol=1e-12)
    r.close()
    print("rank", r.rank, "ok")
    """
)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_plumbing(tmp_path):
    from openekfmonoslam_amd import build

    build.build_engine()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rank} ok" in o
