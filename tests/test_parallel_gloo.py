"""N > 1 path on CPU: two processes, gloo backend, world_size 2.  The sharding + all-gather logic of
commpy_amd.parallel is exercised with the CPU oracle standing in for the per-rank decoder (tests may
use the oracle; the product path on GPUs passes the HIP decoders instead)."""
import os
import socket
import sys

import numpy as np
import pytest

from commpy_amd.parallel import shard_bounds, shard_counts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 64, 65536, 262144 + 3):
        for w in (1, 2, 3, 4, 8):
            counts = shard_counts(n, w)
            assert sum(counts) == n and max(counts) - min(counts) <= 1
            edges = [shard_bounds(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle
    from helpers import make_trellis
    from commpy_amd.channelcoding.convcode import conv_encode
    from commpy_amd.parallel import all_gather_rows, sharded_decode
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr = make_trellis("t57")
        rs = np.random.RandomState(7)
        B = 11                                            # ragged: 6 + 5
        msgs = rs.randint(0, 2, (B, 40))
        coded = np.stack([conv_encode(m, tr) for m in msgs]).astype(float)
        rx = np.where(rs.rand(*coded.shape) < 0.04, 1 - coded, coded)
        full = sharded_decode(lambda x: oracle.viterbi_decode(x, tr, None, "hard").astype(np.uint8), [rx])
        want = oracle.viterbi_decode(rx, tr, None, "hard").astype(np.uint8)
        ok = full.shape == want.shape and np.array_equal(full, want)
        # collective-free form: every rank keeps exactly its own rows
        from commpy_amd.parallel import shard_bounds
        mine = sharded_decode(lambda x: oracle.viterbi_decode(x, tr, None, "hard").astype(np.uint8), [rx], gather=False)
        a, b = shard_bounds(B, rank, world)
        ok = ok and mine.shape[0] == b - a and np.array_equal(mine, want[a:b])
        # equal shards (no padding branch) + float payload
        lo, hi = B // world * rank, B // world * (rank + 1)
        even = all_gather_rows(np.arange(40, dtype=np.float64).reshape(10, 4)[rank * 5:(rank + 1) * 5], 10)
        ok = ok and np.array_equal(even, np.arange(40, dtype=np.float64).reshape(10, 4))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_decode_world2_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
