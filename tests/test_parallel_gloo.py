"""N > 1 path on CPU: two processes, world_size 2.  commpy_amd.parallel is torch-free; its shard arithmetic
(shard_bounds, padding to equal shards, un-padding of the gathered result), sharded_decode and the counter reduction
are exercised here with a gloo-backed stand-in for the RCCL communicator (same Collective protocol: rank, world,
allgather_rows, allreduce) and with the CPU oracle standing in for the per-rank decoder (tests may use the oracle; the
product path on GPUs passes the HIP decoders and RankComm instead).  The file-based exchange of the RCCL id between the
ranks of a one-process-per-GPU launch is tested with two real processes."""
import os
import socket
import sys

import numpy as np
import pytest

from commpy_amd.parallel import pad_shard, shard_bounds, shard_counts, unpad_gathered

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 64, 65536, 262144 + 3):
        for w in (1, 2, 3, 4, 8):
            counts = shard_counts(n, w)
            assert sum(counts) == n and max(counts) - min(counts) <= 1
            edges = [shard_bounds(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_pad_unpad_round_trip():
    rs = np.random.RandomState(0)
    for n, w in ((11, 2), (10, 2), (3, 8), (262147, 8), (1, 4)):
        full = rs.randint(0, 255, (n, 3)).astype(np.uint8)
        shards = [pad_shard(full[slice(*shard_bounds(n, r, w))], n, r, w) for r in range(w)]
        assert len({s.shape for s in shards}) == 1
        assert np.array_equal(unpad_gathered(np.concatenate(shards), n, w), full)
    with pytest.raises(ValueError):
        pad_shard(np.zeros((3, 2)), 11, 0, 2)


class GlooCollective:
    """Stand-in for commpy_amd.parallel.RankComm on CPU (test infrastructure)."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_rows(self, local, n_total):
        import torch
        padded = pad_shard(local, n_total, self.rank, self.world)
        t = torch.from_numpy(padded)
        full = torch.empty((self.world * padded.shape[0],) + padded.shape[1:], dtype=t.dtype)
        self.dist.all_gather_into_tensor(full, t)
        return unpad_gathered(full.numpy(), n_total, self.world)

    def allreduce(self, arr, op='sum'):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == 'sum' else self.dist.ReduceOp.MAX)
        return t.numpy()


def _worker(rank, world, port, q, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle
    from helpers import make_trellis
    from commpy_amd.channelcoding.convcode import conv_encode
    from commpy_amd.parallel import exchange_unique_id, reduce_counters, sharded_decode
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = GlooCollective(dist)
        tr = make_trellis("t57")
        rs = np.random.RandomState(7)
        B = 11                                            # ragged: 6 + 5
        msgs = rs.randint(0, 2, (B, 40))
        coded = np.stack([conv_encode(m, tr) for m in msgs]).astype(float)
        rx = np.where(rs.rand(*coded.shape) < 0.04, 1 - coded, coded)
        dec = lambda x: oracle.viterbi_decode(x, tr, None, "hard").astype(np.uint8)
        full = sharded_decode(dec, [rx], comm)
        want = oracle.viterbi_decode(rx, tr, None, "hard").astype(np.uint8)
        ok = full.shape == want.shape and np.array_equal(full, want)
        # collective-free form: every rank keeps exactly its own rows
        mine = sharded_decode(dec, [rx], comm, gather=False)
        a, b = shard_bounds(B, rank, world)
        ok = ok and mine.shape[0] == b - a and np.array_equal(mine, want[a:b])
        # equal shards (no padding branch) + float payload
        even = comm.allgather_rows(np.arange(40, dtype=np.float64).reshape(10, 4)[rank * 5:(rank + 1) * 5], 10)
        ok = ok and np.array_equal(even, np.arange(40, dtype=np.float64).reshape(10, 4))
        # config-5 counters: per-rank bit errors / bits sent per SNR point, summed over the ranks (links.py:252-260)
        errs = (mine[:, :40] != msgs[a:b]).sum()
        tot = reduce_counters(np.array([errs, (b - a) * 40, rank + 1], dtype=np.int64), comm)
        ok = ok and tot[0] == (want[:, :40] != msgs).sum() and tot[1] == B * 40 and tot[2] == sum(range(1, world + 1))
        ok = ok and np.array_equal(reduce_counters([3, 4]), [3, 4])                   # single process: identity
        # RCCL id exchange of a one-process-per-GPU launch: rank 0 publishes, the others read the same bytes
        blob = exchange_unique_id(rank, world, lambda: bytes(range(128)), path=os.path.join(tmpdir, "id"), timeout=60)
        ok = ok and blob == bytes(range(128))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_decode_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # a stale id file of a crashed earlier launch sits where this launch will publish its id: it must never be read
    stale = tmp_path / "id"
    stale.write_bytes(b"\xff" * 128)
    os.utime(stale, (1.0e9, 1.0e9))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_parallel_module_is_torch_free():
    """north_star: no PyTorch in the product -- the collective layer is the engine's own RCCL binding."""
    import ast
    for mod in ("parallel.py", "_lib.py", "devicelink.py"):
        tree = ast.parse(open(os.path.join(ROOT, "commpy_amd", mod)).read())
        names = [a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names]
        names += [n.module or "" for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)]
        assert not [n for n in names if n.split(".")[0] == "torch"], (mod, names)


def test_default_id_path_is_per_launch():
    """Ranks of one launcher share their parent pid; two launches (or ports) never see each other's id file."""
    from commpy_amd.parallel import exchange_unique_id
    assert exchange_unique_id(0, 1, lambda: b"x" * 128) == b"x" * 128                # world 1: no file at all
    from commpy_amd import parallel
    assert parallel._default_id_path(0) != parallel._default_id_path(1)              # one file per communicator of a job
