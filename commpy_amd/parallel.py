"""Multi-GPU batch sharding for the decoders (one process per GPU).

Codewords are independent, so the path shards with NO data-path collective: rank r decodes the
contiguous slice ``shard_bounds(B, r, world)`` of the batch on its own GPU.  The only collective is
one all-gather of the decoded bits (uint8) so that every rank holds the full ``[B, L]`` result --
``torch.distributed`` is used purely as launcher/collective plumbing (backend "nccl" = RCCL over
xGMI on MI355X, "gloo" for the CPU tests); the decoders themselves never touch torch.

The reference has no distributed code at all; this module is new work.
"""
import numpy as np

__all__ = ['shard_bounds', 'shard_counts', 'all_gather_rows', 'sharded_decode']


def shard_counts(n_items, world_size):
    """Rows per rank: contiguous blocks, the first ``n_items % world_size`` ranks get one more."""
    base, rem = divmod(int(n_items), int(world_size))
    return [base + (1 if r < rem else 0) for r in range(world_size)]


def shard_bounds(n_items, rank, world_size):
    """``(start, stop)`` of rank's contiguous slice of a batch of ``n_items`` codewords."""
    counts = shard_counts(n_items, world_size)
    start = sum(counts[:rank])
    return start, start + counts[rank]


def _dist():
    import torch.distributed as dist
    return dist


def all_gather_rows(local, n_total, group=None):
    """All-gather row shards (``shard_bounds`` layout) into the full ``[n_total, ...]`` array on every rank.

    ``local`` is a NumPy array (gathered through CPU tensors: gloo) or a torch tensor (CPU or GPU;
    on GPU the transfer is a single RCCL all-gather of equal-size padded shards).  Without an
    initialised process group (single process) the input is returned unchanged.
    """
    import torch
    dist = _dist()
    is_np = isinstance(local, np.ndarray)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if t.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError('local shard has %d rows, expected %d' % (t.shape[0], counts[dist.get_rank(group)]))
    pad_rows = max(counts)
    if t.shape[0] < pad_rows:                                  # equal-size shards for one fused collective
        pad = torch.zeros((pad_rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], 0)
    full = torch.empty((world * pad_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(full, t, group=group)
    if len(set(counts)) != 1:                                  # drop the padding rows
        full = torch.cat([full[r * pad_rows:r * pad_rows + counts[r]] for r in range(world)], 0)
    return full.numpy() if is_np else full


def sharded_decode(decode_fn, batch_inputs, n_total=None, group=None, gather=True):
    """Decode a batch sharded over the ranks of the process group.

    The decode path has no exchange step: with ``gather=False`` every rank returns only the rows
    ``shard_bounds(n_total, rank, world)`` it decoded (no collective at all).  ``gather=True`` (default, the
    convenient form for scripts that continue on every rank) reassembles the full result with one all-gather.

    ``decode_fn(*shard_inputs) -> ndarray [rows, ...]`` is any of the batched decoders (e.g.
    ``lambda x: viterbi_decode(x, trellis, None, 'soft')``); ``batch_inputs`` are arrays whose first
    axis is the codeword index (every rank passes the same full arrays, or arrays it can slice).
    """
    dist = _dist()
    n_total = int(batch_inputs[0].shape[0]) if n_total is None else int(n_total)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n_total, rank, world)
    local = np.asarray(decode_fn(*[a[lo:hi] for a in batch_inputs]))
    if not gather:
        return local
    return all_gather_rows(local, n_total, group)
