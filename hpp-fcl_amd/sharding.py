"""Multi-GPU sharding of pair batches: queries are independent, so rank r owns a
contiguous slice of the pair list; the shape/mesh library is replicated; the only collective is an
all-gather of the fixed-size result records (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of rank `rank`: ceil(n/world) pairs per rank (last ranks may be short)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def padded_shard_len(n, world):
    return (n + world - 1) // world


def all_gather_records(local_records, n_total, record_words, dist, device=None):
    """All-gather per-shard result records (torch int32 tensor [padded_len * record_words]) into the
    full batch order; returns a tensor of n_total * record_words int32 on every rank."""
    import torch
    world = dist.get_world_size()
    per = padded_shard_len(n_total, world)
    buf = local_records
    if buf.numel() != per * record_words:
        pad = torch.zeros(per * record_words, dtype=torch.int32, device=buf.device)
        pad[:buf.numel()] = buf
        buf = pad
    out = torch.empty(world * per * record_words, dtype=torch.int32, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return out[:n_total * record_words]


def records_to_words(records):
    """numpy structured record array -> flat int32 view (what travels over the collective)."""
    return np.ascontiguousarray(records).view(np.int32).reshape(-1)
