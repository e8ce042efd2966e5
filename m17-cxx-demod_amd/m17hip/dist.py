"""Multi-GPU plumbing for the demodulation path: channels are independent (reference: one M17Demodulator object per
channel, no shared state), so a node is sharded by contiguous channel ranges, one process per GPU, with NO data-path
collective.  The only exchange is the gather of the decoded frame records (64-byte PODs, include/m17hip.h) at the end
of a run — an RCCL all_gather over xGMI on GPUs (backend "nccl"), gloo on CPU for the tests.  Volume is tiny
(~25 records/s/channel), so one padded all_gather is enough; no ring tuning is needed."""
import torch
import torch.distributed as dist

REC_BYTES = 64


def shard_range(total_channels, rank, world):
    """Contiguous channel range [lo, hi) of `rank` (GPU g gets channels [g*C/8, (g+1)*C/8))."""
    per = (total_channels + world - 1) // world
    lo = min(total_channels, rank * per)
    return lo, min(total_channels, lo + per)


def gather_records(local, n_local, group=None):
    """All-gather the first `n_local` records of `local` (uint8 tensor, >= n_local*64 bytes, CPU or GPU).
    Returns (records [sum(n), 64] uint8 on the same device, counts list).  Records keep their per-rank order, ranks
    are concatenated in rank order, so with channel-major local order the result is globally (channel, seq) ordered
    when each rank numbered its channels with its shard offset."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    flat = local.reshape(-1)
    if world == 1:
        return flat[: n_local * REC_BYTES].reshape(-1, REC_BYTES), [n_local]
    dev = flat.device
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if nmax == 0:
        return torch.empty((0, REC_BYTES), dtype=torch.uint8, device=dev), counts
    mine = torch.zeros(nmax * REC_BYTES, dtype=torch.uint8, device=dev)
    mine[: n_local * REC_BYTES] = flat[: n_local * REC_BYTES]
    out = torch.empty(world * nmax * REC_BYTES, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.reshape(world, nmax, REC_BYTES)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0), counts
