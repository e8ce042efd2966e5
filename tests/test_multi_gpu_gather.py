"""CPU / gloo, world_size 2: the sharding + frame-record gather used by bench.py for N > 1 GPUs.  Each rank demodulates
its contiguous channel shard (here with the CPU oracle, there is no GPU in this container), numbers its channels with
the shard offset, and the gathered set must equal a single-process run over all channels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from m17hip import dist as mdist

C_TOTAL, T = 6, 24000


def _signals():
    p = ol.gen_params(seed=99, kind=-1, n_frames=7, lead_in=3072, noise_sigma=400.0, tail_sigma=400.0, lead_sigma=40000.0, total=T)
    return ol.generate_batch(p, C_TOTAL, T, threads=2)


def _records(x, chan0):
    recs, counts, _ = ol.demod_batch(x, cap=64, threads=2)
    flat = np.concatenate([recs[c, : counts[c]] for c in range(x.shape[0])]) if counts.sum() else recs[0, :0]
    flat = flat.copy()
    flat["channel"] += chan0
    return flat


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = _signals()
    lo, hi = mdist.shard_range(C_TOTAL, rank, world)
    mine = _records(x[lo:hi], lo)
    buf = torch.zeros(4096 * 64, dtype=torch.uint8)
    buf[: mine.size * 64] = torch.from_numpy(np.frombuffer(mine.tobytes(), dtype=np.uint8).copy())
    allrecs, counts = mdist.gather_records(buf, mine.size)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), allrecs.numpy())
    np.save(os.path.join(outdir, f"counts{rank}.npy"), np.array(counts))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_all_channels():
    for total in (1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_two_rank_gather_equals_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    expect = _records(_signals(), 0)
    assert expect.size > 0
    for rank in range(2):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npy"))
        assert got.tobytes() == expect.tobytes(), rank                     # every rank holds the full, ordered set
        counts = np.load(os.path.join(tmp_path, f"counts{rank}.npy"))
        assert counts.sum() == expect.size and len(counts) == 2


def _worker8(rank, world, port, outdir):
    """world_size 8 over 5 channels: ranks 5..7 hold an EMPTY shard (zero records), rank 3's channel is noise only (zero records
    from a non-empty shard), the others differ in length — the padded all-gather of m17hip/dist.py must still return the one ordered set."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = _signals8()
    lo, hi = mdist.shard_range(x.shape[0], rank, world)
    mine = _records(x[lo:hi], lo) if hi > lo else np.zeros(0, dtype=ol.FRAME_REC)
    buf = torch.zeros(max(1, mine.size) * 64, dtype=torch.uint8)
    if mine.size:
        buf[: mine.size * 64] = torch.from_numpy(np.frombuffer(mine.tobytes(), dtype=np.uint8).copy())
    allrecs, counts = mdist.gather_records(buf, mine.size)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), allrecs.numpy())
    np.save(os.path.join(outdir, f"counts{rank}.npy"), np.array(counts))
    dist.barrier()
    dist.destroy_process_group()


def _signals8():
    rows = []
    for c, (kind, frames) in enumerate([(0, 9), (1, 3), (2, 6), (3, 1), (1, 8)]):   # BERT, short stream, packets, NOISE ONLY, stream
        p = ol.gen_params(seed=300 + c, kind=kind, n_frames=frames, lead_in=3072, noise_sigma=400.0, tail_sigma=400.0, lead_sigma=40000.0, total=T)
        rows.append(ol.generate(p)[:T])
    return np.stack(rows)


def test_eight_rank_gather_with_uneven_and_empty_shards(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    expect = _records(_signals8(), 0)
    assert expect.size > 0 and not (expect["channel"] == 3).any()         # the noise-only channel produced nothing
    for rank in range(8):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npy"))
        counts = np.load(os.path.join(tmp_path, f"counts{rank}.npy"))
        assert got.tobytes() == expect.tobytes(), rank
        assert len(counts) == 8 and counts.sum() == expect.size and counts[3] == 0 and (counts[5:] == 0).all()
