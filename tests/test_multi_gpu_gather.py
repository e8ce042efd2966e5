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
