"""N>1 path on CPU: channel sharding is a partition, and the one exchange step (broadcast of the
IQ block from rank 0) works across processes (gloo, world_size 2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sigdigger_amd import pipeline, synth


def test_shard_channels_is_a_partition():
    fn = synth.raster(512, 0.0036)
    for world in (1, 2, 4, 8):
        parts = [pipeline.shard_channels(fn, r, world) for r in range(world)]
        assert sum(len(p) for p in parts) == 512
        assert all(len(p) == 512 // world for p in parts)
        merged = np.empty(512)
        for r, p in enumerate(parts):
            merged[r::world] = p
        assert np.array_equal(merged, fn)
        for c in (0, 1, 63, 511):
            r, i = pipeline.channel_owner(c, world)
            assert parts[r][i] == fn[c]


def test_shard_frames_is_a_partition():
    """C5 (BASELINE.json configs[4]) shards by frame: dwell f of the sweep on rank f mod G, nothing exchanged"""
    for nframes in (512, 513, 7):
        for world in (1, 2, 4, 8):
            parts = [pipeline.shard_frames(nframes, r, world) for r in range(world)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(nframes))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(np.all(p % world == r) for r, p in enumerate(parts))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << 14
        ref = torch.from_numpy(synth.tone_noise(n, seed=5))
        bufs = [ref.clone() if rank == 0 else torch.zeros(n, dtype=torch.complex64) for _ in range(2)]
        ok = True
        for k in range(4):                                    # double-buffered, as bench.py does
            if rank == 0:
                bufs[(k + 1) & 1].copy_(ref * (k + 2))
            w = pipeline.broadcast_block(torch.view_as_real(bufs[(k + 1) & 1]), dist)
            assert w is not None
            w.wait()
            ok = ok and torch.equal(bufs[(k + 1) & 1], ref * (k + 2))
        fn = synth.raster(16, 0.1)
        mine = pipeline.shard_channels(fn, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine.tolist())
        q.put((rank, ok, gathered))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fn = synth.raster(16, 0.1)
    for rank, ok, gathered in res:
        assert ok, f"rank {rank}: broadcast payload mismatch"
        flat = sorted(v for part in gathered for v in part)
        assert np.allclose(flat, sorted(fn))


def test_single_process_broadcast_is_a_noop():
    assert pipeline.broadcast_block(torch.zeros(4), None) is None
