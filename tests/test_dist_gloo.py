"""N > 1 path on CPU: world_size-2 (and 3, uneven) gloo processes shard the channels,
compute their spectrogram tile (with the oracle standing in for the GPU) and all-gather
it; every rank must end up with the full-tile result."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, channels, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from audian_amd.dist import shard_channels, allgather_tiles, tile_frames
    from oracle import oracle
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rate, nfft, hop, T = 8000.0, 64, 32, 2000
        rng = np.random.default_rng(42)
        x = rng.uniform(-1, 1, size=(T, channels))          # same recording on every rank
        nd = tile_frames((T + hop - 1)//hop, rate, hop, 0.2)
        c0, c1 = shard_channels(channels, rank, world)
        local = np.zeros((nd, c1 - c0, nfft//2 + 1))
        oracle.spectrogram_process(x[:, c0:c1], local, rate, nfft, hop)
        tile = torch.from_numpy(np.ascontiguousarray(local.transpose(1, 0, 2)))   # (C_local, T', F)
        merged = allgather_tiles(tile, channels).numpy()
        full = np.zeros((nd, channels, nfft//2 + 1))
        oracle.spectrogram_process(x, full, rate, nfft, hop)
        ok = merged.shape == (channels, nd, nfft//2 + 1) and \
            np.array_equal(merged, full.transpose(1, 0, 2))
        q.put((rank, bool(ok), (c0, c1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,channels', [(2, 6), (3, 7)])
def test_channel_shard_and_allgather(world, channels):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, channels, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    spans = sorted(s for _, _, s in results)
    assert spans[0][0] == 0 and spans[-1][1] == channels
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_shard_channels_partition():
    from audian_amd.dist import shard_channels
    for channels in (1, 5, 64, 256, 7):
        for world in (1, 2, 3, 8):
            spans = [shard_channels(channels, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == channels
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_channels(4, 2, 2)
