"""Channel-sharded multi-GPU execution: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for CPU tests).

Channels are fully independent in every stage of the hot path (the reference loops per
channel, ``src/audian/bufferedfilter.py:35``), so each rank owns a contiguous block of
channels of the planar (channels, frames) layout and no data-path exchange happens
during compute.  The one collective is the all-gather of each rank's spectrogram tile
(channels_local, frames', F) into the merged (channels, frames', F) tile that every
rank's display needs; because the device layout is channel-major the per-rank chunks
are contiguous and the gather needs no re-layout.

Import order matters in a process that uses both: ``import torch`` must come before
``audian_amd.hipdsp`` (see the note in ``_lib.py``); with ``WORLD_SIZE > 1`` in the
environment ``_lib`` does that itself.
"""


def shard_channels(channels, rank, world):
    """Contiguous channel block [c0, c1) of `rank`; the first `channels % world` ranks
    get one channel more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f'bad rank {rank} of {world}')
    base, extra = divmod(int(channels), int(world))
    c0 = rank*base + min(rank, extra)
    c1 = c0 + base + (1 if rank < extra else 0)
    return c0, c1


def allgather_tiles(local_tile, channels, group=None, out=None, async_op=False):
    """All-gather per-rank tiles (channels_local, ...) along the channel axis.

    `local_tile` is this rank's contiguous torch tensor; `channels` the global channel
    count (so uneven shards can be un-padded).  Returns the merged (channels, ...)
    tensor on every rank.  One collective; equal shards gather straight into the result
    (`out` may be a preallocated result; with `async_op` the call returns
    `(result, work)` and the caller overlaps compute until `work.wait()`).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    c0, c1 = shard_channels(channels, rank, world)
    if local_tile.shape[0] != c1 - c0:
        raise ValueError(f'rank {rank} holds {local_tile.shape[0]} channels, expected {c1 - c0}')
    local_tile = local_tile.contiguous()
    rest = tuple(local_tile.shape[1:])
    if channels % world == 0:
        if out is None:
            out = torch.empty((channels,) + rest, dtype=local_tile.dtype, device=local_tile.device)
        work = dist.all_gather_into_tensor(out, local_tile, group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op or out is not None:
        raise ValueError('async/preallocated gather needs equal channel shards')
    cmax = -(-channels//world)
    padded = torch.zeros((cmax,) + rest, dtype=local_tile.dtype, device=local_tile.device)
    padded[:c1 - c0] = local_tile
    buf = torch.empty((world*cmax,) + rest, dtype=local_tile.dtype, device=local_tile.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for r in range(world):
        a, b = shard_channels(channels, r, world)
        parts.append(buf[r*cmax:r*cmax + (b - a)])
    return torch.cat(parts, dim=0)


def tile_frames(frames_total, rate, hop, tile_seconds):
    """Spectrogram frames in a display tile of `tile_seconds` (the resident window of
    the browser: buffer_time + pre/post-roll, src/audian/data.py:17,168)."""
    import math
    return min(int(frames_total), int(math.ceil(tile_seconds*rate/hop)))
