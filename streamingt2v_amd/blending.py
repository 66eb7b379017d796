"""Randomized blending of the enhancement stage (I2VGen-XL SDEdit), single-GPU and chunk-sharded over ranks.

Mirrors the per-step chunk loop of code/i2v_enhance/pipeline_i2vgen_xl.py:841-909:

    latents_denoised = empty_like(latents)
    for idx, chunk in enumerate(chunks):                    # windows of `chunk_size` frames, stride chunk_size-overlap
        out = denoise(chunk(latents))                       # UNet + CFG + scheduler.step on the window
        r   = 0 if first chunk else random.randint(0, overlap-1)
        latents_denoised[:, :, start+r : start+chunk_size] = out[:, :, r:]          # OVERWRITE, not averaging
    latents = latents_denoised

Every window reads `latents` of the step start, so the windows of one step are independent: this is the part of the
StreamingT2V pipeline that shards across GPUs.  `blend_step_sharded` assigns windows to ranks
round-robin, all-gathers the window outputs (RCCL over xGMI on the node; gloo in the CPU tests) and applies the
overwrites in window order on every rank with identical offsets -- bit-identical to the single-process loop.
The random offsets come from a `random.Random` the caller seeds (the reference uses the global `random` module seeded by
Lightning's seed_everything: 33, config.yaml:2); all ranks must pass generators in the same state.
"""
import random

import torch


def chunk_starts(num_frames, chunk_size, overlap_size, n_chunks):
    """Window start frames; raises like the reference when the video is not divisible into windows
    (pipeline_i2vgen_xl.py:907-909)."""
    starts = [i * (chunk_size - overlap_size) for i in range(n_chunks)]
    end = n_chunks * (chunk_size - overlap_size)
    if end + overlap_size > num_frames:
        raise NotImplementedError(f"Video of size={num_frames} is not dividable into chunks "
                                  f"with size={chunk_size} and overlap={overlap_size}")
    return starts


def draw_offsets(n_chunks, overlap_size, rng):
    """random_offset per window, in window order (pipeline_i2vgen_xl.py:891-898)."""
    offs = []
    for idx in range(n_chunks):
        if idx == 0 or overlap_size == 0:
            offs.append(0)
        else:
            offs.append(rng.randint(0, overlap_size - 1))
    return offs


def _apply(latents_denoised, outs, starts, offsets, chunk_size):
    for out, start, r in zip(outs, starts, offsets):
        latents_denoised[:, :, start + r:start + chunk_size] = out[:, :, r:]
    return latents_denoised


def blend_step(latents, denoise_chunk, chunk_size, overlap_size, n_chunks, rng=random):
    """One denoising step over all windows on one device.  latents [B, C, F, H, W];
    denoise_chunk(idx, window[B,C,chunk,H,W]) -> window of the same shape."""
    starts = chunk_starts(latents.shape[2], chunk_size, overlap_size, n_chunks)
    offsets = draw_offsets(n_chunks, overlap_size, rng)
    outs = [denoise_chunk(idx, latents[:, :, s:s + chunk_size]) for idx, s in enumerate(starts)]
    return _apply(torch.empty_like(latents), outs, starts, offsets, chunk_size)


def blend_step_sharded(latents, denoise_chunk, chunk_size, overlap_size, n_chunks, rng, group=None):
    """Same step with the windows sharded round-robin over the ranks of `group` and one all-gather of the window
    outputs (each 4 x chunk x 90 x 160 fp16 = 4.4 MB at the shipped sizes).  Every rank returns the full latents."""
    import torch.distributed as dist
    from . import parallel
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    starts = chunk_starts(latents.shape[2], chunk_size, overlap_size, n_chunks)
    offsets = draw_offsets(n_chunks, overlap_size, rng)          # identical on all ranks (same generator state)
    per_rank = (n_chunks + world - 1) // world
    mine = []
    for slot in range(per_rank):
        idx = slot * world + rank
        if idx < n_chunks:
            mine.append(denoise_chunk(idx, latents[:, :, starts[idx]:starts[idx] + chunk_size]).contiguous())
        else:                                                      # pad so that every rank contributes equally
            mine.append(torch.zeros_like(latents[:, :, :chunk_size]).contiguous())
    send = torch.stack(mine, 0)                                    # [per_rank, B, C, chunk, H, W]
    recv = [torch.empty_like(send) for _ in range(world)]
    parallel.all_gather(recv, send, group=group)
    outs = [recv[idx % world][idx // world] for idx in range(n_chunks)]
    return _apply(torch.empty_like(latents), outs, starts, offsets, chunk_size)
