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


def blend_step_units_sharded(latents, predict_half, combine, chunk_size, overlap_size, n_chunks, rng, group=None):
    """The step with (window, CFG half) UNITS sharded over the ranks (round 4): the shipped 100-frame job has only 3 blending windows, so window
    sharding (blend_step_sharded) leaves 5 of 8 GPUs idle; the two CFG halves of a window's UNet evaluation never interact (per-sample norms and
    attention, pipeline_i2vgen_xl.py:851-874: one batched call on cat([latents] * 2)), which makes 2 * n_chunks independent units per DDIM step.

    predict_half(idx, half, window) -> raw UNet prediction of CFG half `half` (0 = unconditional, 1 = text) for window idx.  CONTRACT (asserted
    on every rank that holds a unit): frames-major [chunk, C, H, W] in the latents' dtype -- a rank WITHOUT a unit (world > 2 * n_chunks) sizes its
    all-gather buffer from the window alone, and mismatched buffers would hang or corrupt the collective instead of raising;
    combine(idx, window, pred_uncond, pred_text) -> the window after guidance + scheduler step.
    Unit u = 2 * idx + half goes to rank u % world; ONE all-gather of the predictions per step (8.75 MB per unit at the shipped sizes), then
    every rank applies guidance + DDIM (a 9-MB element-wise kernel per window) and the overwrites in window order: all ranks end with the
    full latents, bit-identical to the single-process loop whenever a half evaluated alone equals its half of the batched evaluation."""
    import torch.distributed as dist
    from . import parallel
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    starts = chunk_starts(latents.shape[2], chunk_size, overlap_size, n_chunks)
    offsets = draw_offsets(n_chunks, overlap_size, rng)          # identical on all ranks (same generator state)
    n_units = 2 * n_chunks
    per_rank = (n_units + world - 1) // world
    windows = [latents[:, :, s:s + chunk_size] for s in starts]
    w0 = windows[0]                                               # prediction of a window [B, C, chunk, H, W]: frames-major [chunk, C, H, W]
    want = (w0.shape[2], w0.shape[1]) + tuple(w0.shape[3:])
    mine = []
    for slot in range(per_rank):
        u = slot * world + rank
        if u < n_units:
            p = predict_half(u // 2, u % 2, windows[u // 2]).contiguous()
            if tuple(p.shape) != want or p.dtype != latents.dtype:
                raise ValueError(f"predict_half must return [chunk, C, H, W] = {want} in {latents.dtype} (the all-gather buffers of every rank are "
                                 f"sized from it), got {tuple(p.shape)} {p.dtype}")
            mine.append(p)
        else:
            mine.append(None)
    proto = torch.zeros(want, dtype=latents.dtype, device=latents.device)   # ranks without a unit (world > 2 * n_chunks) join the collective with the same buffer
    send = torch.stack([m if m is not None else torch.zeros_like(proto) for m in mine], 0)      # [per_rank, ...]
    recv = [torch.empty_like(send) for _ in range(world)]
    parallel.all_gather(recv, send, group=group)
    pred = lambda u: recv[u % world][u // world]
    outs = [combine(idx, windows[idx], pred(2 * idx), pred(2 * idx + 1)) for idx in range(n_chunks)]
    return _apply(torch.empty_like(latents), outs, starts, offsets, chunk_size)
