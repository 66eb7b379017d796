"""CPU restatement of the randomized-blending step -- TEST INFRASTRUCTURE (see oracle/svd_oracle.py header).

Follows code/i2v_enhance/pipeline_i2vgen_xl.py:841-909 line by line (the running CHUNK_START, the per-chunk
random.randint draw, the overwrite), with the UNet+CFG+scheduler.step of a window abstracted as `denoise_chunk`.
Index arithmetic only: results must be bit-identical to the product path."""
import torch


def randomized_blending_step(latents, denoise_chunk, chunk_size, overlap_size, n_chunks, rng):
    latents_denoised = torch.empty_like(latents)
    CHUNK_START = 0
    for idx in range(n_chunks):
        latents_chunk = denoise_chunk(idx, latents[:, :, CHUNK_START:CHUNK_START + chunk_size])
        if CHUNK_START == 0:
            random_offset = 0
        else:
            random_offset = rng.randint(0, overlap_size - 1) if overlap_size != 0 else 0
        latents_denoised[:, :, CHUNK_START + random_offset:CHUNK_START + chunk_size] = latents_chunk[:, :, random_offset:]
        CHUNK_START += chunk_size - overlap_size
    if CHUNK_START + overlap_size > latents_denoised.shape[2]:
        raise NotImplementedError("not dividable into chunks")
    return latents_denoised
