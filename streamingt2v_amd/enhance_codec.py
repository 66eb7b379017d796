"""Everything around the enhancer's denoising loop: the encoders / decoder of `I2VGenXLPipeline.__call__`
(code/i2v_enhance/pipeline_i2vgen_xl.py:745-835, 915-922) on the MI355X kernels, in the shape `pipeline.StreamingPipeline.enhance_video`
consumes:

    encode_video(frames)            prepare_video_latents :554-603   VAE-encode 16 frames at a time, posterior SAMPLE, x 0.18215
    window_conditioning(images,..)  :753-800                         per window: CLIP image embedding of the centre-cropped key image
                                                                     (zeros for the unconditional half), VAE latents of the image +
                                                                     frame-position-mask planes (:479-511), [negative | positive]
                                                                     prompt embeddings, fps 38 (the fork's `target_fps` default
                                                                     :630, which i2v_enhance_interface.py:95-133 does not override)
    noise_like(latents)             randn_tensor(..., generator)     :605-606
    decode(latents)                 decode_latents :384-406          one frame at a time, (x / 2 + 0.5).clamp(0, 1) -> uint8

Host-side image handling follows the reference's own helpers: `_center_crop_wide` (PIL BOX resize + centre crop, :965-1000) and
`_resize_bilinear` (:952-962) are restated with PIL; normalisation constants are CLIP's.  Prompts go through `clip_tokenizer.CLIPBPETokenizer` (`set_prompts`; the
vocabulary files come with the checkpoint), or the caller passes token ids (`set_prompts_from_ids`) / ready prompt embeddings.
"""
import numpy as np
import torch

from .clip_vision import CLIP_MEAN, CLIP_STD


def center_crop_wide(img, resolution):
    """pipeline_i2vgen_xl.py:965-1000 for one PIL image: BOX-resize so that the image covers `resolution` (w, h), then centre crop."""
    import PIL.Image
    scale = min(img.size[0] / resolution[0], img.size[1] / resolution[1])
    img = img.resize((round(img.width // scale), round(img.height // scale)), resample=PIL.Image.BOX)
    x1, y1 = (img.width - resolution[0]) // 2, (img.height - resolution[1]) // 2
    return img.crop((x1, y1, x1 + resolution[0], y1 + resolution[1]))


def frame_position_planes(image_latents, num_frames):
    """prepare_image_latents :491-503: frame 0 = the image latents, frame i > 0 = a constant plane i / (num_frames - 1) in all channels.
    image_latents [B, 4, h, w] -> [B, 4, num_frames, h, w]."""
    x = image_latents.unsqueeze(2)
    if num_frames > 1:
        planes = [torch.ones_like(x) * ((i + 1) / (num_frames - 1)) for i in range(num_frames - 1)]
        x = torch.cat([x] + planes, 2)
    return x


class EnhanceCodec:
    def __init__(self, vae, image_tower, text_tower=None, height=720, width=1280, target_fps=38, generator=None, device="cuda"):
        """vae: temporal_ae.AutoencoderKL2D; image_tower: clip_vision.OpenCLIPVisionTower (loaded through
        hf_clip_vision_to_openclip_keys); text_tower: clip_text.CLIPTextTower or None when prompt embeddings are given directly."""
        self.vae, self.image_tower, self.text_tower = vae, image_tower, text_tower
        self.h, self.w, self.fps, self.gen, self.dev = height, width, target_fps, generator, device
        self.prompt_embeds = self.negative_prompt_embeds = None

    # ---- text ---------------------------------------------------------------------------------------------------------------
    def set_prompts_from_ids(self, prompt_ids, negative_ids, clip_skip=1):
        """token ids [1, 77] of the prompt / negative prompt (CLIPTokenizer output, padding='max_length'; encode_prompt :250-347).
        clip_skip = 1 is the default of the fork's __call__ (:645; i2v_enhance_interface.py does not pass it): the LAST encoder layer is
        skipped and the final LayerNorm applied to the layer before it."""
        self.prompt_embeds = self.text_tower(prompt_ids, clip_skip=clip_skip)
        self.negative_prompt_embeds = self.text_tower(negative_ids, clip_skip=clip_skip)

    def set_prompts(self, prompt, negative_prompt, tokenizer):
        """prompt strings through `clip_tokenizer.CLIPBPETokenizer` (or any tokenizer with the transformers call signature)."""
        ids = lambda t: torch.tensor([tokenizer(t, padding="max_length", max_length=tokenizer.model_max_length, truncation=True)["input_ids"]],
                                     dtype=torch.long, device=self.dev)
        self.set_prompts_from_ids(ids(prompt), ids(negative_prompt))

    def set_prompt_embeds(self, prompt_embeds, negative_prompt_embeds):
        self.prompt_embeds, self.negative_prompt_embeds = prompt_embeds.to(self.dev).float(), negative_prompt_embeds.to(self.dev).float()

    # ---- images / video ---------------------------------------------------------------------------------------------------
    def _to_pil(self, img):
        import PIL.Image
        return img if isinstance(img, PIL.Image.Image) else PIL.Image.fromarray(np.asarray(img))

    def _pixels(self, pil_images):
        """VaeImageProcessor.preprocess of already correctly sized images: uint8 -> [-1, 1], NCHW."""
        a = np.stack([np.asarray(p.convert("RGB")) for p in pil_images], 0)
        return torch.from_numpy(a).to(self.dev).permute(0, 3, 1, 2).float() / 127.5 - 1.0

    def image_embedding(self, img):
        """:775-783 + _encode_image :349-383: centre crop to a square of side `width`, PIL BILINEAR resize to 224, CLIP normalisation."""
        import PIL.Image
        sq = center_crop_wide(self._to_pil(img), (self.w, self.w)).resize((224, 224), PIL.Image.BILINEAR)
        x = torch.from_numpy(np.array(sq.convert("RGB"))).to(self.dev).permute(2, 0, 1).float()[None] / 255.0
        mean, std = (torch.tensor(v, device=self.dev).view(1, 3, 1, 1) for v in (CLIP_MEAN, CLIP_STD))
        return self.image_tower((x - mean) / std).float()                       # [1, 1024]

    def encode_video(self, frames):
        """frames: sequence of uint8 [H, W, 3] / PIL images (resized to width x height like inference_i2v.py:196-198)."""
        import PIL.Image
        pil = [self._to_pil(f) for f in frames]
        pil = [p if p.size == (self.w, self.h) else p.resize((self.w, self.h)) for p in pil]
        out = []
        n = len(pil) // 16 if len(pil) > 16 else 1                              # torch.chunk(video, F // 16) for long videos (:586-595):
        size = -(-len(pil) // n)                                                # chunks of ceil(F / n) frames, the last one shorter
        for a in range(0, len(pil), size):
            out.append(self.vae.encode_sample(self._pixels(pil[a:a + size]), self.gen))
        lat = torch.cat(out, 0)                                                 # [F, 4, h, w]
        return lat.permute(1, 0, 2, 3)[None].contiguous()                       # [1, 4, F, h, w]

    def noise_like(self, latents):
        """randn_tensor of the FRAME-major shape [F, 4, h, w] the reference draws (:605-606), returned as [1, 4, F, h, w]."""
        b, c, f, h, w = latents.shape
        return torch.randn((b * f, c, h, w), generator=self.gen, device=latents.device).view(b, f, c, h, w).permute(0, 2, 1, 3, 4).contiguous()

    def window_conditioning(self, images, n_windows, window_len):
        """One conditioning dict per blending window (unconditional half first).  images: one key image per window (or a single
        image used for every window)."""
        assert self.prompt_embeds is not None, "set_prompts_from_ids / set_prompt_embeds first"
        text = torch.cat([self.negative_prompt_embeds, self.prompt_embeds], 0)
        conds = []
        for i in range(n_windows):
            img = self._to_pil(images[i] if len(images) > 1 else images[0])
            emb = self.image_embedding(img)
            wide = center_crop_wide(img, (self.w, self.h))
            il = frame_position_planes(self.vae.encode_sample(self._pixels([wide]), None), window_len)   # sample() WITHOUT the generator (:486)
            conds.append(dict(fps=torch.tensor([self.fps, self.fps]), image_latents=torch.cat([il, il], 0),
                              image_embeddings=torch.cat([torch.zeros_like(emb), emb], 0), text=text))
        return conds

    def decode(self, latents):
        """latents [1, 4, F, h, w] -> uint8 frames [F, 8h, 8w, 3] (decode_chunk_size = 1)."""
        fr = latents[0].permute(1, 0, 2, 3)
        out = []
        for i in range(fr.shape[0]):
            x = self.vae.decode(fr[i:i + 1])
            out.append(((x / 2 + 0.5).clamp(0, 1) * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1))
        return torch.cat(out, 0).cpu().numpy()
