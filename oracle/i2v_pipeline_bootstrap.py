"""Run the reference's REAL `I2VGenXLPipeline.__call__` (code/i2v_enhance/pipeline_i2vgen_xl.py) on CPU -- TEST INFRASTRUCTURE ONLY.

Purpose: pin everything around the enhancer UNet that is the reference's OWN code -- image cropping / resizing, the frame-position planes,
classifier-free-guidance batching, fps conditioning, the SDEdit strength arithmetic, chunked video encoding, the randomized-blending loop
with Python's `random`, DDIM stepping order -- by executing it unmodified with
  * the vendored I2VGenXLUNet (tiny configuration, through oracle/i2v_bootstrap.py),
  * linear stand-ins for the networks that are not vendored (VAE, CLIP image encoder: `FakeVAE`, `FakeImageEncoder` below), shared with
    the oracle side so that both run the same arithmetic,
  * restated diffusers 0.30.2 plumbing that the pipeline imports (diffusers is not installed): DiffusionPipeline (module registry,
    progress bar), VaeImageProcessor / VideoProcessor (PIL <-> tensor, [0,1] -> [-1,1]), randn_tensor, DDIMScheduler (oracle DDIM:
    parity unpinned for the scheduler arithmetic itself).
Used by oracle/make_golden_i2v_pipeline.py.
"""
import contextlib
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import i2v_bootstrap
from .i2vgen_oracle import DDIM


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class FakeVAE(nn.Module):
    """Deterministic stand-in for AutoencoderKL: encode = 8x8 average pooling + fixed 3->4 channel mix (posterior std 0.05), decode =
    fixed 4->3 mix + nearest 8x upsampling.  `latent_dist.sample(generator)` draws like DiagonalGaussianDistribution (randn of the mean's shape)."""

    def __init__(self):
        super().__init__()
        self.config = _Cfg(scaling_factor=0.18215, block_out_channels=(128, 256, 512, 512))
        g = torch.Generator().manual_seed(123)
        self.register_buffer("enc", torch.randn(4, 3, generator=g) * 0.6)
        self.register_buffer("dec", torch.randn(3, 4, generator=g) * 0.6)
        self.std = 0.05
        self.encode_calls = []

    def mean(self, x):
        p = torch.nn.functional.avg_pool2d(x.float(), 8)
        return torch.einsum("oc,bchw->bohw", self.enc, p)

    def encode(self, x):
        vae = self
        self.encode_calls.append(tuple(x.shape))

        class Dist:
            def sample(self, generator=None):
                m = vae.mean(x)
                return m + vae.std * torch.randn(m.shape, generator=generator)

            def mode(self):
                return vae.mean(x)
        return types.SimpleNamespace(latent_dist=Dist())

    def decode(self, z):
        y = torch.einsum("oc,bchw->bohw", self.dec, z.float())
        return types.SimpleNamespace(sample=torch.nn.functional.interpolate(y, scale_factor=8, mode="nearest"))


class FakeImageEncoder(nn.Module):
    """Stand-in for CLIPVisionModelWithProjection: image_embeds = fixed projection of the 8x8-pooled, CLIP-normalised pixels."""

    def __init__(self, dim):
        super().__init__()
        g = torch.Generator().manual_seed(321)
        self.proj = nn.Parameter(torch.randn(dim, 3 * 8 * 8, generator=g) * 0.1, requires_grad=False)

    def embed(self, pixels):
        p = torch.nn.functional.adaptive_avg_pool2d(pixels.float(), 8).flatten(1)
        return p @ self.proj.t()

    def forward(self, pixels):
        return types.SimpleNamespace(image_embeds=self.embed(pixels))


class SchedulerShim:
    """The calls pipeline_i2vgen_xl.py makes on DDIMScheduler, on top of the oracle's DDIM."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.d = DDIM()
        self.begin_index = None

    def set_timesteps(self, n, device=None):
        self.d.set_timesteps(n)
        self.timesteps = self.d.timesteps

    def set_begin_index(self, i):
        self.begin_index = i

    def scale_model_input(self, x, t):
        return x

    def add_noise(self, x0, noise, timesteps):
        a = self.d.alphas_cumprod[timesteps].view(-1, *([1] * (x0.dim() - 1)))
        return a.sqrt() * x0 + (1 - a).sqrt() * noise

    def step(self, model_output, timestep, sample, eta=0.0, generator=None):
        assert eta == 0.0
        return types.SimpleNamespace(prev_sample=self.d.step(model_output, int(timestep), sample))


def install():
    """i2v_bootstrap.install() + the diffusers names pipeline_i2vgen_xl.py imports."""
    i2v_bootstrap.install()
    ident = lambda f: f

    class VaeImageProcessor:
        def __init__(self, vae_scale_factor=8, do_resize=True, **kw):
            self.do_resize = do_resize

        @staticmethod
        def pil_to_numpy(images):
            images = images if isinstance(images, list) else [images]
            return np.stack([np.array(im).astype(np.float32) / 255.0 for im in images], 0)

        @staticmethod
        def numpy_to_pt(images):
            return torch.from_numpy(images.transpose(0, 3, 1, 2))

        @staticmethod
        def pt_to_numpy(images):
            return images.cpu().permute(0, 2, 3, 1).float().numpy()

        @staticmethod
        def numpy_to_pil(images):
            import PIL.Image
            return [PIL.Image.fromarray((im * 255).round().astype("uint8")) for im in images]

        def preprocess(self, image, height=None, width=None):
            assert not self.do_resize
            if isinstance(image, torch.Tensor):
                x = image if image.dim() == 4 else image[None]
            else:
                x = self.numpy_to_pt(self.pil_to_numpy(image))
            return 2.0 * x - 1.0                                              # do_normalize

    class VideoProcessor(VaeImageProcessor):
        def preprocess_video(self, video, height=None, width=None):
            """list of PIL / HWC uint8 arrays (one video) -> [1, C, F, H, W] in [-1, 1]."""
            import PIL.Image
            frames = [f if isinstance(f, PIL.Image.Image) else PIL.Image.fromarray(np.asarray(f)) for f in video]
            return self.preprocess(frames).permute(1, 0, 2, 3)[None]

    class DiffusionPipeline:
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def _execution_device(self):
            return torch.device("cpu")

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield types.SimpleNamespace(update=lambda *a: None)

        def maybe_free_model_hooks(self):
            pass

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(tuple(shape), generator=generator, dtype=dtype).to(device or "cpu")

    def stub(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    stub("diffusers.image_processor", PipelineImageInput=object, VaeImageProcessor=VaeImageProcessor)
    stub("diffusers.models", AutoencoderKL=object)
    stub("diffusers.schedulers", DDIMScheduler=object)
    stub("diffusers.utils", replace_example_docstring=lambda doc: ident)
    stub("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    stub("diffusers.video_processor", VideoProcessor=VideoProcessor)
    stub("diffusers.pipelines")
    stub("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline, StableDiffusionMixin=type("StableDiffusionMixin", (), {}))


def build_pipeline(unet, embed_dim):
    """The reference pipeline object around a vendored UNet and the stand-in encoders (no tokenizer / text encoder: prompt embeddings are
    passed to __call__, which the pipeline supports)."""
    install()
    import importlib
    from transformers import CLIPImageProcessor
    mod = importlib.import_module("i2v_enhance.pipeline_i2vgen_xl")
    fe = CLIPImageProcessor(crop_size={"height": 224, "width": 224}, size={"shortest_edge": 224})
    pipe = mod.I2VGenXLPipeline(vae=FakeVAE(), text_encoder=None, tokenizer=None, image_encoder=FakeImageEncoder(embed_dim), feature_extractor=fe,
                                unet=unet, scheduler=SchedulerShim())
    return pipe, mod
