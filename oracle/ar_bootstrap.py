"""Import the reference's `diffusion_trainer.streaming_svd.StreamingSVD` class on CPU -- TEST INFRASTRUCTURE ONLY (build container).

On top of oracle/ref_bootstrap.py (pytorch_lightning / omegaconf / open_clip / kornia / diffusers.Attention stubs) the module pulls in
imageio, cv2, jsonargparse, gdown, torchvision, matplotlib and IPython through `lib.farancia`, `utils.loader`, `modules.loader`: none of
them touches the arithmetic of the autoregressive outer loop, so they are satisfied by empty auto-attribute modules.  `torchvision.transforms.
ToTensor` (used by image_to_video :392) is restated (uint8 HWC -> float CHW / 255).  transformers is imported FIRST because it probes
`torchvision` with importlib.util.find_spec, which rejects a stub module.
"""
import importlib
import sys
import types

import numpy as np
import torch


class _Auto(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def install():
    import transformers  # noqa: F401
    from transformers import (ByT5Tokenizer, CLIPImageProcessor, CLIPTextModel, CLIPTokenizer, CLIPVisionModelWithProjection,  # noqa: F401
                              T5EncoderModel, T5Tokenizer)
    from . import ref_bootstrap
    ref_bootstrap.install()
    for n in ("imageio", "imageio.v3", "cv2", "jsonargparse", "gdown", "torchvision", "torchvision.utils", "torchvision.datasets",
              "torchvision.datasets.utils", "torchvision.transforms.functional", "matplotlib", "matplotlib.pyplot", "IPython", "IPython.display"):
        if n not in sys.modules:
            m = _Auto(n)
            m.__path__ = []
            sys.modules[n] = m

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic)
            assert a.dtype == np.uint8 and a.ndim == 3
            return torch.from_numpy(a.astype(np.float32).transpose(2, 0, 1) / 255.0)

    tt = types.ModuleType("torchvision.transforms")
    tt.__path__ = []
    tt.ToTensor = tt.PILToTensor = ToTensor
    tt.Compose = lambda fs: (lambda x: [x := f(x) for f in fs][-1])
    tt.functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision.transforms"] = tt
    sys.modules["torchvision"].transforms = tt
    d = sys.modules["diffusers"]
    for n in ("DDPMScheduler", "DiffusionPipeline", "StableVideoDiffusionPipeline"):
        if not hasattr(d, n):
            setattr(d, n, type(n, (), {}))
    return importlib.import_module("diffusion_trainer.streaming_svd").StreamingSVD
