"""tests/golden/frontend_enhance_tiny.pt from the reference's UNMODIFIED `inference_i2v.StreamingPipeline.enhance_video` (build container).

    python oracle/make_golden_frontend_enhance.py

The method (code/inference_i2v.py:192-209) is run unbound on a bare object around a recording stand-in for the enhancement pipeline: it pins
the image handling IN FRONT of the enhancer -- the key image goes through IImage(...).resize((720, 1280)) (PIL BICUBIC to 1280 x 720), every
video frame through PIL's default resize to 1280 x 720 -- before `_center_crop_wide` ever sees them.  Stored: the images the pipeline
received, subsampled on a 40-pixel grid.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ar_bootstrap, i2v_pipeline_bootstrap  # noqa: E402
from oracle.cases import tiny_frontend_enhance_inputs  # noqa: E402


def install_inference_i2v():
    import importlib
    ar_bootstrap.install()
    i2v_pipeline_bootstrap.install()

    class DropPath(nn.Module):
        def forward(self, x):
            return x
    layers = types.ModuleType("timm.models.layers")
    layers.DropPath, layers.to_2tuple, layers.trunc_normal_ = DropPath, (lambda x: (x, x)), (lambda t, std=1.0, **kw: t)
    timm, models = types.ModuleType("timm"), types.ModuleType("timm.models")
    timm.models, models.layers = models, layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    pl = sys.modules["pytorch_lightning"]
    pl.__path__ = []
    for n in ("pytorch_lightning.cli", "pytorch_lightning.utilities", "pytorch_lightning.utilities.types", "pytorch_lightning.callbacks",
              "pytorch_lightning.loggers"):
        m = ar_bootstrap._Auto(n)
        m.__path__ = []
        sys.modules[n] = m
    return importlib.import_module("inference_i2v")


def main():
    import PIL.Image
    mod = install_inference_i2v()
    image, video = tiny_frontend_enhance_inputs()
    got = {}

    def pipeline(**kw):
        got.update(image=[np.asarray(i).copy() for i in kw["image"]], video=[np.asarray(f).copy() for f in kw["video"]],
                   sizes=[i.size for i in kw["image"]] + [f.size for f in kw["video"]])
        return types.SimpleNamespace(frames=[[PIL.Image.fromarray(np.asarray(f)) for f in kw["video"]]])

    # the trainer-side input resize (utils/inference_utils.resize_and_keep, streaming_svd.py:383) == pipeline.resize_and_keep
    from utils.inference_utils import resize_and_keep as ref_resize
    from streamingt2v_amd.pipeline import resize_and_keep
    rs = np.random.default_rng(3)
    for (h, w) in ((1080, 1920), (300, 533), (720, 1281)):
        a = rs.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(np.asarray(ref_resize(PIL.Image.fromarray(a))), resize_and_keep(a)), (h, w)
    print("[front end] pipeline.resize_and_keep == the reference's resize_and_keep on 3 sizes")
    bare = types.SimpleNamespace(use_memopt=False)
    out = mod.StreamingPipeline.enhance_video(bare, image, video, enhance_pipeline=pipeline, enhance_generator=None, chunk_size=len(video),
                                              overlap_size=0, strength=0.97, use_randomized_blending=False)
    assert all(s == (1280, 720) for s in got["sizes"]) and out.shape == (len(video), 720, 1280, 3) and out.dtype == np.uint8
    sub = lambda a: torch.from_numpy(np.ascontiguousarray(a[::40, ::40]))
    path = os.path.join(ROOT, "tests", "golden", "frontend_enhance_tiny.pt")
    torch.save(dict(image=[sub(a) for a in got["image"]], video=[sub(a) for a in got["video"]]), path)
    print(f"[front end] enhance_video handed the pipeline {len(got['image'])} image(s) and {len(got['video'])} frames of {got['sizes'][0]}; wrote {path} "
          f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
