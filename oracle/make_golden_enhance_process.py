"""tests/golden/enhance_process_tiny.pt from the reference's UNMODIFIED `i2v_enhance_interface.i2v_enhance_process` (build container only).

    python oracle/make_golden_enhance_process.py

The function (code/i2v_enhance/i2v_enhance_interface.py:86-138) runs around a RECORDING stand-in for the pipeline object: what is pinned is
the two-stage structure of randomized blending -- which frames are the key frames, the single-window key-frame pre-pass (chunk = number of
windows, overlap 0) whose OUTPUT frames become the per-window images of the main pass, the truncation of the video to whole windows, and the
arguments of both calls (strength, steps 30, guidance 9, decode_chunk_size 1, 720 x 1280, prompts) -- with and without randomized blending.
The stand-in returns 255 - frame so that the data flow is visible.  Frames are identified by a CRC of their bytes.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ar_bootstrap, i2v_pipeline_bootstrap  # noqa: E402
from oracle.cases import tiny_enhance_process_inputs  # noqa: E402

crc = lambda f: zlib.crc32(np.ascontiguousarray(np.asarray(f)).tobytes())


def main():
    import importlib
    import PIL.Image
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
    iface = importlib.import_module("i2v_enhance.i2v_enhance_interface")

    gold = {}
    for name, n_frames, rb, chunk, overlap in (("blend_100", 100, True, 38, 12), ("blend_90", 90, True, 38, 12), ("plain_100", 100, False, 100, 0)):
        image, video = tiny_enhance_process_inputs(n_frames)
        calls = []

        def pipeline(**kw):
            calls.append({k: (v if isinstance(v, (int, float, str, bool, type(None))) else None) for k, v in kw.items()}
                         | dict(image=[crc(i) for i in kw["image"]], video=[crc(f) for f in kw["video"]]))
            return types.SimpleNamespace(frames=[[PIL.Image.fromarray(255 - np.asarray(f)) for f in kw["video"]]])

        out = iface.i2v_enhance_process(image=image, video=video, pipeline=pipeline, generator="GEN", overlap_size=overlap, strength=0.97,
                                        chunk_size=chunk, use_randomized_blending=rb, use_memopt=False)
        for c in calls:
            c["generator"] = "GEN"
        gold[name] = dict(calls=calls, out=[crc(f) for f in out])
        print(f"[enhance_process] {name}: {len(calls)} pipeline call(s); " +
              "; ".join(f"images {len(c['image'])}, frames {len(c['video'])}, chunk {c['chunk_size']}, overlap {c['overlap_size']}, num_frames {c['num_frames']}" for c in calls))
    out_path = os.path.join(ROOT, "tests", "golden", "enhance_process_tiny.pt")
    torch.save(gold, out_path)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
