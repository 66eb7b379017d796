"""tests/golden/vfi_process_tiny.pt from the reference's UNMODIFIED `i2v_enhance_interface.vfi_process` (build container only).

    python oracle/make_golden_vfi_process.py

The function (code/i2v_enhance/i2v_enhance_interface.py:30-61) is imported through oracle/ar_bootstrap.py + oracle/i2v_pipeline_bootstrap.py
(+ a timm stub) and run on CPU with `Tensor.to("cuda", ...)` redirected to the CPU, around a stand-in `vfi.inference`.  It pins the frame
selection (video[: len // 2 + 1]), RGB <-> BGR flips, the / 255. -> fp32 -> * 255 -> uint8 round trip of the pass-through frames (NOT the
identity), the interleaving, the duplicated last frame for even lengths and the final PIL resize to 1280 x 720.  Stored: every output
frame subsampled on a 40-pixel grid (the full frames would be 2.7 MB each) plus the frames `inference` was called with.
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
from oracle.cases import tiny_vfi_process_inputs, tiny_vfi_process_infer  # noqa: E402


def main():
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
    iface = importlib.import_module("i2v_enhance.i2v_enhance_interface")

    seen = []

    class FakeVFI:
        def inference(self, I0, I2, TTA=False, fast_TTA=False):
            assert TTA and fast_TTA
            seen.append((I0.clone(), I2.clone()))
            return tiny_vfi_process_infer(I0, I2)

    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: orig_to(self, *[("cpu" if x == "cuda" else x) for x in a], **k)
    gold = {}
    try:
        for n in (7, 8):
            video = tiny_vfi_process_inputs()[: (n + 1) // 2]
            seen.clear()
            frames = iface.vfi_process(video=list(video), vfi=FakeVFI(), video_len=n)
            assert len(frames) == n and all(f.size == (1280, 720) for f in frames)
            gold[n] = dict(sub=torch.from_numpy(np.stack([np.asarray(f)[::40, ::40] for f in frames], 0).copy()),
                           pairs=[(a.clone(), b.clone()) for a, b in seen])
            print(f"[vfi_process] video_len {n}: {len(video)} input frames -> {len(frames)} frames, {len(seen)} inference calls")
    finally:
        torch.Tensor.to = orig_to
    out = os.path.join(ROOT, "tests", "golden", "vfi_process_tiny.pt")
    torch.save(gold, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
