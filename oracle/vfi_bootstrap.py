"""Import bootstrap for the reference's vendored EMA-VFI (code/i2v_enhance/thirdparty/VFI) -- TEST INFRASTRUCTURE ONLY.

Used by oracle/make_golden_vfi.py inside the build container to run the UNMODIFIED vendored model on CPU.  Two things stand
between the vendored files and a CPU run:
  * ``timm`` is not installed: model/feature_extractor.py:5 and model/refine.py:4 import ``DropPath, to_2tuple, trunc_normal_`` from
    ``timm.models.layers``.  Only initialisation helpers and the (inference-time identity) DropPath are used; they are stubbed.
  * model/flow_estimation.py:82,120 call ``.cuda()`` on a freshly created tensor.  ``torch.Tensor.cuda`` is replaced by the identity
    while the reference runs (``cpu_only()`` context).
The package is imported as ``i2v_enhance.thirdparty.VFI`` without executing ``i2v_enhance/__init__``-level imports of diffusers.
"""
import contextlib
import importlib
import sys
import types

import torch
import torch.nn as nn

REF_CODE = "/root/reference/code"


def install():
    if "timm" not in sys.modules:
        class DropPath(nn.Module):
            def __init__(self, p=0.0):
                super().__init__()

            def forward(self, x):
                return x

        layers = types.ModuleType("timm.models.layers")
        layers.DropPath = DropPath
        layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        layers.trunc_normal_ = lambda t, std=1.0, **kw: nn.init.trunc_normal_(t, std=std)
        timm, models = types.ModuleType("timm"), types.ModuleType("timm.models")
        timm.models, models.layers = models, layers
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    # namespace packages so that "i2v_enhance.thirdparty.VFI.*" resolves without running i2v_enhance's diffusers imports
    for name, path in (("i2v_enhance", f"{REF_CODE}/i2v_enhance"), ("i2v_enhance.thirdparty", f"{REF_CODE}/i2v_enhance/thirdparty"),
                       ("i2v_enhance.thirdparty.VFI", f"{REF_CODE}/i2v_enhance/thirdparty/VFI")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m


@contextlib.contextmanager
def cpu_only():
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def build_reference(F=32, depth=(2, 2, 2, 4, 4), W=7):
    """The network exactly as i2v_enhance_interface.vfi_init builds it (:15-19): Model(-1).net with init_model_config(F, depth)."""
    install()
    cfg = importlib.import_module("i2v_enhance.thirdparty.VFI.config")
    cfg.MODEL_CONFIG["MODEL_ARCH"] = cfg.init_model_config(F=F, W=W, depth=list(depth))
    trainer = importlib.import_module("i2v_enhance.thirdparty.VFI.Trainer")
    model = trainer.Model(-1)
    model.eval()
    return model
