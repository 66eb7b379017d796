"""-m gpu: the per-step network evaluation captured in a hipGraph (sampling.EulerEDMSampler(use_graph=True), round 3) replays to exactly the
bits of the eager launch sequence -- with and without control frames (AR chunk / chunk 0), across chunks (a new capture per chunk, new control
frames), and through the chunk driver incl. decode.  Host-side cost of a replayed step is measured by tools/host_bound_probe.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from oracle import cases
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    ops.set_element_dtype(None)
    tu = cases.TINY_UNET
    cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                     conditioning_embedding_out_channels=tu["cond_embed"])
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    unet.load_state_dict(init_by_name(unet.spec(), seed=1), device="cuda")
    cn.load_state_dict(init_by_name(cn.spec(), seed=2), device="cuda")
    return dict(wrap=StreamingWrapper(unet, cn, tu["Tc"]), cases=cases, T=tu["T"])


def _inputs(cases):
    sin, inp = cases.tiny_sampler_inputs(), cases.tiny_wrapper_inputs()
    dev = lambda d: {k: v.cuda() for k, v in d.items()}
    return dev(sin["c"]), dev(sin["uc"]), sin["noise"].cuda(), inp["ctrl_frames"].cuda()


@pytest.mark.parametrize("with_ctrl", [True, False])
def test_graph_replay_is_bit_identical_to_eager(tiny, with_ctrl):
    from streamingt2v_amd.sampling import EulerEDMSampler
    T, wrap = tiny["T"], tiny["wrap"]
    c, uc, noise, ctrl = _inputs(tiny["cases"])
    kw = dict(batch_size=2, num_video_frames=T, ctrl_frames=ctrl if with_ctrl else None)
    with torch.no_grad():
        z_eager = EulerEDMSampler(num_steps=6, num_frames=T)(wrap, noise.clone(), c, uc, **kw)
        sam = EulerEDMSampler(num_steps=6, num_frames=T, use_graph=True)
        z_graph = sam(wrap, noise.clone(), c, uc, **kw)
        # a second chunk through the same sampler object: new control frames / noise -> a new capture in the same pool
        ctrl2 = (ctrl * 0.5 + 0.1) if with_ctrl else None
        kw2 = dict(kw, ctrl_frames=ctrl2)
        z2_graph = sam(wrap, (noise * 0.9).clone(), c, uc, **kw2)
        z2_eager = EulerEDMSampler(num_steps=6, num_frames=T)(wrap, (noise * 0.9).clone(), c, uc, **kw2)
    torch.cuda.synchronize()
    assert torch.isfinite(z_graph).all()
    assert torch.equal(z_graph, z_eager), (z_graph - z_eager).abs().max().item()
    assert torch.equal(z2_graph, z2_eager), (z2_graph - z2_eager).abs().max().item()


def test_chunk_driver_with_graph(tiny):
    """_generate_conditional_output (sampler + temporal-VAE decode + clamp) with the graphed sampler == eager."""
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VaeConfig, VideoDecoder
    cases, T, wrap = tiny["cases"], tiny["T"], tiny["wrap"]
    tv = cases.TINY_VAE
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    dec.load_state_dict(init_by_name(dec.spec(), seed=3), device="cuda")
    c, uc, noise, ctrl = _inputs(cases)
    with torch.no_grad():
        a = StreamingSVD(wrap, AutoencodingEngineDecoder(dec), EulerEDMSampler(num_steps=4, num_frames=T))._generate_conditional_output(c, uc, ctrl, noise)
        b = StreamingSVD(wrap, AutoencodingEngineDecoder(dec), EulerEDMSampler(num_steps=4, num_frames=T, use_graph=True))._generate_conditional_output(c, uc, ctrl, noise)
    assert torch.equal(a, b)
