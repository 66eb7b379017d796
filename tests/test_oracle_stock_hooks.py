"""The stock-op execution hooks of oracle/make_golden_fullsize_gpu.py (im2col + matmul convolutions, spelled-out attention) are the library calls they
replace: checked on CPU against F.conv2d / F.conv3d / F.scaled_dot_product_attention and through the whole tiny StreamingWrapper + decoder restatement.
(The product-size goldens of a whole chunk are generated with them on the GPU box: tests/golden/chunk30_fullsize.pt, ar_handover_fullsize.pt.)"""
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen_module():
    spec = importlib.util.spec_from_file_location("make_golden_fullsize_gpu", os.path.join(ROOT, "oracle", "make_golden_fullsize_gpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("cin,cout,k,stride,pad", [(8, 12, 3, 1, 1), (8, 12, 3, 2, 1), (16, 4, 1, 1, 0), (6, 5, 3, 2, 0)])
def test_im2col_conv_is_conv2d(cin, cout, k, stride, pad):
    m = _gen_module()
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(3, cin, 11, 14, generator=g), torch.randn(cout, cin, k, k, generator=g), torch.randn(cout, generator=g)
    m.COL_BYTES = cin * k * k * 4 * 40          # forces several frame batches
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    assert torch.allclose(m.conv2d_im2col(x, w, b, stride=stride, padding=pad), ref, atol=2e-5)
    assert torch.allclose(m.conv2d_im2col(x, w, None, stride=stride, padding=pad), F.conv2d(x, w, None, stride=stride, padding=pad), atol=2e-5)


def test_t3_conv_is_conv3d_and_attention_is_sdpa():
    m = _gen_module()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 7, 6, 9, generator=g).transpose(1, 2).contiguous().transpose(1, 2)          # the oracle hands over a transposed view
    w, b = torch.randn(4, 5, 3, 1, 1, generator=g), torch.randn(4, generator=g)
    assert torch.allclose(m.conv3d_t3(x, w, b, padding=(1, 0, 0)), F.conv3d(x, w, b, padding=(1, 0, 0)), atol=2e-5)
    q, k, v = (torch.randn(5, 3, n, 16, generator=g) for n in (10, 7, 7))
    assert torch.allclose(m.sdpa_stock(q, k, v), F.scaled_dot_product_attention(q, k, v), atol=2e-6)


def test_tiny_wrapper_sampler_and_decoder_through_the_hooks(golden_dir):
    """the restatement executed through the hooks reproduces the REFERENCE's goldens like the library-call execution does"""
    from oracle import cases, svd_oracle as O
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VaeConfig, VideoDecoder
    from tests.test_oracle_golden import TOL, _tiny_state
    m = _gen_module()
    saved = (O.conv2d, O.conv3d, O.sdpa)
    torch.set_grad_enabled(False)
    try:
        m.install_hooks()
        sd_u, sd_c, ocfg, tu = _tiny_state(golden_dir)
        inp, sin = cases.tiny_wrapper_inputs(), cases.tiny_sampler_inputs()
        c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
        out = O.streaming_wrapper(sd_u, sd_c, ocfg, inp["x"], inp["t"], c, 2, tu["T"], tu["Tc"], inp["ctrl_frames"])
        assert (out - torch.load(os.path.join(golden_dir, "wrapper_tiny.pt"))["out"]).abs().max().item() <= TOL
        net = lambda a, cn_, cc: O.streaming_wrapper(sd_u, sd_c, ocfg, a, cn_, cc, 2, tu["T"], tu["Tc"], inp["ctrl_frames"])
        z = O.euler_edm_sample(net, sin["noise"].clone(), sin["c"], sin["uc"], 2, tu["T"])
        assert (z - torch.load(os.path.join(golden_dir, "sampler_tiny.pt"))["z"]).abs().max().item() <= 5 * TOL
        tv = cases.TINY_VAE
        sd = init_by_name(VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])).spec(), seed=3)
        zz = cases.tiny_vae_inputs()["z"]
        dec = O.video_decoder(sd, O.VaeCfg(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]), zz, zz.shape[0])
        assert (dec - torch.load(os.path.join(golden_dir, "vae_tiny.pt"))["out"]).abs().max().item() <= TOL
        # the reference's convert_range round trip of the hand-over (streaming_svd.py:263-290) = the product's extract_ctrl_frames
        from streamingt2v_amd.streaming_svd import StreamingSVD
        fr = torch.rand(9, 3, 8, 8) * 2 - 1
        assert torch.equal(m.handover_ctrl(fr, 7), StreamingSVD.extract_ctrl_frames(fr, 7))
    finally:
        O.conv2d, O.conv3d, O.sdpa = saved
