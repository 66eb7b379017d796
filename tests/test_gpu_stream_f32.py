"""Kernel-level parity (-m gpu) of the fp32 RESIDUAL STREAM option (round 3): GEMM epilogues that read the residual / blend partner and
write the sum in fp32 (svd_gemm_args.res_f32, SVD_OUT_F32; specialised kinds 8 / 9 / 13 and the generic pass), the norms and add_rows on
fp32 inputs (SVD_DTYPE_IN_F32), the vectorised fp32 -> 16-bit row cast, the row permute of the sequence-parallel repack -- each through the
C ABI against a plain PyTorch fp32 statement of the same op on the same inputs.  With fp32 in and fp32 out the only 16-bit roundings left are
the GEMM operands, so the tolerances are fp32-accumulation tolerances, not 16-bit output tolerances."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ELEM = torch.float16


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def ops(request):
    global ELEM
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from streamingt2v_amd import ops as o
    ELEM = request.param
    o.set_element_dtype(request.param)
    yield o
    o.set_element_dtype(None)


def rnd(*shape, scale=1.0, seed=0, dtype=None):
    g = torch.Generator(device="cpu"); g.manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype or ELEM).cuda()


def close(name, got, ref, atol, rtol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    worst = (err - (atol + rtol * ref.abs())).max().item()
    print(f"[{name}] max abs err {err.max().item():.3e} (ref absmax {ref.abs().max().item():.3f})")
    assert torch.isfinite(got).all() and worst <= 0, f"{name}: err {err.max().item():.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 4, 8, 9, 13, 16, 17, 18, 20, 21, 22, 23])
def test_gemm_fp32_stream_epilogues(ops, cfg):
    """kind 9 (+residual), 13 (+residual + per-frame vector), 8 (plain fp32 out), and the generic pass (blend partner) with fp32 R / S;
    M large enough for several tiles per workgroup on the big tiles, N = 320 k so that the N = 320 tiles are valid."""
    M, N, K, rpv = 2304 * 3 + 40, 640, 320, 2304
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = rnd(N, seed=3, dtype=torch.float32)
    rowvec = rnd((M + rpv - 1) // rpv, N, seed=4, dtype=torch.float32)
    R, S = rnd(M, N, seed=5, dtype=torch.float32) * 3, rnd(M, N, seed=6, dtype=torch.float32) * 3
    mm = a.float() @ w.float().t() + bias
    out = ops.gemm(a, w, bias=bias, residual=R, out_f32=True, tile_cfg=cfg)
    assert out.dtype == torch.float32
    close(f"cfg{cfg} +R32 -> f32", out, mm + R, 2e-4, 2e-5)
    out = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R, out_f32=True, tile_cfg=cfg)
    close(f"cfg{cfg} +R32+vec -> f32", out, mm + rowvec.repeat_interleave(rpv, 0)[:M] + R, 2e-4, 2e-5)
    out = ops.gemm(a, w, bias=bias, out_f32=True, tile_cfg=cfg)
    close(f"cfg{cfg} plain -> f32", out, mm, 2e-4, 2e-5)
    out = ops.gemm(a, w, bias=bias, residual=R, blend=(0.3, S), out_f32=True, tile_cfg=cfg)
    close(f"cfg{cfg} +R32 blend S32 -> f32", out, 0.3 * S + 0.7 * (mm + R), 2e-4, 2e-5)
    out = ops.gemm(a, w, bias=bias, residual=R, blend=(0.3, S), tile_cfg=cfg)            # the transformer's blend: fp32 stream in, 16-bit operand out
    assert out.dtype == ELEM
    ref = 0.3 * S + 0.7 * (mm + R)
    close(f"cfg{cfg} +R32 blend S32 -> 16 bit", out, ref, 1e-5, 2.0 ** (-7 if ELEM == torch.bfloat16 else -10))
    out = ops.gemm(a, w, bias=bias, residual=R.to(ELEM), out_f32=True, tile_cfg=cfg)      # 16-bit residual, fp32 out: generic pass
    close(f"cfg{cfg} +R16 -> f32", out, mm + R.to(ELEM).float(), 2e-4, 2e-5)


def test_gemm_fp32_stream_conv_and_temporal(ops):
    """The stream's convolution producers: 3x3 conv + fp32 residual (ResBlock out_layers + skip), temporal 3-tap conv + residual + blend."""
    Fr, H, W, C = 4, 18, 32, 64
    x = rnd(Fr * H * W, C, seed=7)
    w = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=8, dtype=torch.float32)
    from streamingt2v_amd.video_model import pack_conv3x3, pack_tconv3
    wp = pack_conv3x3(w.cpu()).to(ELEM).cuda()
    bias = rnd(C, seed=9, dtype=torch.float32)
    R = rnd(Fr * H * W, C, seed=10, dtype=torch.float32)
    out = ops.gemm(x, wp, bias=bias, residual=R, out_f32=True, conv=dict(cin=C, hin=H, win=W, hout=H, wout=W, frames=Fr))
    ref = F.conv2d(x.float().view(Fr, H, W, C).permute(0, 3, 1, 2), wp.float().view(C, 3, 3, C).permute(0, 3, 1, 2), bias, 1, 1)
    close("conv3x3 +R32 -> f32", out, ref.permute(0, 2, 3, 1).reshape(-1, C) + R, 3e-4, 3e-5)
    wt = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=11, dtype=torch.float32)
    wtp = pack_tconv3(wt.cpu()).to(ELEM).cuda()
    out = ops.gemm(x, wtp, bias=bias, residual=R, blend=(0.4, R), out_f32=True, temporal=dict(cin=C, T=Fr, pix=H * W))
    xt = x.float().view(1, Fr, H * W, C).permute(0, 3, 1, 2)[..., None]
    ref = F.conv3d(xt, wtp.float().view(C, 3, C).permute(0, 2, 1)[..., None, None], bias, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, C)
    close("temporal3 +R32 blend -> f32", out, 0.4 * R + 0.6 * (ref + R), 3e-4, 3e-5)


def test_gemm_fp32_stream_narrow_tail(ops):
    """N not a multiple of 8 with an fp32 residual: the element-wise tail of the generic pass."""
    M, N, K = 300, 12, 64
    a, w = rnd(M, K, seed=12), rnd(N, K, scale=K ** -0.5, seed=13)
    R = rnd(M, N, seed=14, dtype=torch.float32)
    out = ops.gemm(a, w, residual=R, out_f32=True)
    close("narrow +R32", out, a.float() @ w.float().t() + R, 2e-4, 2e-5)


@pytest.mark.parametrize("C,pix,frames,fps", [(320, 2304, 4, 1), (640, 576, 6, 3), (1280, 144, 4, 2), (128, 4608, 2, 1)])
def test_groupnorm_on_fp32_stream(ops, C, pix, frames, fps):
    x = rnd(frames * pix, C, seed=15, dtype=torch.float32) * 2 + 0.3
    g, b = rnd(C, seed=16, dtype=torch.float32), rnd(C, seed=17, dtype=torch.float32)
    for silu in (False, True):
        out = ops.groupnorm(x, frames, pix, g, b, 1e-5, frames_per_stat=fps, silu=silu)
        assert out.dtype == ELEM
        v = x.view(frames // fps, fps * pix, 32, C // 32)
        ref = ((v - v.mean((1, 3), keepdim=True)) * torch.rsqrt(v.var((1, 3), unbiased=False, keepdim=True) + 1e-5)).reshape(frames * pix, C) * g + b
        if silu:
            ref = F.silu(ref)
        close(f"groupnorm f32-in C{C} silu{int(silu)}", out, ref, 2e-5, 2.0 ** (-7 if ELEM == torch.bfloat16 else -10))
    # statistics from sums (sequence-parallel form) on the fp32 input
    sums = ops.groupnorm_sums(x, frames, pix, fps)
    out2 = ops.groupnorm_apply_sums(x, frames, pix, g, b, 1e-5, sums, float(fps) * pix * (C // 32), frames_per_stat=fps, silu=True)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm_on_fp32_stream(ops, C):
    rows, rpv = 1030, 100
    x = rnd(rows, C, seed=18, dtype=torch.float32) * 1.5 + 0.2
    g, b = rnd(C, seed=19, dtype=torch.float32), rnd(C, seed=20, dtype=torch.float32)
    out = ops.layernorm(x, g, b)
    assert out.dtype == ELEM
    tol = 2.0 ** (-7 if ELEM == torch.bfloat16 else -10)
    close(f"layernorm f32-in C{C}", out, F.layer_norm(x, (C,), g, b), 2e-5, tol)
    add = rnd((rows + rpv - 1) // rpv, C, seed=21, dtype=torch.float32)
    out, xs = ops.layernorm(x, g, b, addvec=add, rows_per_vec=rpv, want_sum=True)
    ref_sum = x + add.repeat_interleave(rpv, 0)[:rows]
    assert xs.dtype == torch.float32 and torch.equal(xs, ref_sum)               # the sum continues the stream exactly
    close(f"layernorm f32-in +vec C{C}", out, F.layer_norm(ref_sum, (C,), g, b), 2e-5, tol)


def test_stream_elementwise(ops):
    rows, ca, cb = 1000, 320, 640
    a, b = rnd(rows, ca, seed=22, dtype=torch.float32), rnd(rows, cb, seed=23, dtype=torch.float32)
    cat = ops.concat_channels(a, b)
    assert cat.dtype == ELEM and torch.equal(cat, torch.cat([a.to(ELEM), b.to(ELEM)], 1))
    cat = ops.concat_channels(a, b.to(ELEM))                                       # mixed: one side already 16 bit
    assert torch.equal(cat, torch.cat([a.to(ELEM), b.to(ELEM)], 1))
    assert torch.equal(ops.to_elem_rows(a), a.to(ELEM))
    view = b[:, 64:64 + 320]                                                       # strided rows
    assert torch.equal(ops.to_elem_rows(view), view.to(ELEM))
    c16 = rnd(rows, ca, seed=24)
    s = ops.add_rows(a, c16)
    assert s.dtype == torch.float32 and torch.equal(s, a + c16.float())
    s16 = ops.add_rows(a.to(ELEM), c16)                                            # the 16-bit form is unchanged
    assert s16.dtype == ELEM
    close("add_rows 16", s16, a.to(ELEM).float() + c16.float(), 1e-6, 2.0 ** (-7 if ELEM == torch.bfloat16 else -10))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_permute_rows(ops, dtype):
    dims, C = (3, 5, 4, 7), 40
    x = torch.randn(dims[0] * dims[1] * dims[2] * dims[3], C, device="cuda").to(dtype)
    for perm in ((2, 0, 1, 3), (1, 2, 0, 3), (0, 1, 2, 3), (3, 2, 1, 0)):
        got = ops.permute_rows(x, dims, perm)
        ref = x.view(*dims, C).permute(*perm, 4).reshape(-1, C)
        assert torch.equal(got, ref), perm
