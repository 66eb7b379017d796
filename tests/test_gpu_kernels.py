"""Kernel-level parity (-m gpu): every libsvdhip.so kernel through the C ABI vs a plain PyTorch fp32 reference of the
same op on the same bf16-rounded inputs.  Tolerances are bf16-output tolerances: the kernels accumulate in fp32, so
the error budget is one bf16 rounding of the result (rel 2^-8) plus accumulation-order noise."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16      # rebound per parametrisation: the element type under test (bf16 | fp16)
TOLF = 1.0                 # tolerance factor: fp16 has 3 more mantissa bits -> tolerances / 4


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def ops(request):
    """Every kernel test runs once per element type of the C ABI (dtype argument)."""
    global BF16, TOLF
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from streamingt2v_amd import ops as o
    BF16 = request.param
    TOLF = 1.0 if request.param == torch.bfloat16 else 0.25
    o.set_element_dtype(request.param)
    yield o
    o.set_element_dtype(None)


def rnd(*shape, scale=1.0, seed=0, dtype=None):
    g = torch.Generator(device="cpu"); g.manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype or BF16).cuda()


def check(name, got, ref, atol, rtol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bound = TOLF * (atol + rtol * ref.abs()) if atol > 1e-5 else atol + rtol * ref.abs()
    worst = (err - bound).max().item()
    print(f"[{name}] max abs err {err.max().item():.4e} (ref absmax {ref.abs().max().item():.3f}) worst margin {worst:.3e}")
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    assert worst <= 0, f"{name}: err {err.max().item():.4e} exceeds atol {atol} + rtol {rtol}"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 320, 320), (77, 960, 64), (4096, 640, 1280)])
def test_gemm_plain(ops, cfg, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = rnd(N, seed=3, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    out = ops.gemm(a, w, bias=bias, tile_cfg=cfg)
    check(f"gemm cfg{cfg} {M}x{N}x{K}", out, ref, 2e-2, 1e-2)
    # asymmetric structure check: A = "identity-like" selects rows of W (catches transposed tiles)
    eye = torch.zeros(M, K, dtype=BF16, device="cuda"); idx = torch.arange(M, device="cuda") % K
    eye[torch.arange(M, device="cuda"), idx] = 1
    out = ops.gemm(eye, w, tile_cfg=cfg)
    check(f"gemm-select cfg{cfg}", out, w.float().t()[idx], 1e-6, 0)


@pytest.mark.parametrize("cfg", [0, 1, 8, 20, 21, 22])
def test_gemm_w_panel_walk(ops, cfg, monkeypatch):
    """The W-panel tile walk (csrc/gemm_impl.inc, round 6: when W exceeds an XCD's L2 the tiles are walked panel by panel of N tiles instead of N fastest over
    the whole width) visits every output tile exactly once, whatever the panel width: bit-identical to the N-fastest walk (SVD_GEMM_PANEL=0) for the width the
    traffic model picks and for forced widths that leave a narrower last panel, with the per-tile epilogue operands (bias slice, per-frame vector, residual) and
    a ragged last M tile and N tile; and equal to plain fp32 PyTorch."""
    M, N, K, rpv = 16500, 2568, 1280, 1500                 # W = 6.6 MB; 65..130 x 9..21 tiles depending on the configuration
    a, w = rnd(M, K, seed=31), rnd(N, K, scale=K ** -0.5, seed=32)
    bias = rnd(N, seed=33, dtype=torch.float32)
    rowvec = rnd(M // rpv, N, seed=34, dtype=torch.float32)
    R = rnd(M, N, seed=35)
    ref = a.float() @ w.float().t() + bias + rowvec.repeat_interleave(rpv, 0) + R.float()
    outs = {}
    for width in ("0", None, "1", "3", "7"):
        if width is None:
            monkeypatch.delenv("SVD_GEMM_PANEL", raising=False)
        else:
            monkeypatch.setenv("SVD_GEMM_PANEL", width)
        outs[width] = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R, tile_cfg=cfg)
    monkeypatch.delenv("SVD_GEMM_PANEL", raising=False)
    check(f"gemm panel walk cfg{cfg} vs fp32 torch", outs["0"], ref, 3e-2, 1e-2)
    for width, o in outs.items():
        assert torch.equal(o, outs["0"]), f"panel width {width}: differs from the N-fastest walk"
    if cfg in (21, 22):
        return                                             # 320-wide tiles do not take the GEGLU epilogue (test_gemm_geglu's list)
    # GEGLU projection (value | gate column pairs inside a tile) through the walk the model picks
    from streamingt2v_amd.video_model import pack_geglu
    Hd = 2560
    w1 = rnd(2 * Hd, K, scale=K ** -0.5, seed=36).float().cpu()
    b1 = rnd(2 * Hd, seed=37, dtype=torch.float32, scale=0.3).cpu()
    w1p, b1p = pack_geglu(w1, b1)
    w1p, b1p = w1p.to(BF16).cuda(), b1p.cuda()
    monkeypatch.setenv("SVD_GEMM_PANEL", "0")
    g0 = ops.gemm(a, w1p, bias=b1p, geglu=True, tile_cfg=cfg)
    monkeypatch.delenv("SVD_GEMM_PANEL", raising=False)
    g1 = ops.gemm(a, w1p, bias=b1p, geglu=True, tile_cfg=cfg)
    assert torch.equal(g0, g1)
    h = a.float() @ w1.to(BF16).float().cuda().t() + b1.cuda()
    check(f"gemm panel walk cfg{cfg} GEGLU vs fp32 torch", g1, h[:, :Hd] * F.gelu(h[:, Hd:]), 3e-2, 1e-2)


@pytest.mark.parametrize("cfg", [8, 20, 21])
def test_gemm_tail_split(ops, cfg, monkeypatch):
    """Tail split of the persistent GEMM (csrc/gemm_impl.inc launch_kernel, round 6): when the last round of a launch holds only a few 256-row tiles, their rows run as
    a second launch with a small tile.  Every tile configuration accumulates an output element in the same order, so the result must be BIT-IDENTICAL to the unsplit
    launch (SVD_GEMM_TAIL=0; the split is opt-in, SVD_GEMM_TAIL=1: measured neutral on the job) -- plain GEMM with bias / per-frame vector / residual (16-bit and the fp32 stream), GEGLU, the 3x3-convolution and 3-tap temporal views,
    a ragged last tile -- and equal to plain fp32 PyTorch.  Row counts: 1 032 and 520 tiles of 256 rows + a ragged rest (8 tiles past a multiple of 256 and of 512
    resident workgroups, whichever the configuration has)."""
    from streamingt2v_amd.video_model import pack_conv3x3, pack_geglu, pack_tconv3

    def both(fn):
        monkeypatch.setenv("SVD_GEMM_TAIL", "0")
        a = fn()
        monkeypatch.setenv("SVD_GEMM_TAIL", "1")
        b = fn()
        monkeypatch.delenv("SVD_GEMM_TAIL", raising=False)
        assert torch.equal(a, b), "the tail split changes the bits"
        return b

    K, N, rpv = 320, 320, 2064
    for M in (1032 * 256 - 37, 520 * 256 - 37):
        a, w = rnd(M, K, seed=81), rnd(N, K, scale=K ** -0.5, seed=82)
        bias = rnd(N, seed=83, dtype=torch.float32)
        rowvec = rnd((M + rpv - 1) // rpv, N, seed=84, dtype=torch.float32)
        R16, R32 = rnd(M, N, seed=85), rnd(M, N, seed=86, dtype=torch.float32)
        ref = a.float() @ w.float().t() + bias + rowvec.repeat_interleave(rpv, 0)[:M]
        o = both(lambda: ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R16, tile_cfg=cfg))
        check(f"gemm tail split cfg{cfg} M={M} 16-bit", o, ref + R16.float(), 3e-2, 1e-2)
        o = both(lambda: ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R32, out_f32=True, tile_cfg=cfg))
        check(f"gemm tail split cfg{cfg} M={M} fp32 stream", o, ref + R32, 1e-2, 5e-3)
    if cfg != 21:                                          # 320-wide tiles do not take the GEGLU epilogue
        Hd = 640
        w1 = rnd(2 * Hd, K, scale=K ** -0.5, seed=87).float().cpu()
        b1 = rnd(2 * Hd, seed=88, dtype=torch.float32, scale=0.3).cpu()
        w1p, b1p = pack_geglu(w1, b1)
        w1p, b1p = w1p.to(BF16).cuda(), b1p.cuda()
        g = both(lambda: ops.gemm(a, w1p, bias=b1p, geglu=True, tile_cfg=cfg))
        h = a.float() @ w1.to(BF16).float().cuda().t() + b1.cuda()
        check(f"gemm tail split cfg{cfg} GEGLU", g, h[:, :Hd] * F.gelu(h[:, Hd:]), 3e-2, 1e-2)
    # 3x3 convolution view: 43 frames of 48 x 128 pixels = 264 192 rows = 1 032 tiles; 3-tap temporal view over the same rows (T = 43)
    Fr, cin, cout, H, W = 43, 64, 320, 48, 128
    x = rnd(Fr, cin, H, W, seed=89).float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=90).float()
    cb = rnd(cout, seed=91, dtype=torch.float32)
    tok = x.permute(0, 2, 3, 1).reshape(Fr * H * W, cin).to(BF16).contiguous()
    wc = pack_conv3x3(wt).to(BF16).cuda()
    o = both(lambda: ops.gemm(tok, wc, bias=cb, tile_cfg=cfg, conv=dict(cin=cin, hin=H, win=W, hout=H, wout=W, stride=1, ups=0, frames=Fr)))
    check(f"conv3x3 tail split cfg{cfg}", o, F.conv2d(x, wt, cb, padding=1).permute(0, 2, 3, 1).reshape(-1, cout), 3e-2, 1e-2)
    wt3 = rnd(cout, cin, 3, 1, 1, scale=(3 * cin) ** -0.5, seed=92).float()
    w3 = pack_tconv3(wt3).to(BF16).cuda()
    o = both(lambda: ops.gemm(tok, w3, bias=cb, temporal=dict(cin=cin, T=Fr, pix=H * W), tile_cfg=cfg))
    xt = x.permute(1, 0, 2, 3).reshape(1, cin, Fr, H * W, 1)
    check(f"temporal tail split cfg{cfg}", o, F.conv3d(xt, wt3, cb, padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0).reshape(-1, cout), 3e-2, 1e-2)


def test_gemm_k32_and_f32_out(ops):
    M, N, K = 513, 132, 96
    a, w = rnd(M, K, seed=4), rnd(N, K, scale=K ** -0.5, seed=5)
    out = ops.gemm(a, w, out_f32=True)
    assert out.dtype == torch.float32
    check("gemm K=96 (BK32) f32 out", out, a.float() @ w.float().t(), 1e-3, 1e-3)


def test_gemm_epilogues(ops):
    M, N, K, rpv = 640, 320, 320, 64
    a, w = rnd(M, K, seed=6), rnd(N, K, scale=K ** -0.5, seed=7)
    bias = rnd(N, seed=8, dtype=torch.float32)
    rowvec = rnd(M // rpv, N, seed=9, dtype=torch.float32)
    R, S = rnd(M, N, seed=10), rnd(M, N, seed=11)
    alpha = 0.3
    base = a.float() @ w.float().t() + bias + rowvec.repeat_interleave(rpv, 0) + R.float()
    out = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R)
    check("gemm +bias+rowvec+residual", out, base, 3e-2, 1e-2)
    out = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R, blend=(alpha, S))
    check("gemm blend", out, alpha * S.float() + (1 - alpha) * base, 3e-2, 1e-2)
    out = ops.gemm(a, w, bias=bias, silu=True)
    check("gemm silu", out, F.silu(a.float() @ w.float().t() + bias), 2e-2, 1e-2)


@pytest.mark.parametrize("cfg", [1, 2, 8, 17, 19, 20, 21, 22, 23])
@pytest.mark.parametrize("K", [32, 128, 704])
def test_gemm_many_tiles_per_workgroup(ops, cfg, K, M=33000):
    """More tiles than resident workgroups: exercises the persistent loop (tile prologue requested before the previous tile's epilogue,
    the static / dynamic vmcnt waits for 2-, 3- and 4-stage rings) and the aux-slot ring; K shorter / longer than the ring."""
    if K % 64 and cfg not in (17, 19, 23):
        pytest.skip("config needs K % 64 == 0")
    N, rpv = 512, 1000
    a, w = rnd(M, K, seed=30), rnd(N, K, scale=K ** -0.5, seed=31)
    bias = rnd(N, seed=32, dtype=torch.float32)
    rowvec = rnd((M + rpv - 1) // rpv, N, seed=33, dtype=torch.float32)
    R, S = rnd(M, N, seed=34), rnd(M, N, seed=35)
    mm = a.float() @ w.float().t()
    base = mm + bias + rowvec.repeat_interleave(rpv, 0)[:M] + R.float()
    out = ops.gemm(a, w, bias=bias, rowvec=rowvec, rows_per_vec=rpv, residual=R, tile_cfg=cfg)
    check(f"many tiles cfg{cfg} K{K} +bias+rowvec+residual", out, base, 3e-2, 1e-2)
    out = ops.gemm(a, w, bias=bias, residual=R, blend=(0.25, S), silu=True, tile_cfg=cfg)
    check(f"many tiles cfg{cfg} K{K} blend+silu", out, F.silu(0.25 * S.float() + 0.75 * (mm + bias + R.float())), 3e-2, 1e-2)
    out = ops.gemm(a, w, out_f32=True, tile_cfg=cfg)
    check(f"many tiles cfg{cfg} K{K} f32", out, mm, 2e-3, 2e-3)


@pytest.mark.parametrize("cfg", [0, 1, 2, 8, 9, 10, 11, 12, 17, 18, 19, 20, 23])
def test_gemm_geglu(ops, cfg):
    from streamingt2v_amd.video_model import pack_geglu
    M, C = 300, 320
    a = rnd(M, C, seed=12)
    w = rnd(8 * C, C, scale=C ** -0.5, seed=13).float().cpu()
    b = rnd(8 * C, seed=14, dtype=torch.float32).cpu()
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(a, wp.to(BF16).cuda(), bias=bp.cuda(), geglu=True, tile_cfg=cfg)
    h = a.float() @ w.to(BF16).float().cuda().t() + b.cuda()
    v, g = h.chunk(2, -1)
    check(f"gemm geglu cfg{cfg}", out, v * F.gelu(g), 3e-2, 1e-2)


@pytest.mark.parametrize("M", [1, 97, 128, 300, 33000 + 17])
def test_ff_geglu_fused(ops, M):
    """svd_ff_geglu_fused (csrc/ff_fused.hip; FeedForward of attention.py:94-120 at dim 320) in every residual / output / blend variant:
    (1) against the two-launch path it replaces (GEGLU-epilogue GEMM -> down-projection GEMM) to one rounding of the 16-bit hidden tile plus
    accumulation-order noise, (2) against a plain fp32 PyTorch evaluation.  M covers a single row, ragged tiles, many tiles per workgroup."""
    from streamingt2v_amd.video_model import pack_ff_fused, pack_geglu
    C, Hd = 320, 1280
    x = rnd(M, C, seed=41)
    w1 = rnd(2 * Hd, C, scale=C ** -0.5, seed=42).float().cpu()
    b1 = rnd(2 * Hd, seed=43, dtype=torch.float32, scale=0.3).cpu()
    w2 = rnd(C, Hd, scale=Hd ** -0.5, seed=44).float().cpu()
    b2 = rnd(C, seed=45, dtype=torch.float32, scale=0.3)
    img = pack_ff_fused(w1, b1, w2).cuda()
    assert img.numel() == ops._lib.svd_ff_fused_pack_bytes(Hd)
    w1p, b1p = pack_geglu(w1, b1)
    w1p, b1p, w2d = w1p.to(BF16).cuda(), b1p.cuda(), w2.to(BF16).cuda()
    h32 = x.float() @ w1.to(BF16).float().cuda().t() + b1.cuda()
    v, g = h32.chunk(2, -1)
    ff32 = (v * F.gelu(g)) @ w2.to(BF16).float().cuda().t() + b2
    r16, r32 = rnd(M, C, seed=46), rnd(M, C, seed=47, dtype=torch.float32)
    s16, s32 = rnd(M, C, seed=48), rnd(M, C, seed=49, dtype=torch.float32)
    alpha = 0.3125
    for res, blend, out_f32 in [(None, None, False), (None, None, True), (r16, None, False), (r16, None, True), (r32, None, True), (r32, None, False),
                                (r16, (alpha, s16), False), (r32, (alpha, s32), True)]:
        got = ops.ff_geglu_fused(x, img, Hd, b2, residual=res, blend=blend, out_f32=out_f32)
        hid = ops.gemm(x, w1p, bias=b1p, geglu=True)
        two = ops.gemm(hid, w2d, bias=b2, residual=res, blend=blend, out_f32=out_f32)
        ref = ff32 + (res.float() if res is not None else 0)
        if blend is not None:
            ref = alpha * blend[1].float() + (1 - alpha) * ref
        assert got.dtype == two.dtype == (torch.float32 if out_f32 else BF16)
        name = f"ff fused M={M} res={'n' if res is None else res.dtype} blend={blend is not None} out32={out_f32}"
        check(name + " vs two launches", got, two, 1.2e-2, 8e-3 if not out_f32 else 0.0)      # 16-bit output: one flipped output rounding (bf16: 2^-7 relative)
        check(name + " vs fp32 torch", got, ref, 3e-2, 1e-2)
    # per-frame vector (ABI v9: x + time_pos_embed enters as residual + row vector, video_attention.py:318-321): rows_per_vec = 96, a ragged last group
    rpv = 96
    rv = rnd((M + rpv - 1) // rpv, C + 8, seed=50, dtype=torch.float32)[:, :C]
    got = ops.ff_geglu_fused(x, img, Hd, b2, residual=r32, out_f32=True, rowvec=rv, rows_per_vec=rpv)
    check(f"ff fused M={M} + per-frame vector vs fp32 torch", got, ff32 + r32 + rv.repeat_interleave(rpv, 0)[:M], 3e-2, 1e-2)
    base = ops.ff_geglu_fused(x, img, Hd, b2, residual=r32, out_f32=True)
    check(f"ff fused M={M} + per-frame vector vs (without) + vector", got, base + rv.repeat_interleave(rpv, 0)[:M], 2e-6, 2e-6)
    # fused LayerNorm of the result (ABI v10: norm_in over x_spatial + time_pos_embed, norm1 behind ff_in; video_attention.py:125-168,318-321): the fp32 rows
    # must be bit-identical to the launch without it, the normalised rows equal svd_layernorm of them to one flipped 16-bit rounding
    lg, lb = rnd(C, seed=51, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=52, dtype=torch.float32) * 0.1
    av = rnd((M + rpv - 1) // rpv, C + 24, seed=53, dtype=torch.float32)[:, :C]
    for vec, add in [(None, None), (None, av), (rv, None), (rv, av)]:
        kw = dict(residual=r32, out_f32=True, rowvec=vec, rows_per_vec=rpv if vec is not None else 0)
        y, yn = ops.ff_geglu_fused(x, img, Hd, b2, ln=(lg, lb), ln_addvec=add, ln_rows_per_vec=rpv if add is not None else 0, **kw)
        y0 = ops.ff_geglu_fused(x, img, Hd, b2, **kw)
        assert torch.equal(y, y0) and yn.dtype == BF16 and yn.shape == (M, C)
        name = f"ff fused M={M} vec={vec is not None} + LayerNorm(addvec={add is not None})"
        check(name + " vs svd_layernorm", yn, ops.layernorm(y0, lg, lb, addvec=add, rows_per_vec=rpv if add is not None else 0), 1.2e-2, 8e-3)
        ref = y0 + (add.repeat_interleave(rpv, 0)[:M] if add is not None else 0)
        check(name + " vs fp32 torch", yn, F.layer_norm(ref, (C,), lg, lb, 1e-5), 1.2e-2, 8e-3)
        y2, yn2 = ops.ff_geglu_fused(x, img, Hd, b2, ln=(lg, lb), ln_addvec=add, ln_rows_per_vec=rpv if add is not None else 0, **kw)
        assert torch.equal(yn, yn2)                      # run to run: no atomics, fixed exchange order
    with pytest.raises(AssertionError):
        ops.ff_geglu_fused(x, img, Hd, b2, residual=r16, out_f32=True, ln=(lg, lb))          # the fused LayerNorm exists for the fp32 stream only


@pytest.mark.parametrize("M", [1, 31, 32, 97, 4096, 33000 + 17])
def test_rowgemm320(ops, M):
    """svd_rowgemm320 (csrc/rowgemm.hip; the 320 -> 320 projections of SpatialVideoTransformer with the LayerNorm behind them, video_attention.py:260-333,
    attention.py:528-530,567-593) in every variant the host uses: (1) against the launches it replaces -- svd_gemm (+ fp32 residual, per-frame vector) and
    svd_layernorm of its fp32 output -- to fp32 summation order / one flipped 16-bit rounding, (2) against plain fp32 PyTorch.  M covers a single row, ragged
    32-row tiles (the lanes past M store duplicates of row M - 1), many tiles per wave.  An ASYMMETRIC weight catches a transposed fragment."""
    from streamingt2v_amd.video_model import pack_rowgemm320
    C = 320
    x = rnd(M, C, seed=61)
    w = rnd(C, C, scale=C ** -0.5, seed=62).float().cpu()
    w[:, :7] *= 3.0; w[5:9] *= 0.25                                       # rows and columns of different scale
    bias = rnd(C, seed=63, dtype=torch.float32, scale=0.3)
    g, b = rnd(C, seed=64, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=65, dtype=torch.float32) * 0.1
    r32 = rnd(M, C, seed=66, dtype=torch.float32)
    rpv = 32 * 3
    nvec = (M + rpv - 1) // rpv
    rv = rnd(nvec, C + 8, seed=67, dtype=torch.float32)[:, :C]             # a row stride that is not the channel count
    img = pack_rowgemm320(w).cuda()
    assert img.numel() == ops._lib.svd_rowgemm320_pack_bytes()
    wd = w.to(BF16).cuda()
    mm = x.float() @ wd.float().t()
    for res, vec, ln, out_f32 in [(None, None, True, True), (r32, rv, True, True), (r32, None, False, True), (None, None, False, False), (r32, rv, False, True),
                                  (None, rv, True, True)]:
        y, yn = ops.rowgemm320(x, img, bias=bias, rowvec=vec, rows_per_vec=rpv if vec is not None else 0, residual=res, out_f32=out_f32,
                               ln=(g, b) if ln else None)
        ref = mm + bias + (res if res is not None else 0) + (vec.repeat_interleave(rpv, 0)[:M] if vec is not None else 0)
        two = ops.gemm(x, wd, bias=bias, rowvec=vec, rows_per_vec=rpv if vec is not None else 0, residual=res, out_f32=out_f32)
        name = f"rowgemm320 M={M} res={res is not None} vec={vec is not None} ln={ln} out32={out_f32}"
        assert y.dtype == two.dtype == (torch.float32 if out_f32 else BF16) and y.shape == (M, C)
        # fp32 output: summation order only -- plus, with a per-frame vector, its hi + lo 16-bit split (2^-22 relative in fp16, 2^-16 in bf16: |vec| <= 5 here)
        a32 = 2e-5 + (1e-4 if (vec is not None and BF16 == torch.bfloat16) else 0.0)
        check(name + " vs svd_gemm", y, two, a32 if out_f32 else 1.2e-2, 2e-5 if out_f32 else 8e-3)
        check(name + " vs fp32 torch", y, ref, a32 if out_f32 else 1.2e-2, 2e-5 if out_f32 else 8e-3)
        if ln:
            assert yn.dtype == BF16 and yn.shape == (M, C)
            check(name + " LayerNorm vs svd_layernorm(svd_gemm)", yn, ops.layernorm(two, g, b), 1.2e-2, 8e-3)
            check(name + " LayerNorm vs fp32 torch", yn, F.layer_norm(ref, (C,), g, b, 1e-5), 1.2e-2, 8e-3)
        else:
            assert yn is None
    # LayerNorm only (no Y): the CAM-style use; and run-to-run bit identity (independent waves, no atomics)
    _, yn1 = ops.rowgemm320(x, img, bias=bias, residual=r32, ln=(g, b), want_y=False)
    y2, yn2 = ops.rowgemm320(x, img, bias=bias, residual=r32, ln=(g, b))
    assert torch.equal(yn1, yn2)
    y3, yn3 = ops.rowgemm320(x, img, bias=bias, residual=r32, ln=(g, b))
    assert torch.equal(y2, y3) and torch.equal(yn2, yn3)


@pytest.mark.parametrize("M", [1, 127, 128, 300, 33000 + 17])
@pytest.mark.parametrize("N", [64, 640, 960])
def test_rowproj320(ops, M, N):
    """svd_rowproj320 (csrc/rowproj.hip; the q | k and q | k | v projections of the 320-channel transformer blocks, attention.py:246-262, video_attention.py:125-168):
    (1) against svd_gemm, the launch it replaces -- same operands, fp32 accumulation in a different order, one 16-bit output rounding: at most one flipped rounding;
    (2) against plain fp32 PyTorch.  An ASYMMETRIC weight catches a transposed fragment or a wrong channel interleave; M covers a single row, ragged 128-row tiles
    (rows past M store duplicates of row M - 1), many tiles per workgroup; a strided destination (a column range of a wider buffer); run-to-run bit identity."""
    from streamingt2v_amd.video_model import pack_rowproj320
    x = rnd(M, 320, seed=71)
    w = rnd(N, 320, scale=320 ** -0.5, seed=72).float().cpu()
    w[:, :5] *= 3.0; w[3:7] *= 0.25; w[N - 1] *= 2.0
    bias = rnd(N, seed=73, dtype=torch.float32, scale=0.3)
    img = pack_rowproj320(w).cuda()
    assert img.numel() == ops._lib.svd_rowproj320_pack_bytes(N)
    wd = w.to(BF16).cuda()
    ref = x.float() @ wd.float().t()
    for b in (None, bias):
        y = ops.rowproj320(x, img, N, bias=b)
        two = ops.gemm(x, wd, bias=b)
        assert y.dtype == BF16 and y.shape == (M, N)
        check(f"rowproj320 M={M} N={N} bias={b is not None} vs svd_gemm", y, two, 1.2e-2, 8e-3)
        check(f"rowproj320 M={M} N={N} bias={b is not None} vs fp32 torch", y, ref + (b if b is not None else 0), 1.2e-2, 8e-3)
    wide = torch.full((M, N + 70), 7.0, dtype=BF16, device="cuda")
    y2 = ops.rowproj320(x, img, N, bias=bias, out=wide[:, 6:6 + N])
    assert torch.equal(y2, y) and torch.equal(wide[:, :6], torch.full_like(wide[:, :6], 7.0)) and torch.equal(wide[:, 6 + N:], torch.full_like(wide[:, 6 + N:], 7.0))
    assert torch.equal(ops.rowproj320(x, img, N, bias=bias), y)


def test_geglu_gate_function_against_exact_erf(ops):
    """The GEGLU epilogue's gate function alone, on every 16-bit gate value in [-9.5, 9.5]: value = 1 (bias only), gate = x through a unit
    weight, so the output is rn16(gelu(x)) -- the exact-erf GELU of the reference's GEGLU (attention.py:99-101) as csrc/svd_common.h
    gelu_erf_f2 evaluates it (x Phi(x) through q = 2^(r(|x|) |x| - 1), one v_exp_f32), to well inside one output rounding, tails and
    16-bit subnormals included.  (Round 4 also ran this test on an LDS-table form of Phi, |Phi error| <= 2e-6: the 4e-6 |x| slack is that form's.)"""
    from streamingt2v_amd.video_model import pack_geglu
    M, K, n_out = 40960, 64, 64
    x = torch.linspace(-9.5, 9.5, M).to(BF16)
    a = torch.zeros(M, K, dtype=BF16); a[:, 0] = x
    w = torch.zeros(2 * n_out, K); w[n_out:, 0] = 1.0
    b = torch.zeros(2 * n_out); b[:n_out] = 1.0
    wp, bp = pack_geglu(w, b)
    exact = (x.double() * 0.5 * torch.erfc(-x.double() / math.sqrt(2.0)))
    for cfg in (0, 2, 8, 20):
        out = ops.gemm(a.cuda(), wp.to(BF16).cuda(), bias=bp.cuda(), geglu=True, tile_cfg=cfg).double().cpu()
        assert out.shape == (M, n_out) and torch.equal(out[:, :1].expand(-1, n_out), out)
        err = (out[:, 0] - exact).abs()
        mant, emin = (7, -126) if BF16 == torch.bfloat16 else (10, -14)   # bound: half a spacing of the output type at the exact value + the table's 4e-6 |x|
        spacing = torch.exp2(torch.floor(torch.log2(exact.abs().clamp(min=1e-300))).clamp(min=emin) - mant)
        bound = 0.5 * spacing + 4e-6 * x.double().abs().clamp(min=1.0)
        worst = (err - bound).max().item()
        print(f"[geglu gate function cfg{cfg}] max abs err {err.max().item():.3e}, worst margin over (one output rounding + 4e-6 |x|) {worst:.3e}")
        assert worst <= 0, (cfg, worst)


@pytest.mark.parametrize("stride,ups,cin,cout,H,W", [(1, 0, 64, 128, 9, 16), (2, 0, 64, 64, 18, 32), (1, 1, 128, 64, 9, 16),
                                                     (1, 0, 32, 96, 12, 20), (1, 0, 320, 320, 16, 16)])
def test_gemm_conv3x3(ops, stride, ups, cin, cout, H, W):
    from streamingt2v_amd.video_model import pack_conv3x3
    Fr = 3
    x = rnd(Fr, cin, H, W, seed=15).float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=16).float()
    bias = rnd(cout, seed=17, dtype=torch.float32)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xin, wt, bias, stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    tok = x.permute(0, 2, 3, 1).reshape(Fr * H * W, cin).to(BF16).contiguous()
    wp = pack_conv3x3(wt).to(BF16).cuda()
    out = ops.gemm(tok, wp, bias=bias, conv=dict(cin=cin, hin=H, win=W, hout=ho, wout=wo, stride=stride, ups=ups, frames=Fr))
    check(f"conv3x3 s{stride} u{ups} {cin}->{cout}", out, ref.permute(0, 2, 3, 1).reshape(-1, cout), 3e-2, 1e-2)


@pytest.mark.parametrize("cfg", [1, 2, 8, 19, 20, 21, 22, 23])
def test_gemm_implicit_views_many_tiles(ops, cfg):
    """conv3x3 / temporal implicit GEMMs with more tiles than resident workgroups, per tile configuration
    (persistent loop across tiles) + residual + per-frame vector."""
    from streamingt2v_amd.video_model import pack_conv3x3, pack_tconv3
    Fr, cin, cout, H, W = 10, 64, 128, 72, 96
    x = rnd(Fr, cin, H, W, seed=40).float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=41).float()
    bias = rnd(cout, seed=42, dtype=torch.float32)
    rowvec = rnd(Fr, cout, seed=43, dtype=torch.float32)
    ref = F.conv2d(x, wt, bias, padding=1) + rowvec[:, :, None, None]
    tok = x.permute(0, 2, 3, 1).reshape(Fr * H * W, cin).to(BF16).contiguous()
    R = rnd(Fr * H * W, cout, seed=44)
    out = ops.gemm(tok, pack_conv3x3(wt).to(BF16).cuda(), bias=bias, rowvec=rowvec, rows_per_vec=H * W, residual=R, tile_cfg=cfg,
                   conv=dict(cin=cin, hin=H, win=W, hout=H, wout=W, stride=1, ups=0, frames=Fr))
    check(f"conv3x3 many tiles cfg{cfg}", out, ref.permute(0, 2, 3, 1).reshape(-1, cout) + R.float(), 3e-2, 1e-2)
    B, C, T, pix = 2, 128, 5, 72 * 96
    xt = rnd(B, C, T, pix, 1, seed=45).float()
    wt3 = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=46).float()
    ref = F.conv3d(xt, wt3, bias, padding=(1, 0, 0))
    tok = xt[..., 0].permute(0, 2, 3, 1).reshape(B * T * pix, C).to(BF16).contiguous()
    out = ops.gemm(tok, pack_tconv3(wt3).to(BF16).cuda(), bias=bias, temporal=dict(cin=C, T=T, pix=pix), tile_cfg=cfg)
    check(f"temporal many tiles cfg{cfg}", out, ref[..., 0].permute(0, 2, 3, 1).reshape(-1, C), 3e-2, 1e-2)


@pytest.mark.parametrize("C,T,pix", [(64, 8, 40), (320, 25, 16), (32, 4, 64)])
def test_gemm_temporal3(ops, C, T, pix):
    from streamingt2v_amd.video_model import pack_tconv3
    B = 2
    x = rnd(B, C, T, pix, 1, seed=18).float()
    wt = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=19).float()
    bias = rnd(C, seed=20, dtype=torch.float32)
    ref = F.conv3d(x, wt, bias, padding=(1, 0, 0))                                  # b c t p 1
    tok = x[..., 0].permute(0, 2, 3, 1).reshape(B * T * pix, C).to(BF16).contiguous()   # (b t p) c
    out = ops.gemm(tok, pack_tconv3(wt).to(BF16).cuda(), bias=bias, temporal=dict(cin=C, T=T, pix=pix))
    check(f"temporal conv C{C} T{T}", out, ref[..., 0].permute(0, 2, 3, 1).reshape(-1, C), 3e-2, 1e-2)


def test_gemm_trans_out(ops):
    Fr, pix, C, K = 3, 144, 128, 128
    a, w = rnd(Fr * pix, K, seed=21), rnd(C, K, scale=K ** -0.5, seed=22)
    tok_ld = 192
    vt = torch.zeros(Fr, C, tok_ld, dtype=BF16, device="cuda")
    ops.gemm(a, w, trans_out=dict(tok_per_frame=pix, tokens_ld=tok_ld, out=vt))
    ref = (a.float() @ w.float().t()).view(Fr, pix, C).transpose(1, 2)
    check("gemm transposed out", vt[:, :, :pix], ref, 2e-2, 1e-2)
    assert float(vt[:, :, pix:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("Fr,N,heads", [(2, 256, 5), (1, 144, 20), (3, 576, 2), (1, 2304, 3), (1, 1000, 1)])
def test_attn_spatial(ops, Fr, N, heads):
    C = heads * 64
    qk = rnd(Fr * N, 2 * C, seed=23)
    v = rnd(Fr * N, C, seed=24)
    tok_ld = (N + 63) // 64 * 64
    vt = torch.zeros(Fr, C, tok_ld, dtype=BF16, device="cuda")
    vt[:, :, :N] = v.view(Fr, N, C).transpose(1, 2)
    out = torch.empty(Fr * N, C, dtype=BF16, device="cuda")
    ops.attn_spatial(qk[:, :C], qk[:, C:], vt, out, Fr, N, heads)
    q, k = (t.float().view(Fr, N, heads, 64).transpose(1, 2) for t in (qk[:, :C], qk[:, C:]))
    vv = v.float().view(Fr, N, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q, k, vv).transpose(1, 2).reshape(Fr * N, C)
    check(f"attn spatial F{Fr} N{N} h{heads}", out, ref, 2e-2, 2e-2)


def test_attn_spatial_online_softmax_rescale(ops):
    """Force the running max to jump at a late KV tile (spiked key) -- exercises the O rescale path."""
    N, C = 512, 64
    qk = rnd(N, 2 * C, seed=25)
    qk[300, C:] = qk[5, :C] * 4.0                      # key 300 strongly matches query 5
    v = rnd(N, C, seed=26)
    vt = torch.zeros(1, C, N, dtype=BF16, device="cuda"); vt[0] = v.t()
    out = torch.empty(N, C, dtype=BF16, device="cuda")
    ops.attn_spatial(qk[:, :C], qk[:, C:], vt, out, 1, N, 1)
    ref = F.scaled_dot_product_attention(qk[None, None, :, :C].float(), qk[None, None, :, C:].float(), v[None, None].float())[0, 0]
    check("attn spatial spiked key", out, ref, 2e-2, 2e-2)


@pytest.mark.parametrize("B,Tq,Tk,pix,heads", [(2, 25, 25, 33, 5), (2, 25, 7, 20, 10), (1, 7, 7, 9, 20), (2, 8, 3, 16, 5),
                                                 (1, 32, 32, 7, 2), (2, 25, 17, 9, 5), (1, 25, 32, 4, 3), (3, 1, 1, 5, 1), (2, 25, 25, 2304, 10)])
def test_attn_temporal(ops, B, Tq, Tk, pix, heads):
    C = heads * 64
    q = rnd(B * Tq * pix, C, seed=27)
    kv = rnd(B * Tk * pix, 2 * C, seed=28)
    out = torch.empty(B * Tq * pix, C, dtype=BF16, device="cuda")
    ops.attn_temporal(q, kv[:, :C], kv[:, C:], out, B, Tq, Tk, pix, heads)

    def bsthd(t, T):   # (b t p) (h d) -> (b p) h t d
        return t.float().view(B, T, pix, heads, 64).permute(0, 2, 3, 1, 4).reshape(B * pix, heads, T, 64)
    ref = F.scaled_dot_product_attention(bsthd(q, Tq), bsthd(kv[:, :C], Tk), bsthd(kv[:, C:], Tk))
    ref = ref.view(B, pix, heads, Tq, 64).permute(0, 3, 1, 2, 4).reshape(B * Tq * pix, C)
    check(f"attn temporal B{B} {Tq}x{Tk}", out, ref, 2e-2, 2e-2)


def test_softmax_rows(ops):
    s = rnd(300, 1024, scale=3.0, seed=29, dtype=torch.float32)
    p = torch.empty(300, 1024, dtype=BF16, device="cuda")
    ops.softmax_rows(s, p, 0.25)
    check("softmax rows", p, torch.softmax(s * 0.25, -1), 1e-4, 1e-2)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("Fr,pix,C,fps,silu", [(4, 144, 320, 1, True), (6, 100, 640, 3, True), (2, 64, 2560, 1, False),
                                              (4, 576, 32, 4, True), (2, 36, 960, 2, False), (2, 4096, 128, 1, True)])
def test_groupnorm(ops, Fr, pix, C, fps, silu):
    x = (rnd(Fr * pix, C, seed=30).float() * 1.5 + 0.7).to(BF16)
    g, b = rnd(C, seed=31, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=32, dtype=torch.float32) * 0.1
    out = ops.groupnorm(x, Fr, pix, g, b, 1e-5, frames_per_stat=fps, silu=silu)
    x5 = x.float().view(Fr // fps, fps * pix, C).transpose(1, 2)                   # [stat batch, C, fps*pix]
    ref = F.group_norm(x5, 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    check(f"groupnorm C{C} fps{fps}", out, ref.transpose(1, 2).reshape(Fr * pix, C), 2e-2, 1e-2)


@pytest.mark.parametrize("C", [32, 96, 256, 320, 512, 640, 1024, 1280, 2048])
def test_layernorm(ops, C):
    rows, rpv = 1000, 250
    x = rnd(rows, C, seed=33)
    g, b = rnd(C, seed=34, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=35, dtype=torch.float32) * 0.1
    out = ops.layernorm(x, g, b)
    check(f"layernorm C{C}", out, F.layer_norm(x.float(), (C,), g, b, 1e-5), 2e-2, 1e-2)
    av = rnd(rows // rpv, C, seed=36, dtype=torch.float32)
    out, xs = ops.layernorm(x, g, b, addvec=av, rows_per_vec=rpv, want_sum=True, silu=True)
    xsum = x.float() + av.repeat_interleave(rpv, 0)
    check(f"layernorm+add C{C} sum", xs, xsum, 2e-2, 1e-2)
    check(f"layernorm+add+silu C{C}", out, F.silu(F.layer_norm(xsum, (C,), g, b, 1e-5)), 2e-2, 1e-2)


# ------------------------------------------------------------------------------------------------ glue
def test_layout_and_glue(ops):
    Fr, h, w = 3, 6, 10
    x0, x1 = rnd(Fr, 4, h, w, seed=37, dtype=torch.float32), rnd(Fr, 4, h, w, seed=38, dtype=torch.float32)
    sc = rnd(Fr, seed=39, dtype=torch.float32)
    tok = ops.nchw_to_tokens(x0, x1, sc, 32)
    ref = torch.zeros(Fr, h * w, 32, device="cuda")
    ref[..., :4] = (x0 * sc[:, None, None, None]).flatten(2).transpose(1, 2)
    ref[..., 4:8] = x1.flatten(2).transpose(1, 2)
    check("nchw_to_tokens", tok, ref.view(-1, 32), 1e-6, 2 ** -8 * (1 if BF16 == torch.bfloat16 else 0.125))
    back = ops.tokens_to_nchw(tok, 8, Fr, h, w)
    check("tokens_to_nchw", back, ref.view(Fr, h * w, 32)[..., :8].transpose(1, 2).reshape(Fr, 8, h, w).to(BF16), 0, 0)
    a, b = rnd(50, 64, seed=40), rnd(50, 128, seed=41)
    check("concat", ops.concat_channels(a, b), torch.cat([a, b], 1), 0, 0)
    c = rnd(50, 64, seed=42)
    check("add_rows", ops.add_rows(a, c), a.float() + c.float(), 1e-6, 2 ** -8 * (1 if BF16 == torch.bfloat16 else 0.125))
    t = torch.tensor([0.0, 1.5, -0.7, 12.0], device="cuda")
    emb = ops.timestep_embedding(t, 320)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half).cuda()
    args = t[:, None] * freqs[None]
    check("timestep_embedding", emb, torch.cat([args.cos(), args.sin()], -1), 1e-3, 2 ** -8 * (1 if BF16 == torch.bfloat16 else 0.125))
    v = rnd(7, 33, seed=43, dtype=torch.float32)
    check("silu->bf16", ops.to_bf16(v, silu=True), F.silu(v), 1e-6, 2 ** -8 * (1 if BF16 == torch.bfloat16 else 0.125))


def test_edm_euler_step(ops):
    T, C, h, w = 5, 4, 6, 8
    x = rnd(T, C, h, w, seed=44, dtype=torch.float32) * 10
    net = rnd(2 * T * h * w, 4, seed=45, dtype=torch.float32)
    gs = torch.linspace(1.5, 3.0, T).cuda()
    sigma, nxt = 7.5, 2.25
    x_ref = x.clone()
    n = net.view(2, T, h * w, C).permute(0, 1, 3, 2).reshape(2, T, C, h, w)
    c_skip, c_out = 1 / (sigma ** 2 + 1), -sigma / math.sqrt(sigma ** 2 + 1)
    du, dc = n[0] * c_out + x_ref * c_skip, n[1] * c_out + x_ref * c_skip
    den = du + gs[:, None, None, None] * (dc - du)
    x_ref = x_ref + (x_ref - den) / sigma * (nxt - sigma)
    ops.edm_euler_step(x, net, gs, sigma, nxt)
    check("edm euler step", x, x_ref, 1e-4, 1e-5)


def test_ae_time_mix3(ops):
    Fr, h, w = 4, 5, 7
    x = rnd(Fr * h * w, 4, seed=46, dtype=torch.float32)
    wt, b = rnd(3, 3, 3, seed=47, dtype=torch.float32), rnd(3, seed=48, dtype=torch.float32)
    out = ops.ae_time_mix3(x, wt, b, Fr, h, w, True)
    x5 = x[:, :3].view(1, Fr, h * w, 3).permute(0, 3, 1, 2)[..., None]          # b c t p 1
    ref = F.conv3d(x5, wt[..., None, None], b, padding=(1, 0, 0))[0, ..., 0].permute(1, 0, 2).reshape(Fr, 3, h, w)
    check("ae time mix + clamp", out, ref.clamp(-1, 1), 1e-5, 1e-5)


# ------------------------------------------------------------------------------------------------ extended-precision rim (csrc/precision.hip)
def _join3(s3, C):
    """split-3 rows [hi | lo | hi] -> the fp32 values they represent."""
    return s3[:, :C].float() + s3[:, C:2 * C].float()


@pytest.mark.parametrize("C", [32, 96, 256, 320, 512])
def test_rows_split3(ops, C):
    rows = 777
    x = rnd(rows, C, seed=50, dtype=torch.float32) * 3
    eps16 = 2.0 ** -8 if BF16 == torch.bfloat16 else 2.0 ** -11
    s3 = ops.rows_split3(x)
    assert torch.equal(s3[:, :C], s3[:, 2 * C:]) and torch.equal(s3[:, :C], x.to(BF16))
    err = (_join3(s3, C) - x).abs().max().item()
    print(f"[rows_split3 C{C}] max |hi + lo - x| = {err:.3e} (one 16-bit rounding would be ~{3 * 4 * eps16:.1e})")
    assert err <= 16 * eps16 * eps16 * 4 + 1e-7          # lo resolves the residual to its own 16-bit rounding (+ fp16 subnormal floor)
    g, b = rnd(C, seed=51, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=52, dtype=torch.float32) * 0.1
    s3 = ops.rows_split3(x, ln=(g, b), eps=1e-5, silu=True)
    ref = F.silu(F.layer_norm(x, (C,), g, b, 1e-5))
    err = (_join3(s3, C) - ref).abs().max().item()
    print(f"[rows_split3 C{C} LN + SiLU] max abs err {err:.3e}")
    assert err <= (2e-4 if BF16 == torch.bfloat16 else 5e-6)          # hi + lo: 16 significant bits in bf16, 22 in fp16


@pytest.mark.parametrize("cin,cout,stride,H,W", [(3, 32, 1, 24, 40), (32, 96, 2, 24, 40), (96, 96, 1, 12, 20), (256, 512, 2, 12, 20), (8, 320, 1, 9, 16)])
def test_x3_convolution_matches_fp32(ops, cin, cout, stride, H, W):
    from streamingt2v_amd.video_model import _Conv
    Fr = 3
    prev = ops.EXACT_RIM
    ops.set_precision_plan(exact_rim=True)
    try:
        conv = _Conv("c.", cin, cout, stride=stride, x3=True)
        g = torch.Generator(); g.manual_seed(cin * 7 + cout)
        sd = {"c.weight": torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5, "c.bias": torch.randn(cout, generator=g) * 0.1}
        conv.prepare(sd, "cuda")
        x = torch.randn(Fr, cin, H, W, generator=g).cuda()
        ref = F.conv2d(x.double(), sd["c.weight"].double().cuda(), sd["c.bias"].double().cuda(), stride=stride, padding=1).float()
        refm = ref.permute(0, 2, 3, 1).reshape(-1, cout)
        s3 = ops.nchw_to_tokens_x3(x, None, None, conv.cin_pad)
        out, ho, wo = conv.forward(s3, Fr, H, W, split3=True, out_f32=True)          # the layout is stated by the caller, never inferred from the width
        e3 = ((out[:, :cout] - refm).pow(2).mean().sqrt() / refm.pow(2).mean().sqrt()).item()
        out16, _, _ = conv.forward(ops.nchw_to_tokens(x, None, None, conv.cin_pad), Fr, H, W, out_f32=True)
        e1 = ((out16[:, :cout] - refm).pow(2).mean().sqrt() / refm.pow(2).mean().sqrt()).item()
        print(f"[x3 conv {cin}->{cout} s{stride}] relative L2 vs fp64: split-3 {e3:.2e}, plain 16-bit operands {e1:.2e}")
        assert e3 < (2e-5 if BF16 == torch.bfloat16 else 2e-6) and e3 < e1 / 50
        # fp32 rows in -> the convolution splits them itself
        if cin % 32 == 0:
            xr = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
            out2, _, _ = conv.forward(xr, Fr, H, W, out_f32=True)
            assert torch.equal(out2, out)
    finally:
        ops.set_precision_plan(exact_rim=prev)


def test_add_rows_f32b_and_head(ops):
    a, b = rnd(300, 320, seed=60), rnd(300, 320, seed=61, dtype=torch.float32)
    eps16 = 2.0 ** -8 if BF16 == torch.bfloat16 else 2.0 ** -11
    check("add_rows_f32b 16-bit x", ops.add_rows_f32b(a, b), a.float() + b, 1e-6, eps16)
    a32 = a.float() * 1.37
    assert torch.equal(ops.add_rows_f32b(a32, b), a32 + b)
    for Fr, H, W, C, cout in ((3, 9, 16, 320, 4), (2, 17, 37, 64, 4), (1, 8, 32, 32, 3)):
        x = (rnd(Fr * H * W, C, seed=62).float() * 1.5 + 0.3).to(BF16)
        g, be = rnd(C, seed=63, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=64, dtype=torch.float32) * 0.1
        w = rnd(cout, C, 3, 3, seed=65, dtype=torch.float32) * (9 * C) ** -0.5
        bias = rnd(4, seed=66, dtype=torch.float32) * 0.1
        wt = torch.zeros(3, 3, C, 4, device="cuda"); wt[..., :cout] = w.permute(2, 3, 1, 0)
        for xin in (x, x.float()):
            out = ops.head_gn_silu_conv3x3(xin, Fr, H, W, g, be, 1e-5, wt.reshape(9, C, 4).contiguous(), bias, cout)
            v = xin.float().reshape(Fr, H * W, C).transpose(1, 2).reshape(Fr, C, H, W)
            ref = F.conv2d(F.silu(F.group_norm(v, 32, g, be, 1e-5)), w, bias[:cout], padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
            err = (out[:, :cout] - ref).abs().max().item()
            print(f"[head GN+SiLU+conv3x3 {Fr}x{H}x{W}x{C}->{cout}, input {xin.dtype}] max abs err {err:.3e} (ref absmax {ref.abs().max().item():.2f})")
            assert err < 2e-5 * max(1.0, ref.abs().max().item())
