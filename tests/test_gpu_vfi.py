"""EMA-VFI on the GPU (-m gpu): every kernel of csrc/vfi.hip against its fp32 torch statement (tests/vfi_shim.py),
then the whole network against the CPU oracle (oracle/vfi_oracle.py, pinned bit-exactly to the vendored network) and the committed
golden vector of the vendored network's fast-TTA output (tests/golden/vfi_tiny.pt, oracle/make_golden_vfi.py).

Tolerances.  Kernels: one 16-bit rounding of the result (bf16 4e-3 / fp16 5e-4 relative) on top of fp32 arithmetic; fp32 kernels 1e-5.
Network (16-bit activations, fp32 accumulation; measured with 16-bit emulation on CPU: bf16 flow 0.05 px / pred 6e-3, fp16 8x smaller):
features rel. L2 <= 3e-2 (bf16) / 6e-3 (fp16), flow <= 0.15 px / 0.03 px, prediction <= 2e-2 / 4e-3 abs (1/255 = 3.9e-3).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vfi_shim as S  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vfi_tiny.pt")
ELEM = torch.bfloat16
TOLF = 1.0
DEV = "cuda"


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def ops(request):
    global ELEM, TOLF
    from streamingt2v_amd import ops as o
    ELEM = request.param
    TOLF = 1.0 if request.param == torch.bfloat16 else 0.2
    o.set_element_dtype(request.param)
    yield o
    o.set_element_dtype(None)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(); g.manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def close(name, got, ref, atol, rtol, scale_tol=True):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    f = TOLF if scale_tol else 1.0
    print(f"[{name} {str(ELEM)[6:]}] max abs err {err.max():.3e} (ref absmax {ref.abs().max():.3f})")
    assert torch.isfinite(got).all() and (err - f * (atol + rtol * ref.abs())).max().item() <= 0, f"{name}: err {err.max():.4e}"


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("rows,C", [(77, 32), (1000, 96)])
def test_prelu(ops, rows, C):
    x, a = rnd(rows, C, seed=1).to(ELEM), rnd(C, seed=2) * 0.3 + 0.25
    got = ops.prelu_(x.to(DEV).clone(), a.to(DEV))
    close("prelu", got, S.prelu_(x.clone(), a), 1e-6, 4e-3)
    xf = rnd(rows, 8, seed=3)
    close("prelu f32", ops.prelu_(xf.to(DEV).clone(), a[:8].contiguous().to(DEV)), S.prelu_(xf.clone(), a[:8]), 1e-7, 1e-6, scale_tol=False)


@pytest.mark.parametrize("frames,h,w,C", [(2, 5, 7, 64), (1, 6, 10, 256), (4, 3, 5, 32)])
def test_dwconv3x3_gelu(ops, frames, h, w, C):
    x, w9, b = rnd(frames * h * w, C, seed=1).to(ELEM), rnd(9, C, seed=2) * 0.4, rnd(C, seed=3) * 0.1
    got = ops.dwconv3x3_gelu(x.to(DEV), w9.to(DEV), b.to(DEV), frames, h, w)
    close("dwconv3x3_gelu", got, S.dwconv3x3_gelu(x.float(), w9, b, frames, h, w), 2e-3, 4e-3)


@pytest.mark.parametrize("n_win,heads,mph,n_mask", [(8, 2, 8, 4), (4, 4, 8, 0), (12, 2, 16, 2), (2, 1, 4, 1)])
def test_window_attn(ops, n_win, heads, mph, n_mask):
    C, md, R = heads * 32, heads * mph, n_win * 49
    q, kv, ce = rnd(R, C, seed=1).to(ELEM), rnd(R, 2 * C, seed=2).to(ELEM), rnd(R, md, seed=3)
    mask = None
    if n_mask:
        g = torch.Generator(); g.manual_seed(5)
        mask = torch.where(torch.rand(n_mask, 49, 49, generator=g) < 0.3, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()
    ox, oc = ops.window_attn_7x7(q.to(DEV), kv.to(DEV), ce.to(DEV), mask.to(DEV) if mask is not None else None, n_win, heads, mph, 32 ** -0.5)
    rx, rc = S.window_attn_7x7(q.float(), kv.float(), ce, mask, n_win, heads, mph, 32 ** -0.5)
    close("window_attn x", ox, rx, 2e-3, 4e-3)
    close("window_attn motion", oc, rc, 2e-3, 4e-3)


@pytest.mark.parametrize("frames,h,w,C,f32", [(2, 9, 13, 3, True), (1, 16, 24, 32, False), (2, 6, 10, 64, False), (1, 5, 4, 1, True)])
def test_warp_bilinear(ops, frames, h, w, C, f32):
    x = rnd(frames * h * w, C, seed=1) if f32 else rnd(frames * h * w, C, seed=1).to(ELEM)
    flow8 = rnd(frames * h * w, 8, seed=2, scale=3.0)                      # +-9 px: plenty of samples beyond the border
    flow8[:7, 2:4] = torch.tensor([[0.0, 0.0], [0.5, 0.0], [0.0, 0.5], [-100.0, 100.0], [100.0, -100.0], [1.0, 1.0], [-0.25, 0.75]])
    got = ops.warp_bilinear(x.to(DEV), flow8.to(DEV)[:, 2:4], frames, h, w)
    ref = S.warp_bilinear(x.float(), flow8[:, 2:4], frames, h, w)
    if f32:
        close("warp f32", got, ref, 3e-4, 1e-5, scale_tol=False)             # x + flow is formed directly, not through the normalised grid (~1e-5 px)
    else:
        close("warp", got, ref, 1e-3, 4e-3)


@pytest.mark.parametrize("frames,hin,win,C,sf", [(2, 12, 20, 6, 0.25), (1, 6, 10, 4, 0.5), (2, 3, 5, 8, 4.0), (1, 7, 9, 8, 2.0), (1, 45, 80, 4, 0.5)])
def test_resize_bilinear(ops, frames, hin, win, C, sf):
    x, mult = rnd(frames * hin * win, C, seed=1), rnd(C, seed=2)
    got, ho, wo = ops.resize_bilinear(x.to(DEV), frames, hin, win, sf, mult=mult.to(DEV))
    ref, rh, rw = S.resize_bilinear(x, frames, hin, win, sf, mult=mult)
    assert (ho, wo) == (rh, rw)
    close("resize", got, ref, 1e-5, 1e-5, scale_tol=False)
    acc = rnd(frames * ho * wo, 8, seed=3)
    got2, _, _ = ops.resize_bilinear(x.to(DEV), frames, hin, win, sf, out=acc.to(DEV).clone(), accumulate=True)
    ref2, _, _ = S.resize_bilinear(x, frames, hin, win, sf, out=acc.clone(), accumulate=True)
    close("resize accumulate", got2, ref2, 1e-5, 1e-5, scale_tol=False)


def test_merge_and_tta_average(ops):
    h, w = 12, 20
    n = 2 * h * w
    g = torch.Generator(); g.manual_seed(9)
    w0, w1, fm, u = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g), rnd(n, 8, seed=1), rnd(n, 8, seed=2)
    pred, merged = ops.vfi_merge(w0.to(DEV), w1.to(DEV), fm.to(DEV)[:, 4:5], u.to(DEV), want_merged=True)
    rp, rm = S.vfi_merge(w0, w1, fm[:, 4:5], u, want_merged=True)
    close("merged", merged, rm, 1e-5, 1e-6, scale_tol=False)
    close("pred", pred, rp, 1e-5, 1e-6, scale_tol=False)
    out, u8 = ops.vfi_tta_average(rp.to(DEV).contiguous(), h, w, want_uint8=True)
    ro, r8 = S.vfi_tta_average(rp, h, w, want_uint8=True)
    assert torch.equal(out.cpu(), ro) and torch.equal(u8.cpu(), r8)           # exact: one fp32 add, a division by 2, a truncation


# ------------------------------------------------------------------------------------------------ the network
def _tiny():
    from oracle.cases import TINY_VFI, tiny_vfi_inputs, vfi_weights
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    model = EMAVFI(VFIConfig(F=TINY_VFI["F"], depth=TINY_VFI["depth"]))
    sd = vfi_weights(model.spec())
    inp = tiny_vfi_inputs()
    imgs = torch.cat((inp["img0"], inp["img1"]), 1)
    return model, sd, inp, torch.cat((imgs, imgs.flip(2).flip(3)), 0), TINY_VFI


def _rel(name, got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"[{name} {str(ELEM)[6:]}] rel L2 {e:.3e} | max abs {(got - ref).abs().max():.3e} (ref absmax {ref.abs().max():.3f})")
    assert torch.isfinite(got).all() and e <= tol * TOLF, f"{name}: rel L2 {e:.3e}"


def test_vfi_network_matches_oracle_and_golden(ops):
    from oracle import vfi_oracle as O
    torch.set_grad_enabled(False)
    model, sd, inp, x, T = _tiny()
    model.load_state_dict(sd, device=DEV)
    H, W = T["H"], T["W"]
    cl = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    r = model.net_forward(cl(x[:, :3]).to(DEV), cl(x[:, 3:6]).to(DEV), 2, H, W, want=True)
    o = O.net_forward(sd, O.vfi_config(T["F"], T["depth"]), x)
    for lvl in range(5):
        t, C, h, w = r["af"][lvl]
        _rel(f"af{lvl}", t[:, :C], cl(o["af"][lvl]), 3e-2)
    for k in range(2):
        t, C, h, w = r["mf"][k]
        _rel(f"mf{k + 3}", t[:, :C], cl(o["mf"][k]), 2e-2)
    close("flow [px]", r["fm"][:, :4], cl(o["flow"]), 0.15, 0.0)
    close("mask", r["fm"][:, 4:5], cl(o["mask"]), 3e-2, 0.0)
    close("merged", r["merged"], cl(o["merged"]), 2e-2, 0.0)
    close("pred", r["pred"], cl(o["pred"]), 2e-2, 0.0)
    g = torch.load(GOLD)
    mid, u8 = model.inference(inp["img0"][0].permute(1, 2, 0).contiguous().to(DEV), inp["img1"][0].permute(1, 2, 0).contiguous().to(DEV),
                              want_uint8=True)
    gold = g["tta"][0].permute(1, 2, 0)
    close("fast-TTA middle frame vs vendored network", mid, gold, 2e-2, 0.0)
    d = (u8.cpu().int() - (gold * 255.0).to(torch.uint8).int()).abs()
    print(f"[uint8 frame {str(ELEM)[6:]}] max |diff| {d.max().item()} levels, {100.0 * (d > 1).float().mean().item():.2f}% of values differ by more than 1")
    assert d.max().item() <= (6 if ELEM == torch.bfloat16 else 2)
    # batch independence: the pair alone (B = 1) gives the first half of the TTA batch (window attention pairs frames, not batch rows)
    r1 = model.net_forward(cl(x[:1, :3]).to(DEV), cl(x[:1, 3:6]).to(DEV), 1, H, W)
    close("B=1 vs first half of the TTA batch", r1, r["pred"][: H * W], 2e-3, 0.0)


def test_vfi_process_on_device(ops):
    """vfi_process end to end: input frames pass through the reference's (lossy) uint8 round trip, interpolated frames sit between them."""
    import numpy as np
    from oracle import vfi_oracle as O
    from streamingt2v_amd.ema_vfi import vfi_process
    torch.set_grad_enabled(False)
    model, sd, inp, _, T = _tiny()
    model.load_state_dict(sd, device=DEV)
    f = lambda t: (t[0].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()[:, :, ::-1].copy()        # RGB uint8 frames
    video = [f(inp["img0"]), f(inp["img1"]), f(inp["img0"])]
    got = vfi_process(video, model, 6, out_size=(T["W"], T["H"]), device=DEV)
    cfg = O.vfi_config(T["F"], T["depth"])
    ref = O.vfi_process(video, lambda a, b: O.inference_fast_tta(sd, cfg, a, b), 6, out_size=(T["W"], T["H"]))
    assert len(got) == len(ref) == 6
    for i, (a, b) in enumerate(zip(got, ref)):
        d = np.abs(np.asarray(a).astype(int) - np.asarray(b).astype(int))
        assert d.max() <= (0 if i in (0, 2, 4, 5) else (6 if ELEM == torch.bfloat16 else 2)), (i, d.max())


@pytest.mark.parametrize("H,W,B", [(64, 96, 1), (112, 112, 2), (32, 224, 1)])
def test_vfi_network_other_sizes(ops, H, W, B):
    """Other token-grid shapes than the golden case: /8 grids 8x12, 14x14 (no padding: no attention mask without shift), 4x28;
    /16 grids 4x6, 7x7, 2x14 -- padding on one, both or no axis, batch 1 and 2 -- against the CPU oracle."""
    from oracle import vfi_oracle as O
    torch.set_grad_enabled(False)
    model, sd, _, _, T = _tiny()
    model.load_state_dict(sd, device=DEV)
    g = torch.Generator(); g.manual_seed(H * 1000 + W)
    low = torch.rand(B, 6, H // 8 + 2, W // 8 + 2, generator=g)
    x = torch.nn.functional.interpolate(low, scale_factor=8, mode="bicubic", align_corners=False).clamp(0, 1)[:, :, 5:5 + H, 3:3 + W].contiguous()
    cl = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    r = model.net_forward(cl(x[:, :3]).to(DEV), cl(x[:, 3:6]).to(DEV), B, H, W, want=True)
    o = O.net_forward(sd, O.vfi_config(T["F"], T["depth"]), x)
    for lvl in (3, 4):
        t, C, h, w = r["af"][lvl]
        _rel(f"af{lvl} {H}x{W}", t[:, :C], cl(o["af"][lvl]), 3e-2)
    close(f"flow [px] {H}x{W}", r["fm"][:, :4], cl(o["flow"]), 0.2, 0.0)
    close(f"pred {H}x{W}", r["pred"], cl(o["pred"]), 2.5e-2, 0.0)


def test_vfi_shipped_width_vs_reference_golden(ops):
    """EMA-VFI at the SHIPPED width (F = 32, 65.7 M parameters) on a 64x96 frame pair against the fast-TTA output of the UNMODIFIED vendored
    network (tests/golden/vfi_fullarch.pt, oracle/make_golden_fullarch_small.py).  16-bit emulation on CPU predicts 2.5e-3 (bf16) /
    3.2e-4 (fp16) max abs error and at most 1 uint8 level."""
    from oracle.cases import fullarch_small_inputs, vfi_weights
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    torch.set_grad_enabled(False)
    model = EMAVFI(VFIConfig())
    model.load_state_dict(vfi_weights(model.spec(), seed=12), device=DEV)
    inp = fullarch_small_inputs()
    gold = torch.load(os.path.join(os.path.dirname(GOLD), "vfi_fullarch.pt"))["tta"][0].permute(1, 2, 0)
    mid, u8 = model.inference(inp["img0"][0].permute(1, 2, 0).contiguous().to(DEV), inp["img1"][0].permute(1, 2, 0).contiguous().to(DEV), want_uint8=True)
    close("EMA-VFI F=32 fast-TTA vs vendored network", mid, gold, 2e-2, 0.0)
    d = (u8.cpu().int() - (gold * 255.0).to(torch.uint8).int()).abs()
    assert d.max().item() <= (4 if ELEM == torch.bfloat16 else 2)
