"""Plain-torch fp32 statements of the `streamingt2v_amd.ops` launchers that `ema_vfi.py` uses -- TEST INFRASTRUCTURE.

Two uses: (1) CPU tests install them in place of the HIP launchers (`install(monkeypatch)`) to check the HOST logic of EMAVFI (layout,
padding, window partition, weight packing, channel bookkeeping) against oracle/vfi_oracle.py without a GPU; (2) the GPU tests use the same
functions as the per-kernel fp32 reference of csrc/vfi.hip.  Never imported by the product.
"""
import torch
import torch.nn.functional as F


def gemm(a, w, *, bias=None, residual=None, out_f32=False, conv=None, out=None, **kw):
    assert not kw, f"shim: unsupported gemm options {list(kw)}"
    a32, w32 = a.float(), w.float()
    N = w.shape[0]
    if conv is not None:
        n, cin, hin, win = conv["frames"], conv["cin"], conv["hin"], conv["win"]
        assert a.shape == (n * hin * win, cin) and w.shape[1] == 9 * cin and conv.get("ups", 0) == 0 and conv.get("pad_mode", 0) == 0
        x = a32.view(n, hin, win, cin).permute(0, 3, 1, 2)
        y = F.conv2d(x, w32.view(N, 3, 3, cin).permute(0, 3, 1, 2), None, conv.get("stride", 1), 1)
        assert tuple(y.shape[2:]) == (conv["hout"], conv["wout"])
        y = y.permute(0, 2, 3, 1).reshape(-1, N)
    else:
        assert a.shape[1] == w.shape[1] and a.shape[1] % 32 == 0
        y = a32 @ w32.t()
    if bias is not None:
        y = y + bias[:N]
    if residual is not None:
        y = y + residual.float()
    y = y if out_f32 else y.to(a.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def layernorm(x, gamma, beta, *, eps=1e-5, **kw):
    assert not kw
    return F.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps).to(x.dtype)


def to_elem(x, silu=False):
    from streamingt2v_amd import ops
    assert x.dtype == torch.float32 and x.is_contiguous() and not silu
    return x.to(ops.ELEM)


def prelu_(x, slope):
    x.copy_(torch.where(x.float() >= 0, x.float(), x.float() * slope[: x.shape[1]]).to(x.dtype))
    return x


def dwconv3x3_gelu(x, w9, bias, frames, h, w):
    C = x.shape[1]
    y = F.conv2d(x.float().view(frames, h, w, C).permute(0, 3, 1, 2), w9.t().reshape(C, 1, 3, 3), bias, 1, 1, 1, C)
    return F.gelu(y).permute(0, 2, 3, 1).reshape(-1, C).to(x.dtype)


def window_attn_7x7(q, kv, ce, mask, n_win, heads, motion_per_head, scale):
    C, md = heads * 32, heads * motion_per_head
    qh = q.float()[:, :C].view(n_win, 49, heads, 32).permute(0, 2, 1, 3)
    kvf = kv.float()[:, : 2 * C].view(n_win, 49, 2, heads, 32)
    kvf = torch.cat([kvf[n_win // 2:], kvf[: n_win // 2]])                           # window w attends to window (w + n_win / 2) % n_win
    k, v = kvf[:, :, 0].permute(0, 2, 1, 3), kvf[:, :, 1].permute(0, 2, 1, 3)
    ceh = ce[:, :md].view(n_win, 49, heads, motion_per_head).permute(0, 2, 1, 3)
    att = (qh @ k.transpose(-2, -1)) * scale
    if mask is not None:
        att = att + mask.repeat(n_win // mask.shape[0], 1, 1)[:, None]
    att = att.softmax(-1)
    ox = (att @ v).transpose(1, 2).reshape(n_win * 49, C)
    oc = (att @ ceh).transpose(1, 2).reshape(n_win * 49, md) - ce[:, :md]
    return ox.to(q.dtype), oc.to(q.dtype)


def warp_bilinear(x, flow, frames, h, w):
    """grid_sample(bilinear, border, align_corners=True) at (x + fx, y + fy), built the way warplayer.py builds its grid."""
    C = x.shape[1]
    img = x.float().view(frames, h, w, C).permute(0, 3, 1, 2)
    fl = flow.reshape(frames, h, w, 2)
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, w).expand(frames, h, w) + fl[..., 0] / ((w - 1.0) / 2.0)
    gy = torch.linspace(-1.0, 1.0, h).view(1, h, 1).expand(frames, h, w) + fl[..., 1] / ((h - 1.0) / 2.0)
    y = F.grid_sample(img, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border", align_corners=True)
    return y.permute(0, 2, 3, 1).reshape(-1, C).to(x.dtype)


def resize_bilinear(x, frames, hin, win, scale_factor, *, mult=None, out=None, accumulate=False):
    Cc = x.shape[1]
    y = F.interpolate(x.view(frames, hin, win, Cc).permute(0, 3, 1, 2), scale_factor=scale_factor, mode="bilinear", align_corners=False)
    hout, wout = y.shape[2:]
    assert (hout, wout) == (int(hin * scale_factor), int(win * scale_factor))
    y = y.permute(0, 2, 3, 1).reshape(-1, Cc)
    if mult is not None:
        y = y * mult[:Cc]
    if out is None:
        return y.contiguous(), hout, wout
    out[:, :Cc] = (out[:, :Cc] + y) if accumulate else y
    return out, hout, wout


def vfi_merge(warped0, warped1, mask, unet_out, want_merged=False):
    sg = torch.sigmoid(mask[:, :1])
    merged = warped0 * sg + warped1 * (1 - sg)
    pred = torch.clamp(merged + torch.sigmoid(unet_out[:, :3]) * 2 - 1, 0, 1)
    return (pred, merged) if want_merged else pred


def vfi_tta_average(pred2, h, w, want_uint8=False):
    p = pred2.view(2, h, w, 3)
    out = (p[0] + p[1].flip(0).flip(1)) / 2.0
    u8 = (out * 255.0).to(torch.uint8) if want_uint8 else None
    return out.reshape(h * w, 3), u8


NAMES = ("gemm", "layernorm", "to_elem", "prelu_", "dwconv3x3_gelu", "window_attn_7x7", "warp_bilinear", "resize_bilinear", "vfi_merge", "vfi_tta_average")


def install(monkeypatch):
    """Route streamingt2v_amd.ops through the statements above, with fp32 'elements' (CPU host-logic tests only)."""
    import sys
    from streamingt2v_amd import ops
    me = sys.modules[__name__]
    for n in NAMES:
        monkeypatch.setattr(ops, n, getattr(me, n))
    monkeypatch.setattr(ops, "ELEM", torch.float32)
