"""Plain-torch fp32 statements of the `streamingt2v_amd.ops` launchers that the StreamingSVD denoiser path uses (video_model.py,
wrappers.py, sampling.py) -- TEST INFRASTRUCTURE, never imported by the product.

`install(monkeypatch)` routes streamingt2v_amd.ops through them with fp32 "elements", so that the HOST logic of the networks (layouts,
weight packing, epilogue bookkeeping, ControlNet slicing, CAM wiring, the sequence-parallel layout changes and collectives) runs on CPU:
  * tests/test_host_svd_cpu.py  : host logic vs the CPU oracle (oracle/svd_oracle.py) on the tiny configuration;
  * tests/test_distributed_cpu.py: the sequence-parallel forward on gloo ranks == the single-process forward.
Each function follows the argument contract documented in streamingt2v_amd/ops.py / include/svdhip.h.
"""
import math

import torch
import torch.nn.functional as F


def _geglu_unpack(y):
    """columns interleaved (value | gate) in blocks of 32 (video_model.pack_geglu) -> value * gelu_erf(gate)."""
    M, N = y.shape
    y = y.view(M, N // 64, 2, 32)
    return (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(M, N // 2)


def gemm(a, w, *, bias=None, rowvec=None, rows_per_vec=0, residual=None, blend=None, geglu=False, silu=False, out=None, out_f32=False,
         conv=None, temporal=None, trans_out=None, n_out=None, tile_cfg=0, k=None):
    a32, w32 = a.float(), w.float()
    N = w.shape[0]
    K = k if k is not None else w.shape[1]
    w32 = w32[:, :K]
    if conv is not None:
        n, cin, hin, win = conv["frames"], conv["cin"], conv["hin"], conv["win"]
        assert a.shape[0] == n * hin * win and K == 9 * cin
        x = a32[:, :cin].reshape(n, hin, win, cin).permute(0, 3, 1, 2)
        if conv.get("ups", 0):
            x = F.interpolate(x, size=(conv["hout"], conv["wout"]), mode="nearest")
        wk = w32.view(N, 3, 3, cin).permute(0, 3, 1, 2)
        stride = conv.get("stride", 1)
        if conv.get("pad_mode", 0):
            y = F.conv2d(F.pad(x, (0, 1, 0, 1)), wk, None, stride, 0)
        else:
            y = F.conv2d(x, wk, None, stride, 1)
        assert tuple(y.shape[2:]) == (conv["hout"], conv["wout"]), (y.shape, conv)
        y = y.permute(0, 2, 3, 1).reshape(-1, N)
    elif temporal is not None:
        cin, T, pix = temporal["cin"], temporal["T"], temporal["pix"]
        M = a.shape[0]
        B = M // (T * pix)
        x = a32[:, :cin].reshape(B, T, pix, cin).permute(0, 3, 1, 2)[..., None]          # b c t p 1
        wk = w32.view(N, 3, cin).permute(0, 2, 1)[..., None, None]
        y = F.conv3d(x, wk, None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(M, N)
    else:
        y = a32[:, :K] @ w32.t()
    M = y.shape[0]
    if bias is not None:
        y = y + bias[:N]
    if trans_out is not None:
        o = trans_out["out"]
        tpf = trans_out["tok_per_frame"]
        o[:, :, :tpf] = y.view(M // tpf, tpf, N).transpose(1, 2).to(o.dtype)
        return o
    if geglu:
        y = _geglu_unpack(y)
    if rowvec is not None:
        idx = torch.arange(M, device=y.device) // rows_per_vec
        y = y + rowvec[idx][:, : y.shape[1]]
    if residual is not None:
        y = y + residual.float()
    if blend is not None:
        alpha, S = blend
        y = alpha * S.float() + (1.0 - alpha) * y
    if silu:
        y = F.silu(y)
    if n_out is not None:
        y = y[:, :n_out]
    if out is not None:
        out.copy_(y.to(out.dtype))
        return out
    return y if out_f32 else y.to(a.dtype)


def attn_spatial(q, k, vt, out, frames, n_tok, heads):
    C = heads * 64
    qh = q.float()[:, :C].reshape(frames, n_tok, heads, 64).transpose(1, 2)
    kh = k.float()[:, :C].reshape(frames, n_tok, heads, 64).transpose(1, 2)
    vh = vt.float()[:, :, :n_tok].reshape(frames, heads, 64, n_tok).transpose(2, 3)
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(frames * n_tok, C)
    out.copy_(o.to(out.dtype))
    return out


def attn_temporal(q, k, v, out, batch, tq, tk, n_pix, heads):
    C = heads * 64

    def bpthd(t, T):      # rows (b t p), cols (h d) -> (b p) h t d
        return t.float()[:, :C].reshape(batch, T, n_pix, heads, 64).permute(0, 2, 3, 1, 4).reshape(batch * n_pix, heads, T, 64)
    o = F.scaled_dot_product_attention(bpthd(q, tq), bpthd(k, tk), bpthd(v, tk))
    o = o.view(batch, n_pix, heads, tq, 64).permute(0, 3, 1, 2, 4).reshape(batch * tq * n_pix, C)
    out.copy_(o.to(out.dtype))
    return out


def attn_cross(q, k, vt, out, frames, n_q, n_k, frames_per_kv, heads):
    C = heads * 64
    nkv = frames // frames_per_kv
    qh = q.float()[:, :C].reshape(frames, n_q, heads, 64).transpose(1, 2)
    kh = k.float()[:, :C].reshape(nkv, n_k, heads, 64).transpose(1, 2).repeat_interleave(frames_per_kv, 0)
    vh = vt.float()[:, :, :n_k].reshape(nkv, heads, 64, n_k).transpose(2, 3).repeat_interleave(frames_per_kv, 0)
    o = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(frames * n_q, C)
    out.copy_(o.to(out.dtype))
    return out


def groupnorm_sums(x, frames, pix, frames_per_stat, groups=32):
    """[nstat, groups, 2] float64 (sum, sum of squares) over frames_per_stat frames x pix x C/groups."""
    C = x.shape[1]
    v = x.double().reshape(frames // frames_per_stat, frames_per_stat * pix, groups, C // groups)
    return torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1).contiguous()


def groupnorm_apply_sums(x, frames, pix, gamma, beta, eps, sums, count, *, frames_per_stat=1, silu=False, groups=32, out=None):
    C = x.shape[1]
    mean = sums[..., 0] / count
    var = (sums[..., 1] / count - mean * mean).clamp_min(0.0)
    rstd = 1.0 / torch.sqrt(var + eps)
    v = x.float().reshape(frames // frames_per_stat, frames_per_stat * pix, groups, C // groups)
    y = (v - mean.float()[:, None, :, None]) * rstd.float()[:, None, :, None]
    y = y.reshape(frames * pix, C) * gamma + beta
    if silu:
        y = F.silu(y)
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def groupnorm(x, frames, pix, gamma, beta, eps, *, frames_per_stat=1, silu=False, groups=32, out=None):
    sums = groupnorm_sums(x, frames, pix, frames_per_stat, groups)
    count = float(frames_per_stat) * pix * (x.shape[1] // groups)
    return groupnorm_apply_sums(x, frames, pix, gamma, beta, eps, sums, count, frames_per_stat=frames_per_stat, silu=silu, groups=groups, out=out)


def layernorm(x, gamma, beta, *, eps=1e-5, addvec=None, rows_per_vec=0, want_sum=False, silu=False, out=None):
    v = x.float()
    if addvec is not None:
        v = v + addvec[torch.arange(x.shape[0], device=x.device) // rows_per_vec]
    y = F.layer_norm(v, (x.shape[1],), gamma, beta, eps)
    if silu:
        y = F.silu(y)
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        y = out
    return (y, v.to(x.dtype)) if want_sum else y


def nchw_to_tokens(x0, x1, scale, cpad):
    from streamingt2v_amd import ops
    F_, c0 = x0.shape[0], x0.shape[1]
    pix = x0.shape[2] * x0.shape[3]
    out = torch.zeros((F_, pix, cpad), dtype=torch.float32, device=x0.device)
    a = x0.float() * (scale[:, None, None, None] if scale is not None else 1.0)
    out[..., :c0] = a.flatten(2).transpose(1, 2)
    if x1 is not None:
        out[..., c0:c0 + x1.shape[1]] = x1.float().flatten(2).transpose(1, 2)
    return out.reshape(F_ * pix, cpad).to(ops.ELEM)


def _split3(v):
    """fp32 rows -> [hi | lo | hi] in the element type (csrc/precision.hip); with fp32 'elements' lo is exactly zero."""
    from streamingt2v_amd import ops
    hi = v.to(ops.ELEM)
    lo = (v - hi.float()).to(ops.ELEM)
    return torch.cat([hi, lo, hi], 1)


def nchw_to_tokens_x3(x0, x1, scale, cpad):
    return _split3(_tokens_f32(x0, x1, scale, cpad))


def _tokens_f32(x0, x1, scale, cpad):
    F_, c0 = x0.shape[0], x0.shape[1]
    pix = x0.shape[2] * x0.shape[3]
    out = torch.zeros((F_, pix, cpad), dtype=torch.float32, device=x0.device)
    a = x0.float() * (scale[:, None, None, None] if scale is not None else 1.0)
    out[..., :c0] = a.flatten(2).transpose(1, 2)
    if x1 is not None:
        out[..., c0:c0 + x1.shape[1]] = x1.float().flatten(2).transpose(1, 2)
    return out.reshape(F_ * pix, cpad)


def rows_split3(x, ln=None, eps=1e-5, silu=False):
    v = x.float()
    if ln is not None:
        v = F.layer_norm(v, (x.shape[1],), ln[0], ln[1], eps)
    if silu:
        v = F.silu(v)
    return _split3(v)


def add_rows_f32b(x, b):
    return (x.float() + b.float()).to(x.dtype)


def head_gn_silu_conv3x3(x, frames, H, W, gamma, beta, eps, wt, bias, cout, groups=32):
    C = x.shape[1]
    v = x.float().reshape(frames, H * W, C).transpose(1, 2).reshape(frames, C, H, W)
    v = F.silu(F.group_norm(v, groups, gamma, beta, eps))
    w = wt.reshape(3, 3, C, 4).permute(3, 2, 0, 1)                        # [tap(ky, kx)][c][co] -> [co, c, ky, kx]
    b4 = torch.zeros(4, dtype=torch.float32, device=x.device)
    b4[:cout] = bias[:cout]
    y = F.conv2d(v, w, b4, padding=1)
    return y.permute(0, 2, 3, 1).reshape(frames * H * W, 4).contiguous()


def tokens_to_nchw(x, c, frames, h, w):
    return x.float()[:, :c].reshape(frames, h * w, c).transpose(1, 2).reshape(frames, c, h, w).contiguous()


def concat_channels(a, b):
    return torch.cat([a, b], 1)


def add_rows(x, b):
    return (x.float() + b.float()).to(x.dtype)


def softmax_rows(s, out, scale):
    out.copy_(torch.softmax(s.float() * scale, dim=-1).to(out.dtype))
    return out


def ae_time_mix3(x, w, b, frames, h, wd, clamp):
    """AE3DConv.time_mix_conv (temporal_ae.py:99-105): 3-tap conv over frames on the 3 output channels, zero padded; x [frames*h*wd, >=3]
    tokens fp32, w [3, 3, 3] (co, ci, kt) -> NCHW fp32."""
    v = x.float()[:, :3].reshape(1, frames, h * wd, 3).permute(0, 3, 1, 2)[..., None]            # b c t p 1
    y = F.conv3d(v, w.float()[..., None, None], b.float(), padding=(1, 0, 0))[0, :, :, :, 0]      # c t p
    y = y.permute(1, 0, 2).reshape(frames, 3, h, wd)
    return y.clamp(-1.0, 1.0) if clamp else y.contiguous()


def to_elem_rows(x, out=None):
    from streamingt2v_amd import ops
    y = x.to(ops.ELEM)
    if out is not None:
        out.copy_(y)
        return out
    return y


def permute_rows(x, dims, perm, out=None):
    C = x.shape[1]
    y = x.reshape(*dims, C).permute(*perm, 4).reshape(-1, C).contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def to_elem(x, silu=False):
    from streamingt2v_amd import ops
    return (F.silu(x) if silu else x).to(ops.ELEM)


def timestep_embedding(t, dim, max_period=10000.0, f32=False):
    from streamingt2v_amd import ops
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], -1).to(torch.float32 if f32 else ops.ELEM)


def edm_euler_step(x, net, guidance_scale, sigma, sigma_next):
    T, C = x.shape[0], x.shape[1]
    h, w = x.shape[2], x.shape[3]
    n = net[:, :C].reshape(2, T, h * w, C).permute(0, 1, 3, 2).reshape(2, T, C, h, w)
    c_skip, c_out = 1.0 / (sigma * sigma + 1.0), -sigma / math.sqrt(sigma * sigma + 1.0)
    du, dc = n[0] * c_out + x * c_skip, n[1] * c_out + x * c_skip
    den = du + guidance_scale[:, None, None, None] * (dc - du)
    x.copy_(x + (x - den) / sigma * (sigma_next - sigma))
    return x


def _ff_unpack(img, hidden, C, esize, dtype):
    """Inverse of video_model.pack_ff_fused, written from the layout documented in csrc/ff_fused.hip / include/svdhip.h: per chunk of 32 hidden
    units [2 C/16 W1 fragments | 1 KiB bias slot (64 floats) | 2 C/32 W2 fragments]; a fragment = 64 lanes x 8 elements."""
    nch, NS, NO = hidden // 32, C // 16, C // 32
    fb = 64 * 8 * esize
    blob = 2 * NS * fb + 1024 + 2 * NO * fb
    img = img.cpu().view(nch, blob)
    f1 = img[:, :2 * NS * fb].contiguous().view(dtype).view(nch, NS, 2, 64, 8).float()
    bias = img[:, 2 * NS * fb:2 * NS * fb + 256].contiguous().view(torch.float32).view(nch, 2, 32)
    f2 = img[:, 2 * NS * fb + 1024:].contiguous().view(dtype).view(nch, 2, NO, 64, 8).float()
    w1 = torch.zeros(2 * hidden, C); b1 = torch.zeros(2 * hidden); w2 = torch.zeros(C, hidden)
    # scatter by index tensors (one assignment per operand; the element-by-element form of this inverse took ~10 s per layer and made the CPU suite 5x slower):
    # lane l = (m = l % 32, kg = l / 32); W1 fragment (k-step s, tile t): lane holds row m of the tile (16 value rows, then the 16 gate rows of hidden
    # 32 c + 16 t ..), channels 16 s + 8 kg .. + 7;  W2 fragment (tile t, output tile o): lane holds output channel 32 o + m, hidden units 4 kg + e (e < 4) /
    # 8 + 4 kg + (e - 4) of the tile
    ar = torch.arange
    c5, l5, e5 = ar(nch).view(-1, 1, 1, 1, 1), ar(64).view(1, 1, 1, -1, 1), ar(8).view(1, 1, 1, 1, -1)
    m5, kg5 = l5 & 31, l5 >> 5
    s5, t5 = ar(NS).view(1, -1, 1, 1, 1), ar(2).view(1, 1, -1, 1, 1)                            # f1 index order [c, s, t, l, e]
    hid = 32 * c5 + 16 * t5 + (m5 & 15)
    row = torch.where(m5 < 16, hid, hidden + hid) + 0 * (s5 + e5)
    col = 16 * s5 + 8 * kg5 + e5 + 0 * (c5 + t5)
    w1[row.expand_as(f1), col.expand_as(f1)] = f1
    t5b, o5 = ar(2).view(1, -1, 1, 1, 1), ar(NO).view(1, 1, -1, 1, 1)                            # f2 index order [c, t, o, l, e]
    u5 = torch.where(e5 < 4, 4 * kg5 + e5, 8 + 4 * kg5 + (e5 - 4))
    w2[(32 * o5 + m5 + 0 * (c5 + t5b + e5)).expand_as(f2), (32 * c5 + 16 * t5b + u5 + 0 * o5).expand_as(f2)] = f2
    c3, t3, m3 = ar(nch).view(-1, 1, 1), ar(2).view(1, -1, 1), ar(32).view(1, 1, -1)
    hb = 32 * c3 + 16 * t3 + (m3 & 15)
    b1[torch.where(m3 < 16, hb, hidden + hb)] = bias
    return w1, b1, w2


def ff_geglu_fused(x, img, hidden, b2, *, residual=None, blend=None, out_f32=False, out=None, rowvec=None, rows_per_vec=0, ln=None, ln_eps=1e-5, ln_addvec=None,
                   ln_rows_per_vec=0):
    if not hasattr(img, "_ff_unpacked"):          # cached ON the image tensor object (an address can be recycled by another layer's image)
        img._ff_unpacked = _ff_unpack(img, hidden, x.shape[1], x.element_size(), x.dtype)
    w1, b1, w2 = img._ff_unpacked
    y = x.float() @ w1.t() + b1
    y = (y[:, :hidden] * F.gelu(y[:, hidden:])) @ w2.t() + b2
    if rowvec is not None:
        y = y + rowvec[:, : y.shape[1]].repeat_interleave(rows_per_vec, dim=0)[: y.shape[0]]
    if residual is not None:
        y = y + residual.float()
    if blend is not None:
        alpha, S = blend
        y = alpha * S.float() + (1.0 - alpha) * y
    if ln is not None:
        assert out is None and out_f32 and blend is None and residual is not None and residual.dtype == torch.float32
        return y, layernorm(y, ln[0], ln[1], eps=ln_eps, addvec=ln_addvec, rows_per_vec=ln_rows_per_vec).to(x.dtype)
    if out is not None:
        out.copy_(y.to(out.dtype))
        return out
    return y if out_f32 else y.to(x.dtype)


def _rowgemm_unpack(img, dtype):
    """inverse of video_model.pack_rowgemm320: image [20 k-steps, 10 tiles, 64 lanes, 8] -> W [320 out, 320 in] fp32"""
    f = img.view(dtype).view(20, 10, 64, 8).float()
    ar = torch.arange
    s_, o, l, e = ar(20).view(-1, 1, 1, 1), ar(10).view(1, -1, 1, 1), ar(64).view(1, 1, -1, 1), ar(8).view(1, 1, 1, -1)
    w = torch.zeros(320, 320)
    w[(32 * o + (l & 31) + 0 * (s_ + e)).expand_as(f), (16 * s_ + 8 * (l >> 5) + e + 0 * o).expand_as(f)] = f
    return w


def _rowproj_unpack(img, dtype, n_out):
    """inverse of video_model.pack_rowproj320: image [N / 64 chunks, 20 k-steps, 2 tiles, 64 lanes, 8] -> W [N, 320] fp32"""
    f = img.view(dtype).view(n_out // 64, 20, 2, 64, 8).float()
    ar = torch.arange
    ch, s_, t, l, e = (ar(n_out // 64).view(-1, 1, 1, 1, 1), ar(20).view(1, -1, 1, 1, 1), ar(2).view(1, 1, -1, 1, 1), ar(64).view(1, 1, 1, -1, 1),
                       ar(8).view(1, 1, 1, 1, -1))
    w = torch.zeros(n_out, 320)
    w[(64 * ch + 2 * (l & 31) + t + 0 * (s_ + e)).expand_as(f), (16 * s_ + 8 * (l >> 5) + e + 0 * (ch + t)).expand_as(f)] = f
    return w


def rowproj_ok(x, w_img):
    return w_img is not None and x.shape[1] == 320


def rowproj320(x, w_img, n_out, *, bias=None, out=None):
    """csrc/rowproj.hip: y = x W^T + bias (include/svdhip.h svd_rowproj320)."""
    from streamingt2v_amd import ops
    if not hasattr(w_img, "_rp_unpacked"):
        w_img._rp_unpacked = _rowproj_unpack(w_img.cpu(), ops.ELEM, n_out).to(x.device)
    y = x.float() @ w_img._rp_unpacked.t()
    if bias is not None:
        y = y + bias
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def rowgemm320(x, w_img, *, bias=None, rowvec=None, rows_per_vec=0, residual=None, out_f32=True, ln=None, eps=1e-5, want_y=True, out=None):
    """csrc/rowgemm.hip: y = residual + bias + rowvec[row // rows_per_vec] + x W^T; yn = LayerNorm(y) (include/svdhip.h svd_rowgemm320)."""
    if not hasattr(w_img, "_rg_unpacked"):
        w_img._rg_unpacked = _rowgemm_unpack(w_img, x.dtype)
    y = x.float() @ w_img._rg_unpacked.t()
    if bias is not None:
        y = y + bias
    if rowvec is not None:
        y = y + rowvec[:, :320].repeat_interleave(rows_per_vec, dim=0)[: y.shape[0]]
    if residual is not None:
        y = y + residual.float()
    yn = F.layer_norm(y, (320,), ln[0], ln[1], eps).to(x.dtype) if ln is not None else None
    if out is not None:
        out.copy_(y.to(out.dtype))
        return out, yn
    return ((y if out_f32 else y.to(x.dtype)) if want_y else None), yn


def rowgemm_ok(x, w_img, rows_per_vec=0):
    return w_img is not None and x.shape[1] == 320 and (rows_per_vec == 0 or rows_per_vec % 32 == 0)


def adaptive_avgpool(x, frames, hin, win, hout, wout):
    Cc = x.shape[1]
    y = F.adaptive_avg_pool2d(x.float().view(frames, hin, win, Cc).permute(0, 3, 1, 2), (hout, wout))
    return y.permute(0, 2, 3, 1).reshape(-1, Cc).to(x.dtype)


def i2v_image_temporal_encoder(x, params, batch, frames, h, w):
    """I2VGenXLTransformerTemporalEncoder on 4-channel rows (b, f, p) (unet_i2vgen_xl.py:110-160): x += to_out(attn(LN(x))) with 2 heads of
    dim 4 over the frames of a pixel; x += W2 gelu(W1 x + b1) + b2 -> fp32 [(b f), 4, h, w].  Parameter layout: include/svdhip.h."""
    p = params.float()
    ln_w, ln_b, wq, wk, wv = p[0:4], p[4:8], p[8:40].view(8, 4), p[40:72].view(8, 4), p[72:104].view(8, 4)
    wo, bo, w1, b1, w2, b2 = p[104:136].view(4, 8), p[136:140], p[140:204].view(16, 4), p[204:220], p[220:284].view(4, 16), p[284:288]
    pix = h * w
    X = x[:, :4].float().view(batch, frames, pix, 4).permute(0, 2, 1, 3)                    # b p f 4
    n = F.layer_norm(X, (4,), ln_w, ln_b, 1e-5)
    hd = lambda t: t.view(batch, pix, frames, 2, 4).transpose(2, 3)                         # b p head f 4
    a = F.scaled_dot_product_attention(hd(n @ wq.t()), hd(n @ wk.t()), hd(n @ wv.t()))
    xo = X + a.transpose(2, 3).reshape(batch, pix, frames, 8) @ wo.t() + bo
    xo = xo + F.gelu(xo @ w1.t() + b1) @ w2.t() + b2
    return xo.permute(0, 2, 3, 1).reshape(batch * frames, 4, h, w).contiguous()


def ddim_cfg_step(x, pred_uncond, pred_cond, guidance_scale, alpha_t, alpha_prev, v_prediction=True, out=None):
    """csrc/i2v.hip ddim_cfg_step_kernel: guidance, then diffusers DDIMScheduler.step with eta 0 (pipeline_i2vgen_xl.py:872-885)."""
    v = pred_uncond if pred_cond is None else pred_uncond + guidance_scale * (pred_cond - pred_uncond)
    sa, sb, spa, spb = alpha_t ** 0.5, (1.0 - alpha_t) ** 0.5, alpha_prev ** 0.5, (1.0 - alpha_prev) ** 0.5
    x0, eps = (sa * x - sb * v, sa * v + sb * x) if v_prediction else ((x - sb * v) / sa, v)
    res = spa * x0 + spb * eps
    if out is not None:
        out.copy_(res)
        return out
    return res


NAMES = ("gemm", "attn_spatial", "attn_temporal", "attn_cross", "groupnorm", "groupnorm_sums", "groupnorm_apply_sums", "layernorm", "nchw_to_tokens",
         "tokens_to_nchw", "concat_channels", "add_rows", "to_elem", "to_elem_rows", "permute_rows", "timestep_embedding", "edm_euler_step", "softmax_rows", "ae_time_mix3",
         "nchw_to_tokens_x3", "rows_split3", "add_rows_f32b", "head_gn_silu_conv3x3", "adaptive_avgpool", "i2v_image_temporal_encoder", "ff_geglu_fused", "ddim_cfg_step", "rowgemm320", "rowgemm_ok", "rowproj320", "rowproj_ok")


def install(monkeypatch=None):
    """Route streamingt2v_amd.ops through the statements above, with fp32 'elements' (CPU host-logic tests only)."""
    import sys
    from streamingt2v_amd import ops
    me = sys.modules[__name__]
    for n in NAMES + ("to_bf16",):
        fn = getattr(me, "to_elem" if n == "to_bf16" else n)
        if monkeypatch is not None:
            monkeypatch.setattr(ops, n, fn, raising=False)
        else:
            setattr(ops, n, fn)
    if monkeypatch is not None:
        monkeypatch.setattr(ops, "ROWPROJ_MIN_ROWS", 0, raising=False)
        monkeypatch.setattr(ops, "ROWGEMM_PLAIN_MIN_ROWS", 0, raising=False)          # host-logic tests: every rowgemm320 call site, whatever the row count
        monkeypatch.setattr(ops, "ELEM", torch.float32)
    else:
        ops.ROWGEMM_PLAIN_MIN_ROWS = 0
        ops.ROWPROJ_MIN_ROWS = 0
        ops.ELEM = torch.float32
