"""CPU restatement of the reference's vendored EMA-VFI inference path -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Plain functional torch (fp32) over a state_dict with the vendored module's key names.  Pinned: oracle/make_golden_vfi.py runs the
UNMODIFIED vendored network (code/i2v_enhance/thirdparty/VFI) on CPU with the same by-name weights and asserts agreement to <= 1e-4
(appearance/motion features, flows, mask, prediction), then stores the reference outputs in tests/golden/vfi_tiny.pt.

Follows:  model/feature_extractor.py  MotionFormer.forward :464-497, MotionFormerBlock.forward :222-290, InterFrameAttention.forward
          :141-171, Mlp.forward :104-112, CrossScalePatchEmbed :366-410, OverlapPatchEmbed :325-363, pad_if_needed/depad :29-71
          model/flow_estimation.py    Head.forward :27-43, MultiScaleFlow.forward :107-140, warp_features :59-67
          model/refine.py             Unet.forward :61-71
          model/warplayer.py          warp :7-22
          Trainer.py                  Model.inference :84-101 (fast_TTA)
          i2v_enhance_interface.py    vfi_process :30-61
"""
import math

import torch
import torch.nn.functional as F


def vfi_config(Fc=32, depth=(2, 2, 2, 4, 4), W=7):
    """config.py:9-31."""
    return dict(embed_dims=[Fc, 2 * Fc, 4 * Fc, 8 * Fc, 16 * Fc], motion_dims=[0, 0, 0, 8 * Fc // depth[-2], 16 * Fc // depth[-1]],
                num_heads=[8 * Fc // 32, 16 * Fc // 32], depths=list(depth), window=W, scales=[4, 8, 16], hidden_dims=[4 * Fc, 4 * Fc], c=Fc)


def warp(x, flow):
    """warplayer.py:7-22."""
    B, _, H, W = flow.shape
    gx = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).expand(B, -1, H, -1)
    gy = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).expand(B, -1, -1, W)
    fl = torch.cat([flow[:, 0:1] / ((x.shape[3] - 1.0) / 2.0), flow[:, 1:2] / ((x.shape[2] - 1.0) / 2.0)], 1)
    g = (torch.cat([gx, gy], 1) + fl).permute(0, 2, 3, 1)
    return F.grid_sample(x, g, mode="bilinear", padding_mode="border", align_corners=True)


def _conv_prelu(sd, p, x, stride=1):
    return F.prelu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride, 1), sd[p + ".1.weight"])


def _win_part(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _win_rev(win, ws, H, W):
    B = win.shape[0] // ((H // ws) * (W // ws))
    return win.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def _region_mask(Hp, Wp, hs, wsl, ws):
    m = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for h in hs:
        for w in wsl:
            m[:, h, w, :] = cnt
            cnt += 1
    mw = _win_part(m, ws).squeeze(-1)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return d.masked_fill(d != 0, -100.0).masked_fill(d == 0, 0.0)


def window_masks(h, w, ws, shift):
    """(pad_h, pad_w, mask [nW, N, N] or None) of one MotionFormerBlock for an h x w token grid (feature_extractor.py:29-58, 231-262)."""
    ph, pw = math.ceil(h / ws) * ws - h, math.ceil(w / ws) * ws - w
    mask = None
    if ph > 0 or pw > 0:
        mask = _region_mask(h + ph, w + pw, (slice(0, ph // 2), slice(ph // 2, h + ph // 2), slice(h + ph // 2, None)),
                            (slice(0, pw // 2), slice(pw // 2, w + pw // 2), slice(w + pw // 2, None)), ws)
    if shift:
        sm = _region_mask(h + ph, w + pw, (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)),
                          (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)), ws)
        if mask is not None:
            sm = sm.masked_fill(mask != 0, -100.0)
        mask = sm
    return ph, pw, mask


def _block(sd, p, x, cor, H, W, B, heads, ws, shift):
    """MotionFormerBlock.forward: x [2B, H*W, C], cor [2B, H, W, 2] -> (x, motion [2B, H*W, md])."""
    C = x.shape[-1]
    ph, pw, mask = window_masks(H, W, ws, shift)
    pad = (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)
    xp, cp = F.pad(x.view(2 * B, H, W, C), pad), F.pad(cor, pad)
    if shift:
        xp, cp = torch.roll(xp, (-shift, -shift), (1, 2)), torch.roll(cp, (-shift, -shift), (1, 2))
    Hp, Wp = xp.shape[1:3]
    xw, cw = _win_part(xp, ws), _win_part(cp, ws)
    n = xw.shape[0]
    xn = F.layer_norm(xw, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    xr = torch.cat([xn[n // 2:], xn[:n // 2]])
    a = p + "attn."
    N, hd = ws * ws, C // heads
    q = F.linear(xn, sd[a + "q.weight"], sd[a + "q.bias"]).view(n, N, heads, hd).permute(0, 2, 1, 3)
    kv = F.linear(xr, sd[a + "kv.weight"], sd[a + "kv.bias"]).view(n, N, 2, heads, hd).permute(2, 0, 3, 1, 4)
    ce_ = F.linear(cw, sd[a + "cor_embed.weight"], sd[a + "cor_embed.bias"])
    md = ce_.shape[-1]
    ce = ce_.view(n, N, heads, md // heads).permute(0, 2, 1, 3)
    att = (q @ kv[0].transpose(-2, -1)) * hd ** -0.5
    if mask is not None:
        nW = mask.shape[0]
        att = (att.view(n // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    att = att.softmax(-1)
    xa = (att @ kv[1]).transpose(1, 2).reshape(n, N, C)
    crev = (att @ ce).transpose(1, 2).reshape(n, N, md)
    motion = F.linear(crev - ce_, sd[a + "motion_proj.weight"], sd[a + "motion_proj.bias"])
    xa = F.linear(xa, sd[a + "proj.weight"], sd[a + "proj.bias"])
    xb, mo = _win_rev(xn + xa, ws, Hp, Wp), _win_rev(motion, ws, Hp, Wp)        # note: the residual is on the NORMED tokens (:271)
    if shift:
        xb, mo = torch.roll(xb, (shift, shift), (1, 2)), torch.roll(mo, (shift, shift), (1, 2))
    crop = lambda t: t[:, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W].reshape(2 * B, H * W, -1)
    x, mo = crop(xb), crop(mo)
    m = p + "mlp."
    y = F.linear(F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6), sd[m + "fc1.weight"], sd[m + "fc1.bias"])
    Ch = y.shape[-1]
    y = F.conv2d(y.transpose(1, 2).reshape(2 * B, Ch, H, W), sd[m + "dwconv.dwconv.weight"], sd[m + "dwconv.dwconv.bias"], 1, 1, 1, Ch)
    y = F.linear(F.gelu(y.reshape(2 * B, Ch, -1).transpose(1, 2)), sd[m + "fc2.weight"], sd[m + "fc2.bias"])
    return x + y, mo


def feature_bone(sd, cfg, x1, x2, p="feature_bone."):
    """MotionFormer.forward -> (appearance features [5], motion features [5] (empty lists for the conv stages))."""
    B = x1.shape[0]
    x = torch.cat([x1, x2], 0)
    ws, af, mf, xs = cfg["window"], [], [], []
    for i in range(5):
        if i < 3:
            if i > 0:
                x = _conv_prelu(sd, f"{p}patch_embed{i + 1}", x, 2)
            for j in range(cfg["depths"][i]):
                x = F.prelu(F.conv2d(x, sd[f"{p}block{i + 1}.conv.{2 * j}.weight"], sd[f"{p}block{i + 1}.conv.{2 * j}.bias"], 1, 1),
                            sd[f"{p}block{i + 1}.conv.{2 * j + 1}.weight"])
            xs.append(x)
            mf.append([])
        else:
            pe = f"{p}patch_embed{i + 1}."
            if i == 3:
                ys, k = [], 0
                for a in range(3):
                    for j in range(2 ** a):
                        ys.append(F.conv2d(xs[-1 - a], sd[f"{pe}layers.{k}.weight"], sd[f"{pe}layers.{k}.bias"], 2 ** (a + 1), 1 + j, 1 + j))
                        k += 1
                x = F.conv2d(torch.cat(ys, 1), sd[pe + "proj.weight"], sd[pe + "proj.bias"])
            else:
                x = F.conv2d(x, sd[pe + "proj.weight"], sd[pe + "proj.bias"], 2, 1)
            H, W = x.shape[2:]
            C = x.shape[1]
            x = F.layer_norm(x.flatten(2).transpose(1, 2), (C,), sd[pe + "norm.weight"], sd[pe + "norm.bias"], 1e-5)
            cor = torch.cat([torch.linspace(-1.0, 1.0, W).view(1, 1, W, 1).expand(2 * B, H, -1, -1),
                             torch.linspace(-1.0, 1.0, H).view(1, H, 1, 1).expand(2 * B, -1, W, -1)], -1)
            mos = []
            for j in range(cfg["depths"][i]):
                x, mo = _block(sd, f"{p}block{i + 1}.{j}.", x, cor, H, W, B, cfg["num_heads"][i - 3], ws, 0 if j % 2 == 0 else ws // 2)
                mos.append(mo.reshape(2 * B, H, W, -1).permute(0, 3, 1, 2))
            x = F.layer_norm(x, (C,), sd[f"{p}norm{i + 1}.weight"], sd[f"{p}norm{i + 1}.bias"], 1e-6)
            x = x.reshape(2 * B, H, W, C).permute(0, 3, 1, 2).contiguous()
            mf.append(torch.cat(mos, 1))
        af.append(x)
    return af, mf


def _head(sd, p, scale, motion_feature, x, flow):
    mfe = F.pixel_shuffle(F.pixel_shuffle(motion_feature, 2), 2)
    if scale != 4:
        x = F.interpolate(x, scale_factor=4.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        if scale != 4:
            flow = F.interpolate(flow, scale_factor=4.0 / scale, mode="bilinear", align_corners=False) * 4.0 / scale
        x = torch.cat((x, flow), 1)
    x = torch.cat([mfe, x], 1)
    for j in range(3):
        x = _conv_prelu(sd, f"{p}conv.{j}", x)
    if scale != 4:
        x = F.interpolate(x, scale_factor=scale // 4, mode="bilinear", align_corners=False)
        return x[:, :4] * (scale // 4), x[:, 4:5]
    return x[:, :4], x[:, 4:5]


def _unet(sd, p, img0, img1, w0, w1, mask, flow, c0, c1):
    def down(name, x):
        return _conv_prelu(sd, f"{p}{name}.conv2", _conv_prelu(sd, f"{p}{name}.conv1", x, 2))

    def up(name, x):
        return F.prelu(F.conv_transpose2d(x, sd[f"{p}{name}.0.weight"], sd[f"{p}{name}.0.bias"], 2, 1), sd[f"{p}{name}.1.weight"])

    s0 = down("down0", torch.cat((img0, img1, w0, w1, mask, flow, c0[0], c1[0]), 1))
    s1 = down("down1", torch.cat((s0, c0[1], c1[1]), 1))
    s2 = down("down2", torch.cat((s1, c0[2], c1[2]), 1))
    s3 = down("down3", torch.cat((s2, c0[3], c1[3]), 1))
    x = up("up0", torch.cat((s3, c0[4], c1[4]), 1))
    x = up("up1", torch.cat((x, s2), 1))
    x = up("up2", torch.cat((x, s1), 1))
    x = up("up3", torch.cat((x, s0), 1))
    return torch.sigmoid(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 1))


def net_forward(sd, cfg, x, timestep=0.5):
    """MultiScaleFlow.forward :107-140 -> dict(af, mf, flow, mask, merged, pred)."""
    img0, img1 = x[:, :3], x[:, 3:6]
    B = x.shape[0]
    af, mf = feature_bone(sd, cfg, img0, img1)
    flow = mask = None
    w0, w1 = img0, img1
    for i in range(2):
        m, a = mf[-1 - i], af[-1 - i]
        # stage 0 weighs the second image's motion by (1 - t) as a tensor, later stages by the float (:123-128): same value
        feat = torch.cat([timestep * m[:B], (1 - timestep) * m[B:], a[:B], a[B:]], 1)
        if flow is None:
            flow, mask = _head(sd, f"block.{i}.", cfg["scales"][-1 - i], feat, torch.cat((img0, img1), 1), None)
        else:
            fd, md = _head(sd, f"block.{i}.", cfg["scales"][-1 - i], feat, torch.cat((img0, img1, w0, w1, mask), 1), flow)
            flow, mask = flow + fd, mask + md
        w0, w1 = warp(img0, flow[:, :2]), warp(img1, flow[:, 2:4])
    sig = torch.sigmoid(mask)
    merged = w0 * sig + w1 * (1 - sig)
    c0, c1, fl = [], [], flow
    for a in af:
        c0.append(warp(a[:B], fl[:, 0:2]))
        c1.append(warp(a[B:], fl[:, 2:4]))
        fl = F.interpolate(fl, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5
    res = _unet(sd, "unet.", img0, img1, w0, w1, mask, flow, c0, c1)[:, :3] * 2 - 1
    return dict(af=af, mf=mf[3:], flow=flow, mask=mask, merged=merged, pred=torch.clamp(merged + res, 0, 1))


def inference_fast_tta(sd, cfg, img0, img1, timestep=0.5):
    """Trainer.Model.inference(TTA=True, fast_TTA=True) :84-94: the pair and its 180-degree rotation in one batch, averaged."""
    imgs = torch.cat((img0, img1), 1)
    preds = net_forward(sd, cfg, torch.cat((imgs, imgs.flip(2).flip(3)), 0), timestep)["pred"]
    return (preds[0] + preds[1].flip(1).flip(2)).unsqueeze(0) / 2.0


def vfi_process(video, infer, video_len, out_size=(1280, 720)):
    """i2v_enhance_interface.vfi_process :30-61.  video: list/array of uint8 RGB frames [H, W, 3]; infer(I0, I2) -> [1, 3, H, W] in
    [0, 1] on BGR float frames.  Returns video_len PIL frames resized to out_size (PIL default BICUBIC)."""
    import numpy as np
    from PIL import Image
    v = np.stack([f[:, :, :3] / 255.0 for f in video[: video_len // 2 + 1]], 0)[:, :, :, ::-1]
    v = torch.from_numpy(v.copy()).permute(0, 3, 1, 2).float()
    to_u8 = lambda t: (t.permute(1, 2, 0).numpy() * 255.0).astype(np.uint8)[:, :, ::-1]
    frames = []
    for i in range(v.shape[0] - 1):
        frames.append(to_u8(v[i]))
        frames.append(to_u8(infer(v[i:i + 1], v[i + 1:i + 2])[0]))
    frames.append(to_u8(v[-1]))
    if video_len % 2 == 0:
        frames.append(to_u8(v[-1]))
    return [Image.fromarray(np.ascontiguousarray(f)).resize(out_size) for f in frames]
