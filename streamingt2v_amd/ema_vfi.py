"""EMA-VFI frame interpolation on the MI355X kernels: the last stage of inference_i2v.py
(`StreamingPipeline.interpolate_video` :211-224 -> i2v_enhance_interface.vfi_process :30-61 -> Trainer.Model.inference(TTA=True,
fast_TTA=True) :84-94 -> MultiScaleFlow.forward, model/flow_estimation.py:107-140).

    EMAVFI(cfg).load_state_dict(sd)          keys of the vendored network (``Model.net.state_dict()``; the shipped ``ours.pkl`` stores them
                                             under ``module.`` -- `convert_checkpoint` strips that like Trainer.load_model :36-47)
    .inference(img0, img1)                   two frames fp32 [H, W, 3] in [0, 1] (channels-last, the reference's BGR order) -> the middle
                                             frame [H, W, 3]; the pair and its 180-degree rotation run as one batch (fast TTA)
    vfi_process(video, vfi, video_len)       the reference's frame arithmetic: every input frame, the interpolated frame after it, the last
                                             frame (twice for an even target length), resized to 1280 x 720

Layout: every feature map is a channels-last "token" tensor [images * H * W, C32] in the 16-bit element type, channels zero-padded to a
multiple of 32 (the implicit-GEMM conv consumes 32-channel K slices).  Images, flows, the blend mask and warped images are fp32
channels-last (3 / 4 / 1 channels): geometry stays in fp32.
  * 3x3 convolutions (stride 1 / 2), 1x1 and linear layers: `ops.gemm` (svd_gemm).  ConvTranspose2d(4, 2, 1) = a 3x3 convolution to
    4 x Cout channels (each output parity uses a 2x2 subset of the taps) followed by a pixel shuffle.  The six dilated stride-4 / stride-8
    convolutions of CrossScalePatchEmbed gather their 9 taps with strided views (data movement) and run as a plain GEMM.
  * PReLU, depthwise-conv + GELU, 7x7-window inter-frame attention, backward warp, bilinear resize, the final blend: csrc/vfi.hip.
  * window partition / shift (`torch.roll`) / padding, pixel shuffle, channel concatenation: torch indexing (data movement only).
  * `cor_embed` (a Linear(2, motion_dim) over the constant coordinate grid) is folded into a per-resolution fp32 table when first used;
    `timestep` (0.5) is folded into the input channels of each flow head's first convolution.
"""
import math

import numpy as np
import torch

from . import ops
from .params import Spec, check_state_dict
from .video_model import _dev_bf16, _dev_f32, pack_conv3x3


def c32(c):
    return (c + 31) // 32 * 32


class VFIConfig:
    """config.py:9-31 (init_model_config) with the values of i2v_enhance_interface.vfi_init :16-17."""

    def __init__(self, F=32, depth=(2, 2, 2, 4, 4), window=7, timestep=0.5):
        self.F, self.depths, self.window, self.timestep = F, tuple(depth), window, timestep
        self.embed_dims = [F, 2 * F, 4 * F, 8 * F, 16 * F]
        self.motion_dims = [0, 0, 0, 8 * F // depth[-2], 16 * F // depth[-1]]
        self.num_heads = [8 * F // 32, 16 * F // 32]
        self.scales = [4, 8, 16]
        self.hidden_dims = [4 * F, 4 * F]
        self.c = F
        assert window == 7 and all(self.embed_dims[3 + i] // self.num_heads[i] == 32 for i in range(2)), "svd_window_attn_7x7: 7x7 windows, head dim 32"
        assert all(self.motion_dims[3 + i] % self.num_heads[i] == 0 and self.motion_dims[3 + i] // self.num_heads[i] <= 16 for i in range(2))


def _pad_cols(t, width):
    """[rows, c] -> [rows, width] zero-padded (data movement, one concatenation kernel)."""
    if t.shape[1] == width:
        return t.contiguous()
    return torch.cat([t, torch.zeros((t.shape[0], width - t.shape[1]), dtype=t.dtype, device=t.device)], 1)


def _cat_tokens(parts, width=None):
    """parts: [(tokens [rows, >= c], c)] -> [rows, c32(sum c)]: the channel concatenation of torch.cat(..., 1) without the pad channels
    (one concatenation kernel: every output element is written once)."""
    total = sum(c for _, c in parts)
    width = width or c32(total)
    pieces = [t[:, :c] for t, c in parts]
    if width > total:
        pieces.append(torch.zeros((parts[0][0].shape[0], width - total), dtype=parts[0][0].dtype, device=parts[0][0].device))
    return torch.cat(pieces, 1)


class _Conv:
    """nn.Conv2d(cin, cout, 3, stride, 1) [+ nn.PReLU(cout)] on tokens."""

    def __init__(self, wkey, pkey, cin, cout, stride=1, f32_out=False):
        self.wkey, self.pkey, self.cin, self.cout, self.stride, self.f32_out = wkey, pkey, cin, cout, stride, f32_out
        self.cin_pad, self.cout_pad = c32(cin), (8 if f32_out else c32(cout))

    def spec(self, s):
        s.add(self.wkey + ".weight", self.cout, self.cin, 3, 3); s.add(self.wkey + ".bias", self.cout)
        if self.pkey:
            s.add(self.pkey + ".weight", self.cout)

    def load(self, sd, dev, in_scale=None):
        w = sd[self.wkey + ".weight"].detach().float()
        if in_scale is not None:
            w = w * in_scale.view(1, -1, 1, 1)
        self.w = _dev_bf16(pack_conv3x3(w, self.cin_pad, self.cout_pad), dev)
        b = torch.zeros(self.cout_pad); b[: self.cout] = sd[self.wkey + ".bias"].float()
        self.b = _dev_f32(b, dev)
        self.slope = None
        if self.pkey:
            a = torch.zeros(self.cout_pad); a[: self.cout] = sd[self.pkey + ".weight"].float()
            self.slope = _dev_f32(a, dev)

    def __call__(self, x, n, H, W):
        ho, wo = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
        y = ops.gemm(x, self.w, bias=self.b, out_f32=self.f32_out,
                     conv=dict(cin=self.cin_pad, hin=H, win=W, hout=ho, wout=wo, stride=self.stride, ups=0, frames=n))
        if self.slope is not None:
            ops.prelu_(y, self.slope)
        return y, ho, wo


class _Deconv:
    """nn.ConvTranspose2d(cin, cout, 4, 2, 1) + nn.PReLU(cout) (refine.py:15-19)."""

    def __init__(self, key, cin, cout):
        self.key, self.cin, self.cout = key, cin, cout
        self.cin_pad, self.cout_pad = c32(cin), c32(cout)

    def spec(self, s):
        s.add(self.key + ".0.weight", self.cin, self.cout, 4, 4); s.add(self.key + ".0.bias", self.cout); s.add(self.key + ".1.weight", self.cout)

    def load(self, sd, dev):
        w = sd[self.key + ".0.weight"].detach().float()                         # [cin, cout, ky, kx]
        wc = torch.zeros(2, 2, self.cout_pad, self.cin, 3, 3)                   # output parity (py, px), tap (ty, tx): ky = py + 3 - 2 ty
        for py in range(2):
            for px in range(2):
                for ty in range(3):
                    for tx in range(3):
                        ky, kx = py + 3 - 2 * ty, px + 3 - 2 * tx
                        if 0 <= ky <= 3 and 0 <= kx <= 3:
                            wc[py, px, : self.cout, :, ty, tx] = w[:, :, ky, kx].t()
        self.w = _dev_bf16(pack_conv3x3(wc.view(4 * self.cout_pad, self.cin, 3, 3), self.cin_pad), dev)
        b = torch.zeros(4, self.cout_pad); b[:, : self.cout] = sd[self.key + ".0.bias"].float()
        self.b = _dev_f32(b.view(-1), dev)
        a = torch.zeros(self.cout_pad); a[: self.cout] = sd[self.key + ".1.weight"].float()
        self.slope = _dev_f32(a, dev)

    def __call__(self, x, n, H, W):
        y = ops.gemm(x, self.w, bias=self.b, conv=dict(cin=self.cin_pad, hin=H, win=W, hout=H, wout=W, stride=1, ups=0, frames=n))
        cp = self.cout_pad
        y = y.view(n, H, W, 2, 2, cp).permute(0, 1, 3, 2, 4, 5).reshape(n * 2 * H * 2 * W, cp)      # pixel shuffle (data movement)
        return ops.prelu_(y, self.slope), 2 * H, 2 * W


class _Linear:
    def __init__(self, key, cin, cout):
        self.key, self.cin, self.cout = key, cin, cout

    def spec(self, s):
        s.add(self.key + ".weight", self.cout, self.cin); s.add(self.key + ".bias", self.cout)

    def load(self, sd, dev):
        w = torch.zeros(c32(self.cout), c32(self.cin)); w[: self.cout, : self.cin] = sd[self.key + ".weight"].float()
        b = torch.zeros(c32(self.cout)); b[: self.cout] = sd[self.key + ".bias"].float()
        self.w, self.b = _dev_bf16(w, dev), _dev_f32(b, dev)

    def __call__(self, x, **kw):
        return ops.gemm(x, self.w, bias=self.b, **kw)


class _LN:
    def __init__(self, key, c, eps):
        self.key, self.c, self.eps = key, c, eps

    def spec(self, s):
        s.add(self.key + ".weight", self.c); s.add(self.key + ".bias", self.c)

    def load(self, sd, dev):
        self.g, self.b = _dev_f32(sd[self.key + ".weight"], dev), _dev_f32(sd[self.key + ".bias"], dev)

    def __call__(self, x):
        return ops.layernorm(x, self.g, self.b, eps=self.eps)


def _region_mask(Hp, Wp, hs, wsl, ws):
    m = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for h in hs:
        for w in wsl:
            m[:, h, w, :] = cnt
            cnt += 1
    mw = m.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def window_geometry(h, w, ws, shift):
    """Centre padding and the additive attention mask of one block for an h x w token grid (feature_extractor.py:29-58, 231-262):
    -100 between tokens of different regions (padding border / wrapped-around shift regions).  -> (pad_h, pad_w, mask [nW, 49, 49] | None)."""
    ph, pw = math.ceil(h / ws) * ws - h, math.ceil(w / ws) * ws - w
    mask = None
    if ph > 0 or pw > 0:
        mask = _region_mask(h + ph, w + pw, (slice(0, ph // 2), slice(ph // 2, h + ph // 2), slice(h + ph // 2, None)),
                            (slice(0, pw // 2), slice(pw // 2, w + pw // 2), slice(w + pw // 2, None)), ws)
    if shift:
        sl = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
        sm = _region_mask(h + ph, w + pw, sl, sl, ws)
        if mask is not None:
            sm = torch.where(mask != 0, torch.full_like(sm, -100.0), sm)
        mask = sm
    return ph, pw, mask


class _Block:
    """MotionFormerBlock (feature_extractor.py:174-290)."""

    def __init__(self, p, dim, motion_dim, heads, ws, shift):
        self.p, self.dim, self.md, self.heads, self.ws, self.shift = p, dim, motion_dim, heads, ws, shift
        self.norm1, self.norm2 = _LN(p + "norm1", dim, 1e-6), _LN(p + "norm2", dim, 1e-6)
        self.q, self.kv, self.proj = _Linear(p + "attn.q", dim, dim), _Linear(p + "attn.kv", dim, 2 * dim), _Linear(p + "attn.proj", dim, dim)
        self.motion_proj = _Linear(p + "attn.motion_proj", motion_dim, motion_dim)
        self.fc1, self.fc2 = _Linear(p + "mlp.fc1", dim, 4 * dim), _Linear(p + "mlp.fc2", 4 * dim, dim)
        self._geo = {}

    def spec(self, s):
        p = self.p
        self.norm1.spec(s)
        self.q.spec(s); self.kv.spec(s)
        s.add(p + "attn.cor_embed.weight", self.md, 2); s.add(p + "attn.cor_embed.bias", self.md)
        self.proj.spec(s); self.motion_proj.spec(s)
        self.norm2.spec(s); self.fc1.spec(s)
        s.add(p + "mlp.dwconv.dwconv.weight", 4 * self.dim, 1, 3, 3); s.add(p + "mlp.dwconv.dwconv.bias", 4 * self.dim)
        self.fc2.spec(s)

    def load(self, sd, dev):
        for m in (self.norm1, self.norm2, self.q, self.kv, self.proj, self.motion_proj, self.fc1, self.fc2):
            m.load(sd, dev)
        p = self.p
        self.w_ce, self.b_ce = _dev_f32(sd[p + "attn.cor_embed.weight"], dev), _dev_f32(sd[p + "attn.cor_embed.bias"], dev)
        self.w_dw = _dev_f32(sd[p + "mlp.dwconv.dwconv.weight"].float().view(4 * self.dim, 9).t().contiguous(), dev)      # [9, C]
        self.b_dw = _dev_f32(sd[p + "mlp.dwconv.dwconv.bias"], dev)
        self.dev = dev

    def _windows(self, t, n, H, W, ph, pw):
        """[n*H*W, C] -> centre pad, shift, partition -> [n_win*49, C] (data movement)."""
        ws, C = self.ws, t.shape[1]
        x = torch.nn.functional.pad(t.view(n, H, W, C), (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        if self.shift:
            x = torch.roll(x, (-self.shift, -self.shift), (1, 2))
        Hp, Wp = H + ph, W + pw
        return x.view(n, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)

    def _unwindows(self, t, n, H, W, ph, pw):
        ws, C = self.ws, t.shape[1]
        Hp, Wp = H + ph, W + pw
        x = t.view(n, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(n, Hp, Wp, C)
        if self.shift:
            x = torch.roll(x, (self.shift, self.shift), (1, 2))
        return x[:, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W].reshape(n * H * W, C)

    def geometry(self, n, H, W):
        """(pad_h, pad_w, mask, cor_embed table fp32 [n_win*49, md]) -- constants of the (n, H, W) token grid."""
        key = (n, H, W)
        if key not in self._geo:
            ph, pw, mask = window_geometry(H, W, self.ws, self.shift)
            dev = self.dev
            cor = torch.cat([torch.linspace(-1.0, 1.0, W, device=dev).view(1, 1, W, 1).expand(n, H, -1, -1),
                             torch.linspace(-1.0, 1.0, H, device=dev).view(1, H, 1, 1).expand(n, -1, W, -1)], -1).reshape(n * H * W, 2)   # get_cor :452-461
            ce = torch.addmm(self.b_ce, self._windows(cor.contiguous(), n, H, W, ph, pw), self.w_ce.t()).contiguous()   # parameter folding, fp32
            self._geo[key] = (ph, pw, mask.to(dev).contiguous() if mask is not None else None, ce)
        return self._geo[key]

    def __call__(self, x, n, H, W):
        """x [n*H*W, dim] (n = 2 * pairs: first half frame 0, second half frame 1) -> (x, motion [n*H*W, c32(md)])."""
        ph, pw, mask, ce = self.geometry(n, H, W)
        n_win = n * ((H + ph) // self.ws) * ((W + pw) // self.ws)
        xn = self.norm1(self._windows(x, n, H, W, ph, pw))                      # LayerNorm AFTER padding, as the reference (:263)
        q, kv = self.q(xn), self.kv(xn)
        xa, dc = ops.window_attn_7x7(q, kv, ce, mask, n_win, self.heads, self.md // self.heads, 32 ** -0.5)
        motion = self.motion_proj(_pad_cols(dc, c32(self.md)))
        xw = self.proj(xa, residual=xn)                                         # x_norm + attn (:271): the residual is the NORMED tokens
        x = self._unwindows(xw, n, H, W, ph, pw)
        motion = self._unwindows(motion, n, H, W, ph, pw)
        h = ops.dwconv3x3_gelu(self.fc1(self.norm2(x)), self.w_dw, self.b_dw, n, H, W)
        return self.fc2(h, residual=x), motion


class EMAVFI:
    def __init__(self, cfg=None):
        self.cfg = cfg = cfg or VFIConfig()
        F, E, p = cfg.F, cfg.embed_dims, "feature_bone."
        # conv stages 1-3 (ConvBlock + strided patch_embed2/3, feature_extractor.py:293-319, 427-436)
        self.stage = []
        for i in range(3):
            mods = []
            if i > 0:
                mods.append(_Conv(f"{p}patch_embed{i + 1}.0", f"{p}patch_embed{i + 1}.1", E[i - 1], E[i], stride=2))
            for j in range(cfg.depths[i]):
                mods.append(_Conv(f"{p}block{i + 1}.conv.{2 * j}", f"{p}block{i + 1}.conv.{2 * j + 1}", (3 if i == 0 else E[i]) if j == 0 else E[i], E[i]))
            self.stage.append(mods)
        # CrossScalePatchEmbed (:366-410): 1 + 2 + 4 dilated strided convs over stages 3, 2, 1 -> 1x1 proj -> LayerNorm
        self.cross = [(a, j, f"{p}patch_embed4.layers.{k}") for k, (a, j) in enumerate((a, j) for a in range(3) for j in range(2 ** a))]
        self.cross_proj = _Linear(f"{p}patch_embed4.proj", 7 * F, E[3])
        self.pe_norm = [_LN(f"{p}patch_embed4.norm", E[3], 1e-5), _LN(f"{p}patch_embed5.norm", E[4], 1e-5)]
        self.pe5 = _Conv(f"{p}patch_embed5.proj", None, E[3], E[4], stride=2)
        self.blocks = [[_Block(f"{p}block{i + 1}.{j}.", E[i], cfg.motion_dims[i], cfg.num_heads[i - 3], cfg.window, 0 if j % 2 == 0 else cfg.window // 2)
                        for j in range(cfg.depths[i])] for i in (3, 4)]
        self.out_norm = [_LN(f"{p}norm4", E[3], 1e-6), _LN(f"{p}norm5", E[4], 1e-6)]
        # flow heads (flow_estimation.py:17-43, 51-55): stage 0 on (/16 features, scale 16), stage 1 on (/8, scale 8)
        self.heads = []
        for i in range(2):
            feat = cfg.motion_dims[-1 - i] * cfg.depths[-1 - i] + E[-1 - i]
            cin = feat * 2 // 16 + (6 if i == 0 else 17)
            c = cfg.hidden_dims[-1 - i]
            self.heads.append([_Conv(f"block.{i}.conv.0.0", f"block.{i}.conv.0.1", cin, c), _Conv(f"block.{i}.conv.1.0", f"block.{i}.conv.1.1", c, c),
                               _Conv(f"block.{i}.conv.2.0", f"block.{i}.conv.2.1", c, 5, f32_out=True)])
        c = 2 * cfg.c
        self.down = [[_Conv(f"unet.down{k}.conv1.0", f"unet.down{k}.conv1.1", ci, co, stride=2), _Conv(f"unet.down{k}.conv2.0", f"unet.down{k}.conv2.1", co, co)]
                     for k, (ci, co) in enumerate(((17 + c, 2 * c), (4 * c, 4 * c), (8 * c, 8 * c), (16 * c, 16 * c)))]
        self.up = [_Deconv(f"unet.up{k}", ci, co) for k, (ci, co) in enumerate(((32 * c, 8 * c), (16 * c, 4 * c), (8 * c, 2 * c), (4 * c, c)))]
        self.last = _Conv("unet.conv", None, c, 3, f32_out=True)
        self.loaded = False

    # ---- parameters -------------------------------------------------------------------------------------------------------
    def spec(self):
        """Same keys, shapes and ORDER as the vendored ``Model.net.state_dict()``."""
        cfg, s, p = self.cfg, Spec(), "feature_bone."
        F, E = cfg.F, cfg.embed_dims
        for mods in self.stage:
            for m in mods:
                m.spec(s)
        for k in (0, 1):                                                     # registration order of MotionFormer.__init__: norm, patch_embed, block
            self.out_norm[k].spec(s)
            if k == 0:
                for a, j, key in self.cross:
                    s.add(key + ".weight", F, E[2 - a], 3, 3); s.add(key + ".bias", F)
                s.add(f"{p}patch_embed4.proj.weight", E[3], 7 * F, 1, 1); s.add(f"{p}patch_embed4.proj.bias", E[3])
            else:
                self.pe5.spec(s)
            self.pe_norm[k].spec(s)
            for b in self.blocks[k]:
                b.spec(s)
        for h in self.heads:
            for m in h:
                m.spec(s)
        for d in self.down:
            for m in d:
                m.spec(s)
        for u in self.up:
            u.spec(s)
        self.last.spec(s)
        return s

    @staticmethod
    def convert_checkpoint(param):
        """Trainer.Model.load_model :36-42: strip ``module.``, drop cached ``attn_mask`` / ``HW`` buffers."""
        return {k.replace("module.", ""): v for k, v in param.items() if "module." in k and "attn_mask" not in k and "HW" not in k}

    def load_state_dict(self, sd, device="cuda"):
        check_state_dict(self.spec(), sd)
        cfg, dev, p = self.cfg, device, "feature_bone."
        F, E = cfg.F, cfg.embed_dims
        for mods in self.stage:
            for m in mods:
                m.load(sd, dev)
        self.cross_w = []
        for a, j, key in self.cross:
            cin = E[2 - a]
            kp = c32(9 * cin)
            w = torch.zeros(c32(F), kp); w[:F, : 9 * cin] = sd[key + ".weight"].float().permute(0, 2, 3, 1).reshape(F, 9 * cin)      # K = (ky, kx, c)
            b = torch.zeros(c32(F)); b[:F] = sd[key + ".bias"].float()
            self.cross_w.append((_dev_bf16(w, dev), _dev_f32(b, dev)))
        proj_sd = {f"{p}patch_embed4.proj.weight": sd[f"{p}patch_embed4.proj.weight"].float().view(E[3], 7 * F), f"{p}patch_embed4.proj.bias": sd[f"{p}patch_embed4.proj.bias"]}
        self.cross_proj.load(proj_sd, dev)
        for m in self.pe_norm + self.out_norm + [self.pe5]:
            m.load(sd, dev)
        for blks in self.blocks:
            for b in blks:
                b.load(sd, dev)
        t = cfg.timestep
        for i, h in enumerate(self.heads):
            mfc = cfg.motion_dims[-1 - i] * cfg.depths[-1 - i] // 16         # pixel-shuffled motion channels of ONE frame
            sc = torch.ones(h[0].cin); sc[:mfc] = t; sc[mfc:2 * mfc] = 1 - t  # t * mf[:B], (1 - t) * mf[B:]  (flow_estimation.py:123-128)
            h[0].load(sd, dev, in_scale=sc)
            h[1].load(sd, dev); h[2].load(sd, dev)
        for d in self.down:
            for m in d:
                m.load(sd, dev)
        for u in self.up:
            u.load(sd, dev)
        self.last.load(sd, dev)
        self.device, self.loaded = dev, True
        self._mult = {}
        return self

    # ---- forward ----------------------------------------------------------------------------------------------------------
    def _cross_scale(self, xs, n):
        """CrossScalePatchEmbed.forward: xs = [(tokens, C, H, W)] of stages 1..3 -> tokens [n*H8*W8, E3] before LayerNorm."""
        F = self.cfg.F
        outs, ho, wo = [], None, None
        for (a, j, _), (w, b) in zip(self.cross, self.cross_w):
            t, C, H, W = xs[2 - a]
            s, d = 2 ** (a + 1), 1 + j
            Ho, Wo = (H + 2 * d - 2 * d - 1) // s + 1, (W + 2 * d - 2 * d - 1) // s + 1
            xp = torch.nn.functional.pad(t.view(n, H, W, -1)[..., :C], (0, 0, d, d, d, d))
            taps = [xp[:, ky * d: ky * d + s * (Ho - 1) + 1: s, kx * d: kx * d + s * (Wo - 1) + 1: s] for ky in range(3) for kx in range(3)]
            col = _pad_cols(torch.cat(taps, -1).reshape(n * Ho * Wo, 9 * C), w.shape[1])          # im2col gather (data movement)
            outs.append((ops.gemm(col, w, bias=b), F))
            assert ho in (None, Ho) and wo in (None, Wo)
            ho, wo = Ho, Wo
        return self.cross_proj(_cat_tokens(outs, c32(7 * F))), ho, wo

    def feature_bone(self, imgs, n, H, W):
        """MotionFormer.forward (:464-497).  imgs: fp32 [n*H*W, 3] (frame-0 images then frame-1 images).
        -> af: 5 x (tokens, C, h, w);  mf: 2 x (tokens [., md * depth], C, h, w) for stages 4, 5."""
        cfg = self.cfg
        x, h, w = _pad_cols(ops.to_elem(imgs.contiguous()), 32), H, W          # cast the 3 channels, then pad in 16 bit
        af, mf, xs = [], [], []
        for i in range(3):
            for m in self.stage[i]:
                x, h, w = m(x, n, h, w)
            xs.append((x, cfg.embed_dims[i], h, w))
            af.append(xs[-1])
        for k, i in enumerate((3, 4)):
            if i == 3:
                x, h, w = self._cross_scale(xs, n)
            else:
                x, h, w = self.pe5(x, n, h, w)
            x = self.pe_norm[k](x)
            mos = []
            for blk in self.blocks[k]:
                x, mo = blk(x, n, h, w)
                mos.append((mo, blk.md))
            x = self.out_norm[k](x)
            af.append((x, cfg.embed_dims[i], h, w))
            mf.append((_cat_tokens(mos), sum(c for _, c in mos), h, w))
        return af, mf

    def _mults(self, vals):
        key = tuple(vals)
        if key not in self._mult:
            self._mult[key] = torch.tensor(list(vals), dtype=torch.float32, device=self.device)
        return self._mult[key]

    @staticmethod
    def _pixel_shuffle4(t, n, h, w, c):
        """nn.PixelShuffle(2) twice on tokens [n*h*w, >= c] -> [n*4h*4w, c/16] (data movement)."""
        x = t[:, :c].reshape(n, h, w, c)
        for _ in range(2):
            c //= 4
            x = x.reshape(n, h, w, c, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2 * h, 2 * w, c)
            h, w = 2 * h, 2 * w
        return x.reshape(n * h * w, c), c

    def net_forward(self, img0, img1, B, H, W, want=False):
        """MultiScaleFlow.forward (:107-140).  img0 / img1: fp32 [B*H*W, 3].  -> pred fp32 [B*H*W, 3] (+ intermediates when want)."""
        cfg = self.cfg
        assert H % 16 == 0 and W % 16 == 0, "EMA-VFI needs frame sides divisible by 16 (four stride-2 stages)"
        af, mf = self.feature_bone(torch.cat([img0, img1], 0), 2 * B, H, W)
        fm = torch.zeros((B * H * W, 8), dtype=torch.float32, device=self.device)            # flow (4) | mask (1) | pad
        w0, w1 = img0, img1
        half = lambda tok, hw: (tok[: B * hw], tok[B * hw:])
        for i in range(2):
            (mt, mc, h, w), (at, ac, _, _) = mf[-1 - i], af[-1 - i]
            m0, m1 = half(mt, h * w)
            a0, a1 = half(at, h * w)
            feat = _cat_tokens([(m0, mc), (m1, mc), (a0, ac), (a1, ac)], 2 * (mc + ac))     # exact width: the pixel shuffle regroups channels
            mfe, cps = self._pixel_shuffle4(feat, B, h, w, 2 * (mc + ac))
            scale = cfg.scales[-1 - i]
            hq, wq = 4 * h, 4 * w
            x = torch.cat([img0, img1] if i == 0 else [img0, img1, w0, w1, fm[:, 4:5]], 1).contiguous()
            if scale != 4:
                x, _, _ = ops.resize_bilinear(x, B, H, W, 4.0 / scale)
            parts = [(mfe, cps), (ops.to_elem(x), x.shape[1])]
            if i > 0:
                fl = fm[:, :4].contiguous()
                if scale != 4:
                    fl, _, _ = ops.resize_bilinear(fl, B, H, W, 4.0 / scale, mult=self._mults([4.0 / scale] * 4))
                parts.append((ops.to_elem(fl), 4))
            y = _cat_tokens(parts)
            for conv in self.heads[i]:
                y, _, _ = conv(y, B, hq, wq)                                                  # last conv: fp32 [., 8] = flow (4) | mask | 0
            up = scale // 4
            ops.resize_bilinear(y, B, hq, wq, float(up), mult=self._mults([float(up)] * 4 + [1.0, 0.0, 0.0, 0.0]), out=fm, accumulate=i > 0)
            w0, w1 = ops.warp_bilinear(img0, fm[:, 0:2], B, H, W), ops.warp_bilinear(img1, fm[:, 2:4], B, H, W)
        # warp_features (:59-67) + Unet (refine.py:61-71)
        c0, c1, fl, fh, fw = [], [], fm[:, :4].contiguous(), H, W
        for lvl, (t, C, h, w) in enumerate(af):
            assert (h, w) == (fh, fw)
            t0, t1 = half(t, h * w)
            c0.append(ops.warp_bilinear(t0, fl[:, 0:2], B, h, w)); c1.append(ops.warp_bilinear(t1, fl[:, 2:4], B, h, w))
            if lvl < 4:
                fl, fh, fw = ops.resize_bilinear(fl, B, fh, fw, 0.5, mult=self._mults([0.5] * 4))
        geo = ops.to_elem(torch.cat([img0, img1, w0, w1, fm[:, 4:5], fm[:, :4]], 1).contiguous())
        x = _cat_tokens([(geo, 17), (c0[0], af[0][1]), (c1[0], af[0][1])])
        skips, h, w = [], H, W
        for k in range(4):
            for conv in self.down[k]:
                x, h, w = conv(x, B, h, w)
            skips.append((x, self.down[k][1].cout))
            if k < 3:
                x = _cat_tokens([skips[-1], (c0[k + 1], af[k + 1][1]), (c1[k + 1], af[k + 1][1])])
        x = _cat_tokens([skips[3], (c0[4], af[4][1]), (c1[4], af[4][1])])
        for k in range(4):
            x, h, w = self.up[k](x, B, h, w)
            if k < 3:
                x = _cat_tokens([(x, self.up[k].cout), skips[2 - k]])
        u, _, _ = self.last(x, B, H, W)
        if want:
            pred, merged = ops.vfi_merge(w0.contiguous(), w1.contiguous(), fm[:, 4:5], u, want_merged=True)
            return dict(af=af, mf=mf, fm=fm, merged=merged, pred=pred)
        return ops.vfi_merge(w0.contiguous(), w1.contiguous(), fm[:, 4:5], u)

    @torch.no_grad()
    def inference(self, img0, img1, TTA=True, fast_TTA=True, want_uint8=False):
        """Trainer.Model.inference(img0, img1, TTA, timestep = cfg.timestep, fast_TTA) :84-101; the reference calls it with TTA=True,
        fast_TTA=True (i2v_enhance_interface.py:46).  img0 / img1: fp32 [H, W, 3] in [0, 1] on the device.
          fast_TTA        the pair and its 180-degree rotation as ONE batch of two, averaged            (:90-94)
          TTA only        the same average from two separate forwards                                   (:99-101)
          neither         one forward                                                                   (:96-98)
        -> the middle frame fp32 [H, W, 3] (and, with want_uint8, its (x * 255).astype(uint8) truncation, i2v_enhance_interface.py:46-47)."""
        assert self.loaded, "load_state_dict first"
        H, W = img0.shape[:2]
        rot = lambda t: t.flip(0).flip(1)                                                       # imgs.flip(2).flip(3) on NCHW
        rows = lambda ts: torch.stack(ts).reshape(len(ts) * H * W, 3).float().contiguous()
        if fast_TTA:
            pred = self.net_forward(rows([img0, rot(img0)]), rows([img1, rot(img1)]), 2, H, W)
        elif TTA:
            pred = torch.cat([self.net_forward(rows([img0]), rows([img1]), 1, H, W), self.net_forward(rows([rot(img0)]), rows([rot(img1)]), 1, H, W)], 0)
        else:
            pred = self.net_forward(rows([img0]), rows([img1]), 1, H, W)
            u8 = (pred * 255.0).to(torch.uint8).view(H, W, 3) if want_uint8 else None            # same truncation, plain cast
            return (pred.view(H, W, 3), u8) if want_uint8 else pred.view(H, W, 3)
        out, u8 = ops.vfi_tta_average(pred.contiguous(), H, W, want_uint8)
        return (out.view(H, W, 3), u8) if want_uint8 else out.view(H, W, 3)


# (np.float32(k / 255.) * 255.0).astype(uint8) for every uint8 k: the reference's round trip of the pass-through frames is NOT the identity
_UNIT_LUT = (np.arange(256) / 255.0).astype(np.float32)
_PASS_LUT = ((np.arange(256) / 255.0).astype(np.float32) * np.float32(255.0)).astype(np.uint8)


def vfi_process(video, vfi, video_len, out_size=(1280, 720), device="cuda", group=None, sharded=False):
    """i2v_enhance_interface.vfi_process :30-61.  video: sequence of uint8 RGB frames [H, W, 3]; vfi: EMAVFI.
    -> list of `video_len` PIL frames (input frame, interpolated frame, ..., last frame [twice when video_len is even]) at out_size.
    sharded=True (inside an initialised torch.distributed job): the frame pairs are independent, so rank r interpolates pairs r, r + world,
    ... and one all-gather of the uint8 middle frames (2.8 MB each at 720 x 1280) gives every rank the full video -- the reference asserts
    a single device here (i2v_enhance_interface.py:26)."""
    from PIL import Image
    frames = [np.asarray(f)[:, :, :3] for f in video[: video_len // 2 + 1]]
    bgr = [torch.from_numpy(_UNIT_LUT[np.ascontiguousarray(f[:, :, ::-1])]).to(device) for f in frames]          # i / 255. -> fp32, BGR (:33-37)
    n_pairs = len(frames) - 1
    mid = lambda i: vfi.inference(bgr[i], bgr[i + 1], want_uint8=True)[1]
    if sharded and n_pairs > 0:
        import torch.distributed as dist
        from . import parallel
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        per_rank = (n_pairs + world - 1) // world
        H, W = frames[0].shape[:2]
        mine = [mid(s * world + rank) if s * world + rank < n_pairs else torch.zeros((H, W, 3), dtype=torch.uint8, device=device)
                for s in range(per_rank)]                                                                        # padded: equal contributions
        send = torch.stack(mine, 0).contiguous()
        recv = [torch.empty_like(send) for _ in range(world)]
        parallel.all_gather(recv, send, group=group)
        mids = [recv[i % world][i // world] for i in range(n_pairs)]
    else:
        mids = [mid(i) for i in range(n_pairs)]
    out = []
    for i in range(n_pairs):
        out.append(_PASS_LUT[frames[i]])
        out.append(mids[i].cpu().numpy()[:, :, ::-1])
    out.append(_PASS_LUT[frames[-1]])
    if video_len % 2 == 0:
        out.append(_PASS_LUT[frames[-1]])
    return [Image.fromarray(np.ascontiguousarray(f)).resize(out_size) for f in out]
