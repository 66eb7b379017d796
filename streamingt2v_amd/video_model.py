"""Host-side mirror of the StreamingSVD denoiser networks, executing on the libsvdhip.so kernels.

Mirrors (same names, same state_dict keys, same forward contracts):
  * VideoResBlock / VideoUNet        <- code/models/diffusion/video_model.py:16-85, 88-618
  * SpatialVideoTransformer (+ BasicTransformerBlock, VideoTransformerBlock)
                                     <- code/models/svd/sgm/modules/video_attention.py:23-333, attention.py:464-593,679-804
  * ConditionalModel (CAM merger)    <- code/models/cam/conditioning.py:7-146
  * ControlNet (+ cond. embedding)   <- code/models/control/controlnet.py:51-121,124-554

What is different from the reference is the execution plan, not the arithmetic:
  * activations live channels-last as token matrices [frames*H*W, C] in bf16; there is not a single physical
    transpose: the "(b t) c h w <-> b c t h w" and "(b t) s c <-> (b s) t c" rearranges of the reference become
    addressing modes of the kernels (temporal-conv GEMM view, strided temporal attention);
  * bias / timestep-embedding add / residual / GEGLU / alpha-blend are GEMM epilogues;
  * cross-attention to the 1-token CLIP context (attn2) is softmax over ONE key == 1, hence
    attn2(x, ctx) == to_out(to_v(ctx)) exactly: computed as a per-frame vector and folded into the
    attn1 output epilogue.  Contexts with more than one token (APM, disabled in the shipped config.yaml:115) raise.
"""
import math

import torch

from . import ops
from .params import Spec

BF16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------------------------
# weight packing helpers (run once in prepare(); torch here is parameter plumbing, not the hot path)
# ----------------------------------------------------------------------------------------------------------------
def _dev_bf16(t, dev):
    """fp32 parameter -> packed 16-bit weight in the active element type (ops.ELEM)."""
    return t.detach().to(device=dev, dtype=torch.float32).to(ops.ELEM).contiguous()


def _dev_f32(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def pack_conv3x3(w, cin_pad=None, cout_pad=None):
    """[Cout, Cin, 3, 3] -> [Cout(_pad), 9 * Cin(_pad)] with K ordered (ky, kx, c)."""
    co, ci = w.shape[0], w.shape[1]
    cp = cin_pad or ci
    cop = cout_pad or co
    wp = torch.zeros(cop, 3, 3, cp, dtype=torch.float32, device=w.device)
    wp[:co, :, :, :ci] = w.detach().float().permute(0, 2, 3, 1)
    return wp.reshape(cop, 9 * cp)


def pack_tconv3(w):
    """[Cout, Cin, 3, 1, 1] -> [Cout, 3 * Cin] with K ordered (kt, c)."""
    co, ci = w.shape[0], w.shape[1]
    return w.detach().float()[:, :, :, 0, 0].permute(0, 2, 1).reshape(co, 3 * ci)


def pack_geglu(w, b):
    """GEGLU proj [2*inner, K]: rows (value | gate) -> interleaved in blocks of 32 rows (v0 g0 v1 g1 ...)."""
    half = w.shape[0] // 2
    assert half % 32 == 0
    K = w.shape[1]
    w = w.detach().float()
    b = b.detach().float()
    wv, wg = w[:half].reshape(half // 32, 32, K), w[half:].reshape(half // 32, 32, K)
    bv, bg = b[:half].reshape(half // 32, 32), b[half:].reshape(half // 32, 32)
    return torch.stack([wv, wg], 1).reshape(2 * half, K), torch.stack([bv, bg], 1).reshape(2 * half)


def pack_ff_fused(w1, b1, w2):
    """FeedForward(dim 320, GEGLU, mult 4) weights -> the packed image of svd_ff_geglu_fused (csrc/ff_fused.hip), a uint8 tensor.
    w1 [2 * hidden, C] = GEGLU.proj.weight (rows: value half, then gate half; attention.py:94-101), b1 [2 * hidden], w2 [C, hidden] = net[2].weight.
    Per chunk of 32 hidden units: 40 W1 fragments (k-step s, tile t: 16 value rows then the 16 gate rows of hidden 32 c + 16 t ..; lane l holds row
    l % 32, channels 16 s + 8 (l / 32) .. + 7), one 1-KiB slot with the chunk's 64 biases in tile-row order, 20 W2 fragments (tile t, output tile o:
    lane l holds output channel 32 o + l % 32 and the 8 hidden units its accumulator registers hold: 4 kg + e (e < 4), 8 + 4 kg + e - 4 otherwise)."""
    C, Hd = w2.shape
    assert w1.shape == (2 * Hd, C) and b1.shape == (2 * Hd,) and C % 32 == 0 and Hd % 64 == 0
    nch, NS, NO = Hd // 32, C // 16, C // 32
    w1, b1, w2 = w1.detach().float().cpu(), b1.detach().float().cpu(), w2.detach().float().cpu()
    ar = torch.arange
    c = ar(nch).view(-1, 1, 1, 1, 1); s_ = ar(NS).view(1, -1, 1, 1, 1); t = ar(2).view(1, 1, -1, 1, 1); l = ar(64).view(1, 1, 1, -1, 1); e = ar(8).view(1, 1, 1, 1, -1)
    m, kg = l & 31, l >> 5
    hid = 32 * c + 16 * t + (m & 15)
    row = torch.where(m < 16, hid, Hd + hid)
    f1 = w1[row, 16 * s_ + 8 * kg + e].to(ops.ELEM).contiguous().view(nch, -1).view(torch.uint8)                  # [nch, 40 KiB]
    m32 = ar(32).view(1, 1, -1)
    hb = 32 * ar(nch).view(-1, 1, 1) + 16 * ar(2).view(1, -1, 1) + (m32 & 15)
    bias = torch.zeros(nch, 256, dtype=torch.float32)
    bias[:, :64] = b1[torch.where(m32 < 16, hb, Hd + hb)].reshape(nch, 64)
    t2, o = ar(2).view(1, -1, 1, 1, 1), ar(NO).view(1, 1, -1, 1, 1)                                                # index order [c, t, o, l, e]
    u = torch.where(e < 4, 4 * kg + e, 8 + 4 * kg + (e - 4))
    f2 = w2[32 * o + m, 32 * c + 16 * t2 + u].to(ops.ELEM).contiguous().view(nch, -1).view(torch.uint8)           # [nch, 20 KiB]
    img = torch.cat([f1, bias.view(torch.uint8), f2], 1).contiguous()
    esz = torch.empty(0, dtype=ops.ELEM).element_size()          # 2 on the device; the CPU host-logic tests run with fp32 "elements" (tests/svd_shim.py)
    assert img.shape[1] == (2 * NS + 2 * NO) * 512 * esz + 1024
    return img.view(-1)


def pack_rowgemm320(w):
    """Linear(320, 320) weight [out, in] -> the packed image of svd_rowgemm320 (csrc/rowgemm.hip), a uint8 tensor: 200 MFMA A-operand fragments of 1 KiB,
    fragment 10 s + o (k-step s of 20, output tile o of 10): lane l holds W[32 o + l % 32][16 s + 8 (l // 32) .. + 7]."""
    assert tuple(w.shape) == (320, 320)
    w = w.detach().float().cpu()
    ar = torch.arange
    s_, o, l, e = ar(20).view(-1, 1, 1, 1), ar(10).view(1, -1, 1, 1), ar(64).view(1, 1, -1, 1), ar(8).view(1, 1, 1, -1)
    img = w[32 * o + (l & 31), 16 * s_ + 8 * (l >> 5) + e].to(ops.ELEM).contiguous()
    return img.view(-1).view(torch.uint8)


def pack_rowproj320(w):
    """Linear(320, N) weight [N, 320] (N % 64 == 0) -> the packed image of svd_rowproj320 (csrc/rowproj.hip), a uint8 tensor: per chunk of 64 output channels 40 MFMA
    B-operand fragments of 1 KiB, fragment 2 s + t (k-step s of 20, column tile t of 2): lane l holds W[64 ch + 2 (l % 32) + t][16 s + 8 (l // 32) .. + 7] -- the
    chunk's channels interleaved over its two column tiles, so that a lane's two accumulators are ADJACENT channels (one dword store)."""
    N = w.shape[0]
    assert w.shape[1] == 320 and N % 64 == 0
    w = w.detach().float().cpu()
    ar = torch.arange
    ch, s_, t, l, e = (ar(N // 64).view(-1, 1, 1, 1, 1), ar(20).view(1, -1, 1, 1, 1), ar(2).view(1, 1, -1, 1, 1), ar(64).view(1, 1, 1, -1, 1),
                       ar(8).view(1, 1, 1, 1, -1))
    img = w[64 * ch + 2 * (l & 31) + t, 16 * s_ + 8 * (l >> 5) + e].to(ops.ELEM).contiguous()
    return img.view(-1).view(torch.uint8)


class RowProj:
    """A 320 -> 320 projection of the fp32 residual stream, optionally with the LayerNorm that consumes its result: ONE launch of svd_rowgemm320 where that
    kernel applies (dim 320, fp32 stream, the per-frame vector constant inside a 32-row tile), else svd_gemm (+ svd_layernorm).
    __call__(x16, bias=, rowvec=, rows_per_vec=, residual=, ln=(gamma, beta), stream=) -> (y, LayerNorm(y) or None)."""

    def __init__(self, w, dev):
        self.w = _dev_bf16(w, dev)
        self.img = pack_rowgemm320(w).to(dev) if tuple(w.shape) == (320, 320) else None
        self.dtype = ops.ELEM

    def __call__(self, x, bias=None, rowvec=None, rows_per_vec=0, residual=None, ln=None, stream=True):
        fused = (stream and self.img is not None and x.dtype == self.dtype and (residual is None or residual.dtype == torch.float32)
                 and ops.rowgemm_ok(x, self.img, rows_per_vec) and (ln is not None or x.shape[0] >= ops.ROWGEMM_PLAIN_MIN_ROWS))
        if fused:
            return ops.rowgemm320(x, self.img, bias=bias, rowvec=rowvec, rows_per_vec=rows_per_vec, residual=residual, ln=ln)
        y = ops.gemm(x, self.w, bias=bias, rowvec=rowvec, rows_per_vec=rows_per_vec, residual=residual, out_f32=stream)
        return y, (ops.layernorm(y, *ln) if ln is not None else None)


class WideProj:
    """Fused q | k (| v) projection Linear(C, n C) without bias: svd_rowproj320 where that kernel applies (C = 320, enough rows), else svd_gemm.  __call__(x16) -> [M, n C]."""

    def __init__(self, w, dev):
        self.w = _dev_bf16(w, dev)
        self.n_out = w.shape[0]
        self.img = pack_rowproj320(w).to(dev) if (w.shape[1] == 320 and w.shape[0] % 64 == 0) else None
        self.dtype = ops.ELEM

    def __call__(self, x):
        if self.img is not None and x.dtype == self.dtype and ops.rowproj_ok(x, self.img):
            return ops.rowproj320(x, self.img, self.n_out)
        return ops.gemm(x, self.w)


class FeedForward:
    """FeedForward(dim, mult 4, glu=True) = GEGLU.proj -> value * gelu(gate) -> net[2] (attention.py:94-120; diffusers FeedForward "geglu" in the
    enhancer) on the kernel path: ONE launch of svd_ff_geglu_fused where the fused kernel exists (dim 320: the level-0 blocks, whose [M, 1280]
    hidden tensor is the largest byte-mover of a forward), else the GEGLU-epilogue GEMM followed by the down-projection GEMM.
    __call__(x_normed, residual=, out_f32=, blend=) -> residual + ff(x) [blended]."""

    def __init__(self, get, prefix, dev):
        w1r, b1r, w2r = get(prefix + "net.0.proj.weight"), get(prefix + "net.0.proj.bias"), get(prefix + "net.2.weight")
        w1, b1 = pack_geglu(w1r, b1r)
        self.w1, self.b1 = _dev_bf16(w1, dev), _dev_f32(b1, dev)
        self.w2, self.b2 = _dev_bf16(w2r, dev), _dev_f32(get(prefix + "net.2.bias"), dev)
        self.c, self.hidden = w2r.shape
        self.img = pack_ff_fused(w1r, b1r, w2r).to(dev) if ops.ff_fused_ok(self.c, self.hidden) else None
        self.img_dtype = ops.ELEM          # the packed image is an opaque byte blob in THIS element type: the kernel must be launched with the same one

    def __call__(self, x, residual=None, out_f32=False, blend=None, rowvec=None, rows_per_vec=0, ln=None, ln_addvec=None, ln_rows_per_vec=0):
        """rowvec / rows_per_vec: a per-frame vector added to the result (residual + rowvec + ff(x)): `x + time_pos_embed` without materialising the sum.
        ln = (gamma, beta): the LayerNorm that consumes the result, over result + ln_addvec[row // ln_rows_per_vec] -> returns (y, LayerNorm(...)); the fused
        kernel emits it from its accumulators where it can (fp32 stream), else svd_layernorm follows."""
        fused = self.img is not None and ops.FF_FUSED and (rowvec is None or rows_per_vec % 32 == 0)
        if fused:
            if x.dtype != self.img_dtype:
                raise TypeError(f"fused feed-forward weights were packed as {self.img_dtype} (ops.ELEM at load_state_dict) but the activations are {x.dtype}: "
                                f"call ops.set_element_dtype BEFORE load_state_dict")
            if (ln is not None and ops.FF_FUSED_LN and out_f32 and blend is None and residual is not None and residual.dtype == torch.float32
                    and (ln_addvec is None or ln_rows_per_vec % 32 == 0)):
                return ops.ff_geglu_fused(x, self.img, self.hidden, self.b2, residual=residual, out_f32=True, rowvec=rowvec, rows_per_vec=rows_per_vec,
                                          ln=ln, ln_addvec=ln_addvec, ln_rows_per_vec=ln_rows_per_vec)
            y = ops.ff_geglu_fused(x, self.img, self.hidden, self.b2, residual=residual, blend=blend, out_f32=out_f32, rowvec=rowvec, rows_per_vec=rows_per_vec)
        else:
            g = ops.gemm(x, self.w1, bias=self.b1, geglu=True)
            y = ops.gemm(g, self.w2, bias=self.b2, rowvec=rowvec, rows_per_vec=rows_per_vec, residual=residual, blend=blend, out_f32=out_f32)
        if ln is None:
            return y
        return y, ops.layernorm(y, *ln, addvec=ln_addvec, rows_per_vec=ln_rows_per_vec)


def pack_x3(w2d, taps):
    """fp32 GEMM weight [N, taps * cin] (K order (tap, c)) -> the SPLIT-3 weight [N, taps * 3 * cin] in the element type: per tap
    [W_hi | W_hi | W_lo] with W_hi = rn16(W), W_lo = rn16(W - W_hi), matching activation rows [A_hi | A_lo | A_hi] (csrc/precision.hip)."""
    N, K = w2d.shape
    cin = K // taps
    w = w2d.detach().float().reshape(N, taps, cin)
    hi = w.to(ops.ELEM)
    lo = (w - hi.float()).to(ops.ELEM)
    return torch.cat([hi, hi, lo], 2).reshape(N, taps * 3 * cin).contiguous()


def pad_rows(w, n):
    if w.shape[0] >= n:
        return w
    out = torch.zeros((n,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[: w.shape[0]] = w
    return out


def _sigmoid(x):
    return 1.0 / (1.0 + math.exp(-float(x)))


def _gn_pooled(x, frames, pix, w, b, eps, frames_per_stat, count, sp, silu):
    """GroupNorm whose statistics pool over rows that are sharded over the ranks of `sp` (the 5-D norms of time_stack and of the CAM
    merger): local (sum, sum of squares) -> all-reduce -> statistics with the GLOBAL element count -> apply on the local rows."""
    sums = sp.allreduce_sums(ops.groupnorm_sums(x, frames, pix, frames_per_stat))
    return ops.groupnorm_apply_sums(x, frames, pix, w, b, eps, sums, count, frames_per_stat=frames_per_stat, silu=silu)


# ----------------------------------------------------------------------------------------------------------------
class VideoResBlock:
    """ResBlock (2-D) -> time_stack ResBlock (3,1,1) -> AlphaBlender.  video_model.py:66-85, openaimodel.py:328-354."""

    def __init__(self, prefix, cin, cout, emb_ch):
        self.p, self.cin, self.cout, self.emb_ch = prefix, cin, cout, emb_ch

    def spec(self, s):
        p, ci, co, e = self.p, self.cin, self.cout, self.emb_ch
        s.add(p + "in_layers.0.weight", ci); s.add(p + "in_layers.0.bias", ci)
        s.add(p + "in_layers.2.weight", co, ci, 3, 3); s.add(p + "in_layers.2.bias", co)
        s.add(p + "emb_layers.1.weight", co, e); s.add(p + "emb_layers.1.bias", co)
        s.add(p + "out_layers.0.weight", co); s.add(p + "out_layers.0.bias", co)
        s.add(p + "out_layers.3.weight", co, co, 3, 3); s.add(p + "out_layers.3.bias", co)
        if ci != co:
            s.add(p + "skip_connection.weight", co, ci, 1, 1); s.add(p + "skip_connection.bias", co)
        t = p + "time_stack."
        s.add(t + "in_layers.0.weight", co); s.add(t + "in_layers.0.bias", co)
        s.add(t + "in_layers.2.weight", co, co, 3, 1, 1); s.add(t + "in_layers.2.bias", co)
        s.add(t + "emb_layers.1.weight", co, e); s.add(t + "emb_layers.1.bias", co)
        s.add(t + "out_layers.0.weight", co); s.add(t + "out_layers.0.bias", co)
        s.add(t + "out_layers.3.weight", co, co, 3, 1, 1); s.add(t + "out_layers.3.bias", co)
        s.add(p + "time_mixer.mix_factor", 1)

    def prepare(self, sd, dev):
        p = self.p
        g = lambda k: sd[p + k]
        self.n1w, self.n1b = _dev_f32(g("in_layers.0.weight"), dev), _dev_f32(g("in_layers.0.bias"), dev)
        self.w1, self.b1 = _dev_bf16(pack_conv3x3(g("in_layers.2.weight")), dev), _dev_f32(g("in_layers.2.bias"), dev)
        self.we, self.be = _dev_bf16(g("emb_layers.1.weight"), dev), _dev_f32(g("emb_layers.1.bias"), dev)
        self.n2w, self.n2b = _dev_f32(g("out_layers.0.weight"), dev), _dev_f32(g("out_layers.0.bias"), dev)
        self.w2, self.b2 = _dev_bf16(pack_conv3x3(g("out_layers.3.weight")), dev), _dev_f32(g("out_layers.3.bias"), dev)
        if self.cin != self.cout:
            self.ws = _dev_bf16(g("skip_connection.weight")[:, :, 0, 0], dev)
            self.bs = _dev_f32(g("skip_connection.bias"), dev)
        t = "time_stack."
        self.tn1w, self.tn1b = _dev_f32(g(t + "in_layers.0.weight"), dev), _dev_f32(g(t + "in_layers.0.bias"), dev)
        self.tw1, self.tb1 = _dev_bf16(pack_tconv3(g(t + "in_layers.2.weight")), dev), _dev_f32(g(t + "in_layers.2.bias"), dev)
        self.twe, self.tbe = _dev_bf16(g(t + "emb_layers.1.weight"), dev), _dev_f32(g(t + "emb_layers.1.bias"), dev)
        self.tn2w, self.tn2b = _dev_f32(g(t + "out_layers.0.weight"), dev), _dev_f32(g(t + "out_layers.0.bias"), dev)
        self.alpha = _sigmoid(g("time_mixer.mix_factor"))   # image_only_indicator == 0 (util.py:341-357)
        # AlphaBlender of the block: alpha * x_spatial + (1 - alpha) * x_temporal with x_temporal = x_spatial + conv(...) + bias (the time_stack is a
        # ResBlock with an identity skip, video_model.py:75-85) = x_spatial + (1 - alpha) * (conv(...) + bias): the blend weight is folded into the
        # last temporal convolution, which then is a plain "+ residual" GEMM (specialised epilogue, residual prefetched 3 passes ahead) instead of
        # the generic blend pass with a second full-size operand.  (1 - alpha) * W is rounded to 16 bit once, like W itself.
        self.tw2 = _dev_bf16(pack_tconv3(g(t + "out_layers.3.weight")) * (1.0 - self.alpha), dev)
        self.tb2 = _dev_f32(g(t + "out_layers.3.bias").detach().float() * (1.0 - self.alpha), dev)

    def forward(self, x, emb_silu, F, T, H, W, sp=None, emb_full=None, emb_out=None, out16=None):
        """x [F*H*W, C]: the frames this rank holds (all B*T of them without sequence parallelism; then emb_silu is also the
        embedding of all frames).  With `sp` (parallel.SeqParallel): emb_silu = embedding rows of the LOCAL frames (2-D part),
        emb_full = rows of all B*T frames (the time_stack runs in the pixel layout, where every rank sees all frames).
        emb_out = (rows of the local frames, rows of all frames) of the network's PACKED emb_layers output (_EncoderBase._pack_emb_layers):
        this block's two Linear(emb -> cout) results are column slices of it (without it the block applies its own two Linear layers)."""
        pix = H * W
        e_pre = et_pre = None
        if emb_out is not None:
            e_pre = emb_out[0][:, self.e_off:self.e_off + self.cout]
            et_pre = (emb_out[0] if sp is None else emb_out[1])[:, self.et_off:self.et_off + self.cout]
        cv_in = dict(cin=self.cin, hin=H, win=W, hout=H, wout=W, frames=F)
        cv = dict(cin=self.cout, hin=H, win=W, hout=H, wout=W, frames=F)
        st = ops.stream_on(self.cout, "res")   # the block's intermediate sum / output are the residual stream: fp32 between kernels when set (its input is whatever the block before wrote)
        h = ops.groupnorm(x, F, pix, self.n1w, self.n1b, 1e-5, silu=True)
        e = e_pre if e_pre is not None else ops.gemm(emb_silu, self.we, bias=self.be, out_f32=True)
        h = ops.gemm(h, self.w1, bias=self.b1, rowvec=e, rows_per_vec=pix, conv=cv_in)
        h = ops.groupnorm(h, F, pix, self.n2w, self.n2b, 1e-5, silu=True)
        skip = x if self.cin == self.cout else ops.gemm(ops.to_elem_rows(x), self.ws, bias=self.bs, out_f32=st)
        hs = ops.gemm(h, self.w2, bias=self.b2, residual=skip, conv=cv, out_f32=st)
        if sp is None:
            # time_stack: 5-D GroupNorm statistics pool over the T frames of a batch element (video_model.py:75-80)
            tv = dict(cin=self.cout, T=T, pix=pix)
            g = ops.groupnorm(hs, F, pix, self.tn1w, self.tn1b, 1e-5, frames_per_stat=T, silu=True)
            et = et_pre if et_pre is not None else ops.gemm(emb_silu, self.twe, bias=self.tbe, out_f32=True)
            g = ops.gemm(g, self.tw1, bias=self.tb1, rowvec=et, rows_per_vec=pix, temporal=tv)
            g = ops.groupnorm(g, F, pix, self.tn2w, self.tn2b, 1e-5, frames_per_stat=T, silu=True)
            # out = alpha * x_spatial + (1 - alpha) * (conv + bias + identity skip) = x_spatial + [(1 - alpha) conv + (1 - alpha) bias]  (prepare)
            if out16 is not None and ops.ZERO_COPY_GENERIC:          # the block's output feeds a concatenation only: its 16-bit rounding, in place
                return ops.gemm(g, self.tw2, bias=self.tb2, residual=hs, temporal=tv, out=out16)
            return ops.gemm(g, self.tw2, bias=self.tb2, residual=hs, temporal=tv, out_f32=st)
        # sequence parallel: the whole time_stack in the PIXEL layout (all T frames of this rank's pixel range); its two norms pool
        # over every frame and pixel -> all-reduce of the sums
        B = (emb_out[1] if emb_out is not None else emb_full).shape[0] // T
        pl = sp.pix_local(pix)
        hp = sp.to_pixels(hs, B, T, pix)
        tv = dict(cin=self.cout, T=T, pix=pl)
        cnt = float(T) * pix * (self.cout // 32)
        g = _gn_pooled(hp, B * T, pl, self.tn1w, self.tn1b, 1e-5, T, cnt, sp, True)
        et = et_pre if et_pre is not None else ops.gemm(emb_full, self.twe, bias=self.tbe, out_f32=True)
        g = ops.gemm(g, self.tw1, bias=self.tb1, rowvec=et, rows_per_vec=pl, temporal=tv)
        g = _gn_pooled(g, B * T, pl, self.tn2w, self.tn2b, 1e-5, T, cnt, sp, True)
        out = ops.gemm(g, self.tw2, bias=self.tb2, residual=hp, temporal=tv, out_f32=st)
        return sp.to_frames(out, B, T, pix)


# ----------------------------------------------------------------------------------------------------------------
class _Attn1:
    """Parameters of a self-attention (to_q/to_k/to_v no bias, to_out.0 with bias)."""

    @staticmethod
    def spec(s, p, c):
        for n in ("to_q", "to_k", "to_v"):
            s.add(p + n + ".weight", c, c)
        s.add(p + "to_out.0.weight", c, c); s.add(p + "to_out.0.bias", c)


def _spec_attn2(s, p, c, ctx):
    s.add(p + "to_q.weight", c, c)
    s.add(p + "to_k.weight", c, ctx); s.add(p + "to_v.weight", c, ctx)
    s.add(p + "to_out.0.weight", c, c); s.add(p + "to_out.0.bias", c)


def _spec_ln(s, p, c):
    s.add(p + ".weight", c); s.add(p + ".bias", c)


def _spec_ff(s, p, c):
    s.add(p + "net.0.proj.weight", 8 * c, c); s.add(p + "net.0.proj.bias", 8 * c)
    s.add(p + "net.2.weight", c, 4 * c); s.add(p + "net.2.bias", c)


class SpatialVideoTransformer:
    """video_attention.py:174-333 with depth 1, use_linear, ff_in, use_spatial_context (config.yaml:99-113)."""

    def __init__(self, prefix, ch, ctx_dim, use_apm=False, apm_tokens=17):
        assert ch % 64 == 0
        self.p, self.c, self.ctx, self.heads = prefix, ch, ctx_dim, ch // 64
        self.use_apm, self.apm_tokens = use_apm, apm_tokens
        self._vt = {}
        self._temb = {}
        self._a2c = None          # per-chunk constants of the one-token cross-attentions (_attn2_const)

    def spec(self, s):
        p, c, ctx = self.p, self.c, self.ctx
        _spec_ln(s, p + "norm", c)
        s.add(p + "proj_in.weight", c, c); s.add(p + "proj_in.bias", c)
        b = p + "transformer_blocks.0."
        _Attn1.spec(s, b + "attn1.", c)
        _spec_ff(s, b + "ff.", c)
        _spec_attn2(s, b + "attn2.", c, ctx)
        for n in ("norm1", "norm2", "norm3"):
            _spec_ln(s, b + n, c)
        if self.use_apm:        # BasicTransformerBlockWithAPM (attention.py:596-611); registration order: after the parent's modules
            s.add(b + "apm_alpha")
            s.add(b + "apm_conv.weight", 1, self.apm_tokens, 3); s.add(b + "apm_conv.bias", 1)
            _spec_ln(s, b + "apm_ln", ctx)
        s.add(p + "proj_out.weight", c, c); s.add(p + "proj_out.bias", c)
        t = p + "time_stack.0."
        _spec_ln(s, t + "norm_in", c)
        _spec_ff(s, t + "ff_in.", c)
        _Attn1.spec(s, t + "attn1.", c)
        _spec_ff(s, t + "ff.", c)
        _spec_ln(s, t + "norm2", c)
        _spec_attn2(s, t + "attn2.", c, ctx)
        _spec_ln(s, t + "norm1", c)
        _spec_ln(s, t + "norm3", c)
        s.add(p + "time_pos_embed.0.weight", 4 * c, c); s.add(p + "time_pos_embed.0.bias", 4 * c)
        s.add(p + "time_pos_embed.2.weight", c, 4 * c); s.add(p + "time_pos_embed.2.bias", c)
        s.add(p + "time_mixer.mix_factor", 1)

    def prepare(self, sd, dev):
        self._a2c = None
        p = self.p
        g = lambda k: sd[p + k]
        W = lambda k: _dev_bf16(g(k), dev)
        Fv = lambda k: _dev_f32(g(k), dev)
        self.dev = dev
        self.nw, self.nb = Fv("norm.weight"), Fv("norm.bias")
        self.wpi, self.bpi = W("proj_in.weight"), Fv("proj_in.bias")
        self.wpo, self.bpo = W("proj_out.weight"), Fv("proj_out.bias")
        b = "transformer_blocks.0."
        self.s_ln = {n: (Fv(b + n + ".weight"), Fv(b + n + ".bias")) for n in ("norm1", "norm3")}
        self.s_wqk = _dev_bf16(torch.cat([g(b + "attn1.to_q.weight"), g(b + "attn1.to_k.weight")], 0), dev)
        self.s_wv = W(b + "attn1.to_v.weight")
        self.s_wo, self.s_bo = W(b + "attn1.to_out.0.weight"), Fv(b + "attn1.to_out.0.bias")
        self.s_wv2, self.s_wo2, self.s_bo2 = W(b + "attn2.to_v.weight"), W(b + "attn2.to_out.0.weight"), Fv(b + "attn2.to_out.0.bias")
        self.s_ff = FeedForward(g, b + "ff.", dev)
        t = "time_stack.0."
        self.t_ln = {n: (Fv(t + n + ".weight"), Fv(t + n + ".bias")) for n in ("norm_in", "norm1", "norm3")}
        self.t_ff_in = FeedForward(g, t + "ff_in.", dev)
        self.t_wqkv = _dev_bf16(torch.cat([g(t + "attn1.to_q.weight"), g(t + "attn1.to_k.weight"), g(t + "attn1.to_v.weight")], 0), dev)
        self.t_wo, self.t_bo = W(t + "attn1.to_out.0.weight"), Fv(t + "attn1.to_out.0.bias")
        self.t_wv2, self.t_wo2, self.t_bo2 = W(t + "attn2.to_v.weight"), W(t + "attn2.to_out.0.weight"), Fv(t + "attn2.to_out.0.bias")
        self.t_ff = FeedForward(g, t + "ff.", dev)
        # round 6: the 320 -> 320 projections of the 320-channel blocks in the row-owning kernel's fragment order (ops.rowgemm320, csrc/rowgemm.hip)
        self.rg = None
        if self.c == 320:
            R = lambda k: pack_rowgemm320(g(k)).to(dev)
            self.rg = dict(pi=R("proj_in.weight"), so=R(b + "attn1.to_out.0.weight"), to=R(t + "attn1.to_out.0.weight"), po=R("proj_out.weight"), dtype=ops.ELEM)
        # ... and the q | k / q | k | v projections in the row-resident kernel's (ops.rowproj320, csrc/rowproj.hip): the same concatenated weights as s_wqk / t_wqkv
        self.rp = None
        if self.c == 320:
            cat = lambda *ks: torch.cat([g(k) for k in ks], 0)
            self.rp = dict(sqk=pack_rowproj320(cat(b + "attn1.to_q.weight", b + "attn1.to_k.weight")).to(dev),
                           tqkv=pack_rowproj320(cat(t + "attn1.to_q.weight", t + "attn1.to_k.weight", t + "attn1.to_v.weight")).to(dev), dtype=ops.ELEM)
        self.tp_w0, self.tp_b0 = W("time_pos_embed.0.weight"), Fv("time_pos_embed.0.bias")
        self.tp_w2, self.tp_b2 = W("time_pos_embed.2.weight"), Fv("time_pos_embed.2.bias")
        self.alpha = _sigmoid(g("time_mixer.mix_factor"))
        if self.use_apm:
            # appearance-preservation front of the spatial block (attention.py:613-619): Conv1d(17 -> 1, k 3, "same") along the 1024-wide
            # CLIP axis as a [*, 3*17 -> padded 64] GEMM, LayerNorm with the gate silu(alpha) folded into its affine parameters
            nt = self.apm_tokens
            wc = g(b + "apm_conv.weight").detach().float()[0]                            # [17, 3]
            wk = torch.zeros(8, 64)
            wk[0, : 3 * nt] = wc.t().reshape(-1)                                          # K order (tap, token)
            self.apm_w = _dev_bf16(wk, dev)
            self.apm_b = _dev_f32(torch.cat([g(b + "apm_conv.bias").detach().float().reshape(1), torch.zeros(7)]), dev)
            a = float(g(b + "apm_alpha"))
            gate = a / (1.0 + math.exp(-a))                                               # F.silu(alpha)
            self.apm_lnw, self.apm_lnb = _dev_f32(g(b + "apm_ln.weight").detach().float() * gate, dev), _dev_f32(g(b + "apm_ln.bias").detach().float() * gate, dev)
            # the temporal block cross-attends to all tokens of the first frame's context (video_attention.py:281-285): real attention
            self.t_ln["norm2"] = (Fv(t + "norm2.weight"), Fv(t + "norm2.bias"))
            self.t_wq2 = W(t + "attn2.to_q.weight")
            self.t_wk2 = W(t + "attn2.to_k.weight")
            self._vt2 = {}

    def _apm_context(self, ctx_tokens):
        """[F, n_tok, 1024] fp32 -> the spatial block's effective 1-token context [F, 1024]: ctx[:, 0] + LN(conv1d(ctx)) * silu(alpha)."""
        Fn, nt, L = ctx_tokens.shape
        assert nt == self.apm_tokens
        xp = torch.nn.functional.pad(ctx_tokens.float(), (1, 1))                          # "same" padding along the CLIP axis
        cols = torch.stack([xp[:, :, k:k + L] for k in range(3)], 1)                      # [F, 3, nt, L]: im2col (data movement only)
        a = torch.zeros((Fn * L, 64), dtype=torch.float32, device=ctx_tokens.device)
        a[:, : 3 * nt] = cols.reshape(Fn, 3 * nt, L).transpose(1, 2).reshape(Fn * L, 3 * nt)
        mixed = ops.gemm(ops.to_bf16(a), self.apm_w, bias=self.apm_b, out_f32=True)[:, 0].reshape(Fn, L).contiguous()
        ln = ops.layernorm(ops.to_bf16(mixed), self.apm_lnw, self.apm_lnb)
        return ops.add_rows(ln, ops.to_bf16(ctx_tokens[:, 0].float().contiguous()))

    def _vt_buf(self, F, pix, x_dtype):
        """V^T staging buffer [F, C, tok_ld]; the pad beyond `pix` stays zero (never written by the GEMM)."""
        tok_ld = (pix + 63) // 64 * 64
        key = (F, tok_ld, x_dtype)
        b = self._vt.get(key)
        if b is None:
            b = torch.zeros((F, self.c, tok_ld), dtype=x_dtype, device=self.dev)
            self._vt[key] = b
        return b, tok_ld

    def _time_emb(self, F, T):
        """time_pos_embed(timestep_embedding(arange(T))) repeated per batch element -> [F, C] fp32
        (video_attention.py:298-308); input independent, cached."""
        key = (F, T)
        e = self._temb.get(key)
        if e is None:
            idx = torch.arange(T, device=self.dev, dtype=torch.float32)
            te = ops.timestep_embedding(idx, self.c)
            h = ops.gemm(te, self.tp_w0, bias=self.tp_b0, silu=True)
            e1 = ops.gemm(h, self.tp_w2, bias=self.tp_b2, out_f32=True)   # [T, C]
            e = e1.repeat(F // T, 1).contiguous()
            self._temb[key] = e
        return e

    def _attn2_const(self, ctx, tctx):
        """With ONE context token the cross-attention (attn2) of both transformer blocks is softmax over a single key == 1: its output is
        to_out(to_v(context)), one vector per frame (spatial) / per video (temporal), independent of the hidden state (K3 shortcut).  The
        context is the same tensor for every Euler step of a chunk (sampling.py builds the CFG-concatenated conditioning once per chunk and
        _EncoderBase._local_conditioning maps one input tensor to one ctx object), so the four tiny GEMMs run once per chunk, not once per
        step: keyed on the IDENTITY of the ctx / tctx objects (held here, so an address can never be reused for different values)."""
        c = self._a2c
        if c is None or c[0] is not ctx or c[1] is not tctx or c[2] != ops.ELEM:
            v2 = ops.gemm(ops.gemm(ctx, self.s_wv2), self.s_wo2, bias=self.s_bo2, out_f32=True)
            v2t = None if tctx is None else ops.gemm(ops.gemm(tctx, self.t_wv2), self.t_wo2, bias=self.t_bo2, out_f32=True)
            self._a2c = c = (ctx, tctx, ops.ELEM, v2, v2t)
        return c[3], c[4]

    def forward(self, x, ctx, tctx, F, T, H, W, sp=None, out16=None):
        """x [F*H*W, C]; ctx [F, ctx_dim] bf16 (per-frame CLIP token); tctx [B, ctx_dim] (= context[::T]).  With `sp`
        (parallel.SeqParallel) x / ctx hold this rank's frames; the temporal block runs in the pixel layout.
        out16: a 16-bit [rows, C] view (a column range of the next block's concatenation buffer, ops.ZERO_COPY_CONCAT): when the block's output is consumed by that
        concatenation only, proj_out writes its rounding there directly where the row-owning kernel applies; the view is returned then, else the usual tensor."""
        c, heads, pix = self.c, self.heads, H * W
        M, B = F * pix, tctx.shape[0]
        st = ops.stream_on(c, "svt")  # fp32 residual stream: h, xm and the output are fp32 between the kernels; every GEMM / attention operand is 16 bit
        e16 = ops.ELEM if x.dtype == torch.float32 else x.dtype
        tctx_tokens = None
        if ctx.dim() == 3:                     # APM: [F, 17, 1024] tokens (fp32); tctx [B, 17, 1024]
            if not self.use_apm:
                raise NotImplementedError("cross-attention contexts with > 1 token need use_apm (config.yaml:115 ships use_apm: false)")
            tctx_tokens, ctx = tctx, self._apm_context(ctx)
        h = ops.groupnorm(x, F, pix, self.nw, self.nb, 1e-6, silu=False)
        # round 6: with the fp32 stream on, a 320-channel block's proj_in / to_out / proj_out run in the row-owning kernel, which also emits the LayerNorm
        # of its result (norm1 / norm3) -- one launch instead of svd_gemm + svd_layernorm, the fp32 tensor is not read back
        rg = self.rg if (st and self.rg is not None and x.dtype == torch.float32 and h.dtype == self.rg["dtype"] and ops.rowgemm_ok(h, self.rg["pi"], pix)) else None
        # ---- spatial BasicTransformerBlock (attention.py:567-593) ----
        if rg is not None:
            h, n1 = ops.rowgemm320(h, rg["pi"], bias=self.bpi, ln=self.s_ln["norm1"])
        else:
            h = ops.gemm(h, self.wpi, bias=self.bpi, out_f32=st)
            n1 = ops.layernorm(h, *self.s_ln["norm1"])
        rp = self.rp if (self.rp is not None and n1.dtype == self.rp["dtype"]) else None
        qk = ops.rowproj320(n1, rp["sqk"], 2 * c) if (rp is not None and ops.rowproj_ok(n1, rp["sqk"])) else ops.gemm(n1, self.s_wqk)
        vt, tok_ld = self._vt_buf(F, pix, e16)
        ops.gemm(n1, self.s_wv, trans_out=dict(tok_per_frame=pix, tokens_ld=tok_ld, out=vt))
        a = torch.empty((M, c), dtype=e16, device=x.device)
        ops.attn_spatial(qk[:, :c], qk[:, c:], vt, a, F, pix, heads)
        v2, v2t_c = self._attn2_const(ctx, tctx if tctx_tokens is None else None)                    # attn2 == const/frame
        if rg is not None:
            h, n3 = ops.rowgemm320(a, rg["so"], bias=self.s_bo, rowvec=v2, rows_per_vec=pix, residual=h, ln=self.s_ln["norm3"])
        else:
            h = ops.gemm(a, self.s_wo, bias=self.s_bo, rowvec=v2, rows_per_vec=pix, residual=h, out_f32=st)
            n3 = ops.layernorm(h, *self.s_ln["norm3"])
        # round 6: the two LayerNorms that consume a feed-forward's result (norm_in over x_spatial + time_pos_embed, norm1 behind ff_in) come out of the
        # fused feed-forward's own epilogue where the fp32 stream runs; with `sp` the rows change layout in between and norm_in stays a kernel of its own
        temb = self._time_emb(B * T, T)
        nin = None
        if sp is None:
            h, nin = self.s_ff(n3, residual=h, out_f32=st, ln=self.t_ln["norm_in"], ln_addvec=temb, ln_rows_per_vec=pix)             # x_spatial
        else:
            h = self.s_ff(n3, residual=h, out_f32=st)
        # ---- temporal VideoTransformerBlock on the same token layout (video_attention.py:125-168) ----
        # rows (b, t, pixel) with pt pixels per frame: all of them, or this rank's pixel range of ALL T frames (one all-to-all in)
        ht, pt = (h, pix) if sp is None else (sp.to_pixels(h, B, T, pix), sp.pix_local(pix))
        # x_mix = x + time_pos_embed (video_attention.py:318-321) is never written: norm_in normalises the sum on the fly and ff_in takes x and the per-frame
        # embedding as residual + row vector (round 6: one fp32 tensor less written and read per block)
        if nin is None:
            nin = ops.layernorm(ht, *self.t_ln["norm_in"], addvec=temb, rows_per_vec=pt)
        xm, n1 = self.t_ff_in(nin, residual=ht, out_f32=st, rowvec=temb, rows_per_vec=pt, ln=self.t_ln["norm1"])
        qkv = ops.rowproj320(n1, rp["tqkv"], 3 * c) if (rp is not None and ops.rowproj_ok(n1, rp["tqkv"])) else ops.gemm(n1, self.t_wqkv)
        at = torch.empty((B * T * pt, c), dtype=e16, device=x.device)
        ops.attn_temporal(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], at, B, T, T, pt, heads)
        n3 = None
        if tctx_tokens is None:
            v2t = v2t_c                                                                               # [B, C]
            if rg is not None and (T * pt) % 32 == 0:
                xm, n3 = ops.rowgemm320(at, rg["to"], bias=self.t_bo, rowvec=v2t, rows_per_vec=T * pt, residual=xm, ln=self.t_ln["norm3"])
            else:
                xm = ops.gemm(at, self.t_wo, bias=self.t_bo, rowvec=v2t, rows_per_vec=T * pt, residual=xm, out_f32=st)
        else:
            # APM: attn2 of the temporal block is a real cross-attention of every (frame, pixel) token to the n_tok context tokens of its
            # batch element (video_attention.py:150-154 with time_context = context[::T] repeated over pixels)
            xm = ops.gemm(at, self.t_wo, bias=self.t_bo, residual=xm, out_f32=st)
            nt = tctx_tokens.shape[1]
            tk = ops.to_bf16(tctx_tokens.float().reshape(B * nt, -1).contiguous())
            q2 = ops.gemm(ops.layernorm(xm, *self.t_ln["norm2"]), self.t_wq2)
            k2 = ops.gemm(tk, self.t_wk2)
            key = (B, e16)
            vt2 = self._vt2.get(key)
            if vt2 is None:
                vt2 = self._vt2[key] = torch.zeros((B, c, 64), dtype=e16, device=x.device)
            # V^T [B, C, 64]: 17 tokens per batch element are not a multiple of the 4-token store width of the GEMM's transposed-output
            # mode, so the (tiny: B x 17 x C) transpose is a copy here
            vt2[:, :, :nt] = ops.gemm(tk, self.t_wv2).view(B, nt, c).transpose(1, 2)
            a2 = torch.empty((B * T * pt, c), dtype=e16, device=x.device)
            ops.attn_cross(q2, k2, vt2, a2, B, T * pt, nt, 1, heads)
            xm = ops.gemm(a2, self.t_wo2, bias=self.t_bo2, residual=xm, out_f32=st)
        if n3 is None:
            n3 = ops.layernorm(xm, *self.t_ln["norm3"])
        xb = self.t_ff(n3, residual=xm, blend=(self.alpha, ht))                                  # AlphaBlender
        if sp is not None:
            xb = sp.to_frames(xb, B, T, pix)                                                    # one all-to-all out
        # xb (the blend) is consumed by proj_out only: a GEMM operand, 16 bit; proj_out + x continues the stream
        out32 = st or (ops.STREAM_F32_SVT_IO_MIN_CH > 0 and c >= ops.STREAM_F32_SVT_IO_MIN_CH)      # (A/B) block output alone in fp32
        # proj_out has no LayerNorm behind it; the row-owning kernel still moves its rows faster than the 256 x 320 tile on the large token matrices
        # (307 vs 353 us at M = 460 800, 667 vs 781 us at 1 094 400; 111 vs 107 us at 129 024: profiles/r06_rowgemm_probe_v2.txt)
        if rg is not None and M >= ops.ROWGEMM_PLAIN_MIN_ROWS:
            return ops.rowgemm320(xb, rg["po"], bias=self.bpo, residual=x, out=out16)[0]
        if out16 is not None and ops.ZERO_COPY_GENERIC and sp is None:
            return ops.gemm(xb, self.wpo, bias=self.bpo, residual=x, out=out16)
        return ops.gemm(xb, self.wpo, bias=self.bpo, residual=x, out_f32=out32)


# ----------------------------------------------------------------------------------------------------------------
class ConditionalModel:
    """CAM merger: per-pixel temporal cross-attention of T sample frames to Tc ControlNet frames.
    conditioning.py:39-81,117-146 (+ diffusers Attention: to_q/k/v no bias, to_out.0 bias, heads = C/64)."""

    def __init__(self, prefix, ch):
        self.p, self.c, self.heads = prefix + "temporal_transformer.", ch, ch // 64

    def spec(self, s):
        p, c = self.p, self.c
        _Attn1.spec(s, p + "attention.", c)
        _spec_ln(s, p + "norm", c)
        s.add(p + "proj_in.weight", c, c); s.add(p + "proj_in.bias", c)
        s.add(p + "proj_out.weight", c, c); s.add(p + "proj_out.bias", c)

    def prepare(self, sd, dev):
        p = self.p
        g = lambda k: sd[p + k]
        self.nw, self.nb = _dev_f32(g("norm.weight"), dev), _dev_f32(g("norm.bias"), dev)
        self.wpi, self.bpi = _dev_bf16(g("proj_in.weight"), dev), _dev_f32(g("proj_in.bias"), dev)
        self.wpo, self.bpo = _dev_bf16(g("proj_out.weight"), dev), _dev_f32(g("proj_out.bias"), dev)
        self.wq = _dev_bf16(g("attention.to_q.weight"), dev)
        self.wkv = _dev_bf16(torch.cat([g("attention.to_k.weight"), g("attention.to_v.weight")], 0), dev)
        self.wo, self.bo = _dev_bf16(g("attention.to_out.0.weight"), dev), _dev_f32(g("attention.to_out.0.bias"), dev)
        self.rg_po = pack_rowgemm320(g("proj_out.weight")).to(dev) if self.c == 320 else None          # proj_out + residual straight into a concatenation buffer (out16)
        self.rg_dtype = ops.ELEM

    def forward(self, sample, cond, F, T, Tc, H, W, sp=None, out16=None):
        """sample [F*pix, C] (F = B * frames held by this rank), cond [B*Tc(local)*pix, C] ControlNet features.  With `sp` the queries are
        this rank's frames, the 7 conditioning frames' K / V are all-gathered (they are sharded like the ControlNet that made them), and
        the 5-D GroupNorm sums are all-reduced."""
        c, pix = self.c, H * W
        cond = ops.to_elem_rows(cond)          # ControlNet features: a GEMM operand here (16-bit copy of the ControlNet's fp32 stream)
        e16 = cond.dtype
        if sp is None:
            B = F // T
            hn = ops.groupnorm(sample, F, pix, self.nw, self.nb, 1e-6, frames_per_stat=T, silu=False)
            Tq = T
        else:
            Tq = sp.frame_counts(T)[sp.rank]
            B = F // Tq
            hn = _gn_pooled(sample, F, pix, self.nw, self.nb, 1e-6, Tq, float(T) * pix * (c // 32), sp, False)
        hn = ops.gemm(hn, self.wpi, bias=self.bpi)
        q = ops.gemm(hn, self.wq)
        kv = ops.gemm(cond, self.wkv)
        if sp is not None:
            kv = sp.gather_frames(kv, B, Tc, pix)
        a = torch.empty((F * pix, c), dtype=e16, device=sample.device)
        ops.attn_temporal(q, kv[:, :c], kv[:, c:], a, B, Tq, Tc, pix, self.heads)
        a = ops.gemm(a, self.wo, bias=self.bo)
        # dropout(p=.25) on the non-conditional frames is identity in eval mode (conditioning.py:74-75)
        if (out16 is not None and self.rg_po is not None and sample.dtype == torch.float32 and a.dtype == self.rg_dtype and a.shape[0] >= ops.ROWGEMM_PLAIN_MIN_ROWS
                and ops.rowgemm_ok(a, self.rg_po)):
            return ops.rowgemm320(a, self.rg_po, bias=self.bpo, residual=sample, out=out16)[0]         # sample + proj_out(..), rounded once, in its consumer's buffer
        if out16 is not None and ops.ZERO_COPY_GENERIC:
            return ops.gemm(a, self.wpo, bias=self.bpo, residual=sample, out=out16)
        return ops.gemm(a, self.wpo, bias=self.bpo, residual=sample, out_f32=sample.dtype == torch.float32)


# ----------------------------------------------------------------------------------------------------------------
class _Conv:
    """3x3 conv (stem / Downsample.op / Upsample.conv / out) as implicit GEMM."""

    def __init__(self, prefix, cin, cout, stride=1, ups=0, x3=False):
        self.p, self.cin, self.cout, self.stride, self.ups = prefix, cin, cout, stride, ups
        self.cin_pad = cin if cin % 32 == 0 else (cin + 31) // 32 * 32
        self.cout_pad = cout if cout % 4 == 0 else (cout + 3) // 4 * 4
        self.x3 = x3              # a rim convolution: split-3 operands when the precision plan asks for them (ops.EXACT_RIM at prepare time)
        self.w3 = None

    def spec(self, s):
        s.add(self.p + "weight", self.cout, self.cin, 3, 3); s.add(self.p + "bias", self.cout)

    def prepare(self, sd, dev):
        wp = pack_conv3x3(sd[self.p + "weight"], self.cin_pad, self.cout_pad)
        self.w = _dev_bf16(wp, dev)
        self.b = _dev_f32(pad_rows(sd[self.p + "bias"].detach().float(), self.cout_pad), dev)
        self.w3 = pack_x3(wp, 9).to(dev) if (self.x3 and ops.EXACT_RIM) else None

    def forward(self, x, F, H, W, split3=False, **kw):
        """x: 16-bit (or fp32-stream) rows [F*H*W, cin_pad].  A rim convolution (w3) takes SPLIT-3 rows [F*H*W, 3*cin_pad] (ops.rows_split3 /
        ops.nchw_to_tokens_x3) when the caller SAYS so (split3=True: the layout is never inferred from the column count), or fp32 rows, which it
        splits itself."""
        if self.stride == 2:
            ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        elif self.ups:
            ho, wo = 2 * H, 2 * W
        else:
            ho, wo = H, W
        assert not split3 or (self.w3 is not None and x.shape[1] == 3 * self.cin_pad), \
            "split-3 rows handed to a convolution prepared without the rim weights (ops.EXACT_RIM at load time) or of the wrong width"
        if self.w3 is not None and (split3 or x.dtype == torch.float32):
            if not split3:
                assert x.shape[1] == self.cin_pad
                x = ops.rows_split3(x)
            cv = dict(cin=3 * self.cin_pad, hin=H, win=W, hout=ho, wout=wo, stride=self.stride, ups=self.ups, frames=F)
            return ops.gemm(x, self.w3, bias=self.b, conv=cv, **kw), ho, wo
        assert x.shape[1] == self.cin_pad, f"{self.p}: rows of {x.shape[1]} columns, expected {self.cin_pad}"
        cv = dict(cin=self.cin_pad, hin=H, win=W, hout=ho, wout=wo, stride=self.stride, ups=self.ups, frames=F)
        return ops.gemm(ops.to_elem_rows(x), self.w, bias=self.b, conv=cv, **kw), ho, wo


class _EmbedMLP:
    """Linear -> SiLU -> Linear (time_embed / label_emb.0)."""

    def __init__(self, prefix, cin, cmid, cout):
        self.p, self.cin, self.cmid, self.cout = prefix, cin, cmid, cout

    def spec(self, s):
        s.add(self.p + "0.weight", self.cmid, self.cin); s.add(self.p + "0.bias", self.cmid)
        s.add(self.p + "2.weight", self.cout, self.cmid); s.add(self.p + "2.bias", self.cout)

    def prepare(self, sd, dev):
        self.w0, self.b0 = _dev_bf16(sd[self.p + "0.weight"], dev), _dev_f32(sd[self.p + "0.bias"], dev)
        self.w2, self.b2 = _dev_bf16(sd[self.p + "2.weight"], dev), _dev_f32(sd[self.p + "2.bias"], dev)
        # precision plan (ops.EXACT_RIM): the two Linear layers with split-3 operands (a handful of rows per forward: free)
        self.w0_3 = self.w2_3 = None
        if ops.EXACT_RIM:
            self.w0_3, self.w2_3 = pack_x3(sd[self.p + "0.weight"], 1).to(dev), pack_x3(sd[self.p + "2.weight"], 1).to(dev)

    def forward(self, x, add=None):
        """x: 16-bit rows, or fp32 rows for the split-3 form (prepared under ops.EXACT_RIM)."""
        if x.dtype == torch.float32 and self.w0_3 is not None:
            h = ops.gemm(ops.rows_split3(x), self.w0_3, bias=self.b0, silu=True, out_f32=True)
            return ops.gemm(ops.rows_split3(h), self.w2_3, bias=self.b2, rowvec=add, rows_per_vec=1, out_f32=True)
        h = ops.gemm(ops.to_elem_rows(x), self.w0, bias=self.b0, silu=True)
        return ops.gemm(h, self.w2, bias=self.b2, rowvec=add, rows_per_vec=1, out_f32=True)


class UNetConfig:
    """Hyper-parameters of config.yaml:69-115 (network_config)."""

    def __init__(self, in_channels=8, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), context_dim=1024, adm_in_channels=768,
                 num_head_channels=64, controlnet_mode=True,
                 conditioning_embedding_out_channels=(32, 96, 256, 512), use_apm=False, apm_tokens=17):
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.channel_mult, self.context_dim, self.adm_in_channels = tuple(channel_mult), context_dim, adm_in_channels
        self.num_head_channels, self.controlnet_mode = num_head_channels, controlnet_mode
        self.conditioning_embedding_out_channels = tuple(conditioning_embedding_out_channels)
        self.use_apm, self.apm_tokens = use_apm, apm_tokens          # config.yaml:115 ships use_apm: false
        assert num_head_channels == 64, "kernels are specialised for head dim 64 (config.yaml:93)"


class _EncoderBase:
    """Shared encoder half (time/label embedding, input_blocks, middle_block) of VideoUNet and ControlNet."""

    def enable_forward_chunking(self, dim=0, num_chunks=1):
        """Reference API (video_model.py:498-534, controlnet.py): chunked feed-forward to save memory.  Chunking a row-wise
        feed-forward does not change its result; with 288 GB of HBM the full tensors stay resident, so this is a no-op."""
        return None

    def requires_grad_(self, flag=False):
        return self

    def eval(self):
        return self

    @property
    def stem_x3(self):
        """True when the stem convolution was prepared with split-3 weights (ops.EXACT_RIM at load time): x_tok must then be split-3 rows."""
        return self.input_blocks[0][0].w3 is not None

    def input_tokens(self, x0, x1, scale):
        """NCHW fp32 (x0 * scale | x1) -> the stem's input rows: [F*H*W, 32] 16 bit, or split-3 [F*H*W, 96] for a rim stem."""
        return (ops.nchw_to_tokens_x3 if self.stem_x3 else ops.nchw_to_tokens)(x0, x1, scale, 32)

    def reset_caches(self):
        """Drop the per-chunk caches (context tensors, one-token cross-attention constants, the ControlNet's condition embedding): they hold
        strong references to the last chunk's `context` / control frames (~50 MB) and are keyed on tensor identity + version -- call this
        between videos, or after writing an input through .data / set_() (which the version counter does not see)."""
        self._ctx_cache = None
        if hasattr(self, "_cond_cache"):
            self._cond_cache = None
        for m in self._modules():
            if isinstance(m, SpatialVideoTransformer):
                m._a2c = None

    def _build_encoder(self, cfg):
        mc, emb = cfg.model_channels, cfg.model_channels * 4
        self.cfg, self.mc, self.emb_ch = cfg, mc, emb
        self.time_embed = _EmbedMLP("time_embed.", mc, emb, emb)
        self.label_emb = _EmbedMLP("label_emb.0.", cfg.adm_in_channels, emb, emb)
        self.input_blocks = [[_Conv("input_blocks.0.0.", cfg.in_channels, mc, x3=True)]]
        self.input_block_chans = [mc]
        ch, ds, idx = mc, 1, 1
        for level, mult in enumerate(cfg.channel_mult):
            for _ in range(cfg.num_res_blocks):
                layers = [VideoResBlock(f"input_blocks.{idx}.0.", ch, mult * mc, emb)]
                ch = mult * mc
                if ds in cfg.attention_resolutions:
                    layers.append(SpatialVideoTransformer(f"input_blocks.{idx}.1.", ch, cfg.context_dim, cfg.use_apm, cfg.apm_tokens))
                self.input_blocks.append(layers)
                self.input_block_chans.append(ch)
                idx += 1
            if level != len(cfg.channel_mult) - 1:
                self.input_blocks.append([_Conv(f"input_blocks.{idx}.0.op.", ch, ch, stride=2)])
                self.input_block_chans.append(ch)
                idx += 1
                ds *= 2
        self.middle_block = [VideoResBlock("middle_block.0.", ch, ch, emb),
                             SpatialVideoTransformer("middle_block.1.", ch, cfg.context_dim, cfg.use_apm, cfg.apm_tokens),
                             VideoResBlock("middle_block.2.", ch, ch, emb)]
        self._enc_ch, self._enc_ds = ch, ds

    def _embed(self, timesteps, y):
        if self.time_embed.w0_3 is not None:                        # precision plan: fp32 in, split-3 operands (see _EmbedMLP.prepare)
            emb = self.time_embed.forward(ops.timestep_embedding(timesteps, self.mc, f32=True))
            emb = self.label_emb.forward(y, add=emb)
        else:
            emb = self.time_embed.forward(ops.timestep_embedding(timesteps, self.mc))
            emb = self.label_emb.forward(ops.to_bf16(y), add=emb)  # emb + label_emb(y)
        return ops.to_bf16(emb, silu=True)                          # every emb_layers starts with SiLU

    def _pack_emb_layers(self):
        """Every ResBlock (and its time_stack twin) starts its `emb_layers` with SiLU + Linear(emb -> cout) of the SAME timestep embedding
        (openaimodel.py:328-354): ~90 GEMMs of 50 rows per UNet forward, each a launch that streams 0.8-3.3 MB of weights through 5-20
        workgroups (20 us for 1 us of HBM time; 1.2 % of the stage-1 job, round-2 kernel trace).  Their weights are stacked along N here
        and the forward does ONE GEMM [frames, sum cout]; a block's result is a column slice of it (the GEMM's per-row-group vector operand
        takes a leading dimension).  Bit-identical: every output element sees the same K walk."""
        off, ws, bs = 0, [], []
        for m in self._modules():
            if isinstance(m, VideoResBlock):
                m.e_off, m.et_off = off, off + m.cout
                ws += [m.we, m.twe]; bs += [m.be, m.tbe]          # the block keeps its own copies: a block used on its own (tests) still works
                off += 2 * m.cout
        self._emb_w, self._emb_b = torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()

    @staticmethod
    def _run(layers, h, emb_silu, ctx, tctx, F, T, H, W, sp=None, emb_full=None, out16_of=None):
        """emb_silu / emb_full: rows of the local / of all frames of the PACKED emb_layers output (see _local_conditioning).
        out16_of(rows) -> a 16-bit [rows, C] view or None: where the LAST layer may write its output directly (zero-copy concatenation)."""
        for i, m in enumerate(layers):
            last = out16_of is not None and i == len(layers) - 1
            if isinstance(m, VideoResBlock):
                h = m.forward(h, None, F, T, H, W, sp=sp, emb_out=(emb_silu, emb_full), out16=out16_of(h.shape[0]) if (last and sp is None) else None)
            elif isinstance(m, SpatialVideoTransformer):
                h = m.forward(h, ctx, tctx, F, T, H, W, sp=sp, out16=out16_of(h.shape[0]) if last else None)
            else:
                # stem / Downsample / Upsample convolutions write the stream; the only rim convolution here is the stem (x3=True), and a rim stem
                # (w3) is always fed the split-3 rows input_tokens() makes for it
                view = out16_of(h.shape[0] * (4 if m.ups else 1)) if (last and m.w3 is None and sp is None) else None
                if view is not None:
                    h, H, W = m.forward(h, F, H, W, out=view)          # an Upsample output consumed by the concatenation only: 16 bit, in place
                else:
                    h, H, W = m.forward(h, F, H, W, split3=m.w3 is not None, out_f32=ops.stream_on(m.cout))
        return h, H, W

    def _local_conditioning(self, timesteps, context, y, T, sp):
        """(packed emb_layers output of this rank's frames, of all frames, per-frame context of this rank's frames, per-video context, local frame count).
        timesteps / context / y always describe ALL B*T frames (they are tiny); `sp` selects this rank's rows."""
        emb_full = self._embed(timesteps, y)
        emb_full = ops.gemm(emb_full, self._emb_w, bias=self._emb_b, out_f32=True)      # all blocks' emb_layers at once (_pack_emb_layers)
        B = timesteps.numel() // T
        # one `context` tensor -> one (ctx, tctx) pair: the Euler steps of a chunk pass the same tensor object, and the transformer blocks key
        # their per-chunk cross-attention constants on the identity of what they receive here
        cc = getattr(self, "_ctx_cache", None)
        ver = ops.tensor_version(context)
        if cc is None or ver is None or cc[0] is not context or cc[1] != ver or cc[2] != (T, id(sp), ops.ELEM):
            ctx, tctx = self._contexts(context, T)
            self._ctx_cache = cc = (context, ver, (T, id(sp), ops.ELEM), ctx if sp is None else sp.take_frames(ctx, B, T), tctx, sp)
        ctx, tctx = cc[3], cc[4]
        if sp is None:
            return emb_full, emb_full, ctx, tctx, timesteps.numel()
        return sp.take_frames(emb_full, B, T), emb_full, ctx, tctx, B * sp.frame_counts(T)[sp.rank]

    def _contexts(self, context, T):
        """(per-frame context, per-video context = context[::T]).  One CLIP token per frame (the shipped configuration): 16-bit [F, ctx_dim] /
        [B, ctx_dim].  Several tokens (APM, use_apm: true): fp32 token tensors [F, n, ctx_dim] / [B, n, ctx_dim]; the spatial blocks reduce
        them to one effective token (SpatialVideoTransformer._apm_context), the temporal blocks attend to all of them."""
        if context.dim() == 3 and context.shape[1] != 1:
            if not self.cfg.use_apm:
                raise NotImplementedError("cross-attention contexts with >1 token (APM) need UNetConfig(use_apm=True); the shipped "
                                          "configuration disables it (config.yaml:115 use_apm: false)")
            context = context.float().contiguous()
            return context, context[::T].contiguous()
        if context.dim() == 3:
            context = context[:, 0]
        ctx = ops.to_bf16(context.float().contiguous())
        tctx = ops.to_bf16(context[::T].float().contiguous())       # time_context = context[::T]
        return ctx, tctx


class VideoUNet(_EncoderBase):
    """video_model.py:88-618.  forward() keeps the reference signature; tensors in/out are NCHW fp32."""

    def __init__(self, cfg=None):
        cfg = cfg or UNetConfig()
        self._build_encoder(cfg)
        mc, emb = self.mc, self.emb_ch
        self.controlnet_mode = cfg.controlnet_mode
        self.in_channels, self.model_channels, self.out_channels = cfg.in_channels, mc, cfg.out_channels
        if cfg.controlnet_mode:
            self.cross_attention_merger_input_blocks = [
                ConditionalModel(f"cross_attention_merger_input_blocks.{i}.", c) for i, c in enumerate(self.input_block_chans)]
            self.cross_attention_merger_mid_block = ConditionalModel("cross_attention_merger_mid_block.", self._enc_ch)
        self.output_blocks = []
        chans = list(self.input_block_chans)
        ch, ds, idx = self._enc_ch, self._enc_ds, 0
        for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
            for i in range(cfg.num_res_blocks + 1):
                ich = chans.pop()
                layers = [VideoResBlock(f"output_blocks.{idx}.0.", ch + ich, mc * mult, emb)]
                ch = mc * mult
                if ds in cfg.attention_resolutions:
                    layers.append(SpatialVideoTransformer(f"output_blocks.{idx}.1.", ch, cfg.context_dim, cfg.use_apm, cfg.apm_tokens))
                if level and i == cfg.num_res_blocks:
                    layers.append(_Conv(f"output_blocks.{idx}.{len(layers)}.conv.", ch, ch, ups=1))
                    ds //= 2
                self.output_blocks.append(layers)
                idx += 1
        self.out_conv = _Conv("out.2.", mc, cfg.out_channels)
        self.prepared = False

    def _modules(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle_block
        for blk in self.output_blocks:
            yield from blk
        if self.controlnet_mode:
            yield from self.cross_attention_merger_input_blocks
            yield self.cross_attention_merger_mid_block
        yield self.time_embed
        yield self.label_emb
        yield self.out_conv

    def spec(self):
        s = Spec()
        for m in self._modules():
            m.spec(s)
        s.add("out.0.weight", self.mc); s.add("out.0.bias", self.mc)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        from .params import check_state_dict
        check_state_dict(self.spec(), sd)
        for m in self._modules():
            m.prepare(sd, device)
        self.ow, self.ob = _dev_f32(sd["out.0.weight"], device), _dev_f32(sd["out.0.bias"], device)
        # the head in fp32 (ops.EXACT_RIM): out.2 weights as [tap (ky, kx)][c][4] for svd_head_gn_silu_conv3x3
        self.head_wt = None
        if ops.EXACT_RIM and self.out_channels <= 4 and self.mc % 32 == 0:
            w = sd["out.2.weight"].detach().float()                                      # [cout, C, 3, 3]
            wt = torch.zeros(3, 3, self.mc, 4, dtype=torch.float32)
            wt[..., : self.out_channels] = w.permute(2, 3, 1, 0)
            self.head_wt = wt.reshape(9, self.mc, 4).contiguous().to(device)
            self.head_b = _dev_f32(pad_rows(sd["out.2.bias"].detach().float(), 4), device)
        self._pack_emb_layers()
        self.device = device
        self.prepared = True
        return self

    def forward_tokens(self, x_tok, timesteps, context, y, T, H, W, hs_control_input=None, hs_control_mid=None,
                       num_conditional_frames=None, sp=None):
        """x_tok [F*H*W, 32] bf16 (8 latent channels zero-padded to 32).  Returns [B*T*H*W, 4] fp32 tokens of ALL frames.
        With `sp` (parallel.SeqParallel) x_tok and the hs_control_* tensors hold THIS RANK'S frames only (timesteps / context / y still
        describe all B*T frames); the network output is all-gathered over the group before it is returned."""
        emb_silu, emb_full, ctx, tctx, F = self._local_conditioning(timesteps, context, y, T, sp)
        assert x_tok.shape[0] == F * H * W
        kw = dict(sp=sp, emb_full=emb_full)
        # ---- zero-copy concatenation (ops.ZERO_COPY_CONCAT): one 16-bit buffer [rows, cin] per output block; producers whose result only that block's
        # torch.cat consumes write their column range directly, everything else is rounded into it by cat() exactly as ops.concat_channels did
        n_out = len(self.output_blocks)
        bufs = [None] * n_out

        def cat_view(j, rows, ch, right):
            if not ops.ZERO_COPY_CONCAT or j >= n_out:
                return None
            total = self.output_blocks[j][0].cin
            if bufs[j] is None:
                bufs[j] = torch.empty((rows, total), dtype=ops.ELEM, device=x_tok.device)
            b = bufs[j]
            assert b.shape[0] == rows
            return b[:, total - ch:] if right else b[:, :ch]

        def cat(j, a, b):
            buf = bufs[j]
            if buf is None or buf.shape[1] != a.shape[1] + b.shape[1] or buf.shape[0] != a.shape[0]:
                return ops.concat_channels(a, b)
            same = lambda t, v: t.dtype == v.dtype and t.data_ptr() == v.data_ptr() and t.stride(0) == v.stride(0) and t.shape == v.shape
            va, vb = buf[:, :a.shape[1]], buf[:, a.shape[1]:]
            if not same(a, va):
                ops.to_elem_rows(a, out=va)
            if not same(b, vb):
                ops.to_elem_rows(b, out=vb)
            return buf
        hs = []
        h = x_tok
        for blk in self.input_blocks:
            h, H, W = self._run(blk, h, emb_silu, ctx, tctx, F, T, H, W, **kw)
            hs.append((h, H, W))
        if hs_control_input is not None:
            # CAM: merge ControlNet features into every skip tensor (video_model.py:582-591)
            assert len(hs) == len(hs_control_input) == len(self.cross_attention_merger_input_blocks)
            Tc = num_conditional_frames
            merged = []
            for i, ((hh, Hh, Wh), hc, mg) in enumerate(zip(hs, hs_control_input, self.cross_attention_merger_input_blocks)):
                # the merged skip tensor is consumed by the concatenation in front of output block n - 1 - i only: its 16-bit rounding goes straight
                # into the right-hand columns of that block's buffer where the merger can write it (zero-copy concatenation)
                view = cat_view(n_out - 1 - i, hh.shape[0], hh.shape[1], right=True)
                merged.append((mg.forward(hh, hc, F, T, Tc, Hh, Wh, sp=sp, out16=view), Hh, Wh))
            hs = merged
        # middle_block consumes the UN-merged encoder output (video_model.py:593-600)
        h, H, W = self._run(self.middle_block, h, emb_silu, ctx, tctx, F, T, H, W, **kw)
        if hs_control_mid is not None:
            h = self.cross_attention_merger_mid_block.forward(h, hs_control_mid, F, T, num_conditional_frames, H, W, sp=sp)
        for j, blk in enumerate(self.output_blocks):
            skip, Hs, Ws = hs.pop()
            assert (Hs, Ws) == (H, W)
            h = cat(j, h, skip)
            nxt = (lambda rows, j=j, c=blk[0].cout: cat_view(j + 1, rows, c, right=False)) if j + 1 < n_out else None
            h, H, W = self._run(blk, h, emb_silu, ctx, tctx, F, T, H, W, out16_of=nxt, **kw)
        if self.head_wt is not None:        # GroupNorm + SiLU + conv to 4 channels as one fp32 kernel (precision plan; also faster than a 4-wide MFMA tile)
            out = ops.head_gn_silu_conv3x3(h, F, H, W, self.ow, self.ob, 1e-5, self.head_wt, self.head_b, self.out_channels)
        else:
            h = ops.groupnorm(h, F, H * W, self.ow, self.ob, 1e-5, silu=True)
            out, _, _ = self.out_conv.forward(h, F, H, W, out_f32=True)
        if sp is not None:
            out = sp.gather_frames(out, timesteps.numel() // T, T, H * W)
        return out

    def forward(self, x, timesteps, context=None, y=None, time_context=None, num_video_frames=None,
                num_conditional_frames=None, image_only_indicator=None, hs_control_input=None, hs_control_mid=None):
        """Reference signature (video_model.py:540-556).  x [(B T), 8, H, W] fp32 -> [(B T), 4, H, W] fp32."""
        assert time_context is None and y is not None
        if image_only_indicator is not None and bool(image_only_indicator.any()):
            raise NotImplementedError("image_only_indicator != 0 (AlphaBlender 'where' branch) is unused by StreamingSVD")
        F, _, H, W = x.shape
        x_tok = self.input_tokens(x.float().contiguous(), None, None)
        out = self.forward_tokens(x_tok, timesteps.float().contiguous(), context, y.float().contiguous(),
                                  num_video_frames, H, W, hs_control_input, hs_control_mid, num_conditional_frames)
        return ops.tokens_to_nchw(out, self.out_channels, F, H, W)


class ControlNetConditioningEmbedding:
    """controlnet.py:51-121 with use_normalization (per-pixel LayerNorm) and stride-2 downsampling."""

    def __init__(self, prefix, out_ch, block_out=(32, 96, 256, 512)):
        self.p, self.out_ch, self.bo = prefix, out_ch, tuple(block_out)
        self.conv_in = _Conv(prefix + "conv_in.", 3, block_out[0], x3=True)
        self.blocks = []
        for i in range(len(block_out) - 1):
            self.blocks.append(_Conv(prefix + f"blocks.{2 * i}.", block_out[i], block_out[i], x3=True))
            self.blocks.append(_Conv(prefix + f"blocks.{2 * i + 1}.", block_out[i], block_out[i + 1], stride=2, x3=True))
        self.conv_out = _Conv(prefix + "conv_out.", block_out[-1], out_ch, x3=True)

    def spec(self, s):
        self.conv_in.spec(s)
        for b in self.blocks:
            b.spec(s)
        for i, b in enumerate(self.blocks):
            _spec_ln(s, self.p + f"norms.{i}", b.cout)
        self.conv_out.spec(s)

    def prepare(self, sd, dev):
        self.conv_in.prepare(sd, dev)
        self.conv_out.prepare(sd, dev)
        self.norms = []
        for i, b in enumerate(self.blocks):
            b.prepare(sd, dev)
            self.norms.append((_dev_f32(sd[self.p + f"norms.{i}.weight"], dev), _dev_f32(sd[self.p + f"norms.{i}.bias"], dev)))

    def forward(self, cond_nchw):
        """cond [Fc, 3, 8H, 8W] fp32 in [-1,1] -> tokens [Fc*H*W, out_ch]."""
        Fc, _, H, W = cond_nchw.shape
        if self.conv_in.w3 is not None and all(c <= 512 for c in self.bo):
            # precision plan (ops.EXACT_RIM): the whole embedding with split-3 operands, fp32 between the layers, LayerNorm + SiLU in fp32.
            # It runs once per chunk (ControlNet.embed_condition) and carries 15 % of the forward's squared 16-bit error when run in 16 bit.
            s3 = ops.nchw_to_tokens_x3(cond_nchw.float().contiguous(), None, None, 32)
            h, H, W = self.conv_in.forward(s3, Fc, H, W, split3=True, silu=True, out_f32=True)
            s3 = ops.rows_split3(h)
            for b, (nw, nb) in zip(self.blocks, self.norms):
                h, H, W = b.forward(s3, Fc, H, W, split3=True, out_f32=True)
                s3 = ops.rows_split3(h, ln=(nw, nb), silu=True)
            h, H, W = self.conv_out.forward(s3, Fc, H, W, split3=True, out_f32=True)
            return h, H, W                       # fp32 rows: added to the stem output by svd_add_rows_bf32
        h = ops.nchw_to_tokens(cond_nchw.float().contiguous(), None, None, 32)
        h, H, W = self.conv_in.forward(h, Fc, H, W, silu=True)
        for b, (nw, nb) in zip(self.blocks, self.norms):
            h, H, W = b.forward(h, Fc, H, W)
            h = ops.layernorm(h, nw, nb, silu=True)
        h, H, W = self.conv_out.forward(h, Fc, H, W)
        return h, H, W


class ControlNet(_EncoderBase):
    """controlnet.py:124-554: encoder half of the UNet on the conditioning frames + image-condition embedding."""

    def __init__(self, cfg=None):
        cfg = cfg or UNetConfig()
        self._build_encoder(cfg)
        assert cfg.model_channels == 320, "reference hard-codes the cond-embedding width to 320 (controlnet.py:443-446)"
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(
            "controlnet_cond_embedding.", 320, cfg.conditioning_embedding_out_channels)
        self.prepared = False
        self._cond_cache = None

    @classmethod
    def from_unet(cls, unet, **_ignored):
        return cls(unet.cfg)

    def _modules(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle_block
        yield self.time_embed
        yield self.label_emb
        yield self.controlnet_cond_embedding

    def spec(self):
        s = Spec()
        for m in self._modules():
            m.spec(s)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        from .params import check_state_dict
        check_state_dict(self.spec(), sd)
        for m in self._modules():
            m.prepare(sd, device)
        self._pack_emb_layers()
        self.device = device
        self.prepared = True
        return self

    def embed_condition(self, controlnet_cond):
        """The image-condition embedding depends only on ctrl_frames, not on sigma: it is computed once per
        distinct input tensor and reused across the Euler steps of a chunk (the reference recomputes it every
        step, controlnet.py:520; the values are identical)."""
        c = self._cond_cache
        ver = ops.tensor_version(controlnet_cond)
        if c is None or ver is None or c[0] is not controlnet_cond or c[1] != ver:
            # the cache keeps the input tensor alive: identity (not address) decides, so a freed-and-reallocated buffer of the
            # next video can never alias a stale embedding
            self._cond_cache = c = (controlnet_cond, ver, self.controlnet_cond_embedding.forward(controlnet_cond)[0])
        return c[2]

    def forward_tokens(self, x_tok, timesteps, controlnet_cond, context, y, T, H, W, sp=None):
        """With `sp`: x_tok and controlnet_cond hold this rank's share of the B*T conditioning frames; the returned features do too."""
        with ops.stream_scope(ops.CN_STREAM_F32):       # precision plan: the fp32 residual stream inside the ControlNet (its features leave as fp32 rows)
            emb_silu, emb_full, ctx, tctx, F = self._local_conditioning(timesteps, context, y, T, sp)
            assert x_tok.shape[0] == F * H * W
            cond = self.embed_condition(controlnet_cond)
            hs = []
            h = x_tok
            for i, blk in enumerate(self.input_blocks):
                h, H, W = self._run(blk, h, emb_silu, ctx, tctx, F, T, H, W, sp=sp, emb_full=emb_full)
                if i == 0:
                    if cond.shape[0] != h.shape[0]:
                        raise ValueError(f"controlnet_cond embeds to {cond.shape[0]} tokens but the latent batch has {h.shape[0]}: the control "
                                         f"frames must be {controlnet_cond.shape[0]} x 3 x {8 * H} x {8 * W} pixels (8x the latent, controlnet.py:75-102)")
                    # Merger 'addition', frame_expansion none (controlnet.py:23-48); the rim embedding arrives in fp32
                    h = ops.add_rows_f32b(h, cond) if cond.dtype == torch.float32 else ops.add_rows(h, cond)
                hs.append(h)
            h, H, W = self._run(self.middle_block, h, emb_silu, ctx, tctx, F, T, H, W, sp=sp, emb_full=emb_full)
            return hs, h

    def forward(self, x, timesteps, controlnet_cond, context=None, y=None, time_context=None, num_video_frames=None,
                num_video_frames_conditional=None, image_only_indicator=None):
        """Reference signature (controlnet.py:496-507); returns token tensors (consumed by VideoUNet's CAM mergers)."""
        assert num_video_frames == num_video_frames_conditional
        F, _, H, W = x.shape
        x_tok = self.input_tokens(x.float().contiguous(), None, None)
        return self.forward_tokens(x_tok, timesteps.float().contiguous(), controlnet_cond, context,
                                   y.float().contiguous(), num_video_frames, H, W)
