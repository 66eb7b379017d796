"""EMA-VFI on CPU: the oracle against the golden vectors of the vendored network, and the HOST logic of
streamingt2v_amd.ema_vfi (layout, padding, window partition / shift, weight packing, transposed-conv and im2col rewrites, channel
bookkeeping) with the HIP launchers replaced by the fp32 torch statements of tests/vfi_shim.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import vfi_oracle as O  # noqa: E402
from oracle.cases import TINY_VFI, tiny_vfi_inputs, vfi_weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "vfi_tiny.pt")


def _tiny():
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    model = EMAVFI(VFIConfig(F=TINY_VFI["F"], depth=TINY_VFI["depth"]))
    sd = vfi_weights(model.spec())
    inp = tiny_vfi_inputs()
    imgs = torch.cat((inp["img0"], inp["img1"]), 1)
    return model, sd, inp, torch.cat((imgs, imgs.flip(2).flip(3)), 0)


def test_vfi_oracle_matches_vendored_golden():
    torch.set_grad_enabled(False)
    _, sd, inp, x = _tiny()
    g = torch.load(GOLD)
    cfg = O.vfi_config(TINY_VFI["F"], TINY_VFI["depth"])
    o = O.net_forward(sd, cfg, x)
    assert (o["flow"] - g["flow"]).abs().max() <= 1e-4
    for k, v in (("af4", o["af"][4]), ("mf4", o["mf"][1]), ("mask", torch.sigmoid(o["mask"])), ("merged", o["merged"]), ("pred", o["pred"])):
        assert (v - g[k].float()).abs().max() <= 2e-3, k                     # stored as fp16
    assert (O.inference_fast_tta(sd, cfg, inp["img0"], inp["img1"]) - g["tta"]).abs().max() <= 1e-4


def test_vfi_host_logic_matches_oracle(monkeypatch):
    import vfi_shim
    torch.set_grad_enabled(False)
    vfi_shim.install(monkeypatch)
    model, sd, inp, x = _tiny()
    model.load_state_dict(sd, device="cpu")
    H, W = TINY_VFI["H"], TINY_VFI["W"]
    cl = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()                  # NCHW -> channels-last rows
    r = model.net_forward(cl(x[:, :3]), cl(x[:, 3:6]), 2, H, W, want=True)
    o = O.net_forward(sd, O.vfi_config(TINY_VFI["F"], TINY_VFI["depth"]), x)
    for lvl in range(5):
        t, C, h, w = r["af"][lvl]
        assert (t[:, :C] - cl(o["af"][lvl])).abs().max() <= 2e-4, f"af{lvl}"
    for k in range(2):
        t, C, h, w = r["mf"][k]
        assert (t[:, :C] - cl(o["mf"][k])).abs().max() <= 2e-4, f"mf{k + 3}"
    assert (r["fm"][:, :4] - cl(o["flow"])).abs().max() <= 5e-4
    assert (r["fm"][:, 4:5] - cl(o["mask"])).abs().max() <= 5e-4
    assert (r["merged"] - cl(o["merged"])).abs().max() <= 2e-4
    assert (r["pred"] - cl(o["pred"])).abs().max() <= 2e-4
    mid = model.inference(inp["img0"][0].permute(1, 2, 0).contiguous(), inp["img1"][0].permute(1, 2, 0).contiguous())
    g = torch.load(GOLD)
    assert (mid - g["tta"][0].permute(1, 2, 0)).abs().max() <= 2e-4                            # against the VENDORED network's fast-TTA output


def test_vfi_process_frame_arithmetic(monkeypatch):
    """i2v_enhance_interface.vfi_process :30-61: frame selection / interleaving / duplication, BGR flip, uint8 truncation, 1280x720 resize."""
    from streamingt2v_amd import ema_vfi

    class Fake:
        def inference(self, a, b, want_uint8=False):
            m = (a + b) / 2
            return m, (m * 255.0).to(torch.uint8)

    rng = np.random.default_rng(0)
    video = [rng.integers(0, 256, (16, 32, 3), dtype=np.uint8) for _ in range(6)]
    infer = lambda a, b: (a + b) / 2
    for n, have in ((7, 4), (8, 4), (5, 3), (8, 6)):        # the pipeline hands over (n + 1) // 2 frames; surplus frames follow the reference too
        got = ema_vfi.vfi_process(video[:have], Fake(), n, out_size=(64, 32), device="cpu")
        ref = O.vfi_process(video[:have], infer, n, out_size=(64, 32))
        assert len(got) == len(ref) and (have != (n + 1) // 2 or len(got) == n)
        for a, b in zip(got, ref):
            assert a.size == (64, 32) and np.array_equal(np.asarray(a), np.asarray(b))


def test_window_geometry_matches_oracle():
    from streamingt2v_amd.ema_vfi import window_geometry
    for h, w in ((6, 10), (3, 5), (7, 14), (90, 160)):
        for shift in (0, 3):
            a, b = window_geometry(h, w, 7, shift), O.window_masks(h, w, 7, shift)
            assert a[:2] == b[:2]
            assert (a[2] is None) == (b[2] is None) and (a[2] is None or torch.equal(a[2], b[2]))


def test_vfi_inference_tta_modes(monkeypatch):
    """Trainer.Model.inference :84-101: fast TTA, two-pass TTA and no TTA against the oracle's statement of each."""
    import vfi_shim
    torch.set_grad_enabled(False)
    vfi_shim.install(monkeypatch)
    model, sd, inp, x = _tiny()
    model.load_state_dict(sd, device="cpu")
    cfg = O.vfi_config(TINY_VFI["F"], TINY_VFI["depth"])
    a, b = inp["img0"][0].permute(1, 2, 0).contiguous(), inp["img1"][0].permute(1, 2, 0).contiguous()
    imgs = torch.cat((inp["img0"], inp["img1"]), 1)
    plain = O.net_forward(sd, cfg, imgs)["pred"]
    flipped = O.net_forward(sd, cfg, imgs.flip(2).flip(3))["pred"].flip(2).flip(3)
    nchw = lambda t: t.permute(2, 0, 1)[None]
    assert (nchw(model.inference(a, b, TTA=False, fast_TTA=False)) - plain).abs().max() <= 2e-4
    assert (nchw(model.inference(a, b, TTA=True, fast_TTA=False)) - (plain + flipped) / 2).abs().max() <= 2e-4
    assert (nchw(model.inference(a, b)) - O.inference_fast_tta(sd, cfg, inp["img0"], inp["img1"])).abs().max() <= 2e-4
    m, u8 = model.inference(a, b, TTA=False, fast_TTA=False, want_uint8=True)
    assert torch.equal(u8, (m * 255.0).to(torch.uint8))


def test_vfi_process_matches_reference_function_golden():
    """ema_vfi.vfi_process against the output of the reference's UNMODIFIED i2v_enhance_interface.vfi_process (tests/golden/
    vfi_process_tiny.pt, oracle/make_golden_vfi_process.py) around the same stand-in interpolator: the frames handed to `inference`
    (bit-equal: BGR order, k / 255. in fp32) and every output frame at 1280 x 720 (compared on the stored 40-pixel grid)."""
    from oracle.cases import tiny_vfi_process_infer, tiny_vfi_process_inputs
    from streamingt2v_amd import ema_vfi
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "vfi_process_tiny.pt"))
    seen = []

    class Fake:
        def inference(self, a, b, want_uint8=False):
            nchw = lambda t: t.permute(2, 0, 1)[None]
            seen.append((nchw(a).clone(), nchw(b).clone()))
            m = tiny_vfi_process_infer(nchw(a), nchw(b))[0].permute(1, 2, 0).contiguous()
            return m, (m * 255.0).to(torch.uint8)

    for n in (7, 8):
        seen.clear()
        frames = ema_vfi.vfi_process(tiny_vfi_process_inputs()[: (n + 1) // 2], Fake(), n, device="cpu")
        g = gold[n]
        assert len(frames) == n and len(seen) == len(g["pairs"])
        for (a, b), (ra, rb) in zip(seen, g["pairs"]):
            assert torch.equal(a, ra) and torch.equal(b, rb)
        sub = np.stack([np.asarray(f)[::40, ::40] for f in frames], 0)
        assert all(f.size == (1280, 720) for f in frames) and np.array_equal(sub, g["sub"].numpy())


def test_load_vfi_from_reference_style_checkpoint(tmp_path):
    """pipeline.load_vfi: the shipped ours.pkl stores DDP keys (``module.``) plus cached attn_mask / HW buffers (Trainer.load_model :36-47)."""
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    cfg = VFIConfig(F=TINY_VFI["F"], depth=TINY_VFI["depth"])
    sd = vfi_weights(EMAVFI(cfg).spec())
    ckpt = {"module." + k: v for k, v in sd.items()}
    ckpt["module.feature_bone.block4.1.attn_mask"] = torch.zeros(4, 49, 49)
    ckpt["module.feature_bone.block4.1.HW"] = torch.tensor([196.0])
    path = str(tmp_path / "ours.pkl")
    torch.save(ckpt, path)
    m = P.load_vfi(path, device="cpu", cfg=cfg)
    assert m.loaded and m.device == "cpu"
    assert P.load_vfi(sd, device="cpu", cfg=cfg).loaded                   # an already converted state_dict loads too


def test_vfi_host_logic_at_shipped_width(monkeypatch):
    """The shipped configuration (F = 32: 32 .. 512 channels, 8 / 16 heads, channel counts such as 81, 134 and 224 that need padding) on a
    small frame pair: EMAVFI's host logic with the fp32 stand-in launchers against the oracle, and its fast-TTA result against the output
    of the UNMODIFIED vendored network at F = 32 (tests/golden/vfi_fullarch.pt, oracle/make_golden_fullarch_small.py)."""
    import vfi_shim
    from oracle.cases import fullarch_small_inputs
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    torch.set_grad_enabled(False)
    vfi_shim.install(monkeypatch)
    model = EMAVFI(VFIConfig())
    sd = vfi_weights(model.spec(), seed=12)
    model.load_state_dict(sd, device="cpu")
    inp = fullarch_small_inputs()
    H, W = inp["img0"].shape[2:]
    x = torch.cat((inp["img0"], inp["img1"]), 1)
    cl = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    r = model.net_forward(cl(x[:, :3]), cl(x[:, 3:6]), 1, H, W, want=True)
    o = O.net_forward(sd, O.vfi_config(32, (2, 2, 2, 4, 4)), x)
    for lvl in range(5):
        t, C, h, w = r["af"][lvl]
        assert C == 32 * 2 ** lvl and (t[:, :C] - cl(o["af"][lvl])).abs().max() <= 5e-4, f"af{lvl}"
    assert (r["fm"][:, :4] - cl(o["flow"])).abs().max() <= 2e-3
    assert (r["pred"] - cl(o["pred"])).abs().max() <= 5e-4
    mid = model.inference(inp["img0"][0].permute(1, 2, 0).contiguous(), inp["img1"][0].permute(1, 2, 0).contiguous())
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "vfi_fullarch.pt"))["tta"]
    assert (mid - gold[0].permute(1, 2, 0)).abs().max() <= 5e-4
