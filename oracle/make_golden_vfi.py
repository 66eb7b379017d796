"""Generate tests/golden/vfi_tiny.pt from the UNMODIFIED vendored EMA-VFI of the reference (build container only).

    python oracle/make_golden_vfi.py        # needs /root/reference ; writes tests/golden/vfi_tiny.pt

The vendored network (code/i2v_enhance/thirdparty/VFI, imported through oracle/vfi_bootstrap.py) gets by-name deterministic weights
(oracle/cases.vfi_weights; load_state_dict(strict=True) proves the key/shape spec) and the seeded tiny frame pair; the restatement
oracle/vfi_oracle.py must agree with it to <= 1e-4 on features, flow, mask and prediction, and on the fast-TTA inference result.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vfi_bootstrap as vb  # noqa: E402
from oracle import vfi_oracle as O  # noqa: E402
from oracle.cases import TINY_VFI, tiny_vfi_inputs, vfi_weights  # noqa: E402
from streamingt2v_amd.params import Spec  # noqa: E402

TOL = 1e-4


def main():
    torch.set_grad_enabled(False)
    model = vb.build_reference(F=TINY_VFI["F"], depth=TINY_VFI["depth"])
    spec = Spec()
    for k, v in model.net.state_dict().items():
        spec.add(k, *v.shape)
    sd = vfi_weights(spec)
    model.net.load_state_dict(sd, strict=True)
    inp = tiny_vfi_inputs()
    imgs = torch.cat((inp["img0"], inp["img1"]), 1)
    x = torch.cat((imgs, imgs.flip(2).flip(3)), 0)
    with vb.cpu_only():
        af, mf = model.net.feature_bone(x[:, :3], x[:, 3:6])
        flow_list, mask_list, merged, pred = model.net(x, timestep=0.5)
        tta = model.inference(inp["img0"], inp["img1"], TTA=True, fast_TTA=True)
    cfg = O.vfi_config(TINY_VFI["F"], TINY_VFI["depth"])
    o = O.net_forward(sd, cfg, x)
    errs = {f"af{i}": (af[i] - o["af"][i]).abs().max().item() for i in range(5)}
    errs.update({f"mf{i + 3}": (mf[i + 3] - o["mf"][i]).abs().max().item() for i in range(2)})
    errs.update(flow=(flow_list[-1] - o["flow"]).abs().max().item(), merged=(merged[-1] - o["merged"]).abs().max().item(),
                pred=(pred - o["pred"]).abs().max().item(), tta=(tta - O.inference_fast_tta(sd, cfg, inp["img0"], inp["img1"])).abs().max().item())
    print("[vfi] vendored-vs-oracle max abs err:", {k: f"{v:.2e}" for k, v in errs.items()})
    print(f"[vfi] |flow| max {flow_list[-1].abs().max():.2f} px, pred mean {pred.mean():.3f} std {pred.std():.3f}, "
          f"|pred - merged| max {(pred - merged[-1]).abs().max():.3f}")
    assert max(errs.values()) <= TOL, errs
    # the product's spec must equal the vendored module's state_dict (tiny and production size)
    try:
        from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
        assert dict(EMAVFI(VFIConfig(F=TINY_VFI["F"], depth=TINY_VFI["depth"])).spec()) == dict(spec), "tiny spec mismatch"
        full = vb.build_reference(F=32, depth=(2, 2, 2, 4, 4))
        assert dict(EMAVFI(VFIConfig()).spec()) == {k: tuple(v.shape) for k, v in full.net.state_dict().items()}, "full spec mismatch"
        print("[vfi] streamingt2v_amd.ema_vfi spec == vendored state_dict (tiny and F=32)")
    except ImportError as e:
        print("[vfi] product module not importable yet:", e)
    out = os.path.join(ROOT, "tests", "golden", "vfi_tiny.pt")
    torch.save(dict(af4=af[4].half(), mf4=mf[4].half(), flow=flow_list[-1].clone(), mask=(mask_list[-1]).half(), merged=merged[-1].half(),
                    pred=pred.half(), tta=tta.clone()), out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
