"""tests/golden/dropin_tiny.pt: one conditional chunk produced by the reference's UNMODIFIED stage-1 code around its OWN networks, and the same
run with the two hot-path objects swapped for ours as INTEGRATION.md section 1 prescribes (build container only: needs /root/reference).

    python oracle/make_golden_dropin.py

Requires the swapped run to reproduce the all-reference run (fp32 torch statements of the launchers on CPU: summation order only) and stores
the all-reference frames; tests/test_dropin_reference.py re-executes the swap against them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dropin_case  # noqa: E402


def main():
    calls = []
    ref = dropin_case.run(swap=False, record=calls)
    got = dropin_case.run(swap=True)
    e = (got - ref).flatten(1).pow(2).mean(1).sqrt()
    print(f"[executed drop-in] reference code around OUR StreamingWrapper / VideoDecoder vs around its own: per-frame L2 max {e.max():.3e} "
          f"(frames rms {ref.pow(2).mean().sqrt():.3f}, {dropin_case.STEPS} Euler steps + decode + clamp)")
    assert e.max() < 2e-4, e
    path = os.path.join(ROOT, "tests", "golden", "dropin_tiny.pt")
    torch.save(dict(frames=ref.clone(), steps=dropin_case.STEPS, case=dropin_case.CASE), path)
    print("wrote", path, os.path.getsize(path), "bytes")
    # the calls the reference's own sampler / denoiser / decode_first_stage made on the two hot-path objects, for the HIP-side replay
    kinds = [c[0] for c in calls]
    assert kinds.count("network") == dropin_case.STEPS and kinds.count("decoder") >= 1, kinds
    pack = lambda v: v                                                  # fp32 as recorded (the whole fixture is ~1.5 MB)
    path = os.path.join(ROOT, "tests", "golden", "dropin_calls_tiny.pt")
    torch.save(dict(calls=[(k, [pack(a) for a in args], {n: pack(x) for n, x in kw.items()}, pack(out)) for k, args, kw, out in calls],
                    steps=dropin_case.STEPS, case=dropin_case.CASE), path)
    print("wrote", path, os.path.getsize(path), "bytes;", kinds)


if __name__ == "__main__":
    main()
