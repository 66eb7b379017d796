"""tests/golden/ar_loop_tiny.pt from the reference's UNMODIFIED autoregressive outer loop (build container only).

    python oracle/make_golden_ar_loop.py

`StreamingSVD._autoregressive_generation` with its own `extract_anchor_frames` / `extract_ctrl_frames` and
`utils.result_processor.convert_range` / `concat_chunks` (code/diffusion_trainer/streaming_svd.py:228-356) is called as an unbound method
on a bare object whose `_generate_conditional_output` is the deterministic stand-in of oracle/cases.tiny_ar_generate -- so the golden uint8
video pins which frames the loop hands over (anchor = chunk0[6], control frames = last 7 frames of the PREVIOUS chunk's kept frames),
which frames it keeps (result[7:]), the chunk concatenation and the final [-1,1] -> [0,255] truncation.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ar_bootstrap  # noqa: E402
from oracle.cases import TINY_AR, tiny_ar_chunk0, tiny_ar_generate  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    Ref = ar_bootstrap.install()
    params = types.SimpleNamespace(n_autoregressive_generations=TINY_AR["n_ar"], anchor_frames=str(TINY_AR["anchor"]), num_conditional_frames=TINY_AR["Tc"])
    seen = []

    class Bare:
        extract_anchor_frames = Ref.extract_anchor_frames
        extract_ctrl_frames = Ref.extract_ctrl_frames

        def _generate_conditional_output(self, svd_input_frame, inference_params, anchor_frames, ctrl_frames):
            seen.append((svd_input_frame.clone(), ctrl_frames.clone()))
            return tiny_ar_generate(svd_input_frame, ctrl_frames, len(seen) - 1)

    video = Ref._autoregressive_generation(Bare(), tiny_ar_chunk0(), params)
    u8 = torch.from_numpy(video.data.copy())                               # IImage -> uint8 [F, H, W, 3]
    n = TINY_AR["T"] + TINY_AR["n_ar"] * (TINY_AR["T"] - TINY_AR["Tc"])
    assert tuple(u8.shape) == (n, TINY_AR["H"], TINY_AR["W"], 3) and u8.dtype == torch.uint8, u8.shape
    print(f"[ar loop] reference video {tuple(u8.shape)}; control frames seen: {[tuple(c.shape) for _, c in seen]}")
    out = os.path.join(ROOT, "tests", "golden", "ar_loop_tiny.pt")
    torch.save(dict(video=u8, anchors=[a for a, _ in seen], ctrl=[c for _, c in seen]), out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
