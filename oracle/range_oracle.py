"""CPU restatement of the reference's range / quantisation glue.  TEST INFRASTRUCTURE ONLY.

    convert_range(video, output_range=[0, 255], input_range=[-1, 1])       code/utils/result_processor.py:4-14
        (called on every chunk at code/diffusion_trainer/streaming_svd.py:353)
    IImage(chunk, vmin=0, vmax=255) -> torch2np                            code/lib/farancia/libimage/iimage.py:21-39, 119-141
        x = 255 * (x.clip(0, 255) - 0) / (255 - 0) ; permute(0, 2, 3, 1) ; .to(torch.uint8)   -- TRUNCATION, not rounding

Pinned: oracle/make_golden_range.py runs the unmodified reference functions (import stubs for imageio / torchvision / cv2 only)
on seeded frames plus adversarial values next to every integer boundary; tests/golden/range_tiny.pt holds their uint8 output and
this restatement must reproduce it bit for bit (byte work: exact).
"""
import torch


def frames_to_uint8(frames):
    """frames fp32 [F, 3, H, W] in [-1, 1]  ->  uint8 [F, H, W, 3].  Every operation in fp32, in the reference's order."""
    v = frames.float()
    v = (v - (-1)) / (1 - (-1))                 # convert_range: to [0, 1]
    v = v * (255 - 0) + 0                       #                to [0, 255]
    v = 255 * (v.clip(0, 255) - 0) / (255 - 0)  # torch2np: (255 * x) / 255 is NOT the identity in fp32
    return v.permute(0, 2, 3, 1).to(torch.uint8)
