"""tests/golden/range_tiny.pt from the UNMODIFIED reference range glue (row A13); build container only.

    python oracle/make_golden_range.py

Imports code/utils/result_processor.py and code/lib/farancia/libimage/iimage.py from /root/reference with import stubs for
imageio, torchvision and cv2 (none of them touches the arithmetic).  Inputs: seeded frames in [-1, 1] plus, for every integer
k in 0..255, the fp32 values around the pre-image of k (where truncation flips) -- regenerated from the seed by the tests.
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/code")


def _stub(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m


for n, a in (("imageio", {}), ("imageio.v3", {}), ("torchvision", {}), ("torchvision.utils", {"flow_to_image": None}),
             ("torchvision.transforms", {}), ("torchvision.transforms.functional", {}), ("cv2", {"COLORMAP_JET": 2})):
    _stub(n, **a)

from oracle.cases import range_inputs  # noqa: E402
from oracle.range_oracle import frames_to_uint8  # noqa: E402


def main():
    from utils.result_processor import concat_chunks, convert_range
    x = range_inputs()
    ref = concat_chunks([convert_range(x, output_range=[0, 255], input_range=[-1, 1])]).data      # numpy uint8 [F, H, W, 3]
    ref = torch.from_numpy(ref.copy())
    ora = frames_to_uint8(x)
    assert ora.dtype == torch.uint8 and torch.equal(ora, ref), "range oracle != reference"
    naive = ((x + 1) * 127.5).permute(0, 2, 3, 1).to(torch.uint8)
    print(f"[range] oracle == reference on {ref.numel()} bytes; a naive (x+1)*127.5 truncation differs in {(naive != ref).sum().item()} bytes")
    out = os.path.join(ROOT, "tests", "golden", "range_tiny.pt")
    torch.save({"u8": ref}, out)
    print("wrote", out)


if __name__ == "__main__":
    main()
