"""Parameter bookkeeping shared by the host modules: state-dict specs and a by-name deterministic initialiser.

The host modules keep the reference's state_dict key layout (checkpoint
``PAIR/StreamingSVD/model.safetensors`` prefixes ``model.diffusion_model.``, ``controlnet.``,
``first_stage_model.``), so a reference checkpoint loads with ``strict=True``
(reference: code/inference_i2v.py:128-141).
"""
import zlib

import torch


class Spec(list):
    """Ordered list of (name, shape) pairs."""

    def add(self, name, *shape):
        self.append((name, tuple(int(s) for s in shape)))

    def names(self):
        return [n for n, _ in self]

    def numel(self):
        t = 0
        for _, s in self:
            n = 1
            for d in s:
                n *= d
            t += n
        return t


def init_by_name(spec, seed=0, device="cpu"):
    """Random parameters that depend only on (name, shape, seed) -- not on construction order.

    Used for synthetic weights (no checkpoints are available offline) by tests, the golden-vector generator
    (which pushes the SAME values into the reference modules) and bench.py.  Zero-initialised layers of the
    reference (zero_module / CAM proj_out) get non-zero values so that they cannot hide bugs.
    Scales keep activations O(1): weights ~ N(0, 1/fan_in), norm gains ~ 1 + 0.1 N, biases ~ 0.02 N.
    """
    sd = {}
    gdev = "cuda" if str(device).startswith("cuda") else "cpu"
    for name, shape in spec:
        g = torch.Generator(device=gdev)
        g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        r = torch.randn(shape, generator=g, device=gdev, dtype=torch.float32)
        if name.endswith("mix_factor"):
            v = r * 0.5
        elif len(shape) == 1 and name.endswith(".weight"):
            v = 1.0 + 0.1 * r
        elif len(shape) == 1:
            v = 0.02 * r
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            v = r * (1.0 / fan_in ** 0.5)
        sd[name] = v.to(device)
    return sd


def check_state_dict(spec, sd, prefix=""):
    """strict=True semantics of nn.Module.load_state_dict for our spec."""
    want = {prefix + n: s for n, s in spec}
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k.startswith(prefix) and k not in want] if prefix else \
        [k for k in sd if k not in want]
    bad = [k for k in want if k in sd and tuple(sd[k].shape) != want[k]]
    if missing or unexpected or bad:
        raise RuntimeError(f"state_dict mismatch: missing={missing[:5]} (+{max(0, len(missing) - 5)}) "
                           f"unexpected={unexpected[:5]} (+{max(0, len(unexpected) - 5)}) shape={bad[:5]}")
