"""Independent checks of the DDIM restatement (diffusers' DDIMScheduler is neither vendored nor installed: oracle/i2vgen_oracle.DDIM and
streamingt2v_amd.enhance.DDIMSchedule restate it).  Nothing here re-uses the restated formulas: every expectation is derived from the
papers' definitions -- DDIM (Song et al. 2021, eq. 12 with sigma = 0), v-prediction (Salimans & Ho 2022: v = sqrt(abar) eps - sqrt(1-abar) x0),
zero-terminal-SNR rescaling (Lin et al. 2023, Alg. 1) -- or from the closed form of the 'scaled_linear' beta schedule."""
import numpy as np
import pytest
import torch

from oracle.i2vgen_oracle import DDIM
from streamingt2v_amd.enhance import DDIMSchedule


def test_beta_table_closed_form_and_zero_terminal_snr():
    T, b0, b1 = 1000, 0.00085, 0.012
    t = np.arange(T, dtype=np.float64)
    betas = (np.sqrt(b0) + t / (T - 1) * (np.sqrt(b1) - np.sqrt(b0))) ** 2                 # 'scaled_linear': linear in sqrt(beta)
    abar = np.cumprod(1.0 - betas)
    # zero terminal SNR (Lin et al.): shift sqrt(abar) so that the last value is 0, rescale so that the first value is unchanged
    s = np.sqrt(abar)
    s = (s - s[-1]) * s[0] / (s[0] - s[-1])
    want = s ** 2
    for sched in (DDIM(), DDIMSchedule()):
        got = sched.alphas_cumprod.double().numpy()
        assert got[-1] == 0.0 and abs(got[0] - abar[0]) < 1e-7                               # SNR(T) = 0 exactly; abar_0 untouched
        assert np.all(np.diff(got) < 0)
        assert np.abs(got - want).max() < 5e-6                                               # fp32 table vs the float64 closed form
        assert abs(float(sched.final_alpha_cumprod) - abar[0]) < 1e-7                        # set_alpha_to_one = False


def test_leading_timesteps_and_img2img_truncation():
    """set_timesteps(30) with 'leading' spacing + steps_offset 1 on 1000 train steps: t_i = (29 - i) * 33 + 1; strength 0.97 keeps
    int(30 * 0.97) = 29 of them (pipeline_i2vgen_xl.py:541-551)."""
    want = [(29 - i) * 33 + 1 for i in range(30)]
    o = DDIM(); o.set_timesteps(30)
    assert o.timesteps.tolist() == want
    p = DDIMSchedule()
    assert p.set_timesteps(30) == want
    assert p.get_timesteps(30, 0.97) == want[1:] and len(p.get_timesteps(30, 0.97)) == 29
    assert p.get_timesteps(10, 0.35) == [(9 - i) * 100 + 1 for i in range(10)][7:]


@pytest.mark.parametrize("t_index", [0, 1, 15, 28, 29])
def test_deterministic_ddim_step_transports_the_exact_solution(t_index):
    """If the network returns the TRUE v of x_t = sqrt(abar_t) x0 + sqrt(1 - abar_t) eps, one eta = 0 DDIM step must land exactly on
    x_prev = sqrt(abar_prev) x0 + sqrt(1 - abar_prev) eps (the defining property of the deterministic sampler); the last step (prev < 0)
    uses final_alpha_cumprod = abar_0."""
    g = torch.Generator(); g.manual_seed(t_index)
    x0, eps = torch.randn(2, 4, 3, 5, 7, generator=g, dtype=torch.float64), torch.randn(2, 4, 3, 5, 7, generator=g, dtype=torch.float64)
    o = DDIM(); o.set_timesteps(30)
    o.alphas_cumprod = o.alphas_cumprod.double(); o.final_alpha_cumprod = o.final_alpha_cumprod.double()
    t = int(o.timesteps[t_index])
    abar_t = o.alphas_cumprod[t]
    prev = t - 1000 // 30
    abar_p = o.alphas_cumprod[prev] if prev >= 0 else o.alphas_cumprod[0]
    x_t = abar_t.sqrt() * x0 + (1 - abar_t).sqrt() * eps
    assert torch.allclose(o.add_noise(x0, eps, t), x_t, atol=1e-12)
    v = abar_t.sqrt() * eps - (1 - abar_t).sqrt() * x0
    want = abar_p.sqrt() * x0 + (1 - abar_p).sqrt() * eps
    assert torch.allclose(o.step(v, t, x_t), want, atol=1e-10)
    # the product's table hands the same (abar_t, abar_prev) pair to the HIP step kernel
    p = DDIMSchedule(); p.set_timesteps(30)
    a_t, a_p = p.alphas(t)
    assert abs(a_t - float(abar_t)) < 1e-7 and abs(a_p - float(abar_p)) < 1e-7
