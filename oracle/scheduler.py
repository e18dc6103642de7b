"""Oracle (test infrastructure): EulerDiscreteScheduler as configured for SVD-XT
(v_prediction, Karras sigmas, sigma_min 0.002, sigma_max 700, rho 7, continuous timesteps,
"leading" spacing, s_churn 0 => deterministic).

Reference call site: ``num_inference_steps`` at /root/reference/model/depthcrafter.py:86; the
scheduler object itself is loaded from ``pre_train_path`` (:24-29).  Algorithm restated from
un-vendored diffusers; PARITY UNPINNED (closed-form tables below are self-consistent
known-answers: sigma_0 = 700, sigma_{N-1} = 0.002, t_i = 0.25 ln sigma_i).
"""
import numpy as np
import torch


class EulerKarrasVPred:
    def __init__(self, sigma_min=0.002, sigma_max=700.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho
        self.sigmas = None
        self.timesteps = None

    def set_timesteps(self, n: int):
        ramp = np.linspace(0, 1, n)
        lo, hi = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        sig = (hi + ramp * (lo - hi)) ** self.rho                     # float64
        sig = torch.from_numpy(sig).to(torch.float32)
        self.timesteps = torch.tensor([0.25 * float(s.log()) for s in sig], dtype=torch.float32)
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        return self.timesteps

    @property
    def init_noise_sigma(self):
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)            # "leading" spacing

    def scale_model_input(self, sample, i):
        s = self.sigmas[i]
        return sample / ((s ** 2 + 1) ** 0.5)       # stays in sample.dtype (0-dim fp32 divisor)

    def step(self, model_output, i, sample):
        """One Euler step.  Mirrors diffusers' dtype behaviour: ``sample`` is upcast to fp32,
        ``model_output`` is NOT, so ``model_output * c`` rounds to model_output.dtype first."""
        out_dtype = model_output.dtype
        sample = sample.to(torch.float32)
        s = self.sigmas[i]
        pred_x0 = model_output * (-s / (s ** 2 + 1) ** 0.5) + (sample / (s ** 2 + 1))
        d = (sample - pred_x0) / s
        dt = self.sigmas[i + 1] - s
        return (sample + d * dt).to(out_dtype)
