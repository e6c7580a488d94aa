"""Host side of the DDIM scheduler: timestep / alpha-bar tables and the
``step`` duck type of the reference's ``YHCustomScheduler``
(reference ``src/utils/utils.py:300-461``).  Tables are host scalars; the update
itself is the HIP kernel behind ``loco_sched_step``.
"""
from __future__ import annotations

from typing import Optional

import torch


class SchedulerOutput(object):
    """``prev_sample`` / ``x0`` pair -- reference utils.py:300-303."""

    def __init__(self, xt_next, P_xt):
        self.prev_sample = xt_next
        self.x0 = P_xt


class YHCustomScheduler(object):
    """Linear-beta DDIM schedule with float timesteps (reference utils.py:305-423).

    Quirks kept on purpose: timesteps are non-integer
    floats ``linspace(0,1,N)*999``; alpha-bar is looked up at ``floor(t)``;
    the inversion sequence carries a ``+1e-6`` offset; the last forward
    ``timesteps_next`` is 0.0 (alpha-bar[0], never exactly 1).
    """

    t_max = 999

    def __init__(self, args=None, engine=None):
        noise_schedule = getattr(args, "noise_schedule", None) or "linear"
        if noise_schedule != "linear":
            # define_argparser.py:233 forces 'linear' for every unconditional model
            raise ValueError("only the linear noise schedule is on the unconditional hot path")
        self.noise_schedule = noise_schedule
        self.timesteps = None
        self.timesteps_next = None
        self.learn_sigma = False
        self.engine = engine
        betas = torch.linspace(0.0001, 0.02, 1000, dtype=torch.float64)       # utils.py:408-409
        self.betas = betas.to(torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(torch.float32)  # utils.py:401-403 (f64 cumprod)

    def set_timesteps(self, num_inferences, device=None, is_inversion=False):
        """utils.py:316-329.  The tables stay on the host (`device` accepted for
        signature compatibility: the reference only ever reads scalars from them)."""
        seq = torch.linspace(0, 1, num_inferences) * self.t_max
        if is_inversion:
            seq = seq + 1e-6
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]], dim=0)
            self.timesteps = seq_prev[1:]
            self.timesteps_next = seq[1:]
        else:
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]], dim=0)
            self.timesteps = torch.flip(seq[1:], dims=[0])
            self.timesteps_next = torch.flip(seq_prev[1:], dims=[0])

    def get_timesteps(self, t):
        """utils.py:331-337."""
        t_idx = torch.where(self.timesteps == float(t))
        return self.timesteps_next[t_idx]

    def return_alphas_cumprod(self):
        return self.alphas_cumprod

    def alpha_at(self, t) -> float:
        """``extract(alphas_cumprod, t, .)``: gather at ``t.long()`` -- utils.py:444-461."""
        return float(self.alphas_cumprod[int(torch.as_tensor(float(t)).long().item())])

    def index_of(self, t) -> int:
        """``self.timesteps.tolist().index(t)`` -- utils.py:353."""
        return self.timesteps.tolist().index(float(t))

    def step(self, et, t, xt, eta=0.0, noise: Optional[torch.Tensor] = None, **kwargs):
        """utils.py:342-383 (learn_sigma False branch).  ``noise`` lets a caller
        inject the ``randn_like`` draw of :374; otherwise it is drawn here."""
        if self.engine is None:
            raise RuntimeError("scheduler.step needs the HIP engine (no CPU fallback)")
        assert et.shape == xt.shape, 'et, xt shape should be same'
        t_next = self.timesteps_next[self.index_of(t)]
        at, at_next = self.alpha_at(t), self.alpha_at(t_next)
        if eta != 0 and noise is None:
            noise = torch.randn_like(xt)
        nxt, x0 = self.engine.sched_step(xt.contiguous(), et.contiguous(), at, at_next, eta,
                                         noise if eta != 0 else None, want_x0=True)
        return SchedulerOutput(nxt, x0)
