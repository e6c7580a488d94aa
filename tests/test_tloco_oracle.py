"""CPU: the T-LOCO oracle (oracle/tloco_oracle.py) against the fixture the reference's own EditDeepFloydIF methods
produced (tests/golden/tloco_tiny.pt, oracle/make_golden_tloco.py), plus the host logic of the product module."""
import pytest
import torch

import loco_oracle as orc
import tloco_oracle as tl
import loco_edit_amd  # noqa: F401
from loco_edit_amd.config import TINY_ADM, synth_params
from loco_edit_amd.tloco import IFScheduler, cfg_weights, cond_params


@pytest.fixture(scope="module")
def setup(golden):
    g = golden("tloco_tiny")
    p = orc.to_torch(synth_params(TINY_ADM, 0))
    p.update({k: torch.from_numpy(v) for k, v in cond_params(TINY_ADM, g["cond_dim"], 0).items()})
    ot = tl.OracleTLoco(p, TINY_ADM, guidance_scale=g["guidance_scale"], guidance_scale_edit=g["guidance_scale_edit"])
    return g, ot


def test_if_scheduler_tables(setup):
    g, ot = setup
    s = IFScheduler()
    assert torch.equal(s.alphas_cumprod, g["alphas_cumprod"]) and torch.equal(ot.sched.alphas_cumprod, g["alphas_cumprod"])
    s.set_timesteps(100)
    assert torch.equal(s.timesteps, g["timesteps"])
    assert g["edit_t_idx"] == 39 and float(g["t"]) == 600.0 == float(s.timesteps[39])
    assert abs(s.alpha_at(g["t"]) - float(g["alphas_cumprod"][600])) < 1e-12
    assert float(s.timesteps_next[-1]) == 0.0 and float(s.timesteps[0]) == 990.0


def test_cfg_weights_cover_the_reference_modes():
    g_, ge = 7.5, 4.0
    for mode in tl.MODES:
        w = dict(cfg_weights(mode, g_, ge))
        tot = sum(w.values())
        assert abs(tot - (1.0 if mode.startswith("null") else 0.0)) < 1e-12        # "null+..." keeps eps_null's unit weight
    assert cfg_weights("null+(for-null)", g_, ge, do_cfg=False) == [("for", 1.0)]
    assert dict(cfg_weights("null+(for-null)+(edit-null)", g_, ge)) == {"for": 7.5, "edit": 4.0, "null": -10.5}
    with pytest.raises(NotImplementedError):
        cfg_weights("edit-proj[for](edit)", g_, ge)


def test_oracle_cfg_noise_x0_and_directions(setup):
    g, ot = setup
    x, t = g["x"], g["t"]
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    xb = torch.cat([x, x.flip(-1)], dim=0)
    with torch.no_grad():
        for mode, ref in g["eps_modes"].items():
            assert torch.allclose(ot.cfg_noise(xb, t, F, E, N, mode), ref, rtol=1e-4, atol=1e-4)
        assert torch.allclose(ot.cfg_noise(xb, t, F, E, N, "null+(for-null)", do_cfg=False), g["eps_nocfg"], rtol=1e-4, atol=1e-4)
        assert torch.allclose(ot.get_x0(x, t, F, E, N, mask=g["mask"]), g["x0_masked"], rtol=1e-4, atol=1e-4)
        for mode, ref in g["v_direct"].items():
            assert torch.allclose(ot.v_modify_direct(x, t, F, E, N, mode), ref, rtol=1e-4, atol=1e-4)
    v = ot.delta_xt_via_grad(x, t, F, E, N, mask=g["mask"])
    assert torch.allclose(v, g["v_grad"], rtol=1e-3, atol=1e-6) and abs(float(v.norm()) - 1.0) < 1e-5


def test_oracle_cfg_solver(setup):
    g, ot = setup
    mode = "null+(for-null)"
    sv = g["solver"][mode]
    u, s, vT = ot.pullback(g["x"], g["t"], g["for_e"], g["edit_e"], g["null_e"], 3, g["v0"], min_iter=sv["n_iter"],
                           max_iter=sv["n_iter"], mask=sv["mask"], mode=mode)
    assert torch.allclose(s, sv["s"], rtol=1e-3)
    assert (vT.double() * sv["vT"].double()).sum(dim=1).abs().min() > 0.9999
    assert u.shape == sv["u"].shape
