"""FusedAdam as a torch.optim.Optimizer: closure-style step (PL-1.9 automatic optimisation hands training_step +
backward in as the closure) and checkpoint interchange of the optimizer state in torch.optim.Adam's layout
(`optimizer_states` of a Lightning .ckpt; src/models/autoencoder.py:365-377, train.py:228-249)."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(24, 40), nn.Tanh(), nn.Linear(40, 8), nn.Tanh(), nn.Linear(8, 3))


def _data(step):
    g = torch.Generator().manual_seed(900 + step)
    return torch.randn(16, 24, generator=g), torch.randn(16, 3, generator=g)


def test_closure_step_matches_torch_adam(hip_lib):
    from odvae_amd.optim import FusedAdam
    ref = _net(3)
    net = copy.deepcopy(ref).cuda()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=2e-3, betas=(0.5, 0.9))
    opt = FusedAdam(net.parameters(), lr=2e-3, betas=(0.5, 0.9))
    for step in range(4):
        x, y = _data(step)

        def closure_ref():
            opt_ref.zero_grad()
            loss = (ref(x) - y).pow(2).mean()
            loss.backward()
            return loss

        def closure():
            opt.zero_grad()
            loss = (net(x.cuda()) - y.cuda()).pow(2).mean()   # would raise under no_grad: the closure needs autograd
            loss.backward()
            return loss

        l_ref = opt_ref.step(closure_ref)
        l = opt.step(closure)
        assert abs(l.item() - l_ref.item()) <= 1e-5 * max(1.0, abs(l_ref.item()))
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.detach().cpu(), q.detach(), atol=2e-6), (p.detach().cpu() - q.detach()).abs().max()


def test_state_dict_layout_and_resume(hip_lib, tmp_path):
    """3 steps -> save -> fresh model + optimizer -> load -> 3 more steps == 6 uninterrupted steps, bit for bit; the
    saved state reads as torch.optim.Adam's and loads INTO torch.optim.Adam (and back)."""
    from odvae_amd.optim import FusedAdam

    def run(net, opt, steps):
        for step in steps:
            x, y = _data(step)
            opt.zero_grad()
            (net(x.cuda()) - y.cuda()).pow(2).mean().backward()
            opt.step()

    net_a = _net(5).cuda()
    opt_a = FusedAdam(net_a.parameters(), lr=1e-3, betas=(0.5, 0.9))
    run(net_a, opt_a, range(6))

    net_b = _net(5).cuda()
    opt_b = FusedAdam(net_b.parameters(), lr=1e-3, betas=(0.5, 0.9))
    run(net_b, opt_b, range(3))
    path = str(tmp_path / "opt.ckpt")
    torch.save({"state_dict": net_b.state_dict(), "optimizer_states": [opt_b.state_dict()]}, path)

    ck = torch.load(path, map_location="cpu")
    osd = ck["optimizer_states"][0]
    n_params = len(list(net_b.parameters()))
    assert sorted(osd["state"].keys()) == list(range(n_params))
    for st in osd["state"].values():
        assert set(st.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 3.0
    assert osd["param_groups"][0]["betas"] == (0.5, 0.9) and osd["param_groups"][0]["params"] == list(range(n_params))

    net_c = _net(77).cuda()           # different init: everything must come from the checkpoint
    net_c.load_state_dict(ck["state_dict"])
    opt_c = FusedAdam(net_c.parameters(), lr=1e-3, betas=(0.5, 0.9))
    opt_c.load_state_dict(osd)
    run(net_c, opt_c, range(3, 6))
    for p, q in zip(net_c.parameters(), net_a.parameters()):
        assert torch.equal(p, q)

    # the same file resumes the reference's optimizer: torch.optim.Adam on the CPU
    net_d = _net(77)
    net_d.load_state_dict(ck["state_dict"])
    opt_d = torch.optim.Adam(net_d.parameters(), lr=1e-3, betas=(0.5, 0.9))
    opt_d.load_state_dict(osd)
    for step in range(3, 6):
        x, y = _data(step)
        opt_d.zero_grad()
        (net_d(x) - y).pow(2).mean().backward()
        opt_d.step()
    for p, q in zip(net_d.parameters(), net_a.parameters()):
        assert torch.allclose(p.detach(), q.detach().cpu(), atol=2e-6)
    # ... and torch.optim.Adam's own state loads into FusedAdam
    net_e = _net(5).cuda()
    net_e.load_state_dict(net_d.state_dict())
    opt_e = FusedAdam(net_e.parameters(), lr=1e-3, betas=(0.5, 0.9))
    opt_e.load_state_dict(opt_d.state_dict())
    assert opt_e._counts == [6] * n_params
