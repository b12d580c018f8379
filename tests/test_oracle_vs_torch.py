"""The north star's bar restated: kinematic state within 1e-5 of a CPU PyTorch fp32 step.  torchdrivesim is not
installed, so the torch side is a literal torch-op restatement of its published KinematicBicycle.step
(a, beta = action; v += a*dt; x += v*cos(psi+beta)*dt; y += v*sin(psi+beta)*dt; psi += v/lr*sin(beta)*dt;
psi = (pi + psi) % (2*pi) - pi) using torch.sin / torch.cos / torch's % — i.e. libm-class transcendentals instead of
the oracle's own polynomial.  Collision / offroad masks are compared on samples away from the decision boundary."""
import math

import numpy as np
import torch

from oracle import oracle

DT = 0.1


def torch_bicycle(state, lr, action):
    x, y, psi, v = state.unbind(-1)
    a, beta = action.unbind(-1)
    v = v + a * DT
    x = x + v * torch.cos(psi + beta) * DT
    y = y + v * torch.sin(psi + beta) * DT
    psi = psi + (v / lr) * torch.sin(beta) * DT
    psi = (np.pi + psi) % (2 * np.pi) - np.pi
    return torch.stack([x, y, psi, v], -1)


def test_kinematic_state_within_1e5_of_torch_fp32():
    g = torch.Generator().manual_seed(0)
    n = 50_000
    st = torch.stack([torch.rand(n, generator=g) * 400 - 200, torch.rand(n, generator=g) * 400 - 200,
                      torch.rand(n, generator=g) * 2 * math.pi - math.pi, torch.rand(n, generator=g) * 25], -1)
    lr = torch.rand(n, generator=g) * 1.1 + 1.46
    # teacher-forced single steps: both sides start every step from the same state
    for _ in range(5):
        act = torch.stack([torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 0.6 - 0.3], -1)
        want = torch_bicycle(st, lr, act)
        cols = [st[:, k].numpy().copy() for k in range(4)]
        oracle.kinematics_step(*cols, lr.numpy().copy(), None, act.numpy().copy(), DT)
        got = np.stack(cols, -1)
        diff = np.abs(got - want.numpy())
        dpsi = diff[:, 2]
        diff[:, 2] = np.minimum(dpsi, 2 * math.pi - dpsi)          # the wrap point itself may land on either side
        assert diff.max() <= 1e-5 * max(1.0, float(want.abs().max())), diff.max()
        assert diff[:, 3].max() == 0.0                               # v' = v + a*dt has no transcendental: bit-exact
        st = want
    # free 100-step rollout: the two implementations drift apart only by accumulated rounding
    s_t, s_o = st.clone(), [st[:, k].numpy().copy() for k in range(4)]
    for _ in range(100):
        act = torch.stack([torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 0.6 - 0.3], -1)
        s_t = torch_bicycle(s_t, lr, act)
        oracle.kinematics_step(*s_o, lr.numpy().copy(), None, act.numpy().copy(), DT)
    pos_err = np.hypot(s_o[0] - s_t[:, 0].numpy(), s_o[1] - s_t[:, 1].numpy())
    assert pos_err.max() < 5e-3 and np.median(pos_err) < 2e-4


def torch_sat(b0, b1):
    """strict SAT overlap of boxes (x, y, psi, L, W) in torch fp32, batched"""
    def axes(b):
        c, s = torch.cos(b[:, 2]), torch.sin(b[:, 2])
        return torch.stack([c, s], -1), torch.stack([-s, c], -1)
    d = b1[:, :2] - b0[:, :2]
    u0, n0 = axes(b0)
    u1, n1 = axes(b1)
    h0, h1 = b0[:, 3:] * 0.5, b1[:, 3:] * 0.5
    sep = torch.zeros(len(b0), dtype=torch.bool)
    gap = torch.full((len(b0),), -1e9)
    for ax in (u0, n0, u1, n1):
        r0 = h0[:, 0] * (ax * u0).sum(-1).abs() + h0[:, 1] * (ax * n0).sum(-1).abs()
        r1 = h1[:, 0] * (ax * u1).sum(-1).abs() + h1[:, 1] * (ax * n1).sum(-1).abs()
        g = (d * ax).sum(-1).abs() - (r0 + r1)
        gap = torch.maximum(gap, g)
    return gap < 0, gap


def test_collision_mask_matches_torch_away_from_the_boundary():
    g = torch.Generator().manual_seed(1)
    n = 20_000
    b0 = torch.stack([torch.rand(n, generator=g) * 10, torch.rand(n, generator=g) * 10,
                      torch.rand(n, generator=g) * 6.28 - 3.14, torch.rand(n, generator=g) * 2 + 4,
                      torch.rand(n, generator=g) + 1.7], -1)
    b1 = b0.clone()
    b1[:, 0] += torch.rand(n, generator=g) * 12 - 6
    b1[:, 1] += torch.rand(n, generator=g) * 12 - 6
    b1[:, 2] = torch.rand(n, generator=g) * 6.28 - 3.14
    want, gap = torch_sat(b0, b1)
    x = torch.stack([b0[:, 0], b1[:, 0]], -1).reshape(-1).numpy().copy()
    y = torch.stack([b0[:, 1], b1[:, 1]], -1).reshape(-1).numpy().copy()
    psi = torch.stack([b0[:, 2], b1[:, 2]], -1).reshape(-1).numpy().copy()
    L = torch.stack([b0[:, 3], b1[:, 3]], -1).reshape(-1).numpy().copy()
    W = torch.stack([b0[:, 4], b1[:, 4]], -1).reshape(-1).numpy().copy()
    got = oracle.compute_collision(n, 2, x, y, psi, L, W, np.ones(2 * n, np.uint8)).reshape(n, 2)
    assert (got[:, 0] == got[:, 1]).all()
    clear = gap.abs().numpy() > 1e-4                       # knife-edge pairs may legitimately differ by rounding
    assert np.array_equal(got[clear, 0].astype(bool), want.numpy()[clear])
    assert clear.mean() > 0.999 and 0.2 < want.float().mean() < 0.8
