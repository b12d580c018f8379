"""Size-independent properties of the oracle (hypothesis): invariances the domain offers, used as a second pin on the
delegated arithmetic and as the template for the full-size GPU checks."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle
from torchdriveenv_amd import _abi
from torchdriveenv_amd.state import EnvState

F = st.floats
pose = st.tuples(F(-50, 50), F(-50, 50), F(-math.pi, math.pi))


def box(p, L=4.6, W=2.0):
    return (p[0], p[1], math.cos(p[2]), math.sin(p[2]), L / 2, W / 2)


@settings(max_examples=300, deadline=None, derandomize=True, database=None)
@given(pose, pose, F(-100, 100), F(-100, 100), F(-math.pi, math.pi))
def test_sat_is_symmetric_and_rigid_motion_invariant_away_from_the_boundary(a, b, tx, ty, rot):
    r = oracle.obb_overlap(box(a), box(b))
    assert r == oracle.obb_overlap(box(b), box(a))
    # a clearly-overlapping or clearly-separated pair keeps its verdict under a rigid motion
    d = math.hypot(a[0] - b[0], a[1] - b[1])
    if d < 1.9 or d > 5.1:       # inside both incircles' reach / beyond both circumcircles
        def move(p):
            c, s = math.cos(rot), math.sin(rot)
            return (c * p[0] - s * p[1] + tx, s * p[0] + c * p[1] + ty, p[2] + rot)
        assert oracle.obb_overlap(box(move(a)), box(move(b))) == r == (1 if d < 1.9 else 0)


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(F(-200, 200), F(-200, 200), F(-math.pi, math.pi), F(-5, 25), F(1.4, 2.7), F(-1, 1), F(-0.3, 0.3))
def test_bicycle_step_properties(x, y, psi, v, lr, a, beta):
    nx, ny, npsi, nv = oracle.bicycle(x, y, psi, v, lr, a, beta)
    assert abs(nv - (v + a * 0.1)) < 1e-5
    assert abs(math.hypot(nx - np.float32(x), ny - np.float32(y)) - abs(nv) * 0.1) < 1e-3   # arc length = |v'| dt
    assert -math.pi - 1e-6 <= npsi < math.pi + 1e-6
    want = (math.pi + (psi + nv / lr * math.sin(beta) * 0.1)) % (2 * math.pi) - math.pi
    err = abs(npsi - want)
    assert min(err, 2 * math.pi - err) < 1e-4


@settings(max_examples=100, deadline=None, derandomize=True, database=None)
@given(F(-30, 30), F(-30, 30))
def test_point_triangle_distance_is_translation_consistent(px, py):
    tri = np.array([[0, 0, 10, 0, 0, 10]], np.float32)
    d0 = oracle.point_mesh_d2(px, py, tri)
    sh = np.array([[100, 50, 110, 50, 100, 60]], np.float32)
    d1 = oracle.point_mesh_d2(px + 100, py + 50, sh)
    assert abs(math.sqrt(d0) - math.sqrt(d1)) < 1e-3
    inside = px >= 0 and py >= 0 and px + py <= 10
    assert (d0 == 0.0) == inside or abs(px) < 1e-4 or abs(py) < 1e-4 or abs(px + py - 10) < 1e-4


def test_rollout_equals_repeated_steps_and_reward_bounds(small_world):
    cfg = _abi.default_config(seed=5, distance_cutoff=0.25)
    B, A, K = 48, 16, 230
    a, b = EnvState(B, A), EnvState(B, A)
    oracle.env_reset(cfg, small_world, a)
    oracle.env_reset(cfg, small_world, b)
    rng = np.random.default_rng(0)
    actions = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
    r, d = oracle.env_rollout(cfg, small_world, a, actions)
    for k in range(K):
        b["action"][...] = actions[k]
        oracle.env_step(cfg, small_world, b)
        assert np.array_equal(b["reward"].view(np.uint32), r[k].view(np.uint32))
        assert np.array_equal(b["terminated"] | (b["truncated"] << 1), d[k] & 3)
    ha, hb = a.host(), b.host()
    for k in ha:
        # (the Monitor-style episode statistics and done_bits belong to the closed-loop step API; a rollout leaves them
        #  alone: tde_abi.h, "tde_env_step only")
        if k not in ("action", "done_bits", "magnitudes") and not k.startswith("ep_"):
            assert np.array_equal(ha[k].view(np.uint8), hb[k].view(np.uint8)), k
    assert not ha["ep_return"].any() and hb["ep_final_len"].max() > 0 and not ha["magnitudes"].any()
    # reward = waypoint_bonus*[reach] + distance_bonus*[moved] - heading_penalty*(1 - cos dpsi)
    assert r.max() <= 101.0 + 1e-6 and r.min() >= -50.0 - 1e-6
    assert ((r > 50) == (r > 99 - 50)).all()
    # episodes never exceed max_steps, and every truncation happens exactly at max_steps
    assert a["steps"].max() <= cfg.max_steps
    # determinism: same seed, same result; different seed, different episodes
    c = EnvState(B, A)
    oracle.env_reset(cfg, small_world, c)
    r2, _ = oracle.env_rollout(cfg, small_world, c, actions)
    assert np.array_equal(r.view(np.uint32), r2.view(np.uint32))
    cfg3 = _abi.default_config(seed=6, distance_cutoff=0.25)
    e, f0 = EnvState(B, A), EnvState(B, A)
    oracle.env_reset(cfg3, small_world, e)
    oracle.env_reset(cfg, small_world, f0)
    assert not np.array_equal(e["x"], f0["x"])


def test_step_magnitudes_equal_the_ungated_operator(small_world):
    """tde_state.magnitudes, which the step fills only for the egos it flagged, equals tde_ego_infractions' brute force on EVERY
    env: a magnitude is non-zero only under its flag (collision: the mask's own predicate; offroad: a corner beyond the threshold
    has d^2 > thr^2 and sqrt(d^2) <= thr otherwise) - the gate never changes a value.  Both readings of the threshold."""
    from torchdriveenv_amd.synth import synthetic_world

    for squared, world in ((0, small_world), (1, synthetic_world(n_scn=8, A=16, seed=0, n_maps=2, threshold=float(np.sqrt(0.5))))):
        cfg = _abi.default_config(seed=9, flags=_abi.F_ALL & ~_abi.F_AUTORESET, terminated_at_infraction=0, max_steps=10_000,
                                  offroad_threshold_squared=squared)
        B, A = 64, 16
        hs = EnvState(B, A)
        oracle.env_reset(cfg, world, hs)
        rng = np.random.default_rng(squared)
        n_off = n_col = 0
        for t in range(60):
            hs["action"][...] = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            oracle.env_step(cfg, world, hs)
            want = oracle.ego_infractions(cfg, world, hs)
            assert np.array_equal(hs["magnitudes"].view(np.uint32), want.view(np.uint32)), t
            n_off += int((want[:, 0] > 0).sum()); n_col += int((want[:, 1] > 0).sum())
        assert n_off > 50 and n_col > 0


def test_first_step_npc_rule_known_answers_and_the_opt_out(small_world):
    """DESIGN R14 / tde_abi.h TDE_F_NPC_FIRST_STEP.  Without the flag (the opt-out): on the FIRST step of an episode the NPCs coast
    with the zero action (known answer, independent of any controller: v' = v bit for bit, x' = x + v cos(psi) dt, psi unchanged up
    to its wrap) and act from step two on.  With the flag (part of TDE_F_ALL since round 6: the default) the controller acts on step
    one too, as the reference's NPCs do (gym_env.py:285-294: IAIWrapper drives them from the first simulator.step) - checked against
    the batched-tensor restatement of the controller (oracle/torch_step.py: torch ops, another code path than the C loops) to 1e-5,
    and it DIFFERS from coasting for the NPCs that have a route."""
    assert _abi.F_ALL & _abi.F_NPC_FIRST_STEP                         # the default is the reference's timing
    from oracle.torch_step import TorchWorld, torch_env_step

    B, A = 48, 16
    tw = TorchWorld(small_world)
    for flag in (0, _abi.F_NPC_FIRST_STEP):
        cfg = _abi.default_config(seed=6, flags=(_abi.F_ALL & ~_abi.F_AUTORESET & ~_abi.F_NPC_FIRST_STEP) | flag)
        hs, ht = EnvState(B, A), EnvState(B, A)
        oracle.env_reset(cfg, small_world, hs)
        ht.load(hs.host())
        pre = hs.host()
        act = np.zeros((B, 2), np.float32)
        hs["action"][...] = act; ht["action"][...] = act
        oracle.env_step(cfg, small_world, hs)
        torch_env_step(cfg, small_world, tw, ht)
        npc = (np.arange(B * A) % A != 0) & (pre["present"] != 0)
        routed = npc & (small_world.arrays["spawn"].reshape(-1, A)["route"][pre["scn"]].reshape(-1) >= 0)
        replayed = npc & (small_world.arrays["spawn"].reshape(-1, A)["replay"][pre["scn"]].reshape(-1) >= 0)
        free = npc & ~replayed
        for k in ("x", "y", "psi", "v"):
            assert np.abs(hs[k] - ht[k]).max() <= 1e-5 * max(1.0, np.abs(pre[k]).max()), (flag, k)
        if not flag:
            assert np.array_equal(hs["v"][free].view(np.uint32), pre["v"][free].view(np.uint32))          # coasting: v' = v
            dpsi = np.abs(hs["psi"][free] - pre["psi"][free])                                              # (the wrap (pi + psi) % 2 pi - pi
            assert np.minimum(dpsi, 2 * np.pi - dpsi).max() < 1e-6                                          #  re-rounds the heading)
            dt = np.float32(0.1)
            s, c = oracle.sincosf(pre["psi"])
            assert np.array_equal((pre["x"] + (pre["v"] * c) * dt)[free].view(np.uint32), hs["x"][free].view(np.uint32))
            coast_v = hs["v"].copy()
        else:
            moved = routed & ~replayed
            assert moved.sum() > 100 and (hs["v"][moved] != coast_v[moved]).mean() > 0.5                    # the controller acted
        # from step two on the two rules agree on WHAT runs (the controller), not on the state it runs from
        oracle.env_step(cfg, small_world, hs)
        assert (hs["v"][free] != pre["v"][free]).mean() > 0.3
