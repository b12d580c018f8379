"""Second opinions for the rows of SURVEY 8(a) that no reference vector can pin (torchdrivesim is absent: R9 collision,
R10 offroad, R12's device-dependent smoothness terms).  Each test restates the PUBLISHED form of the upstream metric with
independent arithmetic (float64 polygon clipping, torch fp32 distances) and checks that the oracle's decision is the same
outside a thin band around the decision boundary, where upstream's own rounding decides.

  * R9   CollisionMetric.nograd (gym_env.py:48, 143): exact IoU of the two oriented rectangles, `> 0`.  Restated here by
         Sutherland-Hodgman clipping of one rectangle by the other's four edges + the shoelace formula; the oracle uses a
         strict separating-axis test.  Positive-area overlap <=> no separating axis.
  * R10  compute_offroad (gym_env.py:142): sum over the four corners of clamp(dist - threshold, min=0), `> 0`, under both
         readings of `dist` (Euclidean, or pytorch3d's squared point-mesh distance).
  * R12  psi_smoothness / speed_smoothness (gym_env.py:432, 435): `(a - b) / 0.1` on fp32 tensors is a true division on
         torch's CPU path (what tests/golden was captured with) and a multiplication by fl32(1 / 0.1f) = 10.0f on CUDA
         (ATen's BinaryDivTrue kernel special-cases a CPU-scalar divisor).  The two agree except for an occasional last bit."""
import json
import math
import os

import numpy as np
import pytest
import torch
from hypothesis import example, given, settings
from hypothesis import strategies as st

from oracle import oracle
from oracle.torch_step import point_mesh_d2 as torch_point_mesh_d2
from tests.golden_util import case_expected, case_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------------
# R9: polygon-clip IoU
# ---------------------------------------------------------------------------------------------------------------------
def corners(x, y, psi, L, W):
    c, s = math.cos(psi), math.sin(psi)
    hl, hw = 0.5 * L, 0.5 * W
    return [(x + hl * c - hw * s, y + hl * s + hw * c), (x - hl * c - hw * s, y - hl * s + hw * c),
            (x - hl * c + hw * s, y - hl * s - hw * c), (x + hl * c + hw * s, y + hl * s - hw * c)]      # counter-clockwise


def clip(poly, a, b):
    """Sutherland-Hodgman: the part of `poly` on the left of the directed edge a -> b"""
    def side(p):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
    out = []
    for i, p in enumerate(poly):
        q = poly[(i + 1) % len(poly)]
        sp, sq = side(p), side(q)
        if sp >= 0:
            out.append(p)
        if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def area(poly):
    return 0.5 * abs(sum(p[0] * q[1] - q[0] * p[1] for p, q in zip(poly, poly[1:] + poly[:1]))) if len(poly) >= 3 else 0.0


def iou(b0, b1):
    p0, p1 = corners(*b0), corners(*b1)
    inter = p0
    for i in range(4):
        if not inter:
            break
        inter = clip(inter, p1[i], p1[(i + 1) % 4])
    ai = area(inter)
    return ai / (b0[3] * b0[4] + b1[3] * b1[4] - ai)


def sat_gap(b0, b1):
    """largest separation over the four candidate axes in float64 (< 0: overlap, the overlap depth)"""
    def ax(b):
        c, s = math.cos(b[2]), math.sin(b[2])
        return (c, s), (-s, c)
    d = (b1[0] - b0[0], b1[1] - b0[1])
    u0, n0 = ax(b0)
    u1, n1 = ax(b1)
    g = -1e9
    for a in (u0, n0, u1, n1):
        r0 = 0.5 * b0[3] * abs(a[0] * u0[0] + a[1] * u0[1]) + 0.5 * b0[4] * abs(a[0] * n0[0] + a[1] * n0[1])
        r1 = 0.5 * b1[3] * abs(a[0] * u1[0] + a[1] * u1[1]) + 0.5 * b1[4] * abs(a[0] * n1[0] + a[1] * n1[1])
        g = max(g, abs(d[0] * a[0] + d[1] * a[1]) - (r0 + r1))
    return g


def oracle_collides(b0, b1):
    f = np.float32
    arr = lambda k: np.array([b0[k], b1[k]], f)                                                   # noqa: E731
    got = oracle.compute_collision(1, 2, arr(0), arr(1), arr(2), arr(3), arr(4), np.ones(2, np.uint8))
    assert got[0] == got[1]
    return bool(got[0])


box = st.tuples(st.floats(-8, 8, width=32), st.floats(-8, 8, width=32), st.floats(-3.25, 3.25, width=32),
                st.floats(3.75, 7.0, width=32), st.floats(1.5, 3.125, width=32))


@settings(max_examples=400, deadline=None, derandomize=True, database=None)
@given(box, box)
@example((0.0, 0.0, 0.0, 4.0, 2.0), (4.0, 0.0, 0.0, 4.0, 2.0))                    # touching end to end: IoU == 0, no collision
@example((0.0, 0.0, 0.0, 4.0, 2.0), (0.0, 2.0, 0.0, 4.0, 2.0))                    # touching side by side
@example((0.0, 0.0, 0.3, 6.0, 3.0), (0.2, 0.1, 1.1, 3.9, 1.7))                    # one box inside the other
@example((0.0, 0.0, 0.0, 4.0, 2.0), (0.0, 0.0, 0.0, 4.0, 2.0))                    # identical
@example((0.0, 0.0, 0.0, 4.0, 2.0), (4.0, 0.0, math.pi / 4, 4.0, 2.0))            # a rotated corner pokes in
def test_iou_positive_iff_oracle_sat_overlap(b0, b1):
    # the oracle sees fp32 poses; the float64 restatement takes the same fp32 values
    b0 = tuple(float(np.float32(v)) for v in b0)
    b1 = tuple(float(np.float32(v)) for v in b1)
    gap = sat_gap(b0, b1)
    got = oracle_collides(b0, b1)
    if gap == 0.0:                     # exactly touching (representable): zero-area contact is not a collision on either side
        assert iou(b0, b1) < 1e-12 and got is False        # (the float64 clip of a rotated touching pair may leave rounding dust)
        return
    if abs(gap) < 1e-4:                # the knife edge: fp32 rounding of either implementation may decide
        return
    assert (iou(b0, b1) > 0.0) == got == (gap < 0.0)


def test_iou_vs_sat_on_a_dense_sample_around_contact():
    """20 000 pairs concentrated around first contact: outside a 1e-4 m band IoU > 0 <=> the oracle's mask, and the IoU of
    a barely overlapping pair is of the order of overlap depth x contact length / area (no spurious zeros)"""
    rng = np.random.default_rng(2)
    n = 20_000
    b0 = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-np.pi, np.pi, n),
                   rng.uniform(3.8, 6.9, n), rng.uniform(1.6, 3.1, n)], -1).astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, n)
    dist = rng.uniform(1.5, 7.5, n)
    b1 = np.stack([b0[:, 0] + dist * np.cos(ang), b0[:, 1] + dist * np.sin(ang), rng.uniform(-np.pi, np.pi, n),
                   rng.uniform(3.8, 6.9, n), rng.uniform(1.6, 3.1, n)], -1).astype(np.float32)
    il = lambda k: np.stack([b0[:, k], b1[:, k]], -1).reshape(-1).copy()                          # noqa: E731
    got = oracle.compute_collision(n, 2, il(0), il(1), il(2), il(3), il(4), np.ones(2 * n, np.uint8)).reshape(n, 2)[:, 0]
    checked = agree = 0
    for i in range(n):
        p, q = tuple(map(float, b0[i])), tuple(map(float, b1[i]))
        g = sat_gap(p, q)
        if abs(g) < 1e-4:
            continue
        checked += 1
        v = iou(p, q)
        agree += (v > 0.0) == bool(got[i]) == (g < 0.0)
        if g < -1e-3:
            assert v > 1e-9            # (a corner 1 mm deep: ~depth^2 / area)
    assert agree == checked and checked > 0.999 * n and 0.25 < got.mean() < 0.75


# ---------------------------------------------------------------------------------------------------------------------
# R10: sum_corners clamp(dist - thr, 0) > 0, both readings
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("squared", [False, True])
def test_published_offroad_form_matches_the_oracle_mask(small_world, squared):
    w = small_world
    rng = np.random.default_rng(13)
    B, A = 400, 4
    n = B * A
    m = w.arrays["maps"][0]
    tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
    s = rng.uniform(-120, 120, n)
    lat = rng.choice([0.0, 1.75, 2.4, 2.9, 3.2, 3.6, 4.2, 6.0], n) * rng.choice([-1, 1], n)
    x, y = (s + rng.normal(0, 0.2, n)).astype(np.float32), (lat + rng.normal(0, 0.2, n)).astype(np.float32)
    psi = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    L = rng.uniform(3.8, 6.9, n).astype(np.float32)
    W = rng.uniform(1.7, 3.0, n).astype(np.float32)
    thr = np.float32(0.5)
    # the published form in torch fp32: corners from torch.cos / torch.sin (libm class), distance to the mesh per corner
    tx, ty, tp = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(psi)
    c, s_ = torch.cos(tp), torch.sin(tp)
    hl, hw = torch.from_numpy(L) * 0.5, torch.from_numpy(W) * 0.5
    cx = torch.stack([tx + hl * c - hw * s_, tx + hl * c + hw * s_, tx - hl * c + hw * s_, tx - hl * c - hw * s_], -1)
    cy = torch.stack([ty + hl * s_ + hw * c, ty + hl * s_ - hw * c, ty - hl * s_ - hw * c, ty - hl * s_ + hw * c], -1)
    d2 = torch_point_mesh_d2(torch.stack([cx.reshape(-1), cy.reshape(-1)], -1), torch.from_numpy(tri.reshape(-1, 3, 2))).reshape(n, 4)
    dist = d2 if squared else d2.sqrt()
    loss = (dist - float(thr)).clamp(min=0.0).sum(-1)                      # what info["offroad"] holds upstream
    want = (loss > 0).numpy()
    # margin of the decision: how far the deciding corner is from the threshold
    margin = (dist - float(thr)).abs().min(-1).values.numpy()
    eff = float(np.sqrt(thr)) if squared else float(thr)                  # the oracle operator squares what it is given
    got = oracle.compute_offroad(B, A, x, y, psi, L, W, np.ones(n, np.uint8), w, np.zeros(B, np.int32), threshold=eff)
    clear = margin > 2e-4
    assert np.array_equal(got.astype(bool)[clear], want[clear])
    assert clear.mean() > 0.99 and 0.2 < want.mean() < 0.8


# ---------------------------------------------------------------------------------------------------------------------
# R12: (a - b) / 0.1 on the CPU path vs * fl32(1 / 0.1f) on the CUDA path
# ---------------------------------------------------------------------------------------------------------------------
def test_smoothness_terms_cpu_division_vs_cuda_reciprocal_multiply():
    """tests/golden/reward_golden.json was captured with torch_device = cpu (oracle/gen_golden.py): psi_smoothness and
    speed_smoothness are |(last - now) / 0.1f| there - a true fp32 division, which is what oracle and kernels evaluate.  A
    reference run on CUDA multiplies by fl32(1.0f / 0.1f) = 10.0f instead.  On the fixture's own inputs the two agree on
    most steps and differ by exactly one ulp on the rest: info-only terms (not in reward / termination), so a CUDA-run
    reference matches this build up to that last bit."""
    with open(os.path.join(ROOT, "tests", "golden", "reward_golden.json")) as f:
        golden = json.load(f)
    inv = np.float32(1.0) / np.float32(0.1)
    assert inv == np.float32(10.0)
    n = differ = 0
    for case in golden["cases"]:
        i = case_inputs(case)
        pre, post = i["pre"], i["post"]
        e = case_expected(case)
        # (the fixture itself holds the CPU form: the division)
        assert np.array_equal(np.abs((pre[:, 2] - post[:, 2]) / np.float32(0.1)).astype(np.float64), e["info"][:, 0])
        for col in (2, 3):
            d = pre[:, col] - post[:, col]
            cpu = np.abs(d / np.float32(0.1))
            cuda = np.abs(d * inv)
            ulp = np.spacing(np.maximum(cpu, cuda))
            assert np.all(np.abs(cpu.astype(np.float64) - cuda.astype(np.float64)) <= ulp)
            n += len(d)
            differ += int((cpu != cuda).sum())
    assert n > 2000 and 0 < differ < 0.5 * n


def test_oracle_box_iou_matches_the_float64_clip():
    """the fp32 IoU the oracle (and the kernel) sum into the collision MAGNITUDE of info["collision"] - CollisionMetric.nograd's
    published form - against the float64 Sutherland-Hodgman construction above, 20 000 random pairs"""
    import ctypes as C

    L = oracle.lib()
    L.tde_oracle_box_iou.argtypes = [C.c_float] * 12
    L.tde_oracle_box_iou.restype = C.c_float
    rng = np.random.default_rng(0)
    f = np.float32
    worst, npos = 0.0, 0
    for _ in range(20_000):
        b = [tuple(float(f(v)) for v in (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-3.1, 3.1), rng.uniform(3.8, 6.9),
                                          rng.uniform(1.6, 3.1))) for _ in range(2)]
        args = []
        for q in b:
            args += [q[0], q[1], f(math.cos(q[2])), f(math.sin(q[2])), f(0.5) * f(q[3]), f(0.5) * f(q[4])]
        v, w = L.tde_oracle_box_iou(*args), iou(b[0], b[1])
        worst = max(worst, abs(v - w))
        npos += w > 0
        assert 0.0 <= v <= 1.0 + 1e-6
    assert worst < 2e-6 and 10_000 < npos < 19_000
