"""Known-answer tests that pin the oracle's delegated arithmetic: trig accuracy, bicycle closed forms,
SAT edge cases, point/triangle distances, Philox KAT, and grid index == brute force."""
import math

import numpy as np

from oracle import oracle
from torchdriveenv_amd import _abi


def test_sincos_within_2ulp_of_libm():
    x = np.concatenate([np.linspace(-8, 8, 400001), np.linspace(-1e-3, 1e-3, 2001), [0.0, np.pi, -np.pi, np.pi / 2]])
    x = x.astype(np.float32)
    s, c = oracle.sincosf(x)
    for got, ref in ((s, np.sin(x.astype(np.float64))), (c, np.cos(x.astype(np.float64)))):
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - ref) / ulp) <= 2.0
        assert np.max(np.abs(got - ref)) < 1.5e-7
    assert oracle.sincosf(np.zeros(1, np.float32))[0][0] == 0.0 and oracle.sincosf(np.zeros(1, np.float32))[1][0] == 1.0


def test_bicycle_straight_line_and_rest():
    x, y, psi, v = oracle.bicycle(1.0, 2.0, 0.0, 5.0, 1.9, 0.0, 0.0)
    assert (x, y, psi, v) == (np.float32(1.0) + np.float32(5.0) * np.float32(0.1), 2.0, 0.0, 5.0)
    assert oracle.bicycle(3.0, -4.0, 0.7, 0.0, 1.9, 0.0, 0.2)[:2] == (3.0, -4.0)       # zero speed: pose constant
    x, y, psi, v = oracle.bicycle(0.0, 0.0, 0.0, 2.0, 1.9, 1.0, 0.0)                   # semi-implicit: uses v'
    assert abs(v - 2.1) < 1e-6 and abs(x - 0.21) < 1e-6


def test_bicycle_constant_steer_is_a_circle_of_radius_lr_over_sin_beta():
    lr, beta, v = 1.9, 0.2, 4.0
    R = lr / math.sin(beta)
    st = (0.0, 0.0, 0.0, v)
    pts = []
    for _ in range(400):
        st = oracle.bicycle(*st, lr, 0.0, beta)
        pts.append(st[:2])
        assert -math.pi - 1e-6 <= st[2] < math.pi + 1e-6                                 # wrap to [-pi, pi)
    pts = np.asarray(pts)
    # fit circle centre: the motion starts at the origin heading (cos b, sin b)
    cx, cy = -R * math.sin(beta), R * math.cos(beta)
    r = np.hypot(pts[:, 0] - cx, pts[:, 1] - cy)
    assert np.max(np.abs(r - R)) < 0.03 * R


def test_heading_wraps_like_torch_remainder():
    _, _, psi, _ = oracle.bicycle(0.0, 0.0, 3.1, 10.0, 1.5, 0.0, 0.3)
    want = (math.pi + (3.1 + 10.0 / 1.5 * math.sin(0.3) * 0.1)) % (2 * math.pi) - math.pi
    assert abs(psi - want) < 1e-5 and psi < 0


def test_sat_known_answers():
    def box(x, y, psi, L=4.0, W=2.0):
        return (x, y, math.cos(psi), math.sin(psi), L / 2, W / 2)
    assert oracle.obb_overlap(box(0, 0, 0), box(0, 0, 0)) == 1
    assert oracle.obb_overlap(box(0, 0, 0), box(4, 0, 0)) == 0            # edge touching: no collision
    assert oracle.obb_overlap(box(0, 0, 0), box(3.999, 0, 0)) == 1
    assert oracle.obb_overlap(box(0, 0, 0), box(0, 2, 0)) == 0
    assert oracle.obb_overlap(box(0, 0, 0), box(4.0, 0, math.pi / 4)) == 1  # rotated corner pokes in
    assert oracle.obb_overlap(box(0, 0, 0), box(4.2, 0, math.pi / 4)) == 0
    assert oracle.obb_overlap(box(0, 0, 0), box(100, 100, 1)) == 0
    rng = np.random.default_rng(0)
    for _ in range(2000):                                                    # symmetric
        a = box(*rng.uniform(-4, 4, 2), rng.uniform(-3, 3))
        b = box(*rng.uniform(-4, 4, 2), rng.uniform(-3, 3))
        assert oracle.obb_overlap(a, b) == oracle.obb_overlap(b, a)


def test_point_mesh_distance_known_answers():
    tri = np.array([[0, 0, 10, 0, 0, 10]], np.float32)
    assert oracle.point_mesh_d2(1, 1, tri) == 0.0                            # inside
    assert oracle.point_mesh_d2(5, 0, tri) == 0.0                            # on an edge
    assert abs(oracle.point_mesh_d2(5, -0.5, tri) - 0.25) < 1e-7             # 0.5 m below the edge
    assert abs(oracle.point_mesh_d2(-3, -4, tri) - 25.0) < 1e-5              # nearest feature = vertex
    assert oracle.point_mesh_d2(5, -0.5 - 1e-4, tri) > 0.25 > oracle.point_mesh_d2(5, -0.5 + 1e-4, tri)


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, counter 0, key 0 / all-ones / pi digits
    assert oracle.philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox((0x299f31d0 << 32) | 0xa4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def grid_offroad_numpy(world, map_id, px, py, thr, use_sub=False):
    """float32 emulation of the kernel's cell lookup + candidate test (tde_device.h: cell_lookup / box_offroad)"""
    f = np.float32
    m = world.arrays["maps"][map_id]
    words, recs = world.arrays["cell_word"], world.arrays["cell_tri"]
    out = np.zeros(len(px), bool)
    for i, (x, y) in enumerate(zip(px.astype(f), py.astype(f))):
        fx, fy = f((x - m["ox"]) * m["inv_cell"]), f((y - m["oy"]) * m["inv_cell"])
        if not (fx >= 0 and fy >= 0 and fx < m["nx"] and fy < m["ny"]):
            out[i] = True
            continue
        ix, iy = int(fx), int(fy)
        wd = int(words[m["cell_base"] + (iy << m["row_shift"]) + ix])
        cls = wd & 3
        if cls != _abi.CELL_MIXED:
            out[i] = cls == _abi.CELL_EMPTY
            continue
        if use_sub:                                        # sub-cell classes of the MIXED cell (world.py: subcell_classes)
            tile = ((iy >> 2) << (int(m["row_shift"]) - 3)) + (ix >> 3)
            bm = int(world.arrays["cell_sub"][m["cell_base"] + ((tile << 5) | ((iy & 3) << 3) | (ix & 7))])
            sx, sy = min(int((fx - f(ix)) * f(4)), 3), min(int((fy - f(iy)) * f(4)), 3)
            sc = (bm >> (2 * (4 * sy + sx))) & 3
            if sc != _abi.CELL_MIXED:
                out[i] = sc == _abi.CELL_EMPTY
                continue
        ok = False
        first = int(m["rec_base"]) + (wd >> 10)         # record offsets count from the map's rec_base (ABI 9)
        for k in range(first, first + ((wd >> 2) & 255)):
            if oracle.point_mesh_d2(x, y, recs[k, :6]) <= f(thr) * f(thr):
                ok = True
                break
        out[i] = not ok
    return out


def test_grid_index_equals_brute_force(small_world):
    w = small_world
    rng = np.random.default_rng(7)
    for map_id in range(w.ints["n_maps"]):
        m = w.arrays["maps"][map_id]
        tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
        n = 6000
        # points concentrated around the road edges (|lateral| ~ 3.5 +- 0.5 +- eps) plus uniform ones
        s = rng.uniform(-135, 135, n)
        lat = (3.5 + 0.5 + rng.normal(0, 0.02, n)) * rng.choice([-1, 1], n)
        px, py = s.copy(), lat.copy()
        u = rng.uniform(size=n) < 0.4
        px[u], py[u] = rng.uniform(-150, 150, u.sum()), rng.uniform(-150, 150, u.sum())
        want = np.array([oracle.point_mesh_d2(x, y, tri) > np.float32(0.5) * np.float32(0.5)
                         for x, y in zip(px.astype(np.float32), py.astype(np.float32))])
        got = grid_offroad_numpy(w, map_id, px, py, 0.5)
        assert np.array_equal(got, want)
        got = grid_offroad_numpy(w, map_id, px, py, 0.5, use_sub=True)     # what the rasteriser does in MIXED cells
        assert np.array_equal(got, want)
        assert 0.2 < want.mean() < 0.95


def _edge_points(tri, n, rng, spread=0.6):
    """points scattered around the mesh's vertices and edge midpoints (where the offroad predicate flips) plus a few far ones"""
    t = np.asarray(tri, np.float64).reshape(-1, 3, 2)
    k = rng.integers(len(t), size=n)
    a, b = t[k, rng.integers(3, size=n)], t[k, rng.integers(3, size=n)]
    p = a + (b - a) * rng.uniform(size=(n, 1)) + rng.normal(0, spread, (n, 2))
    far = rng.uniform(size=n) < 0.1
    lo, hi = t.reshape(-1, 2).min(0) - 5, t.reshape(-1, 2).max(0) + 5
    p[far] = rng.uniform(lo, hi, (int(far.sum()), 2))
    return p[:, 0].astype(np.float32), p[:, 1].astype(np.float32)


def test_near_mesh_predicate_equals_brute_force_minimum(small_world, small_town):
    """the oracle's offroad predicate (bounding-box reject, early exit) == min over EVERY triangle of d2 <= thr2, for both
    readings of the threshold, on points around the road edges of a junction map and of the town"""
    rng = np.random.default_rng(11)
    for w, n in ((small_world, 3000), (small_town, 1500)):
        m = w.arrays["maps"][0]
        tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
        px, py = _edge_points(tri, n, rng)
        for thr2 in (np.float32(0.25), np.float32(0.5)):
            want = np.array([oracle.point_mesh_d2(x, y, tri) <= thr2 for x, y in zip(px, py)])
            got = np.array([oracle.point_near_mesh(x, y, tri, thr2) for x, y in zip(px, py)])
            assert np.array_equal(got, want)
            assert 0.15 < want.mean() < 0.95


def test_town_grid_index_equals_brute_force(small_town):
    """the natively built index (tde_grid_build: shared candidate lists, per-map record base) of a warped street grid gives
    the brute-force mask, through the cell words alone and through the sub-cell classes"""
    w = small_town
    m = w.arrays["maps"][0]
    tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
    px, py = _edge_points(tri, 5000, np.random.default_rng(5))
    want = np.array([oracle.point_mesh_d2(x, y, tri) > np.float32(0.25) for x, y in zip(px, py)])
    assert np.array_equal(grid_offroad_numpy(w, 0, px, py, 0.5), want)
    assert np.array_equal(grid_offroad_numpy(w, 0, px, py, 0.5, use_sub=True), want)
    assert 0.1 < want.mean() < 0.9


def near_list_d2_numpy(w, map_id, px, py):
    """what the magnitude kernels do with the near lists (csrc/tde_magnitudes.h: ego_offroad_mag_wave), restated in numpy on the
    world's tables: the corner's coarse tile -> tile_near word -> the list's records (its length in the first record) -> the minimum
    of the CPU checker's point-triangle distance over the list.  Returns (d2, code) with code 0 = no list, 1 = all FULL, 2 = listed"""
    m = w.arrays["maps"][map_id]
    fx = np.clip((np.float32(px) - m["ox"]) * m["inv_cell"], 0, m["nx"] - 1)
    fy = np.clip((np.float32(py) - m["oy"]) * m["inv_cell"], 0, m["ny"] - 1)
    ix, iy = int(fx), int(fy)
    inside = m["ox"] <= px < m["ox"] + np.float32(m["nx"]) * m["cell"] and m["oy"] <= py < m["oy"] + np.float32(m["ny"]) * m["cell"]
    tw = int(w.arrays["tile_near"][m["near_base"] + (iy // 4) * (m["nx"] // 4) + ix // 4]) if inside else 0
    if tw == 0:
        return None, 0 if inside else 3
    if tw == 0xFFFFFFFF:
        return None, 1
    recs = w.arrays["cell_tri"].reshape(-1, 12)
    first = m["rec_base"] + tw - 1
    n = int(recs[first, 9:10].view(np.int32)[0])
    assert n >= 1
    tris = recs[first:first + n, :6]
    return min(oracle.point_mesh_d2(px, py, t[None]) for t in tris), 2


def test_near_lists_give_the_brute_force_distance(small_world, small_town):
    """ABI 10: the minimum of the point-triangle distances over a coarse tile's NEAR LIST equals the CPU checker's minimum over
    EVERY triangle of the map, bit for bit, for points anywhere in the tile - the magnitude of the offroad infraction
    (gym_env.py:427) reads two table entries instead of scanning the grid.  Tiles coded 'all FULL' hold only points within the
    threshold; tiles without a list lie farther than threshold + near_range from the mesh."""
    from torchdriveenv_amd.world import NEAR_RANGE

    rng = np.random.default_rng(5)
    for w, n in ((small_world, 2500), (small_town, 1500)):
        for map_id in range(min(2, w.ints["n_maps"])):
            m = w.arrays["maps"][map_id]
            tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]]
            px, py = _edge_points(tri, n, rng, spread=1.2)
            codes = np.zeros(4, int)                       # (code 3: outside the grid - the kernels scan)
            for x, y in zip(px, py):
                want = oracle.point_mesh_d2(x, y, tri)
                got, code = near_list_d2_numpy(w, map_id, x, y)
                codes[code] += 1
                if code == 2:
                    assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (x, y, got, want)
                elif code == 1:
                    assert want <= np.float32(0.5) * np.float32(0.5)
                elif code == 0:
                    assert np.sqrt(want) > 0.5 + NEAR_RANGE - 1.0        # (a tile's centre decides: a point of it may be a tile diagonal closer)
            assert codes[2] > 0.1 * n and codes[1] > 0 and codes[0] > 0, codes
    # every list's length rides in its first record, and the table ends with 16 spare records
    recs = small_town.arrays["cell_tri"].reshape(-1, 12)
    assert (recs[-16:] == 0).all()
    lens = recs[:, 9].view(np.int32)
    assert lens.max() < 200 and (lens >= 0).all()


def test_near_lists_on_random_triangle_soups():
    """the near lists' guarantee does not lean on road-like meshes: random triangle soups (overlapping, slivers, a degenerate
    triangle, isolated islands), random points - wherever a tile carries a list, the minimum over the list is the brute-force
    distance bit for bit; an all-FULL tile only holds points within the threshold; a tile without a list is far from every
    triangle"""
    from torchdriveenv_amd.world import build_grid_index

    rng = np.random.default_rng(42)
    for trial in range(6):
        n = int(rng.integers(8, 60))
        c = rng.uniform(-20, 20, (n, 1, 2))
        tri = (c + rng.normal(0, rng.uniform(0.3, 4.0, (n, 1, 1)), (n, 3, 2))).astype(np.float32)
        tri[0, 2] = tri[0, 1]                                  # a degenerate triangle (two equal vertices)
        tri[1] = np.array([[0, 0], [6, 0.001], [12, 0]], np.float32) + rng.uniform(-5, 5, 2).astype(np.float32)   # a sliver
        near_range = float(rng.choice([0.5, 2.0, 4.0]))
        g = build_grid_index(tri, threshold=0.5, cell=float(rng.choice([0.25, 0.5])), near_range=near_range, n_threads=2)
        flat = tri.reshape(-1, 6)
        ntx = g["nx"] // 4
        pts = np.concatenate([rng.uniform(-26, 26, (250, 2)), (c[rng.integers(n, size=250), 0] + rng.normal(0, 2.5, (250, 2)))]).astype(np.float32)
        seen = np.zeros(3, int)
        for x, y in pts:
            ix, iy = int((x - g["ox"]) / g["cell"]), int((y - g["oy"]) / g["cell"])
            if not (0 <= ix < g["nx"] and 0 <= iy < g["ny"]):
                continue
            tw = int(g["tile_near"][(iy // 4) * ntx + ix // 4])
            want = oracle.point_mesh_d2(x, y, flat)
            if tw == 0:
                seen[0] += 1
                assert np.sqrt(want) > 0.5 + near_range - 2.0 * g["cell"] * 4
            elif tw == 0xFFFFFFFF:
                seen[1] += 1
                assert want <= np.float32(0.25)
            else:
                seen[2] += 1
                first, ln = tw - 1, int(g["rec_len"][tw - 1])
                ids = g["rec_tri"][first:first + ln]
                got = min(oracle.point_mesh_d2(x, y, flat[k:k + 1]) for k in ids)
                assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (trial, x, y, got, want)
        assert seen[2] > 50, seen
