"""Static world tables of the batched env: drivable meshes + their grid index, scenarios (WaypointSuite entries),
NPC routes and replay rows.  Host side (numpy); `World.to_device()` uploads them once per GPU (they are
replicated per GPU) and returns the `tde_world` struct of device pointers the C-ABI takes.

Reference anchors:
  * drivable mesh = `map_cfg.road_mesh` handed to `Simulator(road_mesh=...)`      ref gym_env.py:184,260
  * waypoints / car sequences / scenarios = `WaypointSuite`                         ref gym_env.py:63-68,314-316
  * replay rows = `car_sequences` -> replay_states/replay_mask                      ref gym_env.py:275-283
  * agent ordering: slot 0 ego, then scenario agents                                ref gym_env.py:219-228
The real CARLA meshes/lanelets live inside torchdrivesim's package data, which is not part of the reference
repository, so meshes here are synthetic ("Out of scope").
"""
import math

import numpy as np

from . import _abi

# margin that makes the grid classification robust to fp32 evaluation and to the fp32 cell lookup
GRID_MARGIN = 0.05
# grid sizes are rounded up to multiples of GRID_TILE cells
GRID_TILE = 8


def row_shift_of(nx):
    """log2 of the row pitch the cell words of an nx-cell-wide grid are stored with (next power of two >= nx, at least 32:
    the 2-bit class tiles are 32 cells wide)"""
    return max(5, max(0, int(nx) - 1).bit_length())


CLS2_TILE_W, CLS2_TILE_H = 32, 16          # cells per 128-byte tile of tde_world.cell_cls2


def sub_tiles(a, nx, ny):
    """row-major per-cell words [ny * nx] -> the same words in 8 x 4-cell tiles (tde_abi.h: cell_sub), as many as
    pitch_cells() gives (ny is a multiple of GRID_TILE)"""
    assert ny % 4 == 0
    pitch = 1 << row_shift_of(nx)
    full = np.zeros((ny, pitch), dtype=a.dtype)
    full[:, :nx] = a.reshape(ny, nx)
    return np.ascontiguousarray(full.reshape(ny // 4, 4, pitch // 8, 8).transpose(0, 2, 1, 3)).reshape(-1)


def class_tiles(cls, nx, ny):
    """cell classes [ny * nx] -> uint32 words of the 2-bit class map in 32 x 16-cell tiles (tde_abi.h: cell_cls2); the
    padding holds EMPTY cells.  Returns (words, number of tiles)."""
    pitch = 1 << row_shift_of(nx)
    tx, ty = pitch // CLS2_TILE_W, -(-ny // CLS2_TILE_H)
    full = np.full((ty * CLS2_TILE_H, pitch), _abi.CELL_EMPTY, dtype=np.uint32)
    full[:ny, :nx] = np.asarray(cls, dtype=np.uint32).reshape(ny, nx)
    # 16 cells -> one word, cell ix at bits 2 * (ix & 15)
    w = (full.reshape(ty * CLS2_TILE_H, pitch // 16, 16) << (2 * np.arange(16, dtype=np.uint32))[None, None]).sum(2)
    w = w.astype(np.uint32).reshape(ty, CLS2_TILE_H, tx, 2)                 # [tile row][row in tile][tile col][word]
    return np.ascontiguousarray(w.transpose(0, 2, 1, 3)).reshape(-1), tx * ty


def pitch_cells(a, nx, ny):
    """row-major [ny*nx] cell array -> rows of 2^row_shift words (the padding holds EMPTY cells): the kernels form a
    cell's index as (iy << row_shift) + ix, one instruction (round 1 stored 8x8-cell tiles: ten per lookup with a
    quarter-rate integer multiply, for a cache-line saving that the 4-byte words of a 140 x 140-cell view never showed)"""
    pitch = 1 << row_shift_of(nx)
    out = np.full((ny, pitch), _abi.CELL_EMPTY, dtype=a.dtype)
    out[:, :nx] = a.reshape(ny, nx)
    return out.reshape(-1)


# ------------------------------------------------------------------------------------------------
# geometry helpers (float64, host, build time only)
# ------------------------------------------------------------------------------------------------
def _pairs_point_tri_dist(p, tri, want_depth=False):
    """p [P,L,2], tri [P,3,2] -> distance [P,L] of every lattice point to its pair's triangle (0 inside); with
    `want_depth` also the distance of inside points to the triangle's boundary (0 outside).
    Component-wise (x and y as separate arrays): reductions over a length-2 axis are what numpy is slowest at."""
    px, py = p[..., 0], p[..., 1]
    vx = [tri[:, k, 0][:, None] for k in range(3)]
    vy = [tri[:, k, 1][:, None] for k in range(3)]
    d2 = None
    pos = neg = None
    depth2 = None
    for k in range(3):
        ax, ay, bx, by = vx[k], vy[k], vx[(k + 1) % 3], vy[(k + 1) % 3]
        abx, aby = bx - ax, by - ay
        apx, apy = px - ax, py - ay
        e = abx * apy - aby * apx
        pos = (e >= 0) if pos is None else pos & (e >= 0)
        neg = (e <= 0) if neg is None else neg & (e <= 0)
        l2 = abx * abx + aby * aby
        t = np.clip((apx * abx + apy * aby) / np.where(l2 > 0, l2, 1.0), 0.0, 1.0)
        qx, qy = apx - t * abx, apy - t * aby
        s = qx * qx + qy * qy
        d2 = s if d2 is None else np.minimum(d2, s)
        if want_depth:
            h2 = e * e / np.where(l2 > 0, l2, 1.0)
            depth2 = h2 if depth2 is None else np.minimum(depth2, h2)
    inside = pos | neg
    if want_depth:
        return np.where(inside, 0.0, np.sqrt(d2)), np.where(inside, np.sqrt(depth2), 0.0)
    return np.where(inside, 0.0, np.sqrt(d2))


def build_grid_index(tri, threshold=0.5, cell=0.5, margin=GRID_MARGIN, lattice=4):
    """Uniform-grid index over a triangle soup `tri` [n,3,2] for the offroad test.

    For every cell: the list of triangles that can be within `threshold` of some point of the cell, and a class:
      EMPTY  no such triangle  -> every corner falling in the cell is offroad,
      FULL   every point of the cell is within `threshold` of the mesh -> never offroad,
      MIXED  test the candidates.
    Both classifications are conservative: distances are sampled on a (lattice+1)^2 lattice spanning the cell grown
    by `margin` (which absorbs the fp32 cell lookup), and the distance field is 1-Lipschitz, so between lattice points
    it moves by at most h*sqrt(2)/2; on top of that `margin` (>> fp32 evaluation error at |coords| <~ 1e3 m) is kept
    on both decisions.  Hence the HIP kernel's mask equals the oracle's brute-force mask.
    """
    tri = np.asarray(tri, dtype=np.float64).reshape(-1, 3, 2)
    R = threshold + margin
    lo = tri.reshape(-1, 2).min(0) - (R + 2 * cell)
    hi = tri.reshape(-1, 2).max(0) + (R + 2 * cell)
    # the origin must be exactly representable in fp32 (the kernel subtracts it in fp32)
    ox, oy = float(np.float32(math.floor(lo[0]))), float(np.float32(math.floor(lo[1])))
    nx = GRID_TILE * int(math.ceil((hi[0] - ox) / cell / GRID_TILE))
    ny = GRID_TILE * int(math.ceil((hi[1] - oy) / cell / GRID_TILE))
    h = (cell + 2 * margin) / lattice
    slack = h * math.sqrt(2.0) / 2.0
    # (triangle, cell) pairs from dilated triangle bounding boxes
    pt, pc = [], []
    for k, t in enumerate(tri):
        bx0, by0 = t.min(0) - (R + slack)
        bx1, by1 = t.max(0) + (R + slack)
        ix0 = max(0, int(math.floor((bx0 - ox) / cell)) - 1)
        ix1 = min(nx - 1, int(math.floor((bx1 - ox) / cell)) + 1)
        iy0 = max(0, int(math.floor((by0 - oy) / cell)) - 1)
        iy1 = min(ny - 1, int(math.floor((by1 - oy) / cell)) + 1)
        iy, ix = np.mgrid[iy0:iy1 + 1, ix0:ix1 + 1]
        c = (iy * nx + ix).ravel()
        pc.append(c)
        pt.append(np.full(c.shape, k, dtype=np.int64))
    pt, pc = np.concatenate(pt), np.concatenate(pc)
    g = np.arange(lattice + 1, dtype=np.float64) * h - margin
    lat = np.stack(np.meshgrid(g, g, indexing="xy"), -1).reshape(-1, 2)          # [L,2] offsets inside a cell
    L = lat.shape[0]
    ucell, row = np.unique(pc, return_inverse=True)                             # lattice minima only for touched cells
    dmin = np.full((len(ucell), L), np.inf)
    keep = np.zeros(pt.shape, dtype=bool)
    covered = np.zeros(nx * ny, dtype=bool)
    # the lattice is only evaluated for pairs the cell centre cannot decide: with r = half diagonal of the grown cell,
    # a centre farther than R + slack + r keeps every lattice point beyond R + slack (the pair matters to neither
    # decision); a centre at depth >= r inside the triangle puts the whole grown cell inside it (cell is FULL)
    r = (0.5 * cell + margin) * math.sqrt(2.0) + 1e-9
    ctr = np.array([[0.5 * cell, 0.5 * cell]])
    CH = 400_000
    for s0 in range(0, len(pt), CH):
        t_, c_ = pt[s0:s0 + CH], pc[s0:s0 + CH]
        org = np.stack([ox + (c_ % nx) * cell, oy + (c_ // nx) * cell], -1)      # [P,2]
        dc, depth = _pairs_point_tri_dist(org[:, None, :] + ctr[None], tri[t_], want_depth=True)
        dc, depth = dc[:, 0], depth[:, 0]
        deep = depth >= r
        covered[c_[deep]] = True
        band = np.nonzero(~deep & (dc <= R + slack + r))[0]
        if len(band):
            d = _pairs_point_tri_dist(org[band][:, None, :] + lat[None], tri[t_[band]])   # [P',L]
            keep[s0 + band] = d.min(1) <= R + slack
            np.minimum.at(dmin, row[s0 + band], d)
    full = covered
    full[ucell] |= (dmin <= (threshold - margin) - slack).all(1)
    pt, pc = pt[keep], pc[keep]
    order = np.lexsort((pt, pc))
    pt, pc = pt[order], pc[order]
    has = np.zeros(nx * ny, dtype=bool)
    has[pc] = True
    cls = np.where(full, _abi.CELL_FULL, np.where(has, _abi.CELL_MIXED, _abi.CELL_EMPTY)).astype(np.uint8)
    mixed_pair = cls[pc] == _abi.CELL_MIXED                                       # FULL cells need no list
    pt, pc = pt[mixed_pair], pc[mixed_pair]
    counts = np.bincount(pc, minlength=nx * ny)
    start = np.zeros(nx * ny + 1, dtype=np.int32)
    start[1:] = np.cumsum(counts)
    return dict(ox=ox, oy=oy, cell=float(cell), nx=nx, ny=ny, cell_class=cls, cell_start=start,
                cell_tris=pt.astype(np.int32))


# MIXED cells are split once more: SUB x SUB sub-cells, each with a 2-bit class of its own (EMPTY / MIXED / FULL) packed
# into one word, bits 2 * (sy * SUB + sx), kept in a tiled per-cell array of its own (tde_world.cell_sub).  The
# rasteriser resolves most pixels of a MIXED cell from it without a triangle test (the band of truly undecided points
# shrinks from ~0.6 m to ~0.15 m around the road edge at 0.25 m cells).  SUB_MARGIN absorbs the fp32 evaluation of the
# sub-cell coordinate and of the distances (both ~1e-4 m at |coordinates| of a few hundred metres).
SUB = 4
SUB_MARGIN = 0.002


def subcell_classes(tri, g, threshold):
    """uint32 per MIXED cell (in cell order): the classes of its SUB x SUB sub-cells.  Conservative like the cell classes:
    a 3 x 3 lattice over the sub-cell grown by SUB_MARGIN, the 1-Lipschitz slack between lattice points and SUB_MARGIN
    on both decisions; distances are taken to the cell's candidate triangles, which hold every triangle within
    `threshold` (+ the cell margin) of any point of the cell."""
    cell, nx = g["cell"], g["nx"]
    cls = g["cell_class"]
    start = g["cell_start"]
    mixed = np.nonzero(cls == _abi.CELL_MIXED)[0]
    if len(mixed) == 0:
        return np.zeros(0, np.uint32)
    counts = np.diff(start)[mixed]
    row = np.repeat(np.arange(len(mixed)), counts)                      # candidate entry -> index into `mixed`
    ent = np.concatenate([np.arange(start[c], start[c + 1]) for c in mixed]) if len(mixed) < 4096 else \
        (np.repeat(start[mixed], counts) + (np.arange(counts.sum()) - np.repeat(np.cumsum(counts) - counts, counts)))
    tri_of = g["cell_tris"][ent]
    sub = cell / SUB
    h = (sub + 2 * SUB_MARGIN) / 2.0
    slack = h * math.sqrt(2.0) / 2.0
    gl = np.array([-SUB_MARGIN, 0.5 * sub, sub + SUB_MARGIN])
    # lattice offsets inside the cell: [SUB*SUB sub-cells][9 points][2]
    lat = np.zeros((SUB * SUB, 9, 2))
    for sy in range(SUB):
        for sx in range(SUB):
            xs, ys = np.meshgrid(sx * sub + gl, sy * sub + gl, indexing="xy")
            lat[sy * SUB + sx] = np.stack([xs.ravel(), ys.ravel()], -1)
    lat = lat.reshape(-1, 2)
    dmin = np.full((len(mixed), lat.shape[0]), np.inf)
    CH = 100_000
    for s0 in range(0, len(row), CH):
        r_, t_ = row[s0:s0 + CH], tri_of[s0:s0 + CH]
        c_ = mixed[r_]
        org = np.stack([g["ox"] + (c_ % nx) * cell, g["oy"] + (c_ // nx) * cell], -1)
        d = _pairs_point_tri_dist(org[:, None, :] + lat[None], tri[t_])
        np.minimum.at(dmin, r_, d)
    dm = dmin.reshape(len(mixed), SUB * SUB, 9)
    full = (dm <= (threshold - SUB_MARGIN) - slack).all(2)
    empty = (dm > (threshold + SUB_MARGIN) + slack).all(2)
    code = np.where(full, _abi.CELL_FULL, np.where(empty, _abi.CELL_EMPTY, _abi.CELL_MIXED)).astype(np.uint32)
    return (code << (2 * np.arange(SUB * SUB, dtype=np.uint32))[None]).sum(1).astype(np.uint32)


CLEARANCE_UNIT = 0.125   # metres per count of the clearance field


def cell_clearance(g, cell):
    """per cell, floor(rho / CLEARANCE_UNIT) clipped to 255, where rho is the distance between the cell's rectangle and
    the nearest cell rectangle of another class (FULL or EMPTY cells; 0 for MIXED): every point within rho of any
    point of the cell lies in a cell of the same class.  Exact rectangle-to-rectangle distance = centre distance to
    the other-class set dilated by one cell (Chebyshev), from a Euclidean distance transform."""
    from scipy.ndimage import binary_dilation, distance_transform_edt

    cls = g["cell_class"].reshape(g["ny"], g["nx"])
    out = np.zeros(cls.shape, np.float64)
    for c in (_abi.CELL_FULL, _abi.CELL_EMPTY):
        m = cls == c
        if m.any():
            other = binary_dilation(~m, structure=np.ones((3, 3), bool))
            d = distance_transform_edt(~other) * cell - 1e-3
            out = np.where(m, d, out)
    return np.clip(np.floor(np.maximum(out, 0.0) / CLEARANCE_UNIT), 0, 255).astype(np.int64).reshape(-1)


def pack_triangles(tri32):
    """Device-side triangle records [n,12] fp32: ax,ay,bx,by | cx,cy,inv|ab|^2,inv|bc|^2 | inv|ca|^2,0,0,0.
    The reciprocals are computed in fp32 exactly as the oracle's `1.0f / len2` so the kernel's distances keep every
    bit while doing no division."""
    t = np.asarray(tri32, dtype=np.float32).reshape(-1, 3, 2)
    out = np.zeros((len(t), 12), dtype=np.float32)
    out[:, 0:6] = t.reshape(-1, 6)
    one = np.float32(1.0)
    for e, (i, j) in enumerate(((0, 1), (1, 2), (2, 0))):
        abx = t[:, j, 0] - t[:, i, 0]
        aby = t[:, j, 1] - t[:, i, 1]
        len2 = abx * abx + aby * aby
        with np.errstate(divide="ignore"):
            inv = np.where(len2 > 0, one / np.where(len2 > 0, len2, one), np.float32(0.0)).astype(np.float32)
        out[:, 6 + e] = inv
    return out


# ------------------------------------------------------------------------------------------------
# synthetic drivable meshes
# ------------------------------------------------------------------------------------------------
def strip_mesh(polyline, width, seg_len=5.0):
    """Triangulated road strip of `width` metres around a polyline (list of (x,y)); quads of ~seg_len."""
    pts = np.asarray(polyline, dtype=np.float64)
    tris = []
    for a, b in zip(pts[:-1], pts[1:]):
        d = b - a
        L = float(np.hypot(*d))
        if L == 0:
            continue
        t = d / L
        n = np.array([-t[1], t[0]]) * (0.5 * width)
        k = max(1, int(round(L / seg_len)))
        for i in range(k):
            p = a + d * (i / k)
            q = a + d * ((i + 1) / k)
            tris.append([p - n, q - n, q + n])
            tris.append([p - n, q + n, p + n])
    return np.asarray(tris, dtype=np.float64).reshape(-1, 3, 2)


def disc_mesh(center, radius, n=12):
    c = np.asarray(center, dtype=np.float64)
    ang = np.linspace(0, 2 * math.pi, n + 1)
    ring = c + radius * np.stack([np.cos(ang), np.sin(ang)], -1)
    return np.asarray([[c, ring[i], ring[i + 1]] for i in range(n)], dtype=np.float64)


def corridor_mesh(polylines, width=12.0, seg_len=6.0, joint_radius=None):
    """Drivable corridor around a set of polylines (used to give the shipped WaypointSuite scenarios a mesh:
    their CARLA town meshes are not in the reference repository)."""
    parts = []
    for pl in polylines:
        pl = np.asarray(pl, dtype=np.float64).reshape(-1, 2)
        keep = [0] + [i for i in range(1, len(pl)) if np.hypot(*(pl[i] - pl[i - 1])) > 1.0]   # drop repeated points
        dedup = [keep[0]]
        for i in keep[1:]:
            if np.hypot(*(pl[i] - pl[dedup[-1]])) > 1.0:
                dedup.append(i)
        pl = pl[dedup]
        if len(pl) >= 2:
            parts.append(strip_mesh(pl, width, seg_len))
        r = joint_radius if joint_radius is not None else 0.5 * width
        for p in pl:
            parts.append(disc_mesh(p, r, 8))
    return np.concatenate(parts, 0)


# ------------------------------------------------------------------------------------------------
# World container
# ------------------------------------------------------------------------------------------------
class World:
    """Host copy of every static table the step path reads (numpy, dtypes of _abi.WORLD_DTYPES)."""

    def __init__(self, arrays, ints, threshold=None):
        self.arrays = {k: np.ascontiguousarray(arrays[k], dtype=_abi.WORLD_DTYPES[k]) for k in _abi.WORLD_PTRS}
        self.ints = {k: int(ints[k]) for k in _abi.WORLD_INTS}
        # The offroad distance the grid index was built for: cell classes and candidate lists bake it in
        # (build_grid_index), so kernels run with another threshold would silently return wrong masks.  Checked by
        # check_threshold() wherever a threshold meets a World.  None: unknown (tables assembled by hand).
        self.threshold = None if threshold is None else float(threshold)
        self.has_lights = bool(np.asarray(arrays["maps"])["cycle_steps"].max() > 0)
        # zero-length tables still need a valid pointer
        for k, a in self.arrays.items():
            if a.size == 0:
                self.arrays[k] = np.zeros(max(1, 1), dtype=a.dtype)
        self._host_struct = None

    def __getstate__(self):                      # the cached ctypes struct holds raw pointers: never pickled
        d = dict(self.__dict__)
        d["_host_struct"] = None
        return d

    @property
    def A(self):
        return self.ints["A"]

    @property
    def n_scn(self):
        return self.ints["n_scn"]

    def host_struct(self):
        """tde_world of HOST pointers (what the CPU oracle takes in the tests)."""
        if self._host_struct is None:
            self._host_struct = _abi.fill_world_struct(self.arrays, self.ints)
        return self._host_struct

    def to_device(self, device):
        import torch

        tens = {}
        for k, a in self.arrays.items():
            if a.dtype.names is not None:      # record arrays travel as raw bytes
                t = torch.from_numpy(a.reshape(-1).view(np.uint8).copy())
            else:
                t = torch.from_numpy(a.copy())
            tens[k] = t.to(device)
        return DeviceWorld(tens, self.ints, self.threshold)

    def map_of_scn(self):
        return np.ascontiguousarray(self.arrays["scn"]["map"])

    def save(self, path):
        """cache the assembled tables (grid indexes of town-sized meshes take seconds per scenario to build)"""
        blobs = {f"a_{k}": a.reshape(-1).view(np.uint8) if a.dtype.names is not None else a
                 for k, a in self.arrays.items()}
        ints = np.array([self.ints[k] for k in _abi.WORLD_INTS], dtype=np.int64)
        with open(path, "wb") as f:
            np.savez(f, abi=np.int64(_abi.TDE_ABI_VERSION), ints=ints,
                     threshold=np.float64(-1.0 if self.threshold is None else self.threshold), **blobs)

    @classmethod
    def load(cls, path):
        with np.load(path) as z:
            if int(z["abi"]) != _abi.TDE_ABI_VERSION:
                raise ValueError(f"{path}: built for ABI {int(z['abi'])}, this library is ABI {_abi.TDE_ABI_VERSION}")
            ints = dict(zip(_abi.WORLD_INTS, (int(v) for v in z["ints"])))
            arrays = {}
            for k in _abi.WORLD_PTRS:
                dt = np.dtype(_abi.WORLD_DTYPES[k])
                a = z[f"a_{k}"]
                arrays[k] = a.view(dt) if dt.names is not None else a
            thr = float(z["threshold"]) if "threshold" in z.files else -1.0
        return cls(arrays, ints, None if thr < 0 else thr)


def effective_offroad_distance(threshold, squared=False):
    """the distance a box corner may be from the mesh: `threshold`, or sqrt(threshold) when the threshold is applied to
    the SQUARED distance (tde_config.offroad_threshold_squared)"""
    return float(np.sqrt(threshold)) if squared else float(threshold)


def check_threshold(world, threshold, squared=False, what="offroad_threshold"):
    """raise if `world`'s grid index was built for another offroad distance than (threshold, squared) asks for"""
    built = getattr(world, "threshold", None)
    want = effective_offroad_distance(threshold, squared)
    if built is not None and abs(built - want) > 1e-6 * max(1.0, want):
        raise ValueError(f"{what} asks for an offroad distance of {want:g} m but the World's grid index was built for "
                         f"{built:g} m: rebuild the World (assemble_world(..., threshold={want:g}))")


class DeviceWorld:
    def __init__(self, tensors, ints, threshold=None):
        self.tensors = tensors  # keeps the device memory alive
        self.ints = dict(ints)
        self.threshold = threshold
        self.struct = _abi.fill_world_struct(tensors, ints)


def assemble_world(meshes, scenarios, A, threshold=0.5, cell=0.5, lights=None):
    """meshes: list of [n,3,2] triangle arrays; scenarios: list of dicts with keys
         map (int), waypoints [(x,y)...], start_heading (float),
         agents: list (slots 1..) of dict(state=(x,y,psi,v), attr=(L,W,lr), vdes, route=[(x,y)..] or None,
                                         replay=[(x,y,psi,v)...] or None)
         ego_attr (L,W,lr)
    """
    assert A >= 1 and (A & (A - 1)) == 0 and A <= _abi.TDE_MAX_AGENTS, "A must be a power of two <= 64"
    # lights: per map None or dict(stoplines=[(x, y, psi, length, width, light)...], phases=[(n_steps, red_lights)...])
    lights = lights or [None] * len(meshes)
    stop_all, phase_all = [], []
    maps = np.zeros(len(meshes), dtype=_abi.MAP_DTYPE)
    tri_all, word_all, rec_all, cls2_all, sub_all = [], [], [], [], []
    tri_base = cell_base = rec_base = cls2_base = 0
    for m, tri in enumerate(meshes):
        tri = np.asarray(tri, dtype=np.float64).reshape(-1, 3, 2)
        # the kernels see fp32 vertices: index the fp32-rounded mesh
        tri32 = tri.astype(np.float32)
        g = build_grid_index(tri32.astype(np.float64), threshold, cell)
        lt = lights[m]
        n_stop = n_phase = cycle = 0
        if lt:
            n_stop, n_phase = len(lt["stoplines"]), len(lt["phases"])
            assert n_phase >= 1 and max(sl[5] for sl in lt["stoplines"]) < 32
            for (x, y, psi, length, width, light) in lt["stoplines"]:
                stop_all.append((x, y, math.cos(psi), math.sin(psi), 0.5 * length, 0.5 * width, light, 0))
            for (n_steps, red) in lt["phases"]:
                cycle += int(n_steps)
                phase_all.append((cycle, sum(1 << int(i) for i in red)))
        maps[m] = (g["ox"], g["oy"], g["cell"], np.float32(1.0) / np.float32(g["cell"]), g["nx"], g["ny"], cell_base,
                   tri_base, len(tri), len(stop_all) - n_stop, n_stop, len(phase_all) - n_phase, n_phase, cycle,
                   row_shift_of(g["nx"]), cls2_base)
        packed = pack_triangles(tri32)
        counts = np.diff(g["cell_start"]).astype(np.int64)
        # FULL / EMPTY cells carry no candidate list: their count field holds a clearance instead (quarter metres,
        # rounded down): every point within that distance of ANY point of the cell lies in a cell of the same class.
        # The rasteriser uses it to classify a whole 4x4 pixel block with one lookup.
        counts = np.where(g["cell_class"] == _abi.CELL_MIXED, counts, cell_clearance(g, cell))
        assert counts.max(initial=0) <= 255, "more than 255 candidate triangles in one grid cell: use a smaller cell"
        start = g["cell_start"][:-1].astype(np.int64) + rec_base
        assert start.max(initial=0) < (1 << 22), "grid index too large for the 22-bit record offset"
        word_all.append(pitch_cells((g["cell_class"].astype(np.uint32) | (counts.astype(np.uint32) << 2) |
                                    (start.astype(np.uint32) << 10)).astype(np.uint32), g["nx"], g["ny"]))
        rec_all.append(packed[g["cell_tris"]])          # per-cell copies: one dependent load less in the kernel
        sub = np.zeros(g["nx"] * g["ny"], np.uint32)    # sub-cell classes of the MIXED cells, tiled 8 x 4 cells per line
        mixed = np.nonzero(g["cell_class"] == _abi.CELL_MIXED)[0]
        if len(mixed):
            sub[mixed] = subcell_classes(tri32.astype(np.float64), g, threshold)
        sub_all.append(sub_tiles(sub, g["nx"], g["ny"]))
        c2, ntile = class_tiles(g["cell_class"], g["nx"], g["ny"])
        cls2_all.append(c2)
        cls2_base += ntile
        tri_all.append(tri32.reshape(-1, 6))
        tri_base += len(tri)
        cell_base += (1 << row_shift_of(g["nx"])) * g["ny"]
        rec_base += len(g["cell_tris"])
    S = len(scenarios)
    NW = max(2, max(len(s["waypoints"]) for s in scenarios))
    wp_xy = np.zeros((S, NW, 2), np.float64)
    scn = np.zeros(S, _abi.SCN_DTYPE)
    spawn = np.zeros((S, A), _abi.SPAWN_DTYPE)
    spawn["len"], spawn["wid"], spawn["lr"] = 1.0, 1.0, 1.0
    spawn["route"], spawn["replay"] = -1, -1
    routes, replays = [], []
    for si, s in enumerate(scenarios):
        w = np.asarray(s["waypoints"], np.float64)
        assert len(w) >= 2, "a scenario needs at least two waypoints (gym_env.py:353-354)"
        wp_xy[si, :len(w)] = w
        scn[si] = (s["map"], len(w), s["start_heading"], 0)
        ego = spawn[si, 0]
        ego["present"] = 1
        ego["len"], ego["wid"], ego["lr"] = s.get("ego_attr", (5.0, 2.0, 1.9))
        ego["x"], ego["y"], ego["psi"] = w[0, 0], w[0, 1], s["start_heading"]
        ags = s.get("agents", [])
        assert len(ags) <= A - 1, f"scenario {si} has {len(ags)} NPCs but only {A - 1} NPC slots"
        for k, ag in enumerate(ags):
            r = spawn[si, k + 1]
            r["present"] = 1
            r["x"], r["y"], r["psi"], r["v"] = ag["state"]
            r["len"], r["wid"], r["lr"] = ag["attr"]
            r["vdes"] = ag.get("vdes", ag["state"][3])
            if ag.get("route") is not None and len(ag["route"]) > 0:
                r["route"], r["route_n"], r["route_wp"] = len(routes), len(ag["route"]), ag.get("route_wp", 0)
                routes.append(np.asarray(ag["route"], np.float32))
                if r["route_wp"] < r["route_n"]:
                    r["tgx0"], r["tgy0"] = routes[-1][r["route_wp"]]          # the first target rides in the record
            if ag.get("replay") is not None and len(ag["replay"]) > 0:
                r["replay"], r["replay_len"] = len(replays), len(ag["replay"])
                replays.append(np.asarray(ag["replay"], np.float32))
    RW = max([len(r) for r in routes], default=1)
    route_xy = np.zeros((max(1, len(routes)), RW, 2), np.float32)
    for i, r in enumerate(routes):
        route_xy[i, :len(r)] = r
    RT = max([len(r) for r in replays], default=1)
    replay_states = np.zeros((max(1, len(replays)), RT, 4), np.float32)
    for i, r in enumerate(replays):
        replay_states[i, :len(r)] = r
    rec_cat = np.concatenate(rec_all, 0) if rec_base else np.zeros((1, 12), np.float32)
    arrays = dict(maps=maps, tri=np.concatenate(tri_all, 0), cell_word=np.concatenate(word_all), cell_tri=rec_cat,
                  cell_cls2=np.concatenate(cls2_all), cell_sub=np.concatenate(sub_all),
                  scn=scn, wp_xy=wp_xy, spawn=spawn, route_xy=route_xy, replay_states=replay_states,
                  stoplines=np.asarray(stop_all, dtype=_abi.STOPLINE_DTYPE) if stop_all
                  else np.zeros(1, _abi.STOPLINE_DTYPE),
                  phases=np.asarray(phase_all, dtype=_abi.PHASE_DTYPE) if phase_all else np.zeros(1, _abi.PHASE_DTYPE))
    ints = dict(n_maps=len(meshes), n_scn=S, NW=NW, A=A, n_routes=len(routes), RW=RW, n_replay=len(replays), RT=RT)
    return World(arrays, ints, threshold)
