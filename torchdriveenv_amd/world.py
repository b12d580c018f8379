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
    """log2 of the row pitch the cell words of an nx-cell-wide grid are stored with (next power of two >= nx, at least 128:
    a line of the coarse table is 64 cells wide, a 2-bit class tile 32)"""
    return max(7, max(0, int(nx) - 1).bit_length())


CLS2_TILE_W, CLS2_TILE_H = 32, 16          # cells per 128-byte tile of tde_world.cell_cls2
CLEARANCE_UNIT = 0.125   # metres per count of the clearance field (TDE_CLEARANCE_UNIT)


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


COARSE, COARSE_UNIT = 4, 0.25               # TDE_COARSE_CELLS, TDE_COARSE_UNIT
LARGE_GRID_CELLS = 1 << 21                  # a map with more cells sets TDE_WORLD_LARGE_GRID (tde_abi.h)


def coarse_tiles(cls, count, nx, ny):
    """cell classes and count fields [ny * nx] -> bytes of the coarse table (tde_abi.h: cell_coarse) in 128-byte lines of
    16 x 8 coarse tiles, and the number of lines.  A tile of 4 x 4 cells is FULL / EMPTY when all its cells are, with the
    smallest of their clearances (the distance between the tile and the nearest cell of another class), else MIXED / 0."""
    pitch = 1 << row_shift_of(nx)
    lx, ly = pitch // 64, -(-ny // 32)
    c = np.full((ly * 32, pitch), _abi.CELL_EMPTY, np.uint8)
    k = np.zeros((ly * 32, pitch), np.uint8)
    c[:ny, :nx] = np.asarray(cls, np.uint8).reshape(ny, nx)
    k[:ny, :nx] = np.asarray(count, np.uint8).reshape(ny, nx)
    c4 = c.reshape(ly * 8, COARSE, pitch // COARSE, COARSE)
    lo, hi = c4.min((1, 3)), c4.max((1, 3))
    uni = (lo == hi) & (lo != _abi.CELL_MIXED)
    clear = np.floor(k.reshape(ly * 8, COARSE, pitch // COARSE, COARSE).min((1, 3)) * (CLEARANCE_UNIT / COARSE_UNIT))
    byte = np.where(uni, lo | (np.minimum(clear, 63).astype(np.uint8) << 2), _abi.CELL_MIXED).astype(np.uint8)
    return np.ascontiguousarray(byte.reshape(ly, 8, lx, 16).transpose(0, 2, 1, 3)).reshape(-1), lx * ly


def pitch_cells(a, nx, ny):
    """row-major [ny*nx] cell array -> rows of 2^row_shift words (the padding holds EMPTY cells): the kernels form a
    cell's index as (iy << row_shift) + ix, one instruction (round 1 stored 8x8-cell tiles: ten per lookup with a
    quarter-rate integer multiply, for a cache-line saving that the 4-byte words of a 140 x 140-cell view never showed)"""
    pitch = 1 << row_shift_of(nx)
    out = np.full((ny, pitch), _abi.CELL_EMPTY, dtype=a.dtype)
    out[:, :nx] = a.reshape(ny, nx)
    return out.reshape(-1)


# ------------------------------------------------------------------------------------------------
# grid index of a drivable mesh: built by the library's host code (csrc/tde_gridbuild.h)
# ------------------------------------------------------------------------------------------------
SUB = 4                  # a MIXED cell carries SUB x SUB sub-cell classes (TDE_CELL_SUB)


_GRID_LIB = None


def _grid_lib():
    """the library that holds tde_grid_build / tde_grid_free: libtde_hip.so - or, under TDE_GRID_LIB, a host-compiler build of
    the same source (csrc/tde_gridbuild.h through csrc/tde_grid_host.cpp) with AddressSanitizer / UBSan in it
    (`make -C torchdriveenv_amd/csrc san` -> torchdriveenv_amd/_san/; tests/test_sanitizers.py).  The tables do not depend on which
    one built them."""
    global _GRID_LIB
    import ctypes as C
    import os

    from . import _lib

    path = os.environ.get("TDE_GRID_LIB")
    if not path:
        return _lib.load()
    if _GRID_LIB is None:
        L = C.CDLL(path)
        if L.tde_abi_version() != _abi.TDE_ABI_VERSION:
            raise _lib.TdeError(f"{path}: ABI {L.tde_abi_version()}, expected {_abi.TDE_ABI_VERSION}")
        L.tde_last_error.restype = C.c_char_p
        L.tde_grid_build.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                     C.POINTER(C.POINTER(_abi.TdeGrid))]
        L.tde_grid_build.restype = C.c_int
        L.tde_grid_free.argtypes = [C.POINTER(_abi.TdeGrid)]
        L.tde_grid_free.restype = None
        _GRID_LIB = L
    return _GRID_LIB


NEAR_RANGE = 2.0         # metres beyond the threshold the coarse tiles carry a near list for (tde_world.tile_near)


def build_grid_index(tri32, threshold=0.5, cell=0.5, margin=GRID_MARGIN, n_threads=0, near_range=NEAR_RANGE):
    """Uniform-grid index over a triangle soup `tri32` [n,3,2] (fp32 vertices, what kernels and oracle see) for the offroad
    test, through `tde_grid_build` (include/tde_hip.h; rounds 1-3 did this in numpy, seconds per 200-triangle junction).

    Per cell a class - EMPTY: every corner falling in the cell is offroad; FULL: never offroad; MIXED: test the cell's
    candidate triangles - plus, for MIXED cells, the candidate list (identical lists share their records) and the classes
    of its SUB x SUB sub-cells, and for FULL / EMPTY cells a clearance.  All conservative by `margin`, so the HIP kernels'
    masks equal the oracle's brute force (tests/test_oracle_math.py::test_grid_index_equals_brute_force).
    `near_range`: coarse tiles (4 x 4 cells) within threshold + near_range of the mesh also get a NEAR LIST - the triangles among
    which the nearest triangle of any point of the tile is found - for the MAGNITUDE of the offroad infraction
    (ref gym_env.py:427); `tile_near` [ny/4 * nx/4] holds 1 + its first record, `rec_len` its length at that record.
    Returns row-major [ny * nx] numpy arrays (copies) and `rec_tri`, the triangle of every record."""
    import ctypes as C

    from . import _lib

    L = _grid_lib()
    tri32 = np.ascontiguousarray(np.asarray(tri32, dtype=np.float32).reshape(-1, 6))
    gp = C.POINTER(_abi.TdeGrid)()
    rc = L.tde_grid_build(tri32.ctypes.data, len(tri32), float(threshold), float(cell), float(margin), float(near_range),
                          int(n_threads), C.byref(gp))
    if rc != 0:
        raise _lib.TdeError("tde_grid_build: " + (L.tde_last_error().decode() or f"error {rc}"))
    try:
        g = gp.contents
        n = g.nx * g.ny
        out = dict(ox=float(g.ox), oy=float(g.oy), cell=float(g.cell), nx=int(g.nx), ny=int(g.ny), n_lists=int(g.n_lists),
                   cell_class=np.ctypeslib.as_array(g.cell_class, (n,)).copy(),
                   cell_count=np.ctypeslib.as_array(g.cell_count, (n,)).copy(),
                   cell_first=np.ctypeslib.as_array(g.cell_first, (n,)).copy(),
                   cell_sub=np.ctypeslib.as_array(g.cell_sub, (n,)).copy(),
                   rec_tri=np.ctypeslib.as_array(g.rec_tri, (max(1, int(g.n_records)),))[:int(g.n_records)].copy(),
                   rec_len=np.ctypeslib.as_array(g.rec_len, (max(1, int(g.n_records)),))[:int(g.n_records)].copy(),
                   tile_near=np.ctypeslib.as_array(g.tile_near, ((g.nx // COARSE) * (g.ny // COARSE),)).copy(),
                   n_near_lists=int(g.n_near_lists))
    finally:
        L.tde_grid_free(gp)
    return out


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


def cached_world(key, builder, cache_dir=None):
    """`builder()` -> World, built ONCE per machine for a given `key` and shared through a file: the first caller (a file lock
    decides) builds and saves it (World.save), everybody else - the other ranks of a multi-GPU job, whose grid-index builds would
    otherwise run side by side on the same host cores before the first launch - loads the tables (World.load: a read of the page
    cache).  The key is extended by the ABI version and the modification times of the library and of the Python that assembles
    worlds, so a rebuilt package never meets a stale file.  Returns (world, "built" | "loaded")."""
    import fcntl
    import hashlib
    import os
    import tempfile

    here = os.path.dirname(os.path.abspath(__file__))
    stamp = [str(_abi.TDE_ABI_VERSION)]
    for f in ("libtde_hip.so", "world.py", "synth.py", "_abi.py"):
        try:
            stamp.append(str(os.stat(os.path.join(here, f)).st_mtime_ns))
        except OSError:
            stamp.append("-")
    h = hashlib.sha1(("|".join([str(key)] + stamp)).encode()).hexdigest()[:20]
    d = cache_dir or os.environ.get("TDE_WORLD_CACHE") or os.path.join(tempfile.gettempdir(), f"tde_worlds_{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"world_{h}.npz")
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(path):
                try:
                    return World.load(path), "loaded"
                except Exception:
                    os.unlink(path)                         # (a torn or foreign file: rebuild)
            w = builder()
            tmp = path + f".tmp{os.getpid()}"
            w.save(tmp)
            os.replace(tmp, path)
            return w, "built"
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


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


def assemble_world(meshes, scenarios, A, threshold=0.5, cell=0.5, lights=None, light_groups=None, near_range=NEAR_RANGE):
    """meshes: list of [n,3,2] triangle arrays; scenarios: list of dicts with keys
         map (int), waypoints [(x,y)...], start_heading (float),
         start_headings (optional): the lane direction at NH points along the first waypoint segment, entry j at
           p0 + (j + 0.5) / NH * (p1 - p0) - the heading field the reference samples with find_lanelet_directions at the drawn
           start point (ref gym_env.py:359-361); every scenario that gives one must give the same NH, the others get their
           start_heading repeated.  An episode that starts at fraction f of the segment reads entry floor(f * NH)
         agents: list (slots 1..) of dict(state=(x,y,psi,v), attr=(L,W,lr), vdes, route=[(x,y)..] or None,
                                         replay=[(x,y,psi,v)...] or None)
         ego_attr (L,W,lr)
         lights (int, optional): index into `light_groups`
       lights: per mesh None or dict(stoplines=[(x, y, psi, length, width, light)...], phases=[(n_steps, red_lights)...]) - the
         traffic lights every scenario on that mesh sees.
       light_groups: list of dict(map=mesh index, stoplines=..., phases=...) - the lights of ONE NEIGHBOURHOOD of a large map.  The
         kernels walk all stop lines of a scenario's map descriptor and a light is a bit of a 32-bit mask, so a town with hundreds of
         signals is cut up at build time: every group becomes a map descriptor of its own (tde_map: 80 bytes) that SHARES the mesh's
         grid tables and carries only its stop lines and phases; a scenario with `lights=k` is bound to descriptor n_meshes + k.
    """
    assert A >= 1 and (A & (A - 1)) == 0 and A <= _abi.TDE_MAX_AGENTS, f"A must be a power of two <= {_abi.TDE_MAX_AGENTS}"
    lights = lights or [None] * len(meshes)
    light_groups = list(light_groups or [])
    stop_all, phase_all = [], []

    def add_lights(lt):
        """-> (stop_base, n_stop, phase_base, n_phase, cycle_steps) of one set of lights appended to the world's tables"""
        if not lt:
            return len(stop_all), 0, len(phase_all), 0, 0
        n_stop, n_phase, cycle = len(lt["stoplines"]), len(lt["phases"]), 0
        assert n_phase >= 1 and n_stop >= 1 and max(sl[5] for sl in lt["stoplines"]) < 32, "light indices are bits of a 32-bit mask"
        for (x, y, psi, length, width, light) in lt["stoplines"]:
            stop_all.append((x, y, math.cos(psi), math.sin(psi), 0.5 * length, 0.5 * width, light, 0))
        for (n_steps, red) in lt["phases"]:
            cycle += int(n_steps)
            phase_all.append((cycle, sum(1 << int(i) for i in red)))
        return len(stop_all) - n_stop, n_stop, len(phase_all) - n_phase, n_phase, cycle

    maps = np.zeros(len(meshes) + len(light_groups), dtype=_abi.MAP_DTYPE)
    tri_all, word_all, rec_all, cls2_all, sub_all, coarse_all, near_all = [], [], [], [], [], [], []
    tri_base = cell_base = rec_base = cls2_base = coarse_base = near_base = 0
    for m, tri in enumerate(meshes):
        tri = np.asarray(tri, dtype=np.float64).reshape(-1, 3, 2)
        # the kernels see fp32 vertices: index the fp32-rounded mesh
        tri32 = tri.astype(np.float32)
        g = build_grid_index(tri32, threshold, cell, near_range=near_range)
        stop_base, n_stop, phase_base, n_phase, cycle = add_lights(lights[m])
        maps[m] = (g["ox"], g["oy"], g["cell"], np.float32(1.0) / np.float32(g["cell"]), g["nx"], g["ny"], cell_base,
                   tri_base, len(tri), stop_base, n_stop, phase_base, n_phase, cycle,
                   row_shift_of(g["nx"]), cls2_base, rec_base, coarse_base, near_base, 0)
        # cell word = class | count << 2 | first record << 10: the count of a MIXED cell is the length of its candidate list
        # (records from the map's rec_base + first on), that of a FULL / EMPTY cell its clearance (TDE_CLEARANCE_UNITs, rounded
        # down): every point within that distance of ANY point of the cell lies in a cell of the same class - the rasteriser
        # classifies a whole block of pixels with one look-up
        word_all.append(pitch_cells(g["cell_class"].astype(np.uint32) | (g["cell_count"].astype(np.uint32) << 2) |
                                    (g["cell_first"] << 10), g["nx"], g["ny"]))
        recs = pack_triangles(tri32)[g["rec_tri"]]           # per-list copies: one dependent load less in the kernel
        recs[:, 9] = g["rec_len"].astype(np.int32).view(np.float32)   # (the length of a near list rides in its first record)
        rec_all.append(recs)
        near_all.append(g["tile_near"])
        near_base += len(g["tile_near"])
        sub_all.append(sub_tiles(g["cell_sub"], g["nx"], g["ny"]))   # sub-cell classes of the MIXED cells, 8 x 4 cells per line
        c2, ntile = class_tiles(g["cell_class"], g["nx"], g["ny"])
        cls2_all.append(c2)
        cls2_base += ntile
        cb, nline = coarse_tiles(g["cell_class"], g["cell_count"], g["nx"], g["ny"])
        coarse_all.append(cb)
        coarse_base += nline
        tri_all.append(tri32.reshape(-1, 6))
        tri_base += len(tri)
        cell_base += (1 << row_shift_of(g["nx"])) * g["ny"]
        rec_base += len(g["rec_tri"])
        assert cell_base < (1 << 30) and cls2_base < (1 << 26), "world too large for 32-bit cell indices"
    for k, lt in enumerate(light_groups):            # a descriptor per light group: the mesh's grid, the group's lights
        assert 0 <= lt["map"] < len(meshes), "light group on an unknown mesh"
        maps[len(meshes) + k] = maps[lt["map"]]
        d = maps[len(meshes) + k:len(meshes) + k + 1]
        d["stop_base"], d["n_stop"], d["phase_base"], d["n_phase"], d["cycle_steps"] = add_lights(lt)
    S = len(scenarios)
    NW = max(2, max(len(s["waypoints"]) for s in scenarios))
    wp_xy = np.zeros((S, NW, 2), np.float64)
    scn = np.zeros(S, _abi.SCN_DTYPE)
    nhs = {len(sc["start_headings"]) for sc in scenarios if sc.get("start_headings") is not None}
    assert len(nhs) <= 1 and 0 not in nhs, f"scenarios disagree on the number of start-heading samples: {sorted(nhs)}"
    NH = nhs.pop() if nhs else 0
    start_psi = np.zeros((S, max(NH, 1)), np.float32)
    spawn = np.zeros((S, A), _abi.SPAWN_DTYPE)
    spawn["len"], spawn["wid"], spawn["lr"] = 1.0, 1.0, 1.0
    spawn["route"], spawn["replay"] = -1, -1
    routes, replays = [], []
    for si, s in enumerate(scenarios):
        w = np.asarray(s["waypoints"], np.float64)
        assert len(w) >= 2, "a scenario needs at least two waypoints (gym_env.py:353-354)"
        wp_xy[si, :len(w)] = w
        m_of = s["map"]
        if s.get("lights") is not None:
            assert light_groups[s["lights"]]["map"] == s["map"], f"scenario {si}: its light group belongs to another mesh"
            m_of = len(meshes) + s["lights"]
        scn[si] = (m_of, len(w), s["start_heading"], 0)
        start_psi[si] = s["start_heading"] if s.get("start_headings") is None else np.asarray(s["start_headings"], np.float64)
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
    # (16 zero records at the end: the magnitude kernels fetch the first 16 records of a near list before they know its length)
    rec_cat = np.concatenate(rec_all + [np.zeros((16, 12), np.float32)], 0)
    arrays = dict(maps=maps, tri=np.concatenate(tri_all, 0), cell_word=np.concatenate(word_all), cell_tri=rec_cat,
                  cell_cls2=np.concatenate(cls2_all), cell_sub=np.concatenate(sub_all), cell_coarse=np.concatenate(coarse_all),
                  tile_near=np.concatenate(near_all), scn=scn, wp_xy=wp_xy, spawn=spawn, route_xy=route_xy, replay_states=replay_states,
                  stoplines=np.asarray(stop_all, dtype=_abi.STOPLINE_DTYPE) if stop_all
                  else np.zeros(1, _abi.STOPLINE_DTYPE),
                  phases=np.asarray(phase_all, dtype=_abi.PHASE_DTYPE) if phase_all else np.zeros(1, _abi.PHASE_DTYPE),
                  start_psi=start_psi,
                  # the first-step gap cache (tde_first_gap [S][A]): scratch of the device copy, all entries invalid
                  first_gap=np.zeros((S * A, 2), np.uint32))
    large = bool((maps["nx"].astype(np.int64) * maps["ny"]).max() > LARGE_GRID_CELLS)
    ints = dict(n_maps=len(maps), n_scn=S, NW=NW, A=A, n_routes=len(routes), RW=RW, n_replay=len(replays), RT=RT,
                hints=_abi.WORLD_LARGE_GRID if large else 0, NH=NH)
    return World(arrays, ints, threshold)
