"""Decision margins of the two masks, for the PyTorch-witness tests (the north star: "bit-exact for collision/offroad masks and within
1e-5 on fp32 kinematic state" against a PyTorch step).  Two correct fp32 evaluations of a threshold test may disagree only where the
quantity tested lies within rounding of its threshold; these functions compute, in float64 from the fp32 state, how far every
slot's decision is from flipping, so that the tests can demand EQUAL masks on every slot outside a stated band and count the rest:

  collision  margin_i = min over the other present slots j of |s_ij|, s_ij = max over the four separating axes of (|projected centre
             offset| - sum of the projected half extents): the boxes overlap iff s_ij < 0 (strict SAT, ref gym_env.py:143 via the
             nograd IoU > 0)
  offroad    margin_i = min over the four box corners of |distance to the mesh - threshold| (ref gym_env.py:142)
"""
import numpy as np

BAND = 1e-4          # metres: the band around a mask's decision threshold inside which two fp32 evaluations may disagree


def collision_margin(st, B, A):
    x, y, psi = (np.asarray(st[k], np.float64).reshape(B, A) for k in ("x", "y", "psi"))
    hl, hw = 0.5 * np.asarray(st["len"], np.float64).reshape(B, A), 0.5 * np.asarray(st["wid"], np.float64).reshape(B, A)
    present = np.asarray(st["present"]).reshape(B, A) != 0
    c, s = np.cos(psi), np.sin(psi)
    dx, dy = x[:, None, :] - x[:, :, None], y[:, None, :] - y[:, :, None]            # [B, i, j]: j relative to i
    ci, si, cj, sj = c[:, :, None], s[:, :, None], c[:, None, :], s[:, None, :]
    hli, hwi, hlj, hwj = hl[:, :, None], hw[:, :, None], hl[:, None, :], hw[:, None, :]
    cc, ss = np.abs(ci * cj + si * sj), np.abs(ci * sj - si * cj)
    a0 = np.abs(dx * ci + dy * si) - (hli + hlj * cc + hwj * ss)
    a1 = np.abs(dy * ci - dx * si) - (hwi + hlj * ss + hwj * cc)
    a2 = np.abs(dx * cj + dy * sj) - (hlj + hli * cc + hwi * ss)
    a3 = np.abs(dy * cj - dx * sj) - (hwj + hli * ss + hwi * cc)
    slack = np.maximum(np.maximum(a0, a1), np.maximum(a2, a3))
    ok = present[:, :, None] & present[:, None, :] & ~np.eye(A, dtype=bool)[None]
    return np.where(ok, np.abs(slack), np.inf).min(2).reshape(-1)


def _seg_d2(px, py, ax, ay, bx, by):
    abx, aby = bx - ax, by - ay
    apx, apy = px[:, None] - ax, py[:, None] - ay
    t = np.clip((apx * abx + apy * aby) / np.maximum(abx * abx + aby * aby, 1e-300), 0.0, 1.0)
    qx, qy = apx - t * abx, apy - t * aby
    return qx * qx + qy * qy


def _mesh_dist(px, py, tri):
    """distance of points (float64 [n]) to a triangle soup [T, 3, 2] (0 inside), brute force in numpy"""
    ax, ay, bx, by, cx, cy = (tri[None, :, i, j] for i in range(3) for j in range(2))
    x, y = px[:, None], py[:, None]
    e0 = (bx - ax) * (y - ay) - (by - ay) * (x - ax)
    e1 = (cx - bx) * (y - by) - (cy - by) * (x - bx)
    e2 = (ax - cx) * (y - cy) - (ay - cy) * (x - cx)
    inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
    d2 = np.minimum(np.minimum(_seg_d2(px, py, ax, ay, bx, by), _seg_d2(px, py, bx, by, cx, cy)), _seg_d2(px, py, cx, cy, ax, ay))
    return np.sqrt(np.where(inside, 0.0, d2).min(1))


def offroad_margin(st, B, A, tw, scn_map, thr):
    x, y, psi = (np.asarray(st[k], np.float64).reshape(B, A) for k in ("x", "y", "psi"))
    hl, hw = 0.5 * np.asarray(st["len"], np.float64).reshape(B, A), 0.5 * np.asarray(st["wid"], np.float64).reshape(B, A)
    c, s = np.cos(psi), np.sin(psi)
    out = np.full((B, A), np.inf)
    maps = scn_map[np.asarray(st["scn"]).astype(np.int64)]
    for m in np.unique(maps):
        e = np.flatnonzero(maps == m)
        tri = tw.tris[int(m)].numpy().astype(np.float64)
        best = np.full((len(e), A), np.inf)
        for sl, sw in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
            px = x[e] + sl * hl[e] * c[e] - sw * hw[e] * s[e]
            py = y[e] + sl * hl[e] * s[e] + sw * hw[e] * c[e]
            d = _mesh_dist(px.reshape(-1), py.reshape(-1), tri).reshape(len(e), A)
            best = np.minimum(best, np.abs(d - thr))
        out[e] = best
    return out.reshape(-1)


class Observed:
    """running maxima of what a witness test observes, written as a small JSON record"""

    def __init__(self):
        self.rec = dict(state_abs=0.0, state_rel=0.0, reward_abs=0.0, slot_steps=0, collided_in_band=0, offroad_in_band=0,
                        collided_disagree_outside_band=0, offroad_disagree_outside_band=0, collided_disagree_in_band=0, offroad_disagree_in_band=0)

    def state(self, got, ref):
        d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
        if d.size:
            self.rec["state_abs"] = max(self.rec["state_abs"], float(d.max()))
            self.rec["state_rel"] = max(self.rec["state_rel"], float((d / np.maximum(1.0, np.abs(np.asarray(ref, np.float64)))).max()))

    def reward(self, got, ref):
        self.rec["reward_abs"] = max(self.rec["reward_abs"], float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()))

    def mask(self, name, got, ref, margin, live):
        differ = (np.asarray(got) != np.asarray(ref)) & live
        inside = (margin <= BAND) & live
        self.rec[f"{name}_in_band"] += int(inside.sum())
        self.rec[f"{name}_disagree_in_band"] += int((differ & inside).sum())
        self.rec[f"{name}_disagree_outside_band"] += int((differ & ~inside).sum())
        return int((differ & ~inside).sum())

    def write(self, path, **extra):
        import json
        import os

        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dict(self.rec, band_m=BAND, **extra), f, indent=1)
