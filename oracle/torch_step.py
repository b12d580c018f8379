"""Batched-tensor CPU form of the env step: the SAME algorithm as oracle/tde_oracle.c restated as B x A
torch ops on the host, the way a PyTorch simulator (torchdrivesim) evaluates a batch.  TEST INFRASTRUCTURE ONLY: it is the
second CPU baseline of bench.py (`cpu_baseline.batched_torch_*`) next to the scalar C oracle, and tests/ checks it against
that oracle step by step (libm-class sin / cos instead of the oracle's polynomial, so state agrees to ~1e-5 and masks
away from their decision boundaries).  Traffic lights are not restated here (the bench workload runs without them).

Reference anchors are those of the oracle: gym_env.py:369-389 (step order), :391-437 (reward / termination / info),
:275-283 (replay), torchdrivesim's KinematicBicycle / collision / offroad as restated in tde_oracle.c."""
import numpy as np
import torch

from torchdriveenv_amd import _abi


class TorchWorld:
    """host tensors of the static tables, per map triangle lists included"""

    def __init__(self, world):
        a = world.arrays
        self.A = world.A
        self.scn_map = torch.from_numpy(a["scn"]["map"].astype(np.int64))
        self.scn_wp_n = torch.from_numpy(a["scn"]["wp_n"].astype(np.int64))
        NW = world.ints["NW"]
        self.wp_xy = torch.from_numpy(np.asarray(a["wp_xy"], np.float64).reshape(-1, NW, 2))
        sp = a["spawn"].reshape(-1, self.A)
        self.sp = {k: torch.from_numpy(sp[k].astype(np.int64)) for k in ("route", "route_n", "replay", "replay_len")}
        RW, RT = max(1, world.ints["RW"]), max(1, world.ints["RT"])
        self.route_xy = torch.from_numpy(np.asarray(a["route_xy"], np.float32).reshape(-1, RW, 2))
        self.replay = torch.from_numpy(np.asarray(a["replay_states"], np.float32).reshape(-1, RT, 4))
        tri = np.asarray(a["tri"], np.float32).reshape(-1, 3, 2)
        self.tris = [torch.from_numpy(tri[m["tri_base"]:m["tri_base"] + m["n_tri"]].copy()) for m in a["maps"]]


def _seg_d2(p, a, b):
    ab, ap = b - a, p[:, None, :] - a
    len2 = (ab * ab).sum(-1)
    t = ((ap * ab).sum(-1) / len2.clamp_min(1e-30)).clamp(0.0, 1.0)
    q = ap - t[..., None] * ab
    return (q * q).sum(-1)


def point_mesh_d2(p, tri, chunk=16384):
    """squared distance of points [n,2] to a triangle soup [T,3,2] (0 inside), brute force"""
    out = torch.empty(len(p), dtype=p.dtype)
    a, b, c = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
    for i in range(0, len(p), chunk):
        q = p[i:i + chunk]
        qq = q[:, None, :]
        e0 = (b[..., 0] - a[..., 0]) * (qq[..., 1] - a[..., 1]) - (b[..., 1] - a[..., 1]) * (qq[..., 0] - a[..., 0])
        e1 = (c[..., 0] - b[..., 0]) * (qq[..., 1] - b[..., 1]) - (c[..., 1] - b[..., 1]) * (qq[..., 0] - b[..., 0])
        e2 = (a[..., 0] - c[..., 0]) * (qq[..., 1] - c[..., 1]) - (a[..., 1] - c[..., 1]) * (qq[..., 0] - c[..., 0])
        inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
        d = torch.minimum(torch.minimum(_seg_d2(q, a, b), _seg_d2(q, b, c)), _seg_d2(q, c, a))
        out[i:i + chunk] = torch.where(inside, torch.zeros_like(d), d).min(1).values
    return out


def torch_env_step(cfg, world, tw, hs, oracle_reset=None):
    """one timestep of every env of `hs` (host EnvState), in place; hs['action'] holds the ego actions"""
    B, A = hs.B, hs.A
    F = int(cfg.flags)
    T = lambda k: torch.from_numpy(hs[k])                                     # noqa: E731  (views: writes land in hs)
    x, y, psi, v = (T(k).view(B, A) for k in ("x", "y", "psi", "v"))
    L, W, lr, vdes = (T(k).view(B, A) for k in ("len", "wid", "lr", "vdes"))
    present = T("present").view(B, A).bool()
    route_wp = T("route_wp").view(B, A)
    scn = T("scn").long()
    steps = T("steps")
    slot = torch.arange(A)[None, :].expand(B, A)
    pre = torch.stack([x[:, 0], y[:, 0], psi[:, 0], v[:, 0]], -1).clone()
    steps += 1                                                               # :116
    k = steps.long()
    c, s = torch.cos(psi), torch.sin(psi)
    route, route_n = tw.sp["route"][scn], tw.sp["route_n"][scn]
    replay, replay_len = tw.sp["replay"][scn], tw.sp["replay_len"][scn]
    dt, amax, smax = float(cfg.dt), float(cfg.npc_max_accel), float(cfg.npc_max_steer)
    act = T("action").view(B, 2)
    acc = torch.zeros(B, A)
    beta = torch.zeros(B, A)
    if F & _abi.F_NPC:
        npc = present & (slot > 0)
        has = npc & (route >= 0) & (route_wp.long() < route_n)
        tg = tw.route_xy[route.clamp_min(0), route_wp.long().clamp(0, tw.route_xy.shape[1] - 1)]
        dx, dy = tg[..., 0] - x, tg[..., 1] - y
        fwd, lat = dx * c + dy * s, dy * c - dx * s
        sin_err = lat / torch.sqrt(dx * dx + dy * dy).clamp_min(1e-3)
        b_t = torch.where(fwd < 0, torch.copysign(torch.full_like(lat, smax), lat), (cfg.npc_k_steer * sin_err).clamp(-smax, smax))
        ex, ey = x[:, None, :] - x[:, :, None], y[:, None, :] - y[:, :, None]      # [B, i, j] = (j) - (i)
        ci, si = c[:, :, None], s[:, :, None]
        fj, lj = ex * ci + ey * si, ey * ci - ex * si
        halfw = (cfg.npc_lane_half + 0.5 * W)[:, None, :]
        al = lj.abs()
        hd = ci * c[:, None, :] + si * s[:, None, :]
        jj, ii = slot[:, None, :], slot[:, :, None]
        cone = (jj < ii) & (fj < cfg.npc_cone_range) & (al < halfw + cfg.npc_cone_k * fj) & (hd > -0.5)
        ok = (fj > 0) & ((al < halfw) | cone) & present[:, None, :] & (jj != ii)
        g = fj - 0.5 * (L[:, :, None] + L[:, None, :])
        gap = torch.where(ok, g, torch.full_like(g, 1e30)).min(-1).values
        vd = torch.minimum(vdes, torch.sqrt(amax * (gap - cfg.npc_gap_s0).clamp_min(0.0)))
        a_t = (cfg.npc_k_speed * (vd - v)).clamp(-amax, amax)
        a_n = (cfg.npc_k_speed * (0.0 - v)).clamp(-amax, amax)
        acc = torch.where(has, a_t, torch.where(npc, a_n, acc))
        beta = torch.where(has, b_t, beta)
        first = (k == 1)[:, None] & (not (F & _abi.F_NPC_FIRST_STEP))         # first step of an episode: the NPCs coast
        acc = torch.where(first, torch.zeros_like(acc), acc)
        beta = torch.where(first, torch.zeros_like(beta), beta)
    acc[:, 0], beta[:, 0] = act[:, 0], act[:, 1]
    # R4 bicycle
    v1 = v + acc * dt
    x1 = x + v1 * torch.cos(psi + beta) * dt
    y1 = y + v1 * torch.sin(psi + beta) * dt
    p1 = psi + (v1 / lr) * torch.sin(beta) * dt
    p1 = torch.remainder(np.float32(np.pi) + p1, np.float32(2 * np.pi)) - np.float32(np.pi)
    if F & _abi.F_REPLAY:
        on = present & (slot > 0) & (replay >= 0) & (k[:, None] < replay_len)
        rs = tw.replay[replay.clamp_min(0), k.clamp(0, tw.replay.shape[1] - 1)[:, None].expand(B, A)]
        x1, y1, p1, v1 = (torch.where(on, rs[..., i], t) for i, t in enumerate((x1, y1, p1, v1)))
    for dst, src in ((x, x1), (y, y1), (psi, p1), (v, v1)):
        dst.copy_(torch.where(present, src, dst))
    if F & _abi.F_NPC:
        d2 = (tg[..., 0] - x) ** 2 + (tg[..., 1] - y) ** 2
        route_wp += (has & (d2 < cfg.npc_reach * cfg.npc_reach)).to(route_wp.dtype)
    # R9 collision: 4-axis SAT on all pairs
    c, s = torch.cos(psi), torch.sin(psi)
    hl, hw = 0.5 * L, 0.5 * W
    dx, dy = x[:, None, :] - x[:, :, None], y[:, None, :] - y[:, :, None]
    ci, si, cj, sj = c[:, :, None], s[:, :, None], c[:, None, :], s[:, None, :]
    cc, ss = (ci * cj + si * sj).abs(), (ci * sj - si * cj).abs()
    hli, hwi, hlj, hwj = hl[:, :, None], hw[:, :, None], hl[:, None, :], hw[:, None, :]
    ov = ((dx * ci + dy * si).abs() < hli + (hlj * cc + hwj * ss)) & ((dy * ci - dx * si).abs() < hwi + (hlj * ss + hwj * cc)) & \
         ((dx * cj + dy * sj).abs() < hlj + (hli * cc + hwi * ss)) & ((dy * cj - dx * sj).abs() < hwj + (hli * ss + hwi * cc))
    ov &= present[:, :, None] & present[:, None, :] & (slot[:, None, :] != slot[:, :, None])
    collided = ov.any(-1)
    T("collided").view(B, A).copy_(collided.to(torch.uint8))
    # R10 offroad: the four corners against every triangle of the env's map
    offroad = torch.zeros(B, A, dtype=torch.bool)
    if F & _abi.F_OFFROAD:
        lx, ly, wx, wy = hl * c, hl * s, hw * s, hw * c
        cx = torch.stack([(x + lx) - wx, (x + lx) + wx, (x - lx) + wx, (x - lx) - wx], -1)
        cy = torch.stack([(y + ly) + wy, (y + ly) - wy, (y - ly) - wy, (y - ly) + wy], -1)
        thr2 = cfg.offroad_threshold if cfg.offroad_threshold_squared else cfg.offroad_threshold ** 2
        emap = tw.scn_map[scn]
        for m, tri in enumerate(tw.tris):
            sel = emap == m
            if sel.any():
                pts = torch.stack([cx[sel].reshape(-1), cy[sel].reshape(-1)], -1)
                offroad[sel] = (point_mesh_d2(pts, tri).view(-1, A, 4) > thr2).any(-1)
        offroad &= present
    T("offroad").view(B, A).copy_(offroad.to(torch.uint8))
    # R6-R8, R11, R12: float64 on the fp32 ego state
    if F & _abi.F_REWARD:
        post = torch.stack([x[:, 0], y[:, 0], psi[:, 0], v[:, 0]], -1)
        d = torch.sqrt((post[:, 0].double() - pre[:, 0].double()) ** 2 + (post[:, 1].double() - pre[:, 1].double()) ** 2)
        dist_r = torch.where(d > cfg.distance_cutoff, cfg.distance_bonus, 0.0).double()
        psi_r = (1.0 - torch.cos((post[:, 2] - pre[:, 2]).double())) * (-cfg.heading_penalty)
        ti, reached = T("target_idx"), T("reached")
        n_wp = tw.scn_wp_n[scn]
        has_t = ti.long() < n_wp
        w = tw.wp_xy[scn, ti.long().clamp(0, tw.wp_xy.shape[1] - 1)]
        reach = has_t & (torch.sqrt((post[:, 0].double() - w[:, 0]) ** 2 + (post[:, 1].double() - w[:, 1]) ** 2) < cfg.reach_radius)
        reached += reach.to(reached.dtype)
        T("reward").copy_(((torch.where(reach, cfg.waypoint_bonus, 0.0).double() + dist_r) + psi_r).float())
        term = (offroad[:, 0] | collided[:, 0]) & bool(cfg.terminated_at_infraction)
        trunc = steps >= cfg.max_steps
        T("terminated").copy_(term.to(torch.uint8))
        T("truncated").copy_(trunc.to(torch.uint8))
        if hs["info"] is not None:
            inf = T("info")
            inf[:, 0] = ((pre[:, 2] - post[:, 2]) / np.float32(0.1)).abs().double()
            inf[:, 1] = ((pre[:, 3] - post[:, 3]) / np.float32(0.1)).abs().double()
            inf[:, 2], inf[:, 3] = psi_r, dist_r
            T("info_reached").copy_(reached)
        ti += reach.to(ti.dtype)
        if (F & _abi.F_AUTORESET) and oracle_reset is not None:
            done = (term | trunc)
            if done.any():
                oracle_reset(cfg, world, hs, done.to(torch.uint8).numpy())
