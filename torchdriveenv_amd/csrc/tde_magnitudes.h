// tde_magnitudes.h — infraction MAGNITUDES of every env's ego (tde_ego_infractions, include/tde_hip.h): what the reference's
// info dict carries under "offroad" and "collision" (ref gym_env.py:427-428: simulator.compute_offroad() / compute_collision()
// for the exposed agent; Monitor logs them, examples/rl_training.py:128), where the step path only needs `> 0`.
//
//   offroad    sum over the four box corners of clamp(dist - threshold, min = 0), dist = distance of the corner to the drivable
//              mesh (0 inside); under tde_config.offroad_threshold_squared dist is the SQUARED distance, as the threshold then is;
//   collision  sum over the other present agents whose box overlaps the ego's (strict SAT, the predicate of the collision mask) of
//              the IoU of the two boxes - the published form of CollisionMetric.nograd; also the number of such agents.
//
// Upstream's values are unpinned here (torchdrivesim absent): the CPU checker defines them (its ego-infractions restatement: brute force over
// every triangle) and this kernel returns the same bits.  Not on the step path: an env that wants magnitudes steps without
// TDE_F_AUTORESET and follows the step with tde_env_post_step - magnitudes of the envs the step flagged + the re-spawn of the finished
// ones in one launch (BatchedWaypointEnv(info_magnitudes=True)); tde_ego_infractions is the operator on an arbitrary state.
//
// One wavefront per env.  The distance of a corner needs the NEAREST triangle, which the grid index only lists for points within
// the threshold band: a corner in a FULL cell contributes 0; in a MIXED cell whose nearest candidate is within the band radius,
// that candidate is the nearest triangle; otherwise all 64 lanes scan the cells of a square around the corner for MIXED cells
// and their candidate lists - the segment from the corner to its nearest mesh point crosses a MIXED cell that lists the
// triangle it ends on (half a metre before it ends, the distance to the mesh is half a metre: neither FULL nor EMPTY) - growing
// the square until it covers the best distance found less the band the lists cover.
#pragma once
#include "tde_device.h"

namespace tde {

// IoU of two oriented boxes: Sutherland-Hodgman clipping of box 0 by the four edges of box 1 + the shoelace formula, fp32, the
// CPU checker's expression trees (the published form of CollisionMetric.nograd sums this over the other agents)
TDE_DEV void box_corners_ccw(float x, float y, float c, float s, float hl, float hw, float *px, float *py)
{
    const float lx = hl * c, ly = hl * s, wx = hw * s, wy = hw * c;
    px[0] = (x + lx) - wx; py[0] = (y + ly) + wy;
    px[1] = (x - lx) - wx; py[1] = (y - ly) + wy;
    px[2] = (x - lx) + wx; py[2] = (y - ly) - wy;
    px[3] = (x + lx) + wx; py[3] = (y + ly) - wy;
}

// `poly`: 32 floats of LDS, the two vertex lists of the clipping (the overlapping pairs of a wavefront take turns: there is rarely
// more than one).  As private arrays they are indexed dynamically and live in scratch MEMORY - every access a global-memory round trip
// on a chain of ~300 of them: 15 us for one pair of boxes, which the whole launch then waits for (profiles/r04_z_magnitudes_cost.txt).
TDE_DEV float box_iou(float x0, float y0, float c0, float s0, float hl0, float hw0, float x1, float y1, float c1,
                      float s1, float hl1, float hw1, float *poly)
{
#define TDE_AX(i) poly[(i)]
#define TDE_AY(i) poly[8 + (i)]
#define TDE_BX(i) poly[16 + (i)]
#define TDE_BY(i) poly[24 + (i)]
    float px[4], py[4], qx[4], qy[4];
    int n = 4;
    box_corners_ccw(x0, y0, c0, s0, hl0, hw0, px, py);
    box_corners_ccw(x1, y1, c1, s1, hl1, hw1, qx, qy);
#pragma unroll
    for (int i = 0; i < 4; ++i) { TDE_AX(i) = px[i]; TDE_AY(i) = py[i]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n <= 0) break;
        const float ex = qx[(e + 1) & 3] - qx[e], ey = qy[(e + 1) & 3] - qy[e];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int i2 = (i + 1 == n) ? 0 : i + 1;
            const float axi = TDE_AX(i), ayi = TDE_AY(i), axj = TDE_AX(i2), ayj = TDE_AY(i2);
            const float sp = ex * (ayi - qy[e]) - ey * (axi - qx[e]);
            const float sq = ex * (ayj - qy[e]) - ey * (axj - qx[e]);
            if (sp >= 0.0f && m < 8) { TDE_BX(m) = axi; TDE_BY(m) = ayi; ++m; }
            if (((sp > 0.0f && sq < 0.0f) || (sp < 0.0f && sq > 0.0f)) && m < 8) {
                const float t = sp / (sp - sq);
                TDE_BX(m) = axi + t * (axj - axi);
                TDE_BY(m) = ayi + t * (ayj - ayi);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { TDE_AX(i) = TDE_BX(i); TDE_AY(i) = TDE_BY(i); }
    }
    if (n < 3) return 0.0f;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        const int i2 = (i + 1 == n) ? 0 : i + 1;
        acc = acc + (TDE_AX(i) * TDE_AY(i2) - TDE_AX(i2) * TDE_AY(i));
    }
#undef TDE_AX
#undef TDE_AY
#undef TDE_BX
#undef TDE_BY
    const float ai = 0.5f * fabsf(acc);
    const float a0 = (2.0f * hl0) * (2.0f * hw0), a1 = (2.0f * hl1) * (2.0f * hw1);
    return ai / ((a0 + a1) - ai);
}

TDE_DEV float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// squared distance of the wave-uniform point (px, py) to the mesh of map m, exactly min over ALL its triangles of
// point_tri_d2 (the CPU checker's brute-force minimum); -1 when the point lies in a FULL cell (within the threshold: the
// caller's clamp is 0 and the exact value is not needed).  Every lane of the wavefront calls it.
TDE_DEV float point_mesh_d2_wave(const tde_world &w, const tde_map &m, float px, float py, float band2, int lane)
{
    const float fx = __builtin_amdgcn_fmed3f((px - m.ox) * m.inv_cell, 0.0f, (float)(m.nx - 1));
    const float fy = __builtin_amdgcn_fmed3f((py - m.oy) * m.inv_cell, 0.0f, (float)(m.ny - 1));
    const int ix = (int)fx, iy = (int)fy;
    const uint32_t wd = w.cell_word[(uint32_t)m.cell_base + (((uint32_t)iy << m.row_shift) + (uint32_t)ix)];
    const uint32_t cls = wd & 3u;
    // (a point outside the grid was clamped into a border cell: EMPTY, and the scan below is about distances, not cells)
    const bool inside = px >= m.ox && py >= m.oy && px < m.ox + (float)m.nx * m.cell && py < m.oy + (float)m.ny * m.cell;
    if (cls == TDE_CELL_FULL && inside) return -1.0f;
    const float4 *recs = reinterpret_cast<const float4 *>(w.cell_tri) + 3 * (size_t)(uint32_t)m.rec_base;
    float best = 3.0e38f;
    if (cls == TDE_CELL_MIXED && inside) {
        const int n = (int)((wd >> 2) & 255u);
        for (int k = lane; k < n; k += 64) best = fminf(best, point_tri_d2_packed(px, py, recs + 3 * (size_t)((wd >> 10) + (uint32_t)k)));
        best = wave_min(best);
        if (best <= band2) return best;          // every triangle this close to a point of the cell is in the cell's list
    }
    // Scan squares of growing half width (in cells) until the square is known to hold a cell that lists the nearest triangle T:
    // with q = T's point nearest to p and d = |p - q| <= sqrt(best), the point of the segment p -> q at distance `band` from q
    // lies in a cell that lists T (the lists cover `band` around every point of a cell), at most d - band from p.
    // (round 4's first form started at 2 m and asked for sqrt(best) + 1 m: 441 cells for a corner a metre off the road where 81 +
    //  121 do, and the launch waits for its slowest wavefront: 21 -> 12 us per call at 8192 envs)
    const float bandw = __builtin_sqrtf(band2);
    float cover = ((cls == TDE_CELL_EMPTY && inside) ? (float)((wd >> 2) & 255u) * TDE_CLEARANCE_UNIT : 0.0f) + 0.5f;
    for (;;) {
        const int hw = (int)(cover * m.inv_cell) + 2;
        const int x0 = max(ix - hw, 0), x1 = min(ix + hw, m.nx - 1), y0 = max(iy - hw, 0), y1 = min(iy + hw, m.ny - 1);
        const int nxs = x1 - x0 + 1, ncell = nxs * (y1 - y0 + 1);
        float b = 3.0e38f;
        for (int c0 = 0; c0 < ncell; c0 += 64) {
            const int c = c0 + lane;
            const int cy = y0 + c / nxs, cx = x0 + c % nxs;
            uint32_t cw = 0u;                                        // (class EMPTY: nothing to do)
            if (c < ncell) cw = w.cell_word[(uint32_t)m.cell_base + (((uint32_t)cy << m.row_shift) + (uint32_t)cx)];
            // neighbouring cells mostly SHARE their list (world.py stores equal lists once): a cell whose word equals that of the
            // cell to its left or above it in this batch of 64 leaves the list to that one
            const uint32_t left = (uint32_t)__shfl_up((int)cw, 1), up = (uint32_t)__shfl_up((int)cw, nxs < 64 ? nxs : 0);
            const bool dup = (lane >= 1 && cx > x0 && left == cw) || (nxs < 64 && lane >= nxs && up == cw);
            if ((cw & 3u) != TDE_CELL_MIXED || dup) continue;
            const int n = (int)((cw >> 2) & 255u);
            for (int k = 0; k < n; k += 2) {                         // two records per trip: their loads and tests side by side
                const float d0 = point_tri_d2_packed(px, py, recs + 3 * (size_t)((cw >> 10) + (uint32_t)k));
                const float d1 = point_tri_d2_packed(px, py, recs + 3 * (size_t)((cw >> 10) + (uint32_t)(k + 1 < n ? k + 1 : k)));
                b = fminf(b, fminf(d0, d1));
            }
        }
        best = fminf(best, wave_min(b));
        const bool whole = x0 == 0 && y0 == 0 && x1 == m.nx - 1 && y1 == m.ny - 1;
        if (whole) break;                                            // every cell of the map was looked at
        if (best < 3.0e38f) {
            const float need = __builtin_sqrtf(best) - bandw + (m.cell + 0.05f);
            // the square must reach `need` metres from the point in every direction that stays inside the grid
            const float have = ((float)hw - 1.0f) * m.cell;
            if (have >= need) break;
            cover = need;
        } else {
            cover = 2.0f * cover + 1.0f;
        }
    }
    return best;
}

// ---- the two magnitudes in pieces: one wavefront each ---------------------------------------------------------------------------
struct EgoBox { float x, y, c, s, hl, hw; };
TDE_DEV EgoBox ego_box(const tde_state &st, int64_t g0)
{
    EgoBox b;
    b.x = st.x[g0]; b.y = st.y[g0];
    sincos_f32(st.psi[g0], b.s, b.c);
    b.hl = 0.5f * st.len[g0]; b.hw = 0.5f * st.wid[g0];
    return b;
}

// collision: (sum of the IoUs with the overlapping agents, in slot order; their number).  iou_of: A floats, poly: 32 floats of
// LDS of this wavefront
TDE_DEV float2 ego_collision_mag(const tde_state &st, int64_t g0, const EgoBox &eb, int lane, float *iou_of, float *poly)
{
    const int A = st.A;
    int nhit = 0;
    for (int j0 = 0; j0 < A; j0 += 64) {                             // lanes take the other slots, 64 at a time
        const int j = j0 + lane;
        bool hit = false;
        float v = 0.0f;
        if (j > 0 && j < A && st.present[g0 + j]) {
            float sj, cj;
            sincos_f32(st.psi[g0 + j], sj, cj);
            const float xj = st.x[g0 + j], yj = st.y[g0 + j], hlj = 0.5f * st.len[g0 + j], hwj = 0.5f * st.wid[g0 + j];
            hit = obb_overlap(eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, xj, yj, cj, sj, hlj, hwj);
        }
        const unsigned long long hm = __ballot(hit);
        for (unsigned long long rest = hm; rest; rest &= rest - 1) {    // one overlapping pair at a time through the 32 floats of LDS
            if (lane == __ffsll((long long)rest) - 1) {
                float sj, cj;
                sincos_f32(st.psi[g0 + j], sj, cj);
                v = box_iou(eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, st.x[g0 + j], st.y[g0 + j], cj, sj, 0.5f * st.len[g0 + j], 0.5f * st.wid[g0 + j], poly);
            }
        }
        if (j < A) iou_of[j] = v;
        nhit += (int)__popcll(hm);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wavefront's own LDS stores have landed
    float cmag = 0.0f;
    if (nhit) for (int j = 1; j < A; ++j) { const float v = iou_of[j]; if (v != 0.0f) cmag = cmag + v; }
    return make_float2(cmag, (float)nhit);
}

// offroad: the term of corner c (0..3) of the ego's box - clamp(dist - threshold, 0) - by the whole wavefront
TDE_DEV float ego_corner_term(const tde_config &cfg, const tde_world &w, const tde_map &m, const EgoBox &eb, int c, int lane)
{
    const float thr = cfg.offroad_threshold, thr2 = thr2_of(cfg);
    const float band = __builtin_sqrtf(thr2) + 0.04f;            // (the grid's lists cover threshold + 0.05: world.py GRID_MARGIN)
    Corners k;
    offroad_issue<false>(w, m, false, eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, k);   // (corner coordinates only)
    const float px = c == 0 ? k.px0 : c == 1 ? k.px1 : c == 2 ? k.px2 : k.px3;
    const float py = c == 0 ? k.py0 : c == 1 ? k.py1 : c == 2 ? k.py2 : k.py3;
    const float d2 = point_mesh_d2_wave(w, m, px, py, band * band, lane);
    if (d2 < 0.0f) return 0.0f;                                      // inside the mesh
    const float dist = cfg.offroad_threshold_squared ? d2 : __builtin_sqrtf(d2);
    return fmaxf(dist - thr, 0.0f);
}

// out[e] = (offroad magnitude, collision magnitude = sum of IoUs, number of overlapping agents, 0) of env e's ego on the CURRENT
// state, whatever its flags say; one wavefront per env (every env has work: the operator form, tde_ego_infractions)
__global__ __launch_bounds__(kBlock) void ego_infractions_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    __shared__ float iou_of[kBlock / kWave][TDE_MAX_AGENTS];     // per env: the IoU with every slot, summed in slot order
    __shared__ float poly[kBlock / kWave][32];                     // per wavefront: box_iou's vertex lists
    const int lane = (int)(threadIdx.x & 63u);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (e >= st.B) return;                                           // (wave-uniform)
    const int64_t g0 = (int64_t)e * st.A;
    float omag = 0.0f;
    float2 cm = make_float2(0.0f, 0.0f);
    if (st.present[g0]) {
        const EgoBox eb = ego_box(st, g0);
        cm = ego_collision_mag(st, g0, eb, lane, iou_of[threadIdx.x >> 6], poly[threadIdx.x >> 6]);
        if (cfg.flags & TDE_F_OFFROAD) {
            const tde_map m = w.maps[reinterpret_cast<const int4 *>(w.scn)[st.scn[e]].x];
#pragma unroll
            for (int c = 0; c < 4; ++c) omag = omag + ego_corner_term(cfg, w, m, eb, c, lane);
        }
    }
    if (lane == 0) reinterpret_cast<float4 *>(out)[e] = make_float4(omag, cm.x, cm.y, 0.0f);
}

// tde_env_post_step: what follows a step that was launched WITHOUT TDE_F_AUTORESET, in one launch -
// (a) out[e] = the magnitudes of the ego's infractions on the state that step left, GATED by the flags it stored: a magnitude is
//     non-zero only under its flag (collision: the same predicate; offroad: a corner beyond the threshold has d^2 > thr^2, and
//     sqrt(d^2) <= thr for d^2 <= RN(thr * thr) since RN(sqrt(RN(x * x))) = x), so the 98 % of the envs without an infraction
//     are not looked at;
// (b) with TDE_F_AUTORESET the re-spawn of the envs it finished (env_reset_kernel's stores for mask = terminated | truncated)
//     and, when the state carries the compact observation, that of the new episode (state_obs_kernel's expression).
// One wavefront per env (a workgroup of four wavefronts per env - a corner each - was tried: 32 768 wavefronts to launch for the
// ~150 that have work, 36 vs 31 us per step of the env; profiles/r04_z_magnitudes_cost.txt).
template <int A>
__global__ __launch_bounds__(kBlock) void env_post_step_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    __shared__ float iou_of[kBlock / kWave][TDE_MAX_AGENTS];
    __shared__ float poly[kBlock / kWave][32];
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + wv);
    if (e >= st.B) return;                                           // (wave-uniform, like every branch below but the slot guards)
    const int64_t g0 = (int64_t)e * A;
    if (out) {
        float omag = 0.0f;
        float2 cm = make_float2(0.0f, 0.0f);
        const bool do_coll = st.present[g0] && st.collided[g0] != 0;
        const bool do_off = st.present[g0] && (cfg.flags & TDE_F_OFFROAD) && st.offroad[g0] != 0;
        if (do_coll || do_off) {
            const EgoBox eb = ego_box(st, g0);
            if (do_coll) cm = ego_collision_mag(st, g0, eb, lane, iou_of[wv], poly[wv]);
            if (do_off) {
                const tde_map m = w.maps[reinterpret_cast<const int4 *>(w.scn)[st.scn[e]].x];
#pragma unroll
                for (int c = 0; c < 4; ++c) omag = omag + ego_corner_term(cfg, w, m, eb, c, lane);
            }
        }
        if (lane == 0) reinterpret_cast<float4 *>(out)[e] = make_float4(omag, cm.x, cm.y, 0.0f);
    }
    if (!(cfg.flags & TDE_F_AUTORESET) || !(st.terminated[e] | st.truncated[e])) return;
    Cold cold;
    fill_cold(cold, cfg, w);
    const int episode = st.episode[e];
    for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + lane;
        if (a >= A) continue;
        Agent ag;
        EnvRegs er{0, 0, 0, 0, episode};
        reset_lane<A, false>(cfg, cold, e, a, ag, er);
        const int64_t g = g0 + a;
        store_agent_dynamic(st, g, ag);
        store_agent_static(st, g, ag);
        st.collided[g] = 0;
        st.offroad[g] = 0;
        if (a == 0) {
            st.scn[e] = er.scn; st.steps[e] = 0; st.target_idx[e] = 1; st.reached[e] = 0; st.episode[e] = er.episode;
            if (st.ep_return) st.ep_return[e] = 0.0;
            if (st.obs) {
                const bool has = 1 < reinterpret_cast<const int4 *>(w.scn)[er.scn].y;
                float fwd = 0.0f, lat = 0.0f;
                if (has) {
                    const double2 t = reinterpret_cast<const double2 *>(w.wp_xy)[(int64_t)er.scn * w.NW + 1];
                    float s, c;
                    sincos_f32(ag.psi, s, c);
                    const float dx = (float)t.x - ag.x, dy = (float)t.y - ag.y;
                    fwd = dx * c + dy * s;
                    lat = dy * c - dx * s;
                }
                float4 *ob = reinterpret_cast<float4 *>(st.obs) + 2 * (int64_t)e;
                ob[0] = make_float4(ag.x, ag.y, ag.psi, ag.v);
                ob[1] = make_float4(fwd, lat, has ? 1.0f : 0.0f, 0.0f);
            }
        }
    }
}

}  // namespace tde
