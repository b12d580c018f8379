// tde_magnitudes.h — infraction MAGNITUDES of every env's ego (tde_ego_infractions, include/tde_hip.h): what the reference's
// info dict carries under "offroad" and "collision" (ref gym_env.py:427-428: simulator.compute_offroad() / compute_collision()
// for the exposed agent; Monitor logs them, examples/rl_training.py:128), where the step path only needs `> 0`.
//
//   offroad    sum over the four box corners of clamp(dist - threshold, min = 0), dist = distance of the corner to the drivable
//              mesh (0 inside); under tde_config.offroad_threshold_squared dist is the SQUARED distance, as the threshold then is;
//   collision  number of other present agents whose box overlaps the ego's (strict SAT, the predicate of the collision mask).
//
// Upstream's values are unpinned here (torchdrivesim absent): the CPU checker defines them (its ego-infractions restatement: brute force over
// every triangle) and this kernel returns the same bits.  Not on the step path: an env that wants magnitudes steps without
// TDE_F_AUTORESET, calls this, then re-spawns the finished envs with tde_env_reset (BatchedWaypointEnv(info_magnitudes=True)).
//
// One wavefront per env.  The distance of a corner needs the NEAREST triangle, which the grid index only lists for points within
// the threshold band: a corner in a FULL cell contributes 0; in a MIXED cell whose nearest candidate is within the band radius,
// that candidate is the nearest triangle; otherwise all 64 lanes scan the cells of a square around the corner for MIXED cells
// and their candidate lists - the segment from the corner to its nearest mesh point crosses a MIXED cell that lists the
// triangle it ends on (half a metre before it ends, the distance to the mesh is half a metre: neither FULL nor EMPTY) - growing
// the square until it covers the best distance found plus a metre.
#pragma once
#include "tde_device.h"

namespace tde {

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
    // scan squares of growing half width (in cells) until the square covers sqrt(best) + 1 m
    float cover = (cls == TDE_CELL_EMPTY && inside) ? (float)((wd >> 2) & 255u) * TDE_CLEARANCE_UNIT + 2.0f : 2.0f;
    for (;;) {
        const int hw = (int)(cover * m.inv_cell) + 2;
        const int x0 = max(ix - hw, 0), x1 = min(ix + hw, m.nx - 1), y0 = max(iy - hw, 0), y1 = min(iy + hw, m.ny - 1);
        const int nxs = x1 - x0 + 1, ncell = nxs * (y1 - y0 + 1);
        float b = 3.0e38f;
        for (int c = lane; c < ncell; c += 64) {
            const int cy = y0 + c / nxs, cx = x0 + c % nxs;
            const uint32_t cw = w.cell_word[(uint32_t)m.cell_base + (((uint32_t)cy << m.row_shift) + (uint32_t)cx)];
            if ((cw & 3u) != TDE_CELL_MIXED) continue;
            const int n = (int)((cw >> 2) & 255u);
            for (int k = 0; k < n; ++k) b = fminf(b, point_tri_d2_packed(px, py, recs + 3 * (size_t)((cw >> 10) + (uint32_t)k)));
        }
        best = fminf(best, wave_min(b));
        const bool whole = x0 == 0 && y0 == 0 && x1 == m.nx - 1 && y1 == m.ny - 1;
        if (whole) break;                                            // every cell of the map was looked at
        if (best < 3.0e38f) {
            const float need = __builtin_sqrtf(best) + 1.0f;
            // the square must reach `need` metres from the point in every direction that stays inside the grid
            const float have = ((float)hw - 1.0f) * m.cell;
            if (have >= need) break;
            cover = need;
        } else {
            cover = 2.0f * cover + 4.0f;
        }
    }
    return best;
}

// out[e] = (offroad magnitude, collision magnitude) of env e's ego; one wavefront per env
__global__ __launch_bounds__(kBlock) void ego_infractions_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (e >= st.B) return;                                           // (wave-uniform)
    const int A = st.A;
    const int64_t g0 = (int64_t)e * A;
    float omag = 0.0f, cmag = 0.0f;
    if (st.present[g0]) {
        const float ex = st.x[g0], ey = st.y[g0];
        float se, ce;
        sincos_f32(st.psi[g0], se, ce);
        const float hl = 0.5f * st.len[g0], hw = 0.5f * st.wid[g0];
        // collision: lanes take the other slots, 64 at a time
        int nhit = 0;
        for (int j0 = 0; j0 < A; j0 += 64) {
            const int j = j0 + lane;
            bool hit = false;
            if (j > 0 && j < A && st.present[g0 + j]) {
                float sj, cj;
                sincos_f32(st.psi[g0 + j], sj, cj);
                hit = obb_overlap(ex, ey, ce, se, hl, hw, st.x[g0 + j], st.y[g0 + j], cj, sj, 0.5f * st.len[g0 + j], 0.5f * st.wid[g0 + j]);
            }
            nhit += (int)__popcll(__ballot(hit));
        }
        cmag = (float)nhit;
        // offroad: the four corners one after the other, each by the whole wavefront
        if (cfg.flags & TDE_F_OFFROAD) {
            const tde_map m = w.maps[reinterpret_cast<const int4 *>(w.scn)[st.scn[e]].x];
            const float thr = cfg.offroad_threshold, thr2 = thr2_of(cfg);
            const float band = __builtin_sqrtf(thr2) + 0.04f;        // (the grid's lists cover threshold + 0.05: world.py GRID_MARGIN)
            Corners k;
            offroad_issue<false>(w, m, false, ex, ey, ce, se, hl, hw, k);   // (corner coordinates only)
            const float cxs[4] = {k.px0, k.px1, k.px2, k.px3}, cys[4] = {k.py0, k.py1, k.py2, k.py3};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d2 = point_mesh_d2_wave(w, m, cxs[c], cys[c], band * band, lane);
                if (d2 >= 0.0f) {
                    const float dist = cfg.offroad_threshold_squared ? d2 : __builtin_sqrtf(d2);
                    omag = omag + fmaxf(dist - thr, 0.0f);
                }
            }
        }
    }
    if (lane == 0) { out[2 * (int64_t)e] = omag; out[2 * (int64_t)e + 1] = cmag; }
}

}  // namespace tde
