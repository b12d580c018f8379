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

__device__ __noinline__ float box_iou(float x0, float y0, float c0, float s0, float hl0, float hw0, float x1, float y1, float c1,
                                      float s1, float hl1, float hw1)
{
    float ax[8], ay[8], bx[8], by[8], qx[4], qy[4];
    int n = 4;
    box_corners_ccw(x0, y0, c0, s0, hl0, hw0, ax, ay);
    box_corners_ccw(x1, y1, c1, s1, hl1, hw1, qx, qy);
    for (int e = 0; e < 4 && n > 0; ++e) {
        const float ex = qx[(e + 1) & 3] - qx[e], ey = qy[(e + 1) & 3] - qy[e];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int i2 = (i + 1 == n) ? 0 : i + 1;
            const float sp = ex * (ay[i] - qy[e]) - ey * (ax[i] - qx[e]);
            const float sq = ex * (ay[i2] - qy[e]) - ey * (ax[i2] - qx[e]);
            if (sp >= 0.0f && m < 8) { bx[m] = ax[i]; by[m] = ay[i]; ++m; }
            if (((sp > 0.0f && sq < 0.0f) || (sp < 0.0f && sq > 0.0f)) && m < 8) {
                const float t = sp / (sp - sq);
                bx[m] = ax[i] + t * (ax[i2] - ax[i]);
                by[m] = ay[i] + t * (ay[i2] - ay[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { ax[i] = bx[i]; ay[i] = by[i]; }
    }
    if (n < 3) return 0.0f;
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) {
        const int i2 = (i + 1 == n) ? 0 : i + 1;
        acc = acc + (ax[i] * ay[i2] - ax[i2] * ay[i]);
    }
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

// out[e] = (offroad magnitude, collision magnitude = sum of IoUs, number of overlapping agents, 0) of env e's ego; one wavefront per env
__global__ __launch_bounds__(kBlock) void ego_infractions_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    __shared__ float iou_of[kBlock / kWave][TDE_MAX_AGENTS];     // per env: the IoU with every slot, summed in slot order by lane 0
    const int lane = (int)(threadIdx.x & 63u);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (e >= st.B) return;                                           // (wave-uniform)
    const int A = st.A;
    const int64_t g0 = (int64_t)e * A;
    float omag = 0.0f, cmag = 0.0f, nmag = 0.0f;
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
            float v = 0.0f;
            if (j > 0 && j < A && st.present[g0 + j]) {
                float sj, cj;
                sincos_f32(st.psi[g0 + j], sj, cj);
                const float xj = st.x[g0 + j], yj = st.y[g0 + j], hlj = 0.5f * st.len[g0 + j], hwj = 0.5f * st.wid[g0 + j];
                hit = obb_overlap(ex, ey, ce, se, hl, hw, xj, yj, cj, sj, hlj, hwj);
                if (hit) v = box_iou(ex, ey, ce, se, hl, hw, xj, yj, cj, sj, hlj, hwj);
            }
            if (j < A) iou_of[threadIdx.x >> 6][j] = v;
            nhit += (int)__popcll(__ballot(hit));
        }
        nmag = (float)nhit;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wavefront's own LDS stores have landed
        if (nhit) for (int j = 1; j < A; ++j) { const float v = iou_of[threadIdx.x >> 6][j]; if (v != 0.0f) cmag = cmag + v; }
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
    if (lane == 0) reinterpret_cast<float4 *>(out)[e] = make_float4(omag, cmag, nmag, 0.0f);
}

}  // namespace tde
