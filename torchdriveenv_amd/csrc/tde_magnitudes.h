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
// every triangle) and these functions return the same bits.  Round 5: ON the step path - the one-launch step kernels write
// tde_state.magnitudes for the egos they flagged, in the tail of the launch (judge O of the three-role kernel, the ego's wavefront
// of the one-role kernel), before any in-place re-spawn; tde_ego_infractions is the operator on an arbitrary state and
// tde_env_post_step the round-4 form (step without TDE_F_AUTORESET, then magnitudes + re-spawn in a second launch).
//
// The whole wavefront works on one ego.  The distance of a corner needs the NEAREST triangle, which the cells' candidate lists
// only hold for points within the threshold band.  ABI 10: every coarse tile (4 x 4 cells) within threshold + near_range of the
// mesh carries a NEAR LIST (tde_world.tile_near; csrc/tde_gridbuild.h) - the triangles among which the nearest one of any point of
// the tile is found - so a corner is two dependent look-ups: the tile's word, then the list's records, the four corners side by
// side on 16 lanes each.  A corner whose tile has no list (farther out than near_range, or outside the grid) falls back to the
// round-4 scan: all 64 lanes walk the cells of a square around the corner for MIXED cells and their candidate lists - the segment
// from the corner to its nearest mesh point crosses a MIXED cell that lists the triangle it ends on - growing the square until it
// covers the best distance found less the band the lists cover.
#pragma once
#include "tde_device.h"

namespace tde {

// IoU of two oriented boxes: Sutherland-Hodgman clipping of box 0 by the four edges of box 1 + the shoelace formula, fp32, the
// CPU checker's expression trees (the published form of CollisionMetric.nograd sums this over the other agents).
TDE_DEV void box_corners_ccw(float x, float y, float c, float s, float hl, float hw, float *px, float *py)
{
    const float lx = hl * c, ly = hl * s, wx = hw * s, wy = hw * c;
    px[0] = (x + lx) - wx; py[0] = (y + ly) + wy;
    px[1] = (x - lx) - wx; py[1] = (y - ly) + wy;
    px[2] = (x - lx) + wx; py[2] = (y - ly) - wy;
    px[3] = (x + lx) + wx; py[3] = (y + ly) - wy;
}

template <int N> TDE_DEV float row_ror_f(float v)       // lane l of every 16-lane row receives the value of lane (l - N) mod 16 of its row
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}
TDE_DEV float readlane_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
// (this wavefront's LDS stores are ordered before its later LDS loads: the LDS unit serves a wavefront's instructions in order;
//  the fence keeps the compiler from moving them across)
TDE_DEV void wave_lds_fence() { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

// The whole wavefront calls it with wave-uniform arguments; `poly` = 32 floats of LDS of this wavefront.  The clip runs ONE
// POLYGON VERTEX PER LANE (a convex quadrilateral cut by four half planes has at most eight): every stage evaluates the vertex
// tests side by side, ranks the surviving / inserted vertices with two ballots and writes them to the other vertex list in LDS -
// four stages of ~40 instructions instead of a chain of ~300 dependent LDS round trips on one lane (round 4: box_iou on the lane of
// the overlapping slot, 15 us per pair from scratch memory, ~3 us from LDS - and the launch waits for it).  Same operations on the
// same operands in the same order per vertex, and the shoelace sum runs over the lanes in vertex order: the CPU checker's bits.
TDE_DEV float box_iou_wave(float x0, float y0, float c0, float s0, float hl0, float hw0, float x1, float y1, float c1, float s1,
                           float hl1, float hw1, int lane, float *poly)
{
    float px[4], py[4], qx[4], qy[4];
    box_corners_ccw(x0, y0, c0, s0, hl0, hw0, px, py);
    box_corners_ccw(x1, y1, c1, s1, hl1, hw1, qx, qy);
    float *cur = poly, *nxt = poly + 16;                   // vertex lists: x at [0, 8), y at [8, 16)
    if (lane < 4) { cur[lane] = TDE_SEL4(lane, px[0], px[1], px[2], px[3]); cur[8 + lane] = TDE_SEL4(lane, py[0], py[1], py[2], py[3]); }
    wave_lds_fence();
    int n = 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n <= 0) break;                                     // (wave-uniform: n comes from ballots)
        const float ex = qx[(e + 1) & 3] - qx[e], ey = qy[(e + 1) & 3] - qy[e];
        const bool act = lane < n;
        const int i = act ? lane : 0, i2 = (i + 1 == n) ? 0 : i + 1;
        const float axi = cur[i], ayi = cur[8 + i], axj = cur[i2], ayj = cur[8 + i2];
        const float sp = ex * (ayi - qy[e]) - ey * (axi - qx[e]);
        const float sq = ex * (ayj - qy[e]) - ey * (axj - qx[e]);
        const bool keep = act && sp >= 0.0f;
        const bool cross = act && ((sp > 0.0f && sq < 0.0f) || (sp < 0.0f && sq > 0.0f));
        const unsigned long long km = __ballot(keep), cm = __ballot(cross);
        const int pos = lane_prefix(km) + lane_prefix(cm);       // (survivors / crossings on the lanes below this one: v_mbcnt)
        if (keep && pos < 8) { nxt[pos] = axi; nxt[8 + pos] = ayi; }
        const int pos2 = pos + (keep ? 1 : 0);
        if (cross && pos2 < 8) {
            const float t = sp / (sp - sq);
            nxt[pos2] = axi + t * (axj - axi);
            nxt[8 + pos2] = ayi + t * (ayj - ayi);
        }
        const int total = (int)__popcll(km) + (int)__popcll(cm);
        n = total < 8 ? total : 8;
        wave_lds_fence();
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (n < 3) return 0.0f;
    const bool act = lane < n;
    const int i = act ? lane : 0, i2 = (i + 1 == n) ? 0 : i + 1;
    const float term = cur[i] * cur[8 + i2] - cur[i2] * cur[8 + i];
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) acc = acc + readlane_f(term, k);
    wave_lds_fence();                                          // (the next pair of this wavefront reuses `poly`)
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

// point -> triangle from the RAW vertices (tde_world.tri: what the CPU checker's brute force reads), the edge reciprocals formed here
// with IEEE divisions - the values the packed records carry (world.py: pack_triangles; 0 for a degenerate edge)
TDE_DEV float point_tri_d2_raw(float px, float py, const float *__restrict__ t)
{
    const float ax = t[0], ay = t[1], bx = t[2], by = t[3], cx = t[4], cy = t[5];
    const float e0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    const float e1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
    const float e2 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);
    if ((e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) || (e0 <= 0.0f && e1 <= 0.0f && e2 <= 0.0f)) return 0.0f;
    auto inv = [](float ux, float uy) { const float l2 = ux * ux + uy * uy; return l2 > 0.0f ? 1.0f / l2 : 0.0f; };
    float d = seg_d2_inv(px, py, ax, ay, bx, by, inv(bx - ax, by - ay));
    d = fminf(d, seg_d2_inv(px, py, bx, by, cx, cy, inv(cx - bx, cy - by)));
    d = fminf(d, seg_d2_inv(px, py, cx, cy, ax, ay, inv(ax - cx, ay - cy)));
    return d;
}

// the CPU checker's own definition, 64 triangles per trip: what the grid scans below switch to when the square they would have to
// walk holds more cells than (eight times) the map has triangles - a corner tens of metres off a small map (a junction, the corridor
// of a WaypointSuite scenario) is then a dozen trips instead of a thousand.  A minimum of the same values: the same bits.
TDE_DEV float point_mesh_d2_brute_wave(const tde_world &w, int tri_base, int n_tri, float px, float py, int lane)
{
    const float *T = w.tri + 6 * (size_t)(uint32_t)tri_base;
    float best = 3.0e38f;
    for (int k = lane; k < n_tri; k += 64) best = fminf(best, point_tri_d2_raw(px, py, T + 6 * (size_t)k));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
    return best;
}
TDE_DEV bool scan_prefers_brute(int n_tri, int hw) { return (long long)(2 * hw) * (2 * hw) > 8ll * (long long)n_tri; }
// (tri_base, n_tri) of the map from its descriptor: what the scans take by default; the three-role step kernel, which keeps the
// descriptor's other words in scalar registers, hands in a functor that reads the two from LDS when - if ever - they are needed
struct TriSpanOfMap {
    const tde_map &m;
    TDE_DEV int2 operator()() const { return make_int2(m.tri_base, m.n_tri); }
};

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
        if (w.tri && scan_prefers_brute(m.n_tri, hw)) return fminf(best, point_mesh_d2_brute_wave(w, m.tri_base, m.n_tri, px, py, lane));
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

// The same value with a fraction of the registers, for the fall-back INSIDE the three-role step kernel: point_mesh_d2_wave takes
// ~90 VGPRs and the kernel has 80 - 43 spilled registers and a private segment for EVERY launch (+1.3 us to dispatch), for a
// path that only a corner beyond the near lists takes.  The 64 lanes are an 8 x 8 block of cells; the square of side 2 hw cells
// around the corner's cell is walked block by block, one record per trip, and grown (scanned again from scratch: this path is
// rare and its cost is the cell words of a corner that is metres off the road anyway) until it is known to hold a cell that lists
// the nearest triangle - the covering argument of point_mesh_d2_wave.  A minimum of the same values: the same bits.
template <typename SPAN>
TDE_DEV float point_mesh_d2_scan_lean(const tde_world &w, const tde_map &m, float px, float py, float bandw, int lane, SPAN &&tri_span)
{
    const float fx = __builtin_amdgcn_fmed3f((px - m.ox) * m.inv_cell, 0.0f, (float)(m.nx - 1));
    const float fy = __builtin_amdgcn_fmed3f((py - m.oy) * m.inv_cell, 0.0f, (float)(m.ny - 1));
    const int ix = (int)fx, iy = (int)fy;
    const float4 *recs = reinterpret_cast<const float4 *>(w.cell_tri) + 3 * (size_t)(uint32_t)m.rec_base;
    // the corner's own cell: an EMPTY cell's clearance says how far the first square has to reach at least
    const uint32_t wd = w.cell_word[(uint32_t)m.cell_base + (((uint32_t)iy << m.row_shift) + (uint32_t)ix)];
    // (a point outside the grid was clamped into a border cell: EMPTY, and the scan is about distances, not cells)
    const bool inside = px >= m.ox && py >= m.oy && px < m.ox + (float)m.nx * m.cell && py < m.oy + (float)m.ny * m.cell;
    // a corner in a FULL cell lies within the threshold: its term is 0 and the squares below - which find the nearest triangle
    // through the MIXED cells between the corner and the mesh - are not for it (a world without near lists sends it here)
    if ((wd & 3u) == TDE_CELL_FULL && inside) return -1.0f;
    const float clear = ((wd & 3u) == TDE_CELL_EMPTY && inside) ? (float)((wd >> 2) & 255u) * TDE_CLEARANCE_UNIT : 0.0f;
    int hw = (((int)((clear + 0.5f) * m.inv_cell) + 2) + 3) & ~3;                  // a multiple of 4: the square is whole 8 x 8 blocks
    const int lx = lane & 7, ly = lane >> 3;
    float best = 3.0e38f;
    for (;;) {
        if (w.tri && hw >= 32) {                                     // (a square of 64 x 64 cells or more: rare, and only then is the span read)
            const int2 ts = tri_span();
            if (scan_prefers_brute(ts.y, hw)) return fminf(best, point_mesh_d2_brute_wave(w, ts.x, ts.y, px, py, lane));
        }
        float b = 3.0e38f;
        for (int by = iy - hw; by < iy + hw; by += 8)
            for (int bx = ix - hw; bx < ix + hw; bx += 8) {
                const int cx = bx + lx, cy = by + ly;
                uint32_t cw = 0u;
                if (cx >= 0 && cx < m.nx && cy >= 0 && cy < m.ny) cw = w.cell_word[(uint32_t)m.cell_base + (((uint32_t)cy << m.row_shift) + (uint32_t)cx)];
                if ((cw & 3u) != TDE_CELL_MIXED) continue;
                const int n = (int)((cw >> 2) & 255u);
                for (int k = 0; k < n; ++k) b = fminf(b, point_tri_d2_packed(px, py, recs + 3 * (size_t)((cw >> 10) + (uint32_t)k)));
            }
        best = fminf(best, wave_min(b));
        if (ix - hw <= 0 && iy - hw <= 0 && ix + hw >= m.nx && iy + hw >= m.ny) break;   // every cell of the map was looked at
        if (best < 3.0e38f) {
            const float need = __builtin_sqrtf(best) - bandw + (m.cell + 0.05f);
            if (((float)hw - 1.0f) * m.cell >= need) break;
            hw = (((int)(need * m.inv_cell) + 2) + 3) & ~3;
        } else {
            hw = 2 * hw + 4;
        }
    }
    return best;
}

// ---- the two magnitudes, by the whole wavefront for ONE ego ---------------------------------------------------------------------
struct EgoBox { float x, y, c, s, hl, hw; };

// collision: (sum of the IoUs with the overlapping agents, in slot order; their number).  `row(j, x, y, c, s, hl, hw)` fetches the box
// of slot j of the ego's env and returns whether the slot is present (global state arrays or the step kernels' LDS tile rows);
// poly: 32 floats of LDS of this wavefront.  Wave-uniform arguments except `lane`.
template <typename R>
TDE_DEV float2 ego_collision_mag_of(int A, int lane, const EgoBox &eb, R &&row, float *poly)
{
    float cmag = 0.0f;
    int nhit = 0;
    for (int j0 = 0; j0 < A; j0 += 64) {                             // lanes take the other slots, 64 at a time
        const int j = j0 + lane;
        bool hit = false;
        float xj = 0.0f, yj = 0.0f, cj = 1.0f, sj = 0.0f, hlj = 0.0f, hwj = 0.0f;
        if (j > 0 && j < A && row(j, xj, yj, cj, sj, hlj, hwj))
            hit = obb_overlap(eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, xj, yj, cj, sj, hlj, hwj);
        const unsigned long long hm = __ballot(hit);
        for (unsigned long long rest = hm; rest; rest &= rest - 1) {    // one overlapping pair at a time, in slot order (rarely > 1)
            const int src = __ffsll((long long)rest) - 1;
            const float v = box_iou_wave(eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, readlane_f(xj, src), readlane_f(yj, src),
                                         readlane_f(cj, src), readlane_f(sj, src), readlane_f(hlj, src), readlane_f(hwj, src), lane, poly);
            cmag = cmag + v;
        }
        nhit += (int)__popcll(hm);
    }
    return make_float2(cmag, (float)nhit);
}

// offroad: sum over the four corners (FL, FR, RR, RL - the order of the CPU checker's corner list) of clamp(dist - threshold, 0).
// Lanes 16 c .. 16 c + 15 take corner c: the tile word, the first 16 records of its near list (the list's length rides in the
// first record: fetched before it is known, the table ends with 16 spare records), the rest of a longer list, a minimum over the 16
// lanes; a corner without a near list goes through point_mesh_d2_wave, all 64 lanes, one such corner at a time.
// The pieces, so that a caller with time to fill (the three-role step kernel: the ego lanes fetch their corners' tile words beside
// the offroad test, the record loads of a flagged ego are issued ahead of the barrier behind which the magnitudes are due) can
// start them early: each piece of the chain tile word -> records is a round trip to HBM / the fabric for these rarely touched lines.
// the tile_near word of the coarse tile a point lies in (0: no list - also for a point outside the grid, which the clamp would put
// into a border tile it does not lie in)
TDE_DEV uint32_t near_tile_word(const tde_world &w, const tde_map &m, float px, float py)
{
    const float fx = __builtin_amdgcn_fmed3f((px - m.ox) * m.inv_cell, 0.0f, (float)(m.nx - 1));
    const float fy = __builtin_amdgcn_fmed3f((py - m.oy) * m.inv_cell, 0.0f, (float)(m.ny - 1));
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    const bool inside = px >= m.ox && py >= m.oy && px < m.ox + (float)m.nx * m.cell && py < m.oy + (float)m.ny * m.cell;
    uint32_t tw = 0u;
    if (w.tile_near && inside) tw = w.tile_near[(uint32_t)m.near_base + (iy / TDE_COARSE_CELLS) * ((uint32_t)m.nx / TDE_COARSE_CELLS) + ix / TDE_COARSE_CELLS];
    return tw;
}

// one ego's four corners in flight: lanes 16 c .. 16 c + 15 hold corner c's point, its tile word and record r = lane & 15 of its list
struct NearFetch {
    float px, py;
    uint32_t tw;
    float4 t0, t1, t2;
};
TDE_DEV bool near_full(uint32_t tw) { return tw == 0xFFFFFFFFu; }
TDE_DEV bool near_listed(uint32_t tw) { return tw != 0u && tw != 0xFFFFFFFFu; }

// fetch record r of the corner's near list (before its length is known: the table ends with 16 spare records)
TDE_DEV void near_issue(const tde_world &w, int rec_base, int lane, NearFetch &nf)
{
    const int r = lane & 15;
    const float4 *recs = reinterpret_cast<const float4 *>(w.cell_tri) + 3 * (size_t)((uint32_t)rec_base + (near_listed(nf.tw) ? nf.tw - 1u : 0u));
    nf.t0 = recs[3 * r]; nf.t1 = recs[3 * r + 1]; nf.t2 = recs[3 * r + 2];
}

// the magnitude from the fetched records: the rest of a list longer than 16, the minimum over the corner's 16 lanes, the scan for a
// corner without a list, clamp(dist - threshold, 0), the sum over the corners in the CPU checker's order
template <bool LEAN = false, typename SPAN>
TDE_DEV float near_finish(const tde_config &cfg, const tde_world &w, const tde_map &m, const NearFetch &nf, int lane, SPAN &&tri_span)
{
    const float thr = cfg.offroad_threshold, thr2 = thr2_of(cfg);
    const float band = __builtin_sqrtf(thr2) + 0.04f;            // (the cells' lists cover threshold + 0.05: world.py GRID_MARGIN)
    const int c = lane >> 4, r = lane & 15;
    const float px = nf.px, py = nf.py;
    const bool full = near_full(nf.tw), listed = near_listed(nf.tw);
    float best = 3.0e38f;
    if (__ballot(listed)) {
        // the list's length, from the lane that holds its first record (four v_readlane + selects: a ds_bpermute is an LDS round trip)
        const int n = TDE_SEL4(c, __builtin_amdgcn_readlane(__float_as_int(nf.t2.y), 0), __builtin_amdgcn_readlane(__float_as_int(nf.t2.y), 16),
                               __builtin_amdgcn_readlane(__float_as_int(nf.t2.y), 32), __builtin_amdgcn_readlane(__float_as_int(nf.t2.y), 48));
        if (listed && r < n) best = point_tri_d2_words(px, py, nf.t0, nf.t1, nf.t2);
        if (__ballot(listed && 16 < n)) {                            // (a list of more than 16 triangles: rare)
            const float4 *recs = reinterpret_cast<const float4 *>(w.cell_tri) + 3 * (size_t)((uint32_t)m.rec_base + (listed ? nf.tw - 1u : 0u));
            for (int q = 16; __ballot(listed && q < n); q += 16)
                if (listed && q + r < n) best = fminf(best, point_tri_d2_packed(px, py, recs + 3 * (size_t)(q + r)));
        }
        // minimum over the corner's 16 lanes = one DPP row: rotations by 8, 4, 2, 1 (a minimum of the same values)
        best = fminf(best, row_ror_f<8>(best)); best = fminf(best, row_ror_f<4>(best));
        best = fminf(best, row_ror_f<2>(best)); best = fminf(best, row_ror_f<1>(best));
    }
    float d2c = full ? -1.0f : best;
    const unsigned long long scan = __ballot(!full && !listed);
#pragma unroll 1
    for (int cc = 0; cc < 4; ++cc) {
        if (!((scan >> (16 * cc)) & 1ull)) continue;                 // (wave-uniform)
        const float sx = readlane_f(px, 16 * cc), sy = readlane_f(py, 16 * cc);
        float d2;
        if constexpr (LEAN) d2 = point_mesh_d2_scan_lean(w, m, sx, sy, band, lane, tri_span);
        else d2 = point_mesh_d2_wave(w, m, sx, sy, band * band, lane);
        if (c == cc) d2c = d2;
    }
    float term = 0.0f;
    if (d2c >= 0.0f) {
        const float dist = cfg.offroad_threshold_squared ? d2c : __builtin_sqrtf(d2c);
        term = fmaxf(dist - thr, 0.0f);
    }
    float omag = 0.0f;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) omag = omag + readlane_f(term, 16 * cc);
    return omag;
}

template <bool LEAN = false>
TDE_DEV float ego_offroad_mag_wave(const tde_config &cfg, const tde_world &w, const tde_map &m, const EgoBox &eb, int lane)
{
    Corners k;
    offroad_issue<false>(w, m, false, eb.x, eb.y, eb.c, eb.s, eb.hl, eb.hw, k);   // (corner coordinates only)
    const int c = lane >> 4;
    NearFetch nf;
    nf.px = TDE_SEL4(c, k.px0, k.px1, k.px2, k.px3); nf.py = TDE_SEL4(c, k.py0, k.py1, k.py2, k.py3);
    nf.tw = near_tile_word(w, m, nf.px, nf.py);
    nf.t0 = nf.t1 = nf.t2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (__ballot(near_listed(nf.tw))) near_issue(w, m.rec_base, lane, nf);
    return near_finish<LEAN>(cfg, w, m, nf, lane, TriSpanOfMap{m});
}

template <bool LEAN = false>
TDE_DEV float near_finish(const tde_config &cfg, const tde_world &w, const tde_map &m, const NearFetch &nf, int lane)
{
    return near_finish<LEAN>(cfg, w, m, nf, lane, TriSpanOfMap{m});
}

}  // namespace tde
