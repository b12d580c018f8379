// tde_raster.h — ego-centred birdview rasteriser (R13: get_obs -> simulator.render_egocentric(), ref gym_env.py:122-124,
// obs space :95; layer / palette definitions: include/tde_abi.h), ONE WAVEFRONT PER VIEW.
//
// Round 3 rewrite.  The round-2 kernel gave a view to a 256-thread workgroup: five barrier-separated passes, 16 KB of
// LDS lists per view (7 views per CU in flight) and every wavefront repeating the per-view set-up; it sat at 60 us per
// 8192 views with two thirds of its wave-cycles parked at barriers and waitcnts.  Here a view belongs to one wavefront:
// no barrier anywhere (LDS operations of a wavefront execute in program order), 5 KB of LDS and at most 64 VGPRs per view
// (32 views per CU in flight - all 8192 of BASELINE configs[4] at once - each an independent instruction stream that hides
// the others' memory phases).  raster_view() takes its agents through a small source interface (fetch / unpack), so the
// caller decides where the poses come from; render_views_kernel (tde_kernels.hip) reads them from the state arrays.  (A
// step kernel rendering from the rows it holds in LDS was costed and dropped: DESIGN.md section 5, round 3 - the step and
// the rasteriser run side by side on HIP streams instead, tde_env_step_render.)
//
// Specification (shared with oracle/tde_oracle.c: tde_render_env): pixel (r, c) is sampled at its centre,
//     u = (H/2 - 0.5) - r,  v = (W/2 - 0.5) - c,  world = ego + u * (ax, ay) + v * (bx, by)
// evaluated as fmaf(v, b, fmaf(u, a, e)); every per-object quantity tested per pixel is the affine function of (u, v) it
// is in exact arithmetic, coefficients formed once per object (box_coeffs below = the oracle's expressions), evaluated
// the same way.  Everything else here is a result-preserving shortcut:
//   * base layer hierarchically: an 8x8 block of pixel centres is uniform when the coarse tile (4 x 4 grid cells) under its
//     centre is FULL / EMPTY with a clearance of at least the block's half diagonal (tde_world.cell_coarse); blocks that are not split
//     into 4x4; the 4x4 blocks that still straddle a road edge are listed and their pixels take the class of their own
//     cell from the 2-bit class map (tde_world.cell_cls2: 128-byte tiles of 8 m x 4 m, so the 16 pixels of a block share
//     one or two cache lines - the rasteriser is bound by the cache lines its look-ups touch, ~2 cycles of the CU's L1 per
//     line: profiles/r03_a_render_lines.txt); pixels in MIXED cells are compacted once more, take the class of their
//     sub-cell (4 x 4 per cell) and only what is left gets its candidate-triangle tests, densely;
//   * the cell under a point comes from its own affine map (pixel -> cell coordinates): it may differ from the cell of
//     the fp32 world point by ~1e-4 m, which the 5 cm classification margin of the grid absorbs;
//   * objects are culled to the view circle and painted over conservative pixel spans, in layer order.
#pragma once
#include <type_traits>

#include "tde_device.h"

namespace tde {

// ablation switches of tuning builds (scripts/build_variant.sh -DTDE_RASTER_SKIP=mask; results are then wrong on purpose):
// 1 no triangle tests, 2 no per-pixel look-ups of listed blocks, 4 no object painting, 8 no stream-out, 32 every cell-word
// look-up twice
#ifndef TDE_RASTER_SKIP
#define TDE_RASTER_SKIP 0
#endif

constexpr int kRasterMaxPix = 4096;              // padded H * W limit: the view is staged in LDS as one layer byte per pixel
constexpr int kRasterBlockQ = 256;               // listed 4x4 blocks (an image has at most 256)
constexpr int kRasterMixQ = 384;                 // pixels in MIXED cells awaiting their sub-cell class / triangle tests (a trip
                                                 // of the pixel stage appends <= 256: resolved early when that might not fit)

// LDS of one view: 5120 B = 160 KiB / 32, i.e. eight wavefronts (views) per SIMD.  (Object records never touch LDS: the
// paint passes broadcast them from the lane that formed them.)
struct RasterQueues {
    uint8_t blockq[kRasterBlockQ];               // index of a 4x4 block: (r0 / 4) * (padded W / 4) + c0 / 4
    uint16_t mixq[kRasterMixQ];                  // (r << 8) | c of a pixel
};
struct RasterScratch {
    uint32_t plane[kRasterMaxPix / 4];
    RasterQueues q;
};
static_assert(sizeof(RasterScratch) == 5120, "eight views per SIMD");

// what one view needs (wave-uniform)
struct RasterJob {
    // world
    const uint32_t *cell_word;
    const float *cell_tri;
    const uint32_t *cell_cls2;
    const uint32_t *cell_sub;
    const uint8_t *cell_coarse;
    const tde_stopline *stoplines;               // of the env's map (already offset by stop_base)
    const double *wp;                            // waypoints of the env's scenario
    tde_map m;
    uint32_t red;                                // lights of the map that are red now (0 without TDE_F_TRAFFIC_LIGHTS)
    bool lights;
    int n_wp, ti;                                // remaining waypoints: [ti, n_wp)
    int A;
    // the ego
    float ex, ey, ce, se;
    // image
    int H, W, ns, phase, flags;
    float res, inv_res, thr2;
    int K8, K4;                                  // clearance units that make an 8x8 / 4x4 block of pixel centres uniform (in
                                                 // TDE_COARSE_UNITs; TDE_CLEARANCE_UNITs in the TDE_RASTER_COARSE=0 tuning build)
    uint8_t *out;                                // this view's [3 * ns][H][W]
    uint8_t *ring;                               // this view's [ns][H * W] layer planes, or nullptr
    bool fresh;
};

// Round 4: the block pyramid reads the COARSE table (tde_world.cell_coarse: one byte per 4 x 4 cells, 16 x 8 tiles per 128-byte
// line) instead of the clearance field of the cell words: its 320 look-ups per view touched ~320 cache lines of a table of
// 66 MB per km^2, now ~20 lines of a table of 1 MB per km^2.  TDE_RASTER_COARSE=0 keeps the cell-word form (A/B builds).
#ifndef TDE_RASTER_COARSE
#define TDE_RASTER_COARSE 1
#endif

// clearance (in units of `unit` metres, rounded up) at which an n x n block of pixel centres is uniform: the centres lie
// within half a diagonal of the block's centre; 1 % + 2 cm cover the fp32 evaluation of both
inline int raster_block_clearance(int n, float res, float unit = TDE_RASTER_COARSE ? TDE_COARSE_UNIT : TDE_CLEARANCE_UNIT)
{
    const float r = 0.5f * (float)(n - 1) * 1.41421356f * res * 1.01f + 0.02f;
    int k = (int)(r / unit);
    while ((float)k * unit < r) ++k;
    return k;
}

// Phases of a view communicate through LDS inside ONE wavefront: its LDS operations execute in program order, so no
// barrier or wait is needed - only the compiler must keep them in that order across accesses of different types (the
// plane is written as bytes, 16-, 32- and 64-bit words and read back as 16-byte vectors).
TDE_DEV void wave_phase() { asm volatile("" ::: "memory"); }

// (lane_prefix: tde_device.h)

// per-view pixel maps
struct RasterView {
    float ex, ey, ax, ay, bx, by;                // (u, v) -> world (the specification)
    float hu, hv;                                // u = hu - r, v = hv - c
    float c0x, cax, cbx, c0y, cay, cby;          // (u, v) -> cell coordinates (look-up only)
    float nxm1, nym1;
};

TDE_DEV RasterView raster_view_maps(const RasterJob &J)
{
    RasterView V;
    const float lsign = (J.flags & TDE_RENDER_LEFT_HANDED) ? -1.0f : 1.0f;   // left-handed world: lateral image axis mirrored
    const float rs = J.res * lsign;
    V.ex = J.ex; V.ey = J.ey;
    V.ax = J.res * J.ce; V.ay = J.res * J.se; V.bx = (-rs) * J.se; V.by = rs * J.ce;
    V.hu = 0.5f * (float)J.H - 0.5f; V.hv = 0.5f * (float)J.W - 0.5f;
    V.c0x = (J.ex - J.m.ox) * J.m.inv_cell; V.cax = V.ax * J.m.inv_cell; V.cbx = V.bx * J.m.inv_cell;
    V.c0y = (J.ey - J.m.oy) * J.m.inv_cell; V.cay = V.ay * J.m.inv_cell; V.cby = V.by * J.m.inv_cell;
    V.nxm1 = (float)(J.m.nx - 1); V.nym1 = (float)(J.m.ny - 1);
    return V;
}

TDE_DEV void raster_world(const RasterView &V, float u, float v, float &wx, float &wy)
{
    wx = __builtin_fmaf(v, V.bx, __builtin_fmaf(u, V.ax, V.ex));
    wy = __builtin_fmaf(v, V.by, __builtin_fmaf(u, V.ay, V.ey));
}

// cell word under pixel (u, v): the grid is padded by >= 2 EMPTY cells, so clamping replaces the bounds tests (cell_lookup);
// (fx, fy) = the clamped cell coordinates (what subcell_class takes)
TDE_DEV uint32_t raster_lookup(const RasterJob &J, const RasterView &V, float u, float v, float &fx, float &fy)
{
    fx = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cbx, __builtin_fmaf(u, V.cax, V.c0x)), 0.0f, V.nxm1);
    fy = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cby, __builtin_fmaf(u, V.cay, V.c0y)), 0.0f, V.nym1);
    const uint32_t idx = (uint32_t)J.m.cell_base + (((uint32_t)(int)fy << J.m.row_shift) + (uint32_t)(int)fx);
#if TDE_RASTER_SKIP & 32          // tuning probe: every look-up twice (a second, distant line) - what a cache line costs
    { const uint32_t dup = J.cell_word[idx ^ 0x8000u]; asm volatile("" :: "v"(dup)); }
#endif
    return J.cell_word[idx];
}
TDE_DEV uint32_t raster_lookup(const RasterJob &J, const RasterView &V, float u, float v)
{
    float fx, fy;
    return raster_lookup(J, V, u, v, fx, fy);
}

// class | clearance << 2 of the coarse tile under pixel (u, v) (tde_abi.h: tde_world.cell_coarse); the pyramid's look-up
TDE_DEV uint32_t raster_coarse(const RasterJob &J, const RasterView &V, float u, float v)
{
#if TDE_RASTER_COARSE
    const float fx = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cbx, __builtin_fmaf(u, V.cax, V.c0x)), 0.0f, V.nxm1);
    const float fy = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cby, __builtin_fmaf(u, V.cay, V.c0y)), 0.0f, V.nym1);
    const uint32_t cx = (uint32_t)(int)fx >> 2, cy = (uint32_t)(int)fy >> 2;
    static_assert(TDE_COARSE_CELLS == 4, "coarse tiles of 4 x 4 cells");
    const uint32_t line = (uint32_t)J.m.coarse_base + ((cy >> 3) << (J.m.row_shift - 6)) + (cx >> 4);
    return J.cell_coarse[(line << 7) | (((cy & 7u) << 4) | (cx & 15u))];
#else
    return raster_lookup(J, V, u, v) & 1023u;
#endif
}

// class of the sub-cell of a MIXED cell that holds the clamped cell coordinates (fx, fy): two bits of the cell's word in
// the tiled sub-cell array (tde_abi.h: tde_world.cell_sub; world.py: subcell_classes)
TDE_DEV uint32_t subcell_class(const RasterJob &J, float fx, float fy)
{
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    const uint32_t tile = ((iy >> 2) << (J.m.row_shift - 3)) + (ix >> 3);
    const uint32_t bm = J.cell_sub[(uint32_t)J.m.cell_base + ((tile << 5) | (((iy & 3u) << 3) | (ix & 7u)))];
    const int sx = min((int)(__builtin_amdgcn_fractf(fx) * (float)TDE_CELL_SUB), TDE_CELL_SUB - 1);
    const int sy = min((int)(__builtin_amdgcn_fractf(fy) * (float)TDE_CELL_SUB), TDE_CELL_SUB - 1);
    return (bm >> (2 * (sy * TDE_CELL_SUB + sx))) & 3u;
}

// class of the cell under pixel (u, v) from the 2-bit class map (tde_abi.h: tde_world.cell_cls2)
TDE_DEV uint32_t raster_class(const RasterJob &J, const RasterView &V, float u, float v)
{
    const float fx = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cbx, __builtin_fmaf(u, V.cax, V.c0x)), 0.0f, V.nxm1);
    const float fy = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cby, __builtin_fmaf(u, V.cay, V.c0y)), 0.0f, V.nym1);
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    const uint32_t tile = (uint32_t)J.m.cls2_base + ((iy >> 4) << (J.m.row_shift - 5)) + (ix >> 5);
    const uint32_t word = J.cell_cls2[(tile << 5) | (((iy & 15u) << 1) | ((ix >> 4) & 1u))];
    return (word >> ((ix & 15u) << 1)) & 3u;
}

// road / not road of a pixel in a MIXED cell: its candidate triangles, two records in flight per trip (the loop is a
// chain of dependent L2 round trips).  Divergent-safe (no wave-level operations).
TDE_DEV bool raster_mixed_pixel(const RasterJob &J, const RasterView &V, float u, float v, uint32_t wd)
{
    if (TDE_RASTER_SKIP & 1) return false;
    float wx, wy;
    raster_world(V, u, v, wx, wy);
    const float4 *recs = reinterpret_cast<const float4 *>(J.cell_tri) + 3 * (size_t)((wd >> 10) + (uint32_t)J.m.rec_base);
    const int n = (int)((wd >> 2) & 255u);
    bool road = false;
    for (int k = 0; k < n && !road; k += 2) {
        const float4 *r0 = recs + 3 * k, *r1 = recs + 3 * (k + 1 < n ? k + 1 : k);
        const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        road = point_tri_d2_words(wx, wy, a0, a1, a2) <= J.thr2;
        if (!road && k + 1 < n) road = point_tri_d2_words(wx, wy, b0, b1, b2) <= J.thr2;
    }
    return road;
}

// coefficients of a box's frame coordinates p (along) and q (across) as functions of (u, v): the oracle's expressions
TDE_DEV void box_coeffs(const RasterView &V, float x, float y, float cb, float sb, float4 &P, float4 &Q)
{
    const float dx = V.ex - x, dy = V.ey - y;
    P.x = dx * cb + dy * sb; P.y = V.ax * cb + V.ay * sb; P.z = V.bx * cb + V.by * sb;
    Q.x = dy * cb - dx * sb; Q.y = V.ay * cb - V.ax * sb; Q.z = V.by * cb - V.bx * sb;
}

// conservative pixel span (inclusive, one pixel of slack for rounding) of a box / disc, clipped to the image, packed
TDE_DEV uint32_t raster_span(const RasterJob &J, const RasterView &V, float x, float y, float ef, float el)
{
    const float lsign = (J.flags & TDE_RENDER_LEFT_HANDED) ? -1.0f : 1.0f;
    const float dx = x - J.ex, dy = y - J.ey;
    const float f = dx * J.ce + dy * J.se, l = (dy * J.ce - dx * J.se) * lsign;
    const float rc = V.hu - f * J.inv_res, cc = V.hv - l * J.inv_res;
    const float pr = ef * J.inv_res + 1.0f, pc = el * J.inv_res + 1.0f;
    const int rmin = max((int)floorf(rc - pr), 0), rmax = min((int)ceilf(rc + pr), J.H - 1);
    const int cmin = max((int)floorf(cc - pc), 0), cmax = min((int)ceilf(cc + pc), J.W - 1);
    // an empty span (the object lies outside the image) is stored as rmin = 255 > rmax
    if (rmin > rmax || cmin > cmax) return 255u;
    return (uint32_t)rmin | ((uint32_t)rmax << 8) | ((uint32_t)cmin << 16) | ((uint32_t)cmax << 24);
}

TDE_DEV uint32_t box_span(const RasterJob &J, const RasterView &V, float x, float y, float cb, float sb, float hl, float hw)
{
    const float cr = cb * J.ce + sb * J.se, sr = sb * J.ce - cb * J.se;      // box heading relative to the ego's
    return raster_span(J, V, x, y, fabsf(cr) * hl + fabsf(sr) * hw, fabsf(sr) * hl + fabsf(cr) * hw);
}

// paint one box record over its span: the wavefront walks the span as an 8 x 8 tile of lanes
TDE_DEV void raster_paint_box(uint8_t *p8, const RasterView &V, int W, const float4 &P, const float4 &Q, const float4 &X,
                              int lane)
{
    const uint32_t span = __float_as_uint(X.x);
    const int rmin = (int)(span & 255u), rmax = (int)((span >> 8) & 255u), cmin = (int)((span >> 16) & 255u),
              cmax = (int)(span >> 24);
    const uint8_t lay = (uint8_t)__float_as_uint(X.y);
    const int tr = lane >> 3, tc = lane & 7;
    for (int r = rmin + tr; r <= rmax; r += 8) {
        const float u = V.hu - (float)r;
        const float pu = __builtin_fmaf(u, P.y, P.x), qu = __builtin_fmaf(u, Q.y, Q.x);
        for (int c = cmin + tc; c <= cmax; c += 8) {
            const float v = V.hv - (float)c;
            const float p = __builtin_fmaf(v, P.z, pu), q = __builtin_fmaf(v, Q.z, qu);
            if (fabsf(p) <= P.w && fabsf(q) <= Q.w) p8[r * W + c] = lay;
        }
    }
}

// layers -> colours.  v_perm_b32 is a byte look-up in an 8-entry table: entries 0-4 the palette, 5 = TDE_LAYER_BLANK
// (0, 0, 0), 6 / 7 the stop-line colours; 16 layer bytes -> 3 x 16 colour bytes, streaming 16-byte stores
TDE_DEV void raster_expand(const uint4 &v, uint8_t *frame, int plane, int i)
{
    const uint32_t BG[3] = {TDE_RGB_BACKGROUND}, ROAD[3] = {TDE_RGB_ROAD}, WP[3] = {TDE_RGB_WAYPOINT},
                   NPC[3] = {TDE_RGB_NPC}, EGO[3] = {TDE_RGB_EGO}, SRED[3] = {TDE_RGB_STOP_RED}, SGO[3] = {TDE_RGB_STOP_GO};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const uint32_t lo = BG[ch] | (ROAD[ch] << 8) | (WP[ch] << 16) | (NPC[ch] << 24);
        const uint32_t hi = EGO[ch] | (SRED[ch] << 16) | (SGO[ch] << 24);          // entry 5 = TDE_LAYER_BLANK = 0
        uint4 o;
        o.x = __builtin_amdgcn_perm(hi, lo, v.x); o.y = __builtin_amdgcn_perm(hi, lo, v.y);
        o.z = __builtin_amdgcn_perm(hi, lo, v.z); o.w = __builtin_amdgcn_perm(hi, lo, v.w);
        // streaming stores: the observation is consumed by the policy, not by this kernel
        u32x4_t t;
        t.x = o.x; t.y = o.y; t.z = o.z; t.w = o.w;
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t *>(frame + ch * plane) + i);
    }
}

// One view by one wavefront; every lane of the wavefront calls it, converged.  `agent.fetch(j)` -> Raw (the loads of slot j of
// the env, slot 0 = the ego), `agent.unpack(raw, x, y, c, s, hl, hw)` -> present: its pose and half extents.
// SIZE: 64 = the image is 64 x 64 (the reference's observation, every stride a constant); 0 = J.H x J.W.
template <int SIZE, typename AgentSrc>
TDE_DEV void raster_view(RasterScratch &S, const RasterJob &J, AgentSrc &&agent, int *dbg = nullptr)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int H = SIZE ? SIZE : J.H, W = SIZE ? SIZE : J.W, plane = H * W;
    // the block pyramid covers the image rounded up to multiples of 8 (tde_render_ego checks that it fits the plane); the
    // pixels of the padding are computed like any other and never leave LDS
    const int Wp = (W + 7) & ~7, Hp = (H + 7) & ~7;
    const RasterView V = raster_view_maps(J);
    uint8_t *p8 = reinterpret_cast<uint8_t *>(S.plane);
    const int ego_layer = (J.flags & TDE_RENDER_PLAIN_EGO) ? TDE_LAYER_NPC : TDE_LAYER_EGO;
    const float rview = 0.75f * J.res * (float)(H > W ? H : W) + 1.0f;        // view circle: the culled lists are supersets

    // the first 64 candidates of every kind of object (stop lines, waypoints, agent slots) are fetched together (one memory round
    // trip instead of three).  (Issued earlier - ahead of the last MIXED-pixel resolution - their ten registers spill: 21 VGPR
    // spills under the 64-VGPR cap of eight views per SIMD.)
    float4 la0 = make_float4(0.0f, 0.0f, 1.0f, 0.0f), lb0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    double2 wp0 = make_double2(0.0, 0.0);
    typename std::remove_reference<AgentSrc>::type::Raw raw0{};
    auto fetch_objects = [&]() {
        if (J.lights && lane < J.m.n_stop) {
            la0 = reinterpret_cast<const float4 *>(J.stoplines + lane)[0];
            lb0 = reinterpret_cast<const float4 *>(J.stoplines + lane)[1];
        }
        if (J.ti + lane < J.n_wp) wp0 = reinterpret_cast<const double2 *>(J.wp)[J.ti + lane];
        if (lane < J.A) raw0 = agent.fetch(lane);
    };
#ifndef TDE_RASTER_EARLY_OBJECTS
#define TDE_RASTER_EARLY_OBJECTS 0     // 1: the object records' loads ahead of the pyramid (A/B: profiles/r04_raster_floor.md)
#endif
    if (TDE_RASTER_EARLY_OBJECTS) fetch_objects();
    // ---- base layer, 8x8 -> 4x4 -> 2x2 blocks -> pixels; lane b owns 8x8 block b ---------------------------------------
    {
        const int nbw = Wp >> 3, nblk = (Hp >> 3) * nbw, nb4w = Wp >> 2;      // nblk <= 64
        const bool have = lane < nblk;
        const int br = SIZE ? lane >> 3 : (have ? lane / nbw : 0), bc = SIZE ? lane & 7 : (have ? lane - br * nbw : 0);
        const int r8 = br * 8, c8 = bc * 8;
        const float u8 = V.hu - (float)r8 - 3.5f, v8 = V.hv - (float)c8 - 3.5f;   // the block's centre
        uint32_t need1 = 0;                                                   // bit s: 4x4 sub-block s needs splitting
        {
            // the block's own look-up and those of its four 4x4 sub-blocks are issued TOGETHER (the sub-blocks' are wasted
            // when the 8x8 block turns out uniform - about half of them - but a dependent round trip is what a view pays for)
#ifndef TDE_RASTER_SPEC_L1
#define TDE_RASTER_SPEC_L1 1
#endif
            uint32_t wd = 0u, w4[4] = {0u, 0u, 0u, 0u};
            if (have) {
                wd = raster_coarse(J, V, u8, v8);
                if (TDE_RASTER_SPEC_L1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        w4[s] = raster_coarse(J, V, u8 + ((s >> 1) ? -2.0f : 2.0f), v8 + ((s & 1) ? -2.0f : 2.0f));
                }
            }
            const uint32_t cls = wd & 3u;
            const bool uni = cls != TDE_CELL_MIXED && (int)(wd >> 2) >= J.K8;
            if (!TDE_RASTER_SPEC_L1 && have && !uni) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    w4[s] = raster_coarse(J, V, u8 + ((s >> 1) ? -2.0f : 2.0f), v8 + ((s & 1) ? -2.0f : 2.0f));
            }
            if (have && uni) {
                const uint2 val = cls == TDE_CELL_FULL ? make_uint2(0x01010101u, 0x01010101u) : make_uint2(0u, 0u);
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<uint2 *>(p8 + (r8 + i) * Wp + c8) = val;
            }
            if (have && !uni) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t c4 = w4[s] & 3u;
                    if (c4 != TDE_CELL_MIXED && (int)(w4[s] >> 2) >= J.K4) {
                        const uint32_t val = c4 == TDE_CELL_FULL ? 0x01010101u : 0u;
                        const int r4 = r8 + 4 * (s >> 1), c4o = c8 + 4 * (s & 1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint32_t *>(p8 + (r4 + i) * Wp + c4o) = val;
                    } else {
                        need1 |= 1u << s;
                    }
                }
            }
        }
        // ---- list the 4x4 blocks that straddle a road edge, lane by lane (the blocks of one 8x8 parent stay adjacent: the 64
        // pixels of a trip then share two or three lines of the class map): wave prefix sum of popc(need1) (0..4) from the
        // ballots of its three bits --------------------------------------------------------------------------------------
        int nq;
        {
            const uint32_t n = (uint32_t)__popc(need1);
            const unsigned long long b0 = __ballot((n & 1u) != 0u), b1 = __ballot((n & 2u) != 0u), b2 = __ballot((n & 4u) != 0u);
            const int at = lane_prefix(b0) + 2 * lane_prefix(b1) + 4 * lane_prefix(b2);
            nq = (int)__popcll(b0) + 2 * (int)__popcll(b1) + 4 * (int)__popcll(b2);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if ((need1 >> s4) & 1u)
                    S.q.blockq[at + __popc(need1 & ((1u << s4) - 1u))] = (uint8_t)((2 * br + (s4 >> 1)) * nb4w + 2 * bc + (s4 & 1));
            }
        }
        wave_phase();
        // A view under load is bound by the LATENCY of its chain of dependent memory round trips (8192 views in flight: a
        // round trip costs 1 - 2 us; profiles/r03_b_render_views_per_wave.txt), so every stage below keeps as many
        // independent look-ups in flight per lane as registers allow and the stages are few.
        // ---- the pixels of the listed blocks, 16 consecutive lanes per block, kPx blocks per lane and trip: the class of
        // the pixel's own cell; pixels in MIXED cells are compacted once more (mixq) ------------------------------------
        constexpr int kPx = 4;
        int nmix = 0;
        const int nitems = (TDE_RASTER_SKIP & 2) ? 0 : 16 * nq;
        // mixq entries [0, n): sub-cell classes, kPx entries per lane in flight; what is still undecided is compacted to
        // the front of the queue and gets its cell word and candidate triangles, one pixel per lane
        auto resolve_mixed = [&](int n) {
            for (int base = 0; base < n; base += 64 * kPx) {
                uint32_t px[kPx], cm[kPx];
                bool act[kPx];
#pragma unroll
                for (int h = 0; h < kPx; ++h) {
                    const int i = base + 64 * h + lane;
                    act[h] = i < n;
                    px[h] = S.q.mixq[act[h] ? i : 0];
                    const float u = V.hu - (float)(px[h] >> 8), v = V.hv - (float)(px[h] & 255u);
                    const float fx = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cbx, __builtin_fmaf(u, V.cax, V.c0x)), 0.0f, V.nxm1);
                    const float fy = __builtin_amdgcn_fmed3f(__builtin_fmaf(v, V.cby, __builtin_fmaf(u, V.cay, V.c0y)), 0.0f, V.nym1);
                    cm[h] = subcell_class(J, fx, fy);
                }
                wave_phase();                                     // every lane has read its entries: the front may be reused
                int ns = 0;
#pragma unroll
                for (int h = 0; h < kPx; ++h) {
                    const bool und = act[h] && cm[h] == TDE_CELL_MIXED;
                    if (act[h] && !und) p8[(int)(px[h] >> 8) * Wp + (int)(px[h] & 255u)] = cm[h] == TDE_CELL_FULL ? 1 : 0;
                    const unsigned long long um = __ballot(und);
                    if (und) S.q.mixq[base + ns + lane_prefix(um)] = (uint16_t)px[h];      // (ns + prefix <= entries read so far)
                    ns += (int)__popcll(um);
                }
                wave_phase();
                // (fetching the cell word beside the sub-cell word and carrying it through LDS - one dependent round trip
                //  less for the undecided pixels, 130 more look-ups per view - is a wash: 39.1 vs 38.6 us)
                for (int b2 = 0; b2 < ns; b2 += 64) {
                    const int i = b2 + lane;
                    if (i < ns) {
                        const uint32_t q = S.q.mixq[base + i];
                        const int rr = (int)(q >> 8), cc = (int)(q & 255u);
                        const float u = V.hu - (float)rr, v = V.hv - (float)cc;
                        p8[rr * Wp + cc] = (uint8_t)(raster_mixed_pixel(J, V, u, v, raster_lookup(J, V, u, v)) ? 1 : 0);
                    }
                }
                wave_phase();
            }
        };
        for (int base = 0; base < nitems; base += 64 * kPx) {
            if (nmix > kRasterMixQ - 64 * kPx) { if (dbg) dbg[1] += nmix; resolve_mixed(nmix); nmix = 0; }     // (room for a whole trip)
            bool act[kPx];
            int r[kPx], c[kPx];
            uint32_t cls[kPx];
#pragma unroll
            for (int h = 0; h < kPx; ++h) {
                const int item = base + 64 * h + lane;
                act[h] = item < nitems;
                const int blk = S.q.blockq[act[h] ? (item >> 4) : 0];
                const int b4r = SIZE ? blk >> 4 : blk / nb4w, b4c = SIZE ? blk & 15 : blk - b4r * nb4w;
                r[h] = 4 * b4r + ((item >> 2) & 3); c[h] = 4 * b4c + (item & 3);
                cls[h] = raster_class(J, V, V.hu - (float)r[h], V.hv - (float)c[h]);
            }
#pragma unroll
            for (int h = 0; h < kPx; ++h) {
                const bool mx = act[h] && cls[h] == TDE_CELL_MIXED;
                if (act[h] && !mx) p8[r[h] * Wp + c[h]] = cls[h] == TDE_CELL_FULL ? 1 : 0;
                const unsigned long long mm = __ballot(mx);
                if (mx) S.q.mixq[nmix + lane_prefix(mm)] = (uint16_t)((r[h] << 8) | c[h]);
                nmix += (int)__popcll(mm);
            }
        }
        wave_phase();
        if (dbg) { dbg[0] = nq; dbg[1] += nmix; }
        resolve_mixed(nmix);
    }
    wave_phase();

    // ---- objects over the base, in layer order: stop lines (index order: a later line wins where two overlap), waypoint
    // discs, NPC boxes, the ego.  Each kind is culled to the view circle 64 candidates at a time, every lane forming the
    // record of its own candidate; the kept ones are then painted one after the other in lane (= index) order, their records
    // broadcast from the owning lane with v_readlane - no LDS round trip, no list, any number of objects per view -------------
    auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    auto paint_boxes = [&](bool keep, const float4 &P, const float4 &Q, uint32_t span, uint32_t lay) {
        unsigned long long km = (TDE_RASTER_SKIP & 4) ? 0ull : __ballot(keep);
        if (dbg) dbg[2] += (int)__popcll(km);
        while (km) {
            const int l = __ffsll((long long)km) - 1;
            km &= km - 1ull;
            const float4 Pb = make_float4(rl(P.x, l), rl(P.y, l), rl(P.z, l), rl(P.w, l));
            const float4 Qb = make_float4(rl(Q.x, l), rl(Q.y, l), rl(Q.z, l), rl(Q.w, l));
            const float4 Xb = make_float4(rl(__uint_as_float(span), l), rl(__uint_as_float(lay), l), 0.0f, 0.0f);
            raster_paint_box(p8, V, Wp, Pb, Qb, Xb, lane);
        }
    };
    if (!TDE_RASTER_EARLY_OBJECTS) fetch_objects();
    if (J.lights) {
        for (int q0 = 0; q0 < J.m.n_stop; q0 += 64) {
            const int q = q0 + lane;
            bool keep = false;
            float4 P = make_float4(0.0f, 0.0f, 0.0f, 0.0f), Q = P;
            uint32_t span = 255u, lay = 0u;
            if (q < J.m.n_stop) {
                float4 la = la0, lb = lb0;                                                        // hl, hw, light, -
                if (q0 > 0) {
                    la = reinterpret_cast<const float4 *>(J.stoplines + q)[0];
                    lb = reinterpret_cast<const float4 *>(J.stoplines + q)[1];
                }
                const float dx = la.x - J.ex, dy = la.y - J.ey, rr = rview + (lb.x + lb.y);
                keep = dx * dx + dy * dy <= rr * rr;
                box_coeffs(V, la.x, la.y, la.z, la.w, P, Q);
                P.w = lb.x; Q.w = lb.y;
                lay = ((J.red >> __float_as_int(lb.z)) & 1u) ? TDE_LAYER_STOP_RED : TDE_LAYER_STOP_GO;
                span = box_span(J, V, la.x, la.y, la.z, la.w, lb.x, lb.y);
            }
            paint_boxes(keep, P, Q, span, lay);
        }
    }
    for (int k0 = J.ti; k0 < J.n_wp; k0 += 64) {
        const int k = k0 + lane;
        bool keep = false;
        float dx0 = 0.0f, dy0 = 0.0f;
        uint32_t span = 255u;
        if (k < J.n_wp) {
            double2 t = wp0;
            if (k0 > J.ti) t = reinterpret_cast<const double2 *>(J.wp)[k];
            const float tx = (float)t.x, ty = (float)t.y;
            const float dx = tx - J.ex, dy = ty - J.ey, rr = rview + TDE_WAYPOINT_RADIUS;
            keep = dx * dx + dy * dy <= rr * rr;
            dx0 = V.ex - tx; dy0 = V.ey - ty;
            span = raster_span(J, V, tx, ty, TDE_WAYPOINT_RADIUS, TDE_WAYPOINT_RADIUS);
        }
        unsigned long long km = (TDE_RASTER_SKIP & 4) ? 0ull : __ballot(keep);
        while (km) {
            const int l = __ffsll((long long)km) - 1;
            km &= km - 1ull;
            const float wx0 = rl(dx0, l), wy0 = rl(dy0, l);
            const uint32_t sp = (uint32_t)__builtin_amdgcn_readlane((int)span, l);
            const int rmin = (int)(sp & 255u), rmax = (int)((sp >> 8) & 255u), cmin = (int)((sp >> 16) & 255u), cmax = (int)(sp >> 24);
            for (int r = rmin + (lane >> 3); r <= rmax; r += 8) {
                const float u = V.hu - (float)r;
                const float xu = __builtin_fmaf(u, V.ax, wx0), yu = __builtin_fmaf(u, V.ay, wy0);
                for (int c = cmin + (lane & 7); c <= cmax; c += 8) {
                    const float v = V.hv - (float)c;
                    const float dx = __builtin_fmaf(v, V.bx, xu), dy = __builtin_fmaf(v, V.by, yu);
                    if (__builtin_fmaf(dx, dx, dy * dy) <= TDE_WAYPOINT_RADIUS * TDE_WAYPOINT_RADIUS)
                        p8[r * Wp + c] = TDE_LAYER_WAYPOINT;
                }
            }
        }
    }
    {
        // NPC boxes (slots 1 .. A-1), 64 slots at a time, in one pass with the ego, which is painted last: its lane is moved to the
        // end of the paint order by handling bit 0 of the first chunk's mask after everything else (the NPCs share one colour,
        // so their order among themselves does not show)
        float4 Pe = make_float4(0.0f, 0.0f, 0.0f, 0.0f), Qe = Pe;
        uint32_t span_e = 255u;
        bool keep_e = false;
        for (int j0 = 0; j0 < J.A; j0 += 64) {
            float x = 0.0f, y = 0.0f, cb = 1.0f, sb = 0.0f, hl = 0.0f, hw = 0.0f;
            auto raw = raw0;
            if (j0 > 0 && j0 + lane < J.A) raw = agent.fetch(j0 + lane);         // (more than 64 slots per env: a further fetch)
            const bool pres = j0 + lane < J.A && agent.unpack(raw, x, y, cb, sb, hl, hw);
            const bool ego = j0 == 0 && lane == 0;
            bool keep = ego && pres;                                          // (an absent ego is not painted: the oracle skips it)
            if (pres && !ego) {
                const float dx = x - J.ex, dy = y - J.ey, rr = rview + (hl + hw);
                keep = dx * dx + dy * dy <= rr * rr;
            }
            float4 P, Q;
            box_coeffs(V, x, y, cb, sb, P, Q);
            P.w = hl; Q.w = hw;
            const uint32_t span = box_span(J, V, x, y, cb, sb, hl, hw);
            paint_boxes(keep && !ego, P, Q, span, (uint32_t)TDE_LAYER_NPC);
            if (j0 == 0) { Pe = P; Qe = Q; span_e = span; keep_e = keep && ego; }
        }
        paint_boxes(keep_e, Pe, Qe, span_e, (uint32_t)ego_layer);
    }

    // ---- layers -> colours, streamed out --------------------------------------------------------------------------
    wave_phase();
    // 16 layer bytes of flat pixel chunk i: one 16-byte read when the plane is not padded, else four 4-byte reads (W is
    // a multiple of 4, so a dword never straddles two image rows)
    auto chunk = [&](int i) -> uint4 {
        if (Wp == W) return reinterpret_cast<const uint4 *>(S.plane)[i];
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 16 * i + 4 * k, r = p / W, c = p - r * W;
            d[k] = *reinterpret_cast<const uint32_t *>(p8 + r * Wp + c);
        }
        return make_uint4(d[0], d[1], d[2], d[3]);
    };
    const int nv = (TDE_RASTER_SKIP & 8) ? 0 : plane / 16;
    const int ns = J.ns;
    if (ns > 1 && J.ring) {
        // frame stack from the ring of layer planes: slot of the new frame = phase % ns; output frame j (oldest first)
        // is ring slot (phase + 1 + j) % ns.  Nothing is shifted: every frame of `out` is written from its layer plane.
        const int slot_new = J.phase % ns;
        for (int i = lane; i < nv; i += 64) {
            const uint4 v = chunk(i);
            reinterpret_cast<uint4 *>(J.ring + (int64_t)slot_new * plane)[i] = v;
            raster_expand(v, J.out + 3 * (ns - 1) * plane, plane, i);
        }
        const uint32_t bl = TDE_LAYER_BLANK * 0x01010101u;
        for (int j = 0; j < ns - 1; ++j) {
            const int slot = (J.phase + 1 + j) % ns;
            uint4 *old = reinterpret_cast<uint4 *>(J.ring + (int64_t)slot * plane);
            for (int i = lane; i < nv; i += 64) {
                uint4 v;
                if (J.fresh) { v = make_uint4(bl, bl, bl, bl); old[i] = v; }      // VecFrameStack: the stack restarts blank
                else v = old[i];
                raster_expand(v, J.out + 3 * j * plane, plane, i);
            }
        }
    } else {
        if (SIZE == 64 && !(TDE_RASTER_SKIP & 8)) {          // the four 16-byte reads of a lane in flight together
            uint4 ch[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ch[k] = reinterpret_cast<const uint4 *>(S.plane)[lane + 64 * k];
#pragma unroll
            for (int k = 0; k < 4; ++k) raster_expand(ch[k], J.out + 3 * (ns - 1) * plane, plane, lane + 64 * k);
        } else {
            for (int i = lane; i < nv; i += 64) raster_expand(chunk(i), J.out + 3 * (ns - 1) * plane, plane, i);
        }
        if (J.fresh && ns > 1) {                     // in-place stack (no ring): blank the older frames of this view
            uint4 *o4 = reinterpret_cast<uint4 *>(J.out);
            for (int i = lane; i < 3 * (ns - 1) * nv; i += 64) o4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

}  // namespace tde
