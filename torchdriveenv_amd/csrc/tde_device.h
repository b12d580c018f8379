// tde_device.h — device-side building blocks of the env step path (gfx950 / CDNA4, wave64).
//
// Every fp32 expression here is written to round exactly like the CPU oracle's (oracle/tde_oracle.c): the library is
// compiled with -ffp-contract=off, without fast-math, with IEEE-correct fp32 divide/sqrt (hipcc default), so masks AND
// kinematic state are bit-identical to the oracle.  Reference anchors (file:line into inverted-ai/torchdriveenv) are
// given per function; torchdrivesim internals are restated from the published algorithm ("Oracle").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tde_abi.h"

#define TDE_DEV __device__ __forceinline__

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

namespace tde {

constexpr float kPi = 3.14159265358979323846f;      // float(np.pi)
constexpr float kTwoPi = 6.28318530717958647692f;   // float(2*np.pi)
constexpr float k2OverPi = 0.636619772367581343f;
constexpr float kPio2A = 1.5703125f;                // Cody-Waite split of pi/2 (k*A, k*B exact for |k| < 2^13)
constexpr float kPio2B = 4.837512969970703125e-4f;
constexpr float kPio2C = 7.54978995489188216e-8f;

// ---- 64-bit lane masks without 64-bit shifts by a VGPR amount ---------------------------------------------------------
// MI355X computes v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 wrong when the shift AMOUNT sits in the wavefront's last allocated
// VGPR (profiles/r05_a32_respawn_anomaly.md; scripts/ubench/shift64_last_vgpr.hip reproduces it) - and which register holds an
// amount is the allocator's choice.  So no kernel shifts a 64-bit value by a per-lane amount: a lane's bit of a wave mask is taken
// from the 32-bit HALF that holds it with one 32-bit shift, masks are assembled from two 32-bit words, and a lane's rank in a
// mask comes from v_mbcnt.  (Shifts by a wave-uniform amount are scalar instructions; 32-bit shifts are not affected.)
// torchdriveenv_amd/isa_audit.py fails the build if a 64-bit shift by a VGPR amount appears anywhere in the library.
TDE_DEV uint32_t mask_lo(unsigned long long m) { return (uint32_t)m; }
TDE_DEV uint32_t mask_hi(unsigned long long m) { return (uint32_t)(m >> 32); }
// "is bit 5 of i set" as a value the optimiser cannot see through: left transparent, `(i & 32) ? hi : lo` is recognised as
// trunc(m >> (i & 32)) and comes back as the very v_lshrrev_b64 by a VGPR amount this is here to avoid
TDE_DEV bool upper_half(int i)
{
    int up = i & 32;
    asm("" : "+v"(up));
    return up != 0;
}
// bit i (0 .. 63) of m
TDE_DEV uint32_t mask_bit(unsigned long long m, int i) { return ((upper_half(i) ? mask_hi(m) : mask_lo(m)) >> (i & 31)) & 1u; }
TDE_DEV uint32_t mask_bit(uint32_t m, int i) { return (m >> i) & 1u; }
// the bits of m from bit s (0 .. 63) up that lie in the same 32-bit half, at bit 0: what (uint32_t)(m >> s) gives when the field
// read does not straddle bit 32 (an env's lanes: a power of two <= 32, aligned)
TDE_DEV uint32_t mask_field(unsigned long long m, int s) { return (upper_half(s) ? mask_hi(m) : mask_lo(m)) >> (s & 31); }
// 1 << i (0 .. 63) as a 64-bit mask
TDE_DEV unsigned long long one_bit64(int i)
{
    const uint32_t b = 1u << (i & 31);
    const bool up = upper_half(i);
    return ((unsigned long long)(up ? b : 0u) << 32) | (up ? 0u : b);
}
// number of set bits of m below this lane's own bit (v_mbcnt_lo / _hi): popcount(m & ((1 << lane) - 1))
TDE_DEV int lane_prefix(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi(mask_hi(m), __builtin_amdgcn_mbcnt_lo(mask_lo(m), 0u));
}

// sin and cos of an fp32 angle.  ONE specification shared with the CPU checker: Cody-Waite reduction
// by pi/2 and degree-7/8 minimax polynomials, every multiply-add an explicit fused multiply-add (fmaf: one rounding, the
// same on CPU and GPU), so CPU and GPU agree bit for bit (<= 2 ulp vs libm).  Half the instructions of the unfused form.
TDE_DEV void sincos_f32(float xin, float &s, float &c)
{
    const float kf = __builtin_rintf(xin * k2OverPi);
    float r = __builtin_fmaf(-kf, kPio2A, xin);
    r = __builtin_fmaf(-kf, kPio2B, r);
    r = __builtin_fmaf(-kf, kPio2C, r);
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    const uint32_t q = (uint32_t)(int)kf;   // q mod 4: 0 (sn,cs) 1 (cs,-sn) 2 (-sn,-cs) 3 (-cs,sn)
    const bool odd = (q & 1u) != 0u;
    const float a = odd ? cs : sn;
    const float b = odd ? sn : cs;
    // negation = sign-bit flip: integer ops instead of two more compare/select pairs
    s = __uint_as_float(__float_as_uint(a) ^ ((q & 2u) << 30));
    c = __uint_as_float(__float_as_uint(b) ^ (((q + 1u) & 2u) << 30));
}

// sin of a small angle, same bits as sincos_f32: for |x| < 0.75 the reduction index rint(x*2/pi) is 0, the three
// Cody-Waite steps are exact no-ops (fma(-0, A, x) = x) and the quadrant is 0, so only the sine polynomial is left.
TDE_DEV float sin_small_f32(float x)
{
    if (fabsf(x) < 0.75f) {
        const float z = x * x;
        float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
        ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
        return __builtin_fmaf(x * z, ps, x);
    }
    float s, c;
    sincos_f32(x, s, c);
    return s;
}

// Natural logarithm of a normal fp32 u > 0, the specification shared with the CPU checker (its logf): same
// operations, every multiply-add an explicit fmaf, one IEEE division.
TDE_DEV float log_f32(float u)
{
    const uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;
    if (mb > 0x3fb504f3u) { mb -= 0x00800000u; e += 1; }
    const float m = __uint_as_float(mb);
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = __builtin_fmaf(z, 0.11111111f, 0.14285715f);
    p = __builtin_fmaf(p, z, 0.2f);
    p = __builtin_fmaf(p, z, 0.33333334f);
    p = __builtin_fmaf(p, z, 1.0f);
    const float lm = (s + s) * p;
    const float fe = (float)e;
    return __builtin_fmaf(fe, 6.9314575195e-1f, __builtin_fmaf(fe, 1.4286067653e-6f, lm));
}

// standard normal from two 32-bit random words (Box-Muller on the shared log / sincos specifications; tde_normal of the checker)
TDE_DEV float normal_f32(uint32_t ra, uint32_t rb)
{
    const float u1 = ((float)(ra >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = (float)(rb >> 8) * (1.0f / 16777216.0f);
    float sn, cs;
    sincos_f32(kTwoPi * u2, sn, cs);
    return __builtin_sqrtf(-2.0f * log_f32(u1)) * cs;
}

// the squared threshold d^2 is compared with (tde_config.offroad_threshold_squared selects the reading of upstream)
TDE_DEV float thr2_of(const tde_config &cfg)
{
    return cfg.offroad_threshold_squared ? cfg.offroad_threshold : cfg.offroad_threshold * cfg.offroad_threshold;
}

// fminf(fmaxf(v, lo), hi) for lo <= hi as ONE v_med3_f32 (the two-instruction form also canonicalises each operand first:
// four instructions per clamp with SGPR bounds).  Same value for every v including NaN (both return lo).
TDE_DEV float clampf(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

// torch.remainder(a, b) for b > 0: result in [0, b).  fmodf is exact, so the two fast paths (|a| < b: a itself;
// b <= |a| < 2b: a -+ b, exact by Sterbenz) return the very same bits as the generic call.
TDE_DEV float pymodf_pos(float a, float b)
{
    float aa = fabsf(a);
    float r;
    if (aa < b) r = a;
    else if (aa < 2.0f * b) r = a - copysignf(b, a);
    else r = fmodf(a, b);
    if (r < 0.0f) r += b;
    return r;
}

// Correctly rounded square root for x == 0 or x in [2^-96, FLT_MAX]: v_sqrt_f32 (1 ulp) and the usual one-ulp
// correction - of y - 1ulp, y, y + 1ulp the one whose neighbours' residuals x - n * y change sign - with the two
// decisions taken from the SIGN BITS of the (negated) residuals and added to the bit pattern: 8 instructions and no
// compare / select pair, against the 17 (+ four round trips through VCC) of the compiler's sqrtf, which also rescales
// arguments below 2^-96 and passes inf / NaN through.  The controller's two arguments cannot be in those ranges: a
// squared distance whose root only matters above 1e-3 m, and amax * max(gap - s0, 0), which is zero or a multiple of
// an ulp of a length of metres.  x == 0: y - 1ulp saturates at 0 and every residual is +0, so the result is 0.
TDE_DEV float sqrt_cr_f32(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const uint32_t yb = __float_as_uint(y);
    const uint32_t dn = __builtin_elementwise_sub_sat(yb, 1u), up = yb + 1u;
    const float t1 = __builtin_fmaf(__uint_as_float(dn), y, -x);      // -(x - dn * y): sign set <=> y - 1ulp is too small
    const float t2 = __builtin_fmaf(__uint_as_float(up), y, -x);      // -(x - up * y): sign set <=> y is too small
    return __uint_as_float(dn + (__float_as_uint(t1) >> 31) + (__float_as_uint(t2) >> 31));
}

// R4: KinematicBicycle.step — called through simulator.step(action), ref gym_env.py:117; model built at :245-247.
// `inv_lr` = 1.0f / rear_axis_offset, one correctly rounded division per agent and episode instead of one per step (the
// persistent kernels keep it in a register; the oracle forms the same product).
// TDE_KIN_EXPLICIT_EULER / TDE_KIN_LEFT_HANDED: the two readings of upstream that SURVEY R4 lists as undecidable here, as
// compile-time switches spelled like the CPU checker's (its bicycle restatement); defaults = the documented choice.
#ifndef TDE_KIN_EXPLICIT_EULER
#define TDE_KIN_EXPLICIT_EULER 0
#endif
#ifndef TDE_KIN_LEFT_HANDED
#define TDE_KIN_LEFT_HANDED 0
#endif
TDE_DEV void bicycle(float &x, float &y, float &psi, float &v, float inv_lr, float a, float beta, float dt)
{
    if (TDE_KIN_LEFT_HANDED) beta = -beta;
    float v1 = v + a * dt;
    const float vp = TDE_KIN_EXPLICIT_EULER ? v : v1;
    float sn, cs;
    sincos_f32(psi + beta, sn, cs);
    float x1 = x + (vp * cs) * dt;
    float y1 = y + (vp * sn) * dt;
    float sb = sin_small_f32(beta);                  // steering is bounded by 0.3 rad in the action space
    float p1 = psi + ((vp * inv_lr) * sb) * dt;
    p1 = pymodf_pos(kPi + p1, kTwoPi) - kPi;
    x = x1; y = y1; psi = p1; v = v1;
}

// R9: strict separating-axis overlap of two oriented boxes (compute_collision() > 0, ref gym_env.py:143,415).
TDE_DEV bool obb_overlap(float xi, float yi, float ci, float si, float hli, float hwi, float xj, float yj, float cj,
                         float sj, float hlj, float hwj)
{
    float dx = xj - xi, dy = yj - yi;
    float c = ci * cj + si * sj;
    float s = ci * sj - si * cj;
    float ac = fabsf(c), as = fabsf(s);
    float p = dx * ci + dy * si;
    bool ok = fabsf(p) < hli + (hlj * ac + hwj * as);
    float q = dy * ci - dx * si;
    ok = ok && (fabsf(q) < hwi + (hlj * as + hwj * ac));
    float p2 = dx * cj + dy * sj;
    ok = ok && (fabsf(p2) < hlj + (hli * ac + hwi * as));
    float q2 = dy * cj - dx * sj;
    ok = ok && (fabsf(q2) < hwj + (hli * as + hwi * ac));
    return ok;
}

// R10 building blocks: squared distance point -> segment / triangle (0 inside), from the packed triangle record
// (ax,ay,bx,by | cx,cy,1/|ab|^2,1/|bc|^2 | 1/|ca|^2,-,-,-): the reciprocals were computed on the host in fp32 exactly like the
// oracle's `1.0f / len2`, so the distances keep every bit while the kernel performs no division.
TDE_DEV float seg_d2_inv(float px, float py, float ax, float ay, float bx, float by, float inv)
{
    float abx = bx - ax, aby = by - ay;
    float apx = px - ax, apy = py - ay;
    float t = (apx * abx + apy * aby) * inv;
    t = clampf(t, 0.0f, 1.0f);
    float qx = apx - t * abx, qy = apy - t * aby;
    return qx * qx + qy * qy;
}

// (the record as three 16-byte words held in registers)
TDE_DEV float point_tri_d2_words(float px, float py, const float4 &t0, const float4 &t1, const float4 &t2)
{
    const float ax = t0.x, ay = t0.y, bx = t0.z, by = t0.w, cx = t1.x, cy = t1.y;
    float e0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    float e1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
    float e2 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);
    if ((e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) || (e0 <= 0.0f && e1 <= 0.0f && e2 <= 0.0f)) return 0.0f;
    float d = seg_d2_inv(px, py, ax, ay, bx, by, t1.z);
    d = fminf(d, seg_d2_inv(px, py, bx, by, cx, cy, t1.w));
    d = fminf(d, seg_d2_inv(px, py, cx, cy, ax, ay, t2.x));
    return d;
}

TDE_DEV float point_tri_d2_packed(float px, float py, const float4 *__restrict__ T)
{
    const float4 t0 = T[0], t1 = T[1];
    const float ax = t0.x, ay = t0.y, bx = t0.z, by = t0.w, cx = t1.x, cy = t1.y;
    float e0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    float e1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
    float e2 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);
    if ((e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) || (e0 <= 0.0f && e1 <= 0.0f && e2 <= 0.0f)) return 0.0f;
    const float inv_ca = T[2].x;
    float d = seg_d2_inv(px, py, ax, ay, bx, by, t1.z);
    d = fminf(d, seg_d2_inv(px, py, bx, by, cx, cy, t1.w));
    d = fminf(d, seg_d2_inv(px, py, cx, cy, ax, ay, inv_ca));
    return d;
}

// ---- offroad through the grid index ---------------------------------------------------------------------------------
// cell word: bits 0-1 class (EMPTY / MIXED / FULL), bits 2-9 number of candidate triangles, bits 10-31 first record in
// w.cell_tri counted from the map's rec_base (ABI 9).  The classification is conservative by GRID_MARGIN (world.py: build_grid_index), so the mask equals the
// oracle's brute force over every triangle.
TDE_DEV uint32_t cell_lookup(const tde_world &w, const tde_map &m, float px, float py)
{
    // The grid is padded by >= 2 EMPTY cells on every side, so clamping the cell coordinate to the grid (one v_med3_f32
    // per axis, ahead of the conversion) replaces the four bounds tests: anything outside lands in an EMPTY border cell
    // (+-inf clamp like any other value; a NaN coordinate comes out of v_med3_f32 as the minimum of the other two, 0).
    // Truncation after the clamp equals the clamp after truncation: (-1, 0) truncates to cell 0 either way.
    const float fx = __builtin_amdgcn_fmed3f((px - m.ox) * m.inv_cell, 0.0f, (float)(m.nx - 1));
    const float fy = __builtin_amdgcn_fmed3f((py - m.oy) * m.inv_cell, 0.0f, (float)(m.ny - 1));
    // rows are stored with a power-of-two pitch: the index is one shift-add
    return w.cell_word[(uint32_t)m.cell_base + (((uint32_t)(int)fy << m.row_shift) + (uint32_t)(int)fx)];
}

// two-level select on the bits of i (three v_cndmask; the comparison chain was compiled into nested exec-mask branches)
template <typename T> TDE_DEV T sel4(int i, T a0, T a1, T a2, T a3)
{
    const bool b0 = (i & 1) != 0, b1 = (i & 2) != 0;
    const T lo = b0 ? a1 : a0, hi = b0 ? a3 : a2;
    return b1 ? hi : lo;
}
#define TDE_SEL4(i, a0, a1, a2, a3) sel4((i), (a0), (a1), (a2), (a3))

// compute_offroad() > 0 for one box: any of the corners FL, FR, RR, RL farther than sqrt(thr2) from the mesh.
// Split in two so that the four dependent cell-word loads are in flight while other work (the collision sweep) runs:
//   offroad_issue   computes the corners and fetches their cell words,
//   offroad_resolve classifies them; corners in MIXED cells are resolved by a per-lane state machine that performs
//                   ONE triangle test per loop trip, so a wavefront iterates max-over-lanes(sum of tests) times
//                   instead of sum-over-corners(max-over-lanes).  Must be called by all lanes of the wavefront.
struct Corners {
    float px0, py0, px1, py1, px2, py2, px3, py3;
    uint32_t w0, w1, w2, w3;
};

// The class of the cell alone, from the 2-bit class map (tde_world.cell_cls2: 1/16 of the footprint of cell_word, so the maps of
// a world stay resident in every XCD's L2 and the four corners of a box fall into one 128-byte tile), with the cell's index in
// cell_word above it: class | index << 2.
TDE_DEV uint32_t cell_class_lookup(const tde_world &w, const tde_map &m, float px, float py)
{
    const float fx = __builtin_amdgcn_fmed3f((px - m.ox) * m.inv_cell, 0.0f, (float)(m.nx - 1));
    const float fy = __builtin_amdgcn_fmed3f((py - m.oy) * m.inv_cell, 0.0f, (float)(m.ny - 1));
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    const uint32_t tile = (uint32_t)m.cls2_base + ((iy >> 4) << (m.row_shift - 5)) + (ix >> 5);
    const uint32_t word = w.cell_cls2[(tile << 5) | (((iy & 15u) << 1) | ((ix >> 4) & 1u))];
    return ((word >> ((ix & 15u) << 1)) & 3u) | (((uint32_t)m.cell_base + ((iy << m.row_shift) + ix)) << 2);
}

#ifndef TDE_STEP_CLS2
#define TDE_STEP_CLS2 1
#endif
// the same choice for the judges of the two- and three-role rollout kernels (A/B on the town map, profiles/r04_*)
#ifndef TDE_ROLLOUT_CLS2
#define TDE_ROLLOUT_CLS2 0
#endif

// CLS2 (the one-step kernels, whose state is not register-resident and whose every launch therefore pays the HBM / fabric
// traffic of its lookups): the corners' classes come from the class map and only a corner in a MIXED cell fetches its cell word
// (offroad_resolve).  k.w* = class | cell index << 2 then.
template <bool CLS2 = false>
TDE_DEV void offroad_issue(const tde_world &w, const tde_map &m, bool live, float x, float y, float c, float s, float hl,
                           float hw, Corners &k)
{
    const float lx = hl * c, ly = hl * s, wx = hw * s, wy = hw * c;
    k.px0 = (x + lx) - wx; k.py0 = (y + ly) + wy;
    k.px1 = (x + lx) + wx; k.py1 = (y + ly) - wy;
    k.px2 = (x - lx) + wx; k.py2 = (y - ly) - wy;
    k.px3 = (x - lx) - wx; k.py3 = (y - ly) + wy;
    k.w0 = k.w1 = k.w2 = k.w3 = TDE_CELL_FULL;
    if (live) {
        if constexpr (CLS2) {
            k.w0 = cell_class_lookup(w, m, k.px0, k.py0);
            k.w1 = cell_class_lookup(w, m, k.px1, k.py1);
            k.w2 = cell_class_lookup(w, m, k.px2, k.py2);
            k.w3 = cell_class_lookup(w, m, k.px3, k.py3);
        } else {
            k.w0 = cell_lookup(w, m, k.px0, k.py0);
            k.w1 = cell_lookup(w, m, k.px1, k.py1);
            k.w2 = cell_lookup(w, m, k.px2, k.py2);
            k.w3 = cell_lookup(w, m, k.px3, k.py3);
        }
    }
}

// PAIR: two candidate records are fetched per trip of the MIXED-corner loop and the second one is tested only when the first
// did not settle the corner - half the dependent memory round trips for six more registers.  For the one-step kernels,
// whose launch ends with its slowest wavefront (a corner in a MIXED cell somewhere in the batch, every step); the
// persistent kernels keep one record per trip (their 80-VGPR budget, and their wavefronts drift apart anyway).
// rec_base = the map's first record in w.cell_tri (tde_map.rec_base): a cell word's record offset counts from there.
template <bool PAIR = false, bool CLS2 = false>
TDE_DEV bool offroad_resolve(const tde_world &w, const Corners &k, float thr2, int rec_base)
{
    const uint32_t w0 = k.w0, w1 = k.w1, w2 = k.w2, w3 = k.w3;
    // classes are 0 (EMPTY), 1 (MIXED), 2 (FULL): any EMPTY <=> the minimum class is 0; MIXED <=> bit 0.  Plain integer
    // arithmetic: the four compare / select pairs this replaces each went through VCC (tde_kernels.hip, sweep notes).
    static_assert(TDE_CELL_EMPTY == 0u && TDE_CELL_MIXED == 1u && TDE_CELL_FULL == 2u, "cell classes");
    bool off = min(min(w0 & 3u, w1 & 3u), min(w2 & 3u, w3 & 3u)) == 0u;
    uint32_t pending = off ? 0u : ((w0 & 1u) | ((w1 & 1u) << 1) | ((w2 & 1u) << 2) | ((w3 & 1u) << 3));
    bool work = false;
    uint32_t cur = 0, end = 0;
    float qx = 0.0f, qy = 0.0f;
    const float4 *recs = reinterpret_cast<const float4 *>(w.cell_tri);
    for (;;) {
        if (!work && pending) {
            const int ci = __ffs((int)pending) - 1;
            pending &= pending - 1u;
            uint32_t wd = TDE_SEL4(ci, w0, w1, w2, w3);
            if constexpr (CLS2) wd = w.cell_word[wd >> 2];    // (one more dependent load, for the corners in MIXED cells only)
            cur = (wd >> 10) + (uint32_t)rec_base;
            end = cur + ((wd >> 2) & 255u);
            qx = TDE_SEL4(ci, k.px0, k.px1, k.px2, k.px3);
            qy = TDE_SEL4(ci, k.py0, k.py1, k.py2, k.py3);
            work = true;
        }
        if (!__ballot(work)) break;
        if (work) {
            if constexpr (PAIR) {
                const float4 *r0 = recs + 3 * (size_t)cur, *r1 = recs + 3 * (size_t)(cur + 1 < end ? cur + 1 : cur);
                const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
                bool on = point_tri_d2_words(qx, qy, a0, a1, a2) <= thr2;
                if (!on && cur + 1 < end) on = point_tri_d2_words(qx, qy, b0, b1, b2) <= thr2;
                cur += 2;
                if (on) work = false;                         // this corner is on the road
                else if (cur >= end) { off = true; work = false; pending = 0u; }
            } else {
                const float d2 = point_tri_d2_packed(qx, qy, recs + 3 * (size_t)cur);
                if (d2 <= thr2) work = false;                     // this corner is on the road
                else if (++cur == end) { off = true; work = false; pending = 0u; }
            }
        }
    }
    return off;
}

template <bool PAIR = false, bool CLS2 = false>
TDE_DEV bool box_offroad(const tde_world &w, const tde_map &m, bool live, float x, float y, float c, float s, float hl,
                         float hw, float thr2)
{
    Corners k;
    offroad_issue<CLS2>(w, m, live, x, y, c, s, hl, hw, k);
    return offroad_resolve<PAIR, CLS2>(w, k, thr2, m.rec_base);
}

// Philox4x32-10, key = seed, counter = (c0,c1,c2,c3) — the reset RNG (R16).  Returned by value (uint4) so the four
// words live in registers: an output array would be placed in scratch memory.
TDE_DEV uint4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

TDE_DEV double u01(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

// R6/R7/R8/R11/R12 — the reference-owned reward/termination logic, ref gym_env.py:391-437 (see the oracle's
// tde_reward_core for the line-by-line citations).  float64 math on fp32 state, as the reference's Python does.
// sqrt(s) > r and sqrt(s) >= r for s >= 0, r >= 0 in float64, deciding on s against r*r whenever s is farther than
// 1e-12 (relative) from the boundary, i.e. farther than any rounding of the correctly rounded square root could
// matter; the exact sqrt is only evaluated inside that sliver.  Same truth value as the reference's
// `math.dist(..) > cutoff` / `math.dist(..) < 3` (gym_env.py:394,402) at a fraction of the instructions.
// `lo` / `hi` = r*r*(1 -+ 1e-12): the sliver, formed once (Cold) instead of with three float64 multiplies per call
struct Sliver { double lo, hi; };
__host__ TDE_DEV Sliver sliver_of(double r) { const double r2 = r * r; return Sliver{r2 * (1.0 - 1e-12), r2 * (1.0 + 1e-12)}; }
TDE_DEV bool sqrt_gt(double s, double r, const Sliver &v)
{
    if (s > v.hi) return true;
    if (s < v.lo) return false;
    return sqrt(s) > r;
}
TDE_DEV bool sqrt_ge(double s, double r, const Sliver &v)
{
    if (s > v.hi) return true;
    if (s < v.lo) return false;
    return sqrt(s) >= r;
}

// float64 cosine of a heading change for the reward's psi term (gym_env.py:403: math.cos on the float64 of an fp32
// difference, |x| < 2 pi + rounding).  OCML's cos carries the full-range argument reduction (v_trig_preop_f64, ~170
// float64 instructions in the kernel image, a few dozen on the common path) on a wavefront where only the ego lanes
// compute it; this is the fdlibm kernel pair restricted to the range that can occur: |x| < pi/4 - every step without a
// wrap of psi - evaluates one degree-12 polynomial, anything else goes through a two-term Cody-Waite reduction
// (pi/2 = pio2_1 + pio2_1t, the product k * pio2_1 is exact for |k| <= 2^20).  Faithfully rounded like OCML's and
// glibc's: against glibc it differs in the last bit on 1.2 % of random arguments (2e7 samples, max 1.1e-16) and never
// after the reward's rounding to fp32 (tests/test_gpu_parity.py::test_reward_cos_bits_agree_between_libm_and_ocml).
TDE_DEV double cos_kernel_f64(double x, double y)
{
    const double z = x * x;
    const double r = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                         -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                         2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
TDE_DEV double sin_kernel_f64(double x, double y)
{
    const double z = x * x, v = z * x;
    const double r = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10,
                         -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                         8.33333333332248946124e-03);
    return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01);
}
TDE_DEV double cos_heading_f64(double x)
{
    if (__builtin_fabs(x) < 0.78539816339744830962) return cos_kernel_f64(x, 0.0);
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    const double r = x - fn * 1.57079632673412561417e+00, w = fn * 6.07710050650619224932e-11;
    const double y0 = r - w, y1 = (r - y0) - w;
    const double v = (n & 1) ? sin_kernel_f64(y0, y1) : cos_kernel_f64(y0, y1);
    return (((n + 1) & 2) != 0) ? -v : v;                  // n & 3: 0 cos, 1 -sin, 2 -cos, 3 sin
}

// fp32 bounds of reward_core's pre-test: r^2 * (1 -+ 1e-6); a cut-off whose square leaves the normal fp32 range (or a NaN)
// gets (0, +inf), which sends every step to the float64 path
__host__ TDE_DEV void cut2f_bounds(double r, float &lo, float &hi)
{
    const double r2 = r * r;
    lo = 0.0f; hi = __builtin_inff();
    if (r2 > 1e-30 && r2 < 1e30) { lo = (float)(r2 * (1.0 - 1e-6)); hi = (float)(r2 * (1.0 + 1e-6)); }
}

// Kernel arguments that only the rare paths (reset, waypoint switches) or the few ego lanes read.  They are parked in
// LDS at kernel start: by-value argument structs of this path need ~110 SGPRs, more than the 102-SGPR file, and what
// does not fit is spilled to VGPR lanes and re-read with v_readlane on every use inside the step loop.
struct Cold {
    double waypoint_bonus, heading_penalty, distance_bonus, distance_cutoff, reach_radius;
    Sliver cut_sliver, reach_sliver;     // of distance_cutoff / reach_radius (sqrt_gt / sqrt_ge)
    float cut2f_lo, cut2f_hi;            // distance_cutoff^2 * (1 -+ 1e-6) in fp32 (reward_core's fp32 pre-test)
    uint64_t seed;
    const tde_spawn *spawn;
    const tde_scenario *scn;
    const double *wp_xy;
    const tde_map *maps;
    const float *route_xy;
    const float *start_psi;              // the heading table along the first waypoint segments (tde_world.start_psi), NH per scenario
    uint32_t env_base;
    int n_scn, NW, RW, max_steps, terminated_at_infraction, NH;
};

__host__ TDE_DEV void fill_cold(Cold &c, const tde_config &cfg, const tde_world &w)
{
    c.waypoint_bonus = cfg.waypoint_bonus; c.heading_penalty = cfg.heading_penalty;
    c.distance_bonus = cfg.distance_bonus; c.distance_cutoff = cfg.distance_cutoff;
    c.reach_radius = cfg.reach_radius; c.seed = cfg.seed;
    c.cut_sliver = sliver_of(cfg.distance_cutoff); c.reach_sliver = sliver_of(cfg.reach_radius);
    cut2f_bounds(cfg.distance_cutoff, c.cut2f_lo, c.cut2f_hi);
    c.spawn = w.spawn; c.scn = w.scn; c.wp_xy = w.wp_xy; c.maps = w.maps; c.route_xy = w.route_xy;
    c.start_psi = w.start_psi; c.NH = w.NH;
    c.env_base = cfg.env_base; c.n_scn = w.n_scn; c.NW = w.NW; c.RW = w.RW;
    c.max_steps = cfg.max_steps; c.terminated_at_infraction = cfg.terminated_at_infraction;
}

// the thresholds of reward_core from either carrier: the LDS block holds them precomputed, the operator kernel that takes
// tde_config forms them per call
struct RewardBounds { Sliver cut, reach; float cut2f_lo, cut2f_hi; };
TDE_DEV RewardBounds reward_bounds(const Cold &c) { return RewardBounds{c.cut_sliver, c.reach_sliver, c.cut2f_lo, c.cut2f_hi}; }
TDE_DEV RewardBounds reward_bounds(const tde_config &c)
{
    RewardBounds b{sliver_of(c.distance_cutoff), sliver_of(c.reach_radius), 0.0f, 0.0f};
    cut2f_bounds(c.distance_cutoff, b.cut2f_lo, b.cut2f_hi);
    return b;
}

struct RewardOut {
    float reward;
    uint8_t terminated, truncated;
    double psi_smooth, speed_smooth, psi_r, dist_r;
};

// The pieces of get_reward (:396-411), shared by reward_core (one step of one env) and by the three-role rollout
// kernel's batched evaluation (judge C: a window of steps of an env at once, one lane per step).
// dist_r (:402) and psi_r (:403): independent of the waypoint target.
// dist_r (:402)
template <typename CFG>
TDE_DEV double reward_dist_term(const CFG &cfg, const RewardBounds &rb, float lx, float ly, float x, float y)
{
    // math.dist(..) > cutoff (:402) decided in fp32 whenever that is safe: each fp32 difference is correctly rounded and
    // the sum of the two squares is within 4 * 2^-24 of the float64 value the reference forms from the same fp32 state, so
    // outside +-1e-6 (relative) of cutoff^2 the fp32 comparison cannot disagree with it; inside, the float64 path decides
    const float fdx = x - lx, fdy = y - ly;
    const float s32 = fdx * fdx + fdy * fdy;
    bool moved = s32 > rb.cut2f_hi;
    if (!moved && !(s32 < rb.cut2f_lo)) {
        const double ddx = (double)x - (double)lx, ddy = (double)y - (double)ly;
        moved = sqrt_gt(ddx * ddx + ddy * ddy, cfg.distance_cutoff, rb.cut);
    }
    return moved ? cfg.distance_bonus : 0.0;
}
// psi_r (:403): the subtraction in fp32 tensor arithmetic, the cosine in float64
template <typename CFG> TDE_DEV double reward_psi_term(const CFG &cfg, float lpsi, float psi)
{
    const float dpsi = psi - lpsi;
    return (1.0 - cos_heading_f64((double)dpsi)) * (-cfg.heading_penalty);
}
template <typename CFG>
TDE_DEV void reward_motion_terms(const CFG &cfg, const RewardBounds &rb, float lx, float ly, float lpsi, float x, float y,
                                 float psi, double &dist_r, double &psi_r)
{
    dist_r = reward_dist_term(cfg, rb, lx, ly, x, y);
    psi_r = reward_psi_term(cfg, lpsi, psi);
}
// check_reach_target (:391-394) for a target that exists (target_idx < n_wp)
template <typename CFG>
TDE_DEV bool reward_reach(const CFG &cfg, const RewardBounds &rb, float x, float y, double wtx, double wty)
{
    const double tx = (double)x - wtx, ty = (double)y - wty;
    return !sqrt_ge(tx * tx + ty * ty, cfg.reach_radius, rb.reach);
}
// the sum of :409-411, rounded to fp32 once
template <typename CFG> TDE_DEV float reward_sum(const CFG &cfg, bool reach, double dist_r, double psi_r)
{
    return (float)(((reach ? cfg.waypoint_bonus : 0.0) + dist_r) + psi_r);
}

// (wtx, wty) = waypoint[target_idx], only read when target_idx < n_wp (current_target is not None, :394).
template <typename CFG>
TDE_DEV RewardOut reward_core(const CFG &cfg, int n_wp, double wtx, double wty, float lx, float ly, float lpsi,
                              float lv, float x, float y, float psi, float v, bool off, bool col, bool tl, int k,
                              int &target_idx, int &reached, bool want_info = true)
{
    RewardOut o;
    const RewardBounds rb = reward_bounds(cfg);
    reward_motion_terms(cfg, rb, lx, ly, lpsi, x, y, psi, o.dist_r, o.psi_r);
    const int ti = target_idx;
    const bool reach = ti < n_wp && reward_reach(cfg, rb, x, y, wtx, wty);
    if (reach) reached += 1;
    o.reward = reward_sum(cfg, reach, o.dist_r, o.psi_r);
    o.terminated = (uint8_t)(cfg.terminated_at_infraction && (off || col || tl));
    o.truncated = (uint8_t)(k >= cfg.max_steps);
    o.psi_smooth = o.speed_smooth = 0.0;
    if (want_info) {
        o.psi_smooth = (double)fabsf((lpsi - psi) / 0.1f);
        o.speed_smooth = (double)fabsf((lv - v) / 0.1f);
    }
    if (reach) target_idx = ti + 1;
    return o;
}

}  // namespace tde
