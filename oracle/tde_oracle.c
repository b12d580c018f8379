/*
 * tde_oracle.c — CPU ORACLE for the batched driving-env step path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (torchdriveenv_amd + libtde_hip.so) never links, imports or falls back to it.
 *
 * It restates, one fp32/fp64 operation at a time, the per-timestep algorithm of inverted-ai/torchdriveenv:
 *   - the parts the reference OWNS (torchdriveenv/gym_env.py) are pinned by golden vectors captured from the
 *     reference source itself (oracle/gen_golden.py -> tests/golden/reward_golden.json):
 *       step order / last-state snapshot        gym_env.py:369-389, 115-120
 *       WaypointSuite reward                    gym_env.py:396-411
 *       waypoint reach / advance                gym_env.py:391-394, 378-383
 *       termination / truncation                gym_env.py:413-417, 134-135
 *       info terms                              gym_env.py:419-437
 *       reset / start sampling                  gym_env.py:319-367, 192-198 (RNG stream is ours, see below)
 *   - the parts the reference DELEGATES to the third-party package `torchdrivesim` (pyproject.toml:30 pins
 *     >=0.2.1, requirements.txt:74 locks commit 6c7957c780404980d9f69a00b40cb98eab0a87d5; source absent from
 *     /root/reference and from this image) are restated from the published algorithm and anchored on the
 *     reference's call sites.  PARITY UNPINNED for: bicycle integration (gym_env.py:117,245-247), collision
 *     (gym_env.py:143 with CollisionMetric.nograd :48), offroad (gym_env.py:142), replay time indexing
 *     (gym_env.py:275-294).  The heuristic NPC controller has no reference counterpart (it replaces the
 *     remote invertedai.api.drive call behind IAIWrapper, gym_env.py:285-294); it is defined here.
 *
 * Floating-point contract (what makes HIP-vs-oracle comparisons bit-exact):
 *   - every fp32 expression below is evaluated exactly as written, left to right, one IEEE-754 rounding per
 *     operation: build with -ffp-contract=off and without -ffast-math (oracle/Makefile does);
 *   - sin/cos of fp32 angles use tde_oracle_sincosf (Cody-Waite reduction + minimax polynomials evaluated with
 *     explicit fmaf, one rounding each) rather than libm, so that the HIP kernel can reproduce every bit; its error against
 *     libm sinf/cosf is <= 2 ulp on [-8, 8] (tests/test_oracle_math.py), far inside the 1e-5 state tolerance
 *     the north star allows against torch.sin/torch.cos;
 *   - the reward is float64 arithmetic on fp32 state, as in the reference (math.dist / math.cos on Python
 *     floats, gym_env.py:401-403), rounded to fp32 by `r += ...` on a zeros_like(x) tensor (:409-410).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/tde_abi.h"

#define TDE_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------ */
/* fp32 helpers                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

static const float TDE_PI_F = 3.14159265358979323846f;          /* float(np.pi)   */
static const float TDE_TWO_PI_F = 6.28318530717958647692f;      /* float(2*np.pi) */

/* Cody-Waite split of pi/2: the first two parts carry few significant bits so that k*part is exact for
 * |k| < 2^13. */
static const float TDE_2_OVER_PI = 0.636619772367581343f;
static const float TDE_PIO2_A = 1.5703125f;
static const float TDE_PIO2_B = 4.837512969970703125e-4f;
static const float TDE_PIO2_C = 7.54978995489188216e-8f;

/* sin and cos of an fp32 angle: Cody-Waite reduction by pi/2, then sin(r) and cos(r) for |r| <= pi/4 (+ reduction
 * slop) from minimax polynomials.  Every multiply-add is an explicit fmaf (ONE rounding; exact on any IEEE platform,
 * hardware FMA or libm's software one), which is what lets the HIP kernels reproduce every bit with v_fma_f32. */
TDE_EXPORT void tde_oracle_sincosf(float xin, float *s_out, float *c_out)
{
    float kf = rintf(xin * TDE_2_OVER_PI);
    float r = fmaf(-kf, TDE_PIO2_A, xin);
    r = fmaf(-kf, TDE_PIO2_B, r);
    r = fmaf(-kf, TDE_PIO2_C, r);
    float z = r * r;
    /* sin: r + r*z*(S1 + z*(S2 + z*S3)) */
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float sn = fmaf(r * z, ps, r);
    /* cos: 1 - z/2 + z*z*(C1 + z*(C2 + z*C3)) */
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    int q = ((int)kf) & 3;
    float s, c;
    if (q == 0)      { s = sn;  c = cs;  }
    else if (q == 1) { s = cs;  c = -sn; }
    else if (q == 2) { s = -sn; c = -cs; }
    else             { s = -cs; c = sn;  }
    *s_out = s;
    *c_out = c;
}

TDE_EXPORT void tde_oracle_sincosf_array(int64_t n, const float *x, float *s, float *c)
{
    for (int64_t i = 0; i < n; ++i) tde_oracle_sincosf(x[i], &s[i], &c[i]);
}

/* torch.remainder(a, b) for floats: result takes the sign of b (ATen BinaryOpsKernel remainder). */
static inline float tde_pymodf(float a, float b)
{
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

static inline float tde_clampf(float v, float lo, float hi)
{
    return fminf(fmaxf(v, lo), hi);
}

/* ------------------------------------------------------------------------------------------------ */
/* R4: KinematicBicycle.step (torchdrivesim/kinematic.py, called through simulator.step, gym_env.py:117) */
/*     a, beta = action; v += a*dt; x += v*cos(psi+beta)*dt; y += v*sin(psi+beta)*dt;               */
/*     psi += v*(1/lr)*sin(beta)*dt; psi = (pi + psi) % (2*pi) - pi.   left_handed=False (gym_env.py:245). */
/* ------------------------------------------------------------------------------------------------ */
/* The two readings of upstream this restatement had to choose between (torchdrivesim's source is absent: SURVEY R4) are
 * compile-time switches, spelled the same in csrc/tde_device.h (build both sides with the same -D and the parity suite holds):
 *   TDE_KIN_EXPLICIT_EULER  0 (default): the position update uses the NEW speed v' = v + a*dt (semi-implicit Euler);
 *                           1: it uses the old speed v (explicit Euler).
 *   TDE_KIN_LEFT_HANDED     0 (default; KinematicBicycle() is built with its default arguments, gym_env.py:245): steering as
 *                           given; 1: the steering angle is negated (a left-handed world frame).
 * Not switchable because it needs a quantity the env does not pass: steering as a front-wheel angle delta with
 * beta = atan(lr / (lf + lr) * tan(delta)) - the env hands over rear_axis_offset only (gym_env.py:246). */
#ifndef TDE_KIN_EXPLICIT_EULER
#define TDE_KIN_EXPLICIT_EULER 0
#endif
#ifndef TDE_KIN_LEFT_HANDED
#define TDE_KIN_LEFT_HANDED 0
#endif
TDE_EXPORT void tde_oracle_bicycle(float *x, float *y, float *psi, float *v, float lr, float a, float beta, float dt)
{
    if (TDE_KIN_LEFT_HANDED) beta = -beta;
    float v1 = *v + a * dt;
    float vp = TDE_KIN_EXPLICIT_EULER ? *v : v1;
    float sn, cs;
    tde_oracle_sincosf(*psi + beta, &sn, &cs);
    float x1 = *x + (vp * cs) * dt;
    float y1 = *y + (vp * sn) * dt;
    float sb, cb;
    tde_oracle_sincosf(beta, &sb, &cb);
    (void)cb;
    float inv_lr = 1.0f / lr;                      /* one rounding of the reciprocal, then products: the kernels keep
                                                    * inv_lr per agent instead of dividing every step */
    float p1 = *psi + ((vp * inv_lr) * sb) * dt;
    p1 = tde_pymodf(TDE_PI_F + p1, TDE_TWO_PI_F) - TDE_PI_F;
    *x = x1; *y = y1; *psi = p1; *v = v1;
}

/* Batched operator form: SimulatorInterface.step(action (B,A,2)) restricted to kinematics. */
TDE_EXPORT void tde_oracle_kinematics_step(int64_t n, float *x, float *y, float *psi, float *v, const float *lr,
                                           const uint8_t *present, const float *action, float dt)
{
    for (int64_t i = 0; i < n; ++i) {
        if (present && !present[i]) continue;
        tde_oracle_bicycle(&x[i], &y[i], &psi[i], &v[i], lr[i], action[2 * i], action[2 * i + 1], dt);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* R9: compute_collision() > 0 — strict separating-axis overlap of two oriented boxes centred at     */
/*     (x,y) (the env passes `center`, gym_env.py:219) with extents (length,width) (gym_env.py:261).  */
/*     Touching boxes do not collide (IoU > 0 needs positive area).                                  */
/* ------------------------------------------------------------------------------------------------ */
TDE_EXPORT int tde_oracle_obb_overlap(float xi, float yi, float ci, float si, float hli, float hwi,
                                      float xj, float yj, float cj, float sj, float hlj, float hwj)
{
    float dx = xj - xi, dy = yj - yi;
    float c = ci * cj + si * sj;
    float s = ci * sj - si * cj;
    float ac = fabsf(c), as = fabsf(s);
    float p = dx * ci + dy * si;
    if (!(fabsf(p) < hli + (hlj * ac + hwj * as))) return 0;
    float q = dy * ci - dx * si;
    if (!(fabsf(q) < hwi + (hlj * as + hwj * ac))) return 0;
    float p2 = dx * cj + dy * sj;
    if (!(fabsf(p2) < hlj + (hli * ac + hwi * as))) return 0;
    float q2 = dy * cj - dx * sj;
    if (!(fabsf(q2) < hwj + (hli * as + hwi * ac))) return 0;
    return 1;
}

TDE_EXPORT void tde_oracle_compute_collision(int32_t B, int32_t A, const float *x, const float *y, const float *psi,
                                             const float *len, const float *wid, const uint8_t *present,
                                             uint8_t *out)
{
    for (int32_t e = 0; e < B; ++e) {
        float c[TDE_MAX_AGENTS], s[TDE_MAX_AGENTS];
        for (int32_t a = 0; a < A; ++a) tde_oracle_sincosf(psi[e * A + a], &s[a], &c[a]);
        for (int32_t i = 0; i < A; ++i) {
            int64_t gi = (int64_t)e * A + i;
            uint8_t hit = 0;
            if (present[gi]) {
                for (int32_t j = 0; j < A; ++j) {
                    int64_t gj = (int64_t)e * A + j;
                    if (j == i || !present[gj]) continue;
                    if (tde_oracle_obb_overlap(x[gi], y[gi], c[i], s[i], 0.5f * len[gi], 0.5f * wid[gi],
                                               x[gj], y[gj], c[j], s[j], 0.5f * len[gj], 0.5f * wid[gj]))
                        hit = 1;
                }
            }
            out[gi] = hit;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* R10: compute_offroad() > 0 — a box corner farther than offroad_threshold from the drivable mesh.  */
/*      Brute force over every triangle of the map (the HIP side uses a grid index; same mask).       */
/* ------------------------------------------------------------------------------------------------ */
static inline float tde_seg_d2(float px, float py, float ax, float ay, float bx, float by)
{
    float abx = bx - ax, aby = by - ay;
    float apx = px - ax, apy = py - ay;
    float len2 = abx * abx + aby * aby;
    float t = 0.0f;
    if (len2 > 0.0f) {
        float inv = 1.0f / len2;
        t = (apx * abx + apy * aby) * inv;
        t = tde_clampf(t, 0.0f, 1.0f);
    }
    float qx = apx - t * abx, qy = apy - t * aby;
    return qx * qx + qy * qy;
}

TDE_EXPORT float tde_oracle_point_tri_d2(float px, float py, const float *t)
{
    float ax = t[0], ay = t[1], bx = t[2], by = t[3], cx = t[4], cy = t[5];
    float e0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    float e1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
    float e2 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);
    if ((e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) || (e0 <= 0.0f && e1 <= 0.0f && e2 <= 0.0f)) return 0.0f;
    float d = tde_seg_d2(px, py, ax, ay, bx, by);
    d = fminf(d, tde_seg_d2(px, py, bx, by, cx, cy));
    d = fminf(d, tde_seg_d2(px, py, cx, cy, ax, ay));
    return d;
}

TDE_EXPORT float tde_oracle_point_mesh_d2(float px, float py, const float *tri, int32_t n_tri)
{
    float best = INFINITY;
    for (int32_t k = 0; k < n_tri; ++k) best = fminf(best, tde_oracle_point_tri_d2(px, py, tri + 6 * (int64_t)k));
    return best;
}

/* Whether some triangle of the mesh lies within sqrt(thr2) of the point: the truth value of
 * `tde_oracle_point_mesh_d2(...) <= thr2`, still a pass over EVERY triangle (no index), with one exact shortcut that makes a
 * town mesh (6e4 triangles) affordable: a triangle whose bounding box is more than R = sqrt(thr2) + 1 cm away from the point
 * along an axis is farther than R from it, and its computed d2 (relative error ~1e-6, coordinates of kilometres rounded to
 * ~1e-4 m) exceeds thr2 - it could not have been the triangle that decides.  tests/test_oracle_math.py holds the two forms
 * against each other on points around the road edges of the junction maps and of the town. */
TDE_EXPORT int tde_oracle_point_near_mesh(float px, float py, const float *tri, int32_t n_tri, float thr2)
{
    const float R = sqrtf(thr2) + 0.01f;
    const float xl = px - R, xh = px + R, yl = py - R, yh = py + R;
    for (int32_t k = 0; k < n_tri; ++k) {
        const float *t = tri + 6 * (int64_t)k;
        if ((t[0] < xl && t[2] < xl && t[4] < xl) || (t[0] > xh && t[2] > xh && t[4] > xh) ||
            (t[1] < yl && t[3] < yl && t[5] < yl) || (t[1] > yh && t[3] > yh && t[5] > yh))
            continue;
        if (tde_oracle_point_tri_d2(px, py, t) <= thr2) return 1;
    }
    return 0;
}

/* the four box corners, in the order FL, FR, RR, RL */
static inline void tde_corners(float x, float y, float c, float s, float hl, float hw, float *cx, float *cy)
{
    float lx = hl * c, ly = hl * s, wx = hw * s, wy = hw * c;
    cx[0] = (x + lx) - wx; cy[0] = (y + ly) + wy;
    cx[1] = (x + lx) + wx; cy[1] = (y + ly) - wy;
    cx[2] = (x - lx) + wx; cy[2] = (y - ly) - wy;
    cx[3] = (x - lx) - wx; cy[3] = (y - ly) + wy;
}

static int tde_agent_offroad(const tde_world *w, int32_t map_id, float x, float y, float c, float s, float hl, float hw,
                             float thr2)
{
    const tde_map *m = &w->maps[map_id];
    const float *tri = w->tri + 6 * (int64_t)m->tri_base;
    float cx[4], cy[4];
    tde_corners(x, y, c, s, hl, hw, cx, cy);
    for (int k = 0; k < 4; ++k)
        if (!tde_oracle_point_near_mesh(cx[k], cy[k], tri, m->n_tri, thr2)) return 1;
    return 0;
}

/* the squared threshold both modes compare d^2 with (tde_config.offroad_threshold_squared) */
static inline float tde_thr2(const tde_config *cfg)
{
    return cfg->offroad_threshold_squared ? cfg->offroad_threshold : cfg->offroad_threshold * cfg->offroad_threshold;
}

/* operator form; map_of_env[e] selects the map of env e */
TDE_EXPORT void tde_oracle_compute_offroad(int32_t B, int32_t A, const float *x, const float *y, const float *psi,
                                           const float *len, const float *wid, const uint8_t *present,
                                           const tde_world *w, const int32_t *map_of_env, float threshold,
                                           uint8_t *out)
{
    float thr2 = threshold * threshold;
    for (int32_t e = 0; e < B; ++e)
        for (int32_t a = 0; a < A; ++a) {
            int64_t g = (int64_t)e * A + a;
            uint8_t off = 0;
            if (present[g]) {
                float s, c;
                tde_oracle_sincosf(psi[g], &s, &c);
                off = (uint8_t)tde_agent_offroad(w, map_of_env[e], x[g], y[g], c, s, 0.5f * len[g], 0.5f * wid[g], thr2);
            }
            out[g] = off;
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* Counter-based RNG for reset (R16): Philox4x32-10 (Salmon et al., SC'11), key = seed, counter =     */
/* (env, episode, block, 0x7DE).  The reference draws from numpy's global Mersenne Twister            */
/* (gym_env.py:320,357-361,194-196); a batched env needs a per-env reproducible stream instead.       */
/* ------------------------------------------------------------------------------------------------ */
TDE_EXPORT void tde_oracle_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4])
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in [0,1) with 24 random bits, exact in fp32 and fp64 */
static inline double tde_u01(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

/* Natural logarithm of a normal fp32 u > 0 - ONE specification shared with the HIP kernels (tde_device.h: log_f32), like
 * tde_oracle_sincosf: u = m * 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m - 1) / (m + 1) (one IEEE division),
 * odd polynomial of degree 9 in s evaluated with explicit fmaf, e * ln 2 added in two parts.  Absolute error < 2e-7 on
 * (0, 1]; what matters here is that CPU and GPU return the same bits. */
TDE_EXPORT float tde_oracle_logf(float u)
{
    uint32_t b;
    memcpy(&b, &u, 4);
    int32_t e = (int32_t)(b >> 23) - 127;
    uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;           /* m in [1, 2) */
    if (mb > 0x3fb504f3u) { mb -= 0x00800000u; e += 1; }     /* m > sqrt(2): halve it */
    float m;
    memcpy(&m, &mb, 4);
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = fmaf(z, 0.11111111f, 0.14285715f);
    p = fmaf(p, z, 0.2f);
    p = fmaf(p, z, 0.33333334f);
    p = fmaf(p, z, 1.0f);
    const float lm = (s + s) * p;
    const float fe = (float)e;
    return fmaf(fe, 6.9314575195e-1f, fmaf(fe, 1.4286067653e-6f, lm));      /* ln 2 = hi + lo, e * hi exact */
}

/* standard normal from two 32-bit random words (Box-Muller, fp32; ref gym_env.py:361 np.random.normal): |z| < 5.8 */
static float tde_normal(uint32_t ra, uint32_t rb)
{
    const float u1 = ((float)(ra >> 8) + 0.5f) * (1.0f / 16777216.0f);       /* (0, 1): the logarithm is finite */
    const float u2 = (float)(rb >> 8) * (1.0f / 16777216.0f);                /* [0, 1) */
    float sn, cs;
    tde_oracle_sincosf(TDE_TWO_PI_F * u2, &sn, &cs);
    return sqrtf(-2.0f * tde_oracle_logf(u1)) * cs;
}

/* WaypointSuiteEnv.reset (gym_env.py:319-349) + set_start_pos (:351-367) + build_simulator's initial
 * tensors (:192-198, :241-247) for one env. */
static void tde_reset_env(const tde_config *cfg, const tde_world *w, tde_state *st, int32_t e)
{
    const int32_t A = st->A;
    uint32_t ep = (uint32_t)st->episode[e];
    const uint32_t ge = cfg->env_base + (uint32_t)e;   /* global env index keys the stream */
    uint32_t r0[4], r1[4];
    tde_oracle_philox(cfg->seed, ge, ep, 0u, 0x7DEu, r0);
    tde_oracle_philox(cfg->seed, ge, ep, 1u, 0x7DEu, r1);
    /* np.random.randint(len(waypoint_suite))  :320 */
    int32_t scn = (int32_t)(((uint64_t)r0[0] * (uint64_t)w->n_scn) >> 32);
    const double *wp = w->wp_xy + (int64_t)scn * w->NW * 2;
    /* start_point = p0 + rand()*(p1 - p0); start_speed = rand()*10  :357-358 */
    double f = tde_u01(r0[1]);
    double sx = wp[0] + f * (wp[2] - wp[0]);
    double sy = wp[1] + f * (wp[3] - wp[1]);
    double speed = tde_u01(r0[2]) * 10.0;
    /* start_orientation = lanelet direction + normal(0, 0.1)  :359-361 (round 3: a true Gaussian, Box-Muller on the shared
     * fp32 log / sincos specifications; rounds 1-2 used Irwin-Hall(12) - 6, whose tails end at +-0.6 rad) */
    /* the lane direction at the start point: find_lanelet_directions(lanelet_map, x, y)[0] (:359) = the world's heading table along
     * the first waypoint segment read at the drawn fraction (tde_world.start_psi, NH entries per scenario), or - without one - the
     * scenario's start_heading (the direction of that segment) */
    float lane_psi = w->scn[scn].start_heading;
    if (w->NH > 0) lane_psi = w->start_psi[(int64_t)scn * w->NH + (int32_t)(f * (double)w->NH)];
    double psi0 = (double)lane_psi + (double)tde_normal(r1[2], r1[3]) * 0.1;

    st->scn[e] = scn;
    st->steps[e] = 0;          /* :339 */
    st->target_idx[e] = 1;     /* :325 */
    st->reached[e] = 0;        /* :338 */
    st->episode[e] = (int32_t)(ep + 1u);
    if (st->ep_return) st->ep_return[e] = 0.0;   /* Monitor.reset: the reward list restarts */

    for (int32_t a = 0; a < A; ++a) {
        int64_t g = (int64_t)e * A + a;
        const tde_spawn *sp = &w->spawn[(int64_t)scn * A + a];
        st->x[g] = sp->x;
        st->y[g] = sp->y;
        st->psi[g] = sp->psi;
        st->v[g] = sp->v;
        st->len[g] = sp->len;
        st->wid[g] = sp->wid;
        st->lr[g] = sp->lr;
        st->vdes[g] = sp->vdes;
        st->route_wp[g] = sp->route_wp;
        st->present[g] = (uint8_t)(sp->present != 0);
        st->collided[g] = 0;
        st->offroad[g] = 0;
    }
    /* slot 0 = ego (:219, :269-271) */
    int64_t g0 = (int64_t)e * A;
    st->x[g0] = (float)sx;
    st->y[g0] = (float)sy;
    st->psi[g0] = (float)psi0;
    st->v[g0] = (float)speed;
    st->present[g0] = 1;
    st->vdes[g0] = 0.0f;
    if (cfg->flags & TDE_F_EGO_ONLY_ATTRS) {
        /* :194-196 */
        st->len[g0] = (float)(tde_u01(r0[3]) * (5.5 - 4.8) + 4.8);
        st->wid[g0] = (float)(tde_u01(r1[0]) * (2.2 - 1.8) + 1.8);
        st->lr[g0] = (float)(tde_u01(r1[1]) * (0.97 - 0.82) + 0.82);
    }
}

TDE_EXPORT int tde_oracle_env_reset(const tde_config *cfg, const tde_world *w, tde_state *st, const uint8_t *mask)
{
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < st->B; ++e)
        if (!mask || mask[e]) tde_reset_env(cfg, w, st, e);
    return 0;
}

/* lights of map m that are red at env step k (the cycle restarts with the episode) */
static uint32_t tde_red_mask(const tde_world *w, const tde_map *m, int32_t k)
{
    if (m->cycle_steps <= 0) return 0u;
    int32_t t = k % m->cycle_steps;
    for (int32_t p = 0; p < m->n_phase; ++p)
        if (t < w->phases[m->phase_base + p].end_step) return w->phases[m->phase_base + p].red_mask;
    return 0u;
}

/* ------------------------------------------------------------------------------------------------ */
/* Heuristic NPC controller (R14 slot): pure-pursuit steering on the NPC's route + gap-keeping speed. */
/* Reads the PRE-step state of every agent of the env.                                               */
/* ------------------------------------------------------------------------------------------------ */
static void tde_npc_action(const tde_config *cfg, const tde_world *w, int32_t A, int32_t i, const float *x,
                           const float *y, const float *c, const float *s, const float *v, const float *len,
                           const float *wid, const uint8_t *present, float vdes, int32_t route, int32_t route_n,
                           int32_t wpi, const tde_map *m, uint32_t red, float *acc_out, float *beta_out)
{
    float amax = cfg->npc_max_accel, smax = cfg->npc_max_steer;
    if (route < 0 || wpi >= route_n) {
        /* no route (left): brake to a stop, wheels straight */
        *acc_out = tde_clampf(cfg->npc_k_speed * (0.0f - v[i]), -amax, amax);
        *beta_out = 0.0f;
        return;
    }
    const float *t = w->route_xy + ((int64_t)route * w->RW + wpi) * 2;
    float cp = c[i], sp = s[i];
    float dx = t[0] - x[i], dy = t[1] - y[i];
    float fwd = dx * cp + dy * sp;
    float lat = dy * cp - dx * sp;
    float dist = sqrtf(dx * dx + dy * dy);
    float sin_err = lat / fmaxf(dist, 1e-3f);
    float beta;
    if (fwd < 0.0f) beta = copysignf(smax, lat);
    else beta = tde_clampf(cfg->npc_k_steer * sin_err, -smax, smax);
    /* gap to the nearest agent that is (a) ahead in the own lane corridor, or (b) ahead inside the yield cone,
     * not oncoming, and of higher priority (lower slot index) */
    float gap = 1e30f;
    for (int32_t j = 0; j < A; ++j) {
        if (j == i || !present[j]) continue;
        float ex = x[j] - x[i], ey = y[j] - y[i];
        float fj = ex * cp + ey * sp;
        float lj = ey * cp - ex * sp;
        if (fj > 0.0f) {
            float halfw = cfg->npc_lane_half + 0.5f * wid[j];
            float al = fabsf(lj);
            int inlane = al < halfw;
            float hd = cp * c[j] + sp * s[j];
            int cone = (j < i) && (fj < cfg->npc_cone_range) && (al < halfw + cfg->npc_cone_k * fj) && (hd > -0.5f);
            if (inlane || cone) {
                float g = fj - 0.5f * (len[i] + len[j]);
                gap = fminf(gap, g);
            }
        }
    }
    /* a red stop line ahead in the own lane (same travel direction) counts as a standing leader; the IAI NPCs of the
     * reference are fed the light state too (gym_env.py:290-291).  Only evaluated while the front bumper has not yet
     * crossed the line centre, so a car caught on the line by the phase change drives on. */
    if (m && red) {
        for (int32_t k = 0; k < m->n_stop; ++k) {
            const tde_stopline *sl = &w->stoplines[m->stop_base + k];
            if (!((red >> sl->light) & 1u)) continue;
            float ex = sl->x - x[i], ey = sl->y - y[i];
            float fj = ex * cp + ey * sp;
            float lj = ey * cp - ex * sp;
            float hd = cp * sl->c + sp * sl->s;
            float g = fj - 0.5f * len[i];
            if (g > 0.0f && fabsf(lj) < sl->hw && hd > 0.5f) gap = fminf(gap, g + cfg->npc_gap_s0 - 1.0f);
        }
    }
    /* speed from which a brake at amax/2 stops inside the gap */
    float vd = fminf(vdes, sqrtf(amax * fmaxf(gap - cfg->npc_gap_s0, 0.0f)));
    *acc_out = tde_clampf(cfg->npc_k_speed * (vd - v[i]), -amax, amax);
    *beta_out = beta;
}

/* ------------------------------------------------------------------------------------------------ */
/* R6/R7/R8/R11/R12: the part of the step the reference owns (pinned by tests/golden/reward_golden.json). */
/*   pre  = ego state before simulator.step  (last_x,last_y,last_psi,last_speed, gym_env.py:371-375)  */
/*   post = ego state after it; k = environment_steps after the increment at :116                     */
/* ------------------------------------------------------------------------------------------------ */
static void tde_reward_core(const tde_config *cfg, const double *wp, int32_t n_wp, const float pre[4],
                            const float post[4], int off, int col, int tl, int32_t k, int32_t *target_idx,
                            int32_t *reached, float *reward, uint8_t *terminated, uint8_t *truncated, double *info,
                            int32_t *info_reached)
{
    /* get_reward :396-411 (float64 on fp32 state) */
    double ddx = (double)post[0] - (double)pre[0], ddy = (double)post[1] - (double)pre[1];
    double d = sqrt(ddx * ddx + ddy * ddy);                                  /* math.dist :401 */
    double dist_r = (d > cfg->distance_cutoff) ? cfg->distance_bonus : 0.0;  /* :402 */
    float dpsi = post[2] - pre[2];                                           /* fp32 tensor subtraction */
    double psi_r = (1.0 - cos((double)dpsi)) * (-cfg->heading_penalty);      /* :403 */
    /* check_reach_target :391-394 */
    int reach = 0;
    int32_t ti = *target_idx;
    if (ti < n_wp) {
        double tx = (double)post[0] - wp[2 * ti], ty = (double)post[1] - wp[2 * ti + 1];
        reach = sqrt(tx * tx + ty * ty) < cfg->reach_radius;
    }
    double reach_r = 0.0;
    if (reach) { reach_r = cfg->waypoint_bonus; *reached += 1; }             /* :404-408 */
    *reward = (float)((reach_r + dist_r) + psi_r);                           /* :409-411 */
    /* is_terminated :413-417 */
    *terminated = (uint8_t)(cfg->terminated_at_infraction && (off || col || tl));
    /* is_truncated :134-135 */
    *truncated = (uint8_t)(k >= cfg->max_steps);
    /* get_info :419-437 */
    if (info) {
        info[0] = (double)fabsf((pre[2] - post[2]) / 0.1f);   /* psi_smoothness :432 */
        info[1] = (double)fabsf((pre[3] - post[3]) / 0.1f);   /* speed_smoothness :435 */
        info[2] = psi_r;                                      /* :433 */
        info[3] = dist_r;                                     /* :434 */
    }
    if (info_reached) *info_reached = *reached;                              /* :425,431 */
    /* :378-383 advance after reward/info were produced */
    if (reach) *target_idx = ti + 1;
}

/* Batched operator form over n independent envs (SoA): increments steps[i] (gym_env.py:116) and applies the
 * reward/termination logic to the given pre/post ego states and infraction flags. */
TDE_EXPORT int tde_oracle_waypoint_reward(const tde_config *cfg, int32_t n, const float *pre_x, const float *pre_y,
                                          const float *pre_psi, const float *pre_v, const float *x, const float *y,
                                          const float *psi, const float *v, const uint8_t *offroad,
                                          const uint8_t *collided, const uint8_t *tl_violation, const double *wp_xy,
                                          const int32_t *wp_n, int32_t NW, const int32_t *scn, int32_t *steps,
                                          int32_t *target_idx, int32_t *reached, float *reward, uint8_t *terminated,
                                          uint8_t *truncated, double *info, int32_t *info_reached)
{
    for (int32_t i = 0; i < n; ++i) {
        const float pre[4] = {pre_x[i], pre_y[i], pre_psi[i], pre_v[i]};
        const float post[4] = {x[i], y[i], psi[i], v[i]};
        steps[i] += 1;
        int32_t s = scn[i];
        tde_reward_core(cfg, wp_xy + (int64_t)s * NW * 2, wp_n[s], pre, post, offroad[i], collided[i],
                        tl_violation ? tl_violation[i] : 0, steps[i], &target_idx[i], &reached[i], &reward[i],
                        &terminated[i], &truncated[i], info ? info + 4 * (int64_t)i : NULL,
                        info_reached ? &info_reached[i] : NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* One env, one timestep: WaypointSuiteEnv.step (gym_env.py:369-389) over GymEnv.step (:115-120).     */
/* ------------------------------------------------------------------------------------------------ */
/* compute_traffic_lights_violations() > 0 for one box (gym_env.py:144,415,429): the box overlaps a stop line whose
 * light is red at env step k.  PARITY UNPINNED (torchdrivesim internals); the light cycle restarts with the episode. */
static int tde_tl_violation(const tde_world *w, const tde_map *m, int32_t k, float x, float y, float c, float s, float hl,
                            float hw)
{
    if (m->cycle_steps <= 0 || m->n_stop <= 0) return 0;
    const uint32_t red = tde_red_mask(w, m, k);
    for (int32_t i = 0; i < m->n_stop; ++i) {
        const tde_stopline *sl = &w->stoplines[m->stop_base + i];
        if (((red >> sl->light) & 1u) && tde_oracle_obb_overlap(x, y, c, s, hl, hw, sl->x, sl->y, sl->c, sl->s, sl->hl, sl->hw))
            return 1;
    }
    return 0;
}

typedef struct {
    float reward;
    uint8_t terminated, truncated, tl;
} tde_env_out;

static tde_env_out tde_step_env(const tde_config *cfg, const tde_world *w, tde_state *st, int32_t e, float a_acc,
                                float a_steer)
{
    const int32_t A = st->A;
    const int64_t g0 = (int64_t)e * A;
    const uint32_t F = cfg->flags;
    float *X = st->x + g0, *Y = st->y + g0, *P = st->psi + g0, *V = st->v + g0;
    const float *L = st->len + g0, *W = st->wid + g0, *LR = st->lr + g0;
    const uint8_t *present = st->present + g0;
    tde_env_out out = {0.0f, 0, 0, 0};

    /* :371-375 snapshot of the pre-step state (all agents: the NPC controller reads it too) */
    float px[TDE_MAX_AGENTS], py[TDE_MAX_AGENTS], pp[TDE_MAX_AGENTS], pv[TDE_MAX_AGENTS];
    float pc[TDE_MAX_AGENTS], ps[TDE_MAX_AGENTS];
    for (int32_t a = 0; a < A; ++a) {
        px[a] = X[a]; py[a] = Y[a]; pp[a] = P[a]; pv[a] = V[a];
        tde_oracle_sincosf(pp[a], &ps[a], &pc[a]);
    }

    /* :116 */
    st->steps[e] += 1;
    const int32_t k = st->steps[e];

    /* :117 simulator.step(action): ego takes the external action; NPC slots take the controller's action
     * (or coast with zero action, as NPCWrapper does before teleporting), then replayed agents are
     * overwritten with their recorded state at time k. */
    const tde_spawn *spawn = w->spawn + (int64_t)st->scn[e] * A;   /* route / replay ids of the env's slots */
    const tde_map *lights_map = (F & TDE_F_TRAFFIC_LIGHTS) ? &w->maps[w->scn[st->scn[e]].map] : NULL;
    const uint32_t red_now = lights_map ? tde_red_mask(w, lights_map, st->steps[e]) : 0u;   /* lights at decision time */
    for (int32_t a = 0; a < A; ++a) {
        if (!present[a]) continue;
        float acc = 0.0f, beta = 0.0f;
        if (a == 0) { acc = a_acc; beta = a_steer; }
        else if ((F & TDE_F_NPC) && (k > 1 || (F & TDE_F_NPC_FIRST_STEP)))
            /* (k == 1, the first step of an episode: the NPCs coast with the zero action - the controller reads the scene of
             * the previous step, which a fresh episode does not have; DESIGN.md section 2, R14 - unless TDE_F_NPC_FIRST_STEP asks
             * for the reference's behaviour: its NPCs are driven from step one, gym_env.py:285-294) */
            tde_npc_action(cfg, w, A, a, px, py, pc, ps, pv, L, W, present, st->vdes[g0 + a], spawn[a].route,
                           spawn[a].route_n, st->route_wp[g0 + a], lights_map, red_now, &acc, &beta);
        float nx = px[a], ny = py[a], np_ = pp[a], nv = pv[a];
        tde_oracle_bicycle(&nx, &ny, &np_, &nv, LR[a], acc, beta, cfg->dt);
        if ((F & TDE_F_REPLAY) && a > 0) {
            int32_t row = spawn[a].replay;
            if (row >= 0 && k < spawn[a].replay_len) {
                const float *r = w->replay_states + ((int64_t)row * w->RT + k) * 4;
                nx = r[0]; ny = r[1]; np_ = r[2]; nv = r[3];
            }
        }
        X[a] = nx; Y[a] = ny; P[a] = np_; V[a] = nv;
        /* NPC route waypoint switch, judged on the post-step position */
        if ((F & TDE_F_NPC) && a > 0) {
            int32_t route = spawn[a].route, wpi = st->route_wp[g0 + a];
            if (route >= 0 && wpi < spawn[a].route_n) {
                const float *t = w->route_xy + ((int64_t)route * w->RW + wpi) * 2;
                float dx = t[0] - nx, dy = t[1] - ny;
                if (dx * dx + dy * dy < cfg->npc_reach * cfg->npc_reach) st->route_wp[g0 + a] = wpi + 1;
            }
        }
    }

    /* infractions on the post-step state (:142-143) */
    float c[TDE_MAX_AGENTS], s[TDE_MAX_AGENTS];
    for (int32_t a = 0; a < A; ++a) tde_oracle_sincosf(P[a], &s[a], &c[a]);
    for (int32_t i = 0; i < A; ++i) {
        uint8_t hit = 0;
        if (present[i])
            for (int32_t j = 0; j < A; ++j) {
                if (j == i || !present[j]) continue;
                if (tde_oracle_obb_overlap(X[i], Y[i], c[i], s[i], 0.5f * L[i], 0.5f * W[i], X[j], Y[j], c[j], s[j],
                                           0.5f * L[j], 0.5f * W[j]))
                    hit = 1;
            }
        st->collided[g0 + i] = hit;
    }
    if (F & TDE_F_OFFROAD) {
        float thr2 = tde_thr2(cfg);
        int32_t map_id = w->scn[st->scn[e]].map;
        for (int32_t a = 0; a < A; ++a)
            st->offroad[g0 + a] = present[a] ? (uint8_t)tde_agent_offroad(w, map_id, X[a], Y[a], c[a], s[a],
                                                                          0.5f * L[a], 0.5f * W[a], thr2)
                                             : 0;
    } else {
        for (int32_t a = 0; a < A; ++a) st->offroad[g0 + a] = 0;
    }

    if (F & TDE_F_TRAFFIC_LIGHTS)
        out.tl = (uint8_t)tde_tl_violation(w, &w->maps[w->scn[st->scn[e]].map], k, X[0], Y[0], c[0], s[0], 0.5f * L[0],
                                           0.5f * W[0]);
    if (st->tl_violation) st->tl_violation[e] = out.tl;
    if (F & TDE_F_REWARD) {
        const int32_t scn = st->scn[e];
        const float pre[4] = {px[0], py[0], pp[0], pv[0]};
        const float post[4] = {X[0], Y[0], P[0], V[0]};
        tde_reward_core(cfg, w->wp_xy + (int64_t)scn * w->NW * 2, w->scn[scn].wp_n, pre, post, st->offroad[g0],
                        st->collided[g0], out.tl, k, &st->target_idx[e], &st->reached[e], &out.reward, &out.terminated,
                        &out.truncated, st->info ? st->info + 4 * (int64_t)e : NULL,
                        st->info_reached ? &st->info_reached[e] : NULL);
    }
    return out;
}

static void tde_ego_magnitudes(const tde_config *cfg, const tde_world *w, const tde_state *st, int32_t e, int do_coll, int do_off,
                               float *out);

TDE_EXPORT int tde_oracle_env_step(const tde_config *cfg, const tde_world *w, tde_state *st)
{
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < st->B; ++e) {
        tde_env_out o = tde_step_env(cfg, w, st, e, st->action[2 * e], st->action[2 * e + 1]);
        /* tde_state.magnitudes: what get_info reports under "offroad" / "collision" (gym_env.py:427-428), of the state the step
         * left, before any re-spawn.  A magnitude is non-zero only under its flag (collision: the same predicate; offroad: a
         * corner beyond the threshold has d^2 > thr^2 and sqrt(d^2) <= thr otherwise), so the brute-force pass over every
         * triangle only runs for the egos the step flagged - the same values as tde_oracle_ego_infractions on every env
         * (tests/test_oracle_properties.py) */
        if (st->magnitudes)
            tde_ego_magnitudes(cfg, w, st, e, st->collided[(int64_t)e * st->A], st->offroad[(int64_t)e * st->A],
                               st->magnitudes + 4 * (int64_t)e);
        st->reward[e] = o.reward;
        st->terminated[e] = o.terminated;
        st->truncated[e] = o.truncated;
        if (st->done_bits)                                  /* the ego's flags before a re-spawn clears them */
            st->done_bits[e] = (uint8_t)(o.terminated | (o.truncated << 1) | (st->offroad[(int64_t)e * st->A] << 2) |
                                         (st->collided[(int64_t)e * st->A] << 3) | (o.tl << 4));
        if (st->ep_return) {                                /* Monitor.step: rewards.append(float(reward)); ep_rew = sum(rewards) */
            st->ep_return[e] = st->ep_return[e] + (double)o.reward;
            if (o.terminated || o.truncated) {
                if (st->ep_final) st->ep_final[e] = st->ep_return[e];
                if (st->ep_final_len) st->ep_final_len[e] = st->steps[e];
            }
        }
        if ((cfg->flags & TDE_F_AUTORESET) && (o.terminated || o.truncated)) tde_reset_env(cfg, w, st, e);
    }
    return 0;
}

/* K consecutive steps with actions taken from a resident [K][B][2] buffer. */
TDE_EXPORT int tde_oracle_env_rollout(const tde_config *cfg, const tde_world *w, tde_state *st, const tde_rollout *ro)
{
    const int32_t B = st->B;
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < B; ++e) {
        for (int32_t k = 0; k < ro->K; ++k) {
            const float *act = ro->actions + ((int64_t)k * B + e) * 2;
            int64_t g0 = (int64_t)e * st->A;
            tde_env_out o = tde_step_env(cfg, w, st, e, act[0], act[1]);
            ro->reward[(int64_t)k * B + e] = o.reward;
            ro->done[(int64_t)k * B + e] = (uint8_t)(o.terminated | (o.truncated << 1) | (st->offroad[g0] << 2) |
                                                     (st->collided[g0] << 3) | (o.tl << 4));
            st->reward[e] = o.reward;
            st->terminated[e] = o.terminated;
            st->truncated[e] = o.truncated;
            if ((cfg->flags & TDE_F_AUTORESET) && (o.terminated || o.truncated)) {
                /* the Monitor-style episode statistics belong to the closed-loop step API: a rollout leaves them alone */
                const double keep = st->ep_return ? st->ep_return[e] : 0.0;
                tde_reset_env(cfg, w, st, e);
                if (st->ep_return) st->ep_return[e] = keep;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* R13: get_obs -> simulator.render_egocentric() (gym_env.py:122-124), restated as a point-sampled     */
/* raster (layer definitions in include/tde_abi.h).  PARITY UNPINNED: torchdrivesim's renderer is not   */
/* in the reference repository; palette, sampling and draw order are defined here.                     */
/* ------------------------------------------------------------------------------------------------ */
static void tde_render_env(const tde_config *cfg, const tde_world *w, const tde_state *st, const tde_render *rd,
                           int32_t e)
{
    static const uint8_t BG[3] = {TDE_RGB_BACKGROUND}, ROAD[3] = {TDE_RGB_ROAD}, WP[3] = {TDE_RGB_WAYPOINT},
                         NPC[3] = {TDE_RGB_NPC}, EGO[3] = {TDE_RGB_EGO}, STOP_RED[3] = {TDE_RGB_STOP_RED},
                         STOP_GO[3] = {TDE_RGB_STOP_GO};
    if (rd->only && !rd->only[e]) return;                 /* masked call: this view is left as it is */
    const int32_t A = st->A, H = rd->H, W = rd->W;
    const int32_t ns = rd->n_stack > 1 ? rd->n_stack : 1;
    const int64_t g0 = (int64_t)e * A;
    const int64_t plane = (int64_t)H * W;
    uint8_t *out = rd->out + (int64_t)e * 3 * ns * plane;
    /* frame stack: a full call shifts the older frames down; a masked call (rd->only) re-renders the newest frame in
     * place; a view whose episode just started (rd->fresh) shows blank older frames (VecFrameStack after a reset) */
    if (ns > 1 && !rd->only) memmove(out, out + 3 * plane, (size_t)(3 * (ns - 1) * plane));
    if (ns > 1 && rd->fresh && (rd->fresh[e] & 3)) memset(out, 0, (size_t)(3 * (ns - 1) * plane));
    uint8_t *img = out + 3 * (ns - 1) * plane;
    const int32_t scn = st->scn[e];
    const tde_map *m = &w->maps[w->scn[scn].map];
    const float *tri = w->tri + 6 * (int64_t)m->tri_base;
    const float thr2 = tde_thr2(cfg);
    const int lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const uint32_t red = lights ? tde_red_mask(w, m, st->steps[e]) : 0u;     /* light state at the env's current step */
    const float lsign = (rd->flags & TDE_RENDER_LEFT_HANDED) ? -1.0f : 1.0f;
    const uint8_t *ego_col = (rd->flags & TDE_RENDER_PLAIN_EGO) ? NPC : EGO;
    /* Pixel -> world as ONE affine map per view, evaluated with explicit fused multiply-adds (round 3; the unfused chain
     * cost the rasteriser ~10 operations per pixel and object):
     *     u = (H/2 - 0.5) - r,   v = (W/2 - 0.5) - c          (exact: small half-integers)
     *     world = ego + u * (ax, ay) + v * (bx, by),  (ax, ay) = res * (cos, sin) of the ego's heading (one image row up),
     *                                                 (bx, by) = (-rs * sin, rs * cos), rs = res * lsign (one column left)
     * and every per-object quantity tested per pixel (box-frame coordinates p, q; offset to a waypoint) is the affine
     * function of (u, v) it is in exact arithmetic, its three coefficients formed once per object (plain fp32
     * operations, written out below) and then evaluated as fmaf(v, cb, fmaf(u, ca, c0)).  The HIP rasteriser evaluates
     * the same expressions (csrc/tde_raster.h), so pixels are bit-identical. */
    const float ex = st->x[g0], ey = st->y[g0];
    float ce, se;
    tde_oracle_sincosf(st->psi[g0], &se, &ce);
    const float res = rd->fov / (float)W;
    const float rs = res * lsign;
    const float ax = res * ce, ay = res * se, bx = (-rs) * se, by = rs * ce;
    const float hu = 0.5f * (float)H - 0.5f, hv = 0.5f * (float)W - 0.5f;
    /* box-frame coefficients of agent / stop-line boxes: p = along the box, q = across it */
    float bp0[TDE_MAX_AGENTS], bpa[TDE_MAX_AGENTS], bpb[TDE_MAX_AGENTS], bq0[TDE_MAX_AGENTS], bqa[TDE_MAX_AGENTS],
        bqb[TDE_MAX_AGENTS];
    for (int32_t a = 0; a < A; ++a) {
        float cb, sb;
        tde_oracle_sincosf(st->psi[g0 + a], &sb, &cb);
        const float dx = ex - st->x[g0 + a], dy = ey - st->y[g0 + a];
        bp0[a] = dx * cb + dy * sb; bpa[a] = ax * cb + ay * sb; bpb[a] = bx * cb + by * sb;
        bq0[a] = dy * cb - dx * sb; bqa[a] = ay * cb - ax * sb; bqb[a] = by * cb - bx * sb;
    }
    const double *wp = w->wp_xy + (int64_t)scn * w->NW * 2;
    const int32_t n_wp = w->scn[scn].wp_n, ti = st->target_idx[e];
    for (int32_t r = 0; r < H; ++r)
        for (int32_t c = 0; c < W; ++c) {
            const float u = hu - (float)r, v = hv - (float)c;
            const float wx = fmaf(v, bx, fmaf(u, ax, ex));
            const float wy = fmaf(v, by, fmaf(u, ay, ey));
            const uint8_t *col = BG;
            if (tde_oracle_point_near_mesh(wx, wy, tri, m->n_tri, thr2)) col = ROAD;
            if (lights)
                for (int32_t k = 0; k < m->n_stop; ++k) {
                    const tde_stopline *sl = &w->stoplines[m->stop_base + k];
                    const float dx = ex - sl->x, dy = ey - sl->y;
                    const float p0 = dx * sl->c + dy * sl->s, pa = ax * sl->c + ay * sl->s, pb = bx * sl->c + by * sl->s;
                    const float q0 = dy * sl->c - dx * sl->s, qa = ay * sl->c - ax * sl->s, qb = by * sl->c - bx * sl->s;
                    const float p = fmaf(v, pb, fmaf(u, pa, p0)), q = fmaf(v, qb, fmaf(u, qa, q0));
                    if (fabsf(p) <= sl->hl && fabsf(q) <= sl->hw) col = ((red >> sl->light) & 1u) ? STOP_RED : STOP_GO;
                }
            for (int32_t k = ti; k < n_wp; ++k) {
                const float dx0 = ex - (float)wp[2 * k], dy0 = ey - (float)wp[2 * k + 1];
                const float dx = fmaf(v, bx, fmaf(u, ax, dx0)), dy = fmaf(v, by, fmaf(u, ay, dy0));
                if (fmaf(dx, dx, dy * dy) <= TDE_WAYPOINT_RADIUS * TDE_WAYPOINT_RADIUS) col = WP;
            }
            for (int32_t a = A - 1; a >= 0; --a) {
                if (!st->present[g0 + a]) continue;
                const float p = fmaf(v, bpb[a], fmaf(u, bpa[a], bp0[a])), q = fmaf(v, bqb[a], fmaf(u, bqa[a], bq0[a]));
                if (fabsf(p) <= 0.5f * st->len[g0 + a] && fabsf(q) <= 0.5f * st->wid[g0 + a]) col = a ? NPC : ego_col;
            }
            for (int ch = 0; ch < 3; ++ch) img[ch * plane + (int64_t)r * W + c] = col[ch];
        }
}

TDE_EXPORT int tde_oracle_render_ego(const tde_config *cfg, const tde_world *w, const tde_state *st, const tde_render *rd)
{
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < st->B; ++e) tde_render_env(cfg, w, st, rd, e);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* Infraction MAGNITUDES of the ego (what info["offroad"] / info["collision"] hold in the reference,  */
/* gym_env.py:427-428: simulator.compute_offroad() / compute_collision() for the exposed agent).       */
/* PARITY UNPINNED (torchdrivesim absent); defined here:                                               */
/*   offroad   = sum over the four corners of clamp(dist - threshold, min = 0), dist = distance of the   */
/*               corner to the mesh (brute force over every triangle; the SQUARED distance under         */
/*               offroad_threshold_squared), accumulated in fp32 in corner order FL, FR, RR, RL;         */
/*   collision = sum over the other present agents whose box overlaps the ego's (strict SAT) of the IoU  */
/*               of the two boxes (tde_oracle_box_iou), in slot order; also their number.               */
/* out = float32 [B][4] = offroad, collision (sum of IoUs), number of overlapping agents, 0.           */
/* ------------------------------------------------------------------------------------------------ */
/* IoU of two oriented boxes by Sutherland-Hodgman clipping of box 0 by the four edges of box 1 and the shoelace formula, fp32, one
 * operation at a time (the HIP kernel repeats it: csrc/tde_magnitudes.h).  The published form of torchdrivesim's
 * CollisionMetric.nograd is this value summed over the other agents; tests/test_second_opinions.py holds a float64 version of
 * the same construction against the SAT mask. */
static void tde_box_corners_ccw(float x, float y, float c, float s, float hl, float hw, float *px, float *py)
{
    const float lx = hl * c, ly = hl * s, wx = hw * s, wy = hw * c;
    px[0] = (x + lx) - wx; py[0] = (y + ly) + wy;       /* front left  */
    px[1] = (x - lx) - wx; py[1] = (y - ly) + wy;       /* rear left   */
    px[2] = (x - lx) + wx; py[2] = (y - ly) - wy;       /* rear right  */
    px[3] = (x + lx) + wx; py[3] = (y + ly) - wy;       /* front right */
}

TDE_EXPORT float tde_oracle_box_iou(float x0, float y0, float c0, float s0, float hl0, float hw0, float x1, float y1, float c1,
                                    float s1, float hl1, float hw1)
{
    float ax[8], ay[8], bx[8], by[8], qx[4], qy[4];
    int n = 4;
    tde_box_corners_ccw(x0, y0, c0, s0, hl0, hw0, ax, ay);
    tde_box_corners_ccw(x1, y1, c1, s1, hl1, hw1, qx, qy);
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

/* the four magnitude values of env e's ego on the CURRENT state; do_coll / do_off = 0 skips that magnitude (left 0) */
static void tde_ego_magnitudes(const tde_config *cfg, const tde_world *w, const tde_state *st, int32_t e, int do_coll, int do_off,
                               float *out)
{
    const int32_t A = st->A;
    const int64_t g0 = (int64_t)e * A;
    float omag = 0.0f, cmag = 0.0f, nmag = 0.0f;
    if (st->present[g0]) {
        float se, ce;
        tde_oracle_sincosf(st->psi[g0], &se, &ce);
        const float hl = 0.5f * st->len[g0], hw = 0.5f * st->wid[g0];
        int n = 0;
        for (int32_t j = 1; j < A && do_coll; ++j) {
            if (!st->present[g0 + j]) continue;
            float sj, cj;
            tde_oracle_sincosf(st->psi[g0 + j], &sj, &cj);
            const float hlj = 0.5f * st->len[g0 + j], hwj = 0.5f * st->wid[g0 + j];
            if (tde_oracle_obb_overlap(st->x[g0], st->y[g0], ce, se, hl, hw, st->x[g0 + j], st->y[g0 + j], cj, sj, hlj, hwj)) {
                n += 1;     /* (IoU only for pairs the mask's predicate calls overlapping: magnitude > 0 <=> collided) */
                cmag = cmag + tde_oracle_box_iou(st->x[g0], st->y[g0], ce, se, hl, hw, st->x[g0 + j], st->y[g0 + j], cj, sj, hlj, hwj);
            }
        }
        nmag = (float)n;
        if (do_off && (cfg->flags & TDE_F_OFFROAD)) {
            const tde_map *m = &w->maps[w->scn[st->scn[e]].map];
            const float *tri = w->tri + 6 * (int64_t)m->tri_base;
            float cx[4], cy[4];
            tde_corners(st->x[g0], st->y[g0], ce, se, hl, hw, cx, cy);
            for (int k = 0; k < 4; ++k) {
                const float d2 = tde_oracle_point_mesh_d2(cx[k], cy[k], tri, m->n_tri);
                const float dist = cfg->offroad_threshold_squared ? d2 : sqrtf(d2);
                omag = omag + fmaxf(dist - cfg->offroad_threshold, 0.0f);
            }
        }
    }
    out[0] = omag; out[1] = cmag; out[2] = nmag; out[3] = 0.0f;
}

TDE_EXPORT int tde_oracle_ego_infractions(const tde_config *cfg, const tde_world *w, const tde_state *st, float *out)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t e = 0; e < st->B; ++e) tde_ego_magnitudes(cfg, w, st, e, 1, 1, out + 4 * (int64_t)e);
    return 0;
}

TDE_EXPORT int tde_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

TDE_EXPORT void tde_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

TDE_EXPORT int tde_oracle_abi_version(void) { return TDE_ABI_VERSION; }
