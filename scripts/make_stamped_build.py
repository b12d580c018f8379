"""Builds ab/libS.so: the library with s_memtime stamps / counters patched into ONE kernel, plus the debug entry point
`tde_debug_stamps` that scripts/phase_stamps.py, render_stamps.py and trip_counts.py read.  The instrumentation is
applied to a copy of csrc/tde_kernels.h (built as the single-unit form of the library) by text substitution (asserted), so the shipped kernels stay clean.

    python scripts/make_stamped_build.py duo|trio|step3|trips      # then  TDE_HIP_LIB=$PWD/ab/libS.so python scripts/...

duo    : driver / judge phases of env_rollout_duo_kernel        -> scripts/phase_stamps.py
trio   : driver / judge C / judge O phases of env_rollout_trio_kernel -> scripts/phase_stamps.py trio
step3  : driver / judge C / judge O milestones of env_step_trio_kernel -> scripts/step_stamps.py
trips  : exact-loop trip counts of the controller / collision   -> scripts/trip_counts.py
The stamps cost 10-40 % themselves: read the shares, not the totals.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "torchdriveenv_amd", "csrc", "tde_kernels.h")    # (the device code of every unit of the library)
sys.path.insert(0, ROOT)
from torchdriveenv_amd.build import FLAGS  # noqa: E402

PRELUDE = '''struct Stamps {
    unsigned long long last, acc[12];
    __device__ void start() { for (int i = 0; i < 12; ++i) acc[i] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ void mark(int k) { unsigned long long n = __builtin_amdgcn_s_memtime(); acc[k] += n - last; last = n; }
};
__device__ unsigned long long g_stamps[24];
// light-weight marker (trio patch): only `last` lives in registers; lane 0 of the wavefront adds the elapsed ticks to
// a per-workgroup LDS accumulator (every index is written by one role only), flushed to g_stamps once per launch
__shared__ unsigned long long s_acc[24];
__device__ __forceinline__ void tde_mark(unsigned long long *last, int k)
{
    if (!last) return;
    const unsigned long long n = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) s_acc[k] += n - *last;
    *last = n;
}
// per-workgroup slots, plain read-modify-write by one lane (global atomics from every wavefront of every launch
// serialise at ~12 ns each and distort what they measure); tde_debug_stamps sums the slots on the host
__device__ unsigned long long g_wg[4096][24];
__device__ __forceinline__ void tde_flush(int lo, int hi)
{
    if ((threadIdx.x & 63) == 0) for (int i = lo; i < hi; ++i) g_wg[blockIdx.x & 4095][i] += s_acc[i];
}
'''
EPILOGUE = '''
extern "C" __attribute__((visibility("default"))) int tde_debug_stamps(unsigned long long *out, int clear)
{
    unsigned long long z[24] = {0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tde::g_stamps), sizeof(z));
    if (e == hipSuccess && clear) e = hipMemcpyToSymbol(HIP_SYMBOL(tde::g_stamps), z, sizeof(z));
    static unsigned long long wg[4096][24];
    if (e == hipSuccess) e = hipMemcpyFromSymbol(wg, HIP_SYMBOL(tde::g_wg), sizeof(wg));
    if (e == hipSuccess) {
        for (int b = 0; b < 4096; ++b) for (int i = 0; i < 24; ++i) out[i] += wg[b][i];
        if (clear) { memset(wg, 0, sizeof(wg)); e = hipMemcpyToSymbol(HIP_SYMBOL(tde::g_wg), wg, sizeof(wg)); }
    }
    return (int)e;
}
// the raw per-workgroup slots (wide patch: slots 20 / 21 / 22 = s_memrealtime, 100 MHz, at the workgroup's entry / the drive wavefront's end /
// the judge wavefront's end of the LAST launch)
extern "C" __attribute__((visibility("default"))) int tde_debug_wg(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tde::g_wg), sizeof(unsigned long long) * 4096 * 24);
}
'''
ANCHOR = "template <int A> struct MaskOf { using type = uint32_t; };"


def sub(text, old, new, count=1):
    assert old in text, f"anchor not found (the kernel source moved on):\n{old[:120]}"
    return text.replace(old, new, count)


def kernel_span(s, name):
    a = s.index(f"void {name}(")
    b = s.index("\n}\n", a) + 3
    return a, b


def patch_duo(s):
    a, b = kernel_span(s, "env_rollout_duo_kernel")
    k = s[a:b]
    k = sub(k, "        RewardOut rw{};\n        for (int i = 0; i < ro.K; ++i) {",
            "        RewardOut rw{};\n        Stamps stp; stp.start();\n        for (int i = 0; i < ro.K; ++i) {")
    k = sub(k, "                sincos_f32(npsi, ns, nc);\n", "                sincos_f32(npsi, ns, nc);\n                stp.mark(0);\n")
    k = sub(k, "                if (pass) break;\n                lds_barrier();                               // A: done(i-1) is published\n",
            "                stp.mark(1);\n                if (pass) break;\n                lds_barrier();                               // A: done(i-1) is published\n                stp.mark(2);\n")
    k = sub(k, "            write_rows(sh, p, lane, live, ag, c0, s0, cfg.npc_lane_half);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n",
            "            write_rows(sh, p, lane, live, ag, c0, s0, cfg.npc_lane_half);\n            stp.mark(3);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n            stp.mark(4);\n")
    k = sub(k, "            act = act_next;\n        }\n", "            act = act_next;\n        }\n        if (lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[i], stp.acc[i]);\n")
    k = sub(k, "        lds_barrier();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1;\n            lds_barrier();                                   // A\n            lds_barrier();                                   // B: rows of step i are in buffer p\n",
            "        lds_barrier();\n        Stamps stp; stp.start();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1;\n            lds_barrier();                                   // A\n            stp.mark(0);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n            stp.mark(1);\n")
    k = sub(k, "            bool off = false;\n            if (F & TDE_F_OFFROAD) off = offroad_resolve<false, BIG || TDE_ROLLOUT_CLS2>(w, corners, thr2, cx.m.rec_base);\n",
            "            stp.mark(2);\n            bool off = false;\n            if (F & TDE_F_OFFROAD) off = offroad_resolve<false, BIG || TDE_ROLLOUT_CLS2>(w, corners, thr2, cx.m.rec_base);\n            stp.mark(3);\n")
    k = sub(k, "            if (lane == 0) sh.done = any;\n", "            stp.mark(4);\n            if (lane == 0) sh.done = any;\n")
    k = sub(k, "                o.respawned = true;\n            }\n        }\n",
            "                o.respawned = true;\n            }\n            stp.mark(5);\n        }\n        if (lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[12 + i], stp.acc[i]);\n")
    return s[:a] + k + s[b:]


def patch_trio(s):
    """driver phases of env_rollout_trio_kernel (indices 0-9) and the two judges (12-16, 18-21)"""
    # npc_action gets an optional stamp cursor
    s = sub(s, "float g_far, float red_gap, float &acc,\n                        float &beta)\n{\n    const float gap = npc_gap<A>(cfg, ra, rb, i, bit_of_row<A>(i), ag, cp, sp, has_target, g_far);\n",
            "float g_far, float red_gap, float &acc,\n                        float &beta, unsigned long long *stl = nullptr)\n{\n    const float gap = npc_gap<A>(cfg, ra, rb, i, bit_of_row<A>(i), ag, cp, sp, has_target, g_far, stl);\n    tde_mark(stl, 2);\n")
    s = sub(s, "                      float cp, float sp, bool has_target, float g_far)\n{\n    using mask_t = typename MaskOf<A>::type;\n    constexpr int C = A < SB ? A : SB;\n    mask_t cand = 0;\n    const float hl_i = 0.5f * ag.len;\n    if (has_target) {",
            "                      float cp, float sp, bool has_target, float g_far, unsigned long long *stl = nullptr)\n{\n    using mask_t = typename MaskOf<A>::type;\n    constexpr int C = A < SB ? A : SB;\n    mask_t cand = 0;\n    const float hl_i = 0.5f * ag.len;\n    if (has_target) {")
    s = sub(s, "        cand &= ~own_bit;\n    }\n    float gap = 1e30f;\n",
            "        cand &= ~own_bit;\n    }\n    tde_mark(stl, 1);\n    float gap = 1e30f;\n")
    a, b = kernel_span(s, "env_rollout_trio_kernel")
    k = s[a:b]
    k = sub(k, "    if (threadIdx.x == 0) {\n        fill_cold(cold, cfg, w); sh.done = 0ull;",
            "    if (threadIdx.x < 24) s_acc[threadIdx.x] = 0ull;\n    if (threadIdx.x == 0) {\n        fill_cold(cold, cfg, w); sh.done = 0ull;")
    # ---- driver
    k = sub(k, "        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1, q = p ^ 1;\n            const float2 act = sh.act[p][base];",
            "        unsigned long long stl = __builtin_amdgcn_s_memtime();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1, q = p ^ 1;\n            const float2 act = sh.act[p][base];")
    k = sub(k, "                if (F & TDE_F_NPC) {\n                    // A second pass means that an env of this wavefront finished and its lanes were re-spawned: they are at the",
            "                tde_mark(&stl, 0);\n                if (F & TDE_F_NPC) {\n                    // A second pass means that an env of this wavefront finished and its lanes were re-spawned: they are at the")
    k = sub(k, "                        const float gap = npc_gap<A>(cfg, &sh.a[q][base], &sh.b[q][base], a, bit_of_row<A>(a), ag, c0, s0, has_target, cx.g_far);\n                        float red_gap = 1e30f;\n                        if (lights) {\n                            if (pass == 0) {",
            "                        const float gap = npc_gap<A>(cfg, &sh.a[q][base], &sh.b[q][base], a, bit_of_row<A>(a), ag, c0, s0, has_target, cx.g_far, &stl);\n                        tde_mark(&stl, 2);\n                        float red_gap = 1e30f;\n                        if (lights) {\n                            if (pass == 0) {")
    k = sub(k, "                nx = ag.x; ny = ag.y; npsi = ag.psi; nv = ag.v;\n", "                tde_mark(&stl, 3);\n                nx = ag.x; ny = ag.y; npsi = ag.psi; nv = ag.v;\n")
    k = sub(k, "                switched = false;\n                nwp = ag.route_wp;", "                tde_mark(&stl, 4);\n                switched = false;\n                nwp = ag.route_wp;")
    k = sub(k, "                sincos_f32(npsi, ns, nc);\n                TDE_PROBE(TDE_DUMMY_D, nx);\n                if (pass) break;\n                lds_barrier();                               // A: the judges' masks of step i-1 are published\n",
            "                sincos_f32(npsi, ns, nc);\n                tde_mark(&stl, 5);\n                if (pass) break;\n                lds_barrier();                               // A: the judges' masks of step i-1 are published\n                tde_mark(&stl, 6);\n")
    k = sub(k, "            write_rows(sh, p, lane, live, ag, c0, s0, cfg.npc_lane_half);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n            if (switched) load_route_target(cold, ag, cx);\n        }\n",
            "            write_rows(sh, p, lane, live, ag, c0, s0, cfg.npc_lane_half);\n            tde_mark(&stl, 7);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n            tde_mark(&stl, 8);\n            if (switched) load_route_target(cold, ag, cx);\n            tde_mark(&stl, 9);\n        }\n        tde_flush(0, 10);\n")
    # ---- judge C
    k = sub(k, "        if (lights) publish_red_gaps(0, 1, er.steps + 1);\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1, q = p ^ 1;\n            lds_barrier();                                   // A: masks of step i-1 are complete\n",
            "        if (lights) publish_red_gaps(0, 1, er.steps + 1);\n        unsigned long long stl = __builtin_amdgcn_s_memtime();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1, q = p ^ 1;\n            lds_barrier();                                   // A: masks of step i-1 are complete\n            tde_mark(&stl, 12);\n")
    k = sub(k, "            lds_barrier();                                   // B: rows of step i are in buffer p\n            er.steps += 1;\n            const int k = er.steps;\n            if (lights && i + 1 < ro.K) publish_red_gaps(i + 1, p, k + 1);      // the driver is computing step i + 1 from these rows now\n            const float4 ra = sh.a[p][lane], rb = sh.b[p][lane], rc = sh.c[p][lane];\n            if constexpr (A == 16 && TDE_COLLIDE_DPP)",
            "            tde_mark(&stl, 13);\n            lds_barrier();                                   // B: rows of step i are in buffer p\n            tde_mark(&stl, 14);\n            er.steps += 1;\n            const int k = er.steps;\n            if (lights && i + 1 < ro.K) publish_red_gaps(i + 1, p, k + 1);      // the driver is computing step i + 1 from these rows now\n            const float4 ra = sh.a[p][lane], rb = sh.b[p][lane], rc = sh.c[p][lane];\n            if constexpr (A == 16 && TDE_COLLIDE_DPP)")
    k = sub(k, "            if (lane == 0) sh.hit_mask = m;\n", "            if (lane == 0) sh.hit_mask = m;\n            tde_mark(&stl, 15);\n")
    k = sub(k, "                ro.reward[(int64_t)i * LB + e] = 0.0f;\n            }\n        }\n        lds_barrier();                                       // A'",
            "                ro.reward[(int64_t)i * LB + e] = 0.0f;\n            }\n            tde_mark(&stl, 16);\n        }\n        tde_flush(12, 17);\n        lds_barrier();                                       // A'")
    # ---- judge O
    k = sub(k, "            if (ego) act2 = acts[(int64_t)(i + 2 < ro.K ? i + 2 : ro.K - 1) * LB + e];   // in flight during this step\n            lds_barrier();                                   // A: masks of step i-1 are complete\n",
            "            if (ego) act2 = acts[(int64_t)(i + 2 < ro.K ? i + 2 : ro.K - 1) * LB + e];   // in flight during this step\n            lds_barrier();                                   // A: masks of step i-1 are complete\n            tde_mark(&stl, 18);\n")
    k = sub(k, "        lds_barrier();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1;\n            float2 act2",
            "        lds_barrier();\n        unsigned long long stl = __builtin_amdgcn_s_memtime();\n        for (int i = 0; i < ro.K; ++i) {\n            const int p = i & 1;\n            float2 act2")
    # (the done test moved behind barrier B: "done test + re-spawn" = slot 19 is what runs between B and the row reads)
    k = sub(k, "            lds_barrier();                                   // B: rows of step i are in buffer p\n            if (i > 0) {\n                // done(i-1) as the driver formed it",
            "            lds_barrier();                                   // B: rows of step i are in buffer p\n            tde_mark(&stl, 20);\n            if (i > 0) {\n                // done(i-1) as the driver formed it")
    k = sub(k, "            er.steps += 1;\n            const int k = er.steps;\n            const float4 ra = sh.a[p][lane], rb = sh.b[p][lane], rc = sh.c[p][lane];\n            const bool live = rc.z != 0.0f;\n            off = false;",
            "            tde_mark(&stl, 19);\n            er.steps += 1;\n            const int k = er.steps;\n            const float4 ra = sh.a[p][lane], rb = sh.b[p][lane], rc = sh.c[p][lane];\n            const bool live = rc.z != 0.0f;\n            off = false;")
    k = sub(k, "            if (ego) sh.act[p][lane] = act2;", "            if (ego) sh.act[p][lane] = act2;\n            tde_mark(&stl, 21);")
    k = sub(k, "        lds_barrier();                                       // A'\n        lds_barrier();                                       // done(K-1) is in sh.done\n        if (!valid) return;\n        st.offroad[g]",
            "        tde_flush(18, 22);\n        lds_barrier();                                       // A'\n        lds_barrier();                                       // done(K-1) is in sh.done\n        if (!valid) return;\n        st.offroad[g]")
    return s[:a] + k + s[b:]


def patch_step3(s):
    """milestones of env_step_trio_kernel per role (ticks between marks): drive 0-5, C 8-12, O 16-19"""
    a, b = kernel_span(s, "env_step_trio_kernel")
    k = s[a:b]
    k = sub(k, "    const int lane = threadIdx.x & (kWave - 1);\n    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n",
            "    const int lane = threadIdx.x & (kWave - 1);\n    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n    unsigned long long stl = __builtin_amdgcn_s_memtime();\n    if (threadIdx.x < 24) s_acc[threadIdx.x] = 0ull;\n    if (threadIdx.x == 0) g_wg[blockIdx.x & 4095][22] = __builtin_amdgcn_s_memrealtime();\n")
    # drive
    k = sub(k, "        float c0, s0;\n        const bool live = valid && ag.present;\n        const int k = er.steps + 1;",
            "        tde_mark(&stl, 0);\n        float c0, s0;\n        const bool live = valid && ag.present;\n        const int k = er.steps + 1;")
    k = sub(k, "        if (live) {\n            bicycle(ag.x, ag.y, ag.psi, ag.v, ag.inv_lr, acc, beta, cfg.dt);     // :117",
            "        tde_mark(&stl, 1);\n        if (live) {\n            bicycle(ag.x, ag.y, ag.psi, ag.v, ag.inv_lr, acc, beta, cfg.dt);     // :117")
    k = sub(k, "        write_rows(sh, 0, lane, live, ag, c0, s0, cfg.npc_lane_half);\n        lds_barrier();                                       // B: rows of this step are in buffer 0\n",
            "        write_rows(sh, 0, lane, live, ag, c0, s0, cfg.npc_lane_half);\n        tde_mark(&stl, 2);\n        lds_barrier();                                       // B: rows of this step are in buffer 0\n        tde_mark(&stl, 3);\n")
    k = sub(k, "        lds_barrier();                                       // A: the judges' masks are published\n",
            "        tde_mark(&stl, 6);\n        lds_barrier();                                       // A: the judges' masks are published\n        tde_mark(&stl, 4);\n")
    k = sub(k, "        if (respawned || switched || rebuilt || need_tg2) store_slot_cache(st, g, ag, er, cx, cfg.flags, respawned);\n",
            "        if (respawned || switched || rebuilt || need_tg2) store_slot_cache(st, g, ag, er, cx, cfg.flags, respawned);\n        tde_mark(&stl, 5);\n        tde_flush(0, 7);\n        if (lane == 0) { g_wg[blockIdx.x & 4095][23] = __builtin_amdgcn_s_memrealtime(); g_wg[blockIdx.x & 4095][7] = dn ? 1ull : 0ull; }\n")
    # judge C
    k = sub(k, "        lds_barrier();                                       // B\n        if (TDE_STEP_PRIO_SWITCH) __builtin_amdgcn_s_setprio(TDE_STEP_PRIO_C2);",
            "        tde_mark(&stl, 8);\n        lds_barrier();                                       // B\n        tde_mark(&stl, 9);\n        if (TDE_STEP_PRIO_SWITCH) __builtin_amdgcn_s_setprio(TDE_STEP_PRIO_C2);")
    k = sub(k, "        RewardOut rw{};\n        const int ti0 = er.target_idx;\n        bool reach = false;\n",
            "        tde_mark(&stl, 12);\n        RewardOut rw{};\n        const int ti0 = er.target_idx;\n        bool reach = false;\n")
    k = sub(k, "        lds_barrier();                                       // A: off / tl masks and the ego's psi term are in\n",
            "        tde_mark(&stl, 10);\n        lds_barrier();                                       // A: off / tl masks and the ego's psi term are in\n        tde_mark(&stl, 11);\n")
    # judge O
    k = sub(k, "        }\n        lds_barrier();                                       // B\n        if (TDE_STEP_PRIO_SWITCH) __builtin_amdgcn_s_setprio(TDE_STEP_PRIO_O2);\n",
            "        }\n        tde_mark(&stl, 16);\n        lds_barrier();                                       // B\n        tde_mark(&stl, 17);\n        if (TDE_STEP_PRIO_SWITCH) __builtin_amdgcn_s_setprio(TDE_STEP_PRIO_O2);\n")
    k = sub(k, "        if (lane == 0) { sh.off_mask = om; sh.tl_mask = tm; }\n",
            "        if (lane == 0) { sh.off_mask = om; sh.tl_mask = tm; }\n        tde_mark(&stl, 18);\n")
    # (round 5: the magnitudes section - slot 21 = from barrier A to its end; slot 20 now includes the pre-A touches)
    k = sub(k, "        lds_barrier();                                       // A\n        unsigned long long term_m, trunc_m;\n        const unsigned long long dn = done_of(k, term_m, trunc_m);\n        if (valid) {\n            st.offroad[g]",
            "        tde_mark(&stl, 20);\n        lds_barrier();                                       // A\n        tde_mark(&stl, 19);\n        unsigned long long term_m, trunc_m;\n        const unsigned long long dn = done_of(k, term_m, trunc_m);\n        if (valid) {\n            st.offroad[g]")
    k = sub(k, "                if (lane == src) out_e[0] = omag;\n            }\n#endif\n        }\n    }\n}\n",
            "                if (lane == src) out_e[0] = omag;\n            }\n#endif\n        }\n        tde_mark(&stl, 21);\n        tde_flush(16, 22);\n    }\n}\n")
    k = sub(k, "        if (!valid) return;\n        st.collided[g] = respawned ? 0 : (hit ? 1 : 0);", "        tde_flush(8, 13);\n        if (!valid) return;\n        st.collided[g] = respawned ? 0 : (hit ? 1 : 0);")
    return s[:a] + k + s[b:]


def patch_trips(s):
    """exact-loop trip counts: controller (g_stamps 0 trips, 1 calls, 3 candidates) and collision sweep (4, 5, 6)"""
    s = sub(s, "    // taken (fj = 0).\n    while (__ballot(cand != 0)) {\n",
            "    // taken (fj = 0).\n    if (threadIdx.x % 64 == 0) atomicAdd(&g_stamps[1], 1ull);\n"
            "    atomicAdd(&g_stamps[3], (unsigned long long)__popcll((unsigned long long)cand));\n"
            "    while (__ballot(cand != 0)) {\n        if (threadIdx.x % 64 == 0) atomicAdd(&g_stamps[0], 1ull);\n")
    s = sub(s, "    bool hit = false;\n    while (__ballot(cand != 0)) {\n        if (cand) {\n            const int j = row_of_bit<A>(lowest_bit(cand));\n",
            "    bool hit = false;\n    if (threadIdx.x % 64 == 0) atomicAdd(&g_stamps[5], 1ull);\n"
            "    atomicAdd(&g_stamps[6], (unsigned long long)__popcll((unsigned long long)cand));\n"
            "    while (__ballot(cand != 0)) {\n        if (threadIdx.x % 64 == 0) atomicAdd(&g_stamps[4], 1ull);\n        if (cand) {\n            const int j = row_of_bit<A>(lowest_bit(cand));\n")
    return s


def patch_wide(s):
    """milestones of env_step_wide_kernel (128 slots, two roles), wavefronts 0 (drive) and 2 (judge) only: drive 0-7, judge 8-15"""
    a, b = kernel_span(s, "env_step_wide_kernel")
    k = s[a:b]
    k = sub(k, "    const int a = ((wv & 1) << 6) | lane;                   // the lane's slot\n",
            "    const int a = ((wv & 1) << 6) | lane;                   // the lane's slot\n    unsigned long long stl = __builtin_amdgcn_s_memtime();\n"
            "    unsigned long long *sp = (wv & 1) ? nullptr : &stl;\n    if (threadIdx.x < 24) s_acc[threadIdx.x] = 0ull;\n"
            "    if (threadIdx.x == 0) g_wg[blockIdx.x & 4095][20] = __builtin_amdgcn_s_memrealtime();\n")
    k = sub(k, "        if (st.slot_cache) { sc0 = reinterpret_cast<const int4 *>(st.slot_cache + g)[0]; sc1 = reinterpret_cast<const int4 *>(st.slot_cache + g)[1]; }\n",
            "        if (st.slot_cache) { sc0 = reinterpret_cast<const int4 *>(st.slot_cache + g)[0]; sc1 = reinterpret_cast<const int4 *>(st.slot_cache + g)[1]; }\n        tde_mark(sp, 17);\n")
    # drive
    k = sub(k, "        TDE_WIDE_COLD_BARRIER();                             // cold is published\n        Ctx cx;\n        bool rebuilt;\n",
            "        TDE_WIDE_COLD_BARRIER();                             // cold is published\n        tde_mark(sp, 0);\n        Ctx cx;\n        bool rebuilt;\n")
    k = sub(k, "        lds_barrier();                                       // E: does a drive wavefront lack stored actions?\n",
            "        tde_mark(sp, 1);\n        lds_barrier();                                       // E: does a drive wavefront lack stored actions?\n        tde_mark(sp, 2);\n")
    k = sub(k, "        lds_barrier();                                       // B: rows of this step are in buffer 0\n        __builtin_amdgcn_s_setprio(NW == 8 ? 2 : TDE_WIDE_PRIO_D2);",
            "        tde_mark(sp, 3);\n        lds_barrier();                                       // B: rows of this step are in buffer 0\n        tde_mark(sp, 4);\n        __builtin_amdgcn_s_setprio(NW == 8 ? 2 : TDE_WIDE_PRIO_D2);")
    k = sub(k, "        lds_barrier();                                       // A: the env's done flag is published\n",
            "        tde_mark(sp, 5);\n        lds_barrier();                                       // A: the env's done flag is published\n        tde_mark(sp, 6);\n")
    k = sub(k, "            if (a == 0) reinterpret_cast<int2 *>(ap)[A] = make_int2(((F & TDE_F_NPC) && !respawned) ? er.episode : -1, act_key_steps(act_hash, er.steps));\n        }\n",
            "            if (a == 0) reinterpret_cast<int2 *>(ap)[A] = make_int2(((F & TDE_F_NPC) && !respawned) ? er.episode : -1, act_key_steps(act_hash, er.steps));\n        }\n"
            "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        tde_mark(sp, 7);\n        if (sp) { tde_flush(0, 8); tde_flush(17, 18); }\n        if (wv == 0 && lane == 0) g_wg[blockIdx.x & 4095][21] = __builtin_amdgcn_s_memrealtime();\n")
    # judge
    k = sub(k, "        lds_barrier();                                       // E\n        if (sh.early[0] | sh.early[1]) lds_barrier();        // E2\n",
            "        tde_mark(sp, 8);\n        lds_barrier();                                       // E\n        if (sh.early[0] | sh.early[1]) lds_barrier();        // E2\n")
    k = sub(k, "        lds_barrier();                                       // B: rows of this step are in buffer 0\n        __builtin_amdgcn_s_setprio(NW == 8 ? 2 : TDE_WIDE_PRIO_J2);\n",
            "        lds_barrier();                                       // B: rows of this step are in buffer 0\n        tde_mark(sp, 9);\n        __builtin_amdgcn_s_setprio(NW == 8 ? 2 : TDE_WIDE_PRIO_J2);\n")
    k = sub(k, "        wide_sym_publish(sh, wv & 1, lane, 1);\n", "        wide_sym_publish(sh, wv & 1, lane, 1);\n        tde_mark(sp, 10);\n")
    k = sub(k, "        hit |= wide_sym_joined(sh, wv & 1, a, 1);\n", "        tde_mark(sp, 11);\n        hit |= wide_sym_joined(sh, wv & 1, a, 1);\n        tde_mark(sp, 12);\n")
    k = sub(k, "        lds_barrier();                                       // A\n        const bool respawned = sh.done != 0;\n        st.collided[g]",
            "        tde_mark(sp, 13);\n        lds_barrier();                                       // A\n        tde_mark(sp, 14);\n        const bool respawned = sh.done != 0;\n        st.collided[g]")
    k = sub(k, "        if constexpr (MAG) {\n            // tde_state.magnitudes for a flagged ego (get_info's \"collision\" / \"offroad\", :427-428), by the ego's wavefront from the\n",
            "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        tde_mark(sp, 15);\n        if (sp) tde_flush(8, 17);\n        if (wv == 2 && lane == 0) g_wg[blockIdx.x & 4095][22] = __builtin_amdgcn_s_memrealtime();\n        if constexpr (MAG) {\n            // tde_state.magnitudes for a flagged ego (get_info's \"collision\" / \"offroad\", :427-428), by the ego's wavefront from the\n")
    return s[:a] + k + s[b:]


PATCHES = {"duo": patch_duo, "trio": patch_trio, "step3": patch_step3, "trips": patch_trips, "wide": patch_wide}


def patched_source(mode):
    """the kernel source with the stamps of `mode` applied (asserts that every anchor still exists:
    tests/test_boundary.py::test_stamp_patches_apply_to_the_current_source)"""
    s = open(SRC).read()
    s = sub(s, ANCHOR, PRELUDE + ANCHOR)
    # the patched device code first (it defines the header's include guard, so the units' own #include of tde_kernels.h is a
    # no-op), then every unit of the library as csrc/tde_kernels.hip lists them, then the debug entry point
    units = [ln for ln in open(SRC[:-2] + ".hip").read().splitlines() if ln.startswith('#include "tde_') and ln.endswith('.hip"')]
    assert len(units) >= 5, "csrc/tde_kernels.hip no longer lists the library's units"
    return "#define TDE_TU_API 1\n" + PATCHES[mode](s) + "\n".join(units) + "\n" + EPILOGUE


def main():
    """python scripts/make_stamped_build.py MODE [OUT.so] [-DNAME=VAL ...]"""
    mode = sys.argv[1] if len(sys.argv) > 1 else "duo"
    rest = sys.argv[2:]
    extra = [a for a in rest if a.startswith("-")]
    outs = [a for a in rest if not a.startswith("-")]
    s = patched_source(mode)
    os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
    out = os.path.abspath(outs[0]) if outs else os.path.join(ROOT, "ab", "libS.so")
    tmp = os.path.join(ROOT, "ab", f"tde_kernels_{mode}_stamped_{os.path.basename(out)[:-3]}.hip")
    open(tmp, "w").write(s)
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-I" + os.path.join(ROOT, "torchdriveenv_amd", "csrc"),
                                                      "-I" + os.path.join(ROOT, "include"), "-o", out, tmp]
    subprocess.run(cmd, check=True)
    print(out)


if __name__ == "__main__":
    main()
