// Micro-test: does a 64-bit VALU shift give the right answer when its shift AMOUNT sits in the wavefront's last allocated VGPR?
// (LLVM's GCNHazardRecognizer::fixShift64HighRegBug works around such an erratum for gfx11; round 5 met the symptom on gfx950:
// profiles/r05_a32_respawn_anomaly.md.)  Each kernel is compiled to a fixed number of VGPRs (8 k, the allocation granule is 8) and does
//     r = m >> amt      with the amount moved by hand into v(8k-1) [LAST] or v(8k-2) [control], the instruction written in asm.
// Other wavefronts of a differently sized kernel run beside it so that the register after the allocation holds foreign data.
//   hipcc --offload-arch=gfx950 -O3 -o shift64_last_vgpr shift64_last_vgpr.hip && ./shift64_last_vgpr
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

// NV = the kernel's VGPR budget (a multiple of 8).  The clobber list forces the allocation up to v(NV-1); nothing else may exceed it.
#define SHIFT_KERNEL_OP(NAME, OP, CEXPR, NV, LASTREG, CTRLREG)                                                                                          \
    __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(NV))) void NAME(const uint64_t *m, const int *amt,        \
                                                                                           uint64_t *r_last, uint64_t *r_ctrl,     \
                                                                                           uint64_t *r_c, int rounds)                \
    {                                                                                                                              \
        const int i = blockIdx.x * 64 + threadIdx.x;                                                                               \
        const uint64_t mv = m[i];                                                                                                  \
        const int a = amt[i];                                                                                                      \
        uint64_t x = 0, y = 0;                                                                                                     \
        for (int r = 0; r < rounds; ++r) {                                                                                         \
            uint64_t t0, t1;                                                                                                       \
            asm volatile("v_mov_b32 " LASTREG ", %1\n\t" OP " %0, " LASTREG ", %2" : "=&v"(t0) : "v"(a), "v"(mv) : LASTREG);   \
            asm volatile("v_mov_b32 " CTRLREG ", %1\n\t" OP " %0, " CTRLREG ", %2" : "=&v"(t1) : "v"(a), "v"(mv) : CTRLREG);   \
            x |= t0 ^ (uint64_t)r * 0;                                                                                             \
            y |= t1;                                                                                                               \
        }                                                                                                                          \
        r_last[i] = x;                                                                                                             \
        r_ctrl[i] = y;                                                                                                             \
        r_c[i] = CEXPR;                                                                                                            \
    }

#define SHIFT_KERNEL(NV, LASTREG, CTRLREG) SHIFT_KERNEL_OP(k_shift_##NV, "v_lshrrev_b64", mv >> a, NV, LASTREG, CTRLREG)
SHIFT_KERNEL(8, "v7", "v6")
SHIFT_KERNEL(16, "v15", "v14")
SHIFT_KERNEL(80, "v79", "v78")
SHIFT_KERNEL_OP(k_shl_16, "v_lshlrev_b64", mv << a, 16, "v15", "v14")
SHIFT_KERNEL_OP(k_sar_16, "v_ashrrev_i64", (uint64_t)((int64_t)mv >> a), 16, "v15", "v14")
// the 32-bit shift, for comparison (no erratum expected): the low dword only
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(16))) void k_shr32_16(const uint64_t *m, const int *amt, uint64_t *r_last,
                                                                                      uint64_t *r_ctrl, uint64_t *r_c, int rounds)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const uint32_t mv = (uint32_t)m[i];
    const int a = amt[i];
    uint32_t x = 0, y = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t t0, t1;
        asm volatile("v_mov_b32 v15, %1\n\tv_lshrrev_b32 %0, v15, %2" : "=&v"(t0) : "v"(a), "v"(mv) : "v15");
        asm volatile("v_mov_b32 v14, %1\n\tv_lshrrev_b32 %0, v14, %2" : "=&v"(t1) : "v"(a), "v"(mv) : "v14");
        x |= t0;
        y |= t1;
    }
    r_last[i] = x;
    r_ctrl[i] = y;
    r_c[i] = mv >> (a & 31);
}

// Scope of the erratum (round 6): the same 64-bit shift with the amount NOT in a VGPR - a scalar register (a wave-uniform amount)
// and an inline literal - while the 64-bit VALUE sits in the wavefront's last register pair v[14:15] of 16, and the value in the last
// pair with the amount in a low VGPR.  Expected: fine (the hazard LLVM names for gfx11 is about the register AFTER the amount).
#define SHIFT_KERNEL_SCOPE(NAME, AMT_OPERAND, AMT_CONSTRAINT, AMT_EXPR, CEXPR)                                                      \
    __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(16))) void NAME(const uint64_t *m, const int *amt, uint64_t *r_last,   \
                                                                                    uint64_t *r_ctrl, uint64_t *r_c, int rounds)            \
    {                                                                                                                              \
        const int i = blockIdx.x * 64 + threadIdx.x;                                                                               \
        const uint64_t mv = m[i];                                                                                                  \
        const int a = AMT_EXPR;                                                                                                    \
        uint64_t x = 0, y = 0;                                                                                                     \
        for (int r = 0; r < rounds; ++r) {                                                                                         \
            uint64_t t0, t1;                                                                                                       \
            asm volatile("v_mov_b32 v14, %2\n\tv_mov_b32 v15, %3\n\tv_lshrrev_b64 %0, " AMT_OPERAND ", v[14:15]"                     \
                         : "=&v"(t0) : AMT_CONSTRAINT(a), "v"((uint32_t)mv), "v"((uint32_t)(mv >> 32)) : "v14", "v15");            \
            asm volatile("v_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\tv_lshrrev_b64 %0, " AMT_OPERAND ", v[12:13]"                     \
                         : "=&v"(t1) : AMT_CONSTRAINT(a), "v"((uint32_t)mv), "v"((uint32_t)(mv >> 32)) : "v12", "v13");            \
            x |= t0;                                                                                                               \
            y |= t1;                                                                                                               \
        }                                                                                                                          \
        r_last[i] = x;                                                                                                             \
        r_ctrl[i] = y;                                                                                                             \
        r_c[i] = CEXPR;                                                                                                            \
    }
SHIFT_KERNEL_SCOPE(k_shr_sgpr_16, "%1", "s", __builtin_amdgcn_readfirstlane(amt[blockIdx.x * 64]) & 63, mv >> a)
SHIFT_KERNEL_SCOPE(k_shr_lit_16, "20", "s", 20, mv >> 20)
SHIFT_KERNEL_SCOPE(k_shr_lowv_16, "%1", "v", amt[i], mv >> a)

// the neighbours: wavefronts with a different allocation whose registers hold all-ones patterns for a while
__global__ __launch_bounds__(64) void k_noise(uint32_t *out, int spin)
{
    uint32_t v[24];
    for (int k = 0; k < 24; ++k) v[k] = 0xFFFFFFF0u + k + threadIdx.x;
    for (int s = 0; s < spin; ++s)
        for (int k = 0; k < 24; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(v[(k + 1) % 24] | 0x3Fu));
    uint32_t acc = 0;
    for (int k = 0; k < 24; ++k) acc ^= v[k];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <typename K> static int run(const char *name, K kernel, int nblk, bool mask32 = false)
{
    const int n = nblk * 64;
    uint64_t *m, *r0, *r1, *rc;
    int *amt;
    uint32_t *noise;
    CHECK(hipMallocManaged(&m, n * 8)); CHECK(hipMallocManaged(&r0, n * 8)); CHECK(hipMallocManaged(&r1, n * 8));
    CHECK(hipMallocManaged(&rc, n * 8)); CHECK(hipMallocManaged(&amt, n * 4)); CHECK(hipMalloc(&noise, 4096 * 64 * 4));
    srand(1);
    for (int i = 0; i < n; ++i) {
        m[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ (uint64_t)rand();
        amt[i] = (i & 1) ? 32 : (rand() & 63);
    }
    hipStream_t s0, s1;
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    long bad_last = 0, bad_ctrl = 0, total = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(k_noise, dim3(4096), dim3(64), 0, s1, noise, 200);
        hipLaunchKernelGGL(kernel, dim3(nblk), dim3(64), 0, s0, m, amt, r0, r1, rc, 50);
        CHECK(hipDeviceSynchronize());
        for (int i = 0; i < n; ++i) {
            uint64_t want = rc[i];                       // (the same operation as the compiler emits it)
            if (mask32) want &= 0xFFFFFFFFull;
            const uint64_t g0 = mask32 ? (r0[i] & 0xFFFFFFFFull) : r0[i], g1 = mask32 ? (r1[i] & 0xFFFFFFFFull) : r1[i];
            bad_last += g0 != want;
            bad_ctrl += g1 != want;
            if (g0 != want && bad_last <= 3)
                printf("  %s: lane %d amt %d m %016llx: got %016llx want %016llx\n", name, i, amt[i], (unsigned long long)m[i], (unsigned long long)r0[i], (unsigned long long)want);
        }
        total += n;
    }
    printf("%-10s amount in the LAST allocated VGPR: %ld of %ld wrong; in the one before it: %ld wrong\n", name, bad_last, total, bad_ctrl);
    return bad_last != 0;
}

int main()
{
    int rc = 0;
    rc |= run("8 VGPRs", k_shift_8, 8192);
    rc |= run("16 VGPRs", k_shift_16, 8192);
    rc |= run("80 VGPRs", k_shift_80, 8192);
    rc |= run("shl64/16", k_shl_16, 8192);
    rc |= run("sar64/16", k_sar_16, 8192);
    // (not part of the verdict: the amount is not in the last VGPR in these three)
    const int rs = run("sgpr amt", k_shr_sgpr_16, 8192), rl = run("literal", k_shr_lit_16, 8192), rv = run("value last", k_shr_lowv_16, 8192);
    printf("amount in an SGPR / an inline literal / a low VGPR, the VALUE in the last register pair: %s / %s / %s\n", rs ? "WRONG" : "fine",
           rl ? "WRONG" : "fine", rv ? "WRONG" : "fine");
    const int r32 = run("shr32/16", k_shr32_16, 8192, true);
    printf("32-bit shift with its amount in the last VGPR: %s\n", r32 ? "WRONG too" : "fine");
    printf(rc ? "ERRATUM REPRODUCED\n" : "not reproduced by this test\n");
    return 0;
}
