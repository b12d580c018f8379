// Micro-benchmark: issue cost of dependent vs independent fp32 VALU chains for ONE wavefront on a SIMD (gfx950),
// measured with s_memtime.  hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip && ./valu_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N 1024

template <int CHAINS>
__global__ void k_fma(float *out, uint64_t *ticks, float a, float b)
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 0.001f + c;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
__global__ void k_mul_add(float *out, uint64_t *ticks, float a, float b)   // v_mul (VOP2) + v_add (VOP2), dependent
{
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 0.001f + c;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ void k_cmp_sel(float *out, uint64_t *ticks, float a)   // v_cmp -> v_cndmask dependent pairs (VCC hazard)
{
    float x = threadIdx.x * 0.001f;
    uint32_t m = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a), "v"(x) : "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + m;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ void k_lds(float *out, uint64_t *ticks, int stride)   // dependent ds_read_b128 chain (pointer chase)
{
    __shared__ float4 tile[256];
    for (int i = threadIdx.x; i < 256; i += 64) tile[i] = make_float4(__int_as_float((i + stride) & 255), 0, 0, 0);
    __syncthreads();
    int idx = threadIdx.x & 15;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) idx = __float_as_int(tile[idx].x);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ void k_empty(float *out, uint64_t *ticks)
{
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <typename F> static void run(const char *name, int blocks, int ops, F launch)
{
    float *out; uint64_t *ticks;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMalloc(&ticks, blocks * sizeof(uint64_t));
    launch(out, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(out, ticks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t *h = (uint64_t *)malloc(blocks * sizeof(uint64_t));
    hipMemcpy(h, ticks, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += h[i];
    avg /= blocks;
    printf("%-44s blocks %6d  ticks %9.0f  ticks/op %6.2f   (kernel %.1f us)\n", name, blocks, avg, ops ? avg / ops : avg, ms * 1e3);
    free(h); hipFree(out); hipFree(ticks);
}

int main()
{
    for (int blocks : {256, 1024, 2048, 4096, 8192}) {
        printf("---- %d one-wave workgroups (1024 SIMDs)\n", blocks);
        run("memtime pair", blocks, 0, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_empty, blocks, 64, 0, 0, o, t); });
        run("v_fma_f32 dependent chain x1", blocks, N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_fma<1>, blocks, 64, 0, 0, o, t, 1.0001f, 0.5f); });
        run("v_fma_f32 2 independent chains", blocks, 2 * N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_fma<2>, blocks, 64, 0, 0, o, t, 1.0001f, 0.5f); });
        run("v_fma_f32 4 independent chains", blocks, 4 * N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_fma<4>, blocks, 64, 0, 0, o, t, 1.0001f, 0.5f); });
        run("v_mul+v_add dependent x1", blocks, N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_mul_add<1>, blocks, 64, 0, 0, o, t, 1.0001f, 0.5f); });
        run("v_mul+v_add 4 independent chains", blocks, 4 * N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_mul_add<4>, blocks, 64, 0, 0, o, t, 1.0001f, 0.5f); });
        run("v_cmp;s_nop 1;v_cndmask dependent (per pair)", blocks, N / 2, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_cmp_sel, blocks, 64, 0, 0, o, t, 0.5f); });
        run("ds_read_b128 dependent chain", blocks, N, [&](float *o, uint64_t *t) { hipLaunchKernelGGL(k_lds, blocks, 64, 0, 0, o, t, 1); });
    }
    return 0;
}
