// How long does the dispatcher take to START every workgroup of a launch that fills the chip once, and what does that depend on?
// Each workgroup records s_memrealtime (100 MHz) at entry, then spins ~6 us (every workgroup of the launch is resident at once, as in the
// one-step kernels) and leaves.  Reported: the spread of the start times (last - first), per shape.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/dispatch_ramp scripts/ubench/dispatch_ramp.hip && scripts/ubench/dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int WAVES_PER_EU>
__global__ __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void ramp_kernel(unsigned long long *start, unsigned long long *stop, int spin_ticks)
{
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = (char)t0; }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) stop[blockIdx.x] = __builtin_amdgcn_s_memrealtime() + (lds[0] & 0);
}

template <int LDS_BYTES, int WPE>
static void run(const char *what, int blocks, int threads)
{
    unsigned long long *d0, *d1;
    hipMalloc(&d0, sizeof(unsigned long long) * blocks);
    hipMalloc(&d1, sizeof(unsigned long long) * blocks);
    std::vector<unsigned long long> h0(blocks), h1(blocks);
    double ramp = 0, p50 = 0, p99 = 0, total = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        ramp_kernel<LDS_BYTES, WPE><<<blocks, threads>>>(d0, d1, 600);
        ramp_kernel<LDS_BYTES, WPE><<<blocks, threads>>>(d0, d1, 600);      // back to back: the second launch is the one measured
        hipDeviceSynchronize();
        hipMemcpy(h0.data(), d0, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), d1, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        if (r < 3) continue;
        std::vector<unsigned long long> s = h0;
        std::sort(s.begin(), s.end());
        ramp += (s.back() - s.front()) * 0.01;
        p50 += (s[blocks / 2] - s.front()) * 0.01;
        p99 += (s[blocks * 99 / 100] - s.front()) * 0.01;
        total += (*std::max_element(h1.begin(), h1.end()) - s.front()) * 0.01;
    }
    printf("%-64s %5d workgroups x %4d threads: starts spread over %5.2f us (median start %5.2f, 99 %% %5.2f); first start -> last end %5.2f us\n", what, blocks,
           threads, ramp / reps, p50 / reps, p99 / reps, total / reps);
    hipFree(d0); hipFree(d1);
}

int main()
{
    run<17640, 6>("the one-step kernel's shape (3 wavefronts, 17.6 KB LDS, 80 VGPRs)", 2048, 192);
    run<35280, 6>("two groups per workgroup (6 wavefronts, 35 KB LDS)", 1024, 384);
    run<70560, 6>("four groups per workgroup (12 wavefronts, 70 KB LDS)", 512, 768);
    run<0, 6>("3 wavefronts, no LDS", 2048, 192);
    run<17640, 8>("3 wavefronts, 17.6 KB LDS, 64 VGPRs", 2048, 192);
    run<17640, 4>("3 wavefronts, 17.6 KB LDS, 128 VGPRs (4 per SIMD: 2 rounds)", 1365, 192);
    run<17328, 4>("the 128-slot step kernel's shape (4 wavefronts, 17.3 KB LDS)", 1024, 256);
    run<17328, 4>("... eight wavefronts", 512, 512);
    run<0, 8>("one wavefront per workgroup, no LDS", 6144, 64);
    return 0;
}
