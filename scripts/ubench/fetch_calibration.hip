// FETCH_SIZE calibration on gfx950 for the access patterns of the one-step kernels (MI355X_MICROARCH.md: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"): every kernel reads each byte of a 512 MiB buffer
// exactly once; run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calibration  and compare KiB counted with KiB read.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fetch_calibration scripts/ubench/fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __device__ unsigned fold(T v);
template <> __device__ unsigned fold<uint8_t>(uint8_t v) { return v; }
template <> __device__ unsigned fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ unsigned fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ unsigned fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// coalesced: lane l of a wavefront reads element (wave's base + l): sizeof(T) bytes per lane, contiguous over the wavefront
template <typename T> __global__ void read_coalesced(const T *p, size_t n, unsigned *sink)
{
    unsigned acc = 0x9e377900u;             // (bytes only reach the low 8 bits: start where the sink's test stays undecidable)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= fold(p[i]);
    if (acc == 0x9e3779b9u) *sink = acc;
}
// the first 16 bytes of every 32-byte entry (a slot-cache word per lane): every 128-byte line is touched, half of its bytes are used
__global__ void read_first_half_of_32(const uint4 *p, size_t n_entries, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_entries; i += (size_t)gridDim.x * blockDim.x) acc ^= fold(p[2 * i]);
    if (acc == 0x9e3779b9u) *sink = acc;
}
// both 16-byte words of every 32-byte entry, as two loads per lane
__global__ void read_both_halves_of_32(const uint4 *p, size_t n_entries, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_entries; i += (size_t)gridDim.x * blockDim.x)
        acc ^= fold(p[2 * i]) ^ fold(p[2 * i + 1]);
    if (acc == 0x9e3779b9u) *sink = acc;
}
// struct-of-arrays at 16 slots per env: 4 bytes per lane from EIGHT arrays in turn (the state's x, y, psi, v, len, wid, lr, vdes)
__global__ void read_soa8(const uint32_t *p, size_t n_per_array, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_array; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 8; ++a) acc ^= p[(size_t)a * n_per_array + i];
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

// sparse: 4 bytes per lane every STRIDE bytes (a scattered table look-up: one dword of a line, or of a 64-byte half line)
template <int STRIDE> __global__ void read_sparse(const uint8_t *p, size_t bytes, unsigned *sink)
{
    unsigned acc = 0;
    const size_t n = bytes / STRIDE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc ^= *reinterpret_cast<const uint32_t *>(p + i * STRIDE);
    if (acc == 0x9e3779b9u) *sink = acc;
}

int main()
{
    const size_t bytes = 512ull << 20;
    void *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess || hipMemset(buf, 1, bytes) != hipSuccess) return 1;
    (void)hipDeviceSynchronize();
    const int blocks = 256 * 8, threads = 256;
    for (int rep = 0; rep < 3; ++rep) {
        read_coalesced<uint4><<<blocks, threads>>>((const uint4 *)buf, bytes / 16, sink);
        read_coalesced<uint2><<<blocks, threads>>>((const uint2 *)buf, bytes / 8, sink);
        read_coalesced<uint32_t><<<blocks, threads>>>((const uint32_t *)buf, bytes / 4, sink);
        read_coalesced<uint8_t><<<blocks, threads>>>((const uint8_t *)buf, bytes, sink);
        read_first_half_of_32<<<blocks, threads>>>((const uint4 *)buf, bytes / 32, sink);
        read_both_halves_of_32<<<blocks, threads>>>((const uint4 *)buf, bytes / 32, sink);
        read_soa8<<<blocks, threads>>>((const uint32_t *)buf, bytes / 32, sink);
        read_sparse<32><<<blocks, threads>>>((const uint8_t *)buf, bytes, sink);
        read_sparse<64><<<blocks, threads>>>((const uint8_t *)buf, bytes, sink);
        read_sparse<128><<<blocks, threads>>>((const uint8_t *)buf, bytes, sink);
        read_sparse<256><<<blocks, threads>>>((const uint8_t *)buf, bytes, sink);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
    }
    printf("ok: every kernel read %zu KiB (read_first_half_of_32: %zu KiB used, %zu KiB of lines touched)\n", bytes >> 10, bytes >> 11, bytes >> 10);
    return 0;
}
