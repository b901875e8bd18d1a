// Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (see MI355X_MICROARCH.md, HBM):
// every kernel streams a 1 GiB buffer (4x the 256 MiB Infinity Cache) once, with loads of 4, 8 or 16 bytes per lane
// (coalesced, one wave reads 256 / 512 / 1024 contiguous bytes per instruction), or stores of the same widths.
// Build: hipcc -O3 --offload-arch=gfx950 -o pmc_calibrate tools/pmc/pmc_calibrate.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- ./pmc_calibrate     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T>
__global__ void __launch_bounds__(256) k_read(const T* __restrict__ src, size_t n, unsigned long long* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) {
        const T v = src[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += w[k];
    }
    if (acc == 0x1234567887654321ull) *sink = acc;      // never true: keeps the loads
}
template <typename T>
__global__ void __launch_bounds__(256) k_write(T* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v; unsigned* w = reinterpret_cast<unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = (unsigned)i + k;
    for (; i < n; i += stride) dst[i] = v;
}
// byte-wide per-lane gathers through a 64-byte record per lane (one lane = one record): the stop-node extras of dp_wave.hip
__global__ void __launch_bounds__(256) k_read_rec64(const uint4* __restrict__ src, size_t nrec, int every, unsigned long long* __restrict__ sink) {
    size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * every;
    const size_t stride = (size_t)gridDim.x * blockDim.x * every;
    unsigned long long acc = 0;
    for (; r < nrec; r += stride) {
        const uint4 a = src[4 * r], b = src[4 * r + 1], c = src[4 * r + 2], d = src[4 * r + 3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf; unsigned long long* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const dim3 grid(256 * 16), blk(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read<uint32_t>, grid, blk, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_read<uint2>, grid, blk, 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_read<uint4>, grid, blk, 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_read_rec64, grid, blk, 0, 0, (const uint4*)buf, bytes / 64, 1, sink);      // every record: 1 GiB
        hipLaunchKernelGGL(k_read_rec64, grid, blk, 0, 0, (const uint4*)buf, bytes / 64, 6, sink);      // every 6th record: 1/6 GiB asked for
        hipLaunchKernelGGL(k_write<uint32_t>, grid, blk, 0, 0, (uint32_t*)buf, bytes / 4);
        hipLaunchKernelGGL(k_write<uint2>, grid, blk, 0, 0, (uint2*)buf, bytes / 8);
        hipLaunchKernelGGL(k_write<uint4>, grid, blk, 0, 0, (uint4*)buf, bytes / 16);
        hipDeviceSynchronize();
    }
    printf("pmc_calibrate: every kernel moved %zu bytes (k_read_rec64 with every=6: %zu bytes requested)\n", bytes, bytes / 6);
    hipFree(buf); hipFree(sink);
    return 0;
}
