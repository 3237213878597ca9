// What rocprofv3's FETCH_SIZE reports for reads of known size on this GPU: a streaming read (16 B per lane,
// coalesced), and scattered reads of 16 B and of 64 B from pseudo-randomly chosen distinct 128-byte lines of a
// 1 GiB buffer (far more than the 256 MiB Infinity Cache holds).  Measurement tool, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 -o tools/calib/fetch_calib tools/calib/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- tools/calib/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ void stream16(const u4* __restrict__ buf, int64_t n16, unsigned* sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { const u4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// line j of the walk = (j * 2654435761) mod n_lines, n_lines a power of two: a permutation of the lines
template <int QUADS>
__global__ void scatter(const u4* __restrict__ buf, int64_t n_lines, unsigned* sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += stride) {
        const int64_t line = (j * 2654435761LL) & (n_lines - 1);
        for (int q = 0; q < QUADS; ++q) { const u4 v = buf[line * 8 + q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int64_t bytes = 1LL << 30, n_lines = bytes / 128;
    u4* buf; unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, buf, bytes / 16, sink);
        hipLaunchKernelGGL(scatter<1>, dim3(8192), dim3(256), 0, 0, buf, n_lines, sink);
        hipLaunchKernelGGL(scatter<4>, dim3(8192), dim3(256), 0, 0, buf, n_lines, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    printf("stream16: %lld bytes read; scatter<1>: %lld lines x 16 B = %lld bytes asked for; scatter<4>: %lld lines x 64 B = %lld bytes\n",
           (long long)bytes, (long long)n_lines, (long long)n_lines * 16, (long long)n_lines, (long long)n_lines * 64);
    return 0;
}
