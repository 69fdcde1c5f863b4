// fetch_calib.hip — what rocprofv3's FETCH_SIZE reports for the read patterns of this library's kernels, each pattern
// reading every byte of a buffer far larger than the Infinity Cache EXACTLY ONCE (MI355X_MICROARCH.md, HBM: "FETCH_SIZE
// reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  One kernel name per pattern, so that the per-kernel counter rows
// can be compared with the bytes printed here:
//   wide16      16 bytes per lane, lanes contiguous (the convolutions' loaders, the max pools' rows)
//   planes16    the three-plane average pool's reads (csrc/pool.hip avgpool3x3s1_p3x8): thread t takes the 16-byte halves
//               (t & 1) of the three 32-byte plane rows of 16-channel group t >> 1 — a wave covers 3 KiB contiguous bytes
//               with three loads of 16 bytes at a 32-byte pitch
//   half64      only the first 64 bytes of every 128-byte line (4 lanes x 16 bytes), then — a second kernel, after the whole
//               buffer went by — the other halves (the chain kernel's first form, profiles/r6_chain_full_lines_ab.txt)
//   dword4      4 bytes per lane, lanes contiguous (fp32 element-wise kernels)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- ./fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned fold(u32x4 v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }

__global__ __launch_bounds__(256) void calib_wide16(const char* __restrict__ x, unsigned* __restrict__ out, size_t bytes) {
    unsigned acc = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * 256 * 16)
        acc ^= fold(*reinterpret_cast<const u32x4*>(x + o));
    if (acc == 0x12345677u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_planes16(const char* __restrict__ x, unsigned* __restrict__ out, size_t bytes) {
    unsigned acc = 0;
    const size_t groups = bytes / 96;                       // 16-channel groups of [plane][16] 16-bit values
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < groups * 2; t += (size_t)gridDim.x * 256) {
        const char* src = x + (t >> 1) * 96 + (t & 1) * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) acc ^= fold(*reinterpret_cast<const u32x4*>(src + p * 32));
    }
    if (acc == 0x12345677u) out[0] = acc;
}

template <int HALF>
__global__ __launch_bounds__(256) void calib_half64(const char* __restrict__ x, unsigned* __restrict__ out, size_t bytes) {
    unsigned acc = 0;
    const size_t lines = bytes / 128;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < lines * 4; t += (size_t)gridDim.x * 256)
        acc ^= fold(*reinterpret_cast<const u32x4*>(x + (t >> 2) * 128 + HALF * 64 + (t & 3) * 16));
    if (acc == 0x12345677u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_dword4(const char* __restrict__ x, unsigned* __restrict__ out, size_t bytes) {
    unsigned acc = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; o < bytes; o += (size_t)gridDim.x * 256 * 4)
        acc ^= *reinterpret_cast<const unsigned*>(x + o);
    if (acc == 0x12345677u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1536 << 20;                // 1.5 GiB = 6 x the Infinity Cache; a multiple of 96 and of 128
    char* x;
    unsigned* out;
    CHECK(hipMalloc(&x, bytes));
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(x, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char* name, size_t touched, auto launch) {
        launch();                                            // (twice: the second is the one a reader takes from the counter rows)
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-16s reads %.1f MB exactly once per launch: %.3f ms = %.0f GB/s\n", name, touched / 1e6, ms, touched / ms / 1e6);
    };
    timed("calib_wide16", bytes, [&] { hipLaunchKernelGGL(calib_wide16, dim3(grid), dim3(256), 0, 0, x, out, bytes); });
    timed("calib_planes16", bytes, [&] { hipLaunchKernelGGL(calib_planes16, dim3(grid), dim3(256), 0, 0, x, out, bytes); });
    timed("calib_half64<0>", bytes / 2, [&] { hipLaunchKernelGGL(calib_half64<0>, dim3(grid), dim3(256), 0, 0, x, out, bytes); });
    timed("calib_half64<1>", bytes / 2, [&] { hipLaunchKernelGGL(calib_half64<1>, dim3(grid), dim3(256), 0, 0, x, out, bytes); });
    timed("calib_dword4", bytes, [&] { hipLaunchKernelGGL(calib_dword4, dim3(grid), dim3(256), 0, 0, x, out, bytes); });
    CHECK(hipDeviceSynchronize());
    printf("buffer %.1f MB; FETCH_SIZE rows are KiB\n", bytes / 1e6);
    return 0;
}
