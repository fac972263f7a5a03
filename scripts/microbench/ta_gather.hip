// How the gfx950 texture-address path prices the gather shapes the registration kernels use: map points are 24 contiguous bytes
// (x, y, z doubles). Every pattern moves the same 64 points per "step" of a wave; the table is L2-resident (the point is the address
// path, not HBM). Build: hipcc -O3 --offload-arch=gfx950 ta_gather.hip -o ta_gather ; run: ./ta_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// region: npts points of 24 B. Each pattern returns an xor of everything it loaded so nothing is dropped.
template <int PAT>
__global__ __launch_bounds__(256) void k(const char *base, uint32_t npts, int iters, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t step = (uint32_t) it * 8 + u;
            if (PAT == 1) {            // one scattered point per lane: dwordx4 + dwordx2
                const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + lane) % npts;
                const char *a = base + (size_t) p * 24;
                const u4 v = *reinterpret_cast<const u4 *>(a);
                const u2 w = *reinterpret_cast<const u2 *>(a + 16);
                acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y;
            } else if (PAT == 2) {     // one scattered point per lane PAIR, two instructions (64 points): dwordx4 each, 8 bytes wasted
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + h * 32 + (lane >> 1)) % (npts - 1);
                    const char *a = base + (size_t) p * 24 + (lane & 1) * 16;
                    const u4 v = *reinterpret_cast<const u4 *>(a);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
            } else if (PAT == 3) {     // lane pairs, dwordx3 each (exactly the 24 bytes)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + h * 32 + (lane >> 1)) % npts;
                    const char *a = base + (size_t) p * 24 + (lane & 1) * 12;
                    const u3 v = *reinterpret_cast<const u3 *>(a);
                    acc ^= v.x ^ v.y ^ v.z;
                }
            } else if (PAT == 4) {     // four rows of 16 CONTIGUOUS points (stride 24 B) at scattered bases: dwordx4 + dwordx2 per lane
                const uint32_t p = (mix(wave * 0x9e3779b9u + step * 4 + (lane >> 4)) % (npts / 16 - 1)) * 16 + (lane & 15);
                const char *a = base + (size_t) p * 24;
                const u4 v = *reinterpret_cast<const u4 *>(a);
                const u2 w = *reinterpret_cast<const u2 *>(a + 16);
                acc ^= v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y;
            } else if (PAT == 5) {     // the same rows as three dense dwordx2 loads (lane s: bytes 8 s .. 8 s + 8 of each 128-byte third)
                const uint32_t p = (mix(wave * 0x9e3779b9u + step * 4 + (lane >> 4)) % (npts / 16 - 1)) * 16;
                const char *a = base + (size_t) p * 24 + (lane & 15) * 8;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const u2 w = *reinterpret_cast<const u2 *>(a + 128 * m);
                    acc ^= w.x ^ w.y;
                }
            } else if (PAT == 6) {     // the same rows as two dense dwordx3 loads (lane s: bytes 12 s .. 12 s + 12 of each 192-byte half)
                const uint32_t p = (mix(wave * 0x9e3779b9u + step * 4 + (lane >> 4)) % (npts / 16 - 1)) * 16;
                const char *a = base + (size_t) p * 24 + (lane & 15) * 12;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const u3 v = *reinterpret_cast<const u3 *>(a + 192 * m);
                    acc ^= v.x ^ v.y ^ v.z;
                }
            } else if (PAT == 7) {     // one scattered point per lane, three dwordx2 (what a compiler may emit for three doubles)
                const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + lane) % npts;
                const char *a = base + (size_t) p * 24;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const u2 w = *reinterpret_cast<const u2 *>(a + 8 * m);
                    acc ^= w.x ^ w.y;
                }
            } else if (PAT == 8) {     // scattered point per lane, dwordx4 only (16 of the 24 bytes): one instruction's worth of lookups
                const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + lane) % npts;
                const u4 v = *reinterpret_cast<const u4 *>(base + (size_t) p * 24);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            } else if (PAT == 9) {     // lane QUADS: four lanes x dwordx2 cover 32 bytes of one point, four instructions per 64 points
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const uint32_t p = mix(wave * 0x9e3779b9u + step * 64 + h * 16 + (lane >> 2)) % (npts - 1);
                    const u2 w = *reinterpret_cast<const u2 *>(base + (size_t) p * 24 + (lane & 3) * 8);
                    acc ^= w.x ^ w.y;
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT>
int run(const char *name, const char *d, uint32_t npts, uint32_t *sink, int grid, int iters) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, d, npts, iters, sink);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, d, npts, iters, sink);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double steps = (double) grid * 4 * iters * 8;             // wave-steps of 64 points
    printf("%-62s %8.3f ms  %7.1f G points/s  %6.1f clk per 64-point step and CU (2.4 GHz)\n", name, best, steps * 64 / best * 1e-6,
           best * 1e-3 * 2.4e9 / (steps / 256));
    return 0;
}

int main(int argc, char **argv) {
    for (int pass = 0; pass < 4; ++pass) {
        // 12 KB (every CU's L1) | 1.5 MB (every XCD's L2) | 24 MB (memory-side cache) | 288 MB (the B2 level's size: HBM behind it)
        const uint32_t npts = pass == 0 ? 512u : pass == 1 ? (1u << 16) : pass == 2 ? (1u << 20) : (12u << 20);
        char *d; uint32_t *sink;
        CHK(hipMalloc(&d, (size_t) npts * 24 + 256));
        CHK(hipMemset(d, 1, (size_t) npts * 24 + 256));
        CHK(hipMalloc(&sink, 64));
        const int grid = 256 * 8, iters = pass < 3 ? 64 : 16;          // 8 blocks of 4 waves per CU: 32 waves per CU, 8 per SIMD
        printf("table of %u points (%.0f MB), %d waves per CU\n", npts, npts * 24.0 / 1e6, grid * 4 / 256);
        if (run<1>("1 scattered, lane = point: dwordx4 + dwordx2", d, npts, sink, grid, iters)) return 1;
        if (run<7>("7 scattered, lane = point: 3 x dwordx2", d, npts, sink, grid, iters)) return 1;
        if (run<8>("8 scattered, lane = point: dwordx4 only (16 of 24 bytes)", d, npts, sink, grid, iters)) return 1;
        if (run<2>("2 scattered, lane PAIR = point: dwordx4 each", d, npts, sink, grid, iters)) return 1;
        if (run<3>("3 scattered, lane PAIR = point: dwordx3 each", d, npts, sink, grid, iters)) return 1;
        if (run<9>("9 scattered, lane QUAD = point: dwordx2 each", d, npts, sink, grid, iters)) return 1;
        if (run<4>("4 rows of 16 contiguous points, lane = point: x4 + x2", d, npts, sink, grid, iters)) return 1;
        if (run<5>("5 rows of 16 contiguous points, dense: 3 x dwordx2", d, npts, sink, grid, iters)) return 1;
        if (run<6>("6 rows of 16 contiguous points, dense: 2 x dwordx3", d, npts, sink, grid, iters)) return 1;
        CHK(hipFree(d)); CHK(hipFree(sink));
    }
    return 0;
}
