// Microbenchmark (development aid): write bandwidth of store patterns a decode kernel could use.
//   0: coalesced — each wave writes 1 KB contiguous per dwordx4 store instruction
//   1: per-lane streams, 16 B per lane per instruction, lane streams `stream_ints` apart (4-byte aligned, +1 int skew)
//   2: per-lane streams, 8 B stores      3: per-lane streams, 4 B stores
//   4: per-lane streams, mixed 4/8/16 B in rotation (what variable-size codewords would do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) a4x4 { u32x4 v; };
struct __attribute__((packed, aligned(4))) a4x2 { u32x2 v; };

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, size_t ints_per_wave, unsigned stream_ints) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    unsigned* base = out + wave * ints_per_wave;
    if (MODE == 0) {
        for (size_t i = 0; i + 256 <= ints_per_wave; i += 256) {
            u32x4 v = {unsigned(i), lane, 2, 3};
            reinterpret_cast<a4x4*>(base + i + 1 + 4 * lane)->v = v;
        }
    } else {
        unsigned* p = base + size_t(lane) * stream_ints + 1;  // +1: not 16-byte aligned, like real output
        unsigned left = stream_ints - 4;
        unsigned it = 0;
        while (left >= 4) {
            unsigned n = MODE == 1 ? 4 : MODE == 2 ? 2 : MODE == 3 ? 1 : (1u << ((it + lane) % 3));
            if (n == 4) { u32x4 v = {it, lane, 2, 3}; reinterpret_cast<a4x4*>(p)->v = v; }
            else if (n == 2) { u32x2 v = {it, lane}; reinterpret_cast<a4x2*>(p)->v = v; }
            else { *p = it; }
            p += n; left -= n; ++it;
        }
    }
}

int main(int argc, char** argv) {
    const size_t total_ints = size_t(1) << 30;  // 4 GB
    unsigned* d; hipMalloc(&d, total_ints * 4 + 4096);
    const int waves = 256 * 16;
    const size_t per_wave = total_ints / waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned stream_ints : {per_wave / 64 > 0 ? unsigned(per_wave / 64) : 64u, 1024u, 256u}) {
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e9;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0);
                size_t pw = mode == 0 ? per_wave : size_t(stream_ints) * 64;
                // modes >0: every wave covers 64 streams of stream_ints; loop over chunks via more waves
                size_t nw = mode == 0 ? waves : total_ints / pw;
                dim3 grid((nw * 64 + 1023) / 1024);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, grid, dim3(1024), 0, 0, d, pw, stream_ints); break;
                    case 1: hipLaunchKernelGGL(k<1>, grid, dim3(1024), 0, 0, d, pw, stream_ints); break;
                    case 2: hipLaunchKernelGGL(k<2>, grid, dim3(1024), 0, 0, d, pw, stream_ints); break;
                    case 3: hipLaunchKernelGGL(k<3>, grid, dim3(1024), 0, 0, d, pw, stream_ints); break;
                    case 4: hipLaunchKernelGGL(k<4>, grid, dim3(1024), 0, 0, d, pw, stream_ints); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("stream_ints %7u mode %d: %.3f ms  %.1f GB/s\n", stream_ints, mode, best, total_ints * 4.0 / best / 1e6);
        }
    }
    return 0;
}
