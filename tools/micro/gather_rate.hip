// Microbenchmark (development aid): how many random L2-resident gathers per second can a CU issue?
// 16 waves per CU (one 1024-thread workgroup, like the decode kernel). Per iteration every lane
// issues `MLP` independent buffer loads at hashed offsets into a table of `table_kb` KB, then
// uses them. Modes: 0 = 4-byte loads, 1 = 16-byte loads (dword-aligned), 2 = 4-byte loads of which
// only every 4th lane is active (sparse exec mask), 3 = 4-byte loads + a 1 KB/wave coalesced
// store per iteration (the decode kernel's mix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE, int MLP>
__global__ __launch_bounds__(1024) void k(const unsigned* table, unsigned table_words, unsigned iters, unsigned* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, int(table_words * 4), 0x00020000);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    unsigned* obase = out + size_t(tid >> 6) * 256 * 64;  // 64 KB per wave, rewritten
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 256 * 64 * 4, 0x00020000);
    unsigned acc = 0, h = mix(tid * 2654435761u + 1);
    for (unsigned it = 0; it != iters; ++it) {
        unsigned v[MLP];
#pragma unroll
        for (int j = 0; j != MLP; ++j) {
            h = mix(h + j + 1);
            const unsigned off = (h % (table_words - 4)) * 4;
            if (MODE == 1) {
                u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                v[j] = q.x + q.y + q.z + q.w;
            } else if (MODE == 2) {
                v[j] = 0;
                if ((lane & 3) == 0) v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
            } else {
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j != MLP; ++j) acc += v[j];
        if (MODE == 3) {
            u32x4 s = {acc, it, lane, 1};
            __builtin_amdgcn_raw_buffer_store_b128(s, ro, 16 * lane, (it & 63) * 1024, 0);
        }
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int MODE, int MLP>
float run(const unsigned* t, unsigned words, unsigned iters, unsigned* out, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, MLP>), dim3(cus), dim3(1024), 0, 0, t, words, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    unsigned* out; hipMalloc(&out, size_t(cus) * 16 * 256 * 64 * 4);
    const unsigned iters = 2000;
    for (unsigned kb : {16u, 256u, 1280u, 16384u}) {
        const unsigned words = kb * 256;
        unsigned* t; hipMalloc(&t, words * 4); hipMemset(t, 1, words * 4);
        struct { const char* name; float ms; double per_lane; } rows[6];
        rows[0] = {"b32  mlp4        ", run<0, 4>(t, words, iters, out, cus), 4.0};
        rows[1] = {"b32  mlp16       ", run<0, 16>(t, words, iters, out, cus), 16.0};
        rows[2] = {"b128 mlp4        ", run<1, 4>(t, words, iters, out, cus), 4.0};
        rows[3] = {"b32  mlp16 1/4   ", run<2, 16>(t, words, iters, out, cus), 4.0};
        rows[4] = {"b32  mlp4 +store ", run<3, 4>(t, words, iters, out, cus), 4.0};
        rows[5] = {"b32  mlp1        ", run<0, 1>(t, words, iters, out, cus), 1.0};
        for (auto& r : rows) {
            const double lanes = double(cus) * 1024 * iters * r.per_lane;
            printf("table %6u KB  %s %8.3f ms  %7.1f G lane-loads/s  %.3f per CU-cycle (%.2f GHz)\n", kb, r.name, r.ms,
                   lanes / r.ms / 1e6, lanes / cus / (r.ms * 1e-3 * ghz * 1e9), ghz);
        }
        hipFree(t);
    }
    return 0;
}
