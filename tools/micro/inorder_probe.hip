// Microbenchmark (development aid): is a CU's vector-memory return path in order ACROSS waves?
// One 1024-thread workgroup per CU. Waves [0, S) are "streamers": each keeps DEPTH coalesced 512-byte loads of a
// buffer far larger than every cache in flight (HBM latency). Waves [S, 16) are "gatherers": dependent random 16-byte
// gathers from a 256 KB table (L2 hits), one at a time, timed with s_memtime. If the return path is one in-order pipe
// per CU, a gatherer's latency rises to the HBM latency as soon as one streamer runs; if it is in order per wave only,
// it does not. Control: the streamers read an L2-resident buffer instead (same instruction mix, no HBM latency).
// Also: do SCALAR loads bring a line into L2 for a later vector load (mode "sprefetch")?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/inorder_probe.hip -o /tmp/inorder_probe ; run: /tmp/inorder_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// res[cu * 16 + wave] = {cycles spent, operations done}
template <int DEPTH, int MLP = 1>
__global__ __launch_bounds__(1024) void k(const unsigned* big, size_t big_bytes, const unsigned* table, unsigned table_words,
                                          unsigned streamers, unsigned iters, unsigned long long* res, unsigned* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, int(table_words * 4), 0x00020000);
    unsigned acc = 0;
    __shared__ unsigned done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), ops = 0;
    if (wave < streamers) {
        // coalesced 512-byte loads, each from its own far-apart place; run until the gatherers are done
        size_t pos = (size_t(blockIdx.x) * 16 + wave) * (big_bytes / (256 * 16)) & ~size_t(511);
        const size_t lim = big_bytes - 4096;
        volatile unsigned* dn = &done;
        while (*dn < 16 - streamers) {
            u32x2 v[DEPTH];
#pragma unroll
            for (int j = 0; j != DEPTH; ++j) {
                v[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(big) + pos + 8 * lane);
                pos += 512 * 257;  // (another DRAM page every time)
                if (pos >= lim) pos -= lim & ~size_t(511);
            }
#pragma unroll
            for (int j = 0; j != DEPTH; ++j) acc += v[j].x + v[j].y;
            ops += DEPTH;
        }
    } else {
        unsigned h = mix((blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 1);
        for (unsigned it = 0; it != iters; ++it) {
            u32x4 q[MLP];
#pragma unroll
            for (int j = 0; j != MLP; ++j) {  // MLP independent gathers, then all of them used
                const unsigned off = ((h + 977u * j) & (table_words - 1) & ~3u) * 4;
                q[j] = __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, 0);
            }
            unsigned x = 0;
#pragma unroll
            for (int j = 0; j != MLP; ++j) x += q[j].x;
            h = h * 1664525u + 1013904223u + (x & 1u);  // dependent: the next addresses need this data
        }
        acc += h;
        ops = iters;
        if (lane == 0) atomicAdd(&done, 1u);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        res[(size_t(blockIdx.x) * 16 + wave) * 2] = t1 - t0;
        res[(size_t(blockIdx.x) * 16 + wave) * 2 + 1] = ops;
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

// Latency-dominated form: gatherer waves [12, 16) only, 4 active lanes each (a gather instruction occupies the CU's
// gather path for ~9 cycles), dependent chain; streamers are waves [0, S). Waves in between exit.
template <int DEPTH>
__global__ __launch_bounds__(1024) void k4(const unsigned* big, size_t big_bytes, const unsigned* table, unsigned table_words,
                                           unsigned streamers, unsigned iters, unsigned long long* res, unsigned* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, int(table_words * 4), 0x00020000);
    unsigned acc = 0;
    __shared__ unsigned done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), ops = 0;
    if (wave < streamers) {
        size_t pos = (size_t(blockIdx.x) * 16 + wave) * (big_bytes / (256 * 16)) & ~size_t(511);
        const size_t lim = big_bytes - 4096;
        volatile unsigned* dn = &done;
        while (*dn < 4) {
            u32x2 v[DEPTH];
#pragma unroll
            for (int j = 0; j != DEPTH; ++j) {
                v[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(big) + pos + 8 * lane);
                pos += 512 * 257;
                if (pos >= lim) pos -= lim & ~size_t(511);
            }
#pragma unroll
            for (int j = 0; j != DEPTH; ++j) acc += v[j].x + v[j].y;
            ops += DEPTH;
        }
    } else if (wave >= 12) {
        unsigned h = mix((blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 1);
        if (lane < 4) {
            for (unsigned it = 0; it != iters; ++it) {
                const unsigned off = (h & (table_words - 1) & ~3u) * 4;
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, 0);
                h = h * 1664525u + 1013904223u + (q.x & 1u);
            }
        }
        acc += h;
        ops = iters;
        if (lane == 0) atomicAdd(&done, 1u);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        res[(size_t(blockIdx.x) * 16 + wave) * 2] = (wave < streamers || wave >= 12) ? t1 - t0 : 0;
        res[(size_t(blockIdx.x) * 16 + wave) * 2 + 1] = (wave < streamers || wave >= 12) ? ops : 0;
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

// scalar prefetch: wave 0 of each workgroup touches its 64 KB region with s_load_dword (one per 64 bytes), waits, then
// every wave reads the region with vector loads, timed; against the same without the scalar touches (cold) and a second
// vector pass (warm).
__global__ __launch_bounds__(64) void sp(const unsigned* big, unsigned region_bytes, int mode, unsigned long long* res, unsigned* sink) {
    const unsigned lane = threadIdx.x;
    const char* base = reinterpret_cast<const char*>(big) + size_t(blockIdx.x) * region_bytes;
    unsigned acc = 0;
    if (mode == 3) {  // s_atc_probe: "probe or prefetch an address into the scalar data cache" (no destination register)
        for (unsigned o = 0; o < region_bytes; o += 64) {
            const unsigned long long a = reinterpret_cast<unsigned long long>(base + o);
            asm volatile("s_atc_probe 0, %0, 0x0" : : "s"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (mode == 4) {  // scalar loads, up to 15 in flight (no wait in between), one per 128 bytes
        unsigned v = 0;
        for (unsigned o = 0; o < region_bytes; o += 128) {
            const unsigned long long a = reinterpret_cast<unsigned long long>(base + o);
            asm volatile("s_load_dword %0, %1, 0x0" : "+s"(v) : "s"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
        acc += v & 0;
    }
    if (mode == 1) {
        for (unsigned o = 0; o < region_bytes; o += 64) {
            unsigned v;
            const unsigned long long a = reinterpret_cast<unsigned long long>(base + o);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
            acc += v;
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned o = 0; o < region_bytes; o += 1024) {  // dependent chain: one 1 KB wave load at a time
        const u32x4 q = *reinterpret_cast<const u32x4*>(base + o + 16 * lane + (acc & 0));
        acc += q.x + q.y + q.z + q.w;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long warm = 0;
    if (mode == 2) {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        for (unsigned o = 0; o < region_bytes; o += 1024) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(base + o + 16 * lane + (acc & 0));
            acc += q.x + q.y + q.z + q.w;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        warm = __builtin_amdgcn_s_memtime() - t2;
    }
    if (lane == 0) {
        res[blockIdx.x * 2] = t1 - t0;
        res[blockIdx.x * 2 + 1] = warm;
    }
    if (acc == 0x12345678u) sink[lane] = acc;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t big_bytes = size_t(8) << 30;
    unsigned *big, *table, *sink; unsigned long long* res;
    hipMalloc(&big, big_bytes); hipMemset(big, 1, big_bytes);
    const unsigned table_words = 256 * 1024 / 4;
    hipMalloc(&table, table_words * 4); hipMemset(table, 3, table_words * 4);
    hipMalloc(&sink, 4096); hipMalloc(&res, size_t(cus) * 16 * 16);
    std::vector<unsigned long long> h(size_t(cus) * 32);
    const unsigned iters = 2000;
    auto report = [&](const char* what, unsigned streamers) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), res, h.size() * 8, hipMemcpyDeviceToHost);
        double gl = 0, gn = 0, sl = 0, sn = 0;
        for (int c = 0; c < cus; ++c)
            for (unsigned w = 0; w < 16; ++w) {
                const double cyc = double(h[(size_t(c) * 16 + w) * 2]), ops = double(h[(size_t(c) * 16 + w) * 2 + 1]);
                if (ops == 0) continue;
                if (w < streamers) sl += cyc, sn += ops; else gl += cyc, gn += ops;
            }
        printf("%-40s streamers %2u: gather latency %7.0f memtime units", what, streamers, gl / gn);
        if (sn > 0) printf("   streamer: %7.0f ticks per load", sl / sn);
        printf("\n");
    };
    printf("times in s_memtime units (shader clock cycles on this part)\n");
    for (unsigned s : {0u, 1u, 2u, 4u, 8u}) {
        hipLaunchKernelGGL((k<1>), dim3(cus), dim3(1024), 0, 0, big, big_bytes, table, table_words, s, iters, res, sink);
        report("streamers from HBM, depth 1", s);
    }
    for (unsigned s : {1u, 4u}) {
        hipLaunchKernelGGL((k<4>), dim3(cus), dim3(1024), 0, 0, big, big_bytes, table, table_words, s, iters, res, sink);
        report("streamers from HBM, depth 4", s);
    }
    for (unsigned s : {1u, 4u}) {  // control: the streamers' "big" buffer is 1 MB (L2-resident)
        hipLaunchKernelGGL((k<4>), dim3(cus), dim3(1024), 0, 0, big, size_t(1) << 20, table, table_words, s, iters, res, sink);
        report("streamers from L2 (control), depth 4", s);
    }
    for (unsigned s : {0u, 1u, 4u, 8u}) {
        hipLaunchKernelGGL((k4<4>), dim3(cus), dim3(1024), 0, 0, big, big_bytes, table, table_words, s, 4000u, res, sink);
        report("SPARSE gatherers x4, HBM streamers d4", s);
    }
    for (unsigned s : {1u, 4u, 8u}) {
        hipLaunchKernelGGL((k4<4>), dim3(cus), dim3(1024), 0, 0, big, size_t(1) << 20, table, table_words, s, 4000u, res, sink);
        report("SPARSE gatherers x4, L2 streamers d4", s);
    }
    for (unsigned s : {0u, 1u, 2u}) {  // throughput: 4 independent gathers per gatherer and iteration
        hipLaunchKernelGGL((k<8, 4>), dim3(cus), dim3(1024), 0, 0, big, big_bytes, table, table_words, s, iters, res, sink);
        report("MLP-4 gatherers, HBM streamers depth 8", s);
    }
    for (unsigned s : {1u, 2u}) {
        hipLaunchKernelGGL((k<8, 4>), dim3(cus), dim3(1024), 0, 0, big, size_t(1) << 20, table, table_words, s, iters, res, sink);
        report("MLP-4 gatherers, L2 streamers depth 8", s);
    }
    // scalar prefetch into L2
    const unsigned region = 64 * 1024;
    for (int mode : {0, 1, 2, 3, 4}) {
        // a fresh part of the big buffer every time (nothing cached)
        const unsigned* b = big + (size_t(1) << 28) * (mode + 1);
        hipLaunchKernelGGL(sp, dim3(cus), dim3(64), 0, 0, b, region, mode, res, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), res, size_t(cus) * 16, hipMemcpyDeviceToHost);
        double a = 0, w = 0;
        for (int c = 0; c < cus; ++c) a += double(h[c * 2]), w += double(h[c * 2 + 1]);
        printf("vector pass over 64 KB, %-28s %7.1f ticks per 1 KB load%s", mode == 0 ? "cold" : mode == 1 ? "after scalar touches" : mode == 2 ? "cold, then again (warm):" : mode == 3 ? "after s_atc_probe touches" : "after s_load per 128 B, unwaited",
               a / cus / (region / 1024), mode == 2 ? "" : "\n");
        if (mode == 2) printf("   warm %7.1f\n", w / cus / (region / 1024));
    }
    return 0;
}
