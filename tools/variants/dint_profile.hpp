// Diagnostic build only (-DDINT_PROFILE -Itools/variants): where does a wavefront of the decode kernels spend
// its cycles? Every SECTION() mark reads the shader clock (s_memtime) and adds the cycles since the previous
// mark to that section's per-wave accumulator in LDS (a no-return ds_add from lane 0: no round trip); at the end
// of the kernel the waves add their accumulators into g_prof. Not part of the product: dint_kernels.hpp
// includes this file only under DINT_PROFILE, and the plain build compiles the marks away.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (included from inside namespace dint_dev)
constexpr uint32_t kProfSections = 16;
__device__ unsigned long long g_prof[kProfSections];
// Per compute unit (XCC x the SE / SH / CU fields of HW_ID): waves that ran there, their lifetime in cycles, the
// cycles they spent decoding (every section but 0), the work items they drew — is the dynamic queue's work evenly
// spread over the chip?
constexpr uint32_t kProfCus = 8 * 256;
__device__ unsigned long long g_cu[kProfCus * 4];

// The one-launch query form (query_fused_body): thread 0's clock at its phase boundaries, of the LAST launch.
__device__ unsigned long long g_qtrace[32];
#define QTRACE(k) do { if (threadIdx.x == 0 && (k) < 32) g_qtrace[(k)] = __builtin_amdgcn_s_memtime(); } while (0)

struct prof_t {
    uint32_t last = 0, id = 0;
    uint32_t* acc = nullptr;  // kProfSections words in LDS, this wave's
    uint32_t lane = 0;
    uint32_t t0 = 0, items = 0;
};

__device__ __forceinline__ void prof_stamp(prof_t& p, uint32_t id) {
    const uint32_t now = uint32_t(__builtin_amdgcn_s_memtime());
    if (p.acc && p.lane == 0) __hip_atomic_fetch_add(p.acc + p.id, now - p.last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    p.last = now;
    p.id = id;
    if (id == 14) ++p.items;  // (the queue's draw: one work item)
}
__device__ __forceinline__ void prof_begin(prof_t& p, uint32_t* acc, uint32_t lane) {
    p.acc = acc;
    p.lane = lane;
    if (lane < kProfSections) acc[lane] = 0;
    p.last = uint32_t(__builtin_amdgcn_s_memtime());
    p.t0 = p.last;
    p.id = 0;
}
__device__ __forceinline__ void prof_end(prof_t& p) {
    prof_stamp(p, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p.lane < kProfSections) atomicAdd(&g_prof[p.lane], (unsigned long long)p.acc[p.lane]);
    if (p.lane == 0) {
        uint32_t hw, xcc, busy = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        for (uint32_t i = 1; i != kProfSections; ++i) busy += p.acc[i];
        unsigned long long* const cu = g_cu + 4 * ((xcc & 7u) * 256 + ((hw >> 8) & 255u));
        atomicAdd(cu + 0, 1ull);
        atomicAdd(cu + 1, (unsigned long long)(p.last - p.t0));
        atomicAdd(cu + 2, (unsigned long long)busy);
        atomicAdd(cu + 3, (unsigned long long)p.items);
    }
}
#define SECTION(pf, id, name) do { MARK(name); prof_stamp(pf, id); } while (0)
