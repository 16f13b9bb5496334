// LDS access patterns of the FAVOR+ slice kernels / attention kernels in isolation: one kernel instance per pattern, 1000 repetitions per wave,
// to be read with `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS` (conflict cycles per instruction per pattern).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_patterns tools/ubench/lds_patterns.hip      (diagnostics only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int ROWB = 128, REP = 1000;

__device__ __forceinline__ int sw_old(int row) { return row & 7; }
__device__ __forceinline__ int sw_new(int row) { return ((row >> 2) & 1) | (row & 2) | ((((row >> 2) ^ (row >> 3)) & 1) << 2); }

template <int P>
__global__ __launch_bounds__(256) void lds_pat(float* sink, const char* gsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    for (int i = tid; i < 16384; i += 256) ((float*)smem)[i] = (float)i;
    __syncthreads();
    char* base = smem + w * 16384;          // a wave-private 16 KB region
    u32x4 acc = {0, 0, 0, 0};
    int off = 0;
    if (P == 0) off = c * ROWB + (((g) ^ sw_old(c)) << 4);                                   // ring fragment, 16 B
    if (P == 1) off = c * ROWB + (g & 1) * 8 + ((((g >> 1)) ^ sw_old(c)) << 4);              // ring permuted 8-B read, r03 swizzle
    if (P == 2 || P == 12) {                                                                  // ring transpose read
        const int row = (lane >> 4) * 4 + (c >> 2), col = (c & 3) * 4;
        off = row * ROWB + (((col >> 3) ^ (P == 2 ? sw_old(row) : sw_new(row))) << 4) + (col & 7) * 2;
    }
    if (P == 3) off = (c * 136 + 16 * w + 4 * g) * 2;                                         // feature store, 8 B, LDF 136
    if (P == 4) off = (c * 136 + 4 * g) * 2;                                                  // load_perm 8-B read, LDF 136
    if (P == 5) off = (((lane >> 4) * 4 + (c >> 2)) * 136 + (c & 3) * 4) * 2;                 // load_perm_tr on the feature image, LDF 136
    if (P == 7) off = c * 4;                                                                  // (bpermute index)
    if (P == 8) off = 4 * g * 4;                                                              // broadcast f32x4 read
    if (P == 9) off = c * ROWB + (g & 1) * 8 + ((((g >> 1)) ^ sw_old(c)) << 4);              // dU store 8 B, ring layout
    if (P == 11) off = c * ROWB + (g & 1) * 8 + ((((g >> 1)) ^ sw_new(c)) << 4);             // ring permuted 8-B read, candidate swizzle
    if (P == 13) off = c * ROWB + (((g) ^ sw_new(c)) << 4);                                   // ring fragment, candidate swizzle
    if (P == 15) {                                                                            // load_perm_tr, unpadded image with the 8-B unit swizzle
        const int r = (lane >> 4) * 4 + (c >> 2), col = (c & 3) * 4;
        off = r * 256 + ((((col >> 2)) ^ ((4 * (r & 7)) ^ (2 * ((r >> 3) & 1)))) << 3);
    }
    if (P == 14) off = c * 256 + (((g) ^ ((4 * (c & 7)) ^ (2 * ((c >> 3) & 1)))) << 3);      // load_perm on the same image
    const __attribute__((address_space(3))) char* lp = (const __attribute__((address_space(3))) char*)(base + off);
    for (int it = 0; it < REP; ++it) {
        if (P == 0 || P == 8 || P == 13) { u32x4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lp) : "memory"); acc += v; }
        if (P == 1 || P == 4 || P == 11 || P == 14) { u32x2 v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lp) : "memory"); acc[0] += v[0] + v[1]; }
        if (P == 2 || P == 5 || P == 12 || P == 15) { u32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lp) : "memory"); acc[0] += v[0] + v[1]; }
        if (P == 3 || P == 9) { u32x2 v = {(unsigned)it, (unsigned)lane}; asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(lp), "v"(v) : "memory"); }
        if (P == 6) {                                                                         // LDS-DMA, 1 KB per wave instruction
            const char* src = gsrc + (size_t)(blockIdx.x * 4 + w) * 1024 + lane * 16;
            const uint32_t dst = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)base;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
        }
        if (P == 7) { unsigned v; asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off), "v"(lane) : "memory"); acc[0] += v; }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[tid] = 1.f;
}

template <int P> static void run(float* sink, const char* g) {
    hipFuncSetAttribute((const void*)lds_pat<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(lds_pat<P>, dim3(256), dim3(256), 65536, 0, sink, g);
    hipDeviceSynchronize();
}
int main() {
    float* sink; char* g;
    hipMalloc(&sink, 4096); hipMalloc(&g, 4 << 20);
    run<0>(sink, g); run<1>(sink, g); run<2>(sink, g); run<3>(sink, g); run<4>(sink, g); run<5>(sink, g); run<6>(sink, g); run<7>(sink, g);
    run<8>(sink, g); run<9>(sink, g); run<11>(sink, g); run<12>(sink, g); run<13>(sink, g); run<14>(sink, g); run<15>(sink, g);
    printf("done\n");
    return 0;
}
