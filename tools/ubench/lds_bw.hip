// LDS -> VGPR read bandwidth of one CU for the fragment-read instructions of the GEMM kernels: W waves per workgroup (one workgroup per CU), each wave
// issues back-to-back conflict-free reads of its own 16-KB region.  Prints bytes per clock per CU (shader clock from clock64()).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_bw tools/ubench/lds_bw.hip      (diagnostics only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
constexpr int REP = 2000;
template <int MODE>      // 0: ds_read_b128, lane-linear (16 B per lane); 1: ds_read_b64 lane-linear; 2: ds_read_b128 in the A-stationary kernel's swizzled fragment pattern
__global__ __launch_bounds__(1024) void lds_bw(unsigned long long* out, int nw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 32768; i += blockDim.x) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const char* base = smem + (w & 7) * 16384;
    int off;
    if (MODE == 0) off = lane * 16;
    else if (MODE == 1) off = lane * 8;
    else { const int r = lane & 15, g = lane >> 4, row = 8 * (r >> 2) + (r & 3); off = row * 256 + ((g ^ ((row & 3) | ((row >> 1) & 12))) << 4); }
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 1) { u32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(off), "i"(k * 1024)); asm volatile("" :: "v"(v)); }
            else { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(off), "i"(k * 1024 + 0)); asm volatile("" :: "v"(v)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[w] = (unsigned long long)(t1 - t0);
    if (acc[0] == 12345) out[63] = 1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    for (int mode = 0; mode < 3; ++mode)
        for (int nw : {1, 4, 8, 16}) {
            hipMemset(d, 0, 64 * 8);
            auto k = mode == 0 ? lds_bw<0> : mode == 1 ? lds_bw<1> : lds_bw<2>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipLaunchKernelGGL(k, dim3(256), dim3(64 * nw), 131072, 0, d, nw);
            hipDeviceSynchronize();
            unsigned long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const double bytes = (double)REP * 8 * 64 * (mode == 1 ? 8 : 16) * nw;
            printf("mode %d (%s) waves/CU %2d: %.1f bytes per clock per CU (%llu clocks)\n", mode, mode == 0 ? "ds_read_b128 linear" : mode == 1 ? "ds_read_b64 linear" : "ds_read_b128 A-stationary fragment pattern", nw, bytes / (double)h[0], h[0]);
        }
    return 0;
}
