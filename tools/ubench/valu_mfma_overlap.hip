// Do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?  (r06: the A-stationary GEMM's column-tile epilogue takes a wave
// 8.2 k cycles for ~1.7 k cycles of VALU content while the other wave of the SIMD runs its MFMA stages.)  One workgroup of 8 waves per CU = 2 waves per
// SIMD (waves w and w + 4 share SIMD w % 4): waves 0-3 run independent v_mfma_f32_16x16x32_bf16 — back to back, or each followed by s_nop GAP so that the
// next one is not waiting at the issue stage for the busy matrix pipe —, waves 4-7 run independent VALU instructions.  Each role is timed alone and
// together, with and without s_setprio 3 on the VALU waves: shader clocks per instruction.  Diagnostics only, not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/valu_mfma_overlap tools/ubench/valu_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int REP = 4000;
template <int MODE, int GAP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int run_mfma, int run_valu, float seed, int prio) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long t0 = clock64();
    if (w < 4) {
        if (run_mfma) {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + lane); b[i] = (__bf16)(seed * 2 + i); }
            f32x4 c[8];
            for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
            for (int it = 0; it < REP; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (GAP == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
                    else if constexpr (GAP == 8) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(c[j]) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 11" : "+v"(c[j]) : "v"(a), "v"(b));
                }
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            float s = 0; for (int i = 0; i < 8; ++i) s += c[i][0];
            if (s == 1234.5f) out[63] = 1;
        }
    } else if (run_valu) {
        if (prio) asm volatile("s_setprio 3");
        float x[8]; unsigned u[8];
        for (int i = 0; i < 8; ++i) { x[i] = seed + i + lane; u[i] = (unsigned)(lane * 7 + i); }
        for (int it = 0; it < REP; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(seed));
                else if (MODE == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(0x7feb352du));
                else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            }
        }
        float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
        if (s == 1234.5f) out[62] = 1;
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[w] = (unsigned long long)(t1 - t0);
}
template <int MODE, int GAP> void run(unsigned long long* d, const char* name) {
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int rm = cfg != 1, rv = cfg != 0, prio = cfg == 3;
        (void)hipMemset(d, 0, 64 * 8);
        hipLaunchKernelGGL((k<MODE, GAP>), dim3(256), dim3(512), 0, 0, d, rm, rv, 1.0f, prio);
        (void)hipDeviceSynchronize();
        unsigned long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-12s MFMA + s_nop %-2d  %-15s: MFMA wave %6.1f clocks per MFMA, VALU wave %6.2f clocks per instruction\n", name, GAP, cfg == 0 ? "MFMA only" : cfg == 1 ? "VALU only" : cfg == 2 ? "both" : "both, VALU prio",
               rm ? (double)h[0] / (REP * 8.0) : 0.0, rv ? (double)h[4] / (REP * 8.0) : 0.0);
    }
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64 * 8);
    run<0, 0>(d, "v_fma_f32"); run<0, 8>(d, "v_fma_f32"); run<0, 12>(d, "v_fma_f32");
    run<1, 0>(d, "v_mul_lo_u32"); run<2, 0>(d, "v_xor_b32"); run<2, 12>(d, "v_xor_b32");
    return 0;
}
