// Pure-MFMA loop (no memory traffic) at one wave per SIMD: sustained rate of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 on random
// operands under the package power limit.  hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power; ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 16 + i) & 65535]; b[i] = src[(tid * 16 + 8 + i) & 65535]; }
    float s = 0.f;
    if (KIND == 0) {
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j]), "v"(a[i]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0];
    } else if (KIND == 2) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(b[j + 4 * (rep & 1)]), "a"(a[i + 4 * (rep >> 1)]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    } else {
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)                      // two 16-deep steps = the same 32-deep slice of a 128 x 128 quadrant
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[4 * kk + j]), "v"(a[4 * kk + i]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    }
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& x : h) { float f = ((rand() & 0xFFFF) / 32768.0f - 1.0f); unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
    bf16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 1024 * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 1024;
    for (int kind = 0; kind < 3; ++kind)
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < 20; ++l) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 20.0 * blocks * 4 * iters * 64.0 * 16 * 16 * 32 * 2;
            if (rep % 3 == 2) printf("%s rep %d: %.0f TFLOP/s\n", kind == 0 ? "16x16x32 (acc in AGPR)" : kind == 1 ? "32x32x16" : "16x16x32 (source in AGPR, acc in VGPR, 16 accumulators)", rep, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
