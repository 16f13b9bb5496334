// Output-store patterns of the A-stationary GEMM in isolation (r05): how fast does the memory system drain M x N bf16 of output when it is written
// the way the kernel writes it — 2 workgroups per CU, each owning 128 rows and sweeping the columns in 64-column tiles, a wave's store
// instruction covering 16 rows x 64 B (half lines, row pitch N * 2 B) — against full-line and linear variants?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_patterns tools/ubench/store_patterns.hip      (diagnostics only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// P = 0: the kernel's pattern (per tile and wave: four stores of 16 rows x 64 B);  1: the same bytes as two stores of 8 rows x 128 B per 16 rows
// (full lines);  2: a wave stores 32 rows x 128 B of a tile as 4 stores of 8 rows x 128 B but TWO tiles (256 B per row) back to back;
// 3: linear — block b writes its 128 x N x 2 bytes as one contiguous run (what a row-major output would be if a block owned whole rows
// and wrote them at once);  NT: non-temporal stores.  `pace`: s_sleep units between column tiles (0 = as fast as the stores issue).
template <int P, bool NT>
__global__ __launch_bounds__(256, 2) void store_pat(char* out, int64_t N, int pace) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pitch = N * 2, m0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const u32x4 v = {(unsigned)lane, (unsigned)blockIdx.x, 3u, 4u};
    auto st = [&](char* p) {
        if (NT) __builtin_nontemporal_store(v, (u32x4*)p);
        else *(u32x4*)p = v;
    };
    const int n_tiles = (int)(N / 64);
    if (P == 3) {
        char* base = out + (int64_t)blockIdx.x * 128 * pitch;
        for (int64_t o = (int64_t)threadIdx.x * 16; o < 128 * pitch; o += 256 * 16) st(base + o);
        return;
    }
    for (int nt = 0; nt < n_tiles; nt += (P == 2 ? 2 : 1)) {
        if (P == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) st(out + (m0 + 16 * i + (lane & 15)) * pitch + nt * 128 + 64 * h + 16 * (lane >> 4));
        } else if (P == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) st(out + (m0 + 8 * q + (lane >> 3)) * pitch + nt * 128 + 16 * (lane & 7));
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) st(out + (m0 + 4 * q + (lane >> 4)) * pitch + nt * 128 + 16 * (lane & 15));
        }
        if (pace) __builtin_amdgcn_s_sleep(127);
        for (int r = 1; r < pace; ++r) __builtin_amdgcn_s_sleep(127);
    }
}

template <int P, bool NT> static void run(const char* name, char* buf, int64_t M, int64_t N, int pace) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((store_pat<P, NT>), dim3((unsigned)(M / 128)), dim3(256), 0, 0, buf, N, pace);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("N=%5lld pace %d  %-44s %7.1f us  %6.2f TB/s\n", (long long)N, pace, name, best * 1e3, (double)M * N * 2 / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int64_t M = 131072;
    char* buf;
    if (hipMalloc(&buf, (size_t)M * 2048 * 2) != hipSuccess) return 1;
    for (int64_t N : {2048, 512})
        for (int pace : {0, 4}) {
            run<0, false>("kernel pattern: 16 rows x 64 B per store", buf, M, N, pace);
            run<0, true>("kernel pattern, non-temporal", buf, M, N, pace);
            run<1, false>("full lines: 8 rows x 128 B per store", buf, M, N, pace);
            run<2, false>("two tiles at once: 4 rows x 256 B per store", buf, M, N, pace);
            if (!pace) run<3, false>("linear: block's rows as one run", buf, M, N, pace);
        }
    return 0;
}
