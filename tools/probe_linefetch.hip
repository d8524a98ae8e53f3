// Stand-alone probe (not part of the library):  hipcc -O3 --offload-arch=gfx950 tools/probe_linefetch.hip -o /tmp/linefetch
// How many distinct 128-byte lines per microsecond does a wave / the chip pull from HBM through the scalar path (s_load)
// and through the vector path (one dword per line)?  MI355X: vector 165 lines/us for one wave per CU (= 64 outstanding lines
// per CU / 0.39 us), 42-55 k lines/us for the chip (5.4-7.1 TB/s of lines); scalar 28 lines/us per wave, 14-16 k lines/us chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// How fast can ONE wave pull distinct 128-byte lines into L2 through the SCALAR path (s_load), and through the vector path?
__global__ void scalar_touch(const int* __restrict__ base, long stride_words, int n, int* out) {
    const long w = blockIdx.x;
    int acc = 0;
    const int* p = base + w * (long)n * stride_words;
    for (int i = 0; i < n; i += 8) {
        int a0, a1, a2, a3, a4, a5, a6, a7;
        const int* q = p + (long)i * stride_words;
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a0) : "s"(q));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a1) : "s"(q + stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a2) : "s"(q + 2 * stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a3) : "s"(q + 3 * stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a4) : "s"(q + 4 * stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a5) : "s"(q + 5 * stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a6) : "s"(q + 6 * stride_words));
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(a7) : "s"(q + 7 * stride_words));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ void vector_touch(const int* __restrict__ base, long stride_words, int n, int* out) {
    const long w = blockIdx.x;
    int acc = 0;
    const int* p = base + w * (long)n * stride_words;
    for (int i = threadIdx.x; i < n; i += 64) acc += __builtin_nontemporal_load(p + (long)i * stride_words);
    if (acc == 12345) out[blockIdx.x] = acc;
}
int main() {
    const long stride = 32;       // words: one int per 128-byte line
    const int n = 4096;           // lines per wave
    for (int waves : {256, 1024, 4096}) {
        const size_t words = (size_t)waves * n * stride;
        int *buf, *out;
        hipMalloc(&buf, words * 4);
        hipMalloc(&out, waves * 4);
        hipMemset(buf, 0, words * 4);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                // flush caches by touching another big buffer is skipped: the buffer (>= 512 MB at 256 waves) exceeds L2+MALL for waves >= 1024
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(scalar_touch, dim3(waves), dim3(64), 0, 0, buf, stride, n, out);
                else hipLaunchKernelGGL(vector_touch, dim3(waves), dim3(64), 0, 0, buf, stride, n, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("%s waves=%d lines/wave=%d: %.3f ms -> %.1f lines/us total, %.2f lines/us per wave (%.1f GB/s of 128B lines)\n", mode == 0 ? "scalar" : "vector", waves, n, best,
                   (double)waves * n / (best * 1e3), (double)n / (best * 1e3), (double)waves * n * 128 / (best * 1e6));
        }
        hipFree(buf); hipFree(out);
    }
    return 0;
}
