// Stand-alone probe (not part of the library):  hipcc -O3 --offload-arch=gfx950 tools/probe_lds.hip -o build/probe_lds
// Facts the lattice sweep kernel (csrc/lattice_impl.h) relies on, checked on the box:
//   A  16-byte LDS-DMA (global_load_lds_dwordx4) into LDS byte offsets beyond 64 KB (one workgroup, 160 KB dynamic LDS)
//   B  the same from global addresses that are only 4-byte aligned
//   C  what a ds_read_b128 beyond the workgroup's LDS allocation returns
//   D  4-byte aligned global_store_dwordx4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__global__ void dma_probe(const float* __restrict__ src, int src_shift_words, int lds_off_bytes, float* out) {
    extern __shared__ uint4 smem[];
    char* base = reinterpret_cast<char*>(smem);
    const int lane = threadIdx.x;
    const float* g = src + src_shift_words + lane * 4;
    __builtin_amdgcn_global_load_lds((glb_ptr)g, (lds_ptr)(base + lds_off_bytes), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4 v = *reinterpret_cast<const float4*>(base + lds_off_bytes + lane * 16);
    out[lane * 4 + 0] = v.x;
    out[lane * 4 + 1] = v.y;
    out[lane * 4 + 2] = v.z;
    out[lane * 4 + 3] = v.w;
}

__global__ void dma_probe_asm(const float* __restrict__ src, int src_shift_words, int lds_off_bytes, float* out) {
    extern __shared__ uint4 smem[];
    char* base = reinterpret_cast<char*>(smem);
    const int lane = threadIdx.x;
    const float* g = src + src_shift_words + lane * 4;
    const unsigned dst = (unsigned)(size_t)(lds_ptr)(base + lds_off_bytes);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4 v = *reinterpret_cast<const float4*>(base + lds_off_bytes + lane * 16);
    out[lane * 4 + 0] = v.x;
    out[lane * 4 + 1] = v.y;
    out[lane * 4 + 2] = v.z;
    out[lane * 4 + 3] = v.w;
}

__global__ void oob_probe(int off_bytes, float* out) {
    extern __shared__ uint4 smem[];
    float* s = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = 7.0f;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(lds_ptr)smem + (unsigned)off_bytes + threadIdx.x * 16;
    f4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = v.x;
    out[threadIdx.x * 4 + 1] = v.y;
    out[threadIdx.x * 4 + 2] = v.z;
    out[threadIdx.x * 4 + 3] = v.w;
}

__global__ void store_probe(float* dst, int shift_words) {
    f4 v = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
    float* p = dst + shift_words + threadIdx.x * 4;
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

int main() {
    const int N = 1 << 16;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *src, *out;
    hipMalloc(&src, N * 4);
    hipMalloc(&out, N * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> r(256 * 4);
    const int big = 160 * 1024;
    if (hipFuncSetAttribute((const void*)dma_probe, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) printf("set attr failed (builtin)\n");
    if (hipFuncSetAttribute((const void*)dma_probe_asm, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) printf("set attr failed (asm)\n");
    for (int form = 0; form < 2; ++form) {
        for (int off : {0, 4096, 65536 - 1024, 65536, 100000 / 16 * 16, 160 * 1024 - 1024}) {
            for (int shift : {0, 1, 2, 3, 5}) {
                hipMemset(out, 0xff, 1024);
                if (form == 0) hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), big, 0, src, shift, off, out);
                else hipLaunchKernelGGL(dma_probe_asm, dim3(1), dim3(64), big, 0, src, shift, off, out);
                hipError_t e = hipDeviceSynchronize();
                hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
                int bad = 0;
                for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
                printf("A/B %s lds_off=%6d src_shift=%d words: %s (%d mismatches, first %.0f %.0f %.0f %.0f) err=%d\n", form ? "asm    " : "builtin", off, shift,
                       bad ? "FAIL" : "ok", bad, r[0], r[1], r[2], r[3], (int)e);
            }
        }
    }
    for (int alloc : {4096, 65536}) {
        if (hipFuncSetAttribute((const void*)oob_probe, hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess) printf("set attr failed\n");
        for (int off : {0, alloc - 1024, alloc, alloc + 4096, 65536, 131072, 163840, 262144, 524288, 1 << 20}) {
            hipMemset(out, 0xff, 1024);
            hipLaunchKernelGGL(oob_probe, dim3(1), dim3(64), alloc, 0, off, out);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
            printf("C alloc=%d read at +%d: %g %g %g %g ... %g (err=%d)\n", alloc, off, r[0], r[1], r[2], r[3], r[255], (int)e);
        }
    }
    for (int shift : {0, 1, 2, 3}) {
        hipMemset(out, 0, 2048);
        hipLaunchKernelGGL(store_probe, dim3(1), dim3(64), 0, 0, out, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
        printf("D store shift=%d: %g %g %g %g %g %g (err=%d)\n", shift, r[0], r[1], r[2], r[3], r[4], r[5], (int)e);
    }
    return 0;
}
