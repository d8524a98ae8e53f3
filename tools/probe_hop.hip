// Developer probe (not part of the product): the floor of one dependency hop of the sync-free triangular solve —
// an agent-scope store by one workgroup seen by an agent-scope polling load of another.  Two workgroups play ping-pong on
// two words; the round trip / 2 is the hop.  Workgroups are dispatched round-robin over the 8 XCDs, so workgroups 0 and 1
// sit on different XCDs (the common case in the solve: 256 persistent workgroups), 0 and 8 on the same one.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_hop.hip -o /tmp/probe_hop && /tmp/probe_hop
#include <hip/hip_runtime.h>

#include <cstdio>

template <int SLEEP>
__global__ void pingpong(unsigned* a, unsigned* b, int partner, int rounds, long long* ticks) {
    const int me = blockIdx.x;
    if (me != 0 && me != partner) return;
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (int k = 1; k <= rounds; ++k) {
        if (me == 0) {
            __hip_atomic_store(a, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)k) {
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
        } else {
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)k) {
                if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            }
            __hip_atomic_store(b, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) *ticks = wall_clock64() - t0;
}

template <int SLEEP>
static void run(const char* what, int partner, unsigned* a, unsigned* b, long long* ticks, int rate_khz) {
    const int rounds = 20000;
    hipMemset(a, 0, 4);
    hipMemset(b, 0, 4);
    hipLaunchKernelGGL(pingpong<SLEEP>, dim3(partner + 1), dim3(64), 0, 0, a, b, partner, rounds, ticks);
    hipDeviceSynchronize();
    long long t = 0;
    hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    const double us = (double)t / rate_khz * 1e3;
    std::printf("%-44s s_sleep %d: %.3f us per hop (%d round trips, %.1f us)\n", what, SLEEP, us / (2.0 * rounds), rounds, us);
}

int main() {
    unsigned *a, *b;
    long long* ticks;
    hipMalloc(&a, 256);
    hipMalloc(&b, 256);   // (separate allocations: different cache lines)
    hipMalloc(&ticks, 8);
    int rate_khz = 100000;   // wall_clock64 ticks at 100 MHz on gfx9
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    std::printf("wall clock %d kHz\n", rate_khz);
    run<0>("different XCDs (workgroups 0 and 1)", 1, a, b, ticks, rate_khz);
    run<1>("different XCDs (workgroups 0 and 1)", 1, a, b, ticks, rate_khz);
    run<4>("different XCDs (workgroups 0 and 1)", 1, a, b, ticks, rate_khz);
    run<0>("same XCD (workgroups 0 and 8)", 8, a, b, ticks, rate_khz);
    run<4>("same XCD (workgroups 0 and 8)", 8, a, b, ticks, rate_khz);
    run<0>("far XCD (workgroups 0 and 4)", 4, a, b, ticks, rate_khz);
    return 0;
}
