// Micro-probe: can a kernel start while its predecessor ON THE SAME STREAM still runs (hipExtAnyOrderLaunch: the dispatch packet
// without the barrier bit), and are its workgroups dispatched only after ALL of the predecessor's?  And the same question for a
// second stream with no event in between.  A = 4096 workgroups that each spin ~20 us (longer than the dispatch of the grid),
// B = 512 workgroups that stamp their start.  Stamps: wall_clock64 (100 MHz).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void kernel_a(unsigned long long* start, unsigned long long* stop, unsigned spin_ticks, unsigned* flags, unsigned stamp) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        stop[blockIdx.x] = wall_clock64();
        __hip_atomic_store(flags + blockIdx.x, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// B: workgroup b waits for the flags of A's workgroups [8b, 8b+8) (bounded), stamps when it started and when it saw them
__global__ __launch_bounds__(256) void kernel_b(unsigned long long* start, unsigned long long* seen, const unsigned* flags, unsigned stamp, int per) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    if (threadIdx.x < per) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + blockIdx.x * per + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != stamp && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (threadIdx.x == 0) seen[blockIdx.x] = wall_clock64();
}

int main() {
    const int NA = 4096, NB = 512, PER = NA / NB;
    unsigned long long *a0, *a1, *b0, *b1; unsigned* flags;
    CK(hipMalloc(&a0, NA * 8)); CK(hipMalloc(&a1, NA * 8)); CK(hipMalloc(&b0, NB * 8)); CK(hipMalloc(&b1, NB * 8)); CK(hipMalloc(&flags, NA * 4));
    CK(hipMemset(flags, 0, NA * 4));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<unsigned long long> ha0(NA), ha1(NA), hb0(NB), hb1(NB);
    unsigned stamp = 0;
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 3; ++rep) {
        ++stamp;
        hipLaunchKernelGGL(kernel_a, dim3(NA), dim3(256), 0, s, a0, a1, 2000u /* 20 us */, flags, stamp);
        if (mode == 0)      hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(256), 0, s, b0, b1, flags, stamp, PER);                                   // ordered
        else if (mode == 1) hipExtLaunchKernelGGL(kernel_b, dim3(NB), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, b0, b1, flags, stamp, PER);
        else                hipLaunchKernelGGL(kernel_b, dim3(NB), dim3(256), 0, s2, b0, b1, flags, stamp, PER);                                  // second stream, no event
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(ha0.data(), a0, NA * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ha1.data(), a1, NA * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb0.data(), b0, NB * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb1.data(), b1, NB * 8, hipMemcpyDeviceToHost));
        const unsigned long long t0 = *std::min_element(ha0.begin(), ha0.end());
        const unsigned long long a_last_start = *std::max_element(ha0.begin(), ha0.end()), a_end = *std::max_element(ha1.begin(), ha1.end());
        const unsigned long long b_first = *std::min_element(hb0.begin(), hb0.end()), b_last = *std::max_element(hb0.begin(), hb0.end());
        const unsigned long long b_seen_last = *std::max_element(hb1.begin(), hb1.end());
        int b_before_a_dispatched = 0;
        for (int i = 0; i < NB; ++i) b_before_a_dispatched += hb0[i] < a_last_start;
        std::printf("%s rep %d: A last start %.2f us, A end %.2f | B first start %.2f, last start %.2f, last seen %.2f | B workgroups started before A's last: %d\n",
                    mode == 0 ? "ordered   " : mode == 1 ? "any-order " : "2nd stream", rep, (a_last_start - t0) / 100.0, (a_end - t0) / 100.0,
                    (b_first - t0) / 100.0, (b_last - t0) / 100.0, (b_seen_last - t0) / 100.0, b_before_a_dispatched);
    }
    return 0;
}
