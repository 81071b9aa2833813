// Micro-benchmark behind DESIGN.md section 4: what would PRE-ENQUEUED passes buy?  A pass of the iterated update starts
// when the host has the new pose; today that is a kernel launch (API call + doorbell + dispatch).  Alternative: the
// next pass's kernels are enqueued while the current one runs, behind a one-wavefront GATE kernel that polls a word in
// host memory; the host then only stores the pose and the word.
//   A  direct: t0, launch work kernel (grid of a k-NN launch), its first workgroup publishes into host-mapped memory,
//              host sees it: t1
//   B  gated:  [gate][work] already enqueued and the gate resident (it has told the host so); t0, host stores the
//              word, gate sees it and ends, work starts, publishes, host sees it: t1
// Both end with the same device -> host store, so B - A is what the gate saves (or costs) per pass.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/gate_latency.hip -o /tmp/gate && /tmp/gate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctrl {                                   // one 64-byte line in coherent host memory
    double pose[7];
    unsigned long long go;                      // sequence number of the pass that may start
};

// the first workgroup publishes at its START (what we time is the way to the first wavefront)
__global__ __launch_bounds__(256) void work(float* buf, int n, unsigned seq, unsigned* out, const double* pose_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) __hip_atomic_store(out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i < n) {
        float v = buf[i] + (pose_dev ? (float)pose_dev[i & 7] : 0.0f);
        for (int k = 0; k < 64; ++k) v = v * 1.0001f + 0.5f;
        buf[i] = v;
    }
}
// one wavefront: tell the host we are resident, poll the line (bounded), hand the pose to device memory
__global__ __launch_bounds__(64) void gate(const Ctrl* ctrl, unsigned long long want, unsigned* armed, double* pose_dev, unsigned* state,
                                           unsigned max_polls) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(armed, (unsigned)want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* line = reinterpret_cast<const unsigned long long*>(ctrl);
    unsigned long long v = 0;
    unsigned polls = 0;
    bool ok = false;
    while (polls++ < max_polls) {
        if (lane < 8) v = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the whole line, one request
        const unsigned long long g = __shfl(v, 7, 64);
        if (g == want) { ok = true; break; }
    }
    if (lane < 7) pose_dev[lane] = __longlong_as_double((long long)v);
    if (lane == 7) { pose_dev[7] = 0.0; *state = ok ? 1u : 2u; }
}

// D: no gate kernel -- the pass's own kernel is dispatched early, every workgroup spins on a device word that workgroup 0 sets
// after it has seen the host's word (and copied the pose); the first workgroup publishes when released
__global__ __launch_bounds__(256) void work_self_gated(float* buf, int n, unsigned seq, unsigned* out, const Ctrl* ctrl, unsigned* armed,
                                                      double* pose_dev, unsigned* go_dev, unsigned max_polls) {
    __shared__ unsigned s_ok;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (blockIdx.x == 0) {
            if (lane == 0) __hip_atomic_store(armed, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long* line = reinterpret_cast<const unsigned long long*>(ctrl);
            unsigned long long v = 0;
            unsigned polls = 0;
            while (polls++ < max_polls) {
                if (lane < 8) v = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (__shfl(v, 7, 64) == (unsigned long long)seq) break;
            }
            if (lane < 7) __hip_atomic_store(reinterpret_cast<unsigned long long*>(pose_dev) + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(go_dev, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(go_dev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq && polls++ < max_polls) __builtin_amdgcn_s_sleep(8);
            s_ok = 1u;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) __hip_atomic_store(out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i < n) {
        const double pz = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(pose_dev) + (i & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float v = buf[i] + (float)pz + (float)s_ok;
        for (int k = 0; k < 64; ++k) v = v * 1.0001f + 0.5f;
        buf[i] = v;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int n = 800000, iters = 1500;        // 3125 workgroups of 256: the grid of a 100 k-query k-NN launch
    float* buf; CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMemset(buf, 0, n * sizeof(float)));
    double* pose_dev; CK(hipMalloc(&pose_dev, 64));
    unsigned* state; CK(hipMalloc(&state, 64));
    Ctrl* h_ctrl; unsigned *h_out, *h_armed;
    CK(hipHostMalloc(&h_ctrl, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_armed, 64, hipHostMallocMapped | hipHostMallocCoherent));
    Ctrl* d_ctrl; unsigned *d_out, *d_armed;
    CK(hipHostGetDevicePointer((void**)&d_ctrl, h_ctrl, 0)); CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    CK(hipHostGetDevicePointer((void**)&d_armed, h_armed, 0));
    *h_out = 0; *h_armed = 0; h_ctrl->go = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const dim3 grid((n + 255) / 256), block(256);
    auto wait_out = [&](unsigned v) { while (__atomic_load_n(h_out, __ATOMIC_ACQUIRE) != v) { } };
    unsigned seq = 0;
    unsigned* go_dev; CK(hipMalloc(&go_dev, 64)); CK(hipMemset(go_dev, 0, 64));
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<double> t(iters);
        for (int it = -100; it < iters; ++it) {
            ++seq;
            double t0;
            if (mode == 0) {
                t0 = now_us();
                hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, seq, d_out, (const double*)nullptr);
            } else {
                // mode 1: the previous kernel is long gone (the queue is idle when the gate is enqueued)
                // mode 2: a 30 us kernel is still running in front of the gate (a pass in flight), wait for it, then 5 us of "host algebra"
                if (mode == 2) hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 0u, d_armed + 8, (const double*)nullptr);
                if (mode >= 3) {
                    if (mode == 4) hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 0u, d_armed + 8, (const double*)nullptr);
                    hipLaunchKernelGGL(work_self_gated, grid, block, 0, s, buf, n, seq, d_out, (const Ctrl*)d_ctrl, d_armed, pose_dev, go_dev, 400000u);
                } else {
                hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, s, (const Ctrl*)d_ctrl, (unsigned long long)seq, d_armed, pose_dev, state, 400000u);
                hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, seq, d_out, (const double*)pose_dev);
                }
                while (__atomic_load_n(h_armed, __ATOMIC_ACQUIRE) != seq) { }           // the gate is resident
                const double tw = now_us();
                while (now_us() - tw < 5.0) { }                                          // the host's own work
                for (int k = 0; k < 7; ++k) h_ctrl->pose[k] = (double)seq + k;
                t0 = now_us();
                __atomic_store_n(&h_ctrl->go, (unsigned long long)seq, __ATOMIC_RELEASE);
            }
            wait_out(seq);
            const double t1 = now_us();
            if (it >= 0) t[it] = t1 - t0;
            CK(hipStreamSynchronize(s));
        }
        std::sort(t.begin(), t.end());
        std::printf("%s: median %.2f us  p10 %.2f  p90 %.2f\n", mode == 0 ? "A direct launch -> first workgroup -> host" :
                    mode == 1 ? "B gate (idle queue): host word -> first workgroup -> host" :
                    mode == 2 ? "C gate (behind a running kernel): host word -> first workgroup -> host" :
                    mode == 3 ? "D self-gated kernel (idle queue): host word -> released workgroup -> host" :
                                "E self-gated kernel (behind a running kernel): host word -> released workgroup -> host", t[iters / 2], t[iters / 10], t[iters * 9 / 10]);
    }
    unsigned st = 0; CK(hipMemcpy(&st, state, 4, hipMemcpyDeviceToHost));
    std::printf("last gate state %u (1 = released by the host, 2 = ran out)\n", st);
    return 0;
}
