// Micro-benchmark behind DESIGN.md section 4: is a hipGraph of the four kernels of a search pass cheaper to start
// than four direct launches on one stream?  Each "pass" = 4 short dependent kernels, the last one publishes a
// sequence number into host-mapped memory which the host polls (the shape of lio_measure).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/graph_vs_direct.hip -o /tmp/gvd && /tmp/gvd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(float* buf, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = buf[i];
        for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
        buf[i] = v;
    }
}
__global__ void publish(float* buf, int n, volatile unsigned* seq_src, unsigned* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] += 1.0f;
    if (i == 0) {
        __threadfence_system();
        __hip_atomic_store(out, *seq_src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int n = 100000, iters = 2000;
    float* buf; CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMemset(buf, 0, n * sizeof(float)));
    unsigned *h_seq, *h_out, *d_seq, *d_out;
    CK(hipHostMalloc(&h_seq, 64, hipHostMallocMapped)); CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&d_seq, h_seq, 0)); CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    *h_seq = 0; *h_out = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const dim3 grid((n + 255) / 256), block(256);
    auto direct = [&]() {
        hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 8);
        hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 64);
        hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 8);
        hipLaunchKernelGGL(publish, grid, block, 0, s, buf, n, (volatile unsigned*)d_seq, d_out);
    };
    // graph of the same four launches; the sequence number travels through mapped memory, so no node parameter
    // changes between launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    direct();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    unsigned seq = 0;
    auto wait = [&](unsigned v) { while (__atomic_load_n(h_out, __ATOMIC_ACQUIRE) != v) { } };
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> t(iters);
        for (int it = -200; it < iters; ++it) {
            const double t0 = now_us();
            __atomic_store_n(h_seq, ++seq, __ATOMIC_RELEASE);
            if (mode == 0) direct(); else CK(hipGraphLaunch(ge, s));
            wait(seq);
            if (it >= 0) t[it] = now_us() - t0;
        }
        std::sort(t.begin(), t.end());
        std::printf("%s: median %.2f us, p10 %.2f, p90 %.2f per pass of 4 kernels (launch -> host sees the result)\n",
                    mode == 0 ? "direct launches" : "hipGraphLaunch ", t[iters / 2], t[iters / 10], t[iters * 9 / 10]);
    }
    CK(hipStreamSynchronize(s));
    return 0;
}
