// Micro-benchmark: where should the host's word to an armed gate kernel live?  Today (lio_pass.hip, gate_write) the mail slot is in
// host-mapped memory and the one-wavefront gate polls it over PCIe (a read round trip per poll).  Alternative: the slot in
// FINE-GRAINED DEVICE memory that the CPU writes through the PCIe BAR -- the gate then polls local memory and the word travels once.
//   1  can the CPU store into hipExtMallocWithFlags(hipDeviceMallocFinegrained) memory at all on this box?  (forked: a fault kills
//      the child only)
//   2  host word -> gate sees it -> work kernel's first workgroup -> host sees its word: median latency, mail in host memory vs
//      mail in device memory
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mail_in_vram.hip -o scripts/micro/mail_in_vram && scripts/micro/mail_in_vram
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Mail { unsigned long long w[8]; };      // [0..6] pose words, [7] sequence number (written last)

__global__ __launch_bounds__(256) void work(float* buf, int n, unsigned seq, unsigned* out, const double* pose_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) __hip_atomic_store(out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i < n) { float v = buf[i] + (float)pose_dev[i & 7]; for (int k = 0; k < 64; ++k) v = v * 1.0001f + 0.5f; buf[i] = v; }
}
__global__ __launch_bounds__(64) void gate(const Mail* mail, unsigned long long want, unsigned* armed, double* pose_dev, unsigned max_polls) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(armed, (unsigned)want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long v = 0;
    unsigned polls = 0;
    while (polls++ < max_polls) {
        if (lane < 8) v = __hip_atomic_load(&mail->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__shfl(v, 7, 64) == want) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane < 8) pose_dev[lane] = lane < 7 ? __longlong_as_double((long long)v) : 0.0;
}

int main() {
    // ---- 1: CPU stores into fine-grained device memory (in a child process)
    int works = 0;
    {
        int fd[2];
        if (pipe(fd) != 0) return 1;
        const pid_t pid = fork();
        if (pid == 0) {
            Mail* m = nullptr;
            if (hipExtMallocWithFlags((void**)&m, 4096, hipDeviceMallocFinegrained) != hipSuccess) _exit(3);
            volatile unsigned long long* p = m->w;
            p[0] = 0x1234ull;                                   // faults here when the BAR does not map it
            const unsigned long long back = p[0];
            unsigned long long dev = 0;
            if (hipMemcpy(&dev, m, 8, hipMemcpyDeviceToHost) != hipSuccess) _exit(4);
            const char ok = (back == 0x1234ull && dev == 0x1234ull) ? 1 : 0;
            if (write(fd[1], &ok, 1) != 1) _exit(5);
            _exit(0);
        }
        int status = 0;
        waitpid(pid, &status, 0);
        char ok = 0;
        close(fd[1]);
        if (WIFEXITED(status) && WEXITSTATUS(status) == 0 && read(fd[0], &ok, 1) == 1) works = ok;
        std::printf("CPU store into hipDeviceMallocFinegrained memory: %s (child %s %d)\n", works ? "WORKS (device sees the value)" : "does not work",
                    WIFSIGNALED(status) ? "killed by signal" : "exit code", WIFSIGNALED(status) ? WTERMSIG(status) : WEXITSTATUS(status));
    }
    const int n = 800000, iters = 1500;
    float* buf; CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMemset(buf, 0, n * sizeof(float)));
    double* pose_dev; CK(hipMalloc(&pose_dev, 64));
    Mail* h_mail; unsigned *h_out, *h_armed;
    CK(hipHostMalloc(&h_mail, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_armed, 64, hipHostMallocMapped | hipHostMallocCoherent));
    Mail* d_mail_host; unsigned *d_out, *d_armed;
    CK(hipHostGetDevicePointer((void**)&d_mail_host, h_mail, 0)); CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
    CK(hipHostGetDevicePointer((void**)&d_armed, h_armed, 0));
    Mail* vram_mail = nullptr;
    if (works) CK(hipExtMallocWithFlags((void**)&vram_mail, 4096, hipDeviceMallocFinegrained));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const dim3 grid((n + 255) / 256), block(256);
    unsigned seq = 0;
    for (int mode = 0; mode < (works ? 2 : 1); ++mode) {
        Mail* cpu_side = mode == 0 ? h_mail : vram_mail;         // where the CPU writes
        const Mail* gpu_side = mode == 0 ? d_mail_host : vram_mail;
        for (int k = 0; k < 8; ++k) ((volatile unsigned long long*)cpu_side->w)[k] = 0;
        for (int behind = 0; behind < 2; ++behind) {
            std::vector<double> t(iters);
            for (int it = -100; it < iters; ++it) {
                ++seq;
                if (behind) hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, 0u, d_armed + 8, (const double*)pose_dev);   // a pass still running in front
                hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, s, gpu_side, (unsigned long long)seq, d_armed, pose_dev, 4000000u);
                hipLaunchKernelGGL(work, grid, block, 0, s, buf, n, seq, d_out, (const double*)pose_dev);
                while (__atomic_load_n(h_armed, __ATOMIC_ACQUIRE) != seq) { }
                const double tw = now_us();
                while (now_us() - tw < 5.0) { }
                volatile unsigned long long* w = cpu_side->w;
                const double t0 = now_us();
                for (int k = 0; k < 7; ++k) w[k] = (unsigned long long)seq + k;
                __atomic_store_n(&cpu_side->w[7], (unsigned long long)seq, __ATOMIC_RELEASE);
                while (__atomic_load_n(h_out, __ATOMIC_ACQUIRE) != seq) { }
                const double t1 = now_us();
                if (it >= 0) t[it] = t1 - t0;
                CK(hipStreamSynchronize(s));
            }
            std::sort(t.begin(), t.end());
            std::printf("mail in %s, %s: host word -> first workgroup of the pass -> host: median %.2f us  p10 %.2f  p90 %.2f\n",
                        mode == 0 ? "HOST memory (gate polls over PCIe)" : "DEVICE memory (CPU writes through the BAR)",
                        behind ? "gate behind a running kernel" : "idle queue", t[iters / 2], t[iters / 10], t[iters * 9 / 10]);
        }
    }
    return 0;
}
