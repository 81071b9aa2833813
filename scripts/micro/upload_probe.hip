// Micro-benchmark behind DESIGN.md section 4 (the drop-in callback path): what is the fastest way to get a host-resident
// PCL cloud (pcl::PointXYZINormal, 48-byte records, x y z first) into the float4 scan buffer in HBM?
//   A  host pack 48 -> 16 bytes into pinned staging (scalar / AVX2), then ONE copy of the packed scan (today's lio_scan_set)
//   B  the same in chunks: the copy of chunk i travels while chunk i+1 is packed
//   C  the raw records through the copy engine (registered memory, 3x the bytes), repacked by a kernel
//   D  a kernel that PULLS the records over PCIe itself (registered + mapped memory): one lane per record (16 of 48 bytes),
//      or every lane 16 consecutive bytes (whole lines, perfectly coalesced)
//   E  hipMemcpy2DAsync width 16 / pitch 48
// plus the one-off cost of hipHostRegister.  Times are host wall clock from "cloud ready" to "scan resident" (a one-wavefront
// kernel publishes a word into host-mapped memory behind the work; the host polls it), median of REPS.
//   hipcc --offload-arch=gfx950 -O2 -mavx2 scripts/micro/upload_probe.hip -o /tmp/upload_probe && /tmp/upload_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void publish(unsigned* out, unsigned seq) { if (threadIdx.x == 0) __hip_atomic_store(out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// one lane per record: 16 of its 48 bytes
__global__ __launch_bounds__(256) void pull_records(const float4* __restrict__ src48, float4* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 v = src48[3 * (size_t)i]; v.w = 0.0f; dst[i] = v; }
}
// every lane 16 consecutive bytes; lanes 0, 3, 6 ... hold the head of a record
__global__ __launch_bounds__(256) void pull_lines(const float4* __restrict__ src48, float4* __restrict__ dst, int n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * (size_t)n) { float4 v = src48[i]; if (i % 3 == 0) { v.w = 0.0f; dst[i / 3] = v; } }
}
// device-side repack of raw records already in HBM
__global__ __launch_bounds__(256) void repack(const float4* __restrict__ raw, float4* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 v = raw[3 * (size_t)i]; v.w = 0.0f; dst[i] = v; }
}

static void pack_scalar(const char* src, float4* dst, int n) {
    for (int i = 0; i < n; ++i) { const float* p = reinterpret_cast<const float*>(src + (size_t)i * 48); dst[i] = make_float4(p[0], p[1], p[2], 0.0f); }
}
static void pack_sse(const char* src, float4* dst, int n) {
    const __m128 mask = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1));
    for (int i = 0; i < n; ++i) {
        if ((i & 7) == 0) _mm_prefetch(src + (size_t)(i + 16) * 48, _MM_HINT_T0);
        const __m128 v = _mm_and_ps(_mm_loadu_ps(reinterpret_cast<const float*>(src + (size_t)i * 48)), mask);
        _mm_stream_ps(reinterpret_cast<float*>(dst + i), v);
    }
    _mm_sfence();
}
static void trash_caches() {
    static std::vector<char> big(768u << 20, 1);
    volatile long s = 0;
    for (size_t i = 0; i < big.size(); i += 64) s += big[i];
}

int main() {
    const int REPS = 15;
    unsigned* h_flag; unsigned* d_flag;
    CK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&d_flag, h_flag, 0));
    *h_flag = 0;
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned seq = 0;
    auto wait_pub = [&](hipStream_t st) { ++seq; hipLaunchKernelGGL(publish, 1, 64, 0, st, d_flag, seq); while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) {} };

    for (int n : {8600, 24000, 100000}) {
        const size_t raw_bytes = (size_t)n * 48;
        char* cloud = (char*)aligned_alloc(4096, (raw_bytes + 4095) / 4096 * 4096);
        for (int i = 0; i < n; ++i) { float* p = (float*)(cloud + (size_t)i * 48); for (int k = 0; k < 12; ++k) p[k] = (float)(i * 12 + k); }
        float4 *h_stage, *d_body, *d_raw;
        CK(hipHostMalloc((void**)&h_stage, (size_t)n * 16, hipHostMallocDefault));
        CK(hipMalloc((void**)&d_body, (size_t)n * 16));
        CK(hipMalloc((void**)&d_raw, raw_bytes));
        std::printf("---- n = %d points (%.2f MB raw, %.2f MB packed)\n", n, raw_bytes / 1e6, n * 16 / 1e6);
        // one-off: hipHostRegister
        {
            double t0 = now_us();
            CK(hipHostRegister(cloud, raw_bytes, hipHostRegisterMapped));
            double t1 = now_us();
            std::printf("hipHostRegister(mapped)            %8.1f us (one-off per buffer)\n", t1 - t0);
        }
        float4* d_cloud_alias = nullptr;
        CK(hipHostGetDevicePointer((void**)&d_cloud_alias, cloud, 0));
        auto report = [&](const char* name, std::vector<double>& v) { std::sort(v.begin(), v.end()); std::printf("%-58s median %8.1f us  min %8.1f\n", name, v[v.size() / 2], v[0]); };
        auto verify = [&](const char* what) {
            std::vector<float4> back(n);
            (void)hipMemcpy(back.data(), d_body, (size_t)n * 16, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; i += 997) if (back[i].x != (float)(i * 12) || back[i].z != (float)(i * 12 + 2)) { std::printf("  !! %s: wrong data at %d\n", what, i); return; }
            (void)hipMemset(d_body, 0, (size_t)n * 16);
        };
        for (int cold = 0; cold < 2; ++cold) {
            std::printf("  (cloud %s in the CPU caches)\n", cold ? "COLD: evicted before every repetition" : "as the last writer left it: warm");
            std::vector<double> v;
            // A: pack + one copy
            for (int variant = 0; variant < 2; ++variant) {
                v.clear();
                std::vector<double> vp;
                for (int r = 0; r < REPS; ++r) {
                    if (cold) trash_caches();
                    double t0 = now_us();
                    if (variant == 0) pack_scalar(cloud, h_stage, n); else pack_sse(cloud, h_stage, n);
                    double t1 = now_us();
                    (void)hipMemcpyAsync(d_body, h_stage, (size_t)n * 16, hipMemcpyHostToDevice, s);
                    wait_pub(s);
                    v.push_back(now_us() - t0); vp.push_back(t1 - t0);
                }
                report(variant ? "A2 pack (SSE, streaming stores) + one copy" : "A1 pack (scalar) + one copy", v);
                report(variant ? "     of which the pack" : "     of which the pack", vp);
                verify("A");
            }
            // B: chunked
            for (int chunks : {4, 8}) {
                v.clear();
                for (int r = 0; r < REPS; ++r) {
                    if (cold) trash_caches();
                    double t0 = now_us();
                    for (int c = 0; c < chunks; ++c) {
                        const int a = (int)((long long)n * c / chunks), b = (int)((long long)n * (c + 1) / chunks);
                        pack_sse(cloud + (size_t)a * 48, h_stage + a, b - a);
                        (void)hipMemcpyAsync(d_body + a, h_stage + a, (size_t)(b - a) * 16, hipMemcpyHostToDevice, s);
                    }
                    wait_pub(s);
                    v.push_back(now_us() - t0);
                }
                char nm[96]; std::snprintf(nm, sizeof nm, "B  pack + copy in %d chunks (copy behind the next pack)", chunks);
                report(nm, v);
                verify("B");
            }
            // C: raw through the copy engine + repack kernel
            v.clear();
            for (int r = 0; r < REPS; ++r) {
                if (cold) trash_caches();
                double t0 = now_us();
                (void)hipMemcpyAsync(d_raw, cloud, raw_bytes, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(repack, (n + 255) / 256, 256, 0, s, d_raw, d_body, n);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            report("C  raw records by the copy engine + repack kernel", v);
            verify("C");
            // D: kernel pulls
            v.clear();
            for (int r = 0; r < REPS; ++r) {
                if (cold) trash_caches();
                double t0 = now_us();
                hipLaunchKernelGGL(pull_records, (n + 255) / 256, 256, 0, s, d_cloud_alias, d_body, n);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            report("D1 kernel pulls 16 of every 48 bytes (lane per record)", v);
            verify("D1");
            v.clear();
            for (int r = 0; r < REPS; ++r) {
                if (cold) trash_caches();
                double t0 = now_us();
                hipLaunchKernelGGL(pull_lines, (3 * n + 255) / 256, 256, 0, s, d_cloud_alias, d_body, n);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            report("D2 kernel pulls whole lines (16 consecutive bytes a lane)", v);
            verify("D2");
            // D3: half by the kernel, half packed by the host meanwhile
            v.clear();
            for (int r = 0; r < REPS; ++r) {
                if (cold) trash_caches();
                const int half = n / 2;
                double t0 = now_us();
                hipLaunchKernelGGL(pull_lines, (3 * half + 255) / 256, 256, 0, s, d_cloud_alias, d_body, half);
                pack_sse(cloud + (size_t)half * 48, h_stage + half, n - half);
                (void)hipMemcpyAsync(d_body + half, h_stage + half, (size_t)(n - half) * 16, hipMemcpyHostToDevice, s2);
                wait_pub(s2);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            report("D3 first half pulled by a kernel, second half packed + copied", v);
            verify("D3");
            // E: 2-D copy
            v.clear();
            for (int r = 0; r < REPS; ++r) {
                if (cold) trash_caches();
                double t0 = now_us();
                (void)hipMemcpy2DAsync(d_body, 16, cloud, 48, 16, n, hipMemcpyHostToDevice, s);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            report("E  hipMemcpy2DAsync width 16 pitch 48", v);
            verify("E");
        }
        // plain copies for the PCIe rate
        for (size_t bytes : {(size_t)n * 12, (size_t)n * 16, raw_bytes}) {
            std::vector<double> v;
            for (int r = 0; r < REPS; ++r) {
                double t0 = now_us();
                (void)hipMemcpyAsync(d_raw, cloud, bytes, hipMemcpyHostToDevice, s);
                wait_pub(s);
                v.push_back(now_us() - t0);
            }
            char nm[96]; std::snprintf(nm, sizeof nm, "   plain copy of %.2f MB (registered memory)", bytes / 1e6);
            report(nm, v);
        }
        CK(hipHostUnregister(cloud));
        (void)hipFree(d_body); (void)hipFree(d_raw); (void)hipHostFree(h_stage); std::free(cloud);
    }
    return 0;
}
