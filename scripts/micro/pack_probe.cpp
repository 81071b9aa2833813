// Micro-probe (host only): how fast can ONE host thread turn the front of a 48-byte-stride cloud into 16-byte records?
// (assist_pack, lio_pull.hip: the faster it is, the larger the share of a scan that crosses PCIe as 16 instead of 48 bytes.)
// Variants: plain 16-byte load/mask/store; + software prefetch; unrolled x4; 32-byte stores of two records; streaming stores;
// two threads.  Warm = the same cloud again and again (a node's voxel filter has just written it); cold = 8 clouds in turn behind
// a 512 MB sweep.   g++ -O2 -mavx2 -pthread scripts/micro/pack_probe.cpp -o scripts/micro/pack_probe && scripts/micro/pack_probe
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now_ns() { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int N = 100000, STRIDE = 48;
static const __m128 keep() { return _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1)); }
static void v_plain(const char* s, float* d, int a, int b) {
    const __m128 k = keep();
    for (int i = a; i < b; ++i) _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), k));
}
template <int PF> static void v_prefetch(const char* s, float* d, int a, int b) {
    const __m128 k = keep();
    for (int i = a; i < b; ++i) {
        _mm_prefetch(s + (size_t)(i + PF) * STRIDE, _MM_HINT_T0);
        _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), k));
    }
}
static void v_unroll4(const char* s, float* d, int a, int b) {
    const __m128 k = keep();
    int i = a;
    for (; i + 4 <= b; i += 4) {
        const __m128 r0 = _mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), r1 = _mm_loadu_ps((const float*)(s + (size_t)(i + 1) * STRIDE));
        const __m128 r2 = _mm_loadu_ps((const float*)(s + (size_t)(i + 2) * STRIDE)), r3 = _mm_loadu_ps((const float*)(s + (size_t)(i + 3) * STRIDE));
        _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(r0, k)); _mm_store_ps(d + 4 * (size_t)i + 4, _mm_and_ps(r1, k));
        _mm_store_ps(d + 4 * (size_t)i + 8, _mm_and_ps(r2, k)); _mm_store_ps(d + 4 * (size_t)i + 12, _mm_and_ps(r3, k));
    }
    for (; i < b; ++i) _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), k));
}
static void v_avx2(const char* s, float* d, int a, int b) {
    const __m256 k = _mm256_castsi256_ps(_mm256_set_epi32(0, -1, -1, -1, 0, -1, -1, -1));
    int i = a;
    for (; i + 4 <= b; i += 4) {
        const __m256 p0 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 1) * STRIDE), (const float*)(s + (size_t)i * STRIDE));
        const __m256 p1 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 3) * STRIDE), (const float*)(s + (size_t)(i + 2) * STRIDE));
        _mm256_store_ps(d + 4 * (size_t)i, _mm256_and_ps(p0, k)); _mm256_store_ps(d + 4 * (size_t)i + 8, _mm256_and_ps(p1, k));
    }
    for (; i < b; ++i) _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), keep()));
}
static void v_stream(const char* s, float* d, int a, int b) {
    const __m256 k = _mm256_castsi256_ps(_mm256_set_epi32(0, -1, -1, -1, 0, -1, -1, -1));
    int i = a;
    for (; i + 4 <= b; i += 4) {
        const __m256 p0 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 1) * STRIDE), (const float*)(s + (size_t)i * STRIDE));
        const __m256 p1 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 3) * STRIDE), (const float*)(s + (size_t)(i + 2) * STRIDE));
        _mm256_stream_ps(d + 4 * (size_t)i, _mm256_and_ps(p0, k)); _mm256_stream_ps(d + 4 * (size_t)i + 8, _mm256_and_ps(p1, k));
    }
    for (; i < b; ++i) _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), keep()));
    _mm_sfence();
}
template <int PF> static void v_stream_pf(const char* s, float* d, int a, int b) {
    const __m256 k = _mm256_castsi256_ps(_mm256_set_epi32(0, -1, -1, -1, 0, -1, -1, -1));
    int i = a;
    for (; i + 4 <= b; i += 4) {
        _mm_prefetch(s + (size_t)(i + PF) * STRIDE, _MM_HINT_T0); _mm_prefetch(s + (size_t)(i + PF) * STRIDE + 64, _MM_HINT_T0); _mm_prefetch(s + (size_t)(i + PF) * STRIDE + 128, _MM_HINT_T0);
        const __m256 p0 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 1) * STRIDE), (const float*)(s + (size_t)i * STRIDE));
        const __m256 p1 = _mm256_loadu2_m128((const float*)(s + (size_t)(i + 3) * STRIDE), (const float*)(s + (size_t)(i + 2) * STRIDE));
        _mm256_stream_ps(d + 4 * (size_t)i, _mm256_and_ps(p0, k)); _mm256_stream_ps(d + 4 * (size_t)i + 8, _mm256_and_ps(p1, k));
    }
    for (; i < b; ++i) _mm_store_ps(d + 4 * (size_t)i, _mm_and_ps(_mm_loadu_ps((const float*)(s + (size_t)i * STRIDE)), keep()));
    _mm_sfence();
}
typedef void (*Fn)(const char*, float*, int, int);
int main() {
    std::vector<char*> clouds;
    for (int c = 0; c < 8; ++c) { char* p = (char*)aligned_alloc(64, (size_t)(N + 64) * STRIDE); memset(p, c + 1, (size_t)(N + 64) * STRIDE); clouds.push_back(p); }
    float* dst = (float*)aligned_alloc(64, (size_t)N * 16 + 4096);
    memset(dst, 0, (size_t)N * 16);
    const size_t SWEEP = 512u << 20; char* sweep = (char*)aligned_alloc(64, SWEEP); memset(sweep, 1, SWEEP);
    struct V { const char* name; Fn f; } vs[] = {{"plain", v_plain}, {"prefetch +16", v_prefetch<16>}, {"prefetch +48", v_prefetch<48>}, {"unroll x4", v_unroll4},
                                                 {"avx2 two records", v_avx2}, {"avx2 + streaming stores", v_stream}, {"streaming + prefetch +32", v_stream_pf<32>}, {"streaming + prefetch +96", v_stream_pf<96>}};
    for (auto& v : vs) {
        double warm = 1e30, cold = 0;
        for (int r = 0; r < 30; ++r) { const double t0 = now_ns(); v.f(clouds[0], dst, 0, N); warm = std::min(warm, now_ns() - t0); }
        int nc = 0;
        for (int r = 0; r < 6; ++r) {
            volatile char sink = 0; for (size_t i = 0; i < SWEEP; i += 64) sink += sweep[i];
            const double t0 = now_ns(); v.f(clouds[r % 8], dst, 0, N); cold += now_ns() - t0; ++nc;
        }
        std::printf("%-28s warm %.3f ns/record (%.1f us per 100 k), cold %.3f ns/record\n", v.name, warm / N, warm * 1e-3, cold / nc / N);
    }
    // two threads: a helper that spins on a flag takes the second half
    {
        std::atomic<int> go{0}, done{0}; std::atomic<bool> quit{false};
        const char* src = clouds[0];
        std::thread helper([&] { int seen = 0; while (!quit.load(std::memory_order_relaxed)) { const int g = go.load(std::memory_order_acquire); if (g != seen) { seen = g; v_avx2(src, dst, N / 2, N); done.store(g, std::memory_order_release); } } });
        double best = 1e30;
        for (int r = 1; r <= 30; ++r) { const double t0 = now_ns(); go.store(r, std::memory_order_release); v_avx2(src, dst, 0, N / 2); while (done.load(std::memory_order_acquire) != r) {} best = std::min(best, now_ns() - t0); }
        quit = true; helper.join();
        std::printf("%-28s warm %.3f ns/record (%.1f us per 100 k)\n", "two threads (spinning helper)", best / N, best * 1e-3);
    }
    return 0;
}
