// Host-side check of the compressed batch keys (csrc/lio_mapadd.h): frame_key / frame_decode are inverse, the all-ones key belongs to
// the spare coordinates only, and the ORDER of the keys does not depend on the frame (two boxes around the same points sort them alike).
// Compiled with hipcc (host code only) by tests/test_host_math.py.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "lio_mapadd.h"
using namespace lio;
int main() {
    srand(5);
    for (int t = 0; t < 1000000; ++t) {
        KeyFrame f; f.ox = (rand() % 2000 - 1000) & ~31; f.oy = (rand() % 2000 - 1000) & ~31; f.oz = (rand() % 100 - 50) & ~31;
        f.bx = 1 + rand() % 20; f.by = 1 + rand() % 20; f.bz = 1 + rand() % 12;
        int x = f.ox + (int)(((long long)rand() * rand()) % (1ll << f.bx)), y = f.oy + (int)(((long long)rand() * rand()) % (1ll << f.by)), z = f.oz + (int)(((long long)rand() * rand()) % (1ll << f.bz));
        uint64_t k = frame_key(f, x, y, z);
        if (k >> frame_bits(f)) { printf("key too wide\n"); return 1; }
        int a, b, c; frame_decode(f, k, a, b, c);
        if (a != x || b != y || c != z) { printf("mismatch\n"); return 1; }
        int mx = f.ox + (1 << f.bx) - 1, my = f.oy + (1 << f.by) - 1, mz = f.oz + (1 << f.bz) - 1;
        const uint64_t ones = (1ull << frame_bits(f)) - 1ull;
        if ((k == ones) != (x == mx && y == my && z == mz)) { printf("junk key clash\n"); return 1; }
    }
    // order independent of the frame: random points, two boxes that hold them
    for (int t = 0; t < 300; ++t) {
        std::vector<int> px, py, pz; int n = 200;
        int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
        for (int i = 0; i < n; ++i) { int p[3] = {rand() % 700 - 300, rand() % 900 - 500, rand() % 40 - 10}; px.push_back(p[0]); py.push_back(p[1]); pz.push_back(p[2]); for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
        int lo2[3] = {lo[0] - rand() % 500, lo[1] - rand() % 70, lo[2] - rand() % 9}, hi2[3] = {hi[0] + rand() % 900, hi[1] + rand() % 5, hi[2] + rand() % 300};
        KeyFrame f1 = frame_of(lo, hi), f2 = frame_of(lo2, hi2);
        std::vector<int> o1(n), o2(n);
        for (int i = 0; i < n; ++i) o1[i] = o2[i] = i;
        std::stable_sort(o1.begin(), o1.end(), [&](int a, int b) { return frame_key(f1, px[a], py[a], pz[a]) < frame_key(f1, px[b], py[b], pz[b]); });
        std::stable_sort(o2.begin(), o2.end(), [&](int a, int b) { return frame_key(f2, px[a], py[a], pz[a]) < frame_key(f2, px[b], py[b], pz[b]); });
        if (o1 != o2) { printf("order depends on the frame (t %d)\n", t); return 1; }
        for (int i = 0; i < n; ++i) if (!frame_holds(f1, px[i], py[i], pz[i]) || !frame_holds(f2, px[i], py[i], pz[i])) { printf("not held\n"); return 1; }
    }
    printf("ok\n");
    return 0;
}
