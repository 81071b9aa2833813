// cell_frame_check.cpp -- host check of the compressed cell keys of the rebuild by cell runs (csrc/lio_rebuild.h): for random
// bounding boxes of cells, compress() keeps the order of the absolute Morton codes, expand() inverts it, cells outside the box's
// parents are refused, and the key has the bits the frame says.  Compiled with hipcc (host code only) by tests/test_host_math.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "lio_rebuild.h"

int main() {
    using namespace lio;
    std::mt19937_64 rng(12345);
    int frames = 0;
    for (int trial = 0; trial < 400; ++trial) {
        uint32_t lo[3], hi[3];
        const int kind = trial % 4;
        if (trial == 399) {                                       // the whole grid: two parents per axis at level 10, 63-bit keys
            for (int i = 0; i < 3; ++i) { lo[i] = 0u; hi[i] = (1u << 21) - 1u; }
        } else
        for (int i = 0; i < 3; ++i) {
            // boxes around the bias (a map around the origin), anywhere, tiny, and kilometres wide
            const uint32_t centre = kind == 0 ? (1u << 20) : (uint32_t)(rng() % ((1u << 21) - 70000u)) + 35000u;
            const uint32_t ext = kind == 2 ? (uint32_t)(rng() % 3) : kind == 3 ? (uint32_t)(rng() % 30000) : (uint32_t)(rng() % (i == 2 ? 120 : 1500));
            lo[i] = centre - ext / 2; hi[i] = lo[i] + ext;
        }
        const CellKeyFrame f = cell_key_frame(lo, hi);
        if (f.a < 1 || f.a > 10 || f.n_parents < 1 || f.n_parents > CELL_FRAME_PARENTS || f.bits > 63) { std::printf("bad frame\n"); return 1; }
        if (trial == 399 && (f.a != 10 || f.n_parents != 8 || f.bits != 63)) { std::printf("whole-grid frame: a %d parents %d bits %d\n", f.a, f.n_parents, f.bits); return 1; }
        for (int i = 1; i < f.n_parents; ++i) if (!(f.parent[i - 1] < f.parent[i])) { std::printf("parents not ascending\n"); return 1; }
        std::vector<std::pair<uint64_t, uint64_t>> keys;
        for (int k = 0; k < 3000; ++k) {
            uint32_t c[3];
            for (int i = 0; i < 3; ++i) c[i] = lo[i] + (uint32_t)(rng() % (uint64_t)(hi[i] - lo[i] + 1u));
            if (k < 8) for (int i = 0; i < 3; ++i) c[i] = (k >> i) & 1 ? hi[i] : lo[i];          // the corners
            const uint64_t mk = morton3(c[0], c[1], c[2]);
            bool ok = false;
            const uint64_t ck = cell_key_compress(f, c[0], c[1], c[2], &ok);
            if (!ok) { std::printf("cell of the box refused\n"); return 1; }
            if (f.bits < 64 && (ck >> f.bits) != 0) { std::printf("key wider than the frame says\n"); return 1; }
            if (cell_key_expand(f, ck) != mk) { std::printf("expand does not invert compress\n"); return 1; }
            keys.push_back({mk, ck});
        }
        std::sort(keys.begin(), keys.end());
        for (size_t i = 1; i < keys.size(); ++i)
            if ((keys[i - 1].first < keys[i].first) != (keys[i - 1].second < keys[i].second)) { std::printf("order not kept\n"); return 1; }
        // a cell whose parent is outside the box's parents
        const uint32_t step = 1u << (2 * f.a);
        if (hi[0] + 2 * step < (1u << 21)) {
            bool ok = true;
            (void)cell_key_compress(f, hi[0] + 2 * step, lo[1], lo[2], &ok);
            if (ok) { std::printf("cell outside the frame accepted\n"); return 1; }
        }
        ++frames;
    }
    std::printf("ok %d frames\n", frames);
    return 0;
}
