#include "rt_mi355x.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
// The host side of the library (KD build on 1, 3 and 8 threads, flat view, refusal of bad indices, camera) driven directly, for
// builds with -fsanitize=address,undefined or -fsanitize=thread (tests/test_host.py, tools/host_sanitizers.sh).  argv[1]: rounds.
int main(int argc, char** argv)
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    for (int round = 0; round < rounds; ++round) {
        // displaced grid plus random small triangles, indexed vertices
        const int n = 30 + 40 * round;
        std::vector<float> v; std::vector<uint32_t> t;
        for (int j = 0; j <= n; ++j) for (int i = 0; i <= n; ++i) {
            float x = -1.5f + 3.0f * i / n, z = -3.0f + 3.0f * j / n;
            v.push_back(x); v.push_back(0.3f + 0.15f * std::sin(4 * x) * std::cos(5 * z)); v.push_back(z);
        }
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
            uint32_t a = j * (n + 1) + i, b = a + 1, c = a + n + 1, d = c + 1;
            t.insert(t.end(), {a, b, c, b, d, c});
        }
        for (int k = 0; k < 200; ++k) {
            uint32_t base = (uint32_t)(v.size() / 3);
            float cx = U(rng), cy = U(rng) + 1.0f, cz = U(rng) - 1.5f;
            for (int q = 0; q < 3; ++q) { v.push_back(cx + 0.05f * U(rng)); v.push_back(cy + 0.05f * U(rng)); v.push_back(cz + 0.05f * U(rng)); }
            t.insert(t.end(), {base, base + 1, base + 2});
        }
        for (int threads : {1, 3, 8}) {
            rt_host_kd* kd = rt_host_kd_build(v.data(), (uint32_t)(v.size() / 3), t.data(), (uint32_t)(t.size() / 3), 1u << 24, threads);
            if (!kd) { printf("build failed\n"); return 1; }
            rt_kd_flat f; uint32_t depth = 0;
            if (rt_host_kd_flat(kd, &f, &depth) != 0) { printf("flat failed\n"); return 1; }
            printf("round %d threads %d: %u tris -> %u nodes, %u leaf refs, depth %u\n", round, threads, (uint32_t)(t.size() / 3), f.n_nodes, f.n_leaf_idx, depth);
            rt_host_kd_free(kd);
        }
    }
    // bad input must be refused, not read out of bounds
    float v[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0}; uint32_t bad[3] = {0, 1, 7};
    rt_host_kd* kd = rt_host_kd_build(v, 3, bad, 1, 1000, 1);
    printf("bad index: %s\n", kd ? "accepted (wrong)" : "refused");
    if (kd) rt_host_kd_free(kd);
    float fwd[3]; rt_host_camera_default_forward(fwd);
    return 0;
}
