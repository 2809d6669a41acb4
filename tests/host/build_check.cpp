// Host-side check of the scene builder: prints a hash of every table build_host_scene produces and the time it took.
// The reference's tree (same std::sort calls, src/acceleration/bvh.h:20-48) and the traversal hierarchy are built by
// several threads that each own a subtree; the tables must not depend on how many there are (BDPT_BUILD_THREADS).
//   build_check <n_triangles> <seed>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../bidirectional-path-tracer_amd/csrc/scene_build.h"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
template <class T> static uint64_t hash_of(const std::vector<T>& v) { return v.empty() ? 0 : fnv(v.data(), v.size() * sizeof(T)); }

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atoll(argv[1]) : 100000u;
    uint64_t state = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 1u;
    auto rnd = [&state]() { state = state * 6364136223846793005ull + 1442695040888963407ull; return (double)(state >> 11) / 9007199254740992.0; };
    // a mix of scattered triangles and a regular grid of quads (many equal sort keys: the order of equal keys is
    // libstdc++'s, and must be the same from every thread)
    std::vector<double> v((size_t)n * 9);
    std::vector<int32_t> mat(n);
    const uint32_t side = 64;
    for (uint32_t i = 0; i < n; i++) {
        double* t = &v[(size_t)i * 9];
        if (i % 3 == 0) {
            const uint32_t q = i / 3, gx = q % side, gz = (q / side) % side, gy = q / (side * side);
            const double x = gx, y = 0.5 * gy, z = gz;
            t[0] = x; t[1] = y; t[2] = z; t[3] = x + 1; t[4] = y; t[5] = z; t[6] = x; t[7] = y; t[8] = z + 1;
        } else {
            const double cx = 64 * rnd(), cy = 16 * rnd(), cz = 64 * rnd();
            for (int k = 0; k < 3; k++) { t[3 * k] = cx + rnd() - 0.5; t[3 * k + 1] = cy + rnd() - 0.5; t[3 * k + 2] = cz + rnd() - 0.5; }
        }
        mat[i] = (int32_t)(i % 2);
    }
    bdpt_material mats[2];
    std::memset(mats, 0, sizeof mats);
    mats[0].type = BDPT_LAMBERTIAN; mats[0].rgb[0] = mats[0].rgb[1] = mats[0].rgb[2] = 0.5;
    mats[1].type = BDPT_DIFFUSE_LIGHT; mats[1].rgb[0] = mats[1].rgb[1] = mats[1].rgb[2] = 4.0;
    const uint32_t light = 1;
    bdpt::HostScene hs;
    const auto t0 = std::chrono::steady_clock::now();
    const std::string err = bdpt::build_host_scene(v.data(), mat.data(), n, mats, 2, &light, 1, hs);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!err.empty()) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::printf("{\"triangles\": %u, \"seconds\": %.3f, \"nodes\": \"%016llx\", \"stris\": \"%016llx\", \"sleaf\": \"%016llx\", \"node_parent\": \"%016llx\", "
                "\"xnodes\": \"%016llx\", \"xtris\": \"%016llx\", \"tri_xpos\": \"%016llx\", \"shade\": \"%016llx\", \"dfs_leaves\": \"%016llx\", "
                "\"n_nodes\": %zu, \"n_xnodes\": %zu, \"sah_stack\": %u, \"ref_nested\": %d}\n",
                n, secs, (unsigned long long)hash_of(hs.nodes), (unsigned long long)hash_of(hs.stris), (unsigned long long)hash_of(hs.sleaf),
                (unsigned long long)hash_of(hs.node_parent), (unsigned long long)hash_of(hs.xnodes), (unsigned long long)hash_of(hs.xtris),
                (unsigned long long)hash_of(hs.tri_xpos), (unsigned long long)hash_of(hs.shade), (unsigned long long)hash_of(hs.dfs_leaves),
                hs.nodes.size(), hs.xnodes.size(), hs.sah_stack, hs.ref_nested ? 1 : 0);
    return 0;
}
