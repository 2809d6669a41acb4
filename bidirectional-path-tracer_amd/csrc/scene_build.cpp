// Host-side scene preparation (plain C++, no device code).  Built with -ffp-contract=off: the
// triangle precompute and the box unions must round like the reference's.
#include "scene_build.h"
#include "composite.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>

namespace bdpt {
namespace {

struct Box3 { double lo[3], hi[3]; };

struct BuildTri {
    Box3 box;      // aabb(point, point) of the triangle, NOT padded (src/acceleration/aabb.h:18-25)
};

struct BuildNode {
    Box3 box;
    int32_t left, right;   // >= 0: node index, < 0: ~triangle id
};

// interval::expand via aabb::pad_to_minimums, src/acceleration/aabb.h:81-88, src/core/interval.h:36-39
inline void pad_axis(double& lo, double& hi) {
    const double delta = 0.0001;
    if (hi - lo < delta) {
        const double padding = delta / 2;
        lo = lo - padding;
        hi = hi + padding;
    }
}

// aabb(const aabb&, const aabb&), aabb.h:27-33 + interval(a, b), interval.h:12-16
inline Box3 unite(const Box3& a, const Box3& b) {
    Box3 r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = a.lo[k] <= b.lo[k] ? a.lo[k] : b.lo[k];
        r.hi[k] = a.hi[k] >= b.hi[k] ? a.hi[k] : b.hi[k];
        pad_axis(r.lo[k], r.hi[k]);
    }
    return r;
}

inline int longest_axis(const Box3& b) {   // aabb.h:68-75
    const double sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}

// Host threads a scene build may use (BDPT_BUILD_THREADS; by default what the machine reports, at most 16).  The two
// builders below split into independent subtrees; which thread builds which subtree changes nothing of the result.
inline unsigned build_threads() {
    if (const char* v = std::getenv("BDPT_BUILD_THREADS")) { const int n = std::atoi(v); return n < 1 ? 1u : (n > 64 ? 64u : (unsigned)n); }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw < 1 ? 1u : (hw > 16 ? 16u : hw);
}
// levels of the recursion at which both halves of a range of at least kParallelSpan are built by two threads
inline int parallel_levels() {
    int levels = 0;
    for (unsigned t = build_threads(); t > 1; t >>= 1) levels++;
    return levels;
}
constexpr size_t kParallelSpan = 1u << 15;

struct Builder {
    const std::vector<BuildTri>& tris;
    std::vector<BuildNode> nodes;
    std::vector<uint32_t> order;           // the hittable_list being sorted in place (bvh.h:41)

    explicit Builder(const std::vector<BuildTri>& t) : tris(t) {}

    // number of bvh_node objects over a span (bvh.h:29-46: one for a span of one or two, otherwise one plus the halves')
    // (the halves of a span differ by at most one, so a tree has two distinct spans per level: the table is filled once,
    // before any thread starts, and only read afterwards)
    std::unordered_map<size_t, size_t> span_nodes;
    size_t fill_span_nodes(size_t span) {
        if (span <= 2) return 1;
        auto it = span_nodes.find(span);
        if (it != span_nodes.end()) return it->second;
        const size_t n = 1 + fill_span_nodes(span / 2) + fill_span_nodes(span - span / 2);
        span_nodes.emplace(span, n);
        return n;
    }
    size_t nodes_of_span(size_t span) const { return span <= 2 ? 1 : span_nodes.at(span); }

    // bvh_node(objects, start, end), src/acceleration/bvh.h:20-48, into the pre-order array: the node of a range is
    // `idx`, its left subtree follows at idx + 1, its right one behind that (sizes known from the spans alone), so
    // the two halves - each sorts its own part of `order` with the reference's own call - can be built side by side.
    void build_at(size_t start, size_t end, size_t idx, int levels) {
        const double inf = std::numeric_limits<double>::infinity();
        Box3 box;                          // aabb::empty (aabb.h:92): padding leaves the infinities as they are
        for (int k = 0; k < 3; k++) { box.lo[k] = inf; box.hi[k] = -inf; }
        for (size_t i = start; i < end; i++) box = unite(box, tris[order[i]].box);
        const int axis = longest_axis(box);
        const size_t span = end - start;
        int32_t left, right;
        if (span == 1) {
            left = right = ~(int32_t)order[start];
        } else if (span == 2) {
            left = ~(int32_t)order[start];
            right = ~(int32_t)order[start + 1];
        } else {
            const std::vector<BuildTri>& t = tris;
            std::sort(order.begin() + start, order.begin() + end,
                      [&t, axis](uint32_t a, uint32_t b) { return t[a].box.lo[axis] < t[b].box.lo[axis]; });
            const size_t mid = start + span / 2;
            const size_t li = idx + 1, ri = li + nodes_of_span(mid - start);
            if (levels > 0 && span >= kParallelSpan) {
                std::thread other([this, start, mid, li, levels] { build_at(start, mid, li, levels - 1); });
                build_at(mid, end, ri, levels - 1);
                other.join();
            } else {
                build_at(start, mid, li, 0);
                build_at(mid, end, ri, 0);
            }
            left = (int32_t)li;
            right = (int32_t)ri;
        }
        nodes[idx].box = box;
        nodes[idx].left = left;
        nodes[idx].right = right;
    }
    void build_all(size_t n) {
        nodes.assign(fill_span_nodes(n), BuildNode());
        build_at(0, n, 0, parallel_levels());
    }

    // the same, nodes appended one by one (small trees: the objects of a composite scene)
    int32_t build(size_t start, size_t end) {
        const int32_t idx = (int32_t)nodes.size();
        nodes.push_back(BuildNode());

        const double inf = std::numeric_limits<double>::infinity();
        Box3 box;                          // aabb::empty (aabb.h:92): padding leaves the infinities as they are
        for (int k = 0; k < 3; k++) { box.lo[k] = inf; box.hi[k] = -inf; }
        for (size_t i = start; i < end; i++) box = unite(box, tris[order[i]].box);

        const int axis = longest_axis(box);
        const size_t span = end - start;
        int32_t left, right;
        if (span == 1) {
            left = right = ~(int32_t)order[start];
        } else if (span == 2) {
            left = ~(int32_t)order[start];
            right = ~(int32_t)order[start + 1];
        } else {
            // Same comparator (strict < on the triangle boxes' minimum along the axis, bvh.h:68-74) and
            // the same libstdc++ std::sort as the reference, so equal keys land in the same places.
            const std::vector<BuildTri>& t = tris;
            std::sort(order.begin() + start, order.begin() + end,
                      [&t, axis](uint32_t a, uint32_t b) { return t[a].box.lo[axis] < t[b].box.lo[axis]; });
            const size_t mid = start + span / 2;
            left = build(start, mid);
            right = build(mid, end);
        }
        nodes[idx].box = box;
        nodes[idx].left = left;
        nodes[idx].right = right;
        return idx;
    }
};

// std::fmin / std::fmax as the C library the reference is built against evaluates them (glibc, x86-64: one
// minsd / maxsd instruction with the first argument as destination): of two equal operands - +0 and -0 compare
// equal - the SECOND is returned.  A compiler's inline expansion may return the other zero, and the box tables are
// compared bit for bit.
inline double second_fmin(double x, double y) { return (x < y) ? x : ((y != y) ? x : y); }
inline double second_fmax(double x, double y) { return (x > y) ? x : ((y != y) ? x : y); }


// ---------------------------------------------------------------------------------------------
// Binned surface-area-heuristic hierarchy for closest-hit queries.
struct SahRef { double lo[3], hi[3], c[3]; uint32_t pos; };
// binary record of the builder: both child boxes (FP32, rounded outward) and child references
struct SahBinNode { float lo0[3], hi0[3], lo1[3], hi1[3]; int32_t child0, child1; };

inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

struct SahBuilder {
    std::vector<SahRef> own_refs;
    std::vector<SahRef>& refs;             // own_refs, or those of the builder this one works for
    std::vector<SahBinNode> nodes;
    std::vector<uint32_t> leaf_order;      // positions (in stris) in this hierarchy's leaf order
    uint32_t max_depth = 0;
    size_t min_leaf = 2;       // ranges of at most this many triangles always become a leaf
    double box_cost = 1.0;     // cost of one record visit in units of one triangle test
    int parallel_levels_left = 0;

    SahBuilder() : refs(own_refs), parallel_levels_left(parallel_levels()) {}
    // a helper over part of `parent`'s references
    SahBuilder(SahBuilder& parent, int levels) : refs(parent.refs), min_leaf(parent.min_leaf), box_cost(parent.box_cost), parallel_levels_left(levels) {}
    SahBuilder(const SahBuilder&) = delete;

    // Appends what `part` built: its records (interior child references and leaf references moved behind this builder's
    // own) and its leaf order.  `ref`: the child reference `part` returned for its range; returns it as this builder's.
    int32_t adopt(const SahBuilder& part, int32_t ref) {
        const int32_t node_off = (int32_t)nodes.size();
        const uint32_t leaf_off = (uint32_t)leaf_order.size();
        auto moved = [node_off, leaf_off](int32_t r) -> int32_t {
            if (r >= 0) return r + node_off;
            if (r == kQuadEmpty) return r;
            const uint32_t code = (uint32_t)~r;
            return ~(int32_t)((((code >> 2) + leaf_off) << 2) | (code & 3u));
        };
        nodes.reserve(nodes.size() + part.nodes.size());
        for (SahBinNode nd : part.nodes) { nd.child0 = moved(nd.child0); nd.child1 = moved(nd.child1); nodes.push_back(nd); }
        leaf_order.insert(leaf_order.end(), part.leaf_order.begin(), part.leaf_order.end());
        if (part.max_depth > max_depth) max_depth = part.max_depth;
        return moved(ref);
    }

    static double area(const double* lo, const double* hi) {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    void bounds(size_t a, size_t b, double* lo, double* hi) const {
        for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
        for (size_t i = a; i < b; i++)
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], refs[i].lo[k]); hi[k] = std::max(hi[k], refs[i].hi[k]); }
    }
    int32_t make_leaf(size_t a, size_t b) {
        const uint32_t first = (uint32_t)leaf_order.size();
        for (size_t i = a; i < b; i++) leaf_order.push_back(refs[i].pos);
        return ~(int32_t)((first << 2) | (uint32_t)(b - a - 1));
    }
    // returns the child reference of the subtree over refs[a, b)
    int32_t build(size_t a, size_t b, uint32_t depth) {
        if (depth > max_depth) max_depth = depth;
        const size_t n = b - a;
        if (n <= min_leaf) return make_leaf(a, b);
        double clo[3], chi[3];
        for (int k = 0; k < 3; k++) { clo[k] = std::numeric_limits<double>::infinity(); chi[k] = -clo[k]; }
        for (size_t i = a; i < b; i++)
            for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], refs[i].c[k]); chi[k] = std::max(chi[k], refs[i].c[k]); }
        constexpr int kBins = 16;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_split = 0;
        for (int axis = 0; axis < 3; axis++) {
            const double ext = chi[axis] - clo[axis];
            if (!(ext > 0)) continue;
            size_t cnt[kBins] = {0};
            double blo[kBins][3], bhi[kBins][3];
            for (int q = 0; q < kBins; q++)
                for (int k = 0; k < 3; k++) { blo[q][k] = std::numeric_limits<double>::infinity(); bhi[q][k] = -blo[q][k]; }
            const double scale = kBins / ext;
            for (size_t i = a; i < b; i++) {
                int q = (int)((refs[i].c[axis] - clo[axis]) * scale);
                q = q < 0 ? 0 : (q >= kBins ? kBins - 1 : q);
                cnt[q]++;
                for (int k = 0; k < 3; k++) { blo[q][k] = std::min(blo[q][k], refs[i].lo[k]); bhi[q][k] = std::max(bhi[q][k], refs[i].hi[k]); }
            }
            double right_area[kBins];
            size_t right_cnt[kBins];
            double rlo[3], rhi[3];
            for (int k = 0; k < 3; k++) { rlo[k] = std::numeric_limits<double>::infinity(); rhi[k] = -rlo[k]; }
            size_t rc = 0;
            for (int q = kBins - 1; q >= 1; q--) {
                for (int k = 0; k < 3; k++) { rlo[k] = std::min(rlo[k], blo[q][k]); rhi[k] = std::max(rhi[k], bhi[q][k]); }
                rc += cnt[q];
                right_area[q] = rc ? area(rlo, rhi) : 0.0;
                right_cnt[q] = rc;
            }
            double llo[3], lhi[3];
            for (int k = 0; k < 3; k++) { llo[k] = std::numeric_limits<double>::infinity(); lhi[k] = -llo[k]; }
            size_t lc = 0;
            for (int q = 0; q < kBins - 1; q++) {
                for (int k = 0; k < 3; k++) { llo[k] = std::min(llo[k], blo[q][k]); lhi[k] = std::max(lhi[k], bhi[q][k]); }
                lc += cnt[q];
                if (lc == 0 || right_cnt[q + 1] == 0) continue;
                const double cost = area(llo, lhi) * (double)lc + right_area[q + 1] * (double)right_cnt[q + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = q + 1; }
            }
        }
        size_t mid;
        if (best_axis < 0) {
            if (n <= 4) return make_leaf(a, b);
            mid = a + n / 2;                               // all centroids coincide
        } else {
            double lo[3], hi[3];
            bounds(a, b, lo, hi);
            const double parent = area(lo, hi);
            // leaf cost n * 1.0 against split cost 1.0 + children (box test taken as one triangle test)
            if (n <= 4 && parent > 0 && best_cost / parent + box_cost >= (double)n) return make_leaf(a, b);
            const double ext = chi[best_axis] - clo[best_axis];
            const double scale = kBins / ext;
            const double c0 = clo[best_axis];
            const int axis = best_axis, split = best_split;
            auto it = std::partition(refs.begin() + a, refs.begin() + b, [=](const SahRef& r) {
                int q = (int)((r.c[axis] - c0) * scale);
                q = q < 0 ? 0 : (q >= kBins ? kBins - 1 : q);
                return q < split;
            });
            mid = (size_t)(it - refs.begin());
            if (mid == a || mid == b) mid = a + n / 2;
        }
        const int32_t idx = (int32_t)nodes.size();
        nodes.push_back(SahBinNode());
        double lo0[3], hi0[3], lo1[3], hi1[3];
        bounds(a, mid, lo0, hi0);
        bounds(mid, b, lo1, hi1);
        int32_t c0r, c1r;
        if (parallel_levels_left > 0 && n >= kParallelSpan) {
            // Both halves at once: the right one by a builder of its own over the same `refs` (the ranges are disjoint),
            // whose records and leaf order are appended behind the left half's afterwards - the order in which one thread
            // alone would have produced them.
            SahBuilder right(*this, parallel_levels_left - 1);
            int32_t right_ref = 0;
            std::thread other([&right, &right_ref, mid, b, depth] { right_ref = right.build(mid, b, depth + 1); });
            parallel_levels_left--;
            c0r = build(a, mid, depth + 1);
            parallel_levels_left++;
            other.join();
            c1r = adopt(right, right_ref);
        } else {
            c0r = build(a, mid, depth + 1);
            c1r = build(mid, b, depth + 1);
        }
        SahBinNode& nd = nodes[idx];
        for (int k = 0; k < 3; k++) {
            nd.lo0[k] = round_down(lo0[k]); nd.hi0[k] = round_up(hi0[k]);
            nd.lo1[k] = round_down(lo1[k]); nd.hi1[k] = round_up(hi1[k]);
        }
        nd.child0 = c0r; nd.child1 = c1r;
        return idx;
    }
};

// Puts the `top` records nearest the root first (breadth-first), the rest in their previous order,
// and rewrites the interior child references.  The traversal kernels keep that prefix in LDS.
// `kids(node)` returns a pointer to the record's child references, N of them.
template <int N, class NodeT, class Kids>
void hoist_top_levels(std::vector<NodeT>& nodes, size_t top, Kids kids) {
    const size_t n = nodes.size();
    if (n <= 2 || top <= 1) return;
    std::vector<int32_t> order;
    order.reserve(n);
    std::vector<char> taken(n, 0);
    order.push_back(0);
    taken[0] = 1;
    for (size_t head = 0; head < order.size() && order.size() < top; head++) {
        const int32_t* c = kids(nodes[order[head]]);
        for (int k = 0; k < N; k++)
            if (c[k] >= 0 && !taken[c[k]] && order.size() < top) { order.push_back(c[k]); taken[c[k]] = 1; }
    }
    for (size_t i = 0; i < n; i++) if (!taken[i]) order.push_back((int32_t)i);
    std::vector<int32_t> new_index(n);
    for (size_t k = 0; k < n; k++) new_index[order[k]] = (int32_t)k;
    std::vector<NodeT> out(n);
    for (size_t k = 0; k < n; k++) {
        NodeT nd = nodes[order[k]];
        int32_t* c = kids(nd);
        for (int q = 0; q < N; q++) if (c[q] >= 0) c[q] = new_index[c[q]];
        out[k] = nd;
    }
    nodes.swap(out);
}

// Collapses the binary surface-area tree to four children per record: starting from a record's two
// children, the interior child with the largest box is replaced by its own two children until there
// are four (or only leaves are left).  `stack_need` becomes the most entries a depth-first walk that
// pushes every child but the one it enters can have on its stack.
struct QuadCollapse {
    const std::vector<SahBinNode>& bin;
    std::vector<GpuQuadNode> out;
    uint32_t stack_need = 0;
    struct Kid { float lo[3], hi[3]; int32_t ref; };
    static double area(const Kid& k) {
        const double dx = (double)k.hi[0] - k.lo[0], dy = (double)k.hi[1] - k.lo[1], dz = (double)k.hi[2] - k.lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
    static void kids_of(const SahBinNode& b, Kid& k0, Kid& k1) {
        for (int a = 0; a < 3; a++) { k0.lo[a] = b.lo0[a]; k0.hi[a] = b.hi0[a]; k1.lo[a] = b.lo1[a]; k1.hi[a] = b.hi1[a]; }
        k0.ref = b.child0; k1.ref = b.child1;
    }
    explicit QuadCollapse(const std::vector<SahBinNode>& b) : bin(b) {}
    int32_t collapse(int32_t bin_index, uint32_t stack_above) {
        Kid kid[4];
        int n = 2;
        kids_of(bin[bin_index], kid[0], kid[1]);
        while (n < 4) {
            int pick = -1;
            double best = -1.0;
            for (int k = 0; k < n; k++)
                if (kid[k].ref >= 0 && area(kid[k]) > best) { best = area(kid[k]); pick = k; }
            if (pick < 0) break;
            Kid a, b2;
            kids_of(bin[kid[pick].ref], a, b2);
            kid[pick] = a;
            kid[n++] = b2;
        }
        const int32_t idx = (int32_t)out.size();
        out.push_back(GpuQuadNode());
        const uint32_t below = stack_above + (uint32_t)(n - 1);
        if (below > stack_need) stack_need = below;
        int32_t refs[4];
        for (int k = 0; k < n; k++) refs[k] = kid[k].ref >= 0 ? collapse(kid[k].ref, below) : kid[k].ref;
        GpuQuadNode& q = out[idx];
        for (int k = 0; k < 4; k++) {
            for (int a = 0; a < 3; a++) {
                q.box[k][2 * a] = k < n ? kid[k].lo[a] : std::numeric_limits<float>::infinity();
                q.box[k][2 * a + 1] = k < n ? kid[k].hi[a] : -std::numeric_limits<float>::infinity();
            }
            q.child[k] = k < n ? refs[k] : kQuadEmpty;
            q.pad[k] = 0;
        }
        return idx;
    }
};

// Closest-hit hierarchy over the entries of `out.stris` (referenced by their position there).  world_box[p]: the
// box of entry p where rays meet it; own_box[p]: the box of its triangle in the coordinates its test runs in (the two
// differ for a triangle of an instanced mesh).
static void build_traversal_hierarchy(HostScene& out, const std::vector<Box3>& world_box, const std::vector<Box3>& own_box) {
    const uint32_t n_tri = (uint32_t)out.tris.size();
    {
        SahBuilder sb;
        if (const char* v = std::getenv("BDPT_SAH_MIN_LEAF")) {      // a leaf reference has room for 1..4 triangles
            const int m = std::atoi(v);
            sb.min_leaf = (size_t)(m < 1 ? 1 : (m > 4 ? 4 : m));
        }
        if (const char* v = std::getenv("BDPT_SAH_BOX_COST")) sb.box_cost = std::atof(v);
        sb.refs.resize(out.stris.size());
        for (size_t p = 0; p < out.stris.size(); p++) {
            SahRef& r = sb.refs[p];
            r.pos = (uint32_t)p;
            for (int k = 0; k < 3; k++) {
                // the triangle's exact box, padded on every axis: at least the reference's own padding of
                // thin boxes (aabb.h:81-88), and well above the rounding of an accepted hit point
                const double pad = std::fmax(5e-5, 1e-9 * std::fmax(std::fabs(world_box[p].lo[k]), std::fabs(world_box[p].hi[k])));
                r.lo[k] = world_box[p].lo[k] - pad;
                r.hi[k] = world_box[p].hi[k] + pad;
                r.c[k] = 0.5 * (r.lo[k] + r.hi[k]);
            }
        }
        sb.nodes.reserve(out.stris.size());
        sb.leaf_order.reserve(out.stris.size());
        int32_t root = sb.build(0, sb.refs.size(), 1);
        if (root < 0) {                        // one leaf only: wrap it so that record 0 exists
            SahBinNode w{};
            double lo[3], hi[3];
            sb.bounds(0, sb.refs.size(), lo, hi);
            for (int k = 0; k < 3; k++) {
                w.lo0[k] = round_down(lo[k]); w.hi0[k] = round_up(hi[k]);
                w.lo1[k] = std::numeric_limits<float>::infinity(); w.hi1[k] = -std::numeric_limits<float>::infinity();
            }
            w.child0 = root;
            w.child1 = kQuadEmpty;
            sb.nodes.push_back(w);
        }
        QuadCollapse qc(sb.nodes);
        qc.collapse(0, 0);
        hoist_top_levels<4>(qc.out, kLdsTopQuadNodes, [](GpuQuadNode& q) { return &q.child[0]; });
        out.xnodes.swap(qc.out);
        out.sah_stack = qc.stack_need;
        out.coord_bound = 0.0;
        for (const SahRef& r : sb.refs)
            for (int k = 0; k < 3; k++) out.coord_bound = std::fmax(out.coord_bound, std::fmax(std::fabs(r.lo[k]), std::fabs(r.hi[k])));
        for (const Box3& bx : own_box)       // (a mesh's own coordinates, where its triangle tests run)
            for (int k = 0; k < 3; k++) out.coord_bound = std::fmax(out.coord_bound, std::fmax(std::fabs(bx.lo[k]), std::fabs(bx.hi[k])));
        out.coord_bound = (double)round_up(out.coord_bound * 1.0000002);     // covers the outward FP32 rounding of the boxes
        // Hits the traversal kernels re-check against the reference's own tree (wavefront.hip, "exact ties"): a hit
        // within bary_margin of the triangle's outline, or by a ray within 1e-4 rad of the triangle's plane
        // (|det| = |d . (e1 x e2)| below 1e-4 |d| |e1 x e2|; the threshold without the |d| is kept per triangle: the
        // high 32 bits of that double, in the otherwise unused material field).  The margin argument needs a hit inside the outline
        // to be at least ~64 ulp of the coordinates inside the triangle's box on every axis the box is not thin
        // (< 1e-4, where the reference pads) on: bary_margin * 1e-4 >= 64 ulp(4 * coord_bound).  A triangle whose
        // box is thin on two axes has no such axis pair and is always re-checked (threshold +infinity).
        out.bary_margin = std::fmax(1e-6, 64.0 * 0x1p-52 * 4.0 * out.coord_bound / 1e-4);
        out.xtris.resize(sb.leaf_order.size());
        out.tri_graze.assign(n_tri, 0);
        // (indexed by what a path vertex keeps in Vertex::tri: the triangle id, or - in a scene with instances, where an
        // entry is an (object, triangle) pair - the position in `stris`)
        out.tri_xpos.assign(out.vertex_names_entry ? (uint32_t)out.stris.size() : n_tri, 0);
        for (size_t k = 0; k < sb.leaf_order.size(); k++) {
            GpuTri t = out.stris[sb.leaf_order[k]];
            const uint32_t id = (uint32_t)t.pad;
            t.pad = (int32_t)sb.leaf_order[k];
            int thin_axes = 0;
            for (int a = 0; a < 3; a++) thin_axes += (own_box[sb.leaf_order[k]].hi[a] - own_box[sb.leaf_order[k]].lo[a] < 1e-4) ? 1 : 0;
            double graze = 1e-4 * 2.0 * out.aux[id].area;                // |e1 x e2| = 2 area
            if (thin_axes >= 2 || !(graze == graze)) graze = std::numeric_limits<double>::infinity();
            uint64_t bits;
            std::memcpy(&bits, &graze, sizeof bits);
            // Which faces of the reference's boxes above the triangle lie (almost) in the triangle's own plane: bit
            // 2a for the lower face on axis a, 2a + 1 for the upper one.  A visibility ray that ends on the triangle
            // meets such a box at the very end of its interval, where rounding decides whether the reference gets to test
            // the triangle at all (aabb.h:56-63) - but only if it ENTERS the box through that face.  The triangle's own
            // box is not padded (triangle.h:38 builds it from two points); a bvh_node's is when it is thin
            // (aabb.h:26-32,81-88), so the boxes above a flat triangle come within 5e-5 of its plane or coincide with
            // it.  On an axis where the triangle is not thin, a hit away from its outline lies inside every box above it
            // by the margin of wavefront.hip's exact rule: no bit.
            // ... and on which sides every such face lies EXACTLY in the plane of an exactly flat triangle: the plane's
            // coordinate is then the triangle's own, all those boxes give the same distance to it, and the kernels
            // settle the tie with the one product of aabb.h:49-50 that decides it
            const uint32_t pos = sb.leaf_order[k];
            const int32_t ob = out.vertex_names_entry ? out.pos_obj[pos] : -1;
            int thin_world = 0;
            for (int a = 0; a < 3; a++) thin_world += (world_box[pos].hi[a] - world_box[pos].lo[a] < 1e-4) ? 1 : 0;
            uint32_t faces = 0u, inexact = 0u;
            // the boxes of one tree above `first` against the triangle's box `tb` in that tree's coordinates; `same`: the
            // numbers of that tree's slab test on axis a are the very ones the kernels form (see below)
            auto chain = [&](uint32_t first, const Box3& tb, const bool same[3]) {
                // (boxes that contain one another: the one next to the triangle decides, as in reference_reaches)
                for (uint32_t n = first; n != kEndOfTree; n = out.ref_nested ? kEndOfTree : out.node_parent[n])
                    for (int a = 0; a < 3; a++) {
                        if (!(tb.hi[a] - tb.lo[a] < 1e-4)) continue;
                        const bool flat = tb.lo[a] == tb.hi[a];
                        if (!(tb.lo[a] - out.nodes[n].lo[a] >= 1e-5)) {
                            faces |= 1u << (2 * a);
                            if (!(flat && same[a] && out.nodes[n].lo[a] == tb.lo[a])) inexact |= 1u << (2 * a);
                        }
                        if (!(out.nodes[n].hi[a] - tb.hi[a] >= 1e-5)) {
                            faces |= 2u << (2 * a);
                            if (!(flat && same[a] && out.nodes[n].hi[a] == tb.hi[a])) inexact |= 2u << (2 * a);
                        }
                    }
            };
            const bool all_same[3] = {true, true, true};
            if (thin_axes > 0) chain(out.sleaf[pos] & 0x7FFFFFFFu, own_box[pos], all_same);
            if (ob >= 0) {
                // An entry that belongs to an object: the chain runs through the mesh's boxes with the ray as the mesh sees
                // it (above) and goes on through the world's boxes, from the node that holds the object, with the ray
                // itself.  The kernels look at the ray as the mesh sees it: a bit found in the world's tree means the same
                // there on an axis the object's transform leaves alone (every axis of an object that is only moved, y of one
                // that is also turned), and the product that settles an exact tie is the same number in both trees when,
                // in addition, the object is not moved along that axis (x - 0 = x) and the triangle's plane has the same
                // coordinate in both.  Anything else: all bits, none exact - the reference's boxes are asked.
                const CxObject& obj = out.objs[ob];
                const Box3& wb = world_box[pos];
                const Box3& tb = own_box[pos];
                bool same[3], usable = true;
                for (int a = 0; a < 3; a++) {
                    const bool thin_w = wb.hi[a] - wb.lo[a] < 1e-4;
                    const bool axis_kept = !(obj.flags & kCxRotate) || a == 1;
                    if (thin_w && !axis_kept) usable = false;
                    const bool unmoved = !(obj.flags & kCxTranslate) || obj.offset[a] == 0.0;
                    same[a] = axis_kept && unmoved && tb.lo[a] == tb.hi[a] && wb.lo[a] == tb.lo[a] && wb.hi[a] == tb.hi[a];
                }
                if (!usable) { faces = 63u; inexact = 63u; }
                else if (thin_world > 0) chain(out.obj_leaf[ob] & 0x7FFFFFFFu, wb, same);
            }
            // high word of the threshold, rounded up to a multiple of 4096; the face bits ride in the twelve bits below
            t.material = (int32_t)((((uint32_t)(bits >> 32) + 1u + 4095u) & ~4095u) | ((faces & ~inexact) << 6) | faces);
            out.tri_graze[id] = t.material;
            out.tri_xpos[out.vertex_names_entry ? sb.leaf_order[k] : id] = (uint32_t)k;
            out.xtris[k] = t;
        }
    }

}

}  // namespace

std::string build_host_scene(const double* tv, const int32_t* tri_material, uint32_t n_tri,
                             const bdpt_material* materials, uint32_t n_mat, const uint32_t* light_triangles,
                             uint32_t n_lights, HostScene& out) {
    if (!tv || !tri_material || !materials) return "null table pointer";
    if (n_tri == 0) return "scene has no triangles (the reference cannot build a BVH over an empty list either)";
    if (n_tri >= (1u << 29)) return "too many triangles";
    if (n_mat == 0) return "scene has no materials";
    if (n_lights > 0 && !light_triangles) return "null light list";

    out = HostScene();
    out.mats.resize(n_mat);
    for (uint32_t i = 0; i < n_mat; i++) {
        const bdpt_material& m = materials[i];
        if (m.type < BDPT_LAMBERTIAN || m.type > BDPT_DIFFUSE_LIGHT) return "unknown material type";
        GpuMaterial g;
        g.rgb[0] = m.rgb[0]; g.rgb[1] = m.rgb[1]; g.rgb[2] = m.rgb[2];
        g.fuzz = m.fuzz < 1 ? m.fuzz : 1;            // metal::metal, src/materials/material.h:71
        g.ior = m.ior;
        g.type = m.type;
        g.pad = 0;
        out.mats[i] = g;
    }

    // triangle::triangle, src/objects/primatives/triangle.h:21-39
    out.tris.resize(n_tri);
    out.aux.resize(n_tri);
    std::vector<BuildTri> btris(n_tri);
    for (uint32_t i = 0; i < n_tri; i++) {
        const double* v0 = tv + 9 * (size_t)i;
        const double* v1 = v0 + 3;
        const double* v2 = v0 + 6;
        if (tri_material[i] < 0 || (uint32_t)tri_material[i] >= n_mat) return "triangle material index out of range";
        GpuTri& t = out.tris[i];
        for (int k = 0; k < 3; k++) {
            t.v0[k] = v0[k];
            t.e1[k] = v1[k] - v0[k];
            t.e2[k] = v2[k] - v0[k];
        }
        t.material = tri_material[i];
        t.pad = 0;
        if (materials[tri_material[i]].type == BDPT_LAMBERTIAN) out.has_lambertian = true;
        const double nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
        const double ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
        const double nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
        const double nlen = std::sqrt(nx * nx + ny * ny + nz * nz);
        const double inv = 1 / nlen;                 // unit_vector: v / len == (1/len) * v, vec3.h:93-95,109-111
        GpuTriAux& a = out.aux[i];
        a.normal[0] = inv * nx; a.normal[1] = inv * ny; a.normal[2] = inv * nz;
        a.area = 0.5 * nlen;
        for (int k = 0; k < 3; k++) {
            const double lo = second_fmin(v0[k], second_fmin(v1[k], v2[k]));      // triangle.h:31-36
            const double hi = second_fmax(v0[k], second_fmax(v1[k], v2[k]));
            if (lo <= hi) { btris[i].box.lo[k] = lo; btris[i].box.hi[k] = hi; }
            else { btris[i].box.lo[k] = hi; btris[i].box.hi[k] = lo; }
        }
    }

    // bvh_node(world.to_hittable_list()), src/main.cpp:66
    Builder b(btris);
    b.order.resize(n_tri);
    for (uint32_t i = 0; i < n_tri; i++) b.order[i] = i;
    b.build_all(n_tri);

    // thread the pre-order array: skip = first node after the subtree
    const uint32_t n_nodes = (uint32_t)b.nodes.size();
    std::vector<uint32_t> subtree_end(n_nodes);
    std::vector<uint32_t> depth(n_nodes, 1);
    for (int64_t i = (int64_t)n_nodes - 1; i >= 0; i--) {
        const BuildNode& n = b.nodes[i];
        subtree_end[i] = (n.left >= 0) ? subtree_end[n.right] : (uint32_t)i + 1;
    }
    out.nodes.resize(n_nodes);
    out.tree_depth = 1;
    for (uint32_t i = 0; i < n_nodes; i++) {
        const BuildNode& n = b.nodes[i];
        GpuNode& g = out.nodes[i];
        for (int k = 0; k < 3; k++) { g.lo[k] = n.box.lo[k]; g.hi[k] = n.box.hi[k]; }
        const uint32_t end = subtree_end[i];
        g.skip = (end >= n_nodes) ? kEndOfTree : end;
        if (n.left >= 0) {
            g.next = (uint32_t)n.left;               // == i + 1 in pre-order
            g.tri0 = g.tri1 = -1;
            depth[n.left] = depth[n.right] = depth[i] + 1;
            if (depth[i] + 1 > out.tree_depth) out.tree_depth = depth[i] + 1;
        } else {
            g.next = g.skip;
            g.tri0 = ~n.left;
            g.tri1 = ~n.right;
            out.dfs_leaves.push_back((uint32_t)g.tri0);
            out.dfs_leaves.push_back((uint32_t)g.tri1);
        }
    }

    // triangles in the reference's left-to-right leaf order (the tie-break key of closest-hit queries), and for
    // each of them the bvh_node that holds it; parents of all nodes (visibility queries re-check the reference's
    // own boxes from that node up to the root)
    {
        std::vector<uint32_t> node_of(n_tri, 0);
        out.node_parent.assign(n_nodes, kEndOfTree);
        for (uint32_t i = 0; i < n_nodes; i++) {
            const BuildNode& n = b.nodes[i];
            if (n.left >= 0) { out.node_parent[n.left] = i; out.node_parent[n.right] = i; }
            else { node_of[(uint32_t)~n.left] = i; node_of[(uint32_t)~n.right] = i; }
        }
        out.stris.reserve(n_tri);
        out.sleaf.reserve(n_tri);
        for (size_t k = 0; k < out.dfs_leaves.size(); k++) {
            const uint32_t id = out.dfs_leaves[k];
            if (k > 0 && out.dfs_leaves[k - 1] == id && (k & 1)) continue;      // second mention by a one-triangle node
            GpuTri t = out.tris[id];
            t.pad = (int32_t)id;
            out.stris.push_back(t);
            out.sleaf.push_back(node_of[id]);
        }
        out.tri_leaf.swap(node_of);
        out.shade.resize(out.stris.size());
        out.tri_pos.assign(n_tri, 0);
        for (size_t k = 0; k < out.stris.size(); k++) {
            const uint32_t id = (uint32_t)out.stris[k].pad;
            const GpuMaterial& m = out.mats[out.tris[id].material];
            GpuShadeRec& r = out.shade[k];
            for (int c = 0; c < 3; c++) { r.normal[c] = out.aux[id].normal[c]; r.rgb[c] = m.rgb[c]; }
            r.param = m.type == 1 ? m.fuzz : m.ior;
            r.tri = id;
            r.mat = (uint32_t)out.tris[id].material | ((uint32_t)m.type << 30);
            out.tri_pos[id] = (uint32_t)k;
        }
        // Every box contains the boxes below it, coordinate by coordinate (a parent's box is the min/max of its
        // children's, padded outwards at most).  The device relies on it (wavefront.hip: reference_reaches), so it is
        // checked rather than assumed.
        out.ref_nested = true;
        for (uint32_t i = 0; i < n_nodes; i++) {
            const uint32_t p = out.node_parent[i];
            if (p == kEndOfTree) continue;
            for (int k = 0; k < 3; k++)
                if (!(out.nodes[p].lo[k] <= out.nodes[i].lo[k] && out.nodes[i].hi[k] <= out.nodes[p].hi[k])) out.ref_nested = false;
        }
        if (out.ref_nested) for (uint32_t& leaf : out.sleaf) leaf |= 0x80000000u;      // where the traversal kernels read it
    }

    // closest-hit hierarchy over the same triangles (referenced by their position in `stris`)
    {
        std::vector<Box3> boxes(out.stris.size());
        for (size_t p = 0; p < out.stris.size(); p++) boxes[p] = btris[(uint32_t)out.stris[p].pad].box;
        build_traversal_hierarchy(out, boxes, boxes);
    }

    // the light collection handed to camera::render: `lights`, or the whole world when it is empty (main.cpp:67)
    if (n_lights == 0) {
        out.light_tris.resize(n_tri);
        for (uint32_t i = 0; i < n_tri; i++) out.light_tris[i] = i;
    } else {
        out.light_tris.assign(light_triangles, light_triangles + n_lights);
        for (uint32_t id : out.light_tris) if (id >= n_tri) return "light triangle index out of range";
    }
    // running area sums in the order triangle_collection::sample_surface adds them (triangle.h:203-219)
    out.light_accum.resize(out.light_tris.size());
    double accum = 0.0;
    for (size_t i = 0; i < out.light_tris.size(); i++) {
        accum += out.aux[out.light_tris[i]].area;
        out.light_accum[i] = accum;
    }
    out.light_total_area = accum;
    return std::string();
}

// ---------------------------------------------------------------------------------------------
// Composite scenes (composite.h)
namespace {

// triangle::triangle, src/objects/primatives/triangle.h:21-39
void precompute_triangle(const double* v0, int32_t material, GpuTri& t, GpuTriAux& a, Box3& box) {
    const double* v1 = v0 + 3;
    const double* v2 = v0 + 6;
    for (int k = 0; k < 3; k++) {
        t.v0[k] = v0[k];
        t.e1[k] = v1[k] - v0[k];
        t.e2[k] = v2[k] - v0[k];
    }
    t.material = material;
    t.pad = 0;
    const double nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
    const double ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
    const double nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    const double nlen = std::sqrt(nx * nx + ny * ny + nz * nz);
    const double inv = 1 / nlen;
    a.normal[0] = inv * nx; a.normal[1] = inv * ny; a.normal[2] = inv * nz;
    a.area = 0.5 * nlen;
    for (int k = 0; k < 3; k++) {
        const double lo = second_fmin(v0[k], second_fmin(v1[k], v2[k]));
        const double hi = second_fmax(v0[k], second_fmax(v1[k], v2[k]));
        if (lo <= hi) { box.lo[k] = lo; box.hi[k] = hi; }
        else { box.lo[k] = hi; box.hi[k] = lo; }
    }
}

// One bvh_node tree in pre-order -> threaded records appended to `out`.  item_ref[i]: the leaf reference of list
// entry i (triangle id, or -(object + 2)).  `leaves` receives the references in visiting order.
void append_threaded(const std::vector<BuildNode>& bn, const std::vector<int32_t>& item_ref, std::vector<GpuNode>& out,
                     uint32_t& tree_depth, std::vector<int32_t>* leaves, std::vector<uint32_t>& parents) {
    const uint32_t base = (uint32_t)out.size();
    const uint32_t n = (uint32_t)bn.size();
    parents.resize(base + n, kEndOfTree);
    for (uint32_t i = 0; i < n; i++)
        if (bn[i].left >= 0) { parents[base + (uint32_t)bn[i].left] = base + i; parents[base + (uint32_t)bn[i].right] = base + i; }
    std::vector<uint32_t> subtree_end(n), depth(n, 1);
    for (int64_t i = (int64_t)n - 1; i >= 0; i--) subtree_end[i] = (bn[i].left >= 0) ? subtree_end[bn[i].right] : (uint32_t)i + 1;
    for (uint32_t i = 0; i < n; i++) {
        GpuNode g;
        for (int k = 0; k < 3; k++) { g.lo[k] = bn[i].box.lo[k]; g.hi[k] = bn[i].box.hi[k]; }
        const uint32_t end = subtree_end[i];
        g.skip = (end >= n) ? kEndOfTree : base + end;
        if (bn[i].left >= 0) {
            g.next = base + (uint32_t)bn[i].left;
            g.tri0 = g.tri1 = -1;
            depth[bn[i].left] = depth[bn[i].right] = depth[i] + 1;
            if (depth[i] + 1 > tree_depth) tree_depth = depth[i] + 1;
        } else {
            g.next = g.skip;
            g.tri0 = item_ref[(uint32_t)~bn[i].left];
            g.tri1 = item_ref[(uint32_t)~bn[i].right];
            if (leaves) { leaves->push_back(g.tri0); leaves->push_back(g.tri1); }
        }
        out.push_back(g);
    }
}

// every texture reachable from `t` (through checker children)
bool reaches_image(const std::vector<CxTexture>& tex, int32_t t) {
    if (tex[t].type == BDPT_TEX_IMAGE) return true;
    if (tex[t].type == BDPT_TEX_CHECKER) return reaches_image(tex, tex[t].a) || reaches_image(tex, tex[t].b);
    return false;
}

}  // namespace

std::string build_composite_scene(const double* tv, const int32_t* tri_material, uint32_t n_tri, const bdpt_material* materials,
                                  uint32_t n_mat, const bdpt_texture* textures, uint32_t n_tex, const bdpt_object* objects,
                                  uint32_t n_obj, const double* sin_theta, const double* cos_theta,
                                  const uint32_t* light_triangles, uint32_t n_lights, CxHostScene& out) {
    if (!tv || !tri_material || !materials) return "null table pointer";
    if (n_tri == 0) return "scene has no triangles";
    if (n_tri >= (1u << 29)) return "too many triangles";
    if (n_mat == 0) return "scene has no materials";
    if (n_tex > 0 && !textures) return "null texture table";
    if (n_obj > 0 && (!objects || !sin_theta || !cos_theta)) return "null object table";
    if (n_lights > 0 && !light_triangles) return "null light list";
    out = CxHostScene();

    // textures, src/materials/textures/texture.h
    out.tex.resize(n_tex);
    for (uint32_t i = 0; i < n_tex; i++) {
        const bdpt_texture& t = textures[i];
        CxTexture& c = out.tex[i];
        c = CxTexture();
        c.type = t.type;
        for (int k = 0; k < 3; k++) c.rgb[k] = t.rgb[k];
        switch (t.type) {
            case BDPT_TEX_SOLID: break;
            case BDPT_TEX_CHECKER:
                // children are constructed before the texture that holds them (texture.h:30-34)
                if (t.even < 0 || t.odd < 0 || (uint32_t)t.even >= i || (uint32_t)t.odd >= i) return "checker texture: sub-texture index must be smaller than its own";
                c.scale = 1.0 / t.scale;                       // inv_scale, texture.h:31
                c.a = t.even; c.b = t.odd;
                break;
            case BDPT_TEX_NOISE: {
                if (!t.data) return "noise texture without perlin tables";
                const bdpt_perlin_tables* p = static_cast<const bdpt_perlin_tables*>(t.data);
                CxPerlin cp;
                std::memcpy(cp.randvec, p->randvec, sizeof cp.randvec);
                std::memcpy(cp.perm_x, p->perm_x, sizeof cp.perm_x);
                std::memcpy(cp.perm_y, p->perm_y, sizeof cp.perm_y);
                std::memcpy(cp.perm_z, p->perm_z, sizeof cp.perm_z);
                for (int k = 0; k < 256; k++)
                    if ((uint32_t)cp.perm_x[k] > 255u || (uint32_t)cp.perm_y[k] > 255u || (uint32_t)cp.perm_z[k] > 255u) return "perlin permutation entry out of range";
                c.scale = t.scale;
                c.data = (uint32_t)out.perlin.size();
                out.perlin.push_back(cp);
                break;
            }
            case BDPT_TEX_IMAGE: {
                c.a = t.width; c.b = t.height;
                if (t.height > 0) {
                    if (t.width <= 0 || !t.data) return "image texture without pixel data";
                    const size_t bytes = (size_t)t.width * (size_t)t.height * 3;
                    if (out.image_bytes.size() + bytes >= ((size_t)1 << 32)) return "image textures too large";
                    c.data = (uint32_t)out.image_bytes.size();
                    const uint8_t* src = static_cast<const uint8_t*>(t.data);
                    out.image_bytes.insert(out.image_bytes.end(), src, src + bytes);
                }
                break;
            }
            default: return "unknown texture type";
        }
    }
    if (out.image_bytes.empty()) out.image_bytes.push_back(0);

    out.mats.resize(n_mat);
    for (uint32_t i = 0; i < n_mat; i++) {
        const bdpt_material& m = materials[i];
        if (m.type < BDPT_LAMBERTIAN || m.type > BDPT_ISOTROPIC) return "unknown material type";
        if (m.reserved < 0 || (uint32_t)m.reserved > n_tex) return "material texture index out of range";
        if (m.reserved > 0 && (m.type == BDPT_METAL || m.type == BDPT_DIELECTRIC)) return "metal and dielectric take no texture";
        if (m.type == BDPT_ISOTROPIC && m.reserved > 0 && reaches_image(out.tex, m.reserved - 1))
            return "isotropic material with an image texture: the reference reads unset texture coordinates there";
        GpuMaterial g;
        g.rgb[0] = m.rgb[0]; g.rgb[1] = m.rgb[1]; g.rgb[2] = m.rgb[2];
        g.fuzz = m.fuzz < 1 ? m.fuzz : 1;
        g.ior = m.ior;
        g.type = m.type;
        g.pad = m.reserved;
        out.mats[i] = g;
    }

    // triangles; which object each belongs to
    out.tris.resize(n_tri);
    out.aux.resize(n_tri);
    out.tri_object.assign(n_tri, -1);
    std::vector<BuildTri> btris(n_tri);
    for (uint32_t i = 0; i < n_tri; i++) {
        if (tri_material[i] < 0 || (uint32_t)tri_material[i] >= n_mat) return "triangle material index out of range";
        if (materials[tri_material[i]].type == BDPT_ISOTROPIC) return "isotropic is the phase function of a medium, not a surface material";
        precompute_triangle(tv + 9 * (size_t)i, tri_material[i], out.tris[i], out.aux[i], btris[i].box);
    }
    // Objects may share a mesh (the same range: one bvh_node behind several wrappers, as shared_ptr allows in the
    // reference); two different ranges must not overlap.
    std::vector<uint32_t> by_first(n_obj);
    for (uint32_t k = 0; k < n_obj; k++) by_first[k] = k;
    std::stable_sort(by_first.begin(), by_first.end(), [objects](uint32_t a, uint32_t b) { return objects[a].first_triangle < objects[b].first_triangle; });
    for (uint32_t q = 0; q < n_obj; q++) {
        const bdpt_object& ob = objects[by_first[q]];
        if (ob.n_triangles == 0) return "object without triangles";
        if ((uint64_t)ob.first_triangle + ob.n_triangles > n_tri) return "object triangle range outside the triangle array";
        if (q > 0) {
            const bdpt_object& pv = objects[by_first[q - 1]];
            const bool same = pv.first_triangle == ob.first_triangle && pv.n_triangles == ob.n_triangles;
            if (!same && pv.first_triangle + pv.n_triangles > ob.first_triangle) return "object triangle ranges must be identical (a shared mesh) or disjoint";
        }
        for (uint32_t i = 0; i < ob.n_triangles; i++) out.tri_object[ob.first_triangle + i] = (int32_t)by_first[q];
        if (ob.flags & ~(kCxRotate | kCxTranslate | kCxMedium)) return "unknown object flag";
        if (ob.flags & kCxMedium) {
            if (ob.phase_material < 0 || (uint32_t)ob.phase_material >= n_mat || materials[ob.phase_material].type != BDPT_ISOTROPIC)
                return "medium without an isotropic phase material";
        }
    }

    // each object's own bvh_node(hittable_list of its triangles), and its box in the world
    std::vector<std::vector<BuildNode>> obj_nodes(n_obj);     // filled for the first object of every mesh
    std::vector<int32_t> mesh_of(n_obj, -1);                   // the object that built the mesh this object uses
    std::vector<Box3> obj_box(n_obj);
    out.objs.resize(n_obj);
    for (uint32_t q = 0; q < n_obj; q++) {
        const uint32_t k = by_first[q];
        const bdpt_object& ob = objects[k];
        if (q > 0 && objects[by_first[q - 1]].first_triangle == ob.first_triangle) {
            mesh_of[k] = mesh_of[by_first[q - 1]];
        } else {
            mesh_of[k] = (int32_t)k;
            std::vector<BuildTri> local(btris.begin() + ob.first_triangle, btris.begin() + ob.first_triangle + ob.n_triangles);
            Builder b(local);
            b.order.resize(ob.n_triangles);
            for (uint32_t i = 0; i < ob.n_triangles; i++) b.order[i] = i;
            b.build(0, ob.n_triangles);
            obj_nodes[k].swap(b.nodes);
        }
        Box3 box = obj_nodes[mesh_of[k]][0].box;                // bvh_node::bounding_box
        CxObject& co = out.objs[k];
        co = CxObject();
        co.flags = ob.flags;
        co.phase_material = (ob.flags & kCxMedium) ? ob.phase_material : -1;
        co.neg_inv_density = (ob.flags & kCxMedium) ? -1 / ob.density : 0.0;      // constant_medium.h:11
        co.sin_theta = sin_theta[k]; co.cos_theta = cos_theta[k];
        for (int a = 0; a < 3; a++) co.offset[a] = ob.offset[a];
        if (ob.flags & kCxRotate) {                             // rotate_y::rotate_y, hittable.h:82-107
            const double inf = std::numeric_limits<double>::infinity();
            double mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int q = 0; q < 2; q++) {
                        const double x = i * box.hi[0] + (1 - i) * box.lo[0];
                        const double y = j * box.hi[1] + (1 - j) * box.lo[1];
                        const double z = q * box.hi[2] + (1 - q) * box.lo[2];
                        const double newx = co.cos_theta * x + co.sin_theta * z;
                        const double newz = -co.sin_theta * x + co.cos_theta * z;
                        const double tester[3] = {newx, y, newz};
                        for (int c = 0; c < 3; c++) {
                            mn[c] = second_fmin(mn[c], tester[c]);
                            mx[c] = second_fmax(mx[c], tester[c]);
                        }
                    }
            for (int c = 0; c < 3; c++) {                       // aabb(point, point), aabb.h:18-25: no padding
                if (mn[c] <= mx[c]) { box.lo[c] = mn[c]; box.hi[c] = mx[c]; }
                else { box.lo[c] = mx[c]; box.hi[c] = mn[c]; }
            }
        }
        if (ob.flags & kCxTranslate) {                          // bbox + offset, aabb.h:95-97 -> aabb(interval x3) pads
            for (int c = 0; c < 3; c++) {
                box.lo[c] = box.lo[c] + ob.offset[c];
                box.hi[c] = box.hi[c] + ob.offset[c];
                pad_axis(box.lo[c], box.hi[c]);
            }
        }
        obj_box[k] = box;
    }

    // the world's list: triangles of no object and the objects, each object at the place of its first triangle
    // (objects that share a mesh in the order of their indices)
    std::vector<BuildTri> items;
    std::vector<int32_t> item_ref;
    {
        uint32_t q = 0;
        for (uint32_t i = 0; i < n_tri;) {
            BuildTri it;
            if (out.tri_object[i] < 0) { it.box = btris[i].box; items.push_back(it); item_ref.push_back((int32_t)i); i++; continue; }
            uint32_t span = 0;
            while (q < n_obj && objects[by_first[q]].first_triangle == i) {
                const uint32_t k = by_first[q++];
                it.box = obj_box[k]; items.push_back(it); item_ref.push_back(-((int32_t)k + 2));
                span = objects[k].n_triangles;
            }
            i += span;
        }
    }
    Builder wb(items);
    wb.order.resize(items.size());
    for (size_t i = 0; i < items.size(); i++) wb.order[i] = (uint32_t)i;
    wb.build(0, items.size());

    out.tree_depth = 1;
    append_threaded(wb.nodes, item_ref, out.nodes, out.tree_depth, &out.world_leaves, out.node_parent);
    out.n_world_nodes = (uint32_t)out.nodes.size();
    for (uint32_t k = 0; k < n_obj; k++) {
        if (mesh_of[k] != (int32_t)k) continue;
        std::vector<int32_t> refs(objects[k].n_triangles);
        for (uint32_t i = 0; i < objects[k].n_triangles; i++) refs[i] = (int32_t)(objects[k].first_triangle + i);
        out.objs[k].root = (uint32_t)out.nodes.size();
        uint32_t d = 1;
        append_threaded(obj_nodes[k], refs, out.nodes, d, nullptr, out.node_parent);
    }
    for (uint32_t k = 0; k < n_obj; k++) out.objs[k].root = out.objs[mesh_of[k]].root;

    // the light collection: triangles of the world's list only
    if (n_lights == 0) {
        for (uint32_t i = 0; i < n_tri; i++) if (out.tri_object[i] < 0) out.light_tris.push_back(i);
    } else {
        out.light_tris.assign(light_triangles, light_triangles + n_lights);
        for (uint32_t id : out.light_tris) {
            if (id >= n_tri) return "light triangle index out of range";
            if (out.tri_object[id] >= 0) return "light triangles must not belong to an object";
        }
    }
    out.light_accum.resize(out.light_tris.size());
    double accum = 0.0;
    for (size_t i = 0; i < out.light_tris.size(); i++) {
        accum += out.aux[out.light_tris[i]].area;
        out.light_accum[i] = accum;
    }
    out.light_total_area = accum;
    out.n_lights = (uint32_t)out.light_tris.size();
    if (out.light_tris.empty()) { out.light_tris.push_back(0); out.light_accum.push_back(0.0); out.light_total_area = 0.0; }
    out.first_triangle.resize(n_obj);
    out.mesh_triangles.resize(n_obj);
    for (uint32_t k = 0; k < n_obj; k++) { out.first_triangle[k] = objects[k].first_triangle; out.mesh_triangles[k] = objects[k].n_triangles; }
    out.pipeline_ok = true;                                   // no medium: see composite.h
    for (const CxObject& ob : out.objs) if (ob.flags & kCxMedium) out.pipeline_ok = false;
    out.textured = false;
    for (const GpuMaterial& m : out.mats) if (m.pad > 0) out.textured = true;
    return std::string();
}

// Tables of the render pipeline (SceneDev) for a composite scene without media.  The leaf order is the
// reference's visiting order over the two-level structure: the leaves of the world's tree left to right, an object
// standing for the triangles of its mesh in the order of the mesh's own tree.
std::string build_instanced_pipeline(const CxHostScene& cx, HostScene& out) {
    out = HostScene();
    const uint32_t n_tri = (uint32_t)cx.tris.size();
    const uint32_t n_obj = (uint32_t)cx.objs.size();
    out.nodes = cx.nodes;
    out.node_parent = cx.node_parent;
    out.tris = cx.tris;
    out.aux = cx.aux;
    out.mats = cx.mats;                                       // (pad: texture index + 1)
    out.light_tris = cx.light_tris;
    out.light_accum = cx.light_accum;
    out.light_total_area = cx.light_total_area;
    out.n_lights = cx.n_lights;
    out.tree_depth = cx.tree_depth;
    out.objs = cx.objs;
    for (uint32_t i = 0; i < n_tri; i++) if (out.mats[out.tris[i].material].type == BDPT_LAMBERTIAN) out.has_lambertian = true;
    for (const CxObject& ob : cx.objs) if (ob.flags & kCxMedium) out.has_lambertian = true;      // (its phase function forms connections like a lambertian surface)

    // every box of a tree contains the boxes below it (what reference_reaches relies on), checked tree by tree
    out.ref_nested = true;
    for (uint32_t i = 0; i < out.nodes.size(); i++) {
        const uint32_t p = out.node_parent[i];
        if (p == kEndOfTree) continue;
        for (int k = 0; k < 3; k++)
            if (!(out.nodes[p].lo[k] <= out.nodes[i].lo[k] && out.nodes[i].hi[k] <= out.nodes[p].hi[k])) out.ref_nested = false;
    }
    const uint32_t nested_bit = out.ref_nested ? 0x80000000u : 0u;

    // leaves of one tree of the table, in visiting order: (triangle or object reference, node)
    auto leaves_of = [&](uint32_t first, uint32_t count, std::vector<std::pair<int32_t, uint32_t>>& dst) {
        for (uint32_t i = first; i < first + count; i++) {
            const GpuNode& g = out.nodes[i];
            if (g.tri0 == -1) continue;
            dst.push_back({g.tri0, i});
            if (g.tri1 != g.tri0) dst.push_back({g.tri1, i});
        }
    };
    // node count of every mesh tree: the trees follow each other in the table in the order of their roots
    std::vector<uint32_t> roots;
    for (uint32_t k = 0; k < n_obj; k++) roots.push_back(cx.objs[k].root);
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
    auto tree_size = [&](uint32_t root) {
        auto it = std::upper_bound(roots.begin(), roots.end(), root);
        const uint32_t end = (it == roots.end()) ? (uint32_t)out.nodes.size() : *it;
        return end - root;
    };

    out.tri_pos.assign(n_tri, 0);
    out.tri_leaf.assign(n_tri, 0);
    out.obj_leaf.assign(n_obj, 0);
    out.obj_pos0.assign(n_obj, 0);
    std::vector<Box3> world_box, own_box;
    std::vector<std::pair<int32_t, uint32_t>> world;
    leaves_of(0, cx.n_world_nodes, world);
    auto tri_box = [&](uint32_t id, const CxObject* ob, Box3& wbox, Box3& obox) {
        const GpuTri& t = out.tris[id];
        double v[3][3];
        for (int k = 0; k < 3; k++) { v[0][k] = t.v0[k]; v[1][k] = t.v0[k] + t.e1[k]; v[2][k] = t.v0[k] + t.e2[k]; }
        for (int k = 0; k < 3; k++) {
            obox.lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k]));
            obox.hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
        }
        if (ob) {                                  // where the mesh's triangle stands in the world (hittable.h:135-139,63)
            for (int q = 0; q < 3; q++) {
                double x = v[q][0], y = v[q][1], z = v[q][2];
                if (ob->flags & kCxRotate) {
                    const double nx = ob->cos_theta * x + ob->sin_theta * z, nz = -ob->sin_theta * x + ob->cos_theta * z;
                    x = nx; z = nz;
                }
                if (ob->flags & kCxTranslate) { x += ob->offset[0]; y += ob->offset[1]; z += ob->offset[2]; }
                v[q][0] = x; v[q][1] = y; v[q][2] = z;
            }
        }
        for (int k = 0; k < 3; k++) {
            wbox.lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k]));
            wbox.hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
        }
    };
    auto push_entry = [&](uint32_t id, int32_t obj, uint32_t node) {
        GpuTri t = out.tris[id];
        t.pad = (int32_t)id;
        out.stris.push_back(t);
        out.sleaf.push_back(node | nested_bit);
        out.pos_obj.push_back(obj);
        Box3 wbox, obox;
        tri_box(id, obj >= 0 ? &cx.objs[obj] : nullptr, wbox, obox);
        world_box.push_back(wbox);
        own_box.push_back(obox);
    };
    std::vector<char> seen(n_obj, 0);
    for (const auto& leaf : world) {
        if (leaf.first >= 0) {
            const uint32_t id = (uint32_t)leaf.first;
            out.tri_pos[id] = (uint32_t)out.stris.size();
            out.tri_leaf[id] = leaf.second;
            push_entry(id, -1, leaf.second);
            continue;
        }
        const uint32_t k = (uint32_t)(-(leaf.first + 2));
        if (seen[k]) continue;                     // (a node over one entry lists it twice)
        seen[k] = 1;
        out.obj_leaf[k] = leaf.second | nested_bit;
        out.obj_pos0[k] = (uint32_t)out.stris.size();
        std::vector<std::pair<int32_t, uint32_t>> mesh;
        leaves_of(cx.objs[k].root, tree_size(cx.objs[k].root), mesh);
        uint32_t rank = 0;
        for (const auto& ml : mesh) {
            const uint32_t id = (uint32_t)ml.first;
            out.tri_pos[id] = rank++;              // (the same for every object that shares the mesh)
            out.tri_leaf[id] = ml.second;
            push_entry(id, (int32_t)k, ml.second);
        }
    }
    if (out.stris.size() >= (1u << 29)) return "too many triangles once the instances are counted";
    out.shade.resize(out.stris.size());
    for (size_t p = 0; p < out.stris.size(); p++) {
        const uint32_t id = (uint32_t)out.stris[p].pad;
        const GpuMaterial& m = out.mats[out.tris[id].material];
        GpuShadeRec& r = out.shade[p];
        for (int c = 0; c < 3; c++) { r.normal[c] = out.aux[id].normal[c]; r.rgb[c] = m.rgb[c]; }
        r.param = m.type == 1 ? m.fuzz : m.ior;
        r.tri = id;
        r.mat = (uint32_t)out.tris[id].material | ((uint32_t)m.type << 30);
    }
    out.vertex_names_entry = !cx.objs.empty() || cx.textured || !cx.pipeline_ok;       // (what the host uploads as SceneDev::objs != null)
    build_traversal_hierarchy(out, world_box, own_box);
    out.xobj.resize(out.xtris.size());
    for (size_t k = 0; k < out.xtris.size(); k++) out.xobj[k] = out.pos_obj[(uint32_t)out.xtris[k].pad];
    return std::string();
}

std::string camera_initialize(const bdpt_camera_public& c, bdpt_camera_state& out) {
    if (c.image_width <= 0) return "image_width must be positive";
    if (!(c.aspect_ratio > 0)) return "aspect_ratio must be positive";
    if (c.samples_per_pixel < 1) return "samples_per_pixel must be at least 1";
    const double pi = 3.1415926535897932385;          // src/main.h:20
    int image_height = int(c.image_width / c.aspect_ratio);
    image_height = (image_height < 1) ? 1 : image_height;
    const int sqrt_spp = int(std::sqrt(c.samples_per_pixel));
    const double* lf = c.lookfrom;
    const double theta = c.vfov * pi / 180.0;
    const double h = std::tan(theta / 2);
    const double viewport_height = 2 * h * c.focus_dist;
    const double viewport_width = viewport_height * (double(c.image_width) / image_height);
    auto unit3 = [](const double* v, double* o) {
        const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double inv = 1 / len;
        o[0] = inv * v[0]; o[1] = inv * v[1]; o[2] = inv * v[2];
    };
    auto cross3 = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    double d[3] = {lf[0] - c.lookat[0], lf[1] - c.lookat[1], lf[2] - c.lookat[2]};
    double w[3], u[3], v[3], t[3];
    unit3(d, w);
    cross3(c.vup, w, t);
    unit3(t, u);
    cross3(w, u, v);
    double viewport_u[3], viewport_v[3];
    for (int k = 0; k < 3; k++) { viewport_u[k] = viewport_width * u[k]; viewport_v[k] = viewport_height * (-v[k]); }
    const double inv_w = 1.0 / c.image_width, inv_h = 1.0 / image_height;      // vec / int == (1/t) * vec
    for (int k = 0; k < 3; k++) {
        out.pixel_delta_u[k] = inv_w * viewport_u[k];
        out.pixel_delta_v[k] = inv_h * viewport_v[k];
    }
    const double half = 1.0 / 2;
    for (int k = 0; k < 3; k++) {
        const double upper_left = ((lf[k] - (c.focus_dist * w[k])) - half * viewport_u[k]) - half * viewport_v[k];
        out.pixel00_loc[k] = upper_left + 0.5 * (out.pixel_delta_u[k] + out.pixel_delta_v[k]);
        out.center[k] = lf[k];
        out.background[k] = c.background[k];
    }
    // camera.h:193-196: defocus_radius = focus_dist * tan(degrees_to_radians(defocus_angle / 2)); u * r == r * u
    const double defocus_radius = c.focus_dist * std::tan((c.defocus_angle / 2) * pi / 180.0);
    for (int k = 0; k < 3; k++) {
        out.defocus_disk_u[k] = defocus_radius * u[k];
        out.defocus_disk_v[k] = defocus_radius * v[k];
    }
    out.defocus_angle = c.defocus_angle;
    out.image_width = c.image_width;
    out.image_height = image_height;
    out.sqrt_spp = sqrt_spp;
    out.max_depth = c.max_depth;
    return std::string();
}

}  // namespace bdpt
