// accel_build.cpp — host builders for the acceleration structures the device path
// consumes: binned-SAH BVH2 with optional spatial splits (SBVH, parameter alpha),
// greedy BVH2 -> BVH4 collapse.  Restates the algorithm of the reference
// (src/bvh.cpp:46-154 build loop, :157-261 object splits, :264-610 spatial splits and
// clipping, :695-803 4-wide collapse) so that node arrays and the primIdx permutation
// come out in the same order: LIFO work stack with the right child on top, 8 bins over
// reference-box centres, strict '<' when comparing costs, split only when it beats the
// leaf cost or the node holds more than MIN_LEAF_PRIMS references.
//
// Parity note: the reference's builder cannot be compiled in this environment (it
// needs the Windows/GL/OpenCL template headers), so node-array parity with it is
// UNPINNED; traversal parity does not depend on it because the oracle and the HIP path
// consume the same arrays.
#include <atomic>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <cstring>
#include <future>
#include <mutex>
#include <utility>
#include "rt_host.h"

namespace rt355 {

// ---- small math (template/precomp.h:815-865 semantics: unfused, 1/sqrt normalise) ----
float  dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3 cross(const float3& a, const float3& b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
float  length(const float3& v) { return sqrtf(dot(v, v)); }
float3 normalize(const float3& v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }

static inline float lo(float a, float b) { return a < b ? a : b; } // _mm_min_ps / template min()
static inline float hi(float a, float b) { return a > b ? a : b; } // _mm_max_ps / template max()

void Aabb::Grow(const float3& p)
{
    bmin[0] = lo(bmin[0], p.x); bmin[1] = lo(bmin[1], p.y); bmin[2] = lo(bmin[2], p.z); bmin[3] = lo(bmin[3], 0.0f);
    bmax[0] = hi(bmax[0], p.x); bmax[1] = hi(bmax[1], p.y); bmax[2] = hi(bmax[2], p.z); bmax[3] = hi(bmax[3], 0.0f);
}
void Aabb::Grow(const RtFloat4& p) { Grow(float3(p)); }
void Aabb::Grow(const Aabb& b)
{
    for (int i = 0; i < 4; i++) { bmin[i] = lo(bmin[i], b.bmin[i]); bmax[i] = hi(bmax[i], b.bmax[i]); }
}
Aabb Aabb::Union(const Aabb& b) const { Aabb r = *this; r.Grow(b); return r; }
Aabb Aabb::Intersection(const Aabb& b) const
{
    Aabb r;
    for (int i = 0; i < 4; i++) { r.bmin[i] = hi(bmin[i], b.bmin[i]); r.bmax[i] = lo(bmax[i], b.bmax[i]); }
    return r;
}
float Aabb::Area() const
{
    float e0 = bmax[0] - bmin[0], e1 = bmax[1] - bmin[1], e2 = bmax[2] - bmin[2];
    return hi(0.0f, e0 * e1 + e0 * e2 + e1 * e2);
}

// ------------------------------------------------------------------ BVH2
BVH2::BVH2(std::vector<RtPrimitive>& prims, std::vector<RtBVHInstance>& blas) : blasNodes(blas), primitives_(prims) {}

uint32_t BVH2::Depth(uint32_t n) const
{
    const RtBVHNode2& node = bvhNodes[n];
    if (node.count > 0) return 0;
    uint32_t l = Depth(node.first), r = Depth(node.first + 1);
    return (l > r ? l : r) + 1;
}
uint32_t BVH2::Count(uint32_t n) const
{
    const RtBVHNode2& node = bvhNodes[n];
    return node.count > 0 ? node.count : Count(node.first) + Count(node.first + 1);
}
float BVH2::TotalCost(uint32_t n) const
{
    const RtBVHNode2& node = bvhNodes[n];
    return node.count > 0 ? CalculateNodeCost(node, node.count) : TotalCost(node.first) + TotalCost(node.first + 1);
}
float BVH2::CalculateNodeCost(const RtBVHNode2& node, uint32_t count) const
{
    float ex = node.aabbMax.x - node.aabbMin.x, ey = node.aabbMax.y - node.aabbMin.y, ez = node.aabbMax.z - node.aabbMin.z;
    return (float)count * (ex * ey + ey * ez + ez * ex);
}

BVH2::Refs BVH2::CreateBVHPrimData(int startIdx) const
{
    Refs refs;
    refs.reserve(primitives_.size() - startIdx);
    for (uint32_t i = (uint32_t)startIdx; i < primitives_.size(); i++) {
        const RtPrimitive& p = primitives_[i];
        BVHPrimData d;
        d.idx = i;
        if (p.objType == RT_PRIM_TRIANGLE) {
            d.box.Grow(p.obj.triangle.v0); d.box.Grow(p.obj.triangle.v1); d.box.Grow(p.obj.triangle.v2);
        } else if (p.objType == RT_PRIM_SPHERE) {
            float3 c(p.obj.sphere.pos);
            d.box.Grow(c + p.obj.sphere.r); d.box.Grow(c - p.obj.sphere.r);
        } // planes keep the empty box (reference behaviour: planes are not supported under a BVH)
        refs.push_back(d);
    }
    return refs;
}

void BVH2::UpdateNodeBounds(uint32_t nodeIdx, const Refs& refs)
{
    if (nodeIdx >= bvhNodes.size()) bvhNodes.resize((size_t)(bvhNodes.size() * 1.5));
    RtBVHNode2& n = bvhNodes[nodeIdx];
    n.aabbMin = RtFloat4{ RT_REALLYFAR, RT_REALLYFAR, RT_REALLYFAR, 0 };
    n.aabbMax = RtFloat4{ -RT_REALLYFAR, -RT_REALLYFAR, -RT_REALLYFAR, 0 };
    for (const BVHPrimData& r : refs) {
        n.aabbMin = RtFloat4{ fminf(n.aabbMin.x, r.box.bmin[0]), fminf(n.aabbMin.y, r.box.bmin[1]),
                              fminf(n.aabbMin.z, r.box.bmin[2]), fminf(n.aabbMin.w, r.box.bmin[3]) };
        n.aabbMax = RtFloat4{ fmaxf(n.aabbMax.x, r.box.bmax[0]), fmaxf(n.aabbMax.y, r.box.bmax[1]),
                              fmaxf(n.aabbMax.z, r.box.bmax[2]), fmaxf(n.aabbMax.w, r.box.bmax[3]) };
    }
}

void BVH2::BuildBLAS(bool statistics, int startIdx)
{
    auto t0 = std::chrono::steady_clock::now();
    RtBVHInstance inst;
    memset(&inst, 0, sizeof inst);
    inst.bvhIdx = rootNodeIdx_;
    inst.invT[0] = inst.invT[5] = inst.invT[10] = inst.invT[15] = 1.0f;
    blasNodes.push_back(inst);
    Refs refs = CreateBVHPrimData(startIdx);
    bvhNodes.resize(bvhNodes.size() + (primitives_.size() - startIdx) * 8);
    bvhNodes[rootNodeIdx_].count = (uint32_t)refs.size();
    nodesUsed_++;
    UpdateNodeBounds(rootNodeIdx_, refs);
    if (buildThreads > 1 && refs.size() > 2048) {
        const RtBVHNode2& rn = bvhNodes[rootNodeIdx_];
        float d0 = rn.aabbMax.x - rn.aabbMin.x, d1 = rn.aabbMax.y - rn.aabbMin.y, d2 = rn.aabbMax.z - rn.aabbMin.z;
        int budget = buildThreads - 1;
        TNode* tree = BuildSubtree(std::move(refs), hi(0.f, d0 * d1 + d0 * d2 + d1 * d2), 0, budget);
        FlattenLIFO(rootNodeIdx_, tree);
    } else
        BuildBVH(rootNodeIdx_, std::move(refs));
    if (statistics) {
        stat_build_time += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stat_node_count = nodesUsed_;
        uint32_t d = Depth(rootNodeIdx_);
        if (d > stat_depth) stat_depth = d;
        stat_sah_cost += TotalCost(rootNodeIdx_);
        stat_prim_count = (uint32_t)primitives_.size();
    }
    bvhNodes.resize(nodesUsed_);
    rootNodeIdx_ = nodesUsed_;
}

void BVH2::BuildBVH(uint32_t root, Refs data)
{
    const uint32_t blasRoot = rootNodeIdx_;
    std::vector<std::pair<uint32_t, Refs>> work;
    work.emplace_back(root, std::move(data));
    while (!work.empty()) {
        uint32_t nodeIdx = work.back().first;
        Refs refs = std::move(work.back().second);
        work.pop_back();

        int objectAxis = 0, spatialAxis = -1;
        float objectPos = 0, overlap = 0, spatialPos = RT_REALLYFAR, spatialCost = RT_REALLYFAR;
        float objectCost = FindBestObjectSplitPlane(objectAxis, objectPos, overlap, refs);
        float leafCost = CalculateNodeCost(bvhNodes[nodeIdx], (uint32_t)refs.size());
        const RtBVHNode2& rn = bvhNodes[blasRoot];
        float d0 = rn.aabbMax.x - rn.aabbMin.x, d1 = rn.aabbMax.y - rn.aabbMin.y, d2 = rn.aabbMax.z - rn.aabbMin.z;
        float rootArea = hi(0.f, d0 * d1 + d0 * d2 + d1 * d2);
        if (overlap / rootArea > alpha) spatialCost = FindBestSpatialSplitPlane(spatialAxis, spatialPos, refs);

        if (refs.size() <= RT_MIN_LEAF_PRIMS || (leafCost < objectCost && leafCost < spatialCost)) {
            RtBVHNode2& n = bvhNodes[nodeIdx];
            n.first = (uint32_t)primIdx.size();
            n.count = (uint32_t)refs.size();
            for (const BVHPrimData& r : refs) primIdx.push_back(r.idx);
            continue;
        }
        Refs left, right;
        if (objectCost < spatialCost) ObjectSplit(objectAxis, objectPos, refs, left, right);
        else { stat_spatial_splits++; SpatialSplit(spatialAxis, spatialPos, refs, left, right, stat_prims_clipped); }
        // Termination guard (deviation from the reference, which recurses without bound here): a
        // spatial split that hands every reference to one child or duplicates all of them into both
        // makes no progress and would be chosen again for the child -> close the node as a leaf.
        if (left.empty() || right.empty() || (left.size() >= refs.size() && right.size() >= refs.size())) {
            RtBVHNode2& n = bvhNodes[nodeIdx];
            n.first = (uint32_t)primIdx.size();
            n.count = (uint32_t)refs.size();
            for (const BVHPrimData& r : refs) primIdx.push_back(r.idx);
            stat_forced_leaves++;
            continue;
        }
        uint32_t leftId = nodesUsed_++, rightId = nodesUsed_++;
        UpdateNodeBounds(leftId, left);
        UpdateNodeBounds(rightId, right);
        bvhNodes[nodeIdx].first = leftId;
        bvhNodes[nodeIdx].count = 0;
        work.emplace_back(leftId, std::move(left));
        work.emplace_back(rightId, std::move(right)); // popped first
    }
}

// ---- parallel build --------------------------------------------------------------------
// A node's split decision depends only on its own references (and the BLAS root area / alpha), so subtrees can be built
// by independent tasks into a pointer tree.  Node ids and the primIdx order are then produced by FlattenLIFO, which
// replays the reference's work-stack order (children get the next two ids when their parent is popped, the right child
// is popped first, leaves append their references when popped: bvh.cpp:101-154) — the arrays are identical to the
// sequential build, index for index.
struct BVH2::TNode {
    float bmin[4], bmax[4];
    TNode* left = nullptr; TNode* right = nullptr;
    Refs refs;                    // leaf only
    ~TNode() { delete left; delete right; }
};
static std::mutex g_statMutex;
BVH2::TNode* BVH2::BuildSubtree(Refs refs, float rootArea, int depth, int& budget)
{
    TNode* n = new TNode();
    n->bmin[0] = n->bmin[1] = n->bmin[2] = RT_REALLYFAR; n->bmin[3] = 0;
    n->bmax[0] = n->bmax[1] = n->bmax[2] = -RT_REALLYFAR; n->bmax[3] = 0;
    for (const BVHPrimData& r : refs) for (int k = 0; k < 4; k++) { n->bmin[k] = fminf(n->bmin[k], r.box.bmin[k]); n->bmax[k] = fmaxf(n->bmax[k], r.box.bmax[k]); }
    int objectAxis = 0, spatialAxis = -1;
    float objectPos = 0, overlap = 0, spatialPos = RT_REALLYFAR, spatialCost = RT_REALLYFAR;
    float objectCost = FindBestObjectSplitPlane(objectAxis, objectPos, overlap, refs);
    float ex = n->bmax[0] - n->bmin[0], ey = n->bmax[1] - n->bmin[1], ez = n->bmax[2] - n->bmin[2];
    float leafCost = (float)(uint32_t)refs.size() * (ex * ey + ey * ez + ez * ex);
    if (overlap / rootArea > alpha) spatialCost = FindBestSpatialSplitPlane(spatialAxis, spatialPos, refs);
    if (refs.size() <= RT_MIN_LEAF_PRIMS || (leafCost < objectCost && leafCost < spatialCost)) { n->refs = std::move(refs); return n; }
    Refs left, right;
    uint32_t clipped = 0;
    bool spatial = !(objectCost < spatialCost);
    if (!spatial) ObjectSplit(objectAxis, objectPos, refs, left, right);
    else {
        SpatialSplit(spatialAxis, spatialPos, refs, left, right, clipped);
        std::lock_guard<std::mutex> lock(g_statMutex);
        stat_prims_clipped += clipped;
    }
    if (left.empty() || right.empty() || (left.size() >= refs.size() && right.size() >= refs.size())) {
        { std::lock_guard<std::mutex> lock(g_statMutex); stat_forced_leaves++; if (spatial) { /* the sequential build counts the split before it is discarded */ stat_spatial_splits++; } }
        n->refs = std::move(refs);
        return n;
    }
    if (spatial) { std::lock_guard<std::mutex> lock(g_statMutex); stat_spatial_splits++; }
    refs.clear(); refs.shrink_to_fit();
    bool fork = false;
    { std::lock_guard<std::mutex> lock(g_statMutex); if (budget > 0 && left.size() > 1024 && right.size() > 1024) { budget--; fork = true; } }
    if (fork) {
        auto fut = std::async(std::launch::async, [&, this]() { return BuildSubtree(std::move(left), rootArea, depth + 1, budget); });
        n->right = BuildSubtree(std::move(right), rootArea, depth + 1, budget);
        n->left = fut.get();
        std::lock_guard<std::mutex> lock(g_statMutex); budget++;
    } else {
        n->left = BuildSubtree(std::move(left), rootArea, depth + 1, budget);
        n->right = BuildSubtree(std::move(right), rootArea, depth + 1, budget);
    }
    return n;
}
void BVH2::FlattenLIFO(uint32_t root, TNode* tree)
{
    std::vector<std::pair<uint32_t, TNode*>> work;
    work.emplace_back(root, tree);
    auto setBounds = [&](uint32_t id, const TNode* t) {
        if (id >= bvhNodes.size()) bvhNodes.resize((size_t)(bvhNodes.size() * 1.5));
        bvhNodes[id].aabbMin = RtFloat4{ t->bmin[0], t->bmin[1], t->bmin[2], t->bmin[3] };
        bvhNodes[id].aabbMax = RtFloat4{ t->bmax[0], t->bmax[1], t->bmax[2], t->bmax[3] };
    };
    while (!work.empty()) {
        auto [id, t] = work.back();
        work.pop_back();
        if (!t->left) {
            bvhNodes[id].first = (uint32_t)primIdx.size();
            bvhNodes[id].count = (uint32_t)t->refs.size();
            for (const BVHPrimData& r : t->refs) primIdx.push_back(r.idx);
            continue;
        }
        uint32_t leftId = nodesUsed_++, rightId = nodesUsed_++;
        setBounds(leftId, t->left); setBounds(rightId, t->right);
        bvhNodes[id].first = leftId; bvhNodes[id].count = 0;
        work.emplace_back(leftId, t->left);
        work.emplace_back(rightId, t->right);
    }
    delete tree;
}

// ---- object splits -------------------------------------------------------------
float BVH2::FindBestObjectSplitPlane(int& axis, float& splitPos, float& overlap, const Refs& refs) const
{
    float best = RT_REALLYFAR;
    for (int a = 0; a < 3; a++) {
        float cmin = RT_REALLYFAR, cmax = -RT_REALLYFAR;
        for (const BVHPrimData& r : refs) { float c = r.box.Center(a); cmin = lo(cmin, c); cmax = hi(cmax, c); }
        if (cmin == cmax) continue;
        Aabb binBox[RT_BVH_BINS]; int binCount[RT_BVH_BINS] = { 0 };
        float scale = (float)RT_BVH_BINS / (cmax - cmin);
        for (const BVHPrimData& r : refs) {
            int b = (int)((r.box.Center(a) - cmin) * scale);
            if (b > RT_BVH_BINS - 1) b = RT_BVH_BINS - 1;
            binCount[b]++; binBox[b].Grow(r.box);
        }
        float lArea[RT_BVH_BINS - 1], rArea[RT_BVH_BINS - 1];
        Aabb  lBox[RT_BVH_BINS - 1], rBox[RT_BVH_BINS - 1];
        int   lCount[RT_BVH_BINS - 1], rCount[RT_BVH_BINS - 1];
        Aabb accL, accR; int sumL = 0, sumR = 0;
        for (int i = 0; i < RT_BVH_BINS - 1; i++) {
            sumL += binCount[i]; lCount[i] = sumL; accL.Grow(binBox[i]); lArea[i] = accL.Area(); lBox[i] = accL;
            int j = RT_BVH_BINS - 1 - i;
            sumR += binCount[j]; rCount[j - 1] = sumR; accR.Grow(binBox[j]); rArea[j - 1] = accR.Area(); rBox[j - 1] = accR;
        }
        scale = (cmax - cmin) / (float)RT_BVH_BINS;
        for (int i = 0; i < RT_BVH_BINS - 1; i++) {
            float cost = (float)lCount[i] * lArea[i] + (float)rCount[i] * rArea[i]; // 0*inf = NaN for empty sides: never '<'
            if (cost < best) {
                best = cost; axis = a; splitPos = cmin + scale * (float)(i + 1);
                overlap = lBox[i].Intersection(rBox[i]).Area();
            }
        }
    }
    return best;
}
void BVH2::ObjectSplit(int axis, float splitPos, const Refs& refs, Refs& left, Refs& right) const
{
    for (const BVHPrimData& r : refs) (r.box.Center(axis) <= splitPos ? left : right).push_back(r);
}

// ---- clipping ------------------------------------------------------------------
static float3 cutEdge(float3 p, float3 q, int axis, float plane) // src/bvh.cpp:288-301
{
    float3 s = p[axis] < q[axis] ? p : q, e = p[axis] < q[axis] ? q : p;
    float3 d = e - s;
    float f = (plane - s[axis]) / d[axis];
    return s + d * f;
}
bool BVH2::ClipTriangleToAABB(const Aabb& bounds, float3 v0, float3 v1, float3 v2, Aabb& out) const
{
    std::vector<float3> poly = { v0, v1, v2 }, next;
    for (int a = 0; a < 3; a++) for (int side = 0; side < 2; side++) {
        float plane = side == 0 ? bounds.bmin[a] : bounds.bmax[a];
        float sign = side == 0 ? 1.0f : -1.0f;
        next.clear();
        for (size_t i = 0; i < poly.size(); i++) {
            float3 cur = poly[i], nxt = poly[(i + 1) % poly.size()];
            bool inCur = (cur[a] - plane) * sign >= 0, inNxt = (nxt[a] - plane) * sign >= 0;
            if (inCur) next.push_back(cur);
            if (inCur != inNxt) next.push_back(cutEdge(cur, nxt, a, plane));
        }
        poly = next;
    }
    if (poly.size() < 3) return false;
    for (const float3& p : poly) out.Grow(p);
    return true;
}
static void sphereSlice(float3 pos, float d, int axis, float plane, float3 pts[4]) // src/bvh.cpp:388-437
{
    int n = 0;
    for (int axisL = 0; axisL < 3; axisL++) {
        if (axisL == axis) continue;
        int axisF = 0;
        while (axisF == axisL || axisF == axis) axisF++;
        float3 p;
        p[axis] = plane; p[axisL] = pos[axisL];
        float a = -2 * pos[axisF];
        float by = -2 * pos[axisL] * p[axisL];
        float cz = -2 * pos[axis] * p[axis];
        float y2 = p[axisL] * p[axisL], z2 = p[axis] * p[axis];
        float D = a * a - 4 * (y2 + z2 + by + cz + d);
        float s = sqrtf(D);
        p[axisF] = (-a + s) * 0.5f; pts[n++] = p;
        p[axisF] = (-a - s) * 0.5f; pts[n++] = p;
    }
}
bool BVH2::ClipSphereToAABB(const Aabb& bounds, float3 pos, float r, Aabb& out) const
{
    out.Grow(pos + r); out.Grow(pos - r);
    for (int a = 0; a < 3; a++) for (int side = 0; side < 2; side++) {
        float plane = side == 0 ? bounds.bmin[a] : bounds.bmax[a];
        float sign = side == 0 ? 1.0f : -1.0f;
        float farPos = pos[a] + r * sign;
        if (!(farPos * sign > plane * sign)) return false; // sphere entirely outside
        float nearPos = pos[a] - r * sign;
        if (nearPos * sign < plane * sign) {
            float d = pos.x * pos.x + pos.y * pos.y + pos.z * pos.z - r * r;
            float3 pts[4];
            sphereSlice(pos, d, a, plane, pts);
            Aabb tight;
            for (int k = 0; k < 4; k++) tight.Grow(pts[k]);
            float3 farVec = pos; farVec[a] = farPos;
            tight.Grow(farVec);
            out = out.Intersection(tight);
        }
    }
    return true;
}

// ---- spatial splits --------------------------------------------------------------
namespace {
struct SpatialBin {
    Aabb bounds; int entries = 0, exits = 0; float left = RT_REALLYFAR, right = -RT_REALLYFAR;
    SpatialBin merged(const SpatialBin& o) const
    {
        SpatialBin r;
        r.bounds = bounds.Union(o.bounds); r.entries = entries + o.entries; r.exits = exits + o.exits;
        r.left = lo(left, o.left); r.right = hi(right, o.right);
        return r;
    }
};
}
float BVH2::FindBestSpatialSplitPlane(int& axis, float& splitPos, const Refs& refs) const
{
    const int NB = RT_BVH_BINS;
    float best = RT_REALLYFAR;
    for (int a = 0; a < 3; a++) {
        float bmin = RT_REALLYFAR, bmax = -RT_REALLYFAR;
        for (const BVHPrimData& r : refs) { bmin = lo(bmin, r.box.bmin[a]); bmax = hi(bmax, r.box.bmax[a]); }
        if (bmin == bmax) continue;
        SpatialBin bins[RT_BVH_BINS];
        float scale = (float)NB / (bmax - bmin);
        for (int b = 0; b < NB; b++) {
            bins[b].left = bmin + (float)b * (1 / scale);
            bins[b].right = b == NB - 1 ? bmax : bmin + (float)(b + 1) * (1 / scale);
        }
        for (const BVHPrimData& r : refs) {
            const Aabb& box = r.box;
            int lb = (int)(scale * (box.bmin[a] - bmin)); if (lb > NB - 1) lb = NB - 1;
            int rb = (int)(scale * (box.bmax[a] - bmin)); if (rb > NB - 1) rb = NB - 1;
            while (box.bmin[a] <= bins[lb].left && lb > 0) lb--;
            while (box.bmin[a] > bins[lb].right && lb != NB - 1) lb++;
            while (box.bmax[a] < bins[rb].left && rb > 0) rb--;
            while (box.bmax[a] >= bins[rb].right && rb != NB - 1) rb++;
            if (lb == rb) {
                bins[lb].entries++; bins[rb].exits++; bins[lb].bounds.Grow(box);
                continue;
            }
            int first = NB, last = -1;
            const RtPrimitive& prim = primitives_[r.idx];
            for (int b = lb; b <= rb; b++) {
                Aabb slab = box, clipped;
                slab.bmin[a] = bins[b].left; slab.bmax[a] = bins[b].right;
                bool hit = false;
                if (prim.objType == RT_PRIM_TRIANGLE)
                    hit = ClipTriangleToAABB(slab, float3(prim.obj.triangle.v0), float3(prim.obj.triangle.v1), float3(prim.obj.triangle.v2), clipped);
                else if (prim.objType == RT_PRIM_SPHERE)
                    hit = ClipSphereToAABB(slab, float3(prim.obj.sphere.pos), prim.obj.sphere.r, clipped);
                if (hit) {
                    if (b < first) first = b;
                    if (b > last) last = b;
                    bins[b].bounds.Grow(clipped);
                }
            }
            if (first <= last) { bins[first].entries++; bins[last].exits++; }
        }
        SpatialBin prefix[RT_BVH_BINS], suffix[RT_BVH_BINS], accL, accR;
        for (int i = 0; i < NB; i++) {
            accL = accL.merged(bins[i]); prefix[i] = accL;
            accR = accR.merged(bins[NB - 1 - i]); suffix[NB - 1 - i] = accR;
        }
        for (int i = 0; i < NB - 1; i++) {
            const SpatialBin& L = prefix[i]; const SpatialBin& R = suffix[i + 1];
            if (L.entries == 0 || R.exits == 0) continue;
            float cost = (float)L.entries * L.bounds.Area() + (float)R.exits * R.bounds.Area();
            if (cost < best) { best = cost; axis = a; splitPos = L.right; }
        }
    }
    return best;
}
void BVH2::SpatialSplit(int axis, float splitPos, const Refs& refs, Refs& left, Refs& right, uint32_t& clippedCount) const
{
    for (const BVHPrimData& r : refs) {
        float mn = r.box.bmin[axis], mx = r.box.bmax[axis];
        if (mn < splitPos && mx > splitPos) {
            Aabb lclip = r.box, rclip = r.box, lout, rout;
            lclip.bmax[axis] = splitPos; rclip.bmin[axis] = splitPos;
            bool lok = false, rok = false;
            const RtPrimitive& prim = primitives_[r.idx];
            if (prim.objType == RT_PRIM_TRIANGLE) {
                float3 a(prim.obj.triangle.v0), b(prim.obj.triangle.v1), c(prim.obj.triangle.v2);
                lok = ClipTriangleToAABB(lclip, a, b, c, lout);
                rok = ClipTriangleToAABB(rclip, a, b, c, rout);
            } else if (prim.objType == RT_PRIM_SPHERE) {
                lok = ClipSphereToAABB(lclip, float3(prim.obj.sphere.pos), prim.obj.sphere.r, lout);
                rok = ClipSphereToAABB(rclip, float3(prim.obj.sphere.pos), prim.obj.sphere.r, rout);
            }
            clippedCount++;
            if (lok) left.push_back({ lout, r.idx });
            if (rok) right.push_back({ rout, r.idx });
        } else if (mx <= splitPos) left.push_back(r);
        else right.push_back(r);
    }
}

// ------------------------------------------------------------------ BVH4 (src/bvh.cpp:613-803)
BVH4::BVH4(BVH2& b) : bvh2(b) { Convert(); }

int BVH4::GetChildCount(const RtBVHNode4& n) const
{
    int c = 0;
    while (c < 4 && n.count[c] != RT_INVALID) c++;
    return c;
}
void BVH4::Convert()
{
    const std::vector<RtBVHNode2>& src = bvh2.bvhNodes;
    bvhNodes.assign(src.size(), RtBVHNode4{}); // same index space as the BVH2 array; leaf slots stay zero
    for (size_t i = 0; i < src.size(); i++) {
        if (src[i].count > 0) continue;
        RtBVHNode4& q = bvhNodes[i];
        for (int k = 0; k < 2; k++) {
            const RtBVHNode2& ch = src[src[i].first + k];
            q.aabbMin[k] = ch.aabbMin; q.aabbMax[k] = ch.aabbMax;
            if (ch.count > 0) { q.first[k] = (int32_t)ch.first; q.count[k] = (int32_t)ch.count; }
            else { q.first[k] = (int32_t)(src[i].first + k); q.count[k] = 0; }
        }
        for (int k = 2; k < 4; k++) q.first[k] = q.count[k] = RT_INVALID;
    }
    for (const RtBVHInstance& inst : bvh2.blasNodes) {
        uint32_t root = inst.bvhIdx;
        if (src[root].count > 0) { // a BLAS whose root is a leaf
            RtBVHNode4& q = bvhNodes[root];
            q.aabbMin[0] = src[root].aabbMin; q.aabbMax[0] = src[root].aabbMax;
            q.first[0] = (int32_t)src[root].first; q.count[0] = (int32_t)src[root].count;
            for (int k = 1; k < 4; k++) q.first[k] = q.count[k] = RT_INVALID;
        } else Collapse((int)root);
    }
}
void BVH4::Collapse(int index)
{
    RtBVHNode4& node = bvhNodes[index];
    for (;;) {
        int n = GetChildCount(node);
        float bestArea = -INFINITY; int pick = RT_INVALID;
        for (int i = 0; i < n; i++) {
            if (node.count[i] > 0) continue;
            int nc = GetChildCount(bvhNodes[node.first[i]]);
            if (!(n - 1 + nc <= 4)) continue;
            float dx = node.aabbMax[i].x - node.aabbMin[i].x, dy = node.aabbMax[i].y - node.aabbMin[i].y,
                  dz = node.aabbMax[i].z - node.aabbMin[i].z;
            float half = dx * dy + dy * dz + dz * dx;
            if (half > bestArea) { bestArea = half; pick = i; }
        }
        if (pick == RT_INVALID) break;
        const RtBVHNode4 child = bvhNodes[node.first[pick]];
        int nc = GetChildCount(child);
        node.aabbMin[pick] = child.aabbMin[0]; node.aabbMax[pick] = child.aabbMax[0];
        node.first[pick] = child.first[0]; node.count[pick] = child.count[0];
        for (int i = 1; i < nc; i++) {
            node.aabbMin[n - 1 + i] = child.aabbMin[i]; node.aabbMax[n - 1 + i] = child.aabbMax[i];
            node.first[n - 1 + i] = child.first[i]; node.count[n - 1 + i] = child.count[i];
        }
    }
    for (int i = 0; i < 4; i++) {
        if (node.count[i] == RT_INVALID) break;
        if (node.count[i] == 0) Collapse(node.first[i]);
    }
}
uint32_t BVH4::Depth(uint32_t idx) const
{
    const RtBVHNode4& n = bvhNodes[idx];
    uint32_t d = 0;
    for (int i = 0; i < 4; i++) if (n.count[i] == 0) { uint32_t c = Depth((uint32_t)n.first[i]) + 1; if (c > d) d = c; }
    return d;
}
uint32_t BVH4::Count(uint32_t idx) const
{
    const RtBVHNode4& n = bvhNodes[idx];
    uint32_t c = 0;
    for (int i = 0; i < 4; i++) {
        if (n.count[i] > 0) c += (uint32_t)n.count[i];
        else if (n.count[i] == 0) c += Count((uint32_t)n.first[i]);
    }
    return c;
}

// ------------------------------------------------------------------ TLAS (src/tlas.cpp:3-52)
TLAS::TLAS(BVH2& b) : bvh2_(b) { tlasNodes.assign(b.blasNodes.size() * 2, RtTLASNode{}); }

int TLAS::FindBestMatch(const int* list, int N, int A) const
{
    float smallest = RT_REALLYFAR; int best = -1;
    for (int B = 0; B < N; B++) {
        if (B == A) continue;
        const RtTLASNode& a = tlasNodes[list[A]]; const RtTLASNode& b = tlasNodes[list[B]];
        float ex = fmaxf(a.aabbMax.x, b.aabbMax.x) - fminf(a.aabbMin.x, b.aabbMin.x);
        float ey = fmaxf(a.aabbMax.y, b.aabbMax.y) - fminf(a.aabbMin.y, b.aabbMin.y);
        float ez = fmaxf(a.aabbMax.z, b.aabbMax.z) - fminf(a.aabbMin.z, b.aabbMin.z);
        float area = ex * ey + ey * ez + ez * ex;
        if (area < smallest) { smallest = area; best = B; }
    }
    return best;
}
// World-space bounds of an instance: the reference copies the BLAS root's OBJECT-space box into the TLAS leaf (tlas.cpp:15-17),
// which is only right for the identity transforms it ever uses (bvh.cpp:53-58; the one writer is commented out, scene.cpp:82).
// With a real transform the world ray would be slab-tested against the wrong box and the instance culled.  Identity instances
// keep the reference's copy bit for bit; for any other invT the eight corners of the root box are mapped by inverse(invT) and
// bounded, padded by a few ulp of the box size (the leaf test must never be tighter than the instance's own root test).
static bool is_identity(const float* T)
{
    static const float I[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    return memcmp(T, I, sizeof I) == 0 || (T[0] == 1 && T[5] == 1 && T[10] == 1 && T[1] == 0 && T[2] == 0 && T[3] == 0 && T[4] == 0 &&
                                           T[6] == 0 && T[7] == 0 && T[8] == 0 && T[9] == 0 && T[11] == 0);
}
static void instance_world_box(const float* invT, const RtBVHNode2& root, RtFloat4& mn, RtFloat4& mx)
{
    // inverse of the affine map p' = A p + t (rows 0-2 of invT; row-major, translation in cells 3/7/11): p = A^-1 (p' - t)
    const double a[3][3] = { { invT[0], invT[1], invT[2] }, { invT[4], invT[5], invT[6] }, { invT[8], invT[9], invT[10] } };
    const double t[3] = { invT[3], invT[7], invT[11] };
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (!(fabs(det) > 1e-30)) throw std::runtime_error("TLAS::Build: an instance transform is singular");
    double inv[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
        inv[r][c] = (a[r1][c1] * a[r2][c2] - a[r1][c2] * a[r2][c1]) / det;
    }
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    const float bx[2][3] = { { root.aabbMin.x, root.aabbMin.y, root.aabbMin.z }, { root.aabbMax.x, root.aabbMax.y, root.aabbMax.z } };
    for (int k = 0; k < 8; k++) {
        const double p[3] = { bx[k & 1][0] - t[0], bx[(k >> 1) & 1][1] - t[1], bx[(k >> 2) & 1][2] - t[2] };
        for (int r = 0; r < 3; r++) {
            const double w = inv[r][0] * p[0] + inv[r][1] * p[1] + inv[r][2] * p[2];
            lo[r] = std::min(lo[r], w); hi[r] = std::max(hi[r], w);
        }
    }
    float fl[3], fh[3];
    for (int r = 0; r < 3; r++) {
        const double pad = 1e-5 * (hi[r] - lo[r]) + 1e-6 * std::max(fabs(lo[r]), fabs(hi[r])) + 1e-30;
        fl[r] = nextafterf((float)(lo[r] - pad), -INFINITY); fh[r] = nextafterf((float)(hi[r] + pad), INFINITY);
    }
    mn = RtFloat4{ fl[0], fl[1], fl[2], 0.0f }; mx = RtFloat4{ fh[0], fh[1], fh[2], 0.0f };
}

void TLAS::Build()
{
    int slot[256], live = (int)bvh2_.blasNodes.size();
    // reference limit: nodeIdx[256] and 16-bit child ids (tlas.cpp:11, common.h:111-116); it overruns the array beyond that
    if (live > 256) throw std::runtime_error("TLAS::Build: " + std::to_string(live) + " BLAS instances, at most 256 are supported");
    if (live == 0) throw std::runtime_error("TLAS::Build: the scene has no BLAS (BuildBLAS comes first)");   // the reference reads slot[0] uninitialised here
    nodesUsed_ = 1;
    for (int i = 0; i < live; i++) {
        const RtBVHNode2& root = bvh2_.bvhNodes[bvh2_.blasNodes[i].bvhIdx];
        RtTLASNode& leaf = tlasNodes[nodesUsed_];
        if (is_identity(bvh2_.blasNodes[i].invT)) { leaf.aabbMin = root.aabbMin; leaf.aabbMax = root.aabbMax; }   // tlas.cpp:15-17, bit for bit
        else instance_world_box(bvh2_.blasNodes[i].invT, root, leaf.aabbMin, leaf.aabbMax);
        leaf.BLASidx = (uint32_t)i; leaf.leftRight = 0;
        slot[i] = (int)nodesUsed_++;
    }
    int A = 0, B = FindBestMatch(slot, live, A);
    while (live > 1) {
        int C = FindBestMatch(slot, live, B);
        if (A == C) {
            int ia = slot[A], ib = slot[B];
            RtTLASNode joined{};
            const RtTLASNode& na = tlasNodes[ia]; const RtTLASNode& nb = tlasNodes[ib];
            joined.leftRight = (uint32_t)ia + ((uint32_t)ib << 16);
            joined.aabbMin = RtFloat4{ fminf(na.aabbMin.x, nb.aabbMin.x), fminf(na.aabbMin.y, nb.aabbMin.y), fminf(na.aabbMin.z, nb.aabbMin.z), fminf(na.aabbMin.w, nb.aabbMin.w) };
            joined.aabbMax = RtFloat4{ fmaxf(na.aabbMax.x, nb.aabbMax.x), fmaxf(na.aabbMax.y, nb.aabbMax.y), fmaxf(na.aabbMax.z, nb.aabbMax.z), fmaxf(na.aabbMax.w, nb.aabbMax.w) };
            tlasNodes[nodesUsed_] = joined;
            slot[A] = (int)nodesUsed_++;
            slot[B] = slot[live - 1];
            B = FindBestMatch(slot, --live, A);
        } else { A = B; B = C; }
    }
    tlasNodes[0] = tlasNodes[slot[A]];
}

} // namespace rt355
