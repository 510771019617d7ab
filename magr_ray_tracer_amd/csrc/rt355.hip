// rt355.hip — C-ABI implementation (include/rt355.h): device memory, streams, launch
// sequence.  This is the replacement for the reference's OpenCL Kernel/Buffer dispatch in
// Renderer (src/renderer.cpp:64-94,142-263,289-301).  No CPU fallback exists here.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/rt355.h"
#include "rt355_kernels.h"

using namespace rt355dev;

// workgroups of 256 threads the hardware admits per CU whatever the occupancy query says: any kernel / kernels with <= 96 SGPRs
static constexpr int kAdmitAnySgpr = 6, kAdmit96Sgpr = 7;
static constexpr int kSpillCap = 20, kFitSeven = 22;   // LDS stack entries per lane: 22 x 1 KB per workgroup is the most with which seven workgroups share a CU's 160 KB; a spilling kernel keeps 20
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(RT_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// The device copy of a scene (uploaded arrays + derived layouts).  Contexts that render the same scene - sample-stream lanes, the
// row bands of one frame - can hold ONE copy (rt_share_scene): less HBM, and one working set in the L2s / Infinity Cache instead of one per context.
struct SceneBag {
    std::vector<void*> allocs;
    int device = 0;
    ~SceneBag() { (void)hipSetDevice(device); for (void* p : allocs) (void)hipFree(p); }
};

struct RtCtx {
    RtConfig cfg{};
    hipStream_t stream = nullptr;
    DevScene sc{};
    DevQueues q{};
    DevVariant var{};
    int nPix = 0, firstPixel = 0, gridMax = 0;
    bool sceneLoaded = false, ownAccum = true;
    std::shared_ptr<SceneBag> scene;      // shared by the contexts of rt_share_scene, freed with the last of them
    bool singleBlas = false;              // the TLAS root is a leaf
    std::vector<void*> queueAllocs;
    float* dFocus = nullptr;
    RtRay* dRayIO = nullptr; // debug import/export staging (lazy)
    uint64_t frames = 0, primaryRays = 0;
    // profiling
    struct Ev { hipEvent_t a, b; int stage; };
    std::vector<Ev> evPool; size_t evUsed = 0;
    RtStageTimes times{};
    int maxDepth2 = 0, tlasDepth = 0;
    int layout = 0;   // 0 = traverse the reference arrays as uploaded, 1 = derived pair/triangle-record layout
    bool persist = false;   // persistent-wavefront traversal (layout 1, single BLAS)
    bool persist4 = false;  // ... over the BVH4
    bool persistTlas = false;   // ... through a multi-BLAS TLAS (BVH2, layout 1): k_trace_persist_tlas
    bool spillStack = false;    // ... with the deep end of the traversal stacks in global memory (trees deeper than the LDS share of 7 workgroups per CU)
    uint32_t* dSpill = nullptr; size_t spillWords = 0;
    int xcdFirst = -1;      // the XCD this context's sparse queues start on (PersistTune.xcdFirst)
    int spillCap = kSpillCap;   // LDS entries per lane of a spilling kernel (RT355_SPILL_CAP: tests force the spill path with a tiny cap, >= 6)
    int nInterior = 0;          // records of the dense pair table (their ids must fit the 29-bit field of the tagged stack entries)
    bool cursorUsed[2 * (RT_MAX_BOUNCES + 2)] = {};   // work-queue heads consumed since the last k_begin_frame
    bool shadeRun[RT_MAX_BOUNCES + 1] = {};           // shade(b) launched since the last k_begin_frame
    bool generated = false;                           // generate launched since the last k_begin_frame
    int stackEntries = RT_BVH2_STACK, persistGrid = 0, persistGridConnect = 0;
    PersistTune tune{ 112, 24, 6, 8, 0, 0 }, tuneConnect{ 128, 32, 6, 16, 0, 0 }, tune4{ 64, 20, 6, 8, 0, 0 };   // extend (BVH2), connect, extend (BVH4): measured optima (tools/tune_extend.sh, tune_connect.sh, tune_persist.sh)
    float4* dPostF = nullptr; uchar4* dPostB = nullptr;   // post-processing outputs (lazy)
    int32_t* dSteps = nullptr;   // per-ray `steps` buffer, only bound while rt_debug_enable_steps is on
    int shadeTile = kTile;  // k_shade tile = workgroup size: kTile (512), or 256 for contexts that share the GPU (RtConfig.shade_blocks_per_cu > 0)
    int shadeGrid = 1024;   // workgroups of k_shade (what the CUs hold at once; the kernel does not depend on it); set in rt_create
};
enum { ST_GENERATE, ST_EXTEND, ST_SHADE, ST_COMPACT, ST_CONNECT, ST_ACCUM };

extern "C" const char* rt_last_error(void) { return g_err.c_str(); }
extern "C" int rt_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int rt_kernel_info(RtCtx* ctx, RtKernelInfo* out)
{
    if (!ctx || !out) return fail(RT_E_INVALID, "rt_kernel_info: null argument");
    if (!ctx->sceneLoaded) return fail(RT_E_INVALID, "rt_kernel_info: no scene uploaded");
    *out = RtKernelInfo{ ctx->layout, ctx->persist ? 1 : (ctx->persistTlas ? (ctx->spillStack ? 3 : 2) : 0), ctx->persist4 ? 1 : 0, ctx->spillStack ? ctx->spillCap : ctx->stackEntries, ctx->persistGrid, ctx->persistGridConnect,
                         ctx->shadeGrid, ctx->sc.nBlas };
    return RT_OK;
}

template <class T> static int dalloc(std::vector<void*>& bag, T** p, size_t count)
{
    void* v = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&v, bytes);
    if (e != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    bag.push_back(v);
    *p = (T*)v;
    return RT_OK;
}
static void free_bag(std::vector<void*>& bag) { for (void* p : bag) (void)hipFree(p); bag.clear(); }

// LDS traversal stack: one column per lane; sized at upload to what this scene's trees can need
// (never more than the reference kernels' 32 / 64 entries).
static size_t stack_bytes(const RtCtx* c) { return (size_t)c->stackEntries * kBlock * sizeof(uint32_t); }
// k_trace_persist_tlas keeps its pending TLAS siblings (<= one per level) on the same column; with spillStack only the first kSpillCap
// entries of a column live in LDS
static int tlas_stack_entries(const RtCtx* c) { return c->stackEntries + c->tlasDepth + 1; }
static constexpr int kBackupWords = 10;   // the world ray (O, D, 1/D) + the TLAS level's pruning distance, kept in LDS across an instance visit (PersistTune.backup)
static int tlas_lds_entries(const RtCtx* c) { return c->spillStack ? c->spillCap : tlas_stack_entries(c); }
static size_t tlas_stack_bytes(const RtCtx* c) { return (size_t)(tlas_lds_entries(c) + (c->tune.backup ? kBackupWords : 0)) * kBlock * sizeof(uint32_t); }

// ---- profiling brackets --------------------------------------------------------------
// Stage timing: a fixed ring of HIP event pairs on the context's stream.  Recording never forces a device sync: when the ring
// wraps, the oldest pair is harvested (it completed thousands of launches ago; if not, the HOST waits on that one event while the
// GPU keeps draining its queue).
static constexpr size_t kEvRing = 4096;
static void ev_account(RtCtx* c, RtCtx::Ev& e)
{
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) { (void)hipEventSynchronize(e.b); (void)hipEventElapsedTime(&ms, e.a, e.b); }
    switch (e.stage) {
    case ST_GENERATE: c->times.generate_ms += ms; c->times.generate_launches++; break;
    case ST_EXTEND:   c->times.extend_ms += ms; c->times.extend_launches++; break;
    case ST_SHADE:    c->times.shade_ms += ms; c->times.shade_launches++; break;
    case ST_COMPACT:  c->times.compact_ms += ms; c->times.compact_launches++; break;
    case ST_CONNECT:  c->times.connect_ms += ms; c->times.connect_launches++; break;
    case ST_ACCUM:    c->times.accumulate_ms += ms; c->times.accumulate_launches++; break;
    }
    e.stage = -1;
}
static void ev_init(RtCtx* c)   // rt_create, so that no event is created inside a timed region
{
    if (!c->cfg.profile || !c->evPool.empty()) return;
    c->evPool.resize(kEvRing);
    for (auto& e : c->evPool) { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); e.stage = -1; }
}
static inline bool ev_on(const RtCtx* c, int stage) { return c->cfg.profile >= 2 || (c->cfg.profile == 1 && stage == ST_EXTEND); }
static void ev_begin(RtCtx* c, int stage)
{
    if (!ev_on(c, stage)) return;
    ev_init(c);
    RtCtx::Ev& e = c->evPool[c->evUsed % kEvRing];
    if (e.stage >= 0) ev_account(c, e);   // ring wrapped: harvest the oldest pair first
    e.stage = stage;
    (void)hipEventRecord(e.a, c->stream);
}
static void ev_end(RtCtx* c, int stage)
{
    if (!ev_on(c, stage)) return;
    (void)hipEventRecord(c->evPool[c->evUsed % kEvRing].b, c->stream);
    c->evUsed++;
}
// A single launch is timed by the events the dispatch itself carries (hipExtLaunchKernelGGL: start/stop are the kernel's own begin/end
// timestamps, no marker packets on the stream - bracketing a launch with hipEventRecord costs ~2 % of a frame at 24 launches).
static RtCtx::Ev& ev_slot(RtCtx* c, int stage)
{
    ev_init(c);
    RtCtx::Ev& e = c->evPool[c->evUsed % kEvRing];
    if (e.stage >= 0) ev_account(c, e);   // ring wrapped: harvest the oldest pair first
    e.stage = stage;
    c->evUsed++;
    return e;
}
#define LAUNCHB(ctx, stage, kernel, grid, block, shmem, ...) do { \
        if (ev_on(ctx, stage)) { RtCtx::Ev& e_ = ev_slot(ctx, stage); hipExtLaunchKernelGGL(kernel, grid, dim3(block), (uint32_t)(shmem), (ctx)->stream, e_.a, e_.b, 0, __VA_ARGS__); } \
        else hipLaunchKernelGGL(kernel, grid, dim3(block), shmem, (ctx)->stream, __VA_ARGS__); \
    } while (0)
#define LAUNCH(ctx, stage, kernel, grid, shmem, ...) LAUNCHB(ctx, stage, kernel, grid, kBlock, shmem, __VA_ARGS__)
static void ev_collect(RtCtx* c) // call after a stream sync
{
    for (auto& e : c->evPool) if (e.stage >= 0) ev_account(c, e);
}

// ---- create / destroy ----------------------------------------------------------------
static void ctx_free(RtCtx* ctx);
extern "C" int rt_create(const RtConfig* cfg, RtCtx** out)
{
    if (!cfg || !out) return fail(RT_E_INVALID, "rt_create: null argument");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(RT_E_INVALID, "rt_create: bad resolution %dx%d", cfg->width, cfg->height);
    RtConfig c = *cfg;
    if (c.y1 <= 0) c.y1 = c.height;
    if (c.y0 < 0 || c.y0 >= c.y1 || c.y1 > c.height) return fail(RT_E_INVALID, "rt_create: bad row band [%d,%d)", c.y0, c.y1);
    if (c.max_bounces <= 0) c.max_bounces = RT_MAX_BOUNCES;
    if (c.max_bounces > RT_MAX_BOUNCES) return fail(RT_E_INVALID, "rt_create: max_bounces %d > %d", c.max_bounces, RT_MAX_BOUNCES);
    if ((c.shading != RT_SHADING_SIMPLE && c.shading != RT_SHADING_NEE) || (c.sampling != RT_SAMPLING_HEMISPHERE && c.sampling != RT_SAMPLING_COSINE) ||
        (c.accel != RT_ACCEL_BVH2 && c.accel != RT_ACCEL_BVH4))
        return fail(RT_E_INVALID, "rt_create: unknown kernel variant");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RT_E_DEVICE, "rt_create: no HIP device visible (this library has no CPU path)");
    if (c.device < 0 || c.device >= ndev) return fail(RT_E_INVALID, "rt_create: device %d out of range (%d visible)", c.device, ndev);
    HIPCHK(hipSetDevice(c.device));
    std::unique_ptr<RtCtx, void (*)(RtCtx*)> guard(new RtCtx(), ctx_free);   // released on success; any early return frees it
    RtCtx* ctx = guard.get();
    ctx->cfg = c;
    ctx->nPix = (c.y1 - c.y0) * c.width;
    ctx->firstPixel = c.y0 * c.width;
    ctx->gridMax = (ctx->nPix * std::max(1, c.max_bounces) + kBlock - 1) / kBlock; // connect may cover max_bounces*nPix shadow rays
    ctx->gridMax = std::max(ctx->gridMax, 4096);                                    // and the persistent grid (<= 256 CUs x 8 blocks)
    ctx->var = DevVariant{ c.shading, c.sampling, c.accel, c.russian_roulette ? 1 : 0, c.filter_fireflies ? 1 : 0, c.max_bounces };
    {   // k_shade: as many workgroups as the CUs hold at once.  Its ordered scan does not depend on that (tiles go by ticket to
        // running workgroups), so the size only matters for speed.  The occupancy query knows the VGPR, LDS and wave-slot limits
        // but not the SGPR file: 256-thread workgroups are admitted up to min(query, 8, 800 / (ceil16(sgprs) + 16)) per CU
        // (MI355X_MICROARCH.md, residency) = 6 for any kernel (<= 112 SGPRs), 7 up to 96 SGPRs.  k_shade: 2 workgroups of 512 threads (registers, 78 KB LDS).
        hipDeviceProp_t prop; int perCU = 0;
        HIPCHK(hipGetDeviceProperties(&prop, c.device));
        ctx->shadeTile = c.shade_blocks_per_cu > 0 ? 256 : kTile;
        if (const char* t = getenv("RT355_SHADE_TILE")) { const int v = atoi(t); if (v == 256 || v == 512) ctx->shadeTile = v; }
        if (c.shading == RT_SHADING_NEE) {
            if (ctx->shadeTile == 256) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_shade<true, 256>), 256, 0));
            else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_shade<true, kTile>), kTile, 0));
        } else {
            if (ctx->shadeTile == 256) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_shade<false, 256>), 256, 0));
            else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_shade<false, kTile>), kTile, 0));
        }
        perCU = std::min(perCU, kAdmitAnySgpr);
        ctx->shadeGrid = prop.multiProcessorCount * std::max(1, perCU);
        if (c.shade_blocks_per_cu > 0 && c.shade_blocks_per_cu <= 16) ctx->shadeGrid = prop.multiProcessorCount * c.shade_blocks_per_cu;
        if (const char* g = getenv("RT355_SHADE_PER_CU")) { int v = atoi(g); if (v > 0 && v <= 16) ctx->shadeGrid = prop.multiProcessorCount * v; }   // tuning / over-subscription tests
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(e));
    DevQueues& q = ctx->q;
    const size_t n = (size_t)ctx->nPix, nS = n * (size_t)c.max_bounces;
    int rc = RT_OK;
    auto& bag = ctx->queueAllocs;
#define QA(field, count) if (rc == RT_OK) rc = dalloc(bag, &q.field, count)
    const size_t nTiles = (n + kBlock - 1) / kBlock;
    for (int k = 0; k < 2; k++) { QA(O[k], n); QA(D[k], n); QA(inten[k], n); QA(meta[k], n); QA(tile[k], nTiles + 2); QA(super[k], nTiles / 64 + 2); QA(supAcc[k], nTiles / 64 + 2); }
    QA(hit, n);
    QA(sA, nS); QA(sB, nS); QA(sC, nS);
    QA(nRays, RT_MAX_BOUNCES + 2); QA(nShadow, RT_MAX_BOUNCES + 2); QA(cursor, kCursorWords); QA(shadeTicket, (size_t)(RT_MAX_BOUNCES + 1) * kTicketClasses * kTicketStride); QA(fault, 1);
    QA(seeds, n); QA(accum, (size_t)c.width * c.height);
    if (rc == RT_OK) rc = dalloc(bag, &ctx->dSteps, n);
    q.steps = nullptr;
    QA(ctrExtend, (size_t)ctx->gridMax * kCtrCols); QA(ctrConnect, (size_t)ctx->gridMax * kCtrCols);
#undef QA
    if (rc == RT_OK) rc = dalloc(bag, &ctx->dFocus, 1);
    if (rc != RT_OK) return rc;
    q.nPix = ctx->nPix; q.firstPixel = ctx->firstPixel; q.width = c.width; q.height = c.height;
    (void)hipMemsetAsync(q.accum, 0, sizeof(float4) * (size_t)c.width * c.height, ctx->stream);
    (void)hipMemsetAsync(q.nRays, 0, sizeof(int32_t) * (RT_MAX_BOUNCES + 2), ctx->stream);
    (void)hipMemsetAsync(q.nShadow, 0, sizeof(int32_t) * (RT_MAX_BOUNCES + 2), ctx->stream);
    (void)hipMemsetAsync(q.cursor, 0, sizeof(int32_t) * kCursorWords, ctx->stream);
    (void)hipMemsetAsync(q.shadeTicket, 0, sizeof(int32_t) * (size_t)(RT_MAX_BOUNCES + 1) * kTicketClasses * kTicketStride, ctx->stream);
    (void)hipMemsetAsync(q.fault, 0, sizeof(int32_t), ctx->stream);
    (void)hipMemsetAsync(q.ctrExtend, 0, sizeof(unsigned long long) * (size_t)ctx->gridMax * kCtrCols, ctx->stream);
    (void)hipMemsetAsync(q.ctrConnect, 0, sizeof(unsigned long long) * (size_t)ctx->gridMax * kCtrCols, ctx->stream);
    (void)hipMemsetAsync(q.seeds, 0, sizeof(uint32_t) * n, ctx->stream);
    for (int k = 0; k < 2; k++) {
        (void)hipMemsetAsync(q.tile[k], 0, sizeof(unsigned long long) * (nTiles + 2), ctx->stream);
        (void)hipMemsetAsync(q.super[k], 0, sizeof(unsigned long long) * (nTiles / 64 + 2), ctx->stream);
        (void)hipMemsetAsync(q.supAcc[k], 0, sizeof(unsigned long long) * (nTiles / 64 + 2), ctx->stream);
    }
    (void)hipMemsetAsync(q.hit, 0, sizeof(float4) * n, ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_init(ctx);
    *out = guard.release();
    return RT_OK;
}

static void ctx_free(RtCtx* ctx)   // every owned resource; safe on a partially constructed context
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->scene.reset(); free_bag(ctx->queueAllocs);
    if (ctx->dRayIO) (void)hipFree(ctx->dRayIO);
    if (ctx->dSpill) (void)hipFree(ctx->dSpill);
    if (ctx->dPostF) (void)hipFree(ctx->dPostF);
    if (ctx->dPostB) (void)hipFree(ctx->dPostB);
    for (auto& e : ctx->evPool) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" int rt_destroy(RtCtx* ctx)
{
    ctx_free(ctx);
    return RT_OK;
}

// ---- scene upload ----------------------------------------------------------------------
static int configure_traversal(RtCtx* ctx);
template <class T> static int upload(RtCtx* c, const T** dst, const T* src, size_t count)
{
    T* d = nullptr;
    int rc = dalloc(c->scene->allocs, &d, count);
    if (rc != RT_OK) return rc;
    if (count) HIPCHK(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = d;
    return RT_OK;
}
static int bvh2_depth(const RtBVHNode2* n, int32_t nNodes, uint32_t root)
{
    // iterative depth of the subtree at `root`; also validates child indices
    std::vector<std::pair<uint32_t, int>> st; st.push_back({ root, 0 });
    int best = 0; size_t visited = 0;
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        if (i >= (uint32_t)nNodes || ++visited > (size_t)nNodes * 2 + 2) return -1;
        if (d > best) best = d;
        if (n[i].count == 0) { st.push_back({ n[i].first, d + 1 }); st.push_back({ n[i].first + 1, d + 1 }); }
    }
    return best;
}
static int bvh4_stack_need(const RtBVHNode4* n, int32_t nNodes, uint32_t root)
{
    // worst-case live stack entries of the unordered 4-wide traversal (push every interior child, pop one)
    std::vector<std::pair<uint32_t, int>> st; st.push_back({ root, 0 });
    int best = 0; size_t visited = 0;
    while (!st.empty()) {
        auto [i, base] = st.back(); st.pop_back();
        if (i >= (uint32_t)nNodes || ++visited > (size_t)nNodes + 1) return -1;
        int kids = 0;
        for (int k = 0; k < 4; k++) if (n[i].first[k] != RT_INVALID && n[i].count[k] == 0) kids++;
        if (base + kids > best) best = base + kids;
        int pushed = 0;
        for (int k = 0; k < 4; k++) if (n[i].first[k] != RT_INVALID && n[i].count[k] == 0) {
            // child k is popped when the (kids-1-pushed) later siblings are gone: entries below it = base + pushed
            st.push_back({ (uint32_t)n[i].first[k], base + pushed });
            pushed++;
        }
    }
    return best;
}

// Host-side shape checks of a scene (no device needed; rt_upload_scene runs them first): a kernel that walks a malformed tree can
// fault the GPU.  Also sizes the LDS traversal stack and the texture padding.
static int validate_scene(int accel, const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                          const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                          const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                          const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas,
                          int* stackEntriesOut, int64_t* texPadOut, int* tlasDepthOut)
{
    if (accel != RT_ACCEL_BVH2 && accel != RT_ACCEL_BVH4) return fail(RT_E_INVALID, "rt_upload_scene: unknown accel %d", accel);
    if (!prims || nPrims <= 0 || !mats || nMats <= 0 || !bvhNodes || nNodes <= 0 || !primIdx || nIdx <= 0 || !tlas || nTlas <= 0 || !blas || nBlas <= 0)
        return fail(RT_E_INVALID, "rt_upload_scene: missing array (prims/materials/bvh/primIdx/tlas/blas are required)");
    if (nLights > 0 && !lights) return fail(RT_E_INVALID, "rt_upload_scene: nLights > 0 but lights == NULL");
    if (nTexels > 0 && !textures) return fail(RT_E_INVALID, "rt_upload_scene: nTexels > 0 but textures == NULL");
    // Child ids and instance ids of the TLAS travel as 15-bit values on the traversal stacks (bit 15 = leaf; the reference's own
    // TLASNode packs two 16-bit child ids into leftRight and TLAS::Build stops at 256 instances, tlas.cpp:11): larger trees are refused.
    if (nTlas > 0x8000 || nBlas > 0x8000) return fail(RT_E_UNSUPPORTED, "rt_upload_scene: %d TLAS nodes / %d instances exceed the 32768 the traversal stacks encode", nTlas, nBlas);
    for (int32_t i = 0; i < nPrims; i++) {
        if (prims[i].matIdx < 0 || prims[i].matIdx >= nMats) return fail(RT_E_INVALID, "primitive %d: matIdx %d out of range", i, prims[i].matIdx);
        if (prims[i].objType < 0 || prims[i].objType > 2) return fail(RT_E_INVALID, "primitive %d: objType %d", i, prims[i].objType);
    }
    for (int32_t i = 0; i < nIdx; i++) if (primIdx[i] >= (uint32_t)nPrims) return fail(RT_E_INVALID, "primIdx[%d] = %u out of range", i, primIdx[i]);
    for (int32_t i = 0; i < nLights; i++) if (lights[i] >= (uint32_t)nPrims) return fail(RT_E_INVALID, "lights[%d] out of range", i);
    int64_t texPad = 2; // the reference's lookup can land one row + one texel past a texture (uv == 1): pad the atlas
    for (int32_t i = 0; i < nMats; i++) if (mats[i].texIdx != -1) {
        if (mats[i].texIdx < 0 || mats[i].texW <= 0 || mats[i].texH <= 0 ||
            (int64_t)mats[i].texIdx + (int64_t)mats[i].texW * mats[i].texH > (int64_t)nTexels)
            return fail(RT_E_INVALID, "material %d: texture window exceeds the atlas", i);
        texPad = std::max<int64_t>(texPad, (int64_t)mats[i].texW + 2);
    }
    for (int32_t i = 0; i < nTlas; i++) {
        const uint32_t lr = tlas[i].leftRight;
        if (lr == 0) { if (tlas[i].BLASidx >= (uint32_t)nBlas) return fail(RT_E_INVALID, "tlas node %d: BLASidx out of range", i); }
        else if ((lr & 0xffffu) >= (uint32_t)nTlas || (lr >> 16) >= (uint32_t)nTlas) return fail(RT_E_INVALID, "tlas node %d: child out of range", i);
    }
    int tlasDepth = 0;
    {   // walk the TLAS from node 0: a back reference would make traverse_tlas spin forever, and its private stack holds
        // RT_TLAS_STACK entries (the ordered descent keeps at most one pending sibling per level, so depth bounds the stack)
        std::vector<std::pair<uint32_t, int>> st; st.push_back({ 0u, 0 });
        size_t visited = 0;
        while (!st.empty()) {
            auto [i, d] = st.back(); st.pop_back();
            if (++visited > (size_t)nTlas) return fail(RT_E_INVALID, "tlas: a node is reachable twice (cycle or shared child)");
            tlasDepth = std::max(tlasDepth, d);
            const uint32_t lr = tlas[i].leftRight;
            if (lr != 0) { st.push_back({ lr & 0xffffu, d + 1 }); st.push_back({ lr >> 16, d + 1 }); }
        }
        if (tlasDepth > RT_TLAS_STACK) return fail(RT_E_UNSUPPORTED, "tlas: depth %d exceeds the %d-entry traversal stack", tlasDepth, RT_TLAS_STACK);
    }
    // The reference kernels give BVH2 32 and BVH4 64 stack entries (bvh.cl:15,57) and overflow silently beyond that
    // (SBVH trees at alpha = 0 do get deeper than 32); this library sizes the LDS stack to the tree, up to 64 entries.
    const int stackCap = RT_BVH4_STACK;
    int stackNeed = 1;
    for (int32_t b = 0; b < nBlas; b++) {
        if (blas[b].bvhIdx >= (uint32_t)nNodes) return fail(RT_E_INVALID, "instance %d: bvhIdx out of range", b);
        int need = accel == RT_ACCEL_BVH4 ? bvh4_stack_need((const RtBVHNode4*)bvhNodes, nNodes, blas[b].bvhIdx)
                                          : bvh2_depth((const RtBVHNode2*)bvhNodes, nNodes, blas[b].bvhIdx);
        if (need < 0) return fail(RT_E_INVALID, "instance %d: malformed BVH (child index out of range or cycle)", b);
        if (need > stackCap) return fail(RT_E_UNSUPPORTED, "instance %d: traversal needs %d stack entries, at most %d are supported", b, need, stackCap);
        stackNeed = std::max(stackNeed, need);
    }
    if (accel == RT_ACCEL_BVH2) {
        const RtBVHNode2* n2 = (const RtBVHNode2*)bvhNodes;
        for (int32_t i = 0; i < nNodes; i++) if (n2[i].count > 0 && (uint64_t)n2[i].first + n2[i].count > (uint64_t)nIdx)
            return fail(RT_E_INVALID, "bvh node %d: leaf range exceeds primIdx", i);
    } else {
        const RtBVHNode4* n4 = (const RtBVHNode4*)bvhNodes;
        for (int32_t i = 0; i < nNodes; i++) for (int k = 0; k < 4; k++) if (n4[i].first[k] != RT_INVALID && n4[i].count[k] > 0 &&
            (int64_t)n4[i].first[k] + n4[i].count[k] > (int64_t)nIdx) return fail(RT_E_INVALID, "bvh4 node %d: leaf range exceeds primIdx", i);
    }
    if (stackEntriesOut) *stackEntriesOut = std::min(stackCap, std::max(stackNeed + 1, 6)); // >= 6: flush_counters reuses 36 words of it
    if (texPadOut) *texPadOut = texPad;
    if (tlasDepthOut) *tlasDepthOut = tlasDepth;
    return RT_OK;
}
extern "C" int rt_validate_scene(int32_t accel, const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                                 const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                                 const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                                 const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas)
{
    return validate_scene(accel, prims, nPrims, mats, nMats, textures, nTexels, lights, nLights, bvhNodes, nNodes, primIdx, nIdx, tlas, nTlas, blas, nBlas,
                          nullptr, nullptr, nullptr);
}

extern "C" int rt_upload_scene(RtCtx* ctx, const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                               const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                               const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                               const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_upload_scene: null context");
    int stackEntries = RT_BVH2_STACK, tlasDepth = 0, nInterior = 0; int64_t texPad = 2;
    {   // nothing of the context changes until the arrays have passed (a failed upload leaves the bound scene usable)
        const int vrc = validate_scene(ctx->cfg.accel, prims, nPrims, mats, nMats, textures, nTexels, lights, nLights, bvhNodes, nNodes, primIdx, nIdx,
                                       tlas, nTlas, blas, nBlas, &stackEntries, &texPad, &tlasDepth);
        if (vrc != RT_OK) return vrc;
    }
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->scene = std::make_shared<SceneBag>();   // (a copy shared with other contexts lives on with them)
    ctx->scene->device = ctx->cfg.device;
    ctx->sceneLoaded = false; ctx->persist = false; ctx->persist4 = false; ctx->layout = 0;   // nothing usable until this upload has succeeded
    ctx->sc = DevScene{};
    DevScene sc{};
    int rc = upload(ctx, &sc.prims, prims, (size_t)nPrims);
    if (rc == RT_OK) rc = upload(ctx, &sc.mats, mats, (size_t)nMats);
    if (rc == RT_OK) { // zero-padded atlas (see texPad above)
        float4* t = nullptr;
        rc = dalloc(ctx->scene->allocs, &t, (size_t)nTexels + (size_t)texPad);
        if (rc == RT_OK) {
            HIPCHK(hipMemset(t, 0, sizeof(float4) * ((size_t)nTexels + (size_t)texPad)));
            if (nTexels) HIPCHK(hipMemcpy(t, textures, sizeof(float4) * (size_t)nTexels, hipMemcpyHostToDevice));
            sc.tex = t;
        }
    }
    if (rc == RT_OK) rc = upload(ctx, &sc.lights, lights, (size_t)nLights);
    if (rc == RT_OK) {
        if (ctx->cfg.accel == RT_ACCEL_BVH4) rc = upload(ctx, &sc.bvh4, (const RtBVHNode4*)bvhNodes, (size_t)nNodes);
        else rc = upload(ctx, &sc.bvh2, (const RtBVHNode2*)bvhNodes, (size_t)nNodes);
    }
    if (rc == RT_OK) rc = upload(ctx, &sc.primIdx, primIdx, (size_t)nIdx);
    if (rc == RT_OK) rc = upload(ctx, &sc.tlas, tlas, (size_t)nTlas);
    if (rc == RT_OK) rc = upload(ctx, &sc.blas, blas, (size_t)nBlas);
    if (rc == RT_OK) { // dense shading records (k_shade): geometric normal + material id + type per primitive
        std::vector<float4> recs((size_t)nPrims);
        for (int32_t i = 0; i < nPrims; i++) {
            const RtPrimitive& p = prims[i];
            const RtFloat4 N = p.objType == RT_PRIM_TRIANGLE ? p.obj.triangle.N : (p.objType == RT_PRIM_PLANE ? p.obj.plane.N : RtFloat4{ 0, 0, 0, 0 });
            uint32_t tag = ((uint32_t)p.objType << 28) | ((uint32_t)p.matIdx & 0x07ffffffu) | (std::signbit(N.w) ? 0x08000000u : 0u);
            // a triangle/plane normal with a non-zero w lane cannot be represented: mark it like a sphere (reference-layout path)
            if (p.objType != RT_PRIM_SPHERE && N.w != 0.0f) tag = ((uint32_t)RT_PRIM_SPHERE << 28) | ((uint32_t)p.matIdx & 0x07ffffffu);
            float w; memcpy(&w, &tag, 4);
            recs[(size_t)i] = make_float4(N.x, N.y, N.z, w);
        }
        rc = upload(ctx, &sc.shadeRecs, recs.data(), recs.size());
    }
    if (rc == RT_OK) { // light records (k_shade, NEE): the first 64 bytes of the light's Primitive, {objType, area}, its material's emittance
        std::vector<float4> lr(std::max<size_t>((size_t)nLights, 1) * 8, make_float4(0, 0, 0, 0));
        for (int32_t i = 0; i < nLights; i++) {
            const RtPrimitive& p = prims[lights[i]];
            memcpy(&lr[(size_t)i * 8], &p.obj, 64);
            float t; int32_t ty = p.objType; memcpy(&t, &ty, 4);
            lr[(size_t)i * 8 + 4] = make_float4(t, p.area, 0, 0);
            const RtFloat4& e = mats[p.matIdx].emittance;
            lr[(size_t)i * 8 + 5] = make_float4(e.x, e.y, e.z, e.w);
        }
        rc = upload(ctx, &sc.lightRecs, lr.data(), lr.size());
    }
    // Derived layout 1 (rt355_kernels.h, traverse_bvh2_packed): only for BVH2, when the encodings fit.
    ctx->layout = 0;
    if (rc == RT_OK && ctx->cfg.accel == RT_ACCEL_BVH2 && ctx->cfg.extend_variant != 1 && nIdx < (1 << 24)) {
        const RtBVHNode2* n2 = (const RtBVHNode2*)bvhNodes;
        bool fits = true;
        for (int32_t i = 0; i < nNodes && fits; i++) if (n2[i].count > 127) fits = false;
        if (fits) {
            // Interior nodes are renumbered breadth-first, BLAS by BLAS, and stored densely: the reference array interleaves leaves
            // and interior nodes (children are allocated in pairs), so a table indexed by the reference's node id would be half
            // holes; breadth-first puts the top levels of the (first) tree, which every ray visits, into the first records.
            std::vector<uint32_t> newId((size_t)nNodes, 0xffffffffu), order;
            order.reserve((size_t)nNodes / 2 + 1);
            for (int32_t b = 0; b < nBlas; b++) {
                const uint32_t root = blas[b].bvhIdx;
                if (n2[root].count > 0 || newId[root] != 0xffffffffu) continue;
                size_t head = order.size();
                newId[root] = (uint32_t)order.size(); order.push_back(root);
                for (; head < order.size(); head++) {
                    const uint32_t i = order[head];
                    for (uint32_t c = n2[i].first; c <= n2[i].first + 1; c++)
                        if (n2[c].count == 0 && newId[c] == 0xffffffffu) { newId[c] = (uint32_t)order.size(); order.push_back(c); }
                }
            }
            auto entry = [&](uint32_t i) { return n2[i].count > 0 ? (0x80000000u | (n2[i].count << 24) | n2[i].first) : newId[i]; };
            auto f2u = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
            std::vector<float4> pairs(std::max<size_t>(order.size(), 1) * 4, make_float4(0, 0, 0, 0));
            for (size_t k = 0; k < order.size(); k++) {
                const uint32_t i = order[k];
                const RtBVHNode2& a = n2[n2[i].first]; const RtBVHNode2& b = n2[n2[i].first + 1];
                pairs[k * 4 + 0] = make_float4(a.aabbMin.x, a.aabbMin.y, a.aabbMin.z, a.aabbMax.x);
                pairs[k * 4 + 1] = make_float4(a.aabbMax.y, a.aabbMax.z, b.aabbMin.x, b.aabbMin.y);
                pairs[k * 4 + 2] = make_float4(b.aabbMin.z, b.aabbMax.x, b.aabbMax.y, b.aabbMax.z);
                pairs[k * 4 + 3] = make_float4(f2u(entry(n2[i].first)), f2u(entry(n2[i].first + 1)), 0, 0);
            }
            std::vector<float4> recs((size_t)nIdx * 3);
            for (int32_t s = 0; s < nIdx; s++) {
                const RtPrimitive& p = prims[primIdx[s]];
                const RtTriangle& t = p.obj.triangle;
                const bool plain = p.objType == RT_PRIM_TRIANGLE && t.v0.w == 0.0f && t.v1.w == 0.0f && t.v2.w == 0.0f;
                // v0 and the edges v1 - v0, v2 - v0: the first two operations of the reference's triangle test (primitives.cl:49-50), done
                // here once with the same IEEE subtraction (this file is built with -ffp-contract=off like the kernels)
                const float e1x = t.v1.x - t.v0.x, e1y = t.v1.y - t.v0.y, e1z = t.v1.z - t.v0.z;
                const float e2x = t.v2.x - t.v0.x, e2y = t.v2.y - t.v0.y, e2z = t.v2.z - t.v0.z;
                recs[(size_t)s * 3 + 0] = make_float4(t.v0.x, t.v0.y, t.v0.z, e1x);
                recs[(size_t)s * 3 + 1] = make_float4(e1y, e1z, e2x, e2y);
                recs[(size_t)s * 3 + 2] = make_float4(e2z, f2u(primIdx[s]), f2u(plain ? 0u : 1u), 0);
            }
            std::vector<uint32_t> roots((size_t)nBlas);
            for (int32_t b = 0; b < nBlas; b++) roots[b] = entry(blas[b].bvhIdx);
            rc = upload(ctx, &sc.pairs, pairs.data(), pairs.size());
            if (rc == RT_OK) rc = upload(ctx, &sc.triRecs, recs.data(), recs.size());
            if (rc == RT_OK) rc = upload(ctx, &sc.rootEntry, roots.data(), roots.size());
            if (rc == RT_OK) { ctx->layout = 1; nInterior = (int)order.size(); }
        }
    }
    if (rc == RT_OK && ctx->cfg.accel == RT_ACCEL_BVH4 && ctx->cfg.extend_variant != 1 && nIdx < (1 << 24)) {
        const RtBVHNode4* n4 = (const RtBVHNode4*)bvhNodes;
        bool fits = true;
        for (int32_t i = 0; i < nNodes && fits; i++) for (int k = 0; k < 4; k++) if (n4[i].first[k] != RT_INVALID && n4[i].count[k] > 127) fits = false;
        if (fits) {
            auto f2u = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
            // The collapse (bvh.cpp:695-803) leaves the absorbed BVH2 nodes in the array: only the nodes still reachable from a BLAS
            // root are kept, renumbered breadth-first and stored densely (on the bench scene 1 node in 4 is alive).
            std::vector<uint32_t> newId((size_t)nNodes, 0xffffffffu), order;
            for (int32_t b = 0; b < nBlas; b++) {
                const uint32_t root = blas[b].bvhIdx;
                if (newId[root] != 0xffffffffu) continue;
                size_t head = order.size();
                newId[root] = (uint32_t)order.size(); order.push_back(root);
                for (; head < order.size(); head++) {
                    const uint32_t i = order[head];
                    for (int k = 0; k < 4; k++) {
                        if (n4[i].first[k] == RT_INVALID || n4[i].count[k] > 0) continue;
                        const uint32_t c = (uint32_t)n4[i].first[k];
                        if (newId[c] == 0xffffffffu) { newId[c] = (uint32_t)order.size(); order.push_back(c); }
                    }
                }
            }
            std::vector<float4> quads(std::max<size_t>(order.size(), 1) * 8, make_float4(0, 0, 0, 0));
            for (size_t q = 0; q < order.size(); q++) {
                const uint32_t i = order[q];
                float b[24]; uint32_t e[4];
                for (int k = 0; k < 4; k++) {
                    const RtFloat4& mn = n4[i].aabbMin[k]; const RtFloat4& mx = n4[i].aabbMax[k];
                    b[k * 6 + 0] = mn.x; b[k * 6 + 1] = mn.y; b[k * 6 + 2] = mn.z; b[k * 6 + 3] = mx.x; b[k * 6 + 4] = mx.y; b[k * 6 + 5] = mx.z;
                    if (n4[i].first[k] == RT_INVALID) e[k] = 0xffffffffu;
                    else if (n4[i].count[k] > 0) e[k] = 0x80000000u | ((uint32_t)n4[i].count[k] << 24) | (uint32_t)n4[i].first[k];
                    else e[k] = newId[(uint32_t)n4[i].first[k]];
                }
                for (int v = 0; v < 6; v++) quads[q * 8 + v] = make_float4(b[v * 4], b[v * 4 + 1], b[v * 4 + 2], b[v * 4 + 3]);
                quads[q * 8 + 6] = make_float4(f2u(e[0]), f2u(e[1]), f2u(e[2]), f2u(e[3]));
            }
            std::vector<uint32_t> roots((size_t)nBlas);
            for (int32_t b = 0; b < nBlas; b++) roots[b] = newId[blas[b].bvhIdx];
            if (rc == RT_OK) rc = upload(ctx, &sc.rootEntry, roots.data(), roots.size());
            std::vector<float4> recs((size_t)nIdx * 3);
            for (int32_t s = 0; s < nIdx; s++) {
                const RtPrimitive& p = prims[primIdx[s]];
                const RtTriangle& t = p.obj.triangle;
                const bool plain = p.objType == RT_PRIM_TRIANGLE && t.v0.w == 0.0f && t.v1.w == 0.0f && t.v2.w == 0.0f;
                // v0 and the edges v1 - v0, v2 - v0: the first two operations of the reference's triangle test (primitives.cl:49-50), done
                // here once with the same IEEE subtraction (this file is built with -ffp-contract=off like the kernels)
                const float e1x = t.v1.x - t.v0.x, e1y = t.v1.y - t.v0.y, e1z = t.v1.z - t.v0.z;
                const float e2x = t.v2.x - t.v0.x, e2y = t.v2.y - t.v0.y, e2z = t.v2.z - t.v0.z;
                recs[(size_t)s * 3 + 0] = make_float4(t.v0.x, t.v0.y, t.v0.z, e1x);
                recs[(size_t)s * 3 + 1] = make_float4(e1y, e1z, e2x, e2y);
                recs[(size_t)s * 3 + 2] = make_float4(e2z, f2u(primIdx[s]), f2u(plain ? 0u : 1u), 0);
            }
            rc = upload(ctx, &sc.quads, quads.data(), quads.size());
            if (rc == RT_OK) rc = upload(ctx, &sc.triRecs, recs.data(), recs.size());
            if (rc == RT_OK) ctx->layout = 1;
        }
    }
    if (rc == RT_OK) { // derived TLAS records and instance records (traverse_tlas / traverse_instance)
        auto f2u = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        auto enc = [&](uint32_t n) { return tlas[n].leftRight == 0 ? (0x80000000u | tlas[n].BLASidx) : n; };
        std::vector<float4> tp((size_t)nTlas * 4, make_float4(0, 0, 0, 0));
        for (int32_t i = 0; i < nTlas; i++) {
            const uint32_t lr = tlas[i].leftRight;
            if (lr == 0) continue;
            const RtTLASNode& a = tlas[lr & 0xffffu]; const RtTLASNode& b = tlas[lr >> 16];
            tp[(size_t)i * 4 + 0] = make_float4(a.aabbMin.x, a.aabbMin.y, a.aabbMin.z, a.aabbMax.x);
            tp[(size_t)i * 4 + 1] = make_float4(a.aabbMax.y, a.aabbMax.z, b.aabbMin.x, b.aabbMin.y);
            tp[(size_t)i * 4 + 2] = make_float4(b.aabbMin.z, b.aabbMax.x, b.aabbMax.y, b.aabbMax.z);
            tp[(size_t)i * 4 + 3] = make_float4(f2u(enc(lr & 0xffffu)), f2u(enc(lr >> 16)), 0, 0);
        }
        std::vector<uint32_t> roots((size_t)nBlas, 0u);
        if (ctx->layout == 1) HIPCHK(hipMemcpy(roots.data(), sc.rootEntry, sizeof(uint32_t) * (size_t)nBlas, hipMemcpyDeviceToHost));
        std::vector<float4> ir((size_t)nBlas * 4);
        for (int32_t b = 0; b < nBlas; b++) {
            const float* T = blas[b].invT;
            ir[(size_t)b * 4 + 0] = make_float4(T[0], T[1], T[2], T[3]);
            ir[(size_t)b * 4 + 1] = make_float4(T[4], T[5], T[6], T[7]);
            ir[(size_t)b * 4 + 2] = make_float4(T[8], T[9], T[10], T[11]);
            ir[(size_t)b * 4 + 3] = make_float4(f2u(roots[(size_t)b]), f2u(blas[b].bvhIdx), 0, 0);
        }
        rc = upload(ctx, &sc.tlasPairs, tp.data(), tp.size());
        if (rc == RT_OK) rc = upload(ctx, &sc.instRecs, ir.data(), ir.size());
        sc.tlasRoot = enc(0);
        // the same records with the children in the tagged encoding of k_trace_persist_tlas (TLAS interior / instance ids on the BLAS stack)
        auto encP = [&](uint32_t n) { return tlas[n].leftRight == 0 ? (kTagInst | tlas[n].BLASidx) : (kTagTlas | n); };
        for (int32_t i = 0; i < nTlas; i++) {
            const uint32_t lr = tlas[i].leftRight;
            if (lr != 0) tp[(size_t)i * 4 + 3] = make_float4(f2u(encP(lr & 0xffffu)), f2u(encP(lr >> 16)), 0, 0);
        }
        if (rc == RT_OK) rc = upload(ctx, &sc.tlasPairsP, tp.data(), tp.size());
        sc.tlasRootP = encP(0);
    }
    if (rc != RT_OK) { ctx->scene.reset(); ctx->sceneLoaded = false; return rc; }
    sc.nLights = nLights; sc.nPrims = nPrims; sc.nBlas = nBlas; sc.nTex = nTexels;
    ctx->singleBlas = tlas[0].leftRight == 0;
    ctx->sc = sc;
    ctx->stackEntries = stackEntries; ctx->tlasDepth = tlasDepth; ctx->nInterior = nInterior;
    rc = configure_traversal(ctx);
    if (rc != RT_OK) { ctx->scene.reset(); return rc; }
    ctx->sceneLoaded = true;
    return RT_OK;
}

// What a context derives from its configuration once it has a scene: which traversal kernels run and how their persistent grids are sized.
static int configure_traversal(RtCtx* ctx)
{
    // persistent-wavefront traversal: layout 1 and a TLAS whose root is a leaf (one BLAS)
    ctx->persist = ctx->layout == 1 && ctx->cfg.accel == RT_ACCEL_BVH2 && ctx->singleBlas && ctx->cfg.extend_variant != 2;
    ctx->persist4 = ctx->layout == 1 && ctx->cfg.accel == RT_ACCEL_BVH4 && ctx->singleBlas && ctx->cfg.extend_variant != 2;
    // ... and through a TLAS with several BLAS (BASELINE config 5): TLAS entries ride on the BLAS stack column, so the TLAS must be shallow
    // (<= 8 levels: <= 256 instances in a balanced tree) and the pair-table ids must leave the three tag bits free; extend_variant 4 keeps
    // the one-ray-per-lane nested loops (A/B runs)
    ctx->persistTlas = ctx->layout == 1 && ctx->cfg.accel == RT_ACCEL_BVH2 && !ctx->singleBlas && ctx->cfg.extend_variant != 2 && ctx->cfg.extend_variant != 4 &&
                       ctx->tlasDepth <= 8 && ctx->nInterior < (1 << 29);
    // deep trees (an SBVH at alpha = 0: config 5's second BLAS has 63 levels): a full LDS column per lane would leave two workgroups per
    // CU, so the column is capped and its deep end spills to global memory (rt355_kernels.h, stk_push / stk_pop)
    // the world ray of a lane waits in LDS while the lane is inside an instance (10 words per lane) instead of being fetched back from the
    // queue on the way out: one global round trip less per instance visit (config 5: 1.5 visits among a ray's dozen events).  The stack
    // column's LDS share shrinks accordingly (12 + 10 words per lane keep seven workgroups per CU).  RT355_TLAS_BACKUP=0 switches it off.
    const bool backup = !(getenv("RT355_TLAS_BACKUP") && atoi(getenv("RT355_TLAS_BACKUP")) == 0);
    ctx->tune.backup = ctx->tuneConnect.backup = ctx->persistTlas && backup ? 1 : 0;   // (the LDS size of the occupancy query below depends on it)
    ctx->spillCap = backup ? kSpillCap - kBackupWords + 2 : kSpillCap;
    bool forceSpill = false;
    if (const char* t = getenv("RT355_SPILL_CAP")) { const int v = atoi(t); if (v >= 6 && v <= 64) { ctx->spillCap = v; forceSpill = true; } }
    // (columns of up to 22 entries stay whole in LDS, backup beside them: 5-6 workgroups per CU without the spill branches beat 7 with them,
    // 1,923 against 1,849 and 1,327 against 1,310 M samples/s on two-BLAS scenes of 16 and 22 entries - tools/middepth_tlas.py)
    ctx->spillStack = ctx->persistTlas && (tlas_stack_entries(ctx) > kFitSeven || forceSpill) && tlas_stack_entries(ctx) > ctx->spillCap &&
                      !(getenv("RT355_NO_SPILL") && atoi(getenv("RT355_NO_SPILL")));
    if (ctx->persistTlas && !ctx->spillStack && tlas_stack_entries(ctx) > RT_BVH4_STACK + 9) ctx->persistTlas = false;
    if (ctx->persist || ctx->persist4 || ctx->persistTlas) {
        int perCU = 0; hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ctx->cfg.device));
        if (ctx->persist) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_persist<false>, kBlock, stack_bytes(ctx)));
        else if (ctx->persistTlas && ctx->spillStack) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (k_trace_persist_tlas<false, false, true>), kBlock, tlas_stack_bytes(ctx)));
        else if (ctx->persistTlas) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_persist_tlas<false>, kBlock, tlas_stack_bytes(ctx)));
        else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_persist4<false>, kBlock, stack_bytes(ctx)));
        // the closest-hit instantiations use ~90 SGPRs, the any-hit ones ~100: the hardware admits 7 resp. 6 workgroups per CU where
        // the occupancy query may say more (see rt_create); a surplus workgroup would strand its static first chunk until another exits
        ctx->persistGrid = std::min(ctx->gridMax, std::max(1, std::min(perCU, kAdmit96Sgpr)) * prop.multiProcessorCount);
        ctx->persistGridConnect = std::min(ctx->gridMax, std::max(1, std::min(perCU, kAdmitAnySgpr)) * prop.multiProcessorCount);
        if (ctx->cfg.persist_blocks_per_cu > 0) {
            const int d = std::min(ctx->cfg.persist_blocks_per_cu, std::max(1, perCU));
            ctx->persistGrid = std::min(ctx->persistGrid, std::min(ctx->gridMax, d * prop.multiProcessorCount));
            ctx->persistGridConnect = std::min(ctx->persistGridConnect, std::min(ctx->gridMax, d * prop.multiProcessorCount));
            ctx->tune.leafK = 16;   // contexts sharing the GPU: hold triangle events back until 16 lanes wait on a leaf (+1 % with three lanes, -0.6 % alone)
        } else {
            // a context with the GPU to itself: chunks after the first are dealt round-robin too (no atomic, no round trip per dequeue):
            // 716 -> 725 M samples/s; with three contexts sharing the GPU the dynamic queue is 0.9 % better (profiles/r02_fixed_chunks.txt)
            ctx->tune.fixedChunks = ctx->tuneConnect.fixedChunks = ctx->tune4.fixedChunks = 1;
        }
        if (const char* t = getenv("RT355_TUNE")) { // "chunk,refill,inner,leafK[,blocksPerCU]" (tuning aid)
            int a = 0, b = 0, c = 0, l = 0, d = 0;
            int k = sscanf(t, "%d,%d,%d,%d,%d", &a, &b, &c, &l, &d);
            if (k >= 4 && a > 0 && b > 0 && b <= 64 && c > 0 && l > 0 && l <= 64) ctx->tune = ctx->tuneConnect = ctx->tune4 = PersistTune{ a, b, c, l, 0, 0 };
            if (k == 5 && d > 0) ctx->persistGrid = ctx->persistGridConnect = std::min(ctx->gridMax, std::min(d, std::max(1, perCU)) * prop.multiProcessorCount);
        }
        if (const char* t = getenv("RT355_CONNECT_BLOCKS")) { const int d = atoi(t); if (d > 0) ctx->persistGridConnect = std::min(ctx->gridMax, std::min(d, std::max(1, perCU)) * prop.multiProcessorCount); }   // (lab)
        if (const char* t = getenv("RT355_FIXED_CHUNKS")) { int a = 0, b = 0; if (sscanf(t, "%d,%d", &a, &b) == 2) { ctx->tune.fixedChunks = ctx->tune4.fixedChunks = a; ctx->tuneConnect.fixedChunks = b; } }   // extend, connect (tuning aid)
        if (const char* t = getenv("RT355_TUNE_CONNECT")) { // same fields, connect launches only
            int a = 0, b = 0, c = 0, l = 0;
            if (sscanf(t, "%d,%d,%d,%d", &a, &b, &c, &l) == 4 && a > 0 && b > 0 && b <= 64 && c > 0 && l > 0 && l <= 64) ctx->tuneConnect = PersistTune{ a, b, c, l, 0, 0 };
        }
    }
    ctx->tune.backup = ctx->tuneConnect.backup = 0;
    if (ctx->persistTlas) {
        // Multi-BLAS scenes so far are open scenes whose rays take a dozen events (config 5: 1 TLAS visit, 1.5 instance entries, 7.7 box
        // pairs, 1.9 triangles per ray): extend runs the kernel's one-ray-per-lane branch over every queue (measured per bounce at 4K:
        // 522 / 446 / 198 us against 654 / 562 / 194 through the event loop and 730 / 643 / 237 through the nested loops at two
        // workgroups per CU), connect - unoccluded shadow rays cross the whole scene - the event loop (646 against 690 / 1,418 us).
        // RT355_TLAS_FLAT="e,c" overrides (A/B runs).  profiles/r03_config5_per_bounce.txt
        ctx->tune.flat = 1; ctx->tuneConnect.flat = 0;
        ctx->tune.backup = ctx->tuneConnect.backup = backup ? 1 : 0;   // (set here, after RT355_TUNE has been parsed: that assignment resets the struct)
        if (const char* t = getenv("RT355_TLAS_FLAT")) { int a = 0, b = 0; if (sscanf(t, "%d,%d", &a, &b) == 2) { ctx->tune.flat = a; ctx->tuneConnect.flat = b; } }
    }
    // sparse queues (the one-ray-per-lane branches) stay on as few XCDs as hold them at 1,024 rays each, so that their rays share an L2
    // (EXPERIMENTS.md (54): 16,384 before the thinning below made spreading the better default); contexts start on different XCDs.
    // (set here, after RT355_TUNE has been parsed: that assignment resets the struct)
    {
        static std::atomic<int> serial{ 0 };
        if (ctx->xcdFirst < 0) ctx->xcdFirst = serial.fetch_add(1) & 7;
        int rays = 1024;
        if (const char* t = getenv("RT355_XCD_RAYS")) rays = std::max(0, atoi(t));
        ctx->tune.xcdRays = ctx->tuneConnect.xcdRays = ctx->tune4.xcdRays = rays;
        ctx->tune.xcdFirst = ctx->tuneConnect.xcdFirst = ctx->tune4.xcdFirst = ctx->xcdFirst;
        // ... and on few lanes of every participating wave when they hold at most 16 rays per wave (sparse_slot; RT355_THIN=0: off)
        int thin = 16;
        if (const char* t = getenv("RT355_THIN")) thin = std::min(64, std::max(0, atoi(t)));
        ctx->tune.thin = ctx->tuneConnect.thin = ctx->tune4.thin = thin;
    }
    ctx->q.spill = nullptr; ctx->q.spillStride = 0; ctx->q.stackCap = 0;
    ctx->q.tlasLdsEntries = ctx->persistTlas ? (uint32_t)tlas_lds_entries(ctx) : 0u;
    if (ctx->spillStack) {   // every SPILL launch runs on a persistent grid (bounce 0 too), so the global columns are bounded by the grids
        const size_t stride = (size_t)std::max(ctx->persistGrid, ctx->persistGridConnect) * kBlock;
        const size_t words = stride * (size_t)(tlas_stack_entries(ctx) - ctx->spillCap);
        if (words > ctx->spillWords) {
            if (ctx->dSpill) (void)hipFree(ctx->dSpill);
            ctx->dSpill = nullptr; ctx->spillWords = 0;
            if (hipMalloc((void**)&ctx->dSpill, words * sizeof(uint32_t)) != hipSuccess) return fail(RT_E_NOMEM, "hipMalloc of the spill stacks (%zu bytes) failed", words * sizeof(uint32_t));
            ctx->spillWords = words;
        }
        ctx->q.spill = ctx->dSpill; ctx->q.spillStride = (uint32_t)stride; ctx->q.stackCap = (uint32_t)ctx->spillCap;
    }
    return RT_OK;
}

// A second context on the same device renders the scene `from` holds: it takes the device copy (uploaded arrays and derived layouts)
// instead of uploading its own.  Both contexts must agree on what the derived layout depends on (accel, extend_variant).
extern "C" int rt_share_scene(RtCtx* ctx, RtCtx* from)
{
    if (!ctx || !from) return fail(RT_E_INVALID, "rt_share_scene: null context");
    if (ctx == from) return RT_OK;
    if (!from->sceneLoaded) return fail(RT_E_INVALID, "rt_share_scene: the source context has no scene");
    if (ctx->cfg.device != from->cfg.device) return fail(RT_E_INVALID, "rt_share_scene: contexts on different devices (%d, %d)", ctx->cfg.device, from->cfg.device);
    if (ctx->cfg.accel != from->cfg.accel || ctx->cfg.extend_variant != from->cfg.extend_variant)
        return fail(RT_E_INVALID, "rt_share_scene: the contexts differ in accel / extend_variant, which the derived layout depends on");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->sceneLoaded = false;
    ctx->scene = from->scene;
    ctx->sc = from->sc;
    ctx->layout = from->layout; ctx->maxDepth2 = from->maxDepth2; ctx->stackEntries = from->stackEntries; ctx->singleBlas = from->singleBlas; ctx->tlasDepth = from->tlasDepth; ctx->nInterior = from->nInterior;
    const int rc = configure_traversal(ctx);
    if (rc != RT_OK) { ctx->scene.reset(); return rc; }
    ctx->sceneLoaded = true;
    return RT_OK;
}

// ---- seeds / accumulator ---------------------------------------------------------------
extern "C" int rt_set_seeds(RtCtx* ctx, const uint32_t* seeds, int64_t n)
{
    if (!ctx || !seeds) return fail(RT_E_INVALID, "rt_set_seeds: null argument");
    if (n != ctx->nPix) return fail(RT_E_INVALID, "rt_set_seeds: expected %d seeds (band pixels), got %lld", ctx->nPix, (long long)n);
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(ctx->q.seeds, seeds, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
    return RT_OK;
}
extern "C" int rt_seed_default(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_seed_default: null context");
    std::vector<uint32_t> s((size_t)ctx->nPix);
    uint32_t x = 0x12345678u; // template/template.cpp:711
    auto next = [&x]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int64_t i = 0; i < ctx->firstPixel; i++) next();
    for (auto& v : s) v = next();
    return rt_set_seeds(ctx, s.data(), (int64_t)s.size());
}
extern "C" int rt_get_seeds(RtCtx* ctx, uint32_t* out, int64_t n)
{
    if (!ctx || !out || n != ctx->nPix) return fail(RT_E_INVALID, "rt_get_seeds: bad argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, ctx->q.seeds, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_bind_accum(RtCtx* ctx, void* devicePtr)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_bind_accum: null context");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (devicePtr) { ctx->q.accum = (float4*)devicePtr; ctx->ownAccum = false; }
    else if (!ctx->ownAccum) {
        float4* a = nullptr;
        int rc = dalloc(ctx->queueAllocs, &a, (size_t)ctx->cfg.width * ctx->cfg.height);
        if (rc != RT_OK) return rc;
        HIPCHK(hipMemset(a, 0, sizeof(float4) * (size_t)ctx->cfg.width * ctx->cfg.height));
        ctx->q.accum = a; ctx->ownAccum = true;
    }
    return RT_OK;
}
extern "C" void* rt_accum_device_ptr(RtCtx* ctx) { return ctx ? (void*)ctx->q.accum : nullptr; }
extern "C" void* rt_stream(RtCtx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int rt_reset(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_reset: null context");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(k_reset, dim3((ctx->nPix + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, ctx->q.accum, ctx->firstPixel, ctx->nPix);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

// ---- stages --------------------------------------------------------------------------------
static int need_scene(RtCtx* ctx, const char* who)
{
    if (!ctx) return fail(RT_E_INVALID, "%s: null context", who);
    if (!ctx->sceneLoaded) return fail(RT_E_INVALID, "%s: no scene uploaded", who);
    return RT_OK;
}
static inline dim3 grid_for(int n) { return dim3((unsigned)std::max(1, (n + kBlock - 1) / kBlock)); }

static void frame_state_reset(RtCtx* ctx)
{
    memset(ctx->cursorUsed, 0, sizeof ctx->cursorUsed);
    memset(ctx->shadeRun, 0, sizeof ctx->shadeRun);
    ctx->generated = false;
}
extern "C" int rt_stage_begin_frame(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_stage_begin_frame: null context");
    hipLaunchKernelGGL(k_begin_frame, dim3(1), dim3(256), 0, ctx->stream, ctx->q);
    HIPCHK(hipGetLastError());
    frame_state_reset(ctx);
    return RT_OK;
}
static int generate(RtCtx* ctx, const RtCamera* cam, const RtSettings* s, int beginFrame)
{
    if (beginFrame) frame_state_reset(ctx);
    LAUNCH(ctx, ST_GENERATE, k_generate, grid_for(ctx->nPix), 0, ctx->q, *cam, s ? s->antiAliasing : 1, beginFrame);
    HIPCHK(hipGetLastError());
    ctx->primaryRays += (uint64_t)ctx->nPix;
    ctx->generated = true;
    return RT_OK;
}
extern "C" int rt_stage_generate(RtCtx* ctx, const RtCamera* cam, const RtSettings* s)
{
    if (!ctx || !cam) return fail(RT_E_INVALID, "rt_stage_generate: null argument");
    return generate(ctx, cam, s, 0);
}
extern "C" int rt_stage_extend(RtCtx* ctx, int32_t bounce, int32_t renderBVH)
{
    int rc = need_scene(ctx, "rt_stage_extend"); if (rc) return rc;
    if (bounce < 0 || bounce > ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_stage_extend: bounce %d outside [0, %d]", bounce, ctx->cfg.max_bounces);
    if (ctx->persist || ctx->persist4 || ctx->persistTlas) { // a queue head is good for one launch per frame; re-arm it if this stage is run again
        if (ctx->cursorUsed[bounce]) HIPCHK(hipMemsetAsync(ctx->q.cursor + bounce, 0, sizeof(int32_t), ctx->stream));
        ctx->cursorUsed[bounce] = true;
    }
    const bool wantSteps = renderBVH != 0 || ctx->q.steps != nullptr;   // only then does the event loop keep the per-ray `steps`
    // bounce 0: primary rays are coherent and finish together, refilling buys nothing -> one ray per lane
    if (ctx->persistTlas) {
        // bounce 0 with one workgroup per 256 rays: the kernel's short-queue branch = the nested one-ray-per-lane loops (coherent primary rays)
        const dim3 g = bounce > 0 || ctx->spillStack ? dim3(ctx->persistGrid) : grid_for(ctx->nPix);
        // bounce 0 through the one-ray-per-lane branch: wave-uniform node records through the scalar cache (RT355_COHERENT=0: A/B runs)
        static const int cohOn = getenv("RT355_COHERENT") ? atoi(getenv("RT355_COHERENT")) : 1;   // (2: every bounce - lab)
        const bool coh = cohOn && (bounce == 0 || cohOn == 2) && (ctx->tune.flat || !ctx->spillStack);
        if (ctx->spillStack) {
            if (wantSteps) LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false, true, true>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
            else if (coh) LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false, false, true, true>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
            else LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false, false, true>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        }
        else if (wantSteps) LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false, true>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        else if (coh) LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false, false, false, true>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        else LAUNCH(ctx, ST_EXTEND, (k_trace_persist_tlas<false>), g, tlas_stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
    } else if (ctx->persist4)
        LAUNCH(ctx, ST_EXTEND, (k_trace_persist4<false>), bounce > 0 ? dim3(ctx->persistGrid) : grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune4);
    else if (ctx->persist && (bounce > 0 || ctx->cfg.extend_variant == 3)) {
        if (wantSteps) LAUNCH(ctx, ST_EXTEND, (k_trace_persist<false, false, true>), dim3(ctx->persistGrid), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        else LAUNCH(ctx, ST_EXTEND, (k_trace_persist<false>), dim3(ctx->persistGrid), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
    } else if (ctx->persist && ctx->cfg.extend_variant != 5)
        // bounce 0 through the same kernel with one workgroup per 256 rays: its "queue not longer than the grid" branch is the plain
        // one-ray-per-lane loop without the TLAS code of k_extend (60 instead of 86 VGPRs: 8 instead of 5 waves per SIMD)
    {
        // bounce 0: wave-uniform node records come through the scalar cache (traverse_bvh2_packed_coherent; RT355_COHERENT=0 switches it off for A/B runs)
        static const bool coherent = !(getenv("RT355_COHERENT") && atoi(getenv("RT355_COHERENT")) == 0);
        if (coherent && bounce == 0) LAUNCH(ctx, ST_EXTEND, (k_trace_persist<false, true>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        else if (wantSteps) LAUNCH(ctx, ST_EXTEND, (k_trace_persist<false, false, true>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
        else LAUNCH(ctx, ST_EXTEND, (k_trace_persist<false>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, bounce, renderBVH, ctx->tune);
    }
    else if (ctx->cfg.accel == RT_ACCEL_BVH4 && ctx->layout == 1)
        LAUNCH(ctx, ST_EXTEND, (k_extend<RT_ACCEL_BVH4, 1>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, renderBVH);
    else if (ctx->cfg.accel == RT_ACCEL_BVH4)
        LAUNCH(ctx, ST_EXTEND, (k_extend<RT_ACCEL_BVH4, 0>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, renderBVH);
    else if (ctx->layout == 1)
        LAUNCH(ctx, ST_EXTEND, (k_extend<RT_ACCEL_BVH2, 1>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, renderBVH);
    else
        LAUNCH(ctx, ST_EXTEND, (k_extend<RT_ACCEL_BVH2, 0>), grid_for(ctx->nPix), stack_bytes(ctx), ctx->sc, ctx->q, bounce, renderBVH);
    HIPCHK(hipGetLastError());
    return RT_OK;
}
extern "C" int rt_stage_shade(RtCtx* ctx, int32_t bounce)
{
    int rc = need_scene(ctx, "rt_stage_shade"); if (rc) return rc;
    // the shadow queue and the counter rows are sized by cfg.max_bounces, not by the compile-time maximum
    if (bounce < 0 || bounce >= ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_stage_shade: bounce %d outside [0, %d)", bounce, ctx->cfg.max_bounces);
    // The scan state of bounce b is armed by generate (b = 0) or by shade(b-1); re-arm it by hand when this
    // stage is run out of sequence (stage-level API used by the tests) or twice for the same bounce.
    if (ctx->shadeRun[bounce] || (bounce > 0 && !ctx->shadeRun[bounce - 1]) || (bounce == 0 && !ctx->generated)) {
        const size_t nTiles = ((size_t)ctx->nPix + kBlock - 1) / kBlock;
        HIPCHK(hipMemsetAsync(ctx->q.tile[bounce & 1], 0, sizeof(unsigned long long) * (nTiles + 2), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->q.super[bounce & 1], 0, sizeof(unsigned long long) * (nTiles / 64 + 2), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->q.supAcc[bounce & 1], 0, sizeof(unsigned long long) * (nTiles / 64 + 2), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->q.shadeTicket + (size_t)bounce * kTicketClasses * kTicketStride, 0, sizeof(int32_t) * (size_t)kTicketClasses * kTicketStride, ctx->stream));
    }
    const int tileSz = ctx->shadeTile;
    const dim3 sg((unsigned)std::max(1, std::min(ctx->shadeGrid, (ctx->nPix + tileSz - 1) / tileSz)));
    if (ctx->cfg.shading == RT_SHADING_NEE) {
        if (tileSz == 256) LAUNCHB(ctx, ST_SHADE, (k_shade<true, 256>), sg, 256, 0, ctx->sc, ctx->q, ctx->var, bounce);
        else LAUNCHB(ctx, ST_SHADE, (k_shade<true, kTile>), sg, kTile, 0, ctx->sc, ctx->q, ctx->var, bounce);
    } else {
        if (tileSz == 256) LAUNCHB(ctx, ST_SHADE, (k_shade<false, 256>), sg, 256, 0, ctx->sc, ctx->q, ctx->var, bounce);
        else LAUNCHB(ctx, ST_SHADE, (k_shade<false, kTile>), sg, kTile, 0, ctx->sc, ctx->q, ctx->var, bounce);
    }
    ctx->shadeRun[bounce] = true;
    HIPCHK(hipGetLastError());
    return RT_OK;
}
extern "C" int rt_stage_connect(RtCtx* ctx, int32_t b0, int32_t b1)
{
    int rc = need_scene(ctx, "rt_stage_connect"); if (rc) return rc;
    if (b0 < 0 || b1 < b0 || b1 >= ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_stage_connect: bounce range [%d,%d] outside [0, %d)", b0, b1, ctx->cfg.max_bounces);
    const int cap = ctx->nPix * (b1 - b0 + 1);
    if (ctx->persist || ctx->persist4 || ctx->persistTlas) {
        const int ci = (RT_MAX_BOUNCES + 2) + b0;
        if (ctx->cursorUsed[ci]) HIPCHK(hipMemsetAsync(ctx->q.cursor + ci, 0, sizeof(int32_t), ctx->stream));
        ctx->cursorUsed[ci] = true;
    }
    if (ctx->persistTlas && ctx->spillStack)
        LAUNCH(ctx, ST_CONNECT, (k_trace_persist_tlas<true, false, true>), dim3(ctx->persistGridConnect), tlas_stack_bytes(ctx), ctx->sc, ctx->q, b0, b1, 0, ctx->tuneConnect);
    else if (ctx->persistTlas)
        LAUNCH(ctx, ST_CONNECT, (k_trace_persist_tlas<true>), dim3(ctx->persistGridConnect), tlas_stack_bytes(ctx), ctx->sc, ctx->q, b0, b1, 0, ctx->tuneConnect);
    else if (ctx->persist4)
        LAUNCH(ctx, ST_CONNECT, (k_trace_persist4<true>), dim3(ctx->persistGridConnect), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1, 0, ctx->tuneConnect);
    else if (ctx->persist)
        LAUNCH(ctx, ST_CONNECT, (k_trace_persist<true>), dim3(ctx->persistGridConnect), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1, 0, ctx->tuneConnect);
    else if (ctx->cfg.accel == RT_ACCEL_BVH4 && ctx->layout == 1)
        LAUNCH(ctx, ST_CONNECT, (k_connect<RT_ACCEL_BVH4, 1>), grid_for(cap), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1);
    else if (ctx->cfg.accel == RT_ACCEL_BVH4)
        LAUNCH(ctx, ST_CONNECT, (k_connect<RT_ACCEL_BVH4, 0>), grid_for(cap), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1);
    else if (ctx->layout == 1)
        LAUNCH(ctx, ST_CONNECT, (k_connect<RT_ACCEL_BVH2, 1>), grid_for(cap), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1);
    else
        LAUNCH(ctx, ST_CONNECT, (k_connect<RT_ACCEL_BVH2, 0>), grid_for(cap), stack_bytes(ctx), ctx->sc, ctx->q, b0, b1);
    ev_begin(ctx, ST_ACCUM);
    for (int b = b0; b <= b1; b++)
        hipLaunchKernelGGL(k_accumulate, dim3(std::min(grid_for(ctx->nPix).x, 2048u)), dim3(kBlock), 0, ctx->stream, ctx->q, b);
    ev_end(ctx, ST_ACCUM);
    HIPCHK(hipGetLastError());
    return RT_OK;
}

// Renderer::RayTrace() (renderer.cpp:64-94), `frames` times.
extern "C" int rt_render(RtCtx* ctx, const RtCamera* cam, const RtSettings* settings, int32_t frames)
{
    int rc = need_scene(ctx, "rt_render"); if (rc) return rc;
    if (!cam) return fail(RT_E_INVALID, "rt_render: null camera");
    if (frames <= 0) return fail(RT_E_INVALID, "rt_render: frames must be > 0");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const bool nee = ctx->cfg.shading == RT_SHADING_NEE, rr = ctx->cfg.russian_roulette != 0;
    const int renderBVH = settings ? settings->renderBVH : 0;
    for (int f = 0; f < frames; f++) {
        if ((rc = generate(ctx, cam, settings, 1))) return rc;   // the frame's counter reset rides in k_generate's first workgroup
        for (int b = 0; b < ctx->cfg.max_bounces; b++) {
            if ((rc = rt_stage_extend(ctx, b, renderBVH))) return rc;
            if (renderBVH) break;                                  // renderer.cpp:79
            if ((rc = rt_stage_shade(ctx, b))) return rc;
            if (!rr && nee) if ((rc = rt_stage_connect(ctx, b, b))) return rc;   // renderer.cpp:85-87
        }
        if (rr && nee && !renderBVH) if ((rc = rt_stage_connect(ctx, 0, ctx->cfg.max_bounces - 1))) return rc; // renderer.cpp:91-92
        ctx->frames++;
    }
    return RT_OK;
}
static int check_fault(RtCtx* ctx) // after a stream sync: did a bounded device-side wait expire?
{
    int32_t f = 0;
    HIPCHK(hipMemcpy(&f, ctx->q.fault, sizeof f, hipMemcpyDeviceToHost));
    if (f) {
        (void)hipMemset(ctx->q.fault, 0, sizeof f);
        return fail(RT_E_DEVICE, "device fault 0x%08x: a bounded wait of the ordered scan in k_shade expired (results of this render are invalid)", (unsigned)f);
    }
    return RT_OK;
}
extern "C" int rt_synchronize(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_synchronize: null context");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    return check_fault(ctx);
}

extern "C" int rt_focus(RtCtx* ctx, int32_t x, int32_t y, const RtCamera* cam, float* t)
{
    int rc = need_scene(ctx, "rt_focus"); if (rc) return rc;
    if (!cam || !t) return fail(RT_E_INVALID, "rt_focus: null argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    if (ctx->cfg.accel == RT_ACCEL_BVH4)
        hipLaunchKernelGGL((k_focus<RT_ACCEL_BVH4>), dim3(1), dim3(kBlock), stack_bytes(ctx), ctx->stream, ctx->sc, *cam, x, y, ctx->cfg.width, ctx->cfg.height, ctx->dFocus);
    else
        hipLaunchKernelGGL((k_focus<RT_ACCEL_BVH2>), dim3(1), dim3(kBlock), stack_bytes(ctx), ctx->stream, ctx->sc, *cam, x, y, ctx->cfg.width, ctx->cfg.height, ctx->dFocus);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(t, ctx->dFocus, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

extern "C" int rt_read_accum(RtCtx* ctx, RtFloat4* out)
{
    if (!ctx || !out) return fail(RT_E_INVALID, "rt_read_accum: null argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    int rc = check_fault(ctx); if (rc) return rc;
    HIPCHK(hipMemcpy(out, ctx->q.accum, sizeof(float4) * (size_t)ctx->cfg.width * ctx->cfg.height, hipMemcpyDeviceToHost));
    return RT_OK;
}
// Restore half of a checkpoint: the running-sum accumulator (SURVEY.md §5 "Checkpoint / resume": the state a render
// carries across frames is {accum, seeds, frames}; seeds go through rt_set_seeds, frames live in the caller's Settings).
extern "C" int rt_write_accum(RtCtx* ctx, const RtFloat4* in)
{
    if (!ctx || !in) return fail(RT_E_INVALID, "rt_write_accum: null argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(ctx->q.accum, in, sizeof(float4) * (size_t)ctx->cfg.width * ctx->cfg.height, hipMemcpyHostToDevice));
    return RT_OK;
}
static int sum_table(RtCtx* ctx, const unsigned long long* dev, uint64_t out[kCtrCols])
{
    std::vector<unsigned long long> h((size_t)ctx->gridMax * kCtrCols);
    HIPCHK(hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < kCtrCols; k++) out[k] = 0;
    for (size_t b = 0; b < (size_t)ctx->gridMax; b++) for (int k = 0; k < kCtrCols; k++) out[k] += h[b * kCtrCols + k];
    return RT_OK;
}
extern "C" int rt_read_counters(RtCtx* ctx, RtCounters* out)
{
    if (!ctx || !out) return fail(RT_E_INVALID, "rt_read_counters: null argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    uint64_t e[kCtrCols], c[kCtrCols];
    int rc = sum_table(ctx, ctx->q.ctrExtend, e); if (rc) return rc;
    rc = sum_table(ctx, ctx->q.ctrConnect, c); if (rc) return rc;
    memset(out, 0, sizeof *out);
    out->extend_rays = e[0]; out->extend_tlas_visits = e[1]; out->extend_inst_visits = e[2]; out->extend_node_visits = e[3]; out->extend_prim_tests = e[4];
    out->connect_rays = c[0]; out->connect_tlas_visits = c[1]; out->connect_inst_visits = c[2]; out->connect_node_visits = c[3]; out->connect_prim_tests = c[4];
    out->primary_rays = ctx->primaryRays; out->shadow_rays = c[0]; out->frames = ctx->frames;
    out->extend_node_issues = e[5]; out->extend_leaf_issues = e[6]; out->connect_node_issues = c[5]; out->connect_leaf_issues = c[6];
    out->extend_loop_node_events = e[7]; out->extend_loop_leaf_events = e[8]; out->connect_loop_node_events = c[7]; out->connect_loop_leaf_events = c[8];
    return RT_OK;
}
extern "C" int rt_reset_counters(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_reset_counters: null context");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemset(ctx->q.ctrExtend, 0, sizeof(unsigned long long) * (size_t)ctx->gridMax * kCtrCols));
    HIPCHK(hipMemset(ctx->q.ctrConnect, 0, sizeof(unsigned long long) * (size_t)ctx->gridMax * kCtrCols));
    ctx->frames = 0; ctx->primaryRays = 0;
    return RT_OK;
}
extern "C" int rt_read_stage_times(RtCtx* ctx, RtStageTimes* out)
{
    if (!ctx || !out) return fail(RT_E_INVALID, "rt_read_stage_times: null argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    *out = ctx->times;
    return RT_OK;
}
extern "C" int rt_set_profile(RtCtx* ctx, int32_t level)
{
    if (!ctx || level < 0 || level > 2) return fail(RT_E_INVALID, "rt_set_profile: level must be 0, 1 or 2");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    ctx->cfg.profile = level;
    ev_init(ctx);
    return RT_OK;
}
extern "C" int rt_reset_stage_times(RtCtx* ctx)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_reset_stage_times: null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    memset(&ctx->times, 0, sizeof ctx->times);
    return RT_OK;
}

// ---- debug import/export ----------------------------------------------------------------------
static int ray_io(RtCtx* ctx)
{
    if (!ctx->dRayIO) HIPCHK(hipMalloc((void**)&ctx->dRayIO, sizeof(RtRay) * (size_t)ctx->nPix));
    return RT_OK;
}
static int read_count(RtCtx* ctx, const int32_t* dev, int32_t* out)
{
    HIPCHK(hipMemcpyAsync(out, dev, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}
extern "C" int rt_debug_get_rays(RtCtx* ctx, int32_t bounce, RtRay* out, int32_t capacity, int32_t* n)
{
    int rc = need_scene(ctx, "rt_debug_get_rays"); if (rc) return rc;
    if (!n || bounce < 0 || bounce > ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_debug_get_rays: bad argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    if ((rc = read_count(ctx, ctx->q.nRays + bounce, n))) return rc;
    if (!out) return RT_OK;
    if (*n > capacity) return fail(RT_E_INVALID, "rt_debug_get_rays: capacity %d < %d rays", capacity, *n);
    if ((rc = ray_io(ctx))) return rc;
    if (*n > 0) {
        hipLaunchKernelGGL(k_export_rays, grid_for(*n), dim3(kBlock), 0, ctx->stream, ctx->sc, ctx->q, bounce, ctx->dRayIO);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, ctx->dRayIO, sizeof(RtRay) * (size_t)*n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}
extern "C" int rt_debug_set_rays(RtCtx* ctx, int32_t bounce, const RtRay* in, int32_t n)
{
    if (!ctx || !in || n < 0 || n > ctx->nPix || bounce < 0 || bounce > ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_debug_set_rays: bad argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int rc = ray_io(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        HIPCHK(hipMemcpy(ctx->dRayIO, in, sizeof(RtRay) * (size_t)n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_import_rays, grid_for(n), dim3(kBlock), 0, ctx->stream, ctx->q, ctx->dRayIO, n, bounce & 1);
    }
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, ctx->stream, ctx->q.nRays + bounce, n);
    memset(ctx->shadeRun, 0, sizeof ctx->shadeRun); ctx->generated = false;   // injected queue: scan state must be re-armed
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}
extern "C" int rt_debug_get_shadow(RtCtx* ctx, int32_t b0, int32_t b1, RtShadowRecord* out, int32_t capacity, int32_t* n)
{
    if (!ctx || !n || b0 < 0 || b1 < b0 || b1 >= ctx->cfg.max_bounces) return fail(RT_E_INVALID, "rt_debug_get_shadow: bad argument");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    int32_t lo = 0, hi = 0, rc;
    if ((rc = read_count(ctx, ctx->q.nShadow + b0, &lo))) return rc;
    if ((rc = read_count(ctx, ctx->q.nShadow + b1 + 1, &hi))) return rc;
    *n = hi - lo;
    if (!out) return RT_OK;
    if (*n > capacity) return fail(RT_E_INVALID, "rt_debug_get_shadow: capacity %d < %d", capacity, *n);
    std::vector<float4> a((size_t)*n), b((size_t)*n), c((size_t)*n);
    if (*n > 0) {
        HIPCHK(hipMemcpy(a.data(), ctx->q.sA + lo, sizeof(float4) * a.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), ctx->q.sB + lo, sizeof(float4) * b.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c.data(), ctx->q.sC + lo, sizeof(float4) * c.size(), hipMemcpyDeviceToHost));
    }
    for (int32_t i = 0; i < *n; i++) {
        RtShadowRecord r;
        r.ox = a[i].x; r.oy = a[i].y; r.oz = a[i].z; r.tmax = a[i].w;
        r.lx = b[i].x; r.ly = b[i].y; r.lz = b[i].z; memcpy(&r.pixelIdx, &b[i].w, 4);
        r.radiance = RtFloat4{ c[i].x, c[i].y, c[i].z, c[i].w };
        out[i] = r;
    }
    return RT_OK;
}
extern "C" int rt_debug_enable_steps(RtCtx* ctx, int32_t on)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_debug_enable_steps: null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->q.steps = on ? ctx->dSteps : nullptr;
    return RT_OK;
}
extern "C" int rt_debug_get_steps(RtCtx* ctx, int32_t* out, int32_t capacity, int32_t* n)
{
    if (!ctx || !n) return fail(RT_E_INVALID, "rt_debug_get_steps: bad argument");
    if (!ctx->q.steps) return fail(RT_E_INVALID, "rt_debug_get_steps: call rt_debug_enable_steps(ctx, 1) before the extend stage");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *n = ctx->nPix;
    if (!out) return RT_OK;
    if (capacity < ctx->nPix) return fail(RT_E_INVALID, "rt_debug_get_steps: capacity too small");
    HIPCHK(hipMemcpy(out, ctx->dSteps, sizeof(int32_t) * (size_t)ctx->nPix, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- post-processing chain (renderer.cpp:95-124 PostProc, :303-308 SaveFrame) ----------------------------------------
extern "C" int rt_postproc(RtCtx* ctx, int32_t frames, float vignette, float gamma, float chromatic, RtFloat4* outF32, uint8_t* outRGBA8)
{
    if (!ctx) return fail(RT_E_INVALID, "rt_postproc: null context");
    if (frames <= 0) return fail(RT_E_INVALID, "rt_postproc: frames must be > 0 (it is the divisor of prep())");
    HIPCHK(hipSetDevice(ctx->cfg.device));
    const size_t px = (size_t)ctx->cfg.width * ctx->cfg.height;
    if (!ctx->dPostF) { HIPCHK(hipMalloc((void**)&ctx->dPostF, px * sizeof(float4))); HIPCHK(hipMalloc((void**)&ctx->dPostB, px * 4)); }
    PostParams pp{ 1 / (float)frames, vignette, gamma, chromatic, ctx->cfg.width, ctx->cfg.height };
    hipLaunchKernelGGL(k_postproc, grid_for((int)px), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->q.accum, ctx->dPostF, ctx->dPostB, pp);
    HIPCHK(hipGetLastError());
    if (outF32) HIPCHK(hipMemcpyAsync(outF32, ctx->dPostF, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (outRGBA8) HIPCHK(hipMemcpyAsync(outRGBA8, ctx->dPostB, px * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ev_collect(ctx);
    return RT_OK;
}

// ---- lanes: several sample streams of one accumulation behind ONE handle -------------------------------------------------------
// Every launch of a frame ends in a tail of a few long rays during which most of the chip idles, and the frames of ONE seed stream cannot
// overlap (each continues the RNG state of the one before).  A group renders the accumulation as `lanes` independent sample streams
// instead - own context, HIP stream, queues and seed slice each, ONE device copy of the scene - and interleaves their frames, so the
// tails of one lane's launches are filled by the others' kernels.  The group's accumulator is the sum of its lanes' accumulators in
// lane order.  (Reference: one Renderer, one in-order queue, renderer.cpp:26-94; the group is what stands behind Renderer::Tick here.)
//
// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included) and runs kernels of streams
// that share one after the other.  The library asks for 16 when it is loaded (before HIP initialises, unless the application already
// did; eight lanes + the process's own streams + RCCL's need more than 8: with 8 queues eight lanes run four at a time), and rt_group_create MEASURES how many of its streams really run side by side, so a caller is told instead of silently serialised.
__attribute__((constructor)) static void rt355_request_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

__global__ void k_spin(long long ticks, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) *sink = 1;
}
__global__ __launch_bounds__(kBlock) void k_sum_lanes(float4* out, const float4* a0, const float4* a1, const float4* a2, const float4* a3,
                                                       const float4* a4, const float4* a5, const float4* a6, const float4* a7, int lanes, int first, int n)
{
    // only the group's own rows [first, first + n): the bands of one frame can be summed into one buffer without touching each other
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    i += first;
    const float4* a[8] = { a0, a1, a2, a3, a4, a5, a6, a7 };
    float4 s = a[0][i];
    for (int m = 1; m < lanes; m++) s = add4(s, a[m][i]);   // lane order, left to right
    out[i] = s;
}

struct RtGroup {
    std::vector<RtCtx*> lane;
    float4* sum = nullptr;              // lane-ordered sum of the lanes' accumulators (own buffer)
    std::vector<hipEvent_t> done;       // one per lane: "this lane's queued frames are finished", for the sum on lane 0's stream
    uint64_t frames = 0;                // frames rendered by all lanes since the last reset (= the divisor of prep())
    int concurrent = 0;                 // streams measured to run side by side at creation
    int nextLane = 0;                   // round-robin position, so that successive one-frame calls visit all lanes
};
static constexpr int kMaxLanes = 8;

static void group_free(RtGroup* g)
{
    if (!g) return;
    for (RtCtx* c : g->lane) ctx_free(c);
    for (hipEvent_t e : g->done) (void)hipEventDestroy(e);
    if (g->sum) (void)hipFree(g->sum);
    delete g;
}
// How many of the group's streams execute concurrently: one single-wave kernel that naps for ~1 ms, first on one stream, then on
// every stream at once, both timed on the host.  Streams that share a hardware queue run their naps one after the other, so the
// second figure is `depth` times the first, depth = the longest chain of serialised streams; the answer is lanes / depth.
static int measure_concurrency(RtGroup* g)
{
    const int n = (int)g->lane.size();
    if (n <= 1) return n;
    int rate = 0;
    if (hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, g->lane[0]->cfg.device) != hipSuccess || rate <= 0) rate = 100000;   // kHz
    const long long ticks = (long long)rate;        // 1 ms
    int* sink = (int*)g->lane[0]->q.fault;          // never written (ticks >= 0)
    auto run = [&](int streams) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int m = 0; m < streams; m++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, g->lane[(size_t)m]->stream, ticks, sink);
        for (int m = 0; m < streams; m++) (void)hipStreamSynchronize(g->lane[(size_t)m]->stream);
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    (void)run(n);                                    // warms the code object up on every stream
    // the best of three each: a host thread that is descheduled for a millisecond must not read as a serialised stream
    float one = run(1), all = run(n);
    for (int k = 0; k < 2; k++) { one = std::min(one, run(1)); all = std::min(all, run(n)); }
    if (one <= 0 || all <= 0) return 0;
    const int depth = std::max(1, std::min(n, (int)std::lround(all / one)));
    return std::max(1, n / depth);
}

extern "C" int rt_group_create(const RtConfig* cfg, int32_t lanes, RtGroup** out)
{
    if (!cfg || !out) return fail(RT_E_INVALID, "rt_group_create: null argument");
    if (lanes < 1 || lanes > kMaxLanes) return fail(RT_E_INVALID, "rt_group_create: lanes must be 1..%d", kMaxLanes);
    std::unique_ptr<RtGroup, void (*)(RtGroup*)> guard(new RtGroup(), group_free);
    RtGroup* g = guard.get();
    for (int m = 0; m < lanes; m++) {
        RtConfig c = *cfg;
        if (lanes > 1) {   // contexts that share the GPU get the footprints that fit BESIDE each other
            if (c.shade_blocks_per_cu == 0) c.shade_blocks_per_cu = 1;
            if (c.persist_blocks_per_cu == 0) c.persist_blocks_per_cu = 2;
        }
        if (m > 0) c.profile = 0;   // HIP-event brackets on the first lane only
        RtCtx* ctx = nullptr;
        const int rc = rt_create(&c, &ctx);
        if (rc != RT_OK) return rc;
        g->lane.push_back(ctx);
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(RT_E_DEVICE, "rt_group_create: hipEventCreate failed");
        g->done.push_back(e);
    }
    const size_t px = (size_t)cfg->width * cfg->height;
    if (hipMalloc((void**)&g->sum, px * sizeof(float4)) != hipSuccess) return fail(RT_E_NOMEM, "rt_group_create: hipMalloc of the group accumulator failed");
    HIPCHK(hipMemset(g->sum, 0, px * sizeof(float4)));
    g->concurrent = measure_concurrency(g);
    if (g->concurrent < lanes) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "librt355: %d lanes requested but only %d of their HIP streams run concurrently (GPU_MAX_HW_QUEUES=%s; HIP serialises streams that "
                            "share a hardware queue - set GPU_MAX_HW_QUEUES >= lanes + 1 before the process initialises HIP, or load librt355 first)\n",
                    lanes, g->concurrent, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset");
        }
    }
    *out = guard.release();
    return RT_OK;
}
extern "C" int rt_group_destroy(RtGroup* g) { group_free(g); return RT_OK; }
extern "C" int rt_group_lanes(RtGroup* g) { return g ? (int)g->lane.size() : 0; }
extern "C" int rt_group_concurrency(RtGroup* g) { return g ? g->concurrent : 0; }
extern "C" RtCtx* rt_group_lane(RtGroup* g, int32_t m) { return g && m >= 0 && m < (int)g->lane.size() ? g->lane[(size_t)m] : nullptr; }
extern "C" uint64_t rt_group_frames(RtGroup* g) { return g ? g->frames : 0; }

extern "C" int rt_group_upload_scene(RtGroup* g, const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                                     const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                                     const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                                     const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_upload_scene: null group");
    int rc = rt_upload_scene(g->lane[0], prims, nPrims, mats, nMats, textures, nTexels, lights, nLights, bvhNodes, nNodes, primIdx, nIdx, tlas, nTlas, blas, nBlas);
    for (size_t m = 1; m < g->lane.size() && rc == RT_OK; m++) rc = rt_share_scene(g->lane[m], g->lane[0]);   // ONE device copy
    return rc;
}
// Another group on the same device (e.g. another row band of the frame) renders from the device copy `from` holds.
extern "C" int rt_group_share_scene(RtGroup* g, RtGroup* from)
{
    if (!g || !from) return fail(RT_E_INVALID, "rt_group_share_scene: null group");
    for (RtCtx* c : g->lane) { const int rc = rt_share_scene(c, from->lane[0]); if (rc != RT_OK) return rc; }
    return RT_OK;
}
// Lane m renders sample stream `firstStream + m`: its seeds are outputs (firstStream + m) * W*H + firstPixel + i + 1 of the reference's
// host xorshift32 stream (renderer.cpp:195-196).  A single Renderer has firstStream 0; rank r of a sample-partitioned job r * lanes.
extern "C" int rt_group_seed(RtGroup* g, uint64_t firstStream)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_seed: null group");
    const RtCtx* c0 = g->lane[0];
    const uint64_t P = (uint64_t)c0->cfg.width * (uint64_t)c0->cfg.height;
    uint32_t x = 0x12345678u; // template/template.cpp:711
    auto next = [&x]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    uint64_t pos = 0;
    std::vector<uint32_t> s((size_t)c0->nPix);
    for (size_t m = 0; m < g->lane.size(); m++) {
        const uint64_t first = (firstStream + m) * P + (uint64_t)c0->firstPixel;
        for (; pos < first; pos++) next();
        for (auto& v : s) v = next();
        pos += s.size();
        const int rc = rt_set_seeds(g->lane[m], s.data(), (int64_t)s.size());
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}
extern "C" int rt_group_reset(RtGroup* g)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_reset: null group");
    for (RtCtx* c : g->lane) { const int rc = rt_reset(c); if (rc != RT_OK) return rc; }
    g->frames = 0; g->nextLane = 0;
    return RT_OK;
}
// `frames` frames in all, dealt to the lanes round-robin (continuing where the last call stopped) and queued interleaved, so that the
// lanes' kernels overlap on the GPU.  Asynchronous.  After k frames in total the group accumulator holds the sum of k samples per
// pixel: prep() divides by k (postproc.cl:71), exactly as with one stream.
extern "C" int rt_group_render(RtGroup* g, const RtCamera* cam, const RtSettings* settings, int32_t frames)
{
    if (!g || !cam) return fail(RT_E_INVALID, "rt_group_render: null argument");
    if (frames <= 0) return fail(RT_E_INVALID, "rt_group_render: frames must be > 0");
    const int n = (int)g->lane.size();
    for (int f = 0; f < frames; f++) {
        const int rc = rt_render(g->lane[(size_t)g->nextLane], cam, settings, 1);
        if (rc != RT_OK) return rc;
        g->nextLane = (g->nextLane + 1) % n;
    }
    g->frames += (uint64_t)frames;
    return RT_OK;
}
extern "C" int rt_group_synchronize(RtGroup* g)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_synchronize: null group");
    for (RtCtx* c : g->lane) { const int rc = rt_synchronize(c); if (rc != RT_OK) return rc; }
    return RT_OK;
}
// The lane-ordered sum of the lanes' accumulators, on the device: into `devicePtr` (float4[width*height], e.g. the tensor a
// torch.distributed all_reduce then works on) or, when NULL, into the group's own buffer.  Queued on lane 0's stream behind every
// lane's pending frames; rt_group_synchronize (or rt_group_read_accum) waits for it.
extern "C" int rt_group_sum(RtGroup* g, void* devicePtr)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_sum: null group");
    RtCtx* c0 = g->lane[0];
    HIPCHK(hipSetDevice(c0->cfg.device));
    const int n = (int)g->lane.size();
    for (int m = 1; m < n; m++) { HIPCHK(hipEventRecord(g->done[(size_t)m], g->lane[(size_t)m]->stream)); HIPCHK(hipStreamWaitEvent(c0->stream, g->done[(size_t)m], 0)); }
    const float4* a[kMaxLanes];
    for (int m = 0; m < kMaxLanes; m++) a[m] = g->lane[(size_t)std::min(m, n - 1)]->q.accum;
    hipLaunchKernelGGL(k_sum_lanes, grid_for(c0->nPix), dim3(kBlock), 0, c0->stream, devicePtr ? (float4*)devicePtr : g->sum,
                       a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], n, c0->firstPixel, c0->nPix);
    HIPCHK(hipGetLastError());
    // the lanes must not start overwriting their accumulators before the sum has read them
    HIPCHK(hipEventRecord(g->done[0], c0->stream));
    for (int m = 1; m < n; m++) HIPCHK(hipStreamWaitEvent(g->lane[(size_t)m]->stream, g->done[0], 0));
    return RT_OK;
}
extern "C" int rt_group_read_accum(RtGroup* g, RtFloat4* out)
{
    if (!g || !out) return fail(RT_E_INVALID, "rt_group_read_accum: null argument");
    int rc = rt_group_sum(g, nullptr); if (rc != RT_OK) return rc;
    rc = rt_group_synchronize(g); if (rc != RT_OK) return rc;
    const RtCtx* c0 = g->lane[0];
    HIPCHK(hipMemcpy(out, g->sum, sizeof(float4) * (size_t)c0->cfg.width * c0->cfg.height, hipMemcpyDeviceToHost));
    return RT_OK;
}
extern "C" int rt_group_focus(RtGroup* g, int32_t x, int32_t y, const RtCamera* cam, float* t) { return g ? rt_focus(g->lane[0], x, y, cam, t) : fail(RT_E_INVALID, "rt_group_focus: null group"); }
// Renderer::PostProc + SaveFrame over the group's accumulator; `frames` is the group's frame count (rt_group_frames) unless > 0.
extern "C" int rt_group_postproc(RtGroup* g, int32_t frames, float vignette, float gamma, float chromatic, RtFloat4* outF32, uint8_t* outRGBA8)
{
    if (!g) return fail(RT_E_INVALID, "rt_group_postproc: null group");
    RtCtx* c0 = g->lane[0];
    if (g->lane.size() == 1) return rt_postproc(c0, frames > 0 ? frames : (int32_t)std::max<uint64_t>(g->frames, 1), vignette, gamma, chromatic, outF32, outRGBA8);
    int rc = rt_group_sum(g, nullptr); if (rc != RT_OK) return rc;
    float4* own = c0->q.accum;
    c0->q.accum = g->sum;                  // k_postproc reads the accumulator it is handed: the summed one
    rc = rt_postproc(c0, frames > 0 ? frames : (int32_t)std::max<uint64_t>(g->frames, 1), vignette, gamma, chromatic, outF32, outRGBA8);
    c0->q.accum = own;
    return rc;
}

#ifdef RT355_TAIL_PROBE
// lab build only: copies the per-wave probe records of the last persistent launches out (9 x 8192 x 4 uint64, see rt355_kernels.h)
extern "C" int rt_lab_tail_probe(void* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(rt355dev::g_tp), sizeof(rt355dev::g_tp)) == hipSuccess ? 0 : 1; }
#endif
