/* oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's wavefront path tracer hot path:
 * generate -> extend -> shade -> connect with TLAS / BVH2 / BVH4 traversal
 * (reference: src/cl/wavefront.cl, tlas.cl, bvh.cl, primitives.cl, ray.cl,
 * shading.cl, camera.cl, glass.cl, skydome.cl, util.cl; launch sequence
 * src/renderer.cpp:64-94).  It is the checker for the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (librt355.so) never links or calls anything in this directory.
 *
 * Pinning: see oracle/README.md — the restatement is checked stage by stage against
 * the reference's own OpenCL kernels, compiled unmodified for gfx950 by
 * oracle/build_ref.sh and run on the MI355X (outputs committed under tests/golden/).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/rt355_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SHADING_SIMPLE = 0, ORC_SHADING_NEE = 1 };       /* renderer.h:6-7   */
enum { ORC_SAMPLING_HEMISPHERE = 0, ORC_SAMPLING_COSINE = 1 }; /* renderer.h:12-13 */
enum { ORC_ACCEL_BVH2 = 0, ORC_ACCEL_BVH4 = 1 };            /* renderer.h:15-16 */
/* Schedules (legal executions of the reference's persistent-thread kernels):
 * S1: every work-item claims exactly queue slot g (RNG stream seeds[g]); appends
 *     keep the relative order of their source slots.
 * S0: ONE work-item (global size 1) drains the whole queue: slots are taken in
 *     descending order by atomic_dec, all draws come from seeds[0], appends happen
 *     in processing order.  This is what the reference kernels do when launched
 *     with a global size of 1, and is used to pin shade() against them. */
enum { ORC_SCHED_S1 = 1, ORC_SCHED_S0 = 0 };

typedef struct OrcScene {
    const RtPrimitive*   prims;     int32_t nPrims;
    const RtMaterial*    mats;      int32_t nMats;
    const RtFloat4*      tex;       int32_t nTex;
    const uint32_t*      lights;    int32_t nLights;
    const RtBVHNode2*    bvh2;      /* used when accel == ORC_ACCEL_BVH2 */
    const RtBVHNode4*    bvh4;      /* used when accel == ORC_ACCEL_BVH4 */
    int32_t              nNodes;
    const uint32_t*      primIdx;   int32_t nIdx;
    const RtTLASNode*    tlas;      int32_t nTlas;
    const RtBVHInstance* blas;      int32_t nBlas;
} OrcScene;

typedef struct OrcConfig {
    int32_t width, height;
    int32_t max_bounces;            /* host loop count, renderer.cpp:75 (MAX_BOUNCES) */
    int32_t shading, sampling, accel;
    int32_t russian_roulette, filter_fireflies;
    int32_t schedule;               /* ORC_SCHED_S1 / ORC_SCHED_S0 */
} OrcConfig;

/* Work counters behind the roofline formula of SURVEY.md §8(d). */
typedef struct OrcCounters {
    uint64_t rays;        /* R      rays traced                       */
    uint64_t tlas_visits; /* V_tlas TLAS interior visits               */
    uint64_t inst_visits; /* L_inst instance (TLAS leaf) visits        */
    uint64_t node_visits; /* V_int  BVH2 interior visits / V4 BVH4 pops */
    uint64_t prim_tests;  /* T_prim primitive tests                    */
} OrcCounters;

uint32_t orc_xorshift32(uint32_t* state);
void     orc_seed_stream(uint32_t* seeds, int64_t first, int64_t n);   /* renderer.cpp:195-196 */

void orc_generate(RtRay* rays, int32_t n, int32_t firstPixel, const OrcConfig* cfg,
                  const RtCamera* cam, int32_t antiAliasing, uint32_t* seeds);
void orc_extend(RtRay* rays, int32_t n, const OrcScene* sc, const OrcConfig* cfg,
                int32_t renderBVH, RtFloat4* accum, int32_t* steps, OrcCounters* ctr);
void orc_shade(RtRay* in, int32_t nIn, RtRay* out, int32_t* nOut,
               RtShadowRay* shadow, int32_t* nShadow, const OrcScene* sc,
               const OrcConfig* cfg, RtFloat4* accum, uint32_t* seeds);
void orc_connect(const RtShadowRay* shadow, int32_t n, const OrcScene* sc,
                 const OrcConfig* cfg, RtFloat4* accum, OrcCounters* ctr);
float orc_focus(int32_t x, int32_t y, const OrcScene* sc, const OrcConfig* cfg, const RtCamera* cam);

/* One Renderer::RayTrace() (renderer.cpp:64-94) over pixels [firstPixel, firstPixel+n).
 * accum is indexed by global pixel index; seeds by band-local slot.  work must hold
 * 2*n RtRay + (max_bounces)*n RtShadowRay (query with orc_frame_work_bytes). */
size_t orc_frame_work_bytes(int32_t n, const OrcConfig* cfg);
void   orc_render_frame(const OrcScene* sc, const OrcConfig* cfg, const RtCamera* cam,
                        int32_t antiAliasing, int32_t firstPixel, int32_t n,
                        RtFloat4* accum, uint32_t* seeds, void* work,
                        OrcCounters* extendCtr, OrcCounters* connectCtr);

/* CPU baseline driver: renders `frames` frames of rows [y0,y1) split into `threads`
 * independent row bands (OpenMP), each band with its own queues and seed slice. */
void orc_render_bands(const OrcScene* sc, const OrcConfig* cfg, const RtCamera* cam,
                      int32_t antiAliasing, int32_t y0, int32_t y1, int32_t frames,
                      int32_t threads, RtFloat4* accum, uint32_t* seeds,
                      OrcCounters* extendCtr, OrcCounters* connectCtr);

/* Template-style CPU tracer (CPU-B of BASELINE.md §3): one primary ray per pixel,
 * nearest hit through the BVH, normal visualisation into out[W*H]. */
void orc_trace_normals(const OrcScene* sc, const OrcConfig* cfg, const RtCamera* cam,
                       int32_t threads, RtFloat4* out);

/* Renderer::PostProc + SaveFrame (renderer.cpp:95-124,303-308; src/cl/postproc.cl:7-86). */
void orc_postproc(const RtFloat4* accum, int32_t W, int32_t H, int32_t frames, float vignette, float gamma, float chromatic,
                  RtFloat4* outF, uint8_t* outRGBA8);

/* Unit hooks for the known-answer tests. */
void     orc_test_random_float3(uint32_t* seed, float out[4]);
void     orc_test_cosine_hemisphere(const float N[4], uint32_t* seed, float out[4]);
void     orc_test_triangle(const float v0[3], const float v1[3], const float v2[3], const float O[3], const float D[3], float out[4]);
uint32_t orc_test_wang_hash(uint32_t s);

#ifdef __cplusplus
}
#endif
#endif
