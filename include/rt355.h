/* rt355.h — C-ABI of the MI355X device path (librt355.so).
 *
 * The reference has no FFI: its "operator API" is the Kernel/Buffer call sequence that
 * Renderer issues (reference: src/renderer.cpp:64-94 RayTrace, :142-209 InitBuffers,
 * :211-263 InitWavefrontKernels, :289-301 FocusCamera, :126-140 ComputeEnergy) plus the
 * POD arrays of src/common.h.  Each entry point below replaces the cited piece of that
 * sequence; INTEGRATION.md shows the Renderer-side binding.
 *
 * Conventions: every function returns 0 on success and a negative RT_E_* code on
 * failure; rt_last_error() returns a thread-local message (the reference aborts through
 * FatalError(), template/template.cpp:949-962 — this library never aborts).  Host
 * pointers are copied during the call and never retained (the reference's Buffer keeps
 * a non-owning alias and copies in CopyToDevice(), template.cpp:1133-1137).  A context
 * is bound to one GPU and is not re-entrant: one host thread (or process) per GPU.
 * There is NO CPU fallback: without a HIP device rt_create() fails.
 */
#ifndef RT355_H
#define RT355_H
#include "rt355_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK            0
#define RT_E_INVALID    (-1)   /* bad argument / call order                       */
#define RT_E_DEVICE     (-2)   /* HIP error (message in rt_last_error)            */
#define RT_E_NOMEM      (-3)
#define RT_E_UNSUPPORTED (-4)

/* Kernel variants — the reference's prepended #defines (renderer.h:6-19, renderer.cpp:213-215). */
#define RT_SHADING_SIMPLE       0   /* SHADING_SIMPLE ("Kajiya")  */
#define RT_SHADING_NEE          1   /* SHADING_NEE (default)      */
#define RT_SAMPLING_HEMISPHERE  0   /* SAMPLING_HEMISPHERE        */
#define RT_SAMPLING_COSINE      1   /* SAMPLING_COSINE (default)  */
#define RT_ACCEL_BVH2           0   /* USE_BVH2 (default)         */
#define RT_ACCEL_BVH4           1   /* USE_BVH4                   */

typedef struct RtCtx RtCtx;

/* Replaces the compile-time macros of src/constants.h:3-7,28-31 and the ImGuiData
 * variant selection (renderer.h:23-36) with run-time values. */
typedef struct RtConfig {
    int32_t width, height;      /* SCRWIDTH, SCRHEIGHT                                        */
    int32_t y0, y1;             /* row band [y0,y1) this context renders; 0,height = all rows  */
    int32_t max_bounces;        /* MAX_BOUNCES host loop count (7); must be 1..RT_MAX_BOUNCES  */
    int32_t shading, sampling, accel;
    int32_t russian_roulette;   /* RUSSIAN_ROULETTE                                           */
    int32_t filter_fireflies;   /* FILTER_FIREFLIES                                           */
    int32_t device;             /* HIP device ordinal                                         */
    int32_t extend_variant;     /* traversal kernels: 0 = best available (derived node/triangle layout + persistent
                                 * wavefronts when the TLAS has one BLAS), 1 = traverse the reference arrays as uploaded,
                                 * one ray per lane, 2 = derived layout, one ray per lane, 4 = as 0 but multi-BLAS scenes keep the
                                 * one-ray-per-lane nested TLAS loops instead of k_trace_persist_tlas (A/B runs)      */
    int32_t profile;            /* HIP-event brackets on the context's stream: 0 none, 1 extend launches only
                                 * (what the roofline needs; ~1 % overhead), 2 every stage launch (~3.5 %)        */
    int32_t shade_blocks_per_cu;/* k_shade workgroups per CU: 0 = what the CUs hold (2, best for one context with the GPU to itself);
                                 * > 0 also selects 256-slot tiles (39 KB of LDS instead of 78): 1 leaves room for the kernels of
                                 * other contexts (best when several sample streams share the GPU)                               */
    int32_t persist_blocks_per_cu; /* workgroups per CU of the persistent traversal grids: 0 = what the hardware admits (7 extend / 6 connect);
                                 * 4 is best when three contexts share the GPU (their workgroups then fit beside each other)      */
    int32_t reserved[1];
} RtConfig;

/* Device-side work counters (per-kernel-family totals since the last rt_reset_counters).
 * They define the algorithmic bytes of SURVEY.md §8(d). */
typedef struct RtCounters {
    uint64_t extend_rays, extend_tlas_visits, extend_inst_visits, extend_node_visits, extend_prim_tests;
    uint64_t connect_rays, connect_tlas_visits, connect_inst_visits, connect_node_visits, connect_prim_tests;
    uint64_t primary_rays;       /* pixels generated                     */
    uint64_t shadow_rays;        /* shadow rays appended by shade        */
    uint64_t frames;
    /* how often the persistent event loops issued their two code paths (wave level) and how many events those issues carried:
     * loop_node_events / (64 * node_issues) is the share of the lanes that had a box pair to test when the node path ran,
     * loop_leaf_events / (64 * leaf_issues) the same for the triangle path.  Counted by the extend instantiation that also keeps the
     * per-ray `steps` (rt_debug_enable_steps(ctx, 1) or renderBVH): the production kernel does not pay for them; launches that ran one
     * ray per lane count nothing here */
    uint64_t extend_node_issues, extend_leaf_issues, connect_node_issues, connect_leaf_issues;
    uint64_t extend_loop_node_events, extend_loop_leaf_events, connect_loop_node_events, connect_loop_leaf_events;   /* the events those issues carried */
} RtCounters;

/* Accumulated stage times in milliseconds (HIP events on the context's stream) and
 * launch counts; only filled when RtConfig.profile != 0. */
typedef struct RtStageTimes {
    double  generate_ms, extend_ms, shade_ms, compact_ms, connect_ms, accumulate_ms;
    int64_t generate_launches, extend_launches, shade_launches, compact_launches, connect_launches, accumulate_launches;
} RtStageTimes;

/* Which kernels a context runs for the uploaded scene (chosen at rt_upload_scene; bench.py names the roofline's kernel from it). */
typedef struct RtKernelInfo {
    int32_t layout;               /* 0 = the reference arrays as uploaded, 1 = derived pair / quad / triangle records */
    int32_t persist, persist4;    /* persistent-wavefront traversal over the BVH2 (1: one BLAS, 2: through a multi-BLAS
                                   * TLAS, k_trace_persist_tlas, 3: the same with the deep end of the traversal stacks
                                   * spilled to global memory) / over the BVH4 (one BLAS)                             */
    int32_t stack_entries;        /* LDS traversal stack entries per lane                                             */
    int32_t persist_grid, persist_grid_connect, shade_grid;   /* workgroups of the persistent launches                */
    int32_t n_blas;
} RtKernelInfo;

const char* rt_last_error(void);
int rt_device_count(void);
int rt_kernel_info(RtCtx* ctx, RtKernelInfo* out);

/* new Buffer(...) x11 + new Kernel(...) x6 (renderer.cpp:145-157, :218-223). */
int rt_create(const RtConfig* cfg, RtCtx** out);
int rt_destroy(RtCtx* ctx);

/* primBuffer/matBuffer/texBuffer/lightBuffer/bvhNodeBuffer/bvhIdxBuffer/tlasNodeBuffer/
 * blasNodeBuffer ->CopyToDevice() (renderer.cpp:160-208).  bvhNodes is RtBVHNode2[nNodes]
 * for RT_ACCEL_BVH2 and RtBVHNode4[nNodes] for RT_ACCEL_BVH4, uploaded unchanged. */
int rt_upload_scene(RtCtx* ctx,
                    const RtPrimitive* prims, int32_t nPrims,
                    const RtMaterial* mats, int32_t nMats,
                    const RtFloat4* textures, int32_t nTexels,
                    const uint32_t* lights, int32_t nLights,
                    const void* bvhNodes, int32_t nNodes,
                    const uint32_t* primIdx, int32_t nIdx,
                    const RtTLASNode* tlas, int32_t nTlas,
                    const RtBVHInstance* blas, int32_t nBlas);

/* The host-side shape checks rt_upload_scene runs before it touches the device (index ranges, tree cycles and depths, stack needs,
 * the 32768-node / instance bound of the TLAS encodings), callable without a GPU.  `accel` says how bvhNodes is to be read. */
int rt_validate_scene(int32_t accel,
                      const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                      const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                      const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                      const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas);

/* A second context on the same device renders the scene `from` holds: it takes `from`'s device copy (uploaded arrays + derived
 * layouts) instead of uploading its own - one copy in HBM and in the caches for the sample streams of a GPU or the row bands of a
 * frame.  The contexts must agree in accel and extend_variant; the copy lives until the last context holding it is destroyed or
 * uploads another scene.  (The reference has one Renderer and one set of buffers, renderer.cpp:160-208; several contexts per device
 * are this library's way to keep a GPU full.) */
int rt_share_scene(RtCtx* ctx, RtCtx* from);

/* ---- lanes: one accumulation as several interleaved sample streams behind one handle -------------------------------------------
 * What stands behind Renderer::Tick() (renderer.cpp:26-63) when a GPU is to be kept full: `lanes` contexts (own HIP stream, queues,
 * accumulator, seed slice) that share ONE device copy of the scene; their frames are queued interleaved so that the tails of one
 * lane's launches are filled by the others' kernels (1 lane: 733, 4 lanes: 1,000 M samples/s on the bench scene).  Lane m renders
 * sample stream firstStream + m (seeds = that slice of the reference's host xorshift32 stream, renderer.cpp:195-196); the group's
 * accumulator is the sum of the lanes' accumulators in lane order and, after k frames in all, holds k samples per pixel - prep()
 * divides by k exactly as with one stream (postproc.cl:71).  A group of ONE lane is the reference's single Renderer bit for bit.
 * HIP runs kernels of streams that share a hardware queue one after the other: the library asks for GPU_MAX_HW_QUEUES=16 when it is
 * loaded (effective if HIP has not been initialised yet), rt_group_create measures how many of the group's streams really run side
 * by side (rt_group_concurrency) and writes one line to stderr when that is fewer than `lanes`. */
typedef struct RtGroup RtGroup;
int rt_group_create(const RtConfig* cfg, int32_t lanes, RtGroup** out);      /* lanes 1..8; cfg as for rt_create (row band included)   */
int rt_group_destroy(RtGroup* g);
int rt_group_lanes(RtGroup* g);
int rt_group_concurrency(RtGroup* g);                                        /* streams measured to run concurrently at creation        */
RtCtx* rt_group_lane(RtGroup* g, int32_t m);                                 /* lane m's context (counters, stage times, debug stages)  */
uint64_t rt_group_frames(RtGroup* g);                                        /* frames rendered by all lanes since the last reset       */
int rt_group_upload_scene(RtGroup* g,
                          const RtPrimitive* prims, int32_t nPrims, const RtMaterial* mats, int32_t nMats,
                          const RtFloat4* textures, int32_t nTexels, const uint32_t* lights, int32_t nLights,
                          const void* bvhNodes, int32_t nNodes, const uint32_t* primIdx, int32_t nIdx,
                          const RtTLASNode* tlas, int32_t nTlas, const RtBVHInstance* blas, int32_t nBlas);
int rt_group_share_scene(RtGroup* g, RtGroup* from);                         /* e.g. the row bands of one frame: one device copy        */
int rt_group_seed(RtGroup* g, uint64_t firstStream);                         /* a single Renderer: 0; rank r of a sample split: r*lanes */
int rt_group_reset(RtGroup* g);                                              /* resetKernel on every lane; frames = 0                   */
int rt_group_render(RtGroup* g, const RtCamera* cam, const RtSettings* settings, int32_t frames);   /* `frames` in all, round-robin   */
int rt_group_synchronize(RtGroup* g);
int rt_group_sum(RtGroup* g, void* devicePtr);                               /* lane-ordered sum of the group's rows into devicePtr (a
                                                                              * full-frame float4 buffer; NULL: the group's own)       */
int rt_group_read_accum(RtGroup* g, RtFloat4* out);
int rt_group_focus(RtGroup* g, int32_t x, int32_t y, const RtCamera* cam, float* t);
int rt_group_postproc(RtGroup* g, int32_t frames, float vignette, float gamma, float chromatic, RtFloat4* outF32, uint8_t* outRGBA8);

/* seedBuffer (renderer.cpp:195-196,200).  rt_set_seeds takes the band's slice
 * (one uint per band pixel); rt_seed_default fills seeds[i] with the (firstPixel+i+1)-th
 * xorshift32 output from 0x12345678 like the reference's host loop. */
int rt_set_seeds(RtCtx* ctx, const uint32_t* seeds, int64_t n);
int rt_seed_default(RtCtx* ctx);
int rt_get_seeds(RtCtx* ctx, uint32_t* out, int64_t n);

/* Optional: render into a caller-owned device accumulator float4[width*height]
 * (e.g. a torch tensor, so that torch.distributed can reduce it). NULL = own buffer. */
int rt_bind_accum(RtCtx* ctx, void* devicePtr);
void* rt_accum_device_ptr(RtCtx* ctx);
void* rt_stream(RtCtx* ctx);

/* resetKernel->Run(PIXELS) (renderer.cpp:41-46): clears this context's band. */
int rt_reset(RtCtx* ctx);

/* Renderer::RayTrace() x frames (renderer.cpp:64-94): generate, then max_bounces x
 * (extend, shade[, connect when RR is off]), then connect.  Uses settings->antiAliasing
 * and settings->renderBVH; asynchronous on the context's stream. */
int rt_render(RtCtx* ctx, const RtCamera* cam, const RtSettings* settings, int32_t frames);
int rt_synchronize(RtCtx* ctx);

/* focusKernel->Run(1) + settingsBuffer->CopyFromDevice() (renderer.cpp:289-301). */
int rt_focus(RtCtx* ctx, int32_t x, int32_t y, const RtCamera* cam, float* t);

/* accumBuffer->CopyFromDevice() (renderer.cpp:128): full frame float4[width*height]. */
int rt_read_accum(RtCtx* ctx, RtFloat4* out);
/* Checkpoint restore: overwrite the accumulator (the state carried across frames is {accum, seeds, settings->frames};
 * the reference has no checkpointing — SURVEY.md §5). */
int rt_write_accum(RtCtx* ctx, const RtFloat4* in);
int rt_read_counters(RtCtx* ctx, RtCounters* out);
int rt_reset_counters(RtCtx* ctx);
int rt_set_profile(RtCtx* ctx, int32_t level);      /* change RtConfig.profile at run time (synchronises) */
int rt_read_stage_times(RtCtx* ctx, RtStageTimes* out);
int rt_reset_stage_times(RtCtx* ctx);

/* Renderer::PostProc() + SaveFrame() (renderer.cpp:95-124,303-308; src/cl/postproc.cl): prep (divide by `frames`, clamp 1),
 * vignetting if vignette > 0, gammaCorr if gamma != 1, chromatic if chromatic > 0, then min(color,1).  Writes the float
 * image (float4[width*height], w = 1) and/or the 8-bit image (RGBA, bytes (uchar)(c*255) as SaveImageF); either may be NULL. */
int rt_postproc(RtCtx* ctx, int32_t frames, float vignette, float gamma, float chromatic, RtFloat4* outF32, uint8_t* outRGBA8);

/* ---- stage-level entry points (one kernel of the reference each), used by the parity
 * tests to feed identical inputs to one stage at a time ------------------------------ */
int rt_stage_begin_frame(RtCtx* ctx);                                           /* renderer.cpp:66-69   */
int rt_stage_generate(RtCtx* ctx, const RtCamera* cam, const RtSettings* s);    /* wavefront.cl:14-34   */
int rt_stage_extend(RtCtx* ctx, int32_t bounce, int32_t renderBVH);             /* wavefront.cl:35-75   */
int rt_stage_shade(RtCtx* ctx, int32_t bounce);                                 /* wavefront.cl:76-142  */
int rt_stage_connect(RtCtx* ctx, int32_t firstBounce, int32_t lastBounce);      /* wavefront.cl:144-201 */
/* Ray queue of `bounce` as reference-layout Ray structs (I and N as extend leaves them). */
int rt_debug_get_rays(RtCtx* ctx, int32_t bounce, RtRay* out, int32_t capacity, int32_t* n);
int rt_debug_set_rays(RtCtx* ctx, int32_t bounce, const RtRay* in, int32_t n);
/* Shadow rays appended by shade() of bounces [firstBounce,lastBounce], in queue order:
 * origin = I + L*EPSILON (xyz), tmax = dist - 2*EPSILON, dir = L (xyz), pixel index and the
 * radiance the ray carries if unoccluded. */
typedef struct RtShadowRecord { float ox, oy, oz, tmax; float lx, ly, lz; int32_t pixelIdx; RtFloat4 radiance; } RtShadowRecord;
int rt_debug_get_shadow(RtCtx* ctx, int32_t firstBounce, int32_t lastBounce, RtShadowRecord* out, int32_t capacity, int32_t* n);
/* Per-ray `steps` (the value the reference's heat map shows, wavefront.cl:66-67) of the last extend; recording is off
 * by default (it costs 4 B written per ray) and is switched on with rt_debug_enable_steps(ctx, 1). */
int rt_debug_enable_steps(RtCtx* ctx, int32_t on);
int rt_debug_get_steps(RtCtx* ctx, int32_t* out, int32_t capacity, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
