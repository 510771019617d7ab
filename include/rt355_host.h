/* rt355_host.h — C-ABI of the host-side producers (librt355_host.so): the Scene primitive /
 * material factory, the BVH2 / SBVH / BVH4 / TLAS builders, the camera maths and a Renderer
 * mirror.  These wrap the C++ classes of magr_ray_tracer_amd/host/rt_host.h, which mirror the
 * reference's Scene (src/scene.h:5-34), BVH2/BVH4 (src/bvh.h:4-56), TLAS (src/tlas.h:2-12),
 * CameraManager (src/camera.h:7-122) and Renderer (src/renderer.h:44-120).
 * All functions return 0 on success, negative on failure (rth_last_error() has the text).
 */
#ifndef RT355_HOST_H
#define RT355_HOST_H
#include "rt355_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthScene RthScene;
typedef struct RthRenderer RthRenderer;

const char* rth_last_error(void);

/* Scene::Scene / ~Scene (scene.cpp:8-10,74) */
RthScene* rth_scene_create(void);
void      rth_scene_destroy(RthScene* s);
/* Scene::AddMaterial (scene.cpp:84-100): pushes a zeroed material (texIdx -1), then copies the
 * fields of *init if given; returns the material index. */
int rth_add_material(RthScene* s, const char* name, const RtMaterial* init);
/* Scene::LoadTexture minus the file read (scene.cpp:244-256): appends texels, adds a material. */
int rth_add_texture(RthScene* s, const char* name, const RtFloat4* texels, int width, int height);
/* Scene::LoadTexture (scene.cpp:244-256) for PNG, JPEG, TGA and Radiance HDR files, texel rule of stbi_loadf; returns the material index or -1. */
int rth_load_texture(RthScene* s, const char* filename, const char* name);
int rth_add_sphere(RthScene* s, const float pos[3], float radius, const char* material);       /* scene.cpp:125-138 */
int rth_add_plane(RthScene* s, const float N[3], float d, const char* material);               /* scene.cpp:140-150 */
int rth_add_triangle(RthScene* s, const float v0[3], const float v1[3], const float v2[3],
                     const float uv0[2], const float uv1[2], const float uv2[2], const char* material, int flipNormal); /* scene.cpp:158-176 */
int rth_add_quad(RthScene* s, const float v0[3], const float v1[3], const float v2[3], const float v3[3],
                 const char* material, int flipNormal);                                       /* scene.cpp:152-156, default uvs */
/* n x AddTriangle; verts is n*9 floats (v0,v1,v2), uvs n*6 floats or NULL (all zero). */
int rth_add_triangles(RthScene* s, const float* verts, const float* uvs, int n, const char* material, int flipNormal);
/* Scene::LoadModel (scene.cpp:178-243): OBJ (+MTL map_Kd names); returns the number of triangles added or -1. */
int rth_load_model(RthScene* s, const char* filename, const char* defaultMaterial, const float pos[3], int forceDefaultMat);
/* SaveImageF (template/template.cpp:1629-1644): float4 image -> 8-bit RGB PNG, bytes (uchar)(min(c,1)*255). */
int rth_save_png(const char* file, int width, int height, const RtFloat4* data);
/* BVH2::BuildBLAS(true, startIdx) with bvh2->alpha = alpha (bvh.cpp:46-82). */
int rth_build_blas(RthScene* s, int startIdx, float alpha);
/* Threads for the following BuildBLAS calls: 1 = the reference's sequential loop; > 1 = task-parallel subtrees numbered
 * afterwards in the reference's LIFO order (identical arrays). */
int rth_set_build_threads(RthScene* s, int threads);
int rth_build_bvh4(RthScene* s);            /* new BVH4(*bvh2) (scene.cpp:71)                    */
int rth_build_tlas(RthScene* s);            /* new TLAS(*bvh2); Build() (renderer.cpp:12-13)     */
/* BVH4::Convert + Collapse (bvh.cpp:695-787) on a caller-provided BVH2 node array, one BLAS rooted at node 0; out[n] */
int rth_bvh4_from_nodes(const RtBVHNode2* nodes, int n, RtBVHNode4* out);
int rth_set_instance_transform(RthScene* s, int blas, const float invT[16]); /* scene.cpp:82 (commented out there) */

/* Borrowed views of the arrays (valid until the scene changes). */
const RtPrimitive*   rth_primitives(RthScene* s, int* n);
const RtMaterial*    rth_materials(RthScene* s, int* n);
const RtFloat4*      rth_textures(RthScene* s, int* n);
const uint32_t*      rth_lights(RthScene* s, int* n);
const RtBVHNode2*    rth_bvh2_nodes(RthScene* s, int* n);
const RtBVHNode4*    rth_bvh4_nodes(RthScene* s, int* n);
const uint32_t*      rth_prim_idx(RthScene* s, int* n);
const RtTLASNode*    rth_tlas_nodes(RthScene* s, int* n);
const RtBVHInstance* rth_blas_nodes(RthScene* s, int* n);
/* BVH statistics (bvh.h:21-22): depth, node count, spatial splits, clipped prims, prim count, SAH cost, build ms. */
int rth_bvh_stats(RthScene* s, uint32_t out_u[5], float out_f[2]);

/* CameraManager(vfov,type) + origin/forward/aperture/focalLength + UpdateCamVec() (camera.h:24-34,101-121). */
int rth_camera(int width, int height, float vfov, int type, const float origin[3], const float forward[3],
               float aperture, float focalLength, RtCamera* out);

/* seeds[i] = (first+i+1)-th xorshift32 output from 0x12345678: the host seed loop of renderer.cpp:195-196
 * (RandomUInt, template/template.cpp:711,724-730), with a start offset for row bands / sample partitions. */
int rth_seed_stream(uint32_t* out, int64_t first, int64_t n);

/* Renderer mirror (renderer.cpp:6-63): owns a context of librt355.so. */
RthRenderer* rth_renderer_create(RthScene* scene /* adopted */, int width, int height, int device, int y0, int y1,
                                 int shading, int sampling, int bvh, int russianRoulette, int filterFireflies);
void rth_renderer_destroy(RthRenderer* r);
int  rth_renderer_init(RthRenderer* r);                                   /* Renderer::Init  */
int  rth_renderer_set_camera(RthRenderer* r, const float origin[3], const float forward[3], float fov, float aperture);
int  rth_renderer_tick(RthRenderer* r, int frames);                       /* Renderer::Tick x frames */
int  rth_renderer_read(RthRenderer* r, RtFloat4* out, float* energy);     /* accumBuffer read-back + ComputeEnergy */
int  rth_renderer_camera(RthRenderer* r, RtCamera* out);
int  rth_renderer_save_frame(RthRenderer* r, const char* file);
/* CameraManager::Move / MouseMove / Zoom (camera.h:47-99); the next Tick() sees camera.moved and resets (renderer.cpp:41-46). */
int  rth_renderer_camera_move(RthRenderer* r, int camdir /* 0 Forward 1 Backwards 2 Left 3 Right 4 Up 5 Down */);
int  rth_renderer_camera_mouse(RthRenderer* r, float xOffset, float yOffset);
int  rth_renderer_camera_zoom(RthRenderer* r, float offset);
int rth_renderer_set_lanes(RthRenderer* r, int lanes);   /* before rth_renderer_init: Renderer::lanes (Tick = `lanes` overlapping frames) */
int  rth_renderer_frames(RthRenderer* r);                                 /* settings->frames */          /* Renderer::SaveFrame (renderer.cpp:303-308) */

#ifdef __cplusplus
}
#endif
#endif
