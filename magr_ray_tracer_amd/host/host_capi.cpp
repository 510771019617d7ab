// host_capi.cpp — extern "C" wrappers (include/rt355_host.h) over the C++ host classes.
#include <cstring>
#include <exception>
#include <string>
#include "../../include/rt355.h"
#include "../../include/rt355_host.h"
#include "rt_host.h"

using namespace rt355;

static thread_local std::string g_herr;
struct RthScene { Scene scene; TLAS* tlas = nullptr; ~RthScene() { delete tlas; } };
struct RthRenderer { Renderer* r = nullptr; RthScene* adopted = nullptr; };

#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { g_herr = e.what(); return -1; }
static float3 f3(const float* p) { return float3(p[0], p[1], p[2]); }
static float2 f2(const float* p) { float2 r; if (p) { r.x = p[0]; r.y = p[1]; } return r; }

extern "C" {

const char* rth_last_error(void) { return g_herr.c_str(); }

RthScene* rth_scene_create(void) { try { return new RthScene(); } catch (...) { return nullptr; } }
void rth_scene_destroy(RthScene* s) { delete s; }

int rth_add_material(RthScene* s, const char* name, const RtMaterial* init)
{
    if (!s || !name) { g_herr = "rth_add_material: null argument"; return -1; }
    try {
        RtMaterial& m = s->scene.AddMaterial(name);
        if (init) m = *init;
        return (int)s->scene.materials.size() - 1;
    } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_add_texture(RthScene* s, const char* name, const RtFloat4* texels, int w, int h)
{
    if (!s || !name || !texels || w <= 0 || h <= 0) { g_herr = "rth_add_texture: bad argument"; return -1; }
    try { return s->scene.AddTexture(texels, w, h, name); } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_load_texture(RthScene* s, const char* filename, const char* name)
{
    if (!s || !filename || !name) { g_herr = "rth_load_texture: null argument"; return -1; }
    try { return s->scene.LoadTexture(filename, name); } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_add_sphere(RthScene* s, const float pos[3], float radius, const char* material) { GUARD(s->scene.AddSphere(f3(pos), radius, material)) }
int rth_add_plane(RthScene* s, const float N[3], float d, const char* material) { GUARD(s->scene.AddPlane(f3(N), d, material)) }
int rth_add_triangle(RthScene* s, const float v0[3], const float v1[3], const float v2[3], const float uv0[2], const float uv1[2],
                     const float uv2[2], const char* material, int flip)
{
    GUARD(s->scene.AddTriangle(f3(v0), f3(v1), f3(v2), f2(uv0), f2(uv1), f2(uv2), material, flip != 0))
}
int rth_add_quad(RthScene* s, const float v0[3], const float v1[3], const float v2[3], const float v3[3], const char* material, int flip)
{
    GUARD(s->scene.AddQuad(f3(v0), f3(v1), f3(v2), f3(v3), material, flip != 0))
}
int rth_add_triangles(RthScene* s, const float* verts, const float* uvs, int n, const char* material, int flip)
{
    if (!s || !verts || n < 0 || !material) { g_herr = "rth_add_triangles: bad argument"; return -1; }
    try {
        const std::string mat(material);
        s->scene.primitives.reserve(s->scene.primitives.size() + (size_t)n);
        for (int i = 0; i < n; i++) {
            const float* v = verts + (size_t)i * 9;
            const float* t = uvs ? uvs + (size_t)i * 6 : nullptr;
            s->scene.AddTriangle(f3(v), f3(v + 3), f3(v + 6), f2(t), f2(t ? t + 2 : nullptr), f2(t ? t + 4 : nullptr), mat, flip != 0);
        }
        return 0;
    } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_build_blas(RthScene* s, int startIdx, float alpha)
{
    if (!s || startIdx < 0 || startIdx >= (int)s->scene.primitives.size()) { g_herr = "rth_build_blas: bad start index"; return -1; }
    GUARD(s->scene.bvh2->alpha = alpha; s->scene.bvh2->BuildBLAS(true, startIdx))
}
int rth_set_build_threads(RthScene* s, int threads) { if (!s) return -1; s->scene.bvh2->buildThreads = threads < 1 ? 1 : threads; return 0; }
int rth_build_bvh4(RthScene* s) { GUARD(s->scene.BuildBVH4()) }
// BVH4::Convert + Collapse (bvh.cpp:695-787) on a caller-provided BVH2 node array with ONE BLAS rooted at node 0: how the tests feed
// the reference's own hand-built 13-node tree (bvh.cpp:615-674) through the collapse.
int rth_bvh4_from_nodes(const RtBVHNode2* nodes, int n, RtBVHNode4* out)
{
    if (!nodes || !out || n <= 0) { g_herr = "rth_bvh4_from_nodes: bad argument"; return -1; }
    {   // Convert / Collapse index children unchecked and recurse: refuse child ids out of range and nodes reachable twice (a cycle)
        std::vector<uint32_t> st{ 0u };
        size_t visited = 0;
        while (!st.empty()) {
            const uint32_t i = st.back(); st.pop_back();
            if (i >= (uint32_t)n || ++visited > (size_t)n) { g_herr = "rth_bvh4_from_nodes: child index out of range or a node reachable twice"; return -1; }
            if (nodes[i].count == 0) { st.push_back(nodes[i].first); st.push_back(nodes[i].first + 1); }
        }
    }
    try {
        std::vector<RtPrimitive> prims; std::vector<RtBVHInstance> blas(1);
        memset(&blas[0], 0, sizeof blas[0]);
        blas[0].bvhIdx = 0;
        BVH2 b2(prims, blas);
        b2.bvhNodes.assign(nodes, nodes + n);
        BVH4 b4(b2);
        memcpy(out, b4.Nodes().data(), sizeof(RtBVHNode4) * (size_t)n);
        return 0;
    } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_build_tlas(RthScene* s) { GUARD(delete s->tlas; s->tlas = new TLAS(*s->scene.bvh2); s->tlas->Build()) }
int rth_set_instance_transform(RthScene* s, int blas, const float invT[16])
{
    if (!s || blas < 0 || blas >= (int)s->scene.blasNodes.size()) { g_herr = "rth_set_instance_transform: bad index"; return -1; }
    memcpy(s->scene.blasNodes[blas].invT, invT, sizeof(float) * 16);
    return 0;
}

#define VIEW(vec) do { if (n) *n = (int)(vec).size(); return (vec).data(); } while (0)
const RtPrimitive*   rth_primitives(RthScene* s, int* n) { VIEW(s->scene.primitives); }
const RtMaterial*    rth_materials(RthScene* s, int* n) { VIEW(s->scene.materials); }
const RtFloat4*      rth_textures(RthScene* s, int* n) { VIEW(s->scene.textures); }
const uint32_t*      rth_lights(RthScene* s, int* n) { VIEW(s->scene.lights); }
const RtBVHNode2*    rth_bvh2_nodes(RthScene* s, int* n) { VIEW(s->scene.bvh2->bvhNodes); }
const RtBVHNode4*    rth_bvh4_nodes(RthScene* s, int* n) { if (!s->scene.bvh4) { if (n) *n = 0; return nullptr; } VIEW(s->scene.bvh4->Nodes()); }
const uint32_t*      rth_prim_idx(RthScene* s, int* n) { VIEW(s->scene.bvh2->primIdx); }
const RtTLASNode*    rth_tlas_nodes(RthScene* s, int* n) { if (!s->tlas) { if (n) *n = 0; return nullptr; } VIEW(s->tlas->tlasNodes); }
const RtBVHInstance* rth_blas_nodes(RthScene* s, int* n) { VIEW(s->scene.blasNodes); }

int rth_bvh_stats(RthScene* s, uint32_t u[5], float f[2])
{
    if (!s) return -1;
    const BVH2& b = *s->scene.bvh2;
    u[0] = b.stat_depth; u[1] = b.stat_node_count; u[2] = b.stat_spatial_splits; u[3] = b.stat_prims_clipped; u[4] = b.stat_prim_count;
    f[0] = b.stat_sah_cost; f[1] = b.stat_build_time;
    return 0;
}

int rth_camera(int width, int height, float vfov, int type, const float origin[3], const float forward[3], float aperture,
               float focalLength, RtCamera* out)
{
    if (!out || width <= 0 || height <= 0) { g_herr = "rth_camera: bad argument"; return -1; }
    CameraManager cm(width, height, vfov, type);
    if (origin) cm.cam.origin = RtFloat4{ origin[0], origin[1], origin[2], 0 };
    if (forward) cm.cam.forward = RtFloat4{ forward[0], forward[1], forward[2], 0 };
    cm.cam.aperture = aperture; cm.cam.focalLength = focalLength;
    cm.UpdateCamVec();
    *out = cm.cam;
    return 0;
}

// Renderer mirror: the scene handed in is adopted (its arrays are moved into Renderer::scene).
RthRenderer* rth_renderer_create(RthScene* scene, int width, int height, int device, int y0, int y1, int shading, int sampling,
                                 int bvh, int rr, int fireflies)
{
    if (!scene) { g_herr = "rth_renderer_create: null scene"; return nullptr; }
    try {
        RthRenderer* h = new RthRenderer();
        h->r = new Renderer(width, height, device, y0, y1);
        Scene& dst = h->r->scene; Scene& src = scene->scene;
        dst.primitives = src.primitives; dst.materials = src.materials; dst.lights = src.lights; dst.textures = src.textures;
        dst.blasNodes = src.blasNodes;
        dst.bvh2->bvhNodes = src.bvh2->bvhNodes; dst.bvh2->primIdx = src.bvh2->primIdx; dst.bvh2->alpha = src.bvh2->alpha;
        h->r->imgui.shading = shading; h->r->imgui.sampling = sampling; h->r->imgui.bvh = bvh;
        h->r->imgui.use_russian_roulette = rr != 0; h->r->imgui.filter_fireflies = fireflies != 0;
        return h;
    } catch (const std::exception& e) { g_herr = e.what(); return nullptr; }
}
void rth_renderer_destroy(RthRenderer* r) { if (r) { delete r->r; delete r; } }
int rth_renderer_init(RthRenderer* r) { GUARD(r->r->Init()) }
int rth_renderer_set_camera(RthRenderer* r, const float origin[3], const float forward[3], float fov, float aperture)
{
    try {
        CameraManager& c = r->r->camera;
        if (origin) c.cam.origin = RtFloat4{ origin[0], origin[1], origin[2], 0 };
        if (forward) c.cam.forward = RtFloat4{ forward[0], forward[1], forward[2], 0 };
        c.cam.fov = fov; c.cam.aperture = aperture; c.moved = true; c.UpdateCamVec();
        return 0;
    } catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
int rth_renderer_tick(RthRenderer* r, int frames) { GUARD(for (int i = 0; i < frames; i++) r->r->Tick(0.0f)) }
int rth_renderer_read(RthRenderer* r, RtFloat4* out, float* energy)
{
    GUARD(if (out) r->r->ReadAccum(out); if (energy) { r->r->ComputeEnergy(); *energy = r->r->energy_total; })
}
int rth_renderer_camera(RthRenderer* r, RtCamera* out) { GUARD(*out = r->r->camera.cam) }

} // extern "C"

extern "C" int rth_load_model(RthScene* s, const char* filename, const char* defaultMaterial, const float pos[3], int forceDefaultMat)
{
    if (!s || !filename || !defaultMaterial) { g_herr = "rth_load_model: null argument"; return -1; }
    try { return s->scene.LoadModel(filename, defaultMaterial, pos ? f3(pos) : float3(0, 0, 0), forceDefaultMat != 0); }
    catch (const std::exception& e) { g_herr = e.what(); return -1; }
}
extern "C" int rth_save_png(const char* file, int w, int h, const RtFloat4* data)
{
    if (!file || !data || w <= 0 || h <= 0) { g_herr = "rth_save_png: bad argument"; return -1; }
    GUARD(SavePNG(file, w, h, data))
}
// Renderer input half that touches the path (renderer.cpp:310-365): camera controller, then Tick() resets the accumulator
extern "C" int rth_renderer_camera_move(RthRenderer* r, int camdir) { GUARD(r->r->camera.Move(camdir, 0.0f)) }
extern "C" int rth_renderer_camera_mouse(RthRenderer* r, float dx, float dy) { GUARD(r->r->camera.MouseMove(dx, dy)) }
extern "C" int rth_renderer_camera_zoom(RthRenderer* r, float offset) { GUARD(r->r->camera.Zoom(offset)) }
// sample streams behind the Renderer (before Init): Tick() then renders `lanes` frames whose kernels overlap (include/rt355.h, rt_group_*)
extern "C" int rth_renderer_set_lanes(RthRenderer* r, int lanes) { if (!r || lanes < 1 || lanes > 8) { g_herr = "rth_renderer_set_lanes: lanes must be 1..8"; return -1; } r->r->lanes = lanes; return 0; }
extern "C" int rth_renderer_frames(RthRenderer* r) { return r && r->r->settings ? r->r->settings->frames : -1; }
extern "C" int rth_renderer_save_frame(RthRenderer* r, const char* file) { GUARD(r->r->SaveFrame(file)) }

// seeds[i] = (first+i+1)-th xorshift32 output from 0x12345678 — the reference's host seed loop
// (src/renderer.cpp:195-196 over template/template.cpp:711,724-730).
extern "C" int rth_seed_stream(uint32_t* out, int64_t first, int64_t n)
{
    if (!out || first < 0 || n < 0) return -1;
    uint32_t s = 0x12345678u;
    auto next = [&s]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    for (int64_t i = 0; i < first; i++) next();
    for (int64_t i = 0; i < n; i++) out[i] = next();
    return 0;
}
