// rt_host.h — host-side mirror of the reference's Scene / BVH2 / BVH4 / TLAS /
// CameraManager / Renderer surface (reference: src/scene.h, src/bvh.h, src/tlas.h,
// src/camera.h, src/renderer.h).  Same class and method names, same argument meaning;
// the arrays these classes produce use the wire format of include/rt355_types.h and
// are handed to the device path through the C-ABI of include/rt355.h.
//
// All arithmetic is strict binary32 in source order (build with -ffp-contract=off):
// the reference's own host build used MSVC /fp:fast and is not reproducible
// (SURVEY.md Appendix B #13).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include "../../include/rt355_types.h"

struct RtCtx;
struct RtGroup;

namespace rt355 {

struct float2 { float x = 0, y = 0; };
struct float3 {
    float x = 0, y = 0, z = 0;
    float3() = default;
    float3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit float3(float a) : x(a), y(a), z(a) {}
    explicit float3(const RtFloat4& v) : x(v.x), y(v.y), z(v.z) {}
    float  operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline float3 operator+(const float3& a, const float3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline float3 operator-(const float3& a, const float3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float3 operator*(const float3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline float3 operator*(float s, const float3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline float3 operator+(const float3& a, float s) { return { a.x + s, a.y + s, a.z + s }; }
inline float3 operator-(const float3& a, float s) { return { a.x - s, a.y - s, a.z - s }; }
inline RtFloat4 to4(const float3& v, float w = 0.0f) { return RtFloat4{ v.x, v.y, v.z, w }; }
float  dot(const float3& a, const float3& b);
float3 cross(const float3& a, const float3& b);
float3 normalize(const float3& v);
float  length(const float3& v);

// Axis-aligned box with the min/max semantics of the template's SSE aabb
// (reference: template/precomp.h:889-941): 4 lanes, w held at 0, empty = (+1e34, -1e34).
struct Aabb {
    float bmin[4] = { 1e34f, 1e34f, 1e34f, 0 };
    float bmax[4] = { -1e34f, -1e34f, -1e34f, 0 };
    void  Grow(const float3& p);
    void  Grow(const RtFloat4& p);
    void  Grow(const Aabb& b);
    Aabb  Union(const Aabb& b) const;
    Aabb  Intersection(const Aabb& b) const;
    float Area() const;
    float Center(int axis) const { return (bmin[axis] + bmax[axis]) * 0.5f; }
};

struct BVHPrimData { Aabb box; uint32_t idx = 0; };

// reference: src/bvh.h:4-40
class BVH2 {
public:
    BVH2(std::vector<RtPrimitive>& prims, std::vector<RtBVHInstance>& blasNodes);
    void     BuildBLAS(bool statistics, int startIdx);
    int      buildThreads = 1;   // > 1: subtrees are built by parallel tasks, then numbered in the reference's LIFO order (same arrays)
    uint32_t Depth(uint32_t nodeIdx) const;
    uint32_t Count(uint32_t nodeIdx) const;
    float    TotalCost(uint32_t nodeIdx) const;
    std::vector<RtBVHNode2> bvhNodes;
    std::vector<uint32_t>   primIdx;
    float    alpha = 1.f;
    uint32_t stat_depth = 0, stat_node_count = 0, stat_spatial_splits = 0, stat_prims_clipped = 0, stat_prim_count = 0, stat_forced_leaves = 0;
    float    stat_sah_cost = 0, stat_build_time = 0;
    std::vector<RtBVHInstance>& blasNodes;
private:
    using Refs = std::vector<BVHPrimData>;
    void  BuildBVH(uint32_t root, Refs data);
    struct TNode;                                       // temporary pointer tree of the parallel build
    TNode* BuildSubtree(Refs refs, float rootArea, int depth, int& budget);
    void  FlattenLIFO(uint32_t root, TNode* tree);
    void  UpdateNodeBounds(uint32_t nodeIdx, const Refs& prims);
    Refs  CreateBVHPrimData(int startIdx) const;
    float CalculateNodeCost(const RtBVHNode2& node, uint32_t count) const;
    float FindBestObjectSplitPlane(int& axis, float& splitPos, float& overlap, const Refs& prims) const;
    void  ObjectSplit(int axis, float splitPos, const Refs& prims, Refs& left, Refs& right) const;
    float FindBestSpatialSplitPlane(int& axis, float& splitPos, const Refs& prims) const;
    void  SpatialSplit(int axis, float splitPos, const Refs& prims, Refs& left, Refs& right, uint32_t& clippedCount) const;
    bool  ClipTriangleToAABB(const Aabb& bounds, float3 v0, float3 v1, float3 v2, Aabb& out) const;
    bool  ClipSphereToAABB(const Aabb& bounds, float3 pos, float r, Aabb& out) const;
    std::vector<RtPrimitive>& primitives_;
    uint32_t rootNodeIdx_ = 0, nodesUsed_ = 0;
};

// reference: src/bvh.h:41-56
class BVH4 {
public:
    explicit BVH4(BVH2& bvh2);
    std::vector<RtBVHNode4>& Nodes() { return bvhNodes; }
    std::vector<uint32_t>&   Idx() { return bvh2.primIdx; }
    uint32_t Depth(uint32_t nodeIdx) const;
    uint32_t Count(uint32_t nodeIdx) const;
private:
    BVH2& bvh2;
    std::vector<RtBVHNode4> bvhNodes;
    void Convert();
    void Collapse(int index);
    int  GetChildCount(const RtBVHNode4& node) const;
};

// reference: src/tlas.h:2-12
class TLAS {
public:
    explicit TLAS(BVH2& bvh2);
    void Build();
    std::vector<RtTLASNode> tlasNodes;
private:
    int FindBestMatch(const int* list, int N, int A) const;
    BVH2& bvh2_;
    uint32_t nodesUsed_ = 0;
};

// reference: src/scene.h:5-34 (LoadModel / LoadTexture file IO: scene_io.cpp, image_io.cpp)
class Scene {
public:
    Scene();
    ~Scene();
    RtMaterial& AddMaterial(const std::string& name);
    void AddSphere(float3 pos, float radius, const std::string& material);
    void AddPlane(float3 N, float d, const std::string& material);
    void AddQuad(float3 v0, float3 v1, float3 v2, float3 v3, const std::string& material, bool flipNormal = false,
                 float2 uv0 = { 0, 0 }, float2 uv1 = { 1, 0 }, float2 uv2 = { 0, 1 }, float2 uv3 = { 1, 1 });
    void AddTriangle(float3 v0, float3 v1, float3 v2, float2 uv0, float2 uv1, float2 uv2, const std::string& material,
                     bool flipNormal = false);
    int  AddTexture(const RtFloat4* texels, int width, int height, const std::string& name); // LoadTexture minus the file read
    // reference: Scene::LoadTexture (scene.cpp:244-256); PNG, JPEG, TGA and Radiance HDR files (image_io.cpp, jpeg_io.cpp); returns the material index
    int  LoadTexture(const std::string& filename, const std::string& name);
    int  MaterialIndex(const std::string& name);
    bool HasMaterial(const std::string& name) const { return matMap_.count(name) != 0; }
    // reference: Scene::LoadModel (scene.cpp:178-243), OBJ + MTL diffuse-texture names; returns the triangles added
    int  LoadModel(const std::string& filename, const std::string& defaultMaterial, float3 pos = float3(0, 0, 0), bool forceDefaultMat = false);
    void BuildBVH4();
    std::vector<RtPrimitive>   primitives;
    std::vector<RtMaterial>    materials;
    std::vector<uint32_t>      lights;
    std::vector<RtFloat4>      textures;
    std::vector<RtBVHInstance> blasNodes;
    BVH2* bvh2 = nullptr;
    BVH4* bvh4 = nullptr;
private:
    std::map<std::string, int> matMap_;
    int matIdx_ = 0;
};

// reference: SaveImageF (template/template.cpp:1629-1644) behind Renderer::SaveFrame (renderer.cpp:303-308)
void SavePNG(const std::string& file, int w, int h, const RtFloat4* data);
// LoadImageF (template/template.cpp:1613-1627): w*h RGB float triples, top row first (image_io.cpp)
constexpr int64_t kMaxTexturePixels = int64_t(1) << 26;   // larger headers are treated as corrupt files, not as allocation requests
std::vector<float> LoadImageF(const std::string& file, int& w, int& h);
// ITU-T T.81 baseline / progressive Huffman JPEG -> 8-bit RGB, top row first (jpeg_io.cpp)
void DecodeJpeg(const std::vector<uint8_t>& bytes, const std::string& file, int& w, int& h, std::vector<uint8_t>& rgb);

// reference: src/camera.h:7-122 (aspect = width/height is a run-time value here)
class CameraManager {
public:
    CameraManager(int width, int height, float vfov = 110, int type = RT_CAM_PROJECTION);
    RtCamera cam;
    float viewportHeight = 0, viewportWidth = 0, aspect = 1;
    bool  moved = true;
    void Fov(float vfov);
    void UpdateCamVec();
    // controller (camera.h:47-99): dir 0..5 = Forward, Backwards, Left, Right, Up, Down
    void Move(int camdir, float deltaTime);
    void MouseMove(float xOffset, float yOffset);
    void Zoom(float offset);
    float mouseSensivity = 0.5f, speed = 1.f;
private:
    float yaw_ = 0, pitch_ = 0;   // uninitialised in the reference (camera.h:20-21); zero here
};

// reference: src/renderer.h:23-26,36 (ImGuiData kernel variants)
struct RenderOptions {
    int  shading = 1;          // 0 SHADING_SIMPLE, 1 SHADING_NEE
    int  sampling = 1;         // 0 SAMPLING_HEMISPHERE, 1 SAMPLING_COSINE
    int  bvh = 0;              // 0 USE_BVH2, 1 USE_BVH4
    bool use_russian_roulette = true;
    bool filter_fireflies = true;
    bool reset_every_frame = false;
};

// reference: src/renderer.h:44-120, src/renderer.cpp:6-94,126-140,289-301.
// Owns the Scene, camera, TLAS and Settings like the reference's Renderer and drives
// the device path exclusively through the C-ABI (rt_create / rt_upload_scene /
// rt_render / ...).  Resolution and row band are run-time values.
class Renderer {
public:
    Renderer(int width, int height, int device = 0, int y0 = 0, int y1 = -1);
    ~Renderer();
    void Init();                       // renderer.cpp:6-21
    void Tick(float deltaTime);        // renderer.cpp:26-63 (one accumulated frame)
    void RayTrace();                   // renderer.cpp:64-94
    void ComputeEnergy();              // renderer.cpp:126-140
    void FocusCamera(int x, int y);    // renderer.cpp:289-301
    void ReadAccum(RtFloat4* out);     // accumBuffer->CopyFromDevice()
    void SaveFrame(const char* file);  // renderer.cpp:303-308 (PostProc chain + saveImage + SaveImageF)
    float vignet_strength = 0, chromatic_strength = 0, gamma_strength = .9f;   // renderer.h:28-30
    Scene          scene;
    CameraManager  camera;
    RtSettings*    settings = nullptr;
    TLAS*          tlas = nullptr;
    RenderOptions  imgui;
    float          energy_total = 0;
    RtGroup*       group = nullptr;    // the lanes behind this Renderer (include/rt355.h, rt_group_*)
    RtCtx*         ctx = nullptr;      // lane 0 (focus pick, counters, stage-level debugging)
    int            lanes = 1;          // set before Init(): sample streams whose frames overlap on the GPU; Tick() = `lanes` frames
    int width, height, device, y0, y1;
};

} // namespace rt355
