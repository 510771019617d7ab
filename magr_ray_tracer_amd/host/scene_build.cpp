// scene_build.cpp — Scene primitive/material factory and CameraManager maths.
// Mirrors reference src/scene.cpp:84-176 (AddMaterial, AddSphere, AddPlane, AddQuad,
// AddTriangle: geometric normal, centroid, Heron area, light list) and
// src/camera.h:24-121 (Fov, UpdateCamVec).  File IO (LoadModel, LoadTexture) lives in
// scene_io.cpp / image_io.cpp; AddTexture takes texels that are already in memory.
#include <cmath>
#include <cstring>
#include "rt_host.h"

namespace rt355 {

Scene::Scene() { bvh2 = new BVH2(primitives, blasNodes); }
Scene::~Scene() { delete bvh4; delete bvh2; }

RtMaterial& Scene::AddMaterial(const std::string& name) // scene.cpp:84-100
{
    RtMaterial m;
    memset(&m, 0, sizeof m);
    m.texIdx = -1;
    materials.push_back(m);
    matMap_[name] = matIdx_++;
    return materials.back();
}
int Scene::MaterialIndex(const std::string& name) { return matMap_[name]; } // unknown names fall back to 0 like matMap_[name]

int Scene::AddTexture(const RtFloat4* texels, int width, int height, const std::string& name) // scene.cpp:244-256
{
    int texIdx = (int)textures.size();
    textures.insert(textures.end(), texels, texels + (size_t)width * height);
    RtMaterial& m = AddMaterial(name);
    m.texIdx = texIdx; m.isDielectric = 0; m.texW = width; m.texH = height;
    return matIdx_ - 1;
}

void Scene::AddSphere(float3 pos, float radius, const std::string& material) // scene.cpp:125-138
{
    RtPrimitive p;
    memset(&p, 0, sizeof p);
    p.objType = RT_PRIM_SPHERE;
    p.obj.sphere.pos = to4(pos);
    p.obj.sphere.r = radius; p.obj.sphere.r2 = radius * radius; p.obj.sphere.invr = 1 / radius;
    p.matIdx = matMap_[material];
    p.area = 4 * 3.14159265358979323846264f * p.obj.sphere.r2;
    primitives.push_back(p);
    if (materials[p.matIdx].isLight) lights.push_back((uint32_t)primitives.size() - 1);
}
void Scene::AddPlane(float3 N, float d, const std::string& material) // scene.cpp:140-150
{
    RtPrimitive p;
    memset(&p, 0, sizeof p);
    p.objType = RT_PRIM_PLANE;
    p.obj.plane.N = to4(N); p.obj.plane.d = d;
    p.matIdx = matMap_[material];
    primitives.push_back(p);
    if (materials[p.matIdx].isLight) lights.push_back((uint32_t)primitives.size() - 1);
}
void Scene::AddQuad(float3 v0, float3 v1, float3 v2, float3 v3, const std::string& material, bool flip,
                    float2 uv0, float2 uv1, float2 uv2, float2 uv3) // scene.cpp:152-156
{
    AddTriangle(v0, v1, v2, uv0, uv1, uv2, material, flip);
    AddTriangle(v2, v3, v0, uv2, uv3, uv1, material, flip);
}
void Scene::AddTriangle(float3 v0, float3 v1, float3 v2, float2 uv0, float2 uv1, float2 uv2,
                        const std::string& material, bool flip) // scene.cpp:158-176
{
    RtPrimitive p;
    memset(&p, 0, sizeof p);
    p.objType = RT_PRIM_TRIANGLE;
    RtTriangle& t = p.obj.triangle;
    t.v0 = to4(v0); t.v1 = to4(v1); t.v2 = to4(v2);
    t.uv0 = RtFloat2{ uv0.x, uv0.y }; t.uv1 = RtFloat2{ uv1.x, uv1.y }; t.uv2 = RtFloat2{ uv2.x, uv2.y };
    t.N = to4(normalize(cross(v1 - v0, v2 - v0)));
    if (flip) { t.N.x *= -1; t.N.y *= -1; t.N.z *= -1; t.N.w *= -1; }
    t.centroid = to4((v0 + v1 + v2) * (1 / 3.f));
    p.matIdx = matMap_[material];
    float a = length(v1 - v0), b = length(v1 - v2), c = length(v2 - v0); // Heron, scene.cpp:107-123
    float s = 0.5f * (a + b + c);
    p.area = sqrtf(s * (s - a) * (s - b) * (s - c));
    primitives.push_back(p);
    if (materials[p.matIdx].isLight) lights.push_back((uint32_t)primitives.size() - 1);
}
void Scene::BuildBVH4() { delete bvh4; bvh4 = new BVH4(*bvh2); } // scene.cpp:71

// ------------------------------------------------------------------ camera
CameraManager::CameraManager(int width, int height, float vfov, int type) // camera.h:24-34
{
    memset(&cam, 0, sizeof cam);
    aspect = (float)width / (float)height;
    cam.type = type;
    cam.origin = RtFloat4{ -10, 10, 15, 0 };
    cam.forward = RtFloat4{ 0, 0, 1, 0 };
    cam.right = RtFloat4{ 1, 0, 0, 0 };
    cam.up = RtFloat4{ 0, 1, 0, 0 };
    cam.aperture = 0.1f;
    cam.focalLength = 1;
    Fov(vfov);
}
void CameraManager::Fov(float vfov) // camera.h:101-109 (theta is a double there: float*PI / 180.0)
{
    cam.fov = vfov;
    double theta = (double)(cam.fov * 3.14159265358979323846264f) / 180.0;
    double h = tan(theta / 2);
    viewportHeight = (float)(2 * h);
    viewportWidth = aspect * viewportHeight;
}
void CameraManager::Move(int camdir, float) // camera.h:47-73 (velocity = speed, deltaTime unused there too)
{
    moved = true;
    const float v = speed;
    auto axpy = [&](const RtFloat4& d, float s) { cam.origin.x += d.x * s; cam.origin.y += d.y * s; cam.origin.z += d.z * s; cam.origin.w += d.w * s; };
    switch (camdir) {
    case 0: axpy(cam.forward, -v); break;   // Forward: the camera looks along -forward
    case 1: axpy(cam.forward, v); break;
    case 2: axpy(cam.right, -v); break;
    case 3: axpy(cam.right, v); break;
    case 4: axpy(cam.up, v); break;
    case 5: axpy(cam.up, -v); break;
    }
}
void CameraManager::MouseMove(float xOffset, float yOffset) // camera.h:75-93
{
    moved = true;
    xOffset *= mouseSensivity; yOffset *= mouseSensivity;
    yaw_ = fmodf(yaw_ + xOffset, 360.f);
    pitch_ += yOffset;
    pitch_ = pitch_ < -89.f ? -89.f : (pitch_ > 89.f ? 89.f : pitch_);
    const double yaw = (double)(yaw_ * 3.14159265358979323846264f) / 180.0, pitch = (double)(pitch_ * 3.14159265358979323846264f) / 180.0;
    float3 f((float)(cos(yaw) * cos(pitch)), (float)sin(pitch), (float)(sin(yaw) * cos(pitch)));
    cam.forward = to4(normalize(f));
}
void CameraManager::Zoom(float offset) { moved = true; Fov(cam.fov + offset); } // camera.h:95-99

void CameraManager::UpdateCamVec() // camera.h:111-121
{
    Fov(cam.fov);
    const float3 worldUp(0, 1, 0);
    float3 fwd(cam.forward);
    float3 right = normalize(cross(worldUp, fwd));
    float3 up = normalize(cross(right, fwd));
    cam.right = to4(right); cam.up = to4(up);
    cam.horizontal = to4(viewportWidth * right);
    cam.vertical = to4(viewportHeight * up);
    RtFloat4 o = cam.origin, hz = cam.horizontal, vt = cam.vertical, f = cam.forward;
    cam.topLeft = RtFloat4{ o.x - hz.x / 2 - vt.x / 2 - f.x, o.y - hz.y / 2 - vt.y / 2 - f.y,
                            o.z - hz.z / 2 - vt.z / 2 - f.z, o.w - hz.w / 2 - vt.w / 2 - f.w };
}

} // namespace rt355
