// headless_tick — the reference's main loop without its window: build a scene with the Scene API, BuildBLAS, Renderer::Init,
// Tick() N times, SaveFrame.  This is what a maintainer's renderer.cpp looks like after the swap described in INTEGRATION.md:
// the host code is the reference's call sequence (src/renderer.cpp:6-63, template main loop), the device work goes through
// librt355.so.  There is no CPU path: without a HIP device Init() throws.
//
//   headless_tick [--obj model.obj] [--tex image.png] [--size W H] [--spp N] [--bvh4] [--kajiya] [--decorrelate] [--lanes N] [--out frame.png]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <stdexcept>
#include <vector>
#include <algorithm>
#include <memory>
#include "../include/rt355.h"
#include "../include/rt355_host.h"
#include "../magr_ray_tracer_amd/host/rt_host.h"

using namespace rt355;

static void cornell_like(Scene& s, const std::string& wallMat)
{
    // a 10x10x10 room open towards +z, one emissive quad under the ceiling, two boxes worth of triangles on the floor
    const float a = 5.f;   // AddQuad takes the corners in perimeter order: (v0,v1,v2) + (v2,v3,v0), scene.cpp:152-156
    s.AddQuad(float3(-a, 0, -a), float3(-a, 0, a), float3(a, 0, a), float3(a, 0, -a), wallMat);                    // floor
    s.AddQuad(float3(-a, 2 * a, -a), float3(a, 2 * a, -a), float3(a, 2 * a, a), float3(-a, 2 * a, a), "white");   // ceiling
    s.AddQuad(float3(-a, 0, -a), float3(a, 0, -a), float3(a, 2 * a, -a), float3(-a, 2 * a, -a), "white");         // back
    s.AddQuad(float3(-a, 0, -a), float3(-a, 2 * a, -a), float3(-a, 2 * a, a), float3(-a, 0, a), "red");           // left
    s.AddQuad(float3(a, 0, -a), float3(a, 0, a), float3(a, 2 * a, a), float3(a, 2 * a, -a), "green");             // right
    s.AddQuad(float3(-1.5f, 2 * a - 0.01f, -1.5f), float3(1.5f, 2 * a - 0.01f, -1.5f), float3(1.5f, 2 * a - 0.01f, 1.5f),
              float3(-1.5f, 2 * a - 0.01f, 1.5f), "light");
    for (int k = 0; k < 2; k++) {   // two tetrahedra
        const float3 c(k ? 2.f : -2.f, 0.f, k ? -1.f : 1.f);
        const float h = k ? 3.f : 4.5f;
        const float3 p0 = c + float3(-1.5f, 0, -1.f), p1 = c + float3(1.5f, 0, -1.f), p2 = c + float3(0, 0, 1.6f), top = c + float3(0, h, 0);
        const float2 z{ 0, 0 };
        s.AddTriangle(p0, p1, top, z, z, z, k ? "mirror" : "white");
        s.AddTriangle(p1, p2, top, z, z, z, k ? "mirror" : "white");
        s.AddTriangle(p2, p0, top, z, z, z, k ? "mirror" : "white");
    }
}

int main(int argc, char** argv)
{
    int W = 1280, H = 720, spp = 64;
    std::string obj, tex, out = "frame.png";
    bool bvh4 = false, kajiya = false, decorrelate = false;
    int lanes = 1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--obj" && i + 1 < argc) obj = argv[++i];
        else if (a == "--tex" && i + 1 < argc) tex = argv[++i];
        else if (a == "--size" && i + 2 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (a == "--spp" && i + 1 < argc) spp = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--bvh4") bvh4 = true;
        else if (a == "--kajiya") kajiya = true;
        else if (a == "--decorrelate") decorrelate = true;
        else if (a == "--lanes" && i + 1 < argc) lanes = std::max(1, atoi(argv[++i]));
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        // ONE Renderer, as in the reference.  `lanes` > 1 makes it render the accumulation as that many interleaved sample streams
        // (rt_group_*, include/rt355.h): Tick() then adds `lanes` frames to the accumulator, whose kernels
        // overlap on the GPU.  The library arranges the hardware queues itself and says so on stderr if it could not.
        Renderer r(W, H);
        r.lanes = lanes;
        Scene& s = r.scene;
        // materials the way the reference's Scene constructor sets them up (scene.cpp:14-43)
        { RtMaterial& mt = s.AddMaterial("white"); mt.color = RtFloat4{ 0.9f, 0.9f, 0.9f, 0 }; }
        { RtMaterial& mt = s.AddMaterial("red"); mt.color = RtFloat4{ 0.9f, 0.15f, 0.1f, 0 }; }
        { RtMaterial& mt = s.AddMaterial("green"); mt.color = RtFloat4{ 0.15f, 0.8f, 0.2f, 0 }; }
        { RtMaterial& mt = s.AddMaterial("mirror"); mt.color = RtFloat4{ 0.9f, 0.9f, 0.9f, 0 }; mt.specular = 0.5f; }
        { RtMaterial& mt = s.AddMaterial("light"); mt.color = RtFloat4{ 1, 1, 1, 0 }; mt.isLight = 1; mt.emittance = RtFloat4{ 40, 40, 40, 0 }; }
        std::string floorMat = "white";
        if (!tex.empty()) { s.LoadTexture(tex, "floor-texture"); floorMat = "floor-texture"; }
        cornell_like(s, floorMat);
        if (!obj.empty()) { const int n = s.LoadModel(obj, "white", float3(0, 0, 0), false); printf("loaded %d triangles from %s\n", n, obj.c_str()); }
        const auto t0 = std::chrono::steady_clock::now();
        s.bvh2->BuildBLAS(true, 0);
        const double buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        r.imgui.bvh = bvh4 ? 1 : 0;
        r.imgui.shading = kajiya ? 0 : 1;
        r.camera.cam.origin = RtFloat4{ 0, 5, 14, 0 };
        r.camera.cam.forward = RtFloat4{ 0, 0, 1, 0 };     // the camera looks along -forward
        r.camera.cam.aperture = 0.0f;
        r.camera.Fov(60);
        r.Init();
        if (decorrelate) {
            // The reference seeds pixel i with the (i+1)-th output of ONE xorshift32 stream and then advances every pixel with the same
            // xorshift32 (template.cpp:724-730, util.cl:50-56): neighbouring pixels draw the same numbers one step apart, which shows
            // as horizontal streaks at low sample counts.  That is what parity reproduces by default; a caller who does not need
            // parity hands over independent seeds (here the reference's own, unused, initSeed = WangHash((i + 1) * 17), util.cl:37-48).
            std::vector<uint32_t> seeds((size_t)W * H);
            for (int m = 0; m < lanes; m++) {
                for (size_t i = 0; i < seeds.size(); i++) {
                    uint32_t v = ((uint32_t)(i + (size_t)m * seeds.size()) + 1u) * 17u;
                    v = (v ^ 61u) ^ (v >> 16); v *= 9u; v = v ^ (v >> 4); v *= 0x27d4eb2du; v = v ^ (v >> 15);
                    seeds[i] = v ? v : 1u;
                }
                if (rt_set_seeds(rt_group_lane(r.group, m), seeds.data(), (int64_t)seeds.size())) throw std::runtime_error(rt_last_error());
            }
        }
        const auto t1 = std::chrono::steady_clock::now();
        const int ticks = (spp + lanes - 1) / lanes;                     // a Tick() is `lanes` frames
        for (int i = 0; i < ticks; i++) r.Tick(0.016f);                  // Tick() only enqueues
        r.ComputeEnergy();                                               // (reads the accumulator: synchronises)
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        r.SaveFrame(out.c_str());
        const int done = ticks * lanes;
        printf("headless_tick: %zu primitives, BVH build %.1f ms, %d spp at %dx%d in %.1f ms (%.1f M samples/s, %d lane%s, %d concurrent), energy %.6g, wrote %s\n",
               r.scene.primitives.size(), buildMs, done, W, H, ms, (double)W * H * done / ms / 1e3, lanes, lanes > 1 ? "s" : "",
               rt_group_concurrency(r.group), (double)r.energy_total, out.c_str());
    } catch (const std::exception& e) {
        fprintf(stderr, "headless_tick: %s\n", e.what());
        return 1;
    }
    return 0;
}
