// renderer.cpp — Renderer mirror (reference: src/renderer.cpp:6-94,126-140,289-301).
// Same call shape as the reference (Init, Tick, RayTrace, FocusCamera, ComputeEnergy), but
// every Kernel::Run / Buffer::CopyToDevice of the reference becomes one call of the C-ABI in
// include/rt355.h.  There is no CPU path: if the device library cannot create a context the
// error is reported and nothing is rendered.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "../../include/rt355.h"
#include "rt_host.h"

namespace rt355 {

static void check(int rc, const char* what)
{
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

Renderer::Renderer(int w, int h, int dev, int band0, int band1)
    : camera(w, h), width(w), height(h), device(dev), y0(band0), y1(band1 < 0 ? h : band1)
{
}
Renderer::~Renderer()
{
    if (group) rt_group_destroy(group);
    delete tlas;
    delete settings;
}

void Renderer::Init() // renderer.cpp:6-21 (+ InitBuffers :142-209, InitWavefrontKernels :211-263)
{
    settings = new RtSettings();
    memset(settings, 0, sizeof *settings);
    settings->tracerType = 1; // KAJIYA
    settings->antiAliasing = 1;
    settings->renderBVH = 0;
    settings->frames = 1;
    delete tlas;
    tlas = new TLAS(*scene.bvh2);
    tlas->Build();
    if (imgui.bvh == RT_ACCEL_BVH4 && !scene.bvh4) scene.BuildBVH4();
    settings->numPrimitives = (int)scene.primitives.size();
    settings->numLights = (int)scene.lights.size();

    RtConfig cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = width; cfg.height = height; cfg.y0 = y0; cfg.y1 = y1;
    cfg.max_bounces = RT_MAX_BOUNCES;
    cfg.shading = imgui.shading; cfg.sampling = imgui.sampling; cfg.accel = imgui.bvh;
    cfg.russian_roulette = imgui.use_russian_roulette; cfg.filter_fireflies = imgui.filter_fireflies;
    cfg.device = device;
    // `lanes` sample streams behind this one Renderer (rt_group_*, include/rt355.h): 1 = the reference's single in-order queue, bit for
    // bit; more keep the GPU full (the lanes' frames overlap) and turn one Tick() into `lanes` frames of the same accumulation
    if (group) { rt_group_destroy(group); group = nullptr; ctx = nullptr; }
    check(rt_group_create(&cfg, lanes < 1 ? 1 : lanes, &group), "rt_group_create");
    ctx = rt_group_lane(group, 0);
    const bool q = imgui.bvh == RT_ACCEL_BVH4;
    check(rt_group_upload_scene(group, scene.primitives.data(), (int)scene.primitives.size(), scene.materials.data(), (int)scene.materials.size(),
                          scene.textures.data(), (int)scene.textures.size(), scene.lights.data(), (int)scene.lights.size(),
                          q ? (const void*)scene.bvh4->Nodes().data() : (const void*)scene.bvh2->bvhNodes.data(),
                          q ? (int)scene.bvh4->Nodes().size() : (int)scene.bvh2->bvhNodes.size(),
                          scene.bvh2->primIdx.data(), (int)scene.bvh2->primIdx.size(),
                          tlas->tlasNodes.data(), (int)tlas->tlasNodes.size(), scene.blasNodes.data(), (int)scene.blasNodes.size()),
          "rt_group_upload_scene");
    check(rt_group_seed(group, 0), "rt_group_seed");   // lane m: the (m * PIXELS + i + 1)-th outputs of the host stream (renderer.cpp:195-196)
    camera.UpdateCamVec();
    FocusCamera(width / 2, height / 2);
}

void Renderer::Tick(float) // renderer.cpp:26-63
{
    camera.UpdateCamVec();
    if (camera.moved || imgui.reset_every_frame) {
        check(rt_group_reset(group), "rt_group_reset");
        camera.moved = false;
        settings->frames = 1;
    }
    if (settings->renderBVH) settings->frames = 1;
    RayTrace();
    settings->frames += rt_group_lanes(group);   // renderer.cpp:53 `frames++`, once per frame a lane has added to the accumulation
}
void Renderer::RayTrace() // renderer.cpp:64-94, once per lane: the lanes' launches are queued interleaved and overlap on the GPU
{
    check(rt_group_render(group, &camera.cam, settings, rt_group_lanes(group)), "rt_group_render");
}
void Renderer::FocusCamera(int x, int y) // renderer.cpp:289-301
{
    float t = RT_REALLYFAR;
    check(rt_focus(ctx, x, y, &camera.cam, &t), "rt_focus");
    settings->focalLength = t;
    if (t != RT_REALLYFAR) camera.cam.focalLength = t;
}
void Renderer::ReadAccum(RtFloat4* out) { check(rt_group_read_accum(group, out), "rt_group_read_accum"); }
void Renderer::SaveFrame(const char* file) // renderer.cpp:303-308 after PostProc (:95-124)
{
    std::vector<RtFloat4> img((size_t)width * height);
    // Tick() has already advanced settings->frames past the frame whose image is shown (renderer.cpp:49-53)
    int shown = settings->frames > 1 ? settings->frames - 1 : 1;
    check(rt_group_postproc(group, shown, vignet_strength, gamma_strength, chromatic_strength, img.data(), nullptr), "rt_group_postproc");
    SavePNG(file, width, height, img.data());
}
void Renderer::ComputeEnergy() // renderer.cpp:126-140
{
    std::vector<RtFloat4> px((size_t)width * height);
    ReadAccum(px.data());
    energy_total = 0;
    for (const RtFloat4& p : px) { energy_total += p.x; energy_total += p.y; energy_total += p.z; }
    energy_total *= 1 / (float)(settings->frames);
}

} // namespace rt355
