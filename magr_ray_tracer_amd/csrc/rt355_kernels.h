// rt355_kernels.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the wavefront
// path tracer hot path: generate -> extend -> shade -> (compact) -> connect.
//
// What each kernel replaces in the reference (paths relative to the reference repo):
//   k_generate  src/cl/wavefront.cl:14-34  + camera.cl:6-46, ray.cl:4-19
//   k_extend    src/cl/wavefront.cl:35-75  + tlas.cl:3-77, bvh.cl:3-96, primitives.cl:11-89
//   k_shade     src/cl/wavefront.cl:76-142 + shading.cl:7-169, ray.cl:21-72, glass.cl:4-49,
//               primitives.cl:91-189, skydome.cl:5-7
//   k_connect   src/cl/wavefront.cl:144-201
//   k_focus     src/cl/wavefront.cl:203-224, k_reset :226-229
//
// Design (DESIGN.md has the full rationale):
//  * Ray queues are struct-of-arrays of 16-byte elements (O, D, intensity, {t,prim,u,v},
//    8-byte meta) so a wave reads 1 KiB contiguous per instruction, instead of the
//    reference's 128-byte AoS Ray updated in place in global memory.
//  * Schedule S1 of SURVEY.md §8(c): queue slot g is shaded with RNG stream seeds[g] and
//    survivors keep their relative order.  The reference's global atomic_inc/atomic_dec
//    counters become per-wave __ballot masks + popcount prefix inside k_shade and a
//    single-pass ordered scan across workgroups (decoupled look-back): survivors are
//    written straight to their final, stable queue position — deterministic, one kernel.
//  * The traversal stack lives in LDS (one column per lane, conflict-free); the ray
//    lives in registers; results are written once.
//  * Float discipline = oracle/oracle.c header: IEEE + - * /, sqrt; dot/cross as the fma
//    chains of ROCm's OpenCL library; compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt355_types.h"

namespace rt355dev {

#define RT_FORCEINLINE __device__ __forceinline__
static constexpr int   kBlock = 256;           // threads per workgroup = 4 wave64
static constexpr float kEps = RT_EPSILON;
static constexpr float kFar = RT_REALLYFAR;
static constexpr float kPi = 3.14159274101257f;     // M_PI_F
static constexpr float kInvPi = 0.31830987334251f;  // M_1_PI_F

// ------------------------------------------------------------------ device views
struct DevScene {
    const RtPrimitive*   prims;
    const RtMaterial*    mats;
    const float4*        tex;
    const uint32_t*      lights;
    const RtBVHNode2*    bvh2;
    const RtBVHNode4*    bvh4;
    const uint32_t*      primIdx;
    const RtTLASNode*    tlas;
    const RtBVHInstance* blas;
    // derived at upload from the unchanged BVH2 arrays (layout 1, see traverse_bvh2_packed)
    const float4*        pairs;      // [nNodes][4]  both child boxes + encoded child entries of interior node i
    const float4*        triRecs;    // [nIdx][3]    leaf-ordered triangle vertices + primitive id
    const uint32_t*      rootEntry;  // [nBlas]      encoded root of every instance
    const float4*        shadeRecs;  // [nPrims]     {N.xyz, bits(matIdx | (N.w is -0) << 27 | objType << 28)}: what shade() needs of a 128-B Primitive, in 16 B
    const float4*        tlasPairs;  // [nTlas][4]    TLAS interior node i: both child boxes + encoded children (leaf: kLeafBit | BLASidx), one 64-B fetch
    const float4*        instRecs;   // [nBlas][4]    rows 0..2 of invT + {encoded BLAS root (layout 1), bvhIdx}: one 64-B fetch per instance visit
    uint32_t             tlasRoot;   // encoded TLAS root (a leaf when the scene has one BLAS)
    const float4*        tlasPairsP; // [nTlas][4]    the same records with the children in the tagged encoding of k_trace_persist_tlas (kTagTlas / kTagInst)
    uint32_t             tlasRootP;  // TLAS root in that encoding
    const float4*        lightRecs;  // [nLights][8]  what NEE needs of light li in one place: objData[0..63], {objType, area}, emittance of its material
    const float4*        quads;      // [nNodes][8]  layout 1 of the BVH4: four child boxes + four encoded child entries (128 B)
    int32_t nLights, nPrims, nBlas, nTex;
};
struct DevQueues {
    // ray queues (compacted), capacity nPix each; bounce b lives in set b & 1 (shade reads one, writes the other)
    float4* O[2]; float4* D[2]; float4* inten[2]; uint2* meta[2];
    float4* hit;       // {t, primIdx, u, v} of the queue extend() just traced
    // shadow queue (compacted), capacity max_bounces * nPix
    float4* sA; float4* sB; float4* sC;
    // single-pass scan state of k_shade, one set per bounce parity: [1 + t] = status of tile t
    unsigned long long* tile[2];
    unsigned long long* super[2];   // [s] = status of super-tile s (64 consecutive tiles), same encoding
    unsigned long long* supAcc[2];  // [s] = running total of super-tile s: [63:40] tiles arrived, [39:20] extension rays, [19:0] shadow rays
    int32_t* nRays;    // [RT_MAX_BOUNCES+2]  rays entering bounce b
    int32_t* nShadow;  // [RT_MAX_BOUNCES+2]  shadow rays of bounce b occupy [nShadow[b], nShadow[b+1])
    int32_t* fault;    // [1] set to 1 by a kernel whose bounded wait expired (host turns it into RT_E_DEVICE)
    int32_t* cursor;   // [2*(RT_MAX_BOUNCES+2)] work-queue heads of the persistent kernels (extend: [b], connect: [9+b])
    int32_t* shadeTicket; // [(RT_MAX_BOUNCES+1) * kTicketClasses * kTicketStride] k_shade's tile tickets, one counter per bounce and class
    uint32_t* seeds;   // one RNG stream per band slot
    float4* accum;     // full frame, indexed by global pixel index
    int32_t* steps;    // per-ray steps of the last extend (debug / heat map), may be null
    unsigned long long* ctrExtend;  // [gridMax][kCtrCols] per-block partial work counters
    unsigned long long* ctrConnect; // [gridMax][kCtrCols]
    uint32_t* spill;   // [spillEntries][spillStride] deep ends of the traversal stacks (SPILL instantiations), may be null
    uint32_t spillStride, stackCap;   // lanes of the largest SPILL launch; LDS entries per lane of those launches
    uint32_t tlasLdsEntries;          // LDS stack entries per lane of k_trace_persist_tlas (the world-ray backup, if any, sits behind them)
    int32_t nPix, firstPixel, width, height;
};
// k_shade hands its tiles out by ticket.  ONE counter: any running workgroup draws the smallest tile not drawn yet, so the ordered scan
// depends on nothing but "some workgroup of the launch is running".  (A counter per class of workgroups - class = blockIdx mod 32,
// tiles c, c + 32, ... - takes the tickets off one memory channel and was 5 % faster for k_shade, but workgroups go to the XCDs
// round-robin, so an XCD then serves only 4 of the 32 classes, and two processes with two contexts each dead-locked within a few
// frames: four partially resident k_shade grids, each holding the XCD another one needed.  tests: test_two_processes_with_two_lanes_...)
static constexpr int kTicketClasses = 1, kTicketStride = 1024, kCursorWords = 2 * (RT_MAX_BOUNCES + 2);
struct DevVariant { int32_t shading, sampling, accel, rr, fireflies, maxBounces; };

// meta.y bit layout
static constexpr uint32_t kMetaBounceMask = 0xffu, kMetaInside = 0x100u, kMetaLastSpec = 0x200u;

// ------------------------------------------------------------------ float4 helpers
// Non-temporal accesses for queue data at its LAST use (k_shade's input entries and hit records, k_accumulate's shadow records) and for
// k_generate's primary rays: what streams through once does not push node and triangle records out of the L2s.  Measured with the
// accesses switched one group at a time (profiles/r02_nt_streams.txt): +1.0 % with three lanes, +1.4 % for one context; the same hint on
// k_shade's STORES costs a context alone 1 % (the next extend reads them back soon), on the traversal kernels' ray loads and hit stores nothing.
typedef float nt_v4f __attribute__((ext_vector_type(4)));
typedef unsigned nt_v2u __attribute__((ext_vector_type(2)));
RT_FORCEINLINE float4 ldnt4(const float4* p) { nt_v4f t = __builtin_nontemporal_load((const nt_v4f*)p); return *(float4*)&t; }
RT_FORCEINLINE uint2 ldnt2(const uint2* p) { nt_v2u t = __builtin_nontemporal_load((const nt_v2u*)p); return *(uint2*)&t; }
RT_FORCEINLINE void stnt4(float4* p, float4 v) { __builtin_nontemporal_store(*(nt_v4f*)&v, (nt_v4f*)p); }
RT_FORCEINLINE void stnt2(uint2* p, uint2 v) { __builtin_nontemporal_store(*(nt_v2u*)&v, (nt_v2u*)p); }
RT_FORCEINLINE float4 mk4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
RT_FORCEINLINE float4 splat(float s) { return make_float4(s, s, s, s); }
RT_FORCEINLINE float4 add4(float4 a, float4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RT_FORCEINLINE float4 sub4(float4 a, float4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
RT_FORCEINLINE float4 mul4(float4 a, float4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
RT_FORCEINLINE float4 muls(float4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
RT_FORCEINLINE float4 neg4(float4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }
RT_FORCEINLINE float4 ld4(const RtFloat4& v) { return *reinterpret_cast<const float4*>(&v); }
RT_FORCEINLINE float dot3(float4 a, float4 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
RT_FORCEINLINE float dot4(float4 a, float4 b) { return __fmaf_rn(a.w, b.w, __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x))); }
RT_FORCEINLINE float4 cross4(float4 a, float4 b)
{
    return mk4(__fmaf_rn(a.y, b.z, b.y * (-a.z)), __fmaf_rn(a.z, b.x, b.z * (-a.x)), __fmaf_rn(a.x, b.y, b.x * (-a.y)), 0.0f);
}
// -DRT355_REF_BUILTINS (a second library, librt355_refb.so, built by build.py; NOT the shipped default): the builtins below evaluate the
// very instruction sequences ROCm's OpenCL library gives the reference's kernels - normalize() = v * rsqrt(dot) with the hardware
// v_rsq_f32 (opencl.bc _Z9normalizeDv4_f -> __ocml_rsqrt_f32), length() = the hardware v_sqrt_f32 (llvm.sqrt with !fpmath 3 ulp in
// _Z6lengthDv4_f), exp / sin / cos / acospi / atan2pi = the same ocml bitcode functions the OpenCL builtins resolve to (HIP links the same
// ocml.bc with the same control constants: correctly rounded sqrt on, denormals on, no unsafe math).  With them the HIP path is held to
// the reference's own kernels WITHOUT the few-ulp allowance of DESIGN.md section 2, on uncurated frames (tests/test_gpu_reference.py).
// The default build keeps IEEE 1/sqrt and the Cephes sequences: those are what a CPU (the oracle) can reproduce bit for bit.
#ifdef RT355_REF_BUILTINS
extern "C" __device__ float __ocml_acospi_f32(float);
extern "C" __device__ float __ocml_atan2pi_f32(float, float);
RT_FORCEINLINE float rt_len_sqrt(float d) { return __builtin_amdgcn_sqrtf(d); }
RT_FORCEINLINE float rt_rsqrt(float d) { return __ocml_rsqrt_f32(d); }
#else
RT_FORCEINLINE float rt_len_sqrt(float d) { return sqrtf(d); }
RT_FORCEINLINE float rt_rsqrt(float d) { return 1.0f / sqrtf(d); }
#endif
RT_FORCEINLINE float length4(float4 v)
{
    float d = dot4(v, v);
    if (d < 1.17549435e-38f) { float4 s = muls(v, 0x1p+86f); return rt_len_sqrt(dot4(s, s)) * 0x1p-86f; }
    if (d == INFINITY) { float4 s = muls(v, 0x1p-66f); return rt_len_sqrt(dot4(s, s)) * 0x1p+66f; }
    return rt_len_sqrt(d);
}
RT_FORCEINLINE float sel_inf(float x) { return copysignf(isinf(x) ? 1.0f : 0.0f, x); }
RT_FORCEINLINE float4 normalize4(float4 v)
{
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f) return v;
    float d = dot4(v, v);
    if (d < 1.17549435e-38f) { v = muls(v, 0x1p+86f); d = dot4(v, v); }
    else if (d == INFINITY) {
        v = muls(v, 0x1p-66f); d = dot4(v, v);
        if (d == INFINITY) { v = mk4(sel_inf(v.x), sel_inf(v.y), sel_inf(v.z), sel_inf(v.w)); d = dot4(v, v); }
    }
    float r = rt_rsqrt(d);
    return muls(v, r);
}

// ------------------------------------------------------------------ transcendentals
// exp (Beer's law, glass.cl:4-9), sin / cos (fisheye camera, sphere-light sampling), acos / atan2 (sphere texture lookup).  The
// device library's versions lean on hardware approximations (v_exp_f32, v_sin_f32 ...) that no CPU reproduces, and one last-bit
// difference in a Fresnel draw or a texel index flips a whole path; so this path and the CPU oracle both evaluate the single-precision
// Cephes algorithms (S. Moshier: expf.c, sinf.c, asinf.c, atanf.c) as plain sequences of IEEE + - * / sqrt - transcribed separately
// here and in oracle/oracle.c - and agree bit for bit on every scene.  They are within 2 ulp of the correctly rounded value, i.e. as
// close to the reference's builtins as those are to each other.
#ifdef RT355_REF_BUILTINS
RT_FORCEINLINE float rt_expf(float x) { return __ocml_exp_f32(x); }
RT_FORCEINLINE float rt_sinf(float x) { return __ocml_sin_f32(x); }
RT_FORCEINLINE float rt_cosf(float x) { return __ocml_cos_f32(x); }
#else
RT_FORCEINLINE float rt_expf(float x)
{
    if (x != x) return x;
    if (x > 88.7228394f) return INFINITY;
    if (x < -103.972076f) return 0.0f;
    const float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = p * r + 1.3981999507E-3f;
    p = p * r + 8.3334519073E-3f;
    p = p * r + 4.1665795894E-2f;
    p = p * r + 1.6666665459E-1f;
    p = p * r + 5.0000001201E-1f;
    float e = p * z;
    e = e + r;
    e = e + 1.0f;
    int k = (int)n;
    if (k > 127) { e = e * 0x1p127f; k -= 127; }
    else if (k < -126) { e = e * 0x1p-126f; k += 126; }
    return e * __uint_as_float((uint32_t)(k + 127) << 23);
}
RT_FORCEINLINE float rt_trig_reduce(float ax, int& j)   // octant (made even-up) and remainder in [-pi/4, pi/4], three-term Cody-Waite
{
    j = (int)(1.27323954473516f * ax);
    float y = (float)j;
    if (j & 1) { j += 1; y = y + 1.0f; }
    j &= 7;
    float r = ax - y * 0.78515625f;
    r = r - y * 2.4187564849853515625e-4f;
    r = r - y * 3.77489497744594108e-8f;
    return r;
}
RT_FORCEINLINE float rt_sin_kernel(float x, float z) { float p = -1.9515295891E-4f * z + 8.3321608736E-3f; p = p * z - 1.6666654611E-1f; p = p * z; p = p * x; return p + x; }
RT_FORCEINLINE float rt_cos_kernel(float z) { float p = 2.443315711809948E-005f * z - 1.388731625493765E-003f; p = p * z + 4.166664568298827E-002f; p = p * z; p = p * z; p = p - 0.5f * z; return p + 1.0f; }
RT_FORCEINLINE float rt_sinf(float x)
{
    if (x != x) return x;
    bool neg = x < 0.0f;
    const float ax = neg ? -x : x;
    if (ax > 8192.0f) return 0.0f;
    int j; const float r = rt_trig_reduce(ax, j);
    if (j > 3) { neg = !neg; j -= 4; }
    const float z = r * r;
    const float y = (j == 1 || j == 2) ? rt_cos_kernel(z) : rt_sin_kernel(r, z);
    return neg ? -y : y;
}
RT_FORCEINLINE float rt_cosf(float x)
{
    if (x != x) return x;
    const float ax = x < 0.0f ? -x : x;
    if (ax > 8192.0f) return 0.0f;
    int j; const float r = rt_trig_reduce(ax, j);
    bool neg = false;
    if (j > 3) { neg = !neg; j -= 4; }
    if (j > 1) neg = !neg;
    const float z = r * r;
    const float y = (j == 1 || j == 2) ? rt_sin_kernel(r, z) : rt_cos_kernel(z);
    return neg ? -y : y;
}
RT_FORCEINLINE float rt_asinf(float x)
{
    const bool neg = x < 0.0f;
    const float a = neg ? -x : x;
    if (a > 1.0f) return (x - x) / (x - x);
    if (a < 1.0e-4f) return x;
    float z, w; bool flag = false;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); w = sqrtf(z); flag = true; }
    else { w = a; z = w * w; }
    float p = 4.2163199048E-2f * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    p = p * z;
    p = p * w;
    p = p + w;
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return neg ? -p : p;
}
RT_FORCEINLINE float rt_acosf(float x)
{
    if (x != x) return x;
    if (x < -1.0f || x > 1.0f) return (x - x) / (x - x);   // NaN: a non-unit sphere normal (w-lane pollution) gets here; f2i_gpu then reads texel row 0
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * rt_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * rt_asinf(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - rt_asinf(x);
}
RT_FORCEINLINE float rt_atanf(float x)
{
    const bool neg = x < 0.0f;
    float a = neg ? -x : x, y;
    if (a > 2.414213562373095f) { y = 1.5707963267948966192f; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = 0.7853981633974483096f; a = (a - 1.0f) / (a + 1.0f); }
    else y = 0.0f;
    const float z = a * a;
    float p = 8.05374449538e-2f * z - 1.38776856032E-1f;
    p = p * z + 1.99777106478E-1f;
    p = p * z - 3.33329491539E-1f;
    p = p * z;
    p = p * a;
    p = p + a;
    y = y + p;
    return neg ? -y : y;
}
RT_FORCEINLINE float rt_atan2f(float y, float x)
{
    if (x != x || y != y) return x + y;
    if (x == 0.0f) {
        if (y == 0.0f) return signbit(x) ? copysignf(3.14159265358979323846f, y) : y;
        return y > 0.0f ? 1.5707963267948966192f : -1.5707963267948966192f;
    }
    float z = rt_atanf(y / x);
    if (x < 0.0f) z = signbit(y) ? z - 3.14159265358979323846f : z + 3.14159265358979323846f;
    return z;
}
#endif

// ------------------------------------------------------------------ RNG (util.cl:50-59)
RT_FORCEINLINE uint32_t rng_next(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
RT_FORCEINLINE float rnd_float(uint32_t& s) { return (float)rng_next(s) * 2.3283064365387e-10f; }
RT_FORCEINLINE float rnd_abs(uint32_t& s) { return fabsf(rnd_float(s)); }
RT_FORCEINLINE float4 rnd_float3(uint32_t& s) { float x = rnd_float(s), y = rnd_float(s), z = rnd_float(s); return mk4(x, y, z, 0.0f); }

// ------------------------------------------------------------------ traversal ray (registers)
// Inside an instance the reference's transformRay() zeroes the w lanes (tlas.cl:5-7), and
// AABB tests only read xyz, so traversal carries xyz only and w == 0 in primitive tests.
struct TRay {
    float ox, oy, oz, dx, dy, dz, rx, ry, rz;
    float t; int prim; float u, v;
};
struct WorkCtr { uint32_t tlas, inst, node, prim, nodeIss = 0, leafIss = 0, evNode = 0, evPrim = 0; };   // nodeIss / leafIss: wave-level issues of the node path / the triangle path by the event loops (x 64 lanes = the slots those events had); evNode / evPrim: the events they carried
static constexpr int kCtrCols = 9;

RT_FORCEINLINE float slab(const TRay& r, float4 bmin, float4 bmax) // bvh.cl:3-12
{
    float tx1 = (bmin.x - r.ox) * r.rx, tx2 = (bmax.x - r.ox) * r.rx;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (bmin.y - r.oy) * r.ry, ty2 = (bmax.y - r.oy) * r.ry;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (bmin.z - r.oz) * r.rz, tz2 = (bmax.z - r.oz) * r.rz;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    return (tmax >= tmin && tmin < r.t && tmax > 0) ? tmin : kFar;
}

// Any-hit (connect) form of the same test: `hit` is exactly the reference's "visit this child" condition (slab() < t_light; in an
// occlusion traversal ray.t stays t_light until the first accepted primitive ends it, bvh.cl:25,40), and the exit distance comes
// for free.  Whether a shadow ray is occluded does not depend on the ORDER in which the children that pass this test are visited
// (a node is reached iff all its ancestors pass, whatever the order; the first accepted primitive returns), so connect is free to
// pick its own: the child the ray leaves LATER first - nearer the light, where most occluders of this kind of scene sit - finds
// an occluder after 27.6 instead of 34.9 node visits per shadow ray on the bench scene (tools/lab/anyhit_lab.c; 85 % of its
// shadow rays are occluded), with the accumulator unchanged bit for bit.
RT_FORCEINLINE bool slab_any(const TRay& r, float4 bmin, float4 bmax, float& exitT)
{
    float tx1 = (bmin.x - r.ox) * r.rx, tx2 = (bmax.x - r.ox) * r.rx;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (bmin.y - r.oy) * r.ry, ty2 = (bmax.y - r.oy) * r.ry;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (bmin.z - r.oz) * r.rz, tz2 = (bmax.z - r.oz) * r.rz;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    exitT = tmax;
    return tmax >= tmin && tmin < r.t && tmax > 0;
}

RT_FORCEINLINE void test_prim(const DevScene& sc, int idx, TRay& r) // primitives.cl:11-89
{
    const RtPrimitive* p = sc.prims + idx;
    const int type = p->objType;
    const float4 O = mk4(r.ox, r.oy, r.oz, 0.0f), D = mk4(r.dx, r.dy, r.dz, 0.0f);
    if (type == RT_PRIM_TRIANGLE) {
        const float4 v0 = ld4(p->obj.triangle.v0), v1 = ld4(p->obj.triangle.v1), v2 = ld4(p->obj.triangle.v2);
        float4 v0v1 = sub4(v1, v0), v0v2 = sub4(v2, v0);
        float4 pvec = cross4(D, v0v2);
        float det = dot4(v0v1, pvec);
        if (fabsf(det) < 1e-8f) return;
        float invDet = 1.0f / det;
        float4 tvec = sub4(O, v0);
        float u = dot4(tvec, pvec) * invDet;
        if (u < 0 || u > 1) return;
        float4 qvec = cross4(tvec, v0v1);
        float v = dot4(D, qvec) * invDet;
        if (v < 0 || u + v > 1) return;
        float t = dot4(v0v2, qvec) * invDet;
        if (t > r.t || t < 0) return;
        r.t = t; r.prim = idx; r.u = u; r.v = v;
    } else if (type == RT_PRIM_SPHERE) {
        const float4 pos = ld4(p->obj.sphere.pos);
        const float r2 = p->obj.sphere.r2;
        float4 oc = sub4(O, pos);
        float b = dot4(oc, D);
        float c = dot4(oc, oc) - r2;
        float d = b * b - c;
        if (d <= 0) return;
        d = sqrtf(d);
        float t = -b - d;
        if (t < r.t && t > 0) { r.t = t; r.prim = idx; return; }
        t = d - b;
        if (t < r.t && t > 0) { r.t = t; r.prim = idx; return; }
    } else if (type == RT_PRIM_PLANE) {
        const float4 N = ld4(p->obj.plane.N);
        float t = -(dot4(O, N) + p->obj.plane.d) / dot4(D, N);
        if (t > r.t || t < 0) return;
        r.t = t; r.prim = idx;
        float4 uAxis = mk4(N.y, N.z, -N.x, 0.0f);
        float4 vAxis = cross4(uAxis, N);
        float4 I = add4(O, muls(D, r.t));
        r.u = dot4(I, uAxis); r.v = dot4(I, vAxis);
    }
}

// Stack column of this lane in LDS: entry e lives at stk[e * kBlock + tid].
#define STK(e) stk[(e) * kBlock + threadIdx.x]

// BVH2 traversal, bvh.cl:13-54.  Returns `steps` (or -1 when OCC and a hit closer than
// t_light exists).  `root` is the instance's bvhIdx; child ids are absolute.
template <bool OCC>
RT_FORCEINLINE int traverse_bvh2(const DevScene& sc, TRay& r, uint32_t root, uint32_t* stk, WorkCtr& wc)
{
    const RtBVHNode2* nodes = sc.bvh2;
    uint32_t node = root, sp = 0;
    int steps = 0;
    const float tLight = r.t;
    for (;;) {
        const uint2 fc = *reinterpret_cast<const uint2*>(&nodes[node].first);
        if (fc.y > 0) {
            for (uint32_t i = 0; i < fc.y; i++) {
                int index = (int)sc.primIdx[fc.x + i];
                wc.prim++;
                test_prim(sc, index, r);
                if (OCC && r.t < tLight) return -1;
            }
            if (sp == 0) break;
            node = STK(--sp);
            continue;
        }
        wc.node++;
        uint32_t c1 = fc.x, c2 = fc.x + 1;
        if (OCC) {   // any-hit: own visit order (see slab_any)
            float x1, x2;
            const bool h1 = slab_any(r, ld4(nodes[c1].aabbMin), ld4(nodes[c1].aabbMax), x1), h2 = slab_any(r, ld4(nodes[c2].aabbMin), ld4(nodes[c2].aabbMax), x2);
            if (h1 && h2) { const bool firstIs2 = x2 > x1; node = firstIs2 ? c2 : c1; STK(sp) = firstIs2 ? c1 : c2; sp++; }
            else if (h1 || h2) node = h1 ? c1 : c2;
            else { if (sp == 0) break; node = STK(--sp); }
            continue;
        }
        float d1 = slab(r, ld4(nodes[c1].aabbMin), ld4(nodes[c1].aabbMax));
        float d2 = slab(r, ld4(nodes[c2].aabbMin), ld4(nodes[c2].aabbMax));
        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t c = c1; c1 = c2; c2 = c; }
        if (d1 >= tLight) {
            if (sp == 0) break;
            node = STK(--sp);
        } else {
            steps++;
            node = c1;
            if (d2 < tLight) { STK(sp) = c2; sp++; steps++; }
        }
    }
    return steps;
}

// ---- layout 1: the same BVH2, re-laid-out at upload for one dependent fetch per step ---------------
// The reference node array is uploaded unchanged; from it rt_upload_scene derives
//   pairs[i]   (64 B, one cache-line-aligned fetch) for every interior node i:
//              q0 = (c1.min.xyz, c1.max.x) q1 = (c1.max.yz, c2.min.xy) q2 = (c2.min.z, c2.max.xyz)
//              q3 = (entry(c1), entry(c2), -, -) with c1 = nodes[i].first, c2 = c1 + 1
//   entry(n)   = n                                   interior node id
//              = 0x80000000 | count << 24 | first    leaf (count <= 127, first < 2^24)
//   triRecs[s] (48 B) for every primIdx slot s: v0, the edges v1 - v0 and v2 - v0 (xyz) and the primitive id, in leaf
//              order, so a leaf's triangles are contiguous and the primIdx indirection disappears.
// Visit order, slab arithmetic, tie rules and `steps` are those of traverse_bvh2 (bvh.cl:13-54): results
// are bit-identical; what changes is that a step costs one dependent fetch instead of two (node header,
// then child boxes) and a triangle test one instead of two (index, then 128-byte Primitive).
static constexpr uint32_t kLeafBit = 0x80000000u;
RT_FORCEINLINE void test_tri_packed(const DevScene& sc, uint32_t slot, TRay& r)
{
    const float4* rec = sc.triRecs + (size_t)slot * 3;
    const float4 a = rec[0], b = rec[1], c = rec[2];
    // Keep the three loads of the record together: left alone, the compiler fetches the flag word first, waits, branches on it and only
    // then fetches the vertices - two dependent round trips per triangle instead of one.
    asm volatile("" : : "v"(a.x), "v"(b.x), "v"(c.x));
    const int idx = __float_as_int(c.y);
    if (__float_as_int(c.z) != 0) { test_prim(sc, idx, r); return; } // not a plain triangle: reference-layout test
    const float4 O = mk4(r.ox, r.oy, r.oz, 0.0f), D = mk4(r.dx, r.dy, r.dz, 0.0f);
    // the record holds v0 and the two edges v1 - v0, v2 - v0 (the reference's first two operations, primitives.cl:49-50, done once at
    // upload with the same IEEE subtraction)
    const float4 v0 = mk4(a.x, a.y, a.z, 0.0f), v0v1 = mk4(a.w, b.x, b.y, 0.0f), v0v2 = mk4(b.z, b.w, c.x, 0.0f);
    float4 pvec = cross4(D, v0v2);
    float det = dot4(v0v1, pvec);
    if (fabsf(det) < 1e-8f) return;
    float invDet = 1.0f / det;
    float4 tvec = sub4(O, v0);
    float u = dot4(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return;
    float4 qvec = cross4(tvec, v0v1);
    float v = dot4(D, qvec) * invDet;
    if (v < 0 || u + v > 1) return;
    float t = dot4(v0v2, qvec) * invDet;
    if (t > r.t || t < 0) return;
    r.t = t; r.prim = idx; r.u = u; r.v = v;
}
template <bool OCC>
RT_FORCEINLINE int traverse_bvh2_packed(const DevScene& sc, TRay& r, uint32_t rootEntry, uint32_t* stk, WorkCtr& wc)
{
    uint32_t cur = rootEntry, sp = 0;
    int steps = 0;
    const float tLight = r.t;
    for (;;) {
        if (cur & kLeafBit) {
            // ONE triangle of the leaf per iteration (the rest of the leaf stays in `cur`): a wave of this loop runs the leaf body whenever
            // any of its lanes is on a leaf, and with the whole leaf in that body every such iteration cost the lanes on a box pair two or
            // three dependent triangle fetches (config 3's late bounces 195 / 147 / 127 -> 184 / 138 / 116 us, EXPERIMENTS.md (57))
            const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
            wc.prim++;
            test_tri_packed(sc, first, r);
            if (OCC && r.t < tLight) return -1;
            if (count > 1) { cur = kLeafBit | ((count - 1) << 24) | (first + 1); continue; }
            if (sp == 0) break;
            cur = STK(--sp);
            continue;
        }
        wc.node++;
        const float4* p = sc.pairs + (size_t)cur * 4;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        uint32_t e1 = __float_as_uint(q3.x), e2 = __float_as_uint(q3.y);
        if (OCC) {   // any-hit: own visit order (see slab_any)
            float x1, x2;
            const bool h1 = slab_any(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f), x1);
            const bool h2 = slab_any(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f), x2);
            if (h1 && h2) { const bool firstIs2 = x2 > x1; cur = firstIs2 ? e2 : e1; STK(sp) = firstIs2 ? e1 : e2; sp++; }
            else if (h1 || h2) cur = h1 ? e1 : e2;
            else { if (sp == 0) break; cur = STK(--sp); }
            continue;
        }
        float d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
        float d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t e = e1; e1 = e2; e2 = e; }
        if (d1 >= tLight) {
            if (sp == 0) break;
            cur = STK(--sp);
        } else {
            steps++;
            cur = e1;
            if (d2 < tLight) { STK(sp) = e2; sp++; steps++; }
        }
    }
    return steps;
}

// Primary rays (an 8x8 pixel tile per wave): for the first ~10 levels every lane of the wave sits on the SAME node.  While that holds
// the 64-byte pair record is fetched once through the scalar cache (constant address space + a wave-uniform index = s_load) instead of
// 64 times through the vector memory pipeline; arithmetic, visit order, `steps` and counters are those of traverse_bvh2_packed.
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) fvec4* ConstF4;
RT_FORCEINLINE int traverse_bvh2_packed_coherent(const DevScene& sc, TRay& r, uint32_t rootEntry, uint32_t* stk, WorkCtr& wc)
{
    uint32_t cur = rootEntry, sp = 0;
    int steps = 0;
    const float tLight = r.t;
    const ConstF4 cpairs = (ConstF4)(uintptr_t)sc.pairs;
    for (;;) {
        if (cur & kLeafBit) {
            const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
            wc.prim++; test_tri_packed(sc, first, r);        // one triangle per iteration, as in traverse_bvh2_packed
            if (count > 1) { cur = kLeafBit | ((count - 1) << 24) | (first + 1); continue; }
            if (sp == 0) break;
            cur = STK(--sp);
            continue;
        }
        wc.node++;
        float d1, d2;
        uint32_t e1, e2;
        const uint32_t ucur = __builtin_amdgcn_readfirstlane(cur);
        if (__ballot(cur != ucur) == 0ull) {   // every lane that is on an interior node right now is on this one
            // the slab tests are written out in this branch too, so that they read the record from the scalar registers it was loaded
            // into (joined with the other branch first, the record would be copied into 14 vector registers per lane and step)
            const ConstF4 p = cpairs + (size_t)ucur * 4;
            const fvec4 a = p[0], b = p[1], c = p[2], d = p[3];
            d1 = slab(r, mk4(a.x, a.y, a.z, 0.0f), mk4(a.w, b.x, b.y, 0.0f));
            d2 = slab(r, mk4(b.z, b.w, c.x, 0.0f), mk4(c.y, c.z, c.w, 0.0f));
            e1 = __float_as_uint(d.x); e2 = __float_as_uint(d.y);
        } else {
            const float4* p = sc.pairs + (size_t)cur * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
            d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
            e1 = __float_as_uint(q3.x); e2 = __float_as_uint(q3.y);
        }
        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t e = e1; e1 = e2; e2 = e; }
        if (d1 >= tLight) {
            if (sp == 0) break;
            cur = STK(--sp);
        } else {
            steps++;
            cur = e1;
            if (d2 < tLight) { STK(sp) = e2; sp++; steps++; }
        }
    }
    return steps;
}

// BVH4 traversal, bvh.cl:55-96 (children visited in slot order, all four distances taken
// at node entry).
template <bool OCC>
RT_FORCEINLINE int traverse_bvh4(const DevScene& sc, TRay& r, uint32_t root, uint32_t* stk, WorkCtr& wc)
{
    const RtBVHNode4* nodes = sc.bvh4;
    uint32_t node = root, sp = 0;
    int steps = 0;
    const float tLight = r.t;
    for (;;) {
        steps++; wc.node++;
        const RtBVHNode4* n = nodes + node;
        const int4 first = *reinterpret_cast<const int4*>(n->first);
        const int4 count = *reinterpret_cast<const int4*>(n->count);
        const int f[4] = { first.x, first.y, first.z, first.w };
        const int c[4] = { count.x, count.y, count.z, count.w };
        float dist[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            dist[k] = f[k] != RT_INVALID ? slab(r, ld4(n->aabbMin[k]), ld4(n->aabbMax[k])) : kFar;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (f[k] == RT_INVALID) continue;
            if (dist[k] >= tLight) continue;
            if (c[k] > 0) {
                for (uint32_t j = 0; j < (uint32_t)c[k]; j++) {
                    int index = (int)sc.primIdx[f[k] + j];
                    wc.prim++;
                    test_prim(sc, index, r);
                    if (OCC && r.t < tLight) return -1;
                }
            } else {
                STK(sp) = (uint32_t)f[k]; sp++;
            }
        }
        if (sp == 0) break;
        node = STK(--sp);
    }
    return steps;
}

// Layout 1 of the BVH4 (derived at upload from the unchanged BVHNode4 array): quads[i] = 128 B,
//   floats 0..23 = {min.xyz, max.xyz} of child slots 0..3, q6 = entry(slot 0..3) (0xffffffff = unused slot, else as
//   for the BVH2 layout: interior node id, or leaf bit | count << 24 | first).  7 sixteen-byte loads from two 64-B lines
// instead of 10 from three, and triangles come from the leaf-ordered triRecs.  Order of evaluation as bvh.cl:55-96.
static constexpr uint32_t kNoChild = 0xffffffffu;
template <bool OCC>
RT_FORCEINLINE int traverse_bvh4_packed(const DevScene& sc, TRay& r, uint32_t root, uint32_t* stk, WorkCtr& wc)
{
    uint32_t node = root, sp = 0;
    int steps = 0;
    const float tLight = r.t;
    for (;;) {
        steps++; wc.node++;
        const float4* p = sc.quads + (size_t)node * 8;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5], q6 = p[6];
        asm volatile("" : : "v"(q0.x), "v"(q1.x), "v"(q2.x), "v"(q3.x), "v"(q4.x), "v"(q5.x), "v"(q6.x));   // one round trip (see test_tri_packed)
        const uint32_t e[4] = { __float_as_uint(q6.x), __float_as_uint(q6.y), __float_as_uint(q6.z), __float_as_uint(q6.w) };
        float dist[4];
        dist[0] = e[0] != kNoChild ? slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f)) : kFar;
        dist[1] = e[1] != kNoChild ? slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f)) : kFar;
        dist[2] = e[2] != kNoChild ? slab(r, mk4(q3.x, q3.y, q3.z, 0.0f), mk4(q3.w, q4.x, q4.y, 0.0f)) : kFar;
        dist[3] = e[3] != kNoChild ? slab(r, mk4(q4.z, q4.w, q5.x, 0.0f), mk4(q5.y, q5.z, q5.w, 0.0f)) : kFar;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e[k] == kNoChild) continue;
            if (dist[k] >= tLight) continue;
            if (e[k] & kLeafBit) {
                const uint32_t first = e[k] & 0x00ffffffu, count = (e[k] >> 24) & 0x7fu;
                for (uint32_t j = 0; j < count; j++) {
                    wc.prim++;
                    test_tri_packed(sc, first + j, r);
                    if (OCC && r.t < tLight) return -1;
                }
            } else {
                STK(sp) = e[k]; sp++;
            }
        }
        if (sp == 0) break;
        node = STK(--sp);
    }
    return steps;
}

// instanceIntersect, tlas.cl:9-26 with transformRay :3-8 and util.cl:61-87.  The instance record (rows 0..2 of invT + the encoded BLAS
// root) is one 64-byte fetch; round 1 read BVHInstance.invT and rootEntry[] separately.
template <int ACCEL, int LAYOUT, bool OCC>
RT_FORCEINLINE int traverse_instance(const DevScene& sc, TRay& r, uint32_t instIdx, uint32_t* stk, WorkCtr& wc)
{
    const float4* ir = sc.instRecs + (size_t)instIdx * 4;
    const float4 t0 = ir[0], t1 = ir[1], t2 = ir[2], t3 = ir[3];
    const float bx = r.ox, by = r.oy, bz = r.oz, bdx = r.dx, bdy = r.dy, bdz = r.dz, brx = r.rx, bry = r.ry, brz = r.rz;
    const float4 Dv = mk4(bdx, bdy, bdz, 0.0f), Ov = mk4(bx, by, bz, 0.0f);
    r.dx = dot3(mk4(t0.x, t0.y, t0.z, 0), Dv); r.dy = dot3(mk4(t1.x, t1.y, t1.z, 0), Dv); r.dz = dot3(mk4(t2.x, t2.y, t2.z, 0), Dv);
    r.ox = dot3(mk4(t0.x, t0.y, t0.z, 0), Ov) + t0.w; r.oy = dot3(mk4(t1.x, t1.y, t1.z, 0), Ov) + t1.w;
    r.oz = dot3(mk4(t2.x, t2.y, t2.z, 0), Ov) + t2.w;
    r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
    wc.inst++;
    const uint32_t rootEnc = __float_as_uint(t3.x), bvhIdx = __float_as_uint(t3.y);
    int steps;
    if (ACCEL == RT_ACCEL_BVH4) steps = LAYOUT == 1 ? traverse_bvh4_packed<OCC>(sc, r, rootEnc, stk, wc) : traverse_bvh4<OCC>(sc, r, bvhIdx, stk, wc);
    else if (LAYOUT == 1) steps = traverse_bvh2_packed<OCC>(sc, r, rootEnc, stk, wc);
    else steps = traverse_bvh2<OCC>(sc, r, bvhIdx, stk, wc);
    r.ox = bx; r.oy = by; r.oz = bz; r.dx = bdx; r.dy = bdy; r.dz = bdz; r.rx = brx; r.ry = bry; r.rz = brz;
    return steps;
}

// intersectTLAS, tlas.cl:28-77, over the derived TLAS records: an interior visit is ONE fetch (both child boxes and both encoded
// children; the reference array costs the node's leftRight word, then the two child nodes), a leaf costs none (its BLAS index sits in
// the parent's record).  Visit order, pruning and counters are the reference's.  The TLAS stack (<= 32 entries) is a private array;
// with a single BLAS the root is a leaf and it is never touched.
template <int ACCEL, int LAYOUT, bool OCC>
RT_FORCEINLINE int traverse_tlas(const DevScene& sc, TRay& r, uint32_t* stk, WorkCtr& wc)
{
    uint16_t tstack[RT_TLAS_STACK];   // 16-bit entries as in the reference (bit 15 = leaf; <= 256 instances, <= 512 nodes)
    uint32_t cur = sc.tlasRoot, sp = 0;
    auto pack16 = [](uint32_t c) { return (uint16_t)((c >> 16) | (c & 0x7fffu)); };
    auto unpack16 = [](uint32_t v) { return ((v & 0x8000u) << 16) | (v & 0x7fffu); };
    int steps = 0;
    const float tLight = r.t;
    for (;;) {
        if (cur & kLeafBit) {
            int value = traverse_instance<ACCEL, LAYOUT, OCC>(sc, r, cur & 0x7fffffffu, stk, wc);
            if (OCC && value == -1) return -1;
            steps += value;
            if (sp == 0) break;
            cur = unpack16(tstack[--sp]);
            continue;
        }
        wc.tlas++;
        const float4* p = sc.tlasPairs + (size_t)cur * 4;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        float d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
        float d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
        uint32_t c1 = __float_as_uint(q3.x), c2 = __float_as_uint(q3.y);
        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t c = c1; c1 = c2; c2 = c; }
        if (d1 >= tLight) {
            if (sp == 0) break;
            cur = unpack16(tstack[--sp]);
        } else {
            cur = c1;
            if (d2 < tLight) tstack[sp++] = pack16(c2);
        }
    }
    return steps;
}

// Sum a lane's work counters over the workgroup and add them to this block's row of the
// per-block partial table (no atomics: one row per blockIdx, launches are stream-ordered).
RT_FORCEINLINE void flush_counters(unsigned long long* table, uint32_t rays, const WorkCtr& wc, uint32_t* red /* >= kCtrCols*4 words LDS */)
{
    uint32_t v[kCtrCols] = { rays, wc.tlas, wc.inst, wc.node, wc.prim, wc.nodeIss, wc.leafIss, wc.evNode, wc.evPrim };
#pragma unroll
    for (int k = 0; k < kCtrCols; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads(); // LDS stack columns are dead past this point
    if (lane == 0) for (int k = 0; k < kCtrCols; k++) red[wave * kCtrCols + k] = v[k];
    __syncthreads();
    if (threadIdx.x < kCtrCols) {
        unsigned long long s = 0;
        for (int w = 0; w < kBlock / 64; w++) s += red[w * kCtrCols + threadIdx.x];
        table[(size_t)blockIdx.x * kCtrCols + threadIdx.x] += s;
    }
}

// ------------------------------------------------------------------ k_reset / k_begin
__global__ void k_reset(float4* accum, int32_t first, int32_t n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) accum[first + i] = splat(0.0f);
}
RT_FORCEINLINE void begin_frame(const DevQueues& q) // renderer.cpp:66-69 (one workgroup)
{
    if (threadIdx.x == 0) { q.nRays[0] = q.nPix; q.nShadow[0] = 0; }
    if (threadIdx.x < kCursorWords) q.cursor[threadIdx.x] = 0;
    for (int k = threadIdx.x; k < (RT_MAX_BOUNCES + 1) * kTicketClasses; k += blockDim.x) q.shadeTicket[(size_t)k * kTicketStride] = 0;
}
__global__ void k_begin_frame(DevQueues q) { begin_frame(q); }   // stage-level API; rt_render folds it into k_generate

// ------------------------------------------------------------------ k_generate
RT_FORCEINLINE void primary_ray(const RtCamera& cam, int x, int y, int W, int H, int aa, uint32_t& seed, float4& O, float4& D)
{
    if (cam.type == RT_CAM_PROJECTION) { // camera.cl:9-24
        float u = (float)x * (1.0f / (float)W);
        float v = (float)y * (1.0f / (float)H);
        if (aa) { u += rnd_float(seed) / (float)W; v += rnd_float(seed) / (float)H; }
        float4 P = add4(add4(ld4(cam.topLeft), muls(ld4(cam.horizontal), u)), muls(ld4(cam.vertical), v));
        float4 dir = normalize4(sub4(P, ld4(cam.origin)));
        float4 focalPoint = add4(ld4(cam.origin), muls(dir, cam.focalLength));
        O = add4(ld4(cam.origin), muls(sub4(rnd_float3(seed), splat(0.5f)), cam.aperture));
        D = normalize4(sub4(focalPoint, O));
    } else { // camera.cl:25-44
        float u = ((float)x - (float)W * .5f) * (2.f / (float)W);
        float v = ((float)y - (float)H * .5f) * (2.f / (float)H);
        if (aa) { u += rnd_float(seed) / (float)W; v += rnd_float(seed) / (float)H; }
        float r2 = u * u + v * v;
        if (r2 > 1.0f) { O = splat(0.0f); D = splat(0.0f); return; }
        float rr = sqrtf(r2);
        float psi = rr * cam.fov * kPi / 180.0f;
        float sinPsi = rt_sinf(psi), cosPsi = rt_cosf(psi);
        float sinAlpha = u / rr, cosAlpha = v / rr;
        D = sub4(add4(muls(ld4(cam.up), sinPsi * cosAlpha), muls(ld4(cam.right), sinPsi * sinAlpha)), muls(ld4(cam.forward), cosPsi));
        O = ld4(cam.origin);
    }
}
__global__ __launch_bounds__(kBlock) void k_generate(DevQueues q, RtCamera cam, int aa, int beginFrame)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (beginFrame && blockIdx.x == 0) begin_frame(q);   // the counters it resets are read by later launches only
    if (threadIdx.x == 0) { q.tile[0][1 + blockIdx.x] = 0ull; if ((blockIdx.x & 63) == 0) { q.super[0][blockIdx.x >> 6] = 0ull; q.supAcc[0][blockIdx.x >> 6] = 0ull; } }   // arm shade(0)'s scan
    if (i >= q.nPix) return;
    const int idx = q.firstPixel + i;
    uint32_t seed = q.seeds[i];
    float4 O, D;
    primary_ray(cam, idx % q.width, idx / q.width, q.width, q.height, aa, seed, O, D);
    q.seeds[i] = seed;
    stnt4(&q.O[0][i], O); stnt4(&q.D[0][i], D); stnt4(&q.inten[0][i], splat(1.0f));
    stnt2(&q.meta[0][i], make_uint2((uint32_t)idx, kMetaLastSpec)); // bounces 0, inside 0, lastSpecular 1
}

// ------------------------------------------------------------------ k_extend (variant 0: one ray per lane)
template <int ACCEL, int LAYOUT>
__global__ __launch_bounds__(kBlock) void k_extend(DevScene sc, DevQueues q, int bounce, int renderBVH)
{
    extern __shared__ uint32_t stk[];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int n = q.nRays[bounce];
    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    if (i < n) {
        const float4 O = q.O[bounce & 1][i], D = q.D[bounce & 1][i];
        TRay r;
        r.ox = O.x; r.oy = O.y; r.oz = O.z; r.dx = D.x; r.dy = D.y; r.dz = D.z;
        r.rx = 1.0f / D.x; r.ry = 1.0f / D.y; r.rz = 1.0f / D.z;
        r.t = kFar; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
        int steps = traverse_tlas<ACCEL, LAYOUT, false>(sc, r, stk, wc);
        rays = 1;
        q.hit[i] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
        if (q.steps) q.steps[i] = steps;
        if (renderBVH) q.accum[q.firstPixel + i] = splat((float)(uint32_t)steps / 255.f); // wavefront.cl:67
    }
    flush_counters(q.ctrExtend, rays, wc, stk);
}

// ------------------------------------------------------------------ k_trace_persist (layout 1, single BLAS)
// Persistent wavefronts (north star; Aila & Laine style): a fixed grid of resident waves pulls rays from
// the bounce's queue.  A wave dequeues a chunk of rays with ONE atomic (lane 0, broadcast), hands them to
// its idle lanes by ballot + prefix count, and keeps traversing; whenever at least kRefill lanes have
// finished their ray it tops the idle lanes up again, so SIMD lanes stay busy although traversal lengths
// differ by an order of magnitude between rays.  Each lane advances its own ray by one event per
// iteration (one interior visit, or one triangle test) with the exact per-ray visit order of
// traverse_bvh2_packed, so hits, `steps` and the work counters are bit-identical to the one-ray-per-lane
// kernels; only the lane<->ray assignment differs.  Every wave reaches the exit condition (queue empty and
// all its lanes idle), so the grid always drains.
// Work distribution: the first chunk of every wave is static (chunk id = global wave id, no atomic, so the
// launch does not start with thousands of waves hammering one counter); further chunks are dequeued.
#ifdef RT355_TAIL_PROBE
// lab build only (tools/lab/tail_probe.sh): when does the queue of a persistent launch run dry, and when do its waves exit?  One record per
// launch slot (extend of bounce b: b, connect: 8) and wave, plain stores to distinct addresses (nothing shared, so the probe does not
// disturb what it measures): wall_clock64 at start, when the wave found the queue dry (0: never), at exit.  Read by rt_lab_tail_probe.
static constexpr int kTpWaves = 8192;
__device__ unsigned long long g_tp[9][kTpWaves][4];
#endif
struct PersistTune { int chunk, refill, inner, leafK, fixedChunks, flat = 0, backup = 0, xcdRays = 0, xcdFirst = 0, thin = 0; };   // thin: a queue of at most `thin` rays per participating wave is spread evenly over the waves (few lanes of each) instead of filling the first waves   // xcdRays: a sparse queue's rays are kept on as few XCDs as hold them at this many rays each (one-ray-per-lane branches; 0 = spread over all eight)   // backup: k_trace_persist_tlas keeps the world ray in LDS across an instance visit (10 words per lane behind the stack column) instead of fetching it back from the queue   // flat: k_trace_persist_tlas runs every queue through its one-ray-per-lane branch, 64 rays per wave and round (short traversals: config 5's open scene)   // rays per dequeue, idle lanes that trigger a top-up, events between checks, lanes on a leaf that trigger the triangle path, chunks dealt round-robin instead of dequeued

// A sparse queue on few XCDs: workgroup ids go to the eight XCDs round-robin and every XCD has its own L2, so the few thousand rays of a late
// bounce spread over all of them fetch every node record from the Infinity Cache once PER XCD.  With `xcdRays` > 0 only the workgroups of
// the first nx XCDs take rays (nx = as many as hold the queue at xcdRays rays each), in the order of their rank among those: returns the
// wave's rank and sets `waves` to the number of participating waves, or -1 for a wave that sits this launch out.
RT_FORCEINLINE int xcd_pack(int n, int xcdRays, int xcdFirst, int& waves)
{
    const int wpb = kBlock / 64, wave = threadIdx.x >> 6;
    waves = gridDim.x * wpb;
    if (xcdRays <= 0 || (gridDim.x & 7u) != 0u) return blockIdx.x * wpb + wave;
    const int perXcd = (int)(gridDim.x >> 3) * kBlock;
    int nx = min(8, max(1, (n + xcdRays - 1) / xcdRays));
    while (nx < 8 && (long long)nx * perXcd < (long long)n) nx++;
    if (nx == 8) return blockIdx.x * wpb + wave;
    const int xcd = (int)(blockIdx.x + 8u - (uint32_t)xcdFirst) & 7;   // contexts that share the GPU start on different XCDs
    waves = (int)(gridDim.x >> 3) * nx * wpb;
    if (xcd >= nx) return -1;
    return ((int)(blockIdx.x >> 3) * nx + xcd) * wpb + wave;
}

// The queue slot of this lane in a launch that is not shorter than its queue (the one-ray-per-lane loops), or n for a lane without one.
// Sparse queues are (1) kept on few XCDs (xcd_pack) and (2) THINNED: in the one-ray-per-lane loop a wave pays for the union of its lanes'
// states every iteration - a node fetch AND a leaf's triangle fetches, one after the other - so a ray advances at the pace of its 63
// wave-mates; a queue of at most `thin` rays per participating wave therefore takes a few lanes of EVERY wave instead of all lanes of the
// first ones (config 2's late launches 93 / 88 / 72 / 65 / 48 -> 83 / 60 / 38 / 33 / 29 us, EXPERIMENTS.md (55)).
RT_FORCEINLINE int sparse_slot(int n, const PersistTune& t, int waveId, int lane)
{
    if (t.xcdRays <= 0 && t.thin <= 0) return waveId * 64 + lane;
    int waves;
    const int w = xcd_pack(n, t.xcdRays, t.xcdFirst, waves);
    if (w < 0) return n;
    const int per = (n + waves - 1) / waves;      // (<= 64: the queue is not longer than the launch)
    if (t.thin > 0 && per <= t.thin) return lane < per ? w * per + lane : n;
    return w * 64 + lane;
}

// Short queue (late bounces, and bounce 0 when it is launched with one workgroup per 256 rays): every wave gets at most one 64-ray chunk
// and nothing is left to refill from, so run the plain one-ray-per-lane loop, which has less per-step overhead than the refill machine.
template <bool OCC, bool COH>
RT_FORCEINLINE void trace_short_queue(const DevScene& sc, const DevQueues& q, int b0, int qFirst, int n, int renderBVH, const float* T,
                                      uint32_t rootEntry, uint32_t* stk, int waveId, int lane, const PersistTune& tune)
{
    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    TRay r; r.t = 0; r.prim = -1; r.u = r.v = 0; r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.rx = r.ry = r.rz = 0;
    int idx = waveId * 64 + lane;
    if (!(b0 == 0 && n == q.nPix)) idx = sparse_slot(n, tune, waveId, lane);
    if (!OCC && b0 == 0 && ((q.width | (q.nPix / q.width)) & 7) == 0 && n == q.nPix && (long long)gridDim.x * kBlock >= (long long)q.nPix) {
        // primary rays: a wave takes an 8x8 pixel tile instead of a 64x1 strip (the queue of bounce 0 is the pixel grid - only when it IS
        // the whole grid and the launch has a wave for every tile; an injected shorter queue or a smaller grid keeps the strip mapping)
        const int tilesX = q.width >> 3, ty = waveId / tilesX, tx = waveId - ty * tilesX;
        idx = ((ty << 3) + (lane >> 3)) * q.width + (tx << 3) + (lane & 7);
    }
    if (idx < n) {
        float4 O, D; float tmax;
        if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tmax = a.w; }
        else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tmax = kFar; }
        const float4 Dv = mk4(D.x, D.y, D.z, 0.0f), Ov = mk4(O.x, O.y, O.z, 0.0f);
        r.dx = dot3(mk4(T[0], T[1], T[2], 0), Dv); r.dy = dot3(mk4(T[4], T[5], T[6], 0), Dv); r.dz = dot3(mk4(T[8], T[9], T[10], 0), Dv);
        r.ox = dot3(mk4(T[0], T[1], T[2], 0), Ov) + T[3]; r.oy = dot3(mk4(T[4], T[5], T[6], 0), Ov) + T[7];
        r.oz = dot3(mk4(T[8], T[9], T[10], 0), Ov) + T[11];
        r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
        r.t = tmax; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
        rays = 1; wc.inst = 1;
        const int st = COH ? traverse_bvh2_packed_coherent(sc, r, rootEntry, stk, wc) : traverse_bvh2_packed<OCC>(sc, r, rootEntry, stk, wc);
        if (OCC) { if (st == -1) q.sC[qFirst + idx] = splat(0.0f); }
        else {
            q.hit[idx] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
            if (q.steps) q.steps[idx] = st;
            if (renderBVH) q.accum[q.firstPixel + idx] = splat((float)(uint32_t)st / 255.f);
        }
    }
    flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
}


// STEPS: the per-ray `steps` value of the heat map (wavefront.cl:66-67) is kept only by the instantiation that has a reader for it
// (renderBVH or rt_debug_enable_steps); the work counters of a wave are kept in scalar registers (population counts of the masks the
// event loop forms anyway), not per lane.
template <bool OCC, bool COH = false, bool STEPS = false>
__global__ __launch_bounds__(kBlock) void k_trace_persist(DevScene sc, DevQueues q, int b0, int b1, int renderBVH, PersistTune tune)
{
    const int kChunk = tune.chunk, kRefill = tune.refill, kInner = tune.inner, kLeafK = tune.leafK;
    extern __shared__ uint32_t stk[];
    const int lane = threadIdx.x & 63;
    // queue window: extend -> rays [0, nRays[b0]); connect -> shadow rays [nShadow[b0], nShadow[b1+1])
    const int qFirst = OCC ? q.nShadow[b0] : 0;
    const int n = OCC ? q.nShadow[b1 + 1] - qFirst : q.nRays[b0];
    int32_t* cursor = q.cursor + (OCC ? (RT_MAX_BOUNCES + 2) + b0 : b0);
    const RtBVHInstance* inst = sc.blas + sc.tlas[0].BLASidx;
    const uint32_t rootEntry = sc.rootEntry[sc.tlas[0].BLASidx];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = inst->invT[k];

    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    uint32_t wRays = 0, wNode = 0, wPrim = 0, wNodeIss = 0, wLeafIss = 0;    // this wave's work (wave-uniform): events by kind, and how often each path was issued
    TRay r; r.t = 0; r.prim = -1; r.u = r.v = 0; r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.rx = r.ry = r.rz = 0;
    uint32_t cur = 0, sp = 0;
    int slot = -1, steps = 0;
    float tLight = 0;
    const int nWaves = gridDim.x * (kBlock / 64), waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
#ifdef RT355_TAIL_PROBE
    const unsigned long long tp0 = wall_clock64();
    unsigned long long tpDry = 0;
#define TAIL_PROBE_EXIT() if (lane == 0 && waveId < kTpWaves) { unsigned long long* w = g_tp[OCC ? 8 : b0][waveId]; w[0] = tp0; w[1] = tpDry; w[2] = wall_clock64(); w[3] = rays; }
#endif
    if (n <= nWaves * 64) {
        trace_short_queue<OCC, COH>(sc, q, b0, qFirst, n, renderBVH, T, rootEntry, stk, waveId, lane, tune);
#ifdef RT355_TAIL_PROBE
        TAIL_PROBE_EXIT()
#endif
        return;
    }
    int chunkNext = min(waveId * kChunk, n), chunkEnd = min(waveId * kChunk + kChunk, n);   // wave-uniform
    bool exhausted = false;                                                                  // wave-uniform
    int round = 0;

    for (;;) {
        const unsigned long long idleMask = __ballot(slot < 0);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 && exhausted && chunkNext >= chunkEnd) break;
        if (nIdle >= kRefill && !(exhausted && chunkNext >= chunkEnd)) {
            if (chunkNext >= chunkEnd) {           // dequeue a chunk for this wave
                int c = 0;
                if (tune.fixedChunks) { round++; c = round * nWaves * kChunk + waveId * kChunk; }   // chunks dealt round-robin: no atomic, no round trip (a context with the GPU to itself)
                else {                                                                           // dequeued: balances waves that other contexts' kernels slow down
                    if (lane == 0) c = atomicAdd(cursor, kChunk);
                    c = __shfl(c, 0, 64) + nWaves * kChunk;
                }
                chunkNext = c; chunkEnd = min(c + kChunk, n);
                if (c >= n) { exhausted = true; chunkNext = chunkEnd = 0; }
#ifdef RT355_TAIL_PROBE
                if (exhausted) tpDry = wall_clock64();
#endif
            }
            if (chunkNext < chunkEnd) {
                const int rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                const int idx = chunkNext + rank;
                if (slot < 0 && idx < chunkEnd) {
                    float4 O, D; float tmax;
                    if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tmax = a.w; }
                    else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tmax = kFar; }
                    // transformRay (tlas.cl:3-8) of the single instance, same arithmetic as traverse_instance
                    const float4 Dv = mk4(D.x, D.y, D.z, 0.0f), Ov = mk4(O.x, O.y, O.z, 0.0f);
                    r.dx = dot3(mk4(T[0], T[1], T[2], 0), Dv); r.dy = dot3(mk4(T[4], T[5], T[6], 0), Dv); r.dz = dot3(mk4(T[8], T[9], T[10], 0), Dv);
                    r.ox = dot3(mk4(T[0], T[1], T[2], 0), Ov) + T[3]; r.oy = dot3(mk4(T[4], T[5], T[6], 0), Ov) + T[7];
                    r.oz = dot3(mk4(T[8], T[9], T[10], 0), Ov) + T[11];
                    r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
                    r.t = tmax; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
                    tLight = tmax; cur = rootEntry; sp = 0; steps = 0; slot = idx;
                }
                wRays += (uint32_t)min(nIdle, chunkEnd - chunkNext);
                chunkNext = min(chunkNext + nIdle, chunkEnd);
            }
        }
#pragma unroll 1
        for (int it = 0; it < kInner; it++) {
            // One event per lane and iteration, but the wave issues only ONE of the two code paths: triangle tests
            // are held back until kLeafK lanes sit on a leaf (or no lane has a box test left), so neither path runs
            // with a handful of lanes while the rest of the wave waits (the per-ray event order is unchanged).
            const bool act = slot >= 0, atLeaf = act && (cur & kLeafBit) != 0u;
            const unsigned long long lm = __ballot(atLeaf), im = __ballot(act && !atLeaf);
            if ((lm | im) == 0ull) break;
            const bool doLeaf = im == 0ull || __popcll(lm) >= kLeafK;
            bool done = false, occluded = false;
            // (the path-issue counters behind RtCounters.*_issues live in the STEPS instantiation only: two scalar adds per iteration cost the
            // production kernel 2-4 % on bounces 1-3)
            if (!doLeaf) { wNode += (uint32_t)__popcll(im); if (STEPS) wNodeIss++; }
            if (doLeaf) {
                wPrim += (uint32_t)__popcll(lm); if (STEPS) wLeafIss++;
                if (atLeaf) {
                    const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
                    test_tri_packed(sc, first, r);
                    if (OCC && r.t < tLight) { done = true; occluded = true; }
                    else if (count > 1) cur = kLeafBit | ((count - 1) << 24) | (first + 1);
                    else if (sp == 0) done = true;
                    else cur = STK(--sp);
                }
            } else if (act && !atLeaf) {
                const float4* p = sc.pairs + (size_t)cur * 4;
                const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                uint32_t e1 = __float_as_uint(q3.x), e2 = __float_as_uint(q3.y);
                if (OCC) {   // any-hit: the child the ray leaves later first (see slab_any); no `steps`, no near / far sort
                    float x1, x2;
                    const bool h1 = slab_any(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f), x1);
                    const bool h2 = slab_any(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f), x2);
                    if (h1 && h2) { const bool firstIs2 = x2 > x1; cur = firstIs2 ? e2 : e1; STK(sp) = firstIs2 ? e1 : e2; sp++; }
                    else if (h1 || h2) cur = h1 ? e1 : e2;
                    else if (sp == 0) done = true;
                    else cur = STK(--sp);
                } else {
                    float d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
                    float d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
                    if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t e = e1; e1 = e2; e2 = e; }
                    if (d1 >= tLight) {
                        if (sp == 0) done = true;
                        else cur = STK(--sp);
                    } else {
                        if (STEPS) steps++;
                        cur = e1;
                        if (d2 < tLight) { STK(sp) = e2; sp++; if (STEPS) steps++; }
                    }
                }
            }
            if (done) {
                if (OCC) { if (occluded) q.sC[qFirst + slot] = splat(0.0f); }
                else {
                    q.hit[slot] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
                    if (STEPS) {
                        if (q.steps) q.steps[slot] = steps;
                        if (renderBVH) q.accum[q.firstPixel + slot] = splat((float)(uint32_t)steps / 255.f);
                    }
                }
                slot = -1;
            }
        }
    }
#ifdef RT355_TAIL_PROBE
    TAIL_PROBE_EXIT()
#endif
    if (lane == 0) { rays = wRays; wc.inst = wRays; wc.node = wNode; wc.prim = wPrim; if (STEPS) { wc.nodeIss = wNodeIss; wc.leafIss = wLeafIss; wc.evNode = wNode; wc.evPrim = wPrim; } }   // the wave's totals enter the reduction once
    flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
}

// ------------------------------------------------------------------ k_trace_persist_tlas: persistent wavefronts through a multi-BLAS TLAS (BVH2, layout 1)
// BASELINE config 5 (two BLAS under a TLAS; tlas.cl:9-77).  Same work distribution and event loop as k_trace_persist; what is new is
// that a lane's traversal has two levels.  Everything a lane still has to visit lives on ONE LDS stack column, the entries tagged by
// the space they belong to:
//     bit 31 set                      BLAS leaf        count << 24 | first          (as in k_trace_persist)
//     bits 31..29 = 000               BLAS interior    id in the dense pair table
//     bits 31..29 = 010  (kTagTlas)   TLAS interior    id in tlasPairsP
//     bits 31..29 = 011  (kTagInst)   TLAS leaf        instance id
// A lane is either in TLAS space (world ray in its registers, spBase = 0) or inside an instance (object-space ray in its registers;
// spBase = the stack height at entry, so "sp == spBase" means the instance's tree is exhausted).  The ray is transformed ONCE, at the
// moment the lane enters the instance (transformRay, tlas.cl:3-8, the arithmetic of traverse_instance), kept in registers over all
// events of that instance, and the world ray is fetched back from the queue when the lane leaves it (tlas.cl:21-23 restores a 128-byte
// backup; <= nBlas times per ray).  TLAS interior nodes go through the same code path as BLAS interior nodes - one 64-byte record,
// two slab tests, near child first, far child pushed (tlas.cl:48-75 is bvh.cl:41-52 with other names) - only the table differs.
// Visit order, pruning distances (tLight = ray.t on entering the level: tlas.cl:31, bvh.cl:19), `steps` (BLAS levels only: tlas.cl:44)
// and all work counters are those of traverse_tlas, so hits and counters are bit-identical to the one-ray-per-lane kernels.
// connect (OCC): inside an instance the any-hit order of k_trace_persist (later exit first); TLAS nodes keep the reference's
// near-first order, so the TLAS-visit and instance-visit counts stay the reference's (which instance is entered first decides whether
// the second one is entered at all).
static constexpr uint32_t kTagTlas = 0x40000000u, kTagInst = 0x60000000u, kTagMask = 0xe0000000u, kIdMask = 0x1fffffffu;
RT_FORCEINLINE bool slab_both(const TRay& r, float4 bmin, float4 bmax, float& tminOut, float& tmaxOut)
{
    float tx1 = (bmin.x - r.ox) * r.rx, tx2 = (bmax.x - r.ox) * r.rx;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (bmin.y - r.oy) * r.ry, ty2 = (bmax.y - r.oy) * r.ry;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (bmin.z - r.oz) * r.rz, tz2 = (bmax.z - r.oz) * r.rz;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    tminOut = tmin; tmaxOut = tmax;
    return tmax >= tmin && tmin < r.t && tmax > 0;
}
// SPILL: the LDS column holds the first q.stackCap entries of a lane's stack, deeper ones live in a global per-lane column (q.spill,
// entry-major so that the lanes of a wave write neighbouring words).  An SBVH at alpha = 0 gets DEEP (config 5's terrarium: 63 levels):
// 64 entries x 1 KB per workgroup would leave two workgroups per CU; capped at 20 entries seven fit, and the traversal - a chain of
// dependent fetches that lives on occupancy - rarely goes deeper than the cap (near child first keeps one pending sibling per level
// actually forked).  The two extra branches per push / pop cost a scene that does not need them ~9 % (EXPERIMENTS.md (19)), so the
// instantiation is chosen per scene at upload.
template <bool SPILL> RT_FORCEINLINE void stk_push(uint32_t* stk, const DevQueues& q, uint32_t gl, uint32_t& sp, uint32_t e)
{
    if (!SPILL || sp < q.stackCap) stk[sp * kBlock + threadIdx.x] = e;
    else q.spill[(size_t)(sp - q.stackCap) * q.spillStride + gl] = e;
    sp++;
}
template <bool SPILL> RT_FORCEINLINE uint32_t stk_pop(const uint32_t* stk, const DevQueues& q, uint32_t gl, uint32_t& sp)
{
    --sp;
    return (!SPILL || sp < q.stackCap) ? stk[sp * kBlock + threadIdx.x] : q.spill[(size_t)(sp - q.stackCap) * q.spillStride + gl];
}
// COH (extend of bounce 0 through the one-ray-per-lane branch): while every lane of the wave that is on an interior node is on the SAME
// one - primary rays of an 8x8 pixel tile, the first levels of the TLAS and of each BLAS - its record comes once through the scalar cache
// instead of 64 times through the vector memory pipeline (as in traverse_bvh2_packed_coherent; same arithmetic, order and counters).
template <bool OCC, bool STEPS = false, bool SPILL = false, bool COH = false>
__global__ __launch_bounds__(kBlock, 7) void k_trace_persist_tlas(DevScene sc, DevQueues q, int b0, int b1, int renderBVH, PersistTune tune)
{
    const int kChunk = tune.chunk, kRefill = tune.refill, kInner = tune.inner, kLeafK = tune.leafK;
    extern __shared__ uint32_t stk[];
    const int lane = threadIdx.x & 63;
    const int qFirst = OCC ? q.nShadow[b0] : 0;
    const int n = OCC ? q.nShadow[b1 + 1] - qFirst : q.nRays[b0];
    int32_t* cursor = q.cursor + (OCC ? (RT_MAX_BOUNCES + 2) + b0 : b0);
    const int nWaves = gridDim.x * (kBlock / 64), waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t gl = blockIdx.x * kBlock + threadIdx.x;    // this lane's column of the spill stack

    if (n <= nWaves * 64 || tune.flat) {
        // Short queue (late bounces; bounce 0 when launched with one workgroup per 256 rays): one ray per lane and nothing to refill
        // from, so every lane just runs its own state machine to the end - the same states and the same single stack column as the
        // event loop below (world ray re-fetched on leaving an instance instead of a nine-register backup), without the wave-level
        // path selection.  tune.flat: ANY queue this way, the persistent grid striding over it 64 rays per wave and round - scenes whose
        // rays take a dozen events (an open scene: most rays leave through the TLAS root or end on the floor) finish before the event
        // loop's bookkeeping pays, but still want the seven workgroups per CU that the capped stack column allows.
        WorkCtr wc = { 0, 0, 0, 0 };
        uint32_t rays = 0;
        // primary rays: a wave takes an 8x8 pixel tile instead of 64 pixels of a scan line (see trace_short_queue) - any slot -> lane map
        // is legal, every slot is traced on its own
        const bool tiled = !OCC && b0 == 0 && n == q.nPix && ((q.width | (q.nPix / q.width)) & 7) == 0;
        const int tilesX = q.width >> 3;
        int packWaves = nWaves;
        const int packId = (tune.xcdRays > 0 || tune.thin) && !tiled ? xcd_pack(n, tune.xcdRays, tune.xcdFirst, packWaves) : waveId;   // a sparse queue on few XCDs
        // ... and on few lanes of every participating wave (see trace_short_queue)
        int per = 64;
        if (tune.thin > 0 && !tiled && (long long)packWaves * tune.thin >= (long long)n) per = max(1, (n + packWaves - 1) / packWaves);
        for (int item = packId < 0 ? n : packId; (long long)item * per < (long long)n; item += packWaves) {
            int idx = item * per + lane;
            if (tiled) { const int ty = item / tilesX, tx = item - ty * tilesX; idx = ((ty << 3) + (lane >> 3)) * q.width + (tx << 3) + (lane & 7); }
            if (lane >= per || idx >= n) continue;
            TRay r;
            float tmax;
            {
                float4 O, D;
                if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tmax = a.w; }
                else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tmax = kFar; }
                r.ox = O.x; r.oy = O.y; r.oz = O.z; r.dx = D.x; r.dy = D.y; r.dz = D.z;
                r.rx = 1.0f / D.x; r.ry = 1.0f / D.y; r.rz = 1.0f / D.z;
            }
            r.t = tmax; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
            rays++;
            uint32_t cur = sc.tlasRootP, sp = 0, spBase = 0;
            bool inInst = false, occluded = false;
            float tLight = tmax;
            int steps = 0;
            for (;;) {
                bool needPop = false;
                if (cur & kLeafBit) {
                    const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
                    for (uint32_t i = 0; i < count; i++) {
                        wc.prim++;
                        test_tri_packed(sc, first + i, r);
                        if (OCC && r.t < tLight) { occluded = true; break; }
                    }
                    if (OCC && occluded) break;
                    needPop = true;
                } else if ((cur & kTagMask) == kTagInst) {
                    const float4* ir = sc.instRecs + (size_t)(cur & kIdMask) * 4;
                    const float4 t0 = ir[0], t1 = ir[1], t2 = ir[2], t3 = ir[3];
                    const float4 Dv = mk4(r.dx, r.dy, r.dz, 0.0f), Ov = mk4(r.ox, r.oy, r.oz, 0.0f);
                    r.dx = dot3(mk4(t0.x, t0.y, t0.z, 0), Dv); r.dy = dot3(mk4(t1.x, t1.y, t1.z, 0), Dv); r.dz = dot3(mk4(t2.x, t2.y, t2.z, 0), Dv);
                    r.ox = dot3(mk4(t0.x, t0.y, t0.z, 0), Ov) + t0.w; r.oy = dot3(mk4(t1.x, t1.y, t1.z, 0), Ov) + t1.w;
                    r.oz = dot3(mk4(t2.x, t2.y, t2.z, 0), Ov) + t2.w;
                    if (tune.backup) {   // the world ray waits in LDS (the reference keeps a 128-byte copy of the Ray, tlas.cl:12,21-23)
                        uint32_t* bk = stk + q.tlasLdsEntries * kBlock + threadIdx.x;
                        bk[0] = __float_as_uint(Ov.x); bk[kBlock] = __float_as_uint(Ov.y); bk[2 * kBlock] = __float_as_uint(Ov.z);
                        bk[3 * kBlock] = __float_as_uint(Dv.x); bk[4 * kBlock] = __float_as_uint(Dv.y); bk[5 * kBlock] = __float_as_uint(Dv.z);
                        bk[6 * kBlock] = __float_as_uint(r.rx); bk[7 * kBlock] = __float_as_uint(r.ry); bk[8 * kBlock] = __float_as_uint(r.rz);
                    }
                    r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
                    wc.inst++;
                    cur = __float_as_uint(t3.x);
                    spBase = sp; inInst = true; tLight = r.t;
                    continue;
                } else {
                    const bool isT = (cur & kTagTlas) != 0u;
                    if (isT) wc.tlas++; else wc.node++;
                    if (OCC) {
                        const float4* p = (isT ? sc.tlasPairsP : sc.pairs) + (size_t)(cur & kIdMask) * 4;
                        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                        const uint32_t e1 = __float_as_uint(q3.x), e2 = __float_as_uint(q3.y);
                        float n1, x1, n2, x2;
                        const bool h1 = slab_both(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f), n1, x1);
                        const bool h2 = slab_both(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f), n2, x2);
                        if (h1 && h2) { const bool firstIs2 = isT ? (n2 < n1) : (x2 > x1); cur = firstIs2 ? e2 : e1; stk_push<SPILL>(stk, q, gl, sp, firstIs2 ? e1 : e2); }
                        else if (h1 || h2) cur = h1 ? e1 : e2;
                        else needPop = true;
                    } else {
                        float d1, d2;
                        uint32_t e1, e2;
                        const uint32_t ucur = COH ? __builtin_amdgcn_readfirstlane(cur) : 0u;
                        if (COH && __ballot(cur != ucur) == 0ull) {   // (the slab tests are written out here too: they read the record from scalar registers)
                            const ConstF4 cp = (ConstF4)(uintptr_t)((ucur & kTagTlas) != 0u ? sc.tlasPairsP : sc.pairs) + (size_t)(ucur & kIdMask) * 4;
                            const fvec4 a = cp[0], b = cp[1], c = cp[2], d = cp[3];
                            d1 = slab(r, mk4(a.x, a.y, a.z, 0.0f), mk4(a.w, b.x, b.y, 0.0f));
                            d2 = slab(r, mk4(b.z, b.w, c.x, 0.0f), mk4(c.y, c.z, c.w, 0.0f));
                            e1 = __float_as_uint(d.x); e2 = __float_as_uint(d.y);
                        } else {
                            const float4* p = (isT ? sc.tlasPairsP : sc.pairs) + (size_t)(cur & kIdMask) * 4;
                            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                            d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
                            d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
                            e1 = __float_as_uint(q3.x); e2 = __float_as_uint(q3.y);
                        }
                        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t e = e1; e1 = e2; e2 = e; }
                        if (d1 >= tLight) needPop = true;
                        else {
                            if (!isT) steps++;
                            cur = e1;
                            if (d2 < tLight) { stk_push<SPILL>(stk, q, gl, sp, e2); if (!isT) steps++; }
                        }
                    }
                }
                if (needPop) {
                    if (inInst && sp == spBase) {
                        if (tune.backup) {
                            const uint32_t* bk = stk + q.tlasLdsEntries * kBlock + threadIdx.x;
                            r.ox = __uint_as_float(bk[0]); r.oy = __uint_as_float(bk[kBlock]); r.oz = __uint_as_float(bk[2 * kBlock]);
                            r.dx = __uint_as_float(bk[3 * kBlock]); r.dy = __uint_as_float(bk[4 * kBlock]); r.dz = __uint_as_float(bk[5 * kBlock]);
                            r.rx = __uint_as_float(bk[6 * kBlock]); r.ry = __uint_as_float(bk[7 * kBlock]); r.rz = __uint_as_float(bk[8 * kBlock]);
                            tLight = tmax;
                        } else {
                            float4 O, D;
                            if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tLight = a.w; }
                            else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tLight = kFar; }
                            r.ox = O.x; r.oy = O.y; r.oz = O.z; r.dx = D.x; r.dy = D.y; r.dz = D.z;
                            r.rx = 1.0f / D.x; r.ry = 1.0f / D.y; r.rz = 1.0f / D.z;
                        }
                        inInst = false; spBase = 0;
                    }
                    if (sp == 0) break;
                    cur = stk_pop<SPILL>(stk, q, gl, sp);
                }
            }
            if (OCC) { if (occluded) q.sC[qFirst + idx] = splat(0.0f); }
            else {
                q.hit[idx] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
                if (q.steps) q.steps[idx] = steps;
                if (renderBVH) q.accum[q.firstPixel + idx] = splat((float)(uint32_t)steps / 255.f);
            }
        }
        flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
        return;
    }

    uint32_t wRays = 0, wNode = 0, wPrim = 0, wTlas = 0, wInst = 0, wNodeIss = 0, wLeafIss = 0;    // this wave's work (wave-uniform)
    TRay r; r.t = 0; r.prim = -1; r.u = r.v = 0; r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.rx = r.ry = r.rz = 0;
    uint32_t cur = 0, sp = 0, spBase = 0;
    bool inInst = false;
    int slot = -1, steps = 0;
    float tLight = 0;
    int chunkNext = min(waveId * kChunk, n), chunkEnd = min(waveId * kChunk + kChunk, n);   // wave-uniform
    bool exhausted = false;
    int round = 0;

    // the world ray of queue slot `idx` into r (k_extend / k_connect: rD = 1 / D, three IEEE divides)
    auto world_ray = [&](int idx, float& tmax) {
        float4 O, D;
        if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tmax = a.w; }
        else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tmax = kFar; }
        r.ox = O.x; r.oy = O.y; r.oz = O.z; r.dx = D.x; r.dy = D.y; r.dz = D.z;
        r.rx = 1.0f / D.x; r.ry = 1.0f / D.y; r.rz = 1.0f / D.z;
    };

    for (;;) {
        const unsigned long long idleMask = __ballot(slot < 0);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 && exhausted && chunkNext >= chunkEnd) break;
        if (nIdle >= kRefill && !(exhausted && chunkNext >= chunkEnd)) {
            if (chunkNext >= chunkEnd) {
                int c = 0;
                if (tune.fixedChunks) { round++; c = round * nWaves * kChunk + waveId * kChunk; }
                else {
                    if (lane == 0) c = atomicAdd(cursor, kChunk);
                    c = __shfl(c, 0, 64) + nWaves * kChunk;
                }
                chunkNext = c; chunkEnd = min(c + kChunk, n);
                if (c >= n) { exhausted = true; chunkNext = chunkEnd = 0; }
            }
            if (chunkNext < chunkEnd) {
                const int rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                const int idx = chunkNext + rank;
                if (slot < 0 && idx < chunkEnd) {
                    float tmax;
                    world_ray(idx, tmax);
                    r.t = tmax; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
                    tLight = tmax; cur = sc.tlasRootP; sp = 0; spBase = 0; inInst = false; steps = 0; slot = idx;
                }
                wRays += (uint32_t)min(nIdle, chunkEnd - chunkNext);
                chunkNext = min(chunkNext + nIdle, chunkEnd);
            }
        }
#pragma unroll 1
        for (int it = 0; it < kInner; it++) {
            const bool act = slot >= 0, atLeaf = act && (cur & kLeafBit) != 0u;
            const bool atInst = act && (cur & kTagMask) == kTagInst;
            const bool atNode = act && !atLeaf && !atInst;
            const unsigned long long lm = __ballot(atLeaf), im = __ballot(atNode), xm = __ballot(atInst);
            if ((lm | im | xm) == 0ull) break;
            if (xm != 0ull) {
                // enter an instance (instanceIntersect, tlas.cl:9-26): rare (<= nBlas per ray) and short, so it goes first and alone
                wInst += (uint32_t)__popcll(xm);
                if (atInst) {
                    const float4* ir = sc.instRecs + (size_t)(cur & kIdMask) * 4;
                    const float4 t0 = ir[0], t1 = ir[1], t2 = ir[2], t3 = ir[3];
                    const float4 Dv = mk4(r.dx, r.dy, r.dz, 0.0f), Ov = mk4(r.ox, r.oy, r.oz, 0.0f);
                    r.dx = dot3(mk4(t0.x, t0.y, t0.z, 0), Dv); r.dy = dot3(mk4(t1.x, t1.y, t1.z, 0), Dv); r.dz = dot3(mk4(t2.x, t2.y, t2.z, 0), Dv);
                    r.ox = dot3(mk4(t0.x, t0.y, t0.z, 0), Ov) + t0.w; r.oy = dot3(mk4(t1.x, t1.y, t1.z, 0), Ov) + t1.w;
                    r.oz = dot3(mk4(t2.x, t2.y, t2.z, 0), Ov) + t2.w;
                    if (tune.backup) {
                        uint32_t* bk = stk + q.tlasLdsEntries * kBlock + threadIdx.x;
                        bk[0] = __float_as_uint(Ov.x); bk[kBlock] = __float_as_uint(Ov.y); bk[2 * kBlock] = __float_as_uint(Ov.z);
                        bk[3 * kBlock] = __float_as_uint(Dv.x); bk[4 * kBlock] = __float_as_uint(Dv.y); bk[5 * kBlock] = __float_as_uint(Dv.z);
                        bk[6 * kBlock] = __float_as_uint(r.rx); bk[7 * kBlock] = __float_as_uint(r.ry); bk[8 * kBlock] = __float_as_uint(r.rz);
                        bk[9 * kBlock] = __float_as_uint(tLight);       // the TLAS level's pruning distance (connect: t_light; extend: 1e30)
                    }
                    r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
                    cur = __float_as_uint(t3.x);          // encoded BLAS root (interior id or leaf)
                    spBase = sp; inInst = true; tLight = r.t;   // intersectBVH2 prunes against ray.t on entry (bvh.cl:19)
                }
                continue;
            }
            const bool doLeaf = im == 0ull || __popcll(lm) >= kLeafK;
            bool done = false, occluded = false, needPop = false;
            if (doLeaf) {
                wPrim += (uint32_t)__popcll(lm); if (STEPS) wLeafIss++;
                if (atLeaf) {
                    const uint32_t first = cur & 0x00ffffffu, count = (cur >> 24) & 0x7fu;
                    test_tri_packed(sc, first, r);
                    if (OCC && r.t < tLight) { done = true; occluded = true; }
                    else if (count > 1) cur = kLeafBit | ((count - 1) << 24) | (first + 1);
                    else needPop = true;
                }
            } else {
                if (STEPS) wNodeIss++;
                const bool isT = (cur & kTagTlas) != 0u;
                const unsigned long long tm = __ballot(atNode && isT);
                wTlas += (uint32_t)__popcll(tm); wNode += (uint32_t)(__popcll(im) - __popcll(tm));
                if (atNode) {
                    const float4* p = (isT ? sc.tlasPairsP : sc.pairs) + (size_t)(cur & kIdMask) * 4;
                    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                    uint32_t e1 = __float_as_uint(q3.x), e2 = __float_as_uint(q3.y);
                    if (OCC) {
                        float n1, x1, n2, x2;
                        const bool h1 = slab_both(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f), n1, x1);
                        const bool h2 = slab_both(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f), n2, x2);
                        if (h1 && h2) {
                            const bool firstIs2 = isT ? (n2 < n1) : (x2 > x1);   // TLAS: the reference's near-first (tlas.cl:58-63); BLAS: later exit first
                            cur = firstIs2 ? e2 : e1; stk_push<SPILL>(stk, q, gl, sp, firstIs2 ? e1 : e2);
                        }
                        else if (h1 || h2) cur = h1 ? e1 : e2;
                        else needPop = true;
                    } else {
                        float d1 = slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f));
                        float d2 = slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f));
                        if (d1 > d2) { float d = d1; d1 = d2; d2 = d; uint32_t e = e1; e1 = e2; e2 = e; }
                        if (d1 >= tLight) needPop = true;
                        else {
                            if (STEPS && !isT) steps++;
                            cur = e1;
                            if (d2 < tLight) { stk_push<SPILL>(stk, q, gl, sp, e2); if (STEPS && !isT) steps++; }
                        }
                    }
                }
            }
            if (needPop) {
                if (inInst && sp == spBase) {   // the instance's tree is exhausted: back to world space (tlas.cl:21-23)
                    if (tune.backup) {
                        const uint32_t* bk = stk + q.tlasLdsEntries * kBlock + threadIdx.x;
                        r.ox = __uint_as_float(bk[0]); r.oy = __uint_as_float(bk[kBlock]); r.oz = __uint_as_float(bk[2 * kBlock]);
                        r.dx = __uint_as_float(bk[3 * kBlock]); r.dy = __uint_as_float(bk[4 * kBlock]); r.dz = __uint_as_float(bk[5 * kBlock]);
                        r.rx = __uint_as_float(bk[6 * kBlock]); r.ry = __uint_as_float(bk[7 * kBlock]); r.rz = __uint_as_float(bk[8 * kBlock]);
                        tLight = __uint_as_float(bk[9 * kBlock]);
                    } else {
                        float tmax;
                        world_ray(slot, tmax);
                        tLight = tmax;
                    }
                    inInst = false; spBase = 0;   // the TLAS level prunes against the ray.t of ITS entry (tlas.cl:31)
                }
                if (sp == 0) done = true;
                else cur = stk_pop<SPILL>(stk, q, gl, sp);
            }
            if (done) {
                if (OCC) { if (occluded) q.sC[qFirst + slot] = splat(0.0f); }
                else {
                    q.hit[slot] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
                    if (STEPS) {
                        if (q.steps) q.steps[slot] = steps;
                        if (renderBVH) q.accum[q.firstPixel + slot] = splat((float)(uint32_t)steps / 255.f);
                    }
                }
                slot = -1;
            }
        }
    }
    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    if (lane == 0) { rays = wRays; wc.tlas = wTlas; wc.inst = wInst; wc.node = wNode; wc.prim = wPrim; if (STEPS) { wc.nodeIss = wNodeIss; wc.leafIss = wLeafIss; wc.evNode = wNode + wTlas; wc.evPrim = wPrim; } }
    flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
}

// ------------------------------------------------------------------ k_trace_persist4: persistent wavefronts over the BVH4 (layout 1, one BLAS)
// Same work distribution as k_trace_persist.  A lane is either on a NODE event (fetch the 128-byte quad record, four slab tests
// with the ray->t of node entry, push the interior children that were hit in slot order, remember the hit leaf children) or on a
// LEAF event (one triangle of the lowest pending leaf child).  The reference interleaves pushes and leaf tests in slot order
// (bvh.cl:78-92), but a push depends only on the distances taken at node entry, never on the shrinking ray->t, so doing the pushes
// first leaves the stack contents, the order of triangle tests, `steps` and the counters unchanged.
template <bool OCC>
__global__ __launch_bounds__(kBlock) void k_trace_persist4(DevScene sc, DevQueues q, int b0, int b1, int renderBVH, PersistTune tune)
{
    const int kChunk = tune.chunk, kRefill = tune.refill, kInner = tune.inner, kLeafK = tune.leafK;
    extern __shared__ uint32_t stk[];
    const int lane = threadIdx.x & 63;
    const int qFirst = OCC ? q.nShadow[b0] : 0;
    const int n = OCC ? q.nShadow[b1 + 1] - qFirst : q.nRays[b0];
    int32_t* cursor = q.cursor + (OCC ? (RT_MAX_BOUNCES + 2) + b0 : b0);
    const RtBVHInstance* inst = sc.blas + sc.tlas[0].BLASidx;
    const uint32_t rootNode = sc.rootEntry[sc.tlas[0].BLASidx];   // id of the root in the dense quad table
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = inst->invT[k];

    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    TRay r; r.t = 0; r.prim = -1; r.u = r.v = 0; r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.rx = r.ry = r.rz = 0;
    uint32_t cur = 0, sp = 0, leafMask = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    int slot = -1, steps = 0;
    float tLight = 0;
    const int nWaves = gridDim.x * (kBlock / 64), waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);

    auto setup = [&](int idx) {
        float4 O, D; float tmax;
        if (OCC) { const float4 a = q.sA[qFirst + idx], b = q.sB[qFirst + idx]; O = a; D = b; tmax = a.w; }
        else { O = q.O[b0 & 1][idx]; D = q.D[b0 & 1][idx]; tmax = kFar; }
        const float4 Dv = mk4(D.x, D.y, D.z, 0.0f), Ov = mk4(O.x, O.y, O.z, 0.0f);
        r.dx = dot3(mk4(T[0], T[1], T[2], 0), Dv); r.dy = dot3(mk4(T[4], T[5], T[6], 0), Dv); r.dz = dot3(mk4(T[8], T[9], T[10], 0), Dv);
        r.ox = dot3(mk4(T[0], T[1], T[2], 0), Ov) + T[3]; r.oy = dot3(mk4(T[4], T[5], T[6], 0), Ov) + T[7];
        r.oz = dot3(mk4(T[8], T[9], T[10], 0), Ov) + T[11];
        r.rx = 1.0f / r.dx; r.ry = 1.0f / r.dy; r.rz = 1.0f / r.dz;
        r.t = tmax; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
        tLight = tmax;
    };
    auto finish = [&](int idx, int st, bool occluded) {
        if (OCC) { if (occluded) q.sC[qFirst + idx] = splat(0.0f); }
        else {
            q.hit[idx] = mk4(r.t, __int_as_float(r.prim), r.u, r.v);
            if (q.steps) q.steps[idx] = st;
            if (renderBVH) q.accum[q.firstPixel + idx] = splat((float)(uint32_t)st / 255.f);
        }
    };

    if (n <= nWaves * 64) {   // short queue: plain one-ray-per-lane loop
        int idx = b0 == 0 && n == q.nPix ? waveId * 64 + lane : sparse_slot(n, tune, waveId, lane);
        if (!OCC && b0 == 0 && ((q.width | (q.nPix / q.width)) & 7) == 0 && n == q.nPix && (long long)gridDim.x * kBlock >= (long long)q.nPix) {   // primary rays: 8x8 pixel tile per wave (see trace_short_queue)
            const int tilesX = q.width >> 3, ty = waveId / tilesX, tx = waveId - ty * tilesX;
            idx = ((ty << 3) + (lane >> 3)) * q.width + (tx << 3) + (lane & 7);
        }
        if (idx < n) {
            setup(idx);
            rays = 1; wc.inst = 1;
            const int st = traverse_bvh4_packed<OCC>(sc, r, rootNode, stk, wc);
            finish(idx, st, st == -1);
        }
        flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
        return;
    }
    int chunkNext = min(waveId * kChunk, n), chunkEnd = min(waveId * kChunk + kChunk, n);
    bool exhausted = false;
    int round = 0;

    for (;;) {
        const unsigned long long idleMask = __ballot(slot < 0);
        const int nIdle = __popcll(idleMask);
        if (nIdle == 64 && exhausted && chunkNext >= chunkEnd) break;
        if (nIdle >= kRefill && !(exhausted && chunkNext >= chunkEnd)) {
            if (chunkNext >= chunkEnd) {
                int c = 0;
                if (tune.fixedChunks) { round++; c = round * nWaves * kChunk + waveId * kChunk; }   // see k_trace_persist
                else {
                    if (lane == 0) c = atomicAdd(cursor, kChunk);
                    c = __shfl(c, 0, 64) + nWaves * kChunk;
                }
                chunkNext = c; chunkEnd = min(c + kChunk, n);
                if (c >= n) { exhausted = true; chunkNext = chunkEnd = 0; }
            }
            if (chunkNext < chunkEnd) {
                const int rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                const int idx = chunkNext + rank;
                if (slot < 0 && idx < chunkEnd) {
                    setup(idx);
                    cur = rootNode; sp = 0; steps = 0; leafMask = 0; slot = idx;
                    rays++; wc.inst++;
                }
                chunkNext = min(chunkNext + nIdle, chunkEnd);
            }
        }
#pragma unroll 1
        for (int it = 0; it < kInner; it++) {
            const bool act = slot >= 0, atLeaf = act && leafMask != 0u;
            const unsigned long long lm = __ballot(atLeaf), im = __ballot(act && !atLeaf);
            if ((lm | im) == 0ull) break;
            const bool doLeaf = im == 0ull || __popcll(lm) >= kLeafK;
            bool done = false, occluded = false;
            if (doLeaf) {
                if (atLeaf) {
                    const int k = __ffs((int)leafMask) - 1;
                    uint32_t ek = k == 0 ? e0 : (k == 1 ? e1 : (k == 2 ? e2 : e3));
                    const uint32_t first = ek & 0x00ffffffu, count = (ek >> 24) & 0x7fu;
                    wc.prim++;
                    test_tri_packed(sc, first, r);
                    if (OCC && r.t < tLight) { done = true; occluded = true; }
                    else if (count > 1) {
                        ek = kLeafBit | ((count - 1) << 24) | (first + 1);
                        if (k == 0) e0 = ek; else if (k == 1) e1 = ek; else if (k == 2) e2 = ek; else e3 = ek;
                    } else {
                        leafMask &= leafMask - 1u;
                        if (leafMask == 0u) { if (sp == 0) done = true; else cur = STK(--sp); }
                    }
                }
            } else if (act && !atLeaf) {
                steps++; wc.node++;
                const float4* p = sc.quads + (size_t)cur * 8;
                const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5], q6 = p[6];
                asm volatile("" : : "v"(q0.x), "v"(q1.x), "v"(q2.x), "v"(q3.x), "v"(q4.x), "v"(q5.x), "v"(q6.x));   // one round trip (see test_tri_packed)
                e0 = __float_as_uint(q6.x); e1 = __float_as_uint(q6.y); e2 = __float_as_uint(q6.z); e3 = __float_as_uint(q6.w);
                const float d0 = e0 != kNoChild ? slab(r, mk4(q0.x, q0.y, q0.z, 0.0f), mk4(q0.w, q1.x, q1.y, 0.0f)) : kFar;
                const float d1 = e1 != kNoChild ? slab(r, mk4(q1.z, q1.w, q2.x, 0.0f), mk4(q2.y, q2.z, q2.w, 0.0f)) : kFar;
                const float d2 = e2 != kNoChild ? slab(r, mk4(q3.x, q3.y, q3.z, 0.0f), mk4(q3.w, q4.x, q4.y, 0.0f)) : kFar;
                const float d3 = e3 != kNoChild ? slab(r, mk4(q4.z, q4.w, q5.x, 0.0f), mk4(q5.y, q5.z, q5.w, 0.0f)) : kFar;
                uint32_t m = 0;
                if (e0 != kNoChild && d0 < tLight) { if (e0 & kLeafBit) m |= 1u; else { STK(sp) = e0; sp++; } }
                if (e1 != kNoChild && d1 < tLight) { if (e1 & kLeafBit) m |= 2u; else { STK(sp) = e1; sp++; } }
                if (e2 != kNoChild && d2 < tLight) { if (e2 & kLeafBit) m |= 4u; else { STK(sp) = e2; sp++; } }
                if (e3 != kNoChild && d3 < tLight) { if (e3 & kLeafBit) m |= 8u; else { STK(sp) = e3; sp++; } }
                leafMask = m;
                if (m == 0u) { if (sp == 0) done = true; else cur = STK(--sp); }
            }
            if (done) { finish(slot, steps, occluded); slot = -1; leafMask = 0; }
        }
    }
    flush_counters(OCC ? q.ctrConnect : q.ctrExtend, rays, wc, stk);
}

// ------------------------------------------------------------------ shading helpers
struct SRay { // the reference Ray fields shade() reads and writes
    float4 O, D, N, I, inten;
    float t, u, v; int prim, bounces, pixel, matIdx; bool inside, lastSpec;
};
struct ExtRay { float4 O, D, inten; int bounces; bool inside, lastSpec, valid; };

RT_FORCEINLINE float4 prim_normal(const RtPrimitive* p, float4 I) // primitives.cl:91-105
{
    const int type = p->objType;
    if (type == RT_PRIM_SPHERE) return muls(sub4(I, ld4(p->obj.sphere.pos)), p->obj.sphere.invr);
    if (type == RT_PRIM_PLANE) return ld4(p->obj.plane.N);
    return ld4(p->obj.triangle.N);
}
// Texel `i` of the atlas.  The reference indexes `textures` unchecked (primitives.cl:124,134,145): uv == 1 lands one texel or one row
// past a texture and a plane with negative u or v up to a whole texture past it - inside the atlas that reads a neighbouring
// texture's texel (reproduced), past its end it is undefined behaviour.  Here, and in the oracle, texels outside the atlas are zero.
// float -> int with the hardware's own rule (v_cvt_i32_f32 / v_cvt_u32_f32: NaN -> 0, out of range saturates), spelled out so that it
// is defined C++ rather than a poison fptosi.  It matters: a sphere hit found with w-lane-polluted dots has a non-unit normal,
// acos(N.y) is then NaN and the reference's kernels read texel row 0 (tests/test_gpu_reference.py, whole-frame comparison).
RT_FORCEINLINE int f2i_gpu(float x) { return x != x ? 0 : (x >= 2147483648.0f ? 2147483647 : (x <= -2147483648.0f ? (int)(-2147483647 - 1) : (int)x)); }
RT_FORCEINLINE uint32_t f2u_gpu(float x) { return !(x > 0.0f) ? 0u : (x >= 4294967296.0f ? 0xffffffffu : (uint32_t)x); }
RT_FORCEINLINE float4 texel(const DevScene& sc, long long i) { return i >= 0 && i < (long long)sc.nTex ? sc.tex[i] : splat(0.0f); }
// The scalar fields of a Material (bytes 32..63: specular, n1, n2, isDielectric | texIdx, texW, texH, isLight) fetched as two 16-byte
// loads kept together; read field by field where they are used, the compiler turns them into a chain of up to eight dependent fetches.
struct MatCtl { float specular, n1, n2; uint32_t isDielectric; int32_t texIdx, texW, texH; uint32_t isLight; };
RT_FORCEINLINE MatCtl load_mat_ctl(const RtMaterial* mat)
{
    const uint4* p = reinterpret_cast<const uint4*>(mat);
    const uint4 a = p[2], b = p[3];
    asm volatile("" : : "v"(a.x), "v"(b.x));
    MatCtl m;
    m.specular = __uint_as_float(a.x); m.n1 = __uint_as_float(a.y); m.n2 = __uint_as_float(a.z); m.isDielectric = a.w & 0xffu;
    m.texIdx = (int32_t)b.x; m.texW = (int32_t)b.y; m.texH = (int32_t)b.z; m.isLight = b.w & 0xffu;
    return m;
}
RT_FORCEINLINE float4 albedo_of(const DevScene& sc, const RtPrimitive* prim, const RtMaterial* mat, const MatCtl& mc, const SRay& ray) // primitives.cl:107-148
{
    float4 albedo = ld4(mat->color);
    const int texIdx = mc.texIdx;
    if (texIdx != -1) {
        const int texW = mc.texW, texH = mc.texH, type = prim->objType;
        if (type == RT_PRIM_TRIANGLE) {
            const RtTriangle* t = &prim->obj.triangle;
            float w2 = 1 - ray.u - ray.v;
            float ux = fmodf(ray.u * t->uv1.x + ray.v * t->uv0.x + w2 * t->uv2.x, 1.f);
            float uy = fmodf(ray.u * t->uv1.y + ray.v * t->uv0.y + w2 * t->uv2.y, 1.f);
            if (ux < 0) ux = 1 + ux;
            if (uy < 0) uy = 1 + uy;
            int x = f2i_gpu(ux * (float)texW), y = f2i_gpu(uy * (float)texH);
            albedo = texel(sc, (long long)texIdx + x + (long long)y * texW);
        } else if (type == RT_PRIM_SPHERE) {
#ifdef RT355_REF_BUILTINS
            float ux = (1.0f + __ocml_atan2pi_f32(ray.N.z, ray.N.x)) * 0.5f;   // primitives.cl:130 (int + float is a float add; * 0.5 is exact in any precision)
            float uy = __ocml_acospi_f32(ray.N.y);                             // :131
#else
            float ux = (float)((1 + rt_atan2f(ray.N.z, ray.N.x) / 3.14159265358979323846) * 0.5);
            float uy = rt_acosf(ray.N.y) / 3.14159265358979323846f;
#endif
            int x = f2i_gpu(ux * (float)texW), y = f2i_gpu(uy * (float)texH);
            albedo = texel(sc, (long long)texIdx + x + (long long)y * texW);
        } else {
            float u = fmodf(ray.u, 1.f), v = fmodf(ray.v, 1.f);
            if (u < 0) u = 1 - u;
            if (v < 0) v = 1 - v;
            int x = f2i_gpu(u * (float)texW), y = f2i_gpu(v * (float)texH);
            albedo = texel(sc, (long long)texIdx + x + (long long)y * texW);
        }
    }
    return albedo;
}
RT_FORCEINLINE float survival_prob(float4 a) { return fminf(fmaxf(fmaxf(a.x, fmaxf(a.y, a.z)), 0.f), 1.f); } // primitives.cl:150-153
RT_FORCEINLINE float4 random_point_on(const RtPrimitive* p, uint32_t& seed) // primitives.cl:155-189
{
    if (p->objType == RT_PRIM_SPHERE) {
        float theta = rnd_abs(seed) * 2.0f * kPi;
        float u = rnd_abs(seed) * 2.0f - 1.0f;
        float pre = sqrtf(1 - u * u);
        float x = rt_cosf(theta) * pre, y = rt_sinf(theta) * pre;
        return add4(muls(mk4(x, y, u, 0.0f), p->obj.sphere.r), ld4(p->obj.sphere.pos));
    }
    const float4 v0 = ld4(p->obj.triangle.v0), v1 = ld4(p->obj.triangle.v1), v2 = ld4(p->obj.triangle.v2);
    float u1 = rnd_abs(seed), u2 = rnd_abs(seed);
    if (u1 + u2 > 1) { u1 = 1 - u1; u2 = 1 - u2; }
    float4 a = sub4(v1, v0), b = sub4(v2, v0);
    return add4(add4(v0, muls(a, u1)), muls(b, u2));
}
RT_FORCEINLINE float4 sample_ball(uint32_t& seed) // ray.cl:49-53,62-69 (w lane = -1 before normalising)
{
    float4 p = sub4(muls(rnd_float3(seed), 2.0f), splat(1.0f));
    while (p.x * p.x + p.y * p.y + p.z * p.z > 1.0f) p = sub4(muls(rnd_float3(seed), 2.0f), splat(1.0f));
    return normalize4(p);
}
RT_FORCEINLINE float4 sample_dir(int sampling, float4 N, uint32_t& seed) // ray.cl:46-72
{
    float4 p = sample_ball(seed);
    if (sampling == RT_SAMPLING_HEMISPHERE) return dot4(N, p) < 0.0f ? neg4(p) : p;
    return normalize4(add4(N, p));
}
RT_FORCEINLINE ExtRay reflect_ray(const SRay& ray) // ray.cl:21-29 (inside resets to false)
{
    float dnd = dot4(ray.N, ray.D);
    float4 reflected = sub4(ray.D, muls(muls(ray.N, 2.0f), dnd));
    ExtRay e;
    e.O = add4(ray.I, muls(muls(reflected, 2.0f), kEps));
    e.D = reflected; e.inten = ray.inten; e.bounces = ray.bounces + 1; e.inside = false; e.lastSpec = false; e.valid = true;
    return e;
}
RT_FORCEINLINE ExtRay transmit_ray(const SRay& ray, float4 T) // ray.cl:31-39
{
    ExtRay e;
    e.O = add4(ray.I, muls(T, kEps));
    e.D = T; e.inten = ray.inten; e.bounces = ray.bounces + 1; e.inside = !ray.inside; e.lastSpec = false; e.valid = true;
    return e;
}
RT_FORCEINLINE float fresnel(SRay& ray, const RtMaterial* mat, const MatCtl& mc, float4& outT) // glass.cl:4-49
{
    float costhetai = dot4(ray.N, muls(ray.D, -1.0f));
    float n1 = mc.n1, n2 = mc.n2;
    if (ray.inside) {
        n1 = mc.n2; n2 = mc.n1;
        ray.inten.x *= rt_expf(-mat->absorption.x * ray.t);
        ray.inten.y *= rt_expf(-mat->absorption.y * ray.t);
        ray.inten.z *= rt_expf(-mat->absorption.z * ray.t);
    }
    float frac = n1 * (1 / n2);
    float k = 1 - frac * frac * (1 - costhetai * costhetai);
    if (k < 0) return 1.f;
    outT = normalize4(add4(muls(ray.D, frac), muls(ray.N, frac * costhetai - sqrtf(k))));
    float costhetat = dot4(neg4(ray.N), outT);
    float n1ci = n1 * costhetai, n2ci = n2 * costhetai, n1ct = n1 * costhetat, n2ct = n2 * costhetat;
    float frac1 = (n1ci - n2ct) / (n1ci + n2ct);
    float frac2 = (n1ct - n2ci) / (n1ct + n2ci);
    float Fr = 0.5f * (frac1 * frac1 + frac2 * frac2);
    return mc.specular + (1 - mc.specular) * Fr;
}
RT_FORCEINLINE float4 firefly(int on, float4 c) // wavefront.cl:125-127,196-198
{
    if (on && dot4(c, c) > 25) return muls(normalize4(c), 5.0f);
    return c;
}

struct ShadowOut { float4* a; float4* b; float4* c; bool valid; };   // a/b/c: this lane's LDS slots (stored as soon as known)

// neeShading (shading.cl:72-169) and kajiyaShading (:7-70) in one body; NEE selects the
// light-sampling block and the lastSpecular rules.
template <bool NEE>
RT_FORCEINLINE float4 shade_hit(const DevScene& sc, const DevVariant& var, SRay& ray, uint32_t& seed, ExtRay& ext, ShadowOut& sh)
{
    const RtPrimitive* prim = sc.prims + ray.prim;
    const RtMaterial* mat = sc.mats + ray.matIdx;
    const MatCtl mc = load_mat_ctl(mat);
    if (mc.isLight) {
        if (NEE && !ray.lastSpec) return splat(0.0f);
        return mul4(ray.inten, ld4(mat->emittance));
    }
    const float rnd = rnd_float(seed);
    if (mc.isDielectric) {
        float4 T = splat(0.0f);
        float Fr = fresnel(ray, mat, mc, T);
        ext = rnd < Fr ? reflect_ray(ray) : transmit_ray(ray, T);
        if (NEE) ext.lastSpec = true;
    } else if (rnd < mc.specular) {
        ext = reflect_ray(ray);
        if (NEE) ext.lastSpec = true;
    } else {
        const float4 albedo = albedo_of(sc, prim, mat, mc, ray);
        const float4 BRDF = muls(albedo, kInvPi);
        if (NEE && sc.nLights > 0) {
            uint32_t li = f2u_gpu(floorf(rnd_abs(seed) * (float)sc.nLights));
            if (li >= (uint32_t)sc.nLights) li = (uint32_t)sc.nLights - 1; // reference reads out of bounds here (draw == 1.0)
            // One round trip for everything NEE reads of the light (round 1 walked lights[li] -> Primitive.objType -> vertices -> normal ->
            // matIdx / area -> Material.emittance: six dependent fetches); same arithmetic as getRandomPoint / getNormal (primitives.cl:91-189).
            const float4* LR = sc.lightRecs + (size_t)li * 8;
            const float4 l0 = LR[0], l1 = LR[1], l2 = LR[2], l3 = LR[3], l4 = LR[4], l5 = LR[5];
            asm volatile("" : : "v"(l0.x), "v"(l1.x), "v"(l2.x), "v"(l3.x), "v"(l4.x), "v"(l5.x));
            const int ltype = __float_as_int(l4.x);
            const float larea = l4.y;
            float4 pl, Nl;
            if (ltype == RT_PRIM_SPHERE) {            // Sphere { pos; r, r2, invr }
                float theta = rnd_abs(seed) * 2.0f * kPi;
                float u = rnd_abs(seed) * 2.0f - 1.0f;
                float pre = sqrtf(1 - u * u);
                float x = rt_cosf(theta) * pre, y = rt_sinf(theta) * pre;
                pl = add4(muls(mk4(x, y, u, 0.0f), l1.x), l0);
                Nl = muls(sub4(pl, l0), l1.z);
            } else {                                  // Triangle { v0, v1, v2, N } (a plane light samples like the reference: as a triangle)
                float u1 = rnd_abs(seed), u2 = rnd_abs(seed);
                if (u1 + u2 > 1) { u1 = 1 - u1; u2 = 1 - u2; }
                float4 a = sub4(l1, l0), b = sub4(l2, l0);
                pl = add4(add4(l0, muls(a, u1)), muls(b, u2));
                Nl = ltype == RT_PRIM_PLANE ? l0 : l3;
            }
            float4 dirToLight = sub4(pl, ray.I);
            float dist = length4(dirToLight);
            float4 L = muls(dirToLight, 1 / dist);
            float dotNL = dot4(ray.N, L);
            if (dotNL > 0 && dot4(Nl, neg4(L)) > 0) {
                // The shadow ray carries what connect() needs (wavefront.cl:175-199): its
                // origin/direction/t_max and the radiance it adds when unoccluded.
                float4 sInt = muls(ray.inten, (float)sc.nLights);
                float solidAngle = dot4(Nl, neg4(L)) * larea * (1 / (dist * dist));
                float4 lightColor = l5;
                float4 Ld = muls(mul4(muls(lightColor, solidAngle), BRDF), dotNL);
                float4 color = firefly(var.fireflies, mul4(Ld, sInt));
                float4 so = add4(ray.I, muls(L, kEps));
                *sh.a = mk4(so.x, so.y, so.z, dist - 2 * kEps);
                *sh.b = mk4(L.x, L.y, L.z, __int_as_float(ray.pixel));
                *sh.c = color;
                sh.valid = true;
            }
        }
        if (var.rr) {
            float rr_p = survival_prob(albedo);
            if (rr_p < rnd_float(seed)) return splat(0.0f);
            ray.inten = muls(ray.inten, 1 / rr_p);
        }
        float4 refl = sample_dir(var.sampling, ray.N, seed);
        float dotNR = dot4(ray.N, refl);
        float I_PDF = var.sampling == RT_SAMPLING_HEMISPHERE ? 2 * kPi : dotNR * kPi;
        ext.O = add4(ray.I, muls(refl, kEps));
        ext.D = refl;
        if (NEE) ext.inten = muls(muls(mul4(ray.inten, BRDF), I_PDF), dotNR);         // shading.cl:161
        else     ext.inten = mul4(ray.inten, muls(muls(BRDF, I_PDF), dotNR));         // shading.cl:60
        ext.bounces = ray.bounces + 1; ext.inside = ray.inside; ext.lastSpec = false; ext.valid = true;
    }
    return splat(0.0f);
}

// ------------------------------------------------------------------ k_shade (single pass, stable compaction)
// Tile status word of the decoupled look-back scan: [63:62] flag (0 empty, 1 aggregate, 2 inclusive
// prefix), [61:31] extension-ray count, [30:0] shadow-ray count.  Flag and payload travel in ONE 8-byte
// word written/read with agent-scope atomics (L2-coherent across XCDs), so no separate fence is needed.
static constexpr int kTile = 512;   // queue slots per tile = threads of a k_shade workgroup (a multiple of the 256 slots k_generate arms per workgroup)
static constexpr unsigned long long kTileAgg = 1ull << 62, kTilePrefix = 2ull << 62, kTilePrefixZero = 2ull << 62;
RT_FORCEINLINE unsigned long long tile_pack(unsigned long long flag, uint32_t e, uint32_t s) { return flag | ((unsigned long long)e << 31) | (unsigned long long)s; }
RT_FORCEINLINE uint32_t tile_ext(unsigned long long v) { return (uint32_t)((v >> 31) & 0x7fffffffull); }
RT_FORCEINLINE uint32_t tile_sh(unsigned long long v) { return (uint32_t)(v & 0x7fffffffull); }

// Persistent workgroups take tiles (kTile consecutive queue slots) by ticket (see the kernel).  Publishing a tile's counts waits
// for nothing; closing a super-tile and resolving a tile's position only wait for words of smaller tile / super-tile ids.
// Bounded wait on a status word: every spin in this library has an upper bound (~seconds), after which the kernel raises
// q.fault and carries on with a zero payload instead of hanging the GPU; the host reports RT_E_DEVICE.
static constexpr uint32_t kSpinLimit = 1u << 22;   // x ~1-2 us per poll: several seconds
RT_FORCEINLINE unsigned long long wait_word(const unsigned long long* p, unsigned long long needFlag /*0: any non-empty, 2: prefix*/, int32_t* fault, int32_t what = 1)
{
    unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (needFlag == 2ull ? (v >> 62) != 2ull : (v >> 62) == 0ull) {
        // give up when the bound is reached - or when another wave already has: once the flag is up the launch's results are void
        if (++spins > kSpinLimit || ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { atomicCAS(fault, 0, what); return kTilePrefixZero; }   // `what`: which word was waited for (diagnostic)
        // back off: a few quick polls, then ~1 us naps, so that thousands of waiting waves do not flood the L2 with polls
        if (spins < 4) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}

// TILE = queue slots per tile = threads of the workgroup: 512 for a context that has the GPU to itself (half the tickets, publishes and
// drains per ray), 256 for contexts that share it (39 instead of 78 KB of LDS: such a workgroup fits beside the other contexts' traversal
// workgroups instead of waiting for them to leave the CU: +2 % with three lanes, -4.6 % alone).
template <bool NEE, int TILE>
__global__ __launch_bounds__(TILE, 4) void k_shade(DevScene sc, DevQueues q, DevVariant var, int bounce)
{
    // [2]: a tile's outputs are written one tile late (see the loop), so the counts and the shadow records of two tiles are alive
    __shared__ uint32_t sWaveE[2][TILE / 64], sWaveS[2][TILE / 64], sBaseE, sBaseS;
    // Survivors wait in LDS (104 B per lane) while the ordered scan resolves, instead of in ~28 registers: the
    // kernel's occupancy is set by the shading code, not by values that are merely parked across the scan.
    __shared__ float4 sExtO[TILE], sExtD[TILE], sExtI[TILE], sShA[2][TILE], sShB[2][TILE], sShC[2][TILE];
    __shared__ uint2 sExtM[TILE];
    const int cur = bounce & 1, nxt = cur ^ 1;
    unsigned long long* state = q.tile[cur];
    unsigned long long* const* super = q.super;
    const int n = q.nRays[bounce];
    if (n <= 0) { // empty queue: still publish the (empty) next queue
        if (blockIdx.x == 0 && threadIdx.x == 0) { q.nRays[bounce + 1] = 0; q.nShadow[bounce + 1] = q.nShadow[bounce]; }
        return;
    }
    const uint32_t numTiles = (uint32_t)((n + TILE - 1) / TILE);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int shadowBase = q.nShadow[bounce];

    // Resolve the ordered scan for tile `t` (its counts are in sWave*[pp], its aggregate has been published) and write its
    // survivors to their final queue positions.  Called by the whole workgroup.
    auto drain = [&](uint32_t t, int pp, unsigned long long em, unsigned long long sm, bool extValid, bool shValid) {
        if (wave == 0) {
            uint32_t aggE = 0, aggS = 0;
#pragma unroll
            for (int w = 0; w < TILE / 64; w++) { aggE += sWaveE[pp][w]; aggS += sWaveS[pp][w]; }
            // Level 1 of the ordered scan: the counts of the earlier tiles of the own super-tile (one 64-lane read) plus the
            // inclusive prefix of the previous super-tile (level 2, resolved by whichever workgroup closed that super-tile).
            const uint32_t sup = t >> 6, inSup = t & 63u;
            // lanes 0..inSup-1 fetch the earlier tiles' words, lane 63 the previous super-tile's: one round trip for both levels
            unsigned long long v = 0ull;
            if ((uint32_t)lane < inSup) v = wait_word(&state[1 + (sup << 6) + (uint32_t)lane], 0ull, q.fault, 0x10000000 | (int32_t)((sup << 6) + (uint32_t)lane));
            else if (lane == 63 && sup > 0) v = wait_word(&super[cur][sup - 1], 2ull, q.fault, 0x20000000 | (int32_t)(sup - 1));
            const unsigned long long vp = __shfl(v, 63, 64);
            const uint32_t preE = sup > 0 ? tile_ext(vp) : 0u, preS = sup > 0 ? tile_sh(vp) : 0u;   // inclusive prefix of super-tiles [0, sup)
            uint32_t inE = (uint32_t)lane < inSup ? tile_ext(v) : 0u, inS = (uint32_t)lane < inSup ? tile_sh(v) : 0u;   // counts of tiles [64*sup, t)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { inE += __shfl_xor(inE, off, 64); inS += __shfl_xor(inS, off, 64); }
            if (lane == 0) {
                sBaseE = preE + inE; sBaseS = preS + inS;
                if (t == numTiles - 1) { q.nRays[bounce + 1] = (int)(preE + inE + aggE); q.nShadow[bounce + 1] = shadowBase + (int)(preS + inS + aggS); }
            }
        }
        __syncthreads();
        uint32_t baseE = sBaseE, baseS = sBaseS + (uint32_t)shadowBase;
        for (int w = 0; w < wave; w++) { baseE += sWaveE[pp][w]; baseS += sWaveS[pp][w]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (extValid) {
            const uint32_t dst = baseE + (uint32_t)__popcll(em & below);
            q.O[nxt][dst] = sExtO[threadIdx.x]; q.D[nxt][dst] = sExtD[threadIdx.x]; q.inten[nxt][dst] = sExtI[threadIdx.x];
            q.meta[nxt][dst] = sExtM[threadIdx.x];
        }
        if (shValid) {
            const uint32_t dst = baseS + (uint32_t)__popcll(sm & below);
            q.sA[dst] = sShA[pp][threadIdx.x]; q.sB[dst] = sShB[pp][threadIdx.x]; q.sC[dst] = sShC[pp][threadIdx.x];
        }
        __syncthreads();   // sExt*, sBase* and sSh*[pp] are free again
    };

    // Software pipeline over this workgroup's tiles: a tile's aggregate is published as soon as it is shaded, but its look-back
    // is resolved - and its survivors are written - only after the NEXT tile has been shaded.  By then the workgroups that own
    // the preceding tiles have long published theirs, so the scan no longer stalls the workgroup for the skew between its
    // neighbours (measured: 9-10 us of waiting per tile against 5-8 us of shading when resolved immediately).
#ifdef RT355_SHADE_TIMING
    long long tShade = 0, tDrain = 0, tBegin = wall_clock64(); int nT = 0;
#endif
    bool pend = false, pExt = false, pSh = false;
    uint32_t pTile = 0;
    unsigned long long pEm = 0ull, pSm = 0ull;
    int par = 0;
    // the queue entries of the NEXT tile are fetched before the previous tile is drained, so their latency hides behind the drain
    struct TileIn { float4 hit, O, D, inten; uint2 meta; uint32_t seed; };
    auto fetch = [&](uint32_t t, TileIn& in) {
        const int j = (int)t * TILE + threadIdx.x;
        if (t < numTiles && j < n) {
            in.hit = ldnt4(&q.hit[j]); in.meta = ldnt2(&q.meta[cur][j]); in.O = ldnt4(&q.O[cur][j]); in.D = ldnt4(&q.D[cur][j]); in.inten = ldnt4(&q.inten[cur][j]);   // last use
            in.seed = q.seeds[j];
        }
    };
    // Tiles are handed out by ticket to workgroups that are RUNNING, in increasing order (kTicketClasses = 1: one counter, see there;
    // the class arithmetic below is kept general).  What a tile waits for are words of smaller tile / super-tile ids; the smallest
    // unpublished tile is either held by a running workgroup, which publishes before it waits for anything newer, or it is the next
    // ticket, which any running workgroup draws as soon as its own (smaller, hence published) tile is done.  So the scan needs no
    // co-resident grid and also gets through when the GPU is shared with other contexts and processes.
    __shared__ uint32_t sTicket;
    const uint32_t cls = blockIdx.x % (uint32_t)kTicketClasses;
    int32_t* ticket = q.shadeTicket + ((size_t)bounce * kTicketClasses + cls) * kTicketStride;
    uint32_t tkNext = 0;                                   // thread 0: ticket of the next tile (drawn while this one is shaded)
    if (threadIdx.x == 0) sTicket = (uint32_t)atomicAdd(ticket, 1);
    __syncthreads();
    uint32_t tile = sTicket * (uint32_t)kTicketClasses + cls;
    __syncthreads();                                       // sTicket is rewritten inside the loop
    TileIn in;
    fetch(tile, in);
    for (; tile < numTiles; par ^= 1) {
        const int i = (int)tile * TILE + threadIdx.x;
        if (threadIdx.x == 0) {
            q.tile[nxt][1 + tile] = 0ull;     // arm shade(bounce+1)'s scan (its queue is never longer)
            tkNext = (uint32_t)atomicAdd(ticket, 1);
        }
#ifdef RT355_SHADE_TIMING
        const long long c0 = wall_clock64();
#endif

        bool extValid = false, shValid = false;
        ExtRay ext; ext.valid = false;
        uint2 extMeta = make_uint2(0u, 0u);
        if (i < n) {
            ShadowOut sh; sh.valid = false; sh.a = &sShA[par][threadIdx.x]; sh.b = &sShB[par][threadIdx.x]; sh.c = &sShC[par][threadIdx.x];
            const float4 hit = in.hit;
            const uint2 meta = in.meta;
            SRay ray;
            ray.O = in.O; ray.D = in.D; ray.inten = in.inten;
            ray.t = hit.x; ray.prim = __float_as_int(hit.y); ray.u = hit.z; ray.v = hit.w;
            ray.pixel = (int)meta.x; ray.bounces = (int)(meta.y & kMetaBounceMask);
            ray.inside = (meta.y & kMetaInside) != 0; ray.lastSpec = (meta.y & kMetaLastSpec) != 0;
            if (ray.prim == -1) { // wavefront.cl:109-112, sky = skydome.cl:7
                float4 c = mul4(ray.inten, mk4(0.0784f, 0.0941f, 0.3215f, 0.0f));
                q.accum[ray.pixel] = add4(q.accum[ray.pixel], c);
            } else {
                // what extend() leaves in the ray (wavefront.cl:69-72)
                ray.I = add4(ray.O, muls(ray.D, ray.t));
                // normal and material id from the dense 16-byte shading record (4 MB for 265k primitives: L2-resident) instead
                // of two fields 68 bytes apart in the 128-byte Primitive (34 MB); spheres need the hit point: reference layout
                const float4 rec = sc.shadeRecs[ray.prim];
                asm volatile("" : : "v"(rec.x), "v"(rec.w));   // one 16-byte fetch, not the tag word first and the normal after the branch on it
                const uint32_t tag = __float_as_uint(rec.w);
                ray.matIdx = (int)(tag & 0x07ffffffu);
                ray.N = (tag >> 28) == RT_PRIM_SPHERE ? prim_normal(sc.prims + ray.prim, ray.I)
                                                      : mk4(rec.x, rec.y, rec.z, (tag & 0x08000000u) ? -0.0f : 0.0f);   // flipped normals carry w = -0
                if (dot4(ray.N, neg4(ray.D)) < 0) ray.N = muls(ray.N, -1.0f);
                uint32_t seed = in.seed;
                float4 color = shade_hit<NEE>(sc, var, ray, seed, ext, sh);
                q.seeds[i] = seed;
                color = firefly(var.fireflies, color);
                // one path per pixel and launch: the add is race-free; adding an exact zero is skipped
                if (color.x != 0.0f || color.y != 0.0f || color.z != 0.0f || color.w != 0.0f)
                    q.accum[ray.pixel] = add4(q.accum[ray.pixel], color);
                if (ext.valid && ext.bounces <= RT_MAX_BOUNCES) {   // wavefront.cl:129
                    extValid = true;
                    extMeta = make_uint2((uint32_t)ray.pixel, (uint32_t)ext.bounces | (ext.inside ? kMetaInside : 0u) | (ext.lastSpec ? kMetaLastSpec : 0u));
                }
                shValid = sh.valid;
            }
        }
        // wave votes; publish the tile's counts at once (replaces atomic_inc on numOutRays / shadowRays, wavefront.cl:131,136)
        const unsigned long long em = __ballot(extValid), sm = __ballot(shValid);
        if (lane == 0) { sWaveE[par][wave] = (uint32_t)__popcll(em); sWaveS[par][wave] = (uint32_t)__popcll(sm); }
        if (threadIdx.x == 0) sTicket = tkNext;
        __syncthreads();
        if (wave == 0) {
            uint32_t aggE = 0, aggS = 0;
#pragma unroll
            for (int w = 0; w < TILE / 64; w++) { aggE += sWaveE[par][w]; aggS += sWaveS[par][w]; }
            const uint32_t sup = tile >> 6;
            unsigned long long acc = 0ull;
            if (lane == 0) {
                __hip_atomic_store(&state[1 + tile], tile_pack(kTileAgg, aggE, aggS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((tile & 63u) == 0u) { super[nxt][sup] = 0ull; q.supAcc[nxt][sup] = 0ull; }   // arm shade(bounce+1)'s level 2
                // one relaxed add carries this tile's counts AND its arrival, so the 64th arrival holds the super-tile's totals
                // in the value the add returns: nothing else has to be visible to it, no fence is needed
                acc = __hip_atomic_fetch_add(&q.supAcc[cur][sup], (1ull << 40) | ((unsigned long long)aggE << 20) | (unsigned long long)aggS,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc = __shfl(acc, 0, 64);
            if ((acc >> 40) == 63ull) {
                // Level 2: the LAST of its 64 tiles to arrive closes a super-tile, at once and whichever tile it is: publish the
                // total, resolve the inclusive prefix by decoupled look-back over the earlier super-tile words (<= a few hundred;
                // waits only on smaller ids) and publish it.
                const uint32_t totE = (uint32_t)((acc >> 20) & 0xfffffull) + aggE, totS = (uint32_t)(acc & 0xfffffull) + aggS;
                uint32_t preE = 0, preS = 0;
                if (sup > 0) {
                    if (lane == 0) __hip_atomic_store(&super[cur][sup], tile_pack(kTileAgg, totE, totS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int top = (int)sup - 1;
                    for (;;) {
                        const int tt = top - lane;
                        unsigned long long v = kTilePrefix;
                        if (tt >= 0) v = wait_word(&super[cur][tt], 0ull, q.fault, 0x30000000 | tt);
                        const unsigned long long isPre = __ballot((v >> 62) == 2ull);
                        const int stop = isPre ? __ffsll((long long)isPre) - 1 : 64;
                        uint32_t e = lane <= stop ? tile_ext(v) : 0u, s2 = lane <= stop ? tile_sh(v) : 0u;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) { e += __shfl_xor(e, off, 64); s2 += __shfl_xor(s2, off, 64); }
                        preE += e; preS += s2;
                        if (isPre) break;
                        top -= 64;
                    }
                }
                if (lane == 0) __hip_atomic_store(&super[cur][sup], tile_pack(kTilePrefix, preE + totE, preS + totS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef RT355_SHADE_TIMING
        const long long c1 = wall_clock64();
#endif
        const uint32_t nextTile = sTicket * (uint32_t)kTicketClasses + cls;
        TileIn inNext;
        fetch(nextTile, inNext);
        if (pend) drain(pTile, par ^ 1, pEm, pSm, pExt, pSh);
        else __syncthreads();   // everybody has read sTicket before thread 0 takes the next one
        in = inNext;
#ifdef RT355_SHADE_TIMING
        tShade += c1 - c0; tDrain += wall_clock64() - c1; nT++;
#endif
        // park this tile's extension ray (its shadow record went to sSh*[par] while shading)
        if (extValid) { sExtO[threadIdx.x] = ext.O; sExtD[threadIdx.x] = ext.D; sExtI[threadIdx.x] = ext.inten; sExtM[threadIdx.x] = extMeta; }
        pend = true; pTile = tile; pEm = em; pSm = sm; pExt = extValid; pSh = shValid;
        tile = nextTile;
    }
    if (pend) {
        __syncthreads();   // the parked extension rays of the last tile
        drain(pTile, par ^ 1, pEm, pSm, pExt, pSh);
    }
#ifdef RT355_SHADE_TIMING
    if (threadIdx.x == 0 && (blockIdx.x & 127) == 5 && bounce <= 1)
        printf("shade b%d block %d tiles %d: shade %lld drain %lld total %lld (x10 ns)\n", bounce, blockIdx.x, nT, tShade, tDrain, wall_clock64() - tBegin);
#endif
}

// ------------------------------------------------------------------ k_connect: any-hit over shadow rays [nShadow[b0], nShadow[b1+1])
template <int ACCEL, int LAYOUT>
__global__ __launch_bounds__(kBlock) void k_connect(DevScene sc, DevQueues q, int b0, int b1)
{
    extern __shared__ uint32_t stk[];
    const int first = q.nShadow[b0], n = q.nShadow[b1 + 1] - first;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    WorkCtr wc = { 0, 0, 0, 0 };
    uint32_t rays = 0;
    if (i < n) {
        const float4 a = q.sA[first + i], b = q.sB[first + i];
        TRay r;
        r.ox = a.x; r.oy = a.y; r.oz = a.z; r.dx = b.x; r.dy = b.y; r.dz = b.z;
        r.rx = 1.0f / b.x; r.ry = 1.0f / b.y; r.rz = 1.0f / b.z;
        r.t = a.w; r.prim = -1; r.u = 0.0f; r.v = 0.0f;
        rays = 1;
        if (traverse_tlas<ACCEL, LAYOUT, true>(sc, r, stk, wc) == -1) q.sC[first + i] = splat(0.0f); // occluded: contributes nothing
    }
    flush_counters(q.ctrConnect, rays, wc, stk);
}

// Ordered accumulation of one bounce's shadow contributions: within a bounce every pixel owns
// at most one shadow ray, so the plain add is race-free; bounces are launched in order, which
// reproduces the accumulation order of schedule S1 bit for bit.
__global__ __launch_bounds__(kBlock) void k_accumulate(DevQueues q, int bounce)
{
    const int first = q.nShadow[bounce], n = q.nShadow[bounce + 1] - first;
    // grid-stride: the launch is sized for a typical queue, not for the worst case of one shadow ray per pixel
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 c = ldnt4(&q.sC[first + i]);                                 // last use
        if (c.x == 0.0f && c.y == 0.0f && c.z == 0.0f && c.w == 0.0f) continue;
        const int pix = __float_as_int(ldnt4(&q.sB[first + i]).w);
        q.accum[pix] = add4(q.accum[pix], c);
    }
}

// ------------------------------------------------------------------ k_focus (wavefront.cl:203-224)
template <int ACCEL>
__global__ void k_focus(DevScene sc, RtCamera cam, int x, int y, int W, int H, float* out)
{
    extern __shared__ uint32_t stk[];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float u = (float)x * (1.0f / (float)W), v = (float)y * (1.0f / (float)H); // camera.cl:48-55
    float4 P = add4(add4(ld4(cam.topLeft), muls(ld4(cam.horizontal), u)), muls(ld4(cam.vertical), v));
    float4 D = normalize4(sub4(P, ld4(cam.origin)));
    float4 O = ld4(cam.origin);
    TRay r;
    r.ox = O.x; r.oy = O.y; r.oz = O.z; r.dx = D.x; r.dy = D.y; r.dz = D.z;
    r.rx = 1.0f / D.x; r.ry = 1.0f / D.y; r.rz = 1.0f / D.z;
    r.t = kFar; r.prim = -1; r.u = r.v = 0.0f;
    WorkCtr wc = { 0, 0, 0, 0 };
    traverse_tlas<ACCEL, 0, false>(sc, r, stk, wc);
    *out = r.t;
}

// ------------------------------------------------------------------ post-processing (src/cl/postproc.cl)
// The reference runs prep -> [vignetting] -> [gammaCorr] -> [chromatic] -> saveImage as separate full-frame
// passes over two swap buffers (renderer.cpp:95-124, 303-308).  They are per-pixel streaming operations (chromatic
// also reads the left neighbour), so one kernel evaluates the chain for a pixel and, when chromatic aberration is on,
// for its left neighbour: 16 B read + 16 B written per pixel instead of up to 5 x 32 B.
struct PostParams { float invFrames, vignette, gamma, chromatic; int width, height; };
RT_FORCEINLINE float3 post_chain(const float4* accum, int idx, const PostParams& pp)
{
    const float4 a = accum[idx];
    float3 c = make_float3(fminf(a.x * pp.invFrames, 1.0f), fminf(a.y * pp.invFrames, 1.0f), fminf(a.z * pp.invFrames, 1.0f)); // prep, postproc.cl:65-75
    if (pp.vignette > 0) { // postproc.cl:18-32
        const int x = idx % pp.width, y = idx / pp.width;
        const float px = (float)x / (float)pp.width - 0.5f, py = (float)y / (float)pp.height - 0.5f;
        // length(float2) as ROCm's OpenCL library computes it for the reference's kernel (read from its gfx950 code object):
        // fma(y, y, x*x), rescaled by 2^+-n when tiny / infinite, and the HARDWARE square root v_sqrt_f32 (1 ulp, not the correctly
        // rounded one).  Post-processing is compared with the reference's own postproc kernels bit for bit (tests/test_gpu_reference.py),
        // so k_postproc issues the same instruction; the CPU oracle has no such instruction and is held to 2 ulp on vignetted images.
        float d = __fmaf_rn(py, py, px * px);
        float len;
        if (d < 1.17549435e-38f) { const float sx = px * 0x1p+86f, sy = py * 0x1p+86f; len = __builtin_amdgcn_sqrtf(__fmaf_rn(sy, sy, sx * sx)) * 0x1p-86f; }
        else if (d == INFINITY) { const float sx = px * 0x1p-65f, sy = py * 0x1p-65f; len = __builtin_amdgcn_sqrtf(__fmaf_rn(sy, sy, sx * sx)) * 0x1p+65f; }
        else len = __builtin_amdgcn_sqrtf(d);
        float t = fminf(fmaxf(len, 0.0f), 1.0f);                                 // smoothstep(0,1,len) = t*t*fma(t,-2,3)
        float vig = 1 - (t * t) * __fmaf_rn(t, -2.0f, 3.0f);
        c = make_float3(__fmaf_rn(c.x * vig - c.x, pp.vignette, c.x), __fmaf_rn(c.y * vig - c.y, pp.vignette, c.y),
                        __fmaf_rn(c.z * vig - c.z, pp.vignette, c.z));           // mix(a,b,s) = fma(b-a, s, a)
    }
    if (pp.gamma != 1.0f) c = make_float3(powf(c.x, pp.gamma), powf(c.y, pp.gamma), powf(c.z, pp.gamma)); // postproc.cl:34-40
    return c;
}
__global__ __launch_bounds__(kBlock) void k_postproc(const float4* accum, float4* out, uchar4* rgba8, PostParams pp)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= pp.width * pp.height) return;
    float3 c = post_chain(accum, idx, pp);
    if (pp.chromatic > 0 && idx % pp.width != 0) { // postproc.cl:42-63
        const float3 prev = post_chain(accum, idx - 1, pp);
        const float o = pp.chromatic;
        c = make_float3(c.x, c.y * (1 - o) + prev.y * o, c.z * (1 - 2 * o) + prev.z * 2 * o);
    }
    // display (target is RGBA8) + saveImage: min(color, 1) (postproc.cl:7-16,77-86); bytes as SaveImageF (template.cpp:1629-1644)
    const float4 o4 = mk4(fminf(c.x, 1.0f), fminf(c.y, 1.0f), fminf(c.z, 1.0f), 1.0f);
    if (out) out[idx] = o4;
    if (rgba8) rgba8[idx] = make_uchar4((unsigned char)(o4.x * 255), (unsigned char)(o4.y * 255), (unsigned char)(o4.z * 255), 255);
}

// ------------------------------------------------------------------ debug import/export (parity tests)
__global__ void k_export_rays(DevScene sc, DevQueues q, int bounce, RtRay* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.nRays[bounce]) return;
    RtRay r;
    const int set = bounce & 1;
    const float4 O = q.O[set][i], D = q.D[set][i], hit = q.hit[i], inten = q.inten[set][i];
    const uint2 meta = q.meta[set][i];
    float4 rD = mk4(1.0f / D.x, 1.0f / D.y, 1.0f / D.z, 1.0f / D.w);
    float4 I = splat(0.0f), N = splat(0.0f);
    const int prim = __float_as_int(hit.y);
    if (prim != -1) {
        I = add4(O, muls(D, hit.x));
        N = prim_normal(sc.prims + prim, I);
        if (dot4(N, neg4(D)) < 0) N = muls(N, -1.0f);
    }
    *reinterpret_cast<float4*>(&r.O) = O; *reinterpret_cast<float4*>(&r.D) = D; *reinterpret_cast<float4*>(&r.rD) = rD;
    *reinterpret_cast<float4*>(&r.N) = N; *reinterpret_cast<float4*>(&r.I) = I; *reinterpret_cast<float4*>(&r.intensity) = inten;
    r.t = hit.x; r.primIdx = prim; r.bounces = (int)(meta.y & kMetaBounceMask); r.pixelIdx = (int)meta.x;
    r.inside = (meta.y & kMetaInside) ? 1 : 0; r.lastSpecular = (meta.y & kMetaLastSpec) ? 1 : 0;
    r._pad0[0] = r._pad0[1] = 0; r.u = hit.z; r.v = hit.w; r._pad1 = 0;
    out[i] = r;
}
__global__ void k_import_rays(DevQueues q, const RtRay* in, int n, int set)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtRay r = in[i];
    q.O[set][i] = ld4(r.O); q.D[set][i] = ld4(r.D); q.inten[set][i] = ld4(r.intensity);
    q.hit[i] = mk4(r.t, __int_as_float(r.primIdx), r.u, r.v);
    q.meta[set][i] = make_uint2((uint32_t)r.pixelIdx, (uint32_t)r.bounces | (r.inside ? kMetaInside : 0u) | (r.lastSpecular ? kMetaLastSpec : 0u));
}
__global__ void k_set_count(int32_t* p, int32_t v) { *p = v; }

} // namespace rt355dev
