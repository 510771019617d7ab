/* rt355_types.h — wire format shared by host code, the C-ABI and the device kernels.
 *
 * Every struct here is layout-compatible (size and field offsets, checked below)
 * with the POD of the same role in the reference's host/device header
 * (reference: src/common.h:3-116, constants src/constants.h:3-31), so that arrays
 * produced by the reference's Scene / BVH2 / BVH4 / TLAS classes can be handed to
 * rt_upload_scene() unchanged.  Offsets are those of SURVEY.md Appendix A.
 *
 * Plain C, no vendor types: float4 is four packed floats, 16-byte aligned.
 */
#ifndef RT355_TYPES_H
#define RT355_TYPES_H

#include <stddef.h>
#include <stdint.h>

#define RT_ALIGNAS(n) __attribute__((aligned(n)))
#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
extern "C" {
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* ---- constants (reference: src/constants.h:7-25) ------------------------- */
#define RT_MAX_BOUNCES      7          /* MAX_BOUNCES                          */
#define RT_EPSILON          0.0001f    /* EPSILON                              */
#define RT_REALLYFAR        1e30f      /* REALLYFAR                            */
#define RT_INVALID          (-1)       /* INVALID                              */
#define RT_PRIM_SPHERE      0
#define RT_PRIM_PLANE       1
#define RT_PRIM_TRIANGLE    2
#define RT_CAM_PROJECTION   0
#define RT_CAM_FISHEYE      1
#define RT_BVH_BINS         8          /* BVH_BINS                             */
#define RT_MIN_LEAF_PRIMS   2          /* MIN_LEAF_PRIMS                       */
#define RT_BVH2_STACK       32         /* src/cl/bvh.cl:15  stack[32]          */
#define RT_BVH4_STACK       64         /* src/cl/bvh.cl:57  stack[64]          */
#define RT_TLAS_STACK       32         /* src/cl/tlas.cl:42 stack[32]          */

typedef struct RT_ALIGNAS(16) RtFloat4 { float x, y, z, w; } RtFloat4;
typedef struct RtFloat2 { float x, y; } RtFloat2;

/* reference: src/common.h:3-12 (Ray, 128 B) */
typedef struct RtRay {
    RtFloat4 O, D, rD;
    RtFloat4 N, I, intensity;
    float    t;
    int32_t  primIdx, bounces, pixelIdx;
    uint8_t  inside, lastSpecular;
    uint8_t  _pad0[2];
    float    u, v;
    uint32_t _pad1;
} RtRay;

/* reference: src/common.h:14-19 (ShadowRay, 96 B) */
typedef struct RtShadowRay {
    RtFloat4 I, L, Nl, intensity, BRDF;
    int32_t  lightIdx, pixelIdx;
    float    dotNL, dist;
} RtShadowRay;

/* reference: src/common.h:21-32 (Material, 80 B) */
typedef struct RtMaterial {
    RtFloat4 color, absorption;
    float    specular, n1, n2;
    uint8_t  isDielectric;
    uint8_t  _pad0[3];
    int32_t  texIdx, texW, texH;
    uint8_t  isLight;
    uint8_t  _pad1[3];
    RtFloat4 emittance;
} RtMaterial;

/* reference: src/common.h:34-51 */
typedef struct RtSphere   { RtFloat4 pos; float r, r2, invr; float _pad; } RtSphere;          /* 32 B  */
typedef struct RtPlane    { RtFloat4 N; float d; float _pad[3]; } RtPlane;                    /* 32 B  */
typedef struct RtTriangle { RtFloat4 v0, v1, v2, N, centroid; RtFloat2 uv0, uv1, uv2;
                            float _pad[2]; } RtTriangle;                                     /* 112 B */

/* reference: src/common.h:53-65 (Primitive, 128 B) */
typedef struct RtPrimitive {
    union { RtTriangle triangle; RtSphere sphere; RtPlane plane; } obj;
    int32_t objType, matIdx;
    float   area;
    uint32_t _pad;
} RtPrimitive;

/* reference: src/common.h:77-83 (Camera, 128 B, passed by value to generate/focus) */
typedef struct RtCamera {
    int32_t  type;
    float    fov, aperture, focalLength;
    RtFloat4 forward, right, up;
    RtFloat4 origin, horizontal, vertical, topLeft;
} RtCamera;

/* reference: src/common.h:85-91 (Settings, 40 B) */
typedef struct RtSettings {
    int32_t numPrimitives, numLights, tracerType, frames, antiAliasing;
    int32_t numInRays, numOutRays, shadowRays;
    int32_t renderBVH;
    float   focalLength;
} RtSettings;

/* reference: src/common.h:93-97 (BVHNode2, 48 B). Leaf iff count>0: first indexes
 * primIdx[]; otherwise the children are nodes first and first+1. */
typedef struct RtBVHNode2 {
    RtFloat4 aabbMin, aabbMax;
    uint32_t first, count;
    uint32_t _pad[2];
} RtBVHNode2;

/* reference: src/common.h:99-103 (BVHNode4, 160 B). Slot unused: first=count=-1;
 * interior child: count==0 (first = node id); leaf child: count>0 (first -> primIdx[]). */
typedef struct RtBVHNode4 {
    RtFloat4 aabbMin[4], aabbMax[4];
    int32_t  first[4], count[4];
} RtBVHNode4;

/* reference: src/common.h:105-109 (BVHInstance, 68 B, 4-byte aligned) */
typedef struct RtBVHInstance {
    uint32_t bvhIdx;
    float    invT[16];
} RtBVHInstance;

/* reference: src/common.h:111-116 (TLASNode, 48 B) */
typedef struct RtTLASNode {
    RtFloat4 aabbMin, aabbMax;
    uint32_t leftRight;   /* lo16 = left child, hi16 = right child, 0 => leaf */
    uint32_t BLASidx;
    uint32_t _pad[2];
} RtTLASNode;

RT_STATIC_ASSERT(sizeof(RtRay) == 128 && offsetof(RtRay, t) == 96 && offsetof(RtRay, primIdx) == 100 &&
                 offsetof(RtRay, pixelIdx) == 108 && offsetof(RtRay, inside) == 112 &&
                 offsetof(RtRay, lastSpecular) == 113 && offsetof(RtRay, u) == 116 && offsetof(RtRay, v) == 120,
                 "Ray layout");
RT_STATIC_ASSERT(sizeof(RtShadowRay) == 96 && offsetof(RtShadowRay, lightIdx) == 80 && offsetof(RtShadowRay, dist) == 92,
                 "ShadowRay layout");
RT_STATIC_ASSERT(sizeof(RtMaterial) == 80 && offsetof(RtMaterial, specular) == 32 && offsetof(RtMaterial, isDielectric) == 44 &&
                 offsetof(RtMaterial, texIdx) == 48 && offsetof(RtMaterial, isLight) == 60 && offsetof(RtMaterial, emittance) == 64,
                 "Material layout");
RT_STATIC_ASSERT(sizeof(RtTriangle) == 112 && offsetof(RtTriangle, uv0) == 80 && offsetof(RtTriangle, uv2) == 96, "Triangle layout");
RT_STATIC_ASSERT(sizeof(RtSphere) == 32 && sizeof(RtPlane) == 32, "Sphere/Plane layout");
RT_STATIC_ASSERT(sizeof(RtPrimitive) == 128 && offsetof(RtPrimitive, objType) == 112 && offsetof(RtPrimitive, matIdx) == 116 &&
                 offsetof(RtPrimitive, area) == 120, "Primitive layout");
RT_STATIC_ASSERT(sizeof(RtCamera) == 128 && offsetof(RtCamera, forward) == 16 && offsetof(RtCamera, topLeft) == 112, "Camera layout");
RT_STATIC_ASSERT(sizeof(RtSettings) == 40 && offsetof(RtSettings, numInRays) == 20 && offsetof(RtSettings, focalLength) == 36,
                 "Settings layout");
RT_STATIC_ASSERT(sizeof(RtBVHNode2) == 48 && offsetof(RtBVHNode2, first) == 32, "BVHNode2 layout");
RT_STATIC_ASSERT(sizeof(RtBVHNode4) == 160 && offsetof(RtBVHNode4, first) == 128 && offsetof(RtBVHNode4, count) == 144, "BVHNode4 layout");
RT_STATIC_ASSERT(sizeof(RtBVHInstance) == 68 && offsetof(RtBVHInstance, invT) == 4, "BVHInstance layout");
RT_STATIC_ASSERT(sizeof(RtTLASNode) == 48 && offsetof(RtTLASNode, leftRight) == 32 && offsetof(RtTLASNode, BLASidx) == 36, "TLASNode layout");

#ifdef __cplusplus
}
#endif
#endif /* RT355_TYPES_H */
