/* anyhit_lab.c - CPU experiment (not part of the product): how many node records does an any-hit (shadow) ray have to fetch under
 * different traversal orders over the SAME BVH2 and the same slab / triangle arithmetic?  Input: raw arrays dumped by anyhit_lab.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float mn[4], mx[4]; uint32_t first, count; uint32_t pad[2]; } Node;      /* RtBVHNode2, 48 B */
typedef struct { float v0[4], v1[4], v2[4]; float rest[16]; int32_t type, mat; float area; int32_t pad; } Prim; /* 128 B */
typedef struct { float o[3], tmax, d[3]; int32_t pix; } SRay;

static const Node* N; static const Prim* P; static const uint32_t* IDX; static uint32_t* CNT; static float* OCCP;
static uint32_t count_prims(uint32_t i) { const Node* n = &N[i]; uint32_t c = n->count > 0 ? n->count : count_prims(n->first) + count_prims(n->first + 1); CNT[i] = c; return c; }
static float area_of(const Node* n) { float x = n->mx[0] - n->mn[0], y = n->mx[1] - n->mn[1], z = n->mx[2] - n->mn[2]; return x * y + y * z + z * x; }
static float g_exit;
static float slab(const float* o, const float* r, float t, const Node* n)
{
    float tx1 = (n->mn[0] - o[0]) * r[0], tx2 = (n->mx[0] - o[0]) * r[0];
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (n->mn[1] - o[1]) * r[1], ty2 = (n->mx[1] - o[1]) * r[1];
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (n->mn[2] - o[2]) * r[2], tz2 = (n->mx[2] - o[2]) * r[2];
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    g_exit = tmax;
    return (tmax >= tmin && tmin < t && tmax > 0) ? tmin : 1e30f;
}
static int tri_hit(const float* o, const float* d, float tmax, const Prim* p)
{
    float e1[3], e2[3], pv[3], tv[3], qv[3];
    for (int k = 0; k < 3; k++) { e1[k] = p->v1[k] - p->v0[k]; e2[k] = p->v2[k] - p->v0[k]; }
    pv[0] = d[1] * e2[2] - d[2] * e2[1]; pv[1] = d[2] * e2[0] - d[0] * e2[2]; pv[2] = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (fabsf(det) < 1e-8f) return 0;
    float inv = 1 / det;
    for (int k = 0; k < 3; k++) tv[k] = o[k] - p->v0[k];
    float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0 || u > 1) return 0;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
    if (v < 0 || u + v > 1) return 0;
    float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    return !(t > tmax || t < 0);
}
/* order: 0 near-first (reference), 1 far-first, 2 larger area first, 3 smaller-index... 4 = child nearer to the ray END (light) first */
static int trace(const SRay* s, int order, long* nodes, long* tris)
{
    float r[3] = { 1 / s->d[0], 1 / s->d[1], 1 / s->d[2] };
    uint32_t stack[128]; int sp = 0; uint32_t node = 0;
    for (;;) {
        const Node* n = &N[node];
        if (n->count > 0) {
            for (uint32_t i = 0; i < n->count; i++) { (*tris)++; if (tri_hit(s->o, s->d, s->tmax, &P[IDX[n->first + i]])) return 1; }
            if (!sp) return 0;
            node = stack[--sp]; continue;
        }
        (*nodes)++;
        uint32_t c1 = n->first, c2 = c1 + 1;
        float d1 = slab(s->o, r, s->tmax, &N[c1]); float x1 = g_exit; float d2 = slab(s->o, r, s->tmax, &N[c2]); float x2 = g_exit;
        int swap = 0;
        if (order == 6) swap = x1 < x2;            /* later exit first */
        if (order == 7) swap = fminf(x1, s->tmax) - d1 < fminf(x2, s->tmax) - d2;   /* longer chord first */
        if (order == 8) swap = CNT[c1] < CNT[c2];                                   /* more primitives first */
        if (order == 9) swap = CNT[c1] / (area_of(&N[c1]) + 1e-9f) < CNT[c2] / (area_of(&N[c2]) + 1e-9f);   /* denser first */
        if (order == 10) swap = (fminf(x1, s->tmax) - fmaxf(d1, 0)) * CNT[c1] / (area_of(&N[c1]) + 1e-9f) < (fminf(x2, s->tmax) - fmaxf(d2, 0)) * CNT[c2] / (area_of(&N[c2]) + 1e-9f);
        if (order == 11) swap = OCCP[c1] < OCCP[c2];                                /* learned: fraction of visiting rays that found their occluder below this node */
        if (order == 12) swap = (x1 < x2) ? (OCCP[c2] > 0.5f * OCCP[c1]) : !(OCCP[c1] > 0.5f * OCCP[c2]);
        if (order == 0) swap = d1 > d2;
        else if (order == 1) swap = d1 < d2;
        else if (order == 2) swap = area_of(&N[c1]) < area_of(&N[c2]);
        else if (order == 4) {   /* exit distance: the child the ray leaves LATER is nearer the light */
            swap = 0;
            float e[3] = { s->o[0] + s->d[0] * s->tmax, s->o[1] + s->d[1] * s->tmax, s->o[2] + s->d[2] * s->tmax };
            float q1 = 0, q2 = 0;
            for (int k = 0; k < 3; k++) { float c = 0.5f * (N[c1].mn[k] + N[c1].mx[k]) - e[k]; q1 += c * c; c = 0.5f * (N[c2].mn[k] + N[c2].mx[k]) - e[k]; q2 += c * c; }
            swap = q1 > q2;
        } else if (order == 5) swap = N[c1].count == 0 && N[c2].count > 0;   /* leaf first */
        if (swap) { float d = d1; d1 = d2; d2 = d; uint32_t c = c1; c1 = c2; c2 = c; }
        int h1 = d1 < 1e29f, h2 = d2 < 1e29f;
        if (!h1 && !h2) { if (!sp) return 0; node = stack[--sp]; }
        else if (h1) { node = c1; if (h2) stack[sp++] = c2; }
        else node = c2;
    }
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[4]; if (fread(hdr, 4, 4, f) != 4) return 1;
    Node* n = malloc(sizeof(Node) * hdr[0]); Prim* p = malloc(sizeof(Prim) * hdr[1]); uint32_t* ix = malloc(4 * hdr[2]); SRay* s = malloc(sizeof(SRay) * hdr[3]);
    if (fread(n, sizeof(Node), hdr[0], f) != (size_t)hdr[0] || fread(p, sizeof(Prim), hdr[1], f) != (size_t)hdr[1] || fread(ix, 4, hdr[2], f) != (size_t)hdr[2] ||
        fread(s, sizeof(SRay), hdr[3], f) != (size_t)hdr[3]) return 2;
    N = n; P = p; IDX = ix;
    const char* names[] = { "near-first (reference)", "far-first", "larger-area first", "-", "nearer-the-light first", "leaf child first", "later exit first", "longer chord first", "more primitives first", "denser first", "chord x density first", "learned hit rate first", "later exit unless much lower hit rate" };
    CNT = calloc(hdr[0], 4); OCCP = calloc(hdr[0], 4); count_prims(0);
    {   /* "learned" order: per node, the fraction of the rays that visit it (exhaustive traversal, no early exit) for which some primitive below it occludes - from the first half of the rays */
        uint32_t* vis = calloc(hdr[0], 4); uint32_t* occ = calloc(hdr[0], 4);
        for (int i = 0; i < hdr[3] / 2; i++) {
            const SRay* sr = &s[i]; float r[3] = { 1 / sr->d[0], 1 / sr->d[1], 1 / sr->d[2] };
            /* recursive exhaustive traversal with explicit stack of (node, state) is overkill: do a post-order via recursion */
            struct F { uint32_t node; int state; int hit; } st[128]; int sp = 0; st[sp++] = (struct F){ 0, 0, 0 };
            while (sp) {
                struct F* f = &st[sp - 1]; const Node* nd = &N[f->node];
                if (nd->count > 0) { int h = 0; for (uint32_t j = 0; j < nd->count; j++) h |= tri_hit(sr->o, sr->d, sr->tmax, &P[IDX[nd->first + j]]); vis[f->node]++; occ[f->node] += h; int hh = h; sp--; if (sp) st[sp - 1].hit |= hh; continue; }
                if (f->state == 0) { f->state = 1; if (slab(sr->o, r, sr->tmax, &N[nd->first]) < 1e29f) { st[sp++] = (struct F){ nd->first, 0, 0 }; } continue; }
                if (f->state == 1) { f->state = 2; if (slab(sr->o, r, sr->tmax, &N[nd->first + 1]) < 1e29f) { st[sp++] = (struct F){ nd->first + 1, 0, 0 }; } continue; }
                vis[f->node]++; occ[f->node] += f->hit; int hh = f->hit; sp--; if (sp) st[sp - 1].hit |= hh;
            }
        }
        for (int i = 0; i < hdr[0]; i++) OCCP[i] = vis[i] ? (float)occ[i] / vis[i] : 0.0f;
    }
    for (int order = 0; order < 13; order++) {
        if (order == 3) continue;
        long nodes[2] = { 0, 0 }, tris[2] = { 0, 0 }, cnt[2] = { 0, 0 };
        for (int i = 0; i < hdr[3]; i++) { long a = 0, b = 0; int h = trace(&s[i], order, &a, &b); nodes[h] += a; tris[h] += b; cnt[h]++; }
        printf("%-26s occluded %ld rays: %.1f nodes %.1f tris | free %ld rays: %.1f nodes %.1f tris | all: %.2f nodes %.2f tris\n", names[order],
               cnt[1], (double)nodes[1] / cnt[1], (double)tris[1] / cnt[1], cnt[0], (double)nodes[0] / (cnt[0] ? cnt[0] : 1), (double)tris[0] / (cnt[0] ? cnt[0] : 1),
               (double)(nodes[0] + nodes[1]) / hdr[3], (double)(tris[0] + tris[1]) / hdr[3]);
    }
    /* MRU occluder cache over queue order: test the last K distinct occluder triangles first (order 4 traversal otherwise) */
    for (int K = 1; K <= 64; K *= 4) {
        int32_t mru[64]; int nm = 0; long hits = 0, occ = 0, nodes = 0, tris = 0;
        for (int i = 0; i < hdr[3]; i++) {
            int found = -1;
            for (int k = 0; k < nm && found < 0; k++) { tris++; if (tri_hit(s[i].o, s[i].d, s[i].tmax, &P[mru[k]])) found = k; }
            if (found >= 0) { hits++; occ++; int32_t t = mru[found]; memmove(mru + 1, mru, found * sizeof(int32_t)); mru[0] = t; continue; }
            /* full traversal; need the occluder id: re-run trace variant that reports it */
            float r[3] = { 1 / s[i].d[0], 1 / s[i].d[1], 1 / s[i].d[2] };
            uint32_t stack[128]; int sp = 0; uint32_t node = 0; int32_t hitPrim = -1;
            for (;;) {
                const Node* nd = &N[node];
                if (nd->count > 0) {
                    for (uint32_t j = 0; j < nd->count && hitPrim < 0; j++) { tris++; if (tri_hit(s[i].o, s[i].d, s[i].tmax, &P[IDX[nd->first + j]])) hitPrim = (int32_t)IDX[nd->first + j]; }
                    if (hitPrim >= 0 || !sp) break;
                    node = stack[--sp]; continue;
                }
                nodes++;
                uint32_t c1 = nd->first, c2 = c1 + 1;
                float d1 = slab(s[i].o, r, s[i].tmax, &N[c1]), d2 = slab(s[i].o, r, s[i].tmax, &N[c2]);
                if (d1 < d2) { float d = d1; d1 = d2; d2 = d; uint32_t c = c1; c1 = c2; c2 = c; }   /* far first */
                int h1 = d1 < 1e29f, h2 = d2 < 1e29f;
                if (!h1 && !h2) { if (!sp) break; node = stack[--sp]; }
                else if (h1) { node = c1; if (h2) stack[sp++] = c2; }
                else node = c2;
            }
            if (hitPrim >= 0) { occ++; if (nm < K) nm++; memmove(mru + 1, mru, (nm - 1) * sizeof(int32_t)); mru[0] = hitPrim; }
        }
        printf("MRU cache K=%2d: %.1f %% of occluded rays answered by the cache; per ray: %.2f nodes %.2f tris\n", K, 100.0 * hits / occ, (double)nodes / hdr[3], (double)tris / hdr[3]);
    }
    return 0;
}
