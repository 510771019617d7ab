/* wide_anyhit_lab.c - CPU experiment (not part of the product): a compressed WIDE tree for shadow rays, costed in vector-memory requests.
 * Input: the file tools/lab/anyhit_lab.py writes (reference BVH2 arrays + the shadow rays of a bench frame).
 *
 * The reference decides "occluded" by walking its BVH2 with float slab tests.  Slab tests are monotone in the box (IEEE subtraction and
 * multiplication are monotone, min / max exact): a ray that passes the test of a box passes the test of every box containing it, so a leaf
 * is reached by the reference iff ITS OWN box passes.  Any conservative hierarchy above the leaves therefore gives the same answer as
 * long as a leaf's exact box is tested before its triangles.  Built here: the BVH2 collapsed into nodes of up to WIDTH children (the
 * child of largest area is opened first), child boxes quantised to 8 bits per plane against the node's box (rounded outwards, checked
 * after decoding with the same float operations the traversal uses).  Reported: that every ray gets the reference's answer, visits per
 * ray, and 16-byte requests per ray under the record sizes given below, against the BVH2 any-hit traversal the product runs today. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float mn[4], mx[4]; uint32_t first, count; uint32_t pad[2]; } Node;      /* RtBVHNode2, 48 B */
typedef struct { float v0[4], v1[4], v2[4]; float rest[16]; int32_t type, mat; float area; int32_t pad; } Prim; /* 128 B */
typedef struct { float o[3], tmax, d[3]; int32_t pix; } SRay;
#define MAXW 8
typedef struct { float lo[3], scale[3]; uint8_t qlo[MAXW][3], qhi[MAXW][3]; int32_t child[MAXW]; /* >= 0: wide node, < 0: ~bvh2 leaf */ int n; } Wide;

static const Node* N; static const Prim* P; static const uint32_t* IDX;
static Wide* WN; static int nWide = 0, capWide = 0, WIDTH = 8;

static float area_of(const Node* n) { float x = n->mx[0] - n->mn[0], y = n->mx[1] - n->mn[1], z = n->mx[2] - n->mn[2]; return x * y + y * z + z * x; }
static int slab(const float* o, const float* r, float t, const float* mn, const float* mx, float* exitT)
{
    float tx1 = (mn[0] - o[0]) * r[0], tx2 = (mx[0] - o[0]) * r[0];
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    float ty1 = (mn[1] - o[1]) * r[1], ty2 = (mx[1] - o[1]) * r[1];
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (mn[2] - o[2]) * r[2], tz2 = (mx[2] - o[2]) * r[2];
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    *exitT = tmax;
    return tmax >= tmin && tmin < t && tmax > 0;
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
/* decoded planes, with the float operations the traversal would use */
static float dec(const Wide* w, int axis, uint8_t q) { return w->lo[axis] + (float)q * w->scale[axis]; }

static int build(uint32_t root)
{
    if (nWide == capWide) { capWide = capWide ? capWide * 2 : 1024; WN = realloc(WN, sizeof(Wide) * (size_t)capWide); }
    const int me = nWide++;
    uint32_t kids[MAXW]; int nk = 2;
    kids[0] = N[root].first; kids[1] = N[root].first + 1;
    for (;;) {                                           /* open the interior child of largest area until the node is full */
        int best = -1; float ba = -1;
        for (int k = 0; k < nk; k++) if (N[kids[k]].count == 0 && area_of(&N[kids[k]]) > ba) { ba = area_of(&N[kids[k]]); best = k; }
        if (best < 0 || nk == WIDTH) break;
        const uint32_t c = kids[best];
        kids[best] = N[c].first; kids[nk++] = N[c].first + 1;
    }
    Wide w; memset(&w, 0, sizeof w); w.n = nk;
    for (int a = 0; a < 3; a++) {
        w.lo[a] = N[root].mn[a];
        const float ext = N[root].mx[a] - N[root].mn[a];
        w.scale[a] = ext > 0 ? ext / 255.0f : 0.0f;
        while (w.scale[a] > 0 && w.lo[a] + 255.0f * w.scale[a] < N[root].mx[a]) w.scale[a] = nextafterf(w.scale[a], INFINITY);   /* the last code must reach the node's upper plane */
        for (int k = 0; k < nk; k++) {
            const float lo = N[kids[k]].mn[a], hi = N[kids[k]].mx[a];
            int ql = w.scale[a] > 0 ? (int)floorf((lo - w.lo[a]) / w.scale[a]) : 0, qh = w.scale[a] > 0 ? (int)ceilf((hi - w.lo[a]) / w.scale[a]) : 0;
            if (ql < 0) ql = 0; if (ql > 255) ql = 255; if (qh < 0) qh = 0; if (qh > 255) qh = 255;
            while (ql > 0 && dec(&w, a, (uint8_t)ql) > lo) ql--;                 /* rounded outwards AFTER decoding */
            while (qh < 255 && dec(&w, a, (uint8_t)qh) < hi) qh++;
            if (dec(&w, a, (uint8_t)ql) > lo || dec(&w, a, (uint8_t)qh) < hi) { fprintf(stderr, "quantisation cannot cover a child box\n"); exit(3); }
            w.qlo[k][a] = (uint8_t)ql; w.qhi[k][a] = (uint8_t)qh;
        }
    }
    for (int k = 0; k < nk; k++) w.child[k] = N[kids[k]].count > 0 ? ~(int32_t)kids[k] : 0;
    WN[me] = w;
    for (int k = 0; k < nk; k++) if (N[kids[k]].count == 0) { const int c = build(kids[k]); WN[me].child[k] = c; }
    return me;
}
/* any-hit over the wide tree: children that pass their (decoded) box are visited in order of LATER exit first */
static int trace_wide(const SRay* s, long* wide, long* leafBoxes, long* tris)
{
    const float r[3] = { 1 / s->d[0], 1 / s->d[1], 1 / s->d[2] };
    int32_t stack[256]; int sp = 0; int32_t cur = 0;
    for (;;) {
        if (cur < 0) {                                   /* a reference leaf: its EXACT box first, then its triangles */
            const Node* n = &N[~cur]; float x;
            (*leafBoxes)++;
            if (slab(s->o, r, s->tmax, n->mn, n->mx, &x))
                for (uint32_t i = 0; i < n->count; i++) { (*tris)++; if (tri_hit(s->o, s->d, s->tmax, &P[IDX[n->first + i]])) return 1; }
        } else {
            const Wide* w = &WN[cur];
            (*wide)++;
            int32_t hit[MAXW]; float ex[MAXW]; int nh = 0;
            for (int k = 0; k < w->n; k++) {
                float mn[3], mx[3], x;
                for (int a = 0; a < 3; a++) { mn[a] = dec(w, a, w->qlo[k][a]); mx[a] = dec(w, a, w->qhi[k][a]); }
                if (slab(s->o, r, s->tmax, mn, mx, &x)) { int j = nh++; while (j > 0 && ex[j - 1] > x) { ex[j] = ex[j - 1]; hit[j] = hit[j - 1]; j--; } ex[j] = x; hit[j] = w->child[k]; }
            }
            for (int k = 0; k < nh; k++) stack[sp++] = hit[k];          /* ascending exit distance: the latest exit is popped first */
        }
        if (!sp) return 0;
        cur = stack[--sp];
    }
}
/* the product's any-hit traversal of the BVH2 (later exit first) */
static int trace_bvh2(const SRay* s, long* nodes, long* tris)
{
    const float r[3] = { 1 / s->d[0], 1 / s->d[1], 1 / s->d[2] };
    uint32_t stack[128]; int sp = 0; uint32_t node = 0;
    for (;;) {
        const Node* n = &N[node];
        if (n->count > 0) {
            for (uint32_t i = 0; i < n->count; i++) { (*tris)++; if (tri_hit(s->o, s->d, s->tmax, &P[IDX[n->first + i]])) return 1; }
            if (!sp) return 0;
            node = stack[--sp]; continue;
        }
        (*nodes)++;
        uint32_t c1 = n->first, c2 = c1 + 1; float x1, x2;
        int h1 = slab(s->o, r, s->tmax, N[c1].mn, N[c1].mx, &x1), h2 = slab(s->o, r, s->tmax, N[c2].mn, N[c2].mx, &x2);
        if (h1 && h2) { if (x2 > x1) { node = c2; stack[sp++] = c1; } else { node = c1; stack[sp++] = c2; } }
        else if (h1 || h2) node = h1 ? c1 : c2;
        else { if (!sp) return 0; node = stack[--sp]; }
    }
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int32_t hdr[4]; if (!f || fread(hdr, 4, 4, f) != 4) return 1;
    Node* n = malloc(sizeof(Node) * hdr[0]); Prim* p = malloc(sizeof(Prim) * hdr[1]); uint32_t* ix = malloc(4 * hdr[2]); SRay* s = malloc(sizeof(SRay) * hdr[3]);
    if (fread(n, sizeof(Node), hdr[0], f) != (size_t)hdr[0] || fread(p, sizeof(Prim), hdr[1], f) != (size_t)hdr[1] || fread(ix, 4, hdr[2], f) != (size_t)hdr[2] ||
        fread(s, sizeof(SRay), hdr[3], f) != (size_t)hdr[3]) return 2;
    N = n; P = p; IDX = ix;
    long bn = 0, bt = 0, bocc = 0; char* ref = malloc((size_t)hdr[3]);
    for (int i = 0; i < hdr[3]; i++) { ref[i] = (char)trace_bvh2(&s[i], &bn, &bt); bocc += ref[i]; }
    const double R = hdr[3];
    printf("%d shadow rays, %.1f %% occluded\n", hdr[3], 100.0 * bocc / R);
    printf("BVH2 any-hit (product): %.2f node visits, %.2f triangle tests per ray -> %.1f requests (4 per node record, 3 per triangle record), %.1f round trips\n",
           bn / R, bt / R, (4.0 * bn + 3.0 * bt) / R, (bn + bt) / R);
    for (WIDTH = 4; WIDTH <= 8; WIDTH += 4) {
        nWide = 0; build(0);
        long wv = 0, lb = 0, tt = 0, diff = 0;
        for (int i = 0; i < hdr[3]; i++) diff += trace_wide(&s[i], &wv, &lb, &tt) != ref[i];
        /* record sizes: origin + scale 24 B, 6 B per child box, child base indices 8 B + 1 B per child */
        const int bytes = 24 + 6 * WIDTH + 8 + WIDTH, req = (bytes + 15) / 16;
        printf("width %d, 8-bit boxes: %d wide nodes of %d B (%d requests): %ld rays answered differently; per ray %.2f wide visits, %.2f exact leaf boxes (2 requests), "
               "%.2f triangle tests -> %.1f requests, %.1f round trips\n", WIDTH, nWide, bytes, req, diff, wv / R, lb / R, tt / R,
               ((double)req * wv + 2.0 * lb + 3.0 * tt) / R, (double)(wv + lb + tt) / R);
    }
    return 0;
}
