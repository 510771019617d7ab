// gather_probe: does a cooperative (quad-coalesced) fetch of per-lane 64-byte records beat four 16-byte loads per lane on MI355X?
// Mimics the node event of k_trace_persist: a dependent chain of random 64-B record fetches out of a 32 MB table, ~50 VALU ops per step.
//   A: lane l issues 4 x global_load_dwordx4 at its own record (what the kernels do today; 4 x 64 line requests per wave step)
//   B: in load j, lane l fetches chunk (l & 3) of the record wanted by lane j*16 + (l >> 2); the quad's 64 B are contiguous, so the
//      texture addresser sees 16 line requests per load; the chunks go through a 4 KB/wave LDS transposition back to their owners.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o gpurun_out/gather_probe ; run: gather_probe [extraLdsKB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float work(float4 q0, float4 q1, float4 q2, float ox, float oy, float oz, float rx, float ry, float rz)
{
    // two slab tests, same op mix as the traversal step
    float a0 = (q0.x - ox) * rx, a1 = (q0.w - ox) * rx, b0 = (q0.y - oy) * ry, b1 = (q1.x - oy) * ry, c0 = (q0.z - oz) * rz, c1 = (q1.y - oz) * rz;
    float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
    float d0 = (q1.z - ox) * rx, d1 = (q2.y - ox) * rx, e0 = (q1.w - oy) * ry, e1 = (q2.z - oy) * ry, f0 = (q2.x - oz) * rz, f1 = (q2.w - oz) * rz;
    float un = fmaxf(fmaxf(fminf(d0, d1), fminf(e0, e1)), fminf(f0, f1)), uf = fminf(fminf(fmaxf(d0, d1), fmaxf(e0, e1)), fmaxf(f0, f1));
    return (tf >= tn ? tn : 1e30f) - (uf >= un ? un : 1e30f);
}

__global__ __launch_bounds__(256) void gatherA(const float4* __restrict__ tab, int steps, float* out, uint32_t mask)
{
    extern __shared__ uint32_t pad[];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = ((uint32_t)gid * 2654435761u >> 13) & mask;
    float acc = 0, ox = gid * 1e-6f, oy = 0.5f, oz = 0.25f, rx = 1.5f, ry = -0.7f, rz = 0.9f;
    for (int s = 0; s < steps; s++) {
        const float4* p = tab + (size_t)idx * 4;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        const float w = work(q0, q1, q2, ox, oy, oz, rx, ry, rz);
        acc += w;
        idx = w > 0.0f ? __float_as_uint(q3.x) : __float_as_uint(q3.y);
    }
    out[gid] = acc + (float)pad[0] * 0.0f;
}

__global__ __launch_bounds__(256) void gatherB(const float4* __restrict__ tab, int steps, float* out, uint32_t mask)
{
    extern __shared__ float4 xb[];   // [wave][chunk][lane]
    const int gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* x = xb + wave * 256;
    uint32_t idx = ((uint32_t)gid * 2654435761u >> 13) & mask;
    float acc = 0, ox = gid * 1e-6f, oy = 0.5f, oz = 0.25f, rx = 1.5f, ry = -0.7f, rz = 0.9f;
    const int g = lane >> 2, c = lane & 3;
    for (int s = 0; s < steps; s++) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t ni = (uint32_t)__shfl((int)idx, j * 16 + g, 64);
            v[j] = tab[(size_t)ni * 4 + c];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) x[c * 64 + j * 16 + g] = v[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 q0 = x[lane], q1 = x[64 + lane], q2 = x[128 + lane], q3 = x[192 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float w = work(q0, q1, q2, ox, oy, oz, rx, ry, rz);
        acc += w;
        idx = w > 0.0f ? __float_as_uint(q3.x) : __float_as_uint(q3.y);
    }
    out[gid] = acc;
}

// C: TWO lanes per chain.  The record is laid out as two 32-byte halves (child box + child entry each); the even lane of a pair fetches
// and tests the first half, the odd lane the second (2 x global_load_dwordx4 per lane, both lanes of a pair in the same 64-B line), and
// the two results are exchanged by DPP quad_perm [1,0,3,2].  32 chains per wave, half the slab work per lane.
__device__ __forceinline__ float swap1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ uint32_t swap1u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }

__global__ __launch_bounds__(256) void gatherC(const float4* __restrict__ tab2, int steps, float* out, uint32_t mask)
{
    extern __shared__ uint32_t pad[];
    const int tid = blockIdx.x * 256 + threadIdx.x, gid = tid >> 1, half = tid & 1;
    uint32_t idx = ((uint32_t)gid * 2654435761u >> 13) & mask;
    float acc = 0, ox = gid * 1e-6f, oy = 0.5f, oz = 0.25f, rx = 1.5f, ry = -0.7f, rz = 0.9f;
    for (int s = 0; s < steps; s++) {
        const float4* p = tab2 + (size_t)idx * 4 + half * 2;
        const float4 q0 = p[0], q1 = p[1];          // (lo.xyz, hi.x) (hi.yz, entry, -)
        const float a0 = (q0.x - ox) * rx, a1 = (q0.w - ox) * rx, b0 = (q0.y - oy) * ry, b1 = (q1.x - oy) * ry, c0 = (q0.z - oz) * rz, c1 = (q1.y - oz) * rz;
        const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
        const float mine = tf >= tn ? tn : 1e30f, other = swap1(mine);
        const float w = half ? other - mine : mine - other;
        acc += w;
        const uint32_t e = __float_as_uint(q1.z), eo = swap1u(e);
        idx = (w > 0.0f) == (half == 0) ? e : eo;
    }
    if (!half) out[gid] = acc + (float)pad[0] * 0.0f;
}

// D: one chain per lane as in A, but the lanes of a pair fetch TOGETHER: in load group I both lanes read the record of the EVEN lane's
// chain (even lane: half 0, odd lane: half 1 - one 64-byte line per pair), in group II the record of the ODD lane's chain.  Each lane
// tests "its" child for both chains (with a resident copy of the partner's ray), then the foreign result is handed over by DPP.
// 64 chains per wave, the slab arithmetic of A, 4 loads per lane as in A - but every load instruction touches 32 lines instead of 64.
__device__ __forceinline__ uint32_t bcastEven(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0, 0xf, 0xf, true); }  // quad_perm [0,0,2,2]
__device__ __forceinline__ uint32_t bcastOdd(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5, 0xf, 0xf, true); }   // quad_perm [1,1,3,3]
__device__ __forceinline__ float halfSlab(float4 q0, float4 q1, float ox, float oy, float oz, float rx, float ry, float rz)
{
    const float a0 = (q0.x - ox) * rx, a1 = (q0.w - ox) * rx, b0 = (q0.y - oy) * ry, b1 = (q1.x - oy) * ry, c0 = (q0.z - oz) * rz, c1 = (q1.y - oz) * rz;
    const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
    return tf >= tn ? tn : 1e30f;
}
__global__ __launch_bounds__(256) void gatherD(const float4* __restrict__ tab2, int steps, float* out, uint32_t mask)
{
    extern __shared__ uint32_t pad[];
    const int gid = blockIdx.x * 256 + threadIdx.x, half = gid & 1;
    uint32_t idx = ((uint32_t)gid * 2654435761u >> 13) & mask;
    float acc = 0;
    const float ox = gid * 1e-6f, oy = 0.5f, oz = 0.25f, rx = 1.5f, ry = -0.7f, rz = 0.9f;
    const float oxE = __uint_as_float(bcastEven(__float_as_uint(ox))), oxO = __uint_as_float(bcastOdd(__float_as_uint(ox)));   // resident copies of both rays
    const float4* base = tab2 + half * 2;
    for (int s = 0; s < steps; s++) {
        const uint32_t idxE = bcastEven(idx), idxO = bcastOdd(idx);
        const float4* pE = base + (size_t)idxE * 4;
        const float4* pO = base + (size_t)idxO * 4;
        const float4 e0 = pE[0], e1 = pE[1], o0 = pO[0], o1 = pO[1];
        const float rE = halfSlab(e0, e1, oxE, oy, oz, rx, ry, rz);      // my child of the even chain's record
        const float rO = halfSlab(o0, o1, oxO, oy, oz, rx, ry, rz);      // my child of the odd chain's record
        const uint32_t enE = __float_as_uint(e1.z), enO = __float_as_uint(o1.z);
        // hand the foreign result to its owner: the even lane sends (rO, enO), the odd lane (rE, enE)
        const float got = swap1(half ? rE : rO);
        const uint32_t gotE = swap1u(half ? enE : enO);
        const float mine = half ? rO : rE;
        const uint32_t mineE = half ? enO : enE;
        const float w = half ? got - mine : mine - got;                  // child 1 minus child 2
        acc += w;
        idx = (w > 0.0f) == (half == 0) ? mineE : gotE;
    }
    out[gid] = acc + (float)pad[0] * 0.0f;
}

int main(int argc, char** argv)
{
    const int nrec = 1 << (argc > 1 ? atoi(argv[1]) : 19), steps = 64;   // table = nrec * 64 B
    printf("table %d records = %.2f MB\n", nrec, nrec * 64.0 / 1048576.0);
    std::vector<float> h((size_t)nrec * 16);
    uint32_t s = 12345u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    for (int i = 0; i < nrec; i++) {
        float* r = &h[(size_t)i * 16];
        for (int k = 0; k < 12; k++) r[k] = (float)(rnd() & 0xffff) / 65536.0f;
        uint32_t a = rnd() & (nrec - 1), b = rnd() & (nrec - 1);
        memcpy(&r[12], &a, 4); memcpy(&r[13], &b, 4); r[14] = r[15] = 0;
    }
    // layout for C: half 0 = (q0.x q0.y q0.z | q0.w q1.x q1.y -> lo.xyz, hi.xyz), entry a; half 1 likewise from (q1.z q1.w q2.x | q2.y q2.z q2.w), entry b
    std::vector<float> h2((size_t)nrec * 16);
    for (int i = 0; i < nrec; i++) {
        const float* r = &h[(size_t)i * 16]; float* d = &h2[(size_t)i * 16];
        // A's work(): box 0 uses x: r0,r3  y: r1,r4  z: r2,r5 ; box 1 uses x: r6,r9  y: r7,r10  z: r8,r11
        d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3]; d[4] = r[4]; d[5] = r[5]; d[6] = r[12]; d[7] = 0;
        d[8] = r[6]; d[9] = r[7]; d[10] = r[8]; d[11] = r[9]; d[12] = r[10]; d[13] = r[11]; d[14] = r[13]; d[15] = 0;
    }
    float4* tab; float* out; float4* tab2;
    CHK(hipMalloc(&tab2, h2.size() * 4)); CHK(hipMemcpy(tab2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> rc(256 * 256 * 8);
    const int maxThreads = 256 * 256 * 8;
    CHK(hipMalloc(&tab, h.size() * 4)); CHK(hipMalloc(&out, (size_t)maxThreads * 4));
    CHK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::vector<float> ra(maxThreads), rb(maxThreads);
    for (int perCU = 4; perCU <= 8; perCU += 3) {
        // LDS per block chosen so that exactly perCU blocks fit a CU (160 KB), like the traversal stack does
        const size_t lds = (size_t)(160 * 1024 / perCU) & ~(size_t)1023;
        if (lds > 64 * 1024) continue;
        const int blocks = 256 * perCU, total = blocks * 256;
        for (int which = 0; which < 2; which++) {
            if (which == 1 && lds < 16 * 1024) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(gatherA, dim3(blocks), dim3(256), lds, 0, tab, steps, out, (uint32_t)(nrec - 1));
                else hipLaunchKernelGGL(gatherB, dim3(blocks), dim3(256), lds, 0, tab, steps, out, (uint32_t)(nrec - 1));
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            CHK(hipMemcpy(which ? rb.data() : ra.data(), out, (size_t)total * 4, hipMemcpyDeviceToHost));
            printf("%s blocks/CU %d (lds %zu KB): %.3f ms, %.2f G records/s\n", which ? "B coop" : "A 4xld", perCU, lds >> 10, best,
                   (double)total * steps / best / 1e6);
        }
        {
            // C with the same number of THREADS (half the chains per launch) and with the same number of CHAINS (twice the workgroups)
            for (int mult = 1; mult <= 2; mult++) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    CHK(hipEventRecord(e0));
                    hipLaunchKernelGGL(gatherC, dim3(blocks * mult), dim3(256), lds / mult, 0, tab2, steps, out, (uint32_t)(nrec - 1));
                    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                const int chains = total * mult / 2;
                CHK(hipMemcpy(rc.data(), out, (size_t)chains * 4, hipMemcpyDeviceToHost));
                int badc = 0; for (int i = 0; i < chains && i < total; i++) if (ra[i] != rc[i]) badc++;
                printf("C pair blocks/CU %d x%d: %.3f ms, %.2f G records/s  mismatches vs A %d\n", perCU, mult, best, (double)chains * steps / best / 1e6, badc);
            }
        }
        {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(gatherD, dim3(blocks), dim3(256), lds, 0, tab2, steps, out, (uint32_t)(nrec - 1));
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            CHK(hipMemcpy(rc.data(), out, (size_t)total * 4, hipMemcpyDeviceToHost));
            int badd = 0; for (int i = 0; i < total; i++) if (ra[i] != rc[i]) badd++;
            printf("D pair-fetch blocks/CU %d: %.3f ms, %.2f G records/s  mismatches vs A %d\n", perCU, best, (double)total * steps / best / 1e6, badd);
        }
        int bad = 0;
        if (lds >= 16 * 1024) for (int i = 0; i < total; i++) if (ra[i] != rb[i]) bad++;
        printf("   mismatches A vs B: %d\n", bad);
    }
    return 0;
}
