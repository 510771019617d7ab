// pmc_calib: what do the rocprofv3 memory counters count, per byte, for the two access shapes of this library?
//   calib_stream  a coalesced streaming read, 16 B per lane (the ray-queue streams): N bytes known exactly
//   calib_gather  dependent random 64-byte record fetches, 4 x global_load_dwordx4 per lane and record (the traversal's node event),
//                 from a table that fits the L2s (4 MB) / the Infinity Cache only (64 MB): lanes x steps x 64 B known exactly
//   calib_store   a coalesced streaming store, 16 B per lane
//   calib_reread  <0> and <1>: the same 2 MB table read by every workgroup in two back-to-back launches - do the L2s keep it?  (no: both
//                 miss 2 MB / 128 B x 8 XCDs times)
//   calib_valu    nothing but vector arithmetic: 4 independent chains of v_fma_f32 per lane, 8 waves per SIMD; the number of wave-level
//                 vector instructions is known exactly (waves x iterations x 64 + a few) -> the unit of SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU /
//                 SQ_WAVE_CYCLES (quad-cycles?) and the issue peak of the chip (one wave64 instruction per SIMD every 4 clocks)
// Run under `rocprofv3 --pmc <set> --kernel-trace` (tools/r3_pmc.sh); tools/make_traffic.py divides the counter values of these three
// kernels by the byte counts printed here to get bytes-per-count for TCP_TOTAL_CACHE_ACCESSES, TCP_TCC_READ_REQ, TCC_HIT + TCC_MISS,
// FETCH_SIZE and WRITE_SIZE in THIS access pattern (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
// access pattern before trusting an absolute").
// Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ src, size_t n, float* out)
{
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_store(float4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
template <int TAG>   // TAG only gives the two table sizes different kernel names in the counter CSV
__global__ __launch_bounds__(256) void calib_gather(const float4* __restrict__ tab, int steps, float* out, uint32_t mask)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = ((uint32_t)gid * 2654435761u >> 9) & mask;
    float acc = 0;
    for (int s = 0; s < steps; s++) {
        const float4* p = tab + (size_t)idx * 4;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        asm volatile("" : : "v"(q0.x), "v"(q1.x), "v"(q2.x), "v"(q3.x));
        acc += q0.x + q1.y + q2.z;
        idx = __float_as_uint(q3.x) & mask;
    }
    out[gid] = acc;
}

__global__ __launch_bounds__(256) void calib_valu(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

// the slab test's own mix: subtract, multiply, min, max, compare, select - are they all issued at the rate of the fused multiply-add?
__global__ __launch_bounds__(256) void calib_valu_mix(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_sub_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5\n\tv_min_f32 %2, %2, %0\n\tv_max_f32 %3, %3, %1\n\t"
                         "v_cmp_lt_f32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %2, %2, %3\n\tv_and_b32 %3, %3, %2"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

// Does an XCD's L2 keep a read-only table from one launch to the next?  Two launches of the same read (different names), every workgroup
// reads the whole 2 MB table: TCC_MISS of the second against the first.
template <int TAG>
__global__ __launch_bounds__(256) void calib_reread(const float4* __restrict__ tab, int n4, float* out)
{
    float acc = 0;
    for (int rep = 0; rep < 2; rep++)
        for (int i = threadIdx.x; i < n4; i += 256) { const float4 v = tab[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main()
{
    const size_t streamBytes = (size_t)1 << 30;          // 1 GiB
    const size_t n4 = streamBytes / 16;
    float4* buf; float* out;
    CHK(hipMalloc(&buf, streamBytes)); CHK(hipMalloc(&out, sizeof(float) * (1 << 22)));
    CHK(hipMemset(buf, 0, streamBytes));
    const int lanes = 256 * 7 * 256, steps = 64;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_stream, dim3(256 * 8), dim3(256), 0, 0, buf, n4, out);
        hipLaunchKernelGGL(calib_store, dim3(256 * 8), dim3(256), 0, 0, buf, n4);
    }
    CHK(hipDeviceSynchronize());
    for (int big = 0; big < 2; big++) {
        const uint32_t recs = big ? (1u << 20) : (1u << 16);   // 64 MB / 4 MB of 64-byte records
        std::vector<float4> h((size_t)recs * 4);
        uint32_t s = 12345u;
        for (uint32_t i = 0; i < recs; i++) {
            for (int k = 0; k < 3; k++) h[(size_t)i * 4 + k] = make_float4(1.f, 2.f, 3.f, 4.f);
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            uint32_t nx = s & (recs - 1); float f; memcpy(&f, &nx, 4);
            h[(size_t)i * 4 + 3] = make_float4(f, 0, 0, 0);
        }
        CHK(hipMemcpy(buf, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; rep++) {
            if (big) hipLaunchKernelGGL(calib_gather<64>, dim3(lanes / 256), dim3(256), 0, 0, buf, steps, out, recs - 1);
            else hipLaunchKernelGGL(calib_gather<4>, dim3(lanes / 256), dim3(256), 0, 0, buf, steps, out, recs - 1);
        }
        CHK(hipDeviceSynchronize());
    }
    CHK(hipMemset(buf, 0, 2 << 20));
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_reread<0>, dim3(256), dim3(256), 0, 0, buf, (2 << 20) / 16, out);
    hipLaunchKernelGGL(calib_reread<1>, dim3(256), dim3(256), 0, 0, buf, (2 << 20) / 16, out);
    CHK(hipDeviceSynchronize());
    printf("calib_reread<0>, <1>: 256 workgroups each read a 2 MB table twice; <1> is launched right after <0> on the same stream\n");
    const int valuIters = 4096, valuBlocks = 256 * 8;
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(calib_valu, dim3(valuBlocks), dim3(256), 0, 0, out, valuIters, 0.999f, 0.001f);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(calib_valu_mix, dim3(valuBlocks), dim3(256), 0, 0, out, valuIters, 0.999f, 0.001f);
    CHK(hipDeviceSynchronize());
    printf("calib_valu_mix wave_instructions_per_dispatch %zu (waves %d x iterations %d x 64: sub, mul, min, max, cmp, cndmask, add_u32, and_b32 in equal parts)\n", (size_t)valuBlocks * 4 * valuIters * 64, valuBlocks * 4, valuIters);
    printf("calib_valu wave_instructions_per_dispatch %zu (waves %d x iterations %d x 64 v_fma_f32; + loop and epilogue)\n", (size_t)valuBlocks * 4 * valuIters * 64, valuBlocks * 4, valuIters);
    printf("calib_stream bytes_per_dispatch %zu\ncalib_store bytes_per_dispatch %zu\ncalib_gather bytes_per_dispatch %zu (lanes %d x steps %d x 64 B; tables 4 MB <4> and 64 MB <64>)\n",
           streamBytes, streamBytes, (size_t)lanes * steps * 64, lanes, steps);
    return 0;
}
