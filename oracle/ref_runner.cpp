// ref_runner.cpp — TEST INFRASTRUCTURE.  Loads a code object produced by oracle/build_ref.sh
// (the reference's OpenCL kernels compiled for gfx950) through the HIP module API and launches
// its kernels with caller-provided arguments, so that tests can run the REFERENCE itself on the
// MI355X and compare stage outputs.  Generic: device buffers, copies, launch-by-name.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static std::string g_err;
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { g_err = std::string(#e) + ": " + hipGetErrorString(r_); return -1; } } while (0)

extern "C" {
const char* ref_last_error() { return g_err.c_str(); }
int ref_init(int device) { CHK(hipSetDevice(device)); CHK(hipFree(nullptr)); return 0; }
int ref_load(const char* path, void** module)
{
    FILE* f = fopen(path, "rb");
    if (!f) { g_err = std::string("cannot open ") + path; return -1; }
    std::vector<char> buf; char tmp[65536]; size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    hipModule_t m;
    CHK(hipModuleLoadData(&m, buf.data()));
    *module = m;
    return 0;
}
int ref_unload(void* module) { CHK(hipModuleUnload((hipModule_t)module)); return 0; }
int ref_malloc(void** p, size_t bytes) { CHK(hipMalloc(p, bytes ? bytes : 16)); CHK(hipMemset(*p, 0, bytes ? bytes : 16)); return 0; }
int ref_free(void* p) { CHK(hipFree(p)); return 0; }
int ref_h2d(void* dst, const void* src, size_t bytes) { CHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int ref_d2h(void* dst, const void* src, size_t bytes) { CHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
// args: array of pointers to the explicit kernel argument values (OpenCL order)
int ref_launch(void* module, const char* kernel, unsigned global, unsigned local, void** args)
{
    hipFunction_t f;
    CHK(hipModuleGetFunction(&f, (hipModule_t)module, kernel));
    if (local == 0 || global % local) { g_err = "global size must be a multiple of local size"; return -1; }
    CHK(hipModuleLaunchKernel(f, global / local, 1, 1, local, 1, 1, 0, nullptr, args, nullptr));
    CHK(hipDeviceSynchronize());
    return 0;
}
// same, bracketed by HIP events: *ms = kernel time (used to time the reference's own extend on the MI355X)
int ref_launch_timed(void* module, const char* kernel, unsigned global, unsigned local, void** args, float* ms)
{
    hipFunction_t f;
    CHK(hipModuleGetFunction(&f, (hipModule_t)module, kernel));
    if (local == 0 || global % local) { g_err = "global size must be a multiple of local size"; return -1; }
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipEventRecord(a, nullptr));
    CHK(hipModuleLaunchKernel(f, global / local, 1, 1, local, 1, 1, 0, nullptr, args, nullptr));
    CHK(hipEventRecord(b, nullptr));
    CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}
}
