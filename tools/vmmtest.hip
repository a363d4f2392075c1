// does the HIP virtual-memory API (reserve a range, map physical chunks on demand) work on this box? (developer probe for Arena)
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("FAIL %s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); return 1; } } while (0)
__global__ void fill(float* p, size_t n, float v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + (float)(i & 1023); }
__global__ void sum(const float* p, size_t n, double* out) { double a = 0; for (size_t i = threadIdx.x; i < n; i += blockDim.x) a += p[i]; atomicAdd(out, a); }
int main() {
    int dev = 0;
    HC(hipSetDevice(dev));
    int vmm = 0;
    HC(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("VirtualMemoryManagementSupported = %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    HC(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    size_t freeb = 0, total = 0;
    HC(hipMemGetInfo(&freeb, &total));
    printf("free before reserve %.2f GB\n", freeb / 1e9);
    const size_t reserve = (size_t)12 << 30;
    void* base = nullptr;
    HC(hipMemAddressReserve(&base, reserve, gran, nullptr, 0));
    HC(hipMemGetInfo(&freeb, &total));
    printf("free after reserving 12 GB of addresses %.2f GB\n", freeb / 1e9);
    const size_t chunk = ((size_t)256 << 20) / gran * gran;
    hipMemGenericAllocationHandle_t h[4];
    for (int i = 0; i < 4; ++i) {
        HC(hipMemCreate(&h[i], chunk, &prop, 0));
        HC(hipMemMap((char*)base + i * chunk, chunk, 0, h[i], 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        HC(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
    }
    HC(hipMemGetInfo(&freeb, &total));
    printf("free after mapping 1 GB %.2f GB\n", freeb / 1e9);
    const size_t n = 4 * chunk / 4;
    hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, (float*)base, n, 1.f);
    double* d; HC(hipMalloc(&d, 8)); HC(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(sum, dim3(1), dim3(1024), 0, 0, (const float*)base + (n - 4096), (size_t)4096, d);
    double hsum = 0; HC(hipMemcpy(&hsum, d, 8, hipMemcpyDeviceToHost));
    printf("tail sum %.1f (expect %.1f)\n", hsum, 4096.0 + 4.0 * (1023.0 * 1024.0 / 2.0));
    // a kernel in a captured graph touching the mapped range
    hipStream_t s; HC(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    HC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, s, (float*)base, n, 2.f);
    HC(hipStreamEndCapture(s, &g));
    HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HC(hipGraphLaunch(ge, s));
    HC(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) { HC(hipMemUnmap((char*)base + i * chunk, chunk)); HC(hipMemRelease(h[i])); }
    HC(hipMemAddressFree(base, reserve));
    HC(hipMemGetInfo(&freeb, &total));
    printf("free after release %.2f GB\nVMM OK\n", freeb / 1e9);
    return 0;
}
