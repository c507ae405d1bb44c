// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the hop kernels use
// (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of the bytes of a 16-B-per-lane streaming read; other widths are
// uncalibrated). Each kernel streams a buffer far larger than the Infinity Cache exactly once with W bytes per lane:
//   hipcc --offload-arch=gfx950 -O3 tools/fetchcal.hip -o /tmp/fetchcal
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetchcal   (then WRITE_SIZE in a second pass)
// and compare the counter (KiB) with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <class V>
__global__ __launch_bounds__(256) void rd_kernel(const V *src, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const V v = src[i];
        if constexpr (sizeof(V) == 4) acc += v;
        else acc += v.x;
    }
    if (acc == 1.2345f) *sink = acc;
}
template <class V>
__global__ __launch_bounds__(256) void wr_kernel(V *dst, size_t n, int nt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        V v;
        if constexpr (sizeof(V) == 4) v = (float)i;
        else v = V((float)i);
        if (nt) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB: 8x the Infinity Cache
    void *d; float *sink;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 0, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipDeviceSynchronize());
    const dim3 g(256 * 8), b(256);
    hipLaunchKernelGGL(rd_kernel<float>, g, b, 0, 0, (const float *)d, bytes / 4, sink);
    hipLaunchKernelGGL(rd_kernel<v2f>, g, b, 0, 0, (const v2f *)d, bytes / 8, sink);
    hipLaunchKernelGGL(rd_kernel<v4f>, g, b, 0, 0, (const v4f *)d, bytes / 16, sink);
    hipLaunchKernelGGL(wr_kernel<float>, g, b, 0, 0, (float *)d, bytes / 4, 0);
    hipLaunchKernelGGL(wr_kernel<v2f>, g, b, 0, 0, (v2f *)d, bytes / 8, 0);
    hipLaunchKernelGGL(wr_kernel<v2f>, g, b, 0, 0, (v2f *)d, bytes / 8, 1);
    hipLaunchKernelGGL(wr_kernel<v4f>, g, b, 0, 0, (v4f *)d, bytes / 16, 0);
    CHECK(hipDeviceSynchronize());
    printf("each kernel moves %zu bytes = %zu KiB (rd 4/8/16 B per lane, wr 4/8/8nt/16)\n", bytes, bytes >> 10);
    return 0;
}
