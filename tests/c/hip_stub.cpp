// TEST INFRASTRUCTURE (host sanitizer builds only; never linked into the product library).
// A stand-in for the HIP runtime and for the kernel launchers of rocoder_amd/csrc/*.hip, so that the engine's HOST code
// (rc_engine.cpp: worker pools, the three-set pinned pipeline of the frequency-kernel path, the persistent rc_multi
// workers, the streaming seam's look-ahead, every copy's bounds) links without a GPU and runs under
// AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer (rocoder_amd/csrc/host/sanitize.mk, engine_*).
//   "device" memory = calloc'd host memory (so every hipMemcpy* is a real memcpy the sanitizer checks),
//   streams execute at enqueue time (everything is synchronous), events are empty objects,
//   kernel launches compute NOTHING (outputs stay zero): results are not checked here, only the host logic.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../../rocoder_amd/csrc/rc_kernels.h"

namespace {
std::mutex g_mu;
std::map<const void *, std::pair<size_t, int>> g_dev;  // "device" allocations: base -> (bytes, device)
thread_local int t_device = 0;
constexpr int kDevices = 4;  // a list such as {0, 1, 2} exercises the peer-copy path
struct Obj {
    int tag;
};
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) {
    *n = kDevices;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= kDevices) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = t_device;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof *p);
    p->multiProcessorCount = 256;
    strcpy(p->gcnArchName, "gfx950:stub");
    p->clockRate = 2400000;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
    *p = calloc(1, n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[*p] = {n, t_device};
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev.erase(p);
    }
    free(p);
    return hipSuccess;
}
static std::map<const void *, size_t> g_host;  // page-locked blocks (hipPointerGetAttributes: hipMemoryTypeHost)
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = calloc(1, n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_host[*p] = n ? n : 1;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (p) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_host.erase(p)) return hipErrorInvalidValue;
    }
    free(p);
    return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorInvalidDevice; }  // (no sysfs entry to look up here)
// the engine page-locks anonymous memory it mapped itself (pinned_alloc): the block joins the page-locked set
hipError_t hipHostRegister(void *p, size_t n, unsigned) {
    if (!p || !n) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_host.count(p)) return hipErrorHostMemoryAlreadyRegistered;
    g_host[p] = n;
    return hipSuccess;
}
hipError_t hipHostUnregister(void *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_host.erase(p) ? hipSuccess : hipErrorHostMemoryNotRegistered;
}
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) {
    *d = h;
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dev.upper_bound(p);
    if (it != g_dev.begin()) {
        --it;
        if ((const char *)p < (const char *)it->first + (it->second.first ? it->second.first : 1)) {
            memset(a, 0, sizeof *a);
            a->type = hipMemoryTypeDevice;
            a->device = it->second.second;
            a->devicePointer = const_cast<void *>(p);
            return hipSuccess;
        }
    }
    auto ih = g_host.upper_bound(p);
    if (ih != g_host.begin()) {
        --ih;
        if ((const char *)p < (const char *)ih->first + ih->second) {
            memset(a, 0, sizeof *a);
            a->type = hipMemoryTypeHost;
            a->hostPointer = const_cast<void *>(p);
            return hipSuccess;
        }
    }
    return hipErrorInvalidValue;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    if (n) memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    if (n) memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) {
    if (n) memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    if (n) memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = (hipStream_t) new Obj{1};
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    delete (Obj *)s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) {
    *e = (hipEvent_t) new Obj{2};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete (Obj *)e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hip stub"; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) {
    *can = 1;
    return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
}  // extern "C"

// ---- the kernel launchers (rc_kernels.h): geometry answers as the real ones, launches do nothing
namespace rc {
bool hop_geometry(int log2n, int *threads, size_t *lds_bytes) {
    if (log2n < 5 || log2n > 14) return false;
    const int M = 1 << (log2n - 1);
    const int T = M <= 128 ? std::max(2, M / 8) : std::max(M / 32, std::min(64, M / 4));
    if (threads) *threads = T;
    if (lds_bytes) *lds_bytes = sizeof(float2) * (size_t)(M + (M >> 5) + 1);
    return true;
}
int hop_workgroups_per_cu(int log2n, bool default_window, bool pitch1) { return log2n == 14 ? ((default_window || pitch1) ? 3 : 2) : 0; }
int hop_resident_workgroups(int log2n, bool default_window) {
    if (!default_window) return 0;
    return log2n == 13 ? 6 : log2n == 12 ? 12 : (log2n >= 9 && log2n <= 11) ? 16 : 0;
}
int hop_slots(int log2n) {
    if (log2n < 5 || log2n > 8) return 1;
    const int M = 1 << (log2n - 1), T = M <= 128 ? std::max(2, M / 8) : std::max(M / 32, std::min(64, M / 4));
    return T < 64 ? 64 / T : 1;
}
hipError_t launch_hop(int, HopMode, const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hop16k(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hop16k_prev(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hopw(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hopw9(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hopw10(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hopw11(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_hopw2(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_ola(const OlaParams &, hipStream_t, bool) { return hipSuccess; }
hipError_t launch_resample_slower(const ResampleParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_calib_valu(float *, int, hipStream_t) { return hipSuccess; }
hipError_t launch_big(int, const BigParams &, hipStream_t, HopMode) { return hipSuccess; }
hipError_t launch_big_cr(const BigOlaParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_big4(int, const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_big5(const HopParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_big5s(const HopParams &, hipStream_t) { return hipSuccess; }
size_t big5_lds_bytes() { return 0; }
size_t big4_tail_scratch_floats(int) { return 0; }
hipError_t launch_dev_kernel(const DevKernelParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_prep(const PrepParams &, hipStream_t) { return hipSuccess; }
hipError_t launch_gen(int, const HopParams &, hipStream_t) { return hipSuccess; }
}  // namespace rc
