// devmem.cpp -- every device allocation of the library (dev_malloc / dev_free, common.hpp).
//
// Product mode: hipMalloc / hipFree.
//
// Diagnostic modes, chosen once per process by the environment (tests/test_gpu_guard.py and tools/gpu_accept.sh run the GPU
// suite and the soaks under them; NOTEBOOK.md R5.1 has what they found):
//
//   HFNET_GUARD_ALLOC=1   every allocation is its own virtual-memory mapping whose LAST byte is the buffer's last byte
//                         (16-byte granularity), followed by reserved-but-unmapped address space: a kernel that reads or writes
//                         one element past the end of any buffer faults on the spot instead of silently touching a neighbour.
//   HFNET_GUARD_ALLOC=2   the same with the buffer at the START of its mapping and unmapped space in front (row -1, negative
//                         offsets).
//   HFNET_GUARD_FILL=xx   (hex byte, with or without the guard) fresh allocations are filled with that byte before they are
//                         handed out, device-synchronised: code that relies on "new memory is zero" or reads a buffer before
//                         its first write computes with 0xFF.. (NaN / -1) or 0x7F.. (NaN / 2^31-ish counts) instead of zeros.
//
// A device fault still takes the process down (ROCr aborts on "Memory access fault"); the point of the modes is that it then
// happens in the FIRST test that has the defect, on every box, rather than once in a thousand runs on somebody else's.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace hfnet {
namespace {

struct Mapping { void* va; size_t reserved; hipMemGenericAllocationHandle_t handle; void* mapped; size_t mapped_bytes; };

struct GuardState {
    int mode = 0;          // 0 off, 1 guard behind the end, 2 guard in front of the start
    int fill = -1;         // -1 off, else the byte
    std::mutex mu;
    std::unordered_map<void*, Mapping> live;
    GuardState() {
        if (const char* g = std::getenv("HFNET_GUARD_ALLOC")) mode = std::atoi(g);
        if (mode < 0 || mode > 2) mode = 0;
        if (const char* f = std::getenv("HFNET_GUARD_FILL")) { if (*f) fill = (int)(std::strtol(f, nullptr, 16) & 0xff); }
    }
};
GuardState& state() { static GuardState s; return s; }

hipError_t guarded_malloc(GuardState& s, void** out, size_t bytes) {
    int dev = 0;
    hipError_t r = hipGetDevice(&dev);
    if (r != hipSuccess) return r;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    r = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (r != hipSuccess) return r;
    if (gran == 0) return hipErrorInvalidValue;
    const size_t user = (bytes + 15) / 16 * 16;
    Mapping m = {};
    m.mapped_bytes = (user + gran - 1) / gran * gran;
    m.reserved = m.mapped_bytes + 2 * gran;                      // one unmapped granule on either side
    r = hipMemAddressReserve(&m.va, m.reserved, gran, nullptr, 0);
    if (r != hipSuccess) return r;
    r = hipMemCreate(&m.handle, m.mapped_bytes, &prop, 0);
    if (r != hipSuccess) { (void)hipMemAddressFree(m.va, m.reserved); return r; }
    m.mapped = (char*)m.va + gran;
    r = hipMemMap(m.mapped, m.mapped_bytes, 0, m.handle, 0);
    if (r != hipSuccess) { (void)hipMemRelease(m.handle); (void)hipMemAddressFree(m.va, m.reserved); return r; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    r = hipMemSetAccess(m.mapped, m.mapped_bytes, &acc, 1);
    if (r != hipSuccess) { (void)hipMemUnmap(m.mapped, m.mapped_bytes); (void)hipMemRelease(m.handle); (void)hipMemAddressFree(m.va, m.reserved); return r; }
    void* p = s.mode == 1 ? (void*)((char*)m.mapped + m.mapped_bytes - user) : m.mapped;
    { std::lock_guard<std::mutex> lk(s.mu); s.live[p] = m; }
    *out = p;
    return hipSuccess;
}

}  // namespace

int dev_guard_mode() { return state().mode; }

bool trace_launches() {
    static const bool on = [] { const char* t = std::getenv("HFNET_TRACE_LAUNCHES"); return t && *t && *t != '0'; }();
    return on;
}
void trace_launch(const char* name, hipStream_t s, bool after) {
    if (!after) { std::fprintf(stderr, "[hfnet] launch %s\n", name); std::fflush(stderr); return; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cs == hipStreamCaptureStatusNone) (void)hipStreamSynchronize(s);
}

// Blocking copies between HOST memory of any kind and the device through one process-wide pinned block (8 MB pieces): set-up paths (weights,
// resize tables) and odd small read-backs.  No pageable pointer reaches the runtime (HostBounce, engine.hpp, has the why).
namespace {
struct Staging { std::mutex mu; unsigned char* p = nullptr; static constexpr size_t cap = (size_t)8 << 20; };
Staging& staging() { static Staging s; return s; }
}  // namespace
hipError_t copy_h2d_blocking(void* dst_dev, const void* src_host, size_t bytes) {
    Staging& st = staging();
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.p) { void* hp = nullptr; const hipError_t r = hipHostMalloc(&hp, Staging::cap, hipHostMallocDefault); if (r != hipSuccess) return r; st.p = (unsigned char*)hp; }
    for (size_t off = 0; off < bytes; off += Staging::cap) {
        const size_t n = bytes - off < Staging::cap ? bytes - off : Staging::cap;
        std::memcpy(st.p, (const unsigned char*)src_host + off, n);
        const hipError_t r = hipMemcpy((unsigned char*)dst_dev + off, st.p, n, hipMemcpyHostToDevice);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}
hipError_t copy_d2h_blocking(void* dst_host, const void* src_dev, size_t bytes) {
    Staging& st = staging();
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.p) { void* hp = nullptr; const hipError_t r = hipHostMalloc(&hp, Staging::cap, hipHostMallocDefault); if (r != hipSuccess) return r; st.p = (unsigned char*)hp; }
    for (size_t off = 0; off < bytes; off += Staging::cap) {
        const size_t n = bytes - off < Staging::cap ? bytes - off : Staging::cap;
        const hipError_t r = hipMemcpy(st.p, (const unsigned char*)src_dev + off, n, hipMemcpyDeviceToHost);
        if (r != hipSuccess) return r;
        std::memcpy((unsigned char*)dst_host + off, st.p, n);
    }
    return hipSuccess;
}

hipError_t dev_malloc(void** out, size_t bytes) {
    GuardState& s = state();
    if (bytes == 0) bytes = 1;
    hipError_t r = s.mode ? guarded_malloc(s, out, bytes) : hipMalloc(out, bytes);
    if (r != hipSuccess) return r;
    if (s.fill >= 0) {
        r = hipMemset(*out, s.fill, bytes);
        if (r == hipSuccess) r = hipDeviceSynchronize();         // (hipMemset is not host-synchronous on this runtime)
    }
    return r;
}

hipError_t dev_free(void* p) {
    if (!p) return hipSuccess;
    GuardState& s = state();
    if (!s.mode) return hipFree(p);
    Mapping m;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.live.find(p);
        if (it == s.live.end()) return hipFree(p);
        m = it->second;
        s.live.erase(it);
    }
    hipError_t r = hipDeviceSynchronize();                        // hipFree's implicit synchronisation
    hipError_t q = hipMemUnmap(m.mapped, m.mapped_bytes);
    if (r == hipSuccess) r = q;
    q = hipMemRelease(m.handle);
    // The address range stays reserved for the rest of the process (never handed out again): an access through a stale pointer
    // faults as well, and no later allocation can alias a range the GPU's translation caches have seen under another mapping.
    return r == hipSuccess ? q : r;
}

}  // namespace hfnet
