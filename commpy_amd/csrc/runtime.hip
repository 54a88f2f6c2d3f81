// Runtime plumbing of libcommpy_amd.so: error string, device selection, stream, memory and
// HIP-event timers.  No compute here.
#include "cpx_internal.h"

#include <algorithm>
#include <atomic>
#include <dlfcn.h>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <utility>

namespace cpx {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static hipStream_t g_streams[64] = {};
static std::mutex g_streams_mu;

hipStream_t lib_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_streams_mu);              // two host threads may meet here (ctypes drops the GIL)
    if (!g_streams[dev]) {
        if (hipStreamCreateWithFlags(&g_streams[dev], hipStreamNonBlocking) != hipSuccess) g_streams[dev] = nullptr;
    }
    return g_streams[dev];
}

static thread_local char g_kernel[256] = "";

// "detect and redo" made visible (round 5): a decoder whose redo launch counts the items it decoded again gets a device word here,
// the count is copied to a pinned host word behind the launch (same stream), and cpx_last_kernel appends "redo: n of N" READING THAT
// WORD AT THAT MOMENT: meaningful once the stream the decode was issued on has been synchronised.
static thread_local unsigned *g_redo_word = nullptr;
static thread_local long long g_redo_total = -1;
static thread_local char g_redo_what[48] = "";

void note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
    g_redo_total = -1;
}

// {device words [count, workgroups done] -- zero between launches: the kernel's last workgroup resets them --, pinned host word}.
// No memset and no copy on the stream per decode: the redo kernel's LAST workgroup stores the count to the host word itself (round 5b;
// a memset + a 4-byte D2H copy per decode cost ~8 us of a 0.33 ms map_decode).  Round 6 (advisor): the device words belong to the
// (device, STREAM) the decode is issued on -- scratch slot 12 of that stream, zeroed once when the block is created, released with the
// stream's other scratch blocks (cpx_stream_destroy / cpx_release_workspace) -- so two redo launches in flight on two streams never
// share a counter, and launches on one stream are ordered.  The pinned host word stays per calling thread: it is what
// cpx_last_kernel(), a per-thread string, reads.
RedoCounter redo_counter(hipStream_t st) {
    RedoCounter rc{nullptr, nullptr};
    if (!g_redo_word) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return rc; }
        g_redo_word = static_cast<unsigned *>(p);
        *g_redo_word = 0;
    }
    void *w = nullptr;
    bool fresh = false;
    if (workspace(st, 12, 64, &w, &fresh) != CPX_OK) return rc;
    if (fresh && hipMemsetAsync(w, 0, 64, st) != hipSuccess) { (void)hipGetLastError(); return rc; }
    rc.dev = static_cast<unsigned *>(w);
    rc.host = g_redo_word;
    return rc;
}

void note_redo(long long total, const char *what) {
    g_redo_total = g_redo_word ? total : -1;
    snprintf(g_redo_what, sizeof(g_redo_what), "%s", what);
}

const char *last_kernel_name() { return g_kernel; }

int check_handle_device(int handle_device, const char *what) {
    int dev = -1;
    CPX_HIP(hipGetDevice(&dev));
    CPX_REQUIRE(dev == handle_device, CPX_EINVAL,
                "%s: the handle's tables live on device %d but device %d is current (handles belong to the device "
                "that was current when they were created)", what, handle_device, dev);
    return CPX_OK;
}

struct WsEntry { int dev; hipStream_t st; int slot; void *p; size_t cap; };
static std::vector<WsEntry> g_ws;
static std::mutex g_ws_mu;

int workspace(hipStream_t stream, int slot, size_t bytes, void **out, bool *fresh) {
    if (fresh) *fresh = false;
    int dev = 0;
    CPX_HIP(hipGetDevice(&dev));
    if (bytes == 0) bytes = 8;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &e : g_ws)
        if (e.dev == dev && e.st == stream && e.slot == slot) {
            if (e.cap < bytes) {
                CPX_HIP(hipStreamSynchronize(stream));          // nothing may still use the old block
                CPX_HIP(hipFree(e.p));
                e.p = nullptr; e.cap = 0;
                hipError_t er = hipMalloc(&e.p, bytes);
                if (er != hipSuccess) { set_error("workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(er)); return CPX_ENOMEM; }
                e.cap = bytes;
                if (fresh) *fresh = true;
            }
            *out = e.p;
            return CPX_OK;
        }
    WsEntry e{dev, stream, slot, nullptr, 0};
    hipError_t er = hipMalloc(&e.p, bytes);
    if (er != hipSuccess) { set_error("workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(er)); return CPX_ENOMEM; }
    e.cap = bytes;
    g_ws.push_back(e);
    *out = e.p;
    if (fresh) *fresh = true;
    return CPX_OK;
}

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libcommpy_amd has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return CPX_ENODEV;
    }
    return CPX_OK;
}

}  // namespace cpx

using namespace cpx;

static std::atomic<int> g_precision{-1};   // -1: not read from CPX_PRECISION yet, 0 fp64-parity, 1 fp32-fast

extern "C" {

const char *cpx_last_error(void) { return g_err; }

int cpx_version(void) { return 300; }  // 0.3.0

#ifndef CPX_BUILD_ID
#define CPX_BUILD_ID "unknown"
#endif
// "full:<sha16>;viterbi:<sha16>": digests of the sources this library was compiled from (commpy_amd/build.py)
const char *cpx_build_id(void) { return CPX_BUILD_ID; }

int cpx_device_count(int *n) {
    CPX_REQUIRE(n, CPX_EINVAL, "cpx_device_count: null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *n = c;
    return CPX_OK;
}

int cpx_set_device(int device) {
    int rc = ensure_device();
    if (rc) return rc;
    CPX_HIP(hipSetDevice(device));
    return CPX_OK;
}

int cpx_get_device(int *device) {
    CPX_REQUIRE(device, CPX_EINVAL, "cpx_get_device: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    CPX_HIP(hipGetDevice(device));
    return CPX_OK;
}

int cpx_last_kernel(char *name, int cap) {
    CPX_REQUIRE(name && cap > 0, CPX_EINVAL, "cpx_last_kernel: null buffer");
    if (g_redo_total >= 0 && g_redo_word)
        snprintf(name, (size_t)cap, "%s; redo: %u of %lld %s", g_kernel, *(volatile unsigned *)g_redo_word, g_redo_total, g_redo_what);
    else
        snprintf(name, (size_t)cap, "%s", g_kernel);
    return CPX_OK;
}

int cpx_set_precision(const char *mode) {
    int v = -1;
    if (!mode || !mode[0] || strcmp(mode, "fp64-parity") == 0 || strcmp(mode, "fp64") == 0) v = 0;
    else if (strcmp(mode, "fp32-fast") == 0 || strcmp(mode, "fp32") == 0) v = 1;
    CPX_REQUIRE(v >= 0, CPX_EINVAL, "cpx_set_precision: unknown mode '%s' (fp64-parity | fp32-fast)", mode);
    g_precision.store(v, std::memory_order_relaxed);
    return CPX_OK;
}

int cpx_get_precision(void) { return cpx::precision_fast() ? 1 : 0; }

}  // extern "C"

namespace cpx {
// roctx ranges: when CPX_TRACE=1, every decoder entry point -- and the upload / kernels / download phases of the
// host-buffer entry points -- is bracketed by roctxRangePush / roctxRangePop, resolved at run time from the profiler's roctx
// library (rocprofv3 --marker-trace shows them next to the kernels); off by default: no dlopen, no calls.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("CPX_TRACE");
        if (!e || e[0] != '1') return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
}  // namespace

// Host threads may call into the library concurrently (ctypes drops the GIL).  Kernels of one stream serialise, but the
// scratch arena is shared per (device, stream, slot): between `workspace()` handing out a block and the launches that use it,
// another thread growing the same slot would free it (the grow path synchronises the stream first -- which only protects work
// that has already been issued).  Every entry point that takes arena memory therefore holds the device's issue lock from
// before `workspace()` until its launches are queued; it is recursive (host-buffer entry points call the device ones).
namespace { std::recursive_mutex g_issue_mu[64]; }
IssueGuard::IssueGuard() : dev(0) {
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    g_issue_mu[dev].lock();
}
IssueGuard::~IssueGuard() { g_issue_mu[dev].unlock(); }
void issue_lock(int dev, bool lock) { if (lock) g_issue_mu[dev].lock(); else g_issue_mu[dev].unlock(); }

// Download into PAGEABLE host memory, fast: the runtime's own pageable device-to-host path moved a fresh 0.57 GB NumPy result
// at 24 GB/s (page faults of the destination inside the copy); here the data crosses PCIe into two pinned 64 MB blocks,
// alternately, and host threads copy a finished block into the caller's array (first touch spread over the cores) while the
// next one is in flight.  Blocks until the data is in `dst`.
int d2h_pageable(void *dst, const void *d_src, size_t bytes, hipStream_t st) {
    constexpr size_t CH = (size_t)64 << 20;
    if (bytes < 2 * CH) {
        CPX_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        CPX_HIP(hipStreamSynchronize(st));
        return CPX_OK;
    }
    static std::mutex mu;
    static unsigned char *pin[2] = {nullptr, nullptr};
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < 2; i++)
        if (!pin[i]) CPX_HIP(hipHostMalloc((void **)&pin[i], CH, hipHostMallocDefault));
    hipEvent_t ev[2];
    CPX_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const size_t n = (bytes + CH - 1) / CH;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t len = std::min(CH, bytes - c * CH);
        hipError_t e = hipMemcpyAsync(pin[c & 1], static_cast<const unsigned char *>(d_src) + c * CH, len, hipMemcpyDeviceToHost, st);
        return e != hipSuccess ? e : hipEventRecord(ev[c & 1], st);
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 8 ? 8 : (nt < 1 ? 1 : nt);
    hipError_t err = issue(0);
    for (size_t c = 0; c < n && err == hipSuccess; c++) {
        if (c + 1 < n) err = issue(c + 1);                        // its pinned block was emptied in the previous round
        if (err == hipSuccess) err = hipEventSynchronize(ev[c & 1]);
        if (err != hipSuccess) break;
        const size_t len = std::min(CH, bytes - c * CH);
        unsigned char *to = static_cast<unsigned char *>(dst) + c * CH;
        const unsigned char *from = pin[c & 1];
        auto work = [&](unsigned i) { const size_t lo = len * i / nt, hi = len * (i + 1) / nt; memcpy(to + lo, from + lo, hi - lo); };
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; i++) th.emplace_back(work, i);
        work(0);
        for (auto &x : th) x.join();
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (err != hipSuccess) { set_error("download failed: %s", hipGetErrorString(err)); (void)hipStreamSynchronize(st); return CPX_EHIP; }
    return CPX_OK;
}

TraceRange::TraceRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
TraceRange::~TraceRange() { if (on) roctx().pop(); }
bool trace_enabled() { return roctx().push != nullptr; }

bool precision_fast() {
    int v = g_precision.load(std::memory_order_relaxed);
    if (v < 0) {
        static std::once_flag once;
        std::call_once(once, [] {
            const char *e = getenv("CPX_PRECISION");
            g_precision.store(e && (strcmp(e, "fp32-fast") == 0 || strcmp(e, "fp32") == 0) ? 1 : 0, std::memory_order_relaxed);
        });
        v = g_precision.load(std::memory_order_relaxed);
    }
    return v == 1;
}
int device_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

int resident_blocks(const void *fn, int threads) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> cache;
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto key = std::make_pair(fn, dev * 4096 + threads);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    const int r = per_cu * device_cus();
    cache[key] = r;
    return r;
}
}  // namespace cpx

extern "C" {

int cpx_device_info(char *name, int name_cap, int *compute_units, int64_t *hbm_bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    CPX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    CPX_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return CPX_OK;
}

int cpx_malloc(void **dptr, size_t bytes) {
    CPX_REQUIRE(dptr, CPX_EINVAL, "cpx_malloc: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return CPX_ENOMEM; }
    return CPX_OK;
}

int cpx_free(void *dptr) {
    if (dptr) CPX_HIP(hipFree(dptr));
    return CPX_OK;
}

int cpx_memset(void *dptr, int value, size_t bytes) {
    CPX_HIP(hipMemsetAsync(dptr, value, bytes, lib_stream()));
    CPX_HIP(hipStreamSynchronize(lib_stream()));
    return CPX_OK;
}

// The synchronous copies run ON the library stream and wait for it: they are ordered after every `_dev` call that was given
// stream = NULL.  (The library stream is non-blocking, so a plain hipMemcpy on the null stream is NOT ordered with it: a
// cpx_memcpy_d2h right after an asynchronous `_dev` call could read the buffer before the kernel had written it.)
int cpx_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    CPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, lib_stream()));
    CPX_HIP(hipStreamSynchronize(lib_stream()));
    return CPX_OK;
}

int cpx_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    CPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, lib_stream()));
    CPX_HIP(hipStreamSynchronize(lib_stream()));
    return CPX_OK;
}

int cpx_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream) {
    CPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick_stream(stream)));
    return CPX_OK;
}

int cpx_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream) {
    CPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick_stream(stream)));
    return CPX_OK;
}

int cpx_memcpy_d2d_async(void *dst, const void *src, size_t bytes, void *stream) {
    CPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick_stream(stream)));
    return CPX_OK;
}

int cpx_stream_create(void **stream) {
    CPX_REQUIRE(stream, CPX_EINVAL, "cpx_stream_create: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t st = nullptr;
    CPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return CPX_OK;
}

// The scratch blocks a stream's calls grew go with it: an entry that outlived its stream made cpx_release_workspace()
// synchronise a destroyed stream (undefined in HIP: std::bad_variant_access from inside the runtime and an abort, found in round 5
// by running the collectives tests before a test that releases the workspace).
int cpx_stream_destroy(void *stream) {
    if (!stream) return CPX_OK;
    hipStream_t st = (hipStream_t)stream;
    (void)hipStreamSynchronize(st);                               // outside the lock: other threads' workspace() calls do not wait for this stream
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        for (size_t i = 0; i < g_ws.size();) {
            if (g_ws[i].st != st) { i++; continue; }
            (void)hipFree(g_ws[i].p);
            g_ws.erase(g_ws.begin() + i);
        }
    }
    CPX_HIP(hipStreamDestroy(st));
    return CPX_OK;
}

int cpx_release_workspace(void) {
    // no entry point may sit between workspace() and its launches while the blocks are freed: all issue locks, in order
    struct AllIssueLocks {
        AllIssueLocks() { for (int d = 0; d < 64; d++) cpx::issue_lock(d, true); }
        ~AllIssueLocks() { for (int d = 63; d >= 0; d--) cpx::issue_lock(d, false); }
    } all;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto &e : g_ws) {
        (void)hipStreamSynchronize(e.st);
        (void)hipFree(e.p);
    }
    g_ws.clear();
    return CPX_OK;
}

int cpx_stream_sync(void *stream) {
    CPX_HIP(hipStreamSynchronize(pick_stream(stream)));
    return CPX_OK;
}

void *cpx_default_stream(void) { return (void *)lib_stream(); }

struct cpx_timer_t {
    hipEvent_t a, b;
};

int cpx_timer_create(void **timer) {
    CPX_REQUIRE(timer, CPX_EINVAL, "cpx_timer_create: null pointer");
    int rc = ensure_device();
    if (rc) return rc;
    cpx_timer_t *t = new cpx_timer_t;
    CPX_HIP(hipEventCreate(&t->a));
    CPX_HIP(hipEventCreate(&t->b));
    *timer = t;
    return CPX_OK;
}

int cpx_timer_start(void *timer, void *stream) {
    CPX_HIP(hipEventRecord(((cpx_timer_t *)timer)->a, pick_stream(stream)));
    return CPX_OK;
}

int cpx_timer_stop(void *timer, void *stream) {
    CPX_HIP(hipEventRecord(((cpx_timer_t *)timer)->b, pick_stream(stream)));
    return CPX_OK;
}

int cpx_timer_elapsed_ms(void *timer, float *ms) {
    cpx_timer_t *t = (cpx_timer_t *)timer;
    CPX_HIP(hipEventSynchronize(t->b));
    CPX_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return CPX_OK;
}

int cpx_timer_destroy(void *timer) {
    cpx_timer_t *t = (cpx_timer_t *)timer;
    if (!t) return CPX_OK;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    return CPX_OK;
}

// ---- shader-clock probe (round 5) ------------------------------------------------------------------------------------------
// One wavefront on a stream of its own that mostly sleeps and, when `spin_ms` of the constant-rate reference clock
// (s_memrealtime; hipDeviceAttributeWallClockRate) have passed, stores how far the SHADER-clock counter (s_memtime: on the gfx9
// family a free-running counter of the shader core clock) advanced meanwhile: their quotient is the average sclk of that interval,
// i.e. of whatever ran on the other streams during it.  profiles/README.md used to ARGUE that rocprofv3 pins ~2.1 GHz where an
// unprofiled run boosts to 2.4 GHz (same cycle count, 1.69 vs 1.55 ms per launch); bench.py now RECORDS the clock next to the time.
struct cpx_sclk_probe_t {
    hipStream_t st;
    uint64_t *d;          // device: r0, c0, r1, c1
    int ref_khz;
    int dev, slot;        // the result slot this probe owns (cpx_sclk_probe_start)
};

__global__ void sclk_probe_kernel(uint64_t *out, uint64_t ref_ticks) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    uint64_t r;
    do {
        __builtin_amdgcn_s_sleep(32);                            // ~2000 cycles asleep per poll: the probe takes no issue slots to speak of
        r = __builtin_amdgcn_s_memrealtime();
    } while (r - r0 < ref_ticks);
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = r0; out[1] = c0; out[2] = r; out[3] = c1; }
}

// (stream and result buffer are created once per thread and device and kept: the first version created them per call, and the host
//  time that took -- plus the probe kernel's first-launch cost -- left the GPU idle long enough to drop its clock right before the
//  interval it was meant to measure: 2246 MHz / 1.60 ms in the first repetition of scripts/micro/sclk_probe_check.py, 2380 MHz / 1.54 ms
//  in every later one)
// Round 6 (advisor): one probe stream and one block of 16 result slots PER DEVICE (not per thread): every outstanding probe owns a
// slot of its own, a seventeenth is refused, a thread that changes devices leaks nothing, and a probe that is never read is given back
// with cpx_sclk_probe_destroy.
constexpr int PROBE_SLOTS = 16;
static std::mutex g_probe_mu;
static hipStream_t g_probe_stream[64] = {};
static uint64_t *g_probe_buf[64] = {};
static unsigned g_probe_busy[64] = {};                            // bit i: slot i belongs to an outstanding probe

int cpx_sclk_probe_start(void **probe, double spin_ms) {
    CPX_REQUIRE(probe && spin_ms > 0.0 && spin_ms <= 10000.0, CPX_EINVAL, "cpx_sclk_probe_start: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0, khz = 0;
    CPX_HIP(hipGetDevice(&dev));
    CPX_REQUIRE(dev >= 0 && dev < 64, CPX_ELIMIT, "cpx_sclk_probe_start: device index %d", dev);
    CPX_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    CPX_REQUIRE(khz > 0, CPX_EHIP, "cpx_sclk_probe_start: the device reports no wall-clock rate");
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(g_probe_mu);
        if (!g_probe_stream[dev]) {
            CPX_HIP(hipStreamCreateWithFlags(&g_probe_stream[dev], hipStreamNonBlocking));
            CPX_HIP(hipMalloc((void **)&g_probe_buf[dev], PROBE_SLOTS * 4 * sizeof(uint64_t)));
        }
        for (int i = 0; i < PROBE_SLOTS && slot < 0; i++)
            if (!(g_probe_busy[dev] & (1u << i))) slot = i;
        CPX_REQUIRE(slot >= 0, CPX_ELIMIT, "cpx_sclk_probe_start: %d probes are outstanding on device %d (read or destroy them)", PROBE_SLOTS, dev);
        g_probe_busy[dev] |= 1u << slot;
    }
    cpx_sclk_probe_t *q = new cpx_sclk_probe_t;
    q->ref_khz = khz;
    q->st = g_probe_stream[dev];
    q->d = g_probe_buf[dev] + 4 * slot;
    q->dev = dev;
    q->slot = slot;
    hipLaunchKernelGGL(sclk_probe_kernel, dim3(1), dim3(64), 0, q->st, q->d, (uint64_t)(spin_ms * (double)khz));
    if (hipGetLastError() != hipSuccess) {
        cpx_sclk_probe_destroy(q);
        set_error("cpx_sclk_probe_start: launch failed");
        return CPX_EHIP;
    }
    *probe = q;
    return CPX_OK;
}

int cpx_sclk_probe_destroy(void *probe) {
    cpx_sclk_probe_t *q = (cpx_sclk_probe_t *)probe;
    if (!q) return CPX_OK;
    (void)hipStreamSynchronize(q->st);                            // the slot may be handed out again: its kernel must be done
    {
        std::lock_guard<std::mutex> lk(g_probe_mu);
        g_probe_busy[q->dev] &= ~(1u << q->slot);
    }
    delete q;
    return CPX_OK;
}

int cpx_sclk_probe_read(void *probe, double *sclk_mhz, double *interval_ms) {
    cpx_sclk_probe_t *q = (cpx_sclk_probe_t *)probe;
    CPX_REQUIRE(q, CPX_EINVAL, "cpx_sclk_probe_read: null probe");
    uint64_t h[4] = {0, 0, 0, 0};
    hipError_t e = hipStreamSynchronize(q->st);
    if (e == hipSuccess) e = hipMemcpy(h, q->d, sizeof(h), hipMemcpyDeviceToHost);
    const int khz = q->ref_khz;
    cpx_sclk_probe_destroy(q);                                    // (stream and slots stay with the device)
    if (e != hipSuccess) { set_error("cpx_sclk_probe_read: %s", hipGetErrorString(e)); return CPX_EHIP; }
    const double dt_ms = (double)(h[2] - h[0]) / (double)khz;
    if (interval_ms) *interval_ms = dt_ms;
    if (sclk_mhz) *sclk_mhz = dt_ms > 0.0 ? (double)(h[3] - h[1]) / dt_ms * 1e-3 : 0.0;
    return CPX_OK;
}

}  // extern "C"
