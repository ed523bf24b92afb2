// Runtime side of the C ABI: errors, device selection, cached workspaces, memory/stream/event
// wrappers.  Nothing here is on the hot path.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "pdc_internal.h"

namespace pdc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct DeviceState {
    void *buf[SLOT_COUNT] = {};
    int64_t cap[SLOT_COUNT] = {};
    std::mutex call_mutex;
    hipStream_t host_stream = nullptr;   // what the host entry points enqueue on (created on first use)
};

static std::mutex g_mutex;
static std::vector<DeviceState *> g_devices;
static int g_count = -1;

static int device_count_cached() {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (g_count < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            return -1;  // not cached: a later call may succeed
        }
        g_count = n;
        g_devices.resize(n, nullptr);
        for (int i = 0; i < n; ++i) g_devices[i] = new DeviceState();
    }
    return g_count;
}

int use_device(int device) {
    int n = device_count_cached();
    if (n <= 0) {
        if (n == 0) set_error("no HIP device is visible to this process");
        return PDC_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return PDC_ERR_INVALID;
    }
    PDC_HIP(hipSetDevice(device));
    return PDC_OK;
}

static std::atomic<int64_t> g_device_allocs{0}, g_pinned_allocs{0};

int device_alloc(void **dptr, int64_t bytes) {
    PDC_HIP(hipMalloc(dptr, (size_t)(bytes > 0 ? bytes : 1)));
    g_device_allocs.fetch_add(1);
    return PDC_OK;
}

int pinned_alloc(void **hptr, int64_t bytes) {
    PDC_HIP(hipHostMalloc(hptr, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault));
    g_pinned_allocs.fetch_add(1);
    return PDC_OK;
}

int cached(int device, Slot slot, int64_t bytes, void **dptr) {
    DeviceState *st = g_devices[device];
    if (bytes < 256) bytes = 256;
    if (st->cap[slot] < bytes) {
        if (st->buf[slot]) {
            PDC_HIP(hipFree(st->buf[slot]));
            st->buf[slot] = nullptr;
            st->cap[slot] = 0;
        }
        int64_t want = bytes + bytes / 8;  // head-room so a slowly growing caller does not thrash
        if (device_alloc(&st->buf[slot], want) != PDC_OK) {
            (void)hipGetLastError();
            want = bytes;
            PDC_TRY(device_alloc(&st->buf[slot], want));
        }
        st->cap[slot] = want;
    }
    *dptr = st->buf[slot];
    return PDC_OK;
}

int64_t work_budget() {
    // (read at every call: a test - or a caller - may set it between calls; the size query and the launch of ONE scan
    // must see the same value)
    const char *e = getenv("PDC_WORK_BUDGET_GB");
    if (!e || !e[0]) return 0;
    const double gb = atof(e);
    return gb > 0.0 ? (int64_t)(gb * (double)(1 << 30)) : 0;
}

int64_t host_work_budget(int device) {
    int64_t b = work_budget();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const DeviceState *st = g_devices[device];
        const int64_t avail = (int64_t)((double)((int64_t)free_b + (st ? st->cap[SLOT_WORK] : 0)) * 0.9);
        if (avail > 0 && (b == 0 || avail < b)) b = avail;
    } else {
        (void)hipGetLastError();
    }
    return b;
}

static thread_local double t_work_scale = 1.0;
static thread_local int t_work_depth = 0;
double work_scale() { return t_work_scale; }
double *work_scale_slot(int **depth) {
    *depth = &t_work_depth;
    return &t_work_scale;
}

// The host entry points (numpy in, numpy out) run on ONE non-blocking stream per device, not on the legacy NULL
// stream: a thread that drives its own streams through the `_dev` entries is no longer serialised against them by
// the default stream's implicit synchronisation.  The caller holds the device's DeviceLock.
int host_stream(int device, hipStream_t *st) {
    DeviceState *d = g_devices[device];
    if (!d->host_stream) PDC_HIP(hipStreamCreateWithFlags(&d->host_stream, hipStreamNonBlocking));
    *st = d->host_stream;
    return PDC_OK;
}

struct StreamScratch {
    int device;
    hipStream_t stream;
    void *buf;
    int64_t cap;
    int pins;   // 1 between stream_scratch() and stream_scratch_done() (the holder's launches are not enqueued yet), else 0
};
static std::mutex g_scratch_mutex;
static std::condition_variable g_scratch_cv;   // signalled by stream_scratch_done()
static std::vector<StreamScratch> g_scratch;   // least recently used entry first
constexpr size_t kScratchPerDevice = 64;

// (the caller holds g_scratch_mutex) hipFree synchronises the device: no kernel still uses the block
static int free_scratch_entry(size_t i) {
    StreamScratch s = g_scratch[i];
    g_scratch.erase(g_scratch.begin() + (long)i);
    if (s.buf) {
        int now = -1;
        PDC_HIP(hipGetDevice(&now));
        PDC_HIP(hipSetDevice(s.device));
        PDC_HIP(hipFree(s.buf));
        PDC_HIP(hipSetDevice(now));
    }
    return PDC_OK;
}

// An entry may go only when nobody can still be using it: no caller holds it pinned (its launches would come
// AFTER the free) and its stream has run dry (or no longer exists).
static bool scratch_entry_idle(const StreamScratch &s) {
    if (s.pins > 0) return false;
    const hipError_t q = hipStreamQuery(s.stream);
    if (q != hipSuccess) (void)hipGetLastError();
    return q != hipErrorNotReady;
}

// The block comes back PINNED: call stream_scratch_done() once every launch that uses it has been enqueued.
int stream_scratch(int device, hipStream_t stream, int64_t bytes, void **dptr) {
    std::unique_lock<std::mutex> lk(g_scratch_mutex);
    if (bytes < 256) bytes = 256;
    // ONE caller at a time between stream_scratch() and stream_scratch_done() on a (device, stream): two host threads
    // that enqueue on the same stream share its block, and launches of theirs that interleaved in the stream would
    // read each other's partial results (the legacy NULL stream is the likely shared one).  Stream order then keeps
    // the block's users apart; a caller that needs a larger block finds the entry unpinned.
    size_t at;
    for (;;) {
        at = g_scratch.size();
        for (size_t i = 0; i < g_scratch.size(); ++i)
            if (g_scratch[i].device == device && g_scratch[i].stream == stream) at = i;
        if (at == g_scratch.size() || g_scratch[at].pins == 0) break;
        g_scratch_cv.wait(lk);
    }
    if (at == g_scratch.size()) {
        // a stream handle this table has not seen: the caller may be cycling through transient streams (and
        // HIP may never hand the same address out again), so a device's entries are bounded - the least
        // recently used IDLE one goes; entries in use are never freed, the table rather grows
        size_t mine = 0;
        for (const StreamScratch &e : g_scratch) mine += e.device == device;
        for (size_t i = 0; mine >= kScratchPerDevice && i < g_scratch.size();) {
            if (g_scratch[i].device == device && scratch_entry_idle(g_scratch[i])) {
                PDC_TRY(free_scratch_entry(i));
                --mine;
            } else {
                ++i;
            }
        }
        g_scratch.push_back({device, stream, nullptr, 0, 0});
    } else if (at + 1 != g_scratch.size()) {   // a hit moves to the back: true LRU order
        const StreamScratch hit = g_scratch[at];
        g_scratch.erase(g_scratch.begin() + (long)at);
        g_scratch.push_back(hit);
    }
    StreamScratch *e = &g_scratch.back();
    if (e->cap < bytes) {
        // (the entry is unpinned here - the wait above -: nobody is between taking the block and enqueuing on it, and
        // hipFree synchronises the device, so no kernel still uses it)
        if (e->buf) PDC_HIP(hipFree(e->buf));
        e->buf = nullptr;
        e->cap = 0;
        const int64_t want = bytes + bytes / 4;
        PDC_TRY(device_alloc(&e->buf, want));
        e->cap = want;
    }
    ++e->pins;
    *dptr = e->buf;
    return PDC_OK;
}

void stream_scratch_done(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    g_scratch_cv.notify_all();
    for (StreamScratch &e : g_scratch)
        if (e.device == device && e.stream == stream && e.pins > 0) --e.pins;
}

int drop_stream_scratch(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    for (size_t i = 0; i < g_scratch.size(); ++i)
        if (g_scratch[i].device == device && g_scratch[i].stream == stream) return free_scratch_entry(i);
    return PDC_OK;
}

static int release_stream_scratch() {
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    for (StreamScratch &s : g_scratch) {
        if (!s.buf) continue;
        PDC_HIP(hipSetDevice(s.device));
        PDC_HIP(hipFree(s.buf));
        s.buf = nullptr;
        s.cap = 0;
    }
    g_scratch.clear();
    return PDC_OK;
}

int allow_dynamic_lds(const void *kernel, int bytes) {
    struct Done {
        int device;
        const void *kernel;
        int bytes;
    };
    static std::mutex mu;
    static std::vector<Done> done;
    int device = -1;
    PDC_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(mu);
    for (Done &d : done)
        if (d.device == device && d.kernel == kernel) {
            if (d.bytes >= bytes) return PDC_OK;
            PDC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            d.bytes = bytes;
            return PDC_OK;
        }
    PDC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.push_back({device, kernel, bytes});
    return PDC_OK;
}

DeviceLock::DeviceLock(int d) : device(d) {
    if (d >= 0 && d < (int)g_devices.size()) g_devices[d]->call_mutex.lock();
}
DeviceLock::~DeviceLock() {
    if (device >= 0 && device < (int)g_devices.size()) g_devices[device]->call_mutex.unlock();
}


// ---- clock probe ------------------------------------------------------------------------------------------------
// A fixed count of fp64 fmas with nothing else in the way: 16 independent chains per lane, 4 waves per SIMD (1024
// workgroups of 256 threads, 40 KB of LDS each so that a CU takes exactly four).  fp64 fma = 4 issue cycles per
// wave64 instruction, so  wave-instructions per SIMD x 4 / HIP-event time  = the clock the chip SUSTAINS under fp64
// load on this box, in this minute (the chip is power-limited there: 2.04-2.13 GHz seen across the pool against the
// nominal 2.4).  One wave also reads s_memtime / s_memrealtime around its loop (shader clock against the 100 MHz
// reference clock).
__global__ __launch_bounds__(256) void clock_probe_kernel(double *out, unsigned long long *stamps, double a, double b, int iters) {
    extern __shared__ double probe_lds[];
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = threadIdx.x * 1e-3 + c;
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        c0 = __builtin_readcyclecounter();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = __builtin_fma(x[c], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += x[c];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = __builtin_readcyclecounter() - c0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (s == 12345.678) probe_lds[threadIdx.x] = s;   // (never: keeps the LDS allocation alive)
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace pdc

using namespace pdc;

extern "C" {

const char *pdc_last_error(void) { return g_err; }

int pdc_version(void) { return 1; }

int pdc_device_count(int *count) {
    PDC_REQUIRE(count != nullptr, "count is NULL");
    int n = device_count_cached();
    if (n < 0) {
        *count = 0;
        return PDC_ERR_NODEVICE;
    }
    *count = n;
    return PDC_OK;
}

int pdc_device_info(int device, char *name, int name_len, int *cu_count, int64_t *hbm_bytes,
                    int *clock_khz) {
    PDC_TRY(use_device(device));
    hipDeviceProp_t prop;
    PDC_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (clock_khz) *clock_khz = prop.clockRate;
    return PDC_OK;
}

int pdc_release(void) {
    release_multi();
    PDC_TRY(release_stream_scratch());
    std::lock_guard<std::mutex> lk(g_mutex);
    for (size_t d = 0; d < g_devices.size(); ++d) {
        DeviceState *st = g_devices[d];
        std::lock_guard<std::mutex> lk2(st->call_mutex);
        bool any = st->host_stream != nullptr;
        for (int s = 0; s < SLOT_COUNT; ++s) any |= st->buf[s] != nullptr;
        if (!any) continue;
        PDC_HIP(hipSetDevice((int)d));
        if (st->host_stream) {   // (the host entry points' stream: idle - every host entry ends with a synchronise and holds call_mutex)
            PDC_HIP(hipStreamDestroy(st->host_stream));
            st->host_stream = nullptr;
        }
        for (int s = 0; s < SLOT_COUNT; ++s) {
            if (st->buf[s]) PDC_HIP(hipFree(st->buf[s]));
            st->buf[s] = nullptr;
            st->cap[s] = 0;
        }
    }
    return PDC_OK;
}

// Test hooks for the per-(device, stream) scratch table behind the `_dev` entry points that take no workspace
// (pdc_internal.h: stream_scratch): pin the stream's block / unpin it.  Not for callers - tests/test_multi_gpu.py
// drives the table's waiting rule with them (they replace the mangled C++ names the test bound to up to round 5).
int pdc_test_scratch_pin(int device, void *stream, int64_t bytes, void **dptr) {
    PDC_REQUIRE(dptr && bytes >= 0, "test_scratch_pin: bad argument");
    PDC_TRY(use_device(device));
    return stream_scratch(device, (hipStream_t)stream, bytes, dptr);
}
int pdc_test_scratch_unpin(int device, void *stream) {
    PDC_TRY(use_device(device));
    stream_scratch_done(device, (hipStream_t)stream);
    return PDC_OK;
}

int pdc_alloc_counts(int64_t *device_allocs, int64_t *pinned_allocs) {
    if (device_allocs) *device_allocs = g_device_allocs.load();
    if (pinned_allocs) *pinned_allocs = g_pinned_allocs.load();
    return PDC_OK;
}

int pdc_malloc(int device, int64_t bytes, void **dptr) {
    PDC_REQUIRE(dptr != nullptr && bytes >= 0, "pdc_malloc: bad arguments");
    PDC_TRY(use_device(device));
    return device_alloc(dptr, bytes);
}

int pdc_free(int device, void *dptr) {
    PDC_TRY(use_device(device));
    if (dptr) PDC_HIP(hipFree(dptr));
    return PDC_OK;
}

int pdc_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes) {
    PDC_REQUIRE(bytes >= 0 && (bytes == 0 || (dst && src)), "pdc_memcpy_h2d: bad arguments");
    PDC_TRY(use_device(device));
    if (bytes) PDC_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return PDC_OK;
}

int pdc_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes) {
    PDC_REQUIRE(bytes >= 0 && (bytes == 0 || (dst && src)), "pdc_memcpy_d2h: bad arguments");
    PDC_TRY(use_device(device));
    if (bytes) PDC_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return PDC_OK;
}

int pdc_memset(int device, void *dst, int value, int64_t bytes) {
    PDC_REQUIRE(bytes >= 0 && (bytes == 0 || dst), "pdc_memset: bad arguments");
    PDC_TRY(use_device(device));
    if (bytes) PDC_HIP(hipMemset(dst, value, (size_t)bytes));
    return PDC_OK;
}

int pdc_stream_create(int device, void **stream) {
    PDC_REQUIRE(stream != nullptr, "stream is NULL");
    PDC_TRY(use_device(device));
    hipStream_t s;
    PDC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return PDC_OK;
}

int pdc_stream_destroy(int device, void *stream) {
    PDC_TRY(use_device(device));
    if (stream) {
        PDC_TRY(drop_stream_scratch(device, (hipStream_t)stream));   // its split-mode scratch goes with it
        PDC_HIP(hipStreamDestroy((hipStream_t)stream));
    }
    return PDC_OK;
}

int pdc_stream_sync(int device, void *stream) {
    PDC_TRY(use_device(device));
    PDC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PDC_OK;
}

int pdc_device_sync(int device) {
    PDC_TRY(use_device(device));
    PDC_HIP(hipDeviceSynchronize());
    return PDC_OK;
}

int pdc_event_create(int device, void **event) {
    PDC_REQUIRE(event != nullptr, "event is NULL");
    PDC_TRY(use_device(device));
    hipEvent_t e;
    PDC_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return PDC_OK;
}

int pdc_event_destroy(int device, void *event) {
    PDC_TRY(use_device(device));
    if (event) PDC_HIP(hipEventDestroy((hipEvent_t)event));
    return PDC_OK;
}

int pdc_event_record(int device, void *event, void *stream) {
    PDC_REQUIRE(event != nullptr, "event is NULL");
    PDC_TRY(use_device(device));
    PDC_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return PDC_OK;
}

int pdc_event_elapsed_ms(int device, void *start, void *stop, float *ms) {
    PDC_REQUIRE(start && stop && ms, "pdc_event_elapsed_ms: NULL argument");
    PDC_TRY(use_device(device));
    PDC_HIP(hipEventSynchronize((hipEvent_t)stop));
    PDC_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PDC_OK;
}

int pdc_clock_probe(int device, void *stream, int iters, float *ms, double *wave_instr_per_simd, double *memtime_ratio) {
    PDC_REQUIRE(ms && iters > 0 && iters <= 1000000, "pdc_clock_probe: bad argument");
    PDC_TRY(use_device(device));
    hipStream_t st = (hipStream_t)stream;
    constexpr int kBlocks = 1024, kLds = 40 * 1024;
    void *buf = nullptr;
    PDC_HIP(hipMalloc(&buf, (size_t)kBlocks * 256 * 8 + 64));
    unsigned long long *stamps = (unsigned long long *)((char *)buf + (size_t)kBlocks * 256 * 8);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PDC_OK;
    auto body = [&]() -> int {
        PDC_TRY(allow_dynamic_lds((const void *)clock_probe_kernel, kLds));
        PDC_HIP(hipEventCreate(&e0));
        PDC_HIP(hipEventCreate(&e1));
        clock_probe_kernel<<<kBlocks, 256, kLds, st>>>((double *)buf, stamps, 0.999999, 1e-9, 64);   // (code object loaded)
        PDC_HIP(hipEventRecord(e0, st));
        clock_probe_kernel<<<kBlocks, 256, kLds, st>>>((double *)buf, stamps, 0.999999, 1e-9, iters);
        PDC_HIP(hipEventRecord(e1, st));
        PDC_HIP(hipEventSynchronize(e1));
        PDC_HIP(hipEventElapsedTime(ms, e0, e1));
        unsigned long long h[2] = {0, 0};
        PDC_HIP(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
        // 4 waves per SIMD, each iters x 8 x 16 fmas
        if (wave_instr_per_simd) *wave_instr_per_simd = 4.0 * (double)iters * 128.0;
        if (memtime_ratio) *memtime_ratio = h[1] ? (double)h[0] / (double)h[1] : 0.0;
        return PDC_OK;
    };
    rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return rc;
}

}  // extern "C"
