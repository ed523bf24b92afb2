// Scans sharded over the GPUs of a node.  Every trial frequency / trial period / light curve is
// independent given the (small, replicated) sample set, so a grid is cut into contiguous equal
// slabs, one per device slot; one process drives all devices (hipSetDevice + streams
// per slot); the reference's multiprocessing.Pool fan-out
// (/root/reference/src/periodicity/phase.py:69-70,185-186) is not reproduced.
//
//   GlsPlan    one periodogram, frequency slabs, ONE exchange: an all-gather of the power array - RCCL
//              over xGMI (ncclCommInitAll + grouped in-place ncclAllGather), or, for "loopback" plans
//              (several logical slots on one physical device: how the N > 1 logic runs on a 1-GPU box)
//              and with PDC_PLAN_EXCHANGE=copy, the equivalent device-to-device copies;
//   PhasePlan  PDM / AoV / conditional entropy / StringLength, period slabs, no exchange at all;
//   batches    curves dealt to the slots in contiguous groups (pdc_gls_scan_batch_multi), no exchange.
//
// Everything with a fixed cost - per-slot buffers (grow-only), streams, events, pinned staging, the
// communicators - is created once and kept: by the caller's plan handle, or, for the one-shot `_multi`
// entry points, in a small cache keyed by the device list.
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pdc_internal.h"

using namespace pdc;

#define PDC_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t _r = (call);                                                               \
        if (_r != ncclSuccess) {                                                                \
            set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(_r), __FILE__, __LINE__); \
            return PDC_ERR_RCCL;                                                                \
        }                                                                                       \
    } while (0)

namespace {

// ---- grow-only buffers ---------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    int64_t cap = 0;
};

// (on the current device) hipFree synchronises it, so no kernel still uses the old block
int ensure(DevBuf &b, int64_t bytes) {
    if (bytes < 256) bytes = 256;
    if (b.cap >= bytes) return PDC_OK;
    if (b.p) PDC_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const int64_t want = bytes + bytes / 8;
    PDC_TRY(device_alloc(&b.p, want));
    b.cap = want;
    return PDC_OK;
}

struct PinBuf {   // page-locked host staging: copies from/to it are truly asynchronous
    void *p = nullptr;
    int64_t cap = 0;
};

int ensure(PinBuf &b, int64_t bytes) {
    if (bytes < 256) bytes = 256;
    if (b.cap >= bytes) return PDC_OK;
    if (b.p) PDC_HIP(hipHostFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const int64_t want = bytes + bytes / 8;
    PDC_TRY(pinned_alloc(&b.p, want));
    b.cap = want;
    return PDC_OK;
}

int check_devices(const char *who, const int *devices, int n_devices, bool distinct) {
    PDC_REQUIRE(devices, "%s: NULL device list", who);
    PDC_REQUIRE(n_devices >= 1 && n_devices <= 64, "%s: between 1 and 64 device slots", who);
    int count = 0;
    PDC_TRY(pdc_device_count(&count));
    for (int a = 0; a < n_devices; ++a) {
        PDC_REQUIRE(devices[a] >= 0 && devices[a] < count, "%s: device %d is not one of the %d visible", who,
                    devices[a], count);
        for (int b = a + 1; distinct && b < n_devices; ++b)
            PDC_REQUIRE(devices[a] != devices[b], "%s: device %d listed twice", who, devices[a]);
    }
    return PDC_OK;
}

// contiguous equal slabs: slot i owns [i*per, min((i+1)*per, total)); trailing slots may own nothing
struct Slab {
    int64_t begin, count;
};
Slab slab_of(int64_t total, int n_slots, int i) {
    const int64_t per = (total + n_slots - 1) / n_slots;
    const int64_t b = (int64_t)i * per;
    return {b, b >= total ? 0 : (total - b < per ? total - b : per)};
}

// ======================================================================================================
// GlsPlan
// ======================================================================================================
enum Exchange { EX_NONE = 0, EX_RCCL = 1, EX_COPY = 2 };

struct PlanDev {
    int device = -1;
    void *t = nullptr, *y = nullptr, *dy = nullptr, *work = nullptr;
    void *pow[2] = {nullptr, nullptr};
    hipStream_t compute = nullptr, comm = nullptr;
    hipEvent_t scanned[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
    hipEvent_t k0 = nullptr, k1 = nullptr, uploaded = nullptr;
};

struct GlsPlan {
    std::vector<PlanDev> dev;
    std::vector<ncclComm_t> comms;
    Exchange exchange = EX_NONE;
    bool loopback = false;
    std::string init_error;   // why the RCCL communicators could not be built (the plan then exchanges by copies)
    PinBuf stage;        // (t, y, dy) staged once, then copied to every device asynchronously
    int64_t n_cap = 0, slab_cap = 0, work_cap = 0;
    int64_t n = 0, nf = 0, slab = 0;
    bool has_dy = false, timed = false;
    int gen = 0;         // generation the NEXT scan writes
    int last = -1;       // generation of the latest scan
    bool used[2] = {false, false};
    std::mutex mu;
};

bool force_rccl() {
    static const bool f = [] { const char *e = getenv("PDC_FORCE_RCCL"); return e && e[0] == '1'; }();
    return f;
}

// PDC_FORCE_RCCL_FAIL=1: behave as if ncclCommInitAll had failed (exercises the fallback on any box)
bool force_rccl_fail() {
    static const bool f = [] { const char *e = getenv("PDC_FORCE_RCCL_FAIL"); return e && e[0] == '1'; }();
    return f;
}

bool exchange_by_copy() {
    static const bool f = [] { const char *e = getenv("PDC_PLAN_EXCHANGE"); return e && !strcmp(e, "copy"); }();
    return f;
}

void plan_free(GlsPlan *p) {
    for (PlanDev &d : p->dev) {   // drain every stream before the communicators go
        if (d.device < 0 || hipSetDevice(d.device) != hipSuccess) continue;
        if (d.compute) (void)hipStreamSynchronize(d.compute);
        if (d.comm) (void)hipStreamSynchronize(d.comm);
    }
    for (ncclComm_t c : p->comms)
        if (c) ncclCommDestroy(c);
    for (PlanDev &d : p->dev) {
        if (d.device < 0 || hipSetDevice(d.device) != hipSuccess) continue;
        for (hipEvent_t e : {d.scanned[0], d.scanned[1], d.gathered[0], d.gathered[1], d.k0, d.k1, d.uploaded})
            if (e) (void)hipEventDestroy(e);
        if (d.compute) (void)hipStreamDestroy(d.compute);
        if (d.comm) (void)hipStreamDestroy(d.comm);
        for (void *q : {d.t, d.y, d.dy, d.work, d.pow[0], d.pow[1]})
            if (q) (void)hipFree(q);
    }
    if (p->stage.p) (void)hipHostFree(p->stage.p);
    delete p;
}

int plan_build(GlsPlan *p, const int *devices, int n_devices, int64_t n_max, int64_t nf_max, bool loopback) {
    p->n_cap = n_max;
    p->slab_cap = (nf_max + n_devices - 1) / n_devices;
    p->work_cap = pdc_gls_work_bytes(n_max, 1, p->slab_cap);
    p->loopback = loopback;
    p->dev.resize(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        PlanDev &d = p->dev[i];
        PDC_TRY(use_device(devices[i]));   // PDC_ERR_INVALID for an ordinal outside pdc_device_count()
        d.device = devices[i];
        PDC_HIP(hipStreamCreateWithFlags(&d.compute, hipStreamNonBlocking));
        PDC_HIP(hipStreamCreateWithFlags(&d.comm, hipStreamNonBlocking));
        for (int g = 0; g < 2; ++g) {
            PDC_HIP(hipEventCreateWithFlags(&d.scanned[g], hipEventDisableTiming));
            PDC_HIP(hipEventCreateWithFlags(&d.gathered[g], hipEventDisableTiming));
            PDC_TRY(device_alloc(&d.pow[g], p->slab_cap * n_devices * 8 + 8));
        }
        PDC_HIP(hipEventCreate(&d.k0));
        PDC_HIP(hipEventCreate(&d.k1));
        PDC_HIP(hipEventCreateWithFlags(&d.uploaded, hipEventDisableTiming));
        PDC_TRY(device_alloc(&d.t, n_max * 8 + 8));
        PDC_TRY(device_alloc(&d.y, n_max * 8 + 8));
        PDC_TRY(device_alloc(&d.dy, n_max * 8 + 8));
        PDC_TRY(device_alloc(&d.work, p->work_cap));
    }
    PDC_TRY(ensure(p->stage, 3 * n_max * 8));
    if (loopback || (n_devices > 1 && exchange_by_copy())) {
        p->exchange = n_devices > 1 ? EX_COPY : EX_NONE;
        if (loopback && force_rccl_fail()) {   // (a 1-GPU box: the injected failure lands where a real one would)
            p->init_error = "ncclCommInitAll failed: PDC_FORCE_RCCL_FAIL=1 (injected; loopback plan)";
            fprintf(stderr, "periodicity_hip: WARNING: %s - the plan exchanges slabs by device-to-device copies instead of an RCCL all-gather\n",
                    p->init_error.c_str());
        }
    } else if (n_devices > 1 || force_rccl()) {
        p->comms.assign(n_devices, nullptr);
        const ncclResult_t rc = force_rccl_fail() ? ncclSystemError : ncclCommInitAll(p->comms.data(), n_devices, devices);
        if (rc == ncclSuccess) {
            p->exchange = EX_RCCL;
        } else if (rc == ncclInvalidArgument || rc == ncclInvalidUsage) {
            // a caller bug (a device listed twice, an ordinal that does not exist): an error, not something to paper
            // over with the copy exchange
            for (ncclComm_t &c : p->comms)
                if (c) (void)ncclCommAbort(c);
            p->comms.clear();
            set_error("ncclCommInitAll failed: %s (check the device list)", ncclGetErrorString(rc));
            return PDC_ERR_RCCL;
        } else {
            // No communicator: the job must not die for it.  The all-gather is the same data movement as the copy
            // exchange (slot i pulls slab j from slot j: hipMemcpyPeerAsync over xGMI once peer access is on), so
            // the plan falls back to that - LOUDLY: stderr here, pdc_gls_plan_init_error / plan_info for the caller.
            p->init_error = std::string("ncclCommInitAll failed: ") + (force_rccl_fail() ? "PDC_FORCE_RCCL_FAIL=1 (injected)" : ncclGetErrorString(rc));
            fprintf(stderr, "periodicity_hip: WARNING: %s - the plan exchanges slabs by device-to-device copies instead of an RCCL all-gather\n",
                    p->init_error.c_str());
            for (ncclComm_t &c : p->comms) {   // (communicators a failed ncclCommInitAll left half-made)
                if (c) (void)ncclCommAbort(c);
                c = nullptr;
            }
            p->comms.clear();
            for (int i = 0; i < n_devices; ++i)
                for (int j = 0; j < n_devices; ++j) {
                    if (devices[i] == devices[j]) continue;
                    // (peer access is an optimisation: without it hipMemcpyPeerAsync stages through the host - a failure
                    // to query or enable it must not abort the plan after the fallback has been announced)
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) {
                        (void)hipGetLastError();
                        continue;
                    }
                    PDC_TRY(use_device(devices[i]));
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                        fprintf(stderr, "periodicity_hip: WARNING: peer access %d -> %d not enabled (%s): copies between them go through the host\n",
                                devices[i], devices[j], hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            p->exchange = n_devices > 1 ? EX_COPY : EX_NONE;
        }
    }
    return PDC_OK;
}

int plan_upload(GlsPlan *p, const double *t, const double *y, const double *dy, int64_t n) {
    PDC_REQUIRE(t && y, "gls_plan_upload: t and y must not be NULL");
    PDC_REQUIRE(n >= 0 && n <= p->n_cap, "gls_plan_upload: %lld samples exceed the plan's %lld",
                (long long)n, (long long)p->n_cap);
    // The staging block is free again once the previous upload's copies have run (a scan of the old
    // samples may still be in flight; it reads device memory only).
    for (PlanDev &d : p->dev) {
        PDC_TRY(use_device(d.device));
        PDC_HIP(hipEventSynchronize(d.uploaded));
    }
    // one pass over the caller's (pageable) arrays, then every device is fed from page-locked memory at
    // the same time instead of one blocking pageable copy per device
    double *s = static_cast<double *>(p->stage.p);
    memcpy(s, t, (size_t)n * 8);
    memcpy(s + n, y, (size_t)n * 8);
    if (dy) memcpy(s + 2 * n, dy, (size_t)n * 8);
    for (PlanDev &d : p->dev) {
        PDC_TRY(use_device(d.device));
        PDC_HIP(hipMemcpyAsync(d.t, s, n * 8, hipMemcpyHostToDevice, d.compute));
        PDC_HIP(hipMemcpyAsync(d.y, s + n, n * 8, hipMemcpyHostToDevice, d.compute));
        if (dy) PDC_HIP(hipMemcpyAsync(d.dy, s + 2 * n, n * 8, hipMemcpyHostToDevice, d.compute));
        PDC_HIP(hipEventRecord(d.uploaded, d.compute));
    }
    p->n = n;
    p->has_dy = dy != nullptr;
    return PDC_OK;
}

// The all-gather of generation g as RCCL runs it: grouped, in place, on the communication streams.
// A failing call inside the group still closes the group before the error is returned.
int exchange_rccl(GlsPlan *p, int g, int64_t slab) {
    const int nd = (int)p->dev.size();
    PDC_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < nd && bad == ncclSuccess; ++i) {
        double *buf = (double *)p->dev[i].pow[g];
        bad = ncclAllGather(buf + (int64_t)i * slab, buf, (size_t)slab, ncclDouble, p->comms[i], p->dev[i].comm);
    }
    const ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
        set_error("ncclAllGather failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : end));
        return PDC_ERR_RCCL;
    }
    return PDC_OK;
}

// The same data movement as device-to-device copies: slot i pulls slab j from slot j's buffer, for every
// j, on ITS communication stream, once slot j's scan of this generation has been recorded.
int exchange_copy(GlsPlan *p, int g, int64_t slab) {
    const int nd = (int)p->dev.size();
    for (int i = 0; i < nd; ++i) {
        PlanDev &di = p->dev[i];
        PDC_TRY(use_device(di.device));
        for (int j = 0; j < nd; ++j) {
            if (j == i) continue;
            PlanDev &dj = p->dev[j];
            PDC_HIP(hipStreamWaitEvent(di.comm, dj.scanned[g], 0));
            double *dst = (double *)di.pow[g] + (int64_t)j * slab;
            const double *src = (const double *)dj.pow[g] + (int64_t)j * slab;
            if (di.device == dj.device)
                PDC_HIP(hipMemcpyAsync(dst, src, (size_t)slab * 8, hipMemcpyDeviceToDevice, di.comm));
            else
                PDC_HIP(hipMemcpyPeerAsync(dst, di.device, src, dj.device, (size_t)slab * 8, di.comm));
        }
    }
    return PDC_OK;
}

int plan_scan(GlsPlan *p, double f0, double delta, int64_t nf, int fit_mean, int psd) {
    const int nd = (int)p->dev.size();
    const int64_t slab = (nf + nd - 1) / nd;   // equal counts; the tail of the last slab(s) is padding
    PDC_REQUIRE(nf >= 0 && slab <= p->slab_cap, "gls_plan_scan: %lld frequencies exceed the plan",
                (long long)nf);
    if (nf == 0) return PDC_OK;
    const int g = p->gen;
    for (int i = 0; i < nd; ++i) {
        PlanDev &d = p->dev[i];
        PDC_TRY(use_device(d.device));
        // generation g is free again once its previous gather (if any) has completed: with RCCL the
        // collective's completion on this device's stream covers it; with copies every OTHER slot's
        // pull of this slot's slab has to be through as well
        if (p->used[g] && p->exchange == EX_RCCL) PDC_HIP(hipStreamWaitEvent(d.compute, d.gathered[g], 0));
        if (p->used[g] && p->exchange == EX_COPY)
            for (int j = 0; j < nd; ++j) PDC_HIP(hipStreamWaitEvent(d.compute, p->dev[j].gathered[g], 0));
        const Slab s = slab_of(nf, nd, i);
        double *out = (double *)d.pow[g] + (int64_t)i * slab;
        if (s.count < slab)
            PDC_HIP(hipMemsetAsync(out + s.count, 0, (size_t)((slab - s.count) * 8), d.compute));
        PDC_HIP(hipEventRecord(d.k0, d.compute));
        if (s.count > 0)
            PDC_TRY(pdc_gls_scan_dev(d.device, d.compute, (double *)d.t, (double *)d.y,
                                     p->has_dy ? (double *)d.dy : nullptr, nullptr, p->n, 1, 0, f0,
                                     delta, s.begin, s.count, fit_mean, psd, out, nullptr, nullptr, d.work,
                                     p->work_cap));
        PDC_HIP(hipEventRecord(d.k1, d.compute));
        if (p->exchange != EX_NONE) {
            PDC_HIP(hipEventRecord(d.scanned[g], d.compute));
            PDC_HIP(hipStreamWaitEvent(d.comm, d.scanned[g], 0));
        }
    }
    if (p->exchange == EX_RCCL) PDC_TRY(exchange_rccl(p, g, slab));
    if (p->exchange == EX_COPY) PDC_TRY(exchange_copy(p, g, slab));
    if (p->exchange != EX_NONE) {
        for (int i = 0; i < nd; ++i) {
            PDC_TRY(use_device(p->dev[i].device));
            PDC_HIP(hipEventRecord(p->dev[i].gathered[g], p->dev[i].comm));
        }
    }
    p->used[g] = true;
    p->timed = true;
    p->nf = nf;
    p->slab = slab;
    p->last = g;
    p->gen = g ^ 1;
    return PDC_OK;
}

int plan_wait(GlsPlan *p) {
    for (PlanDev &d : p->dev) {
        PDC_TRY(use_device(d.device));
        PDC_HIP(hipStreamSynchronize(d.compute));
        PDC_HIP(hipStreamSynchronize(d.comm));
    }
    return PDC_OK;
}

int plan_create(const int *devices, int n_devices, int64_t n_max, int64_t nf_max, bool loopback, void **plan) {
    PDC_REQUIRE(plan, "gls_plan_create: NULL argument");
    PDC_REQUIRE(n_max >= 0 && nf_max >= 0, "gls_plan_create: bad size");
    PDC_TRY(check_devices("gls_plan_create", devices, n_devices, !loopback));
    GlsPlan *p = new GlsPlan();
    const int rc = plan_build(p, devices, n_devices, n_max, nf_max, loopback);
    if (rc != PDC_OK) {
        plan_free(p);
        return rc;
    }
    *plan = p;
    return PDC_OK;
}

// the one-shot host entry point keeps its plan between calls
std::mutex g_oneshot_mutex;
GlsPlan *g_oneshot = nullptr;
std::vector<int> g_oneshot_devices;

}  // namespace

extern "C" {

int pdc_gls_plan_create(const int *devices, int n_devices, int64_t n_max, int64_t nf_max, void **plan) {
    return plan_create(devices, n_devices, n_max, nf_max, false, plan);
}

int pdc_gls_plan_create_loopback(int device, int n_slots, int64_t n_max, int64_t nf_max, void **plan) {
    PDC_REQUIRE(n_slots >= 1 && n_slots <= 64, "gls_plan_create_loopback: between 1 and 64 slots");
    const std::vector<int> devices((size_t)n_slots, device);
    return plan_create(devices.data(), n_slots, n_max, nf_max, true, plan);
}

int pdc_gls_plan_info(void *plan, int *n_slots, int *rccl_ranks, int *exchange) {
    PDC_REQUIRE(plan, "gls_plan_info: NULL plan");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    if (n_slots) *n_slots = (int)p->dev.size();
    if (exchange) *exchange = (int)p->exchange;
    if (rccl_ranks) {
        *rccl_ranks = 0;
        if (!p->comms.empty()) PDC_NCCL(ncclCommCount(p->comms[0], rccl_ranks));
    }
    return PDC_OK;
}

int pdc_gls_plan_init_error(void *plan, char *buf, int buf_len) {
    PDC_REQUIRE(plan && buf && buf_len > 0, "gls_plan_init_error: NULL argument");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    snprintf(buf, (size_t)buf_len, "%s", p->init_error.c_str());
    return PDC_OK;
}

int pdc_gls_plan_upload(void *plan, const double *t, const double *y, const double *dy, int64_t n) {
    PDC_REQUIRE(plan, "gls_plan_upload: NULL plan");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return plan_upload(p, t, y, dy, n);
}

int pdc_gls_plan_scan(void *plan, double f0, double delta, int64_t nf, int fit_mean, int psd) {
    PDC_REQUIRE(plan, "gls_plan_scan: NULL plan");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return plan_scan(p, f0, delta, nf, fit_mean, psd);
}

int pdc_gls_plan_wait(void *plan) {
    PDC_REQUIRE(plan, "gls_plan_wait: NULL plan");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return plan_wait(p);
}

int pdc_gls_plan_download(void *plan, double *power_out, int64_t nf, int which) {
    PDC_REQUIRE(plan && (power_out || nf == 0), "gls_plan_download: NULL argument");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    PDC_REQUIRE(p->last >= 0 && nf == p->nf, "gls_plan_download: no scan of %lld frequencies is pending",
                (long long)nf);
    PDC_REQUIRE(which >= 0 && which < (int)p->dev.size(), "gls_plan_download: bad device slot %d", which);
    PDC_TRY(plan_wait(p));
    PlanDev &d = p->dev[which];
    PDC_TRY(use_device(d.device));
    if (nf) PDC_HIP(hipMemcpy(power_out, d.pow[p->last], (size_t)(nf * 8), hipMemcpyDeviceToHost));
    return PDC_OK;
}

int pdc_gls_plan_kernel_ms(void *plan, float *ms) {
    PDC_REQUIRE(plan && ms, "gls_plan_kernel_ms: NULL argument");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    PDC_REQUIRE(p->timed, "gls_plan_kernel_ms: no scan has been enqueued");
    PDC_TRY(use_device(p->dev[0].device));
    PDC_HIP(hipEventSynchronize(p->dev[0].k1));
    PDC_HIP(hipEventElapsedTime(ms, p->dev[0].k0, p->dev[0].k1));
    return PDC_OK;
}

int pdc_gls_plan_slot_ms(void *plan, float *ms_out, int n_slots) {
    PDC_REQUIRE(plan && ms_out, "gls_plan_slot_ms: NULL argument");
    GlsPlan *p = static_cast<GlsPlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    PDC_REQUIRE(p->timed, "gls_plan_slot_ms: no scan has been enqueued");
    PDC_REQUIRE(n_slots == (int)p->dev.size(), "gls_plan_slot_ms: the plan has %d slots, not %d", (int)p->dev.size(),
                n_slots);
    for (int i = 0; i < n_slots; ++i) {
        PlanDev &d = p->dev[i];
        PDC_TRY(use_device(d.device));
        PDC_HIP(hipEventSynchronize(d.k1));
        PDC_HIP(hipEventElapsedTime(&ms_out[i], d.k0, d.k1));
    }
    return PDC_OK;
}

int pdc_gls_plan_destroy(void *plan) {
    if (plan) plan_free(static_cast<GlsPlan *>(plan));
    return PDC_OK;
}

int pdc_gls_scan_multi(const double *t, const double *y, const double *dy, int64_t n, double f0,
                       double delta, int64_t nf, int fit_mean, int psd, double *power_out,
                       const int *devices, int n_devices) {
    PDC_REQUIRE(t && y && devices, "gls_multi: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0 && n_devices >= 1 && n_devices <= 64, "gls_multi: bad size");
    PDC_REQUIRE(power_out || nf == 0, "gls_multi: power_out is NULL");
    if (nf == 0) return PDC_OK;
    std::lock_guard<std::mutex> lk(g_oneshot_mutex);
    const std::vector<int> want(devices, devices + n_devices);
    const int64_t slab = (nf + n_devices - 1) / n_devices;
    if (!g_oneshot || want != g_oneshot_devices || n > g_oneshot->n_cap || slab > g_oneshot->slab_cap) {
        if (g_oneshot) plan_free(g_oneshot);
        g_oneshot = nullptr;
        void *fresh = nullptr;
        // head-room so that a caller sweeping sizes does not rebuild the plan on every call
        PDC_TRY(pdc_gls_plan_create(devices, n_devices, n + n / 8, nf + nf / 8, &fresh));
        g_oneshot = static_cast<GlsPlan *>(fresh);
        g_oneshot_devices = want;
    }
    PDC_TRY(plan_upload(g_oneshot, t, y, dy, n));
    PDC_TRY(plan_scan(g_oneshot, f0, delta, nf, fit_mean, psd));
    PDC_TRY(plan_wait(g_oneshot));
    PlanDev &d0 = g_oneshot->dev[0];
    PDC_TRY(use_device(d0.device));
    PDC_HIP(hipMemcpy(power_out, d0.pow[g_oneshot->last], (size_t)(nf * 8), hipMemcpyDeviceToHost));
    return PDC_OK;
}

}  // extern "C"

// ======================================================================================================
// PhasePlan: PDM / AoV / conditional entropy / StringLength over period slabs; batches over curve groups
// ======================================================================================================
namespace {

enum { B_T = 0, B_V, B_PER, B_OUT, B_WORK, B_DY, B_OFF, B_POW, B_AMAX, B_ARG, B_PICK, B_COUNT };

struct DevSlot {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;
    DevBuf b[B_COUNT];
    bool timed = false;
};

struct PhasePlan {
    std::vector<int> devices;
    std::vector<DevSlot> slot;
    PinBuf pin_in, pin_per, pin_out;   // samples (t, v) / trial periods / results
    int64_t n = 0, n_periods = 0;
    bool scanned = false;
    std::mutex mu;
};

void phase_free(PhasePlan *p) {
    for (DevSlot &s : p->slot) {
        if (s.device < 0 || hipSetDevice(s.device) != hipSuccess) continue;
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        for (hipEvent_t e : {s.k0, s.k1})
            if (e) (void)hipEventDestroy(e);
        if (s.stream) {
            (void)drop_stream_scratch(s.device, s.stream);
            (void)hipStreamDestroy(s.stream);
        }
        for (DevBuf &b : s.b)
            if (b.p) (void)hipFree(b.p);
    }
    for (PinBuf *b : {&p->pin_in, &p->pin_per, &p->pin_out})
        if (b->p) (void)hipHostFree(b->p);
    delete p;
}

int phase_build(PhasePlan *p, const int *devices, int n_devices, int64_t n_max, int64_t n_periods_max) {
    p->devices.assign(devices, devices + n_devices);
    p->slot.resize(n_devices);
    const int64_t slab = (n_periods_max + n_devices - 1) / n_devices;
    for (int i = 0; i < n_devices; ++i) {
        DevSlot &s = p->slot[i];
        PDC_TRY(use_device(devices[i]));
        s.device = devices[i];
        PDC_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        PDC_HIP(hipEventCreate(&s.k0));
        PDC_HIP(hipEventCreate(&s.k1));
        if (n_max > 0) {
            PDC_TRY(ensure(s.b[B_T], n_max * 8));
            PDC_TRY(ensure(s.b[B_V], n_max * 8));
        }
        if (slab > 0) {
            PDC_TRY(ensure(s.b[B_PER], slab * 8));
            PDC_TRY(ensure(s.b[B_OUT], slab * 8));
        }
    }
    if (n_max > 0) PDC_TRY(ensure(p->pin_in, 2 * n_max * 8));
    if (n_periods_max > 0) {
        PDC_TRY(ensure(p->pin_per, n_periods_max * 8));
        PDC_TRY(ensure(p->pin_out, n_periods_max * 8));
    }
    return PDC_OK;
}

int phase_wait(PhasePlan *p) {
    for (DevSlot &s : p->slot) {
        PDC_TRY(use_device(s.device));
        PDC_HIP(hipStreamSynchronize(s.stream));
    }
    return PDC_OK;
}

int phase_upload(PhasePlan *p, const double *t, const double *v, int64_t n) {
    PDC_REQUIRE(t && v, "phase_plan_upload: t and v must not be NULL");
    PDC_REQUIRE(n >= 0, "phase_plan_upload: negative size");
    PDC_TRY(phase_wait(p));   // the staging block and the device copies may still be in use
    PDC_TRY(ensure(p->pin_in, 2 * n * 8));
    double *s = static_cast<double *>(p->pin_in.p);
    memcpy(s, t, (size_t)n * 8);
    memcpy(s + n, v, (size_t)n * 8);
    for (DevSlot &sl : p->slot) {
        PDC_TRY(use_device(sl.device));
        PDC_TRY(ensure(sl.b[B_T], n * 8));
        PDC_TRY(ensure(sl.b[B_V], n * 8));
        PDC_HIP(hipMemcpyAsync(sl.b[B_T].p, s, n * 8, hipMemcpyHostToDevice, sl.stream));
        PDC_HIP(hipMemcpyAsync(sl.b[B_V].p, s + n, n * 8, hipMemcpyHostToDevice, sl.stream));
    }
    p->n = n;
    return PDC_OK;
}

// kind 0 = PDM (v = x), 1 = AoV (v = x; nb phase bins), 2 = conditional entropy (v = magnitude bins; nb x nc
// cells), 3 = StringLength (v = m), 4 = Gregory-Loredo (v unused; nb = m * offsets fine bins, nc = m).  Enqueues on every slot: its slab of the trial periods H2D, the scan,
// the slab of results D2H into the page-locked result block - and returns.
int phase_scan_enqueue(PhasePlan *p, int kind, const double *periods, int64_t n_periods, int nb, int nc,
                       double sigma) {
    const int nd = (int)p->slot.size();
    if (n_periods == 0) return PDC_OK;
    PDC_TRY(ensure(p->pin_per, n_periods * 8));
    PDC_TRY(ensure(p->pin_out, n_periods * 8));
    memcpy(p->pin_per.p, periods, (size_t)n_periods * 8);
    const double *pp = static_cast<const double *>(p->pin_per.p);
    double *po = static_cast<double *>(p->pin_out.p);
    for (int i = 0; i < nd; ++i) {
        const Slab sb = slab_of(n_periods, nd, i);
        if (sb.count == 0) continue;
        DevSlot &s = p->slot[i];
        PDC_TRY(use_device(s.device));
        PDC_REQUIRE(s.b[B_T].p && s.b[B_V].p, "phase_plan_scan: no samples have been uploaded");
        PDC_TRY(ensure(s.b[B_PER], sb.count * 8));
        PDC_TRY(ensure(s.b[B_OUT], sb.count * 8));
        // (the scans that sort every period: the samples are still in the staging block, so the host can see whether
        // this slab's periods need the streamed kernels' bin lists at all - ~12 GB per slot at a million samples)
        const bool sorted_kind = kind == 3 || kind == 5;
        const int hints = sorted_kind ? sorted_scan_hints(kind, static_cast<const double *>(p->pin_in.p), p->n, pp + sb.begin, sb.count) : 0;
        // (round 6: this slot's workspace fitted to PDC_WORK_BUDGET_GB and to this slot's share of what its device has
        // free - eight loopback slots of one GPU used to ask for eight times the built-in caps; the scope stays open
        // over the launch below, so the scan lays the workspace out with the same scale)
        int64_t budget = work_budget();
        if (sorted_kind) {
            int sharing = 0;
            for (int j = i; j < nd; ++j)
                if (p->slot[j].device == s.device && slab_of(n_periods, nd, j).count > 0) ++sharing;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const int64_t avail = (int64_t)(0.9 * ((double)free_b / (double)(sharing > 0 ? sharing : 1) + (double)s.b[B_WORK].cap));
                if (avail > 0 && (budget == 0 || avail < budget)) budget = avail;
            } else {
                (void)hipGetLastError();
            }
        }
        WorkScale ws(sorted_kind ? budget : 0, [&] {
            return sorted_kind ? sorted_scan_work_bytes(kind, p->n, sb.count, hints) : pdc_phase_work_bytes(kind, p->n, sb.count, nb, nc);
        });
        PDC_REQUIRE_FITS(ws, "phase_plan_scan");
        const int64_t wb = ws.need;
        PDC_REQUIRE(wb >= 0, "phase_plan_scan: bad size");
        PDC_TRY(ensure(s.b[B_WORK], wb + 8));
        PDC_HIP(hipMemcpyAsync(s.b[B_PER].p, pp + sb.begin, sb.count * 8, hipMemcpyHostToDevice, s.stream));
        PDC_HIP(hipEventRecord(s.k0, s.stream));
        if (sorted_kind)
            PDC_TRY(sorted_scan_dev(kind, s.device, s.stream, (double *)s.b[B_T].p, (double *)s.b[B_V].p, p->n, (double *)s.b[B_PER].p,
                                    sb.count, sigma, (double *)s.b[B_OUT].p, s.b[B_WORK].p, s.b[B_WORK].cap, hints));
        else
            PDC_TRY(pdc_phase_scan_dev(kind, s.device, s.stream, (double *)s.b[B_T].p, (double *)s.b[B_V].p, p->n,
                                       (double *)s.b[B_PER].p, sb.count, nb, nc, sigma, (double *)s.b[B_OUT].p,
                                       s.b[B_WORK].p, s.b[B_WORK].cap));
        PDC_HIP(hipEventRecord(s.k1, s.stream));
        s.timed = true;
        PDC_HIP(hipMemcpyAsync(po + sb.begin, s.b[B_OUT].p, sb.count * 8, hipMemcpyDeviceToHost, s.stream));
    }
    return PDC_OK;
}

// A scan counts as pending only once EVERY slot has been enqueued: a failure half-way drains what is in
// flight (copies from / into the staging blocks) and leaves the plan with nothing to download.
int phase_scan(PhasePlan *p, int kind, const double *periods, int64_t n_periods, int nb, int nc, double sigma) {
    PDC_REQUIRE(kind >= 0 && kind <= 5, "phase_plan_scan: kind must be 0 (PDM), 1 (AoV), 2 (conditional "
                                        "entropy), 3 (StringLength), 4 (Gregory-Loredo) or 5 (Supersmoother)");
    PDC_REQUIRE((periods || n_periods == 0) && n_periods >= 0, "phase_plan_scan: bad period grid");
    PDC_TRY(phase_wait(p));   // the period / result staging blocks of the previous scan are free after this
    p->scanned = false;
    for (DevSlot &s : p->slot) s.timed = false;
    const int rc = phase_scan_enqueue(p, kind, periods, n_periods, nb, nc, sigma);
    if (rc != PDC_OK) {
        const std::string why = pdc_last_error();
        (void)phase_wait(p);
        for (DevSlot &s : p->slot) s.timed = false;
        set_error("%s", why.c_str());
        return rc;
    }
    p->n_periods = n_periods;
    p->scanned = true;
    return PDC_OK;
}

int phase_download(PhasePlan *p, double *out, int64_t n_periods) {
    PDC_REQUIRE(p->scanned && n_periods == p->n_periods, "phase_plan_download: no scan of %lld periods is pending",
                (long long)n_periods);
    PDC_REQUIRE(out || n_periods == 0, "phase_plan_download: NULL argument");
    PDC_TRY(phase_wait(p));
    if (n_periods) memcpy(out, p->pin_out.p, (size_t)n_periods * 8);
    return PDC_OK;
}

// one-shot callers: a few plans cached by device list, most recently used last
std::mutex g_phase_mutex;
std::vector<PhasePlan *> g_phase_cache;
constexpr size_t kPhaseCache = 4;

int cached_phase_plan(const char *who, const int *devices, int n_devices, PhasePlan **out) {
    PDC_TRY(check_devices(who, devices, n_devices, false));
    const std::vector<int> want(devices, devices + n_devices);
    for (size_t i = 0; i < g_phase_cache.size(); ++i)
        if (g_phase_cache[i]->devices == want) {
            PhasePlan *p = g_phase_cache[i];
            g_phase_cache.erase(g_phase_cache.begin() + (long)i);
            g_phase_cache.push_back(p);
            *out = p;
            return PDC_OK;
        }
    while (g_phase_cache.size() >= kPhaseCache) {
        phase_free(g_phase_cache.front());
        g_phase_cache.erase(g_phase_cache.begin());
    }
    PhasePlan *p = new PhasePlan();
    const int rc = phase_build(p, devices, n_devices, 0, 0);
    if (rc != PDC_OK) {
        phase_free(p);
        return rc;
    }
    g_phase_cache.push_back(p);
    *out = p;
    return PDC_OK;
}

int phase_multi_entry(int kind, const char *who, const double *t, const double *v, int64_t n,
                      const double *periods, int64_t n_periods, int nb, int nc, double sigma,
                      double *out, const int *devices, int n_devices) {
    PDC_REQUIRE(t && (v || kind == 4) && devices && (periods || n_periods == 0) && (out || n_periods == 0),
                "%s: NULL argument", who);
    if (kind == 4) v = t;   // (the plan replicates two arrays; arrival times alone are scanned)
    PDC_REQUIRE(n >= 0 && n_periods >= 0 && n_devices >= 1 && n_devices <= 64, "%s: bad size", who);
    if (kind == 2)
        for (int64_t i = 0; i < n; ++i)
            PDC_REQUIRE(v[i] >= 0.0 && v[i] < (double)nc, "%s: mag_bin[%lld] = %g is not a bin index in 0 .. %d",
                        who, (long long)i, v[i], nc - 1);
    if (n_periods == 0) return PDC_OK;
    std::lock_guard<std::mutex> lk(g_phase_mutex);
    PhasePlan *p = nullptr;
    PDC_TRY(cached_phase_plan(who, devices, n_devices, &p));
    PDC_TRY(phase_upload(p, t, v, n));
    PDC_TRY(phase_scan(p, kind, periods, n_periods, nb, nc, sigma));
    return phase_download(p, out, n_periods);
}

}  // namespace

extern "C" {

int pdc_phase_plan_create(const int *devices, int n_devices, int64_t n_max, int64_t n_periods_max, void **plan) {
    PDC_REQUIRE(plan && n_max >= 0 && n_periods_max >= 0, "phase_plan_create: bad argument");
    PDC_TRY(check_devices("phase_plan_create", devices, n_devices, false));
    PhasePlan *p = new PhasePlan();
    const int rc = phase_build(p, devices, n_devices, n_max, n_periods_max);
    if (rc != PDC_OK) {
        phase_free(p);
        return rc;
    }
    *plan = p;
    return PDC_OK;
}

int pdc_phase_plan_upload(void *plan, const double *t, const double *v, int64_t n) {
    PDC_REQUIRE(plan, "phase_plan_upload: NULL plan");
    PhasePlan *p = static_cast<PhasePlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return phase_upload(p, t, v, n);
}

int pdc_phase_plan_scan(void *plan, int kind, const double *periods, int64_t n_periods, int nb, int nc,
                        double sigma) {
    PDC_REQUIRE(plan, "phase_plan_scan: NULL plan");
    PhasePlan *p = static_cast<PhasePlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return phase_scan(p, kind, periods, n_periods, nb, nc, sigma);
}

int pdc_phase_plan_wait(void *plan) {
    PDC_REQUIRE(plan, "phase_plan_wait: NULL plan");
    PhasePlan *p = static_cast<PhasePlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return phase_wait(p);
}

int pdc_phase_plan_download(void *plan, double *out, int64_t n_periods) {
    PDC_REQUIRE(plan, "phase_plan_download: NULL plan");
    PhasePlan *p = static_cast<PhasePlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    return phase_download(p, out, n_periods);
}

int pdc_phase_plan_kernel_ms(void *plan, float *ms) {
    PDC_REQUIRE(plan && ms, "phase_plan_kernel_ms: NULL argument");
    PhasePlan *p = static_cast<PhasePlan *>(plan);
    std::lock_guard<std::mutex> lk(p->mu);
    float worst = -1.0f;
    for (DevSlot &s : p->slot) {
        if (!s.timed) continue;
        PDC_TRY(use_device(s.device));
        PDC_HIP(hipEventSynchronize(s.k1));
        float one = 0.0f;
        PDC_HIP(hipEventElapsedTime(&one, s.k0, s.k1));
        worst = one > worst ? one : worst;
    }
    PDC_REQUIRE(worst >= 0.0f, "phase_plan_kernel_ms: no scan has been enqueued");
    *ms = worst;
    return PDC_OK;
}

int pdc_phase_plan_destroy(void *plan) {
    if (plan) phase_free(static_cast<PhasePlan *>(plan));
    return PDC_OK;
}

int pdc_pdm_scan_multi(const double *t, const double *x, int64_t n, const double *periods,
                       int64_t n_periods, int nb, int nc, double sigma, double *theta_out,
                       const int *devices, int n_devices) {
    return phase_multi_entry(0, "pdm_multi", t, x, n, periods, n_periods, nb, nc, sigma, theta_out,
                             devices, n_devices);
}

int pdc_aov_scan_multi(const double *t, const double *x, int64_t n, const double *periods, int64_t n_periods,
                       int n_bins, double *theta_out, const int *devices, int n_devices) {
    return phase_multi_entry(1, "aov_multi", t, x, n, periods, n_periods, n_bins, 1, 1.0, theta_out, devices,
                             n_devices);
}

int pdc_cond_entropy_scan_multi(const double *t, const double *mag_bin, int64_t n, const double *periods,
                                int64_t n_periods, int n_phase, int n_mag, double *entropy_out,
                                const int *devices, int n_devices) {
    return phase_multi_entry(2, "cond_entropy_multi", t, mag_bin, n, periods, n_periods, n_phase, n_mag, 1.0,
                             entropy_out, devices, n_devices);
}

int pdc_gl_scan_multi(const double *t, int64_t n, const double *periods, int64_t n_periods, int m, int n_offsets,
                      double *log_s_out, const int *devices, int n_devices) {
    PDC_REQUIRE(m >= 1 && n_offsets >= 1 && (int64_t)m * n_offsets <= 190, "gregory_loredo: m * n_offsets must be 1..190");
    return phase_multi_entry(4, "gl_multi", t, nullptr, n, periods, n_periods, m * n_offsets, m, 1.0, log_s_out,
                             devices, n_devices);
}

int pdc_supersmoother_scan_multi(const double *t, const double *y, int64_t n, const double *periods, int64_t n_periods,
                                 double alpha, double *stat_out, const int *devices, int n_devices) {
    return phase_multi_entry(5, "supersmoother_multi", t, y, n, periods, n_periods, 1, 1, alpha, stat_out, devices,
                             n_devices);
}

int pdc_stringlength_scan_multi(const double *t, const double *m, int64_t n,
                                const double *periods, int64_t n_periods, double *ell_out,
                                const int *devices, int n_devices) {
    return phase_multi_entry(3, "stringlength_multi", t, m, n, periods, n_periods, 1, 1, 0.0, ell_out,
                             devices, n_devices);
}

// Batch of light curves dealt to the device slots in contiguous groups of ceil(B / slots) curves (SURVEY.md
// §8e: "Batched C3 can alternatively shard over curves (no exchange at all)"); the shape of GLS.bootstrap
// over several GPUs when shared_t != 0 (/root/reference/src/periodicity/spectral.py:140-152).
int pdc_gls_scan_batch_multi(const double *t, const double *y, const double *dy, const int64_t *offsets,
                             int64_t n_curves, int shared_t, double f0, double delta, int64_t nf, int fit_mean,
                             int psd, double *power_out, double *amax_out, int64_t *argmax_out,
                             const int *devices, int n_devices) {
    PDC_REQUIRE(t && y && offsets && devices, "gls_batch_multi: NULL argument");
    PDC_REQUIRE(n_curves >= 1, "gls_batch_multi: a batch holds at least one curve (n_curves = %lld)", (long long)n_curves);
    PDC_REQUIRE(nf >= 0, "gls_batch_multi: negative size");
    PDC_REQUIRE(power_out || amax_out || argmax_out, "gls_batch_multi: no output requested");
    PDC_REQUIRE(offsets[0] == 0, "gls_batch_multi: offsets[0] must be 0");
    for (int64_t b = 0; b < n_curves; ++b) {
        PDC_REQUIRE(offsets[b + 1] >= offsets[b], "gls_batch_multi: offsets must be non-decreasing");
        PDC_REQUIRE(!shared_t || offsets[b + 1] - offsets[b] == offsets[1] - offsets[0],
                    "gls_batch_multi: with a shared time axis every curve must have the same length");
    }
    std::lock_guard<std::mutex> lk(g_phase_mutex);
    PhasePlan *p = nullptr;
    PDC_TRY(cached_phase_plan("gls_batch_multi", devices, n_devices, &p));
    PDC_TRY(phase_wait(p));
    if (nf == 0) return PDC_OK;
    const int nd = n_devices;
    std::vector<std::vector<int64_t>> rebased_of((size_t)nd);   // alive until the final wait
    // launches first, on every slot; the results come back afterwards (a copy into the caller's pageable
    // arrays blocks the host until that slot is done, and issued inside this loop it would run the
    // devices one after another).  Whatever fails half-way, every slot already enqueued is drained before
    // the call returns: the copies read `rebased_of` and the caller's arrays.
    auto enqueue = [&]() -> int {
    for (int i = 0; i < nd; ++i) {
        const Slab sb = slab_of(n_curves, nd, i);
        if (sb.count == 0) continue;
        DevSlot &s = p->slot[i];
        PDC_TRY(use_device(s.device));
        const int64_t s0 = offsets[sb.begin], s1 = offsets[sb.begin + sb.count];
        const int64_t n_total = s1 - s0;
        const int64_t n_t = shared_t ? offsets[1] : n_total;
        const int64_t wb = pdc_gls_work_bytes(n_total, sb.count, nf);
        PDC_REQUIRE(wb >= 0, "gls_batch_multi: bad size");
        PDC_TRY(ensure(s.b[B_T], n_t * 8));
        PDC_TRY(ensure(s.b[B_V], n_total * 8));
        if (dy) PDC_TRY(ensure(s.b[B_DY], n_total * 8));
        PDC_TRY(ensure(s.b[B_OFF], (sb.count + 1) * 8));
        if (power_out) PDC_TRY(ensure(s.b[B_POW], sb.count * nf * 8));
        if (amax_out) PDC_TRY(ensure(s.b[B_AMAX], sb.count * 8));
        if (argmax_out) PDC_TRY(ensure(s.b[B_ARG], sb.count * 8));
        PDC_TRY(ensure(s.b[B_WORK], wb));
        std::vector<int64_t> &rebased = rebased_of[(size_t)i];
        rebased.resize((size_t)sb.count + 1);
        for (int64_t b = 0; b <= sb.count; ++b) rebased[(size_t)b] = offsets[sb.begin + b] - s0;
        PDC_HIP(hipMemcpyAsync(s.b[B_T].p, shared_t ? t : t + s0, n_t * 8, hipMemcpyHostToDevice, s.stream));
        PDC_HIP(hipMemcpyAsync(s.b[B_V].p, y + s0, n_total * 8, hipMemcpyHostToDevice, s.stream));
        if (dy) PDC_HIP(hipMemcpyAsync(s.b[B_DY].p, dy + s0, n_total * 8, hipMemcpyHostToDevice, s.stream));
        PDC_HIP(hipMemcpyAsync(s.b[B_OFF].p, rebased.data(), (sb.count + 1) * 8, hipMemcpyHostToDevice, s.stream));
        PDC_TRY(pdc_gls_scan_dev(s.device, s.stream, (double *)s.b[B_T].p, (double *)s.b[B_V].p,
                                 dy ? (double *)s.b[B_DY].p : nullptr, (int64_t *)s.b[B_OFF].p, n_total, sb.count,
                                 shared_t, f0, delta, 0, nf, fit_mean, psd,
                                 power_out ? (double *)s.b[B_POW].p : nullptr,
                                 amax_out ? (double *)s.b[B_AMAX].p : nullptr,
                                 argmax_out ? (int64_t *)s.b[B_ARG].p : nullptr, s.b[B_WORK].p, s.b[B_WORK].cap));
    }
    for (int i = 0; i < nd; ++i) {
        const Slab sb = slab_of(n_curves, nd, i);
        if (sb.count == 0) continue;
        DevSlot &s = p->slot[i];
        PDC_TRY(use_device(s.device));
        if (power_out)
            PDC_HIP(hipMemcpyAsync(power_out + sb.begin * nf, s.b[B_POW].p, sb.count * nf * 8, hipMemcpyDeviceToHost,
                                   s.stream));
        if (amax_out)
            PDC_HIP(hipMemcpyAsync(amax_out + sb.begin, s.b[B_AMAX].p, sb.count * 8, hipMemcpyDeviceToHost, s.stream));
        if (argmax_out)
            PDC_HIP(hipMemcpyAsync(argmax_out + sb.begin, s.b[B_ARG].p, sb.count * 8, hipMemcpyDeviceToHost, s.stream));
    }
    return PDC_OK;
    };
    const int rc = enqueue();
    if (rc != PDC_OK) {
        const std::string why = pdc_last_error();
        (void)phase_wait(p);
        set_error("%s", why.c_str());
        return rc;
    }
    return phase_wait(p);
}

// GLS.bootstrap (/root/reference/src/periodicity/spectral.py:140-152) with the replicates given BY INDEX:
// picks[b n + i] = the sample replicate b draws at position i (rng.integers(0, n, n) per replicate, drawn
// by the caller exactly as upstream draws them).  Only (t, y, dy) of the ONE curve and the 4-byte indices
// cross PCIe; the prologue on the device gathers y[picks], dy[picks] while it builds the weight table.
// Replicates are dealt to the device slots in contiguous groups (no exchange); every replicate's NaN-aware
// maximum (and its bin) comes back.  method 0 = exact direct sums (f0, delta = grid start and step),
// 1 = the reference's FFT/extirpolation path (one device).
int pdc_gls_bootstrap(const double *t, const double *y, const double *dy, int64_t n, const int32_t *picks,
                      int64_t n_boot, double f0, double delta, int64_t nf, int fit_mean, int psd, int method,
                      double *amax_out, int64_t *argmax_out, const int *devices, int n_devices) {
    PDC_REQUIRE(t && y && devices && (picks || n_boot == 0 || n == 0), "gls_bootstrap: NULL argument");
    PDC_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && n_boot >= 0 && nf >= 0, "gls_bootstrap: bad size");
    PDC_REQUIRE(method == 0 || method == 1, "gls_bootstrap: method must be 0 (direct sums) or 1 (FFT path)");
    PDC_REQUIRE(amax_out || argmax_out || n_boot == 0, "gls_bootstrap: no output requested");
    if (n_boot == 0 || nf == 0) return PDC_OK;
    {   // (one vectorisable pass; an index outside the curve would be an out-of-bounds read on the device)
        unsigned worst = 0;
        for (int64_t i = 0; i < n_boot * n; ++i) worst = (unsigned)picks[i] > worst ? (unsigned)picks[i] : worst;
        PDC_REQUIRE(n_boot * n == 0 || (int64_t)worst < n, "gls_bootstrap: a pick (%u as unsigned) is not a sample index "
                    "in 0 .. %lld", worst, (long long)n - 1);
    }
    if (method == 1) {
        PDC_TRY(check_devices("gls_bootstrap", devices, n_devices, false));
        return gls_bootstrap_fft(t, y, dy, n, picks, n_boot, f0, delta, nf, fit_mean, psd, amax_out, argmax_out,
                                 devices[0]);
    }
    std::lock_guard<std::mutex> lk(g_phase_mutex);
    PhasePlan *p = nullptr;
    PDC_TRY(cached_phase_plan("gls_bootstrap", devices, n_devices, &p));
    PDC_TRY(phase_wait(p));
    const int nd = n_devices;
    auto enqueue = [&]() -> int {
        for (int i = 0; i < nd; ++i) {
            const Slab sb = slab_of(n_boot, nd, i);
            if (sb.count == 0) continue;
            DevSlot &s = p->slot[i];
            PDC_TRY(use_device(s.device));
            const int64_t wb = pdc_gls_bootstrap_work_bytes(n, sb.count, nf);
            PDC_REQUIRE(wb >= 0, "gls_bootstrap: bad size");
            PDC_TRY(ensure(s.b[B_T], n * 8));
            PDC_TRY(ensure(s.b[B_V], n * 8));
            if (dy) PDC_TRY(ensure(s.b[B_DY], n * 8));
            PDC_TRY(ensure(s.b[B_PICK], sb.count * n * 4));
            PDC_TRY(ensure(s.b[B_AMAX], sb.count * 8));
            PDC_TRY(ensure(s.b[B_ARG], sb.count * 8));
            PDC_TRY(ensure(s.b[B_WORK], wb));
            PDC_HIP(hipMemcpyAsync(s.b[B_T].p, t, n * 8, hipMemcpyHostToDevice, s.stream));
            PDC_HIP(hipMemcpyAsync(s.b[B_V].p, y, n * 8, hipMemcpyHostToDevice, s.stream));
            if (dy) PDC_HIP(hipMemcpyAsync(s.b[B_DY].p, dy, n * 8, hipMemcpyHostToDevice, s.stream));
            PDC_HIP(hipMemcpyAsync(s.b[B_PICK].p, picks + sb.begin * n, sb.count * n * 4, hipMemcpyHostToDevice,
                                   s.stream));
            PDC_TRY(pdc_gls_bootstrap_dev(s.device, s.stream, (double *)s.b[B_T].p, (double *)s.b[B_V].p,
                                          dy ? (double *)s.b[B_DY].p : nullptr, n, (int32_t *)s.b[B_PICK].p, sb.count,
                                          f0, delta, nf, fit_mean, psd, (double *)s.b[B_AMAX].p,
                                          (int64_t *)s.b[B_ARG].p, s.b[B_WORK].p, s.b[B_WORK].cap));
        }
        for (int i = 0; i < nd; ++i) {
            const Slab sb = slab_of(n_boot, nd, i);
            if (sb.count == 0) continue;
            DevSlot &s = p->slot[i];
            PDC_TRY(use_device(s.device));
            if (amax_out)
                PDC_HIP(hipMemcpyAsync(amax_out + sb.begin, s.b[B_AMAX].p, sb.count * 8, hipMemcpyDeviceToHost, s.stream));
            if (argmax_out)
                PDC_HIP(hipMemcpyAsync(argmax_out + sb.begin, s.b[B_ARG].p, sb.count * 8, hipMemcpyDeviceToHost, s.stream));
        }
        return PDC_OK;
    };
    const int rc = enqueue();
    if (rc != PDC_OK) {
        const std::string why = pdc_last_error();
        (void)phase_wait(p);
        set_error("%s", why.c_str());
        return rc;
    }
    return phase_wait(p);
}

}  // extern "C"

// Frees the plans the one-shot `_multi` entry points keep between calls (pdc_release()).
void pdc::release_multi() {
    {
        std::lock_guard<std::mutex> lk(g_oneshot_mutex);
        if (g_oneshot) plan_free(g_oneshot);
        g_oneshot = nullptr;
        g_oneshot_devices.clear();
    }
    std::lock_guard<std::mutex> lk(g_phase_mutex);
    for (PhasePlan *p : g_phase_cache) phase_free(p);
    g_phase_cache.clear();
}
