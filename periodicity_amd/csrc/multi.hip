// One periodogram sharded over the GPUs of a node: every trial frequency is independent given
// the (small, replicated) sample set, so the grid is cut into contiguous equal slabs, one per
// device, and the only exchange is one RCCL all-gather of the power array over xGMI
// (SURVEY.md §8e).  One process drives all devices (hipSetDevice + one stream per device,
// ncclCommInitAll + grouped calls); the reference's multiprocessing.Pool fan-out
// (/root/reference/src/periodicity/phase.py:69-70,185-186) is not reproduced.
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
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

// A multi-GPU GLS plan: everything with a fixed cost (per-device buffers, two streams and the events
// per device, the RCCL communicators) is created once and reused by every scan (SURVEY.md §8e
// "create the communicator once and cache it").  Outputs are double-buffered: the all-gather of scan
// i runs on the communication streams while the compute streams already scan i+1.
struct PlanDev {
    int device = -1;
    void *t = nullptr, *y = nullptr, *dy = nullptr, *work = nullptr;
    void *pow[2] = {nullptr, nullptr};
    hipStream_t compute = nullptr, comm = nullptr;
    hipEvent_t scanned[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
    hipEvent_t k0 = nullptr, k1 = nullptr;
};

struct GlsPlan {
    std::vector<PlanDev> dev;
    std::vector<ncclComm_t> comms;
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
        for (hipEvent_t e : {d.scanned[0], d.scanned[1], d.gathered[0], d.gathered[1], d.k0, d.k1})
            if (e) (void)hipEventDestroy(e);
        if (d.compute) (void)hipStreamDestroy(d.compute);
        if (d.comm) (void)hipStreamDestroy(d.comm);
        for (void *q : {d.t, d.y, d.dy, d.work, d.pow[0], d.pow[1]})
            if (q) (void)hipFree(q);
    }
    delete p;
}

int plan_build(GlsPlan *p, const int *devices, int n_devices, int64_t n_max, int64_t nf_max) {
    p->n_cap = n_max;
    p->slab_cap = (nf_max + n_devices - 1) / n_devices;
    p->work_cap = pdc_gls_work_bytes(n_max, 1, p->slab_cap);
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
            PDC_HIP(hipMalloc(&d.pow[g], (size_t)(p->slab_cap * n_devices * 8 + 8)));
        }
        PDC_HIP(hipEventCreate(&d.k0));
        PDC_HIP(hipEventCreate(&d.k1));
        PDC_HIP(hipMalloc(&d.t, (size_t)(n_max * 8 + 8)));
        PDC_HIP(hipMalloc(&d.y, (size_t)(n_max * 8 + 8)));
        PDC_HIP(hipMalloc(&d.dy, (size_t)(n_max * 8 + 8)));
        PDC_HIP(hipMalloc(&d.work, (size_t)p->work_cap));
    }
    if (n_devices > 1 || force_rccl()) {
        p->comms.assign(n_devices, nullptr);
        PDC_NCCL(ncclCommInitAll(p->comms.data(), n_devices, devices));
    }
    return PDC_OK;
}

int plan_upload(GlsPlan *p, const double *t, const double *y, const double *dy, int64_t n) {
    PDC_REQUIRE(t && y, "gls_plan_upload: t and y must not be NULL");
    PDC_REQUIRE(n >= 0 && n <= p->n_cap, "gls_plan_upload: %lld samples exceed the plan's %lld",
                (long long)n, (long long)p->n_cap);
    for (PlanDev &d : p->dev) {
        PDC_TRY(use_device(d.device));
        PDC_HIP(hipMemcpyAsync(d.t, t, n * 8, hipMemcpyHostToDevice, d.compute));
        PDC_HIP(hipMemcpyAsync(d.y, y, n * 8, hipMemcpyHostToDevice, d.compute));
        if (dy) PDC_HIP(hipMemcpyAsync(d.dy, dy, n * 8, hipMemcpyHostToDevice, d.compute));
    }
    p->n = n;
    p->has_dy = dy != nullptr;
    return PDC_OK;
}

int plan_scan(GlsPlan *p, double f0, double delta, int64_t nf, int fit_mean, int psd) {
    const int nd = (int)p->dev.size();
    const int64_t slab = (nf + nd - 1) / nd;   // equal counts; the tail of the last slab is padding
    PDC_REQUIRE(nf >= 0 && slab <= p->slab_cap, "gls_plan_scan: %lld frequencies exceed the plan",
                (long long)nf);
    if (nf == 0) return PDC_OK;
    const int g = p->gen;
    for (int i = 0; i < nd; ++i) {
        PlanDev &d = p->dev[i];
        PDC_TRY(use_device(d.device));
        // generation g is free again once its previous gather (if any) has completed
        if (p->used[g] && !p->comms.empty()) PDC_HIP(hipStreamWaitEvent(d.compute, d.gathered[g], 0));
        const int64_t j0 = (int64_t)i * slab;
        const int64_t cnt = j0 >= nf ? 0 : (nf - j0 < slab ? nf - j0 : slab);
        double *out = (double *)d.pow[g] + j0;
        if (cnt < slab)
            PDC_HIP(hipMemsetAsync(out + cnt, 0, (size_t)((slab - cnt) * 8), d.compute));
        if (i == 0) PDC_HIP(hipEventRecord(d.k0, d.compute));
        if (cnt > 0)
            PDC_TRY(pdc_gls_scan_dev(d.device, d.compute, (double *)d.t, (double *)d.y,
                                     p->has_dy ? (double *)d.dy : nullptr, nullptr, p->n, 1, 0, f0,
                                     delta, j0, cnt, fit_mean, psd, out, nullptr, nullptr, d.work,
                                     p->work_cap));
        if (i == 0) PDC_HIP(hipEventRecord(d.k1, d.compute));
        if (!p->comms.empty()) {
            PDC_HIP(hipEventRecord(d.scanned[g], d.compute));
            PDC_HIP(hipStreamWaitEvent(d.comm, d.scanned[g], 0));
        }
    }
    if (!p->comms.empty()) {
        PDC_NCCL(ncclGroupStart());
        for (int i = 0; i < nd; ++i) {
            double *buf = (double *)p->dev[i].pow[g];
            PDC_NCCL(ncclAllGather(buf + (int64_t)i * slab, buf, (size_t)slab, ncclDouble, p->comms[i],
                                   p->dev[i].comm));
        }
        PDC_NCCL(ncclGroupEnd());
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

// the one-shot host entry point keeps its plan between calls
std::mutex g_oneshot_mutex;
GlsPlan *g_oneshot = nullptr;
std::vector<int> g_oneshot_devices;

}  // namespace

extern "C" {

int pdc_gls_plan_create(const int *devices, int n_devices, int64_t n_max, int64_t nf_max, void **plan) {
    PDC_REQUIRE(devices && plan, "gls_plan_create: NULL argument");
    PDC_REQUIRE(n_devices >= 1 && n_devices <= 64 && n_max >= 0 && nf_max >= 0, "gls_plan_create: bad size");
    int count = 0;
    PDC_TRY(pdc_device_count(&count));
    for (int a = 0; a < n_devices; ++a) {
        PDC_REQUIRE(devices[a] >= 0 && devices[a] < count, "gls_plan_create: device %d is not one of the %d visible",
                    devices[a], count);
        for (int b = a + 1; b < n_devices; ++b)
            PDC_REQUIRE(devices[a] != devices[b], "gls_plan_create: device %d listed twice", devices[a]);
    }
    GlsPlan *p = new GlsPlan();
    const int rc = plan_build(p, devices, n_devices, n_max, nf_max);
    if (rc != PDC_OK) {
        plan_free(p);
        return rc;
    }
    *plan = p;
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
    PDC_REQUIRE(which == 0 || !p->comms.empty() || p->dev.size() == 1, "gls_plan_download: bad slot");
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
    if (g_oneshot->comms.empty() && n_devices > 1) return PDC_ERR_RCCL;  // (cannot happen: comms exist for n > 1)
    PDC_HIP(hipMemcpy(power_out, d0.pow[g_oneshot->last], (size_t)(nf * 8), hipMemcpyDeviceToHost));
    return PDC_OK;
}

}  // extern "C"

// Frees the plan pdc_gls_scan_multi keeps between calls (pdc_release()).
void pdc::release_multi() {
    std::lock_guard<std::mutex> lk(g_oneshot_mutex);
    if (g_oneshot) plan_free(g_oneshot);
    g_oneshot = nullptr;
    g_oneshot_devices.clear();
}

// ---- phase scans over several GPUs ------------------------------------------------------------------
// The trial-period grid is cut into contiguous slabs, one per listed device (a device may be listed
// more than once: its slabs then run on separate streams); samples are replicated, every slab comes
// back with its own D2H copy, and there is no exchange step at all.
namespace {

struct PhaseSlot {
    void *t = nullptr, *v = nullptr, *periods = nullptr, *out = nullptr, *work = nullptr;
    hipStream_t stream = nullptr;
};

void free_slots(std::vector<PhaseSlot> &slots, const int *devices) {
    for (size_t d = 0; d < slots.size(); ++d) {
        if (hipSetDevice(devices[d]) != hipSuccess) continue;
        if (slots[d].stream) (void)hipStreamDestroy(slots[d].stream);
        for (void *p : {slots[d].t, slots[d].v, slots[d].periods, slots[d].out, slots[d].work})
            if (p) (void)hipFree(p);
    }
}

// kind 0 = PDM (v = x), 1 = StringLength (v = m)
int phase_multi(int kind, const double *t, const double *v, int64_t n, const double *periods,
                int64_t n_periods, int nb, int nc, double sigma, double *out, const int *devices,
                int n_devices, std::vector<PhaseSlot> &slots) {
    const int64_t slab = (n_periods + n_devices - 1) / n_devices;
    for (int d = 0; d < n_devices; ++d) {
        const int64_t p0 = (int64_t)d * slab;
        const int64_t cnt = p0 >= n_periods ? 0 : (n_periods - p0 < slab ? n_periods - p0 : slab);
        if (cnt == 0) continue;
        PDC_TRY(use_device(devices[d]));
        PhaseSlot &s = slots[d];
        PDC_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        PDC_HIP(hipMalloc(&s.t, (size_t)(n * 8 + 8)));
        PDC_HIP(hipMalloc(&s.v, (size_t)(n * 8 + 8)));
        PDC_HIP(hipMalloc(&s.periods, (size_t)(cnt * 8)));
        PDC_HIP(hipMalloc(&s.out, (size_t)(cnt * 8)));
        PDC_HIP(hipMemcpyAsync(s.t, t, n * 8, hipMemcpyHostToDevice, s.stream));
        PDC_HIP(hipMemcpyAsync(s.v, v, n * 8, hipMemcpyHostToDevice, s.stream));
        PDC_HIP(hipMemcpyAsync(s.periods, periods + p0, cnt * 8, hipMemcpyHostToDevice, s.stream));
        if (kind == 0) {
            PDC_TRY(pdc_pdm_scan_dev(devices[d], s.stream, (double *)s.t, (double *)s.v, n,
                                     (double *)s.periods, cnt, nb, nc, sigma, (double *)s.out));
        } else {
            const int64_t wb = pdc_stringlength_work_bytes(n, cnt);
            PDC_REQUIRE(wb >= 0, "stringlength_multi: bad size");
            PDC_HIP(hipMalloc(&s.work, (size_t)(wb + 8)));
            PDC_TRY(pdc_stringlength_scan_dev(devices[d], s.stream, (double *)s.t, (double *)s.v, n,
                                              (double *)s.periods, cnt, (double *)s.out, s.work, wb));
        }
    }
    // Results come back only after EVERY slab has been launched: a copy into the caller's pageable
    // buffer blocks the host until that slab's scan has finished, and issued inside the loop above it
    // would run the devices one after another.
    for (int d = 0; d < n_devices; ++d) {
        if (!slots[d].stream) continue;
        const int64_t p0 = (int64_t)d * slab;
        const int64_t cnt = n_periods - p0 < slab ? n_periods - p0 : slab;
        PDC_TRY(use_device(devices[d]));
        PDC_HIP(hipMemcpyAsync(out + p0, slots[d].out, cnt * 8, hipMemcpyDeviceToHost, slots[d].stream));
    }
    for (int d = 0; d < n_devices; ++d) {
        if (!slots[d].stream) continue;
        PDC_TRY(use_device(devices[d]));
        PDC_HIP(hipStreamSynchronize(slots[d].stream));
    }
    return PDC_OK;
}

int phase_multi_entry(int kind, const char *who, const double *t, const double *v, int64_t n,
                      const double *periods, int64_t n_periods, int nb, int nc, double sigma,
                      double *out, const int *devices, int n_devices) {
    PDC_REQUIRE(t && v && devices && (periods || n_periods == 0) && (out || n_periods == 0),
                "%s: NULL argument", who);
    PDC_REQUIRE(n >= 0 && n_periods >= 0 && n_devices >= 1 && n_devices <= 64, "%s: bad size", who);
    if (n_periods == 0) return PDC_OK;
    std::vector<PhaseSlot> slots(n_devices);
    const int rc = phase_multi(kind, t, v, n, periods, n_periods, nb, nc, sigma, out, devices,
                               n_devices, slots);
    free_slots(slots, devices);
    return rc;
}

}  // namespace

extern "C" int pdc_pdm_scan_multi(const double *t, const double *x, int64_t n, const double *periods,
                                  int64_t n_periods, int nb, int nc, double sigma, double *theta_out,
                                  const int *devices, int n_devices) {
    return phase_multi_entry(0, "pdm_multi", t, x, n, periods, n_periods, nb, nc, sigma, theta_out,
                             devices, n_devices);
}

extern "C" int pdc_stringlength_scan_multi(const double *t, const double *m, int64_t n,
                                           const double *periods, int64_t n_periods, double *ell_out,
                                           const int *devices, int n_devices) {
    return phase_multi_entry(1, "stringlength_multi", t, m, n, periods, n_periods, 0, 0, 0.0, ell_out,
                             devices, n_devices);
}
