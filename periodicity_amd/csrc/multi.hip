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

std::mutex g_comm_mutex;
std::vector<int> g_comm_devices;
std::vector<ncclComm_t> g_comms;

int get_comms(const int *devices, int n, ncclComm_t **out) {
    std::vector<int> want(devices, devices + n);
    if (want != g_comm_devices) {
        for (ncclComm_t c : g_comms) ncclCommDestroy(c);
        g_comms.assign(n, nullptr);
        g_comm_devices.clear();
        PDC_NCCL(ncclCommInitAll(g_comms.data(), n, devices));
        g_comm_devices = want;
    }
    *out = g_comms.data();
    return PDC_OK;
}

struct PerDevice {
    void *t = nullptr, *y = nullptr, *dy = nullptr, *pow = nullptr, *work = nullptr;
    hipStream_t stream = nullptr;
};

int free_all(std::vector<PerDevice> &pd, const int *devices) {
    for (size_t d = 0; d < pd.size(); ++d) {
        if (hipSetDevice(devices[d]) != hipSuccess) continue;
        if (pd[d].stream) (void)hipStreamDestroy(pd[d].stream);
        for (void *p : {pd[d].t, pd[d].y, pd[d].dy, pd[d].pow, pd[d].work})
            if (p) (void)hipFree(p);
    }
    return PDC_OK;
}

int scan_multi(const double *t, const double *y, const double *dy, int64_t n, double f0,
               double delta, int64_t nf, int fit_mean, int psd, double *power_out,
               const int *devices, int n_devices, std::vector<PerDevice> &pd) {
    const int64_t slab = (nf + n_devices - 1) / n_devices;  // equal counts; the tail is padding
    const int64_t wb = pdc_gls_work_bytes(n, 1, slab);
    for (int d = 0; d < n_devices; ++d) {
        PDC_TRY(use_device(devices[d]));
        PDC_HIP(hipStreamCreateWithFlags(&pd[d].stream, hipStreamNonBlocking));
        PDC_HIP(hipMalloc(&pd[d].t, (size_t)(n * 8 + 8)));
        PDC_HIP(hipMalloc(&pd[d].y, (size_t)(n * 8 + 8)));
        if (dy) PDC_HIP(hipMalloc(&pd[d].dy, (size_t)(n * 8 + 8)));
        PDC_HIP(hipMalloc(&pd[d].pow, (size_t)(slab * n_devices * 8 + 8)));
        PDC_HIP(hipMalloc(&pd[d].work, (size_t)wb));
        hipStream_t st = pd[d].stream;
        PDC_HIP(hipMemcpyAsync(pd[d].t, t, n * 8, hipMemcpyHostToDevice, st));
        PDC_HIP(hipMemcpyAsync(pd[d].y, y, n * 8, hipMemcpyHostToDevice, st));
        if (dy) PDC_HIP(hipMemcpyAsync(pd[d].dy, dy, n * 8, hipMemcpyHostToDevice, st));
        const int64_t j0 = (int64_t)d * slab;
        const int64_t cnt = j0 >= nf ? 0 : (nf - j0 < slab ? nf - j0 : slab);
        if (cnt < slab)  // keep the padding defined
            PDC_HIP(hipMemsetAsync((double *)pd[d].pow + j0 + cnt, 0, (size_t)((slab - cnt) * 8), st));
        if (cnt > 0)
            PDC_TRY(pdc_gls_scan_dev(devices[d], st, (double *)pd[d].t, (double *)pd[d].y,
                                     (double *)pd[d].dy, nullptr, n, 1, 0, f0, delta, j0, cnt,
                                     fit_mean, psd, (double *)pd[d].pow + j0, nullptr, nullptr,
                                     pd[d].work, wb));
    }
    // PDC_FORCE_RCCL=1 runs the collective even for one device (exercises the RCCL path on 1-GPU boxes)
    static const bool force_rccl = [] { const char *e = getenv("PDC_FORCE_RCCL"); return e && e[0] == '1'; }();
    if (n_devices > 1 || force_rccl) {
        ncclComm_t *comms;
        PDC_TRY(get_comms(devices, n_devices, &comms));
        PDC_NCCL(ncclGroupStart());
        for (int d = 0; d < n_devices; ++d) {
            double *buf = (double *)pd[d].pow;
            PDC_NCCL(ncclAllGather(buf + (int64_t)d * slab, buf, (size_t)slab, ncclDouble, comms[d],
                                   pd[d].stream));
        }
        PDC_NCCL(ncclGroupEnd());
    }
    PDC_TRY(use_device(devices[0]));
    PDC_HIP(hipMemcpyAsync(power_out, pd[0].pow, nf * 8, hipMemcpyDeviceToHost, pd[0].stream));
    for (int d = 0; d < n_devices; ++d) {
        PDC_TRY(use_device(devices[d]));
        PDC_HIP(hipStreamSynchronize(pd[d].stream));
    }
    return PDC_OK;
}

}  // namespace

extern "C" int pdc_gls_scan_multi(const double *t, const double *y, const double *dy, int64_t n,
                                  double f0, double delta, int64_t nf, int fit_mean, int psd,
                                  double *power_out, const int *devices, int n_devices) {
    PDC_REQUIRE(t && y && devices, "gls_multi: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0 && n_devices >= 1 && n_devices <= 64, "gls_multi: bad size");
    PDC_REQUIRE(power_out || nf == 0, "gls_multi: power_out is NULL");
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            PDC_REQUIRE(devices[a] != devices[b], "gls_multi: device %d listed twice", devices[a]);
    if (nf == 0) return PDC_OK;
    std::lock_guard<std::mutex> lk(g_comm_mutex);
    std::vector<PerDevice> pd(n_devices);
    const int rc = scan_multi(t, y, dy, n, f0, delta, nf, fit_mean, psd, power_out, devices,
                              n_devices, pd);
    free_all(pd, devices);
    return rc;
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
        PDC_HIP(hipMemcpyAsync(out + p0, s.out, cnt * 8, hipMemcpyDeviceToHost, s.stream));
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
