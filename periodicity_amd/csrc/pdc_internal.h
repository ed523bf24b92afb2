// Internal helpers shared by the translation units of libperiodicity_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/periodicity_hip.h"

namespace pdc {

// Per-thread message returned by pdc_last_error().
void set_error(const char *fmt, ...);

// Grow-only device scratch cached per (device, slot); released by pdc_release().
// Slots keep the buffers of one host-level call apart.
enum Slot { SLOT_IN0 = 0, SLOT_IN1, SLOT_IN2, SLOT_IN3, SLOT_OUT0, SLOT_OUT1, SLOT_OUT2, SLOT_WORK,
            SLOT_COUNT };
int cached(int device, Slot slot, int64_t bytes, void **dptr);

// Frees the multi-GPU plans the one-shot `_multi` entry points cache (multi.hip).
void release_multi();

// Every device / pinned-host allocation of the library goes through these two, so that
// pdc_alloc_counts() can show a caller (and the tests) that a cached path allocates nothing on
// its second call.
int device_alloc(void **dptr, int64_t bytes);
int pinned_alloc(void **hptr, int64_t bytes);

// Grow-only device scratch cached per (device, stream) for the `_dev` entry points that need a
// workspace the ABI does not pass in: work enqueued on one stream is ordered, so reuse is safe, and two
// streams never share a buffer.  (Stream-ordered pool memory - hipMallocAsync - is NOT used: on ROCm 7.2 a
// block handed out again by the pool gave kernels of the next call stale partial results unless the block
// was memset first)  Released by pdc_release(); the entry of a stream handed to
// pdc_stream_destroy() goes with it, and a device's entries are capped (kScratchPerDevice, true LRU) so that
// a caller cycling through raw HIP streams cannot grow the table without bound - but only entries that are
// not pinned and whose stream has run dry are ever evicted.  stream_scratch() returns the block PINNED; the
// caller unpins it (stream_scratch_done, or the ScratchPin guard) once all launches that use it are enqueued:
// from then on the stream itself is busy until they have run.  Entry points that take an
// explicit workspace (pdc_phase_scan_dev, pdc_gls_scan_dev, pdc_stringlength_scan_dev) never come here.
// RULE: no nested pin on one (device, stream) - a caller that takes stream_scratch() twice before stream_scratch_done()
// waits for itself.  Every `_dev` entry takes the block once, enqueues, unpins (ScratchPin).
int host_stream(int device, hipStream_t *st);   // the device's stream for host entry points (caller holds DeviceLock); destroyed by pdc_release()
int stream_scratch(int device, hipStream_t stream, int64_t bytes, void **dptr);
void stream_scratch_done(int device, hipStream_t stream);
int drop_stream_scratch(int device, hipStream_t stream);
struct ScratchPin {   // unpins on scope exit (also on the error returns of PDC_TRY / PDC_HIP)
    int device = -1;
    hipStream_t stream = nullptr;
    bool held = false;
    ~ScratchPin() {
        if (held) stream_scratch_done(device, stream);
    }
};

// The phase-fold statistics with an explicit workspace (pdm.hip): kind 0 = PDM theta, 1 = AoV,
// 2 = conditional entropy.  work == NULL falls back to stream_scratch().
int64_t phase_stat_work_bytes(int kind, int64_t n, int64_t n_periods, int nb, int nc);
int phase_stat_dev(int kind, int device, hipStream_t st, const double *d_t, const double *d_x, int64_t n,
                   const double *d_periods, int64_t n_periods, int nb, int nc, double sigma, double *d_out,
                   void *work, int64_t work_bytes);

// StringLength (kind 3) / Supersmoother (kind 5) for callers that hold the host arrays too (stringlength.hip): whether
// the streamed kernels will need their bin lists (hints bit 0) and whether t is in order (bit 1: no time sort to
// launch), the workspace for those hints, the scan.  hints = 1 is what a caller that has not looked passes.
int sorted_scan_hints(int kind, const double *t, int64_t n, const double *periods, int64_t n_periods);
int64_t sorted_scan_work_bytes(int kind, int64_t n, int64_t n_periods, int hints);
int sorted_scan_dev(int kind, int device, void *stream, const double *d_t, const double *d_v, int64_t n, const double *d_periods,
                    int64_t n_periods, double alpha, double *d_out, void *work, int64_t work_bytes, int hints);

// ---- workspace budget (round 6) -------------------------------------------------------------------------------
// The StringLength / Supersmoother workspaces are sized by built-in caps (12 GB of bin lists, 2 GB of sorted curves,
// 1 GB pools, 1024 workgroups' scratch ...) chosen for a 288 GB device that the caller owns.  On a shared device, or
// with eight loopback slots on one GPU, those caps are what made hipMalloc fail.  A budget - PDC_WORK_BUDGET_GB for
// every entry point, and for the HOST entry points also 0.9 x (free device memory + the cached workspace they would
// replace) from hipMemGetInfo - scales ALL of those caps by the largest power of two <= 1 for which the workspace
// fits: smaller batches, fewer resident workgroups, same results.  The scale is a thread-local set for the duration
// of one entry point (WorkScale; an inner entry keeps the outer one's), and every size function reads it, so
// `pdc_*_work_bytes` and the launch that follows agree as long as PDC_WORK_BUDGET_GB does not change in between.
int64_t work_budget();                          // PDC_WORK_BUDGET_GB in bytes (0: none)
int64_t host_work_budget(int device);           // min(work_budget(), 0.9 x (free + cached SLOT_WORK)) - caller holds the DeviceLock
double work_scale();                            // 1 outside a WorkScale scope
double *work_scale_slot(int **depth);           // (the thread-local pair behind WorkScale)
struct WorkScale {
    int64_t budget, need;
    template <typename Total>
    WorkScale(int64_t budget_, Total total) : budget(budget_), need(0) {
        int *depth;
        double *scale = work_scale_slot(&depth);
        if ((*depth)++ == 0) {
            *scale = 1.0;
            if (budget > 0)
                while (total() > budget && *scale > 1.0 / 65536.0) *scale *= 0.5;
        }
        need = total();
    }
    ~WorkScale() {
        int *depth;
        double *scale = work_scale_slot(&depth);
        if (--(*depth) == 0) *scale = 1.0;
    }
    bool fits() const { return budget <= 0 || need <= budget; }
};
#define PDC_REQUIRE_FITS(ws, what)                                                                                      \
    PDC_REQUIRE((ws).fits(), "%s: even the smallest batch needs %lld bytes of workspace, over the budget of %.3f GB "     \
                             "(PDC_WORK_BUDGET_GB / free device memory)", what, (long long)(ws).need, (double)(ws).budget / (double)(1 << 30))

// hipSetDevice + range check; every entry point starts here.
int use_device(int device);

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) on the CURRENT device,
// once per (device, kernel): the attribute belongs to the device's copy of the kernel, so a process that
// drives several GPUs has to set it on each of them.
int allow_dynamic_lds(const void *kernel, int bytes);

// GLS.bootstrap through the FFT path with the replicates given by index (glsfft.hip; host buffers, one device).
int gls_bootstrap_fft(const double *t, const double *y, const double *dy, int64_t n, const int32_t *picks,
                      int64_t n_boot, double fmin, double df, int64_t nf, int fit_mean, int psd,
                      double *amax_out, int64_t *argmax_out, int device);

// Serialises the host-level entry points that share the cached workspace of one device.
struct DeviceLock {
    explicit DeviceLock(int device);
    ~DeviceLock();
    int device;
};

}  // namespace pdc

#define PDC_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            pdc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                     \
            return _e == hipErrorOutOfMemory ? PDC_ERR_NOMEM : PDC_ERR_HIP;               \
        }                                                                                 \
    } while (0)

#define PDC_TRY(call)          \
    do {                       \
        int _s = (call);       \
        if (_s != PDC_OK) return _s; \
    } while (0)

#define PDC_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            pdc::set_error(__VA_ARGS__); \
            return PDC_ERR_INVALID;     \
        }                               \
    } while (0)

#include "pdc_device.h"
