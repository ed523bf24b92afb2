#include "pdc_internal.h"
using namespace pdc;
extern "C" {
int64_t pdc_stringlength_work_bytes(int64_t, int64_t) { return 0; }
int pdc_stringlength_scan_dev(int, void *, const double *, const double *, int64_t, const double *,
                              int64_t, double *, void *, int64_t) {
    set_error("stringlength: not implemented yet");
    return PDC_ERR_INVALID;
}
int pdc_stringlength_scan(const double *, const double *, int64_t, const double *, int64_t, double *,
                          int) {
    set_error("stringlength: not implemented yet");
    return PDC_ERR_INVALID;
}
}
