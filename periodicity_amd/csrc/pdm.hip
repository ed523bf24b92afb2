#include "pdc_internal.h"
using namespace pdc;
extern "C" {
int pdc_pdm_scan_dev(int, void *, const double *, const double *, int64_t, const double *, int64_t,
                     int, int, double, double *) {
    set_error("pdm: not implemented yet");
    return PDC_ERR_INVALID;
}
int pdc_pdm_scan(const double *, const double *, int64_t, const double *, int64_t, int, int, double,
                 double *, int) {
    set_error("pdm: not implemented yet");
    return PDC_ERR_INVALID;
}
}
