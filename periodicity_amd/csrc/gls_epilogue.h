// Epilogue of GLS.__call__ (/root/reference/src/periodicity/spectral.py:113-132), written in the
// reference's own operation order (build with -ffp-contract=off), shared by the direct-sum scan
// (gls.hip) and the FFT-extirpolation path (glsfft.hip).
#pragma once

namespace pdc {

// Sh, Ch: sum (w y) sin/cos(omega t); S, C: sum w sin/cos(omega t) (fit_mean only);
// S2, C2: sum w sin/cos(2 omega t).  YY = sum w y^2, Werr = sum err^-2.
template <bool FIT_MEAN>
__device__ __forceinline__ double gls_power_from_sums(double Sh, double Ch, double S, double C,
                                                      double S2, double C2, double YY, double Werr,
                                                      int psd) {
    double tan2;
    if (FIT_MEAN) {
        tan2 = (S2 - 2.0 * S * C) / (C2 - (C * C - S * S));
    } else {
        tan2 = S2 / C2;
    }
    const double nrm = __builtin_sqrt(1.0 + tan2 * tan2);
    const double S2w = tan2 / nrm;
    const double C2w = 1.0 / nrm;
    const double rh = __builtin_sqrt(0.5);
    const double Cw = rh * __builtin_sqrt(1.0 + C2w);
    const double sgn = (S2w != S2w) ? S2w : (double)((S2w > 0.0) - (S2w < 0.0));  // np.sign
    const double Sw = rh * sgn * __builtin_sqrt(1.0 - C2w);
    const double YC = Ch * Cw + Sh * Sw;
    const double YS = Sh * Cw - Ch * Sw;
    double CC = 0.5 * (1.0 + C2 * C2w + S2 * S2w);
    double SSw = 0.5 * (1.0 - C2 * C2w - S2 * S2w);
    if (FIT_MEAN) {
        const double a = C * Cw + S * Sw;
        const double b = S * Cw - C * Sw;
        CC -= a * a;
        SSw -= b * b;
    }
    double power = YC * YC / CC + YS * YS / SSw;
    if (psd) {
        power *= 0.5 * Werr;
    } else {
        power /= YY;
    }
    return power;
}

}  // namespace pdc
