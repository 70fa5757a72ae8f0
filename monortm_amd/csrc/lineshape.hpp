// lineshape.hpp - device-side line shapes (Humlicek w(z), speed-dependent Voigt, LSF_SDVOIGT), the prepared-line
// records staged in LDS, and the TIPS interpolation.  Reference: src/modm.f90:567-704, :965-1251; src/tips_2003.f90:4610.
#pragma once
#include "device_common.hpp"

namespace monortm_dev {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
struct cx {
    double re, im;
};
__device__ __forceinline__ cx cmk(double r, double i) { return cx{r, i}; }
__device__ __forceinline__ cx operator+(cx a, cx b) { return cmk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cx operator-(cx a, cx b) { return cmk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cx operator*(cx a, cx b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cx operator*(cx a, double s) { return cmk(a.re * s, a.im * s); }
__device__ __forceinline__ cx operator+(double s, cx a) { return cmk(s + a.re, a.im); }
__device__ __forceinline__ cx operator-(double s, cx a) { return cmk(s - a.re, -a.im); }
__device__ __forceinline__ cx operator/(cx a, cx b) {
    if (fabs(b.re) >= fabs(b.im)) {
        double r = b.im / b.re, d = b.re + b.im * r;
        return cmk((a.re + a.im * r) / d, (a.im - a.re * r) / d);
    }
    double r = b.re / b.im, d = b.re * r + b.im;
    return cmk((a.re * r + a.im) / d, (a.im * r - a.re) / d);
}
__device__ __forceinline__ cx cexpd(cx a) {
    double e = exp(a.re), s, c;
    sincos(a.im, &s, &c);
    return cmk(e * c, e * s);
}

// Humlicek (1982) rational approximations, coefficient strings of src/modm.f90:1107-1128
__device__ inline cx hum_r1(cx T) { return (T * .5641896) / (.5 + T * T); }
__device__ inline cx hum_r2(cx T) {
    cx U = T * T;
    return (T * (1.410474 + U * .5641896)) / (.75 + U * (3. + U));
}
__device__ inline cx hum_r3(cx T) {
    cx n = 16.4955 + T * (20.20933 + T * (11.96482 + T * (3.778987 + T * .5642236)));
    cx d = 16.4955 + T * (38.82363 + T * (39.27121 + T * (21.69274 + T * (6.699398 + T))));
    return n / d;
}
__device__ inline cx hum_r4(cx T) {
    cx U = T * T;
    cx n = T * (36183.31 - U * (3321.9905 - U * (1540.787 - U * (219.0313 - U * (35.76683 - U * (1.320522 - U * .56419))))));
    cx d = 32066.6 - U * (24322.84 - U * (9022.228 - U * (2186.181 - U * (364.2191 - U * (61.57037 - U * (1.841439 - U))))));
    return cexpd(U) - n / d;
}
// The region tests are evaluated as the reference evaluates them - product, then difference, each rounded - so that an
// argument exactly on a boundary picks the reference's region (a fused multiply-add differs in the last bit there).
__device__ inline cx w4(double x, double y) {  // src/modm.f90:1100-1130
#pragma clang fp contract(off)
    cx T = cmk(y, -x);
    double S = fabs(x) + y;
    if (S >= 15.) return hum_r1(T);
    if (S >= 5.5) return hum_r2(T);
    if (y >= 0.195 * fabs(x) - 0.176) return hum_r3(T);
    return hum_r4(T);
}
__device__ inline int hum_region_sd(double x, double y) {  // src/modm.f90:1161-1179 (II/III boundary at 6)
#pragma clang fp contract(off)
    double S = fabs(x) + y;
    if (S >= 15.0) return 1;
    if (S >= 6.0) return 2;
    return (y < 0.195 * fabs(x) - 0.176) ? 4 : 3;
}
__device__ inline cx sd_humlicek(double x1, double y1, double x2, double y2) {  // src/modm.f90:1150-1251
    cx T1 = cmk(y1, -x1), T2 = cmk(y2, -x2);
    int R1 = hum_region_sd(x1, y1), R2 = hum_region_sd(x2, y2);
    int R = R1 > R2 ? R1 : R2;
    if (R == 1) return hum_r1(T1) - hum_r1(T2);
    if (R == 2) return hum_r2(T1) - hum_r2(T2);
    if (R == 3) return hum_r3(T1) - hum_r3(T2);
    cx W1 = (R1 == 4) ? hum_r4(T1) : hum_r3(T1);
    cx W2 = (R2 == 4) ? hum_r4(T2) : hum_r3(T2);
    return W1 - W2;
}

// SDVOIGT, src/modm.f90:965-1087
__device__ inline double sdvoigt(double deltnu, double alphal, double alphad, double sdep, int *errflag) {
#pragma clang fp contract(off)
    const double TINY = 1.0e-4;
    double zeta = alphal / (alphal + alphad);
    double AL = 0., dnu = 0.;
    if (zeta < 1.00) {
        AL = alphal / alphad;
        dnu = deltnu / alphad;
    }
    if (zeta == 1.00 && fabs(sdep) < TINY) return alphal / (K_PI * (alphal * alphal + deltnu * deltnu));
    cx v;
    if (fabs(sdep) > TINY) {  // Boone et al. 2011 speed-dependent Voigt
        double gamma2 = alphal * sdep;
        double alfa = (alphal / gamma2) - 1.5;
        double beta = deltnu / gamma2;
        double delta = (1.0 / 4.0 / log(2.)) * (alphad * alphad / gamma2 / gamma2);
        double alfadelta = alfa + delta;
        double temp = sqrt(alfadelta * alfadelta + beta * beta);
        double x1 = (1.0 / sqrt(2.0)) * sqrt(temp + alfadelta) - sqrt(delta);
        double x2 = x1 + 2.0 * sqrt(delta);
        double sign = beta > 0.0 ? 1. : (beta == 0.0 ? 0. : -1.);
        double y1 = sign * sqrt((temp - delta - alfa) / 2.0);
        v = sd_humlicek(y1, x1, y1, x2);  // (y1,x1,y2,x2): the reference's argument order, modm.f90:1058
        if (v.re < 0.0) atomicOr(errflag, ERRBIT_SDV);  // reference: STOP (modm.f90:1062)
    } else {
        double x = sqrt(log(2.)) * dnu;
        double y = 1000.;
        if (zeta < 1.000) y = sqrt(log(2.)) * AL;
        v = w4(x, y);
    }
    double anorm1 = sqrt(log(2.) / K_PI) / alphad;
    return v.re * anorm1;
}

// SDVOIGT for an argument far from the centre - the negative resonance (WN + Xnu) and the 25 cm-1 pedestal: a plain Voigt
// (no speed dependence) whose argument falls into Humlicek's region I (|x| + y >= 15) is the closed form below, formed with
// the very operations of sdvoigt() -> w4() -> hum_r1(); everything else takes the call.  sdvoigt() is a function of some
// thousand instructions that the compiler keeps out of line: a call walks all of it for the few lanes that need it.
__device__ __forceinline__ double sdvoigt_far(double deltnu, double alphal, double alphad, double sdep, int *errflag) {
#pragma clang fp contract(off)
    double r = 0.;
    bool done = false;
    if (!(fabs(sdep) > 1.0e-4)) {
        const double zeta = alphal / (alphal + alphad);
        if (zeta < 1.00) {
            const double AL = alphal / alphad, dnu = deltnu / alphad;
            const double x = sqrt(log(2.)) * dnu, y = sqrt(log(2.)) * AL;
            if (fabs(x) + y >= 15.) {
                const cx v = hum_r1(cmk(y, -x));
                const double anorm1 = sqrt(log(2.) / K_PI) / alphad;
                r = v.re * anorm1;
                done = true;
            }
        }
    }
    if (!done) r = sdvoigt(deltnu, alphal, alphad, sdep, errflag);
    return r;
}

// ------------------------------------------------------------------------------------------------
// prepared line: what the per-wavenumber loop needs, staged in LDS
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) HotA {  // read by every evaluation
    double xnu;   // shifted line centre                                   modm.f90:375-380
    double hw2;   // HWHM_C^2
    double a2;    // S~ HWHM_C / pi
    double pa;    // generic: pedestal of the (+) resonance a2/(625+hw2)*Y1P; CO2: bare pedestal;
                  // O2: cut limit on |WN-Xnu| (25, or +inf for a coupled line)
};
// single-precision build (real_kind = 4): the reference keeps Xnu and WN REAL*8 there too (src/modm.f90:287, :139) and
// forms WN - Xnu in double before anything is rounded to REAL*4.  The record carries the centre as a float pair,
// xh = float(Xnu), xl = float(Xnu - xh) (48 bits of Xnu), the lane its wavenumber likewise, and
//     d = (wh - xh) + (wl - xl)
// in float arithmetic: the first difference is EXACT wherever it matters (Sterbenz: the operands are within a factor of two of
// each other next to a line centre), so d carries the rounding of one float operation like float(WN - Xnu) does - without a
// double subtraction and a conversion per evaluation (3 packed float operations per two wavenumbers instead of 2 + 2).  The
// amplitudes are float and carry the column amount W so that they stay inside the float range.
struct __attribute__((aligned(8))) HotAf {
    float xh, xl;
    float hw2, a2, pa, pb;
};
template <typename R> struct HotOf { using type = HotA; };
template <> struct HotOf<float> { using type = HotAf; };

struct __attribute__((aligned(16))) HotB {  // read only by the variants that need it
    double pb;    // generic: pedestal of the (-) resonance (x Y2P); O2: limit on WN+Xnu for the (-) resonance
    double d100;  // 100 * HWHM_D, or -1 when zeta > 0.99 or no wavenumber of the tile is that close (modm.f90:427)
    double c1;    // AIP * (1/HWHM_C) * RP   (0 when the shape carries no Y factor)
    double gp1;   // 1 + BIP * RP2           (1 when ...)
};
struct __attribute__((aligned(8))) ColdLine {  // Voigt candidates only (32 bytes: the one-wave kernel's LDS budget is 10 KB)
    double stild, hw, hwd;
    float sdep;
    uint32_t info;  // bits 0-5 molecule, 6-7 coupling code
};
// XL3 = SDVOIGT(25, HWHM, AD, SDEP), the pedestal of a Voigt pair (modm.f90:596, :639, :651): it depends on the line and the layer only,
// but it is formed where the pairs are worked off (one pair per lane, dense) and not in the prepare stage - there ONE candidate among
// the 64 lines of a pass made the whole wave walk sdvoigt_far (and, for a line with speed dependence, all of sdvoigt): 3.7 % of
// configs[3] (round 6, ablation).  The same function with the same arguments as the reference's call: identical bits.
// (O2 shapes have no pedestal: lsf_sdvoigt never reads it there)
__device__ __forceinline__ double cold_xl3(const ColdLine &c, int mol, int *errflag) {
    return (mol != 7) ? sdvoigt_far(25., c.hw, c.hwd, (double)c.sdep, errflag) : 0.;
}

// x**y for x > 0 (the reference's REAL ** REAL): exp(y log x) keeps the register footprint small, the result is
// within a few ulp of libm pow
__device__ __forceinline__ double powpos(double x, double y) { return exp(y * log(x)); }

__device__ __forceinline__ double xlq(double z) { return 1.0 / (1.0 + z * z); }  // pi * XLORENTZ(z)

// Full LSF_SDVOIGT for one (wavenumber, line): src/modm.f90:567-704.  mol 7 = O2, 2 = CO2.
// XL3 = SDVOIGT(deltnuC, HWHM, AD, SDEP) is handed in: the reference evaluates it inside every call, but it depends on the
// line and the layer only, so the prepare stage forms it once per (layer, line) with the same function
__device__ inline double lsf_sdvoigt(int mol, int code, double RP, double RP2, double AIP, double BIP, double HWHM, double WN,
                              double Xnu, double AD, double SDEP, const double XL3in, int *errflag) {
    const double deltnuC = 25.;
    const double DIFF = (WN + Xnu) - deltnuC;
    double SLS = 0.;
    const bool lc = code != 0;
    if (mol != 7 && mol != 2) {
        double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
        double XL3 = XL3in;
        if (lc) {
            double Y1 = (1. + (AIP * (1 / HWHM) * RP * (WN - Xnu)) + (BIP * RP2));
            double Y1P = (1. + (AIP * (1 / HWHM) * RP * (deltnuC)) + (BIP * RP2));
            if (DIFF <= 0.) {
                double XL2 = sdvoigt_far(WN + Xnu, HWHM, AD, SDEP, errflag);
                double Y2 = (1. - (AIP * (1 / HWHM) * RP * (WN + Xnu)) + (BIP * RP2));
                double Y2P = (1. - (AIP * (1 / HWHM) * RP * (deltnuC)) + (BIP * RP2));
                SLS = (Y1 * (XL1)-Y1P * (XL3) + Y2 * (XL2)-Y2P * (XL3));
            } else
                SLS = Y1 * (XL1)-Y1P * (XL3);
        } else {
            if (DIFF <= 0.) {
                double XL2 = sdvoigt_far(WN + Xnu, HWHM, AD, SDEP, errflag);
                SLS = (XL1 + XL2 - (2 * XL3));
            } else
                SLS = (XL1 - XL3);
        }
    } else if (fabs(WN - Xnu) <= deltnuC && !lc) {
        double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
        if (mol == 7) {
            if (DIFF <= 0.) SLS = XL1 + sdvoigt_far(WN + Xnu, HWHM, AD, SDEP, errflag);
            else SLS = XL1;
        } else {
            double dx = WN - Xnu;
            double XL3 = XL3in;
            XL3 = XL3 * (2. - ((dx * dx) / (deltnuC * deltnuC)));
            SLS = XL1 - XL3;  // chi == 1 (modm.f90:1286)
        }
    } else if (mol == 7) {
        if (lc) {
            double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
            double XL2 = sdvoigt_far(WN + Xnu, HWHM, AD, SDEP, errflag);
            if (code == 1) {
                double Y1 = (1. + (AIP * (1 / HWHM) * RP * (WN - Xnu)) + (BIP * RP2));
                double Y2 = (1. - (AIP * (1 / HWHM) * RP * (WN + Xnu)) + (BIP * RP2));
                SLS = (XL1 * (Y1) + XL2 * (Y2));
            } else
                SLS = XL1 + XL2;
        }
    } else {
        // CO2 with coupling.  Literal reference condition (XF.EQ.-1).or.(XF.EQ.-3).or.(XF.NE.-5)
        // (modm.f90:659): an XF = -5 line gets SLS = 0 on the Voigt side.
        if (code != 3) {
            double dx = WN - Xnu;
            double XL1 = sdvoigt(dx, HWHM, AD, SDEP, errflag);
            double XL3 = XL3in;
            double f = (2. - (dx * dx) / (deltnuC * deltnuC));
            if (code == 1) {  // XF == -1 (-5 cannot reach here)
                double Y1 = (1. + (AIP * (1 / HWHM) * RP * (dx)) + (BIP * RP2));
                SLS = (XL1 * (Y1)-XL3 * f - XL3 * ((Y1 - 1.) * f));
            } else if (lc)
                SLS = XL1 - XL3 * f;
        }
    }
    return SLS;
}

// RADFN: the radiation term (src/lblrtm_sub.f90:36-97)
__device__ __forceinline__ double radfn(double VI, double XKT) {
    if (XKT > 0.0) {
        double x = VI / XKT;
        if (x <= 0.01) return 0.5 * x * VI;
        if (x <= 10.0) {
            double e = exp(-x);
            return VI * (1. - e) / (1. + e);
        }
    }
    return VI;
}

// 3- / 4-point Lagrange of TIPS (AtoB, src/tips_2003.f90:4610-4700).  The temperature grid is uniform (60 K + 25 K
// steps, tips_2003.f90:312-336), so the node index follows from aa directly and the Lagrange denominators are the
// constants (+-25)(+-50)(+-75): no search, no divisions.  Host and device share this function (Q(296) is tabulated
// once on the host).
// Split in two so that a caller can issue the four table reads early (lines_kernel's prologue): tips_nodes() = which nodes,
// tips_interp() = the Lagrange form on their values; tips_atob() is the two back to back.
struct TipsNodes {
    int J;      // 1-based node index: values B[J-3 .. J] (three-point ends: B[J-3 .. J-1])
    int ends;   // 1: the three-point form of the table's first / last interval
    int none;   // 1: aa lies beyond the table (the reference's QT <= 0 -> STOP)
};
__host__ __device__ inline TipsNodes tips_nodes(double aa) {
    const int npt = 119;
    int I = (int)ceil((aa - 60.) / 25.) + 1;   // first node with A(I) >= aa
    if (I < 2) I = 2;
    TipsNodes n = {I, 0, 0};
    if (I > npt) { n.J = npt; n.none = 1; return n; }
    if (I < 3 || I == npt) { n.J = (I < 3) ? 3 : npt; n.ends = 1; }
    return n;
}
__host__ __device__ inline double tips_interp(double aa, TipsNodes n, double b0, double b1, double b2, double b3) {
    if (n.none) return 0.;
    const int J = n.J;
    if (n.ends) {
        const double a0 = 60. + 25. * (J - 3), a1 = a0 + 25., a2 = a0 + 50.;
        const double A0 = (aa - a1) * (aa - a2) * (1. / 1250.);    // (a0-a1)(a0-a2) = (-25)(-50)
        const double A1 = (aa - a0) * (aa - a2) * (-1. / 625.);    // (a1-a0)(a1-a2) = (25)(-25)
        const double A2 = (aa - a0) * (aa - a1) * (1. / 1250.);    // (a2-a0)(a2-a1) = (50)(25)
        return A0 * b0 + A1 * b1 + A2 * b2;
    }
    const double a0 = 60. + 25. * (J - 3), a1 = a0 + 25., a2 = a0 + 50., a3 = a0 + 75.;
    const double A0 = (aa - a1) * (aa - a2) * (aa - a3) * (-1. / 93750.);   // (-25)(-50)(-75)
    const double A1 = (aa - a0) * (aa - a2) * (aa - a3) * (1. / 31250.);    // (25)(-25)(-50)
    const double A2 = (aa - a0) * (aa - a1) * (aa - a3) * (-1. / 31250.);   // (50)(25)(-25)
    const double A3 = (aa - a0) * (aa - a1) * (aa - a2) * (1. / 93750.);    // (75)(50)(25)
    return A0 * b0 + A1 * b1 + A2 * b2 + A3 * b3;
}
__host__ __device__ inline double tips_atob(double aa, const double *B) {
    const TipsNodes n = tips_nodes(aa);
    if (n.none) return 0.;
    const int J = n.J;
    return tips_interp(aa, n, B[J - 3], B[J - 2], B[J - 1], n.ends ? 0. : B[J]);
}

// scor(mol, iso) = Q(296)/Q(T) as TIPS_2003 leaves it (src/tips_2003.f90:60-292); the caller has checked 70 <= T <= 3000.
// q296: Q(296 K) of every isotopologue, tabulated once per context with the same interpolation.  *bad is set when the
// reference would STOP on a partition sum <= 0 (:272-277).
//   molecule 34 (O): Q = 1 at both temperatures.
//   molecule 39 (CH3OH): :260-266 set 296 and (T/296)**1.5 and jump to label 100, where :287-288 overwrite both with QT -
//   still the value molecule 38 left behind (the loop visits 38 first) - so the reference returns exactly 1.
//   isotopologues beyond min(9, ISONM(mol)) are never written by the reference (uninitialised there): 0.
__device__ inline double tips_scor(const int *isonm, const int *offset, const double *qoft, const double *q296, int mol, int iso,
                                   double Tk, bool *bad) {
    if (iso > min(9, isonm[mol - 1])) return 0.;
    if (mol == 34 || mol == 39) return 1.;
    const int slot = offset[mol - 1] + iso - 1;
    const double qt = tips_atob(Tk, qoft + (size_t)slot * 119);
    if (qt <= 0.) *bad = true;
    return q296[slot] / qt;
}

}  // namespace monortm_dev
