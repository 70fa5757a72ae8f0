// monortm_hip.hip - MI355X (gfx950 / CDNA4) implementation of monoRTM's optical-depth and
// radiative-transfer hot path behind the C ABI of include/monortm_hip.h.
//
// Reference behaviour being replaced (paths relative to /root/reference):
//   MODM      src/modm.f90:21-274      LINES  src/modm.f90:277-440
//   line shapes src/modm.f90:567-831, :888-895, :965-1251
//   CONTNM    src/contnm.f90:25-1142 (+ accessors)   XINT/RADFN src/lblrtm_sub.f90
//   TIPS_2003 src/tips_2003.f90:2-298, :4610         ODCLW_TKC src/CloudOptProp.f90:29-157
//   CALCTMR / RTM / RAD_UP_DN  src/RTMmono.f90
//
// Design (DESIGN.md has the full account):
//   * one process = one GPU; a context owns the device line table (44 B per line, SoA);
//   * lines_kernel: workgroup = (profile, layer, tile of NW*64 wavenumbers), lane = wavenumber.
//     Everything of a line that does not depend on the wavenumber (shifted centre, S~, Lorentz and
//     Doppler widths, coupling factors, pedestal) is prepared ONCE per (layer, line) by one lane,
//     staged in LDS, and then broadcast-read by every wave: the inner loop is one FP64 reciprocal
//     and ~15 FP64 FMAs per (wavenumber, layer, line).  The reference recomputes all of it per
//     wavenumber (6 exp, 2 pow, 3 sqrt per evaluation).
//   * finish_kernel: workgroup = (profile, layer); MT_CKD continuum on the 1 cm-1 ABSRB grid in LDS,
//     second interpolation to the wavenumbers, TKC cloud liquid, totals.
//   * rtm_kernel: lane = (profile, wavenumber); CALCTMR + RAD_UP_DN + RTM recurrences in registers.
// No MFMA (nothing here is a dense contraction), no Triton, no CUDA compatibility layer.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/monortm_hip.h"
#include "line_table.hpp"
#include "tables/monortm_tables.h"

namespace {

// ------------------------------------------------------------------------------------------------
// constants: the literal decimal strings of the reference as doubles (src/PhysConstants.f90:19-39)
// ------------------------------------------------------------------------------------------------
#define K_PI 3.1415926535898
#define K_PLANCK 6.62606876E-27
#define K_BOLTZ 1.3806503E-16
#define K_CLIGHT 2.99792458E+10
#define K_AVOGAD 6.02214199E+23
#define K_RADCN1 1.191042722E-12
#define K_RADCN2 1.4387752
#define K_ONEPL 1.001
#define K_ONEMI 0.999
#define K_T0 296.0
#define K_P0 1013.25

constexpr int MXMOL = 39;
constexpr int MXBRD = 7;
constexpr int NSCOR = MXMOL * 9;

enum : int { ERRBIT_TEMP = 1, ERRBIT_SDV = 2 };

struct DevTables {  // device copies of monortm_tables.h
    const double *self296, *self260, *frgn296, *fco2, *n2c296, *n2sf296, *n2c220, *n2sf220, *xfac_rhu, *xfacco2,
        *tdep_bandhead, *tips_qoft, *tips_q296, *smass;
    // branches above 1340 cm-1
    const double *o3ch_x, *o3ch_y, *o3ch_z, *o3hh0, *o3hh1, *o3hh2, *o3huv, *o2f_x, *o2f_t, *o2inf1, *o2inf3, *o2vis, *o2fuv,
        *n2f_272, *n2f_228, *n2f_ah2o, *n2f1;
    const int *tips_isonm, *tips_offset;
};

struct DevLines {
    const double *vnu, *s0adj, *lc;
    const float *alfa, *hwhm, *epp, *tmpalf, *pshift, *sdep, *brd_dat;
    const uint32_t *meta;
    const int32_t *brd_flg;
    int mol_start[MXMOL + 2];
    unsigned long long sorted_mask;
    unsigned long long lc_mask;  // molecules that own at least one line-coupled entry
    double max_abs_shift;
};

struct ModmArgs {
    int nprof, nwn, nlay_max, nmol, ibrd;
    double dvset, sclcpl, sclhw, y0res;
    double cntnm[7];
    const double *wn, *P, *T, *CLW, *WKL, *WBRODL;
    const int *nlay;
    double *O, *O_BY_MOL, *OC, *O_CLW;
    int *errflag;
    // line slicing (few workgroups otherwise): nslice blocks share one (profile, layer, tile); each writes its
    // partial sums to partial[slice][profile][layer][mol][wn], finish_kernel adds them in slice order
    int nslice;
    double *partial;
};

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
__device__ cx hum_r1(cx T) { return (T * .5641896) / (.5 + T * T); }
__device__ cx hum_r2(cx T) {
    cx U = T * T;
    return (T * (1.410474 + U * .5641896)) / (.75 + U * (3. + U));
}
__device__ cx hum_r3(cx T) {
    cx n = 16.4955 + T * (20.20933 + T * (11.96482 + T * (3.778987 + T * .5642236)));
    cx d = 16.4955 + T * (38.82363 + T * (39.27121 + T * (21.69274 + T * (6.699398 + T))));
    return n / d;
}
__device__ cx hum_r4(cx T) {
    cx U = T * T;
    cx n = T * (36183.31 - U * (3321.9905 - U * (1540.787 - U * (219.0313 - U * (35.76683 - U * (1.320522 - U * .56419))))));
    cx d = 32066.6 - U * (24322.84 - U * (9022.228 - U * (2186.181 - U * (364.2191 - U * (61.57037 - U * (1.841439 - U))))));
    return cexpd(U) - n / d;
}
__device__ cx w4(double x, double y) {  // src/modm.f90:1100-1130
    cx T = cmk(y, -x);
    double S = fabs(x) + y;
    if (S >= 15.) return hum_r1(T);
    if (S >= 5.5) return hum_r2(T);
    if (y >= 0.195 * fabs(x) - 0.176) return hum_r3(T);
    return hum_r4(T);
}
__device__ int hum_region_sd(double x, double y) {  // src/modm.f90:1161-1179 (II/III boundary at 6)
    double S = fabs(x) + y;
    if (S >= 15.0) return 1;
    if (S >= 6.0) return 2;
    return (y < 0.195 * fabs(x) - 0.176) ? 4 : 3;
}
__device__ cx sd_humlicek(double x1, double y1, double x2, double y2) {  // src/modm.f90:1150-1251
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
__device__ double sdvoigt(double deltnu, double alphal, double alphad, double sdep, int *errflag) {
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
struct __attribute__((aligned(16))) HotB {  // read only by the variants that need it
    double pb;    // generic: pedestal of the (-) resonance (x Y2P); O2: limit on WN+Xnu for the (-) resonance
    double d100;  // 100 * HWHM_D, or -1 when zeta > 0.99 or no wavenumber of the tile is that close (modm.f90:427)
    double c1;    // AIP * (1/HWHM_C) * RP   (0 when the shape carries no Y factor)
    double gp1;   // 1 + BIP * RP2           (1 when ...)
};
struct __attribute__((aligned(16))) ColdLine {
    double stild, hw, hwd;
    float sdep;
    uint32_t info;  // bits 0-5 molecule, 6-7 coupling code
};

// x**y for x > 0 (the reference's REAL ** REAL): exp(y log x) keeps the register footprint small, the result is
// within a few ulp of libm pow
__device__ __forceinline__ double powpos(double x, double y) { return exp(y * log(x)); }

__device__ __forceinline__ double xlq(double z) { return 1.0 / (1.0 + z * z); }  // pi * XLORENTZ(z)

// Full LSF_SDVOIGT for one (wavenumber, line): src/modm.f90:567-704.  mol 7 = O2, 2 = CO2.
__device__ double lsf_sdvoigt(int mol, int code, double RP, double RP2, double AIP, double BIP, double HWHM, double WN,
                              double Xnu, double AD, double SDEP, int *errflag) {
    const double deltnuC = 25.;
    const double DIFF = (WN + Xnu) - deltnuC;
    double SLS = 0.;
    const bool lc = code != 0;
    if (mol != 7 && mol != 2) {
        double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
        double XL3 = sdvoigt(deltnuC, HWHM, AD, SDEP, errflag);
        if (lc) {
            double Y1 = (1. + (AIP * (1 / HWHM) * RP * (WN - Xnu)) + (BIP * RP2));
            double Y1P = (1. + (AIP * (1 / HWHM) * RP * (deltnuC)) + (BIP * RP2));
            if (DIFF <= 0.) {
                double XL2 = sdvoigt(WN + Xnu, HWHM, AD, SDEP, errflag);
                double Y2 = (1. - (AIP * (1 / HWHM) * RP * (WN + Xnu)) + (BIP * RP2));
                double Y2P = (1. - (AIP * (1 / HWHM) * RP * (deltnuC)) + (BIP * RP2));
                SLS = (Y1 * (XL1)-Y1P * (XL3) + Y2 * (XL2)-Y2P * (XL3));
            } else
                SLS = Y1 * (XL1)-Y1P * (XL3);
        } else {
            if (DIFF <= 0.) {
                double XL2 = sdvoigt(WN + Xnu, HWHM, AD, SDEP, errflag);
                SLS = (XL1 + XL2 - (2 * XL3));
            } else
                SLS = (XL1 - XL3);
        }
    } else if (fabs(WN - Xnu) <= deltnuC && !lc) {
        double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
        if (mol == 7) {
            if (DIFF <= 0.) SLS = XL1 + sdvoigt(WN + Xnu, HWHM, AD, SDEP, errflag);
            else SLS = XL1;
        } else {
            double dx = WN - Xnu;
            double XL3 = sdvoigt(deltnuC, HWHM, AD, SDEP, errflag);
            XL3 = XL3 * (2. - ((dx * dx) / (deltnuC * deltnuC)));
            SLS = XL1 - XL3;  // chi == 1 (modm.f90:1286)
        }
    } else if (mol == 7) {
        if (lc) {
            double XL1 = sdvoigt(WN - Xnu, HWHM, AD, SDEP, errflag);
            double XL2 = sdvoigt(WN + Xnu, HWHM, AD, SDEP, errflag);
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
            double XL3 = sdvoigt(deltnuC, HWHM, AD, SDEP, errflag);
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

// 3- / 4-point Lagrange of TIPS (AtoB, src/tips_2003.f90:4610-4700).  The temperature grid is uniform (60 K + 25 K
// steps, tips_2003.f90:312-336), so the node index follows from aa directly and the Lagrange denominators are the
// constants (+-25)(+-50)(+-75): no search, no divisions.  Host and device share this function (Q(296) is tabulated
// once on the host).
__host__ __device__ inline double tips_atob(double aa, const double *B) {
    const int npt = 119;
    int I = (int)ceil((aa - 60.) / 25.) + 1;   // first node with A(I) >= aa
    if (I < 2) I = 2;
    if (I > npt) return 0.;
    if (I < 3 || I == npt) {
        const int J = (I < 3) ? 3 : npt;
        const double a0 = 60. + 25. * (J - 3), a1 = a0 + 25., a2 = a0 + 50.;
        const double A0 = (aa - a1) * (aa - a2) * (1. / 1250.);    // (a0-a1)(a0-a2) = (-25)(-50)
        const double A1 = (aa - a0) * (aa - a2) * (-1. / 625.);    // (a1-a0)(a1-a2) = (25)(-25)
        const double A2 = (aa - a0) * (aa - a1) * (1. / 1250.);    // (a2-a0)(a2-a1) = (50)(25)
        return A0 * B[J - 3] + A1 * B[J - 2] + A2 * B[J - 1];
    }
    const int J = I;
    const double a0 = 60. + 25. * (J - 3), a1 = a0 + 25., a2 = a0 + 50., a3 = a0 + 75.;
    const double A0 = (aa - a1) * (aa - a2) * (aa - a3) * (-1. / 93750.);   // (-25)(-50)(-75)
    const double A1 = (aa - a0) * (aa - a2) * (aa - a3) * (1. / 31250.);    // (25)(-25)(-50)
    const double A2 = (aa - a0) * (aa - a1) * (aa - a3) * (-1. / 31250.);   // (50)(25)(-25)
    const double A3 = (aa - a0) * (aa - a1) * (aa - a2) * (1. / 93750.);    // (75)(50)(25)
    return A0 * B[J - 3] + A1 * B[J - 2] + A2 * B[J - 1] + A3 * B[J];
}

// FP64 reciprocal: v_rcp_f64 seed (relative error 4.6e-8 measured on gfx950, tools/rcp_accuracy.hip) + one
// Newton step -> 2.2e-15.  Operands are positive normal numbers (d^2 + HWHM^2 and products of two of them),
// so no scaling / special cases are needed; an IEEE-correct division costs ~3x as many issue slots.
__device__ __forceinline__ double frcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
// two Newton steps = exact to 1 ulp (prepare stage: widths, S~ denominators)
__device__ __forceinline__ double frcp_any(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// The Lorentz shapes of src/modm.f90:706-831, regrouped.  With a2 = S~ HWHM/pi and hw2 = HWHM^2:
//     S~ * XLORENTZ(d/HWHM)/HWHM = a2 / (d^2 + hw2)
// so one evaluation is (d, d^2+hw2, one reciprocal, one FMA for the pedestal); two resonances share a
// single reciprocal:  a2*(Y1*den2 + Y2*den1)/(den1*den2).
//   KIND : 0 generic molecule, 1 O2 (no pedestal; coupled lines exempt from the 25 cm-1 rule), 2 CO2
//          (pedestal x (2 - d^2/625), no negative resonance)

// ---- fast path: molecule run without coupled lines and without any Voigt candidate in this chunk ----------
//   M2 : some line of the run can have its negative resonance within 25 cm-1 of zero for a wavenumber of the tile
template <int KIND, bool M2>
__device__ __forceinline__ double eval_one_fast(const HotA h, const double pb_or_lim, double WN) {
    const double d = WN - h.xnu;
    const double den1 = fma(d, d, h.hw2);
    const double cutlim = (KIND == 1) ? h.pa : 25.;
    const bool live = !(fabs(d) > cutlim);  // modm.f90:384 (O2: inside the shape function, :755)
    double term;
    if (KIND == 2) {
        const double f = fma(-(d * d), 1.0 / 625., 2.);
        term = fma(-h.pa, f, h.a2 * frcp(den1));
    } else if (!M2) {
        term = (KIND == 0) ? fma(h.a2, frcp(den1), -h.pa) : h.a2 * frcp(den1);
    } else {
        // 1/den1 + [m2]/den2 = (den2 + [m2] den1) / (den1 den2): the condition enters as a 0/1 factor, no selects
        const double dp = WN + h.xnu;
        const double m2f = (dp <= ((KIND == 1) ? pb_or_lim : 25.)) ? 1.0 : 0.0;  // DIFF = (WN+Xnu) - 25 <= 0 (modm.f90:713)
        const double den2 = fma(dp, dp, h.hw2);
        const double num = fma(m2f, den1, den2);
        const double t = h.a2 * num;
        if (KIND == 0) term = fma(t, frcp(den1 * den2), -fma(m2f, pb_or_lim, h.pa));
        else term = t * frcp(den1 * den2);
    }
    return live ? term : 0.;
}

template <int KIND, bool M2>
__device__ __forceinline__ double eval_fast(const HotA *sA, const HotB *sB, int j0, int j1, double WN, double SF) {
    // two lines per trip, LDS records fetched one line ahead (ping-pong registers, no copies)
    constexpr bool needB = M2 && KIND != 2;
    HotA h0 = sA[j0];
    double b0 = needB ? sB[j0].pb : 0.;
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const HotA h1 = sA[j + 1];
        const double b1 = needB ? sB[j + 1].pb : 0.;
        SF += eval_one_fast<KIND, M2>(h0, b0, WN);
        const int jn = (j + 2 < j1) ? j + 2 : j + 1;
        h0 = sA[jn];
        if (needB) b0 = sB[jn].pb;
        SF += eval_one_fast<KIND, M2>(h1, b1, WN);
    }
    if (j < j1) SF += eval_one_fast<KIND, M2>(h0, b0, WN);
    return SF;
}

// ---- general path: coupled lines (Y factors) and / or Voigt candidates ------------------------------------
template <int KIND, bool VOIGT>
__device__ __forceinline__ double eval_general(const HotA *sA, const HotB *sB, const ColdLine *sCold, int j0, int j1, double WN,
                                               int mol, double SF, int *errflag) {
    HotA h = sA[j0];
    HotB b = sB[j0];
    for (int j = j0; j < j1; j++) {
        const int jn = (j + 1 < j1) ? j + 1 : j;
        const HotA hnext = sA[jn];  // software prefetch of the next line's LDS records
        const HotB bnext = sB[jn];
        const double d = WN - h.xnu, dp = WN + h.xnu;
        const double ad = fabs(d);
        const double den1 = fma(d, d, h.hw2);
        const double Y1 = fma(b.c1, d, b.gp1);
        double term;
        bool live;
        if (KIND == 2) {
            live = !(ad > 25.);
            const double f = fma(-(d * d), 1.0 / 625., 2.);
            term = Y1 * fma(-h.pa, f, h.a2 * frcp(den1));
        } else {
            const double cutlim = (KIND == 1) ? h.pa : 25.;
            const double dplim = (KIND == 1) ? b.pb : 25.;
            live = !(ad > cutlim);
            const bool m2 = dp <= dplim;
            if (__builtin_amdgcn_ballot_w64(m2 && live) == 0ull) {
                term = (KIND == 0) ? fma(h.a2 * Y1, frcp(den1), -h.pa) : (h.a2 * Y1) * frcp(den1);
            } else {
                const double den2 = m2 ? fma(dp, dp, h.hw2) : 1.0;
                const double Y2 = m2 ? fma(-b.c1, dp, b.gp1) : 0.0;
                term = (h.a2 * fma(Y1, den2, Y2 * den1)) * frcp(den1 * den2);
                if (KIND == 0) term -= (m2 ? h.pa + b.pb : h.pa);
            }
        }
        if (VOIGT) {
            const bool useV = live && !(ad > b.d100);  // modm.f90:427
            if (__builtin_amdgcn_ballot_w64(useV) != 0ull) {
                if (useV) {
                    const ColdLine c = sCold[j];
                    // the shape functions only use the products AIP*(1/HW)*RP = c1 and BIP*RP2 = gp1-1:
                    // hand them over as AIP' = c1*HW, BIP' = gp1-1 with RP' = RP2' = 1
                    const double SLS = lsf_sdvoigt(mol, (int)((c.info >> 6) & 3), 1.0, 1.0, b.c1 * c.hw, b.gp1 - 1., c.hw, WN, h.xnu,
                                                   c.hwd, (double)c.sdep, errflag);
                    term = c.stild * SLS;
                }
            }
        }
        SF += live ? term : 0.;
        h = hnext;
        b = bnext;
    }
    return SF;
}

template <int KIND>
__device__ __forceinline__ double eval_dispatch(bool lc, bool voigt, bool m2, const HotA *sA, const HotB *sB,
                                                const ColdLine *sCold, int j0, int j1, double WN, int mol, double SF,
                                                int *errflag) {
    if (voigt) return eval_general<KIND, true>(sA, sB, sCold, j0, j1, WN, mol, SF, errflag);
    if (lc) return eval_general<KIND, false>(sA, sB, sCold, j0, j1, WN, mol, SF, errflag);
    if (KIND != 2 && m2) return eval_fast<KIND, true>(sA, sB, j0, j1, WN, SF);
    return eval_fast<KIND, false>(sA, sB, j0, j1, WN, SF);
}

// ------------------------------------------------------------------------------------------------
// lines_kernel: O_BY_MOL(wn, mol, layer) = RFT * W_mol * sum_lines S~ * shape      (modm.f90:253-262)
// grid = (wavenumber tiles, layers, profiles); block = NW waves; lane = wavenumber
// ------------------------------------------------------------------------------------------------
// IBRD: species-by-species broadening data are read (IBRD != 0 and the file carries any); a separate
// instantiation keeps its ~25 VGPRs out of the common kernel (4 instead of 3 waves per SIMD)
template <int NW, bool IBRD>
__global__ __launch_bounds__(NW * 64) void lines_kernel(ModmArgs a, DevLines L, DevTables tb) {
    constexpr int NT = NW * 64;
    __shared__ HotA sA[NT];
    __shared__ HotB sB[NT];
    __shared__ double sWn[NT];  // the tile's wavenumbers (ascending)
    __shared__ double sLay[20];  // layer scalars: parked here so they do not occupy registers during the evaluate loops
    // per chunk parity, one bit per molecule: may a lane of the tile need a Voigt shape / a negative resonance?
    __shared__ unsigned long long sMaskV[2], sMaskM2[2];
    __shared__ ColdLine sCold[NT];
    // per-molecule tables sized by nmol (dynamic LDS, lines_dyn_lds()): a 64-thread block must stay under
    // ~8 KB of LDS or the 160 KB of a CU, not the registers, limit the resident waves
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double *sScor = dyn_lds;                     // [nmol*9] Q(296)/Q(T) per (mol, iso)
    double *sDop = sScor + a.nmol * 9;           // [nmol*9] HWHM_D / Xnu per (mol, iso)
    double *sW = sDop + a.nmol * 9;              // [nmol]   column amounts
    int *sLo = reinterpret_cast<int *>(sW + a.nmol);  // [nmol]   first candidate line
    int *sOff = sLo + a.nmol;                    // [nmol+1] prefix sums of the candidate counts

    const int tid = threadIdx.x;
    const int nslice = a.nslice;
    const int tile = blockIdx.x / nslice, slice = blockIdx.x % nslice, lay = blockIdx.y, prof = blockIdx.z;
    const int nwn = a.nwn, nmol = a.nmol;
    const int iw = tile * NT + tid;
    const bool valid = iw < nwn;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    double *obm = (nslice == 1) ? a.O_BY_MOL + pl * nmol * (size_t)nwn
                                : a.partial + ((size_t)slice * a.nprof * a.nlay_max + pl) * nmol * (size_t)nwn;

    // outputs start from zero: molecules without lines / zero column keep it (modm.f90:314, :318-321)
    if (valid)
        for (int m = 0; m < nmol; m++) obm[(size_t)m * nwn + iw] = 0.;
    if (lay >= a.nlay[prof]) return;

    const double WN = a.wn[valid ? iw : nwn - 1];
    const double Pk = a.P[pl], Tk = a.T[pl], wbrod = a.WBRODL[pl];
    const double *wk = a.WKL + pl * nmol;

    // ---- layer scalars (INITI + head of LINES: modm.f90:868-883, :301-314) -------------------------
    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    const double XN0 = (K_P0 / (K_BOLTZ * K_T0)) * 1.E+3;
    const double Xn = (Pk / (K_BOLTZ * Tk)) * 1.E+3;
    double WTOT = 0.;
    for (int m = 0; m < nmol; m++) WTOT += wk[m];
    WTOT = WTOT + wbrod;
    const double RP = Pk / K_P0, RP2 = RP * RP;
    const double RT = Tk / K_T0, RHORAT = Xn / XN0;
    int ILC = (Tk < 250.0) ? 1 : ((Tk < 296.0) ? 2 : 3);  // TEMPLC = 200,250,296,340
    const double tlo = (ILC == 1) ? 200.0 : (ILC == 2 ? 250.0 : 296.0);
    const double thi = (ILC == 1) ? 250.0 : (ILC == 2 ? 296.0 : 340.0);
    const double RECTLC = 1.0 / (thi - tlo), TMPDIF = Tk - tlo;
    const double RFT = WN * tanh((RADCT * WN) / (2 * Tk));
    const double lnRT = log(RT);
    const double cTk = RADCT / Tk, cT0 = RADCT / K_T0, dTinv = 1.0 / K_T0 - 1.0 / Tk;  // wave-uniform INTENS factors

    for (int m = tid; m < nmol; m += NT) sW[m] = wk[m];
    sWn[tid] = WN;  // lanes past nwn repeat the last wavenumber: still ascending
    if (tid == 0) {
        sLay[0] = RHORAT; sLay[1] = RP; sLay[2] = RP2; sLay[3] = lnRT; sLay[4] = cTk; sLay[5] = cT0; sLay[6] = dTinv;
        sLay[7] = RECTLC; sLay[8] = TMPDIF; sLay[9] = WTOT;
        for (int j = 0; j < MXBRD; j++) sLay[10 + j] = RHORAT * wk[j] / WTOT;  // rho_molec(1:7), modm.f90:313
    }
    if (tid < 2) {
        sMaskV[tid] = 0ull;
        sMaskM2[tid] = 0ull;
    }
    // TIPS + Doppler factor per (mol, iso): src/tips_2003.f90:60-296, src/modm.f90:442-454
    for (int t = tid; t < nmol * 9; t += NT) {
        const int mol = t / 9 + 1, iso = t % 9 + 1;
        double sc = 0., dop = 0.;
        const int niso = min(9, tb.tips_isonm[mol - 1]);
        if (iso <= niso) {
            if (mol == 34) sc = 1.;
            else if (mol == 39) sc = 296. / ((Tk / 296.) * sqrt(Tk / 296.));
            else {
                if (Tk < 70. || Tk > 3000.) atomicOr(a.errflag, ERRBIT_TEMP);
                else {
                    const double *Q = tb.tips_qoft + (size_t)(tb.tips_offset[mol - 1] + iso - 1) * 119;
                    const double q296 = tb.tips_q296[tb.tips_offset[mol - 1] + iso - 1], qt = tips_atob(Tk, Q);
                    if (qt <= 0.) atomicOr(a.errflag, ERRBIT_TEMP);
                    sc = q296 / qt;
                }
            }
        }
        const double M = tb.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = sqrt(2. * log(2.) * ((K_BOLTZ * Tk) / (M / K_AVOGAD))) / K_CLIGHT;
        sScor[t] = sc;
        sDop[t] = dop;
    }

    // ---- candidate range of every active molecule for this wavenumber tile ------------------------
    const double wnlo = a.wn[tile * NT], wnhi = a.wn[min(nwn, (tile + 1) * NT) - 1];
    const double pad = 3.0 * L.max_abs_shift * fmax(RHORAT, 1.0) + 1e-6;
    for (int m = tid; m < nmol; m += NT) {
        const int mol = m + 1;
        int lo = L.mol_start[mol], hi = L.mol_start[mol + 1];
        if (wk[m] == 0.) hi = lo;  // W_SPECIES == 0 -> OL = 0 (modm.f90:318-321)
        else if (mol != 7 && ((L.sorted_mask >> mol) & 1ull)) {
            // 25 cm-1 rule (modm.f90:384): only lines with |WN - Xnu| <= 25 for some WN of the tile matter
            const double vlo = wnlo - 25.0 - pad, vhi = wnhi + 25.0 + pad;
            int l0 = lo, l1 = hi;
            while (l0 < l1) { int mid = (l0 + l1) >> 1; if (L.vnu[mid] < vlo) l0 = mid + 1; else l1 = mid; }
            const int first = l0;
            l1 = hi;
            while (l0 < l1) { int mid = (l0 + l1) >> 1; if (L.vnu[mid] <= vhi) l0 = mid + 1; else l1 = mid; }
            lo = first;
            hi = l0;
        }
        sLo[m] = lo;
        sOff[m + 1] = hi - lo;  // count, prefix-summed below
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        sOff[0] = 0;
        for (int m = 0; m < nmol; m++) { acc += sOff[m + 1]; sOff[m + 1] = acc; }
    }
    __syncthreads();
    const int total = sOff[nmol];
    // this block's share of the candidate lines (the whole list when nslice == 1)
    const int vbeg = (int)(((long long)total * slice) / nslice), vend = (int)(((long long)total * (slice + 1)) / nslice);


    double SF = 0.;

#ifdef MONORTM_ABLATE_LOOP
    if (a.nwn > 0) return;  // timing experiment: prologue only
#endif
    for (int base = vbeg, ck = 0; base < vend; base += NT, ck++) {
        // ================= prepare: one lane per line ================================================
        const int v = base + tid;
        if (v < vend) {
            const double RHORAT = sLay[0], RP = sLay[1], RP2 = sLay[2], lnRT = sLay[3], cTk = sLay[4], cT0 = sLay[5],
                         dTinv = sLay[6], RECTLC = sLay[7], TMPDIF = sLay[8], WTOT = sLay[9];
            double rho7[MXBRD];
#pragma unroll
            for (int j = 0; j < MXBRD; j++) rho7[j] = IBRD ? sLay[10 + j] : 0.;
            int m = 0;
            while (sOff[m + 1] <= v) m++;
            const int idx = sLo[m] + (v - sOff[m]);
            const int mol = m + 1;
            const uint32_t meta = L.meta[idx];
            const int iso = (meta >> 6) & 15, code = (meta >> 10) & 3;
            const double xnu0 = L.vnu[idx];
            double alpf = L.alfa[idx], alps = L.hwhm[idx], delt = L.pshift[idx];
            const double E = L.epp[idx], XTILD = L.tmpalf[idx];
            if ((meta >> 13) & 1) {  // O2 / N2: air width -> foreign width (lnfl_mod.f90:98-113)
                const double rvmr = (mol == 7) ? 0.21 : 0.79;
                alpf = (alpf - rvmr * alps) / (1.0 - rvmr);
            }
            if ((meta >> 14) & 1) {
                const double rvmr = 0.21;
                delt = (delt - rvmr * (double)L.brd_dat[(size_t)idx * 21 + 3 * 6 + 2]) / (1.0 - rvmr);
            }
            const double rho_self = (mol <= MXBRD) ? sLay[10 + mol - 1] : RHORAT * sW[mol - 1] / WTOT;
            // line-coupling coefficients at the layer temperature (modm.f90:328-368)
            double AIP = 0., BIP = 0.;
            if (code) {
                const double *s = L.lc + (size_t)(meta >> 15) * 8;
                double A0 = s[ILC - 1], A1 = s[ILC], B0 = s[4 + ILC - 1], B1 = s[4 + ILC];
                if ((meta >> 12) & 1) {
                    const double rho_for = (RHORAT - rho_self) / RHORAT, rho_sel = rho_self / RHORAT;
                    A0 = rho_for * A0 + rho_sel * s[8 + ILC - 1];
                    A1 = rho_for * A1 + rho_sel * s[8 + ILC];
                    B0 = rho_for * B0 + rho_sel * s[12 + ILC - 1];
                    B1 = rho_for * B1 + rho_sel * s[12 + ILC];
                }
                AIP = A0 + ((A1 - A0) * RECTLC) * TMPDIF;
                BIP = B0 + ((B1 - B0) * RECTLC) * TMPDIF;
                if (code == 1) { AIP = AIP * a.sclcpl + a.y0res; BIP = BIP * a.sclcpl + a.y0res; }
                if (code == 2) { AIP = AIP * a.sclhw; BIP = BIP * a.sclhw; }
            }
            double Xnu = xnu0 + (delt * RHORAT);
            const bool brd = IBRD && mol <= MXBRD;
            int bf[MXBRD];
            int sflg = 0;
            if (brd) {
                double s = 0.;
#pragma unroll
                for (int j = 0; j < MXBRD; j++) {
                    bf[j] = L.brd_flg[(size_t)idx * 7 + j];
                    sflg += bf[j];
                    s += rho7[j] * bf[j] * ((double)L.brd_dat[(size_t)idx * 21 + 3 * j + 2] - delt);
                }
                Xnu = Xnu + s;
            }
            // INTENS (modm.f90:860-865); exp(a)/exp(b) folded into one exp
            const double XIPSF = iso ? sScor[(mol - 1) * 9 + iso - 1] : 0.;
            const double S = L.s0adj[idx] * exp((RADCT * E) * dTinv) * XIPSF;
            const double STILD = S * ((1 + exp(-(Xnu * cTk))) * frcp_any(Xnu * (1 - exp(-(Xnu * cT0)))));
            // HALFWHM_C (modm.f90:833-857)
            if (mol == 1 && alps == 0.) alps = 5 * alpf;
            const double rtx = exp(XTILD * lnRT);
            const double alfa0i = alpf * rtx, hwhmsi = alps * rtx;
            double HW = alfa0i * (RHORAT - rho_self) + hwhmsi * rho_self;
            if (brd && sflg > 0) {
                double alfsum = 0., rsum = 0.;
#pragma unroll
                for (int j = 0; j < MXBRD; j++) {
                    const double hwj = L.brd_dat[(size_t)idx * 21 + 3 * j], tmj = L.brd_dat[(size_t)idx * 21 + 3 * j + 1];
                    alfsum += rho7[j] * bf[j] * (hwj * exp(tmj * lnRT));
                    rsum += rho7[j] * bf[j];
                }
                HW = (RHORAT - rsum) * alfa0i + alfsum;
                if (bf[mol - 1] == 0) HW = HW + rho7[mol - 1] * (hwhmsi - alfa0i);
            }
            const double HWD = Xnu * (iso ? sDop[(mol - 1) * 9 + iso - 1] : sDop[(mol - 1) * 9]);
            if (code == 2) HW = HW * (1 - (AIP * (RP)) - (BIP * (RP2)));
            const double zeta = HW / (HW + HWD);
            // which shapes carry the Y factors (modm.f90:706-831): every coupled generic / CO2(-1,-5) line,
            // O2 only for XG = -1
            const bool yfac = code != 0 && ((mol != 7 && mol != 2) || (mol == 7 && code == 1) || (mol == 2 && code != 2));
            const double c1 = yfac ? AIP * frcp_any(HW) * RP : 0.;
            const double g = yfac ? BIP * RP2 : 0.;
            const double A2 = STILD * HW * (1.0 / K_PI);
            const double HW2 = HW * HW;
            const double p = A2 * frcp_any(625. + HW2);
            HotA h;
            HotB hb;
            h.xnu = Xnu;
            h.hw2 = HW2;
            h.a2 = A2;
            if (mol == 7) {
                // O2: no pedestal.  Uncoupled lines obey the 25 cm-1 rule inside the shape function and add the
                // negative resonance only when WN+Xnu <= 25; coupled lines use both resonances everywhere
                // (modm.f90:755-792)
                h.pa = code ? __builtin_inf() : 25.;
                hb.pb = code ? __builtin_inf() : 25.;
            } else {
                // generic molecules: pedestal with its coupling factors Y1P / Y2P; CO2: bare pedestal (it is
                // multiplied by (2 - d^2/625) and by Y1 per wavenumber, modm.f90:808-817)
                h.pa = (mol == 2) ? p : p * ((1. + c1 * 25.) + g);
                hb.pb = p * ((1. - c1 * 25.) + g);
            }
            hb.c1 = c1;
            hb.gp1 = 1. + g;
            // Voigt is only possible when zeta <= 0.99 AND some wavenumber of the tile lies within 100 Doppler
            // widths of the centre (modm.f90:427): look up the nearest one (sWn is sorted)
            double d100 = -1.0;
            if (!(zeta > 0.99)) {
                const double lim = 100. * HWD;
                int lo = 0, hi = NT;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sWn[mid] < Xnu) lo = mid + 1;
                    else hi = mid;
                }
                double best = __builtin_inf();
                if (lo < NT) best = fabs(sWn[lo] - Xnu);
                if (lo > 0) best = fmin(best, fabs(sWn[lo - 1] - Xnu));
                if (!(best > lim)) {
                    d100 = lim;
                    atomicOr(&sMaskV[ck & 1], 1ull << mol);
                }
            }
            hb.d100 = d100;
            // negative resonance: WN + Xnu <= 25 (<= +inf for coupled O2) possible for the tile's lowest wavenumber?
            if (mol != 2 && sWn[0] + Xnu <= ((mol == 7 && code) ? __builtin_inf() : 25.)) atomicOr(&sMaskM2[ck & 1], 1ull << mol);
            sA[tid] = h;
            sB[tid] = hb;
            ColdLine c;
            c.stild = STILD;
            c.hw = HW;
            c.hwd = HWD;
            c.sdep = L.sdep[idx];
            c.info = (uint32_t)mol | ((uint32_t)code << 6);
            sCold[tid] = c;
        }
        __syncthreads();

        // ================= evaluate: every wave walks the prepared lines, molecule by molecule =========
        const unsigned long long maskV = sMaskV[ck & 1], maskM2 = sMaskM2[ck & 1];
        if (tid == 0) {  // next chunk's flags; their last readers passed the barrier above
            sMaskV[(ck + 1) & 1] = 0ull;
            sMaskM2[(ck + 1) & 1] = 0ull;
        }
#ifdef MONORTM_ABLATE_EVAL
        if (a.nwn > 0) { __syncthreads(); continue; }  // timing experiment: prologue + prepare only
#endif
        for (int m = 0; m < nmol; m++) {
            // the molecule's run restricted to this block's slice
            const int s0 = max(sOff[m], vbeg), s1 = min(sOff[m + 1], vend);
            if (s1 <= base || s0 >= s1) continue;
            if (s0 >= base + NT) break;
            const int j0 = max(s0, base) - base, j1 = min(s1, base + NT) - base;
            if (s0 >= base) SF = 0.;  // the molecule's run starts in this chunk
            const int mol = m + 1;
            const bool lc = (L.lc_mask >> mol) & 1ull;
            const bool vg = (maskV >> mol) & 1ull, m2 = (maskM2 >> mol) & 1ull;
            if (mol == 7) SF = eval_dispatch<1>(lc, vg, m2, sA, sB, sCold, j0, j1, WN, mol, SF, a.errflag);
            else if (mol == 2) SF = eval_dispatch<2>(lc, vg, m2, sA, sB, sCold, j0, j1, WN, mol, SF, a.errflag);
            else SF = eval_dispatch<0>(lc, vg, m2, sA, sB, sCold, j0, j1, WN, mol, SF, a.errflag);
            // run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438)
            if (s1 <= base + NT && valid) obm[(size_t)m * nwn + iw] = RFT * (sW[m] * SF);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// continuum helpers (device)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double radfn(double VI, double XKT) {  // src/lblrtm_sub.f90:36-97
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

struct AccGrid {
    double V1C, V2C, DVC;
    int NPTC, I1;
};
// grid set-up shared by SL296 / SL260 / FRN296 / FRNCO2 / xn2_r (src/contnm.f90:1441-1459)
__device__ AccGrid acc_grid(double V1ABS, double V2ABS, double V1S, double DVS, int NPTS) {
    AccGrid g;
    g.DVC = DVS;
    g.V1C = V1ABS - g.DVC;
    g.V2C = V2ABS + g.DVC;
    if (g.V1C < V1S) g.I1 = -1;
    else g.I1 = (int)((g.V1C - V1S) / DVS + 0.01);
    g.V1C = V1S + DVS * (double)(g.I1 - 1);
    int I2 = (int)((g.V2C - V1S) / DVS + 0.01);
    g.NPTC = I2 - g.I1 + 3;
    if (g.NPTC > NPTS) g.NPTC = NPTS + 4;
    g.V2C = g.V1C + DVS * (double)(g.NPTC - 1);
    return g;
}

// one interpolated value of XINT (src/lblrtm_sub.f90:22-30); A is 1-based
__device__ __forceinline__ double xint_point(double V1A, double DVA, const double *A, double VI) {
    const double RECDVA = 1. / DVA;
    int J = (int)((VI - V1A) * RECDVA + K_ONEPL);
    double VJ = V1A + DVA * (double)(J - 1);
    double P = RECDVA * (VI - VJ);
    double C = (3. - 2. * P) * P * P;
    double B = 0.5 * P * (1. - P);
    double B1 = B * (1. - P);
    double B2 = B * P;
    return -A[J - 1] * B1 + A[J] * (1. - C + B2) + A[J + 1] * (C + B1) - A[J + 2] * B2;
}

// XINT of the coarse array sC (grid g) accumulated into sAbs[ist..last] on the 1 cm-1 grid
__device__ void xint_to_abs(const AccGrid &g, const double *sC, double V1ABS, double DVABS, int NPTABS, double v1ss,
                            double v2ss, double *sAbs, int ist_min = 1, int last_max = 1 << 30) {
    // pre_xint (src/contnm.f90:1146-1164)
    int ist = (int)(2 + (v1ss - V1ABS) / DVABS + 1.e-5);
    if (ist < 1) ist = 1;
    int last = (int)(1 + (v2ss - V1ABS) / DVABS + 1.e-5);
    if (last > NPTABS) last = NPTABS;
    if (ist < ist_min) ist = ist_min;      // O3 Hartley-Huggins / UV seam at 40800 cm-1 (contnm.f90:579-599, :620-640)
    if (last > last_max) last = last_max;
    int ILO = (int)((g.V1C + g.DVC - V1ABS) / DVABS + 1. + K_ONEMI);
    if (ILO < ist) ILO = ist;
    int IHI = (int)((g.V2C - g.DVC - V1ABS) / DVABS + K_ONEMI);
    if (IHI > last) IHI = last;
    for (int I = ILO + (int)threadIdx.x; I <= IHI; I += blockDim.x) {
        double VI = V1ABS + DVABS * (double)(I - 1);
        sAbs[I] = sAbs[I] + xint_point(g.V1C, g.DVC, sC, VI) * 1.0;
    }
}

// accessor grid with a selectable index fudge and optional table-length cap (O2FUV: 1.e-5, contnm.f90:9968;
// O2HERZ: no table, no cap, :9820)
__device__ AccGrid acc_grid2(double V1ABS, double V2ABS, double V1S, double DVS, int NPTS, double fudge, bool cap) {
    AccGrid g;
    g.DVC = DVS;
    g.V1C = V1ABS - g.DVC;
    g.V2C = V2ABS + g.DVC;
    if (g.V1C < V1S) g.I1 = -1;
    else g.I1 = (int)((g.V1C - V1S) / DVS + fudge);
    g.V1C = V1S + DVS * (double)(g.I1 - 1);
    int I2 = (int)((g.V2C - V1S) / DVS + fudge);
    g.NPTC = I2 - g.I1 + 3;
    if (cap && g.NPTC > NPTS) g.NPTC = NPTS + 4;
    g.V2C = g.V1C + DVS * (double)(g.NPTC - 1);
    return g;
}

// One tabulated continuum branch: coarse coefficients f(I, VJ) on grid g -> XINT onto the 1 cm-1 ABSRB grid.
// Called by the whole block (contains barriers).
template <class F>
__device__ __forceinline__ void cont_branch(const AccGrid &g, double v1ss, double v2ss, double V1ABS, double DVABS, int NPTABS,
                                            int csize, double *sC, double *sAbs, F f, int ist_min = 1, int last_max = 1 << 30) {
    for (int J = threadIdx.x; J <= g.NPTC + 2 && J < csize; J += blockDim.x) {
        double v = 0.;
        if (J >= 1 && J <= g.NPTC) v = f(g.I1 + (J - 1), g.V1C + g.DVC * (double)(J - 1));
        sC[J] = v;
    }
    __syncthreads();
    xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, v1ss, v2ss, sAbs, ist_min, last_max);
    __syncthreads();
}

__device__ double odclw_tkc(double WN, double TEMP, double CLW) {  // src/CloudOptProp.f90:29-157
    const double Hz_per_GHz = 1.e9, cm_per_m = 100.;
    const double a_1 = 8.110808E+01, b_1 = 4.433736E-03, c_1 = 1.301700E-13, d_1 = 6.627126E+02, a_2 = 2.025164E+00,
                 b_2 = 1.072976E-02, c_2 = 1.011945E-14, d_2 = 6.089168E+02, t_c = 1.342433E+02;
    double freq = WN * K_CLIGHT / Hz_per_GHz;
    double temp = TEMP - 273.15;
    double frq = freq * Hz_per_GHz;
    double cl = K_CLIGHT / cm_per_m;
    double eps_s = 87.9144 - 0.404399 * temp + 9.58726e-4 * (temp * temp) - 1.32802e-6 * (temp * temp * temp);
    double delta_1 = a_1 * exp(-b_1 * temp), tau_1 = c_1 * exp(d_1 / (temp + t_c));
    double delta_2 = a_2 * exp(-b_2 * temp), tau_2 = c_2 * exp(d_2 / (temp + t_c));
    double w1 = 2. * K_PI * frq * tau_1, w2 = 2. * K_PI * frq * tau_2, w = 2. * K_PI * frq;
    double t1 = (tau_1 * tau_1 * delta_1) / (1. + w1 * w1);
    double t2 = (tau_2 * tau_2 * delta_2) / (1. + w2 * w2);
    double eps1 = eps_s - (w * w) * (t1 + t2);
    t1 = (tau_1 * delta_1) / (1. + w1 * w1);
    t2 = (tau_2 * delta_2) / (1. + w2 * w2);
    double eps2 = w * (t1 + t2);
    cx eps = cmk(eps1, eps2);
    cx RE = (cmk(eps1 - 1., eps2)) / (2. + eps);
    double alpha = 6. * K_PI * RE.im * frq * 1.e-3 / cl;
    return alpha * CLW;
}

// ------------------------------------------------------------------------------------------------
// finish_kernel: continuum (CONTNM x 6, modm.f90:207-247), cloud (modm.f90:264), totals (:265-269)
// grid = (layers, profiles); dynamic LDS: sAbs[NPTABS+4] + sC[NPTABS/2+24]
// ------------------------------------------------------------------------------------------------
// HIGH: the spectral range reaches above 1340 cm-1, where the O3 / O2 / N2-fundamental continua live; the microwave /
// far-infrared instantiation leaves that code (and its registers) out
template <bool HIGH>
__global__ __launch_bounds__(256) void finish_kernel(ModmArgs a, DevTables tb, double V1ABS, double V2ABS, int NPTABS,
                                                     int csize) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sAbs = smem;               // 1-based, [0..NPTABS+3]
    double *sC = smem + NPTABS + 4;    // 1-based coarse array
    const int lay = blockIdx.x, prof = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int nwn = a.nwn, nmol = a.nmol;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    double *O = a.O + pl * (size_t)nwn, *OCLW = a.O_CLW + pl * (size_t)nwn;
    double *OC = a.OC + pl * MONORTM_NCONT * (size_t)nwn;
    if (lay >= a.nlay[prof]) {
        for (int iw = tid; iw < nwn; iw += nt) {
            O[iw] = 0.;
            OCLW[iw] = 0.;
            for (int s = 0; s < MONORTM_NCONT; s++) OC[(size_t)s * nwn + iw] = 0.;
            for (int m = 0; m < nmol; m++) a.O_BY_MOL[(pl * nmol + m) * (size_t)nwn + iw] = 0.;
        }
        return;
    }
    const double DVABS = 1.0;
    const double PAVE = a.P[pl], TAVE = a.T[pl], WBROAD = a.WBRODL[pl], CLW = a.CLW[pl];
    const double *wk = a.WKL + pl * nmol;
    const double V1 = a.wn[0], V2 = a.wn[nwn - 1];
    const double P0c = 1013., T0c = 296., XLOSMT = 2.68675E+19;
    const double RHOAVE = (PAVE / P0c) * (T0c / TAVE);
    const double XKT = TAVE / K_RADCN2;
    const double amagat = (PAVE / P0c) * (273. / TAVE);
    double WTOT = WBROAD;
    for (int m = 0; m < nmol; m++) WTOT = WTOT + wk[m];
    const double WK1 = wk[0], WK2 = wk[1], WK7 = wk[6];
    const double x_vmr_h2o = WK1 / WTOT, x_vmr_o2 = WK7 / WTOT, x_vmr_n2 = 1. - x_vmr_h2o - x_vmr_o2;
    const double wn2 = x_vmr_n2 * WTOT;
    const double h2o_fac = WK1 / WTOT;

    for (int pass = 0; pass < 6; pass++) {
        // oneMolecCntnm (src/CntnmFactors.f90:95-139): only this pass's factors are non-zero
        const double xself = pass == 0 ? a.cntnm[0] : 0., xfrgn = pass == 0 ? a.cntnm[1] : 0.;
        const double xco2c = pass == 1 ? a.cntnm[2] : 0., xn2cn = pass == 4 ? a.cntnm[5] : 0.;
        const double xo3cn = pass == 2 ? a.cntnm[3] : 0., xo2cn = pass == 3 ? a.cntnm[4] : 0.;
        const double xrayl = pass == 5 ? a.cntnm[6] : 0.;
        // passes whose every branch is switched off (or lies outside the spectral range: O3 and O2 have no
        // continuum below 1340 cm-1) leave ABSRB = 0: store the zeros directly
        const bool active = (pass == 0 && V2 > -20.0 && V1 < 20000. && (xself > 0. || xfrgn > 0.)) ||
                            (pass == 1 && V2 > -20.0 && V1 < 10000. && xco2c > 0.) ||
                            (HIGH && pass == 2 && V2 > 8920.0 && V1 < 54000. && xo3cn > 0.) ||
                            (HIGH && pass == 3 && V2 > 1340.0 && xo2cn > 0.) ||
                            (pass == 4 && xn2cn > 0. && ((V2 > -10.0 && V1 < 350.) || (HIGH && V2 > 2001.77 && V1 < 4910.))) ||
                            (pass == 5 && V2 >= 820. && xrayl > 0.);
        if (!active) {
            for (int iw = tid; iw < nwn; iw += nt) {
                if (pass < 5) OC[(size_t)pass * nwn + iw] = 0.;
                else O[iw] = 0.;
            }
            continue;
        }
        for (int i = tid; i < NPTABS + 4; i += nt) sAbs[i] = 0.;
        __syncthreads();
        if (pass == 0 && V2 > -20.0 && V1 < 20000. && xself > 0.) {  // H2O self, contnm.f90:325-371
            const double Rself = h2o_fac * RHOAVE * 1.e-20 * xself;
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_SELF296_V1, MT_SELF296_DV, MT_SELF296_NPT);
            const double TFAC = (TAVE - T0c) / (260. - T0c);
            for (int J = tid; J <= g.NPTC + 2 && J < csize; J += nt) {
                double v = 0.;
                const int I = g.I1 + (J - 1);
                if (J >= 1 && J <= g.NPTC && I >= 1 && I <= MT_SELF296_NPT) {
                    const double s0 = tb.self296[I - 1], s1 = tb.self260[I - 1];
                    double SH2O = 0.;
                    if (s0 > 0.) SH2O = s0 * powpos(s1 / s0, TFAC);
                    v = WK1 * (SH2O * Rself);
                }
                sC[J] = v;
            }
            __syncthreads();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_SELF296_V1, MT_SELF296_V2, sAbs);
            __syncthreads();
        }
        if (pass == 0 && V2 > -20.0 && V1 < 20000. && xfrgn > 0.) {  // H2O foreign, contnm.f90:380-474
            const double Rfrgn = (1. - h2o_fac) * RHOAVE * 1.e-20 * xfrgn;
            const double f0 = 0.06, V0F1 = 255.67, HWSQ1 = 240. * 240., BETA1 = 57.83, C_1 = -0.42, C_2 = 0.3, BETA2 = 630.;
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_FRGN296_V1, MT_FRGN296_DV, MT_FRGN296_NPT);
            for (int J = tid; J <= g.NPTC + 2 && J < csize; J += nt) {
                double v = 0.;
                const int I = g.I1 + (J - 1);
                if (J >= 1 && J <= g.NPTC) {
                    double FH2O = (I >= 1 && I <= MT_FRGN296_NPT) ? tb.frgn296[I - 1] : 0.;
                    const double VJ = g.V1C + g.DVC * (double)(J - 1);
                    double FSCAL;
                    if (VJ <= 600.) {
                        const int JFAC = (int)((VJ + 10.) / 10. + 0.00001);
                        FSCAL = tb.xfac_rhu[JFAC + 1];
                    } else {
                        const double vdelsq1 = (VJ - V0F1) * (VJ - V0F1), vdelmsq1 = (VJ + V0F1) * (VJ + V0F1);
                        double t = (VJ - V0F1) / BETA1; t = t * t; t = t * t; const double VF1 = t * t;
                        t = (VJ + V0F1) / BETA1; t = t * t; t = t * t; const double VmF1 = t * t;
                        t = VJ / BETA2; t = t * t; t = t * t; const double VF2 = t * t;
                        FSCAL = 1. + (f0 + C_1 * ((HWSQ1 / (vdelsq1 + HWSQ1 + VF1)) + (HWSQ1 / (vdelmsq1 + HWSQ1 + VmF1)))) /
                                         (1. + C_2 * VF2);
                    }
                    FH2O = FH2O * FSCAL;
                    v = (WK1 * FH2O) * Rfrgn;
                }
                sC[J] = v;
            }
            __syncthreads();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_FRGN296_V1, MT_FRGN296_V2, sAbs);
            __syncthreads();
        }
        if (pass == 1 && V2 > -20.0 && V1 < 10000. && xco2c > 0.) {  // CO2, contnm.f90:484-528 + FRNCO2 :2958
            const double WCO2 = WK2 * RHOAVE * 1.0E-20 * xco2c;
            const double trat = TAVE / 246.;
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_FCO2_V1, MT_FCO2_DV, MT_FCO2_NPT);
            for (int J = tid; J <= g.NPTC + 2 && J < csize; J += nt) {
                double v = 0.;
                const int I = g.I1 + (J - 1);
                if (J >= 1 && J <= g.NPTC && I >= 1 && I <= MT_FCO2_NPT) {
                    double tcor = 1.;
                    if (I >= 1196 && I <= 1220) tcor = powpos(trat, tb.tdep_bandhead[I - 1196]);
                    double FCO2 = tcor * tb.fco2[I - 1];
                    const double VJ = g.V1C + g.DVC * (double)(J - 1);
                    double CFAC = 1.;
                    if (VJ >= 2000. && VJ <= 2998.) CFAC = tb.xfacco2[(int)((VJ - 1998.) / 2. + 0.00001) - 1];
                    FCO2 = CFAC * FCO2;
                    v = FCO2 * WCO2;
                }
                sC[J] = v;
            }
            __syncthreads();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_FCO2_V1, MT_FCO2_V2, sAbs);
            __syncthreads();
        }
        if (HIGH && pass == 2) {  // ---------------- O3 (contnm.f90:536-642)
            if (V2 > 8920.0 && V1 <= 24665.0 && xo3cn > 0.) {  // Chappuis / Wulf, XO3CHP :4685
                const double WO3 = wk[2] * 1.0E-20 * xo3cn, DT = TAVE - 273.15;
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O3CH_V1, MT_O3CH_DV, MT_O3CH_NPT);
                cont_branch(g, MT_O3CH_V1, MT_O3CH_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    double c0 = 0., c1 = 0., c2 = 0.;
                    if (I >= 1 && I <= MT_O3CH_NPT) { c0 = tb.o3ch_x[I - 1] / VJ; c1 = tb.o3ch_y[I - 1] / VJ; c2 = tb.o3ch_z[I - 1] / VJ; }
                    return (c0 + (c1 + c2 * DT) * DT) * WO3;
                });
            }
            const int I_FIX = (int)((40800. - V1ABS) / DVABS + 1.001);
            if (V2 > 27370. && V1 < 40800. && xo3cn > 0.) {  // Hartley-Huggins, O3HHT0/1/2 :6850-8216
                const double WO3 = wk[2] * 1.E-20 * xo3cn, TC = TAVE - 273.15;
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O3HH0_V1, MT_O3HH0_DV, MT_O3HH0_NPT);
                const bool seam = (g.V2C > 40815.) && (V2 > 40800);  // keep it below 40800 cm-1 (:579-599)
                cont_branch(g, MT_O3HH0_V1, MT_O3HH0_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    double c0 = 0., ct1 = 0., ct2 = 0.;
                    if (I >= 1 && I <= MT_O3HH0_NPT) { c0 = tb.o3hh0[I - 1] / VJ; ct1 = tb.o3hh1[I - 1]; ct2 = tb.o3hh2[I - 1]; }
                    double c = c0 * WO3;
                    return c * (1. + ct1 * TC + ct2 * TC * TC);
                }, 1, seam ? I_FIX - 1 : (1 << 30));
            }
            if (V2 > 40800. && V1 < 54000. && xo3cn > 0.) {  // UV, O3HHUV :8826 (no 1e-20 here)
                const double WO3 = wk[2] * xo3cn;
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O3HUV_V1, MT_O3HUV_DV, MT_O3HUV_NPT);
                cont_branch(g, MT_O3HUV_V1, MT_O3HUV_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    return ((I >= 1 && I <= MT_O3HUV_NPT) ? tb.o3huv[I - 1] / VJ : 0.) * WO3;
                }, (V1 < 40800) ? I_FIX : 1);
            }
        }
        if (HIGH && pass == 3) {  // ---------------- O2 (contnm.f90:657-878)
            if (V2 > 1340.0 && V1 < 1850. && xo2cn > 0.) {  // collision-induced fundamental, o2_ver_1 :8917
                const double tau_fac = xo2cn * WK7 * 1.e-20 * amagat;
                const double xktfac = (1. / 296.) - (1. / TAVE), factor = (1.e+20 / XLOSMT);
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O2F_V1, MT_O2F_DV, MT_O2F_NPT);
                cont_branch(g, MT_O2F_V1, MT_O2F_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    double c0 = 0.;
                    if (I >= 1 && I <= MT_O2F_NPT) c0 = factor * tb.o2f_x[I - 1] * exp(tb.o2f_t[I - 1] * xktfac) / VJ;
                    return tau_fac * c0;
                });
            }
            if (V2 > 7536.0 && V1 < 8500. && xo2cn > 0.) {  // 1.27 micron, O2INF1 :9047
                const double tau_fac = xo2cn * (WK7 / XLOSMT) * amagat *
                                       ((1. / 0.446) * x_vmr_o2 + (0.3 / 0.446) * x_vmr_n2 + 1. * x_vmr_h2o);
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O2INF1_V1, MT_O2INF1_DV, MT_O2INF1_NPT);
                cont_branch(g, MT_O2INF1_V1, MT_O2INF1_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    return tau_fac * ((I >= 1 && I <= MT_O2INF1_NPT) ? tb.o2inf1[I - 1] / VJ : 0.);
                });
            }
            if (V2 > 9100.0 && V1 < 11000. && xo2cn > 0.) {  // 1.06 micron, analytic: O2INF2 :9227
                const double V1S = 9100., V2S = 11000., DVS = 2.;
                const double WO2 = xo2cn * (WK7 * 1.e-20) * RHOAVE;
                const double ADJWO2 = (WK7 / WTOT) * (1. / 0.209) * WO2;
                AccGrid g;
                g.DVC = DVS;
                g.V1C = V1ABS - g.DVC;
                g.V2C = V2ABS + g.DVC;
                if (g.V1C < V1S) g.V1C = V1S - 2. * DVS;
                if (g.V2C > V2S) g.V2C = V2S + 2. * DVS;
                g.NPTC = (int)((g.V2C - g.V1C) / g.DVC + 3.01);
                g.V2C = g.V1C + g.DVC * (double)(g.NPTC - 1);
                g.I1 = 0;
                cont_branch(g, V1S, V2S, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int, double VJ) {
                    double c0 = 0.;
                    if (VJ > V1S && VJ < V2S) {
                        const double DV1 = VJ - 9375., DV2 = VJ - 9439., HW1 = 58.96, HW2 = 45.04;
                        const double DAMP1 = (DV1 < 0.0) ? exp(DV1 / 176.1) : 1.0, DAMP2 = (DV2 < 0.0) ? exp(DV2 / 176.1) : 1.0;
                        const double O2INF = 0.31831 * (((1.166E-04 * DAMP1 / HW1) / (1. + (DV1 / HW1) * (DV1 / HW1))) +
                                                        ((3.086E-05 * DAMP2 / HW2) / (1. + (DV2 / HW2) * (DV2 / HW2)))) * 1.054;
                        c0 = O2INF / VJ;
                    }
                    return c0 * ADJWO2;
                });
            }
            if (V2 > 12961.5 && V1 < 13221.5 && xo2cn > 0.) {  // A band, O2INF3 :9282
                const double tau_fac = xo2cn * (WK7 / XLOSMT) * amagat;
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O2INF3_V1, MT_O2INF3_DV, MT_O2INF3_NPT);
                cont_branch(g, MT_O2INF3_V1, MT_O2INF3_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    return tau_fac * ((I >= 1 && I <= MT_O2INF3_NPT) ? tb.o2inf3[I - 1] / VJ : 0.);
                });
            }
            if (V2 > 15000.0 && V1 < 29870. && xo2cn > 0.) {  // visible, O2_vis :9400
                const double WO2 = WK7 * 1.e-20 * ((PAVE / 1013.) * (273. / TAVE)) * xo2cn;
                const double ADJWO2 = (WK7 / WTOT) * WO2;
                const double t55 = (55. * 273. / 296.);
                const double factor = 1. / ((XLOSMT * 1.e-20 * (t55 * t55)) * 89.5);
                const AccGrid g = acc_grid(V1ABS, V2ABS, MT_O2VIS_V1, MT_O2VIS_DV, MT_O2VIS_NPT);
                cont_branch(g, MT_O2VIS_V1, MT_O2VIS_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    return ((I >= 1 && I <= MT_O2VIS_NPT) ? factor * tb.o2vis[I - 1] / VJ : 0.) * ADJWO2;
                });
            }
            if (V2 > 36000.0 && xo2cn > 0.) {  // Herzberg, O2HERZ / HERTDA / HERPRS :9808-9948
                const double WO2 = WK7 * 1.e-20 * xo2cn;
                const AccGrid g = acc_grid2(V1ABS, V2ABS, 36000., 10., 0, 0.01, false);
                cont_branch(g, 36000., 99999., V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    double c0 = 0.;
                    if (I >= 1) {
                        double HERZ = 0.0;
                        if (VJ > 36000.00) {
                            double CORR = 0.;
                            if (VJ <= 40000.) CORR = ((40000. - VJ) / 4000.) * 7.917E-07;
                            const double YRATIO = VJ / 48811.0, lg = log(YRATIO);
                            HERZ = 6.884E-04 * (YRATIO)*exp(-69.738 * (lg * lg)) - CORR;
                        }
                        HERZ = HERZ * (1. + .83 * (PAVE / 1013.) * (273.16 / TAVE));
                        c0 = HERZ / VJ;
                    }
                    return c0 * WO2;
                });
            }
            if (V2 > 56740.0 && xo2cn > 0.) {  // far UV (Schumann-Runge), O2FUV :9952
                const double WO2 = WK7 * 1.e-20 * xo2cn;
                const AccGrid g = acc_grid2(V1ABS, V2ABS, MT_O2FUV_V1, MT_O2FUV_DV, MT_O2FUV_NPT, 1.e-5, true);
                cont_branch(g, MT_O2FUV_V1, MT_O2FUV_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                    return ((I >= 1 && I <= MT_O2FUV_NPT) ? tb.o2fuv[I - 1] / VJ : 0.) * WO2;
                });
            }
        }
        if (pass == 4 && V2 > -10.0 && V1 < 350. && xn2cn > 0.) {  // N2 roto-translational, contnm.f90:906-943
            const double tau_fac = xn2cn * (wn2 / XLOSMT) * amagat;
            const double tfac = (TAVE - 296.) / (220. - 296.);
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_N2RT296_V1, MT_N2RT296_DV, MT_N2RT296_NPT);
            for (int J = tid; J <= g.NPTC + 2 && J < csize; J += nt) {
                double v = 0.;
                const int I = g.I1 + (J - 1);
                if (J >= 1 && J <= g.NPTC) {
                    double c0 = 0., c1 = 0.;
                    if (I >= 1 && I <= MT_N2RT296_NPT) {
                        c0 = tb.n2c296[I - 1] * powpos(tb.n2c220[I - 1] / tb.n2c296[I - 1], tfac);
                        const double sf_T = tb.n2sf296[I - 1] * powpos(tb.n2sf220[I - 1] / tb.n2sf296[I - 1], tfac);
                        c1 = (sf_T - 1.) * (0.79) / (0.21);
                    }
                    v = tau_fac * c0 * (x_vmr_n2 + c1 * x_vmr_o2 + 1. * x_vmr_h2o);
                }
                sC[J] = v;
            }
            __syncthreads();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_N2RT296_V1, MT_N2RT296_V2, sAbs);
            __syncthreads();
        }
        if (HIGH && pass == 4 && V2 > 2001.77 && V1 < 2897.59 && xn2cn > 0.) {  // N2 fundamental, contnm.f90:963-1009, n2_ver_1 :4331
            const double tau_fac = xn2cn * (wn2 / XLOSMT) * amagat;
            const double xtfac = ((1. / TAVE) - (1. / 272.)) / ((1. / 228.) - (1. / 272.));
            const double xt_lin = (TAVE - 272.) / (228. - 272.);
            const double a_o2 = 1.294 - 0.4545 * TAVE / 296.;
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_N2F_V1, MT_N2F_DV, MT_N2F_NPT);
            cont_branch(g, MT_N2F_V1, MT_N2F_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                double cn0 = 0., cn1 = 0., cn2 = 0.;
                if (I >= 1 && I <= MT_N2F_NPT) {
                    const double x272 = tb.n2f_272[I - 1], x228 = tb.n2f_228[I - 1];
                    if (x272 > 0. && x228 > 0.) cn0 = x272 * powpos(x228 / x272, xtfac);
                    else cn0 = x272 + (x228 - x272) * xt_lin;
                    cn0 = cn0 / VJ;
                    cn1 = a_o2 * cn0;
                    cn2 = (9. / 7.) * tb.n2f_ah2o[I - 1] * cn0;
                }
                return tau_fac * (x_vmr_n2 * cn0 + x_vmr_o2 * cn1 + x_vmr_h2o * cn2);
            });
        }
        if (HIGH && pass == 4 && V2 > 4340.0 && V1 < 4910. && xn2cn > 0.) {  // N2 first overtone, contnm.f90:1022-1068, :4579
            const double tau_fac = xn2cn * (wn2 / XLOSMT) * amagat * (x_vmr_n2 + 1. * x_vmr_o2 + 1. * x_vmr_h2o);
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_N2F1_V1, MT_N2F1_DV, MT_N2F1_NPT);
            cont_branch(g, MT_N2F1_V1, MT_N2F1_V2, V1ABS, DVABS, NPTABS, csize, sC, sAbs, [=](int I, double VJ) {
                return tau_fac * ((I >= 1 && I <= MT_N2F1_NPT) ? tb.n2f1[I - 1] / VJ : 0.);
            });
        }
        if (pass == 5 && V2 >= 820. && xrayl > 0.) {  // Rayleigh, contnm.f90:1107-1131 (JRAD = 0)
            const double conv_cm2mol = xrayl * 1.E-20 / (2.68675e-1 * 1.e5);
            for (int i = 1 + tid; i <= NPTABS; i += nt) {
                const double vr = V1ABS + (i - 1) * DVABS;
                const double xv = vr / 1.e4;
                double ray_ext = (xv * xv * xv / (9.38076E2 - 10.8426 * (xv * xv))) * (WTOT * conv_cm2mol);
                ray_ext = ray_ext * xv / radfn(vr, XKT);
                sAbs[i] = sAbs[i] + ray_ext;
            }
            __syncthreads();
        }
        // second interpolation ABSRB -> wavenumbers (modm.f90:216-246)
        for (int iw = tid; iw < nwn; iw += nt) {
            const double wnv = a.wn[iw];
            double val = 0.;
            if (a.dvset != 0.) {
                const int I = iw + 1;
                int ILO = (int)((V1ABS + DVABS - V1) / a.dvset + 1. + K_ONEMI);
                if (ILO < 1) ILO = 1;
                int IHI = (int)((V2ABS - DVABS - V1) / a.dvset + K_ONEMI);
                if (IHI > nwn) IHI = nwn;
                if (I >= ILO && I <= IHI) val = xint_point(V1ABS, DVABS, sAbs, V1 + a.dvset * (double)(I - 1));
            } else {
                int ILO = (int)((V1ABS + DVABS - wnv) / 1.0 + 1. + K_ONEMI);
                if (ILO < 1) ILO = 1;
                int IHI = (int)((V2ABS - DVABS - wnv) / 1.0 + K_ONEMI);
                if (IHI > 1) IHI = 1;
                if (ILO <= 1 && IHI >= 1) val = xint_point(V1ABS, DVABS, sAbs, wnv);
            }
            if (pass < 5) OC[(size_t)pass * nwn + iw] = val * radfn(wnv, XKT);
            else O[iw] = val * wnv / 1.0e4;  // oc_rayl parked in O until the totals below
        }
        __syncthreads();
    }
    // cloud liquid water + totals (modm.f90:264-269); same thread <-> same iw as above
    double *obm = a.O_BY_MOL + pl * nmol * (size_t)nwn;
    if (a.nslice > 1) {  // add the line slices in slice (= line) order
        const size_t sstride = (size_t)a.nprof * a.nlay_max * nmol * nwn;
        const double *part = a.partial + pl * nmol * (size_t)nwn;
        for (int iw = tid; iw < nwn; iw += nt)
            for (int m = 0; m < nmol; m++) {
                double acc = 0.;
                for (int sl = 0; sl < a.nslice; sl++) acc += part[(size_t)sl * sstride + (size_t)m * nwn + iw];
                obm[(size_t)m * nwn + iw] = acc;
            }
    }
    for (int iw = tid; iw < nwn; iw += nt) {
        const double wnv = a.wn[iw];
        const double oclw = (CLW == 0.) ? 0. : odclw_tkc(wnv, TAVE, CLW);  // alpha * 0 = 0 in the reference
        OCLW[iw] = oclw;
        double o = 0.;
        for (int m = 0; m < nmol; m++) o = o + obm[(size_t)m * nwn + iw];
        double soc = 0.;
        for (int s = 0; s < MONORTM_NCONT; s++) soc += OC[(size_t)s * nwn + iw];
        o = o + 0. + O[iw] + soc + oclw;
        O[iw] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// rtm_kernel: CALCTMR (RTMmono.f90:239-325) + RAD_UP_DN (:157-221) + RTM (:13-155); lane = (profile, wn)
// ------------------------------------------------------------------------------------------------
struct RtmArgs {
    int nprof, nwn, nlay_max, iout;
    const double *wn, *T, *TZ, *O, *emiss, *reflc;
    const int *nlay, *irt;
    double *tmpsfc, *RUP, *RDN, *TRTOT, *RAD, *TB, *TMR;
};

__device__ __forceinline__ double bb_fn(double v, double fbeta) { return K_RADCN1 * (v * v * v) / (exp(v * fbeta) - 1.); }

// Block = 64 wavenumbers x G layer groups.  The recurrences of RAD_UP_DN are sums of independent terms once
// the optical depth above / below a layer is known:  ODT after the reference's running subtraction equals the
// optical depth of the layers not yet visited.  Every thread walks its contiguous group of layers exactly like
// the reference (same running subtraction, same term formula), group partial sums are combined through LDS in
// the reference's visiting order (surface->top for RUP, top->surface for RDN / TMR).
template <int G>
__global__ __launch_bounds__(64 * G) void rtm_kernel(RtmArgs a) {
    __shared__ double sPart[G][64];
    __shared__ double sUp[G][64], sDn[G][64], sEx[G][64];
    const int lane = threadIdx.x, g = threadIdx.y;
    const int iw0 = blockIdx.x * 64 + lane, prof = blockIdx.y;
    const int nwn = a.nwn;
    const bool valid = iw0 < nwn;
    const int iw = valid ? iw0 : nwn - 1;
    const int nlay = a.nlay[prof], irt = a.irt[prof];
    const double VV = a.wn[iw];
    const double *O = a.O + (size_t)prof * a.nlay_max * nwn + iw;
    const double *T = a.T + (size_t)prof * a.nlay_max, *TZ = a.TZ + (size_t)prof * (a.nlay_max + 1);
    const int chunk = (nlay + G - 1) / G;
    const int l0 = min(nlay, g * chunk), l1 = min(nlay, l0 + chunk);  // 0-based layer range [l0, l1)

    double part = 0.;
    for (int l = l0; l < l1; l++) part = part + O[(size_t)l * nwn];
    sPart[g][lane] = part;
    __syncthreads();
    double below = 0., ODTOT = 0.;
    for (int gg = 0; gg < G; gg++) {
        if (gg < g) below = below + sPart[gg][lane];
        ODTOT = ODTOT + sPart[gg][lane];
    }
    const double above = ODTOT - below - part;

    double RUP = 0., RDN = 0., sumexp = 0.;
    if (irt != 3) {  // RTMmono.f90:193-205, layers l0+1 .. l1 (1-based) of the upward sweep
        double ODT = ODTOT - below;
        for (int l = l0 + 1; l <= l1; l++) {
            const double bb = bb_fn(VV, K_RADCN2 / T[l - 1]), bba = bb_fn(VV, K_RADCN2 / TZ[l]);
            const double ODVI = O[(size_t)(l - 1) * nwn];
            const double TRI = exp(-ODVI);
            ODT = ODT - ODVI;
            const double TR = exp(-ODT);
            const double pade = 0.193 * ODVI + 0.013 * (ODVI * ODVI);
            RUP = RUP + TR * (1. - TRI) * (bb + pade * bba) / (1. + pade);
        }
    }
    {  // RTMmono.f90:207-217 (and CALCTMR :302-315), layers l1 .. l0+1 of the downward sweep
        double ODT = ODTOT - above;
        for (int l = l1; l >= l0 + 1; l--) {
            const double bb = bb_fn(VV, K_RADCN2 / T[l - 1]), bba = bb_fn(VV, K_RADCN2 / TZ[l - 1]);
            const double ODVI = O[(size_t)(l - 1) * nwn];
            ODT = ODT - ODVI;
            const double TRI = exp(-ODVI);
            const double TR = exp(-ODT);
            const double pade = 0.193 * ODVI + 0.013 * (ODVI * ODVI);
            RDN = RDN + TR * (1. - TRI) * (bb + pade * bba) / (1. + pade);
            const double beff = (bb + pade * bba) / (1. + pade);
            sumexp = sumexp + beff * TR * (1 - TRI);
        }
    }
    sUp[g][lane] = RUP;
    sDn[g][lane] = RDN;
    sEx[g][lane] = sumexp;
    __syncthreads();
    if (g != 0 || !valid) return;
    RUP = 0.;
    RDN = 0.;
    sumexp = 0.;
    for (int gg = 0; gg < G; gg++) RUP = RUP + sUp[gg][lane];
    for (int gg = G - 1; gg >= 0; gg--) {
        RDN = RDN + sDn[gg][lane];
        sumexp = sumexp + sEx[gg][lane];
    }
    const double TRTOT = exp(-ODTOT);
    const size_t o = (size_t)prof * nwn + iw;
    if (a.TMR) {
        const double radtmr = sumexp / (1. - exp(-1 * ODTOT));
        const double x = K_RADCN1 * (VV * VV * VV) / radtmr + 1.;
        a.TMR[o] = K_RADCN2 * VV / log(x);
    }
    const double TSKY = 2.75;
    double tmpsfc = a.tmpsfc[prof];
    if (irt == 3 || irt == 2) tmpsfc = TSKY;  // RTMmono.f90:113-124
    const double SURFRAD = bb_fn(VV, K_RADCN2 / tmpsfc), COSMOS = bb_fn(VV, K_RADCN2 / TSKY);
    const double ESFC = a.emiss[o], RSFC = a.reflc[o];
    double RAD = 0.;
    if (irt == 1) RAD = RUP + TRTOT * (ESFC * SURFRAD + RSFC * (RDN + TRTOT * COSMOS));
    if (irt == 2) RAD = RUP + TRTOT * (RDN + TRTOT * COSMOS);
    if (irt == 3) RAD = RDN + (TRTOT * COSMOS);
    // TMPSFC is an in/out argument of the reference's RTM (RTMmono.f90:122).  Lanes of this profile that still
    // read the old value ignore it exactly when it is overwritten (irt = 2,3), so the store needs no ordering.
    if (iw == 0 && (irt == 3 || irt == 2)) a.tmpsfc[prof] = TSKY;
    a.RUP[o] = RUP;
    a.RDN[o] = RDN;
    a.TRTOT[o] = TRTOT;
    a.RAD[o] = RAD;
    if (a.iout == 1) {
        const double X = K_RADCN1 * (VV * VV * VV) / RAD + 1.;
        a.TB[o] = K_RADCN2 * VV / log(X);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: context, uploads, launches
// ------------------------------------------------------------------------------------------------
thread_local std::string g_init_error;

struct Ctx {
    int device = 0;
    std::string err;
    monortm::LineTable host;
    DevLines lines{};
    DevTables tables{};
    std::vector<void *> owned;
    int *errflag = nullptr;
    double *partial = nullptr;  // line-slice workspace, grown on demand
    size_t partial_elems = 0;
    int profiling = 0;  // bit k set: record events around kernel k
    struct Ev {
        hipEvent_t a, b;
        int k;
    };
    std::vector<Ev> events;
    double tot_ms[3] = {0, 0, 0};
    long long launches[3] = {0, 0, 0};
};

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return MONORTM_EHIP;                                                                     \
        }                                                                                            \
    } while (0)

template <class T>
int upload(Ctx *c, const T *src, size_t n, const T **dst) {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(c, hipMalloc(&p, bytes));
    c->owned.push_back(p);
    if (n) HIPCHK(c, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return MONORTM_OK;
}

void prof_begin(Ctx *c, hipStream_t s, int k, Ctx::Ev &ev) {
    ev.k = -1;
    if (!((c->profiling >> k) & 1)) return;
    hipEventCreate(&ev.a);
    hipEventCreate(&ev.b);
    ev.k = k;
    hipEventRecord(ev.a, s);
}
void prof_end(Ctx *c, hipStream_t s, Ctx::Ev &ev) {
    if (ev.k < 0) return;
    hipEventRecord(ev.b, s);
    c->events.push_back(ev);
}

int check_modm_args(Ctx *c, int nprof, int nwn, int nlay_max, int nmol, int ibrd, int ixsect, double v2) {
    if (nprof < 1 || nwn < 1 || nlay_max < 1 || nlay_max > 603) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    if (nmol < 7 || nmol > MXMOL) { c->err = "nmol must be 7..39 (LINES reads WK(1:7), modm.f90:313)"; return MONORTM_EARG; }
    if (nwn > 80000) { c->err = "nwn exceeds NWNMX=80000 (RTMmono.f90:10)"; return MONORTM_EARG; }
    if (ixsect != 0) { c->err = "IXSECT=1 (cross-section molecules) is outside the built path: no FSCDXS/xs data"; return MONORTM_EUNSUPPORTED; }
    if (ibrd != 0 && !c->host.any_brd) { /* nothing to do: flags all zero, same as ibrd = 0 */ }
    (void)v2;
    return MONORTM_OK;
}

}  // namespace

extern "C" {

const char *monortm_hip_last_error(void *ctx) {
    if (!ctx) return g_init_error.c_str();
    return static_cast<Ctx *>(ctx)->err.c_str();
}

int monortm_hip_init(const char *tape3_path, double v1, double v2, int icp, int real_kind, int device, void **out) {
    (void)icp;  // passed through to GET_LNFL by the reference and unused there (lnfl_mod.f90:22)
    *out = nullptr;
    if (real_kind != 8) { g_init_error = "real_kind must be 8 (double precision build)"; return MONORTM_EUNSUPPORTED; }
    Ctx *c = new Ctx;
    auto failed = [&](int rc) { g_init_error = c->err; for (void *p : c->owned) hipFree(p); delete c; return rc; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { c->err = "no HIP device available: the MI355X path has no CPU fallback"; return failed(MONORTM_EHIP); }
    if (device >= 0) { if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return failed(MONORTM_EHIP); } }
    hipGetDevice(&c->device);
    // an empty path gives a context without a line table (RTM / CALCTMR need no TAPE3)
    int rc = MONORTM_OK;
    if (tape3_path && tape3_path[0]) rc = monortm::load_tape3(tape3_path, v1, v2, c->host, c->err);
    if (rc) return failed(rc);
    const monortm::LineTable &h = c->host;
    DevLines &L = c->lines;
#define UP(field, vec) if ((rc = upload(c, (vec).data(), (vec).size(), &L.field))) return failed(rc)
    UP(vnu, h.vnu); UP(s0adj, h.s0adj); UP(lc, h.lc); UP(alfa, h.alfa); UP(hwhm, h.hwhm); UP(epp, h.epp);
    UP(tmpalf, h.tmpalf); UP(pshift, h.pshift); UP(sdep, h.sdep); UP(meta, h.meta); UP(brd_flg, h.brd_flg); UP(brd_dat, h.brd_dat);
#undef UP
    for (int m = 0; m <= MXMOL + 1; m++) L.mol_start[m] = h.mol_start[m];
    L.sorted_mask = 0;
    for (int m = 1; m <= MXMOL; m++) if (h.sorted[m]) L.sorted_mask |= (1ull << m);
    L.max_abs_shift = h.max_abs_shift;
    L.lc_mask = 0;
    for (size_t i = 0; i < h.meta.size(); i++)
        if ((h.meta[i] >> 10) & 3) L.lc_mask |= (1ull << (h.meta[i] & 63));
    DevTables &t = c->tables;
#define UT(field, arr) if ((rc = upload(c, arr, sizeof(arr) / sizeof(arr[0]), &t.field))) return failed(rc)
    UT(self296, MT_SELF296); UT(self260, MT_SELF260); UT(frgn296, MT_FRGN296); UT(fco2, MT_FCO2);
    UT(n2c296, MT_N2RT296_C); UT(n2sf296, MT_N2RT296_SF); UT(n2c220, MT_N2RT220_C); UT(n2sf220, MT_N2RT220_SF);
    UT(xfac_rhu, MT_XFAC_RHU); UT(xfacco2, MT_XFACCO2); UT(tdep_bandhead, MT_TDEP_BANDHEAD);
    UT(tips_qoft, TIPS_QOFT); UT(smass, ISO_SMASS); UT(tips_isonm, TIPS_ISONM); UT(tips_offset, TIPS_OFFSET);
    UT(o3ch_x, MT_O3CH_X); UT(o3ch_y, MT_O3CH_Y); UT(o3ch_z, MT_O3CH_Z); UT(o3hh0, MT_O3HH0); UT(o3hh1, MT_O3HH1);
    UT(o3hh2, MT_O3HH2); UT(o3huv, MT_O3HUV); UT(o2f_x, MT_O2F_XO2); UT(o2f_t, MT_O2F_XO2T); UT(o2inf1, MT_O2INF1);
    UT(o2inf3, MT_O2INF3); UT(o2vis, MT_O2VIS); UT(o2fuv, MT_O2FUV); UT(n2f_272, MT_N2F_272); UT(n2f_228, MT_N2F_228);
    UT(n2f_ah2o, MT_N2F_AH2O); UT(n2f1, MT_N2F1);
#undef UT
    {   // Q(296 K) of every isotopologue, interpolated exactly like Q(T)
        std::vector<double> q296(sizeof(TIPS_QOFT) / sizeof(double) / 119);
        for (size_t i = 0; i < q296.size(); i++) q296[i] = tips_atob(296., &TIPS_QOFT[i * 119]);
        if ((rc = upload(c, q296.data(), q296.size(), &t.tips_q296))) return failed(rc);
    }
    void *ef = nullptr;
    if (hipMalloc(&ef, sizeof(int)) != hipSuccess || hipMemset(ef, 0, sizeof(int)) != hipSuccess) { c->err = "hipMalloc(errflag) failed"; return failed(MONORTM_EHIP); }
    c->owned.push_back(ef);
    c->errflag = static_cast<int *>(ef);
    *out = c;
    return MONORTM_OK;
}

void monortm_hip_finalize(void *ctx) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return;
    hipSetDevice(c->device);
    for (auto &e : c->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (void *p : c->owned) hipFree(p);
    if (c->partial) hipFree(c->partial);
    delete c;
}

int monortm_hip_tape3_probe(const char *tape3_path, double v1, double v2, long long *n_physical, long long *n_entries,
                            long long *n_coupled) {
    monortm::LineTable t;
    std::string err;
    int rc = monortm::load_tape3(tape3_path ? tape3_path : "", v1, v2, t, err);
    if (rc) {
        g_init_error = err;
        return rc;
    }
    for (int m = 0; m <= MXMOL; m++) {
        n_physical[m] = t.n_physical[m];
        n_entries[m] = (m == 0) ? (long long)t.size() : t.mol_start[m + 1] - t.mol_start[m];
        n_coupled[m] = 0;
    }
    for (size_t i = 0; i < t.meta.size(); i++)
        if ((t.meta[i] >> 10) & 3) {
            n_coupled[t.meta[i] & 63]++;
            n_coupled[0]++;
        }
    return MONORTM_OK;
}

long long monortm_hip_line_count(void *ctx, int mol) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c || mol < 0 || mol > MXMOL) return -1;
    return c->host.n_physical[mol];
}

int monortm_hip_profile(void *ctx, int enable) {
    Ctx *c = static_cast<Ctx *>(ctx);
    c->profiling = enable;
    return MONORTM_OK;
}

int monortm_hip_kernel_time(void *ctx, int kernel, double *total_ms, long long *launches) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (kernel < 0 || kernel > 2) { c->err = "kernel id must be 0..2"; return MONORTM_EARG; }
    for (auto &e : c->events) {
        float ms = 0.f;
        HIPCHK(c, hipEventSynchronize(e.b));
        HIPCHK(c, hipEventElapsedTime(&ms, e.a, e.b));
        c->tot_ms[e.k] += ms;
        c->launches[e.k]++;
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    c->events.clear();
    *total_ms = c->tot_ms[kernel];
    *launches = c->launches[kernel];
    return MONORTM_OK;
}

int monortm_hip_check(void *ctx, void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->errflag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIPCHK(c, hipMemsetAsync(c->errflag, 0, sizeof(int), (hipStream_t)stream));
        if (flag & ERRBIT_TEMP) { c->err = "TIPS: layer temperature outside 70-3000 K / partition sum <= 0 (reference STOP, tips_2003.f90:277)"; return MONORTM_ETEMP; }
        c->err = "SDVOIGT: REAL(v) < 0 (reference STOP, modm.f90:1062)";
        return MONORTM_ESDV;
    }
    return MONORTM_OK;
}

int monortm_hip_modm_dev(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                         int nmol, const double *P, const double *T, const double *CLW, const double *WKL,
                         const double *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res,
                         int ibrd, int ixsect, double *O, double *O_BY_MOL, double *OC, double *O_CLW, void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    hipStream_t s = (hipStream_t)stream;
    // first / last wavenumber decide the ABSRB grid (modm.f90:180-185); they live in device memory
    double vends[2];
    HIPCHK(c, hipMemcpyAsync(&vends[0], wn, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&vends[1], wn + (nwn > 0 ? nwn - 1 : 0), sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    int rc = check_modm_args(c, nprof, nwn, nlay_max, nmol, ibrd, ixsect, vends[1]);
    if (rc) return rc;
    ModmArgs a{};
    a.nprof = nprof; a.nwn = nwn; a.nlay_max = nlay_max; a.nmol = nmol; a.ibrd = ibrd;
    a.dvset = dvset; a.sclcpl = sclcpl; a.sclhw = sclhw; a.y0res = y0res;
    for (int i = 0; i < 7; i++) a.cntnm[i] = cntnm_fac[i];
    a.wn = wn; a.P = P; a.T = T; a.CLW = CLW; a.WKL = WKL; a.WBRODL = WBRODL; a.nlay = nlay;
    a.O = O; a.O_BY_MOL = O_BY_MOL; a.OC = OC; a.O_CLW = O_CLW; a.errflag = c->errflag;

    const double DVABS = 1.0;
    const double V1ABS = (int)(vends[0]) - 3. * DVABS;
    const double V2ABS = (int)(vends[1] + 3. * DVABS + 0.5);
    const int NPTABS = (int)((V2ABS - V1ABS) / DVABS + 1.5);
    if (NPTABS > 5050) { c->err = "wavenumber span exceeds the 5050-point continuum grid (N_ABSRB, lblparams.f90:35)"; return MONORTM_EARG; }

    // few workgroups (single profiles): slice the line list over several blocks per (profile, layer, tile)
    const int NTw = (nwn <= 64) ? 64 : 256;
    const long long nblocks = (long long)((nwn + NTw - 1) / NTw) * nlay_max * nprof;
    const long long nlines = (long long)c->host.size();
    int nslice = 1;
    if (nblocks < 1024 && nlines >= 2 * NTw) {
        nslice = (int)std::min<long long>(16, std::min<long long>((2048 + nblocks - 1) / nblocks, nlines / NTw));
        if (nslice < 1) nslice = 1;
    }
    if (nslice > 1) {
        const size_t need = (size_t)nslice * nprof * nlay_max * nmol * nwn;
        if (need > c->partial_elems) {
            if (c->partial) HIPCHK(c, hipFree(c->partial));
            c->partial = nullptr;
            c->partial_elems = 0;
            HIPCHK(c, hipMalloc((void **)&c->partial, need * sizeof(double)));
            c->partial_elems = need;
        }
    }
    a.nslice = nslice;
    a.partial = c->partial;
    Ctx::Ev ev{};
    const bool use_brd = ibrd != 0 && c->host.any_brd;
    const size_t dyn = sizeof(double) * (size_t)(19 * nmol) + sizeof(int) * (size_t)(2 * nmol + 2);
    if (nwn <= 64) {
        dim3 grid(((nwn + 63) / 64) * nslice, nlay_max, nprof);
        prof_begin(c, s, 0, ev);
        if (use_brd) hipLaunchKernelGGL((lines_kernel<1, true>), grid, dim3(64), dyn, s, a, c->lines, c->tables);
        else hipLaunchKernelGGL((lines_kernel<1, false>), grid, dim3(64), dyn, s, a, c->lines, c->tables);
        prof_end(c, s, ev);
    } else {
        dim3 grid(((nwn + 255) / 256) * nslice, nlay_max, nprof);
        prof_begin(c, s, 0, ev);
        if (use_brd) hipLaunchKernelGGL((lines_kernel<4, true>), grid, dim3(256), dyn, s, a, c->lines, c->tables);
        else hipLaunchKernelGGL((lines_kernel<4, false>), grid, dim3(256), dyn, s, a, c->lines, c->tables);
        prof_end(c, s, ev);
    }
    HIPCHK(c, hipGetLastError());
    // finest coarse grid: 1 cm-1 (O2 A band) above 1340 cm-1, 2 cm-1 (CO2) below
    const int csize = (vends[1] > 1340.0 ? NPTABS : NPTABS / 2) + 24;
    const size_t lds = sizeof(double) * (size_t)(NPTABS + 4 + csize);
    prof_begin(c, s, 1, ev);
    const bool high = vends[1] > 1340.0;
    if (lds > 48 * 1024) {
        const void *fn = high ? reinterpret_cast<const void *>(finish_kernel<true>) : reinterpret_cast<const void *>(finish_kernel<false>);
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int fin_threads = (NPTABS <= 256 && nwn <= 128) ? 64 : 256;  // microwave-sized grids: one wave, cheap barriers
    prof_begin(c, s, 1, ev);
    if (high) hipLaunchKernelGGL(finish_kernel<true>, dim3(nlay_max, nprof), dim3(fin_threads), lds, s, a, c->tables, V1ABS, V2ABS, NPTABS, csize);
    else hipLaunchKernelGGL(finish_kernel<false>, dim3(nlay_max, nprof), dim3(fin_threads), lds, s, a, c->tables, V1ABS, V2ABS, NPTABS, csize);
    prof_end(c, s, ev);
    HIPCHK(c, hipGetLastError());
    return MONORTM_OK;
}

int monortm_hip_rtm_dev(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max, const int *irt,
                        int iout, const double *T, const double *TZ, const double *O, double *tmpsfc, const double *emiss,
                        const double *reflc, double *RUP, double *RDN, double *TRTOT, double *RAD, double *TB, double *TMR,
                        void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    hipStream_t s = (hipStream_t)stream;
    if (nprof < 1 || nwn < 1 || nlay_max < 1) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    RtmArgs a{};
    a.nprof = nprof; a.nwn = nwn; a.nlay_max = nlay_max; a.iout = iout;
    a.wn = wn; a.T = T; a.TZ = TZ; a.O = O; a.emiss = emiss; a.reflc = reflc; a.nlay = nlay; a.irt = irt;
    a.tmpsfc = tmpsfc; a.RUP = RUP; a.RDN = RDN; a.TRTOT = TRTOT; a.RAD = RAD; a.TB = TB; a.TMR = TMR;
    Ctx::Ev ev{};
    prof_begin(c, s, 2, ev);
    if (nlay_max >= 24) hipLaunchKernelGGL(rtm_kernel<8>, dim3((nwn + 63) / 64, nprof), dim3(64, 8), 0, s, a);
    else hipLaunchKernelGGL(rtm_kernel<2>, dim3((nwn + 63) / 64, nprof), dim3(64, 2), 0, s, a);
    prof_end(c, s, ev);
    HIPCHK(c, hipGetLastError());
    return MONORTM_OK;
}

// ---- host-buffer front ends (what the Fortran shim calls): stage through device memory -----------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

#define H2D(buf, src, bytes)                                                         \
    HIPCHK(c, hipMalloc(&(buf).p, std::max<size_t>((bytes), 8)));                    \
    if (src) HIPCHK(c, hipMemcpy((buf).p, (src), (bytes), hipMemcpyHostToDevice))

int monortm_hip_modm(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                     int nmol, const double *P, const double *T, const double *CLW, const double *WKL,
                     const double *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd,
                     int ixsect, double *O, double *O_BY_MOL, double *OC, double *O_CLW) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (nprof < 1 || nwn < 1 || nlay_max < 1 || nmol < 1) { c->err = "bad nprof/nwn/nlay_max/nmol"; return MONORTM_EARG; }
    for (int i = 1; i < nwn; i++)
        if (!(wn[i] >= wn[i - 1])) { c->err = "wavenumbers must be ascending (the reference takes v1 = wn(1), v2 = wn(nwn), modm.f90:180-181)"; return MONORTM_EARG; }
    for (int p = 0; p < nprof; p++)
        if (nlay[p] < 1 || nlay[p] > nlay_max) { c->err = "nlay[p] outside 1..nlay_max"; return MONORTM_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npl = (size_t)nprof * nlay_max, d = sizeof(double);
    DevBuf dwn, dnl, dP, dT, dC, dW, dB, dO, dOM, dOC, dOL;
    H2D(dwn, wn, nwn * d); H2D(dnl, nlay, nprof * sizeof(int));
    H2D(dP, P, npl * d); H2D(dT, T, npl * d); H2D(dC, CLW, npl * d); H2D(dW, WKL, npl * nmol * d); H2D(dB, WBRODL, npl * d);
    H2D(dO, (const void *)nullptr, npl * nwn * d); H2D(dOM, (const void *)nullptr, npl * nmol * nwn * d);
    H2D(dOC, (const void *)nullptr, npl * MONORTM_NCONT * nwn * d); H2D(dOL, (const void *)nullptr, npl * nwn * d);
    int rc = monortm_hip_modm_dev(ctx, nprof, nwn, (double *)dwn.p, dvset, (int *)dnl.p, nlay_max, nmol, (double *)dP.p,
                                  (double *)dT.p, (double *)dC.p, (double *)dW.p, (double *)dB.p, cntnm_fac, sclcpl, sclhw,
                                  y0res, ibrd, ixsect, (double *)dO.p, (double *)dOM.p, (double *)dOC.p, (double *)dOL.p,
                                  nullptr);
    if (rc) return rc;
    rc = monortm_hip_check(ctx, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(O, dO.p, npl * nwn * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(O_BY_MOL, dOM.p, npl * nmol * nwn * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(OC, dOC.p, npl * MONORTM_NCONT * nwn * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(O_CLW, dOL.p, npl * nwn * d, hipMemcpyDeviceToHost));
    return MONORTM_OK;
}

int monortm_hip_rtm(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max, const int *irt,
                    int iout, const double *T, const double *TZ, const double *O, double *tmpsfc, const double *emiss,
                    const double *reflc, double *RUP, double *RDN, double *TRTOT, double *RAD, double *TB, double *TMR) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (nprof < 1 || nwn < 1 || nlay_max < 1) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npl = (size_t)nprof * nlay_max, d = sizeof(double), pw = (size_t)nprof * nwn;
    DevBuf dwn, dnl, dirt, dT, dTZ, dO, dts, dem, drf, o1, o2, o3, o4, o5, o6;
    H2D(dwn, wn, nwn * d); H2D(dnl, nlay, nprof * sizeof(int)); H2D(dirt, irt, nprof * sizeof(int));
    H2D(dT, T, npl * d); H2D(dTZ, TZ, (size_t)nprof * (nlay_max + 1) * d); H2D(dO, O, npl * nwn * d);
    H2D(dts, tmpsfc, nprof * d); H2D(dem, emiss, pw * d); H2D(drf, reflc, pw * d);
    H2D(o1, (const void *)nullptr, pw * d); H2D(o2, (const void *)nullptr, pw * d); H2D(o3, (const void *)nullptr, pw * d);
    H2D(o4, (const void *)nullptr, pw * d); H2D(o5, (const void *)nullptr, pw * d); H2D(o6, (const void *)nullptr, pw * d);
    HIPCHK(c, hipMemset(o5.p, 0, pw * d));
    int rc = monortm_hip_rtm_dev(ctx, nprof, nwn, (double *)dwn.p, (int *)dnl.p, nlay_max, (int *)dirt.p, iout, (double *)dT.p,
                                 (double *)dTZ.p, (double *)dO.p, (double *)dts.p, (double *)dem.p, (double *)drf.p,
                                 (double *)o1.p, (double *)o2.p, (double *)o3.p, (double *)o4.p, (double *)o5.p,
                                 TMR ? (double *)o6.p : nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(RUP, o1.p, pw * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(RDN, o2.p, pw * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(TRTOT, o3.p, pw * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(RAD, o4.p, pw * d, hipMemcpyDeviceToHost));
    if (iout == 1) HIPCHK(c, hipMemcpy(TB, o5.p, pw * d, hipMemcpyDeviceToHost));
    if (TMR) HIPCHK(c, hipMemcpy(TMR, o6.p, pw * d, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(tmpsfc, dts.p, nprof * d, hipMemcpyDeviceToHost));
    return MONORTM_OK;
}

}  // extern "C"
