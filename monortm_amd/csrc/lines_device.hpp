// lines_device.hpp - device functions of the line-sum kernel (lines_kernel.hip): reciprocal, the Lorentz fast loops by
// line class, the general (coupled / Voigt) loop, the run dispatcher and the per-(layer, line) prepare stage.  Reference: src/modm.f90:277-440, :706-831.  See DESIGN.md section 3.1.
#pragma once
#include "lineshape.hpp"
#include "lines_asm.hpp"

namespace {
using namespace monortm_dev;

__device__ __forceinline__ unsigned long long uni64(unsigned long long x) {  // wave-uniform value -> SGPR pair
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// FP64 reciprocal: v_rcp_f64 seed (relative error 4.6e-8 measured on gfx950, tools/rcp_accuracy.hip) + one
// Newton step -> 2.2e-15.  Operands are positive normal numbers (d^2 + HWHM^2 and products of two of them),
// so no scaling / special cases are needed; an IEEE-correct division costs ~3x as many issue slots.
__device__ __forceinline__ double frcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32: 1 ulp
// two Newton steps = exact to 1 ulp (prepare stage: widths, S~ denominators)
__device__ __forceinline__ double frcp_any(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// Sum over the 64 lanes with DPP moves (no LDS traffic): butterflies inside each row of 16 lanes (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror), then row_bcast15 / row_bcast31 carry the row sums upwards; lane 63 holds the total, which
// is returned wave-uniform.  The order of the additions is fixed.
// lanes of rows that ROW_MASK leaves out receive `idle` (the identity of the reduction)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v, double idle = 0.) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int mlo = __builtin_amdgcn_update_dpp(__double2loint(idle), lo, CTRL, ROW_MASK, 0xf, false);
    const int mhi = __builtin_amdgcn_update_dpp(__double2hiint(idle), hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(mhi, mlo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);  // row_mirror
    v += dpp_move<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v) {
    const double inf = __builtin_inf();
    v = fmin(v, dpp_move<0xB1, 0xf>(v));
    v = fmin(v, dpp_move<0x4E, 0xf>(v));
    v = fmin(v, dpp_move<0x141, 0xf>(v));
    v = fmin(v, dpp_move<0x140, 0xf>(v));
    v = fmin(v, dpp_move<0x142, 0xa>(v, inf));
    v = fmin(v, dpp_move<0x143, 0xc>(v, inf));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// exp for the prepare stage (4 per line and layer): Cody-Waite reduction x = n ln2 + r, |r| <= 0.347, degree-13 Taylor
// polynomial (truncation 0.347^14 / 14! = 4e-18), ldexp.  ~20 instructions, about half the library call; agrees with
// it to 1-2 ulp.  Arguments here lie in (-1200, 140): underflow goes to 0 through ldexp, overflow cannot happen.
#ifdef MONORTM_EXP_SGPR_CONSTANTS
// one Horner step p r + c with the constant in a scalar register pair: left to itself the compiler puts every 64-bit constant into
// the destination of a v_fmac_f64 with two v_mov_b32 (three vector instructions a term), or - where its loop-invariant code motion
// runs - keeps thirteen register pairs alive across the whole chunk loop (lines_ms_kernel.hip: they do not fit beside the class
// loops' fixed registers).  The same operations in the same order: identical bits.
__device__ __forceinline__ double horner_s(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
    return o;
}
#else
__device__ __forceinline__ double horner_s(double p, double r, double c) { return fma(p, r, c); }
#endif
__device__ __forceinline__ double exp_prep(double x) {
    const double n = rint(x * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
#ifdef MONORTM_EXP_SGPR_CONSTANTS
    {
        double p = 1.0 / 6227020800.0;
        p = horner_s(p, r, 1.0 / 479001600.0);
        p = horner_s(p, r, 1.0 / 39916800.0);
        p = horner_s(p, r, 1.0 / 3628800.0);
        p = horner_s(p, r, 1.0 / 362880.0);
        p = horner_s(p, r, 1.0 / 40320.0);
        p = horner_s(p, r, 1.0 / 5040.0);
        p = horner_s(p, r, 1.0 / 720.0);
        p = horner_s(p, r, 1.0 / 120.0);
        p = horner_s(p, r, 1.0 / 24.0);
        p = horner_s(p, r, 1.0 / 6.0);
        p = fma(p, r, 0.5);
        p = fma(p, r, 1.0);
        p = fma(p, r, 1.0);
        return ldexp(p, (int)n);
    }
#endif
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

#ifdef MONORTM_EXP_SGPR_CONSTANTS
// Four exponentials side by side (the four of line_physics_core: INTENS' Boltzmann factor, the two radiation-field factors of S~ and
// the temperature exponent of the width): one Horner step of ALL FOUR per scalar constant - the constant is materialised once
// (two s_mov_b32) instead of four times, and the four chains are independent (FP64 latency covered inside the wave).  A wave pays for
// scalar instructions as for vector ones (one issue per wave and turn): the prepare stage of lines_ms_kernel ran at VALU busy 0.65
// with 0.6 scalar instructions per vector one.  Per chain the operations of exp_prep in the same order: identical bits.
__device__ __forceinline__ void horner4_s(double (&p)[4], const double (&r)[4], double c) {
    asm("v_fma_f64 %0, %0, %4, %8\n\tv_fma_f64 %1, %1, %5, %8\n\tv_fma_f64 %2, %2, %6, %8\n\tv_fma_f64 %3, %3, %7, %8"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "s"(c));
}
__device__ __forceinline__ void exp_prep4(const double (&x)[4], double (&out)[4]) {
    double n[4], r[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        n[i] = rint(x[i] * 1.44269504088896338700e+00);
        r[i] = fma(-n[i], 6.93147180369123816490e-01, x[i]);
        r[i] = fma(-n[i], 1.90821492927058770002e-10, r[i]);
        p[i] = 1.0 / 6227020800.0;
    }
    horner4_s(p, r, 1.0 / 479001600.0);
    horner4_s(p, r, 1.0 / 39916800.0);
    horner4_s(p, r, 1.0 / 3628800.0);
    horner4_s(p, r, 1.0 / 362880.0);
    horner4_s(p, r, 1.0 / 40320.0);
    horner4_s(p, r, 1.0 / 5040.0);
    horner4_s(p, r, 1.0 / 720.0);
    horner4_s(p, r, 1.0 / 120.0);
    horner4_s(p, r, 1.0 / 24.0);
    horner4_s(p, r, 1.0 / 6.0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        p[i] = fma(p[i], r[i], 0.5);
        p[i] = fma(p[i], r[i], 1.0);
        p[i] = fma(p[i], r[i], 1.0);
        out[i] = ldexp(p[i], (int)n[i]);
    }
}
#endif

// tanh for the radiation term RFT = WN tanh(hc WN / 2kT) (modm.f90:436-438), argument >= 0: the library call is ~225
// instructions per wavenumber of a lane (2 % of a c4shard workgroup).  Below 1/8 (every microwave channel) the odd Taylor
// series up to x^13 (next term 1.5e-3 x^14 <= 3e-16 relative); above, (1 - e) / (1 + e) with e = exp(-2x) <= 0.78, no
// cancellation.  Agrees with the library to 1-2 ulp.
__device__ __forceinline__ double tanh_pos(double x) {
    if (x < 0.125) {
        const double z = x * x;
        double p = 21844.0 / 6081075.0;
        p = fma(p, z, -1382.0 / 155925.0);
        p = fma(p, z, 62.0 / 2835.0);
        p = fma(p, z, -17.0 / 315.0);
        p = fma(p, z, 2.0 / 15.0);
        p = fma(p, z, -1.0 / 3.0);
        return fma(x * z, p, x);
    }
    const double e = exp_prep(-2.0 * x);
    return (1.0 - e) / (1.0 + e);
}

// The Lorentz shapes of src/modm.f90:706-831, regrouped.  With a2 = S~ HWHM/pi and hw2 = HWHM^2:
//     S~ * XLORENTZ(d/HWHM)/HWHM = a2 / (d^2 + hw2)
// so one evaluation is (d, d^2+hw2, one reciprocal, one FMA for the pedestal); two resonances share a
// single reciprocal:  a2*(Y1*den2 + Y2*den1)/(den1*den2).
//   KIND : 0 generic molecule, 1 O2 (no pedestal; coupled lines exempt from the 25 cm-1 rule), 2 CO2
//          (pedestal x (2 - d^2/625), no negative resonance)

// ---- fast path: molecule run without coupled lines and without any Voigt candidate in this chunk ----------
// A run is cut into sub-runs of lines of one class (bit masks built by the prepare stage):
//   M2   : the negative resonance of the line is within reach (WN + Xnu <= 25) of some wavenumber of the tile
//   TEST : the 25 cm-1 test can fail for some wavenumber of the tile (otherwise every lane is live: no compare / select)
// a b - c clamped to [0, 1] in ONE instruction (VOP3 clamp).  For a generic molecule a bracket a2 / den - pedestal is >= 0
// exactly where its rule admits the line (|WN - Xnu| <= 25: modm.f90:384; WN + Xnu <= 25: :713), so the clamp IS the test; the
// upper bound is out of reach (a2 / den <= S~ / (pi HW) < 1e-12).  See lines_asm.hpp, LA_FIN_T_K0.
__device__ __forceinline__ double fma_clamp0(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, -%3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ double eval_one_fast(const HotA h, const double pb_or_lim, double WN) {
    const double d = WN - h.xnu;
    const double den1 = fma(d, d, h.hw2);
    const double cutlim = (KIND == 1) ? h.pa : 25.;
    const bool live = !(fabs(d) > cutlim);  // modm.f90:384 (O2: inside the shape function, :755)
    double term;
    if constexpr (KIND == 0 && !M2 && TEST) return fma_clamp0(h.a2, frcp(den1), h.pa);
    if constexpr (KIND == 0 && M2) {  // both brackets over one reciprocal, each clamped: tested or not, within reach or not
        const double dp = WN + h.xnu;
        const double den2 = fma(dp, dp, h.hw2);
        const double q = h.a2 * frcp(den1 * den2);
        return fma_clamp0(q, den2, h.pa) + fma_clamp0(q, den1, pb_or_lim);
    }
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
    return (!TEST || live) ? term : 0.;
}

template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ double eval_fast(const HotA *sA, const HotB *sB, int j0, int j1, double WN, double SF) {
    // two lines per trip, LDS records fetched one line ahead (ping-pong registers, no copies)
    constexpr bool needB = M2 && KIND != 2;
    if (j0 >= j1) return SF;
    HotA h0 = sA[j0];
    double b0 = needB ? sB[j0].pb : 0.;
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const HotA h1 = sA[j + 1];
        const double b1 = needB ? sB[j + 1].pb : 0.;
        SF += eval_one_fast<KIND, M2, TEST>(h0, b0, WN);
        const int jn = (j + 2 < j1) ? j + 2 : j + 1;
        h0 = sA[jn];
        if (needB) b0 = sB[jn].pb;
        SF += eval_one_fast<KIND, M2, TEST>(h1, b1, WN);
    }
    if (j < j1) SF += eval_one_fast<KIND, M2, TEST>(h0, b0, WN);
    return SF;
}

// Sub-runs with one resonance and no test (generic molecules and uncoupled O2): two LINES share one reciprocal,
//   a2_0/den_0 + a2_1/den_1 = (a2_0 den_1 + a2_1 den_0) / (den_0 den_1)
template <int KIND>
__device__ __forceinline__ double eval_pair_fast(const HotA *sA, int j0, int j1, double WN, double SF) {
    if (j0 >= j1) return SF;
    HotA h0 = sA[j0];
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const HotA h1 = sA[j + 1];
        const HotA hn = sA[(j + 2 < j1) ? j + 2 : j + 1];
        const double d0 = WN - h0.xnu, d1 = WN - h1.xnu;
        const double den0 = fma(d0, d0, h0.hw2), den1 = fma(d1, d1, h1.hw2);
        const double num = fma(h0.a2, den1, h1.a2 * den0);
        const double r = frcp(den0 * den1);
        if (KIND == 0) SF += fma(num, r, -(h0.pa + h1.pa));
        else SF += num * r;
        h0 = hn;
    }
    if (j < j1) SF += eval_one_fast<KIND, false, false>(h0, 0., WN);
    return SF;
}

// The same for the other classes of a sub-run (two resonances and / or the 25 cm-1 test): the two lines' denominators
// P_i (= den1_i, or den1_i den2_i with both resonances) share one reciprocal - v_rcp_f64 plus its Newton step cost as
// much as six FMAs, the three extra products half of that:
//   n_0 / P_0 + n_1 / P_1 = (n_0 P_1) r + (n_1 P_0) r,   r = 1 / (P_0 P_1)
// The terms are added to SF one after the other, in line order, each with its own test.
template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ double eval_pair(const HotA *sA, const HotB *sB, int j0, int j1, double WN, double SF) {
    static_assert(KIND != 2, "CO2 keeps eval_fast");
    if (j0 >= j1) return SF;
    HotA h0 = sA[j0];
    double b0 = M2 ? sB[j0].pb : 0.;
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const HotA h1 = sA[j + 1];
        const double b1 = M2 ? sB[j + 1].pb : 0.;
        const int jn = (j + 2 < j1) ? j + 2 : j + 1;
        const HotA hn = sA[jn];
        const double bn = M2 ? sB[jn].pb : 0.;
        const double d0 = WN - h0.xnu, d1 = WN - h1.xnu;
        const double den0 = fma(d0, d0, h0.hw2), den1 = fma(d1, d1, h1.hw2);
        double n0 = h0.a2, n1 = h1.a2, P0 = den0, P1 = den1, ped0 = h0.pa, ped1 = h1.pa;
        if (M2) {
            const double dp0 = WN + h0.xnu, dp1 = WN + h1.xnu;
            const double m0 = (dp0 <= ((KIND == 1) ? b0 : 25.)) ? 1.0 : 0.0, m1 = (dp1 <= ((KIND == 1) ? b1 : 25.)) ? 1.0 : 0.0;
            const double e0 = fma(dp0, dp0, h0.hw2), e1 = fma(dp1, dp1, h1.hw2);
            n0 *= fma(m0, den0, e0);
            n1 *= fma(m1, den1, e1);
            P0 *= e0;
            P1 *= e1;
            if (KIND == 0) {
                ped0 = fma(m0, b0, ped0);
                ped1 = fma(m1, b1, ped1);
            }
        }
        const double r = frcp(P0 * P1);
        const double t0 = (KIND == 0) ? fma(n0 * P1, r, -ped0) : (n0 * P1) * r;
        const double t1 = (KIND == 0) ? fma(n1 * P0, r, -ped1) : (n1 * P0) * r;
        if (TEST) {
            // modm.f90:384 (O2: inside the shape function, :755) as a 0/1 factor inside the accumulation: fma(t, 1, SF) is the
            // rounded sum, fma(t, 0, SF) is SF (t is finite) - one select of a half register instead of two selects and an add
            const double l0 = !(fabs(d0) > ((KIND == 1) ? h0.pa : 25.)) ? 1.0 : 0.0;
            const double l1 = !(fabs(d1) > ((KIND == 1) ? h1.pa : 25.)) ? 1.0 : 0.0;
            SF = fma(t0, l0, SF);
            SF = fma(t1, l1, SF);
        } else {
            SF += t0;
            SF += t1;
        }
        h0 = hn;
        b0 = bn;
    }
    if (j < j1) SF += eval_one_fast<KIND, M2, TEST>(h0, b0, WN);
    return SF;
}

// ---- round 4: the ordinary lines of a run in class steps written in assembly (lines_asm.hpp; double precision, one
// wavenumber per lane).  The C++ class loops above restart at every class change with copies of their own, carry 2-3 v_mov of
// wave-uniform LDS addresses per two lines and, in the tested classes, v_cmp + v_cndmask + FMA per line for the 25 cm-1 rule.
// The class steps keep ONE address register, read every record at an immediate offset, hold two pairs in flight in fixed
// alternating register sets and apply the per-lane conditions as EXEC masks (v_cmpx).  What is left here: the dispatcher and
// the odd line at the end of a run.
typedef const __attribute__((address_space(3))) double *lds_cdp;
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}

// SF += t0 for the lanes with !(|d0| > l0) (modm.f90:384; O2: inside the shape function, :755)
template <bool LIM_UNIFORM>
__device__ __forceinline__ void acc1_tested(double &SF, double d0, double l0, double t0) {
    unsigned long long sv, cm;
    if constexpr (LIM_UNIFORM)
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_ngt_f64_e64 %[cm], |%[d0]|, %[l0]\n\t"
                     "v_add_f64 %[sf], %[sf], %[t0]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sf] "+v"(SF), [sv] "=&s"(sv), [cm] "=&s"(cm)
                     : [d0] "v"(d0), [l0] "s"(l0), [t0] "v"(t0));
    else
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_ngt_f64_e64 %[cm], |%[d0]|, %[l0]\n\t"
                     "v_add_f64 %[sf], %[sf], %[t0]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sf] "+v"(SF), [sv] "=&s"(sv), [cm] "=&s"(cm)
                     : [d0] "v"(d0), [l0] "v"(l0), [t0] "v"(t0));
}
// negative resonance within reach (DIFF = (WN + Xnu) - 25 <= 0, modm.f90:713; O2: the limit of the record): for those lanes
//   e = den2 becomes den2 + den1   (numerator of 1/den1 + 1/den2 over the common denominator den1 den2)
//   pa becomes pa + pb             (generic molecules: the second pedestal)
template <int KIND>
__device__ __forceinline__ void m2_lanes(double dp, double lim, double den, double &e, double &pa, double pb) {
    unsigned long long sv, cm;
    if constexpr (KIND == 0)
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_le_f64_e64 %[cm], %[dp], %[lim]\n\t"
                     "v_add_f64 %[e], %[e], %[den]\n\t"
                     "v_add_f64 %[pa], %[pa], %[pb]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [e] "+v"(e), [pa] "+v"(pa), [sv] "=&s"(sv), [cm] "=&s"(cm)
                     : [dp] "v"(dp), [lim] "s"(lim), [den] "v"(den), [pb] "v"(pb));
    else
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_le_f64_e64 %[cm], %[dp], %[lim]\n\t"
                     "v_add_f64 %[e], %[e], %[den]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [e] "+v"(e), [sv] "=&s"(sv), [cm] "=&s"(cm)
                     : [dp] "v"(dp), [lim] "v"(lim), [den] "v"(den));
}
// the odd line at the end of a run: the arithmetic of a class step's pair for one line
template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ double uni_single(HotA h0, double b0, double WN, double SF) {
    const double d0 = WN - h0.xnu;
    const double den0 = fma(d0, d0, h0.hw2);
    if constexpr (KIND == 0 && (M2 || TEST)) {  // generic molecule: the clamped brackets of the pair loops (lines_asm.hpp)
        if constexpr (!M2) return SF + fma_clamp0(h0.a2, frcp(den0), h0.pa);
        const double dp0 = WN + h0.xnu;
        const double e0 = fma(dp0, dp0, h0.hw2);
        const double q = h0.a2 * frcp(den0 * e0);
        SF += fma_clamp0(q, e0, h0.pa);
        return SF + fma_clamp0(q, den0, h0.pa);
    }
    double n0 = h0.a2, P0 = den0, ped0 = h0.pa;
    if constexpr (M2) {
        const double dp0 = WN + h0.xnu;
        double e0 = fma(dp0, dp0, h0.hw2);
        P0 *= e0;
        m2_lanes<KIND>(dp0, (KIND == 1) ? b0 : 25., den0, e0, ped0, b0);
        n0 *= e0;
    }
    const double t0 = (KIND == 0) ? fma(n0, frcp(P0), -ped0) : n0 * frcp(P0);
    if constexpr (TEST) acc1_tested<KIND == 0>(SF, d0, (KIND == 1) ? h0.pa : 25., t0);  // (O2: the limit sits in the pa slot)
    else SF += t0;
    return SF;
}
template <int KIND>
__device__ __forceinline__ double uni_single_any(unsigned cls, HotA h0, double b0, double WN, double SF) {
    if constexpr (KIND == 2) {  // CO2: one resonance, the pedestal x (2 - d^2 / 625)
        if (cls & 1u) return SF + eval_one_fast<2, false, true>(h0, 0., WN);
        return SF + eval_one_fast<2, false, false>(h0, 0., WN);
    }
    if (cls & 2u) {
        if (cls & 1u) return uni_single<KIND, true, true>(h0, b0, WN, SF);
        return uni_single<KIND, true, false>(h0, b0, WN, SF);
    }
    if (cls & 1u) return uni_single<KIND, false, true>(h0, b0, WN, SF);
    return uni_single<KIND, false, false>(h0, b0, WN, SF);
}
// class of the pair at bits 0, 1 of the shifted masks: bit 0 = tested, bit 1 = two resonances (scalar)
__device__ __forceinline__ unsigned pair_cls(unsigned long long T, unsigned long long M) {
    return (unsigned)((T & 3ull) != 0ull) | ((unsigned)((M & 3ull) != 0ull) << 1);
}

// sA: the chunk's HotA records; the HotB records sit BOFF bytes behind them in the same LDS object, both arrays padded by two
// entries (the read-ahead of a class step may run two records past the run).  Lines j .. j + n - 1 of one 64-line group;
// T / M: its "tested" and "two resonances" masks shifted so that bit 0 belongs to line j.  A pair takes the class step of the
// more general of its two lines (the per-lane conditions decide); lines are added in file order.
template <int KIND, unsigned BOFF>
__device__ __forceinline__ double eval_unified(const HotA *sA, int j, int n, unsigned long long T, unsigned long long M, double WN,
                                               double SF) {
    unsigned addr = lds_addr(sA + j);
    // wave-uniform by construction; the class steps take them in scalar registers
    n = __builtin_amdgcn_readfirstlane(n);
    T = uni64(T);
    M = uni64(M);
    if (n >= 2) asm_run<KIND, BOFF>(addr, n, T, M, WN, SF);
    if (n == 1) {
        const unsigned cls = (unsigned)(T & 1ull) | ((unsigned)(M & 1ull) << 1);
        const lds_cdp q = (lds_cdp)addr;  // (field by field: a struct copy out of address space 3 needs a generic reference)
        const HotA h{q[0], q[1], q[2], q[3]};
        const double b = (cls & 2u) ? *(lds_cdp)(addr + BOFF) : 0.;
        SF = uni_single_any<KIND>(cls, h, b, WN, SF);
    }
    return SF;
}

// ---- the same fast path in single precision: d = WN - Xnu from the float pairs of centre and wavenumber (lineshape.hpp:
// the accuracy of the reference's double difference rounded to float), everything after it in float; pedestal / limit of
// the negative resonance sit in the same 24-byte record
__device__ __forceinline__ float pair_hi(double x) { return (float)x; }
__device__ __forceinline__ float pair_lo(double x) { return (float)(x - (double)(float)x); }
template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ float eval_one_fast(const HotAf h, double WN) {
    const float wh = pair_hi(WN), wl = pair_lo(WN);  // (loop-invariant: formed once per lane)
    const float d = (wh - h.xh) + (wl - h.xl);
    const float den1 = fmaf(d, d, h.hw2);
    const float cutlim = (KIND == 1) ? h.pa : 25.f;
    const bool live = !(fabsf(d) > cutlim);
    float term;
    if (KIND == 2) {
        const float f = fmaf(-(d * d), 1.0f / 625.f, 2.f);
        term = fmaf(-h.pa, f, h.a2 * frcp(den1));
    } else if (!M2) {
        term = (KIND == 0) ? fmaf(h.a2, frcp(den1), -h.pa) : h.a2 * frcp(den1);
        if constexpr (KIND == 0 && TEST) return fmaxf(term, 0.f);  // (>= 0 exactly inside the 25 cm-1 window: fma_clamp0)
    } else if constexpr (KIND == 0) {  // both brackets over one reciprocal, each >= 0 exactly where its rule admits it
        const float dp = (wh + h.xh) + (wl + h.xl);
        const float den2 = fmaf(dp, dp, h.hw2);
        const float q = h.a2 * frcp(den1 * den2);
        return fmaxf(fmaf(q, den2, -h.pa), 0.f) + fmaxf(fmaf(q, den1, -h.pb), 0.f);
    } else {
        const float dp = (wh + h.xh) + (wl + h.xl);
        const float m2f = (dp <= ((KIND == 1) ? h.pb : 25.f)) ? 1.0f : 0.0f;
        const float den2 = fmaf(dp, dp, h.hw2);
        const float num = fmaf(m2f, den1, den2);
        const float t = h.a2 * num;
        if (KIND == 0) term = fmaf(t, frcp(den1 * den2), -fmaf(m2f, h.pb, h.pa));
        else term = t * frcp(den1 * den2);
    }
    return (!TEST || live) ? term : 0.f;
}

template <int KIND, bool M2, bool TEST>
__device__ __forceinline__ float eval_fast(const HotAf *sA, const HotB *, int j0, int j1, double WN, float SF) {
    if (j0 >= j1) return SF;
    HotAf h0 = sA[j0];
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const HotAf h1 = sA[j + 1];
        SF += eval_one_fast<KIND, M2, TEST>(h0, WN);
        h0 = sA[(j + 2 < j1) ? j + 2 : j + 1];
        SF += eval_one_fast<KIND, M2, TEST>(h1, WN);
    }
    if (j < j1) SF += eval_one_fast<KIND, M2, TEST>(h0, WN);
    return SF;
}

// ---- single precision, two wavenumbers per lane: the pair is one float2 and the arithmetic packed (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per instruction; the reciprocals and the selects stay scalar)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { return (f2){x, x}; }

// FULL (round 4; two resonances, untested): EVERY wavenumber of the tile is within reach of the negative resonance (WN + Xnu <=
// limit at the tile's upper end - sounder channels below 6.5 cm-1 against lines below 18.5).  No lane needs the 0/1 factor
// then, and with t = d d+ + HW^2 (d+ = WN + Xnu, d+ - d = 2 Xnu) both resonances are two FMAs:
//   den1 + den2 = 2 t + g,   den1 den2 = t^2 + HW^2 g,   g = (2 Xnu)^2
// g, HW^2 g and the sum of the two pedestals are formed per line from the ordinary 24-byte record (four scalar instructions:
// a line whose short FULL run was smoothed away must still fit the generic loop; parking them in the HotB record instead cost
// an LDS read that was slower than the four instructions, configs[4] whole 1.177 against 1.202 ms).  d+ is formed from the
// high parts alone (a sum of positives: one float rounding, like the generic loop's three-term sum).  17 vector instructions
// per line and pair of wavenumbers instead of 22.5 (profiles/r04_isa_census_f14).
template <int KIND, bool M2, bool TEST, bool FULL = false>
__device__ __forceinline__ f2 eval_one_fast2(const HotAf h, const double (&WN)[2]) {
    const f2 wh = {pair_hi(WN[0]), pair_hi(WN[1])}, wl = {pair_lo(WN[0]), pair_lo(WN[1])};  // (loop-invariant)
    const f2 d = (wh - splat(h.xh)) + (wl - splat(h.xl));
    const f2 hw2 = splat(h.hw2), a2 = splat(h.a2);
    if constexpr (FULL && M2 && !TEST && KIND != 2) {
        const f2 dp = wh + splat(h.xh);
        const float x2 = h.xh + h.xh, g = x2 * x2;
        const f2 t = pk_fma(d, dp, hw2);
        const f2 pr = pk_fma(t, t, splat(h.hw2 * g));
        const f2 n = a2 * pk_fma(splat(2.f), t, splat(g));
        const f2 r = {frcp(pr.x), frcp(pr.y)};
        return (KIND == 0) ? pk_fma(n, r, splat(-(h.pa + h.pb))) : n * r;
    }
    const f2 den1 = pk_fma(d, d, hw2);
    const float cutlim = (KIND == 1) ? h.pa : 25.f;
    f2 term;
    if (KIND == 2) {
        const f2 f = pk_fma(-(d * d), splat(1.0f / 625.f), splat(2.f));
        const f2 r = {frcp(den1.x), frcp(den1.y)};
        term = pk_fma(splat(-h.pa), f, a2 * r);
    } else if (!M2) {
        const f2 r = {frcp(den1.x), frcp(den1.y)};
        term = (KIND == 0) ? pk_fma(a2, r, splat(-h.pa)) : a2 * r;
        if constexpr (KIND == 0 && TEST) return (f2){fmaxf(term.x, 0.f), fmaxf(term.y, 0.f)};  // (the 25 cm-1 test: fma_clamp0)
    } else if constexpr (KIND == 0) {  // generic molecule, two resonances, tested or not: the clamped brackets of lines_asm.hpp
        const f2 dp = (wh + splat(h.xh)) + (wl + splat(h.xl));
        const f2 den2 = pk_fma(dp, dp, hw2);
        const f2 pr = den1 * den2;
        const f2 q = a2 * (f2){frcp(pr.x), frcp(pr.y)};
        const f2 b1 = pk_fma(q, den2, splat(-h.pa)), b2 = pk_fma(q, den1, splat(-h.pb));
        return (f2){fmaxf(b1.x, 0.f), fmaxf(b1.y, 0.f)} + (f2){fmaxf(b2.x, 0.f), fmaxf(b2.y, 0.f)};
    } else {
        const f2 dp = (wh + splat(h.xh)) + (wl + splat(h.xl));
        const float lim = (KIND == 1) ? h.pb : 25.f;
        const f2 m2f = {(dp.x <= lim) ? 1.0f : 0.0f, (dp.y <= lim) ? 1.0f : 0.0f};
        const f2 den2 = pk_fma(dp, dp, hw2);
        const f2 t = a2 * pk_fma(m2f, den1, den2);
        const f2 pr = den1 * den2;
        const f2 r = {frcp(pr.x), frcp(pr.y)};
        if (KIND == 0) term = pk_fma(t, r, -pk_fma(m2f, splat(h.pb), splat(h.pa)));
        else term = t * r;
    }
    if (TEST) {
        term.x = !(fabsf(d.x) > cutlim) ? term.x : 0.f;
        term.y = !(fabsf(d.y) > cutlim) ? term.y : 0.f;
    }
    return term;
}

__device__ __forceinline__ HotA widen(const HotA &h) { return h; }
__device__ __forceinline__ HotA widen(const HotAf &h) { return HotA{(double)h.xh + (double)h.xl, (double)h.hw2, (double)h.a2, (double)h.pa}; }
__device__ __forceinline__ double rec_xnu(const HotA &h) { return h.xnu; }
__device__ __forceinline__ double rec_xnu(const HotAf &h) { return (double)h.xh + (double)h.xl; }

// ---- general path: coupled lines (Y factors) and / or Voigt candidates ------------------------------------
// VOIGT: the (wavenumber, line) pairs that take a (speed-dependent) Voigt shape are rare and scattered - in a sub-run of Voigt
// candidates one or two lanes of a wave lie within 100 Doppler widths of a centre - while the shape costs thousands of
// instructions in several Humlicek regions.  Evaluated in place, every such line made the whole wave walk that code for one
// lane (measured: 59 % of the evaluate time of c3, 18 % of c4shard's).  The lanes therefore only QUEUE their pair (line, lane)
// in LDS (vq, 64 entries per wave and wavenumber of the lane) and take the Lorentz term of everybody else; the queue is
// worked off one pair per lane - dense - when it is full and when the walk over the molecule's lines of the chunk ends
// (eval_dispatch), and each value is handed to its wavenumber's lane in queue order (fixed order: deterministic; the Voigt
// terms join the sum after the Lorentz terms).
// rec_off: lines_packed_kernel only (lanes of several layers in one wave: sA / sB / sCold arrive as the lane's own layer's
// records, rec_off is their offset from the first layer's) - the records and the amplitude scale of a queued pair are its
// OWNER's; zero (and a wave-uniform wscale) everywhere else.
template <int KIND, bool PACKED = false, typename H>
__device__ __forceinline__ double voigt_flush(const H *sA, const HotB *sB, const ColdLine *sCold, const unsigned short *vq, int n,
                                              double WN, int mol, double SF, double wscale, int *errflag, int rec_off = 0) {
    const int lane = (int)__lane_id();
    const unsigned rec = (lane < n) ? vq[lane] : 0u;
    const int j = (int)(rec >> 6), owner = (int)(rec & 63u);
    const double WNi = __shfl(WN, owner);
    int ro = 0;
    if constexpr (PACKED) {
        ro = __shfl(rec_off, owner) - rec_off;  // from this lane's layer to the owner's
        wscale = __shfl(wscale, owner);
    }
    double val = 0.;
    if (lane < n) {
        const HotA h = widen(sA[ro + j]);
        const HotB b = sB[ro + j];
        const ColdLine c = sCold[ro + j];
        // the shape functions only use the products AIP*(1/HW)*RP = c1 and BIP*RP2 = gp1-1:
        // hand them over as AIP' = c1*HW, BIP' = gp1-1 with RP' = RP2' = 1
        const double SLS = lsf_sdvoigt(mol, (int)((c.info >> 6) & 3), 1.0, 1.0, b.c1 * c.hw, b.gp1 - 1., c.hw, WNi, h.xnu, c.hwd,
                                       (double)c.sdep, cold_xl3(c, mol, errflag), errflag);
        val = (c.stild * wscale) * SLS;
    }
    for (int it = 0; it < n; it++) {  // wave-uniform trip count and indices
        const int lo = __builtin_amdgcn_readlane(__double2loint(val), it), hi = __builtin_amdgcn_readlane(__double2hiint(val), it);
        const int ow = __builtin_amdgcn_readlane((int)rec, it) & 63;
        if (lane == ow) SF += __hiloint2double(hi, lo);
    }
    return SF;
}

// the Lorentz term of ONE line with Y factors for one wavenumber, with every per-lane condition explicit: the arithmetic of
// eval_general's loop body (its wave-level shortcut for "no lane has the negative resonance" differs in rounding only)
template <int KIND>
__device__ __forceinline__ double general_term(const HotA &h, const HotB &b, double WN) {
    const double d = WN - h.xnu, dp = WN + h.xnu;
    const double ad = fabs(d);
    const double den1 = fma(d, d, h.hw2);
    const double Y1 = fma(b.c1, d, b.gp1);
    if constexpr (KIND == 2) {
        const double f = fma(-(d * d), 1.0 / 625., 2.);
        const double term = Y1 * fma(-h.pa, f, h.a2 * frcp(den1));
        return !(ad > 25.) ? term : 0.;
    } else {
        const double cutlim = (KIND == 1) ? h.pa : 25.;
        const double dplim = (KIND == 1) ? b.pb : 25.;
        const bool m2 = dp <= dplim;
        const double den2 = m2 ? fma(dp, dp, h.hw2) : 1.0;
        const double Y2 = m2 ? fma(-b.c1, dp, b.gp1) : 0.0;
        double term = (h.a2 * fma(Y1, den2, Y2 * den1)) * frcp(den1 * den2);
        if (KIND == 0) term -= (m2 ? h.pa + b.pb : h.pa);
        return !(ad > cutlim) ? term : 0.;
    }
}

template <int KIND, bool VOIGT, bool PACKED = false, typename H>
__device__ __forceinline__ double eval_general(const H *sA, const HotB *sB, const ColdLine *sCold, int j0, int j1, double WN,
                                               int mol, double SF, double wscale, int *errflag, unsigned short *vq, int &nq,
                                               int rec_off = 0) {
    HotA h = widen(sA[j0]);
    HotB b = sB[j0];
    for (int j = j0; j < j1; j++) {  // nq: Voigt pairs queued so far (wave-uniform; the caller works off the rest)
        const int jn = (j + 1 < j1) ? j + 1 : j;
        const HotA hnext = widen(sA[jn]);  // software prefetch of the next line's LDS records
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
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(useV);
            if (mv != 0ull) {
                const int add = __popcll(mv);
                if (nq + add > 64) {  // (a line queues at most 64 pairs)
                    SF = voigt_flush<KIND, PACKED>(sA, sB, sCold, vq, nq, WN, mol, SF, wscale, errflag, rec_off);
                    nq = 0;
                }
                const int lane = (int)__lane_id();
                if (useV) {
                    vq[nq + __popcll(mv & ((1ull << lane) - 1ull))] = (unsigned short)((j << 6) | lane);
                    term = 0.;  // the Voigt value replaces the Lorentz term (modm.f90:427-432)
                }
                nq += add;
            }
        }
        SF += live ? term : 0.;
        h = hnext;
        b = bnext;
    }
    return SF;
}

// ---- first-order coupled O2 lines (XG = -1; the 60 GHz complex): Lorentz shapes with Y factors, both resonances for
// every wavenumber, no cut and no pedestal (modm.f90:755-776).  One reciprocal serves both resonances:
//   a2 (Y1/den1 + Y2/den2) = a2 (Y1 den2 + Y2 den1) / (den1 den2),  Y1 = (1+g) + c1 (WN-Xnu),  Y2 = (1+g) - c1 (WN+Xnu)
template <typename H>
__device__ __forceinline__ double eval_o2_coupled(const H *sA, const HotB *sB, int j0, int j1, double WN, double SF) {
    HotA h = widen(sA[j0]);
    HotB b = sB[j0];
    for (int j = j0; j < j1; j++) {
        const int jn = (j + 1 < j1) ? j + 1 : j;
        const HotA hn = widen(sA[jn]);
        const HotB bn = sB[jn];
        const double d = WN - h.xnu, dp = WN + h.xnu;
        const double den1 = fma(d, d, h.hw2), den2 = fma(dp, dp, h.hw2);
        const double Y1 = fma(b.c1, d, b.gp1), Y2 = fma(-b.c1, dp, b.gp1);
        const double num = fma(Y1, den2, Y2 * den1);
        SF += (h.a2 * num) * frcp(den1 * den2);
        h = hn;
        b = bn;
    }
    return SF;
}

// ---- two wavenumbers per lane (tiles of 2 x NW x 64): one LDS record read serves two evaluations, the prepare stage is
// paid once per two wavenumber tiles.  For one-resonance untested lines the two wavenumbers share the reciprocal:
//   q = a2 / (den_a den_b);  a2/den_a = q den_b,  a2/den_b = q den_a
// LUMP: the pedestals of the sub-run are subtracted once, after the loop (eval_fast2)
template <int KIND, bool M2, bool TEST, bool LUMP, bool FULL = false, typename R, typename H>
__device__ __forceinline__ void eval_one2(const H &h, double b, const double (&WN)[2], R (&SF)[2]) {
    if constexpr (sizeof(R) == 8) {
        if constexpr (!M2 && KIND != 2) {
            const double da = WN[0] - h.xnu, db = WN[1] - h.xnu;
            const double dena = fma(da, da, h.hw2), denb = fma(db, db, h.hw2);
            const double q = h.a2 * frcp(dena * denb);
            if constexpr (TEST && KIND == 0) {  // the two wavenumbers share the reciprocal, the clamp is each one's 25 cm-1 test
                SF[0] += fma_clamp0(q, denb, h.pa);
                SF[1] += fma_clamp0(q, dena, h.pa);
            } else if constexpr (TEST) {
                const double cutlim = (KIND == 1) ? h.pa : 25.;
                const double ta = (KIND == 0) ? fma(q, denb, -h.pa) : q * denb, tb = (KIND == 0) ? fma(q, dena, -h.pa) : q * dena;
                SF[0] = fma(ta, !(fabs(da) > cutlim) ? 1.0 : 0.0, SF[0]);  // 0/1 factor: see eval_pair
                SF[1] = fma(tb, !(fabs(db) > cutlim) ? 1.0 : 0.0, SF[1]);
            } else if (KIND == 0 && !LUMP) {
                SF[0] += fma(q, denb, -h.pa);
                SF[1] += fma(q, dena, -h.pa);
            } else {
                SF[0] = fma(q, denb, SF[0]);
                SF[1] = fma(q, dena, SF[1]);
            }
        } else {
            SF[0] += eval_one_fast<KIND, M2, TEST>(h, b, WN[0]);
            SF[1] += eval_one_fast<KIND, M2, TEST>(h, b, WN[1]);
        }
    } else {
        const f2 t = eval_one_fast2<KIND, M2, TEST, FULL>(h, WN);
        SF[0] += t.x;
        SF[1] += t.y;
    }
}

template <int KIND, bool M2, bool TEST, bool LUMP, bool FULL = false, typename R, typename H>
__device__ __forceinline__ void eval_loop2(const H *sA, const HotB *sB, int j0, int j1, const double (&WN)[2], R (&SF)[2]) {
    constexpr bool needB = sizeof(R) == 8 && M2 && KIND != 2;
    // two lines per trip, records fetched one line ahead into ping-pong registers (no copies)
    H h0 = sA[j0];
    double b0 = needB ? sB[j0].pb : 0.;
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        const H h1 = sA[j + 1];
        const double b1 = needB ? sB[j + 1].pb : 0.;
        eval_one2<KIND, M2, TEST, LUMP, FULL>(h0, b0, WN, SF);
        const int jn = (j + 2 < j1) ? j + 2 : j + 1;
        h0 = sA[jn];
        if (needB) b0 = sB[jn].pb;
        eval_one2<KIND, M2, TEST, LUMP, FULL>(h1, b1, WN, SF);
    }
    if (j < j1) eval_one2<KIND, M2, TEST, LUMP, FULL>(h0, b0, WN, SF);
}

template <int KIND, bool M2, bool TEST, bool FULL = false, typename R, typename H>
__device__ __forceinline__ void eval_fast2(const H *sA, const HotB *sB, int j0, int j1, const double (&WN)[2], R (&SF)[2]) {
    if (j0 >= j1) return;
    if constexpr (FULL) {
        eval_loop2<KIND, M2, TEST, false, true>(sA, sB, j0, j1, WN, SF);
        return;
    }
    if constexpr (sizeof(R) == 8 && KIND == 0 && !M2 && !TEST) {
        // untested one-resonance sub-runs (<= 64 lines: they never cross a mask word): the pedestal is the same for
        // every lane, so its sum is formed once per wave (one LDS read per lane + a wave reduction) and subtracted
        // after the loop - one FMA per evaluation instead of FMA + add
        if (j1 - j0 >= 16) {
            const int lane = (int)__lane_id();
            const double ped = wave_sum((j0 + lane < j1) ? sA[j0 + lane].pa : 0.);
            eval_loop2<KIND, M2, TEST, true>(sA, sB, j0, j1, WN, SF);
            SF[0] -= ped;
            SF[1] -= ped;
            return;
        }
    }
    eval_loop2<KIND, M2, TEST, false>(sA, sB, j0, j1, WN, SF);
}

// ---- round 5: Voigt candidates WITHOUT Y factors as ordinary lines + a correction (double precision) ------------------------------
// The general loop above costs ~40 instructions per (line, wavenumber of the lane) against ~7 in the class loops, for lines of
// which one or two wavenumbers of a whole tile take the Voigt shape (c3: 1.3 % of the lines, 24 % of the evaluate time).  Such a
// line now walks the class loops like any other - every lane takes its Lorentz term - and a SCAN of the candidates (one LDS
// record, one compare and one ballot per line and wavenumber of the lane) queues the (line, lane, k) triples within 100 Doppler
// widths (modm.f90:427).  voigt_flush_corr evaluates the queued shapes densely and hands each owner lane
//     S~ SLS_Voigt - (the Lorentz term the class loops added for that lane),
// the second by eval_one_fast<.., true, true>, i.e. with the per-lane conditions of the general formula.  The two Lorentz values
// differ by rounding only (paired reciprocals in the loops): an error of 1e-16 of the LORENTZ term, which exceeds the Voigt value
// by HWHM_D / HWHM_C at most (<= 1e3 in any atmosphere): far inside the tolerance.  Single precision keeps the general loop
// (float terms would leave 1e-7 x that ratio).  Candidates that carry Y factors walk the Y-factor loop (eval_general without its
// Voigt test, eval_o2_coupled) and are corrected against general_term.  One queue serves all wavenumbers of a lane:
// entry = Y << 15 | line << 7 | k << 6 | lane.
template <int KIND, int WPL, typename H>
__device__ __forceinline__ void voigt_flush_corr(const H *sA, const HotB *sB, const ColdLine *sCold, const unsigned short *vq, int n,
                                                 const double (&WNk)[WPL], int mol, double (&SFk)[WPL], double wscale, int *errflag) {
    const int lane = (int)__lane_id();
    const unsigned rec = (lane < n) ? vq[lane] : 0u;
    const int j = (int)((rec >> 7) & 255u), kk = (int)((rec >> 6) & 1u), owner = (int)(rec & 63u);
    double WNi = __shfl(WNk[0], owner);
    if constexpr (WPL >= 2) {
        const double w1 = __shfl(WNk[1], owner);
        WNi = kk ? w1 : WNi;
    }
    double val = 0.;
    if (lane < n) {
        const HotA h = widen(sA[j]);
        const HotB b = sB[j];
        const ColdLine c = sCold[j];
        const double SLS = lsf_sdvoigt(mol, (int)((c.info >> 6) & 3), 1.0, 1.0, b.c1 * c.hw, b.gp1 - 1., c.hw, WNi, h.xnu, c.hwd,
                                       (double)c.sdep, cold_xl3(c, mol, errflag), errflag);
        double lor;
        if (rec >> 15) lor = general_term<KIND>(h, b, WNi);          // the line walked the Y-factor loop (eval_general / eval_o2_coupled)
        else if constexpr (KIND == 2) lor = eval_one_fast<2, false, true>(h, 0., WNi);
        else lor = eval_one_fast<KIND, true, true>(h, b.pb, WNi);
        val = (c.stild * wscale) * SLS - lor;
    }
    for (int it = 0; it < n; it++) {  // wave-uniform trip count and indices; queue order = summation order (deterministic)
        const int lo = __builtin_amdgcn_readlane(__double2loint(val), it), hi = __builtin_amdgcn_readlane(__double2hiint(val), it);
        const int r = __builtin_amdgcn_readlane((int)rec, it);
        const double v = __hiloint2double(hi, lo);
        if (lane == (r & 63)) {
            if (WPL >= 2 && ((r >> 6) & 1)) SFk[WPL >= 2 ? 1 : 0] += v;
            else SFk[0] += v;
        }
    }
}

// scan of the Voigt candidates `cand` (bit i = line jbase + i of the chunk; all of one molecule run): queue what lies within 100
// Doppler widths of a wavenumber of the lane
template <int KIND, int WPL, typename H>
__device__ __forceinline__ void voigt_scan(unsigned long long cand, unsigned long long ymask, int jbase, const H *sA, const HotB *sB, const ColdLine *sCold,
                                           const double (&WNk)[WPL], int mol, double (&SFk)[WPL], double wscale, int *errflag,
                                           unsigned short *vq, int &nq) {
    static_assert(WPL <= 2, "one bit for k in a queue entry");
    const int lane = (int)__lane_id();
    while (cand) {
        const int i = (int)__builtin_ctzll(cand);
        cand &= cand - 1ull;
        const int j = jbase + i;
        const unsigned ybit = (unsigned)((ymask >> i) & 1ull) << 15;
        const double xnu = rec_xnu(sA[j]), d100 = sB[j].d100;   // (wave-uniform address: one LDS read serves the wave)
        const double cutlim = (KIND == 1) ? (double)sA[j].pa : 25.;   // (O2: the limit sits in the pa slot, +inf for a coupled line)
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            const double ad = fabs(WNk[k] - xnu);
            const bool useV = !(ad > cutlim) && !(ad > d100);     // modm.f90:384 / :755, :427 (d100 >= 0 for a candidate)
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(useV);
            if (mv != 0ull) {
                const int add = __popcll(mv);
                if (nq + add > 64) {
                    voigt_flush_corr<KIND, WPL>(sA, sB, sCold, vq, nq, WNk, mol, SFk, wscale, errflag);
                    nq = 0;
                }
                if (useV) vq[nq + __popcll(mv & ((1ull << lane) - 1ull))] = (unsigned short)(ybit | (j << 7) | (k << 6) | lane);
                nq += add;
            }
        }
    }
}

// Class masks of 64 lines: runs of ones shorter than 8 are cleared / runs of zeros shorter than 8 are filled.  A sub-run
// switch costs about as much as ten evaluations; a line may always take the more general loop (tested instead of untested,
// two resonances with a per-lane 0/1 factor instead of one), so short islands join their neighbours.
__device__ __forceinline__ unsigned long long open_runs8(unsigned long long x) {
    unsigned long long e = x & (x >> 1);
    e &= e >> 2;
    e &= e >> 4;  // bit i set: ones at i .. i+7
    e |= e << 1;
    e |= e << 2;
    e |= e << 4;
    return e;
}
__device__ __forceinline__ unsigned long long close_runs8(unsigned long long x) { return ~open_runs8(~x); }


// mAL / mM2: per 64 lines of the chunk one bit per line (all lanes live / two resonances); the run [j0, j1) is walked
// in sub-runs of constant class, in line order - the summation order stays the reference's
// mFar (may be null): lines whose contribution has been moved into the far-field moments of the tile (far_moments below);
// their LDS records are null and the sub-run is skipped
// mV: Voigt candidates (zeta <= 0.99 and some wavenumber of the tile within 100 Doppler widths, modm.f90:427) - ONLY these
// lines take the general loop whose lanes ballot for the (speed-dependent) Voigt shapes; mY: lines whose shapes carry
// line-coupling Y factors (general loop without the Voigt test; first-order coupled O2 has a loop of its own)
#ifdef LINES_CLASS_STATS
__device__ unsigned long long g_eval_stat[32];  // per class: cycles, sub-runs, lines (debug builds only; the contended atomics slow the kernel several times)
#define EVAL_STAT(c) do { if (__lane_id() == 0) { atomicAdd(&g_eval_stat[(c)], (unsigned long long)(__builtin_readcyclecounter() - t_sub)); \
    atomicAdd(&g_eval_stat[8 + (c)], 1ull); atomicAdd(&g_eval_stat[16 + (c)], (unsigned long long)len); } } while (0)
#else
#define EVAL_STAT(c)
#endif
template <int KIND, typename R, typename H, int WPL, bool PACKED = false, unsigned UNI_BOFF = 0u>
__device__ __forceinline__ void eval_dispatch(const unsigned long long *mAL, const unsigned long long *mM2,
                                              const unsigned long long *mFar, const unsigned long long *mV,
                                              const unsigned long long *mY, const H *sA, const HotB *sB, const ColdLine *sCold,
                                              int j0, int j1,
                                              const double (&WNk)[WPL], int mol, R (&SFk)[WPL], double wscale, int *errflag,
                                              unsigned short *vq, int rec_off = 0, const unsigned long long *mFull = nullptr,
                                              const double *tst = nullptr, const double *wlim = nullptr) {
    // tst / wlim (may be null; two wavenumbers per lane and pass): tst[2 g], tst[2 g + 1]
    // = least and largest centre among the plain tested lines of 64-line group g; wlim[2 k], wlim[2 k + 1] = least and largest k-th
    // wavenumber of THIS wave.  A tested sub-run whose every line is more than 25 cm-1 from every k-th wavenumber of the wave adds
    // nothing to those: on a dense tile the lines cut by the rule sit 25 cm-1 beyond one edge, and for half of the (wave, k)
    // pairs all of them are out of reach - the walk used to evaluate them for all and let the clamp return zero.
    // mFull (may be null; single precision, two wavenumbers per lane): two-resonance untested lines whose negative resonance
    // is within reach of EVERY wavenumber of the tile - a subset of mAL & mM2 (eval_one_fast2 FULL)
    constexpr bool HAS_FULL = sizeof(R) == 4 && WPL == 2 && KIND != 2 && !PACKED;
    int nq[WPL];  // Voigt pairs queued per wavenumber of the lane (vq + 64 k)
#pragma unroll
    for (int k = 0; k < WPL; k++) nq[k] = 0;
#ifdef MONORTM_NO_UNIFIED   // A/B builds: the per-class loops of rounds 1-3
    constexpr bool UNIFIED = false;
#else
    // UNI_BOFF: byte distance from sA to sB when both live in one padded LDS object (lines_kernel), else 0
    constexpr bool UNIFIED = UNI_BOFF != 0u && WPL == 1 && sizeof(R) == 8 && !PACKED;
#endif
    // VSCAN (round 5, double precision): Voigt candidates walk the loops of the Lorentz shapes like any other line and are
    // corrected afterwards (voigt_scan); the general loop with its Voigt test is not instantiated then
#ifdef MONORTM_NO_VSCAN
    constexpr bool VSCAN = false;
#else
    constexpr bool VSCAN = sizeof(R) == 8 && !PACKED && WPL <= 2;
#endif
    int nqs = 0;  // (VSCAN: one queue for all wavenumbers of the lane)
    int j = j0, wc = -1;  // wc: the 64-line group whose masks are held in scalar registers
    unsigned long long a = 0ull, m = 0ull, f = 0ull, v = 0ull, y = 0ull, fu = 0ull;
    while (j < j1) {
#ifdef LINES_CLASS_STATS
        const unsigned long long t_sub = __builtin_readcyclecounter();
#endif
        const int w = j >> 6, bit = j & 63;
        if (w != wc) {
            f = (mFar == nullptr) ? 0ull : uni64(mFar[w]);
            if (mFar != nullptr) {
                // every line of this group that belongs to the run sits in the far-field sums (most groups of a dense tile: four
                // fifths of its window): nothing to evaluate, and none of the other masks is needed - the walk over such groups was
                // 18 % of c3's evaluate time when all five masks were fetched first
                const int end = min(64, j1 - (w << 6));
                const unsigned long long span = ((end >= 64) ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << bit) - 1ull);
                if ((f & span) == span) {
#ifdef LINES_CLASS_STATS
                    const int len = end - bit;
#endif
                    EVAL_STAT(2);
                    j = (w << 6) + end;
                    continue;
                }
            }
            a = uni64(mAL[w]);
            m = (KIND == 2) ? 0ull : uni64(mM2[w]);
            v = uni64(mV[w]);
            y = uni64(mY[w]);
            if constexpr (VSCAN) {
                const int end = min(64, j1 - (w << 6));
                const unsigned long long span = ((end >= 64) ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << bit) - 1ull);
                const unsigned long long cand = v & span;   // (a group is entered once per run: its candidates are scanned here)
                if (cand) voigt_scan<KIND, WPL>(cand, y, w << 6, sA, sB, sCold, WNk, mol, SFk, wscale, errflag, vq + 64 * WPL, nqs);
                v = 0ull;   // ... and walk as what they otherwise are: fast-class lines, or lines with Y factors
            }
    #ifndef MONORTM_NO_FULL
            if constexpr (HAS_FULL) fu = (mFull == nullptr) ? 0ull : uni64(mFull[w]);
#endif
            wc = w;
        }
        const bool al = (a >> bit) & 1ull, m2 = (m >> bit) & 1ull, far = (f >> bit) & 1ull, vg = VSCAN ? false : (bool)((v >> bit) & 1ull),
                   yf = (y >> bit) & 1ull;
        if constexpr (UNIFIED) {
            // every ordinary line up to the next rare shape / far-field line in ONE loop, whatever its fast class
            if (!vg && !yf && !far) {
                const unsigned long long special = (v | y | f) >> bit;
                int len = special ? (int)__builtin_ctzll(special) : 64;
                len = min(min(len, 64 - bit), j1 - j);
                if constexpr (UNIFIED) SFk[0] = eval_unified<KIND, UNI_BOFF>(sA, j, len, ~a >> bit, m >> bit, WNk[0], SFk[0]);
                EVAL_STAT(7);
                j += len;
                continue;
            }
        }
        // a rare shape cuts a sub-run whatever the fast classes say; among ordinary lines the fast classes cut it too
        unsigned long long diff = (vg ? ~v : v) | (yf ? ~y : y);
        if (!vg && !yf) diff |= (al ? ~a : a) | (m2 ? ~m : m) | (far ? ~f : f);
        const bool full = HAS_FULL && ((fu >> bit) & 1ull);
        if (HAS_FULL && !vg && !yf) diff |= full ? ~fu : fu;
        diff >>= bit;
        int len = diff ? (int)__builtin_ctzll(diff) : 64;
        len = min(min(len, 64 - bit), j1 - j);
        const int je = j + len;
        if (vg || yf) {  // one wavenumber at a time
#pragma unroll
            for (int k = 0; k < WPL; k++) {
                if (vg) SFk[k] = (R)eval_general<KIND, true, PACKED>(sA, sB, sCold, j, je, WNk[k], mol, (double)SFk[k], wscale, errflag, vq + 64 * k, nq[k], rec_off);
                else if (KIND == 1) SFk[k] = (R)eval_o2_coupled(sA, sB, j, je, WNk[k], (double)SFk[k]);
                else {
                    int none = 0;
                    SFk[k] = (R)eval_general<KIND, false>(sA, sB, sCold, j, je, WNk[k], mol, (double)SFk[k], wscale, errflag, nullptr, none);
                }
            }
            EVAL_STAT(vg ? 0 : 1);
            j = je;
            continue;
        }
        if (far) {
            EVAL_STAT(2);
            j = je;
            continue;
        }
        if constexpr (WPL == 1) {
            const double WN = WNk[0];
            R SF = SFk[0];
            constexpr bool pairs = sizeof(R) == 8 && KIND != 2;  // double precision: two lines share a reciprocal
            if (m2) {
                if constexpr (pairs) {
                    if (al) SF = eval_pair<KIND, true, false>(sA, sB, j, je, WN, SF);
                    else SF = eval_pair<KIND, true, true>(sA, sB, j, je, WN, SF);
                } else {
                    if (al) SF = eval_fast<KIND, true, false>(sA, sB, j, je, WN, SF);
                    else SF = eval_fast<KIND, true, true>(sA, sB, j, je, WN, SF);
                }
            } else if (al) {
                if constexpr (pairs) SF = eval_pair_fast<KIND>(sA, j, je, WN, SF);
                else SF = eval_fast<KIND, false, false>(sA, sB, j, je, WN, SF);
            } else {
                if constexpr (pairs) SF = eval_pair<KIND, false, true>(sA, sB, j, je, WN, SF);
                else SF = eval_fast<KIND, false, true>(sA, sB, j, je, WN, SF);
            }
            SFk[0] = SF;
        } else {
            if (m2) {
                if (al) {
                    if constexpr (HAS_FULL) {
                        if (full) eval_fast2<KIND, true, false, true>(sA, sB, j, je, WNk, SFk);
                        else eval_fast2<KIND, true, false>(sA, sB, j, je, WNk, SFk);
                    } else eval_fast2<KIND, true, false>(sA, sB, j, je, WNk, SFk);
                } else eval_fast2<KIND, true, true>(sA, sB, j, je, WNk, SFk);
            } else if (al) {
                eval_fast2<KIND, false, false>(sA, sB, j, je, WNk, SFk);
            } else {
                bool done = false;
#ifdef MONORTM_NO_SGL_TSKIP
                if constexpr (WPL == 2 && !PACKED && sizeof(R) == 8) {
#else
                if constexpr (WPL == 2 && !PACKED) {
#endif
                    if (tst != nullptr) {
                        const double tlo = tst[2 * w], thi = tst[2 * w + 1];   // (wave-uniform LDS reads)
                        // modm.f90:384 (O2: :755, the same 25 cm-1 for an uncoupled line): out of reach <=> centre - WN > 25 for the
                        // largest WN, or WN - centre > 25 for the least
                        const bool ex0 = (tlo - wlim[1] > 25.) || (wlim[0] - thi > 25.), ex1 = (tlo - wlim[3] > 25.) || (wlim[2] - thi > 25.);
                        if (ex0 || ex1) {
                            done = true;
                            if (!(ex0 && ex1)) {
                                const int k = ex0 ? 1 : 0;
                                if constexpr (KIND == 2 || sizeof(R) == 4) SFk[k] = eval_fast<KIND, false, true>(sA, sB, j, je, WNk[k], SFk[k]);
                                else SFk[k] = eval_pair<KIND, false, true>(sA, sB, j, je, WNk[k], SFk[k]);
                            }
                        }
                    }
                }
                if (!done) eval_fast2<KIND, false, true>(sA, sB, j, je, WNk, SFk);
            }
        }
        EVAL_STAT(m2 ? (al ? 3 : 4) : (al ? 5 : 6));
        j = je;
    }
#pragma unroll
    for (int k = 0; k < WPL; k++)
        if (nq[k] > 0) SFk[k] = (R)voigt_flush<KIND, PACKED>(sA, sB, sCold, vq + 64 * k, nq[k], WNk[k], mol, (double)SFk[k], wscale, errflag, rec_off);
    if constexpr (VSCAN) {
        if (nqs > 0) voigt_flush_corr<KIND, WPL>(sA, sB, sCold, vq + 64 * WPL, nqs, WNk, mol, SFk, wscale, errflag);
    }
}

// ------------------------------------------------------------------------------------------------
// Far field of a tile.  A one-resonance line whose centre lies at least FAR_KAPPA half-widths r of the tile away from the
// tile's centre w0 contributes a smooth function of x = (WN - w0) / r in [-1, 1] to every wavenumber of the tile.  Round 5: the
// function is expanded in CHEBYSHEV polynomials instead of powers of t = WN - w0.  With the pole z = (delta + i h) / r
// (delta = Xnu - w0, h = HWHM):
//     a2 / ((t - delta)^2 + h^2) = (a2 / (r h)) Im[1 / (x - z)],     1 / (z - x) = (2 / s) sum'_n w^n T_n(x),
//     s = sqrt(z^2 - 1),  w = z - s = 1 / (z + s),  |w| = 1 / rho,  rho = |delta| / r + sqrt((delta / r)^2 - 1)
// (sum' = the n = 0 term halved).  The coefficients are a geometric sequence like those of the Taylor series were - the same
// four multiply-adds per term - but the ratio is 1 / rho instead of r / |delta|: at |delta| = 1.6 r the terms have decayed to
// 1e-15 after 35 of them where the power series needed 75 (and was cut at 60, truncation 6e-13), at 3 r after 22 instead of 33,
// at 19 r (the 25 cm-1 distance of a dense tile) after 12 instead of 14.  So the series of a tile cost ~0.6 of what they did, and
// the least distance can come down (kappa 1.6 -> 1.2: 59 terms), which moves more of the near lines out of the direct loops.
// Everything stays real: the imaginary parts are carried divided by h (im' = im / h), so that no h appears in a denominator:
//     (a + i h a')(b + i h b') = (a b - h^2 a' b') + i h (a b' + a' b).
// The prepare stage adds c_n = -(a2 / r) Im'[g w^n], g = 2 / s, of such lines to FAR_P sums per molecule; a lane then evaluates
// one Chebyshev sum per molecule run (Clenshaw) instead of one Lorentzian per line.  Checked against the direct formula: 3e-14
// at kappa = 1.6, 9e-14 at 1.3 (relative to the term, h from 1e-5 to 0.2; tests/test_hip_parity.py::test_far_field_dense_grid).
// ------------------------------------------------------------------------------------------------
// (FAR_P = MONORTM_FAR_P, 56 sums: device_common.hpp - the host sizes far_kernel's buffers with it)
#ifndef MONORTM_FAR_KAPPA
#define MONORTM_FAR_KAPPA 1.2
#endif
constexpr double FAR_KAPPA = MONORTM_FAR_KAPPA;
// terms until |w|^n < 1e-15 at the least distance kappa: 34.5 / ln(kappa + sqrt(kappa^2 - 1)) + 3 - 59 at 1.2 (round 5, first session: 60 sums; now 56, 1.86^-56 = 7e-16: a
// multiple of four, the butterfly forms four at a time; measured on c3: kappa / sums 1.6 / 36: 5.54 ms, 1.45 / 44: 5.48, 1.3 / 52:
// 5.55, 1.2 / 60: 5.36), 27 at 2.25 for the one-wave tiles (they evaluate a line 2-4 times only:
// a far line must cost its owner lane less than the direct evaluation costs every wave; round 4's distance, 28 sums instead of
// its 34 powers).  At most 63 sums: one lane per sum.
#ifndef MONORTM_FAR_P1
#define MONORTM_FAR_P1 28
#endif
#ifndef MONORTM_FAR_KAPPA1
#define MONORTM_FAR_KAPPA1 2.25
#endif
__host__ __device__ constexpr int far_p(int evals_per_line) { return evals_per_line >= 8 ? FAR_P : MONORTM_FAR_P1; }
__host__ __device__ constexpr double far_kappa(int evals_per_line) { return evals_per_line >= 8 ? FAR_KAPPA : MONORTM_FAR_KAPPA1; }

// index-space description of the lines of one molecule that are FAR for an interval [a, b] of wavenumbers (centre c, half-width
// rho, all in the units of the sorted table centres vnu, pad = the largest pressure shift of the layer + 1e-6):
//   low  side  [lowS, lowE)   = vnu in [b - 25 + pad, c - kappa rho - pad]   every wavenumber of the interval within 25 cm-1, pole far
//   high side  [highS, highE) = vnu in [c + kappa rho + pad, a + 25 - pad]
//   minus [.., e0)     = vnu <  kappa rho - c + pad     (the negative resonance's pole at -vnu would be near)
//   minus [e1s, e1e)   = vnu in (25 - b - pad, 25 - a + pad]   (negative resonance within reach of some wavenumbers of the interval only)
// lines below e1s carry both resonances (WN + Xnu <= 25 for every wavenumber: modm.f90:713), lines from e1e on the positive one alone.
// A child interval's set contains its parent's (kappa > 1), so an interval expands its set minus its parent's and lines_kernel walks
// the candidates minus the tile's set.  All zeros = no far lines.
struct FarGeom { int lowS, lowE, highS, highE, e0, e1s, e1e, spare; };
__host__ __device__ inline bool far_contains(const FarGeom &g, int i) {
    return ((i >= g.lowS && i < g.lowE) || (i >= g.highS && i < g.highE)) && i >= g.e0 && !(i >= g.e1s && i < g.e1e);
}

// All 64 lanes call this; `on` marks the lanes that own a far line, `on2` those whose negative resonance (centre -Xnu,
// i.e. delta2 = -(w0 + Xnu)) is included for every wavenumber of the tile and is expanded with it.
// mom[0..FAR_P-1] += sum over lanes of the Chebyshev coefficients (mom[0] holds TWICE the coefficient of T_0: Clenshaw's
// convention) (+ the quadratic of the CO2 pedestal, c0..c2 in powers of t), mom[FAR_P] += sum of the constant pedestals.  The
// series is cut where the largest |w| of the wave has decayed below 1e-15 (lines arrive sorted, so a wave's lines sit at similar
// distances).  The wave sums are formed in a fixed order (deterministic); one lane per coefficient collects its sum in a register
// and the lanes add theirs to LDS at the end.
// Sums over the 64 lanes of FOUR values at a time by a halving butterfly: v_permlane32_swap / v_permlane16_swap (gfx950)
// exchange halves of two registers, so one add finishes the (lane, lane + 32) sums of two values and another the row sums of
// both; the 16-lane row sums of the one register left are four DPP steps.  21 instructions per four sums instead of 80;
// afterwards row r (lanes 16 r .. 16 r + 15) holds the total of value r.
typedef unsigned far_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double swap_add32(double a, double b) {  // lanes < 32: a(l) + a(l + 32); lanes >= 32: b(l - 32) + b(l)
    const far_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const far_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double swap_add16(double a, double b) {  // rows 0, 2: a(row) + a(row + 1); rows 1, 3: b(row - 1) + b(row)
    const far_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const far_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double row_sum16(double v) {  // every lane: the sum over its row of 16 lanes
    v += dpp_move<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);  // row_mirror
    return v;
}
// coefficient n of the series ends up in the lane L with far_moment_of_lane(L) == n: four coefficients per trip, row r of the
// trip G holds coefficient 4 G + r and its lane with (lane & 15) == G keeps it
__device__ __forceinline__ int far_moment_of_lane(int lane) { return 4 * (lane & 15) + (lane >> 4); }

// sqrt of a positive normal double to 1 ulp: v_rsq_f64 seed (2^-26 relative) and two coupled Newton steps - the library sqrt
// spends twice the instructions on range checks these operands (O(1) .. O(1e3)) do not need
__device__ __forceinline__ double fsqrt_pos(double x) {
    double r = __builtin_amdgcn_rsq(x);
    double y = x * r;             // ~ sqrt(x)
    double e = 0.5 * r;           // ~ 1 / (2 sqrt(x))
    y = fma(fma(-y, y, x), e, y);
    e = fma(fma(-2.0 * e, y, 1.0), e, e);   // refresh 1 / (2 y) for the second step: e (2 - 2 e y)  ->  e + e (1 - 2 e y)
    return fma(fma(-y, y, x), e, y);
}

// The generator of one pole: w = 1 / (z + s) and g = 2 / s in (re, im / h) form, for z = (delta + i h) / r, |delta| >= kappa r.
// rinv = 1 / r, hw2 = h^2.  `off`: a lane without a far line (every output 0).
struct FarPole { double wre, wim, gre, gim, kw; };   // kw = h^2 wim (the recurrence's -h^2 a' b' term)
__device__ __forceinline__ FarPole far_pole(bool on, double delta, double hw2, double rinv) {
    FarPole p{0., 0., 0., 0., 0.};
    if (on) {
        const double zr = delta * rinv, r2 = rinv * rinv;
        const double A = fma(zr, zr, -fma(hw2, r2, 1.0));   // Re(z^2 - 1) = (delta^2 - h^2) / r^2 - 1  (> 0: |delta| >= kappa r > r)
        const double Bp = 2.0 * zr * rinv;                  // Im(z^2 - 1) / h
        const double mod = fsqrt_pos(fma(A, A, (Bp * Bp) * hw2));
        // (A < 0 - a Lorentz width beyond 0.66 r: mod + A cancels; the same number is (Im(z^2 - 1))^2 / (2 (mod - A)))
        double sre = fsqrt_pos((A < 0.) ? 0.5 * ((Bp * Bp) * hw2) * frcp_any(mod - A) : 0.5 * (mod + A));            // Re sqrt: the branch with |w| < 1 has the sign of delta
        sre = (delta < 0.) ? -sre : sre;
        const double sim = Bp * (0.5 * frcp_any(sre));      // Im sqrt / h
        const double ure = zr + sre, uim = rinv + sim;      // z + s (no cancellation: same signs)
        const double dinv = frcp_any(fma(ure, ure, (uim * uim) * hw2));
        p.wre = ure * dinv;
        p.wim = -uim * dinv;
        const double sinv = 2.0 * frcp_any(fma(sre, sre, (sim * sim) * hw2));
        p.gre = sre * sinv;
        p.gim = -sim * sinv;
        p.kw = hw2 * p.wim;
    }
    return p;
}

template <bool TWO>
__device__ __forceinline__ double far_series(int order, double amp, const FarPole &p1, const FarPole &p2) {
    double re = p1.gre, im = p1.gim, re2 = p2.gre, im2 = p2.gim, mine = 0.;
    const int slot = (int)__lane_id() & 15;
#pragma unroll 1
    for (int G = 0; 4 * G < order; G++) {
        double x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            x[i] = TWO ? amp * (im + im2) : amp * im;
            const double ren = fma(re, p1.wre, -(im * p1.kw));
            im = fma(re, p1.wim, im * p1.wre);
            re = ren;
            if (TWO) {
                const double ren2 = fma(re2, p2.wre, -(im2 * p2.kw));
                im2 = fma(re2, p2.wim, im2 * p2.wre);
                re2 = ren2;
            }
        }
        // lanes < 32: x0 | lanes >= 32: x2;  then rows: x0, x1, x2, x3
        const double w = row_sum16(swap_add16(swap_add32(x[0], x[2]), swap_add32(x[1], x[3])));
        mine = (slot == G) ? w : mine;
    }
    return mine;
}

// terms of the series of a pole at |delta| = dmin (the nearest of the wave) until rho^-n < 1e-15
__device__ __forceinline__ int far_order(double dmin, double rinv, int P) {
    const float z = (float)(dmin * rinv);
    const float rho = z + __builtin_sqrtf(fmaxf(z * z - 1.f, 0.f));
    return min(P, max(8, (int)(34.5f / __logf(fmaxf(rho, 1.0001f))) + 3));
}

// STORE: the sums REPLACE what mom holds (a wave's own per-chunk sums) instead of being added to it
template <int P = FAR_P, bool STORE = false>
__device__ __forceinline__ void far_moments(bool on, double delta, bool on2, double delta2, double hw2, double a2, double ped,
                                            bool quad, double c0, double c1, double c2, double rr, double *mom) {
    const double rinv = frcp_any(rr);
    const FarPole p1 = far_pole(on, delta, hw2, rinv), p2 = far_pole(on2, delta2, hw2, rinv);
    const int lane = (int)__lane_id();
    const double amp = (on || on2) ? -(a2 * rinv) : 0.0;   // (on2 alone: a half-far line - only its negative resonance is expanded)
    const double dmin = wave_min(fmin(on ? fabs(delta) : __builtin_inf(), on2 ? fabs(delta2) : __builtin_inf()));
    const int order = far_order(dmin, rinv, P);
    // (a wave of half-far lines alone has one pole per line as well: the second one)
    double mine = (__ballot(on2) == 0ull) ? far_series<false>(order, amp, p1, p2)
                                          : ((__ballot(on) == 0ull) ? far_series<false>(order, amp, p2, p1) : far_series<true>(order, amp, p1, p2));
    if (quad) {
        // wave-uniform: the CO2 pedestal -pa (2 - (t - delta)^2 / 625) = c0 + c1 t + c2 t^2 with t = r x, t^2 = r^2 (T_2 + 1) / 2:
        // T_0: c0 + c2 r^2 / 2 (stored twice: Clenshaw's convention), T_1: c1 r, T_2: c2 r^2 / 2  (coefficients 0, 1, 2 = lanes 0, 16, 32)
        const double s0 = wave_sum(c0), s1 = wave_sum(c1), s2 = wave_sum(c2) * (0.5 * rr * rr);
        mine += (lane == 0) ? 2.0 * (s0 + s2) : ((lane == 16) ? s1 * rr : ((lane == 32) ? s2 : 0.));
    }
    const double tp = wave_sum((on || on2) ? ped : 0.0);
    const int n = far_moment_of_lane(lane);
    if constexpr (STORE) {
        if (n < P) mom[n] = (n < order || (quad && n < 3)) ? mine : 0.;
        if (lane == P) mom[P] = tp;
    } else {
        if (n < order || (quad && n < 3)) mom[n] += mine;
        if (lane == P) mom[P] += tp;
    }
}

// The far field of a lane: sum'_n mom[n] T_n(x) by Clenshaw's recurrence (mom[0] = twice the coefficient of T_0), x in [-1, 1].
// NW waves' sums are added per coefficient in wave order.  stride = distance between the waves' arrays in doubles.
// gm (may be null): the sums of far_kernel for this (profile, layer, tile, molecule) in global memory, added as one more source.
template <int P, int NW, int WPL>
__device__ __forceinline__ void far_eval(const double *mom, int stride, const double (&x)[WPL], double (&out)[WPL], const double *gm = nullptr) {
    double b1[WPL], b2[WPL], x2[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) { b1[k] = 0.; b2[k] = 0.; x2[k] = x[k] + x[k]; }
#pragma unroll 2
    for (int n = P - 1; n >= 1; n--) {
        double mn = gm ? gm[n] : 0.;
#pragma unroll
        for (int w = 0; w < NW; w++) mn += mom[w * stride + n];
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            const double t = fma(x2[k], b1[k], mn - b2[k]);
            b2[k] = b1[k];
            b1[k] = t;
        }
    }
    double m0 = gm ? gm[0] : 0.;
#pragma unroll
    for (int w = 0; w < NW; w++) m0 += mom[w * stride];
#pragma unroll
    for (int k = 0; k < WPL; k++) out[k] = fma(x[k], b1[k], 0.5 * m0 - b2[k]);
}

// ------------------------------------------------------------------------------------------------
// prepare_line: everything of a line that does not depend on the wavenumber, for one layer (modm.f90:324-415, hoisted out
// of the wavenumber loop): shifted centre, S~ (INTENS), Lorentz and Doppler widths (HALFWHM_C / _D), coupling factors
// at the layer temperature, pedestal -> the LDS records of lineshape.hpp.
//   lay  : the layer scalars parked in LDS (RHORAT, RP, RP2, ln(T/T0), ..., rho_molec(1:7), temperature bracket)
//   scor, dop : Q(296)/Q(T) and the Doppler factor per (molecule, isotopologue);  sWl : column amounts of the layer
//   sWn[TW]   : the tile's wavenumbers (ascending)
//   fAL / fM2 : class flags of the line for the fast loops (all lanes live / negative resonance within reach)
//   fV / fY   : Voigt candidate for this tile / shape carries line-coupling Y factors (general loops, this line only)
// ------------------------------------------------------------------------------------------------
// The function is split where the tile comes in: line_physics() is everything that depends on the line and the layer only
// (what physics_kernel can form once per (layer, line) for all tiles of a dense grid), line_records() the classes and LDS
// records of the line for one tile.
struct LinePhys {
    double xnu, hw, hwd, stild;  // shifted centre, Lorentz and Doppler half widths, S~
    double c1, g;                // AIP (1/HW) RP and BIP RP2 of the shapes that carry Y factors, else 0
};
// In memory (physics_kernel -> lines_kernel, far_kernel) the record is split: 32 bytes that every reader wants, [state][line], and
// behind all of them 16 bytes of coupling factors that exist for coupled lines only (coupling code != 0 in the line's meta word) -
// an uncoupled line costs 32 bytes of traffic per reader instead of 48, and far_kernel (uncoupled molecules only) never sees the rest.
struct LinePhysM { double xnu, hw, hwd, stild; };
struct LinePhysY { double c1, g; };
struct PhysView {
    const LinePhysM *m;   // null: no physics pass
    const LinePhysY *y;
};
// state = profile * nlay_max + layer; nstates = nprof * nlay_max
__device__ __forceinline__ PhysView phys_view(const void *base, size_t state, size_t nstates, size_t nlines) {
    if (!base) return PhysView{nullptr, nullptr};
    const LinePhysM *m0 = reinterpret_cast<const LinePhysM *>(base);
    return PhysView{m0 + state * nlines, reinterpret_cast<const LinePhysY *>(m0 + nstates * nlines) + state * nlines};
}

// The layer scalars of LINES (INITI + head of LINES, modm.f90:868-883, :301-314) as values: lines_kernel parks them in LDS
// (one layer per workgroup), lines_state_kernel holds them per lane (lane = atmospheric state).
struct LayerScalars {
    double RHORAT, RP, RP2, lnRT, cTk, cT0, dTinv, RECTLC, TMPDIF;
    int ILC;
};

// line_physics_core: everything of a line that depends on the line and the layer only.  idx may be per lane (lines_kernel,
// physics_kernel: one lane per line) or wave-uniform (lines_state_kernel: one lane per layer - the table loads are then
// scalar loads).  rho_self = RHORAT W(mol) / WTOT; rho7 = rho_molec(1:7) (read only with IBRD); XIPSF = Q(296)/Q(T) of the
// line's isotopologue (0 for an unknown one); dopfac = HWHM_D / Xnu of the isotopologue.
// the table fields of one line that every evaluation needs (the coupling coefficients and the species-broadening data are
// read from the table on demand, through idx)
struct LineFields {
    double xnu0, s0adj;
    float alfa, hwhm, epp, tmpalf, pshift;
    uint32_t meta;
};
__device__ __forceinline__ LineFields load_line_fields(const DevLines &L, int idx) {
    LineFields f;
    f.xnu0 = L.vnu[idx]; f.s0adj = L.s0adj[idx];
    f.alfa = L.alfa[idx]; f.hwhm = L.hwhm[idx]; f.epp = L.epp[idx]; f.tmpalf = L.tmpalf[idx]; f.pshift = L.pshift[idx];
    f.meta = L.meta[idx];
    return f;
}

// what line_physics_core reads besides the line and the layer: the coupling scale factors of the call and the table arrays it
// indexes on demand (coupling coefficients, species-broadening data)
struct PhysParams {
    double sclcpl, sclhw, y0res;
    const double *lc;
    const int32_t *brd_flg;
    const float *brd_dat;
};
__device__ __forceinline__ PhysParams phys_params(const ModmArgs &a, const DevLines &L) {
    return PhysParams{a.sclcpl, a.sclhw, a.y0res, L.lc, L.brd_flg, L.brd_dat};
}

// PLAIN: the caller has seen that NO lane of the wave holds a line with a coupling code, an air-width or an air-shift conversion
// (meta bits 10-11, 13, 14): those blocks are left out - the divergent regions around them (three scalar instructions each, walked
// by every wave whether a lane needs them or not) go with them.  What remains is what such a line executes anyway: identical bits.
template <bool IBRD, bool PLAIN = false>
__device__ __forceinline__ LinePhys line_physics_core(const PhysParams &pp, int idx, int mol, const LineFields &lf,
                                                      const LayerScalars &ly, double rho_self, const double (&rho7)[MXBRD],
                                                      double XIPSF, double dopfac) {
    // the reference's expression order, each operation rounded (its build does not contract a*b+c): the shifted centre and
    // the widths feed the 25 cm-1, zeta and 100-Doppler-width decisions; the exponentials use explicit fma() of their own
#pragma clang fp contract(off)
    const uint32_t meta = lf.meta;
    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    const int ILC = ly.ILC;
    const double RHORAT = ly.RHORAT, RP = ly.RP, RP2 = ly.RP2, lnRT = ly.lnRT, cTk = ly.cTk, cT0 = ly.cT0, dTinv = ly.dTinv,
                 RECTLC = ly.RECTLC, TMPDIF = ly.TMPDIF;
    const int code = PLAIN ? 0 : (int)((meta >> 10) & 3);
    const double xnu0 = lf.xnu0;
    double alpf = lf.alfa, alps = lf.hwhm, delt = lf.pshift;
    const double E = lf.epp, XTILD = lf.tmpalf;
    if (!PLAIN && ((meta >> 13) & 1)) {  // O2 / N2: air width -> foreign width (lnfl_mod.f90:98-113)
        const double rvmr = (mol == 7) ? 0.21 : 0.79;
        alpf = (alpf - rvmr * alps) / (1.0 - rvmr);
    }
    if (!PLAIN && ((meta >> 14) & 1)) {
        const double rvmr = 0.21;
        delt = (delt - rvmr * (double)pp.brd_dat[(size_t)idx * 21 + 3 * 6 + 2]) / (1.0 - rvmr);
    }
    // line-coupling coefficients at the layer temperature (modm.f90:328-368)
    double AIP = 0., BIP = 0.;
#ifdef MONORTM_ABLATE_COUPLE
    if (false) {  // timing experiment: no coupling coefficients (wrong results)
#else
    if (code) {
#endif
        const double *s = pp.lc + (size_t)(meta >> 15) * 8;
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
        if (code == 1) { AIP = AIP * pp.sclcpl + pp.y0res; BIP = BIP * pp.sclcpl + pp.y0res; }
        if (code == 2) { AIP = AIP * pp.sclhw; BIP = BIP * pp.sclhw; }
    }
    double Xnu = xnu0 + (delt * RHORAT);
    const bool brd = IBRD && mol <= MXBRD;
    int bf[MXBRD];
    int sflg = 0;
    if (brd) {
        double s = 0.;
#pragma unroll
        for (int j = 0; j < MXBRD; j++) {
            bf[j] = pp.brd_flg[(size_t)idx * 7 + j];
            sflg += bf[j];
            s += rho7[j] * bf[j] * ((double)pp.brd_dat[(size_t)idx * 21 + 3 * j + 2] - delt);
        }
        Xnu = Xnu + s;
    }
    // INTENS (modm.f90:860-865); exp(a)/exp(b) folded into one exp
#ifdef MONORTM_EXP_SGPR_CONSTANTS
    double ex[4];
    {
        const double xa[4] = {(RADCT * E) * dTinv, -(Xnu * cTk), -(Xnu * cT0), XTILD * lnRT};
#ifdef MONORTM_ABLATE_EXP4
        for (int i = 0; i < 4; i++) ex[i] = fma(xa[i], 1e-3, 1.0);  // timing experiment: the four exponentials priced (wrong results)
#else
        exp_prep4(xa, ex);
#endif
    }
    const double S = lf.s0adj * ex[0] * XIPSF;
    const double STILD = S * ((1 + ex[1]) * frcp_any(Xnu * (1 - ex[2])));
    // HALFWHM_C (modm.f90:833-857)
    if (mol == 1 && alps == 0.) alps = 5 * alpf;
    const double rtx = ex[3];
#else
    const double S = lf.s0adj * exp_prep((RADCT * E) * dTinv) * XIPSF;
    const double STILD = S * ((1 + exp_prep(-(Xnu * cTk))) * frcp_any(Xnu * (1 - exp_prep(-(Xnu * cT0)))));
    // HALFWHM_C (modm.f90:833-857)
    if (mol == 1 && alps == 0.) alps = 5 * alpf;
    const double rtx = exp_prep(XTILD * lnRT);
#endif
    const double alfa0i = alpf * rtx, hwhmsi = alps * rtx;
    double HW = alfa0i * (RHORAT - rho_self) + hwhmsi * rho_self;
    if (brd && sflg > 0) {
        double alfsum = 0., rsum = 0.;
#pragma unroll
        for (int j = 0; j < MXBRD; j++) {
            const double hwj = pp.brd_dat[(size_t)idx * 21 + 3 * j], tmj = pp.brd_dat[(size_t)idx * 21 + 3 * j + 1];
            alfsum += rho7[j] * bf[j] * (hwj * exp_prep(tmj * lnRT));
            rsum += rho7[j] * bf[j];
        }
        HW = (RHORAT - rsum) * alfa0i + alfsum;
        // (brd implies mol <= 7; the select keeps the index in range for the compiler's sake)
        double rho_m = rho7[0];
#pragma unroll
        for (int j = 1; j < MXBRD; j++) rho_m = (mol - 1 == j) ? rho7[j] : rho_m;
        int bf_m = bf[0];
#pragma unroll
        for (int j = 1; j < MXBRD; j++) bf_m = (mol - 1 == j) ? bf[j] : bf_m;
        if (bf_m == 0) HW = HW + rho_m * (hwhmsi - alfa0i);
    }
    const double HWD = Xnu * dopfac;
    if (code == 2) HW = HW * (1 - (AIP * (RP)) - (BIP * (RP2)));
    // which shapes carry the Y factors (modm.f90:706-831): every coupled generic / CO2(-1,-5) line,
    // O2 only for XG = -1
    const bool yfac = code != 0 && ((mol != 7 && mol != 2) || (mol == 7 && code == 1) || (mol == 2 && code != 2));
    LinePhys ph;
    ph.xnu = Xnu;
    ph.hw = HW;
    ph.hwd = HWD;
    ph.stild = STILD;
    ph.c1 = yfac ? AIP * frcp_any(HW) * RP : 0.;
    ph.g = yfac ? BIP * RP2 : 0.;
    return ph;
}

// pre: the line's table fields, read ahead by the caller (lines_kernel: while the previous chunk is evaluated), or null
template <bool IBRD, bool PLAIN = false>
__device__ __forceinline__ LinePhys line_physics(const ModmArgs &a, const DevLines &L, int idx, int m, uint32_t meta, const double *lay,
                                                 const double *scor, const double *dop, const double *sWl,
                                                 const LineFields *pre = nullptr) {
    LayerScalars ly;
    ly.ILC = (int)lay[17];
    ly.RHORAT = lay[0]; ly.RP = lay[1]; ly.RP2 = lay[2]; ly.lnRT = lay[3]; ly.cTk = lay[4]; ly.cT0 = lay[5];
    ly.dTinv = lay[6]; ly.RECTLC = lay[7]; ly.TMPDIF = lay[8];
    const double WTOT = lay[9];
    double rho7[MXBRD];
#pragma unroll
    for (int j = 0; j < MXBRD; j++) rho7[j] = IBRD ? lay[10 + j] : 0.;
    const int mol = m + 1;
    const int iso = (meta >> 6) & 15;
    const double rho_self = (mol <= MXBRD) ? lay[10 + mol - 1] : ly.RHORAT * sWl[mol - 1] / WTOT;
    const double XIPSF = iso ? scor[(mol - 1) * 9 + iso - 1] : 0.;
    const double dopfac = iso ? dop[(mol - 1) * 9 + iso - 1] : dop[(mol - 1) * 9];
    LineFields lf = pre ? *pre : load_line_fields(L, idx);
    lf.meta = meta;
    return line_physics_core<IBRD, PLAIN>(phys_params(a, L), idx, mol, lf, ly, rho_self, rho7, XIPSF, dopfac);
}

// PLAIN: no lane of the wave holds a line with a coupling code (see line_physics_core)
template <typename R, bool PLAIN = false>
__device__ __forceinline__ void line_records(const ModmArgs &a, const DevLines &L, int idx, int m, uint32_t meta, const LinePhys &ph,
                                             const double *sWl, const double *sWn, int TW, typename HotOf<R>::type &outA, HotB &outB,
                                             ColdLine &outC, bool &fAL, bool &fM2, bool &fV, bool &fY, double near_lb = -1.) {
#pragma clang fp contract(off)
    constexpr bool SGL = sizeof(R) == 4;
    const int mol = m + 1, code = PLAIN ? 0 : (int)((meta >> 10) & 3);
    const double Xnu = ph.xnu, HW = ph.hw, HWD = ph.hwd, STILD = ph.stild, c1 = PLAIN ? 0. : ph.c1, g = PLAIN ? 0. : ph.g;
    const bool yfac = code != 0 && ((mol != 7 && mol != 2) || (mol == 7 && code == 1) || (mol == 2 && code != 2));
    // zeta = HW / (HW + HWD) > 0.99 (modm.f90:427) decided without the division unless the quotient is within 1e-12
    // of the threshold, where the reference's own rounded quotient is formed
    const double zsum = HW + HWD, zthr = 0.99 * zsum;
#ifdef MONORTM_ABLATE_ZSEARCH
    const bool zeta_gt = zthr == zthr;  // timing experiment: no Voigt search, no candidates (wrong results)
#else
    const bool zeta_gt = (HW > zthr * (1. + 1e-12)) ? true : ((HW < zthr * (1. - 1e-12)) ? false : (HW / zsum > 0.99));
#endif
    const double A2 = STILD * HW * (1.0 / K_PI);
    const double HW2 = HW * HW;
    const double p = A2 * frcp_any(625. + HW2);
    HotA h;
    HotB hb;
    // single precision: the amplitudes carry the column amount W (keeps them inside the float range)
    const double wsc = SGL ? sWl[mol - 1] : 1.0;
    h.xnu = Xnu;
    h.hw2 = HW2;
    h.a2 = A2 * wsc;
    if (mol == 7) {
        // O2: no pedestal.  Uncoupled lines obey the 25 cm-1 rule inside the shape function and add the
        // negative resonance only when WN+Xnu <= 25; coupled lines use both resonances everywhere
        // (modm.f90:755-792)
        h.pa = code ? __builtin_inf() : 25.;
        hb.pb = code ? __builtin_inf() : 25.;
    } else {
        // generic molecules: pedestal with its coupling factors Y1P / Y2P; CO2: bare pedestal (it is
        // multiplied by (2 - d^2/625) and by Y1 per wavenumber, modm.f90:808-817)
        h.pa = ((mol == 2) ? p : p * ((1. + c1 * 25.) + g)) * wsc;
        hb.pb = (p * ((1. - c1 * 25.) + g)) * wsc;
    }
    hb.c1 = c1;
    hb.gp1 = 1. + g;
    // Voigt is only possible when zeta <= 0.99 AND some wavenumber of the tile lies within 100 Doppler
    // widths of the centre (modm.f90:427): look up the nearest one (sWn is sorted)
    double d100 = -1.0;
    // near_lb: a lower bound of the distance to the nearest wavenumber, where the caller has one (lines_ms_kernel: from a table
    // formed once per launch) - beyond the limit nothing can be a candidate and the search is skipped (a NaN limit searches)
    if (!zeta_gt && !(near_lb > 100. * HWD)) {
        const double lim = 100. * HWD;
        double best = __builtin_inf();
        if (a.dvset != 0.) {
            // a grid V1 + i DVSET (dense grids: 512 wavenumbers per tile): the nearest wavenumber is at the rounded index or beside
            // it - three reads instead of the nine dependent ones of the search (positions past nwn repeat the last wavenumber)
            const int jc = min(max((int)rint((Xnu - sWn[0]) / a.dvset), 0), TW - 1);
            best = fabs(sWn[jc] - Xnu);
            best = fmin(best, fabs(sWn[max(jc - 1, 0)] - Xnu));
            best = fmin(best, fabs(sWn[min(jc + 1, TW - 1)] - Xnu));
        } else {
            int lo = 0, hi = TW;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sWn[mid] < Xnu) lo = mid + 1;
                else hi = mid;
            }
            if (lo < TW) best = fabs(sWn[lo] - Xnu);
            if (lo > 0) best = fmin(best, fabs(sWn[lo - 1] - Xnu));
        }
        if (!(best > lim)) d100 = lim;
    }
#ifdef MONORTM_ABLATE_VOIGT
    d100 = -1.0;  // timing experiment: no Voigt candidates (wrong results)
#endif
    hb.d100 = d100;
    fV = d100 >= 0.;
    // Y factors: every coupled generic / CO2(-1,-5) line, O2 for XG = -1 (see yfac).  A coupled O2 line with XG = -3 / -5
    // has none: its limits are +inf and the ordinary O2 loops take both resonances everywhere (modm.f90:777-792)
    // ... and a line whose brackets the clamp of the fast loops would falsify.  The fast loops of generic molecules form
    // a2 / den - pedestal with the [0, 1] clamp of the FMA (fma_clamp0; max(t, 0) in single precision), which IS the 25 cm-1 test
    // for 0 <= a2 and a peak a2 / HW^2 = S~ / (pi HW) below 1.  No physical line list gets within nine orders of magnitude of
    // the upper bound; a NEGATIVE amplitude (negative strength or column: the bracket is then negative inside the window and
    // positive outside) or a NaN one (NaN column, pressure, temperature: the clamp returns 0 where the reference adds NaN,
    // modm.f90:384,432) takes the general loop with its explicit test and stays exact.
#ifdef MONORTM_NO_CLAMP_GUARD
    fY = yfac;
#else
    {
        const double A2w = A2 * wsc;
        fY = yfac || (mol != 7 && mol != 2 && !(A2w >= 0. && HW2 == HW2 && (SGL || A2 <= 0.25 * HW2)));
    }
#endif
    // negative resonance: WN + Xnu <= 25 (<= +inf for coupled O2) possible for the tile's lowest wavenumber?
    const double cutlim = (mol == 7 && code) ? __builtin_inf() : 25.;
    fM2 = mol != 2 && sWn[0] + Xnu <= cutlim;
    // 25 cm-1 rule (modm.f90:384, :755) passed by the whole tile?  |WN - Xnu| is largest at one of its ends
    fAL = !(fabs(sWn[0] - Xnu) > cutlim) && !(fabs(sWn[TW - 1] - Xnu) > cutlim);
    if constexpr (SGL) outA = HotAf{pair_hi(h.xnu), pair_lo(h.xnu), (float)h.hw2, (float)h.a2, (float)h.pa, (float)hb.pb};
    else outA = h;
    outB = hb;
    ColdLine c;
    c.stild = STILD;
    c.hw = HW;
    c.hwd = HWD;
    c.sdep = L.sdep[idx];
    c.info = (uint32_t)mol | ((uint32_t)code << 6);
    outC = c;
}

// phys: the line's LinePhys formed by physics_kernel for this (profile, layer), or null: form it here
template <typename R, bool IBRD>
__device__ __forceinline__ void prepare_line(const ModmArgs &a, const DevLines &L, int idx, int m, const double *lay,
                                             const double *scor, const double *dop, const double *sWl, const double *sWn, int TW,
                                             const PhysView phys, typename HotOf<R>::type &outA, HotB &outB, ColdLine &outC,
                                             bool &fAL, bool &fM2, bool &fV, bool &fY, const LineFields *pre = nullptr) {
    const uint32_t meta = pre ? pre->meta : L.meta[idx];
    // a wave without a coupled line or an air-width / air-shift conversion among its 64 lines (meta bits 10-11, 13, 14) takes the
    // instantiations without those blocks (line_physics_core): the divergent regions around them cost every wave their scalar
    // instructions whether a lane needs them or not (round 6: 4 % of configs[3] in lines_ms_kernel)
    const bool plain = ((meta >> 10) & 3u) == 0u && ((meta >> 13) & 3u) == 0u;
    if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
        LinePhys ph;
        if (phys.m) {
            const LinePhysM pm = phys.m[idx];
            ph.xnu = pm.xnu; ph.hw = pm.hw; ph.hwd = pm.hwd; ph.stild = pm.stild;
            ph.c1 = 0.;
            ph.g = 0.;
        } else ph = line_physics<IBRD, true>(a, L, idx, m, meta, lay, scor, dop, sWl, pre);
        line_records<R, true>(a, L, idx, m, meta, ph, sWl, sWn, TW, outA, outB, outC, fAL, fM2, fV, fY);
        return;
    }
    LinePhys ph;
    if (phys.m) {
        const LinePhysM pm = phys.m[idx];
        ph.xnu = pm.xnu; ph.hw = pm.hw; ph.hwd = pm.hwd; ph.stild = pm.stild;
        ph.c1 = 0.;
        ph.g = 0.;
        if ((meta >> 10) & 3) { const LinePhysY py = phys.y[idx]; ph.c1 = py.c1; ph.g = py.g; }
    } else ph = line_physics<IBRD>(a, L, idx, m, meta, lay, scor, dop, sWl, pre);
    line_records<R>(a, L, idx, m, meta, ph, sWl, sWn, TW, outA, outB, outC, fAL, fM2, fV, fY);
}

}  // namespace
