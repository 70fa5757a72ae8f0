// continuum_kernel.hip - MT_CKD continuum (CONTNM x 6 passes), TKC cloud liquid and the layer totals of MODM for gfx950
// (reference src/modm.f90:200-247, :264-269; src/contnm.f90:25-1142; src/lblrtm_sub.f90; src/CloudOptProp.f90:29-157).
// See DESIGN.md section 3.2.
#include "lineshape.hpp"
#include <cstring>
#include "tables/monortm_tables.h"

namespace {
using namespace monortm_dev;

// ------------------------------------------------------------------------------------------------
// continuum helpers (device)
// ------------------------------------------------------------------------------------------------
struct AccGrid {
    double V1C, V2C, DVC;
    int NPTC, I1;
};
// grid set-up shared by SL296 / SL260 / FRN296 / FRNCO2 / xn2_r (src/contnm.f90:1441-1459)
__host__ __device__ inline AccGrid acc_grid(double V1ABS, double V2ABS, double V1S, double DVS, int NPTS) {
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

// A "team" works on one continuum pass: the whole workgroup (BLOCK = true: large grids, infrared) or one wave of it
// (microwave-sized grids: the passes of the six molecules run side by side in different waves).  A wave executes its
// LDS instructions in order, so a team of one wave only has to stop the compiler from moving accesses across the point.
template <bool BLOCK>
__device__ __forceinline__ void team_sync() {
    if (BLOCK) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// XINT of the coarse array sC (grid g) accumulated into sAbs[ist..last] on the 1 cm-1 grid; tid / nt: rank and size in the team
__device__ void xint_to_abs(const AccGrid &g, const double *sC, double V1ABS, double DVABS, int NPTABS, double v1ss,
                            double v2ss, double *sAbs, int tid, int nt, int ist_min = 1, int last_max = 1 << 30) {
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
    for (int I = ILO + tid; I <= IHI; I += nt) {
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
    const int tid = threadIdx.x, nt = blockDim.x;  // only used by the infrared branches: the team is the workgroup
    for (int J = tid; J <= g.NPTC + 2 && J < csize; J += nt) {
        double v = 0.;
        if (J >= 1 && J <= g.NPTC) v = f(g.I1 + (J - 1), g.V1C + g.DVC * (double)(J - 1));
        sC[J] = v;
    }
    __syncthreads();
    xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, v1ss, v2ss, sAbs, tid, nt, ist_min, last_max);
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
// R: element type of the REAL arrays (real_kind 8 / 4); all arithmetic is double, the stores round to R
// PAR (only without HIGH): the passes run side by side in the four waves of a 256-thread workgroup - shorter dependency
// chain per workgroup, used when the grid is too small to fill the chip (single profiles); a full grid is served better
// by one pass after the other in one team.
// Q4 (only without HIGH and PAR, 64 threads, large microwave batches): a wave serves FOUR layers, each with a team of 16
// lanes and its own ABSRB / coarse arrays.  The coarse stages of the microwave continua touch ~10 grid points per layer, so
// one layer per wave leaves most lanes idle and 8192 one-wave workgroups need two rounds over the chip; four layers per
// wave run the same instruction stream once for all four and fit in one round.
template <typename R, bool HIGH, bool PAR, bool Q4 = false>
__global__ __launch_bounds__(256, HIGH ? 1 : 4) void finish_kernel(ModmArgs a, DevTables tb, double V1ABS, double V2ABS, int NPTABS,
                                                     int csize) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // !PAR: one team = the workgroup, the six passes one after the other.  PAR (256 threads): wave 0 takes the passes of
    // H2O and O3, wave 1 CO2 and O2, wave 2 N2, wave 3 Rayleigh, each with its own ABSRB / coarse arrays.
    static_assert(!(HIGH && PAR), "the infrared branches synchronise the whole workgroup");
    static_assert(!Q4 || (!HIGH && !PAR), "four layers per wave: microwave instantiation only");
    const int wave = threadIdx.x >> 6, quarter = (threadIdx.x >> 4) & 3;
    // rank / size inside the whole workgroup as far as ONE layer is concerned (the zero fill and the totals)
    const int btid = Q4 ? (threadIdx.x & 15) : (int)threadIdx.x, bnt = Q4 ? 16 : (int)blockDim.x;
    const int tid = PAR ? ((int)threadIdx.x & 63) : btid, nt = PAR ? 64 : bnt;
    double *sAbs = smem + ((PAR ? wave : (Q4 ? quarter : 0)) * (NPTABS + 4 + csize));  // 1-based, [0..NPTABS+3]
    double *sC = sAbs + NPTABS + 4;                                  // 1-based coarse array
    const int lay = Q4 ? (int)blockIdx.x * 4 + quarter : (int)blockIdx.x, prof = blockIdx.y;
    if (Q4 && lay >= a.nlay_max) return;  // last workgroup of a layer count that is not a multiple of four
    const int nwn = a.nwn, nmol = a.nmol;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    R *O = wp<R>(a.O) + pl * (size_t)nwn, *OCLW = wp<R>(a.O_CLW) + pl * (size_t)nwn;
    R *OC = wp<R>(a.OC) + pl * MONORTM_NCONT * (size_t)nwn;
    if (lay >= a.nlay[prof]) {
        for (int iw = btid; iw < nwn; iw += bnt) {
            O[iw] = (R)0;
            OCLW[iw] = (R)0;
            for (int s = 0; s < MONORTM_NCONT; s++) OC[(size_t)s * nwn + iw] = (R)0;
            for (int m = 0; m < nmol; m++) wp<R>(a.O_BY_MOL)[(pl * nmol + m) * (size_t)nwn + iw] = (R)0;
        }
        return;
    }
    const double DVABS = 1.0;
    const double PAVE = rp<R>(a.P)[pl], TAVE = rp<R>(a.T)[pl], WBROAD = rp<R>(a.WBRODL)[pl], CLW = rp<R>(a.CLW)[pl];
    const R *wk = rp<R>(a.WKL) + pl * nmol;
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

    for (int pi = 0; pi < (PAR ? 2 : 6); pi++) {
        const int pass = !PAR ? pi : (pi == 0 ? (wave < 2 ? wave : wave + 2) : (wave < 2 ? wave + 2 : 6));
        if (pass > 5) break;
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
        team_sync<!PAR && !Q4>();
        if (pass == 0 && V2 > -20.0 && V1 < 20000. && xself > 0.) {  // H2O self, contnm.f90:325-371
            const double Rself = h2o_fac * RHOAVE * 1.e-20 * xself;
            const AccGrid g = acc_grid(V1ABS, V2ABS, MT_SELF296_V1, MT_SELF296_DV, MT_SELF296_NPT);
            const double TFAC = (TAVE - T0c) * (1. / (260. - T0c));
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
            team_sync<!PAR && !Q4>();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_SELF296_V1, MT_SELF296_V2, sAbs, tid, nt);
            team_sync<!PAR && !Q4>();
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
            team_sync<!PAR && !Q4>();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_FRGN296_V1, MT_FRGN296_V2, sAbs, tid, nt);
            team_sync<!PAR && !Q4>();
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
            team_sync<!PAR && !Q4>();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_FCO2_V1, MT_FCO2_V2, sAbs, tid, nt);
            team_sync<!PAR && !Q4>();
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
            team_sync<!PAR && !Q4>();
            xint_to_abs(g, sC, V1ABS, DVABS, NPTABS, MT_N2RT296_V1, MT_N2RT296_V2, sAbs, tid, nt);
            team_sync<!PAR && !Q4>();
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
            team_sync<!PAR && !Q4>();
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
            if (pass < 5) OC[(size_t)pass * nwn + iw] = (R)(val * radfn(wnv, XKT));
            else O[iw] = (R)(val * wnv / 1.0e4);  // oc_rayl parked in O until the totals below
        }
        team_sync<!PAR && !Q4>();
    }
    // cloud liquid water + totals (modm.f90:264-269) by the whole workgroup, after every team has stored its continua
    __syncthreads();
    R *obm = wp<R>(a.O_BY_MOL) + pl * nmol * (size_t)nwn;
    if (a.nslice > 1 && !a.slices_reduced) {  // add the line slices in slice (= line) order; one thread per (molecule, wavenumber)
        const size_t sstride = (size_t)a.nprof * a.nlay_max * nmol * nwn;
        const R *part = rp<R>(a.partial) + pl * nmol * (size_t)nwn;
        for (int idx = btid; idx < nmol * nwn; idx += bnt) {
            double acc = 0.;
            for (int sl = 0; sl < a.nslice; sl++) acc += (double)part[(size_t)sl * sstride + idx];
            obm[idx] = (R)acc;
        }
        __syncthreads();  // the totals below read every molecule of a wavenumber
    }
    for (int iw = btid; iw < nwn; iw += bnt) {
        const double wnv = a.wn[iw];
        const double oclw = (CLW == 0.) ? 0. : odclw_tkc(wnv, TAVE, CLW);  // alpha * 0 = 0 in the reference
        OCLW[iw] = (R)oclw;
        double o = 0.;
        for (int m = 0; m < nmol; m++) o = o + (double)obm[(size_t)m * nwn + iw];
        double soc = 0.;
        for (int s = 0; s < MONORTM_NCONT; s++) soc += (double)OC[(size_t)s * nwn + iw];
        const double odx = a.ODXSEC ? (double)rp<R>(a.ODXSEC)[pl * (size_t)nwn + iw] : 0.;  // cross-section molecules (modm.f90:197, :268)
        o = o + odx + (double)O[iw] + soc + oclw;
        O[iw] = (R)o;
    }
}

// ------------------------------------------------------------------------------------------------
// finish_mw_kernel: the same work as finish_kernel for spectral ranges that end below 820 cm-1 (microwave to far infrared),
// where only H2O self / foreign, CO2 and the N2 roto-translational continuum are alive (O3, O2 and Rayleigh have no
// contribution there: their slots are zero; the CO2 band-head temperature factor and the CO2 chi factor sit above 2000
// cm-1).  finish_kernel walks the six oneMolecCntnm passes one after the other, each a chain of coarse array -> XINT
// onto ABSRB -> XINT onto the wavenumbers with ~10 busy lanes in the coarse stages; with one wave per layer that chain of
// dependent table loads, log() and exp() is pure latency (measured with s_memtime: 17 us per wave of which 12 in the set-up
// and the coarse stage).  Here the four branches run SIDE BY SIDE:
//   A  every coarse point of the four branches, flattened over the lanes: ONE round of table loads and two exp(y log x)
//      for all of them (no divergence: a lane picks its tables and its formula by selects)
//   B  every (branch, ABSRB point): 4-point XINT, contnm.f90:1146-1164
//   C  every wavenumber of this workgroup's chunk: second XINT of the three passes (self + foreign added in CONTNM's
//      order) x RADFN -> OC
//   D  cloud liquid water, line-slice sums and the totals of modm.f90:264-269
// The grids of the branches depend on the spectral range only: the launcher prepares them on the host (MwSetup, the same
// acc_grid / pre_xint arithmetic on small integers).  Same formulas and the same order of additions per value as
// finish_kernel (src/contnm.f90:325-528, :906-943; modm.f90:207-247).
// grid = (layers, profiles, wavenumber chunks of blockDim.x); dynamic LDS = MwSetup::lds doubles.
// ------------------------------------------------------------------------------------------------
struct MwSetup {
    double V1C[4], DVC[4], RDVC[4];     // coarse grids of self, foreign, CO2, N2 (acc_grid); 1 / DVC (the host's IEEE quotient)
    int NPTC[4], I1[4], NPT[4];         // points on the coarse grid, first table index, table length
    int ILO[4], IHI[4];                 // ABSRB points each branch reaches (pre_xint + XINT's window)
    int off[5];                         // offsets of the coarse arrays in LDS (flattened item index of stage A)
    int alive[4];                       // the spectral-range tests of contnm.f90 (the scale factors are tested in the kernel)
    double V1, V2;                      // first and last wavenumber
    int lds;                            // doubles of dynamic LDS
};
template <typename T>
__device__ __forceinline__ T sel4(int b, T x0, T x1, T x2, T x3) { return b < 2 ? (b == 0 ? x0 : x1) : (b == 2 ? x2 : x3); }

__global__ void logratio_kernel(const double *t296, const double *tlow, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log(tlow[i] / t296[i]);  // the log(x) of powpos(x, y) = exp(y log x)
}

// Everything in stages A and B that depends on the spectral range alone - which branch and table entry an item is, the table
// values themselves, the foreign scale factor FSCAL, the 4-point weights and LDS index of an ABSRB point - is formed once per
// range by mw_items_kernel (MwCache, launch_finish_mw) and read back as one coalesced record per item: finish_mw_kernel
// keeps the layer-dependent arithmetic only (round 4: no select chains, no dependent table loads in stages A / B).
struct MwItemA { double ta, tc, te, tf; int flags, pad_; };  // flags: branch (bits 0-1), inside the coarse grid (2), inside the table (3)
struct MwItemB { double B1, c1, c2, B2; int j, flags; };    // j: LDS index of A[J-1]; flags: branch (0-1), inside the branch's reach (2)

__global__ void mw_items_kernel(DevTables tb, MwSetup q, double V1ABS, int NPTABS, MwItemA *ia, MwItemB *ib) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    const int NA = NPTABS + 4;
    const double DVABS = 1.0;
    if (it < q.off[4]) {
        const int b = (it >= q.off[1]) + (it >= q.off[2]) + (it >= q.off[3]);
        const int J = it - q.off[b];
        const int I = q.I1[b] + (J - 1);
        const bool live = q.alive[b] && J >= 1 && J <= q.NPTC[b];
        const bool intab = I >= 1 && I <= q.NPT[b];
        const double VJ = q.V1C[b] + q.DVC[b] * (double)(J - 1);
        const int ix = (live && intab) ? I - 1 : 0;
        int jf = 0;
        if (b == 1 && VJ <= 600.) jf = max((int)((VJ + 10.) / 10. + 0.00001) + 1, 0);
        MwItemA r;
        r.ta = sel4(b, tb.self296, tb.frgn296, tb.fco2, tb.n2c296)[ix];
        r.tc = sel4(b, tb.lr_self, tb.xfac_rhu, tb.fco2, tb.lr_n2c)[b == 1 ? jf : ix];
        const int ix3 = (b == 3) ? ix : 0;
        r.te = tb.n2sf296[ix3], r.tf = tb.lr_n2sf[ix3];
        if (b == 1 && VJ > 600.) {  // contnm.f90:380-474: the foreign scale factor above 600 cm-1 is a closed form of VJ
            const double f0 = 0.06, V0F1 = 255.67, HWSQ1 = 240. * 240., BETA1 = 57.83, C_1 = -0.42, C_2 = 0.3, BETA2 = 630.;
            const double vdelsq1 = (VJ - V0F1) * (VJ - V0F1), vdelmsq1 = (VJ + V0F1) * (VJ + V0F1);
            double t = (VJ - V0F1) / BETA1; t = t * t; t = t * t; const double VF1 = t * t;
            t = (VJ + V0F1) / BETA1; t = t * t; t = t * t; const double VmF1 = t * t;
            t = VJ / BETA2; t = t * t; t = t * t; const double VF2 = t * t;
            r.tc = 1. + (f0 + C_1 * ((HWSQ1 / (vdelsq1 + HWSQ1 + VF1)) + (HWSQ1 / (vdelmsq1 + HWSQ1 + VmF1)))) / (1. + C_2 * VF2);
        }
        r.flags = b | (live ? 4 : 0) | (intab ? 8 : 0);
        r.pad_ = 0;
        ia[it] = r;
    }
    if (it < 4 * NA) {
        const int b = it / NA, I = it - b * NA;
        const bool inside = q.alive[b] && I >= q.ILO[b] && I <= q.IHI[b];
        MwItemB r = {0., 0., 0., 0., 0, b};
        if (inside) {  // xint_point() with RECDVA = 1 / DVA from the host
            const double VI = V1ABS + DVABS * (double)(I - 1);
            const double V1A = q.V1C[b], DVA = q.DVC[b], RECDVA = q.RDVC[b];
            const int J = (int)((VI - V1A) * RECDVA + K_ONEPL);
            const double VJ = V1A + DVA * (double)(J - 1);
            const double P = RECDVA * (VI - VJ);
            const double C = (3. - 2. * P) * P * P;
            const double B = 0.5 * P * (1. - P);
            const double B1 = B * (1. - P), B2 = B * P;
            r.B1 = B1, r.c1 = (1. - C + B2), r.c2 = (C + B1), r.B2 = B2;
            r.j = q.off[b] + J - 1;
            r.flags = b | 4;
        }
        ib[it] = r;
    }
}

template <typename R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void finish_mw_kernel(ModmArgs a, MwSetup q, const MwItemA *__restrict__ ia, const MwItemB *__restrict__ ib,
                                                       double V1ABS, double V2ABS, int NPTABS) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lay = blockIdx.x, prof = blockIdx.y, iw0 = blockIdx.z * nt;
    const int nwn = a.nwn, nmol = a.nmol;
    const int iw = iw0 + tid;  // this thread's wavenumber in stages C / D (one per thread)
    const size_t pl = (size_t)prof * a.nlay_max + lay;
#ifdef MW_TIMING
    long long tq[8]; int ntq = 0;
#define MW_T() tq[ntq++] = (long long)__builtin_readcyclecounter()
    MW_T();
#else
#define MW_T()
#endif
    R *O = wp<R>(a.O) + pl * (size_t)nwn, *OCLW = wp<R>(a.O_CLW) + pl * (size_t)nwn;
    R *OC = wp<R>(a.OC) + pl * MONORTM_NCONT * (size_t)nwn;
    R *obm = wp<R>(a.O_BY_MOL) + pl * nmol * (size_t)nwn;
    // every load of the set-up is issued before the first use: one round trip to memory instead of four
    const int nl = a.nlay[prof];
    const double PAVE = rp<R>(a.P)[pl], TAVE = rp<R>(a.T)[pl], WBROAD = rp<R>(a.WBRODL)[pl], CLW = rp<R>(a.CLW)[pl];
    const R *wk = rp<R>(a.WKL) + pl * nmol;
    double WTOT = WBROAD;
    for (int m = 0; m < nmol; m++) WTOT = WTOT + wk[m];
    const double WK1 = wk[0], WK2 = wk[1], WK7 = wk[6];
    const bool sliced = a.nslice > 1 && !a.slices_reduced;
    double o_lines = 0.;  // sum of the molecules' line optical depths of this wavenumber (first terms of modm.f90:264-269)
    if (!sliced && iw < nwn) {
        if (a.osum) o_lines = a.osum[pl * (size_t)nwn + iw];  // formed by lines_kernel in the same order
        else
            for (int m = 0; m < nmol; m++) o_lines = o_lines + (double)obm[(size_t)m * nwn + iw];
    }
    if (lay >= nl) {
        if (iw < nwn) {
            O[iw] = (R)0;
            OCLW[iw] = (R)0;
            for (int sl = 0; sl < MONORTM_NCONT; sl++) OC[(size_t)sl * nwn + iw] = (R)0;
            for (int m = 0; m < nmol; m++) obm[(size_t)m * nwn + iw] = (R)0;
        }
        return;
    }
    if (sliced) {  // partial sums of the line slices, in slice (= line) order; all threads of the workgroup share the work
        const int cw = min(nt, nwn - iw0);
        const size_t sstride = (size_t)a.nprof * a.nlay_max * nmol * nwn;
        const R *part = rp<R>(a.partial) + pl * nmol * (size_t)nwn;
        for (int it = tid; it < nmol * cw; it += nt) {
            const size_t at = (size_t)(it / cw) * nwn + iw0 + (it % cw);
            double acc = 0.;
            for (int sl = 0; sl < a.nslice; sl++) acc += (double)part[(size_t)sl * sstride + at];
            obm[at] = (R)acc;  // read back in stage D, two barriers later
        }
    }
    const double DVABS = 1.0;
    const double V1 = q.V1, V2 = q.V2;
    const double P0c = 1013., T0c = 296., XLOSMT = 2.68675E+19;
    // (round 4: the layer scalars through reciprocals - v_rcp_f64 + two Newton steps, within an ulp of the IEEE quotients that
    // finish_kernel forms - and the exponentials through exp_cw: 1250 -> ~1050 instructions per wave)
    const double rTAVE = rcp2(TAVE), rWTOT = rcp2(WTOT), pr = PAVE * (1. / P0c);
    const double RHOAVE = pr * (T0c * rTAVE);
    const double XKT = TAVE * (1. / K_RADCN2);
    const double amagat = pr * (273. * rTAVE);
    const double x_vmr_h2o = WK1 * rWTOT, x_vmr_o2 = WK7 * rWTOT, x_vmr_n2 = 1. - x_vmr_h2o - x_vmr_o2;
    const double wn2 = x_vmr_n2 * WTOT;
    const double h2o_fac = x_vmr_h2o;
    const double xself = a.cntnm[0], xfrgn = a.cntnm[1], xco2c = a.cntnm[2], xn2cn = a.cntnm[5];
    const bool on0 = q.alive[0] && xself > 0., on1 = q.alive[1] && xfrgn > 0., on2 = q.alive[2] && xco2c > 0.,
               on3 = q.alive[3] && xn2cn > 0.;
    // LDS: coarse arrays of the four branches (q.off), then their four ABSRB grids [NPTABS + 4], 1-based
    double *sC = smem;
    double *sAbs = smem + q.off[4];
    const int NA = NPTABS + 4;

    MW_T();
    // ---- stage A: coarse coefficients ------------------------------------------------------------
    {
        const double Rself = h2o_fac * RHOAVE * 1.e-20 * xself;            // contnm.f90:325-371
        const double TFAC = (TAVE - T0c) * (1. / (260. - T0c));
        const double Rfrgn = (1. - h2o_fac) * RHOAVE * 1.e-20 * xfrgn;     // contnm.f90:380-474
        const double WCO2 = WK2 * RHOAVE * 1.0E-20 * xco2c;                // contnm.f90:484-528 + FRNCO2 :2958
        const double tau_fac = xn2cn * (wn2 / XLOSMT) * amagat;            // contnm.f90:906-943
        const double tfac = (TAVE - 296.) * (1. / (220. - 296.));
        const int onmask = (on0 ? 1 : 0) | (on1 ? 2 : 0) | (on2 ? 4 : 0) | (on3 ? 8 : 0);
        for (int it = tid; it < q.off[4]; it += nt) {
            const MwItemA r = ia[it];
            const int b = r.flags & 3;
            const bool live = ((r.flags >> 2) & 1) && ((onmask >> b) & 1), intab = (r.flags >> 3) & 1;
            const double ta = r.ta, tc = r.tc, te = r.te, tf = r.tf;
            // the temperature interpolations powpos(x, y) = exp(y log x) with the tabulated log x: self (260 K / 296 K), N2
            // (220 K / 296 K) and its scale factor
            const double pw1 = exp_cw((b == 0 ? TFAC : tfac) * ((b == 0 || b == 3) ? tc : 0.));
            const double pw2 = exp_cw(tfac * (b == 3 ? tf : 0.));
            double v = 0.;
            if (live) {
                if (b == 0) {
                    if (intab) {
                        double SH2O = 0.;
                        if (ta > 0.) SH2O = ta * pw1;
                        v = WK1 * (SH2O * Rself);
                    }
                } else if (b == 1) {
                    double FH2O = intab ? ta : 0.;
                    const double FSCAL = tc;  // xfac_rhu below 600 cm-1, the closed form above (mw_items_kernel)
                    FH2O = FH2O * FSCAL;
                    v = (WK1 * FH2O) * Rfrgn;
                } else if (b == 2) {
                    if (intab) v = ta * WCO2;  // (band-head temperature factor and chi factor are 1 below 2000 cm-1)
                } else {
                    double c0 = 0., c1 = 0.;
                    if (intab) {
                        c0 = ta * pw1;
                        const double sf_T = te * pw2;
                        c1 = (sf_T - 1.) * (0.79) / (0.21);
                    }
                    v = tau_fac * c0 * (x_vmr_n2 + c1 * x_vmr_o2 + 1. * x_vmr_h2o);
                }
            }
            sC[it] = v;
        }
    }
    __syncthreads();

    MW_T();
    // ---- stage B: XINT of every branch onto its ABSRB grid -----------------------------------------
    const int onmaskB = (on0 ? 1 : 0) | (on1 ? 2 : 0) | (on2 ? 4 : 0) | (on3 ? 8 : 0);
    for (int it = tid; it < 4 * NA; it += nt) {
        const MwItemB r = ib[it];
        const bool inside = ((r.flags >> 2) & 1) && ((onmaskB >> (r.flags & 3)) & 1);
        double v = 0.;
        if (inside) {
            const double *A = sC + r.j;
            v = (-A[0] * r.B1 + A[1] * r.c1 + A[2] * r.c2 - A[3] * r.B2) * 1.0;
        }
        sAbs[it] = v;
    }
    __syncthreads();

    MW_T();
    // ---- stage C: second interpolation ABSRB -> wavenumbers (modm.f90:216-246) x RADFN -> OC -------------------
    double soc = 0.;  // sum of the continuum slots as stored (rounded to R)
    if (iw < nwn) {
        const double wnv = a.wn[iw];
        bool inside;
        double vint;
        if (a.dvset != 0.) {
            const int I = iw + 1;
            int ilo = (int)((V1ABS + DVABS - V1) / a.dvset + 1. + K_ONEMI);
            if (ilo < 1) ilo = 1;
            int ihi = (int)((V2ABS - DVABS - V1) / a.dvset + K_ONEMI);
            if (ihi > nwn) ihi = nwn;
            inside = I >= ilo && I <= ihi;
            vint = V1 + a.dvset * (double)(I - 1);
        } else {
            int ilo = (int)((V1ABS + DVABS - wnv) / 1.0 + 1. + K_ONEMI);
            if (ilo < 1) ilo = 1;
            int ihi = (int)((V2ABS - DVABS - wnv) / 1.0 + K_ONEMI);
            if (ihi > 1) ihi = 1;
            inside = ilo <= 1 && ihi >= 1;
            vint = wnv;
        }
        double rad = wnv;  // RADFN (src/lblrtm_sub.f90:36-97) with XKT > 0
        {
            const double x = (wnv * K_RADCN2) * rTAVE;
            if (x <= 0.01) rad = 0.5 * x * wnv;
            else if (x <= 10.0) {
                const double e = exp_cw(-x);
                rad = wnv * (1. - e) * rcp2(1. + e);
            }
        }
        // the 4-point weights of XINT are the same for the three passes (same grid): xint_point on the sum of self and
        // foreign needs ABSRB = (0 + self) + foreign per point, as CONTNM accumulates it
        const double RECDVA = 1. / DVABS;
        const int J = (int)((vint - V1ABS) * RECDVA + K_ONEPL);
        const double VJ = V1ABS + DVABS * (double)(J - 1);
        const double P = RECDVA * (vint - VJ);
        const double C = (3. - 2. * P) * P * P;
        const double B = 0.5 * P * (1. - P);
        const double B1 = B * (1. - P), B2 = B * P;
        double val[3] = {0., 0., 0.};
        if (inside) {
            const double *A0 = sAbs, *A1 = sAbs + NA, *A2 = sAbs + 2 * NA, *A3 = sAbs + 3 * NA;
            const double h0 = A0[J - 1] + A1[J - 1], h1 = A0[J] + A1[J], h2 = A0[J + 1] + A1[J + 1], h3 = A0[J + 2] + A1[J + 2];
            val[0] = -h0 * B1 + h1 * (1. - C + B2) + h2 * (C + B1) - h3 * B2;
            val[1] = -A2[J - 1] * B1 + A2[J] * (1. - C + B2) + A2[J + 1] * (C + B1) - A2[J + 2] * B2;
            val[2] = -A3[J - 1] * B1 + A3[J] * (1. - C + B2) + A3[J + 1] * (C + B1) - A3[J + 2] * B2;
        }
        const R s0 = (on0 || on1) ? (R)(val[0] * rad) : (R)0;
        const R s1 = on2 ? (R)(val[1] * rad) : (R)0;
        const R s4 = on3 ? (R)(val[2] * rad) : (R)0;
        OC[iw] = s0;
        OC[(size_t)nwn + iw] = s1;
        OC[(size_t)2 * nwn + iw] = (R)0;
        OC[(size_t)3 * nwn + iw] = (R)0;
        OC[(size_t)4 * nwn + iw] = s4;
        soc += (double)s0;
        soc += (double)s1;
        soc += 0.;
        soc += 0.;
        soc += (double)s4;
    }

    MW_T();
    // ---- stage D: cloud liquid water and the totals (modm.f90:264-269) ---------------------------------------
    if (iw < nwn) {
        if (sliced)
            for (int m = 0; m < nmol; m++) o_lines = o_lines + (double)obm[(size_t)m * nwn + iw];
        const double oclw = (CLW == 0.) ? 0. : odclw_tkc(a.wn[iw], TAVE, CLW);
        OCLW[iw] = (R)oclw;
        const double odx = a.ODXSEC ? (double)rp<R>(a.ODXSEC)[pl * (size_t)nwn + iw] : 0.;  // cross-section molecules (modm.f90:197, :268)
        const double o = o_lines + odx + 0. + soc + oclw;  // (the Rayleigh term of modm.f90:243-245 is zero below 820 cm-1)
        O[iw] = (R)o;
    }
#ifdef LINES_TIMING
    if (a.osum && tid < 12 && blockIdx.z == 0) OCLW[8 + tid] = (R)a.osum[pl * (size_t)nwn + tid];
#endif
#ifdef MW_TIMING
    MW_T();
    if (tid == 0) for (int i = 1; i < ntq; i++) OCLW[i - 1] = (R)(double)(tq[i] - tq[i - 1]);
    if (tid == 0) OCLW[ntq - 1] = (R)(double)(tq[ntq - 1] - tq[0]);
#endif
}

// Wide grids (nmol x nwn large): the slice sums as a bandwidth-bound kernel of their own, ahead of finish_kernel
// grid = (blocks of 256 over nmol*nwn, layers, profiles)
template <typename R>
__global__ __launch_bounds__(256) void reduce_slices_kernel(ModmArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x, lay = blockIdx.y, prof = blockIdx.z;
    const int n = a.nmol * a.nwn;
    if (idx >= n || lay >= a.nlay[prof]) return;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    const size_t sstride = (size_t)a.nprof * a.nlay_max * n;
    const R *part = rp<R>(a.partial) + pl * n;
    double acc = 0.;
    for (int sl = 0; sl < a.nslice; sl++) acc += (double)part[(size_t)sl * sstride + idx];
    wp<R>(a.O_BY_MOL)[pl * n + idx] = (R)acc;
}

// Known-answer hook (tests only reach it through monortm_hip_kat): the device versions of the small functions of the path,
// one evaluation per thread.  in: n x 4 arguments, out: n x 2 - the same convention as the CPU restatement's orc_kat.
__global__ void kat_kernel(int which, int n, const double *in, const double *tab, double *out, int *errflag, DevTables tips) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *a = in + 4 * i;
    double r0 = 0., r1 = 0.;
    if (which == 1) { const cx z = w4(a[0], a[1]); r0 = z.re; r1 = z.im; }
    else if (which == 2) { const cx z = sd_humlicek(a[0], a[1], a[2], a[3]); r0 = z.re; r1 = z.im; }
    else if (which == 3) r0 = sdvoigt(a[0], a[1], a[2], a[3], errflag);
    else if (which == 4) r0 = radfn(a[0], a[1]);
    else if (which == 5) r0 = tips_atob(a[0], tab);
    else if (which == 6) r0 = odclw_tkc(a[0], a[1], a[2]);
    else if (which == 7) {  // scor(mol, iso) of TIPS_2003(39, T, scor): tab = the context's TIPS tables (see monortm_hip_kat)
        const int mol = (int)a[1], iso = (int)a[2];
        bool bad = a[0] < 70. || a[0] > 3000.;
        if (!bad && mol >= 1 && mol <= MXMOL && iso >= 1 && iso <= 9) r0 = tips_scor(tips.tips_isonm, tips.tips_offset, tips.tips_qoft, tips.tips_q296, mol, iso, a[0], &bad);
        r1 = bad ? 1. : 0.;
        if (bad) r0 = 0.;
    }
    else if (which == 8) {  // HALFWHM_D(mol, iso, xnu, T): the kernels' Doppler factor x wavenumber; tips.smass = the context's mass table
        const int mol = (int)a[0], iso = (int)a[1];
        if (mol >= 1 && mol <= MXMOL && iso >= 1 && iso <= 9) r0 = a[2] * doppler_factor(tips.smass[(mol - 1) * 9 + iso - 1], a[3]);
    } else if (which == 9) r0 = planck(K_RADCN1 * (a[0] * a[0] * a[0]), a[0], a[1]);  // bb_fn(v, fbeta) as rtm_kernel forms it
    out[2 * i] = r0;
    out[2 * i + 1] = r1;
}

}  // namespace

namespace monortm_dev {
void launch_kat(int which, int n, const double *in, const double *tab, double *out, int *errflag, const DevTables &tb, hipStream_t s) {
    hipLaunchKernelGGL(kat_kernel, dim3((n + 63) / 64), dim3(64), 0, s, which, n, in, tab, out, errflag, tb);
}
void launch_logratio(const double *t296, const double *tlow, double *out, int n, hipStream_t s) {
    hipLaunchKernelGGL(logratio_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t296, tlow, out, n);
}
hipError_t launch_finish_mw(const ModmArgs &a, const DevTables &tb, double V1, double V2, double V1ABS, double V2ABS, int NPTABS,
                            MwCache &cache, hipStream_t s) {
    MwSetup q;
    const double v1s[4] = {MT_SELF296_V1, MT_FRGN296_V1, MT_FCO2_V1, MT_N2RT296_V1};
    const double v2s[4] = {MT_SELF296_V2, MT_FRGN296_V2, MT_FCO2_V2, MT_N2RT296_V2};
    const double dvs[4] = {MT_SELF296_DV, MT_FRGN296_DV, MT_FCO2_DV, MT_N2RT296_DV};
    const int npt[4] = {MT_SELF296_NPT, MT_FRGN296_NPT, MT_FCO2_NPT, MT_N2RT296_NPT};
    const double DVABS = 1.0;
    q.off[0] = 0;
    for (int b = 0; b < 4; b++) {
        const AccGrid g = acc_grid(V1ABS, V2ABS, v1s[b], dvs[b], npt[b]);
        q.V1C[b] = g.V1C, q.DVC[b] = g.DVC, q.RDVC[b] = 1. / g.DVC, q.NPTC[b] = g.NPTC, q.I1[b] = g.I1, q.NPT[b] = npt[b];
        // pre_xint (contnm.f90:1146-1164) and the index window of XINT (lblrtm_sub.f90:14-21)
        int ist = (int)(2 + (v1s[b] - V1ABS) / DVABS + 1.e-5);
        if (ist < 1) ist = 1;
        int last = (int)(1 + (v2s[b] - V1ABS) / DVABS + 1.e-5);
        if (last > NPTABS) last = NPTABS;
        q.ILO[b] = (int)((g.V1C + g.DVC - V1ABS) / DVABS + 1. + K_ONEMI);
        if (q.ILO[b] < ist) q.ILO[b] = ist;
        q.IHI[b] = (int)((g.V2C - g.DVC - V1ABS) / DVABS + K_ONEMI);
        if (q.IHI[b] > last) q.IHI[b] = last;
        // stage A fills entries 0 .. NPTC + 2 (XINT reads one point before and two after an interval)
        q.off[b + 1] = q.off[b] + (g.NPTC > 0 ? g.NPTC : 0) + 4;
    }
    q.alive[0] = V2 > -20.0 && V1 < 20000.;
    q.alive[1] = q.alive[0];
    q.alive[2] = V2 > -20.0 && V1 < 10000.;
    q.alive[3] = V2 > -10.0 && V1 < 350.;
    q.V1 = V1, q.V2 = V2;
    q.lds = q.off[4] + 4 * (NPTABS + 4);
    // one wave per layer when the wavenumbers fit; partial sums of line slices are shared by four waves
    const bool sliced = a.nslice > 1 && !a.slices_reduced;
    const int threads = (a.nwn <= 64 && !sliced) ? 64 : ((a.nwn <= 128 && !sliced) ? 128 : 256);
    const dim3 grid(a.nlay_max, a.nprof, (a.nwn + threads - 1) / threads);
    const size_t lds = sizeof(double) * (size_t)q.lds;
    if (lds > 60000) return hipErrorInvalidValue;
    // the per-item constants of this spectral range: built by a kernel on this stream when the context meets the range for
    // the first time, kept per range afterwards (MwCache).  Building allocates, so it cannot happen inside a stream capture:
    // the warm step that precedes a capture fills the cache.
    const int nA = q.off[4], nB = 4 * (NPTABS + 4);
    const double key[5] = {V1, V2, V1ABS, V2ABS, (double)NPTABS};
    cache.why = nullptr;
    MwCache::Entry *en = nullptr;
    for (int i = 0; i < cache.n; i++)
        if (memcmp(cache.e[i].key, key, sizeof(key)) == 0) en = &cache.e[i];
    if (!en) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap != hipStreamCaptureStatusNone) {
            cache.why = "a spectral range this context has not served yet cannot be set up inside a stream capture: run one step outside the capture first";
            return hipErrorStreamCaptureUnsupported;
        }
        if (cache.n == MwCache::kMax) {  // evict the range used longest ago (rare: the device is drained so that nobody reads its items)
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            int lru = 0;
            for (int i = 1; i < cache.n; i++)
                if (cache.e[i].last_use < cache.e[lru].last_use) lru = i;
            (void)hipFree(cache.e[lru].items);
            (void)hipEventDestroy(cache.e[lru].built);
            cache.e[lru] = cache.e[cache.n - 1];
            cache.n--;
        }
        MwCache::Entry ne{};
        memcpy(ne.key, key, sizeof(key));
        hipError_t e = hipMalloc(&ne.items, sizeof(MwItemA) * (size_t)nA + sizeof(MwItemB) * (size_t)nB);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&ne.built, hipEventDisableTiming);
        if (e != hipSuccess) { (void)hipFree(ne.items); return e; }
        MwItemA *ia = static_cast<MwItemA *>(ne.items);
        MwItemB *ib = reinterpret_cast<MwItemB *>(ia + nA);
        const int n = nA > nB ? nA : nB;
        hipLaunchKernelGGL(mw_items_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tb, q, V1ABS, NPTABS, ia, ib);
        e = hipEventRecord(ne.built, s);
        if (e != hipSuccess) { (void)hipFree(ne.items); (void)hipEventDestroy(ne.built); return e; }
        ne.stream = s;
        ne.ready = false;
        cache.e[cache.n] = ne;
        en = &cache.e[cache.n++];
    } else if (!en->ready) {
        // the items may still be in the making: wait for them here (legal inside a capture too), and stop waiting once they are seen
        // done.  On WHICHEVER stream the call comes - comparing stream handles would let a new stream that reuses the builder's
        // handle skip the wait, and calls on the builder's own stream are what sets `ready` in a one-stream program
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap == hipStreamCaptureStatusNone && hipEventQuery(en->built) == hipSuccess) en->ready = true;
        else {
            (void)hipGetLastError();  // (hipErrorNotReady of the query is not an error of this call)
            hipError_t e = hipStreamWaitEvent(s, en->built, 0);
            if (e != hipSuccess) return e;
        }
    }
    en->last_use = ++cache.clock;
    const MwItemA *ia = static_cast<const MwItemA *>(en->items);
    const MwItemB *ib = reinterpret_cast<const MwItemB *>(ia + nA);
    if (a.real_kind == 4) hipLaunchKernelGGL(finish_mw_kernel<float>, grid, dim3(threads), lds, s, a, q, ia, ib, V1ABS, V2ABS, NPTABS);
    else hipLaunchKernelGGL(finish_mw_kernel<double>, grid, dim3(threads), lds, s, a, q, ia, ib, V1ABS, V2ABS, NPTABS);
    return hipGetLastError();
}
void MwCache::release() {
    for (int i = 0; i < n; i++) {
        (void)hipFree(e[i].items);
        (void)hipEventDestroy(e[i].built);
    }
    n = 0;
}
void launch_reduce_slices(const ModmArgs &a, hipStream_t s) {
    const dim3 grid((a.nmol * a.nwn + 255) / 256, a.nlay_max, a.nprof);
    if (a.real_kind == 4) hipLaunchKernelGGL(reduce_slices_kernel<float>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(reduce_slices_kernel<double>, grid, dim3(256), 0, s, a);
}
template <typename R, bool HIGH, bool PAR, bool Q4 = false>
static hipError_t launch_finish_t(const ModmArgs &a, const DevTables &tb, double V1ABS, double V2ABS, int NPTABS, int csize,
                                  int threads, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(finish_kernel<R, HIGH, PAR, Q4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((finish_kernel<R, HIGH, PAR, Q4>), dim3(Q4 ? (a.nlay_max + 3) / 4 : a.nlay_max, a.nprof), dim3(threads), lds, s,
                       a, tb, V1ABS, V2ABS, NPTABS, csize);
    return hipSuccess;
}
template <typename R>
static hipError_t launch_finish_r(const ModmArgs &a, const DevTables &tb, double V1ABS, double V2ABS, int NPTABS, int csize, bool high,
                                  bool par, int threads, size_t lds, int lds_sets, hipStream_t s) {
    if (high) return launch_finish_t<R, true, false>(a, tb, V1ABS, V2ABS, NPTABS, csize, threads, lds, s);
    if (par) return launch_finish_t<R, false, true>(a, tb, V1ABS, V2ABS, NPTABS, csize, threads, lds, s);
    // one-wave workgroups on a grid that fills the chip: four layers per wave (lds holds four sets of grids)
    if (threads == 64 && lds_sets == 4) return launch_finish_t<R, false, false, true>(a, tb, V1ABS, V2ABS, NPTABS, csize, threads, lds, s);
    return launch_finish_t<R, false, false>(a, tb, V1ABS, V2ABS, NPTABS, csize, threads, lds, s);
}
// par: passes side by side in the waves of a 256-thread workgroup (lds = 4 sets of grids); only without `high`
hipError_t launch_finish(const ModmArgs &a, const DevTables &tb, double V1ABS, double V2ABS, int NPTABS, int csize, bool high,
                         bool par, int threads, size_t lds, int lds_sets, hipStream_t s) {
    if (a.real_kind == 4) return launch_finish_r<float>(a, tb, V1ABS, V2ABS, NPTABS, csize, high, par, threads, lds, lds_sets, s);
    return launch_finish_r<double>(a, tb, V1ABS, V2ABS, NPTABS, csize, high, par, threads, lds, lds_sets, s);
}
}  // namespace monortm_dev
