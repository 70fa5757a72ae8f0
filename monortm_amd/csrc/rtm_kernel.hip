// rtm_kernel.hip - CALCTMR + RAD_UP_DN + RTM (reference src/RTMmono.f90) for gfx950.  See DESIGN.md section 3.3.
#include "device_common.hpp"

namespace {
using namespace monortm_dev;

// ------------------------------------------------------------------------------------------------
// rtm_kernel: CALCTMR (RTMmono.f90:239-325) + RAD_UP_DN (:157-221) + RTM (:13-155); lane = (profile, wn)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double bb_fn(double v, double fbeta) { return K_RADCN1 * (v * v * v) / (exp(v * fbeta) - 1.); }

// Block = 64 wavenumbers x G layer groups.  The recurrences of RAD_UP_DN are sums of independent terms once the optical depth
// above / below a layer is known.  Every thread walks its contiguous group of layers ONCE, from the top of the group down, and
// forms the terms of both sweeps and of CALCTMR from shared pieces (round 4; rounds 1-3 walked the group twice and formed every
// exponential and Planck value per sweep):
//   per layer: tau, exp(-tau), the Pade weight, B(T_layer); B at the lower level (the downward sweep's edge) - the upward sweep's
//   edge, B at the upper level, is the value the layer above has just used;
//   exp(-tau above) for the upward and exp(-tau below) for the downward term.
// hc / kT of the layers and levels of the profile are formed once per workgroup (LDS).  The group sums are combined through LDS in
// the reference's visiting order (surface -> top for RUP, top -> surface for RDN / TMR); inside a group the upward terms are added
// top-down and the optical depth above a layer is a running sum instead of the reference's running difference - differences of
// the order of the last bit of the double-precision sums.
// R: element type of the REAL arrays (real_kind 8 / 4); the recurrences themselves run in double
template <typename R, int G>
__global__ __launch_bounds__(64 * G) void rtm_kernel(RtmArgs a) {
    __shared__ double sPart[G][64];
    __shared__ double sUp[G][64], sDn[G][64], sEx[G][64];
    extern __shared__ __attribute__((aligned(16))) double sBeta[];  // [nlay_max] hc/kT of the layers, [nlay_max + 1] of the levels
    const int lane = threadIdx.x, g = threadIdx.y;
    const int iw0 = blockIdx.x * 64 + lane, prof = blockIdx.y;
    const int nwn = a.nwn;
    const bool valid = iw0 < nwn;
    const int iw = valid ? iw0 : nwn - 1;
    const int nlay = max(0, min(a.nlay[prof], a.nlay_max)), irt = a.irt[prof];  // out-of-range counts are flagged by lines_kernel / the host
    const double VV = a.wn[iw];
    const R *O = rp<R>(a.O) + (size_t)prof * a.nlay_max * nwn + iw;
    const R *T = rp<R>(a.T) + (size_t)prof * a.nlay_max, *TZ = rp<R>(a.TZ) + (size_t)prof * (a.nlay_max + 1);
    double *sBl = sBeta, *sBz = sBeta + a.nlay_max;
    for (int l = g * 64 + lane; l < 2 * nlay + 1; l += 64 * G) {
        if (l < nlay) sBl[l] = K_RADCN2 / (double)T[l];
        else sBz[l - nlay] = K_RADCN2 / (double)TZ[l - nlay];
    }
    const int chunk = (nlay + G - 1) / G;
    const int l0 = min(nlay, g * chunk), l1 = min(nlay, l0 + chunk);  // 0-based layer range [l0, l1)

    double part = 0.;
    for (int l = l0; l < l1; l++) part = part + (double)O[(size_t)l * nwn];
    sPart[g][lane] = part;
    __syncthreads();
    double below = 0., ODTOT = 0.;
    for (int gg = 0; gg < G; gg++) {
        if (gg < g) below = below + sPart[gg][lane];
        ODTOT = ODTOT + sPart[gg][lane];
    }
    const double above = ODTOT - below - part;
    const double c3 = K_RADCN1 * (VV * VV * VV);
    const bool up = irt != 3;

    double RUP = 0., RDN = 0., sumexp = 0.;
    {  // RTMmono.f90:193-217 and CALCTMR :302-315, layers l1 .. l0+1 (1-based)
        double ODTd = ODTOT - above;        // downward sweep: optical depth from the surface up to and including the layer, running difference
        double ODTu = above;                // upward sweep: optical depth above the layer
        double bb_top = (up && l1 > l0) ? planck(c3, VV, sBz[l1]) : 0.;  // B at the upper level of the group's top layer
        for (int l = l1; l >= l0 + 1; l--) {
            const double ODVI = (double)O[(size_t)(l - 1) * nwn];
            const double bb = planck(c3, VV, sBl[l - 1]), bbz = planck(c3, VV, sBz[l - 1]);
            const double TRI = exp_cw(-ODVI);
            const double pade = 0.193 * ODVI + 0.013 * (ODVI * ODVI);
            const double rp1 = rcp2(1. + pade), emis = 1. - TRI;
            ODTd = ODTd - ODVI;
            const double TRd = exp_cw(-ODTd);
            const double bnum = bb + pade * bbz;
            RDN = RDN + ((TRd * emis) * bnum) * rp1;      // TR (1 - TRI) (bb + pade bba) / (1 + pade), RTMmono.f90:216
            sumexp = sumexp + ((bnum * rp1) * TRd) * emis;  // beff TR (1 - TRI), RTMmono.f90:312-313
            if (up) {
                const double TRu = exp_cw(-ODTu);
                RUP = RUP + ((TRu * emis) * (bb + pade * bb_top)) * rp1;  // RTMmono.f90:203
                ODTu = ODTu + ODVI;
                bb_top = bbz;
            }
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
        wp<R>(a.TMR)[o] = (R)(K_RADCN2 * VV / log(x));
    }
    const double TSKY = 2.75;
    double tmpsfc = (double)wp<R>(a.tmpsfc)[prof];
    if (irt == 3 || irt == 2) tmpsfc = TSKY;  // RTMmono.f90:113-124
    const double SURFRAD = bb_fn(VV, K_RADCN2 / tmpsfc), COSMOS = bb_fn(VV, K_RADCN2 / TSKY);
    const double ESFC = (double)rp<R>(a.emiss)[o], RSFC = (double)rp<R>(a.reflc)[o];
    double RAD = 0.;
    if (irt == 1) RAD = RUP + TRTOT * (ESFC * SURFRAD + RSFC * (RDN + TRTOT * COSMOS));
    if (irt == 2) RAD = RUP + TRTOT * (RDN + TRTOT * COSMOS);
    if (irt == 3) RAD = RDN + (TRTOT * COSMOS);
    // TMPSFC is an in/out argument of the reference's RTM (RTMmono.f90:122).  Lanes of this profile that still
    // read the old value ignore it exactly when it is overwritten (irt = 2,3), so the store needs no ordering.
    if (iw == 0 && (irt == 3 || irt == 2)) wp<R>(a.tmpsfc)[prof] = (R)TSKY;
    wp<R>(a.RUP)[o] = (R)RUP;
    wp<R>(a.RDN)[o] = (R)RDN;
    wp<R>(a.TRTOT)[o] = (R)TRTOT;
    wp<R>(a.RAD)[o] = (R)RAD;
    if (a.iout == 1) {
        const double X = K_RADCN1 * (VV * VV * VV) / RAD + 1.;
        wp<R>(a.TB)[o] = (R)(K_RADCN2 * VV / log(X));
    }
}

}  // namespace

namespace monortm_dev {
void launch_rtm(const RtmArgs &a, hipStream_t s) {
    dim3 grid((a.nwn + 63) / 64, a.nprof);
    // few workgroups (single profiles) and many layers: 16 layer groups shorten each thread's chain of exponentials
    const bool few = (long long)grid.x * grid.y < 256 && a.nlay_max >= 48;
    const size_t lds = sizeof(double) * (size_t)(2 * a.nlay_max + 1);  // hc / kT of the layers and levels
    if (a.real_kind == 4) {
        if (few) hipLaunchKernelGGL((rtm_kernel<float, 16>), grid, dim3(64, 16), lds, s, a);
        else if (a.nlay_max >= 24) hipLaunchKernelGGL((rtm_kernel<float, 8>), grid, dim3(64, 8), lds, s, a);
        else hipLaunchKernelGGL((rtm_kernel<float, 2>), grid, dim3(64, 2), lds, s, a);
    } else {
        if (few) hipLaunchKernelGGL((rtm_kernel<double, 16>), grid, dim3(64, 16), lds, s, a);
        else if (a.nlay_max >= 24) hipLaunchKernelGGL((rtm_kernel<double, 8>), grid, dim3(64, 8), lds, s, a);
        else hipLaunchKernelGGL((rtm_kernel<double, 2>), grid, dim3(64, 2), lds, s, a);
    }
}
}  // namespace monortm_dev
