// rtm_kernel.hip - CALCTMR + RAD_UP_DN + RTM (reference src/RTMmono.f90) for gfx950.  See DESIGN.md section 3.3.
#include "device_common.hpp"

namespace {
using namespace monortm_dev;

// ------------------------------------------------------------------------------------------------
// rtm_kernel: CALCTMR (RTMmono.f90:239-325) + RAD_UP_DN (:157-221) + RTM (:13-155); lane = (profile, wn)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double bb_fn(double v, double fbeta) { return K_RADCN1 * (v * v * v) / (exp(v * fbeta) - 1.); }

// Block = 64 wavenumbers x G layer groups.  The recurrences of RAD_UP_DN are sums of independent terms once
// the optical depth above / below a layer is known:  ODT after the reference's running subtraction equals the
// optical depth of the layers not yet visited.  Every thread walks its contiguous group of layers exactly like
// the reference (same running subtraction, same term formula), group partial sums are combined through LDS in
// the reference's visiting order (surface->top for RUP, top->surface for RDN / TMR).
// R: element type of the REAL arrays (real_kind 8 / 4); the recurrences themselves run in double
template <typename R, int G>
__global__ __launch_bounds__(64 * G) void rtm_kernel(RtmArgs a) {
    __shared__ double sPart[G][64];
    __shared__ double sUp[G][64], sDn[G][64], sEx[G][64];
    const int lane = threadIdx.x, g = threadIdx.y;
    const int iw0 = blockIdx.x * 64 + lane, prof = blockIdx.y;
    const int nwn = a.nwn;
    const bool valid = iw0 < nwn;
    const int iw = valid ? iw0 : nwn - 1;
    const int nlay = max(0, min(a.nlay[prof], a.nlay_max)), irt = a.irt[prof];  // out-of-range counts are flagged by lines_kernel / the host
    const double VV = a.wn[iw];
    const R *O = rp<R>(a.O) + (size_t)prof * a.nlay_max * nwn + iw;
    const R *T = rp<R>(a.T) + (size_t)prof * a.nlay_max, *TZ = rp<R>(a.TZ) + (size_t)prof * (a.nlay_max + 1);
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

    double RUP = 0., RDN = 0., sumexp = 0.;
    if (irt != 3) {  // RTMmono.f90:193-205, layers l0+1 .. l1 (1-based) of the upward sweep
        double ODT = ODTOT - below;
        for (int l = l0 + 1; l <= l1; l++) {
            const double bb = bb_fn(VV, K_RADCN2 / (double)T[l - 1]), bba = bb_fn(VV, K_RADCN2 / (double)TZ[l]);
            const double ODVI = (double)O[(size_t)(l - 1) * nwn];
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
            const double bb = bb_fn(VV, K_RADCN2 / (double)T[l - 1]), bba = bb_fn(VV, K_RADCN2 / (double)TZ[l - 1]);
            const double ODVI = (double)O[(size_t)(l - 1) * nwn];
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
    if (a.real_kind == 4) {
        if (few) hipLaunchKernelGGL((rtm_kernel<float, 16>), grid, dim3(64, 16), 0, s, a);
        else if (a.nlay_max >= 24) hipLaunchKernelGGL((rtm_kernel<float, 8>), grid, dim3(64, 8), 0, s, a);
        else hipLaunchKernelGGL((rtm_kernel<float, 2>), grid, dim3(64, 2), 0, s, a);
    } else {
        if (few) hipLaunchKernelGGL((rtm_kernel<double, 16>), grid, dim3(64, 16), 0, s, a);
        else if (a.nlay_max >= 24) hipLaunchKernelGGL((rtm_kernel<double, 8>), grid, dim3(64, 8), 0, s, a);
        else hipLaunchKernelGGL((rtm_kernel<double, 2>), grid, dim3(64, 2), 0, s, a);
    }
}
}  // namespace monortm_dev
