// xsec_kernel.hip - optical depth of the cross-section molecules (IXSECT = 1) for gfx950: MONORTM_XSEC_SUB and convolve of the
// reference (src/monortm_sub.F90:1540-1750, :1751-1834) with the tables already parsed (monortm_hip_xsec_tables).
//
// grid = (wavenumbers, layers, profiles); block = one wave.  Everything about a (layer, spectral region) - the temperature
// bracket, the pressure of the blended measurement, the extra Lorentz width hwb, the step of the resampled grid - is
// wave-uniform.  The reference then walks outwards from the wavenumber, one pair of grid points per trip, until a pair adds
// less than ratio x 1e-6 of the running sum (:1797-1817): here the 64 lanes take 64 consecutive trips at once, an inclusive
// scan gives every lane the running sum the sequential walk would hold before its trip, and the first lane whose pair meets
// the criterion ends the walk - the same stopping point (the tail that is cut is NOT negligible, so it has to be the same).
// The resampled spectrum xspd_int (:1779-1785, up to 10^7 points per layer and region in the reference) is never stored:
// a grid value is two table reads per temperature, formed where it is needed.
#include "lineshape.hpp"

namespace {
using namespace monortm_dev;

__device__ __forceinline__ double wave_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

struct XsLayer {  // one (layer, region): what :1677-1716 and :1762-1777 leave behind
    double coef1, coef2, xkt1, xkt2, v1x, delvx, step;
    const double *d1, *d2;
    int nptsx;
};
// xspd(i1), i1 1-based (:1711-1715); beyond the data the reference reads static storage that nothing has written: zero
__device__ __forceinline__ double xs_xspd(const XsLayer &q, int i1) {
    if (i1 < 1 || i1 > q.nptsx) return 0.;
    const double vv = q.v1x + (double)(i1 - 1) * q.delvx;
    return q.coef1 * q.d1[i1 - 1] / radfn(vv, q.xkt1) + q.coef2 * q.d2[i1 - 1] / radfn(vv, q.xkt2);
}
// xspd_int(i) = (1-coef) xspd(ind+1) + coef xspd(ind+2), ind = int(i step / delvx)   (:1779-1785)
__device__ __forceinline__ double xs_int(const XsLayer &q, int i) {
    const double vv = q.v1x + (double)i * q.step;
    const double delvv = vv - q.v1x;
    const int ind = (int)(delvv / q.delvx);
    const double cf = (delvv - (double)ind * q.delvx) / q.delvx;
    return (1. - cf) * xs_xspd(q, ind + 1) + cf * xs_xspd(q, ind + 2);
}

template <typename R>
__global__ __launch_bounds__(64) void xsec_kernel(ModmArgs a, DevXsec x) {
    const int iw = blockIdx.x, lay = blockIdx.y, prof = blockIdx.z, lane = threadIdx.x;
    const int nwn = a.nwn;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    R *out = wp<R>(a.ODXSEC) + pl * (size_t)nwn + iw;
    if (lay >= a.nlay[prof]) {
        if (lane == 0) *out = (R)0;
        return;
    }
    const double pave = (double)rp<R>(a.P)[pl], tave = (double)rp<R>(a.T)[pl];
    const double wnv = a.wn[iw];
    const R *xam = rp<R>(a.XAMNT) + pl * (size_t)x.nxs;
    const double dvbuf = 1.0, p0 = 1013.;
    double xstot = 0.;
    for (int ixmol = 0; ixmol < x.nxs; ixmol++) {
        double xsmol = 0.;
        for (int r = 0; r < x.nreg; r++) {
            const double *rg = x.reg + (size_t)r * 8;
            if ((int)rg[0] != ixmol) continue;
            const double v1fx = rg[1], v2fx = rg[2];   // FSCDXS entry: is the region processed at all (:1645)
            const double v1x = rg[6], v2x = rg[7];     // header of the last file read: the grid and the in-range test (:1663-1666)
            const int nptsx = (int)rg[3], ntemp = (int)rg[4];
            const double xdoplr = rg[5];
            // the region is processed when SOME wavenumber of the call lies within 1 cm-1 of it (:1645-1653)
            // (the wavenumbers ascend - modm.f90:180-181, checked by lines_kernel - so the first one at or above the lower bound
            // decides: a binary search instead of a scan of all nwn values per workgroup and region)
            int lo = 0, hi = nwn;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.wn[mid] < v1fx - dvbuf) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= nwn || !(a.wn[lo] <= v2fx + dvbuf)) continue;
            double res = 0.;
            if (!(wnv < v1x || wnv > v2x)) {  // (:1789-1792)
                const double *tx = x.temps + (size_t)r * 6, *pdx = x.pres + (size_t)r * 6;
                // temperature bracket, the tables are in ascending temperature (:1677-1704)
                double coef1 = 1., coef2 = 0.;
                int ind1, ind2 = 1, it = 1;
                if (ntemp == 1 || tave <= tx[it - 1]) ind1 = 1;
                else {
                    for (;;) {
                        it = it + 1;
                        if (it > ntemp) { ind1 = ntemp; ind2 = ntemp; break; }
                        else if (tave <= tx[it - 1]) {
                            ind1 = it - 1; ind2 = it;
                            coef1 = (tave - tx[it - 1]) / (tx[it - 2] - tx[it - 1]);
                            coef2 = 1. - coef1;
                            break;
                        }
                    }
                }
                const double pd = coef1 * pdx[ind1 - 1] + coef2 * pdx[ind2 - 1];
                XsLayer q;
                q.coef1 = coef1; q.coef2 = coef2;
                q.xkt1 = tx[ind1 - 1] / K_RADCN2; q.xkt2 = tx[ind2 - 1] / K_RADCN2;
                q.v1x = v1x; q.nptsx = nptsx;
                q.delvx = (v2x - v1x) / (double)(nptsx - 1);
                q.d1 = x.pool + x.offs[(size_t)r * 6 + ind1 - 1];
                q.d2 = x.pool + x.offs[(size_t)r * 6 + ind2 - 1];
                const double hwdop = xdoplr * sqrt(tave / 296.);
                // convolve: half widths, step of the resampled grid (:1762-1777)
                double hwpave = 0.1 * (pave / p0) * (273.15 / tave);
                double hwd = 0.1 * (pd / p0) * (273.15 / tave);
                hwd = fmax(hwd, hwdop);
                if (hwd > hwpave) hwpave = 1.001 * hwd;
                const double hwb = hwpave - hwd;
                double ratio = 0.25, step = ratio * hwb;
                if (step > q.delvx) step = q.delvx;
                const int npts = (int)((v2x - v1x) / step);
                step = (v2x - v1x) / (double)npts;
                ratio = step / hwb;
                q.step = step;
                if (hwb / hwd > 0.1) {
                    // Lorentzian of width hwb over the resampled spectrum (:1788-1821)
                    const double hwb2 = hwb * hwb;
                    const double wn_v1x = wnv - v1x;
                    const int ind = (int)(wn_v1x / step);
                    const double dvlo0 = wnv - (v1x + (double)ind * step), dvhi0 = wnv - (v1x + (double)(ind + 1) * step);
                    double answer = (hwb / (hwb2 + dvlo0 * dvlo0)) * xs_int(q, ind) + (hwb / (hwb2 + dvhi0 * dvhi0)) * xs_int(q, ind + 1);
                    const double thr = ratio * 1e-6;
                    for (int jb = 0;; jb += 64) {
                        const int j = jb + lane + 1;
                        double contlo = 0., conthi = 0.;
                        const double vlo = v1x + (double)(ind - j) * step;
                        if (vlo > v1x) {
                            const double dvlo = wnv - vlo;
                            contlo = (hwb / (hwb2 + dvlo * dvlo)) * xs_int(q, ind - j);
                        }
                        const double vhi = v1x + (double)(ind + j + 1) * step;
                        if (vhi < v2x) {
                            const double dvhi = wnv - vhi;
                            conthi = (hwb / (hwb2 + dvhi * dvhi)) * xs_int(q, ind + j + 1);
                        }
                        const double xincr = contlo + conthi;
                        // running sum BEFORE this lane's trip: answer + the trips of the lanes below (inclusive scan, then shift)
                        double incl = xincr;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const double up = __shfl_up(incl, d);
                            if (lane >= d) incl += up;
                        }
                        const double before = answer + (incl - xincr);
                        const unsigned long long stop = __ballot((xincr / before) < thr);
                        if (stop != 0ull) {
                            const int first = (int)__builtin_ctzll(stop);
                            answer = wave_bcast(before, first);
                            break;
                        }
                        answer = wave_bcast(answer + incl, 63);
                        // past both ends of the spectrum every later pair is zero: the reference would spin on 0/0 when the sum
                        // itself is zero (no data) - every wave leaves here
                        if (ind - (jb + 64) < 0 && ind + jb + 65 > npts) break;
                    }
                    res = answer * step / 3.14159;   // (:1820; the reference's own value of pi here)
                } else {
                    // linearly interpolated values (:1822-1828) - with xspd(ind), xspd(ind+1): one element below the convention of
                    // the resampling above, as the reference has it
                    const double wn_v1x = wnv - v1x;
                    const int ind = (int)(wn_v1x / q.delvx);
                    const double coef = (wn_v1x - (double)ind * q.delvx) / q.delvx;
                    res = (1. - coef) * xs_xspd(q, ind) + coef * xs_xspd(q, ind + 1);
                }
            }
            xsmol += res;
        }
        xstot += (double)xam[ixmol] * xsmol;   // (:1729-1733)
    }
    if (lane == 0) *out = (R)(xstot * radfn(wnv, tave / K_RADCN2));   // the radiation term back in (:1738-1744)
}

}  // namespace

namespace monortm_dev {
void launch_xsec(const ModmArgs &a, const DevXsec &x, hipStream_t s) {
    const dim3 grid((unsigned)a.nwn, (unsigned)a.nlay_max, (unsigned)a.nprof);
    if (a.real_kind == 4) hipLaunchKernelGGL(xsec_kernel<float>, grid, dim3(64), 0, s, a, x);
    else hipLaunchKernelGGL(xsec_kernel<double>, grid, dim3(64), 0, s, a, x);
}
}  // namespace monortm_dev
