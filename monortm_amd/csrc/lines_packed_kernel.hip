// lines_packed_kernel.hip - the line sum of MODM / LINES (reference src/modm.f90:253-262, :277-440) for channel sets that
// leave lines_kernel's 64-lane tile partly empty (BASELINE configs[1], configs[3]: 50 channels = 50 of 64 lanes).
//
// lines_kernel gives one (profile, layer) to a one-wave workgroup, lane = wavenumber.  Here a four-wave workgroup takes
// NL = 256 / nwn consecutive layers of a profile and its 256 lanes are the (layer, wavenumber) pairs, wavenumber fastest
// (50 channels: 5 layers = 250 of 256 lanes).  Everything else is lines_kernel's: the same prepare_line() per (layer, line)
// - one lane per (layer, line), CL = 256 / NL lines of every layer per chunk - the same class masks, sub-run walker and
// loops (eval_dispatch); a lane reads the records of ITS layer.  A line's class is the strictest over the block's layers.
// Per (layer, wavenumber, molecule) the lines are added in table order, as everywhere.  DESIGN.md section 3.1c.
#include "lines_device.hpp"

namespace {
using namespace monortm_dev;

constexpr int PK_NW = 4;         // waves per workgroup
constexpr int PK_NT = PK_NW * 64;
constexpr int PK_MAXL = 8;       // layers per workgroup, at most (CL = 256 / NL >= 32 lines per chunk, <= 64: one mask word)

template <typename R, bool IBRD>
__global__ __launch_bounds__(PK_NT, 4) void lines_packed_kernel(ModmArgs a, DevLines L, DevTables tb, int NL, int CL) {
    constexpr bool SGL = sizeof(R) == 4;
    using Hot = typename HotOf<R>::type;
    __shared__ Hot sA[PK_NT];        // [layer of the block][line of the chunk]
    __shared__ HotB sB[PK_NT];
    __shared__ ColdLine sCold[PK_NT];
    __shared__ double sWn[64];       // the wavenumbers (ascending; positions past nwn repeat the last)
    __shared__ double sLay[PK_MAXL][20];  // layer scalars per layer of the block ([18] = temperature, [19] = 1 when the layer is part of the profile)
    __shared__ unsigned char sFlg[PK_MAXL][64];  // class flags per (layer, line of the chunk): AL | M2 << 1 | V << 2 | Y << 3
    __shared__ unsigned long long sMask[PK_NW][4];  // per wave (each forms them itself): all-live, two resonances, Voigt, Y factors
    __shared__ unsigned short sVq[PK_NW][64];
    __shared__ double sPad;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const int nmol = a.nmol, nwn = a.nwn;
    double *sScor = dyn_lds;                       // [NL][nmol*9] Q(296)/Q(T)
    double *sDop = sScor + NL * nmol * 9;          // [NL][nmol*9] HWHM_D / Xnu
    double *sW = sDop + NL * nmol * 9;             // [NL][nmol]   column amounts
    int *sLo = reinterpret_cast<int *>(sW + NL * nmol);  // [nmol]   first candidate line
    int *sOff = sLo + nmol;                        // [nmol+1] prefix sums of the candidate counts

    const int tid = threadIdx.x, wave = tid >> 6;
    // top layers first (their prepare stage is the longest: Voigt proximity searches), as in lines_kernel
    const int prof = blockIdx.y, grp = (int)gridDim.x - 1 - (int)blockIdx.x;
    const int lay0 = grp * NL;
    const int nlayp = a.nlay[prof];
    // ---- evaluate role: lane = (layer ls, wavenumber iw) ----
    const int ls_raw = tid / nwn, iw = tid - ls_raw * nwn;
    const bool lane_on = ls_raw < NL && lay0 + ls_raw < a.nlay_max;      // the lane owns an output column
    const int ls = lane_on ? ls_raw : 0;
    const int lay = lay0 + ls;
    const bool in_prof = lane_on && lay < nlayp;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    R *obm = wp<R>(a.O_BY_MOL) + pl * nmol * (size_t)nwn;
    // ---- prepare role: lane = (layer lp, line jl of the chunk) ----
    const int lp_raw = tid / CL, jl = tid - lp_raw * CL;
    const bool prep_on = lp_raw < NL && lay0 + lp_raw < nlayp;
    const int lp = (lp_raw < NL) ? lp_raw : 0;

    // arguments that live in device memory cannot be validated by the host side of a *_dev call: flag them here
    if (grp == 0) {
        if (tid == 0 && (nlayp < 1 || nlayp > a.nlay_max)) atomicOr(a.errflag, ERRBIT_ARG);
        if (prof == 0 && tid + 1 < nwn && a.wn[tid + 1] < a.wn[tid]) atomicOr(a.errflag, ERRBIT_ARG);  // modm.f90:180-181
    }
    // layers beyond nlay[p]: zeros (modm.f90:314 starts every output from zero)
    if (lane_on && !in_prof)
        for (int m = 0; m < nmol; m++) obm[(size_t)m * nwn + iw] = (R)0;
    if (lay0 >= nlayp) return;  // (block-uniform)

    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    if (tid < 64) sWn[tid] = a.wn[min(tid, nwn - 1)];
    // ---- layer scalars (INITI + head of LINES: modm.f90:868-883, :301-314): one lane per layer of the block ----
    if (tid < NL) {
        const int l = lay0 + tid;
        const bool on = l < nlayp;
        const size_t q = (size_t)prof * a.nlay_max + (on ? l : lay0);
        const double Pk = rp<R>(a.P)[q], Tk = rp<R>(a.T)[q], wbrod = rp<R>(a.WBRODL)[q];
        const R *wk = rp<R>(a.WKL) + q * nmol;
        // MODM calls TIPS_2003 for every layer and all nmol molecules (modm.f90:250): outside 70-3000 K the reference STOPs
        if (on && (Tk < 70. || Tk > 3000.)) atomicOr(a.errflag, ERRBIT_TEMP);
        const double XN0 = (K_P0 / (K_BOLTZ * K_T0)) * 1.E+3;
        const double Xn = (Pk / (K_BOLTZ * Tk)) * 1.E+3;
        double WTOT = 0.;
        for (int m = 0; m < nmol; m++) WTOT += wk[m];
        WTOT = WTOT + wbrod;
        const double RP = Pk / K_P0, RP2 = RP * RP;
        const double RT = Tk / K_T0, RHORAT = Xn / XN0;
        const int ILC = (Tk < 250.0) ? 1 : ((Tk < 296.0) ? 2 : 3);  // TEMPLC = 200,250,296,340
        const double tlo = (ILC == 1) ? 200.0 : (ILC == 2 ? 250.0 : 296.0);
        const double thi = (ILC == 1) ? 250.0 : (ILC == 2 ? 296.0 : 340.0);
        double *sl = sLay[tid];
        sl[0] = RHORAT; sl[1] = RP; sl[2] = RP2; sl[3] = log(RT); sl[4] = RADCT / Tk; sl[5] = RADCT / K_T0;
        sl[6] = 1.0 / K_T0 - 1.0 / Tk;
        sl[7] = 1.0 / (thi - tlo); sl[8] = Tk - tlo; sl[9] = WTOT; sl[17] = (double)ILC;
        for (int j = 0; j < MXBRD; j++) sl[10 + j] = RHORAT * wk[j] / WTOT;  // rho_molec(1:7), modm.f90:313
        sl[18] = Tk;
        sl[19] = on ? 1. : 0.;
    }
    for (int t = tid; t < NL * nmol; t += PK_NT) {
        const int l = t / nmol, m = t - l * nmol;
        const bool on = lay0 + l < nlayp;
        sW[t] = on ? (double)rp<R>(a.WKL)[((size_t)prof * a.nlay_max + lay0 + l) * nmol + m] : 0.;
    }
    __syncthreads();
    if (tid == 0) {
        // |Xnu - XNU0| <= max_abs_shift * RHORAT for every entry, with or without species broadening (line_table.cpp)
        double mx = 1.0;
        for (int l = 0; l < NL; l++)
            if (sLay[l][19] != 0.) mx = fmax(mx, sLay[l][0]);
        sPad = L.max_abs_shift * mx + 1e-6;
    }
    __syncthreads();
    // ---- candidate range of every molecule (one tile: all wavenumbers), common to the layers of the block ----
    {
        const double wnlo = sWn[0], wnhi = sWn[nwn - 1], pad = sPad;
        for (int m = tid; m < nmol; m += PK_NT) {
            const int mol = m + 1;
            int lo = L.mol_start[mol], hi = L.mol_start[mol + 1];
            bool any = false;  // W_SPECIES == 0 -> OL = 0 without a walk (modm.f90:318-321): when that holds for every layer here
            for (int l = 0; l < NL; l++) any = any || sW[l * nmol + m] != 0.;
            if (!any) hi = lo;
            // coupled O2 lines are exempt from the rule (modm.f90:755-792); an O2 list without any obeys it like the others
            else if ((mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull)) {
                const double vlo = wnlo - 25.0 - pad, vhi = wnhi + 25.0 + pad;  // 25 cm-1 rule (modm.f90:384)
                if (!(hi > lo && !(L.vnu[lo] < vlo) && L.vnu[hi - 1] <= vhi)) {
                    int l0 = lo, l1 = hi;
                    while (l0 < l1) { int mid = (l0 + l1) >> 1; if (L.vnu[mid] < vlo) l0 = mid + 1; else l1 = mid; }
                    const int first = l0;
                    l1 = hi;
                    while (l0 < l1) { int mid = (l0 + l1) >> 1; if (L.vnu[mid] <= vhi) l0 = mid + 1; else l1 = mid; }
                    lo = first;
                    hi = l0;
                }
            }
            sLo[m] = lo;
            sOff[m + 1] = hi - lo;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        sOff[0] = 0;
        for (int m = 0; m < nmol; m++) { acc += sOff[m + 1]; sOff[m + 1] = acc; }
    }
    __syncthreads();
    const int total = sOff[nmol];
    // TIPS + Doppler factor per (layer, molecule, isotopologue) of the molecules that have candidate lines
    // (src/tips_2003.f90:60-296, src/modm.f90:442-454)
    for (int t = tid; t < NL * nmol * 9; t += PK_NT) {
        const int l = t / (nmol * 9), r = t - l * (nmol * 9);
        const int mol = r / 9 + 1, iso = r % 9 + 1;
        if (sOff[mol] == sOff[mol - 1]) continue;
        const double Tk = sLay[l][18];
        double sc = 0., dop = 0.;
        if (sLay[l][19] != 0. && !(Tk < 70. || Tk > 3000.)) {
            bool bad = false;
            sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
            if (bad) atomicOr(a.errflag, ERRBIT_TEMP);
        }
        const double M = tb.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = sqrt(2. * log(2.) * ((K_BOLTZ * Tk) / (M / K_AVOGAD))) / K_CLIGHT;
        sScor[t] = sc;
        sDop[t] = dop;
    }
    // molecules without candidate lines / without column in every layer: OL = 0 (modm.f90:314, :318-321)
    if (in_prof)
        for (int m = 0; m < nmol; m++)
            if (sOff[m + 1] == sOff[m]) obm[(size_t)m * nwn + iw] = (R)0;
    __syncthreads();

    double WNk[1] = {sWn[iw]};
    const double RFT = WNk[0] * tanh_pos((RADCT * WNk[0]) / (2 * sLay[ls][18]));
    R SFk[1] = {(R)0};
    double osum = 0.;  // sum over the molecules of O_BY_MOL as stored (written once, at the end)
    const int rec_off = ls * CL;
    const Hot *myA = sA + rec_off;
    const HotB *myB = sB + rec_off;
    const ColdLine *myC = sCold + rec_off;

    for (int base = 0, ck = 0; base < total; base += CL, ck++) {
        // ================= prepare: one lane per (layer, line of the chunk) =========================
        const int v = base + jl;
        {
            bool fAL = true, fM2 = false, fV = false, fY = false;  // (a layer outside the profile does not restrict the classes)
            Hot hA{};
            HotB hB{};
            ColdLine cC{};
            if (prep_on && v < total) {
                int m = 0;
                while (sOff[m + 1] <= v) m++;
                const int idx = sLo[m] + (v - sOff[m]);
                prepare_line<R, IBRD>(a, L, idx, m, sLay[lp], sScor + lp * nmol * 9, sDop + lp * nmol * 9, sW + lp * nmol, sWn, nwn, nullptr, hA,
                                      hB, cC, fAL, fM2, fV, fY);
            }
            if (lp_raw < NL) {
                sA[lp * CL + jl] = hA;
                sB[lp * CL + jl] = hB;
                sCold[lp * CL + jl] = cC;
                sFlg[lp][jl] = (unsigned char)((fAL ? 1 : 0) | (fM2 ? 2 : 0) | (fV ? 4 : 0) | (fY ? 8 : 0));
            }
        }
        __syncthreads();
        // ---- class masks of the chunk's lines, the strictest over the layers (every wave forms them for itself) ----
        {
            const int lane = tid & 63;
            unsigned f = 1u;  // AL: and;  M2, V, Y: or
            if (lane < CL) {
                for (int l = 0; l < NL; l++) {
                    const unsigned x = sFlg[l][lane];
                    f = (f & x & 1u) | ((f | x) & 14u);
                }
            }
            const bool inl = lane < CL && base + lane < total;
            const unsigned long long bA = __ballot(inl && (f & 1u)), bM = __ballot(inl && (f & 2u));
            const unsigned long long bV = __ballot(inl && (f & 4u)), bY = __ballot(inl && (f & 8u));
            if (lane == 0) {
                // short all-live islands take the tested loop of their neighbours, short one-resonance gaps the two-resonance
                // loop (0/1 factor per lane), as in lines_kernel
                sMask[wave][0] = open_runs8(bA);
                sMask[wave][1] = close_runs8(bM);
                sMask[wave][2] = bV;
                sMask[wave][3] = bY;
            }
        }
        // ================= evaluate: every wave walks the prepared lines, molecule by molecule =========
        const int nch = min(CL, total - base);
        for (int m = 0; m < nmol; m++) {
            const int s0 = sOff[m], s1 = sOff[m + 1];
            if (s1 <= base || s0 >= s1) continue;
            if (s0 >= base + nch) break;
            const int j0 = max(s0, base) - base, j1 = min(s1, base + nch) - base;
            if (s0 >= base) SFk[0] = (R)0;  // the molecule's run starts in this chunk
            const int mol = m + 1;
            const double wsc = SGL ? sW[ls * nmol + m] : 1.0;
            const unsigned long long *mk = sMask[wave];
            if (mol == 7) eval_dispatch<1, R, Hot, 1, true>(mk, mk + 1, nullptr, mk + 2, mk + 3, myA, myB, myC, j0, j1, WNk, mol, SFk, wsc, a.errflag, sVq[wave], rec_off);
            else if (mol == 2) eval_dispatch<2, R, Hot, 1, true>(mk, mk + 1, nullptr, mk + 2, mk + 3, myA, myB, myC, j0, j1, WNk, mol, SFk, wsc, a.errflag, sVq[wave], rec_off);
            else eval_dispatch<0, R, Hot, 1, true>(mk, mk + 1, nullptr, mk + 2, mk + 3, myA, myB, myC, j0, j1, WNk, mol, SFk, wsc, a.errflag, sVq[wave], rec_off);
            // run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438); in single precision W is already inside SF
            if (s1 <= base + nch && in_prof) {
                const double Wm = sW[ls * nmol + m];
                const R od = (Wm == 0.) ? (R)0 : (R)(SGL ? RFT * (double)SFk[0] : RFT * (Wm * (double)SFk[0]));
                obm[(size_t)m * nwn + iw] = od;
                osum += (double)od;  // molecules complete in ascending order: the sum of modm.f90:264-269
            }
        }
        __syncthreads();  // the records are overwritten by the next chunk
    }
    if (a.osum && in_prof) a.osum[pl * (size_t)nwn + iw] = osum;
}

}  // namespace

namespace monortm_dev {
// layers per workgroup for nwn wavenumbers (0: the packed kernel has nothing to offer for this channel count)
int lines_packed_layers(int nwn) {
    if (nwn < 1 || nwn > 64) return 0;
    const int nl = std::min(PK_MAXL, PK_NT / nwn);
    if (nl < 4) return 0;  // (nwn > 64 cannot happen here; 4 layers keep a chunk of <= 64 lines per layer)
    // worth it when the (layer, wavenumber) pairs fill the workgroup clearly better than the wavenumbers fill a wave
    const double packed = (double)(nl * nwn) / PK_NT, plain = (double)nwn / 64.0;
    return packed > plain + 0.08 ? nl : 0;
}
void launch_lines_packed(const ModmArgs &a, const DevLines &L, const DevTables &tb, bool ibrd, hipStream_t s) {
    const int NL = lines_packed_layers(a.nwn), CL = PK_NT / NL;
    const dim3 grid((unsigned)((a.nlay_max + NL - 1) / NL), (unsigned)a.nprof);
    const size_t dyn = sizeof(double) * (size_t)(NL * 19 * a.nmol) + sizeof(int) * (size_t)(2 * a.nmol + 2);
    if (a.real_kind == 4) {
        if (ibrd) hipLaunchKernelGGL((lines_packed_kernel<float, true>), grid, dim3(PK_NT), dyn, s, a, L, tb, NL, CL);
        else hipLaunchKernelGGL((lines_packed_kernel<float, false>), grid, dim3(PK_NT), dyn, s, a, L, tb, NL, CL);
    } else {
        if (ibrd) hipLaunchKernelGGL((lines_packed_kernel<double, true>), grid, dim3(PK_NT), dyn, s, a, L, tb, NL, CL);
        else hipLaunchKernelGGL((lines_packed_kernel<double, false>), grid, dim3(PK_NT), dyn, s, a, L, tb, NL, CL);
    }
}
}  // namespace monortm_dev
