// lines_ms_kernel.hip - the line sum of MODM / LINES (reference src/modm.f90:253-262, :277-440) for BATCHES OF STATES ON SPARSE
// CHANNEL SETS (configs[3]: 1024 profiles x 64 layers x 50 channels), gfx950, double precision.  Round 6; DESIGN.md section 3.1m.
//
// lines_kernel<double,1,1> gives a one-wave workgroup ONE atmospheric state (profile, layer) and a lane one channel: 50 channels
// leave 14 of 64 lanes idle in every evaluate instruction, a record read from LDS serves one evaluation per lane, and each wave
// pays the prologue (layer scalars, partition sums, candidate windows) of its state alone.  Here a one-wave workgroup takes
// G states - the same layer of G consecutive profiles - and a lane is (state, WPS = 5 channels of it):
//   * evaluate: a lane reads the prepared records of ITS state (per-lane LDS address; the class of a line is common to the
//     wave: the most general over the G states), a record that has been read serves five evaluations, 60 of 64 lanes work
//     (G = 6, 10 lanes x 5 channels a state), four one-resonance lines share one reciprocal (lines_ms_asm.hpp);
//   * prepare: one lane per (state, line) as before - the same functions (line_physics_core, line_records of lines_device.hpp);
//   * prologue: one pass over (state, molecule) and (state, isotopologue) items for all G states.
// The G states share ONE candidate window per molecule (the union of theirs: a line outside a state's own window lies beyond
// 25 cm-1 of every channel and adds nothing - the clamp / the EXEC mask of its class says so), so a line index means the same
// line for every lane.  No barrier anywhere: the workgroup is one wave.  Results: those of lines_kernel up to the rounding of
// the shared reciprocals (1e-15 of a term); tests/test_ms_kernel.py holds the two kernels together and both to the oracle.
#include "lines_device.hpp"
#include "lines_ms_asm.hpp"

namespace {
using namespace monortm_dev;

constexpr int WPS = MS_WPS;   // wavenumbers per lane

__device__ __forceinline__ void ms_sync() {  // one wave: order the compiler, the LDS unit keeps program order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ... and for the records of the rare shapes that travel through global memory (written by one lane, read by another of the same
// wave): the stores have left the wave before a load is issued
__device__ __forceinline__ void ms_sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// what a lane needs to find the records of a state: HotA in LDS, HotB / ColdLine (rare shapes only) in the workgroup's scratch
struct MsState {
    const HotA *sA;
    const HotB *gB;
    const ColdLine *gC;
    int item0;   // first item (state * CL) of the state: bit item0 + j of `spec` says whether gB / gC of line j were written
};
struct MsSpec { unsigned long long w[MS_MAXSTEPS]; };
__device__ __forceinline__ bool ms_own(const MsSpec &sp, int item) {
    unsigned long long x = sp.w[0];
#pragma unroll
    for (int t = 1; t < MS_MAXSTEPS; t++) x = ((item >> 6) == t) ? sp.w[t] : x;
    return (x >> (item & 63)) & 1ull;
}
// HotB of line j for this lane's state: the stored one where the state itself flagged the line (Voigt candidate / Y factors),
// else what line_records would have stored for an ordinary line (pb = pa, no Doppler limit, no Y factors)
__device__ __forceinline__ HotB ms_hotb(const MsState &st, const MsSpec &sp, int j, const HotA &h) {
    if (ms_own(sp, st.item0 + j)) return st.gB[j];
    return HotB{h.pa, -1., 0., 1.};
}

// the Voigt candidates of a molecule run (VSCAN of lines_device.hpp for five wavenumbers and G states): the lines walked the
// Lorentz loops like any other; here every (line, lane, k) within 100 Doppler widths is queued, and the queue is worked off one
// pair per lane: Voigt value minus the Lorentz term that was added
template <int KIND>
__device__ __forceinline__ void ms_voigt_flush(const HotA *sA, const HotB *gB, const ColdLine *gC, const MsArgs &ms, const unsigned short *vq, int n,
                                               const double (&W)[WPS], int mol, double (&S)[WPS], int *errflag) {
    const int lane = (int)__lane_id();
    const unsigned rec = (lane < n) ? vq[lane] : 0u;
    const int j = (int)((rec >> 9) & 63u), kk = (int)((rec >> 6) & 7u), owner = (int)(rec & 63u);
    double WNi = __shfl(W[0], owner);
#pragma unroll
    for (int k = 1; k < WPS; k++) {
        const double wk = __shfl(W[k], owner);
        WNi = (kk == k) ? wk : WNi;
    }
    double val = 0.;
    if (lane < n) {
        const int so = owner / ms.LPS;   // the owner's state: its records (the pair was queued by a lane that saw its own flag)
        const HotA h = sA[so * ms.sa_stride + j];
        const HotB b = gB[so * ms.CL + j];
        const ColdLine c = gC[so * ms.CL + j];
        const double SLS = lsf_sdvoigt(mol, (int)((c.info >> 6) & 3), 1.0, 1.0, b.c1 * c.hw, b.gp1 - 1., c.hw, WNi, h.xnu, c.hwd,
                                       (double)c.sdep, c.xl3, errflag);
        double lor;
        if (rec >> 15) lor = general_term<KIND>(h, b, WNi);
        else if constexpr (KIND == 2) lor = eval_one_fast<2, false, true>(h, 0., WNi);
        else lor = eval_one_fast<KIND, true, true>(h, b.pb, WNi);
        val = c.stild * SLS - lor;
    }
    for (int it = 0; it < n; it++) {  // wave-uniform trip count and indices; queue order = summation order (deterministic)
        const int lo = __builtin_amdgcn_readlane(__double2loint(val), it), hi = __builtin_amdgcn_readlane(__double2hiint(val), it);
        const int r = __builtin_amdgcn_readlane((int)rec, it);
        const double v = __hiloint2double(hi, lo);
        if (lane == (r & 63)) {
            const int k = (r >> 6) & 7;
#pragma unroll
            for (int q = 0; q < WPS; q++)
                if (k == q) S[q] += v;
        }
    }
}
template <int KIND>
__device__ __forceinline__ void ms_voigt_scan(unsigned long long cand, unsigned long long ymask, const MsState &st, const MsSpec &sp, const HotA *sA,
                                              const HotB *gB, const ColdLine *gC, const MsArgs &ms, const double (&W)[WPS], unsigned kvalid, int mol,
                                              double (&S)[WPS], int *errflag, unsigned short *vq) {
    const int lane = (int)__lane_id();
    int nq = 0;
    while (cand) {
        const int j = (int)__builtin_ctzll(cand);
        cand &= cand - 1ull;
        const unsigned ybit = (unsigned)((ymask >> j) & 1ull) << 15;
        const HotA h = st.sA[j];
        const bool own = ms_own(sp, st.item0 + j);
        const double d100 = own ? st.gB[j].d100 : -1.;   // (not a candidate for this lane's state: nothing within -1)
        const double cutlim = (KIND == 1) ? h.pa : 25.;
#pragma unroll
        for (int k = 0; k < WPS; k++) {
            const double ad = fabs(W[k] - h.xnu);
            const bool useV = ((kvalid >> k) & 1u) && !(ad > cutlim) && !(ad > d100);   // modm.f90:384 / :755, :427
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(useV);
            if (mv != 0ull) {
                const int add = __popcll(mv);
                if (nq + add > 64) {
                    ms_voigt_flush<KIND>(sA, gB, gC, ms, vq, nq, W, mol, S, errflag);
                    nq = 0;
                }
                if (useV) vq[nq + __popcll(mv & ((1ull << lane) - 1ull))] = (unsigned short)(ybit | (j << 9) | (k << 6) | lane);
                nq += add;
            }
        }
    }
    if (nq > 0) ms_voigt_flush<KIND>(sA, gB, gC, ms, vq, nq, W, mol, S, errflag);
}

// the lines [j0, j1) of a molecule run in the chunk: sub-runs of ordinary lines through the assembly loops, lines with Y factors
// one by one (general_term: every per-lane condition explicit), Voigt candidates corrected afterwards
// NT: the 25 cm-1 test can fail for some (state, channel); M2: negative resonance within reach of some; V / Y: rare shapes
template <int KIND>
__device__ __forceinline__ void ms_eval_run(const MsState &st, const MsSpec &sp, const HotA *sA, const HotB *gB, const ColdLine *gC, const MsArgs &ms,
                                            unsigned long long NT, unsigned long long M2, unsigned long long V, unsigned long long Y, int j0, int j1,
                                            const double (&W)[WPS], unsigned kvalid, int mol, double (&S)[WPS], int *errflag, unsigned short *vq) {
    int j = j0;
    while (j < j1) {
        const unsigned long long ysh = Y >> j;
        if (ysh & 1ull) {  // a line with Y factors (coupled lines; amplitudes the clamp would falsify): one wavenumber at a time
            const HotA h = st.sA[j];
            const HotB b = ms_hotb(st, sp, j, h);
#pragma unroll
            for (int k = 0; k < WPS; k++) S[k] += general_term<KIND>(h, b, W[k]);
            j++;
            continue;
        }
        int len = ysh ? (int)__builtin_ctzll(ysh) : 64;
        len = min(len, j1 - j);
        unsigned addr = lds_addr(st.sA + j);
        if constexpr (KIND == 0) {
            int n = __builtin_amdgcn_readfirstlane(len);
            unsigned long long M = uni64(M2 >> j);
            if (n >= 2) ms_run_k0(addr, n, M, W, S);
            if (n == 1) {   // the odd line at the end of the sub-run
                const lds_cdp q = (lds_cdp)addr;
                const HotA h{q[0], q[1], q[2], q[3]};
                if (M & 1ull) {
#pragma unroll
                    for (int k = 0; k < WPS; k++) S[k] = uni_single<0, true, true>(h, h.pa, W[k], S[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < WPS; k++) S[k] = uni_single<0, false, true>(h, h.pa, W[k], S[k]);
                }
            }
        } else {
            // O2 / CO2 (an eighth of a line list each): the one-wavenumber loops of lines_asm.hpp, once per wavenumber of the lane.
            // Their second pedestal / limit is read BOFF bytes behind a record: 24 = the record's own pa slot (an ordinary line has pb = pa)
#pragma unroll 1
            for (int k = 0; k < WPS; k++) {
                unsigned ak = addr;
                int n = __builtin_amdgcn_readfirstlane(len);
                unsigned long long T = uni64(NT >> j), M = (KIND == 2) ? 0ull : uni64(M2 >> j);
                double w = W[0], s = S[0];
#pragma unroll
                for (int q = 1; q < WPS; q++) { w = (k == q) ? W[q] : w; s = (k == q) ? S[q] : s; }
                if (n >= 2) asm_run<KIND, 24u>(ak, n, T, M, w, s);
                if (n == 1) {
                    const unsigned cls = (unsigned)(T & 1ull) | ((unsigned)(M & 1ull) << 1);
                    const lds_cdp q = (lds_cdp)ak;
                    const HotA h{q[0], q[1], q[2], q[3]};
                    s = uni_single_any<KIND>(cls, h, h.pa, w, s);
                }
#pragma unroll
                for (int q = 0; q < WPS; q++) S[q] = (k == q) ? s : S[q];
            }
        }
        j += len;
    }
    const unsigned long long span = ((j1 >= 64) ? ~0ull : ((1ull << j1) - 1ull)) & ~((1ull << j0) - 1ull);
    const unsigned long long cand = V & span;
    if (cand) ms_voigt_scan<KIND>(cand, Y, st, sp, sA, gB, gC, ms, W, kvalid, mol, S, errflag, vq);
}

// grid = (groups of G profiles x layers); block = one wave
template <bool IBRD>
__global__ __launch_bounds__(64, 4) void lines_ms_kernel(ModmArgs a, DevLines L, DevTables tb, MsArgs ms) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    __shared__ unsigned short sVq[64];
    const int G = ms.G, LPS = ms.LPS, CL = ms.CL, nmol = a.nmol, nwn = a.nwn, nslot = ms.nslot;
    // LDS layout (launch_lines_ms sizes it)
    HotA *sA = reinterpret_cast<HotA *>(dyn_lds);                  // [G][sa_stride]
    double *sWn = reinterpret_cast<double *>(sA + G * ms.sa_stride);   // [64] the channels, ascending, the last one repeated
    double *sLay = sWn + 64;                                       // [G][20] layer scalars; [18] = T, [19] = active
    double *sW = sLay + G * 20;                                    // [G][nmol] column amounts
    double *sScor = sW + G * nmol;                                 // [G][nslot] Q(296)/Q(T)
    double *sDop = sScor + G * nslot;                              // [G][nslot] HWHM_D / Xnu
    int *sLo = reinterpret_cast<int *>(sDop + G * nslot);          // [nmol] first candidate line of the wave (union over its states)
    int *sOff = sLo + nmol;                                        // [nmol + 1] prefix sums of the candidate counts
    int *sSlot = sOff + nmol + 1;                                  // [nmol + 1] slot of (molecule, isotopologue 1)
    unsigned char *sFlag = reinterpret_cast<unsigned char *>(sSlot + nmol + 1);   // [nsteps * 64] class flags per item

    const int lane = threadIdx.x;
    const int npg = ms.npg;
    const int lay = a.nlay_max - 1 - (int)blockIdx.x / npg;   // top layer first (the long prepare stages start early)
    const int pg = (int)blockIdx.x % npg;
    // ---- the lane in the evaluate stage: state se, channels ce + LPS k ------------------------------------------------------------
    const int se_raw = lane / LPS;
    const bool lane_in = se_raw < G;
    const int se = lane_in ? se_raw : 0, ce = lane_in ? lane - se_raw * LPS : 0;
    const int prof_e = pg * G + se;
    const bool prof_ok = lane_in && prof_e < a.nprof;
    const int nl_e = prof_ok ? a.nlay[prof_e] : 0;
    const bool act_e = prof_ok && lay < nl_e;
    const size_t pl_e = (size_t)(prof_ok ? prof_e : 0) * a.nlay_max + lay;
    double *obm = static_cast<double *>(a.O_BY_MOL) + pl_e * nmol * (size_t)nwn;
    unsigned kvalid = 0u;   // bit k: channel ce + LPS k exists
#pragma unroll
    for (int k = 0; k < WPS; k++) kvalid |= (unsigned)(ce + LPS * k < nwn) << k;
    if (!prof_ok) kvalid = 0u;

    // layers beyond nlay[p] are zeroed here (modm.f90:314); argument checks as in lines_kernel
    if (prof_ok && lay >= nl_e) {
#pragma unroll
        for (int k = 0; k < WPS; k++)
            if ((kvalid >> k) & 1u)
                for (int m = 0; m < nmol; m++) obm[(size_t)m * nwn + ce + LPS * k] = 0.;
    }
    if (lay == 0) {
        if (prof_ok && ce == 0 && (nl_e < 1 || nl_e > a.nlay_max)) atomicOr(a.errflag, ERRBIT_ARG);
        if (pg == 0)
            for (int i = lane; i + 1 < nwn; i += 64) {
                const double w0 = a.wn[i], w1 = a.wn[i + 1];
                if (w1 < w0) atomicOr(a.errflag, ERRBIT_ARG);  // modm.f90:180-181
                if (a.dvset != 0. && !(fabs(w1 - (a.wn[0] + (double)(i + 1) * a.dvset)) <= 0.25 * fabs(a.dvset))) atomicOr(a.errflag, ERRBIT_ARG);
            }
    }
    if (__builtin_amdgcn_ballot_w64(act_e) == 0ull) return;   // no state of the wave has this layer

    // ---- prologue: the layer scalars of every state (INITI + head of LINES: modm.f90:868-883, :301-314; the expressions of
    // lines_kernel, every lane for its own state, lane ce == 0 of a state writes them) --------------------------------------------
    sWn[lane] = a.wn[min(lane, nwn - 1)];
    if (lane <= nmol) sSlot[lane] = ms.slot_base[lane];   // (device array: a per-lane index into the kernel arguments would go through scratch)
    {
        const double Pk = act_e ? static_cast<const double *>(a.P)[pl_e] : K_P0, Tk = act_e ? static_cast<const double *>(a.T)[pl_e] : K_T0,
                     wbrod = act_e ? static_cast<const double *>(a.WBRODL)[pl_e] : 1.;
        const double *wk = static_cast<const double *>(a.WKL) + pl_e * nmol;
        const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
        const double XN0 = (K_P0 / (K_BOLTZ * K_T0)) * 1.E+3;
        const double Xn = (Pk / (K_BOLTZ * Tk)) * 1.E+3;
        double WTOT = 0.;
        for (int m = 0; m < nmol; m++) WTOT += act_e ? wk[m] : 0.;
        WTOT = WTOT + wbrod;
        const double RP = Pk / K_P0, RP2 = RP * RP;
        const double RT = Tk / K_T0, RHORAT = Xn / XN0;
        const int ILC = (Tk < 250.0) ? 1 : ((Tk < 296.0) ? 2 : 3);  // TEMPLC = 200,250,296,340
        const double tlo = (ILC == 1) ? 200.0 : (ILC == 2 ? 250.0 : 296.0);
        const double RECTLC = (ILC == 1) ? 1.0 / (250.0 - 200.0) : ((ILC == 2) ? 1.0 / (296.0 - 250.0) : 1.0 / (340.0 - 296.0));
        const double TMPDIF = Tk - tlo;
        const double lnRT = log(RT);
        const double cTk = RADCT / Tk, cT0 = RADCT / K_T0, dTinv = 1.0 / K_T0 - 1.0 / Tk;
        if (lane_in) {
            double *ly = sLay + se * 20;
            if (ce == 0) {
                ly[0] = RHORAT; ly[1] = RP; ly[2] = RP2; ly[3] = lnRT; ly[4] = cTk; ly[5] = cT0; ly[6] = dTinv;
                ly[7] = RECTLC; ly[8] = TMPDIF; ly[9] = WTOT; ly[17] = (double)ILC; ly[18] = Tk; ly[19] = act_e ? 1. : 0.;
            }
            for (int j = ce; j < MXBRD; j += LPS) ly[10 + j] = RHORAT * ((act_e && j < nmol) ? wk[j] : 0.) / WTOT;  // rho_molec(1:7), modm.f90:313
            for (int m = ce; m < nmol; m += LPS) sW[se * nmol + m] = act_e ? wk[m] : 0.;
        }
        // MODM calls TIPS_2003 for every layer and all nmol molecules (modm.f90:250): the layer temperature alone decides the stop
        if (act_e && ce == 0 && (Tk < 70. || Tk > 3000.)) atomicOr(a.errflag, ERRBIT_TEMP);
    }
    if (lane < nmol) { sLo[lane] = 0x7fffffff; sOff[lane + 1] = 0; }
    // null records behind every state's chunk (the read-ahead of the class loops runs two records past a run)
    for (int i = lane; i < 2 * G; i += 64) sA[(i >> 1) * ms.sa_stride + CL + (i & 1)] = HotA{0., 1., 0., 0.};
    ms_sync();

    // ---- candidate window of every (state, molecule); the wave walks the union -----------------------------------------------------
    for (int i = lane; i < G * nmol; i += 64) {
        const int s = i / nmol, m = i - s * nmol, mol = m + 1;
        const double *ly = sLay + s * 20;
        if (ly[19] == 0.) continue;
        const double wkq = sW[s * nmol + m];
        int lo = L.mol_start[m + 1], hi = L.mol_start[m + 2];
        if (wkq == 0.) continue;  // W_SPECIES == 0 -> OL = 0 (modm.f90:318-321): nothing to walk for this state
        const double RHORAT = ly[0], WTOT = ly[9], Tk = ly[18];
        // (a coupled O2 list ignores the rule, an unsorted molecule cannot be searched, a state with a NaN keeps every line: lines_kernel)
        if ((mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull) && WTOT == WTOT && RHORAT == RHORAT && Tk == Tk) {
            const double pad = L.max_abs_shift * fmax(RHORAT, 1.0) + 1e-6;
            const double vlo = sWn[0] - 25.0 - pad, vhi = sWn[63] + 25.0 + pad;
            int l0 = lo, l1 = hi;
            while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (L.vnu[mid] < vlo) l0 = mid + 1; else l1 = mid; }
            const int first = l0;
            l1 = hi;
            while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (L.vnu[mid] <= vhi) l0 = mid + 1; else l1 = mid; }
            lo = first;
            hi = l0;
        }
        if (hi > lo) {
            atomicMin(&sLo[m], lo);
            atomicMax(&sOff[m + 1], hi);
        }
    }
    ms_sync();
    if (lane == 0) {
        int acc = 0;
        sOff[0] = 0;
        for (int m = 0; m < nmol; m++) {
            const int cnt = (sOff[m + 1] > sLo[m]) ? sOff[m + 1] - sLo[m] : 0;
            acc += cnt;
            sOff[m + 1] = acc;
        }
    }
    ms_sync();
    const int total = __builtin_amdgcn_readfirstlane(sOff[nmol]);
    // ---- TIPS + Doppler factor per (state, molecule, isotopologue of the table): src/tips_2003.f90:60-296, src/modm.f90:442-454 -----
    for (int i = lane; i < G * nslot; i += 64) {
        const int s = i / nslot, slot = i - s * nslot;
        int m = 0;
        while (m + 1 < nmol && sSlot[m + 1] <= slot) m++;
        const int mol = m + 1, iso = slot - sSlot[m] + 1;
        const double *ly = sLay + s * 20;
        double sc = 0., dop = 0.;
        if (ly[19] != 0. && sOff[m + 1] > sOff[m]) {
            const double Tk = ly[18];
            if (!(Tk < 70. || Tk > 3000.)) {
                bool bad = false;
                sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
                if (bad) atomicOr(a.errflag, ERRBIT_TEMP);
            }
            const double M = tb.smass[(mol - 1) * 9 + iso - 1];
            if (M > 0.) dop = doppler_factor(M, Tk);
        }
        sScor[i] = sc;
        sDop[i] = dop;
    }
    // molecules of which the wave walks no line: OL = 0 (modm.f90:314, :318-321)
    if (act_e)
        for (int m = 0; m < nmol; m++)
            if (sOff[m + 1] == sOff[m]) {
#pragma unroll
                for (int k = 0; k < WPS; k++)
                    if ((kvalid >> k) & 1u) obm[(size_t)m * nwn + ce + LPS * k] = 0.;
            }
    ms_sync();

    const double Te = sLay[se * 20 + 18];
    // the records of the rare shapes of a chunk (HotB + ColdLine per item) in this workgroup's scratch
    const size_t nitem = (size_t)G * CL;
    HotB *gB = reinterpret_cast<HotB *>(static_cast<char *>(ms.scratch) + (size_t)blockIdx.x * nitem * (sizeof(HotB) + sizeof(ColdLine)));
    ColdLine *gC = reinterpret_cast<ColdLine *>(gB + nitem);
    MsState st{sA + se * ms.sa_stride, gB + se * CL, gC + se * CL, se * CL};

    double S[WPS];
#pragma unroll
    for (int k = 0; k < WPS; k++) S[k] = 0.;
    bool osum_first = true;   // (wave-uniform: molecules complete in the same order for every state)
    int mchunk = 0;
    const int nchunks = max(1, (total + CL - 1) / CL);
    const int fair_t1 = (nchunks + 3) >> 2, fair_t2 = (2 * nchunks + 3) >> 2, fair_t3 = (3 * nchunks + 3) >> 2;
    for (int base = 0, ck = 0; base < total; base += CL, ck++) {
        if (a.fair) {   // progress-ordered wave priorities (lines_kernel.hip)
            const int q = (ck >= fair_t1) + (ck >= fair_t2) + (ck >= fair_t3);
            if (q <= 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ================= prepare: one lane per (state, line) item, ms.nsteps passes of 64 ============================================
        while (mchunk + 1 < nmol && sOff[mchunk + 1] <= base) mchunk++;
        MsSpec sp;
#pragma unroll
        for (int t = 0; t < MS_MAXSTEPS; t++) sp.w[t] = 0ull;
#pragma unroll 1
        for (int t = 0; t < ms.nsteps; t++) {
            int ltid = lane;
            asm volatile("" : "+v"(ltid));
            const int item = t * 64 + ltid;
            const int s = (int)(((unsigned)item * (unsigned)ms.inv_cl) >> 16), l = item - s * CL;   // item = s CL + l
            const int v = base + l;
            const bool in = s < G && v < total && sLay[s * 20 + 19] != 0.;
            HotA hA{0., 1., 0., 0.};
            unsigned flag = 0u;
            bool special = false;
            if (in) {
                int m = mchunk;
                while (sOff[m + 1] <= v) m++;
                const int idx = sLo[m] + (v - sOff[m]);
                const int mol = m + 1;
                const double *ly = sLay + s * 20;
                const uint32_t meta = L.meta[idx];
                LayerScalars lys;
                lys.ILC = (int)ly[17];
                lys.RHORAT = ly[0]; lys.RP = ly[1]; lys.RP2 = ly[2]; lys.lnRT = ly[3]; lys.cTk = ly[4]; lys.cT0 = ly[5];
                lys.dTinv = ly[6]; lys.RECTLC = ly[7]; lys.TMPDIF = ly[8];
                double rho7[MXBRD];
#pragma unroll
                for (int j = 0; j < MXBRD; j++) rho7[j] = IBRD ? ly[10 + j] : 0.;
                const int iso = (meta >> 6) & 15;
                const double rho_self = (mol <= MXBRD) ? ly[10 + mol - 1] : lys.RHORAT * sW[s * nmol + mol - 1] / ly[9];
                const int sl = s * nslot + sSlot[m];
                const double XIPSF = iso ? sScor[sl + iso - 1] : 0.;
                const double dopfac = iso ? sDop[sl + iso - 1] : sDop[sl];
                LineFields lf = load_line_fields(L, idx);
                lf.meta = meta;
                const LinePhys ph = line_physics_core<IBRD>(phys_params(a, L), idx, mol, lf, lys, rho_self, rho7, XIPSF, dopfac);
                HotB hB;
                ColdLine cC;
                bool fAL, fM2, fV, fY;
                line_records<double>(a, L, idx, m, meta, ph, sW + s * nmol, sWn, 64, hA, hB, cC, fAL, fM2, fV, fY);
                flag = (fAL ? 0u : 1u) | (fM2 ? 2u : 0u) | (fV ? 4u : 0u) | (fY ? 8u : 0u);
                special = fV || fY;
                if (special) {
                    gB[item] = hB;
                    gC[item] = cC;
                }
            }
            if (s < G) sA[s * ms.sa_stride + l] = hA;
            sFlag[t * 64 + ltid] = (unsigned char)flag;
            const unsigned long long bs = __builtin_amdgcn_ballot_w64(special);
#pragma unroll
            for (int q = 0; q < MS_MAXSTEPS; q++)
                if (t == q) sp.w[q] = bs;
        }
        ms_sync_global();
        // the class of a line for the wave: the most general over its states
        unsigned long long NT, M2, V, Y;
        {
            unsigned f = 0u;
            if (lane < CL)
                for (int s = 0; s < G; s++) f |= sFlag[s * CL + lane];
            NT = __builtin_amdgcn_ballot_w64(f & 1u);
            M2 = __builtin_amdgcn_ballot_w64(f & 2u);
            V = __builtin_amdgcn_ballot_w64(f & 4u);
            Y = __builtin_amdgcn_ballot_w64(f & 8u);
        }
        // ================= evaluate: molecule by molecule, in file order ================================================================
        double W[WPS];
        {
            int lc = ce;
            asm volatile("" : "+v"(lc));
#pragma unroll
            for (int k = 0; k < WPS; k++) W[k] = sWn[min(lc + LPS * k, 63)];
        }
        for (int m = __builtin_amdgcn_readfirstlane(mchunk); m < nmol; m++) {
            const int o0 = __builtin_amdgcn_readfirstlane(sOff[m]), o1 = __builtin_amdgcn_readfirstlane(sOff[m + 1]);
            if (o1 <= base || o0 >= o1) continue;
            if (o0 >= base + CL) break;
            const int j0 = max(o0, base) - base, j1 = min(o1, base + CL) - base;
            if (o0 >= base) {
#pragma unroll
                for (int k = 0; k < WPS; k++) S[k] = 0.;
            }
            const int mol = m + 1;
            if (mol == 7) ms_eval_run<1>(st, sp, sA, gB, gC, ms, NT, M2, V, Y, j0, j1, W, kvalid, mol, S, a.errflag, sVq);
            else if (mol == 2) ms_eval_run<2>(st, sp, sA, gB, gC, ms, NT, M2, V, Y, j0, j1, W, kvalid, mol, S, a.errflag, sVq);
            else ms_eval_run<0>(st, sp, sA, gB, gC, ms, NT, M2, V, Y, j0, j1, W, kvalid, mol, S, a.errflag, sVq);
            if (o1 <= base + CL) {   // run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438)
                const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
                const double wm = sW[se * nmol + m];
#pragma unroll
                for (int k = 0; k < WPS; k++)
                    if (act_e && ((kvalid >> k) & 1u)) {
                        const int iw = ce + LPS * k;
                        const double rft = W[k] * tanh_pos((RADCT * W[k]) / (2 * Te));
                        // (a state without a column of this molecule: the reference does not walk the lines at all, modm.f90:318-321)
                        const double od = (wm == 0.) ? 0. : rft * (wm * S[k]);
                        obm[(size_t)m * nwn + iw] = od;
                        if (a.osum) {   // sum over the molecules as stored, in molecule order (modm.f90:264-269): a lane's own slot
                            double *os = a.osum + pl_e * (size_t)nwn + iw;
                            *os = osum_first ? od : *os + od;
                        }
                    }
                osum_first = false;
            }
        }
        ms_sync();
    }
    if (a.osum && osum_first && act_e) {
#pragma unroll
        for (int k = 0; k < WPS; k++)
            if ((kvalid >> k) & 1u) a.osum[pl_e * (size_t)nwn + ce + LPS * k] = 0.;
    }
}

}  // namespace

namespace monortm_dev {
size_t lines_ms_lds(const MsArgs &ms, int nmol) {
    return sizeof(HotA) * (size_t)(ms.G * ms.sa_stride) + sizeof(double) * (size_t)(64 + ms.G * 20 + ms.G * nmol + 2 * ms.G * ms.nslot) +
           sizeof(int) * (size_t)(3 * nmol + 2) + (size_t)ms.nsteps * 64 + 16;
}
size_t lines_ms_scratch(const MsArgs &ms, long long nwg) { return (size_t)nwg * ms.G * ms.CL * (sizeof(HotB) + sizeof(ColdLine)); }
void launch_lines_ms(const ModmArgs &a, const DevLines &L, const DevTables &tb, const MsArgs &ms, bool ibrd, hipStream_t s) {
    const dim3 grid((unsigned)(ms.npg * a.nlay_max));
    const size_t lds = lines_ms_lds(ms, a.nmol);
    if (ibrd) hipLaunchKernelGGL((lines_ms_kernel<true>), grid, dim3(64), lds, s, a, L, tb, ms);
    else hipLaunchKernelGGL((lines_ms_kernel<false>), grid, dim3(64), lds, s, a, L, tb, ms);
}
}  // namespace monortm_dev
