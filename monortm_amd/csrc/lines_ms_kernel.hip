// lines_ms_kernel.hip - the line sum of MODM / LINES (reference src/modm.f90:253-262, :277-440) for BATCHES OF STATES ON SPARSE
// CHANNEL SETS (configs[3]: 1024 profiles x 64 layers x 50 channels), gfx950, double precision.  Round 6; DESIGN.md section 3.1m.
//
// lines_kernel<double,1,1> gives a one-wave workgroup ONE atmospheric state (profile, layer) and a lane one channel: 50 channels
// leave 14 of 64 lanes idle in every evaluate instruction, a record read from LDS serves one evaluation per lane, and each wave
// pays the prologue (layer scalars, partition sums, candidate windows) of its state alone.  Here a one-wave workgroup takes
// G states - the same layer of G consecutive profiles - and a lane is (state, WPS = 5 channels of it):
//   * evaluate: a lane reads the prepared records of ITS state (per-lane LDS address; the class of a line is common to the
//     wave: the most general over the G states), a record that has been read serves five evaluations, 60 of 64 lanes work
//     (G = 6, 10 lanes x 5 channels a state), four one-resonance lines share one reciprocal, and a block (group of lines x slot of
//     ten consecutive channels of every state) that no line of the group can reach is jumped over (lines_ms_asm.hpp);
//   * prepare: one lane per (state, line) as before - the same functions (line_physics_core, line_records of lines_device.hpp);
//   * prologue: one pass over (state, molecule) and (state, isotopologue) items for all G states.
// The G states share ONE candidate window per molecule (the union of theirs: a line outside a state's own window lies beyond
// 25 cm-1 of every channel and adds nothing - the clamp / the EXEC mask of its class says so), so a line index means the same
// line for every lane.  No barrier anywhere: the workgroup is one wave.  Results: those of lines_kernel up to the rounding of
// the shared reciprocals (1e-15 of a term; 1e-12 of a cell's optical depth); tests/test_ms_kernel.py holds the two kernels
// together and both to the oracle.  A profile's last bits depend on the profiles that share its waves.
// Memory besides the outputs: 10 KB of LDS and 21 KB of scratch in global memory per wave (records of rare shapes, radiation terms,
// the sums of the run in progress - L2-resident), two bytes per table line (ms_reach_kernel).
#define MONORTM_EXP_SGPR_CONSTANTS 1   // (exp_prep of lines_device.hpp: see there)
#include "lines_device.hpp"
#include "lines_ms_asm.hpp"

namespace {
using namespace monortm_dev;

constexpr int WPS = MS_WPS;   // wavenumbers per lane
constexpr double MS_REACH_RHO = 2.0;   // densest state (RHORAT = density over that at 1013.25 hPa, 296 K) the slot masks allow for

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
                                       (double)c.sdep, cold_xl3(c, mol, errflag), errflag);
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
// (out of line: a rare path whose registers - the Voigt function's - must not shape the allocation of the class loops; the sums
// and wavenumbers travel through two small arrays of the caller)
template <int KIND>
__device__ __attribute__((noinline)) void ms_voigt_scan(unsigned long long cand, unsigned long long ymask, MsState st, MsSpec sp, const HotA *sA,
                                              const HotB *gB, const ColdLine *gC, MsArgs ms, const double (&W)[WPS], unsigned kvalid, int mol,
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
                                            unsigned long long NT, unsigned long long M2, unsigned long long V, unsigned long long Y,
                                            const unsigned long long (&RS)[WPS], const unsigned long long (&QS)[WPS], int j0, int j1, const double *sWn, int ce, unsigned kvalid, int mol, double *sS, bool fresh, int *errflag, unsigned short *vq) {
    // the lane's wavenumbers and the sums of the run, read here and not held across the chunk loop (ten LDS reads per run against
    // twenty registers that would be live across the prepare stage)
    double W[WPS], S[WPS];
    {
        int c = ce, ln = (int)__lane_id();
        asm volatile("" : "+v"(c), "+v"(ln));
#pragma unroll
        for (int k = 0; k < WPS; k++) {
            W[k] = sWn[min(c + ms.LPS * k, 63)];
            S[k] = fresh ? 0. : sS[k * 64 + ln];
        }
    }
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
            unsigned long long R[WPS], Q[WPS];
#pragma unroll
            for (int k = 0; k < WPS; k++) { R[k] = uni64(RS[k] >> j); Q[k] = uni64(QS[k] >> j); }
            ms_run_k0(addr, n, M, R, Q, W, S);   // (the odd last line included)
        } else {
            // O2 / CO2: their own five-wavenumber loops, always the tested forms; an ordinary line's second limit is its first
            // (pb = pa), so the record alone serves
            int n = __builtin_amdgcn_readfirstlane(len);
            unsigned long long M = (KIND == 2) ? 0ull : uni64(M2 >> j);
            if (n >= 2) {
                unsigned long long R[WPS];
#pragma unroll
                for (int k = 0; k < WPS; k++) R[k] = uni64(RS[k] >> j);
                if constexpr (KIND == 1) ms_run_k1(addr, n, M, R, W, S);
                else ms_run_k2(addr, n, R, W, S);
            }
            if (n == 1) {
                const unsigned cls = 1u | ((unsigned)(M & 1ull) << 1);
                const lds_cdp q = (lds_cdp)addr;
                const HotA h{q[0], q[1], q[2], q[3]};
#pragma unroll
                for (int k = 0; k < WPS; k++) S[k] = uni_single_any<KIND>(cls, h, h.pa, W[k], S[k]);
            }
        }
        j += len;
    }
    const unsigned long long span = ((j1 >= 64) ? ~0ull : ((1ull << j1) - 1ull)) & ~((1ull << j0) - 1ull);
    const unsigned long long cand = V & span;
    bool scan = cand != 0ull;
#ifdef MONORTM_EXPERIMENT
    if (ms.ablate == 4) scan = false;
#endif
    if (scan) {
        double Wt[WPS], St[WPS];   // (copies: the arrays handed to an out-of-line function live in memory)
#pragma unroll
        for (int k = 0; k < WPS; k++) { Wt[k] = W[k]; St[k] = S[k]; }
        ms_voigt_scan<KIND>(cand, Y, st, sp, sA, gB, gC, ms, Wt, kvalid, mol, St, errflag, vq);
#pragma unroll
        for (int k = 0; k < WPS; k++) S[k] = St[k];
    }
    {
        int ln = (int)__lane_id();
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < WPS; k++) sS[k * 64 + ln] = S[k];
    }
}

// per workgroup: HotB[G CL] (16-byte aligned), ColdLine[G CL], then three blocks of [WPS][64] doubles - radiation terms, the sums of the run
// in progress, the sum over the molecules; rounded to 16 bytes so that every workgroup's HotB array starts aligned
__host__ __device__ inline size_t ms_scratch_per_wg(int G, int CL) {
    return ((size_t)G * CL * (sizeof(HotB) + sizeof(ColdLine)) + sizeof(double) * 3 * WPS * 64 + 15) / 16 * 16;
}

// ---- LDS layout of a workgroup (launch_lines_ms sizes it: lines_ms_lds) ------------------------------------------------------------
struct MsLds {
    HotA *sA;              // [G][sa_stride] prepared records of the chunk, per state
    double *sWn;           // [64] the channels, ascending, the last one repeated
    double *sLay;          // [G][20] layer scalars; [18] = T, [19] = state has this layer
    double *sW;            // [G][nmol] column amounts
    double *sScor, *sDop;  // [G][nslot] Q(296)/Q(T), HWHM_D / Xnu per (molecule, isotopologue) of the table
    int *sLo, *sOff;       // [nmol] first candidate line of the wave (union over its states), [nmol + 1] prefix sums of the counts
    int *sSlot;            // [nmol + 1] slot of (molecule, isotopologue 1)
    unsigned long long *sMask;   // [4 + MS_MAXSTEPS + 2 WPS + 1] class masks of the chunk (NT, M2, V, Y), the items whose rare-shape records exist,
                                 // per slot k the lines that reach one of its channels (lines_ms_asm.hpp, MS_IFK), and a word that
                                 // is non-zero when a state of the wave is denser than ms_reach_kernel's margin allows; behind it per slot
                                 // the lines whose NEGATIVE resonance reaches the slot (MS_IFQ)
    unsigned char *sFlag;  // [nsteps * 64] class flags per item: bits 0-3 NT, M2, V, Y
    int *sRole;            // [64] the lane in the evaluate stage: se | ce << 8 | kvalid << 16 | profile exists << 24 | state active << 25
    double *sS;            // [WPS][64] the sums of the molecule run in progress (a lane's own slots; global memory): in registers only inside a run's walk
};
__device__ __forceinline__ MsLds ms_lds(double *dyn, int G, int sa_stride, int nmol, int nslot, double *gS = nullptr) {
    MsLds l;
    l.sA = reinterpret_cast<HotA *>(dyn);
    l.sWn = reinterpret_cast<double *>(l.sA + G * sa_stride);
    // (the sums of the run in progress are parked in the workgroup's scratch, a lane's own slots: ten global accesses per walk of a
    // run, served by the L2 - in LDS they took the 2.5 KB that a chunk of 32 lines instead of 21 needs: configs[3] 1.055 -> 1.02 ms)
    l.sS = gS;
    l.sLay = l.sWn + 64;
    l.sW = l.sLay + G * 20;
    l.sScor = l.sW + G * nmol;
    l.sDop = l.sScor + G * nslot;
    l.sMask = reinterpret_cast<unsigned long long *>(l.sDop + G * nslot);
    l.sLo = reinterpret_cast<int *>(l.sMask + 4 + MS_MAXSTEPS + 2 * WPS + 1);
    l.sOff = l.sLo + nmol;
    l.sSlot = l.sOff + nmol + 1;
    l.sRole = l.sSlot + nmol + 1;
    l.sFlag = reinterpret_cast<unsigned char *>(l.sRole + 64);
    return l;
}
// kernel arguments in the kernarg segment (ModmArgs at 0, DevLines, DevTables, MsArgs aligned behind each other: the layout is
// checked against the by-value parameters once per launch).  The out-of-line stages read them from there - a handful of scalar
// loads where they are needed - instead of receiving 600 bytes of structs in registers.
typedef const __attribute__((address_space(4))) char *kseg_t;
constexpr unsigned ms_align_up(unsigned x, unsigned a) { return (x + a - 1) / a * a; }
constexpr unsigned KA_LINES = ms_align_up((unsigned)sizeof(ModmArgs), (unsigned)alignof(DevLines));
constexpr unsigned KA_TABLES = ms_align_up(KA_LINES + (unsigned)sizeof(DevLines), (unsigned)alignof(DevTables));
constexpr unsigned KA_MS = ms_align_up(KA_TABLES + (unsigned)sizeof(DevTables), (unsigned)alignof(MsArgs));
// (in a callee the builtin returns null: the kernel hands the address over as an integer - arguments travel in vector registers - and
// the callee makes it wave-uniform again, so that its loads from the segment are scalar loads)
__device__ __forceinline__ kseg_t ms_kseg_from(unsigned long long bits) {
    return (kseg_t)(size_t)uni64(bits);
}
__device__ __forceinline__ kseg_t ms_kseg() {
    kseg_t k = (kseg_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

// the lane in the evaluate stage: state se (of G), channels ce + LPS k of it; lanes beyond G x LPS idle along with state 0
struct MsLane { int lane, se, ce, prof; unsigned kvalid; bool act; };
__device__ __forceinline__ MsLane ms_lane(const MsArgs &mc, int pg, const int *sRole) {
    MsLane l;
    l.lane = (int)__lane_id();
    asm volatile("" : "+v"(l.lane));
    const unsigned r = (unsigned)sRole[l.lane];   // (formed once, by ms_prologue)
    l.se = (int)(r & 255u);
    l.ce = (int)((r >> 8) & 255u);
    l.kvalid = (r >> 16) & 255u;
    l.prof = ((r >> 24) & 1u) ? pg * mc.G + l.se : 0;
    l.act = (r >> 25) & 1u;
    return l;
}

// ---- prologue: the layer scalars of every state (INITI + head of LINES: modm.f90:868-883, :301-314; the expressions of
// lines_kernel), the candidate window of the wave, partition sums and Doppler factors.  Returns the number of candidate lines.
// (ks: the kernel's kernarg segment - the builtin that returns it is null in a callee)
__device__ __attribute__((noinline)) int ms_prologue(const unsigned long long *sKseg, int lay, int pg) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const kseg_t ks = ms_kseg_from(*sKseg);
    const ModmArgs &a = *(const ModmArgs *)ks;
    const DevLines &L = *(const DevLines *)(ks + KA_LINES);
    const DevTables &tb = *(const DevTables *)(ks + KA_TABLES);
    const MsArgs &ms = *(const MsArgs *)(ks + KA_MS);
    const int G = ms.G, LPS = ms.LPS, CL = ms.CL, nmol = a.nmol, nwn = a.nwn, nslot = ms.nslot;
    const MsLds l = ms_lds(dyn_lds, G, ms.sa_stride, nmol, nslot, nullptr);   // (sS itself is not used here)
    const int lane = threadIdx.x;
    const int se_raw = (int)(((unsigned)lane * (unsigned)ms.inv_lps) >> 16);
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
    if (__builtin_amdgcn_ballot_w64(act_e) == 0ull) return -1;   // no state of the wave has this layer
    l.sRole[lane] = (int)((unsigned)se | ((unsigned)ce << 8) | (kvalid << 16) | ((unsigned)prof_ok << 24) | ((unsigned)act_e << 25));

    l.sWn[lane] = a.wn[min(lane, nwn - 1)];
    if (lane <= nmol) l.sSlot[lane] = ms.slot_base[lane];   // (device array: a per-lane index into the kernel arguments would go through scratch)
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
            double *ly = l.sLay + se * 20;
            if (ce == 0) {
                ly[0] = RHORAT; ly[1] = RP; ly[2] = RP2; ly[3] = lnRT; ly[4] = cTk; ly[5] = cT0; ly[6] = dTinv;
                ly[7] = RECTLC; ly[8] = TMPDIF; ly[9] = WTOT; ly[17] = (double)ILC; ly[18] = Tk; ly[19] = act_e ? 1. : 0.;
            }
            for (int j = ce; j < MXBRD; j += LPS) ly[10 + j] = RHORAT * ((act_e && j < nmol) ? wk[j] : 0.) / WTOT;  // rho_molec(1:7), modm.f90:313
            for (int m = ce; m < nmol; m += LPS) l.sW[se * nmol + m] = act_e ? wk[m] : 0.;
        }
        {   // radiation term RFT = WN tanh(hc WN / 2kT) of the lane's channels (modm.f90:436-438) -> the workgroup's scratch
            double *gR = reinterpret_cast<double *>(static_cast<char *>(ms.scratch) + (size_t)blockIdx.x * ms_scratch_per_wg(G, CL) +
                                                    (size_t)G * CL * (sizeof(HotB) + sizeof(ColdLine)));
#pragma unroll 1
            for (int k = 0; k < WPS; k++) {
                const double w = a.wn[min(ce + LPS * k, nwn - 1)];
                gR[k * 64 + lane] = w * tanh_pos((RADCT * w) / (2 * Tk));
            }
        }
        // MODM calls TIPS_2003 for every layer and all nmol molecules (modm.f90:250): the layer temperature alone decides the stop
        if (act_e && ce == 0 && (Tk < 70. || Tk > 3000.)) atomicOr(a.errflag, ERRBIT_TEMP);
    }
    if (lane < nmol) { l.sLo[lane] = 0x7fffffff; l.sOff[lane + 1] = 0; }
    // null records behind every state's chunk (the read-ahead of the class loops runs two records past a run)
    for (int i = lane; i < 2 * G; i += 64) l.sA[(i >> 1) * ms.sa_stride + CL + (i & 1)] = HotA{0., 1., 0., 0.};
    ms_sync();

    if (lane == 0) {   // (a state whose density exceeds the margin of ms_reach_kernel's masks - or is NaN: no masks for this wave)
        unsigned long long dense = 0ull;
        for (int s = 0; s < G; s++) {
            const double rh = l.sLay[s * 20];
            if (l.sLay[s * 20 + 19] != 0. && !(rh <= MS_REACH_RHO)) dense = 1ull;
        }
        l.sMask[4 + MS_MAXSTEPS + WPS] = dense;
    }
    // ---- candidate window of every (state, molecule); the wave walks the union -----------------------------------------------------
    for (int i = lane; i < G * nmol; i += 64) {
        const int s = i / nmol, m = i - s * nmol, mol = m + 1;
        const double *ly = l.sLay + s * 20;
        if (ly[19] == 0.) continue;
        const double wkq = l.sW[s * nmol + m];
        int lo = L.mol_start[m + 1], hi = L.mol_start[m + 2];
        if (wkq == 0.) continue;  // W_SPECIES == 0 -> OL = 0 (modm.f90:318-321): nothing to walk for this state
        const double RHORAT = ly[0], WTOT = ly[9], Tk = ly[18];
        // (a coupled O2 list ignores the rule, an unsorted molecule cannot be searched, a state with a NaN keeps every line: lines_kernel)
        if ((mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull) && WTOT == WTOT && RHORAT == RHORAT && Tk == Tk) {
            const double pad = L.max_abs_shift * fmax(RHORAT, 1.0) + 1e-6;
            const double vlo = l.sWn[0] - 25.0 - pad, vhi = l.sWn[63] + 25.0 + pad;
            int l0 = lo, l1 = hi;
            while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (L.vnu[mid] < vlo) l0 = mid + 1; else l1 = mid; }
            const int first = l0;
            l1 = hi;
            while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (L.vnu[mid] <= vhi) l0 = mid + 1; else l1 = mid; }
            lo = first;
            hi = l0;
        }
        if (hi > lo) {
            atomicMin(&l.sLo[m], lo);
            atomicMax(&l.sOff[m + 1], hi);
        }
    }
    ms_sync();
    if (lane == 0) {
        int acc = 0;
        l.sOff[0] = 0;
        for (int m = 0; m < nmol; m++) {
            const int cnt = (l.sOff[m + 1] > l.sLo[m]) ? l.sOff[m + 1] - l.sLo[m] : 0;
            acc += cnt;
            l.sOff[m + 1] = acc;
        }
    }
    ms_sync();
    // ---- TIPS + Doppler factor per (state, molecule, isotopologue of the table): src/tips_2003.f90:60-296, src/modm.f90:442-454 -----
    for (int i = lane; i < G * nslot; i += 64) {
        const int s = i / nslot, slot = i - s * nslot;
        int m = 0;
        while (m + 1 < nmol && l.sSlot[m + 1] <= slot) m++;
        const int mol = m + 1, iso = slot - l.sSlot[m] + 1;
        const double *ly = l.sLay + s * 20;
        double sc = 0., dop = 0.;
        if (ly[19] != 0. && l.sOff[m + 1] > l.sOff[m]) {
            const double Tk = ly[18];
            if (!(Tk < 70. || Tk > 3000.)) {
                bool bad = false;
                sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
                if (bad) atomicOr(a.errflag, ERRBIT_TEMP);
            }
            const double M = tb.smass[(mol - 1) * 9 + iso - 1];
            if (M > 0.) dop = doppler_factor(M, Tk);
        }
        l.sScor[i] = sc;
        l.sDop[i] = dop;
    }
    // molecules of which the wave walks no line: OL = 0 (modm.f90:314, :318-321)
    if (act_e)
        for (int m = 0; m < nmol; m++)
            if (l.sOff[m + 1] == l.sOff[m]) {
#pragma unroll
                for (int k = 0; k < WPS; k++)
                    if ((kvalid >> k) & 1u) obm[(size_t)m * nwn + ce + LPS * k] = 0.;
            }
    ms_sync();
    return l.sOff[nmol];
}

// ---- prepare: the chunk's lines [base, base + CL) for every state, one lane per (state, line) item, ms.nsteps passes of 64.
// Leaves the records in sA (+ HotB / ColdLine of the rare shapes in the workgroup's scratch), the class of every line for the wave
// - the most general over its states - and the items whose rare-shape records exist in sMask.
template <bool IBRD>
__device__ __forceinline__ void ms_prepare(const unsigned long long *sKseg, int base, int mchunk, int total, HotB *gB, ColdLine *gC) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const kseg_t ks = ms_kseg_from(*sKseg);
    const ModmArgs &a = *(const ModmArgs *)ks;
    const DevLines &L = *(const DevLines *)(ks + KA_LINES);
    const MsArgs &ms = *(const MsArgs *)(ks + KA_MS);
    const int G = ms.G, CL = ms.CL, nmol = a.nmol, nslot = ms.nslot;
    const MsLds ld = ms_lds(dyn_lds, G, ms.sa_stride, nmol, nslot, nullptr);   // (sS itself is not used here)
    const int lane = threadIdx.x;
#pragma unroll 1
    for (int t = 0; t < ms.nsteps; t++) {
        const int item = t * 64 + lane;
        const int s = (int)(((unsigned)item * (unsigned)ms.inv_cl) >> 16), l = item - s * CL;   // item = s CL + l
        const int v = base + l;
        const bool in = s < G && v < total && ld.sLay[s * 20 + 19] != 0.;
        HotA hA{0., 1., 0., 0.};
        unsigned flag = 0u;
        bool special = false;
        if (in) {
            int m = mchunk;
            while (ld.sOff[m + 1] <= v) m++;
            const int idx = ld.sLo[m] + (v - ld.sOff[m]);
            const int mol = m + 1;
            const double *ly = ld.sLay + s * 20;
            const uint32_t meta = L.meta[idx];
            LayerScalars lys;
            lys.ILC = (int)ly[17];
            lys.RHORAT = ly[0]; lys.RP = ly[1]; lys.RP2 = ly[2]; lys.lnRT = ly[3]; lys.cTk = ly[4]; lys.cT0 = ly[5];
            lys.dTinv = ly[6]; lys.RECTLC = ly[7]; lys.TMPDIF = ly[8];
            double rho7[MXBRD];
#pragma unroll
            for (int j = 0; j < MXBRD; j++) rho7[j] = IBRD ? ly[10 + j] : 0.;
            const int iso = (meta >> 6) & 15;
            const double rho_self = (mol <= MXBRD) ? ly[10 + mol - 1] : lys.RHORAT * ld.sW[s * nmol + mol - 1] / ly[9];
            const int sl = s * nslot + ld.sSlot[m];
            const double XIPSF = iso ? ld.sScor[sl + iso - 1] : 0.;
            const double dopfac = iso ? ld.sDop[sl + iso - 1] : ld.sDop[sl];
            LineFields lf = load_line_fields(L, idx);
            lf.meta = meta;
            HotB hB;
            ColdLine cC;
            bool fAL, fM2, fV, fY;
            // a pass without a coupled line or an air-width / air-shift conversion (meta bits 10-11, 13, 14: most passes of most
            // tables) takes the instantiation without those blocks - fifteen divergent regions less to step through
            const bool plain = ((meta >> 10) & 3u) == 0u && ((meta >> 13) & 3u) == 0u;
            // near_lb: no channel nearer to the shifted centre than this (<= 0: unknown) - the table centre's distance less the shift,
            // with a margin far above the roundings involved
            if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
                const LinePhys ph = line_physics_core<IBRD, true>(phys_params(a, L), idx, mol, lf, lys, rho_self, rho7, XIPSF, dopfac);
                const double near_lb = (double)ms.near0[idx] * (1. - 1e-6) - fabs(ph.xnu - lf.xnu0) - 1e-9;
                line_records<double, true>(a, L, idx, m, meta, ph, ld.sW + s * nmol, ld.sWn, 64, hA, hB, cC, fAL, fM2, fV, fY, near_lb);
            } else {
                const LinePhys ph = line_physics_core<IBRD>(phys_params(a, L), idx, mol, lf, lys, rho_self, rho7, XIPSF, dopfac);
                const double near_lb = (double)ms.near0[idx] * (1. - 1e-6) - fabs(ph.xnu - lf.xnu0) - 1e-9;
                line_records<double>(a, L, idx, m, meta, ph, ld.sW + s * nmol, ld.sWn, 64, hA, hB, cC, fAL, fM2, fV, fY, near_lb);
            }
            flag = (fAL ? 0u : 1u) | (fM2 ? 2u : 0u) | (fV ? 4u : 0u) | (fY ? 8u : 0u);
            special = fV || fY;
            if (special) {
                gB[item] = hB;
                gC[item] = cC;
            }
        }
        if (s < G) ld.sA[s * ms.sa_stride + l] = hA;
        ld.sFlag[item] = (unsigned char)flag;
        const unsigned long long bs = __builtin_amdgcn_ballot_w64(special);
        if (lane == 0) ld.sMask[4 + t] = bs;
    }
    ms_sync_global();
    // the class of a line for the wave: the most general over its states
    unsigned f = 0u;
    if (lane < CL)
        for (int s = 0; s < G; s++) f |= ld.sFlag[s * CL + lane];
    const unsigned long long NT = __builtin_amdgcn_ballot_w64(f & 1u), M2 = __builtin_amdgcn_ballot_w64(f & 2u),
                             V = __builtin_amdgcn_ballot_w64(f & 4u), Y = __builtin_amdgcn_ballot_w64(f & 8u);
    if (lane == 0) { ld.sMask[0] = NT; ld.sMask[1] = M2; ld.sMask[2] = V; ld.sMask[3] = Y; }
    // ... and the slots it reaches (slot k = the k-th wavenumbers of the lanes = channels k LPS .. k LPS + LPS - 1 of every state): a
    // property of the table line and the channel set up to the pressure shift, formed once per launch by ms_reach_kernel with a
    // margin for the shifts of states up to RHORAT = MS_REACH_RHO (a wave with a denser state - ms_prologue notes it in sMask's
    // last word - uses no mask); rare shapes reach every slot
    {
        unsigned r = 0u;
        const int v = base + lane;
        if (lane < CL && v < total) {
            int m = mchunk;
            while (ld.sOff[m + 1] <= v) m++;
            r = ms.reach[ld.sLo[m] + (v - ld.sOff[m])];
            if ((f & (4u | 8u)) || ld.sMask[4 + MS_MAXSTEPS + WPS] != 0ull) r = 0x1f1fu;
        }
#pragma unroll
        for (int k = 0; k < WPS; k++) {
            const unsigned long long rk = __builtin_amdgcn_ballot_w64((r >> k) & 1u), qk = __builtin_amdgcn_ballot_w64((r >> (8 + k)) & 1u);
            if (lane == 0) { ld.sMask[4 + MS_MAXSTEPS + k] = rk; ld.sMask[4 + MS_MAXSTEPS + WPS + 1 + k] = qk; }
        }
    }
    ms_sync();
}

// ---- ms_reach_kernel: per table line the slots (of LPS consecutive channels) it can reach, |WN - Xnu| <= 25 cm-1 for some channel
// of the slot with the shifted centre anywhere within max_abs_shift x MS_REACH_RHO of the table's (line_table.cpp: |Xnu - XNU0| <=
// max_abs_shift x RHORAT); a coupled O2 line (no rule, modm.f90:755-792) and a NaN centre reach every slot that holds a channel.
// One byte per line, once per launch (the channels are the caller's device array).
__global__ void ms_reach_kernel(const double *wn, int nwn, int LPS, DevLines L, int nlines, unsigned short *reach, float *near0) {
    // (the channels - at most 64 where this kernel runs - through LDS: the search below is a chain of dependent reads)
    __shared__ double swn[64];
    if (threadIdx.x < 64) swn[threadIdx.x] = wn[min((int)threadIdx.x, max(nwn, 1) - 1)];
    __syncthreads();
    wn = swn;
    nwn = min(nwn, 64);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nlines) return;
    const uint32_t meta = L.meta[idx];
    const double x = L.vnu[idx], pad = L.max_abs_shift * MS_REACH_RHO + 1e-6;
    const bool every = ((meta & 63u) == 7u && ((meta >> 10) & 3u)) || !(x == x);
    unsigned r = 0u;
    for (int k = 0; k < WPS; k++) {
        const int c0 = LPS * k;
        if (c0 >= nwn) break;
        const double wlo = wn[c0], whi = wn[min(c0 + LPS, nwn) - 1];
        if (every || (!(x - pad - 25. > whi) && !(wlo - 25. > x + pad))) r |= 1u << k;
        // bits 8 .. 12: the NEGATIVE resonance reaches the slot (WN + Xnu <= 25 for its lowest channel, modm.f90:713 / :757)
        if (every || !(wlo + (x - pad) > 25.)) r |= 256u << k;
    }
    reach[idx] = (unsigned short)r;
    // ... and the distance of the table centre to the nearest channel, rounded down: line_records' Voigt test (some channel within
    // 100 Doppler widths of the SHIFTED centre, modm.f90:427) cannot pass where this distance less the shift exceeds the limit
    // (triangle inequality), so that the search over the channels is left to the few lines near a channel.  0 (= search) for a NaN centre.
    double best = 0.;
    if (x == x && nwn > 0) {
        int lo = 0, hi = nwn;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (wn[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        best = __builtin_inf();
        if (lo < nwn) best = fabs(wn[lo] - x);
        if (lo > 0) best = fmin(best, fabs(wn[lo - 1] - x));
    }
    near0[idx] = __double2float_rd(best);
}

// grid = (groups of G profiles x layers); block = one wave
template <bool IBRD>
__global__ __launch_bounds__(64, 4) void lines_ms_kernel(ModmArgs a, DevLines L, DevTables tb, MsArgs ms) {
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    __shared__ unsigned short sVq[64];
    __shared__ unsigned long long sKseg;   // address of the kernarg segment for the out-of-line stages (the builtin is null in a callee)
    if (threadIdx.x == 0) sKseg = (unsigned long long)(size_t)__builtin_amdgcn_kernarg_segment_ptr();
    ms_sync();
    const int npg = ms.npg;
    const int lay = a.nlay_max - 1 - (int)blockIdx.x / npg;   // top layer first (the long prepare stages start early)
    const int pg = (int)blockIdx.x % npg;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the out-of-line stages read the arguments from the kernarg segment at offsets computed from the struct sizes: if a compiler
        // ever laid the segment out differently they would read garbage silently - compare with the by-value parameters once per launch
        const kseg_t ks = ms_kseg();
        const ModmArgs &ak = *(const ModmArgs *)ks;
        const DevLines &Lk = *(const DevLines *)(ks + KA_LINES);
        const DevTables &tk = *(const DevTables *)(ks + KA_TABLES);
        const MsArgs &mk = *(const MsArgs *)(ks + KA_MS);
        if (ak.wn != a.wn || ak.errflag != a.errflag || ak.nmol != a.nmol || Lk.vnu != L.vnu || Lk.meta != L.meta || Lk.mol_start[MXMOL + 1] != L.mol_start[MXMOL + 1] ||
            tk.tips_qoft != tb.tips_qoft || tk.smass != tb.smass || mk.scratch != ms.scratch || mk.inv_cl != ms.inv_cl || mk.slot_base != ms.slot_base)
            atomicOr(a.errflag, ERRBIT_ARG);
    }
    const int total = __builtin_amdgcn_readfirstlane(ms_prologue(&sKseg, lay, pg));
    if (total < 0) return;
#ifdef MONORTM_EXPERIMENT   // timing experiments (wrong results): tools/build_variant.sh
    if (ms.ablate == 1) return;
#endif

    // the records of the rare shapes of a chunk (HotB + ColdLine per item) in this workgroup's scratch
    const size_t nitem = (size_t)ms.G * ms.CL;
    HotB *gB = reinterpret_cast<HotB *>(static_cast<char *>(ms.scratch) + (size_t)blockIdx.x * ms_scratch_per_wg(ms.G, ms.CL));
    ColdLine *gC = reinterpret_cast<ColdLine *>(gB + nitem);
    const double *gR = reinterpret_cast<const double *>(gC + nitem);   // [WPS][64] radiation terms (ms_prologue)

    bool osum_first = true;   // (wave-uniform: molecules complete in the same order for every state)
    int mchunk = 0;
    const int CLk = ms.CL;
    const int nchunks = max(1, (total + CLk - 1) / CLk);
    const int fair_t1 = (nchunks + 3) >> 2, fair_t2 = (2 * nchunks + 3) >> 2, fair_t3 = (3 * nchunks + 3) >> 2;
    for (int base = 0, ck = 0; base < total; base += CLk, ck++) {
        if (a.fair) {   // progress-ordered wave priorities (lines_kernel.hip)
            const int q = (ck >= fair_t1) + (ck >= fair_t2) + (ck >= fair_t3);
            if (q <= 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // (the arguments per chunk from the kernarg segment through an opaque copy of its address: hoisted out of the chunk loop they
        // were live across both stages - SGPRs spilled to VGPR lanes)
        const kseg_t ks = ms_kseg();
        const ModmArgs &ac = *(const ModmArgs *)ks;
        const MsArgs &mc = *(const MsArgs *)(ks + KA_MS);
        const int G = mc.G, LPS = mc.LPS, CL = mc.CL, nmol = ac.nmol, nwn = ac.nwn;
        const MsLds ld = ms_lds(dyn_lds, G, mc.sa_stride, nmol, mc.nslot, const_cast<double *>(gR) + WPS * 64);
        while (mchunk + 1 < nmol && ld.sOff[mchunk + 1] <= base) mchunk++;
        mchunk = __builtin_amdgcn_readfirstlane(mchunk);
        ms_prepare<IBRD>(&sKseg, base, mchunk, total, gB, gC);

        // ================= evaluate: molecule by molecule, in file order ================================================================
        // The lane's role in this stage - state se, channels ce + LPS k - is formed where it is used (ms_lane: a handful of
        // instructions on an opaque lane id), not once per chunk: values that live across a run's walk are spilled around the
        // 48 fixed registers of the class loops.
#ifdef MONORTM_EXPERIMENT
        if (mc.ablate == 2) continue;
#endif
        MsSpec sp;
#pragma unroll
        for (int t = 0; t < MS_MAXSTEPS; t++) sp.w[t] = uni64(ld.sMask[4 + t]);
        const unsigned long long NT = uni64(ld.sMask[0]), M2 = uni64(ld.sMask[1]), V = uni64(ld.sMask[2]), Y = uni64(ld.sMask[3]);
        unsigned long long RS[WPS], QS[WPS];
#pragma unroll
        for (int k = 0; k < WPS; k++) { RS[k] = uni64(ld.sMask[4 + MS_MAXSTEPS + k]); QS[k] = uni64(ld.sMask[4 + MS_MAXSTEPS + WPS + 1 + k]); }
        for (int m = mchunk; m < nmol; m++) {
            const int o0 = __builtin_amdgcn_readfirstlane(ld.sOff[m]), o1 = __builtin_amdgcn_readfirstlane(ld.sOff[m + 1]);
            if (o1 <= base || o0 >= o1) continue;
            if (o0 >= base + CL) break;
            const int j0 = max(o0, base) - base, j1 = min(o1, base + CL) - base;
            const int mol = m + 1;
            {
                const MsLane ln = ms_lane(mc, pg, ld.sRole);
                const MsState st{ld.sA + ln.se * mc.sa_stride, gB + ln.se * CL, gC + ln.se * CL, ln.se * CL};
#ifdef MONORTM_EXPERIMENT
                if ((mc.ablate == 3 && (mol == 7 || mol == 2)) || (mc.ablate == 5 && mol != 7 && mol != 2)) continue;
#endif
                if (mol == 7) ms_eval_run<1>(st, sp, ld.sA, gB, gC, mc, NT, M2, V, Y, RS, QS, j0, j1, ld.sWn, ln.ce, ln.kvalid, mol, ld.sS, o0 >= base, ac.errflag, sVq);
                else if (mol == 2) ms_eval_run<2>(st, sp, ld.sA, gB, gC, mc, NT, M2, V, Y, RS, QS, j0, j1, ld.sWn, ln.ce, ln.kvalid, mol, ld.sS, o0 >= base, ac.errflag, sVq);
                else ms_eval_run<0>(st, sp, ld.sA, gB, gC, mc, NT, M2, V, Y, RS, QS, j0, j1, ld.sWn, ln.ce, ln.kvalid, mol, ld.sS, o0 >= base, ac.errflag, sVq);
            }
            if (o1 <= base + CL) {   // run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438)
                const MsLane ln = ms_lane(mc, pg, ld.sRole);
                const double wm = ld.sW[ln.se * nmol + m];
                const size_t pl_e = (size_t)ln.prof * ac.nlay_max + lay;
                double *obm = static_cast<double *>(ac.O_BY_MOL) + (pl_e * nmol + m) * (size_t)nwn;
                // sum over the molecules as stored, in molecule order (modm.f90:264-269), accumulated in the workgroup's scratch (a lane's
                // own slots, L2-resident) and stored once at the end of the wave
                double *os = const_cast<double *>(gR) + 2 * WPS * 64;
#pragma unroll
                for (int k = 0; k < WPS; k++)
                    if (ln.act && ((ln.kvalid >> k) & 1u)) {
                        const int iw = ln.ce + LPS * k;
                        const double rft = gR[k * 64 + ln.lane];   // radiation term of (state, channel): formed once, in the prologue
                        // (a state without a column of this molecule: the reference does not walk the lines at all, modm.f90:318-321)
                        const double od = (wm == 0.) ? 0. : rft * (wm * ld.sS[k * 64 + ln.lane]);
                        obm[iw] = od;
                        if (ac.osum) os[k * 64 + ln.lane] = osum_first ? od : os[k * 64 + ln.lane] + od;
                    }
                osum_first = false;
            }
        }
        ms_sync();
    }
    if (a.osum) {
        const int lane = threadIdx.x;
        const int se_raw = (int)(((unsigned)lane * (unsigned)ms.inv_lps) >> 16);
        const int se = se_raw < ms.G ? se_raw : 0, ce = se_raw < ms.G ? lane - se_raw * ms.LPS : 0;
        const int prof_e = pg * ms.G + se;
        const MsLds ld = ms_lds(dyn_lds, ms.G, ms.sa_stride, a.nmol, ms.nslot, nullptr);
        const double *os = gR + 2 * WPS * 64;
        if (se_raw < ms.G && prof_e < a.nprof && ld.sLay[se * 20 + 19] != 0.) {
#pragma unroll
            for (int k = 0; k < WPS; k++)
                if (ce + ms.LPS * k < a.nwn)
                    a.osum[((size_t)prof_e * a.nlay_max + lay) * (size_t)a.nwn + ce + ms.LPS * k] = osum_first ? 0. : os[k * 64 + lane];
        }
    }
}

}  // namespace

namespace monortm_dev {
size_t lines_ms_lds(const MsArgs &ms, int nmol) {
    return sizeof(HotA) * (size_t)(ms.G * ms.sa_stride) + sizeof(double) * (size_t)(64 + ms.G * 20 + ms.G * nmol + 2 * ms.G * ms.nslot) +
           sizeof(unsigned long long) * (size_t)(4 + MS_MAXSTEPS + 2 * MS_WPS + 1) + sizeof(int) * (size_t)(3 * nmol + 2 + 64) + (size_t)ms.nsteps * 64 + 16;
}
size_t lines_ms_scratch(const MsArgs &ms, long long nwg) { return (size_t)nwg * ms_scratch_per_wg(ms.G, ms.CL); }
void launch_lines_ms(const ModmArgs &a, const DevLines &L, const DevTables &tb, const MsArgs &ms, bool ibrd, hipStream_t s) {
    const int nlines = L.mol_start[MXMOL + 1];
    if (nlines > 0) hipLaunchKernelGGL(ms_reach_kernel, dim3((nlines + 255) / 256), dim3(256), 0, s, a.wn, a.nwn, ms.LPS, L, nlines, ms.reach, ms.near0);
    const dim3 grid((unsigned)(ms.npg * a.nlay_max));
    const size_t lds = lines_ms_lds(ms, a.nmol);
    if (ibrd) hipLaunchKernelGGL((lines_ms_kernel<true>), grid, dim3(64), lds, s, a, L, tb, ms);
    else hipLaunchKernelGGL((lines_ms_kernel<false>), grid, dim3(64), lds, s, a, L, tb, ms);
}
}  // namespace monortm_dev
