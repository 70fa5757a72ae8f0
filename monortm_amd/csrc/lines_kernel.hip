// lines_kernel.hip - the line sum of MODM / LINES (reference src/modm.f90:253-262, :277-440) for gfx950.
// See DESIGN.md section 3.1.
#include "lines_device.hpp"

namespace {
using namespace monortm_dev;

// ------------------------------------------------------------------------------------------------
// lines_kernel: O_BY_MOL(wn, mol, layer) = RFT * W_mol * sum_lines S~ * shape      (modm.f90:253-262)
// grid = (wavenumber tiles x line slices, layers, profiles); block = NW waves; lane = WPL wavenumbers (tile = WPL x NW x 64)
// ------------------------------------------------------------------------------------------------
// IBRD: species-by-species broadening data are read (IBRD != 0 and the file carries any); a separate
// instantiation keeps its ~25 VGPRs out of the common kernel.  With one wavenumber per lane it also fits 128 VGPRs with two
// spilled registers (4 waves per SIMD: c4brd 0.278 -> 0.257 ms); with two wavenumbers per lane it keeps 3 waves
// R: double (real_kind 8) or float (real_kind 4: float I/O and float evaluation of the Lorentz fast path; the prepare
// stage and the rare coupled / Voigt shapes stay double)
// Synchronisation of the workgroup's LDS traffic.  A one-wave workgroup needs no barrier and - more to the point - no fence at
// workgroup scope: __syncthreads() waits for every outstanding GLOBAL access of the wave too (s_waitcnt vmcnt(0): the O_BY_MOL
// stores of a finished molecule run, the table loads read ahead for the next chunk), once or twice per chunk.  The LDS unit
// takes the DS instructions of one wave in program order, so ordering the compiler is all that is needed.
template <int NW>
__device__ __forceinline__ void tile_sync() {
    if constexpr (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// PLANT: a one-wave two-wavenumber tile that takes its candidate runs and far field from far_kernel (dense grids; the tile has no
// far field of its own).  A separate instantiation: the tiles of sparse channel sets keep their registers.
template <typename R, int NW, int WPL, bool IBRD, bool PLANT = false>
__global__ __launch_bounds__(NW * 64, (IBRD && WPL >= 2) ? 3 : 4) void lines_kernel(ModmArgs a, DevLines L, DevTables tb) {
    // kernarg layout (checked against the code object's metadata): ModmArgs at 0, DevLines right behind it
    constexpr unsigned KARG_LINES = (unsigned)((sizeof(ModmArgs) + alignof(DevLines) - 1) / alignof(DevLines) * alignof(DevLines));
    constexpr int NT = NW * 64;   // threads = lines per chunk
    constexpr int TW = NT * WPL;  // wavenumbers per tile: lane tid owns tile positions tid, tid + NT, ...
    constexpr bool SGL = sizeof(R) == 4;
    using Hot = typename HotOf<R>::type;
    // one object: sB sits at a fixed positive distance behind sA (eval_unified reads both at immediate offsets from one address
    // register); + 2: its read-ahead may run two records past a run
#ifdef MONORTM_LDS_ROOMY
    constexpr int RA = 2;
#else
    constexpr int RA = SGL ? 0 : 2;  // (single precision has no assembly loops and no read-ahead)
#endif
    __shared__ struct { Hot a[NT + RA]; HotB b[NT + RA]; } sRec;
    Hot *const sA = sRec.a;
    HotB *const sB = sRec.b;
    __shared__ double sWn[TW];  // the tile's wavenumbers (ascending)
    // radiation term and the sum over the molecules of O_BY_MOL per wavenumber of the tile: touched once per molecule run, so
    // they sit here instead of in registers that stay live across the evaluate loops (the assembly loops of lines_asm.hpp pin
    // 56 registers; with these four in VGPRs the prepare stage went to scratch: 89 scratch accesses per wave)
    constexpr bool LDS_STATE = WPL == 1 && !SGL;  // (the wider tiles have neither the assembly loops nor LDS to spare)
    __shared__ double sRft[LDS_STATE ? TW : 1], sOsum[LDS_STATE ? TW : 1];
    __shared__ double sLay[20];  // layer scalars: parked here so they do not occupy registers during the evaluate loops
    // per chunk parity and wave of the prepare stage, one bit per line: every lane of the tile within 25 cm-1 / negative
    // resonance within reach of some lane / Voigt candidate for this tile / shape with line-coupling Y factors
    __shared__ unsigned long long sAL[2][NW], sM2[2][NW], sVg[2][NW], sYf[2][NW];
    // far field (two wavenumbers per lane = dense grids): per chunk parity and wave the lines moved into the moments; per
    // wave and molecule parity the moments themselves (two consecutive molecules can be open at a time).  Moments and the
    // polynomial are double in both builds; the single-precision build adds the rounded polynomial to its float sums.
    // (round 5: tiles of several waves only.  On a one-wave tile a line is evaluated by ONE wave, 2-4 times a lane: the series of its
    // 64 lines cost that wave as much as the evaluations they replace - measured with the Chebyshev sums at every distance from 1.3
    // to 4 tile half-widths and without: configs[4] whole 1.116 / 1.101 / 1.029 / 1.018 ms against 1.010 without, its 32-profile share
    // 0.190 ... 0.169 against 0.150)
    constexpr bool FAR = WPL >= 2 && NW >= 2;
    constexpr bool PLAN = FAR || PLANT;   // far_kernel may serve the tile
    static_assert(!PLANT || (NW == 1 && WPL == 2), "PLANT: the one-wave two-wavenumber tile");
    __shared__ unsigned long long sFar[2][NW];
    __shared__ int sAllFar[2][NW];  // ... and per preparing wave: every line of its 64 went into the sums (or lies past the slice)
    __shared__ unsigned long long sFull[2][NW];  // single precision: two-resonance lines within reach of every wavenumber of the tile
    // dense grids in double precision: per chunk parity and preparing wave, least and largest centre among its plain tested lines
    // (eval_dispatch skips them for the (wave, k) pairs they cannot reach)
    constexpr bool EDGE = WPL >= 2;   // (one-wave tiles keep the two numbers in scalar registers: kTst below)
    __shared__ double sTst[(EDGE && NW > 1) ? 2 : 1][(EDGE && NW > 1) ? NW : 1][2];
    constexpr int FARP = far_p(NW * WPL);            // moments per molecule parity / least distance in tile half-widths:
    constexpr double FARK = far_kappa(NW * WPL);     // by the evaluations a workgroup makes per line (lines_device.hpp)
    __shared__ double sMom[FAR ? NW : 1][2][FAR ? FARP + 1 : 1];
    __shared__ int sMomUsed[2];  // per molecule parity: moments were added since the slot was cleared
    __shared__ ColdLine sCold[NT];
    // per wave and wavenumber of the lane: queued (line, lane) pairs that take a Voigt shape (four wavenumbers per lane: the two
    // passes of a chunk empty their queues before they return and share the entries)
    // (+ 64 in double precision: the queue of voigt_scan, lines_device.hpp - one for all wavenumbers of the lane, behind the others)
#ifdef MONORTM_LDS_ROOMY
    __shared__ unsigned short sVq[NW][(WPL >= 4 ? 2 : WPL) * 64 + 192];
#else
    __shared__ unsigned short sVq[NW][(WPL >= 4 ? 2 : WPL) * 64 + (SGL ? 0 : 64)];
#endif
    // per-molecule tables sized by nmol (dynamic LDS, lines_dyn_lds()): a 64-thread block must stay under
    // ~8 KB of LDS or the 160 KB of a CU, not the registers, limit the resident waves
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double *sScor = dyn_lds;                     // [nmol*9] Q(296)/Q(T) per (mol, iso)
    double *sDop = sScor + a.nmol * 9;           // [nmol*9] HWHM_D / Xnu per (mol, iso)
    double *sW = sDop + a.nmol * 9;              // [nmol]   column amounts
    int *sLo = reinterpret_cast<int *>(sW + a.nmol);  // [nmol]   first candidate line
    int *sOff = sLo + a.nmol;                    // [nmol+1] prefix sums of the candidate counts
    // far field formed by far_kernel (a.farseg != null): the candidates of a molecule are up to FAR_SEGS runs of table lines
    int *sSegBase = sOff + a.nmol + 1;           // [nmol*FAR_SEGS] first line of run k - candidates of the molecule before it
    int *sSegCum = sSegBase + a.nmol * FAR_SEGS; // [nmol*FAR_SEGS] candidates of the molecule up to and including run k

    const int tid = threadIdx.x;
    const int nslice = a.nslice;
#ifdef LINES_TIMING
    long long tq0 = (long long)__builtin_readcyclecounter(), tqP = 0, tqE = 0, tqx;
    const long long trt0 = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz, the same counter on every CU
    int nFar = 0, nAL = 0, nM2 = 0, nV = 0;
#endif
    // dispatch order = x, then y, then z: layers are the slowest index and the TOP layer comes first - the layers whose prepare
    // stage is longest (low pressure: Voigt proximity searches) start in the first round of workgroups, the uniform
    // lower layers fill the last round, so the grid drains evenly (c4shard: 8192 workgroups over 4096 resident slots)
    // Placement: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one L2), so the ntile tiles that walk
    // the SAME records (one layer, profile and slice) are given to one XCD back to back - they stream through the slice together
    // and one of them fetches a record from HBM for all (c3: 20 tiles per slice). Speed only: any placement gives the same sums.
    // With one tile per group (c4, c5) the map is the identity.
    const int ntile = (int)gridDim.x / nslice;
    int tile, slice, prof, lay;
    if (ntile == 1) {  // (no integer divisions on the sparse-channel grids: every instruction of the prologue is paid per wave)
        tile = 0;
        slice = (int)blockIdx.x;
        prof = (int)blockIdx.y;
        lay = (int)gridDim.z - 1 - (int)blockIdx.z;
    } else {
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned ngroups = (unsigned)nslice * gridDim.y * gridDim.z, main_wg = (ngroups & ~7u) * (unsigned)ntile;
        unsigned grp, t;
        if (L < main_wg) {
            const unsigned q = L >> 3;
            grp = (q / (unsigned)ntile) * 8u + (L & 7u);
            t = q % (unsigned)ntile;
        } else {
            grp = L / (unsigned)ntile;
            t = L % (unsigned)ntile;
        }
        tile = (int)t;
        slice = (int)(grp % (unsigned)nslice);
        const unsigned pz = grp / (unsigned)nslice;
        prof = (int)(pz % gridDim.y);
        lay = (int)gridDim.z - 1 - (int)(pz / gridDim.y);
    }
    const int nwn = a.nwn, nmol = a.nmol;
    int iwk[WPL];
    bool validk[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) {
        iwk[k] = tile * TW + k * NT + tid;
        validk[k] = iwk[k] < nwn;
    }
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    R *obm = (nslice == 1) ? wp<R>(a.O_BY_MOL) + pl * nmol * (size_t)nwn
                           : wp<R>(a.partial) + ((size_t)slice * a.nprof * a.nlay_max + pl) * nmol * (size_t)nwn;

    // ---- prologue, stage 0: every load whose address follows from the block index alone is issued HERE, before anything is
    // waited for.  (Round 4: a `-DLINES_TIMING` build showed the prologue at 30 k of a configs[3] workgroup's 190 k cycles for
    // 7 % of its instructions - a chain of eight dependent round trips: nlay, the layer's state, the tile's ends, mol_start,
    // the run's end lines, then ISONM -> offset -> the four TIPS nodes.  Now: stage 0 (this block), stage 1 (what needs the
    // temperature or mol_start), and the first chunk's table fields.)
    static_assert(NT >= 64 && MXMOL < 64, "one lane per molecule in the window search");
    const int nl = a.nlay[prof];
    double WNk[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) WNk[k] = a.wn[validk[k] ? iwk[k] : nwn - 1];
    const double Pk = rp<R>(a.P)[pl], Tk = rp<R>(a.T)[pl], wbrod = rp<R>(a.WBRODL)[pl];
    const R *wk = rp<R>(a.WKL) + pl * nmol;
    const double wnlo = a.wn[tile * TW], wnhi = a.wn[min(nwn, (tile + 1) * TW) - 1];
    const int mq = tid < nmol ? tid : nmol - 1;     // the molecule whose candidate range this lane finds
    const double wkq = (double)wk[mq];
    const int msq0 = L.mol_start[mq + 1], msq1 = L.mol_start[mq + 2];
    const int tq = tid < nmol * 9 ? tid : 0;         // the (molecule, isotopologue) slot of this lane in the first TIPS pass
    const int isnq = tb.tips_isonm[tq / 9], offq = tb.tips_offset[tq / 9];
    const double Mq = tb.smass[tq];

    // outputs start from zero (modm.f90:314): layers beyond nlay[p] are zeroed here; inside the profile only the molecules
    // whose run is not walked by this block are (below, once the candidate ranges are known) - the others are written once,
    // when their run is complete
    if (lay >= nl) {
#pragma unroll
        for (int k = 0; k < WPL; k++)
            if (validk[k])
                for (int m = 0; m < nmol; m++) obm[(size_t)m * nwn + iwk[k]] = (R)0;
    }
    // arguments that live in device memory cannot be validated by the host side of a *_dev call: flag them here
    if (lay == 0 && slice == 0) {
        if (tile == 0 && tid == 0 && (nl < 1 || nl > a.nlay_max)) atomicOr(a.errflag, ERRBIT_ARG);
        if (tile == 0 && tid == 0 && prof == 0) {
            // the chunk loop re-reads the arguments from the kernarg segment at offsets computed from the struct sizes (ModmArgs at
            // 0, DevLines aligned behind it): if a compiler ever laid the segment out differently the table pointers read there
            // would be wrong silently - compare them with the by-value parameters once per launch
            const __attribute__((address_space(4))) char *ks = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
            const ModmArgs &ak = *(const ModmArgs *)ks;
            const DevLines &Lk = *(const DevLines *)(ks + KARG_LINES);
            if (ak.wn != a.wn || ak.errflag != a.errflag || ak.phys != a.phys || Lk.vnu != L.vnu || Lk.meta != L.meta || Lk.brd_dat != L.brd_dat ||
                Lk.mol_start[MXMOL + 1] != L.mol_start[MXMOL + 1])
                atomicOr(a.errflag, ERRBIT_ARG);
        }
        if (prof == 0) {
#pragma unroll
            for (int k = 0; k < WPL; k++)
                if (iwk[k] + 1 < nwn) {
                    const double w0 = a.wn[iwk[k]], w1 = a.wn[iwk[k] + 1];
                    if (w1 < w0) atomicOr(a.errflag, ERRBIT_ARG);  // modm.f90:180-181
                    // DVSET /= 0 promises the grid V1 + i DVSET (the continuum interpolation and the nearest-wavenumber lookup of
                    // line_records rely on it, as the reference's CONTNM call does): the cumulative drift stays below a quarter
                    // step (the sgl driver's REAL*4 product (J-1)*DVSET moves point J by 6e-8 J DVSET, src/monortm_sub.F90:287)
                    if (a.dvset != 0. && !(fabs(w1 - (a.wn[0] + (double)(iwk[k] + 1) * a.dvset)) <= 0.25 * fabs(a.dvset)))
                        atomicOr(a.errflag, ERRBIT_ARG);
                }
        }
    }
    if (lay >= nl) return;

    // MODM calls TIPS_2003 for every layer and all nmol molecules whatever the line file holds (modm.f90:250), and each QT_*
    // routine returns -1 outside 70-3000 K -> STOP (tips_2003.f90:272-277): the layer temperature alone decides
    const bool t_bad = Tk < 70. || Tk > 3000.;
    if (t_bad && tile == 0 && slice == 0 && tid == 0) atomicOr(a.errflag, ERRBIT_TEMP);
    // ---- stage 1: the first and last line of this lane's molecule (decides whether the tile keeps the whole run) and the
    // four TIPS nodes + Q(296) of this lane's isotopologue; they travel while the layer scalars are formed
    const bool runq = tid < nmol && msq1 > msq0;
    double vq0 = 0., vq1 = 0.;
    if (runq) {
        vq0 = L.vnu[msq0];
        vq1 = L.vnu[msq1 - 1];
    }
    const int molq = tq / 9 + 1, isoq = tq % 9 + 1;
    const TipsNodes tnq = tips_nodes(Tk);
    const bool tipsq = tid < nmol * 9 && !t_bad && isoq <= min(9, isnq) && molq != 34 && molq != 39 && !tnq.none;
    double bq0 = 0., bq1 = 0., bq2 = 0., bq3 = 0., q296q = 0.;
    if (tipsq) {
        const int slot = offq + isoq - 1;
        const double *B = tb.tips_qoft + (size_t)slot * 119;
        bq0 = B[tnq.J - 3];
        bq1 = B[tnq.J - 2];
        bq2 = B[tnq.J - 1];
        bq3 = B[min(tnq.J, 118)];
        q296q = tb.tips_q296[slot];
    }

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
    // (1 / (thi - tlo): the three quotients as constants - the compiler rounds them as the division would)
    const double RECTLC = (ILC == 1) ? 1.0 / (250.0 - 200.0) : ((ILC == 2) ? 1.0 / (296.0 - 250.0) : 1.0 / (340.0 - 296.0));
    const double TMPDIF = Tk - tlo;
    // LEAN (four wavenumbers per lane): the per-wavenumber state that the loops do not touch stays out of the registers - the
    // wavenumbers are re-read from sWn where they are needed, the radiation term is kept as four floats, and the launcher keeps
    // a.osum null (the finish kernel sums O_BY_MOL itself)
    constexpr bool LEAN = WPL >= 4;
    constexpr int NS = LEAN ? 1 : WPL;
    double RFTk[NS], osumk[NS];  // (registers unless LDS_STATE / LEAN)
    // LEAN (single precision only): the radiation term rounded to REAL*4 and the sum over the molecules accumulated in REAL*4 -
    // both as the sgl reference holds them (default REAL: modm.f90:264-269, :436-438)
    float RFTf[LEAN ? WPL : 1], osumf[LEAN ? WPL : 1];
    if constexpr (LEAN) {
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            RFTf[k] = (float)(WNk[k] * tanh_pos((RADCT * WNk[k]) / (2 * Tk)));
            osumf[k] = 0.f;
        }
    }
    if constexpr (!LEAN) {
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            RFTk[k] = WNk[k] * tanh_pos((RADCT * WNk[k]) / (2 * Tk));
            osumk[k] = 0.;
            if constexpr (LDS_STATE) {
                sRft[k * NT + tid] = RFTk[k];
                sOsum[k * NT + tid] = 0.;
            }
        }
    }
    const double lnRT = log(RT);
    const double cTk = RADCT / Tk, cT0 = RADCT / K_T0, dTinv = 1.0 / K_T0 - 1.0 / Tk;  // wave-uniform INTENS factors

    if (tid < nmol) sW[tid] = wkq;  // (the column read in stage 0; nmol <= 39 < NT)
#pragma unroll
    for (int k = 0; k < WPL; k++) sWn[k * NT + tid] = WNk[k];  // positions past nwn repeat the last wavenumber: still ascending
    if (tid == 0) {
        sLay[0] = RHORAT; sLay[1] = RP; sLay[2] = RP2; sLay[3] = lnRT; sLay[4] = cTk; sLay[5] = cT0; sLay[6] = dTinv;
        sLay[7] = RECTLC; sLay[8] = TMPDIF; sLay[9] = WTOT; sLay[17] = (double)ILC;
    }
    if (tid < MXBRD) sLay[10 + tid] = RHORAT * ((tid < nmol) ? wkq : (double)wk[tid]) / WTOT;  // rho_molec(1:7), modm.f90:313 (one division per lane, not seven in lane 0)
    if (tid < 2) sMomUsed[tid] = 0;
    if (FAR)
        for (int t = tid; t < NW * 2 * (FARP + 1); t += NT) (&sMom[0][0][0])[t] = 0.;
    // ---- candidate range of every active molecule for this wavenumber tile ------------------------
    // |Xnu - XNU0| <= max_abs_shift * RHORAT for every entry, with or without species broadening (line_table.cpp)
    const double pad = L.max_abs_shift * fmax(RHORAT, 1.0) + 1e-6;
    // (FAR tiles with the far field of far_kernel: the candidate runs come from far_plan_kernel, far lines left out)
    const bool planned = PLAN && a.farseg != nullptr;
    const double *gmom = planned ? a.farmom + ((pl * (size_t)a.far_ni + tile) * nmol) * FAR_MOM_STRIDE : nullptr;
    if (planned) {
        if (tid < nmol) {
            const int *sg = a.farseg + ((pl * (size_t)ntile + tile) * nmol + tid) * FAR_SEG_INTS;
#pragma unroll
            for (int k = 0; k < FAR_SEGS; k++) {
                sSegBase[tid * FAR_SEGS + k] = sg[k];
                sSegCum[tid * FAR_SEGS + k] = sg[FAR_SEGS + k];
            }
            sLo[tid] = 0;
            sOff[tid + 1] = sg[FAR_SEG_INTS - 1];  // (runs that are not used repeat the total)
        }
    } else if (tid < nmol) {
        const int m = tid, mol = m + 1;
        int lo = msq0, hi = msq1;
        if (wkq == 0.) hi = lo;  // W_SPECIES == 0 -> OL = 0 (modm.f90:318-321)
        // coupled O2 lines are exempt from the rule (modm.f90:755-792); an O2 list without any obeys it like the others
        // (a layer whose state holds a NaN - column, pressure or temperature - keeps every line: its shifted centres or widths
        // are NaN, the reference's test ABS(WN-Xnu).GT.25 is false for them and every line of the molecule adds its NaN)
        else if ((mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull) && WTOT == WTOT && RHORAT == RHORAT && Tk == Tk) {
            // 25 cm-1 rule (modm.f90:384): only lines with |WN - Xnu| <= 25 for some WN of the tile matter
            const double vlo = wnlo - 25.0 - pad, vhi = wnhi + 25.0 + pad;
            // a tile that spans the whole list (few scattered channels) keeps all of it: the two end lines read in stage 1
            // instead of two chains of dependent reads
            if (!(hi > lo && !(vq0 < vlo) && vq1 <= vhi)) {
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
        sOff[m + 1] = hi - lo;  // count, prefix-summed below
    }
    tile_sync<NW>();
    if (tid == 0) {
        int acc = 0;
        sOff[0] = 0;
        for (int m = 0; m < nmol; m++) { acc += sOff[m + 1]; sOff[m + 1] = acc; }
    }
    tile_sync<NW>();
    const int total = __builtin_amdgcn_readfirstlane(sOff[nmol]);  // (wave-uniform: chunk loop and run bounds in scalar registers)
    // TIPS + Doppler factor per (mol, iso): src/tips_2003.f90:60-296, src/modm.f90:442-454 - of the molecules that have
    // candidate lines only (the reference evaluates them per line; the others' entries are never read)
    for (int t = tid; t < nmol * 9; t += NT) {
        const int mol = t / 9 + 1, iso = t % 9 + 1;
        if (sOff[mol] == sOff[mol - 1]) continue;
        double sc = 0., dop = 0.;
        if (!t_bad) {
            bool bad = false;
            if (t == tid) {  // first pass: the values read in stages 0 / 1 (same arithmetic as tips_scor)
                if (iso > min(9, isnq)) sc = 0.;
                else if (mol == 34 || mol == 39) sc = 1.;
                else {
                    const double qt = tips_interp(Tk, tnq, bq0, bq1, bq2, tnq.ends ? 0. : bq3);
                    if (qt <= 0.) bad = true;
                    sc = q296q / qt;
                }
            } else sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
            if (bad) atomicOr(a.errflag, ERRBIT_TEMP);
        }
        const double M = (t == tid) ? Mq : tb.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = doppler_factor(M, Tk);
        sScor[t] = sc;
        sDop[t] = dop;
    }
    tile_sync<NW>();

    // this block's share of the candidate lines (the whole list when nslice == 1)
    int vbeg = 0, vend = total;
    if (nslice > 1) {
        vbeg = (int)(((long long)total * slice) / nslice);
        vend = (int)(((long long)total * (slice + 1)) / nslice);
    }
    // molecules without lines in this share / zero column: OL = 0 (modm.f90:314, :318-321)
    for (int m = 0; m < nmol; m++)
        if (min(sOff[m + 1], vend) <= max(sOff[m], vbeg)) {
            // a molecule of which the tile walks no line at all may still have a far field (far_kernel): slice 0 writes it
            bool far_only = false;
            if constexpr (PLAN && !LEAN) {
                if (planned && slice == 0 && sOff[m + 1] == sOff[m] && gmom[(size_t)m * FAR_MOM_STRIDE + FAR_P + 1] != 0.) {
                    far_only = true;
                    const double *gm = gmom + (size_t)m * FAR_MOM_STRIDE;
                    const double w0 = 0.5 * (sWn[0] + sWn[TW - 1]), rinv = (sWn[TW - 1] > sWn[0]) ? frcp_any(0.5 * (sWn[TW - 1] - sWn[0])) : 0. /* one wavenumber: x = 0, not 0 * inf */;
                    double poly[WPL], xk[WPL];
#pragma unroll
                    for (int k = 0; k < WPL; k++) xk[k] = (WNk[k] - w0) * rinv;
                    far_eval<FAR_P, 0, WPL>(nullptr, 0, xk, poly, gm);
#pragma unroll
                    for (int k = 0; k < WPL; k++)
                        if (validk[k]) {
                            const R sf = (R)(poly[k] - gm[FAR_P]);
                            const R od = (R)(SGL ? RFTk[k] * (double)sf : RFTk[k] * (sW[m] * (double)sf));
                            obm[(size_t)m * nwn + iwk[k]] = od;
                            osumk[k] += (double)od;
                        }
                }
            }
            if (!far_only) {
#pragma unroll
                for (int k = 0; k < WPL; k++)
                    if (validk[k]) obm[(size_t)m * nwn + iwk[k]] = (R)0;
            }
        }


    R SFk[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) SFk[k] = (R)0;
    // dense grids: the tile-independent part of every line of this (profile, layer), formed once by physics_kernel
    const PhysView phys = phys_view(a.phys, pl, (size_t)a.nprof * a.nlay_max, (size_t)a.phys_lines);

#ifdef MONORTM_ABLATE_LOOP
    if (a.nwn > 0) return;  // timing experiment: prologue only
#endif
#ifdef LINES_TIMING
    const long long tq1 = (long long)__builtin_readcyclecounter();
#endif
    int mchunk = 0;
    // one-wave tiles that form the line physics in place: the next chunk's table fields are read ahead (10 registers)
    // (measured on configs[3] with the assembly loops: 1.314 -> 1.402 ms - the ten registers push the prepare stage back into
    // scratch; kept as a switch for builds with more register room)
    constexpr bool PREFETCH = false;
    static_assert(!PREFETCH || !PLAN, "the read-ahead indexes sLo + (v - sOff): wrong for the candidate runs of far_plan_kernel (farseg)");
    LineFields nxt{};
    // (the quarter of its chunks a wave is in, without a division: (4 ck) / nchunks >= k  <=>  ck >= ceil(k nchunks / 4))
    const int nchunks = max(1, (vend - vbeg + NT - 1) / NT);
    const int fair_t1 = (nchunks + 3) >> 2, fair_t2 = (2 * nchunks + 3) >> 2, fair_t3 = (3 * nchunks + 3) >> 2;
    for (int base = vbeg, ck = 0; base < vend; base += NT, ck++) {
        if (a.fair) {
            // A grid of a few rounds of workgroups (api.hip decides): the SIMD arbitrates oldest-first among equal priorities, so
            // the four waves of a SIMD finish one after the other and the last round drains with three, two, one wave per SIMD -
            // a tail of a third of a workgroup's duration at low issue rates.  A wave that is further along yields to the waves
            // behind it (priority 3 .. 0 by the quarter of its chunks it is in): the waves of a SIMD end together and the slots
            // refill together (c4shard: 0.2135 -> 0.199 ms; c5, one round of two-wave workgroups: 0.137 -> 0.121 ms; nothing to
            // gain on many-round grids, where it costs a per cent)
            const int q = (ck >= fair_t1) + (ck >= fair_t2) + (ck >= fair_t3);  // = (4 * ck) / nchunks
            if (q <= 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#ifdef LINES_TIMING
        tqx = (long long)__builtin_readcyclecounter();
#endif
        // ================= prepare: one lane per line ================================================
        // (the lane id as an opaque value per chunk: the LDS addresses derived from it are formed here, one instruction each,
        // instead of being hoisted out of the chunk loop and kept - or spilled - across the evaluate stage)
        int ltid = tid;
        asm volatile("" : "+v"(ltid));
        unsigned long long kAL[1] = {0ull}, kM2[1] = {0ull}, kFar[1] = {0ull}, kFull[1] = {0ull}, kVg[1] = {0ull}, kYf[1] = {0ull};  // (NW == 1)
        double kTst[2] = {__builtin_inf(), -__builtin_inf()};
        // The same for the kernel arguments the prepare stage reads (table pointers, coupling scale factors): read from the
        // kernarg segment per chunk through an opaque copy of its address.  Hoisted out of the chunk loop they were live across
        // the evaluate stage - 60 SGPRs spilled to VGPR lanes, a 16-register block of table pointers restored with 16
        // v_readlane per chunk (the scalar cache serves a reload for one s_load)
        const __attribute__((address_space(4))) char *kseg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kseg));
        const ModmArgs &ac = *(const ModmArgs *)kseg;
        const DevLines &Lc = *(const DevLines *)(kseg + KARG_LINES);
        const int v = base + ltid;
        bool fAL = false, fM2 = false, fFar = false, fHalf = false, fV = false, fY = false;
        int mline = -1;
        Hot hA{};
        HotB hB{};
        ColdLine cC{};
        // molecule of the chunk's first line (wave-uniform, carried from chunk to chunk): a lane starts its search there - a
        // chunk crosses one or two molecule boundaries, the scan from molecule 1 took up to nmol trips for every lane
        while (mchunk + 1 < nmol && sOff[mchunk + 1] <= base) mchunk++;
        if (v < vend) {
            int m = mchunk;
            while (sOff[m + 1] <= v) m++;
            int idx = sLo[m] + (v - sOff[m]);
            if constexpr (PLAN) {
                if (planned) {
                    const int o = v - sOff[m];
                    int k = 0;
                    while (o >= sSegCum[m * FAR_SEGS + k]) k++;
                    idx = sSegBase[m * FAR_SEGS + k] + o;
                }
            }
            mline = m;
            prepare_line<R, IBRD>(ac, Lc, idx, m, sLay, sScor, sDop, sW, sWn, TW, phys, hA, hB, cC, fAL, fM2, fV, fY,
                                  (PREFETCH && ck > 0) ? &nxt : nullptr);
        }
        if constexpr (PREFETCH) {
            // the table fields of the NEXT chunk's line of this lane: eight independent loads that travel while this chunk is
            // evaluated (a one-wave workgroup has nobody else to hide them behind at the head of its next prepare stage)
            const int vn = v + NT;
            if (vn < vend) {
                int m = mchunk;
                while (sOff[m + 1] <= vn) m++;
                nxt = load_line_fields(Lc, sLo[m] + (vn - sOff[m]));
            }
        }
        if constexpr (FAR) {
            // Far field: untested one-resonance lines of uncoupled generic molecules / O2, at least FAR_KAPPA tile half-widths
            // from the tile centre, not Voigt candidates.  A wave serves one molecule (that of its first lane) and a chunk the
            // two molecule parities of its first line's molecule and the next; other lines are evaluated directly.
            const int mw = __builtin_amdgcn_readfirstlane(mline);
            const int mf = mchunk;  // (molecule of the chunk's first line)
            const double w0 = 0.5 * (sWn[0] + sWn[TW - 1]), rr = 0.5 * (sWn[TW - 1] - sWn[0]);
            if (mw >= 0 && mw - mf <= 1 && rr > 0. && !((L.lc_mask >> (mw + 1)) & 1ull)) {
                // the negative resonance goes along when every wavenumber of the tile includes it (WN + Xnu <= 25 at the
                // tile's upper end; uncoupled O2 has the same limit), provided it is far as well (|w0 + Xnu| >= FAR_KAPPA r)
                const double xnu = rec_xnu(hA), hw2 = hA.hw2, a2 = hA.a2, pa = hA.pa, pb = hB.pb;  // float fields widen here
                const bool m2all = fM2 && sWn[TW - 1] + xnu <= 25.;
                fFar = mline == mw && fAL && (!fM2 || m2all) && !(hB.d100 >= 0.) && !(fabs(xnu - w0) < FARK * rr) &&
                       (!m2all || !(fabs(xnu + w0) < FARK * rr));
                // HALF far (round 5): a NEAR line whose negative resonance every wavenumber of the tile includes.  Its pole at
                // -Xnu lies |w0 + Xnu| >= w0 away from the tile - far whenever the positive one is not - so that resonance (with its
                // pedestal) joins the sums and the line walks the one-resonance loop: 19 instead of 26 instructions per line and
                // wave for c3's near lines below 25 cm-1, 11 instead of 17 for the sounder channels of configs[4], where four fifths
                // of the window lines are such - but there ONE wave evaluates a line, and the series of its 64 lines cost that wave
                // more than the shorter loop saves (measured: configs[4] whole 1.028 -> 1.084 ms): tiles of several waves only.
                // Not for Voigt candidates (their correction recomputes both resonances) nor for lines with Y factors.
#ifndef MONORTM_NO_HALF
                fHalf = NW * WPL >= 8 && mline == mw && !fFar && fAL && m2all && !(hB.d100 >= 0.) && !fY && mw + 1 != 2 && !(fabs(xnu + w0) < FARK * rr);
#endif
                // the moments of a wave cost about as much as 16 lines evaluated directly by the four waves (a half-far line saves
                // half as much as a far one)
                if (2 * __popcll(__ballot(fFar)) + __popcll(__ballot(fHalf)) < 32) { fFar = false; fHalf = false; }
                if (__ballot(fFar || fHalf) != 0ull) {
                    if ((tid & 63) == 0) sMomUsed[mw & 1] = 1;
                    // pedestals: none for O2; CO2: -pa (2 - d^2/625) with d = t - delta is a quadratic in t (modm.f90:808-817)
                    const bool co2 = mw + 1 == 2;
                    const double dl = xnu - w0;
                    const double ped = (mw + 1 == 7 || co2) ? 0. : (fHalf ? pb : (m2all ? pa + pb : pa));
                    const double pq = (co2 && fFar) ? pa : 0.;
                    far_moments<FARP>(fFar, dl, (fFar && m2all) || fHalf, -(xnu + w0), hw2, a2, ped, co2, -pq * (2. - dl * dl * (1. / 625.)),
                                -pq * (2. * dl * (1. / 625.)), pq * (1. / 625.), rr, sMom[tid >> 6][mw & 1]);
                    if (fFar) {  // the record that is left adds nothing in any loop
                        hA.a2 = 0;
                        if (mw + 1 != 7) {
                            hA.pa = 0;
                            hB.pb = 0.;
                            if constexpr (SGL) hA.pb = 0;
                        }
                    }
                    if (fHalf) fM2 = false;  // one resonance left (the class masks keep such a line out of the two-resonance loops: cM below)
                }
            }
        }
        // single precision, two wavenumbers per pass: untested two-resonance lines whose negative resonance EVERY wavenumber of
        // the tile includes (WN + Xnu <= 25, +inf for coupled O2, at the tile's upper end) take the loop without per-lane factors
        bool fFull = false;
        if constexpr (SGL && WPL >= 2) {
            if (v < vend && mline + 1 != 2)
                fFull = fM2 && fAL && !fV && !fY && !fFar && sWn[TW - 1] + rec_xnu(hA) <= ((mline + 1 == 7) ? hB.pb : 25.);
        }
        if (v < vend) {
            sA[ltid] = hA;
            sB[ltid] = hB;
            if (fV) sCold[ltid] = cC;  // (read by voigt_flush alone, for Voigt candidates)
        }
        {
            const unsigned long long bA = __ballot(fAL), bM = __ballot(fM2), bF = __ballot(fFar), bV = __ballot(fV), bY = __ballot(fY);
            const unsigned long long bFu = __ballot(fFull);
            const unsigned long long bIn = __ballot(v < vend);   // (formed by the whole wave: the branch below is lane 0's alone)
#ifdef LINES_TIMING
            nFar += __popcll(bF); nAL += __popcll(bA & ~bF); nM2 += __popcll(bM & ~bF); nV += __popcll(bV);
#endif
            // short all-live islands take the tested loop of their neighbours, short one-resonance gaps the two-resonance
            // loop (0/1 factor per lane).  Not for two wavenumbers per lane in double: there the untested one-resonance
            // loop (shared reciprocal, lumped pedestal) is worth more than the switch (c3 +1.3 % with the smoothing)
            constexpr bool SMOOTH = WPL == 1 || SGL;
            // (a half-far line must not be drawn into a two-resonance loop by the smoothing: its negative resonance is in the sums)
            const unsigned long long cA = SMOOTH ? open_runs8(bA) : bA, cM = (SMOOTH ? close_runs8(bM) : bM) & ~__ballot(fHalf);
            const unsigned long long cFu = (SGL && WPL >= 2) ? open_runs8(bFu & cA) : 0ull;
            if constexpr (EDGE) {
                // least and largest centre among the lines that will walk the tested one-resonance loop (after the smoothing: an
                // all-live line drawn into it only widens the range, i.e. prevents a skip) - eval_dispatch leaves the loop out for
                // the wavenumbers of a wave that none of them can reach
                // (CO2 has no negative resonance: eval_dispatch ignores the two-resonance mask for it, and so must this - the
                // smoothing sets that bit for a short CO2 run between two-resonance neighbours)
                const bool tl = v < vend && !((cA >> (tid & 63)) & 1ull) && (mline + 1 == 2 || !((cM >> (tid & 63)) & 1ull)) && !fFar && !fY &&
                                !(SGL && fV);
                double lo = __builtin_inf(), hi = -__builtin_inf();
                if (__ballot(tl) != 0ull) {
                    const double xn = rec_xnu(hA);
                    lo = wave_min(tl ? xn : __builtin_inf());
                    hi = -wave_min(tl ? -xn : __builtin_inf());
                }
                if constexpr (NW == 1) { kTst[0] = lo; kTst[1] = hi; }
                else if ((tid & 63) == 0) { sTst[ck & 1][tid >> 6][0] = lo; sTst[ck & 1][tid >> 6][1] = hi; }
            }
            if constexpr (NW == 1) {
                // one-wave tile: the wave that ballots is the wave that walks - the masks stay in scalar registers (round 4: the
                // LDS round trip and ten v_readfirstlane per sub-run walk were ~4 % of a configs[3] wave's instructions)
                kAL[0] = cA; kM2[0] = cM; kFar[0] = bF; kFull[0] = cFu; kVg[0] = bV; kYf[0] = bY;
            } else if ((tid & 63) == 0) {
                sAL[ck & 1][tid >> 6] = cA;
                sM2[ck & 1][tid >> 6] = cM;
                sFar[ck & 1][tid >> 6] = bF;
                if constexpr (FAR) sAllFar[ck & 1][tid >> 6] = ((bF | ~bIn) == ~0ull) ? 1 : 0;
                if constexpr (SGL && WPL >= 2) sFull[ck & 1][tid >> 6] = cFu;
                sVg[ck & 1][tid >> 6] = bV;
                sYf[ck & 1][tid >> 6] = bY;
            }
        }
        tile_sync<NW>();

#ifdef LINES_TIMING
        { const long long t = (long long)__builtin_readcyclecounter(); tqP += t - tqx; tqx = t; }
#endif
        // ================= evaluate: every wave walks the prepared lines, molecule by molecule =========
#ifdef MONORTM_ABLATE_EVAL
        if (a.nwn > 0) { tile_sync<NW>(); continue; }  // timing experiment: prologue + prepare only
#endif
        // (from the molecule of the chunk's first line; the prefix sums as scalars, so that the run bounds and the branches on
        // them are wave-uniform code: the loop from molecule 1 with per-lane compares was ~1000 of a configs[3] wave's 16.6 k
        // instructions)
        // dense tiles: four fifths of the window lines are far, and so are all 256 lines of most chunks - then no wave has anything
        // to walk (the run bookkeeping below still happens: a run may start or end in the chunk)
        bool chunk_far = false;
        if constexpr (FAR && NW > 1) {
            int af = 1;
#pragma unroll
            for (int w = 0; w < NW; w++) af &= sAllFar[ck & 1][w];
#ifndef MONORTM_NO_CHUNKFAR
            chunk_far = __builtin_amdgcn_readfirstlane(af) != 0;
#endif
        }
        for (int m = __builtin_amdgcn_readfirstlane(mchunk); m < nmol; m++) {
            // the molecule's run restricted to this block's slice
            const int o0 = __builtin_amdgcn_readfirstlane(sOff[m]), o1 = __builtin_amdgcn_readfirstlane(sOff[m + 1]);
            const int s0 = max(o0, vbeg), s1 = min(o1, vend);
            if (s1 <= base || s0 >= s1) continue;
            if (s0 >= base + NT) break;
            const int j0 = max(s0, base) - base, j1 = min(s1, base + NT) - base;
            if (s0 >= base) {  // the molecule's run starts in this chunk
#pragma unroll
                for (int k = 0; k < WPL; k++) SFk[k] = (R)0;
            }
            const int mol = m + 1;
            const unsigned long long *mAL = (NW == 1) ? kAL : sAL[ck & 1], *mM2 = (NW == 1) ? kM2 : sM2[ck & 1];
            const unsigned long long *mFar = FAR ? ((NW == 1) ? kFar : sFar[ck & 1]) : nullptr;
            const unsigned long long *mV = (NW == 1) ? kVg : sVg[ck & 1], *mY = (NW == 1) ? kYf : sYf[ck & 1];
            const unsigned long long *mFu = (SGL && WPL >= 2) ? ((NW == 1) ? kFull : sFull[ck & 1]) : nullptr;
            const double wsc = SGL ? sW[m] : 1.0;
            // the class loops in assembly (lines_asm.hpp).  With species broadening too (round 4, with the 40-register version:
            // 64 B of scratch in that instantiation's prepare stage, c4brd 0.226 -> 0.213 ms; the first, 56-register version
            // lost there: 0.236 -> 0.269 ms)
            constexpr unsigned UB = (WPL == 1 && !SGL) ? (unsigned)sizeof(sRec.a) : 0u;
            double WNe[WPL];  // the lane's wavenumbers (LEAN: read from LDS where they are needed)
            if constexpr (!LEAN) {
#pragma unroll
                for (int k = 0; k < WPL; k++) WNe[k] = WNk[k];   // (the far field of a run that ends here reads them too)
            }
            if (chunk_far) {
                // (nothing to evaluate)
            } else if constexpr (!LEAN) {
                const double *tst = nullptr;
                double wlim[EDGE ? 4 : 1];
                if constexpr (EDGE) {   // (the lane's wavenumbers ascend with the lane: lanes 0 and 63 hold the wave's extremes)
                    tst = (NW == 1) ? kTst : &sTst[ck & 1][0][0];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        wlim[2 * k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(WNk[k])), __builtin_amdgcn_readfirstlane(__double2loint(WNk[k])));
                        wlim[2 * k + 1] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(WNk[k]), 63), __builtin_amdgcn_readlane(__double2loint(WNk[k]), 63));
                    }
                }
                if (mol == 7) eval_dispatch<1, R, Hot, WPL, false, UB>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WNe, mol, SFk, wsc, a.errflag, sVq[tid >> 6], 0, mFu, tst, wlim);
                else if (mol == 2) eval_dispatch<2, R, Hot, WPL, false, UB>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WNe, mol, SFk, wsc, a.errflag, sVq[tid >> 6], 0, mFu, tst, wlim);
                else eval_dispatch<0, R, Hot, WPL, false, UB>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WNe, mol, SFk, wsc, a.errflag, sVq[tid >> 6], 0, mFu, tst, wlim);
            } else {
                // four wavenumbers per lane = two passes of the two-wavenumber loops over the same prepared records: the
                // registers of the loops are those of the two-wavenumber tile (one copy of the code: the pass is a loop, the
                // sums of the other pair wait in two registers), the prologue and the prepare stage are paid once
#pragma unroll 1
                for (int hf = 0; hf < 2; hf++) {
                    const double WN2[2] = {sWn[(2 * hf) * NT + tid], sWn[(2 * hf + 1) * NT + tid]};
                    R SF2[2] = {hf ? SFk[2] : SFk[0], hf ? SFk[3] : SFk[1]};
                    unsigned short *vq = sVq[tid >> 6];
                    double wlim[4];   // (this pass's wavenumbers: positions 2 hf, 2 hf + 1 of every lane)
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        wlim[2 * k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(WN2[k])), __builtin_amdgcn_readfirstlane(__double2loint(WN2[k])));
                        wlim[2 * k + 1] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(WN2[k]), 63), __builtin_amdgcn_readlane(__double2loint(WN2[k]), 63));
                    }
                    const double *tst = (NW == 1) ? kTst : &sTst[ck & 1][0][0];
                    if (mol == 7) eval_dispatch<1, R, Hot, 2>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WN2, mol, SF2, wsc, a.errflag, vq, 0, mFu, tst, wlim);
                    else if (mol == 2) eval_dispatch<2, R, Hot, 2>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WN2, mol, SF2, wsc, a.errflag, vq, 0, mFu, tst, wlim);
                    else eval_dispatch<0, R, Hot, 2>(mAL, mM2, mFar, mV, mY, sA, sB, sCold, j0, j1, WN2, mol, SF2, wsc, a.errflag, vq, 0, mFu, tst, wlim);
                    if (hf) { SFk[2] = SF2[0]; SFk[3] = SF2[1]; }
                    else { SFk[0] = SF2[0]; SFk[1] = SF2[1]; }
                }
            }
            // run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438); in single precision W is already inside SF
            if (s1 <= base + NT) {
                // (the sums of far_kernel join in the slice that holds the molecule's last candidate)
                const double *gm = nullptr;
                if constexpr (PLAN) {
                    if (planned && s1 == o1 && gmom[(size_t)m * FAR_MOM_STRIDE + FAR_P + 1] != 0.) gm = gmom + (size_t)m * FAR_MOM_STRIDE;
                }
                if ((FAR && sMomUsed[m & 1] != 0) || gm != nullptr) {  // the far field of the run: one Chebyshev sum in x = (WN - w0) / r, the waves' sums added in wave order
                    const double w0 = 0.5 * (sWn[0] + sWn[TW - 1]), rinv = (sWn[TW - 1] > sWn[0]) ? frcp_any(0.5 * (sWn[TW - 1] - sWn[0])) : 0. /* one wavenumber: x = 0, not 0 * inf */;
                    double poly[WPL], xk[WPL];
                    if constexpr (LEAN) {  // (a fresh read: the copies of the evaluate stage are dead by now)
                        int lt = tid;
                        asm volatile("" : "+v"(lt));
#pragma unroll
                        for (int k = 0; k < WPL; k++) WNe[k] = sWn[k * NT + lt];
                    }
#pragma unroll
                    for (int k = 0; k < WPL; k++) xk[k] = (WNe[k] - w0) * rinv;
                    // (far_kernel's sums are FAR_P long whatever the tile; the tile's own have FARP entries - fewer on two-wave tiles)
                    if constexpr (!FAR) far_eval<FAR_P, 0, WPL>(nullptr, 0, xk, poly, gm);   // (PLANT: far_kernel's sums alone)
                    else if constexpr (FARP == FAR_P) far_eval<FARP, NW, WPL>(&sMom[0][m & 1][0], 2 * (FARP + 1), xk, poly, gm);
                    else {
                        far_eval<FARP, NW, WPL>(&sMom[0][m & 1][0], 2 * (FARP + 1), xk, poly);
                        if (gm) {
                            double pg[WPL];
                            far_eval<FAR_P, 0, WPL>(nullptr, 0, xk, pg, gm);
#pragma unroll
                            for (int k = 0; k < WPL; k++) poly[k] += pg[k];
                        }
                    }
                    double ped = gm ? gm[FAR_P] : 0.;
                    if constexpr (FAR) {
#pragma unroll
                        for (int w = 0; w < NW; w++) ped += sMom[w][m & 1][FARP];
                    }
#pragma unroll
                    for (int k = 0; k < WPL; k++) SFk[k] += (R)(poly[k] - ped);
                    if constexpr (FAR) {
                        tile_sync<NW>();  // every lane has read the moments: free the slot for the molecule after next
                        for (int t = tid; t < NW * (FARP + 1); t += NT) sMom[t / (FARP + 1)][m & 1][t % (FARP + 1)] = 0.;
                        if (tid == 0) sMomUsed[m & 1] = 0;
                        tile_sync<NW>();  // a later molecule of the same parity that ends in this chunk must see the cleared slot
                    }
                }
#pragma unroll
                for (int k = 0; k < WPL; k++)
                    if (validk[k]) {
                        double rft;
                        if constexpr (LEAN) rft = (double)RFTf[k];
                        else rft = LDS_STATE ? sRft[k * NT + tid] : RFTk[k];
                        const R od = (R)(SGL ? rft * (double)SFk[k] : rft * (sW[m] * (double)SFk[k]));
                        obm[(size_t)m * nwn + iwk[k]] = od;
                        // molecules complete in ascending order: the sum of modm.f90:264-269 (a lane's own slot: no race)
                        if constexpr (LDS_STATE) sOsum[k * NT + tid] += (double)od;
                        else if constexpr (LEAN) osumf[k] += (float)od;
                        else osumk[k] += (double)od;
                    }
            }
        }
        tile_sync<NW>();
#ifdef LINES_TIMING
        tqE += (long long)__builtin_readcyclecounter() - tqx;
#endif
    }
    if (a.osum) {
#pragma unroll
        for (int k = 0; k < WPL; k++)
            if (validk[k]) a.osum[pl * (size_t)nwn + iwk[k]] = LEAN ? (double)osumf[k] : (LDS_STATE ? sOsum[k * NT + tid] : osumk[k < NS ? k : 0]);
    }
#ifdef LINES_TIMING
    if (a.osum && tid == 0 && tile == ntile / 2 && slice == nslice / 2) {
        double *d = a.osum + pl * (size_t)nwn;
        d[0] = (double)(tq1 - tq0); d[1] = (double)tqP; d[2] = (double)tqE; d[3] = (double)((long long)__builtin_readcyclecounter() - tq0);
        d[4] = (double)total; d[5] = (double)(vend - vbeg); d[6] = nFar; d[7] = nAL; d[8] = nM2; d[9] = nV;
        d[10] = (double)trt0; d[11] = (double)(long long)__builtin_amdgcn_s_memrealtime();  // start / end of the workgroup, 10 ns units
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// physics_kernel: line_physics() of every line of the table for every (profile, layer), once per call - for dense grids, where
// lines_kernel would otherwise repeat it in each of the ~20 wavenumber tiles whose window holds the line (c3: 39 % of its
// instructions were the prepare stage).  32 B per (layer, line) + 16 B for a coupled one (LinePhysM / LinePhysY); lines_kernel reads the record instead of the nine table
// fields.  Same functions, same inputs as the in-place path: identical bits.
// grid = (blocks of per_block lines, profiles, layers); dynamic LDS = [nmol*9] Q(296)/Q(T), [nmol*9] Doppler factors, [nmol] W.
// ------------------------------------------------------------------------------------------------
template <typename R, bool IBRD>
__global__ __launch_bounds__(256) void physics_kernel(ModmArgs a, DevLines L, DevTables tb, int nlines, int per_block) {
    __shared__ double sLay[20];
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const int nmol = a.nmol;
    double *sScor = dyn_lds, *sDop = sScor + nmol * 9, *sW = sDop + nmol * 9;
    const int tid = threadIdx.x, prof = blockIdx.y, lay = blockIdx.z;
    if (lay >= a.nlay[prof]) return;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    const double Pk = rp<R>(a.P)[pl], Tk = rp<R>(a.T)[pl], wbrod = rp<R>(a.WBRODL)[pl];
    const R *wk = rp<R>(a.WKL) + pl * nmol;
    // ---- layer scalars: the expressions of lines_kernel, verbatim (INITI + head of LINES: modm.f90:868-883, :301-314)
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
    const double lnRT = log(RT);
    const double cTk = RADCT / Tk, cT0 = RADCT / K_T0, dTinv = 1.0 / K_T0 - 1.0 / Tk;
    for (int m = tid; m < nmol; m += 256) sW[m] = wk[m];
    if (tid == 0) {
        sLay[0] = RHORAT; sLay[1] = RP; sLay[2] = RP2; sLay[3] = lnRT; sLay[4] = cTk; sLay[5] = cT0; sLay[6] = dTinv;
        sLay[7] = RECTLC; sLay[8] = TMPDIF; sLay[9] = WTOT; sLay[17] = (double)ILC;
        for (int j = 0; j < MXBRD; j++) sLay[10 + j] = RHORAT * wk[j] / WTOT;  // rho_molec(1:7), modm.f90:313
    }
    // TIPS + Doppler factor per (mol, iso) of the molecules that have lines and a column (as in lines_kernel)
    for (int t = tid; t < nmol * 9; t += 256) {
        const int mol = t / 9 + 1, iso = t % 9 + 1;
        if (L.mol_start[mol + 1] == L.mol_start[mol] || wk[mol - 1] == 0.) continue;
        double sc = 0., dop = 0.;
        if (!(Tk < 70. || Tk > 3000.)) {  // (out of range / Q <= 0: flagged by lines_kernel, which every call launches)
            bool bad = false;
            sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
        }
        const double M = tb.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = doppler_factor(M, Tk);
        sScor[t] = sc;
        sDop[t] = dop;
    }
    __syncthreads();
    const size_t nstates = (size_t)a.nprof * a.nlay_max;
    LinePhysM *out = reinterpret_cast<LinePhysM *>(a.phys) + pl * (size_t)nlines;
    LinePhysY *outy = reinterpret_cast<LinePhysY *>(reinterpret_cast<LinePhysM *>(a.phys) + nstates * (size_t)nlines) + pl * (size_t)nlines;
    const int end = min(nlines, ((int)blockIdx.x + 1) * per_block);
    for (int idx = (int)blockIdx.x * per_block + tid; idx < end; idx += 256) {
        int m = 0;
        while (m + 1 < nmol && L.mol_start[m + 2] <= idx) m++;
        if (idx < L.mol_start[m + 1] || idx >= L.mol_start[m + 2] || sW[m] == 0.) continue;  // (lines of molecules beyond nmol)
        const uint32_t meta = L.meta[idx];
        const LinePhys ph = line_physics<IBRD>(a, L, idx, m, meta, sLay, sScor, sDop, sW);
        out[idx] = LinePhysM{ph.xnu, ph.hw, ph.hwd, ph.stild};
        if ((meta >> 10) & 3) outy[idx] = LinePhysY{ph.c1, ph.g};   // (coupled lines only: see LinePhysM)
    }
}

}  // namespace

namespace monortm_dev {
void launch_physics(const ModmArgs &a, const DevLines &L, const DevTables &tb, int nlines, bool ibrd, hipStream_t s) {
    const int per_block = 2048;
    const dim3 grid((nlines + per_block - 1) / per_block, a.nprof, a.nlay_max);
    const size_t dyn = sizeof(double) * (size_t)(19 * a.nmol);
    if (a.real_kind == 4) {
        if (ibrd) hipLaunchKernelGGL((physics_kernel<float, true>), grid, dim3(256), dyn, s, a, L, tb, nlines, per_block);
        else hipLaunchKernelGGL((physics_kernel<float, false>), grid, dim3(256), dyn, s, a, L, tb, nlines, per_block);
    } else {
        if (ibrd) hipLaunchKernelGGL((physics_kernel<double, true>), grid, dim3(256), dyn, s, a, L, tb, nlines, per_block);
        else hipLaunchKernelGGL((physics_kernel<double, false>), grid, dim3(256), dyn, s, a, L, tb, nlines, per_block);
    }
}
template <typename R, int NW, int WPL>
static void launch_lines_cfg(const ModmArgs &a, const DevLines &L, const DevTables &tb, bool ibrd, dim3 grid, size_t dyn_lds,
                             hipStream_t s) {
    if (ibrd) hipLaunchKernelGGL((lines_kernel<R, NW, WPL, true>), grid, dim3(NW * 64), dyn_lds, s, a, L, tb);
    else hipLaunchKernelGGL((lines_kernel<R, NW, WPL, false>), grid, dim3(NW * 64), dyn_lds, s, a, L, tb);
}
template <typename R>
static void launch_lines_t(const ModmArgs &a, const DevLines &L, const DevTables &tb, int nw, int wpl, bool ibrd, dim3 grid,
                           size_t dyn_lds, hipStream_t s) {
    if (nw == 1 && wpl == 1) launch_lines_cfg<R, 1, 1>(a, L, tb, ibrd, grid, dyn_lds, s);
    else if (nw == 1 && wpl == 4) {
        if constexpr (sizeof(R) == 4) launch_lines_cfg<R, 1, 4>(a, L, tb, ibrd, grid, dyn_lds, s);
    }
    else if (nw == 1 && a.farseg != nullptr) {   // (dense grid served by far_kernel)
        if (ibrd) hipLaunchKernelGGL((lines_kernel<R, 1, 2, true, true>), grid, dim3(64), dyn_lds, s, a, L, tb);
        else hipLaunchKernelGGL((lines_kernel<R, 1, 2, false, true>), grid, dim3(64), dyn_lds, s, a, L, tb);
    }
    else if (nw == 1) launch_lines_cfg<R, 1, 2>(a, L, tb, ibrd, grid, dyn_lds, s);
    else if (nw == 2) launch_lines_cfg<R, 2, 2>(a, L, tb, ibrd, grid, dyn_lds, s);
    else launch_lines_cfg<R, 4, 2>(a, L, tb, ibrd, grid, dyn_lds, s);
}
void lines_config(int nwn, int real_kind, long long states, double span, int *nw, int *wpl) {
    if (nwn <= 64) { *nw = 1; *wpl = 1; }
    // single precision, 129 - 256 wavenumbers, a batch that fills the chip twice over even so (states = profiles x layers): ONE
    // one-wave tile with four wavenumbers per lane - the prologue and the prepare stage (double precision arithmetic in this
    // build too) are paid once per layer instead of once per tile of 128 (configs[4] whole: lines 1.398 -> 1.242 ms; its 32-
    // profile share, 4096 states, is better off with the 8192 workgroups of two tiles: 0.183 against 0.215 ms)
    // ... for channel sets that span less than half the 25 cm-1 rule (a sounder's: configs[4], 0.3 - 6.5 cm-1): there the lines'
    // classes are the same for all four wavenumbers of a lane and most two-resonance lines are FULL.  A set spread over
    // 0.3 - 30 cm-1 is better off with two tiles of half the spread each (round 3's workload: 0.843 against 0.882 ms)
    else if (real_kind == 4 && nwn > 128 && nwn <= 256 && states >= 8192 && span <= 12.5) { *nw = 1; *wpl = 4; }
    // up to 256 wavenumbers: one or two one-wave tiles of 128.  Two tiles repeat the prepare stage, but a one-wave workgroup has
    // no barrier to wait at (configs[4] whole, 200 channels: 0.860 -> 0.819 ms against one two-wave tile of 256)
    else if (nwn <= 256) { *nw = 1; *wpl = 2; }
    else { *nw = 4; *wpl = 2; }
}
void launch_lines(const ModmArgs &a, const DevLines &L, const DevTables &tb, int nw, int wpl, bool ibrd, dim3 grid, size_t dyn_lds,
                  hipStream_t s) {
    if (a.real_kind == 4) launch_lines_t<float>(a, L, tb, nw, wpl, ibrd, grid, dyn_lds, s);
    else launch_lines_t<double>(a, L, tb, nw, wpl, ibrd, grid, dyn_lds, s);
}
}  // namespace monortm_dev
#ifdef LINES_CLASS_STATS
extern "C" void monortm_dbg_stats(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eval_stat), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_eval_stat), z, sizeof(z)); }
}
#endif
