// lines_state_kernel.hip - the line sum of MODM / LINES (reference src/modm.f90:253-262, :277-440) for BATCHES of atmospheric
// states on sparse channel sets (the microwave-radiometer use of monoRTM: BASELINE configs[3], configs[4]).
//
// lines_kernel (lane = wavenumber) leaves lanes idle on such inputs: 50 channels fill 50 of 64 lanes, and a line that the
// 25 cm-1 rule (modm.f90:384) cuts for part of the channels is still walked by every lane (c4shard: 79 % of the visits pass
// the test).  Here the roles are turned round:
//
//     lane  = one atmospheric STATE (profile, layer)          -> every lane has work, whatever the number of channels
//     wave  = <= 8 wavenumbers of the tile, held in SGPRs      -> the sums of a lane stay in 8 register pairs, no reduction
//     line  = wave-uniform                                     -> its table fields are scalar loads; whether a (line,
//                                                                wavenumber) pair is inside 25 cm-1 / has a negative
//                                                                resonance is decided ONCE per workgroup, as bit masks
//
// Workgroup = 64 states x 8 waves (one tile of <= 64 wavenumbers).  The chunks of CH = 8 lines of all molecules are one
// sequence; the prepared records exist twice in LDS: while every wave EVALUATES the 8 lines of chunk k for its own
// wavenumbers (the lane reading the record of ITS state), it PREPARES its line of chunk k + 1 for the 64 states (the same
// line_physics_core() as lines_kernel: shifted centre, S~, widths, coupling factors) into the other half; one barrier per
// chunk.  Two wavenumbers of a line share one reciprocal.  Per (state, wavenumber, molecule) the lines are added in table
// order = the reference's order (the Voigt terms of a line after its Lorentz terms).  state_tips_kernel forms the partition
// sums and Doppler factors per (state, molecule, isotopologue) ahead of the launch.
// Opt-in (MONORTM_LINES_KERNEL=state): fewer vector instructions than lines_kernel, but slower - DESIGN.md section 3.1b.
#include "lines_device.hpp"

namespace {
using namespace monortm_dev;

constexpr int SK_WAVES = 8;    // waves per workgroup
constexpr int SK_KW = 8;       // wavenumbers per wave (accumulator pairs per lane)
constexpr int SK_TILE = SK_WAVES * SK_KW;  // wavenumbers per workgroup
constexpr int SK_CH = 8;       // lines per chunk (one per wave)
constexpr int SK_VQ = 128;     // Voigt queue of a wave: worked off whenever it holds more than 64 triples (one step adds <= 64)
// record fields per (line, state) in LDS, [field][lane]
enum : int { F_XNU = 0, F_HW2, F_A2, F_PA, F_PB, F_D100, F_C1, F_GP1, F_N };
// line flags (wave-uniform)
enum : unsigned { LF_GENERAL = 1u, LF_VOIGT = 2u, LF_HASB = 4u };

__device__ __forceinline__ double uni_d(double x) {  // wave-uniform double -> SGPR pair
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned uni_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }

__device__ __forceinline__ double wave_max_d(double v) {
    v = fmax(v, dpp_move<0xB1, 0xf>(v));
    v = fmax(v, dpp_move<0x4E, 0xf>(v));
    v = fmax(v, dpp_move<0x141, 0xf>(v));
    v = fmax(v, dpp_move<0x140, 0xf>(v));
    v = fmax(v, dpp_move<0x142, 0xa>(v, 0.));
    v = fmax(v, dpp_move<0x143, 0xc>(v, 0.));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Eight doubles as a struct with named members and compile-time getters: an array here is turned into one <8 x double>
// value by the AMDGPU alloca-to-vector promotion, and every join of the evaluate loops then copies all sixteen registers.
struct D8 {
    double a, b, c, d, e, f, g, h;
};
template <int I>
__device__ __forceinline__ double &el(D8 &s) {
    if constexpr (I == 0) return s.a;
    else if constexpr (I == 1) return s.b;
    else if constexpr (I == 2) return s.c;
    else if constexpr (I == 3) return s.d;
    else if constexpr (I == 4) return s.e;
    else if constexpr (I == 5) return s.f;
    else if constexpr (I == 6) return s.g;
    else return s.h;
}
template <int I>
__device__ __forceinline__ double el(const D8 &s) {
    if constexpr (I == 0) return s.a;
    else if constexpr (I == 1) return s.b;
    else if constexpr (I == 2) return s.c;
    else if constexpr (I == 3) return s.d;
    else if constexpr (I == 4) return s.e;
    else if constexpr (I == 5) return s.f;
    else if constexpr (I == 6) return s.g;
    else return s.h;
}

// The additions into a lane's sums are written as tied-operand instructions: the eight sums are updated on many wave-uniform
// paths, and without the tie the compiler renames them at every join (copies, and a register pair per live copy).
__device__ __forceinline__ void acc_add(double &acc, double t) { asm("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(t)); }
__device__ __forceinline__ void acc_fma(double &acc, double a, double b) { asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }

// One Lorentz evaluation of an ordinary line (no Y factors, not a Voigt candidate) for one wavenumber, flags wave-uniform:
//   TEST: the 25 cm-1 test may fail for some state (|WN - xnu0| within the largest pressure shift of 25)
//   M2  : the negative resonance is included for some state; M2T: ... but possibly not for all (per-lane 0/1 factor)
// Same arithmetic as eval_one_fast of lines_device.hpp.  KIND: 0 generic, 1 O2 (no pedestal), 2 CO2.
template <int KIND>
__device__ __forceinline__ double sk_one(double xnu, double hw2, double a2, double pa, double pb, double WN, bool TEST, bool M2,
                                         bool M2T) {
    const double d = WN - xnu;
    const double den1 = fma(d, d, hw2);
    double term;
    if (KIND == 2) {
        const double f = fma(-(d * d), 1.0 / 625., 2.);
        term = fma(-pa, f, a2 * frcp(den1));
    } else if (!M2) {
        term = (KIND == 0) ? fma(a2, frcp(den1), -pa) : a2 * frcp(den1);
    } else {
        const double dp = WN + xnu;
        const double m2f = M2T ? ((dp <= 25.) ? 1.0 : 0.0) : 1.0;  // DIFF = (WN+Xnu) - 25 <= 0 (modm.f90:713)
        const double den2 = fma(dp, dp, hw2);
        const double num = fma(m2f, den1, den2);
        const double t = a2 * num;
        if (KIND == 0) term = fma(t, frcp(den1 * den2), -fma(m2f, pb, pa));
        else term = t * frcp(den1 * den2);
    }
    if (TEST) term = !(fabs(d) > 25.) ? term : 0.;  // modm.f90:384 (O2: inside the shape function, :755)
    return term;
}

// Two wavenumbers of one line, both inside 25 cm-1 for every state, one resonance: they share the reciprocal,
//   q = a2 / (den_a den_b);  a2/den_a = q den_b,  a2/den_b = q den_a
template <int KIND>
__device__ __forceinline__ void sk_pair(double xnu, double hw2, double a2, double pa, double WNa, double WNb, double &sa, double &sb) {
    const double da = WNa - xnu, db = WNb - xnu;
    const double dena = fma(da, da, hw2), denb = fma(db, db, hw2);
    const double q = a2 * frcp(dena * denb);
    if (KIND == 0) {
        acc_add(sa, fma(q, denb, -pa));
        acc_add(sb, fma(q, dena, -pa));
    } else if (KIND == 1) {
        acc_fma(sa, q, denb);
        acc_fma(sb, q, dena);
    } else {
        const double fa = fma(-(da * da), 1.0 / 625., 2.), fb = fma(-(db * db), 1.0 / 625., 2.);
        acc_add(sa, fma(-pa, fa, q * denb));
        acc_add(sb, fma(-pa, fb, q * dena));
    }
}

// A line whose shape carries line-coupling Y factors and / or may take a Voigt shape for some state: the per-lane logic of
// eval_general (lines_device.hpp).  cutlim / dplim: 25, or +inf for a coupled O2 line.  Returns the Lorentz term; *useV
// says that this (state, wavenumber) takes the Voigt shape instead (modm.f90:427) - the caller queues it.
template <int KIND>
__device__ __forceinline__ double sk_general(double xnu, double hw2, double a2, double pa, double pb, double d100, double c1,
                                             double gp1, double cutlim, double dplim, double WN, bool voigt, bool *useV) {
    const double d = WN - xnu, dp = WN + xnu;
    const double ad = fabs(d);
    const double den1 = fma(d, d, hw2);
    const double Y1 = fma(c1, d, gp1);
    double term;
    bool live;
    if (KIND == 2) {
        live = !(ad > 25.);
        const double f = fma(-(d * d), 1.0 / 625., 2.);
        term = Y1 * fma(-pa, f, a2 * frcp(den1));
    } else {
        live = !(ad > cutlim);
        const bool m2 = dp <= dplim;
        const double den2 = m2 ? fma(dp, dp, hw2) : 1.0;
        const double Y2 = m2 ? fma(-c1, dp, gp1) : 0.0;
        term = (a2 * fma(Y1, den2, Y2 * den1)) * frcp(den1 * den2);
        if (KIND == 0) term -= (m2 ? pa + pb : pa);
    }
    *useV = voigt && live && !(ad > d100);
    return live ? term : 0.;
}

// ------------------------------------------------------------------------------------------------
// LDS of the workgroup (file scope: the out-of-line stages below reach it without arguments).  The records, class bits and
// flags of a chunk exist twice: while the waves evaluate chunk k from one half, each of them prepares its line of chunk k + 1
// into the other, and ONE barrier per chunk separates the roles of the halves.
// ------------------------------------------------------------------------------------------------
// layer scalars, one copy for the eight waves (they all serve the same 64 states): the prepare stage loads them per line, the
// evaluate stage does not carry them
enum : int { LY_RHORAT = 0, LY_RP, LY_LNRT, LY_CTK, LY_DTINV, LY_RECTLC, LY_TMPDIF, LY_ILC, LY_TK, LY_WTOT, LY_N };
__shared__ double sRec[2][SK_CH][F_N][64];    // prepared records, [half][line of the chunk][field][state]
__shared__ double sLy[LY_N][64];              // layer scalars per state
__shared__ double sRho7[MXBRD][64];           // rho_molec(1:7) per state (referenced by the species-broadening instantiation only)
__shared__ double sWn[SK_TILE];               // the tile's wavenumbers (ascending; positions past the end repeat the last)
__shared__ unsigned sBits[2][SK_CH][SK_WAVES];  // per (line, wave): live | test << 8 | m2 << 16 | m2test << 24 for its wavenumbers
__shared__ unsigned sFlag[2][SK_CH];          // line flags (LF_*) | coupling code << 8
__shared__ int sB0[MXMOL], sB1[MXMOL];        // this slice's lines of a molecule: table indices [sB0, sB1); empty: no lines here or no column for any state
__shared__ float sSdep[2][SK_CH];
__shared__ unsigned short sVq[SK_WAVES][SK_VQ + 64];  // per wave: queued (line, wavenumber, state) triples that take a Voigt shape (+ 64 scratch slots)
struct SkConst {  // what the out-of-line stages need of the launch: they read it here, the main loop carries none of it
    PhysParams pp;
    double padS;
    int ntw, kbase, krem;
    // line table (the fields of DevLines the prepare stage reads)
    const double *vnu, *s0adj;
    const float *alfa, *hwhm, *epp, *tmpalf, *pshift, *sdep;
    const uint32_t *meta;
    // per call
    const void *WKL;
    const double *iso_grp;   // this group's rows of state_tips_kernel's output
    const double *rft;
    double *osum;            // null, or where the sum over the molecules goes (finish kernel of the microwave range)
    void *obm;               // O_BY_MOL, or this slice's partial sums
    long long st0;           // first state of the group
    int nmol, nwn, t0, nprof, nlay_max;
};
__shared__ SkConst sC;

// Work off the queued (line, wavenumber, state) triples that take a (speed-dependent) Voigt shape, one triple per lane, and
// hand each value to its state's lane in queue order (fixed: deterministic) - voigt_flush of lines_device.hpp for this
// layout.  The cold quantities are rebuilt from the record: HW = sqrt(HW^2) (exact), HWD = d100 / 100 and
// S~ = a2 pi / HW (each within an ulp or two of the prepare stage's value); the pedestal SDVOIGT(25, ...) is formed here.
// Out of line (one copy for the three molecule kinds): the shapes are thousands of instructions and rare.
__device__ __noinline__ double sk_voigt_values(const double (*sRec)[F_N][64], const unsigned short *vq, int nq, const float *sSdep,
                                               const unsigned *sFlag, const double *sWn, int k0, int mol, int *errflag) {
    const int lane = (int)__lane_id();
    double val = 0.;
    if (lane < nq) {
        const unsigned rec = (unsigned)vq[lane];
        const int qj = (int)(rec >> 9), qi = (int)(rec >> 6) & 7, owner = (int)(rec & 63u);
        const double qx = sRec[qj][F_XNU][owner], qhw2 = sRec[qj][F_HW2][owner], qa2 = sRec[qj][F_A2][owner];
        const double qd100 = sRec[qj][F_D100][owner], qc1 = sRec[qj][F_C1][owner], qgp1 = sRec[qj][F_GP1][owner];
        const double qhw = sqrt(qhw2), qhwd = qd100 * 0.01, qst = qa2 * (K_PI / qhw);
        const double qsd = (double)sSdep[qj];
        const int qcode = (int)(sFlag[qj] >> 8) & 3;
        const double xl3 = sdvoigt_far(25., qhw, qhwd, qsd, errflag);
        const double WNi = sWn[k0 + qi];
        // the shape functions only use the products AIP*(1/HW)*RP = c1 and BIP*RP2 = gp1-1: AIP' = c1*HW, BIP' = gp1-1, RP' = RP2' = 1
        const double SLS = lsf_sdvoigt(mol, qcode, 1.0, 1.0, qc1 * qhw, qgp1 - 1., qhw, WNi, qx, qhwd, qsd, xl3, errflag);
        val = qst * SLS;
    }
    return val;
}
// Evaluate the prepared lines [0, nch) of the chunk for this wave's wavenumbers.  KIND: 0 generic molecule, 1 O2, 2 CO2.
struct SkLine {  // the lane's record of one line + the wave-uniform class bits of its (line, wavenumber) pairs
    double xnu, hw2, a2, pa;
    unsigned live, test, m2, m2t;
};
// ordinary line whose wavenumbers of this wave are not all of one class (some cut by the 25 cm-1 rule, or cut for some states
// only, or with the negative resonance for some): wavenumber I with the per-lane tests; only "does any state have the
// negative resonance" stays a wave-uniform branch
template <int KIND, int I>
__device__ __forceinline__ void sk_mixed(const SkLine &l, const D8 &WN, D8 &acc) {
    if ((l.live >> I) & 1u)
        acc_add(el<I>(acc), sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, el<I>(WN), true, (KIND != 2) && ((l.m2 >> I) & 1u), true));
}

// A line whose shape carries line-coupling Y factors and / or may take a Voigt shape for some state, for the wave's
// wavenumbers: the per-lane logic of eval_general (lines_device.hpp).  Rare, and the (speed-dependent) Voigt function behind it
// needs ~100 registers: out of line, and the wave's sums stay in its scratch area `save` ([8][64] doubles, the caller parks them
// there) - nothing of the hot loops' register budget is spent here.  Lanes queue their (line, wavenumber, state) triples that
// take the Voigt shape; the queue is worked off one triple per lane (sk_voigt_values) and every value goes to its state's lane
// in queue order (fixed: deterministic), after the Lorentz terms of the line.
template <int KIND>
__device__ __noinline__ void sk_general_line(int buf, int jc, unsigned live, unsigned fl, int wv, int k0, int cnt, int mol, bool valid,
                                             int *errflag, volatile double *save) {
    const int lane = (int)__lane_id();
    buf = (int)uni_u((unsigned)buf); jc = (int)uni_u((unsigned)jc); live = uni_u(live); fl = uni_u(fl); wv = (int)uni_u((unsigned)wv);
    k0 = (int)uni_u((unsigned)k0); cnt = (int)uni_u((unsigned)cnt); mol = (int)uni_u((unsigned)mol);
    const double xnu = sRec[buf][jc][F_XNU][lane], hw2 = sRec[buf][jc][F_HW2][lane], a2 = sRec[buf][jc][F_A2][lane];
    const double pa = (KIND == 1) ? 0. : sRec[buf][jc][F_PA][lane];
    const double pb = sRec[buf][jc][F_PB][lane], d100 = sRec[buf][jc][F_D100][lane], c1 = sRec[buf][jc][F_C1][lane], gp1 = sRec[buf][jc][F_GP1][lane];
    const int code = (int)(fl >> 8) & 3;
    const bool voigt = (fl & LF_VOIGT) != 0u;
    const double lim = (KIND == 1 && code) ? __builtin_inf() : 25.;
    unsigned short *vq = sVq[wv];
    int nq = 0;
    auto flush = [&]() {
        for (int b0 = 0; b0 < nq; b0 += 64) {  // 64 triples at a time, one per lane
            const int nb = min(64, nq - b0);
            const double val = sk_voigt_values(sRec[buf], vq + b0, nb, sSdep[buf], sFlag[buf], sWn, k0, mol, errflag);
            const unsigned rec = (lane < nb) ? (unsigned)vq[b0 + lane] : 0u;
            for (int it = 0; it < nb; it++) {  // wave-uniform trip count and indices
                const int vlo = __builtin_amdgcn_readlane(__double2loint(val), it), vhi = __builtin_amdgcn_readlane(__double2hiint(val), it);
                const int rr = __builtin_amdgcn_readlane((int)rec, it);
                const int ow = rr & 63, wi = (rr >> 6) & 7;
                if (lane == ow) save[wi * 64 + lane] = save[wi * 64 + lane] + __hiloint2double(vhi, vlo);
            }
        }
        nq = 0;
    };
    for (int I = 0; I < cnt; I++) {
        if (!((live >> I) & 1u)) continue;
        const double WNi = uni_d(sWn[k0 + I]);
        bool useV = false;
        double term = sk_general<KIND>(xnu, hw2, a2, pa, pb, d100, c1, gp1, lim, lim, WNi, voigt, &useV);
        if (voigt) {
            useV = useV && valid;
            const unsigned long long mv = __ballot(useV);
            if (mv != 0ull) {
                // (no per-lane branch: lanes that queue nothing write to a scratch slot)
                vq[useV ? nq + __popcll(mv & ((1ull << lane) - 1ull)) : SK_VQ + lane] = (unsigned short)((jc << 9) | (I << 6) | lane);
                term = useV ? 0. : term;  // the Voigt value replaces the Lorentz term (modm.f90:427-432)
                nq += __popcll(mv);
            }
        }
        save[I * 64 + lane] = save[I * 64 + lane] + term;
        if (nq > SK_VQ - 64) flush();  // the next wavenumber may add 64 triples
    }
    if (nq > 0) flush();
}
template <int KIND>
__device__ __forceinline__ void sk_eval_chunk(int buf, int nch, int wv, int lane, int k0, int cnt, int mol, bool valid, const D8 &WN,
                                              D8 &acc, int *errflag, volatile double *save) {
    const unsigned full = (1u << cnt) - 1u;  // (cnt >= 1 whenever a line is live)
    // class bits and flags of the chunk's lines: one LDS read per chunk (lane j holds line j), handed out by v_readlane;
    // the lane's record of line j + 1 is fetched while line j is evaluated
    const unsigned vbits = sBits[buf][lane & (SK_CH - 1)][wv], vflag = sFlag[buf][lane & (SK_CH - 1)];
    double nx = sRec[buf][0][F_XNU][lane], nh = sRec[buf][0][F_HW2][lane], na = sRec[buf][0][F_A2][lane], np = (KIND == 1) ? 0. : sRec[buf][0][F_PA][lane];
    {
        // Lines come in table order = ascending wavenumber, so the classes of (line, this wave's wavenumbers) are the same over
        // long stretches: a chunk whose lines are ALL of one of the two common classes takes a loop without any per-line
        // decision (lane j < nch looks at line j's bits)
        const unsigned lv = vbits & 0xffu, ts = (vbits >> 8) & 0xffu, m2 = (vbits >> 16) & 0xffu, m2t = vbits >> 24;
        const bool gen = (vflag & LF_GENERAL) != 0u;
        const unsigned long long inm = (1ull << nch) - 1ull;
        const unsigned long long mD = __ballot(lv == 0u) & inm;
        if (mD == inm) return;  // nothing of this chunk reaches this wave's wavenumbers
        const unsigned long long mA = __ballot(!gen && lv == full && (ts | m2) == 0u) & inm;
        const unsigned long long mB = (KIND == 2) ? 0ull : (__ballot(!gen && lv == full && m2 == full && (ts | m2t) == 0u) & inm);
        if (mA == inm) {  // one resonance, every wavenumber of the wave inside 25 cm-1 for every state
            for (int jc = 0; jc < nch; jc++) {
                const double x = nx, h = nh, aa = na, p = np;
                const int jn = min(jc + 1, nch - 1);
                nx = sRec[buf][jn][F_XNU][lane]; nh = sRec[buf][jn][F_HW2][lane]; na = sRec[buf][jn][F_A2][lane];
                if (KIND != 1) np = sRec[buf][jn][F_PA][lane];
                sk_pair<KIND>(x, h, aa, p, WN.a, WN.b, acc.a, acc.b);
                sk_pair<KIND>(x, h, aa, p, WN.c, WN.d, acc.c, acc.d);
                sk_pair<KIND>(x, h, aa, p, WN.e, WN.f, acc.e, acc.f);
                if (cnt > 6) sk_pair<KIND>(x, h, aa, p, WN.g, WN.h, acc.g, acc.h);  // (wave-uniform; positions past the count repeat the last wavenumber)
            }
            return;
        }
        if (mB == inm) {  // ... and both resonances
            for (int jc = 0; jc < nch; jc++) {
                const double x = nx, h = nh, aa = na, p = np;
                const int jn = min(jc + 1, nch - 1);
                nx = sRec[buf][jn][F_XNU][lane]; nh = sRec[buf][jn][F_HW2][lane]; na = sRec[buf][jn][F_A2][lane];
                if (KIND != 1) np = sRec[buf][jn][F_PA][lane];
                acc_add(acc.a, sk_one<KIND>(x, h, aa, p, p, WN.a, false, true, false));
                acc_add(acc.b, sk_one<KIND>(x, h, aa, p, p, WN.b, false, true, false));
                acc_add(acc.c, sk_one<KIND>(x, h, aa, p, p, WN.c, false, true, false));
                acc_add(acc.d, sk_one<KIND>(x, h, aa, p, p, WN.d, false, true, false));
                acc_add(acc.e, sk_one<KIND>(x, h, aa, p, p, WN.e, false, true, false));
                acc_add(acc.f, sk_one<KIND>(x, h, aa, p, p, WN.f, false, true, false));
                if (cnt > 6) {
                    acc_add(acc.g, sk_one<KIND>(x, h, aa, p, p, WN.g, false, true, false));
                    acc_add(acc.h, sk_one<KIND>(x, h, aa, p, p, WN.h, false, true, false));
                }
            }
            return;
        }
    }
    for (int jc = 0; jc < nch; jc++) {
        const unsigned u = (unsigned)__builtin_amdgcn_readlane((int)vbits, jc);
        const unsigned fl = (unsigned)__builtin_amdgcn_readlane((int)vflag, jc);
        SkLine l;
        l.live = u & 0xffu;
        l.test = (u >> 8) & 0xffu;
        l.m2 = (u >> 16) & 0xffu;
        l.m2t = u >> 24;
        l.xnu = nx; l.hw2 = nh; l.a2 = na; l.pa = np;
        const int jn = min(jc + 1, nch - 1);
        nx = sRec[buf][jn][F_XNU][lane]; nh = sRec[buf][jn][F_HW2][lane]; na = sRec[buf][jn][F_A2][lane];
        if (KIND != 1) np = sRec[buf][jn][F_PA][lane];
        if (l.live != 0u) {
            if (fl & LF_GENERAL) {
                // Y factors and / or Voigt candidates (rare): out of line, with the sums handed over in the wave's scratch area
                save[0 * 64 + lane] = acc.a; save[1 * 64 + lane] = acc.b; save[2 * 64 + lane] = acc.c; save[3 * 64 + lane] = acc.d;
                save[4 * 64 + lane] = acc.e; save[5 * 64 + lane] = acc.f; save[6 * 64 + lane] = acc.g; save[7 * 64 + lane] = acc.h;
                sk_general_line<KIND>(buf, jc, l.live, fl, wv, k0, cnt, mol, valid, errflag, save);
                acc.a = save[0 * 64 + lane]; acc.b = save[1 * 64 + lane]; acc.c = save[2 * 64 + lane]; acc.d = save[3 * 64 + lane];
                acc.e = save[4 * 64 + lane]; acc.f = save[5 * 64 + lane]; acc.g = save[6 * 64 + lane]; acc.h = save[7 * 64 + lane];
            } else if (l.live == full && (l.test | l.m2) == 0u) {
                // the common case in ONE basic block (the scheduler interleaves the four reciprocal chains): every wavenumber
                // of the wave inside 25 cm-1 for every state, one resonance.  Positions past the wave's count repeat its last
                // wavenumber; their sums are never stored
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.a, WN.b, acc.a, acc.b);
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.c, WN.d, acc.c, acc.d);
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.e, WN.f, acc.e, acc.f);
                if (cnt > 6) sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.g, WN.h, acc.g, acc.h);   // (wave-uniform)
            } else if (KIND != 2 && l.live == full && l.m2 == full && (l.test | l.m2t) == 0u) {
                // ... and both resonances for every wavenumber and state
                acc_add(acc.a, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.a, false, true, false));
                acc_add(acc.b, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.b, false, true, false));
                acc_add(acc.c, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.c, false, true, false));
                acc_add(acc.d, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.d, false, true, false));
                acc_add(acc.e, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.e, false, true, false));
                acc_add(acc.f, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.f, false, true, false));
                if (cnt > 6) {
                    acc_add(acc.g, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.g, false, true, false));
                    acc_add(acc.h, sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.h, false, true, false));
                }
            } else {
                // ordinary line (no Y factors: c1 = g = 0, so both pedestals equal pa; no Voigt candidate), mixed classes
                sk_mixed<KIND, 0>(l, WN, acc); sk_mixed<KIND, 1>(l, WN, acc); sk_mixed<KIND, 2>(l, WN, acc); sk_mixed<KIND, 3>(l, WN, acc);
                sk_mixed<KIND, 4>(l, WN, acc); sk_mixed<KIND, 5>(l, WN, acc); sk_mixed<KIND, 6>(l, WN, acc); sk_mixed<KIND, 7>(l, WN, acc);
            }
        }
    }
}

__device__ __forceinline__ double bcast_d(double v, int src) {  // value of lane src (wave-uniform index) -> SGPR pair
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }


// ---- prepare one line of a chunk for the 64 states: records, class bits, flags -> LDS half `buf`.  Out of line with its own
// register allocation; it fetches what it needs itself (the line's table fields: one field per lane in two loads, made scalar
// by v_readlane; the column amount of the lane's state), so that the main loop carries nothing for it.
template <typename R, bool IBRD>
__device__ __noinline__ void sk_prepare_line(int idx, int mol, int jc, int buf, bool valid) {
    const int lane = (int)__lane_id();
    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    idx = (int)uni_u((unsigned)idx); mol = (int)uni_u((unsigned)mol); jc = (int)uni_u((unsigned)jc); buf = (int)uni_u((unsigned)buf);
    double fd;
    uint32_t fw;
    {
        const double *pd = (lane == 1) ? sC.s0adj : sC.vnu;
        fd = pd[idx];
        const void *pw = sC.alfa;
        pw = (lane == 1) ? (const void *)sC.hwhm : pw;
        pw = (lane == 2) ? (const void *)sC.epp : pw;
        pw = (lane == 3) ? (const void *)sC.tmpalf : pw;
        pw = (lane == 4) ? (const void *)sC.pshift : pw;
        pw = (lane == 5) ? (const void *)sC.sdep : pw;
        pw = (lane == 6) ? (const void *)sC.meta : pw;
        fw = static_cast<const uint32_t *>(pw)[idx];
    }
    const long long st = sC.st0 + lane;
    const double Wm = valid ? (double)static_cast<const R *>(sC.WKL)[(size_t)st * sC.nmol + (mol - 1)] : 0.;
    const double *iso_g = sC.iso_grp + (size_t)(mol - 1) * (9 * 128);
    LineFields lf;
    lf.xnu0 = bcast_d(fd, 0); lf.s0adj = bcast_d(fd, 1);
    lf.alfa = bcast_f(__uint_as_float(fw), 0); lf.hwhm = bcast_f(__uint_as_float(fw), 1);
    lf.epp = bcast_f(__uint_as_float(fw), 2); lf.tmpalf = bcast_f(__uint_as_float(fw), 3);
    lf.pshift = bcast_f(__uint_as_float(fw), 4);
    const float sdep_j = bcast_f(__uint_as_float(fw), 5);
    lf.meta = (uint32_t)__builtin_amdgcn_readlane((int)fw, 6);
    const uint32_t meta = lf.meta;
    const int iso = (meta >> 6) & 15, code = (meta >> 10) & 3;
    // Q(296)/Q(T) and HWHM_D / Xnu of the isotopologue per state: state_tips_kernel's rows [iso][2][64] of this molecule
    const bool isok = iso >= 1 && iso <= 9;
    const double *ig = iso_g + (size_t)((isok ? iso : 1) - 1) * 128;
    const double XIPSF = isok ? ig[lane] : 0.;
    const double dopfac = ig[64 + lane];
    LayerScalars ly;
    ly.RHORAT = sLy[LY_RHORAT][lane];
    ly.RP = sLy[LY_RP][lane];
    ly.RP2 = ly.RP * ly.RP;
    ly.lnRT = sLy[LY_LNRT][lane];
    ly.cTk = sLy[LY_CTK][lane];
    ly.cT0 = RADCT / K_T0;
    ly.dTinv = sLy[LY_DTINV][lane];
    ly.RECTLC = code ? sLy[LY_RECTLC][lane] : 0.;
    ly.TMPDIF = code ? sLy[LY_TMPDIF][lane] : 0.;
    ly.ILC = code ? (int)sLy[LY_ILC][lane] : 1;
    const double rho_self = ly.RHORAT * Wm / sLy[LY_WTOT][lane];
    double rho7[MXBRD];
#pragma unroll
    for (int j = 0; j < MXBRD; j++) rho7[j] = IBRD ? sRho7[j][lane] : 0.;
    const LinePhys ph = line_physics_core<IBRD>(sC.pp, idx, mol, lf, ly, rho_self, rho7, XIPSF, dopfac);
    {
#pragma clang fp contract(off)
        // records (line_records of lines_device.hpp, without the tile classes)
        const bool o2 = mol == 7, co2 = mol == 2;
        const double padS = sC.padS;
        const int ntw = sC.ntw, kbase = sC.kbase, krem = sC.krem;
        const double Xnu = ph.xnu, HW = ph.hw, HWD = ph.hwd, c1 = ph.c1, g = ph.g;
        const bool yfac = code != 0 && ((mol != 7 && mol != 2) || (mol == 7 && code == 1) || (mol == 2 && code != 2));
        const double zsum = HW + HWD, zthr = 0.99 * zsum;
        const bool zeta_gt = (HW > zthr * (1. + 1e-12)) ? true : ((HW < zthr * (1. - 1e-12)) ? false : (HW / zsum > 0.99));
        const double A2 = ph.stild * HW * (1.0 / K_PI);
        const double HW2 = HW * HW;
        const double p = A2 * frcp_any(625. + HW2);
        const double pa = o2 ? 0. : (co2 ? p : p * ((1. + c1 * 25.) + g));
        // Voigt is only possible when zeta <= 0.99 AND some wavenumber of the tile lies within 100 Doppler widths of the centre
        // (modm.f90:427).  The centres of the 64 states differ by at most padS from XNU0, so the nearest wavenumber of any
        // state is one of those around [XNU0 - padS, XNU0 + padS]
        double d100 = -1.0;
        const double xnu0 = lf.xnu0;
        const double wl = sWn[lane];
        const bool in_t = lane < ntw;
        if (__ballot(valid && !zeta_gt) != 0ull) {
            const int i0 = __popcll(__ballot(in_t && wl < xnu0 - padS));
            const int i1 = __popcll(__ballot(in_t && wl <= xnu0 + padS));
            double best = __builtin_inf();
            for (int i = max(i0 - 1, 0); i <= min(i1, ntw - 1); i++) best = fmin(best, fabs(sWn[i] - Xnu));
            if (!zeta_gt && !(best > 100. * HWD)) d100 = 100. * HWD;
        }
        const bool anyV = __ballot(valid && d100 >= 0.) != 0ull;
        sRec[buf][jc][F_XNU][lane] = Xnu;
        sRec[buf][jc][F_HW2][lane] = HW2;
        sRec[buf][jc][F_A2][lane] = A2;
        sRec[buf][jc][F_PA][lane] = pa;
        if (yfac || anyV) {  // read by the general path only
            sRec[buf][jc][F_PB][lane] = o2 ? 0. : (p * ((1. - c1 * 25.) + g));
            sRec[buf][jc][F_D100][lane] = d100;
            sRec[buf][jc][F_C1][lane] = c1;
            sRec[buf][jc][F_GP1][lane] = 1. + g;
        }
        // ---- classes of the (line, wavenumber) pairs, lane = position in the tile ----
        const bool exempt = o2 && code != 0;  // coupled O2: both resonances everywhere, no cut (modm.f90:755-792)
        const double dk = fabs(wl - xnu0), sp = wl + xnu0;
        const unsigned long long mLive = exempt ? ~0ull : __ballot(in_t && !(dk > 25. + padS));
        const unsigned long long mSure = exempt ? ~0ull : __ballot(in_t && !(dk > 25. - padS));
        const unsigned long long mM2 = co2 ? 0ull : (exempt ? ~0ull : __ballot(in_t && sp <= 25. + padS));
        const unsigned long long mM2s = co2 ? 0ull : (exempt ? ~0ull : __ballot(in_t && sp <= 25. - padS));
        if (lane < SK_WAVES) {
            // lane w packs the bits of wave w's wavenumbers [k0w, k0w + cntw)
            const int k0w = lane * kbase + min(lane, krem), cntw = kbase + (lane < krem ? 1 : 0);
            const unsigned msk = (1u << cntw) - 1u;
            const unsigned lv = (unsigned)(mLive >> k0w) & msk, su = (unsigned)(mSure >> k0w) & msk;
            const unsigned m2 = (unsigned)(mM2 >> k0w) & msk & lv, m2s = (unsigned)(mM2s >> k0w) & msk;
            sBits[buf][jc][lane] = lv | ((lv & ~su) << 8) | (m2 << 16) | ((m2 & ~m2s) << 24);
        }
        if (lane == 0) {
            sFlag[buf][jc] = ((yfac || anyV) ? LF_GENERAL : 0u) | (anyV ? LF_VOIGT : 0u) | ((unsigned)code << 8);
            sSdep[buf][jc] = sdep_j;
        }
    }
}

// ---- a molecule's run is complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438) for this wave's wavenumbers, the sums taken
// from the wave's scratch area.  Out of line: once per molecule, and its addresses stay out of the main loop.
template <typename R>
__device__ __noinline__ void sk_store_molecule(int m, int wv, bool valid, volatile double *save) {
    const int lane = (int)__lane_id();
    m = (int)uni_u((unsigned)m); wv = (int)uni_u((unsigned)wv);
    const int k0 = wv * sC.kbase + min(wv, sC.krem), cnt = sC.kbase + (wv < sC.krem ? 1 : 0);
    const long long st = sC.st0 + lane;
    if (st >= (long long)sC.nprof * sC.nlay_max) return;  // (layers beyond nlay[p] receive zeros, as from lines_kernel)
    const size_t pl = (size_t)st;
    const double Wm = valid ? (double)static_cast<const R *>(sC.WKL)[pl * sC.nmol + m] : 0.;
    R *obm = static_cast<R *>(sC.obm) + pl * sC.nmol * (size_t)sC.nwn + (size_t)m * sC.nwn;
    for (int I = 0; I < cnt; I++) {
        const size_t iw = (size_t)(sC.t0 + k0 + I);
        obm[iw] = (Wm == 0. || !valid) ? (R)0 : (R)(sC.rft[pl * (size_t)sC.nwn + iw] * (Wm * save[I * 64 + lane]));
    }
}

// ---- molecules without lines in this slice (or without column for any state of the group): zeros for this wave's wavenumbers
template <typename R>
__device__ __noinline__ void sk_zero_molecules(int wv) {
    const int lane = (int)__lane_id();
    wv = (int)uni_u((unsigned)wv);
    const int k0 = wv * sC.kbase + min(wv, sC.krem), cnt = sC.kbase + (wv < sC.krem ? 1 : 0);
    const long long st = sC.st0 + lane;
    if (st >= (long long)sC.nprof * sC.nlay_max) return;
    R *obm = static_cast<R *>(sC.obm) + (size_t)st * sC.nmol * (size_t)sC.nwn;
    for (int m = 0; m < sC.nmol; m++) {
        if ((int)uni_u((unsigned)sB1[m]) > (int)uni_u((unsigned)sB0[m])) continue;
        for (int I = 0; I < cnt; I++) obm[(size_t)m * sC.nwn + (size_t)(sC.t0 + k0 + I)] = (R)0;
    }
}
// ---- sum over the molecules of O_BY_MOL as stored, in molecule order (modm.f90:264-269), for the finish kernel: read back from
// what this thread has written
template <typename R>
__device__ __noinline__ void sk_molecule_sum(int wv, bool valid) {
    const int lane = (int)__lane_id();
    wv = (int)uni_u((unsigned)wv);
    if (!sC.osum || !valid) return;
    const int k0 = wv * sC.kbase + min(wv, sC.krem), cnt = sC.kbase + (wv < sC.krem ? 1 : 0);
    const size_t pl = (size_t)(sC.st0 + lane);
    const R *obm = static_cast<const R *>(sC.obm) + pl * sC.nmol * (size_t)sC.nwn;
    for (int I = 0; I < cnt; I++) {
        const size_t iw = (size_t)(sC.t0 + k0 + I);
        double sm = 0.;
        for (int m = 0; m < sC.nmol; m++) sm += (double)obm[(size_t)m * sC.nwn + iw];
        sC.osum[pl * (size_t)sC.nwn + iw] = sm;
    }
}

// ================= the chunks of all molecules as one sequence ======================================
// chunk = (molecule index, first table line); the following one: the next SK_CH lines of the molecule or the first of the next
// molecule that has any.  While the waves evaluate chunk k from one half of the records, each of them prepares its line of
// chunk k + 1 into the other half; one barrier per chunk.  A function of its own, so that nothing of the kernel's prologue
// competes with the eight sums, the two records and the reciprocal chains for registers.
template <typename R, bool IBRD>
__device__ __noinline__ void sk_main_loop(D8 WNv, int wv, int k0, int cnt, bool valid, int *errflag, volatile double *vsave) {
    const int lane = (int)__lane_id();
    wv = (int)uni_u((unsigned)wv); k0 = (int)uni_u((unsigned)k0); cnt = (int)uni_u((unsigned)cnt);
    D8 WN;  // wave-uniform: scalar registers
    WN.a = uni_d(WNv.a); WN.b = uni_d(WNv.b); WN.c = uni_d(WNv.c); WN.d = uni_d(WNv.d);
    WN.e = uni_d(WNv.e); WN.f = uni_d(WNv.f); WN.g = uni_d(WNv.g); WN.h = uni_d(WNv.h);
    const int nmol = (int)uni_u((unsigned)sC.nmol);
#ifdef SK_TIMING
    long long tq_prep = 0, tq_eval = 0, tq_bar = 0, tq_flush = 0, tq_x = (long long)__builtin_readcyclecounter();
    const long long tq_0 = tq_x;
#define SK_T(acc_) do { const long long t_ = (long long)__builtin_readcyclecounter(); acc_ += t_ - tq_x; tq_x = t_; } while (0)
#else
#define SK_T(acc_)
#endif
    auto seek = [&](int &m, int &b) {
        while (m < nmol && b >= (int)uni_u((unsigned)sB1[m])) {
            m++;
            if (m < nmol) b = (int)uni_u((unsigned)sB0[m]);
        }
    };
    int cm = 0, cb = (int)uni_u((unsigned)sB0[0]);
    seek(cm, cb);
    int nm = cm, nb = cb + SK_CH;
    if (cm < nmol) {
        seek(nm, nb);
        if (cb + wv < (int)uni_u((unsigned)sB1[cm])) sk_prepare_line<R, IBRD>(cb + wv, cm + 1, wv, 0, valid);
    }
    SK_T(tq_prep);
    __syncthreads();
    SK_T(tq_bar);
    int buf = 0;
    D8 acc = {0., 0., 0., 0., 0., 0., 0., 0.};
    while (cm < nmol) {
        const int mol = cm + 1;
        const int nch = min(SK_CH, (int)uni_u((unsigned)sB1[cm]) - cb);
        auto do_prepare = [&]() {
            // ---- prepare: this wave's line of the NEXT chunk, for the 64 states, into the other half ----
#ifdef SK_ABL_PREP
            if (buf == 0 && cb == (int)uni_u((unsigned)sB0[cm]) && nm < nmol && nb + wv < (int)uni_u((unsigned)sB1[nm]) && nm == cm)
#else
            if (nm < nmol && nb + wv < (int)uni_u((unsigned)sB1[nm]))
#endif
                sk_prepare_line<R, IBRD>(nb + wv, nm + 1, wv, buf ^ 1, valid);
            SK_T(tq_prep);
        };
        auto do_evaluate = [&]() {
            // ---- evaluate: every line of the CURRENT chunk for this wave's wavenumbers ----
#ifdef SK_ABL_EVAL
            if (cnt < 0)
#endif
            if (mol == 7) sk_eval_chunk<1>(buf, nch, wv, lane, k0, cnt, mol, valid, WN, acc, errflag, vsave);
            else if (mol == 2) sk_eval_chunk<2>(buf, nch, wv, lane, k0, cnt, mol, valid, WN, acc, errflag, vsave);
            else sk_eval_chunk<0>(buf, nch, wv, lane, k0, cnt, mol, valid, WN, acc, errflag, vsave);
            SK_T(tq_eval);
        };
        // (the two stages touch different halves of the records; taking them in the opposite order in half of the waves, so that
        // prepare and evaluate stages overlap on a SIMD, was measured: 1.69 -> 2.14 ms)
        do_prepare();
        do_evaluate();
        if (nm != cm) {  // the molecule's run is complete
            vsave[0 * 64 + lane] = acc.a; vsave[1 * 64 + lane] = acc.b; vsave[2 * 64 + lane] = acc.c; vsave[3 * 64 + lane] = acc.d;
            vsave[4 * 64 + lane] = acc.e; vsave[5 * 64 + lane] = acc.f; vsave[6 * 64 + lane] = acc.g; vsave[7 * 64 + lane] = acc.h;
            sk_store_molecule<R>(cm, wv, valid, vsave);
            acc = D8{0., 0., 0., 0., 0., 0., 0., 0.};
            SK_T(tq_flush);
        }
#ifndef SK_ABL_NOBAR
        __syncthreads();  // chunk k + 1 is complete in the other half; this half may be overwritten
#endif
        SK_T(tq_bar);
        cm = nm; cb = nb;
        nb += SK_CH;
        if (nm < nmol) seek(nm, nb);
        buf ^= 1;
    }
#ifdef SK_TIMING
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0)
        printf("SK_TIMING block %d wave %d cnt %d: prepare %lld evaluate %lld barrier %lld flush %lld loop %lld\n", (int)blockIdx.x, wv, cnt,
               tq_prep, tq_eval, tq_bar, tq_flush, (long long)__builtin_readcyclecounter() - tq_0);
#endif
}

// ------------------------------------------------------------------------------------------------
// Partition-sum ratios and Doppler factors ahead of the line sum: out[group][molecule][isotopologue][2][64 states]
// (src/tips_2003.f90:60-296, modm.f90:442-454), once per (state, molecule, isotopologue) whatever the number of line slices.
// grid = groups of 64 states, block = 4 waves sharing the (molecule, isotopologue) pairs.
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(256) void state_tips_kernel(ModmArgs a, DevTables tb) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, group = blockIdx.x;
    const long long st = (long long)group * 64 + lane;
    const int prof = (int)(st / a.nlay_max), lay = (int)(st % a.nlay_max);
    const bool inb = prof < a.nprof;
    const bool valid = inb && lay < a.nlay[inb ? prof : 0];
    const double Tk = valid ? (double)rp<R>(a.T)[(size_t)st] : K_T0;
    const bool ok = valid && !(Tk < 70. || Tk > 3000.);
    double *out = a.sk_iso + (size_t)group * a.nmol * (9 * 128);
    for (int q = wv; q < a.nmol * 9; q += 4) {
        const int m = q / 9, iso = q % 9 + 1, mol = m + 1;
        const int niso = min(9, tb.tips_isonm[mol - 1]);
        double sc = 0., dop = 0.;
        if (iso <= niso && ok) {
            bool bad = false;
            sc = tips_scor(tb.tips_isonm, tb.tips_offset, tb.tips_qoft, tb.tips_q296, mol, iso, Tk, &bad);
            if (bad) atomicOr(a.errflag, ERRBIT_TEMP);
        }
        const double M = tb.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = sqrt(2. * log(2.) * ((K_BOLTZ * Tk) / (M / K_AVOGAD))) / K_CLIGHT;
        out[(size_t)q * 128 + lane] = sc;
        out[(size_t)q * 128 + 64 + lane] = dop;
    }
}

// ------------------------------------------------------------------------------------------------
// grid = (groups of 64 states x line slices, wavenumber tiles); block = 8 waves; dynamic LDS = the molecule windows
// ------------------------------------------------------------------------------------------------
template <typename R, bool IBRD>
__global__ __launch_bounds__(SK_WAVES * 64, 4) void lines_state_kernel(ModmArgs a, DevLines L, DevTables tb, int tile_w) {
    extern __shared__ __attribute__((aligned(16))) int dyn_lds_i[];
    int *sLo = dyn_lds_i;            // [nmol]   first candidate line of the molecule
    int *sOff = sLo + a.nmol;        // [nmol+1] prefix sums of the candidate counts

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: tell the compiler (scalar loops and loads)
#ifdef SK_TIMING
    const long long tq_start = (long long)__builtin_readcyclecounter();
#endif
    const int nslice = a.nslice, nwn = a.nwn, nmol = a.nmol;
    const int group = blockIdx.x / nslice, slice = blockIdx.x % nslice, tile = blockIdx.y;
    const int t0 = tile * tile_w, ntw = min(tile_w, nwn - t0);  // this tile's wavenumbers [t0, t0 + ntw)
    // this wave's wavenumbers: positions [k0, k0 + cnt) of the tile, spread evenly over the waves
    const int kbase = ntw / SK_WAVES, krem = ntw % SK_WAVES;
    const int k0 = wv * kbase + min(wv, krem), cnt = kbase + (wv < krem ? 1 : 0);

    // ---- the lane's state -----------------------------------------------------------------------
    const long long st = (long long)group * 64 + lane;
    const int prof = (int)(st / a.nlay_max), lay = (int)(st % a.nlay_max);
    const bool inb = prof < a.nprof;
    const bool valid = inb && lay < a.nlay[inb ? prof : 0];
    const size_t pl = inb ? (size_t)st : 0;  // = prof * nlay_max + lay
    if (tile == 0 && slice == 0 && wv == 0) {  // arguments that live in device memory (see lines_kernel)
        if (inb && lay == 0 && (a.nlay[prof] < 1 || a.nlay[prof] > a.nlay_max)) atomicOr(a.errflag, ERRBIT_ARG);
        if (group == 0) {
            for (int i = lane; i + 1 < nwn; i += 64)
                if (a.wn[i + 1] < a.wn[i]) atomicOr(a.errflag, ERRBIT_ARG);  // modm.f90:180-181
        }
    }
    const R *wk = rp<R>(a.WKL) + pl * nmol;
    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    if (tid < SK_TILE) sWn[tid] = a.wn[t0 + min(tid, ntw - 1)];
    {
        const double Pk = valid ? (double)rp<R>(a.P)[pl] : K_P0, Tk = valid ? (double)rp<R>(a.T)[pl] : K_T0;
        // MODM calls TIPS_2003 for every layer and all nmol molecules (modm.f90:250): outside 70-3000 K the reference STOPs
        const bool t_bad = valid && (Tk < 70. || Tk > 3000.);
        if (t_bad && tile == 0 && slice == 0 && wv == 0) atomicOr(a.errflag, ERRBIT_TEMP);
        if (wv == 0) {
            // layer scalars (INITI + head of LINES: modm.f90:868-883, :301-314) - the expressions of lines_kernel
            const double wbrod = valid ? (double)rp<R>(a.WBRODL)[pl] : 1.;
            const double XN0 = (K_P0 / (K_BOLTZ * K_T0)) * 1.E+3;
            const double Xn = (Pk / (K_BOLTZ * Tk)) * 1.E+3;
            double WTOT = 0.;
            for (int m = 0; m < nmol; m++) WTOT += valid ? (double)wk[m] : 0.;
            WTOT = WTOT + wbrod;
            const double RHORAT = Xn / XN0;
            const int ILC = (Tk < 250.0) ? 1 : ((Tk < 296.0) ? 2 : 3);  // TEMPLC = 200,250,296,340
            const double tlo = (ILC == 1) ? 200.0 : (ILC == 2 ? 250.0 : 296.0);
            const double thi = (ILC == 1) ? 250.0 : (ILC == 2 ? 296.0 : 340.0);
            sLy[LY_RHORAT][lane] = RHORAT;
            sLy[LY_RP][lane] = Pk / K_P0;
            sLy[LY_LNRT][lane] = log(Tk / K_T0);
            sLy[LY_CTK][lane] = RADCT / Tk;
            sLy[LY_DTINV][lane] = 1.0 / K_T0 - 1.0 / Tk;
            sLy[LY_RECTLC][lane] = 1.0 / (thi - tlo);
            sLy[LY_TMPDIF][lane] = Tk - tlo;
            sLy[LY_ILC][lane] = (double)ILC;
            sLy[LY_TK][lane] = Tk;
            sLy[LY_WTOT][lane] = WTOT;
            if (IBRD) {
#pragma unroll
                for (int j = 0; j < MXBRD; j++) sRho7[j][lane] = valid ? RHORAT * (double)wk[j] / WTOT : 0.;  // rho_molec(1:7), modm.f90:313
            }
            const double mx = wave_max_d(valid ? RHORAT : 0.);
            if (lane == 0) {
                // |Xnu - XNU0| <= max_abs_shift * RHORAT for every entry, with or without species broadening (line_table.cpp);
                // the margin covers the roundings of the sums and differences involved
                sC.padS = L.max_abs_shift * fmax(1.0, mx) + 1e-9;
                sC.pp = phys_params(a, L);
                sC.ntw = ntw; sC.kbase = kbase; sC.krem = krem;
                sC.vnu = L.vnu; sC.s0adj = L.s0adj; sC.alfa = L.alfa; sC.hwhm = L.hwhm; sC.epp = L.epp; sC.tmpalf = L.tmpalf;
                sC.pshift = L.pshift; sC.sdep = L.sdep; sC.meta = L.meta;
                sC.WKL = a.WKL;
                sC.iso_grp = a.sk_iso + (size_t)group * nmol * (9 * 128);
                sC.rft = a.rft;
                sC.osum = a.osum;
                sC.obm = (nslice == 1) ? a.O_BY_MOL
                                       : (void *)(static_cast<char *>(a.partial) + (size_t)slice * a.nprof * a.nlay_max * nmol * (size_t)nwn * sizeof(R));
                sC.st0 = (long long)group * 64;
                sC.nmol = nmol; sC.nwn = nwn; sC.t0 = t0; sC.nprof = a.nprof; sC.nlay_max = a.nlay_max;
            }
        }
    }
    __syncthreads();
    const double padS = uni_d(sC.padS);
    // this wave's wavenumbers in scalar registers; RFT per (state, wavenumber) (modm.f90:436-438) goes to the scratch array
    // a.rft and is read back when a molecule's run is complete (same thread: program order)
    D8 WN;
    {
        const double Tk = sLy[LY_TK][lane];
#define SK_W(I)                                                                                          \
    el<I>(WN) = uni_d(sWn[min(k0 + min(I, max(cnt - 1, 0)), SK_TILE - 1)]);                               \
    if (I < cnt && inb) a.rft[pl * (size_t)nwn + (t0 + k0 + I)] = el<I>(WN) * tanh_pos((RADCT * el<I>(WN)) / (2 * Tk));
        SK_W(0) SK_W(1) SK_W(2) SK_W(3) SK_W(4) SK_W(5) SK_W(6) SK_W(7)
#undef SK_W
    }
    // ---- candidate range of every molecule for this tile (as lines_kernel) ----
    {
        const double wnlo = sWn[0], wnhi = sWn[ntw - 1];
        for (int m = tid; m < nmol; m += SK_WAVES * 64) {
            const int mol = m + 1;
            int lo = L.mol_start[mol], hi = L.mol_start[mol + 1];
            // coupled O2 lines are exempt from the rule (modm.f90:755-792); an O2 list without any obeys it like the others
            if ((mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull)) {
                const double vlo = wnlo - 25.0 - padS, vhi = wnhi + 25.0 + padS;
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
    {
        // this slice's share [vbeg, vend) of the concatenated candidate lists, per molecule as table indices.  W_SPECIES == 0 ->
        // OL = 0 without a walk (modm.f90:318-321): a molecule is skipped when that holds for every state of the group
        const int total = sOff[nmol];
        const int vbeg = (int)(((long long)total * slice) / nslice), vend = (int)(((long long)total * (slice + 1)) / nslice);
        for (int m = wv; m < nmol; m += SK_WAVES) {
            const double Wm = valid ? (double)wk[m] : 0.;
            const bool act = __ballot(Wm != 0.) != 0ull;
            if (lane == 0) {
                const int s0 = max(sOff[m], vbeg), s1 = min(sOff[m + 1], vend);
                const int lo_m = sLo[m] - sOff[m];
                sB0[m] = lo_m + s0;
                sB1[m] = (act && s1 > s0) ? lo_m + s1 : lo_m + s0;
            }
        }
    }
    __syncthreads();

    // this wave's 4 KB of the scratch array through which the sums are handed to the out-of-line stages
    volatile double *vsave = a.vsave + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SK_WAVES + wv) * (SK_KW * 64);
    sk_zero_molecules<R>(wv);
#ifdef SK_TIMING
    const long long tq_prol = (long long)__builtin_readcyclecounter() - tq_start;
#endif
    sk_main_loop<R, IBRD>(WN, wv, k0, cnt, valid, a.errflag, vsave);
#ifdef SK_TIMING
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0)
        printf("SK_TIMING block %d wave %d cnt %d: prologue %lld total %lld\n", (int)blockIdx.x, wv, cnt, tq_prol,
               (long long)__builtin_readcyclecounter() - tq_start);
#endif
    sk_molecule_sum<R>(wv, valid);
}

}  // namespace

namespace monortm_dev {
// states per workgroup / wavenumbers per tile of the state-lane kernel (api.hip sizes the grid with them)
int lines_state_tile(int nwn, int *ntiles) {
    const int nt = (nwn + SK_TILE - 1) / SK_TILE;
    *ntiles = nt;
    return (nwn + nt - 1) / nt;  // tiles of equal width <= 64
}
void launch_lines_state(const ModmArgs &a, const DevLines &L, const DevTables &tb, bool ibrd, hipStream_t s) {
    int ntiles = 1;
    const int tile_w = lines_state_tile(a.nwn, &ntiles);
    const long long nstates = (long long)a.nprof * a.nlay_max;
    const unsigned ngroups = (unsigned)((nstates + 63) / 64);
    const dim3 grid(ngroups * a.nslice, (unsigned)ntiles);
    const size_t dyn = sizeof(int) * (size_t)(2 * a.nmol + 2);
    if (a.real_kind == 4) {
        hipLaunchKernelGGL((state_tips_kernel<float>), dim3(ngroups), dim3(256), 0, s, a, tb);
        if (ibrd) hipLaunchKernelGGL((lines_state_kernel<float, true>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
        else hipLaunchKernelGGL((lines_state_kernel<float, false>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
    } else {
        hipLaunchKernelGGL((state_tips_kernel<double>), dim3(ngroups), dim3(256), 0, s, a, tb);
        if (ibrd) hipLaunchKernelGGL((lines_state_kernel<double, true>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
        else hipLaunchKernelGGL((lines_state_kernel<double, false>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
    }
}
}  // namespace monortm_dev
