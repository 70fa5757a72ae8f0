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
// Workgroup = 64 states x 8 waves (one tile of <= 64 wavenumbers).  Per chunk of CH lines: each wave PREPARES the records
// of its lines for the 64 states (the same line_physics_core() as lines_kernel: shifted centre, S~, widths, coupling
// factors) into LDS; after a barrier every wave EVALUATES all CH lines for its own wavenumbers, the lane reading the
// record of ITS state.  Two wavenumbers of a line share one reciprocal.  Per (state, wavenumber, molecule) the lines are
// added in table order = the reference's order (Voigt terms after the Lorentz terms of their chunk, as in lines_kernel).
// Chosen by api.hip for calls with many states; single profiles and dense grids keep lines_kernel.  DESIGN.md section 3.1b.
#include "lines_device.hpp"

namespace {
using namespace monortm_dev;

constexpr int SK_WAVES = 8;    // waves per workgroup
constexpr int SK_KW = 8;       // wavenumbers per wave (accumulator pairs per lane)
constexpr int SK_TILE = SK_WAVES * SK_KW;  // wavenumbers per workgroup
constexpr int SK_CH = 8;       // lines per chunk (one per wave)
constexpr int SK_VQ = 8 * 64;  // Voigt queue of a wave: what one line can produce
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
        sa += fma(q, denb, -pa);
        sb += fma(q, dena, -pa);
    } else if (KIND == 1) {
        sa = fma(q, denb, sa);
        sb = fma(q, dena, sb);
    } else {
        const double fa = fma(-(da * da), 1.0 / 625., 2.), fb = fma(-(db * db), 1.0 / 625., 2.);
        sa += fma(-pa, fa, q * denb);
        sb += fma(-pa, fb, q * dena);
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
template <int I>
__device__ __forceinline__ void sk_add_if(D8 &acc, int wi, bool mine, double vv) {
    el<I>(acc) += (wi == I && mine) ? vv : 0.;
}
// The out-of-line call needs ~100 registers of its own: the eight sums are parked in a scratch array in global memory
// around it (volatile: the compiler must not keep the values in registers across the call), so that the kernel as a whole
// stays within 128 VGPRs = four waves per SIMD.  Rare path.
__device__ __forceinline__ void sk_voigt_flush(const double (*sRec)[F_N][64], const unsigned short *vq, int nq, const float *sSdep,
                                               const unsigned *sFlag, const double *sWn, int k0, int mol, D8 &acc, int *errflag,
                                               volatile double *save /* [8][64] of this wave */) {
    const int lane = (int)__lane_id();
    save[0 * 64 + lane] = acc.a; save[1 * 64 + lane] = acc.b; save[2 * 64 + lane] = acc.c; save[3 * 64 + lane] = acc.d;
    save[4 * 64 + lane] = acc.e; save[5 * 64 + lane] = acc.f; save[6 * 64 + lane] = acc.g; save[7 * 64 + lane] = acc.h;
    for (int b0 = 0; b0 < nq; b0 += 64) {  // 64 triples at a time, one per lane
        const int nb = min(64, nq - b0);
        const double val = sk_voigt_values(sRec, vq + b0, nb, sSdep, sFlag, sWn, k0, mol, errflag);
        const unsigned rec = (lane < nb) ? (unsigned)vq[b0 + lane] : 0u;
        D8 t;
        t.a = save[0 * 64 + lane]; t.b = save[1 * 64 + lane]; t.c = save[2 * 64 + lane]; t.d = save[3 * 64 + lane];
        t.e = save[4 * 64 + lane]; t.f = save[5 * 64 + lane]; t.g = save[6 * 64 + lane]; t.h = save[7 * 64 + lane];
        for (int it = 0; it < nb; it++) {  // wave-uniform trip count and indices
            const int vlo = __builtin_amdgcn_readlane(__double2loint(val), it), vhi = __builtin_amdgcn_readlane(__double2hiint(val), it);
            const int rr = __builtin_amdgcn_readlane((int)rec, it);
            const int ow = rr & 63, wi = (rr >> 6) & 7;
            const double vv = __hiloint2double(vhi, vlo);
            const bool mine = lane == ow;
            sk_add_if<0>(t, wi, mine, vv); sk_add_if<1>(t, wi, mine, vv); sk_add_if<2>(t, wi, mine, vv); sk_add_if<3>(t, wi, mine, vv);
            sk_add_if<4>(t, wi, mine, vv); sk_add_if<5>(t, wi, mine, vv); sk_add_if<6>(t, wi, mine, vv); sk_add_if<7>(t, wi, mine, vv);
        }
        save[0 * 64 + lane] = t.a; save[1 * 64 + lane] = t.b; save[2 * 64 + lane] = t.c; save[3 * 64 + lane] = t.d;
        save[4 * 64 + lane] = t.e; save[5 * 64 + lane] = t.f; save[6 * 64 + lane] = t.g; save[7 * 64 + lane] = t.h;
    }
    acc.a = save[0 * 64 + lane]; acc.b = save[1 * 64 + lane]; acc.c = save[2 * 64 + lane]; acc.d = save[3 * 64 + lane];
    acc.e = save[4 * 64 + lane]; acc.f = save[5 * 64 + lane]; acc.g = save[6 * 64 + lane]; acc.h = save[7 * 64 + lane];
}

// Evaluate the prepared lines [0, nch) of the chunk for this wave's wavenumbers.  KIND: 0 generic molecule, 1 O2, 2 CO2.
struct SkLine {  // the lane's record of one line + the wave-uniform class bits of its (line, wavenumber) pairs
    double xnu, hw2, a2, pa;
    unsigned live, test, m2, m2t;
};
// ordinary line, wavenumbers I and I + 1 of the wave
template <int KIND, int I>
__device__ __forceinline__ void sk_step2(const SkLine &l, const D8 &WN, D8 &acc) {
    const unsigned b = (l.live >> I) & 3u;
    const unsigned special = ((l.test | l.m2) >> I) & 3u;
    if (b == 3u && special == 0u) {
        sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, el<I>(WN), el<I + 1>(WN), el<I>(acc), el<I + 1>(acc));
    } else {
        if (b & 1u)
            el<I>(acc) += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, el<I>(WN), (l.test >> I) & 1u, (KIND != 2) && ((l.m2 >> I) & 1u), (l.m2t >> I) & 1u);
        if (b & 2u)
            el<I + 1>(acc) += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, el<I + 1>(WN), (l.test >> (I + 1)) & 1u,
                                           (KIND != 2) && ((l.m2 >> (I + 1)) & 1u), (l.m2t >> (I + 1)) & 1u);
    }
}
// line with Y factors and / or Voigt candidates, wavenumber I of the wave
template <int KIND, int I>
__device__ __forceinline__ void sk_stepg(const SkLine &l, double pb, double d100, double c1, double gp1, double lim, bool voigt,
                                         bool valid, int jc, int lane, const D8 &WN, D8 &acc, unsigned short *vq, int &nq) {
    if ((l.live >> I) & 1u) {
        bool useV = false;
        double term = sk_general<KIND>(l.xnu, l.hw2, l.a2, l.pa, pb, d100, c1, gp1, lim, lim, el<I>(WN), voigt, &useV);
        if (voigt) {
            useV = useV && valid;
            const unsigned long long mv = __ballot(useV);
            if (mv != 0ull) {
                const int add = __popcll(mv);
                // (no per-lane branch: lanes that queue nothing write to a scratch slot; the queue holds the 8 x 64 triples one
                // line can produce and is worked off after the line, see sk_eval_chunk)
                vq[useV ? nq + __popcll(mv & ((1ull << lane) - 1ull)) : SK_VQ + lane] = (unsigned short)((jc << 9) | (I << 6) | lane);
                term = useV ? 0. : term;  // the Voigt value replaces the Lorentz term (modm.f90:427-432)
                nq += add;
            }
        }
        el<I>(acc) += term;
    }
}

template <int KIND>
__device__ __forceinline__ void sk_eval_chunk(const double (*sRec)[F_N][64], const unsigned (*sBits)[64], const unsigned *sFlag,
                                              const float *sSdep, const double *sWn, unsigned short *vq, int nch, int wv, int lane,
                                              int k0, int cnt, int mol, bool valid, const D8 &WN, D8 &acc, int *errflag,
                                              volatile double *save) {
    int nq = 0;
    const unsigned full = (1u << cnt) - 1u;  // (cnt >= 1 whenever a line is live)
    // class bits and flags of the chunk's lines: one LDS read per chunk (lane j holds line j), handed out by v_readlane;
    // the lane's record of line j + 1 is fetched while line j is evaluated
    const unsigned vbits = sBits[lane & (SK_CH - 1)][wv], vflag = sFlag[lane & (SK_CH - 1)];
    double nx = sRec[0][F_XNU][lane], nh = sRec[0][F_HW2][lane], na = sRec[0][F_A2][lane], np = (KIND == 1) ? 0. : sRec[0][F_PA][lane];
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
        nx = sRec[jn][F_XNU][lane]; nh = sRec[jn][F_HW2][lane]; na = sRec[jn][F_A2][lane];
        if (KIND != 1) np = sRec[jn][F_PA][lane];
        if (l.live != 0u) {
            if (!(fl & LF_GENERAL) && l.live == full && (l.test | l.m2) == 0u) {
                // the common case in ONE basic block (the scheduler interleaves the four reciprocal chains): every wavenumber
                // of the wave inside 25 cm-1 for every state, one resonance.  Positions past the wave's count repeat its last
                // wavenumber; their sums are never stored
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.a, WN.b, acc.a, acc.b);
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.c, WN.d, acc.c, acc.d);
                sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.e, WN.f, acc.e, acc.f);
                if (cnt > 6) sk_pair<KIND>(l.xnu, l.hw2, l.a2, l.pa, WN.g, WN.h, acc.g, acc.h);   // (wave-uniform)
            } else if (KIND != 2 && !(fl & LF_GENERAL) && l.live == full && l.m2 == full && (l.test | l.m2t) == 0u) {
                // ... and both resonances for every wavenumber and state
                acc.a += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.a, false, true, false);
                acc.b += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.b, false, true, false);
                acc.c += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.c, false, true, false);
                acc.d += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.d, false, true, false);
                acc.e += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.e, false, true, false);
                acc.f += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.f, false, true, false);
                if (cnt > 6) {
                    acc.g += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.g, false, true, false);
                    acc.h += sk_one<KIND>(l.xnu, l.hw2, l.a2, l.pa, l.pa, WN.h, false, true, false);
                }
            } else if (!(fl & LF_GENERAL)) {
                // ordinary line: no Y factors (c1 = g = 0, so both pedestals equal pa), not a Voigt candidate for any state
                sk_step2<KIND, 0>(l, WN, acc);
                sk_step2<KIND, 2>(l, WN, acc);
                sk_step2<KIND, 4>(l, WN, acc);
                sk_step2<KIND, 6>(l, WN, acc);
            } else {
                // Y factors and / or Voigt candidates: per-lane tests, one wavenumber at a time
                const double pb = sRec[jc][F_PB][lane], d100 = sRec[jc][F_D100][lane], c1 = sRec[jc][F_C1][lane], gp1 = sRec[jc][F_GP1][lane];
                const int code = (int)(fl >> 8) & 3;
                const bool voigt = (fl & LF_VOIGT) != 0u;
                const double lim = (KIND == 1 && code) ? __builtin_inf() : 25.;
#define SK_G(I) sk_stepg<KIND, I>(l, pb, d100, c1, gp1, lim, voigt, valid, jc, lane, WN, acc, vq, nq)
                SK_G(0); SK_G(1); SK_G(2); SK_G(3); SK_G(4); SK_G(5); SK_G(6); SK_G(7);
#undef SK_G
                if (nq > 0) {  // the line's Voigt shapes (they read this chunk's records): after its Lorentz terms
                    sk_voigt_flush(sRec, vq, nq, sSdep, sFlag, sWn, k0, mol, acc, errflag, save);
                    nq = 0;
                }
            }
        }
    }
}

__device__ __forceinline__ double bcast_d(double v, int src) {  // value of lane src (wave-uniform index) -> SGPR pair
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// ------------------------------------------------------------------------------------------------
// LDS of the workgroup (file scope: the out-of-line stages below reach it without arguments)
// ------------------------------------------------------------------------------------------------
// layer scalars, one copy for the eight waves (they all serve the same 64 states): the prepare stage loads them per line, the
// evaluate stage does not carry them
enum : int { LY_RHORAT = 0, LY_RP, LY_LNRT, LY_CTK, LY_DTINV, LY_RECTLC, LY_TMPDIF, LY_ILC, LY_TK, LY_WTOT, LY_N };
__shared__ double sRec[SK_CH][F_N][64];       // prepared records, [line of the chunk][field][state]
__shared__ double sIso[2][9][64];             // Q(296)/Q(T) and HWHM_D / Xnu per isotopologue of the current molecule, per state
__shared__ double sLy[LY_N][64];              // layer scalars per state
__shared__ double sRho7[MXBRD][64];           // rho_molec(1:7) per state (read with species broadening only)
__shared__ double sWn[SK_TILE];               // the tile's wavenumbers (ascending; positions past the end repeat the last)
__shared__ unsigned sBits[SK_CH][64];         // per (line, wave < 8): live | test << 8 | m2 << 16 | m2test << 24 for its wavenumbers
__shared__ unsigned sFlag[SK_CH];             // line flags (LF_*) | coupling code << 8
// table fields of the lines of the current / next chunk, [parity][field][line]: loaded by one wave (a coalesced load per
// field, one line per lane) while the previous chunk is evaluated, read back wave-uniformly by the wave that prepares a line
__shared__ double sFldD[2][2][SK_CH];         // XNU0, S0adj
__shared__ float sFldF[2][6][SK_CH];          // alfa, hwhm, epp, tmpalf, pshift, sdep
__shared__ unsigned sFldM[2][SK_CH];          // meta
__shared__ float sSdep[SK_CH];
__shared__ unsigned short sVq[SK_WAVES][SK_VQ + 64];  // per wave: queued (line, wavenumber, state) triples that take a Voigt shape (+ 64 scratch slots)
struct SkConst {  // constants of the launch the out-of-line stages need
    PhysParams pp;
    const double *tips_qoft, *tips_q296, *smass;
    const int *tips_isonm, *tips_offset;
    int *errflag;
    double padS;
    int ntw, kbase, krem;
};
__shared__ SkConst sC;

// ---- per molecule: TIPS + Doppler factor of its isotopologues per state (src/tips_2003.f90:60-296, modm.f90:442-454).
// Out of line: its table gathers and their registers stay out of the main loop.
__device__ __noinline__ void sk_molecule_setup(int mol, int wv, bool ok /* valid state, temperature in range */) {
    const int lane = (int)__lane_id();
    const int niso = min(9, sC.tips_isonm[mol - 1]);
    const double Tk = sLy[LY_TK][lane];
    for (int iso = wv + 1; iso <= 9; iso += SK_WAVES) {
        double sc = 0., dop = 0.;
        if (iso <= niso && ok) {
            bool bad = false;
            sc = tips_scor(sC.tips_isonm, sC.tips_offset, sC.tips_qoft, sC.tips_q296, mol, iso, Tk, &bad);
            if (bad) atomicOr(sC.errflag, ERRBIT_TEMP);
        }
        const double M = sC.smass[(mol - 1) * 9 + iso - 1];
        if (M > 0.) dop = sqrt(2. * log(2.) * ((K_BOLTZ * Tk) / (M / K_AVOGAD))) / K_CLIGHT;
        sIso[0][iso - 1][lane] = sc;
        sIso[1][iso - 1][lane] = dop;
    }
}

// ---- prepare one line of the chunk for the 64 states: records, class bits, flags -> LDS.  Out of line with its own register
// allocation: the main loop keeps only its eight sums across the call (in registers the callee preserves).
template <bool IBRD>
__device__ __noinline__ void sk_prepare_line(int idx, int mol, int jc, int par, double Wm, bool valid) {
    const int lane = (int)__lane_id();
    const double RADCT = K_PLANCK * K_CLIGHT / K_BOLTZ;
    LineFields lf;
    lf.xnu0 = uni_d(sFldD[par][0][jc]); lf.s0adj = uni_d(sFldD[par][1][jc]);
    lf.alfa = __uint_as_float(uni_u(__float_as_uint(sFldF[par][0][jc]))); lf.hwhm = __uint_as_float(uni_u(__float_as_uint(sFldF[par][1][jc])));
    lf.epp = __uint_as_float(uni_u(__float_as_uint(sFldF[par][2][jc]))); lf.tmpalf = __uint_as_float(uni_u(__float_as_uint(sFldF[par][3][jc])));
    lf.pshift = __uint_as_float(uni_u(__float_as_uint(sFldF[par][4][jc])));
    lf.meta = uni_u(sFldM[par][jc]);
    const float sdep_j = __uint_as_float(uni_u(__float_as_uint(sFldF[par][5][jc])));
    const uint32_t meta = lf.meta;
    const int iso = (meta >> 6) & 15, code = (meta >> 10) & 3;
    const double XIPSF = (iso >= 1 && iso <= 9) ? sIso[0][iso - 1][lane] : 0.;
    const double dopfac = sIso[1][((iso >= 1 && iso <= 9) ? iso : 1) - 1][lane];
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
        sRec[jc][F_XNU][lane] = Xnu;
        sRec[jc][F_HW2][lane] = HW2;
        sRec[jc][F_A2][lane] = A2;
        sRec[jc][F_PA][lane] = pa;
        if (yfac || anyV) {  // read by the general path only
            sRec[jc][F_PB][lane] = o2 ? 0. : (p * ((1. - c1 * 25.) + g));
            sRec[jc][F_D100][lane] = d100;
            sRec[jc][F_C1][lane] = c1;
            sRec[jc][F_GP1][lane] = 1. + g;
        }
        // ---- classes of the (line, wavenumber) pairs, lane = position in the tile ----
        const bool exempt = o2 && code != 0;  // coupled O2: both resonances everywhere, no cut (modm.f90:755-792)
        const double dk = fabs(wl - xnu0), sp = wl + xnu0;
        const unsigned long long mLive = exempt ? ~0ull : __ballot(in_t && !(dk > 25. + padS));
        const unsigned long long mSure = exempt ? ~0ull : __ballot(in_t && !(dk > 25. - padS));
        const unsigned long long mM2 = co2 ? 0ull : (exempt ? ~0ull : __ballot(in_t && sp <= 25. + padS));
        const unsigned long long mM2s = co2 ? 0ull : (exempt ? ~0ull : __ballot(in_t && sp <= 25. - padS));
        {
            // lane w < 8 packs the bits of wave w's wavenumbers [k0w, k0w + cntw) (the other lanes fill slots nobody reads: no
            // per-lane branch)
            const int lw = lane & (SK_WAVES - 1);
            const int k0w = lw * kbase + min(lw, krem), cntw = kbase + (lw < krem ? 1 : 0);
            const unsigned msk = (1u << cntw) - 1u;
            const unsigned lv = (unsigned)(mLive >> k0w) & msk, su = (unsigned)(mSure >> k0w) & msk;
            const unsigned m2 = (unsigned)(mM2 >> k0w) & msk & lv, m2s = (unsigned)(mM2s >> k0w) & msk;
            sBits[jc][lane] = lv | ((lv & ~su) << 8) | (m2 << 16) | ((m2 & ~m2s) << 24);
            sFlag[jc] = ((yfac || anyV) ? LF_GENERAL : 0u) | (anyV ? LF_VOIGT : 0u) | ((unsigned)code << 8);  // (same value from every lane)
            sSdep[jc] = sdep_j;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// grid = (groups of 64 states x line slices, wavenumber tiles); block = 8 waves; dynamic LDS = the molecule windows
// ------------------------------------------------------------------------------------------------
template <typename R, bool IBRD>
__global__ __launch_bounds__(SK_WAVES * 64, 4) void lines_state_kernel(ModmArgs a, DevLines L, DevTables tb, int tile_w) {
    __shared__ double sRed[SK_WAVES];
    extern __shared__ __attribute__((aligned(16))) int dyn_lds_i[];
    int *sLo = dyn_lds_i;            // [nmol]   first candidate line of the molecule
    int *sOff = sLo + a.nmol;        // [nmol+1] prefix sums of the candidate counts

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: tell the compiler (scalar loops and loads)
#ifdef SK_TIMING
    long long tq_start = (long long)__builtin_readcyclecounter(), tq_prep = 0, tq_b1 = 0, tq_eval = 0, tq_b2 = 0, tq_mol = 0, tq_flush = 0, tq_x;
#define SK_T(acc_) do { const long long t_ = (long long)__builtin_readcyclecounter(); acc_ += t_ - tq_x; tq_x = t_; } while (0)
#else
#define SK_T(acc_)
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
    bool t_bad;
    {
        const double Pk = valid ? (double)rp<R>(a.P)[pl] : K_P0, Tk = valid ? (double)rp<R>(a.T)[pl] : K_T0;
        // MODM calls TIPS_2003 for every layer and all nmol molecules (modm.f90:250): outside 70-3000 K the reference STOPs
        t_bad = valid && (Tk < 70. || Tk > 3000.);
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
                sRed[0] = mx;
                // |Xnu - XNU0| <= max_abs_shift * RHORAT for every entry, with or without species broadening (line_table.cpp);
                // the margin covers the roundings of the sums and differences involved
                sC.padS = L.max_abs_shift * fmax(1.0, mx) + 1e-9;
                sC.pp = phys_params(a, L);
                sC.tips_qoft = tb.tips_qoft; sC.tips_q296 = tb.tips_q296; sC.smass = tb.smass;
                sC.tips_isonm = tb.tips_isonm; sC.tips_offset = tb.tips_offset;
                sC.errflag = a.errflag;
                sC.ntw = ntw; sC.kbase = kbase; sC.krem = krem;
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
    if (I < cnt && inb) a.rft[pl * (size_t)nwn + (t0 + k0 + I)] = el<I>(WN) * tanh((RADCT * el<I>(WN)) / (2 * Tk));
        SK_W(0) SK_W(1) SK_W(2) SK_W(3) SK_W(4) SK_W(5) SK_W(6) SK_W(7)
#undef SK_W
    }
    // ---- candidate range of every molecule for this tile (as lines_kernel; zero columns are skipped per molecule below) ----
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
    const int total = sOff[nmol];
    const int vbeg = (int)(((long long)total * slice) / nslice), vend = (int)(((long long)total * (slice + 1)) / nslice);

    // this wave's 4 KB of the scratch array in which the sums are parked around the out-of-line Voigt shapes (sk_voigt_flush)
    volatile double *vsave = a.vsave + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * SK_WAVES + wv) * (SK_KW * 64);
    R *obm = (nslice == 1) ? wp<R>(a.O_BY_MOL) + pl * nmol * (size_t)nwn
                           : wp<R>(a.partial) + ((size_t)slice * a.nprof * a.nlay_max + pl) * nmol * (size_t)nwn;

#ifdef SK_TIMING
    const long long tq_prol = (long long)__builtin_readcyclecounter() - tq_start;
    tq_x = (long long)__builtin_readcyclecounter();
#endif
    // ================= molecule by molecule =========================================================
    for (int m = 0; m < nmol; m++) {
        const int mol = m + 1;
        const double Wm = valid ? (double)wk[m] : 0.;
        const int s0 = (int)uni_u((unsigned)max(sOff[m], vbeg)), s1 = (int)uni_u((unsigned)min(sOff[m + 1], vend));
        D8 acc = {0., 0., 0., 0., 0., 0., 0., 0.};
        // W_SPECIES == 0 -> OL = 0 without a walk (modm.f90:318-321): skipped when that holds for every state of the group
        if (s1 > s0 && __ballot(Wm != 0.) != 0ull) {
            __syncthreads();  // (the previous molecule's readers of sIso are done)
            sk_molecule_setup(mol, wv, valid && !t_bad);
            const int lo_m = (int)uni_u((unsigned)(sLo[m] - sOff[m]));
            // The table fields of a chunk's lines: one wave loads them, one line per lane (coalesced), WHILE the previous chunk is
            // evaluated (the duty rotates over the waves) and leaves them in LDS; the wave that prepares a line reads them back
            // wave-uniformly.  Nothing of the line table stays in registers.
            LineFields vf;
            float vsdep = 0.f;
            auto load_fields = [&](int first) {
                const int li = lo_m + min(first + (lane & (SK_CH - 1)), s1 - 1);
                vf = load_line_fields(L, li);
                vsdep = L.sdep[li];
            };
            auto store_fields = [&](int par) {
                if (lane < SK_CH) {
                    sFldD[par][0][lane] = vf.xnu0; sFldD[par][1][lane] = vf.s0adj;
                    sFldF[par][0][lane] = vf.alfa; sFldF[par][1][lane] = vf.hwhm; sFldF[par][2][lane] = vf.epp;
                    sFldF[par][3][lane] = vf.tmpalf; sFldF[par][4][lane] = vf.pshift; sFldF[par][5][lane] = vsdep;
                    sFldM[par][lane] = vf.meta;
                }
            };
            if (wv == 0) { load_fields(s0); store_fields(0); }
            __syncthreads();
            SK_T(tq_mol);
            int ck = 0;
            for (int base = s0; base < s1; base += SK_CH, ck++) {
                const int nch = min(SK_CH, s1 - base);
                const int par = ck & 1;
                // ================= prepare: this wave's lines of the chunk, for the 64 states ==================
                for (int jc = wv; jc < nch; jc += SK_WAVES) sk_prepare_line<IBRD>(lo_m + base + jc, mol, jc, par, Wm, valid);
                SK_T(tq_prep);
                __syncthreads();
                SK_T(tq_b1);
                // next chunk's fields: the loads are in flight during the evaluate stage of the wave on duty, which has registers
                // to spare; they go to the other half of the staging area (whose last readers passed this barrier) after it
                const bool duty = base + SK_CH < s1 && wv == ((ck + 1) & (SK_WAVES - 1));
                if (duty) load_fields(base + SK_CH);
                // ================= evaluate: every line of the chunk for this wave's wavenumbers ===============
                if (mol == 7) sk_eval_chunk<1>(sRec, sBits, sFlag, sSdep, sWn, sVq[wv], nch, wv, lane, k0, cnt, mol, valid, WN, acc, a.errflag, vsave);
                else if (mol == 2) sk_eval_chunk<2>(sRec, sBits, sFlag, sSdep, sWn, sVq[wv], nch, wv, lane, k0, cnt, mol, valid, WN, acc, a.errflag, vsave);
                else sk_eval_chunk<0>(sRec, sBits, sFlag, sSdep, sWn, sVq[wv], nch, wv, lane, k0, cnt, mol, valid, WN, acc, a.errflag, vsave);
                if (duty) store_fields(par ^ 1);
                SK_T(tq_eval);
                __syncthreads();  // the records are overwritten by the next chunk
                SK_T(tq_b2);
            }
        }
        // ---- run complete: O_BY_MOL = RFT * (W * SF)   (modm.f90:436-438); zero for molecules without lines / column ----
        if (inb) {  // (layers beyond nlay[p] receive zeros, as from lines_kernel)
#define SK_O(I)                                                                                                           \
    if (I < cnt) {                                                                                                         \
        const size_t iw = (size_t)(t0 + k0 + I);                                                                           \
        const R od = (Wm == 0. || !valid) ? (R)0 : (R)(a.rft[pl * (size_t)nwn + iw] * (Wm * el<I>(acc)));                  \
        obm[(size_t)m * nwn + iw] = od;                                                                                    \
    }
            SK_O(0) SK_O(1) SK_O(2) SK_O(3) SK_O(4) SK_O(5) SK_O(6) SK_O(7)
#undef SK_O
        }
    }
#ifdef SK_TIMING
    SK_T(tq_flush);
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0)
        printf("SK_TIMING block %d wave %d cnt %d: prologue %lld molecule-setup %lld prepare %lld barrier1 %lld evaluate %lld barrier2 %lld flush+rest %lld total %lld\n",
               (int)blockIdx.x, wv, cnt, tq_prol, tq_mol, tq_prep, tq_b1, tq_eval, tq_b2, tq_flush, (long long)__builtin_readcyclecounter() - tq_start);
#endif
    if (a.osum && valid) {
        // sum over the molecules of O_BY_MOL as stored, in molecule order (modm.f90:264-269), for the finish kernel: read back
        // from what this thread has just written (no registers held across the kernel for it)
#define SK_S(I)                                                                       \
    if (I < cnt) {                                                                    \
        const size_t iw = (size_t)(t0 + k0 + I);                                      \
        double sm = 0.;                                                               \
        for (int m = 0; m < nmol; m++) sm += (double)obm[(size_t)m * nwn + iw];       \
        a.osum[pl * (size_t)nwn + iw] = sm;                                           \
    }
        SK_S(0) SK_S(1) SK_S(2) SK_S(3) SK_S(4) SK_S(5) SK_S(6) SK_S(7)
#undef SK_S
    }
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
    const dim3 grid((unsigned)(((nstates + 63) / 64) * a.nslice), (unsigned)ntiles);
    const size_t dyn = sizeof(int) * (size_t)(2 * a.nmol + 2);
    if (a.real_kind == 4) {
        if (ibrd) hipLaunchKernelGGL((lines_state_kernel<float, true>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
        else hipLaunchKernelGGL((lines_state_kernel<float, false>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
    } else {
        if (ibrd) hipLaunchKernelGGL((lines_state_kernel<double, true>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
        else hipLaunchKernelGGL((lines_state_kernel<double, false>), grid, dim3(SK_WAVES * 64), dyn, s, a, L, tb, tile_w);
    }
}
}  // namespace monortm_dev
