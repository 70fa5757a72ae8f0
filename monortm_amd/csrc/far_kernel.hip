// far_kernel.hip - the far field of dense wavenumber grids, formed once per interval of wavenumbers instead of chunk by chunk inside
// lines_kernel (round 5).  Reference arithmetic: the Lorentz terms of src/modm.f90:706-831 for lines whose 25 cm-1 window holds
// the whole interval; see lines_device.hpp ("Far field of a tile") for the Chebyshev series.
//
// Why.  On configs[2] (10000 wavenumbers x 100000 lines) four fifths of the lines in a tile's window are far lines, and lines_kernel
// spent 46 % of its vector instructions in its prepare stage - reading their records, classifying them, and adding their series to
// the tile's sums with a cross-lane butterfly per four coefficients and 64 lines (profiles/r05_k_abl_c3.txt).  Here
//   * a lane walks MANY far lines and keeps the 56 sums to itself (32 in registers, 24 in LDS): three instructions per pole and
//     term, no butterfly until the very end (one per interval and molecule);
//   * intervals come in levels - tiles, pairs of tiles, groups of four, ... - and a line is expanded ONCE by the largest interval
//     for which it is far; a child adds its parent's series, re-expanded about its own centre (exactly: a polynomial of degree
//     FAR_P - 1 sampled at FAR_P Chebyshev nodes), to its own;
//   * lines_kernel never reads a far line: far_plan_kernel hands it the runs of table lines that are left (farseg), and the
//     molecule's finished series (farmom) is one more source of its Clenshaw sum.
// Workgroups of one molecule land on one XCD, or on a few in proportion to its lines (far_xcd_share, device_common.hpp): the
// intervals of a layer share the molecule's records in that L2.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "lines_device.hpp"

namespace {
using namespace monortm_dev;

struct FarIv { double a, b, c, rho; };
__device__ __forceinline__ double uni_f64(double x) {   // wave-uniform value -> SGPR pair
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// interval (level l, index j): tiles [j 2^l, (j + 1) 2^l) and its ends on the wavenumber grid
__device__ __forceinline__ FarIv far_interval(const ModmArgs &a, int l, int j) {
    const int ntile = a.far_ntile, tw = a.far_tw, nwn = a.nwn;
    const int t0 = j << l, t1 = min(ntile, (j + 1) << l);
    FarIv v;
    v.a = a.wn[t0 * tw];
    v.b = a.wn[min(nwn, t1 * tw) - 1];
    v.c = 0.5 * (v.a + v.b);       // (lines_kernel forms the centre and half-width of its tile with these expressions)
    v.rho = 0.5 * (v.b - v.a);
    return v;
}

// ------------------------------------------------------------------------------------------------
// far_plan_kernel: grid = (ceil(intervals of all levels x molecules / 64), profiles, layers), lane = (interval, molecule).  FarGeom of
// every (interval, molecule) and, for the tiles, the runs of candidate lines that are not far (what lines_kernel walks).
// ------------------------------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(64) void far_plan_kernel(ModmArgs a, DevLines L, DevTables tb) {
    const int prof = blockIdx.y, lay = blockIdx.z;
    const int ntile = a.far_ntile, nmol = a.nmol;
    const int item = (int)blockIdx.x * 64 + (int)threadIdx.x, gi = item / nmol, m = item - gi * nmol;
    if (lay >= a.nlay[prof] || gi >= a.far_ni) return;
    int l = 0, off = 0;
    while (l + 1 < a.far_levels && gi >= off + far_level_count(ntile, l)) { off += far_level_count(ntile, l); l++; }
    const FarIv iv = far_interval(a, l, gi - off);
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    // the layer state exactly as lines_kernel's prologue forms it (modm.f90:868-883, :301-314)
    const double Pk = rp<R>(a.P)[pl], Tk = rp<R>(a.T)[pl], wbrod = rp<R>(a.WBRODL)[pl];
    const R *wk = rp<R>(a.WKL) + pl * nmol;
    const double XN0 = (K_P0 / (K_BOLTZ * K_T0)) * 1.E+3;
    const double Xn = (Pk / (K_BOLTZ * Tk)) * 1.E+3;
    double WTOT = 0.;
    for (int q = 0; q < nmol; q++) WTOT += wk[q];
    WTOT = WTOT + wbrod;
    const double RHORAT = Xn / XN0;
    const double pad = L.max_abs_shift * fmax(RHORAT, 1.0) + 1e-6;
    const int mol = m + 1, ms0 = L.mol_start[mol], ms1 = L.mol_start[mol + 1];
    const double W = (double)wk[m];
    const bool windowed = (mol != 7 || !((L.lc_mask >> 7) & 1ull)) && ((L.sorted_mask >> mol) & 1ull) && WTOT == WTOT && RHORAT == RHORAT && Tk == Tk;
    // far lines: sorted uncoupled molecules in a finite state, an interval of more than one wavenumber, and no far line within 100
    // Doppler widths of any wavenumber of the interval (modm.f90:427: such a line could take the Voigt shape) - bounded with the
    // lightest isotopologue and the molecule's last line: a far line is (kappa - 1) rho away from every wavenumber at least.
    // An interval that fails the test (narrow tiles in the infrared, the stub of a last tile) INHERITS the far lines of its nearest
    // ancestor that passes - the ancestor expands them for all its wavenumbers, the descendant expands nothing of its own and its
    // tile does not walk them.  (A wider interval passes whenever a narrower one inside it does: a child's set always contains its
    // parent's.)
    const bool molok = W != 0. && windowed && !((L.lc_mask >> mol) & 1ull) && ms1 > ms0;
    double hwdmax = 0.;
    if (molok) {
        double dopmax = 0.;
        for (int i = 0; i < 9; i++) {
            const double M = tb.smass[(mol - 1) * 9 + i];
            if (M > 0.) dopmax = fmax(dopmax, doppler_factor(M, Tk));
        }
        hwdmax = (L.vnu[ms1 - 1] + pad) * dopmax;
    }
    // nine searches in lock step: s < 7 the FarGeom entries, 7 / 8 the candidate window of a tile
    FarGeom g{0, 0, 0, 0, 0, 0, 0, 0};
    int clo = ms0, chi = ms1;
    FarIv cur = iv;
    for (int ll = l, jj = gi - off; ll < a.far_levels; ll++, jj >>= 1) {
        if (ll != l) cur = far_interval(a, ll, jj);
        const bool farok = molok && cur.rho > 0. && (FAR_KAPPA - 1.0) * cur.rho - 2. * pad > 100. * hwdmax * 1.000001;
        if (!farok && ll != l) continue;   // (the searches of the interval itself also find a tile's candidate window)
        const double kr = FAR_KAPPA * cur.rho;
        const double key[9] = {cur.b - 25. + pad, cur.c - kr - pad, cur.c + kr + pad, cur.a + 25. - pad, kr - cur.c + pad, 25. - cur.b - pad,
                               25. - cur.a + pad, cur.a - 25.0 - pad, cur.b + 25.0 + pad};
        const bool incl[9] = {false, true, false, true, false, true, true, false, true};   // count vnu <= key (true) or vnu < key
        int lo[9], hi[9];
#pragma unroll
        for (int s = 0; s < 9; s++) { lo[s] = ms0; hi[s] = ms1; }
        for (int it = 0; it < 40; it++) {
            bool any = false;
#pragma unroll
            for (int s = 0; s < 9; s++)
                if (lo[s] < hi[s]) {
                    const int mid = (lo[s] + hi[s]) >> 1;
                    const double v = L.vnu[mid];
                    if (incl[s] ? (v <= key[s]) : (v < key[s])) lo[s] = mid + 1;
                    else hi[s] = mid;
                    any = true;
                }
            if (!any) break;
        }
        if (ll == l) { clo = lo[7]; chi = max(lo[7], lo[8]); }
        if (farok) {
            g.lowS = lo[0]; g.lowE = lo[1]; g.highS = lo[2]; g.highE = lo[3];
            if (mol == 2) { g.e0 = ms0; g.e1s = ms0; g.e1e = ms0; }   // CO2 has no negative resonance (modm.f90:808-817)
            else { g.e0 = lo[4]; g.e1s = lo[5]; g.e1e = max(lo[5], lo[6]); }
            break;
        }
    }
    int *go = a.fargeom + ((pl * (size_t)a.far_ni + gi) * nmol + m) * FAR_GEOM_INTS;
    go[0] = g.lowS; go[1] = g.lowE; go[2] = g.highS; go[3] = g.highE; go[4] = g.e0; go[5] = g.e1s; go[6] = g.e1e; go[7] = 0;
    // "nothing there" until far_kernel says otherwise: it starts no workgroup for a molecule without lines, and lines_kernel
    // reads the flag of every molecule of its tile
    {
        double *mo = a.farmom + ((pl * (size_t)a.far_ni + gi) * nmol + m) * FAR_MOM_STRIDE;
        mo[FAR_P] = 0.;
        mo[FAR_P + 1] = 0.;
    }
    if (l != 0) return;
    // tile: candidates (lines_kernel's rule: W = 0 -> none, modm.f90:318-321; the 25 cm-1 window for sorted molecules without
    // coupled O2 in a finite state, else the whole run) minus the far lines, as at most FAR_SEGS runs
    if (W == 0.) { clo = ms0; chi = ms0; }
    else if (!windowed) { clo = ms0; chi = ms1; }
    int seg_base[FAR_SEGS] = {0, 0, 0, 0, 0}, seg_cum[FAR_SEGS] = {0, 0, 0, 0, 0};
    int nseg = 0, cum = 0, pos = clo;
    bool over = false;
    const int bp[7] = {g.lowS, g.lowE, g.highS, g.highE, g.e0, g.e1s, g.e1e};
    while (pos < chi) {
        int next = chi;
#pragma unroll
        for (int s = 0; s < 7; s++)
            if (bp[s] > pos && bp[s] < next) next = bp[s];
        if (!far_contains(g, pos)) {
            // (a run that continues the previous one - a breakpoint without a change of status - extends it)
            bool joined = false;
#pragma unroll
            for (int k = 0; k < FAR_SEGS; k++)
                if (k == nseg - 1 && seg_base[k] + seg_cum[k] == pos) { seg_cum[k] += next - pos; joined = true; }
            if (!joined) {
                if (nseg >= FAR_SEGS) over = true;
                else {
#pragma unroll
                    for (int k = 0; k < FAR_SEGS; k++)
                        if (k == nseg) { seg_base[k] = pos - cum; seg_cum[k] = cum + (next - pos); }
                    nseg++;
                }
            }
            cum += next - pos;
        }
        pos = next;
    }
    if (over) atomicOr(a.errflag, ERRBIT_ARG);   // (cannot happen: two sides, each cut at most once - at most four far runs)
    int *so = a.farseg + ((pl * (size_t)ntile + gi) * nmol + m) * FAR_SEG_INTS;
#pragma unroll
    for (int k = 0; k < FAR_SEGS; k++) {
        so[k] = (k < nseg) ? seg_base[k] : 0;
        so[FAR_SEGS + k] = (k < nseg) ? seg_cum[k] : cum;
    }
}

// The series of one wave step.  A lane carries up to four POLES at a time - both resonances of two lines, or the one resonance of
// two lines in slots 0 and 1 - and adds u_n = amp Im'[g w^n] of each to its own FAR_P sums.  g w^n and its conjugate solve
//   u_{n+1} = 2 Re(w) u_n - |w|^2 u_{n-1},
// so a term costs an add, a product and a multiply-add per pole where the complex product of far_series (lines_device.hpp) takes
// five instructions; both roots have modulus |w| < 1, the rounding errors decay with the sequence.  x = u_n and y = u_{n+1} swap
// roles from term to term (no moves):  c_n += x;  x <- A y + B x.
// A group of four terms is ONE asm statement, its wave-uniform branches ("order reached?", "two slots or four?") included: every
// sum is tied to a register ("+v").  Written in C++ with a predicate per group the compiler gave each group's results fresh
// registers and copied them back on the path around it - two sets of sums, 1.8 KB of scratch per lane.  Inside a term
// the products of all slots come first, then their multiply-adds: an instruction does not read the result of the one before it
// (a wave issues in order, and two waves per SIMD hide little).
struct FarSlot { double x, y, A, B; };
#define FAR_AM(X, S, C)                                   \
    "v_add_f64 " C ", " C ", %[" X S "]\n\t"               \
    "v_mul_f64 %[t" S "], %[B" S "], %[" X S "]\n\t"
#define FAR_FM(X, Y, S) "v_fma_f64 %[" X S "], %[A" S "], %[" Y S "], %[t" S "]\n\t"
#define FAR_T4(X, Y, C) FAR_AM(X, "0", C) FAR_AM(X, "1", C) FAR_AM(X, "2", C) FAR_AM(X, "3", C) FAR_FM(X, Y, "0") FAR_FM(X, Y, "1") FAR_FM(X, Y, "2") FAR_FM(X, Y, "3")
#define FAR_T2(X, Y, C) FAR_AM(X, "0", C) FAR_AM(X, "1", C) FAR_FM(X, Y, "0") FAR_FM(X, Y, "1")
template <int LIM>
__device__ __forceinline__ void far_group4(double &c0, double &c1, double &c2, double &c3, FarSlot &s0, FarSlot &s1, FarSlot &s2, FarSlot &s3,
                                           int order, int four) {
    double t0, t1, t2, t3;
    asm volatile("s_cmp_le_i32 %[ord], %[lim]\n\ts_cbranch_scc1 9f\n\t"
                 "s_cmp_eq_u32 %[four], 0\n\ts_cbranch_scc1 5f\n\t"
                 FAR_T4("x", "y", "%[c0]") FAR_T4("y", "x", "%[c1]") FAR_T4("x", "y", "%[c2]") FAR_T4("y", "x", "%[c3]")
                 "s_branch 9f\n\t"
                 "5:\n\t"
                 FAR_T2("x", "y", "%[c0]") FAR_T2("y", "x", "%[c1]") FAR_T2("x", "y", "%[c2]") FAR_T2("y", "x", "%[c3]")
                 "9:"
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [x0] "+v"(s0.x), [y0] "+v"(s0.y), [x1] "+v"(s1.x), [y1] "+v"(s1.y),
                   [x2] "+v"(s2.x), [y2] "+v"(s2.y), [x3] "+v"(s3.x), [y3] "+v"(s3.y), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [A0] "v"(s0.A), [B0] "v"(s0.B), [A1] "v"(s1.A), [B1] "v"(s1.B), [A2] "v"(s2.A), [B2] "v"(s2.B), [A3] "v"(s3.A), [B3] "v"(s3.B),
                   [ord] "s"(order), [four] "s"(four), [lim] "n"(LIM)
                 : "scc");
}
// The sums FAR_PREG .. FAR_P - 1 live in LDS, one slot per lane and sum ([sum][lane]: no bank conflicts), and take their terms
// through ds_add_f64: only a step whose nearest line lies within ~1.8 half-widths gets that far (a fifth of the steps below the
// top level), and with 64 instead of 112 registers for the sums the kernel runs three waves per SIMD instead of two - it is bound
// by the latency of its dependent FP64 chains, not by their number (measured with the series cut at 32 terms: 2 -> 3 waves per
// SIMD took every level from 0.28-0.47 to 0.21-0.36 ms; cutting the terms alone changed nothing).
constexpr int FAR_PREG = 32;
#define FAR_LS4(X, OFF)                                                          \
    "v_add_f64 %[t2], %[" X "0], %[" X "1]\n\t"                                    \
    "v_add_f64 %[t3], %[" X "2], %[" X "3]\n\t"                                    \
    "v_add_f64 %[t2], %[t2], %[t3]\n\t"                                            \
    "ds_add_f64 %[addr], %[t2] offset:%[" OFF "]\n\t"
#define FAR_LS2(X, OFF)                                                          \
    "v_add_f64 %[t2], %[" X "0], %[" X "1]\n\t"                                    \
    "ds_add_f64 %[addr], %[t2] offset:%[" OFF "]\n\t"
#define FAR_MU(X, S) "v_mul_f64 %[t" S "], %[B" S "], %[" X S "]\n\t"
#define FAR_L4(X, Y, OFF) FAR_LS4(X, OFF) FAR_MU(X, "0") FAR_MU(X, "1") FAR_MU(X, "2") FAR_MU(X, "3") FAR_FM(X, Y, "0") FAR_FM(X, Y, "1") FAR_FM(X, Y, "2") FAR_FM(X, Y, "3")
#define FAR_L2(X, Y, OFF) FAR_LS2(X, OFF) FAR_MU(X, "0") FAR_MU(X, "1") FAR_FM(X, Y, "0") FAR_FM(X, Y, "1")
// addr: LDS byte address of this lane's slot of sum FAR_PREG; LIM: first term of the group
template <int LIM>
__device__ __forceinline__ void far_group4_lds(unsigned addr, FarSlot &s0, FarSlot &s1, FarSlot &s2, FarSlot &s3, int order, int four) {
    double t0, t1, t2, t3;
    asm volatile("s_cmp_le_i32 %[ord], %[lim]\n\ts_cbranch_scc1 9f\n\t"
                 "s_cmp_eq_u32 %[four], 0\n\ts_cbranch_scc1 5f\n\t"
                 FAR_L4("x", "y", "o0") FAR_L4("y", "x", "o1") FAR_L4("x", "y", "o2") FAR_L4("y", "x", "o3")
                 "s_branch 9f\n\t"
                 "5:\n\t"
                 FAR_L2("x", "y", "o0") FAR_L2("y", "x", "o1") FAR_L2("x", "y", "o2") FAR_L2("y", "x", "o3")
                 "9:"
                 : [x0] "+v"(s0.x), [y0] "+v"(s0.y), [x1] "+v"(s1.x), [y1] "+v"(s1.y), [x2] "+v"(s2.x), [y2] "+v"(s2.y), [x3] "+v"(s3.x), [y3] "+v"(s3.y),
                   [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [A0] "v"(s0.A), [B0] "v"(s0.B), [A1] "v"(s1.A), [B1] "v"(s1.B), [A2] "v"(s2.A), [B2] "v"(s2.B), [A3] "v"(s3.A), [B3] "v"(s3.B),
                   [addr] "v"(addr), [ord] "s"(order), [four] "s"(four), [lim] "n"(LIM), [o0] "n"((LIM - FAR_PREG) * 512), [o1] "n"((LIM - FAR_PREG + 1) * 512),
                   [o2] "n"((LIM - FAR_PREG + 2) * 512), [o3] "n"((LIM - FAR_PREG + 3) * 512)
                 : "scc", "memory");
}
template <int P, int G = 0>
__device__ __forceinline__ void far_accumulate(double (&c)[FAR_PREG], unsigned addr, int order, int four, FarSlot &s0, FarSlot &s1, FarSlot &s2, FarSlot &s3) {
    if constexpr (4 * G < P) {
        if constexpr (4 * G < FAR_PREG) far_group4<4 * G>(c[4 * G], c[4 * G + 1], c[4 * G + 2], c[4 * G + 3], s0, s1, s2, s3, order, four);
        else far_group4_lds<4 * G>(addr, s0, s1, s2, s3, order, four);
        far_accumulate<P, G + 1>(c, addr, order, four, s0, s1, s2, s3);
    }
}
// The slot of one pole at distance delta from the interval's centre (|delta| >= kappa rho; far_pole of lines_device.hpp restated
// without its branch and with one reciprocal instead of three): z = (delta + i h) / rho, s = sqrt(z^2 - 1), w = 1 / (z + s), g = 2 / s
// in (re, im / h) form, then u_0 = amp Im'(g), u_1 = amp Im'(g w) (Im'[(a + i h a')(b + i h b')] = a b' + a' b), A = 2 Re w, B = -|w|^2.
// No branch: the four poles of a step are independent chains of ~50 instructions with two square roots and a reciprocal each, and
// under `if (on)` they ran one after the other (a lane without a line passes delta = 2 rho, h = 0, amp = 0).
__device__ __forceinline__ FarSlot far_slot_of(double delta, double hw2, double rinv, double amp) {
    const double zr = delta * rinv, r2 = rinv * rinv;
    const double A = fma(zr, zr, -fma(hw2, r2, 1.0));   // Re(z^2 - 1) = (delta^2 - h^2) / rho^2 - 1
    const double Bp = 2.0 * zr * rinv;                  // Im(z^2 - 1) / h
    const double mod = fsqrt_pos(fma(A, A, (Bp * Bp) * hw2));
    // Re sqrt = sqrt(X), X = (|z^2 - 1| + A) / 2, with e = 1 / (2 sqrt(X)) from the same Newton steps (fsqrt_pos)
    // (A < 0 - a Lorentz width beyond 0.66 rho, narrow infrared tiles in the lowest layers: mod + A cancels, the same number without
    // the cancellation is (Im(z^2 - 1))^2 / (2 (mod - A)); wave-uniform test, rarely true)
    double X = 0.5 * (mod + A);
    if (__ballot(A < 0.) != 0ull) X = (A < 0.) ? 0.5 * ((Bp * Bp) * hw2) * frcp_any(mod - A) : X;
    const double r = __builtin_amdgcn_rsq(X);
    double y = X * r, e = 0.5 * r;
    y = fma(fma(-y, y, X), e, y);
    e = fma(fma(-2.0 * e, y, 1.0), e, e);
    y = fma(fma(-y, y, X), e, y);
    e = fma(fma(-2.0 * e, y, 1.0), e, e);
    const double sre = (delta < 0.) ? -y : y, einv = (delta < 0.) ? -e : e;   // the branch with |w| < 1 has the sign of delta
    const double sim = Bp * einv;                       // Im sqrt / h
    const double ure = zr + sre, uim = rinv + sim;      // z + s (no cancellation: same signs)
    const double D1 = fma(ure, ure, (uim * uim) * hw2), D2 = fma(sre, sre, (sim * sim) * hw2);
    const double rr = frcp_any(D1 * D2);
    const double dinv = rr * D2, sinv = (rr + rr) * D1;
    const double wre = ure * dinv, wim = -uim * dinv, gre = sre * sinv, gim = -sim * sinv;
    return FarSlot{amp * gim, amp * fma(gre, wim, gim * wre), wre + wre, -fma(wre, wre, (hw2 * wim) * wim)};
}

// ------------------------------------------------------------------------------------------------
// far_kernel: grid = (intervals of ONE level x molecules padded to a multiple of 8, profiles, layers), one wave per workgroup.
// Levels are launched top first: a child reads its parent's finished sums.
// ------------------------------------------------------------------------------------------------
// NWF waves per workgroup share the far lines of their (interval, molecule): wave w takes the steps w, w + NWF, ... of every run.
// One wave alone walks up to 7000 lines of a group of four tiles, 110 steps of ~600 dependent instructions = 0.3 ms - longer than
// the whole kernel should take; the waves' sums are added in wave order (deterministic).
#ifndef FAR_OCC
#define FAR_OCC 3   // waves per SIMD the kernel is compiled for (168 vector registers)
#endif
template <typename R, int NWF>
__global__ __launch_bounds__(NWF * 64, FAR_OCC) void far_kernel(ModmArgs a, DevLines L, int level, FarPlace place) {
    constexpr int P = FAR_P;
    __shared__ double sPart[NWF][P + 4];   // per wave: its sums, pedestal sum, the three CO2 sums
    __shared__ double sHi[NWF][P - FAR_PREG][64];   // the sums FAR_PREG .. P - 1 of every lane (ds_add_f64)
    static_assert(P % 4 == 0 && P <= 64, "groups of four sums, one Chebyshev node per lane");
    constexpr bool SGL = sizeof(R) == 4;
    const int nmol = a.nmol, prof = blockIdx.y, lay = blockIdx.z;
    if (lay >= a.nlay[prof]) return;
    // workgroup -> (interval j, molecule m): the slot-th item of XCD k (far_xcd_share, device_common.hpp)
    int m = -1, j = 0;
    {
        const int k = (int)blockIdx.x & 7;
        int slot = (int)blockIdx.x >> 3;
        for (int q = 0; q < nmol; q++) {
            const int cnt = place.cnt[q][k];
            if (slot < cnt) { m = q; j = (k - (int)place.xlo[q]) + slot * (int)place.nx[q]; break; }
            slot -= cnt;
        }
    }
    if (m < 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), ntile = a.far_ntile;
    const int gi = far_level_offset(ntile, level) + j;
    const bool has_parent = level + 1 < a.far_levels;
    const int gip = has_parent ? far_level_offset(ntile, level + 1) + (j >> 1) : 0;
    const size_t pl = (size_t)prof * a.nlay_max + lay;
    const int *gq = a.fargeom + ((pl * (size_t)a.far_ni + gi) * nmol + m) * FAR_GEOM_INTS;
    const int *gpq = a.fargeom + ((pl * (size_t)a.far_ni + gip) * nmol + m) * FAR_GEOM_INTS;
    // (wave-uniform values into scalar registers: the sums need the vector registers)
#define FAR_UNI(x) __builtin_amdgcn_readfirstlane(x)
    FarGeom g{FAR_UNI(gq[0]), FAR_UNI(gq[1]), FAR_UNI(gq[2]), FAR_UNI(gq[3]), FAR_UNI(gq[4]), FAR_UNI(gq[5]), FAR_UNI(gq[6]), 0}, gp{0, 0, 0, 0, 0, 0, 0, 0};
    if (has_parent) gp = FarGeom{FAR_UNI(gpq[0]), FAR_UNI(gpq[1]), FAR_UNI(gpq[2]), FAR_UNI(gpq[3]), FAR_UNI(gpq[4]), FAR_UNI(gpq[5]), FAR_UNI(gpq[6]), 0};
#undef FAR_UNI
    FarIv iv = far_interval(a, level, j);
    iv.c = uni_f64(iv.c);
    iv.rho = uni_f64(iv.rho);
    const double c0 = iv.c, rinv = uni_f64((iv.rho > 0.) ? frcp_any(iv.rho) : 0.);
    const int mol = m + 1;
    const bool co2 = mol == 2, o2 = mol == 7;
    const double wsc = SGL ? uni_f64((double)(rp<R>(a.WKL) + pl * nmol)[m]) : 1.0;   // single precision: the amplitudes carry the column (line_records)
    const LinePhysM *phys = reinterpret_cast<const LinePhysM *>(a.phys) + pl * (size_t)a.phys_lines;   // (uncoupled molecules: no Y factors)
#ifdef FAR_ABL_HOT   // timing experiment (wrong results): every record read comes from the same 48 KB
#define FAR_REC(i) ((i) & 1023)
#else
#define FAR_REC(i) (i)
#endif

    static_assert(FAR_PREG % 4 == 0 && FAR_PREG < P, "the first sums in registers, the others in LDS");
    constexpr int PL = P - FAR_PREG;
    double c[FAR_PREG];
#pragma unroll
    for (int n = 0; n < FAR_PREG; n++) c[n] = 0.;
#pragma unroll
    for (int n = 0; n < PL; n++) sHi[wave][n][lane] = 0.;
    const unsigned hi_addr = lds_addr(&sHi[wave][0][lane]);
    double ped = 0., q0 = 0., q1 = 0., q2 = 0.;
    bool any = false;
    const int end = __builtin_amdgcn_readfirstlane(L.mol_start[mol + 1]);
    int pos = __builtin_amdgcn_readfirstlane(L.mol_start[mol]);
    const int bp[14] = {g.lowS, g.lowE, g.highS, g.highE, g.e0, g.e1s, g.e1e, gp.lowS, gp.lowE, gp.highS, gp.highE, gp.e0, gp.e1s, gp.e1e};
    while (pos < end) {
        int next = end;
#pragma unroll
        for (int s = 0; s < 14; s++)
            if (bp[s] > pos && bp[s] < next) next = bp[s];
#ifdef FAR_ABL_STEPS
        if (far_contains(g, pos) && !far_contains(gp, pos)) any = true;
        if (false) {
#else
        if (far_contains(g, pos) && !far_contains(gp, pos)) {
#endif
            any = true;
            const bool two = !co2 && pos < g.e1s;   // both resonances for every wavenumber of the interval (modm.f90:713)
            // two lines a lane and step: i and i + 64.  (The records of the NEXT step travel while this one is expanded: the first
            // use of a record is an HBM read, and two waves per SIMD hide nothing.)
            constexpr int STEP = 128 * NWF;
            double nx[2] = {0., 0.}, nh[2] = {1., 1.}, ns[2] = {0., 0.};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = pos + 128 * wave + 64 * u + lane;
                if (i < next) { nx[u] = phys[FAR_REC(i)].xnu; nh[u] = phys[FAR_REC(i)].hw; ns[u] = phys[FAR_REC(i)].stild; }
            }
            for (int i0 = pos + 128 * wave; i0 < next; i0 += STEP) {
                double xnu[2], hw[2], st[2];
                bool on[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = i0 + 64 * u + lane, in = i + STEP;
                    on[u] = i < next;
                    xnu[u] = nx[u]; hw[u] = nh[u]; st[u] = ns[u];
                    nx[u] = 0.; nh[u] = 1.; ns[u] = 0.;
                    if (in < next) { nx[u] = phys[FAR_REC(in)].xnu; nh[u] = phys[FAR_REC(in)].hw; ns[u] = phys[FAR_REC(in)].stild; }
                }
                FarSlot sl[4];
                double dm = __builtin_inf();
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // amplitude and pedestal as line_records forms them (no Y factors: uncoupled molecules only)
                    const double A2 = (st[u] * hw[u]) * (1.0 / K_PI), HW2 = hw[u] * hw[u];
                    const double p = (A2 * frcp_any(625. + HW2)) * wsc, a2 = A2 * wsc;
                    const double d1 = xnu[u] - c0, d2 = -(xnu[u] + c0);
                    const double amp = on[u] ? -(a2 * rinv) : 0.0, hq = on[u] ? HW2 : 0.;
                    // (two resonances: slots 0, 1 = the first line's poles, 2, 3 = the second's; one: slots 0, 1 = the two lines)
                    if (two) {
                        sl[2 * u] = far_slot_of(on[u] ? d1 : 2. * iv.rho, hq, rinv, amp);
                        sl[2 * u + 1] = far_slot_of(on[u] ? d2 : -2. * iv.rho, hq, rinv, amp);
                    } else {
                        sl[u] = far_slot_of(on[u] ? d1 : 2. * iv.rho, hq, rinv, amp);
                        sl[2 + u] = FarSlot{0., 0., 0., 0.};
                    }
                    if (on[u]) {
                        dm = fmin(dm, two ? fmin(fabs(d1), fabs(d2)) : fabs(d1));
                        if (co2) {   // -pa (2 - (t - d1)^2 / 625) in powers of t = WN - c0 (modm.f90:808-817)
                            q0 -= p * (2. - d1 * d1 * (1. / 625.));
                            q1 -= p * (2. * d1 * (1. / 625.));
                            q2 += p * (1. / 625.);
                        } else if (!o2) ped += two ? p + p : p;
                    }
                }
                const int ord = __builtin_amdgcn_readfirstlane(far_order(wave_min(dm), rinv, P));
                far_accumulate<P>(c, hi_addr, ord, __builtin_amdgcn_readfirstlane((int)two), sl[0], sl[1], sl[2], sl[3]);
            }
        }
        pos = next;
    }
    // the parent's series, re-expanded about this interval's centre: lane n < P evaluates it at the n-th Chebyshev node of this
    // interval and adds f T_k(x_n) 2 / P to its sums - after the sum over the lanes that is the discrete Chebyshev transform,
    // exact for the polynomial of degree P - 1 that the parent's series is
    // (the sums kept in LDS come back into registers first: the slots of the step loop are dead by now)
    double ch[PL];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < PL; n++) ch[n] = sHi[wave][n][lane];
    double pped = 0.;
#ifdef FAR_ABL_TRANS   // timing experiments (wrong results): without the parent's series / without the far lines
    if (false) {
#else
    if (has_parent && wave == 0) {
#endif
        const double *pm = a.farmom + ((pl * (size_t)a.far_ni + gip) * nmol + m) * FAR_MOM_STRIDE;
        if (pm[P + 1] != 0.) {
            any = true;
            pped = pm[P];
            const FarIv ip = far_interval(a, level + 1, j >> 1);
            const double xn = place.node[lane];   // cos(pi (lane + 1/2) / P)
            // (a parent of ONE wavenumber - the stub of a last tile - holds a constant: x = 0 there, not 0 * inf)
            const double xp = (ip.rho > 0.) ? ((c0 + iv.rho * xn) - ip.c) * frcp_any(ip.rho) : 0., x2 = xp + xp;
            double b1 = 0., b2 = 0.;
#pragma unroll
            for (int n = P - 1; n >= 1; n--) {   // (unrolled: the parent's sums arrive as a few wide scalar loads)
                const double t = fma(x2, b1, pm[n] - b2);
                b2 = b1;
                b1 = t;
            }
            double f = fma(xp, b1, 0.5 * pm[0] - b2);
            f = (lane < P) ? f * (2.0 / (double)P) : 0.;
            double t0 = 1., t1 = xn;
            const double xn2 = xn + xn;
            c[0] += f;
            c[1] = fma(f, xn, c[1]);
#pragma unroll
            for (int k = 2; k < P; k++) {
                const double tk = fma(xn2, t1, -t0);
                if (k < FAR_PREG) c[k < FAR_PREG ? k : 0] = fma(f, tk, c[k < FAR_PREG ? k : 0]);
                else ch[k >= FAR_PREG ? k - FAR_PREG : 0] = fma(f, tk, ch[k >= FAR_PREG ? k - FAR_PREG : 0]);
                t0 = t1;
                t1 = tk;
            }
        }
    }
    double *mo = a.farmom + ((pl * (size_t)a.far_ni + gi) * nmol + m) * FAR_MOM_STRIDE;
    // (`any` of wave 0 knows the parent; the runs are the same for every wave)
    if constexpr (NWF > 1) {
        if (threadIdx.x == 0) sPart[0][P + 3] = any ? 1. : 0.;
        __syncthreads();
        any = sPart[0][P + 3] != 0.;
        __syncthreads();
    }
    if (!any) {   // (the flag alone: lines_kernel does not read the sums then)
        if (threadIdx.x == 0) { mo[P] = 0.; mo[P + 1] = 0.; }
        return;
    }
    const double pedw = wave_sum(ped) + pped;
    const double s0 = co2 ? wave_sum(q0) : 0., s1 = co2 ? wave_sum(q1) : 0., s2 = co2 ? wave_sum(q2) : 0.;
    double tot[P / 4];   // row r of group G: the wave's total of sum 4 G + r
#pragma unroll
    for (int G = 0; G < P / 4; G++) {
        constexpr int GR = FAR_PREG / 4;
        if (G < GR) tot[G] = row_sum16(swap_add16(swap_add32(c[G < GR ? 4 * G : 0], c[G < GR ? 4 * G + 2 : 0]), swap_add32(c[G < GR ? 4 * G + 1 : 0], c[G < GR ? 4 * G + 3 : 0])));
        else {
            const int h = G >= GR ? 4 * (G - GR) : 0;
            tot[G] = row_sum16(swap_add16(swap_add32(ch[h], ch[h + 2]), swap_add32(ch[h + 1], ch[h + 3])));
        }
    }
    if constexpr (NWF > 1) {
        if ((lane & 15) == 0) {
#pragma unroll
            for (int G = 0; G < P / 4; G++) sPart[wave][4 * G + (lane >> 4)] = tot[G];
        }
        if (lane == 0) { sPart[wave][P] = pedw; sPart[wave][P + 1] = s0; sPart[wave][P + 2] = s1; sPart[wave][P + 3] = s2; }
        __syncthreads();
        if (wave != 0) return;
    }
    // wave 0: lane n < P + 4 adds the waves' values of entry n in wave order
    double v = 0.;
    if constexpr (NWF > 1) {
        if (lane < P + 4)
            for (int w = 0; w < NWF; w++) v += sPart[w][lane];
    }
    const double pedt = NWF > 1 ? __shfl(v, P) : pedw;
    // CO2: c0 + c1 t + c2 t^2 with t = rho x, t^2 = rho^2 (T_2 + 1) / 2 -> T_0: c0 + c2 rho^2 / 2 (stored twice), T_1: c1 rho, T_2: c2 rho^2 / 2
    double adj = 0.;
    if (co2) {
        const double S0 = NWF > 1 ? __shfl(v, P + 1) : s0, S1 = NWF > 1 ? __shfl(v, P + 2) : s1, S2 = (NWF > 1 ? __shfl(v, P + 3) : s2) * (0.5 * iv.rho * iv.rho);
        if constexpr (NWF > 1) adj = (lane == 0) ? 2.0 * (S0 + S2) : ((lane == 1) ? S1 * iv.rho : ((lane == 2) ? S2 : 0.));
        else adj = (lane < 16) ? 2.0 * (S0 + S2) : ((lane < 32) ? S1 * iv.rho : ((lane < 48) ? S2 : 0.));
    }
    if constexpr (NWF > 1) {
        if (lane < P) mo[lane] = v + adj;
    } else {
#pragma unroll
        for (int G = 0; G < P / 4; G++)
            if ((lane & 15) == 0) mo[4 * G + (lane >> 4)] = tot[G] + (G == 0 ? adj : 0.);
    }
    if (lane == 0) { mo[P] = pedt; mo[P + 1] = 1.; }
}

}  // namespace

namespace monortm_dev {
void launch_far_plan(const ModmArgs &a, const DevLines &L, const DevTables &tb, hipStream_t s) {
    const dim3 pgrid((a.far_ni * a.nmol + 63) / 64, a.nprof, a.nlay_max);
    if (a.real_kind == 4) hipLaunchKernelGGL(far_plan_kernel<float>, pgrid, dim3(64), 0, s, a, L, tb);
    else hipLaunchKernelGGL(far_plan_kernel<double>, pgrid, dim3(64), 0, s, a, L, tb);
}
// rho_tile: half-width of a tile, lines_per_cm: lines of the table per cm-1 (both estimates of the host: they size the workgroups)
void launch_far(const ModmArgs &a, const DevLines &L, const DevTables &tb, double rho_tile, double lines_per_cm, hipStream_t s) {
    static const int nwf_env = getenv("MONORTM_FAR_WAVES") ? atoi(getenv("MONORTM_FAR_WAVES")) : 0;   // A/B switch for measurements
    for (int l = a.far_levels - 1; l >= 0; l--) {
        const int nint = far_level_count(a.far_ntile, l);
        int most = 0;
        FarPlace place{};
        for (int n = 0; n < 64; n++) place.node[n] = std::cos(3.14159265358979323846 * ((double)n + 0.5) / (double)FAR_P);
        for (int k = 0; k < 8; k++) {
            int items = 0;
            for (int q = 0; q < a.nmol; q++) {
                int xlo, nx;
                far_xcd_share(L.mol_start, a.nmol, q, &xlo, &nx);
                place.xlo[q] = (unsigned char)xlo;
                place.nx[q] = (unsigned char)nx;
                // (16 bits: api.hip gives a grid of more than 65535 tiles no far levels at all - an interval count that wrapped here would
                // leave intervals without a workgroup whose far lines far_plan_kernel has already taken off the tiles' runs)
                if (far_xcd_items(nint, k, xlo, nx) > 65535) { fprintf(stderr, "monortm_amd: far_kernel placement overflow (%d intervals)\n", nint); abort(); }
                place.cnt[q][k] = (unsigned short)far_xcd_items(nint, k, xlo, nx);
                items += place.cnt[q][k];
            }
            most = std::max(most, items);
        }
        if (most == 0) continue;   // (no molecule has lines: the plan holds no far line either)
        const dim3 grid(8 * most, a.nprof, a.nlay_max);
        // waves per workgroup by the lines an (interval, molecule) expands - about 4.4 half-widths of them, the top level everything
        // out to the 25 cm-1 rule - at ~16 steps of 128 lines a wave: fewer, longer waves run better than many short ones as long
        // as no single wave becomes the critical path (configs[2], half-width 1.28: one wave 0.26 ms, two 0.28, four 0.43; the top
        // level at 2.56: four 0.44, eight 0.50, one 0.51)
        int nactive = 0;
        for (int q = 0; q < a.nmol; q++) nactive += L.mol_start[q + 2] > L.mol_start[q + 1];
        const double rho_l = rho_tile * (double)(1 << l);
        const double width = (l == a.far_levels - 1) ? std::max(50.0 - 2.4 * rho_l, 4.4 * rho_l) : 4.4 * rho_l;
        const double steps = lines_per_cm * width / std::max(nactive, 1) / 128.0;
        const int nwf = nwf_env ? nwf_env : (steps <= 16. ? 1 : (steps <= 32. ? 2 : (steps <= 128. ? 4 : 8)));
#define FAR_LAUNCH(N)                                                                                              \
    do {                                                                                                           \
        if (a.real_kind == 4) hipLaunchKernelGGL((far_kernel<float, N>), grid, dim3(64 * N), 0, s, a, L, l, place);        \
        else hipLaunchKernelGGL((far_kernel<double, N>), grid, dim3(64 * N), 0, s, a, L, l, place);                        \
    } while (0)
        if (nwf <= 1) FAR_LAUNCH(1);
        else if (nwf == 2) FAR_LAUNCH(2);
        else if (nwf <= 4) FAR_LAUNCH(4);
        else FAR_LAUNCH(8);
#undef FAR_LAUNCH
    }
}
}  // namespace monortm_dev
