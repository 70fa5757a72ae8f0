// device_common.hpp - shared declarations of the MI355X (gfx950 / CDNA4) implementation:
// monortm_amd/csrc - MI355X (gfx950 / CDNA4) implementation of monoRTM's optical-depth and
// radiative-transfer hot path behind the C ABI of include/monortm_hip.h.
//
// Reference behaviour being replaced (paths relative to /root/reference):
//   MODM      src/modm.f90:21-274      LINES  src/modm.f90:277-440
//   line shapes src/modm.f90:567-831, :888-895, :965-1251
//   CONTNM    src/contnm.f90:25-1142 (+ accessors)   XINT/RADFN src/lblrtm_sub.f90
//   TIPS_2003 src/tips_2003.f90:2-298, :4610         ODCLW_TKC src/CloudOptProp.f90:29-157
//   CALCTMR / RTM / RAD_UP_DN  src/RTMmono.f90
//
// Design (DESIGN.md has the full account):
//   * one process = one GPU; a context owns the device line table (44 B per line, SoA);
//   * lines_kernel: workgroup = (profile, layer, tile of wavenumbers, slice of the line list), lane = 1 or 2 wavenumbers.
//     Everything of a line that does not depend on the wavenumber (shifted centre, S~, Lorentz and
//     Doppler widths, coupling factors, pedestal) is prepared ONCE per (layer, line) by one lane,
//     staged in LDS, and then broadcast-read by every wave: the inner loops are one FP64 reciprocal
//     and 7-18 FP64 instructions per (wavenumber, layer, line), chosen per sub-run of lines of one class.
//     The reference recomputes all of it per wavenumber (6 exp, 2 pow, 3 sqrt per evaluation).  On dense
//     grids distant lines enter through per-molecule far-field moments of the tile instead.
//   * finish_kernel: workgroup = (profile, layer); MT_CKD continuum on the 1 cm-1 ABSRB grid in LDS,
//     second interpolation to the wavenumbers, TKC cloud liquid, line-slice sums, totals.
//   * rtm_kernel: workgroup = 64 wavenumbers x G layer groups of a profile; CALCTMR + RAD_UP_DN + RTM recurrences in
//     registers, group sums combined through LDS in the reference's order.
// No MFMA (nothing here is a dense contraction), no Triton, no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>

// Timing-experiment switches change RESULTS (stages left out, guards removed, series cut short): a build that defines one of
// them must say that it is an experiment (tools/build_variant.sh NAME -DMONORTM_EXPERIMENT=1 -D...), so that none can slip into
// the shipped library through an environment's compiler flags.  __graft_entry__.build() / _build.py define none.
#if (defined(FAR_ABL_HOT) || defined(FAR_ABL_STEPS) || defined(FAR_ABL_TRANS) || defined(MONORTM_ABLATE_LOOP) || defined(MONORTM_ABLATE_EVAL) || \
     defined(MONORTM_ABLATE_VOIGT) || defined(MONORTM_ABLATE_ZSEARCH) || defined(MONORTM_ABLATE_EXP4) || defined(MONORTM_ABLATE_COUPLE) ||     \
     defined(MONORTM_NO_CLAMP_GUARD) || defined(MONORTM_NO_UNIFIED) || defined(MONORTM_NO_VSCAN) ||              \
     defined(MONORTM_NO_HALF) || defined(MONORTM_NO_FULL) || defined(MONORTM_NO_CHUNKFAR) || defined(MONORTM_NO_SGL_TSKIP) || defined(LINES_TIMING) || \
     defined(LINES_CLASS_STATS) || defined(MW_TIMING)) &&                                                                                          \
    !defined(MONORTM_EXPERIMENT)
#error "a timing-experiment / A-B switch of monortm_amd is defined: such builds are measurements, not the product - add -DMONORTM_EXPERIMENT=1"
#endif

#include <cstdint>

#include "../../include/monortm_hip.h"

namespace monortm_dev {

// ------------------------------------------------------------------------------------------------
// constants: the literal decimal strings of the reference as doubles (src/PhysConstants.f90:19-39)
// ------------------------------------------------------------------------------------------------
#define K_PI 3.1415926535898
#define K_PLANCK 6.62606876E-27
#define K_BOLTZ 1.3806503E-16
#define K_CLIGHT 2.99792458E+10
#define K_AVOGAD 6.02214199E+23
#define K_RADCN1 1.191042722E-12
#define K_RADCN2 1.4387752
#define K_ONEPL 1.001
#define K_ONEMI 0.999
#define K_T0 296.0
#define K_P0 1013.25

constexpr int MXMOL = 39;
constexpr int MXBRD = 7;
constexpr int NSCOR = MXMOL * 9;

#ifndef MONORTM_FAR_P
#define MONORTM_FAR_P 56
#endif
constexpr int FAR_P = MONORTM_FAR_P;   // Chebyshev sums of a far field (lines_device.hpp: "Far field of a tile")
// ---- the far field of dense grids formed OUTSIDE lines_kernel (far_kernel.hip, round 5) -------------------------------------------
// Intervals of wavenumbers in up to FAR_MAXLEV levels: level 0 = the tiles of lines_kernel (tw wavenumbers each), level l = groups
// of 2^l consecutive tiles.  gi = far_level_offset(l) + j numbers them; the parent of (l, j) is (l + 1, j / 2).
constexpr int FAR_MAXLEV = 6;
constexpr int FAR_GEOM_INTS = 8;            // per (profile, layer, interval, molecule): lowS, lowE, highS, highE, e0, e1s, e1e, unused
constexpr int FAR_SEGS = 5;                 // per (profile, layer, tile, molecule): up to 5 runs of table lines that the tile walks itself
constexpr int FAR_SEG_INTS = 2 * FAR_SEGS;  // (base_k = first line - lines before the run, cum_k = lines up to and including the run)
constexpr int FAR_MOM_STRIDE = FAR_P + 2;   // Chebyshev sums (Clenshaw's convention), sum of the constant pedestals, "anything there" flag
__host__ __device__ inline int far_level_count(int ntile, int l) { return (ntile + (1 << l) - 1) >> l; }
__host__ __device__ inline int far_level_offset(int ntile, int l) {
    int o = 0;
    for (int k = 0; k < l; k++) o += far_level_count(ntile, k);
    return o;
}

// Placement of far_kernel's workgroups (one per interval and molecule): workgroups are dealt round-robin over the 8 XCDs, so
// workgroup x runs on XCD x mod 8.  A molecule gets a share of the XCDs in proportion to its lines (table order), and its
// intervals go round its share: the intervals of a molecule re-read the same records from ONE L2 (or a few), and a molecule
// with a third of the lines does not hold one XCD three times as long as the others (measured: 0.64 ms per level with
// molecule = x mod 8 on a list where one of five molecules has 37 % of the lines).
// mol_start: DevLines::mol_start.  far_xcd_share: XCDs [xlo, xlo + nx) of molecule m (0-based); nx = 0: no lines.
__host__ __device__ inline void far_xcd_share(const int *mol_start, int nmol, int m, int *xlo, int *nx) {
    const long long base = mol_start[1], N = (long long)mol_start[nmol + 1] - base;
    const long long s0 = mol_start[m + 1] - base, s1 = mol_start[m + 2] - base;
    if (N <= 0 || s1 <= s0) { *xlo = 0; *nx = 0; return; }
    const int lo = (int)((16 * s0 + N) / (2 * N)) < 7 ? (int)((16 * s0 + N) / (2 * N)) : 7;
    int hi = (int)((16 * s1 + N) / (2 * N));
    if (hi > 8) hi = 8;
    if (hi < lo + 1) hi = lo + 1;
    *xlo = lo;
    *nx = hi - lo;
}
// the same as a table for one level, formed on the host and passed by value (a workgroup finds its item with scalar compares;
// the divisions of far_xcd_share per molecule and workgroup were a third of a tile-level wave's instructions)
struct FarPlace {
    unsigned char xlo[MXMOL], nx[MXMOL];
    unsigned short cnt[MXMOL][8];   // workgroups of molecule m on XCD k
    double node[64];                // Chebyshev nodes cos(pi (n + 1/2) / FAR_P) of the re-expansion (formed by the host: a cos() per lane
                                    // and workgroup was a tenth of a tile-level workgroup's instructions)
};
// workgroups of molecule m on XCD k at a level of nint intervals: intervals (k - xlo), (k - xlo) + nx, ...
__host__ __device__ inline int far_xcd_items(int nint, int k, int xlo, int nx) {
    if (nx <= 0 || k < xlo || k >= xlo + nx || k - xlo >= nint) return 0;
    return (nint - (k - xlo) + nx - 1) / nx;
}

enum : int { ERRBIT_TEMP = 1, ERRBIT_SDV = 2, ERRBIT_ARG = 4 };  // ARG: nlay[p] outside 1..nlay_max or wn not ascending (device arrays)

struct DevTables {  // device copies of monortm_tables.h
    const double *self296, *self260, *frgn296, *fco2, *n2c296, *n2sf296, *n2c220, *n2sf220, *xfac_rhu, *xfacco2,
        *tdep_bandhead, *tips_qoft, *tips_q296, *smass;
    // branches above 1340 cm-1
    const double *o3ch_x, *o3ch_y, *o3ch_z, *o3hh0, *o3hh1, *o3hh2, *o3huv, *o2f_x, *o2f_t, *o2inf1, *o2inf3, *o2vis, *o2fuv,
        *n2f_272, *n2f_228, *n2f_ah2o, *n2f1;
    const int *tips_isonm, *tips_offset;
    // log(T-low table / 296 K table) of the three temperature interpolations exp(tfac * log(ratio)) below 820 cm-1 (H2O self,
    // N2 rototranslational and its scale factor): formed once per context by logratio_kernel with the device's own log(),
    // so finish_mw_kernel gets the bits finish_kernel computes in place
    const double *lr_self, *lr_n2c, *lr_n2sf;
};

struct DevLines {
    const double *vnu, *s0adj, *lc;
    const float *alfa, *hwhm, *epp, *tmpalf, *pshift, *sdep, *brd_dat;
    const uint32_t *meta;
    const int32_t *brd_flg;
    int mol_start[MXMOL + 2];
    unsigned long long sorted_mask;
    unsigned long long lc_mask;  // molecules that own at least one line-coupled entry
    double max_abs_shift;
};

struct ModmArgs {
    int nprof, nwn, nlay_max, nmol, ibrd;
    double dvset, sclcpl, sclhw, y0res;
    double cntnm[7];
    // "real" arrays are REAL(real_kind): double (8) or float (4); wavenumbers are always double
    // (the reference keeps WN REAL*8 in both builds, src/modm.f90:139)
    int real_kind;
    const double *wn;
    const void *P, *T, *CLW, *WKL, *WBRODL;
    const int *nlay;
    void *O, *O_BY_MOL, *OC, *O_CLW;
    int *errflag;
    // line slicing (few workgroups otherwise): nslice blocks share one (profile, layer, tile); each writes its
    // partial sums to partial[slice][profile][layer][mol][wn], finish_kernel adds them in slice order
    int nslice;
    void *partial;
    int slices_reduced;  // the slice sums were formed by reduce_slices_kernel (wide grids), not inside finish_kernel
    // nslice == 1: lines_kernel leaves sum_mol O_BY_MOL (as stored, added in molecule order) per (profile, layer, wn) here, so
    // the finish kernel of the microwave range reads nwn values per layer instead of nmol x nwn; null otherwise
    double *osum;
    int fair;       // lines_kernel, one-wave workgroups: waves lower their issue priority as they progress (grids of a few rounds)
    // dense grids: LinePhysM (32 B; + LinePhysY, 16 B, for coupled lines) of every table line for every (profile, layer), formed by physics_kernel before lines_kernel
    // (phys_lines = lines of the table); null: lines_kernel forms them in place, per tile
    void *phys;
    int phys_lines;
    // cross-section molecules (IXSECT = 1): column amounts [profile][layer][nxs] in, their total optical depth
    // [profile][layer][wn] out (xsec_kernel), added into O by the finish kernels; both null otherwise
    const void *XAMNT;
    void *ODXSEC;
    int nxs;
    // dense grids with the physics pass: the far field of every tile formed by far_kernel (far_kernel.hip) before lines_kernel.
    // farmom [profile][layer][interval][molecule][FAR_MOM_STRIDE] Chebyshev sums, fargeom [..][interval][molecule][FAR_GEOM_INTS]
    // the far lines of every interval in index space, farseg [profile][layer][tile][molecule][FAR_SEG_INTS] the runs of table
    // lines that lines_kernel still walks.  All null: lines_kernel forms the far field of a tile itself, chunk by chunk.
    double *farmom;
    int *fargeom, *farseg;
    int far_levels, far_ni, far_tw, far_ntile;   // levels in use, intervals of all levels, wavenumbers per tile, tiles
};

// device copy of the cross-section tables (monortm_hip_xsec_tables): per (molecule, spectral region) a row of
// reg = (molecule, V1FX, V2FX of FSCDXS, points, temperatures, XDOPLR, V1, V2 of the last file header), its temperatures (ascending) and measurement pressures [mbar], and
// the offsets of its spectra in pool
struct DevXsec {
    const double *reg, *temps, *pres, *pool;
    const long long *offs;
    int nxs, nreg;
};

struct RtmArgs {
    int nprof, nwn, nlay_max, iout;
    int real_kind;
    const double *wn;
    const void *T, *TZ, *O, *emiss, *reflc;
    const int *nlay, *irt;
    void *tmpsfc, *RUP, *RDN, *TRTOT, *RAD, *TB, *TMR;
};

// exp() and 1/x for the small kernels behind the line sum (radiance recurrences, microwave continuum): Cody-Waite reduction +
// degree-13 polynomial (the routine of the line kernel's prepare stage: 20 instructions, 1-2 ulp like the library call at ~35)
// and v_rcp_f64 + two Newton steps (1 ulp; an IEEE division is ~11 instructions).  exp_cw: |n| < 2^31 for every argument that
// occurs (optical depths, hc v / kT); ldexp saturates to 0 / +inf beyond the exponent range; fmax keeps the conversion defined
// for a NaN argument - the polynomial is NaN then, and so is the result.
__device__ __forceinline__ double exp_cw(double x) {
    const double n = rint(x * 1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)fmax(n, -2200.));
}
__device__ __forceinline__ double rcp2(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// Planck function bb_fn (RTMmono.f90:223-237) with v^3 RADCN1 = c3 formed once per wavenumber
__device__ __forceinline__ double planck(double c3, double v, double fbeta) {
    const double e = exp_cw(v * fbeta) - 1.;
    return (e == __builtin_inf()) ? 0. : c3 * rcp2(e);
}
// Doppler half width per unit wavenumber, HALFWHM_D / XNU (modm.f90:442-454) for an isotopologue of mass M [g/mol]
__device__ __forceinline__ double doppler_factor(double M, double T) {
    return sqrt(2. * log(2.) * ((K_BOLTZ * T) / (M / K_AVOGAD))) / K_CLIGHT;
}

template <typename R>
__host__ __device__ inline const R *rp(const void *p) { return static_cast<const R *>(p); }
template <typename R>
__host__ __device__ inline R *wp(void *p) { return static_cast<R *>(p); }

// ---- lines_ms_kernel (round 6): G states per one-wave workgroup, MS_WPS wavenumbers per lane (lines_ms_kernel.hip) -------------------
#define MS_WPS 5
#define MS_MAXSTEPS 4
struct MsArgs {
    int G, LPS, CL, nsteps;   // states per wave, lanes per state, lines per chunk, prepare passes of 64 items (ceil(G CL / 64))
    int sa_stride;            // HotA records per state in LDS (CL + 2: the class loops read two records ahead)
    int npg;                  // groups of G profiles (ceil(nprof / G)); grid = npg x nlay_max
    int nslot;                // (molecule, isotopologue) pairs of the line table: slot_base[m] + iso - 1
    int inv_cl;               // ceil(65536 / CL): item / CL = (item * inv_cl) >> 16 for item < 256
    int inv_lps;              // ceil(65536 / LPS): lane / LPS likewise
    const int *slot_base;     // [nmol + 1] on the device
    void *scratch;            // per workgroup G x CL x (HotB + ColdLine): the records of the rare shapes of a chunk
    unsigned short *reach;    // [lines of the table] slots of channels each line / its negative resonance can reach (ms_reach_kernel, once per launch)
    float *near0;             // [lines of the table] distance of the table centre to the nearest channel, rounded down (ms_reach_kernel)
    int ablate;               // MONORTM_EXPERIMENT builds only (option ms_ablate; wrong results): 1 = prologue only, 2 = no evaluate stage, 3-5 parts of it
};
size_t lines_ms_lds(const MsArgs &ms, int nmol);
size_t lines_ms_scratch(const MsArgs &ms, long long nwg);
void launch_lines_ms(const ModmArgs &a, const DevLines &L, const DevTables &tb, const MsArgs &ms, bool ibrd, hipStream_t s);

// ---- launchers: one translation unit per kernel family -------------------------------------------------------
// lines_kernel.hip: block = nw waves, lane = wpl wavenumbers (tile = wpl * nw * 64), as chosen by lines_config();
// ibrd selects the species-broadening instantiation
void lines_config(int nwn, int real_kind, long long states, double span, int *nw, int *wpl);  // span = wn[nwn-1] - wn[0]
void launch_physics(const ModmArgs &a, const DevLines &L, const DevTables &tb, int nlines, bool ibrd, hipStream_t s);
// far_kernel.hip: far_plan_kernel (which lines are far for which interval; independent of physics_kernel), then far_kernel per level,
// top level first (reads the plan and physics_kernel's records)
void launch_far_plan(const ModmArgs &a, const DevLines &L, const DevTables &tb, hipStream_t s);
void launch_far(const ModmArgs &a, const DevLines &L, const DevTables &tb, double rho_tile, double lines_per_cm, hipStream_t s);
void launch_lines(const ModmArgs &a, const DevLines &L, const DevTables &tb, int nw, int wpl, bool ibrd, dim3 grid, size_t dyn_lds,
                  hipStream_t s);
// continuum_kernel.hip: high = spectral range reaches above 1340 cm-1; par = passes side by side in the waves of a
// 256-thread workgroup (small grids below 1340 cm-1; lds holds 4 sets of grids); threads = 64 with lds_sets = 4: four layers
// per one-wave workgroup (large microwave batches)
hipError_t launch_finish(const ModmArgs &a, const DevTables &tb, double V1ABS, double V2ABS, int NPTABS, int csize, bool high,
                         bool par, int threads, size_t lds, int lds_sets, hipStream_t s);
void launch_reduce_slices(const ModmArgs &a, hipStream_t s);
// spectral ranges that end below 820 cm-1: the continuum passes side by side in three stages (finish_mw_kernel)
void launch_logratio(const double *t296, const double *tlow, double *out, int n, hipStream_t s);
// per context: the spectral-range constants of finish_mw_kernel's stages A and B (continuum_kernel.hip: MwItemA / MwItemB)
struct MwCache {
    // one entry per spectral range seen by the context (a caller that alternates between two channel sets keeps both; round 4
    // held a single slot and re-built it - with a device-wide synchronisation - at every change).  The items of an entry are
    // built by a kernel on the stream of the call that first needs them; later calls on the same stream are ordered behind it
    // by the stream, calls on another stream wait for `built` once.
    struct Entry {
        double key[5];       // V1, V2, V1ABS, V2ABS, NPTABS
        void *items;
        hipEvent_t built;
        hipStream_t stream;  // the stream that built the items
        bool ready;          // `built` was seen complete: no stream has to wait any more
        unsigned long long last_use;
    };
    static constexpr int kMax = 8;
    Entry e[kMax];
    int n = 0;
    unsigned long long clock = 0;
    const char *why = nullptr;  // reason of the last failure, where hipGetErrorString would not say it
    void release();             // frees every entry (context teardown; the device is idle)
};
hipError_t launch_finish_mw(const ModmArgs &a, const DevTables &tb, double V1, double V2, double V1ABS, double V2ABS, int NPTABS, MwCache &cache,
                            hipStream_t s);
// known-answer hook: device versions of W4, SD_Humlicek, SDVOIGT, RADFN, AtoB, ODCLW_TKC, TIPS scor (continuum_kernel.hip)
void launch_kat(int which, int n, const double *in, const double *tab, double *out, int *errflag, const DevTables &tb, hipStream_t s);
// xsec_kernel.hip: MONORTM_XSEC_SUB + convolve (src/monortm_sub.F90:1540-1834) -> a.ODXSEC
void launch_xsec(const ModmArgs &a, const DevXsec &x, hipStream_t s);
// rtm_kernel.hip
void launch_rtm(const RtmArgs &a, hipStream_t s);

}  // namespace monortm_dev
