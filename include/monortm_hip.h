/* monortm_hip.h - C ABI of the MI355X-native MODM / CALCTMR / RTM hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its boundary is the Fortran
 * module-procedure interface used by PROGRAM MONORTM
 *     CALL MODM(...)     reference src/monortm.f90:557-561  ->  src/modm.f90:21-25
 *     CALL CALCTMR(...)  reference src/monortm.f90:567      ->  src/RTMmono.f90:239
 *     CALL RTM(...)      reference src/monortm.f90:573-574  ->  src/RTMmono.f90:13-14
 * and module procedures are compiler-mangled with compiler-specific array descriptors, so the
 * replacement is source level: monortm_amd/fortran/{modm_hip,rtmmono_hip}.f90 define modules
 * ModmMod / RTMmono with the reference's public names and argument lists and forward to the
 * entry points below through ISO_C_BINDING (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types, caller-owned contiguous buffers, no library types in any signature;
 *   - every function returns 0 on success, a MONORTM_E* code otherwise; the text of the last
 *     error is available from monortm_hip_last_error() (the reference has no status codes: every
 *     failure is a Fortran STOP; the Fortran shim turns a non-zero status into STOP);
 *   - arrays are C-ordered with the WAVENUMBER AXIS FASTEST, i.e. element (wn m, layer k) of the
 *     reference's O(m,k) is O[k*nwn + m]; a batch adds a leading profile axis;
 *   - monortm_real arrays hold the caller's default REAL: real_kind = 8 (IEEE double, the reference's "dbl" build:
 *     default REAL = 8 bytes, build/makefile.common:195-198) or real_kind = 4 (float, the "sgl" build).  The kind is
 *     fixed per context at init; wavenumbers are REAL*8 in both builds (src/modm.f90:139) and so are the scalars
 *     passed by value here.  With real_kind = 4 the Lorentz line sum is evaluated in float (one v_rcp_f32 per
 *     line), the per-(layer, line) preparation, the coupled / Voigt shapes, the continuum and the radiance
 *     recurrences in double with float loads / stores;
 *   - one context = one loaded TAPE3 on one GPU (monortm_hip_init) or on several (monortm_hip_init_multi); calls on a
 *     context are serialised by the caller
 *     (the reference's MODM is non-reentrant: SAVE / COMMON state, src/modm.f90:161-163).
 *   - the *_dev entry points take DEVICE pointers and a hipStream_t (as void*): inputs stay resident
 *     in HBM, nothing is copied, the call is asynchronous on that stream.
 */
#ifndef MONORTM_HIP_H
#define MONORTM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MONORTM_OK = 0,
    MONORTM_EIO = 1,          /* TAPE3 missing / unreadable      (reference: lnfl_mod.f90:131-132 STOP) */
    MONORTM_EFORMAT = 2,      /* TAPE3 malformed / no isotope tag (reference: lnfl_mod.f90:297-302 STOP) */
    MONORTM_EUNSUPPORTED = 3, /* option outside the entry point called (IXSECT=1 through monortm_hip_modm: use _modm_xs; real_kind not 4 / 8) */
    MONORTM_ETEMP = 4,        /* layer temperature outside 70-3000 K (reference: tips_2003.f90:277 STOP) */
    MONORTM_ESDV = 5,         /* speed-dependent Voigt gave Re(v)<0 (reference: modm.f90:1062 STOP) */
    MONORTM_EARG = 6,         /* bad argument */
    MONORTM_EHIP = 7          /* HIP runtime error */
};

typedef void monortm_real; /* element type selected by real_kind: double or float */

#define MONORTM_NCONT 5 /* continuum slots returned in OC: molecules 1,2,3,7,22 (index_cont, modm.f90:166) */

/* Replaces the once-per-process GET_LNFL(IPR,ICP,HFILE,v1,v2) inside MODM (src/modm.f90:187-190,
 * src/lnfl_mod.f90:22-133): parses TAPE3 on the host with the reference's block skip / stop rules for
 * [v1-25, v2+25], builds the device line table.  device = HIP device ordinal (or -1: current). */
int monortm_hip_init(const char *tape3_path, double v1, double v2, int icp, int real_kind, int device, void **ctx);

/* The same for ngpu devices of this node (ngpu <= 0: every visible device): one line table, stream and set of staging
 * arenas per device.  monortm_hip_modm / monortm_hip_rtm on such a context shard the batch into contiguous blocks of
 * ceil(nprof / ngpu) profiles - the reference's only parallel axis, the independent-profile loop of src/monortm.f90:357 -
 * enqueue every device's block before waiting for the first, and each block comes back over its own device's PCIe link
 * straight into the caller's arrays (results are bit-identical to a one-device context: profiles do not interact).
 * The *_dev, profiling and check entry points need a one-device context.  MONORTM_DEVICES="0,1,.." overrides the device
 * list (an ordinal may repeat, which exercises the sharding on a one-GPU box). */
int monortm_hip_init_multi(const char *tape3_path, double v1, double v2, int icp, int real_kind, int ngpu, void **ctx);

int monortm_hip_device_count(void *ctx); /* devices (shards) behind a context: 1 for monortm_hip_init */

/* One process per GPU: the single gather of a profile-sharded job over RCCL / xGMI.
 * Reference axis: the independent-profile loop of src/monortm.f90:357 (the reference itself is serial and has no such step);
 * every rank runs monortm_hip_modm_dev / _rtm_dev on its contiguous block of ceil(P / G) profiles, then ONE gather of the
 * device-resident outputs to `root` - the call the Python layer makes through torch.distributed (monortm_amd/distributed.py),
 * here for C / Fortran callers.  RCCL is loaded on first use (dlopen), never by single-GPU callers.
 *   monortm_hip_comm_unique_id : rank 0 fills the 128-byte id (ncclGetUniqueId); the caller hands it to every rank (MPI, a file)
 *   monortm_hip_comm_init      : ncclCommInitRank on the context's device; collective over the `world` ranks
 *   monortm_hip_gather_dev     : `bytes` bytes from every rank's `send` (device) into recv[rank * bytes ..] on `root` (device;
 *                                may be NULL elsewhere), asynchronous on `stream` (ncclGather)
 * Errors: MONORTM_EHIP with the RCCL message in monortm_hip_last_error. */
int monortm_hip_comm_unique_id(void *id128);
int monortm_hip_comm_init(void *ctx, int world, int rank, const void *id128);
int monortm_hip_gather_dev(void *ctx, const void *send, size_t bytes, void *recv, int root, void *stream);

void monortm_hip_finalize(void *ctx);

const char *monortm_hip_last_error(void *ctx); /* ctx may be NULL: error of the last failed init */

/* Host-only (no GPU needed): parse TAPE3 exactly as monortm_hip_init would and report, per molecule m = 1..39
 * (index 0 = all): physical line records kept (IFLG >= 0), table entries (records the LINES walk treats as a
 * line) and entries that carry line-coupling coefficients.  Each array has 40 elements. */
int monortm_hip_tape3_probe(const char *tape3_path, double v1, double v2, long long *n_physical, long long *n_entries,
                            long long *n_coupled);

/* 1 when the context holds a line table (a TAPE3 path was given at init), else 0.  A context created with an empty
 * path serves CALCTMR / RTM only; MODM on it returns MONORTM_EARG (the reference always loads the line file on its
 * first MODM call: INIT flag, src/modm.f90:187-190). */
int monortm_hip_has_lines(void *ctx);

/* Number of cross-section regions the context holds (monortm_hip_xsec_tables), 0 before the first upload.  The Fortran
 * shim uses it to skip the per-profile re-upload that would mirror the reference's per-call re-read of the xs files
 * (src/monortm_sub.F90:1659-1673) when neither the context nor the /XSECTR/, /XSECTF/ entries have changed. */
int monortm_hip_xsec_regions(void *ctx);

/* Known-answer hook for the parity tests: evaluates ONE of the small device functions of the path for n argument sets on
 * the GPU.  args[n][4] in, out[n][2].  which: 1 W4(x,y) -> re,im (src/modm.f90:1100)  2 SD_Humlicek(x1,y1,x2,y2) -> re,im
 * (:1150)  3 SDVOIGT(deltnu,alphal,alphad,sdep) (:965)  4 RADFN(vi,xkt) (src/lblrtm_sub.f90:36)  5 AtoB(aa) on the TIPS
 * temperature grid with the 119-point table tab119 (src/tips_2003.f90:4610)  6 ODCLW_TKC(wn,temp,clw)
 * (src/CloudOptProp.f90:29)  7 scor(mol,iso) of TIPS_2003(39,T,scor) for args (T, mol, iso) (src/tips_2003.f90:2-298; out[.][1] = 1
 * where the reference would STOP, 0 where it leaves scor untouched).  Returns MONORTM_ESDV when SDVOIGT meets the
 * reference's STOP condition. */
int monortm_hip_kat(void *ctx, int which, int n, const double *args, const double *tab119, double *out);

/* Diagnostics: which = 0 -> number of monortm_hip_rtm calls on this context that found the optical depths O of the
 * preceding monortm_hip_modm call still resident on the device (the caller handed back exactly what MODM returned, as
 * PROGRAM MONORTM does at src/monortm.f90:567-574) and skipped the upload.  -1 for an unknown selector / NULL. */
long long monortm_hip_counter(void *ctx, int which);

/* Physical line records (IFLG >= 0) held for molecule mol (1..39); mol = 0 -> all molecules.
 * This is NBLM(mol) minus the coupling records (src/lnfl_mod.f90:66) and is what the
 * (wavenumber x layer x line) evaluation count of BASELINE.json is made of. */
long long monortm_hip_line_count(void *ctx, int mol);

/* MODM, host buffers (what the Fortran shim calls).  Replaces src/modm.f90:21-274.
 *   wn[nwn] ascending cm-1;  dvset: COMMON /MANE/ DVSET of the caller (0 => explicit channels; /= 0 promises wn[i] = wn[0] + i dvset,
 *   which the continuum interpolation relies on as the reference's does - a grid that breaks it is MONORTM_EARG);
 *   nlay[nprof] layers per profile (<= nlay_max);  nmol molecules (7..39), same for the batch;
 *   P,T,CLW,WBRODL [nprof][nlay_max];  WKL [nprof][nlay_max][nmol];
 *   cntnm_fac[7] = XSELF,XFRGN,XCO2C,XO3CN,XO2CN,XN2CN,XRAYL (CntnmFactors_t, CntnmFactors.f90:17-19);
 * outputs (zero-filled for layers >= nlay[p]):
 *   O [nprof][nlay_max][nwn], O_BY_MOL [nprof][nlay_max][nmol][nwn],
 *   OC [nprof][nlay_max][MONORTM_NCONT][nwn], O_CLW [nprof][nlay_max][nwn]. */
int monortm_hip_modm(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay,
                     int nlay_max, int nmol, const monortm_real *P, const monortm_real *T,
                     const monortm_real *CLW, const monortm_real *WKL, const monortm_real *WBRODL,
                     const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd, int ixsect,
                     monortm_real *O, monortm_real *O_BY_MOL, monortm_real *OC, monortm_real *O_CLW);

/* Cross-section molecules (IXSECT = 1; replaces MONORTM_XSEC_SUB + convolve, src/monortm_sub.F90:1540-1834, called from
 * src/modm.f90:197).  The reference keeps the tables in COMMON /XSECTR/, /XSECTF/ (filled by XSREAD from FSCDXS, :1246-1421)
 * and re-reads the xs files in every call; here the caller hands the parsed tables over once per context (or whenever they
 * change):
 *   nxs molecules (in the order of the request = the second axis of XAMNT), nreg (molecule, spectral region) rows;
 *   reg[nreg][8] = molecule (0-based position in the request), V1FX, V2FX of the FSCDXS entry (a region is processed when some
 *   wavenumber of the call lies within 1 cm-1 of them, :1645), points per spectrum, temperatures (1..6), XDOPLR (:1383-1387),
 *   V1 and V2 on the header of the LAST xs file of the region (with the point count they define the grid of every spectrum of
 *   the region and the in-range test of a wavenumber, :1663-1666, :1709, :1789); temps[nreg][6] K ascending; pres_mb[nreg][6] measurement pressures in millibar (torr x 1013/760, :1626);
 *   offs[nreg][6] offsets of the spectra in pool[npool].  A multi-device context uploads to every device. */
int monortm_hip_xsec_tables(void *ctx, int nxs, int nreg, const double *reg, const double *temps, const double *pres_mb,
                            const long long *offs, const double *pool, long long npool);

/* MODM with the hidden inputs / the output of the cross-section path as arguments: XAMNT [nprof][nlay_max][nxs] = the
 * reference's COMMON /PATHX/ XAMNT (src/monortm.f90:233,:526-528), ODXSEC [nprof][nlay_max][nwn] = its ODXSEC argument
 * (src/modm.f90:24; total over the molecules, added into O at :268).  ixsect = 0: both may be NULL and the call is
 * monortm_hip_modm.  Not reproduced: the reference's own driver allocates ODXSEC(nwn, .) while MONORTM_XSEC_SUB indexes it
 * as (NWNMX, MXLAY) (:1611) and so writes out of bounds for every layer but the first; and convolve() overruns its
 * 10^7-point work array for layers whose pressure is below that of the measurement (:1758,:1773-1786) - the formulas are
 * evaluated as written, without those arrays. */
int monortm_hip_modm_xs(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay,
                        int nlay_max, int nmol, const monortm_real *P, const monortm_real *T,
                        const monortm_real *CLW, const monortm_real *WKL, const monortm_real *WBRODL,
                        const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd, int ixsect,
                        const monortm_real *XAMNT, monortm_real *ODXSEC,
                        monortm_real *O, monortm_real *O_BY_MOL, monortm_real *OC, monortm_real *O_CLW);

/* CALCTMR + RTM, host buffers.  Replaces src/RTMmono.f90:239-325 and :13-221.
 *   irt[nprof] 1 up / 2 limb / 3 down;  iout = 1 => TB computed;  T [nprof][nlay_max], TZ [nprof][nlay_max+1];
 *   O [nprof][nlay_max][nwn];  tmpsfc[nprof] is IN/OUT exactly like the reference's TMPSFC argument
 *   (set to 2.75 K for irt = 2,3; RTMmono.f90:113-124);  emiss, reflc [nprof][nwn];
 * outputs [nprof][nwn]: RUP, RDN, TRTOT, RAD, TB, TMR (TMR may be NULL to skip CALCTMR). */
int monortm_hip_rtm(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max,
                    const int *irt, int iout, const monortm_real *T, const monortm_real *TZ,
                    const monortm_real *O, monortm_real *tmpsfc, const monortm_real *emiss,
                    const monortm_real *reflc, monortm_real *RUP, monortm_real *RDN, monortm_real *TRTOT,
                    monortm_real *RAD, monortm_real *TB, monortm_real *TMR);

/* Same operations on DEVICE pointers, asynchronous on `stream` (hipStream_t, may be NULL).
 * nlay / irt are device int arrays, tmpsfc a device monortm_real array.  The context's device must be the calling
 * thread's current device (checked: MONORTM_EARG otherwise); the host-buffer entry points select it themselves.
 * wn_ends: HOST array {wn[0], wn[nwn-1]} (they size the continuum grid, modm.f90:180-185), or NULL - then the two
 * values are read back from device memory, which synchronises `stream` once per call. */
int monortm_hip_modm_dev(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay,
                         int nlay_max, int nmol, const monortm_real *P, const monortm_real *T,
                         const monortm_real *CLW, const monortm_real *WKL, const monortm_real *WBRODL,
                         const double *cntnm_fac /*host*/, double sclcpl, double sclhw, double y0res, int ibrd,
                         int ixsect, monortm_real *O, monortm_real *O_BY_MOL, monortm_real *OC,
                         monortm_real *O_CLW, const double *wn_ends, void *stream);

int monortm_hip_modm_xs_dev(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay,
                            int nlay_max, int nmol, const monortm_real *P, const monortm_real *T,
                            const monortm_real *CLW, const monortm_real *WKL, const monortm_real *WBRODL,
                            const double *cntnm_fac /*host*/, double sclcpl, double sclhw, double y0res, int ibrd,
                            int ixsect, const monortm_real *XAMNT, monortm_real *ODXSEC, monortm_real *O,
                            monortm_real *O_BY_MOL, monortm_real *OC, monortm_real *O_CLW, const double *wn_ends,
                            void *stream);

int monortm_hip_rtm_dev(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max,
                        const int *irt, int iout, const monortm_real *T, const monortm_real *TZ,
                        const monortm_real *O, monortm_real *tmpsfc, const monortm_real *emiss,
                        const monortm_real *reflc, monortm_real *RUP, monortm_real *RDN, monortm_real *TRTOT,
                        monortm_real *RAD, monortm_real *TB, monortm_real *TMR, void *stream);

/* Device-side failure flags raised by the kernels of earlier *_dev calls (temperature range, SD-Voigt
 * sign): synchronises `stream`, returns MONORTM_OK or the first error and clears the flags. */
int monortm_hip_check(void *ctx, void *stream);

/* Kernel timing: HIP events recorded on the launch stream around the kernel launches selected by the bit mask
 * `enable` (bit 0 = line sum, bit 1 = continuum+cloud+total, bit 2 = rtm; 0 = off).  Bits 8 and up: sample stride n -
 * only every n-th launch of a selected kernel is bracketed (an event pair keeps the next launch from being queued behind
 * the running kernel, a few microseconds per step; 0 or 1 = every launch).
 * monortm_hip_kernel_time(kernel = 0,1,2) synchronises the recorded events and returns the running totals. */
int monortm_hip_profile(void *ctx, int enable);

/* Measurement switches of a context - how the line-sum launch is shaped.  No reference counterpart (the reference has no
 * tuning knobs on this path); results are the same to rounding whatever is chosen.  The environment variables
 * MONORTM_NSLICE / MONORTM_FAIR / MONORTM_TILE_WAVES / MONORTM_FAR_LEVELS give the defaults once, at monortm_hip_init (a value
 * that does not parse fails the init with MONORTM_EARG).
 *   "nslice" = "auto" | 1..16;  "fair" = "auto" | 0 | 1;  "tile_waves" = "auto" | 1 | 2 | 4;
 *   "far_levels" = "auto" | 0..6: dense grids (>= 4 tiles of wavenumbers) - levels of intervals (tiles, pairs of tiles, fours, eights ...)
 *       whose far lines far_kernel expands before the line sum; 0 = the far field of a tile is formed inside the line-sum kernel;
 *   "lines_kernel" = "auto" | "wn" (the one kernel; the round-3 alternatives "state" / "p" were removed in round 5).
 * Values are parsed strictly (whole string, in range).  Unknown names / values: MONORTM_EARG. */
int monortm_hip_set_option(void *ctx, const char *name, const char *value);
int monortm_hip_kernel_time(void *ctx, int kernel, double *total_ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif
