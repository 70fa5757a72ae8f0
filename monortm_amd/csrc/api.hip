// api.hip - host side of the C ABI (include/monortm_hip.h): context, TAPE3 -> device line table, model tables,
// launch configuration, host-buffer front ends for the Fortran shim.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <string>
#include <vector>

#include "device_common.hpp"
#include "line_table.hpp"
#include "lineshape.hpp"
#include "tables/monortm_tables.h"

namespace {
using namespace monortm_dev;

// ------------------------------------------------------------------------------------------------
// host side: context, uploads, launches
// ------------------------------------------------------------------------------------------------
thread_local std::string g_init_error;

struct Ctx {
    int device = 0;
    int cus = 256;            // compute units of the device (MI355X: 256): the launch heuristics count wave slots with it
    // workspaces of dense grids may grow to a share of the device's memory (MI355X: 288 GB -> 36 GB of line-physics records, 9 GB of
    // far-field sums; a grid that needs more has its line physics formed in place and its far field inside lines_kernel)
    size_t phys_cap = 2ull << 30, far_cap = 1ull << 30;
    // multi-device context (monortm_hip_init_multi): no device resources of its own, one full context per device
    std::vector<Ctx *> shards;
    // measurement switches (monortm_hip_set_option; the environment variables MONORTM_NSLICE / _FAIR /
    // _TILE_WAVES give their defaults ONCE, when the context is created - nothing on the launch path calls getenv)
    struct Opt {
        int nslice = 0;           // 0 = chosen per call
        int fair = -1;            // -1 = chosen per call, 0 / 1 wave priorities off / on
        int tile_waves = 0;       // 0 = lines_config(); 1 / 2 / 4 waves per workgroup of two-wavenumber tiles
        int far_levels = -1;      // -1 = chosen per call; 0 = lines_kernel forms the far field of dense grids itself; 1 .. 6 levels of far_kernel
        int lines_ms = -1;        // -1 = chosen per call; 0 = never lines_ms_kernel; 1 = whenever its layout fits (tests, measurements)
        int ms_ablate = 0;        // timing experiments: lines_ms_kernel without its later stages (wrong results)
        int ms_items = 0;         // 0 = chosen per call; 64 / 128 / 192 (state, line) items per chunk of lines_ms_kernel
    } opt;
    void *comm = nullptr;     // RCCL communicator of a multi-process job (monortm_hip_comm_init), one rank per context
    int comm_rank = 0, comm_world = 1;
    bool has_lines = false;  // a TAPE3 was loaded (a context created with an empty path serves RTM / CALCTMR only)
    int real_kind = 8;  // element size of the caller's REAL arrays (the reference's "dbl" / "sgl" builds)
    std::string err;
    monortm::LineTable host;
    DevLines lines{};
    DevTables tables{};
    std::vector<void *> owned;
    int *errflag = nullptr;
    void *partial = nullptr;  // line-slice workspace, grown on demand
    monortm_dev::MwCache mw_cache;  // spectral-range constants of finish_mw_kernel (launch_finish_mw)
    void *phys = nullptr;     // per (profile, layer, line) LinePhys records of dense grids (physics_kernel), grown on demand
    size_t phys_bytes = 0;
    void *far = nullptr;      // far_kernel's sums, interval geometry and candidate runs of dense grids, grown on demand
    size_t far_bytes = 0;
    double lines_per_cm = -1.;          // lines of the table per cm-1 (sizes far_kernel's workgroups), formed at the first dense call
    hipStream_t far_stream = nullptr;   // far_plan_kernel runs beside physics_kernel (fork / join with far_ev)
    hipEvent_t far_ev[2] = {nullptr, nullptr};
    // lines_ms_kernel (batches of states on sparse channel sets): slot of (molecule, isotopologue 1) among the isotopologues the
    // table holds - on the device and here - and the scratch of the rare shapes, grown on demand
    double lc_frac = 0.;      // share of the table's lines that carry line-coupling coefficients
    const int *ms_slot_base = nullptr;
    int ms_slot_host[MXMOL + 1] = {0};
    void *ms_scratch = nullptr;
    unsigned short *ms_reach = nullptr;  // per table line: the slots of channels it / its negative resonance can reach (ms_reach_kernel)
    size_t ms_scratch_bytes = 0;
    double *osum = nullptr;   // per (profile, layer, wn) line sums handed from lines_kernel to finish_mw_kernel, grown on demand
    size_t osum_elems = 0;
    DevXsec xs{};             // cross-section tables (monortm_hip_xsec_tables); xs_buf holds them, replaced as a whole
    std::vector<void *> xs_buf;
    // staging buffers of the host-buffer entry points, one per argument, grown on demand and kept: a caller that loops
    // over profiles (the reference's driver does) pays for device allocations once, not per call
    struct Stage {
        void *p = nullptr;
        size_t bytes = 0;
    };
    Stage stage[8];  // modm: host in, device in, host out, device out; rtm: the same four
    // The host-buffer entry points run on their own stream: one asynchronous upload, the kernels, one asynchronous
    // download (+ the 4-byte error flag) and a single synchronisation per call.
    hipStream_t hs = nullptr;
    int *errflag_host = nullptr;  // pinned
    // What the last host-buffer MODM left behind: the total optical depths O on the device and their pinned host copy.
    // The reference's driver hands exactly these values to CALCTMR and RTM next (monortm.f90:557-574); when the caller's
    // O compares equal to the copy, monortm_hip_rtm reads the resident array instead of uploading it again.
    struct {
        const void *dev = nullptr, *host = nullptr;
        size_t bytes = 0;
        int nprof = 0, nwn = 0, nlay_max = 0;
    } lastO;
    long long o_reused = 0;  // calls of monortm_hip_rtm that found O resident
    // MONORTM_HOST_TIMING=1: wall time of the host-buffer calls by phase (pack, enqueue, wait, unpack), printed at finalize
    bool host_timing = false;
    double ht[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    long long ht_calls[2] = {0, 0};
    size_t partial_elems = 0;
    int profiling = 0;  // bit k set: record events around kernel k
    int prof_stride = 1;  // ... around every prof_stride-th launch of it (an event pair costs a few microseconds of the stream)
    long long prof_calls[3] = {0, 0, 0};
    struct Ev {
        hipEvent_t a, b;
        int k;
    };
    std::vector<Ev> events;
    std::vector<hipEvent_t> event_pool;  // events are created once and reused: no hipEventCreate inside a timed region
    double tot_ms[3] = {0, 0, 0};
    long long launches[3] = {0, 0, 0};
};

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return MONORTM_EHIP;                                                                     \
        }                                                                                            \
    } while (0)

// Small host-buffer calls (one profile per call - the reference driver's pattern) move their arenas with a copy KERNEL
// that reads / writes the pinned host arena directly over PCIe: a DMA-engine copy costs tens of microseconds of latency
// each, a few hundred KB through a kernel a few.  Large batches keep hipMemcpyAsync.
constexpr size_t kKernelCopyMax = 4u << 20;
__global__ __launch_bounds__(256) void copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                     const uint4 *__restrict__ src2, uint4 *__restrict__ dst2, size_t n16b,
                                                     const int *flag_src, int *flag_dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == gridDim.x - 1)
        for (size_t i = threadIdx.x; i < n16b; i += 256) dst2[i] = src2[i];
    if (flag_dst && blockIdx.x == 0 && threadIdx.x == 0) *flag_dst = *flag_src;  // error flags of the kernels before this one
}
// arenas and their pieces are multiples of 256 bytes (Arena::add); the optional second piece (src2 -> dst2) is small
hipError_t move_arena(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s, const int *flag_src = nullptr,
                      int *flag_dst = nullptr, void *dst2 = nullptr, const void *src2 = nullptr, size_t bytes2 = 0) {
    if (bytes <= kKernelCopyMax) {
        const size_t n16 = bytes / 16;
        const unsigned blocks = (unsigned)std::min<size_t>(256, std::max<size_t>(1, (n16 + 255) / 256));
        hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16,
                           static_cast<const uint4 *>(src2), static_cast<uint4 *>(dst2), bytes2 / 16, flag_src, flag_dst);
        return hipGetLastError();
    }
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
    if (e == hipSuccess && flag_dst) e = hipMemcpyAsync(flag_dst, flag_src, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && bytes2) e = hipMemcpyAsync(dst2, src2, bytes2, hipMemcpyDefault, s);
    return e;
}

void comm_release(Ctx *c);  // (RCCL communicator of the context, defined with the gather entry points)

// one measurement switch from its textual value; unknown names / values are refused (MONORTM_EARG)
constexpr double MONORTM_FAR_KAPPA_HOST = 1.2;   // = MONORTM_FAR_KAPPA of lines_device.hpp (least distance of a far line in tile half-widths)

int set_option(Ctx *c, const char *name, const char *value) {
    const std::string n = name ? name : "", v = value ? value : "";
    const bool autov = v.empty() || v == "auto";
    long iv = 0;
    bool isint = false;
    if (!autov) {
        char *end = nullptr;
        iv = strtol(v.c_str(), &end, 10);
        isint = end && end != v.c_str() && *end == '\0';
    }
    if (n == "nslice" && (autov || (isint && iv >= 1 && iv <= 16))) c->opt.nslice = autov ? 0 : (int)iv;
    else if (n == "fair" && (autov || (isint && (iv == 0 || iv == 1)))) c->opt.fair = autov ? -1 : (int)iv;
    else if (n == "tile_waves" && (autov || (isint && (iv == 1 || iv == 2 || iv == 4)))) c->opt.tile_waves = autov ? 0 : (int)iv;
    else if (n == "far_levels" && (autov || (isint && iv >= 0 && iv <= FAR_MAXLEV))) c->opt.far_levels = autov ? -1 : (int)iv;
    // lines_kernel: auto = by batch size; wn = lines_kernel (one state per wave) always; ms = lines_ms_kernel (several states per wave,
    // five wavenumbers per lane) wherever its layout fits (double precision, <= 64 wavenumbers)
    else if (n == "lines_kernel" && (autov || v == "wn" || v == "ms")) c->opt.lines_ms = autov ? -1 : (v == "ms" ? 1 : 0);
#ifdef MONORTM_EXPERIMENT   // stage ablations of lines_ms_kernel: timing only, wrong results - experiment builds alone know the option
    else if (n == "ms_ablate" && isint && iv >= 0 && iv <= 9) c->opt.ms_ablate = (int)iv;
#endif
    else if (n == "ms_items" && (autov || (isint && (iv == 64 || iv == 128 || iv == 192 || iv == 256)))) c->opt.ms_items = autov ? 0 : (int)iv;
    else { c->err = "unknown option or value: " + n + " = " + v; return MONORTM_EARG; }
    return MONORTM_OK;
}

template <class T>
int upload(Ctx *c, const T *src, size_t n, const T **dst) {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(c, hipMalloc(&p, bytes));
    c->owned.push_back(p);
    if (n) HIPCHK(c, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return MONORTM_OK;
}

void prof_begin(Ctx *c, hipStream_t s, int k, Ctx::Ev &ev) {
    ev.k = -1;
    if (!((c->profiling >> k) & 1)) return;
    if (c->prof_stride > 1 && (c->prof_calls[k]++ % c->prof_stride) != 0) return;
    auto take = [&](hipEvent_t *e) {
        if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); }
        else hipEventCreate(e);
    };
    take(&ev.a);
    take(&ev.b);
    ev.k = k;
    hipEventRecord(ev.a, s);
}
void prof_end(Ctx *c, hipStream_t s, Ctx::Ev &ev) {
    if (ev.k < 0) return;
    hipEventRecord(ev.b, s);
    c->events.push_back(ev);
}

// the *_dev entry points launch on the calling thread's current device: it must be the context's
int check_device(Ctx *c) {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d != c->device) {
        c->err = "the context was created on device " + std::to_string(c->device) + " but device " + std::to_string(d) +
                 " is current: call hipSetDevice (torch.cuda.set_device) before the *_dev entry points";
        return MONORTM_EARG;
    }
    return MONORTM_OK;
}

int check_modm_args(Ctx *c, int nprof, int nwn, int nlay_max, int nmol, int ibrd, int ixsect, double v2) {
    if (nprof < 1 || nwn < 1 || nlay_max < 1 || nlay_max > 603) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    if (nmol < 7 || nmol > MXMOL) { c->err = "nmol must be 7..39 (LINES reads WK(1:7), modm.f90:313)"; return MONORTM_EARG; }
    if (nwn > 80000) { c->err = "nwn exceeds NWNMX=80000 (RTMmono.f90:10)"; return MONORTM_EARG; }
    if (ixsect != 0 && ixsect != 1) { c->err = "ixsect must be 0 or 1"; return MONORTM_EARG; }
    if (ibrd != 0 && !c->host.any_brd) { /* nothing to do: flags all zero, same as ibrd = 0 */ }
    (void)v2;
    return MONORTM_OK;
}

Ctx *g_timing_ctx = nullptr;
void print_host_timing(Ctx *c) {
    if (!c->host_timing) return;
    for (int k = 0; k < 2; k++)
        if (c->ht_calls[k])
            fprintf(stderr, "monortm_hip %s: %lld calls, us per call: pack %.1f enqueue %.1f wait %.1f unpack %.1f\n", k ? "rtm " : "modm",
                    c->ht_calls[k], c->ht[k][0] / c->ht_calls[k] * 1e6, c->ht[k][1] / c->ht_calls[k] * 1e6,
                    c->ht[k][2] / c->ht_calls[k] * 1e6, c->ht[k][3] / c->ht_calls[k] * 1e6);
}

// device pointers, streams and kernel timers belong to ONE device: a multi-device context serves the host-buffer calls
int multi_only_host(Ctx *c) {
    c->err = "multi-device context: only monortm_hip_modm / monortm_hip_rtm (host buffers) shard over devices; create one "
             "context per device with monortm_hip_init for the *_dev / profiling entry points";
    return MONORTM_EARG;
}

int null_ctx() {
    g_init_error = "null context";
    return MONORTM_EARG;
}

// The host-buffer entry points and the multi-device set-up switch devices (hipSetDevice is per thread): the caller's current
// device is put back on every return path, so that allocations and *_dev calls made afterwards land where they did before.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

}  // namespace

// ---- host-buffer front ends (what the Fortran shim calls): stage through device memory -----------
// All inputs of a call are packed into one pinned host arena and travel in ONE host-to-device copy, all outputs come back
// in ONE device-to-host copy: a caller that loops over profiles (the reference's driver does) pays two synchronous
// copies per call instead of one per argument.  The arenas grow on demand and stay with the context.
namespace {
struct Arena {  // layout helper: 256-byte aligned pieces of one buffer
    size_t size = 0;
    size_t add(size_t bytes) {
        const size_t off = size;
        size = (size + std::max<size_t>(bytes, 8) + 255) & ~size_t(255);
        return off;
    }
};
struct HostClock {  // phase timer of a host-buffer call (only when MONORTM_HOST_TIMING=1)
    Ctx *c;
    int k;
    std::chrono::steady_clock::time_point t;
    HostClock(Ctx *c_, int k_) : c(c_), k(k_) {
        if (c->host_timing) { t = std::chrono::steady_clock::now(); c->ht_calls[k]++; }
    }
    void mark(int phase) {
        if (!c->host_timing) return;
        const auto n = std::chrono::steady_clock::now();
        c->ht[k][phase] += std::chrono::duration<double>(n - t).count();
        t = n;
    }
};
hipError_t stage_get(Ctx *c, int slot, size_t bytes, bool pinned_host, void **out) {
    Ctx::Stage &st = c->stage[slot];
    if (st.bytes < bytes) {
        if (st.p) {
            if (pinned_host) hipHostFree(st.p);
            else hipFree(st.p);
        }
        st.p = nullptr;
        st.bytes = 0;
        const size_t want = bytes + bytes / 4;  // some head room: profiles of a run differ little in size
        const hipError_t e = pinned_host ? hipHostMalloc(&st.p, want, hipHostMallocDefault) : hipMalloc(&st.p, want);
        if (e != hipSuccess) return e;
        st.bytes = want;
    }
    *out = st.p;
    return hipSuccess;
}
}  // namespace

namespace {
int decode_flag(Ctx *c, int flag, hipStream_t s) {
    if (!flag) return MONORTM_OK;
    HIPCHK(c, hipMemsetAsync(c->errflag, 0, sizeof(int), s));
    if (flag & ERRBIT_ARG) { c->err = "device arguments: nlay[p] outside 1..nlay_max, wavenumbers not ascending or not the grid dvset promises"; return MONORTM_EARG; }
    if (flag & ERRBIT_TEMP) { c->err = "TIPS: layer temperature outside 70-3000 K / partition sum <= 0 (reference STOP, tips_2003.f90:277)"; return MONORTM_ETEMP; }
    c->err = "SDVOIGT: REAL(v) < 0 (reference STOP, modm.f90:1062)";
    return MONORTM_ESDV;
}
}  // namespace

namespace {
// One device's share of a host-buffer MODM call.  With `defer` the call returns once everything is enqueued on the
// context's stream and *defer completes it (wait, error flags, unpack): a multi-device context enqueues on every device
// before it waits for the first.
static int modm_host(Ctx *c, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                     int nmol, const void *P, const void *T, const void *CLW, const void *WKL,
                     const void *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd,
                     int ixsect, const void *XAMNT, void *ODXSEC, void *O, void *O_BY_MOL, void *OC, void *O_CLW,
                     std::function<int()> *defer) {
    void *ctx = c;
    if (!wn || !nlay || !P || !T || !CLW || !WKL || !WBRODL || !cntnm_fac || !O || !O_BY_MOL || !OC || !O_CLW) { c->err = "null array argument"; return MONORTM_EARG; }
    if (nprof < 1 || nwn < 1 || nlay_max < 1 || nmol < 1) { c->err = "bad nprof/nwn/nlay_max/nmol"; return MONORTM_EARG; }
    for (int i = 1; i < nwn; i++)
        if (!(wn[i] >= wn[i - 1])) { c->err = "wavenumbers must be ascending (the reference takes v1 = wn(1), v2 = wn(nwn), modm.f90:180-181)"; return MONORTM_EARG; }
    // dvset /= 0: what the kernels rely on is the nearest grid index within +-1 (line_records) and the continuum at V1 + i dvset
    // as the reference's own CONTNM call places it - so the bound is on the cumulative drift, not on consecutive differences: the
    // reference's "sgl" driver forms WN(J) = V1 + (J-1)*DVSET with a REAL*4 product (src/monortm_sub.F90:287), which moves
    // point J by up to 6e-8 J DVSET
    if (dvset != 0.)
        for (int i = 1; i < nwn; i++)
            if (!(std::fabs(wn[i] - (wn[0] + i * dvset)) <= 0.25 * std::fabs(dvset))) { c->err = "dvset /= 0 promises the grid wn[i] = wn[0] + i dvset (COMMON /MANE/ DVSET of the reference driver)"; return MONORTM_EARG; }
    for (int p = 0; p < nprof; p++)
        if (nlay[p] < 1 || nlay[p] > nlay_max) { c->err = "nlay[p] outside 1..nlay_max"; return MONORTM_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    c->lastO.dev = nullptr;  // before any arena can move or fail to regrow: nothing may point into a freed arena afterwards
    c->lastO.host = nullptr;
    const double ends[2] = {wn[0], wn[nwn - 1]};
    const size_t npl = (size_t)nprof * nlay_max, d = (size_t)c->real_kind;
    Arena in, out;
    const size_t b_wn = nwn * sizeof(double), b_nl = nprof * sizeof(int), b_l = npl * d, b_w = npl * nmol * d;
    const size_t i_wn = in.add(b_wn), i_nl = in.add(b_nl), i_P = in.add(b_l), i_T = in.add(b_l), i_C = in.add(b_l), i_W = in.add(b_w),
                 i_B = in.add(b_l);
    const bool xs = ixsect == 1;
    if (xs && (!XAMNT || !ODXSEC || c->xs.nxs < 1)) { c->err = "IXSECT = 1: XAMNT / ODXSEC / cross-section tables missing"; return MONORTM_EARG; }
    const size_t b_xa = xs ? npl * (size_t)c->xs.nxs * d : 0, i_XA = xs ? in.add(b_xa) : 0;
    const size_t b_o = npl * nwn * d, b_om = npl * nmol * nwn * d, b_oc = npl * MONORTM_NCONT * nwn * d;
    const size_t o_O = out.add(b_o), o_OM = out.add(b_om), o_OC = out.add(b_oc), o_OL = out.add(b_o), o_OX = xs ? out.add(b_o) : 0;
    void *hin = nullptr, *din = nullptr, *hout = nullptr, *dout = nullptr;
    HIPCHK(c, stage_get(c, 0, in.size, true, &hin));
    HIPCHK(c, stage_get(c, 1, in.size, false, &din));
    HIPCHK(c, stage_get(c, 2, out.size, true, &hout));
    HIPCHK(c, stage_get(c, 3, out.size, false, &dout));
    char *h = static_cast<char *>(hin), *dv = static_cast<char *>(din), *dz = static_cast<char *>(dout);
    HostClock hc(c, 0);
    memcpy(h + i_wn, wn, b_wn); memcpy(h + i_nl, nlay, b_nl); memcpy(h + i_P, P, b_l); memcpy(h + i_T, T, b_l);
    memcpy(h + i_C, CLW, b_l); memcpy(h + i_W, WKL, b_w); memcpy(h + i_B, WBRODL, b_l);
    if (xs) memcpy(h + i_XA, XAMNT, b_xa);
    hc.mark(0);
    HIPCHK(c, move_arena(din, hin, in.size, hipMemcpyHostToDevice, c->hs));
    int rc = monortm_hip_modm_xs_dev(ctx, nprof, nwn, (double *)(dv + i_wn), dvset, (int *)(dv + i_nl), nlay_max, nmol, dv + i_P, dv + i_T,
                                     dv + i_C, dv + i_W, dv + i_B, cntnm_fac, sclcpl, sclhw, y0res, ibrd, ixsect,
                                     xs ? dv + i_XA : nullptr, xs ? dz + o_OX : nullptr, dz + o_O, dz + o_OM, dz + o_OC, dz + o_OL, ends, c->hs);
    if (rc) return rc;
    HIPCHK(c, move_arena(hout, dout, out.size, hipMemcpyDeviceToHost, c->hs, c->errflag, c->errflag_host));
    hc.mark(1);
    auto complete = [=]() mutable -> int {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->hs));
        hc.mark(2);
        if (int rcf = decode_flag(c, *c->errflag_host, c->hs)) return rcf;
        const char *ho = static_cast<const char *>(hout);
        memcpy(O, ho + o_O, b_o); memcpy(O_BY_MOL, ho + o_OM, b_om); memcpy(OC, ho + o_OC, b_oc); memcpy(O_CLW, ho + o_OL, b_o);
        if (xs) memcpy(ODXSEC, ho + o_OX, b_o);
        hc.mark(3);
        c->lastO.dev = dz + o_O; c->lastO.host = ho + o_O; c->lastO.bytes = b_o;
        c->lastO.nprof = nprof; c->lastO.nwn = nwn; c->lastO.nlay_max = nlay_max;
        return MONORTM_OK;
    };
    if (defer) { *defer = complete; return MONORTM_OK; }
    return complete();
}



static int rtm_host(Ctx *c, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max, const int *irt,
                    int iout, const void *T, const void *TZ, const void *O, void *tmpsfc, const void *emiss,
                    const void *reflc, void *RUP, void *RDN, void *TRTOT, void *RAD, void *TB, void *TMR,
                    std::function<int()> *defer) {
    void *ctx = c;
    if (!wn || !nlay || !irt || !T || !TZ || !O || !tmpsfc || !emiss || !reflc || !RUP || !RDN || !TRTOT || !RAD || !TB) { c->err = "null array argument"; return MONORTM_EARG; }
    for (int p = 0; p < nprof; p++)
        if (nlay[p] < 1 || nlay[p] > nlay_max) { c->err = "nlay[p] outside 1..nlay_max"; return MONORTM_EARG; }
    if (nprof < 1 || nwn < 1 || nlay_max < 1) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npl = (size_t)nprof * nlay_max, d = (size_t)c->real_kind, pw = (size_t)nprof * nwn;
    Arena in, out;
    const size_t b_wn = nwn * sizeof(double), b_i = nprof * sizeof(int), b_l = npl * d, b_tz = (size_t)nprof * (nlay_max + 1) * d,
                 b_o = npl * nwn * d, b_p = nprof * d, b_pw = pw * d;
    // O straight from the preceding MODM call?  Then it is still on the device (see Ctx::lastO): no second upload
    const bool resident = c->lastO.dev && c->lastO.nprof == nprof && c->lastO.nwn == nwn && c->lastO.nlay_max == nlay_max &&
                          c->lastO.bytes == npl * nwn * d && memcmp(O, c->lastO.host, c->lastO.bytes) == 0;
    if (resident) c->o_reused++;
    const size_t i_wn = in.add(b_wn), i_nl = in.add(b_i), i_irt = in.add(b_i), i_T = in.add(b_l), i_TZ = in.add(b_tz),
                 i_em = in.add(b_pw), i_rf = in.add(b_pw), i_ts = in.add(b_p), i_O = resident ? 0 : in.add(b_o);
    // tmpsfc is in/out: it lives in the output arena and is seeded from the host before the launch
    const size_t o_up = out.add(b_pw), o_dn = out.add(b_pw), o_tr = out.add(b_pw), o_rad = out.add(b_pw), o_tb = out.add(b_pw),
                 o_tmr = out.add(b_pw), o_ts = out.add(b_p);
    void *hin = nullptr, *din = nullptr, *hout = nullptr, *dout = nullptr;
    HIPCHK(c, stage_get(c, 4, in.size, true, &hin));
    HIPCHK(c, stage_get(c, 5, in.size, false, &din));
    HIPCHK(c, stage_get(c, 6, out.size, true, &hout));
    HIPCHK(c, stage_get(c, 7, out.size, false, &dout));
    char *h = static_cast<char *>(hin), *dv = static_cast<char *>(din), *dz = static_cast<char *>(dout);
    HostClock hc(c, 1);
    memcpy(h + i_wn, wn, b_wn); memcpy(h + i_nl, nlay, b_i); memcpy(h + i_irt, irt, b_i); memcpy(h + i_T, T, b_l);
    memcpy(h + i_TZ, TZ, b_tz); memcpy(h + i_em, emiss, b_pw); memcpy(h + i_rf, reflc, b_pw); memcpy(h + i_ts, tmpsfc, b_p);
    if (!resident) memcpy(h + i_O, O, b_o);
    hc.mark(0);
    // (tmpsfc is in/out: its piece of the input arena also seeds the output arena)
    HIPCHK(c, move_arena(din, hin, in.size, hipMemcpyHostToDevice, c->hs, nullptr, nullptr, dz + o_ts, h + i_ts, (b_p + 255) & ~size_t(255)));
    if (iout != 1) HIPCHK(c, hipMemsetAsync(dz + o_tb, 0, b_pw, c->hs));
    const void *dO = resident ? c->lastO.dev : static_cast<const void *>(dv + i_O);
    int rc = monortm_hip_rtm_dev(ctx, nprof, nwn, (double *)(dv + i_wn), (int *)(dv + i_nl), nlay_max, (int *)(dv + i_irt), iout,
                                 dv + i_T, dv + i_TZ, dO, dz + o_ts, dv + i_em, dv + i_rf, dz + o_up, dz + o_dn, dz + o_tr,
                                 dz + o_rad, dz + o_tb, TMR ? dz + o_tmr : nullptr, c->hs);
    if (rc) return rc;
    HIPCHK(c, move_arena(hout, dout, out.size, hipMemcpyDeviceToHost, c->hs));
    hc.mark(1);
    auto complete = [=]() mutable -> int {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->hs));
        hc.mark(2);
        const char *ho = static_cast<const char *>(hout);
        memcpy(RUP, ho + o_up, b_pw); memcpy(RDN, ho + o_dn, b_pw); memcpy(TRTOT, ho + o_tr, b_pw); memcpy(RAD, ho + o_rad, b_pw);
        if (iout == 1) memcpy(TB, ho + o_tb, b_pw);
        if (TMR) memcpy(TMR, ho + o_tmr, b_pw);
        memcpy(tmpsfc, ho + o_ts, b_p);
        hc.mark(3);
        return MONORTM_OK;
    };
    if (defer) { *defer = complete; return MONORTM_OK; }
    return complete();
}

// ---- sharding of a host-buffer call over the devices of a multi-device context (monortm_hip_init_multi) ------------
// Profiles are independent (src/monortm.f90:357 is a loop without cross-iteration data flow): device g takes the
// contiguous block [g ceil(P/G), (g+1) ceil(P/G)) of the batch, every device holds the whole line table, each block
// comes back over its own device's PCIe link straight into the caller's arrays.
void shard_block(int nprof, int G, int g, int *p0, int *n) {
    const int per = (nprof + G - 1) / G;
    *p0 = std::min(nprof, g * per);
    *n = std::min(nprof, *p0 + per) - *p0;
}
const char *off(const void *p, size_t bytes) { return static_cast<const char *>(p) + bytes; }
char *off(void *p, size_t bytes) { return static_cast<char *>(p) + bytes; }

}  // namespace

extern "C" {

const char *monortm_hip_last_error(void *ctx) {
    if (!ctx) return g_init_error.c_str();
    return static_cast<Ctx *>(ctx)->err.c_str();
}

int monortm_hip_init(const char *tape3_path, double v1, double v2, int icp, int real_kind, int device, void **out) {
    (void)icp;  // passed through to GET_LNFL by the reference and unused there (lnfl_mod.f90:22)
    *out = nullptr;
    if (real_kind != 8 && real_kind != 4) { g_init_error = "real_kind must be 8 (\"dbl\" build) or 4 (\"sgl\" build)"; return MONORTM_EUNSUPPORTED; }
    Ctx *c = new Ctx;
    c->real_kind = real_kind;
    auto failed = [&](int rc) { g_init_error = c->err; for (void *p : c->owned) hipFree(p); delete c; return rc; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { c->err = "no HIP device available: the MI355X path has no CPU fallback"; return failed(MONORTM_EHIP); }
    if (device >= 0) { if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return failed(MONORTM_EHIP); } }
    hipGetDevice(&c->device);
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && n > 0) c->cus = n; }
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) {
            c->phys_cap = std::max<size_t>(c->phys_cap, total_b / 8);
            c->far_cap = std::max<size_t>(c->far_cap, total_b / 32);
        } else (void)hipGetLastError();
        // the caller's say on how much device memory the dense-grid workspaces may take (they are raw hipMalloc, outside any
        // framework's caching allocator, grow on demand and are released when a later call needs less than a quarter of them):
        // MONORTM_PHYS_CAP / MONORTM_FAR_CAP in MiB; 0 = no workspace of that kind (line physics in place / far field inside lines_kernel)
        for (int k = 0; k < 2; k++)
            if (const char *e = getenv(k ? "MONORTM_FAR_CAP" : "MONORTM_PHYS_CAP")) {
                char *end = nullptr;
                const double mib = strtod(e, &end);
                if (end == e || *end != '\0' || !(mib >= 0.)) { c->err = std::string(k ? "MONORTM_FAR_CAP" : "MONORTM_PHYS_CAP") + ": not a number of MiB"; return failed(MONORTM_EARG); }
                (k ? c->far_cap : c->phys_cap) = (size_t)(mib * 1048576.);
            }
    }
    // an empty path gives a context without a line table (RTM / CALCTMR need no TAPE3)
    int rc = MONORTM_OK;
    if (tape3_path && tape3_path[0]) {
        rc = monortm::load_tape3(tape3_path, v1, v2, c->host, c->err, c->real_kind);
        c->has_lines = rc == MONORTM_OK;
    }
    if (rc) return failed(rc);
    const monortm::LineTable &h = c->host;
    DevLines &L = c->lines;
#define UP(field, vec) if ((rc = upload(c, (vec).data(), (vec).size(), &L.field))) return failed(rc)
    UP(vnu, h.vnu); UP(s0adj, h.s0adj); UP(lc, h.lc); UP(alfa, h.alfa); UP(hwhm, h.hwhm); UP(epp, h.epp);
    UP(tmpalf, h.tmpalf); UP(pshift, h.pshift); UP(sdep, h.sdep); UP(meta, h.meta); UP(brd_flg, h.brd_flg); UP(brd_dat, h.brd_dat);
#undef UP
    for (int m = 0; m <= MXMOL + 1; m++) L.mol_start[m] = h.mol_start[m];
    L.sorted_mask = 0;
    for (int m = 1; m <= MXMOL; m++) if (h.sorted[m]) L.sorted_mask |= (1ull << m);
    L.max_abs_shift = h.max_abs_shift;
    L.lc_mask = 0;
    size_t ncoupled = 0;
    for (size_t i = 0; i < h.meta.size(); i++)
        if ((h.meta[i] >> 10) & 3) { L.lc_mask |= (1ull << (h.meta[i] & 63)); ncoupled++; }
    c->lc_frac = h.meta.empty() ? 0. : (double)ncoupled / (double)h.meta.size();
    {   // the isotopologues the table holds per molecule (at least the first: a line without a known one reads its Doppler factor)
        int iso_max[MXMOL + 1] = {0};
        for (size_t i = 0; i < h.meta.size(); i++) {
            const int mol = (int)(h.meta[i] & 63u), iso = (int)((h.meta[i] >> 6) & 15u);
            if (mol >= 1 && mol <= MXMOL) iso_max[mol] = std::max(iso_max[mol], std::max(iso, 1));
        }
        int acc = 0;
        for (int m = 0; m < MXMOL; m++) { c->ms_slot_host[m] = acc; acc += std::min(iso_max[m + 1], 9); }
        c->ms_slot_host[MXMOL] = acc;
        if ((rc = upload(c, c->ms_slot_host, (size_t)MXMOL + 1, &c->ms_slot_base))) return failed(rc);
    }
    DevTables &t = c->tables;
#define UT(field, arr) if ((rc = upload(c, arr, sizeof(arr) / sizeof(arr[0]), &t.field))) return failed(rc)
    UT(self296, MT_SELF296); UT(self260, MT_SELF260); UT(frgn296, MT_FRGN296); UT(fco2, MT_FCO2);
    UT(n2c296, MT_N2RT296_C); UT(n2sf296, MT_N2RT296_SF); UT(n2c220, MT_N2RT220_C); UT(n2sf220, MT_N2RT220_SF);
    UT(xfac_rhu, MT_XFAC_RHU); UT(xfacco2, MT_XFACCO2); UT(tdep_bandhead, MT_TDEP_BANDHEAD);
    UT(tips_qoft, TIPS_QOFT); UT(smass, ISO_SMASS); UT(tips_isonm, TIPS_ISONM); UT(tips_offset, TIPS_OFFSET);
    UT(o3ch_x, MT_O3CH_X); UT(o3ch_y, MT_O3CH_Y); UT(o3ch_z, MT_O3CH_Z); UT(o3hh0, MT_O3HH0); UT(o3hh1, MT_O3HH1);
    UT(o3hh2, MT_O3HH2); UT(o3huv, MT_O3HUV); UT(o2f_x, MT_O2F_XO2); UT(o2f_t, MT_O2F_XO2T); UT(o2inf1, MT_O2INF1);
    UT(o2inf3, MT_O2INF3); UT(o2vis, MT_O2VIS); UT(o2fuv, MT_O2FUV); UT(n2f_272, MT_N2F_272); UT(n2f_228, MT_N2F_228);
    UT(n2f_ah2o, MT_N2F_AH2O); UT(n2f1, MT_N2F1);
#undef UT
    {   // Q(296 K) of every isotopologue, interpolated exactly like Q(T)
        std::vector<double> q296(sizeof(TIPS_QOFT) / sizeof(double) / 119);
        for (size_t i = 0; i < q296.size(); i++) q296[i] = tips_atob(296., &TIPS_QOFT[i * 119]);
        if ((rc = upload(c, q296.data(), q296.size(), &t.tips_q296))) return failed(rc);
    }
    {   // log ratios of the temperature interpolations below 820 cm-1 (DevTables::lr_*), with the device's log()
        const struct { const double *t296, *tlow; const double **out; int n; } lr[3] = {
            {t.self296, t.self260, &t.lr_self, MT_SELF296_NPT},
            {t.n2c296, t.n2c220, &t.lr_n2c, MT_N2RT296_NPT},
            {t.n2sf296, t.n2sf220, &t.lr_n2sf, MT_N2RT296_NPT}};
        for (const auto &e : lr) {
            void *p = nullptr;
            if (hipMalloc(&p, sizeof(double) * (size_t)e.n) != hipSuccess) { c->err = "hipMalloc(log-ratio table) failed"; return failed(MONORTM_EHIP); }
            c->owned.push_back(p);
            launch_logratio(e.t296, e.tlow, static_cast<double *>(p), e.n, nullptr);
            *e.out = static_cast<const double *>(p);
        }
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { c->err = "log-ratio tables could not be formed"; return failed(MONORTM_EHIP); }
    }
    void *ef = nullptr;
    if (hipMalloc(&ef, sizeof(int)) != hipSuccess || hipMemset(ef, 0, sizeof(int)) != hipSuccess) { c->err = "hipMalloc(errflag) failed"; return failed(MONORTM_EHIP); }
    c->owned.push_back(ef);
    c->errflag = static_cast<int *>(ef);
    if (hipDeviceSynchronize() != hipSuccess) { c->err = "device set-up failed"; return failed(MONORTM_EHIP); }  // memsets above: done before any stream uses them
    if (hipStreamCreateWithFlags(&c->hs, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->errflag_host), sizeof(int), hipHostMallocDefault) != hipSuccess) {
        c->err = "stream / pinned flag of the host-buffer entry points could not be created";
        return failed(MONORTM_EHIP);
    }
    if (const char *e = getenv("MONORTM_HOST_TIMING")) c->host_timing = e[0] == '1';
    for (const char *k : {"lines_kernel", "nslice", "fair", "tile_waves", "far_levels", "ms_items",
#ifdef MONORTM_EXPERIMENT
                          "ms_ablate",
#endif
                         }) {
        std::string env = "MONORTM_" + std::string(k);
        for (char &ch : env) ch = (char)toupper((unsigned char)ch);
        if (const char *e = getenv(env.c_str()))
            if (set_option(c, k, e)) { c->err = env + ": " + c->err; return failed(MONORTM_EARG); }  // a mistyped switch must not run silently
    }
    if (c->host_timing && !g_timing_ctx) {  // a Fortran caller never finalizes: report at exit
        g_timing_ctx = c;
        static bool registered = false;
        if (!registered) { atexit([] { if (g_timing_ctx) print_host_timing(g_timing_ctx); }); registered = true; }
    }
    *out = c;
    return MONORTM_OK;
}

void monortm_hip_finalize(void *ctx) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return;
    if (!c->shards.empty()) {
        for (Ctx *sh : c->shards) monortm_hip_finalize(sh);
        delete c;
        return;
    }
    print_host_timing(c);
    if (g_timing_ctx == c) g_timing_ctx = nullptr;
    DeviceGuard guard;
    hipSetDevice(c->device);
    for (auto &e : c->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto &e : c->event_pool) hipEventDestroy(e);
    comm_release(c);
    if (c->hs) hipStreamDestroy(c->hs);
    if (c->errflag_host) hipHostFree(c->errflag_host);
    for (void *p : c->owned) hipFree(p);
    if (c->partial) hipFree(c->partial);
    if (c->osum) hipFree(c->osum);
    if (c->ms_scratch) hipFree(c->ms_scratch);
    for (void *p : c->xs_buf) hipFree(p);
    if (c->phys) hipFree(c->phys);
    if (c->far) hipFree(c->far);
    if (c->far_stream) hipStreamDestroy(c->far_stream);
    for (hipEvent_t e : c->far_ev)
        if (e) hipEventDestroy(e);
    c->mw_cache.release();
    for (int i = 0; i < 8; i++)
        if (c->stage[i].p) {
            if (i % 2 == 0) hipHostFree(c->stage[i].p);  // even slots: pinned host arenas
            else hipFree(c->stage[i].p);
        }
    delete c;
}

int monortm_hip_init_multi(const char *tape3_path, double v1, double v2, int icp, int real_kind, int ngpu, void **out) {
    if (!out) { g_init_error = "null output pointer"; return MONORTM_EARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_init_error = "no HIP device available: the MI355X path has no CPU fallback"; return MONORTM_EHIP; }
    // device list: MONORTM_DEVICES="0,1,..." (an ordinal may repeat: several shards then share that device, which is how
    // the sharding is exercised on a one-GPU box) or the first ngpu visible devices; ngpu <= 0 means all of them
    std::vector<int> devs;
    if (const char *e = getenv("MONORTM_DEVICES")) {
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            devs.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
        }
        if (ngpu > 0 && (int)devs.size() > ngpu) devs.resize(ngpu);
    }
    if (devs.empty()) {
        const int n = (ngpu <= 0) ? ndev : ngpu;
        if (n > ndev) { g_init_error = "ngpu = " + std::to_string(n) + " but only " + std::to_string(ndev) + " device(s) visible"; return MONORTM_EARG; }
        for (int g = 0; g < n; g++) devs.push_back(g);
    }
    for (int dv : devs)
        if (dv < 0 || dv >= ndev) { g_init_error = "MONORTM_DEVICES names device " + std::to_string(dv) + " outside 0.." + std::to_string(ndev - 1); return MONORTM_EARG; }
    DeviceGuard guard;  // monortm_hip_init selects each device in turn
    Ctx *m = new Ctx;
    m->real_kind = real_kind;
    for (int dv : devs) {
        void *sh = nullptr;
        const int rc = monortm_hip_init(tape3_path, v1, v2, icp, real_kind, dv, &sh);
        if (rc) {  // g_init_error holds the text
            for (Ctx *x : m->shards) monortm_hip_finalize(x);
            delete m;
            return rc;
        }
        m->shards.push_back(static_cast<Ctx *>(sh));
    }
    m->has_lines = m->shards[0]->has_lines;
    m->device = m->shards[0]->device;
    *out = m;
    return MONORTM_OK;
}

int monortm_hip_device_count(void *ctx) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return 0;
    return c->shards.empty() ? 1 : (int)c->shards.size();
}

int monortm_hip_tape3_probe(const char *tape3_path, double v1, double v2, long long *n_physical, long long *n_entries,
                            long long *n_coupled) {
    monortm::LineTable t;
    std::string err;
    int rc = monortm::load_tape3(tape3_path ? tape3_path : "", v1, v2, t, err);
    if (rc) {
        g_init_error = err;
        return rc;
    }
    for (int m = 0; m <= MXMOL; m++) {
        n_physical[m] = t.n_physical[m];
        n_entries[m] = (m == 0) ? (long long)t.size() : t.mol_start[m + 1] - t.mol_start[m];
        n_coupled[m] = 0;
    }
    for (size_t i = 0; i < t.meta.size(); i++)
        if ((t.meta[i] >> 10) & 3) {
            n_coupled[t.meta[i] & 63]++;
            n_coupled[0]++;
        }
    return MONORTM_OK;
}

// ---- the single output gather of a profile-sharded job (north_star), for callers without Python --------------
// RCCL is opened on first use (dlopen: a one-GPU caller never loads it, and a process that already holds a copy - torch ships
// one - keeps using that one).  Only the handful of entry points the gather needs are bound.
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Gather)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *init_rank = nullptr;
};
struct NcclId { char b[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
Rccl g_rccl;
bool rccl_open(std::string &err) {
    if (g_rccl.h) return true;
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) {
        const char *de = dlerror();  // (one call: dlerror() clears the message it returns)
        err = std::string("RCCL cannot be loaded: ") + (de ? de : "librccl.so.1 not found");
        return false;
    }
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void *)>(dlsym(g_rccl.h, "ncclGetUniqueId"));
    g_rccl.init_rank = dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(g_rccl.h, "ncclCommDestroy"));
    g_rccl.Gather = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(g_rccl.h, "ncclGather"));
    g_rccl.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(g_rccl.h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.init_rank || !g_rccl.CommDestroy || !g_rccl.Gather) {
        err = "RCCL lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGather";
        dlclose(g_rccl.h);
        g_rccl = Rccl{};
        return false;
    }
    return true;
}
void comm_release(Ctx *c) {
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}
int rccl_fail(Ctx *c, const char *what, int rc) {
    c->err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
    return MONORTM_EHIP;
}
}  // namespace

int monortm_hip_comm_unique_id(void *id128) {
    // no context yet: the reason of a failure is left where monortm_hip_last_error(NULL) finds it
    if (!id128) { g_init_error = "monortm_hip_comm_unique_id: null id buffer"; return MONORTM_EARG; }
    std::string err;
    if (!rccl_open(err)) { g_init_error = err; return MONORTM_EHIP; }
    const int rc = g_rccl.GetUniqueId(id128);
    if (rc != 0) {
        g_init_error = std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
        return MONORTM_EHIP;
    }
    return MONORTM_OK;
}

int monortm_hip_comm_init(void *ctx, int world, int rank, const void *id128) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    if (world < 1 || rank < 0 || rank >= world || !id128) { c->err = "bad world / rank / id"; return MONORTM_EARG; }
    if (!rccl_open(c->err)) return MONORTM_EHIP;
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    NcclId id;
    memcpy(id.b, id128, sizeof id.b);
    auto init = reinterpret_cast<int (*)(void **, int, NcclId, int)>(g_rccl.init_rank);
    const int rc = init(&c->comm, world, id, rank);
    if (rc != 0) { c->comm = nullptr; return rccl_fail(c, "ncclCommInitRank", rc); }
    c->comm_rank = rank;
    c->comm_world = world;
    return MONORTM_OK;
}

int monortm_hip_gather_dev(void *ctx, const void *send, size_t bytes, void *recv, int root, void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->comm) { c->err = "no communicator on this context: call monortm_hip_comm_init first"; return MONORTM_EARG; }
    if (!send || root < 0 || root >= c->comm_world || (c->comm_rank == root && !recv)) { c->err = "bad gather arguments"; return MONORTM_EARG; }
    if (int rcd = check_device(c)) return rcd;
    const int rc = g_rccl.Gather(send, recv, bytes, /* ncclUint8 */ 1, root, c->comm, (hipStream_t)stream);
    if (rc != 0) return rccl_fail(c, "ncclGather", rc);
    return MONORTM_OK;
}

int monortm_hip_set_option(void *ctx, const char *name, const char *value) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    int rc = set_option(c, name, value);
    for (Ctx *sh : c->shards)
        if (int r = set_option(sh, name, value)) rc = r;
    return rc;
}

int monortm_hip_has_lines(void *ctx) {
    Ctx *c = static_cast<Ctx *>(ctx);
    return (c && c->has_lines) ? 1 : 0;
}

int monortm_hip_xsec_regions(void *ctx) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return 0;
    if (!c->shards.empty()) c = c->shards[0];
    return c->xs.nreg;
}

int monortm_hip_kat(void *ctx, int which, int n, const double *args, const double *tab119, double *out) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) c = c->shards[0];
    if (which < 1 || which > 9 || n < 1 || !args || !out || (which == 5 && !tab119)) { c->err = "bad known-answer request"; return MONORTM_EARG; }
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    double *din = nullptr, *dtab = nullptr, *dout = nullptr;
    auto run = [&]() -> int {
        HIPCHK(c, hipMalloc(&din, sizeof(double) * 4 * n));
        HIPCHK(c, hipMalloc(&dtab, sizeof(double) * 119));
        HIPCHK(c, hipMalloc(&dout, sizeof(double) * 2 * n));
        HIPCHK(c, hipMemcpy(din, args, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
        if (tab119) HIPCHK(c, hipMemcpy(dtab, tab119, sizeof(double) * 119, hipMemcpyHostToDevice));
        launch_kat(which, n, din, dtab, dout, c->errflag, c->tables, nullptr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpy(out, dout, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
        return MONORTM_OK;
    };
    const int rc = run();
    hipFree(din); hipFree(dtab); hipFree(dout);  // (hipFree(nullptr) is a no-op: every exit path releases what was allocated)
    return rc ? rc : monortm_hip_check(c, nullptr);
}

long long monortm_hip_counter(void *ctx, int which) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return -1;
    if (!c->shards.empty()) {
        long long t = 0;
        for (Ctx *sh : c->shards) t += monortm_hip_counter(sh, which);
        return t;
    }
    if (which == 0) return c->o_reused;
    return -1;
}

long long monortm_hip_line_count(void *ctx, int mol) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c || mol < 0 || mol > MXMOL) return -1;
    if (!c->shards.empty()) c = c->shards[0];  // every device holds the same table
    return c->host.n_physical[mol];
}

int monortm_hip_profile(void *ctx, int enable) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    c->profiling = enable & 7;
    c->prof_stride = std::max(1, enable >> 8);
    for (auto &n : c->prof_calls) n = 0;
    return MONORTM_OK;
}

int monortm_hip_kernel_time(void *ctx, int kernel, double *total_ms, long long *launches) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    if (!total_ms || !launches) { c->err = "null output pointer"; return MONORTM_EARG; }
    if (kernel < 0 || kernel > 2) { c->err = "kernel id must be 0..2"; return MONORTM_EARG; }
    for (auto &e : c->events) {
        float ms = 0.f;
        HIPCHK(c, hipEventSynchronize(e.b));
        HIPCHK(c, hipEventElapsedTime(&ms, e.a, e.b));
        c->tot_ms[e.k] += ms;
        c->launches[e.k]++;
        c->event_pool.push_back(e.a);
        c->event_pool.push_back(e.b);
    }
    c->events.clear();
    *total_ms = c->tot_ms[kernel];
    *launches = c->launches[kernel];
    return MONORTM_OK;
}

int monortm_hip_check(void *ctx, void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->errflag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    return decode_flag(c, flag, (hipStream_t)stream);
}

int monortm_hip_modm_dev(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                         int nmol, const void *P, const void *T, const void *CLW, const void *WKL,
                         const void *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res,
                         int ibrd, int ixsect, void *O, void *O_BY_MOL, void *OC, void *O_CLW, const double *wn_ends,
                         void *stream) {
    if (ixsect != 0) {
        Ctx *c = static_cast<Ctx *>(ctx);
        if (!c) return null_ctx();
        c->err = "IXSECT = 1 needs the column amounts of the cross-section molecules and an ODXSEC array: call monortm_hip_modm_xs_dev "
                 "(the reference passes them through COMMON /PATHX/ and its ODXSEC argument, src/modm.f90:24,197)";
        return MONORTM_EUNSUPPORTED;
    }
    return monortm_hip_modm_xs_dev(ctx, nprof, nwn, wn, dvset, nlay, nlay_max, nmol, P, T, CLW, WKL, WBRODL, cntnm_fac, sclcpl, sclhw,
                                   y0res, ibrd, 0, nullptr, nullptr, O, O_BY_MOL, OC, O_CLW, wn_ends, stream);
}

int monortm_hip_xsec_tables(void *ctx, int nxs, int nreg, const double *reg, const double *temps, const double *pres_mb,
                            const long long *offs, const double *pool, long long npool) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) {
        for (Ctx *sh : c->shards)
            if (int rc = monortm_hip_xsec_tables(sh, nxs, nreg, reg, temps, pres_mb, offs, pool, npool)) { c->err = sh->err; return rc; }
        return MONORTM_OK;
    }
    if (nxs < 0 || nxs > 38 || nreg < 0 || npool < 0 || (nreg > 0 && (!reg || !temps || !pres_mb || !offs || !pool))) { c->err = "bad cross-section tables"; return MONORTM_EARG; }
    for (int r = 0; r < nreg; r++) {   // shapes the kernel relies on
        const int m = (int)reg[r * 8], npts = (int)reg[r * 8 + 3], nt = (int)reg[r * 8 + 4];
        if (m < 0 || m >= nxs || npts < 2 || nt < 1 || nt > 6 || !(reg[r * 8 + 2] > reg[r * 8 + 1]) || !(reg[r * 8 + 7] > reg[r * 8 + 6])) { c->err = "bad cross-section region"; return MONORTM_EARG; }
        for (int k = 0; k < nt; k++)
            if (offs[r * 6 + k] < 0 || offs[r * 6 + k] + npts > npool) { c->err = "cross-section spectrum outside the pool"; return MONORTM_EARG; }
    }
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());   // no kernel may still read the tables that are replaced
    for (void *p : c->xs_buf) hipFree(p);
    c->xs_buf.clear();
    c->xs = DevXsec{};
    auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
        void *p = nullptr;
        HIPCHK(c, hipMalloc(&p, std::max<size_t>(bytes, 8)));
        c->xs_buf.push_back(p);
        if (bytes) HIPCHK(c, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        *dst = p;
        return MONORTM_OK;
    };
    const size_t n6 = (size_t)nreg * 6;
    int rc;
    if ((rc = up(reg, (size_t)nreg * 8 * 8, reinterpret_cast<const void **>(&c->xs.reg)))) return rc;
    if ((rc = up(temps, n6 * 8, reinterpret_cast<const void **>(&c->xs.temps)))) return rc;
    if ((rc = up(pres_mb, n6 * 8, reinterpret_cast<const void **>(&c->xs.pres)))) return rc;
    if ((rc = up(offs, n6 * 8, reinterpret_cast<const void **>(&c->xs.offs)))) return rc;
    if ((rc = up(pool, (size_t)npool * 8, reinterpret_cast<const void **>(&c->xs.pool)))) return rc;
    c->xs.nxs = nxs;
    c->xs.nreg = nreg;
    return MONORTM_OK;
}

int monortm_hip_modm_xs_dev(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                            int nmol, const void *P, const void *T, const void *CLW, const void *WKL,
                            const void *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res,
                            int ibrd, int ixsect, const void *XAMNT, void *ODXSEC, void *O, void *O_BY_MOL, void *OC,
                            void *O_CLW, const double *wn_ends, void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    hipStream_t s = (hipStream_t)stream;
    if (!c->has_lines) { c->err = "this context holds no line table (created without a TAPE3 path): MODM needs one (GET_LNFL, modm.f90:187-190)"; return MONORTM_EARG; }
    if (!wn || !nlay || !P || !T || !CLW || !WKL || !WBRODL || !cntnm_fac || !O || !O_BY_MOL || !OC || !O_CLW) { c->err = "null array argument"; return MONORTM_EARG; }
    if (int rcd = check_device(c)) return rcd;
    // first / last wavenumber decide the ABSRB grid (modm.f90:180-185).  A caller that knows them passes them in
    // wn_ends (host) and the call stays asynchronous; otherwise they are fetched from device memory (one sync).
    double vends[2];
    if (wn_ends) {
        vends[0] = wn_ends[0];
        vends[1] = wn_ends[1];
    } else {
        HIPCHK(c, hipMemcpyAsync(&vends[0], wn, sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&vends[1], wn + (nwn > 0 ? nwn - 1 : 0), sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    int rc = check_modm_args(c, nprof, nwn, nlay_max, nmol, ibrd, ixsect, vends[1]);
    if (rc) return rc;
    ModmArgs a{};
    a.real_kind = c->real_kind;
    a.nprof = nprof; a.nwn = nwn; a.nlay_max = nlay_max; a.nmol = nmol; a.ibrd = ibrd;
    a.dvset = dvset; a.sclcpl = sclcpl; a.sclhw = sclhw; a.y0res = y0res;
    for (int i = 0; i < 7; i++) a.cntnm[i] = cntnm_fac[i];
    a.wn = wn; a.P = P; a.T = T; a.CLW = CLW; a.WKL = WKL; a.WBRODL = WBRODL; a.nlay = nlay;
    a.O = O; a.O_BY_MOL = O_BY_MOL; a.OC = OC; a.O_CLW = O_CLW; a.errflag = c->errflag;
    if (ixsect == 1) {
        if (!XAMNT || !ODXSEC) { c->err = "IXSECT = 1: XAMNT / ODXSEC missing"; return MONORTM_EARG; }
        if (c->xs.nxs < 1) { c->err = "IXSECT = 1: no cross-section tables on this context (monortm_hip_xsec_tables)"; return MONORTM_EARG; }
        a.XAMNT = XAMNT; a.ODXSEC = ODXSEC; a.nxs = c->xs.nxs;
    }

    const double DVABS = 1.0;
    const double V1ABS = (int)(vends[0]) - 3. * DVABS;
    const double V2ABS = (int)(vends[1] + 3. * DVABS + 0.5);
    const int NPTABS = (int)((V2ABS - V1ABS) / DVABS + 1.5);
    if (NPTABS > 5050) { c->err = "wavenumber span exceeds the 5050-point continuum grid (N_ABSRB, lblparams.f90:35)"; return MONORTM_EARG; }

    // few workgroups (single profiles): slice the line list over several blocks per (profile, layer, tile)
    int nw, wpl;  // waves per workgroup, wavenumbers per lane
    lines_config(nwn, c->real_kind, (long long)nprof * nlay_max, vends[1] - vends[0], &nw, &wpl);
    // dense grids whose far field comes from far_kernel (the physics pass below: >= 4 tiles, records fit): tiles of two waves, 256
    // wavenumbers.  The lines a tile evaluates directly are those within 1.2 of ITS half-widths, so halving the tile halves that
    // work, and the level of intervals it adds to far_kernel costs less (configs[2]: lines_kernel 2.92 -> 1.84 ms, far_kernel
    // 0.71 -> 1.06 ms).  Without far_kernel the four-wave tile with its own far field stays the better one.
    static const bool phys_off_cfg = getenv("MONORTM_NO_PHYSICS_PASS") != nullptr;
    // levels of far_kernel for tiles of tw wavenumbers - tiles, pairs, fours ... up to a half-width of ~3 cm-1 (configs[2], tiles of
    // 0.64: three levels 1.01 ms, four 1.06, two 1.63) - and the bytes of its workspace (sums, interval geometry, candidate runs)
    auto far_levels_for = [&](int tw) {
        if (nwn < 2) return 0;
        const int nt = (nwn + tw - 1) / tw;
        const double rho_tile = 0.5 * tw * (vends[1] - vends[0]) / (double)(nwn - 1);
        // no interval for which a line can be far: a far line lies >= kappa half-widths from the tile's centre AND holds the whole
        // tile inside its 25 cm-1 window, i.e. kappa rho <= 25 - rho (wide sparse channel sets: the plan, the levels and the
        // workspace would be paid for nothing - ADVICE r5).  far_kernel counts the workgroups of a molecule on an XCD in 16 bits
        // (FarPlace::cnt): a grid of more tiles than that keeps the far field inside lines_kernel
        if (MONORTM_FAR_KAPPA_HOST * rho_tile >= 25.0 - rho_tile || nt > 65535) return 0;
        int levels = 1;
        while (levels < FAR_MAXLEV && rho_tile * (double)(1 << levels) <= 3.0 && far_level_count(nt, levels - 1) > 1) levels++;
        if (c->opt.far_levels >= 0) levels = std::min(c->opt.far_levels, FAR_MAXLEV);
        return levels;
    };
    auto far_bytes_for = [&](int tw, int levels, size_t *mom_b, size_t *geom_b) {
        const int nt = (nwn + tw - 1) / tw;
        const size_t ni = (size_t)far_level_offset(nt, levels), states = (size_t)nprof * nlay_max;
        *mom_b = states * ni * nmol * FAR_MOM_STRIDE * sizeof(double);
        *geom_b = states * ni * nmol * FAR_GEOM_INTS * sizeof(int);
        return *mom_b + *geom_b + states * (size_t)nt * nmol * FAR_SEG_INTS * sizeof(int);
    };
    const size_t kFarCap = c->far_cap;   // (beyond it lines_kernel forms the far field itself)
    bool far_tiles = false;   // tiles of 128 / 256 wavenumbers chosen because far_kernel serves the grid
    if (nw == 4 && wpl == 2 && c->opt.far_levels != 0 && !phys_off_cfg && (size_t)nprof * nlay_max * c->host.size() * 48 <= c->phys_cap) {
        if (c->lines_per_cm < 0.) {   // (once per context: the table does not change)
            double vlo = 0., vhi = 0.;
            if (!c->host.vnu.empty()) {
                const auto mm = std::minmax_element(c->host.vnu.begin(), c->host.vnu.end());
                vlo = *mm.first;
                vhi = *mm.second;
            }
            c->lines_per_cm = (double)c->host.size() / std::max(vhi - vlo, 1.0);
        }
        // the smallest tile - one wave (128 wavenumbers), two, four - that still has ~1000 lines within 1.2 of its half-widths (the
        // ones it evaluates directly: below that the prologue of a tile outweighs them) and at least four tiles on the grid.
        // configs[2] (1800 lines per cm-1, 0.005 cm-1 steps), line-sum segment: 512 wavenumbers 3.75 ms, 256: 3.02, 128: 2.69
        const double dvm = (vends[1] - vends[0]) / (double)std::max(nwn - 1, 1);
        for (int cand = 1; cand <= 2; cand *= 2) {
            const int tw = 128 * cand;
            size_t mb, gb;
            const int lv = far_levels_for(tw);
            if (nwn >= 4 * tw && c->lines_per_cm * 2.4 * (0.5 * tw * dvm) >= 1000. && lv > 0 && far_bytes_for(tw, lv, &mb, &gb) <= kFarCap) { nw = cand; far_tiles = true; break; }
        }
    }
    if (c->opt.tile_waves && wpl >= 2) { nw = c->opt.tile_waves; wpl = 2; }  // measurements only: waves per workgroup of the two-wavenumber tiles
    const int NTw = 64 * nw, TW = NTw * wpl;  // lines per chunk, wavenumbers per tile
    const long long nblocks = (long long)((nwn + TW - 1) / TW) * nlay_max * nprof;
    const long long nlines = (long long)c->host.size();
    const long long cus = c->cus;  // (16 one-wave workgroups of lines_kernel are resident per compute unit: 128 VGPRs, 10 KB of LDS)
    // batches of states on a sparse channel set, double precision: G states per wave, five wavenumbers per lane
    // (lines_ms_kernel.hip) - when the batch makes whole rounds of such waves and the layout fits the LDS (decided before the line
    // slices: this kernel walks the whole list of its states)
    bool use_ms = false;
    int ms_nprof = nprof;   // profiles [0, ms_nprof) go through lines_ms_kernel (all of them unless the batch is split, below)
    MsArgs ms{};
    if (c->real_kind == 8 && nw == 1 && wpl == 1 && nlines > 0 && c->opt.lines_ms != 0 && c->opt.nslice == 0) {
        const int LPS = (nwn + MS_WPS - 1) / MS_WPS;
        int G = std::min(std::min(12, 64 / LPS), nprof);
        // slots of the molecules of this call: slot_base[nmol] pairs
        ms.nslot = c->ms_slot_host[nmol];
        // lines per chunk: three passes of 64 (state, line) items, two, or one - the largest that leaves <= 10 KB of LDS a wave
        // (16 waves per compute unit)
        for (int items = c->opt.ms_items ? c->opt.ms_items : 192; items >= 64 && !use_ms && G >= 1 && ms.nslot > 0; items -= 64) {
            const int CL = std::min(64, items / G);
            if (CL < 8) break;
            ms.G = G; ms.LPS = LPS; ms.CL = CL; ms.nsteps = (G * CL + 63) / 64;
            // records per state in LDS: the chunk + 2 read ahead, padded so that the states' arrays start 24 banks apart (a lane reads
            // 16 bytes of ITS state's record: eight states then touch eight disjoint groups of four banks)
            ms.sa_stride = CL + 2;
            while (ms.sa_stride % 8 != 3) ms.sa_stride++;
            ms.npg = (nprof + G - 1) / G;
            ms.inv_cl = (65536 + CL - 1) / CL;
            ms.inv_lps = (65536 + LPS - 1) / LPS;
            bool exact = ms.nsteps <= MS_MAXSTEPS;
            for (int ln = 0; ln < 64 && exact; ln++) exact = (int)(((unsigned)ln * (unsigned)ms.inv_lps) >> 16) == ln / LPS;
            for (int item = 0; item < ms.nsteps * 64 && exact; item++) exact = (int)(((unsigned)item * (unsigned)ms.inv_cl) >> 16) == item / CL;
            if (exact && lines_ms_lds(ms, nmol) <= 10240 - 160) use_ms = true;
        }
        const long long groups = (long long)ms.npg * nlay_max;
        if (use_ms && c->opt.lines_ms < 0) {
            // auto: a wave of lines_ms_kernel carries G states, so a batch is a few ROUNDS of such waves over the 16 wave slots of
            // every compute unit, and a round that is only part full costs half a round + half its share (measured on configs[3]'s
            // shape, 50 channels, final kernels: 128 / 256 / 320 / 384 / 512 / 640 / 768 / 1024 profiles = 0.34 / 0.67 / 0.84 / 1.0 / 1.34 /
            // 1.68 / 2.0 / 2.67 rounds take 0.69 / 0.83 / 0.98 / 1.0 / 1.63 / 1.74 / 1.96 / 2.64 times the 0.35 ms of a full round;
            // lines_kernel takes 1.26 of that per round of states whatever the batch: tools/rounds_sweep.sh).
            // From two rounds on the part-full round hides behind the others (2.0 -> 1.96, 2.67 -> 2.65).  Lists with many coupled lines
            // (their shapes go one wavenumber at a time here, Voigt pairs through a queue per wave): bench's c2lc shape at 384 profiles
            // 0.524 against 0.551 ms - a margin of 5 % instead of 20 %, so only whole rounds and large batches take this kernel.
            const double r = (double)groups / (double)(16 * cus), fr = r - std::floor(r);
            const double cost_ms = (r >= 2.0) ? r : std::floor(r) + (fr > 0.02 ? 0.5 + 0.5 * fr : 0.0);
            // What lines_kernel costs per round of lines_ms_kernel's states: 1.26 at six states a wave (50 channels).  A wave's evaluate stage
            // and prologue cost the same for four states as for nine, its prepare stage goes with the states: per state 0.055 + 0.60 / G of
            // a six-state wave's time (stage times of DESIGN 3.1m), lines_kernel the same per state whatever the channel count
            // (tools/rounds_sweep.sh with MONORTM_BENCH_CHANNELS: 64 channels = four states a wave 1.01, 40 = eight 1.42, 32 = nine 1.33).
            // Species broadening (IBRD = 1: both kernels walk the seven-species blocks in every pass - 0.472 against 0.534 ms at 384
            // profiles of bench's c4brd shape) and coupled lists (above) narrow the margin.
            double gain = std::min(1.35, 1.26 * 0.155 / (0.055 + 0.60 / (double)G));
            if (c->lc_frac > 0.02) gain *= 1.04 / 1.26;
            else if (ibrd != 0 && c->host.any_brd) gain *= 1.13 / 1.26;
            // between one and two rounds: the whole rounds through lines_ms_kernel, the rest of the profiles through lines_kernel (two
            // launches on the stream; 512 profiles of configs[3]'s shape: 384 + 128 = 0.37 + 0.16 ms against 0.60 either way)
            const long long npg_round = (16 * cus) / std::max(nlay_max, 1);   // groups of G profiles that fill the wave slots once
            const long long n_whole = (long long)G * npg_round * (long long)std::floor(r);
            const double cost_split = (r >= 1.0 && r < 2.0 && n_whole >= G && n_whole < nprof)
                                          ? std::floor(r) + gain * (double)(((nprof - n_whole + G - 1) / G) * nlay_max) / (double)(16 * cus) : 1e30;
            if (G * nwn * 10 < 64 * MS_WPS * 7) use_ms = false;
            else if (cost_split < 0.97 * std::min(cost_ms, gain * r)) ms_nprof = (int)n_whole;
            else if (cost_ms >= gain * r) use_ms = false;
        }
        if (use_ms && ms_nprof < nprof) ms.npg = (ms_nprof + G - 1) / G;
    }
    int nslice = 1;
    if (nblocks < 4 * cus && nlines >= 2 * NTw) {
        // at least ~40 lines per slice: below that the prologue of a workgroup outweighs its share of the lines
        nslice = (int)std::min<long long>(16, std::min<long long>((8 * cus + nblocks - 1) / nblocks, nlines / 40));
        if (nslice < 1) nslice = 1;
    } else {
        // long line lists: a block walks hundreds of chunks, and with only a few rounds of blocks over the chip
        // (resident: 16 one-wave or 4 four-wave blocks per CU) the last round runs half empty.  Slices of >= 32 chunks
        // until there are >= 8 rounds.
        // (a grid served by far_kernel walks a quarter of its window at most: four rounds - configs[2], one-wave tiles: three
        // slices 2.33 ms for the line-sum segment, five 2.34, seven 2.36, two 2.40, one 2.64)
        const long long resident = cus * (16 / nw), chunks = nlines / NTw;
        const long long want = ((far_tiles ? 4 : 8) * resident + nblocks - 1) / nblocks;
        nslice = (int)std::max<long long>(1, std::min<long long>(16, std::min<long long>(want, chunks / 32)));
        // a grid that fills at most five eighths of the wave slots: two slices use the rest (c4 shape, 32 profiles: 0.096 ->
        // 0.089 ms per step).  A full round is better off unsliced since the waves order themselves by progress (a.fair below):
        // 64 profiles 0.155 -> 0.127 ms, c5 0.188 -> 0.151 ms per step
        if (nslice == 1 && nblocks * nw <= 10 * cus && nlines >= 3 * NTw) nslice = 2;
    }
    if (c->opt.nslice) nslice = c->opt.nslice;  // measurements only
    if (use_ms) nslice = 1;   // (lines_ms_kernel walks the whole list of its states)
    if (nslice > 1) {
        const size_t need = (size_t)nslice * nprof * nlay_max * nmol * nwn;
        if (need > c->partial_elems) {
            if (c->partial) HIPCHK(c, hipFree(c->partial));
            c->partial = nullptr;
            c->partial_elems = 0;
            HIPCHK(c, hipMalloc(&c->partial, need * (size_t)c->real_kind));
            c->partial_elems = need;
        }
    }
    a.nslice = nslice;
    a.partial = c->partial;
    // progress-ordered wave priorities (lines_kernel.hip): grids of at most a few rounds over the 4096 wave slots that 128 VGPRs
    // leave on 256 CUs.  Re-measured in round 4 with the shorter kernels: one-wave double-precision tiles gain at every size
    // (4 / 8 / 16 rounds: -3 % / -0.6 % / -0.5 %), the four-wavenumber float tile loses at 8 rounds (configs[4] whole: 1.127
    // against 1.099 ms), multi-wave tiles lose 1 % at 8, the one-wave two-wavenumber tile gains 2 % at 8 (round 3's configs[4]
    // workload: 0.823 against 0.841 ms) - so: always for the first, up to 8 rounds for the last, up to 4 for the others
    // (round 5, after the tested sub-runs are skipped per (wave, k): the four-wavenumber float tile gains at 8 rounds as well -
    // configs[4] whole 0.992 -> 0.948 ms - so every one-wave tile up to 8 rounds)
    a.fair = ((nw == 1 && wpl == 1 && c->real_kind == 8) || nblocks * nslice * nw <= ((nw == 1) ? 8 : 4) * 16 * cus) ? 1 : 0;
    if (c->opt.fair >= 0) a.fair = c->opt.fair;  // measurements only
    static const bool mw_off = getenv("MONORTM_FINISH_GENERIC") != nullptr;  // A/B switch for measurements
    // microwave to far infrared (last wavenumber below 820 cm-1: no O3 / O2 / Rayleigh term anywhere): the fused finish kernel
    const bool mw = vends[1] < 820.0 && NPTABS <= 1000 && !mw_off;
#ifdef LINES_TIMING
    if (mw) {
#else
    if (mw && nslice == 1) {
#endif
        const size_t need = (size_t)nprof * nlay_max * nwn;
        if (need > c->osum_elems) {
            if (c->osum) HIPCHK(c, hipFree(c->osum));
            c->osum = nullptr;
            c->osum_elems = 0;
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->osum), need * sizeof(double)));
            c->osum_elems = need;
        }
        a.osum = c->osum;
    }
    Ctx::Ev ev{};
    const bool use_brd = ibrd != 0 && c->host.any_brd;
    size_t dyn = sizeof(double) * (size_t)(19 * nmol) + sizeof(int) * (size_t)(2 * nmol + 2 + 2 * FAR_SEGS * nmol);
#ifdef MONORTM_EXTRA_LDS   // occupancy experiment (tools/build_variant.sh): fewer resident workgroups per CU
    dyn += MONORTM_EXTRA_LDS;
#endif
    if (nprof > 65535) { c->err = "more than 65535 profiles in one call: split the batch"; return MONORTM_EARG; }
    dim3 grid(((nwn + TW - 1) / TW) * nslice, nprof, nlay_max);  // (tile x slice, profile, layer): see lines_kernel
    // dense grids (a line sits in the window of many tiles): its tile-independent part once per (profile, layer) - 48 B of room (32 written for an uncoupled line) per
    // (layer, line), up to an eighth of the device memory; lines_kernel then reads the record instead of forming it in every tile
    const long long ntiles = (nwn + TW - 1) / TW;
    static const bool phys_off = getenv("MONORTM_NO_PHYSICS_PASS") != nullptr;  // A/B switch for measurements
    const size_t phys_need = (size_t)nprof * nlay_max * (size_t)nlines * 48;
    prof_begin(c, s, 0, ev);
    // (a workspace a later call needs less than a quarter of - or not at all - goes back to the device: a single large dense call
    // must not pin tens of GB away from the caller's framework for the life of the context)
    const bool want_phys = ntiles >= 4 && nlines > 0 && phys_need <= c->phys_cap && !phys_off;
    if (c->phys && (!want_phys || phys_need < c->phys_bytes / 4) && c->phys_bytes > (64u << 20)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap == hipStreamCaptureStatusNone) {
            HIPCHK(c, hipDeviceSynchronize());   // (nobody reads the records any more)
            HIPCHK(c, hipFree(c->phys));
            c->phys = nullptr;
            c->phys_bytes = 0;
            if (c->far) { HIPCHK(c, hipFree(c->far)); c->far = nullptr; c->far_bytes = 0; }
        }
    }
    if (want_phys) {
        if (phys_need > c->phys_bytes) {
            if (c->phys) HIPCHK(c, hipFree(c->phys));
            c->phys = nullptr;
            c->phys_bytes = 0;
            if (hipMalloc(&c->phys, phys_need) == hipSuccess) c->phys_bytes = phys_need;
            else {  // no room for the records: lines_kernel forms them in place, as on sparse grids
                c->phys = nullptr;
                (void)hipGetLastError();
            }
        }
        if (c->phys) {
            a.phys = c->phys;
            a.phys_lines = (int)nlines;
            // multi-wave tiles (the ones that have a far field): the far lines of every tile through far_kernel
            const int levels = (wpl == 2) ? far_levels_for(TW) : 0;
            if (levels > 0) {
                size_t mom_b, geom_b;
                const size_t need = far_bytes_for(TW, levels, &mom_b, &geom_b), ni = (size_t)far_level_offset((int)ntiles, levels);
                if (need > c->far_bytes && need <= kFarCap) {
                    if (c->far) HIPCHK(c, hipFree(c->far));
                    c->far = nullptr;
                    c->far_bytes = 0;
                    if (hipMalloc(&c->far, need) == hipSuccess) {
                        c->far_bytes = need;
                        // all-ones bytes = NaN sums, -1 indices: an entry that a kernel reads without another having written it
                        // shows in the results at once instead of depending on what the allocation held (once per growth)
                        HIPCHK(c, hipMemsetAsync(c->far, 0xFF, need, s));
                    } else (void)hipGetLastError();   // no room: lines_kernel forms the far field itself
                }
                if (c->far && need <= c->far_bytes) {
                    a.farmom = static_cast<double *>(c->far);
                    a.fargeom = reinterpret_cast<int *>(static_cast<char *>(c->far) + mom_b);
                    a.farseg = reinterpret_cast<int *>(static_cast<char *>(c->far) + mom_b + geom_b);
                    a.far_levels = levels;
                    a.far_ni = (int)ni;
                    a.far_tw = TW;
                    a.far_ntile = (int)ntiles;
                }
            }
            // far_plan_kernel (a chain of dependent table reads, ~65 us whatever the grid) needs nothing physics_kernel writes: it
            // runs beside it on a stream of the context, forked from and joined to the caller's stream with two events (the pattern a
            // stream capture follows as well)
            const bool far_on = a.farmom != nullptr;
            if (far_on && !c->far_stream) {
                if (hipStreamCreateWithFlags(&c->far_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->far_ev[0], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&c->far_ev[1], hipEventDisableTiming) != hipSuccess) {
                    c->err = "stream / events of the far-field plan could not be created";
                    return MONORTM_EHIP;
                }
            }
            if (far_on) {
                HIPCHK(c, hipEventRecord(c->far_ev[0], s));
                HIPCHK(c, hipStreamWaitEvent(c->far_stream, c->far_ev[0], 0));
                launch_far_plan(a, c->lines, c->tables, c->far_stream);
                HIPCHK(c, hipEventRecord(c->far_ev[1], c->far_stream));
            }
            launch_physics(a, c->lines, c->tables, (int)nlines, use_brd, s);
            if (far_on) {
                HIPCHK(c, hipStreamWaitEvent(s, c->far_ev[1], 0));
                launch_far(a, c->lines, c->tables, 0.5 * TW * (vends[1] - vends[0]) / (double)std::max(nwn - 1, 1),
                           c->lines_per_cm > 0. ? c->lines_per_cm : 1., s);
            }
        }
    }
    if (use_ms) {
        const long long groups = (long long)ms.npg * nlay_max;
        const size_t need = lines_ms_scratch(ms, groups);
        if (need > c->ms_scratch_bytes) {
            if (c->ms_scratch) HIPCHK(c, hipFree(c->ms_scratch));
            c->ms_scratch = nullptr;
            c->ms_scratch_bytes = 0;
            HIPCHK(c, hipMalloc(&c->ms_scratch, need));
            c->ms_scratch_bytes = need;
        }
        const size_t nl4 = (std::max<size_t>(c->host.size(), 1) + 3) / 4 * 4;
        if (!c->ms_reach) {   // two bytes + one float per table line (the table does not change)
            void *p = nullptr;
            HIPCHK(c, hipMalloc(&p, 6 * nl4));
            c->owned.push_back(p);
            c->ms_reach = static_cast<unsigned short *>(p);
        }
        ms.reach = c->ms_reach;
        ms.near0 = reinterpret_cast<float *>(c->ms_reach + nl4);
        ms.scratch = c->ms_scratch;
        ms.slot_base = c->ms_slot_base;
        ms.ablate = c->opt.ms_ablate;
    }
    if (use_ms && ms_nprof < nprof) {
        // a split batch (double precision, one tile, one slice, no dense-grid workspaces): the same arguments with nprof = the whole
        // rounds for lines_ms_kernel, and shifted by those profiles for lines_kernel
        ModmArgs am = a, aw = a;
        am.nprof = ms_nprof;
        const size_t st = (size_t)ms_nprof * nlay_max;   // states ahead of the second part
        auto shift = [](const void *p, size_t elems) { return p ? static_cast<const void *>(static_cast<const double *>(p) + elems) : nullptr; };
        aw.nprof = nprof - ms_nprof;
        aw.P = shift(a.P, st); aw.T = shift(a.T, st); aw.CLW = shift(a.CLW, st); aw.WBRODL = shift(a.WBRODL, st);
        aw.WKL = shift(a.WKL, st * nmol);
        aw.nlay = a.nlay + ms_nprof;
        aw.O = const_cast<void *>(shift(a.O, st * nwn)); aw.O_CLW = const_cast<void *>(shift(a.O_CLW, st * nwn));
        aw.O_BY_MOL = const_cast<void *>(shift(a.O_BY_MOL, st * nmol * nwn));
        aw.OC = const_cast<void *>(shift(a.OC, st * MONORTM_NCONT * nwn));
        aw.osum = a.osum ? a.osum + st * nwn : nullptr;
        launch_lines_ms(am, c->lines, c->tables, ms, use_brd, s);
        launch_lines(aw, c->lines, c->tables, nw, wpl, use_brd, dim3(grid.x, (unsigned)aw.nprof, grid.z), dyn, s);
    } else if (use_ms) launch_lines_ms(a, c->lines, c->tables, ms, use_brd, s);
    else launch_lines(a, c->lines, c->tables, nw, wpl, use_brd, grid, dyn, s);
    prof_end(c, s, ev);
    HIPCHK(c, hipGetLastError());
    // finest coarse grid: 1 cm-1 (O2 A band) above 1340 cm-1, 2 cm-1 (CO2) below
    const bool high = vends[1] > 1340.0;
    const int csize = (high ? NPTABS : NPTABS / 2) + 24;
    // few workgroups (single profiles) below 1340 cm-1: the four waves of a workgroup run the passes side by side, each
    // with its own grids; a grid that fills the chip is served better by one pass after the other
    const bool par = !high && (long long)nlay_max * nprof < 16 * cus;
    const int fin_threads = par ? 256 : ((NPTABS <= 256 && nwn <= 128) ? 64 : 256);  // microwave-sized grids: one wave
    // one-wave workgroups on a grid of >= 4096 of them: four layers per wave, a 16-lane team each (continuum_kernel.hip)
    const bool quad = !high && !par && fin_threads == 64 && NPTABS <= 64;
    const int lds_sets = (par || quad) ? 4 : 1;
    const size_t lds = sizeof(double) * (size_t)(NPTABS + 4 + csize) * lds_sets;
    prof_begin(c, s, 1, ev);
    if (ixsect == 1) launch_xsec(a, c->xs, s);   // MONORTM_XSEC_SUB comes first (modm.f90:197); the finish kernel adds its sum into O
    a.slices_reduced = 0;
    if (nslice > 1 && (long long)nmol * nwn > 4096) {  // wide grids: the slice sums at full memory bandwidth
        launch_reduce_slices(a, s);
        a.slices_reduced = 1;
    }
    if (mw) {
        const hipError_t e = launch_finish_mw(a, c->tables, vends[0], vends[1], V1ABS, V2ABS, NPTABS, c->mw_cache, s);
        if (e != hipSuccess) {
            c->err = std::string("launch_finish_mw: ") + (c->mw_cache.why ? c->mw_cache.why : hipGetErrorString(e));
            return c->mw_cache.why ? MONORTM_EUNSUPPORTED : MONORTM_EHIP;
        }
    }
    else HIPCHK(c, launch_finish(a, c->tables, V1ABS, V2ABS, NPTABS, csize, high, par, fin_threads, lds, lds_sets, s));
    prof_end(c, s, ev);
    HIPCHK(c, hipGetLastError());
    return MONORTM_OK;
}

int monortm_hip_rtm_dev(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max, const int *irt,
                        int iout, const void *T, const void *TZ, const void *O, void *tmpsfc, const void *emiss,
                        const void *reflc, void *RUP, void *RDN, void *TRTOT, void *RAD, void *TB, void *TMR,
                        void *stream) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    if (!c->shards.empty()) return multi_only_host(c);
    hipStream_t s = (hipStream_t)stream;
    if (!wn || !nlay || !irt || !T || !TZ || !O || !tmpsfc || !emiss || !reflc || !RUP || !RDN || !TRTOT || !RAD || !TB) { c->err = "null array argument"; return MONORTM_EARG; }
    if (int rcd = check_device(c)) return rcd;
    if (nprof < 1 || nwn < 1 || nlay_max < 1) { c->err = "bad nprof/nwn/nlay_max"; return MONORTM_EARG; }
    RtmArgs a{};
    a.real_kind = c->real_kind;
    a.nprof = nprof; a.nwn = nwn; a.nlay_max = nlay_max; a.iout = iout;
    a.wn = wn; a.T = T; a.TZ = TZ; a.O = O; a.emiss = emiss; a.reflc = reflc; a.nlay = nlay; a.irt = irt;
    a.tmpsfc = tmpsfc; a.RUP = RUP; a.RDN = RDN; a.TRTOT = TRTOT; a.RAD = RAD; a.TB = TB; a.TMR = TMR;
    Ctx::Ev ev{};
    prof_begin(c, s, 2, ev);
    launch_rtm(a, s);
    prof_end(c, s, ev);
    HIPCHK(c, hipGetLastError());
    return MONORTM_OK;
}


int monortm_hip_modm(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                     int nmol, const void *P, const void *T, const void *CLW, const void *WKL,
                     const void *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd,
                     int ixsect, void *O, void *O_BY_MOL, void *OC, void *O_CLW) {
    if (ixsect != 0) {
        Ctx *c = static_cast<Ctx *>(ctx);
        if (!c) return null_ctx();
        c->err = "IXSECT = 1 needs the column amounts of the cross-section molecules and an ODXSEC array: call monortm_hip_modm_xs";
        return MONORTM_EUNSUPPORTED;
    }
    return monortm_hip_modm_xs(ctx, nprof, nwn, wn, dvset, nlay, nlay_max, nmol, P, T, CLW, WKL, WBRODL, cntnm_fac, sclcpl, sclhw, y0res,
                               ibrd, 0, nullptr, nullptr, O, O_BY_MOL, OC, O_CLW);
}

int monortm_hip_modm_xs(void *ctx, int nprof, int nwn, const double *wn, double dvset, const int *nlay, int nlay_max,
                        int nmol, const void *P, const void *T, const void *CLW, const void *WKL,
                        const void *WBRODL, const double *cntnm_fac, double sclcpl, double sclhw, double y0res, int ibrd,
                        int ixsect, const void *XAMNT, void *ODXSEC, void *O, void *O_BY_MOL, void *OC, void *O_CLW) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    DeviceGuard guard;
    if (c->shards.empty())
        return modm_host(c, nprof, nwn, wn, dvset, nlay, nlay_max, nmol, P, T, CLW, WKL, WBRODL, cntnm_fac, sclcpl, sclhw, y0res, ibrd,
                         ixsect, XAMNT, ODXSEC, O, O_BY_MOL, OC, O_CLW, nullptr);
    if (!nlay || !P || !T || !CLW || !WKL || !WBRODL || !O || !O_BY_MOL || !OC || !O_CLW || nprof < 1 || nwn < 1 || nlay_max < 1 || nmol < 1) {
        c->err = "bad or null argument";
        return MONORTM_EARG;
    }
    const int G = (int)c->shards.size();
    const size_t d = (size_t)c->real_kind, l = (size_t)nlay_max * d, w = (size_t)nlay_max * nwn * d;
    std::vector<std::function<int()>> fin(G);
    int rc = MONORTM_OK;
    for (int g = 0; g < G; g++) {
        int p0, n;
        shard_block(nprof, G, g, &p0, &n);
        if (n < 1) continue;
        Ctx *s = c->shards[g];
        const int nxs = s->xs.nxs;
        const int r = modm_host(s, n, nwn, wn, dvset, nlay + p0, nlay_max, nmol, off(P, p0 * l), off(T, p0 * l), off(CLW, p0 * l),
                                off(WKL, p0 * l * nmol), off(WBRODL, p0 * l), cntnm_fac, sclcpl, sclhw, y0res, ibrd, ixsect,
                                XAMNT ? off(XAMNT, p0 * l * nxs) : nullptr, ODXSEC ? off(ODXSEC, p0 * w) : nullptr, off(O, p0 * w),
                                off(O_BY_MOL, p0 * w * nmol), off(OC, p0 * w * MONORTM_NCONT), off(O_CLW, p0 * w), &fin[g]);
        if (r && !rc) { rc = r; c->err = "device " + std::to_string(s->device) + ": " + s->err; }
    }
    for (int g = 0; g < G; g++)
        if (fin[g]) {
            const int r = fin[g]();
            if (r && !rc) { rc = r; c->err = "device " + std::to_string(c->shards[g]->device) + ": " + c->shards[g]->err; }
        }
    return rc;
}

int monortm_hip_rtm(void *ctx, int nprof, int nwn, const double *wn, const int *nlay, int nlay_max, const int *irt,
                    int iout, const void *T, const void *TZ, const void *O, void *tmpsfc, const void *emiss,
                    const void *reflc, void *RUP, void *RDN, void *TRTOT, void *RAD, void *TB, void *TMR) {
    Ctx *c = static_cast<Ctx *>(ctx);
    if (!c) return null_ctx();
    DeviceGuard guard;
    if (c->shards.empty())
        return rtm_host(c, nprof, nwn, wn, nlay, nlay_max, irt, iout, T, TZ, O, tmpsfc, emiss, reflc, RUP, RDN, TRTOT, RAD, TB, TMR, nullptr);
    if (!nlay || !irt || !T || !TZ || !O || !tmpsfc || !emiss || !reflc || !RUP || !RDN || !TRTOT || !RAD || !TB || nprof < 1 || nwn < 1 ||
        nlay_max < 1) {
        c->err = "bad or null argument";
        return MONORTM_EARG;
    }
    const int G = (int)c->shards.size();
    const size_t d = (size_t)c->real_kind, l = (size_t)nlay_max * d, w = (size_t)nlay_max * nwn * d, v = (size_t)nwn * d;
    std::vector<std::function<int()>> fin(G);
    int rc = MONORTM_OK;
    for (int g = 0; g < G; g++) {
        int p0, n;
        shard_block(nprof, G, g, &p0, &n);
        if (n < 1) continue;
        Ctx *s = c->shards[g];
        const int r = rtm_host(s, n, nwn, wn, nlay + p0, nlay_max, irt + p0, iout, off(T, p0 * l), off(TZ, p0 * (l + d)), off(O, p0 * w),
                               off(tmpsfc, p0 * d), off(emiss, p0 * v), off(reflc, p0 * v), off(RUP, p0 * v), off(RDN, p0 * v),
                               off(TRTOT, p0 * v), off(RAD, p0 * v), off(TB, p0 * v), TMR ? off(TMR, p0 * v) : nullptr, &fin[g]);
        if (r && !rc) { rc = r; c->err = "device " + std::to_string(s->device) + ": " + s->err; }
    }
    for (int g = 0; g < G; g++)
        if (fin[g]) {
            const int r = fin[g]();
            if (r && !rc) { rc = r; c->err = "device " + std::to_string(c->shards[g]->device) + ": " + c->shards[g]->err; }
        }
    return rc;
}

}  // extern "C"
