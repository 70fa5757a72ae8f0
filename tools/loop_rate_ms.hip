// loop_rate_ms.hip - what the five-wavenumber loops of lines_ms_asm.hpp deliver against the one-wavenumber loops of lines_asm.hpp
// (measurement tool, not part of the library; LABNOTES round 6 reads it).
//   hipcc --offload-arch=gfx950 -O3 -I monortm_amd/csrc tools/loop_rate_ms.hip -o monortm_amd/lib/loop_rate_ms && monortm_amd/lib/loop_rate_ms
// Every workgroup is one wave.  "bcast": the lanes read ONE record array (lines_kernel<double,1,1>: one state per wave, a lane =
// a channel).  "lane6": six groups of ten lanes read six arrays STRIDE records apart (six states per wave) with the same
// one-wavenumber loops (a record read serves one evaluation).  "ms": the same six arrays, five wavenumbers per lane
// (ms_run_k0).  Reported: cycles per (line, wavenumber of a lane) of one wave, and (state, line, channel) evaluations per 1000
// cycles and SIMD, counting 50 useful lanes of 64 for bcast and 60 for the six-state layouts.  The sums are checked against
// the host's arithmetic.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lines_asm.hpp"
#include "lines_ms_asm.hpp"

struct RecA { double xnu, hw2, a2, pa; };
constexpr int NL = 32, G = 6, LPS = 10;

__host__ __device__ inline RecA make_rec(int g, int j) {
    const double hw = 0.02 + 0.001 * j + 0.003 * g, a2 = 1e-3 * (1 + 0.1 * g) * hw;
    return RecA{1.0 + 0.31 * j + 0.01 * g, hw * hw, a2, a2 / (625. + hw * hw)};
}
__host__ __device__ inline double make_wn(int lane, int k, int mode) {
    const int c = (mode == 0) ? lane : (lane % LPS) + LPS * k;   // channel of the state
    return 0.4 + 0.57 * c;
}

// mode 0 bcast, 1 lane6 (one-wavenumber loops, five passes), 2 ms
template <int MODE>
__global__ __launch_bounds__(64, 4) void walk(double *out, long long *cyc, int reps, unsigned long long T, unsigned long long M, int stride) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RecA *rec = reinterpret_cast<RecA *>(lds);
    const int lane = threadIdx.x;
    for (int g = 0; g < G; g++)
        for (int j = lane; j < NL + 2; j += 64) rec[g * stride + j] = make_rec(g, j < NL ? j : 0);
    __syncthreads();
    const int g = (MODE == 0) ? 0 : min(lane / LPS, G - 1);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(void *)&rec[g * stride];
    double W[5], S[5] = {0., 0., 0., 0., 0.};
    for (int k = 0; k < 5; k++) W[k] = make_wn(lane, k, MODE);
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        if constexpr (MODE == 2) {
            unsigned addr = base;
            int n = __builtin_amdgcn_readfirstlane(NL);
            unsigned long long Mc = M;
            asm volatile("" : "+s"(Mc));
            unsigned long long R[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};   // (every slot within reach: no block is skipped)
            asm volatile("" : "+s"(R[0]), "+s"(R[1]), "+s"(R[2]), "+s"(R[3]), "+s"(R[4]));
            ms_run_k0(addr, n, Mc, R, W, S);
        } else {
#pragma unroll 1
            for (int k = 0; k < (MODE == 0 ? 1 : 5); k++) {
                unsigned addr = base;
                int n = __builtin_amdgcn_readfirstlane(NL);
                unsigned long long Tc = T, Mc = M;
                asm volatile("" : "+s"(Tc), "+s"(Mc));
                double w = W[0], s = S[0];
                if (MODE == 1) { w = k == 0 ? W[0] : k == 1 ? W[1] : k == 2 ? W[2] : k == 3 ? W[3] : W[4]; s = k == 0 ? S[0] : k == 1 ? S[1] : k == 2 ? S[2] : k == 3 ? S[3] : S[4]; }
                asm_run<0, 0u>(addr, n, Tc, Mc, w, s);
                if (MODE == 0) S[0] = s;
                else { if (k == 0) S[0] = s; else if (k == 1) S[1] = s; else if (k == 2) S[2] = s; else if (k == 3) S[3] = s; else S[4] = s; }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int k = 0; k < 5; k++) out[((size_t)blockIdx.x * 5 + k) * 64 + lane] = S[k];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

static double host_sum(int g, double wn, unsigned long long M, int reps) {
    double s = 0.;
    for (int r = 0; r < reps; r++)
        for (int j = 0; j < NL; j++) {
            const RecA h = make_rec(g, j);
            const bool two = ((M >> (j & ~1)) & 3ull) != 0ull;   // a pair takes the class of the more general of its lines
            const double d = wn - h.xnu, den1 = d * d + h.hw2;
            s += fmax(h.a2 / den1 - h.pa, 0.);
            if (two) {
                const double dp = wn + h.xnu, den2 = dp * dp + h.hw2;
                s += fmax(h.a2 / den2 - h.pa, 0.);
            }
        }
    return s;
}

template <int MODE>
static void run(const char *name, int cus, unsigned long long T, unsigned long long M, const char *cls, int stride, double *out, long long *cyc) {
    const int reps = 1000;
    const size_t rec_bytes = sizeof(RecA) * (size_t)(G * stride + 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(walk<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int occ = 1; occ <= 4; occ += 3) {
        size_t lds = (size_t)(160 * 1024) / (4 * occ) - 1024;
        if (lds < rec_bytes) lds = rec_bytes;
        const int wgs = cus * 4 * occ;
        hipLaunchKernelGGL(walk<MODE>, dim3(wgs), dim3(64), lds, 0, out, cyc, 3, T, M, stride);
        hipDeviceSynchronize();
        // correctness of the 3-rep warm run
        std::vector<double> h(5 * 64);
        hipMemcpy(h.data(), out, sizeof(double) * 5 * 64, hipMemcpyDeviceToHost);
        double worst = 0.;
        for (int lane = 0; lane < (MODE == 0 ? 64 : G * LPS); lane++)
            for (int k = 0; k < (MODE == 0 ? 1 : 5); k++) {
                const int g = (MODE == 0) ? 0 : lane / LPS;
                const double ref = host_sum(g, make_wn(lane, k, MODE), M, 3), got = h[k * 64 + lane];
                worst = fmax(worst, fabs(got - ref) / fmax(fabs(ref), 1e-300));
            }
        hipLaunchKernelGGL(walk<MODE>, dim3(wgs), dim3(64), lds, 0, out, cyc, reps, T, M, stride);
        hipDeviceSynchronize();
        std::vector<long long> c(wgs);
        hipMemcpy(c.data(), cyc, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
        double mean = 0.;
        for (long long v : c) mean += (double)v;
        mean /= wgs;
        const int nwl = (MODE == 0) ? 1 : 5;
        const double per = mean / ((double)reps * NL * nwl);             // cycles per (line, wavenumber of a lane), one wave
        const double useful = (MODE == 0) ? 50. : 60.;                    // channels x states in the 64 lanes
        printf("%-6s %-12s stride %3d  waves/SIMD %d  cycles per (line, wn) %7.2f   evals per 1000 cycles and SIMD %8.1f   worst rel err %.1e\n", name, cls,
               stride, occ, per, 1000.0 * occ / per * useful, worst);
    }
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double *out;
    long long *cyc;
    hipMalloc(&out, sizeof(double) * 5 * 64 * cus * 16);
    hipMalloc(&cyc, sizeof(long long) * cus * 16);
    struct Cls { const char *name; unsigned long long T, M; };
    const Cls cls[] = {{"one-res", ~0ull, 0ull}, {"two-res", ~0ull, ~0ull}, {"mixed", ~0ull, 0x00ff00ff0f0f3333ull}};
    for (const Cls &c : cls) {
        run<0>("bcast", cus, c.T, c.M, c.name, 34, out, cyc);
        run<1>("lane6", cus, c.T, c.M, c.name, 34, out, cyc);
        run<1>("lane6", cus, c.T, c.M, c.name, 35, out, cyc);
        run<2>("ms", cus, c.T, c.M, c.name, 34, out, cyc);
        run<2>("ms", cus, c.T, c.M, c.name, 35, out, cyc);
    }
    return 0;
}
