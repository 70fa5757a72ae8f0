// loop_rate.hip - how fast ONE wave walks the hand-written class loops of lines_asm.hpp, and what 1 .. 4 such waves per SIMD
// deliver together (measurement tool, not part of the library; DESIGN.md section 5 reads it).
//   hipcc --offload-arch=gfx950 -O3 -I monortm_amd/csrc tools/loop_rate.hip -o monortm_amd/lib/loop_rate && monortm_amd/lib/loop_rate
// Every workgroup is one wave with 64 prepared line records in LDS (as lines_kernel leaves them) and calls asm_run() on them
// `reps` times for a given pair of class masks; the number of resident waves per SIMD is set through the dynamic LDS size.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lines_asm.hpp"

struct RecA { double xnu, hw2, a2, pa; };
struct RecB { double pb, d100, c1, gp1; };

template <int KIND>
__global__ __launch_bounds__(64, 4) void walk(double *out, long long *cyc, int reps, unsigned long long T, unsigned long long M) {
    __shared__ struct { RecA a[66]; RecB b[66]; } rec;
    extern __shared__ double pad_[];
    const int lane = threadIdx.x;
    rec.a[lane] = RecA{2.0 + 0.3 * lane, 1e-4 + 1e-6 * lane, 1e-3, 1e-9};
    rec.b[lane] = RecB{1e-9, -1., 0., 1.};
    if (lane < 2) { rec.a[64 + lane] = rec.a[0]; rec.b[64 + lane] = rec.b[0]; }
    __syncthreads();
    const double WN = 1.5 + 0.4 * lane;
    double SF = 0.;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(void *)&rec.a[0];
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        unsigned addr = base;
        int n = __builtin_amdgcn_readfirstlane(64);
        unsigned long long Tc = T, Mc = M;
        asm volatile("" : "+s"(Tc), "+s"(Mc));
        asm_run<KIND, (unsigned)sizeof(rec.a)>(addr, n, Tc, Mc, WN, SF);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = SF + pad_[0] * 0.;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int reps = 2000;
    double *out;
    long long *cyc;
    hipMalloc(&out, sizeof(double) * 64 * cus * 16);
    hipMalloc(&cyc, sizeof(long long) * cus * 16);
    hipFuncSetAttribute(reinterpret_cast<const void *>(walk<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cls { const char *name; unsigned long long T, M; int valu_per_4; };
    const Cls cls[] = {{"one resonance, untested", 0ull, 0ull, 23},
                       {"one resonance, tested  ", ~0ull, 0ull, 29},
                       {"two resonances, untested", 0ull, ~0ull, 53},
                       {"two resonances, tested ", ~0ull, ~0ull, 53}};
    printf("class                      waves/SIMD  cycles per line (one wave)  lines per 1000 cycles and SIMD  VALU issue share\n");
    for (const Cls &c : cls)
        for (int occ = 1; occ <= 4; occ++) {
            // 4 * occ workgroups per CU: the dynamic LDS leaves room for exactly that many
            const size_t lds = (size_t)(160 * 1024) / (4 * occ) - 4608;
            const int wgs = cus * 4 * occ;
            hipLaunchKernelGGL(walk<0>, dim3(wgs), dim3(64), lds, 0, out, cyc, 10, c.T, c.M);  // warm
            hipLaunchKernelGGL(walk<0>, dim3(wgs), dim3(64), lds, 0, out, cyc, reps, c.T, c.M);
            hipDeviceSynchronize();
            std::vector<long long> h(wgs);
            hipMemcpy(h.data(), cyc, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
            double mean = 0.;
            for (long long v : h) mean += (double)v;
            mean /= wgs;
            const double per_line = mean / (reps * 64.0);
            printf("%s  %d           %8.2f                    %8.2f                        %.2f\n", c.name, occ, per_line, 1000.0 * occ / per_line,
                   occ * (c.valu_per_4 / 4.0 * 4.0 + 0.5 * 12.0) / per_line);
        }
    return 0;
}
