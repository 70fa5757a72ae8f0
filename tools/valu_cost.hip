// valu_cost.hip - issue cost (shader cycles per wave64 instruction, one wave per SIMD) of the FP64 instructions the
// line-sum loop is made of, and of wave-uniform ("broadcast") LDS reads.  Measurement tool, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o monortm_amd/lib/valu_cost && monortm_amd/lib/valu_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ __launch_bounds__(64) void probe(double *out, long long *cyc, int iters, double seed) {
    __shared__ double lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = seed + i;
    __syncthreads();
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 1e-9;
    int idx = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) {  // v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (OP == 1) {  // v_rcp_f64
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                         "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 2) {  // v_add_f64
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == 3) {  // v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                         "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (OP == 4) {  // v_cmp_f64 + 2 v_cndmask_b32 (the "live" select), 4 per trip
            a0 = (a0 > c) ? a0 : 0.; asm volatile("" : "+v"(a0));
            a1 = (a1 > c) ? a1 : 0.; asm volatile("" : "+v"(a1));
            a2 = (a2 > c) ? a2 : 0.; asm volatile("" : "+v"(a2));
            a3 = (a3 > c) ? a3 : 0.; asm volatile("" : "+v"(a3));
        } else if (OP == 5) {  // wave-uniform ds_read_b128 x 8 (address in a VGPR, same for all lanes)
            double2 r0, r1, r2, r3, r4, r5, r6, r7;
            const int ad = (idx & 15) * 128;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(ad));
            a0 += r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y;
            idx++;
        } else if (OP == 6) {  // the same reads with the 8 VALU adds only (baseline to subtract) - no LDS
            a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
            idx++;
        } else if (OP == 8) {  // v_cmp_gt_f64 alone
            asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %2, %1\n v_cmp_gt_f64 vcc, %3, %1\n v_cmp_gt_f64 vcc, %4, %1\n"
                         "v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %2, %1\n v_cmp_gt_f64 vcc, %3, %1\n v_cmp_gt_f64 vcc, %4, %1\n"
                         : : "v"(a0), "v"(c), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if (OP == 9) {  // v_cndmask_b32 alone (vcc set once)
            int x0 = (int)a0, x1 = (int)a1, x2 = (int)a2, x3 = (int)a3;
            asm volatile("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc\n"
                         "v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
            a0 += x0; a1 += x1; a2 += x2; a3 += x3;
        } else if (OP == 10) {  // v_cmp_gt_u32
            int x0 = (int)a0, x1 = (int)a1;
            asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %0\n v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %0\n"
                         "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %0\n v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %0\n"
                         : : "v"(x0), "v"(x1) : "vcc");
        } else if (OP == 11) {  // v_cvt_f32_f64
            float f0, f1, f2, f3;
            asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                         "v_cvt_f32_f64 %0, %5\n v_cvt_f32_f64 %1, %6\n v_cvt_f32_f64 %2, %7\n v_cvt_f32_f64 %3, %4\n"
                         : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            a4 += f0 + f1 + f2 + f3;
        } else if (OP == 12) {  // v_cmp_gt_u64
            asm volatile("v_cmp_gt_u64 vcc, %0, %1\n v_cmp_gt_u64 vcc, %2, %1\n v_cmp_gt_u64 vcc, %3, %1\n v_cmp_gt_u64 vcc, %4, %1\n"
                         "v_cmp_gt_u64 vcc, %0, %1\n v_cmp_gt_u64 vcc, %2, %1\n v_cmp_gt_u64 vcc, %3, %1\n v_cmp_gt_u64 vcc, %4, %1\n"
                         : : "v"(a0), "v"(c), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else if (OP == 13) {  // v_cmp_gt_f32
            float f0 = (float)a0, f1 = (float)a1;
            asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %0\n v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %0\n"
                         "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %0\n v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %0\n"
                         : : "v"(f0), "v"(f1) : "vcc");
        } else if (OP == 14) {  // v_max_f64 (candidate for branch-free clamps)
            asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                         "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (OP == 7) {  // v_rcp_f32
            float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            a0 = f0; a1 = f1; a2 = f2; a3 = f3;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + idx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int per_iter, int waves_per_simd) {
    const int nblk = 256 * 4 * waves_per_simd, iters = 4096;
    double *out;
    long long *cyc;
    hipMalloc(&out, nblk * 64 * sizeof(double));
    hipMalloc(&cyc, nblk * sizeof(long long));
    hipLaunchKernelGGL(probe<OP>, dim3(nblk), dim3(64), 0, 0, out, cyc, iters, 1.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<OP>, dim3(nblk), dim3(64), 0, 0, out, cyc, iters, 1.5);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= nblk;
    printf("%-28s waves/SIMD %d: %7.2f counter ticks per instruction per wave, kernel %.3f ms -> %.2f ns per instr per SIMD\n", name,
           waves_per_simd, avg / ((double)iters * per_iter), ms, ms * 1e6 / ((double)iters * per_iter * waves_per_simd));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {4}) {
        run<0>("v_fma_f64", 8, w);
        run<2>("v_add_f64", 8, w);
        run<3>("v_mul_f64", 8, w);
        run<1>("v_rcp_f64", 8, w);
        run<7>("v_rcp_f32", 8, w);
        run<4>("v_cmp_f64+v_cndmask (pair)", 4, w);
        run<5>("ds_read_b128 uniform (+add)", 8, w);
        run<6>("(the 8 adds alone)", 8, w);
        run<8>("v_cmp_gt_f64", 8, w);
        run<9>("v_cndmask_b32", 8, w);
        run<10>("v_cmp_gt_u32", 8, w);
        run<13>("v_cmp_gt_f32", 8, w);
        run<12>("v_cmp_gt_u64", 8, w);
        run<11>("v_cvt_f32_f64", 8, w);
        run<14>("v_max_f64", 8, w);
    }
    return 0;
}
