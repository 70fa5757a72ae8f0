// pk_count.hip - does SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 count a packed instruction (v_pk_fma_f32 = two lanes' worth of FP32 FMA
// per lane) once or twice?  (bench.py prices the f32 work of the single-precision line sum from these counters.)
//   hipcc --offload-arch=gfx950 -O2 -o monortm_amd/lib/pk_count tools/pk_count.hip
//   rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU -d gpurun_out/pk -- monortm_amd/lib/pk_count
// Four kernels, each one workgroup of 64 lanes x 256 workgroups, each wave issuing exactly 4096 instructions of ONE opcode:
//   k_fma_f32 (v_fma_f32), k_pk_fma_f32 (v_pk_fma_f32), k_pk_mul_f32 (v_pk_mul_f32), k_pk_add_f32 (v_pk_add_f32).
// Per wave the counter reads 4096 if a packed instruction counts once, 8192 if it counts per FP32 operation.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k_fma_f32(float *out) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-7f;
    for (int i = 0; i < 256; i++) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ void k_pk_fma_f32(float *out) {
    f2 a = {threadIdx.x * 1e-3f, 1.f}, b = {1.0001f, 0.9999f}, c = {1e-7f, 1e-7f};
    for (int i = 0; i < 256; i++) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y;
}
__global__ void k_pk_mul_f32(float *out) {
    f2 a = {threadIdx.x * 1e-3f, 1.f}, b = {1.0001f, 0.9999f};
    for (int i = 0; i < 256; i++) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y;
}
__global__ void k_pk_add_f32(float *out) {
    f2 a = {threadIdx.x * 1e-3f, 1.f}, b = {1e-7f, -1e-7f};
    for (int i = 0; i < 256; i++) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y;
}

int main() {
    float *d;
    if (hipMalloc(&d, 256 * 64 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    hipLaunchKernelGGL(k_fma_f32, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k_pk_fma_f32, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k_pk_mul_f32, dim3(256), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k_pk_add_f32, dim3(256), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    printf("pk_count: 4 kernels x 256 waves x 4096 instructions of one opcode each: %s\n", hipGetErrorString(e));
    hipFree(d);
    return e == hipSuccess ? 0 : 1;
}
