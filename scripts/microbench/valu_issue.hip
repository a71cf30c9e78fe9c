#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* t, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f pa = {a, a}, pb = {b, -b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {   // 8 scalar fma
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (MODE == 1) {   // 4 pk fma (same flops)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (MODE == 2) {   // 8 scalar add
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (MODE == 3) {   // 4 pk add
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (MODE == 4) {   // 4 pk mul with op_sel swizzle + neg
                asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]\n v_pk_mul_f32 %1, %1, %4 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %4 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %4 op_sel:[1,0] op_sel_hi:[0,1]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    __shared__ unsigned long long tmin, tmax;
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, t0); atomicMax(&tmax, t1); }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[MODE] = tmax - tmin;
}
int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 4 << 20); hipMalloc(&t, 64);
    for (int wg : {64, 256, 512, 1024}) {   // 256 threads = 1 wave/SIMD, 512 threads = 2 waves/SIMD
        hipLaunchKernelGGL(k<0>, dim3(64), dim3(wg), 0, 0, out, t, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<1>, dim3(64), dim3(wg), 0, 0, out, t, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<2>, dim3(64), dim3(wg), 0, 0, out, t, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<3>, dim3(64), dim3(wg), 0, 0, out, t, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<4>, dim3(64), dim3(wg), 0, 0, out, t, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        const double ops = 64.0 * REP * ((wg + 255) / 256);   // scalar-equivalent float ops per lane
        printf("threads/WG %d (waves/SIMD x256): WG span / (float-ops per lane x waves per SIMD): fma %.2f  pk_fma %.2f  add %.2f  pk_add %.2f  pk_mul(op_sel) %.2f\n", wg,
               h[0] / ops, h[1] / ops, h[2] / ops, h[3] / ops, h[4] / ops);
    }
    return 0;
}
