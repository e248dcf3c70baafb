// What the matrix cores sustain on f32 / f64 16x16x4 MFMAs from registers alone (no memory): the ceiling the dense-gate
// kernels (csrc/dq_dense.hip) are measured against.  hipcc --offload-arch=gfx950 -O3 mb_mfma.hip -o mb_mfma && ./mb_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k64(double* out, int iters) {
    f64x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    void* buf; hipMalloc(&buf, 1 << 26);
    const int iters = 20000, blocks = 256 * 4;     // 4 workgroups (16 waves) per CU
    for (int wg : {1, 2, 4}) {
        const int nb = 256 * wg;
        float ms = timeit([&] { hipLaunchKernelGGL(k32<8>, dim3(nb), dim3(256), 0, 0, (float*)buf, iters); });
        double flop = 2.0 * 16 * 16 * 4 * 8.0 * iters * (double)nb * 4;
        printf("f32 16x16x4, 8 accumulators, %d waves/SIMD: %.1f TFLOP/s (%.3f of 157.3)\n", wg, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
        ms = timeit([&] { hipLaunchKernelGGL(k64<8>, dim3(nb), dim3(256), 0, 0, (double*)buf, iters); });
        printf("f64 16x16x4, 8 accumulators, %d waves/SIMD: %.1f TFLOP/s (%.3f of 78.6)\n", wg, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
    }
    (void)blocks;
    return 0;
}
