// What one MFMA wave per SIMD can sustain on this chip, apart from any kernel design (round 4): the persistent conv kernels hold
// 0.92-0.94 PFLOP/s on the 64->64 / 128->128 layers with ONE consumer (matrix) wave per SIMD, whatever is done to the nest's schedule.
// Back-to-back v_mfma_f32_32x32x16_bf16 on four independent accumulators, operands in registers: the pipe's ceiling at the clock the chip
// holds under that load, with 1 wave per SIMD (256 threads) and 2 (512 threads), 256 workgroups.  Measured: 2,008 / 2,170 TFLOP/s (0.80 /
// 0.87 of the 2.5 PFLOP/s the 2.4 GHz peak assumes).  (The nest WITH its LDS reads, barriers and stores: ws_interference.hip.)     build: hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pure_mfma(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 18; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kernel, int threads, size_t lds, int iters, float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), lds, 0, out, iters);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double mfmas = 256.0 * (threads / 64) * (double)iters * 72.0;
        if (rep == 2) printf("%-44s %7.3f ms  %7.1f TFLOP/s  (%.3f of 2.5 PFLOP/s)\n", name, ms, mfmas * 32768.0 / (ms * 1e-3) / 1e12, mfmas * 32768.0 / (ms * 1e-3) / 2.5e15);
    }
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 6000;   // 72 MFMAs per iteration and wave: ~7 ms of matrix work at one wave per SIMD
    run("A pure MFMA, 1 wave / SIMD", pure_mfma<256>, 256, 0, iters, out);
    run("A pure MFMA, 2 waves / SIMD", pure_mfma<512>, 512, 0, iters / 2, out);
    return 0;
}
