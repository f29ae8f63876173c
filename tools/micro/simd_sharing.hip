// What two waves on ONE SIMD cost each other (round 4, after the two-consumer-team experiment: with the MFMA phase of one team and the
// epilogue of the other on the same four SIMDs, each phase took 1.4-1.8 x as long as alone).  512-thread workgroups, one per CU: waves 0-3
// (one per SIMD) run activity A, waves 4-7 (their SIMD partners) activity B, each for a fixed amount of its own work, timed per wave
// with s_memtime; a run with the partner idle gives the wave's time alone.
//   A: M = back-to-back v_mfma_f32_32x32x16_bf16 (four independent accumulators)   L = M with one conflict-free ds_read_b128 per MFMA
//   PRIO 1: the partner wave (B) at s_setprio 3; 2: the MFMA wave
//   B: V = dependent-free v_fma_f32 on eight chains   W = ds_write_b128 stream   G = 16-byte global loads (L2-resident)   S = 16-byte global stores
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/simd_sharing.hip -o /tmp/simd_sharing
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// A: 0 idle, 1 MFMA, 2 MFMA + LDS reads.  B: 0 idle, 1 VALU, 2 LDS writes, 3 global loads, 4 global stores
template <int A, int B, int PRIO = 0>
__global__ __launch_bounds__(512) void k(long long* times, float* out, const u32x4* src, u32x4* dst, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    float sink = 0.f;
    if (wave < 4) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        if (A) {
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            const char* base = smem + wave * 8192 + lane * 16;
            t0 = wall_clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 18; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (A == 2) b = *reinterpret_cast<const bf16x8*>(base + ((u * 4 + t) & 7) * 1024);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                        // A = 3 / 4 / 5: the wave steps back from the issue stage for 16 / 24 / 32 cycles behind every MFMA (s_nop: no
                        // instruction of this wave is presented to the arbiter meanwhile) — does the partner get the VALU port then?
                        if (A == 3) asm volatile("s_nop 15");
                        if (A == 4) asm volatile("s_nop 15\n\ts_nop 7");
                        if (A == 5) asm volatile("s_nop 15\n\ts_nop 15");
                    }
            }
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sink += acc[t][r];
            t1 = wall_clock64();
        }
    } else {
        const int tid = threadIdx.x - 256;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);   // the partner outranks the MFMA wave at the issue arbiter
        if (B == 1) {
            float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
            t0 = wall_clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 36; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            }
            for (int j = 0; j < 8; ++j) sink += v[j];
            t1 = wall_clock64();
        } else if (B == 2) {
            u32x4 r = {(unsigned)tid, 1u, 2u, 3u};
            char* base = smem + 32768 + tid * 16;
            t0 = wall_clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { *reinterpret_cast<u32x4*>(base + (j & 7) * 4096) = r; r.x += 1; }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            t1 = wall_clock64();
            sink = (float)r.x;
        } else if (B == 3) {
            u32x4 r[8];
            const u32x4* p = src + (size_t)blockIdx.x * 65536 + tid;
            t0 = wall_clock64();
            unsigned x = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = p[((it * 8 + j) & 255) * 256];
#pragma unroll
                for (int j = 0; j < 8; ++j) x ^= r[j].x;
            }
            t1 = wall_clock64();
            sink = (float)x;
        } else if (B == 4) {
            u32x4 r = {(unsigned)tid, 1u, 2u, 3u};
            u32x4* p = dst + (size_t)blockIdx.x * 65536 + tid;
            t0 = wall_clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { p[((it * 8 + j) & 255) * 256] = r; r.x += 1; }
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
            t1 = wall_clock64();
            sink = (float)r.x;
        }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
    if (lane == 0) times[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <class K>
static void run(const char* name, K kernel, int iters, long long* times, float* out, const u32x4* src, u32x4* dst) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kernel, dim3(256), dim3(512), 64 * 1024, 0, times, out, src, dst, iters);
    (void)hipDeviceSynchronize();
    long long h[256 * 8];
    (void)hipMemcpy(h, times, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int w = 0; w < 256; ++w) for (int v = 0; v < 8; ++v) (v < 4 ? a : b) += (double)h[w * 8 + v];
    printf("%-34s A %8.1f us   B %8.1f us\n", name, a / 1024 / 100.0, b / 1024 / 100.0);   // 100 MHz counter
}

int main() {
    long long* times; float* out; u32x4 *src, *dst;
    (void)hipMalloc(&times, 256 * 8 * 8); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&src, (size_t)256 * 65536 * 16 + 4096); (void)hipMalloc(&dst, (size_t)256 * 65536 * 16 + 4096);
    (void)hipMemset(src, 1, (size_t)256 * 65536 * 16);
    const int it = 1000;
    run("M alone", k<1, 0>, it, times, out, src, dst);
    run("L (MFMA + LDS reads) alone", k<2, 0>, it, times, out, src, dst);
    run("V (VALU) alone", k<0, 1>, it, times, out, src, dst);
    run("W (LDS writes) alone", k<0, 2>, it, times, out, src, dst);
    run("G (global loads) alone", k<0, 3>, it, times, out, src, dst);
    run("S (global stores) alone", k<0, 4>, it, times, out, src, dst);
    run("M + V", k<1, 1>, it, times, out, src, dst);
    run("M + W", k<1, 2>, it, times, out, src, dst);
    run("M + G", k<1, 3>, it, times, out, src, dst);
    run("M + S", k<1, 4>, it, times, out, src, dst);
    run("L + V", k<2, 1>, it, times, out, src, dst);
    run("L + W", k<2, 2>, it, times, out, src, dst);
    run("L + G", k<2, 3>, it, times, out, src, dst);
    run("L + S", k<2, 4>, it, times, out, src, dst);
    run("M + V, V at priority 3", k<1, 1, 1>, it, times, out, src, dst);
    run("M + G, G at priority 3", k<1, 3, 1>, it, times, out, src, dst);
    run("M + S, S at priority 3", k<1, 4, 1>, it, times, out, src, dst);
    run("L + V, V at priority 3", k<2, 1, 1>, it, times, out, src, dst);
    run("L + G, G at priority 3", k<2, 3, 1>, it, times, out, src, dst);
    run("L + S, S at priority 3", k<2, 4, 1>, it, times, out, src, dst);
    run("M + V, M at priority 3", k<1, 1, 2>, it, times, out, src, dst);
    run("M/nop16 alone", k<3, 0>, it, times, out, src, dst);
    run("M/nop24 alone", k<4, 0>, it, times, out, src, dst);
    run("M/nop32 alone", k<5, 0>, it, times, out, src, dst);
    run("M/nop16 + V", k<3, 1>, it, times, out, src, dst);
    run("M/nop24 + V", k<4, 1>, it, times, out, src, dst);
    run("M/nop32 + V", k<5, 1>, it, times, out, src, dst);
    run("M/nop24 + G", k<4, 3>, it, times, out, src, dst);
    run("M/nop24 + S", k<4, 4>, it, times, out, src, dst);
    run("M/nop24 + V, V at priority 3", k<4, 1, 1>, it, times, out, src, dst);
    return 0;
}
