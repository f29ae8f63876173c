// Which companion activity slows a consumer wave's MFMA phase (round 4)?  The stride-1 nest's shape (72 MFMAs + 60 LDS fragment reads
// per item) runs at the full matrix rate by itself (mfma_ceiling.hip: 2.05 PFLOP/s at one wave per SIMD); inside conv3x3_ws the same nest
// runs at 55-60 % of that.  Here the nest runs in waves 0-3 of a 512-thread workgroup (one per SIMD) with one workgroup barrier per item,
// and waves 4-7 play the producers with one ingredient at a time:
//   0 nothing (they only meet the barrier)      1 + 6 x ds_write_b128 per thread and item (the patch commit)
//   2 + ~300 VALU operations per item            3 + 6 x 16-byte global loads per thread and item (the fetch), consumed next item
//   4 all of 1-3                                 5 all of 1-3 and the consumers store 8 x 16 bytes per lane every second item (the epilogue)
//   6 / 7 / 8: as 0 with ONE barrier per 2 / 3 / 4 items
//   12 / 13: TWO consumer teams (768 threads, three waves per SIMD) taking turns: in the interval in which one team runs the nest, the other
//   issues the four stores of its previous item (12), after ~150 VALU operations (13: the epilogue's arithmetic); producers as in 4.
//   To be read against 10 (one team: nest, then four stores, every item).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__((MODE >= 12 ? 768 : 512), (MODE >= 12 ? 3 : 2)) void k(float* out, const u32x4* src, u32x4* dst, int iters, size_t src_elems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = MODE >= 12 ? 768 : 512;
    for (int i = threadIdx.x; i < 120 * 1024 / 4; i += NTHR) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, hw_wave = threadIdx.x >> 6, wave = hw_wave & 3, team = MODE >= 12 ? hw_wave >> 2 : 0;
    if (hw_wave < NTHR / 64 - 4) {
        f32x16 acc[2][2];
        for (int g = 0; g < 2; ++g) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (MODE < 6 || MODE > 8 || (it % (MODE - 4)) == 0) __syncthreads();   // MODE 6 / 7 / 8: one barrier per 2 / 3 / 4 items
            if (MODE >= 12 && (it & 1) != team) {   // the other team's nest: this team's previous item leaves
                if (it > 0) {
                    if (MODE == 13) {
#pragma unroll
                        for (int r = 0; r < 9; ++r)
#pragma unroll
                            for (int j = 0; j < 16; ++j) acc[0][0][j] = fmaf(acc[0][0][j], 1.0001f, 0.5f);
                    }
                    u32x4* o = dst + ((size_t)blockIdx.x * 64 + (it >> 1) % 64) * 2048 + (team * 1024 + wave * 64 + lane);
                    for (int s = 0; s < 4; ++s)
                        o[s * 256] = u32x4{__float_as_uint(acc[0][0][s]), __float_as_uint(acc[0][1][s]), __float_as_uint(acc[1][0][s]), __float_as_uint(acc[1][1][s])};
                }
                continue;
            }
            // the conv kernel's conflict-free layout: 64-byte pixel records, 16-byte chunk index XOR-ed with (column >> 2) & 3
            const int col = lane & 31, half = lane >> 5;
            const char* xbase = smem + (it & 1) * 22 * 1024;
            const char* wbase = smem + 44 * 1024;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int rec = wave * 2 * 34 + col + kx, key = ((col + kx) >> 2) & 3;
                    const char* xb = xbase + rec * 64 + ((((ks << 1) | half) ^ key) << 4);
                    const char* wb = wbase + col * 64 + ((((ks << 1) | half) ^ ((col >> 2) & 3)) << 4);
                    bf16x8 xf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) xf[r] = *reinterpret_cast<const bf16x8*>(xb + r * (34 * 64));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + ((ky * 3 + kx) * 64 + nt * 32) * 64);
#pragma unroll
                            for (int g = 0; g < 2; ++g) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[g + ky], acc[g][nt], 0, 0, 0);
                        }
                }
            // the epilogue's stores: 8 x 16 bytes per lane (a wave instruction = 1 KB contiguous).  5: all eight every second item, every
            // workgroup; 9: the same from ONE workgroup only (no chip-wide burst); 10: four every item; 11: one of the eight after every 9th MFMA group
            const bool st5 = (MODE == 5 && (it & 1)) || (MODE == 9 && (it & 1) && blockIdx.x == 0);
            if (st5 || MODE == 10) {
                u32x4* o = dst + ((size_t)blockIdx.x * 64 + (it >> 1) % 64) * 2048 + (wave * 64 + lane);
                const int n_st = MODE == 10 ? 4 : 8, s0 = MODE == 10 ? (it & 1) * 4 : 0;
                for (int s = s0; s < s0 + n_st; ++s)
                    o[s * 256] = u32x4{__float_as_uint(acc[0][0][s]), __float_as_uint(acc[0][1][s]), __float_as_uint(acc[1][0][s]), __float_as_uint(acc[1][1][s])};
            }
        }
        float s = 0;
        for (int g = 0; g < 2; ++g) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[g][t][r];
        out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        const int tid = threadIdx.x - (NTHR - 256);
        u32x4 regs[6];
        for (int j = 0; j < 6; ++j) regs[j] = u32x4{(unsigned)tid, (unsigned)j, 0u, 0u};
        float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
        size_t cursor = ((size_t)blockIdx.x * 977 + tid) % (src_elems - 6 * 4096);
        for (int it = 0; it < iters; ++it) {
            char* lb = smem + ((it + 1) & 1) * 22 * 1024;
            if (MODE == 1 || MODE == 4 || MODE == 5 || MODE >= 9) {
#pragma unroll
                for (int j = 0; j < 6; ++j) if ((tid >> 2) + 64 * j < 340) *reinterpret_cast<u32x4*>(lb + ((tid >> 2) + 64 * j) * 64 + (tid & 3) * 16) = regs[j];
            }
            if (MODE == 2 || MODE == 4 || MODE == 5 || MODE >= 9) {
#pragma unroll
                for (int r = 0; r < 38; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            }
            if (MODE == 3 || MODE == 4 || MODE == 5 || MODE >= 9) {
#pragma unroll
                for (int j = 0; j < 6; ++j) regs[j] = src[cursor + (size_t)j * 4096];
                cursor = (cursor + 256 * 6 * 97) % (src_elems - 6 * 4096);
            }
            if (MODE < 6 || MODE > 8 || (it % (MODE - 4)) == 0) __syncthreads();
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += v[j];
        for (int j = 0; j < 6; ++j) s += (float)regs[j][0];
        out[(size_t)blockIdx.x * 256 + tid + 131072] = s;
    }
}

template <class K>
static void run(const char* name, K kernel, int iters, float* out, const u32x4* src, u32x4* dst, size_t n, int threads = 512) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 120 * 1024, 0, out, src, dst, iters, n);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double mfmas = 256.0 * 4 * (double)iters * 72.0;
        if (rep == 2) printf("%-58s %7.3f ms  %6.3f us/item  %7.1f TFLOP/s (%.3f of peak)\n", name, ms, ms * 1e3 / iters, mfmas * 32768.0 / (ms * 1e-3) / 1e12, mfmas * 32768.0 / (ms * 1e-3) / 2.5e15);
    }
}

int main() {
    float* out; u32x4 *src, *dst;
    const size_t n = (size_t)64 << 20;   // 1 GiB of 16-byte elements to fetch from
    (void)hipMalloc(&out, 4 * 131072 * 2); /* consumers' sums: 256 workgroups x 512; producers' behind them */ (void)hipMalloc(&src, n * 16); (void)hipMalloc(&dst, (size_t)256 * 64 * 2048 * 16);
    (void)hipMemset(src, 1, n * 16);
    const int iters = 2000;
    run("0 consumers + barrier per item only", k<0>, iters, out, src, dst, n);
    run("1 + producers write the patch to LDS", k<1>, iters, out, src, dst, n);
    run("2 + producers run ~300 VALU per item", k<2>, iters, out, src, dst, n);
    run("3 + producers fetch 6 x 16 B per thread per item (HBM)", k<3>, iters, out, src, dst, n);
    run("4 all three", k<4>, iters, out, src, dst, n);
    run("5 all three + consumers store a tile every 2nd item", k<5>, iters, out, src, dst, n);
    run("9 as 5, but ONE workgroup stores (no chip-wide burst)", k<9>, iters, out, src, dst, n);
    run("10 as 5, four stores EVERY item instead of eight every 2nd", k<10>, iters, out, src, dst, n);
    run("6 as 0, ONE barrier per TWO items", k<6>, iters, out, src, dst, n);
    run("7 as 0, one barrier per three items", k<7>, iters, out, src, dst, n);
    run("8 as 0, one barrier per four items", k<8>, iters, out, src, dst, n);
    run("12 two consumer teams taking turns: nest | four stores", k<12>, iters, out, src, dst, n, 768);
    run("13 ... nest | ~150 VALU + four stores", k<13>, iters, out, src, dst, n, 768);
    return 0;
}
