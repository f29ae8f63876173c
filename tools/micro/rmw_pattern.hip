// Micro-benchmark: does the lane -> address pattern of the conv epilogue (16 B per lane, neighbouring lanes 64+ B apart)
// cost bandwidth against a lane-contiguous pattern?  256 WGs x 4 waves, every wave read-modify-writes 8 KiB per item.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rmw_pattern tools/micro/rmw_pattern.hip ; run: /tmp/rmw_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: contiguous (lane i -> chunk k*64 + i); 1: stride-1 conv, 32 channels (lane (n, half), instr s2 -> chunk n*4 + s2*2 + half);
// 2: transposed conv, 32 channels (pixel 2n + px); 3: stride-1 conv with 64-channel records (chunk n*8 + nt*4 + s2*2 + half)
template <int MODE, bool RMW>
__global__ __launch_bounds__(256) void k(u32x4* buf, long items) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        u32x4* base = buf + (it * 4 + wave) * 512;  // 8 KiB per wave per item
        int idx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) idx[j] = j * 64 + lane;
            else if (MODE == 1) idx[j] = (j >> 1) * 128 + n * 4 + (j & 1) * 2 + half;
            else if (MODE == 2) idx[j] = (j >> 2) * 256 + (2 * n + ((j >> 1) & 1)) * 4 + (j & 1) * 2 + half;
            else idx[j] = (j >> 2) * 256 + n * 8 + ((j >> 1) & 1) * 4 + (j & 1) * 2 + half;
        }
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = RMW ? base[idx[j]] : u32x4{(unsigned)it, 1u, 2u, 3u};
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j].x += 1; base[idx[j]] = v[j]; }
    }
}

template <int MODE, bool RMW>
void run(u32x4* buf, long items, const char* name) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, RMW>), dim3(256), dim3(256), 0, 0, buf, items);
    CHECK(hipEventRecord(a, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, RMW>), dim3(256), dim3(256), 0, 0, buf, items);
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)items * 32768 * (RMW ? 2 : 1);
    printf("%-28s %s  %.1f us  %.2f TB/s\n", name, RMW ? "rmw  " : "store", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const long items = 3221;  // 227*227*32 pixels * 64 B / 32 KiB
    u32x4* buf; CHECK(hipMalloc(&buf, items * 32768)); CHECK(hipMemset(buf, 0, items * 32768));
    run<0, true>(buf, items, "contiguous");
    run<1, true>(buf, items, "s1 32ch (now)");
    run<2, true>(buf, items, "up 32ch (now)");
    run<3, true>(buf, items, "s1 64ch (now)");
    run<0, false>(buf, items, "contiguous");
    run<1, false>(buf, items, "s1 32ch (now)");
    run<2, false>(buf, items, "up 32ch (now)");
    run<3, false>(buf, items, "s1 64ch (now)");
    return 0;
}
