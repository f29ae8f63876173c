// What does a conv launch's BYTE MOVEMENT cost by itself (round 4, the experiment DESIGN 7.R4 asks for first)?  The 32->32 stride-1 conv at
// side 227, batch 32 (7,424 tiles of 8 x 32 pixels, 64-byte NHWC pixel records): per item a workgroup fetches a 10 x 34 pixel patch
// (21,760 bytes: ten row segments of 2,176 bytes) and stores an 8 x 32 tile (16,384 bytes: eight row segments of 2,048 bytes).  This
// kernel does exactly that and nothing else — producer half of the workgroup: global -> registers -> LDS one item ahead; consumer half:
// LDS -> global; one barrier per item, the band walk of conv3x3_ws — and varies what the conv kernel cannot vary cheaply:
//   PATTERN 0 the conv's addresses, 1 the same bytes as ONE contiguous block per item (a tile-major activation layout)
//   THREADS / workgroups per CU: 512 x 1 (the conv kernel), 512 x 2, 256 x 2, 128 x 4 (4-row tiles)
//   delay: the consumers sleep `delay` x 64 cycles per item before they store (stand-in for the MFMA phase: 1.5 us = 47 at 2 GHz)
//   DIRECT: no roles, no LDS: every thread loads its chunks of the patch, then stores its chunks of the tile
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/conv_bytes.hip -o /tmp/conv_bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { int N, H, W, tiles_x, tiles_y, n_tiles; };

template <int TR>
__device__ __forceinline__ void tile_of(const Shape& s, int tile, int& n, int& ty, int& tx) {
    const int per = s.tiles_x * s.tiles_y;
    n = tile / per;
    const int r = tile - n * per;
    ty = r / s.tiles_x;
    tx = r - ty * s.tiles_x;
}

template <int THREADS, int TR, int PATTERN, bool DIRECT>
__global__ __launch_bounds__(THREADS) void mover(const u32x4* __restrict__ in, u32x4* __restrict__ out, Shape s, int delay) {
    constexpr int P = DIRECT ? THREADS : THREADS / 2;           // loading threads
    constexpr int PCH = (TR + 2) * 34 * 4, TCH = TR * 32 * 4;   // 16-byte chunks of a patch / a tile
    constexpr int NL = (PCH + P - 1) / P, NS = (TCH + P - 1) / P;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the band walk of conv3x3_ws: the workgroups of one XCD (blockIdx & 7) share a contiguous eighth of the tile list
    const int band_len = (s.n_tiles + 7) >> 3;
    const int first = (blockIdx.x & 7) * band_len + (blockIdx.x >> 3), step = gridDim.x >> 3;
    const int last = min(s.n_tiles, ((int)(blockIdx.x & 7) + 1) * band_len);
    const size_t in_chunks = (size_t)s.N * s.H * s.W * 4;

    auto src_of = [&](int tile, int q) -> size_t {
        if (PATTERN == 1) return ((size_t)tile * PCH + q) % in_chunks;
        int n, ty, tx;
        tile_of<TR>(s, tile, n, ty, tx);
        const int row = q / 136, rem = q - row * 136;
        const int y = min(max(ty * TR - 1 + row, 0), s.H - 1), x = min(max(tx * 32 - 1 + (rem >> 2), 0), s.W - 1);
        return ((size_t)(n * s.H + y) * s.W + x) * 4 + (rem & 3);
    };
    auto dst_of = [&](int tile, int c, bool& ok) -> size_t {
        if (PATTERN == 1) { ok = true; return ((size_t)tile * TCH + c) % in_chunks; }
        int n, ty, tx;
        tile_of<TR>(s, tile, n, ty, tx);
        const int row = c >> 7, rem = c & 127;
        const int y = ty * TR + row, x = tx * 32 + (rem >> 2);
        ok = y < s.H && x < s.W;
        return ((size_t)(n * s.H + y) * s.W + x) * 4 + (rem & 3);
    };

    if (DIRECT) {
        unsigned keep = 0;
        for (int tile = first; tile < last; tile += step) {
            u32x4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { const int q = min((int)threadIdx.x + j * P, PCH - 1); v[j] = in[src_of(tile, q)]; }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int c = threadIdx.x + j * P;
                bool ok; const size_t d = dst_of(tile, min(c, TCH - 1), ok);
                if (ok && c < TCH) out[d] = v[j];
            }
#pragma unroll
            for (int j = NS; j < NL; ++j) keep ^= v[j].x;
        }
        if (keep == 0x12345678u) out[0].x = keep;
        return;
    }

    const bool producer = threadIdx.x >= P;
    const int t = producer ? threadIdx.x - P : threadIdx.x;
    u32x4 v[NL];
    if (producer) {
        if (first < last) {
#pragma unroll
            for (int j = 0; j < NL; ++j) v[j] = in[src_of(first, min(t + j * P, PCH - 1))];
#pragma unroll
            for (int j = 0; j < NL; ++j) if (t + j * P < PCH) *reinterpret_cast<u32x4*>(smem + (size_t)(t + j * P) * 16) = v[j];
        }
        if (first + step < last)
#pragma unroll
            for (int j = 0; j < NL; ++j) v[j] = in[src_of(first + step, min(t + j * P, PCH - 1))];
    }
    __syncthreads();
    int it = 0;
    for (int tile = first; tile < last; tile += step, ++it) {
        if (producer) {
            // commit item it + 1 (fetched during item it - 1 ... it), then fetch item it + 2
            if (tile + step < last) {
                char* buf = smem + ((it + 1) & 1) * (PCH * 16);
#pragma unroll
                for (int j = 0; j < NL; ++j) if (t + j * P < PCH) *reinterpret_cast<u32x4*>(buf + (size_t)(t + j * P) * 16) = v[j];
            }
            if (tile + 2 * step < last)
#pragma unroll
                for (int j = 0; j < NL; ++j) v[j] = in[src_of(tile + 2 * step, min(t + j * P, PCH - 1))];
        } else {
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
            const char* buf = smem + (it & 1) * (PCH * 16);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int c = t + j * P;
                if (c < TCH) {
                    bool ok; const size_t d = dst_of(tile, c, ok);
                    const int row = c >> 7, rem = c & 127;
                    const u32x4 val = *reinterpret_cast<const u32x4*>(buf + (size_t)(((row + 1) * 34 + (rem >> 2) + 1) * 4 + (rem & 3)) * 16);
                    if (ok) out[d] = val;
                }
            }
        }
        __syncthreads();
    }
}

template <int THREADS, int TR, int PATTERN, bool DIRECT>
static void run(const char* name, int wg_per_cu, int delay, const u32x4* in, u32x4* out, int N, int H, int W) {
    Shape s{N, H, W, (W + 31) / 32, (H + TR - 1) / TR, 0};
    s.n_tiles = N * s.tiles_x * s.tiles_y;
    const int grid = 256 * wg_per_cu;
    const size_t lds = DIRECT ? 0 : 2 * (TR + 2) * 34 * 64;
    auto kern = mover<THREADS, TR, PATTERN, DIRECT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, 0, in, out, s, delay);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, 0, in, out, s, delay);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    const double bytes = (double)s.n_tiles * ((TR + 2) * 34 * 64 + TR * 32 * 64);   // requested at the L1 level (edge tiles included)
    const double tensor = 2.0 * N * H * W * 64;                                       // one read + one write of the tensor
    printf("%-58s grid %4d x %3d  delay %3d  %7.1f us/launch  %5.2f us/item  L1-level %5.2f TB/s  tensor-level %5.2f TB/s\n", name, grid, THREADS, delay,
           us, us * grid / s.n_tiles, bytes / us * 1e-6, tensor / us * 1e-6);
}

int main() {
    const int N = 32, H = 227, W = 227;
    const size_t bytes = (size_t)N * H * W * 64;
    u32x4 *in, *out;
    CK(hipMalloc(&in, bytes + (1 << 20))); CK(hipMalloc(&out, bytes + (1 << 20)));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    for (int delay : {0, 24, 47}) {
        run<512, 8, 0, false>("conv pattern, 4 + 4 waves, 1 workgroup per CU", 1, delay, in, out, N, H, W);
        run<512, 8, 1, false>("contiguous blocks, 4 + 4 waves, 1 workgroup per CU", 1, delay, in, out, N, H, W);
        run<512, 8, 0, false>("conv pattern, 4 + 4 waves, 2 workgroups per CU", 2, delay, in, out, N, H, W);
        run<512, 8, 1, false>("contiguous blocks, 4 + 4 waves, 2 workgroups per CU", 2, delay, in, out, N, H, W);
        run<256, 8, 0, false>("conv pattern, 2 + 2 waves, 2 workgroups per CU", 2, delay, in, out, N, H, W);
        run<256, 8, 1, false>("contiguous blocks, 2 + 2 waves, 2 workgroups per CU", 2, delay, in, out, N, H, W);
        run<256, 8, 0, false>("conv pattern, 2 + 2 waves, 3 workgroups per CU", 3, delay, in, out, N, H, W);
        run<128, 4, 0, false>("conv pattern (4-row tiles), 1 + 1 waves, 4 workgroups per CU", 4, delay, in, out, N, H, W);
        run<128, 4, 1, false>("contiguous blocks (4-row tiles), 1 + 1 waves, 4 per CU", 4, delay, in, out, N, H, W);
    }
    run<512, 8, 0, true>("direct (no LDS, no roles), conv pattern, 1 per CU", 1, 0, in, out, N, H, W);
    run<512, 8, 1, true>("direct, contiguous, 1 per CU", 1, 0, in, out, N, H, W);
    run<512, 8, 0, true>("direct, conv pattern, 2 per CU", 2, 0, in, out, N, H, W);
    run<256, 8, 0, true>("direct, conv pattern, 256 threads, 4 per CU", 4, 0, in, out, N, H, W);
    run<256, 8, 1, true>("direct, contiguous, 256 threads, 4 per CU", 4, 0, in, out, N, H, W);
    run<256, 8, 0, true>("direct, conv pattern, 256 threads, 8 per CU", 8, 0, in, out, N, H, W);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
