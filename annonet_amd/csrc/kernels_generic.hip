// kernels_generic.hip — shape-agnostic HIP kernels for gfx950.
//
// These carry (a) the fp32 parity mode, where every convolution output is ONE k-ordered fmaf chain
// (taps row-major, reduction channels innermost) so that inference is bit-exact against the oracle, and
// (b) every layer shape the MFMA kernels (kernels_mfma.hip) do not cover: the 3-channel stem, the K-channel
// head, widths that are not multiples of 32.  Wavefronts are 64 lanes: one lane = one output pixel, one
// wave = 64 consecutive pixels x 8 output channels, so weight addresses are wave-uniform (scalar loads) and
// activation loads are 16/32-byte vectors per lane.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "bnacc.h"
#include "common.h"
#include "kernels.h"

namespace anh {
namespace {

typedef __bf16 bf16;

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// In ANH_BF16 mode a conv consumes bf16 operands (the MFMA A fragment), so the value produced by the bn+relu(+skip)
// prologue is rounded to bf16 here too: the generic and the MFMA kernels then see identical inputs.
template <typename T> __device__ __forceinline__ float operand_round(float v) { return v; }
template <> __device__ __forceinline__ float operand_round<bf16>(float v) { return (float)(bf16)v; }

__device__ __forceinline__ float relu_affine(float y, float s, float t) {
    const float z = fmaf(y, s, t);
    return z > 0.f ? z : 0.f;
}

__device__ __forceinline__ double wave_sum(double v) {  // fixed shuffle tree: deterministic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float v[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
// eight storage elements kept RAW in registers (a prefetched value costs 4 VGPRs in bf16), converted on use
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<bf16> { uint4 a; };
__device__ __forceinline__ Raw8<float> raw_load8(const float* p) { return Raw8<float>{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
__device__ __forceinline__ Raw8<bf16> raw_load8(const bf16* p) { return Raw8<bf16>{*reinterpret_cast<const uint4*>(p)}; }
__device__ __forceinline__ void raw_to_float(const Raw8<float>& r, float v[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void raw_to_float(const Raw8<bf16>& r, float v[8]) {
    const unsigned w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float v[8]) {
    bf16 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(t);
}

// ---- reading one input value / eight consecutive channels through the producer's bn+relu ----
template <typename T, int KIND>
__device__ __forceinline__ float fetch1(const Src& s, int n, int y, int x, int h, int w, int c_total, int c) {
    if (KIND == SRC_IMAGE) {
        const int sy = min(max(s.win_top(n) + y, 0), s.img_h - 1);
        const int sx = min(max(s.win_left(n) + x, 0), s.img_w - 1);
        const uint8_t v = s.img[(size_t)n * s.img_sample_stride + ((size_t)sy * s.img_w + sx) * c_total + c];
        return (float)v * (1.0f / 256.0f);  // dlib input<>::to_tensor
    }
    const size_t i = (((size_t)n * h + y) * w + x) * c_total + c;
    const float a = to_f<T>(reinterpret_cast<const T*>(s.a)[i]);
    if (KIND == SRC_RAW) return a;
    float v = relu_affine(a, s.a_scale[c], s.a_shift[c]);
    if (KIND == SRC_ACT2) v += relu_affine(to_f<T>(reinterpret_cast<const T*>(s.b)[i]), s.b_scale[c], s.b_shift[c]);
    return operand_round<T>(v);
}

template <typename T, int KIND>
__device__ __forceinline__ void fetch8(const Src& s, size_t pixel, int c_total, int c8, float v[8]) {
    const size_t i = pixel * c_total + c8;
    load8<T>(reinterpret_cast<const T*>(s.a) + i, v);
    if (KIND == SRC_RAW) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = relu_affine(v[j], s.a_scale[c8 + j], s.a_shift[c8 + j]);
    if (KIND == SRC_ACT2) {
        float u[8];
        load8<T>(reinterpret_cast<const T*>(s.b) + i, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += relu_affine(u[j], s.b_scale[c8 + j], s.b_shift[c8 + j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = operand_round<T>(v[j]);
}

__device__ __forceinline__ bool tap_source(int o, int kk, int stride, int pad, int gather, int limit, int& src) {
    if (gather == 0) { src = o * stride + kk - pad; }
    else {
        const int t = o + pad - kk;
        if (t < 0 || (t % stride) != 0) { src = 0; return false; }
        src = t / stride;
    }
    return src >= 0 && src < limit;
}

// ---------------------------------------------------------------------------------------------------
// conv_generic: block = 64 pixels x (blockDim.y groups of 8 output channels)
// ---------------------------------------------------------------------------------------------------
template <typename TIN, typename TOUT, int KIND>
__global__ __launch_bounds__(256) void conv_generic_kernel(ConvArgs a) {
    const int64_t total = (int64_t)a.n * a.h_out * a.w_out;
    const int64_t p_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int co0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * blockDim.y + threadIdx.y) * 8);
    if (co0 >= a.c_out) return;
    const bool live = p_raw < total;
    const int64_t p = live ? p_raw : total - 1;
    const int plane = a.h_out * a.w_out;
    const int n = (int)(p / plane);
    const int rem = (int)(p - (int64_t)n * plane);
    const int oy = rem / a.w_out, ox = rem - oy * a.w_out;

    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    const bool vec_in = KIND != SRC_IMAGE && (a.c_red % 8) == 0;
    const bool vec_w = (a.c_out % 8) == 0;

    for (int ky = 0; ky < a.k; ++ky) {
        int iy;
        const bool vy = tap_source(oy, ky, a.stride, a.pad, a.gather, a.h_in, iy);
        for (int kx = 0; kx < a.k; ++kx) {
            int ix;
            const bool valid = tap_source(ox, kx, a.stride, a.pad, a.gather, a.w_in, ix) && vy;
            if (!__any(valid)) continue;  // adding fmaf(0,w,acc) is the identity, so skipping is exact
            const float* wt = a.w_f32 + (size_t)(ky * a.k + kx) * a.c_red * a.c_out + co0;
            const size_t pixel = valid ? ((size_t)n * a.h_in + iy) * a.w_in + ix : 0;
            if (vec_in) {
                for (int c8 = 0; c8 < a.c_red; c8 += 8) {
                    float v[8];
                    if (valid) fetch8<TIN, KIND>(a.src, pixel, a.c_red, c8, v);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float* wr = wt + (size_t)(c8 + j) * a.c_out;
                        if (vec_w) {
                            const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
                            acc[0] = fmaf(v[j], w0.x, acc[0]); acc[1] = fmaf(v[j], w0.y, acc[1]);
                            acc[2] = fmaf(v[j], w0.z, acc[2]); acc[3] = fmaf(v[j], w0.w, acc[3]);
                            acc[4] = fmaf(v[j], w1.x, acc[4]); acc[5] = fmaf(v[j], w1.y, acc[5]);
                            acc[6] = fmaf(v[j], w1.z, acc[6]); acc[7] = fmaf(v[j], w1.w, acc[7]);
                        } else {
#pragma unroll
                            for (int o = 0; o < 8; ++o)
                                if (co0 + o < a.c_out) acc[o] = fmaf(v[j], wr[o], acc[o]);
                        }
                    }
                }
            } else {
                for (int c = 0; c < a.c_red; ++c) {
                    const float v = valid ? fetch1<TIN, KIND>(a.src, n, iy, ix, a.h_in, a.w_in, a.c_red, c) : 0.f;
                    const float* wr = wt + (size_t)c * a.c_out;
                    if (vec_w) {
                        const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
                        acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                        acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                        acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                        acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
                    } else {
#pragma unroll
                        for (int o = 0; o < 8; ++o)
                            if (co0 + o < a.c_out) acc[o] = fmaf(v, wr[o], acc[o]);
                    }
                }
            }
        }
    }
    if (!live) return;
    if (a.bias) {
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (co0 + o < a.c_out) acc[o] = acc[o] + a.bias[co0 + o];
    }
    if (a.out_nchw) {
        float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (co0 + o < a.c_out) out[(((size_t)n * a.c_out + co0 + o) * a.h_out + oy) * a.w_out + ox] = acc[o];
        return;
    }
    const size_t base = (size_t)p * a.c_out + co0;
#pragma unroll
    for (int dst = 0; dst < 2; ++dst) {
        TOUT* out = reinterpret_cast<TOUT*>(dst == 0 ? a.out : a.out2);
        const int accumulate = dst == 0 ? a.out_accumulate : a.out2_accumulate;
        if (!out) continue;
        if (vec_w) {
            float r[8];
            if (accumulate) {
                load8<TOUT>(out + base, r);
#pragma unroll
                for (int o = 0; o < 8; ++o) r[o] += acc[o];
            } else {
#pragma unroll
                for (int o = 0; o < 8; ++o) r[o] = acc[o];
            }
            store8<TOUT>(out + base, r);
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (co0 + o < a.c_out) {
                    const float r = accumulate ? to_f<TOUT>(out[base + o]) + acc[o] : acc[o];
                    out[base + o] = from_f<TOUT>(r);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// conv_f32_mfma: the fp32 parity mode on the matrix cores.  v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain
// (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product, no wider accumulation: MI355X_MICROARCH.md § Matrix cores),
// so chaining it over (tap row-major, input channel innermost, two channels per instruction) reproduces conv_generic's — and
// the oracle's — sums exactly, at the fp32 MATRIX rate instead of one fmaf per lane and output.
//   one wave = 32 consecutive output pixels of one output row x NT*32 output channels; any con / cont geometry (the tap ->
//   source pixel rule is conv_generic's); A = weights (row = output channel, k = channel parity = lane >> 5), B = pixels.
//   Operands come straight from global memory: a lane fetches 8 consecutive channels of ITS pixel (32 B, through the producer's
//   bn + relu (+ skip add)) per four instructions and one weight dword per instruction (32 output channels contiguous); an MFMA
//   holds the pipe for 64 cycles, so 4+ waves per SIMD hide those L1 / L2 round trips without LDS staging.
//   Padding taps, channels beyond c_red and output channels beyond c_out multiply by an exact 0 (fma(w, 0, acc) = acc: acc is
//   never -0 because every chain starts at +0), so the result is the same as skipping them.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// exact zeroing without a select: a "cond ? loaded : 0" makes hipcc branch around the LOAD and wait for it on the spot
// (cdna_hip_programming.md §5, trap (c)); an integer AND keeps every load unconditional and in flight together
__device__ __forceinline__ float and_mask(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

template <int KIND, int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(ConvArgs a, int tiles_x) {
    const int lane = threadIdx.x & 63, j = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_tiles = (int64_t)a.n * a.h_out * tiles_x;
    if (tile >= n_tiles) return;   // wave-uniform; the kernel has no barrier
    const int tx = (int)(tile % tiles_x);
    const int64_t row = tile / tiles_x;
    const int oy = (int)(row % a.h_out), n = (int)(row / a.h_out);
    // con: 32 consecutive output columns.  cont (transposed, stride s): 32 columns of ONE residue class mod s — a tap is then valid
    // for every lane or for none (t = ox + pad - kx must be a multiple of s), so no instruction multiplies structural zeros
    const int xs = a.gather ? a.stride : 1;
    const int ox = a.gather ? ((tx / xs) * 32 + j) * xs + (tx % xs) : tx * 32 + j;
    const bool px_live = ox < a.w_out;
    const int co_base = blockIdx.y * (NT * 32);
    const int c_red = a.c_red, c_out = a.c_out;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    // this lane's output channel per tile (A operand row), clamped for the load, and the mask that zeroes rows beyond c_out
    int co_l[NT];
    unsigned co_m[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { const int co = co_base + nt * 32 + j; co_m[nt] = co < c_out ? 0xffffffffu : 0u; co_l[nt] = min(co, c_out - 1); }
    const float* xa = reinterpret_cast<const float*>(a.src.a);
    const float* xb = reinterpret_cast<const float*>(a.src.b);

    for (int ky = 0; ky < a.k; ++ky) {
        int iy;
        const bool vy = tap_source(oy, ky, a.stride, a.pad, a.gather, a.h_in, iy);   // wave-uniform
        if (!vy) continue;
        for (int kx = 0; kx < a.k; ++kx) {
            int ix;
            const bool valid = tap_source(ox, kx, a.stride, a.pad, a.gather, a.w_in, ix) && px_live;
            if (!__any(valid)) continue;   // every lane would multiply by 0: the identity
            const float* wt = a.w_f32 + (size_t)(ky * a.k + kx) * c_red * c_out;
            const int ixc = valid ? ix : 0;
            const unsigned vm = valid ? 0xffffffffu : 0u;
            const size_t pixel = ((size_t)n * a.h_in + iy) * a.w_in + ixc;   // always inside the tensor: the load is unconditional
            if (VEC) {
                for (int c8 = 0; c8 < c_red; c8 += 8) {
                    // every load of the group first: 8 channels of this lane's pixel (both inputs of a skip add) and the 4 NT weights
                    const Raw8<float> ra = raw_load8(xa + pixel * c_red + c8);
                    Raw8<float> rb{};
                    if (KIND == SRC_ACT2) rb = raw_load8(xb + pixel * c_red + c8);
                    float wv[4][NT];
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) wv[st][nt] = wt[(size_t)(c8 + 2 * st + hf) * c_out + co_l[nt]];
                    float v[8];
                    raw_to_float(ra, v);
                    if (KIND != SRC_RAW) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = relu_affine(v[q], a.src.a_scale[c8 + q], a.src.a_shift[c8 + q]);
                        if (KIND == SRC_ACT2) {
                            float u[8];
                            raw_to_float(rb, u);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += relu_affine(u[q], a.src.b_scale[c8 + q], a.src.b_shift[c8 + q]);
                        }
                    }
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const float b = and_mask(hf ? v[2 * st + 1] : v[2 * st], vm);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(wv[st][nt], co_m[nt]), b, acc[nt], 0, 0, 0);
                    }
                }
            } else {   // channel counts that are not multiples of 8 (the image: 1 or 3 channels; odd widths): one value at a time
                for (int c8 = 0; c8 < c_red; c8 += 8) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = and_mask(fetch1<float, KIND>(a.src, n, iy, ixc, a.h_in, a.w_in, c_red, min(c8 + q, c_red - 1)), (c8 + q < c_red) ? vm : 0u);
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        if (c8 + 2 * st >= c_red) break;          // wave-uniform: nothing but padding left in this group
                        const int c = c8 + 2 * st + hf;
                        const float b = hf ? v[2 * st + 1] : v[2 * st];
                        const unsigned cm = c < c_red ? 0xffffffffu : 0u;
                        const float* wr = wt + (size_t)min(c, c_red - 1) * c_out;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(wr[co_l[nt]], co_m[nt] & cm), b, acc[nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!px_live) return;
    // accumulator: column = pixel j; register r of lane half hf = output channel (r & 3) + 8 (r >> 2) + 4 hf of the tile
    const size_t opix = ((size_t)n * a.h_out + oy) * a.w_out + ox;
    float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co0 = co_base + nt * 32 + 8 * g + 4 * hf;
            if (co0 >= c_out) continue;
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = acc[nt][4 * g + q] + ((a.bias && co0 + q < c_out) ? a.bias[co0 + q] : 0.f);
            if (a.out_nchw) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (co0 + q < c_out) out[(((size_t)n * c_out + co0 + q) * a.h_out + oy) * a.w_out + ox] = r[q];
            } else if ((c_out & 3) == 0) *reinterpret_cast<float4*>(out + opix * c_out + co0) = make_float4(r[0], r[1], r[2], r[3]);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (co0 + q < c_out) out[opix * c_out + co0 + q] = r[q];
            }
        }
}

// The same computation for channel counts that are multiples of 8 (every layer but the image stem), software-pipelined: the
// (tap, 8-channel group) items of a wave are walked by a wave-uniform state machine, and the operands of item i+1 (32 B of the
// lane's pixel per input tensor + 4 NT weight dwords) are requested BEFORE the four MFMA steps of item i issue, so their L1 / L2
// round trip hides behind 4 NT x 64 matrix cycles instead of being exposed once per item; the producer's (scale, shift) tables
// sit in LDS.  Two register sets used alternately (A computes while B loads, then the reverse): no copies.  The instruction
// ORDER — and with it every sum — is exactly that of conv_f32_mfma_kernel.
template <int KIND, int NT>
__global__ __launch_bounds__(256) void conv_f32_mfma_vec_kernel(ConvArgs a, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) float f32_tab[];   // [a_scale | a_shift | b_scale | b_shift][c_red]
    const int c_red = a.c_red, c_out = a.c_out;
    if (KIND != SRC_RAW) {
        for (int i = threadIdx.x; i < c_red; i += 256) {
            f32_tab[i] = a.src.a_scale[i]; f32_tab[c_red + i] = a.src.a_shift[i];
            if (KIND == SRC_ACT2) { f32_tab[2 * c_red + i] = a.src.b_scale[i]; f32_tab[3 * c_red + i] = a.src.b_shift[i]; }
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, j = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_tiles = (int64_t)a.n * a.h_out * tiles_x;
    if (tile >= n_tiles) return;   // wave-uniform, after the only barrier
    const int tx = (int)(tile % tiles_x);
    const int64_t row = tile / tiles_x;
    const int oy = (int)(row % a.h_out), n = (int)(row / a.h_out);
    const int xs = a.gather ? a.stride : 1;
    const int ox = a.gather ? ((tx / xs) * 32 + j) * xs + (tx % xs) : tx * 32 + j;
    const bool px_live = ox < a.w_out;
    const int co_base = blockIdx.y * (NT * 32);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    int co_l[NT];
    unsigned co_m[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { const int co = co_base + nt * 32 + j; co_m[nt] = co < c_out ? 0xffffffffu : 0u; co_l[nt] = min(co, c_out - 1); }
    const float* xa = reinterpret_cast<const float*>(a.src.a);
    const float* xb = reinterpret_cast<const float*>(a.src.b);

    struct Item { Raw8<float> ra, rb; float wv[4][NT]; unsigned vm; int c8; };
    // the walk over (filter row, filter column, channel group): wave-uniform except for the lane's pixel and validity
    int ky = 0, kx = -1, c8 = c_red;
    size_t pix = 0;
    unsigned vm = 0;
    const float* wt = a.w_f32;
    auto next = [&]() __attribute__((always_inline)) -> bool {
        c8 += 8;
        if (c8 < c_red) return true;
        for (;;) {
            if (++kx >= a.k) { kx = 0; ++ky; }
            if (ky >= a.k) return false;
            int iy, ix;
            if (!tap_source(oy, ky, a.stride, a.pad, a.gather, a.h_in, iy)) { kx = a.k - 1; continue; }   // the whole filter row misses the image
            const bool valid = tap_source(ox, kx, a.stride, a.pad, a.gather, a.w_in, ix) && px_live;
            if (!__any(valid)) continue;   // every lane would multiply by 0: the identity
            wt = a.w_f32 + (size_t)(ky * a.k + kx) * c_red * c_out;
            vm = valid ? 0xffffffffu : 0u;
            pix = (((size_t)n * a.h_in + iy) * a.w_in + (valid ? ix : 0)) * c_red;   // always inside the tensor: unconditional loads
            c8 = 0;
            return true;
        }
    };
    auto load = [&](Item& it) __attribute__((always_inline)) {
        it.c8 = c8; it.vm = vm;
        it.ra = raw_load8(xa + pix + c8);
        if (KIND == SRC_ACT2) it.rb = raw_load8(xb + pix + c8);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) it.wv[st][nt] = wt[(size_t)(c8 + 2 * st + hf) * c_out + co_l[nt]];
    };
    auto compute = [&](const Item& it) __attribute__((always_inline)) {
        float v[8];
        raw_to_float(it.ra, v);
        if (KIND != SRC_RAW) {
            float sc[8], sh[8];
            load8<float>(f32_tab + it.c8, sc);
            load8<float>(f32_tab + c_red + it.c8, sh);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = relu_affine(v[q], sc[q], sh[q]);
            if (KIND == SRC_ACT2) {
                float u[8];
                raw_to_float(it.rb, u);
                load8<float>(f32_tab + 2 * c_red + it.c8, sc);
                load8<float>(f32_tab + 3 * c_red + it.c8, sh);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += relu_affine(u[q], sc[q], sh[q]);
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float b = and_mask(hf ? v[2 * st + 1] : v[2 * st], it.vm);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(it.wv[st][nt], co_m[nt]), b, acc[nt], 0, 0, 0);
        }
    };
    // Four items in flight per wave: an item is only four 64-cycle matrix instructions, and at 3-5 waves per SIMD two items per wave
    // cover a fraction of a memory round trip (the kernel ran at a third of the fp32 matrix rate).  The items are consumed strictly in
    // walk order — the oracle's summation order.
    Item I0, I1, I2, I3;
    bool m0 = next();
    if (m0) load(I0);
    bool m1 = m0 && next();
    if (m1) load(I1);
    bool m2 = m1 && next();
    if (m2) load(I2);
    while (m0) {
        const bool m3 = m2 && next();
        if (m3) load(I3);
        compute(I0);
        if (!m1) break;
        m0 = m3 && next();
        if (m0) load(I0);
        compute(I1);
        if (!m2) break;
        m1 = m0 && next();
        if (m1) load(I1);
        compute(I2);
        if (!m3) break;
        m2 = m1 && next();
        if (m2) load(I2);
        compute(I3);
    }
    if (!px_live) return;
    const size_t opix = ((size_t)n * a.h_out + oy) * a.w_out + ox;
    float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co0 = co_base + nt * 32 + 8 * g + 4 * hf;
            if (co0 >= c_out) continue;
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = acc[nt][4 * g + q] + ((a.bias && co0 + q < c_out) ? a.bias[co0 + q] : 0.f);
            if (a.out_nchw) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (co0 + q < c_out) out[(((size_t)n * c_out + co0 + q) * a.h_out + oy) * a.w_out + ox] = r[q];
            } else if ((c_out & 3) == 0) *reinterpret_cast<float4*>(out + opix * c_out + co0) = make_float4(r[0], r[1], r[2], r[3]);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (co0 + q < c_out) out[opix * c_out + co0 + q] = r[q];
            }
        }
}

bool conv_f32_mfma_selected(const ConvArgs& a) {
    static const int on = getenv("ANH_FP32_MFMA") ? atoi(getenv("ANH_FP32_MFMA")) : 1;
    const bool in_f32 = a.src.kind == SRC_IMAGE || a.src.dtype == DT_F32;
    const bool out_f32 = a.out_nchw || a.out_dtype == DT_F32;
    return on && in_f32 && out_f32 && !a.out_accumulate && !a.out2 && !a.stat_partials && !a.bnred_partials;
}

template <int NT>
void launch_conv_f32_mfma_nt(const ConvArgs& a, hipStream_t s) {
    const int xs = a.gather ? a.stride : 1;   // cont: one tile = 32 columns of one residue class mod stride
    const int tiles_x = (((a.w_out + xs - 1) / xs + 31) / 32) * xs;
    const int64_t n_tiles = (int64_t)a.n * a.h_out * tiles_x;
    const dim3 grid((unsigned)((n_tiles + 3) / 4), (unsigned)((a.c_out + NT * 32 - 1) / (NT * 32))), block(256);
    const bool vec = a.src.kind != SRC_IMAGE && (a.c_red % 8) == 0;
    static const int pipelined = getenv("ANH_FP32_MFMA_PIPELINED") ? atoi(getenv("ANH_FP32_MFMA_PIPELINED")) : 1;
    if (vec && pipelined) {
        const size_t lds = (size_t)a.c_red * 16;
        switch (a.src.kind) {
            case SRC_RAW: hipLaunchKernelGGL((conv_f32_mfma_vec_kernel<SRC_RAW, NT>), grid, block, lds, s, a, tiles_x); break;
            case SRC_ACT: hipLaunchKernelGGL((conv_f32_mfma_vec_kernel<SRC_ACT, NT>), grid, block, lds, s, a, tiles_x); break;
            default: hipLaunchKernelGGL((conv_f32_mfma_vec_kernel<SRC_ACT2, NT>), grid, block, lds, s, a, tiles_x); break;
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    switch (a.src.kind) {
        case SRC_RAW: if (vec) hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_RAW, NT, true>), grid, block, 0, s, a, tiles_x); else hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_RAW, NT, false>), grid, block, 0, s, a, tiles_x); break;
        case SRC_ACT: if (vec) hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_ACT, NT, true>), grid, block, 0, s, a, tiles_x); else hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_ACT, NT, false>), grid, block, 0, s, a, tiles_x); break;
        case SRC_ACT2: if (vec) hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_ACT2, NT, true>), grid, block, 0, s, a, tiles_x); else hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_ACT2, NT, false>), grid, block, 0, s, a, tiles_x); break;
        default: hipLaunchKernelGGL((conv_f32_mfma_kernel<SRC_IMAGE, NT, false>), grid, block, 0, s, a, tiles_x); break;
    }
    HIP_CHECK(hipGetLastError());
}

template <typename TIN, typename TOUT>
void conv_generic_dispatch(const ConvArgs& a, dim3 grid, dim3 block, hipStream_t s) {
    switch (a.src.kind) {
        case SRC_RAW: hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, SRC_RAW>), grid, block, 0, s, a); break;
        case SRC_ACT: hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, SRC_ACT>), grid, block, 0, s, a); break;
        case SRC_ACT2: hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, SRC_ACT2>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((conv_generic_kernel<TIN, TOUT, SRC_IMAGE>), grid, block, 0, s, a); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// stem_forward: the 5x5 stem on the u8 image (CIN = 1 or 3 -> 32 channels).  One thread = one output pixel x 32
// channels; the (8+4)x(32+4) image patch (already /256, clamp-to-edge window, zero padding) and the 25*CIN x 32 filter
// block live in LDS, filter rows are read as wave-wide broadcasts.  Same k-ordered fmaf chain as conv_generic / the
// oracle (taps row-major, input channel innermost), so the fp32 mode stays bit-exact.
// ---------------------------------------------------------------------------------------------------
template <int CIN, typename TOUT>
__global__ __launch_bounds__(256) void stem_forward_kernel(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int PH = 12, PW = 36, ROWS = 25 * CIN;
    __shared__ float xs[PH * PW * CIN];
    __shared__ __attribute__((aligned(16))) float ws[ROWS * 32];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int x0 = tx * 32, y0 = ty * 8;
    for (int i = tid; i < ROWS * 32; i += 256) ws[i] = a.w_f32[i];
    for (int i = tid; i < PH * PW * CIN; i += 256) {
        const int c = i % CIN, px = (i / CIN) % PW, py = i / (CIN * PW);
        const int iy = y0 - 2 + py, ix = x0 - 2 + px;
        xs[i] = (iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in) ? fetch1<float, SRC_IMAGE>(a.src, n, iy, ix, a.h_in, a.w_in, CIN, c) : 0.f;
    }
    __syncthreads();
    const int py = tid >> 5, px = tid & 31;
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.f;
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) {
            const float* xp = xs + ((py + ky) * PW + px + kx) * CIN;
            const float* wp = ws + (ky * 5 + kx) * CIN * 32;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float x = xp[ci];
#pragma unroll
                for (int o4 = 0; o4 < 8; ++o4) {
                    const float4 w = *reinterpret_cast<const float4*>(wp + ci * 32 + o4 * 4);
                    acc[o4 * 4 + 0] = fmaf(x, w.x, acc[o4 * 4 + 0]); acc[o4 * 4 + 1] = fmaf(x, w.y, acc[o4 * 4 + 1]);
                    acc[o4 * 4 + 2] = fmaf(x, w.z, acc[o4 * 4 + 2]); acc[o4 * 4 + 3] = fmaf(x, w.w, acc[o4 * 4 + 3]);
                }
            }
        }
    const int oy = y0 + py, ox = x0 + px;
    if (oy >= a.h_out || ox >= a.w_out) return;
    TOUT* out = reinterpret_cast<TOUT*>(a.out) + (((size_t)n * a.h_out + oy) * a.w_out + ox) * 32;
#pragma unroll
    for (int o8 = 0; o8 < 32; o8 += 8) store8<TOUT>(out + o8, acc + o8);
}

bool stem_forward_ok(const ConvArgs& a) {
    return a.src.kind == SRC_IMAGE && a.k == 5 && a.stride == 1 && a.pad == 2 && a.gather == 0 && a.c_out == 32 && (a.c_red == 1 || a.c_red == 3) &&
           a.h_in == a.h_out && a.w_in == a.w_out && !a.bias && !a.out_nchw && !a.out_accumulate && !a.out2;
}

// ---------------------------------------------------------------------------------------------------
// wgrad_generic: grid = (pixel split, tap, 32x32 (ci,co) tile); 256 threads, 2x2 outputs each; pixel chunks of 32
// staged in LDS; per-split partials, reduced in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------
template <typename TIN, typename TDY, int KIND>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(WgradArgs a, int64_t pix_per_split, int n_co_tiles) {
    __shared__ float xs[32][33];
    __shared__ float ds[32][33];
    const int tid = threadIdx.x;
    const int tap = blockIdx.y, ky = tap / a.k, kx = tap - ky * a.k;
    const int ci0 = (blockIdx.z / n_co_tiles) * 32, co0 = (blockIdx.z % n_co_tiles) * 32;
    const int64_t total = (int64_t)a.n * a.h_out * a.w_out;
    const int64_t p_begin = (int64_t)blockIdx.x * pix_per_split;
    const int64_t p_end = min(total, p_begin + pix_per_split);
    const int plane = a.h_out * a.w_out;
    const int tci = (tid >> 4) * 2, tco = (tid & 15) * 2;
    float acc00 = 0.f, acc01 = 0.f, acc10 = 0.f, acc11 = 0.f;
    const TDY* dy = reinterpret_cast<const TDY*>(a.dy);

    for (int64_t p0 = p_begin; p0 < p_end; p0 += 32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            const int px = e >> 5, ch = e & 31;
            const int64_t p = p0 + px;
            float xv = 0.f, dv = 0.f;
            if (p < p_end) {
                const int n = (int)(p / plane);
                const int rem = (int)(p - (int64_t)n * plane);
                const int oy = rem / a.w_out, ox = rem - oy * a.w_out;
                int iy, ix;
                const bool vy = tap_source(oy, ky, a.stride, a.pad, a.gather, a.h_in, iy);
                const bool valid = tap_source(ox, kx, a.stride, a.pad, a.gather, a.w_in, ix) && vy;
                if (valid && ci0 + ch < a.c_in) xv = fetch1<TIN, KIND>(a.src, n, iy, ix, a.h_in, a.w_in, a.c_in, ci0 + ch);
                if (co0 + ch < a.c_out) dv = to_f<TDY>(dy[(size_t)p * a.c_out + co0 + ch]);
            }
            xs[px][ch] = xv;
            ds[px][ch] = dv;
        }
        __syncthreads();
#pragma unroll 8
        for (int px = 0; px < 32; ++px) {
            const float x0 = xs[px][tci], x1 = xs[px][tci + 1];
            const float d0 = ds[px][tco], d1 = ds[px][tco + 1];
            acc00 = fmaf(x0, d0, acc00); acc01 = fmaf(x0, d1, acc01);
            acc10 = fmaf(x1, d0, acc10); acc11 = fmaf(x1, d1, acc11);
        }
        __syncthreads();
    }
    const size_t nw = (size_t)a.k * a.k * a.c_in * a.c_out;
    float* out = a.partials + (size_t)blockIdx.x * nw + (size_t)tap * a.c_in * a.c_out;
    const int ci = ci0 + tci, co = co0 + tco;
    if (ci < a.c_in && co < a.c_out) out[(size_t)ci * a.c_out + co] = acc00;
    if (ci < a.c_in && co + 1 < a.c_out) out[(size_t)ci * a.c_out + co + 1] = acc01;
    if (ci + 1 < a.c_in && co < a.c_out) out[(size_t)(ci + 1) * a.c_out + co] = acc10;
    if (ci + 1 < a.c_in && co + 1 < a.c_out) out[(size_t)(ci + 1) * a.c_out + co + 1] = acc11;
}

// out[i] = sum over splits of partials[split][i], in a fixed order.  A workgroup owns 256 consecutive elements (64 lanes
// x float4) and deals the splits round-robin to its 16 lane rows; each row sums its splits in order (double), then
// the 16 rows are summed in order.  nw is a multiple of 4 for every filter this build has (3x3 and 5x5 with >= 4
// channel products); the scalar tail handles anything else.
// XW lanes x 4 elements per workgroup, YR rows of threads each summing every YR-th partial.  (64, 16) streams wide rows; (16, 64) — four times
// the workgroups, a quarter of the partials per thread — is for short sums: the 32 -> 64 stride-2 layer's 18,432 elements ran as 72
// workgroups of the wide form, 51 us beside the stem's filter gradient at the tail of a step.
template <int XW, int YR>
__global__ __launch_bounds__(XW * YR) void reduce_partials_kernel(const float* __restrict__ partials, int splits, int64_t nw, float* __restrict__ out) {
    __shared__ double sh[YR][XW][4];
    const int64_t i = ((int64_t)blockIdx.x * XW + threadIdx.x) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i + 3 < nw && (nw & 3) == 0) {
#pragma unroll 4
        for (int k = threadIdx.y; k < splits; k += YR) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (size_t)k * nw + i);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
    } else {
        for (int k = threadIdx.y; k < splits; k += YR) {
            const float* row = partials + (size_t)k * nw;
            if (i < nw) s0 += (double)row[i];
            if (i + 1 < nw) s1 += (double)row[i + 1];
            if (i + 2 < nw) s2 += (double)row[i + 2];
            if (i + 3 < nw) s3 += (double)row[i + 3];
        }
    }
    double* mine = sh[threadIdx.y][threadIdx.x];
    mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3;
    __syncthreads();
    if (threadIdx.y < 4) {  // lane row e finishes element i + e
        const int e = threadIdx.y;
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < YR; ++g) t += sh[g][threadIdx.x][e];
        if (i + e < nw) out[i + e] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad_stem: filter gradient of the 5x5 stem on the u8 image.  dw[tap][ci][co] = sum_p img(p + tap - 2)[ci]/256 * dy[p][co]:
// only 25*CIN x 32 outputs but a reduction over every pixel, so it is a streaming kernel: one thread owns one (tap, ci) row
// and 8 output channels; an 8x32 pixel tile (image patch as fp32, dy as fp32) is staged in LDS; workgroups are persistent
// over a strided set of tiles and write one partial each (fixed-order reduction afterwards).
// ---------------------------------------------------------------------------------------------------
constexpr int kStemTH = 8, kStemTW = 32, kStemPH = kStemTH + 4, kStemPW = kStemTW + 4;

template <int CIN, typename TDY>
__global__ __launch_bounds__(320) void wgrad_stem_kernel(WgradArgs a, int tiles_x, int tiles_y, int total_tiles, int splits) {
    constexpr int ROWS = 25 * CIN;
    __shared__ float xs[kStemPH * kStemPW * CIN];
    __shared__ __attribute__((aligned(16))) float gs[kStemTH * kStemTW * 32];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int row = tid >> 2, co0 = (tid & 3) * 8;
    const bool active = row < ROWS;
    const int tap = row / CIN, ci = row - tap * CIN;
    const int ky = tap / 5, kx = tap - ky * 5;
    const TDY* dy = reinterpret_cast<const TDY*>(a.dy);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;

    for (int tile = blockIdx.x; tile < total_tiles; tile += splits) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * kStemTW, y0 = ty * kStemTH;
        __syncthreads();
        for (int i = tid; i < kStemPH * kStemPW * CIN; i += nthreads) {
            const int c = i % CIN, px = (i / CIN) % kStemPW, py = i / (CIN * kStemPW);
            const int iy = y0 - 2 + py, ix = x0 - 2 + px;
            xs[i] = (iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in) ? fetch1<float, SRC_IMAGE>(a.src, n, iy, ix, a.h_in, a.w_in, CIN, c) : 0.f;
        }
        for (int i = tid; i < kStemTH * kStemTW * 4; i += nthreads) {
            const int p = i >> 2, c8 = (i & 3) * 8;
            const int oy = y0 + (p >> 5), ox = x0 + (p & 31);
            float v[8];
            if (oy < a.h_out && ox < a.w_out) load8<TDY>(dy + (((size_t)n * a.h_out + oy) * a.w_out + ox) * 32 + c8, v);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            store8<float>(gs + p * 32 + c8, v);
        }
        __syncthreads();
        if (active) {
            for (int py = 0; py < kStemTH; ++py) {
                const float* xr = xs + ((py + ky) * kStemPW + kx) * CIN + ci;
                const float* gr = gs + (py * kStemTW) * 32 + co0;
#pragma unroll 8
                for (int px = 0; px < kStemTW; ++px) {
                    const float x = xr[px * CIN];
                    const float4 g0 = *reinterpret_cast<const float4*>(gr + px * 32), g1 = *reinterpret_cast<const float4*>(gr + px * 32 + 4);
                    acc[0] = fmaf(x, g0.x, acc[0]); acc[1] = fmaf(x, g0.y, acc[1]); acc[2] = fmaf(x, g0.z, acc[2]); acc[3] = fmaf(x, g0.w, acc[3]);
                    acc[4] = fmaf(x, g1.x, acc[4]); acc[5] = fmaf(x, g1.y, acc[5]); acc[6] = fmaf(x, g1.z, acc[6]); acc[7] = fmaf(x, g1.w, acc[7]);
                }
            }
        }
    }
    if (active) {
        float* out = a.partials + (size_t)blockIdx.x * ROWS * 32 + (size_t)row * 32 + co0;
        store8<float>(out, acc);
    }
}

bool stem_wgrad_ok(const WgradArgs& a) {
    return a.src.kind == SRC_IMAGE && a.k == 5 && a.stride == 1 && a.pad == 2 && a.gather == 0 && a.c_out == 32 && (a.c_in == 1 || a.c_in == 3) &&
           a.h_in == a.h_out && a.w_in == a.w_out;
}
int stem_wgrad_splits(const WgradArgs& a) {
    const int tiles = ((a.w_out + kStemTW - 1) / kStemTW) * ((a.h_out + kStemTH - 1) / kStemTH) * a.n;
    return std::max(1, std::min(tiles, 1024));
}

void wgrad_plan(const WgradArgs& a, int& splits, int64_t& pix_per_split, int& n_ci_tiles, int& n_co_tiles) {
    const int64_t total = (int64_t)a.n * a.h_out * a.w_out;
    n_ci_tiles = (a.c_in + 31) / 32;
    n_co_tiles = (a.c_out + 31) / 32;
    const int64_t per_split_blocks = (int64_t)a.k * a.k * n_ci_tiles * n_co_tiles;
    int64_t want = (2048 + per_split_blocks - 1) / per_split_blocks;
    const int64_t max_by_pixels = std::max<int64_t>(1, total / 512);
    want = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, 512), max_by_pixels));
    pix_per_split = ((total + want - 1) / want + 31) / 32 * 32;
    splits = (int)((total + pix_per_split - 1) / pix_per_split);
}

template <typename TIN, typename TDY>
void wgrad_generic_dispatch(const WgradArgs& a, dim3 grid, int64_t pps, int nco, hipStream_t s) {
    switch (a.src.kind) {
        case SRC_RAW: hipLaunchKernelGGL((wgrad_generic_kernel<TIN, TDY, SRC_RAW>), grid, dim3(256), 0, s, a, pps, nco); break;
        case SRC_ACT: hipLaunchKernelGGL((wgrad_generic_kernel<TIN, TDY, SRC_ACT>), grid, dim3(256), 0, s, a, pps, nco); break;
        case SRC_ACT2: hipLaunchKernelGGL((wgrad_generic_kernel<TIN, TDY, SRC_ACT2>), grid, dim3(256), 0, s, a, pps, nco); break;
        default: hipLaunchKernelGGL((wgrad_generic_kernel<TIN, TDY, SRC_IMAGE>), grid, dim3(256), 0, s, a, pps, nco); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// batch norm
// ---------------------------------------------------------------------------------------------------
// pixels per workgroup: small layers get smaller slabs so that the grid still covers the chip
inline int bn_pixels_per_block(int64_t pixels) {
    int64_t ppb = (pixels + 1023) / 1024;  // aim at ~1024 workgroups (4 per CU); fewer partials keep the finalize short
    ppb = (ppb + 63) / 64 * 64;
    if (ppb < 256) ppb = 256;
    if (ppb > 4096) ppb = 4096;
    return (int)ppb;
}

// block (cx, 256/cx): thread (c, r) walks pixels r, r+rows, ... of this block's slab for channels c, c+cx, ...
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* y, int64_t pixels, int c, double* partials, int ppb) {
    extern __shared__ float sh[];  // [rows][2*c]
    const int cx = blockDim.x, rows = blockDim.y;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = min(pixels, p0 + ppb);
    for (int ch = threadIdx.x; ch < c; ch += cx) {
        float s = 0.f, q = 0.f;
        for (int64_t p = p0 + threadIdx.y; p < p1; p += rows) {
            const float v = to_f<T>(y[(size_t)p * c + ch]);
            s += v; q = fmaf(v, v, q);
        }
        sh[(threadIdx.y * c + ch) * 2] = s;
        sh[(threadIdx.y * c + ch) * 2 + 1] = q;
    }
    __syncthreads();
    const int tid = threadIdx.y * cx + threadIdx.x;
    for (int ch = tid; ch < c; ch += cx * rows) {
        double s = 0, q = 0;
        for (int r = 0; r < rows; ++r) { s += sh[(r * c + ch) * 2]; q += sh[(r * c + ch) * 2 + 1]; }
        partials[((size_t)ch * 2) * gridDim.x + blockIdx.x] = s;
        partials[((size_t)ch * 2 + 1) * gridDim.x + blockIdx.x] = q;
    }
}

// one wave per channel: lanes stride over the per-workgroup partials, then a shuffle tree (fixed order => deterministic)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* partials, int blocks, int64_t pixels, int c, const float* gamma, const float* beta,
                                                         float eps, float* mean, float* invstd, float* scale, float* shift, double* var_out,
                                                         float* rmean, float* rvar, double af, double unbias) {
    const int ch = blockIdx.x;
    // this kernel is one dependent chain on the critical path of every layer: the scalar operands of its tail are
    // requested up front, so the chain holds ONE memory round trip (the partials) and not two
    const float gm = gamma[ch], bt = beta[ch];
    const float rm0 = rmean ? rmean[ch] : 0.f, rv0 = rmean ? rvar[ch] : 0.f;
    double s = 0, q = 0;
    const double* ps = partials + (size_t)ch * 2 * blocks;  // [channel][sum | sum of squares][workgroup]: contiguous per channel
#pragma unroll 8
    for (int b = threadIdx.x; b < blocks; b += 64) { s += ps[b]; q += ps[blocks + b]; }
    s = wave_sum(s); q = wave_sum(q);
    if (threadIdx.x != 0) return;
    const double m = s / (double)pixels;
    double var = q / (double)pixels - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gm * is;
    mean[ch] = mf; invstd[ch] = is; scale[ch] = sc; shift[ch] = fmaf(-mf, sc, bt);
    var_out[ch] = var;
    if (rmean) {  // dlib bn_ updates its running statistics in the training forward
        rmean[ch] = (float)((1.0 - af) * (double)rm0 + af * (double)mf);
        rvar[ch] = (float)((1.0 - af) * (double)rv0 + af * unbias * var);
    }
}

__global__ void bn_running_kernel(const float* mean, const double* var, float* rmean, float* rvar, int c, double af, double unbias) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    rmean[ch] = (float)((1.0 - af) * (double)rmean[ch] + af * (double)mean[ch]);
    rvar[ch] = (float)((1.0 - af) * (double)rvar[ch] + af * unbias * var[ch]);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* da, const T* y, int64_t pixels, int c, const float* mean, const float* invstd,
                                                            const float* scale, const float* shift, double* partials, int ppb) {
    extern __shared__ float sh[];
    const int cx = blockDim.x, rows = blockDim.y;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = min(pixels, p0 + ppb);
    for (int ch = threadIdx.x; ch < c; ch += cx) {
        const float m = mean[ch], is = invstd[ch], sc = scale[ch], sf = shift[ch];
        float sg = 0.f, sb = 0.f;
        for (int64_t p = p0 + threadIdx.y; p < p1; p += rows) {
            const size_t i = (size_t)p * c + ch;
            const float yv = to_f<T>(y[i]);
            const float dz = fmaf(yv, sc, sf) > 0.f ? to_f<T>(da[i]) : 0.f;
            sg = fmaf(dz, (yv - m) * is, sg);
            sb += dz;
        }
        sh[(threadIdx.y * c + ch) * 2] = sg;
        sh[(threadIdx.y * c + ch) * 2 + 1] = sb;
    }
    __syncthreads();
    const int tid = threadIdx.y * cx + threadIdx.x;
    for (int ch = tid; ch < c; ch += cx * rows) {
        double g = 0, b = 0;
        for (int r = 0; r < rows; ++r) { g += sh[(r * c + ch) * 2]; b += sh[(r * c + ch) * 2 + 1]; }
        partials[((size_t)ch * 2) * gridDim.x + blockIdx.x] = g;
        partials[((size_t)ch * 2 + 1) * gridDim.x + blockIdx.x] = b;
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* partials, int blocks, int64_t pixels, int c, const float* gamma, const float* invstd,
                                                             float* dgamma, float* dbeta, float* coef) {
    const int ch = blockIdx.x;
    const float gi = gamma[ch] * invstd[ch];   // requested before the partials (see bn_finalize_kernel)
    double g = 0, b = 0;
    const double* ps = partials + (size_t)ch * 2 * blocks;
#pragma unroll 8
    for (int k = threadIdx.x; k < blocks; k += 64) { g += ps[k]; b += ps[blocks + k]; }
    g = wave_sum(g); b = wave_sum(b);
    if (threadIdx.x != 0) return;
    dgamma[ch] = (float)g; dbeta[ch] = (float)b;
    coef[ch] = gi;
    coef[c + ch] = (float)(b / (double)pixels);
    coef[2 * c + ch] = (float)(g / (double)pixels);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* da, T* out, const T* y, int64_t total, int c, const float* mean, const float* invstd,
                                                           const float* scale, const float* shift, const float* coef) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c);
        const float yv = to_f<T>(y[i]);
        const float dz = fmaf(yv, scale[ch], shift[ch]) > 0.f ? to_f<T>(da[i]) : 0.f;
        const float xhat = (yv - mean[ch]) * invstd[ch];
        out[i] = from_f<T>(coef[ch] * (dz - coef[c + ch] - xhat * coef[2 * c + ch]));
    }
}

// ---- vectorized batch-norm kernels (C a multiple of 8 that divides 2048): every thread owns one fixed 8-channel group
// and walks pixels with 16-byte (bf16) / 32-byte (fp32) loads; consecutive threads read consecutive chunks, so a wave
// streams 1 KiB of contiguous NHWC per instruction.  Partials go out as doubles; the finalize kernels are shared. ----
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_vec_kernel(const T* y, int64_t pixels, int c, double* partials, int ppb) {
    __shared__ float sh[256][17];
    const int groups = c >> 3, cg = threadIdx.x % groups, lane_px = threadIdx.x / groups, px_step = 256 / groups;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = min(pixels, p0 + ppb);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
#pragma unroll 4
    for (int64_t p = p0 + lane_px; p < p1; p += px_step) {
        float v[8];
        load8<T>(y + (size_t)p * c + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] = fmaf(v[j], v[j], q[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[threadIdx.x][j] = s[j]; sh[threadIdx.x][8 + j] = q[j]; }
    __syncthreads();
    // thread t < 2*c finalises (channel, which) = (t >> 1, t & 1)
    for (int t = threadIdx.x; t < 2 * c; t += 256) {
        const int ch = t >> 1, which = t & 1;
        const int g = ch >> 3, j = ch & 7;
        double acc = 0;
        for (int r = 0; r < px_step; ++r) acc += sh[r * groups + g][which * 8 + j];
        partials[((size_t)ch * 2 + which) * gridDim.x + blockIdx.x] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const T* da, const T* y, int64_t pixels, int c, const float* mean, const float* invstd,
                                                                const float* scale, const float* shift, double* partials, int ppb, BnBwdFinish fin) {
    long long* acc_table = fin.acc;
    __shared__ float sh[256][17];
    const int groups = c >> 3, cg = threadIdx.x % groups, lane_px = threadIdx.x / groups, px_step = 256 / groups;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = min(pixels, p0 + ppb);
    float m[8], is[8], sc[8], sf[8], sg[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; sc[j] = scale[cg * 8 + j]; sf[j] = shift[cg * 8 + j];
        sg[j] = 0.f; sb[j] = 0.f;
    }
#pragma unroll 2
    for (int64_t p = p0 + lane_px; p < p1; p += px_step) {
        float yv[8], dv[8];
        load8<T>(y + (size_t)p * c + cg * 8, yv);
        load8<T>(da + (size_t)p * c + cg * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dz = fmaf(yv[j], sc[j], sf[j]) > 0.f ? dv[j] : 0.f;
            sg[j] = fmaf(dz, (yv[j] - m[j]) * is[j], sg[j]);
            sb[j] += dz;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[threadIdx.x][j] = sg[j]; sh[threadIdx.x][8 + j] = sb[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * c; t += 256) {
        const int ch = t >> 1, which = t & 1;
        const int g = ch >> 3, j = ch & 7;
        double acc = 0;
        for (int r = 0; r < px_step; ++r) acc += sh[r * groups + g][which * 8 + j];
        if (acc_table) bnacc_add(acc_table, BNACC_SUM_DZ_XHAT + which, c, ch, acc);
        else partials[((size_t)ch * 2 + which) * gridDim.x + blockIdx.x] = acc;
    }
    if (acc_table) bnacc_finish_backward(fin, (int)gridDim.x);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const T* da, T* out, const T* y, int64_t chunks, int c, const float* mean, const float* invstd,
                                                               const float* scale, const float* shift, const float* coef) {
    const int groups = c >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;  // a multiple of `groups`: a thread keeps its channel group
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(first % groups);
    float m[8], is[8], sc[8], sf[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cg * 8 + j;
        m[j] = mean[ch]; is[j] = invstd[ch]; sc[j] = scale[ch]; sf[j] = shift[ch];
        k0[j] = coef[ch]; k1[j] = coef[c + ch]; k2[j] = coef[2 * c + ch];
    }
    // (written on pairs with v_pk_* instructions this pass measured +1.1 % per step in round 4: profiles/r04_ab_log.txt)
    for (int64_t i = first; i < chunks; i += stride) {
        float yv[8], dv[8], r[8];
        load8<T>(y + (size_t)i * 8, yv);
        load8<T>(da + (size_t)i * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dz = fmaf(yv[j], sc[j], sf[j]) > 0.f ? dv[j] : 0.f;
            const float xhat = (yv[j] - m[j]) * is[j];
            r[j] = k0[j] * (dz - k1[j] - xhat * k2[j]);
        }
        store8<T>(out + (size_t)i * 8, r);
    }
}

inline bool bn_vec_ok(int c) { return c >= 8 && (c % 8) == 0 && (256 % (c / 8)) == 0; }

// ---------------------------------------------------------------------------------------------------
// loss_multiclass_log_per_pixel_weighted
// ---------------------------------------------------------------------------------------------------
constexpr int kLossPixelsPerBlock = 2048;

template <int KMAX>
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
    __shared__ double red[256];
    const int64_t p0 = (int64_t)blockIdx.x * kLossPixelsPerBlock;
    const int64_t p1 = min(a.pixels, p0 + kLossPixelsPerBlock);
    const int K = a.k;
    double loss = 0.0;
    float dbias[KMAX], z[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) dbias[k] = 0.f;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const uint16_t y = a.labels[p];
        float* g = a.dlogits + (size_t)p * K;
        if (y == ANH_LABEL_IGNORE || y >= K) {
            if (y != ANH_LABEL_IGNORE && a.error_flag) *a.error_flag = 1;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) g[k] = 0.f;
            continue;
        }
        const float* zp = a.logits + (size_t)p * K;
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) { z[k] = zp[k]; m = fmaxf(m, z[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) { z[k] = expf(z[k] - m); sum += z[k]; }
        const float sw = (float)a.scale * a.weights[p];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) {
            const float pk = z[k] / sum;
            float gk;
            if (k == y) { loss += (double)sw * (double)(-logf(fmaxf(pk, 1e-10f))); gk = sw * (pk - 1.f); }
            else gk = sw * pk;
            g[k] = gk;
            dbias[k] += gk;
        }
    }
    for (int slot = 0; slot <= K; ++slot) {
        double mine = loss;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (slot == k + 1) mine = (double)dbias[k];
        red[threadIdx.x] = mine;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.partials[(size_t)blockIdx.x * (K + 1) + slot] = red[0];
        __syncthreads();
    }
}

// one wave per slot (0 = loss, 1..k = bias gradient)
__global__ __launch_bounds__(64) void loss_finalize_kernel(const double* partials, int blocks, int k, double* loss_out, float* loss_out_f32, float* dbias) {
    const int slot = blockIdx.x;
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += 64) s += partials[(size_t)b * (k + 1) + slot];
    s = wave_sum(s);
    if (threadIdx.x != 0) return;
    if (slot == 0) { *loss_out = s; if (loss_out_f32) *loss_out_f32 = (float)s; }
    else dbias[slot - 1] = (float)s;
}

// ---------------------------------------------------------------------------------------------------
// head_train: the whole tail of a training step in one pass over the last hidden tensor —
//   logits = relu(bn(y)) . W + b   (1x1 head, fp32, same k-ordered fmaf chain as conv_generic)
//   weighted per-pixel softmax log-loss and its gradient (as loss_kernel)
//   dA   = dlogits . W^T            (head backward-data, written in the storage type)
//   dW  += a (x) dlogits, db += dlogits   (head filter / bias gradient, per-thread registers -> wave tree -> partials)
// One thread = one pixel at a time (C = 32 channels = one 64-byte record), grid-stride.
// ---------------------------------------------------------------------------------------------------
constexpr int kHeadC = 32, kHeadKMax = 4;

// Table mode (bnacc.h): one fold job = one bn layer's arrays (mean, invstd, scale, shift, var) and running statistics from its
// accumulator table, with bn_finalize_kernel's arithmetic.  Run by a whole workgroup; nothing in the same launch reads the results.
__device__ __forceinline__ void bn_fold_job(const BnFoldJob& j) {
    for (int ch = threadIdx.x; ch < j.c; ch += blockDim.x) {
        const float rm0 = j.rmean ? j.rmean[ch] : 0.f, rv0 = j.rmean ? j.rvar[ch] : 0.f;
        const BnFolded f = bnacc_fold_forward(j.acc, j.c, ch, j.pixels, j.gamma[ch], j.beta[ch], j.eps);
        j.mean[ch] = f.mean; j.invstd[ch] = f.invstd; j.scale[ch] = f.scale; j.shift[ch] = f.shift;
        j.var[ch] = f.var;
        if (j.rmean) {  // dlib bn_ updates its running statistics in the training forward
            j.rmean[ch] = (float)((1.0 - j.af) * (double)rm0 + j.af * (double)f.mean);
            j.rvar[ch] = (float)((1.0 - j.af) * (double)rv0 + j.af * j.unbias * f.var);
        }
    }
}
__global__ __launch_bounds__(128) void bn_fold_all_kernel(BnFoldJobs jobs) {
    if ((int)blockIdx.x < jobs.n) bn_fold_job(jobs.job[blockIdx.x]);
}

// the same pass for the layer under the fused 1x1 head, whose da is not in memory: a thread recomputes its 8 channels of
// da = round_T(sum_k g[p][k] * w_tm[ch][k]) from the pixel's dlogits (head_train_kernel's expression and order)
template <typename T, int KM>
__global__ __launch_bounds__(256) void bn_bwd_apply_head_kernel(const float* g, const float* w_tm, int K, T* out, const T* y, int64_t chunks, const float* mean,
                                                                const float* invstd, const float* scale, const float* shift, const float* coef) {
    constexpr int C = kHeadC, groups = C >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(first % groups);
    float m[8], is[8], sc[8], sf[8], k0[8], k1[8], k2[8], w[8][KM];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cg * 8 + j;
        m[j] = mean[ch]; is[j] = invstd[ch]; sc[j] = scale[ch]; sf[j] = shift[ch];
        k0[j] = coef[ch]; k1[j] = coef[C + ch]; k2[j] = coef[2 * C + ch];
#pragma unroll
        for (int k = 0; k < KM; ++k) w[j][k] = k < K ? w_tm[ch * K + k] : 0.f;
    }
    for (int64_t i = first; i < chunks; i += stride) {
        const int64_t p = i / groups;
        float gk[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) gk[k] = k < K ? g[(size_t)p * K + k] : 0.f;
        float yv[8], r[8];
        load8<T>(y + (size_t)i * 8, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) acc = fmaf(gk[k], w[j][k], acc);
            const float dz = fmaf(yv[j], sc[j], sf[j]) > 0.f ? operand_round<T>(acc) : 0.f;
            const float xhat = (yv[j] - m[j]) * is[j];
            r[j] = k0[j] * (dz - k1[j] - xhat * k2[j]);
        }
        store8<T>(out + (size_t)i * 8, r);
    }
}


// value of another lane of the same quad (DPP quad_perm: a VALU modifier, no LDS crossbar)
template <int CTRL> __device__ __forceinline__ float quad_perm(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_bcast(float v, int k) {   // k: a constant after unrolling
    switch (k) {
        case 0: return quad_perm<0x00>(v);
        case 1: return quad_perm<0x55>(v);
        case 2: return quad_perm<0xAA>(v);
        default: return quad_perm<0xFF>(v);
    }
}

struct HeadArgs {
    Src src;                       // SRC_ACT or SRC_ACT2, 32 channels
    const float* w_tm;             // [ci][k]   (tap-major with one tap), bf16-rounded values in bf16 mode
    const float* w_km;             // [k][ci]
    const float* bias;
    const uint16_t* labels; const float* weights;
    float* logits;                 // [P][K] fp32 (kept for inspection)
    void* da;                      // [P][32] storage type (null: not materialised)
    float* dlogits;                // [P][K] fp32 or null
    int64_t pixels; int k;
    double scale;
    double* partials;              // [blocks][1 + K + 32*K]
    int* error_flag;
    const float* bn_mean; const float* bn_invstd; double* bn_partials;   // fused bn backward sums of the input layer (SRC_ACT only)
    long long* bn_acc;             // table mode: those sums are added to the input layer's accumulator table instead (bnacc.h)
    BnBwdFinish bn_finish;         // ... and the last workgroup folds them
};

// Four lanes share a pixel: lane `sub` owns channels 8*sub..8*sub+7 (one 16-byte chunk), so a wave reads / writes 1 KiB
// of contiguous NHWC per instruction; the K partial logits are combined with two xor-shuffles.
// KM = register slots per class dimension.  2, 3 and 4 classes get exact instantiations (K = KM known at compile time: no
// per-class predicates, no dead slots); a one-class net runs the two-slot body with K at run time.
template <typename T, int KIND, int KM, bool EXACT>
__global__ __launch_bounds__(256, 2) void head_train_kernel(HeadArgs a, BnFoldJobs jobs) {
    constexpr int C = kHeadC;
    const int K = EXACT ? KM : a.k;
    const int sub = threadIdx.x & 3, c0 = sub * 8;
    // table mode: the first workgroups also form every bn layer's arrays and running statistics (read by the backward kernels, all
    // of which are launched after this one); this kernel itself folds what it needs of its input layers below
    if ((int)blockIdx.x < jobs.n) bn_fold_job(jobs.job[blockIdx.x]);
    __shared__ float sfold[6][C];   // a_scale, a_shift, b_scale, b_shift, a_mean, a_invstd
    const bool tables = a.src.a_tab.acc != nullptr;
    if (tables) {
        const int sides = KIND == SRC_ACT2 ? 2 : 1;
        if ((int)threadIdx.x < sides * C) {
            const int side = threadIdx.x >= C, ch = threadIdx.x - side * C;
            const BnTable& t = side ? a.src.b_tab : a.src.a_tab;
            const BnFolded f = bnacc_fold_forward(t.acc, t.c, ch, t.pixels, t.gamma[ch], t.beta[ch], t.eps);
            sfold[2 * side][ch] = f.scale; sfold[2 * side + 1][ch] = f.shift;
            if (!side) { sfold[4][ch] = f.mean; sfold[5][ch] = f.invstd; }
        }
        __syncthreads();
    }
    // w_tm[ci][k] and w_km[k][ci] are two layouts of the same filter: one register copy serves forward and backward-data
    float w[8][KM], bias[KM], sa[8], ta[8], sb[8], tb[8];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        bias[k] = k < K ? a.bias[k] : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c][k] = k < K ? a.w_tm[(c0 + c) * K + k] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (tables) {
            sa[c] = sfold[0][c0 + c]; ta[c] = sfold[1][c0 + c];
            sb[c] = KIND == SRC_ACT2 ? sfold[2][c0 + c] : 0.f; tb[c] = KIND == SRC_ACT2 ? sfold[3][c0 + c] : 0.f;
        } else {
            sa[c] = a.src.a_scale[c0 + c]; ta[c] = a.src.a_shift[c0 + c];
            sb[c] = KIND == SRC_ACT2 ? a.src.b_scale[c0 + c] : 0.f; tb[c] = KIND == SRC_ACT2 ? a.src.b_shift[c0 + c] : 0.f;
        }
    }
    const bool bnred = KIND == SRC_ACT && (a.bn_partials != nullptr || a.bn_acc != nullptr);
    // bn backward sums of the input layer.  The loop keeps sg = sum dz*y and sb2 = sum dz; the workgroup's partial of
    // sum dz*xhat = invstd * (sum dz*y - mean * sum dz) is formed once, in double, after the folds (8 fewer VALU per
    // pixel and 16 fewer registers than carrying mean / invstd through the loop).
    float sg[8], sb2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sg[c] = 0.f; sb2[c] = 0.f; }
    float dw[8][KM], db_mine = 0.f;   // db of class `sub`
#pragma unroll
    for (int k = 0; k < KM; ++k) {
#pragma unroll
        for (int c = 0; c < 8; ++c) dw[c][k] = 0.f;
    }
    float loss = 0.f;
    const T* xa = reinterpret_cast<const T*>(a.src.a);
    const T* xb = reinterpret_cast<const T*>(a.src.b);
    T* da = reinterpret_cast<T*>(a.da);
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 2;

    // Two pixels per iteration, and the loads of the NEXT two are issued (unconditionally, from clamped indices) before
    // the current two are processed: four pixel records per thread in flight keep HBM busy at 2 waves per SIMD.
    struct Pre { Raw8<T> xa, xb; uint16_t y; float wgt; };
    auto issue = [&](int64_t p, Pre& r) {
        const int64_t pc = min(p, a.pixels - 1);
        r.xa = raw_load8(xa + (size_t)pc * C + c0);
        if (KIND == SRC_ACT2) r.xb = raw_load8(xb + (size_t)pc * C + c0);
        r.y = a.labels[pc];
        r.wgt = a.weights[pc];
    };
    auto process = [&](int64_t p, const Pre& r) {
        float x[8], yraw[8];
        raw_to_float(r.xa, x);
#pragma unroll
        for (int c = 0; c < 8; ++c) { yraw[c] = x[c]; x[c] = relu_affine(x[c], sa[c], ta[c]); }
        bool pos[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pos[c] = x[c] > 0.f;   // relu(z) > 0  <=>  z = y*scale+shift > 0
        if (KIND == SRC_ACT2) {
            float u[8];
            raw_to_float(r.xb, u);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] += relu_affine(u[c], sb[c], tb[c]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = operand_round<T>(x[c]);
        float z[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc = fmaf(x[c], w[c][k], acc);
            acc += quad_perm<0xB1>(acc);   // lane ^ 1
            acc += quad_perm<0x4E>(acc);   // lane ^ 2
            z[k] = acc + bias[k];
        }
        // From here lane `sub` of the pixel's four owns class `sub`: one exp, one division and one 4-byte store per lane instead of
        // all K on every lane; what every lane needs comes back with quad broadcasts.  Same expressions in the same order as the
        // plain form (sum over k ascending, e / sum), and no branch: a pixel without a label takes the selects' zero side.
        float z_mine = z[0];
#pragma unroll
        for (int k = 1; k < KM; ++k) z_mine = sub == k ? z[k] : z_mine;
        const bool mine = sub < K;
        if (mine) a.logits[(size_t)p * K + sub] = z_mine;
        const uint16_t y = r.y;
        const bool valid = y != ANH_LABEL_IGNORE && y < K;
        if (!valid && y != ANH_LABEL_IGNORE && a.error_flag) *a.error_flag = 1;
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < KM; ++k) if (k < K) m = fmaxf(m, z[k]);
        const float e_mine = mine ? expf(z_mine - m) : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) sum += quad_bcast(e_mine, k);
        const float pk_mine = e_mine / sum;
        const float sw = (float)a.scale * r.wgt;
        const bool is_y = sub == (int)y;
        const float g_mine = (valid && mine) ? (is_y ? sw * (pk_mine - 1.f) : sw * pk_mine) : 0.f;
        float py = (valid && is_y) ? pk_mine : 0.f;   // one lane of the four holds p[y]; x + 0 is exact
        py += quad_perm<0xB1>(py);
        py += quad_perm<0x4E>(py);
        if (valid && sub == 0) loss += sw * (-logf(fmaxf(py, 1e-10f)));
        float g[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) g[k] = quad_bcast(g_mine, k);
        float dx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) acc = fmaf(g[k], w[c][k], acc);
            dx[c] = acc;
#pragma unroll
            for (int k = 0; k < KM; ++k) dw[c][k] = fmaf(x[c], g[k], dw[c][k]);
        }
        db_mine += g_mine;
        if (da) store8<T>(da + (size_t)p * C + c0, dx);
        if (a.dlogits && mine) a.dlogits[(size_t)p * K + sub] = g_mine;
        if (bnred) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float dz = pos[c] ? operand_round<T>(dx[c]) : 0.f;   // the STORED da, as bn_bwd_reduce would read it
                sg[c] = fmaf(dz, yraw[c], sg[c]);
                sb2[c] += dz;
            }
        }
    };
    int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    Pre r0, r1;
    issue(p, r0);
    issue(p + stride, r1);
    for (; p < a.pixels; p += 2 * stride) {
        const Pre q0 = r0, q1 = r1;
        issue(p + 2 * stride, r0);
        issue(p + 3 * stride, r1);
        process(p, q0);
        if (p + stride < a.pixels) process(p + stride, q1);  // the four lanes of a pixel agree on this branch
    }
    // lanes with equal `sub` hold the same (channel, class) slots: fold them (offsets 4..32 keep `sub`), then across waves
    __shared__ double red[4][1 + KM + C * KM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto fold = [](float v) {
        double d = (double)v;
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) d += __shfl_down(d, off, 64);
        return d;
    };
    {
        const double l = fold(loss);
        if (lane == 0) red[wave][0] = l;
        { const double v = fold(db_mine); if (lane < KM) red[wave][1 + lane] = v; }   // lanes 0..3 = sub 0..3 = class
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < KM; ++k) { const double v = fold(dw[c][k]); if (lane < 4) red[wave][1 + KM + (lane * 8 + c) * KM + k] = v; }
    }
    __shared__ double redb[4][2][C];
    if (bnred) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const double v0 = fold(sg[c]), v1 = fold(sb2[c]);
            if (lane < 4) { redb[wave][0][lane * 8 + c] = v0; redb[wave][1][lane * 8 + c] = v1; }
        }
    }
    __syncthreads();
    if (bnred && threadIdx.x < 2 * C) {
        const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
        const double sy = ((redb[0][0][ch] + redb[1][0][ch]) + redb[2][0][ch]) + redb[3][0][ch];
        const double sd = ((redb[0][1][ch] + redb[1][1][ch]) + redb[2][1][ch]) + redb[3][1][ch];
        const double mean_ = tables ? (double)sfold[4][ch] : (double)a.bn_mean[ch], invstd_ = tables ? (double)sfold[5][ch] : (double)a.bn_invstd[ch];
        const double v = which ? sd : invstd_ * (sy - mean_ * sd);
        if (a.bn_acc) bnacc_add(a.bn_acc, BNACC_SUM_DZ_XHAT + which, C, ch, v);
        else a.bn_partials[((size_t)ch * 2 + which) * gridDim.x + blockIdx.x] = v;
    }
    if (bnred && a.bn_acc) {
        // The finish forms k0 = gamma * invstd.  The layer's invstd ARRAY is written by a fold job of THIS launch (another workgroup,
        // possibly behind another XCD's L2): a plain load of it here may still see what memory held before — zero on a trainer's first
        // step (every gradient below this layer then comes out zero), last step's value afterwards.  This workgroup folded the same
        // number from the table itself (sfold[5], bnacc_fold_forward: the fold job's own arithmetic): use that.
        BnBwdFinish fin = a.bn_finish;
        if (tables) fin.invstd = sfold[5];
        bnacc_finish_backward(fin, (int)gridDim.x);
    }
    const int slots = 1 + K + C * K;
    for (int sidx = threadIdx.x; sidx < slots; sidx += blockDim.x) {
        int src;
        if (sidx <= K) src = sidx;  // loss, db[k]
        else { const int e = sidx - 1 - K, c = e / K, k = e - c * K; src = 1 + KM + c * KM + k; }
        a.partials[(size_t)sidx * gridDim.x + blockIdx.x] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// slot 0 -> loss, 1..K -> dbias, then dW[ci][k]
__global__ __launch_bounds__(64) void head_finalize_kernel(const double* partials, int blocks, int slots, int k, double* loss_out, float* loss_out_f32,
                                                           float* dbias, float* dw) {
    const int slot = blockIdx.x;
    double s = 0;
    const double* ps = partials + (size_t)slot * blocks;  // [slot][workgroup]
#pragma unroll 8
    for (int b = threadIdx.x; b < blocks; b += 64) s += ps[b];
    s = wave_sum(s);
    if (threadIdx.x != 0) return;
    if (slot == 0) { *loss_out = s; if (loss_out_f32) *loss_out_f32 = (float)s; }
    else if (slot <= k) dbias[slot - 1] = (float)s;
    else dw[slot - 1 - k] = (float)s;
}

// ---------------------------------------------------------------------------------------------------
// SGD + layout refresh
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const ParamSegment* seg, int n, int64_t i) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// canonical index within a filter segment -> (tap-major, k-major) indices
__device__ __forceinline__ void filter_indices(const ParamSegment& sg, int64_t local64, int64_t& tm, int64_t& km) {
    // 32-bit arithmetic: a filter segment holds k^2 cin cout < 2^31 elements (a 64-bit division by a run-time divisor is ~100
    // instructions on this ISA, three of them per parameter were a quarter of the update kernel)
    const unsigned local = (unsigned)local64, kk = (unsigned)(sg.k * sg.k);
    const unsigned r = local / kk, t = local - r * kk;
    unsigned ci, co;
    if (sg.type == 0) { co = r / (unsigned)sg.cin; ci = r - co * (unsigned)sg.cin; }   // [co][ci][t]
    else { ci = r / (unsigned)sg.cout; co = r - ci * (unsigned)sg.cout; }              // [ci][co][t]
    tm = (int64_t)((t * (unsigned)sg.cin + ci) * (unsigned)sg.cout + co);
    km = (int64_t)((t * (unsigned)sg.cout + co) * (unsigned)sg.cin + ci);
}

__global__ __launch_bounds__(256) void sgd_kernel(SgdArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t z = i; z < a.zero_words16; z += (int64_t)gridDim.x * blockDim.x) reinterpret_cast<uint4*>(a.zero)[z] = make_uint4(0u, 0u, 0u, 0u);
    if (i == 0 && a.loss_post)
        __hip_atomic_store(a.loss_post, ((unsigned long long)a.loss_tag << 32) | (unsigned long long)__float_as_uint(*a.loss_src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (i >= a.n_params) return;
    const ParamSegment sg = a.segments[find_segment(a.segments, a.n_segments, i)];
    const int64_t local = i - sg.start;
    int64_t tm = local, km = local;
    if (sg.kind == 0) filter_indices(sg, local, tm, km);
    float w = a.master[i];
    if (a.apply) {
        const double g = (double)a.grad_tm[sg.start + tm] * a.grad_scale;
        const double wd = sg.kind == 0 ? a.weight_decay : 0.0;
        const float v = (float)(a.momentum_coef * (double)a.momentum[i] - wd * a.lr * (double)w - a.lr * g);
        a.momentum[i] = v;
        w += v;
        a.master[i] = w;
    }
    // bf16 mode: the fp32 compute copies of the FILTERS carry the bf16-rounded value, so every kernel multiplies by
    // the same weights as the MFMA path; bias / gamma / beta stay fp32
    const float wq = (a.w_tm_bf16 && sg.kind == 0) ? (float)(bf16)w : w;
    a.w_tm_f32[sg.start + tm] = wq;
    a.w_km_f32[sg.start + km] = wq;
    if (a.w_tm_bf16) reinterpret_cast<bf16*>(a.w_tm_bf16)[sg.start + tm] = (bf16)w;
    if (a.w_km_bf16) reinterpret_cast<bf16*>(a.w_km_bf16)[sg.start + km] = (bf16)w;
}

__global__ void tm_to_canonical_kernel(const ParamSegment* segments, int n_segments, int64_t n_params, const float* tm_blob, float* canon) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_params) return;
    const ParamSegment sg = segments[find_segment(segments, n_segments, i)];
    const int64_t local = i - sg.start;
    int64_t tm = local, km = local;
    if (sg.kind == 0) filter_indices(sg, local, tm, km);
    canon[i] = tm_blob[sg.start + tm];
}

// ---------------------------------------------------------------------------------------------------
// inference glue
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ramp(long long coordinate, long long first_possible, long long first_in, long long last_in, long long last_possible) {
    // get_t (annonet_infer.cpp:102-114)
    if (coordinate < first_in) return (double)(coordinate - first_possible) / (double)(first_in - first_possible);
    if (coordinate > last_in) return (double)(last_possible - coordinate) / (double)(last_possible - last_in);
    return 1.0;
}

// one thread per (x of the tile, y of the tile); loops over classes.  annonet_infer.cpp:116-164
__global__ __launch_bounds__(256) void blend_kernel(BlendArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= a.tile_w) return;
    const long long bx = (long long)a.tile_left + x, by = (long long)a.tile_top + y;
    if (by < a.full[1] || by > a.full[3] || by < 0 || by >= a.img_h) return;
    if (bx < a.full[0] || bx > a.full[2] || bx < 0 || bx >= a.img_w) return;
    const bool inside_unique = bx >= a.unique[0] && bx <= a.unique[2] && by >= a.unique[1] && by <= a.unique[3];
    double t = 1.0;
    if (!inside_unique) {
        const double th = ramp(bx, a.full[0], a.unique[0], a.unique[2], a.full[2]);
        const double tv = ramp(by, a.full[1], a.unique[1], a.unique[3], a.full[3]);
        t = __dmul_rn(th, tv);
    }
    for (int k = 0; k < a.k; ++k) {
        const float in = a.logits_nchw[((size_t)k * a.tile_h + y) * a.tile_w + x];
        float* out = a.blended + ((size_t)k * a.img_h + by) * a.img_w + bx;
        if (inside_unique) *out = in;
        else *out = (float)__dadd_rn((double)*out, __dmul_rn(t, (double)in));  // float += double * float
    }
}

// One launch for the tiles of a batch (BlendBatchArgs): blockIdx.z = tile.  Same arithmetic as blend_kernel — float += double * float without
// contraction — in the same order per pixel: tiles of EARLIER batches have been added by earlier launches, the batch's own in list order here.
__global__ __launch_bounds__(256) void blend_batch_kernel(BlendBatchArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, t = blockIdx.z;
    if (x >= a.tile_w) return;
    const int bx = a.left[t] + x, by = a.top[t] + y;
    if (by < a.full[t][1] || by > a.full[t][3] || by < 0 || by >= a.img_h) return;
    if (bx < a.full[t][0] || bx > a.full[t][2] || bx < 0 || bx >= a.img_w) return;
    const size_t tile_px = (size_t)a.tile_h * a.tile_w, img_px = (size_t)a.img_h * a.img_w;
    float* out = a.blended + (size_t)by * a.img_w + bx;
    if (bx >= a.unique[t][0] && bx <= a.unique[t][2] && by >= a.unique[t][1] && by <= a.unique[t][3]) {
        const float* in = a.logits + (size_t)t * a.k * tile_px + (size_t)y * a.tile_w + x;
        for (int k = 0; k < a.k; ++k) out[(size_t)k * img_px] = in[(size_t)k * tile_px];
        return;
    }
    // only the tiles whose full rectangles intersect this one's can cover the pixel (a.nbr[t]: at most eight on a regular tiling, against
    // `count` candidates per loop before: the frame path — 16 % of the pixels, a third of the waves — was most of this kernel's instructions)
    const unsigned near = a.nbr[t];
    for (unsigned e = near & ((1u << t) - 1u); e; e &= e - 1u) {   // an earlier tile of the batch covers this pixel: it gathers
        const int j = __builtin_ctz(e);
        if (bx >= a.full[j][0] && bx <= a.full[j][2] && by >= a.full[j][1] && by <= a.full[j][3]) return;
    }
    float acc[4];
    for (int k = 0; k < a.k; ++k) acc[k] = out[(size_t)k * img_px];
    for (unsigned e = (near | (1u << t)) & ~((1u << t) - 1u); e; e &= e - 1u) {   // this tile, then the later ones, in list order
        const int j = __builtin_ctz(e);
        if (!(bx >= a.full[j][0] && bx <= a.full[j][2] && by >= a.full[j][1] && by <= a.full[j][3])) continue;
        const double th = ramp(bx, a.full[j][0], a.unique[j][0], a.unique[j][2], a.full[j][2]);
        const double tv = ramp(by, a.full[j][1], a.unique[j][1], a.unique[j][3], a.full[j][3]);
        const double w = __dmul_rn(th, tv);
        const float* in = a.logits + (size_t)j * a.k * tile_px + (size_t)(by - a.top[j]) * a.tile_w + (bx - a.left[j]);
        for (int k = 0; k < a.k; ++k) acc[k] = (float)__dadd_rn((double)acc[k], __dmul_rn(w, (double)in[(size_t)k * tile_px]));
    }
    for (int k = 0; k < a.k; ++k) out[(size_t)k * img_px] = acc[k];
}

// head_blend (bf16 inference): logits = relu(bn(y)) . W + b for one tile pixel, blended straight into the resident planes.
// Four lanes share a pixel (8 channels = one 16-byte chunk each; a wave reads 1 KiB of contiguous NHWC per instruction) and
// combine their partial logits with two xor-shuffles, as head_train does; lane `sub` then blends class `sub`.
template <int KIND>
__global__ __launch_bounds__(256) void head_blend_kernel(HeadBlendArgs a) {
    constexpr int C = kHeadC, KM = kHeadKMax;
    const int K = a.k;
    const int sub = threadIdx.x & 3, c0 = sub * 8;
    const BlendArgs& b = a.blend;
    float w[8][KM], bias[KM], sa[8], ta[8], sb[8], tb[8];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        bias[k] = k < K ? a.bias[k] : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c][k] = k < K ? a.w_tm[(c0 + c) * K + k] : 0.f;
    }
    constexpr bool PRE = KIND == SRC_ACT || KIND == SRC_ACT2;   // the input still needs its producer's bn + relu (else: stored post-activation)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        sa[c] = PRE ? a.src.a_scale[c0 + c] : 0.f; ta[c] = PRE ? a.src.a_shift[c0 + c] : 0.f;
        sb[c] = KIND == SRC_ACT2 ? a.src.b_scale[c0 + c] : 0.f; tb[c] = KIND == SRC_ACT2 ? a.src.b_shift[c0 + c] : 0.f;
    }
    const bf16* xa = reinterpret_cast<const bf16*>(a.src.a);
    const bf16* xb = reinterpret_cast<const bf16*>(a.src.b);
    const int64_t pixels = (int64_t)b.tile_h * b.tile_w;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 2;
    const int f0 = (int)b.full[0], f1 = (int)b.full[1], f2 = (int)b.full[2], f3 = (int)b.full[3];
    const int u0 = (int)b.unique[0], u1 = (int)b.unique[1], u2 = (int)b.unique[2], u3 = (int)b.unique[3];
    const float inv_w = 1.0f / (float)b.tile_w;
    const bool small_tile = (int64_t)b.tile_h * b.tile_w <= ((int64_t)1 << 24);
    auto process = [&](int64_t p, const Raw8<bf16>& ra, const Raw8<bf16>& rb) __attribute__((always_inline)) {
        float x[8];
        raw_to_float(ra, x);
        if (PRE) {
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = relu_affine(x[c], sa[c], ta[c]);
        }
        if (KIND == SRC_ACT2 || KIND == SRC_SUM2) {
            float u[8];
            raw_to_float(rb, u);
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] += KIND == SRC_ACT2 ? relu_affine(u[c], sb[c], tb[c]) : u[c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = operand_round<bf16>(x[c]);
        float z[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc = fmaf(x[c], w[c][k], acc);
            acc += quad_perm<0xB1>(acc);   // lane ^ 1
            acc += quad_perm<0x4E>(acc);   // lane ^ 2
            z[k] = acc + bias[k];
        }
        // blend (annonet_infer.cpp:116-164): lane `sub` owns class `sub`
        // (a tile has < 2^31 pixels and its rectangles lie within int range: 32-bit arithmetic — the 64-bit division and compares
        // of the first version were a third of this kernel's instructions)
        const unsigned up = (unsigned)p, uw = (unsigned)b.tile_w;
        // p / tile_w without the 32-bit division sequence on tiles of up to 2^24 pixels: p is exact in fp32 and the float quotient lies
        // within one of the answer (error <= y * 2^-23 < 1 for tile_w >= 3; tile_w = 1, 2 are exact)
        int y, xx;
        if (small_tile) {
            y = (int)((float)up * inv_w);
            xx = (int)up - y * (int)uw;
            if (xx < 0) { --y; xx += (int)uw; } else if (xx >= (int)uw) { ++y; xx -= (int)uw; }
        } else { y = (int)(up / uw); xx = (int)(up - (unsigned)y * uw); }
        const int bx = b.tile_left + xx, by = b.tile_top + y;
        if (sub >= K || by < f1 || by > f3 || by < 0 || by >= b.img_h || bx < f0 || bx > f2 || bx < 0 || bx >= b.img_w) return;
        float in = z[0];
#pragma unroll
        for (int k = 1; k < KM; ++k) in = sub == k ? z[k] : in;
        float* out = b.blended + ((size_t)sub * b.img_h + by) * b.img_w + bx;
        const bool inside_unique = bx >= u0 && bx <= u2 && by >= u1 && by <= u3;
        if (inside_unique) *out = in;
        else {
            const double th = ramp(bx, b.full[0], b.unique[0], b.unique[2], b.full[2]);
            const double tv = ramp(by, b.full[1], b.unique[1], b.unique[3], b.full[3]);
            *out = (float)__dadd_rn((double)*out, __dmul_rn(__dmul_rn(th, tv), (double)in));  // float += double * float
        }
    };
    // four pixel records per thread in flight (unconditional loads from clamped indices), then the four are processed: the pass is a
    // stream of 64-byte records, and one dependent load -> compute -> store chain per iteration left it at a third of the HBM rate
    constexpr int U = 4;
    for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; p0 < pixels; p0 += U * stride) {
        Raw8<bf16> ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pc = min(p0 + u * stride, pixels - 1);
            ra[u] = raw_load8(xa + (size_t)pc * C + c0);
            if (KIND == SRC_ACT2 || KIND == SRC_SUM2) rb[u] = raw_load8(xb + (size_t)pc * C + c0); else rb[u] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (p0 + u * stride < pixels) process(p0 + u * stride, ra[u], rb[u]);   // the four lanes of a pixel agree on this branch
    }
}

// find_label (annonet_infer.cpp:170-185): strict '>' from -inf, start label 65535, gain added in double
__device__ __forceinline__ uint16_t find_label(const float* v, int k, const double* gains) {
    uint16_t label = ANH_LABEL_IGNORE;
    float best = -INFINITY;
    for (int c = 0; c < k; ++c) {
        const float value = (float)__dadd_rn((double)v[c], gains ? gains[c] : 0.0);
        if (value > best) { label = (uint16_t)c; best = value; }
    }
    return label;
}
// four pixels per thread while the class count allows (k <= 8: 16-byte plane loads, dword-aligned; one 8-byte label store)
typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned short u16x4a2 __attribute__((ext_vector_type(4), aligned(2)));
__global__ __launch_bounds__(256) void argmax_kernel(const float* blended, int k, int64_t pixels, int64_t p0, int64_t p1, const double* gains, uint16_t* labels) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (k <= 8) {
        double g[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = (gains && c < k) ? gains[c] : 0.0;
        for (int64_t p = p0 + 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); p < p1; p += 4 * stride) {
            if (p + 3 < p1) {
                float v[4][8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < k) {
                        const f32x4a4 q = *reinterpret_cast<const f32x4a4*>(blended + (size_t)c * pixels + p);
                        v[0][c] = q[0]; v[1][c] = q[1]; v[2][c] = q[2]; v[3][c] = q[3];
                    }
                u16x4a2 l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint16_t label = ANH_LABEL_IGNORE;
                    float best = -INFINITY;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < k) {
                            const float value = (float)__dadd_rn((double)v[i][c], g[c]);
                            if (value > best) { label = (uint16_t)c; best = value; }
                        }
                    l[i] = label;
                }
                *reinterpret_cast<u16x4a2*>(labels + p) = l;
            } else {
                for (int64_t q = p; q < p1; ++q) {
                    float v[8];
                    for (int c = 0; c < k; ++c) v[c] = blended[(size_t)c * pixels + q];
                    labels[q] = find_label(v, k, gains);
                }
            }
        }
        return;
    }
    for (int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < p1; p += stride) {
        uint16_t label = ANH_LABEL_IGNORE;
        float best = -INFINITY;
        for (int c = 0; c < k; ++c) {
            const double gain = gains ? gains[c] : 0.0;
            const float value = (float)__dadd_rn((double)blended[(size_t)c * pixels + p], gain);
            if (value > best) { label = (uint16_t)c; best = value; }
        }
        labels[p] = label;
    }
}

// ---------------------------------------------------------------------------------------------------
// detection-level filter (annonet_infer.cpp:187-239) on the device.  A blob = 8-connected pixels of one non-zero label; a blob
// without a seed pixel (class score above "clean" by more than its detection level) is relabelled 0.  Instead of labelling
// components, the seed property is flooded through them: 32x32 tiles (+1 halo) iterate in LDS until nothing changes inside
// the tile; the launch is repeated until no tile changed.  Flags only ever go 0 -> 1, so the in-place sweeps are race-benign.
// ---------------------------------------------------------------------------------------------------
// transposed = 0: the seed at (r, c) marks the blob that holds (r, c) — the intended behaviour.
// transposed = 1 (ANH_DET_SEED_REFERENCE_ORDER=1): the reference stores the seed as point(r, c) (annonet_infer.cpp:210) and looks the
//   blob up at (point.y(), point.x()) = (c, r) (:222): the seed marks the blob that holds the TRANSPOSED pixel.  Out-of-range transposed
//   positions (non-square images: undefined behaviour in the reference) are dropped.  flags must be zero on entry.
__global__ __launch_bounds__(256) void det_seed_kernel(const float* blended, const uint16_t* labels, int k, int64_t pixels, const double* det, uint8_t* flags,
                                                       int transposed, int h, int w) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += stride) {
        const uint16_t lab = labels[p];
        uint8_t f = 0;
        if (lab != 0 && lab != ANH_LABEL_IGNORE && lab < k) {
            const float clean = blended[p], mine = blended[(size_t)lab * pixels + p];
            if ((double)(mine - clean) > det[lab] - det[0]) f = 1;
        }
        if (!transposed) flags[p] = f;
        else if (f) {
            const int64_t r = p / w, c = p - r * w;
            if (c < h && r < w) flags[c * w + r] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void det_flood_kernel(const uint16_t* labels, uint8_t* flags, int h, int w, int* changed) {
    constexpr int T = 32, S = T + 2;
    __shared__ uint16_t lab[S][S];
    __shared__ uint8_t flg[S][S];
    const int x0 = blockIdx.x * T - 1, y0 = blockIdx.y * T - 1;
    for (int i = threadIdx.x; i < S * S; i += 256) {
        const int ly = i / S, lx = i - ly * S, y = y0 + ly, x = x0 + lx;
        const bool in = y >= 0 && y < h && x >= 0 && x < w;
        lab[ly][lx] = in ? labels[(size_t)y * w + x] : (uint16_t)0;   // label 0 never joins a blob
        flg[ly][lx] = in ? flags[(size_t)y * w + x] : (uint8_t)0;
    }
    __syncthreads();
    bool mine_changed = false;
    for (;;) {
        bool sweep = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + 256 * q, ly = 1 + (i >> 5), lx = 1 + (i & 31);
            const uint16_t l = lab[ly][lx];
            if (l != 0 && !flg[ly][lx]) {
                bool any = false;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) any = any || (lab[ly + dy][lx + dx] == l && flg[ly + dy][lx + dx]);
                if (any) { flg[ly][lx] = 1; sweep = true; }
            }
        }
        mine_changed = mine_changed || sweep;
        if (!__syncthreads_or(sweep)) break;
    }
    if (mine_changed) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = threadIdx.x + 256 * q, ly = 1 + (i >> 5), lx = 1 + (i & 31), y = y0 + ly, x = x0 + lx;
            if (y < h && x < w && flg[ly][lx]) flags[(size_t)y * w + x] = 1;
        }
        *changed = 1;
    }
}

__global__ __launch_bounds__(256) void det_apply_kernel(uint16_t* labels, const uint8_t* flags, int64_t pixels) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += stride)
        if (labels[p] != 0 && !flags[p]) labels[p] = 0;
}

}  // namespace

// Relabels every blob without a seed to 0, in place.  d_flags: H*W bytes of scratch; d_changed: one int; det: K doubles on the device.
void run_detection_filter(const float* d_blended, uint16_t* d_labels, int k, int h, int w, const double* d_det, uint8_t* d_flags, int* d_changed,
                          hipStream_t s) {
    const int64_t pixels = (int64_t)h * w;
    if (pixels == 0) return;
    const int blocks = (int)std::min<int64_t>((pixels + 255) / 256, 256 * 16);
    static const int reference_order = getenv("ANH_DET_SEED_REFERENCE_ORDER") ? atoi(getenv("ANH_DET_SEED_REFERENCE_ORDER")) : 0;
    if (reference_order) HIP_CHECK(hipMemsetAsync(d_flags, 0, (size_t)pixels, s));
    hipLaunchKernelGGL(det_seed_kernel, dim3(blocks), dim3(256), 0, s, d_blended, d_labels, k, pixels, d_det, d_flags, reference_order, h, w);
    HIP_CHECK(hipGetLastError());
    const dim3 grid((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32));
    for (int round = 0;; ++round) {
        ANH_REQUIRE(round < 100000, "detection filter did not converge");
        HIP_CHECK(hipMemsetAsync(d_changed, 0, sizeof(int), s));
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(det_flood_kernel, grid, dim3(256), 0, s, d_labels, d_flags, h, w, d_changed);
        HIP_CHECK(hipGetLastError());
        int changed = 0;
        HIP_CHECK(hipMemcpyAsync(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (!changed) break;
    }
    hipLaunchKernelGGL(det_apply_kernel, dim3(blocks), dim3(256), 0, s, d_labels, d_flags, pixels);
    HIP_CHECK(hipGetLastError());
}

// ===================================================================================================
// launchers
// ===================================================================================================
bool conv_f32_mfma_ok(const ConvArgs& a) { return conv_f32_mfma_selected(a); }

void launch_conv_generic(const ConvArgs& a, hipStream_t s) {
    const int64_t total = (int64_t)a.n * a.h_out * a.w_out;
    if (total == 0) return;
    if (conv_f32_mfma_selected(a)) {   // fp32 storage: the same k-ordered chains on v_mfma_f32_32x32x2_f32
        if (a.c_out > 64) launch_conv_f32_mfma_nt<4>(a, s);
        else if (a.c_out > 32) launch_conv_f32_mfma_nt<2>(a, s);
        else launch_conv_f32_mfma_nt<1>(a, s);
        return;
    }
    if (stem_forward_ok(a)) {
        const int tiles_x = (a.w_out + 31) / 32, tiles_y = (a.h_out + 7) / 8;
        const dim3 grid((unsigned)(tiles_x * tiles_y * a.n));
        const bool bf = a.out_dtype == DT_BF16;
        if (a.c_red == 3) {
            if (bf) hipLaunchKernelGGL((stem_forward_kernel<3, bf16>), grid, dim3(256), 0, s, a, tiles_x, tiles_y);
            else hipLaunchKernelGGL((stem_forward_kernel<3, float>), grid, dim3(256), 0, s, a, tiles_x, tiles_y);
        } else {
            if (bf) hipLaunchKernelGGL((stem_forward_kernel<1, bf16>), grid, dim3(256), 0, s, a, tiles_x, tiles_y);
            else hipLaunchKernelGGL((stem_forward_kernel<1, float>), grid, dim3(256), 0, s, a, tiles_x, tiles_y);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    const int groups = (a.c_out + 7) / 8;
    const int gy = std::min(groups, 4);
    dim3 block(64, gy), grid((unsigned)((total + 63) / 64), (groups + gy - 1) / gy);
    const bool in_bf16 = a.src.kind != SRC_IMAGE && a.src.dtype == DT_BF16;
    const bool out_bf16 = !a.out_nchw && a.out_dtype == DT_BF16;
    if (in_bf16) {
        if (out_bf16) conv_generic_dispatch<bf16, bf16>(a, grid, block, s);
        else conv_generic_dispatch<bf16, float>(a, grid, block, s);
    } else {
        if (out_bf16) conv_generic_dispatch<float, bf16>(a, grid, block, s);
        else conv_generic_dispatch<float, float>(a, grid, block, s);
    }
    HIP_CHECK(hipGetLastError());
}

int64_t wgrad_generic_scratch_floats(const WgradArgs& a) {
    if (stem_wgrad_ok(a)) return (int64_t)stem_wgrad_splits(a) * a.k * a.k * a.c_in * a.c_out;
    int splits, nci, nco; int64_t pps;
    wgrad_plan(a, splits, pps, nci, nco);
    return (int64_t)splits * a.k * a.k * a.c_in * a.c_out;
}

void launch_wgrad_generic(const WgradArgs& a, hipStream_t s) {
    if (stem_wgrad_ok(a)) {
        const int splits = stem_wgrad_splits(a);
        const int64_t nw = (int64_t)a.k * a.k * a.c_in * a.c_out;
        ANH_REQUIRE((int64_t)splits * nw <= a.partials_capacity, "wgrad scratch too small");
        const int tiles_x = (a.w_out + kStemTW - 1) / kStemTW, tiles_y = (a.h_out + kStemTH - 1) / kStemTH;
        const int total = tiles_x * tiles_y * a.n;
        const bool bf = a.dy_dtype == DT_BF16;
        if (a.c_in == 3) {
            if (bf) hipLaunchKernelGGL((wgrad_stem_kernel<3, bf16>), dim3(splits), dim3(320), 0, s, a, tiles_x, tiles_y, total, splits);
            else hipLaunchKernelGGL((wgrad_stem_kernel<3, float>), dim3(splits), dim3(320), 0, s, a, tiles_x, tiles_y, total, splits);
        } else {
            if (bf) hipLaunchKernelGGL((wgrad_stem_kernel<1, bf16>), dim3(splits), dim3(128), 0, s, a, tiles_x, tiles_y, total, splits);
            else hipLaunchKernelGGL((wgrad_stem_kernel<1, float>), dim3(splits), dim3(128), 0, s, a, tiles_x, tiles_y, total, splits);
        }
        HIP_CHECK(hipGetLastError());
        if (a.splits_out) *a.splits_out = splits; else launch_reduce_partials(a.partials, splits, nw, a.dw, s);
        HIP_CHECK(hipGetLastError());
        return;
    }
    int splits, nci, nco; int64_t pps;
    wgrad_plan(a, splits, pps, nci, nco);
    const int64_t nw = (int64_t)a.k * a.k * a.c_in * a.c_out;
    ANH_REQUIRE((int64_t)splits * nw <= a.partials_capacity, "wgrad scratch too small");
    dim3 grid(splits, a.k * a.k, nci * nco);
    const bool in_bf16 = a.src.kind != SRC_IMAGE && a.src.dtype == DT_BF16;
    const bool dy_bf16 = a.dy_dtype == DT_BF16;
    if (in_bf16) {
        if (dy_bf16) wgrad_generic_dispatch<bf16, bf16>(a, grid, pps, nco, s);
        else wgrad_generic_dispatch<bf16, float>(a, grid, pps, nco, s);
    } else {
        if (dy_bf16) wgrad_generic_dispatch<float, bf16>(a, grid, pps, nco, s);
        else wgrad_generic_dispatch<float, float>(a, grid, pps, nco, s);
    }
    HIP_CHECK(hipGetLastError());
    if (a.splits_out) *a.splits_out = splits; else launch_reduce_partials(a.partials, splits, nw, a.dw, s);
    HIP_CHECK(hipGetLastError());
}

void launch_reduce_partials(const float* partials, int splits, int64_t nw, float* out, hipStream_t s) {
    if (nw < (int64_t)256 * 256 && splits >= 64)   // fewer than one wide workgroup per CU: the narrow form (a fixed order of its own: chosen by the shape alone)
        hipLaunchKernelGGL((reduce_partials_kernel<16, 64>), dim3((unsigned)((nw + 63) / 64)), dim3(16, 64), 0, s, partials, splits, nw, out);
    else
        hipLaunchKernelGGL((reduce_partials_kernel<64, 16>), dim3((unsigned)((nw + 255) / 256)), dim3(64, 16), 0, s, partials, splits, nw, out);
    HIP_CHECK(hipGetLastError());
}

bool head_train_supported(const HeadTrainArgs& a) {
    return a.c_in == kHeadC && a.k >= 1 && a.k <= kHeadKMax && (a.src.kind == SRC_ACT || a.src.kind == SRC_ACT2);
}
// 768 = 3 workgroups per CU (the two-class kernel fits 3 waves per SIMD): one resident round, so the per-workgroup
// reduction tail (about 50 double wave-folds per thread) is paid once per CU slot and not once per 12 pixels.
int head_train_blocks(int64_t pixels) {
    // two workgroups per CU are resident (194-226 VGPRs): 512 is ONE round of them on 256 CUs; every further workgroup pays the closing
    // fold again (43 double-precision wave reductions + its row of partials).  Measured 256 / 384 / 512 / 640 / 768 / 1024 workgroups:
    // 89 / 82 / 68 / 85 / 80 / 78 us
    static const int cap = getenv("ANH_HEAD_BLOCKS") ? std::max(1, atoi(getenv("ANH_HEAD_BLOCKS"))) : 512;
    return (int)std::max<int64_t>(1, std::min<int64_t>((pixels * 4 + 255) / 256, cap));
}
int64_t head_train_partial_doubles(const HeadTrainArgs& a) { return (int64_t)head_train_blocks(a.pixels) * (1 + a.k + kHeadC * a.k); }

void launch_head_train(const HeadTrainArgs& t, hipStream_t s) {
    ANH_REQUIRE(head_train_supported(t), "head_train: unsupported shape");
    HeadArgs a;
    a.src = t.src; a.w_tm = t.w_tm; a.w_km = t.w_km; a.bias = t.bias; a.labels = t.labels; a.weights = t.weights;
    a.logits = t.logits; a.da = t.da; a.dlogits = t.dlogits; a.pixels = t.pixels; a.k = t.k; a.scale = t.scale; a.partials = t.partials; a.error_flag = t.error_flag;
    a.bn_mean = t.bnred_mean; a.bn_invstd = t.bnred_invstd; a.bn_partials = t.src.kind == SRC_ACT ? t.bnred_partials : nullptr;
    a.bn_acc = t.src.kind == SRC_ACT ? t.bnred_acc : nullptr;
    a.bn_finish = t.bnred_finish;
    ANH_REQUIRE(!a.bn_acc || t.src.a_tab.acc, "head_train: table-mode sums need the input layer's accumulator table");
    const int blocks = head_train_blocks(t.pixels);
    const bool bf = t.src.dtype == DT_BF16;
    BnFoldJobs jobs;
    if (t.fold) jobs = *t.fold;
    ANH_REQUIRE(jobs.n <= blocks, "head_train: more fold jobs than workgroups");
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, s, a, jobs); };
    auto pick = [&](auto tag_t, auto tag_kind) {
        using T = decltype(tag_t); constexpr int KIND = decltype(tag_kind)::value;
        switch (t.k) {
            case 2: launch(head_train_kernel<T, KIND, 2, true>); break;
            case 3: launch(head_train_kernel<T, KIND, 3, true>); break;
            case 4: launch(head_train_kernel<T, KIND, 4, true>); break;
            default: launch(head_train_kernel<T, KIND, 2, false>); break;   // one class: the two-slot body with a run-time count
        }
    };
    if (t.src.kind == SRC_ACT) { if (bf) pick(bf16{}, std::integral_constant<int, SRC_ACT>{}); else pick(float{}, std::integral_constant<int, SRC_ACT>{}); }
    else { if (bf) pick(bf16{}, std::integral_constant<int, SRC_ACT2>{}); else pick(float{}, std::integral_constant<int, SRC_ACT2>{}); }
    HIP_CHECK(hipGetLastError());
    const int slots = 1 + t.k + kHeadC * t.k;
    hipLaunchKernelGGL(head_finalize_kernel, dim3(slots), dim3(64), 0, s, t.partials, blocks, slots, t.k, t.loss_out, t.loss_out_f32, t.dbias, t.dw);
    HIP_CHECK(hipGetLastError());
}

bool bn_table_mode_ok(int c) { return bn_vec_ok(c) && c <= 2048; }

int bn_partial_blocks(int64_t pixels) { const int ppb = bn_pixels_per_block(pixels); return (int)((pixels + ppb - 1) / ppb); }

static dim3 bn_block(int c) {
    int cx = 1;
    while (cx < c && cx < 64) cx <<= 1;
    return dim3(cx, 256 / cx);
}

void launch_bn_forward_finalize(const BnFwdArgs& a, int blocks, hipStream_t s);
// statistics pass alone (partials only); returns the number of partials per channel for launch_bn_forward_finalize
int launch_bn_forward_partials(const BnFwdArgs& a, hipStream_t s) {
    const int blocks = bn_partial_blocks(a.pixels);
    if (bn_vec_ok(a.c)) {
        if (a.dtype == DT_BF16) hipLaunchKernelGGL(bn_stats_vec_kernel<bf16>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16*>(a.y), a.pixels, a.c, a.partials, bn_pixels_per_block(a.pixels));
        else hipLaunchKernelGGL(bn_stats_vec_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(a.y), a.pixels, a.c, a.partials, bn_pixels_per_block(a.pixels));
    } else {
        const dim3 block = bn_block(a.c);
        const size_t shmem = (size_t)block.y * a.c * 2 * sizeof(float);
        if (a.dtype == DT_BF16)
            hipLaunchKernelGGL(bn_stats_kernel<bf16>, dim3(blocks), block, shmem, s, reinterpret_cast<const bf16*>(a.y), a.pixels, a.c, a.partials, bn_pixels_per_block(a.pixels));
        else
            hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(blocks), block, shmem, s, reinterpret_cast<const float*>(a.y), a.pixels, a.c, a.partials, bn_pixels_per_block(a.pixels));
    }
    HIP_CHECK(hipGetLastError());
    return blocks;
}

void launch_bn_forward_stats(const BnFwdArgs& a, hipStream_t s) { launch_bn_forward_finalize(a, launch_bn_forward_partials(a, s), s); }

void launch_bn_forward_finalize(const BnFwdArgs& a, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(a.c), dim3(64), 0, s, a.partials, blocks, a.pixels, a.c, a.gamma, a.beta, a.eps,
                       a.mean, a.invstd, a.scale, a.shift, a.var, a.running_mean, a.running_var, a.averaging_factor, a.unbias);
    HIP_CHECK(hipGetLastError());
}

void launch_bn_fold_all(const BnFoldJobs& jobs, hipStream_t s) {
    if (jobs.n == 0) return;
    hipLaunchKernelGGL(bn_fold_all_kernel, dim3(jobs.n), dim3(128), 0, s, jobs);
    HIP_CHECK(hipGetLastError());
}

void launch_bn_running_update(const float* mean, const double* var, float* running_mean, float* running_var, int c,
                              double averaging_factor, double unbias, hipStream_t s) {
    hipLaunchKernelGGL(bn_running_kernel, dim3((c + 63) / 64), dim3(64), 0, s, mean, var, running_mean, running_var, c, averaging_factor, unbias);
    HIP_CHECK(hipGetLastError());
}

// The three passes of batch-norm + relu backward; launch_bn_backward runs them in order.
void launch_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s) {
    const int blocks = bn_partial_blocks(a.pixels);
    const bool bf = a.dtype == DT_BF16;
    BnBwdFinish fin = a.finish;
    fin.acc = a.acc;   // null: partials + finalize kernel
    if (bn_vec_ok(a.c)) {
        if (bf) hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel<bf16>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16*>(a.da), reinterpret_cast<const bf16*>(a.y),
                                   a.pixels, a.c, a.mean, a.invstd, a.scale, a.shift, a.partials, bn_pixels_per_block(a.pixels), fin);
        else hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(a.da), reinterpret_cast<const float*>(a.y),
                                a.pixels, a.c, a.mean, a.invstd, a.scale, a.shift, a.partials, bn_pixels_per_block(a.pixels), fin);
    } else {
        ANH_REQUIRE(!a.acc, "bn_bwd_reduce: table mode needs the vectorised kernel");
        const dim3 block = bn_block(a.c);
        const size_t shmem = (size_t)block.y * a.c * 2 * sizeof(float);
        if (bf) hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16>, dim3(blocks), block, shmem, s, reinterpret_cast<const bf16*>(a.da), reinterpret_cast<const bf16*>(a.y),
                                   a.pixels, a.c, a.mean, a.invstd, a.scale, a.shift, a.partials, bn_pixels_per_block(a.pixels));
        else hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), block, shmem, s, reinterpret_cast<const float*>(a.da), reinterpret_cast<const float*>(a.y),
                                a.pixels, a.c, a.mean, a.invstd, a.scale, a.shift, a.partials, bn_pixels_per_block(a.pixels));
    }
    HIP_CHECK(hipGetLastError());
}

void launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a.c), dim3(64), 0, s, a.partials, a.partial_blocks > 0 ? a.partial_blocks : bn_partial_blocks(a.pixels), a.pixels, a.c, a.gamma, a.invstd,
                       a.dgamma, a.dbeta, a.coef);
    HIP_CHECK(hipGetLastError());
}

static int apply_block_cap() {   // 10 workgroups per CU; 1024 / 1280 / 2048 / 2304 / 2560 / 3072 / 4096 / 8192: +0.5 / +0.2 / 0 / -0.2 / -0.4 / 0 / +0.5 / +3 % of the step
    static const int cap = getenv("ANH_APPLY_BLOCKS") ? std::max(1, atoi(getenv("ANH_APPLY_BLOCKS"))) : 256 * 10;
    return cap;
}

void launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s) {
    const int64_t total = a.pixels * a.c;
    const bool bf = a.dtype == DT_BF16;
    void* out = a.dy_out ? a.dy_out : a.da;
    if (a.head_g) {   // da of the layer under the fused head is recomputed from the head's dlogits
        ANH_REQUIRE(a.c == kHeadC && a.head_k >= 1 && a.head_k <= kHeadKMax && out != nullptr, "bn_bwd_apply: head form needs 32 channels and at most 4 classes");
        const int64_t chunks = total / 8;
        const int blocks = (int)std::min<int64_t>((chunks + 255) / 256, apply_block_cap());
        auto go = [&](auto kernel, auto tag) {
            using T = decltype(tag);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, s, a.head_g, a.head_w_tm, a.head_k, reinterpret_cast<T*>(out), reinterpret_cast<const T*>(a.y), chunks,
                               a.mean, a.invstd, a.scale, a.shift, a.coef);
        };
        if (a.head_k <= 2) { if (bf) go(bn_bwd_apply_head_kernel<bf16, 2>, bf16{}); else go(bn_bwd_apply_head_kernel<float, 2>, float{}); }
        else { if (bf) go(bn_bwd_apply_head_kernel<bf16, 4>, bf16{}); else go(bn_bwd_apply_head_kernel<float, 4>, float{}); }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (bn_vec_ok(a.c)) {
        const int64_t chunks = total / 8;
        const int apply_blocks = (int)std::min<int64_t>((chunks + 255) / 256, apply_block_cap());  // 256 threads: a multiple of every group count
        if (bf) hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<bf16>, dim3(apply_blocks), dim3(256), 0, s, reinterpret_cast<const bf16*>(a.da), reinterpret_cast<bf16*>(out),
                                   reinterpret_cast<const bf16*>(a.y), chunks, a.c, a.mean, a.invstd, a.scale, a.shift, a.coef);
        else hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<float>, dim3(apply_blocks), dim3(256), 0, s, reinterpret_cast<const float*>(a.da), reinterpret_cast<float*>(out),
                                reinterpret_cast<const float*>(a.y), chunks, a.c, a.mean, a.invstd, a.scale, a.shift, a.coef);
    } else {
        const int apply_blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
        if (bf) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, dim3(apply_blocks), dim3(256), 0, s, reinterpret_cast<const bf16*>(a.da), reinterpret_cast<bf16*>(out),
                                   reinterpret_cast<const bf16*>(a.y), total, a.c, a.mean, a.invstd, a.scale, a.shift, a.coef);
        else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(apply_blocks), dim3(256), 0, s, reinterpret_cast<const float*>(a.da), reinterpret_cast<float*>(out),
                                reinterpret_cast<const float*>(a.y), total, a.c, a.mean, a.invstd, a.scale, a.shift, a.coef);
    }
    HIP_CHECK(hipGetLastError());
}

void launch_bn_backward(const BnBwdArgs& a, hipStream_t s) {
    launch_bn_bwd_reduce(a, s);
    launch_bn_bwd_finalize(a, s);
    launch_bn_bwd_apply(a, s);
}

int loss_partial_blocks(int64_t pixels) { return (int)((pixels + kLossPixelsPerBlock - 1) / kLossPixelsPerBlock); }

void launch_loss(const LossArgs& a, hipStream_t s) {
    ANH_REQUIRE(a.k >= 1 && a.k <= 64, "class count out of range for the loss kernel");
    const int blocks = loss_partial_blocks(a.pixels);
    if (a.k <= 4) hipLaunchKernelGGL(loss_kernel<4>, dim3(blocks), dim3(256), 0, s, a);
    else if (a.k <= 8) hipLaunchKernelGGL(loss_kernel<8>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(loss_kernel<64>, dim3(blocks), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(a.k + 1), dim3(64), 0, s, a.partials, blocks, a.k, a.loss_out, a.loss_out_f32, a.dbias);
    HIP_CHECK(hipGetLastError());
}

void launch_sgd(const SgdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((a.n_params + 255) / 256)), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

void launch_tm_to_canonical(const ParamSegment* segments, int n_segments, int64_t n_params, const float* tm, float* canonical, hipStream_t s) {
    hipLaunchKernelGGL(tm_to_canonical_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, segments, n_segments, n_params, tm, canonical);
    HIP_CHECK(hipGetLastError());
}

bool head_blend_supported(const HeadBlendArgs& a) {
    static const bool on = !(getenv("ANH_FUSE_HEAD_BLEND") && atoi(getenv("ANH_FUSE_HEAD_BLEND")) == 0);
    return on && a.c_in == kHeadC && a.k >= 1 && a.k <= kHeadKMax && a.src.dtype == DT_BF16 &&
           (a.src.kind == SRC_ACT || a.src.kind == SRC_ACT2 || a.src.kind == SRC_RAW || a.src.kind == SRC_SUM2);
}
void launch_head_blend(const HeadBlendArgs& a, hipStream_t s) {
    ANH_REQUIRE(head_blend_supported(a), "head_blend: unsupported shape");
    const int64_t pixels = (int64_t)a.blend.tile_h * a.blend.tile_w;
    if (pixels <= 0) return;
    // ONE round of resident workgroups (94-115 VGPRs: five per CU on 256 CUs).  4096^2 tiled inference, 1024 / 1280 / 2048 / 2560 / 3840
    // workgroups: 3,799 / 3,838 / 3,759 / 3,801 / 3,763 Mpx/s
    static const int cap = getenv("ANH_HEAD_BLEND_BLOCKS") ? std::max(1, atoi(getenv("ANH_HEAD_BLEND_BLOCKS"))) : 1280;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((pixels * 4 + 255) / 256, cap));
    switch (a.src.kind) {
        case SRC_ACT: hipLaunchKernelGGL(head_blend_kernel<SRC_ACT>, dim3(blocks), dim3(256), 0, s, a); break;
        case SRC_ACT2: hipLaunchKernelGGL(head_blend_kernel<SRC_ACT2>, dim3(blocks), dim3(256), 0, s, a); break;
        case SRC_RAW: hipLaunchKernelGGL(head_blend_kernel<SRC_RAW>, dim3(blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(head_blend_kernel<SRC_SUM2>, dim3(blocks), dim3(256), 0, s, a); break;
    }
    HIP_CHECK(hipGetLastError());
}

// The kernel's shortcut — a pixel inside a tile's unique rectangle is assigned by that tile and by nobody else — holds when no OTHER tile's
// full rectangle reaches into a unique rectangle, which is what tiling::get_tiles produces (the reference asserts it: `out == 0.f` before the
// assignment, annonet_infer.cpp:158).  A caller's own tile list may break it; such a batch takes the per-tile launches, which are the
// reference's loop literally.
bool blend_batch_ok(const BlendBatchArgs& a) {
    if (a.count > 16 || a.k < 1 || a.k > 4) return false;
    for (int i = 0; i < a.count; ++i)
        for (int j = 0; j < a.count; ++j)
            if (j != i && a.unique[i][0] <= a.full[j][2] && a.full[j][0] <= a.unique[i][2] && a.unique[i][1] <= a.full[j][3] && a.full[j][1] <= a.unique[i][3]) return false;
    return true;
}

void launch_blend_batch(BlendBatchArgs a, hipStream_t s) {
    if (a.tile_w <= 0 || a.tile_h <= 0 || a.count <= 0) return;
    ANH_REQUIRE(a.count <= 16 && a.k >= 1 && a.k <= 4, "blend_batch: at most 16 tiles of at most 4 classes");
    for (int i = 0; i < a.count; ++i) {
        a.nbr[i] = 0;
        for (int j = 0; j < a.count; ++j)
            if (j != i && a.full[i][0] <= a.full[j][2] && a.full[j][0] <= a.full[i][2] && a.full[i][1] <= a.full[j][3] && a.full[j][1] <= a.full[i][3]) a.nbr[i] |= 1u << j;
    }
    hipLaunchKernelGGL(blend_batch_kernel, dim3((a.tile_w + 255) / 256, a.tile_h, a.count), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

void launch_blend(const BlendArgs& a, hipStream_t s) {
    if (a.tile_w <= 0 || a.tile_h <= 0) return;
    hipLaunchKernelGGL(blend_kernel, dim3((a.tile_w + 255) / 256, a.tile_h), dim3(256), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

void launch_argmax(const float* blended, int k, int64_t pixels, const double* gains_or_null, uint16_t* labels, hipStream_t s) {
    launch_argmax_range(blended, k, pixels, 0, pixels, gains_or_null, labels, s);
}
// labels of the pixels [p0, p1) of planes that hold `pixels` pixels each
void launch_argmax_range(const float* blended, int k, int64_t pixels, int64_t p0, int64_t p1, const double* gains_or_null, uint16_t* labels, hipStream_t s) {
    if (p1 <= p0) return;
    const int64_t per = k <= 8 ? 1024 : 256;   // pixels per workgroup and pass
    const int blocks = (int)std::min<int64_t>((p1 - p0 + per - 1) / per, 256 * 16);
    hipLaunchKernelGGL(argmax_kernel, dim3(blocks), dim3(256), 0, s, blended, k, pixels, p0, p1, gains_or_null, labels);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// training crops from HBM-resident full images
// ---------------------------------------------------------------------------------------------------
// One workgroup walks a strip of one crop in UNFLIPPED row-major order (that is the order set_weights scans in, and the
// order the first-occurrence positions refer to); flips only move where a pixel is written.
__global__ __launch_bounds__(256) void crop_pixels_kernel(const CropSource* specs, int dim, int channels, uint8_t* out_img, uint16_t* out_lab,
                                                          unsigned* hist, unsigned* firstpos, int* bad_label, int classes) {
    __shared__ unsigned sh_hist[kCropMaxClasses], sh_first[kCropMaxClasses];
    const int crop = blockIdx.y;
    const CropSource sp = specs[crop];
    for (int i = threadIdx.x; i < kCropMaxClasses; i += blockDim.x) { sh_hist[i] = 0; sh_first[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const int plane = dim * dim;
    uint8_t* oi = out_img + (size_t)crop * plane * channels;
    uint16_t* ol = out_lab + (size_t)crop * plane;
    const double gain = sp.gain;
    const bool resize = sp.src_dim != dim;
    // dlib::resize_image's corner-aligned grid [UPSTREAM-UNVERIFIED, restated in annonet_host.h and the oracle]: output (r, c) samples
    // the src_dim x src_dim chip at (r, c) * (src_dim - 1) / max(dim - 1, 1)
    const double scale = (double)(sp.src_dim - 1) / (double)max(dim - 1, 1);
    auto chip_label = [&](int yy, int xx) -> uint16_t {   // the chip's label image: "ignore" outside the full image (:150-158)
        const int sy = sp.top + yy, sx = sp.left + xx;
        const bool inside = sy >= 0 && sy < sp.height && sx >= 0 && sx < sp.width;
        return inside ? sp.labels[(size_t)sy * sp.width + sx] : (uint16_t)ANH_LABEL_IGNORE;
    };
    auto chip_pixel = [&](int yy, int xx, int ch) -> uint8_t {   // the chip after outpaint: clamp to the nearest image pixel
        const int cy = min(max(sp.top + yy, 0), sp.height - 1), cx = min(max(sp.left + xx, 0), sp.width - 1);
        return sp.image[((size_t)cy * sp.width + cx) * channels + ch];
    };
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += gridDim.x * blockDim.x) {
        const int r = p / dim, c = p - r * dim;
        uint16_t label;
        int y0 = r, x0 = c, y1 = r, x1 = c;
        float fy = 0.f, fx = 0.f;
        if (!resize) label = chip_label(r, c);
        else {
            const double y = r * scale, x = c * scale;
            label = chip_label((int)floor(y + 0.5), (int)floor(x + 0.5));   // interpolate_nearest_neighbor
            y0 = (int)floor(y); x0 = (int)floor(x);
            y1 = min(y0 + 1, sp.src_dim - 1); x1 = min(x0 + 1, sp.src_dim - 1);
            fy = (float)(y - y0); fx = (float)(x - x0);
        }
        if (label != ANH_LABEL_IGNORE) {
            if (label < classes) { atomicAdd(&sh_hist[label], 1u); atomicMin(&sh_first[label], (unsigned)p); }
            else *bad_label = 1;
        }
        const int fr = sp.flip_ud ? dim - 1 - r : r, fc = sp.flip_lr ? dim - 1 - c : c;
        const size_t dst = (size_t)fr * dim + fc;
        ol[dst] = label;
        for (int ch = 0; ch < channels; ++ch) {
            int v;
            if (!resize) v = chip_pixel(r, c, ch);
            else {   // interpolate_bilinear in float, rounded half up
                const float tl = chip_pixel(y0, x0, ch), tr = chip_pixel(y0, x1, ch), bl = chip_pixel(y1, x0, ch), br = chip_pixel(y1, x1, ch);
                const float top = (1.f - fx) * tl + fx * tr, bot = (1.f - fx) * bl + fx * br;
                v = (int)((1.f - fy) * top + fy * bot + 0.5f);
            }
            // tuc::round<unsigned char>(tuc::clamp(value * change, 0.0, 255.0)) (annonet_train_main.cpp:203-205), in double
            if (gain != 1.0) v = (int)floor(fmin(fmax((double)v * gain, 0.0), 255.0) + 0.5);
            if (sp.noise_level > 0) v = min(max(v + crop_noise_draw(sp.noise_seed, (unsigned long long)(dst * channels + ch), sp.noise_level), 0), 255);
            if (channels == 3 && sp.color_offset[ch] != 0) v = min(max(v + sp.color_offset[ch], 0), 255);
            oi[dst * channels + ch] = (uint8_t)v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < classes; i += blockDim.x) {   // integer atomics: the sums do not depend on the order
        if (sh_hist[i]) { atomicAdd(&hist[(size_t)crop * kCropMaxClasses + i], sh_hist[i]); atomicMin(&firstpos[(size_t)crop * kCropMaxClasses + i], sh_first[i]); }
    }
}

__global__ __launch_bounds__(256) void crop_weights_kernel(const uint16_t* labels, const float* table, int plane, float* weights) {
    const int crop = blockIdx.y;
    const uint16_t* l = labels + (size_t)crop * plane;
    float* w = weights + (size_t)crop * plane;
    const float* t = table + (size_t)crop * kCropMaxClasses;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += gridDim.x * blockDim.x) {
        const uint16_t v = l[p];
        w[p] = (v == ANH_LABEL_IGNORE || v >= kCropMaxClasses) ? 0.f : t[v];
    }
}

void launch_crop_pixels(const CropSource* d_specs, int n, int dim, int channels, uint8_t* d_images, uint16_t* d_labels,
                        unsigned* d_hist, unsigned* d_firstpos, int* d_bad_label, int classes, hipStream_t s) {
    ANH_REQUIRE(classes >= 1 && classes <= kCropMaxClasses, "crop: class count out of range");
    HIP_CHECK(hipMemsetAsync(d_hist, 0, (size_t)n * kCropMaxClasses * sizeof(unsigned), s));
    HIP_CHECK(hipMemsetAsync(d_firstpos, 0xFF, (size_t)n * kCropMaxClasses * sizeof(unsigned), s));
    const int blocks = std::max(1, std::min((dim * dim + 255) / 256, 64));
    hipLaunchKernelGGL(crop_pixels_kernel, dim3(blocks, n), dim3(256), 0, s, d_specs, dim, channels, d_images, d_labels, d_hist, d_firstpos, d_bad_label, classes);
    HIP_CHECK(hipGetLastError());
}

void launch_crop_weights(const uint16_t* d_labels, const float* d_table, int n, int dim, float* d_weights, hipStream_t s) {
    const int blocks = std::max(1, std::min((dim * dim + 255) / 256, 64));
    hipLaunchKernelGGL(crop_weights_kernel, dim3(blocks, n), dim3(256), 0, s, d_labels, d_table, dim * dim, d_weights);
    HIP_CHECK(hipGetLastError());
}

// Zero fill as a streaming kernel: the runtime's fill (hipMemsetAsync -> fillBufferAligned) clears the 201 MB of class planes of a
// 4096^2 image in ~560 us (0.36 TB/s, profiles/r03_infer_kernel_stats.csv) — a seventh of that image's time; 16-byte stores from
// 2048 workgroups run at the HBM write rate.
namespace {
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4* p, size_t n16, unsigned char* tail, int tail_bytes) {
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < tail_bytes) tail[threadIdx.x] = 0;
}
}  // namespace

// zero the pixels of `n` rectangles (inclusive, inside the image) in every one of k planes [k][H][W]: one workgroup per (rectangle, plane, slice)
namespace {
__global__ __launch_bounds__(256) void zero_rects_kernel(float* planes, int H, int W, const anh_rect* rects) {
    const anh_rect r = rects[blockIdx.x];
    const int w = (int)(r.right - r.left + 1), h = (int)(r.bottom - r.top + 1);
    if (w <= 0 || h <= 0) return;
    float* base = planes + ((size_t)blockIdx.y * H + r.top) * W + r.left;
    for (int i = (int)blockIdx.z * 256 + (int)threadIdx.x; i < w * h; i += (int)gridDim.z * 256) {
        const int y = i / w, x = i - y * w;
        base[(size_t)y * W + x] = 0.f;
    }
}
}  // namespace
void launch_zero_rects(float* planes, int k, int H, int W, const anh_rect* d_rects, int n, hipStream_t s) {
    if (n <= 0 || k <= 0) return;
    hipLaunchKernelGGL(zero_rects_kernel, dim3((unsigned)n, (unsigned)k, 8), dim3(256), 0, s, planes, H, W, d_rects);
    HIP_CHECK(hipGetLastError());
}

void launch_fill_zero(void* p, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (bytes < (size_t)1 << 20 || (reinterpret_cast<uintptr_t>(p) & 15)) { HIP_CHECK(hipMemsetAsync(p, 0, bytes, s)); return; }
    const size_t n16 = bytes >> 4;
    const int blocks = (int)std::min<size_t>(2048, (n16 + 255) / 256);
    hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n16, reinterpret_cast<unsigned char*>(p) + (n16 << 4), (int)(bytes & 15));
    HIP_CHECK(hipGetLastError());
}

}  // namespace anh
