// kernels_mfma.hip — bf16 MFMA implicit-GEMM kernels for the dense 3x3 convolutions (gfx950 / CDNA4).
//
// conv3x3s1_mfma: 3x3, stride 1, pad 1 — the forward of enc/dec layers and (with the taps flipped) their
// backward-data.  One workgroup = 4 waves = an 8x32 tile of output pixels x all output channels.
//   * The (8+2)x(32+2) input patch of a 32-channel slab is read from HBM ONCE (16 B per lane, the producer's
//     bn+relu(+skip add) applied in registers), rounded to bf16 and laid down in LDS as 64-byte pixel records;
//     all 9 taps then read it from LDS — no im2col expansion ever reaches memory.
//   * The 16-byte chunk index inside a record is XOR-ed with (record>>2)&3: the 16 lanes of a ds_read_b128
//     group hit 16 distinct 16-byte slots of the 256-byte bank row (conflict-free for every tap offset).
//   * Weights ([tap][c_out][c_red] bf16, k contiguous) are staged the same way, 9 or 3 taps at a time.
//   * MFMA 32x32x16 bf16, A = weights (rows = output channel), B = pixels (cols = pixel): the accumulator puts a
//     pixel on the lane and 4 consecutive channels in 4 registers, so the epilogue packs to bf16, swaps halves with
//     v_permlane32_swap and stores 16 B per lane straight to NHWC — no LDS transpose.
//   * fp32 accumulate; 2 workgroups per CU (<= 59 KB LDS, <= 256 VGPR) overlap one tile's staging with the other's MFMAs.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <vector>

#include "bnacc.h"
#include "common.h"
#include "kernels.h"

namespace anh {

void ensure_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return;   // the default limit
    static std::mutex mu;   // handles may be driven from different threads
    static std::unordered_map<const void*, size_t> configured[64];   // the attribute is per (device, kernel)
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = configured[dev & 63][kernel];
    if (bytes > have) {
        HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
}

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int TH = 8, TW = 32;            // output tile (rows x cols); one wave = 2 rows = 2 MFMA pixel groups
constexpr int PH = TH + 2, PW = TW + 2;   // input patch
constexpr int PATCH_PIX = PH * PW;        // 340 pixel records of 64 B
constexpr int X_BYTES = PATCH_PIX * 64;   // 21,760

// two fp32 -> one packed bf16 pair, round to nearest even (element-wise conversions: the one-instruction vector conversion measured
// slower in round 4, profiles/r04_ab_log.txt)
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    bf16x2 t;
    t[0] = (bf16)lo; t[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ uint4 load16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void store16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ float relu_affine(float y, float s, float t) {
    const float z = fmaf(y, s, t);
    return z > 0.f ? z : 0.f;
}

// relu of two packed bf16: rounding to bf16 keeps the sign, so relu(round(z)) == round(relu(z)) bit for bit, and on the packed pair
// the relu is ONE v_pk_max_i16 against zero (a negative bf16, -0 included, is a negative int16)
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ unsigned relu_bf16x2(unsigned packed) {
    const s16x2 zero = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, packed), zero));
}
// one chunk through a single producer's bn + relu: affine in fp32, pack, relu on the packed pairs (5 VALU per pair instead of 7)
__device__ __forceinline__ uint4 affine_relu_pack8(const uint4& raw, const float* sc, const float* sh) {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = relu_bf16x2(pack2(fmaf(lo_f(w[i]), sc[2 * i], sh[2 * i]), fmaf(hi_f(w[i]), sc[2 * i + 1], sh[2 * i + 1])));
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// 8 bf16 (one 16-byte chunk) through the producer's bn+relu
__device__ __forceinline__ void affine8(const uint4& raw, const float* sc, const float* sh, float v[8]) {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = relu_affine(lo_f(w[i]), sc[2 * i], sh[2 * i]);
        v[2 * i + 1] = relu_affine(hi_f(w[i]), sc[2 * i + 1], sh[2 * i + 1]);
    }
}


__device__ __forceinline__ uint4 add_bf16x8(const uint4& p, const uint4& q) {
    return make_uint4(pack2(lo_f(p.x) + lo_f(q.x), hi_f(p.x) + hi_f(q.x)), pack2(lo_f(p.y) + lo_f(q.y), hi_f(p.y) + hi_f(q.y)),
                      pack2(lo_f(p.z) + lo_f(q.z), hi_f(p.z) + hi_f(q.z)), pack2(lo_f(p.w) + lo_f(q.w), hi_f(p.w) + hi_f(q.w)));
}

// Epilogue buffer (conv3x3_ws, producer-side bn backward sums): pixel slot q holds the stored values of its NT*32 channels as
// 16-byte chunks k = channel / 8, chunk index XOR-ed so that the eight lanes of a ds_write_b128 group hit distinct banks.
template <int NT> __device__ __forceinline__ int ebuf_swizzle(int q) { return NT == 1 ? (q >> 2) & 3 : (q >> 1) & 7; }

// NT 32-channel accumulator tiles of one pixel per lane -> NHWC bf16.  Lane = pixel x half; registers 4q..4q+3 of a tile
// hold channels 8q + 4*half + 0..3, so two v_permlane32_swap per 16 channels leave 8 consecutive channels (16 bytes) in
// every lane.  With "accumulate" the old values are fetched by one batch of unconditional loads (pix is clamped by the
// caller) before any add/store, instead of a load -> wait -> store chain per 16 bytes.
// Epilogue statistics kept per lane across all tiles of a persistent workgroup (reduced across lanes once, at kernel end):
//   mode 1  forward bn statistics of the stored values:  stat[..][0..7] += v,        stat[..][8..15] += v*v
//   mode 2  bn + relu backward sums for `out` = da:      stat[..][0..7] += dz*xhat,  stat[..][8..15] += dz
// bnc = this workgroup's [scale | shift | mean | invstd][C_WG] table (LDS), yraw = the layer's raw output at this pixel.
template <int NT>
__device__ __forceinline__ void store_pixel_tiles_rmw(const f32x16 (&acc)[NT], const ConvArgs& a, size_t pix, bool valid, int half, int co_base,
                                                      const u32x4 (&prefetched)[NT][2], bool use_prefetched,
                                                      float (*stat)[2][16] = nullptr, int stat_mode = 0,
                                                      const u32x4 (*yraw)[2] = nullptr, const float* bnc = nullptr,
                                                      char* ebuf = nullptr, int eq = 0, bool store_global = true) {
    const int C_OUT = a.c_out;  // a workgroup may own only NT*32 of the layer's output channels, starting at co_base
    uint4 q[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x16& v = acc[nt];
            unsigned a0 = pack2(v[8 * s + 0], v[8 * s + 1]), a1 = pack2(v[8 * s + 2], v[8 * s + 3]);
            unsigned b0 = pack2(v[8 * s + 4], v[8 * s + 5]), b1 = pack2(v[8 * s + 6], v[8 * s + 7]);
            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            q[nt][s] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    const size_t base = pix * C_OUT + co_base + 8 * half;
    // ---- first destination: its final stored value is what the statistics see ----
    bf16* out = reinterpret_cast<bf16*>(a.out);
    uint4 fin[NT][2];
    if (a.out_accumulate) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 old;
                if (use_prefetched) { const u32x4 v = prefetched[nt][s]; old = make_uint4(v[0], v[1], v[2], v[3]); }
                else old = *reinterpret_cast<const uint4*>(out + base + nt * 32 + 16 * s);
                fin[nt][s] = add_bf16x8(old, q[nt][s]);
            }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) fin[nt][s] = q[nt][s];
    }
    if (valid && store_global) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) store16(out + base + nt * 32 + 16 * s, fin[nt][s]);
    }
    if (ebuf) {   // the final values also go to the workgroup's epilogue buffer: the staging waves form the bn backward sums from them
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<uint4*>(ebuf + eq * (64 * NT) + (((nt * 4 + 2 * s + half) ^ ebuf_swizzle<NT>(eq)) << 4)) = fin[nt][s];
    }
    if (stat_mode == 1 && valid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned w[4] = {fin[nt][s].x, fin[nt][s].y, fin[nt][s].z, fin[nt][s].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = lo_f(w[i]), hi = hi_f(w[i]);
                    stat[nt][s][2 * i] += lo; stat[nt][s][8 + 2 * i] = fmaf(lo, lo, stat[nt][s][8 + 2 * i]);
                    stat[nt][s][2 * i + 1] += hi; stat[nt][s][8 + 2 * i + 1] = fmaf(hi, hi, stat[nt][s][8 + 2 * i + 1]);
                }
            }
    }
    if (stat_mode == 2 && valid) {
        const int cw = NT * 32;  // channels of this workgroup's table
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* t0 = bnc + nt * 32 + 16 * s + 8 * half;
                const unsigned wd[4] = {fin[nt][s].x, fin[nt][s].y, fin[nt][s].z, fin[nt][s].w};
                const u32x4 yv4 = yraw[nt][s];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int j = 2 * i + h2;
                        const float dv = h2 ? hi_f(wd[i]) : lo_f(wd[i]), yv = h2 ? hi_f(yv4[i]) : lo_f(yv4[i]);
                        // running sums of dz*y and dz: sum dz*xhat = invstd * (sum dz*y - mean * sum dz) is formed once per workgroup, in
                        // double, by the closing reduction (two VALU operations and two table reads fewer per value than carrying xhat)
                        const float dz = fmaf(yv, t0[j], t0[cw + j]) > 0.f ? dv : 0.f;
                        stat[nt][s][j] = fmaf(dz, yv, stat[nt][s][j]);
                        stat[nt][s][8 + j] += dz;
                    }
            }
    }
    // ---- optional second destination (skip gradient) ----
    bf16* out2 = reinterpret_cast<bf16*>(a.out2);
    if (out2) {
        if (a.out2_accumulate) {
            uint4 old[NT][2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < 2; ++s) old[nt][s] = *reinterpret_cast<const uint4*>(out2 + base + nt * 32 + 16 * s);
            if (valid) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s = 0; s < 2; ++s) *reinterpret_cast<uint4*>(out2 + base + nt * 32 + 16 * s) = add_bf16x8(old[nt][s], q[nt][s]);
            }
        } else if (valid) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s = 0; s < 2; ++s) *reinterpret_cast<uint4*>(out2 + base + nt * 32 + 16 * s) = q[nt][s];
        }
    }
}


// Inference epilogue: this layer's folded bn and relu on the fp32 accumulators, then bf16 and the same 16-byte NHWC stores.
// Register r of a tile holds channel 8 (r >> 2) + 4 half + (r & 3) BEFORE the permlane swap; act = [scale | shift][cw] of the
// workgroup's channels (LDS table or global memory).
template <int NT>
__device__ __forceinline__ void store_pixel_tiles_act(const f32x16 (&acc)[NT], const ConvArgs& a, size_t pix, bool valid, int half, int co_base, const float* act, int cw) {
    bf16* out = reinterpret_cast<bf16*>(a.out);
    const size_t base = pix * a.c_out + co_base + 8 * half;
    uint4 fin[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sc = *reinterpret_cast<const float4*>(act + nt * 32 + 8 * q + 4 * half);
            const float4 sh = *reinterpret_cast<const float4*>(act + cw + nt * 32 + 8 * q + 4 * half);
            v[4 * q + 0] = fmaf(acc[nt][4 * q + 0], sc.x, sh.x);
            v[4 * q + 1] = fmaf(acc[nt][4 * q + 1], sc.y, sh.y);
            v[4 * q + 2] = fmaf(acc[nt][4 * q + 2], sc.z, sh.z);
            v[4 * q + 3] = fmaf(acc[nt][4 * q + 3], sc.w, sh.w);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned a0 = relu_bf16x2(pack2(v[8 * s + 0], v[8 * s + 1])), a1 = relu_bf16x2(pack2(v[8 * s + 2], v[8 * s + 3]));
            const unsigned b0 = relu_bf16x2(pack2(v[8 * s + 4], v[8 * s + 5])), b1 = relu_bf16x2(pack2(v[8 * s + 6], v[8 * s + 7]));
            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            fin[nt][s] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    }
    if (valid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s) store16(out + base + nt * 32 + 16 * s, fin[nt][s]);
    }
}

// Inference epilogue of the layer under the 1x1 head: the activation of store_pixel_tiles_act (same rounding), not stored.  The head is
// two more MFMAs on the matrix core the conv just used (round 5; 64 multiply-adds, 16 unpacks and 12 cross-half operations per lane
// before: that epilogue was as long as the tile's MFMA phase).  BEFORE the permlane swap of store_pixel_tiles_act the lane (pixel `col`,
// half h) holds, as packed pairs, channels 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7 of step s: a valid B operand of a 32x32x16 MFMA
// whose reduction index is a PERMUTATION of the channels.  The A operand (head_operand below) holds the head weights in that same
// permutation, split in two bf16 terms so that an fp32 weight keeps 16 mantissa bits: row k (k < 4) = bf16(w[.][k]), row 4 + k =
// bf16(w - bf16(w)); every other row zero.  Rows 0..3 land in register k of the lower half, rows 4..7 in register k of the upper half:
// logit = (hi + lo) + bias after one permlane swap per class.  Half 0 stores classes 0 and 2, half 1 classes 1 and 3.
// (The separate head kernels sum in another order: tests compare the two forms within a tolerance, not bit for bit.)
__device__ __forceinline__ void head_operand(bf16x8 (&ha)[2], const ConvArgs& a, int col, int half) {
    const int k = col & 3, part = col >> 2;   // part 0: leading term, 1: remainder, >= 2: zero row
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
            const float w = (k < a.head_k && part < 2) ? a.head_w[ch * a.head_k + k] : 0.f;
            const bf16 hi = (bf16)w;
            ha[s][j] = part == 0 ? hi : (bf16)(w - (float)hi);
        }
}
__device__ __forceinline__ void store_pixel_tiles_head(const f32x16& acc, const ConvArgs& a, size_t pix, int n, bool valid, int half, const float* act, int cw,
                                                       const bf16x8 (&ha)[2], const float (&hbias)[4]) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 sc = *reinterpret_cast<const float4*>(act + 8 * q + 4 * half);
        const float4 sh = *reinterpret_cast<const float4*>(act + cw + 8 * q + 4 * half);
        v[4 * q + 0] = fmaf(acc[4 * q + 0], sc.x, sh.x);
        v[4 * q + 1] = fmaf(acc[4 * q + 1], sc.y, sh.y);
        v[4 * q + 2] = fmaf(acc[4 * q + 2], sc.z, sh.z);
        v[4 * q + 3] = fmaf(acc[4 * q + 3], sc.w, sh.w);
    }
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const u32x4 w4 = {relu_bf16x2(pack2(v[8 * s + 0], v[8 * s + 1])), relu_bf16x2(pack2(v[8 * s + 2], v[8 * s + 3])),
                          relu_bf16x2(pack2(v[8 * s + 4], v[8 * s + 5])), relu_bf16x2(pack2(v[8 * s + 6], v[8 * s + 7]))};
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[s], __builtin_bit_cast(bf16x8, w4), d, 0, 0, 0);
    }
    const size_t plane = (size_t)a.h_out * a.w_out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[k]), __float_as_uint(d[k]), false, false);
        const float other = __uint_as_float(half ? x[0] : x[1]);
        const float z = (half ? other + d[k] : d[k] + other) + hbias[k];   // (leading + remainder) + bias in both halves
        if (valid && k < a.head_k && (k & 1) == half) a.head_out[pix + ((size_t)n * (a.head_k - 1) + k) * plane] = z;
    }
}

template <int NT>
__device__ __forceinline__ void store_pixel_tiles(const f32x16 (&acc)[NT], const ConvArgs& a, size_t pix, bool valid, int half, int co_base = 0) {
    u32x4 none[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s) none[nt][s] = u32x4{0u, 0u, 0u, 0u};
    store_pixel_tiles_rmw<NT>(acc, a, pix, valid, half, co_base, none, false);
}

// ---- staging helpers shared by every MFMA kernel --------------------------------------------------------------------
// A "side" is a tensor read through its producer's bn+relu (+ skip add).  Staging a batch of 16-byte chunks is done in
// two phases: (1) EVERY global load of the batch is issued, unconditionally, from a clamped (always valid) address —
// a predicated load makes hipcc branch around it and wait vmcnt(0) per element, which serialises the memory round
// trips; (2) the prologue runs on the values, out-of-range chunks are zeroed, and the chunks go to LDS as 64-byte
// records whose 16-byte chunk index is XOR-ed with (record >> 2) & 3.

struct WgSide {
    const bf16* a; const float* a_scale; const float* a_shift;
    const bf16* b; const float* b_scale; const float* b_shift;
    int h, w, c;
};

template <int KIND> struct RawChunk { uint4 a, b; };

template <int KIND>
__device__ __forceinline__ RawChunk<KIND> side_load(const WgSide& s, size_t pix, int ch) {
    RawChunk<KIND> r;
    const size_t off = pix * s.c + ch;
    r.a = *reinterpret_cast<const uint4*>(s.a + off);
    if (KIND == SRC_ACT2) r.b = *reinterpret_cast<const uint4*>(s.b + off);
    return r;
}

template <int KIND>
__device__ __forceinline__ uint4 side_convert(const WgSide& s, const RawChunk<KIND>& r, int ch) {
    if (KIND == SRC_RAW) return r.a;
    if (KIND == SRC_ACT) return affine_relu_pack8(r.a, s.a_scale + ch, s.a_shift + ch);
    float v[8];
    affine8(r.a, s.a_scale + ch, s.a_shift + ch, v);
    if (KIND == SRC_ACT2) {
        float u[8];
        affine8(r.b, s.b_scale + ch, s.b_shift + ch, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += u[j];
    }
    return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// Stages ITEMS chunks (= ITEMS/4 records) with 256 threads, at most BATCH chunks per thread in flight.
// geo(rec, pix, ch0) -> bool: global pixel index (clamped into the tensor) and first channel of record `rec`; returns
// whether the record lies inside the tensor (otherwise it is zero-filled).
template <int KIND, int ITEMS, int BATCH, typename Geo>
__device__ __forceinline__ void stage_side(char* lds, const WgSide& s, int tid, const Geo& geo) {
    constexpr int N = (ITEMS + 255) / 256;
    const int c16 = tid & 3;
#pragma unroll
    for (int b0 = 0; b0 < N; b0 += BATCH) {
        RawChunk<KIND> raw[BATCH];
        int ch[BATCH];
        unsigned okmask = 0;
#pragma unroll
        for (int jj = 0; jj < BATCH; ++jj) {
            if (b0 + jj < N) {
                const int item = min(tid + 256 * (b0 + jj), ITEMS - 1);
                size_t pix; int ch0;
                const bool ok = geo(item >> 2, pix, ch0);
                ch[jj] = ch0 + c16 * 8;
                raw[jj] = side_load<KIND>(s, pix, ch[jj]);
                okmask |= (ok ? 1u : 0u) << jj;
            }
        }
#pragma unroll
        for (int jj = 0; jj < BATCH; ++jj) {
            if (b0 + jj < N) {
                const int item = tid + 256 * (b0 + jj);
                const int rec = item >> 2;
                uint4 v = side_convert<KIND>(s, raw[jj], ch[jj]);
                if (!((okmask >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
                if (item < ITEMS) *reinterpret_cast<uint4*>(lds + rec * 64 + ((c16 ^ ((rec >> 2) & 3)) << 4)) = v;
            }
        }
    }
}

// The same staging split in two calls, for software-pipelined kernels: side_fetch issues EVERY global load of the side
// into registers (nothing waits on them), side_commit — called one loop iteration later — runs the prologue and writes
// the records to LDS; conv(raw, j) is the prologue of the thread's j-th chunk.
template <int KIND, int ITEMS> struct SideRegs {
    static constexpr int N = (ITEMS + 255) / 256;
    RawChunk<KIND> raw[N];
    unsigned okmask;
};
template <int KIND, int ITEMS, typename Geo>
__device__ __forceinline__ void side_fetch(SideRegs<KIND, ITEMS>& R, const WgSide& s, int tid, const Geo& geo) {
    const int c16 = tid & 3;
    R.okmask = 0;
#pragma unroll
    for (int jj = 0; jj < SideRegs<KIND, ITEMS>::N; ++jj) {
        const int item = min(tid + 256 * jj, ITEMS - 1);
        size_t pix; int ch0;
        const bool ok = geo(item >> 2, pix, ch0);
        R.raw[jj] = side_load<KIND>(s, pix, ch0 + c16 * 8);
        R.okmask |= (ok ? 1u : 0u) << jj;
    }
}
template <int KIND, int ITEMS, typename Conv>
__device__ __forceinline__ void side_commit(const SideRegs<KIND, ITEMS>& R, char* lds, int tid, const Conv& conv) {
    const int c16 = tid & 3;
#pragma unroll
    for (int jj = 0; jj < SideRegs<KIND, ITEMS>::N; ++jj) {
        const int item = tid + 256 * jj;
        const int rec = item >> 2;
        uint4 v = conv(R.raw[jj], jj);  // jj is a compile-time constant after unrolling
        if (!((R.okmask >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
        if (item < ITEMS) *reinterpret_cast<uint4*>(lds + rec * 64 + ((c16 ^ ((rec >> 2) & 3)) << 4)) = v;
    }
}
// prologue of one 16-byte chunk with the per-channel constants given as pointers (registers or an LDS table)
template <int KIND>
__device__ __forceinline__ uint4 chunk_convert(const RawChunk<KIND>& r, const float* sa, const float* ta, const float* sb, const float* tb) {
    if (KIND == SRC_RAW) return r.a;
    if (KIND == SRC_SUM2) return add_bf16x8(r.a, r.b);
    if (KIND == SRC_ACT) return affine_relu_pack8(r.a, sa, ta);
    float v[8];
    affine8(r.a, sa, ta, v);
    if (KIND == SRC_ACT2) {
        float u[8];
        affine8(r.b, sb, tb, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += u[j];
    }
    return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// bn + relu backward of one 16-byte chunk (the stem's filter gradient applies it while staging): a = da, b = y; the expression and its order are those of bn_bwd_apply_vec_kernel
__device__ __forceinline__ uint4 chunk_bnbwd(const uint4& da, const uint4& y, const float* sc, const float* sf, const float* m, const float* is,
                                             const float* k0, const float* k1, const float* k2) {
    const unsigned wd[4] = {da.x, da.y, da.z, da.w}, wy[4] = {y.x, y.y, y.z, y.w};
    float r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * i + h;
            const float yv = h ? hi_f(wy[i]) : lo_f(wy[i]), dv = h ? hi_f(wd[i]) : lo_f(wd[i]);
            const float dz = fmaf(yv, sc[j], sf[j]) > 0.f ? dv : 0.f;
            const float xhat = (yv - m[j]) * is[j];
            r[j] = k0[j] * (dz - k1[j] - xhat * k2[j]);
        }
    }
    return make_uint4(pack2(r[0], r[1]), pack2(r[2], r[3]), pack2(r[4], r[5]), pack2(r[6], r[7]));
}

// weights of NTAPS taps for one 32-channel slab -> LDS records [tap][co][32 ci]; loads first, then writes
template <int C_OUT, int NTAPS>
__device__ __forceinline__ void stage_weights(char* lds_w, const bf16* wsrc, int t0, int c_red, int cc, int tid, int c_out_total = C_OUT, int co_base = 0) {
    constexpr int ITEMS = NTAPS * C_OUT * 4, WI = (ITEMS + 255) / 256;
    const int c16 = tid & 3;
    uint4 r[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int item = min(tid + 256 * j, ITEMS - 1);  // clamped: the load stays unconditional
        const int rec = item >> 2;
        const int tl = rec / C_OUT, co = rec - tl * C_OUT;
        r[j] = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(t0 + tl) * c_out_total + co_base + co) * c_red + cc + c16 * 8);
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int item = tid + 256 * j;
        const int rec = item >> 2;
        const int co = rec % C_OUT;
        if (item < ITEMS) *reinterpret_cast<uint4*>(lds_w + rec * 64 + ((c16 ^ ((co >> 2) & 3)) << 4)) = r[j];
    }
}

__host__ __device__ inline WgSide side_of_src(const Src& src, int h, int w, int c) {
    return WgSide{reinterpret_cast<const bf16*>(src.a), src.a_scale, src.a_shift, reinterpret_cast<const bf16*>(src.b), src.b_scale, src.b_shift, h, w, c};
}

template <int NT, int KIND, int TAPS>
__global__ __launch_bounds__(256, (NT == 4 ? 1 : 2)) void conv3x3s1_mfma_kernel(ConvArgs a, int tiles_x, int tiles_y, int flip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_x = smem;
    char* lds_w = smem + X_BYTES;
    constexpr int C_OUT = NT * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int tile = blockIdx.x;
    const int co_base = blockIdx.y * C_OUT;  // 128-channel layers are split over two workgroups of 64 channels (occupancy)
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const int H = a.h_out, W = a.w_out;  // stride 1, pad 1: input and output planes have the same size
    const int c_red = a.c_red;
    const bf16* wsrc = reinterpret_cast<const bf16*>(a.w_bf16);

    f32x16 acc[2][NT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][nt][r] = 0.f;

    const WgSide side = side_of_src(a.src, H, W, c_red);

    for (int cc = 0; cc < c_red; cc += 32) {
        __syncthreads();  // every wave is done reading the previous slab's patch and weights
        // ---- stage the input patch of this 32-channel slab ----
        stage_side<KIND, PATCH_PIX * 4, 6>(lds_x, side, tid, [&](int rec, size_t& pix, int& ch0) {
            const int py = rec / PW, pxx = rec - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
            pix = ((size_t)n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1);
            ch0 = cc;
            return iy >= 0 && iy < H && ix >= 0 && ix < W;
        });
        for (int t0 = 0; t0 < 9; t0 += TAPS) {
            if (t0 > 0) __syncthreads();  // the previous tap group's weight reads are done
            stage_weights<C_OUT, TAPS>(lds_w, wsrc, t0, c_red, cc, tid, a.c_out, co_base);
            __syncthreads();
            // ---- MFMA over the staged taps (2 waves per SIMD: the partner wave covers the LDS read latency) ----
#pragma unroll
            for (int tl = 0; tl < TAPS; ++tl) {
                const int tap = t0 + tl;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int pky = flip ? 2 - ky : ky, pkx = flip ? 2 - kx : kx;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int chunk = (ks << 1) | half;
                    bf16x8 xf[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int pidx = (wave * 2 + g + pky) * PW + col + pkx;
                        xf[g] = *reinterpret_cast<const bf16x8*>(lds_x + pidx * 64 + ((chunk ^ ((pidx >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = nt * 32 + col;
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds_w + (tl * C_OUT + co) * 64 + ((chunk ^ ((co >> 2) & 3)) << 4));
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[g], acc[g][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int oy = y0 + wave * 2 + g, ox = x0 + col;
        const bool valid = oy < H && ox < W;
        const size_t pix = ((size_t)n * H + (valid ? oy : 0)) * W + (valid ? ox : 0);
        store_pixel_tiles<NT>(acc[g], a, pix, valid, half, co_base);
    }
}

template <int NT, int TAPS>
void launch_s1(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.w_out + TW - 1) / TW, tiles_y = (a.h_out + TH - 1) / TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y * a.n), (unsigned)(a.c_out / (NT * 32))), block(256);
    const size_t lds = X_BYTES + (size_t)TAPS * NT * 32 * 64;
    const int flip = a.gather;  // backward-data of a stride-1 conv = the same conv with the taps mirrored
    switch (a.src.kind) {
        case SRC_RAW: hipLaunchKernelGGL((conv3x3s1_mfma_kernel<NT, SRC_RAW, TAPS>), grid, block, lds, s, a, tiles_x, tiles_y, flip); break;
        case SRC_ACT: hipLaunchKernelGGL((conv3x3s1_mfma_kernel<NT, SRC_ACT, TAPS>), grid, block, lds, s, a, tiles_x, tiles_y, flip); break;
        default: hipLaunchKernelGGL((conv3x3s1_mfma_kernel<NT, SRC_ACT2, TAPS>), grid, block, lds, s, a, tiles_x, tiles_y, flip); break;
    }
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad3x3_mfma: filter gradient of the 3x3 layers.
//   con, stride 1 pad 1 :  dw[t][ci][co] = sum_p x[p + t - 1][ci] * dy[p][co]
//   con, stride 2 pad 0 :  dw[t][ci][co] = sum_o x[2o + t][ci]   * dy[o][co]
//   cont, stride 2 pad 0:  dw[t][ci][co] = sum_i x[i][ci]        * dy[2i + t][co]
// All three are "sum over the pixels of a low-res TILE tensor of  PATCH[stride*p + t + origin] (x) TILE[p]".
// The reduction runs over PIXELS, which NHWC keeps strided — so both MFMA operands are fetched with the transposing
// LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major): tensors are staged
// as in the forward kernel (64-byte pixel records, XOR-swizzled chunks, producer bn+relu applied on the way in) and are
// never transposed in memory.  For stride 2 the patch is stored split by column parity, so a tap still reads 16
// CONSECUTIVE records.  One workgroup = (32-channel patch slab) x (all tile channels) x a strided set of pixel tiles;
// its 9*NTC 32x32 output tiles are dealt round-robin to the 4 waves and stay in accumulators across all of its pixel
// tiles; it writes ONE partial, and a fixed-order pass sums the partials (deterministic, no float atomics).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgParams {
    WgSide patch, tile;
    int n, c_in, c_out, transpose_out;  // transpose_out: patch channels are the OUTPUT channels (cont)
    float* partials;
};

template <int STRIDE> struct WgGeom;
template <> struct WgGeom<1> { static constexpr int TH = 8, TW = 32, RECS = 10 * 34, ORIGIN = -1; };
template <> struct WgGeom<2> { static constexpr int TH = 4, TW = 32, RECS = 9 * 66, ORIGIN = 0; };

__device__ __forceinline__ bf16x8 tr_read8(const char* base, int rec_first, int lane) {
    // this lane's share of a 16(k = pixel) x 32(channel) operand: records rec_first + 8*half + {0..7}, channel = lane & 31
    // (records swizzled by their INDEX: 32-pixel-wide tiles, where index and x-coordinate agree mod 16)
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, cg = (lane >> 4) & 1, half = lane >> 5;
    s16x4 r[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int rec = rec_first + 8 * half + 4 * rr + q;
        const char* addr = base + rec * 64 + ((((cg << 1) | (p >> 1)) ^ ((rec >> 2) & 3)) << 4) + ((p & 1) << 3);
        r[rr] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(addr));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3]};
    return __builtin_bit_cast(bf16x8, v);
}

// The same operand with the address arithmetic taken out of the inner loop.  The wgrad kernel swizzles a record by its
// X-COORDINATE ((x >> 2) & 3, x = column in the patch row / parity plane / tile row) instead of its index, so the
// swizzle of "16 consecutive pixels starting at column x_first" depends on the lane and on x_first mod 16 only — not
// on the row.  tr_lane_offset is that lane-constant byte offset (read rr covers record offsets 8*half + 4*rr + q);
// a k-step or tap row then only adds a compile-time constant, which the LDS instruction carries as its immediate.
__device__ __forceinline__ int tr_lane_offset(int lane, int rr, int x_first) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, cg = (lane >> 4) & 1, half = lane >> 5;
    const int L = 8 * half + 4 * rr + q;
    const int key = ((x_first + L) >> 2) & 3;
    return L * 64 + ((((cg << 1) | (p >> 1)) ^ key) << 4) + ((p & 1) << 3);
}
__device__ __forceinline__ bf16x8 tr_read8_at(const char* a0, const char* a1) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// k-steps KS0, KS0+1 of the wgrad MFMA phase (template recursion = fully unrolled with compile-time LDS offsets):
// fragments of step ks+1 are requested before the MFMAs of step ks issue.
// SWAP: accumulate the TRANSPOSED output tile (A = tile fragment, B = patch fragment), so that the accumulator's lane
// (column) index is the patch channel — the contiguous index of dW when patch channels are the output channels (cont).
template <int KS0, int KS, int TPW, int STRIDE, bool SWAP = false>
__device__ __forceinline__ void wg_ksteps(const char* ta0, const char* ta1, const char* (&pa0)[TPW], const char* (&pa1)[TPW],
                                          bf16x8& tf0, bf16x8 (&pf0)[TPW], bf16x8& tf1, bf16x8 (&pf1)[TPW], f32x16 (&acc)[TPW]) {
    if constexpr (KS0 < KS) {
        {
            constexpr int row = (KS0 + 1) >> 1, xh = ((KS0 + 1) & 1) << 4;
            constexpr int timm = (row * 32 + xh) * 64, pimm = (STRIDE == 1 ? row * 34 + xh : row * 132 + xh) * 64;
            tf1 = tr_read8_at(ta0 + timm, ta1 + timm);
#pragma unroll
            for (int i = 0; i < TPW; ++i) pf1[i] = tr_read8_at(pa0[i] + pimm, pa1[i] + pimm);
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf0, pf0[i], acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf0[i], tf0, acc[i], 0, 0, 0);
        if constexpr (KS0 + 2 < KS) {
            constexpr int row = (KS0 + 2) >> 1, xh = ((KS0 + 2) & 1) << 4;
            constexpr int timm = (row * 32 + xh) * 64, pimm = (STRIDE == 1 ? row * 34 + xh : row * 132 + xh) * 64;
            tf0 = tr_read8_at(ta0 + timm, ta1 + timm);
#pragma unroll
            for (int i = 0; i < TPW; ++i) pf0[i] = tr_read8_at(pa0[i] + pimm, pa1[i] + pimm);
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf1, pf1[i], acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf1[i], tf1, acc[i], 0, 0, 0);
        wg_ksteps<KS0 + 2, KS, TPW, STRIDE, SWAP>(ta0, ta1, pa0, pa1, tf0, pf0, tf1, pf1, acc);
    }
}

template <int KIND>
__device__ __forceinline__ RawChunk<KIND> side_load_at(const bf16* a, const bf16* b, int off) {
    RawChunk<KIND> r;
    r.a = load16(a + off);
    if (KIND == SRC_ACT2 || KIND == SRC_SUM2) r.b = load16(b + off);
    return r;
}

template <int NTC, int KP, int KT, int STRIDE>
__global__ __launch_bounds__(256, 1) void wgrad3x3_mfma_kernel(WgParams a, int tiles_x, int tiles_y, int total_tiles, int splits) {
    using G = WgGeom<STRIDE>;
    constexpr int TH = G::TH, TW = G::TW, TILE_PIX = TH * TW, KS = TILE_PIX / 16;
    constexpr int P_ITEMS = G::RECS * 4, NP = (P_ITEMS + 255) / 256;   // patch chunks, per thread
    constexpr int JPN = TILE_PIX / 64, NT_ = NTC * JPN;                // tile chunks per thread: JPN per 32-channel group
    constexpr int BUF_BYTES = G::RECS * 64 + TILE_PIX * NTC * 64;
    constexpr int TAB_BYTES = NTC * 32 * 4 * 4;            // the tile side's bn constants: [a_scale | a_shift | b_scale | b_shift][NTC*32]
    constexpr bool DB = 2 * BUF_BYTES + TAB_BYTES <= 160 * 1024;  // two LDS buffers when they fit: one barrier per pixel tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TPW = (9 * NTC + 3) / 4;  // output tiles per wave

    const int tid = threadIdx.x, lane = tid & 63, c16 = tid & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile assignment below stays scalar
    const int slab = blockIdx.y, split = blockIdx.x;
    const int cc = slab * 32;
    const int nt_mine = wave % NTC;

    // ---- per-channel bn constants: the patch side's 8 channels of this thread live in registers, the tile side's in LDS ----
    float psa[8], pta[8], psb[8], ptb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        psa[j] = KP != SRC_RAW ? a.patch.a_scale[cc + c16 * 8 + j] : 0.f;
        pta[j] = KP != SRC_RAW ? a.patch.a_shift[cc + c16 * 8 + j] : 0.f;
        psb[j] = KP == SRC_ACT2 ? a.patch.b_scale[cc + c16 * 8 + j] : 0.f;
        ptb[j] = KP == SRC_ACT2 ? a.patch.b_shift[cc + c16 * 8 + j] : 0.f;
    }
    float* tab = reinterpret_cast<float*>(smem + (DB ? 2 : 1) * BUF_BYTES);
    if (KT != SRC_RAW) {
        for (int i = tid; i < NTC * 32; i += 256) {
            tab[i] = a.tile.a_scale[i];
            tab[NTC * 32 + i] = a.tile.a_shift[i];
            tab[2 * NTC * 32 + i] = KT == SRC_ACT2 ? a.tile.b_scale[i] : 0.f;
            tab[3 * NTC * 32 + i] = KT == SRC_ACT2 ? a.tile.b_shift[i] : 0.f;
        }
        __syncthreads();
    }

    // ---- staging geometry, fixed per thread for the whole kernel ----
    // patch chunk jj = record (tid >> 2) + 64 jj -> (row py, column px); packed as py | px << 8.  Its LDS slot is
    // swizzled by the column within the row (stride 1) or within the parity plane (stride 2).
    int pgeo[NP], pdst[NP];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
        const int rec = min((tid >> 2) + 64 * jj, G::RECS - 1);
        int py, px, key;
        if (STRIDE == 1) { py = rec / 34; px = rec - py * 34; key = (px >> 2) & 3; }
        else { py = rec / 66; const int rem = rec - py * 66; const int par = rem >= 33; const int u = rem - 33 * par; px = 2 * u + par; key = (u >> 2) & 3; }
        pgeo[jj] = py | (px << 8);
        pdst[jj] = rec * 64 + ((c16 ^ key) << 4);
    }
    // tile chunk jj = 32-channel group jj / JPN, row (tid >> 7) + 2 (jj % JPN), column (tid >> 2) & 31
    const int t_x = (tid >> 2) & 31, t_row0 = tid >> 7;
    const int tdst0 = G::RECS * 64 + (t_row0 * 32 + t_x) * 64 + ((c16 ^ ((t_x >> 2) & 3)) << 4);

    // ---- this wave's output tiles j = wave + 4i -> (tap, nt_mine); a wave short of tiles repeats its previous one
    //      (never written), which keeps the MFMA phase free of branches.  pofs / tofs: lane part of the operand addresses ----
    int pofs[TPW][2], tofs[2];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int j = wave + 4 * i, jc = j < 9 * NTC ? j : j - 4;
        const int tap = jc / NTC, ky = tap / 3, kx = tap - ky * 3;
        const int first = STRIDE == 1 ? ky * 34 + kx : (ky * 2 + (kx & 1)) * 33 + (kx >> 1);
        const int x_first = STRIDE == 1 ? kx : (kx >> 1);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) pofs[i][rr] = first * 64 + tr_lane_offset(lane, rr, x_first);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) tofs[rr] = G::RECS * 64 + nt_mine * (TILE_PIX * 64) + tr_lane_offset(lane, rr, 0);

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // Software pipeline over this workgroup's pixel tiles: the global loads of tile i+1 are in flight (in registers)
    // while the MFMAs of tile i run; the prologue + LDS write of tile i+1 follow.  Pixels outside a tensor are zeros.
    RawChunk<KP> praw[NP];
    RawChunk<KT> traw[NT_];
    unsigned pok = 0, tok = 0;
    const size_t p_plane = (size_t)a.patch.h * a.patch.w * a.patch.c, t_plane = (size_t)a.tile.h * a.tile.w * a.tile.c;  // < 2^31 elements (host check)
    auto fetch = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
        const bf16* pa = a.patch.a + (size_t)n * p_plane;
        const bf16* pb = KP == SRC_ACT2 ? a.patch.b + (size_t)n * p_plane : nullptr;
        const int yb = STRIDE * y0 + G::ORIGIN, xb = STRIDE * x0 + G::ORIGIN;
        pok = 0;
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const int iy = yb + (pgeo[jj] & 255), ix = xb + (pgeo[jj] >> 8);
            const int cy = min(max(iy, 0), a.patch.h - 1), cx = min(max(ix, 0), a.patch.w - 1);
            praw[jj] = side_load_at<KP>(pa, pb, (cy * a.patch.w + cx) * a.patch.c + cc + c16 * 8);
            pok |= ((iy == cy && ix == cx) ? 1u : 0u) << jj;
        }
        const bf16* ta = a.tile.a + (size_t)n * t_plane;
        const bf16* tb = KT == SRC_ACT2 ? a.tile.b + (size_t)n * t_plane : nullptr;
        const int ox = x0 + t_x, cx = min(ox, a.tile.w - 1);
        tok = 0;
#pragma unroll
        for (int jj = 0; jj < NT_; ++jj) {
            const int oy = y0 + t_row0 + 2 * (jj % JPN), cy = min(oy, a.tile.h - 1);
            traw[jj] = side_load_at<KT>(ta, tb, (cy * a.tile.w + cx) * a.tile.c + (jj / JPN) * 32 + c16 * 8);
            tok |= ((oy == cy && ox == cx) ? 1u : 0u) << jj;
        }
    };

    int tile = split;
    if (tile < total_tiles) fetch(tile);
    int buf = 0;
    for (; tile < total_tiles; tile += splits) {
        char* lbuf = smem + buf * BUF_BYTES;
        if (!DB) __syncthreads();  // single buffer: every wave is done reading the previous tile
        // ---- prologue + LDS write of the fetched tile ----
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            uint4 v = chunk_convert<KP>(praw[jj], psa, pta, psb, ptb);
            if (!((pok >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
            if ((tid >> 2) + 64 * jj < G::RECS) *reinterpret_cast<uint4*>(lbuf + pdst[jj]) = v;
        }
#pragma unroll
        for (int jj = 0; jj < NT_; ++jj) {
            const float* t0 = tab + (jj / JPN) * 32 + c16 * 8;
            uint4 v = chunk_convert<KT>(traw[jj], t0, t0 + NTC * 32, t0 + 2 * NTC * 32, t0 + 3 * NTC * 32);
            if (!((tok >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(lbuf + tdst0 + ((jj / JPN) * TILE_PIX + 2 * (jj % JPN) * 32) * 64) = v;
        }
        __syncthreads();
        if (tile + splits < total_tiles) fetch(tile + splits);

        // ---- MFMA phase: the fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks issue; every
        //      address is (lane offset computed once) + (compile-time constant) ----
        const char* pa0[TPW];
        const char* pa1[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) { pa0[i] = lbuf + pofs[i][0]; pa1[i] = lbuf + pofs[i][1]; }
        const char* ta0 = lbuf + tofs[0];
        const char* ta1 = lbuf + tofs[1];
        bf16x8 tf0, tf1, pf0[TPW], pf1[TPW];
        tf0 = tr_read8_at(ta0, ta1);  // k-step 0
#pragma unroll
        for (int i = 0; i < TPW; ++i) pf0[i] = tr_read8_at(pa0[i], pa1[i]);
        wg_ksteps<0, KS, TPW, STRIDE>(ta0, ta1, pa0, pa1, tf0, pf0, tf1, pf1, acc);
        if (DB) buf ^= 1;
    }
    // ---- one partial per workgroup, laid out as dw: [tap][ci][co] ----
    const size_t nw = (size_t)9 * a.c_in * a.c_out;
    float* out = a.partials + (size_t)split * nw;
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int j = wave + 4 * i;
        if (j < 9 * NTC) {
            const int tap = j / NTC;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pch = cc + (r & 3) + 8 * (r >> 2) + 4 * half;  // patch channel (accumulator row)
                const int tch = nt_mine * 32 + col;                      // tile channel (accumulator column)
                const int ci = a.transpose_out ? tch : pch, co = a.transpose_out ? pch : tch;
                out[((size_t)tap * a.c_in + ci) * a.c_out + co] = acc[i][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad3x3_ws: the same filter-gradient computation, operand layout and output as wgrad3x3_mfma, warp-specialised like
// conv3x3_ws: 512 threads, waves 4..7 (producers) fetch pixel tile i+1 into registers, apply the prologues and write LDS
// buffer (i+1)&1 while waves 0..3 (consumers) run the transposing reads + MFMAs of tile i from buffer i&1; one barrier
// per tile.  128 tile channels (NTC = 4) at stride 1 use 4x32 pixel tiles so that two buffers fit in LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int NTC, int KP, int KT, int STRIDE, bool TRANS>
__global__ __launch_bounds__(512, 2) void wgrad3x3_ws_kernel(WgParams a, int tiles_x, int tiles_y, int total_tiles_all, int splits_all) {
    constexpr int TH = STRIDE == 1 ? (NTC == 4 ? 4 : 8) : 4, TW = 32, TILE_PIX = TH * TW, KS = TILE_PIX / 16;
    constexpr int RECS = STRIDE == 1 ? (TH + 2) * 34 : 9 * 66, ORIGIN = STRIDE == 1 ? -1 : 0;
    constexpr int P_ITEMS = RECS * 4, NP = (P_ITEMS + 255) / 256;   // patch chunks per producer thread
    constexpr int JPN = TILE_PIX / 64, NT_ = NTC * JPN;            // tile chunks per producer thread: JPN per 32-channel group
    constexpr int BUF_BYTES = RECS * 64 + TILE_PIX * NTC * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TPW = (9 * NTC + 3) / 4;  // output tiles per consumer wave

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255, lane = tid & 63, c16 = tid & 3;
    const int slab = blockIdx.y, split = blockIdx.x;
    const int cc = slab * 32;
    const int tch0 = blockIdx.z * (NTC * 32);   // 256 tile channels run as two workgroup groups of 128 (levels = 3)

    float* tab = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);  // tile side's bn constants: [a_scale | a_shift | b_scale | b_shift][NTC*32]
    if (KT != SRC_RAW) {
        for (int i = threadIdx.x; i < NTC * 32; i += 512) {
            tab[i] = a.tile.a_scale[tch0 + i];
            tab[NTC * 32 + i] = a.tile.a_shift[tch0 + i];
            tab[2 * NTC * 32 + i] = KT == SRC_ACT2 ? a.tile.b_scale[tch0 + i] : 0.f;
            tab[3 * NTC * 32 + i] = KT == SRC_ACT2 ? a.tile.b_shift[tch0 + i] : 0.f;
        }
    }
    __syncthreads();

    // XCD bands (splits_all bit 30; ANH_WGRAD_XCD_BANDS): as conv3x3_ws — the workgroups of one XCD (split & 7: gridDim.x is a multiple of
    // 8, so every slab / channel group of a split sits on that XCD too) walk one contiguous eighth of the tile list together, so the halo
    // rows and columns neighbouring patches share are re-read from that XCD's L2.  Which tiles a split sums changes with it (fp32 partials
    // in another — still fixed — order).
    const int wbands = (splits_all >> 30) & 1, n_splits = splits_all & 0x3fffffff;
    const int band_len = (total_tiles_all + 7) >> 3;
    const int splits = wbands ? n_splits >> 3 : n_splits;                                                  // this split's step through the list
    const int total_tiles = wbands ? min(total_tiles_all, ((split & 7) + 1) * band_len) : total_tiles_all;   // ... and its end
    int tile = wbands ? (split & 7) * band_len + (split >> 3) : split, it = 0;
    if (producer) {
        // per-channel bn constants of the patch side: this thread's 8 channels, in registers
        float psa[8], pta[8], psb[8], ptb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            psa[j] = KP != SRC_RAW ? a.patch.a_scale[cc + c16 * 8 + j] : 0.f;
            pta[j] = KP != SRC_RAW ? a.patch.a_shift[cc + c16 * 8 + j] : 0.f;
            psb[j] = KP == SRC_ACT2 ? a.patch.b_scale[cc + c16 * 8 + j] : 0.f;
            ptb[j] = KP == SRC_ACT2 ? a.patch.b_shift[cc + c16 * 8 + j] : 0.f;
        }
        // staging geometry, fixed per thread: patch chunk jj = record (tid >> 2) + 64 jj -> (row, column), swizzled by column
        int pgeo[NP], pdst[NP];
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const int rec = min((tid >> 2) + 64 * jj, RECS - 1);
            int py, px, key;
            if (STRIDE == 1) { py = rec / 34; px = rec - py * 34; key = (px >> 2) & 3; }
            else { py = rec / 66; const int rem = rec - py * 66; const int par = rem >= 33; const int u = rem - 33 * par; px = 2 * u + par; key = (u >> 2) & 3; }
            pgeo[jj] = py | (px << 8);
            pdst[jj] = rec * 64 + ((c16 ^ key) << 4);
        }
        // tile chunk jj = 32-channel group jj / JPN, row (tid >> 7) + 2 (jj % JPN), column (tid >> 2) & 31
        const int t_x = (tid >> 2) & 31, t_row0 = tid >> 7;
        const int tdst0 = RECS * 64 + (t_row0 * 32 + t_x) * 64 + ((c16 ^ ((t_x >> 2) & 3)) << 4);
        // TWO register sets: the loads of tiles i+1 and i+2 are in flight while tile i is committed — one tile of MFMA
        // time does not cover the memory latency under load
        struct Regs { RawChunk<KP> p[NP]; RawChunk<KT> t[NT_]; unsigned pok, tok; };
        Regs ra, rb;
        const size_t p_plane = (size_t)a.patch.h * a.patch.w * a.patch.c, t_plane = (size_t)a.tile.h * a.tile.w * a.tile.c;  // < 2^31 elements (host check)
        // Buffer loads, as the conv kernels' producers (round 4: a SIMD's issue port is what these kernels spend): one descriptor per
        // image and operand, a 32-bit byte offset per chunk, positions outside the image at an offset outside the descriptor (zeros
        // by themselves: the plain-copy kinds need neither clamps nor masks).
        constexpr bool P_MASK = KP == SRC_ACT || KP == SRC_ACT2, T_MASK = KT == SRC_ACT || KT == SRC_ACT2;   // bn kinds: relu(shift) is not zero
        auto buffer_side = [&](const __amdgpu_buffer_rsrc_t& ra_, const __amdgpu_buffer_rsrc_t& rb_, int off, auto& dst, auto kind_tag) __attribute__((always_inline)) {
            constexpr int K = decltype(kind_tag)::value;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_, off, 0, 0);
            dst.a = make_uint4(v[0], v[1], v[2], v[3]);
            if constexpr (K == SRC_ACT2 || K == SRC_SUM2) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rb_, off, 0, 0);
                dst.b = make_uint4(w[0], w[1], w[2], w[3]);
            }
        };
        auto fetch = [&](Regs& R, int tile_) __attribute__((always_inline)) {
            const int tx = tile_ % tiles_x, ty = (tile_ / tiles_x) % tiles_y, n = tile_ / (tiles_x * tiles_y);
            const int x0 = tx * TW, y0 = ty * TH;
            const bf16* pa = a.patch.a + (size_t)n * p_plane;
            const bf16* pb = KP == SRC_ACT2 ? a.patch.b + (size_t)n * p_plane : nullptr;
            const int yb = STRIDE * y0 + ORIGIN, xb = STRIDE * x0 + ORIGIN;
            const bf16* ta = a.tile.a + (size_t)n * t_plane;
            const bf16* tb = KT == SRC_ACT2 ? a.tile.b + (size_t)n * t_plane : nullptr;
            R.pok = 0;
            R.tok = 0;
            {
                const __amdgpu_buffer_rsrc_t rpa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(pa), 0, (int)(unsigned)(p_plane * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t rpb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(KP == SRC_ACT2 ? pb : pa), 0, (int)(unsigned)(p_plane * 2), 0x00020000);
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) {
                    const int iy = yb + (pgeo[jj] & 255), ix = xb + (pgeo[jj] >> 8);
                    const bool ok = (unsigned)iy < (unsigned)a.patch.h && (unsigned)ix < (unsigned)a.patch.w;
                    buffer_side(rpa, rpb, ok ? ((iy * a.patch.w + ix) * a.patch.c + cc + c16 * 8) * 2 : (int)0xFFFFF000u, R.p[jj], std::integral_constant<int, KP>{});
                    if constexpr (P_MASK) R.pok |= (ok ? 1u : 0u) << jj;
                }
                const __amdgpu_buffer_rsrc_t rta = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(ta), 0, (int)(unsigned)(t_plane * 2), 0x00020000);
                const __amdgpu_buffer_rsrc_t rtb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(KT == SRC_ACT2 ? tb : ta), 0, (int)(unsigned)(t_plane * 2), 0x00020000);
                const int ox = x0 + t_x;
#pragma unroll
                for (int jj = 0; jj < NT_; ++jj) {
                    const int oy = y0 + t_row0 + 2 * (jj % JPN);
                    const bool ok = oy < a.tile.h && ox < a.tile.w;
                    buffer_side(rta, rtb, ok ? ((oy * a.tile.w + ox) * a.tile.c + tch0 + (jj / JPN) * 32 + c16 * 8) * 2 : (int)0xFFFFF000u, R.t[jj], std::integral_constant<int, KT>{});
                    if constexpr (T_MASK) R.tok |= (ok ? 1u : 0u) << jj;
                }
            }
        };
        auto commit = [&](const Regs& R, char* lbuf) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                uint4 v = chunk_convert<KP>(R.p[jj], psa, pta, psb, ptb);
                if constexpr (P_MASK) { if (!((R.pok >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u); }
                if ((tid >> 2) + 64 * jj < RECS) *reinterpret_cast<uint4*>(lbuf + pdst[jj]) = v;
            }
#pragma unroll
            for (int jj = 0; jj < NT_; ++jj) {
                const float* t0 = tab + (jj / JPN) * 32 + c16 * 8;
                uint4 v = chunk_convert<KT>(R.t[jj], t0, t0 + NTC * 32, t0 + 2 * NTC * 32, t0 + 3 * NTC * 32);
                if constexpr (T_MASK) { if (!((R.tok >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u); }
                *reinterpret_cast<uint4*>(lbuf + tdst0 + ((jj / JPN) * TILE_PIX + 2 * (jj % JPN) * 32) * 64) = v;
            }
        };
        // Skip-add patches (two inputs per chunk) and the 128-tile-channel forms keep ONE set.  With two, those forms hold 195-232
        // VGPRs, and what this kernel's waves leave of a SIMD's register file decides how many waves of the bn backward apply pass —
        // which runs beside this kernel on the other stream and IS on the step's critical path — fit next to them: none beside 232,
        // one 82-VGPR wave beside 195, two beside <= 168.  Same-box A/B: 1.7405 -> 1.7217 ms per step; this
        // kernel's own time does not change (it is not what the step waits for).  (Every form lean / only the skip-add forms lean measured
        // 1.725 / 1.726-1.733 against 1.721-1.723 for this choice: profiles/r03 notes.)
        constexpr bool ONE_SET = KP == SRC_ACT2 || NTC == 4;
        if (tile < total_tiles) fetch(ra, tile);
        if constexpr (ONE_SET) {
            while (tile < total_tiles) {
                commit(ra, smem + (it & 1) * BUF_BYTES);
                if (tile + splits < total_tiles) fetch(ra, tile + splits);
                __syncthreads();  // buffer it & 1 is full; the consumers are done with the other one
                tile += splits; ++it;
            }
        } else {
            if (tile + splits < total_tiles) fetch(rb, tile + splits);
            while (tile < total_tiles) {   // unrolled by two so that the register sets are addressed statically
                commit(ra, smem);
                if (tile + 2 * splits < total_tiles) fetch(ra, tile + 2 * splits);
                __syncthreads();  // buffer 0 is full; the consumers are done with buffer 1
                tile += splits;
                if (tile >= total_tiles) break;
                commit(rb, smem + BUF_BYTES);
                if (tile + 2 * splits < total_tiles) fetch(rb, tile + 2 * splits);
                __syncthreads();
                tile += splits;
            }
        }
    } else {
        const int nt_mine = wave % NTC;
        // this wave's output tiles j = wave + 4i -> (tap, nt_mine); a wave short of tiles repeats its previous one (never
        // written), which keeps the MFMA phase free of branches.  pofs / tofs: lane part of the operand addresses
        int pofs[TPW][2], tofs[2];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int j = wave + 4 * i, jc = j < 9 * NTC ? j : j - 4;
            const int tap = jc / NTC, ky = tap / 3, kx = tap - ky * 3;
            const int first = STRIDE == 1 ? ky * 34 + kx : (ky * 2 + (kx & 1)) * 33 + (kx >> 1);
            const int x_first = STRIDE == 1 ? kx : (kx >> 1);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) pofs[i][rr] = first * 64 + tr_lane_offset(lane, rr, x_first);
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) tofs[rr] = RECS * 64 + nt_mine * (TILE_PIX * 64) + tr_lane_offset(lane, rr, 0);
        f32x16 acc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (; tile < total_tiles; tile += splits, ++it) {
            const char* lbuf = smem + (it & 1) * BUF_BYTES;
            const char* pa0[TPW];
            const char* pa1[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) { pa0[i] = lbuf + pofs[i][0]; pa1[i] = lbuf + pofs[i][1]; }
            const char* ta0 = lbuf + tofs[0];
            const char* ta1 = lbuf + tofs[1];
            __syncthreads();  // buffer it & 1 is full
            bf16x8 tf0, tf1, pf0[TPW], pf1[TPW];
            tf0 = tr_read8_at(ta0, ta1);  // k-step 0
#pragma unroll
            for (int i = 0; i < TPW; ++i) pf0[i] = tr_read8_at(pa0[i], pa1[i]);
            // TRANS (cont: patch channels = output channels): accumulate transposed, so that the partial's stores run along co
            wg_ksteps<0, KS, TPW, STRIDE, TRANS>(ta0, ta1, pa0, pa1, tf0, pf0, tf1, pf1, acc);
        }
        // one partial per workgroup, laid out as dw: [tap][ci][co]
        const size_t nw = (size_t)9 * a.c_in * a.c_out;
        float* out = a.partials + (size_t)split * nw;
        const int col = lane & 31, half = lane >> 5;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int j = wave + 4 * i;
            if (j < 9 * NTC) {
                const int tap = j / NTC;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowc = (r & 3) + 8 * (r >> 2) + 4 * half;      // accumulator row -> channel within the 32-channel tile
                    const int pch = cc + (TRANS ? col : rowc);               // patch channel
                    const int tch = tch0 + nt_mine * 32 + (TRANS ? rowc : col);   // tile channel
                    const int ci = TRANS ? tch : pch, co = TRANS ? pch : tch;
                    out[((size_t)tap * a.c_in + ci) * a.c_out + co] = acc[i][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Geometry policies of the persistent 3x3 conv kernel (conv3x3_ws below): stride-1 con and its mirrored backward-data;
// stride-2 con / cont backward-data; cont / stride-2 con backward-data.
// LDS pixel records are swizzled by the patch COLUMN (column within the parity plane for stride 2), so every operand
// address in the MFMA phase is (lane offset computed once per kernel) + (compile-time constant carried by the ds_read's
// immediate): no address arithmetic between MFMAs.  Mirrored taps are handled by staging the filter taps in reverse
// order.  Single-slab layers stage their filter block once per workgroup.
//   GeoS1   8x32 output tile, 10x34 patch; wave = 2 output rows.
//   GeoDown 4x32 output tile, 9x65 patch stored split by column parity (a tap reads consecutive records); wave = 1 row.
//   GeoUp   4x32 tile of LOW-RES positions, 5x33 patch; the four output parity classes take 4 / 2 / 2 / 1 taps and
//           each owns an accumulator, so no MFMA multiplies by a structural zero; wave = 1 low-res row.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const char* swz_addr(const char* base, int rec, int key, int ks, int half) {
    return base + rec * 64 + ((((ks << 1) | half) ^ (key & 3)) << 4);
}
__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

struct GeoS1 {
    static constexpr int RECS = PATCH_PIX, ACC = 2;
    static constexpr bool RMW_PREFETCH = false;  // read-modify-write destinations prefetched before the MFMA phase (register cost)
    __device__ static void decode(int rec, int& py, int& px, int& key) { py = rec / PW; px = rec - py * PW; key = (px >> 2) & 3; }
    __device__ static int in_y0(int ty) { return ty * TH - 1; }
    __device__ static int in_x0(int tx) { return tx * TW - 1; }
    static constexpr int NB = 3;  // one lane base per kx
    struct Bases { const char* x[3][2]; };
    __device__ static void init(Bases& b, const char* lds_x, int wave, int col, int half) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b.x[kx][ks] = swz_addr(lds_x, wave * 2 * PW + col + kx, (col + kx) >> 2, ks, half);
    }
    // The two output rows of a wave and the three filter rows touch patch rows 0..3: for each (kx, k-step) those four pixel
    // fragments are read ONCE and reused by the three ky (10 LDS reads per 12 MFMAs at NT = 2, instead of 12).
    // (Round 3: a hand-pinned two-register-set form of this nest — all ten fragments of step i + 1 requested behind the second MFMA of
    // step i — shortens the instrumented MFMA phase by 17 % (27 -> 22.7 us on the 64->64 forward) and leaves the step where it was:
    // 1.741 vs 1.738 ms for this plain nest, same box.  What did pay is registers: see HAS_FWD_FORM in launch_ws.)
    template <int NT>
    __device__ static void mfma(f32x16 (&acc)[ACC][NT], const Bases& b, const char* (&wb)[2]) {
        constexpr int C_OUT = NT * 32;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xf[r] = lds_frag(b.x[kx][ks] + r * (PW * 64));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int tl = ky * 3 + kx;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 wf = lds_frag(wb[ks] + (tl * C_OUT + nt * 32) * 64);
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[g + ky], acc[g][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // The 32 -> 32 layer (one reduction slab, one channel tile): its 18 filter fragments (9 taps x 2 k-steps, 72 VGPRs) are the same for
    // every tile — read from LDS once per workgroup and kept in registers, the nest then reads the 24 pixel fragments of an item only
    // (42 LDS reads per 36 MFMAs before: the phase ran at half the matrix rate on LDS bandwidth).
    __device__ static void load_filter_regs(bf16x8 (&wreg)[18], const char* (&wb)[2]) {
#pragma unroll
        for (int tl = 0; tl < 9; ++tl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wreg[tl * 2 + ks] = lds_frag(wb[ks] + (tl * 32) * 64);
    }
    __device__ static void mfma_wreg(f32x16 (&acc)[ACC][1], const Bases& b, const bf16x8 (&wreg)[18]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xf[r] = lds_frag(b.x[kx][ks] + r * (PW * 64));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[(ky * 3 + kx) * 2 + ks], xf[g + ky], acc[g][0], 0, 0, 0);
            }
        }
    }
    // output pixel of accumulator group g (one lane = one pixel): row 2*wave + g of the tile
    __device__ static void out_pixel(int g, const ConvArgs& a, int n, int ty, int tx, int wave, int col, size_t& pix, bool& valid) {
        const int oy = ty * TH + wave * 2 + g, ox = tx * TW + col;
        valid = oy < a.h_out && ox < a.w_out;
        pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
    }
};

struct GeoDown {
    static constexpr int RECS = 9 * 66, ACC = 1;
    static constexpr bool RMW_PREFETCH = false;
    __device__ static void decode(int rec, int& py, int& px, int& key) {
        py = rec / 66;
        const int rem = rec - py * 66, par = rem >= 33, u = rem - 33 * par;
        px = 2 * u + par; key = (u >> 2) & 3;
    }
    __device__ static int in_y0(int ty) { return ty * 8; }
    __device__ static int in_x0(int tx) { return tx * 64; }
    static constexpr int NB = 2;
    struct Bases { const char* x[2][2]; };  // [kx >> 1][ks]
    __device__ static void init(Bases& b, const char* lds_x, int wave, int col, int half) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b.x[kh][ks] = swz_addr(lds_x, wave * 4 * 33 + col + kh, (col + kh) >> 2, ks, half);
    }
    template <int NT>
    __device__ static void mfma(f32x16 (&acc)[ACC][NT], const Bases& b, const char* (&wb)[2]) {
        constexpr int C_OUT = NT * 32;
#pragma unroll
        for (int tl = 0; tl < 9; ++tl) {
            const int ky = tl / 3, kx = tl - ky * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 xf = lds_frag(b.x[kx >> 1][ks] + (ky * 2 + (kx & 1)) * (33 * 64));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 wf = lds_frag(wb[ks] + (tl * C_OUT + nt * 32) * 64);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[0][nt], 0, 0, 0);
                }
            }
        }
    }
    __device__ static void out_pixel(int, const ConvArgs& a, int n, int ty, int tx, int wave, int col, size_t& pix, bool& valid) {
        const int oy = ty * 4 + wave, ox = tx * 32 + col;
        valid = oy < a.h_out && ox < a.w_out;
        pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
    }
};

struct GeoUp {
    static constexpr int RECS = 5 * 33, ACC = 4;
    static constexpr bool RMW_PREFETCH = true;   // the skip-gradient accumulation of this net lands in stride-2 con backward-data
    __device__ static void decode(int rec, int& py, int& px, int& key) { py = rec / 33; px = rec - py * 33; key = (px >> 2) & 3; }
    __device__ static int in_y0(int ty) { return ty * 4 - 1; }
    __device__ static int in_x0(int tx) { return tx * 32 - 1; }
    static constexpr int NB = 2;
    struct Bases { const char* x[2][2]; };  // [ib][ks]: input column j - ib, row i - 1 (the immediate adds a row for ia = 0)
    __device__ static void init(Bases& b, const char* lds_x, int wave, int col, int half) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b.x[ib][ks] = swz_addr(lds_x, wave * 33 + col + 1 - ib, (col + 1 - ib) >> 2, ks, half);
    }
    template <int NT>
    __device__ static void mfma(f32x16 (&acc)[ACC][NT], const Bases& b, const char* (&wb)[2]) {
        constexpr int C_OUT = NT * 32;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[2][2];  // [ia][ib]: input (i - ia, j - ib)
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) xf[ia][ib] = lds_frag(b.x[ib][ks] + (1 - ia) * (33 * 64));
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int ia = 0; ia < 2 - py; ++ia)
#pragma unroll
                        for (int ib = 0; ib < 2 - px; ++ib) {
                            const int tap = (py + 2 * ia) * 3 + px + 2 * ib;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const bf16x8 wf = lds_frag(wb[ks] + (tap * C_OUT + nt * 32) * 64);
                                acc[py * 2 + px][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[ia][ib], acc[py * 2 + px][nt], 0, 0, 0);
                            }
                        }
        }
    }
    // accumulator group g = output parity class (py, px) = (g >> 1, g & 1) of low-res position (i, j)
    __device__ static void out_pixel(int g, const ConvArgs& a, int n, int ty, int tx, int wave, int col, size_t& pix, bool& valid) {
        const int oy = 2 * (ty * 4 + wave) + (g >> 1), ox = 2 * (tx * 32 + col) + (g & 1);
        valid = oy < a.h_out && ox < a.w_out;
        pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// conv3x3_ws: the 3x3 layers as ONE persistent, warp-specialised kernel.  A workgroup walks (pixel tile, 32-channel slab)
// items with the two halves of the work on different waves of a 512-thread workgroup (1 workgroup per CU, 2 waves per SIMD):
//   waves 4..7 (producers): global loads of item i+1 into registers, prologue (bn + relu, skip add), LDS write into
//                           buffer (i+1) & 1;
//   waves 0..3 (consumers): MFMAs of item i from buffer i & 1, epilogue stores.
// One workgroup barrier per item hands a filled buffer over and releases the other one, so the staging VALU work, the
// global-load latency and the MFMA/LDS-read work of a CU overlap instead of adding up.
// ---------------------------------------------------------------------------------------------------------------
// FWD: the forward-only form — no prefetched epilogue operands (old values of an accumulating destination, y of the
// layer behind `out`), whose registers the four-accumulator-group geometry at NT = 2 needs for its bn statistics sums.
// ACT (inference, with FWD): the epilogue applies this layer's own folded bn + relu (ConvArgs::out_scale) and stores the activation.
// HEAD (with ACT; GeoS1, NT = 1): the epilogue forms the 1x1 head's logits instead of storing the activation.
// PS (backward-data forms with the bn backward sums fused; plain-copy staging): what the PRODUCER waves take over from the consumers'
// epilogue.  The consumers leave the final values of a tile in an LDS epilogue buffer (e_off); two items later a producer thread —
// which owns a fixed 16-byte channel chunk — reads them back, consecutive lanes holding consecutive chunks:
//   PS = 1  the producers form the bn backward sums (16 running sums per thread instead of the consumers' NT x 32; y fetched meanwhile);
//   PS = 4  the producers issue the tile's global stores, the consumers keep the sums (stride-1 / down geometries);
//   PS = 5  both (the four-accumulator-group geometry at 32 channels).
// In the backward-data convs the producers wait at the hand-over barrier for a third to half of the kernel while the consumers'
// epilogue — mostly stores waiting to be issued — is as long as their MFMA phase: this moves that work to where the slack is.
// Forms measured and removed again (rounds 2-4; profiles/r04_ab_log.txt, DESIGN.md "Measured and rejected"): sums split between the roles,
// y operands through LDS, LDS-DMA staging, two consumer teams, one barrier per two items, deferred / transposed / non-temporal stores,
// a second staging register set, a pinned MFMA nest, 16 x 32 tiles, role-per-SIMD maps, wave priorities, staggered starts.
template <class G, int NT, int KIND, bool FWD = false, bool ACT = false, int PS = 0, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(ConvArgs a, int tiles_x, int tiles_y, int flip, long long* prof, int wres_, int e_off) {
    // per-phase wall-clock accounting, compiled in with -DANH_WS_PROFILE (ANH_WS_PROF=1 then prints one line per launch)
#ifdef ANH_WS_PROFILE
    long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, t0_;
    float prof_sink = 0.f;
    const bool prof_nostore = (wres_ >> 6) & 1;   // ANH_WS_PROF_NOSTORE=1: the epilogue computes but stores nothing (timing only, results wrong)
    const long long t_entry = wall_clock64();   // absolute (the counter is chip-wide): launch skew, prologue and tail of the kernel
    long long t_loop = 0, t_loop_end = 0, c_loop = 0;   // c_loop: shader cycles (s_memtime) over the producers' loop -> the clock the kernel holds
#define TICK() (t0_ = wall_clock64())
#define TOCK(acc_) (acc_ += wall_clock64() - t0_)
#else
#define TICK()
#define TOCK(acc_)
#endif
    static_assert(PS == 0 || PS == 1 || PS == 4 || PS == 5, "conv3x3_ws: producer-side forms 1 (sums), 4 (stores), 5 (both)");
    constexpr int NTHR = 512;
    constexpr int C_OUT = NT * 32, NP = (G::RECS * 4 + 255) / 256, W_ITEMS = 9 * C_OUT * 4, NW = (W_ITEMS + 255) / 256;
    constexpr int X_BYTES_ = G::RECS * 64, W_BYTES = 9 * C_OUT * 64, BUF = X_BYTES_ + W_BYTES;
    const int wres = wres_ & 1, bands = (wres_ >> 4) & 1;   // wres_ bit 0: filter-resident form; bit 4: XCD-band tile walk; bit 19: filter fragments NOT kept in registers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS layout.  Streaming form: [X | W] [X | W] tables — the filter slab of every item travels with its patch.
    // Resident form (wres; the filter slabs of ALL reduction slabs fit beside two patches, i.e. 64 reduction channels):
    // [X] [X] [W slab 0] [W slab 1] ... tables — the producers stage the filter once and then move patches only.
    const int n_slabs_l = a.c_red >> 5;
    const int x_stride = wres ? X_BYTES_ : BUF;                     // between the two patch buffers
    const int w_base = 2 * X_BYTES_;                               // resident filter slabs
    const int tab_off = wres ? w_base + n_slabs_l * W_BYTES : 2 * BUF;
    float* tab = reinterpret_cast<float*>(smem + tab_off);  // [a_scale | a_shift | b_scale | b_shift][c_red]

    // The hardware deals the eight waves of a workgroup round the four SIMDs (wave w -> SIMD w & 3): one producer and one consumer
    // wave on every SIMD, all four matrix cores in use.
    const int hw_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool producer = hw_wave >= 4;
    const int wave = hw_wave & 3;   // index within the role (0..3)
    const int lane = threadIdx.x & 63, tid = wave * 64 + lane, c16 = tid & 3;
    const int half = lane >> 5, col = lane & 31;
    const int co_base = blockIdx.y * C_OUT;
    const int H = a.h_in, W = a.w_in, c_red = a.c_red;
    const int n_slabs = c_red >> 5, n_tiles_all = tiles_x * tiles_y * a.n;
    // Which tiles this workgroup walks.  Plain: tile blockIdx.x, then every gridDim.x-th.  XCD bands (`bands`; gridDim.x a multiple of 8):
    // the hardware deals workgroups round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md: blocks b and b + 8 share one — used for
    // speed only, any placement gives the same result), so the workgroups with equal blockIdx.x & 7 take ONE contiguous eighth of the
    // row-major tile list and step through it together, gridDim.x / 8 consecutive tiles at a time: the halo rows and columns a patch
    // shares with its neighbours are then re-read from that XCD's own L2 instead of once per XCD from the memory side.
    const int band_len = (n_tiles_all + 7) >> 3;
    const int tile_first = bands ? (int)(blockIdx.x & 7) * band_len + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int gstep = bands ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int n_tiles = bands ? min(n_tiles_all, ((int)(blockIdx.x & 7) + 1) * band_len) : n_tiles_all;   // end of this workgroup's range
    constexpr bool CAN_STATS = G::ACC * NT <= 4 || FWD;  // register budget of the consumer waves
    const bool fuse_stats = !PS && !ACT && CAN_STATS && (a.stat_partials != nullptr || a.stat_acc != nullptr);   // forward: bn statistics of the output
    const bool fuse_bnred = !FWD && (PS || CAN_STATS) && (a.bnred_partials != nullptr || a.bnred_acc != nullptr);  // backward-data: dgamma / dbeta sums of the layer `out` belongs to
    constexpr bool PST = PS == 4 || PS == 5;
    const int stat_mode = fuse_stats ? 1 : (fuse_bnred && PS != 1 && PS != 5) ? 2 : 0;   // sums kept by the CONSUMER waves
    const bool ps = (PS == 1 || PS == 5) && fuse_bnred;                    // sums kept by the producer waves
    const bool pst = PST && fuse_bnred;                                    // global stores issued by the producer waves
    constexpr int E_BYTES = G::ACC * 128 * 64 * NT;                        // epilogue buffer: ACC x 128 pixel slots of NT x 64 bytes
    constexpr int EK = 4 * NT, EQ_STEP = 256 / EK, ECH = G::ACC * 128 / EQ_STEP;   // chunks per pixel; pixel slots between a thread's chunks; chunks per thread
    float est[16];   // PS: this producer thread's running sums (sum dz*y | sum dz of its 8 channels), reduced after the stat_mode reduction
    float* bnc = tab + c_red * 4;            // [scale | shift | mean | invstd][C_OUT] of this workgroup's channels

    int tile = tile_first, slab = 0, it = 0;
    auto init_tables = [&]() __attribute__((always_inline)) {   // every thread of the workgroup, from either branch below
        if (KIND == SRC_ACT || KIND == SRC_ACT2) {
            if (a.src.a_tab.acc) {   // table mode: the producers' (scale, shift) are folded here from their accumulator tables (bnacc.h)
                const int sides = KIND == SRC_ACT2 ? 2 : 1;
                for (int i = threadIdx.x; i < sides * c_red; i += NTHR) {
                    const int side = i >= c_red, ch = i - side * c_red;
                    const BnTable& t = side ? a.src.b_tab : a.src.a_tab;
                    const BnFolded f = bnacc_fold_forward(t.acc, t.c, ch, t.pixels, t.gamma[ch], t.beta[ch], t.eps);
                    tab[2 * side * c_red + ch] = f.scale;
                    tab[(2 * side + 1) * c_red + ch] = f.shift;
                }
                if (KIND != SRC_ACT2) for (int i = threadIdx.x; i < c_red; i += NTHR) { tab[2 * c_red + i] = 0.f; tab[3 * c_red + i] = 0.f; }
            } else {
                for (int i = threadIdx.x; i < c_red; i += NTHR) {
                    tab[i] = a.src.a_scale[i];
                    tab[c_red + i] = a.src.a_shift[i];
                    tab[2 * c_red + i] = KIND == SRC_ACT2 ? a.src.b_scale[i] : 0.f;
                    tab[3 * c_red + i] = KIND == SRC_ACT2 ? a.src.b_shift[i] : 0.f;
                }
            }
        }
        if (ACT) {
            for (int i = threadIdx.x; i < C_OUT; i += NTHR) {
                bnc[i] = a.out_scale[co_base + i];
                bnc[C_OUT + i] = a.out_shift[co_base + i];
            }
        }
        if (fuse_bnred) {
            for (int i = threadIdx.x; i < C_OUT; i += NTHR) {
                bnc[i] = a.bnred_scale[co_base + i];
                bnc[C_OUT + i] = a.bnred_shift[co_base + i];
                bnc[2 * C_OUT + i] = a.bnred_mean[co_base + i];
                bnc[3 * C_OUT + i] = a.bnred_invstd[co_base + i];
            }
        }
        __syncthreads();
    };
    if (producer) {
        // The producers request their FIRST patch and the filter blocks before the tables are loaded and the workgroup meets:
        // those round trips overlap instead of adding up.
        const bf16* wsrc = reinterpret_cast<const bf16*>(a.w_bf16);
        const bf16* xa = reinterpret_cast<const bf16*>(a.src.a);
        const bf16* xb = reinterpret_cast<const bf16*>(a.src.b);
        const size_t plane = (size_t)H * W * c_red;  // < 2^31 elements (host check)
        // ---- staging geometry, fixed per thread: patch chunk jj = record (tid >> 2) + 64 jj ----
        int pgeo[NP], pdst[NP];
    #pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const int rec = min((tid >> 2) + 64 * jj, G::RECS - 1);
            int py, px, key;
            G::decode(rec, py, px, key);
            pgeo[jj] = py | (px << 8);
            pdst[jj] = rec * 64 + ((c16 ^ key) << 4);
        }
        // filter chunk j = record (tid >> 2) + 64 j = (tap slot, co); slot t holds tap t, or 8 - t for mirrored taps
        int wsrc_off[NW], wdst[NW];
    #pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int rec = min((tid >> 2) + 64 * j, 9 * C_OUT - 1);
            const int tl = rec / C_OUT, co = rec - tl * C_OUT;
            const int tap = flip ? 8 - tl : tl;
            wsrc_off[j] = (tap * a.c_out + co_base + co) * c_red + c16 * 8;
            wdst[j] = rec * 64 + ((c16 ^ ((co >> 2) & 3)) << 4);   // within a filter slab
        }
        // ---- ONE item of loads in flight: the loads of item k + 1 are issued right after item k is committed, so a load has a whole
        // item period to arrive (a second register set measured slower three times: rounds 2, 3, 4).  The filter slab of the NEXT item
        // is fetched one item ahead into one shared register set: it comes from L2 (every workgroup reads the same filter). ----
        u32x4 wraw[NW];  // a native vector type: hipcc keeps a HIP uint4 that is only copied (never unpacked) in scratch
        struct Fetched {
            RawChunk<KIND> praw[NP];
            unsigned pok;
        };
        Fetched R0;
        // the fetch cursor runs ahead of the commits.  Its tile coordinates advance incrementally (no division per item)
        // and the chunk offsets / validity bits of its tile are computed once per tile, not once per (tile, slab) item.
        const int per_img = tiles_x * tiles_y;
        const int step_x = gstep % tiles_x, step_y = (gstep / tiles_x) % tiles_y, step_n = gstep / per_img;
        int ftile = tile_first, fslab = 0;
        int ftx = ftile % tiles_x, fty = (ftile / tiles_x) % tiles_y, fn = ftile / per_img;
        // Round 4, "VALU diet" (tools/micro/simd_sharing.hip: on a SIMD, MFMAs and the VALU instructions of EVERY wave share one issue port —
        // a conv item's time is the sum of its MFMA cycles and 4 x the VALU instructions of both roles): the patch and filter fetches are
        // BUFFER loads — address = per-image descriptor (SGPRs) + a 32-bit byte offset computed once per tile (VGPR) + the slab's offset
        // (SGPR): no sign extension, no 64-bit add per load (3 VALU each before); a padding pixel's offset lies outside the descriptor, so
        // the load returns zeros by itself and the plain-copy kinds need neither clamped addresses nor a mask (5 VALU per chunk before).
        constexpr bool NEED_MASK = KIND == SRC_ACT || KIND == SRC_ACT2;   // relu(0 * scale + shift) is not zero: the bn kinds zero their padding after the prologue
        const int plane_bytes = (int)(unsigned)(plane * 2);   // (< 0xFFFFF000: host check, conv_plan)
        int foff[NP];   // byte offset within the image (0xFFFFF000 = padding)
        unsigned fpok = 0;
        auto enter_tile = [&]() __attribute__((always_inline)) {
            const int x0 = G::in_x0(ftx), y0 = G::in_y0(fty);
            fpok = 0;
    #pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const int iy = y0 + (pgeo[jj] & 255), ix = x0 + (pgeo[jj] >> 8);
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                foff[jj] = ok ? ((iy * W + ix) * c_red + c16 * 8) * 2 : (int)0xFFFFF000u;   // (+ the slab's offset: still outside the image)
                if constexpr (NEED_MASK) fpok |= (ok ? 1u : 0u) << jj;
            }
        };
        auto fetch = [&](Fetched& R) __attribute__((always_inline)) {
            const bf16* pa = xa + (size_t)fn * plane;
            const bf16* pb = (KIND == SRC_ACT2 || KIND == SRC_SUM2) ? xb + (size_t)fn * plane : nullptr;
            const int cc = fslab * 32;
            R.pok = fpok;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(pa), 0, plane_bytes, 0x00020000);
    #pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, foff[jj], cc * 2, 0);
                R.praw[jj].a = make_uint4(v[0], v[1], v[2], v[3]);
            }
            if constexpr (KIND == SRC_ACT2 || KIND == SRC_SUM2) {
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(pb), 0, plane_bytes, 0x00020000);
    #pragma unroll
                for (int jj = 0; jj < NP; ++jj) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, foff[jj], cc * 2, 0);
                    R.praw[jj].b = make_uint4(v[0], v[1], v[2], v[3]);
                }
            }
            if (++fslab == n_slabs) {   // cursor -> the workgroup's next tile
                fslab = 0; ftile += gstep;
                ftx += step_x; if (ftx >= tiles_x) { ftx -= tiles_x; ++fty; }
                fty += step_y; if (fty >= tiles_y) { fty -= tiles_y; ++fn; }
                fn += step_n;
                if (ftile < n_tiles) enter_tile();
            }
        };
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(wsrc), 0, 9 * a.c_out * c_red * 2, 0x00020000);
        auto fetch_w = [&](int slab_) __attribute__((always_inline)) {
    #pragma unroll
            for (int j = 0; j < NW; ++j) wraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, wsrc_off[j] * 2, slab_ * 64, 0);
        };
        auto commit = [&](Fetched& R) __attribute__((always_inline)) {
            char* lbuf = smem + (it & 1) * x_stride;
            char* wbuf = lbuf + X_BYTES_;     // streaming form only
            // a single-slab layer's filter block goes into both buffers once (its registers are not refetched)
            const bool stage_w = !wres && (it < 2 || n_slabs > 1);
            float sa[8], ta[8], sb[8], tb[8];
            if (KIND == SRC_ACT || KIND == SRC_ACT2) {
                const float* t0 = tab + slab * 32 + c16 * 8;
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sa[j] = t0[j]; ta[j] = t0[c_red + j];
                    sb[j] = KIND == SRC_ACT2 ? t0[2 * c_red + j] : 0.f;
                    tb[j] = KIND == SRC_ACT2 ? t0[3 * c_red + j] : 0.f;
                }
            }
    #pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                uint4 v = chunk_convert<KIND>(R.praw[jj], sa, ta, sb, tb);
                if constexpr (NEED_MASK) { if (!((R.pok >> jj) & 1u)) v = make_uint4(0u, 0u, 0u, 0u); }
                if ((tid >> 2) + 64 * jj < G::RECS) *reinterpret_cast<uint4*>(lbuf + pdst[jj]) = v;
            }
            if (stage_w) {
    #pragma unroll
                for (int j = 0; j < NW; ++j)
                    if ((tid >> 2) + 64 * j < 9 * C_OUT) *reinterpret_cast<u32x4*>(wbuf + wdst[j]) = wraw[j];
            }
        };
        if (ftile < n_tiles) { enter_tile(); fetch(R0); if (!wres) fetch_w(0); }
        if (wres && tile < n_tiles) {   // resident form: every filter slab goes to its own LDS block once
            for (int sl = 0; sl < n_slabs; ++sl) {
                fetch_w(sl);
#pragma unroll
                for (int j = 0; j < NW; ++j)
                    if ((tid >> 2) + 64 * j < 9 * C_OUT) *reinterpret_cast<u32x4*>(smem + w_base + sl * W_BYTES + wdst[j]) = wraw[j];
            }
        }
        init_tables();
        // ---- PS: bn backward sums / global stores of the tile whose epilogue the consumers finished two items ago ----
        float esc[8], esh[8];
        const int ek = tid & (EK - 1), eq0 = tid / EK;   // this thread's chunk (channels 8 ek .. 8 ek + 7 of the workgroup's) and first pixel slot
        int hist1 = -1, hist2 = -1, hist1_it = 0, hist2_it = 0;   // tile (or -1) and item index of the last two items whose epilogue leaves a buffer
        if (PS == 1 || PS == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                esc[j] = bnc[ek * 8 + j]; esh[j] = bnc[C_OUT + ek * 8 + j];
                est[j] = 0.f; est[8 + j] = 0.f;
            }
        }
        const bf16* ylayer = reinterpret_cast<const bf16*>(a.bnred_y);
        unsigned epix[PST ? ECH : 1];
        auto estat_begin = [&](int et, u32x4 (&yv)[ECH], unsigned& evalid) __attribute__((always_inline)) {   // the y operands are requested ...
            const int tx = et % tiles_x, ty = (et / tiles_x) % tiles_y, n = et / per_img;
            evalid = 0;
#pragma unroll
            for (int j = 0; j < ECH; ++j) {
                const int q = eq0 + EQ_STEP * j;
                size_t pix; bool valid;
                G::out_pixel(q >> 7, a, n, ty, tx, (q >> 5) & 3, q & 31, pix, valid);
                if (PS != 4) yv[j] = *reinterpret_cast<const u32x4*>(ylayer + pix * a.c_out + co_base + ek * 8);
                if constexpr (PST) epix[j] = (unsigned)pix;   // (< 2^31 elements per tensor: host check) where this chunk's pixel is stored
                evalid |= (valid ? 1u : 0u) << j;
            }
        };
        auto estat_end = [&](int eit, const u32x4 (&yv)[ECH], unsigned evalid) __attribute__((always_inline)) {   // ... and meet the stored values here
            const char* eb = smem + e_off + (n_slabs == 1 ? (eit & 1) * E_BYTES : 0);
#pragma unroll
            for (int j = 0; j < ECH; ++j) {
                const int q = eq0 + EQ_STEP * j;
                const u32x4 dv = *reinterpret_cast<const u32x4*>(eb + q * (64 * NT) + ((ek ^ ebuf_swizzle<NT>(q)) << 4));
                if constexpr (PST) {   // the tile's global store, issued here
                    if ((evalid >> j) & 1u) store16(reinterpret_cast<bf16*>(a.out) + (size_t)epix[j] * a.c_out + co_base + ek * 8, make_uint4(dv[0], dv[1], dv[2], dv[3]));
                }
                if (PS != 4 && ((evalid >> j) & 1u)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int c = 2 * i + h2;
                            const float d = h2 ? hi_f(dv[i]) : lo_f(dv[i]), y = h2 ? hi_f(yv[j][i]) : lo_f(yv[j][i]);
                            const float dz = fmaf(y, esc[c], esh[c]) > 0.f ? d : 0.f;
                            est[c] = fmaf(dz, y, est[c]);   // sum dz*y (-> sum dz*xhat in the closing reduction, as the consumer-side form)
                            est[8 + c] += dz;
                        }
                }
            }
        };
        auto one_item = [&](Fetched& R) __attribute__((always_inline)) {
            u32x4 eyv[PS ? ECH : 1];
            unsigned evalid = 0;
            const int et = hist2, eit = hist2_it;
            if constexpr (PS != 0) { if ((ps || pst) && et >= 0) estat_begin(et, eyv, evalid); }
            TICK();
            commit(R);
#ifdef ANH_WS_PROFILE
            __builtin_amdgcn_s_waitcnt(0xc07f);  // the commit figure includes the LDS write drain
#endif
            TOCK(t_a);
            TICK();
            if (ftile < n_tiles) fetch(R);
            if (!wres && n_slabs > 1 && (slab + 1 < n_slabs || tile + gstep < n_tiles)) fetch_w(slab + 1 < n_slabs ? slab + 1 : 0);
            if constexpr (PS != 0) { if ((ps || pst) && et >= 0) estat_end(eit, eyv, evalid); }
            TOCK(t_b);
            TICK();
            __syncthreads();   // buffer `it` is full; the consumers are done with the buffer the next item goes to
            TOCK(t_c);
            hist2 = hist1; hist2_it = hist1_it;
            hist1 = slab == n_slabs - 1 ? tile : -1; hist1_it = it;
            if (++slab == n_slabs) { slab = 0; tile += gstep; }
            ++it;
        };
#ifdef ANH_WS_PROFILE
        t_loop = wall_clock64();
        c_loop = clock64();
#endif
        while (tile < n_tiles) one_item(R0);
#ifdef ANH_WS_PROFILE
        t_loop_end = wall_clock64();
        c_loop = clock64() - c_loop;
#endif
        if constexpr (PS != 0) {
            if (ps || pst) {   // the last two items: the one before the last is ready, the last one after the consumers' closing barrier
                u32x4 eyv[ECH];
                unsigned evalid = 0;
                if (hist2 >= 0) { estat_begin(hist2, eyv, evalid); estat_end(hist2_it, eyv, evalid); }
                __syncthreads();
                if (hist1 >= 0) { estat_begin(hist1, eyv, evalid); estat_end(hist1_it, eyv, evalid); }
            }
        }
    } else {
        init_tables();
        typename G::Bases b0;
        G::init(b0, smem, wave, col, half);
        const char* wb0[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wb0[ks] = swz_addr(smem, col, col >> 2, ks, half);   // + the slab's block offset per item
        f32x16 acc[G::ACC][NT];
        bf16x8 hw[2];   // HEAD: this lane's rows of the 1x1 head's MFMA operand (head_operand)
        float hbias[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HEAD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hbias[k] = k < a.head_k ? a.head_bias[k] : 0.f;
            head_operand(hw, a, col, half);
        }
        u32x4 old[FWD ? 1 : G::ACC][NT][2];   // prefetched old values of a read-modify-write destination (GeoUp)
        u32x4 yraw[FWD ? 1 : G::ACC][NT][2];  // prefetched raw outputs y of the layer whose da is written (fused bn backward reduction)
        // The 32-channel read-modify-write kernel (GeoUp, NT = 1: HBM-bound, short MFMA phase) keeps the old values a whole
        // tile ahead: those of tile t + 1 are requested while tile t runs, so their latency is covered by a full item and
        // not by the MFMA phase alone (measured: 133 -> 120 us on the 32x64 stride-2 backward-data; the same depth for the
        // y operand of the stride-1 kernels measured slower, 93 -> 105 us, and is not compiled).
        constexpr bool DEEP = !FWD && NT == 1 && G::RMW_PREFETCH;
        u32x4 old_n[DEEP ? G::ACC : 1][NT][2];
        const bool rmw_any = !FWD && G::RMW_PREFETCH && a.out_accumulate;
        const bool cons_bnred = fuse_bnred && PS != 1 && PS != 5;   // the consumers keep the bn backward sums (and fetch y for them)
        const bool pre_any = rmw_any || cons_bnred;
        auto prefetch_epilogue = [&](int t, auto& o, auto& y, bool want_old, bool want_y) __attribute__((always_inline)) {
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            const bf16* out = reinterpret_cast<const bf16*>(a.out);
            const bf16* yl = reinterpret_cast<const bf16*>(a.bnred_y);
#pragma unroll
            for (int g = 0; g < G::ACC; ++g) {
                size_t pix; bool valid;
                G::out_pixel(g, a, n, ty, tx, wave, col, pix, valid);
                const size_t e0 = pix * a.c_out + co_base + 8 * half;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        if constexpr (!FWD) {
                            if (want_old) o[g][nt][s2] = *reinterpret_cast<const u32x4*>(out + e0 + nt * 32 + 16 * s2);
                            if (want_y) y[g][nt][s2] = *reinterpret_cast<const u32x4*>(yl + e0 + nt * 32 + 16 * s2);
                        }
                    }
            }
        };
        if (DEEP && pre_any && tile < n_tiles) prefetch_epilogue(tile, old, yraw, rmw_any, false);
        float stat[NT][2][16];      // per-lane running sums of this lane's 8 channels per (nt, s): see store_pixel_tiles_rmw
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 16; ++e) stat[nt][s2][e] = 0.f;
        // filter fragments in registers (GeoS1, one channel tile, one reduction slab; wres_ bit 19 switches it off: ANH_WS_FILTER_REGS=0)
        constexpr bool WREG = std::is_same<G, GeoS1>::value && NT == 1;
        const bool wreg_on = WREG && n_slabs == 1 && !((wres_ >> 19) & 1);
        bf16x8 wreg[WREG ? 18 : 1];
        for (;;) {
            // the epilogue of tile `et`, whose last item was `eit`: right behind its MFMA phase
            auto epilogue = [&](int et, int eit) __attribute__((always_inline)) {
                const int tx = et % tiles_x, ty = (et / tiles_x) % tiles_y, n = et / (tiles_x * tiles_y);
                const bool rmw = rmw_any;
#pragma unroll
                for (int g = 0; g < G::ACC; ++g) {
                    size_t pix; bool valid;
                    G::out_pixel(g, a, n, ty, tx, wave, col, pix, valid);
#ifdef ANH_WS_PROFILE
                    if (prof_nostore) valid = false;
#endif
                    if constexpr (HEAD) store_pixel_tiles_head(acc[g][0], a, pix, n, valid, half, bnc, C_OUT, hw, hbias);
                    else if constexpr (ACT) store_pixel_tiles_act<NT>(acc[g], a, pix, valid, half, co_base, bnc, C_OUT);
                    else {
                        const bool to_ebuf = ps || pst;   // this tile's sums are the producers' / its stores are
                        store_pixel_tiles_rmw<NT>(acc[g], a, pix, valid, half, co_base, old[FWD ? 0 : g], rmw, stat, stat_mode, yraw[FWD ? 0 : g], bnc,
                                                  to_ebuf ? smem + e_off + (n_slabs == 1 ? (eit & 1) * E_BYTES : 0) : nullptr, (g * 4 + wave) * 32 + col, !pst);
                    }
                }
                if constexpr (DEEP) {
                    if (pre_any) {
#pragma unroll
                        for (int g = 0; g < G::ACC; ++g)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) old[g][0][s2] = old_n[g][0][s2];
                    }
                }
            };
            if (tile >= n_tiles) break;
            int ntile = tile, nslab = slab + 1;
            if (nslab == n_slabs) { nslab = 0; ntile += gstep; }
            const int boff = (it & 1) * x_stride;
            const int woff = wres ? w_base + slab * W_BYTES : boff + X_BYTES_;
            typename G::Bases b;
#pragma unroll
            for (int i = 0; i < G::NB; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b.x[i][ks] = b0.x[i][ks] + boff;
            const char* wb[2] = {wb0[0] + woff, wb0[1] + woff};
            if (slab == 0) {
#pragma unroll
                for (int g = 0; g < G::ACC; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[g][nt][r] = 0.f;
            }
            TICK();
            __syncthreads();  // buffer `it` is full
            TOCK(t_c);
            TICK();
            const bool last_slab = slab == n_slabs - 1;
            if (pre_any && last_slab) {  // epilogue operands travel while the MFMAs run
                if constexpr (DEEP) {
                    if (tile + gstep < n_tiles) prefetch_epilogue(tile + gstep, old_n, yraw, rmw_any, false);
                    prefetch_epilogue(tile, old, yraw, false, cons_bnred);
                } else prefetch_epilogue(tile, old, yraw, rmw_any, cons_bnred);
            }
            if constexpr (WREG) {
                if (wreg_on) {
                    if (it == 0) G::load_filter_regs(wreg, wb);   // (the filter block of a single-slab layer is in both patch buffers from items 0 / 1 on)
                    G::mfma_wreg(acc, b, wreg);
                } else G::template mfma<NT>(acc, b, wb);
            } else G::template mfma<NT>(acc, b, wb);
#ifdef ANH_WS_PROFILE
            __builtin_amdgcn_sched_barrier(0);
#endif
            TOCK(t_a);
#ifdef ANH_WS_PROFILE
            // how long the wave waits for its last MFMAs before the first accumulator register can be read (the matrix pipe's drain)
            TICK();
            if (last_slab) { asm volatile("v_mov_b32 %0, %1" : "=v"(prof_sink) : "v"(acc[0][0][0])); asm volatile("v_mov_b32 %0, %1" : "=v"(prof_sink) : "v"(acc[G::ACC - 1][NT - 1][15])); }
            __builtin_amdgcn_sched_barrier(0);
            TOCK(t_d);
#endif
            TICK();
            if (last_slab) epilogue(tile, it);
            TOCK(t_b);
            tile = ntile; slab = nslab; ++it;
        }
        if constexpr (PS != 0) { if (ps || pst) __syncthreads(); }   // the consumers' closing barrier: the last tile's epilogue buffer is complete
        if (stat_mode) {
            __syncthreads();  // (matched by the producers) every wave is done with the staging buffers
            float* red = reinterpret_cast<float*>(smem) + (size_t)(wave * 64 + lane) * (32 * NT);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(nt * 2 + s2) * 16 + e] = stat[nt][s2][e];
        }
    }
    if (stat_mode) {
        if (producer) __syncthreads();
        __syncthreads();
        // (channel, which) = 64 NT sums over the 4 consumer waves x 32 pixel columns, in a fixed order, in double: a group of
        // 512 / (64 NT) adjacent threads shares one sum (strided columns, then an xor-shuffle tree)
        constexpr int GRP = 512 / (64 * NT);   // 8 or 4 threads per sum
        const int t = threadIdx.x / GRP, part = threadIdx.x % GRP;
        const int ch = t >> 1, which = t & 1;
        const int nt = ch >> 5, s2 = (ch >> 4) & 1, hf = (ch >> 3) & 1, j = ch & 7;
        const float* red = reinterpret_cast<const float*>(smem);
        double sum = 0.0;
#pragma unroll 4
        for (int e = part; e < 128; e += GRP) {
            const int w = e >> 5, c = e & 31;
            sum += (double)red[(size_t)(w * 64 + hf * 32 + c) * (32 * NT) + (nt * 2 + s2) * 16 + which * 8 + j];
        }
#pragma unroll
        for (int off = 1; off < GRP; off <<= 1) sum += __shfl_xor(sum, off, 64);
        if (stat_mode == 2) {   // (sum dz*y, sum dz) of a channel sit in adjacent thread groups: -> (sum dz*xhat, sum dz)
            const double sum_dz = __shfl_down(sum, GRP, 64);
            if (which == 0) sum = (double)bnc[3 * C_OUT + ch] * (sum - (double)bnc[2 * C_OUT + ch] * sum_dz);
        }
        if (part == 0) {
            long long* table = fuse_stats ? a.stat_acc : a.bnred_acc;
            double* dst = fuse_stats ? a.stat_partials : a.bnred_partials;
            if (table) bnacc_add(table, (fuse_stats ? BNACC_SUM_Y : BNACC_SUM_DZ_XHAT) + which, a.c_out, co_base + ch, sum);
            else dst[((size_t)(co_base + ch) * 2 + which) * gridDim.x + blockIdx.x] = sum;
        }
    }
    if constexpr (PS == 1 || PS == 5) {
        if (ps) {   // the producers' 16 sums per thread -> (channel, which) = 64 NT sums, each over the 256 / EK threads that own the chunk, fixed order, in double
            __syncthreads();   // every wave is done with the staging buffers
            if (producer) {
                float* redw = reinterpret_cast<float*>(smem) + (size_t)tid * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) redw[e] = est[e];
            }
            __syncthreads();
            constexpr int GRP = 512 / (64 * NT);
            const int t = threadIdx.x / GRP, part = threadIdx.x % GRP;
            const int ch = t >> 1, which = t & 1, k = ch >> 3, j = ch & 7;
            const float* red = reinterpret_cast<const float*>(smem);
            double sum = 0.0;
#pragma unroll 4
            for (int m = part; m < EQ_STEP; m += GRP) sum += (double)red[(size_t)(k + EK * m) * 16 + which * 8 + j];
#pragma unroll
            for (int off = 1; off < GRP; off <<= 1) sum += __shfl_xor(sum, off, 64);
            {
                const double sum_dz = __shfl_down(sum, GRP, 64);
                if (which == 0) sum = (double)bnc[3 * C_OUT + ch] * (sum - (double)bnc[2 * C_OUT + ch] * sum_dz);
            }
            if (part == 0) {
                if (a.bnred_acc) bnacc_add(a.bnred_acc, BNACC_SUM_DZ_XHAT + which, a.c_out, co_base + ch, sum);
                else a.bnred_partials[((size_t)(co_base + ch) * 2 + which) * gridDim.x + blockIdx.x] = sum;
            }
        }
    }
    // table mode, backward sums: the last workgroup to finish leaves dgamma / dbeta / the apply coefficients of that layer
    if (fuse_bnred && a.bnred_acc) bnacc_finish_backward(a.bnred_finish, (int)(gridDim.x * gridDim.y));
#ifdef ANH_WS_PROFILE
    if (prof && lane == 0) {
        long long* o = prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (producer ? 4 : 0) + wave) * 8;
        o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = producer ? c_loop : it; o[4] = t_entry; o[5] = producer ? t_loop : t_d + (prof_sink == 12345.f ? 1 : 0); o[6] = t_loop_end; o[7] = wall_clock64();
    }
#endif
#undef TICK
#undef TOCK
}

int ws_target_wgs();

// LDS layout of a persistent conv launch and whether it takes a producer-side form of the fused bn backward sums / stores (PS).
// with_bnred: the sums are wanted (conv_fused_bnred_blocks asks before ConvArgs::bnred_* are set).
struct WsLayout { int wres; size_t lds; int ps /* 0, 1 (producer sums), 4 (producer stores), 5 (both) */; int e_off; };
WsLayout ws_layout(const ConvArgs& a, int recs, int acc, int nt, bool with_bnred) {
    const size_t tables = (size_t)a.c_red * 16 + (size_t)nt * 32 * 16;
    const size_t x_bytes = (size_t)recs * 64, w_bytes = (size_t)9 * nt * 32 * 64;
    // filter-resident form: two reduction slabs (64 channels) whose filter blocks fit beside two patches; the end-of-kernel
    // statistics reduction borrows 32 NT KiB from the start of the buffer
    static const bool resident_on = !(getenv("ANH_WS_WEIGHT_RESIDENT") && atoi(getenv("ANH_WS_WEIGHT_RESIDENT")) == 0);
    const int n_slabs = a.c_red >> 5;
    const size_t resident_lds = 2 * x_bytes + (size_t)n_slabs * w_bytes + tables;
    WsLayout L{};
    L.wres = resident_on && n_slabs >= 2 && resident_lds <= 160 * 1024 && 2 * x_bytes + (size_t)n_slabs * w_bytes >= (size_t)32 * 1024 * nt;
    L.lds = L.wres ? resident_lds : 2 * (x_bytes + w_bytes) + tables;
    // Producer-side forms: plain-copy staging with the fused reduction, when the epilogue buffer fits — one buffer of ACC x 128 pixel
    // slots, two (alternating by item) for single-slab layers whose every item ends a tile.  A single-slab layer may take the resident
    // layout to make room (one filter block instead of a copy per patch buffer; whatever ANH_WS_WEIGHT_RESIDENT says — which waves keep
    // the sums must not depend on a tuning switch, the summation order follows it).
    // ANH_WS_PSTAT: 7 (default) = the producer waves issue the global stores: consumer sums (PS = 4) for the stride-1 / down geometries,
    // producer sums + stores (PS = 5) for the four-group geometry at 32 channels (at 64 it spills: PS = 1, producer sums only);
    // 8 = the same with the four-group geometry as in round 3 (PS = 1); 1 = round 3's default (PS = 1 for the four-group geometry, the
    // consumers keep sums and stores elsewhere).  All three are bit-identical (tests/test_gpu_schedules.py).  Measured, same box:
    // 1 -> 7: 1.7356 / 1.7400 / 1.7374 / 1.7360 / 1.7383 -> 1.7264 / 1.7179 / 1.7178 / 1.7224 / 1.7187 ms per step (-1.0 %).
    static const int ps_env = getenv("ANH_WS_PSTAT") ? atoi(getenv("ANH_WS_PSTAT")) : 7;
    int mode = 0;
    if (ps_env == 1) mode = acc == 4 ? 1 : 0;
    else if (ps_env == 8) mode = acc == 4 ? 1 : (acc * nt <= 4 ? 4 : 0);
    else mode = acc == 4 ? (nt == 1 ? 5 : 1) : (acc * nt <= 4 ? 4 : 0);
    const size_t e_total = (size_t)acc * 128 * 64 * nt * (n_slabs == 1 ? 2 : 1);
    if (mode && with_bnred && a.src.kind == SRC_RAW && !a.out_scale && !a.stat_partials && !a.stat_acc) {
        if (L.lds + e_total <= 160 * 1024) L.ps = mode;
        else if (n_slabs == 1 && resident_lds + e_total <= 160 * 1024 && resident_lds - tables >= 16 * 1024) { L.wres = 1; L.lds = resident_lds; L.ps = mode; }
    }
    L.e_off = L.ps ? (int)((L.lds + 15) / 16 * 16) : 0;
    if (L.ps) L.lds = (size_t)L.e_off + e_total;
    return L;
}

template <class G, int NT>
void launch_ws(const ConvArgs& a, int tiles_x, int tiles_y, int flip, hipStream_t s) {
    const int n_tiles = tiles_x * tiles_y * a.n, groups = a.c_out / (NT * 32);
    const dim3 grid((unsigned)std::max(1, std::min(n_tiles, ws_target_wgs() / groups)), (unsigned)groups);
    const WsLayout lay = ws_layout(a, G::RECS, G::ACC, NT, a.bnred_partials != nullptr || a.bnred_acc != nullptr);
    const int wres = lay.wres;
    const size_t lds = lay.lds;
    const int e_off = lay.e_off;
    static const int filter_regs_env = getenv("ANH_WS_FILTER_REGS") ? atoi(getenv("ANH_WS_FILTER_REGS")) : 1;
    const int ps = lay.ps;
    // ANH_WS_XCD_BANDS (1): the workgroups of one XCD walk one contiguous eighth of the tile list (see the kernel); 0 = strided walk
    static const int bands_env = getenv("ANH_WS_XCD_BANDS") ? atoi(getenv("ANH_WS_XCD_BANDS")) : 1;
    const int bands = bands_env && grid.x % 8 == 0 && grid.x >= 8 ? 1 : 0;
    const int flags = wres | (bands << 4) | ((filter_regs_env ? 0 : 1) << 19);   // (bit 6: the profiling build's no-store run)
    auto launch = [&](auto kernel) {
        const dim3 block(512);
        ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
#ifndef ANH_WS_PROFILE
        hipLaunchKernelGGL(kernel, grid, block, lds, s, a, tiles_x, tiles_y, flip, (long long*)nullptr, flags, e_off);
#else
        static const int prof_on = getenv("ANH_WS_PROF") ? atoi(getenv("ANH_WS_PROF")) : 0;
        static long long* prof = nullptr;
        if (prof_on && !prof) HIP_CHECK(hipMalloc(&prof, 1024 * 12 * 8 * sizeof(long long)));
        static const int nostore = getenv("ANH_WS_PROF_NOSTORE") ? atoi(getenv("ANH_WS_PROF_NOSTORE")) : 0;
        hipLaunchKernelGGL(kernel, grid, block, lds, s, a, tiles_x, tiles_y, flip, prof_on ? prof : nullptr, flags | (nostore << 6), e_off);
        if (prof_on) {
            HIP_CHECK(hipStreamSynchronize(s));
            const int nwg = grid.x * grid.y, nw = 8, nc = 4;   // waves per workgroup, consumer waves among them
            std::vector<long long> h((size_t)nwg * nw * 8);
            HIP_CHECK(hipMemcpy(h.data(), prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double pa = 0, pb = 0, pc = 0, ca = 0, cb = 0, cc = 0, cd = 0, items = 0;
            long long first_entry = h[4], last_exit = h[7];
            double entry = 0, pro = 0, loop = 0, tail = 0, cycles = 0;   // over the producer waves: entry skew, entry -> loop, loop, loop end -> exit; shader cycles of the loop
            for (int w = 0; w < nwg; ++w) {
                for (int v = 0; v < nw; ++v) { first_entry = std::min(first_entry, h[(w * nw + v) * 8 + 4]); last_exit = std::max(last_exit, h[(w * nw + v) * 8 + 7]); }
                for (int v = 0; v < nc; ++v) { ca += h[(w * nw + v) * 8]; cb += h[(w * nw + v) * 8 + 1]; cc += h[(w * nw + v) * 8 + 2]; cd += h[(w * nw + v) * 8 + 5]; }
                for (int v = nc; v < nw; ++v) { pa += h[(w * nw + v) * 8]; pb += h[(w * nw + v) * 8 + 1]; pc += h[(w * nw + v) * 8 + 2]; }
                items += h[(w * nw) * 8 + 3];
            }
            for (int w = 0; w < nwg; ++w)
                for (int v = nc; v < nw; ++v) {   // producer waves carry the loop stamps
                    const long long* o = &h[(w * nw + v) * 8];
                    entry += (double)(o[4] - first_entry); pro += (double)(o[5] - o[4]); loop += (double)(o[6] - o[5]); tail += (double)(o[7] - o[6]);
                    cycles += (double)o[3];
                }
            const double k = 1.0 / (4.0 * nwg) / 100.0;  // wall clock = 100 MHz -> us per wave
            fprintf(stderr, "[ws prof] geo_recs=%d NT=%d kind=%d c_red=%d wgs=%d items/wg=%.1f | producer us: commit %.1f fetch %.1f barrier %.1f | consumer us: mfma %.1f drain %.1f store %.1f barrier %.1f"
                            " | kernel %.1f us = launch skew %.1f + prologue %.1f + loop %.1f + tail %.1f (+ drain to the last wave) | clock in the loop %.0f MHz\n",
                    G::RECS, NT, a.src.kind, a.c_red, nwg, items / nwg, pa * k, pb * k, pc * k, ca * k, cd * k, cb * k, cc * k,
                    (double)(last_exit - first_entry) / 100.0, entry * k, pro * k, loop * k, tail * k, loop > 0 ? cycles / loop * 100.0 : 0.0);
        }
#endif
    };
    if (a.out_scale) {   // inference: activation-storing epilogue, plain-copy (or skip-add) staging
        ANH_REQUIRE(a.out_shift && !a.stat_partials && !a.bnred_partials && !a.stat_acc && !a.bnred_acc && !a.out_accumulate && !a.out2, "conv_ws: the activation-storing form takes no training epilogue");
        if (a.head_out) {   // the layer under the 1x1 head: logits instead of the activation
            if constexpr (std::is_same<G, GeoS1>::value && NT == 1) {
                ANH_REQUIRE(conv_head_in_epilogue_ok(a) && a.head_w && a.head_bias, "conv_ws: this layer cannot take the head in its epilogue");
                switch (a.src.kind) {
                    case SRC_RAW: launch(conv3x3_ws_kernel<G, NT, SRC_RAW, true, true, 0, true>); break;
                    case SRC_SUM2: launch(conv3x3_ws_kernel<G, NT, SRC_SUM2, true, true, 0, true>); break;
                    default: fail(ANH_ERR_INTERNAL, "conv_ws: the activation-storing form reads post-activation tensors");
                }
            } else fail(ANH_ERR_INTERNAL, "conv_ws: the head-in-epilogue form exists for the stride-1 32-channel kernel only");
            HIP_CHECK(hipGetLastError());
            return;
        }
        switch (a.src.kind) {
            case SRC_RAW: launch(conv3x3_ws_kernel<G, NT, SRC_RAW, true, true>); break;
            case SRC_SUM2: launch(conv3x3_ws_kernel<G, NT, SRC_SUM2, true, true>); break;
            default: fail(ANH_ERR_INTERNAL, "conv_ws: the activation-storing form reads post-activation tensors");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    // The forward-only form (every geometry has one): a training forward that runs the general form carries the registers of the
    // backward epilogues (old values, y operands: 64 VGPRs at two accumulator groups x two channel tiles) through its MFMA phase.
    // Same-box A/B (round 3): 1.757 -> 1.738 ms per step; the skip-add 64->64 forward 72.7 -> 63.1 us.
    const bool fwd_form = (a.stat_partials || a.stat_acc) && !a.bnred_partials && !a.bnred_acc && !a.out_accumulate && !a.out2;
    if (fwd_form) {
        switch (a.src.kind) {
            case SRC_RAW: launch(conv3x3_ws_kernel<G, NT, SRC_RAW, true>); break;
            case SRC_ACT: launch(conv3x3_ws_kernel<G, NT, SRC_ACT, true>); break;
            case SRC_ACT2: launch(conv3x3_ws_kernel<G, NT, SRC_ACT2, true>); break;
            default: fail(ANH_ERR_INTERNAL, "conv_ws: no forward-only form for this source kind");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (ps == 1) { launch(conv3x3_ws_kernel<G, NT, SRC_RAW, false, false, 1>); HIP_CHECK(hipGetLastError()); return; }
    if constexpr (G::ACC < 4 && G::ACC * NT <= 4) {   // (ws_layout gives the four-group geometry 5 or 1, never 4)
        if (ps == 4 && !a.out2) { launch(conv3x3_ws_kernel<G, NT, SRC_RAW, false, false, 4>); HIP_CHECK(hipGetLastError()); return; }
    }
    if (ps == 5 && !a.out2) { launch(conv3x3_ws_kernel<G, NT, SRC_RAW, false, false, 5>); HIP_CHECK(hipGetLastError()); return; }
    if (ps == 5) { launch(conv3x3_ws_kernel<G, NT, SRC_RAW, false, false, 1>); HIP_CHECK(hipGetLastError()); return; }
    switch (a.src.kind) {
        case SRC_RAW: launch(conv3x3_ws_kernel<G, NT, SRC_RAW>); break;
        case SRC_ACT: launch(conv3x3_ws_kernel<G, NT, SRC_ACT>); break;
        default: launch(conv3x3_ws_kernel<G, NT, SRC_ACT2>); break;
    }
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad_stem_mfma: filter gradient of the 5x5 stem on the u8 image, bf16 dy.
//   dW[(tap,ci)][co] = sum_p img(p + tap - 2)[ci]/256 * dy[p][co]      -> a (25*CIN -> 96 or 32) x 32 MFMA output, K = pixels.
// dy is fetched with the transposing LDS read as in wgrad3x3_mfma.  The image operand needs, per lane, 8 consecutive
// pixels of one (tap, ci) row: the patch is kept in LDS as bf16 planes [ci][row][kx][32] — five copies shifted by the
// tap's kx — so that every fragment is ONE 16-byte-aligned ds_read_b128 (u8/256 is exact in bf16).  Rows beyond
// 25*CIN read an all-zero plane.  The 16 k-steps of an 8x32 tile are split over the 4 waves; accumulators persist
// over the workgroup's tiles; every wave writes its own partial (fixed-order reduction afterwards).
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, bool BNBWD>
__global__ __launch_bounds__(256, 2) void wgrad_stem_mfma_kernel(WgradArgs a, WgSide dy_side, int tiles_x, int tiles_y, int total_tiles, int splits) {
    constexpr int ROWS = 25 * CIN, RT = (ROWS + 31) / 32, PLANE = 12 * 5 * 64 + 64;  // bytes of one channel plane (+ 64: planes start 16 banks apart)
    constexpr int IMG_THREADS = 12 * 4 * CIN;   // image staging: one thread per (patch row, 8-pixel segment, channel)
    constexpr int BUF = 256 * 64 + (CIN + 1) * PLANE;  // one buffer: dy tile (256 pixel records) + CIN planes + an all-zero plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31, c16 = tid & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < PLANE / 4; i += 256) {  // the zero planes, read by the rows beyond 25*CIN
        reinterpret_cast<unsigned*>(smem + 256 * 64 + CIN * PLANE)[i] = 0u;
        reinterpret_cast<unsigned*>(smem + BUF + 256 * 64 + CIN * PLANE)[i] = 0u;
    }

    // ---- operand addresses: lane part once, k-step part as compile-time constants.  A k-step = 16 pixels of one tile
    //      row; wave w owns rows 2w, 2w+1 ----
    int a_off[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = rt * 32 + col;
        const int tap = r / CIN, ci = r - tap * CIN;
        const int ky = tap / 5, kx = tap - ky * 5;
        a_off[rt] = 256 * 64 + (r < ROWS ? ci * PLANE + (ky * 5 + kx) * 64 : CIN * PLANE) + half * 16 + wave * 2 * 320;
    }
    int g_off[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) g_off[rr] = wave * 2 * 32 * 64 + tr_lane_offset(lane, rr, 0);

    // ---- staging geometry, fixed per thread: image byte jj -> (channel, column, row) of the 12 x 36 patch; dy chunk jj ->
    //      row (tid >> 7) + 2 jj, column (tid >> 2) & 31 ----
    // Image staging (round 5): thread (patch row, segment q, channel) loads the 12 pixels 8 q .. 8 q + 11 of its 36-pixel line — byte loads
    // through a buffer descriptor of the image, so a pixel outside the net's input reads as zero by itself — converts them to six packed
    // bf16 pairs and writes the five kx-shifted 8-pixel chunks as 16-byte stores (even shifts are four of the six words, odd shifts four
    // v_alignbit).  No branch, ~50 VALU per thread and tile on 144 threads.  The form before handled one byte per thread with five
    // predicated 2-byte stores each (hipcc emits a branch per store: ~60 instructions and 5 branches per byte, 1,296 bytes per tile).
    const bool img_thread = tid < IMG_THREADS;
    const int i_ci = tid % CIN, i_q = (tid / CIN) & 3, i_py = min(tid / (4 * CIN), 11);
    const int i_dst = 256 * 64 + i_ci * PLANE + i_py * 320 + i_q * 16;   // + kx * 64
    const int img_bytes = a.src.img_h * a.src.img_w * CIN;   // (< 2^31: stem_wgrad_mfma_ok)
    const int t_x = (tid >> 2) & 31, t_row0 = tid >> 7;
    const int gdst0 = (t_row0 * 32 + t_x) * 64 + ((c16 ^ ((t_x >> 2) & 3)) << 4);

    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

    // ---- software pipeline over this workgroup's tiles: the loads of tile i + 1 are issued behind the barrier of tile i and fly while
    // its MFMAs run.  Round 5 measured two deeper forms, same box, four alternating rounds (profiles/r05_ab_log.txt): a second register
    // set (loads of tile i + 2 in flight; does not fit 256 VGPRs: 70 spilled) 1.678 -> 1.846 ms per step; the one set refilled chunk by
    // chunk as soon as a chunk has been converted 1.678 -> 1.686 (noise).  The kernel is not waiting for its loads: its waves issue
    // VALU instructions for 27 % of their cycles at two waves per SIMD (profiles/r04_sq_counters.txt) — the staging arithmetic is the limit. ----
    struct Regs {
        unsigned char ipx[12];
        unsigned gok;
        uint4 graw[4], yraw[4];
    };
    Regs R0;
    // BNBWD: dy is computed from (da, y) while staging; this thread's 8 channels' constants stay in registers
    float bsc[8], bsf[8], bm[8], bis[8], bk0[8], bk1[8], bk2[8];
    if (BNBWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = c16 * 8 + j;
            bsc[j] = a.dy_scale[ch]; bsf[j] = a.dy_shift[ch]; bm[j] = a.dy_mean[ch]; bis[j] = a.dy_invstd[ch];
            bk0[j] = a.dy_coef[ch]; bk1[j] = a.dy_coef[32 + ch]; bk2[j] = a.dy_coef[64 + ch];
        }
    }
    const bf16* dyy = reinterpret_cast<const bf16*>(a.dy_y);
    const bf16* dyp = reinterpret_cast<const bf16*>(a.dy);
    const size_t dy_plane = (size_t)a.h_out * a.w_out * 32;
    struct TileAt { int x0, y0, n; };
    auto tile_at = [&](int tile) __attribute__((always_inline)) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        return TileAt{tx * 32, ty * 8, n};
    };
    auto fetch_img = [&](Regs& R, const TileAt& t) __attribute__((always_inline)) {
        if (!img_thread) return;
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.src.img + (size_t)t.n * a.src.img_sample_stride), 0, img_bytes, 0x00020000);
        const int iy = t.y0 - 2 + i_py, ix0 = t.x0 - 2 + 8 * i_q;
        const int top = a.src.win_top(t.n), left = a.src.win_left(t.n);
        const bool interior = iy >= 0 && iy < a.h_in && ix0 >= 0 && ix0 + 11 < a.w_in && top + iy >= 0 && top + iy < a.src.img_h && left + ix0 >= 0 && left + ix0 + 11 < a.src.img_w;
        if (interior) {   // (nearly wave-uniform: 70 % of the tiles are interior ones)
            const int off = ((top + iy) * a.src.img_w + left + ix0) * CIN + i_ci;
#pragma unroll
            for (int k = 0; k < 12; ++k) R.ipx[k] = __builtin_amdgcn_raw_buffer_load_b8(ri, off + k * CIN, 0, 0);
        } else {
            const int sy = min(max(top + iy, 0), a.src.img_h - 1);
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int ix = ix0 + k, sx = min(max(left + ix, 0), a.src.img_w - 1);
                const bool ok = iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;   // the conv's zero padding
                R.ipx[k] = __builtin_amdgcn_raw_buffer_load_b8(ri, ok ? (sy * a.src.img_w + sx) * CIN + i_ci : (int)0x7FFFFFF0, 0, 0);
            }
        }
    };
    auto fetch_dy = [&](Regs& R, const TileAt& t, int jj) __attribute__((always_inline)) {
        const bf16* g = dyp + (size_t)t.n * dy_plane;
        const int ox = t.x0 + t_x, cx = min(ox, a.w_out - 1);
        const int oy = t.y0 + t_row0 + 2 * jj, cy = min(oy, a.h_out - 1);
        R.graw[jj] = *reinterpret_cast<const uint4*>(g + (cy * a.w_out + cx) * 32 + c16 * 8);
        if (BNBWD) R.yraw[jj] = *reinterpret_cast<const uint4*>(dyy + (size_t)t.n * dy_plane + (cy * a.w_out + cx) * 32 + c16 * 8);
        const unsigned ok = (oy == cy && ox == cx) ? 1u : 0u;
        R.gok = (R.gok & ~(1u << jj)) | (ok << jj);
    };
    auto fetch = [&](Regs& R, int tile) __attribute__((always_inline)) {
        const TileAt t = tile_at(tile);
        R.gok = 0;
        fetch_img(R, t);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fetch_dy(R, t, jj);
    };

    int tile = blockIdx.x;
    if (tile < total_tiles) fetch(R0, tile);
    int buf = 0;
    auto one_tile = [&](Regs& R) __attribute__((always_inline)) {
        char* lbuf = smem + buf * BUF;
        if (img_thread) {   // image patch (12 x 36 x CIN) -> five shifted bf16 copies per row; u8 / 256 is exact in bf16
            unsigned d[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = pack2((float)R.ipx[2 * k] * (1.0f / 256.0f), (float)R.ipx[2 * k + 1] * (1.0f / 256.0f));
            char* dst = lbuf + i_dst;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                uint4 v;
                if (kx & 1) {
                    const int b = kx >> 1;
                    v = make_uint4(__builtin_amdgcn_alignbit(d[b + 1], d[b], 16), __builtin_amdgcn_alignbit(d[b + 2], d[b + 1], 16),
                                   __builtin_amdgcn_alignbit(d[b + 3], d[b + 2], 16), __builtin_amdgcn_alignbit(d[b + 4], d[b + 3], 16));
                } else v = make_uint4(d[kx >> 1], d[(kx >> 1) + 1], d[(kx >> 1) + 2], d[(kx >> 1) + 3]);
                *reinterpret_cast<uint4*>(dst + kx * 64) = v;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const uint4 dyv = BNBWD ? chunk_bnbwd(R.graw[jj], R.yraw[jj], bsc, bsf, bm, bis, bk0, bk1, bk2) : R.graw[jj];
            const uint4 v = ((R.gok >> jj) & 1u) ? dyv : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(lbuf + gdst0 + jj * (2 * 32 * 64)) = v;
        }
        __syncthreads();  // two buffers: the writes above cannot race with a slower wave still reading the other one
        if (tile + splits < total_tiles) fetch(R, tile + splits);
        const char* g0 = lbuf + g_off[0];
        const char* g1 = lbuf + g_off[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // k-steps 4*wave + i: row 2*wave + (i >> 1), columns 16*(i & 1)..
            const int gimm = ((i >> 1) * 32 + ((i & 1) << 4)) * 64, aimm = (i >> 1) * 320 + ((i & 1) << 4) * 2;
            const bf16x8 gf = tr_read8_at(g0 + gimm, g1 + gimm);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(lbuf + a_off[rt] + aimm);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, gf, acc[rt], 0, 0, 0);
            }
        }
        buf ^= 1;
        tile += splits;
    };
    while (tile < total_tiles) one_tile(R0);

    // ---- sum the four waves' accumulators through LDS (fixed order) and write ONE partial per workgroup ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][rt][r][lane], 4 * RT * 16 * 64 floats <= 2 * BUF
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * RT + rt) * 16 + r) * 64 + lane] = acc[rt][r];
    __syncthreads();
    float* out = a.partials + (size_t)blockIdx.x * ROWS * 32;
    for (int e = tid; e < RT * 16 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, rt = e >> 10;
        const float sum = ((red[e] + red[RT * 1024 + e]) + red[2 * RT * 1024 + e]) + red[3 * RT * 1024 + e];
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (row < ROWS) out[row * 32 + (l & 31)] = sum;
    }
}

bool stem_wgrad_mfma_ok(const WgradArgs& a) {
    return a.src.kind == SRC_IMAGE && a.k == 5 && a.stride == 1 && a.pad == 2 && a.gather == 0 && a.c_out == 32 && (a.c_in == 1 || a.c_in == 3) &&
           a.h_in == a.h_out && a.w_in == a.w_out && a.dy_dtype == DT_BF16 &&
           (int64_t)a.src.img_h * a.src.img_w * a.c_in < 0x7FFFFFF0ll;   // (one image behind a buffer descriptor, 32-bit byte offsets)
}
int stem_wgrad_mfma_blocks(const WgradArgs& a) {
    const int tiles = ((a.w_out + 31) / 32) * ((a.h_out + 7) / 8) * a.n;
    static const int target = getenv("ANH_STEM_WGRAD_BLOCKS") ? atoi(getenv("ANH_STEM_WGRAD_BLOCKS")) : 512;
    return std::max(1, std::min(tiles, target));
}
void launch_wgrad_stem_mfma(const WgradArgs& a, hipStream_t s) {
    const int tiles_x = (a.w_out + 31) / 32, tiles_y = (a.h_out + 7) / 8;
    const int total = tiles_x * tiles_y * a.n, blocks = stem_wgrad_mfma_blocks(a);
    const WgSide dy{reinterpret_cast<const bf16*>(a.dy), nullptr, nullptr, nullptr, nullptr, nullptr, a.h_out, a.w_out, 32};
    const size_t lds = 2 * (256 * 64 + (size_t)(a.c_in + 1) * (12 * 5 * 64 + 64));  // two buffers, each with its zero plane
    if (a.dy_y) {
        if (a.c_in == 3) hipLaunchKernelGGL((wgrad_stem_mfma_kernel<3, true>), dim3(blocks), dim3(256), lds, s, a, dy, tiles_x, tiles_y, total, blocks);
        else hipLaunchKernelGGL((wgrad_stem_mfma_kernel<1, true>), dim3(blocks), dim3(256), lds, s, a, dy, tiles_x, tiles_y, total, blocks);
    } else if (a.c_in == 3) hipLaunchKernelGGL((wgrad_stem_mfma_kernel<3, false>), dim3(blocks), dim3(256), lds, s, a, dy, tiles_x, tiles_y, total, blocks);
    else hipLaunchKernelGGL((wgrad_stem_mfma_kernel<1, false>), dim3(blocks), dim3(256), lds, s, a, dy, tiles_x, tiles_y, total, blocks);
    HIP_CHECK(hipGetLastError());
    if (a.splits_out) *a.splits_out = blocks; else launch_reduce_partials(a.partials, blocks, (int64_t)25 * a.c_in * 32, a.dw, s);
}

struct WgPlan { int stride, ntc, zgroups, slabs, tiles_x, tiles_y, total, splits; size_t lds; bool cont, ws; };

WgPlan wgrad_plan_mfma(const WgradArgs& a) {
    WgPlan p{};
    p.stride = a.stride;
    p.cont = a.gather == 1;
    static const int ws_on = getenv("ANH_WGRAD_WS") ? atoi(getenv("ANH_WGRAD_WS")) : 1;
    p.ws = ws_on != 0;
    // tile channels per workgroup: up to 128 (ntc 4: nine accumulator tiles per consumer wave, 216-232 VGPRs; two workgroup groups of 64
    // measured 3-7 us slower on the filter gradients of the 56^2 level, round 3)
    const int ntc_ = std::min((p.cont ? a.c_in : a.c_out) / 32, 4);
    const int th = a.stride == 1 ? ((p.ws && ntc_ == 4) ? 4 : 8) : 4;
    const int lr_h = p.cont ? a.h_in : a.h_out, lr_w = p.cont ? a.w_in : a.w_out;   // the low-res (tile) tensor
    const int c_tile = p.cont ? a.c_in : a.c_out, c_patch = p.cont ? a.c_out : a.c_in;
    p.ntc = ntc_;
    p.zgroups = c_tile / (p.ntc * 32);   // 256 tile channels: two groups of 128 (warp-specialised kernel only)
    p.slabs = c_patch / 32;
    p.tiles_x = (lr_w + 31) / 32;
    p.tiles_y = (lr_h + th - 1) / th;
    p.total = p.tiles_x * p.tiles_y * a.n;
    p.lds = (size_t)(a.stride == 1 ? (th + 2) * 34 : 594) * 64 + (size_t)p.ntc * th * 32 * 64;
    const size_t tab = (size_t)p.ntc * 32 * 16;            // the tile side's bn constants
    if (p.ws) p.lds = 2 * p.lds + tab;                     // the warp-specialised kernel always runs two buffers
    else p.lds = (2 * p.lds + tab <= 160 * 1024 ? 2 * p.lds : p.lds) + tab;  // double-buffered when it fits (the kernel makes the same decision)
    // one workgroup per CU: these kernels share the chip with the backward-data chain (second stream), and every
    // workgroup writes a full partial, so fewer workgroups also means less partial-sum traffic
    static const int target = getenv("ANH_WGRAD_WGS") ? atoi(getenv("ANH_WGRAD_WGS")) : 256;
    p.splits = std::max(1, std::min(p.total, target / (p.slabs * p.zgroups)));
    return p;
}

template <int NTC, int KP, int KT, int STRIDE>
void launch_wg(const WgParams& prm, const WgPlan& p, hipStream_t s) {
    if (p.ws) {
        auto launch = [&](auto kernel) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), p.lds);
            // MEASURED (13 same-box rounds in two calls): 1.7314 -> 1.7217 and 1.7026 -> 1.6960 ms per step (-0.4 ... -0.6 %).  Default 1.
            static const int wbands_env = getenv("ANH_WGRAD_XCD_BANDS") ? atoi(getenv("ANH_WGRAD_XCD_BANDS")) : 1;
            const int wbands = wbands_env && p.splits % 8 == 0 && p.splits >= 8 ? 1 : 0;
            hipLaunchKernelGGL(kernel, dim3(p.splits, p.slabs, p.zgroups), dim3(512), p.lds, s, prm, p.tiles_x, p.tiles_y, p.total, p.splits | (wbands << 30));
        };
        constexpr bool can_be_cont = STRIDE == 2 && KP == SRC_RAW;   // cont: the patch is dy (raw), its channels are the output channels
        if constexpr (can_be_cont) {
            if (p.cont) { launch(wgrad3x3_ws_kernel<NTC, KP, KT, STRIDE, true>); return; }
        }
        ANH_REQUIRE(!p.cont, "wgrad_mfma: transposed layer on a non-transposing kernel");
        if constexpr (KT == SRC_RAW) launch(wgrad3x3_ws_kernel<NTC, KP, KT, STRIDE, false>);
        return;
    }
    ANH_REQUIRE(p.zgroups == 1, "wgrad_mfma: 256 tile channels need the warp-specialised kernel");
    auto kernel = wgrad3x3_mfma_kernel<NTC, KP, KT, STRIDE>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), p.lds);
    hipLaunchKernelGGL(kernel, dim3(p.splits, p.slabs), dim3(256), p.lds, s, prm, p.tiles_x, p.tiles_y, p.total, p.splits);
}

template <int NTC, int STRIDE>
void launch_wg_kinds(const WgParams& prm, const WgPlan& p, int kp, int kt, hipStream_t s) {
    if (kt == SRC_RAW) {
        if (kp == SRC_RAW) launch_wg<NTC, SRC_RAW, SRC_RAW, STRIDE>(prm, p, s);
        else if (kp == SRC_ACT) launch_wg<NTC, SRC_ACT, SRC_RAW, STRIDE>(prm, p, s);
        else launch_wg<NTC, SRC_ACT2, SRC_RAW, STRIDE>(prm, p, s);
    } else if (STRIDE == 2) {  // cont: the patch is dy (raw), the tile carries the prologue
        if (kt == SRC_ACT) launch_wg<NTC, SRC_RAW, SRC_ACT, 2>(prm, p, s);
        else launch_wg<NTC, SRC_RAW, SRC_ACT2, 2>(prm, p, s);
    }
}

void launch_wgrad_any(const WgradArgs& a, hipStream_t s) {
    const WgPlan p = wgrad_plan_mfma(a);
    WgSide x{reinterpret_cast<const bf16*>(a.src.a), a.src.a_scale, a.src.a_shift, reinterpret_cast<const bf16*>(a.src.b), a.src.b_scale, a.src.b_shift,
             a.h_in, a.w_in, a.c_in};
    WgSide g{reinterpret_cast<const bf16*>(a.dy), nullptr, nullptr, nullptr, nullptr, nullptr, a.h_out, a.w_out, a.c_out};
    WgParams prm{};
    prm.patch = p.cont ? g : x;
    prm.tile = p.cont ? x : g;
    prm.n = a.n; prm.c_in = a.c_in; prm.c_out = a.c_out; prm.transpose_out = p.cont ? 1 : 0;
    prm.partials = a.partials;
    const int kp = p.cont ? SRC_RAW : a.src.kind, kt = p.cont ? a.src.kind : SRC_RAW;
    if (a.stride == 1) {
        if (p.ntc == 1) launch_wg_kinds<1, 1>(prm, p, kp, kt, s);
        else if (p.ntc == 2) launch_wg_kinds<2, 1>(prm, p, kp, kt, s);
        else launch_wg_kinds<4, 1>(prm, p, kp, kt, s);
    } else {
        if (p.ntc == 1) launch_wg_kinds<1, 2>(prm, p, kp, kt, s);
        else if (p.ntc == 2) launch_wg_kinds<2, 2>(prm, p, kp, kt, s);
        else launch_wg_kinds<4, 2>(prm, p, kp, kt, s);
    }
    HIP_CHECK(hipGetLastError());
    const int64_t nw = (int64_t)9 * a.c_in * a.c_out;
    if (a.splits_out) *a.splits_out = p.splits; else launch_reduce_partials(a.partials, p.splits, nw, a.dw, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 layers, same machinery as conv3x3s1_mfma (patch slab in LDS as swizzled 64-byte records, weights staged by
// tap group, A = weights, B = pixels, permlane-swapped 16-byte NHWC stores).
//   conv3x3_down_mfma  (gather 0: con s2 forward, cont backward-data):  out[o] = sum_t in[2o + t] * W[t]
//       4x32 output tile; the 9x65 input patch is stored split by column parity so that the lanes of a wave (consecutive
//       output columns) read consecutive records for every tap.
//   conv3x3_up_mfma    (gather 1: cont forward, con s2 backward-data):  out[2i + t] += in[i] * W[t]
//       4x32 tile of LOW-RES positions; the four output parity classes (even/odd row x column) take 4 / 2 / 2 / 1 taps
//       and each owns an accumulator, so no MFMA multiplies by a structural zero.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, int KIND, int TAPS>
__global__ __launch_bounds__(256, 2) void conv3x3_down_mfma_kernel(ConvArgs a, WgSide src, int tiles_x, int tiles_y) {
    constexpr int DTH = 4, RECS = 9 * 66, C_OUT = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_x = smem;
    char* lds_w = smem + RECS * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int x0 = tx * 32, y0 = ty * DTH;
    const int co_base = blockIdx.y * C_OUT;
    const bf16* wsrc = reinterpret_cast<const bf16*>(a.w_bf16);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    for (int cc = 0; cc < a.c_red; cc += 32) {
        __syncthreads();
        stage_side<KIND, RECS * 4, 5>(lds_x, src, tid, [&](int rec, size_t& pix, int& ch0) {
            const int py = rec / 66, rem = rec - py * 66, par = rem >= 33, pxx = 2 * (rem - 33 * par) + par;
            const int iy = 2 * y0 + py, ix = 2 * x0 + pxx;
            pix = ((size_t)n * a.h_in + min(iy, a.h_in - 1)) * a.w_in + min(ix, a.w_in - 1);
            ch0 = cc;
            return iy < a.h_in && ix < a.w_in;
        });
        for (int t0 = 0; t0 < 9; t0 += TAPS) {
            if (t0 > 0) __syncthreads();
            stage_weights<C_OUT, TAPS>(lds_w, wsrc, t0, a.c_red, cc, tid, a.c_out, co_base);
            __syncthreads();
#pragma unroll
            for (int tl = 0; tl < TAPS; ++tl) {
                const int tap = t0 + tl;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int rec = ((2 * wave + ky) * 2 + (kx & 1)) * 33 + col + (kx >> 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int chunk = (ks << 1) | half;
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds_x + rec * 64 + ((chunk ^ ((rec >> 2) & 3)) << 4));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = nt * 32 + col;
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds_w + (tl * C_OUT + co) * 64 + ((chunk ^ ((co >> 2) & 3)) << 4));
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    const int oy = y0 + wave, ox = x0 + col;
    const bool valid = oy < a.h_out && ox < a.w_out;
    const size_t pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
    store_pixel_tiles<NT>(acc, a, pix, valid, half, co_base);
}

template <int NT, int KIND>
__global__ __launch_bounds__(256, 2) void conv3x3_up_mfma_kernel(ConvArgs a, WgSide src, int tiles_x, int tiles_y) {
    constexpr int UTH = 4, UPW = 33, RECS = (UTH + 1) * UPW, C_OUT = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_x = smem;
    char* lds_w = smem + RECS * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int j0 = tx * 32, i0 = ty * UTH;  // low-res positions; output pixel = (2i + py, 2j + px)
    const bf16* wsrc = reinterpret_cast<const bf16*>(a.w_bf16);

    f32x16 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][nt][r] = 0.f;

    for (int cc = 0; cc < a.c_red; cc += 32) {
        __syncthreads();
        stage_side<KIND, RECS * 4, 3>(lds_x, src, tid, [&](int rec, size_t& pix, int& ch0) {
            const int py = rec / UPW, pxx = rec - py * UPW;
            const int iy = i0 - 1 + py, ix = j0 - 1 + pxx;
            pix = ((size_t)n * a.h_in + min(max(iy, 0), a.h_in - 1)) * a.w_in + min(max(ix, 0), a.w_in - 1);
            ch0 = cc;
            return iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
        });
        stage_weights<C_OUT, 9>(lds_w, wsrc, 0, a.c_red, cc, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = (ks << 1) | half;
            bf16x8 xf[2][2];  // [a][b]: input (i - a, j - b)
#pragma unroll
            for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int rec = (wave + 1 - ia) * UPW + col + 1 - ib;
                    xf[ia][ib] = *reinterpret_cast<const bf16x8*>(lds_x + rec * 64 + ((chunk ^ ((rec >> 2) & 3)) << 4));
                }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int ia = 0; ia < 2 - py; ++ia)
#pragma unroll
                        for (int ib = 0; ib < 2 - px; ++ib) {
                            const int tap = (py + 2 * ia) * 3 + px + 2 * ib;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const int co = nt * 32 + col;
                                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lds_w + (tap * C_OUT + co) * 64 + ((chunk ^ ((co >> 2) & 3)) << 4));
                                acc[py * 2 + px][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[ia][ib], acc[py * 2 + px][nt], 0, 0, 0);
                            }
                        }
        }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int oy = 2 * (i0 + wave) + py, ox = 2 * (j0 + col) + px;
            const bool valid = oy < a.h_out && ox < a.w_out;
            const size_t pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
            store_pixel_tiles<NT>(acc[py * 2 + px], a, pix, valid, half);
        }
}

WgSide side_of(const ConvArgs& a) { return side_of_src(a.src, a.h_in, a.w_in, a.c_red); }

// ---------------------------------------------------------------------------------------------------------------
// stem_mfma: the 5x5 stem (u8 image, 1 or 3 channels -> 32) on the matrix cores.
// K is laid out as 5 filter rows x 32, a filter row being 8 pixels x 4 channels (RGB0): positions beyond 5 pixels / 3
// channels carry ZERO weights, so whatever finite pixel data sits there does not matter.  The image patch is kept in
// LDS as 8-byte RGB0 bf16 pixels (u8/256 is exact in bf16), which makes a lane's B operand for (filter row, k-step)
// 16 contiguous, 8-byte aligned bytes: two ds_read_b64.  The padded weights (A operand, 10 fragments) are gathered
// once per workgroup into registers; workgroups are persistent over a strided set of 8x32 tiles.
// ---------------------------------------------------------------------------------------------------------------
// ACT: the inference form — the epilogue stores relu(bn(y)) (ConvArgs::out_scale), no statistics: 32 fewer registers and none of
// the 32 KB reduction buffer, so four workgroups instead of three share a CU.
template <int CIN, bool ACT = false>
__global__ __launch_bounds__(256, (ACT ? 4 : 2)) void stem_mfma_kernel(ConvArgs a, int tiles_x, int tiles_y, int total_tiles) {
    constexpr int SPW = 48;                      // patch pixels per row: 36 used + padding read by the zero-weight positions
    __shared__ __attribute__((aligned(16))) unsigned long long patch[12 * SPW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;

    // A fragments: W'[co = col][k = ky*32 + ks*16 + 8*half + j], k -> (kx = k'/4, ci = k'%4) within the filter row
    bf16x8 wf[5][2];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = ks * 16 + 8 * half + j;
                const int kx = kk >> 2, ci = kk & 3;
                float w = 0.f;
                if (kx < 5 && ci < CIN) w = a.w_f32[((size_t)(ky * 5 + kx) * CIN + ci) * 32 + col];
                wf[ky][ks][j] = (bf16)w;
            }

    const bool fuse_stats = !ACT && (a.stat_partials != nullptr || a.stat_acc != nullptr);   // training forward: bn statistics of the stored output, as conv3x3_ws
    float stat[1][2][16];
    u32x4 none[1][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        none[0][s2] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) stat[0][s2][e] = 0.f;
    }
    for (int i = tid; i < 12 * SPW; i += 256) patch[i] = 0ull;
    // The patch pixels of the NEXT tile are requested while the MFMAs and the epilogue of the current one run (two 12x36-pixel
    // slots per thread, kept in registers as raw bytes): a tile no longer waits a memory round trip before its first MFMA.
    // A pixel is ONE unaligned 4-byte load (its CIN bytes + bytes of the next pixel, masked off); the image's very last pixel, whose
    // fourth byte would lie outside the buffer, is read byte by byte.
    constexpr int NPX = 2;                        // ceil(12 * 36 / 256)
    unsigned praw[NPX];
    unsigned pvalid = 0;
    int ppy[NPX], ppx[NPX];                       // this thread's patch positions: the same for every tile
#pragma unroll
    for (int u = 0; u < NPX; ++u) { const int i = min(tid + 256 * u, 12 * 36 - 1); ppy[u] = i / 36; ppx[u] = i - ppy[u] * 36; }
    const size_t img_pixels = (size_t)a.src.img_h * a.src.img_w;
    auto fetch = [&](int tile_) __attribute__((always_inline)) {
        const int tx = tile_ % tiles_x, ty = (tile_ / tiles_x) % tiles_y, n = tile_ / (tiles_x * tiles_y);
        const int x0 = tx * 32, y0 = ty * 8;
        const uint8_t* img = a.src.img + (size_t)n * a.src.img_sample_stride;
        pvalid = 0;
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int iy = y0 - 2 + ppy[u], ix = x0 - 2 + ppx[u];
            const int sy = min(max(a.src.win_top(n) + iy, 0), a.src.img_h - 1), sx = min(max(a.src.win_left(n) + ix, 0), a.src.img_w - 1);
            const size_t pixel = (size_t)sy * a.src.img_w + sx;
            const uint8_t* src = img + pixel * CIN;
            unsigned v;
            if (CIN == 3 && pixel + 1 < img_pixels) {
                unsigned w;
                __builtin_memcpy(&w, src, 4);     // one unaligned dword load
                v = w & 0x00ffffffu;
            } else {
                v = src[0];
                if (CIN == 3) v |= ((unsigned)src[1] << 8) | ((unsigned)src[2] << 16);
            }
            praw[u] = v;
            pvalid |= ((iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in) ? 1u : 0u) << u;
        }
    };
    int tile = blockIdx.x;
    if (tile < total_tiles) fetch(tile);
    for (; tile < total_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * 32, y0 = ty * 8;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const int i = tid + 256 * u;
            if (i < 12 * 36) {
                unsigned long long v = 0ull;
                if ((pvalid >> u) & 1u) {
                    unsigned short c[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int ch = 0; ch < CIN; ++ch) c[ch] = __builtin_bit_cast(unsigned short, (bf16)((float)((praw[u] >> (8 * ch)) & 0xffu) * (1.0f / 256.0f)));
                    v = (unsigned long long)c[0] | ((unsigned long long)c[1] << 16) | ((unsigned long long)c[2] << 32) | ((unsigned long long)c[3] << 48);
                }
                patch[ppy[u] * SPW + ppx[u]] = v;
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total_tiles) fetch(tile + (int)gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    // pixels (row wave*2+g+ky, col + 4*ks + 2*half + {0,1}) = 8 bf16 = k positions ks*16 + 8*half + 0..7
                    const unsigned long long* p = patch + (wave * 2 + g + ky) * SPW + col + 4 * ks + 2 * half;
                    typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
                    u64x2 raw = {p[0], p[1]};
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], __builtin_bit_cast(bf16x8, raw), acc[g], 0, 0, 0);
                }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int oy = y0 + wave * 2 + g, ox = x0 + col;
            const bool valid = oy < a.h_out && ox < a.w_out;
            const size_t pix = ((size_t)n * a.h_out + (valid ? oy : 0)) * a.w_out + (valid ? ox : 0);
            f32x16 one[1] = {acc[g]};
            if constexpr (ACT) store_pixel_tiles_act<1>(one, a, pix, valid, half, 0, a.out_scale, 32);   // inference: [scale | shift][32] in global memory
            else store_pixel_tiles_rmw<1>(one, a, pix, valid, half, 0, none, false, stat, fuse_stats ? 1 : 0);
        }
    }
    if constexpr (!ACT) if (fuse_stats) {   // per-lane running sums -> one partial per workgroup, [channel][sum | sum of squares][workgroup] (as conv3x3_ws)
        __shared__ float red[256 * 32];
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[tid * 32 + s2 * 16 + e] = stat[0][s2][e];
        __syncthreads();
        const int t = tid >> 2, part = tid & 3;   // 64 sums (32 channels x 2), four threads each
        const int ch = t >> 1, which = t & 1;
        const int s2 = (ch >> 4) & 1, hf = (ch >> 3) & 1, j = ch & 7;
        double sum = 0.0;
#pragma unroll 4
        for (int e = part; e < 128; e += 4) {
            const int w = e >> 5, c = e & 31;
            sum += (double)red[(w * 64 + hf * 32 + c) * 32 + s2 * 16 + which * 8 + j];
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (part == 0) {
            if (a.stat_acc) bnacc_add(a.stat_acc, BNACC_SUM_Y + which, 32, ch, sum);
            else a.stat_partials[((size_t)ch * 2 + which) * gridDim.x + blockIdx.x] = sum;
        }
    }
}

bool stem_mfma_ok(const ConvArgs& a) {
    return a.src.kind == SRC_IMAGE && a.k == 5 && a.stride == 1 && a.pad == 2 && a.gather == 0 && a.c_out == 32 && (a.c_red == 1 || a.c_red == 3) &&
           a.h_in == a.h_out && a.w_in == a.w_out && !a.bias && !a.out_nchw && a.out_dtype == DT_BF16 && a.w_f32;
}
int stem_mfma_blocks(const ConvArgs& a) {
    const int total = ((a.w_out + 31) / 32) * ((a.h_out + 7) / 8) * a.n;
    static const int env = getenv("ANH_STEM_BLOCKS") ? atoi(getenv("ANH_STEM_BLOCKS")) : 0;
    // ONE round of resident workgroups: four per CU for the inference form, three for the training form (152 VGPRs)
    return std::min(total, env > 0 ? env : a.out_scale ? 1024 : 768);
}
void launch_stem_mfma(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.w_out + 31) / 32, tiles_y = (a.h_out + 7) / 8;
    const int total = tiles_x * tiles_y * a.n;
    const dim3 grid((unsigned)stem_mfma_blocks(a)), block(256);
    ANH_REQUIRE(!a.out_scale || (a.out_shift == a.out_scale + 32 && !a.stat_partials), "stem_mfma: the activation-storing form takes [scale | shift] as one table");
    if (a.out_scale) {
        if (a.c_red == 3) hipLaunchKernelGGL((stem_mfma_kernel<3, true>), grid, block, 0, s, a, tiles_x, tiles_y, total);
        else hipLaunchKernelGGL((stem_mfma_kernel<1, true>), grid, block, 0, s, a, tiles_x, tiles_y, total);
    } else if (a.c_red == 3) hipLaunchKernelGGL((stem_mfma_kernel<3>), grid, block, 0, s, a, tiles_x, tiles_y, total);
    else hipLaunchKernelGGL((stem_mfma_kernel<1>), grid, block, 0, s, a, tiles_x, tiles_y, total);
    HIP_CHECK(hipGetLastError());
}



template <int NT, int TAPS>
void launch_down(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.w_out + 31) / 32, tiles_y = (a.h_out + 3) / 4;
    const dim3 grid((unsigned)(tiles_x * tiles_y * a.n), (unsigned)(a.c_out / (NT * 32))), block(256);
    const size_t lds = (size_t)9 * 66 * 64 + (size_t)TAPS * NT * 32 * 64;
    const WgSide src = side_of(a);
    switch (a.src.kind) {
        case SRC_RAW: hipLaunchKernelGGL((conv3x3_down_mfma_kernel<NT, SRC_RAW, TAPS>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
        case SRC_ACT: hipLaunchKernelGGL((conv3x3_down_mfma_kernel<NT, SRC_ACT, TAPS>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL((conv3x3_down_mfma_kernel<NT, SRC_ACT2, TAPS>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
    }
    HIP_CHECK(hipGetLastError());
}

template <int NT>
void launch_up(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.w_in + 1 + 31) / 32, tiles_y = (a.h_in + 1 + 3) / 4;
    const dim3 grid((unsigned)(tiles_x * tiles_y * a.n)), block(256);
    const size_t lds = (size_t)5 * 33 * 64 + (size_t)9 * NT * 32 * 64;
    const WgSide src = side_of(a);
    switch (a.src.kind) {
        case SRC_RAW: hipLaunchKernelGGL((conv3x3_up_mfma_kernel<NT, SRC_RAW>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
        case SRC_ACT: hipLaunchKernelGGL((conv3x3_up_mfma_kernel<NT, SRC_ACT>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL((conv3x3_up_mfma_kernel<NT, SRC_ACT2>), grid, block, lds, s, a, src, tiles_x, tiles_y); break;
    }
    HIP_CHECK(hipGetLastError());
}

}  // namespace

namespace { bool ws_form_ok(const ConvArgs& a); }

bool mfma_conv_supported(const ConvArgs& a) {
    if (stem_mfma_ok(a)) return true;
    if (a.k != 3 || !a.w_bf16) return false;
    if (a.src.kind == SRC_IMAGE || a.src.dtype != DT_BF16 || a.out_dtype != DT_BF16 || a.out_nchw || a.bias) return false;
    if ((a.src.kind == SRC_SUM2 || a.out_scale) && !ws_form_ok(a)) return false;   // inference forms: persistent kernels only
    if (a.c_red % 32 != 0) return false;
    // output channels: 32, or any multiple of 64 up to 256 (workgroup groups of 64 channels: levels = 3 reaches 256)
    const bool c_ok = a.c_out == 32 || a.c_out == 64 || a.c_out == 128 || a.c_out == 256;
    if (a.stride == 1 && a.pad == 1)
        return c_ok && a.h_in == a.h_out && a.w_in == a.w_out;
    if (a.stride == 2 && a.pad == 0 && a.gather == 0)
        return c_ok && a.h_in >= 3 && a.w_in >= 3 && a.h_out == (a.h_in - 3) / 2 + 1 && a.w_out == (a.w_in - 3) / 2 + 1;
    if (a.stride == 2 && a.pad == 0 && a.gather == 1)   // the classic up kernel has no channel groups: 128 only on the persistent kernel
        return (a.c_out == 32 || a.c_out == 64 || (a.c_out == 128 && ws_form_ok(a))) && a.h_out == 2 * a.h_in + 1 && a.w_out == 2 * a.w_in + 1;
    return false;
}

namespace {
// which kernel a supported 3x3 layer takes, and its launch geometry
struct ConvPlan { int geo /*0 s1, 1 down, 2 up*/, nt, tiles_x, tiles_y, flip, form /*0 classic one-tile kernels, 2 warp-specialised*/, grid_x, groups; };

int ws_target_wgs() { static const int t = getenv("ANH_WS_WGS") ? atoi(getenv("ANH_WS_WGS")) : 256; return t; }  // 1 per CU

ConvPlan conv_plan(const ConvArgs& a) {
    static const int wsm = getenv("ANH_CONV_WS") ? atoi(getenv("ANH_CONV_WS")) : 7;  // bit 0: stride 1, bit 1: down, bit 2: up (0 = the classic one-tile kernels)
    // the persistent kernels index within one image with 32-bit element offsets
    // (and fetch through buffer descriptors of one image each: its bytes, and the padding offset 0xFFFFF000 + a slab's offset, stay below 2^32)
    const bool small_plane = (int64_t)a.h_in * a.w_in * a.c_red * 2 < 0xFFFFF000ll;
    ConvPlan p{};
    p.geo = a.stride == 1 ? 0 : (a.gather == 0 ? 1 : 2);
    p.nt = a.c_out == 32 ? 1 : 2;  // 64, or 128 as two workgroup groups of 64 output channels
    p.flip = p.geo == 0 ? a.gather : 0;
    if (p.geo == 0) { p.tiles_x = (a.w_out + TW - 1) / TW; p.tiles_y = (a.h_out + TH - 1) / TH; }
    else if (p.geo == 1) { p.tiles_x = (a.w_out + 31) / 32; p.tiles_y = (a.h_out + 3) / 4; }
    else { p.tiles_x = (a.w_in + 1 + 31) / 32; p.tiles_y = (a.h_in + 1 + 3) / 4; }
    const int bit = 1 << p.geo;
    p.form = (small_plane && (wsm & bit)) ? 2 : 0;
    p.groups = a.c_out / (p.nt * 32);
    const int n_tiles = p.tiles_x * p.tiles_y * a.n;
    p.grid_x = std::max(1, std::min(n_tiles, ws_target_wgs() / p.groups));
    return p;
}
}  // namespace

namespace {
bool ws_form_ok(const ConvArgs& a) { return a.k == 3 && conv_plan(a).form == 2; }
}  // namespace

bool conv_stores_activation(const ConvArgs& a) {
    if (a.out_dtype != DT_BF16 || !mfma_conv_supported(a)) return false;
    if (stem_mfma_ok(a)) return a.c_out == 32;
    return ws_form_ok(a) && (a.src.kind == SRC_RAW || a.src.kind == SRC_SUM2);
}

int conv_fused_stat_blocks(const ConvArgs& a) {
    if (!mfma_conv_supported(a) || (int64_t)a.n * a.h_out * a.w_out == 0) return 0;
    static const int on = getenv("ANH_FUSE_BN_STATS") ? atoi(getenv("ANH_FUSE_BN_STATS")) : 1;
    if (stem_mfma_ok(a)) return on && !a.out_accumulate && !a.out2 ? stem_mfma_blocks(a) : 0;
    const ConvPlan p = conv_plan(a);
    const int acc = p.geo == 0 ? 2 : p.geo == 1 ? 1 : 4;
    if (!on || p.form != 2 || a.out_accumulate || a.out2) return 0;
    if (acc * p.nt > 4 && a.src.kind != SRC_RAW && a.src.kind != SRC_ACT && a.src.kind != SRC_ACT2) return 0;   // (the forward-only form)
    return p.grid_x;
}

// the persistent kernels fold Src::a_tab / b_tab in their table prologue; the classic one-tile kernels read arrays only
bool conv_folds_bn_tables(const ConvArgs& a) { return mfma_conv_supported(a) && !stem_mfma_ok(a) && ws_form_ok(a); }

bool conv_head_in_epilogue_ok(const ConvArgs& a) {
    static const bool on = !(getenv("ANH_HEAD_IN_EPILOGUE") && atoi(getenv("ANH_HEAD_IN_EPILOGUE")) == 0);
    if (!on || !mfma_conv_supported(a) || stem_mfma_ok(a) || !ws_form_ok(a)) return false;
    return a.k == 3 && a.stride == 1 && a.c_out == 32 && a.out_scale && a.out_dtype == DT_BF16 && a.head_k >= 1 && a.head_k <= 4 &&
           (a.src.kind == SRC_RAW || a.src.kind == SRC_SUM2);
}

int conv_fused_bnred_blocks(const ConvArgs& a) {
    if (!mfma_conv_supported(a) || stem_mfma_ok(a) || (int64_t)a.n * a.h_out * a.w_out == 0) return 0;
    static const int on = getenv("ANH_FUSE_BN_BWD_REDUCE") ? atoi(getenv("ANH_FUSE_BN_BWD_REDUCE")) : 1;
    const ConvPlan p = conv_plan(a);
    const int acc = p.geo == 0 ? 2 : p.geo == 1 ? 1 : 4;
    if (!on || p.form != 2) return 0;
    // more than four accumulator tiles per consumer wave leave no registers for the sums there: only the producer-side form keeps them
    // (64 output channels of the up geometry: 128->64 backward-data 48 -> 77 us, and the layer's own reduce pass over (da, y), 42 us,
    // disappears; two workgroup groups of 32 channels each in that form cost the same 77 us)
    static const int wide_ps = getenv("ANH_WS_WIDE_PS") ? atoi(getenv("ANH_WS_WIDE_PS")) : 1;
    if (acc * p.nt > 4 && !(wide_ps && ws_layout(a, p.geo == 0 ? GeoS1::RECS : p.geo == 1 ? GeoDown::RECS : GeoUp::RECS, acc, p.nt, true).ps)) return 0;
    return p.grid_x;
}

void launch_conv_mfma(const ConvArgs& a, hipStream_t s) {
    if (!mfma_conv_supported(a)) fail(ANH_ERR_INTERNAL, "conv_mfma: unsupported shape");
    if ((int64_t)a.n * a.h_out * a.w_out == 0) return;
    if (stem_mfma_ok(a)) { launch_stem_mfma(a, s); return; }
    const ConvPlan p = conv_plan(a);
    ANH_REQUIRE(!(a.stat_partials || a.stat_acc) || conv_fused_stat_blocks(a) > 0, "conv_mfma: this layer's kernel does not fuse bn statistics");
    ANH_REQUIRE(!a.bnred_acc || (conv_fused_bnred_blocks(a) > 0 && !a.stat_partials && !a.stat_acc), "conv_mfma: this layer's kernel does not fuse the bn backward reduction");
    ANH_REQUIRE(!a.src.a_tab.acc || conv_folds_bn_tables(a), "conv_mfma: this layer's kernel does not fold bn accumulator tables");
    ANH_REQUIRE(!a.bnred_partials || (conv_fused_bnred_blocks(a) > 0 && !a.stat_partials), "conv_mfma: this layer's kernel does not fuse the bn backward reduction");
    if (p.geo == 0) {
        if (p.form == 2) {
            if (p.nt == 1) launch_ws<GeoS1, 1>(a, p.tiles_x, p.tiles_y, p.flip, s);
            else launch_ws<GeoS1, 2>(a, p.tiles_x, p.tiles_y, p.flip, s);
        }
        else if (a.c_out == 32) launch_s1<1, 9>(a, s);
        else launch_s1<2, 9>(a, s);
    } else if (p.geo == 1) {
        if (p.form == 2) { if (p.nt == 1) launch_ws<GeoDown, 1>(a, p.tiles_x, p.tiles_y, 0, s); else launch_ws<GeoDown, 2>(a, p.tiles_x, p.tiles_y, 0, s); }
        else if (a.c_out == 32) launch_down<1, 9>(a, s);
        else if (a.c_out == 128) launch_down<4, 3>(a, s);
        else launch_down<2, 9>(a, s);   // 64, or 256 as four groups of 64
    } else {
        if (p.form == 2) { if (p.nt == 1) launch_ws<GeoUp, 1>(a, p.tiles_x, p.tiles_y, 0, s); else launch_ws<GeoUp, 2>(a, p.tiles_x, p.tiles_y, 0, s); }
        else if (a.c_out == 32) launch_up<1>(a, s);
        else launch_up<2>(a, s);
    }
}

bool wgrad_accepts_bnbwd(const WgradArgs& a, DType mode) {
    static const int on = getenv("ANH_FUSE_STEM_BN_APPLY") ? atoi(getenv("ANH_FUSE_STEM_BN_APPLY")) : 1;
    return on && mode == DT_BF16 && stem_wgrad_mfma_ok(a) && a.c_out == 32;
}

bool mfma_wgrad_supported(const WgradArgs& a) {
    if (a.dy_y && !stem_wgrad_mfma_ok(a)) return false;   // only the stem kernel computes dy on the fly
    if (stem_wgrad_mfma_ok(a)) return true;
    if (a.k != 3 || a.src.kind == SRC_IMAGE || a.src.dtype != DT_BF16 || a.dy_dtype != DT_BF16) return false;
    const bool s1 = a.stride == 1 && a.pad == 1 && a.gather == 0 && a.h_in == a.h_out && a.w_in == a.w_out;
    const bool s2 = a.stride == 2 && a.pad == 0;
    if (!s1 && !s2) return false;
    const int c_tile = a.gather == 1 ? a.c_in : a.c_out, c_patch = a.gather == 1 ? a.c_out : a.c_in;
    const int64_t plane_in = (int64_t)a.h_in * a.w_in * a.c_in, plane_out = (int64_t)a.h_out * a.w_out * a.c_out;
    if (plane_in * 2 >= 0xFFFFF000ll || plane_out * 2 >= 0xFFFFF000ll) return false;  // the kernel addresses within one image through a buffer descriptor (32-bit byte offsets; 0xFFFFF000 marks padding)
    static const int ws_on = getenv("ANH_WGRAD_WS") ? atoi(getenv("ANH_WGRAD_WS")) : 1;
    return c_patch % 32 == 0 && (c_tile == 32 || c_tile == 64 || c_tile == 128 || (c_tile == 256 && ws_on));
}

int64_t wgrad_mfma_scratch_floats(const WgradArgs& a) {
    if (stem_wgrad_mfma_ok(a)) return (int64_t)stem_wgrad_mfma_blocks(a) * 25 * a.c_in * 32;
    return (int64_t)wgrad_plan_mfma(a).splits * 9 * a.c_in * a.c_out;
}

void launch_wgrad_mfma(const WgradArgs& a, hipStream_t s) {
    if (!mfma_wgrad_supported(a)) fail(ANH_ERR_INTERNAL, "wgrad_mfma: unsupported shape");
    ANH_REQUIRE(wgrad_mfma_scratch_floats(a) <= a.partials_capacity, "wgrad scratch too small");
    if ((int64_t)a.n * a.h_out * a.w_out == 0) return;
    if (stem_wgrad_mfma_ok(a)) { launch_wgrad_stem_mfma(a, s); return; }
    launch_wgrad_any(a, s);
}

}  // namespace anh
