// annonet_oracle.cpp — CPU restatement of the annonet hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product (annonet_amd/csrc, libannonet_hip.so) never links or calls it.
//
// PARITY PINNING STATUS
//   * set_weights / random_rect_containing_point: pinned by the reference's own
//     known-answer tests (test/annonet_test.cpp:54-130), re-hosted in tests/.
//   * tile geometry, clamp-to-edge crop, blend, argmax, detection filter: restated line
//     by line from first-party reference source that is present
//     (annonet_infer.cpp:26-240, annonet.h:74-120); no reference test pins them.
//   * net forward/backward, loss, SGD, BN, tiler: "parity unpinned".  The reference keeps
//     this arithmetic in un-vendored, un-pinned submodules (dlib, dlib-dnn-pimpl-wrapper,
//     tiling; .gitmodules:1-27, all empty in the snapshot).  The restatement follows
//     dlib's published layer semantics (con / cont / bn_con / affine / relu /
//     loss_multiclass_log_per_pixel_weighted / sgd) as design intent and is cross-checked
//     against PyTorch-CPU as an independent second opinion (tests/test_oracle_vs_torch.py).
//
// Numerical conventions (chosen so that the fp32 GPU parity mode can be bit-exact):
//   * activations fp32, NHWC internally; NCHW at the boundary (annonet_infer.cpp:26-30).
//   * every convolution output element is ONE k-ordered fmaf chain starting from +0:
//     taps in (ky,kx) row-major order, input channels innermost; out-of-image taps are
//     skipped; bias (if any) is added after the chain.
//   * reductions over pixels (BN statistics, loss, dgamma/dbeta, filter gradients) are
//     accumulated in double.
//   * compiled with -ffp-contract=off: only explicit fmaf/fma calls fuse.
//
// Build: see oracle/Makefile (g++ -O3 -mavx2 -mfma -fopenmp -ffp-contract=off).

#include <algorithm>
#include <cmath>
#include <immintrin.h>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

constexpr uint16_t kIgnore = 65535;  // dlib::loss_multiclass_log_per_pixel_::label_to_ignore (annonet.cpp:25)
constexpr float kBnEps = 1e-4f;      // dlib DEFAULT_BATCH_NORM_EPS [UPSTREAM-UNVERIFIED]

thread_local std::string g_err;

// bf16 emulation (round-to-nearest-even, as v_cvt_pk_bf16_f32): used to restate the ANH_BF16 storage points
//  so that the throughput mode can be held to a tight bar too.
inline float bf16r(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v;  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    std::memcpy(&v, &u, 4);
    return v;
}
inline void round_tensor(std::vector<float>& d) { for (float& v : d) v = bf16r(v); }

// ------------------------------------------------------------------------------------------
// Net specification.  Stands in for dlib-dnn-pimpl-wrapper/NetStructure.h [ABSENT]
// (annonet_train.h:16, annonet_train_cuda.vcxproj:247).  DESIGN.md §2 states the same list.
// ------------------------------------------------------------------------------------------
struct Layer {
    int type;  // 0 = con (cross-correlation), 1 = cont (transposed convolution)
    int k, stride, pad;
    int cin, cout;
    int in_a, in_b;  // producing layer index; -1 = the u8 image; in_b = -2 when there is no skip
    int has_bn, has_bias;
    int64_t w_off, b_off, g_off, beta_off;  // offsets into the canonical parameter blob
    int64_t rs_off;                         // offset into the running-stats blob (mean[C], var[C])
};

struct Spec {
    int levels, in_ch, classes;
    std::vector<Layer> layers;
    int64_t n_params = 0, n_running = 0;
};

int width_of(int level, double scaler, int min_filters) {
    static const int base[4] = {32, 64, 128, 256};
    return std::max(min_filters, (int)std::lround(scaler * base[level]));
}

Spec build_spec(int levels, int in_ch, int classes, double scaler, int min_filters) {
    if (levels < 0 || levels > 3) throw std::runtime_error("level count must be 0..3");
    if (in_ch != 1 && in_ch != 3) throw std::runtime_error("input channels must be 1 or 3");
    if (classes < 1) throw std::runtime_error("class count must be >= 1");
    Spec s;
    s.levels = levels; s.in_ch = in_ch; s.classes = classes;
    auto add = [&](int type, int k, int stride, int pad, int cin, int cout, int in_a, int in_b, int bn, int bias) {
        Layer L{};
        L.type = type; L.k = k; L.stride = stride; L.pad = pad; L.cin = cin; L.cout = cout;
        L.in_a = in_a; L.in_b = in_b; L.has_bn = bn; L.has_bias = bias;
        L.w_off = s.n_params; s.n_params += (int64_t)k * k * cin * cout;
        L.b_off = -1; L.g_off = -1; L.beta_off = -1; L.rs_off = -1;
        if (bias) { L.b_off = s.n_params; s.n_params += cout; }
        if (bn) {
            L.g_off = s.n_params; s.n_params += cout;
            L.beta_off = s.n_params; s.n_params += cout;
            L.rs_off = s.n_running; s.n_running += 2 * cout;
        }
        s.layers.push_back(L);
        return (int)s.layers.size() - 1;
    };
    std::vector<int> C(levels + 1);
    for (int l = 0; l <= levels; ++l) C[l] = width_of(l, scaler, min_filters);
    std::vector<int> enc(levels + 1);
    enc[0] = add(0, 5, 1, 2, in_ch, C[0], -1, -2, 1, 0);  // stem
    for (int l = 1; l <= levels; ++l) {
        int d = add(0, 3, 2, 0, C[l - 1], C[l], enc[l - 1], -2, 1, 0);  // down_l
        enc[l] = add(0, 3, 1, 1, C[l], C[l], d, -2, 1, 0);              // enc_l
    }
    int cur = enc[levels];
    for (int l = levels; l >= 1; --l) {
        int u = add(1, 3, 2, 0, C[l], C[l - 1], cur, -2, 1, 0);          // up_l
        cur = add(0, 3, 1, 1, C[l - 1], C[l - 1], u, enc[l - 1], 1, 0);  // dec_{l-1}: input = up + skip
    }
    add(0, 1, 1, 0, C[0], classes, cur, -2, 0, 1);  // head
    return s;
}

int out_dim(const Layer& L, int in) {
    if (L.type == 0) return in + 2 * L.pad < L.k ? 0 : (in + 2 * L.pad - L.k) / L.stride + 1;
    return L.stride * (in - 1) + L.k - 2 * L.pad;
}

// smallest valid input side >= n: every stride-2 3x3 p0 conv must divide exactly and the
// transposed conv must restore the size: d = 2^L * m + 2^L - 1, m >= 1.
int recommended_dim(int levels, int n) {
    const int q = 1 << levels;
    int m = (n - (q - 1) + q - 1) / q;
    if (n <= q - 1) m = 1;
    if (m < 1) m = 1;
    return q * m + q - 1;
}

int required_dim(const Spec& s) {  // receptive field side of one output pixel
    int rf = 1, jump = 1;
    for (const Layer& L : s.layers) {
        if (L.type == 0) { rf += (L.k - 1) * jump; jump *= L.stride; }
        else { rf += ((L.k + L.stride - 1) / L.stride - 1) * jump; jump /= L.stride; }
    }
    return rf;
}

// ------------------------------------------------------------------------------------------
// Tensors and layer arithmetic
// ------------------------------------------------------------------------------------------
struct Tensor {  // NHWC fp32
    int n = 0, h = 0, w = 0, c = 0;
    std::vector<float> d;
    // all zeros afterwards; a tensor that keeps its size (scratch reused from step to step) is cleared by the thread team instead of
    // being re-created by one thread (which, once the convs run as GEMMs, was a third of a training step)
    void resize(int n_, int h_, int w_, int c_) {
        n = n_; h = h_; w = w_; c = c_;
        const size_t total = (size_t)n * h * w * c;
        if (d.size() != total) { d.assign(total, 0.f); return; }
        float* p = d.data();
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < total; ++i) p[i] = 0.f;
    }
    void copy_from(const Tensor& o) {   // *this = o, element copies spread over the thread team
        n = o.n; h = o.h; w = o.w; c = o.c;
        if (d.size() != o.d.size()) d.resize(o.d.size());
        float* p = d.data(); const float* q = o.d.data();
        const size_t total = o.d.size();
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < total; ++i) p[i] = q[i];
    }
    size_t pixels() const { return (size_t)n * h * w; }
    float* px(int in, int y, int x) { return d.data() + (((size_t)in * h + y) * w + x) * c; }
    const float* px(int in, int y, int x) const { return d.data() + (((size_t)in * h + y) * w + x) * c; }
};

// canonical filter layouts: con = [cout][cin][ky][kx] (dlib con_ filters,
// torch Conv2d); cont = [cin][cout][ky][kx] (dlib cont_, torch ConvTranspose2d).
// Working layout here: [tap][cin][cout].
std::vector<float> to_tap_major(const Layer& L, const float* canon) {
    const int k = L.k, ci = L.cin, co = L.cout;
    std::vector<float> w((size_t)k * k * ci * co);
    for (int t = 0; t < k * k; ++t)
        for (int i = 0; i < ci; ++i)
            for (int o = 0; o < co; ++o) {
                size_t src = L.type == 0 ? ((size_t)o * ci + i) * k * k + t : ((size_t)i * co + o) * k * k + t;
                w[((size_t)t * ci + i) * co + o] = canon[src];
            }
    return w;
}
void from_tap_major_add(const Layer& L, const double* wt, float* canon_grad) {
    const int k = L.k, ci = L.cin, co = L.cout;
    for (int t = 0; t < k * k; ++t)
        for (int i = 0; i < ci; ++i)
            for (int o = 0; o < co; ++o) {
                size_t dst = L.type == 0 ? ((size_t)o * ci + i) * k * k + t : ((size_t)i * co + o) * k * k + t;
                canon_grad[dst] = (float)wt[((size_t)t * ci + i) * co + o];
            }
}

// forward of con / cont.  x: input activations; y: raw output (before bias/bn).
void conv_forward(const Layer& L, const std::vector<float>& wt, const Tensor& x, Tensor& y) {
    const int k = L.k, ci = L.cin, co = L.cout, s = L.stride, p = L.pad;
    y.resize(x.n, out_dim(L, x.h), out_dim(L, x.w), co);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < y.n; ++n)
        for (int oy = 0; oy < y.h; ++oy) {
            std::vector<float> acc(co);
            for (int ox = 0; ox < y.w; ++ox) {
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        int iy, ix;
                        if (L.type == 0) { iy = oy * s + ky - p; ix = ox * s + kx - p; }
                        else {
                            int ty = oy + p - ky, tx = ox + p - kx;
                            if (ty < 0 || tx < 0 || ty % s || tx % s) continue;
                            iy = ty / s; ix = tx / s;
                        }
                        if (iy < 0 || iy >= x.h || ix < 0 || ix >= x.w) continue;
                        const float* xp = x.px(n, iy, ix);
                        const float* wp = wt.data() + (size_t)(ky * k + kx) * ci * co;
                        for (int i = 0; i < ci; ++i) {
                            const float a = xp[i];
                            const float* wr = wp + (size_t)i * co;
                            for (int o = 0; o < co; ++o) acc[o] = fmaf(a, wr[o], acc[o]);
                        }
                    }
                std::memcpy(y.px(n, oy, ox), acc.data(), sizeof(float) * co);
            }
        }
}

// dx += (transpose of conv_forward)(dy)
void conv_backward_data(const Layer& L, const std::vector<float>& wt, const Tensor& dy, Tensor& dx) {
    const int k = L.k, ci = L.cin, co = L.cout, s = L.stride, p = L.pad;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < dx.n; ++n)
        for (int iy = 0; iy < dx.h; ++iy) {
            std::vector<float> acc(ci);
            for (int ix = 0; ix < dx.w; ++ix) {
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        int oy, ox;
                        if (L.type == 0) {
                            int ty = iy + p - ky, tx = ix + p - kx;
                            if (ty < 0 || tx < 0 || ty % s || tx % s) continue;
                            oy = ty / s; ox = tx / s;
                        } else { oy = iy * s + ky - p; ox = ix * s + kx - p; }
                        if (oy < 0 || oy >= dy.h || ox < 0 || ox >= dy.w) continue;
                        const float* gp = dy.px(n, oy, ox);
                        const float* wp = wt.data() + (size_t)(ky * k + kx) * ci * co;
                        for (int i = 0; i < ci; ++i) {
                            const float* wr = wp + (size_t)i * co;
                            float a = acc[i];
                            for (int o = 0; o < co; ++o) a = fmaf(gp[o], wr[o], a);
                            acc[i] = a;
                        }
                    }
                float* d = dx.px(n, iy, ix);
                for (int i = 0; i < ci; ++i) d[i] += acc[i];
            }
        }
}

// filter gradient, tap-major double accumulators: dw[t][ci][co] = sum_p x[p@t][ci] * dy[p][co]
void conv_backward_filter(const Layer& L, const Tensor& x, const Tensor& dy, std::vector<double>& dw) {
    const int k = L.k, ci = L.cin, co = L.cout, s = L.stride, p = L.pad;
    const size_t nw = (size_t)k * k * ci * co;
    dw.assign(nw, 0.0);
    const int rows = dy.n * dy.h;
#pragma omp parallel
    {
        std::vector<double> local(nw, 0.0);
        std::vector<float> rowacc(nw);
#pragma omp for schedule(static)
        for (int r = 0; r < rows; ++r) {
            const int n = r / dy.h, oy = r % dy.h;
            std::fill(rowacc.begin(), rowacc.end(), 0.f);
            for (int ox = 0; ox < dy.w; ++ox) {
                const float* gp = dy.px(n, oy, ox);
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        int iy, ix;
                        if (L.type == 0) { iy = oy * s + ky - p; ix = ox * s + kx - p; }
                        else {
                            int ty = oy + p - ky, tx = ox + p - kx;
                            if (ty < 0 || tx < 0 || ty % s || tx % s) continue;
                            iy = ty / s; ix = tx / s;
                        }
                        if (iy < 0 || iy >= x.h || ix < 0 || ix >= x.w) continue;
                        const float* xp = x.px(n, iy, ix);
                        float* wa = rowacc.data() + (size_t)(ky * k + kx) * ci * co;
                        for (int i = 0; i < ci; ++i) {
                            const float a = xp[i];
                            float* wr = wa + (size_t)i * co;
                            for (int o = 0; o < co; ++o) wr[o] = fmaf(a, gp[o], wr[o]);
                        }
                    }
            }
            for (size_t i = 0; i < nw; ++i) local[i] += rowacc[i];
        }
#pragma omp critical
        for (size_t i = 0; i < nw; ++i) dw[i] += local[i];
    }
}

// ------------------------------------------------------------------------------------------
// im2col + blocked SGEMM (conv_algo = 1): the CPU design SURVEY.md §8d names as the baseline to time —
// dlib's CPU path lowers con / cont to im2col (per sample) + BLAS sgemm (cpu_dlib.cpp + BLAS;
// /root/reference/annonet_train_cpu.vcxproj:93,113,230 list those files).  No BLAS exists in this image, so the
// SGEMM is written here: a 6 x 16 AVX2 register tile, K unblocked (the operands of one strip fit the core's L2).
// Every layer is handled through its "con geometry": a big tensor B (cb channels), a small tensor S (cs channels),
// filters W[(tap, cb)][cs]:
//   gather   S  = im2col(B) . W             con forward,        cont backward-data
//   scatter  B += col2im(S . W^T)           con backward-data,  cont forward
//   filter   dW = im2col(B)^T . S           both filter gradients
// The direct loops above stay the parity oracle (their summation order is what the GPU parity mode reproduces);
// this path is checked against them (tests/test_oracle_gemm.py) and is what bench.py times as cpu_baseline.
// ------------------------------------------------------------------------------------------
struct ConGeo { int k, s, p, cb, cs; };
// per-thread scratch that lives as long as the thread: a 0.25-1 MB vector created inside every parallel region is an mmap + page faults
// + munmap per thread and call, and the unmaps' TLB shootdowns serialise a 16-thread team (measured on the GPU box's host: 16 threads
// no faster than one)
static std::vector<float>& tl_floats(int which, size_t n) {
    static thread_local std::vector<float> buf[2];
    if (buf[which].size() < n) buf[which].resize(n);
    return buf[which];
}
static std::vector<double>& tl_doubles(size_t n) {
    static thread_local std::vector<double> buf;
    if (buf.size() < n) buf.resize(n);
    return buf;
}

// C[M x N] (+)= A . B with A(i, k) = a[i * ars + k * aks], B and C row-major.
template <int MR>
static inline void sgemm_tile16(int K, const float* a, ptrdiff_t ars, ptrdiff_t aks, const float* b, int ldb, float* c, int ldc, bool accumulate) {
    __m256 acc[MR][2];
    for (int i = 0; i < MR; ++i) {
        acc[i][0] = accumulate ? _mm256_loadu_ps(c + (size_t)i * ldc) : _mm256_setzero_ps();
        acc[i][1] = accumulate ? _mm256_loadu_ps(c + (size_t)i * ldc + 8) : _mm256_setzero_ps();
    }
    for (int k = 0; k < K; ++k) {
        const __m256 b0 = _mm256_loadu_ps(b + (size_t)k * ldb), b1 = _mm256_loadu_ps(b + (size_t)k * ldb + 8);
        for (int i = 0; i < MR; ++i) {
            const __m256 av = _mm256_broadcast_ss(a + i * ars + k * aks);
            acc[i][0] = _mm256_fmadd_ps(av, b0, acc[i][0]);
            acc[i][1] = _mm256_fmadd_ps(av, b1, acc[i][1]);
        }
    }
    for (int i = 0; i < MR; ++i) { _mm256_storeu_ps(c + (size_t)i * ldc, acc[i][0]); _mm256_storeu_ps(c + (size_t)i * ldc + 8, acc[i][1]); }
}
template <int MR>
static inline void sgemm_tile8(int K, const float* a, ptrdiff_t ars, ptrdiff_t aks, const float* b, int ldb, float* c, int ldc, bool accumulate) {
    __m256 acc[MR];
    for (int i = 0; i < MR; ++i) acc[i] = accumulate ? _mm256_loadu_ps(c + (size_t)i * ldc) : _mm256_setzero_ps();
    for (int k = 0; k < K; ++k) {
        const __m256 b0 = _mm256_loadu_ps(b + (size_t)k * ldb);
        for (int i = 0; i < MR; ++i) acc[i] = _mm256_fmadd_ps(_mm256_broadcast_ss(a + i * ars + k * aks), b0, acc[i]);
    }
    for (int i = 0; i < MR; ++i) _mm256_storeu_ps(c + (size_t)i * ldc, acc[i]);
}
static void sgemm_block(int M, int N, int K, const float* a, ptrdiff_t ars, ptrdiff_t aks, const float* b, int ldb, float* c, int ldc, bool accumulate) {
    for (int i0 = 0; i0 < M; i0 += 6) {
        const int mr = std::min(6, M - i0);
        const float* ai = a + (ptrdiff_t)i0 * ars;
        float* ci = c + (size_t)i0 * ldc;
        int j0 = 0;
        for (; j0 + 16 <= N; j0 += 16) {
            switch (mr) {
                case 6: sgemm_tile16<6>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 5: sgemm_tile16<5>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 4: sgemm_tile16<4>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 3: sgemm_tile16<3>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 2: sgemm_tile16<2>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                default: sgemm_tile16<1>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
            }
        }
        for (; j0 + 8 <= N; j0 += 8) {
            switch (mr) {
                case 6: sgemm_tile8<6>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 5: sgemm_tile8<5>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 4: sgemm_tile8<4>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 3: sgemm_tile8<3>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                case 2: sgemm_tile8<2>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
                default: sgemm_tile8<1>(K, ai, ars, aks, b + j0, ldb, ci + j0, ldc, accumulate); break;
            }
        }
        for (; j0 < N; ++j0)   // narrow remainders (the 1x1 head's K = 3 classes)
            for (int i = 0; i < mr; ++i) {
                float acc = accumulate ? ci[(size_t)i * ldc + j0] : 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(ai[i * ars + k * aks], b[(size_t)k * ldb + j0], acc);
                ci[(size_t)i * ldc + j0] = acc;
            }
    }
}
// K is walked in blocks of 256 so that the A and B panels of a register tile stay in L1; C carries the partial sums between blocks
// exactly (an fp32 store and reload), so every output is still ONE k-ordered fmaf chain from +0.
static void sgemm(int M, int N, int K, const float* a, ptrdiff_t ars, ptrdiff_t aks, const float* b, int ldb, float* c, int ldc, bool accumulate) {
    constexpr int KC = 256;
    for (int k0 = 0; k0 < K; k0 += KC)
        sgemm_block(M, N, std::min(KC, K - k0), a + (ptrdiff_t)k0 * aks, ars, aks, b + (size_t)k0 * ldb, ldb, c, ldc, accumulate || k0 > 0);
    if (K == 0 && !accumulate) for (int i = 0; i < M; ++i) std::memset(c + (size_t)i * ldc, 0, sizeof(float) * N);
}

// one row `sy` of the small tensor: col[sx][(tap, cb)] = B(n, sy*s + ky - p, sx*s + kx - p) or 0 outside the tensor
static void im2col_row(const ConGeo& g, const Tensor& B, int n, int sy, int ws, float* col) {
    const int kk = g.k * g.k * g.cb;
    for (int sx = 0; sx < ws; ++sx) {
        float* dst = col + (size_t)sx * kk;
        for (int ky = 0; ky < g.k; ++ky)
            for (int kx = 0; kx < g.k; ++kx, dst += g.cb) {
                const int by = sy * g.s + ky - g.p, bx = sx * g.s + kx - g.p;
                if (by < 0 || by >= B.h || bx < 0 || bx >= B.w) std::memset(dst, 0, sizeof(float) * g.cb);
                else std::memcpy(dst, B.px(n, by, bx), sizeof(float) * g.cb);
            }
    }
}
static void gemm_gather(const ConGeo& g, const float* W /*[(tap,cb)][cs]*/, const Tensor& B, Tensor& S) {
    const int kk = g.k * g.k * g.cb, rows = S.n * S.h;
#pragma omp parallel
    {
        std::vector<float>& col = tl_floats(0, (size_t)S.w * kk);
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < rows; ++r) {
            const int n = r / S.h, sy = r % S.h;
            im2col_row(g, B, n, sy, S.w, col.data());
            sgemm(S.w, g.cs, kk, col.data(), kk, 1, W, g.cs, S.px(n, sy, 0), g.cs, false);
        }
    }
}
// B += col2im(S . W^T).  Rows of S whose scatter footprints overlap run in different passes (ceil(k / s) colours).
static void gemm_scatter(const ConGeo& g, const float* WT /*[cs][(tap,cb)]*/, const Tensor& S, Tensor& B) {
    const int kk = g.k * g.k * g.cb, rows = S.n * S.h, colours = (g.k + g.s - 1) / g.s;
#pragma omp parallel
    {
        std::vector<float>& col = tl_floats(0, (size_t)S.w * kk);
        for (int colour = 0; colour < colours; ++colour) {
#pragma omp for schedule(dynamic, 1)
            for (int r = 0; r < rows; ++r) {
                const int n = r / S.h, sy = r % S.h;
                if (sy % colours != colour) continue;
                sgemm(S.w, kk, g.cs, S.px(n, sy, 0), g.cs, 1, WT, kk, col.data(), kk, false);
                for (int sx = 0; sx < S.w; ++sx) {
                    const float* src = col.data() + (size_t)sx * kk;
                    for (int ky = 0; ky < g.k; ++ky)
                        for (int kx = 0; kx < g.k; ++kx, src += g.cb) {
                            const int by = sy * g.s + ky - g.p, bx = sx * g.s + kx - g.p;
                            if (by < 0 || by >= B.h || bx < 0 || bx >= B.w) continue;
                            float* d = B.px(n, by, bx);
                            for (int c = 0; c < g.cb; ++c) d[c] += src[c];
                        }
                }
            }   // (implicit barrier: the next colour touches rows this one has finished)
        }
    }
}
// dW[(tap,cb)][cs] = sum over pixels; float within one row of S, double across rows (as the direct loop)
static void gemm_filter(const ConGeo& g, const Tensor& B, const Tensor& S, std::vector<double>& dw) {
    const int kk = g.k * g.k * g.cb, rows = S.n * S.h;
    const size_t nw = (size_t)kk * g.cs;
    dw.assign(nw, 0.0);
#pragma omp parallel
    {
        std::vector<float>& col = tl_floats(0, (size_t)S.w * kk);
        std::vector<float>& part = tl_floats(1, nw);
        std::vector<double>& local = tl_doubles(nw);
        std::fill(local.begin(), local.begin() + nw, 0.0);
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < rows; ++r) {
            const int n = r / S.h, sy = r % S.h;
            im2col_row(g, B, n, sy, S.w, col.data());
            sgemm(kk, g.cs, S.w, col.data(), 1, kk, S.px(n, sy, 0), g.cs, part.data(), g.cs, false);
            for (size_t i = 0; i < nw; ++i) local[i] += part[i];
        }
#pragma omp critical
        for (size_t i = 0; i < nw; ++i) dw[i] += local[i];
    }
}
// tap-major filters wt[tap][ci][co] of a layer -> the con-geometry operands
static ConGeo con_geo(const Layer& L) { return ConGeo{L.k, L.stride, L.pad, L.type == 0 ? L.cin : L.cout, L.type == 0 ? L.cout : L.cin}; }
static std::vector<float> con_w(const Layer& L, const std::vector<float>& wt) {   // [(tap, cb)][cs]
    if (L.type == 0) return wt;
    std::vector<float> w(wt.size());
    const int t = L.k * L.k;
    for (int tt = 0; tt < t; ++tt) for (int i = 0; i < L.cin; ++i) for (int o = 0; o < L.cout; ++o) w[((size_t)tt * L.cout + o) * L.cin + i] = wt[((size_t)tt * L.cin + i) * L.cout + o];
    return w;
}
static std::vector<float> con_wt(const Layer& L, const std::vector<float>& wt) {  // [cs][(tap, cb)]
    std::vector<float> w(wt.size());
    const int t = L.k * L.k, kk = t * (L.type == 0 ? L.cin : L.cout);
    for (int tt = 0; tt < t; ++tt) for (int i = 0; i < L.cin; ++i) for (int o = 0; o < L.cout; ++o) {
        const float v = wt[((size_t)tt * L.cin + i) * L.cout + o];
        if (L.type == 0) w[(size_t)o * kk + tt * L.cin + i] = v; else w[(size_t)i * kk + tt * L.cout + o] = v;
    }
    return w;
}
void conv_forward_gemm(const Layer& L, const std::vector<float>& wt, const Tensor& x, Tensor& y) {
    y.resize(x.n, out_dim(L, x.h), out_dim(L, x.w), L.cout);
    const ConGeo g = con_geo(L);
    if (L.type == 0) gemm_gather(g, wt.data(), x, y);
    else { const std::vector<float> w = con_wt(L, wt); gemm_scatter(g, w.data(), x, y); }
}
void conv_backward_data_gemm(const Layer& L, const std::vector<float>& wt, const Tensor& dy, Tensor& dx) {   // dx += ...
    const ConGeo g = con_geo(L);
    if (L.type == 0) { const std::vector<float> w = con_wt(L, wt); gemm_scatter(g, w.data(), dy, dx); }
    else {
        const std::vector<float> w = con_w(L, wt);
        Tensor t; t.resize(dx.n, dx.h, dx.w, dx.c);
        gemm_gather(g, w.data(), dy, t);
        for (size_t i = 0; i < dx.d.size(); ++i) dx.d[i] += t.d[i];
    }
}
void conv_backward_filter_gemm(const Layer& L, const Tensor& x, const Tensor& dy, std::vector<double>& dw) {   // dw[tap][ci][co]
    const ConGeo g = con_geo(L);
    if (L.type == 0) { gemm_filter(g, x, dy, dw); return; }
    std::vector<double> t;
    gemm_filter(g, dy, x, t);   // [(tap, co)][ci]
    dw.assign(t.size(), 0.0);
    const int taps = L.k * L.k;
    for (int tt = 0; tt < taps; ++tt) for (int i = 0; i < L.cin; ++i) for (int o = 0; o < L.cout; ++o) dw[((size_t)tt * L.cin + i) * L.cout + o] = t[((size_t)tt * L.cout + o) * L.cin + i];
}

// ------------------------------------------------------------------------------------------
// The net object
// ------------------------------------------------------------------------------------------
struct Net {
    Spec spec;
    std::vector<float> params, momentum, grads, running;  // canonical blobs
    std::vector<double> running_count;                    // per bn layer: number of updates so far
    // hyper-parameters (annonet_train_main.cpp:396-410)
    double lr = 0.1, weight_decay = 0.0005, mom = 0.9;
    unsigned long bn_window = 100;
    // bf16 storage points of the HIP path (0 = fp32): weights, conv inputs and stored gradients rounded to bf16, and
    //   training passes: the stored RAW conv outputs y (consumers re-apply bn + relu to the rounded y);
    //   inference passes (1): the stored ACTIVATIONS relu(bn(y)) — the epilogue applies the layer's folded bn + relu to the fp32
    //   accumulators and rounds once; (2): raw-output storage in inference as well (the library with ANH_INFER_POST_ACT=0)
    int emulate_bf16 = 0;
    // 0: direct convolution loops (the parity oracle); 1: im2col + blocked SGEMM with OpenMP reductions (the timed CPU baseline)
    int conv_algo = 0;
    // scratch
    std::vector<Tensor> raw, act, dact;
    Tensor image, sx, sdy, sdx;   // sx / sdy / sdx: a layer's gathered input, dy and dx (reused across layers and steps)
    std::vector<float> logits_nchw;
    double last_loss = 0;
};

void u8_to_tensor(const uint8_t* img, int n, int h, int w, int c, Tensor& t) {
    // dlib input<matrix<rgb_pixel>>::to_tensor: value / 256.0, channel order R,G,B [UPSTREAM-UNVERIFIED]
    t.resize(n, h, w, c);
    const size_t total = (size_t)n * h * w * c;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; ++i) t.d[i] = (float)img[i] * (1.0f / 256.0f);
}

// inference-mode scale/shift of a bn layer ("affine"): folded from running stats
void affine_from_running(const Net& net, const Layer& L, std::vector<float>& scale, std::vector<float>& shift) {
    scale.resize(L.cout); shift.resize(L.cout);
    const float* g = net.params.data() + L.g_off;
    const float* b = net.params.data() + L.beta_off;
    const float* rm = net.running.data() + L.rs_off;
    const float* rv = rm + L.cout;
    for (int c = 0; c < L.cout; ++c) {
        const float invstd = 1.0f / std::sqrt(rv[c] + kBnEps);
        scale[c] = g[c] * invstd;
        shift[c] = fmaf(-rm[c], scale[c], b[c]);
    }
}

void gather_input(const Net& net, const Layer& L, const std::vector<Tensor>& act, Tensor& x) {
    const Tensor& a = L.in_a < 0 ? net.image : act[L.in_a];
    if (L.in_b == -2) x.copy_from(a);
    else {
        const Tensor& b = act[L.in_b];
        x.resize(a.n, a.h, a.w, a.c);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < x.d.size(); ++i) x.d[i] = a.d[i] + b.d[i];
    }
    if (net.emulate_bf16) round_tensor(x.d);  // the MFMA A operand is bf16
}

std::vector<float> layer_weights(const Net& net, const Layer& L) {
    std::vector<float> wt = to_tap_major(L, net.params.data() + L.w_off);
    if (net.emulate_bf16) round_tensor(wt);
    return wt;
}

// per-channel double sums over pixels of f(p, c) -> (s0, s1).  Serial in the parity mode (one fixed summation order);
// OpenMP with per-thread partials in the timed baseline mode (conv_algo = 1), where the order is that of the thread team.
template <typename F>
static void channel_sums2(size_t P, int C, bool parallel, std::vector<double>& s0, std::vector<double>& s1, const F& f) {
    s0.assign(C, 0.0); s1.assign(C, 0.0);
    if (!parallel) {
        for (size_t p = 0; p < P; ++p)
            for (int c = 0; c < C; ++c) { double a, b; f(p, c, a, b); s0[c] += a; s1[c] += b; }
        return;
    }
#pragma omp parallel
    {
        std::vector<double> l0(C, 0.0), l1(C, 0.0);
#pragma omp for schedule(static)
        for (size_t p = 0; p < P; ++p)
            for (int c = 0; c < C; ++c) { double a, b; f(p, c, a, b); l0[c] += a; l1[c] += b; }
#pragma omp critical
        for (int c = 0; c < C; ++c) { s0[c] += l0[c]; s1[c] += l1[c]; }
    }
}

// forward; training=false uses running stats, training=true uses batch stats and records
// mean / invstd for the backward pass.
struct BnBatch { std::vector<float> mean, invstd, scale, shift; std::vector<double> var; };

void forward(Net& net, bool training, std::vector<BnBatch>* bn_out) {
    const Spec& s = net.spec;
    const int nl = (int)s.layers.size();
    net.raw.resize(nl); net.act.resize(nl);
    if (bn_out) bn_out->assign(nl, BnBatch());
    Tensor& x = net.sx;
    for (int li = 0; li < nl; ++li) {
        const Layer& L = s.layers[li];
        gather_input(net, L, net.act, x);
        std::vector<float> wt = layer_weights(net, L);
        if (net.conv_algo == 1) conv_forward_gemm(L, wt, x, net.raw[li]); else conv_forward(L, wt, x, net.raw[li]);
        Tensor& y = net.raw[li];
        const bool act_storage = net.emulate_bf16 == 1 && !training;
        if (net.emulate_bf16 && !act_storage && L.has_bn) round_tensor(y.d);  // raw conv outputs are stored as bf16; logits stay fp32
        Tensor& a = net.act[li];
        a.resize(y.n, y.h, y.w, y.c);
        const size_t P = y.pixels();
        const int C = y.c;
        if (L.has_bn) {
            std::vector<float> scale, shift;
            if (!training) affine_from_running(net, L, scale, shift);
            else {
                BnBatch& bb = (*bn_out)[li];
                std::vector<double> sum, sq;
                channel_sums2(P, C, net.conv_algo == 1, sum, sq, [&](size_t p, int c, double& a, double& b) { const double v = y.d[p * C + c]; a = v; b = v * v; });
                bb.mean.resize(C); bb.invstd.resize(C); bb.scale.resize(C); bb.shift.resize(C); bb.var.resize(C);
                scale.resize(C); shift.resize(C);
                const float* g = net.params.data() + L.g_off;
                const float* b = net.params.data() + L.beta_off;
                for (int c = 0; c < C; ++c) {
                    const double m = sum[c] / (double)P;
                    double var = sq[c] / (double)P - m * m;
                    if (var < 0) var = 0;
                    bb.var[c] = var;
                    bb.mean[c] = (float)m;
                    bb.invstd[c] = (float)(1.0 / std::sqrt(var + (double)kBnEps));
                    scale[c] = g[c] * bb.invstd[c];
                    shift[c] = fmaf(-bb.mean[c], scale[c], b[c]);
                }
                bb.scale = scale; bb.shift = shift;
            }
#pragma omp parallel for schedule(static)
            for (size_t p = 0; p < P; ++p)
                for (int c = 0; c < C; ++c) {
                    const float z = fmaf(y.d[p * C + c], scale[c], shift[c]);
                    a.d[p * C + c] = z > 0.f ? z : 0.f;  // relu
                }
            if (act_storage) round_tensor(a.d);   // the activation is what is stored
        } else {
            const float* b = L.has_bias ? net.params.data() + L.b_off : nullptr;
            for (size_t p = 0; p < P; ++p)
                for (int c = 0; c < C; ++c) a.d[p * C + c] = b ? y.d[p * C + c] + b[c] : y.d[p * C + c];
        }
    }
    // logits NHWC -> NCHW (annonet_infer.cpp:26-30)
    const Tensor& lg = net.act[nl - 1];
    net.logits_nchw.resize(lg.d.size());
    for (int n = 0; n < lg.n; ++n)
        for (int y = 0; y < lg.h; ++y)
            for (int xx = 0; xx < lg.w; ++xx)
                for (int k = 0; k < lg.c; ++k)
                    net.logits_nchw[(((size_t)n * lg.c + k) * lg.h + y) * lg.w + xx] = lg.px(n, y, xx)[k];
}

// loss_multiclass_log_per_pixel_weighted [UPSTREAM-UNVERIFIED semantics, SURVEY §8 a19]:
//   scale = 1/(N*nr*nc); loss = sum scale*w*(-log p_y); dlogit = scale*w*(p - 1_y); ignored -> 0
double loss_and_grad(const Tensor& logits, const uint16_t* labels, const float* weights, double scale, Tensor& dlogits) {
    const size_t P = logits.pixels();
    const int K = logits.c;
    dlogits.resize(logits.n, logits.h, logits.w, K);
    double loss = 0;
#pragma omp parallel for reduction(+ : loss) schedule(static)
    for (size_t p = 0; p < P; ++p) {
        const uint16_t y = labels[p];
        float* g = dlogits.d.data() + p * K;
        if (y == kIgnore) { for (int k = 0; k < K; ++k) g[k] = 0.f; continue; }
        if (y >= K) continue;  // validated by the caller
        const float* z = logits.d.data() + p * K;
        float m = z[0];
        for (int k = 1; k < K; ++k) m = std::max(m, z[k]);
        float e[64]; float sum = 0.f;
        for (int k = 0; k < K; ++k) { e[k] = std::exp(z[k] - m); sum += e[k]; }
        const float w = weights[p];
        const float sw = (float)scale * w;
        for (int k = 0; k < K; ++k) {
            const float pk = e[k] / sum;
            if (k == y) { loss += (double)sw * -std::log(std::max(pk, 1e-10f)); g[k] = sw * (pk - 1.f); }
            else g[k] = sw * pk;
        }
    }
    return loss;
}

void train_step(Net& net, const uint8_t* images, const uint16_t* labels, const float* weights,
                int n, int h, int w, double loss_scale_n, bool apply_update) {
    const Spec& s = net.spec;
    const int nl = (int)s.layers.size();
    u8_to_tensor(images, n, h, w, s.in_ch, net.image);
    std::vector<BnBatch> bn;
    forward(net, true, &bn);
    const Tensor& logits = net.act[nl - 1];
    if (logits.h != h || logits.w != w) throw std::runtime_error("input size is not a valid net input dimension");
    for (size_t p = 0; p < (size_t)n * h * w; ++p)
        if (labels[p] != kIgnore && labels[p] >= s.classes) throw std::runtime_error("label out of range");
    net.dact.resize(nl);   // (each tensor is cleared by the resize below)
    for (int li = 0; li < nl; ++li) { const Tensor& a = net.act[li]; net.dact[li].resize(a.n, a.h, a.w, a.c); }
    const double scale = 1.0 / (loss_scale_n * (double)h * (double)w);
    net.last_loss = loss_and_grad(logits, labels, weights, scale, net.dact[nl - 1]);
    net.grads.assign(net.params.size(), 0.f);
    Tensor& x = net.sx;
    Tensor& dy = net.sdy;
    std::vector<double> dw;
    for (int li = nl - 1; li >= 0; --li) {
        const Layer& L = s.layers[li];
        const Tensor& y = net.raw[li];
        const size_t P = y.pixels();
        const int C = y.c;
        dy.resize(y.n, y.h, y.w, C);
        const Tensor& da = net.dact[li];
        if (L.has_bn) {
            const BnBatch& bb = bn[li];
            const float* g = net.params.data() + L.g_off;
            std::vector<double> dg, db;
            channel_sums2(P, C, net.conv_algo == 1, dg, db, [&](size_t p, int c, double& a, double& b) {
                const size_t i = p * C + c;
                const float dz = net.act[li].d[i] > 0.f ? da.d[i] : 0.f;
                const float xhat = (y.d[i] - bb.mean[c]) * bb.invstd[c];
                a = (double)dz * xhat; b = dz;
            });
            for (int c = 0; c < C; ++c) { net.grads[L.g_off + c] = (float)dg[c]; net.grads[L.beta_off + c] = (float)db[c]; }
            const double invP = 1.0 / (double)P;
#pragma omp parallel for schedule(static)
            for (size_t p = 0; p < P; ++p)
                for (int c = 0; c < C; ++c) {
                    const size_t i = p * C + c;
                    const float dz = net.act[li].d[i] > 0.f ? da.d[i] : 0.f;
                    const float xhat = (y.d[i] - bb.mean[c]) * bb.invstd[c];
                    dy.d[i] = (float)((double)g[c] * bb.invstd[c] * ((double)dz - db[c] * invP - (double)xhat * dg[c] * invP));
                }
            if (net.emulate_bf16) round_tensor(dy.d);
        } else {
            dy.copy_from(da);
            if (L.has_bias) {
                std::vector<double> dbias(C, 0.0);
                for (size_t p = 0; p < P; ++p) for (int c = 0; c < C; ++c) dbias[c] += da.d[p * C + c];
                for (int c = 0; c < C; ++c) net.grads[L.b_off + c] = (float)dbias[c];
            }
        }
        gather_input(net, L, net.act, x);
        if (net.conv_algo == 1) conv_backward_filter_gemm(L, x, dy, dw); else conv_backward_filter(L, x, dy, dw);
        from_tap_major_add(L, dw.data(), net.grads.data() + L.w_off);
        if (L.in_a >= 0) {
            std::vector<float> wt = layer_weights(net, L);
            Tensor& dx = net.sdx; dx.resize(x.n, x.h, x.w, x.c);
            if (net.conv_algo == 1) conv_backward_data_gemm(L, wt, dy, dx); else conv_backward_data(L, wt, dy, dx);
            const bool r = net.emulate_bf16;
            Tensor& ta = net.dact[L.in_a];
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < dx.d.size(); ++i) { ta.d[i] += dx.d[i]; if (r) ta.d[i] = bf16r(ta.d[i]); }
            if (L.in_b >= 0) {
                Tensor& tb = net.dact[L.in_b];
#pragma omp parallel for schedule(static)
                for (size_t i = 0; i < dx.d.size(); ++i) { tb.d[i] += dx.d[i]; if (r) tb.d[i] = bf16r(tb.d[i]); }
            }
        }
    }
    if (!apply_update) return;
    // running statistics, dlib bn_ [UPSTREAM-UNVERIFIED]: averaging factor 1/(updates+1); the update count is incremented
    // after use and capped AT the window (steady-state factor 1/(window+1)); variance stored unbiased (N/(N-1)).
    int bn_idx = 0;
    for (int li = 0; li < nl; ++li) {
        const Layer& L = s.layers[li];
        if (!L.has_bn) continue;
        const BnBatch& bb = bn[li];
        const double P = (double)net.raw[li].pixels();
        double& cnt = net.running_count[bn_idx++];
        const double af = 1.0 / (cnt + 1.0);
        if (cnt < (double)net.bn_window) cnt += 1.0;
        float* rm = net.running.data() + L.rs_off;
        float* rv = rm + L.cout;
        const double unb = P > 1 ? P / (P - 1.0) : 1.0;
        for (int c = 0; c < L.cout; ++c) {
            rm[c] = (float)((1.0 - af) * rm[c] + af * (double)bb.mean[c]);
            rv[c] = (float)((1.0 - af) * rv[c] + af * unb * bb.var[c]);
        }
    }
    // dlib sgd [UPSTREAM-UNVERIFIED]: v = m*v - wd*lr*w - lr*g ; w += v ; no weight decay on bias / bn params
    for (int li = 0; li < nl; ++li) {
        const Layer& L = s.layers[li];
        auto upd = [&](int64_t off, int64_t cnt, double wd) {
            for (int64_t i = off; i < off + cnt; ++i) {
                const float v = (float)(net.mom * net.momentum[i] - wd * net.lr * net.params[i] - net.lr * net.grads[i]);
                net.momentum[i] = v;
                net.params[i] += v;
            }
        };
        upd(L.w_off, (int64_t)L.k * L.k * L.cin * L.cout, net.weight_decay);
        if (L.has_bias) upd(L.b_off, L.cout, 0.0);
        if (L.has_bn) { upd(L.g_off, L.cout, 0.0); upd(L.beta_off, L.cout, 0.0); }
    }
}

// ------------------------------------------------------------------------------------------
// Tiler.  Stands in for tiling/tiling.{h,cpp} [ABSENT]; contract inferred from
// annonet_infer.cpp:42-164: full rects cover the image, unique_rect is the
// part of full_rect no other tile covers, a single tile has unique == full.
// ------------------------------------------------------------------------------------------
struct Rect { long l, t, r, b; };  // inclusive, as dlib::rectangle
struct Tile { Rect full, unique; };

void split_axis(long size, long max_tile, long overlap, std::vector<std::pair<long, long>>& spans) {
    spans.clear();
    if (size <= 0) return;
    if (size <= max_tile) { spans.push_back({0, size - 1}); return; }
    if (max_tile <= 2 * overlap) throw std::runtime_error("max tile size must exceed twice the overlap");
    const long count = (size - overlap + (max_tile - overlap) - 1) / (max_tile - overlap);
    const long len = (size + (count - 1) * overlap + count - 1) / count;  // even split, rounded up
    for (long i = 0; i < count; ++i) {
        const long start = (i * (size - len)) / (count - 1);
        spans.push_back({start, start + len - 1});
    }
    // every tile must keep a non-empty part no other tile covers (annonet_infer.cpp:102-114 divides by
    // the distance between the full and the unique edge)
    for (long i = 0; i + 2 < count; ++i)
        if (spans[i + 2].first <= spans[i].second + 1) throw std::runtime_error("max tile size is too small for this overlap");
}

std::vector<Tile> get_tiles(long width, long height, long max_w, long max_h, long ov_x, long ov_y) {
    std::vector<std::pair<long, long>> xs, ys;
    split_axis(width, max_w, ov_x, xs);
    split_axis(height, max_h, ov_y, ys);
    auto uniq = [](const std::vector<std::pair<long, long>>& s, size_t i, long size) {
        long lo = i == 0 ? 0 : s[i - 1].second + 1;
        long hi = i + 1 == s.size() ? size - 1 : s[i + 1].first - 1;
        return std::make_pair(lo, hi);
    };
    std::vector<Tile> tiles;
    for (size_t j = 0; j < ys.size(); ++j)
        for (size_t i = 0; i < xs.size(); ++i) {
            Tile t;
            t.full = {xs[i].first, ys[j].first, xs[i].second, ys[j].second};
            auto ux = uniq(xs, i, width), uy = uniq(ys, j, height);
            t.unique = {ux.first, uy.first, ux.second, uy.second};
            tiles.push_back(t);
        }
    return tiles;
}

// clamp-to-edge crop = extract_image_chip at scale 1 + outpaint (annonet_infer.cpp:68-75, annonet.h:74-120)
void crop_clamped(const uint8_t* img, int h, int w, int c, long left, long top, int th, int tw, uint8_t* out) {
    for (int y = 0; y < th; ++y) {
        const long sy = std::min<long>(std::max<long>(top + y, 0), h - 1);
        for (int x = 0; x < tw; ++x) {
            const long sx = std::min<long>(std::max<long>(left + x, 0), w - 1);
            std::memcpy(out + ((size_t)y * tw + x) * c, img + ((size_t)sy * w + sx) * c, c);
        }
    }
}

}  // namespace

// ==========================================================================================
// C interface (ctypes-friendly)
// ==========================================================================================
#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; }

extern "C" {

struct orc_layer {
    int type, k, stride, pad, cin, cout, in_a, in_b, has_bn, has_bias;
    int64_t w_off, b_off, g_off, beta_off, rs_off;
};
struct orc_tile { long full[4]; long unique[4]; };  // l,t,r,b inclusive

const char* orc_last_error() { return g_err.c_str(); }

void* orc_net_create(int levels, int in_ch, int classes, double scaler, int min_filters) {
    try {
        Net* n = new Net();
        n->spec = build_spec(levels, in_ch, classes, scaler, min_filters);
        n->params.assign(n->spec.n_params, 0.f);
        n->momentum.assign(n->spec.n_params, 0.f);
        n->grads.assign(n->spec.n_params, 0.f);
        n->running.assign(n->spec.n_running, 0.f);
        int nbn = 0;
        for (const Layer& L : n->spec.layers) {
            if (!L.has_bn) continue;
            ++nbn;
            for (int c = 0; c < L.cout; ++c) { n->params[L.g_off + c] = 1.f; n->running[L.rs_off + L.cout + c] = 1.f; }
        }
        n->running_count.assign(nbn, 0.0);
        return n;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_net_destroy(void* h) { delete (Net*)h; }
int orc_net_layer_count(void* h) { return (int)((Net*)h)->spec.layers.size(); }
int orc_net_layer(void* h, int i, orc_layer* out) {
    const Layer& L = ((Net*)h)->spec.layers[i];
    *out = {L.type, L.k, L.stride, L.pad, L.cin, L.cout, L.in_a, L.in_b, L.has_bn, L.has_bias, L.w_off, L.b_off, L.g_off, L.beta_off, L.rs_off};
    return 0;
}
int64_t orc_net_param_count(void* h) { return ((Net*)h)->spec.n_params; }
int64_t orc_net_running_count(void* h) { return ((Net*)h)->spec.n_running; }
float* orc_net_params(void* h) { return ((Net*)h)->params.data(); }
float* orc_net_momentum(void* h) { return ((Net*)h)->momentum.data(); }
float* orc_net_grads(void* h) { return ((Net*)h)->grads.data(); }
float* orc_net_running(void* h) { return ((Net*)h)->running.data(); }
double* orc_net_running_updates(void* h) { return ((Net*)h)->running_count.data(); }
void orc_net_set_hyper(void* h, double lr, double wd, double mom, unsigned long bn_window) {
    Net* n = (Net*)h; n->lr = lr; n->weight_decay = wd; n->mom = mom; n->bn_window = bn_window;
}
void orc_net_set_conv_algorithm(void* h, int algo) { ((Net*)h)->conv_algo = algo == 1 ? 1 : 0; }
void orc_net_set_bf16_emulation(void* h, int mode) { ((Net*)h)->emulate_bf16 = mode < 0 ? 0 : mode; }
int orc_required_input_dim(void* h) { return required_dim(((Net*)h)->spec); }
int orc_recommended_input_dim(int levels, int n) { return recommended_dim(levels, n); }

// RuntimeNet::Forward restatement: u8 HWC batch -> logits NCHW fp32 (inference mode: bn as affine)
int orc_forward(void* h, const uint8_t* img, int n, int hh, int ww, float* out_nchw) {
    ORC_TRY
    Net* net = (Net*)h;
    u8_to_tensor(img, n, hh, ww, net->spec.in_ch, net->image);
    forward(*net, false, nullptr);
    const Tensor& lg = net->act.back();
    if (lg.h != hh || lg.w != ww) throw std::runtime_error("input size is not a valid net input dimension");
    std::memcpy(out_nchw, net->logits_nchw.data(), sizeof(float) * net->logits_nchw.size());
    return 0;
    ORC_CATCH
}

// raw / activation taps for per-layer parity tests (NHWC fp32); which: 0 raw conv output, 1 post-activation
int orc_layer_output(void* h, int layer, int which, float* out, int64_t cap) {
    Net* net = (Net*)h;
    const Tensor& t = which == 0 ? net->raw[layer] : net->act[layer];
    if ((int64_t)t.d.size() > cap) { g_err = "buffer too small"; return -1; }
    std::memcpy(out, t.d.data(), sizeof(float) * t.d.size());
    return 0;
}
int orc_layer_dims(void* h, int layer, int* dims4) {
    const Tensor& t = ((Net*)h)->raw[layer];
    dims4[0] = t.n; dims4[1] = t.h; dims4[2] = t.w; dims4[3] = t.c;
    return 0;
}
int orc_layer_dact(void* h, int layer, float* out, int64_t cap) {
    Net* net = (Net*)h;
    const Tensor& t = net->dact[layer];
    if ((int64_t)t.d.size() > cap) { g_err = "buffer too small"; return -1; }
    std::memcpy(out, t.d.data(), sizeof(float) * t.d.size());
    return 0;
}

// TrainingNet::StartTraining restatement: one SGD step (annonet_train_main.cpp:609).
// loss_scale_n: the N in the loss scale 1/(N*nr*nc) (global batch under data parallelism).
int orc_train_step(void* h, const uint8_t* images, const uint16_t* labels, const float* weights,
                   int n, int hh, int ww, double loss_scale_n, int apply_update, double* loss) {
    ORC_TRY
    Net* net = (Net*)h;
    train_step(*net, images, labels, weights, n, hh, ww, loss_scale_n, apply_update != 0);
    if (loss) *loss = net->last_loss;
    return 0;
    ORC_CATCH
}

// ---- set_weights (annonet_train.h:20-83) ----
int orc_set_weights(const uint16_t* labels, int nr, int nc, double class_weight, double image_weight, float* weights_out) {
    ORC_TRY
    std::vector<size_t> label_counts;
    auto ensure = [](auto& v, size_t index) { if (index >= v.size()) v.resize(index * 2 + 16); };
    for (int i = 0; i < nr * nc; ++i) {
        const uint16_t label = labels[i];
        if (label != kIgnore) { ensure(label_counts, label); ++label_counts[label]; }
    }
    const size_t total_count = std::accumulate(label_counts.begin(), label_counts.end(), (size_t)0);
    std::vector<double> label_weights;
    if (total_count > 0) {
        const double average_count = total_count / (double)label_counts.size();
        double total_unnormalized_weight = 0.0;
        for (size_t label = 0; label < label_counts.size(); ++label) {
            const size_t c = label_counts[label];
            if (c > 0) {
                const double u = std::pow(average_count / c, class_weight);
                ensure(label_weights, label);
                label_weights[label] = u;
                total_unnormalized_weight += c * u;
            }
        }
        const double target = total_count * std::pow(nr * nc / (double)total_count, image_weight);
        for (double& w : label_weights) w *= target / total_unnormalized_weight;
    }
    for (int i = 0; i < nr * nc; ++i) {
        const uint16_t label = labels[i];
        weights_out[i] = (float)(label == kIgnore ? 0.0 : label_weights[label]);
    }
    return 0;
    ORC_CATCH
}

// ---- random_rect_containing_point (annonet_train.h:85-105); the two 32-bit draws are passed in ----
int orc_random_rect_containing_point(uint32_t draw_x, uint32_t draw_y, long px, long py, long rw, long rh, long* rect_ltrb) {
    ORC_TRY
    const long min_cx = px - (rw - 1) / 2, max_cx = px + rw / 2;
    const long min_cy = py - (rh - 1) / 2, max_cy = py + rh / 2;
    if (max_cx < min_cx || max_cy < min_cy) throw std::runtime_error("bad rect size");
    const long cx = min_cx + draw_x % (max_cx - min_cx + 1);
    const long cy = min_cy + draw_y % (max_cy - min_cy + 1);
    // dlib::centered_rect(p, w, h): left = x - w/2, top = y - h/2, right = left + w - 1 [UPSTREAM-UNVERIFIED]
    rect_ltrb[0] = cx - rw / 2; rect_ltrb[1] = cy - rh / 2;
    rect_ltrb[2] = rect_ltrb[0] + rw - 1; rect_ltrb[3] = rect_ltrb[1] + rh - 1;
    if (!(px >= rect_ltrb[0] && px <= rect_ltrb[2] && py >= rect_ltrb[1] && py <= rect_ltrb[3])) throw std::runtime_error("rect does not contain point");
    return 0;
    ORC_CATCH
}

// ---- outpaint (annonet.h:74-120): replicate the edge of `inside` outwards, in place ----
int orc_outpaint(uint8_t* img, int nr, int nc, int ch, long il, long it, long ir, long ib) {
    ORC_TRY
    il = std::max(il, 0L); it = std::max(it, 0L); ir = std::min<long>(ir, nc - 1); ib = std::min<long>(ib, nr - 1);
    if (il > ir || it > ib) return 0;
    auto P = [&](long r, long c) { return img + ((size_t)r * nc + c) * ch; };
    auto cp = [&](long r, long c, long sr, long sc) { std::memmove(P(r, c), P(sr, sc), ch); };
    for (long r = 0; r < it; ++r) {
        for (long c = 0; c < il; ++c) cp(r, c, it, il);
        for (long c = il; c <= ir; ++c) cp(r, c, it, c);
        for (long c = ir + 1; c < nc; ++c) cp(r, c, it, ir);
    }
    for (long r = it; r <= ib; ++r) {
        for (long c = 0; c < il; ++c) cp(r, c, r, il);
        for (long c = ir + 1; c < nc; ++c) cp(r, c, r, ir);
    }
    for (long r = ib + 1; r < nr; ++r) {
        for (long c = 0; c < il; ++c) cp(r, c, ib, il);
        for (long c = il; c <= ir; ++c) cp(r, c, ib, c);
        for (long c = ir + 1; c < nc; ++c) cp(r, c, ib, ir);
    }
    return 0;
    ORC_CATCH
}

// ---- tiler ----
int64_t orc_get_tiles(long width, long height, long max_w, long max_h, long ov_x, long ov_y, orc_tile* out, int64_t cap) {
    ORC_TRY
    std::vector<Tile> t = get_tiles(width, height, max_w, max_h, ov_x, ov_y);
    if (out) {
        if ((int64_t)t.size() > cap) throw std::runtime_error("tile buffer too small");
        for (size_t i = 0; i < t.size(); ++i) {
            out[i].full[0] = t[i].full.l; out[i].full[1] = t[i].full.t; out[i].full[2] = t[i].full.r; out[i].full[3] = t[i].full.b;
            out[i].unique[0] = t[i].unique.l; out[i].unique[1] = t[i].unique.t; out[i].unique[2] = t[i].unique.r; out[i].unique[3] = t[i].unique.b;
        }
    }
    return (int64_t)t.size();
    ORC_CATCH
}

// ---- annonet_infer() (annonet_infer.cpp:32-240) ----
// image u8 HWC -> labels u16 HW.  blended_out (optional): K planes H*W fp32.
// tiles_in (optional): explicit tile list (used to test blending with hand-made tiles).
int orc_infer(void* h, const uint8_t* image, int H, int W, const double* gains, const double* detection_levels,
              long max_w, long max_h, long ov_x, long ov_y, const orc_tile* tiles_in, int64_t n_tiles_in,
              uint16_t* result, float* blended_out) {
    ORC_TRY
    Net* net = (Net*)h;
    const int C = net->spec.in_ch, K = net->spec.classes, levels = net->spec.levels;
    std::vector<Tile> tiles;
    if (tiles_in) {
        for (int64_t i = 0; i < n_tiles_in; ++i) {
            Tile t;
            t.full = {tiles_in[i].full[0], tiles_in[i].full[1], tiles_in[i].full[2], tiles_in[i].full[3]};
            t.unique = {tiles_in[i].unique[0], tiles_in[i].unique[1], tiles_in[i].unique[2], tiles_in[i].unique[3]};
            tiles.push_back(t);
        }
    } else tiles = get_tiles(W, H, max_w, max_h, ov_x, ov_y);

    std::vector<float> blended((size_t)K * H * W, 0.f);
    std::vector<uint8_t> tile_img;
    std::vector<float> out;

    auto get_t = [](long long coordinate, long long first_possible, long long first_in, long long last_in, long long last_possible) {
        if (coordinate < first_in) return (coordinate - first_possible) / static_cast<double>(first_in - first_possible);
        else if (coordinate > last_in) return (last_possible - coordinate) / static_cast<double>(last_possible - last_in);
        else return 1.0;
    };

    for (const Tile& tile : tiles) {
        const long fw = tile.full.r - tile.full.l + 1, fh = tile.full.b - tile.full.t + 1;
        const long cx = tile.full.l + fw / 2, cy = tile.full.t + fh / 2;        // :47
        const int tw = recommended_dim(levels, (int)fw), th = recommended_dim(levels, (int)fh);  // :49-50
        const long left = cx - tw / 2, top = cy - th / 2;                       // :51-52
        tile_img.resize((size_t)th * tw * C);
        crop_clamped(image, H, W, C, left, top, th, tw, tile_img.data());       // :68-75
        out.resize((size_t)K * th * tw);
        if (orc_forward(h, tile_img.data(), 1, th, tw, out.data()) != 0) return -1;  // :77
        for (long long y = 0, by = top; y < th; ++y, ++by) {                    // :116-164
            if (by < tile.full.t || by > tile.full.b) continue;
            if (by < 0 || by >= H) continue;
            for (long long x = 0, bx = left; x < tw; ++x, ++bx) {
                if (bx < tile.full.l || bx > tile.full.r) continue;
                if (bx < 0 || bx >= W) continue;
                const bool inside_unique = bx >= tile.unique.l && bx <= tile.unique.r && by >= tile.unique.t && by <= tile.unique.b;
                for (int k = 0; k < K; ++k) {
                    const float in = out[((size_t)k * th + y) * tw + x];
                    float& o = blended[((size_t)k * H + by) * W + bx];
                    if (!inside_unique) {
                        const double thh = get_t(bx, tile.full.l, tile.unique.l, tile.unique.r, tile.full.r);
                        const double tvv = get_t(by, tile.full.t, tile.unique.t, tile.unique.b, tile.full.b);
                        const double t = thh * tvv;
                        o = (float)((double)o + t * (double)in);  // "out += t * in[in_index]" (:154): float += double*float
                    } else o = in;
                }
            }
        }
    }
    if (blended_out) std::memcpy(blended_out, blended.data(), sizeof(float) * blended.size());

    bool use_det = false;
    if (detection_levels) for (int k = 0; k < K; ++k) if (detection_levels[k] > 0.0) use_det = true;
    std::vector<std::pair<long, long>> seeds;  // (r, c)
    for (long r = 0; r < H; ++r)
        for (long c = 0; c < W; ++c) {
            uint16_t label = kIgnore;                                           // :172
            float max_value = -std::numeric_limits<float>::infinity();
            for (int k = 0; k < K; ++k) {
                const double gain = gains ? gains[k] : 0.0;
                const float value = (float)((double)blended[((size_t)k * H + r) * W + c] + gain);  // :177
                if (value > max_value) { label = (uint16_t)k; max_value = value; }
            }
            result[(size_t)r * W + c] = label;
            if (use_det && label > 0 && label != kIgnore) {
                const float clean = blended[(size_t)r * W + c];
                const float lab = blended[((size_t)label * H + r) * W + c];
                if ((double)(lab - clean) > detection_levels[label] - detection_levels[0]) seeds.push_back({r, c});  // :208
            }
        }
    if (use_det) {
        // 8-connected blobs of equal non-zero label (:217); seeds are looked up at (r,c) — the intended
        // behaviour; the reference reads them transposed (:210 vs :222), see DESIGN.md "known divergences".
        std::vector<unsigned> blob((size_t)H * W, 0);
        unsigned next = 1;
        std::vector<std::pair<long, long>> stack;
        for (long r = 0; r < H; ++r)
            for (long c = 0; c < W; ++c) {
                if (result[(size_t)r * W + c] == 0 || blob[(size_t)r * W + c]) continue;
                const uint16_t lab = result[(size_t)r * W + c];
                const unsigned id = next++;
                stack.push_back({r, c}); blob[(size_t)r * W + c] = id;
                while (!stack.empty()) {
                    auto [rr, cc] = stack.back(); stack.pop_back();
                    for (int dr = -1; dr <= 1; ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            const long r2 = rr + dr, c2 = cc + dc;
                            if (r2 < 0 || r2 >= H || c2 < 0 || c2 >= W) continue;
                            if (blob[(size_t)r2 * W + c2] || result[(size_t)r2 * W + c2] != lab) continue;
                            blob[(size_t)r2 * W + c2] = id; stack.push_back({r2, c2});
                        }
                }
            }
        std::unordered_set<unsigned> detected;
        for (auto& s : seeds) detected.insert(blob[(size_t)s.first * W + s.second]);
        for (size_t i = 0; i < (size_t)H * W; ++i)
            if (blob[i] > 0 && !detected.count(blob[i])) result[i] = 0;
    }
    return 0;
    ORC_CATCH
}


// ---- single-layer ops on caller tensors (kernel-level parity tests).  Inputs are a producer's raw output plus its
// folded bn (scale/shift; NULL = use the tensor as it is, no relu), optionally a second (skip) input that is added.
// bf16 != 0 restates the ANH_BF16 storage points: operands and filters rounded to bf16, result rounded (not dw). ----
static void op_input(const float* xa, const float* sa, const float* ta, const float* xb, const float* sb, const float* tb,
                     int n, int h, int w, int c, bool bf16, Tensor& x) {
    x.resize(n, h, w, c);
    const size_t P = x.pixels();
    for (size_t p = 0; p < P; ++p)
        for (int ch = 0; ch < c; ++ch) {
            const size_t i = p * c + ch;
            float v = xa[i];
            if (sa) { const float z = fmaf(v, sa[ch], ta[ch]); v = z > 0.f ? z : 0.f; }
            if (xb) {
                float u = xb[i];
                if (sb) { const float z = fmaf(u, sb[ch], tb[ch]); u = z > 0.f ? z : 0.f; }
                v += u;
            }
            x.d[i] = bf16 ? bf16r(v) : v;
        }
}
static Layer op_layer(int type, int k, int stride, int pad, int cin, int cout) {
    Layer L{};
    L.type = type; L.k = k; L.stride = stride; L.pad = pad; L.cin = cin; L.cout = cout;
    return L;
}

int orc_op_conv_forward(int type, int k, int stride, int pad, int cin, int cout, int n, int h, int w,
                        const float* xa, const float* sa, const float* ta, const float* xb, const float* sb, const float* tb,
                        const float* filters, const float* bias, int bf16, float* y_out) {
    ORC_TRY
    const Layer L = op_layer(type, k, stride, pad, cin, cout);
    Tensor x, y;
    op_input(xa, sa, ta, xb, sb, tb, n, h, w, cin, bf16 != 0, x);
    std::vector<float> wt = to_tap_major(L, filters);
    if (bf16) round_tensor(wt);
    conv_forward(L, wt, x, y);
    for (size_t i = 0; i < y.d.size(); ++i) {
        float v = y.d[i];
        if (bias) v = v + bias[i % cout];
        y_out[i] = (bf16 && !bias) ? bf16r(v) : v;  // biased (head) outputs stay fp32
    }
    return 0;
    ORC_CATCH
}

int orc_op_conv_backward_data(int type, int k, int stride, int pad, int cin, int cout, int n, int h_in, int w_in,
                              const float* dy, const float* filters, int bf16, float* dx_out) {
    ORC_TRY
    const Layer L = op_layer(type, k, stride, pad, cin, cout);
    Tensor g, dx;
    g.resize(n, out_dim(L, h_in), out_dim(L, w_in), cout);
    for (size_t i = 0; i < g.d.size(); ++i) g.d[i] = bf16 ? bf16r(dy[i]) : dy[i];
    dx.resize(n, h_in, w_in, cin);
    std::vector<float> wt = to_tap_major(L, filters);
    if (bf16) round_tensor(wt);
    conv_backward_data(L, wt, g, dx);
    for (size_t i = 0; i < dx.d.size(); ++i) dx_out[i] = bf16 ? bf16r(dx.d[i]) : dx.d[i];
    return 0;
    ORC_CATCH
}

int orc_op_conv_backward_filter(int type, int k, int stride, int pad, int cin, int cout, int n, int h_in, int w_in,
                                const float* xa, const float* sa, const float* ta, const float* xb, const float* sb, const float* tb,
                                const float* dy, int bf16, float* dw_canonical) {
    ORC_TRY
    const Layer L = op_layer(type, k, stride, pad, cin, cout);
    Tensor x, g;
    op_input(xa, sa, ta, xb, sb, tb, n, h_in, w_in, cin, bf16 != 0, x);
    g.resize(n, out_dim(L, h_in), out_dim(L, w_in), cout);
    for (size_t i = 0; i < g.d.size(); ++i) g.d[i] = bf16 ? bf16r(dy[i]) : dy[i];
    std::vector<double> dw;
    conv_backward_filter(L, x, g, dw);
    from_tap_major_add(L, dw.data(), dw_canonical);
    return 0;
    ORC_CATCH
}

// ---- learning-rate schedule helper: dlib count_steps_without_decrease [UPSTREAM-UNVERIFIED] ----
// Walks the loss history backwards, fits a line to the values seen so far and returns the
// largest suffix length for which P(slope < 0 in forward time) < probability_of_decrease.
int64_t orc_count_steps_without_decrease(const double* values, int64_t n, double probability_of_decrease) {
    double sn = 0, sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;
    int64_t count = 0, j = 0;
    for (int64_t i = n - 1; i >= 0; --i) {
        ++j;
        const double x = (double)sn, y = values[i];
        sn += 1; sx += x; sy += y; sxx += x * x; sxy += x * y; syy += y * y;
        if (sn > 2) {
            const double Sxx = sxx - sx * sx / sn, Sxy = sxy - sx * sy / sn, Syy = syy - sy * sy / sn;
            const double slope = Sxy / Sxx;
            double res = (Syy - slope * Sxy) / (sn - 2);
            if (res < 0) res = 0;
            const double se = std::sqrt(res / Sxx);
            double prob_pos;  // P(slope > 0) in reversed time == P(decreasing) in forward time
            if (se == 0) prob_pos = slope > 0 ? 1.0 : 0.0;
            else prob_pos = 1.0 - 0.5 * std::erfc(-(0.0 - slope) / se / std::sqrt(2.0));
            if (prob_pos < probability_of_decrease) count = j;
        }
    }
    return count;
}

}  // extern "C"
