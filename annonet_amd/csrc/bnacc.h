// bnacc.h — batch-norm accumulator tables: the per-channel sums of a bn layer (forward: sum y, sum y^2 of the stored output;
// backward: sum dz*xhat, sum dz) are ADDED by every workgroup of the kernel that sees the values — order-independent integer
// atomics, so the totals are bit-reproducible and identical on every rank — and FOLDED into (mean, invstd, scale, shift) or the
// bn backward coefficients by whichever kernel consumes them, in its table prologue.  The per-workgroup partial buffers and the
// one-wave-per-channel finalize kernels between a layer and its consumer (18 launches of a training step) are not needed then.
//
// A sum is kept as TWO int64 words: hi counts units of 2^-8, lo counts units of 2^-60 of what hi left over.  A workgroup's
// partial v (a double formed in a fixed order) is split exactly: hi = rint(v * 2^8), lo = rint((v - hi * 2^-8) * 2^60); the
// subtraction is exact in double, |lo| <= 2^51 per add, so 2^11 workgroups cannot overflow a word, and |total| < 3.6e16.  The total
// hi * 2^-8 + lo * 2^-60 carries an absolute error <= 2^-61 per add: as good as the double partials it replaces.
// Same-address atomics serialise at the memory side (measured: 256 workgroups adding 256 words each into ONE table cost a conv
// kernel 3-6 us of tail, 1024 workgroups 30 us), so a table is kept as kBnAccReplicas copies; an adder picks the copy by its
// workgroup index and a fold adds the copies up (integers: any order gives the same total).
// A partial that is not finite, or too large for hi, raises the table's poison word: every fold of that table then yields NaN — a
// diverged net shows as NaN parameters, never as finite numbers formed from a wrapped or truncated sum.  (The floating-point
// finalize kernels turn an infinite sum of squares into invstd = 0 and carry on with finite values; NaN partials give NaN there too.)
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace anh {

// table of a layer with c channels: kBnAccReplicas x [4 sums x c channels x {hi, lo}], then the poison word and the ticket counter
// of the backward sums' finish (+ padding to 64 bytes)
enum { BNACC_SUM_Y = 0, BNACC_SUM_YY = 1, BNACC_SUM_DZ_XHAT = 2, BNACC_SUM_DZ = 3 };
constexpr int kBnAccReplicas = 16;
__host__ __device__ inline size_t bnacc_words(int c) { return (size_t)kBnAccReplicas * 8 * c + 8; }

#if defined(__HIPCC__)
__device__ __forceinline__ void bnacc_add(long long* acc, int which, int c, int ch, double v) {
    const double k_hi = 256.0, k_lo = 1152921504606846976.0;   // 2^8, 2^60
    if (!(fabs(v) < 3.0e16)) {   // NaN, infinity or out of range
        atomicOr(reinterpret_cast<unsigned long long*>(acc + (size_t)kBnAccReplicas * 8 * c), 1ull);
        return;
    }
    const double hi = rint(v * k_hi);
    const double lo = rint((v - hi * (1.0 / k_hi)) * k_lo);
    const int replica = (int)(blockIdx.x % kBnAccReplicas);
    unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + (size_t)replica * 8 * c + ((size_t)which * c + ch) * 2);
    atomicAdd(p, (unsigned long long)(long long)hi);
    atomicAdd(p + 1, (unsigned long long)(long long)lo);
}

typedef __attribute__((ext_vector_type(2))) long long bnacc_pair;
__device__ __forceinline__ double bnacc_get(const long long* acc, int which, int c, int ch) {
    const long long* p = acc + ((size_t)which * c + ch) * 2;
    long long hi = 0, lo = 0;
#pragma unroll
    for (int r = 0; r < kBnAccReplicas; ++r) {
        const bnacc_pair v = *reinterpret_cast<const bnacc_pair*>(p + (size_t)r * 8 * c);   // one 16-byte load per copy
        hi += v[0]; lo += v[1];
    }
    const double v = (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 1152921504606846976.0);
    return acc[(size_t)kBnAccReplicas * 8 * c] ? __builtin_nan("") : v;
}

struct BnFolded { float mean, invstd, scale, shift; double var; };
// the arithmetic of bn_finalize_kernel (kernels_generic.hip), on the table's totals
__device__ __forceinline__ BnFolded bnacc_fold_forward(const long long* acc, int c, int ch, double pixels, float gamma, float beta, float eps) {
    const double s = bnacc_get(acc, BNACC_SUM_Y, c, ch), q = bnacc_get(acc, BNACC_SUM_YY, c, ch);
    const double m = s / pixels;
    double var = q / pixels - m * m;
    if (var < 0) var = 0;
    BnFolded f;
    f.mean = (float)m;
    f.invstd = (float)(1.0 / sqrt(var + (double)eps));
    f.scale = gamma * f.invstd;
    f.shift = fmaf(-f.mean, f.scale, beta);
    f.var = var;
    return f;
}

// The end of a kernel that added backward sums with bnacc_add: called by EVERY thread of EVERY workgroup after its adds.  The
// workgroup whose ticket is the last one folds the table — every other workgroup's adds have been acknowledged by then (each
// wave waits for its own before the workgroup draws its ticket) — and writes dgamma, dbeta and the apply coefficients.
// The table is read with agent-scope loads (served from behind this CU's caches).  total_workgroups = the whole grid.
template <typename Finish>
__device__ __forceinline__ void bnacc_finish_backward(const Finish& f, int total_workgroups) {
    __shared__ int is_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's atomics have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        unsigned long long* ticket = reinterpret_cast<unsigned long long*>(f.acc + (size_t)kBnAccReplicas * 8 * f.c + 1);
        is_last = atomicAdd(ticket, 1ull) == (unsigned long long)(total_workgroups - 1);
    }
    __syncthreads();
    if (!is_last) return;
    const int nthreads = blockDim.x * blockDim.y, tid = threadIdx.y * blockDim.x + threadIdx.x;
    const bool poisoned = __hip_atomic_load(f.acc + (size_t)kBnAccReplicas * 8 * f.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (int ch = tid; ch < f.c; ch += nthreads) {
        double sums[2];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const long long* p = f.acc + ((size_t)(BNACC_SUM_DZ_XHAT + which) * f.c + ch) * 2;
            long long hi = 0, lo = 0;
#pragma unroll
            for (int r = 0; r < kBnAccReplicas; ++r) {
                hi += __hip_atomic_load(p + (size_t)r * 8 * f.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo += __hip_atomic_load(p + (size_t)r * 8 * f.c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sums[which] = poisoned ? __builtin_nan("") : (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 1152921504606846976.0);
        }
        const double g = sums[0], b = sums[1];   // bn_bwd_finalize_kernel's arithmetic
        f.dgamma[ch] = (float)g; f.dbeta[ch] = (float)b;
        f.coef[ch] = f.gamma[ch] * f.invstd[ch];
        f.coef[f.c + ch] = (float)(b / f.pixels);
        f.coef[2 * f.c + ch] = (float)(g / f.pixels);
    }
}
#endif

}  // namespace anh
