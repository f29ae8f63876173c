#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* dst, int n_rec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane l of wave w fetches 16 B: record (w*16 + l/4), chunk (l&3) ^ ((rec>>2)&3)  -> LDS rec*64 + (l&3)*16 = w*1024 + l*16
    const int rec = wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((rec >> 2) & 3);
    const unsigned* g = src + rec * 16 + c * 4;
    if (rec < n_rec)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) dst[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
    const int n = 64 * 16;
    std::vector<unsigned> h(n); for (int i = 0; i < n; ++i) h[i] = i;
    unsigned *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, n * 4); hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(d, 0xff, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d, 60);
    std::vector<unsigned> o(n); hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int rec = 0; rec < 60; ++rec) for (int p = 0; p < 4; ++p) for (int j = 0; j < 4; ++j) {
        const int c = p ^ ((rec >> 2) & 3);
        if (o[rec * 16 + p * 4 + j] != (unsigned)(rec * 16 + c * 4 + j)) ++bad;
    }
    printf("bad %d; tail untouched-ish %u\n", bad, o[63 * 16]);
    return bad != 0;
}
