// ops.cpp — single-layer entry points of the C ABI (anh_op_*): one conv / cont forward, backward-data or
// backward-filter on host tensors, through the same kernel choice the net uses.  They exist so that each kernel can be
// checked against the oracle with IDENTICAL inputs (whole-net bf16 gradients decorrelate at the one-ulp level), and for
// per-kernel micro-benchmarks.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "spec.h"

using namespace anh;

namespace {
inline uint16_t to_bf16_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float from_bf16_bits(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

struct Stream {
    hipStream_t s = nullptr;
    Stream() { HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~Stream() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } }
};

// host fp32 -> device tensor in the mode's storage type
void upload(DevBuf& d, const float* host, size_t count, DType dt) {
    if (dt == DT_F32) {
        d.reserve(std::max<size_t>(count, 1) * 4);
        HIP_CHECK(hipMemcpy(d.p, host, count * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint16_t> tmp(count);
        for (size_t i = 0; i < count; ++i) tmp[i] = to_bf16_bits(host[i]);
        d.reserve(std::max<size_t>(count, 1) * 2);
        HIP_CHECK(hipMemcpy(d.p, tmp.data(), count * 2, hipMemcpyHostToDevice));
    }
}
void upload_f32(DevBuf& d, const float* host, size_t count) { upload(d, host, count, DT_F32); }

void download(const DevBuf& d, float* host, size_t count, DType dt) {
    if (dt == DT_F32) HIP_CHECK(hipMemcpy(host, d.p, count * 4, hipMemcpyDeviceToHost));
    else {
        std::vector<uint16_t> tmp(count);
        HIP_CHECK(hipMemcpy(tmp.data(), d.p, count * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < count; ++i) host[i] = from_bf16_bits(tmp[i]);
    }
}

struct Filters {
    DevBuf tm_f32, km_f32, tm_bf16, km_bf16;
};
// canonical -> [tap][ci][co] and [tap][co][ci]; in bf16 mode the fp32 copies carry the rounded values (as the engine's do)
void upload_filters(Filters& f, const anh_conv_desc& d, const float* canon, DType dt) {
    const int kk = d.k * d.k;
    const size_t nw = (size_t)kk * d.cin * d.cout;
    std::vector<float> tm(nw), km(nw);
    std::vector<uint16_t> tmb(nw), kmb(nw);
    for (int t = 0; t < kk; ++t)
        for (int ci = 0; ci < d.cin; ++ci)
            for (int co = 0; co < d.cout; ++co) {
                const size_t src = d.type == 0 ? ((size_t)co * d.cin + ci) * kk + t : ((size_t)ci * d.cout + co) * kk + t;
                float w = canon[src];
                const uint16_t b = to_bf16_bits(w);
                if (dt == DT_BF16) w = from_bf16_bits(b);
                const size_t i_tm = ((size_t)t * d.cin + ci) * d.cout + co, i_km = ((size_t)t * d.cout + co) * d.cin + ci;
                tm[i_tm] = w; km[i_km] = w; tmb[i_tm] = b; kmb[i_km] = b;
            }
    upload_f32(f.tm_f32, tm.data(), nw);
    upload_f32(f.km_f32, km.data(), nw);
    if (dt == DT_BF16) {
        f.tm_bf16.reserve(nw * 2); f.km_bf16.reserve(nw * 2);
        HIP_CHECK(hipMemcpy(f.tm_bf16.p, tmb.data(), nw * 2, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(f.km_bf16.p, kmb.data(), nw * 2, hipMemcpyHostToDevice));
    }
}

struct OpSource {
    DevBuf xa, xb, sa, ta, sb, tb;
    Src src;
};
void make_source(OpSource& o, const anh_op_input* a, const anh_op_input* b, size_t elems, int c, DType dt) {
    ANH_REQUIRE(a && a->x, "null input tensor");
    ANH_REQUIRE(!b || (b->x && a->scale && b->scale), "a skip input needs both inputs to carry scale/shift");
    upload(o.xa, a->x, elems, dt);
    o.src.dtype = dt;
    o.src.a = o.xa.p;
    o.src.kind = SRC_RAW;
    if (a->scale) {
        ANH_REQUIRE(a->shift, "scale without shift");
        upload_f32(o.sa, a->scale, c); upload_f32(o.ta, a->shift, c);
        o.src.a_scale = o.sa.as<float>(); o.src.a_shift = o.ta.as<float>();
        o.src.kind = SRC_ACT;
    }
    if (b) {
        ANH_REQUIRE(b->shift, "scale without shift");
        upload(o.xb, b->x, elems, dt);
        upload_f32(o.sb, b->scale, c); upload_f32(o.tb, b->shift, c);
        o.src.b = o.xb.p; o.src.b_scale = o.sb.as<float>(); o.src.b_shift = o.tb.as<float>();
        o.src.kind = SRC_ACT2;
    }
}

void check_desc(const anh_conv_desc* d, int n, int h, int w) {
    ANH_REQUIRE(d, "null descriptor");
    ANH_REQUIRE((d->type == 0 || d->type == 1) && d->k >= 1 && d->k <= 7 && d->stride >= 1 && d->stride <= 4 && d->pad >= 0 && d->cin >= 1 && d->cout >= 1,
                "bad conv descriptor");
    ANH_REQUIRE(n >= 1 && h >= 1 && w >= 1, "empty tensor");
}
int out_dim(const anh_conv_desc& d, int in) {
    anh_layer_desc L{};
    L.type = d.type; L.k = d.k; L.stride = d.stride; L.pad = d.pad;
    return Spec::out_dim(L, in);
}

template <typename F>
int guarded(F&& f) {
    try { f(); return ANH_OK; }
    catch (const Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return ANH_ERR_INTERNAL; }
}
}  // namespace

extern "C" {

int anh_op_conv_forward(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                        const float* filters, const float* bias, float* y, int* used_mfma) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(filters && y, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        OpSource in;
        make_source(in, a, b, (size_t)n * h_in * w_in * d->cin, d->cin, dt);
        Filters f;
        upload_filters(f, *d, filters, dt);
        DevBuf dbias, out;
        if (bias) upload_f32(dbias, bias, d->cout);
        const DType out_dt = bias ? DT_F32 : dt;  // biased (head) outputs are fp32 logits
        const size_t out_elems = (size_t)n * h_out * w_out * d->cout;
        out.reserve(out_elems * (out_dt == DT_BF16 ? 2 : 4));
        ConvArgs c;
        c.src = in.src;
        c.n = n; c.h_in = h_in; c.w_in = w_in; c.c_red = d->cin; c.h_out = h_out; c.w_out = w_out; c.c_out = d->cout;
        c.k = d->k; c.stride = d->stride; c.pad = d->pad; c.gather = d->type;
        c.w_f32 = f.tm_f32.as<float>(); c.w_bf16 = f.km_bf16.p;
        c.bias = bias ? dbias.as<float>() : nullptr;
        c.out = out.p; c.out_dtype = out_dt;
        const bool fast = conv_takes_mfma(c, dt);
        if (fast) launch_conv_mfma(c, st.s); else launch_conv_generic(c, st.s);
        HIP_CHECK(hipStreamSynchronize(st.s));
        download(out, y, out_elems, out_dt);
        if (used_mfma) *used_mfma = fast ? 1 : 0;
    });
}

int anh_op_conv_backward_data(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const float* dy, const float* filters,
                              float* dx, int* used_mfma) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(dy && filters && dx, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        DevBuf g, out;
        upload(g, dy, (size_t)n * h_out * w_out * d->cout, dt);
        Filters f;
        upload_filters(f, *d, filters, dt);
        const size_t out_elems = (size_t)n * h_in * w_in * d->cin;
        out.reserve(out_elems * (dt == DT_BF16 ? 2 : 4));
        ConvArgs c;
        c.src.kind = SRC_RAW; c.src.dtype = dt; c.src.a = g.p;
        c.n = n; c.h_in = h_out; c.w_in = w_out; c.c_red = d->cout; c.h_out = h_in; c.w_out = w_in; c.c_out = d->cin;
        c.k = d->k; c.stride = d->stride; c.pad = d->pad; c.gather = 1 - d->type;
        c.w_f32 = f.km_f32.as<float>(); c.w_bf16 = f.tm_bf16.p;
        c.out = out.p; c.out_dtype = dt;
        const bool fast = conv_takes_mfma(c, dt);
        if (fast) launch_conv_mfma(c, st.s); else launch_conv_generic(c, st.s);
        HIP_CHECK(hipStreamSynchronize(st.s));
        download(out, dx, out_elems, dt);
        if (used_mfma) *used_mfma = fast ? 1 : 0;
    });
}

int anh_op_conv_backward_filter(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                                const float* dy, float* dw, int* used_mfma) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(dy && dw, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        OpSource in;
        make_source(in, a, b, (size_t)n * h_in * w_in * d->cin, d->cin, dt);
        DevBuf g, out, scratch;
        upload(g, dy, (size_t)n * h_out * w_out * d->cout, dt);
        const int kk = d->k * d->k;
        const size_t nw = (size_t)kk * d->cin * d->cout;
        out.reserve(nw * 4);
        WgradArgs w;
        w.src = in.src; w.dy = g.p; w.dy_dtype = dt;
        w.n = n; w.h_in = h_in; w.w_in = w_in; w.c_in = d->cin; w.h_out = h_out; w.w_out = w_out; w.c_out = d->cout;
        w.k = d->k; w.stride = d->stride; w.pad = d->pad; w.gather = d->type;
        w.dw = out.as<float>();
        const bool fast = wgrad_takes_mfma(w, dt);
        const int64_t need = fast ? wgrad_mfma_scratch_floats(w) : wgrad_generic_scratch_floats(w);
        scratch.reserve((size_t)std::max<int64_t>(need, 1) * 4);
        w.partials = scratch.as<float>(); w.partials_capacity = (int64_t)(scratch.bytes / 4);
        if (fast) launch_wgrad_mfma(w, st.s); else launch_wgrad_generic(w, st.s);
        HIP_CHECK(hipStreamSynchronize(st.s));
        std::vector<float> tm(nw);
        HIP_CHECK(hipMemcpy(tm.data(), out.p, nw * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < kk; ++t)
            for (int ci = 0; ci < d->cin; ++ci)
                for (int co = 0; co < d->cout; ++co) {
                    const size_t dst = d->type == 0 ? ((size_t)co * d->cin + ci) * kk + t : ((size_t)ci * d->cout + co) * kk + t;
                    dw[dst] = tm[((size_t)t * d->cin + ci) * d->cout + co];
                }
        if (used_mfma) *used_mfma = fast ? 1 : 0;
    });
}

// ---- the fused forms (see include/annonet_hip.h) ----
int anh_op_conv_forward_stats(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                              const float* filters, float* y, double* sums, int* fused) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(filters && y && sums, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        OpSource in;
        make_source(in, a, b, (size_t)n * h_in * w_in * d->cin, d->cin, dt);
        Filters f;
        upload_filters(f, *d, filters, dt);
        DevBuf out, partials, stat;
        const size_t out_elems = (size_t)n * h_out * w_out * d->cout;
        const int64_t pixels = (int64_t)n * h_out * w_out;
        out.reserve(out_elems * (dt == DT_BF16 ? 2 : 4));
        ConvArgs c;
        c.src = in.src;
        c.n = n; c.h_in = h_in; c.w_in = w_in; c.c_red = d->cin; c.h_out = h_out; c.w_out = w_out; c.c_out = d->cout;
        c.k = d->k; c.stride = d->stride; c.pad = d->pad; c.gather = d->type;
        c.w_f32 = f.tm_f32.as<float>(); c.w_bf16 = f.km_bf16.p;
        c.out = out.p; c.out_dtype = dt;
        const bool fast = conv_takes_mfma(c, dt);
        int blocks = fast ? conv_fused_stat_blocks(c) : 0;
        const int fused_here = blocks > 0;
        partials.reserve((size_t)std::max(blocks, std::max(bn_partial_blocks(pixels), 1)) * 2 * d->cout * sizeof(double));
        if (fused_here) c.stat_partials = partials.as<double>();
        if (fast) launch_conv_mfma(c, st.s); else launch_conv_generic(c, st.s);
        if (!fused_here) {
            BnFwdArgs bn;
            bn.y = out.p; bn.dtype = dt; bn.pixels = pixels; bn.c = d->cout; bn.partials = partials.as<double>();
            blocks = launch_bn_forward_partials(bn, st.s);
        }
        HIP_CHECK(hipStreamSynchronize(st.s));
        download(out, y, out_elems, dt);
        std::vector<double> p((size_t)blocks * 2 * d->cout);
        HIP_CHECK(hipMemcpy(p.data(), partials.p, p.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int ch = 0; ch < d->cout; ++ch)
            for (int which = 0; which < 2; ++which) {
                double s = 0;
                for (int k = 0; k < blocks; ++k) s += p[((size_t)ch * 2 + which) * blocks + k];
                sums[ch * 2 + which] = s;
            }
        if (fused) *fused = fused_here;
    });
}

int anh_op_conv_backward_data_bn(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const float* dy, const float* filters,
                                 const float* dx_init, const float* y_prev, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, float* dx, double* sums, int* fused) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(dy && filters && dx && y_prev && scale && shift && mean && invstd && sums, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        DevBuf g, out, yp, sc, sf, mn, is, partials, coef;
        upload(g, dy, (size_t)n * h_out * w_out * d->cout, dt);
        Filters f;
        upload_filters(f, *d, filters, dt);
        const size_t out_elems = (size_t)n * h_in * w_in * d->cin;
        const int64_t pixels = (int64_t)n * h_in * w_in;
        if (dx_init) upload(out, dx_init, out_elems, dt); else out.reserve(out_elems * (dt == DT_BF16 ? 2 : 4));
        upload(yp, y_prev, out_elems, dt);
        upload_f32(sc, scale, d->cin); upload_f32(sf, shift, d->cin); upload_f32(mn, mean, d->cin); upload_f32(is, invstd, d->cin);
        ConvArgs c;
        c.src.kind = SRC_RAW; c.src.dtype = dt; c.src.a = g.p;
        c.n = n; c.h_in = h_out; c.w_in = w_out; c.c_red = d->cout; c.h_out = h_in; c.w_out = w_in; c.c_out = d->cin;
        c.k = d->k; c.stride = d->stride; c.pad = d->pad; c.gather = 1 - d->type;
        c.w_f32 = f.km_f32.as<float>(); c.w_bf16 = f.tm_bf16.p;
        c.out = out.p; c.out_dtype = dt; c.out_accumulate = dx_init ? 1 : 0;
        const bool fast = conv_takes_mfma(c, dt);
        int blocks = fast ? conv_fused_bnred_blocks(c) : 0;
        const int fused_here = blocks > 0;
        partials.reserve((size_t)std::max(blocks, std::max(bn_partial_blocks(pixels), 1)) * 2 * d->cin * sizeof(double));
        if (fused_here) {
            c.bnred_y = yp.p; c.bnred_scale = sc.as<float>(); c.bnred_shift = sf.as<float>(); c.bnred_mean = mn.as<float>(); c.bnred_invstd = is.as<float>();
            c.bnred_partials = partials.as<double>();
        }
        if (fast) launch_conv_mfma(c, st.s); else launch_conv_generic(c, st.s);
        if (!fused_here) {
            BnBwdArgs bn;
            bn.da = out.p; bn.y = yp.p; bn.dtype = dt; bn.pixels = pixels; bn.c = d->cin;
            bn.mean = mn.as<float>(); bn.invstd = is.as<float>(); bn.scale = sc.as<float>(); bn.shift = sf.as<float>();
            bn.partials = partials.as<double>();
            launch_bn_bwd_reduce(bn, st.s);
            blocks = bn_partial_blocks(pixels);
        }
        HIP_CHECK(hipStreamSynchronize(st.s));
        download(out, dx, out_elems, dt);
        std::vector<double> p((size_t)blocks * 2 * d->cin);
        HIP_CHECK(hipMemcpy(p.data(), partials.p, p.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int ch = 0; ch < d->cin; ++ch)
            for (int which = 0; which < 2; ++which) {
                double s = 0;
                for (int k = 0; k < blocks; ++k) s += p[((size_t)ch * 2 + which) * blocks + k];
                sums[ch * 2 + which] = s;
            }
        if (fused) *fused = fused_here;
    });
}

int anh_op_conv_backward_filter_bn(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const uint8_t* image,
                                   const anh_op_bn_dy* dy, float* dw, int* computed_in_kernel) {
    return guarded([&] {
        check_desc(d, n, h_in, w_in);
        ANH_REQUIRE(image && dy && dy->da && dy->y && dy->scale && dy->shift && dy->mean && dy->invstd && dy->coef && dw, "null argument");
        const DType dt = precision == ANH_BF16 ? DT_BF16 : DT_F32;
        const int h_out = out_dim(*d, h_in), w_out = out_dim(*d, w_in);
        ANH_REQUIRE(h_out >= 1 && w_out >= 1, "input too small");
        Stream st;
        DevBuf img, da, yy, sc, sf, mn, is, cf, out, scratch;
        const size_t img_bytes = (size_t)n * h_in * w_in * d->cin;
        img.reserve(img_bytes);
        HIP_CHECK(hipMemcpy(img.p, image, img_bytes, hipMemcpyHostToDevice));
        const size_t g_elems = (size_t)n * h_out * w_out * d->cout;
        upload(da, dy->da, g_elems, dt); upload(yy, dy->y, g_elems, dt);
        upload_f32(sc, dy->scale, d->cout); upload_f32(sf, dy->shift, d->cout); upload_f32(mn, dy->mean, d->cout);
        upload_f32(is, dy->invstd, d->cout); upload_f32(cf, dy->coef, (size_t)3 * d->cout);
        const int kk = d->k * d->k;
        const size_t nw = (size_t)kk * d->cin * d->cout;
        out.reserve(nw * 4);
        WgradArgs w;
        w.src.kind = SRC_IMAGE; w.src.img = img.as<uint8_t>(); w.src.img_h = h_in; w.src.img_w = w_in; w.src.img_sample_stride = (int64_t)h_in * w_in * d->cin;
        w.dy = da.p; w.dy_dtype = dt;
        w.n = n; w.h_in = h_in; w.w_in = w_in; w.c_in = d->cin; w.h_out = h_out; w.w_out = w_out; w.c_out = d->cout;
        w.k = d->k; w.stride = d->stride; w.pad = d->pad; w.gather = d->type;
        w.dw = out.as<float>();
        const bool in_kernel = wgrad_accepts_bnbwd(w, dt);
        if (in_kernel) {
            w.dy_y = yy.p; w.dy_scale = sc.as<float>(); w.dy_shift = sf.as<float>(); w.dy_mean = mn.as<float>(); w.dy_invstd = is.as<float>();
            w.dy_coef = cf.as<float>();
        } else {   // the unfused schedule: materialise dy first
            BnBwdArgs bn;
            bn.da = da.p; bn.y = yy.p; bn.dtype = dt; bn.pixels = (int64_t)n * h_out * w_out; bn.c = d->cout;
            bn.mean = mn.as<float>(); bn.invstd = is.as<float>(); bn.scale = sc.as<float>(); bn.shift = sf.as<float>(); bn.coef = cf.as<float>();
            launch_bn_bwd_apply(bn, st.s);
        }
        const bool fast = wgrad_takes_mfma(w, dt);
        const int64_t need = fast ? wgrad_mfma_scratch_floats(w) : wgrad_generic_scratch_floats(w);
        scratch.reserve((size_t)std::max<int64_t>(need, 1) * 4);
        w.partials = scratch.as<float>(); w.partials_capacity = (int64_t)(scratch.bytes / 4);
        if (fast) launch_wgrad_mfma(w, st.s); else launch_wgrad_generic(w, st.s);
        HIP_CHECK(hipStreamSynchronize(st.s));
        std::vector<float> tm(nw);
        HIP_CHECK(hipMemcpy(tm.data(), out.p, nw * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < kk; ++t)
            for (int ci = 0; ci < d->cin; ++ci)
                for (int co = 0; co < d->cout; ++co) {
                    const size_t dst = d->type == 0 ? ((size_t)co * d->cin + ci) * kk + t : ((size_t)ci * d->cout + co) * kk + t;
                    dw[dst] = tm[((size_t)t * d->cin + ci) * d->cout + co];
                }
        if (computed_in_kernel) *computed_in_kernel = in_kernel ? 1 : 0;
    });
}

}  // extern "C"
