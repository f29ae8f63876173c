#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "bnacc.h"
#include "hostlogic.h"

namespace anh {

namespace {
constexpr float kBnEps = 1e-4f;  // dlib DEFAULT_BATCH_NORM_EPS [UPSTREAM-UNVERIFIED]
inline size_t elem_size(DType d) { return d == DT_BF16 ? 2 : 4; }
}  // namespace

// ---------------------------------------------------------------------------------------------------
// Profiler
// ---------------------------------------------------------------------------------------------------
Profiler::~Profiler() {
    for (auto& p : pending_) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto e : pool_) (void)hipEventDestroy(e);
}
hipEvent_t Profiler::get_event() {
    if (!pool_.empty()) { hipEvent_t e = pool_.back(); pool_.pop_back(); return e; }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}
int Profiler::begin(hipStream_t s, const char* name, double flops, double bytes) {
    if (!enabled) return -1;
    order.emplace_back(name);
    if (sample_every > 1 && (pass_index % sample_every) != 0) return -1;
    if (!filter.empty() && std::string(name).find(filter) == std::string::npos) return -1;
    int idx = -1;
    for (size_t i = 0; i < entries.size(); ++i) if (entries[i].name == name) { idx = (int)i; break; }
    if (idx < 0) { entries.push_back(Entry{name}); idx = (int)entries.size() - 1; }
    entries[idx].flops += flops;
    entries[idx].bytes += bytes;
    entries[idx].launches += 1;
    Pending p{idx, get_event(), get_event()};
    HIP_CHECK(hipEventRecord(p.start, s));
    pending_.push_back(p);
    return (int)pending_.size() - 1;
}
void Profiler::end(hipStream_t s, int token) {
    if (token < 0) return;
    HIP_CHECK(hipEventRecord(pending_[token].stop, s));
}
void Profiler::collect() {
    for (auto& p : pending_) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, p.start, p.stop));
        entries[p.entry].total_ms += ms;
        pool_.push_back(p.start);
        pool_.push_back(p.stop);
    }
    pending_.clear();
}
void Profiler::reset() { collect(); entries.clear(); }

// ---------------------------------------------------------------------------------------------------
// construction / parameters
// ---------------------------------------------------------------------------------------------------
Engine::Engine(const anh_net_config& cfg, bool training_) : spec(Spec::build(cfg)), training(training_) {
    dtype = cfg.precision == ANH_BF16 ? DT_BF16 : DT_F32;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); fail(ANH_ERR_DEVICE, "no MI355X / HIP device visible"); }
    HIP_CHECK(hipGetDevice(&device));
    // experiment switch: ANH_STREAM_PRIORITY=1 puts the main (critical-path) stream above the filter-gradient stream
    static const int prio = getenv("ANH_STREAM_PRIORITY") ? atoi(getenv("ANH_STREAM_PRIORITY")) : 0;
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least urgent (numerically greatest)
    if (prio && training) HIP_CHECK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));
    else HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (training) {
        if (prio) HIP_CHECK(hipStreamCreateWithPriority(&aux_stream, hipStreamNonBlocking, prio == 2 ? hi : lo));
        else HIP_CHECK(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_dy_ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_aux_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_early_grads, hipEventDisableTiming));
        const char* e = getenv("ANH_CONCURRENT_WGRAD");
        if (e && e[0] == '0') concurrent_wgrad = false;
    }
    const size_t np = (size_t)spec.n_params;
    master.reserve(np * 4);
    w_tm_f32.reserve(np * 4);
    w_km_f32.reserve(np * 4);
    if (dtype == DT_BF16) { w_tm_bf16.reserve(np * 2); w_km_bf16.reserve(np * 2); }
    running.reserve(std::max<size_t>(1, (size_t)spec.n_running) * 4);
    if (training) {
        momentum.reserve(np * 4);
        grad.reserve((np + 1) * 4);
        HIP_CHECK(hipMemsetAsync(momentum.p, 0, np * 4, stream));
        HIP_CHECK(hipMemsetAsync(grad.p, 0, (np + 1) * 4, stream));
    }
    scalars.reserve(256);   // [loss | error flag | ...] in the first 64 bytes; bytes 64..255 stay zero
    HIP_CHECK(hipMemsetAsync(scalars.p, 0, 256, stream));
    loss_dev = scalars.as<double>();
    error_flag = reinterpret_cast<int*>(scalars.as<char>() + 16);

    for (size_t li = 0; li < spec.layers.size(); ++li) {
        const anh_layer_desc& L = spec.layers[li];
        ParamSegment f{L.w_off, spec.filter_count((int)li), 0, L.type, L.k, L.cin, L.cout};
        segments_host.push_back(f);
        if (L.has_bias) segments_host.push_back({L.b_off, L.cout, 1, 0, 1, 1, L.cout});
        if (L.has_bn) {
            segments_host.push_back({L.g_off, L.cout, 1, 0, 1, 1, L.cout});
            segments_host.push_back({L.beta_off, L.cout, 1, 0, 1, 1, L.cout});
        }
    }
    segments.reserve(segments_host.size() * sizeof(ParamSegment));
    HIP_CHECK(hipMemcpyAsync(segments.p, segments_host.data(), segments_host.size() * sizeof(ParamSegment), hipMemcpyHostToDevice, stream));

    ls.resize(spec.layers.size());
    int max_c = 1;
    for (size_t li = 0; li < spec.layers.size(); ++li) {
        const int C = spec.layers[li].cout;
        max_c = std::max(max_c, C);
        LayerState& s = ls[li];
        s.bn.reserve((size_t)C * (8 * sizeof(float) + sizeof(double)));
        s.mean = s.bn.as<float>(); s.invstd = s.mean + C; s.scale = s.invstd + C; s.shift = s.scale + C;
        s.var = reinterpret_cast<double*>(s.shift + C);
        s.coef = reinterpret_cast<float*>(s.var + C);  // [3][C] (+ C of padding)
    }
    coef.reserve((size_t)max_c * 3 * sizeof(float));
    HIP_CHECK(hipStreamSynchronize(stream));
    random_init(0);
}

// Every stream the engine launched on is drained before anything it owns is destroyed, whatever the host still holds (a torch
// ExternalStream over `stream`, events recorded there, pinned copies in flight): destruction never races work in flight.
Engine::~Engine() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (early_stream) (void)hipStreamSynchronize(early_stream);
    if (aux_stream) { (void)hipStreamSynchronize(aux_stream); (void)hipStreamDestroy(aux_stream); }
    for (StepGraph& c : step_graphs) if (c.exec) (void)hipGraphExecDestroy(c.exec);
    if (ev_dy_ready) (void)hipEventDestroy(ev_dy_ready);
    if (ev_aux_done) (void)hipEventDestroy(ev_aux_done);
    if (ev_early_grads) (void)hipEventDestroy(ev_early_grads);
    if (ev_early_reduced) (void)hipEventDestroy(ev_early_reduced);
    if (early_stream) (void)hipStreamDestroy(early_stream);
    if (stream && own_stream) (void)hipStreamDestroy(stream);
}

void Engine::set_stream(hipStream_t s) {
    synchronize();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    if (s) { stream = s; own_stream = false; }
    else { HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
}

void Engine::synchronize() {
    const bool bounded = bounded_waits;
    wait_stream(stream, bounded);
    if (aux_stream) wait_stream(aux_stream, bounded);
    if (prof.enabled) prof.collect();
}

void Engine::refresh_compute_weights() {
    SgdArgs a;
    a.segments = segments.as<ParamSegment>(); a.n_segments = (int)segments_host.size();
    a.n_params = spec.n_params;
    a.master = master.as<float>();
    a.w_tm_f32 = w_tm_f32.as<float>(); a.w_km_f32 = w_km_f32.as<float>();
    a.w_tm_bf16 = w_tm_bf16.p; a.w_km_bf16 = w_km_bf16.p;
    a.apply = 0;
    launch_sgd(a, stream);
}

// inference: bn layers act as dlib "affine" layers; fold gamma/beta/running stats on the host so that the
// result is bit-identical to the oracle's fold.
void Engine::fold_running_stats() {
    std::vector<float> buf;
    for (size_t li = 0; li < spec.layers.size(); ++li) {
        const anh_layer_desc& L = spec.layers[li];
        if (!L.has_bn) continue;
        const int C = L.cout;
        buf.assign((size_t)2 * C, 0.f);
        const float* g = host_params.data() + L.g_off;
        const float* b = host_params.data() + L.beta_off;
        const float* rm = host_running.data() + L.rs_off;
        const float* rv = rm + C;
        for (int c = 0; c < C; ++c) {
            const float invstd = 1.0f / std::sqrt(rv[c] + kBnEps);
            buf[c] = g[c] * invstd;
            buf[C + c] = std::fmaf(-rm[c], buf[c], b[c]);
        }
        HIP_CHECK(hipMemcpyAsync(ls[li].scale, buf.data(), (size_t)2 * C * sizeof(float), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));  // buf is reused
    }
}

void Engine::set_params(const float* params, const float* running_stats) {
    host_params.assign(params, params + spec.n_params);
    if (running_stats) host_running.assign(running_stats, running_stats + spec.n_running);
    HIP_CHECK(hipMemcpyAsync(master.p, host_params.data(), (size_t)spec.n_params * 4, hipMemcpyHostToDevice, stream));
    if (spec.n_running) HIP_CHECK(hipMemcpyAsync(running.p, host_running.data(), (size_t)spec.n_running * 4, hipMemcpyHostToDevice, stream));
    refresh_compute_weights();
    HIP_CHECK(hipStreamSynchronize(stream));
    if (!training) fold_running_stats();
}

void Engine::get_params(float* params, float* running_stats) {
    synchronize();
    if (params) HIP_CHECK(hipMemcpy(params, master.p, (size_t)spec.n_params * 4, hipMemcpyDeviceToHost));
    if (running_stats && spec.n_running) HIP_CHECK(hipMemcpy(running_stats, running.p, (size_t)spec.n_running * 4, hipMemcpyDeviceToHost));
}

void Engine::get_grads_canonical(float* out) {
    ANH_REQUIRE(training, "not a training net");
    DevBuf tmp;
    tmp.reserve((size_t)spec.n_params * 4);
    launch_tm_to_canonical(segments.as<ParamSegment>(), (int)segments_host.size(), spec.n_params, grad.as<float>(), tmp.as<float>(), stream);
    synchronize();
    HIP_CHECK(hipMemcpy(out, tmp.p, (size_t)spec.n_params * 4, hipMemcpyDeviceToHost));
}

std::vector<double> Engine::get_running_updates() const {
    std::vector<double> v;
    for (size_t li = 0; li < spec.layers.size(); ++li) if (spec.layers[li].has_bn) v.push_back(ls[li].running_updates);
    return v;
}
void Engine::set_running_updates(const std::vector<double>& v) {
    size_t k = 0;
    for (size_t li = 0; li < spec.layers.size(); ++li) if (spec.layers[li].has_bn) { ANH_REQUIRE(k < v.size(), "running update counters: one per bn layer"); ls[li].running_updates = v[k++]; }
}

void Engine::get_momentum(float* out) {
    ANH_REQUIRE(training, "not a training net");
    synchronize();
    HIP_CHECK(hipMemcpy(out, momentum.p, (size_t)spec.n_params * 4, hipMemcpyDeviceToHost));
}
void Engine::set_momentum(const float* in) {
    ANH_REQUIRE(training, "not a training net");
    synchronize();
    HIP_CHECK(hipMemcpy(momentum.p, in, (size_t)spec.n_params * 4, hipMemcpyHostToDevice));
}

// dlib-style initialisation [UPSTREAM-UNVERIFIED]: filters uniform in +-sqrt(6/(fan_in+fan_out)), bias 0, gamma 1, beta 0,
// running mean 0 / variance 1.  splitmix64 keeps it reproducible across hosts.
void Engine::random_init(uint64_t seed) {
    std::vector<float> p((size_t)spec.n_params, 0.f), r((size_t)spec.n_running, 0.f);
    uint64_t state = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    auto next = [&]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (size_t li = 0; li < spec.layers.size(); ++li) {
        const anh_layer_desc& L = spec.layers[li];
        const int64_t nw = spec.filter_count((int)li);
        const double fan = (double)L.k * L.k * (L.cin + L.cout);
        const double lim = std::sqrt(6.0 / fan);
        for (int64_t i = 0; i < nw; ++i) {
            const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
            p[L.w_off + i] = (float)((2.0 * u - 1.0) * lim);
        }
        if (L.has_bn) for (int c = 0; c < L.cout; ++c) { p[L.g_off + c] = 1.f; r[L.rs_off + L.cout + c] = 1.f; }
    }
    set_params(p.data(), r.data());
    for (auto& s : ls) s.running_updates = 0;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// A mini-batch that does not fit (find_max_mini-batch_size.cmd:43-53 probes for exactly that) fails here, before any kernel of the pass
// is enqueued, as ANH_ERR_OOM; the layer tensors grown so far are handed back, so the handle is as usable as before the call and a
// smaller batch trains on it afterwards.
void Engine::plan_dims(int n, int h, int w) {
    try { plan_dims_unguarded(n, h, w); }
    catch (const Error& e) {
        if (e.code == ANH_ERR_OOM) {
            (void)hipStreamSynchronize(stream);
            if (aux_stream) (void)hipStreamSynchronize(aux_stream);
            for (LayerState& s : ls) { s.raw.release(); s.dact.release(); s.bwd_partials.release(); }
            logits.release(); dlogits.release(); loss_partials.release(); bn_partials.release();
            have_forward = false;
        }
        throw;
    }
}

void Engine::plan_dims_unguarded(int n, int h, int w) {
    ANH_REQUIRE(n >= 1, "empty batch");
    ANH_REQUIRE(spec.valid_input_dim(h) && spec.valid_input_dim(w),
                "input size is not a valid net input dimension (see GetRecommendedInputDimension)");
    const size_t es = elem_size(dtype);
    size_t bn_need = 0;
    for (size_t li = 0; li < spec.layers.size(); ++li) {
        const anh_layer_desc& L = spec.layers[li];
        LayerState& s = ls[li];
        s.n = n;
        s.h_in = L.in_a < 0 ? h : ls[L.in_a].h;
        s.w_in = L.in_a < 0 ? w : ls[L.in_a].w;
        s.h = Spec::out_dim(L, s.h_in);
        s.w = Spec::out_dim(L, s.w_in);
        ANH_REQUIRE(s.h >= 1 && s.w >= 1, "input too small for this net depth");
        if (L.in_b >= 0) ANH_REQUIRE(ls[L.in_b].h == s.h_in && ls[L.in_b].w == s.w_in, "skip connection size mismatch");
        const size_t elems = (size_t)n * s.h * s.w * L.cout;
        if (L.has_bn) {
            s.raw.reserve(elems * es);
            if (training) s.dact.reserve(elems * es);
            // the conv kernels that fuse the statistics write one partial per workgroup (at most 1024 workgroups)
            bn_need = std::max(bn_need, (size_t)std::max(bn_partial_blocks((int64_t)n * s.h * s.w), 1024) * 2 * L.cout * sizeof(double));
        }
    }
    const anh_layer_desc& head = spec.layers.back();
    ANH_REQUIRE(ls.back().h == h && ls.back().w == w, "net output size differs from its input size");
    if (training) {
        const size_t P = (size_t)n * h * w;
        logits.reserve(P * head.cout * 4);
        dlogits.reserve(P * head.cout * 4);
        HeadTrainArgs ht;
        ht.c_in = head.cin; ht.k = head.cout; ht.pixels = (int64_t)P; ht.src.kind = SRC_ACT;
        const size_t fused_need = head_train_supported(ht) ? (size_t)head_train_partial_doubles(ht) * sizeof(double) : 0;
        loss_partials.reserve(std::max((size_t)loss_partial_blocks((int64_t)P) * (head.cout + 1) * sizeof(double), fused_need));
        bn_partials.reserve(bn_need);
    }
}

BnBwdFinish Engine::finish_of(int li) {
    const anh_layer_desc& L = spec.layers[li];
    LayerState& s = ls[li];
    BnBwdFinish f;
    f.acc = s.acc; f.gamma = master.as<float>() + L.g_off; f.invstd = s.invstd;
    f.dgamma = grad.as<float>() + L.g_off; f.dbeta = grad.as<float>() + L.beta_off; f.coef = s.coef;
    f.pixels = (double)s.n * s.h * s.w; f.c = L.cout;
    return f;
}

BnTable Engine::table_of(int li) const {
    const anh_layer_desc& L = spec.layers[li];
    const LayerState& s = ls[li];
    BnTable t;
    t.acc = s.acc; t.gamma = master.as<float>() + L.g_off; t.beta = master.as<float>() + L.beta_off;
    t.pixels = (double)s.n * s.h * s.w; t.eps = kBnEps; t.c = L.cout;
    return t;
}

// fold_tables (training forward in table mode): the consumer folds the producers' (scale, shift) from their accumulator tables.
// Backward-time consumers read the arrays, which the fold jobs have written by then.
Src Engine::layer_source(int li, const Src& image, bool fold_tables) const {
    const anh_layer_desc& L = spec.layers[li];
    if (L.in_a < 0) return image;
    Src s;
    s.dtype = dtype;
    if (fold_tables && tables) {
        s.a_tab = table_of(L.in_a);
        if (L.in_b >= 0) s.b_tab = table_of(L.in_b);
    }
    if (infer_post) {   // the producers stored relu(bn(y)) (forward_conv_args): plain reads
        s.kind = L.in_b >= 0 ? SRC_SUM2 : SRC_RAW;
        s.a = ls[L.in_a].raw.p;
        if (L.in_b >= 0) s.b = ls[L.in_b].raw.p;
        return s;
    }
    s.kind = L.in_b >= 0 ? SRC_ACT2 : SRC_ACT;
    s.a = ls[L.in_a].raw.p; s.a_scale = ls[L.in_a].scale; s.a_shift = ls[L.in_a].shift;
    if (L.in_b >= 0) { s.b = ls[L.in_b].raw.p; s.b_scale = ls[L.in_b].scale; s.b_shift = ls[L.in_b].shift; }
    return s;
}

// Arguments of layer li's forward conv.  bf16 inference with infer_post: bn layers store their post-activation output (the
// epilogue applies the layer's own folded bn + relu to the fp32 accumulators), so no consumer re-applies it while staging.
ConvArgs Engine::forward_conv_args(int li, const Src& image, bool training_pass, float* d_out_nchw) const {
    const anh_layer_desc& L = spec.layers[li];
    const LayerState& s = ls[li];
    ConvArgs a;
    a.src = layer_source(li, image, training_pass);
    a.n = s.n; a.h_in = s.h_in; a.w_in = s.w_in; a.c_red = L.cin;
    a.h_out = s.h; a.w_out = s.w; a.c_out = L.cout;
    a.k = L.k; a.stride = L.stride; a.pad = L.pad; a.gather = L.type;
    a.w_f32 = w_tm_f32.as<float>() + L.w_off;
    a.w_bf16 = dtype == DT_BF16 ? (const void*)(w_km_bf16.as<uint16_t>() + L.w_off) : nullptr;
    a.bias = L.has_bias ? master.as<float>() + L.b_off : nullptr;
    if (L.has_bn) { a.out = s.raw.p; a.out_dtype = dtype; if (infer_post) { a.out_scale = s.scale; a.out_shift = s.shift; } }
    else if (training_pass) { a.out = logits.p; a.out_dtype = DT_F32; }
    else { a.out = d_out_nchw; a.out_dtype = DT_F32; a.out_nchw = 1; }
    return a;
}

// Decides infer_post for a pass whose dimensions are planned: every bn layer's kernel must be able to store activations and every
// consumer to read them (the persistent MFMA kernels, the stem kernel, the fused head + blend or the generic 1x1 head).
void Engine::choose_inference_form(const Src& image) {
    static const bool on = !(getenv("ANH_INFER_POST_ACT") && atoi(getenv("ANH_INFER_POST_ACT")) == 0);
    infer_post = false;
    if (!on || training || dtype != DT_BF16 || spec.layers.back().in_b >= 0) return;
    infer_post = true;
    for (size_t li = 0; li + 1 < spec.layers.size(); ++li) {
        const ConvArgs a = forward_conv_args((int)li, image, false, nullptr);
        if (!spec.layers[li].has_bn || !conv_takes_mfma(a, dtype) || !conv_stores_activation(a)) { infer_post = false; return; }
    }
}

// Table mode is all or nothing per pass: every bn layer's forward kernel must add its statistics to a table, every consumer's
// kernel must fold tables, and the bn backward kernels of every width must take them.
bool Engine::choose_table_mode(const Src& image) {
    static const bool on = !(getenv("ANH_BN_TABLES") && atoi(getenv("ANH_BN_TABLES")) == 0);
    tables = false;
    if (!on || !training || dtype != DT_BF16) return false;
    const int nl = (int)spec.layers.size();
    if (nl - 1 > 16) return false;   // BnFoldJobs::job
    for (int li = 0; li + 1 < nl; ++li) {
        const anh_layer_desc& L = spec.layers[li];
        if (!L.has_bn || !bn_table_mode_ok(L.cout)) return false;
        const ConvArgs a = forward_conv_args(li, image, true, nullptr);   // (tables is false: plain arguments)
        if (!conv_takes_mfma(a, dtype) || conv_fused_stat_blocks(a) <= 0) return false;
        if (L.in_a >= 0 && !conv_folds_bn_tables(a)) return false;
    }
    // the head reads its input layers through arrays unless it is the fused kernel (which folds tables itself): both are fine
    size_t words = 0;
    for (int li = 0; li + 1 < nl; ++li) words += bnacc_words(spec.layers[li].cout);
    const void* had = bn_acc.p;
    bn_acc.reserve(words * sizeof(long long));
    if (bn_acc.p != had) tables_clean = false;
    bn_acc_bytes = words * sizeof(long long);
    words = 0;
    for (int li = 0; li + 1 < nl; ++li) { ls[li].acc = bn_acc.as<long long>() + words; words += bnacc_words(spec.layers[li].cout); }
    tables = true;
    return true;
}

void Engine::build_fold_jobs() {
    fold_jobs.n = 0;
    for (size_t li = 0; li + 1 < spec.layers.size(); ++li) {
        const anh_layer_desc& L = spec.layers[li];
        LayerState& s = ls[li];
        BnFoldJob& j = fold_jobs.job[fold_jobs.n++];
        j.acc = s.acc; j.gamma = master.as<float>() + L.g_off; j.beta = master.as<float>() + L.beta_off;
        j.mean = s.mean; j.invstd = s.invstd; j.scale = s.scale; j.shift = s.shift; j.var = s.var;
        j.rmean = s.fold_running ? running.as<float>() + L.rs_off : nullptr;
        j.rvar = s.fold_running ? running.as<float>() + L.rs_off + L.cout : nullptr;
        j.pixels = (double)s.n * s.h * s.w; j.af = s.fold_af; j.unbias = s.fold_unbias; j.eps = kBnEps; j.c = L.cout;
    }
}

void Engine::conv_dispatch(const ConvArgs& a_in, const char* tag, double flops, double bytes) {
    const ConvArgs& a = a_in;
    const bool fast = conv_takes_mfma(a, dtype);
    // the entry names the kernel family that runs: bf16 MFMA, fp32 MFMA (the parity mode on v_mfma_f32_32x32x2_f32), or the VALU kernels
    std::string name = std::string(fast ? "conv_mfma_bf16:" : conv_f32_mfma_ok(a) ? "conv_mfma_f32:" : (dtype == DT_BF16 ? "conv_generic_bf16:" : "conv_generic_f32:")) + tag;
    const int tok = prof.begin(stream, name.c_str(), flops, bytes);
    if (fast) launch_conv_mfma(a, stream);
    else launch_conv_generic(a, stream);
    prof.end(stream, tok);
}

void Engine::wgrad_dispatch(WgradArgs& a, const char* tag, double flops, double bytes, hipStream_t on, DevBuf& scratch) {
    const bool fast = wgrad_takes_mfma(a, dtype);
    const int64_t need = fast ? wgrad_mfma_scratch_floats(a) : wgrad_generic_scratch_floats(a);
    scratch.reserve((size_t)need * 4);
    a.partials = scratch.as<float>();
    a.partials_capacity = (int64_t)(scratch.bytes / 4);
    std::string name = std::string(fast ? "wgrad_mfma_bf16:" : (dtype == DT_BF16 ? "wgrad_generic_bf16:" : "wgrad_generic_f32:")) + tag;
    int splits = 0;
    a.splits_out = &splits;
    int tok = prof.begin(on, name.c_str(), flops, bytes);
    if (fast) launch_wgrad_mfma(a, on);
    else launch_wgrad_generic(a, on);
    prof.end(on, tok);
    a.splits_out = nullptr;
    if (splits > 0) {  // fixed-order sum of the per-workgroup partials
        const int64_t nw = (int64_t)a.k * a.k * a.c_in * a.c_out;
        // (the reduce of a filter gradient that runs on the MAIN stream — the stem's, in the step's tail — has a name of its own: bench.py's critical_path)
        tok = prof.begin(on, (on == stream && aux_stream && concurrent_wgrad) ? "wgrad_reduce_partials_main" : "wgrad_reduce_partials", 0, (double)splits * nw * 4.0);
        launch_reduce_partials(a.partials, splits, nw, a.dw, on);
        prof.end(on, tok);
    }
}

// profiler entry of a layer's kernel: layer kind + channel shape, so that an entry is ONE kernel instantiation at one shape
static std::string layer_tag(int li, const anh_layer_desc& L) {
    const char* kind = L.in_a < 0 ? "stem" : !L.has_bn ? "head" : L.type == 1 ? "cont3x3s2" : L.stride == 2 ? "con3x3s2" : "con3x3s1";
    return "L" + std::to_string(li) + "_" + kind + "_" + std::to_string(L.cin) + "x" + std::to_string(L.cout);
}

void Engine::run_conv_forward(int li, const Src& image, bool training_pass, float* d_out_nchw) {
    const anh_layer_desc& L = spec.layers[li];
    LayerState& s = ls[li];
    ConvArgs a = forward_conv_args(li, image, training_pass, d_out_nchw);
    if (!training_pass && li == head_epi_layer) {   // inference: this conv's epilogue also runs the 1x1 head (Engine::infer_tiles)
        const anh_layer_desc& head = spec.layers.back();
        a.head_w = w_tm_f32.as<float>() + head.w_off; a.head_bias = master.as<float>() + head.b_off; a.head_k = head.cout; a.head_out = head_epi_out;
    }
    const int64_t p_out = (int64_t)s.n * s.h * s.w, p_in = (int64_t)s.n * s.h_in * s.w_in;
    const double flops = 2.0 * L.k * L.k * L.cin * L.cout * (double)(L.type == 0 ? p_out : p_in);
    const double es = (double)elem_size(dtype);
    const double bytes = (double)p_in * L.cin * (L.in_a < 0 ? 1.0 : es) * (L.in_b >= 0 ? 2 : 1) +
                         (a.head_out ? (double)p_out * a.head_k * 4.0 : (double)p_out * L.cout * (L.has_bn ? es : 4.0)) + (double)spec.filter_count(li) * es;
    // training forward of a bn layer: the MFMA kernels that can keep per-lane running sums also write the statistic partials
    int fused_stat_blocks = 0;
    const bool table_layer = tables && training_pass && L.has_bn;
    if (table_layer) a.stat_acc = s.acc;   // the statistics go to the layer's accumulator table; its consumers fold them (bnacc.h)
    else if (L.has_bn && training_pass && conv_takes_mfma(a, dtype)) {
        fused_stat_blocks = conv_fused_stat_blocks(a);
        if (fused_stat_blocks > 0) {
            bn_partials.reserve((size_t)fused_stat_blocks * 2 * L.cout * sizeof(double));
            a.stat_partials = bn_partials.as<double>();
        }
    }
    conv_dispatch(a, (std::string("fwd_") + layer_tag(li, L)).c_str(), flops, bytes);
    if (table_layer) {   // no finalize launch: only this step's running-statistics bookkeeping, applied by the fold job
        s.fold_running = update_running_in_forward;
        if (update_running_in_forward) {
            const double P = (double)p_out;
            s.fold_af = 1.0 / (s.running_updates + 1.0);
            if (s.running_updates < (double)bn_window) s.running_updates += 1.0;
            s.fold_unbias = P > 1 ? P / (P - 1.0) : 1.0;
        }
        return;
    }
    if (L.has_bn && training_pass) {
        BnFwdArgs b;
        b.y = s.raw.p; b.dtype = dtype; b.pixels = p_out; b.c = L.cout;
        b.gamma = master.as<float>() + L.g_off; b.beta = master.as<float>() + L.beta_off;
        b.mean = s.mean; b.invstd = s.invstd; b.scale = s.scale; b.shift = s.shift; b.var = s.var;
        b.partials = bn_partials.as<double>(); b.eps = kBnEps;
        if (update_running_in_forward) {
            const double P = (double)p_out;
            b.averaging_factor = 1.0 / (s.running_updates + 1.0);
            if (s.running_updates < (double)bn_window) s.running_updates += 1.0;   // dlib bn_: count capped AT the window [UPSTREAM-UNVERIFIED]
            b.unbias = P > 1 ? P / (P - 1.0) : 1.0;
            b.running_mean = running.as<float>() + L.rs_off; b.running_var = running.as<float>() + L.rs_off + L.cout;
        }
        if (fused_stat_blocks > 0) {
            const int tok = prof.begin(stream, "bn_forward_finalize", 0, (double)fused_stat_blocks * L.cout * 16.0);
            launch_bn_forward_finalize(b, fused_stat_blocks, stream);
            prof.end(stream, tok);
        } else {
            int tok = prof.begin(stream, "bn_forward_stats", 0, (double)p_out * L.cout * es);
            const int blocks = launch_bn_forward_partials(b, stream);
            prof.end(stream, tok);
            tok = prof.begin(stream, "bn_forward_finalize", 0, (double)blocks * L.cout * 16.0);
            launch_bn_forward_finalize(b, blocks, stream);
            prof.end(stream, tok);
        }
    }
}

// training-time tail: 1x1 head + loss + head backward in one kernel when the shape allows it
bool Engine::head_is_fused() const {
    const anh_layer_desc& head = spec.layers.back();
    HeadTrainArgs t;
    t.src = layer_source((int)spec.layers.size() - 1, Src{});
    t.c_in = head.cin; t.k = head.cout;
    return training && head.k == 1 && head.in_a >= 0 && head_train_supported(t);
}

hipStream_t Engine::early_reduce_stream() {
    if (!early_stream) {
        HIP_CHECK(hipStreamCreateWithFlags(&early_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ev_early_reduced, hipEventDisableTiming));
    }
    return early_stream;
}

int64_t Engine::early_grad_first() const {
    // parameters are laid out in layer order, so "layers >= kEarlyLayer" is a suffix of the bucket; a net too shallow to have an
    // early part reports none
    if (!training || (int)spec.layers.size() <= kEarlyLayer + 1 || spec.layers[kEarlyLayer].in_a < 0) return spec.n_params + 1;
    if (step_graph_enabled()) return spec.n_params + 1;   // (an event recorded inside a captured graph cannot gate a stream outside it: one all-reduce)
    return spec.layers[kEarlyLayer].w_off;
}

void Engine::forward_inference(const Src& image, int n, int h, int w, float* d_out_nchw) {
    prof.start_pass();
    ANH_REQUIRE(!training, "forward_inference on a training net: take a runtime snapshot first");
    plan_dims(n, h, w);
    choose_inference_form(image);
    for (size_t li = 0; li < spec.layers.size(); ++li) run_conv_forward((int)li, image, false, d_out_nchw);
}

void Engine::forward_training(const Src& image, int n, int h, int w) {
    prof.hold_order = false;
    prof.start_pass();
    ANH_REQUIRE(training, "not a training net");
    plan_dims(n, h, w);
    if (choose_table_mode(image)) {
        if (!tables_clean) {   // first step, or a pass that no update followed: clear the tables here
            const int tok = prof.begin(stream, "bn_tables_zero", 0, (double)bn_acc_bytes);
            HIP_CHECK(hipMemsetAsync(bn_acc.p, 0, bn_acc_bytes, stream));
            prof.end(stream, tok);
        }
        tables_clean = false;
    }
    const bool fused_head = head_is_fused();
    const size_t n_bn = spec.layers.size() - 1;
    for (size_t li = 0; li < n_bn; ++li) run_conv_forward((int)li, image, true, nullptr);
    if (tables) {
        build_fold_jobs();
        if (fused_head) fold_pending = true;   // the first workgroups of the fused head kernel run the jobs (Engine::backward)
        else {                                 // the unfused head reads its input layers' arrays: fold them now
            const int tok = prof.begin(stream, "bn_fold_all", 0, 0);
            launch_bn_fold_all(fold_jobs, stream);
            prof.end(stream, tok);
        }
    }
    if (!fused_head) run_conv_forward((int)n_bn, image, true, nullptr);   // (the fused tail computes the logits itself)
    last_image = image; last_n = n; last_h = h; last_w = w;
    have_forward = true;
}

// ---------------------------------------------------------------------------------------------------
// backward + update
// ---------------------------------------------------------------------------------------------------
void Engine::backward(const uint16_t* d_labels, const float* d_weights, double loss_scale_n) {
    ANH_REQUIRE(training && have_forward, "backward without a training forward");
    const int nl = (int)spec.layers.size();
    const anh_layer_desc& head = spec.layers.back();
    const int64_t P = (int64_t)last_n * last_h * last_w;
    const double es = (double)elem_size(dtype);
    const bool fused_head = head_is_fused();
    for (auto& s : ls) { s.dact_written = false; s.da_alias = nullptr; }
    int head_bnred_blocks_ = 0;
    bool head_da_virtual = false;
    if (fused_head) {
        HeadTrainArgs t;
        t.src = layer_source(nl - 1, last_image, true); t.c_in = head.cin; t.k = head.cout;
        if (tables && fold_pending) {   // the fold jobs ride in the head kernel's first workgroups (a tiny pass without enough of them: own launch)
            if (head_train_blocks(P) >= fold_jobs.n) t.fold = &fold_jobs;
            else launch_bn_fold_all(fold_jobs, stream);
            fold_pending = false;
        }
        t.w_tm = w_tm_f32.as<float>() + head.w_off; t.w_km = w_km_f32.as<float>() + head.w_off;
        t.bias = master.as<float>() + head.b_off;
        t.labels = d_labels; t.weights = d_weights;
        t.logits = logits.as<float>(); t.da = ls[head.in_a].dact.p;
        t.pixels = P; t.scale = 1.0 / (loss_scale_n * (double)last_h * (double)last_w);
        t.partials = loss_partials.as<double>();
        t.loss_out = loss_dev; t.loss_out_f32 = grad.as<float>() + spec.n_params;
        t.dbias = grad.as<float>() + head.b_off; t.dw = grad.as<float>() + head.w_off;
        t.error_flag = error_flag;
        // the head's input layer (single bn layer): its da is produced here from its y -> leave its bn backward sums too
        static const bool fuse_head_bnred = !(getenv("ANH_FUSE_BN_BWD_REDUCE") && atoi(getenv("ANH_FUSE_BN_BWD_REDUCE")) == 0);
        LayerState& hp = ls[head.in_a];
        int head_bnred_blocks = 0;
        if (fuse_head_bnred && t.src.kind == SRC_ACT && spec.layers[head.in_a].has_bn) {
            head_bnred_blocks = head_train_blocks(P);
            t.bnred_mean = hp.mean; t.bnred_invstd = hp.invstd;
            if (tables) { t.bnred_acc = hp.acc; t.bnred_finish = finish_of(head.in_a); }
            else {
                hp.bwd_partials.reserve((size_t)head_bnred_blocks * 2 * head.cin * sizeof(double));
                t.bnred_partials = hp.bwd_partials.as<double>();
            }
            // ... and then nothing needs da in memory: the layer's bn_bwd_apply pass recomputes it from the dlogits (k floats per pixel
            // instead of 32 storage elements written here and read there).
            static const bool virtual_da = !(getenv("ANH_HEAD_DA_VIRTUAL") && atoi(getenv("ANH_HEAD_DA_VIRTUAL")) == 0);
            int readers = 0;   // a second consumer of that layer would ADD its gradient to the da in memory
            for (const anh_layer_desc& X : spec.layers) readers += (X.in_a == head.in_a) + (X.in_b == head.in_a);
            if (virtual_da && readers == 1) { t.da = nullptr; t.dlogits = dlogits.as<float>(); head_da_virtual = true; }
        }
        const int tok = prof.begin(stream, "head_fused_fwd_loss_bwd", 2.0 * 3 * head.cin * head.cout * (double)P, (double)P * (head.cin * es * (head_da_virtual ? 1 : 2) + head.cout * (head_da_virtual ? 8.0 : 4.0) + 6.0));
        launch_head_train(t, stream);
        prof.end(stream, tok);
        head_bnred_blocks_ = head_bnred_blocks;
        ls[head.in_a].dact_written = true;
    } else {
        LossArgs a;
        a.logits = logits.as<float>(); a.labels = d_labels; a.weights = d_weights; a.dlogits = dlogits.as<float>();
        a.pixels = P; a.k = head.cout;
        a.scale = 1.0 / (loss_scale_n * (double)last_h * (double)last_w);
        a.partials = loss_partials.as<double>(); a.loss_out = loss_dev;
        a.loss_out_f32 = grad.as<float>() + spec.n_params;
        a.dbias = grad.as<float>() + head.b_off;
        a.error_flag = error_flag;
        const int tok = prof.begin(stream, "softmax_logloss", 0, (double)P * (head.cout * 8.0 + 6.0));
        launch_loss(a, stream);
        prof.end(stream, tok);
    }
    auto run_layers = [&]() {   // everything behind the head: per layer bn backward, filter gradient (second stream), backward-data conv; the join
    for (int i = 0; i < nl; ++i) { ls[i].consumers = 0; ls[i].da_writes = 0; ls[i].fused_bwd_blocks = 0; }
    for (int i = 0; i < nl; ++i) {
        if (spec.layers[i].in_a >= 0) ++ls[spec.layers[i].in_a].consumers;
        if (spec.layers[i].in_b >= 0) ++ls[spec.layers[i].in_b].consumers;
    }
    if (fused_head) { ++ls[head.in_a].da_writes; ls[head.in_a].fused_bwd_blocks = head_bnred_blocks_; }  // the fused head kernel wrote d(dec) already
    for (int li = fused_head ? nl - 2 : nl - 1; li >= 0; --li) {
        const anh_layer_desc& L = spec.layers[li];
        LayerState& s = ls[li];
        const int64_t p_out = (int64_t)s.n * s.h * s.w, p_in = (int64_t)s.n * s.h_in * s.w_in;
        const void* dy;
        DType dy_dt;
        const double flops = 2.0 * L.k * L.k * L.cin * L.cout * (double)(L.type == 0 ? p_out : p_in);
        const bool two_streams = concurrent_wgrad && aux_stream;

        // backward-data conv of this layer (arguments first: whether it can take the bn backward in its prologue decides the schedule)
        ConvArgs dg;
        const bool has_dgrad = L.in_a >= 0;
        if (has_dgrad) {
            dg.src.kind = SRC_RAW; dg.src.dtype = L.has_bn ? dtype : DT_F32;
            dg.n = s.n; dg.h_in = s.h; dg.w_in = s.w; dg.c_red = L.cout;
            dg.h_out = s.h_in; dg.w_out = s.w_in; dg.c_out = L.cin;
            dg.k = L.k; dg.stride = L.stride; dg.pad = L.pad; dg.gather = 1 - L.type;
            dg.w_f32 = w_km_f32.as<float>() + L.w_off;
            dg.w_bf16 = dtype == DT_BF16 ? (const void*)(w_tm_bf16.as<uint16_t>() + L.w_off) : nullptr;
            dg.out = ls[L.in_a].dact.p; dg.out_dtype = dtype; dg.out_accumulate = ls[L.in_a].dact_written ? 1 : 0;
            if (L.in_b >= 0) { dg.out2 = ls[L.in_b].dact.p; dg.out2_accumulate = ls[L.in_b].dact_written ? 1 : 0; }
            // The gradient w.r.t. a skip sum is ONE tensor for both addends.  When this conv is the first writer of both, it is stored
            // once, in the skip source's buffer; the main source's apply pass reads it there and writes its dy out of place
            // (bit-identical; 157 MB less per step; six-round same-box A/B 1.7137 -> 1.7040 ms.  ANH_SKIP_GRAD_ONCE=0: two stores).
            static const bool skip_once = !(getenv("ANH_SKIP_GRAD_ONCE") && atoi(getenv("ANH_SKIP_GRAD_ONCE")) == 0);
            if (skip_once && L.in_b >= 0 && !ls[L.in_a].dact_written && !ls[L.in_b].dact_written && spec.layers[L.in_a].has_bn && spec.layers[L.in_a].in_a >= 0 &&
                !(head_da_virtual && L.in_a == head.in_a) && ls[L.in_a].dact.bytes == ls[L.in_b].dact.bytes) {
                dg.out = ls[L.in_b].dact.p; dg.out_accumulate = 0;
                dg.out2 = nullptr; dg.out2_accumulate = 0;
                ls[L.in_a].da_alias = ls[L.in_b].dact.p;
            }
        }
        // bn + relu backward of this layer: sums (left by the conv that wrote da, else a reduce pass) -> finalize -> one elementwise
        // apply pass that turns da into dy in place, read by the backward-data conv (this stream) and the filter gradient (second
        // stream).  Every attempt to fold that pass into a conv's staging lost; the stem, which has no backward-data
        // conv, folds it into its filter-gradient kernel.
        bool wgrad_computes_dy = false;
        if (L.has_bn) {
            ANH_REQUIRE(s.dact_written, "internal: layer output has no consumer");
            BnBwdArgs b;
            b.da = s.da_alias ? const_cast<void*>(s.da_alias) : s.dact.p; b.y = s.raw.p; b.dtype = dtype; b.pixels = p_out; b.c = L.cout;
            if (s.da_alias) b.dy_out = s.dact.p;
            b.gamma = master.as<float>() + L.g_off; b.mean = s.mean; b.invstd = s.invstd; b.scale = s.scale; b.shift = s.shift;
            b.dgamma = grad.as<float>() + L.g_off; b.dbeta = grad.as<float>() + L.beta_off;
            b.partials = bn_partials.as<double>(); b.coef = s.coef;
            if (head_da_virtual && li == head.in_a) { b.head_g = dlogits.as<float>(); b.head_w_tm = w_tm_f32.as<float>() + head.w_off; b.head_k = head.cout; }
            // one profiler entry per kernel: reduce reads da and y; apply reads both and writes dy
            int tok;
            if (tables) { b.acc = s.acc; b.finish = finish_of(li); }   // table mode: sums in the layer's accumulator table, folded by the last workgroup that adds to it
            if (s.fused_bwd_blocks > 0) {   // the conv that wrote da last left the partial sums
                if (!tables) { b.partials = s.bwd_partials.as<double>(); b.partial_blocks = s.fused_bwd_blocks; }
            } else {
                tok = prof.begin(stream, "bn_bwd_reduce", 0, (double)p_out * L.cout * es * 2);
                launch_bn_bwd_reduce(b, stream);
                prof.end(stream, tok);
            }
            if (!tables) {
                tok = prof.begin(stream, "bn_bwd_finalize", 0, (double)(b.partial_blocks > 0 ? b.partial_blocks : bn_partial_blocks(p_out)) * L.cout * 16.0);
                launch_bn_bwd_finalize(b, stream);
                prof.end(stream, tok);
            }
            // a layer without backward-data conv (the stem): only its filter gradient consumes dy, and the stem wgrad
            // kernel can compute dy from (da, y) while staging -> no apply pass on the critical path
            WgradArgs probe;
            probe.src = layer_source(li, last_image); probe.dy_dtype = dtype;
            probe.n = s.n; probe.h_in = s.h_in; probe.w_in = s.w_in; probe.c_in = L.cin; probe.h_out = s.h; probe.w_out = s.w; probe.c_out = L.cout;
            probe.k = L.k; probe.stride = L.stride; probe.pad = L.pad; probe.gather = L.type;
            wgrad_computes_dy = !has_dgrad && wgrad_accepts_bnbwd(probe, dtype);
            if (!wgrad_computes_dy) {
                tok = prof.begin(stream, "bn_bwd_apply", 0, b.head_g ? (double)p_out * (L.cout * es * 2 + head.cout * 4.0) : (double)p_out * L.cout * es * 3);
                launch_bn_bwd_apply(b, stream);
                prof.end(stream, tok);
            }
            dy = s.dact.p; dy_dt = dtype;
            if (has_dgrad) dg.src.a = dy;
        } else { dy = dlogits.p; dy_dt = DT_F32; if (has_dgrad) dg.src.a = dy; }

        auto run_wgrad = [&]() {   // filter gradient
            WgradArgs g;
            g.src = layer_source(li, last_image);
            g.dy = dy; g.dy_dtype = dy_dt;
            g.n = s.n; g.h_in = s.h_in; g.w_in = s.w_in; g.c_in = L.cin;
            g.h_out = s.h; g.w_out = s.w; g.c_out = L.cout;
            g.k = L.k; g.stride = L.stride; g.pad = L.pad; g.gather = L.type;
            g.dw = grad.as<float>() + L.w_off;
            if (wgrad_computes_dy) {
                g.dy = s.dact.p; g.dy_y = s.raw.p;
                g.dy_scale = s.scale; g.dy_shift = s.shift; g.dy_mean = s.mean; g.dy_invstd = s.invstd; g.dy_coef = s.coef;
            }
            const double bytes = (double)p_in * L.cin * (L.in_a < 0 ? 1.0 : es) * (L.in_b >= 0 ? 2 : 1) + (double)p_out * L.cout * (L.has_bn ? es : 4.0) * (wgrad_computes_dy ? 2 : 1);
            hipStream_t on = stream;
            // A layer without backward-data conv (the stem, last in this loop) leaves the main stream with nothing else to
            // do: its filter gradient runs THERE, beside the previous layer's filter gradient still on the second stream
            // (own partials buffer), instead of queueing behind it.
            static const bool tail_on_main = getenv("ANH_STEM_WGRAD_MAIN") ? atoi(getenv("ANH_STEM_WGRAD_MAIN")) != 0 : true;
            const bool on_main = two_streams && !has_dgrad && tail_on_main;
            if (two_streams && !on_main) {
                HIP_CHECK(hipEventRecord(ev_dy_ready, stream));   // dy of this layer is final on the main stream: let the second stream pick it up
                HIP_CHECK(hipStreamWaitEvent(aux_stream, ev_dy_ready, 0));
                on = aux_stream;
            }
            wgrad_dispatch(g, (std::string("wgrad_") + layer_tag(li, L)).c_str(), flops, bytes, on, on_main ? wgrad_partials_main : wgrad_partials);
        };
        auto run_dgrad = [&]() {   // data gradient -> d(activation of the producing layers)
            LayerState& P = ls[L.in_a];
            const anh_layer_desc& PL = spec.layers[L.in_a];
            P.dact_written = true; ++P.da_writes;
            if (L.in_b >= 0) { ls[L.in_b].dact_written = true; ++ls[L.in_b].da_writes; }
            // `out` becomes final with this conv: let its epilogue do that layer's bn backward reduction
            if (PL.has_bn && P.da_writes == P.consumers && dtype == DT_BF16 && conv_takes_mfma(dg, dtype)) {
                const int blocks = conv_fused_bnred_blocks(dg);
                if (blocks > 0) {
                    dg.bnred_y = P.raw.p; dg.bnred_scale = P.scale; dg.bnred_shift = P.shift; dg.bnred_mean = P.mean; dg.bnred_invstd = P.invstd;
                    if (tables) { dg.bnred_acc = P.acc; dg.bnred_finish = finish_of(L.in_a); }
                    else {
                        P.bwd_partials.reserve((size_t)blocks * 2 * PL.cout * sizeof(double));
                        dg.bnred_partials = P.bwd_partials.as<double>();
                    }
                    P.fused_bwd_blocks = blocks;
                }
            }
            const double bytes = (double)p_out * L.cout * (L.has_bn ? es : 4.0) +
                                 (double)p_in * L.cin * es * (L.in_b >= 0 ? 2 : 1) * (dg.out_accumulate ? 2 : 1);
            conv_dispatch(dg, (std::string("dgrad_") + layer_tag(li, L)).c_str(), flops, bytes);
        };
        run_wgrad();
        if (li == kEarlyLayer && early_grad_first() <= spec.n_params)   // gradients of layers >= kEarlyLayer, head and loss are final here
            HIP_CHECK(hipEventRecord(ev_early_grads, (two_streams && has_dgrad) ? aux_stream : stream));
        if (has_dgrad) run_dgrad();
    }
    if (concurrent_wgrad && aux_stream) {  // gradients are complete on the main stream only after the aux stream drains
        HIP_CHECK(hipEventRecord(ev_aux_done, aux_stream));
        HIP_CHECK(hipStreamWaitEvent(stream, ev_aux_done, 0));
    }
    };
    // ---- ANH_STEP_GRAPH=1 (round 5): the ~36 launches behind the head — both streams, the eight dy hand-overs and
    // the join as graph edges — replayed as ONE hipGraphLaunch.  Nothing in them changes from step to step (the per-step values: the fold
    // jobs' averaging factors ride in the head kernel, lr / loss tag / loss slot in the update kernel; both stay ordinary launches), so an
    // instantiated graph is valid as long as the shapes, the buffers and the input image pointer (the stem's filter gradient reads it) are
    // the same: they are the cache key.  A key runs eagerly twice (allocations, LDS attributes) before it is captured. ----
    if (!step_graph_enabled() || prof.enabled) { run_layers(); return; }
    uint64_t key = 1469598103934665603ull;
    auto mix = [&key](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    mix((uint64_t)last_n); mix((uint64_t)last_h); mix((uint64_t)last_w); mix((uint64_t)(uintptr_t)last_image.img); mix((uint64_t)last_image.img_sample_stride);
    mix((uint64_t)last_image.img_left); mix((uint64_t)last_image.img_top); mix((uint64_t)last_image.img_h); mix((uint64_t)last_image.img_w); mix((uint64_t)last_image.img_nwin);
    for (const LayerState& s : ls) { mix((uint64_t)(uintptr_t)s.raw.p); mix((uint64_t)(uintptr_t)s.dact.p); mix((uint64_t)(uintptr_t)s.bwd_partials.p); }
    mix((uint64_t)(uintptr_t)dlogits.p); mix((uint64_t)(uintptr_t)wgrad_partials.p); mix((uint64_t)(uintptr_t)wgrad_partials_main.p); mix((uint64_t)(uintptr_t)bn_partials.p);
    mix((uint64_t)(uintptr_t)stream); mix((uint64_t)(uintptr_t)aux_stream); mix((uint64_t)tables); mix((uint64_t)concurrent_wgrad); mix((uint64_t)head_da_virtual); mix((uint64_t)head_bnred_blocks_);
    StepGraph* g = nullptr;
    for (StepGraph& c : step_graphs) if (c.key == key) g = &c;
    if (!g) {
        if (step_graphs.size() >= 4) { for (StepGraph& c : step_graphs) if (c.exec) (void)hipGraphExecDestroy(c.exec); step_graphs.clear(); }
        step_graphs.push_back(StepGraph{key, 0, nullptr});
        g = &step_graphs.back();
    }
    if (!g->exec && g->eager_runs < 2) { ++g->eager_runs; run_layers(); return; }
    if (!g->exec) {
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        try { run_layers(); }
        catch (...) { (void)hipStreamEndCapture(stream, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
        HIP_CHECK(hipStreamEndCapture(stream, &graph));
        const hipError_t e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_CHECK(e);
        ++step_graph_captures;
    }
    HIP_CHECK(hipGraphLaunch(g->exec, stream));
    ++step_graph_launches;
}

bool Engine::step_graph_enabled() {
    static const bool on = getenv("ANH_STEP_GRAPH") && atoi(getenv("ANH_STEP_GRAPH")) != 0;
    return on;
}

void Engine::apply_update(double lr, double weight_decay, double momentum_coef, double grad_scale, unsigned long bn_window_arg, unsigned long long* loss_post, unsigned int loss_tag) {
    ANH_REQUIRE(training && have_forward, "apply_update without a step");
    (void)bn_window_arg;  // running statistics are updated by the training forward (bn_finalize), as dlib's bn_ does
    SgdArgs a;
    a.segments = segments.as<ParamSegment>(); a.n_segments = (int)segments_host.size();
    a.n_params = spec.n_params;
    a.master = master.as<float>(); a.momentum = momentum.as<float>(); a.grad_tm = grad.as<float>();
    a.w_tm_f32 = w_tm_f32.as<float>(); a.w_km_f32 = w_km_f32.as<float>();
    a.w_tm_bf16 = w_tm_bf16.p; a.w_km_bf16 = w_km_bf16.p;
    a.lr = lr; a.weight_decay = weight_decay; a.momentum_coef = momentum_coef; a.grad_scale = grad_scale; a.apply = 1;
    if (loss_post) { a.loss_src = grad.as<float>() + spec.n_params; a.loss_post = loss_post; a.loss_tag = loss_tag; }
    if (tables && bn_acc.p && bn_acc_bytes % 16 == 0) { a.zero = bn_acc.p; a.zero_words16 = (int64_t)(bn_acc_bytes / 16); tables_clean = true; }
    const int tok = prof.begin(stream, "sgd_momentum_wd", 0, (double)spec.n_params * 28);
    launch_sgd(a, stream);
    prof.end(stream, tok);
}

double Engine::read_loss() {
    synchronize();
    float v = 0;
    HIP_CHECK(hipMemcpy(&v, grad.as<float>() + spec.n_params, sizeof(float), hipMemcpyDeviceToHost));
    return (double)v;
}

int Engine::read_error_flag_and_clear() {
    synchronize();
    int v = 0;
    HIP_CHECK(hipMemcpy(&v, error_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (v) HIP_CHECK(hipMemset(error_flag, 0, sizeof(int)));
    return v;
}

void Engine::layer_tensor(int layer, int which, float* out_host, int64_t capacity, int dims[4]) {
    ANH_REQUIRE(layer >= 0 && layer < (int)ls.size(), "layer index out of range");
    const anh_layer_desc& L = spec.layers[layer];
    const LayerState& s = ls[layer];
    dims[0] = s.n; dims[1] = s.h; dims[2] = s.w; dims[3] = L.cout;
    const int64_t elems = (int64_t)s.n * s.h * s.w * L.cout;
    if (!out_host) return;
    ANH_REQUIRE(elems <= capacity, "buffer too small");
    synchronize();
    const void* src;
    DType dt = dtype;
    if (!L.has_bn) { src = which == 0 ? logits.p : dlogits.p; dt = DT_F32; ANH_REQUIRE(training, "head taps exist only on a training net"); }
    else src = which == 0 ? s.raw.p : s.dact.p;
    ANH_REQUIRE(src != nullptr, "tensor not available");
    if (dt == DT_F32) HIP_CHECK(hipMemcpy(out_host, src, (size_t)elems * 4, hipMemcpyDeviceToHost));
    else {
        std::vector<uint16_t> tmp((size_t)elems);
        HIP_CHECK(hipMemcpy(tmp.data(), src, (size_t)elems * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < elems; ++i) { uint32_t u = (uint32_t)tmp[i] << 16; std::memcpy(out_host + i, &u, 4); }
    }
}

// ---------------------------------------------------------------------------------------------------
// tiled inference (annonet_infer.cpp:42-214)
// ---------------------------------------------------------------------------------------------------
// one tile of annonet_infer(): clamp-to-edge window of the resident image -> net forward -> blend into the resident planes
void Engine::infer_tile(const anh_tile& t, const uint8_t* d_image, int H, int W, float* d_blended) { infer_tiles(&t, 1, d_image, H, W, d_blended); }

// `count` (<= kMaxTileBatch) tiles whose input windows have the same size run through the net as ONE batch — every conv launch
// then carries count times the work, so its fixed ramp and tail (a third of a 30-us launch on a 1024^2 tile) is paid once —
// and are blended one after the other, in list order: the additive blended_output sees exactly the sequence of the per-tile loop
// (annonet_infer.cpp:116-164), so the result does not depend on the batch size.
void Engine::infer_tiles(const anh_tile* ts, int count, const uint8_t* d_image, int H, int W, float* d_blended) {
    ANH_REQUIRE(count >= 1 && count <= kMaxTileBatch, "infer_tiles: batch size out of range");
    const int K = spec.cfg.classes;
    const TileWindow win = tile_window(ts[0], spec.cfg.levels);
    Src image;
    image.kind = SRC_IMAGE;
    image.img = d_image; image.img_h = H; image.img_w = W; image.img_left = win.left; image.img_top = win.top;
    if (count > 1) {
        image.img_nwin = count;   // (img_sample_stride stays 0: the samples are windows of the same image)
        for (int i = 0; i < count; ++i) {
            const TileWindow wi = tile_window(ts[i], spec.cfg.levels);
            ANH_REQUIRE(wi.height == win.height && wi.width == win.width, "infer_tiles: tiles of one batch must have equal windows");
            image.img_win[2 * i] = wi.left; image.img_win[2 * i + 1] = wi.top;
        }
    }
    // bf16 mode: the 1x1 head and the blend run as one kernel over the last hidden tensor (the tile's logits stay on chip)
    const anh_layer_desc& head = spec.layers.back();
    HeadBlendArgs hb;
    hb.src = layer_source((int)spec.layers.size() - 1, image); hb.c_in = head.cin; hb.k = head.cout;
    const bool fuse_head = !training && head.k == 1 && head.has_bias && head.in_a >= 0 && head_blend_supported(hb);
    bool head_epi = false;   // the head rides in the epilogue of the last hidden layer's conv: its logits, not its activation, go to memory
    DevBuf& tout = tile_out;
    if (fuse_head) {
        prof.start_pass();
        plan_dims(count, win.height, win.width);
        choose_inference_form(image);
        const int hl = head.in_a;
        if (head.in_b < 0 && hl == (int)spec.layers.size() - 2) {
            int readers = 0;
            for (const anh_layer_desc& X : spec.layers) readers += (X.in_a == hl) + (X.in_b == hl);
            ConvArgs probe = forward_conv_args(hl, image, false, nullptr);
            probe.head_k = head.cout;
            head_epi = readers == 1 && conv_head_in_epilogue_ok(probe);
        }
        if (head_epi) {
            tout.reserve((size_t)count * K * win.height * win.width * 4);
            head_epi_layer = hl; head_epi_out = tout.as<float>();
        }
        for (size_t li = 0; li + 1 < spec.layers.size(); ++li) run_conv_forward((int)li, image, false, nullptr);
        head_epi_layer = -1; head_epi_out = nullptr;
    } else {
        tout.reserve((size_t)count * K * win.height * win.width * 4);
        forward_inference(image, count, win.height, win.width, tout.as<float>());
    }
    const size_t es = elem_size(dtype);
    const hipStream_t bs = stream;   // (a batch's blends on a second stream beside the next batch's convs measured no gain in round 4: the persistent conv kernels hold every CU)
    // the whole batch's tiles by ONE blend launch (round 5): unique-rectangle pixels are assignments, a frame pixel is gathered by the first
    // tile of the batch that covers it, contributions in list order — bit for bit what `count` launches in list order leave
    static const bool batch_blend = !(getenv("ANH_BLEND_BATCH") && atoi(getenv("ANH_BLEND_BATCH")) == 0);
    if (batch_blend && count > 1 && !(fuse_head && !head_epi) && K <= 4 && (int64_t)H * W < 0x7fffffffll) {
        BlendBatchArgs bb;
        bb.logits = tout.as<float>(); bb.blended = d_blended; bb.k = K; bb.count = count; bb.tile_h = win.height; bb.tile_w = win.width; bb.img_h = H; bb.img_w = W;
        for (int i = 0; i < count; ++i) {
            const TileWindow wi = tile_window(ts[i], spec.cfg.levels);
            bb.left[i] = wi.left; bb.top[i] = wi.top;
            const anh_rect &f = ts[i].full_rect, &u = ts[i].unique_rect;
            bb.full[i][0] = (int)f.left; bb.full[i][1] = (int)f.top; bb.full[i][2] = (int)f.right; bb.full[i][3] = (int)f.bottom;
            bb.unique[i][0] = (int)u.left; bb.unique[i][1] = (int)u.top; bb.unique[i][2] = (int)u.right; bb.unique[i][3] = (int)u.bottom;
        }
        if (blend_batch_ok(bb)) {
            const int tok = prof.begin(bs, "blend_accumulate", 0, (double)count * K * win.height * win.width * 12);
            launch_blend_batch(bb, bs);
            prof.end(bs, tok);
            return;
        }
    }
    for (int i = 0; i < count; ++i) {
        const anh_tile& t = ts[i];
        const TileWindow wi = tile_window(t, spec.cfg.levels);
        BlendArgs b;
        b.logits_nchw = (fuse_head && !head_epi) ? nullptr : tout.as<float>() + (size_t)i * K * win.height * win.width; b.blended = d_blended;
        b.k = K; b.tile_h = wi.height; b.tile_w = wi.width; b.tile_left = wi.left; b.tile_top = wi.top;
        b.img_h = H; b.img_w = W;
        b.full[0] = t.full_rect.left; b.full[1] = t.full_rect.top; b.full[2] = t.full_rect.right; b.full[3] = t.full_rect.bottom;
        b.unique[0] = t.unique_rect.left; b.unique[1] = t.unique_rect.top; b.unique[2] = t.unique_rect.right; b.unique[3] = t.unique_rect.bottom;
        if (fuse_head && !head_epi) {
            hb.src = layer_source((int)spec.layers.size() - 1, image);   // (scale/shift pointers are stable; the tensors were just written)
            const size_t plane = (size_t)win.height * win.width * head.cin * es;   // sample i of the last hidden tensor
            hb.src.a = static_cast<const char*>(hb.src.a) + (size_t)i * plane;
            if (hb.src.b) hb.src.b = static_cast<const char*>(hb.src.b) + (size_t)i * plane;
            hb.w_tm = w_tm_f32.as<float>() + head.w_off; hb.bias = master.as<float>() + head.b_off;
            hb.blend = b;
            const int tok = prof.begin(stream, "head_blend_fused", 2.0 * head.cin * head.cout * (double)win.height * win.width,
                                       (double)win.height * win.width * (head.cin * 2.0 + K * 8.0));
            launch_head_blend(hb, stream);
            prof.end(stream, tok);
            continue;
        }
        const int tok = prof.begin(bs, "blend_accumulate", 0, (double)K * win.height * win.width * 12);
        launch_blend(b, bs);
        prof.end(bs, tok);
    }
}

// how many tiles with a window of h x w run as one batch: ANH_INFER_TILE_BATCH, or as many (at most kMaxTileBatch) as keep the
// batch's layer tensors under 16 GiB (1024^2 tiles: 8 x 0.4 GB in bf16; a 4096^2 tile alone is 6 GB)
int Engine::tile_batch(int h, int w) const {
    static const int batch_env = getenv("ANH_INFER_TILE_BATCH") ? atoi(getenv("ANH_INFER_TILE_BATCH")) : 0;
    int batch = batch_env;
    if (batch <= 0) {
        double per_px = 0, scale = 1.0;   // storage elements per input pixel over all layer outputs
        for (const anh_layer_desc& L : spec.layers) {
            if (L.stride == 2) scale *= L.type == 0 ? 0.25 : 4.0;
            per_px += L.cout * scale;
        }
        const double bytes = per_px * (double)elem_size(dtype) * (double)h * (double)w;
        batch = (int)std::floor(16.0 * 1024 * 1024 * 1024 / std::max(bytes, 1.0));
    }
    return std::max(1, std::min(kMaxTileBatch, batch));
}

const double* Engine::upload_gains(const double* gains_host) {
    if (!gains_host) return nullptr;
    const int K = spec.cfg.classes;
    gains_dev.reserve((size_t)K * sizeof(double));
    HIP_CHECK(hipMemcpyAsync(gains_dev.p, gains_host, (size_t)K * sizeof(double), hipMemcpyHostToDevice, stream));
    return gains_dev.as<double>();
}

void Engine::infer_device(const uint8_t* d_image, int H, int W, const double* gains_host, const std::vector<anh_tile>& tiles,
                          uint16_t* d_labels, float* d_blended, bool whole_image) {
    ANH_REQUIRE(H >= 1 && W >= 1, "empty image");
    const int K = spec.cfg.classes;
    const int64_t pixels = (int64_t)H * W;
    prof.start_image();
    // The class planes before the blends (annonet_infer.cpp:80-85 allocates them zeroed).  Inside its unique rectangle a tile ASSIGNS
    // (out = in, annonet_infer.cpp:156-161) and only the frame between its full and its unique rectangle is accumulated into, so when the
    // list is the image's complete tiling only those frames need to be zero: 7 % of a 4096^2 image at 1024^2 tiles (14 MB of 201), 3.6 %
    // of a 16384^2 one — round 5; a tile list that is a replica's share leaves pixels no tile of it writes, so that path clears all.
    if (whole_image && tiles.size() > 1) {
        uint64_t key = 1469598103934665603ull;
        auto mix = [&key](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
        mix((uint64_t)H); mix((uint64_t)W); mix((uint64_t)tiles.size());
        for (const anh_tile& t : tiles) for (long v : {t.full_rect.left, t.full_rect.top, t.full_rect.right, t.full_rect.bottom, t.unique_rect.left, t.unique_rect.top, t.unique_rect.right, t.unique_rect.bottom}) mix((uint64_t)v);
        if (key != zero_rects_key || !zero_rects.p) {
            std::vector<anh_rect> frames;
            auto add = [&](long l, long t, long r, long b) {
                l = std::max(l, 0L); t = std::max(t, 0L); r = std::min(r, (long)W - 1); b = std::min(b, (long)H - 1);
                if (l <= r && t <= b) frames.push_back(anh_rect{l, t, r, b});
            };
            for (const anh_tile& t : tiles) {
                const anh_rect &f = t.full_rect, &u = t.unique_rect;
                add(f.left, f.top, f.right, u.top - 1);        // above the unique rectangle
                add(f.left, u.bottom + 1, f.right, f.bottom);  // below
                add(f.left, u.top, u.left - 1, u.bottom);      // left of it
                add(u.right + 1, u.top, f.right, u.bottom);    // right of it
            }
            HIP_CHECK(hipStreamSynchronize(stream));   // (a previous image's clear may still read the old list)
            zero_rects.reserve(std::max<size_t>(frames.size(), 1) * sizeof(anh_rect));
            if (!frames.empty()) HIP_CHECK(hipMemcpy(zero_rects.p, frames.data(), frames.size() * sizeof(anh_rect), hipMemcpyHostToDevice));
            zero_rects_n = (int)frames.size();
            zero_rects_key = key;
        }
        launch_zero_rects(d_blended, K, H, W, zero_rects.as<anh_rect>(), zero_rects_n, stream);
    } else if (!(whole_image && tiles.size() == 1))   // (a single tile assigns every pixel of the image: nothing to clear)
        launch_fill_zero(d_blended, (size_t)K * pixels * 4, stream);
    // consecutive tiles with equal input windows (all of them, on a regular tiling) run as batches
    // — as FEW batches as the cap allows, of equal size (25 tiles at a cap of 8 used to run as 8 + 8 + 8 + 1: every launch has a fixed
    // prologue and tail, and the last batch paid them for one tile)
    for (size_t i = 0; i < tiles.size();) {
        const TileWindow w0 = tile_window(tiles[i], spec.cfg.levels);
        const size_t batch = (size_t)tile_batch(w0.height, w0.width);
        size_t run = 1;
        while (i + run < tiles.size()) {
            const TileWindow wj = tile_window(tiles[i + run], spec.cfg.levels);
            if (wj.height != w0.height || wj.width != w0.width) break;
            ++run;
        }
        const size_t n_batches = (run + batch - 1) / batch, per = (run + n_batches - 1) / n_batches;
        for (size_t done = 0; done < run; done += per)
            infer_tiles(&tiles[i + done], (int)std::min(per, run - done), d_image, H, W, d_blended);
        i += run;
    }
    if (d_labels) argmax_rows(d_blended, H, W, 0, H, gains_host, d_labels);
}

void Engine::argmax_rows(const float* d_blended, int H, int W, int row0, int row1, const double* gains_host, uint16_t* d_labels) {
    const int K = spec.cfg.classes;
    const double* d_gains = upload_gains(gains_host);
    const int tok = prof.begin(stream, "argmax_gain", 0, (double)(row1 - row0) * W * (K * 4.0 + 2.0));
    launch_argmax_range(d_blended, K, (int64_t)H * W, (int64_t)row0 * W, (int64_t)row1 * W, d_gains, d_labels, stream);
    prof.end(stream, tok);
}

}  // namespace anh
