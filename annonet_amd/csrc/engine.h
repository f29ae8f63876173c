// engine.h — owns one net's device state on one GPU and sequences the kernels for
//   * RuntimeNet::Forward and annonet_infer()  (inference: bn folded to affine)
//   * TrainingNet::StartTraining               (forward, loss, backward, SGD)
// Everything is enqueued on one HIP stream; nothing synchronises unless the caller asks for host data.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "spec.h"

namespace anh {

// Per-kernel-class timing with HIP events on the engine's stream (bench.py's roofline line reads this).
class Profiler {
  public:
    struct Entry { std::string name; double total_ms = 0; int64_t launches = 0; double flops = 0, bytes = 0; };
    bool enabled = false;
    std::string filter;  // when non-empty only kernels whose name contains it are timed
    int sample_every = 1;  // time the kernels of every n-th pass only (an event pair costs ~6 us of stream time per launch)
    int64_t pass_index = 0; // advanced by the engine at the start of every forward pass
    // entry names of the most recent pass in host enqueue order, one per begin() (kept whether or not that launch is timed):
    // tools/pmc_traffic.py maps the dispatches of a rocprofv3 trace onto bench.py's per-layer entries with it
    std::vector<std::string> order;
    bool hold_order = false;   // inside annonet_infer(): `order` spans the whole image (every tile batch is a forward pass of its own)
    void start_pass() { ++pass_index; if (enabled && !hold_order) order.clear(); }
    void start_image() { if (enabled) order.clear(); hold_order = true; }
    ~Profiler();
    int begin(hipStream_t s, const char* name, double flops, double bytes);  // returns a token (or -1 when disabled)
    void end(hipStream_t s, int token);
    void collect();  // stream must be idle
    void reset();
    std::vector<Entry> entries;

  private:
    struct Pending { int entry; hipEvent_t start, stop; };
    std::vector<Pending> pending_;
    std::vector<hipEvent_t> pool_;
    hipEvent_t get_event();
};

struct LayerState {
    DevBuf raw;    // raw conv output y (storage dtype), [n][h][w][cout]
    DevBuf dact;   // gradient w.r.t. this layer's post-activation output; becomes dy in place on the unfused path
    DevBuf bn;     // mean, invstd, scale, shift (4*C floats) then var (C doubles)
    float* mean = nullptr; float* invstd = nullptr; float* scale = nullptr; float* shift = nullptr; double* var = nullptr;
    float* coef = nullptr;  // bn backward coefficients [3][C] of this layer (its own slot: two streams read them)
    int n = 0, h_in = 0, w_in = 0, h = 0, w = 0;
    bool dact_written = false;
    const void* da_alias = nullptr;   // backward: da of this layer lives in ANOTHER layer's buffer (skip gradient written once); its apply pass reads it from there
    // backward bookkeeping: a layer's da is final once all its consumers' backward-data convs have written it; the conv
    // that writes it last may also leave the dgamma / dbeta partial sums (fused bn backward reduction)
    int consumers = 0, da_writes = 0, fused_bwd_blocks = 0;
    DevBuf bwd_partials;
    double running_updates = 0;  // dlib bn_: number of running-stat updates so far (capped by the window)
    long long* acc = nullptr;    // table mode (bnacc.h): this layer's accumulator table inside Engine::bn_acc
    double fold_af = 1.0, fold_unbias = 1.0; bool fold_running = false;   // this step's running-statistics update, applied by the fold job
};

class Engine {
  public:
    Engine(const anh_net_config& cfg, bool training);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    Spec spec;
    DType dtype;
    bool training;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    // filter gradients run on a second stream, concurrently with the backward-data / bn chain (both only read dy)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_dy_ready = nullptr, ev_aux_done = nullptr;
    // Data-parallel hosts start the gradient all-reduce before backward has finished: the filter gradients come out last layer
    // first, so once layer kEarlyLayer's is reduced every gradient element from that layer's first parameter to the end of the
    // bucket (deeper layers, head, loss slot) is final.  ev_early_grads fires there (on the stream that ran that filter gradient).
    static constexpr int kEarlyLayer = 2;
    hipEvent_t ev_early_grads = nullptr;
    hipStream_t early_stream = nullptr;       // where the in-library exchange reduces the early part (anh_set_devices), made on first use
    hipEvent_t ev_early_reduced = nullptr;
    hipStream_t early_reduce_stream();
    int64_t early_grad_first() const;   // first bucket element covered by ev_early_grads; n_params + 1 when there is no early part
    bool concurrent_wgrad = true;
    Profiler prof;
    unsigned long bn_window = 100;           // SetAllBatchNormalizationRunningStatsWindowSizes
    bool update_running_in_forward = true;   // training forwards update the running statistics (dlib bn_ semantics)

    // ---- parameters ----
    void set_params(const float* params, const float* running);  // host canonical blobs
    void get_params(float* params, float* running);              // synchronises
    void get_grads_canonical(float* out);
    void get_momentum(float* out);
    void set_momentum(const float* in);
    std::vector<double> get_running_updates() const;   // dlib bn_ num_updates, one per bn layer (trainer state file)
    void set_running_updates(const std::vector<double>& v);
    void random_init(uint64_t seed);
    float* grad_bucket() { return grad.as<float>(); }  // n_params + 1 floats (last = loss)

    // ---- forward ----
    // image: SRC_IMAGE source (device u8).  Inference writes fp32 NCHW logits to d_out_nchw.
    void forward_inference(const Src& image, int n, int h, int w, float* d_out_nchw);
    void forward_training(const Src& image, int n, int h, int w);
    // ---- training ----
    void backward(const uint16_t* d_labels, const float* d_weights, double loss_scale_n);
    // loss_post / loss_tag: SgdArgs::loss_post (the update kernel posts the step's loss to that pinned host word), or null
    void apply_update(double lr, double weight_decay, double momentum, double grad_scale, unsigned long bn_window, unsigned long long* loss_post = nullptr, unsigned int loss_tag = 0);
    double read_loss();  // synchronises
    int read_error_flag_and_clear();
    void layer_tensor(int layer, int which, float* out_host, int64_t capacity, int dims[4]);

    // ---- tiled inference: annonet_infer.cpp:42-214 with image, blended planes and labels resident in HBM ----
    int head_epi_layer = -1; float* head_epi_out = nullptr;   // set around the conv launches of one inference batch (infer_tiles)
    static constexpr int kMaxTileBatch = 16;   // Src::img_win holds 16 windows (4096^2 image, 25 tiles of 1024^2: batches of 8+8+8+1 / 7+6+6+6 / 9+8+8 / 13+12 / 25: 3,838 / 3,865 / 3,917 / 3,960 / 3,938 Mpx/s)
    void infer_tile(const anh_tile& t, const uint8_t* d_image, int H, int W, float* d_blended);
    void infer_tiles(const anh_tile* ts, int count, const uint8_t* d_image, int H, int W, float* d_blended);
    bool infer_post = false;   // this inference pass stores post-activation tensors (choose_inference_form)
    ConvArgs forward_conv_args(int li, const Src& image, bool training_pass, float* d_out_nchw) const;
    void choose_inference_form(const Src& image);
    int tile_batch(int h, int w) const;
    const double* upload_gains(const double* gains_host);   // -> device pointer (or nullptr)
    void argmax_rows(const float* d_blended, int H, int W, int row0, int row1, const double* gains_host, uint16_t* d_labels);   // find_label over rows [row0, row1)
    // whole_image: `tiles` is the image's COMPLETE tiling (then only the frames the blends accumulate into are cleared first)
    void infer_device(const uint8_t* d_image, int H, int W, const double* gains_host, const std::vector<anh_tile>& tiles,
                      uint16_t* d_labels, float* d_blended, bool whole_image = false);
    DevBuf zero_rects; int zero_rects_n = 0; uint64_t zero_rects_key = 0;   // the frames of the last whole-image tiling

    // ANH_STEP_GRAPH=1: the backward pass behind the head replayed as a captured HIP graph (Engine::backward)
    struct StepGraph { uint64_t key; int eager_runs; hipGraphExec_t exec; };
    std::vector<StepGraph> step_graphs;
    long step_graph_captures = 0, step_graph_launches = 0;
    static bool step_graph_enabled();
    void synchronize();
    bool bounded_waits = false;   // one of several replicas behind a handle: host waits take the deadline of common.h (wait_stream)
    void set_stream(hipStream_t s);

    // scratch exposed to the C ABI layer (host-pointer entry points stage through these)
    DevBuf stage_image, stage_labels, stage_weights, stage_out, stage_blended, stage_result;
    std::vector<float> host_out;

  private:
    void plan_dims(int n, int h, int w);
    void plan_dims_unguarded(int n, int h, int w);
    // Table mode (bnacc.h, default in bf16 training when every layer's kernels support it; ANH_BN_TABLES=0 switches it off): the bn
    // sums of every layer are added to accumulator tables by the kernels that see the values and folded by their consumers, so the
    // 18 finalize launches of a step disappear from the critical stream.
    bool tables = false, fold_pending = false;
    bool tables_clean = false;   // the update kernel of the previous step cleared the tables (Engine::apply_update)
    DevBuf bn_acc;
    size_t bn_acc_bytes = 0;
    BnFoldJobs fold_jobs;
    bool choose_table_mode(const Src& image);
    BnTable table_of(int li) const;
    BnBwdFinish finish_of(int li);
    void build_fold_jobs();
    bool head_is_fused() const;
    Src layer_source(int li, const Src& image, bool fold_tables = false) const;
    void run_conv_forward(int li, const Src& image, bool training_pass, float* d_out_nchw);
    void refresh_compute_weights();
    void fold_running_stats();  // inference: scale/shift from running stats, computed on the host
    void ensure_training_buffers();
    void conv_dispatch(const ConvArgs& a, const char* tag, double flops, double bytes);
    void wgrad_dispatch(WgradArgs& a, const char* tag, double flops, double bytes, hipStream_t on, DevBuf& scratch);

    std::vector<LayerState> ls;
    std::vector<ParamSegment> segments_host;
    DevBuf segments;
    DevBuf master, momentum, grad, w_tm_f32, w_km_f32, w_tm_bf16, w_km_bf16, running;
    DevBuf bn_partials, wgrad_partials, wgrad_partials_main, loss_partials, coef, logits, dlogits, scalars, gains_dev, tile_out;
    double* loss_dev = nullptr;
    int* error_flag = nullptr;
    std::vector<float> host_params, host_running;  // mirror kept for inference folding / serialization
    Src last_image{};
    int last_n = 0, last_h = 0, last_w = 0;
    bool have_forward = false;
};

}  // namespace anh
