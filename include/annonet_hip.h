/*
 * annonet_hip.h — C ABI of libannonet_hip.so, the MI355X (gfx950) implementation of annonet's hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * The C++ shims include/NetPimpl.h and include/tiling/tiling.h wrap it back into the class surface
 * annonet's host code uses (dlib-dnn-pimpl-wrapper/NetPimpl.h and tiling/tiling.h, both absent from the
 * reference snapshot and therefore known only from their call sites, cited per function below).
 * All file:line citations are relative to the reference tree.
 *
 * Conventions
 *   - every function returns anh_status (0 = ok) unless stated; on failure anh_last_error() (thread-local)
 *     holds the message.  Nothing aborts: device out-of-memory is ANH_ERR_OOM, so a host that throws on
 *     failure exits with a positive code (find_max_mini-batch_size.cmd:49-53 relies on that).
 *   - images are u8, row-major HWC, channel order R,G,B (dlib::matrix<rgb_pixel>) or one channel.
 *   - network outputs are fp32 NCHW (annonet_infer.cpp:26-30).
 *   - "host" entry points take host pointers and copy; "_device" entry points take pointers that are
 *     already resident in this GPU's HBM and enqueue on the handle's stream without synchronising.
 *   - one thread drives a handle at a time (annonet_train_main.cpp:583-614, annonet_infer_main.cpp:446-494).
 */
#ifndef ANNONET_HIP_H
#define ANNONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ANH_OK = 0,
    ANH_ERR_INVALID = 1,  /* bad argument / precondition (DLIB_CASSERT analogue, annonet_infer.cpp:90-94) */
    ANH_ERR_OOM = 2,      /* device or host allocation failed */
    ANH_ERR_DEVICE = 3,   /* HIP runtime error, no usable GPU */
    ANH_ERR_IO = 4,       /* malformed blob / file error */
    ANH_ERR_INTERNAL = 5
} anh_status;

typedef enum {
    ANH_FP32 = 0, /* parity mode: fp32 storage, fp32 k-ordered fmaf chains (bit-exact inference vs the oracle) */
    ANH_BF16 = 1  /* throughput mode: bf16 storage + bf16 MFMA, fp32 accumulate, fp32 master weights */
} anh_precision;

#define ANH_LABEL_IGNORE 65535 /* dlib::loss_multiclass_log_per_pixel_::label_to_ignore (annonet.cpp:25) */

/* Build-time knobs of the reference made run-time: DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT (appveyor.yml:7-20),
 * DLIB_DNN_PIMPL_WRAPPER_GRAYSCALE_INPUT (annonet_train_main.cpp:93-100), SetNetWidth (annonet_train_main.cpp:402),
 * SetClassCount (:405). */
typedef struct {
    int levels;          /* 0..3 down/up-sampling levels */
    int in_channels;     /* 3 = RGB, 1 = grayscale */
    int classes;         /* K */
    double width_scaler; /* --net-width-scaler */
    int min_filters;     /* --net-width-min-filter-count */
    int precision;       /* anh_precision */
} anh_net_config;

/* One layer of the net (the stand-in for NetStructure.h). Offsets index the canonical
 * parameter blob: con filters [cout][cin][ky][kx], cont filters [cin][cout][ky][kx], bias[cout],
 * gamma[cout], beta[cout]; running-stats blob: mean[cout], var[cout] per bn layer. */
typedef struct {
    int type; /* 0 = con, 1 = cont */
    int k, stride, pad, cin, cout;
    int in_a, in_b; /* producing layer; -1 = image; in_b = -2: no skip input */
    int has_bn, has_bias;
    int64_t w_off, b_off, g_off, beta_off, rs_off;
} anh_layer_desc;

typedef struct { uint16_t label; float weight; } anh_wlabel; /* dlib weighted_label (annonet_train.h:80) */
typedef struct { long left, top, right, bottom; } anh_rect;   /* dlib::rectangle, inclusive */
typedef struct { anh_rect full_rect, unique_rect; } anh_tile; /* tiling::dlib_tile (annonet_infer.cpp:46-47,118,138) */
typedef struct { int max_tile_width, max_tile_height, overlap_x, overlap_y; } anh_tiling_params; /* tiling::parameters (annonet_infer_main.cpp:423-427) */

typedef struct anh_runtime anh_runtime; /* NetPimpl::RuntimeNet */
typedef struct anh_trainer anh_trainer; /* NetPimpl::TrainingNet */

const char* anh_last_error(void);
void anh_free(void* p); /* releases buffers this library returned (serialize, get_tiles) */

/* dlib::cuda::set_device (annonet_train_main.cpp:392-394): device used by handles created afterwards on this thread */
int anh_set_device(int device);
int anh_device_count(void); /* number of visible GPUs; 0 when none (never fails) */
/* One process, several GPUs (the reference has only --primary-cuda-device, annonet_train_main.cpp:307,392-394):
 * handles created afterwards ON THIS THREAD hold one replica per listed device.
 *   TrainingNet::StartTraining (anh_trainer_step) splits the mini-batch along N over the replicas (annonet_train_main.cpp:583-614
 *     stays unchanged), sums the flat gradient buckets with ONE all-reduce (RCCL over xGMI) and applies the identical update;
 *     the loss scale uses the whole batch, batch-norm statistics are per replica;
 *   annonet_infer (anh_infer) splits the tile list into contiguous chunks, exchanges the plane sums where tiles of different
 *     replicas overlap (one RCCL all-reduce) and returns ONE host label map.
 * n = 0 returns to "the thread's current device".  A list that repeats a device is a rehearsal on fewer GPUs (tests): the
 * exchange then runs as a fixed-order sum on the first replica's stream instead of RCCL. */
int anh_set_devices(const int* devices, int n);
int anh_handle_replicas(void* handle, int is_trainer); /* number of replicas (devices) a handle drives */
/* What the exchange step of data-parallel StartTraining (anh_trainer_step on a handle with several replicas) costs: the HOST time of a
 * call (the reference's loop is one thread: /root/reference/annonet_train_main.cpp:583-614 — what it spends inside StartTraining bounds
 * the step rate whatever the GPUs do), and, on every 8th step (ANH_EXCHANGE_SAMPLE), the device time of the all-reduce's two parts on
 * replica 0 (tail = the bulk of the bucket, reduced on a side stream while backward still runs; head = the first layers, on the main
 * stream).  The replicas of a handle are driven by persistent worker threads; worker_calls counts their wake-ups.
 * uses_rccl names the transport of the exchange: 1 = RCCL over xGMI; 2 = peer copies between distinct devices (taken by itself, in the
 * same process, when the RCCL communicators cannot be created, or with ANH_COLLECTIVE_TRANSPORT=peer); 0 = a device list that repeats
 * a device (rehearsal on fewer GPUs).  One host thread per handle; host-side waits of a handle with several replicas give up after
 * ANH_REPLICA_TIMEOUT_S seconds (default 180) with ANH_ERR_DEVICE instead of blocking for ever. */
typedef struct {
    int replicas, early_reduce, uses_rccl, rccl_version;
    int64_t steps, samples, worker_calls, bucket_bytes;
    double host_us_mean, host_us_last, host_wait_us_mean; /* wall time of a call; of which blocked on the GPU (staging set of step k-2 still uploading) */
    double allreduce_tail_us_mean, allreduce_head_us_mean, allreduce_tail_us_last, allreduce_head_us_last;
} anh_exchange_stats;
int anh_trainer_exchange_stats(anh_trainer* h, anh_exchange_stats* out);
void anh_trainer_reset_exchange_stats(anh_trainer* h);
/* Per-rank delivery of a sharded annonet_infer()'s label map (the reference's writer threads consume HOST label maps,
 * /root/reference/annonet_infer_main.cpp:403-419): every rank copies the rectangles it answers for from its device-resident map straight
 * into ONE host map that all ranks of the job map (POSIX shared memory), over its own PCIe link — nothing funnels through rank 0.
 * anh_host_register page-locks host memory the caller owns (the shared map) so the copies run asynchronously; anh_labels_rect_to_host
 * enqueues the copy of rows [top, bottom] x columns [left, right] (inclusive) of a [height x width] u16 map on `stream`. */
int anh_host_register(void* p, size_t bytes);
int anh_host_unregister(void* p);
int anh_labels_rect_to_host(const uint16_t* d_labels, uint16_t* h_labels, int width, int height, int left, int top, int right, int bottom, void* stream);
/* the host logic of the two splits, exported for tests: replica `rank` of `world` takes units [*lo, *hi) of n */
int anh_shard_range(int64_t n, int world, int rank, int64_t* lo, int64_t* hi);

/* ---- static dimension maths ---- */
/* TrainingNet::GetRequiredInputDimension() (annonet_train_main.cpp:376,441; annonet_infer_main.cpp:421): receptive-field side */
int anh_required_input_dim(const anh_net_config* cfg);
/* RuntimeNet::GetRecommendedInputDimension(int) (annonet_infer.cpp:49-50; annonet_train_main.cpp:382): smallest valid side >= n */
int anh_recommended_input_dim(int levels, int n);

/* ---- net description ---- */
int anh_net_layer_count(const anh_net_config* cfg);
int anh_net_layer(const anh_net_config* cfg, int index, anh_layer_desc* out);
int64_t anh_net_param_count(const anh_net_config* cfg);
int64_t anh_net_running_count(const anh_net_config* cfg);

/* ---- RuntimeNet ---- */
int anh_runtime_create(const anh_net_config* cfg, anh_runtime** out); /* default-constructed RuntimeNet (annonet_infer_main.cpp:347) */
void anh_runtime_destroy(anh_runtime* h);
int anh_runtime_config(const anh_runtime* h, anh_net_config* out);
int anh_runtime_set_params(anh_runtime* h, const float* params, int64_t n_params, const float* running, int64_t n_running);
int anh_runtime_get_params(const anh_runtime* h, float* params, int64_t n_params, float* running, int64_t n_running);
/* RuntimeNet::Serialize / Deserialize (annonet_train_main.cpp:558-561; annonet_infer_main.cpp:347-351): opaque blob */
int anh_runtime_serialize(const anh_runtime* h, void** blob, size_t* size);
int anh_runtime_deserialize(const void* blob, size_t size, int precision, anh_runtime** out);
/* RuntimeNet::Forward(const input_type&) (annonet_infer.cpp:77): returns a library-owned fp32 NCHW tensor,
 * valid until the next call on this handle; k = classes, nr x nc = input tile size (annonet_infer.cpp:80-100). */
int anh_runtime_forward(anh_runtime* h, const uint8_t* image_hwc, int n, int height, int width,
                        const float** out_nchw, int* k, int* nr, int* nc);
int anh_runtime_forward_device(anh_runtime* h, const uint8_t* d_image_hwc, int n, int height, int width, float* d_out_nchw);
/* annonet_infer() (annonet_infer.h:34-42, annonet_infer.cpp:32-240) in one call: tile, clamp-pad, forward, blend,
 * argmax(+gain), optional detection-level blob filter.  gains / detection_levels: K doubles or NULL.
 * tiling: NULL = one tile (tiling::parameters() default argument, annonet_infer.h:41).
 * blended_out (optional, host): K planes H*W fp32 = annonet_infer_temp::blended_output (annonet_infer.h:31). */
int anh_infer(anh_runtime* h, const uint8_t* image_hwc, int height, int width,
              const double* gains, const double* detection_levels, const anh_tiling_params* tiling,
              uint16_t* result_labels, float* blended_out);
/* same with the image and the label map resident in HBM; explicit tile list (e.g. one rank's shard) when tiles != NULL */
int anh_infer_device(anh_runtime* h, const uint8_t* d_image_hwc, int height, int width,
                     const double* gains, const anh_tiling_params* tiling,
                     const anh_tile* tiles, size_t n_tiles, uint16_t* d_result_labels, float* d_blended);
/* label rows [row0, row1) of device-resident blended planes (find_label, annonet_infer.cpp:170-185): the second half of a
   sharded annonet_infer(), run after the ranks have exchanged the plane sums of their overlapping tiles */
int anh_argmax_device(anh_runtime* h, const float* d_blended, int height, int width, int row0, int row1, const double* gains, uint16_t* d_result);
/* A handle launches on ONE stream (its own, non-blocking, unless set).  NULL returns the handle to a stream of its own — it does
   NOT mean the legacy default stream.  A host that mixes its own GPU work with the handle's (torch.distributed's all-reduce of the
   gradient bucket, torch kernels on the blended planes) takes the handle's stream with _get_stream and enqueues that work on it. */
int anh_runtime_set_stream(anh_runtime* h, void* hip_stream);
int anh_runtime_get_stream(anh_runtime* h, void** hip_stream);
/* bf16 inference has two forms: when every layer runs on a persistent MFMA kernel, each layer stores its post-activation output
 * relu(bn(y)) (one rounding of the fp32 accumulator); otherwise layers store the raw conv output and consumers re-apply bn + relu.
 * *yes = the form of the handle's LAST inference pass (parity tests pick the matching CPU restatement; ANH_INFER_POST_ACT=0 forces
 * the second form). */
int anh_runtime_stores_activations(anh_runtime* h, int* yes);
int anh_runtime_synchronize(anh_runtime* h);

/* ---- TrainingNet (annonet_train_main.cpp:396-410) ---- */
int anh_trainer_create(anh_trainer** out);                                      /* TrainingNet ctor */
void anh_trainer_destroy(anh_trainer* h);
int anh_trainer_set_net_width(anh_trainer* h, double scaler, int min_filters); /* SetNetWidth */
int anh_trainer_set_class_count(anh_trainer* h, size_t classes);               /* SetClassCount */
int anh_trainer_set_levels(anh_trainer* h, int levels);                        /* LEVEL_COUNT */
int anh_trainer_set_input_channels(anh_trainer* h, int channels);              /* GRAYSCALE_INPUT */
int anh_trainer_set_precision(anh_trainer* h, int precision);
int anh_trainer_set_seed(anh_trainer* h, uint64_t seed);
int anh_trainer_initialize(anh_trainer* h);                                    /* Initialize(): builds the net, random init */
int anh_trainer_set_learning_rate(anh_trainer* h, double lr);                  /* SetLearningRate */
int anh_trainer_set_learning_rate_shrink_factor(anh_trainer* h, double f);     /* SetLearningRateShrinkFactor */
int anh_trainer_set_iterations_without_progress_threshold(anh_trainer* h, unsigned long n);
int anh_trainer_set_previous_loss_values_dump_amount(anh_trainer* h, unsigned long n);
int anh_trainer_set_all_bn_running_stats_window_sizes(anh_trainer* h, unsigned long n);
int anh_trainer_set_synchronization_file(anh_trainer* h, const char* path, double seconds); /* SetSynchronizationFile */
int anh_trainer_be_verbose(anh_trainer* h);                                    /* BeVerbose */
int anh_trainer_set_sgd(anh_trainer* h, double weight_decay, double momentum); /* dlib sgd defaults 0.0005 / 0.9 */
double anh_trainer_get_learning_rate(const anh_trainer* h);                    /* GetLearningRate (annonet_train_main.cpp:570) */
double anh_trainer_get_last_loss(anh_trainer* h);                              /* synchronises */
unsigned long anh_trainer_get_step_count(const anh_trainer* h);
int anh_trainer_config(const anh_trainer* h, anh_net_config* out);
/* StartTraining(samples, labels) (annonet_train_main.cpp:609): one optimiser step; inputs are consumed (copied to
 * HBM) before this returns (annonet_train_main.cpp:585-586). images[i]: height*width*channels u8; labels[i]: height*width. */
int anh_trainer_step(anh_trainer* h, const uint8_t* const* images, const anh_wlabel* const* labels, int n, int height, int width);
/* the same step on inputs resident in HBM, split so that data-parallel ranks can all-reduce the gradient
 * bucket in between: forward_backward -> (RCCL all-reduce of grad_buffer) -> apply_update.
 * loss_scale_n: the N of the loss scale 1/(N*nr*nc), i.e. the GLOBAL batch. */
int anh_trainer_forward_backward_device(anh_trainer* h, const uint8_t* d_images, const uint16_t* d_labels, const float* d_weights,
                                        int n, int height, int width, double loss_scale_n);
int anh_trainer_apply_update(anh_trainer* h, double grad_scale);
int anh_trainer_grad_buffer(anh_trainer* h, void** d_ptr, int64_t* count); /* flat fp32 gradients (+1 trailing slot: loss) */
int anh_trainer_get_params(anh_trainer* h, float* params, int64_t n_params, float* running, int64_t n_running);
int anh_trainer_set_params(anh_trainer* h, const float* params, int64_t n_params, const float* running, int64_t n_running);
int anh_trainer_get_grads(anh_trainer* h, float* grads_canonical, int64_t n_params);
int anh_trainer_replica_params(anh_trainer* h, int replica, float* params, int64_t n_params); /* anh_set_devices: every replica must hold the same weights */
int anh_trainer_get_momentum(anh_trainer* h, float* momentum, int64_t n_params);
int anh_trainer_set_momentum(anh_trainer* h, const float* momentum, int64_t n_params);
/* GetRuntimeNet() (annonet_train_main.cpp:558): independent by-value snapshot; quiesces the device first */
int anh_trainer_snapshot_runtime(anh_trainer* h, int precision, anh_runtime** out);
int anh_trainer_save_state(anh_trainer* h, const char* path); /* trainer synchronization file */
int anh_trainer_load_state(anh_trainer* h, const char* path);
int anh_trainer_set_stream(anh_trainer* h, void* hip_stream);
int anh_trainer_get_stream(anh_trainer* h, void** hip_stream);
/* Overlapping the gradient all-reduce with the end of backward (data-parallel hosts; annonet_train_main.cpp:583-614 hands the
 * mini-batch to dlib's multi-GPU trainer, whose averaging this replaces): after forward_backward the bucket elements
 * [*first, n_params + 1) — every layer but the first two, the head and the loss slot — become final while the last
 * backward-data convs still run.  wait_early_grads makes `hip_stream` (a stream of the caller's) wait for exactly that point;
 * the elements [0, *first) are final when the handle's own stream has drained.  *first > n_params: no early part. */
int anh_trainer_early_grads(anh_trainer* h, int64_t* first);
/* ANH_STEP_GRAPH=1 (off by default): the launches of a training step behind the fused head — bn backward passes, backward-data convs and,
 * on the second stream, the filter gradients with their hand-over events — are captured once per (shape, buffers, input pointer) and
 * replayed with one hipGraphLaunch per StartTraining (/root/reference/annonet_train_main.cpp:609).  captures / launches: how often
 * replica 0 of the handle instantiated / replayed a graph (0 / 0 when the switch is off or the profiler is on). */
int anh_trainer_step_graph_stats(anh_trainer* h, int64_t* captures, int64_t* launches);
int anh_trainer_wait_early_grads(anh_trainer* h, void* hip_stream);
int anh_trainer_synchronize(anh_trainer* h);
/* debugging / parity taps: raw conv output (which=0) or gradient w.r.t. the layer's activation (which=1), as fp32 NHWC */
int anh_trainer_layer_tensor(anh_trainer* h, int layer, int which, float* out, int64_t capacity, int dims4[4]);

/* ---- per-kernel timing (HIP events on the handle's stream), for bench.py's roofline line ---- */
int anh_profile_enable(void* handle, int is_trainer, int enable);
int anh_profile_set_filter(void* handle, int is_trainer, const char* substring); /* NULL or "" = time every kernel */
int anh_profile_set_sampling(void* handle, int is_trainer, int every);         /* time every n-th forward(+backward) pass only */
int anh_profile_reset(void* handle, int is_trainer);
int anh_profile_count(void* handle, int is_trainer);
int anh_profile_entry(void* handle, int is_trainer, int index, char* name, size_t name_cap,
                      double* total_ms, int64_t* launches, double* flops, double* bytes);
/* entry names of the most recent pass (forward [+ backward + update]) in host enqueue order, one per kernel-class launch,
 * newline-separated; *needed = bytes incl. the terminator.  tools/pmc_traffic.py maps rocprofv3 dispatches onto entries with it. */
int anh_profile_launch_order(void* handle, int is_trainer, char* buf, size_t cap, size_t* needed);

/* ---- single-layer ops on host tensors (kernel-level parity tests and micro-benchmarks) ----
 * The same kernels the net runs, on caller-provided data.  Tensors are fp32 NHWC on the host; in ANH_BF16 they are
 * rounded to bf16 on upload and the result is rounded to bf16 storage before it comes back (filter gradients stay fp32).
 * An input is a producer's raw conv output plus its folded batch-norm: value = relu(x*scale+shift); scale == NULL means
 * "use x as it is".  A second input, when given, is added (skip connection). */
typedef struct { int type, k, stride, pad, cin, cout; } anh_conv_desc; /* type 0 = con, 1 = cont */
typedef struct { const float* x; const float* scale; const float* shift; } anh_op_input;
int anh_op_conv_forward(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                        const float* filters_canonical, const float* bias, float* y_nhwc, int* used_mfma);
int anh_op_conv_backward_data(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const float* dy_nhwc,
                              const float* filters_canonical, float* dx_nhwc, int* used_mfma);
int anh_op_conv_backward_filter(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                                const float* dy_nhwc, float* dw_canonical, int* used_mfma);
/* The same ops with the epilogues / prologues the training step fuses into them (kernel-level parity tests):
 *  _forward_stats      also returns the batch-norm statistics of the STORED output: sums[2*cout] = per channel (sum y, sum y*y);
 *                      *fused = 1 when the conv kernel produced them itself (else the separate statistics kernel ran).
 *  _backward_data_bn   dx (+= dx_init when given: the skip-gradient accumulation) and the bn + relu backward sums of the layer
 *                      that receives dx: y_prev = its raw output, (scale, shift, mean, invstd)[cin];
 *                      sums[2*cin] = per channel (sum dz*xhat, sum dz), dz = (y*scale+shift > 0) ? dx : 0, xhat = (y-mean)*invstd.
 *  bn_dy               describes a gradient that is not materialised: dy = coef0*(dz - coef1 - xhat*coef2) from (da, y) */
typedef struct { const float* da; const float* y; const float* scale; const float* shift; const float* mean; const float* invstd; const float* coef; } anh_op_bn_dy;
int anh_op_conv_forward_stats(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const anh_op_input* a, const anh_op_input* b,
                              const float* filters_canonical, float* y_nhwc, double* sums, int* fused);
int anh_op_conv_backward_data_bn(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const float* dy_nhwc,
                                 const float* filters_canonical, const float* dx_init, const float* y_prev, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, float* dx_nhwc, double* sums, int* fused);
int anh_op_conv_backward_filter_bn(int precision, const anh_conv_desc* d, int n, int h_in, int w_in, const uint8_t* image_u8,
                                   const anh_op_bn_dy* dy, float* dw_canonical, int* computed_dy_in_kernel);

/* ---- host logic ---- */
/* tiling::get_tiles(width, height, params) (annonet_infer.cpp:42): *tiles is malloc'd, release with anh_free */
int anh_get_tiles(int width, int height, const anh_tiling_params* params, anh_tile** tiles, size_t* count);
/* the rectangles (inclusive, clipped to the image) in which tiles owned by different replicas overlap when the tile list is split
   into `world` contiguous chunks: the pixels whose plane sums the replicas exchange.  *rects is malloc'd, release with anh_free */
int anh_cross_replica_overlaps(const anh_tile* tiles, size_t n_tiles, int world, int width, int height, anh_rect** rects, size_t* count);
/* set_weights() (annonet_train.h:20-83) */
int anh_set_weights(const uint16_t* labels, int nr, int nc, double class_weight, double image_weight, anh_wlabel* out);
/* random_rect_containing_point() (annonet_train.h:85-105); the two 32-bit draws of dlib::rand are passed in */
int anh_random_rect_containing_point(uint32_t draw_x, uint32_t draw_y, long px, long py, long width, long height, anh_rect* out);
/* outpaint() (annonet.h:74-120), in place on a u8 image with `channels` interleaved channels */
int anh_outpaint(uint8_t* image, int nr, int nc, int channels, const anh_rect* inside);
/* ---- training crops cut on the device ----
   The reference cuts every training crop on the CPU (randomly_crop_image, annonet_train_main.cpp:110-232, in loader threads)
   and hands StartTraining host matrices.  Here the full images of the dataset live in HBM (288 GB) and a mini-batch of
   crops is produced there: extract_image_chip at scale 1 + outpaint (= clamp-to-edge crop, :160-176), labels set to
   "ignore" outside the image (:150-158), set_weights (:178), flips (:183-194) and the multiplicative brightness change
   (:196-216), for further_downscaling_factor = 1.  The random draws stay with the caller (dlib::rand is the host's): a
   crop is described by its rectangle (random_rect_containing_point), the two flip decisions and the brightness factor.
   Further downscaling, add_random_noise and apply_random_color_offset are covered with the host's draws passed in (see anh_crop_spec);
   dlib's own routines are absent from the snapshot, so their arithmetic is restated [UPSTREAM-UNVERIFIED]. */
typedef struct anh_dataset anh_dataset;
typedef struct {
    int image; long left, top; int flip_left_right, flip_upside_down;
    double brightness_change;          /* multiplicative (:196-216); 1 = none */
    double further_downscaling_factor; /* :124-127,160-171: the rectangle is round(dim * factor) wide and is resized to dim (bilinear image,
                                          nearest-neighbour labels, dlib::resize_image's corner-aligned grid); 0 or 1 = none */
    int noise_level;                   /* add_random_noise (:73-105): uniform integer in [-level, level] per channel value; 0 = none.  The draws
                                          come from a counter-based generator keyed by (noise_seed, position), not from dlib::rand's sequence */
    uint64_t noise_seed;
    int color_offset[3];               /* apply_random_color_offset (:226-231): per-channel offsets drawn by the host (RGB nets); 0 = none */
} anh_crop_spec;
int anh_dataset_create(int channels, anh_dataset** out);
void anh_dataset_destroy(anh_dataset* d);
/* uploads one full image (u8, HWC, `channels` interleaved) and its label image (u16) and keeps them resident; *index = its number */
int anh_dataset_add(anh_dataset* d, const uint8_t* image_hwc, const uint16_t* labels, int height, int width, int* index);
/* frees the image with that index (after the crops in flight that read it): the index becomes invalid (a crop spec naming it is
   ANH_ERR_INVALID) and is handed out again by a later anh_dataset_add.  With anh_dataset_resident_bytes a host bounds the HBM the
   resident set may take (the reference bounds its decoded images by --cached-image-count, annonet_train_main.cpp:301,504-518). */
int anh_dataset_remove(anh_dataset* d, int index);
int anh_dataset_resident_bytes(const anh_dataset* d, uint64_t* bytes);
/* n crops of dim x dim into host arrays: images n*dim*dim*channels u8, weighted labels n*dim*dim (tests, non-resident hosts) */
int anh_dataset_crop_batch(anh_dataset* d, const anh_crop_spec* specs, int n, int dim, int classes, double class_weight, double image_weight,
                           uint8_t* images, anh_wlabel* labels);
/* StartTraining on n crops cut on the device: the mini-batch never exists on the host */
int anh_trainer_step_crops(anh_trainer* h, anh_dataset* d, const anh_crop_spec* specs, int n, int dim, double class_weight, double image_weight);

/* ignore_large_nonzero_regions (annonet_train_main.cpp:434-502), in place on a u16 label image: 8-connected blobs of equal
   non-zero, non-ignored label larger than by_area * rf^2 pixels, wider than by_width * rf or taller than by_height * rf
   become ANH_LABEL_IGNORE (infinity = that test off, as the CLI defaults :339-341).  *ignored = pixels relabelled. */
int anh_ignore_large_nonzero_regions(uint16_t* labels, int nr, int nc, double by_area, double by_width, double by_height,
                                     int receptive_field_side, int64_t* ignored);
/* annonet.dnn = dlib serialize framing of (string anno_classes_json, double downscaling_factor, string serialized RuntimeNet)
   (written annonet_train_main.cpp:557-565, read annonet_infer_main.cpp:340-351).  Both calls work on memory images of the
   file; returned buffers are released with anh_free.  The net blob is what anh_runtime_serialize / _deserialize exchange. */
int anh_dnn_envelope_pack(const char* classes_json, size_t json_size, double downscaling_factor, const void* net_blob, size_t net_size,
                          void** file, size_t* file_size);
int anh_dnn_envelope_unpack(const void* file, size_t file_size, char** classes_json, size_t* json_size, double* downscaling_factor,
                            void** net_blob, size_t* net_size);
/* dlib count_steps_without_decrease, used by the learning-rate schedule */
int64_t anh_count_steps_without_decrease(const double* values, int64_t n, double probability_of_decrease);

#ifdef __cplusplus
}
#endif
#endif /* ANNONET_HIP_H */
