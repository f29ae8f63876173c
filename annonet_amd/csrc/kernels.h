// kernels.h — launch interface of the HIP kernels (gfx950).  Layouts: activations NHWC (channels innermost),
// storage type T = float (ANH_FP32) or __bf16 (ANH_BF16); all arithmetic accumulates in fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace anh {

enum DType { DT_F32 = 0, DT_BF16 = 1 };

// How a kernel reads its input activations.  Post-activation tensors are never materialised: a consumer
// re-applies the producer's folded batch-norm (scale, shift) and the relu while loading the raw conv output.
enum SrcKind {
    SRC_RAW = 0,    // value = a[i]
    SRC_ACT = 1,    // value = relu(a[i]*scale_a[c] + shift_a[c])
    SRC_ACT2 = 2,   // value = relu(a[i]*scale_a[c]+shift_a[c]) + relu(b[i]*scale_b[c]+shift_b[c])   (skip add)
    SRC_IMAGE = 3,  // value = u8 image / 256, read through a clamp-to-edge window (annonet_infer.cpp:68-75)
    // Inference only (running statistics are known before a layer runs, so its epilogue CAN apply its own bn + relu:
    // ConvArgs::out_scale): the producers stored post-activation tensors, a consumer reads a[i] (SRC_RAW) or the skip add:
    SRC_SUM2 = 4    // value = a[i] + b[i]
    // (A fifth kind — the bn + relu BACKWARD of a layer applied by a backward-data conv while staging, in three schedules — was built
    // and measured in rounds 1 and 2 and lost every time; it is gone.  The stem's filter gradient, which has no
    // backward-data conv beside it, still applies that expression while staging: WgradArgs::dy_y.)
};

// Batch-norm accumulator table of the layer a tensor comes from (bnacc.h).  With acc != nullptr a consumer folds that layer's
// (scale, shift) from the table's totals in its own prologue instead of reading arrays a finalize kernel would have had to write.
struct BnTable { const long long* acc = nullptr; const float* gamma = nullptr; const float* beta = nullptr; double pixels = 0; float eps = 1e-4f; int c = 0; };

// Table mode, backward sums: the kernel that adds the LAST sums of a layer's (sum dz*xhat, sum dz) also folds them — its last
// workgroup to finish (a ticket counter in the table) writes dgamma, dbeta and the apply coefficients [k0 | k1 | k2][c], with
// bn_bwd_finalize_kernel's arithmetic.  The sums are already at the memory side (atomics), so no fence stands in the way.
struct BnBwdFinish { long long* acc = nullptr; const float* gamma = nullptr; const float* invstd = nullptr;
                     float* dgamma = nullptr; float* dbeta = nullptr; float* coef = nullptr; double pixels = 0; int c = 0; };

struct Src {
    int kind = SRC_RAW;
    int dtype = DT_F32;
    const void* a = nullptr; const float* a_scale = nullptr; const float* a_shift = nullptr;
    const void* b = nullptr; const float* b_scale = nullptr; const float* b_shift = nullptr;
    BnTable a_tab, b_tab;   // table mode: replaces a_scale / a_shift (b_scale / b_shift) when .acc is set
    // SRC_IMAGE: sample n lives at img + n*img_sample_stride; the net input window starts at (img_left, img_top)
    const uint8_t* img = nullptr;
    int img_h = 0, img_w = 0, img_left = 0, img_top = 0;
    int64_t img_sample_stride = 0;
    // several windows of ONE image as the samples of a batch (annonet_infer() runs tiles of equal size together): sample n
    // starts at (img_win[2n], img_win[2n+1]); img_nwin = 0 means the single window above
    int img_nwin = 0;
    int img_win[32] = {};
    __host__ __device__ int win_left(int n) const { return img_nwin ? img_win[2 * n] : img_left; }
    __host__ __device__ int win_top(int n) const { return img_nwin ? img_win[2 * n + 1] : img_top; }
};

// out[n,oy,ox,co] = sum_{ky,kx,cr} src(n, iy, ix, cr) * w[(ky*k+kx)][cr][co]   (+ bias[co])
//   gather = 0 ("con"):        iy = oy*stride + ky - pad
//   gather = 1 ("transposed"): ty = oy + pad - ky, valid iff ty % stride == 0, iy = ty/stride
// Forward con / backward-data cont use gather 0; forward cont / backward-data con use gather 1.
struct ConvArgs {
    Src src;
    int n = 0, h_in = 0, w_in = 0, c_red = 0;
    int h_out = 0, w_out = 0, c_out = 0;
    int k = 0, stride = 1, pad = 0, gather = 0;
    const float* w_f32 = nullptr;     // [tap][c_red][c_out] fp32
    const void* w_bf16 = nullptr;     // [tap][c_out][c_red] bf16 (MFMA B operand, k contiguous)
    const float* bias = nullptr;
    void* out = nullptr; int out_dtype = DT_F32; int out_accumulate = 0;
    void* out2 = nullptr; int out2_accumulate = 0;  // optional second destination (skip-add gradient)
    int out_nchw = 0;                               // fp32 NCHW destination (boundary layout)
    double* stat_partials = nullptr;                // fused bn statistics (MFMA path), else nullptr
    // table mode (bnacc.h): the sums are added to accumulator tables instead of leaving one partial per workgroup — stat_acc = this
    // layer's table (with stat_partials unset), bnred_acc = the table of the layer `out` belongs to (with bnred_partials unset)
    long long* stat_acc = nullptr; long long* bnred_acc = nullptr;
    BnBwdFinish bnred_finish;   // with bnred_acc
    // bf16 inference on the MFMA path: the epilogue stores relu(acc * out_scale[c] + out_shift[c]) — this layer's folded bn and its
    // relu on the fp32 accumulator — instead of the raw output, so that its consumers stage plain copies (conv_stores_activation())
    const float* out_scale = nullptr; const float* out_shift = nullptr;
    // ... and, for the layer under a 1x1 head (32 channels -> head_k <= 4 classes, bias): head_out set = that activation is NOT stored;
    // the epilogue forms the pixel's logits from it (the fused head/blend kernel's expression and order) and stores them as fp32
    // planes [n][head_k][h_out][w_out] — 4 head_k instead of 64 bytes per pixel written here and read back by the blend
    // (conv_head_in_epilogue_ok() decides; head_w = [ci][k] fp32 holding bf16-rounded values)
    const float* head_w = nullptr; const float* head_bias = nullptr; int head_k = 0; float* head_out = nullptr;
    // Fused bn + relu backward REDUCTION (MFMA path, backward-data convs): `out` receives its final value da of a layer with
    // raw output bnred_y and folded constants; the kernel also writes that layer's dgamma / dbeta partial sums
    // ([channel][sum dz*xhat | sum dz][workgroup], dz = (y*scale+shift > 0) ? da : 0, xhat = (y-mean)*invstd).
    const void* bnred_y = nullptr;
    const float* bnred_scale = nullptr; const float* bnred_shift = nullptr; const float* bnred_mean = nullptr; const float* bnred_invstd = nullptr;
    double* bnred_partials = nullptr;
};

// dw[tap][ci][co] = sum_pixels src(n, iy, ix, ci) * dy[n,oy,ox,co]; same gather convention as ConvArgs.
struct WgradArgs {
    Src src;
    const void* dy = nullptr; int dy_dtype = DT_F32;
    int n = 0, h_in = 0, w_in = 0, c_in = 0;
    int h_out = 0, w_out = 0, c_out = 0;
    int k = 0, stride = 1, pad = 0, gather = 0;
    float* dw = nullptr;        // [tap][c_in][c_out] fp32
    float* partials = nullptr;  // scratch, >= splits * k*k*c_in*c_out floats
    int64_t partials_capacity = 0;
    int* splits_out = nullptr;  // when set: the partial-sum pass is left to the caller (launch_reduce_partials with *splits_out)
    // dy not materialised (wgrad_accepts_bnbwd): `dy` points at da, and dy = bn + relu backward of (da, dy_y) is applied
    // while staging — the expression of the bn_bwd_apply kernels with this layer's constants:
    //   dz = (y*scale + shift > 0) ? da : 0,  dy = coef0 * (dz - coef1 - (y - mean)*invstd * coef2)
    const void* dy_y = nullptr;
    const float* dy_scale = nullptr; const float* dy_shift = nullptr; const float* dy_mean = nullptr; const float* dy_invstd = nullptr;
    const float* dy_coef = nullptr;
};

void launch_conv_generic(const ConvArgs& a, hipStream_t s);
bool conv_f32_mfma_ok(const ConvArgs& a);   // launch_conv_generic runs this conv on the fp32 matrix cores (conv_f32_mfma*), not on the VALU kernels
void launch_wgrad_generic(const WgradArgs& a, hipStream_t s);
int64_t wgrad_generic_scratch_floats(const WgradArgs& a);
bool wgrad_accepts_bnbwd(const WgradArgs& a, DType mode);   // decided on the args with dy_y unset

// batch-norm forward statistics over y[P][C]: partial sums -> mean/var -> folded (scale, shift); optional running update
struct BnFwdArgs {
    const void* y = nullptr; int dtype = DT_F32;
    int64_t pixels = 0; int c = 0;
    const float* gamma = nullptr; const float* beta = nullptr;
    float* mean = nullptr; float* invstd = nullptr; float* scale = nullptr; float* shift = nullptr;
    double* var = nullptr;            // biased batch variance (kept for the running update)
    double* partials = nullptr;       // >= bn_partial_blocks(pixels) * 2 * c doubles
    float eps = 1e-4f;
    // running statistics, updated by the finalize kernel when running_mean != nullptr (dlib bn_: factor 1/(updates+1), unbiased var)
    float* running_mean = nullptr; float* running_var = nullptr; double averaging_factor = 1.0, unbias = 1.0;
};
int bn_partial_blocks(int64_t pixels);
bool bn_table_mode_ok(int c);   // the bn backward kernels of a c-channel layer take accumulator tables (BnBwdArgs::acc)
void launch_bn_forward_stats(const BnFwdArgs& a, hipStream_t s);   // = partials, finalize
int launch_bn_forward_partials(const BnFwdArgs& a, hipStream_t s);
// statistics already written by the conv kernel (ConvArgs::stat_partials, `blocks` partials per channel): finalize only
void launch_bn_forward_finalize(const BnFwdArgs& a, int blocks, hipStream_t s);

// running_mean/var update (dlib bn_: averaging factor 1/(updates+1) up to the window; unbiased variance)
void launch_bn_running_update(const float* mean, const double* var, float* running_mean, float* running_var,
                              int c, double averaging_factor, double unbias, hipStream_t s);

// batch-norm + relu backward.  da: gradient w.r.t. the post-activation output; y: raw conv output.
//   reduce:  dgamma = sum dz*xhat, dbeta = sum dz, with dz = da * (y*scale+shift > 0)
//   apply:   dy = gamma*invstd*(dz - dbeta/P - xhat*dgamma/P), written over da
struct BnBwdArgs {
    void* da = nullptr; const void* y = nullptr; int dtype = DT_F32;
    int64_t pixels = 0; int c = 0;
    const float* gamma = nullptr; const float* mean = nullptr; const float* invstd = nullptr;
    const float* scale = nullptr; const float* shift = nullptr;
    float* dgamma = nullptr; float* dbeta = nullptr;  // destinations in the gradient blob
    double* partials = nullptr;
    float* coef = nullptr;  // scratch 3*c floats
    // apply with the gradient of the 1x1 head NOT materialised: da[p][ch] = round(sum_k head_g[p][k] * head_w_tm[ch][k]) is recomputed
    // per pixel from the head's dlogits (k floats per pixel instead of c storage elements), with the fused head kernel's own expression
    const float* head_g = nullptr; const float* head_w_tm = nullptr; int head_k = 0;
    void* dy_out = nullptr; // apply: destination (default: in place over da)
    int partial_blocks = 0; // finalize: partials per channel when the reduction came from a conv epilogue (0 = bn_partial_blocks(pixels))
    // table mode (bnacc.h): reduce ADDS its sums to the table and its last workgroup leaves dgamma / dbeta / coef (no partials, no finalize)
    long long* acc = nullptr;
    BnBwdFinish finish;   // with acc
};
void launch_bn_backward(const BnBwdArgs& a, hipStream_t s);   // = reduce, finalize, apply
void launch_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s);
void launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s);
void launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s);   // writes BnBwdArgs::dy_out when set, else over da

// loss_multiclass_log_per_pixel_weighted on fp32 NHWC logits [P][K]; writes dlogits in place of nothing (separate buffer),
// the summed loss (double) and the bias gradient.
struct LossArgs {
    const float* logits = nullptr; const uint16_t* labels = nullptr; const float* weights = nullptr;
    float* dlogits = nullptr;
    int64_t pixels = 0; int k = 0;
    double scale = 0;  // 1/(N*nr*nc)
    double* partials = nullptr;  // >= loss_partial_blocks(pixels) * (1+k) doubles
    double* loss_out = nullptr;  // device scalar
    float* loss_out_f32 = nullptr;  // copy in the gradient bucket's trailing slot
    float* dbias = nullptr;
    int* error_flag = nullptr;   // set to 1 when a label is >= k and not the ignore label
};
int loss_partial_blocks(int64_t pixels);
void launch_loss(const LossArgs& a, hipStream_t s);

// SGD with momentum and weight decay over the canonical parameter blob; rewrites the compute-layout copies.
struct ParamSegment {
    int64_t start, count;  // canonical range
    int kind;              // 0 filter (weight decay on), 1 bias / gamma / beta (weight decay off)
    int type, k, cin, cout;
};
struct SgdArgs {
    const ParamSegment* segments = nullptr; int n_segments = 0;  // device table
    int64_t n_params = 0;
    float* master = nullptr; float* momentum = nullptr;
    const float* grad_tm = nullptr;   // filters tap-major [t][ci][co]; other segments canonical
    float* w_tm_f32 = nullptr; float* w_km_f32 = nullptr;   // [t][ci][co] / [t][co][ci]
    void* w_tm_bf16 = nullptr; void* w_km_bf16 = nullptr;
    double lr = 0, weight_decay = 0, momentum_coef = 0, grad_scale = 1;
    int apply = 1;  // 0: only refresh the compute-layout copies from master (after set_params)
    // table mode: the update is the last kernel of a step, and the bn accumulator tables are dead by then — it also clears them for
    // the next step (16-byte words), so the next forward starts without a fill on its critical path
    void* zero = nullptr; int64_t zero_words16 = 0;
    // the step's loss (fp32 at loss_src, final by the time this kernel runs) posted to pinned host memory as ONE 64-bit word
    // (tag << 32 | float bits): the host polls the tag — no copy, no event packet on the stream between two steps
    const float* loss_src = nullptr; unsigned long long* loss_post = nullptr; unsigned int loss_tag = 0;
};
void launch_sgd(const SgdArgs& a, hipStream_t s);
// canonical <-> tap-major conversion of a whole blob (get_grads, tests)
void launch_tm_to_canonical(const ParamSegment* segments, int n_segments, int64_t n_params, const float* tm, float* canonical, hipStream_t s);

// inference glue (annonet_infer.cpp:116-214)
struct BlendArgs {
    const float* logits_nchw = nullptr;  // [K][th][tw] of one tile
    float* blended = nullptr;            // [K][H][W]
    int k = 0, tile_h = 0, tile_w = 0, tile_left = 0, tile_top = 0;
    int img_h = 0, img_w = 0;
    long full[4] = {0, 0, 0, 0}, unique[4] = {0, 0, 0, 0};  // l,t,r,b
};
void launch_blend(const BlendArgs& a, hipStream_t s);
// The tiles of one inference batch (equal windows; logits [count][K][th][tw]) blended by ONE launch, with the result of `count` launch_blend
// calls in list order: inside its unique rectangle a tile assigns (no order involved); a frame pixel is handled by the FIRST tile of the
// batch that covers it, which adds the contributions of every covering tile of the batch in list order (annonet_infer.cpp:116-164).
struct BlendBatchArgs {
    const float* logits = nullptr; float* blended = nullptr;
    int k = 0, count = 0, tile_h = 0, tile_w = 0, img_h = 0, img_w = 0;
    int left[16] = {}, top[16] = {};
    int full[16][4] = {}, unique[16][4] = {};   // l, t, r, b (inclusive, image coordinates)
    unsigned nbr[16] = {};                       // bit j: tile j's full rectangle intersects this tile's (launch_blend_batch fills it)
};
bool blend_batch_ok(const BlendBatchArgs& a);   // the batch's unique rectangles are reached by no other tile of it (else: per-tile launches)
void launch_blend_batch(BlendBatchArgs a, hipStream_t s);
// bf16 inference: the 1x1 head (32 -> K <= 4 channels, bias) and the blend in one pass over the last hidden tensor — the tile's
// logits never go to memory.  BlendArgs::logits_nchw is unused.  head_blend_supported() decides.
struct HeadBlendArgs {
    Src src; int c_in = 0, k = 0;
    const float* w_tm = nullptr; const float* bias = nullptr;   // [ci][k] (bf16-rounded values), [k]
    BlendArgs blend;
};
bool head_blend_supported(const HeadBlendArgs& a);
bool conv_head_in_epilogue_ok(const ConvArgs& a);   // a = the conv of the last hidden layer, with out_scale / head_k set
void launch_head_blend(const HeadBlendArgs& a, hipStream_t s);
void launch_argmax(const float* blended, int k, int64_t pixels, const double* gains_or_null, uint16_t* labels, hipStream_t s);
void launch_argmax_range(const float* blended, int k, int64_t pixels, int64_t p0, int64_t p1, const double* gains_or_null, uint16_t* labels, hipStream_t s);

void launch_fill_zero(void* p, size_t bytes, hipStream_t s);
void launch_zero_rects(float* planes, int k, int H, int W, const anh_rect* d_rects, int n, hipStream_t s);   // zero n inclusive rectangles of every plane [k][H][W]

// Training crops cut on the device from full images resident in HBM (randomly_crop_image, annonet_train_main.cpp:110-232,
// for further_downscaling_factor = 1 and given draws).  kCropMaxClasses bounds the label values the histogram covers.
constexpr int kCropMaxClasses = 64;
struct CropSource {
    const uint8_t* image; const uint16_t* labels;   // full image u8 HWC and its label image, device pointers
    int height, width;
    int left, top;                                  // crop rectangle = [left, left+dim) x [top, top+dim), may leave the image
    int flip_lr, flip_ud;
    double gain;                                    // multiplicative brightness change (the reference's double); 1 = none
    // further downscaling (annonet_train_main.cpp:124-127,160-171): the rectangle is src_dim x src_dim (= round(dim * factor)) and
    // is resized to dim x dim — bilinear for the image, nearest neighbour for the labels; src_dim == dim: none
    int src_dim;
    // add_random_noise (:73-105): per channel value + uniform integer in [-noise_level, noise_level], clamped to 0..255; the draws
    // come from a counter-based generator keyed by (noise_seed, position in the final crop) instead of dlib::rand's sequence
    int noise_level;
    unsigned long long noise_seed;
    int color_offset[3];                            // apply_random_color_offset (:226-231): per-channel offsets drawn by the host, clamped add
};
// the counter-based draw of the device's add_random_noise (splitmix64 finaliser): uniform in [-level, level]
__host__ __device__ inline int crop_noise_draw(unsigned long long seed, unsigned long long counter, int level) {
    unsigned long long z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (int)(z % (unsigned long long)(2 * level + 1)) - level;
}
// pass 1: pixels (clamp-to-edge = extract_image_chip at scale 1 + outpaint), labels (ignore outside the image), flips and
// brightness; per crop a histogram of the UNFLIPPED labels and the first row-major position of every label value
void launch_crop_pixels(const CropSource* d_specs, int n, int dim, int channels, uint8_t* d_images, uint16_t* d_labels,
                        unsigned* d_hist, unsigned* d_firstpos, int* d_bad_label, int classes, hipStream_t s);
// pass 2: weights[p] = table[crop][label] (0 for ignored pixels); table from set_weights' arithmetic on the histograms
void launch_crop_weights(const uint16_t* d_labels, const float* d_table, int n, int dim, float* d_weights, hipStream_t s);
// detection-level filter of annonet_infer() on the device (kernels_generic.hip)
void run_detection_filter(const float* d_blended, uint16_t* d_labels, int k, int h, int w, const double* d_det, uint8_t* d_flags, int* d_changed,
                          hipStream_t s);
void launch_reduce_partials(const float* partials, int splits, int64_t nw, float* out, hipStream_t s);

struct BnFoldJobs;
// training-time tail fused into one pass: 1x1 head forward + weighted softmax log-loss + head backward-data + head
// filter / bias gradient (32 input channels, up to 4 classes; other shapes take the separate kernels)
struct HeadTrainArgs {
    Src src; int c_in = 0;
    const float* w_tm = nullptr; const float* w_km = nullptr; const float* bias = nullptr;
    const uint16_t* labels = nullptr; const float* weights = nullptr;
    float* logits = nullptr; void* da = nullptr;   // da may be null when bnred_partials is set: see BnBwdArgs::head_g
    float* dlogits = nullptr;                      // optional: [P][k] fp32, the loss gradient at the logits
    int64_t pixels = 0; int k = 0; double scale = 0;
    double* partials = nullptr;
    double* loss_out = nullptr; float* loss_out_f32 = nullptr; float* dbias = nullptr; float* dw = nullptr;
    int* error_flag = nullptr;
    // optional (single-input head): also leave the bn + relu backward sums of the layer that produced the input —
    // the head computes that layer's da from its raw output y in the same pass.  Layout [channel][sum dz*xhat | sum dz][workgroup].
    const float* bnred_mean = nullptr; const float* bnred_invstd = nullptr; double* bnred_partials = nullptr;
    long long* bnred_acc = nullptr;        // table mode: the sums go to the input layer's accumulator table (bnred_partials unset)
    BnBwdFinish bnred_finish;              // with bnred_acc
    const BnFoldJobs* fold = nullptr;      // table mode: workgroup j < fold->n also runs fold job j
};
// Table mode: the arrays (mean, invstd, scale, shift, var) and the running statistics of every bn layer, formed from the accumulator
// tables by ONE launch per step — or by the first workgroups of the fused head kernel, which runs before any backward kernel reads them.
struct BnFoldJob {
    const long long* acc; const float* gamma; const float* beta;
    float* mean; float* invstd; float* scale; float* shift; double* var;
    float* rmean; float* rvar;            // null: no running update
    double pixels, af, unbias; float eps; int c;
};
struct BnFoldJobs { int n = 0; BnFoldJob job[16]; };
void launch_bn_fold_all(const BnFoldJobs& jobs, hipStream_t s);
bool head_train_supported(const HeadTrainArgs& a);
int head_train_blocks(int64_t pixels);   // workgroups of the launch = partials per channel
int64_t head_train_partial_doubles(const HeadTrainArgs& a);
void launch_head_train(const HeadTrainArgs& a, hipStream_t s);

// kernels_mfma.hip: bf16 MFMA implicit-GEMM kernels; *_supported() says whether a shape is covered
bool mfma_conv_supported(const ConvArgs& a);
void launch_conv_mfma(const ConvArgs& a, hipStream_t s);
// > 0: the MFMA kernel of this layer can also write the bn statistic partials (set ConvArgs::stat_partials; the
// value is the number of partials per channel to pass to launch_bn_forward_finalize)
int conv_fused_stat_blocks(const ConvArgs& a);
int conv_fused_bnred_blocks(const ConvArgs& a);   // same for ConvArgs::bnred_partials (backward-data convs)
bool conv_folds_bn_tables(const ConvArgs& a);     // the layer's kernel reads Src::a_tab / b_tab (the persistent kernels)
bool conv_stores_activation(const ConvArgs& a);   // the layer's MFMA kernel honours ConvArgs::out_scale / out_shift (src.kind SRC_RAW, SRC_SUM2 or SRC_IMAGE)
bool mfma_wgrad_supported(const WgradArgs& a);
void launch_wgrad_mfma(const WgradArgs& a, hipStream_t s);
int64_t wgrad_mfma_scratch_floats(const WgradArgs& a);

// kernel choice shared by the engine and the single-op entry points: MFMA when the mode is bf16 and the shape is covered
inline bool conv_takes_mfma(const ConvArgs& a, DType mode) { return mode == DT_BF16 && mfma_conv_supported(a); }
inline bool wgrad_takes_mfma(const WgradArgs& a, DType mode) { return mode == DT_BF16 && mfma_wgrad_supported(a); }

}  // namespace anh
