// NetPimpl.h — drop-in replacement for dlib-dnn-pimpl-wrapper/NetPimpl.h on MI355X.
//
// annonet's host code (annonet_infer.cpp, annonet_train_main.cpp, annonet_infer_main.cpp, annonet_train.h, annonet.h)
// includes "dlib-dnn-pimpl-wrapper/NetPimpl.h" and uses exactly the surface below (every member is
// cited at its call site).  This header keeps that surface and forwards to libannonet_hip.so's C ABI
// (include/annonet_hip.h).  Where the reference passes dlib value types across the boundary, a host that has dlib
// keeps passing them: the adapters only need row-major storage, which dlib::matrix provides; for hosts without dlib
// (this repository's own tests) minimal stand-ins are provided under ANNONET_HIP_NO_DLIB.
//
//   reference (file:line)                                   this header
//   NetPimpl::input_type (annonet.h:47)                     dlib::matrix<rgb_pixel> / matrix<uint8_t>  (or the stand-in)
//   NetPimpl::training_label_type (annonet_train.h:74)      matrix<weighted_label>                     (or the stand-in)
//   RuntimeNet::Forward (annonet_infer.cpp:77)              anh_runtime_forward -> tensor view (k/nr/nc/host)
//   RuntimeNet::GetRecommendedInputDimension (:49-50)       anh_recommended_input_dim
//   RuntimeNet::Serialize/Deserialize (annonet_train_main.cpp:558-561, annonet_infer_main.cpp:347-351)
//   TrainingNet::* (annonet_train_main.cpp:396-410,570,609) anh_trainer_*
//   TrainingNet::GetRequiredInputDimension (:376)           anh_required_input_dim
#ifndef ANNONET_HIP_NETPIMPL_H
#define ANNONET_HIP_NETPIMPL_H

#include <chrono>
#include <cstdint>
#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "annonet_hip.h"

#ifdef ANNONET_HIP_NO_DLIB
// ---- minimal stand-ins for the dlib value types that cross the boundary ----
namespace dlib {
struct rgb_pixel { unsigned char red, green, blue; };
struct point { long x_, y_; point(long x = 0, long y = 0) : x_(x), y_(y) {} long x() const { return x_; } long y() const { return y_; } };
struct rectangle {  // inclusive l,t,r,b like dlib::rectangle
    long l = 0, t = 0, r = -1, b = -1;
    rectangle() = default;
    rectangle(long l_, long t_, long r_, long b_) : l(l_), t(t_), r(r_), b(b_) {}
    long left() const { return l; } long top() const { return t; } long right() const { return r; } long bottom() const { return b; }
    unsigned long width() const { return r < l ? 0 : (unsigned long)(r - l + 1); }
    unsigned long height() const { return b < t ? 0 : (unsigned long)(b - t + 1); }
    bool contains(long x, long y) const { return x >= l && x <= r && y >= t && y <= b; }
};
template <typename T> class matrix {  // row-major, the subset annonet uses
  public:
    void set_size(long nr, long nc) { nr_ = nr; nc_ = nc; d_.resize((size_t)nr * nc); }
    long nr() const { return nr_; } long nc() const { return nc_; }
    T& operator()(long r, long c) { return d_[(size_t)r * nc_ + c]; }
    const T& operator()(long r, long c) const { return d_[(size_t)r * nc_ + c]; }
    T* begin() { return d_.data(); } T* end() { return d_.data() + d_.size(); }
    const T* begin() const { return d_.data(); } const T* end() const { return d_.data() + d_.size(); }
    size_t size() const { return d_.size(); }
  private:
    long nr_ = 0, nc_ = 0;
    std::vector<T> d_;
};
struct loss_multiclass_log_per_pixel_ { static const uint16_t label_to_ignore = 65535; };
struct loss_multiclass_log_per_pixel_weighted_ {
    struct weighted_label {
        weighted_label() = default;
        weighted_label(uint16_t l, float w = 1.f) : label(l), weight(w) {}
        uint16_t label = 0; float weight = 1.f;
    };
};
}  // namespace dlib
#endif  // ANNONET_HIP_NO_DLIB

namespace NetPimpl {

#ifdef DLIB_DNN_PIMPL_WRAPPER_GRAYSCALE_INPUT
typedef dlib::matrix<uint8_t> input_type;
enum { kInputChannels = 1 };
#else
typedef dlib::matrix<dlib::rgb_pixel> input_type;
enum { kInputChannels = 3 };
#endif
typedef dlib::matrix<dlib::loss_multiclass_log_per_pixel_weighted_::weighted_label> training_label_type;

#ifndef DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT
#define DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT 2
#endif

static_assert(sizeof(dlib::rgb_pixel) == 3, "rgb_pixel must be three packed bytes");
static_assert(sizeof(dlib::loss_multiclass_log_per_pixel_weighted_::weighted_label) == sizeof(anh_wlabel), "weighted_label layout differs from anh_wlabel");

inline void check(int status) {  // the reference's error convention: std::exception subclasses (annonet_train_main.cpp:616-644)
    if (status != ANH_OK) throw std::runtime_error(std::string("annonet_hip: ") + anh_last_error());
}

// Extension (the reference has only --primary-cuda-device, annonet_train_main.cpp:307,392-394): the GPUs that nets created
// afterwards on this thread drive.  TrainingNet::StartTraining then splits every mini-batch over them (one RCCL all-reduce of
// the gradients per step), annonet_infer() splits the tile list; the host code is unchanged.  {} = one device again.
inline void SetDevices(const std::vector<int>& devices) { check(anh_set_devices(devices.data(), (int)devices.size())); }

// read-only view standing in for `const dlib::tensor&` (annonet_infer.cpp:77-100)
class OutputTensor {
  public:
    long long num_samples() const { return 1; }
    long long k() const { return k_; } long long nr() const { return nr_; } long long nc() const { return nc_; }
    const float* host() const { return data_; }
  private:
    friend class RuntimeNet;
    const float* data_ = nullptr;
    int k_ = 0, nr_ = 0, nc_ = 0;
};

class RuntimeNet {
  public:
    RuntimeNet() = default;
    explicit RuntimeNet(anh_runtime* h) : h_(h) {}
    RuntimeNet(const RuntimeNet& o) { copy_from(o); }
    RuntimeNet& operator=(const RuntimeNet& o) { if (this != &o) { reset(); copy_from(o); } return *this; }
    RuntimeNet(RuntimeNet&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~RuntimeNet() { reset(); }

    static int GetRecommendedInputDimension(int minimum) { return anh_recommended_input_dim(DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT, minimum); }

    const OutputTensor& Forward(const input_type& tile) {  // annonet_infer.cpp:77
        require();
        check(anh_runtime_forward(h_, reinterpret_cast<const uint8_t*>(&*tile.begin()), 1, (int)tile.nr(), (int)tile.nc(), &out_.data_, &out_.k_, &out_.nr_, &out_.nc_));
        return out_;
    }
    void Serialize(std::ostream& os) const {  // annonet_train_main.cpp:560-561
        require();
        void* blob = nullptr; size_t n = 0;
        check(anh_runtime_serialize(h_, &blob, &n));
        os.write(static_cast<const char*>(blob), (std::streamsize)n);
        anh_free(blob);
    }
    void Deserialize(std::istream& is, int precision = ANH_BF16) {  // annonet_infer_main.cpp:349-350
        std::string blob((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
        reset();
        check(anh_runtime_deserialize(blob.data(), blob.size(), precision, &h_));
    }
    anh_runtime* handle() const { return h_; }

  private:
    void require() const { if (!h_) throw std::runtime_error("annonet_hip: RuntimeNet holds no network (Deserialize it or take it from TrainingNet::GetRuntimeNet)"); }
    void reset() { if (h_) anh_runtime_destroy(h_); h_ = nullptr; }
    void copy_from(const RuntimeNet& o) {
        if (!o.h_) return;
        void* blob = nullptr; size_t n = 0;
        anh_net_config cfg;
        check(anh_runtime_config(o.h_, &cfg));
        check(anh_runtime_serialize(o.h_, &blob, &n));
        const int rc = anh_runtime_deserialize(blob, n, cfg.precision, &h_);
        anh_free(blob);
        check(rc);
    }
    anh_runtime* h_ = nullptr;
    OutputTensor out_;
};

// Extension (no counterpart in dlib-dnn-pimpl-wrapper): the dataset's full images resident in HBM, so that the loader
// threads of annonet_train_main.cpp:504-553 push crop SPECS (rectangle, flips, brightness — their own dlib::rand draws) instead
// of crops, and TrainingNet::StartTrainingOnCrops cuts the mini-batch on the device (randomly_crop_image, :110-232).
class Dataset {
  public:
    Dataset() { check(anh_dataset_create(kInputChannels, &h_)); }
    Dataset(const Dataset&) = delete;
    Dataset& operator=(const Dataset&) = delete;
    ~Dataset() { if (h_) anh_dataset_destroy(h_); }
    // full image + its label image (sample.input_image, sample.label_image); returns the index crop specs refer to
    template <typename label_image_type>
    int Add(const input_type& image, const label_image_type& label_image) {
        if (image.nr() != label_image.nr() || image.nc() != label_image.nc()) throw std::runtime_error("annonet_hip: image and label image sizes differ");
        int index = -1;
        check(anh_dataset_add(h_, reinterpret_cast<const uint8_t*>(&*image.begin()), reinterpret_cast<const uint16_t*>(&*label_image.begin()),
                              (int)image.nr(), (int)image.nc(), &index));
        return index;
    }
    void Remove(int index) { check(anh_dataset_remove(h_, index)); }   // frees that image's HBM; the index is reused by a later Add
    uint64_t ResidentBytes() const { uint64_t n = 0; check(anh_dataset_resident_bytes(h_, &n)); return n; }
    anh_dataset* handle() const { return h_; }

  private:
    anh_dataset* h_ = nullptr;
};

class TrainingNet {
  public:
    TrainingNet() { check(anh_trainer_create(&h_)); check(anh_trainer_set_levels(h_, DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT)); check(anh_trainer_set_input_channels(h_, kInputChannels)); }
    TrainingNet(const TrainingNet&) = delete;
    TrainingNet& operator=(const TrainingNet&) = delete;
    ~TrainingNet() { if (h_) anh_trainer_destroy(h_); }

    static int GetRequiredInputDimension() {  // annonet_train_main.cpp:376
        anh_net_config cfg{DLIB_DNN_PIMPL_WRAPPER_LEVEL_COUNT, kInputChannels, 2, 1.0, 1, ANH_BF16};
        return anh_required_input_dim(&cfg);
    }
    void Initialize() { check(anh_trainer_initialize(h_)); }                                                      // :400
    void SetNetWidth(double scaler, int min_filter_count) { check(anh_trainer_set_net_width(h_, scaler, min_filter_count)); }  // :402
    void SetSynchronizationFile(const std::string& file, std::chrono::seconds interval) {                          // :403
        check(anh_trainer_set_synchronization_file(h_, file.c_str(), (double)interval.count()));
    }
    void BeVerbose() { check(anh_trainer_be_verbose(h_)); }                                                       // :404
    void SetClassCount(size_t n) { check(anh_trainer_set_class_count(h_, n)); }                                    // :405
    void SetLearningRate(double lr) { check(anh_trainer_set_learning_rate(h_, lr)); }                              // :406
    void SetLearningRateShrinkFactor(double f) { check(anh_trainer_set_learning_rate_shrink_factor(h_, f)); }      // :407
    void SetIterationsWithoutProgressThreshold(unsigned long n) { check(anh_trainer_set_iterations_without_progress_threshold(h_, n)); }  // :408
    void SetPreviousLossValuesDumpAmount(unsigned long n) { check(anh_trainer_set_previous_loss_values_dump_amount(h_, n)); }             // :409
    void SetAllBatchNormalizationRunningStatsWindowSizes(unsigned long n) { check(anh_trainer_set_all_bn_running_stats_window_sizes(h_, n)); }  // :410
    double GetLearningRate() const { return anh_trainer_get_learning_rate(h_); }                                   // :570

    // annonet_train_main.cpp:609.  The inputs are copied to HBM before this returns (the host clears them next, :585-586).
    void StartTraining(const std::vector<input_type>& samples, const std::vector<training_label_type>& labels) {
        if (samples.empty() || samples.size() != labels.size()) throw std::runtime_error("annonet_hip: samples and labels must be non-empty and of equal size");
        std::vector<const uint8_t*> ip(samples.size());
        std::vector<const anh_wlabel*> lp(samples.size());
        const long nr = samples[0].nr(), nc = samples[0].nc();
        for (size_t i = 0; i < samples.size(); ++i) {
            if (samples[i].nr() != nr || samples[i].nc() != nc || labels[i].nr() != nr || labels[i].nc() != nc)
                throw std::runtime_error("annonet_hip: all samples and label images of a mini-batch must share one size");
            ip[i] = reinterpret_cast<const uint8_t*>(&*samples[i].begin());
            lp[i] = reinterpret_cast<const anh_wlabel*>(&*labels[i].begin());
        }
        check(anh_trainer_step(h_, ip.data(), lp.data(), (int)samples.size(), (int)nr, (int)nc));
    }
    // Extension: one optimiser step on crops cut on the device from `dataset` (see Dataset); dim = crop side (:382)
    void StartTrainingOnCrops(const Dataset& dataset, const std::vector<anh_crop_spec>& crops, int dim, double class_weight, double image_weight) {
        if (crops.empty()) throw std::runtime_error("annonet_hip: empty mini-batch");
        check(anh_trainer_step_crops(h_, dataset.handle(), crops.data(), (int)crops.size(), dim, class_weight, image_weight));
    }
    RuntimeNet GetRuntimeNet(int precision = ANH_BF16) const {  // :558, by value
        anh_runtime* rt = nullptr;
        check(anh_trainer_snapshot_runtime(h_, precision, &rt));
        return RuntimeNet(rt);
    }
    anh_trainer* handle() const { return h_; }

  private:
    anh_trainer* h_ = nullptr;
};

}  // namespace NetPimpl

#endif  // ANNONET_HIP_NETPIMPL_H
