// Host-side smoke of the C++ drop-in headers (NetPimpl.h, tiling/, annonet_infer_hip.h) without dlib.
// Reads like the reference's own host code: annonet_train_main.cpp:396-410,583-613 and annonet_infer_main.cpp:347-351,468.
// Without a GPU it exercises only the host logic and checks that compute entry points fail with an exception.
#define ANNONET_HIP_NO_DLIB
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "annonet_infer_hip.h"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    // static dimension maths (annonet_train_main.cpp:376-382)
    const int required = NetPimpl::TrainingNet::GetRequiredInputDimension();
    REQUIRE(required == 35);
    REQUIRE(NetPimpl::RuntimeNet::GetRecommendedInputDimension(227) == 227);
    REQUIRE(NetPimpl::RuntimeNet::GetRecommendedInputDimension(105) == 107);
    // tiling (annonet_infer_main.cpp:423-427)
    tiling::parameters tp;
    tp.max_tile_width = tp.max_tile_height = 1024;
    tp.overlap_x = tp.overlap_y = required;
    const auto tiles = tiling::get_tiles(4096, 4096, tp);
    REQUIRE(tiles.size() == 25);
    REQUIRE(tiles.front().full_rect.left() == 0 && tiles.back().full_rect.right() == 4095);
    REQUIRE(tiling::get_tiles(227, 227).size() == 1);

    if (anh_device_count() == 0) {
        bool threw = false;
        try { NetPimpl::TrainingNet t; t.Initialize(); } catch (const std::exception& e) { threw = true; std::printf("no GPU: %s\n", e.what()); }
        REQUIRE(threw);
        std::printf("shim smoke ok (host logic only)\n");
        return 0;
    }

    NetPimpl::TrainingNet training_net;
    training_net.Initialize();
    training_net.SetNetWidth(0.25, 4);
    training_net.SetClassCount(3);
    training_net.SetLearningRate(0.1);
    training_net.SetLearningRateShrinkFactor(0.1);
    training_net.SetIterationsWithoutProgressThreshold(4000);
    training_net.SetPreviousLossValuesDumpAmount(800);
    training_net.SetAllBatchNormalizationRunningStatsWindowSizes(200);
    const int dim = NetPimpl::RuntimeNet::GetRecommendedInputDimension(43);
    std::vector<NetPimpl::input_type> samples(4);
    std::vector<NetPimpl::training_label_type> labels(4);
    unsigned seed = 1;
    for (int i = 0; i < 4; ++i) {
        samples[i].set_size(dim, dim);
        labels[i].set_size(dim, dim);
        for (long r = 0; r < dim; ++r)
            for (long c = 0; c < dim; ++c) {
                seed = seed * 1664525u + 1013904223u;
                samples[i](r, c) = dlib::rgb_pixel{(unsigned char)(seed >> 8), (unsigned char)(seed >> 16), (unsigned char)(seed >> 24)};
                labels[i](r, c) = dlib::loss_multiclass_log_per_pixel_weighted_::weighted_label((uint16_t)((seed >> 5) % 3), 1.f);
            }
    }
    for (int step = 0; step < 3; ++step) training_net.StartTraining(samples, labels);
    REQUIRE(training_net.GetLearningRate() == 0.1);
    {   // the device-side data path: a full image resident in HBM, a step from crop specs (no host mini-batch)
        NetPimpl::Dataset dataset;
        NetPimpl::input_type full;
        dlib::matrix<uint16_t> full_labels;
        full.set_size(120, 150);
        full_labels.set_size(120, 150);
        for (long r = 0; r < 120; ++r)
            for (long c = 0; c < 150; ++c) {
                seed = seed * 1664525u + 1013904223u;
                full(r, c) = dlib::rgb_pixel{(unsigned char)(seed >> 8), (unsigned char)(seed >> 16), (unsigned char)(seed >> 24)};
                full_labels(r, c) = (uint16_t)((r / 16 + c / 16) % 3);
            }
        REQUIRE(dataset.Add(full, full_labels) == 0);
        std::vector<anh_crop_spec> crops = {{0, 10, 5, 0, 0, 1.0}, {0, -7, 90, 1, 0, 1.1}, {0, 120, -3, 0, 1, 1.0}, {0, 60, 60, 1, 1, 0.9}};
        training_net.StartTrainingOnCrops(dataset, crops, dim, 0.5, 0.5);
        bool bad = false;
        try { crops[0].image = 3; training_net.StartTrainingOnCrops(dataset, crops, dim, 0.5, 0.5); } catch (const std::exception&) { bad = true; }
        REQUIRE(bad);
    }
    const NetPimpl::RuntimeNet runtime_net = training_net.GetRuntimeNet();
    std::ostringstream serialized;
    runtime_net.Serialize(serialized);
    NetPimpl::RuntimeNet net;
    { std::istringstream iss(serialized.str()); net.Deserialize(iss); }
    const auto& out = net.Forward(samples[0]);
    REQUIRE(out.k() == 3 && out.nr() == dim && out.nc() == dim && out.host() != nullptr);
    dlib::matrix<uint16_t> result;
    annonet_infer_temp temp;
    NetPimpl::input_type image;
    image.set_size(150, 170);
    for (auto& p : image) { seed = seed * 1664525u + 1013904223u; p = dlib::rgb_pixel{(unsigned char)(seed >> 8), (unsigned char)(seed >> 16), (unsigned char)(seed >> 24)}; }
    tiling::parameters small;
    small.max_tile_width = small.max_tile_height = 96;
    small.overlap_x = small.overlap_y = required;
    annonet_infer(net, image, result, temp, {}, {}, small);
    REQUIRE(result.nr() == 150 && result.nc() == 170);
    for (auto v : result) REQUIRE(v < 3);
    {   // blended_output (annonet_infer.h:31): stays in HBM unless asked for (or defaulted on: ANNONET_HIP_KEEP_BLENDED_OUTPUT / ANH_KEEP_BLENDED_OUTPUT);
        // when it comes back, the label map is its argmax with the reference's tie rule (annonet_infer.cpp:170-185)
        REQUIRE(temp.keep_blended_output == annonet_hip_keep_blended_default());
        annonet_infer_temp keeping;
        keeping.keep_blended_output = true;
        dlib::matrix<uint16_t> again;
        annonet_infer(net, image, again, keeping, {}, {}, small);
        REQUIRE(keeping.blended_output.size() == 3 && keeping.blended_output[0].nr() == 150 && keeping.blended_output[0].nc() == 170);
        for (long r = 0; r < 150; ++r)
            for (long c = 0; c < 170; ++c) {
                REQUIRE(again(r, c) == result(r, c));
                uint16_t best = 0;
                for (uint16_t k = 1; k < 3; ++k) if (keeping.blended_output[k](r, c) > keeping.blended_output[best](r, c)) best = k;
                REQUIRE(best == result(r, c));
            }
    }
    bool threw = false;
    try { NetPimpl::input_type bad; bad.set_size(24, 24); net.Forward(bad); } catch (const std::exception&) { threw = true; }
    REQUIRE(threw);
    {   // two replicas from this one process (two GPUs when the box has them, else device 0 twice = the rehearsal backend):
        // the training loop and annonet_infer() above, unchanged
        NetPimpl::SetDevices({0, anh_device_count() > 1 ? 1 : 0});
        NetPimpl::TrainingNet dp_net;
        dp_net.Initialize();
        dp_net.SetNetWidth(0.25, 4);
        dp_net.SetClassCount(3);
        dp_net.SetLearningRate(0.1);
        REQUIRE(anh_handle_replicas(dp_net.handle(), 1) == 2);
        for (int step = 0; step < 3; ++step) dp_net.StartTraining(samples, labels);
        const NetPimpl::RuntimeNet snapshot = dp_net.GetRuntimeNet(ANH_FP32);   // ONE replica (the host serializes it, annonet_train_main.cpp:557-565)
        REQUIRE(anh_handle_replicas(snapshot.handle(), 0) == 1);
        std::ostringstream blob;
        snapshot.Serialize(blob);
        NetPimpl::RuntimeNet dp_runtime;                                          // a net the host creates for inference spans the selected devices
        { std::istringstream iss(blob.str()); dp_runtime.Deserialize(iss, ANH_FP32); }
        REQUIRE(anh_handle_replicas(dp_runtime.handle(), 0) == 2);
        dlib::matrix<uint16_t> dp_result;
        annonet_infer(dp_runtime, image, dp_result, temp, {}, {}, small);   // tiles sharded over the replicas
        NetPimpl::SetDevices({});
        NetPimpl::RuntimeNet one_device;
        { std::istringstream iss(blob.str()); one_device.Deserialize(iss, ANH_FP32); }
        REQUIRE(anh_handle_replicas(one_device.handle(), 0) == 1);
        dlib::matrix<uint16_t> one_result;
        annonet_infer(one_device, image, one_result, temp, {}, {}, small);
        long differing = 0;
        for (long r = 0; r < 150; ++r) for (long c = 0; c < 170; ++c) differing += dp_result(r, c) != one_result(r, c);
        REQUIRE(differing <= 3);   // equal except exact near-ties where four tiles meet (summation order)
    }
    std::printf("shim smoke ok (trained 3 steps, serialized %zu bytes, inferred 150x170; 2 replicas: 3 data-parallel steps + sharded inference)\n", serialized.str().size());
    return 0;
}
