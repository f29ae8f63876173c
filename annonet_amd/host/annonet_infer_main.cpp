// annonet_infer_main.cpp — the reference's inference tool (annonet_infer_main.cpp:283-538) on the drop-in headers: reader pool ->
// annonet_infer() on the GPU(s) -> writer pool, the tool's own timing lines, per-pixel and per-region confusion matrices.
// The per-image loop is the reference's (its variable names are kept where it helps reading the two side by side); what differs:
//   * cxxopts / rapidjson / dlib are absent here: a small option reader, annonet_host.h's JSON reader and image_io.h stand in;
//   * --precision fp32|bf16 (extension): ANH_FP32 reproduces the CPU oracle's label maps bit for bit, ANH_BF16 is the fast path;
//   * --devices 0,1,... (extension): one process drives several GPUs (anh_set_devices), the tile list of every image is sharded.
// usage: annonet_infer_hip <input-directory> [-g idx:gain]... [-d idx:level]... [-w N] [-h N] [--full-image-reader-thread-count N]
//                          [--result-image-writer-thread-count N] [--precision fp32|bf16] [--devices list] [--dnn annonet.dnn]
#define ANNONET_HIP_NO_DLIB
#include "../../include/annonet_infer_hip.h"
#include "annonet_host.h"

#include <functional>

using namespace std;
using namespace dlib;

// ---------------------------------------------------------------------------------------- annonet_infer_main.cpp:30-63
struct class_specific_value_type {
    uint16_t class_index = dlib::loss_multiclass_log_per_pixel_::label_to_ignore;
    double value = 0.0;
};

class_specific_value_type parse_class_specific_value(const std::string& string_from_command_line) {
    const auto colon_pos = string_from_command_line.find(':');
    if (colon_pos == std::string::npos || colon_pos < 1 || colon_pos >= string_from_command_line.length() - 1)
        throw std::runtime_error("The gains must be supplied in the format index:gain (e.g., 1:-0.5)");
    class_specific_value_type class_specific_value;
    class_specific_value.class_index = (uint16_t)std::stoul(string_from_command_line.substr(0, colon_pos));
    class_specific_value.value = std::stod(string_from_command_line.substr(colon_pos + 1));
    return class_specific_value;
}

std::vector<double> parse_class_specific_values(const std::vector<std::string>& strings_from_command_line, uint16_t class_count) {
    std::vector<double> class_specific_values(class_count, 0.0);
    for (const auto& string_from_command_line : strings_from_command_line) {
        const auto class_specific_value = parse_class_specific_value(string_from_command_line);
        if (class_specific_value.class_index >= class_count) {
            std::ostringstream error;
            error << "Can't define class-specific value for index " << class_specific_value.class_index << " when there are only " << class_count << " classes";
            throw std::runtime_error(error.str());
        }
        class_specific_values[class_specific_value.class_index] = class_specific_value.value;
    }
    return class_specific_values;
}

void index_label_image_to_rgba_label_image(const matrix<uint16_t>& index_label_image, matrix<rgb_alpha_pixel>& rgba_label_image, const std::vector<AnnoClass>& anno_classes) {   // :74-87
    const long nr = index_label_image.nr(), nc = index_label_image.nc();
    rgba_label_image.set_size(nr, nc);
    for (long r = 0; r < nr; ++r)
        for (long c = 0; c < nc; ++c) {
            const uint16_t index_label = index_label_image(r, c);
            rgba_label_image(r, c) = index_label < anno_classes.size() ? anno_classes[index_label].rgba_label : rgba_ignore_label;   // (65535: an all-NaN pixel)
        }
}

struct result_image_type {   // :275-281
    std::string filename;
    int original_width = 0, original_height = 0;
    matrix<uint16_t> label_image;
};

struct Options {
    std::string input_directory, dnn = "annonet.dnn", precision = "bf16";
    std::vector<std::string> gain, detection;
    std::vector<int> devices;
    int tile_max_width = 1024, tile_max_height = 1024;   // the reference's DLIB_USE_CUDA defaults (:300-303)
    int reader_threads = (int)std::max(1u, std::thread::hardware_concurrency()), writer_threads = (int)std::max(1u, std::thread::hardware_concurrency());
};

static const char* kHelp =
    "Do inference using trained semantic-segmentation networks\nUsage:\n  annonet_infer_hip [OPTION...] <input-directory>\n\n"
    "  -i, --input-directory arg            Input image directory\n  -g, --gain arg                       Supply a class-specific gain, for example: 1:-0.5\n"
    "  -d, --detection arg                  Supply a class-specific detection level that _comes on top of gain_, for example: 1:1.5\n"
    "  -w, --tile-max-width arg             Set max tile width (default: 1024)\n  -h, --tile-max-height arg            Set max tile height (default: 1024)\n"
    "      --full-image-reader-thread-count arg\n      --result-image-writer-thread-count arg\n"
    "      --precision fp32|bf16            fp32 = bit-exact parity mode, bf16 = MFMA throughput mode (default)\n"
    "      --devices 0,1,...                GPUs this process drives (tile lists are sharded over them)\n      --dnn file                       trained net (default: annonet.dnn)\n";

Options parse_options(int argc, char** argv) {
    Options o;
    auto need = [&](int& i) -> std::string { if (i + 1 >= argc) throw std::runtime_error(std::string("Option '") + argv[i] + "' is missing an argument"); return argv[++i]; };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-i" || a == "--input-directory") o.input_directory = need(i);
        else if (a == "-g" || a == "--gain") o.gain.push_back(need(i));
        else if (a == "-d" || a == "--detection") o.detection.push_back(need(i));
        else if (a == "-w" || a == "--tile-max-width") o.tile_max_width = std::stoi(need(i));
        else if (a == "-h" || a == "--tile-max-height") o.tile_max_height = std::stoi(need(i));
        else if (a == "--full-image-reader-thread-count") o.reader_threads = std::stoi(need(i));
        else if (a == "--result-image-writer-thread-count") o.writer_threads = std::stoi(need(i));
        else if (a == "--precision") o.precision = need(i);
        else if (a == "--dnn") o.dnn = need(i);
        else if (a == "--devices") { std::stringstream ss(need(i)); std::string item; while (std::getline(ss, item, ',')) o.devices.push_back(std::stoi(item)); }
        else if (!a.empty() && a[0] == '-') throw std::runtime_error("Option '" + a + "' does not exist");
        else if (o.input_directory.empty()) o.input_directory = a;
        else throw std::runtime_error("Unexpected argument " + a);
    }
    if (o.input_directory.empty()) throw std::runtime_error("Option 'input-directory' is required but not present");
    if (o.precision != "bf16" && o.precision != "fp32") throw std::runtime_error("--precision must be fp32 or bf16");
    return o;
}

int main(int argc, char** argv) try {
    if (argc == 1) {
        cout << "You call this program like this: " << endl << "./annonet_infer_hip /path/to/image/data" << endl << endl << "You will also need a trained 'annonet.dnn' file. " << endl << endl;
        return 1;
    }
    Options options;
    try {
        options = parse_options(argc, argv);
        std::cout << "Input directory = " << options.input_directory << std::endl;
    } catch (std::exception& e) {
        cerr << e.what() << std::endl << std::endl << kHelp << std::endl;
        return 2;
    }

    // deserialize("annonet.dnn") >> anno_classes_json >> downscaling_factor >> serialized_runtime_net   (:340-343)
    double downscaling_factor = 1.0;
    std::string serialized_runtime_net, anno_classes_json;
    {
        const std::string file = annonet_io::slurp(options.dnn);
        char* json = nullptr; void* blob = nullptr; size_t json_size = 0, blob_size = 0;
        NetPimpl::check(anh_dnn_envelope_unpack(file.data(), file.size(), &json, &json_size, &downscaling_factor, &blob, &blob_size));
        anno_classes_json.assign(json, json_size);
        serialized_runtime_net.assign(static_cast<const char*>(blob), blob_size);
        anh_free(json); anh_free(blob);
    }
    std::cout << "Deserializing annonet, downscaling factor = " << downscaling_factor << std::endl;

    if (!options.devices.empty()) NetPimpl::SetDevices(options.devices);
    NetPimpl::RuntimeNet net;
    {
        std::istringstream iss(serialized_runtime_net);
        net.Deserialize(iss, options.precision == "fp32" ? ANH_FP32 : ANH_BF16);
    }
    const std::vector<AnnoClass> anno_classes = parse_anno_classes(anno_classes_json);
    if (anno_classes.size() < 2) throw std::runtime_error("at least two classes are needed");
    {
        anh_net_config cfg;
        NetPimpl::check(anh_runtime_config(net.handle(), &cfg));
        if ((size_t)cfg.classes != anno_classes.size()) throw std::runtime_error("the net's class count differs from the anno classes of the .dnn file");
    }

    const std::vector<double> gains = parse_class_specific_values(options.gain, (uint16_t)anno_classes.size());
    const std::vector<double> detection_levels = parse_class_specific_values(options.detection, (uint16_t)anno_classes.size());
    std::cout << "Using gains:";
    for (size_t class_index = 0, end = gains.size(); class_index < end; ++class_index) std::cout << " " << class_index << ":" << gains[class_index];
    std::cout << std::endl;
    std::cout << "Using detection levels:";
    for (size_t class_index = 0, end = detection_levels.size(); class_index < end; ++class_index) std::cout << " " << class_index << ":" << detection_levels[class_index];
    std::cout << std::endl;

    annonet_infer_temp temp;
    auto files = find_image_files(options.input_directory, false);

    anh_host::pipe<image_filenames_type> full_image_read_requests(std::max<size_t>(files.size(), 1));
    for (const auto& file : files) full_image_read_requests.enqueue(image_filenames_type(file));

    const int full_image_reader_count = std::max(1, options.reader_threads);
    const int result_image_writer_count = std::max(1, options.writer_threads);

    // The reference dequeues read results in completion order and pairs result i with whatever arrived i-th (:446-453); the
    // sample carries its own file names, so that is harmless there and here.
    anh_host::pipe<sample_type> full_image_read_results((size_t)full_image_reader_count);
    std::vector<std::thread> full_image_readers;
    for (int i = 0; i < full_image_reader_count; ++i) {
        full_image_readers.push_back(std::thread([&]() {
            image_filenames_type image_filenames;
            while (full_image_read_requests.dequeue(image_filenames))
                full_image_read_results.enqueue(read_sample(image_filenames, anno_classes, false, downscaling_factor));
        }));
    }

    anh_host::pipe<result_image_type> result_image_write_requests((size_t)result_image_writer_count);
    anh_host::pipe<bool> result_image_write_results(std::max<size_t>(files.size(), 1));
    std::vector<std::thread> result_image_writers;
    struct JoinAll {   // an exception in the image loop must not leave joinable threads behind (std::terminate)
        std::function<void()> f;
        ~JoinAll() { f(); }
    } join_all{[&] {
        full_image_read_requests.disable(); full_image_read_results.disable(); result_image_write_requests.disable(); result_image_write_results.disable();
        for (std::thread& t : full_image_readers) if (t.joinable()) t.join();
        for (std::thread& t : result_image_writers) if (t.joinable()) t.join();
    }};
    for (int i = 0; i < result_image_writer_count; ++i) {
        result_image_writers.push_back(std::thread([&]() {
            result_image_type result_image;
            dlib::matrix<rgb_alpha_pixel> rgba_label_image;
            while (result_image_write_requests.dequeue(result_image)) {
                bool ok = true;
                try {
                    resize_label_image(result_image.label_image, result_image.original_width, result_image.original_height);
                    index_label_image_to_rgba_label_image(result_image.label_image, rgba_label_image, anno_classes);
                    save_png(rgba_label_image, result_image.filename);
                } catch (std::exception& e) { std::cerr << e.what() << std::endl; ok = false; }
                result_image_write_results.enqueue(ok);
            }
        }));
    }

    const int min_input_dimension = NetPimpl::TrainingNet::GetRequiredInputDimension();
    tiling::parameters tiling_parameters;
    tiling_parameters.max_tile_width = options.tile_max_width;
    tiling_parameters.max_tile_height = options.tile_max_height;
    tiling_parameters.overlap_x = min_input_dimension;
    tiling_parameters.overlap_y = min_input_dimension;
    if (tiling_parameters.max_tile_width < min_input_dimension || tiling_parameters.max_tile_height < min_input_dimension)
        throw std::runtime_error("the maximum tile size must not be smaller than the net's required input dimension");

    confusion_matrix_type confusion_matrix_per_pixel, confusion_matrix_per_region;   // first index: ground truth, second index: predicted
    init_confusion_matrix(confusion_matrix_per_pixel, anno_classes.size());
    init_confusion_matrix(confusion_matrix_per_region, anno_classes.size());
    size_t ground_truth_count = 0;

    const auto t0 = std::chrono::steady_clock::now();
    update_confusion_matrix_per_region_temp_type update_confusion_matrix_per_region_temp;
    std::chrono::microseconds total_time_spent_in_actual_inference(0);
    std::chrono::microseconds total_time_spent_in_actual_inference_excluding_first_image(0);
    std::chrono::microseconds max_time_spent_in_actual_inference_per_image_excluding_first_image(0);
    bool write_failed = false;

    for (size_t i = 0, end = files.size(); i < end; ++i) {
        std::cout << "\rProcessing image " << (i + 1) << " of " << end << "...";
        sample_type sample;
        result_image_type result_image;
        full_image_read_results.dequeue(sample);
        if (!sample.error.empty()) throw std::runtime_error(sample.error);
        const auto& input_image = sample.input_image;
        result_image.filename = sample.image_filenames.image_filename + "_result.png";
        result_image.label_image.set_size(input_image.nr(), input_image.nc());
        result_image.original_width = sample.original_width;
        result_image.original_height = sample.original_height;

        const auto t0 = std::chrono::steady_clock::now();
        annonet_infer(net, sample.input_image, result_image.label_image, temp, gains, detection_levels, tiling_parameters);
        const auto t1 = std::chrono::steady_clock::now();

        const auto duration_us = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0);
        total_time_spent_in_actual_inference += duration_us;
        if (i > 0) {
            total_time_spent_in_actual_inference_excluding_first_image += duration_us;
            max_time_spent_in_actual_inference_per_image_excluding_first_image = std::max(max_time_spent_in_actual_inference_per_image_excluding_first_image, duration_us);
        }
        for (const auto& labeled_points : sample.labeled_points_by_class) {
            const uint16_t ground_truth_value = labeled_points.first;
            for (const dlib::point& point : labeled_points.second) {
                const uint16_t predicted_value = result_image.label_image(point.y(), point.x());
                if (predicted_value < anno_classes.size()) ++confusion_matrix_per_pixel[ground_truth_value][predicted_value];
            }
            ground_truth_count += labeled_points.second.size();
        }
        update_confusion_matrix_per_region(confusion_matrix_per_region, sample.labeled_points_by_class, sample.label_image, result_image.label_image, update_confusion_matrix_per_region_temp);
        result_image_write_requests.enqueue(result_image);
    }

    const auto t1 = std::chrono::steady_clock::now();
    std::cout << "\nAll " << files.size() << " images processed in " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() / 1000.0 << " seconds!"
              << " (actual inference: " << total_time_spent_in_actual_inference.count() / 1000000.0 << " seconds)" << std::endl;
    if (files.size() > 1) {
        std::cout << "Processing time excluding the first image: "
                  << "average = " << total_time_spent_in_actual_inference_excluding_first_image.count() / 1000.0 / (files.size() - 1) << " ms, "
                  << "max = " << max_time_spent_in_actual_inference_per_image_excluding_first_image.count() / 1000.0 << " ms" << std::endl;
    }
    for (size_t i = 0, end = files.size(); i < end; ++i) {
        bool ok = true;
        result_image_write_results.dequeue(ok);
        write_failed = write_failed || !ok;
    }
    if (write_failed) throw std::runtime_error("some result images could not be written");
    std::cout << "All result images written!" << std::endl;

    full_image_read_requests.disable();
    result_image_write_requests.disable();
    for (std::thread& image_reader : full_image_readers) image_reader.join();
    for (std::thread& image_writer : result_image_writers) image_writer.join();

    if (ground_truth_count) {
        std::cout << std::endl << "Confusion matrix per pixel:" << std::endl;
        print_confusion_matrix(confusion_matrix_per_pixel, anno_classes);
        std::cout << std::endl << "Confusion matrix per region (two-way):" << std::endl;
        print_confusion_matrix(confusion_matrix_per_region, anno_classes);
    }
    return 0;
} catch (std::exception& e) {
    cout << e.what() << endl;
    return 1;
}
