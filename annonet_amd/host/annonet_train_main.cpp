// annonet_train_main.cpp — the reference's training tool (annonet_train_main.cpp:260-644) on the drop-in headers.
// Same options, same echo lines, same trainer configuration order (:400-410), same mini-batch loop (:583-614), same annonet.dnn
// envelope and 10-minute trainer state file.  What differs:
//   * DEFAULT DATA PATH: the loader threads decode full images through the LRU cache as in the reference, but push crop SPECS (the
//     random draws of randomly_crop_image) instead of crops; the full images are uploaded to HBM once (NetPimpl::Dataset) and every
//     mini-batch is cut on the device (TrainingNet::StartTrainingOnCrops): no per-step host crop work, no per-step upload.
//     --host-crops restores the reference's data path (crops cut by the loader threads, StartTraining on host matrices);
//   * --devices 0,1,...: data-parallel training from this one process (anh_set_devices; host crops, split along N inside StartTraining);
//   * --ignore-large-nonzero-regions-by-*: the reference parses these and defines the filter (:426-502) but never calls it; here it runs
//     when an option is given;
//   * cxxopts / dlib / lru-timday are absent: own option reader, annonet_host.h / annonet_train_host.h stand in.
#define ANNONET_HIP_NO_DLIB
#include "annonet_train_host.h"

using namespace std;
using namespace dlib;

struct Options {
    double initial_downscaling_factor = 1.0, further_downscaling_factor = 1.0;
    std::string input_directory;
    bool allow_flip_upside_down = false, allow_flip_left_right = false, allow_random_color_offset = false, no_empty_label_image_warning = false, host_crops = false;
    double brightness_probability = 0.0, brightness_sigma = 0.1, noise_level_stddev = 0.0;
    std::vector<uint16_t> ignore_class;
    double ignore_by_area = std::numeric_limits<double>::infinity(), ignore_by_width = std::numeric_limits<double>::infinity(), ignore_by_height = std::numeric_limits<double>::infinity();
    double class_weight = 0.5, image_weight = 0.5;
    size_t minibatch_size = 100;
    double input_dimension_multiplier = 3.0, net_width_scaler = 1.0;
    int net_width_min_filter_count = 1;
    double initial_learning_rate = 0.1, learning_rate_shrink_factor = 0.1, min_learning_rate = 1e-6;
    size_t save_interval = 1000;
    double relative_training_length = 2.0;
    bool has_max_total_steps = false; size_t max_total_steps = 0;
    int cached_image_count = 8;
    unsigned data_loader_thread_count = std::max(1u, std::thread::hardware_concurrency());
    bool has_primary_device = false; int primary_device = 0;
    std::vector<int> devices;
    std::string precision = "bf16";
    uint64_t seed = 0; bool has_seed = false;
};

static const char* kHelp =
    "Train semantic-segmentation networks using data generated in anno\nUsage:\n  annonet_train_hip [OPTION...] <input-directory>\n\n"
    "  -d, --initial-downscaling-factor arg  -f, --further-downscaling-factor arg  -i, --input-directory arg\n"
    "  -u, --allow-flip-upside-down  -l, --allow-flip-left-right  -o, --allow-random-color-offset\n"
    "      --multiplicative-brightness-change-probability arg  --multiplicative-brightness-change-sigma arg  -n, --noise-level-stddev arg\n"
    "      --ignore-class arg  --ignore-large-nonzero-regions-by-area|-width|-height arg  --class-weight arg  --image-weight arg\n"
    "  -b, --minibatch-size arg  --input-dimension-multiplier arg  --net-width-scaler arg  --net-width-min-filter-count arg\n"
    "      --initial-learning-rate arg  --learning-rate-shrink-factor arg  --min-learning-rate arg  --save-interval arg\n"
    "  -t, --relative-training-length arg  --max-total-steps arg  -c, --cached-image-count arg  --data-loader-thread-count arg\n"
    "      --no-empty-label-image-warning  --primary-cuda-device arg\n"
    "  extensions: --host-crops  --devices 0,1,...  --precision bf16|fp32  --seed arg\n";

Options parse_options(int argc, char** argv) {
    Options o;
    auto need = [&](int& i) -> std::string { if (i + 1 >= argc) throw std::runtime_error(std::string("Option '") + argv[i] + "' is missing an argument"); return argv[++i]; };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-d" || a == "--initial-downscaling-factor") o.initial_downscaling_factor = std::stod(need(i));
        else if (a == "-f" || a == "--further-downscaling-factor") o.further_downscaling_factor = std::stod(need(i));
        else if (a == "-i" || a == "--input-directory") o.input_directory = need(i);
        else if (a == "-u" || a == "--allow-flip-upside-down") o.allow_flip_upside_down = true;
        else if (a == "-l" || a == "--allow-flip-left-right") o.allow_flip_left_right = true;
        else if (a == "-o" || a == "--allow-random-color-offset") o.allow_random_color_offset = true;
        else if (a == "--multiplicative-brightness-change-probability") o.brightness_probability = std::stod(need(i));
        else if (a == "--multiplicative-brightness-change-sigma") o.brightness_sigma = std::stod(need(i));
        else if (a == "-n" || a == "--noise-level-stddev") o.noise_level_stddev = std::stod(need(i));
        else if (a == "--ignore-class") o.ignore_class.push_back((uint16_t)std::stoul(need(i)));
        else if (a == "--ignore-large-nonzero-regions-by-area") o.ignore_by_area = std::stod(need(i));
        else if (a == "--ignore-large-nonzero-regions-by-width") o.ignore_by_width = std::stod(need(i));
        else if (a == "--ignore-large-nonzero-regions-by-height") o.ignore_by_height = std::stod(need(i));
        else if (a == "--class-weight") o.class_weight = std::stod(need(i));
        else if (a == "--image-weight") o.image_weight = std::stod(need(i));
        else if (a == "-b" || a == "--minibatch-size") o.minibatch_size = std::stoul(need(i));
        else if (a == "--input-dimension-multiplier") o.input_dimension_multiplier = std::stod(need(i));
        else if (a == "--net-width-scaler") o.net_width_scaler = std::stod(need(i));
        else if (a == "--net-width-min-filter-count") o.net_width_min_filter_count = std::stoi(need(i));
        else if (a == "--initial-learning-rate") o.initial_learning_rate = std::stod(need(i));
        else if (a == "--learning-rate-shrink-factor") o.learning_rate_shrink_factor = std::stod(need(i));
        else if (a == "--min-learning-rate") o.min_learning_rate = std::stod(need(i));
        else if (a == "--save-interval") o.save_interval = std::stoul(need(i));
        else if (a == "-t" || a == "--relative-training-length") o.relative_training_length = std::stod(need(i));
        else if (a == "--max-total-steps") { o.max_total_steps = std::stoul(need(i)); o.has_max_total_steps = true; }
        else if (a == "-c" || a == "--cached-image-count") o.cached_image_count = std::stoi(need(i));
        else if (a == "--data-loader-thread-count") o.data_loader_thread_count = (unsigned)std::stoul(need(i));
        else if (a == "--no-empty-label-image-warning") o.no_empty_label_image_warning = true;
        else if (a == "--primary-cuda-device") { o.primary_device = std::stoi(need(i)); o.has_primary_device = true; }
        else if (a == "--host-crops") o.host_crops = true;
        else if (a == "--precision") o.precision = need(i);
        else if (a == "--seed") { o.seed = std::stoull(need(i)); o.has_seed = true; }
        else if (a == "--devices") { std::stringstream ss(need(i)); std::string item; while (std::getline(ss, item, ',')) o.devices.push_back(std::stoi(item)); }
        else if (!a.empty() && a[0] == '-') throw std::runtime_error("Option '" + a + "' does not exist");
        else if (o.input_directory.empty()) o.input_directory = a;
        else throw std::runtime_error("Unexpected argument " + a);
    }
    if (o.input_directory.empty()) throw std::runtime_error("Option 'input-directory' is required but not present");
    if (o.precision != "bf16" && o.precision != "fp32") throw std::runtime_error("--precision must be fp32 or bf16");
    return o;
}

std::string read_anno_classes_file(const std::string& folder) {   // :236-256: the file must be in the root of the input directory
    const std::string path = (std::filesystem::path(folder) / "anno_classes.json").string();
    if (!std::ifstream(path)) {
        std::cout << "Warning: no anno_classes.json file found in " + folder << std::endl;
        std::cout << " --> Using the default anno classes" << std::endl;
        return "";
    }
    return annonet_io::slurp(path);
}

int main(int argc, char** argv) try {
    if (argc == 1) {
        cout << "To run this program you need data annotated using the anno program." << endl << endl;
        cout << "You call this program like this: " << endl << "./annonet_train_hip /path/to/anno/data" << endl;
        return 1;
    }
    Options options;
    try {
        options = parse_options(argc, argv);
        std::cout << "Input directory = " << options.input_directory << std::endl;
        std::cout << "Initial downscaling factor = " << options.initial_downscaling_factor << std::endl;
        std::cout << "Further downscaling factor = " << options.further_downscaling_factor << std::endl;
        if (options.initial_downscaling_factor <= 0.0 || options.further_downscaling_factor <= 0.0) throw std::runtime_error("The downscaling factors have to be strictly positive.");
    } catch (std::exception& e) {
        cerr << e.what() << std::endl << std::endl << kHelp << std::endl;
        return 2;
    }

    const double initial_downscaling_factor = options.initial_downscaling_factor;
    const double further_downscaling_factor = options.further_downscaling_factor;
    const auto minibatch_size = options.minibatch_size;
    const auto relative_training_length = std::max(0.01, options.relative_training_length);
    const auto data_loader_thread_count = std::max(1U, options.data_loader_thread_count);
    const bool warn_about_empty_label_images = !options.no_empty_label_image_warning;

    std::cout << "Allow flipping input images upside down = " << (options.allow_flip_upside_down ? "yes" : "no") << std::endl;
    std::cout << "Minibatch size = " << minibatch_size << std::endl;
    std::cout << "Net width scaler = " << options.net_width_scaler << ", min filter count = " << options.net_width_min_filter_count << std::endl;
    std::cout << "Initial learning rate = " << options.initial_learning_rate << std::endl;
    std::cout << "Learning rate shrink factor = " << options.learning_rate_shrink_factor << std::endl;
    std::cout << "Min learning rate = " << options.min_learning_rate << std::endl;
    std::cout << "Save interval = " << options.save_interval << std::endl;
    std::cout << "Relative training length = " << relative_training_length << std::endl;
    std::cout << "Cached image count = " << options.cached_image_count << std::endl;
    std::cout << "Data loader thread count = " << data_loader_thread_count << std::endl;
    if (!options.ignore_class.empty()) {
        std::cout << "Classes to ignore =";
        for (uint16_t class_to_ignore : options.ignore_class) std::cout << " " << class_to_ignore;
        std::cout << std::endl;
    }

    const int required_input_dimension = NetPimpl::TrainingNet::GetRequiredInputDimension();
    std::cout << "Required input dimension = " << required_input_dimension << std::endl;
    const int requested_input_dimension = static_cast<int>(std::round(options.input_dimension_multiplier * required_input_dimension));
    std::cout << "Requested input dimension = " << requested_input_dimension << std::endl;
    const int actual_input_dimension = NetPimpl::RuntimeNet::GetRecommendedInputDimension(requested_input_dimension);
    std::cout << "Actual input dimension = " << actual_input_dimension << std::endl;

    const auto anno_classes_json = read_anno_classes_file(options.input_directory);
    const auto anno_classes = parse_anno_classes(anno_classes_json);

    const unsigned long iterations_without_progress_threshold = static_cast<unsigned long>(std::round(relative_training_length * 2000));
    const unsigned long previous_loss_values_dump_amount = static_cast<unsigned long>(std::round(relative_training_length * 400));
    const unsigned long batch_normalization_running_stats_window_size = static_cast<unsigned long>(std::round(relative_training_length * 100));

    if (options.has_primary_device) NetPimpl::check(anh_set_device(options.primary_device));   // dlib::cuda::set_device (:392-394)
    if (!options.devices.empty()) NetPimpl::SetDevices(options.devices);
    const bool device_crops = !options.host_crops && options.devices.size() <= 1;
    std::cout << "Mini-batches are cut " << (device_crops ? "on the device from HBM-resident full images" : "on the host by the loader threads") << std::endl;

    NetPimpl::TrainingNet training_net;
    std::vector<NetPimpl::input_type> samples;
    std::vector<NetPimpl::training_label_type> labels;

    if (options.precision == "fp32") NetPimpl::check(anh_trainer_set_precision(training_net.handle(), ANH_FP32));
    if (options.has_seed) NetPimpl::check(anh_trainer_set_seed(training_net.handle(), options.seed));
    training_net.Initialize();
    training_net.SetNetWidth(options.net_width_scaler, options.net_width_min_filter_count);
    training_net.SetSynchronizationFile("annonet_trainer_state_file.dat", std::chrono::seconds(10 * 60));
    training_net.BeVerbose();
    training_net.SetClassCount(anno_classes.size());
    training_net.SetLearningRate(options.initial_learning_rate);
    training_net.SetLearningRateShrinkFactor(options.learning_rate_shrink_factor);
    training_net.SetIterationsWithoutProgressThreshold(iterations_without_progress_threshold);
    training_net.SetPreviousLossValuesDumpAmount(previous_loss_values_dump_amount);
    training_net.SetAllBatchNormalizationRunningStatsWindowSizes(batch_normalization_running_stats_window_size);

    cout << "\nSCANNING ANNO DATASET\n" << endl;
    const auto image_files = find_image_files(options.input_directory, true);
    cout << "images in dataset: " << image_files.size() << endl;
    if (image_files.size() == 0) {
        cout << "Didn't find an anno dataset. " << endl;
        return 1;
    }

    const bool filter_large_regions = std::isfinite(options.ignore_by_area) || std::isfinite(options.ignore_by_width) || std::isfinite(options.ignore_by_height);
    shared_lru_cache<image_filenames_type, std::shared_ptr<sample_type>, image_filenames_hash> full_images_cache(
        [&](const image_filenames_type& image_filenames) {   // :504-510
            auto sample = std::make_shared<sample_type>();
            *sample = read_sample(image_filenames, anno_classes, true, initial_downscaling_factor);
            if (sample->error.empty()) {
                ignore_classes_to_ignore(*sample, options.ignore_class);
                if (filter_large_regions) ignore_large_nonzero_regions(*sample, options.ignore_by_area, options.ignore_by_width, options.ignore_by_height);
            }
            return sample;
        }, (size_t)std::max(1, options.cached_image_count));

    cout << endl << "Now training..." << endl;

    augmentation_options aug;
    aug.further_downscaling_factor = further_downscaling_factor; aug.class_weight = options.class_weight; aug.image_weight = options.image_weight;
    aug.allow_flip_left_right = options.allow_flip_left_right; aug.allow_flip_upside_down = options.allow_flip_upside_down;
    aug.allow_random_color_offset = options.allow_random_color_offset;
    aug.multiplicative_brightness_change_probability = options.brightness_probability; aug.multiplicative_brightness_change_sigma = options.brightness_sigma;
    aug.noise_level_stddev = options.noise_level_stddev;

    // Loader threads (:516-553).  A work item is a crop (host path) or a crop spec plus the decoded full image it refers to (device
    // path: the trainer thread uploads a full image to HBM the first time one of its specs arrives).
    struct work_item { crop c; std::shared_ptr<sample_type> full; };
    anh_host::pipe<work_item> data(2 * minibatch_size);
    std::atomic<bool> data_enabled{true};
    auto pull_crops = [&](uint64_t seed) {
        host_rand rnd((options.has_seed ? options.seed : (uint64_t)time(0)) + seed);
        work_item item;
        while (data_enabled) {
            item.c.error.clear();
            item.c.warning.clear();
            item.c.is_spec = false;
            const size_t index = rnd.get_random_32bit_number() % image_files.size();
            const auto& image_filenames = image_files[index];
            const std::shared_ptr<sample_type> ground_truth_sample = full_images_cache(image_filenames);
            if (!ground_truth_sample->error.empty()) item.c.error = ground_truth_sample->error;
            else if (ground_truth_sample->labeled_points_by_class.empty()) item.c.warning = "Warning: no labeled points in " + ground_truth_sample->image_filenames.label_filename;
            else if (device_crops) {
                item.c.spec = draw_crop_spec(actual_input_dimension, *ground_truth_sample, -1, rnd, aug);
                item.c.is_spec = true;
                item.full = ground_truth_sample;
            } else {
                try { randomly_crop_image(actual_input_dimension, *ground_truth_sample, item.c, rnd, aug); }
                catch (std::exception& e) { item.c.error = e.what(); }
            }
            if (!data.enqueue(item)) break;
        }
    };
    std::vector<std::thread> data_loaders;
    struct JoinAll {
        std::function<void()> f;
        ~JoinAll() { f(); }
    } join_all{[&] { data_enabled = false; data.disable(); for (std::thread& t : data_loaders) if (t.joinable()) t.join(); }};
    for (unsigned int i = 0; i < data_loader_thread_count; ++i) data_loaders.push_back(std::thread([&pull_crops, i]() { pull_crops(i); }));

    size_t minibatch = 0;
    const auto save_inference_net = [&]() {   // :557-565
        const NetPimpl::RuntimeNet runtime_net = training_net.GetRuntimeNet();
        std::ostringstream serialized;
        runtime_net.Serialize(serialized);
        cout << "saving network" << endl;
        const std::string blob = serialized.str();
        void* file = nullptr; size_t file_size = 0;
        NetPimpl::check(anh_dnn_envelope_pack(anno_classes_json.data(), anno_classes_json.size(), initial_downscaling_factor * further_downscaling_factor, blob.data(), blob.size(), &file, &file_size));
        std::ofstream f("annonet.dnn", std::ios::binary | std::ios::trunc);
        f.write(static_cast<const char*>(file), (std::streamsize)file_size);
        anh_free(file);
        if (!f) throw std::runtime_error("Unable to write annonet.dnn");
    };

    std::set<std::string> warnings_already_printed;
    const auto should_continue_training = [&]() {   // :569-577
        if (training_net.GetLearningRate() < options.min_learning_rate) return false;
        if (options.has_max_total_steps && minibatch >= options.max_total_steps) return false;
        return true;
    };

    NetPimpl::Dataset dataset;                                   // full images resident in HBM (device path)
    std::unordered_map<std::string, int> resident;               // image file -> index in `dataset`
    std::vector<anh_crop_spec> specs;
    int return_value = 0;
    try {
        while (should_continue_training()) {   // :583-614
            samples.clear();
            labels.clear();
            specs.clear();
            work_item item;
            while ((device_crops ? specs.size() : samples.size()) < minibatch_size) {
                if (!data.dequeue(item)) throw std::runtime_error("the loader threads stopped");
                if (!item.c.error.empty()) throw std::runtime_error(item.c.error);
                else if (!item.c.warning.empty()) {
                    if (warn_about_empty_label_images && warnings_already_printed.find(item.c.warning) == warnings_already_printed.end()) {
                        std::cout << item.c.warning << std::endl;
                        warnings_already_printed.insert(item.c.warning);
                    }
                } else if (item.c.is_spec) {
                    const std::string& key = item.full->image_filenames.image_filename;
                    auto it = resident.find(key);
                    if (it == resident.end()) it = resident.emplace(key, dataset.Add(item.full->input_image, item.full->label_image)).first;
                    item.c.spec.image = it->second;
                    specs.push_back(item.c.spec);
                } else {
                    samples.push_back(std::move(item.c.input_image));
                    labels.push_back(std::move(item.c.label_image));
                }
            }
            if (device_crops) training_net.StartTrainingOnCrops(dataset, specs, actual_input_dimension, options.class_weight, options.image_weight);
            else training_net.StartTraining(samples, labels);
            if (minibatch++ % options.save_interval == 0) save_inference_net();
        }
    } catch (std::exception& e) {
        cout << e.what() << endl;
        return_value = 2;
        data_enabled = false;
        data.disable();
        for (std::thread& t : data_loaders) if (t.joinable()) t.join();
        exit(return_value);
    }
    data_enabled = false;
    data.disable();
    for (std::thread& t : data_loaders) if (t.joinable()) t.join();
    if (return_value == 0) save_inference_net();
    std::cout << "steps: " << minibatch << ", full images decoded: " << full_images_cache.misses() << ", cache hits: " << full_images_cache.hits()
              << ", evictions: " << full_images_cache.evictions() << ", images resident in HBM: " << resident.size() << std::endl;
    return return_value;
} catch (std::exception& e) {
    cout << e.what() << endl;
    return 1;
}
