// infer_tool.cpp -> annonet_amd/lib/annonet_infer_hip: the job of the reference's inference tool (annonet_infer_main.cpp:283-538) on the
// drop-in headers, as a three-stage pipeline built from this repository's own parts:
//
//     ImageReaders (N threads: read_sample)  ->  Segmenter (this thread: annonet_infer() on the GPU(s))  ->  ResultWriters (M threads: PNG)
//
// Kept from the reference because they are its observable behaviour: the option names and defaults (:308-316), the lines it prints
// (input directory, gains / detection levels, scan summary, "All N images processed in ... (actual inference: ...)", "Processing time
// excluding the first image: average = ... ms, max = ... ms", both confusion matrices), the files it reads (annonet.dnn,
// <image>_mask.png) and writes (<image>_result.png), and its exit codes (2 for option errors, 1 otherwise).
// Extensions: --precision fp32|bf16, --devices 0,1,... (tile lists sharded over several GPUs from this one process), --dnn <file>.
#define ANNONET_HIP_NO_DLIB
#include "../../include/annonet_infer_hip.h"
#include "annonet_host.h"

#include <functional>
#include <map>

namespace {

// ---- command line ---------------------------------------------------------------------------------------------------
struct Settings {
    std::string directory, dnn_file = "annonet.dnn", precision = "bf16";
    std::vector<std::string> gain_args, detection_args;
    std::vector<int> devices;
    int max_tile_w = 1024, max_tile_h = 1024;   // the reference's GPU-build defaults (:300-303)
    int readers = (int)std::max(1u, std::thread::hardware_concurrency()), writers = (int)std::max(1u, std::thread::hardware_concurrency());
};

const char* usage_text() {
    return "Do inference using trained semantic-segmentation networks\nUsage:\n  annonet_infer_hip [OPTION...] <input-directory>\n\n"
           "  -i, --input-directory arg            Input image directory\n"
           "  -g, --gain arg                       Supply a class-specific gain, for example: 1:-0.5\n"
           "  -d, --detection arg                  Supply a class-specific detection level that _comes on top of gain_, for example: 1:1.5\n"
           "  -w, --tile-max-width arg             Set max tile width (default: 1024)\n"
           "  -h, --tile-max-height arg            Set max tile height (default: 1024)\n"
           "      --full-image-reader-thread-count arg\n"
           "      --result-image-writer-thread-count arg\n"
           "      --precision fp32|bf16            fp32 = bit-exact parity mode, bf16 = MFMA throughput mode (default)\n"
           "      --devices 0,1,...                GPUs this process drives (tile lists are sharded over them)\n"
           "      --dnn file                       trained net (default: annonet.dnn)\n";
}

Settings read_command_line(int argc, char** argv) {
    Settings s;
    std::map<std::string, std::function<void(const std::string&)>> with_value = {
        {"-i", [&](const std::string& v) { s.directory = v; }}, {"--input-directory", [&](const std::string& v) { s.directory = v; }},
        {"-g", [&](const std::string& v) { s.gain_args.push_back(v); }}, {"--gain", [&](const std::string& v) { s.gain_args.push_back(v); }},
        {"-d", [&](const std::string& v) { s.detection_args.push_back(v); }}, {"--detection", [&](const std::string& v) { s.detection_args.push_back(v); }},
        {"-w", [&](const std::string& v) { s.max_tile_w = std::stoi(v); }}, {"--tile-max-width", [&](const std::string& v) { s.max_tile_w = std::stoi(v); }},
        {"-h", [&](const std::string& v) { s.max_tile_h = std::stoi(v); }}, {"--tile-max-height", [&](const std::string& v) { s.max_tile_h = std::stoi(v); }},
        {"--full-image-reader-thread-count", [&](const std::string& v) { s.readers = std::stoi(v); }},
        {"--result-image-writer-thread-count", [&](const std::string& v) { s.writers = std::stoi(v); }},
        {"--precision", [&](const std::string& v) { s.precision = v; }}, {"--dnn", [&](const std::string& v) { s.dnn_file = v; }},
        {"--devices", [&](const std::string& v) { std::stringstream list(v); std::string item; while (std::getline(list, item, ',')) s.devices.push_back(std::stoi(item)); }},
    };
    for (int i = 1; i < argc; ++i) {
        const std::string word = argv[i];
        auto option = with_value.find(word);
        if (option != with_value.end()) {
            if (i + 1 >= argc) throw std::runtime_error("Option '" + word + "' is missing an argument");
            option->second(argv[++i]);
        } else if (!word.empty() && word[0] == '-') throw std::runtime_error("Option '" + word + "' does not exist");
        else if (s.directory.empty()) s.directory = word;
        else throw std::runtime_error("Unexpected argument " + word);
    }
    if (s.directory.empty()) throw std::runtime_error("Option 'input-directory' is required but not present");
    if (s.precision != "bf16" && s.precision != "fp32") throw std::runtime_error("--precision must be fp32 or bf16");
    return s;
}

// "index:value" pairs (-g 1:-0.5) -> one value per class, 0 where nothing was given (annonet_infer_main.cpp:36-63)
std::vector<double> per_class_values(const std::vector<std::string>& args, size_t class_count) {
    std::vector<double> values(class_count, 0.0);
    for (const std::string& arg : args) {
        const size_t colon = arg.find(':');
        if (colon == std::string::npos || colon == 0 || colon + 1 >= arg.size()) throw std::runtime_error("The gains must be supplied in the format index:gain (e.g., 1:-0.5)");
        const unsigned long index = std::stoul(arg.substr(0, colon));
        if (index >= class_count) {
            std::ostringstream message;
            message << "Can't define class-specific value for index " << index << " when there are only " << class_count << " classes";
            throw std::runtime_error(message.str());
        }
        values[index] = std::stod(arg.substr(colon + 1));
    }
    return values;
}

void echo_per_class(const char* caption, const std::vector<double>& values) {
    std::cout << caption;
    for (size_t k = 0; k < values.size(); ++k) std::cout << " " << k << ":" << values[k];
    std::cout << std::endl;
}

// ---- the trained net: annonet.dnn = (anno classes json, downscaling factor, serialized RuntimeNet) ---------------------------------
struct TrainedNet {
    NetPimpl::RuntimeNet net;
    std::vector<AnnoClass> classes;
    double downscaling = 1.0;

    static TrainedNet load(const std::string& path, bool fp32) {
        TrainedNet t;
        const std::string bytes = annonet_io::slurp(path);
        char* json = nullptr; void* blob = nullptr; size_t json_size = 0, blob_size = 0;
        NetPimpl::check(anh_dnn_envelope_unpack(bytes.data(), bytes.size(), &json, &json_size, &t.downscaling, &blob, &blob_size));
        const std::string classes_json(json, json_size), net_bytes(static_cast<const char*>(blob), blob_size);
        anh_free(json); anh_free(blob);
        std::cout << "Deserializing annonet, downscaling factor = " << t.downscaling << std::endl;
        std::istringstream stream(net_bytes);
        t.net.Deserialize(stream, fp32 ? ANH_FP32 : ANH_BF16);
        t.classes = parse_anno_classes(classes_json);
        if (t.classes.size() < 2) throw std::runtime_error("at least two classes are needed");
        anh_net_config cfg;
        NetPimpl::check(anh_runtime_config(t.net.handle(), &cfg));
        if ((size_t)cfg.classes != t.classes.size()) throw std::runtime_error("the net's class count differs from the anno classes of the .dnn file");
        return t;
    }
};

// ---- stage 1: readers ----------------------------------------------------------------------------------------------------
class ImageReaders {
  public:
    ImageReaders(const std::vector<image_filenames_type>& files, int threads, const std::vector<AnnoClass>& classes, double downscaling)
        : todo_(std::max<size_t>(files.size(), 1)), done_((size_t)std::max(threads, 1)) {
        for (const auto& f : files) todo_.enqueue(f);
        for (int i = 0; i < std::max(threads, 1); ++i)
            pool_.emplace_back([this, &classes, downscaling] {
                image_filenames_type names;
                while (todo_.dequeue(names)) if (!done_.enqueue(read_sample(names, classes, false, downscaling))) break;
            });
    }
    ~ImageReaders() { todo_.disable(); done_.disable(); for (auto& t : pool_) if (t.joinable()) t.join(); }
    sample_type next() {   // samples arrive in completion order; each carries its own file names
        sample_type s;
        if (!done_.dequeue(s)) throw std::runtime_error("the image readers stopped");
        if (!s.error.empty()) throw std::runtime_error(s.error);
        return s;
    }

  private:
    anh_host::pipe<image_filenames_type> todo_;
    anh_host::pipe<sample_type> done_;
    std::vector<std::thread> pool_;
};

// ---- stage 3: writers ----------------------------------------------------------------------------------------------------
struct LabelMapToWrite {
    std::string path;
    int width = 0, height = 0;          // of the ORIGINAL image: the map is resized back before it is painted (:409-411)
    dlib::matrix<uint16_t> labels;
};

class ResultWriters {
  public:
    ResultWriters(int threads, size_t expected, const std::vector<AnnoClass>& classes) : jobs_((size_t)std::max(threads, 1)), outcomes_(std::max<size_t>(expected, 1)) {
        for (int i = 0; i < std::max(threads, 1); ++i)
            pool_.emplace_back([this, &classes] {
                LabelMapToWrite job;
                dlib::matrix<dlib::rgb_alpha_pixel> painted;
                while (jobs_.dequeue(job)) {
                    bool ok = true;
                    try {
                        resize_label_image(job.labels, job.width, job.height);
                        painted.set_size(job.labels.nr(), job.labels.nc());
                        auto out = painted.begin();
                        for (const uint16_t label : job.labels) *out++ = label < classes.size() ? classes[label].rgba_label : rgba_ignore_label;   // (65535: an all-NaN pixel)
                        save_png(painted, job.path);
                    } catch (std::exception& e) { std::cerr << e.what() << std::endl; ok = false; }
                    outcomes_.enqueue(ok);
                }
            });
    }
    ~ResultWriters() { jobs_.disable(); outcomes_.disable(); for (auto& t : pool_) if (t.joinable()) t.join(); }
    void submit(LabelMapToWrite job) { jobs_.enqueue(std::move(job)); ++submitted_; }
    void wait_for_all() {
        bool all_ok = true;
        for (size_t i = 0; i < submitted_; ++i) { bool ok = true; outcomes_.dequeue(ok); all_ok = all_ok && ok; }
        if (!all_ok) throw std::runtime_error("some result images could not be written");
    }

  private:
    anh_host::pipe<LabelMapToWrite> jobs_;
    anh_host::pipe<bool> outcomes_;
    std::vector<std::thread> pool_;
    size_t submitted_ = 0;
};

// ---- the tool's own clock (annonet_infer_main.cpp:442-444,466-480,498-507) -----------------------------------------------------------
struct InferenceClock {
    using us = std::chrono::microseconds;
    us all{0}, after_first{0}, slowest_after_first{0};
    size_t images = 0;
    void record(us d) {
        all += d;
        if (images++ > 0) { after_first += d; slowest_after_first = std::max(slowest_after_first, d); }   // the first image pays the warm-up
    }
    void report(double wall_seconds) const {
        std::cout << "\nAll " << images << " images processed in " << wall_seconds << " seconds! (actual inference: " << all.count() / 1000000.0 << " seconds)" << std::endl;
        if (images > 1)
            std::cout << "Processing time excluding the first image: average = " << after_first.count() / 1000.0 / (images - 1) << " ms, max = " << slowest_after_first.count() / 1000.0
                      << " ms" << std::endl;
    }
};

int run(const Settings& settings) {
    if (!settings.devices.empty()) NetPimpl::SetDevices(settings.devices);
    TrainedNet trained = TrainedNet::load(settings.dnn_file, settings.precision == "fp32");
    const std::vector<double> gains = per_class_values(settings.gain_args, trained.classes.size());
    const std::vector<double> detection_levels = per_class_values(settings.detection_args, trained.classes.size());
    echo_per_class("Using gains:", gains);
    echo_per_class("Using detection levels:", detection_levels);

    const std::vector<image_filenames_type> files = find_image_files(settings.directory, false);
    const int overlap = NetPimpl::TrainingNet::GetRequiredInputDimension();   // tiles overlap by one receptive field (:421-427)
    tiling::parameters tiles;
    tiles.max_tile_width = settings.max_tile_w; tiles.max_tile_height = settings.max_tile_h;
    tiles.overlap_x = tiles.overlap_y = overlap;
    if (tiles.max_tile_width < overlap || tiles.max_tile_height < overlap) throw std::runtime_error("the maximum tile size must not be smaller than the net's required input dimension");

    ConfusionMatrix per_pixel(trained.classes.size()), per_region(trained.classes.size());
    RegionScorer region_scorer;
    size_t labelled_pixels = 0;
    InferenceClock clock;
    annonet_infer_temp scratch;
    {
        ImageReaders readers(files, settings.readers, trained.classes, trained.downscaling);
        ResultWriters writers(settings.writers, files.size(), trained.classes);
        const auto started = std::chrono::steady_clock::now();
        for (size_t i = 0; i < files.size(); ++i) {
            std::cout << "\rProcessing image " << (i + 1) << " of " << files.size() << "...";
            const sample_type sample = readers.next();
            LabelMapToWrite result;
            result.path = sample.image_filenames.image_filename + "_result.png";
            result.width = sample.original_width; result.height = sample.original_height;

            const auto t0 = std::chrono::steady_clock::now();
            annonet_infer(trained.net, sample.input_image, result.labels, scratch, gains, detection_levels, tiles);
            clock.record(std::chrono::duration_cast<InferenceClock::us>(std::chrono::steady_clock::now() - t0));

            for (const auto& cls_points : sample.labeled_points_by_class) {   // per-pixel score on the annotated pixels (:482-490)
                for (const dlib::point& p : cls_points.second) per_pixel.add(cls_points.first, result.labels(p.y(), p.x()));
                labelled_pixels += cls_points.second.size();
            }
            region_scorer.score(per_region, sample, result.labels);
            writers.submit(std::move(result));
        }
        clock.report(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - started).count() / 1000.0);
        writers.wait_for_all();
        std::cout << "All result images written!" << std::endl;
    }
    if (labelled_pixels) {
        std::cout << std::endl << "Confusion matrix per pixel:" << std::endl;
        per_pixel.print(std::cout, trained.classes);
        std::cout << std::endl << "Confusion matrix per region (two-way):" << std::endl;
        per_region.print(std::cout, trained.classes);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 1) {
        std::cout << "You call this program like this: " << std::endl << "./annonet_infer_hip /path/to/image/data" << std::endl << std::endl
                  << "You will also need a trained 'annonet.dnn' file. " << std::endl << std::endl;
        return 1;
    }
    Settings settings;
    try {
        settings = read_command_line(argc, argv);
        std::cout << "Input directory = " << settings.directory << std::endl;
    } catch (std::exception& e) {
        std::cerr << e.what() << std::endl << std::endl << usage_text() << std::endl;
        return 2;
    }
    try { return run(settings); }
    catch (std::exception& e) { std::cout << e.what() << std::endl; return 1; }
}
