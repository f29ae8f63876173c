// host_selftest.cpp — command-line taps into annonet_host.h / image_io.h for the CPU test-suite (tests/test_host_programs.py):
// each subcommand runs one piece of the host logic on files and leaves its result where a numpy restatement can check it.
#define ANNONET_HIP_NO_DLIB
#include "annonet_train_host.h"

static dlib::matrix<uint16_t> read_u16(const std::string& path, long nr, long nc) {
    const std::string b = annonet_io::slurp(path);
    if (b.size() != (size_t)nr * nc * 2) throw std::runtime_error("raw label file has the wrong size");
    dlib::matrix<uint16_t> m;
    m.set_size(nr, nc);
    std::memcpy(&*m.begin(), b.data(), b.size());
    return m;
}
static void write_u16(const std::string& path, const dlib::matrix<uint16_t>& m) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char*>(&*m.begin()), (std::streamsize)(m.size() * 2));
}
static void print_matrix(const ConfusionMatrix& m) {
    for (size_t t = 0; t < m.classes(); ++t) { for (size_t p = 0; p < m.classes(); ++p) std::cout << m.at(t, p) << ' '; std::cout << '\n'; }
}

int main(int argc, char** argv) try {
    const std::vector<std::string> a(argv + 1, argv + argc);
    if (a.empty()) throw std::runtime_error("usage: host_selftest <png-roundtrip|classes|decode-mask|resize-labels|confusion|print-confusion> ...");
    if (a[0] == "png-roundtrip") {   // in out: decode -> encode (what load_image / save_png do)
        annonet_io::save_raster_png(annonet_io::load_raster(a.at(1)), a.at(2));
    } else if (a[0] == "classes") {   // json-file ("-" = the empty string): index r g b a name
        const std::string json = a.at(1) == "-" ? std::string() : annonet_io::slurp(a.at(1));
        for (const AnnoClass& c : parse_anno_classes(json))
            std::cout << c.index << ' ' << (int)c.rgba_label.red << ' ' << (int)c.rgba_label.green << ' ' << (int)c.rgba_label.blue << ' ' << (int)c.rgba_label.alpha << ' ' << c.classlabel << '\n';
    } else if (a[0] == "decode-mask") {   // mask.png classes.json out.raw: RGBA mask -> u16 index labels (+ count of labeled points per class on stdout)
        const auto classes = parse_anno_classes(a.at(2) == "-" ? std::string() : annonet_io::slurp(a.at(2)));
        dlib::matrix<dlib::rgb_alpha_pixel> rgba;
        load_rgba_image(rgba, a.at(1));
        sample_type s;
        decode_rgba_label_image(rgba, s, classes);
        write_u16(a.at(3), s.label_image);
        for (size_t k = 0; k < classes.size(); ++k) { auto it = s.labeled_points_by_class.find((uint16_t)k); std::cout << (it == s.labeled_points_by_class.end() ? 0 : it->second.size()) << ' '; }
        std::cout << '\n';
    } else if (a[0] == "resize-labels") {   // in.raw nr nc target_width target_height out.raw
        auto m = read_u16(a.at(1), std::stol(a.at(2)), std::stol(a.at(3)));
        resize_label_image(m, std::stoi(a.at(4)), std::stoi(a.at(5)));
        write_u16(a.at(6), m);
    } else if (a[0] == "confusion" || a[0] == "print-confusion") {   // gt.raw result.raw nr nc classes: the per-pixel and per-region matrices of one image
        const long nr = std::stol(a.at(3)), nc = std::stol(a.at(4));
        const size_t K = std::stoul(a.at(5));
        const auto gt = read_u16(a.at(1), nr, nc), res = read_u16(a.at(2), nr, nc);
        sample_type s;
        for (long r = 0; r < nr; ++r) for (long c = 0; c < nc; ++c) if (gt(r, c) != 65535) s.labeled_points_by_class[gt(r, c)].push_back(dlib::point(c, r));
        s.label_image = gt;
        ConfusionMatrix per_pixel(K), per_region(K);
        for (const auto& lp : s.labeled_points_by_class) for (const auto& p : lp.second) per_pixel.add(lp.first, res(p.y(), p.x()));
        RegionScorer scorer;
        scorer.score(per_region, s, res);
        if (a[0] == "confusion") { print_matrix(per_pixel); print_matrix(per_region); }
        else {
            std::vector<AnnoClass> classes;
            for (size_t k = 0; k < K; ++k) classes.push_back(AnnoClass((uint16_t)k, dlib::rgb_alpha_pixel(1, 1, 1, 1), "c"));
            per_pixel.print(std::cout, classes);
        }
    } else if (a[0] == "crop") {   // image.png mask.png left top dim flip_lr flip_ud gain factor off_r off_g off_b prefix: cut_crop_on_host -> prefix.{img,lab,w}.raw
        const auto classes = parse_anno_classes("");
        sample_type s = read_sample(image_filenames_type{a.at(1), a.at(2)}, classes, true, 1.0);
        if (!s.error.empty()) throw std::runtime_error(s.error);
        anh_crop_spec spec{};
        spec.left = std::stol(a.at(3)); spec.top = std::stol(a.at(4));
        const int dim = std::stoi(a.at(5));
        spec.flip_left_right = std::stoi(a.at(6)); spec.flip_upside_down = std::stoi(a.at(7));
        spec.brightness_change = std::stod(a.at(8)); spec.further_downscaling_factor = std::stod(a.at(9));
        for (int c = 0; c < 3; ++c) spec.color_offset[c] = std::stoi(a.at(10 + c));
        augmentation_options o;
        o.further_downscaling_factor = spec.further_downscaling_factor;
        crop c;
        host_rand rnd(1);
        cut_crop_on_host(dim, s, spec, c, rnd, o);
        std::ofstream fi(a.at(13) + ".img.raw", std::ios::binary), fl(a.at(13) + ".lab.raw", std::ios::binary), fw(a.at(13) + ".w.raw", std::ios::binary);
        fi.write(reinterpret_cast<const char*>(&*c.input_image.begin()), (std::streamsize)((size_t)dim * dim * NetPimpl::kInputChannels));
        for (const auto& wl : c.label_image) { fl.write(reinterpret_cast<const char*>(&wl.label), 2); fw.write(reinterpret_cast<const char*>(&wl.weight), 4); }
    } else if (a[0] == "noise") {   // level seed n: add_random_noise statistics on a mid-gray image: min max mean of the deltas
        NetPimpl::input_type img;
        const int n = std::stoi(a.at(3));
        img.set_size(n, n);
        std::memset(&*img.begin(), 128, (size_t)n * n * NetPimpl::kInputChannels);
        host_rand rnd(std::stoull(a.at(2)));
        add_random_noise(img, std::stod(a.at(1)), rnd);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(&*img.begin());
        long lo = 255, hi = -255; double sum = 0;
        for (size_t i = 0; i < (size_t)n * n * 3; ++i) { const long d = (long)p[i] - 128; lo = std::min(lo, d); hi = std::max(hi, d); sum += d; }
        std::cout << lo << ' ' << hi << ' ' << sum / ((double)n * n * 3) << '\n';
    } else if (a[0] == "lru") {   // capacity key key key ...: hits misses evictions and the final size of shared_lru_cache
        shared_lru_cache<std::string, int> cache([](const std::string& k) { return (int)k.size(); }, std::stoul(a.at(1)));
        for (size_t i = 2; i < a.size(); ++i) if (cache(a[i]) != (int)a[i].size()) throw std::runtime_error("cache returned a wrong value");
        std::cout << cache.hits() << ' ' << cache.misses() << ' ' << cache.evictions() << ' ' << cache.size() << '\n';
    } else if (a[0] == "color-offsets") {   // seed n: n draws of apply_random_color_offset's offsets
        host_rand rnd(std::stoull(a.at(1)));
        for (int i = 0; i < std::stoi(a.at(2)); ++i) { int off[3]; draw_color_offset(rnd, off); std::cout << off[0] << ' ' << off[1] << ' ' << off[2] << '\n'; }
    } else throw std::runtime_error("unknown subcommand " + a[0]);
    return 0;
} catch (std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
}
