// host_selftest.cpp — command-line taps into annonet_host.h / image_io.h for the CPU test-suite (tests/test_host_programs.py):
// each subcommand runs one piece of the host logic on files and leaves its result where a numpy restatement can check it.
#define ANNONET_HIP_NO_DLIB
#include "annonet_host.h"

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
static void print_matrix(const confusion_matrix_type& m) {
    for (const auto& row : m) { for (size_t v : row) std::cout << v << ' '; std::cout << '\n'; }
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
        confusion_matrix_type per_pixel, per_region;
        init_confusion_matrix(per_pixel, K); init_confusion_matrix(per_region, K);
        for (const auto& lp : s.labeled_points_by_class) for (const auto& p : lp.second) ++per_pixel[lp.first][res(p.y(), p.x())];
        update_confusion_matrix_per_region_temp_type temp;
        update_confusion_matrix_per_region(per_region, s.labeled_points_by_class, gt, res, temp);
        if (a[0] == "confusion") { print_matrix(per_pixel); print_matrix(per_region); }
        else {
            std::vector<AnnoClass> classes;
            for (size_t k = 0; k < K; ++k) classes.push_back(AnnoClass((uint16_t)k, dlib::rgb_alpha_pixel(1, 1, 1, 1), "c"));
            print_confusion_matrix(per_pixel, classes);
        }
    } else throw std::runtime_error("unknown subcommand " + a[0]);
    return 0;
} catch (std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
}
