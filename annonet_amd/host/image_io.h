// image_io.h — the image files annonet's mains touch, without libpng / libjpeg (absent from this image; zlib is present):
//   load_image(rgb / rgb_alpha matrix, file)  annonet.cpp:150,155   PNG (8-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced),
//                                                                    binary PNM (P5 / P6) and PAM (P7, RGB_ALPHA)
//   save_png(rgba matrix, file)               annonet_infer_main.cpp:413   8-bit RGBA PNG, filter 0, zlib default level
// JPEG inputs are refused with a message (the reference reads them through dlib + libjpeg).
#ifndef ANNONET_HIP_IMAGE_IO_H
#define ANNONET_HIP_IMAGE_IO_H

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace annonet_io {

struct Raster {   // 8-bit interleaved pixels
    int width = 0, height = 0, channels = 0;   // 1 gray, 2 gray+alpha, 3 RGB, 4 RGBA
    std::vector<uint8_t> data;
};

inline std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Unable to open file " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline Raster decode_png(const std::string& bytes, const std::string& name) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    const uint8_t* p = reinterpret_cast<const uint8_t*>(bytes.data());
    const size_t n = bytes.size();
    if (n < 8 || std::memcmp(p, sig, 8) != 0) throw std::runtime_error(name + ": not a PNG file");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, color = -1, interlace = 0;
    std::string idat;
    std::vector<uint8_t> palette, trns;
    bool end = false;
    while (!end && pos + 12 <= n) {
        const uint32_t len = be32(p + pos);
        if (pos + 12 + (size_t)len > n) throw std::runtime_error(name + ": truncated PNG chunk");
        const std::string type(reinterpret_cast<const char*>(p + pos + 4), 4);
        const uint8_t* body = p + pos + 8;
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error(name + ": bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12];
        } else if (type == "PLTE") palette.assign(body, body + len);
        else if (type == "tRNS") trns.assign(body, body + len);
        else if (type == "IDAT") idat.append(reinterpret_cast<const char*>(body), len);
        else if (type == "IEND") end = true;
        pos += 12 + (size_t)len;
    }
    if (color < 0 || w == 0 || h == 0) throw std::runtime_error(name + ": PNG without a header");
    if (depth != 8 || interlace != 0) throw std::runtime_error(name + ": only 8-bit non-interlaced PNG files are supported");
    int spp;   // samples per pixel in the file
    switch (color) { case 0: spp = 1; break; case 2: spp = 3; break; case 3: spp = 1; break; case 4: spp = 2; break; case 6: spp = 4; break;
        default: throw std::runtime_error(name + ": unknown PNG colour type"); }
    const size_t stride = (size_t)w * spp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, reinterpret_cast<const Bytef*>(idat.data()), (uLong)idat.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error(name + ": corrupt PNG data");
    std::vector<uint8_t> px(stride * h);
    for (uint32_t y = 0; y < h; ++y) {   // undo the scanline filters (PNG spec 9.2)
        const uint8_t* src = raw.data() + (stride + 1) * y;
        uint8_t* dst = px.data() + stride * y;
        const uint8_t* up = y ? dst - stride : nullptr;
        const int f = src[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)spp ? dst[i - spp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
            int pred = 0;
            switch (f) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error(name + ": bad PNG filter type");
            }
            dst[i] = (uint8_t)(src[1 + i] + pred);
        }
    }
    Raster r;
    r.width = (int)w; r.height = (int)h;
    if (color == 3) {   // palette -> RGB(A)
        const bool alpha = !trns.empty();
        r.channels = alpha ? 4 : 3;
        r.data.resize((size_t)w * h * r.channels);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const size_t idx = px[i];
            if (idx * 3 + 2 >= palette.size()) throw std::runtime_error(name + ": PNG palette index out of range");
            for (int c = 0; c < 3; ++c) r.data[i * r.channels + c] = palette[idx * 3 + c];
            if (alpha) r.data[i * 4 + 3] = idx < trns.size() ? trns[idx] : 255;
        }
    } else { r.channels = spp; r.data.swap(px); }
    return r;
}

inline std::string png_chunk(const char* type, const std::string& body) {
    std::string out;
    const uint32_t len = (uint32_t)body.size();
    for (int s = 24; s >= 0; s -= 8) out.push_back((char)(len >> s));
    std::string tb = std::string(type, 4) + body;
    out += tb;
    const uint32_t crc = (uint32_t)crc32(0L, reinterpret_cast<const Bytef*>(tb.data()), (uInt)tb.size());
    for (int s = 24; s >= 0; s -= 8) out.push_back((char)(crc >> s));
    return out;
}

inline std::string encode_png(const Raster& r) {
    const int color = r.channels == 1 ? 0 : r.channels == 2 ? 4 : r.channels == 3 ? 2 : 6;
    std::string ihdr(13, '\0');
    for (int i = 0; i < 4; ++i) { ihdr[i] = (char)((uint32_t)r.width >> (24 - 8 * i)); ihdr[4 + i] = (char)((uint32_t)r.height >> (24 - 8 * i)); }
    ihdr[8] = 8; ihdr[9] = (char)color;
    const size_t stride = (size_t)r.width * r.channels;
    std::string raw;
    raw.reserve((stride + 1) * r.height);
    for (int y = 0; y < r.height; ++y) { raw.push_back('\0'); raw.append(reinterpret_cast<const char*>(r.data.data() + stride * y), stride); }
    uLongf cap = compressBound((uLong)raw.size());
    std::string z(cap, '\0');
    if (compress2(reinterpret_cast<Bytef*>(&z[0]), &cap, reinterpret_cast<const Bytef*>(raw.data()), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK)
        throw std::runtime_error("PNG compression failed");
    z.resize(cap);
    return std::string("\x89PNG\r\n\x1a\n", 8) + png_chunk("IHDR", ihdr) + png_chunk("IDAT", z) + png_chunk("IEND", "");
}

// binary PNM: P5 (gray), P6 (RGB); PAM: P7 with DEPTH 1..4, MAXVAL 255
inline Raster decode_pnm(const std::string& bytes, const std::string& name) {
    std::istringstream in(bytes);
    std::string magic;
    in >> magic;
    Raster r;
    auto next_token = [&]() {
        std::string t;
        while (in >> t) { if (t[0] == '#') { std::string rest; std::getline(in, rest); continue; } return t; }
        throw std::runtime_error(name + ": truncated PNM header");
    };
    if (magic == "P5" || magic == "P6") {
        r.channels = magic == "P5" ? 1 : 3;
        r.width = std::stoi(next_token()); r.height = std::stoi(next_token());
        if (std::stoi(next_token()) != 255) throw std::runtime_error(name + ": only 8-bit PNM files are supported");
        in.get();   // the single whitespace byte after MAXVAL
    } else if (magic == "P7") {
        std::string key;
        int maxval = 255;
        while (in >> key && key != "ENDHDR") {
            if (key == "WIDTH") in >> r.width; else if (key == "HEIGHT") in >> r.height; else if (key == "DEPTH") in >> r.channels;
            else if (key == "MAXVAL") in >> maxval; else { std::string rest; std::getline(in, rest); }
        }
        if (maxval != 255) throw std::runtime_error(name + ": only 8-bit PAM files are supported");
        in.get();
    } else throw std::runtime_error(name + ": not a binary PNM / PAM file");
    if (r.width <= 0 || r.height <= 0 || r.channels < 1 || r.channels > 4) throw std::runtime_error(name + ": bad PNM header");
    const size_t need = (size_t)r.width * r.height * r.channels;
    const std::streampos at = in.tellg();
    if (at < 0 || bytes.size() - (size_t)at < need) throw std::runtime_error(name + ": truncated PNM data");
    r.data.assign(bytes.begin() + (size_t)at, bytes.begin() + (size_t)at + need);
    return r;
}

inline bool ends_with(const std::string& s, const std::string& e) { return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0; }

inline Raster load_raster(const std::string& path) {
    const std::string bytes = slurp(path);
    if (bytes.size() >= 8 && (uint8_t)bytes[0] == 0x89 && bytes.compare(1, 3, "PNG") == 0) return decode_png(bytes, path);
    if (bytes.size() >= 2 && bytes[0] == 'P' && (bytes[1] == '5' || bytes[1] == '6' || bytes[1] == '7')) return decode_pnm(bytes, path);
    if (bytes.size() >= 2 && (uint8_t)bytes[0] == 0xff && (uint8_t)bytes[1] == 0xd8)
        throw std::runtime_error(path + ": JPEG files need libjpeg, which this build does not have — convert to PNG or PNM");
    throw std::runtime_error(path + ": unknown image file format");
}

inline void save_raster_png(const Raster& r, const std::string& path) {
    const std::string png = encode_png(r);
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f || !f.write(png.data(), (std::streamsize)png.size())) throw std::runtime_error("Unable to write " + path);
}

}  // namespace annonet_io
#endif
