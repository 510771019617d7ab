// image_io.cpp — texture files for Scene::LoadTexture (reference: src/scene.cpp:244-256 -> LoadImageF,
// template/template.cpp:1613-1627 -> stbi_loadf of the vendored stb_image).  Readers written from the format
// specifications (PNG 1.2 / RFC 1950-1951 through zlib, Radiance RGBE, Truevision TGA 2.0; JPEG: jpeg_io.cpp); what is taken over from the
// reference's pipeline is the pixel rule it applies afterwards:
//   * 8-bit sources become float by stb's ldr->hdr rule, (float)(pow(v / 255.0f, 2.2f) * 1.0f) evaluated in double
//     (lib/stb_image.h:1553,1849); 16-bit PNG samples are first reduced to their high byte;
//   * .hdr texels are mantissa * 2^(e - 136), zero when e == 0;
//   * LoadImageF keeps channels 0..2; grey images are expanded to r = g = b here (the reference indexes past the pixel
//     for 1- and 2-channel files, template.cpp:1621-1623, which is undefined behaviour, not a convention to mirror);
//   * texels are appended to Scene::textures as float4 with w = 0 (float4(float3), template.cpp:810-814).
// JPEG files go through jpeg_io.cpp (reconstruction after stb_image's arithmetic).  Interlaced (Adam7) PNG files are read.
// Pinned: tests/test_ref_io_cpu.py compares LoadTexture with the reference's LoadImageF (its vendored stb_image compiled where it
// lies, oracle/ref_io_runner.cpp) bit for bit on the reference's own image files and on synthetic files of every variant.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <zlib.h>
#include "rt_host.h"

namespace rt355 {

namespace {

std::vector<uint8_t> slurp(const std::string& file)
{
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) throw std::runtime_error("LoadTexture: cannot open " + file);
    std::vector<uint8_t> b;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
    fclose(f);
    return b;
}

struct Image8 { int w = 0, h = 0; std::vector<uint8_t> rgb; };          // 8-bit RGB, top row first
struct ImageF { int w = 0, h = 0; std::vector<float> rgb; };

uint32_t be32at(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// ---- PNG ---------------------------------------------------------------------------------------
Image8 decodePng(const std::vector<uint8_t>& b, const std::string& file)
{
    auto bad = [&](const char* why) { return std::runtime_error("LoadTexture: " + file + ": " + why); };
    if (b.size() < 8 + 25) throw bad("truncated PNG");
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seenIhdr = false, seenEnd = false;
    while (pos + 12 <= b.size() && !seenEnd) {
        const uint32_t len = be32at(&b[pos]);
        const char* type = (const char*)&b[pos + 4];
        if (pos + 12 + (size_t)len > b.size()) throw bad("truncated PNG chunk");
        const uint8_t* d = &b[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) throw bad("bad IHDR");
            w = (int)be32at(d); h = (int)be32at(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            seenIhdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) seenEnd = true;
        pos += 12 + (size_t)len;
    }
    if (!seenIhdr || w <= 0 || h <= 0 || (int64_t)w * h > kMaxTexturePixels) throw bad("bad PNG header");
    if (interlace > 1) throw bad("bad PNG interlace method");
    int chan;
    switch (ctype) { case 0: chan = 1; break; case 2: chan = 3; break; case 3: chan = 1; break; case 4: chan = 2; break; case 6: chan = 4; break; default: throw bad("bad PNG colour type"); }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw bad("unsupported PNG bit depth");
    if (ctype == 3 && (depth == 16 || plte.size() < 3)) throw bad("bad PNG palette");
    const size_t bpp = std::max<size_t>(1, (size_t)chan * depth / 8);            // filter unit in bytes
    auto strideOf = [&](int pw) { return ((size_t)pw * chan * depth + 7) / 8; };
    // the image is one pass, or the seven passes of Adam7 (PNG 1.2 section 8.2): sub-images of every (dx, dy)-th pixel from (x0, y0)
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const Pass* passes = interlace ? adam7 : &whole;
    const int nPasses = interlace ? 7 : 1;
    size_t total = 0;
    for (int p = 0; p < nPasses; p++) {
        const int pw = (w - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (h - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (pw > 0 && ph > 0) total += (strideOf(pw) + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw(total);
    uLongf rawLen = (uLongf)raw.size();
    if (idat.empty() || uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) throw bad("PNG data does not inflate");
    Image8 im; im.w = w; im.h = h; im.rgb.resize((size_t)w * h * 3);
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;   // grey of < 8 bits is stretched to 0..255
    size_t at = 0;
    for (int p = 0; p < nPasses; p++) {
        const Pass& ps = passes[p];
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = strideOf(pw);
        // undo the scanline filters in place (PNG 1.2 section 6)
        std::vector<uint8_t> zero(stride, 0);
        for (int y = 0; y < ph; y++) {
            uint8_t* cur = &raw[at + (stride + 1) * (size_t)y];
            const uint8_t ft = cur[0];
            uint8_t* row = cur + 1;
            const uint8_t* up = y ? cur - stride : zero.data();     // previous row's bytes (already unfiltered), without its filter byte
            for (size_t x = 0; x < stride; x++) {
                const int a = x >= bpp ? row[x - bpp] : 0, bb = up[x], c = x >= bpp ? up[x - bpp] : 0;
                int add;
                switch (ft) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = bb; break;
                case 3: add = (a + bb) >> 1; break;
                case 4: { const int pp = a + bb - c, pa = abs(pp - a), pb = abs(pp - bb), pc = abs(pp - c); add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
                default: throw bad("bad PNG filter");
                }
                row[x] = (uint8_t)(row[x] + add);
            }
        }
        for (int y = 0; y < ph; y++) {
            const uint8_t* row = &raw[at + (stride + 1) * (size_t)y + 1];
            for (int x = 0; x < pw; x++) {
                uint8_t sv[4] = { 0, 0, 0, 0 };
                if (depth >= 8) { const int bytes = depth / 8; for (int k = 0; k < chan; k++) sv[k] = row[((size_t)x * chan + k) * bytes]; }   // 16 bit: high byte
                else { const int per = 8 / depth, sh = (per - 1 - x % per) * depth; sv[0] = (uint8_t)((row[x / per] >> sh) & ((1 << depth) - 1)); }
                uint8_t* o = &im.rgb[((size_t)(ps.y0 + y * ps.dy) * w + (size_t)(ps.x0 + x * ps.dx)) * 3];
                if (ctype == 3) { const size_t e = (size_t)sv[0] * 3; if (e + 3 > plte.size()) throw bad("PNG palette index out of range"); o[0] = plte[e]; o[1] = plte[e + 1]; o[2] = plte[e + 2]; }
                else if (chan <= 2) { const uint8_t g = (uint8_t)(depth < 8 ? sv[0] * scale : sv[0]); o[0] = o[1] = o[2] = g; }
                else { o[0] = sv[0]; o[1] = sv[1]; o[2] = sv[2]; }
            }
        }
        at += (stride + 1) * (size_t)ph;
    }
    return im;
}

// ---- TGA ---------------------------------------------------------------------------------------
Image8 decodeTga(const std::vector<uint8_t>& b, const std::string& file)
{
    auto bad = [&](const char* why) { return std::runtime_error("LoadTexture: " + file + ": " + why); };
    if (b.size() < 18) throw bad("truncated TGA");
    const int idLen = b[0], cmapType = b[1], type = b[2], w = b[12] | (b[13] << 8), h = b[14] | (b[15] << 8), bits = b[16], desc = b[17];
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (cmapType != 0 || !(type == 2 || type == 3 || type == 10 || type == 11)) throw bad("unsupported TGA type (true-colour and grey only)");
    if (w <= 0 || h <= 0 || !((grey && bits == 8) || (!grey && (bits == 24 || bits == 32)))) throw bad("unsupported TGA pixel size");
    if ((int64_t)w * h > kMaxTexturePixels) throw bad("TGA larger than 64 Mpixel");
    const int bytes = bits / 8;
    size_t pos = 18 + (size_t)idLen;
    Image8 im; im.w = w; im.h = h; im.rgb.resize((size_t)w * h * 3);
    const size_t total = (size_t)w * h;
    size_t i = 0;
    auto put = [&](const uint8_t* p) {
        const size_t y = i / w, x = i % w;
        const size_t yy = (desc & 0x20) ? y : (size_t)h - 1 - y, xx = (desc & 0x10) ? (size_t)w - 1 - x : x;   // default origin: bottom left
        uint8_t* o = &im.rgb[(yy * w + xx) * 3];
        if (grey) o[0] = o[1] = o[2] = p[0]; else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }                    // stored B, G, R
        i++;
    };
    while (i < total) {
        if (!rle) { if (pos + bytes > b.size()) throw bad("truncated TGA"); put(&b[pos]); pos += bytes; continue; }
        if (pos >= b.size()) throw bad("truncated TGA");
        const int hdr = b[pos++], n = (hdr & 0x7f) + 1;
        if (hdr & 0x80) { if (pos + bytes > b.size()) throw bad("truncated TGA"); for (int k = 0; k < n && i < total; k++) put(&b[pos]); pos += bytes; }
        else for (int k = 0; k < n && i < total; k++) { if (pos + bytes > b.size()) throw bad("truncated TGA"); put(&b[pos]); pos += bytes; }
    }
    return im;
}

// ---- Radiance .hdr (RGBE) ------------------------------------------------------------------------
ImageF decodeHdr(const std::vector<uint8_t>& b, const std::string& file)
{
    auto bad = [&](const char* why) { return std::runtime_error("LoadTexture: " + file + ": " + why); };
    size_t pos = 0;
    auto line = [&]() { std::string s; while (pos < b.size() && b[pos] != '\n') s.push_back((char)b[pos++]); pos++; return s; };
    std::string l = line();
    if (l != "#?RADIANCE" && l != "#?RGBE") throw bad("not a Radiance file");
    bool fmt = false;
    while (pos < b.size()) { l = line(); if (l.empty()) break; if (l == "FORMAT=32-bit_rle_rgbe") fmt = true; }
    if (!fmt) throw bad("unsupported Radiance format");
    l = line();
    int w = 0, h = 0;
    if (sscanf(l.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) throw bad("unsupported Radiance orientation");
    if ((int64_t)w * h > kMaxTexturePixels) throw bad("Radiance picture larger than 64 Mpixel");
    ImageF im; im.w = w; im.h = h; im.rgb.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto conv = [](const uint8_t* p, float* o) {
        if (p[3] == 0) { o[0] = o[1] = o[2] = 0.0f; return; }
        const float f = (float)ldexp(1.0f, (int)p[3] - (128 + 8));
        o[0] = p[0] * f; o[1] = p[1] * f; o[2] = p[2] * f;
    };
    for (int y = 0; y < h; y++) {
        if (pos + 4 > b.size()) throw bad("truncated Radiance data");
        const bool newRle = w >= 8 && w < 32768 && b[pos] == 2 && b[pos + 1] == 2 && !(b[pos + 2] & 0x80);
        if (newRle) {
            if (((b[pos + 2] << 8) | b[pos + 3]) != w) throw bad("bad Radiance scanline");
            pos += 4;
            for (int k = 0; k < 4; k++)                       // the four components are run-length coded one after the other
                for (int x = 0; x < w;) {
                    if (pos >= b.size()) throw bad("truncated Radiance data");
                    int n = b[pos++];
                    if (n > 128) { n -= 128; if (pos >= b.size() || x + n > w) throw bad("bad Radiance run"); const uint8_t v = b[pos++]; while (n--) scan[(size_t)(x++) * 4 + k] = v; }
                    else { if (n == 0 || pos + n > b.size() || x + n > w) throw bad("bad Radiance run"); while (n--) scan[(size_t)(x++) * 4 + k] = b[pos++]; }
                }
        } else {                                              // flat scanline
            if (pos + (size_t)w * 4 > b.size()) throw bad("truncated Radiance data");
            memcpy(scan.data(), &b[pos], (size_t)w * 4); pos += (size_t)w * 4;
        }
        for (int x = 0; x < w; x++) conv(&scan[(size_t)x * 4], &im.rgb[((size_t)y * w + x) * 3]);
    }
    return im;
}

} // namespace

// LoadImageF (template.cpp:1613-1627) for the formats above; returns w*h RGB float triples, top row first.
std::vector<float> LoadImageF(const std::string& file, int& w, int& h)
{
    const std::vector<uint8_t> b = slurp(file);
    static const uint8_t pngSig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (b.size() >= 8 && !memcmp(b.data(), pngSig, 8)) {
        Image8 im = decodePng(b, file);
        float lut[256];
        for (int v = 0; v < 256; v++) lut[v] = (float)(pow(v / 255.0f, 2.2f) * 1.0f);     // stb's ldr->hdr rule, double pow
        std::vector<float> out(im.rgb.size());
        for (size_t i = 0; i < out.size(); i++) out[i] = lut[im.rgb[i]];
        w = im.w; h = im.h;
        return out;
    }
    if (b.size() >= 2 && b[0] == 0xff && b[1] == 0xd8) {
        Image8 im;
        DecodeJpeg(b, file, im.w, im.h, im.rgb);
        float lut[256];
        for (int v = 0; v < 256; v++) lut[v] = (float)(pow(v / 255.0f, 2.2f) * 1.0f);
        std::vector<float> out(im.rgb.size());
        for (size_t i = 0; i < out.size(); i++) out[i] = lut[im.rgb[i]];
        w = im.w; h = im.h;
        return out;
    }
    if (b.size() >= 2 && b[0] == '#' && b[1] == '?') { ImageF im = decodeHdr(b, file); w = im.w; h = im.h; return im.rgb; }
    const size_t dot = file.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : file.substr(dot + 1);
    for (char& ch : ext) ch = (char)tolower(ch);
    if (ext == "tga") {
        Image8 im = decodeTga(b, file);
        float lut[256];
        for (int v = 0; v < 256; v++) lut[v] = (float)(pow(v / 255.0f, 2.2f) * 1.0f);
        std::vector<float> out(im.rgb.size());
        for (size_t i = 0; i < out.size(); i++) out[i] = lut[im.rgb[i]];
        w = im.w; h = im.h;
        return out;
    }
    throw std::runtime_error("LoadTexture: " + file + ": unsupported image format (PNG, JPEG, TGA and Radiance HDR are read)");
}

int Scene::LoadTexture(const std::string& filename, const std::string& name) // scene.cpp:244-256
{
    int w = 0, h = 0;
    const std::vector<float> rgb = LoadImageF(filename, w, h);
    std::vector<RtFloat4> texels((size_t)w * h);
    for (size_t i = 0; i < texels.size(); i++) texels[i] = RtFloat4{ rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], 0.0f };
    return AddTexture(texels.data(), w, h, name);
}

} // namespace rt355
