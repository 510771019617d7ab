// jpeg_io.cpp — JPEG reader for Scene::LoadTexture (the reference's sponza textures are .JPG files, read there by the vendored
// stb_image through LoadImageF, template/template.cpp:1613-1627).  Written from ITU-T T.81: baseline / extended sequential (SOF0,
// SOF1) and progressive (SOF2) Huffman coding, 8-bit samples, 1 or 3 components (YCbCr per JFIF, or RGB when an Adobe APP14 marker
// says so), any sampling factors, restart intervals.  Arithmetic choices where T.81 leaves freedom: a separable floating-point
// inverse DCT, the triangle ("fancy") chroma interpolation that libjpeg documents for 2:1 factors (pixel replication otherwise) and
// JFIF's YCbCr->RGB equations with rounding.  Decoders differ by +-1..2 levels in exactly these places, stb_image included, and no
// reference test pins its output: parity of JPEG texels is UNPINNED; the unit test holds this reader to Pillow/libjpeg-turbo
// within a small tolerance instead.  Not read: arithmetic coding, lossless and hierarchical modes, 12-bit samples, CMYK.
#include <cmath>
#include <cstring>
#include <stdexcept>
#include "rt_host.h"

namespace rt355 {

namespace {

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huff {
    bool set = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    int16_t look[512];                    // 9-bit prefix -> (length << 8 | value), -1 when the code is longer
    bool build(const uint8_t* bits /*[16]*/, const uint8_t* v, int n)   // false: the code lengths over-subscribe the code space
    {
        memcpy(vals, v, (size_t)n);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l - 1]; k += bits[l - 1];
            if (code > (1 << l)) return false;
            maxcode[l] = bits[l - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int i = 0; i < 512; i++) look[i] = -1;
        code = 0; k = 0;
        for (int l = 1; l <= 9; l++) {
            for (int i = 0; i < bits[l - 1]; i++, k++, code++)
                for (int f = 0; f < (1 << (9 - l)); f++) look[(code << (9 - l)) | f] = (int16_t)((l << 8) | vals[k]);
            code <<= 1;
        }
        set = true;
        return true;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;                   // blocks per row / rows, padded to whole MCUs
    int cw = 0, ch = 0;                   // blocks that carry image data (non-interleaved scans cover exactly these)
    int pred = 0;
    std::vector<int16_t> coef;            // bw * bh * 64, natural (de-zigzagged) order
    std::vector<uint8_t> pix;             // (bw * 8) x (bh * 8) samples after the inverse DCT
};

struct Jpeg {
    const std::vector<uint8_t>& b;
    const std::string& file;
    size_t pos = 0;
    uint16_t qt[4][64] = {};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, mcusX = 0, mcusY = 0, restart = 0;
    bool progressive = false, adobeRGB = false;
    // entropy-coded segment reader
    uint32_t acc = 0; int nbits = 0; bool hitMarker = false;
    int eobrun = 0;

    Jpeg(const std::vector<uint8_t>& bytes, const std::string& f) : b(bytes), file(f) {}
    std::runtime_error bad(const char* why) const { return std::runtime_error("LoadTexture: " + file + ": " + why); }
    int u8() { if (pos >= b.size()) throw bad("truncated JPEG"); return b[pos++]; }
    int u16() { const int a = u8(); return (a << 8) | u8(); }

    void fill()
    {
        while (nbits <= 24) {
            int c = 0;
            if (!hitMarker && pos < b.size()) {
                c = b[pos];
                if (c == 0xff) {
                    const int d = pos + 1 < b.size() ? b[pos + 1] : 0xd9;
                    if (d == 0) pos += 2;                        // stuffed zero
                    else { hitMarker = true; c = 0; }            // a marker ends the segment: feed zeros
                } else pos++;
            }
            acc |= (uint32_t)c << (24 - nbits);
            nbits += 8;
        }
    }
    int bits(int n) { if (n == 0) return 0; if (nbits < n) fill(); const int v = (int)(acc >> (32 - n)); acc <<= n; nbits -= n; return v; }
    int bit() { return bits(1); }
    int decode(const Huff& h)
    {
        if (!h.set) throw bad("JPEG scan uses an undefined Huffman table");
        if (nbits < 16) fill();
        const int l9 = h.look[acc >> 23];
        if (l9 >= 0) { const int l = l9 >> 8; acc <<= l; nbits -= l; return l9 & 0xff; }
        int code = (int)(acc >> 23);
        for (int l = 10; l <= 16; l++) {
            code = (int)(acc >> (32 - l));
            if (code <= h.maxcode[l]) { acc <<= l; nbits -= l; return h.vals[h.valptr[l] + code - h.mincode[l]]; }
        }
        throw bad("bad Huffman code in JPEG data");
    }
    static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    void resetEntropy() { acc = 0; nbits = 0; hitMarker = false; eobrun = 0; for (Comp& c : comps) c.pred = 0; }

    // ---- marker segments ------------------------------------------------------------------
    void dqt(int len)
    {
        const size_t end = pos + (size_t)len;
        while (pos < end) {
            const int pq = u8(), id = pq & 15;
            if (id > 3) throw bad("bad JPEG quantisation table id");
            for (int i = 0; i < 64; i++) qt[id][kZigzag[i]] = (uint16_t)((pq >> 4) ? u16() : u8());
        }
    }
    void dht(int len)
    {
        const size_t end = pos + (size_t)len;
        while (pos < end) {
            const int tc = u8(), id = tc & 15;
            if (id > 3 || (tc >> 4) > 1) throw bad("bad JPEG Huffman table id");
            uint8_t bitsN[16], vals[256]; int n = 0;
            for (int i = 0; i < 16; i++) { bitsN[i] = (uint8_t)u8(); n += bitsN[i]; }
            if (n > 256) throw bad("bad JPEG Huffman table");
            for (int i = 0; i < n; i++) vals[i] = (uint8_t)u8();
            if (!((tc >> 4) ? ac[id] : dc[id]).build(bitsN, vals, n)) throw bad("bad JPEG Huffman table");
        }
    }
    void sof(int marker)
    {
        progressive = marker == 0xc2;
        if (u8() != 8) throw bad("only 8-bit JPEG samples are supported");
        H = u16(); W = u16();
        const int n = u8();
        if (W <= 0 || H <= 0 || (n != 1 && n != 3)) throw bad("unsupported JPEG frame (1 or 3 components)");
        if ((int64_t)W * H > kMaxTexturePixels) throw bad("JPEG frame larger than 64 Mpixel");
        comps.resize((size_t)n);
        for (Comp& c : comps) {
            c.id = u8(); const int hv = u8(); c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw bad("bad JPEG sampling factors");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        mcusX = (W + 8 * hmax - 1) / (8 * hmax); mcusY = (H + 8 * vmax - 1) / (8 * vmax);
        for (Comp& c : comps) {
            c.bw = mcusX * c.h; c.bh = mcusY * c.v;
            c.cw = ((W * c.h + hmax - 1) / hmax + 7) / 8; c.ch = ((H * c.v + vmax - 1) / vmax + 7) / 8;
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
    }

    // ---- one block of a scan ----------------------------------------------------------------
    void blockBaseline(Comp& c, int16_t* q)
    {
        const int t = decode(dc[c.td]);
        if (t > 11) throw bad("bad JPEG DC category");
        c.pred += t ? extend(bits(t), t) : 0;
        if (c.pred < -32768 || c.pred > 32767) throw bad("JPEG DC value out of range");
        q[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r != 15) break; k += 16; continue; }
            k += r;
            if (k > 63) throw bad("JPEG coefficient index out of range");
            q[kZigzag[k++]] = (int16_t)extend(bits(s), s);
        }
    }
    void blockProgressive(Comp& c, int16_t* q, int ss, int se, int ah, int al)
    {
        if (ss == 0) {                                            // DC scan
            if (ah == 0) {
                const int t = decode(dc[c.td]);
                if (t > 11) throw bad("bad JPEG DC category");
                c.pred += t ? extend(bits(t), t) : 0;
                if (c.pred < -32768 || c.pred > 32767) throw bad("JPEG DC value out of range");
                q[0] = (int16_t)(c.pred * (1 << al));
            } else if (bit()) q[0] = (int16_t)(q[0] | (1 << al));
            return;
        }
        if (ah == 0) {                                            // AC, first pass over this band
            if (eobrun > 0) { eobrun--; return; }
            for (int k = ss; k <= se;) {
                const int rs = decode(ac[c.ta]), r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1 + (r ? bits(r) : 0); break; }
                    k += 16; continue;
                }
                k += r;
                if (k > 63) throw bad("JPEG coefficient index out of range");
                q[kZigzag[k++]] = (int16_t)(extend(bits(s), s) * (1 << al));
            }
            return;
        }
        // AC refinement (T.81 G.1.2.3): new coefficients are +-1 << al, known ones may get one more bit
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; k++) {
                const int rs = decode(ac[c.ta]);
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s) { if (s != 1) throw bad("bad JPEG refinement code"); value = bit() ? p1 : m1; }
                else if (r != 15) { eobrun = (1 << r) + (r ? bits(r) : 0); break; }
                for (; k <= se; k++) {
                    int16_t& z = q[kZigzag[k]];
                    if (z != 0) { if (bit() && (z & p1) == 0) z = (int16_t)(z + (z >= 0 ? p1 : m1)); }
                    else { if (r == 0) break; r--; }
                }
                if (value && k <= se) q[kZigzag[k]] = (int16_t)value;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t& z = q[kZigzag[k]];
                if (z != 0 && bit() && (z & p1) == 0) z = (int16_t)(z + (z >= 0 ? p1 : m1));
            }
            eobrun--;
        }
    }

    void sos()
    {
        const int ns = u8();
        if (ns < 1 || ns > (int)comps.size()) throw bad("bad JPEG scan header");
        std::vector<Comp*> sc((size_t)ns);
        for (int i = 0; i < ns; i++) {
            const int id = u8(), t = u8();
            Comp* c = nullptr;
            for (Comp& k : comps) if (k.id == id) c = &k;
            if (!c) throw bad("JPEG scan names an unknown component");
            c->td = t >> 4; c->ta = t & 15;
            if (c->td > 3 || c->ta > 3) throw bad("bad JPEG table selector");
            sc[(size_t)i] = c;
        }
        const int ss = u8(), se = u8(), a = u8(), ah = a >> 4, al = a & 15;
        if (progressive ? (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) : (ss != 0 || se != 63 || a != 0)) throw bad("bad JPEG spectral selection");
        resetEntropy();
        auto doBlock = [&](Comp& c, int bx, int by) {
            int16_t* q = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (progressive) blockProgressive(c, q, ss, se, ah, al); else blockBaseline(c, q);
        };
        const int total = ns == 1 ? sc[0]->cw * sc[0]->ch : mcusX * mcusY;
        for (int m = 0; m < total; m++) {
            if (restart && m && m % restart == 0) {              // RSTn: byte-align, skip the marker, reset the predictors
                // (the bit reader never steps over a marker, so the RSTn is the next thing in the file; at most 7 pad bits are dropped)
                while (pos + 1 < b.size() && b[pos] == 0xff && b[pos + 1] == 0xff) pos++;
                if (pos + 1 >= b.size() || b[pos] != 0xff || (b[pos + 1] & 0xf8) != 0xd0) throw bad("missing JPEG restart marker");
                pos += 2;
                resetEntropy();
            }
            if (ns == 1) doBlock(*sc[0], m % sc[0]->cw, m / sc[0]->cw);
            else {
                const int mx = m % mcusX, my = m / mcusX;
                for (Comp* c : sc) for (int y = 0; y < c->v; y++) for (int x = 0; x < c->h; x++) doBlock(*c, mx * c->h + x, my * c->v + y);
            }
        }
        if (!hitMarker) {                                        // step over padding up to the next marker
            while (pos + 1 < b.size() && !(b[pos] == 0xff && b[pos + 1] != 0 && b[pos + 1] != 0xff)) pos++;
        }
    }

    // ---- reconstruction ---------------------------------------------------------------------
    void idctAll()
    {
        float cosT[8][8];
        for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) cosT[x][u] = (float)((u ? 1.0 : std::sqrt(0.5)) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
        for (Comp& c : comps) {
            c.pix.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            const uint16_t* q = qt[c.tq];
            const size_t stride = (size_t)c.bw * 8;
            for (int by = 0; by < c.bh; by++) for (int bx = 0; bx < c.bw; bx++) {
                const int16_t* z = &c.coef[((size_t)by * c.bw + bx) * 64];
                float f[64], t[64];
                for (int i = 0; i < 64; i++) f[i] = (float)(z[i] * (int)q[i]);
                for (int v = 0; v < 8; v++) for (int x = 0; x < 8; x++) {       // rows: over u
                    float s = 0; for (int u = 0; u < 8; u++) s += cosT[x][u] * f[v * 8 + u];
                    t[v * 8 + x] = s;
                }
                for (int x = 0; x < 8; x++) for (int y = 0; y < 8; y++) {       // columns: over v
                    float s = 0; for (int v = 0; v < 8; v++) s += cosT[y][v] * t[v * 8 + x];
                    const int p = (int)std::lrintf(s) + 128;
                    c.pix[((size_t)by * 8 + y) * stride + (size_t)bx * 8 + x] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));
                }
            }
        }
    }
    // component plane at full frame resolution
    std::vector<uint8_t> upsample(const Comp& c) const
    {
        const int fw = mcusX * 8 * hmax, fh = mcusY * 8 * vmax, hs = hmax / c.h, vs = vmax / c.v;
        const int cwp = c.bw * 8, chp = c.bh * 8;
        std::vector<uint8_t> out((size_t)fw * fh);
        if (hmax % c.h || vmax % c.v) throw bad("fractional JPEG sampling ratios are not supported");
        if (hs == 1 && vs == 1) return c.pix;
        // samples that carry image data (the padding of the last MCU must not bleed into the triangle filter)
        const int vw = std::min(cwp, (W * c.h + hmax - 1) / hmax), vh = std::min(chp, (H * c.v + vmax - 1) / vmax);
        auto at = [&](int x, int y) { x = x < 0 ? 0 : (x >= vw ? vw - 1 : x); y = y < 0 ? 0 : (y >= vh ? vh - 1 : y); return (int)c.pix[(size_t)y * cwp + x]; };
        if (hs == 2 && vs == 1) {                                 // h2v1: 3/4 nearer + 1/4 further sample
            for (int y = 0; y < fh; y++) for (int x = 0; x < fw; x++) {
                const int i = x >> 1, cy = y;
                out[(size_t)y * fw + x] = (uint8_t)((x & 1) ? (3 * at(i, cy) + at(i + 1, cy) + 2) >> 2 : (3 * at(i, cy) + at(i - 1, cy) + 1) >> 2);
            }
        } else if (hs == 2 && vs == 2) {                          // h2v2: the same weights in both directions (9:3:3:1)/16
            for (int y = 0; y < fh; y++) {
                const int j = y >> 1, jn = (y & 1) ? j + 1 : j - 1;
                for (int x = 0; x < fw; x++) {
                    const int i = x >> 1, in = (x & 1) ? i + 1 : i - 1;
                    const int cur = 3 * at(i, j) + at(i, jn), nb = 3 * at(in, j) + at(in, jn);
                    out[(size_t)y * fw + x] = (uint8_t)((3 * cur + nb + ((x & 1) ? 7 : 8)) >> 4);
                }
            }
        } else {
            for (int y = 0; y < fh; y++) for (int x = 0; x < fw; x++) out[(size_t)y * fw + x] = (uint8_t)at(x / hs, y / vs);
        }
        return out;
    }
};

inline uint8_t clamp8(float v) { const int i = (int)std::lrintf(v); return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i)); }

} // namespace

// 8-bit RGB, top row first
void DecodeJpeg(const std::vector<uint8_t>& bytes, const std::string& file, int& w, int& h, std::vector<uint8_t>& rgb)
{
    Jpeg j(bytes, file);
    if (bytes.size() < 4 || bytes[0] != 0xff || bytes[1] != 0xd8) throw j.bad("not a JPEG file");
    j.pos = 2;
    bool frame = false, done = false;
    while (!done) {
        int m = j.u8();
        if (m != 0xff) throw j.bad("JPEG marker expected");
        do m = j.u8(); while (m == 0xff);
        if (m == 0xd9) break;                                       // EOI
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;        // stand-alone markers
        const int len = j.u16() - 2;
        if (len < 0 || j.pos + (size_t)len > bytes.size()) throw j.bad("truncated JPEG segment");
        const size_t next = j.pos + (size_t)len;
        switch (m) {
        case 0xc0: case 0xc1: case 0xc2: if (frame) throw j.bad("second JPEG frame header"); j.sof(m); frame = true; break;
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
            throw j.bad("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
        case 0xc4: j.dht(len); break;
        case 0xdb: j.dqt(len); break;
        case 0xdd: j.restart = j.u16(); break;
        case 0xee: if (len >= 12 && !memcmp(&bytes[j.pos], "Adobe", 5)) j.adobeRGB = bytes[j.pos + 11] == 0; break;
        case 0xda:
            if (!frame) throw j.bad("JPEG scan before the frame header");
            j.sos();                                                // consumes its header and the entropy-coded data
            continue;
        default: break;                                             // APPn, COM, ...
        }
        j.pos = next;
        if (j.pos >= bytes.size()) done = true;
    }
    if (!frame) throw j.bad("JPEG without a frame");
    j.idctAll();
    w = j.W; h = j.H;
    rgb.resize((size_t)w * h * 3);
    const int fw = j.mcusX * 8 * j.hmax;
    if (j.comps.size() == 1) {
        const std::vector<uint8_t> y = j.upsample(j.comps[0]);
        for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) { const uint8_t v = y[(size_t)r * fw + c]; uint8_t* o = &rgb[((size_t)r * w + c) * 3]; o[0] = o[1] = o[2] = v; }
        return;
    }
    const std::vector<uint8_t> p0 = j.upsample(j.comps[0]), p1 = j.upsample(j.comps[1]), p2 = j.upsample(j.comps[2]);
    for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) {
        const size_t i = (size_t)r * fw + c;
        uint8_t* o = &rgb[((size_t)r * w + c) * 3];
        if (j.adobeRGB) { o[0] = p0[i]; o[1] = p1[i]; o[2] = p2[i]; continue; }
        const float Y = p0[i], cb = (float)p1[i] - 128.0f, cr = (float)p2[i] - 128.0f;
        o[0] = clamp8(Y + 1.402f * cr);
        o[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
        o[2] = clamp8(Y + 1.772f * cb);
    }
}

} // namespace rt355
