// jpeg_io.cpp — JPEG reader for Scene::LoadTexture (the reference's sponza textures are .JPG files, read there by the vendored
// stb_image through LoadImageF, template/template.cpp:1613-1627).  The entropy decoding is written from ITU-T T.81: baseline /
// extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman coding, 8-bit samples, 1 or 3 components, any sampling factors,
// restart intervals - coefficients are integers, every conforming decoder gets the same ones.  Where T.81 leaves freedom the
// reference's decoder decides what a texel is, so the RECONSTRUCTION follows the arithmetic of lib/stb_image.h v2.27 operation by
// operation (restated here, checked against that header compiled where it lies: tests/test_ref_io_cpu.py, 21 sponza textures and
// synthetic files bit for bit):
//   * dequantised coefficients wrap to 16 bits (stb_image.h:2231-2262, 3039-3043);
//   * the inverse DCT is the Loeffler-Ligtenberg-Moschytz integer transform of IJG's jidctint with 12-bit constants, two extra bits
//     after the column pass, rounding and the +128 level shift folded into the row pass (:2392-2489);
//   * chroma is interpolated row by row as the image is produced: nearer sample 3/4 + further 1/4 with "+2 >> 2" (h2v1, v2) or
//     (3a + b) per row then (3t0 + t1 + 8) >> 4 (h2v2), first and last sample of a row from one tap, the nearer / further ROW
//     chosen by a per-component step counter that starts half a step in; other factors repeat samples (:3400-3602, 3871-3888);
//   * YCbCr -> RGB in 20-bit fixed point with the green Cb term masked to its high 16 bits (:3603-3632); components named 'R','G','B',
//     or an Adobe APP14 transform of 0 without a JFIF marker, are taken as RGB (:3825).
// Not read: arithmetic coding, lossless and hierarchical modes, 12-bit samples, 4-component (CMYK / YCCK) files.
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
    bool progressive = false, jfif = false;
    int adobeTransform = -1, rgbIds = 0;   // APP14 colour transform (-1: none), components whose id is 'R','G','B' in that order
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
            if (n == 3 && c.id == "RGB"[&c - comps.data()]) rgbIds++;
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

    // ---- reconstruction (the arithmetic of lib/stb_image.h, see the file header) ---------------------------------------------
    static uint8_t clampLevel(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    // one 8-point pass of the LLM inverse DCT, constants = round(c * 4096): even part into x0..x3, odd part into t0..t3.  The sums are
    // kept modulo 2^32 (valid files stay far below; damaged ones wrap as the reference's int arithmetic does in practice, without
    // undefined behaviour here).
    typedef uint32_t U;
    static void idct1d(U s0, U s1, U s2, U s3, U s4, U s5, U s6, U s7, U x[4], U t[4])
    {
        U p1 = (s2 + s6) * 2217u;
        const U e2 = p1 + s6 * (U)-7567, e3 = p1 + s2 * 3135u;
        const U e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
        x[0] = e0 + e3; x[3] = e0 - e3; x[1] = e1 + e2; x[2] = e1 - e2;
        U t0 = s7, t1 = s5, t2 = s3, t3 = s1;
        U p3 = t0 + t2, p4 = t1 + t3, p2 = t1 + t2;
        p1 = t0 + t3;
        const U p5 = (p3 + p4) * 4816u;
        t0 *= 1223u; t1 *= 8410u; t2 *= 12586u; t3 *= 6149u;
        p1 = p5 + p1 * (U)-3685; p2 = p5 + p2 * (U)-10497; p3 *= (U)-8034; p4 *= (U)-1597;
        t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
    }
    static int sar(U v, int n) { return (int)((int32_t)v >> n); }   // arithmetic shift of the two's-complement value
    void idctAll()
    {
        for (Comp& c : comps) {
            c.pix.assign((size_t)c.bw * 8 * c.bh * 8, 0);
            const uint16_t* q = qt[c.tq];
            const size_t stride = (size_t)c.bw * 8;
            for (int by = 0; by < c.bh; by++) for (int bx = 0; bx < c.bw; bx++) {
                const int16_t* z = &c.coef[((size_t)by * c.bw + bx) * 64];
                int16_t d[64]; int val[64]; U x[4], t[4];
                for (int i = 0; i < 64; i++) d[i] = (int16_t)(uint16_t)((U)(int)z[i] * (U)q[i]);
                for (int i = 0; i < 8; i++) {                                   // columns
                    if (!(d[i + 8] | d[i + 16] | d[i + 24] | d[i + 32] | d[i + 40] | d[i + 48] | d[i + 56])) {
                        const int dc = d[i] * 4;
                        for (int r = 0; r < 8; r++) val[r * 8 + i] = dc;
                        continue;
                    }
                    idct1d((U)(int)d[i], (U)(int)d[i + 8], (U)(int)d[i + 16], (U)(int)d[i + 24], (U)(int)d[i + 32], (U)(int)d[i + 40], (U)(int)d[i + 48], (U)(int)d[i + 56], x, t);
                    for (int k = 0; k < 4; k++) { x[k] += 512u; val[k * 8 + i] = sar(x[k] + t[3 - k], 10); val[(7 - k) * 8 + i] = sar(x[k] - t[3 - k], 10); }
                }
                uint8_t* o = &c.pix[(size_t)by * 8 * stride + (size_t)bx * 8];
                for (int r = 0; r < 8; r++, o += stride) {                      // rows: 1 << 17 to remove, rounded, level shift folded in
                    const int* v = val + r * 8;
                    idct1d((U)v[0], (U)v[1], (U)v[2], (U)v[3], (U)v[4], (U)v[5], (U)v[6], (U)v[7], x, t);
                    for (int k = 0; k < 4; k++) { x[k] += 65536u + (128u << 17); o[k] = clampLevel(sar(x[k] + t[3 - k], 17)); o[7 - k] = clampLevel(sar(x[k] - t[3 - k], 17)); }
                }
            }
        }
    }
    // one output row of a component: `near` / `far` are the nearer and the further of the two sample rows around it
    void resampleRow(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) const
    {
        if (hs == 1 && vs == 1) { memcpy(out, near, (size_t)w); return; }
        if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); return; }
        if (hs == 2 && vs == 1) {
            if (w == 1) { out[0] = out[1] = near[0]; return; }
            out[0] = near[0];
            out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
            int i = 1;
            for (; i < w - 1; i++) { const int n = 3 * near[i] + 2; out[i * 2] = (uint8_t)((n + near[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + near[i + 1]) >> 2); }
            out[i * 2] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
            out[i * 2 + 1] = near[w - 1];
            return;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return; }
            int t1 = 3 * near[0] + far[0];
            out[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < w; i++) {
                const int t0 = t1;
                t1 = 3 * near[i] + far[i];
                out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
            return;
        }
        for (int i = 0; i < w; i++) for (int k = 0; k < hs; k++) out[i * hs + k] = near[i];   // any other factor: samples repeated (rows too)
    }
    // the whole image, row by row, as 8-bit RGB (a one-component file: r = g = b)
    void produce(std::vector<uint8_t>& rgb) const
    {
        const size_t n = comps.size();
        struct Step { int hs, vs, ystep, wLo, ypos, rows; const uint8_t* line0; const uint8_t* line1; size_t stride; std::vector<uint8_t> buf; };
        std::vector<Step> st(n);
        for (size_t k = 0; k < n; k++) {
            const Comp& c = comps[k];
            if (hmax % c.h || vmax % c.v) throw bad("fractional JPEG sampling ratios are not supported");
            Step& r = st[k];
            r.hs = hmax / c.h; r.vs = vmax / c.v; r.ystep = r.vs >> 1; r.wLo = (W + r.hs - 1) / r.hs; r.ypos = 0;
            r.rows = (H * c.v + vmax - 1) / vmax;                       // sample rows that carry image data
            r.stride = (size_t)c.bw * 8; r.line0 = r.line1 = c.pix.data();
            r.buf.assign((size_t)W + 8, 0);
        }
        const bool isRgb = n == 3 && (rgbIds == 3 || (adobeTransform == 0 && !jfif));
        rgb.resize((size_t)W * H * 3);
        for (int j = 0; j < H; j++) {
            for (size_t k = 0; k < n; k++) {
                Step& r = st[k];
                const bool bot = r.ystep >= (r.vs >> 1);
                resampleRow(r.buf.data(), bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.wLo, r.hs, r.vs);
                if (++r.ystep >= r.vs) { r.ystep = 0; r.line0 = r.line1; if (++r.ypos < r.rows) r.line1 += r.stride; }
            }
            uint8_t* o = &rgb[(size_t)j * W * 3];
            if (n == 1) { for (int i = 0; i < W; i++, o += 3) o[0] = o[1] = o[2] = st[0].buf[(size_t)i]; continue; }
            const uint8_t *y = st[0].buf.data(), *pcb = st[1].buf.data(), *pcr = st[2].buf.data();
            for (int i = 0; i < W; i++, o += 3) {
                if (isRgb) { o[0] = y[i]; o[1] = pcb[i]; o[2] = pcr[i]; continue; }
                const int yf = (y[i] << 20) + (1 << 19), cr = pcr[i] - 128, cb = pcb[i] - 128;
                const int r = (yf + cr * 1470208) >> 20;
                const int g = (int)(yf + cr * -748800 + (int)((unsigned)(cb * -360960) & 0xffff0000u)) >> 20;
                const int b = (yf + cb * 1858048) >> 20;
                o[0] = clampLevel(r); o[1] = clampLevel(g); o[2] = clampLevel(b);
            }
        }
    }
};

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
        case 0xe0: if (len >= 5 && !memcmp(&bytes[j.pos], "JFIF\0", 5)) j.jfif = true; break;
        case 0xee: if (len >= 12 && !memcmp(&bytes[j.pos], "Adobe\0", 6)) j.adobeTransform = bytes[j.pos + 11]; break;
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
    j.produce(rgb);
}

} // namespace rt355
