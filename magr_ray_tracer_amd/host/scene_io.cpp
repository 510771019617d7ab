// scene_io.cpp — on-disk formats either side of the path:
//   * Scene::LoadModel: Wavefront OBJ (+ MTL map_Kd names) read the way the reference reads it (src/scene.cpp:178-243 on top of its
//     vendored tinyobjloader 2.0.0, src/tiny_obj_loader.h): tinyobjloader's own number reader (NOT strtod: the decimal digits are summed in
//     double with a table of powers of ten, tiny_obj_loader.h:887-1017), its index rules (1-based, negative = relative, 0 refused), its
//     triangulation (a quad is cut along its SHORTER diagonal, larger polygons by its ear clipping, :1484-1926), its grouping (faces
//     leave in file order), its MTL reader (material = rest of the `newmtl` line, texture = rest of the `map_Kd` line after the
//     options, :1243-1322, :2060-2300); then the reference's loop: every MTL material with a diffuse texture has its image loaded into
//     the atlas, in MTL order, before any face is added (scene.cpp:190-195); per triangle the VERTEX list is reversed while the
//     texcoord list is not, v -> 1 - v in double, material = the diffuse texture's name (:204-238).
//     Pinned by tests/test_ref_io_cpu.py against the reference's own header compiled where it lies (oracle/ref_io_runner.cpp).
//   * SavePNG: the float image of rt_postproc / Renderer::SaveFrame as an 8-bit RGB PNG, bytes computed as
//     SaveImageF does (template/template.cpp:1629-1644): clamp to 1, (uchar)(c*255).
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>
#include "rt_host.h"

namespace rt355 {

static std::string dirOf(const std::string& path)
{
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

namespace {

inline bool isBlank(char c) { return c == ' ' || c == '\t'; }
inline bool isDigit(char c) { return (unsigned)(c - '0') < 10u; }
inline bool isEol(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// One line, ended by "\n", "\r\n" or a lone "\r" (tiny_obj_loader.h:762-794); false at the end of the stream.
bool nextLine(std::istream& in, std::string& line)
{
    line.clear();
    if (in.peek() == EOF) return false;
    std::streambuf* sb = in.rdbuf();
    for (;;) {
        const int c = sb->sbumpc();
        if (c == '\n') break;
        if (c == '\r') { if (sb->sgetc() == '\n') sb->sbumpc(); break; }
        if (c == EOF) { if (line.empty()) in.setstate(std::ios::eofbit); break; }
        line += (char)c;
    }
    return true;
}

// tinyobjloader's number reader (tiny_obj_loader.h:887-1017): sign, integer digits accumulated in a double, fraction digits added as
// digit * 10^-k (k < 8 from a table of double literals, std::pow beyond), optional exponent applied as ldexp(m * 5^e, e).  It is not
// correctly rounded - which is the point: coordinates come out as the reference's do.
bool objNumber(const char* s, const char* end, double& out)
{
    if (s >= end) return false;
    static const double tenth[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
    double mant = 0.0;
    int expo = 0, read = 0;
    char sign = '+', esign = '+';
    bool dotFirst = false;
    const char* c = s;
    if (*c == '+' || *c == '-') { sign = *c++; if (c != end && *c == '.') dotFirst = true; }
    else if (isDigit(*c)) {}
    else if (*c == '.') dotFirst = true;
    else return false;
    bool more = c != end;
    if (!dotFirst) {
        while (more && isDigit(*c)) { mant *= 10; mant += (int)(*c - '0'); c++; read++; more = c != end; }
        if (read == 0) return false;
    }
    if (more) {
        bool tail = true;
        if (*c == '.') {
            c++; read = 1; more = c != end;
            while (more && isDigit(*c)) { mant += (int)(*c - '0') * (read < 8 ? tenth[read] : std::pow(10.0, -read)); read++; c++; more = c != end; }
        } else if (*c != 'e' && *c != 'E') tail = false;
        if (tail && more && (*c == 'e' || *c == 'E')) {
            c++; more = c != end;
            if (more && (*c == '+' || *c == '-')) { esign = *c; c++; }
            else if (isDigit(*c)) {}      // (the original looks at *c here even at the token's end: that character is the blank or NUL after it)
            else return false;
            read = 0; more = c != end;
            while (more && isDigit(*c)) { if (expo > INT_MAX / 10) return false; expo *= 10; expo += (int)(*c - '0'); c++; read++; more = c != end; }
            expo *= esign == '+' ? 1 : -1;
            if (read == 0) return false;
        }
    }
    out = (sign == '+' ? 1 : -1) * (expo ? std::ldexp(mant * std::pow(5.0, expo), expo) : mant);
    return true;
}
// parseReal (tiny_obj_loader.h:1019-1027): the next blank-separated token as a float, `dflt` when it is not a number; the token is consumed either way
float objReal(const char*& tok, double dflt = 0.0)
{
    tok += strspn(tok, " \t");
    const char* end = tok + strcspn(tok, " \t\r");
    double v = dflt;
    objNumber(tok, end, v);
    tok = end;
    return (float)v;
}
std::string objWord(const char*& tok)
{
    tok += strspn(tok, " \t");
    const size_t e = strcspn(tok, " \t\r");
    std::string s(tok, tok + e);
    tok += e;
    return s;
}

struct Corner { int v = -1, vt = -1; };
// fixIndex (tiny_obj_loader.h:815-842): 1-based -> 0-based, negative = relative to the elements read so far, 0 only where allowed
bool fixIndex(int idx, int n, int& out, bool allowZero)
{
    if (idx > 0) { out = idx - 1; return true; }
    if (idx == 0) { out = -1; return allowZero; }
    out = n + idx;
    return true;
}
// parseTriple (tiny_obj_loader.h:1157-1208): i, i/j, i//k, i/j/k with atoi's reading of each field
bool objCorner(const char*& tok, int nV, int nVT, Corner& c)
{
    int dummy = -1;
    if (!fixIndex(atoi(tok), nV, c.v, false)) return false;
    tok += strcspn(tok, "/ \t\r");
    if (tok[0] != '/') return true;
    tok++;
    if (tok[0] == '/') {                                  // i//k
        tok++;
        if (!fixIndex(atoi(tok), 0, dummy, true)) return false;
        tok += strcspn(tok, "/ \t\r");
        return true;
    }
    if (!fixIndex(atoi(tok), nVT, c.vt, true)) return false;   // i/j or i/j/k
    tok += strcspn(tok, "/ \t\r");
    if (tok[0] != '/') return true;
    tok++;
    if (!fixIndex(atoi(tok), 0, dummy, true)) return false;
    tok += strcspn(tok, "/ \t\r");
    return true;
}

struct MtlEntry { std::string name, diffuse; };
// The texture name of a map_* statement (tiny_obj_loader.h:1243-1322): options and their arguments are skipped token by token,
// what follows them up to the end of the (right-trimmed) line is the name - blanks included.
bool mtlTextureName(const char* tok, std::string& name)
{
    bool found = false;
    auto opt = [&](const char* o) { const size_t n = strlen(o); return strncmp(tok, o, n) == 0 && isBlank(tok[n]); };
    auto skipWords = [&](size_t optLen, int words) { tok += optLen; for (int k = 0; k < words; k++) { tok += strspn(tok, " \t"); tok += strcspn(tok, " \t\r"); } };
    while (!isEol(*tok)) {
        tok += strspn(tok, " \t");
        if (opt("-blendu") || opt("-blendv")) skipWords(8, 1);
        else if (opt("-clamp") || opt("-boost")) skipWords(7, 1);
        else if (opt("-bm")) skipWords(4, 1);
        else if (opt("-o") || opt("-s") || opt("-t")) skipWords(3, 3);
        else if (opt("-type")) skipWords(5, 1);
        else if (opt("-texres")) skipWords(7, 1);
        else if (opt("-imfchan")) skipWords(9, 1);
        else if (opt("-mm")) skipWords(4, 2);
        else if (opt("-colorspace")) skipWords(12, 1);
        else { name = tok; tok += name.size(); found = true; }
    }
    return found;
}
// LoadMtl (tiny_obj_loader.h:2028-2420) as far as LoadModel uses it: the materials in file order with their diffuse texture names
void parseMtl(std::istream& in, std::vector<MtlEntry>& mats, std::map<std::string, int>& ids)
{
    MtlEntry cur;
    std::string line;
    auto flush = [&]() { ids.insert({ cur.name, (int)mats.size() }); mats.push_back(cur); };
    while (nextLine(in, line)) {
        if (!line.empty()) line = line.substr(0, line.find_last_not_of(" \t") + 1);
        if (line.empty()) continue;
        const char* tok = line.c_str();
        tok += strspn(tok, " \t");
        if (tok[0] == '\0' || tok[0] == '#') continue;
        if (strncmp(tok, "newmtl", 6) == 0 && isBlank(tok[6])) {
            if (!cur.name.empty()) flush();
            cur = MtlEntry();
            cur.name = tok + 7;
        } else if (strncmp(tok, "map_Kd", 6) == 0 && isBlank(tok[6])) {
            std::string t;
            if (mtlTextureName(tok + 7, t)) cur.diffuse = t;
        }
    }
    flush();   // the last material goes in whatever its name (tiny_obj_loader.h:2413-2416)
}

struct ObjFace { std::vector<Corner> c; };
struct ObjTri { Corner c[3]; int material; };

// exportGroupsToShape (tiny_obj_loader.h:1456-1926), faces only: triangles pass, a quad is cut along its shorter diagonal (ties: 1-3),
// a larger polygon is ear-clipped in the plane of its first real corner.  All arithmetic in float, in the original's order.
void triangulate(const std::vector<ObjFace>& faces, int material, const std::vector<float>& v, std::vector<ObjTri>& out)
{
    auto emit = [&](const Corner& a, const Corner& b, const Corner& c) { out.push_back(ObjTri{ { a, b, c }, material }); };
    auto oob = [&](size_t vi) { return 3 * vi + 2 >= v.size(); };
    for (const ObjFace& face : faces) {
        size_t n = face.c.size();
        if (n < 3) continue;                                  // "Degenerated face"
        if (n == 3) { emit(face.c[0], face.c[1], face.c[2]); continue; }   // (indices are checked when the triangles are added)
        if (n == 4) {
            const size_t i0 = (size_t)face.c[0].v, i1 = (size_t)face.c[1].v, i2 = (size_t)face.c[2].v, i3 = (size_t)face.c[3].v;
            if (oob(i0) || oob(i1) || oob(i2) || oob(i3)) continue;   // "Face with invalid vertex index found": skipped
            const float e02x = v[i2 * 3] - v[i0 * 3], e02y = v[i2 * 3 + 1] - v[i0 * 3 + 1], e02z = v[i2 * 3 + 2] - v[i0 * 3 + 2];
            const float e13x = v[i3 * 3] - v[i1 * 3], e13y = v[i3 * 3 + 1] - v[i1 * 3 + 1], e13z = v[i3 * 3 + 2] - v[i1 * 3 + 2];
            const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z, sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
            if (sqr02 < sqr13) { emit(face.c[0], face.c[1], face.c[2]); emit(face.c[0], face.c[2], face.c[3]); }
            else { emit(face.c[0], face.c[1], face.c[3]); emit(face.c[1], face.c[2], face.c[3]); }
            continue;
        }
        // the two axes to work in: drop the axis along which the first corner that is not degenerate has its largest normal component
        size_t ax[2] = { 1, 2 };
        for (size_t k = 0; k < n; k++) {
            const size_t a = (size_t)face.c[k % n].v, b = (size_t)face.c[(k + 1) % n].v, c = (size_t)face.c[(k + 2) % n].v;
            if (oob(a) || oob(b) || oob(c)) continue;
            const float e0x = v[b * 3] - v[a * 3], e0y = v[b * 3 + 1] - v[a * 3 + 1], e0z = v[b * 3 + 2] - v[a * 3 + 2];
            const float e1x = v[c * 3] - v[b * 3], e1y = v[c * 3 + 1] - v[b * 3 + 1], e1z = v[c * 3 + 2] - v[b * 3 + 2];
            const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = 1.1920928955078125e-7f;
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) { ax[0] = 0; if (cz > cx && cz > cy) ax[1] = 1; }
                break;
            }
        }
        std::vector<Corner> rest = face.c;
        size_t guess = 0, budget = n, before = n;
        while (rest.size() > 3 && budget > 0) {
            n = rest.size();
            if (guess >= n) guess -= n;
            if (before != n) { before = n; budget = n; } else budget--;
            Corner ind[3];
            float px[3], py[3];
            for (size_t k = 0; k < 3; k++) {
                ind[k] = rest[(guess + k) % n];
                const size_t vi = (size_t)ind[k].v;
                if (vi * 3 + ax[0] >= v.size() || vi * 3 + ax[1] >= v.size()) px[k] = py[k] = 0.0f;
                else { px[k] = v[vi * 3 + ax[0]]; py[k] = v[vi * 3 + ax[1]]; }
            }
            const float e0x = px[1] - px[0], e0y = py[1] - py[0], e1x = px[2] - px[1], e1y = py[2] - py[1];
            const float crs = e0x * e1y - e0y * e1x;
            const float area = (px[0] * py[1] - py[0] * px[1]) * 0.5f;     // the original's "area": of the first two corners only
            if (crs * area < 0.0f) { guess += 1; continue; }              // "an internal angle"
            bool overlap = false;
            for (size_t o = 3; o < n; o++) {                               // any other corner inside this ear?
                const size_t ovi = (size_t)rest[(guess + o) % n].v;
                if (ovi * 3 + ax[0] >= v.size() || ovi * 3 + ax[1] >= v.size()) continue;
                const float tx = v[ovi * 3 + ax[0]], ty = v[ovi * 3 + ax[1]];
                int in = 0;                                                // pnpoly (tiny_obj_loader.h:1409-1419) over the three corners
                for (int i = 0, j = 2; i < 3; j = i++)
                    if (((py[i] > ty) != (py[j] > ty)) && (tx < (px[j] - px[i]) * (ty - py[i]) / (py[j] - py[i]) + px[i])) in = !in;
                if (in) { overlap = true; break; }
            }
            if (overlap) { guess += 1; continue; }
            emit(ind[0], ind[1], ind[2]);
            rest.erase(rest.begin() + (long)((guess + 1) % n));            // the ear's middle corner leaves the polygon
        }
        if (rest.size() == 3) emit(rest[0], rest[1], rest[2]);             // (a polygon that ran out of budget loses what is left, as in the original)
    }
}

}   // namespace

int Scene::LoadModel(const std::string& filename, const std::string& defaultMat, float3 pos, bool forceDefaultMat)
{
    std::ifstream in(filename);
    if (!in) throw std::runtime_error("LoadModel: cannot open " + filename);
    std::vector<float> v, vt;
    std::vector<MtlEntry> mtl; std::map<std::string, int> mtlId; std::set<std::string> mtlFiles;
    std::vector<ObjFace> group; std::vector<ObjTri> tris;
    int material = -1;
    std::string line;
    size_t lineNo = 0;
    while (nextLine(in, line)) {                                           // LoadObj (tiny_obj_loader.h:2552-3402)
        lineNo++;
        if (line.empty()) continue;
        const char* tok = line.c_str();
        tok += strspn(tok, " \t");
        if (tok[0] == '\0' || tok[0] == '#') continue;
        if (tok[0] == 'v' && isBlank(tok[1])) { tok += 2; const float x = objReal(tok), y = objReal(tok), z = objReal(tok); v.push_back(x); v.push_back(y); v.push_back(z); }
        else if (tok[0] == 'v' && tok[1] == 't' && isBlank(tok[2])) { tok += 3; const float x = objReal(tok), y = objReal(tok); vt.push_back(x); vt.push_back(y); }
        else if (tok[0] == 'f' && isBlank(tok[1])) {
            tok += 2; tok += strspn(tok, " \t");
            ObjFace face;
            while (!isEol(tok[0])) {
                Corner c;
                if (!objCorner(tok, (int)(v.size() / 3), (int)(vt.size() / 2), c))
                    throw std::runtime_error("LoadModel: " + filename + ": failed to parse `f' line " + std::to_string(lineNo) + " (e.g. a zero vertex index)");
                face.c.push_back(c);
                tok += strspn(tok, " \t\r");
            }
            group.push_back(face);
        }
        else if (strncmp(tok, "usemtl", 6) == 0) {
            tok += 6;
            const std::string name = objWord(tok);
            const auto it = mtlId.find(name);
            const int id = it == mtlId.end() ? -1 : it->second;
            if (id != material) { triangulate(group, material, v, tris); group.clear(); material = id; }
        }
        else if (strncmp(tok, "mtllib", 6) == 0 && isBlank(tok[6])) {
            tok += 7;
            std::vector<std::string> names;                                // split at blanks, a backslash escapes the next character
            { std::string cur; bool esc = false; for (const char* c = tok; *c; c++) { if (esc) { cur += *c; esc = false; } else if (*c == '\\') esc = true; else if (*c == ' ') { if (!cur.empty()) names.push_back(cur); cur.clear(); } else cur += *c; } if (!cur.empty()) names.push_back(cur); }
            for (const std::string& nm : names) {                          // the first file that opens is read, once
                if (mtlFiles.count(nm)) continue;
                std::ifstream mf(dirOf(filename) + nm);
                if (!mf) continue;
                parseMtl(mf, mtl, mtlId);
                mtlFiles.insert(nm);
                break;
            }
        }
        else if ((tok[0] == 'g' || tok[0] == 'o') && isBlank(tok[1])) { triangulate(group, material, v, tris); group.clear(); }
    }
    triangulate(group, material, v, tris);

    // scene.cpp:190-195: the image of every MTL material with a diffuse texture goes into the atlas, in MTL order, before any face is
    // added (the material is named after the texture).  One difference kept for callers that register textures themselves
    // (AddTexture): an image file that does not exist is not an error when a material of that name is already there.
    for (const MtlEntry& m : mtl) {
        if (m.diffuse.empty()) continue;
        const std::string path = dirOf(filename) + m.diffuse;
        if (!std::ifstream(path) && HasMaterial(m.diffuse)) continue;
        try { LoadTexture(path, m.diffuse); }
        catch (const std::exception& e) { throw std::runtime_error("LoadModel: " + filename + ": material '" + m.name + "': " + e.what()); }
    }
    // scene.cpp:196-240
    int added = 0;
    for (const ObjTri& t : tris) {
        float3 p[3]; float2 uv[3];
        for (int k = 0; k < 3; k++) {
            const size_t vi = (size_t)t.c[k].v;
            if (t.c[k].v < 0 || vi * 3 + 2 >= v.size()) throw std::runtime_error("LoadModel: " + filename + ": vertex index out of range (the reference reads past its vertex array here)");
            p[k] = float3(v[vi * 3], v[vi * 3 + 1], v[vi * 3 + 2]) + pos;
            if (t.c[k].vt >= 0 && (size_t)t.c[k].vt * 2 + 1 < vt.size()) {
                uv[k].x = vt[(size_t)t.c[k].vt * 2];
                uv[k].y = (float)(1.0 - vt[(size_t)t.c[k].vt * 2 + 1]);   // `ty = 1.0 - ...` is double arithmetic (scene.cpp:218)
            }
        }
        std::string tex = defaultMat;
        if (t.material >= 0) tex = mtl[(size_t)t.material].diffuse;
        if (tex.empty() || forceDefaultMat || !HasMaterial(tex)) tex = defaultMat;
        // the vertex list is reversed, the texcoord list is not (scene.cpp:228,235-237).  The reference hands both lists to AddTriangle
        // through `vertices[v++], vertices[v++], vertices[v++]`, whose evaluation order C++ leaves to the compiler; taken left to right
        // (as here) the triangle is (V2, V1, V0) with (T0, T1, T2), taken right to left (V0, V1, V2) with (T2, T1, T0): the same
        // vertex-texcoord pairs either way, the stored winding differs - and extend() turns the normal towards the ray (wavefront.cl:71-72).
        AddTriangle(p[2], p[1], p[0], uv[0], uv[1], uv[2], tex);
        added++;
    }
    return added;
}

// ---- minimal PNG (stored deflate blocks) ---------------------------------------------------------------------------
static uint32_t crcTable[256]; static bool crcInit = false;
static uint32_t crc32(uint32_t c, const uint8_t* p, size_t n)
{
    if (!crcInit) { for (uint32_t i = 0; i < 256; i++) { uint32_t k = i; for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1; crcTable[i] = k; } crcInit = true; }
    c = ~c;
    for (size_t i = 0; i < n; i++) c = crcTable[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}
static void be32(std::vector<uint8_t>& o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
static void chunk(std::vector<uint8_t>& png, const char* type, const std::vector<uint8_t>& data)
{
    be32(png, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
    png.insert(png.end(), td.begin(), td.end());
    be32(png, crc32(0, td.data(), td.size()));
}
void SavePNG(const std::string& file, int w, int h, const RtFloat4* data)
{
    std::vector<uint8_t> raw; raw.reserve((size_t)h * (w * 3 + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);
        for (int x = 0; x < w; x++) {
            const RtFloat4& p = data[(size_t)y * w + x];
            float c[3] = { p.x > 1 ? 1.0f : p.x, p.y > 1 ? 1.0f : p.y, p.z > 1 ? 1.0f : p.z };
            for (int k = 0; k < 3; k++) raw.push_back((uint8_t)(c[k] * 255));
        }
    }
    std::vector<uint8_t> z = { 0x78, 0x01 };
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521; b = (b + a) % 65521; }
    for (size_t off = 0; off < raw.size(); off += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> png = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a }, ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h); ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr); chunk(png, "IDAT", z); chunk(png, "IEND", {});
    FILE* f = fopen(file.c_str(), "wb");
    if (!f) throw std::runtime_error("SavePNG: cannot write " + file);
    fwrite(png.data(), 1, png.size(), f);
    fclose(f);
}

} // namespace rt355
