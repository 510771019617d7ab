// scene_io.cpp — on-disk formats either side of the path:
//   * Scene::LoadModel: Wavefront OBJ (+MTL map_Kd names) with the conventions of the reference's loader
//     (src/scene.cpp:178-243): faces are fan-triangulated, each face's VERTEX list is reversed while its
//     texcoord list is not, v -> 1-v, material = the diffuse texture's name; the MTL's map_Kd images are loaded
//     into the atlas first (LoadTexture, scene.cpp:192-195), a missing image is an error.
//   * SavePNG: the float image of rt_postproc / Renderer::SaveFrame as an 8-bit RGB PNG, bytes computed as
//     SaveImageF does (template/template.cpp:1629-1644): clamp to 1, (uchar)(c*255).
// Parity note: the reference parses with tinyobjloader 2.0.0 and stb; no reference test pins their output, so
// asset IO parity is UNPINNED (SURVEY.md §8(c) "Third-party arithmetic").
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include "rt_host.h"

namespace rt355 {

static std::string dirOf(const std::string& path)
{
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

static void parseMtl(const std::string& file, std::map<std::string, std::string>& kdMap)
{
    std::ifstream in(file);
    std::string line, cur;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag; ls >> tag;
        if (tag == "newmtl") { ls >> cur; kdMap[cur] = ""; }
        else if (tag == "map_Kd" && !cur.empty()) { std::string tex, last; while (ls >> tex) last = tex; kdMap[cur] = last; }
    }
}

int Scene::LoadModel(const std::string& filename, const std::string& defaultMat, float3 pos, bool forceDefaultMat)
{
    std::ifstream in(filename);
    if (!in) throw std::runtime_error("LoadModel: cannot open " + filename);
    std::vector<float3> V; std::vector<float2> VT;
    std::map<std::string, std::string> kdOf;           // material name -> diffuse texture name
    std::string line, curMtl;
    int added = 0;
    auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (int)n + i; };   // OBJ indices: 1-based or negative
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string tag; ls >> tag;
        if (tag == "v") { float3 p; ls >> p.x >> p.y >> p.z; V.push_back(p); }
        else if (tag == "vt") { float2 t; ls >> t.x >> t.y; VT.push_back(t); }
        else if (tag == "mtllib") {
            std::string m; ls >> m; parseMtl(dirOf(filename) + m, kdOf);
            // scene.cpp:192-195: every MTL material with a diffuse texture gets its image loaded (material name = texture name)
            // before any face is added.  A texture registered already (AddTexture / an earlier LoadTexture) is kept.
            if (!forceDefaultMat)
                for (const auto& kv : kdOf)
                    if (!kv.second.empty() && !HasMaterial(kv.second)) {
                        try { LoadTexture(dirOf(filename) + kv.second, kv.second); }
                        catch (const std::exception& e) { throw std::runtime_error("LoadModel: " + filename + ": material '" + kv.first + "': " + e.what()); }
                    }
        }
        else if (tag == "usemtl") { ls >> curMtl; }
        else if (tag == "f") {
            std::vector<float3> fv; std::vector<float2> ft;
            std::string tok;
            while (ls >> tok) {
                int vi = 0, ti = 0; bool hasT = false;
                size_t s1 = tok.find('/');
                vi = atoi(tok.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) { ti = atoi(t.c_str()); hasT = true; }
                }
                int v = fix(vi, V.size());
                if (v < 0 || v >= (int)V.size()) throw std::runtime_error("LoadModel: vertex index out of range in " + filename);
                fv.push_back(V[v] + pos);
                float2 uv;
                if (hasT) { int t = fix(ti, VT.size()); if (t >= 0 && t < (int)VT.size()) { uv.x = VT[t].x; uv.y = 1.0f - VT[t].y; } }
                ft.push_back(uv);
            }
            if (fv.size() < 3) continue;
            std::string tex = defaultMat;
            auto it = kdOf.find(curMtl);
            if (it != kdOf.end()) tex = it->second;
            if (tex.empty() || forceDefaultMat || !HasMaterial(tex)) tex = defaultMat;
            for (size_t k = 1; k + 1 < fv.size(); k++) {   // fan (0, k, k+1), then the reference's per-face vertex reversal
                float3 tri[3] = { fv[0], fv[k], fv[k + 1] };
                float2 uv[3] = { ft[0], ft[k], ft[k + 1] };
                AddTriangle(tri[2], tri[1], tri[0], uv[0], uv[1], uv[2], tex);   // vertices reversed, texcoords not (scene.cpp:228,235-237)
                added++;
            }
        }
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
