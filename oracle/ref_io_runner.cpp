// ref_io_runner.cpp — TEST INFRASTRUCTURE.  The reference reads its assets through two header-only libraries that it
// vendors as SOURCE: lib/stb_image.h (+ lib/stb_image_write.h) and src/tiny_obj_loader.h.  oracle/build_ref.sh compiles this
// file with g++ against those headers WHERE THEY LIE under /root/reference (nothing is copied into the repository, no stand-in
// headers) into oracle/_ref/libref_io.so, so that tests can run the reference's own decoders and OBJ parser and compare the
// host side of this repository (Scene::LoadTexture / LoadImageF, Scene::LoadModel, SavePNG) with them.
//
// Everything below is this repository's code: it CALLS the libraries the way the reference's call sites do and flattens
// what they return into plain arrays.
//   ref_load_image_f   follows LoadImageF      (template/template.cpp:1613-1627): stbi_loadf(file, &w, &h, &c, 0), then the
//                                               first three floats of every c-float pixel
//   ref_write_png      follows SaveImageF      (template/template.cpp:1629-1644): stbi_write_png(file, w, h, 3, img, 0)
//   ref_obj_*          follow Scene::LoadModel (src/scene.cpp:178-243): tinyobj::ObjReader with a default ObjReaderConfig,
//                                               then the loop over shapes / faces / face vertices that feeds AddTriangle
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

namespace {
std::string g_err;
struct Face { float v[9]; float uv[6]; int tex; };   // the face's `vertices` list AFTER std::reverse (+ _pos) and its `texcoords` list
std::vector<Face> g_faces;
std::vector<std::string> g_texNames;      // distinct `tex` strings handed to AddTriangle, in order of first use
std::vector<std::string> g_mtlNames, g_mtlDiffuse;   // reader.GetMaterials(): name, diffuse_texname (scene.cpp:192-195 loads these)
int texId(const std::string& s)
{
    for (size_t i = 0; i < g_texNames.size(); i++) if (g_texNames[i] == s) return (int)i;
    g_texNames.push_back(s);
    return (int)g_texNames.size() - 1;
}
}

extern "C" {

const char* ref_io_last_error() { return g_err.c_str(); }

// out == NULL: only w / h / c.  Otherwise out receives w*h*3 floats.  The reference reads data[i*c + 0..2] whatever c is; for
// c < 3 that runs past the pixel (and, on the last pixels, past the buffer): indices are clamped to the buffer here and the
// caller is told c, so a test can see that such files have no defined reference result.
int ref_load_image_f(const char* file, int* w, int* h, int* c, float* out)
{
    float* data = stbi_loadf(file, w, h, c, 0);
    if (!data) { g_err = std::string("stbi_loadf: ") + (stbi_failure_reason() ? stbi_failure_reason() : "?"); return -1; }
    if (out) {
        const long long s = (long long)*w * *h, total = s * *c;
        for (long long i = 0; i < s; i++)
            for (int k = 0; k < 3; k++) out[i * 3 + k] = data[std::min(i * *c + k, total - 1)];
    }
    stbi_image_free(data);
    return 0;
}

// 8-bit decode (stbi_load, the path stbi_loadf converts from for LDR files): out receives w*h*c bytes when not NULL
int ref_load_image_u8(const char* file, int* w, int* h, int* c, unsigned char* out)
{
    unsigned char* data = stbi_load(file, w, h, c, 0);
    if (!data) { g_err = std::string("stbi_load: ") + (stbi_failure_reason() ? stbi_failure_reason() : "?"); return -1; }
    if (out) memcpy(out, data, (size_t)*w * *h * *c);
    stbi_image_free(data);
    return 0;
}

int ref_write_png(const char* file, int w, int h, const unsigned char* rgb) { return stbi_write_png(file, w, h, 3, rgb, 0) ? 0 : -1; }

// Scene::LoadModel up to (not including) the AddTriangle calls.  Returns the number of triangles, -1 on a reader error.
int ref_obj_load(const char* filename, const char* defaultMat, float px, float py, float pz, int forceDefaultMat)
{
    g_faces.clear(); g_texNames.clear(); g_mtlNames.clear(); g_mtlDiffuse.clear();
    tinyobj::ObjReaderConfig readerConfig;
    tinyobj::ObjReader reader;
    if (!reader.ParseFromFile(filename, readerConfig)) { g_err = "E/TinyObjReader: " + reader.Error(); return -1; }
    auto& attrib = reader.GetAttrib();
    auto& shapes = reader.GetShapes();
    auto& materials = reader.GetMaterials();
    for (const auto& m : materials) { g_mtlNames.push_back(m.name); g_mtlDiffuse.push_back(m.diffuse_texname); }
    for (size_t s = 0; s < shapes.size(); s++) {
        size_t index_offset = 0;
        for (size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); f++) {
            const size_t fv = size_t(shapes[s].mesh.num_face_vertices[f]);
            std::vector<float> verts, uvs;
            for (size_t v = 0; v < fv; v++) {
                tinyobj::index_t idx = shapes[s].mesh.indices[index_offset + v];
                verts.push_back(attrib.vertices[3 * size_t(idx.vertex_index) + 0] + px);
                verts.push_back(attrib.vertices[3 * size_t(idx.vertex_index) + 1] + py);
                verts.push_back(attrib.vertices[3 * size_t(idx.vertex_index) + 2] + pz);
                tinyobj::real_t tx = 0, ty = 0;
                if (idx.texcoord_index >= 0) {
                    tx = attrib.texcoords[2 * size_t(idx.texcoord_index) + 0];
                    ty = 1.0 - attrib.texcoords[2 * size_t(idx.texcoord_index) + 1];   // double arithmetic, as in scene.cpp:218
                }
                uvs.push_back(tx); uvs.push_back(ty);
            }
            int matIdx = shapes[s].mesh.material_ids[f];
            std::string tex = defaultMat;
            if (matIdx >= 0) tex = materials[matIdx].diffuse_texname;
            if (tex.empty() || forceDefaultMat) tex = defaultMat;
            if (fv == 3) {   // a default ObjReaderConfig triangulates, so every face has three vertices; anything else is reported
                Face fc;
                for (int k = 0; k < 3; k++)   // std::reverse(vertices) (scene.cpp:228): element k is the face's vertex 2 - k
                    for (int a = 0; a < 3; a++) fc.v[k * 3 + a] = verts[(2 - k) * 3 + a];
                for (int k = 0; k < 6; k++) fc.uv[k] = uvs[k];
                fc.tex = texId(tex);
                g_faces.push_back(fc);
            } else { g_err = "face with " + std::to_string(fv) + " vertices after triangulation"; return -1; }
            index_offset += fv;
        }
    }
    return (int)g_faces.size();
}
void ref_obj_faces(float* verts /* n*9 */, float* uvs /* n*6 */, int* tex /* n */)
{
    for (size_t i = 0; i < g_faces.size(); i++) {
        memcpy(verts + i * 9, g_faces[i].v, sizeof g_faces[i].v);
        memcpy(uvs + i * 6, g_faces[i].uv, sizeof g_faces[i].uv);
        tex[i] = g_faces[i].tex;
    }
}
int ref_obj_tex_count() { return (int)g_texNames.size(); }
const char* ref_obj_tex_name(int i) { return g_texNames[i].c_str(); }
int ref_obj_material_count() { return (int)g_mtlNames.size(); }
const char* ref_obj_material_name(int i) { return g_mtlNames[i].c_str(); }
const char* ref_obj_material_diffuse(int i) { return g_mtlDiffuse[i].c_str(); }

}   // extern "C"
