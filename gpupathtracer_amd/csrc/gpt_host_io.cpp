// gpt_host_io.cpp -- the callers and data formats either side of the hot path
// (plain host C++): OBJ loading with the reference loader's semantics, the
// reference's scene/material/camera tables, and headless image output.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/gpt.h"

extern "C" const char* gpt_last_error(void);
// gpt_api.cpp owns the error string; this file reports through a setter there.
extern "C" void gpt_set_last_error_(const char* msg);

namespace {

gpt_status io_fail(gpt_status st, const std::string& msg) {
    gpt_set_last_error_(msg.c_str());
    return st;
}

// SET_DEFAULT_MATERIAL_PROPERTIES (material.h:19-26) with SET_TO_AIR_MEDIUM (medium.h:17-24)
Material default_material() {
    Material m;
    memset(&m, 0, sizeof(m));
    m.medium.refractiveIndex = (float)1.000293;
    return m;
}

void set3(Color& c, double x, double y, double z) { c.x = (float)x; c.y = (float)y; c.z = (float)z; }

struct Preset { const char* name; Material m; };

// path_tracer.cpp:124-215
std::vector<Preset> make_presets() {
    std::vector<Preset> v;
    Material m;
    m = default_material(); set3(m.diffuseColor, 0.87, 0.15, 0.15); set3(m.specularColor, 1, 1, 1);
    m.medium.refractiveIndex = (float)1.491; v.push_back({"red", m});
    m = default_material(); set3(m.diffuseColor, 0.15, 0.87, 0.15); set3(m.specularColor, 1, 1, 1);
    m.medium.refractiveIndex = (float)1.491; v.push_back({"green", m});
    m = default_material(); set3(m.diffuseColor, 0.93, 0.33, 0.04); set3(m.specularColor, 1, 1, 1);
    m.medium.refractiveIndex = (float)1.491; v.push_back({"orange", m});
    // purple: its diffuse colour is overwritten with black at path_tracer.cpp:150
    m = default_material(); set3(m.diffuseColor, 0, 0, 0); set3(m.specularColor, 1, 1, 1);
    m.medium.refractiveIndex = (float)1.491; v.push_back({"purple", m});
    m = default_material(); set3(m.specularColor, 1, 1, 1); m.medium.refractiveIndex = (float)2.42;
    m.hasTransmission = true; v.push_back({"glass", m});
    m = default_material(); set3(m.specularColor, 1, 1, 1); m.hasTransmission = true;
    m.medium.refractiveIndex = (float)1.62;
    set3(m.medium.absorptionAndScatteringProperties.absorptionCoefficient, 1.0, 0.01, 1.0);
    v.push_back({"greenGlass", m});
    m = default_material(); set3(m.specularColor, 1, 1, 1); m.hasTransmission = true;
    m.medium.refractiveIndex = (float)1.486;
    set3(m.medium.absorptionAndScatteringProperties.absorptionCoefficient, 0.6, 0.6, 0.6);
    m.medium.absorptionAndScatteringProperties.reducedScatteringCoefficient = (float)8.0;
    v.push_back({"marble", m});
    m = default_material(); set3(m.specularColor, 1, 1, 1); m.hasTransmission = true;
    m.medium.refractiveIndex = (float)1.333;
    set3(m.medium.absorptionAndScatteringProperties.absorptionCoefficient, 0.9, 0.3, 0.02);
    m.medium.absorptionAndScatteringProperties.reducedScatteringCoefficient = (float)2.0;
    v.push_back({"something", m});
    m = default_material(); set3(m.specularColor, 1, 1, 1); m.hasTransmission = true;
    m.medium.refractiveIndex = (float)1.350;
    set3(m.medium.absorptionAndScatteringProperties.absorptionCoefficient, 0.02, 5.1, 5.7);
    m.medium.absorptionAndScatteringProperties.reducedScatteringCoefficient = (float)9.0;
    v.push_back({"ketchup", m});
    m = default_material(); set3(m.diffuseColor, 0.9, 0.9, 0.9); v.push_back({"white", m});
    m = default_material(); set3(m.diffuseColor, 0.4, 0.6, 0.8); set3(m.specularColor, 1, 1, 1);
    m.medium.refractiveIndex = (float)1.2; v.push_back({"lightBlue", m});
    m = default_material(); set3(m.specularColor, 0.869, 0.621, 0.027);
    m.medium.refractiveIndex = (float)1000.0; v.push_back({"gold", m});
    m = default_material(); set3(m.specularColor, 0.89, 0.89, 0.89);
    m.medium.refractiveIndex = (float)1000.0; v.push_back({"steel", m});
    m = default_material(); set3(m.emittedColor, 13, 13, 11); v.push_back({"light", m});
    return v;
}

const std::vector<Preset>& presets() {
    static const std::vector<Preset> p = make_presets();
    return p;
}

struct SphereRow { double x, y, z, r; const char* mat; };
// path_tracer.cpp:217-271
const SphereRow kSpheres[14] = {
    {-0.9, 0, -0.9, 0.8, "steel"},      {0.9, -0.5, 1.3, 0.3, "greenGlass"}, {-0.5, -0.4, 1.0, 0.4, "lightBlue"},
    {-1.0, -0.7, 1.2, 0.1, "lightBlue"}, {-0.5, -0.7, 1.7, 0.1, "lightBlue"}, {0.3, -0.7, 1.4, 0.1, "lightBlue"},
    {-0.1, -0.7, 0.1, 0.1, "lightBlue"}, {0.2, -0.55, 0.7, 0.25, "lightBlue"}, {0.8, 0, -0.4, 0.8, "green"},
    {0.8, 1.2, -0.4, 0.4, "green"},      {0.8, 1.8, -0.4, 0.2, "green"},      {0.8, 2.1, -0.4, 0.1, "green"},
    {0.8, 2.25, -0.4, 0.05, "green"},    {0.8, 2.325, -0.4, 0.025, "green"}};

// BasicMath (basic_math.cpp:5-36): float constants, double arithmetic inside.
const float kPiF = (float)3.1415926535897932384626422832795028841971;
float radians_to_degrees(float radians) { float degrees = radians * 180.0 / kPiF; return degrees; }
float degrees_to_radians(float degrees) { float radians = degrees / 180.0 * kPiF; return radians; }
float bm_min(float a, float b) { if (a < b) return a; else return b; }
float bm_max(float a, float b) { if (a > b) return a; else return b; }
float bm_clamp(float n, float low, float high) { n = bm_min(n, high); n = bm_max(n, low); return n; }

std::vector<std::string> split_spaces(const std::string& s, size_t from) {
    // std::getline(stream, tok, ' ') semantics: empty tokens between adjacent
    // spaces are kept, a trailing delimiter yields no extra token.
    std::vector<std::string> out;
    std::istringstream ss(s.substr(from));
    std::string tok;
    while (std::getline(ss, tok, ' ')) out.push_back(tok);
    return out;
}

}  // namespace

extern "C" {

void gpt_free(void* p) { free(p); }

int gpt_default_scene_sphere_count(void) { return 14; }

gpt_status gpt_material_preset(const char* name, Material* out) {
    if (!name || !out) return io_fail(GPT_ERR_INVALID_ARGUMENT, "null argument");
    for (const Preset& p : presets())
        if (strcmp(p.name, name) == 0) { *out = p.m; return GPT_OK; }
    return io_fail(GPT_ERR_INVALID_ARGUMENT, std::string("unknown material preset: ") + name);
}

void gpt_default_scene(Sphere* out) {
    for (int i = 0; i < 14; i++) {
        memset(&out[i], 0, sizeof(Sphere));
        out[i].position.x = (float)kSpheres[i].x;
        out[i].position.y = (float)kSpheres[i].y;
        out[i].position.z = (float)kSpheres[i].z;
        out[i].radius = (float)kSpheres[i].r;
        gpt_material_preset(kSpheres[i].mat, &out[i].material);
    }
}

// ---- named scenes ------------------------------------------------------------------------------
// The reference was shipped as five executables -- CornellBox, Diamond-NoDOF, Monkey, Speed, Subsurface --
// each built by editing the scene table, the mesh path and the camera in the source
// (path_tracer.cpp:54, 116-278; interactive_camera.cpp:15-25).  Only their NAMES survive (the binaries
// are not part of the source tree); the committed source is the 14-sphere scene ("Default").  The
// presets below restate what each name says with the reference's own meshes, materials and sphere
// table; mesh placements and material choices are this library's.
namespace {
struct MeshPart { const char* file; const char* material; float scale, tx, ty, tz; };
struct ScenePreset {
    const char* name;
    bool spheres;               // the 14 spheres of the default scene
    MeshPart parts[2];
    int n_parts;
    bool cornell;               // parts[0] is box.obj used as a room (walls coloured per face pair, front and floor left out)
    gpt_preset_view view;
};
const ScenePreset kScenes[] = {
    {"Default", true, {}, 0, false, {0.0f, 0.3f, 4.0f, 0.04f, 45.0f, {0, 0, 0}}},
    // box.obj (a cube of +-0.8) scaled x3 and lifted so that its floor lies in the ground plane: a room around the spheres
    {"CornellBox", true, {{"box.obj", "white", 3.0f, 0.0f, 1.6f, 0.0f}}, 1, true, {0.0f, 0.3f, 4.0f, 0.04f, 45.0f, {0, 0, 0}}},
    {"Diamond-NoDOF", true, {{"diamond.obj", "glass", 0.8f, 0.2f, -2.2f, 2.6f}}, 1, false, {0.0f, 0.3f, 4.0f, 0.0f, 45.0f, {0, 0, 0}}},
    {"Monkey", true, {{"monkey.obj", "glass", 0.5f, 0.0f, -0.2f, 1.5f}}, 1, false, {0.0f, 0.3f, 4.0f, 0.04f, 45.0f, {0, 0, 0}}},
    // the sphere scene without the lens: the cheapest frame the renderer draws
    {"Speed", true, {}, 0, false, {0.0f, 0.3f, 4.0f, 0.0f, 45.0f, {0, 0, 0}}},
    {"Subsurface", true, {{"coil.obj", "marble", 0.25f, -1.2f, -2.2f, 1.2f}, {"diamond.obj", "ketchup", 0.8f, 0.2f, -2.2f, 2.6f}}, 2, false,
     {0.0f, 0.3f, 4.0f, 0.04f, 45.0f, {0, 0, 0}}},
};
const int kNumScenes = (int)(sizeof(kScenes) / sizeof(kScenes[0]));
}  // namespace

int gpt_scene_preset_count(void) { return kNumScenes; }
const char* gpt_scene_preset_name(int index) { return (index >= 0 && index < kNumScenes) ? kScenes[index].name : nullptr; }

gpt_status gpt_scene_preset(const char* name, const char* mesh_dir, Sphere** out_spheres, int* out_numSpheres,
                            Poly** out_polys, int* out_numPolys, gpt_preset_view* out_view) {
    if (!name || !out_spheres || !out_numSpheres || !out_polys || !out_numPolys)
        return io_fail(GPT_ERR_INVALID_ARGUMENT, "null argument");
    *out_spheres = nullptr; *out_numSpheres = 0; *out_polys = nullptr; *out_numPolys = 0;
    const ScenePreset* sp = nullptr;
    for (const ScenePreset& k : kScenes) if (strcmp(k.name, name) == 0) sp = &k;
    if (!sp) return io_fail(GPT_ERR_INVALID_ARGUMENT, std::string("unknown scene preset: ") + name);
    if (sp->n_parts > 0 && !mesh_dir) return io_fail(GPT_ERR_INVALID_ARGUMENT, std::string(name) + " needs the directory of the reference's .obj files");
    std::vector<Poly> all;
    for (int k = 0; k < sp->n_parts; k++) {
        const MeshPart& part = sp->parts[k];
        Material m;
        gpt_status st = gpt_material_preset(part.material, &m);
        if (st != GPT_OK) return st;
        Poly* polys = nullptr;
        int n = 0;
        st = gpt_obj_load((std::string(mesh_dir) + "/" + part.file).c_str(), &m, &polys, &n, nullptr);
        if (st != GPT_OK) return st;
        for (int i = 0; i < n; i++) {
            GPT_FLOAT3* v[3] = {&polys[i].p0, &polys[i].p1, &polys[i].p2};
            for (int c = 0; c < 3; c++) {
                v[c]->x = v[c]->x * part.scale + part.tx; v[c]->y = v[c]->y * part.scale + part.ty; v[c]->z = v[c]->z * part.scale + part.tz;
            }
        }
        if (sp->cornell && k == 0) {
            // box.obj's faces come in pairs: front, top, back, bottom, right, left.  The front wall is left out (the
            // orbit camera stands outside and looks in) and so is the floor (the ground plane is the floor); the
            // ceiling is the only emitter.
            if (n != 12) { gpt_free(polys); return io_fail(GPT_ERR_PARSE, "CornellBox expects the reference's box.obj (12 faces)"); }
            const struct { int face; const char* mat; } walls[8] = {{2, "light"}, {3, "light"}, {4, "white"}, {5, "white"},
                                                                    {8, "green"}, {9, "green"}, {10, "red"}, {11, "red"}};
            for (const auto& w : walls) {
                Poly q = polys[w.face];
                gpt_material_preset(w.mat, &q.material);
                all.push_back(q);
            }
        } else {
            all.insert(all.end(), polys, polys + n);
        }
        gpt_free(polys);
    }
    if (sp->spheres) {
        *out_spheres = (Sphere*)calloc(14, sizeof(Sphere));
        if (!*out_spheres) return io_fail(GPT_ERR_OUT_OF_MEMORY, "out of memory");
        gpt_default_scene(*out_spheres);
        *out_numSpheres = 14;
    }
    if (!all.empty()) {
        *out_polys = (Poly*)malloc(all.size() * sizeof(Poly));
        if (!*out_polys) { gpt_free(*out_spheres); *out_spheres = nullptr; *out_numSpheres = 0; return io_fail(GPT_ERR_OUT_OF_MEMORY, "out of memory"); }
        memcpy(*out_polys, all.data(), all.size() * sizeof(Poly));
        *out_numPolys = (int)all.size();
    }
    if (out_view) *out_view = sp->view;
    return GPT_OK;
}

// InteractiveCamera::setResolution/setFOVX/buildRenderCamera
// (interactive_camera.cpp:61-87)
void gpt_orbit_camera(float width, float height, float fovx_degrees, float yaw, float pitch, float radius,
                      float apertureRadius, const float center[3], Camera* out) {
    memset(out, 0, sizeof(*out));
    float fovx = fovx_degrees;
    float fovy = radians_to_degrees(atan(tan(degrees_to_radians(fovx) * 0.5) * (height / width)) * 2.0);
    float xDirection = sinf(yaw) * cosf(pitch);
    float yDirection = sinf(pitch);
    float zDirection = cosf(yaw) * cosf(pitch);
    out->position.x = center[0] + xDirection * radius;
    out->position.y = center[1] + yDirection * radius;
    out->position.z = center[2] + zDirection * radius;
    out->view.x = -xDirection; out->view.y = -yDirection; out->view.z = -zDirection;
    out->up.x = 0; out->up.y = 1; out->up.z = 0;
    out->resolution.x = width; out->resolution.y = height;
    out->fov.x = fovx; out->fov.y = fovy;
    out->apertureRadius = apertureRadius;
    out->focalDistance = radius;
}

// InteractiveCamera() defaults (interactive_camera.cpp:15-25) + display.cpp:78-85
void gpt_default_camera(float width, float height, Camera* out) {
    const float center[3] = {0, 0, 0};
    gpt_orbit_camera(width, height, 45, 0, (float)0.3, 4, (float)0.04, center, out);
}

// floatTo8Bit(c / passCounter): color.cpp:41-62, display.cpp:198
void gpt_tonemap_8bit(const float* sum_rgb, int numPixels, int passes, unsigned char* out_rgb8) {
    float gamma = 2.2;
    float oneOverGamma = 1.0 / gamma;
    for (size_t i = 0; i < (size_t)numPixels * 3; i++) {
        float c = sum_rgb[i] / (float)passes;
        float g = std::pow(c, oneOverGamma);
        out_rgb8[i] = (unsigned char)bm_clamp(g * 255, 0, 255);
    }
}

gpt_status gpt_write_ppm(const char* path, const float* sum_rgb, int width, int height, int passes) {
    if (!path || !sum_rgb || width <= 0 || height <= 0 || passes <= 0)
        return io_fail(GPT_ERR_INVALID_ARGUMENT, "bad argument to gpt_write_ppm");
    std::vector<unsigned char> px((size_t)width * height * 3);
    gpt_tonemap_8bit(sum_rgb, width * height, passes, px.data());
    FILE* f = fopen(path, "wb");
    if (!f) return io_fail(GPT_ERR_IO, std::string("cannot open ") + path);
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    size_t w = fwrite(px.data(), 1, px.size(), f);
    fclose(f);
    if (w != px.size()) return io_fail(GPT_ERR_IO, std::string("short write to ") + path);
    return GPT_OK;
}

gpt_status gpt_write_pfm(const char* path, const float* sum_rgb, int width, int height, int passes) {
    if (!path || !sum_rgb || width <= 0 || height <= 0 || passes <= 0)
        return io_fail(GPT_ERR_INVALID_ARGUMENT, "bad argument to gpt_write_pfm");
    FILE* f = fopen(path, "wb");
    if (!f) return io_fail(GPT_ERR_IO, std::string("cannot open ") + path);
    fprintf(f, "PF\n%d %d\n-1.0\n", width, height);       // negative scale: little endian
    std::vector<float> row((size_t)width * 3);
    for (int y = height - 1; y >= 0; y--) {               // PFM rows run bottom to top
        for (int i = 0; i < width * 3; i++) row[(size_t)i] = sum_rgb[(size_t)y * width * 3 + i] / (float)passes;
        if (fwrite(row.data(), 4, row.size(), f) != row.size()) { fclose(f); return io_fail(GPT_ERR_IO, "short write"); }
    }
    fclose(f);
    return GPT_OK;
}

// OpenEXR 2 scanline file, uncompressed, three 32-bit float channels (stored in the
// alphabetical order B, G, R the format prescribes), row 0 at the top (lineOrder INCREASING_Y).
namespace {
void exr_put(std::vector<unsigned char>& b, const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    b.insert(b.end(), c, c + n);
}
void exr_i32(std::vector<unsigned char>& b, int32_t v) { exr_put(b, &v, 4); }      // (little-endian host)
void exr_f32(std::vector<unsigned char>& b, float v) { exr_put(b, &v, 4); }
void exr_str(std::vector<unsigned char>& b, const char* s) { exr_put(b, s, strlen(s) + 1); }
void exr_attr(std::vector<unsigned char>& b, const char* name, const char* type, const std::vector<unsigned char>& value) {
    exr_str(b, name);
    exr_str(b, type);
    exr_i32(b, (int32_t)value.size());
    b.insert(b.end(), value.begin(), value.end());
}
}  // namespace

gpt_status gpt_write_exr(const char* path, const float* sum_rgb, int width, int height, int passes) {
    if (!path || !sum_rgb || width <= 0 || height <= 0 || passes <= 0)
        return io_fail(GPT_ERR_INVALID_ARGUMENT, "bad argument to gpt_write_exr");
    std::vector<unsigned char> hd;
    exr_i32(hd, 20000630);     // magic
    exr_i32(hd, 2);            // version 2, single-part scanline, no long names
    {
        std::vector<unsigned char> v;
        for (const char* ch : {"B", "G", "R"}) {
            exr_str(v, ch);
            exr_i32(v, 2);                         // pixel type FLOAT
            v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);   // pLinear + reserved
            exr_i32(v, 1);                         // x sampling
            exr_i32(v, 1);                         // y sampling
        }
        v.push_back(0);
        exr_attr(hd, "channels", "chlist", v);
    }
    { std::vector<unsigned char> v(1, 0); exr_attr(hd, "compression", "compression", v); }            // NO_COMPRESSION
    for (const char* name : {"dataWindow", "displayWindow"}) {
        std::vector<unsigned char> v;
        exr_i32(v, 0); exr_i32(v, 0); exr_i32(v, width - 1); exr_i32(v, height - 1);
        exr_attr(hd, name, "box2i", v);
    }
    { std::vector<unsigned char> v(1, 0); exr_attr(hd, "lineOrder", "lineOrder", v); }                // INCREASING_Y
    { std::vector<unsigned char> v; exr_f32(v, 1.0f); exr_attr(hd, "pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; exr_f32(v, 0.0f); exr_f32(v, 0.0f); exr_attr(hd, "screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; exr_f32(v, 1.0f); exr_attr(hd, "screenWindowWidth", "float", v); }
    hd.push_back(0);           // end of header
    FILE* f = fopen(path, "wb");
    if (!f) return io_fail(GPT_ERR_IO, std::string("cannot open ") + path);
    const size_t row_bytes = 8 + (size_t)width * 12;          // y, byte count, three channel rows
    std::vector<uint64_t> offsets((size_t)height);
    for (int y = 0; y < height; y++) offsets[(size_t)y] = hd.size() + (size_t)height * 8 + (size_t)y * row_bytes;
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size() &&
              fwrite(offsets.data(), 8, offsets.size(), f) == offsets.size();
    std::vector<unsigned char> row;
    for (int y = 0; y < height && ok; y++) {
        row.clear();
        exr_i32(row, y);
        exr_i32(row, width * 12);
        for (int c = 2; c >= 0; c--)                                      // B, G, R
            for (int x = 0; x < width; x++) exr_f32(row, sum_rgb[((size_t)y * width + x) * 3 + c] / (float)passes);
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    fclose(f);
    return ok ? GPT_OK : io_fail(GPT_ERR_IO, "short write");
}

// objLoader::objLoader (objcore/objloader.cpp:18-122) + PathTracer::prepMesh
// (path_tracer.cpp:52-69) + the poly loop of createDeviceData (:275-278).
gpt_status gpt_obj_load(const char* path, const Material* material, Poly** out_polys, int* out_numPolys,
                        float bbox[6]) {
    if (!path || !out_polys || !out_numPolys) return io_fail(GPT_ERR_INVALID_ARGUMENT, "null argument");
    *out_polys = nullptr;
    *out_numPolys = 0;
    std::ifstream in(path);
    if (!in.is_open()) return io_fail(GPT_ERR_IO, std::string("ERROR: ") + path + " could not be found");

    std::vector<float> points, normals;
    std::vector<std::vector<int>> faces, facenormals;
    while (in.good()) {
        std::string line;
        std::getline(in, line);
        if (line.empty()) line = "42";
        const char c0 = line[0], c1 = line.size() > 1 ? line[1] : '\0';
        if (c0 == 'v' && c1 == 't') {
            continue;   // texture coordinates never reach the renderer
        } else if (c0 == 'v') {
            std::vector<std::string> t = split_spaces(line, 0);
            float xyz[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++)
                if ((size_t)(k + 1) < t.size()) xyz[k] = (float)::atof(t[(size_t)k + 1].c_str());
            std::vector<float>& dst = (c1 == 'n') ? normals : points;
            dst.insert(dst.end(), xyz, xyz + 3);
        } else if (c0 == 'f') {
            std::vector<std::string> t = split_spaces(line, 0);
            std::vector<int> pl, nl;
            if (line.find("//") != std::string::npos) {
                for (size_t k = 1; k < t.size(); k++) {
                    std::istringstream fs(t[k]);
                    std::string f;
                    std::getline(fs, f, '/');
                    pl.push_back((int)(::atof(f.c_str()) - 1));
                    std::getline(fs, f, '/');
                    std::getline(fs, f, ' ');
                    nl.push_back((int)(::atof(f.c_str()) - 1));
                }
                faces.push_back(pl);
                facenormals.push_back(nl);
            } else if (line.find("/") != std::string::npos) {
                for (size_t k = 1; k < t.size(); k++) {
                    std::istringstream fs(t[k]);
                    std::string f;
                    int i = 0;
                    while (std::getline(fs, f, '/')) {
                        if (i == 0) pl.push_back((int)(::atof(f.c_str()) - 1));
                        else if (i == 2) nl.push_back((int)(::atof(f.c_str()) - 1));
                        i++;
                    }
                }
                faces.push_back(pl);
                facenormals.push_back(nl);
            } else {
                for (size_t k = 1; k < t.size(); k++) pl.push_back((int)(::atof(t[k].c_str()) - 1));
                faces.push_back(pl);
            }
        }
    }
    const int npts = (int)(points.size() / 3), nnrm = (int)(normals.size() / 3);
    Material mat = material ? *material : default_material();
    Poly* polys = (Poly*)calloc(faces.size() > 0 ? faces.size() : 1, sizeof(Poly));
    if (!polys) return io_fail(GPT_ERR_OUT_OF_MEMORY, "out of memory");
    for (size_t i = 0; i < faces.size(); i++) {
        const std::vector<int>& fc = faces[i];
        if (fc.size() < 3) { free(polys); return io_fail(GPT_ERR_PARSE, "face with fewer than 3 vertices"); }
        for (int k = 0; k < 3; k++)
            if (fc[(size_t)k] < 0 || fc[(size_t)k] >= npts) { free(polys); return io_fail(GPT_ERR_PARSE, "face vertex index out of range"); }
        Poly& p = polys[i];
        p.position.x = (float)0.0; p.position.y = (float)-0.8; p.position.z = (float)0.0;
        const float* a = &points[3 * (size_t)fc[0]];
        const float* b = &points[3 * (size_t)fc[1]];
        const float* c = &points[3 * (size_t)fc[2]];
        p.p0.x = a[0]; p.p0.y = a[1]; p.p0.z = a[2];
        p.p1.x = b[0]; p.p1.y = b[1]; p.p1.z = b[2];
        p.p2.x = c[0]; p.p2.y = c[1]; p.p2.z = c[2];
        if (i < facenormals.size() && !facenormals[i].empty()) {
            int ni = facenormals[i][0];
            if (ni < 0 || ni >= nnrm) { free(polys); return io_fail(GPT_ERR_PARSE, "face normal index out of range"); }
            p.n.x = normals[3 * (size_t)ni]; p.n.y = normals[3 * (size_t)ni + 1]; p.n.z = normals[3 * (size_t)ni + 2];
        } else {
            // no normals in the file: geometric normal, normalize(cross(p1-p0, p2-p0)) (obj.cpp:87-89)
            float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2];
            float vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
            float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
            float len = sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0) { nx /= len; ny /= len; nz /= len; }
            p.n.x = nx; p.n.y = ny; p.n.z = nz;
        }
        p.material = mat;
    }
    if (bbox) {
        for (int k = 0; k < 6; k++) bbox[k] = 0;
        for (int i = 0; i < npts; i++)
            for (int k = 0; k < 3; k++) {
                float v = points[3 * (size_t)i + k];
                if (i == 0 || v < bbox[k]) bbox[k] = v;
                if (i == 0 || v > bbox[3 + k]) bbox[3 + k] = v;
            }
    }
    *out_polys = polys;
    *out_numPolys = (int)faces.size();
    return GPT_OK;
}

}  // extern "C"
