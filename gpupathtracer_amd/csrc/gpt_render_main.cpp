// gpt_render -- headless renderer: what the reference's display.cpp main loop does
// (display.cpp:69-110,183-259: build the camera, call PathTracer::render once per frame,
// show floatTo8Bit(sum / passes)), without a window: the image goes to a PPM (and
// optionally a PFM with raw radiance).  Pure caller of the C-ABI in include/gpt.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpt.h"

static void usage() {
    fprintf(stderr,
            "usage: gpt_render [--width W] [--height H] [--spp N] [--depth D] [--out image.ppm] [--pfm image.pfm] [--exr image.exr]\n"
            "                  [--obj mesh.obj [--material NAME] [--scale S] [--translate X Y Z]] [--no-spheres]\n"
            "                  [--yaw R] [--pitch R] [--radius R] [--aperture R] [--fovx DEG] [--center X Y Z]\n"
            "                  [--scene NAME --assets DIR] [--brute-force] [--device N] [--gpus N]\n"
            "  --scene NAME: one of the named scenes (Default, CornellBox, Diamond-NoDOF, Monkey, Speed, Subsurface); DIR holds\n"
            "            the reference's .obj files.  Sets the spheres, the meshes and the camera; later flags override the camera\n"
            "  --gpus N: the spp passes are split across devices 0..N-1 (one host thread and one scene per GPU,\n"
            "            disjoint ranges of pass counters) and the images summed with one RCCL all-reduce\n");
}

int main(int argc, char** argv) {
    int width = 512, height = 512, spp = 64, depth = 30, device = 0, gpus = 0;   // display.cpp:33-34, MAX_TRACE_DEPTH
    std::string out = "render.ppm", pfm, exr, obj, material = "glass", scene_name, assets;
    float scale = 1, tr[3] = {0, 0, 0};
    float yaw = 0, pitch = 0.3f, radius = 4, aperture = 0.04f, fovx = 45, center[3] = {0, 0, 0};
    bool spheres_on = true, brute = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { usage(); exit(2); } };
        if (a == "--width") { need(1); width = atoi(argv[++i]); }
        else if (a == "--height") { need(1); height = atoi(argv[++i]); }
        else if (a == "--spp") { need(1); spp = atoi(argv[++i]); }
        else if (a == "--depth") { need(1); depth = atoi(argv[++i]); }
        else if (a == "--device") { need(1); device = atoi(argv[++i]); }
        else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--pfm") { need(1); pfm = argv[++i]; }
        else if (a == "--exr") { need(1); exr = argv[++i]; }
        else if (a == "--obj") { need(1); obj = argv[++i]; }
        else if (a == "--scene") { need(1); scene_name = argv[++i]; }
        else if (a == "--assets") { need(1); assets = argv[++i]; }
        else if (a == "--material") { need(1); material = argv[++i]; }
        else if (a == "--scale") { need(1); scale = (float)atof(argv[++i]); }
        else if (a == "--translate") { need(3); for (int k = 0; k < 3; k++) tr[k] = (float)atof(argv[++i]); }
        else if (a == "--center") { need(3); for (int k = 0; k < 3; k++) center[k] = (float)atof(argv[++i]); }
        else if (a == "--yaw") { need(1); yaw = (float)atof(argv[++i]); }
        else if (a == "--pitch") { need(1); pitch = (float)atof(argv[++i]); }
        else if (a == "--radius") { need(1); radius = (float)atof(argv[++i]); }
        else if (a == "--aperture") { need(1); aperture = (float)atof(argv[++i]); }
        else if (a == "--fovx") { need(1); fovx = (float)atof(argv[++i]); }
        else if (a == "--no-spheres") spheres_on = false;
        else if (a == "--brute-force") brute = true;
        else { usage(); return 2; }
    }
    if (width <= 0 || height <= 0 || spp <= 0) { usage(); return 2; }

    std::vector<Sphere> spheres;
    if (spheres_on) { spheres.resize((size_t)gpt_default_scene_sphere_count()); gpt_default_scene(spheres.data()); }
    Poly* polys = nullptr;
    int numPolys = 0;
    if (!scene_name.empty()) {
        // a named scene: its spheres, meshes and camera; camera flags given AFTER --scene on the command line override it
        Sphere* ps = nullptr;
        int ns = 0;
        gpt_preset_view view;
        if (gpt_scene_preset(scene_name.c_str(), assets.empty() ? nullptr : assets.c_str(), &ps, &ns, &polys, &numPolys, &view) != GPT_OK) {
            fprintf(stderr, "%s\n", gpt_last_error());
            return 1;
        }
        spheres.assign(ps, ps + ns);
        gpt_free(ps);
        bool seen = false;
        bool set_yaw = false, set_pitch = false, set_radius = false, set_aperture = false, set_fovx = false, set_center = false;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            if (a == "--scene") seen = true;
            if (!seen) continue;
            set_yaw |= a == "--yaw"; set_pitch |= a == "--pitch"; set_radius |= a == "--radius";
            set_aperture |= a == "--aperture"; set_fovx |= a == "--fovx"; set_center |= a == "--center";
        }
        if (!set_yaw) yaw = view.yaw;
        if (!set_pitch) pitch = view.pitch;
        if (!set_radius) radius = view.radius;
        if (!set_aperture) aperture = view.apertureRadius;
        if (!set_fovx) fovx = view.fovx_degrees;
        if (!set_center) for (int k = 0; k < 3; k++) center[k] = view.center[k];
        printf("Scene %s: %d spheres, %d triangles\n", scene_name.c_str(), ns, numPolys);
    } else if (!obj.empty()) {
        Material m;
        if (gpt_material_preset(material.c_str(), &m) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
        float bbox[6];
        if (gpt_obj_load(obj.c_str(), &m, &polys, &numPolys, bbox) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
        printf("Loaded %d faces from %s\n", numPolys, obj.c_str());
        for (int i = 0; i < numPolys; i++) {
            GPT_FLOAT3* v[3] = {&polys[i].p0, &polys[i].p1, &polys[i].p2};
            for (int k = 0; k < 3; k++) {
                v[k]->x = v[k]->x * scale + tr[0]; v[k]->y = v[k]->y * scale + tr[1]; v[k]->z = v[k]->z * scale + tr[2];
            }
        }
    }
    gpt_render_params rp;
    gpt_render_params_default(&rp);
    rp.max_depth = depth;
    rp.use_kdtree = brute ? 0 : 1;
    Camera cam;
    gpt_orbit_camera((float)width, (float)height, fovx, yaw, pitch, radius, aperture, center, &cam);

    // One scene per GPU.  Without --gpus: a single scene on --device.  With --gpus N: device g renders the
    // contiguous g-th share of the pass counters 0..spp-1 (every sample is a function of its counter alone,
    // so the union is the single-GPU render's sample set), then one RCCL sum makes every image the total.
    const int n_dev = gpus > 0 ? gpus : 1;
    std::vector<gpt_scene*> scenes((size_t)n_dev, nullptr);
    std::vector<std::string> errors((size_t)n_dev);
    auto on_all = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int g = 0; g < n_dev; g++) th.emplace_back([&, g]() { if (fn(g) != GPT_OK) errors[(size_t)g] = gpt_last_error(); });
        for (auto& t : th) t.join();
        for (int g = 0; g < n_dev; g++)
            if (!errors[(size_t)g].empty()) { fprintf(stderr, "GPU %d: %s\n", g, errors[(size_t)g].c_str()); return false; }
        return true;
    };
    auto t0 = std::chrono::steady_clock::now();
    if (!on_all([&](int g) {
            gpt_status st = gpt_scene_create(spheres.data(), (int)spheres.size(), polys, numPolys, &rp, gpus > 0 ? g : device, &scenes[(size_t)g]);
            return st != GPT_OK ? st : gpt_scene_reset(scenes[(size_t)g], width * height);
        })) return 1;
    auto t1 = std::chrono::steady_clock::now();
    if (!on_all([&](int g) {
            const int base = spp / n_dev, extra = spp % n_dev;
            const int begin = g * base + (g < extra ? g : extra), end = begin + base + (g < extra ? 1 : 0);
            return gpt_render_range(scenes[(size_t)g], &cam, begin, end, 1);
        })) return 1;
    if (gpus > 0 && gpt_accum_allreduce(scenes.data(), n_dev) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
    auto t2 = std::chrono::steady_clock::now();
    gpt_scene* scene = scenes[0];
    std::vector<float> img((size_t)width * height * 3);
    int passes = 0;
    gpt_stats st;
    if (gpt_accum_download(scene, img.data(), &passes) != GPT_OK || gpt_get_stats(scene, &st) != GPT_OK) {
        fprintf(stderr, "%s\n", gpt_last_error());
        return 1;
    }
    for (int g = 1; g < n_dev; g++) {
        gpt_stats sg;
        if (gpt_get_stats(scenes[(size_t)g], &sg) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
        st.path_bounces += sg.path_bounces;
    }
    double sec = std::chrono::duration<double>(t2 - t1).count();
    // the reference's overlay: iterations, seconds, FPS (display.cpp:232-250)
    printf("Iterations: %d\nSeconds: %.3f\nFPS: %.2f\n", passes, sec, passes / (sec > 0 ? sec : 1));
    if (gpus > 0) printf("GPUs: %d (spp-sharded, one RCCL sum)\n", n_dev);
    printf("Scene upload + kd-tree: %.3f s; %.1f Mpath-bounces/s\n", std::chrono::duration<double>(t1 - t0).count(),
           (double)st.path_bounces / (sec > 0 ? sec : 1) / 1e6);
    if (gpt_write_ppm(out.c_str(), img.data(), width, height, passes) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
    if (!pfm.empty() && gpt_write_pfm(pfm.c_str(), img.data(), width, height, passes) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
    if (!exr.empty() && gpt_write_exr(exr.c_str(), img.data(), width, height, passes) != GPT_OK) { fprintf(stderr, "%s\n", gpt_last_error()); return 1; }
    for (gpt_scene* sc : scenes) gpt_scene_destroy(sc);
    gpt_shutdown();      // the RCCL communicators of --gpus
    gpt_free(polys);
    return 0;
}
