// Drives the C++ host mirror (nano-kazen_amd/host/kazen_host.hpp) the way nano-kazen's parser drives its object
// model (parser.cpp:116-301: children first, createInstance, addChild, activate), then hands the activated Scene to the ADAPTER a maintainer
// adds to a kazen tree - nano-kazen_amd/host/adapter/renderer_mi355x.cpp + adapter/kazen/mi355x.h, compiled UNCHANGED into this program (the
// mirror answers to the reference's header names through mirror_tree/kazen/*.h) - and either dumps the description the adapter produced as
// JSON (no GPU needed) or renders on the GPU and writes the linear rgb bitmap to a file. The mirror's plugin classes keep their parameters
// private like the reference's: whatever this program prints came out through the describe() virtuals INTEGRATION.md lists.
// Build: g++ -std=c++17 -I include -I nano-kazen_amd/host/mirror_tree -I nano-kazen_amd/host/adapter host_mirror_test.cpp nano-kazen_amd/host/adapter/renderer_mi355x.cpp -lkazen_mi355x
#include <kazen/renderer.h>
#include <kazen/scene.h>
#include <kazen/block.h>
#include <kazen/bitmap.h>
#include <kazen/mi355x.h>
#include <kazen_mi355x_dev.h>                    // (the test also reads the BVH statistics: development surface)
#include "../../nano-kazen_amd/host/kazen_sceneio.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

using namespace kazen;

static Mesh *quadMesh(const float p[4][3], const float n[3]) {
    auto *m = static_cast<Mesh *>(ObjectFactory::createInstance("obj", PropertyList()));
    std::vector<float> V, N, UV = {0, 0, 1, 0, 1, 1, 0, 1};
    for (int i = 0; i < 4; ++i) for (int a = 0; a < 3; ++a) { V.push_back(p[i][a]); N.push_back(n[a]); }
    m->setBuffers(V, {0, 1, 2, 0, 2, 3}, N, UV);
    return m;
}
static Object *constTex(const char *id, float r, float g, float b) {
    PropertyList p; p.setColor("color", Color3f(r, g, b));
    Object *t = ObjectFactory::createInstance("constanttexture", p); t->setId(id); return t;
}

static Scene *buildScene() {
    auto *scene = static_cast<Scene *>(ObjectFactory::createInstance("scene", PropertyList()));
    { PropertyList p; p.setInteger("maxDepth", 4); scene->addChild(ObjectFactory::createInstance("path_mis", p)); }
    { PropertyList p; p.setInteger("sampleCount", 8); p.setInteger("seed", 0); scene->addChild(ObjectFactory::createInstance("independent", p)); }
    { PropertyList p; p.setInteger("width", 64); p.setInteger("height", 48); p.setFloat("fov", 40.f); p.setFloat("nearClip", 0.1f); p.setFloat("farClip", 100.f);
      p.setTransform("toWorld", Transform::lookAt({0, 0.5f, 3.f}, {0, 0, 0}, {0, 1, 0}));
      Object *cam = ObjectFactory::createInstance("perspective", p);
      PropertyList f; f.setFloat("radius", 2.0f); f.setFloat("stddev", 0.5f);
      cam->addChild(ObjectFactory::createInstance("gaussian", f));
      scene->addChild(cam); }
    const float up[3] = {0, 1, 0}, down[3] = {0, -1, 0}, front[3] = {0, 0, 1};
    const float floorP[4][3] = {{-2, -1, -2}, {2, -1, -2}, {2, -1, 2}, {-2, -1, 2}};
    const float wallP[4][3] = {{-2, -1, -2}, {-2, 2, -2}, {2, 2, -2}, {2, -1, -2}};
    const float lightP[4][3] = {{-0.5f, 1.5f, -0.5f}, {-0.5f, 1.5f, 0.5f}, {0.5f, 1.5f, 0.5f}, {0.5f, 1.5f, -0.5f}};
    const float panelP[4][3] = {{-0.8f, -0.6f, 0}, {0.8f, -0.6f, 0}, {0.8f, 0.6f, -0.6f}, {-0.8f, 0.6f, -0.6f}};
    const float panelN[3] = {0, 0.70710678f, 0.70710678f};
    scene->addChild(quadMesh(floorP, up));                                    // no bsdf child: default diffuse 0.5
    { Mesh *m = quadMesh(wallP, front); PropertyList p; p.setColor("albedo", Color3f(0.7f, 0.3f, 0.3f)); m->addChild(ObjectFactory::createInstance("diffuse", p)); scene->addChild(m); }
    { Mesh *m = quadMesh(panelP, panelN); PropertyList p; p.setFloat("clearcoat", 1.0f); p.setFloat("sheen", 0.5f);
      Object *b = ObjectFactory::createInstance("kazenstandard", p);
      b->addChild(constTex("baseColor", 0.8f, 0.6f, 0.2f)); b->addChild(constTex("roughness", 0.4f, 0.4f, 0.4f)); b->addChild(constTex("metallic", 0.f, 0.f, 0.f));
      m->addChild(b); scene->addChild(m); }
    { Mesh *m = quadMesh(lightP, down); PropertyList p; p.setFloat("intensity", 12.f); p.setColor("color", Color3f(1.f, 0.9f, 0.8f)); m->addChild(ObjectFactory::createInstance("area", p)); scene->addChild(m); }
    { PropertyList p; p.setFloat("intensity", 0.25f); Object *bg = ObjectFactory::createInstance("background", p); bg->addChild(constTex("", 0.5f, 0.6f, 1.0f)); scene->addChild(bg); }
    scene->activate();
    return scene;
}

// SURVEY 8f rank 4 through the plugin surface: lambertian + imagetexture (a PPM file), a normalmap (raster handed over with
// setRaster) over a kazenstandard whose roughness is a colorramp of a blend.
static Scene *buildTexturedScene(const char *ppm) {
    auto *scene = static_cast<Scene *>(ObjectFactory::createInstance("scene", PropertyList()));
    { PropertyList p; p.setInteger("maxDepth", 4); p.setBoolean("regularization", true); scene->addChild(ObjectFactory::createInstance("path_mis", p)); }
    { PropertyList p; p.setInteger("sampleCount", 8); p.setInteger("seed", 0); scene->addChild(ObjectFactory::createInstance("independent", p)); }
    { PropertyList p; p.setInteger("width", 64); p.setInteger("height", 48); p.setFloat("fov", 40.f); p.setFloat("nearClip", 0.1f); p.setFloat("farClip", 100.f);
      p.setTransform("toWorld", Transform::lookAt({0, 0.5f, 3.f}, {0, 0, 0}, {0, 1, 0}));
      scene->addChild(ObjectFactory::createInstance("perspective", p)); }
    const float up[3] = {0, 1, 0}, down[3] = {0, -1, 0};
    const float floorP[4][3] = {{-2, -1, -2}, {2, -1, -2}, {2, -1, 2}, {-2, -1, 2}};
    const float lightP[4][3] = {{-0.5f, 1.5f, -0.5f}, {-0.5f, 1.5f, 0.5f}, {0.5f, 1.5f, 0.5f}, {0.5f, 1.5f, -0.5f}};
    const float panelP[4][3] = {{-0.8f, -0.6f, 0}, {0.8f, -0.6f, 0}, {0.8f, 0.6f, -0.6f}, {-0.8f, 0.6f, -0.6f}};
    const float panelN[3] = {0, 0.70710678f, 0.70710678f};
    auto image = [&](const char *id, float scale, const char *colorspace) {
        PropertyList p; p.setString("filename", ppm); p.setFloat("scale", scale); p.setString("colorspace", colorspace);
        Object *t = ObjectFactory::createInstance("imagetexture", p); t->setId(id); return t;
    };
    { Mesh *m = quadMesh(floorP, up); Object *b = ObjectFactory::createInstance("lambertian", PropertyList()); b->addChild(image("", 3.0f, "srgb")); m->addChild(b); scene->addChild(m); }
    { Mesh *m = quadMesh(panelP, panelN);
      Object *nm = ObjectFactory::createInstance("normalmap", PropertyList());
      { PropertyList p; p.setString("colorspace", "linear");
        auto *t = static_cast<ImageTexture *>(ObjectFactory::createInstance("imagetexture", p));
        const uint8_t px[2 * 2 * 3] = {128, 128, 255, 160, 128, 240, 128, 160, 240, 100, 110, 235};
        t->setRaster(2, 2, 3, KZ_PIXEL_U8, px);
        nm->addChild(t); }
      PropertyList p; p.setFloat("clearcoat", 1.0f);
      Object *kiss = ObjectFactory::createInstance("kazenstandard", p);
      kiss->addChild(image("baseColor", 1.0f, "srgb"));
      { PropertyList r; r.setFloat("min", 0.2f); r.setFloat("max", 0.7f); Object *ramp = ObjectFactory::createInstance("colorramp", r); ramp->setId("roughness");
        PropertyList bp; bp.setString("blendmode", "multiply"); Object *bl = ObjectFactory::createInstance("blend", bp);
        bl->addChild(image("input1", 2.0f, "linear")); { Object *c = constTex("input2", 0.9f, 0.9f, 0.9f); bl->addChild(c); }
        ramp->addChild(bl); kiss->addChild(ramp); }
      kiss->addChild(constTex("metallic", 0.f, 0.f, 0.f));
      nm->addChild(kiss); m->addChild(nm); scene->addChild(m); }
    { Mesh *m = quadMesh(lightP, down); PropertyList p; p.setFloat("intensity", 12.f); m->addChild(ObjectFactory::createInstance("area", p)); scene->addChild(m); }
    scene->activate();
    return scene;
}

// what renderer::render does before it writes the file: the adapter's DeviceScene, the full-frame ImageBlock, toBitmap
static std::vector<float> renderRgb(mi355x::DeviceScene &ds, const Scene *scene, const std::vector<int> &devices) {
    ImageBlock result(scene->getCamera()->getOutputSize(), scene->getCamera()->getReconstructionFilter());
    ds.render(result, devices);
    std::unique_ptr<Bitmap> bm(result.toBitmap());
    return std::vector<float>(bm->data(), bm->data() + (size_t)bm->cols() * bm->rows() * 3);
}

template <class F> static std::string thrown(F f) { try { f(); } catch (const Exception &e) { return e.what(); } return ""; }

int main(int argc, char **argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "--render")) {
        std::unique_ptr<Scene> scene(buildScene());
        mi355x::DeviceScene ds(scene.get());
        std::vector<float> rgb = renderRgb(ds, scene.get(), {0});
        std::ofstream(argv[2], std::ios::binary).write((const char *)rgb.data(), (std::streamsize)(rgb.size() * sizeof(float)));
        double s = 0; for (float v : rgb) s += v;
        std::printf("{\"pixels\": %zu, \"mean\": %.6f}\n", rgb.size() / 3, s / rgb.size());
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "--xml")) {               // loadFromXML (parser.cpp:10-305) + the OBJ loader (mesh.cpp:200-343)
        std::string err;
        std::unique_ptr<Object> root;
        try { root.reset(loadFromXML(argv[2])); } catch (const Exception &e) { err = e.what(); }
        if (!root) { for (auto &c : err) if (c == '"') c = '\''; std::printf("{\"error\": \"%s\"}\n", err.c_str()); return 0; }
        Scene *scene = static_cast<Scene *>(root.get());
        std::unique_ptr<mi355x::DeviceScene> dsp;
        try { dsp.reset(new mi355x::DeviceScene(scene)); } catch (const Exception &e) { err = e.what(); }
        if (!dsp) { for (auto &c : err) if (c == '"') c = '\''; std::printf("{\"error\": \"%s\"}\n", err.c_str()); return 0; }
        mi355x::DeviceScene &ds = *dsp;
        const KzSceneDesc &d = ds.desc();
        if (argc >= 4) {
            std::vector<float> rgb = renderRgb(ds, scene, {0});
            std::ofstream(argv[3], std::ios::binary).write((const char *)rgb.data(), (std::streamsize)(rgb.size() * sizeof(float)));
        }
        std::printf("{\"nMeshes\": %u, \"nBsdfs\": %u, \"nLights\": %u, \"nTextures\": %u, \"nImages\": %u, \"meshes\": [", d.nMeshes, d.nBsdfs, d.nLights, d.nTextures, d.nImages);
        for (uint32_t i = 0; i < d.nMeshes; ++i) {
            const KzMesh &m = d.meshes[i];
            double sv = 0, sn = 0, su = 0; unsigned long long sf = 0;
            for (uint32_t k = 0; k < 3 * m.nV; ++k) { sv += m.V[k]; if (m.N) sn += m.N[k]; }
            if (m.UV) for (uint32_t k = 0; k < 2 * m.nV; ++k) su += m.UV[k];
            for (uint32_t k = 0; k < 3 * m.nF; ++k) sf += (unsigned long long)m.F[k] * (k % 7 + 1);
            std::printf("%s[%u, %u, %d, %d, %d, %d, %.9g, %.9g, %.9g, %llu]", i ? ", " : "", m.nV, m.nF, m.N ? 1 : 0, m.UV ? 1 : 0, m.bsdf, m.light, sv, sn, su, sf);
        }
        std::printf("], \"bsdfTypes\": [");
        for (uint32_t i = 0; i < d.nBsdfs; ++i) std::printf("%s%d", i ? ", " : "", d.bsdfs[i].type);
        std::printf("], \"camera\": [%d, %d, %d, %.9g, %.9g, %.9g, %d, %.9g], \"toWorld\": [", d.camera.type, d.camera.width, d.camera.height, d.camera.fov, d.camera.nearClip,
                    d.camera.farClip, d.camera.rfilter.type, d.camera.rfilter.radius);
        for (int i = 0; i < 16; ++i) std::printf("%s%.9g", i ? ", " : "", d.camera.toWorld[i]);
        std::printf("], \"sampler\": [%d, %u, %llu], \"integrator\": [%d, %.9g, %d, %.9g], \"background\": [%d, %.9g, %.9g, %.9g, %.9g], \"backgroundTexture\": %d, \"lights\": [", d.sampler.type,
                    d.sampler.sampleCount, (unsigned long long)d.sampler.seed, d.integrator.maxDepth, d.integrator.traceBias, d.integrator.regularization,
                    d.integrator.accumulatedRoughness, d.background.present, d.background.color[0], d.background.color[1], d.background.color[2], d.background.intensity,
                    d.background.texture);
        for (uint32_t i = 0; i < d.nLights; ++i) std::printf("%s[%.9g, %.9g, %.9g, %.9g, %d]", i ? ", " : "", d.lights[i].color[0], d.lights[i].color[1], d.lights[i].color[2], d.lights[i].intensity, d.lights[i].primaryVisibility);
        std::printf("]}\n");
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "--bitmap")) {            // Bitmap::savePNG / saveEXR of a deterministic 37 x 5 gradient (no GPU)
        const int w = 37, h = 5;
        Bitmap bm(w, h);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int c = 0; c < 3; ++c)
            bm.data()[((size_t)y * w + x) * 3 + c] = (float)(x + 1) / (float)w * (c == 0 ? 1.2f : c == 1 ? 0.5f : 0.01f) + (float)y * 0.003f - (c == 2 && x == 0 ? 0.5f : 0.f);
        bm.savePNG(argv[2]); bm.saveEXR(argv[2]);
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "--render-multi")) {      // renderer::render(scene, devices): argv[3] = device list "0,1,.."
        std::unique_ptr<Scene> scene(buildScene());
        std::vector<int> devs;
        for (const char *c = argv[3]; *c;) { devs.push_back(std::atoi(c)); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
        mi355x::DeviceScene ds(scene.get());
        std::vector<float> rgb = renderRgb(ds, scene.get(), devs);
        std::ofstream(argv[2], std::ios::binary).write((const char *)rgb.data(), (std::streamsize)(rgb.size() * sizeof(float)));
        std::printf("{\"pixels\": %zu, \"devices\": %zu}\n", rgb.size() / 3, devs.size());
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "--render-in-turn")) {    // ONE DeviceScene rendered on each device of the list in turn: argv[2] = output stem, argv[3] = "0,1,.."
        std::unique_ptr<Scene> scene(buildScene());
        mi355x::DeviceScene ds(scene.get());
        int n = 0;
        for (const char *c = argv[3]; *c;) {
            const int dev = std::atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c;
            std::vector<float> rgb = renderRgb(ds, scene.get(), {dev});            // (the film comes from the device that rendered it: kz_render_tiles hands it back)
            std::ofstream(std::string(argv[2]) + "." + std::to_string(n++), std::ios::binary).write((const char *)rgb.data(), (std::streamsize)(rgb.size() * sizeof(float)));
        }
        std::printf("{\"renders\": %d}\n", n);
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "--render-png")) {        // renderer::render(scene, filename) (renderer.cpp:72-153)
        std::unique_ptr<Scene> scene(buildScene());
        renderer::render(scene.get(), std::string(argv[2]));          // the adapter's drop-in itself: every visible GPU, <stem>.png
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "--textured")) {
        std::unique_ptr<Scene> scene(buildTexturedScene(argv[2]));
        mi355x::DeviceScene ds(scene.get());
        const KzSceneDesc &d = ds.desc();
        if (argc >= 4) {
            std::vector<float> rgb = renderRgb(ds, scene.get(), {0});
            std::ofstream(argv[3], std::ios::binary).write((const char *)rgb.data(), (std::streamsize)(rgb.size() * sizeof(float)));
        }
        std::printf("{\"nBsdfs\": %u, \"nTextures\": %u, \"nImages\": %u, \"bsdfs\": [", d.nBsdfs, d.nTextures, d.nImages);
        for (uint32_t i = 0; i < d.nBsdfs; ++i)
            std::printf("%s[%d, %d, %d, %d, %d, %d, %g]", i ? ", " : "", d.bsdfs[i].type, d.bsdfs[i].albedoTex, d.bsdfs[i].roughnessTex, d.bsdfs[i].metallicTex, d.bsdfs[i].normalTex,
                        d.bsdfs[i].nested, d.bsdfs[i].metallic);
        std::printf("], \"textures\": [");
        for (uint32_t i = 0; i < d.nTextures; ++i)
            std::printf("%s[%d, %d, %g, %d, %g, %g, %d, %d, %d, %d]", i ? ", " : "", d.textures[i].type, d.textures[i].image, d.textures[i].scale, d.textures[i].srgb, d.textures[i].rampMin,
                        d.textures[i].rampMax, d.textures[i].blendMode, d.textures[i].child[0], d.textures[i].child[1], d.textures[i].child[2]);
        std::printf("], \"images\": [");
        for (uint32_t i = 0; i < d.nImages; ++i) std::printf("%s[%d, %d, %d, %d]", i ? ", " : "", d.images[i].width, d.images[i].height, d.images[i].channels, d.images[i].format);
        std::string e1 = thrown([] { std::unique_ptr<Object> t(ObjectFactory::createInstance("blend", PropertyList())); t->addChild(constTex("nosuchslot", 0, 0, 0)); });
        std::string e2 = thrown([] { PropertyList p; p.setString("filename", "/nonexistent/file.ppm"); ObjectFactory::createInstance("imagetexture", p); });
        for (std::string *e : {&e1, &e2}) for (auto &c : *e) if (c == '"') c = '\'';
        std::printf("], \"errors\": [\"%s\", \"%s\"]}\n", e1.c_str(), e2.c_str());
        return 0;
    }
    std::unique_ptr<Scene> scene(buildScene());
    mi355x::DeviceScene ds(scene.get());
    const KzSceneDesc &d = ds.desc();
    KzBvhInfo info; kz_scene_bvh_info(ds.handle(), &info);
    std::string e1 = thrown([] { ObjectFactory::createInstance("whitted", PropertyList()); });
    std::string e2 = thrown([] { ObjectFactory::createInstance("nosuchclass", PropertyList()); });
    std::string e3 = thrown([] { Scene s; s.addChild(ObjectFactory::createInstance("path_mis", PropertyList())); s.activate(); });
    std::string e4 = thrown([] { Scene s; s.addChild(ObjectFactory::createInstance("independent", PropertyList())); s.addChild(ObjectFactory::createInstance("independent", PropertyList())); });
    std::string e5 = thrown([] { std::unique_ptr<Object> c(ObjectFactory::createInstance("perspective", PropertyList())); c->addChild(ObjectFactory::createInstance("diffuse", PropertyList())); });
    auto esc = [](std::string s) { for (auto &c : s) if (c == '"') c = '\''; return s; };
    std::printf("{\"nMeshes\": %u, \"nBsdfs\": %u, \"nLights\": %u, \"meshBsdf\": [%d, %d, %d, %d], \"meshLight\": [%d, %d, %d, %d],\n",
                d.nMeshes, d.nBsdfs, d.nLights, d.meshes[0].bsdf, d.meshes[1].bsdf, d.meshes[2].bsdf, d.meshes[3].bsdf,
                d.meshes[0].light, d.meshes[1].light, d.meshes[2].light, d.meshes[3].light);
    std::printf(" \"kiss\": [%d, %g, %g, %g, %g, %g, %g, %g, %g, %g, %g, %g, %g], \"light\": [%g, %g, %g, %g, %d],\n", d.bsdfs[2].type, d.bsdfs[2].baseColor[0],
                d.bsdfs[2].baseColor[1], d.bsdfs[2].baseColor[2], d.bsdfs[2].roughness, d.bsdfs[2].metallic, d.bsdfs[2].anisotropy, d.bsdfs[2].specular, d.bsdfs[2].specularTint,
                d.bsdfs[2].clearcoat, d.bsdfs[2].clearcoatRoughness, d.bsdfs[2].sheen, d.bsdfs[2].sheenTint, d.lights[0].color[0], d.lights[0].color[1], d.lights[0].color[2],
                d.lights[0].intensity, d.lights[0].primaryVisibility);
    std::printf(" \"camera\": [%d, %d, %g, %g, %g, %d, %g, %g], \"sampler\": [%d, %u, %llu], \"integrator\": [%d, %d, %g, %d, %g], \"background\": [%d, %g, %g, %g, %g],\n",
                d.camera.width, d.camera.height, d.camera.fov, d.camera.nearClip, d.camera.farClip, d.camera.rfilter.type, d.camera.rfilter.radius, d.camera.rfilter.stddev,
                d.sampler.type, d.sampler.sampleCount, (unsigned long long)d.sampler.seed, d.integrator.type, d.integrator.maxDepth, d.integrator.traceBias,
                d.integrator.regularization, d.integrator.accumulatedRoughness, d.background.present, d.background.color[0], d.background.color[1], d.background.color[2], d.background.intensity);
    std::printf(" \"toWorld\": [");
    for (int i = 0; i < 16; ++i) std::printf("%s%.9g", i ? ", " : "", d.camera.toWorld[i]);
    std::printf("],\n \"bvhTris\": %u, \"errors\": [\"%s\", \"%s\", \"%s\", \"%s\", \"%s\"]}\n", info.nTris, esc(e1).c_str(), esc(e2).c_str(), esc(e3).c_str(), esc(e4).c_str(), esc(e5).c_str());
    return 0;
}
