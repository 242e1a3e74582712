// src/kazen/renderer_mi355x.cpp - NEW file of the MI355X integration; takes the place of src/kazen/renderer.cpp in the build.
// kazen::renderer::render(Scene*, filename) with the region renderer.cpp:85-133 (tbb::parallel_for over image blocks and everything it calls)
// done by libkazen_mi355x on the GPU(s). Reads the scene ONLY through what the reference exposes - Scene::getMeshes / getCamera / getSampler /
// getIntegrator, Mesh::getVertexPositions / getVertexNormals / getVertexTexCoords / getIndices / getBSDF / getLight, Camera::getOutputSize /
// getReconstructionFilter - plus the describe() virtuals and Scene::getBackground() that INTEGRATION.md adds.
#include <kazen/renderer.h>
#include <kazen/scene.h>
#include <kazen/camera.h>
#include <kazen/sampler.h>
#include <kazen/integrator.h>
#include <kazen/mesh.h>
#include <kazen/bsdf.h>
#include <kazen/light.h>
#include <kazen/texture.h>
#include <kazen/rfilter.h>
#include <kazen/block.h>
#include <kazen/bitmap.h>
#include <kazen/mi355x.h>

#include <cstring>

NAMESPACE_BEGIN(kazen)
NAMESPACE_BEGIN(mi355x)

int Rows::texture(const Texture<Color3f> *t) {
    if (!t) return 0;
    auto it = seen.find(t);
    if (it != seen.end()) return it->second;
    KzTexture row;
    std::memset(&row, 0, sizeof row);
    row.child[0] = row.child[1] = row.child[2] = -1;
    if (!t->describe(row, *this))                                       // (children first: the override calls texture() on them)
        throw Exception("Texture {} is not on the MI355X path (constanttexture, imagetexture, colorramp, blend)", t->toString());
    textures.push_back(row);
    return seen[t] = (int) textures.size();
}

void Rows::color(const Texture<Color3f> *t, float dst[3], int32_t &id) {
    if (!t) throw Exception("a BSDF on the MI355X path lacks one of its texture children");
    KzTexture row;
    std::memset(&row, 0, sizeof row);
    Rows probe;                                                         // (a constanttexture has no children: nothing lands here)
    if (t->describe(row, probe) && row.type == KZ_TEX_CONSTANT) { dst[0] = row.color[0]; dst[1] = row.color[1]; dst[2] = row.color[2]; id = 0; }
    else id = texture(t);
}

void Rows::scalar(const Texture<Color3f> *t, float &dst, int32_t &id) {
    float c[3] = {0.f, 0.f, 0.f};
    color(t, c, id);
    if (id == 0) dst = c[0];
}

int Rows::nested(const BSDF *b) {
    if (!b) throw Exception("normalmap without a nested BSDF");
    nestedBsdfs.push_back(b);
    return nestedBase + (int) nestedBsdfs.size() - 1;
}

int Rows::image(int width, int height, int channels, int format, const void *pixels) {
    const size_t bytes = (size_t) width * height * channels * (format == KZ_PIXEL_F32 ? 4 : 1);
    if (!pixels || bytes == 0) throw Exception("imagetexture without a decoded raster");
    rasters.emplace_back(new unsigned char[bytes]);
    std::memcpy(rasters.back().get(), pixels, bytes);
    KzImage im;
    std::memset(&im, 0, sizeof im);
    im.pixels = rasters.back().get(); im.width = width; im.height = height; im.channels = channels; im.format = format;
    images.push_back(im);
    return (int) images.size() - 1;
}

DeviceScene::DeviceScene(const Scene *scene) {
    std::memset(&m_desc, 0, sizeof m_desc);
    const std::vector<Mesh *> &meshes = scene->getMeshes();                                  // scene.h:42
    m_rows.nestedBase = (int) meshes.size();                                                 // a mesh always has a BSDF after Mesh::activate (mesh.cpp:25-28)
    for (const Mesh *m : meshes) {
        KzMesh k;
        std::memset(&k, 0, sizeof k);
        k.V = m->getVertexPositions().data();  k.nV = m->getVertexCount();                   // Eigen col-major 3 x nV == xyz, stride 12 B (mesh.h:176-179)
        k.F = m->getIndices().data();          k.nF = m->getTriangleCount();
        k.N  = m->getVertexNormals().size()   ? m->getVertexNormals().data()   : nullptr;
        k.UV = m->getVertexTexCoords().size() ? m->getVertexTexCoords().data() : nullptr;
        KzBSDF b;
        std::memset(&b, 0, sizeof b);
        if (!m->getBSDF() || !m->getBSDF()->describe(b, m_rows))
            throw Exception("BSDF {} is not on the MI355X path", m->getBSDF() ? m->getBSDF()->toString() : std::string("<none>"));
        k.bsdf = (int32_t) m_bsdfs.size();
        m_bsdfs.push_back(b);
        k.light = -1;
        if (m->isLight()) {
            KzLight l;
            std::memset(&l, 0, sizeof l);
            if (!m->getLight()->describe(l)) throw Exception("Light {} is not on the MI355X path", m->getLight()->toString());
            k.light = (int32_t) m_lights.size();
            m_lights.push_back(l);
        }
        m_meshes.push_back(k);
    }
    for (size_t i = 0; i < m_rows.nestedBsdfs.size(); ++i) {                                 // the rows normalmaps wrap, behind the meshes' own
        KzBSDF b;
        std::memset(&b, 0, sizeof b);
        if (!m_rows.nestedBsdfs[i]->describe(b, m_rows)) throw Exception("BSDF {} (nested in a normalmap) is not on the MI355X path", m_rows.nestedBsdfs[i]->toString());
        if (b.type == KZ_BSDF_NORMALMAP) throw Exception("a normalmap nested in a normalmap is not on the MI355X path");
        m_bsdfs.push_back(b);
    }
    if (!scene->getCamera()->describe(m_desc.camera)) throw Exception("Camera {} is not on the MI355X path", scene->getCamera()->toString());
    if (!scene->getSampler()->describe(m_desc.sampler)) throw Exception("Sampler {} is not on the MI355X path", scene->getSampler()->toString());
    if (!scene->getIntegrator()->describe(m_desc.integrator)) throw Exception("Integrator {} is not on the MI355X path (path_mis)", scene->getIntegrator()->toString());
    if (scene->getBackground() && !scene->getBackground()->describeBackground(m_desc.background, m_rows))
        throw Exception("Scene background {} is not on the MI355X path", scene->getBackground()->toString());
    m_desc.abiVersion = KZ_ABI_VERSION;
    m_desc.meshes = m_meshes.data();          m_desc.nMeshes = (uint32_t) m_meshes.size();
    m_desc.bsdfs = m_bsdfs.data();            m_desc.nBsdfs = (uint32_t) m_bsdfs.size();
    m_desc.lights = m_lights.data();          m_desc.nLights = (uint32_t) m_lights.size();
    m_desc.textures = m_rows.textures.data(); m_desc.nTextures = (uint32_t) m_rows.textures.size();
    m_desc.images = m_rows.images.data();     m_desc.nImages = (uint32_t) m_rows.images.size();
    if (kz_scene_create(&m_desc, &m_handle) != KZ_OK)                                        // Scene::activate -> Accel::build (accel.cpp:25-61): the host BVH
        throw Exception("kz_scene_create: {}", kz_last_error());
}

DeviceScene::~DeviceScene() {
    if (m_handle) kz_scene_destroy(m_handle);
}

void DeviceScene::render(ImageBlock &result, std::vector<int> devices, const KzRenderOpts *opts) {
    if (devices.empty()) for (int d = 0; d < kz_device_count(); ++d) devices.push_back(d);
    if (devices.empty()) throw Exception("no HIP device visible (the MI355X path has no CPU fallback)");
    int32_t w = 0, h = 0, border = 0;
    kz_film_dims(m_handle, &w, &h, &border);
    const size_t nFloats = (size_t) (w + 2 * border) * (h + 2 * border) * 4;
    if ((size_t) result.size() * 4 != nFloats) throw Exception("ImageBlock of {} texels for a film of {}", (size_t) result.size(), nFloats / 4);
    float *film = (float *) result.data();                                                   // row-major Color4f (rgb * weight, weight), border included (block.cpp:30)
    if (devices.size() == 1) {
        KzRenderOpts o;
        std::memset(&o, 0, sizeof o);                                                        // zeros: every sample, the whole frame, the default pipeline
        if (opts) o = *opts;
        if (kz_scene_upload(m_handle, devices[0]) != KZ_OK) throw Exception("kz_scene_upload: {}", kz_last_error());
        if (kz_render_tiles(m_handle, &o, nullptr, 0, devices[0], film, nFloats) != KZ_OK) throw Exception("kz_render_tiles: {}", kz_last_error());
    } else {
        std::vector<int32_t> devs(devices.begin(), devices.end());
        if (kz_render_multi(m_handle, opts, devs.data(), (uint32_t) devs.size(), /*tileSize*/ 0, film, nFloats, nullptr) != KZ_OK)
            throw Exception("kz_render_multi: {}", kz_last_error());
    }
}

NAMESPACE_END(mi355x)

NAMESPACE_BEGIN(renderer)

void render(Scene *scene, const std::string &filename) {
    mi355x::DeviceScene deviceScene(scene);
    const Camera *camera = scene->getCamera();
    ImageBlock result(camera->getOutputSize(), camera->getReconstructionFilter());          // renderer.cpp:81
    deviceScene.render(result, std::vector<int>());                                          // every GPU of the node; std::vector<int>(1, 0) for one
    std::unique_ptr<Bitmap> bitmap(result.toBitmap());                                       // normalised bitmap, as renderer.cpp:140
    bitmap->savePNG(filename.substr(0, filename.find_last_of('.')));                         // "<scene>.xml" -> "<scene>" (+ ".png" by savePNG), renderer.cpp:143-152
}

NAMESPACE_END(renderer)
NAMESPACE_END(kazen)
