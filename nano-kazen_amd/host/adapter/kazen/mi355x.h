// include/kazen/mi355x.h - NEW file of the MI355X integration: what the describe() overrides of the plugin classes and
// src/kazen/renderer_mi355x.cpp share. Nothing here touches a private member of any plugin: parameters leave the plugins through ONE added
// virtual per interface,
//     bool BSDF::describe(KzBSDF &row, mi355x::Rows &rows) const                  bool Light::describe(KzLight &row) const
//     bool Texture<T>::describe(KzTexture &row, mi355x::Rows &rows) const         bool Camera::describe(KzCamera &row) const
//     bool Texture<T>::describeBackground(KzBackground &row, mi355x::Rows &) const bool ReconstructionFilter::describe(KzFilter &row) const
//     bool Sampler::describe(KzSampler &row) const                                 bool Integrator::describe(KzIntegrator &row) const
// whose default returns false ("this plugin is not on the MI355X path": the adapter throws, never falls back) and which each plugin class on the
// path overrides with a few lines that copy ITS OWN members into the row (INTEGRATION.md lists every override).
#pragma once

#include <kazen/common.h>
#include <kazen_mi355x.h>          // the product header of the library (C ABI), link -lkazen_mi355x

#include <map>
#include <memory>
#include <string>
#include <vector>

NAMESPACE_BEGIN(kazen)

template <typename T> class Texture;
struct Color3f;
class BSDF;
class Scene;
class ImageBlock;

NAMESPACE_BEGIN(mi355x)

/// The tables a scene's BSDF and texture plugins describe themselves into (KzSceneDesc.textures / images and the BSDF rows a normalmap wraps).
struct Rows {
    std::vector<KzTexture> textures;
    std::vector<KzImage> images;

    /// 1-based id of the row of `t` in `textures` (0 for a null pointer); the texture describes itself - children first - the first time it is seen
    int texture(const Texture<Color3f> *t);
    /// A colour-valued texture child of a BSDF (bsdf.cpp:219,255,644,661,1226): a constanttexture is folded into `dst` (id 0), any other goes through the table
    void color(const Texture<Color3f> *t, float dst[3], int32_t &id);
    /// The same for a parameter the BSDF reads as .r() of the colour (bsdf.cpp:1227,1231)
    void scalar(const Texture<Color3f> *t, float &dst, int32_t &id);
    /// Index of the BSDF row a normalmap wraps (KzBSDF.nested): those rows follow the meshes' own rows
    int nested(const BSDF *b);
    /// A decoded raster (copied: the caller's buffer may go away); returns its index in `images`
    int image(int width, int height, int channels, int format, const void *pixels);

    // (state of the builder)
    std::map<const void *, int> seen;
    std::vector<const BSDF *> nestedBsdfs;
    int nestedBase = 0;
    std::vector<std::unique_ptr<unsigned char[]>> rasters;
};

/// An activated Scene flattened into the library's description and built (host BVH); resident on a device from the first render on it.
class DeviceScene {
public:
    /// Walks Scene::getMeshes() / getCamera() / getSampler() / getIntegrator() / getBackground() and the describe() virtuals; throws
    /// kazen::Exception for a plugin that is not on the MI355X path or a description the library rejects
    explicit DeviceScene(const Scene *scene);
    ~DeviceScene();
    DeviceScene(const DeviceScene &) = delete;
    DeviceScene &operator=(const DeviceScene &) = delete;

    const KzSceneDesc &desc() const { return m_desc; }
    KzScene *handle() const { return m_handle; }

    /// renderer.cpp:85-133: every sample of every pixel into `result` (the full-frame ImageBlock of renderer.cpp:81, weighted rgb + weight, border
    /// included). One device: that device renders the frame. Several: tiles dealt over them, the devices' tile rects added on the host in tile
    /// order (block.cpp:87-96). An empty list = every visible device.
    void render(ImageBlock &result, std::vector<int> devices = std::vector<int>(1, 0), const KzRenderOpts *opts = nullptr);

private:
    Rows m_rows;
    std::vector<KzMesh> m_meshes;
    std::vector<KzBSDF> m_bsdfs;
    std::vector<KzLight> m_lights;
    KzSceneDesc m_desc;
    KzScene *m_handle = nullptr;
};

NAMESPACE_END(mi355x)
NAMESPACE_END(kazen)
