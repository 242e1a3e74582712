// kazen_host.hpp — host-side mirror of nano-kazen's plugin surface for the path_mis hot path, above the C ABI
// (include/kazen_mi355x.h). Header-only C++17, no dependencies.
//
// Same class names, registry strings, property names and defaults as the reference:
//   ObjectFactory::createInstance(name, PropertyList)        include/kazen/object.h:133-138, KAZEN_REGISTER_CLASS :144-152
//   Object::addChild / activate / getClassType               include/kazen/object.h:40-75
//   Scene, Mesh, PerspectiveCamera("perspective"), Independent("independent"), PMJ02BN("pmj02bn"),
//   PathMisIntegrator("path_mis"), Diffuse("diffuse"), KazenStandardSurface("kazenstandard"), AreaLight("area"),
//   ConstantTexture("constanttexture"), BackgroundTexture("background"), GaussianFilter("gaussian"),
//   MitchellNetravaliFilter("mitchell"), TentFilter("tent"), BoxFilter("box")
//   renderer::render(Scene*, ...)                              include/kazen/renderer.h:10, src/kazen/renderer.cpp:72-153
// The objects are DESCRIPTIONS: Scene::activate() flattens them into a KzSceneDesc (enum tags instead of vtables)
// and renderer::render() hands that to the library, which does what renderer.cpp:85-133 did with TBB + Embree.
// Error behaviour follows the reference: kazen::Exception (std::runtime_error) from createInstance/addChild/activate
// (scene.cpp:33-35, parser.cpp:295-298); a plugin name that exists in the reference but is outside the hot path throws
// "... is not on the MI355X hot path" — never a silent fallback.
#pragma once
#include "../../include/kazen_mi355x.h"

#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <sstream>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

namespace kazen {

class Exception : public std::runtime_error {       // include/kazen/common.h:124-129
public:
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
};

struct Color3f { float r = 0, g = 0, b = 0; Color3f() {} Color3f(float v) : r(v), g(v), b(v) {} Color3f(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {} };
struct Transform {                                   // include/kazen/transform.h:16-78 (row-major here)
    std::array<float, 16> m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    /// <lookat origin target up> as parser.cpp:268-287 builds it: columns = left, newUp, dir, origin
    static Transform lookAt(const std::array<float, 3> &o, const std::array<float, 3> &t, const std::array<float, 3> &up) {
        auto sub = [](auto a, auto b) { return std::array<float, 3>{a[0] - b[0], a[1] - b[1], a[2] - b[2]}; };
        auto cross = [](auto a, auto b) { return std::array<float, 3>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; };
        auto norm = [](auto a) { float l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); return std::array<float, 3>{a[0] / l, a[1] / l, a[2] / l}; };
        auto d = norm(sub(t, o)); auto left = norm(cross(up, d)); auto nu = cross(d, left);
        Transform r;
        r.m = {left[0], nu[0], d[0], o[0], left[1], nu[1], d[1], o[1], left[2], nu[2], d[2], o[2], 0, 0, 0, 1};
        return r;
    }
    /// this * rhs (parser.cpp:243-290 left-multiplies every transform operation onto the running transform)
    Transform operator*(const Transform &b) const {
        Transform r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = 0.f; for (int k = 0; k < 4; ++k) s += m[4 * i + k] * b.m[4 * k + j]; r.m[4 * i + j] = s; }
        return r;
    }
    /// Transform * Point3f: homogeneous multiply and divide (transform.h:59-62)
    std::array<float, 3> point(const std::array<float, 3> &p) const {
        float q[4];
        for (int i = 0; i < 4; ++i) q[i] = m[4 * i] * p[0] + m[4 * i + 1] * p[1] + m[4 * i + 2] * p[2] + m[4 * i + 3];
        return {q[0] / q[3], q[1] / q[3], q[2] / q[3]};
    }
    /// Transform * Normal3f: inverse transpose of the upper 3x3 (transform.h:54-56); the 4x4 inverse is formed in double
    std::array<float, 3> normal(const std::array<float, 3> &n) const {
        double a[16], inv[16]; for (int i = 0; i < 16; ++i) a[i] = m[i];
        inv[0] = a[5]*a[10]*a[15]-a[5]*a[11]*a[14]-a[9]*a[6]*a[15]+a[9]*a[7]*a[14]+a[13]*a[6]*a[11]-a[13]*a[7]*a[10];
        inv[4] = -a[4]*a[10]*a[15]+a[4]*a[11]*a[14]+a[8]*a[6]*a[15]-a[8]*a[7]*a[14]-a[12]*a[6]*a[11]+a[12]*a[7]*a[10];
        inv[8] = a[4]*a[9]*a[15]-a[4]*a[11]*a[13]-a[8]*a[5]*a[15]+a[8]*a[7]*a[13]+a[12]*a[5]*a[11]-a[12]*a[7]*a[9];
        inv[12] = -a[4]*a[9]*a[14]+a[4]*a[10]*a[13]+a[8]*a[5]*a[14]-a[8]*a[6]*a[13]-a[12]*a[5]*a[10]+a[12]*a[6]*a[9];
        inv[1] = -a[1]*a[10]*a[15]+a[1]*a[11]*a[14]+a[9]*a[2]*a[15]-a[9]*a[3]*a[14]-a[13]*a[2]*a[11]+a[13]*a[3]*a[10];
        inv[5] = a[0]*a[10]*a[15]-a[0]*a[11]*a[14]-a[8]*a[2]*a[15]+a[8]*a[3]*a[14]+a[12]*a[2]*a[11]-a[12]*a[3]*a[10];
        inv[9] = -a[0]*a[9]*a[15]+a[0]*a[11]*a[13]+a[8]*a[1]*a[15]-a[8]*a[3]*a[13]-a[12]*a[1]*a[11]+a[12]*a[3]*a[9];
        inv[2] = a[1]*a[6]*a[15]-a[1]*a[7]*a[14]-a[5]*a[2]*a[15]+a[5]*a[3]*a[14]+a[13]*a[2]*a[7]-a[13]*a[3]*a[6];
        inv[6] = -a[0]*a[6]*a[15]+a[0]*a[7]*a[14]+a[4]*a[2]*a[15]-a[4]*a[3]*a[14]-a[12]*a[2]*a[7]+a[12]*a[3]*a[6];
        inv[10] = a[0]*a[5]*a[15]-a[0]*a[7]*a[13]-a[4]*a[1]*a[15]+a[4]*a[3]*a[13]+a[12]*a[1]*a[7]-a[12]*a[3]*a[5];
        const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
        if (det == 0.0) return n;
        // (M^-1)^T n: row i of the transposed inverse = column i of the inverse (rows/cols 0..2)
        const double r0 = (inv[0] * n[0] + inv[4] * n[1] + inv[8] * n[2]) / det, r1 = (inv[1] * n[0] + inv[5] * n[1] + inv[9] * n[2]) / det,
                     r2 = (inv[2] * n[0] + inv[6] * n[1] + inv[10] * n[2]) / det;
        return {(float)r0, (float)r1, (float)r2};
    }
};
using Vector3 = std::array<float, 3>;
/// File resolver (filesystem/resolver.h as main.cpp uses it: the scene file's directory is prepended before parsing)
inline std::vector<std::string> &fileResolverPaths() { static std::vector<std::string> p; return p; }
inline std::string resolveFile(const std::string &name) {
    if (name.empty() || name[0] == '/') return name;
    for (const std::string &dir : fileResolverPaths()) { const std::string c = dir + "/" + name; if (std::ifstream(c).good()) return c; }
    return name;
}

/// Typed property bag with defaults (include/kazen/proplist.h, src/kazen/proplist.cpp:5-32)
class PropertyList {
public:
    using Value = std::variant<bool, int, float, std::string, Color3f, Transform, Vector3>;
    void setBoolean(const std::string &n, bool v) { m_[n] = v; }
    void setInteger(const std::string &n, int v) { m_[n] = v; }
    void setFloat(const std::string &n, float v) { m_[n] = v; }
    void setString(const std::string &n, const std::string &v) { m_[n] = v; }
    void setColor(const std::string &n, const Color3f &v) { m_[n] = v; }
    void setTransform(const std::string &n, const Transform &v) { m_[n] = v; }
    void setPoint(const std::string &n, const Vector3 &v) { m_[n] = v; }
    void setVector(const std::string &n, const Vector3 &v) { m_[n] = v; }
    Vector3 getPoint(const std::string &n, const Vector3 &d) const { return get<Vector3>(n, d); }
    Vector3 getVector(const std::string &n, const Vector3 &d) const { return get<Vector3>(n, d); }
    bool has(const std::string &n) const { return m_.count(n) != 0; }
    bool getBoolean(const std::string &n, bool d) const { return get<bool>(n, d); }
    int getInteger(const std::string &n, int d) const { return get<int>(n, d); }
    float getFloat(const std::string &n, float d) const { return get<float>(n, d); }
    std::string getString(const std::string &n, const std::string &d) const { return get<std::string>(n, d); }
    Color3f getColor(const std::string &n, const Color3f &d) const { return get<Color3f>(n, d); }
    Transform getTransform(const std::string &n, const Transform &d) const { return get<Transform>(n, d); }
private:
    template <class T> T get(const std::string &n, const T &d) const {
        auto it = m_.find(n);
        if (it == m_.end()) return d;
        if (!std::holds_alternative<T>(it->second)) throw Exception("Property '" + n + "' has the wrong type!");
        return std::get<T>(it->second);
    }
    std::map<std::string, Value> m_;
};

class Object {                                       // include/kazen/object.h:17-100
public:
    enum EClassType { EScene = 0, EMesh, EBSDF, ELight, EMedium, ECamera, EIntegrator, ESampler, EReconstructionFilter, ETexture, EClassTypeCount };
    virtual ~Object() {}
    virtual EClassType getClassType() const = 0;
    virtual void addChild(Object *) { throw Exception("Object::addChild() is not implemented for objects of type '" + classTypeName(getClassType()) + "'!"); }
    virtual void setParent(Object *) {}
    virtual void activate() {}
    virtual std::string toString() const = 0;
    void setId(const std::string &id) { m_id = id; }
    const std::string &getId() const { return m_id; }
    static std::string classTypeName(EClassType t) {
        static const char *n[] = {"scene", "mesh", "bsdf", "light", "medium", "camera", "integrator", "sampler", "rfilter", "texture"};
        return t < EClassTypeCount ? n[t] : "<unknown>";
    }
protected:
    std::string m_id;
};

class ObjectFactory {                                // include/kazen/object.h:108-141
public:
    using Constructor = std::function<Object *(const PropertyList &)>;
    static void registerClass(const std::string &name, const Constructor &c) { table()[name] = c; }
    static Object *createInstance(const std::string &name, const PropertyList &props) {
        auto &t = table();
        auto it = t.find(name);
        if (it != t.end()) return it->second(props);
        static const char *offPath[] = {"normals", "ao", "whitted", "path_mats", "nonscatter"};
        for (const char *o : offPath)
            if (name == o) throw Exception("Class \"" + name + "\" exists in nano-kazen but is not on the MI355X hot path (path_mis; diffuse/lambertian/kazenstandard/mirror/dielectric/ggx/roughconductor/roughplastic/roughdielectric/normalmap; constanttexture/imagetexture/colorramp/blend; independent/pmj02bn/stratified/correlated; perspective/thinlens)");
        throw Exception("A constructor for class \"" + name + "\" could not be found!");
    }
private:
    static std::map<std::string, Constructor> &table() { static std::map<std::string, Constructor> t; return t; }
};
#define KAZEN_MI355X_REGISTER(cls, name) \
    inline bool cls##_registered = (::kazen::ObjectFactory::registerClass(name, [](const ::kazen::PropertyList &p) -> ::kazen::Object * { return new cls(p); }), true)

// ---- textures (src/kazen/texture.cpp) -------------------------------------------------------------------------------
// Rows of one scene flattening: textures / images / the BSDF rows that normalmaps wrap (placed behind the per-mesh rows).
class Texture;
class BSDF;
struct RowBuilder {
    std::vector<KzTexture> textures; std::vector<KzImage> images;
    std::map<const Texture *, int> seen;
    int nestedBase = 0; std::vector<const BSDF *> nested;
    int tex(const Texture *t);                       // 1-based texture id (0 for a null pointer)
    int nestedRow(const BSDF *b) { nested.push_back(b); return nestedBase + (int)nested.size() - 1; }
};
class Texture : public Object {
public:
    EClassType getClassType() const override { return ETexture; }
    virtual KzTexture row(RowBuilder &rb) const = 0;
    virtual bool isConstant() const { return false; }
};
inline int RowBuilder::tex(const Texture *t) {
    if (!t) return 0;
    auto it = seen.find(t);
    if (it != seen.end()) return it->second;
    KzTexture k = t->row(*this);                     // children first
    textures.push_back(k);
    return seen[t] = (int)textures.size();
}
class ConstantTexture : public Texture {             // texture.cpp:10-32
public:
    explicit ConstantTexture(const PropertyList &p) { m_color = p.getColor("color", Color3f(0.5f)); }
    KzTexture row(RowBuilder &) const override { KzTexture k{}; k.type = KZ_TEX_CONSTANT; k.color[0] = m_color.r; k.color[1] = m_color.g; k.color[2] = m_color.b; k.child[0] = k.child[1] = k.child[2] = -1; return k; }
    bool isConstant() const override { return true; }
    std::string toString() const override { return "ConstantTexture[]"; }
    Color3f m_color;
};
/// "imagetexture" (texture.cpp:36-98). The reference decodes the file through OpenImageIO; this dependency-free mirror reads
/// binary PGM / PPM (P5 / P6, 8 or 16 bit) and PFM, and takes any other format as an already decoded raster (setRaster).
class ImageTexture : public Texture {
public:
    explicit ImageTexture(const PropertyList &p) {
        m_filename = p.getString("filename", ""); m_colorspace = p.getString("colorspace", "srgb"); m_scale = p.getFloat("scale", 1.0f);
        // not a property of the reference: which filter stands in for OpenImageIO's TextureSystem::texture (KzTexture.filter; "bilinear" is the declared default)
        const std::string f = p.getString("filter", "bilinear");
        if (f != "bilinear" && f != "bicubic") throw Exception("imagetexture: filter \"" + f + "\" (bilinear or bicubic)");
        m_filter = f == "bicubic" ? KZ_TEXFILTER_BICUBIC : KZ_TEXFILTER_BILINEAR;
        if (!m_filename.empty()) load(resolveFile(m_filename));          // texture.cpp:40: getFileResolver()->resolve(fileName)
    }
    void setRaster(int width, int height, int channels, int format, const void *pixels) {
        m_w = width; m_h = height; m_c = channels; m_fmt = format;
        size_t bytes = (size_t)width * height * channels * (format == KZ_PIXEL_F32 ? 4 : 1);
        m_px.assign((const uint8_t *)pixels, (const uint8_t *)pixels + bytes);
    }
    KzTexture row(RowBuilder &rb) const override {
        if (m_px.empty()) throw Exception("imagetexture \"" + m_filename + "\": no raster (decode the file in the host application and call setRaster)");
        KzImage im{}; im.pixels = m_px.data(); im.width = m_w; im.height = m_h; im.channels = m_c; im.format = m_fmt;
        rb.images.push_back(im);
        KzTexture k{}; k.type = KZ_TEX_IMAGE; k.image = (int32_t)rb.images.size() - 1; k.scale = m_scale; k.srgb = m_colorspace == "srgb" ? 1 : 0;
        k.filter = m_filter;
        k.child[0] = k.child[1] = k.child[2] = -1;
        return k;
    }
    std::string toString() const override { return "ImageTexture[]"; }
private:
    void load(const std::string &fn) {
        FILE *f = std::fopen(fn.c_str(), "rb");
        if (!f) throw Exception("imagetexture: cannot open \"" + fn + "\"");
        char magic[3] = {0, 0, 0};
        auto token = [&](std::string &out) { out.clear(); int ch; while ((ch = std::fgetc(f)) != EOF) { if (ch == '#') { while ((ch = std::fgetc(f)) != EOF && ch != '\n') {} continue; } if (std::isspace(ch)) { if (!out.empty()) break; continue; } out.push_back((char)ch); } };
        if (std::fread(magic, 1, 2, f) != 2) { std::fclose(f); throw Exception("imagetexture: \"" + fn + "\" is empty"); }
        const std::string m(magic);
        std::string a, b, c;
        if (m == "P5" || m == "P6") {
            token(a); token(b); token(c);
            const int w = std::atoi(a.c_str()), h = std::atoi(b.c_str()), maxv = std::atoi(c.c_str()), ch = m == "P6" ? 3 : 1;
            if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 65535) { std::fclose(f); throw Exception("imagetexture: bad PNM header in \"" + fn + "\""); }
            const size_t n = (size_t)w * h * ch;
            if (maxv < 256) { std::vector<uint8_t> px(n); if (std::fread(px.data(), 1, n, f) != n) { std::fclose(f); throw Exception("imagetexture: truncated \"" + fn + "\""); }
                if (maxv != 255) { std::vector<float> fl(n); for (size_t i = 0; i < n; ++i) fl[i] = (float)px[i] / (float)maxv; setRaster(w, h, ch, KZ_PIXEL_F32, fl.data()); } else setRaster(w, h, ch, KZ_PIXEL_U8, px.data()); }
            else { std::vector<uint8_t> px(2 * n); if (std::fread(px.data(), 1, 2 * n, f) != 2 * n) { std::fclose(f); throw Exception("imagetexture: truncated \"" + fn + "\""); }
                std::vector<float> fl(n); for (size_t i = 0; i < n; ++i) fl[i] = (float)((px[2 * i] << 8) | px[2 * i + 1]) / (float)maxv; setRaster(w, h, ch, KZ_PIXEL_F32, fl.data()); }
        } else if (m == "PF" || m == "Pf") {
            token(a); token(b); token(c);
            const int w = std::atoi(a.c_str()), h = std::atoi(b.c_str()), ch = m == "PF" ? 3 : 1; const double sc = std::atof(c.c_str());
            if (w <= 0 || h <= 0 || sc == 0.0) { std::fclose(f); throw Exception("imagetexture: bad PFM header in \"" + fn + "\""); }
            const size_t n = (size_t)w * h * ch; std::vector<float> fl(n), out(n);
            if (std::fread(fl.data(), 4, n, f) != n) { std::fclose(f); throw Exception("imagetexture: truncated \"" + fn + "\""); }
            if (sc > 0) for (size_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, &fl[i], 4); u = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24); std::memcpy(&fl[i], &u, 4); }   // big endian file
            for (int y = 0; y < h; ++y) std::memcpy(&out[(size_t)y * w * ch], &fl[(size_t)(h - 1 - y) * w * ch], (size_t)w * ch * 4);   // PFM rows are bottom-up
            setRaster(w, h, ch, KZ_PIXEL_F32, out.data());
        } else { std::fclose(f); m_px.clear(); return; }     // another container (PNG, EXR, ...): the raster must come through setRaster
        std::fclose(f);
    }
    std::string m_filename, m_colorspace; float m_scale; int m_filter = KZ_TEXFILTER_BILINEAR, m_w = 0, m_h = 0, m_c = 0, m_fmt = KZ_PIXEL_U8; std::vector<uint8_t> m_px;
};
class ColorRampTexture : public Texture {            // texture.cpp:149-195
public:
    explicit ColorRampTexture(const PropertyList &p) { m_min = p.getFloat("min", 0.0f); m_max = p.getFloat("max", 1.0f); }
    ~ColorRampTexture() override { delete m_nested; }
    void addChild(Object *o) override { if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture"); m_nested = static_cast<Texture *>(o); }
    KzTexture row(RowBuilder &rb) const override { KzTexture k{}; k.type = KZ_TEX_COLORRAMP; k.rampMin = m_min; k.rampMax = m_max; k.child[0] = rb.tex(m_nested) - 1; k.child[1] = k.child[2] = -1; return k; }
    std::string toString() const override { return "ColorRampTexture[]"; }
    float m_min, m_max; Texture *m_nested = nullptr;
};
class BlendTexture : public Texture {                // texture.cpp:199-270
public:
    explicit BlendTexture(const PropertyList &p) { m_blendmode = p.getString("blendmode", "mix"); }
    ~BlendTexture() override { delete m_mask; delete m_input1; delete m_input2; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture");
        auto set = [&](Texture *&slot, const char *what) { if (slot) throw Exception(std::string("There is already an ") + what + " defined!"); slot = static_cast<Texture *>(o); };
        if (o->getId() == "mask") set(m_mask, "mask");
        else if (o->getId() == "input1") set(m_input1, "input1");
        else if (o->getId() == "input2") set(m_input2, "input2");
        else throw Exception("The name of this texture does not match any field!");
    }
    KzTexture row(RowBuilder &rb) const override {
        KzTexture k{}; k.type = KZ_TEX_BLEND; k.blendMode = m_blendmode == "mix" ? KZ_BLEND_MIX : m_blendmode == "multiply" ? KZ_BLEND_MULTIPLY : KZ_BLEND_NONE;
        k.child[0] = rb.tex(m_mask) - 1; k.child[1] = rb.tex(m_input1) - 1; k.child[2] = rb.tex(m_input2) - 1;
        return k;
    }
    std::string toString() const override { return "BlendTexture[]"; }
    std::string m_blendmode; Texture *m_mask = nullptr, *m_input1 = nullptr, *m_input2 = nullptr;
};
class BackgroundTexture : public Object {
public:
    explicit BackgroundTexture(const PropertyList &p) { m_intensity = p.getFloat("intensity", 1.0f); }
    ~BackgroundTexture() override { delete m_nested; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture");
        delete m_nested;                       // texture.cpp:128-136: the last texture child is the nested one
        m_nested = static_cast<Texture *>(o);  // constanttexture -> colour, imagetexture -> environment lookup, colorramp / blend -> 0 (texture.h:13)
    }
    EClassType getClassType() const override { return ETexture; }
    std::string toString() const override { return "Background[]"; }
    float m_intensity; Texture *m_nested = nullptr;
};

// ---- BSDFs ----------------------------------------------------------------------------------------------------------
// A texture child that is a constanttexture is folded into the row (texture id 0); any other texture goes through the table.
class BSDF : public Object {
public:
    EClassType getClassType() const override { return EBSDF; }
    virtual KzBSDF row(RowBuilder &rb) const = 0;
protected:
    static void bind3(RowBuilder &rb, const Texture *t, float *dst, int32_t &id) {
        if (t->isConstant()) { const Color3f &c = static_cast<const ConstantTexture *>(t)->m_color; dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; id = 0; }
        else id = rb.tex(t);
    }
    static void bind1(RowBuilder &rb, const Texture *t, float &dst, int32_t &id) {       // .r() of the colour (bsdf.cpp:1227,1231)
        if (t->isConstant()) { dst = static_cast<const ConstantTexture *>(t)->m_color.r; id = 0; }
        else id = rb.tex(t);
    }
};
class Diffuse : public BSDF {                        // src/kazen/bsdf.cpp:20-92
public:
    explicit Diffuse(const PropertyList &p) { m_albedo = p.getColor("albedo", Color3f(0.5f)); }
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_DIFFUSE; b.albedo[0] = m_albedo.r; b.albedo[1] = m_albedo.g; b.albedo[2] = m_albedo.b; return b; }
    std::string toString() const override { return "Diffuse[]"; }
    Color3f m_albedo;
};
class Lambertian : public BSDF {                     // src/kazen/bsdf.cpp:202-276: the diffuse model, albedo through a texture child
public:
    explicit Lambertian(const PropertyList &) {}
    ~Lambertian() override { delete m_albedo; }
    void addChild(Object *o) override { if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than albedi maps"); m_albedo = static_cast<Texture *>(o); }
    void activate() override { if (!m_albedo) throw Exception("lambertian needs an albedo texture"); }
    KzBSDF row(RowBuilder &rb) const override { KzBSDF b{}; b.type = KZ_BSDF_DIFFUSE; bind3(rb, m_albedo, b.albedo, b.albedoTex); return b; }
    std::string toString() const override { return "Lambertian[]"; }
    Texture *m_albedo = nullptr;
};
class NormalMap : public BSDF {                      // src/kazen/bsdf.cpp:281-417
public:
    explicit NormalMap(const PropertyList &) {}
    ~NormalMap() override { delete m_normalMap; delete m_nested; }
    void addChild(Object *o) override {
        switch (o->getClassType()) {
        case ETexture: m_normalMap = static_cast<Texture *>(o); break;
        case EBSDF: m_nested = static_cast<BSDF *>(o); break;
        default: throw Exception("addChild is not supported other than normal maps and nested BSDF");
        }
    }
    void activate() override {
        if (!m_normalMap || !m_nested) throw Exception("normalmap needs a normal texture and a nested BSDF");
        if (dynamic_cast<NormalMap *>(m_nested)) throw Exception("a normalmap nested in a normalmap is not on the MI355X hot path");
        m_nested->activate();
    }
    KzBSDF row(RowBuilder &rb) const override { KzBSDF b{}; b.type = KZ_BSDF_NORMALMAP; b.normalTex = rb.tex(m_normalMap); b.nested = rb.nestedRow(m_nested); return b; }
    std::string toString() const override { return "NormalMap[]"; }
    Texture *m_normalMap = nullptr; BSDF *m_nested = nullptr;
};
class KazenStandardSurface : public BSDF {           // src/kazen/bsdf.cpp:1157-1418
public:
    explicit KazenStandardSurface(const PropertyList &p) {
        m_anisotropy = p.getFloat("anisotropy", 0.0f); m_specular = p.getFloat("specular", 0.5f); m_specularTint = p.getFloat("specularTint", 0.5f);
        m_clearcoat = p.getFloat("clearcoat", 0.0f); m_clearcoatRoughness = p.getFloat("clearcoatRoughness", 0.5f);
        m_sheen = p.getFloat("sheen", 0.0f); m_sheenTint = p.getFloat("sheenTint", 0.5f);
    }
    ~KazenStandardSurface() override { delete m_baseColor; delete m_roughness; delete m_metallic; }
    void addChild(Object *o) override {              // bsdf.cpp:1373-1395: textures by id
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than baseColor maps");
        auto *c = static_cast<Texture *>(o);
        auto set = [&](Texture *&slot, const char *what) { if (slot) throw Exception(std::string("There is already an ") + what + " defined!"); slot = c; };
        if (o->getId() == "baseColor") set(m_baseColor, "baseColor");
        else if (o->getId() == "metallic") set(m_metallic, "metallic");
        else if (o->getId() == "roughness") set(m_roughness, "roughness");
        else throw Exception("kazenstandard: texture id must be baseColor, metallic or roughness");
    }
    void activate() override { if (!m_baseColor || !m_roughness || !m_metallic) throw Exception("kazenstandard needs baseColor, roughness and metallic textures"); }
    KzBSDF row(RowBuilder &rb) const override {
        KzBSDF b{}; b.type = KZ_BSDF_KAZENSTANDARD;
        bind3(rb, m_baseColor, b.baseColor, b.albedoTex); bind1(rb, m_roughness, b.roughness, b.roughnessTex); bind1(rb, m_metallic, b.metallic, b.metallicTex);
        b.anisotropy = m_anisotropy; b.specular = m_specular; b.specularTint = m_specularTint; b.clearcoat = m_clearcoat;
        b.clearcoatRoughness = m_clearcoatRoughness; b.sheen = m_sheen; b.sheenTint = m_sheenTint;
        return b;
    }
    std::string toString() const override { return "KazenStandardSurface"; }
    Texture *m_baseColor = nullptr, *m_roughness = nullptr, *m_metallic = nullptr;
    float m_anisotropy, m_specular, m_specularTint, m_clearcoat, m_clearcoatRoughness, m_sheen, m_sheenTint;
};

class Mirror : public BSDF {                         // src/kazen/bsdf.cpp:161-196
public:
    explicit Mirror(const PropertyList &) {}
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_MIRROR; return b; }
    std::string toString() const override { return "Mirror[]"; }
};
class Dielectric : public BSDF {                     // src/kazen/bsdf.cpp:98-155
public:
    explicit Dielectric(const PropertyList &p) { m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); }
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_DIELECTRIC; b.intIOR = m_intIOR; b.extIOR = m_extIOR; return b; }
    std::string toString() const override { return "Dielectric[]"; }
    float m_intIOR, m_extIOR;
};

class GGX : public BSDF {                            // src/kazen/bsdf.cpp:629-689 (albedo: a texture child)
public:
    explicit GGX(const PropertyList &p) { m_roughness = p.getFloat("roughness", 0.5f); m_anisotropy = p.getFloat("anisotropy", 0.f); }
    ~GGX() override { delete m_albedo; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than albedi maps");
        m_albedo = static_cast<Texture *>(o);
    }
    void activate() override { if (!m_albedo) throw Exception("ggx needs an albedo texture"); }
    KzBSDF row(RowBuilder &rb) const override { KzBSDF b{}; b.type = KZ_BSDF_GGX; bind3(rb, m_albedo, b.albedo, b.albedoTex); b.alpha = m_roughness; b.anisotropy = m_anisotropy; return b; }
    std::string toString() const override { return "GGX[]"; }
    Texture *m_albedo = nullptr; float m_roughness, m_anisotropy;
};
class RoughConductor : public BSDF {                 // src/kazen/bsdf.cpp:692-811
public:
    explicit RoughConductor(const PropertyList &p) {
        m_alpha = p.getFloat("alpha", 0.1f);
        const std::string mat = p.getString("material", "Au");
        static const float T[3][6] = {{0.1431189557f, 0.3749570432f, 1.4424785571f, 3.9831604247f, 2.3857207478f, 1.6032152899f},
                                      {0.2004376970f, 0.9240334304f, 1.1022119527f, 3.9129485033f, 2.4528477015f, 2.1421879552f},
                                      {4.3696828663f, 2.9167024892f, 1.6547005413f, 5.2064337956f, 4.2313645277f, 3.7549467933f}};
        int i = mat == "Au" ? 0 : mat == "Cu" ? 1 : mat == "Cr" ? 2 : -1;
        if (i < 0) throw Exception("roughconductor: unknown material \"" + mat + "\" (the reference leaves eta/k uninitialised here)");
        for (int a = 0; a < 3; ++a) { m_eta[a] = T[i][a]; m_k[a] = T[i][3 + a]; }
    }
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_ROUGHCONDUCTOR; b.alpha = m_alpha; for (int a = 0; a < 3; ++a) { b.condEta[a] = m_eta[a]; b.condK[a] = m_k[a]; } return b; }
    std::string toString() const override { return "RoughConductor[]"; }
    float m_alpha, m_eta[3], m_k[3];
};
class RoughPlastic : public BSDF {                   // src/kazen/bsdf.cpp:814-943
public:
    explicit RoughPlastic(const PropertyList &p) { m_alpha = p.getFloat("alpha", 0.1f); m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); m_kd = p.getColor("kd", Color3f(0.5f)); }
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_ROUGHPLASTIC; b.alpha = m_alpha; b.intIOR = m_intIOR; b.extIOR = m_extIOR; b.albedo[0] = m_kd.r; b.albedo[1] = m_kd.g; b.albedo[2] = m_kd.b; return b; }
    std::string toString() const override { return "RoughPlastic[]"; }
    float m_alpha, m_intIOR, m_extIOR; Color3f m_kd;
};
class RoughDielectric : public BSDF {                // src/kazen/bsdf.cpp:947-1145
public:
    explicit RoughDielectric(const PropertyList &p) { m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); m_roughness = p.getFloat("roughness", 0.1f); }
    KzBSDF row(RowBuilder &) const override { KzBSDF b{}; b.type = KZ_BSDF_ROUGHDIELECTRIC; b.alpha = m_roughness; b.intIOR = m_intIOR; b.extIOR = m_extIOR; return b; }
    std::string toString() const override { return "RoughDielectric"; }
    float m_intIOR, m_extIOR, m_roughness;
};

// ---- light / filter / sampler / integrator / camera -------------------------------------------------------------------
class AreaLight : public Object {                    // src/kazen/light.cpp:7-70
public:
    explicit AreaLight(const PropertyList &p) { m_color = p.getColor("color", Color3f(1.f)); m_intensity = p.getFloat("intensity", 1.f); m_vis = p.getBoolean("lightPrimaryVisibility", false); }
    EClassType getClassType() const override { return ELight; }
    bool getPrimaryVisibility() const { return m_vis; }
    KzLight row() const { KzLight l{}; l.color[0] = m_color.r; l.color[1] = m_color.g; l.color[2] = m_color.b; l.intensity = m_intensity; l.primaryVisibility = m_vis ? 1 : 0; return l; }
    std::string toString() const override { return "AreaLight[]"; }
    Color3f m_color; float m_intensity; bool m_vis;
};
class ReconstructionFilter : public Object {         // include/kazen/rfilter.h:22-37, src/kazen/rfilter.cpp
public:
    EClassType getClassType() const override { return EReconstructionFilter; }
    float getRadius() const { return m_f.radius; }
    KzFilter m_f{};
};
class GaussianFilter : public ReconstructionFilter { public: explicit GaussianFilter(const PropertyList &p) { m_f.type = KZ_FILTER_GAUSSIAN; m_f.radius = p.getFloat("radius", 2.0f); m_f.stddev = p.getFloat("stddev", 0.5f); } std::string toString() const override { return "GaussianFilter[]"; } };
class MitchellNetravaliFilter : public ReconstructionFilter { public: explicit MitchellNetravaliFilter(const PropertyList &p) { m_f.type = KZ_FILTER_MITCHELL; m_f.radius = p.getFloat("radius", 2.0f); m_f.B = p.getFloat("B", 1.0f / 3.0f); m_f.C = p.getFloat("C", 1.0f / 3.0f); } std::string toString() const override { return "MitchellNetravaliFilter[]"; } };
class TentFilter : public ReconstructionFilter { public: explicit TentFilter(const PropertyList &) { m_f.type = KZ_FILTER_TENT; m_f.radius = 1.0f; } std::string toString() const override { return "TentFilter[]"; } };
class BoxFilter : public ReconstructionFilter { public: explicit BoxFilter(const PropertyList &) { m_f.type = KZ_FILTER_BOX; m_f.radius = 0.5f; } std::string toString() const override { return "BoxFilter[]"; } };

class Sampler : public Object {                      // include/kazen/sampler.h:44-107
public:
    EClassType getClassType() const override { return ESampler; }
    uint32_t getSampleCount() const { return m_s.sampleCount; }
    KzSampler m_s{};
};
class Independent : public Sampler {                 // src/kazen/sampler.cpp:18-71 (seed: the reference leaves it uninitialised; 0 here, H2)
public:
    explicit Independent(const PropertyList &p) { m_s.type = KZ_SAMPLER_INDEPENDENT; m_s.sampleCount = (uint32_t)p.getInteger("sampleCount", 1); m_s.seed = (uint64_t)p.getInteger("seed", 0); }
    std::string toString() const override { return "Independent[sampleCount=" + std::to_string(m_s.sampleCount) + "]"; }
};
class PMJ02BN : public Sampler {                     // src/kazen/sampler.cpp:273-390; tables = the arrays of pmj02table.cpp / bluenoise.cpp
public:
    explicit PMJ02BN(const PropertyList &p) { m_s.type = KZ_SAMPLER_PMJ02BN; m_s.seed = (uint64_t)p.getInteger("seed", 1); m_s.sampleCount = (uint32_t)p.getInteger("sampleCount", 16); }
    void setTables(const uint32_t *pmj02bnSamples, const uint16_t *blueNoiseTextures) { m_s.pmj02bnSamples = pmj02bnSamples; m_s.blueNoise = blueNoiseTextures; }
    std::string toString() const override { return "PMJ02BN"; }
};
class Stratified : public Sampler {                  // src/kazen/sampler.cpp:81-156 (the library applies the constructor's rounding)
public:
    explicit Stratified(const PropertyList &p) { m_s.type = KZ_SAMPLER_STRATIFIED; m_s.seed = (uint64_t)p.getInteger("seed", 1); m_s.sampleCount = (uint32_t)p.getInteger("sampleCount", 16); m_s.resolution = p.getInteger("resolution", 4); }
    std::string toString() const override { return "Stratified"; }
};
class Correlated : public Sampler {                  // src/kazen/sampler.cpp:176-269
public:
    explicit Correlated(const PropertyList &p) { m_s.type = KZ_SAMPLER_CORRELATED; m_s.seed = (uint64_t)p.getInteger("seed", 1); m_s.sampleCount = (uint32_t)p.getInteger("sampleCount", 16); m_s.resolution = 4; }
    std::string toString() const override { return "Correlated"; }
};
class Integrator : public Object { public: EClassType getClassType() const override { return EIntegrator; } virtual void preprocess(const class Scene *) {} KzIntegrator m_i{}; };
class PathMisIntegrator : public Integrator {        // src/kazen/integrator.cpp:185-355
public:
    explicit PathMisIntegrator(const PropertyList &p) {
        m_i.type = KZ_INTEGRATOR_PATH_MIS; m_i.maxDepth = std::min(512, p.getInteger("maxDepth", 5)); m_i.traceBias = p.getFloat("traceBias", 0.001f);
        m_i.regularization = p.getBoolean("regularization", false) ? 1 : 0; m_i.accumulatedRoughness = p.getFloat("accumulatedRoughness", 0.5f);
    }
    std::string toString() const override { return "PathMisIntegrator[]"; }
};
class Camera : public Object { public: EClassType getClassType() const override { return ECamera; } KzCamera m_c{}; ReconstructionFilter *m_rfilter = nullptr; ~Camera() override { delete m_rfilter; } };
class PerspectiveCamera : public Camera {            // src/kazen/camera.cpp:14-131
public:
    explicit PerspectiveCamera(const PropertyList &p) {
        m_c.type = KZ_CAMERA_PERSPECTIVE; m_c.width = p.getInteger("width", 1280); m_c.height = p.getInteger("height", 720);
        Transform t = p.getTransform("toWorld", Transform());
        for (int i = 0; i < 16; ++i) m_c.toWorld[i] = t.m[i];
        m_c.fov = p.getFloat("fov", 30.0f); m_c.nearClip = p.getFloat("nearClip", 1e-4f); m_c.farClip = p.getFloat("farClip", 1e4f);
    }
    void addChild(Object *o) override {
        if (o->getClassType() != EReconstructionFilter) throw Exception("Camera::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        if (m_rfilter) throw Exception("Camera: tried to register multiple reconstruction filters!");
        m_rfilter = static_cast<ReconstructionFilter *>(o);
    }
    void activate() override { if (!m_rfilter) m_rfilter = static_cast<ReconstructionFilter *>(ObjectFactory::createInstance("gaussian", PropertyList())); m_c.rfilter = m_rfilter->m_f; }
    std::string toString() const override { return "PerspectiveCamera[]"; }
};

class ThinlensCamera : public PerspectiveCamera {    // src/kazen/camera.cpp:133-270
public:
    explicit ThinlensCamera(const PropertyList &p) : PerspectiveCamera(p) { m_c.type = KZ_CAMERA_THINLENS; m_c.apertureRadius = p.getFloat("apertureRadius", 1.0f); m_c.focusDistance = p.getFloat("focusDistance", 0.0f); }
    std::string toString() const override { return "ThinlensCamera[]"; }
};

// ---- mesh: buffers in the layout of kazen::Mesh (mesh.h:176-179); the OBJ loader itself is host scene I/O, out of scope ----
class Mesh : public Object {
public:
    Mesh() {}
    /// "obj" (WavefrontOBJ, mesh.cpp:200-343): with a "filename" property the file is loaded exactly as the reference does —
    /// v transformed by toWorld, vn by its inverse transpose and normalised, (p, uv, n) triples de-duplicated in encounter
    /// order, quads split into (0,1,2) and (3,0,2); without one the buffers are handed over through setBuffers.
    explicit Mesh(const PropertyList &props) {
        if (!props.has("filename")) return;
        const std::string filename = resolveFile(props.getString("filename", ""));
        std::ifstream is(filename);
        if (is.fail()) throw Exception("Unable to open OBJ file \"" + filename + "\"!");
        const Transform trafo = props.getTransform("toWorld", Transform());
        struct Key { uint32_t p = (uint32_t)-1, n = (uint32_t)-1, uv = (uint32_t)-1; bool operator==(const Key &o) const { return p == o.p && n == o.n && uv == o.uv; } };
        struct KeyHash { size_t operator()(const Key &k) const { size_t h = std::hash<uint32_t>()(k.p); h = h * 37 + std::hash<uint32_t>()(k.uv); h = h * 37 + std::hash<uint32_t>()(k.n); return h; } };
        auto toUInt = [](const std::string &t) { char *e = nullptr; unsigned long v = std::strtoul(t.c_str(), &e, 10); if (*e != '\0') throw Exception("Could not parse integer value \"" + t + "\""); return (uint32_t)v; };
        auto parseKey = [&](const std::string &str) {
            std::vector<std::string> tok; size_t last = 0;
            for (;;) { size_t pos = str.find('/', last); tok.push_back(str.substr(last, pos == std::string::npos ? pos : pos - last)); if (pos == std::string::npos) break; last = pos + 1; }
            if (tok.size() < 1 || tok.size() > 3) throw Exception("Invalid vertex data: \"" + str + "\"");
            Key k; k.p = toUInt(tok[0]);
            if (tok.size() >= 2 && !tok[1].empty()) k.uv = toUInt(tok[1]);
            if (tok.size() >= 3 && !tok[2].empty()) k.n = toUInt(tok[2]);
            return k;
        };
        std::vector<Vector3> positions, normals; std::vector<std::array<float, 2>> texcoords;
        std::vector<Key> vertices; std::unordered_map<Key, uint32_t, KeyHash> vertexMap;
        std::string lineStr;
        while (std::getline(is, lineStr)) {
            std::istringstream line(lineStr);
            std::string prefix; line >> prefix;
            if (prefix == "v") { Vector3 p{0, 0, 0}; line >> p[0] >> p[1] >> p[2]; positions.push_back(trafo.point(p)); }
            else if (prefix == "vt") { std::array<float, 2> tc{0, 0}; line >> tc[0] >> tc[1]; texcoords.push_back(tc); }
            else if (prefix == "vn") {
                Vector3 n{0, 0, 0}; line >> n[0] >> n[1] >> n[2]; n = trafo.normal(n);
                const float l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (l2 > 0.f) { const float l = std::sqrt(l2); n = {n[0] / l, n[1] / l, n[2] / l}; }
                normals.push_back(n);
            } else if (prefix == "f") {
                std::string v[4]; line >> v[0] >> v[1] >> v[2] >> v[3];
                Key verts[6]; int nVertices = 3;
                for (int i = 0; i < 3; ++i) verts[i] = parseKey(v[i]);
                if (!v[3].empty()) { verts[3] = parseKey(v[3]); verts[4] = verts[0]; verts[5] = verts[2]; nVertices = 6; }
                for (int i = 0; i < nVertices; ++i) {
                    auto it = vertexMap.find(verts[i]);
                    if (it == vertexMap.end()) { vertexMap[verts[i]] = (uint32_t)vertices.size(); m_F.push_back((uint32_t)vertices.size()); vertices.push_back(verts[i]); }
                    else m_F.push_back(it->second);
                }
            }
        }
        for (const Key &k : vertices) { const Vector3 &p = positions.at(k.p - 1); m_V.insert(m_V.end(), p.begin(), p.end()); }
        if (!normals.empty()) for (const Key &k : vertices) { const Vector3 &n = normals.at(k.n - 1); m_N.insert(m_N.end(), n.begin(), n.end()); }
        if (!texcoords.empty()) for (const Key &k : vertices) { const auto &t = texcoords.at(k.uv - 1); m_UV.insert(m_UV.end(), t.begin(), t.end()); }
    }
    ~Mesh() override { delete m_bsdf; delete m_light; }
    void setBuffers(std::vector<float> V, std::vector<uint32_t> F, std::vector<float> N = {}, std::vector<float> UV = {}) { m_V = std::move(V); m_F = std::move(F); m_N = std::move(N); m_UV = std::move(UV); }
    void addChild(Object *o) override {              // mesh.cpp:135-165
        switch (o->getClassType()) {
        case EBSDF: if (m_bsdf) throw Exception("Mesh: tried to register multiple BSDF instances!"); m_bsdf = static_cast<BSDF *>(o); break;
        case ELight: if (m_light) throw Exception("Mesh: tried to register multiple light instances!"); m_light = static_cast<AreaLight *>(o); break;
        default: throw Exception("Mesh::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        }
    }
    bool isLight() const { return m_light != nullptr; }
    EClassType getClassType() const override { return EMesh; }
    std::string toString() const override { return "Mesh[]"; }
    std::vector<float> m_V, m_N, m_UV; std::vector<uint32_t> m_F; BSDF *m_bsdf = nullptr; AreaLight *m_light = nullptr;
};

// ---- scene ---------------------------------------------------------------------------------------------------------------
class Scene : public Object {                        // src/kazen/scene.cpp, include/kazen/scene.h
public:
    explicit Scene(const PropertyList & = PropertyList()) {}
    ~Scene() override { if (m_handle) kz_scene_destroy(m_handle); for (auto *m : m_meshes) delete m; delete m_sampler; delete m_camera; delete m_integrator; delete m_background; }
    void addChild(Object *o) override {              // scene.cpp:81-130
        switch (o->getClassType()) {
        case EMesh: m_meshes.push_back(static_cast<Mesh *>(o)); break;
        case ESampler: if (m_sampler) throw Exception("There can only be one sampler per scene!"); m_sampler = static_cast<Sampler *>(o); break;
        case ECamera: if (m_camera) throw Exception("There can only be one camera per scene!"); m_camera = static_cast<Camera *>(o); break;
        case EIntegrator: if (m_integrator) throw Exception("There can only be one integrator per scene!"); m_integrator = static_cast<Integrator *>(o); break;
        case ETexture: {
            auto *b = dynamic_cast<BackgroundTexture *>(o);
            if (!b) throw Exception("Scene::addChild(<texture>): only \"background\" is supported");
            m_background = b; break;
        }
        default: throw Exception("Scene::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        }
    }
    /// scene.cpp:29-52 + the flattening: builds the KzSceneDesc and the library scene (host BVH build included)
    void activate() override {
        if (!m_integrator) throw Exception("No integrator was specified!");
        if (!m_camera) throw Exception("No camera was specified!");
        if (!m_sampler) m_sampler = static_cast<Sampler *>(ObjectFactory::createInstance("independent", PropertyList()));
        m_bsdfRows.clear(); m_lightRows.clear(); m_meshRows.clear();
        m_rb = RowBuilder();
        for (Mesh *m : m_meshes) if (m->m_bsdf) m_rb.nestedBase++;                     // rows wrapped by normalmaps go behind the per-mesh rows
        for (Mesh *m : m_meshes) {
            KzMesh k{}; k.V = m->m_V.data(); k.F = m->m_F.data(); k.N = m->m_N.empty() ? nullptr : m->m_N.data(); k.UV = m->m_UV.empty() ? nullptr : m->m_UV.data();
            k.nV = (uint32_t)(m->m_V.size() / 3); k.nF = (uint32_t)(m->m_F.size() / 3);
            k.bsdf = -1; k.light = -1;
            if (m->m_bsdf) { m->m_bsdf->activate(); k.bsdf = (int32_t)m_bsdfRows.size(); m_bsdfRows.push_back(m->m_bsdf->row(m_rb)); }   // no bsdf: default diffuse (mesh.cpp:25-28)
            if (m->m_light) { k.light = (int32_t)m_lightRows.size(); m_lightRows.push_back(m->m_light->row()); }
            m_meshRows.push_back(k);
        }
        for (size_t i = 0; i < m_rb.nested.size(); ++i) m_bsdfRows.push_back(m_rb.nested[i]->row(m_rb));
        m_camera->activate();
        KzSceneDesc d{};
        d.abiVersion = KZ_ABI_VERSION;
        d.meshes = m_meshRows.data(); d.nMeshes = (uint32_t)m_meshRows.size();
        d.bsdfs = m_bsdfRows.data(); d.nBsdfs = (uint32_t)m_bsdfRows.size();
        d.lights = m_lightRows.data(); d.nLights = (uint32_t)m_lightRows.size();
        d.textures = m_rb.textures.data(); d.nTextures = (uint32_t)m_rb.textures.size();
        d.images = m_rb.images.data(); d.nImages = (uint32_t)m_rb.images.size();
        d.camera = m_camera->m_c; d.sampler = m_sampler->m_s; d.integrator = m_integrator->m_i;
        if (m_background && m_background->m_nested) {
            d.background.present = 1; d.background.intensity = m_background->m_intensity;
            if (auto *c = dynamic_cast<ConstantTexture *>(m_background->m_nested)) {
                d.background.color[0] = c->m_color.r; d.background.color[1] = c->m_color.g; d.background.color[2] = c->m_color.b;
            } else {
                d.background.texture = m_rb.tex(m_background->m_nested);                                // 1-based id of the nested texture's row
                d.textures = m_rb.textures.data(); d.nTextures = (uint32_t)m_rb.textures.size();
                d.images = m_rb.images.data(); d.nImages = (uint32_t)m_rb.images.size();
            }
        }
        m_desc = d;
        if (m_handle) { kz_scene_destroy(m_handle); m_handle = nullptr; }
        int rc = kz_scene_create(&m_desc, &m_handle);
        if (rc != KZ_OK) throw Exception(std::string("kz_scene_create: ") + kz_last_error());
    }
    const std::vector<Mesh *> &getMeshes() const { return m_meshes; }
    const Camera *getCamera() const { return m_camera; }
    const Sampler *getSampler() const { return m_sampler; }
    const Integrator *getIntegrator() const { return m_integrator; }
    size_t getNumLights() const { return m_lightRows.size(); }
    const KzSceneDesc &desc() const { return m_desc; }
    KzScene *handle() const { return m_handle; }
    EClassType getClassType() const override { return EScene; }
    std::string toString() const override { return "Scene[]"; }
private:
    std::vector<Mesh *> m_meshes; Sampler *m_sampler = nullptr; Camera *m_camera = nullptr; Integrator *m_integrator = nullptr; BackgroundTexture *m_background = nullptr;
    std::vector<KzMesh> m_meshRows; std::vector<KzBSDF> m_bsdfRows; std::vector<KzLight> m_lightRows; RowBuilder m_rb;
    KzSceneDesc m_desc{}; KzScene *m_handle = nullptr;
};

KAZEN_MI355X_REGISTER(Scene, "scene");
KAZEN_MI355X_REGISTER(Mesh, "obj");
KAZEN_MI355X_REGISTER(AreaLight, "area");
KAZEN_MI355X_REGISTER(Diffuse, "diffuse");
KAZEN_MI355X_REGISTER(KazenStandardSurface, "kazenstandard");
KAZEN_MI355X_REGISTER(Mirror, "mirror");
KAZEN_MI355X_REGISTER(Dielectric, "dielectric");
KAZEN_MI355X_REGISTER(GGX, "ggx");
KAZEN_MI355X_REGISTER(RoughConductor, "roughconductor");
KAZEN_MI355X_REGISTER(RoughPlastic, "roughplastic");
KAZEN_MI355X_REGISTER(RoughDielectric, "roughdielectric");
KAZEN_MI355X_REGISTER(Lambertian, "lambertian");
KAZEN_MI355X_REGISTER(NormalMap, "normalmap");
KAZEN_MI355X_REGISTER(ConstantTexture, "constanttexture");
KAZEN_MI355X_REGISTER(ImageTexture, "imagetexture");
KAZEN_MI355X_REGISTER(ColorRampTexture, "colorramp");
KAZEN_MI355X_REGISTER(BlendTexture, "blend");
KAZEN_MI355X_REGISTER(BackgroundTexture, "background");
KAZEN_MI355X_REGISTER(PerspectiveCamera, "perspective");
KAZEN_MI355X_REGISTER(GaussianFilter, "gaussian");
KAZEN_MI355X_REGISTER(MitchellNetravaliFilter, "mitchell");
KAZEN_MI355X_REGISTER(TentFilter, "tent");
KAZEN_MI355X_REGISTER(BoxFilter, "box");
KAZEN_MI355X_REGISTER(Independent, "independent");
KAZEN_MI355X_REGISTER(Stratified, "stratified");
KAZEN_MI355X_REGISTER(Correlated, "correlated");
KAZEN_MI355X_REGISTER(ThinlensCamera, "thinlens");
KAZEN_MI355X_REGISTER(PMJ02BN, "pmj02bn");
KAZEN_MI355X_REGISTER(PathMisIntegrator, "path_mis");

// ---- Bitmap (include/kazen/bitmap.h, src/kazen/bitmap.cpp:23-64): the renderer's output files ----------------------------
// The reference writes through OpenImageIO; these writers are self-contained: an 8-bit RGB PNG (stored deflate blocks) and the
// uncompressed scan-line form of OpenEXR with FLOAT channels B, G, R.
class Bitmap {
public:
    Bitmap(int width, int height) : m_w(width), m_h(height), m_rgb((size_t)width * height * 3, 0.f) {}
    Bitmap(int width, int height, std::vector<float> rgb) : m_w(width), m_h(height), m_rgb(std::move(rgb)) {}
    int cols() const { return m_w; } int rows() const { return m_h; }
    float *data() { return m_rgb.data(); } const float *data() const { return m_rgb.data(); }
    void setSRGB8(std::vector<uint8_t> px) { m_rgb8 = std::move(px); }             // the device-side tone map (kz_film_to_srgb8)
    /// bitmap.cpp:39-62: Color3f::toSRGB, clamp(255 v, 0, 255), truncate; ".png" is appended like the reference does
    void savePNG(const std::string &filename) const {
        std::vector<uint8_t> px = m_rgb8;
        if (px.empty()) {
            px.resize(m_rgb.size());
            for (size_t i = 0; i < m_rgb.size(); ++i) {
                const float v = m_rgb[i], t = v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * std::pow(v, 1.0f / 2.4f) - 0.055f, s = 255.f * t;
                px[i] = (uint8_t)(s < 0.f ? 0.f : (s > 255.f ? 255.f : s));
            }
        }
        std::vector<uint8_t> raw; raw.reserve((size_t)m_h * (3 * m_w + 1));
        for (int y = 0; y < m_h; ++y) { raw.push_back(0); raw.insert(raw.end(), px.begin() + (size_t)y * 3 * m_w, px.begin() + (size_t)(y + 1) * 3 * m_w); }
        std::vector<uint8_t> z = {0x78, 0x01};
        for (size_t o = 0; o < raw.size() || o == 0; o += 65535) {
            const size_t n = std::min<size_t>(65535, raw.size() - o);
            z.push_back(o + n >= raw.size() ? 1 : 0); z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
            z.insert(z.end(), raw.begin() + o, raw.begin() + o + n);
            if (raw.empty()) break;
        }
        uint32_t a = 1, b = 0; for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
        for (int s = 24; s >= 0; s -= 8) z.push_back((((b << 16) | a) >> s) & 0xff);
        std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
        auto be32 = [](std::vector<uint8_t> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((x >> s) & 0xff); };
        auto chunk = [&](const char *tag, const std::vector<uint8_t> &d) {
            be32(out, (uint32_t)d.size());
            std::vector<uint8_t> td(tag, tag + 4); td.insert(td.end(), d.begin(), d.end());
            uint32_t crc = 0xffffffffu;
            for (uint8_t c : td) { crc ^= c; for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u))); }
            out.insert(out.end(), td.begin(), td.end()); be32(out, ~crc);
        };
        std::vector<uint8_t> ihdr; be32(ihdr, (uint32_t)m_w); be32(ihdr, (uint32_t)m_h); for (uint8_t c : {8, 2, 0, 0, 0}) ihdr.push_back(c);
        chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
        writeFile(filename + ".png", out);
    }
    /// bitmap.cpp:23-37: three FLOAT channels
    void saveEXR(const std::string &filename) const {
        std::vector<uint8_t> o;
        auto raw = [&](const void *p, size_t n) { o.insert(o.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
        auto i32 = [&](int32_t v) { raw(&v, 4); };
        auto f32 = [&](float v) { raw(&v, 4); };
        auto str = [&](const char *s) { raw(s, std::strlen(s) + 1); };
        auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
        i32(20000630); i32(2);
        attr("channels", "chlist", 3 * 18 + 1);
        for (const char *c : {"B", "G", "R"}) { str(c); i32(2); const uint8_t lin[4] = {0, 0, 0, 0}; raw(lin, 4); i32(1); i32(1); }
        o.push_back(0);
        attr("compression", "compression", 1); o.push_back(0);
        attr("dataWindow", "box2i", 16); i32(0); i32(0); i32(m_w - 1); i32(m_h - 1);
        attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(m_w - 1); i32(m_h - 1);
        attr("lineOrder", "lineOrder", 1); o.push_back(0);
        attr("pixelAspectRatio", "float", 4); f32(1.0f);
        attr("screenWindowCenter", "v2f", 8); f32(0.f); f32(0.f);
        attr("screenWindowWidth", "float", 4); f32(1.0f);
        o.push_back(0);
        const uint64_t line = 8 + 12 * (uint64_t)m_w, first = o.size() + 8 * (uint64_t)m_h;
        for (int y = 0; y < m_h; ++y) { const uint64_t off = first + (uint64_t)y * line; raw(&off, 8); }
        for (int y = 0; y < m_h; ++y) {
            i32(y); i32(12 * m_w);
            for (int c = 2; c >= 0; --c) for (int x = 0; x < m_w; ++x) f32(m_rgb[((size_t)y * m_w + x) * 3 + c]);
        }
        writeFile(filename + ".exr", o);
    }
private:
    static void writeFile(const std::string &path, const std::vector<uint8_t> &bytes) {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw Exception("Bitmap: cannot write \"" + path + "\"");
        const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
        std::fclose(f);
        if (n != bytes.size()) throw Exception("Bitmap: short write to \"" + path + "\"");
    }
    int m_w, m_h; std::vector<float> m_rgb; std::vector<uint8_t> m_rgb8;
};

namespace renderer {
/// The drop-in for kazen::renderer::render (renderer.cpp:72-153): render every sample of every pixel on `device` and
/// return the normalised bitmap (h x w x rgb, linear) — what result.toBitmap() holds before savePNG.
inline std::vector<float> render(Scene *scene, int device = 0) {
    KzScene *h = scene->handle();
    if (!h) throw Exception("renderer::render: scene was not activated");
    if (kz_scene_upload(h, device) != KZ_OK) throw Exception(std::string("kz_scene_upload: ") + kz_last_error());
    KzRenderOpts o{};
    o.device = device;
    if (kz_render(h, &o) != KZ_OK) throw Exception(std::string("kz_render: ") + kz_last_error());
    int32_t w, hh, b;
    kz_film_dims(h, &w, &hh, &b);
    std::vector<float> film((size_t)(w + 2 * b) * (hh + 2 * b) * 4), rgb((size_t)w * hh * 3);
    if (kz_film_download(h, film.data(), film.size()) != KZ_OK)      // (the scene is resident on that one device: its primary replica)
        throw Exception(std::string("kz_film_download: ") + kz_last_error());
    kz_film_to_rgb(film.data(), w, hh, b, rgb.data());
    return rgb;
}
/// The same over several GPUs of one node, the reference's own decomposition one level up (renderer.cpp:94-127 runs one
/// task per 32x32 block and merges with ImageBlock::put(ImageBlock&) under a mutex, block.cpp:87-96): ONE scene (one host
/// BVH) resident on every device of `devices`, one host thread per device rendering its share of 64x64 tiles
/// (kz_deal_tiles: by area), per-device films summed on the host in the order of `devices`. No collective.
/// An empty list means every visible device.
inline std::vector<float> render(Scene *scene, std::vector<int> devices, std::vector<float> *deviceMs = nullptr) {
    KzScene *h = scene->handle();
    if (!h) throw Exception("renderer::render: scene was not activated");
    if (devices.empty()) for (int d = 0; d < kz_device_count(); ++d) devices.push_back(d);
    if (devices.empty()) throw Exception("renderer::render: no HIP device visible (the MI355X path has no CPU fallback)");
    int32_t w, hh, b;
    kz_film_dims(h, &w, &hh, &b);
    std::vector<float> film((size_t)(w + 2 * b) * (hh + 2 * b) * 4), rgb((size_t)w * hh * 3), ms(devices.size());
    std::vector<int32_t> devs(devices.begin(), devices.end());
    if (kz_render_multi(h, nullptr, devs.data(), (uint32_t)devs.size(), 0, film.data(), film.size(), ms.data()) != KZ_OK)
        throw Exception(std::string("kz_render_multi: ") + kz_last_error());
    if (deviceMs) *deviceMs = ms;
    kz_film_to_rgb(film.data(), w, hh, b, rgb.data());
    return rgb;
}
inline void saveBitmap(Scene *scene, std::vector<float> rgb, const std::string &filename, bool deviceToneMap) {
    int32_t w, hh, b;
    kz_film_dims(scene->handle(), &w, &hh, &b);
    Bitmap bitmap(w, hh, std::move(rgb));
    if (deviceToneMap) {
        std::vector<uint8_t> px((size_t)w * hh * 3);
        if (kz_film_to_srgb8(scene->handle(), px.data(), px.size()) != KZ_OK) throw Exception(std::string("kz_film_to_srgb8: ") + kz_last_error());
        bitmap.setSRGB8(std::move(px));
    }
    const size_t lastdot = filename.find_last_of(".");
    bitmap.savePNG(lastdot == std::string::npos ? filename : filename.substr(0, lastdot));
}
/// renderer.cpp:72-153 including the file: renders and writes `<stem of filename>.png` (renderer.cpp:143-152), tone-mapped on the device
inline void render(Scene *scene, const std::string &filename, int device = 0) { saveBitmap(scene, render(scene, device), filename, true); }
/// multi-GPU form; the merged film lives on the host, so Bitmap::savePNG applies the reference's tone map there (bitmap.cpp:45-52)
inline void render(Scene *scene, const std::string &filename, const std::vector<int> &devices) { saveBitmap(scene, render(scene, devices), filename, false); }
} // namespace renderer
} // namespace kazen
