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
// The objects are DESCRIPTIONS, and they keep their parameters PRIVATE exactly as the reference's plugin classes do (every one of them
// is local to its .cpp there: bsdf.cpp:1157-1418, light.cpp:7-65, camera.cpp:14-131, sampler.cpp:18-390, integrator.cpp:185-355).
// What leaves a plugin leaves it through the ONE virtual INTEGRATION.md adds to its interface - BSDF::describe(KzBSDF&, mi355x::Rows&),
// Light::describe(KzLight&), Camera::describe(KzCamera&), ReconstructionFilter::describe(KzFilter&), Sampler::describe(KzSampler&),
// Integrator::describe(KzIntegrator&), Texture<T>::describe(KzTexture&, mi355x::Rows&) / describeBackground(KzBackground&, ...) - and
// Scene::getBackground(). The adapter a maintainer adds to a kazen tree (adapter/renderer_mi355x.cpp + adapter/kazen/mi355x.h, quoted
// verbatim in INTEGRATION.md) is compiled UNCHANGED against this mirror (mirror_tree/kazen/*.h forward the reference's header names here):
// tests/host_cpp builds it, so the adapter provably reads nothing the reference does not expose plus those virtuals.
// Error behaviour follows the reference: kazen::Exception (std::runtime_error) from createInstance/addChild/activate
// (scene.cpp:33-35, parser.cpp:295-298); a plugin name that exists in the reference but is outside the hot path throws
// "... is not on the MI355X hot path" — never a silent fallback.
#pragma once
#include "../../include/kazen_mi355x.h"
#if !defined(NAMESPACE_BEGIN)
#  define NAMESPACE_BEGIN(name) namespace name {          /* include/kazen/define.h:64-69 */
#endif
#if !defined(NAMESPACE_END)
#  define NAMESPACE_END(name) }
#endif

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

class Exception : public std::runtime_error {       // include/kazen/common.h:124-129 (fmt-style "{}" placeholders, as the reference's variadic constructor takes them)
public:
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
    template <typename... Args> Exception(const char *fmt, const Args &... args) : std::runtime_error(format(fmt, args...)) {}
private:
    static std::string format(const char *fmt) { return fmt; }
    template <typename A, typename... Rest> static std::string format(const char *fmt, const A &a, const Rest &... rest) {
        const char *br = std::strstr(fmt, "{}");
        if (!br) return fmt;
        std::ostringstream os; os << std::string(fmt, br) << a;
        return os.str() + format(br + 2, rest...);
    }
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
/// What the adapter reads of Eigen's Vector2i / MatrixXf / MatrixXu (include/kazen/vector.h, common.h:120-121): x() y(), data() size() cols()
struct Vector2i { int v[2] = {0, 0}; Vector2i() {} Vector2i(int x_, int y_) { v[0] = x_; v[1] = y_; } int x() const { return v[0]; } int y() const { return v[1]; } int &x() { return v[0]; } int &y() { return v[1]; } };
template <class S> struct ColMajorMatrix {                           // column-major r x n, contiguous: a mesh buffer of mesh.h:176-179
    std::vector<S> a; size_t r = 3;
    const S *data() const { return a.data(); }
    size_t size() const { return a.size(); }
    size_t rows() const { return r; }
    size_t cols() const { return r ? a.size() / r : 0; }
};
using MatrixXf = ColMajorMatrix<float>;
using MatrixXu = ColMajorMatrix<uint32_t>;
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

} // namespace kazen
// mi355x::Rows, mi355x::DeviceScene: the header INTEGRATION.md adds to a kazen tree, UNCHANGED. Its own includes are <kazen/common.h> (the
// reference's header name; mirror_tree/kazen/common.h forwards it to this file) and <kazen_mi355x.h>: compile with
//   -I include -I nano-kazen_amd/host/mirror_tree -I nano-kazen_amd/host/adapter
#include <kazen/mi355x.h>
namespace kazen {

// ---- textures (src/kazen/texture.cpp) -------------------------------------------------------------------------------
// include/kazen/texture.h:8-24 + the two virtuals INTEGRATION.md adds.
template <typename T> class Texture : public Object {
public:
    EClassType getClassType() const override { return ETexture; }
    /// ADDED: the texture's row for KzSceneDesc.textures; nested textures through rows.texture(child). false = not on the MI355X path
    virtual bool describe(KzTexture &, mi355x::Rows &) const { return false; }
    /// ADDED: only the "background" texture overrides this
    virtual bool describeBackground(KzBackground &, mi355x::Rows &) const { return false; }
};
class ConstantTexture : public Texture<Color3f> {    // texture.cpp:10-32
public:
    explicit ConstantTexture(const PropertyList &p) { m_color = p.getColor("color", Color3f(0.5f)); }
    bool describe(KzTexture &row, mi355x::Rows &) const override { row.type = KZ_TEX_CONSTANT; row.color[0] = m_color.r; row.color[1] = m_color.g; row.color[2] = m_color.b; return true; }
    std::string toString() const override { return "ConstantTexture[]"; }
private:
    Color3f m_color;
};
/// "imagetexture" (texture.cpp:36-98). The reference decodes the file through OpenImageIO; this dependency-free mirror reads
/// binary PGM / PPM (P5 / P6, 8 or 16 bit) and PFM, and takes any other format as an already decoded raster (setRaster).
class ImageTexture : public Texture<Color3f> {
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
    bool describe(KzTexture &row, mi355x::Rows &rows) const override {
        if (m_px.empty()) throw Exception("imagetexture \"" + m_filename + "\": no raster (decode the file in the host application and call setRaster)");
        row.type = KZ_TEX_IMAGE; row.image = rows.image(m_w, m_h, m_c, m_fmt, m_px.data()); row.scale = m_scale; row.srgb = m_colorspace == "srgb" ? 1 : 0; row.filter = m_filter;
        return true;
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
class ColorRampTexture : public Texture<Color3f> {   // texture.cpp:149-195
public:
    explicit ColorRampTexture(const PropertyList &p) { m_min = p.getFloat("min", 0.0f); m_max = p.getFloat("max", 1.0f); }
    ~ColorRampTexture() override { delete m_nested; }
    void addChild(Object *o) override { if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture"); m_nested = static_cast<Texture<Color3f> *>(o); }
    bool describe(KzTexture &row, mi355x::Rows &rows) const override { row.type = KZ_TEX_COLORRAMP; row.rampMin = m_min; row.rampMax = m_max; row.child[0] = rows.texture(m_nested) - 1; return true; }
    std::string toString() const override { return "ColorRampTexture[]"; }
private:
    float m_min, m_max; Texture<Color3f> *m_nested = nullptr;
};
class BlendTexture : public Texture<Color3f> {       // texture.cpp:199-270
public:
    explicit BlendTexture(const PropertyList &p) { m_blendmode = p.getString("blendmode", "mix"); }
    ~BlendTexture() override { delete m_mask; delete m_input1; delete m_input2; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture");
        auto set = [&](Texture<Color3f> *&slot, const char *what) { if (slot) throw Exception(std::string("There is already an ") + what + " defined!"); slot = static_cast<Texture<Color3f> *>(o); };
        if (o->getId() == "mask") set(m_mask, "mask");
        else if (o->getId() == "input1") set(m_input1, "input1");
        else if (o->getId() == "input2") set(m_input2, "input2");
        else throw Exception("The name of this texture does not match any field!");
    }
    bool describe(KzTexture &row, mi355x::Rows &rows) const override {
        row.type = KZ_TEX_BLEND; row.blendMode = m_blendmode == "mix" ? KZ_BLEND_MIX : m_blendmode == "multiply" ? KZ_BLEND_MULTIPLY : KZ_BLEND_NONE;
        row.child[0] = rows.texture(m_mask) - 1; row.child[1] = rows.texture(m_input1) - 1; row.child[2] = rows.texture(m_input2) - 1;
        return true;
    }
    std::string toString() const override { return "BlendTexture[]"; }
private:
    std::string m_blendmode; Texture<Color3f> *m_mask = nullptr, *m_input1 = nullptr, *m_input2 = nullptr;
};
class BackgroundTexture : public Texture<Color3f> {  // texture.cpp:104-145
public:
    explicit BackgroundTexture(const PropertyList &p) { m_intensity = p.getFloat("intensity", 1.0f); }
    ~BackgroundTexture() override { delete m_nested; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than nested Texture");
        delete m_nested;                       // texture.cpp:128-136: the last texture child is the nested one
        m_nested = static_cast<Texture<Color3f> *>(o);  // constanttexture -> colour, imagetexture -> environment lookup, colorramp / blend -> 0 (texture.h:13)
    }
    bool describeBackground(KzBackground &row, mi355x::Rows &rows) const override {
        if (!m_nested) return true;                                     // (Scene::getBackgroundColor then evaluates nothing: present stays 0)
        row.present = 1; row.intensity = m_intensity;
        rows.color(m_nested, row.color, row.texture);                   // a constanttexture folds into the colour, anything else is the nested texture's row
        return true;
    }
    std::string toString() const override { return "Background[]"; }
private:
    float m_intensity; Texture<Color3f> *m_nested = nullptr;
};

// ---- BSDFs ----------------------------------------------------------------------------------------------------------
// include/kazen/bsdf.h:80-125 + the virtual INTEGRATION.md adds. A texture child that is a constanttexture is folded into the row
// (texture id 0) by rows.color / rows.scalar; any other texture goes through the table.
class BSDF : public Object {
public:
    EClassType getClassType() const override { return EBSDF; }
    /// ADDED: this BSDF's row for KzSceneDesc.bsdfs; false = not on the MI355X path
    virtual bool describe(KzBSDF &, mi355x::Rows &) const { return false; }
};
class Diffuse : public BSDF {                        // src/kazen/bsdf.cpp:20-92
public:
    explicit Diffuse(const PropertyList &p) { m_albedo = p.getColor("albedo", Color3f(0.5f)); }
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_DIFFUSE; row.albedo[0] = m_albedo.r; row.albedo[1] = m_albedo.g; row.albedo[2] = m_albedo.b; return true; }
    std::string toString() const override { return "Diffuse[]"; }
private:
    Color3f m_albedo;
};
class Lambertian : public BSDF {                     // src/kazen/bsdf.cpp:202-276: the diffuse model, albedo through a texture child
public:
    explicit Lambertian(const PropertyList &) {}
    ~Lambertian() override { delete m_albedo; }
    void addChild(Object *o) override { if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than albedi maps"); m_albedo = static_cast<Texture<Color3f> *>(o); }
    void activate() override { if (!m_albedo) throw Exception("lambertian needs an albedo texture"); }
    bool describe(KzBSDF &row, mi355x::Rows &rows) const override { row.type = KZ_BSDF_DIFFUSE; rows.color(m_albedo, row.albedo, row.albedoTex); return true; }
    std::string toString() const override { return "Lambertian[]"; }
private:
    Texture<Color3f> *m_albedo = nullptr;
};
class NormalMap : public BSDF {                      // src/kazen/bsdf.cpp:281-417
public:
    explicit NormalMap(const PropertyList &) {}
    ~NormalMap() override { delete m_normalMap; delete m_nested; }
    void addChild(Object *o) override {
        switch (o->getClassType()) {
        case ETexture: m_normalMap = static_cast<Texture<Color3f> *>(o); break;
        case EBSDF: m_nested = static_cast<BSDF *>(o); break;
        default: throw Exception("addChild is not supported other than normal maps and nested BSDF");
        }
    }
    void activate() override {
        if (!m_normalMap || !m_nested) throw Exception("normalmap needs a normal texture and a nested BSDF");
        if (dynamic_cast<NormalMap *>(m_nested)) throw Exception("a normalmap nested in a normalmap is not on the MI355X hot path");
        m_nested->activate();
    }
    bool describe(KzBSDF &row, mi355x::Rows &rows) const override { row.type = KZ_BSDF_NORMALMAP; row.normalTex = rows.texture(m_normalMap); row.nested = rows.nested(m_nested); return true; }
    std::string toString() const override { return "NormalMap[]"; }
private:
    Texture<Color3f> *m_normalMap = nullptr; BSDF *m_nested = nullptr;
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
        auto *c = static_cast<Texture<Color3f> *>(o);
        auto set = [&](Texture<Color3f> *&slot, const char *what) { if (slot) throw Exception(std::string("There is already an ") + what + " defined!"); slot = c; };
        if (o->getId() == "baseColor") set(m_baseColor, "baseColor");
        else if (o->getId() == "metallic") set(m_metallic, "metallic");
        else if (o->getId() == "roughness") set(m_roughness, "roughness");
        else throw Exception("kazenstandard: texture id must be baseColor, metallic or roughness");
    }
    void activate() override { if (!m_baseColor || !m_roughness || !m_metallic) throw Exception("kazenstandard needs baseColor, roughness and metallic textures"); }
    bool describe(KzBSDF &row, mi355x::Rows &rows) const override {
        row.type = KZ_BSDF_KAZENSTANDARD;
        rows.color(m_baseColor, row.baseColor, row.albedoTex); rows.scalar(m_roughness, row.roughness, row.roughnessTex); rows.scalar(m_metallic, row.metallic, row.metallicTex);
        row.anisotropy = m_anisotropy; row.specular = m_specular; row.specularTint = m_specularTint; row.clearcoat = m_clearcoat;
        row.clearcoatRoughness = m_clearcoatRoughness; row.sheen = m_sheen; row.sheenTint = m_sheenTint;
        return true;
    }
    std::string toString() const override { return "KazenStandardSurface"; }
private:
    Texture<Color3f> *m_baseColor = nullptr, *m_roughness = nullptr, *m_metallic = nullptr;
    float m_anisotropy, m_specular, m_specularTint, m_clearcoat, m_clearcoatRoughness, m_sheen, m_sheenTint;
};

class Mirror : public BSDF {                         // src/kazen/bsdf.cpp:161-196
public:
    explicit Mirror(const PropertyList &) {}
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_MIRROR; return true; }
    std::string toString() const override { return "Mirror[]"; }
};
class Dielectric : public BSDF {                     // src/kazen/bsdf.cpp:98-155
public:
    explicit Dielectric(const PropertyList &p) { m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); }
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_DIELECTRIC; row.intIOR = m_intIOR; row.extIOR = m_extIOR; return true; }
    std::string toString() const override { return "Dielectric[]"; }
private:
    float m_intIOR, m_extIOR;
};

class GGX : public BSDF {                            // src/kazen/bsdf.cpp:629-689 (albedo: a texture child)
public:
    explicit GGX(const PropertyList &p) { m_roughness = p.getFloat("roughness", 0.5f); m_anisotropy = p.getFloat("anisotropy", 0.f); }
    ~GGX() override { delete m_albedo; }
    void addChild(Object *o) override {
        if (o->getClassType() != ETexture) throw Exception("addChild is not supported other than albedi maps");
        m_albedo = static_cast<Texture<Color3f> *>(o);
    }
    void activate() override { if (!m_albedo) throw Exception("ggx needs an albedo texture"); }
    bool describe(KzBSDF &row, mi355x::Rows &rows) const override { row.type = KZ_BSDF_GGX; rows.color(m_albedo, row.albedo, row.albedoTex); row.alpha = m_roughness; row.anisotropy = m_anisotropy; return true; }
    std::string toString() const override { return "GGX[]"; }
private:
    Texture<Color3f> *m_albedo = nullptr; float m_roughness, m_anisotropy;
};
// The three rough BSDFs keep what the reference's constructors keep: m_alpha = max(0.001, sqr("alpha")) (bsdf.cpp:696-700, 818-822, 956-959), not
// the property. Their rows say so with KzBSDF.alphaResolved = 1 (the library then takes `alpha` as it is).
inline float roughAlpha(float roughness) { const float MIN_ALPHA = 0.001f; return std::max(MIN_ALPHA, roughness * roughness); }
class RoughConductor : public BSDF {                 // src/kazen/bsdf.cpp:692-811
public:
    explicit RoughConductor(const PropertyList &p) {
        m_alpha = roughAlpha(p.getFloat("alpha", 0.1f));
        const std::string mat = p.getString("material", "Au");
        static const float T[3][6] = {{0.1431189557f, 0.3749570432f, 1.4424785571f, 3.9831604247f, 2.3857207478f, 1.6032152899f},
                                      {0.2004376970f, 0.9240334304f, 1.1022119527f, 3.9129485033f, 2.4528477015f, 2.1421879552f},
                                      {4.3696828663f, 2.9167024892f, 1.6547005413f, 5.2064337956f, 4.2313645277f, 3.7549467933f}};
        int i = mat == "Au" ? 0 : mat == "Cu" ? 1 : mat == "Cr" ? 2 : -1;
        if (i < 0) throw Exception("roughconductor: unknown material \"" + mat + "\" (the reference leaves eta/k uninitialised here)");
        for (int a = 0; a < 3; ++a) { m_eta[a] = T[i][a]; m_k[a] = T[i][3 + a]; }
    }
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_ROUGHCONDUCTOR; row.alpha = m_alpha; row.alphaResolved = 1; for (int a = 0; a < 3; ++a) { row.condEta[a] = m_eta[a]; row.condK[a] = m_k[a]; } return true; }
    std::string toString() const override { return "RoughConductor[]"; }
private:
    float m_alpha, m_eta[3], m_k[3];
};
class RoughPlastic : public BSDF {                   // src/kazen/bsdf.cpp:814-943
public:
    explicit RoughPlastic(const PropertyList &p) { m_alpha = roughAlpha(p.getFloat("alpha", 0.1f)); m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); m_kd = p.getColor("kd", Color3f(0.5f)); }
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_ROUGHPLASTIC; row.alpha = m_alpha; row.alphaResolved = 1; row.intIOR = m_intIOR; row.extIOR = m_extIOR; row.albedo[0] = m_kd.r; row.albedo[1] = m_kd.g; row.albedo[2] = m_kd.b; return true; }
    std::string toString() const override { return "RoughPlastic[]"; }
private:
    float m_alpha, m_intIOR, m_extIOR; Color3f m_kd;
};
class RoughDielectric : public BSDF {                // src/kazen/bsdf.cpp:947-1145
public:
    explicit RoughDielectric(const PropertyList &p) { m_intIOR = p.getFloat("intIOR", 1.5046f); m_extIOR = p.getFloat("extIOR", 1.000277f); m_alpha = roughAlpha(p.getFloat("roughness", 0.1f)); }
    bool describe(KzBSDF &row, mi355x::Rows &) const override { row.type = KZ_BSDF_ROUGHDIELECTRIC; row.alpha = m_alpha; row.alphaResolved = 1; row.intIOR = m_intIOR; row.extIOR = m_extIOR; return true; }
    std::string toString() const override { return "RoughDielectric"; }
private:
    float m_intIOR, m_extIOR, m_alpha;
};

// ---- light / filter / sampler / integrator / camera -------------------------------------------------------------------
class Light : public Object {                        // include/kazen/light.h:43-63 + the virtual INTEGRATION.md adds
public:
    EClassType getClassType() const override { return ELight; }
    virtual bool getPrimaryVisibility() const { return false; }
    /// ADDED: this light's row for KzSceneDesc.lights; false = not on the MI355X path
    virtual bool describe(KzLight &) const { return false; }
};
class AreaLight : public Light {                     // src/kazen/light.cpp:7-70
public:
    explicit AreaLight(const PropertyList &p) { m_color = p.getColor("color", Color3f(1.f)); m_intensity = p.getFloat("intensity", 1.f); m_lightPrimaryVisibility = p.getBoolean("lightPrimaryVisibility", false); }
    bool getPrimaryVisibility() const override { return m_lightPrimaryVisibility; }
    bool describe(KzLight &row) const override { row.color[0] = m_color.r; row.color[1] = m_color.g; row.color[2] = m_color.b; row.intensity = m_intensity; row.primaryVisibility = m_lightPrimaryVisibility ? 1 : 0; return true; }
    std::string toString() const override { return "AreaLight[]"; }
private:
    Color3f m_color; float m_intensity; bool m_lightPrimaryVisibility;
};
class ReconstructionFilter : public Object {         // include/kazen/rfilter.h:22-37, src/kazen/rfilter.cpp
public:
    EClassType getClassType() const override { return EReconstructionFilter; }
    float getRadius() const { return m_radius; }
    /// ADDED: the filter's parameters (the library tabulates it exactly as block.cpp:13-21 does); false = not on the MI355X path
    virtual bool describe(KzFilter &) const { return false; }
protected:
    float m_radius = 0.f;
};
class GaussianFilter : public ReconstructionFilter {           // rfilter.cpp:10-31
public:
    explicit GaussianFilter(const PropertyList &p) { m_radius = p.getFloat("radius", 2.0f); m_stddev = p.getFloat("stddev", 0.5f); }
    bool describe(KzFilter &row) const override { row.type = KZ_FILTER_GAUSSIAN; row.radius = m_radius; row.stddev = m_stddev; return true; }
    std::string toString() const override { return "GaussianFilter[]"; }
private:
    float m_stddev;
};
class MitchellNetravaliFilter : public ReconstructionFilter {  // rfilter.cpp:39-70
public:
    explicit MitchellNetravaliFilter(const PropertyList &p) { m_radius = p.getFloat("radius", 2.0f); m_B = p.getFloat("B", 1.0f / 3.0f); m_C = p.getFloat("C", 1.0f / 3.0f); }
    bool describe(KzFilter &row) const override { row.type = KZ_FILTER_MITCHELL; row.radius = m_radius; row.B = m_B; row.C = m_C; return true; }
    std::string toString() const override { return "MitchellNetravaliFilter[]"; }
private:
    float m_B, m_C;
};
class TentFilter : public ReconstructionFilter {               // rfilter.cpp:73-86
public:
    explicit TentFilter(const PropertyList &) { m_radius = 1.0f; }
    bool describe(KzFilter &row) const override { row.type = KZ_FILTER_TENT; row.radius = m_radius; return true; }
    std::string toString() const override { return "TentFilter[]"; }
};
class BoxFilter : public ReconstructionFilter {                // rfilter.cpp:89-102
public:
    explicit BoxFilter(const PropertyList &) { m_radius = 0.5f; }
    bool describe(KzFilter &row) const override { row.type = KZ_FILTER_BOX; row.radius = m_radius; return true; }
    std::string toString() const override { return "BoxFilter[]"; }
};

class Sampler : public Object {                      // include/kazen/sampler.h:44-107 + the virtual INTEGRATION.md adds
public:
    EClassType getClassType() const override { return ESampler; }
    virtual uint32_t getSampleCount() const { return m_sampleCount; }
    /// ADDED: type, sample count, seed (and tables) for KzSceneDesc.sampler; false = not on the MI355X path
    virtual bool describe(KzSampler &) const { return false; }
protected:
    uint64_t m_seed = 0; uint32_t m_sampleCount = 0;
};
class Independent : public Sampler {                 // src/kazen/sampler.cpp:18-71 (seed: the reference leaves it uninitialised; 0 here, H2)
public:
    explicit Independent(const PropertyList &p) { m_sampleCount = (uint32_t)p.getInteger("sampleCount", 1); m_seed = (uint64_t)p.getInteger("seed", 0); }
    bool describe(KzSampler &row) const override { row.type = KZ_SAMPLER_INDEPENDENT; row.sampleCount = m_sampleCount; row.seed = m_seed; return true; }
    std::string toString() const override { return "Independent[sampleCount=" + std::to_string(m_sampleCount) + "]"; }
};
class PMJ02BN : public Sampler {                     // src/kazen/sampler.cpp:273-390; tables = the arrays of pmj02table.cpp / bluenoise.cpp
public:
    explicit PMJ02BN(const PropertyList &p) { m_seed = (uint64_t)p.getInteger("seed", 1); m_sampleCount = (uint32_t)p.getInteger("sampleCount", 16); }
    /// (the reference links its tables - kazen::pmj02bnSamples, kazen::BlueNoiseTextures - and its override passes those; they are missing from the checkout, so the mirror is handed them)
    void setTables(const uint32_t *pmj02bnSamples, const uint16_t *blueNoiseTextures) { m_pmj = pmj02bnSamples; m_bn = blueNoiseTextures; }
    bool describe(KzSampler &row) const override { row.type = KZ_SAMPLER_PMJ02BN; row.sampleCount = m_sampleCount; row.seed = m_seed; row.pmj02bnSamples = m_pmj; row.blueNoise = m_bn; return true; }
    std::string toString() const override { return "PMJ02BN"; }
private:
    const uint32_t *m_pmj = nullptr; const uint16_t *m_bn = nullptr;
};
class Stratified : public Sampler {                  // src/kazen/sampler.cpp:81-156 (the constructor's rounding: :87-92)
public:
    explicit Stratified(const PropertyList &p) {
        m_seed = (uint64_t)p.getInteger("seed", 1); m_sampleCount = (uint32_t)p.getInteger("sampleCount", 16); m_resolution = p.getInteger("resolution", 4);
        while ((uint32_t)(m_resolution * m_resolution) < m_sampleCount) m_resolution++;
        m_sampleCount = (uint32_t)(m_resolution * m_resolution);
    }
    bool describe(KzSampler &row) const override { row.type = KZ_SAMPLER_STRATIFIED; row.sampleCount = m_sampleCount; row.resolution = m_resolution; row.seed = m_seed; return true; }
    std::string toString() const override { return "Stratified"; }
private:
    int m_resolution;
};
class Correlated : public Sampler {                  // src/kazen/sampler.cpp:176-269 (the constructor's rounding: :181-187)
public:
    explicit Correlated(const PropertyList &p) {
        m_seed = (uint64_t)p.getInteger("seed", 1); m_sampleCount = (uint32_t)p.getInteger("sampleCount", 16);
        m_resolution[1] = (int)std::sqrt((double)m_sampleCount); m_resolution[0] = (int)((m_sampleCount + m_resolution[1] - 1) / m_resolution[1]);
        m_sampleCount = (uint32_t)(m_resolution[0] * m_resolution[1]);
    }
    bool describe(KzSampler &row) const override { row.type = KZ_SAMPLER_CORRELATED; row.sampleCount = m_sampleCount; row.resolution = 4; row.seed = m_seed; return true; }
    std::string toString() const override { return "Correlated"; }
private:
    int m_resolution[2];
};
class Integrator : public Object {                   // include/kazen/integrator.h:14-43 + the virtual INTEGRATION.md adds
public:
    EClassType getClassType() const override { return EIntegrator; }
    virtual void preprocess(const class Scene *) {}
    /// ADDED: the integrator's parameters for KzSceneDesc.integrator; false = not on the MI355X path (only path_mis is)
    virtual bool describe(KzIntegrator &) const { return false; }
};
class PathMisIntegrator : public Integrator {        // src/kazen/integrator.cpp:185-355
public:
    explicit PathMisIntegrator(const PropertyList &p) {
        m_maxDepth = std::min(512, p.getInteger("maxDepth", 5)); m_rayEpsilon = p.getFloat("traceBias", 0.001f);
        m_regularization = p.getBoolean("regularization", false); m_accumulatedRoughness = p.getFloat("accumulatedRoughness", 0.5f);
    }
    bool describe(KzIntegrator &row) const override { row.type = KZ_INTEGRATOR_PATH_MIS; row.maxDepth = m_maxDepth; row.traceBias = m_rayEpsilon; row.regularization = m_regularization ? 1 : 0; row.accumulatedRoughness = m_accumulatedRoughness; return true; }
    std::string toString() const override { return "PathMisIntegrator[]"; }
private:
    int m_maxDepth; float m_rayEpsilon; bool m_regularization; float m_accumulatedRoughness;
};
class Camera : public Object {                       // include/kazen/camera.h:16-56 + the virtual INTEGRATION.md adds
public:
    ~Camera() override { delete m_rfilter; }
    EClassType getClassType() const override { return ECamera; }
    const Vector2i &getOutputSize() const { return m_outputSize; }
    const ReconstructionFilter *getReconstructionFilter() const { return m_rfilter; }
    /// ADDED: size, projection, clips, toWorld and the reconstruction filter for KzSceneDesc.camera; false = not on the MI355X path
    virtual bool describe(KzCamera &) const { return false; }
protected:
    Vector2i m_outputSize; ReconstructionFilter *m_rfilter = nullptr;
};
class PerspectiveCamera : public Camera {            // src/kazen/camera.cpp:14-131
public:
    explicit PerspectiveCamera(const PropertyList &p) {
        m_outputSize.x() = p.getInteger("width", 1280); m_outputSize.y() = p.getInteger("height", 720);
        m_cameraToWorld = p.getTransform("toWorld", Transform());
        m_fov = p.getFloat("fov", 30.0f); m_nearClip = p.getFloat("nearClip", 1e-4f); m_farClip = p.getFloat("farClip", 1e4f);
    }
    void addChild(Object *o) override {
        if (o->getClassType() != EReconstructionFilter) throw Exception("Camera::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        if (m_rfilter) throw Exception("Camera: tried to register multiple reconstruction filters!");
        m_rfilter = static_cast<ReconstructionFilter *>(o);
    }
    void activate() override { if (!m_rfilter) m_rfilter = static_cast<ReconstructionFilter *>(ObjectFactory::createInstance("gaussian", PropertyList())); }     // camera.cpp:64-67
    bool describe(KzCamera &row) const override {
        row.type = KZ_CAMERA_PERSPECTIVE; row.width = m_outputSize.x(); row.height = m_outputSize.y();
        for (int i = 0; i < 16; ++i) row.toWorld[i] = m_cameraToWorld.m[i];
        row.fov = m_fov; row.nearClip = m_nearClip; row.farClip = m_farClip;
        return m_rfilter && m_rfilter->describe(row.rfilter);
    }
    std::string toString() const override { return "PerspectiveCamera[]"; }
private:
    Transform m_cameraToWorld; float m_fov, m_nearClip, m_farClip;
};

class ThinlensCamera : public PerspectiveCamera {    // src/kazen/camera.cpp:133-270
public:
    explicit ThinlensCamera(const PropertyList &p) : PerspectiveCamera(p) { m_apertureRadius = p.getFloat("apertureRadius", 1.0f); m_focusDistance = p.getFloat("focusDistance", 0.0f); }
    bool describe(KzCamera &row) const override { if (!PerspectiveCamera::describe(row)) return false; row.type = KZ_CAMERA_THINLENS; row.apertureRadius = m_apertureRadius; row.focusDistance = m_focusDistance; return true; }
    std::string toString() const override { return "ThinlensCamera[]"; }
private:
    float m_apertureRadius, m_focusDistance;
};

// ---- mesh: buffers in the layout of kazen::Mesh (mesh.h:176-179); the OBJ loader itself is host scene I/O, out of scope ----
class Mesh : public Object {
public:
    Mesh() { m_UV.r = 2; }
    /// "obj" (WavefrontOBJ, mesh.cpp:200-343): with a "filename" property the file is loaded exactly as the reference does —
    /// v transformed by toWorld, vn by its inverse transpose and normalised, (p, uv, n) triples de-duplicated in encounter
    /// order, quads split into (0,1,2) and (3,0,2); without one the buffers are handed over through setBuffers.
    explicit Mesh(const PropertyList &props) {
        m_UV.r = 2;
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
                    if (it == vertexMap.end()) { vertexMap[verts[i]] = (uint32_t)vertices.size(); m_F.a.push_back((uint32_t)vertices.size()); vertices.push_back(verts[i]); }
                    else m_F.a.push_back(it->second);
                }
            }
        }
        for (const Key &k : vertices) { const Vector3 &p = positions.at(k.p - 1); m_V.a.insert(m_V.a.end(), p.begin(), p.end()); }
        if (!normals.empty()) for (const Key &k : vertices) { const Vector3 &n = normals.at(k.n - 1); m_N.a.insert(m_N.a.end(), n.begin(), n.end()); }
        if (!texcoords.empty()) for (const Key &k : vertices) { const auto &t = texcoords.at(k.uv - 1); m_UV.a.insert(m_UV.a.end(), t.begin(), t.end()); }
    }
    ~Mesh() override { delete m_bsdf; delete m_light; }
    void setBuffers(std::vector<float> V, std::vector<uint32_t> F, std::vector<float> N = {}, std::vector<float> UV = {}) { m_V.a = std::move(V); m_F.a = std::move(F); m_N.a = std::move(N); m_UV.a = std::move(UV); }
    void addChild(Object *o) override {              // mesh.cpp:135-165
        switch (o->getClassType()) {
        case EBSDF: if (m_bsdf) throw Exception("Mesh: tried to register multiple BSDF instances!"); m_bsdf = static_cast<BSDF *>(o); break;
        case ELight: if (m_light) throw Exception("Mesh: tried to register multiple light instances!"); m_light = static_cast<Light *>(o); break;
        default: throw Exception("Mesh::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        }
    }
    /// mesh.cpp:24-28: a mesh without a BSDF gets the default diffuse one
    void activate() override { if (!m_bsdf) m_bsdf = static_cast<BSDF *>(ObjectFactory::createInstance("diffuse", PropertyList())); m_bsdf->activate(); }
    // the accessors of include/kazen/mesh.h:66-130 the adapter reads
    uint32_t getTriangleCount() const { return (uint32_t)m_F.cols(); }
    uint32_t getVertexCount() const { return (uint32_t)m_V.cols(); }
    const MatrixXf &getVertexPositions() const { return m_V; }
    const MatrixXf &getVertexNormals() const { return m_N; }
    const MatrixXf &getVertexTexCoords() const { return m_UV; }
    const MatrixXu &getIndices() const { return m_F; }
    bool isLight() const { return m_light != nullptr; }
    const Light *getLight() const { return m_light; }
    const BSDF *getBSDF() const { return m_bsdf; }
    EClassType getClassType() const override { return EMesh; }
    std::string toString() const override { return "Mesh[]"; }
protected:
    MatrixXf m_V, m_N, m_UV; MatrixXu m_F; BSDF *m_bsdf = nullptr; Light *m_light = nullptr;
};

// ---- scene ---------------------------------------------------------------------------------------------------------------
class Scene : public Object {                        // src/kazen/scene.cpp, include/kazen/scene.h
public:
    explicit Scene(const PropertyList & = PropertyList()) {}
    ~Scene() override { for (auto *m : m_meshes) delete m; delete m_sampler; delete m_camera; delete m_integrator; delete m_background; }
    void addChild(Object *o) override {              // scene.cpp:81-130
        switch (o->getClassType()) {
        case EMesh: m_meshes.push_back(static_cast<Mesh *>(o)); break;
        case ESampler: if (m_sampler) throw Exception("There can only be one sampler per scene!"); m_sampler = static_cast<Sampler *>(o); break;
        case ECamera: if (m_camera) throw Exception("There can only be one camera per scene!"); m_camera = static_cast<Camera *>(o); break;
        case EIntegrator: if (m_integrator) throw Exception("There can only be one integrator per scene!"); m_integrator = static_cast<Integrator *>(o); break;
        case ETexture: {
            KzBackground probe{}; mi355x::Rows rows;
            auto *t = static_cast<Texture<Color3f> *>(o);
            if (!t->describeBackground(probe, rows)) throw Exception("Scene::addChild(<texture>): only \"background\" is supported");
            m_background = t; break;
        }
        default: throw Exception("Scene::addChild(<" + classTypeName(o->getClassType()) + ">) is not supported!");
        }
    }
    /// scene.cpp:29-52: checks, the default sampler, the list of emitters. (The reference builds its Embree scene here; the library's host
    /// BVH is built by mi355x::DeviceScene, the adapter's object, from what the describe() virtuals hand over.)
    void activate() override {
        if (!m_integrator) throw Exception("No integrator was specified!");
        if (!m_camera) throw Exception("No camera was specified!");
        if (!m_sampler) m_sampler = static_cast<Sampler *>(ObjectFactory::createInstance("independent", PropertyList()));
        m_camera->activate();
        m_lights.clear();
        for (Mesh *m : m_meshes) { m->activate(); if (m->isLight()) m_lights.push_back(m); }
    }
    // include/kazen/scene.h:24-60
    const std::vector<Mesh *> &getMeshes() const { return m_meshes; }
    const std::vector<Mesh *> &getLights() const { return m_lights; }
    const Camera *getCamera() const { return m_camera; }
    const Sampler *getSampler() const { return m_sampler; }
    const Integrator *getIntegrator() const { return m_integrator; }
    size_t getNumLights() const { return m_lights.size(); }
    /// ADDED: scene.h has no accessor for m_background (only getBackgroundColor(dir))
    const Texture<Color3f> *getBackground() const { return m_background; }
    EClassType getClassType() const override { return EScene; }
    std::string toString() const override { return "Scene[]"; }
private:
    std::vector<Mesh *> m_meshes, m_lights; Sampler *m_sampler = nullptr; Camera *m_camera = nullptr; Integrator *m_integrator = nullptr; Texture<Color3f> *m_background = nullptr;
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

/// include/kazen/block.h:12-85 as far as the adapter needs it: the full-frame accumulation buffer of renderer.cpp:81 - row-major
/// Color4f (rgb x weight, weight), (h + 2 border) x (w + 2 border), border = ceil(radius - 0.5) (block.cpp:13-14,30) - and toBitmap().
struct Color4f { float r = 0, g = 0, b = 0, w = 0; };
class ImageBlock {
public:
    ImageBlock(const Vector2i &size, const ReconstructionFilter *filter) : m_size(size) {
        m_borderSize = filter ? (int)std::ceil(filter->getRadius() - 0.5f) : 0;
        m_px.assign((size_t)(size.x() + 2 * m_borderSize) * (size_t)(size.y() + 2 * m_borderSize), Color4f());
    }
    Color4f *data() { return m_px.data(); }
    const Color4f *data() const { return m_px.data(); }
    size_t size() const { return m_px.size(); }
    const Vector2i &getSize() const { return m_size; }
    int getBorderSize() const { return m_borderSize; }
    /// block.cpp:39-45 + Color4f::divideByFilterWeight (color.h:94-99): rgb / weight, 0 where the weight is 0 (the library's kz_film_to_rgb)
    Bitmap *toBitmap() const {
        std::vector<float> rgb((size_t)m_size.x() * m_size.y() * 3);
        kz_film_to_rgb((const float *)m_px.data(), m_size.x(), m_size.y(), m_borderSize, rgb.data());
        return new Bitmap(m_size.x(), m_size.y(), std::move(rgb));
    }
private:
    Vector2i m_size; int m_borderSize = 0; std::vector<Color4f> m_px;
};

namespace renderer {
/// include/kazen/renderer.h:10 - defined by adapter/renderer_mi355x.cpp (the file INTEGRATION.md adds to a kazen tree), which a program
/// using this mirror compiles and links unchanged
void render(Scene *scene, const std::string &filename);
} // namespace renderer
} // namespace kazen
