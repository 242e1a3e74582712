// kazen_sceneio.hpp — scene-file loader of the C++ host mirror: nano-kazen XML -> object tree.
//
// Follows the reference's parser (src/kazen/parser.cpp:10-305): the same tag table (:73-97), children are parsed before
// their parent, an object node becomes ObjectFactory::createInstance(type, properties) + setId + addChild(children) +
// activate() (:157-199), property nodes fill the parent's PropertyList (:203-232), a <transform> node composes its
// operations by LEFT-multiplication (:233-290), and the same structural checks raise kazen::Exception (:135-151).
// pugixml is not part of this repository: the small XML reader below handles what scene files contain (declaration,
// comments, elements, quoted attributes, self-closing tags). Meshes are loaded by kazen::Mesh ("obj", kazen_host.hpp).
#pragma once
#include "kazen_host.hpp"

#include <set>

namespace kazen {
namespace sceneio {

struct XmlNode {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<XmlNode> children;
    size_t offset = 0;
    const std::string *attr(const std::string &n) const { for (auto &a : attrs) if (a.first == n) return &a.second; return nullptr; }
    std::string value(const std::string &n) const { const std::string *v = attr(n); return v ? *v : std::string(); }
};

class XmlReader {
public:
    XmlReader(const std::string &text, const std::string &filename) : m_s(text), m_file(filename) {}
    XmlNode root() {
        XmlNode doc;
        for (;;) {
            skipMisc();
            if (m_i >= m_s.size()) break;
            doc.children.push_back(element());
        }
        if (doc.children.size() != 1) fail("expected exactly one root element");
        return doc.children[0];
    }
private:
    [[noreturn]] void fail(const std::string &what) const {
        size_t line = 1, col = 1;
        for (size_t k = 0; k < m_i && k < m_s.size(); ++k) { if (m_s[k] == '\n') { ++line; col = 1; } else ++col; }
        throw Exception("Error while parsing \"" + m_file + "\": " + what + " (at line " + std::to_string(line) + ", col " + std::to_string(col) + ")");
    }
    bool starts(const char *lit) const { return m_s.compare(m_i, std::strlen(lit), lit) == 0; }
    void skipSpace() { while (m_i < m_s.size() && std::isspace((unsigned char)m_s[m_i])) ++m_i; }
    void skipMisc() {          // whitespace, text, <?...?>, <!-- ... -->, <!DOCTYPE ...>
        for (;;) {
            while (m_i < m_s.size() && m_s[m_i] != '<') { if (!std::isspace((unsigned char)m_s[m_i])) fail("unexpected content"); ++m_i; }
            if (m_i >= m_s.size()) return;
            if (starts("<?")) { size_t e = m_s.find("?>", m_i); if (e == std::string::npos) fail("unterminated declaration"); m_i = e + 2; }
            else if (starts("<!--")) { size_t e = m_s.find("-->", m_i); if (e == std::string::npos) fail("unterminated comment"); m_i = e + 3; }
            else if (starts("<!")) { size_t e = m_s.find('>', m_i); if (e == std::string::npos) fail("unterminated declaration"); m_i = e + 1; }
            else return;
        }
    }
    std::string name() {
        size_t b = m_i;
        while (m_i < m_s.size() && (std::isalnum((unsigned char)m_s[m_i]) || m_s[m_i] == '_' || m_s[m_i] == '-' || m_s[m_i] == ':' || m_s[m_i] == '.')) ++m_i;
        if (m_i == b) fail("expected a name");
        return m_s.substr(b, m_i - b);
    }
    static std::string unescape(const std::string &v) {
        std::string o; o.reserve(v.size());
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] != '&') { o.push_back(v[k]); continue; }
            static const std::pair<const char *, char> ent[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool done = false;
            for (auto &e : ent) if (v.compare(k, std::strlen(e.first), e.first) == 0) { o.push_back(e.second); k += std::strlen(e.first) - 1; done = true; break; }
            if (!done) o.push_back('&');
        }
        return o;
    }
    XmlNode element() {
        XmlNode n; n.offset = m_i;
        if (m_s[m_i] != '<') fail("expected an element");
        ++m_i;
        n.tag = name();
        for (;;) {
            skipSpace();
            if (m_i >= m_s.size()) fail("unterminated element");
            if (starts("/>")) { m_i += 2; return n; }
            if (m_s[m_i] == '>') { ++m_i; break; }
            std::string an = name();
            skipSpace();
            if (m_i >= m_s.size() || m_s[m_i] != '=') fail("expected '=' after attribute \"" + an + "\"");
            ++m_i; skipSpace();
            if (m_i >= m_s.size() || (m_s[m_i] != '"' && m_s[m_i] != '\'')) fail("expected a quoted attribute value");
            const char q = m_s[m_i++];
            size_t e = m_s.find(q, m_i);
            if (e == std::string::npos) fail("unterminated attribute value");
            n.attrs.emplace_back(an, unescape(m_s.substr(m_i, e - m_i)));
            m_i = e + 1;
        }
        for (;;) {
            skipMisc();
            if (m_i >= m_s.size()) fail("missing </" + n.tag + ">");
            if (starts("</")) {
                m_i += 2;
                if (name() != n.tag) fail("mismatched closing tag for <" + n.tag + ">");
                skipSpace();
                if (m_i >= m_s.size() || m_s[m_i] != '>') fail("malformed closing tag");
                ++m_i;
                return n;
            }
            n.children.push_back(element());
        }
    }
    const std::string &m_s; std::string m_file; size_t m_i = 0;
};

// ---- string helpers of common.cpp:237-300 ----------------------------------------------------------------------------
inline std::vector<std::string> tokenize(const std::string &s, const std::string &delim = ", ") {
    std::vector<std::string> tok; size_t last = 0, pos = s.find_first_of(delim, last);
    while (last != std::string::npos) {
        if (pos != last) tok.push_back(s.substr(last, pos == std::string::npos ? pos : pos - last));
        last = pos;
        if (last != std::string::npos) { last += 1; pos = s.find_first_of(delim, last); }
    }
    if (!tok.empty() && tok.back().empty()) tok.pop_back();
    return tok;
}
inline float toFloat(const std::string &s) { char *e = nullptr; float r = std::strtof(s.c_str(), &e); if (s.empty() || *e != '\0') throw Exception("Could not parse floating point value \"" + s + "\""); return r; }
inline int toInt(const std::string &s) { char *e = nullptr; long r = std::strtol(s.c_str(), &e, 10); if (s.empty() || *e != '\0') throw Exception("Could not parse integer value \"" + s + "\""); return (int)r; }
inline bool toBool(const std::string &s) {
    std::string v = s; for (auto &c : v) c = (char)std::tolower((unsigned char)c);
    if (v == "false") return false;
    if (v == "true") return true;
    throw Exception("Could not parse boolean value \"" + s + "\"");
}
inline Vector3 toVector3f(const std::string &s) {
    std::vector<std::string> t = tokenize(s);
    if (t.size() != 3) throw Exception("Expected 3 values");
    return {toFloat(t[0]), toFloat(t[1]), toFloat(t[2])};
}

enum ETag { /* object tags = Object::EClassType values */ EPhaseFunction = Object::EClassTypeCount, EBoolean, EInteger, EFloat, EString, EPoint, EVector, EColor,
            ETransform, ETranslate, EMatrix, ERotate, EScale, ELookAt, EInvalid };

class Parser {
public:
    explicit Parser(const std::string &filename) : m_file(filename) {
        m_tags = {{"scene", Object::EScene}, {"mesh", Object::EMesh}, {"bsdf", Object::EBSDF}, {"light", Object::ELight}, {"camera", Object::ECamera},
                  {"medium", Object::EMedium}, {"phase", EPhaseFunction}, {"integrator", Object::EIntegrator}, {"sampler", Object::ESampler},
                  {"texture", Object::ETexture}, {"rfilter", Object::EReconstructionFilter}, {"boolean", EBoolean}, {"integer", EInteger}, {"float", EFloat},
                  {"string", EString}, {"point", EPoint}, {"vector", EVector}, {"color", EColor}, {"transform", ETransform}, {"translate", ETranslate},
                  {"matrix", EMatrix}, {"rotate", ERotate}, {"scale", EScale}, {"lookat", ELookAt}};
    }
    Object *parse(const XmlNode &root) { PropertyList list; return parseTag(root, list, EInvalid); }
private:
    void checkAttributes(const XmlNode &n, std::set<std::string> want) const {        // parser.cpp:100-111
        for (auto &a : n.attrs) {
            auto it = want.find(a.first);
            if (it == want.end()) throw Exception("unexpected attribute \"" + a.first + "\" in \"" + n.tag + "\"");
            want.erase(it);
        }
        if (!want.empty()) throw Exception("missing attribute \"" + *want.begin() + "\" in \"" + n.tag + "\"");
    }
    Object *parseTag(const XmlNode &node, PropertyList &list, int parentTag) {
        auto it = m_tags.find(node.tag);
        if (it == m_tags.end()) throw Exception("Error while parsing \"" + m_file + "\": unexpected tag \"" + node.tag + "\"");
        const int tag = it->second;
        const bool hasParent = parentTag != EInvalid, parentIsObject = hasParent && parentTag < Object::EClassTypeCount;
        const bool currentIsObject = tag < Object::EClassTypeCount;
        const bool parentIsTransform = parentTag == ETransform;
        const bool currentIsTransformOp = tag == ETranslate || tag == ERotate || tag == EScale || tag == ELookAt || tag == EMatrix;
        if (!hasParent && !currentIsObject) throw Exception("Error while parsing \"" + m_file + "\": root element \"" + node.tag + "\" must be a kazen object");
        if (parentIsTransform != currentIsTransformOp) throw Exception("Error while parsing \"" + m_file + "\": transform nodes can only contain transform operations");
        if (hasParent && !parentIsObject && !(parentIsTransform && currentIsTransformOp))
            throw Exception("Error while parsing \"" + m_file + "\": node \"" + node.tag + "\" requires a kazen object as parent");
        if (tag == ETransform) m_transform = Transform();
        PropertyList propList;
        std::vector<std::unique_ptr<Object>> children;
        for (const XmlNode &ch : node.children) { Object *c = parseTag(ch, propList, tag); if (c) children.emplace_back(c); }
        Object *result = nullptr;
        try {
            if (currentIsObject) {
                const std::string type = tag == Object::EScene ? std::string("scene") : node.value("type");
                std::unique_ptr<Object> obj(ObjectFactory::createInstance(type, propList));
                if ((int)obj->getClassType() != tag)
                    throw Exception("Unexpectedly constructed an object of type <" + Object::classTypeName(obj->getClassType()) + "> (expected type <" +
                                    Object::classTypeName((Object::EClassType)tag) + ">): " + obj->toString());
                obj->setId(node.value("id"));
                for (auto &ch : children) { Object *c = ch.release(); obj->addChild(c); c->setParent(obj.get()); }      // the parent owns it from here on
                obj->activate();
                result = obj.release();
            } else {
                switch (tag) {
                case EString: checkAttributes(node, {"name", "value"}); list.setString(node.value("name"), node.value("value")); break;
                case EFloat: checkAttributes(node, {"name", "value"}); list.setFloat(node.value("name"), toFloat(node.value("value"))); break;
                case EInteger: checkAttributes(node, {"name", "value"}); list.setInteger(node.value("name"), toInt(node.value("value"))); break;
                case EBoolean: checkAttributes(node, {"name", "value"}); list.setBoolean(node.value("name"), toBool(node.value("value"))); break;
                case EPoint: checkAttributes(node, {"name", "value"}); list.setPoint(node.value("name"), toVector3f(node.value("value"))); break;
                case EVector: checkAttributes(node, {"name", "value"}); list.setVector(node.value("name"), toVector3f(node.value("value"))); break;
                case EColor: { checkAttributes(node, {"name", "value"}); const Vector3 v = toVector3f(node.value("value")); list.setColor(node.value("name"), Color3f(v[0], v[1], v[2])); } break;
                case ETransform: checkAttributes(node, {"name"}); list.setTransform(node.value("name"), m_transform); break;
                case ETranslate: { checkAttributes(node, {"value"}); const Vector3 v = toVector3f(node.value("value")); Transform t; t.m[3] = v[0]; t.m[7] = v[1]; t.m[11] = v[2]; m_transform = t * m_transform; } break;
                case EMatrix: {
                    checkAttributes(node, {"value"});
                    const std::vector<std::string> tok = tokenize(node.value("value"));
                    if (tok.size() != 16) throw Exception("Expected 16 values");
                    Transform t; for (int i = 0; i < 16; ++i) t.m[i] = toFloat(tok[i]);
                    m_transform = t * m_transform;
                } break;
                case EScale: { checkAttributes(node, {"value"}); const Vector3 v = toVector3f(node.value("value")); Transform t; t.m[0] = v[0]; t.m[5] = v[1]; t.m[10] = v[2]; m_transform = t * m_transform; } break;
                case ERotate: {
                    checkAttributes(node, {"angle", "axis"});
                    const float angle = toFloat(node.value("angle")) * (3.14159265358979323846f / 180.0f);
                    Vector3 a = toVector3f(node.value("axis"));
                    const float l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);         // Eigen::AngleAxis expects a unit axis; scene files give one
                    if (l > 0.f) a = {a[0] / l, a[1] / l, a[2] / l};
                    const float c = std::cos(angle), s = std::sin(angle), x = a[0], y = a[1], z = a[2];
                    Transform t;
                    t.m = {c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0,
                           y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0,
                           z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0, 0, 0, 0, 1};
                    m_transform = t * m_transform;
                } break;
                case ELookAt: {
                    checkAttributes(node, {"origin", "target", "up"});
                    m_transform = Transform::lookAt(toVector3f(node.value("origin")), toVector3f(node.value("target")), toVector3f(node.value("up"))) * m_transform;
                } break;
                default: throw Exception("Unhandled element \"" + node.tag + "\"");
                }
            }
        } catch (const Exception &e) {
            const std::string what = e.what();
            if (what.rfind("Error while parsing", 0) == 0) throw;
            throw Exception("Error while parsing \"" + m_file + "\": " + what + " (in <" + node.tag + ">)");
        }
        return result;
    }
    std::string m_file; std::map<std::string, int> m_tags; Transform m_transform;
};

} // namespace sceneio

/// parser.h: load a scene from the specified filename and return its root object (the caller owns it)
inline Object *loadFromXML(const std::string &filename) {
    std::ifstream is(filename);
    if (is.fail()) throw Exception("Error while parsing \"" + filename + "\": file not found");
    std::stringstream ss; ss << is.rdbuf();
    const std::string text = ss.str();
    const size_t slash = filename.find_last_of('/');
    fileResolverPaths().insert(fileResolverPaths().begin(), slash == std::string::npos ? std::string(".") : filename.substr(0, slash));     // main.cpp: resolver->prepend(parent_path)
    sceneio::XmlReader reader(text, filename);
    const sceneio::XmlNode root = reader.root();
    sceneio::Parser parser(filename);
    return parser.parse(root);
}

} // namespace kazen
