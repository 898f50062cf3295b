// parser.cpp — .pbrt (v4 syntax) tokenizer + directive interpreter.  Restates the behaviour of
// src/pbrt/parser.cpp (tokenizer :140-330, parameter lists :434-600, directives :600-1000) and the
// graphics-state bookkeeping of BasicSceneBuilder (src/pbrt/scene.cpp:80-620).  ActiveTransform is tracked (start- and end-time
// CTM) and TransformTimes kept: a CAMERA created under two different CTMs moves over that interval (camera motion blur — what the
// reference's own GPU path supports too); creating a shape, light or medium where the two differ (AnimatedPrimitive) is refused.
#include "scene.h"
#include "hanimated.h"
#include <algorithm>

#include <cctype>
#include <cstdarg>
#include <cstring>
#include <fstream>
#include <sstream>

namespace wf {

[[noreturn]] static void Fatal(const std::string &loc, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    char buf[2048];
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw SceneError("Error: " + loc + ": " + buf);
}

// ---- ParamSet --------------------------------------------------------------------------------------
const Param *ParamSet::Find(const std::string &name) const {
    for (const Param &p : params) if (p.name == name) return &p;
    return nullptr;
}
bool ParamSet::HasParam(const std::string &name) const { return Find(name) != nullptr; }
float ParamSet::GetOneFloat(const std::string &name, float def) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "float") {
            if (p.floats.size() != 1) Fatal(p.loc, "expected one value for \"%s\"", name.c_str());
            p.lookedUp = true;
            return p.floats[0];
        }
    return def;
}
int ParamSet::GetOneInt(const std::string &name, int def) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "integer") {
            if (p.ints.size() != 1) Fatal(p.loc, "expected one value for \"%s\"", name.c_str());
            p.lookedUp = true;
            return p.ints[0];
        }
    return def;
}
bool ParamSet::GetOneBool(const std::string &name, bool def) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "bool") {
            if (p.bools.size() != 1) Fatal(p.loc, "expected one value for \"%s\"", name.c_str());
            p.lookedUp = true;
            return p.bools[0] != 0;
        }
    return def;
}
std::string ParamSet::GetOneString(const std::string &name, const std::string &def) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "string") {
            if (p.strings.size() != 1) Fatal(p.loc, "expected one value for \"%s\"", name.c_str());
            p.lookedUp = true;
            return p.strings[0];
        }
    return def;
}
std::vector<std::string> ParamSet::GetStringArray(const std::string &name) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "string") { p.lookedUp = true; return p.strings; }
    return {};
}
std::vector<float> ParamSet::GetFloatArray(const std::string &name) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "float") { p.lookedUp = true; return p.floats; }
    return {};
}
std::vector<int> ParamSet::GetIntArray(const std::string &name) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "integer") { p.lookedUp = true; return p.ints; }
    return {};
}
std::vector<V3> ParamSet::GetTuple3Array(const std::string &name, const char *type) const {
    for (const Param &p : params)
        if (p.name == name && p.type == type) {
            if (p.floats.size() % 3) Fatal(p.loc, "\"%s\": number of values is not a multiple of 3", name.c_str());
            p.lookedUp = true;
            std::vector<V3> r(p.floats.size() / 3);
            for (size_t i = 0; i < r.size(); ++i) r[i] = V3{p.floats[3 * i], p.floats[3 * i + 1], p.floats[3 * i + 2]};
            return r;
        }
    return {};
}
std::vector<V3> ParamSet::GetPoint3fArray(const std::string &name) const { return GetTuple3Array(name, "point3"); }
std::vector<V2> ParamSet::GetPoint2fArray(const std::string &name) const {
    for (const Param &p : params)
        if (p.name == name && (p.type == "point2" || p.type == "vector2")) {
            if (p.floats.size() % 2) Fatal(p.loc, "\"%s\": number of values is not a multiple of 2", name.c_str());
            p.lookedUp = true;
            std::vector<V2> r(p.floats.size() / 2);
            for (size_t i = 0; i < r.size(); ++i) r[i] = V2{p.floats[2 * i], p.floats[2 * i + 1]};
            return r;
        }
    return {};
}
V3 ParamSet::GetOnePoint3f(const std::string &name, V3 def) const {
    auto v = GetTuple3Array(name, "point3");
    return v.size() == 1 ? v[0] : def;
}
V3 ParamSet::GetOneVector3f(const std::string &name, V3 def) const {
    auto v = GetTuple3Array(name, "vector3");
    return v.size() == 1 ? v[0] : def;
}
std::string ParamSet::GetTexture(const std::string &name) const {
    for (const Param &p : params)
        if (p.name == name && p.type == "texture") {
            if (p.strings.size() != 1) Fatal(p.loc, "expected one texture name for \"%s\"", name.c_str());
            p.lookedUp = true;
            return p.strings[0];
        }
    return "";
}
static SpectrumP ReadSpectrumFile(const std::string &fn) {
    std::ifstream in(fn);
    if (!in) return nullptr;
    std::vector<float> vals;
    std::string tok;
    while (in >> tok) {
        if (tok[0] == '#') { std::string rest; std::getline(in, rest); continue; }
        vals.push_back(strtof(tok.c_str(), nullptr));
    }
    if (vals.empty() || vals.size() % 2) return nullptr;
    std::vector<float> l, v;
    for (size_t i = 0; i < vals.size() / 2; ++i) { l.push_back(vals[2 * i]); v.push_back(vals[2 * i + 1]); }
    return MakePiecewise(l, v);
}
SpectrumP ParamSet::GetOneSpectrum(const std::string &name, SpectrumP def, SpectrumType st) const {
    for (const Param &p : params) {
        if (p.name != name) continue;
        if (p.type == "rgb") {
            if (p.floats.size() != 3) Fatal(p.loc, "\"%s\": expected three RGB values", name.c_str());
            p.lookedUp = true;
            const float *v = p.floats.data();
            const ColorSpace *cs = p.colorSpace ? p.colorSpace : colorSpace;
            if (v[0] < 0 || v[1] < 0 || v[2] < 0) Fatal(p.loc, "RGB parameter \"%s\" has negative component.", name.c_str());
            if (st == SpectrumType::Albedo) {
                if (v[0] > 1 || v[1] > 1 || v[2] > 1) Fatal(p.loc, "RGB parameter \"%s\" has > 1 component.", name.c_str());
                return cs->Albedo(v);
            } else if (st == SpectrumType::Unbounded) return cs->Unbounded(v);
            else return cs->Illuminant(v);
        } else if (p.type == "blackbody") {
            if (p.floats.size() != 1) Fatal(p.loc, "\"%s\": expected one blackbody temperature", name.c_str());
            p.lookedUp = true;
            return MakeBlackbody(p.floats[0]);
        } else if (p.type == "spectrum" && !p.floats.empty()) {
            if (p.floats.size() % 2) Fatal(p.loc, "Found odd number of values for \"%s\"", name.c_str());
            p.lookedUp = true;
            int n = (int)p.floats.size() / 2;
            std::vector<float> l(n), v(n);
            for (int i = 0; i < n; ++i) {
                if (i > 0 && p.floats[2 * i] <= l[i - 1]) Fatal(p.loc, "Spectrum description invalid: wavelengths aren't increasing");
                l[i] = p.floats[2 * i]; v[i] = p.floats[2 * i + 1];
            }
            return MakePiecewise(l, v);
        } else if (p.type == "spectrum" && !p.strings.empty()) {
            p.lookedUp = true;
            SpectrumP s = SpectralData::Get().Named(p.strings[0]);
            if (s) return s;
            s = ReadSpectrumFile(p.strings[0]);
            if (!s) Fatal(p.loc, "%s: unable to read valid spectrum file", p.strings[0].c_str());
            return s;
        }
    }
    return def;
}
// ParameterDictionary::ReportUnused (paramdict.cpp:612-636): a parameter nobody looked up is the reference's ErrorExit — unless a
// parameter of the same type and name in front of it was looked up (a Shape's parameter shadowing the material's, say).  Parameters of
// an Attribute directive may stay unused (scene.cpp:208-212: mayBeUnused; the parser marks them looked-up when it appends them).
void ParamSet::ReportUnused(const std::string &what) const {
    std::vector<const Param *> seen;
    for (const Param &p : params) {
        bool haveSeen = false;
        for (const Param *q : seen) if (q->type == p.type && q->name == p.name) { haveSeen = true; break; }
        if (p.lookedUp) { if (!haveSeen) seen.push_back(&p); }
        else if (!haveSeen) throw SceneError("Error: " + p.loc + ": \"" + p.name + "\": unused parameter.");
    }
    (void)what;
}

// ---- tokenizer ------------------------------------------------------------------------------------
struct Tokenizer {
    std::string text, filename;
    size_t pos = 0;
    int line = 1;
    bool hasUnget = false;
    std::string ungetTok;
    std::string Loc() const { return filename + ":" + std::to_string(line); }
    bool Next(std::string *tok) {
        if (hasUnget) { hasUnget = false; *tok = ungetTok; return true; }
        while (pos < text.size()) {
            char ch = text[pos];
            if (ch == '\n') { ++line; ++pos; }
            else if (ch == ' ' || ch == '\t' || ch == '\r') ++pos;
            else if (ch == '#') { while (pos < text.size() && text[pos] != '\n') ++pos; }
            else break;
        }
        if (pos >= text.size()) return false;
        char ch = text[pos];
        if (ch == '"') {
            size_t start = pos++;
            std::string out = "\"";
            while (pos < text.size() && text[pos] != '"') {
                if (text[pos] == '\n') Fatal(Loc(), "unterminated string");
                if (text[pos] == '\\' && pos + 1 < text.size()) {
                    ++pos;
                    char e = text[pos];
                    switch (e) {
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case '\\': out.push_back('\\'); break;
                    case '\'': out.push_back('\''); break;
                    case '"': out.push_back('"'); break;
                    default: Fatal(Loc(), "unexpected escaped character \"%c\"", e);
                    }
                    ++pos;
                } else out.push_back(text[pos++]);
            }
            if (pos >= text.size()) Fatal(Loc(), "premature EOF in string starting at offset %zu", start);
            ++pos;
            out.push_back('"');
            *tok = out;
            return true;
        }
        if (ch == '[' || ch == ']') { *tok = std::string(1, ch); ++pos; return true; }
        size_t start = pos;
        while (pos < text.size()) {
            char c = text[pos];
            if (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '"' || c == '[' || c == ']') break;
            ++pos;
        }
        *tok = text.substr(start, pos - start);
        return true;
    }
    void Unget(const std::string &t) { hasUnget = true; ungetTok = t; }
};
static bool IsQuoted(const std::string &t) { return t.size() >= 2 && t.front() == '"' && t.back() == '"'; }
static std::string Dequote(const std::string &t, const std::string &loc) {
    if (!IsQuoted(t)) Fatal(loc, "\"%s\": expected quoted string", t.c_str());
    return t.substr(1, t.size() - 2);
}
// parser.cpp:380-417
static bool IsIntegerToken(const std::string &s) {
    if (s.empty()) return false;
    size_t i = (s[0] == '-') ? 1 : 0;
    if (i >= s.size()) return false;
    for (; i < s.size(); ++i) if (!isdigit((unsigned char)s[i])) return false;
    return true;
}
static double ParseFloatTok(const std::string &t, const std::string &loc) {
    if (t.size() == 1 && t[0] >= '0' && t[0] <= '9') return t[0] - '0';
    const char *b = t.c_str();
    char *end;
    double val;
    if (IsIntegerToken(t)) val = double(strtol(b, &end, 10));
    else val = strtof(b, &end);
    if (end == b) Fatal(loc, "%s: expected a number", t.c_str());
    return val;
}
static int ParseIntTok(const std::string &t, const std::string &loc) {
    const char *b = t.c_str();
    char *end;
    long long v = strtoll(b, &end, 10);
    if (end == b || *end) Fatal(loc, "\"%s\": expected an integer", t.c_str());
    return (int)v;
}

// ---- graphics state & interpreter --------------------------------------------------------------------
struct GraphicsState {
    Transform ctm;       // the start-time CTM: what a static render uses
    Transform ctmEnd;    // the end-time CTM (ActiveTransform EndTime); they must agree wherever something is created
    int activeBits = 3;  // ActiveTransform: 1 StartTime, 2 EndTime, 3 All
    bool reverseOrientation = false;
    int currentMaterialIndex = 0;
    std::string currentMaterialName;
    std::string areaLightName;
    ParamSet areaLightParams;
    std::string areaLightLoc;
    std::string currentInsideMedium, currentOutsideMedium;
    const ColorSpace *colorSpace = nullptr;
    std::vector<Param> shapeAttributes, lightAttributes, materialAttributes, mediumAttributes, textureAttributes;
};

struct Interpreter {
    ParsedScene *scene;
    RenderOptions *opt;
    GraphicsState gs;
    std::vector<GraphicsState> pushed;
    std::vector<char> pushKinds;
    std::map<std::string, std::pair<Transform, Transform>> namedCoordinateSystems;   // TransformSet: start- and end-time transformation
    float transformStartTime = 0, transformEndTime = 1;   // TransformTimes
    Transform renderFromWorld;
    bool inWorld = false;
    InstanceDefinition *activeInstance = nullptr;

    ParamSet MakeParams(std::vector<Param> params, const std::vector<Param> &attrs) {
        ParamSet ps;
        ps.colorSpace = gs.colorSpace;
        // ParameterDictionary's constructors (paramdict.cpp:141-160) REVERSE both lists: of two parameters with one name the one written
        // last is the one the look-ups find (differential fuzzing, round 3: "bool remaproughness" given twice); the attributes keep their
        // lower precedence (looked up after the explicit ones)
        ps.params = std::move(params);
        std::reverse(ps.params.begin(), ps.params.end());
        ps.params.insert(ps.params.end(), attrs.rbegin(), attrs.rend());
        return ps;
    }
    // a static render uses the start-time transformation; something created under two different CTMs is animated (AnimatedTransform /
    // AnimatedPrimitive in the reference): refused, not rendered in the wrong place
    void RequireStaticCTM(const std::string &loc) const {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (gs.ctm.m.m[i][j] != gs.ctmEnd.m.m[i][j]) Fatal(loc, "animated transformations (ActiveTransform StartTime / EndTime with different CTMs) are not supported by this build");
    }
    Transform RenderFromObject() const { return Transform((renderFromWorld * gs.ctm).m); }
    bool CTMIsAnimated() const {   // TransformSet::IsAnimated (scene.h): ctm[0] != ctm[1]
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (gs.ctm.m.m[i][j] != gs.ctmEnd.m.m[i][j]) return true;
        return false;
    }

    std::vector<Param> ParseParams(Tokenizer &tz) {
        std::vector<Param> out;
        std::string tok;
        while (tz.Next(&tok)) {
            if (!IsQuoted(tok)) { tz.Unget(tok); break; }
            Param p;
            p.loc = tz.Loc();
            std::string decl = Dequote(tok, tz.Loc());
            std::istringstream ds(decl);
            if (!(ds >> p.type)) Fatal(tz.Loc(), "Parameter \"%s\" doesn't have a type declaration?!", decl.c_str());
            if (!(ds >> p.name)) Fatal(tz.Loc(), "Unable to find parameter name from \"%s\"", decl.c_str());
            p.colorSpace = gs.colorSpace;
            enum { Unknown, String, Bool, Float, Int } valType = (p.type == "integer") ? Int : Unknown;
            auto addVal = [&](const std::string &t) {
                if (IsQuoted(t)) {
                    if (valType != Unknown && valType != String) Fatal(tz.Loc(), "\"%s\": unexpected string value", p.name.c_str());
                    valType = String;
                    p.strings.push_back(Dequote(t, tz.Loc()));
                } else if (t == "true" || t == "false") {
                    if (valType != Unknown && valType != Bool) Fatal(tz.Loc(), "\"%s\": unexpected Boolean value", p.name.c_str());
                    valType = Bool;
                    p.bools.push_back(t == "true");
                } else {
                    if (valType == Unknown) valType = Float;
                    if (valType == String || valType == Bool) Fatal(tz.Loc(), "\"%s\": unexpected numeric value", p.name.c_str());
                    if (valType == Int) p.ints.push_back(ParseIntTok(t, tz.Loc()));
                    else p.floats.push_back((float)ParseFloatTok(t, tz.Loc()));
                }
            };
            std::string val;
            if (!tz.Next(&val)) Fatal(tz.Loc(), "premature EOF in parameter list");
            if (val == "[") {
                while (true) {
                    if (!tz.Next(&val)) Fatal(tz.Loc(), "premature EOF in parameter list");
                    if (val == "]") break;
                    addVal(val);
                }
            } else addVal(val);
            out.push_back(std::move(p));
        }
        return out;
    }

    void Run(Tokenizer &tz) {
        auto nextRequired = [&](const char *what) {
            std::string t;
            if (!tz.Next(&t)) Fatal(tz.Loc(), "premature EOF; expected %s", what);
            return t;
        };
        auto nextFloat = [&]() { return (float)ParseFloatTok(nextRequired("number"), tz.Loc()); };
        auto nextString = [&]() { return Dequote(nextRequired("string"), tz.Loc()); };
        auto basicParamDirective = [&](Entity *dst, const std::vector<Param> &attrs = {}) {
            dst->name = nextString();
            dst->loc = tz.Loc();
            dst->params = MakeParams(ParseParams(tz), attrs);
        };
        std::string tok;
        while (tz.Next(&tok)) {
            const std::string loc = tz.Loc();
            if (tok == "AttributeBegin" || tok == "TransformBegin") {
                pushed.push_back(gs);
                pushKinds.push_back(tok[0] == 'A' ? 'a' : 't');
            } else if (tok == "AttributeEnd" || tok == "TransformEnd") {
                if (pushed.empty()) Fatal(loc, "Unmatched %s encountered.", tok.c_str());
                gs = pushed.back();  // v4 treats the deprecated TransformBegin/End as AttributeBegin/End
                pushed.pop_back();
                pushKinds.pop_back();
            } else if (tok == "Attribute") {
                std::string target = nextString();
                std::vector<Param> ps = ParseParams(tz);
                std::vector<Param> *dst = nullptr;
                if (target == "shape") dst = &gs.shapeAttributes;
                else if (target == "light") dst = &gs.lightAttributes;
                else if (target == "material") dst = &gs.materialAttributes;
                else if (target == "medium") dst = &gs.mediumAttributes;
                else if (target == "texture") dst = &gs.textureAttributes;
                else Fatal(loc, "Unknown attribute target \"%s\".", target.c_str());
                for (Param &p : ps) { p.lookedUp = true; dst->push_back(p); }
            } else if (tok == "ActiveTransform") {
                // BasicSceneBuilder::ActiveTransform{All,StartTime,EndTime} (scene.cpp): which of the two CTMs the following transforms change
                std::string a = nextRequired("ActiveTransform argument");
                if (a == "StartTime") gs.activeBits = 1;
                else if (a == "EndTime") gs.activeBits = 2;
                else if (a == "All") gs.activeBits = 3;
                else Fatal(loc, "Unknown ActiveTransform \"%s\".", a.c_str());
            } else if (tok == "TransformTimes") { transformStartTime = nextFloat(); transformEndTime = nextFloat(); }
            else if (tok == "AreaLightSource") {
                gs.areaLightName = nextString();
                gs.areaLightParams = MakeParams(ParseParams(tz), gs.lightAttributes);
                gs.areaLightLoc = loc;
            } else if (tok == "Accelerator") basicParamDirective(&scene->accelerator);
            else if (tok == "ConcatTransform" || tok == "Transform") {
                if (nextRequired("[") != "[") Fatal(loc, "expected [");
                float m[16];
                for (int i = 0; i < 16; ++i) m[i] = nextFloat();
                if (nextRequired("]") != "]") Fatal(loc, "expected ]");
                Mat4 mm;
                for (int i = 0; i < 16; ++i) mm.m[i / 4][i % 4] = m[i];
                Transform t = TransposeT(Transform(mm));
                if (gs.activeBits & 1) gs.ctm = (tok == "Transform") ? t : gs.ctm * t;
                if (gs.activeBits & 2) gs.ctmEnd = (tok == "Transform") ? t : gs.ctmEnd * t;
            } else if (tok == "CoordinateSystem") namedCoordinateSystems[nextString()] = {gs.ctm, gs.ctmEnd};
            else if (tok == "CoordSysTransform") {
                std::string n = nextString();
                auto it = namedCoordinateSystems.find(n);
                if (it != namedCoordinateSystems.end()) { if (gs.activeBits & 1) gs.ctm = it->second.first; if (gs.activeBits & 2) gs.ctmEnd = it->second.second; }
                else fprintf(stderr, "Warning: %s: Couldn't find named coordinate system \"%s\"\n", loc.c_str(), n.c_str());
            } else if (tok == "ColorSpace") {
                std::string n = nextString();
                const ColorSpace *cs = SpectralData::Get().GetColorSpace(n);
                if (!cs) Fatal(loc, "%s: color space unknown", n.c_str());
                gs.colorSpace = cs;
            } else if (tok == "Camera") {
                basicParamDirective(&scene->camera);
                // BasicSceneBuilder::Camera (scene.cpp:232-262): the camera keeps BOTH CTMs — a camera created under two different
                // ones moves over [TransformTimes] (camera motion blur)
                scene->cameraFromWorld = gs.ctm;
                scene->worldFromCamera = Inverse(gs.ctm);
                scene->worldFromCameraEnd = Inverse(gs.ctmEnd);
                scene->transformStartTime = transformStartTime;
                scene->transformEndTime = transformEndTime;
                namedCoordinateSystems["camera"] = {Inverse(gs.ctm), Inverse(gs.ctmEnd)};
                scene->cameraMedium = gs.currentOutsideMedium;
                // CameraTransform (cameras.cpp:27-57), camera-world rendering space (options default): render space is world space
                // translated to the camera's position at the middle of the time interval
                const wf_animated_transform wfc = MakeAnimatedTransform(scene->worldFromCamera, transformStartTime, scene->worldFromCameraEnd, transformEndTime);
                const float tMid = (transformStartTime + transformEndTime) / 2;
                Transform worldFromRender;   // RenderingCoordinateSystem::World: the identity
                if (opt->renderingSpace == 0) worldFromRender = AnimatedAt(wfc, tMid);   // Camera: worldFromCamera.Interpolate(tMid)
                else if (opt->renderingSpace == 1) worldFromRender = Translate(AnimatedAt(wfc, tMid).Point(V3{0, 0, 0}));   // CameraWorld
                renderFromWorld = Inverse(worldFromRender);
                scene->renderFromWorld = renderFromWorld;
            } else if (tok == "Film") { basicParamDirective(&scene->film); scene->filmColorSpace = gs.colorSpace; }
            else if (tok == "Integrator") basicParamDirective(&scene->integrator);
            else if (tok == "Include" || tok == "Import") {
                std::string fn = nextString();
                if (!fn.empty() && fn[0] != '/') fn = scene->baseDir + "/" + fn;
                std::ifstream in(fn);
                if (!in) Fatal(loc, "%s: unable to open included file", fn.c_str());
                std::stringstream ss;
                ss << in.rdbuf();
                Tokenizer sub;
                sub.text = ss.str();
                sub.filename = fn;
                Run(sub);
            } else if (tok == "Identity") { if (gs.activeBits & 1) gs.ctm = Transform(); if (gs.activeBits & 2) gs.ctmEnd = Transform(); }
            else if (tok == "LightSource") {
                LightEntity e;
                basicParamDirective(&e, gs.lightAttributes);
                RequireStaticCTM(loc);
                e.renderFromLight = RenderFromObject();
                e.medium = gs.currentOutsideMedium;
                scene->lights.push_back(std::move(e));
            } else if (tok == "LookAt") {
                float v[9];
                for (float &f : v) f = nextFloat();
                { Transform t = LookAt(V3{v[0], v[1], v[2]}, V3{v[3], v[4], v[5]}, V3{v[6], v[7], v[8]}); if (gs.activeBits & 1) gs.ctm = gs.ctm * t; if (gs.activeBits & 2) gs.ctmEnd = gs.ctmEnd * t; }
            } else if (tok == "MakeNamedMaterial") {
                Entity e;
                std::string name = nextString();
                e.loc = loc;
                e.params = MakeParams(ParseParams(tz), gs.materialAttributes);
                e.name = e.params.GetOneString("type", "");
                for (auto &nm : scene->namedMaterials) if (nm.first == name) Fatal(loc, "%s: named material redefined.", name.c_str());
                scene->namedMaterials.emplace_back(name, std::move(e));
            } else if (tok == "MakeNamedMedium") {
                Entity e;
                std::string name = nextString();
                e.loc = loc;
                e.params = MakeParams(ParseParams(tz), gs.mediumAttributes);
                e.name = e.params.GetOneString("type", "");
                scene->mediaTransforms[name] = RenderFromObject();
                scene->media.emplace_back(name, std::move(e));
            } else if (tok == "Material") {
                Entity e;
                basicParamDirective(&e, gs.materialAttributes);
                scene->materials.push_back(std::move(e));
                gs.currentMaterialIndex = (int)scene->materials.size() - 1;
                gs.currentMaterialName.clear();
            } else if (tok == "MediumInterface") {
                std::string in = nextString();
                gs.currentInsideMedium = in;
                std::string t2;
                if (tz.Next(&t2)) {
                    if (IsQuoted(t2)) gs.currentOutsideMedium = Dequote(t2, loc);
                    else { gs.currentOutsideMedium = in; tz.Unget(t2); }
                } else gs.currentOutsideMedium = in;
            } else if (tok == "NamedMaterial") {
                gs.currentMaterialName = nextString();
                gs.currentMaterialIndex = -1;
            } else if (tok == "ObjectBegin") {
                std::string name = nextString();
                pushed.push_back(gs);
                pushKinds.push_back('o');
                if (activeInstance) Fatal(loc, "ObjectBegin called inside of instance definition");
                if (scene->instanceDefinitions.count(name)) Fatal(loc, "%s: trying to redefine an object instance", name.c_str());
                scene->instanceDefinitions[name].name = name;
                activeInstance = &scene->instanceDefinitions[name];
            } else if (tok == "ObjectEnd") {
                if (!activeInstance) Fatal(loc, "ObjectEnd called outside of instance definition");
                activeInstance = nullptr;
                gs = pushed.back();
                pushed.pop_back();
                pushKinds.pop_back();
            } else if (tok == "ObjectInstance") {
                std::string name = nextString();
                if (activeInstance) Fatal(loc, "ObjectInstance can't be called inside instance definition");
                InstanceUse u;
                u.name = name;
                // scene.cpp:365-395: renderFromInstance = RenderFromObject() * worldFromRender; under an animated CTM an AnimatedTransform of the
                // two, unless they come out equal
                u.renderFromInstance = RenderFromObject() * Inverse(renderFromWorld);
                u.loc = loc;
                if (CTMIsAnimated()) {
                    u.renderFromInstanceEnd = Transform((renderFromWorld * gs.ctmEnd).m) * Inverse(renderFromWorld);
                    u.startTime = transformStartTime; u.endTime = transformEndTime;
                    u.animated = u.renderFromInstance != u.renderFromInstanceEnd;
                }
                scene->instances.push_back(u);
            } else if (tok == "Option") {
                // parser.cpp:877-880 + BasicSceneBuilder::Option (scene.cpp:489-575): `Option "name" value` — the name without a type, the value a bare
                // token (true / false / number) or a quoted string
                std::string name = nextString(), nName;
                for (char ch : name) if (ch != '_' && ch != '-') nName.push_back((char)tolower((unsigned char)ch));   // normalizeArg
                const std::string value = nextRequired("option value");
                auto boolean = [&](bool *dst) {
                    if (value == "true") *dst = true;
                    else if (value == "false") *dst = false;
                    else Fatal(loc, "%s: expected \"true\" or \"false\" for option value", value.c_str());
                };
                auto quoted = [&]() {
                    if (value.size() < 3 || value.front() != '"' || value.back() != '"') Fatal(loc, "%s: expected quoted string for option value", value.c_str());
                    return value.substr(1, value.size() - 2);
                };
                if (nName == "disablepixeljitter") boolean(&opt->disablePixelJitter);
                else if (nName == "disabletexturefiltering") boolean(&opt->disableTextureFiltering);
                else if (nName == "disablewavelengthjitter") boolean(&opt->disableWavelengthJitter);
                else if (nName == "displacementedgescale") {
                    char *end = nullptr;
                    opt->displacementEdgeScale = strtof(value.c_str(), &end);
                    if (end == value.c_str() || *end) Fatal(loc, "%s: expected floating-point option value", value.c_str());
                } else if (nName == "rendercoordsys") {
                    // (takes effect at the Camera directive, like the reference's CameraTransform)
                    const std::string v = quoted();
                    if (v == "camera") opt->renderingSpace = 0;
                    else if (v == "cameraworld") opt->renderingSpace = 1;
                    else if (v == "world") opt->renderingSpace = 2;
                    else Fatal(loc, "%s: unknown rendering coordinate system.", v.c_str());
                } else if (nName == "seed") opt->seed = atoi(value.c_str());
                else if (nName == "forcediffuse") { bool b = false; boolean(&b); if (b) Fatal(loc, "The wavefront integrator does not support --force-diffuse."); }
                else if (nName == "pixelstats") { bool b = false; boolean(&b); if (b) Fatal(loc, "The wavefront integrator does not support --pixelstats."); }
                else if (nName == "wavefront" || nName == "msereferenceimage" || nName == "msereferenceout") {}   // (this IS the wavefront path; no MSE reference output here)
                else Fatal(loc, "%s: unknown option", name.c_str());
            } else if (tok == "PixelFilter") basicParamDirective(&scene->filter);
            else if (tok == "ReverseOrientation") gs.reverseOrientation = !gs.reverseOrientation;
            else if (tok == "Rotate") {
                float a = nextFloat(), x = nextFloat(), y = nextFloat(), z = nextFloat();
                { Transform t = Rotate(a, V3{x, y, z}); if (gs.activeBits & 1) gs.ctm = gs.ctm * t; if (gs.activeBits & 2) gs.ctmEnd = gs.ctmEnd * t; }
            } else if (tok == "Sampler") basicParamDirective(&scene->sampler);
            else if (tok == "Scale") {
                float x = nextFloat(), y = nextFloat(), z = nextFloat();
                { Transform t = Scale(x, y, z); if (gs.activeBits & 1) gs.ctm = gs.ctm * t; if (gs.activeBits & 2) gs.ctmEnd = gs.ctmEnd * t; }
            } else if (tok == "Shape") {
                ShapeEntity e;
                basicParamDirective(&e, gs.shapeAttributes);
                if (!gs.areaLightName.empty()) {
                    Entity al;
                    al.name = gs.areaLightName;
                    al.params = gs.areaLightParams;
                    al.loc = gs.areaLightLoc;
                    scene->areaLights.push_back(al);
                    e.lightIndex = (int)scene->areaLights.size() - 1;
                    if (activeInstance) fprintf(stderr, "Warning: %s: Area lights not supported with object instancing\n", loc.c_str());
                }
                const bool animated = CTMIsAnimated();
                e.renderFromObject = animated ? Transform() : RenderFromObject();   // (scene.cpp:277-285: an animated shape is created with the identity)
                e.reverseOrientation = gs.reverseOrientation;
                e.materialIndex = gs.currentMaterialIndex;
                e.materialName = gs.currentMaterialName;
                e.insideMedium = gs.currentInsideMedium;
                e.outsideMedium = gs.currentOutsideMedium;
                if (animated) {
                    // AnimatedShapeSceneEntity -> AnimatedPrimitive(BVH of the entity's shapes, renderFromShape) (scene.cpp:1452-1506): here a hidden
                    // instance definition with this one entity, used once with the animated transformation RenderFromObject()
                    if (activeInstance) Fatal(loc, "animated shapes inside an object instance definition are not supported by this build");
                    if (e.lightIndex >= 0) Fatal(loc, "Animated area lights are not supported.");   // scene.cpp:1485-1488
                    InstanceUse u;
                    u.name = std::string("\x01animated-shape#") + std::to_string(scene->animatedShapes.size());
                    u.animated = true;
                    u.renderFromInstance = RenderFromObject();
                    u.renderFromInstanceEnd = Transform((renderFromWorld * gs.ctmEnd).m);
                    u.startTime = transformStartTime; u.endTime = transformEndTime;
                    u.loc = loc;
                    scene->instanceDefinitions[u.name].name = u.name;
                    scene->instanceDefinitions[u.name].shapes.push_back(std::move(e));
                    scene->animatedShapes.push_back(std::move(u));
                } else
                if (activeInstance) activeInstance->shapes.push_back(std::move(e));
                else scene->shapes.push_back(std::move(e));
            } else if (tok == "Texture") {
                TextureEntity e;
                e.texName = nextString();
                e.texType = nextString();
                e.name = nextString();
                e.loc = loc;
                e.params = MakeParams(ParseParams(tz), gs.textureAttributes);
                e.renderFromTexture = RenderFromObject();
                if (e.texType != "float" && e.texType != "spectrum") Fatal(loc, "%s: texture type unknown. Must be \"float\" or \"spectrum\".", e.texType.c_str());
                scene->textures.push_back(std::move(e));
            } else if (tok == "Translate") {
                float x = nextFloat(), y = nextFloat(), z = nextFloat();
                { Transform t = Translate(V3{x, y, z}); if (gs.activeBits & 1) gs.ctm = gs.ctm * t; if (gs.activeBits & 2) gs.ctmEnd = gs.ctmEnd * t; }
            } else if (tok == "WorldBegin") {
                inWorld = true;
                gs.ctm = gs.ctmEnd = Transform();
                gs.activeBits = 3;
                namedCoordinateSystems["world"] = {gs.ctm, gs.ctmEnd};
            } else if (tok == "WorldEnd") {
            } else Fatal(loc, "%s: unknown directive", tok.c_str());
        }
    }
};

static void InitDefaults(ParsedScene *scene, Interpreter *in) {
    in->gs.colorSpace = SpectralData::Get().sRGB();
    scene->filmColorSpace = in->gs.colorSpace;
    // defaults: scene.cpp:86-104
    scene->camera.name = "perspective";
    scene->sampler.name = "zsobol";
    scene->filter.name = "gaussian";
    scene->integrator.name = "volpath";
    scene->accelerator.name = "bvh";
    scene->film.name = "rgb";
    for (Entity *e : {&scene->camera, &scene->sampler, &scene->filter, &scene->integrator, &scene->accelerator, &scene->film})
        e->params.colorSpace = in->gs.colorSpace;
    Entity diffuse;
    diffuse.name = "diffuse";
    diffuse.params.colorSpace = in->gs.colorSpace;
    scene->materials.push_back(diffuse);  // material index 0: default "diffuse" (scene.cpp:97-100)
    scene->renderFromWorld = Inverse(Translate(V3{0, 0, 0}));   // a scene without a Camera directive: the default camera at the origin
}

void ParseFiles(const std::vector<std::string> &files, RenderOptions *opt, ParsedScene *scene) {
    Interpreter in;
    in.scene = scene;
    in.opt = opt;
    InitDefaults(scene, &in);
    for (const std::string &fn : files) {
        std::ifstream f(fn);
        if (!f) Fatal(fn, "unable to open scene file");
        std::stringstream ss;
        ss << f.rdbuf();
        Tokenizer tz;
        tz.text = ss.str();
        tz.filename = fn;
        size_t slash = fn.find_last_of('/');
        scene->baseDir = slash == std::string::npos ? "." : fn.substr(0, slash);
        in.Run(tz);
    }
}
void ParseString(const std::string &text, RenderOptions *opt, ParsedScene *scene) {
    Interpreter in;
    in.scene = scene;
    in.opt = opt;
    InitDefaults(scene, &in);
    Tokenizer tz;
    tz.text = text;
    tz.filename = "<string>";
    scene->baseDir = ".";
    in.Run(tz);
}

}  // namespace wf
