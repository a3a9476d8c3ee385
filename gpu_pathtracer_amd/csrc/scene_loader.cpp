// scene_loader.cpp — host-side scene loading for the path: the reference's
//   LoadScene            src/parsescene.cpp:45-590   (the keys the "pt" path reads)
//   Mesh::processMesh    src/mesh.cpp:29-111         (vertex transform, tangents, triangle soup)
//   Scene::Init          src/scene.h:50-83           (BVH, env bounding sphere, light power CDF)
//   BVH::LoadOrBuildBVH  src/bvh.cpp:189-218         (bvh.cache, same file layout)
//   Camera               src/camera.h:31-46,123-128
// re-implemented without the reference's third-party stack: own JSON reader (rapidjson there), own OBJ
// reader (assimp there; the Windows .lib is all the reference vendors), own 4x4 transform code (glm
// there — the operation ORDER of glm's translate/rotate/scale/operator* is followed so transformed
// vertices round the same way).
//
// OBJ rule (what assimp does for aiProcess_Triangulate on the meshes the reference ships): one vertex
// per face corner, polygons fan-triangulated, no vertex joining.  For files WITHOUT `vn`
// (aiProcess_GenSmoothNormals) the declared rule is: normal of a position index = normalize(sum of the
// un-normalised face normals cross(p1-p0, p2-p0) of every triangle that uses that index).  assimp's exact
// smoothing (it also merges coincident positions with different indices) is not pinned — DESIGN.md.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "host_util.h"
#include "imageio.h"
#include "pathtracer.h"
#include "../../include/gpt_wide_bvh.h"
#include "pt_vec.h"

using pt::V3;

// ================================================================= JSON ==========
namespace {

struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    bool has(const char *k) const
    {
        for (auto &kv : obj) if (kv.first == k) return true;
        return false;
    }
    const Json &at(const char *k) const
    {
        static const Json null_json;
        for (auto &kv : obj) if (kv.first == k) return kv.second;
        return null_json;
    }
    double number(double dflt) const { return kind == Num ? num : dflt; }
};

// The reader follows rapidjson's (the reference parses its scene files with Document::Parse and the default flags,
// src/parsescene.cpp:60-61; rapidjson is vendored under include/rapidjson):
//   * the grammar it accepts and refuses: RFC 8259 values at the root, blanks = space / tab / CR / LF only, no comments, no
//     trailing commas, no NaN / Infinity, escapes \" \\ \/ \b \f \n \r \t \uXXXX (surrogate pairs joined, UTF-8 out), raw
//     bytes below 0x20 refused, anything after the root value refused, the text ends at its first NUL;
//   * its NUMBER ARITHMETIC (reader.h ParseNumber + internal/strtod.h StrtodNormalPrecision, the default "normal precision"
//     mode): the digits are gathered into an integer significand (at most 2^53 before the fraction digits stop being
//     taken exactly, 17 significant digits in all), and the value is that significand times or divided by a power of ten —
//     two roundings, not the correctly rounded strtod.  A scene value is a float made from that double, so the double is
//     reproduced exactly (oracle/ref_json.cpp holds rapidjson itself; tests/test_json_reference.py).
struct JsonParser {
    const char *p, *end;
    std::string err;
    int peek() const { return p < end ? (unsigned char)*p : 0; }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    static double pow10(int n)                          // 1e0 .. 1e308, each the double nearest to the power
    {
        struct Table { double v[309]; };
        static const Table table = [] {                 // (initialised once, also when two threads load their first scene together)
            Table t;
            char text[16];
            for (int i = 0; i <= 308; ++i) { std::snprintf(text, sizeof(text), "1e%d", i); t.v[i] = std::strtod(text, nullptr); }
            return t;
        }();
        return table.v[n];
    }
    static double scale10(double d, int e) { return e < -308 ? 0.0 : (e >= 0 ? d * pow10(e) : d / pow10(-e)); }
    bool digit() const { return p < end && *p >= '0' && *p <= '9'; }
    bool parse_number(Json &j)
    {
        bool minus = false;
        if (peek() == '-') { minus = true; ++p; }
        uint64_t sig = 0;                                // the significand while it is exact
        bool is_double = false;
        double d = 0.0;
        int digits = 0;                                  // significant digits counted so far (the first integer digit is not)
        if (peek() == '0') ++p;
        else if (peek() >= '1' && peek() <= '9') {
            sig = (uint64_t)(*p++ - '0');
            // integers stay exact up to the 64-bit range rapidjson reports them in (a minus sign: up to 2^63)
            const uint64_t tenth = minus ? 0x0CCCCCCCCCCCCCCCull : 0x1999999999999999ull;
            const char last = minus ? '8' : '5';
            while (digit()) {
                if (sig >= tenth && (sig != tenth || *p > last)) { d = (double)sig; is_double = true; break; }
                sig = sig * 10 + (uint64_t)(*p++ - '0');
                ++digits;
            }
            if (is_double)
                while (digit()) {
                    if (d >= 1.7976931348623157e307) return fail("number too big");
                    d = d * 10 + (*p++ - '0');
                }
        } else return fail("invalid value");
        bool integral = !is_double;
        int exp_frac = 0;
        if (peek() == '.') {
            ++p;
            if (!digit()) return fail("missing fraction");
            if (!is_double) {
                while (digit()) {
                    if (sig > 0x1FFFFFFFFFFFFFull) break;       // 2^53 - 1: beyond it the digits go through the double
                    sig = sig * 10 + (uint64_t)(*p++ - '0');
                    --exp_frac;
                    if (sig != 0) ++digits;
                }
                d = (double)sig;
                is_double = true;
            }
            while (digit()) {
                if (digits < 17) {
                    d = d * 10.0 + (*p++ - '0');
                    --exp_frac;
                    if (d > 0.0) ++digits;
                } else ++p;
            }
            integral = false;
        }
        int exp = 0;
        if (peek() == 'e' || peek() == 'E') {
            if (!is_double) { d = (double)sig; is_double = true; }
            integral = false;
            ++p;
            bool exp_minus = false;
            if (peek() == '+') ++p;
            else if (peek() == '-') { ++p; exp_minus = true; }
            if (!digit()) return fail("missing exponent");
            exp = *p++ - '0';
            if (exp_minus) {
                while (digit()) {
                    exp = exp * 10 + (*p++ - '0');
                    if (exp >= 214748364) while (digit()) ++p;
                }
                exp = -exp;
            } else {
                const int max_exp = 308 - exp_frac;
                while (digit()) {
                    exp = exp * 10 + (*p++ - '0');
                    if (exp > max_exp) return fail("number too big");
                }
            }
        }
        j.kind = Json::Num;
        if (integral) {
            // an integer token: GetDouble() converts the 32- or 64-bit integer (two's complement for a minus sign)
            j.num = minus ? (double)(int64_t)(~sig + 1) : (double)sig;
            return true;
        }
        const int e10 = exp + exp_frac;
        d = e10 < -308 ? scale10(scale10(d, -308), e10 + 308) : scale10(d, e10);
        j.num = minus ? -d : d;
        return true;
    }
    bool hex4(unsigned &cp)
    {
        cp = 0;
        for (int i = 0; i < 4; ++i) {
            const int c = peek();
            cp <<= 4;
            if (c >= '0' && c <= '9') cp += (unsigned)(c - '0');
            else if (c >= 'A' && c <= 'F') cp += (unsigned)(c - 'A' + 10);
            else if (c >= 'a' && c <= 'f') cp += (unsigned)(c - 'a' + 10);
            else return fail("incorrect hex digit after \\u escape");
            ++p;
        }
        return true;
    }
    bool parse_string(std::string &out)
    {
        ++p;                                             // the opening quote
        for (;;) {
            const int c = peek();
            if (c == '\\') {
                ++p;
                const int e = peek();
                ++p;
                switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = 0;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        if (peek() != '\\') return fail("invalid surrogate pair");
                        ++p;
                        if (peek() != 'u') return fail("invalid surrogate pair");
                        ++p;
                        unsigned lo = 0;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("invalid surrogate pair");
                        cp = (((cp - 0xD800) << 10) | (lo - 0xDC00)) + 0x10000;
                    }
                    if (cp <= 0x7F) out += (char)cp;
                    else if (cp <= 0x7FF) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else if (cp <= 0xFFFF) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: return fail("invalid escape character in string");
                }
            } else if (c == '"') { ++p; return true; }
            else if (c == 0) return fail("missing a closing quotation mark in string");
            else if (c < 0x20) return fail("invalid character in string");
            else { out += (char)c; ++p; }
        }
    }
    bool literal(const char *word, size_t n)
    {
        if ((size_t)(end - p) < n || std::strncmp(p, word, n)) return fail("invalid value");
        p += n;
        return true;
    }
    int depth = 0;                                      // nesting: bounded (rapidjson recurses until the stack runs out; a scene file is 3 deep)
    struct Nest { int &d; explicit Nest(int &x) : d(x) { ++d; } ~Nest() { --d; } };
    bool parse_value(Json &j)
    {
        Nest nest(depth);
        if (depth > 512) return fail("nested too deeply");
        switch (peek()) {
        case 'n': j.kind = Json::Null; return literal("null", 4);
        case 't': j.kind = Json::Bool; j.b = true; return literal("true", 4);
        case 'f': j.kind = Json::Bool; j.b = false; return literal("false", 5);
        case '"': j.kind = Json::Str; return parse_string(j.str);
        case '{':
            ++p; j.kind = Json::Obj;
            ws();
            if (peek() == '}') { ++p; return true; }
            for (;;) {
                if (peek() != '"') return fail("missing a name for object member");
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (peek() != ':') return fail("missing a colon after a name of object member");
                ++p;
                ws();
                Json v;
                if (!parse_value(v)) return false;
                j.obj.emplace_back(std::move(key), std::move(v));
                ws();
                if (peek() == ',') { ++p; ws(); continue; }
                if (peek() == '}') { ++p; return true; }
                return fail("missing a comma or '}' after an object member");
            }
        case '[':
            ++p; j.kind = Json::Arr;
            ws();
            if (peek() == ']') { ++p; return true; }
            for (;;) {
                Json v;
                if (!parse_value(v)) return false;
                j.arr.push_back(std::move(v));
                ws();
                if (peek() == ',') { ++p; ws(); continue; }
                if (peek() == ']') { ++p; return true; }
                return fail("missing a comma or ']' after an array element");
            }
        default: return parse_number(j);
        }
    }
    // a whole document: one value, blanks around it, nothing else (the text ends at its first NUL, as the reference's buffer does)
    bool parse(Json &j)
    {
        const char *nul = (const char *)std::memchr(p, 0, (size_t)(end - p));
        if (nul) end = nul;
        ws();
        if (p >= end) return fail("the document is empty");
        if (!parse_value(j)) return false;
        ws();
        if (p < end) return fail("the document root must not be followed by other values");
        return true;
    }
};

// ================================================================= mat4 (glm order) ===
struct M4 {
    float c[4][4];   // c[column][row]
};
M4 identity()
{
    M4 m;
    std::memset(&m, 0, sizeof(m));
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.f;
    return m;
}
void col_madd3(float out[4], const M4 &m, float a, float b, float c)   // m[0]*a + m[1]*b + m[2]*c
{
    for (int r = 0; r < 4; ++r) out[r] = m.c[0][r] * a + m.c[1][r] * b + m.c[2][r] * c;
}
M4 translate(const M4 &m, float x, float y, float z)   // glm::translate: Result[3] = m[0]*v0 + m[1]*v1 + m[2]*v2 + m[3]
{
    M4 r = m;
    for (int k = 0; k < 4; ++k) r.c[3][k] = m.c[0][k] * x + m.c[1][k] * y + m.c[2][k] * z + m.c[3][k];
    return r;
}
M4 scale(const M4 &m, float x, float y, float z)
{
    M4 r = m;
    for (int k = 0; k < 4; ++k) { r.c[0][k] = m.c[0][k] * x; r.c[1][k] = m.c[1][k] * y; r.c[2][k] = m.c[2][k] * z; }
    return r;
}
M4 rotate(const M4 &m, float angle, float ax, float ay, float az)   // glm::rotate (axis already unit here)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const float axis[3] = {ax, ay, az};
    const float temp[3] = {(1.f - c) * ax, (1.f - c) * ay, (1.f - c) * az};
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    M4 r;
    col_madd3(r.c[0], m, R[0][0], R[0][1], R[0][2]);
    col_madd3(r.c[1], m, R[1][0], R[1][1], R[1][2]);
    col_madd3(r.c[2], m, R[2][0], R[2][1], R[2][2]);
    for (int k = 0; k < 4; ++k) r.c[3][k] = m.c[3][k];
    return r;
}
M4 mul(const M4 &a, const M4 &b)   // glm operator*(mat4, mat4)
{
    M4 r;
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
            r.c[j][k] = a.c[0][k] * b.c[j][0] + a.c[1][k] * b.c[j][1] + a.c[2][k] * b.c[j][2] + a.c[3][k] * b.c[j][3];
    return r;
}
void mulv(const M4 &m, const float v[4], float out[4])   // glm operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
{
    for (int k = 0; k < 4; ++k) out[k] = (m.c[0][k] * v[0] + m.c[1][k] * v[1]) + (m.c[2][k] * v[2] + m.c[3][k] * v[3]);
}
M4 transpose(const M4 &m)
{
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.c[i][j] = m.c[j][i];
    return r;
}
// glm::inverse(mat4) of the glm the reference vendors (0.9.7, glm/detail/type_mat4x4.inl compute_inverse): adjugate over determinant,
// with glm's grouping of the float operations so that the result is glm's to the last bit (oracle/ref_transform.cpp holds glm itself;
// tests/test_transform_reference.py).  c[j][k] = column j, row k, as in glm.
M4 inverse(const M4 &m)
{
    // 2x2 minors of the columns (j1, j2) in {(2,3), (1,3), (1,2)} over every pair of rows p < q
    auto minor2 = [&](int j1, int j2, int p, int q) { return m.c[j1][p] * m.c[j2][q] - m.c[j2][p] * m.c[j1][q]; };
    float fac[4][4][4];                                  // fac[p][q][k]: the minor that element k of an adjugate column multiplies
    for (int p = 0; p < 4; ++p)
        for (int q = p + 1; q < 4; ++q) {
            fac[p][q][0] = fac[p][q][1] = minor2(2, 3, p, q);
            fac[p][q][2] = minor2(1, 3, p, q);
            fac[p][q][3] = minor2(1, 2, p, q);
        }
    M4 adj;
    for (int r = 0; r < 4; ++r) {                        // column r of the adjugate: expand along the three other rows a < b < e
        int o[3], n = 0;
        for (int k = 0; k < 4; ++k) if (k != r) o[n++] = k;
        const int a = o[0], b = o[1], e = o[2];
        for (int k = 0; k < 4; ++k) {
            const int col = k == 0 ? 1 : 0;              // glm's Vec*: (m[1][row], m[0][row], m[0][row], m[0][row])
            const float v = (m.c[col][a] * fac[b][e][k] - m.c[col][b] * fac[a][e][k]) + m.c[col][e] * fac[a][b][k];
            adj.c[r][k] = ((r + k) & 1) ? v * -1.0f : v * 1.0f;
        }
    }
    const float det = (m.c[0][0] * adj.c[0][0] + m.c[0][1] * adj.c[1][0]) + (m.c[0][2] * adj.c[2][0] + m.c[0][3] * adj.c[3][0]);
    const float inv_det = 1.0f / det;
    M4 r;
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) r.c[j][k] = adj.c[j][k] * inv_det;
    return r;
}
float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

M4 trs_matrix(const float scale_[3], const float translate_[3], const float rotate_[3])   // parsescene.cpp:346-352
{
    M4 s = scale(identity(), scale_[0], scale_[1], scale_[2]);
    M4 t = translate(identity(), translate_[0], translate_[1], translate_[2]);
    M4 r = rotate(identity(), radians(rotate_[0]), 1, 0, 0);
    r = rotate(r, radians(rotate_[1]), 0, 1, 0);
    r = rotate(r, radians(rotate_[2]), 0, 0, 1);
    return mul(mul(t, r), s);
}

// ================================================================= OBJ =================
struct ObjMesh {
    std::vector<V3> v, vn;
    std::vector<pt::V2> vt;
    struct Corner { int v, vt, vn; };
    std::vector<Corner> corners;   // 3 per triangle
};

bool read_obj(const std::string &path, ObjMesh &m)
{
    FILE *f = gpt_fopen_read(path.c_str());
    if (!f) return false;
    char line[4096];
    std::vector<ObjMesh::Corner> face;
    while (std::fgets(line, sizeof(line), f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            V3 a{0, 0, 0};
            char *e = p + 1;
            a.x = std::strtof(e, &e); a.y = std::strtof(e, &e); a.z = std::strtof(e, &e);
            m.v.push_back(a);
        } else if (p[0] == 'v' && p[1] == 'n') {
            V3 a{0, 0, 0};
            char *e = p + 2;
            a.x = std::strtof(e, &e); a.y = std::strtof(e, &e); a.z = std::strtof(e, &e);
            m.vn.push_back(a);
        } else if (p[0] == 'v' && p[1] == 't') {
            pt::V2 a{0, 0};
            char *e = p + 2;
            a.x = std::strtof(e, &e); a.y = std::strtof(e, &e);
            m.vt.push_back(a);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            face.clear();
            char *e = p + 1;
            for (;;) {
                while (*e == ' ' || *e == '\t') ++e;
                if (*e == 0 || *e == '\n' || *e == '\r' || *e == '#') break;
                ObjMesh::Corner c{0, 0, 0};
                c.v = (int)std::strtol(e, &e, 10);
                if (*e == '/') {
                    ++e;
                    if (*e != '/') c.vt = (int)std::strtol(e, &e, 10);
                    if (*e == '/') { ++e; c.vn = (int)std::strtol(e, &e, 10); }
                }
                if (c.v < 0) c.v = (int)m.v.size() + c.v + 1;
                if (c.vt < 0) c.vt = (int)m.vt.size() + c.vt + 1;
                if (c.vn < 0) c.vn = (int)m.vn.size() + c.vn + 1;
                if (c.v <= 0 || c.v > (int)m.v.size()) { std::fclose(f); return false; }
                if (c.vt > (int)m.vt.size() || c.vn > (int)m.vn.size()) { std::fclose(f); return false; }
                face.push_back(c);
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                m.corners.push_back(face[0]);
                m.corners.push_back(face[k]);
                m.corners.push_back(face[k + 1]);
            }
        }
    }
    std::fclose(f);
    return true;
}

// ================================================================= PLY =================
// Stanford PLY (ascii and binary_little_endian): element vertex with x y z [nx ny nz] [s t | u v | texture_u texture_v],
// element face with a vertex index list.  The reference reads meshes through assimp (mesh.cpp:4-27) and one shipped scene
// (veach_bidir) names .ply files; the mesh is handed on in the same per-corner form as an OBJ (normal / uv index = vertex
// index), so everything downstream - fan triangulation, transform, missing-normal rule - is shared.
bool read_ply(const std::string &path, ObjMesh &m)
{
    FILE *f = gpt_fopen_read(path.c_str());
    if (!f) return false;
    struct Prop { std::string name, type, count_type; bool list; };
    struct Elem { std::string name; long count; std::vector<Prop> props; };
    std::vector<Elem> elems;
    bool ascii = false, binary_le = false, header_ok = false;
    char line[1024];
    if (!std::fgets(line, sizeof(line), f) || std::strncmp(line, "ply", 3) != 0) { std::fclose(f); return false; }
    while (std::fgets(line, sizeof(line), f)) {
        char a[64] = "", b[64] = "", c[64] = "", d[64] = "", e[64] = "";
        const int n = std::sscanf(line, "%63s %63s %63s %63s %63s", a, b, c, d, e);
        if (n < 1) continue;
        const std::string key = a;
        if (key == "format") { ascii = std::string(b) == "ascii"; binary_le = std::string(b) == "binary_little_endian"; }
        else if (key == "element" && n >= 3) elems.push_back(Elem{b, std::atol(c), {}});
        else if (key == "property" && !elems.empty()) {
            if (std::string(b) == "list" && n >= 5) elems.back().props.push_back(Prop{e, d, c, true});
            else if (n >= 3) elems.back().props.push_back(Prop{c, b, "", false});
        } else if (key == "end_header") { header_ok = true; break; }
    }
    if (!header_ok || (!ascii && !binary_le)) { std::fclose(f); return false; }
    auto type_size = [](const std::string &t) {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        return 0;
    };
    bool ok = true;
    auto read_scalar = [&](const std::string &t, double &out) {       // one value of type t -> double
        if (ascii) return std::fscanf(f, "%lf", &out) == 1;
        unsigned char buf[8];
        const int sz = type_size(t);
        if (sz == 0 || std::fread(buf, 1, (size_t)sz, f) != (size_t)sz) return false;
        if (t == "char" || t == "int8") out = (signed char)buf[0];
        else if (t == "uchar" || t == "uint8") out = buf[0];
        else if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, buf, 2); out = v; }
        else if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, buf, 2); out = v; }
        else if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, buf, 4); out = v; }
        else if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, buf, 4); out = v; }
        else if (t == "float" || t == "float32") { float v; std::memcpy(&v, buf, 4); out = v; }
        else { double v; std::memcpy(&v, buf, 8); out = v; }
        return true;
    };
    bool have_n = false, have_uv = false;
    for (const Elem &el : elems) {
        for (long i = 0; i < el.count && ok; ++i) {
            V3 pos{0, 0, 0}, nor{0, 0, 0};
            pt::V2 uv{0, 0};
            std::vector<int> face;
            for (const Prop &pr : el.props) {
                if (pr.list) {
                    double cnt = 0;
                    if (!read_scalar(pr.count_type, cnt) || cnt < 0 || cnt > 1e6) { ok = false; break; }
                    const bool is_idx = el.name == "face" && (pr.name == "vertex_indices" || pr.name == "vertex_index");
                    for (int k = 0; k < (int)cnt; ++k) {
                        double v = 0;
                        if (!read_scalar(pr.type, v)) { ok = false; break; }
                        if (is_idx) face.push_back((int)v);
                    }
                } else {
                    double v = 0;
                    if (!read_scalar(pr.type, v)) { ok = false; break; }
                    if (el.name != "vertex") continue;
                    const float fv = (float)v;
                    if (pr.name == "x") pos.x = fv; else if (pr.name == "y") pos.y = fv; else if (pr.name == "z") pos.z = fv;
                    else if (pr.name == "nx") { nor.x = fv; have_n = true; } else if (pr.name == "ny") nor.y = fv; else if (pr.name == "nz") nor.z = fv;
                    else if (pr.name == "s" || pr.name == "u" || pr.name == "texture_u") { uv.x = fv; have_uv = true; }
                    else if (pr.name == "t" || pr.name == "v" || pr.name == "texture_v") uv.y = fv;
                }
            }
            if (!ok) break;
            if (el.name == "vertex") {
                m.v.push_back(pos);
                m.vn.push_back(nor);
                m.vt.push_back(uv);
            } else if (el.name == "face") {
                for (int idx : face)
                    if (idx < 0 || idx >= (int)m.v.size()) ok = false;
                for (size_t k = 1; ok && k + 1 < face.size(); ++k)
                    for (int idx : {face[0], face[k], face[k + 1]})
                        m.corners.push_back(ObjMesh::Corner{idx + 1, have_uv ? idx + 1 : 0, have_n ? idx + 1 : 0});
            }
        }
    }
    std::fclose(f);
    if (!have_n) m.vn.clear();
    if (!have_uv) m.vt.clear();
    return ok;
}

gpt_float3 g3(V3 a) { return gpt_float3{a.x, a.y, a.z}; }
V3 v3of(gpt_float3 a) { return V3{a.x, a.y, a.z}; }

// wrap.h:6-16
void make_coordinate(V3 n, V3 &u, V3 &w)
{
    if (std::fabs(n.x) > std::fabs(n.y)) {
        float invLen = 1.0f / pt::sqrt_rn(n.x * n.x + n.z * n.z);
        w = pt::v3(n.z * invLen, 0.0f, -n.x * invLen);
    } else {
        float invLen = 1.0f / pt::sqrt_rn(n.y * n.y + n.z * n.z);
        w = pt::v3(0.0f, n.z * invLen, -n.y * invLen);
    }
    u = pt::cross(w, n);
}

// Mesh::genTangent, src/mesh.cpp:93-111
V3 gen_tangent(const gpt_vertex &v1, const gpt_vertex &v2, const gpt_vertex &v3)
{
    const pt::V2 duv1{v2.uv.x - v1.uv.x, v2.uv.y - v1.uv.y}, duv2{v3.uv.x - v1.uv.x, v3.uv.y - v1.uv.y};
    const V3 e1 = v3of(v2.v) - v3of(v1.v), e2 = v3of(v3.v) - v3of(v1.v);
    const float det = duv1.x * duv2.y - duv1.y * duv2.x;
    if (!((double)std::fabs(det) < 1e-8)) {
        float invdet = 1.f / det;
        return pt::normalize((-duv2.x * e1 + duv1.y * e2) * invdet);
    }
    V3 uu, ww;
    V3 nn = pt::normalize(pt::cross(e1, e2));
    make_coordinate(nn, uu, ww);
    return uu;
}

// Mesh::LoadObjFromFile + processMesh (src/mesh.cpp:4-91) for one OBJ file
bool load_mesh(const std::string &path, const M4 &trs, int matIdx, int bssrdfIdx, std::vector<Triangle> &out)
{
    ObjMesh m;
    const bool is_ply = path.size() > 4 && (path.compare(path.size() - 4, 4, ".ply") == 0 || path.compare(path.size() - 4, 4, ".PLY") == 0);
    if (!(is_ply ? read_ply(path, m) : read_obj(path, m))) {
        gpt_set_error("Error when import model: cannot read \"%s\"", path.c_str());
        return false;
    }
    const size_t ntri = m.corners.size() / 3;
    std::vector<V3> smooth;
    if (m.vn.empty()) {   // aiProcess_GenSmoothNormals stand-in (see file header)
        smooth.assign(m.v.size(), V3{0, 0, 0});
        for (size_t t = 0; t < ntri; ++t) {
            const V3 p0 = m.v[(size_t)m.corners[3 * t].v - 1], p1 = m.v[(size_t)m.corners[3 * t + 1].v - 1],
                     p2 = m.v[(size_t)m.corners[3 * t + 2].v - 1];
            const V3 fn = pt::cross(p1 - p0, p2 - p0);
            for (int k = 0; k < 3; ++k) smooth[(size_t)m.corners[3 * t + k].v - 1] += fn;
        }
        for (auto &n : smooth) {
            float d = pt::dot(n, n);
            n = d > 0.f ? pt::normalize(n) : V3{0, 1, 0};
        }
    }
    // one vertex per face corner
    std::vector<gpt_vertex> verts(m.corners.size());
    const M4 invT = transpose(inverse(trs));
    for (size_t i = 0; i < m.corners.size(); ++i) {
        const ObjMesh::Corner &c = m.corners[i];
        gpt_vertex vx;
        std::memset(&vx, 0, sizeof(vx));
        const V3 p = m.v[(size_t)c.v - 1];
        const V3 n = c.vn > 0 ? m.vn[(size_t)c.vn - 1] : (m.vn.empty() ? smooth[(size_t)c.v - 1] : V3{0, 1, 0});
        float pv[4] = {p.x, p.y, p.z, 1.f}, nv[4] = {n.x, n.y, n.z, 0.f}, o[4];
        mulv(trs, pv, o);
        vx.v = gpt_float3{o[0], o[1], o[2]};
        mulv(invT, nv, o);
        vx.n = g3(pt::normalize(V3{o[0], o[1], o[2]}));
        if (c.vt > 0) { vx.uv.x = m.vt[(size_t)c.vt - 1].x; vx.uv.y = m.vt[(size_t)c.vt - 1].y; }
        verts[i] = vx;
    }
    // tangents: accumulate per vertex, normalise (mesh.cpp:62-76); unique corners => one face each
    for (size_t t = 0; t < ntri; ++t) {
        V3 tg = gen_tangent(verts[3 * t], verts[3 * t + 1], verts[3 * t + 2]);
        for (int k = 0; k < 3; ++k) verts[3 * t + k].t = g3(pt::normalize(tg));
    }
    out.reserve(out.size() + ntri);
    for (size_t t = 0; t < ntri; ++t) {
        Triangle tri;
        std::memset(&tri, 0, sizeof(tri));
        tri.v1 = verts[3 * t]; tri.v2 = verts[3 * t + 1]; tri.v3 = verts[3 * t + 2];
        tri.matIdx = matIdx;
        tri.bssrdfIdx = bssrdfIdx;
        tri.lightIdx = -1;
        tri.mediumInside = tri.mediumOutside = -1;
        out.push_back(tri);
    }
    std::fprintf(stdout, "Load Model sucessfully: %s\nMerge [%d] triangles\n", path.c_str(), (int)ntri);
    return true;
}

void get3(const Json &j, const char *key, const float dflt[3], float out[3])
{
    out[0] = dflt[0]; out[1] = dflt[1]; out[2] = dflt[2];
    if (j.has(key) && j.at(key).kind == Json::Arr)
        for (size_t i = 0; i < 3 && i < j.at(key).arr.size(); ++i) out[i] = (float)j.at(key).arr[i].num;
}
float getf(const Json &j, const char *key, float dflt) { return j.has(key) ? (float)j.at(key).number(dflt) : dflt; }
bool getb(const Json &j, const char *key, bool dflt) { return j.has(key) && j.at(key).kind == Json::Bool ? j.at(key).b : dflt; }
std::string gets(const Json &j, const char *key, const char *dflt) { return j.has(key) && j.at(key).kind == Json::Str ? j.at(key).str : dflt; }

uint64_t fnv1a(const void *data, size_t n)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

// ================================================================= Camera ================
Camera::Camera() { std::memset(static_cast<gpt_camera *>(this), 0, sizeof(gpt_camera)); medium = -1; }

Camera::Camera(float3_t pos, float3_t uu, float3_t vv, float3_t ww, float2_t res, float dis, float angle, float radius,
               float focal, bool filmic_, int medium_)
{
    std::memset(static_cast<gpt_camera *>(this), 0, sizeof(gpt_camera));
    position = pos; u = uu; v = vv; w = ww;
    resolution = res; distance = dis; fov = angle;
    apertureRadius = radius; focalDistance = focal;
    filmic = filmic_ ? 1 : 0;
    medium = medium_;
    float half_fov = fov * .5f;                                           // camera.h:38-45
    float rad = (float)(half_fov / 180.0 * pt::PI);                      // DegreesToRadians, common.h:46-49
    height = std::tan(rad) * distance;
    width = height * resolution.x / resolution.y;
    area = 4.f * width * height;
    pixel2screen.x = 2.f * width / resolution.x;
    pixel2screen.y = 2.f * height / resolution.y;
    ratio = focalDistance / distance;
}

void Camera::Lookat(const float3_t &eye_pos, const float3_t &dest, const float3_t &up)   // camera.h:123-128
{
    position = eye_pos;
    V3 ww = pt::normalize(v3of(eye_pos) - v3of(dest));
    V3 uu = pt::normalize(pt::cross(v3of(up), ww));
    V3 vv = pt::normalize(pt::cross(ww, uu));
    w = g3(ww); u = g3(uu); v = g3(vv);
}

// ================================================================= BVH ====================
BVH::~BVH() { delete[] linear_root; }

void BVH::Build(std::vector<Primitive> &primitives)
{
    delete[] linear_root;
    linear_root = nullptr;
    total_nodes = 0;
    const int n = (int)primitives.size();
    prims.assign((size_t)n, Primitive());
    if (n == 0) return;
    linear_root = new LinearBVHNode[(size_t)2 * n];
    float box[6];
    int32_t nn = 0;
    if (gpt_bvh_build(primitives.data(), n, prims.data(), linear_root, &nn, box) != GPT_OK) {
        prims.clear();
        return;
    }
    total_nodes = nn;
    root_box.fmin = gpt_float3{box[0], box[1], box[2]};
    root_box.fmax = gpt_float3{box[3], box[4], box[5]};
    primitives.clear();   // bvh.cpp:30
}

void BVH::BuildSplit(std::vector<Primitive> &primitives, float alpha)
{
    delete[] linear_root;
    linear_root = nullptr;
    total_nodes = 0;
    const int n = (int)primitives.size();
    prims.clear();
    prim_origin.clear();
    if (n == 0) return;
    const int cap = 2 * n + 64;
    prims.assign((size_t)cap, Primitive());
    prim_origin.assign((size_t)cap, 0);
    linear_root = new LinearBVHNode[(size_t)2 * cap];
    float box[6];
    int32_t nn = 0, np = 0;
    if (gpt_sbvh_build(primitives.data(), n, alpha, prims.data(), cap, &np, prim_origin.data(), linear_root, 2 * cap, &nn, box) != GPT_OK) {
        // no split tree (a non-finite vertex, or the duplicate budget ran out): the reference's builder takes the scene instead -
        // and refuses what it must refuse with its own message - rather than an empty scene that renders the background
        prim_origin.clear();
        Build(primitives);
        return;
    }
    prims.resize((size_t)np);
    prim_origin.resize((size_t)np);
    total_nodes = nn;
    root_box.fmin = gpt_float3{box[0], box[1], box[2]};
    root_box.fmax = gpt_float3{box[3], box[4], box[5]};
    primitives.clear();
}

// bvh.cache: int total_nodes, int nprims, float[3] min, float[3] max, Primitive[nprims], LinearBVHNode[total_nodes]
// (src/bvh.cpp:189-218), followed here by an 8-byte FNV-1a hash of the INPUT primitives.  The reference keys
// the cache by directory only and silently reuses a stale one; here a cache is used only when the primitive
// count matches and, if the trailing hash is present, the hash matches too.
void BVH::LoadOrBuildBVH(std::vector<Primitive> &primitives, std::string file)
{
    const std::string base = file.substr(0, file.find_last_of('/') + 1);
    const std::string bvhfile = base + "bvh.cache";
    const uint64_t want = fnv1a(primitives.data(), primitives.size() * sizeof(Primitive));
    FILE *fp = std::fopen(bvhfile.c_str(), "rb");
    if (fp) {
        int nodes = 0, size = 0;
        float mn[3], mx[3];
        bool ok = std::fread(&nodes, sizeof(int), 1, fp) == 1 && std::fread(&size, sizeof(int), 1, fp) == 1 &&
                  std::fread(mn, sizeof(float) * 3, 1, fp) == 1 && std::fread(mx, sizeof(float) * 3, 1, fp) == 1;
        ok = ok && size == (int)primitives.size() && nodes > 0 && nodes <= 2 * size;
        std::vector<Primitive> p;
        LinearBVHNode *lr = nullptr;
        if (ok) {
            p.resize((size_t)size);
            lr = new LinearBVHNode[(size_t)nodes];
            ok = std::fread(p.data(), sizeof(Primitive), (size_t)size, fp) == (size_t)size &&
                 std::fread(lr, sizeof(LinearBVHNode), (size_t)nodes, fp) == (size_t)nodes;
            uint64_t have = 0;
            if (ok && std::fread(&have, sizeof(have), 1, fp) == 1) ok = have == want;
        }
        std::fclose(fp);
        if (ok) {
            delete[] linear_root;
            linear_root = lr;
            total_nodes = nodes;
            prims.swap(p);
            root_box.fmin = gpt_float3{mn[0], mn[1], mn[2]};
            root_box.fmax = gpt_float3{mx[0], mx[1], mx[2]};
            primitives.clear();
            return;
        }
        delete[] lr;
    }
    Build(primitives);
    fp = std::fopen(bvhfile.c_str(), "wb");
    if (fp) {
        int size = (int)prims.size();
        std::fwrite(&total_nodes, sizeof(int), 1, fp);
        std::fwrite(&size, sizeof(int), 1, fp);
        std::fwrite(&root_box.fmin.x, sizeof(float) * 3, 1, fp);
        std::fwrite(&root_box.fmax.x, sizeof(float) * 3, 1, fp);
        if (size) std::fwrite(prims.data(), sizeof(Primitive), (size_t)size, fp);
        if (total_nodes) std::fwrite(linear_root, sizeof(LinearBVHNode), (size_t)total_nodes, fp);
        std::fwrite(&want, sizeof(want), 1, fp);
        std::fclose(fp);
    }
}

// ================================================================= Scene ==================
Scene::Scene()
{
    std::memset(&infinite, 0, sizeof(infinite));
    integrator.maxDepth = 5;
}

void Scene::Init(Camera *cam, std::string file)   // scene.h:50-83
{
    camera = cam;
    if (use_sbvh) bvh.BuildSplit(primitives);
    else if (use_bvh_cache) bvh.LoadOrBuildBVH(primitives, file);
    else if (reference_bvh) bvh.Build(primitives);
    else {
        // The default: the reference's tree unless it has oversized leaves.  bvh.cpp:43 makes ONE leaf of any set of primitives whose box is
        // thinner than 1e-4 (a tessellated floor, the flat faces of a subdivided cube): every ray that enters that box tests all of them
        // (the config-3 stand-in: 268 triangle tests per sample against 9 on the split tree, which has no such rule).  Same primitives, same
        // box and triangle arithmetic either way: the films agree within the 1e-4 bar (tests/test_sbvh.py).
        std::vector<Primitive> copy = primitives;
        bvh.Build(primitives);
        int largest = 0;
        for (int i = 0; i < bvh.total_nodes; ++i) {
            const LinearBVHNode &nd = bvh.linear_root[i];
            if (nd.is_leaf && nd.start >= 0 && nd.end - nd.start + 1 > largest) largest = nd.end - nd.start + 1;
        }
        if (largest > GPT_WIDE_LEAF_MAX) {
            std::printf("Bvh: a leaf of %d primitives, split tree instead\n", largest);
            bvh.BuildSplit(copy);
            primitives.clear();
        }
    }
    std::printf("Bvh total nodes:%d\n", bvh.total_nodes);
    std::printf("Scene Bounds [%.3f, %.3f, %.3f]-[%.3f, %.3f, %.3f]\n", bvh.root_box.fmin.x, bvh.root_box.fmin.y,
                bvh.root_box.fmin.z, bvh.root_box.fmax.x, bvh.root_box.fmax.y, bvh.root_box.fmax.z);
    std::fflush(stdout);            // (a caller that redirects the progress lines gets all of them before the call returns)
    if (infinite.isvalid) {
        infinite.data = infinite_data.data();
        float box[6] = {bvh.root_box.fmin.x, bvh.root_box.fmin.y, bvh.root_box.fmin.z,
                        bvh.root_box.fmax.x, bvh.root_box.fmax.y, bvh.root_box.fmax.z};
        gpt_infinite_init(&infinite, box);
    }
    lightDistribution.assign(lights.size() + 2, 0.f);
    int32_t n = 0;
    gpt_light_distribution(lights.data(), (int32_t)lights.size(), infinite.isvalid ? &infinite : nullptr,
                           lightDistribution.data(), &n);
    lightDistribution.resize((size_t)n);
}

void Scene::Describe(gpt_scene_desc &d, std::vector<gpt_texture> &tex) const
{
    std::memset(&d, 0, sizeof(d));
    d.prims = bvh.prims.data();
    d.n_prims = (int32_t)bvh.prims.size();
    d.nodes = bvh.linear_root;
    d.n_nodes = bvh.total_nodes;
    d.materials = materials.data();
    d.n_materials = (int32_t)materials.size();
    d.lights = lights.data();
    d.n_lights = (int32_t)lights.size();
    d.light_distribution = lightDistribution.data();
    d.n_light_distribution = (int32_t)lightDistribution.size();
    d.infinite = infinite.isvalid ? &infinite : nullptr;
    tex.resize(textures.size());
    for (size_t i = 0; i < textures.size(); ++i) {
        tex[i].data = textures[i].data.data();
        tex[i].width = textures[i].width;
        tex[i].height = textures[i].height;
    }
    d.textures = tex.empty() ? nullptr : tex.data();
    d.n_textures = (int32_t)tex.size();
    d.integrator_type = (int32_t)integrator.type;
    d.max_depth = integrator.maxDepth;
    d.mediums = mediums.empty() ? nullptr : mediums.data();
    d.n_mediums = (int32_t)mediums.size();
}

// ================================================================= LoadScene ==============
bool LoadScene(const char *filename, GlobalConfig &config, Scene &scene)
{
    const std::string file = filename;
    const std::string base = file.substr(0, file.find_last_of('/') + 1);
    FILE *f = std::fopen(filename, "rb");
    if (!f) {
        gpt_set_error("Scene file [\"%s\"] is not good", filename);
        return false;
    }
    std::string text;
    char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    std::fclose(f);
    Json doc;
    JsonParser jp{text.data(), text.data() + text.size(), ""};
    if (!jp.parse(doc) || doc.kind != Json::Obj) {
        gpt_set_error("Parse scene error: %s (offset %ld)", jp.err.c_str(), (long)(jp.p - text.data()));
        return false;
    }
    const float zero3[3] = {0, 0, 0}, one3[3] = {1, 1, 1};

    // ---- media (parsescene.cpp:72-137).  Stored as the reference stores them: sigmaA, sigmaS scaled, sigmaT their
    // sum, g; a heterogeneous medium also gets its density grid (a text file of nx*ny*nz floats, x fastest:
    // medium.h:235-244), the box p0..p1 it fills, 1 / its largest density, iterMax and the tracking variant.
    std::vector<std::string> mediumName;
    scene.mediums.clear();
    scene.density_grids.clear();
    if (doc.has("medium") && doc.at("medium").kind == Json::Arr)
        for (auto &m : doc.at("medium").arr) {
            mediumName.push_back(gets(m, "name", ""));
            float a[3], sc3[3];
            get3(m, "sigmaA", one3, a);
            get3(m, "sigmaS", one3, sc3);
            const float scale = getf(m, "scale", 1.f);
            gpt_medium md;
            std::memset(&md, 0, sizeof(md));
            md.g = getf(m, "g", 0.f);
            const bool hom = gets(m, "type", "homogeneous") == "homogeneous";
            md.type = hom ? GPT_MEDIUM_HOMOGENEOUS : GPT_MEDIUM_HETEROGENEOUS;
            gpt_float3 sa = {a[0] * scale, a[1] * scale, a[2] * scale}, ss = {sc3[0] * scale, sc3[1] * scale, sc3[2] * scale};
            gpt_float3 st = {sa.x + ss.x, sa.y + ss.y, sa.z + ss.z};
            md.homogeneous.sigmaA = sa;            // (same offsets in both views of the union)
            md.homogeneous.sigmaS = ss;
            md.homogeneous.sigmaT = st;
            if (!hom) {
                if (st.x != st.y || st.x != st.z) {            // parsescene.cpp:101-104 (the reference exits)
                    gpt_set_error("sigmaA and sigmaS requires uniform attenuation coefficient");
                    return false;
                }
                if (!m.has("nx") || !m.has("ny") || !m.has("nz") || !m.has("p0") || !m.has("p1") || !m.has("density")) {
                    gpt_set_error("heterogeneous medium \"%s\": nx, ny, nz, p0, p1 and density are required", mediumName.back().c_str());
                    return false;
                }
                const int nx = (int)m.at("nx").num, ny = (int)m.at("ny").num, nz = (int)m.at("nz").num;
                if (nx <= 0 || ny <= 0 || nz <= 0 || (int64_t)nx * ny * nz > (int64_t)1 << 30) {
                    gpt_set_error("heterogeneous medium \"%s\": bad grid size %d x %d x %d", mediumName.back().c_str(), nx, ny, nz);
                    return false;
                }
                md.heterogeneous.nx = nx;
                md.heterogeneous.ny = ny;
                md.heterogeneous.nz = nz;
                float p[3];
                get3(m, "p0", zero3, p);
                md.heterogeneous.p0 = {p[0], p[1], p[2]};
                get3(m, "p1", zero3, p);
                md.heterogeneous.p1 = {p[0], p[1], p[2]};
                md.heterogeneous.iterMax = m.has("iterMax") ? (int)m.at("iterMax").num : 1000;
                md.heterogeneous.evalTransmittanceType = m.has("evalTransmittanceType") ? (int)m.at("evalTransmittanceType").num : 1;
                // the density file: whitespace-separated decimal floats (the reference reads them with fscanf "%f")
                const std::string dfile = base + m.at("density").str;
                FILE *df = gpt_fopen_read(dfile.c_str());
                if (!df) {
                    gpt_set_error("density file [\"%s\"] is not good", dfile.c_str());
                    return false;
                }
                std::string dtext;
                char dbuf[65536];
                size_t dgot;
                while ((dgot = std::fread(dbuf, 1, sizeof(dbuf), df)) > 0) dtext.append(dbuf, dgot);
                std::fclose(df);
                const size_t n = (size_t)nx * ny * nz;
                scene.density_grids.emplace_back(n);
                std::vector<float> &grid = scene.density_grids.back();
                const char *cp = dtext.c_str();
                float mx = 0.f;
                for (size_t i = 0; i < n; ++i) {
                    char *endp = nullptr;
                    const float v = std::strtof(cp, &endp);
                    if (endp == cp) {
                        gpt_set_error("density file [\"%s\"] holds %zu values, %zu are needed", dfile.c_str(), i, n);
                        return false;
                    }
                    cp = endp;
                    grid[i] = v;
                    if (v > mx) mx = v;
                }
                md.heterogeneous.invMaxDensity = 1.f / mx;      // parsescene.cpp:126-131
            }
            scene.mediums.push_back(md);
        }
    // the grids are complete (no reallocation from here on): point the records at them
    for (size_t i = 0, g = 0; i < scene.mediums.size(); ++i)
        if (scene.mediums[i].type == GPT_MEDIUM_HETEROGENEOUS) scene.mediums[i].heterogeneous.density = scene.density_grids[g++].data();
    auto getMedium = [&](const std::string &m) {
        for (size_t i = 0; i < mediumName.size(); ++i) if (mediumName[i] == m) return (int)i;
        return -1;
    };

    // ---- global config (parsescene.cpp:150-181)
    if (doc.has("screen_width") && doc.has("screen_height")) {
        config.width = (int)doc.at("screen_width").num;
        config.height = (int)doc.at("screen_height").num;
    } else {
        config.width = 512;
        config.height = 512;
    }
    config.epsilon = getf(doc, "epsilon", 0.001f);
    if (!doc.has("camera")) {
        gpt_set_error("Scene file must define camera");
        return false;
    }
    {
        const Json &cam = doc.at("camera");
        float pos[3], up[3], lookat[3];
        const float up_d[3] = {0, 1, 0}, la_d[3] = {0, 0, -1};
        get3(cam, "position", zero3, pos);
        get3(cam, "up", up_d, up);
        get3(cam, "lookat", la_d, lookat);
        config.camera.environment = getb(cam, "environment", false) ? 1 : 0;
        config.camera.fov = getf(cam, "fov", 60.f);
        config.camera.Lookat(gpt_float3{pos[0], pos[1], pos[2]}, gpt_float3{lookat[0], lookat[1], lookat[2]},
                             gpt_float3{up[0], up[1], up[2]});
        config.camera.apertureRadius = getf(cam, "apertureRadius", 0.f);
        config.camera.focalDistance = getf(cam, "focalDistance", 0.f);
        config.camera_move_speed = getf(cam, "move_speed", 0.1f);
        config.camera.filmic = getb(cam, "filmicTonemap", true) ? 1 : 0;
        config.camera.medium = getMedium(gets(cam, "medium", ""));
    }

    // ---- integrator (parsescene.cpp:184-226)
    {
        const std::string in = gets(doc, "integrator", "pt");
        static const std::map<std::string, IntegratorType> types = {
            {"ao", IT_AO}, {"pt", IT_PT}, {"vpt", IT_VPT}, {"lt", IT_LT}, {"bdpt", IT_BDPT}, {"mlt", IT_MLT}, {"sppm", IT_SPPM}, {"ir", IT_IR}};
        auto it = types.find(in);
        if (it == types.end()) {
            gpt_set_error("Unsupport integrator [%s]; choose one of [ao, pt, vpt, lt, bdpt, mlt, sppm, ir]", in.c_str());
            return false;
        }
        scene.integrator.type = it->second;
        if (it->second == IT_AO) scene.integrator.maxDist = getf(doc, "maxDist", 0.5f);
        else scene.integrator.maxDepth = doc.has("maxDepth") ? (int)doc.at("maxDepth").num : 5;
    }

    // ---- materials (parsescene.cpp:231-330)
    std::vector<std::string> matName, bssrdfName;
    if (doc.has("material")) {
        const Json &mats = doc.at("material");
        if (mats.kind != Json::Arr) {
            gpt_set_error("Invalid material format");
            return false;
        }
        std::map<std::string, int> matMap = {{"lambertian", GPT_MT_LAMBERTIAN}, {"mirror", GPT_MT_MIRROR},
                                             {"dielectric", GPT_MT_DIELECTRIC}, {"roughdielectric", GPT_MT_ROUGHDIELECTRIC},
                                             {"roughconduct", GPT_MT_ROUGHCONDUCTOR}, {"substrate", GPT_MT_SUBSTRATE}};
        std::map<std::string, int> texMap;
        for (auto &m : mats.arr) {
            if (m.has("bssrdf")) {   // BSSRDF tables are not on this path; keep the name so lookups behave
                bssrdfName.push_back(gets(m, "name", ""));
                continue;
            }
            float alphaU, alphaV;
            if (m.has("alpha")) {
                alphaU = (float)m.at("alpha").num;
                alphaV = alphaU;
            } else {
                alphaU = getf(m, "alphaU", 0.01f);
                alphaV = getf(m, "alphaV", 0.01f);
            }
            if (getb(m, "remap", false)) {
                auto Remap = [](float roughness) -> float {   // parsescene.cpp:282-288
                    roughness = std::max(roughness, (float)1e-3);
                    float x = std::log(roughness);
                    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
                };
                alphaU = Remap(alphaU);
                alphaV = Remap(alphaV);
            }
            Material mat;
            std::memset(&mat, 0, sizeof(mat));
            mat.type = matMap[gets(m, "bsdf", "")];   // unknown names map to 0 like std::map::operator[]
            mat.alphaU = alphaU;
            mat.alphaV = alphaV;
            mat.insideIOR = getf(m, "insideIOR", 1.f);
            mat.outsideIOR = getf(m, "outsideIOR", 1.f);
            float t[3];
            get3(m, "k", zero3, t); mat.k = gpt_float3{t[0], t[1], t[2]};
            get3(m, "eta", zero3, t); mat.eta = gpt_float3{t[0], t[1], t[2]};
            get3(m, "specular", one3, t); mat.specular = gpt_float3{t[0], t[1], t[2]};
            mat.diffuse = gpt_float3{1.f, 1.f, 1.f};
            mat.textureIdx = -1;
            if (m.has("diffuse")) {
                if (m.at("diffuse").kind == Json::Str) {
                    const std::string tf = m.at("diffuse").str;
                    if (texMap.find(tf) == texMap.end()) {
                        Texture tex;
                        if (!imageio::load_texture((base + tf).c_str(), tex.width, tex.height, tex.data)) {
                            FILE *probe = gpt_fopen_read((base + tf).c_str());
                            if (probe) {
                                std::fclose(probe);
                                gpt_set_error("Error when load texture [%s] (supported: 8-bit non-interlaced PNG, baseline and progressive JPEG)", (base + tf).c_str());
                            } else {
                                gpt_set_error("Error when load texture [%s]: the file cannot be opened", (base + tf).c_str());
                            }
                            return false;
                        }
                        scene.textures.push_back(std::move(tex));
                        texMap[tf] = (int)scene.textures.size() - 1;
                    }
                    mat.textureIdx = texMap[tf];
                } else {
                    get3(m, "diffuse", one3, t);
                    mat.diffuse = gpt_float3{t[0], t[1], t[2]};
                }
            }
            scene.materials.push_back(mat);
            matName.push_back(gets(m, "name", ""));
        }
    }
    auto find_material = [&](const std::string &name, int &matIdx, int &bssrdfIdx) {   // first match wins
        for (size_t i = 0; i < matName.size(); ++i) if (matName[i] == name) { matIdx = (int)i; return true; }
        for (size_t i = 0; i < bssrdfName.size(); ++i) if (bssrdfName[i] == name) { bssrdfIdx = (int)i; return true; }
        gpt_set_error("There is no material named:[\"%s\"]", name.c_str());
        return false;
    };

    // ---- scene (parsescene.cpp:333-489)
    if (doc.has("scene") && doc.at("scene").kind == Json::Arr) {
        for (auto &unit : doc.at("scene").arr) {
            if (unit.has("mesh")) {
                float sc[3], tr[3], ro[3];
                get3(unit, "scale", one3, sc);
                get3(unit, "translate", zero3, tr);
                get3(unit, "rotate", zero3, ro);
                const std::string mat_name = gets(unit, "material", "");
                const int mi = getMedium(gets(unit, "inside", "")), mo = getMedium(gets(unit, "outside", ""));
                int matIdx = -1, bssrdfIdx = -1;
                if (mat_name != "" || !(mi != -1 || mo != -1))
                    if (!find_material(mat_name, matIdx, bssrdfIdx)) return false;
                std::vector<Triangle> tris;
                if (!load_mesh(base + unit.at("mesh").str, trs_matrix(sc, tr, ro), matIdx, bssrdfIdx, tris)) return false;
                for (auto &t : tris) {
                    Primitive p;
                    std::memset(&p, 0, sizeof(p));
                    p.type = GPT_GT_TRIANGLE;
                    p.triangle = t;
                    p.triangle.mediumInside = mi;
                    p.triangle.mediumOutside = mo;
                    scene.primitives.push_back(p);
                }
            } else if (unit.has("line") || unit.has("sphere")) {
                gpt_set_error("scene unit \"%s\": line and sphere primitives are outside the triangle path tracer",
                              unit.has("line") ? "line" : "sphere");
                return false;
            } else {
                gpt_set_error("Error scene file format");
                return false;
            }
        }
    } else {
        std::fprintf(stderr, "There is no primitives in the scene\n");
    }

    // ---- lights (parsescene.cpp:492-586)
    if (doc.has("light") && doc.at("light").kind == Json::Arr) {
        for (auto &unit : doc.at("light").arr) {
            if (unit.has("mesh")) {
                float sc[3], tr[3], ro[3], rad[3];
                get3(unit, "scale", one3, sc);
                get3(unit, "translate", zero3, tr);
                get3(unit, "rotate", zero3, ro);
                get3(unit, "radiance", zero3, rad);
                int matIdx = -1, bssrdfIdx = -1;
                const std::string mat_name = gets(unit, "material", "matte");
                bool found = false;
                for (size_t i = 0; i < matName.size(); ++i) if (matName[i] == mat_name) { matIdx = (int)i; found = true; break; }
                if (!found) {
                    gpt_set_error("There is no material named:[\"%s\"]", mat_name.c_str());
                    return false;
                }
                std::vector<Triangle> tris;
                if (!load_mesh(base + unit.at("mesh").str, trs_matrix(sc, tr, ro), matIdx, bssrdfIdx, tris)) return false;
                for (auto &t : tris) {
                    t.lightIdx = (int)scene.lights.size();
                    Primitive p;
                    std::memset(&p, 0, sizeof(p));
                    p.type = GPT_GT_TRIANGLE;
                    p.triangle = t;
                    scene.primitives.push_back(p);
                    Area area;
                    std::memset(&area, 0, sizeof(area));
                    area.radiance = gpt_float3{rad[0], rad[1], rad[2]};
                    area.triangle = t;
                    area.medium = getMedium(gets(unit, "medium", ""));
                    scene.lights.push_back(area);
                }
            } else if (unit.has("infinite")) {
                const std::string ef = unit.at("infinite").str;
                int w = 0, h = 0;
                const bool is_exr = ef.size() >= 4 && ef.compare(ef.size() - 4, 4, ".exr") == 0;
                const bool ok = is_exr ? imageio::read_exr_rgb_top_down((base + ef).c_str(), w, h, scene.infinite_data)
                                       : imageio::read_pfm_top_down((base + ef).c_str(), w, h, scene.infinite_data);
                if (!ok) {
                    gpt_set_error("Couldn't load hdr file \"%s\" (supported: scanline .exr with NONE/RLE/ZIPS/ZIP compression, .pfm)", ef.c_str());
                    return false;
                }
                // the reference leaves u,v,w unset without "rotate"/"matrix"; identity axes here
                float uu[4] = {1, 0, 0, 0}, vv[4] = {0, 1, 0, 0}, ww[4] = {0, 0, 1, 0};
                auto apply = [&](const M4 &rs) {
                    const float ex[4] = {1, 0, 0, 0}, ey[4] = {0, 1, 0, 0}, ez[4] = {0, 0, 1, 0};
                    mulv(rs, ex, uu); mulv(rs, ey, vv); mulv(rs, ez, ww);
                };
                if (unit.has("rotate")) {
                    float r[3];
                    get3(unit, "rotate", zero3, r);
                    M4 rs = rotate(identity(), radians(r[0]), 1, 0, 0);
                    rs = rotate(rs, radians(r[1]), 0, 1, 0);
                    rs = rotate(rs, radians(r[2]), 0, 0, 1);
                    apply(rs);
                }
                if (unit.has("matrix") && unit.at("matrix").arr.size() == 16) {
                    M4 rs;
                    for (int i = 0; i < 16; ++i) (&rs.c[0][0])[i] = (float)unit.at("matrix").arr[(size_t)i].num;
                    apply(inverse(rs));
                }
                std::memset(&scene.infinite, 0, sizeof(scene.infinite));
                scene.infinite.u = gpt_float3{uu[0], uu[1], uu[2]};
                scene.infinite.v = gpt_float3{vv[0], vv[1], vv[2]};
                scene.infinite.w = gpt_float3{ww[0], ww[1], ww[2]};
                scene.infinite.width = w;
                scene.infinite.height = h;
                scene.infinite.data = scene.infinite_data.data();
                scene.infinite.isvalid = 1;
            } else {
                std::fprintf(stderr, "Only support area and infinite light\n");
            }
        }
    }
    return true;
}

// Test hook: the scene-file reader on its own.  text: a JSON document (NUL-terminated).  Returns -1 when the reader refuses
// it; otherwise, for an array of numbers, how many there are (their values in out[0..cap)), and 0 for any other document.
extern "C" __attribute__((visibility("default"))) int gpt_debug_json_numbers(const char *text, double *out, int cap)
{
    if (!text) return -1;
    Json doc;
    JsonParser jp{text, text + std::strlen(text), ""};
    if (!jp.parse(doc)) return -1;
    if (doc.kind != Json::Arr) return 0;
    int n = 0;
    for (const Json &v : doc.arr) {
        if (v.kind != Json::Num) return 0;
        if (out && n < cap) out[n] = v.num;
        ++n;
    }
    return n;
}

