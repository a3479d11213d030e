// gltf_loader.hpp — dependency-free glTF 2.0 / GLB reader → flat scene for the C-ABI.
//
// Replaces, for the path tracer, what the reference does with fastgltf 0.8.0 (not available here):
//   gltf::Loader::loadAsset   src/pbr_engine/gltf/pbr/gltf/Loader.cpp:10-32   options LoadExternalBuffers |
//                             DecomposeNodeMatrices | GenerateMeshIndices (:19-21)
//   Asset::loadPrimitive      src/pbr_engine/gltf/pbr/gltf/Asset.cpp:162-208   POSITION, NORMAL, TANGENT, TEXCOORD_0, indices
//   Asset::loadMaterial       Asset.cpp:135-160                                  baseColorFactor, baseColorTexture, normalTexture
//                                                                                (+ here metallic, roughness, emissive, metallicRoughnessTexture)
//   Asset::loadImage2D        Asset.cpp:121-133, ImageDataSourceVisitor :58-101  images from a bufferView or a URI, decoded to RGBA8
//                                                                                (LoadImage.cpp:56-73); samplers are default-constructed
//                                                                                (:116-117: NEAREST, REPEAT) whatever the asset says
//   Asset::loadNode/loadScene Asset.cpp:236-273                                  TRS per node, quaternion (w,x,y,z)
// Deliberate deviations: indices are u32 (reference truncates to u16); missing NORMAL / TANGENT /
// TEXCOORD_0 / indices / material get defaults instead of throwing; parent transforms are composed (the reference
// draws each node with its local TRS — `compose_parents = false` reproduces that); objects are keyed by index.
#pragma once
#include <ptc.h>

#include "image_io.hpp"
#include "jpeg_decode.hpp"
#include "misc_decode.hpp"
#include "png_decode.hpp"

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pbr::gltf {

// ---- minimal JSON ---------------------------------------------------------------------------------
struct JValue {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;

  const JValue* get(const std::string& key) const {
    if (type != Obj) return nullptr;
    for (auto const& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
  double number(const std::string& key, double def) const { const JValue* v = get(key); return v && v->type == Num ? v->num : def; }
  long integer(const std::string& key, long def) const { const JValue* v = get(key); return v && v->type == Num ? (long)v->num : def; }
  std::string string(const std::string& key, const std::string& def = "") const { const JValue* v = get(key); return v && v->type == Str ? v->str : def; }
  const std::vector<JValue>& array(const std::string& key) const {
    static const std::vector<JValue> empty;
    const JValue* v = get(key);
    return v && v->type == Arr ? v->arr : empty;
  }
};

class JParser {
public:
  explicit JParser(const std::string& text) : s(text) {}
  JValue parse() { JValue v = value(); ws(); if (p != s.size()) fail("trailing characters"); return v; }

private:
  const std::string& s;
  size_t p = 0;
  int depth = 0;     // nesting of the value being parsed: bounded, the parser is recursive
  struct Nest { int& d; explicit Nest(int& x) : d(x) { ++d; } ~Nest() { --d; } };
  [[noreturn]] void fail(const std::string& m) const { throw std::runtime_error("JSON: " + m + " at offset " + std::to_string(p)); }
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
  bool lit(const char* w) { size_t n = std::strlen(w); if (s.compare(p, n, w) == 0) { p += n; return true; } return false; }
  JValue value() {
    const Nest nest(depth);
    if (depth > 128) fail("nesting deeper than 128");
    ws();
    if (p >= s.size()) fail("unexpected end");
    JValue v;
    const char c = s[p];
    if (c == '{') {
      v.type = JValue::Obj; ++p; ws();
      if (p < s.size() && s[p] == '}') { ++p; return v; }
      for (;;) {
        ws();
        if (p >= s.size() || s[p] != '"') fail("expected key");
        std::string k = str();
        ws();
        if (p >= s.size() || s[p] != ':') fail("expected ':'");
        ++p;
        v.obj.emplace_back(std::move(k), value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == '}') { ++p; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.type = JValue::Arr; ++p; ws();
      if (p < s.size() && s[p] == ']') { ++p; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == ']') { ++p; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') { v.type = JValue::Str; v.str = str(); }
    else if (lit("true")) { v.type = JValue::Bool; v.b = true; }
    else if (lit("false")) { v.type = JValue::Bool; v.b = false; }
    else if (lit("null")) { v.type = JValue::Null; }
    else {
      size_t e = p;
      while (e < s.size() && (std::isdigit((unsigned char)s[e]) || s[e] == '-' || s[e] == '+' || s[e] == '.' || s[e] == 'e' || s[e] == 'E')) ++e;
      if (e == p) fail("unexpected character");
      v.type = JValue::Num;
      v.num = std::strtod(s.substr(p, e - p).c_str(), nullptr);
      p = e;
    }
    return v;
  }
  std::string str() {
    std::string o;
    ++p;
    while (p < s.size() && s[p] != '"') {
      char c = s[p++];
      if (c == '\\') {
        if (p >= s.size()) fail("bad escape");
        c = s[p++];
        switch (c) {
          case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
          case 'b': o += '\b'; break; case 'f': o += '\f'; break;
          case 'u': {
            if (p + 4 > s.size()) fail("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) o += (char)cp;
            else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
            else { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: o += c;
        }
      } else o += c;
    }
    if (p >= s.size()) fail("unterminated string");
    ++p;
    return o;
  }
};

// ---- flat result ------------------------------------------------------------------------------------
struct Material { float base_color[4] = {1, 1, 1, 1}; float metallic = 1.0f, roughness = 1.0f; float emissive[3] = {0, 0, 0}; int tex_color = -1, tex_normal = -1, tex_mr = -1; };   // tex_*: index into FlatScene::textures
struct Texture { int w = 0, h = 0; std::vector<std::uint8_t> rgba; };
struct Primitive { std::vector<ptc_vertex> vertices; std::vector<std::uint32_t> indices; int material = 0; };
struct Instance { int primitive; std::array<float, 16> model; };   // column-major
struct FlatScene {
  std::vector<Texture> textures;        // decoded images, one per glTF image that a material references (first-use order)
  std::vector<Material> materials;      // glTF materials in index order, plus a trailing default one if any primitive needs it
  std::vector<Primitive> primitives;    // one per glTF mesh primitive, mesh order then primitive order
  std::vector<Instance> instances;      // scene traversal order (depth-first, children before the node's own mesh — Scene.cpp:77-82)
  float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};   // world-space bounds of all instanced vertices
  std::uint64_t n_triangles = 0;
};

namespace detail {

inline std::vector<std::uint8_t> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  return std::vector<std::uint8_t>(s.begin(), s.end());
}
// A relative-reference URI of a buffer or an image (glTF 2.0 §2.8 / RFC 3986): percent-decoded, and resolved against the asset's directory
// only when it stays inside it — no scheme ("file:", "http:"), no absolute path, no ".." segment that climbs out, no NUL.  An asset is
// untrusted input: "../../etc/passwd" or "%2e%2e/x" must not turn the loader into a file reader for the rest of the disk.
inline std::string resolve_uri(const std::string& dir, const std::string& uri) {
  std::string dec;
  for (std::size_t i = 0; i < uri.size(); ++i) {
    if (uri[i] == '%') {
      auto hex = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
      if (i + 2 >= uri.size()) throw std::runtime_error("uri: truncated percent escape in " + uri);
      const int h = hex(uri[i + 1]), l = hex(uri[i + 2]);
      if (h < 0 || l < 0) throw std::runtime_error("uri: bad percent escape in " + uri);
      dec.push_back((char)(h * 16 + l));
      i += 2;
    } else dec.push_back(uri[i]);
  }
  if (dec.empty() || dec.find('\0') != std::string::npos) throw std::runtime_error("uri: empty or contains NUL: " + uri);
  if (dec[0] == '/' || dec[0] == '\\' || dec.find(':') != std::string::npos) throw std::runtime_error("uri: only relative references inside the asset's directory are read: " + uri);
  int depth = 0;                                    // segments below the asset's directory
  std::size_t a = 0;
  while (a <= dec.size()) {
    std::size_t b = dec.find_first_of("/\\", a);
    if (b == std::string::npos) b = dec.size();
    const std::string seg = dec.substr(a, b - a);
    if (seg == "..") { if (--depth < 0) throw std::runtime_error("uri: leaves the asset's directory: " + uri); }
    else if (!seg.empty() && seg != ".") ++depth;
    a = b + 1;
  }
  return dir + dec;
}
inline std::vector<std::uint8_t> base64(const std::string& in) {
  std::vector<std::uint8_t> out;
  unsigned acc = 0; int bits = 0;
  for (char c : in) {
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+') v = 62; else if (c == '/') v = 63; else continue;
    acc = (acc << 6) | (unsigned)v; bits += 6;
    if (bits >= 8) { bits -= 8; out.push_back((std::uint8_t)((acc >> bits) & 0xFF)); }
  }
  return out;
}
inline std::string dir_of(const std::string& path) { const size_t k = path.find_last_of("/\\"); return k == std::string::npos ? std::string() : path.substr(0, k + 1); }

// column-major 4x4 helpers, plain binary32 (glm-style operation order)
using Mat4 = std::array<float, 16>;
inline Mat4 identity() { Mat4 m{}; m[0] = m[5] = m[10] = m[15] = 1.0f; return m; }
inline Mat4 mul(const Mat4& a, const Mat4& b) {   // a·b: column j = a[0]*b[j].x + a[1]*b[j].y + a[2]*b[j].z + a[3]*b[j].w
  Mat4 r{};
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
      r[j * 4 + i] = a[0 * 4 + i] * b[j * 4 + 0] + a[1 * 4 + i] * b[j * 4 + 1] + a[2 * 4 + i] * b[j * 4 + 2] + a[3 * 4 + i] * b[j * 4 + 3];
  return r;
}
inline Mat4 from_trs(const float t[3], const float q_wxyz[4], const float s[3]) {   // same arithmetic as ptc_add_instance
  const float w = q_wxyz[0], x = q_wxyz[1], y = q_wxyz[2], z = q_wxyz[3];
  const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  float R[9];
  R[0] = 1.0f - 2.0f * (yy + zz); R[1] = 2.0f * (xy + wz);        R[2] = 2.0f * (xz - wy);
  R[3] = 2.0f * (xy - wz);        R[4] = 1.0f - 2.0f * (xx + zz); R[5] = 2.0f * (yz + wx);
  R[6] = 2.0f * (xz + wy);        R[7] = 2.0f * (yz - wx);        R[8] = 1.0f - 2.0f * (xx + yy);
  Mat4 m{};
  for (int c = 0; c < 3; ++c) { for (int k = 0; k < 3; ++k) m[c * 4 + k] = R[c * 3 + k] * s[c]; m[c * 4 + 3] = 0.0f; }
  m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; m[15] = 1.0f;
  return m;
}

struct Doc {
  JValue root;
  std::vector<std::vector<std::uint8_t>> buffers;

  static size_t comp_size(long ct) {
    switch (ct) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; }
    throw std::runtime_error("unsupported accessor componentType " + std::to_string(ct));
  }
  static int type_count(const std::string& t) {
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4" || t == "MAT2") return 4;
    if (t == "MAT3") return 9;
    if (t == "MAT4") return 16;
    throw std::runtime_error("unsupported accessor type " + t);
  }
  struct View { const std::uint8_t* base; size_t stride, count; long ct; int ncomp; bool normalized; };
  mutable std::map<long, std::vector<std::uint8_t>> dense;     // sparse accessors, materialised (tightly packed) on first use
  // `n` bytes at byteOffset `off` of bufferView `bv`, bounds-checked (sparse indices / values: tightly packed, no stride)
  const std::uint8_t* view_bytes(long bv, long off, size_t n) const {
    const auto& bvs = root.array("bufferViews");
    if (bv < 0 || (size_t)bv >= bvs.size() || off < 0) throw std::runtime_error("sparse accessor: bufferView index or offset out of range");
    const JValue& v = bvs[(size_t)bv];
    const long buf = v.integer("buffer", -1), off_v = v.integer("byteOffset", 0), len_v = v.integer("byteLength", 0);
    if (buf < 0 || (size_t)buf >= buffers.size() || off_v < 0 || len_v < 0) throw std::runtime_error("buffer index out of range");
    const auto& data = buffers[(size_t)buf];
    const size_t o = (size_t)off_v + (size_t)off;
    if ((size_t)off > (size_t)len_v || n > (size_t)len_v - (size_t)off || o > data.size() || n > data.size() - o) throw std::runtime_error("sparse accessor exceeds its bufferView");
    return data.data() + o;
  }
  // glTF 2.0 §3.6.2.3: a sparse accessor is its bufferView's elements (zeros without one) with `sparse.count` of them replaced: element
  // indices[i] (strictly increasing, < count) takes values[i]
  View sparse_view(long accessor, const JValue& a) const {
    View r;
    r.ct = a.integer("componentType", 0);
    r.ncomp = type_count(a.string("type"));
    const long count = a.integer("count", 0);
    if (count < 0) throw std::runtime_error("accessor with a negative count, stride or offset");
    r.count = (size_t)count;
    r.normalized = a.get("normalized") && a.get("normalized")->b;
    const size_t elem = comp_size(r.ct) * (size_t)r.ncomp;
    r.stride = elem;
    auto it = dense.find(accessor);
    if (it == dense.end()) {
      if (r.count > (size_t)1 << 28) throw std::runtime_error("sparse accessor: too many elements");
      std::vector<std::uint8_t> d(r.count * elem, 0);
      if (a.integer("bufferView", -1) >= 0) {
        JValue base = a;                              // the same accessor without its sparse part
        for (size_t k = 0; k < base.obj.size(); ++k) if (base.obj[k].first == "sparse") { base.obj.erase(base.obj.begin() + (long)k); break; }
        const View b = view_of(accessor, base);
        for (size_t i = 0; i < r.count; ++i) std::memcpy(&d[i * elem], b.base + i * b.stride, elem);
      }
      const JValue& sp = *a.get("sparse");
      const long n = sp.integer("count", 0);
      const JValue* ind = sp.get("indices"); const JValue* val = sp.get("values");
      if (n < 0 || !ind || !val) throw std::runtime_error("sparse accessor: bad count, or indices / values missing");
      const long ict = ind->integer("componentType", 0);
      if (ict != 5121 && ict != 5123 && ict != 5125) throw std::runtime_error("sparse accessor: indices must be UNSIGNED_BYTE / SHORT / INT");
      const size_t isz = comp_size(ict);
      if ((size_t)n > r.count) throw std::runtime_error("sparse accessor: more replacements than elements");
      const std::uint8_t* ip = view_bytes(ind->integer("bufferView", -1), ind->integer("byteOffset", 0), (size_t)n * isz);
      const std::uint8_t* vp = view_bytes(val->integer("bufferView", -1), val->integer("byteOffset", 0), (size_t)n * elem);
      long prev = -1;
      for (long i = 0; i < n; ++i) {
        std::uint32_t k = 0;
        if (isz == 1) k = ip[i]; else if (isz == 2) { std::uint16_t u; std::memcpy(&u, ip + 2 * i, 2); k = u; } else std::memcpy(&k, ip + 4 * i, 4);
        if ((long)k <= prev || k >= r.count) throw std::runtime_error("sparse accessor: indices must be strictly increasing and below count");
        prev = (long)k;
        std::memcpy(&d[(size_t)k * elem], vp + (size_t)i * elem, elem);
      }
      it = dense.emplace(accessor, std::move(d)).first;
    }
    r.base = it->second.data();
    return r;
  }
  View view(long accessor) const {
    const auto& accs = root.array("accessors");
    if (accessor < 0 || (size_t)accessor >= accs.size()) throw std::runtime_error("accessor index out of range");
    const JValue& a = accs[(size_t)accessor];
    if (a.get("sparse")) return sparse_view(accessor, a);
    return view_of(accessor, a);
  }
  View view_of(long accessor, const JValue& a) const {
    (void)accessor;
    const long bv = a.integer("bufferView", -1);
    if (bv < 0) throw std::runtime_error("accessor without bufferView");
    const auto& bvs = root.array("bufferViews");
    if ((size_t)bv >= bvs.size()) throw std::runtime_error("bufferView index out of range");
    const JValue& v = bvs[(size_t)bv];
    const long buf = v.integer("buffer", -1);
    if (buf < 0 || (size_t)buf >= buffers.size()) throw std::runtime_error("buffer index out of range");
    View r;
    r.ct = a.integer("componentType", 0);
    r.ncomp = type_count(a.string("type"));
    const long count = a.integer("count", 0), stride = v.integer("byteStride", 0), off_v = v.integer("byteOffset", 0), off_a = a.integer("byteOffset", 0);
    if (count < 0 || stride < 0 || off_v < 0 || off_a < 0) throw std::runtime_error("accessor with a negative count, stride or offset");
    r.count = (size_t)count;
    r.normalized = a.get("normalized") && a.get("normalized")->b;
    const size_t elem = comp_size(r.ct) * (size_t)r.ncomp;
    r.stride = (size_t)stride;
    if (r.stride == 0) r.stride = elem;
    const auto& data = buffers[(size_t)buf];
    const size_t off = (size_t)off_v + (size_t)off_a;   // both < 2^63: no wrap
    // overflow-safe form of  off + (count-1)*stride + elem <= size
    if (r.count && (off > data.size() || elem > data.size() - off || (r.count - 1) > (data.size() - off - elem) / r.stride)) throw std::runtime_error("accessor exceeds its buffer");
    r.base = data.data() + off;
    return r;
  }
  static float component(const View& v, const std::uint8_t* p, int k) {
    switch (v.ct) {
      case 5126: { float f; std::memcpy(&f, p + 4 * k, 4); return f; }
      case 5121: { const std::uint8_t u = p[k]; return v.normalized ? (float)u / 255.0f : (float)u; }
      case 5120: { std::int8_t i; std::memcpy(&i, p + k, 1); return v.normalized ? std::fmax((float)i / 127.0f, -1.0f) : (float)i; }
      case 5123: { std::uint16_t u; std::memcpy(&u, p + 2 * k, 2); return v.normalized ? (float)u / 65535.0f : (float)u; }
      case 5122: { std::int16_t i; std::memcpy(&i, p + 2 * k, 2); return v.normalized ? std::fmax((float)i / 32767.0f, -1.0f) : (float)i; }
      case 5125: { std::uint32_t u; std::memcpy(&u, p + 4 * k, 4); return (float)u; }
    }
    throw std::runtime_error("unsupported componentType");
  }
  // n-component float read of an accessor (fastgltf::iterateAccessorWithIndex<glm::vecN>)
  std::vector<float> floats(long accessor, int want) const {
    const View v = view(accessor);
    if (v.ncomp < want) throw std::runtime_error("accessor has too few components");
    std::vector<float> out(v.count * (size_t)want);
    for (size_t i = 0; i < v.count; ++i)
      for (int k = 0; k < want; ++k) out[i * (size_t)want + (size_t)k] = component(v, v.base + i * v.stride, k);
    return out;
  }
  std::vector<std::uint32_t> indices(long accessor) const {
    const View v = view(accessor);
    std::vector<std::uint32_t> out(v.count);
    for (size_t i = 0; i < v.count; ++i) {
      const std::uint8_t* p = v.base + i * v.stride;
      switch (v.ct) {
        case 5121: out[i] = p[0]; break;
        case 5123: { std::uint16_t u; std::memcpy(&u, p, 2); out[i] = u; break; }
        case 5125: { std::uint32_t u; std::memcpy(&u, p, 4); out[i] = u; break; }
        default: throw std::runtime_error("index accessor must be UNSIGNED_BYTE / SHORT / INT");
      }
    }
    return out;
  }
};

inline Doc open(const std::string& path) {
  Doc d;
  const std::vector<std::uint8_t> file = read_file(path);
  std::string json;
  std::vector<std::uint8_t> bin;
  bool have_bin = false;
  if (file.size() >= 12 && std::memcmp(file.data(), "glTF", 4) == 0) {   // GLB container
    std::uint32_t version, length;
    std::memcpy(&version, &file[4], 4); std::memcpy(&length, &file[8], 4);
    if (version != 2) throw std::runtime_error("GLB version " + std::to_string(version) + " is not supported");
    if (length > file.size()) throw std::runtime_error("GLB length exceeds the file");
    size_t p = 12;
    while (p + 8 <= length) {
      std::uint32_t clen, ctype;
      std::memcpy(&clen, &file[p], 4); std::memcpy(&ctype, &file[p + 4], 4);
      p += 8;
      if (p + clen > length) throw std::runtime_error("GLB chunk exceeds the file");
      if (ctype == 0x4E4F534Au) json.assign((const char*)&file[p], clen);                      // "JSON"
      else if (ctype == 0x004E4942u && !have_bin) { bin.assign(file.begin() + (long)p, file.begin() + (long)(p + clen)); have_bin = true; }   // "BIN\0"
      p += (clen + 3u) & ~3u;
    }
    if (json.empty()) throw std::runtime_error("GLB without JSON chunk");
  } else json.assign(file.begin(), file.end());
  d.root = JParser(json).parse();
  if (d.root.type != JValue::Obj) throw std::runtime_error("glTF root is not an object");
  const JValue* asset = d.root.get("asset");
  if (!asset || asset->string("version").rfind("2", 0) != 0) throw std::runtime_error("not a glTF 2.x asset");
  const std::string dir = dir_of(path);
  size_t bi = 0;
  for (const JValue& b : d.root.array("buffers")) {   // Options::LoadExternalBuffers
    const std::string uri = b.string("uri");
    if (uri.empty()) {
      if (bi != 0 || !have_bin) throw std::runtime_error("buffer without uri and no GLB BIN chunk");
      d.buffers.push_back(bin);
    } else if (uri.rfind("data:", 0) == 0) {
      const size_t k = uri.find("base64,");
      if (k == std::string::npos) throw std::runtime_error("only base64 data: URIs are supported");
      d.buffers.push_back(base64(uri.substr(k + 7)));
    } else d.buffers.push_back(read_file(resolve_uri(dir, uri)));
    if (d.buffers.back().size() < (size_t)b.integer("byteLength", 0)) throw std::runtime_error("buffer shorter than its byteLength");
    ++bi;
  }
  return d;
}

}  // namespace detail

// Load `scene_index` (-1: the asset's default scene, else 0) of a .gltf / .glb file.
inline FlatScene load(const std::string& path, int scene_index = -1, bool compose_parents = true) {
  using namespace detail;
  const Doc d = open(path);
  FlatScene out;
  // ---- images (Asset::loadImage2D): decoded on first use by a material, cached by image index ----
  const std::string dir = dir_of(path);
  std::vector<int> image_slot(d.root.array("images").size(), -1);
  auto texture_of = [&](const JValue* info) -> int {
    if (!info) return -1;
    if (info->integer("texCoord", 0) != 0) throw std::runtime_error("only TEXCOORD_0 is supported (textureInfo.texCoord != 0)");
    const long ti = info->integer("index", -1);
    const auto& textures = d.root.array("textures");
    if (ti < 0 || (size_t)ti >= textures.size()) throw std::runtime_error("texture index out of range");
    const long ii = textures[(size_t)ti].integer("source", -1);
    if (ii < 0 || (size_t)ii >= image_slot.size()) throw std::runtime_error("texture without a valid image source");
    if (image_slot[(size_t)ii] >= 0) return image_slot[(size_t)ii];
    const JValue& img = d.root.array("images")[(size_t)ii];
    std::vector<std::uint8_t> owned;
    const std::uint8_t* bytes = nullptr; size_t nbytes = 0;
    if (const JValue* bv = img.get("bufferView")) {
      const auto& views = d.root.array("bufferViews");
      if (bv->type != JValue::Num || bv->num < 0 || (size_t)bv->num >= views.size()) throw std::runtime_error("image bufferView out of range");
      const JValue& v = views[(size_t)bv->num];
      const long bi = v.integer("buffer", -1), off = v.integer("byteOffset", 0), len = v.integer("byteLength", 0);
      if (bi < 0 || (size_t)bi >= d.buffers.size() || off < 0 || len < 0 || (size_t)off + (size_t)len > d.buffers[(size_t)bi].size()) throw std::runtime_error("image bufferView exceeds its buffer");
      bytes = d.buffers[(size_t)bi].data() + off; nbytes = (size_t)len;
    } else {
      const std::string uri = img.string("uri");
      if (uri.empty()) throw std::runtime_error("image without bufferView or uri");
      if (uri.rfind("data:", 0) == 0) {
        const size_t k = uri.find("base64,");
        if (k == std::string::npos) throw std::runtime_error("only base64 data: URIs are supported");
        owned = base64(uri.substr(k + 7));
      } else owned = read_file(resolve_uri(dir, uri));
      bytes = owned.data(); nbytes = owned.size();
    }
    Texture t;
    // the order in which the reference's decoder tries the formats it knows (stb_image.h, stbi__load_main)
    if (pbr::image::is_png(bytes, nbytes)) t.rgba = pbr::image::decode_png(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_bmp(bytes, nbytes)) t.rgba = pbr::image::decode_bmp(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_gif(bytes, nbytes)) t.rgba = pbr::image::decode_gif(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_psd(bytes, nbytes)) t.rgba = pbr::image::decode_psd(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_pic(bytes, nbytes)) t.rgba = pbr::image::decode_pic(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_jpeg(bytes, nbytes)) t.rgba = pbr::image::decode_jpeg(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_pnm(bytes, nbytes)) t.rgba = pbr::image::decode_pnm(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_hdr(bytes, nbytes)) t.rgba = pbr::image::decode_hdr_rgba8(bytes, nbytes, t.w, t.h);
    else if (pbr::image::is_tga(bytes, nbytes)) t.rgba = pbr::image::decode_tga(bytes, nbytes, t.w, t.h);
    else throw std::runtime_error("image " + std::to_string(ii) + " is none of PNG, BMP, GIF, PSD, PIC, JPEG, PGM / PPM, Radiance, TGA");
    if (t.w <= 0 || t.h <= 0) throw std::runtime_error("image " + std::to_string(ii) + " is empty");
    out.textures.push_back(std::move(t));
    return image_slot[(size_t)ii] = (int)out.textures.size() - 1;
  };
  // ---- materials (Asset::loadMaterial + the metal-rough / emissive factors the reference ignores) ----
  for (const JValue& m : d.root.array("materials")) {
    Material o;
    if (const JValue* pbr = m.get("pbrMetallicRoughness")) {
      const auto& bc = pbr->array("baseColorFactor");
      for (size_t k = 0; k < 4 && k < bc.size(); ++k) o.base_color[k] = (float)bc[k].num;
      o.metallic = (float)pbr->number("metallicFactor", 1.0);
      o.roughness = (float)pbr->number("roughnessFactor", 1.0);
    }
    const auto& em = m.array("emissiveFactor");
    float strength = 1.0f;
    if (const JValue* ext = m.get("extensions"))
      if (const JValue* es = ext->get("KHR_materials_emissive_strength")) strength = (float)es->number("emissiveStrength", 1.0);
    for (size_t k = 0; k < 3 && k < em.size(); ++k) o.emissive[k] = (float)em[k].num * strength;
    if (const JValue* pbr = m.get("pbrMetallicRoughness")) {
      o.tex_color = texture_of(pbr->get("baseColorTexture"));
      o.tex_mr = texture_of(pbr->get("metallicRoughnessTexture"));
    }
    o.tex_normal = texture_of(m.get("normalTexture"));
    out.materials.push_back(o);
  }
  int default_material = -1;
  // ---- meshes → primitives (Asset::loadPrimitive) ----
  std::vector<std::pair<int, int>> mesh_span;   // mesh → [first primitive, count)
  for (const JValue& mesh : d.root.array("meshes")) {
    const int first = (int)out.primitives.size();
    for (const JValue& prim : mesh.array("primitives")) {
      const long mode = prim.integer("mode", 4);
      if (mode != 4) throw std::runtime_error("primitive mode " + std::to_string(mode) + " is not supported (triangle lists only, like core/PipelineBuilder.hpp:20-22)");
      const JValue* attrs = prim.get("attributes");
      if (!attrs || !attrs->get("POSITION")) throw std::runtime_error("Primitive does not have POSITION attribute");
      const std::vector<float> pos = d.floats(attrs->integer("POSITION", -1), 3);
      const size_t nv = pos.size() / 3;
      Primitive P;
      P.vertices.resize(nv);
      for (size_t i = 0; i < nv; ++i) {
        ptc_vertex& v = P.vertices[i];
        std::memcpy(v.position, &pos[i * 3], 12);
        v.normal[0] = v.normal[1] = 0.0f; v.normal[2] = 1.0f;
        v.tangent[0] = 1.0f; v.tangent[1] = v.tangent[2] = 0.0f; v.tangent[3] = 1.0f;
        v.texcoord[0] = v.texcoord[1] = 0.0f;
      }
      if (prim.get("indices")) P.indices = d.indices(prim.integer("indices", -1));
      else { P.indices.resize(nv); for (size_t i = 0; i < nv; ++i) P.indices[i] = (std::uint32_t)i; }   // Options::GenerateMeshIndices
      if (P.indices.size() % 3) throw std::runtime_error("triangle list with an index count that is not a multiple of 3");
      for (std::uint32_t ix : P.indices) if (ix >= nv) throw std::runtime_error("index out of range");
      if (attrs->get("NORMAL")) {
        const std::vector<float> n = d.floats(attrs->integer("NORMAL", -1), 3);
        if (n.size() != pos.size()) throw std::runtime_error("NORMAL count differs from POSITION count");
        for (size_t i = 0; i < nv; ++i) std::memcpy(P.vertices[i].normal, &n[i * 3], 12);
      } else {   // default: area-weighted vertex normals of the indexed triangles, accumulated in index order
        std::vector<float> acc(nv * 3, 0.0f);
        for (size_t t = 0; t + 2 < P.indices.size(); t += 3) {
          const float* a = P.vertices[P.indices[t]].position; const float* b = P.vertices[P.indices[t + 1]].position; const float* c = P.vertices[P.indices[t + 2]].position;
          const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
          const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
          for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) acc[P.indices[t + (size_t)k] * 3 + (size_t)j] += n[j];
        }
        for (size_t i = 0; i < nv; ++i) {
          const float l = std::sqrt(acc[i * 3] * acc[i * 3] + acc[i * 3 + 1] * acc[i * 3 + 1] + acc[i * 3 + 2] * acc[i * 3 + 2]);
          if (l > 0.0f) for (int j = 0; j < 3; ++j) P.vertices[i].normal[j] = acc[i * 3 + (size_t)j] / l;
        }
      }
      if (attrs->get("TANGENT")) {
        const std::vector<float> t = d.floats(attrs->integer("TANGENT", -1), 4);
        if (t.size() != nv * 4) throw std::runtime_error("TANGENT count differs from POSITION count");
        for (size_t i = 0; i < nv; ++i) std::memcpy(P.vertices[i].tangent, &t[i * 4], 16);
      } else {   // default: any unit vector orthogonal to the normal, handedness +1
        for (size_t i = 0; i < nv; ++i) {
          const float* n = P.vertices[i].normal;
          float t[3];
          if (std::fabs(n[0]) > std::fabs(n[2])) { t[0] = -n[1]; t[1] = n[0]; t[2] = 0.0f; } else { t[0] = 0.0f; t[1] = -n[2]; t[2] = n[1]; }
          const float l = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
          if (l > 0.0f) { P.vertices[i].tangent[0] = t[0] / l; P.vertices[i].tangent[1] = t[1] / l; P.vertices[i].tangent[2] = t[2] / l; }
        }
      }
      if (attrs->get("TEXCOORD_0")) {
        const std::vector<float> uv = d.floats(attrs->integer("TEXCOORD_0", -1), 2);
        if (uv.size() != nv * 2) throw std::runtime_error("TEXCOORD_0 count differs from POSITION count");
        for (size_t i = 0; i < nv; ++i) std::memcpy(P.vertices[i].texcoord, &uv[i * 2], 8);
      }
      const long mat = prim.integer("material", -1);
      if (mat >= 0) {
        if ((size_t)mat >= out.materials.size()) throw std::runtime_error("material index out of range");
        P.material = (int)mat;
      } else {
        if (default_material < 0) { default_material = (int)out.materials.size(); out.materials.push_back(Material{}); }
        P.material = default_material;
      }
      out.primitives.push_back(std::move(P));
    }
    mesh_span.emplace_back(first, (int)out.primitives.size() - first);
  }
  // ---- scene graph (Asset::loadScene / loadNode) ----
  const auto& nodes = d.root.array("nodes");
  const auto& scenes = d.root.array("scenes");
  long si = scene_index >= 0 ? scene_index : d.root.integer("scene", 0);
  if (scenes.empty()) throw std::runtime_error("asset has no scenes");
  if (si < 0 || (size_t)si >= scenes.size()) throw std::runtime_error("scene index out of range");
  bool first_vertex = true;
  struct Frame { long node; Mat4 parent; int depth; };
  for (const JValue& rootIdx : scenes[(size_t)si].array("nodes")) {
    // depth-first; a node's children are emitted before its own mesh (post-order, Scene.cpp:77-82)
    std::vector<std::pair<Frame, bool>> stack;   // (frame, expanded)
    stack.push_back({Frame{(long)rootIdx.num, identity(), 0}, false});
    while (!stack.empty()) {
      auto [fr, expanded] = stack.back();
      stack.pop_back();
      if (fr.node < 0 || (size_t)fr.node >= nodes.size()) throw std::runtime_error("node index out of range");
      if (fr.depth > 256) throw std::runtime_error("node hierarchy too deep (cycle?)");
      const JValue& n = nodes[(size_t)fr.node];
      Mat4 local;
      const auto& mtx = n.array("matrix");
      if (mtx.size() == 16) { for (int k = 0; k < 16; ++k) local[(size_t)k] = (float)mtx[(size_t)k].num; }
      else {
        float t[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, s[3] = {1, 1, 1};
        const auto& T = n.array("translation"); const auto& R = n.array("rotation"); const auto& S = n.array("scale");
        for (size_t k = 0; k < 3 && k < T.size(); ++k) t[k] = (float)T[k].num;
        if (R.size() == 4) { q[0] = (float)R[3].num; q[1] = (float)R[0].num; q[2] = (float)R[1].num; q[3] = (float)R[2].num; }   // glTF (x,y,z,w) → (w,x,y,z), Asset.cpp:242
        for (size_t k = 0; k < 3 && k < S.size(); ++k) s[k] = (float)S[k].num;
        local = from_trs(t, q, s);
      }
      const Mat4 world = compose_parents ? mul(fr.parent, local) : local;
      if (!expanded) {
        stack.push_back({Frame{fr.node, fr.parent, fr.depth}, true});
        const auto& ch = n.array("children");
        for (size_t k = ch.size(); k-- > 0;) stack.push_back({Frame{(long)ch[k].num, world, fr.depth + 1}, false});
        continue;
      }
      const long mesh = n.integer("mesh", -1);
      if (mesh < 0) continue;
      if ((size_t)mesh >= mesh_span.size()) throw std::runtime_error("mesh index out of range");
      for (int k = 0; k < mesh_span[(size_t)mesh].second; ++k) {
        const int pi = mesh_span[(size_t)mesh].first + k;
        out.instances.push_back({pi, world});
        out.n_triangles += out.primitives[(size_t)pi].indices.size() / 3;
        for (const ptc_vertex& v : out.primitives[(size_t)pi].vertices)
          for (int r = 0; r < 3; ++r) {
            const float w = world[0 + (size_t)r] * v.position[0] + world[4 + (size_t)r] * v.position[1] + world[8 + (size_t)r] * v.position[2] + world[12 + (size_t)r];
            if (first_vertex) { out.bbox_lo[r] = out.bbox_hi[r] = w; }
            else { out.bbox_lo[r] = std::fmin(out.bbox_lo[r], w); out.bbox_hi[r] = std::fmax(out.bbox_hi[r], w); }
            if (r == 2) first_vertex = false;
          }
      }
    }
  }
  if (out.instances.empty()) throw std::runtime_error("scene has no mesh instances");
  return out;
}

// Issue the C-ABI calls for a loaded scene (between ptc_scene_begin and ptc_scene_commit).  Returns the first error code.
inline int upload(ptc_ctx* ctx, const FlatScene& s) {
  std::vector<int> tex_id, mat_id, mesh_id;
  for (const Texture& t : s.textures) {
    const int id = ptc_add_texture_rgba8(ctx, t.rgba.data(), t.w, t.h);
    if (id < 0) return id;
    tex_id.push_back(id);
  }
  auto tex = [&](int k) { return k < 0 ? -1 : tex_id[(size_t)k]; };
  for (const Material& m : s.materials) {
    const int id = ptc_add_material(ctx, m.base_color, m.metallic, m.roughness, m.emissive, tex(m.tex_color), tex(m.tex_normal), tex(m.tex_mr));
    if (id < 0) return id;
    mat_id.push_back(id);
  }
  for (const Primitive& p : s.primitives) {
    const int id = ptc_add_mesh(ctx, p.vertices.data(), (std::uint32_t)p.vertices.size(), p.indices.data(), (std::uint32_t)p.indices.size(), mat_id[(size_t)p.material]);
    if (id < 0) return id;
    mesh_id.push_back(id);
  }
  for (const Instance& i : s.instances) {
    const int rc = ptc_add_instance_matrix(ctx, mesh_id[(size_t)i.primitive], i.model.data());
    if (rc < 0) return rc;
  }
  return PTC_OK;
}

}  // namespace pbr::gltf
