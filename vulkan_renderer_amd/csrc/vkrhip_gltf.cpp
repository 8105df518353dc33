// vkrhip_gltf.cpp - the on-disk formats either side of the scene path, as the
// arrays the path consumes: the entities of the reference's scene files
// (vkrhip_scene_read_entities, at the end), and the node hierarchy of a glTF 2.0
// document: per node a parent index, translation / rotation / scale
// (or a matrix), the mesh it draws, and whether the chosen scene reaches it.
// The reference imports glTF through cgltf (vendor/cgltf.h, used by its mesh
// loader); what this path needs of such a file is only "nodes" and "scenes",
// read here by a small JSON walker of its own. Host code, no device work.
#include "../../include/vkr_hip_scene.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int kMaxDepth = 64;  // nesting the walker follows before it gives up

struct Cursor {
  const char *p;
  const char *end;
};

void skip_space(Cursor &c) {
  while (c.p < c.end && (*c.p == ' ' || *c.p == '\t' || *c.p == '\n' || *c.p == '\r')) ++c.p;
}
bool take(Cursor &c, char ch) {
  skip_space(c);
  if (c.p < c.end && *c.p == ch) {
    ++c.p;
    return true;
  }
  return false;
}
bool peek(Cursor &c, char ch) {
  skip_space(c);
  return c.p < c.end && *c.p == ch;
}

int hex_digit(char ch) {
  if (ch >= '0' && ch <= '9') return ch - '0';
  if (ch >= 'a' && ch <= 'f') return ch - 'a' + 10;
  if (ch >= 'A' && ch <= 'F') return ch - 'A' + 10;
  return -1;
}

// A JSON string; `out` may be null (the text is only skipped). Escapes are
// decoded, \uXXXX to UTF-8 (a lone surrogate half is kept as it is).
bool parse_string(Cursor &c, std::string *out) {
  if (!take(c, '"')) return false;
  if (out) out->clear();
  while (c.p < c.end) {
    const char ch = *c.p++;
    if (ch == '"') return true;
    if ((unsigned char)ch < 0x20) return false;
    if (ch != '\\') {
      if (out) out->push_back(ch);
      continue;
    }
    if (c.p >= c.end) return false;
    const char esc = *c.p++;
    char plain = 0;
    switch (esc) {
      case '"': plain = '"'; break;
      case '\\': plain = '\\'; break;
      case '/': plain = '/'; break;
      case 'b': plain = '\b'; break;
      case 'f': plain = '\f'; break;
      case 'n': plain = '\n'; break;
      case 'r': plain = '\r'; break;
      case 't': plain = '\t'; break;
      case 'u': {
        if (c.end - c.p < 4) return false;
        uint32_t code = 0;
        for (int k = 0; k < 4; ++k) {
          const int d = hex_digit(c.p[k]);
          if (d < 0) return false;
          code = code * 16u + (uint32_t)d;
        }
        c.p += 4;
        if (out) {
          if (code < 0x80u) {
            out->push_back((char)code);
          } else if (code < 0x800u) {
            out->push_back((char)(0xC0u | (code >> 6)));
            out->push_back((char)(0x80u | (code & 0x3Fu)));
          } else {
            out->push_back((char)(0xE0u | (code >> 12)));
            out->push_back((char)(0x80u | ((code >> 6) & 0x3Fu)));
            out->push_back((char)(0x80u | (code & 0x3Fu)));
          }
        }
        continue;
      }
      default: return false;
    }
    if (out) out->push_back(plain);
  }
  return false;
}

// A JSON number, as the nearest double (strtod on a bounded copy: the text need
// not be terminated, and only the JSON grammar is accepted).
bool parse_number(Cursor &c, double *out) {
  skip_space(c);
  const char *s = c.p;
  const char *q = s;
  if (q < c.end && *q == '-') ++q;
  const char *digits = q;
  while (q < c.end && *q >= '0' && *q <= '9') ++q;
  if (q == digits) return false;
  if (q - digits > 1 && *digits == '0') return false;  // no leading zeros
  if (q < c.end && *q == '.') {
    const char *frac = ++q;
    while (q < c.end && *q >= '0' && *q <= '9') ++q;
    if (q == frac) return false;
  }
  if (q < c.end && (*q == 'e' || *q == 'E')) {
    ++q;
    if (q < c.end && (*q == '+' || *q == '-')) ++q;
    const char *exp = q;
    while (q < c.end && *q >= '0' && *q <= '9') ++q;
    if (q == exp) return false;
  }
  const std::string text(s, q);
  char *stop = nullptr;
  const double v = std::strtod(text.c_str(), &stop);
  if (stop != text.c_str() + text.size()) return false;
  *out = v;
  c.p = q;
  return true;
}

bool skip_value(Cursor &c, int depth);

// Calls member(key, cursor) for every member; the callback consumes the value.
template <typename F>
bool parse_object(Cursor &c, int depth, F member) {
  if (depth > kMaxDepth || !take(c, '{')) return false;
  if (take(c, '}')) return true;
  std::string key;
  for (;;) {
    if (!parse_string(c, &key) || !take(c, ':')) return false;
    if (!member(key, c)) return false;
    if (take(c, ',')) continue;
    return take(c, '}');
  }
}

// Calls element(index, cursor) for every element; the callback consumes it.
template <typename F>
bool parse_array(Cursor &c, int depth, F element) {
  if (depth > kMaxDepth || !take(c, '[')) return false;
  if (take(c, ']')) return true;
  for (uint32_t index = 0;; ++index) {
    if (!element(index, c)) return false;
    if (take(c, ',')) continue;
    return take(c, ']');
  }
}

bool skip_literal(Cursor &c, const char *word) {
  const size_t n = std::strlen(word);
  if ((size_t)(c.end - c.p) < n || std::memcmp(c.p, word, n) != 0) return false;
  c.p += n;
  return true;
}

bool skip_value(Cursor &c, int depth) {
  skip_space(c);
  if (c.p >= c.end) return false;
  switch (*c.p) {
    case '{': return parse_object(c, depth + 1, [&](const std::string &, Cursor &v) { return skip_value(v, depth + 1); });
    case '[': return parse_array(c, depth + 1, [&](uint32_t, Cursor &v) { return skip_value(v, depth + 1); });
    case '"': return parse_string(c, nullptr);
    case 't': return skip_literal(c, "true");
    case 'f': return skip_literal(c, "false");
    case 'n': return skip_literal(c, "null");
    default: {
      double ignored;
      return parse_number(c, &ignored);
    }
  }
}

// An array of exactly `count` numbers, each rounded once to float.
bool parse_floats(Cursor &c, int depth, float *out, uint32_t count) {
  uint32_t seen = 0;
  const bool ok = parse_array(c, depth, [&](uint32_t index, Cursor &v) {
    double d;
    if (index >= count || !parse_number(v, &d)) return false;
    out[index] = (float)d;
    ++seen;
    return true;
  });
  return ok && seen == count;
}

// A non-negative integer index (glTF indices are JSON integers).
bool parse_index(Cursor &c, uint32_t *out) {
  double d;
  if (!parse_number(c, &d)) return false;
  if (!(d >= 0.0) || d > 4294967294.0 || d != std::floor(d)) return false;
  *out = (uint32_t)d;
  return true;
}

bool parse_indices(Cursor &c, int depth, std::vector<uint32_t> &out) {
  out.clear();
  return parse_array(c, depth, [&](uint32_t, Cursor &v) {
    uint32_t i;
    if (!parse_index(v, &i)) return false;
    out.push_back(i);
    return true;
  });
}

struct Node {
  std::vector<uint32_t> children;
  float translation[3] = {0.f, 0.f, 0.f};
  float rotation[4] = {0.f, 0.f, 0.f, 1.f};  // x, y, z, w
  float scale[3] = {1.f, 1.f, 1.f};
  float matrix[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
  bool has_matrix = false;
  int32_t mesh = -1;
};

}  // namespace

extern "C" {

int vkrhip_gltf_read_nodes(const char *json, uint64_t length, uint32_t scene, uint32_t capacity,
                           vkrhip_gltf_nodes *info, uint32_t *parent, uint8_t *in_scene, float *pos4,
                           float *quat4, float *scale4, int32_t *mesh, uint8_t *has_matrix,
                           float *matrix16) try {
  if (!json || !info) return VKRHIP_ERR_ARG;
  std::memset(info, 0, sizeof(*info));
  std::vector<Node> nodes;
  std::vector<std::vector<uint32_t>> scenes;
  bool have_default = false, have_nodes = false;
  uint32_t default_scene = 0;

  Cursor c{json, json + length};
  const bool parsed = parse_object(c, 0, [&](const std::string &key, Cursor &v) {
    if (key == "scene") {
      have_default = true;
      return parse_index(v, &default_scene);
    }
    if (key == "scenes") {
      scenes.clear();
      return parse_array(v, 1, [&](uint32_t, Cursor &sv) {
        scenes.emplace_back();
        std::vector<uint32_t> &roots = scenes.back();
        return parse_object(sv, 2, [&](const std::string &k, Cursor &mv) {
          return k == "nodes" ? parse_indices(mv, 3, roots) : skip_value(mv, 3);
        });
      });
    }
    if (key == "nodes") {
      have_nodes = true;
      nodes.clear();
      return parse_array(v, 1, [&](uint32_t, Cursor &nv) {
        nodes.emplace_back();
        Node &n = nodes.back();
        return parse_object(nv, 2, [&](const std::string &k, Cursor &mv) {
          if (k == "children") return parse_indices(mv, 3, n.children);
          if (k == "translation") return parse_floats(mv, 3, n.translation, 3);
          if (k == "rotation") return parse_floats(mv, 3, n.rotation, 4);
          if (k == "scale") return parse_floats(mv, 3, n.scale, 3);
          if (k == "matrix") {
            n.has_matrix = true;
            return parse_floats(mv, 3, n.matrix, 16);
          }
          if (k == "mesh") {
            uint32_t m;
            if (!parse_index(mv, &m) || m > 0x7fffffffu) return false;
            n.mesh = (int32_t)m;
            return true;
          }
          return skip_value(mv, 3);
        });
      });
    }
    return skip_value(v, 1);
  });
  skip_space(c);
  if (!parsed || c.p != c.end) return VKRHIP_ERR_ARG;  // not a JSON document
  (void)have_nodes;

  const uint32_t n = (uint32_t)nodes.size();
  // parents from the children lists: a node has at most one (glTF 2.0, 3.5.2)
  std::vector<uint32_t> up(n, 0xffffffffu);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t child : nodes[i].children) {
      if (child >= n || child == i || up[child] != 0xffffffffu) return VKRHIP_ERR_ARG;
      up[child] = i;
    }
  // no cycles: walking up from any node ends within n steps
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t at = i, steps = 0;
    while (up[at] != 0xffffffffu) {
      at = up[at];
      if (++steps > n) return VKRHIP_ERR_ARG;
    }
  }
  // the scene: the given one, else the document's default, else the first; a
  // document without scenes shows every node
  std::vector<uint8_t> reached(n, scenes.empty() ? 1 : 0);
  uint32_t chosen = 0;
  if (!scenes.empty()) {
    chosen = scene != 0xffffffffu ? scene : (have_default ? default_scene : 0u);
    if (chosen >= scenes.size()) return VKRHIP_ERR_ARG;
    std::vector<uint32_t> stack;
    for (uint32_t root : scenes[chosen]) {
      if (root >= n || up[root] != 0xffffffffu) return VKRHIP_ERR_ARG;  // scene nodes are roots
      stack.push_back(root);
    }
    while (!stack.empty()) {
      const uint32_t at = stack.back();
      stack.pop_back();
      if (reached[at]) continue;
      reached[at] = 1;
      for (uint32_t child : nodes[at].children) stack.push_back(child);
    }
  }
  info->node_count = n;
  info->scene_count = (uint32_t)scenes.size();
  info->scene = chosen;
  for (uint32_t i = 0; i < n; ++i) info->scene_nodes += reached[i];
  if (n > capacity) return VKRHIP_ERR_CAPACITY;  // info tells how much room the arrays need

  for (uint32_t i = 0; i < n; ++i) {
    const Node &nd = nodes[i];
    if (parent) parent[i] = up[i];
    if (in_scene) in_scene[i] = reached[i];
    if (pos4) {
      float *o = pos4 + 4 * (size_t)i;
      o[0] = nd.translation[0], o[1] = nd.translation[1], o[2] = nd.translation[2], o[3] = 0.f;
    }
    if (quat4) std::memcpy(quat4 + 4 * (size_t)i, nd.rotation, sizeof(nd.rotation));
    if (scale4) {
      float *o = scale4 + 4 * (size_t)i;
      o[0] = nd.scale[0], o[1] = nd.scale[1], o[2] = nd.scale[2], o[3] = 0.f;
    }
    if (mesh) mesh[i] = nd.mesh;
    if (has_matrix) has_matrix[i] = nd.has_matrix ? 1 : 0;
    if (matrix16) std::memcpy(matrix16 + 16 * (size_t)i, nd.matrix, sizeof(nd.matrix));
  }
  return VKRHIP_OK;
} catch (...) {
  return VKRHIP_ERR_ARG;  // out of host memory while parsing
}

// The bounds of every mesh primitive, from the POSITION accessor's "min" / "max"
// (glTF 2.0 requires both on POSITION accessors): what the reference computes over
// the primitive's vertex positions when it imports the file
// (vkr_mesh_loader_compute_bounds, mesh_loader.c:1159-1179: component-wise minimum
// and maximum, centre = (min + max) * 0.5) and stores as the submesh's center /
// min_extents / max_extents (vkr_resources.h:292-294).
int vkrhip_gltf_read_mesh_bounds(const char *json, uint64_t length, uint32_t capacity,
                                 vkrhip_gltf_meshes *info, uint32_t *mesh, uint32_t *primitive,
                                 int32_t *material, uint8_t *has_bounds, float *bounds12) try {
  if (!json || !info) return VKRHIP_ERR_ARG;
  std::memset(info, 0, sizeof(*info));
  struct Primitive {
    uint32_t mesh, index;
    int64_t position = -1;
    int32_t material = -1;
  };
  struct Accessor {
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    bool have_lo = false, have_hi = false;
  };
  std::vector<Primitive> prims;
  std::vector<Accessor> accessors;
  uint32_t mesh_count = 0;
  Cursor c{json, json + length};
  const bool parsed = parse_object(c, 0, [&](const std::string &key, Cursor &v) {
    if (key == "meshes") {
      prims.clear();
      mesh_count = 0;
      return parse_array(v, 1, [&](uint32_t m, Cursor &mv) {
        mesh_count = m + 1u;
        return parse_object(mv, 2, [&](const std::string &k, Cursor &pv) {
          if (k != "primitives") return skip_value(pv, 3);
          return parse_array(pv, 3, [&](uint32_t pi, Cursor &ev) {
            prims.emplace_back();
            prims.back().mesh = m;
            prims.back().index = pi;
            const size_t at = prims.size() - 1;
            return parse_object(ev, 4, [&](const std::string &pk, Cursor &av) {
              if (pk == "material") {
                uint32_t id;
                if (!parse_index(av, &id) || id > 0x7fffffffu) return false;
                prims[at].material = (int32_t)id;
                return true;
              }
              if (pk != "attributes") return skip_value(av, 5);
              return parse_object(av, 5, [&](const std::string &ak, Cursor &iv) {
                if (ak != "POSITION") return skip_value(iv, 6);
                uint32_t id;
                if (!parse_index(iv, &id)) return false;
                prims[at].position = id;
                return true;
              });
            });
          });
        });
      });
    }
    if (key == "accessors") {
      accessors.clear();
      return parse_array(v, 1, [&](uint32_t, Cursor &av) {
        accessors.emplace_back();
        const size_t at = accessors.size() - 1;
        return parse_object(av, 2, [&](const std::string &k, Cursor &mv) {
          // (only three-component bounds are of interest; others are skipped whole)
          if (k == "min" || k == "max") {
            Cursor probe = mv;
            float three[3];
            if (parse_floats(probe, 3, three, 3)) {
              Accessor &a = accessors[at];
              std::memcpy(k == "min" ? a.lo : a.hi, three, sizeof(three));
              (k == "min" ? a.have_lo : a.have_hi) = true;
              mv = probe;
              return true;
            }
          }
          return skip_value(mv, 3);
        });
      });
    }
    return skip_value(v, 1);
  });
  skip_space(c);
  if (!parsed || c.p != c.end) return VKRHIP_ERR_ARG;  // not a JSON document
  for (const Primitive &p : prims)
    if (p.position >= 0 && (uint64_t)p.position >= accessors.size()) return VKRHIP_ERR_ARG;
  info->mesh_count = mesh_count;
  info->primitive_count = (uint32_t)prims.size();
  info->accessor_count = (uint32_t)accessors.size();
  if (prims.size() > capacity) return VKRHIP_ERR_CAPACITY;  // info tells how much room the arrays need
  for (size_t i = 0; i < prims.size(); ++i) {
    const Primitive &p = prims[i];
    if (mesh) mesh[i] = p.mesh;
    if (primitive) primitive[i] = p.index;
    if (material) material[i] = p.material;
    const Accessor *a = p.position >= 0 ? &accessors[(size_t)p.position] : nullptr;
    const bool have = a && a->have_lo && a->have_hi;
    if (has_bounds) has_bounds[i] = have ? 1 : 0;
    if (have) info->bounded_count += 1u;
    if (bounds12) {
      float *o = bounds12 + 12 * i;
      std::memset(o, 0, 12 * sizeof(float));
      if (have)
        for (int k = 0; k < 3; ++k) {
          o[k] = (a->lo[k] + a->hi[k]) * 0.5f;  // center
          o[4 + k] = a->lo[k];                   // min_extents
          o[8 + k] = a->hi[k];                   // max_extents
        }
    }
  }
  return VKRHIP_OK;
} catch (...) {
  return VKRHIP_ERR_ARG;  // out of host memory while parsing
}


// "entities" of one of the reference's scene files (version 2).
int vkrhip_scene_read_entities(const char *json, uint64_t length, uint32_t capacity,
                               vkrhip_scene_entities *info, uint32_t *parent, float *pos4, float *quat4,
                               float *scale4, uint8_t *kind, float *dimensions3,
                               uint64_t *material_path) try {
  if (!json || !info) return VKRHIP_ERR_ARG;
  std::memset(info, 0, sizeof(*info));
  struct Entity {
    uint32_t parent = 0xffffffffu;
    float pos[3] = {0.f, 0.f, 0.f};
    float rot[4] = {0.f, 0.f, 0.f, 1.f};
    float scale[3] = {1.f, 1.f, 1.f};
    float dims[3] = {1.f, 1.f, 1.f};
    bool shape = false, mesh = false;
    uint64_t path_begin = 0, path_length = 0;  // the material path's text inside the document
  };
  std::vector<Entity> entities;
  double version = 0.0;
  // a member that may be null: true when it was (and has been consumed)
  auto take_null = [](Cursor &v) {
    skip_space(v);
    return v.p < v.end && *v.p == 'n' && skip_literal(v, "null");
  };

  Cursor c{json, json + length};
  const bool parsed = parse_object(c, 0, [&](const std::string &key, Cursor &v) {
    if (key == "version") return parse_number(v, &version);
    if (key != "entities") return skip_value(v, 1);
    entities.clear();
    return parse_array(v, 1, [&](uint32_t, Cursor &ev) {
      entities.emplace_back();
      Entity &e = entities.back();
      return parse_object(ev, 2, [&](const std::string &k, Cursor &mv) {
        if (k == "parent") {
          if (take_null(mv)) return true;
          return parse_index(mv, &e.parent);
        }
        if (k == "transform") {
          if (take_null(mv)) return true;
          return parse_object(mv, 3, [&](const std::string &tk, Cursor &tv) {
            if (tk == "pos") return parse_floats(tv, 4, e.pos, 3);
            if (tk == "rot") return parse_floats(tv, 4, e.rot, 4);
            if (tk == "scale") return parse_floats(tv, 4, e.scale, 3);
            return skip_value(tv, 4);
          });
        }
        if (k == "mesh") {
          if (take_null(mv)) return true;
          e.mesh = true;
          return skip_value(mv, 3);
        }
        if (k == "shape") {
          if (take_null(mv)) return true;
          e.shape = true;
          return parse_object(mv, 3, [&](const std::string &sk, Cursor &sv) {
            if (sk == "dimensions") return parse_floats(sv, 4, e.dims, 3);
            if (sk != "material" || take_null(sv)) return sk == "material" ? true : skip_value(sv, 4);
            return parse_object(sv, 4, [&](const std::string &pk, Cursor &pv) {
              if (pk != "path") return skip_value(pv, 5);
              skip_space(pv);
              const char *open = pv.p;  // at the opening quote
              if (!parse_string(pv, nullptr)) return false;
              e.path_begin = (uint64_t)(open + 1 - json);
              e.path_length = (uint64_t)(pv.p - 1 - (open + 1));
              return true;
            });
          });
        }
        return skip_value(mv, 3);
      });
    });
  });
  skip_space(c);
  if (!parsed || c.p != c.end) return VKRHIP_ERR_ARG;  // not a JSON document
  if (!(version >= 0.0) || version > 4294967295.0) return VKRHIP_ERR_ARG;

  const uint32_t n = (uint32_t)entities.size();
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = entities[i].parent;
    if (p != 0xffffffffu && (p >= n || p == i)) return VKRHIP_ERR_ARG;
  }
  for (uint32_t i = 0; i < n; ++i) {  // no cycles
    uint32_t at = i, steps = 0;
    while (entities[at].parent != 0xffffffffu) {
      at = entities[at].parent;
      if (++steps > n) return VKRHIP_ERR_ARG;
    }
  }
  info->version = (uint32_t)version;
  info->entity_count = n;
  for (const Entity &e : entities) info->renderable_count += (e.shape || e.mesh) ? 1u : 0u;
  if (n > capacity) return VKRHIP_ERR_CAPACITY;  // info tells how much room the arrays need

  for (uint32_t i = 0; i < n; ++i) {
    const Entity &e = entities[i];
    if (parent) parent[i] = e.parent;
    if (pos4) {
      float *o = pos4 + 4 * (size_t)i;
      o[0] = e.pos[0], o[1] = e.pos[1], o[2] = e.pos[2], o[3] = 0.f;
    }
    if (quat4) std::memcpy(quat4 + 4 * (size_t)i, e.rot, sizeof(e.rot));
    if (scale4) {
      float *o = scale4 + 4 * (size_t)i;
      o[0] = e.scale[0], o[1] = e.scale[1], o[2] = e.scale[2], o[3] = 0.f;
    }
    if (kind) kind[i] = e.shape ? 1u : (e.mesh ? 2u : 0u);
    if (dimensions3) std::memcpy(dimensions3 + 3 * (size_t)i, e.dims, sizeof(e.dims));
    if (material_path) {
      material_path[2 * (size_t)i] = e.shape ? e.path_begin : 0u;
      material_path[2 * (size_t)i + 1] = e.shape ? e.path_length : 0u;
    }
  }
  return VKRHIP_OK;
} catch (...) {
  return VKRHIP_ERR_ARG;  // out of host memory while parsing
}

}  // extern "C"
