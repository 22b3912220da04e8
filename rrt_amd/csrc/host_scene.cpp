// Host side of the seam: the text scene format of rogerallen/rrt (scene.h:212-452) parsed into
// the POD tables of include/rrtx.h.  Own implementation of the grammar:
//   camera fx fy fz  ax ay az  ux uy uz  vfov aperture focus [time0 time1]
//   material <name> lambertian r g b | metal r g b fuzz | dielectric ir
//   sphere cx cy cz r <mat>
//   msphere c0x c0y c0z c1x c1y c1z t0 t1 r <mat>
//   obj_beg <nverts> <ntris> / obj_vtx x y z / obj_tri i j k / obj_end
//   obj <obj_idx> <mat> { t x y z | s x y z | r angle_deg ax ay az }*
// A line is dispatched by the keyword that starts at column 0 (in the order above); every other
// line is ignored.  Errors never exit the process here: the reference's exit code is reported
// through rrtx_scene_exit_code() and the CLI (rrt_main.cpp) exits with it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rrtx.h"

namespace {

thread_local int g_exit_code = 0;
thread_local std::string g_scene_error;

template <typename F> struct Tables;
template <> struct Tables<float> {
    typedef rrtx_camera_f32 camera_t;
    typedef rrtx_material_f32 material_t;
    typedef rrtx_sphere_f32 sphere_t;
    typedef rrtx_moving_sphere_f32 msphere_t;
    typedef rrtx_triangle_f32 triangle_t;
};
template <> struct Tables<double> {
    typedef rrtx_camera_f64 camera_t;
    typedef rrtx_material_f64 material_t;
    typedef rrtx_sphere_f64 sphere_t;
    typedef rrtx_moving_sphere_f64 msphere_t;
    typedef rrtx_triangle_f64 triangle_t;
};

struct ParseError {
    int exit_code;
    std::string what;
};

// whitespace-separated tokens of one line, with checked numeric access
class Tokens {
  public:
    explicit Tokens(const std::string &line)
    {
        std::istringstream in(line);
        std::string t;
        while (in >> t) tok_.push_back(t);
    }
    size_t size() const { return tok_.size(); }
    bool more() const { return pos_ < tok_.size(); }
    void skip(size_t n = 1) { pos_ += n; }
    const std::string &word()
    {
        static const std::string empty;
        if (pos_ >= tok_.size()) {
            ++pos_;
            return empty; // the reference reads an empty string here (iss >> s on exhausted stream)
        }
        return tok_[pos_++];
    }
    double real()
    {
        if (pos_ >= tok_.size()) throw ParseError{5, "missing number"};
        const std::string &t = tok_[pos_++];
        char *end = nullptr;
        double v = std::strtod(t.c_str(), &end); // std::stod semantics: leading numeric prefix
        if (end == t.c_str()) throw ParseError{5, "bad number '" + t + "'"};
        return v;
    }
    int integer()
    {
        if (pos_ >= tok_.size()) throw ParseError{5, "missing integer"};
        const std::string &t = tok_[pos_++];
        char *end = nullptr;
        long v = std::strtol(t.c_str(), &end, 10);
        if (end == t.c_str()) throw ParseError{5, "bad integer '" + t + "'"};
        return (int)v;
    }
    char peek_char() const { return tok_[pos_][0]; }

  private:
    std::vector<std::string> tok_;
    size_t pos_ = 0;
};

template <typename F> struct Vec {
    F e[3];
};
template <typename F> Vec<F> vmake(F a, F b, F c) { return Vec<F>{{a, b, c}}; }
template <typename F> Vec<F> vsub(const Vec<F> &a, const Vec<F> &b) { return vmake<F>(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
template <typename F> Vec<F> vadd(const Vec<F> &a, const Vec<F> &b) { return vmake<F>(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
template <typename F> Vec<F> vhad(const Vec<F> &a, const Vec<F> &b) { return vmake<F>(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
template <typename F> Vec<F> vscl(F t, const Vec<F> &v) { return vmake<F>(t * v.e[0], t * v.e[1], t * v.e[2]); }
template <typename F> F vdot(const Vec<F> &a, const Vec<F> &b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
template <typename F> Vec<F> vcrs(const Vec<F> &u, const Vec<F> &v)
{
    return vmake<F>(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
template <typename F> Vec<F> vunit(const Vec<F> &v)
{
    F len = std::sqrt(vdot<F>(v, v));
    return vscl<F>((F)1 / len, v); // vec3.h:113,125
}
template <typename F> Vec<F> read_vec(Tokens &t)
{
    F a = (F)t.real();
    F b = (F)t.real();
    F c = (F)t.real();
    return vmake<F>(a, b, c);
}
template <typename F> void put(F dst[3], const Vec<F> &v) { dst[0] = v.e[0], dst[1] = v.e[1], dst[2] = v.e[2]; }

// camera::camera, camera.h:8-29.  `tan` there resolves to ::tan(double) even for FP_T = float, so
// h, viewport_height and viewport_width are doubles that round to FP_T only in
// focus_dist * viewport_width (* u) — reproduced here because it decides the last bit of
// `horizontal` / `vertical` / `lower_left_corner`.
template <typename F> typename Tables<F>::camera_t build_camera(const Vec<F> &from, const Vec<F> &at, const Vec<F> &up, F vfov, F aspect, F aperture, F focus, F t0, F t1)
{
    const F pi = (F)3.1415926535897932385; // rtweekend.h:56
    const F theta = vfov * pi / (F)180.0;  // rtweekend.h:60
    const double h = std::tan((double)(theta / (F)2));
    const double vh = (F)2.0 * h;
    const double vw = aspect * vh;
    typename Tables<F>::camera_t c;
    const Vec<F> w = vunit<F>(vsub<F>(from, at));
    const Vec<F> u = vunit<F>(vcrs<F>(up, w));
    const Vec<F> v = vcrs<F>(w, u);
    const Vec<F> horizontal = vscl<F>((F)(focus * vw), u);
    const Vec<F> vertical = vscl<F>((F)(focus * vh), v);
    const Vec<F> half_h = vscl<F>((F)1 / (F)2, horizontal);
    const Vec<F> half_v = vscl<F>((F)1 / (F)2, vertical);
    const Vec<F> llc = vsub<F>(vsub<F>(vsub<F>(from, half_h), half_v), vscl<F>(focus, w));
    put<F>(c.origin, from);
    put<F>(c.lower_left_corner, llc);
    put<F>(c.horizontal, horizontal);
    put<F>(c.vertical, vertical);
    put<F>(c.u, u);
    put<F>(c.v, v);
    put<F>(c.w, w);
    c.lens_radius = aperture / 2;
    c.time0 = t0;
    c.time1 = t1;
    return c;
}

template <typename F> struct Mesh {
    int want_verts = 0, want_tris = 0;
    std::vector<Vec<F>> verts;
    std::vector<int> index; // 3 per triangle
};

template <typename F> struct Transform {
    char kind;
    Vec<F> v;
    double angle_deg;
    Vec<F> apply(const Vec<F> &p) const
    {
        switch (kind) {
        case 't': return vadd<F>(p, v); // scene.h:120
        case 's': return vhad<F>(p, v); // scene.h:127
        default: {                      // Rodrigues, scene.h:137-143; cos/sin there are the double ones
            const F pi = (F)3.1415926535897932385;
            const F theta = (F)(angle_deg * (pi / 180));
            const F c = (F)std::cos((double)theta);
            const F s = (F)std::sin((double)theta);
            const Vec<F> a = vscl<F>(c, p);
            const Vec<F> b = vscl<F>(s, vcrs<F>(v, p));
            const Vec<F> d = vscl<F>((F)1 - c, vscl<F>(vdot<F>(v, p), v));
            return vadd<F>(vadd<F>(a, b), d);
        }
        }
    }
};

template <typename F> struct SceneData {
    typename Tables<F>::camera_t camera;
    std::vector<typename Tables<F>::material_t> materials;
    std::vector<typename Tables<F>::sphere_t> spheres;
    std::vector<typename Tables<F>::msphere_t> msph;
    std::vector<typename Tables<F>::triangle_t> tris;
    int n_objs = 0, n_insts = 0;
};

bool begins(const std::string &line, const char *kw) { return line.compare(0, std::strlen(kw), kw) == 0; }

template <typename F> void parse_file(const char *path, int width, int height, SceneData<F> &out)
{
    std::ifstream file(path);
    if (!file.good()) throw ParseError{2, std::string("ERROR: problem with opening file: ") + path};

    std::map<std::string, int> material_index;
    auto lookup = [&](const std::string &name) { return material_index[name]; }; // unknown name -> 0 (scene.h:310)
    std::vector<Mesh<F>> meshes;
    Mesh<F> open_mesh;
    bool mesh_open = false;
    struct Instance {
        int mesh, material;
        std::vector<Transform<F>> xf;
    };
    std::vector<Instance> instances;
    bool have_camera = false;

    std::string line;
    while (std::getline(file, line)) {
        Tokens t(line);
        if (begins(line, "camera")) {
            t.skip();
            const Vec<F> from = read_vec<F>(t), at = read_vec<F>(t), up = read_vec<F>(t);
            const double vfov = t.real(), aperture = t.real(), focus = t.real();
            double t0 = 0.0, t1 = 0.0;
            if (t.more()) { // scene.h:249
                t0 = t.real();
                t1 = t.real();
            }
            const double aspect = double(width) / height; // scene.h:254
            out.camera = build_camera<F>(from, at, up, (F)vfov, (F)aspect, (F)aperture, (F)focus, (F)t0, (F)t1);
            have_camera = true;
        }
        else if (begins(line, "material")) {
            t.skip();
            const std::string name = t.word();
            const std::string type = t.word();
            typename Tables<F>::material_t m;
            std::memset(&m, 0, sizeof m);
            if (type == "lambertian") {
                m.type = RRTX_LAMBERTIAN;
                put<F>(m.mat.lambertian.albedo, read_vec<F>(t));
            }
            else if (type == "metal") {
                m.type = RRTX_METAL;
                put<F>(m.mat.metal.albedo, read_vec<F>(t));
                m.mat.metal.fuzz = t.real(); // kept as double, scene.h:279
            }
            else if (type == "dielectric") {
                m.type = RRTX_DIELECTRIC;
                m.mat.dielectric.ref_idx = (F)t.real(); // scene.h:285
            }
            else
                throw ParseError{3, "ERROR: unknown material type: " + type};
            material_index.insert(std::make_pair(name, (int)out.materials.size())); // first definition wins
            out.materials.push_back(m);
        }
        else if (begins(line, "sphere")) {
            t.skip();
            typename Tables<F>::sphere_t s;
            std::memset(&s, 0, sizeof s);
            put<F>(s.center, read_vec<F>(t));
            s.radius = (F)t.real(); // scene.h:309
            s.material_idx = lookup(t.word());
            out.spheres.push_back(s);
        }
        else if (begins(line, "msphere")) {
            t.skip();
            typename Tables<F>::msphere_t s;
            std::memset(&s, 0, sizeof s);
            put<F>(s.center0, read_vec<F>(t));
            put<F>(s.center1, read_vec<F>(t));
            s.time0 = t.real(); // doubles, not rounded through FP_T: scene.h:334-336
            s.time1 = t.real();
            s.radius = t.real();
            s.material_idx = lookup(t.word());
            out.msph.push_back(s);
        }
        else if (begins(line, "obj_beg")) {
            if (mesh_open) throw ParseError{1, "ERROR: obj_beg called without prior obj_end."};
            t.skip();
            open_mesh = Mesh<F>();
            open_mesh.want_verts = t.integer();
            open_mesh.want_tris = t.integer();
            mesh_open = true;
        }
        else if (begins(line, "obj_vtx")) {
            if (!mesh_open) throw ParseError{1, "ERROR: obj_vtx called without prior obj_beg"};
            if ((int)open_mesh.verts.size() == open_mesh.want_verts) throw ParseError{1, "ERROR: only expected " + std::to_string(open_mesh.want_verts) + " vertices."};
            t.skip();
            open_mesh.verts.push_back(read_vec<F>(t));
        }
        else if (begins(line, "obj_tri")) {
            if (!mesh_open) throw ParseError{1, "ERROR: obj_tri called without prior obj_beg."};
            if ((int)open_mesh.index.size() == 3 * open_mesh.want_tris) throw ParseError{1, "ERROR: only expected " + std::to_string(open_mesh.want_tris) + " triangles."};
            t.skip();
            for (int k = 0; k < 3; ++k) open_mesh.index.push_back(t.integer());
        }
        else if (begins(line, "obj_end")) {
            if (!mesh_open) throw ParseError{1, "ERROR: obj_end called without prior obj_beg."};
            if ((int)open_mesh.verts.size() != open_mesh.want_verts)
                throw ParseError{1, "ERROR: expected " + std::to_string(open_mesh.want_verts) + " vertices, got " + std::to_string(open_mesh.verts.size()) + "."};
            if ((int)open_mesh.index.size() != 3 * open_mesh.want_tris)
                throw ParseError{1, "ERROR: expected " + std::to_string(open_mesh.want_tris) + " triangles, got " + std::to_string(open_mesh.index.size() / 3) + "."};
            meshes.push_back(open_mesh);
            mesh_open = false;
        }
        else if (begins(line, "obj")) {
            if (t.size() < 2) throw ParseError{1, "ERROR: obj called without enough args (count = " + std::to_string(t.size() + 1)}; // scene.h:396: two real tokens suffice
            t.skip();
            Instance inst;
            inst.mesh = t.integer();
            inst.material = lookup(t.word());
            while (t.more()) {
                Transform<F> x;
                x.kind = t.peek_char();
                x.angle_deg = 0;
                t.skip();
                if (x.kind == 't' || x.kind == 's')
                    x.v = read_vec<F>(t);
                else if (x.kind == 'r') {
                    x.angle_deg = (F)t.real(); // xf_rotate(FP_T a, ...), scene.h:136
                    x.v = read_vec<F>(t);
                }
                else // the reference loops forever on such a token (scene.h:400-420); reject it
                    throw ParseError{1, "ERROR: obj: unknown transform selector"};
                inst.xf.push_back(x);
            }
            instances.push_back(inst);
        }
    }
    if (!have_camera) throw ParseError{4, "ERROR: Scene did not have a camera."};
    if (out.materials.empty()) throw ParseError{4, "ERROR: Scene did not have any materials."};
    if (out.spheres.size() + out.msph.size() + instances.size() == 0) throw ParseError{4, "ERROR: Scene did not have any objects."};

    // scene::fill_instance_triangles, scene.h:153-177,467-472
    for (const Instance &inst : instances) {
        if (inst.mesh < 0 || inst.mesh >= (int)meshes.size()) throw ParseError{1, "ERROR: obj refers to an undefined object index"};
        const Mesh<F> &m = meshes[inst.mesh];
        for (int k = 0; k < m.want_tris; ++k) {
            typename Tables<F>::triangle_t tri;
            std::memset(&tri, 0, sizeof tri);
            for (int c = 0; c < 3; ++c) {
                const int vi = m.index[3 * k + c];
                if (vi < 0 || vi >= (int)m.verts.size()) throw ParseError{1, "ERROR: obj_tri vertex index out of range"};
                Vec<F> p = m.verts[vi];
                for (const Transform<F> &x : inst.xf) p = x.apply(p);
                put<F>(tri.vertices[c], p);
            }
            tri.material_idx = inst.material;
            out.tris.push_back(tri);
        }
    }
    out.n_objs = (int)meshes.size();
    out.n_insts = (int)instances.size();
}

} // namespace

struct rrtx_scene {
    int fp64;
    SceneData<float> f;
    SceneData<double> d;
};

extern "C" {

int rrtx_scene_exit_code(void) { return g_exit_code; }

const char *rrtx_scene_error(void) { return g_scene_error.c_str(); }

static int scene_load(const char *path, int image_width, int image_height, int fp64, rrtx_scene **out, bool print)
{
    if (!out) return RRTX_E_INVALID;
    *out = nullptr;
    g_exit_code = 0;
    g_scene_error.clear();
    if (!path || image_width < 1 || image_height < 1) return RRTX_E_INVALID;
    rrtx_scene *s = new rrtx_scene();
    s->fp64 = fp64 ? 1 : 0;
    try {
        if (fp64)
            parse_file<double>(path, image_width, image_height, s->d);
        else
            parse_file<float>(path, image_width, image_height, s->f);
    }
    catch (const ParseError &e) {
        delete s;
        g_exit_code = e.exit_code;
        g_scene_error = e.what;
        if (print) fprintf(stderr, "%s\n", e.what.c_str());
        return e.exit_code == 2 ? RRTX_E_IO : RRTX_E_PARSE;
    }
    *out = s;
    return RRTX_OK;
}

int rrtx_scene_load(const char *path, int image_width, int image_height, int fp64, rrtx_scene **out) { return scene_load(path, image_width, image_height, fp64, out, true); }
int rrtx_scene_load_quiet(const char *path, int image_width, int image_height, int fp64, rrtx_scene **out) { return scene_load(path, image_width, image_height, fp64, out, false); }

void rrtx_scene_free(rrtx_scene *s) { delete s; }

int rrtx_scene_describe(const rrtx_scene *s, rrtx_scene_desc *out)
{
    if (!s || !out) return RRTX_E_INVALID;
    std::memset(out, 0, sizeof *out);
    out->fp64 = s->fp64;
    if (s->fp64) {
        out->camera = &s->d.camera;
        out->num_materials = (int)s->d.materials.size(), out->materials = s->d.materials.data();
        out->num_spheres = (int)s->d.spheres.size(), out->spheres = s->d.spheres.data();
        out->num_moving_spheres = (int)s->d.msph.size(), out->moving_spheres = s->d.msph.data();
        out->num_triangles = (int)s->d.tris.size(), out->triangles = s->d.tris.data();
    }
    else {
        out->camera = &s->f.camera;
        out->num_materials = (int)s->f.materials.size(), out->materials = s->f.materials.data();
        out->num_spheres = (int)s->f.spheres.size(), out->spheres = s->f.spheres.data();
        out->num_moving_spheres = (int)s->f.msph.size(), out->moving_spheres = s->f.msph.data();
        out->num_triangles = (int)s->f.tris.size(), out->triangles = s->f.tris.data();
    }
    return RRTX_OK;
}

int rrtx_scene_counts(const rrtx_scene *s, int32_t counts[6])
{
    if (!s || !counts) return RRTX_E_INVALID;
    rrtx_scene_desc d;
    rrtx_scene_describe(s, &d);
    counts[0] = d.num_materials;
    counts[1] = d.num_spheres;
    counts[2] = d.num_moving_spheres;
    counts[3] = d.num_triangles;
    counts[4] = s->fp64 ? s->d.n_objs : s->f.n_objs;
    counts[5] = s->fp64 ? s->d.n_insts : s->f.n_insts;
    return RRTX_OK;
}

} // extern "C"
