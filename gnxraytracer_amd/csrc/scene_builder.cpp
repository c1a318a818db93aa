// scene_builder.cpp -- host-side scene authoring: the C++ mirror of ui/ModelList.cpp,
// ui/MaterialList.cpp and the scene part of ui/RenderThread.cpp:46-187, producing the flat
// gnxr_scene_desc instead of a pbr::Scene.  Pure host code (no HIP calls).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "host_scene.h"

namespace gnxr {

static thread_local char g_error[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_error; }

// ---- Transform restatement (core/Transform.cpp) ----
Mat4 inverse(const Mat4 &mm) {  // Transform.cpp:54-108
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float minv[4][4];
    memcpy(minv, mm.m, 64);
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0 && std::abs(minv[j][k]) >= big) { big = std::abs(minv[j][k]); irow = j; icol = k; }
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol) for (int k = 0; k < 4; ++k) std::swap(minv[irow][k], minv[icol][k]);
        indxr[i] = irow;
        indxc[i] = icol;
        float pivinv = (float)(1. / minv[icol][icol]);
        minv[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                float save = minv[j][icol];
                minv[j][icol] = 0;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--)
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) std::swap(minv[k][indxr[j]], minv[k][indxc[j]]);
    Mat4 r;
    memcpy(r.m, minv, 64);
    return r;
}
Xf translate(Vec3 d) {
    Xf t;
    t.m.m[0][3] = d.x; t.m.m[1][3] = d.y; t.m.m[2][3] = d.z;
    t.inv.m[0][3] = -d.x; t.inv.m[1][3] = -d.y; t.inv.m[2][3] = -d.z;
    return t;
}
Xf scale(float x, float y, float z) {
    Xf t;
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.inv.m[0][0] = 1 / x; t.inv.m[1][1] = 1 / y; t.inv.m[2][2] = 1 / z;
    return t;
}
static inline float radians(float deg) { return (kPi / 180) * deg; }
Xf rotate_x(float theta) {  // Transform.cpp:130-137
    float s = std::sin(radians(theta)), c = std::cos(radians(theta));
    Xf t;
    t.m.m[1][1] = c; t.m.m[1][2] = -s; t.m.m[2][1] = s; t.m.m[2][2] = c;
    t.inv = transpose(t.m);
    return t;
}
Xf rotate_y(float theta) {  // Transform.cpp:139-145
    float s = std::sin(radians(theta)), c = std::cos(radians(theta));
    Xf t;
    t.m.m[0][0] = c; t.m.m[0][2] = s; t.m.m[2][0] = -s; t.m.m[2][2] = c;
    t.inv = transpose(t.m);
    return t;
}
Xf rotate_axis(float theta, Vec3 axis) {  // Rotate(theta, axis), Transform.cpp:156-179
    Vec3 a = normalize(axis);
    float sinTheta = std::sin(radians(theta)), cosTheta = std::cos(radians(theta));
    Xf t;
    Mat4 &m = t.m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * cosTheta;
    m.m[0][1] = a.x * a.y * (1 - cosTheta) - a.z * sinTheta;
    m.m[0][2] = a.x * a.z * (1 - cosTheta) + a.y * sinTheta;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - cosTheta) + a.z * sinTheta;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * cosTheta;
    m.m[1][2] = a.y * a.z * (1 - cosTheta) - a.x * sinTheta;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - cosTheta) - a.y * sinTheta;
    m.m[2][1] = a.y * a.z * (1 - cosTheta) + a.x * sinTheta;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * cosTheta;
    m.m[2][3] = 0;
    t.inv = transpose(t.m);
    return t;
}
Xf look_at(Vec3 pos, Vec3 look, Vec3 up) {  // Transform.cpp:181-215
    Mat4 c2w;
    c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1;
    Vec3 dir = normalize(look - pos);
    Vec3 right = normalize(cross(normalize(up), dir));
    Vec3 newUp = cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.;
    return {inverse(c2w), c2w};
}
Xf perspective(float fov, float n, float f) {  // Transform.cpp:287-296
    Mat4 persp;
    persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n);
    persp.m[3][2] = 1; persp.m[3][3] = 0;
    float invTanAng = 1 / std::tan(radians(fov) / 2);
    Xf p{persp, inverse(persp)};
    return xmul(scale(invTanAng, invTanAng, 1), p);
}
Vec3 xform_point(const Mat4 &m, Vec3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
    float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
    float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
    float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
    if (wp == 1) return {xp, yp, zp};
    float inv = 1.f / wp;
    return {inv * xp, inv * yp, inv * zp};
}
Vec3 xform_vector(const Mat4 &m, Vec3 v) {
    float x = v.x, y = v.y, z = v.z;
    return {m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
            m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z};
}

// ---- Builder ----
Builder::Builder() {
    // ui/RenderThread.cpp:60-68 + camera/Perspective.cpp:116-134
    camera = gnxr_camera{{0.f, 0.f, 5.f}, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, 90.f, 0.f, 3.f};
}

// TriangleMesh ctor: vertices are transformed to world space once (shape/Triangle.cpp:27-31)
int Builder::add_mesh(const float *verts, int nv, const int32_t *idx, int nt, const Xf &o2w, int material, int med_in, int med_out) {
    int first_vertex = (int)vertices.size() / 3;
    int first_tri = (int)indices.size() / 3;
    for (int i = 0; i < nv; ++i) {
        Vec3 p = xform_point(o2w.m, Vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]));
        vertices.push_back(p.x); vertices.push_back(p.y); vertices.push_back(p.z);
    }
    for (int t = 0; t < nt; ++t) {
        for (int k = 0; k < 3; ++k) {
            int v = idx[3 * t + k];
            if (v < 0 || v >= nv) { set_error("add_mesh: vertex index %d out of range [0,%d)", v, nv); return GNXR_ERR_INVALID; }
            indices.push_back(first_vertex + v);
        }
        tri_material.push_back(material);
        tri_light.push_back(-1);
        tri_med_in.push_back(med_in);
        tri_med_out.push_back(med_out);
        if (!tri_uv.empty()) { const float def[6] = {0, 0, 1, 0, 1, 1}; tri_uv.insert(tri_uv.end(), def, def + 6); }
        if (!tri_n.empty()) tri_n.insert(tri_n.end(), 9, 0.f);
        if (!tri_s.empty()) tri_s.insert(tri_s.end(), 9, 0.f);
    }
    return first_tri;
}

void Builder::fill_desc(gnxr_scene_desc *d) const {
    memset(d, 0, sizeof(*d));
    d->abi_version = GNXR_ABI_VERSION;
    d->n_vertices = (int)vertices.size() / 3;
    d->n_triangles = (int)indices.size() / 3;
    d->n_materials = (int)materials.size();
    d->n_lights = (int)lights.size();
    d->n_media = (int)media.size();
    d->env_width = env_w;
    d->env_height = env_h;
    d->vertices = vertices.data();
    d->indices = indices.data();
    d->tri_material = tri_material.data();
    d->tri_light = tri_light.data();
    d->tri_medium_inside = tri_med_in.data();
    d->tri_medium_outside = tri_med_out.data();
    d->materials = materials.data();
    d->lights = lights.data();
    d->media = media.empty() ? nullptr : media.data();
    d->grid_density = grid_density.empty() ? nullptr : grid_density.data();
    d->env_rgb = env_rgb.empty() ? nullptr : env_rgb.data();
    d->camera = camera;
    d->camera_medium = camera_medium;
    d->n_spheres = (int)spheres.size();
    d->spheres = spheres.empty() ? nullptr : spheres.data();
    d->n_textures = (int)textures.size();
    d->textures = textures.empty() ? nullptr : textures.data();
    d->texels = texels.empty() ? nullptr : texels.data();
    d->tri_uv = tri_uv.empty() ? nullptr : tri_uv.data();
    d->tri_n = tri_n.empty() ? nullptr : tri_n.data();
    d->tri_s = tri_s.empty() ? nullptr : tri_s.data();
    d->bvh_split_method = bvh_split_method;
}

// ---- `.3d` text meshes: shape/plyRead.h:19-48 ----
bool read_model_3d(const char *path, std::vector<float> *verts, std::vector<int32_t> *idx) {
    std::ifstream f(path);
    if (!f) { set_error("cannot open %s", path); return false; }
    int nVertices = -1, nTriangles = -1;
    std::string ed;
    for (int i = 0; i < 2; i++) {
        f >> ed;
        if (ed == "vertex") f >> nVertices;
        else if (ed == "face") f >> nTriangles;
    }
    if (nVertices <= 0 || nTriangles <= 0) { set_error("%s: bad .3d header", path); return false; }
    verts->resize((size_t)nVertices * 3);
    idx->resize((size_t)nTriangles * 3);
    for (int i = 0; i < nVertices; i++) {
        float x, y, z;
        f >> x >> y >> z;
        // `vertexArray[i] *= 20` (plyRead.h:38)
        (*verts)[3 * i] = x * 20; (*verts)[3 * i + 1] = y * 20; (*verts)[3 * i + 2] = z * 20;
    }
    for (int i = 0; i < nTriangles; i++) {
        f >> ed;
        f >> (*idx)[3 * i] >> (*idx)[3 * i + 1] >> (*idx)[3 * i + 2];
    }
    if (!f) { set_error("%s: truncated .3d file", path); return false; }
    return true;
}

// Deterministic stand-in for the missing Resources/dragon.3d: a (2,3) torus-knot tube with a
// seeded multi-octave radial displacement, tessellated to ~target_tris triangles.  Native extent
// about [-0.1,0.1] x [0.03,0.23] x [-0.07,0.07] so that after the loader's x20 scale and AddModel's
// y-2.9 translation (ModelList.cpp:56) it sits inside the Cornell box like the dragon did.
static inline uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
bool write_synthetic_3d(const char *path, int target_tris, uint32_t seed) {
    if (target_tris < 64) target_tris = 64;
    // nu * nv quads, 2 triangles each, nu = 5 nv
    int nv = std::max(8, (int)std::lround(std::sqrt(target_tris / 10.0)));
    int nu = 5 * nv;
    // seeded displacement spectrum
    const int K = 6;
    double amp[K], fu[K], fv[K], ph[K];
    for (int k = 0; k < K; ++k) {
        uint32_t h = hash_u32(seed * 7919u + k * 104729u + 1u);
        amp[k] = 0.12 / (1 + k) * (0.5 + (h & 0xffff) / 65535.0);
        fu[k] = 3 + (int)((h >> 16) % (7 * (k + 1)));
        fv[k] = 1 + (int)(hash_u32(h) % (3 * (k + 1)));
        ph[k] = (hash_u32(h + 17) & 0xffff) / 65535.0 * 6.283185307179586;
    }
    FILE *fp = fopen(path, "w");
    if (!fp) { set_error("cannot write %s", path); return false; }
    fprintf(fp, "vertex %d\nface %d\n", nu * nv, 2 * nu * nv);
    const double TWO_PI = 6.283185307179586;
    auto knot = [&](double t, double *p) {
        double r = 0.055 * (2 + std::cos(3 * t)) / 3.0 + 0.025;
        p[0] = r * std::cos(2 * t) * 1.15;
        p[1] = 0.13 + 0.075 * std::sin(3 * t);
        p[2] = r * std::sin(2 * t) * 0.95;
    };
    for (int i = 0; i < nu; ++i) {
        double t = TWO_PI * i / nu;
        double c[3], c2[3];
        knot(t, c);
        knot(t + 1e-4, c2);
        double T[3] = {c2[0] - c[0], c2[1] - c[1], c2[2] - c[2]};
        double tl = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        for (int k = 0; k < 3; ++k) T[k] /= tl;
        double up[3] = {0, 1, 0};
        double N[3] = {up[1] * T[2] - up[2] * T[1], up[2] * T[0] - up[0] * T[2], up[0] * T[1] - up[1] * T[0]};
        double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        if (nl < 1e-9) { N[0] = 1; N[1] = 0; N[2] = 0; nl = 1; }
        for (int k = 0; k < 3; ++k) N[k] /= nl;
        double B[3] = {T[1] * N[2] - T[2] * N[1], T[2] * N[0] - T[0] * N[2], T[0] * N[1] - T[1] * N[0]};
        for (int j = 0; j < nv; ++j) {
            double s = TWO_PI * j / nv;
            double disp = 1.0;
            for (int k = 0; k < K; ++k) disp += amp[k] * std::sin(fu[k] * t + fv[k] * s + ph[k]);
            double rad = 0.0135 * disp;
            double x = c[0] + rad * (std::cos(s) * N[0] + std::sin(s) * B[0]);
            double y = c[1] + rad * (std::cos(s) * N[1] + std::sin(s) * B[1]);
            double z = c[2] + rad * (std::cos(s) * N[2] + std::sin(s) * B[2]);
            fprintf(fp, "%.6f %.6f %.6f\n", x, y, z);
        }
    }
    for (int i = 0; i < nu; ++i) {
        int i1 = (i + 1) % nu;
        for (int j = 0; j < nv; ++j) {
            int j1 = (j + 1) % nv;
            int a = i * nv + j, b = i1 * nv + j, c = i1 * nv + j1, d = i * nv + j1;
            fprintf(fp, "3 %d %d %d\n", a, b, c);
            fprintf(fp, "3 %d %d %d\n", a, c, d);
        }
    }
    fclose(fp);
    return true;
}

// ---- Radiance RGBE (.hdr) reader.  Produces the floats stbi_loadf (3rd/stb_image.h, used at
// lights/InfiniteAreaLight.cpp:27) returns: value = mantissa * 2^(e - 136), 0 when e == 0, rows top-down.
bool read_rgbe(const char *path, std::vector<float> *rgb, int *w, int *h) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error("cannot open %s", path); return false; }
    char line[512];
    bool ok_magic = false, ok_format = false;
    if (fgets(line, sizeof(line), fp)) ok_magic = (strncmp(line, "#?RADIANCE", 10) == 0 || strncmp(line, "#?RGBE", 6) == 0);
    while (fgets(line, sizeof(line), fp)) {
        if (line[0] == '\n' || line[0] == '\r') break;
        if (strncmp(line, "FORMAT=32-bit_rle_rgbe", 22) == 0) ok_format = true;
    }
    int W = 0, H = 0;
    if (!fgets(line, sizeof(line), fp) || sscanf(line, "-Y %d +X %d", &H, &W) != 2 || !ok_magic || !ok_format || W <= 0 || H <= 0) {
        fclose(fp);
        set_error("%s: unsupported .hdr header", path);
        return false;
    }
    rgb->assign((size_t)W * H * 3, 0.f);
    std::vector<unsigned char> scan((size_t)W * 4);
    auto convert = [](const unsigned char *in, float *out) {
        if (in[3] != 0) {
            float f1 = (float)ldexp(1.0f, in[3] - (int)(128 + 8));
            out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1;
        } else out[0] = out[1] = out[2] = 0;
    };
    bool fail = false;
    for (int j = 0; j < H && !fail; ++j) {
        unsigned char hd[4];
        if (fread(hd, 1, 4, fp) != 4) { fail = true; break; }
        if (W < 8 || W >= 32768 || hd[0] != 2 || hd[1] != 2 || (hd[2] & 0x80)) {
            // flat (non-RLE) data: this pixel is the first of W*H raw pixels
            if (j != 0) { fail = true; break; }
            std::vector<unsigned char> raw((size_t)W * H * 4);
            memcpy(raw.data(), hd, 4);
            if (fread(raw.data() + 4, 1, raw.size() - 4, fp) != raw.size() - 4) { fail = true; break; }
            for (size_t p = 0; p < (size_t)W * H; ++p) convert(&raw[4 * p], &(*rgb)[3 * p]);
            fclose(fp);
            *w = W; *h = H;
            return true;
        }
        if (((hd[2] << 8) | hd[3]) != W) { fail = true; break; }
        for (int k = 0; k < 4 && !fail; ++k) {
            int i = 0;
            while (i < W) {
                int count = fgetc(fp);
                if (count == EOF) { fail = true; break; }
                if (count > 128) {
                    int value = fgetc(fp);
                    count -= 128;
                    if (value == EOF || i + count > W) { fail = true; break; }
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = (unsigned char)value;
                } else {
                    if (count == 0 || i + count > W) { fail = true; break; }
                    for (int z = 0; z < count; ++z) {
                        int value = fgetc(fp);
                        if (value == EOF) { fail = true; break; }
                        scan[(size_t)(i++) * 4 + k] = (unsigned char)value;
                    }
                }
            }
        }
        if (!fail) for (int i = 0; i < W; ++i) convert(&scan[(size_t)i * 4], &(*rgb)[((size_t)j * W + i) * 3]);
    }
    fclose(fp);
    if (fail) { set_error("%s: corrupt .hdr data", path); return false; }
    *w = W; *h = H;
    return true;
}

}  // namespace gnxr

// =====================================================================================
// C ABI: gnxr_builder_* (declared in include/gnxr.h)
// =====================================================================================
using namespace gnxr;
struct gnxr_builder { Builder b; };

static gnxr_material blank_material(int type) {
    gnxr_material m;
    memset(&m, 0, sizeof(m));
    m.type = type;
    m.has_bump = 1;  // every reference material is built with a ConstantTexture<float>(0) bump map
    return m;
}

extern "C" {

int gnxr_builder_create(gnxr_builder **out) {
    if (!out) return GNXR_ERR_INVALID;
    *out = new gnxr_builder();
    return GNXR_OK;
}
void gnxr_builder_destroy(gnxr_builder *b) { delete b; }

int gnxr_builder_add_material(gnxr_builder *b, const gnxr_material *m) {
    if (!b || !m) return GNXR_ERR_INVALID;
    b->b.materials.push_back(*m);
    return (int)b->b.materials.size() - 1;
}
// MatteMaterial(Kd, sigma, bumpMap), ui/RenderThread.cpp:79-99
int gnxr_builder_matte(gnxr_builder *b, const float kd[3], float sigma_deg) {
    gnxr_material m = blank_material(GNXR_MAT_MATTE);
    memcpy(m.kd, kd, 12);
    m.sigma = sigma_deg;
    return gnxr_builder_add_material(b, &m);
}
// MirrorMaterial(Kr, bumpMap), ui/RenderThread.cpp:102
int gnxr_builder_mirror(gnxr_builder *b, const float kr[3]) {
    gnxr_material m = blank_material(GNXR_MAT_MIRROR);
    memcpy(m.kr, kr, 12);
    return gnxr_builder_add_material(b, &m);
}
// getPurplePlasticMaterial, ui/MaterialList.cpp:48-56
int gnxr_builder_purple_plastic(gnxr_builder *b) {
    gnxr_material m = blank_material(GNXR_MAT_PLASTIC);
    float purple[3] = {0.35f, 0.12f, 0.48f};  // Spectrum components assigned from double literals
    purple[0] = (float)0.35; purple[1] = (float)0.12; purple[2] = (float)0.48;
    for (int i = 0; i < 3; ++i) { m.kd[i] = purple[i]; m.ks[i] = 1.f - purple[i]; }
    m.urough = m.vrough = 0.1f;
    m.remap_roughness = 1;
    return gnxr_builder_add_material(b, &m);
}
// getYelloMetalMaterial, ui/MaterialList.cpp:58-69
int gnxr_builder_yellow_metal(gnxr_builder *b) {
    gnxr_material m = blank_material(GNXR_MAT_METAL);
    m.eta[0] = 0.2f; m.eta[1] = 0.2f; m.eta[2] = 0.8f;
    m.k[0] = m.k[1] = m.k[2] = 0.11f;
    m.urough = m.vrough = 0.15f;
    m.remap_roughness = 0;
    return gnxr_builder_add_material(b, &m);
}
// getWhiteGlassMaterial, ui/MaterialList.cpp:71-83
int gnxr_builder_white_glass(gnxr_builder *b) {
    gnxr_material m = blank_material(GNXR_MAT_GLASS);
    for (int i = 0; i < 3; ++i) { m.kr[i] = 0.98f; m.kt[i] = 0.98f; }
    m.eta[0] = 1.5f;
    m.urough = m.vrough = 0.1f;
    m.remap_roughness = 0;
    return gnxr_builder_add_material(b, &m);
}

int gnxr_builder_add_mesh(gnxr_builder *b, const float *vertices, int32_t n_vertices, const int32_t *indices, int32_t n_triangles,
                          const float *o2w16, int32_t material, int32_t medium_inside, int32_t medium_outside) {
    if (!b || !vertices || !indices || n_vertices <= 0 || n_triangles <= 0) return GNXR_ERR_INVALID;
    Xf x;
    if (o2w16) { memcpy(x.m.m, o2w16, 64); x.inv = inverse(x.m); }
    return b->b.add_mesh(vertices, n_vertices, indices, n_triangles, x, material, medium_inside, medium_outside);
}

// AddModel, ui/ModelList.cpp:47-69: tri_Object2World = Translate(0,-2.9,0) * Identity
int gnxr_builder_add_model_3d(gnxr_builder *b, const char *path, int32_t material) {
    if (!b || !path) return GNXR_ERR_INVALID;
    std::vector<float> v;
    std::vector<int32_t> idx;
    if (!read_model_3d(path, &v, &idx)) return GNXR_ERR_IO;
    Xf o2w = xmul(translate(Vec3(0.f, -2.9f, 0.f)), Xf());
    return b->b.add_mesh(v.data(), (int)v.size() / 3, idx.data(), (int)idx.size() / 3, o2w, material, -1, -1);
}

// AddCornell, ui/ModelList.cpp:71-118
int gnxr_builder_add_cornell(gnxr_builder *b, int32_t material1, int32_t material2, int32_t material3) {
    if (!b) return GNXR_ERR_INVALID;
    const int nTrianglesWall = 2 * 5;
    int32_t idx[nTrianglesWall * 3];
    for (int i = 0; i < nTrianglesWall * 3; i++) idx[i] = i;
    const float L = 5.0f;
    const float P[nTrianglesWall * 3][3] = {
        {0.f, 0.f, L}, {L, 0.f, L}, {0.f, 0.f, 0.f}, {L, 0.f, L}, {L, 0.f, 0.f}, {0.f, 0.f, 0.f},      // floor
        {0.f, L, L}, {0.f, L, 0.f}, {L, L, L}, {L, L, L}, {0.f, L, 0.f}, {L, L, 0.f},                  // ceiling
        {0.f, 0.f, 0.f}, {L, 0.f, 0.f}, {L, L, 0.f}, {0.f, 0.f, 0.f}, {L, L, 0.f}, {0.f, L, 0.f},      // back wall
        {0.f, 0.f, 0.f}, {0.f, L, L}, {0.f, 0.f, L}, {0.f, 0.f, 0.f}, {0.f, L, 0.f}, {0.f, L, L},      // right wall
        {L, 0.f, 0.f}, {L, L, L}, {L, 0.f, L}, {L, 0.f, 0.f}, {L, L, 0.f}, {L, L, L}};                 // left wall
    Xf o2w = translate(Vec3(-0.5f * L, -0.5f * L, -0.5f * L));
    int first = b->b.add_mesh(&P[0][0], nTrianglesWall * 3, idx, nTrianglesWall, o2w, material3, -1, -1);
    if (first < 0) return first;
    for (int i = 0; i < nTrianglesWall; ++i) {
        if (i == 6 || i == 7) b->b.tri_material[first + i] = material1;
        else if (i == 8 || i == 9) b->b.tri_material[first + i] = material2;
    }
    return first;
}

// AddFloor, ui/ModelList.cpp:20-45
int gnxr_builder_add_floor(gnxr_builder *b, int32_t material) {
    if (!b) return GNXR_ERR_INVALID;
    int32_t idx[6] = {0, 1, 2, 3, 4, 5};
    const float y = -2.0f;
    const float P[6][3] = {{-6.f, y, 6.f}, {6.f, y, 6.f}, {-6.f, y, -6.f}, {6.f, y, 6.f}, {6.f, y, -6.f}, {-6.f, y, -6.f}};
    return b->b.add_mesh(&P[0][0], 6, idx, 2, Xf(), material, -1, -1);
}

// AddAreaLight, ui/ModelList.cpp:120-147: two triangles at y = 2.45, one DiffuseAreaLight(Le=5) each
int gnxr_builder_add_area_light(gnxr_builder *b, int32_t material) {
    if (!b) return GNXR_ERR_INVALID;
    int32_t idx[6] = {0, 1, 2, 3, 4, 5};
    const float P[6][3] = {{-1.4f, 0.f, 1.4f}, {-1.4f, 0.f, -1.4f}, {1.4f, 0.f, 1.4f}, {1.4f, 0.f, 1.4f}, {-1.4f, 0.f, -1.4f}, {1.4f, 0.f, -1.4f}};
    Xf o2w = translate(Vec3(0.0f, 2.45f, 0.0f));
    int first = b->b.add_mesh(&P[0][0], 6, idx, 2, o2w, material, -1, -1);
    if (first < 0) return first;
    for (int i = 0; i < 2; ++i) {
        gnxr_light l;
        memset(&l, 0, sizeof(l));
        l.type = GNXR_LIGHT_AREA_TRI;
        l.tri = first + i;
        l.two_sided = 0;
        l.n_samples = 5;
        l.le[0] = l.le[1] = l.le[2] = 5.0f;
        b->b.lights.push_back(l);
        b->b.tri_light[first + i] = (int)b->b.lights.size() - 1;
    }
    return first;
}

int gnxr_builder_add_emissive_mesh(gnxr_builder *b, const float *vertices, int32_t n_vertices, const int32_t *indices, int32_t n_triangles,
                                   const float *o2w16, int32_t material, const float lemit[3], int32_t n_samples) {
    if (!b || !lemit) return GNXR_ERR_INVALID;
    int first = gnxr_builder_add_mesh(b, vertices, n_vertices, indices, n_triangles, o2w16, material, -1, -1);
    if (first < 0) return first;
    for (int i = 0; i < n_triangles; ++i) {
        gnxr_light l;
        memset(&l, 0, sizeof(l));
        l.type = GNXR_LIGHT_AREA_TRI;
        l.tri = first + i;
        l.two_sided = 0;
        l.n_samples = n_samples;
        memcpy(l.le, lemit, 12);
        b->b.lights.push_back(l);
        b->b.tri_light[first + i] = (int)b->b.lights.size() - 1;
    }
    return first;
}

// AddSkyLight, ui/ModelList.cpp:163-170 (image "1" never loads -> gradient)
int gnxr_builder_add_sky_light(gnxr_builder *b) {
    if (!b) return GNXR_ERR_INVALID;
    gnxr_light l;
    memset(&l, 0, sizeof(l));
    l.type = GNXR_LIGHT_SKYBOX;
    l.tri = -1;
    l.n_samples = 1;
    l.radius = 10.0f;
    for (int i = 0; i < 4; ++i) l.light_to_world[5 * i] = 1.f;
    b->b.lights.push_back(l);
    return (int)b->b.lights.size() - 1;
}

int gnxr_builder_add_light(gnxr_builder *b, const gnxr_light *l) {
    if (!b || !l) return GNXR_ERR_INVALID;
    if (l->type != GNXR_LIGHT_POINT && l->type != GNXR_LIGHT_SPOT && l->type != GNXR_LIGHT_DISTANT) { set_error("gnxr_builder_add_light: POINT, SPOT or DISTANT"); return GNXR_ERR_INVALID; }
    gnxr_light ll = *l;
    ll.tri = -1;
    b->b.lights.push_back(ll);
    return (int)b->b.lights.size() - 1;
}
// AddSpotLight, ui/ModelList.cpp:149-154: Translate(0, 2.43, 0) * Rotate(90, (1,0,0)), I = 25, totalWidth 45, falloffStart 30
int gnxr_builder_add_spot_light(gnxr_builder *b) {
    if (!b) return GNXR_ERR_INVALID;
    gnxr_light l;
    memset(&l, 0, sizeof(l));
    l.type = GNXR_LIGHT_SPOT; l.tri = -1; l.n_samples = 1;
    l.le[0] = l.le[1] = l.le[2] = 25.0f;
    l.radius = 45.f; l.falloff_start = 30.f;
    Xf t = xmul(translate(Vec3(0.0f, 2.43f, 0.0f)), rotate_axis(90.f, Vec3(1.0f, 0.0f, 0.0f)));
    memcpy(l.light_to_world, &t.m.m[0][0], 64);
    return gnxr_builder_add_light(b, &l);
}
// AddDistLight, ui/ModelList.cpp:156-161: identity, L = 25, wLight = (0, 0, 1)
int gnxr_builder_add_dist_light(gnxr_builder *b) {
    if (!b) return GNXR_ERR_INVALID;
    gnxr_light l;
    memset(&l, 0, sizeof(l));
    l.type = GNXR_LIGHT_DISTANT; l.tri = -1; l.n_samples = 1;
    l.le[0] = l.le[1] = l.le[2] = 25.0f;
    l.center[2] = 1.f;
    for (int i = 0; i < 4; ++i) l.light_to_world[5 * i] = 1.f;
    return gnxr_builder_add_light(b, &l);
}

int gnxr_builder_add_inf_light_data(gnxr_builder *b, const float *rgb, int32_t w, int32_t h, const float *l2w16, const float power[3]) {
    if (!b || !rgb || w <= 0 || h <= 0) return GNXR_ERR_INVALID;
    if (b->b.env_w) { set_error("only one InfiniteAreaLight per scene"); return GNXR_ERR_UNSUPPORTED; }
    b->b.env_rgb.assign(rgb, rgb + (size_t)w * h * 3);
    b->b.env_w = w; b->b.env_h = h;
    gnxr_light l;
    memset(&l, 0, sizeof(l));
    l.type = GNXR_LIGHT_INFINITE;
    l.tri = -1;
    l.n_samples = 10;
    for (int i = 0; i < 3; ++i) l.le[i] = power ? power[i] : 1.f;
    if (l2w16) memcpy(l.light_to_world, l2w16, 64);
    else for (int i = 0; i < 4; ++i) l.light_to_world[5 * i] = 1.f;
    b->b.lights.push_back(l);
    return (int)b->b.lights.size() - 1;
}
// AddInfLight, ui/ModelList.cpp:172-179: RotateX(20) * RotateY(-90) * RotateX(-90), power 1
int gnxr_builder_add_inf_light(gnxr_builder *b, const char *hdr_path) {
    if (!b || !hdr_path) return GNXR_ERR_INVALID;
    std::vector<float> rgb;
    int w, h;
    if (!read_rgbe(hdr_path, &rgb, &w, &h)) return GNXR_ERR_IO;
    Xf t = xmul(xmul(rotate_x(20), rotate_y(-90)), rotate_x(-90));
    float power[3] = {1.f, 1.f, 1.f};
    return gnxr_builder_add_inf_light_data(b, rgb.data(), w, h, &t.m.m[0][0], power);
}

int gnxr_builder_add_medium(gnxr_builder *b, const gnxr_medium *m, const float *density) {
    if (!b || !m) return GNXR_ERR_INVALID;
    gnxr_medium mm = *m;
    if (mm.type == GNXR_MEDIUM_GRID) {
        if (!density || mm.nx <= 0 || mm.ny <= 0 || mm.nz <= 0) return GNXR_ERR_INVALID;
        mm.density_offset = (int64_t)b->b.grid_density.size();
        b->b.grid_density.insert(b->b.grid_density.end(), density, density + (size_t)mm.nx * mm.ny * mm.nz);
    }
    b->b.media.push_back(mm);
    return (int)b->b.media.size() - 1;
}

// GridDensityMedium from a `.volume` text file -- the format of the reference's Resources/density_render.70.volume (which nothing in
// the reference loads: GridDensityMedium is never instantiated, ui/RenderThread.cpp:21 only includes its header): tokens
// `nx N ny N nz N`, `p0 x y z`, `p1 x y z`, `sigma_a r g b`, `sigma_s r g b`, then nx * ny * nz densities, x fastest
// (density[(z * ny + y) * nx + x], media/GridDensityMedium.h:34-38), any whitespace / CRLF.
int gnxr_builder_add_volume_file(gnxr_builder *b, const char *path, float g, float sigma_scale, const float *medium_to_world16) {
    if (!b || !path) return GNXR_ERR_INVALID;
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error("cannot open %s", path); return GNXR_ERR_IO; }
    auto fail = [&](const char *what) { fclose(fp); set_error("%s: %s", path, what); return (int)GNXR_ERR_IO; };
    char key[32];
    int n[3] = {0, 0, 0};
    float p0[3], p1[3], sa[3], ss[3];
    const char *dims[3] = {"nx", "ny", "nz"};
    for (int i = 0; i < 3; ++i)
        if (fscanf(fp, "%31s %d", key, &n[i]) != 2 || strcmp(key, dims[i]) != 0 || n[i] <= 0) return fail("bad .volume header (nx / ny / nz)");
    struct { const char *name; float *v; } rows[4] = {{"p0", p0}, {"p1", p1}, {"sigma_a", sa}, {"sigma_s", ss}};
    for (auto &r : rows)
        if (fscanf(fp, "%31s %f %f %f", key, &r.v[0], &r.v[1], &r.v[2]) != 4 || strcmp(key, r.name) != 0) return fail("bad .volume header (p0 / p1 / sigma_a / sigma_s)");
    // untrusted sizes: each dimension is bounded before the product is formed (three ints near INT_MAX would wrap a 64-bit product),
    // and an allocation failure comes back as a status, never as an exception through the C boundary
    for (int i = 0; i < 3; ++i) if (n[i] > 4096) return fail("density grid dimension above 4096");
    const size_t count = (size_t)n[0] * (size_t)n[1] * (size_t)n[2];
    if (count >= (1ull << 31)) return fail("density grid too large");
    std::vector<float> dens;
    try { dens.resize(count); } catch (const std::bad_alloc &) { fclose(fp); set_error("%s: out of memory for %zu densities", path, count); return GNXR_ERR_OOM; }
    for (size_t i = 0; i < count; ++i)
        if (fscanf(fp, "%f", &dens[i]) != 1) return fail("truncated density data");
    fclose(fp);
    gnxr_medium m;
    memset(&m, 0, sizeof(m));
    m.type = GNXR_MEDIUM_GRID;
    m.nx = n[0]; m.ny = n[1]; m.nz = n[2];
    for (int c = 0; c < 3; ++c) { m.sigma_a[c] = sa[c] * sigma_scale; m.sigma_s[c] = ss[c] * sigma_scale; }
    m.g = g;
    if (medium_to_world16) memcpy(m.medium_to_world, medium_to_world16, 64);
    else {   // the file's own placement: Translate(p0) * Scale(p1 - p0) maps the unit cube of the grid onto [p0, p1]
        const float mm[16] = {p1[0] - p0[0], 0, 0, p0[0], 0, p1[1] - p0[1], 0, p0[1], 0, 0, p1[2] - p0[2], p0[2], 0, 0, 0, 1};
        memcpy(m.medium_to_world, mm, 64);
    }
    return gnxr_builder_add_medium(b, &m, dens.data());
}

// ImageTexture(UVMapping2D, filename, doTrilinear, maxAniso, wrapMode, scale, gamma), textures/ImageTexture.cpp:40-48
int gnxr_builder_add_texture_data(gnxr_builder *b, const gnxr_texture *t, const float *rgb, int32_t w, int32_t h) {
    if (!b || !t || !rgb || w <= 0 || h <= 0) return GNXR_ERR_INVALID;
    if (t->wrap < GNXR_WRAP_REPEAT || t->wrap > GNXR_WRAP_CLAMP) { set_error("unknown ImageWrap %d", t->wrap); return GNXR_ERR_INVALID; }
    gnxr_texture tt = *t;
    tt.width = w; tt.height = h;
    tt.texel_offset = (int64_t)b->b.texels.size();
    b->b.texels.insert(b->b.texels.end(), rgb, rgb + (size_t)w * h * 3);
    b->b.textures.push_back(tt);
    return (int)b->b.textures.size() - 1;
}
int gnxr_builder_add_texture_file(gnxr_builder *b, const gnxr_texture *t, const char *hdr_path) {
    if (!b || !t || !hdr_path) return GNXR_ERR_INVALID;
    std::vector<float> rgb;
    int w = 0, h = 0;
    if (!read_rgbe(hdr_path, &rgb, &w, &h)) return GNXR_ERR_IO;
    return gnxr_builder_add_texture_data(b, t, rgb.data(), w, h);
}
int gnxr_builder_set_material_texture(gnxr_builder *b, int32_t material, int32_t slot, int32_t texture) {
    if (!b || material < 0 || material >= (int)b->b.materials.size() || texture < -1 || texture >= (int)b->b.textures.size() || slot < 0 || slot > 1)
        return GNXR_ERR_INVALID;
    gnxr_material &m = b->b.materials[material];
    if (m.type != GNXR_MAT_PLASTIC && !(m.type == GNXR_MAT_MATTE && slot == 0)) { set_error("image textures: Kd of MATTE, Kd / Ks of PLASTIC"); return GNXR_ERR_UNSUPPORTED; }
    (slot == 0 ? m.kd_texture : m.ks_texture) = texture + 1;
    return GNXR_OK;
}

int gnxr_builder_set_triangle_uv(gnxr_builder *b, int32_t first, int32_t n, const float *tri_uv) {
    if (!b || !tri_uv || first < 0 || n < 0 || (size_t)first + (size_t)n > b->b.indices.size() / 3) return GNXR_ERR_INVALID;
    if (b->b.tri_uv.empty()) {   // Triangle::GetUVs defaults for every triangle so far
        const float def[6] = {0, 0, 1, 0, 1, 1};
        for (size_t t = 0; t < b->b.indices.size() / 3; ++t) b->b.tri_uv.insert(b->b.tri_uv.end(), def, def + 6);
    }
    memcpy(&b->b.tri_uv[(size_t)first * 6], tri_uv, (size_t)n * 6 * sizeof(float));
    return GNXR_OK;
}

int gnxr_builder_set_triangle_normals(gnxr_builder *b, int32_t first, int32_t n, const float *tri_n) {
    if (!b || !tri_n || first < 0 || n < 0 || (size_t)first + (size_t)n > b->b.indices.size() / 3) return GNXR_ERR_INVALID;
    if (b->b.tri_n.empty()) b->b.tri_n.assign(b->b.indices.size() / 3 * 9, 0.f);
    memcpy(&b->b.tri_n[(size_t)first * 9], tri_n, (size_t)n * 9 * sizeof(float));
    return GNXR_OK;
}

int gnxr_builder_set_triangle_tangents(gnxr_builder *b, int32_t first, int32_t n, const float *tri_s) {
    if (!b || !tri_s || first < 0 || n < 0 || (size_t)first + (size_t)n > b->b.indices.size() / 3) return GNXR_ERR_INVALID;
    if (b->b.tri_s.empty()) b->b.tri_s.assign(b->b.indices.size() / 3 * 9, 0.f);
    memcpy(&b->b.tri_s[(size_t)first * 9], tri_s, (size_t)n * 9 * sizeof(float));
    return GNXR_OK;
}

int gnxr_builder_add_sphere(gnxr_builder *b, const float center[3], float radius, int32_t material, int32_t medium_inside, int32_t medium_outside) {
    if (!b || !center || !(radius > 0)) return GNXR_ERR_INVALID;
    gnxr_sphere s;
    memset(&s, 0, sizeof(s));
    memcpy(s.center, center, 12);
    s.radius = radius; s.material = material; s.medium_inside = medium_inside; s.medium_outside = medium_outside;
    b->b.spheres.push_back(s);
    return (int)b->b.spheres.size() - 1;
}

int gnxr_builder_set_camera(gnxr_builder *b, const gnxr_camera *cam) {
    if (!b || !cam) return GNXR_ERR_INVALID;
    b->b.camera = *cam;
    return GNXR_OK;
}

int gnxr_builder_set_bvh_split_method(gnxr_builder *b, int32_t method) {
    if (!b || method < GNXR_BVH_SAH || method > GNXR_BVH_EQUAL_COUNTS) return GNXR_ERR_INVALID;
    b->b.bvh_split_method = method;
    return GNXR_OK;
}

int gnxr_builder_set_camera_medium(gnxr_builder *b, int32_t medium) {
    if (!b || medium < -1 || medium >= (int)b->b.media.size()) return GNXR_ERR_INVALID;
    b->b.camera_medium = medium;
    return GNXR_OK;
}

int gnxr_builder_desc(gnxr_builder *b, gnxr_scene_desc *out) {
    if (!b || !out) return GNXR_ERR_INVALID;
    b->b.fill_desc(out);
    return GNXR_OK;
}

int gnxr_write_synthetic_3d(const char *path, int32_t target_triangles, uint32_t seed) {
    if (!path) return GNXR_ERR_INVALID;
    return write_synthetic_3d(path, target_triangles, seed) ? GNXR_OK : GNXR_ERR_IO;
}

// ---- FrameBuffer::saveToFile: a minimal PNG encoder (RGBA8, filter 0, zlib stream of stored deflate blocks) ----
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void write_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data) {
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    fwrite(head.data(), 1, 4, f);
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    uint32_t crc = crc32_update(0xffffffffu, (const uint8_t *)type, 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    fwrite(tail.data(), 1, 4, f);
}
}  // namespace

int gnxr_framebuffer_save_png(const char *path, const uint8_t *rgba8, int32_t width, int32_t height) {
    if (!path || !rgba8 || width <= 0 || height <= 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s", path); return GNXR_ERR_IO; }
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bit, RGBA, deflate, adaptive, no interlace
    write_chunk(f, "IHDR", ihdr);
    // raw scanlines: filter byte 0 + row
    const size_t stride = (size_t)width * 4, raw_n = (stride + 1) * (size_t)height;
    std::vector<uint8_t> raw(raw_n);
    for (int y = 0; y < height; ++y) {
        raw[(stride + 1) * y] = 0;
        memcpy(&raw[(stride + 1) * y + 1], rgba8 + stride * y, stride);
    }
    std::vector<uint8_t> z;
    z.reserve(raw_n + raw_n / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;   // Adler-32
    for (size_t pos = 0; pos < raw_n;) {
        size_t n = std::min<size_t>(65535, raw_n - pos);
        z.push_back(pos + n == raw_n ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    write_chunk(f, "IDAT", z);
    write_chunk(f, "IEND", {});
    bool ok = fclose(f) == 0;
    if (!ok) { set_error("write to %s failed", path); return GNXR_ERR_IO; }
    return GNXR_OK;
}

const char *gnxr_last_error(void) { return gnxr::get_error(); }
// sizeof() of the ABI structs, in the order they appear in include/gnxr.h (binding self-check)
int gnxr_abi_sizeof(int which) {
    switch (which) {
    case 0: return (int)sizeof(gnxr_material);
    case 1: return (int)sizeof(gnxr_light);
    case 2: return (int)sizeof(gnxr_camera);
    case 3: return (int)sizeof(gnxr_medium);
    case 4: return (int)sizeof(gnxr_scene_desc);
    case 5: return (int)sizeof(gnxr_render_params);
    case 6: return (int)sizeof(gnxr_stats);
    case 7: return (int)sizeof(gnxr_ray);
    case 8: return (int)sizeof(gnxr_hit);
    case 9: return (int)sizeof(gnxr_sphere);
    case 10: return (int)sizeof(gnxr_texture);
    default: return -1;
    }
}
int gnxr_abi_version(void) { return GNXR_ABI_VERSION; }

}  // extern "C"
