// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_scene.h: CPU restatement of the geometric side of the hot path:
//   BVHAccel build (SAH) / flatten / Intersect / IntersectP   accelerator/BVHAccel.cpp:147-189,201-367,628-729
//   Triangle::Intersect / IntersectP / Sample / Area           shape/Triangle.cpp:71-303,305-453,455-492
//   GeometricPrimitive::Intersect                              core/Primitive.cpp:32-46
//   SurfaceInteraction ctor / SetShadingGeometry               core/Interaction.cpp:8-54
//   Interaction::SpawnRay / SpawnRayTo                         core/Interaction.h:33-53
//   Shape::Sample(ref,u,pdf) / Pdf(ref,wi)                     core/Shape.cpp:21-53
//   PerspectiveCamera::GenerateRayDifferential                 camera/Perspective.cpp:62-135
#pragma once
#include <atomic>
#include <vector>

#include "../include/gnxr.h"
#include "o_sampler.h"
#include "o_texture.h"

namespace gnxo {

// What the integrators need from Interaction / SurfaceInteraction (core/Interaction.h).
struct Interaction {
    V3 p, pError, wo, n;
    int mediumInside = -1, mediumOutside = -1;  // MediumInterface
    bool IsSurfaceInteraction() const { return n != V3(); }
    int GetMedium(const V3 &w) const { return Dot(w, n) > 0 ? mediumOutside : mediumInside; }
    // Interaction.h:33-37
    Ray SpawnRay(const V3 &d) const {
        V3 o = OffsetRayOrigin(p, pError, n, d);
        return Ray(o, d, Infinity, GetMedium(d));
    }
    // Interaction.h:39-44
    Ray SpawnRayTo(const V3 &p2) const {
        V3 origin = OffsetRayOrigin(p, pError, n, p2 - p);
        V3 d = p2 - p;
        return Ray(origin, d, 1 - ShadowEpsilon, GetMedium(d));
    }
    // Interaction.h:46-53
    Ray SpawnRayTo(const Interaction &it) const {
        V3 origin = OffsetRayOrigin(p, pError, n, it.p - p);
        V3 target = OffsetRayOrigin(it.p, it.pError, it.n, origin - it.p);
        V3 d = target - origin;
        return Ray(origin, d, 1 - ShadowEpsilon, GetMedium(d));
    }
};

struct SurfaceInteraction : Interaction {
    P2 uv;
    V3 dpdu, dpdv;
    V3 sn, sdpdu, sdpdv;  // shading.n / dpdu / dpdv
    V3 dndu, dndv;        // shading.dndu / dndv: zero unless the triangle has per-vertex normals (Triangle.cpp:262-292)
    int prim = -1;         // triangle index in AUTHORING order (desc order)
    Float b0 = 0, b1 = 0, b2 = 0, t = 0;
    // SurfaceInteraction::ComputeDifferentials, Interaction.cpp:65-112 (mutable members there)
    V3 dpdx, dpdy;
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
    static bool SolveLinearSystem2x2(const Float A[2][2], const Float B[2], Float *x0, Float *x1) {   // Transform.cpp:12-20
        Float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        if (std::abs(det) < 1e-10f) return false;
        *x0 = (A[1][1] * B[0] - A[0][1] * B[1]) / det;
        *x1 = (A[0][0] * B[1] - A[1][0] * B[0]) / det;
        if (std::isnan(*x0) || std::isnan(*x1)) return false;
        return true;
    }
    void ComputeDifferentials(const Ray &ray) {
        bool ok = ray.hasDifferentials;
        if (ok) {
            Float d = Dot(n, V3(p.x, p.y, p.z));
            Float tx = -(Dot(n, ray.rxOrigin) - d) / Dot(n, ray.rxDirection);
            Float ty = 0;
            if (std::isinf(tx) || std::isnan(tx)) ok = false;
            V3 px, py;
            if (ok) {
                px = ray.rxOrigin + tx * ray.rxDirection;
                ty = -(Dot(n, ray.ryOrigin) - d) / Dot(n, ray.ryDirection);
                if (std::isinf(ty) || std::isnan(ty)) ok = false;
            }
            if (ok) {
                py = ray.ryOrigin + ty * ray.ryDirection;
                dpdx = px - p;
                dpdy = py - p;
                int dim[2];
                if (std::abs(n.x) > std::abs(n.y) && std::abs(n.x) > std::abs(n.z)) { dim[0] = 1; dim[1] = 2; }
                else if (std::abs(n.y) > std::abs(n.z)) { dim[0] = 0; dim[1] = 2; }
                else { dim[0] = 0; dim[1] = 1; }
                Float A[2][2] = {{dpdu[dim[0]], dpdv[dim[0]]}, {dpdu[dim[1]], dpdv[dim[1]]}};
                Float Bx[2] = {px[dim[0]] - p[dim[0]], px[dim[1]] - p[dim[1]]};
                Float By[2] = {py[dim[0]] - p[dim[0]], py[dim[1]] - p[dim[1]]};
                if (!SolveLinearSystem2x2(A, Bx, &dudx, &dvdx)) dudx = dvdx = 0;
                if (!SolveLinearSystem2x2(A, By, &dudy, &dvdy)) dudy = dvdy = 0;
                return;
            }
        }
        // no differentials, or the `fail:` label (an auxiliary ray parallel to the surface)
        dudx = dvdx = 0;
        dudy = dvdy = 0;
        dpdx = dpdy = V3(0, 0, 0);
    }
    // Interaction.cpp:36-54 (dndu/dndv are zero on this path)
    void SetShadingGeometry(const V3 &dpdus, const V3 &dpdvs, bool orientationIsAuthoritative) {
        sn = Normalize(Cross(dpdus, dpdvs));
        if (orientationIsAuthoritative) n = Faceforward(n, sn);
        else sn = Faceforward(sn, n);
        sdpdu = dpdus;
        sdpdv = dpdvs;
    }
};

struct LinearBVHNode {  // BVHAccel.cpp:54-65
    Bounds3 bounds;
    int offset;  // primitivesOffset (leaf) / secondChildOffset (interior)
    uint16_t nPrimitives;
    uint8_t axis;
    uint8_t pad;
};

struct TraversalCounters {
    std::atomic<uint64_t> nIntersect{0}, nIntersectP{0}, nNodes{0}, nTris{0};
};

struct Scene {
    std::vector<V3> verts;
    std::vector<int> indices;      // authoring order
    std::vector<int> triMaterial, triLight, triMedIn, triMedOut;
    std::vector<gnxr_material> materials;
    std::vector<gnxr_light> lights;
    std::vector<gnxr_medium> media;
    std::vector<float> gridDensity;
    std::vector<float> envRgb;
    int envW = 0, envH = 0;
    gnxr_camera camera;
    int cameraMedium = -1;
    std::vector<gnxr_sphere> spheres;   // prim index = nTriangles() + sphere index; tested before the triangle BVH
    std::vector<ImageTexture> textures; // gnxr_material::kd_texture / ks_texture - 1
    std::vector<float> triUV;           // empty (Triangle::GetUVs defaults) or 6 floats per triangle, authoring order
    std::vector<float> triN;            // empty or 9 floats per triangle (TriangleMesh::n through the indices; zeros == none)
    std::vector<float> triS;            // the same for TriangleMesh::s
    // BVH
    std::vector<LinearBVHNode> nodes;
    std::vector<int> orderedPrims;  // BVH leaf order -> authoring index (primitives.swap(orderedPrims), BVHAccel.cpp:171)
    int bvhMaxDepth = 0;
    mutable TraversalCounters counters;
    bool countTraversal = false;

    int nTriangles() const { return (int)indices.size() / 3; }
    void Tri(int i, V3 *p0, V3 *p1, V3 *p2) const {
        *p0 = verts[indices[3 * i]]; *p1 = verts[indices[3 * i + 1]]; *p2 = verts[indices[3 * i + 2]];
    }
    Bounds3 TriBound(int i) const {  // Triangle::WorldBound, Triangle.cpp:62-69
        V3 p0, p1, p2; Tri(i, &p0, &p1, &p2);
        return Union(Bounds3(p0, p1), p2);
    }
    Bounds3 WorldBound() const { return nodes.empty() ? sphereOnlyBound : nodes[0].bounds; }

    void Load(const gnxr_scene_desc *d) {
        verts.resize(d->n_vertices);
        for (int i = 0; i < d->n_vertices; ++i) verts[i] = V3(d->vertices[3 * i], d->vertices[3 * i + 1], d->vertices[3 * i + 2]);
        indices.assign(d->indices, d->indices + 3 * (size_t)d->n_triangles);
        triMaterial.assign(d->tri_material, d->tri_material + d->n_triangles);
        triLight.assign(d->tri_light, d->tri_light + d->n_triangles);
        if (d->tri_medium_inside) triMedIn.assign(d->tri_medium_inside, d->tri_medium_inside + d->n_triangles);
        else triMedIn.assign(d->n_triangles, -1);
        if (d->tri_medium_outside) triMedOut.assign(d->tri_medium_outside, d->tri_medium_outside + d->n_triangles);
        else triMedOut.assign(d->n_triangles, -1);
        materials.assign(d->materials, d->materials + d->n_materials);
        lights.assign(d->lights, d->lights + d->n_lights);
        if (d->n_media) media.assign(d->media, d->media + d->n_media);
        if (d->grid_density && d->n_media) {
            int64_t total = 0;
            for (auto &m : media) if (m.type == GNXR_MEDIUM_GRID) total = std::max<int64_t>(total, m.density_offset + (int64_t)m.nx * m.ny * m.nz);
            gridDensity.assign(d->grid_density, d->grid_density + total);
        }
        envW = d->env_width; envH = d->env_height;
        if (envW && envH) envRgb.assign(d->env_rgb, d->env_rgb + (size_t)envW * envH * 3);
        camera = d->camera;
        cameraMedium = d->camera_medium;
        if (d->tri_uv) triUV.assign(d->tri_uv, d->tri_uv + 6 * (size_t)d->n_triangles);
        if (d->tri_n) triN.assign(d->tri_n, d->tri_n + 9 * (size_t)d->n_triangles);
        if (d->tri_s) triS.assign(d->tri_s, d->tri_s + 9 * (size_t)d->n_triangles);
        textures.resize(d->n_textures);
        for (int i = 0; i < d->n_textures; ++i) textures[i].Build(d->textures[i], d->texels + d->textures[i].texel_offset);
        if (d->n_spheres > 0) {
            spheres.assign(d->spheres, d->spheres + d->n_spheres);
            for (const gnxr_sphere &sp : spheres) {   // per-primitive tables continue past the triangles
                triMaterial.push_back(sp.material); triLight.push_back(-1);
                triMedIn.push_back(sp.medium_inside); triMedOut.push_back(sp.medium_outside);
            }
        }
        BuildBVH();
        for (const gnxr_sphere &sp : spheres) {       // Scene::WorldBound covers every primitive
            V3 c(sp.center[0], sp.center[1], sp.center[2]), r(sp.radius, sp.radius, sp.radius);
            Bounds3 b(c - r, c + r);
            if (nodes.empty()) { LinearBVHNode n; n.bounds = b; n.offset = 0; n.nPrimitives = 0; n.axis = 0; sphereOnlyBound = Union(sphereOnlyBound, b); }
            else nodes[0].bounds = Union(nodes[0].bounds, b);
        }
    }
    Bounds3 sphereOnlyBound;

    // ---------------- BVH build: BVHAccel.cpp:147-189, 201-367 (SAH, maxPrimsInNode = 1) -----------
    struct PrimInfo { size_t primitiveNumber; Bounds3 bounds; V3 centroid; };
    struct BuildNode { Bounds3 bounds; BuildNode *children[2]; int splitAxis, firstPrimOffset, nPrimitives; };
    std::vector<BuildNode *> buildPool;
    BuildNode *NewNode() { buildPool.push_back(new BuildNode()); return buildPool.back(); }

    BuildNode *recursiveBuild(std::vector<PrimInfo> &primitiveInfo, int start, int end, int *totalNodes) {
        const int maxPrimsInNode = 1;
        BuildNode *node = NewNode();
        (*totalNodes)++;
        Bounds3 bounds;
        for (int i = start; i < end; ++i) bounds = Union(bounds, primitiveInfo[i].bounds);
        int nPrimitives = end - start;
        auto makeLeaf = [&]() {
            int firstPrimOffset = (int)orderedPrims.size();
            for (int i = start; i < end; ++i) orderedPrims.push_back((int)primitiveInfo[i].primitiveNumber);
            node->firstPrimOffset = firstPrimOffset; node->nPrimitives = nPrimitives; node->bounds = bounds;
            node->children[0] = node->children[1] = nullptr;
            return node;
        };
        if (nPrimitives == 1) return makeLeaf();
        Bounds3 centroidBounds;
        for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, primitiveInfo[i].centroid);
        int dim = centroidBounds.MaximumExtent();
        int mid = (start + end) / 2;
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) return makeLeaf();
        if (nPrimitives <= 2) {
            mid = (start + end) / 2;
            std::nth_element(&primitiveInfo[start], &primitiveInfo[mid], &primitiveInfo[end - 1] + 1,
                             [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
        } else {
            constexpr int nBuckets = 12;
            struct BucketInfo { int count = 0; Bounds3 bounds; };
            BucketInfo buckets[nBuckets];
            for (int i = start; i < end; ++i) {
                int b = nBuckets * centroidBounds.Offset(primitiveInfo[i].centroid)[dim];
                if (b == nBuckets) b = nBuckets - 1;
                buckets[b].count++;
                buckets[b].bounds = Union(buckets[b].bounds, primitiveInfo[i].bounds);
            }
            Float cost[nBuckets - 1];
            for (int i = 0; i < nBuckets - 1; ++i) {
                Bounds3 b0, b1;
                int count0 = 0, count1 = 0;
                for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
                for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
                cost[i] = 1 + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
            }
            Float minCost = cost[0];
            int minCostSplitBucket = 0;
            for (int i = 1; i < nBuckets - 1; ++i)
                if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
            Float leafCost = nPrimitives;
            if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
                PrimInfo *pmid = std::partition(&primitiveInfo[start], &primitiveInfo[end - 1] + 1, [=](const PrimInfo &pi) {
                    int b = nBuckets * centroidBounds.Offset(pi.centroid)[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    return b <= minCostSplitBucket;
                });
                mid = (int)(pmid - &primitiveInfo[0]);
            } else
                return makeLeaf();
        }
        // `node->InitInterior(dim, recursiveBuild(left), recursiveBuild(right))` (BVHAccel.cpp:359-363): g++
        // evaluates call arguments right to left, so the reference builds the RIGHT subtree first and
        // orderedPrims fills from there.  Node (DFS) order is unaffected.
        BuildNode *c1 = recursiveBuild(primitiveInfo, mid, end, totalNodes);
        BuildNode *c0 = recursiveBuild(primitiveInfo, start, mid, totalNodes);
        node->children[0] = c0; node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
        node->splitAxis = dim; node->nPrimitives = 0;
        return node;
    }
    // BVHAccel.cpp:628-646
    int flatten(BuildNode *node, int *offset, int depth) {
        bvhMaxDepth = std::max(bvhMaxDepth, depth);
        LinearBVHNode *ln = &nodes[*offset];
        ln->bounds = node->bounds;
        int myOffset = (*offset)++;
        if (node->nPrimitives > 0) {
            ln->offset = node->firstPrimOffset;
            ln->nPrimitives = (uint16_t)node->nPrimitives;
            ln->axis = 0;
        } else {
            ln->axis = (uint8_t)node->splitAxis;
            ln->nPrimitives = 0;
            flatten(node->children[0], offset, depth + 1);
            ln->offset = flatten(node->children[1], offset, depth + 1);
        }
        ln->pad = 0;
        return myOffset;
    }
    void BuildBVH() {
        nodes.clear(); orderedPrims.clear(); bvhMaxDepth = 0;
        int n = nTriangles();
        if (n == 0) return;
        std::vector<PrimInfo> info(n);
        for (int i = 0; i < n; ++i) {
            Bounds3 b = TriBound(i);
            info[i].primitiveNumber = i;
            info[i].bounds = b;
            info[i].centroid = .5f * b.pMin + .5f * b.pMax;  // BVHAccel.cpp:16
        }
        int totalNodes = 0;
        orderedPrims.reserve(n);
        BuildNode *root = recursiveBuild(info, 0, n, &totalNodes);
        nodes.resize(totalNodes);
        int offset = 0;
        flatten(root, &offset, 0);
        for (BuildNode *b : buildPool) delete b;
        buildPool.clear();
    }

    // ---------------- Triangle::Intersect, Triangle.cpp:71-303 + GeometricPrimitive::Intersect -------------
    // `authoringIndex` selects the triangle; on a hit ray.tMax is shrunk (Primitive.cpp:36).
    bool TriIntersect(int tri, const Ray &ray, SurfaceInteraction *isect) const {
        V3 p0, p1, p2; Tri(tri, &p0, &p1, &p2);
        V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
        int kz = MaxDimension(Abs(ray.d));
        int kx = kz + 1; if (kx == 3) kx = 0;
        int ky = kx + 1; if (ky == 3) ky = 0;
        V3 d = Permute(ray.d, kx, ky, kz);
        p0t = Permute(p0t, kx, ky, kz); p1t = Permute(p1t, kx, ky, kz); p2t = Permute(p2t, kx, ky, kz);
        Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
        p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
        p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
        p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y;
            double p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y;
            double p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y;
            double p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0) return false;
        p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
        Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0 && (tScaled >= 0 || tScaled < ray.tMax * det)) return false;
        else if (det > 0 && (tScaled <= 0 || tScaled > ray.tMax * det)) return false;
        Float invDet = 1 / det;
        Float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
        Float t = tScaled * invDet;
        float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
        float deltaZ = gamma(3) * maxZt;
        float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
        float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
        float deltaX = gamma(5) * (maxXt + maxZt);
        float deltaY = gamma(5) * (maxYt + maxZt);
        float deltaE = 2 * (gamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
        float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
        float deltaT = 3 * (gamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * std::abs(invDet);
        if (t <= deltaT) return false;
        if (!isect) { return true; }  // IntersectP stops here (Triangle.cpp:305-453 has no alpha mask on this path)

        // partial derivatives; GetUVs: mesh->uv through the indices, else (0,0),(1,0),(1,1), Triangle.h:60-74
        V3 dpdu, dpdv;
        P2 uv[3] = {P2(0, 0), P2(1, 0), P2(1, 1)};
        if (!triUV.empty()) for (int k = 0; k < 3; ++k) uv[k] = P2(triUV[6 * (size_t)tri + 2 * k], triUV[6 * (size_t)tri + 2 * k + 1]);
        Float duv02[2] = {uv[0].x - uv[2].x, uv[0].y - uv[2].y}, duv12[2] = {uv[1].x - uv[2].x, uv[1].y - uv[2].y};
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerateUV = std::abs(determinant) < 1e-8;
        if (!degenerateUV) {
            Float invdet = 1 / determinant;
            dpdu = (duv12[1] * dp02 - duv02[1] * dp12) * invdet;
            dpdv = (-duv12[0] * dp02 + duv02[0] * dp12) * invdet;
        }
        if (degenerateUV || Cross(dpdu, dpdv).LengthSquared() == 0) {
            V3 ng = Cross(p2 - p0, p1 - p0);
            if (ng.LengthSquared() == 0) return false;
            CoordinateSystem(Normalize(ng), &dpdu, &dpdv);
        }
        float xAbsSum = (std::abs(b0 * p0.x) + std::abs(b1 * p1.x) + std::abs(b2 * p2.x));
        float yAbsSum = (std::abs(b0 * p0.y) + std::abs(b1 * p1.y) + std::abs(b2 * p2.y));
        float zAbsSum = (std::abs(b0 * p0.z) + std::abs(b1 * p1.z) + std::abs(b2 * p2.z));
        V3 pError = gamma(7) * V3(xAbsSum, yAbsSum, zAbsSum);
        P2 uvHit(b0 * uv[0].x + b1 * uv[1].x + b2 * uv[2].x, b0 * uv[0].y + b1 * uv[1].y + b2 * uv[2].y);
        V3 pHit = b0 * p0 + b1 * p1 + b2 * p2;

        // SurfaceInteraction ctor, Interaction.cpp:8-34 (n from Cross(dpdu,dpdv) is overridden just below)
        SurfaceInteraction si;
        si.p = pHit; si.pError = pError; si.uv = uvHit; si.wo = Normalize(-ray.d);
        si.dpdu = dpdu; si.dpdv = dpdv; si.sdpdu = dpdu; si.sdpdv = dpdv;
        // Triangle.cpp:223-226 (no reverseOrientation / handedness swap on this path)
        si.n = si.sn = Normalize(Cross(dp02, dp12));
        si.p = b0 * p0 + b1 * p1 + b2 * p2;
        // shading geometry of a triangle with per-vertex normals and / or tangents, Triangle.cpp:228-297 (no reverseOrientation)
        {
            auto corner = [&](const std::vector<float> &tab, V3 *a, V3 *b, V3 *c) {
                if (tab.empty()) return false;
                const float *q = &tab[9 * (size_t)tri];
                *a = V3(q[0], q[1], q[2]); *b = V3(q[3], q[4], q[5]); *c = V3(q[6], q[7], q[8]);
                return *a != V3() || *b != V3() || *c != V3();
            };
            V3 n0, n1, n2, s0, s1, s2;
            const bool hasN = corner(triN, &n0, &n1, &n2), hasS = corner(triS, &s0, &s1, &s2);
            if (hasN || hasS) {
                V3 ns;
                if (hasN) {
                    ns = (b0 * n0 + b1 * n1 + b2 * n2);
                    if (ns.LengthSquared() > 0) ns = Normalize(ns);
                    else ns = si.n;
                } else ns = si.n;
                V3 ss;
                if (hasS) {
                    ss = (b0 * s0 + b1 * s1 + b2 * s2);
                    if (ss.LengthSquared() > 0) ss = Normalize(ss);
                    else ss = Normalize(si.dpdu);
                } else ss = Normalize(si.dpdu);
                V3 ts = Cross(ss, ns);
                if (ts.LengthSquared() > 0.f) {
                    ts = Normalize(ts);
                    ss = Cross(ts, ns);
                } else CoordinateSystem(ns, &ss, &ts);
                V3 dndu, dndv;
                if (hasN) {
                    V3 dn1 = n0 - n2, dn2 = n1 - n2;
                    Float determinantN = duv02[0] * duv12[1] - duv02[1] * duv12[0];
                    bool degenerateUVN = std::abs(determinantN) < 1e-8;
                    if (degenerateUVN) {
                        V3 dn = Cross(n2 - n0, n1 - n0);
                        if (dn.LengthSquared() == 0) dndu = dndv = V3(0, 0, 0);
                        else CoordinateSystem(dn, &dndu, &dndv);
                    } else {
                        Float invDet = 1 / determinantN;
                        dndu = (duv12[1] * dn1 - duv02[1] * dn2) * invDet;
                        dndv = (-duv12[0] * dn1 + duv02[0] * dn2) * invDet;
                    }
                } else dndu = dndv = V3(0, 0, 0);
                si.dndu = dndu; si.dndv = dndv;
                si.SetShadingGeometry(ss, ts, true);   // shading.n = Normalize(Cross(ss, ts)); n = Faceforward(n, shading.n)
            }
        }
        si.prim = tri; si.b0 = b0; si.b1 = b1; si.b2 = b2; si.t = t;
        // GeometricPrimitive::Intersect, Primitive.cpp:32-46
        ray.tMax = t;
        int mi = triMedIn[tri], mo = triMedOut[tri];
        if (mi != mo) { si.mediumInside = mi; si.mediumOutside = mo; }  // IsMediumTransition
        else { si.mediumInside = si.mediumOutside = ray.medium; }
        *isect = si;
        return true;
    }

    // ---------------- Sphere::Intersect: pbrt-v3's quadratic sphere (the reference's shape/Sphere.h is an unfinished stub;
    // PARITY UNPINNED).  Full sphere, ObjectToWorld = Translate(center).  The quadratic is solved in double precision (in
    // place of pbrt's EFloat error intervals), everything else follows pbrt-v3 src/shapes/sphere.cpp: nearest root in
    // (0, tMax], hit point re-projected onto the sphere, (phi, theta) parameterisation, dpdu / dpdv, pError = gamma(5)|p|,
    // and the SurfaceInteraction is carried to world space with Transform's error bound for the translation.
    bool SphereIntersect(int si, const Ray &ray, SurfaceInteraction *isect) const {
        const gnxr_sphere &sp = spheres[si];
        const Float radius = sp.radius;
        V3 c(sp.center[0], sp.center[1], sp.center[2]);
        V3 o = ray.o - c, d = ray.d;   // WorldToObject = Translate(-center)
        double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
        double a = dx * dx + dy * dy + dz * dz;
        double b = 2 * (dx * ox + dy * oy + dz * oz);
        double cc = ox * ox + oy * oy + oz * oz - (double)radius * (double)radius;
        double discrim = b * b - 4 * a * cc;
        if (discrim < 0) return false;
        double rootDiscrim = std::sqrt(discrim);
        double q = (b < 0) ? -.5 * (b - rootDiscrim) : -.5 * (b + rootDiscrim);
        double t0 = q / a, t1 = cc / q;
        if (t0 > t1) std::swap(t0, t1);
        if (!(t0 <= (double)ray.tMax) || !(t1 > 0)) return false;
        double tShapeHit = t0;
        if (tShapeHit <= 0) {
            tShapeHit = t1;
            if (tShapeHit > (double)ray.tMax) return false;
        }
        Float tHit = (Float)tShapeHit;
        if (!isect) { ray.tMax = tHit; return true; }
        V3 pHit = o + d * tHit;
        pHit = pHit * (radius / pHit.Length());
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
        Float phi = std::atan2(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
        const Float phiMax = 2 * Pi, thetaMin = Pi, thetaMax = 0;   // zMin = -r, zMax = r
        Float u = phi / phiMax;
        Float theta = std::acos(Clamp(pHit.z / radius, -1, 1));
        Float v = (theta - thetaMin) / (thetaMax - thetaMin);
        Float zRadius = std::sqrt(pHit.x * pHit.x + pHit.y * pHit.y);
        Float invZRadius = 1 / zRadius;
        Float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
        V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
        V3 dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * std::sin(theta));
        V3 pError = gamma(5) * Abs(pHit);
        // (*ObjectToWorld)(SurfaceInteraction): Transform.cpp, point with error (Transform.h:285-308), normals re-normalised
        SurfaceInteraction s;
        s.p = pHit + c;
        s.pError = V3((gamma(3) + 1) * pError.x + gamma(3) * (std::abs(pHit.x) + std::abs(c.x)),
                      (gamma(3) + 1) * pError.y + gamma(3) * (std::abs(pHit.y) + std::abs(c.y)),
                      (gamma(3) + 1) * pError.z + gamma(3) * (std::abs(pHit.z) + std::abs(c.z)));
        s.uv = P2(u, v);
        s.wo = Normalize(-ray.d);
        s.dpdu = dpdu; s.dpdv = dpdv; s.sdpdu = dpdu; s.sdpdv = dpdv;
        s.n = s.sn = Normalize(Normalize(Cross(dpdu, dpdv)));
        s.prim = nTriangles() + si; s.b0 = s.b1 = s.b2 = 0; s.t = tHit;
        ray.tMax = tHit;   // GeometricPrimitive::Intersect, Primitive.cpp:32-46
        int mi = sp.medium_inside, mo = sp.medium_outside;
        if (mi != mo) { s.mediumInside = mi; s.mediumOutside = mo; }
        else { s.mediumInside = s.mediumOutside = ray.medium; }
        *isect = s;
        return true;
    }

    // BVHAccel::Intersect, BVHAccel.cpp:653-691 (spheres, which live outside the triangle BVH, are tested first)
    bool Intersect(const Ray &ray, SurfaceInteraction *isect) const {
        counters.nIntersect.fetch_add(1, std::memory_order_relaxed);
        bool hit = false;
        for (int si = 0; si < (int)spheres.size(); ++si) if (SphereIntersect(si, ray, isect)) hit = true;
        if (nodes.empty()) return hit;
        V3 invDir(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int toVisitOffset = 0, currentNodeIndex = 0;
        int nodesToVisit[64];
        uint64_t nn = 0, nt = 0;
        while (true) {
            const LinearBVHNode *node = &nodes[currentNodeIndex];
            ++nn;
            if (SlabTest(node->bounds, ray, invDir, dirIsNeg)) {
                if (node->nPrimitives > 0) {
                    for (int i = 0; i < node->nPrimitives; ++i) {
                        ++nt;
                        if (TriIntersect(orderedPrims[node->offset + i], ray, isect)) hit = true;
                    }
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->axis]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        if (countTraversal) { counters.nNodes.fetch_add(nn, std::memory_order_relaxed); counters.nTris.fetch_add(nt, std::memory_order_relaxed); }
        return hit;
    }
    // BVHAccel::IntersectP, BVHAccel.cpp:693-729
    bool IntersectP(const Ray &ray) const {
        counters.nIntersectP.fetch_add(1, std::memory_order_relaxed);
        for (int si = 0; si < (int)spheres.size(); ++si) {
            Ray r2 = ray;
            if (SphereIntersect(si, r2, nullptr)) return true;
        }
        if (nodes.empty()) return false;
        V3 invDir(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int nodesToVisit[64];
        int toVisitOffset = 0, currentNodeIndex = 0;
        uint64_t nn = 0, nt = 0;
        bool result = false;
        while (true) {
            const LinearBVHNode *node = &nodes[currentNodeIndex];
            ++nn;
            if (SlabTest(node->bounds, ray, invDir, dirIsNeg)) {
                if (node->nPrimitives > 0) {
                    bool any = false;
                    for (int i = 0; i < node->nPrimitives; ++i) {
                        ++nt;
                        if (TriIntersect(orderedPrims[node->offset + i], ray, nullptr)) { any = true; break; }
                    }
                    if (any) { result = true; break; }
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->axis]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        if (countTraversal) { counters.nNodes.fetch_add(nn, std::memory_order_relaxed); counters.nTris.fetch_add(nt, std::memory_order_relaxed); }
        return result;
    }

    // Triangle::Area, Triangle.cpp:455-462
    Float TriArea(int tri) const {
        V3 p0, p1, p2; Tri(tri, &p0, &p1, &p2);
        return 0.5 * Cross(p1 - p0, p2 - p0).Length();
    }
    // Triangle::Sample(u,pdf), Triangle.cpp:464-492
    Interaction TriSample(int tri, const P2 &u, Float *pdf) const {
        P2 b = UniformSampleTriangle(u);
        V3 p0, p1, p2; Tri(tri, &p0, &p1, &p2);
        Interaction it;
        it.p = b.x * p0 + b.y * p1 + (1 - b.x - b.y) * p2;
        it.n = Normalize(Cross(p1 - p0, p2 - p0));
        V3 pAbsSum = Abs(b.x * p0) + Abs(b.y * p1) + Abs((1 - b.x - b.y) * p2);
        it.pError = gamma(6) * V3(pAbsSum.x, pAbsSum.y, pAbsSum.z);
        *pdf = 1 / TriArea(tri);
        return it;
    }
    // Shape::Sample(ref,u,pdf), Shape.cpp:21-35
    Interaction ShapeSample(int tri, const Interaction &ref, const P2 &u, Float *pdf) const {
        Interaction intr = TriSample(tri, u, pdf);
        V3 wi = intr.p - ref.p;
        if (wi.LengthSquared() == 0) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
            if (std::isinf(*pdf)) *pdf = 0.f;
        }
        return intr;
    }
    // Shape::Pdf(ref,wi), Shape.cpp:37-53: re-intersects the single light triangle
    Float ShapePdf(int tri, const Interaction &ref, const V3 &wi) const {
        Ray ray = ref.SpawnRay(wi);
        SurfaceInteraction isectLight;
        if (!TriIntersect(tri, ray, &isectLight)) return 0;
        Float pdf = DistanceSquared(ref.p, isectLight.p) / (AbsDot(isectLight.n, -wi) * TriArea(tri));
        if (std::isinf(pdf)) pdf = 0.f;
        return pdf;
    }
};

// ---------------- Perspective camera, camera/Perspective.cpp + core/Camera.h:54-75 -----------------
struct Camera {
    M44 rasterToCamera, cameraToWorld;
    V3 dxCamera, dyCamera;
    Float lensRadius, focalDistance;
    int medium = -1;
    bool orthographic = false;   // OrthographicCamera, camera/Orthographic.{h,cpp}
    Camera() {}
    Camera(const gnxr_camera &c, int W, int H) {
        // RenderThread.cpp:62-68
        Xform lookat = LookAt(V3(c.eye[0], c.eye[1], c.eye[2]), V3(c.look[0], c.look[1], c.look[2]), V3(c.up[0], c.up[1], c.up[2]));
        cameraToWorld = lookat.mInv;  // Inverse(lookat)
        // Perspective.cpp:114-135
        float frame = (float)W / (float)H;
        float sxmin, sxmax, symin, symax;
        if (frame > 1.f) { sxmin = -frame; sxmax = frame; symin = -1.f; symax = 1.f; }
        else { sxmin = -1.f; sxmax = 1.f; symin = -1.f / frame; symax = 1.f / frame; }
        lensRadius = c.lens_radius; focalDistance = c.focal_distance;
        orthographic = c.orthographic != 0;
        if (orthographic) {   // CreateOrthographicCamera, Orthographic.cpp:94-121: ScreenScale = 2
            float ScreenScale = 2.0f;
            sxmin *= ScreenScale; sxmax *= ScreenScale; symin *= ScreenScale; symax *= ScreenScale;
        }
        // Orthographic(0, 10) = Scale(1, 1, 1 / (zFar - zNear)) * Translate(0, 0, -zNear), Transform.cpp:282-285
        Xform cameraToScreen = orthographic ? XMul(Scale(1, 1, 1 / (10.f - 0.f)), Translate(V3(0, 0, -0.f))) : Perspective(c.fov_deg, 1e-2f, 1000.f);
        // Camera.h:64-70
        Xform screenToRaster = XMul(XMul(Scale(W, H, 1), Scale(1 / (sxmax - sxmin), 1 / (symin - symax), 1)), Translate(V3(-sxmin, -symax, 0)));
        Xform rasterToScreen = XInverse(screenToRaster);
        Xform r2c = XMul(XInverse(cameraToScreen), rasterToScreen);
        rasterToCamera = r2c.m;
        dxCamera = XPoint(rasterToCamera, V3(1, 0, 0)) - XPoint(rasterToCamera, V3(0, 0, 0));
        dyCamera = XPoint(rasterToCamera, V3(0, 1, 0)) - XPoint(rasterToCamera, V3(0, 0, 0));
        if (orthographic) {   // Orthographic.h:22-23: vector transforms
            dxCamera = XVector(rasterToCamera, V3(1, 0, 0));
            dyCamera = XVector(rasterToCamera, V3(0, 1, 0));
        }
    }
    // GenerateRayDifferential, Perspective.cpp:62-112 + Transform::operator()(RayDifferential), Transform.h:246-256
    Ray GenerateRay(const P2 &pFilm, const P2 &pLens) const {
        V3 pCamera = XPoint(rasterToCamera, V3(pFilm.x, pFilm.y, 0));
        V3 dir = Normalize(V3(pCamera.x, pCamera.y, pCamera.z));
        V3 o(0, 0, 0), d = dir;
        if (orthographic) { o = pCamera; d = V3(0, 0, 1); }   // Orthographic.cpp:45-47
        if (lensRadius > 0) {
            P2 dsk = ConcentricSampleDisk(pLens);
            P2 pl(lensRadius * dsk.x, lensRadius * dsk.y);
            Float ft = focalDistance / d.z;
            V3 pFocus = o + d * ft;
            o = V3(pl.x, pl.y, 0);
            d = Normalize(pFocus - o);
        }
        V3 oError;
        V3 ow = XPointErr(cameraToWorld, o, &oError);
        V3 dw = XVector(cameraToWorld, d);
        Float lengthSquared = dw.LengthSquared();
        Float tMax = Infinity;
        if (lengthSquared > 0) {
            Float dt = Dot(Abs(dw), oError) / lengthSquared;
            ow += dw * dt;
            tMax -= dt;
        }
        Ray ray(ow, dw, tMax, medium);
        // offset rays, Perspective.cpp:86-106 (camera space), then CameraToWorld: plain point / vector transforms
        V3 rxO, ryO, rxD, ryD;
        if (orthographic) {   // Orthographic.cpp:62-78 (`o`, `d`: the camera-space main ray after the lens update)
            if (lensRadius > 0) {
                P2 dsk = ConcentricSampleDisk(pLens);
                P2 pl(lensRadius * dsk.x, lensRadius * dsk.y);
                Float ft = focalDistance / d.z;
                V3 pFocus = pCamera + dxCamera + (ft * V3(0, 0, 1));
                rxO = V3(pl.x, pl.y, 0);
                rxD = Normalize(pFocus - rxO);
                pFocus = pCamera + dyCamera + (ft * V3(0, 0, 1));
                ryO = V3(pl.x, pl.y, 0);
                ryD = Normalize(pFocus - ryO);
            } else {
                rxO = o + dxCamera;
                ryO = o + dyCamera;
                rxD = ryD = d;
            }
        } else if (lensRadius > 0) {
            P2 dsk = ConcentricSampleDisk(pLens);
            P2 pl(lensRadius * dsk.x, lensRadius * dsk.y);
            V3 dx = Normalize(pCamera + dxCamera);
            Float ft = focalDistance / dx.z;
            V3 pFocus = V3(0, 0, 0) + (ft * dx);
            rxO = V3(pl.x, pl.y, 0);
            rxD = Normalize(pFocus - rxO);
            V3 dy = Normalize(pCamera + dyCamera);
            ft = focalDistance / dy.z;
            pFocus = V3(0, 0, 0) + (ft * dy);
            ryO = V3(pl.x, pl.y, 0);
            ryD = Normalize(pFocus - ryO);
        } else {
            rxO = ryO = o;
            rxD = Normalize(pCamera + dxCamera);
            ryD = Normalize(pCamera + dyCamera);
        }
        ray.rxOrigin = XPoint(cameraToWorld, rxO);
        ray.ryOrigin = XPoint(cameraToWorld, ryO);
        ray.rxDirection = XVector(cameraToWorld, rxD);
        ray.ryDirection = XVector(cameraToWorld, ryD);
        ray.hasDifferentials = true;
        return ray;
    }
};

}  // namespace gnxo
