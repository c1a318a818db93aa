// api.hip -- C ABI of libgnxr.so (include/gnxr.h): device scene upload, the wavefront render loop and
// the batched Aggregate-seam entry points.  One process drives one GPU (gnxr_init binds the device);
// multi-GPU runs are one process per GPU with the image rows sharded by gnxr_render_params.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "host_scene.h"
#include "kernels.hip.h"
#include "trace4_kernel.hip.h"
#ifndef GX_WITH_TRACE4D
#define GX_WITH_TRACE4D 0   // the two-rays-per-lane traversal kernel: a measured negative result (profiles/README.md, round 3), built only on request
#endif
#if GX_WITH_TRACE4D
#include "trace4d_kernel.hip.h"
#endif
#include "hlbvh_build.hip.h"
#include "kernel_instances.h"

using namespace gnxr;

// compiled in inst_whitted.hip / inst_whitted_tex.hip / inst_vol.hip
#define X(M, L, S, T) extern template GX_WHITTED_SIGNATURE(M, L, S, T)
GX_WHITTED_INSTANCES(X)
#undef X
#define X(M, L, ST, T) extern template GX_VOL_SIGNATURE(M, L, ST, T)
GX_VOL_INSTANCES(X)
#undef X

namespace { int hip_status(hipError_t e); }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);       \
            return hip_status(e_);                                                                      \
        }                                                                                               \
    } while (0)

namespace {

// hipError_t -> gnxr_status: allocation failures, "there is no (such) device", and everything else (a failed launch, an
// invalid argument, a fault reported at the next synchronisation) as GNXR_ERR_RUNTIME
int hip_status(hipError_t e) {
    if (e == hipErrorOutOfMemory) return GNXR_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver || e == hipErrorNotInitialized) return GNXR_ERR_NO_DEVICE;
    return GNXR_ERR_RUNTIME;
}

int g_device = -1;
std::vector<int> g_devices;        // gnxr_init_devices: every scene is replicated on these and renders shard their rows over them
std::vector<char> g_peer_ok;       // per entry of g_devices: the primary device and this one can address each other's memory (peer access enabled both ways)
int g_num_cus = 256;
int g_profiling = 0;
int g_grid_bpc = 8;   // blocks per CU that cap the grid of a grid-stride kernel (GNXR_GRID_BLOCKS_PER_CU: tuning knob)
int g_trace_blocks_per_cu = 5;   // persistent blocks of the traversal kernel per CU (5 waves per SIMD at its 96 VGPRs, 32 KB of LDS each); GNXR_TRACE_BLOCKS_PER_CU overrides (tuning)

// Per-kernel timing with HIP events on the render stream.  Events are recycled from a pool and resolved
// after the stream has been synchronised.
struct KernelTimer {
    std::vector<hipEvent_t> pool;
    struct Span { int kind; hipEvent_t a, b; };
    std::vector<Span> open;
    size_t used = 0;
    double seconds[3] = {0, 0, 0};
    unsigned launches[3] = {0, 0, 0};
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[used++];
    }
    void begin(int kind, hipStream_t st) { Span s{kind, get(), get()}; if (s.a && s.b) { (void)hipEventRecord(s.a, st); open.push_back(s); } }
    void end(hipStream_t st) { if (!open.empty()) (void)hipEventRecord(open.back().b, st); }
    // call only after the stream has been synchronised
    void collect() {
        for (auto &s : open) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { seconds[s.kind] += ms * 1e-3; launches[s.kind]++; } }
        open.clear();
        used = 0;
    }
    ~KernelTimer() { for (auto e : pool) (void)hipEventDestroy(e); }
};

int ensure_device() {
    // the current device is per host thread in HIP: a call from a thread other than the one that ran gnxr_init (the Qt
    // RenderThread of INTEGRATION.md) must not silently land on device 0
    if (g_device >= 0) { HIP_TRY(hipSetDevice(g_device)); return GNXR_OK; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); libgnxr has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return GNXR_ERR_NO_DEVICE;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_device = dev;
    if (const char *e = getenv("GNXR_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) g_trace_blocks_per_cu = v; }
    if (const char *e = getenv("GNXR_GRID_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 4096) g_grid_bpc = v; }
    return GNXR_OK;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    int alloc(size_t count) {
        if (count <= n && p) return GNXR_OK;
        release();
        if (count == 0) count = 1;
        HIP_TRY(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
        return GNXR_OK;
    }
    int upload(const T *src, size_t count) {
        int rc = alloc(count);
        if (rc) return rc;
        if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        return GNXR_OK;
    }
    template <typename V> int upload(const V &v) { return upload(v.data(), v.size()); }
};

int grid_for(long long n, int blocks_per_cu = 0) {
    if (blocks_per_cu <= 0) blocks_per_cu = g_grid_bpc;
    long long need = (n + kBlock - 1) / kBlock;
    long long cap = (long long)g_num_cus * blocks_per_cu;
    return (int)std::max<long long>(1, std::min(need, cap));
}

}  // namespace

struct gnxr_scene {
    CompiledScene cs;
    // device tables
    DevBuf<DNode> nodes;
    DevBuf<DNode4> nodes4;
    DevBuf<DTri> tris;
    DevBuf<float> leaf_boxes;
    DevBuf<uint8_t> tri_class;
    DevBuf<DSphere> spheres;
    DevBuf<DMaterial> materials, materials_single;
    DevBuf<DTexture> textures;
    DevBuf<float> tex_texels, ewa_lut, tri_uv, tri_n, tri_s;
    DevBuf<DLight> lights;
    DevBuf<int32_t> infinite;
    DevBuf<uint16_t> perms;
    DevBuf<int32_t> primes, prime_sums;
    DevBuf<uint32_t> prime_magic;
    DevBuf<float> env_texels4, env_cond_func, env_cond_cdf, env_cond_int, env_marg_func, env_marg_cdf;
    DevBuf<uint16_t> env_marg_guide, env_cond_guide;
    DevBuf<float> grid_table;
    DevBuf<DMedium> dmedia;
    DevBuf<float> grid_density;
    DevBuf<int32_t> tri_media;
    DLightGrid grid;
    int grid_strategy = -1;
    // per-render state (grown on demand)
    DevBuf<float4> rec[kRecGroups], mis_Y;   // the record groups of the path slots (PathArrays, kernels.hip.h)
    static void record_ptrs(DevBuf<float4> *b, float4 *g[kRecGroups]) { for (int i = 0; i < kRecGroups; ++i) g[i] = b[i].p; }
    DevBuf<float4> L, accum, out;
    DevBuf<int> hit, queue_a, queue_b, queue_nee, queue_c0, queue_c1, queue_c2, queue_c3;
    DevBuf<unsigned char> pflags, pclass;
    DevBuf<unsigned int> nee_vis;
    DevBuf<unsigned int> tile_counts;
    DevBuf<int> trace_spill;   // global part of k_trace's per-lane traversal stacks
    DevBuf<float4> vol_n1, vol_f, vol_Li, vol_Tr, vol_Ld, vol_mres;   // VolPath light-estimate records (vol_kernel.hip.h)
    DevBuf<int4> vol_vs;
    DevBuf<unsigned char> vol_state;
    // VolPath packing (k_vol_pack): the second set of the state arrays, the original slot of every path, the renumbering map and the results
    DevBuf<float4> vol_alt[kVolPackF4], vol_alt_rec[kRecGroups], vol_Lout;
    DevBuf<unsigned char> vol_alt_state;
    DevBuf<int> vol_orig, vol_alt_orig, vol_newslot;
    DevBuf<float4> wh_o, wh_d, wh_L, wh_w;   // Whitted recursion frames (whitted_kernel.hip.h)
    DevBuf<float4> wh_rxo, wh_rxd, wh_ryo, wh_ryd;   // their ray differentials (scenes with image textures)
    DevBuf<float> wh_pdf;
    DevBuf<int> wh_rec;
    DevBuf<Counters> counters;
    Counters *h_counters = nullptr;  // pinned
    // the device-driven PathIntegrator loop: lagging copies of the counters (pinned ring, one event per slot) -- the host reads them
    // without ever waiting for the iteration it has just enqueued
    // k_shade runs one kernel per material class; the classes are independent, so they go to different streams and fill each other's ends
    hipStream_t aux_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    static constexpr int kRing = 8;
    Counters *h_ring = nullptr;      // pinned, kRing entries
    hipEvent_t ring_ev[kRing] = {};
    int stack_size = 32;
    bool wide_ok = true;   // 4-wide traversal usable (leaf sizes / triangle count fit the reference encoding)
    std::recursive_mutex render_mutex;   // one render in flight per handle; gnxr_render holds it around its staging buffer too
    int device = 0;                      // the HIP device the tables live on
    std::vector<std::unique_ptr<gnxr_scene>> replicas;   // the same scene on the other devices of gnxr_init_devices (element 0 of that list is this one)
    DevBuf<float4> shard_out;            // a replica's full-size output plane; its rows are peer-copied into the primary's image
    void *h_stage = nullptr;             // pinned: a replica's rows on their way to the primary when the two devices have no peer access
    size_t h_stage_bytes = 0;

    int bind() const { HIP_TRY(hipSetDevice(device)); return GNXR_OK; }
    ~gnxr_scene() {
        if (h_counters) (void)hipHostFree(h_counters);
        if (h_ring) (void)hipHostFree(h_ring);
        if (h_stage) (void)hipHostFree(h_stage);
        for (hipEvent_t e : ring_ev) if (e) (void)hipEventDestroy(e);
        for (hipStream_t a : aux_stream) if (a) (void)hipStreamDestroy(a);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (hipEvent_t e : ev_join) if (e) (void)hipEventDestroy(e);
    }

    DScene device_scene(int W, int H) {
        DScene d;
        d.nodes = reinterpret_cast<const float4 *>(nodes.p);
        d.nodes4 = reinterpret_cast<const float4 *>(nodes4.p);
        d.root4 = cs.root4;
        d.tris = tris.p;
        d.leaf_box = reinterpret_cast<const float4 *>(leaf_boxes.p);
        d.tri_class = tri_class.p;
        static const bool no_verts = getenv("GNXR_LEAF_BOX_TABLE") != nullptr;   // experiment switch: always read leaf_boxes
        d.leaf1_from_verts = (cs.leaf1_from_verts && !no_verts) ? 1 : 0;
        d.spheres = spheres.p;
        d.n_spheres = cs.n_spheres;
        d.materials = materials.p + 1;   // [0] carries the texture tables
        d.escape_class = 0;              // render_one: 3 for the PathIntegrator in a scene without image-textured materials
        d.lt.lights = lights.p;
        d.lt.n_lights = (int)cs.desc_lights.size();
        d.lt.infinite = infinite.p;
        d.lt.n_infinite = (int)cs.infinite_lights.size();
        d.lt.grid = grid;
        d.lt.grid_table = grid_table.p;
        d.lt.has_env = cs.has_env ? 1 : 0;
        d.lt.env = cs.env;
        d.lt.env_texels = reinterpret_cast<const float4 *>(env_texels4.p);
        d.lt.env_cond_func = env_cond_func.p; d.lt.env_cond_cdf = env_cond_cdf.p; d.lt.env_cond_int = env_cond_int.p;
        d.lt.env_marg_func = env_marg_func.p; d.lt.env_marg_cdf = env_marg_cdf.p;
        d.lt.env_marg_guide = env_marg_guide.p; d.lt.env_cond_guide = env_cond_guide.p;
        d.st.perms = perms.p; d.st.primes = primes.p; d.st.prime_sums = prime_sums.p; d.st.prime_magic = prime_magic.p;
        d.st.h = make_halton(W, H);
        return d;
    }
    DMediaTables media_tables() {
        DMediaTables m;
        m.media = dmedia.p;
        m.density = grid_density.p;
        m.tri_media = cs.tri_media.empty() ? nullptr : reinterpret_cast<const int2 *>(tri_media.p);
        return m;
    }
    // light-selection table (core/LightDistribution.cpp).  The spatial strategy's dense voxel table is filled on the device
    // (k_light_grid, ~1 ms instead of ~1 s of host threads for 64^3 voxels); GNXR_HOST_LIGHT_GRID=1 forces the host
    // restatement, which produces the same bits (tests/test_gpu_parity.py::test_light_grid_device_equals_host).
    int ensure_grid(int strategy, bool force_host = false) {
        if (grid_strategy == strategy && !force_host) return GNXR_OK;
        const int nl = (int)cs.desc_lights.size();
        const bool on_device = strategy == GNXR_LIGHTS_SPATIAL && nl >= 2 && !force_host && getenv("GNXR_HOST_LIGHT_GRID") == nullptr;
        std::vector<float> table;
        build_light_grid(cs, strategy, &grid, &table, /*layout_only=*/true);
        {   // the dense spatial table holds nvox^3 x (2 lights + 1) floats: refuse what cannot fit instead of failing inside an allocation
            const unsigned long long bytes = (unsigned long long)grid.nvox[0] * grid.nvox[1] * grid.nvox[2] * (unsigned long long)grid.stride * sizeof(float);
            size_t free_b = 0, total_b = 0;
            unsigned long long limit = 64ull << 30;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = std::min<unsigned long long>(limit, free_b / 2);
            if (bytes > limit) {
                set_error("spatial light distribution: %d x %d x %d voxels x %d lights need %.1f GB (limit %.1f GB); use GNXR_LIGHTS_POWER or GNXR_LIGHTS_UNIFORM for this many lights",
                          grid.nvox[0], grid.nvox[1], grid.nvox[2], nl, bytes * 1e-9, limit * 1e-9);
                return GNXR_ERR_UNSUPPORTED;
            }
        }
        if (!on_device) build_light_grid(cs, strategy, &grid, &table, false);
        int rc;
        if (on_device) {
            const size_t nv = (size_t)grid.nvox[0] * grid.nvox[1] * grid.nvox[2];
            if ((rc = grid_table.alloc(nv * grid.stride)) != GNXR_OK) return rc;
            HIP_TRY(hipMemset(grid_table.p, 0, nv * grid.stride * sizeof(float)));   // padded records: the pad floats are zero, as in the host-built table
            float ri[5 * 128];
            light_grid_probes(cs, ri);
            DevBuf<float> d_ri;
            if ((rc = d_ri.upload(ri, 5 * 128)) != GNXR_OK) return rc;
            DLightTables lt = device_scene(1, 1).lt;
            bool area_only = true;
            for (const gnxr_light &l : cs.desc_lights) if (l.type != GNXR_LIGHT_AREA_TRI) area_only = false;
            const int blocks = (int)std::min<size_t>((nv + kBlock - 1) / kBlock, (size_t)g_num_cus * 8);
            if (nl > kGridMaxLights) {   // mesh lights: any number of lights, the table is the scratch space
                if (area_only) hipLaunchKernelGGL((k_light_grid_any<LT_AREA>), dim3(blocks), dim3(kBlock), 0, 0, lt, grid, (const float *)d_ri.p, grid_table.p);
                else hipLaunchKernelGGL((k_light_grid_any<LT_ALL>), dim3(blocks), dim3(kBlock), 0, 0, lt, grid, (const float *)d_ri.p, grid_table.p);
            } else if (area_only) hipLaunchKernelGGL((k_light_grid<LT_AREA>), dim3(blocks), dim3(kBlock), 0, 0, lt, grid, (const float *)d_ri.p, grid_table.p);
            else hipLaunchKernelGGL((k_light_grid<LT_ALL>), dim3(blocks), dim3(kBlock), 0, 0, lt, grid, (const float *)d_ri.p, grid_table.p);
            HIP_TRY(hipDeviceSynchronize());
        } else if ((rc = grid_table.upload(table)) != GNXR_OK) return rc;
        grid_strategy = force_host ? -1 : strategy;
        return GNXR_OK;
    }
};

// ---- HLBVH on the device (hlbvh_build.hip.h): Morton codes, own LSD radix sort, the treelets' LBVHs and the SAH over their roots.  The host
// gets the finished build tree back (scene_compile.cpp flattens it and derives the 4-wide layout as for the other split methods).
namespace {
bool device_hlbvh_build(const float *prim_bounds6, const float *centroids3, int n, const float lo[3], const float hi[3], std::vector<HlbvhNode> *nodes_out, int *root_out,
                        uint32_t *prims_sorted) {
    using namespace hlbvh;
    if (ensure_device() != GNXR_OK) return false;
    if (n <= 0) { set_error("HLBVH: no primitives"); return false; }
    DevBuf<float> d_cen, d_pb;
    DevBuf<uint32_t> k_a, k_b, v_a, v_b, hist, sums, head, ukey, ustart, thead, total;
    DevBuf<int> parent, roots, tmp, flags;
    DevBuf<unsigned int> arrived;
    DevBuf<HlbvhNode> d_nodes;
    const int n_tiles = (n + kTile - 1) / kTile;
    const auto fail = [&](const char *what) { set_error("HLBVH device build: %s", what); return false; };
    if (d_cen.upload(centroids3, 3 * (size_t)n) || d_pb.upload(prim_bounds6, 6 * (size_t)n) || k_a.alloc(n) || k_b.alloc(n) || v_a.alloc(n) || v_b.alloc(n) ||
        hist.alloc((size_t)64 * n_tiles) || sums.alloc((size_t)std::max(n_tiles, (64 * n_tiles + kTile - 1) / kTile) + 1) || head.alloc(n) || ukey.alloc(n) || ustart.alloc(n) ||
        thead.alloc(n) || total.alloc(2) || flags.alloc(2))
        return fail("out of device memory");
    if (hipMemset(flags.p, 0, 2 * sizeof(int)) != hipSuccess) return fail("memset");
    const bool verbose = getenv("GNXR_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char *name) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gnxr] hlbvh %-10s %7.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    stage("alloc");
    const int g = grid_for(n);
    hipLaunchKernelGGL(hlbvh::k_morton_codes, dim3(g), dim3(kB), 0, 0, (const float *)d_cen.p, n, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], k_a.p, v_a.p);
    // exclusive scan helper (in place); total (optional) lands in *d_total
    auto scan = [&](uint32_t *v, int m, uint32_t *d_total) {
        const int tiles = (m + kTile - 1) / kTile;
        hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kB), 0, 0, v, m, sums.p);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, 0, sums.p, tiles, d_total);
        hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(kB), 0, 0, v, m, (const uint32_t *)sums.p);
    };
    // RadixSort (BVHAccel.cpp:102-141): 30 bits, 6 per pass, least significant first, stable
    uint32_t *kin = k_a.p, *kout = k_b.p, *vin = v_a.p, *vout = v_b.p;
    for (int pass = 0; pass < 5; ++pass) {
        hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(kB), 0, 0, (const uint32_t *)kin, n, 6 * pass, n_tiles, hist.p);
        scan(hist.p, 64 * n_tiles, nullptr);
        hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(kB), 0, 0, (const uint32_t *)kin, (const uint32_t *)vin, n, 6 * pass, n_tiles, (const uint32_t *)hist.p, kout, vout);
        std::swap(kin, kout); std::swap(vin, vout);
    }
    const uint32_t *codes = kin, *prims = vin;   // sorted
    stage("sort");
    // leaves = runs of equal codes
    hipLaunchKernelGGL(k_hl_flags, dim3(g), dim3(kB), 0, 0, codes, n, head.p);
    scan(head.p, n, total.p);
    hipLaunchKernelGGL(k_hl_runs, dim3(g), dim3(kB), 0, 0, codes, (const uint32_t *)head.p, n, ukey.p, ustart.p);
    uint32_t U = 0;
    if (hipMemcpy(&U, total.p, sizeof(U), hipMemcpyDeviceToHost) != hipSuccess || U == 0 || U > (uint32_t)n) return fail("run count");
    // treelets = runs of equal top 12 bits
    const int gu = grid_for(U);
    hipLaunchKernelGGL(k_hl_tflags, dim3(gu), dim3(kB), 0, 0, (const uint32_t *)ukey.p, (int)U, thead.p);
    scan(thead.p, (int)U, total.p + 1);
    uint32_t T = 0;
    if (hipMemcpy(&T, total.p + 1, sizeof(T), hipMemcpyDeviceToHost) != hipSuccess || T == 0 || T > 4096u) return fail("treelet count");
    const size_t cap = (size_t)2 * U + T;
    if (d_nodes.alloc(cap) || parent.alloc(cap) || arrived.alloc(cap) || roots.alloc(T) || tmp.alloc(T)) return fail("out of device memory");
    if (hipMemset(d_nodes.p, 0, cap * sizeof(HlbvhNode)) != hipSuccess || hipMemset(parent.p, 0xff, cap * sizeof(int)) != hipSuccess ||
        hipMemset(arrived.p, 0, cap * sizeof(unsigned int)) != hipSuccess)
        return fail("memset");
    hipLaunchKernelGGL(k_hl_leaves, dim3(gu), dim3(kB), 0, 0, (const uint32_t *)ustart.p, (int)U, n, prims, (const float *)d_pb.p, d_nodes.p, flags.p);
    if (U > 1) {
        hipLaunchKernelGGL(k_hl_internal, dim3(gu), dim3(kB), 0, 0, (const uint32_t *)ukey.p, (int)U, d_nodes.p, parent.p);
        hipLaunchKernelGGL(k_hl_fit, dim3(gu), dim3(kB), 0, 0, (int)U, d_nodes.p, (const int *)parent.p, arrived.p);
    }
    hipLaunchKernelGGL(k_hl_roots, dim3(gu), dim3(kB), 0, 0, (const uint32_t *)ukey.p, (const uint32_t *)thead.p, (int)U, roots.p);
    stage("treelets");
    // buildUpperSAH level by level: the ranges of a level are split by one wave each
    int h_flags[2] = {0, -1};
    if (T == 1) {
        if (hipMemcpy(flags.p + 1, roots.p, sizeof(int), hipMemcpyDeviceToDevice) != hipSuccess) return fail("copy");
    } else {
        DevBuf<UpRange> q_a, q_b;
        DevBuf<int> counters;   // [0] ranges of the next level, [1] upper nodes allocated
        if (q_a.alloc(T) || q_b.alloc(T) || counters.alloc(2) || hipMemset(counters.p, 0, 2 * sizeof(int)) != hipSuccess) return fail("out of device memory");
        UpRange first{0, (int)T, -1};
        if (hipMemcpy(q_a.p, &first, sizeof(first), hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
        UpRange *qin = q_a.p, *qout = q_b.p;
        int n_in = 1;
        for (int level = 0; n_in > 0; ++level) {
            if (level > (int)T) return fail("upper SAH did not terminate");
            const int blocks = std::max(1, std::min((n_in * 64 + kB - 1) / kB, g_num_cus * 8));
            hipLaunchKernelGGL(k_hl_upper_level, dim3(blocks), dim3(kB), 0, 0, (const UpRange *)qin, n_in, qout, counters.p, roots.p, tmp.p, d_nodes.p, (int)(2 * U), counters.p + 1,
                               flags.p + 1, flags.p);
            if (hipMemcpy(&n_in, counters.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail("level count");
            if (hipMemset(counters.p, 0, sizeof(int)) != hipSuccess) return fail("memset");
            std::swap(qin, qout);
        }
    }
    stage("upper");
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(hipGetErrorString(hipGetLastError()));
    if (hipMemcpy(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost) != hipSuccess) return fail("download");
    if (h_flags[0]) {
        set_error("HLBVH: the reference's build does not terminate on this input (coincident treelet centroids) or a leaf exceeds 65535 primitives");
        return false;
    }
    nodes_out->resize(cap);
    if (hipMemcpy(nodes_out->data(), d_nodes.p, cap * sizeof(HlbvhNode), hipMemcpyDeviceToHost) != hipSuccess) return fail("download");
    if (hipMemcpy(prims_sorted, prims, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("download");
    stage("download");
    *root_out = h_flags[1];
    return *root_out >= 0 && (size_t)*root_out < cap;
}
}  // namespace

namespace {
// Issue-rate probe: 8 independent v_fma_f32 chains per lane with all three operands in vector registers, no memory traffic; with 8 waves
// on every SIMD the loop is bound by VALU issue alone.  Inline asm: the compiler would otherwise keep `a` and `b` in scalar registers (a
// VOP3 with two scalar operands issues at half the rate: 578 instead of ~900 G wave-instructions/s) or pack pairs of chains into
// v_pk_fma_f32 (half the instructions at half the rate).  tools/probes/valu_probe.hip has the same loop for every instruction kind the
// kernels use: add / sub / mul / fma / logic issue in 2 cycles per wave64, min / max / compare / shift / integer multiply / conversion /
// packed and fp64 arithmetic in 4, rcp / sqrt in 8 (profiles/r02_valu_probe.log).
__global__ void __launch_bounds__(256) k_valu_peak(float *out, int iters, float a_in, float b_in) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float a = a_in, b = b_in;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mov_b32 %0, %0" : "+v"(a));
    asm volatile("v_mov_b32 %0, %0" : "+v"(b));
#define GX_FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#else
#define GX_FMA1(x) x = __builtin_fmaf(x, a, b);
#endif
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { GX_FMA1(x0) GX_FMA1(x1) GX_FMA1(x2) GX_FMA1(x3) GX_FMA1(x4) GX_FMA1(x5) GX_FMA1(x6) GX_FMA1(x7) }
    }
#undef GX_FMA1
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}
// Gather-rate probe: every lane reads the 8 dwordx4 of its own pseudo-random 128-byte record (a BVH node visit without the arithmetic),
// the next record depends on what was read (a traversal's dependent chain).
__global__ void __launch_bounds__(256) k_gather_peak(const float4 *__restrict__ tab, unsigned nrec, int iters, float *out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        const float4 *p = tab + (size_t)((idx >> 8) % nrec) * 8;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        idx ^= __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
}  // namespace

extern "C" {

int gnxr_probe_gather_peak(double *giga_lane_loads_per_s) {
    if (!giga_lane_loads_per_s) { set_error("null argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    const unsigned nrec = 8u * 1024 * 1024 / 128;
    const int blocks = g_num_cus * 5, iters = 1000;
    DevBuf<float4> tab;
    DevBuf<float> out;
    if ((rc = tab.alloc((size_t)nrec * 8)) != GNXR_OK || (rc = out.alloc((size_t)blocks * 256)) != GNXR_OK) return rc;
    HIP_TRY(hipMemset(tab.p, 0, (size_t)nrec * 128));
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_gather_peak, dim3(blocks), dim3(256), 0, 0, (const float4 *)tab.p, nrec, iters, out.p);
        (void)hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) break;
        float ms = 0;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0) best = std::max(best, (double)blocks * 256 * (double)iters * 8 / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIP_TRY(hipGetLastError());
    *giga_lane_loads_per_s = best;
    return GNXR_OK;
}

int gnxr_probe_valu_peak(double *giga_wave_insts_per_s) {
    if (!giga_wave_insts_per_s) { set_error("null argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    const int blocks = g_num_cus * 8, iters = 4096;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    DevBuf<float> out;
    if ((rc = out.alloc((size_t)blocks * 256)) != GNXR_OK) return rc;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {   // the first launch warms the clocks up
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_valu_peak, dim3(blocks), dim3(256), 0, 0, out.p, iters, 1.0000001f, 1e-9f);
        (void)hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) break;
        float ms = 0;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0)
            best = std::max(best, (double)blocks * 4 /* waves */ * (double)iters * 64 /* FMAs per iteration */ / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIP_TRY(hipGetLastError());
    *giga_wave_insts_per_s = best;
    return GNXR_OK;
}

int gnxr_init(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("no HIP device available; libgnxr has no CPU fallback"); return GNXR_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= n) { set_error("device %d out of range (%d visible)", device_id, n); return GNXR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device_id));
    g_device = -1;
    g_devices.assign(1, device_id);
    g_peer_ok.assign(1, 1);
    return ensure_device();
}
int gnxr_init_devices(int32_t n_devices, const int32_t *device_ids) {
    if (n_devices <= 0 || n_devices > 64 || !device_ids) { set_error("bad device list"); return GNXR_ERR_INVALID; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("no HIP device available; libgnxr has no CPU fallback"); return GNXR_ERR_NO_DEVICE; }
    for (int i = 0; i < n_devices; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= n) { set_error("device %d out of range (%d visible)", device_ids[i], n); return GNXR_ERR_INVALID; }
    int rc = gnxr_init(device_ids[0]);
    if (rc) return rc;
    g_devices.assign(device_ids, device_ids + n_devices);
    // Peer access between the primary and the others (assembling the image).  Whether it was granted is RECORDED per device: a pair
    // without it assembles its rows through a pinned host buffer (render_sharded), it is never assumed.
    g_peer_ok.assign(n_devices, 1);
    for (int i = 1; i < n_devices; ++i) {
        if (device_ids[i] == device_ids[0]) continue;   // the same card listed twice: plain device-to-device copies
        int can01 = 0, can10 = 0;
        bool ok = hipDeviceCanAccessPeer(&can01, device_ids[0], device_ids[i]) == hipSuccess && can01 &&
                  hipDeviceCanAccessPeer(&can10, device_ids[i], device_ids[0]) == hipSuccess && can10;
        if (ok) {
            hipError_t e0 = hipSetDevice(device_ids[0]) == hipSuccess ? hipDeviceEnablePeerAccess(device_ids[i], 0) : hipErrorInvalidDevice;
            hipError_t e1 = hipSetDevice(device_ids[i]) == hipSuccess ? hipDeviceEnablePeerAccess(device_ids[0], 0) : hipErrorInvalidDevice;
            ok = (e0 == hipSuccess || e0 == hipErrorPeerAccessAlreadyEnabled) && (e1 == hipSuccess || e1 == hipErrorPeerAccessAlreadyEnabled);
        }
        g_peer_ok[i] = ok ? 1 : 0;
        if (getenv("GNXR_VERBOSE")) fprintf(stderr, "[gnxr] device %d <-> %d: %s\n", device_ids[0], device_ids[i], ok ? "peer access" : "no peer access: rows are staged through the host");
    }
    (void)hipGetLastError();
    if (getenv("GNXR_NO_PEER")) for (int i = 1; i < n_devices; ++i) g_peer_ok[i] = 0;   // test switch: take the host-staged route everywhere
    HIP_TRY(hipSetDevice(device_ids[0]));
    return GNXR_OK;
}
void gnxr_shutdown(void) { g_device = -1; g_devices.clear(); g_peer_ok.clear(); }
int gnxr_set_profiling(int flags) { g_profiling = flags; return GNXR_OK; }
#ifdef GX_SHADE_STATS
// development builds only (-DGX_SHADE_STATS): wave-time per section of k_shade
int gnxr_debug_shade_stats(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_shade_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return GNXR_ERR_INVALID;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shade_stats), z, sizeof(z)); }
    return GNXR_OK;
}
#endif
#ifdef GX_TRACE_STATS
// development builds only (-DGX_TRACE_STATS): wave-level occupancy statistics of k_trace
int gnxr_debug_trace_stats(unsigned long long *out16, int reset) {   // 24 slots
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_trace_stats), 24 * sizeof(unsigned long long)) != hipSuccess) return GNXR_ERR_INVALID;
    if (reset) { unsigned long long z[24] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_stats), z, sizeof(z)); }
    return GNXR_OK;
}
#endif

// device side of gnxr_scene_create: everything compile_scene produced goes to the scene's device
static int upload_scene(gnxr_scene *s) {
    CompiledScene &cs = s->cs;
    int rc = s->bind();
    if (rc) return rc;
    if (cs.bvh_max_depth + 1 > 64) { set_error("BVH depth %d exceeds the 64-entry traversal stack (BVHAccel.cpp:661)", cs.bvh_max_depth); return GNXR_ERR_UNSUPPORTED; }
    s->stack_size = cs.bvh_max_depth + 1 <= 32 ? 32 : 64;
    s->wide_ok = cs.tris.size() < (1u << 24) && cs.stack4_need + 1 <= 128 && getenv("GNXR_BINARY_BVH") == nullptr;
    for (const DNode &n : cs.nodes) if ((n.meta & 0xffffu) > 127) s->wide_ok = false;
#define UP(field) if ((rc = s->field.upload(cs.field)) != GNXR_OK) return rc;
    UP(nodes) UP(nodes4) UP(tris) UP(leaf_boxes) UP(tri_class) UP(lights) UP(perms) UP(primes) UP(prime_sums) UP(prime_magic)
    UP(dmedia) UP(grid_density) UP(tri_media) UP(spheres) UP(textures) UP(tex_texels) UP(ewa_lut) UP(tri_uv) UP(tri_n) UP(tri_s)
    UP(env_texels4) UP(env_cond_func) UP(env_cond_cdf) UP(env_cond_int) UP(env_marg_func) UP(env_marg_cdf) UP(env_marg_guide) UP(env_cond_guide)
#undef UP
    {   // materials, preceded by one record that carries the texture tables (tex_tables(), device_texture.h)
        DTexTables tt;
        tt.textures = s->textures.p; tt.texels = reinterpret_cast<const float4 *>(s->tex_texels.p); tt.ewa_lut = s->ewa_lut.p; tt.tri_uv = cs.tri_uv.empty() ? nullptr : s->tri_uv.p; tt.tri_n = cs.tri_n.empty() ? nullptr : s->tri_n.p; tt.tri_s = cs.tri_s.empty() ? nullptr : s->tri_s.p;   // (an empty upload still allocates)
        for (int k = 0; k < 2; ++k) {
            const std::vector<DMaterial> &src = k == 0 ? cs.materials : cs.materials_single;
            std::vector<DMaterial> up(src.size() + 1);
            memset(&up[0], 0, sizeof(DMaterial));
            memcpy(&up[0], &tt, sizeof(tt));
            std::copy(src.begin(), src.end(), up.begin() + 1);
            if ((rc = (k == 0 ? s->materials : s->materials_single).upload(up)) != GNXR_OK) return rc;
        }
    }
    if ((rc = s->infinite.upload(cs.infinite_lights)) != GNXR_OK) return rc;
    if ((rc = s->counters.alloc(1)) != GNXR_OK) return rc;
    if (hipHostMalloc((void **)&s->h_counters, sizeof(Counters)) != hipSuccess) { set_error("hipHostMalloc failed"); return GNXR_ERR_OOM; }
    if (hipHostMalloc((void **)&s->h_ring, sizeof(Counters) * gnxr_scene::kRing) != hipSuccess) { set_error("hipHostMalloc failed"); return GNXR_ERR_OOM; }
    for (hipEvent_t &e : s->ring_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipStream_t &a : s->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    for (hipEvent_t &e : s->ev_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return GNXR_OK;
}

int gnxr_scene_create(const gnxr_scene_desc *desc, gnxr_scene **out) {
    if (!desc || !out) { set_error("null argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    gnxr_scene *s = new (std::nothrow) gnxr_scene();
    if (!s) return GNXR_ERR_OOM;
    s->device = g_device;
    if (!compile_scene(desc, &s->cs, device_hlbvh_build)) { delete s; return GNXR_ERR_INVALID; }
    if ((rc = upload_scene(s)) != GNXR_OK) { delete s; return rc; }
    // gnxr_init_devices: the same tables on every other device of the list (the host-side compilation is shared)
    for (size_t i = 1; i < g_devices.size(); ++i) {
        std::unique_ptr<gnxr_scene> r(new (std::nothrow) gnxr_scene());
        if (!r) { delete s; return GNXR_ERR_OOM; }
        r->device = g_devices[i];
        r->cs = s->cs;
        if ((rc = upload_scene(r.get())) != GNXR_OK) { delete s; (void)hipSetDevice(g_device); return rc; }
        s->replicas.push_back(std::move(r));
    }
    if ((rc = s->bind()) != GNXR_OK) { delete s; return rc; }
    const CompiledScene &cs = s->cs;
    if (getenv("GNXR_VERBOSE"))
        fprintf(stderr, "[gnxr] scene: %zu tris, %zu nodes (depth %d), %zu 4-wide nodes (stack %d), wide=%d, %zu device(s)\n", cs.tris.size(), cs.nodes.size(), cs.bvh_max_depth,
                cs.nodes4.size(), cs.stack4_need, (int)s->wide_ok, 1 + s->replicas.size());
    *out = s;
    return GNXR_OK;
}
void gnxr_scene_destroy(gnxr_scene *s) { delete s; }

int gnxr_scene_info(const gnxr_scene *s, int32_t *n_nodes, int32_t *max_depth, int32_t *n_vox) {
    if (!s) return GNXR_ERR_INVALID;
    if (n_nodes) *n_nodes = (int32_t)s->cs.nodes.size();
    if (max_depth) *max_depth = s->cs.bvh_max_depth;
    if (n_vox) *n_vox = s->grid_strategy >= 0 ? s->grid.nvox[0] * s->grid.nvox[1] * s->grid.nvox[2] : 0;
    return GNXR_OK;
}

static int count_local_rows(const gnxr_render_params *p) {
    int rows = 0;
    for (int y = 0; y < p->height; ++y)
        if ((y / p->shard_rows) % p->shard_count == p->shard_index) ++rows;
    return rows;
}

static thread_local bool g_reserve_only = false;   // gnxr_render_reserve: render_one stops after its allocations

// One device: the wavefront loop over the rows `pin` assigns to this shard, on the device the scene's tables live on.
static int render_one(gnxr_scene *s, const gnxr_render_params *pin, void *d_rgba_out, void *hip_stream, gnxr_stats *stats) {
    if (!s || !pin || !d_rgba_out) { set_error("null argument"); return GNXR_ERR_INVALID; }
    gnxr_render_params p = *pin;
    if (p.shard_count <= 0) p.shard_count = 1;
    if (p.shard_rows <= 0) p.shard_rows = 1;
    if (p.spp_end <= 0) p.spp_end = p.spp;
    if (p.width <= 0 || p.height <= 0 || p.spp <= 0 || p.spp_begin < 0 || p.spp_end > p.spp || p.spp_begin >= p.spp_end || p.shard_index < 0 ||
        p.shard_index >= p.shard_count || p.max_depth < 0 || p.max_depth > 250) {
        set_error("invalid render parameters");
        return GNXR_ERR_INVALID;
    }
    if (p.integrator != GNXR_INTEGRATOR_PATH && p.integrator != GNXR_INTEGRATOR_VOLPATH && p.integrator != GNXR_INTEGRATOR_WHITTED &&
        p.integrator != GNXR_INTEGRATOR_DIRECT) {
        set_error("unknown integrator %d", p.integrator);
        return GNXR_ERR_UNSUPPORTED;
    }
    const bool direct = p.integrator == GNXR_INTEGRATOR_DIRECT;
    if (direct && p.direct_strategy != GNXR_DIRECT_SAMPLE_ALL && p.direct_strategy != GNXR_DIRECT_SAMPLE_ONE) {
        set_error("unknown direct-lighting strategy %d", p.direct_strategy);
        return GNXR_ERR_INVALID;
    }
    // Whitted and DirectLighting share the depth-first state machine of whitted_kernel.hip.h
    const bool volpath = p.integrator == GNXR_INTEGRATOR_VOLPATH, whitted = p.integrator == GNXR_INTEGRATOR_WHITTED || direct;
    const int wmode = !direct ? WM_WHITTED : (p.direct_strategy == GNXR_DIRECT_SAMPLE_ONE ? WM_DIRECT_ONE : WM_DIRECT_ALL);
    const int nL = (int)s->cs.desc_lights.size();
    // NEE records per vertex, and the largest Light::nSamples (the array samples multiply the Halton index by it)
    int n_records = nL, max_light_samples = 1;
    if (wmode == WM_DIRECT_ONE) n_records = 1;
    if (wmode == WM_DIRECT_ALL) {
        n_records = 0;
        for (const gnxr_light &l : s->cs.desc_lights) { n_records += std::max(1, l.n_samples); max_light_samples = std::max(max_light_samples, l.n_samples); }
        n_records = std::max(1, n_records);
    }
    // (media in the scene are fine: these integrators never look at them -- a medium boundary without material is passed
    // through by the main ray, WhittedIntegrator.cpp:34-35, and blocks shadow rays like any other surface, Light.cpp:28-31)
    if (whitted && (n_records > 256 || p.max_depth > 32)) {
        set_error("Whitted / DirectLighting on the device: at most 256 light samples per vertex (every light is sampled at every vertex), depth 32");
        return GNXR_ERR_UNSUPPORTED;
    }
    bool textured_scene = false;
    for (const DMaterial &m : s->cs.materials) if (m.shade_class == 3) textured_scene = true;
    std::lock_guard<std::recursive_mutex> lock(s->render_mutex);
    if (int brc = s->bind()) return brc;
    auto t_start = std::chrono::steady_clock::now();
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc = s->ensure_grid(p.light_strategy);
    if (rc) return rc;
    DScene sc = s->device_scene(p.width, p.height);
    // the device sampler keeps the Halton index in 32 bits
    if ((unsigned long long)sc.st.h.stride * ((unsigned long long)p.spp * max_light_samples + 1) >= (1ull << 32)) { set_error("spp too large for 32-bit Halton indices"); return GNXR_ERR_UNSUPPORTED; }
    // every index this render draws is below stride * (spp * n + 1); reversedDigits of base b stays below b * index (device_sampler.h)
    sc.st.h.base32_max = (int32_t)std::min<unsigned long long>(0x7fffffffull, 0xffffffffull / ((unsigned long long)sc.st.h.stride * ((unsigned long long)p.spp * max_light_samples + 1)));
    DRender r;
    memset(&r, 0, sizeof(r));
    r.cam = make_camera(s->cs.camera, p.width, p.height, s->cs.camera_medium);
    r.W = p.width; r.H = p.height; r.spp = p.spp; r.max_depth = p.max_depth; r.rr_threshold = p.rr_threshold;
    r.shard_index = p.shard_index; r.shard_count = p.shard_count; r.shard_rows = p.shard_rows;
    int local_rows = count_local_rows(&p);
    r.npix = local_rows * p.width;
    if (r.npix == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return GNXR_OK; }
    int nsamples = p.spp_end - p.spp_begin;
    int k = p.samples_per_pass;
    const bool path_int = !whitted && !volpath;
    if (k <= 0) {
        // auto.  PathIntegrator: sub-passes of ~64 M paths, four of them in flight (below) -- launches stay thick because they mix the
        // bounces of different sub-passes.  Measured at 1080p on cfg 3 (1024 spp per call, profiles/README.md round 3): 4 x 32 spp
        // (59 GB of state) renders as fast as round 2's two 128-spp passes (118 GB); 4 x 16 spp (29.5 GB) costs 2 % -- every launch of
        // the persistent traversal kernel pays a ramp and a drain of ~0.3 ms, and the number of launches grows as the resident state shrinks.  VolPath / Whitted / DirectLighting render one pass at a
        // time: big passes keep their thin late rounds from under-filling the GPU, so take up to a quarter of the free HBM for path state
        // (~230 B per path), at most 256 M paths; Whitted / DirectLighting keep max_depth frames and n_records NEE records per path
        long long target = 32ll << 20;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) target = std::max<long long>(target, std::min<long long>(256ll << 20, (long long)(free_b / 4 / 230)));
        if (volpath) target = std::min<long long>(target, 64ll << 20);   // + 8 float4 of VolPath state per path
        if (whitted) target = (4ll << 20) / std::max(1, n_records / 4);
        if (path_int) target = 64ll << 20;
        k = (int)std::max<long long>(1, std::min<long long>(nsamples, target / r.npix));
    }
    k = std::min(k, nsamples);
    // PathIntegrator: up to kMaxRegions sub-passes in flight at once, each in its own region of the state arrays (the device-driven loop
    // below).  passes_in_flight = 0 picks 4, fewer when the call has fewer sub-passes or the state would not fit the 32-bit work indices
    // or ~45 % of the free HBM (~230 B per path slot beyond what this handle already holds).
    static const int regions_env = getenv("GNXR_REGIONS") ? atoi(getenv("GNXR_REGIONS")) : 0;   // tuning knob
    static const bool pipeline = getenv("GNXR_PIPELINE") ? atoi(getenv("GNXR_PIPELINE")) != 0 : true;   // experiment switch: 0 = one pass at a time
    const int kh = k;
    const size_t half = (size_t)r.npix * kh;   // slots of one region
    int in_flight = 1;
    if (path_int && pipeline) {
        const int n_subs = (nsamples + kh - 1) / kh;
        in_flight = p.passes_in_flight > 0 ? p.passes_in_flight : (regions_env > 0 ? regions_env : 4);
        in_flight = std::max(1, std::min(std::min(in_flight, kMaxRegions), n_subs));
        size_t free_b = 0, total_b = 0;
        const bool have_mem = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        for (; in_flight > 1; --in_flight) {
            const unsigned long long want = (unsigned long long)in_flight * half, held = (unsigned long long)s->L.n;
            const bool idx_ok = want < (1ull << 31) && want * 3ull < (1ull << 32);
            bool mem_ok = true;
            if (want > held && have_mem) mem_ok = (want - held) * 238ull < (unsigned long long)(0.45 * (double)free_b);
            // (and never beyond ~150 GB of path state: a 177 GB configuration -- 6 x 64 spp at 1080p -- rendered three times SLOWER than the
            // 118 GB one on the 288 GB card, profiles/r03_shard_efficiency.log)
            if (want * 238ull > 150ull * 1000 * 1000 * 1000) mem_ok = false;
            if (idx_ok && mem_ok) break;
        }
    }
    size_t cap = (size_t)in_flight * half;
    // k_trace's work cursor is 32-bit unsigned: continuation rays + two NEE items per record; record slots are `record * cap + path`
    {
        const unsigned long long recs = whitted ? (unsigned long long)std::max(1, n_records) : 1ull;
        if (cap >= (1ull << 31) || (unsigned long long)cap * recs >= (1ull << 31) || (unsigned long long)cap * (1ull + 2ull * recs) >= (1ull << 32)) {
            set_error("pass too large: %zu paths x %llu NEE records per vertex overflow the 32-bit work indices; lower samples_per_pass", cap, recs);
            return GNXR_ERR_INVALID;
        }
    }
#define AL(f) if ((rc = s->f.alloc(cap)) != GNXR_OK) return rc;
    const size_t nrec = whitted ? (size_t)std::max(1, n_records) : 1;   // Whitted / DirectLighting keep one NEE record per light sample of a vertex
    for (int i = 0; i < kRecGroups; ++i) {
        const size_t per_slot = i < 2 ? 1 : (i < 4 ? nrec : (direct ? nrec : 1));
        if ((rc = s->rec[i].alloc(cap * per_slot * kRS)) != GNXR_OK) return rc;
    }
    if ((rc = s->mis_Y.alloc(cap * (direct ? nrec : 1))) != GNXR_OK) return rc;
    AL(L) AL(hit) AL(queue_a) AL(queue_b) AL(queue_nee) AL(queue_c0) AL(queue_c1) AL(queue_c2) AL(queue_c3) AL(pflags) AL(pclass) AL(nee_vis)
#undef AL
    if (whitted) {
        const size_t nl = (size_t)std::max(1, n_records), md = (size_t)std::max(1, p.max_depth);
        if ((rc = s->wh_rec.alloc(cap * nl)) ||
            (rc = s->wh_o.alloc(cap * md)) || (rc = s->wh_d.alloc(cap * md)) || (rc = s->wh_L.alloc(cap * md)) || (rc = s->wh_w.alloc(cap * md)) ||
            (rc = s->wh_pdf.alloc(cap * md)) || (rc = s->vol_vs.alloc(cap)))
            return rc;
        if (textured_scene && ((rc = s->wh_rxo.alloc(cap * (md + 1))) || (rc = s->wh_rxd.alloc(cap * (md + 1))) || (rc = s->wh_ryo.alloc(cap * (md + 1))) || (rc = s->wh_ryd.alloc(cap * (md + 1)))))
            return rc;
    }
    if (volpath) {
#define AL(f) if ((rc = s->f.alloc(cap)) != GNXR_OK) return rc;
        AL(vol_n1) AL(vol_f) AL(vol_Li) AL(vol_Tr) AL(vol_Ld) AL(vol_mres) AL(vol_vs) AL(vol_state)
        AL(vol_Lout) AL(vol_alt_state) AL(vol_orig) AL(vol_alt_orig) AL(vol_newslot)
        for (int i = 0; i < kVolPackF4; ++i) AL(vol_alt[i])
        for (int i = 0; i < kRecGroups; ++i) if ((rc = s->vol_alt_rec[i].alloc(cap * kRS)) != GNXR_OK) return rc;
#undef AL
    }
    {   // global part of k_trace's traversal stacks (the deepest walk either BVH layout can need), sized for a full grid
        const int entries = std::max(s->cs.stack4_need + 1, s->cs.bvh_max_depth + 2);
        if ((rc = s->trace_spill.alloc((size_t)g_num_cus * g_trace_blocks_per_cu * kBlock * (size_t)entries * 2)) != GNXR_OK) return rc;   // (x 2: k_trace4d keeps two columns per lane)
    }
    if ((rc = s->accum.alloc(r.npix)) != GNXR_OK) return rc;
    const int max_tiles = (int)((cap + kCompactTile - 1) / kCompactTile);
    if ((rc = s->tile_counts.alloc((size_t)5 * max_tiles)) != GNXR_OK) return rc;
    PathArrays pa;
    float4 *rec_primary[kRecGroups];
    gnxr_scene::record_ptrs(s->rec, rec_primary);
    pa.bind_records(rec_primary);
    pa.mis_Y = s->mis_Y.p;
    pa.L = s->L.p; pa.hit = s->hit.p; pa.pflags = s->pflags.p; pa.pclass = s->pclass.p; pa.nee_vis = s->nee_vis.p;
    VolArrays va;
    va.bind_records(rec_primary);
    va.mis_Y = s->mis_Y.p;
    va.vs = s->vol_vs.p; va.n1 = s->vol_n1.p; va.f = s->vol_f.p;
    va.Li = s->vol_Li.p; va.Tr = s->vol_Tr.p; va.Ld = s->vol_Ld.p; va.mres = s->vol_mres.p; va.state = s->vol_state.p;
    va.orig = s->vol_orig.p; va.Lout = s->vol_Lout.p;
    DMediaTables mt = s->media_tables();
    WhittedArrays wa;
    wa.ws = s->vol_vs.p; wa.fr_o = s->wh_o.p; wa.fr_d = s->wh_d.p; wa.fr_L = s->wh_L.p; wa.fr_w = s->wh_w.p; wa.fr_pdf = s->wh_pdf.p;
    wa.fr_rxo = s->wh_rxo.p; wa.fr_rxd = s->wh_rxd.p; wa.fr_ryo = s->wh_ryo.p; wa.fr_ryd = s->wh_ryd.p;
    wa.cap = (int)cap; wa.n_lights = nL; wa.n_records = n_records;
    // DirectLightingIntegrator::Preprocess requests maxDepth x lights x 2 2D arrays (DirectLightingIntegrator.cpp:19-25)
    wa.start_dim = wmode == WM_DIRECT_ALL ? 5 + 2 * (p.max_depth * nL * 2) : 5;
    if (whitted) {
        sc.materials = s->materials_single.p + 1;
    }

    if (g_reserve_only) { if (stats) memset(stats, 0, sizeof(*stats)); return GNXR_OK; }
    HIP_TRY(hipMemsetAsync(s->accum.p, 0, sizeof(float4) * r.npix, stream));
    HIP_TRY(hipMemsetAsync(s->counters.p, 0, sizeof(Counters), stream));
    struct EventPair {   // destroyed on every exit path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    hipEvent_t ev0 = ev.a, ev1 = ev.b;
    HIP_TRY(hipEventRecord(ev0, stream));
    unsigned long long rays_closest = 0, rays_any = 0, rays_mis = 0;
    unsigned int launches = 0, passes = 0;
    // counting: bit 1 = on the reference's binary tree, bit 2 = on the timed (4-wide) walk + the medium kernel's tracking steps
    const bool timing = (g_profiling & 1) != 0, count_wide = (g_profiling & 4) != 0, counting = (g_profiling & 2) != 0 && !count_wide, spheres = s->cs.n_spheres > 0;
    unsigned long long media_segments = 0;
    int class_mask = 0;
    for (const DMaterial &m : s->cs.materials) class_mask |= 1 << m.shade_class;
    const bool textured = (class_mask & 8) != 0;
    // escaped continuation rays of a scene with infinite lights get shade queue 3 to themselves when no image-textured material claims it
    const bool escape_queue = !whitted && !volpath && !textured && !s->cs.infinite_lights.empty() && getenv("GNXR_NO_ESCAPE_QUEUE") == nullptr;
    sc.escape_class = escape_queue ? 3 : 0;
    bool area_only = true, area_env_only = true;
    for (const gnxr_light &l : s->cs.desc_lights) {
        if (l.type != GNXR_LIGHT_AREA_TRI) area_only = false;
        if (l.type != GNXR_LIGHT_AREA_TRI && l.type != GNXR_LIGHT_INFINITE) area_env_only = false;
    }
    KernelTimer timer;
    Counters *dctr = s->counters.p;
    // (device-driven loop: w.n_closest / w.n_nee are upper bounds that size the launch, the kernel reads the counts through
    // w.n_closest_dev / w.n_nee_dev, and the rays are counted on the device: count_rays = false)
    auto launch_trace = [&](TraceWork w, int n_sh, int n_mis, bool count_rays = true) {
        long long total = (long long)w.n_closest + 2ll * w.n_nee;
        if (total <= 0) return;
        w.order = nullptr;
        // (binning the rays of a launch by kind / octant / origin cell with a radix sort was measured in round 2 and lost -- 57.6 - 62.0 ms against
        // 51.7 ms per pass without counting the sort: slot order is already coherent in origin and sorting breaks the coalescing of the state
        // reads; the switch and its library sort are gone, `order` stays in TraceWork for callers that bring their own permutation)
        (void)hipMemsetAsync(&dctr->cursor, 0, sizeof(unsigned int), stream);
        // LDS traversal stack: one column per lane, depth from the BVH (binary walk: depth + 1; 4-wide walk: stack4_need)
        const bool wide = s->wide_ok && !counting;
        const int entries = wide ? s->cs.stack4_need + 1 : s->cs.bvh_max_depth + 2;
        // 5 blocks of 4 waves per CU is what k_trace4's 96 VGPRs allow (5 waves per SIMD); the LDS of a block -- stack levels plus, for
        // the 4-wide kernel, the set-up ray records and the node cache -- must fit 5 times into the 160 KB; deeper levels spill to
        // global memory (LDS levels are worth more than a bigger node cache: profiles/README.md, r02 A/B table)
        const int per_cu = g_trace_blocks_per_cu;
        // besides the stack: the set-up ray records, the top-of-tree node cache and the order table
        const size_t fixed_b = wide ? (size_t)(kRayRecDwords + (spheres ? 1 : 0)) * kRqStride * sizeof(int) + (size_t)kTopCache * 128 + 128 : 0;
        static const int lds_levels_cap = getenv("GNXR_TRACE_LDS_LEVELS") ? std::max(2, atoi(getenv("GNXR_TRACE_LDS_LEVELS"))) : 64;   // tuning knob
        const int lds_entries = std::min(std::min(entries, lds_levels_cap), std::max(2, (int)(((160 * 1024) / per_cu - 1024 - fixed_b) / (kBlock * sizeof(int)))));
        const bool spill_needed = entries > lds_entries;
        const size_t lds = (size_t)lds_entries * kBlock * sizeof(int) + fixed_b;
        const int n_top = (int)std::min<size_t>(kTopCache, s->cs.root4 >= 0 ? s->cs.nodes4.size() : 0);
        // persistent waves: enough blocks to fill the chip, never more than the work needs
        int blocks = (int)std::min<long long>((long long)g_num_cus * per_cu, (total + kBlock - 1) / kBlock);
        if (timing) timer.begin(0, stream);
        // rays per atomic: at most kTraceChunk; the kernel shrinks the chunk for thin launches so that every wave gets one (trace_chunk())
        static const int chunk_max = getenv("GNXR_TRACE_CHUNK") ? std::max(64, atoi(getenv("GNXR_TRACE_CHUNK")) / 64 * 64) : kTraceChunk;   // tuning knob
        const int chunk = chunk_max;
#define GX_TRACE(C, W, S) hipLaunchKernelGGL((k_trace<C, W, S>), dim3(blocks), dim3(kBlock), lds, stream, sc, pa, w, &dctr->cursor, dctr, lds_entries, s->trace_spill.p, chunk)
#define GX_TRACE4(C, S, P) hipLaunchKernelGGL((k_trace4<C, S, P>), dim3(blocks), dim3(kBlock), lds, stream, sc, pa, w, &dctr->cursor, dctr, lds_entries, s->trace_spill.p, chunk, n_top)
#define GX_TRACE4_CS(C, S) do { if (spill_needed) GX_TRACE4(C, S, true); else GX_TRACE4(C, S, false); } while (0)
#if GX_WITH_TRACE4D
        static const bool dual = getenv("GNXR_TRACE_DUAL") ? atoi(getenv("GNXR_TRACE_DUAL")) != 0 : false;   // two rays per lane (trace4d_kernel.hip.h)
        if (wide && dual && !count_wide) {
            const int dper_cu = GX_T4D_WAVES;   // blocks of 4 waves per CU = waves per SIMD
            const size_t dfixed = (size_t)(kRayRecDwords + (spheres ? 1 : 0)) * kRqStride * sizeof(int) + (size_t)kTopCacheD * 128 + 128;
            const int dlds_entries = std::min(std::min(entries, lds_levels_cap), std::max(2, (int)(((160 * 1024) / dper_cu - 1024 - dfixed) / (2 * kBlock * sizeof(int)))));
            const int dspill_levels = std::max(0, entries - dlds_entries);
            const size_t dlds = (size_t)2 * dlds_entries * kBlock * sizeof(int) + dfixed;
            const int dn_top = (int)std::min<size_t>(kTopCacheD, s->cs.root4 >= 0 ? s->cs.nodes4.size() : 0);
            const int dblocks = (int)std::min<long long>((long long)g_num_cus * dper_cu, (total + 2 * kBlock - 1) / (2 * kBlock));
#define GX_TRACE4D(S, P) hipLaunchKernelGGL((k_trace4d<S, P>), dim3(std::max(1, dblocks)), dim3(kBlock), dlds, stream, sc, pa, w, &dctr->cursor, dctr, dlds_entries, dspill_levels, s->trace_spill.p, chunk, dn_top)
            if (spheres) { if (dspill_levels > 0) GX_TRACE4D(true, true); else GX_TRACE4D(true, false); }
            else { if (dspill_levels > 0) GX_TRACE4D(false, true); else GX_TRACE4D(false, false); }
#undef GX_TRACE4D
        } else
#endif
        if (wide) {   // the 4-wide walk (trace4_kernel.hip.h); count_wide: its counting variant
            if (spheres) { if (count_wide) GX_TRACE4_CS(true, true); else GX_TRACE4_CS(false, true); }
            else { if (count_wide) GX_TRACE4_CS(true, false); else GX_TRACE4_CS(false, false); }
        } else {      // the reference's binary tree: counting runs on BVHAccel's own walk, and scenes the 4-wide encoding cannot hold
            const bool cnt = counting || count_wide;
            if (spheres) { if (cnt) GX_TRACE(true, false, true); else GX_TRACE(false, false, true); }
            else { if (cnt) GX_TRACE(true, false, false); else GX_TRACE(false, false, false); }
        }
#undef GX_TRACE4_CS
#undef GX_TRACE4
#undef GX_TRACE
        if (timing) timer.end(stream);
        if (count_rays) {
            rays_closest += (unsigned long long)w.n_closest + (unsigned long long)n_mis;
            rays_any += (unsigned long long)n_sh;
            rays_mis += (unsigned long long)n_mis;
        }
        ++launches;
    };
    // stream compaction (compact_kernel.hip.h): count -> scan -> scatter, no global atomics
    // (n_dev: the item count lives on the device; `nin` then bounds it and sizes the launch)
    auto compact = [&](int mode, const int *qin, int nin, const unsigned char *keys, int nout, int nscatter, unsigned int *totals, int *o0, int *o1, int *o2, int *o3 = nullptr, int split = 0,
                       const unsigned *n_dev = nullptr) {
        int tiles = (nin + kCompactTile - 1) / kCompactTile;
        int g = std::max(1, std::min(tiles, g_num_cus * g_grid_bpc));
        const int *nohit = nullptr; const unsigned char *nocls = nullptr; unsigned char *nokeys = nullptr;
        // FLAGS with a fifth count: the paths that continue AND live in the lower half of the state arrays (slot < split)
        if (mode == COMPACT_FLAGS && nout == 5) hipLaunchKernelGGL((k_compact_count<COMPACT_FLAGS, 5>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, nohit, nocls, nokeys, split, n_dev);
        else if (mode == COMPACT_FLAGS) hipLaunchKernelGGL((k_compact_count<COMPACT_FLAGS, 4>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, nohit, nocls, nokeys, 0, n_dev);
        else if (mode == COMPACT_HITCLASS) {   // class of the triangle a path hit, looked up and left in `keys` for the scatter pass
            if (nout == 4) hipLaunchKernelGGL((k_compact_count<COMPACT_HITCLASS, 4>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, (const int *)s->hit.p, (const unsigned char *)s->tri_class.p, s->pclass.p, 0, n_dev);
            else hipLaunchKernelGGL((k_compact_count<COMPACT_HITCLASS, 3>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, (const int *)s->hit.p, (const unsigned char *)s->tri_class.p, s->pclass.p, 0, n_dev);
            mode = COMPACT_CLASS;
        }
        else if (nout == 4) hipLaunchKernelGGL((k_compact_count<COMPACT_CLASS, 4>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, nohit, nocls, nokeys, 0, n_dev);
        else hipLaunchKernelGGL((k_compact_count<COMPACT_CLASS, 3>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, s->tile_counts.p, tiles, nohit, nocls, nokeys, 0, n_dev);
        hipLaunchKernelGGL(k_compact_scan, dim3(nout), dim3(1024), 0, stream, s->tile_counts.p, tiles, totals, n_dev);
        if (mode == COMPACT_FLAGS && nscatter == 3) hipLaunchKernelGGL((k_compact_scatter<COMPACT_FLAGS, 3>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, (const unsigned int *)s->tile_counts.p, tiles, o0, o1, o2, (int *)nullptr, n_dev);
        else if (mode == COMPACT_FLAGS) hipLaunchKernelGGL((k_compact_scatter<COMPACT_FLAGS, 2>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, (const unsigned int *)s->tile_counts.p, tiles, o0, o1, o2, (int *)nullptr, n_dev);
        else if (nscatter == 4) hipLaunchKernelGGL((k_compact_scatter<COMPACT_CLASS, 4>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, (const unsigned int *)s->tile_counts.p, tiles, o0, o1, o2, o3, n_dev);
        else hipLaunchKernelGGL((k_compact_scatter<COMPACT_CLASS, 3>), dim3(g), dim3(kCompactBlock), 0, stream, qin, nin, keys, (const unsigned int *)s->tile_counts.p, tiles, o0, o1, o2, (int *)nullptr, n_dev);
        launches += 3;
    };
    // ---- PathIntegrator: one vertex of every live path per iteration, two sub-passes in flight.
    // shade_stage: PathIntegrator::Li at the vertices the last trace found (class binning, one k_shade per class, queue compaction), then
    // the counts of what they spawned come back to the host.
    // `n` bounds the number of queued paths (it sizes the launches); the count itself is read on the device through n_dev.
    auto shade_stage = [&](const int *q_in, int n, int *q_out, const unsigned *n_dev) -> int {
    if (timing) timer.begin(2, stream);
    // bin the paths by the shade specialisation of the material they hit (pclass written by k_trace)
    const int n_classes = ((class_mask & 8) || escape_queue) ? 4 : 3;   // image-textured materials -- or, without them, escaped rays -- have a shade queue of their own
    compact(COMPACT_HITCLASS, q_in, n, s->pclass.p, n_classes, n_classes, &dctr->q_class[0], s->queue_c0.p, s->queue_c1.p, s->queue_c2.p, s->queue_c3.p, 0, n_dev);
    {
        int *qc[4] = {s->queue_c0.p, s->queue_c1.p, s->queue_c2.p, s->queue_c3.p};
        // 32 blocks per CU: of a k_shade grid only 2 - 3 blocks per CU are resident at a time (168 - 256 registers), and many short blocks
        // balance the end of the launch better than few long ones (8 / 16 / 32 / 64 / 128 / 1024 per CU: shade 171.4 / 167.1 / 165.7 / 165.4 /
        // 168.4 / 176.4 ms on cfg 3, profiles/r03_ab_shade_grid_cfg3.log; each block refills its LDS tables, which is what the large grids pay)
        static const int shade_bpc = getenv("GNXR_SHADE_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("GNXR_SHADE_BLOCKS_PER_CU"))) : 32;   // tuning knob
        dim3 g(grid_for(n, shade_bpc)), b(kBlock);
        // the Halton tables of the first dimensions go to LDS (device_sampler.h LdsSampler): 64 dimensions (the camera sample + 6 path vertices:
        // 18.8 KB per block; deeper vertices read global memory).  A/B on cfg 3: shade -3 % at 64 / 88 dimensions, +4 % at 112 (occupancy)
        static const int shade_lds_dims = getenv("GNXR_SHADE_LDS_DIMS") ? std::max(0, std::min(128, atoi(getenv("GNXR_SHADE_LDS_DIMS")))) : 64;   // tuning knob
        const int sdims = std::min<int>(shade_lds_dims, (int)s->cs.prime_sums.size() - 1);
        const int snperm = sdims > 0 ? s->cs.prime_sums[sdims] : 0;
        // + the scene's material and light tables when they are small (k_shade: dependent gathers along the BSDF code become LDS reads)
        static const bool shade_lds_tabs = getenv("GNXR_SHADE_LDS_TABLES") ? atoi(getenv("GNXR_SHADE_LDS_TABLES")) != 0 : true;   // experiment switch
        const int lmats = (shade_lds_tabs && s->cs.materials.size() <= 12) ? (int)s->cs.materials.size() : 0;
        const int llights = (shade_lds_tabs && nL > 0 && nL <= 16) ? nL : 0;
        const size_t slds = (sdims > 0 ? ((((size_t)snperm * 2 + 15) & ~(size_t)15) + (size_t)sdims * 32) : 0) + (size_t)lmats * sizeof(DMaterial) + (size_t)llights * sizeof(DLight);
        // the class kernels work on disjoint paths: with three or more of them (cfg 4: diffuse, glossy, Disney, escaped rays) classes 1 - 3 run on
        // two auxiliary streams beside class 0, so that the blocks of one fill the thinning end of another (fork / join with events): cfg 4 shade
        // -4 %; with two kernels of similar size (cfg 3) the same costs 1.5 %, so they stay in line (profiles/r03_ab_shade_streams_*.log;
        // GNXR_SHADE_STREAMS = 0 / 1 forces either)
        static const int shade_streams = getenv("GNXR_SHADE_STREAMS") ? atoi(getenv("GNXR_SHADE_STREAMS")) : -1;
        const int n_class_kernels = 1 + ((class_mask & 2) ? 1 : 0) + ((class_mask & 4) ? 1 : 0) + (((class_mask & 8) || escape_queue) ? 1 : 0);
        const bool fork = (shade_streams < 0 ? n_class_kernels >= 3 : (shade_streams != 0 && n_class_kernels >= 2)) && s->aux_stream[0] && s->aux_stream[1];
        hipStream_t cst[4] = {stream, stream, stream, stream};
        if (fork) {
            (void)hipEventRecord(s->ev_fork, stream);
            (void)hipStreamWaitEvent(s->aux_stream[0], s->ev_fork, 0);
            (void)hipStreamWaitEvent(s->aux_stream[1], s->ev_fork, 0);
            cst[1] = s->aux_stream[0]; cst[2] = s->aux_stream[1]; cst[3] = s->aux_stream[1];
        }
#define GX_SHADE(LMV, LTV, C)                                                                                                                        \
    do {                                                                                                                                             \
if (spheres) hipLaunchKernelGGL((k_shade<LMV, LTV, true>), g, b, slds, cst[C], sc, r, pa, (const int *)qc[C], (const unsigned int *)&dctr->q_class[C], sdims, snperm, lmats, llights); \
else hipLaunchKernelGGL((k_shade<LMV, LTV, false>), g, b, slds, cst[C], sc, r, pa, (const int *)qc[C], (const unsigned int *)&dctr->q_class[C], sdims, snperm, lmats, llights);       \
    } while (0)
#define GX_SHADE_TEX(LTV)                                                                                                                            \
    do {                                                                                                                                             \
if (spheres) hipLaunchKernelGGL((k_shade<LM_ALL, LTV, true, true>), g, b, slds, cst[3], sc, r, pa, (const int *)qc[3], (const unsigned int *)&dctr->q_class[3], sdims, snperm, lmats, llights); \
else hipLaunchKernelGGL((k_shade<LM_ALL, LTV, false, true>), g, b, slds, cst[3], sc, r, pa, (const int *)qc[3], (const unsigned int *)&dctr->q_class[3], sdims, snperm, lmats, llights);       \
    } while (0)
        if (area_only) {
            GX_SHADE(LM_DIFFUSE, LT_AREA, 0);
            if (class_mask & 2) GX_SHADE(LM_GLOSSY, LT_AREA, 1);
            if (class_mask & 4) GX_SHADE(LM_ALL, LT_AREA, 2);
            if (class_mask & 8) GX_SHADE_TEX(LT_AREA);
        } else if (area_env_only && !spheres && !(class_mask & 8)) {
            // BASELINE config 4's light set (area lights + one InfiniteAreaLight): without the delta-light and sky-box code
#define GX_SHADE_AE(LMV, C) hipLaunchKernelGGL((k_shade<LMV, LT_AREA | LT_ENV, false>), g, b, slds, cst[C], sc, r, pa, (const int *)qc[C], (const unsigned int *)&dctr->q_class[C], sdims, snperm, lmats, llights)
            GX_SHADE_AE(LM_DIFFUSE, 0);
            if (class_mask & 2) GX_SHADE_AE(LM_GLOSSY, 1);
            if (class_mask & 4) GX_SHADE_AE(LM_ALL, 2);
#undef GX_SHADE_AE
        } else {
            GX_SHADE(LM_DIFFUSE, LT_ALL, 0);
            if (class_mask & 2) GX_SHADE(LM_GLOSSY, LT_ALL, 1);
            if (class_mask & 4) GX_SHADE(LM_ALL, LT_ALL, 2);
            if (class_mask & 8) GX_SHADE_TEX(LT_ALL);
        }
#undef GX_SHADE
        if (escape_queue) {
            if (area_env_only) hipLaunchKernelGGL((k_shade_escape<LT_AREA | LT_ENV>), g, b, 0, cst[3], sc, pa, (const int *)qc[3], (const unsigned int *)&dctr->q_class[3]);
            else hipLaunchKernelGGL((k_shade_escape<LT_ALL>), g, b, 0, cst[3], sc, pa, (const int *)qc[3], (const unsigned int *)&dctr->q_class[3]);
            ++launches;
        }
        if (fork) {
            (void)hipEventRecord(s->ev_join[0], s->aux_stream[0]);
            (void)hipEventRecord(s->ev_join[1], s->aux_stream[1]);
            (void)hipStreamWaitEvent(stream, s->ev_join[0], 0);
            (void)hipStreamWaitEvent(stream, s->ev_join[1], 0);
        }
        launches += 1 + ((class_mask & 2) ? 1 : 0) + ((class_mask & 4) ? 1 : 0) + ((class_mask & 8) ? 1 : 0);
    }
    // next-vertex queue + NEE queue from the per-path flags; totals also count shadow and MIS rays
    compact(COMPACT_FLAGS, q_in, n, s->pflags.p, 4, 2, &dctr->q_next, q_out, s->queue_nee.p, nullptr, nullptr, 0, n_dev);
    if (timing) timer.end(stream);
        return GNXR_OK;
    };
    unsigned int loop_iterations = 0;
    unsigned long long new_paths = 0;   // PathIntegrator: camera rays started (their count is known to the host; the other rays are counted on the device)
    if (!whitted && !volpath) {
        // The device-driven path loop.  The samples of a call are cut into sub-passes of `kh` samples per pixel; up to `in_flight` of them
        // are alive at once, each in its own region of the state arrays, staggered in time: a launch then mixes the camera rays and first
        // bounces of one sub-pass with the thin late bounces of the others (Russian roulette and escapes leave a few hundred thousand of a
        // sub-pass's paths after five bounces), so launches stay thick while the resident state is in_flight x kh samples per pixel
        // instead of two 128-sample passes.  Queues hold slots of all regions in ascending order; results per path do not depend on who
        // shares a launch, and k_resolve runs per sub-pass in sample order, so images are unchanged bit for bit.
        //
        // The host never waits for the iteration it enqueues: every queue count stays on the device (kernels read them there; launches are
        // sized by upper bounds), and what the host needs for its decisions -- how many paths of each region are left -- it reads from a
        // ring of pinned copies that lag the GPU by up to `lag` iterations.  A stale zero is still a zero (a region only refills when the
        // host starts a sub-pass in it), and a stale count is an upper bound.  Reference loop: core/Integrator.cpp:256-293.
        struct Sub { int s0, kk; };
        std::vector<Sub> subs;
        for (int s0 = p.spp_begin; s0 < p.spp_end; s0 += kh) subs.push_back(Sub{s0, std::min(kh, p.spp_end - s0)});
        const int R = in_flight;
        struct Region { int sub = -1, started = -1; long long paths = 0; } reg[kMaxRegions];
        static const int cut_env = getenv("GNXR_PIPE_CUT") ? atoi(getenv("GNXR_PIPE_CUT")) : -1;   // tuning knob: iterations between sub-pass starts
        static const int lag_env = getenv("GNXR_LOOP_LAG") ? atoi(getenv("GNXR_LOOP_LAG")) : -1;   // tuning knob: iterations the host may run ahead
        const int lag = std::max(1, std::min(gnxr_scene::kRing - 2, lag_env >= 0 ? lag_env : 2));
        // a sub-pass lives max_depth + 2 iterations (+ the lag until the host sees that it has ended): spread the starts over that time
        const int stagger = cut_env >= 0 ? cut_env : std::max(1, (p.max_depth + 2 + lag + R - 1) / R);
        auto pa_at = [&](size_t base) { return pa.at(base); };
        int *qbuf[2] = {s->queue_a.p, s->queue_b.p};
        int in_idx = 0;
        const unsigned *cnt_ptr = &dctr->n_queue;  // where the count of the queue in flight lives on the device (k_loop_tail / k_queue_merge write it)
        size_t next_sub = 0, done_subs = 0;
        int iter = 0, last_start = -(1 << 20), newest_seen = -1;
        int ring_iter[gnxr_scene::kRing];          // iteration whose counters were copied into each ring slot (-1: none)
        for (int &v : ring_iter) v = -1;
        Counters seen;
        memset(&seen, 0, sizeof(seen));
        long long guard = 0;
        while (done_subs < subs.size()) {
            // 1. the newest copy of the counters that has arrived (never the iteration just enqueued, unless the GPU is already through it)
            {
                // at most `lag` iterations ahead of what has been seen: wait for the oldest outstanding copy beyond that
                int oldest_needed = iter - 1 - lag;
                for (int j = newest_seen + 1; j <= oldest_needed; ++j) {
                    const int slot = j % gnxr_scene::kRing;
                    if (j >= 0 && ring_iter[slot] == j) HIP_TRY(hipEventSynchronize(s->ring_ev[slot]));
                }
                for (int j = iter - 1; j > newest_seen; --j) {
                    const int slot = ((j % gnxr_scene::kRing) + gnxr_scene::kRing) % gnxr_scene::kRing;
                    if (j < 0 || ring_iter[slot] != j) continue;
                    if (hipEventQuery(s->ring_ev[slot]) == hipSuccess) { seen = s->h_ring[slot]; newest_seen = j; break; }
                }
                (void)hipGetLastError();   // hipEventQuery reports "not ready" as an error
            }
            // 2. sub-passes none of whose paths continues are complete once their last light estimates are added (stream order: the
            //    k_nee_combine of the iteration that produced the zero is already enqueued): colObj += Li in sample order
            //    -- and in sub-pass order: a sub-pass that ends before an earlier one keeps its region until that one is added
            for (bool progress = true; progress;) {
                progress = false;
                for (int rg = 0; rg < R; ++rg) {
                    Region &g = reg[rg];
                    if (g.sub == (int)done_subs && newest_seen > g.started && seen.region_alive[rg] == 0) {
                        hipLaunchKernelGGL(k_resolve, dim3(grid_for(r.npix)), dim3(kBlock), 0, stream, pa_at((size_t)rg * half), s->accum.p, r.npix, subs[g.sub].kk);
                        ++launches; ++passes; ++done_subs;
                        g.sub = -1;
                        progress = true;
                    }
                }
            }
            if (done_subs == subs.size()) break;
            // upper bound of the paths queued for this iteration's shade stage
            bool any_active = false;
            long long n_upper = 0;
            for (int rg = 0; rg < R; ++rg) {
                const Region &g = reg[rg];
                if (g.sub < 0) continue;
                any_active = true;
                n_upper += newest_seen > g.started ? std::min<long long>(g.paths, seen.region_alive[rg]) : g.paths;
            }
            // A. shade what the last trace found; survivors go to the buffer that does not hold the input queue
            const int out_idx = 1 - in_idx;
            if (any_active) {
                if ((rc = shade_stage(qbuf[in_idx], (int)n_upper, qbuf[out_idx], cnt_ptr)) != GNXR_OK) { set_error("HIP runtime error in the path loop: %s", hipGetErrorString(hipGetLastError())); return rc; }
                hipLaunchKernelGGL(k_loop_tail, dim3(1), dim3(64), 0, stream, (const int *)qbuf[out_idx], dctr, R, (int)half, (unsigned)iter);
                ++launches;
                const int slot = iter % gnxr_scene::kRing;
                HIP_TRY(hipMemcpyAsync(&s->h_ring[slot], dctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipEventRecord(s->ring_ev[slot], stream));
                ring_iter[slot] = iter;
            }
            // B. start the next sub-pass in a free region
            int trace_idx = out_idx;
            long long n_trace_upper = n_upper;
            int free_rg = -1;
            for (int rg = 0; rg < R && free_rg < 0; ++rg) if (reg[rg].sub < 0) free_rg = rg;
            const bool start = next_sub < subs.size() && free_rg >= 0 && (!any_active || iter - last_start >= stagger);
            if (start) {
                const Sub &nw = subs[next_sub];
                const int n_new = r.npix * nw.kk;
                const size_t base = (size_t)free_rg * half;
                hipLaunchKernelGGL(k_raygen, dim3(grid_for(n_new)), dim3(kBlock), 0, stream, sc, r, pa_at(base), n_new, nw.s0);
                // survivors + every slot of the new sub-pass, ascending: into the buffer the shaded queue came from
                hipLaunchKernelGGL(k_queue_merge, dim3(grid_for(n_upper + n_new)), dim3(kBlock), 0, stream, (const int *)qbuf[out_idx], dctr, any_active ? 0 : 1, free_rg, (int)base, n_new, qbuf[in_idx]);
                launches += 2;
                trace_idx = in_idx; n_trace_upper = n_upper + n_new;
                new_paths += (unsigned long long)n_new;
                reg[free_rg].sub = (int)next_sub; reg[free_rg].started = iter; reg[free_rg].paths = n_new;
                last_start = iter;
                ++next_sub;
            }
            // C. continuation rays (and new camera rays), shadow and MIS rays of the vertices just shaded; then their light estimates
            if (any_active || start) {
                TraceWork tw{qbuf[trace_idx], (int)n_trace_upper, s->queue_nee.p, any_active ? (int)n_upper : 0, nullptr, reinterpret_cast<unsigned char *>(s->nee_vis.p), cnt_ptr,
                             any_active ? (const unsigned *)&dctr->q_nee : nullptr};
                launch_trace(tw, 0, 0, false);
                if (any_active) {
                    if (timing) timer.begin(1, stream);
                    hipLaunchKernelGGL(k_nee_combine, dim3(grid_for(n_upper)), dim3(kBlock), 0, stream, pa, (const int *)s->queue_nee.p, (int)n_upper, reinterpret_cast<const unsigned char *>(s->nee_vis.p),
                                       (const unsigned *)&dctr->q_nee);
                    if (timing) timer.end(stream);
                    ++launches;
                }
                in_idx = trace_idx;
            }
            ++iter; ++loop_iterations;
            if (++guard > (1ll << 24)) { set_error("path loop did not terminate"); return GNXR_ERR_INVALID; }
        }
    } else
    for (int s0 = p.spp_begin; s0 < p.spp_end; s0 += k) {
        int kk = std::min(k, p.spp_end - s0);
        int n_paths = r.npix * kk;
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(n_paths)), dim3(kBlock), 0, stream, sc, r, pa, n_paths, s0);
        ++launches;
        int n = n_paths;
        const int *q_in = nullptr;            // paths alive at this vertex (nullptr == identity), ascending
        int *q_cur = s->queue_a.p, *q_other = s->queue_b.p;
        int guard = 0;
        if (whitted) {
            // depth-first recursion per path (whitted_kernel.hip.h): the path's ray + the previous vertex's shadow rays per round
            hipLaunchKernelGGL(k_whitted_init, dim3(grid_for(n_paths)), dim3(kBlock), 0, stream, pa, wa, n_paths);
            ++launches;
            if (textured) { hipLaunchKernelGGL(k_whitted_init_diff, dim3(grid_for(n_paths)), dim3(kBlock), 0, stream, sc, r, pa, wa, n_paths); ++launches; }
            int n_cl = n, n_shp = 0;
            const int *q_cl = nullptr;
            unsigned long long *d_shadow = &dctr->whitted_shadow;
            while (n > 0) {
                if (n_shp > 0) {
                    hipLaunchKernelGGL(k_whitted_expand, dim3(grid_for((long long)n_shp * n_records)), dim3(kBlock), 0, stream, (const int *)s->queue_nee.p, n_shp, n_records, (int)cap, s->wh_rec.p);
                    ++launches;
                }
                launch_trace(TraceWork{q_cl, n_cl, s->wh_rec.p, n_shp * n_records}, 0, 0);
                if (timing) timer.begin(2, stream);
#define GX_WH2(MODEV, LTV, SPHV)                                                                                                                     \
    do {                                                                                                                                             \
        if (textured) hipLaunchKernelGGL((k_whitted_step<MODEV, LTV, SPHV, true>), dim3(grid_for(n)), dim3(kBlock), 0, stream, sc, r, pa, wa, q_in, n, d_shadow); \
        else hipLaunchKernelGGL((k_whitted_step<MODEV, LTV, SPHV>), dim3(grid_for(n)), dim3(kBlock), 0, stream, sc, r, pa, wa, q_in, n, d_shadow);   \
    } while (0)
#define GX_WH(LTV, SPHV) do { if (wmode == WM_WHITTED) GX_WH2(WM_WHITTED, LTV, SPHV); else if (wmode == WM_DIRECT_ONE) GX_WH2(WM_DIRECT_ONE, LTV, SPHV); else GX_WH2(WM_DIRECT_ALL, LTV, SPHV); } while (0)
                if (area_only) { if (spheres) GX_WH(LT_AREA, true); else GX_WH(LT_AREA, false); }
                else { if (spheres) GX_WH(LT_ALL, true); else GX_WH(LT_ALL, false); }
#undef GX_WH
#undef GX_WH2
                ++launches;
                compact(COMPACT_FLAGS, q_in, n, s->pflags.p, 4, 3, &dctr->q_next, q_cur, s->queue_nee.p, s->queue_c0.p);
                if (timing) timer.end(stream);
                HIP_TRY(hipMemcpyAsync(s->h_counters, dctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                n = (int)s->h_counters->q_next;
                n_shp = (int)s->h_counters->q_nee;
                n_cl = (int)s->h_counters->q_shadow;   // count of pflags bit2: paths with a closest-hit ray to trace
                q_cl = s->queue_c0.p;
                q_in = q_cur;
                std::swap(q_cur, q_other);
                if (++guard > (1 << 20)) { set_error("path loop did not terminate"); return GNXR_ERR_INVALID; }
            }
        } else if (volpath) {
            // one closest-hit ray per live path and round: k_trace -> k_vol_step -> compaction (vol_kernel.hip.h)
            hipLaunchKernelGGL(k_vol_init, dim3(grid_for(n_paths)), dim3(kBlock), 0, stream, pa, va, n_paths);
            ++launches;
            int n_media = r.cam.medium >= 0 ? n : 0;      // paths whose ray in flight travels inside a medium
            const int *q_media = nullptr;
            // packing (k_vol_pack, vol_kernel.hip.h): the two sets of the per-path arrays that carry state across rounds
            static const bool vol_pack = getenv("GNXR_VOL_PACK") ? atoi(getenv("GNXR_VOL_PACK")) != 0 : true;   // experiment switch
            auto pack_set = [&](bool alt) {
                VolPackSet ps;
                if (!alt) {
                    float4 *a[kVolPackF4] = {s->L.p, reinterpret_cast<float4 *>(s->vol_vs.p), s->vol_n1.p, s->vol_f.p, s->vol_Li.p, s->vol_Tr.p, s->vol_Ld.p, s->mis_Y.p, s->vol_mres.p};
                    for (int i = 0; i < kVolPackF4; ++i) ps.f4[i] = a[i];
                    gnxr_scene::record_ptrs(s->rec, ps.rec);
                    ps.state = s->vol_state.p; ps.orig = s->vol_orig.p;
                } else {
                    for (int i = 0; i < kVolPackF4; ++i) ps.f4[i] = s->vol_alt[i].p;
                    gnxr_scene::record_ptrs(s->vol_alt_rec, ps.rec);
                    ps.state = s->vol_alt_state.p; ps.orig = s->vol_alt_orig.p;
                }
                return ps;
            };
            auto bind_set = [&](const VolPackSet &ps) {   // point the kernels' views at a set
                pa.bind_records(ps.rec); va.bind_records(ps.rec);
                pa.L = ps.f4[0]; va.vs = reinterpret_cast<int4 *>(ps.f4[1]); va.n1 = ps.f4[2]; va.f = ps.f4[3]; va.Li = ps.f4[4]; va.Tr = ps.f4[5]; va.Ld = ps.f4[6]; pa.mis_Y = va.mis_Y = ps.f4[7]; va.mres = ps.f4[8];
                va.state = ps.state; va.orig = ps.orig;
            };
            bool in_alt = false;
            long long span = n_paths;     // the live paths lie in slots [0, span)
            bind_set(pack_set(false));
            while (n > 0) {
                launch_trace(TraceWork{q_in, n, nullptr, 0}, 0, 0);
                if (n_media > 0) {
                    (void)hipMemsetAsync(&dctr->cursor, 0, sizeof(unsigned int), stream);
                    // (4 / 5 / 8 / 16 / 32 blocks per CU: k_vol_media 0.357 s per 3 x 256 spp of cfg 5 each time -- its waves persist; profiles/r03_ab_vol_step_occupancy_cfg5.log)
                    int blocks = (int)std::min<long long>((long long)g_num_cus * 8, ((long long)n_media + kBlock - 1) / kBlock);
                    const int vm_cap = getenv("GNXR_VOLMEDIA_STEP_CAP") ? std::max(0, atoi(getenv("GNXR_VOLMEDIA_STEP_CAP"))) : 64;   // tuning knob, read per launch so that a test can vary it (0: no cap)
                    if (timing) timer.begin(1, stream);
                    const long long mwaves = (long long)blocks * (kBlock / 64);
                    const int mchunk = (int)std::min<long long>(kMediaChunk, std::max<long long>(64, ((n_media + mwaves - 1) / mwaves + 63) / 64 * 64));
                    if (count_wide) hipLaunchKernelGGL(k_vol_media<true>, dim3(blocks), dim3(kBlock), (size_t)kVmRecDwords * kVmStride * sizeof(int), stream, sc, mt, pa, va, q_media, n_media, &dctr->cursor, mchunk, dctr, vm_cap);
                    else hipLaunchKernelGGL(k_vol_media<false>, dim3(blocks), dim3(kBlock), (size_t)kVmRecDwords * kVmStride * sizeof(int), stream, sc, mt, pa, va, q_media, n_media, &dctr->cursor, mchunk, dctr, vm_cap);
                    media_segments += (unsigned long long)n_media;
                    if (timing) timer.end(stream);
                    ++launches;
                }
                if (timing) timer.begin(2, stream);
// bin the live paths by state (main ray / shadow-ray segment / scattering-ray segment), one k_vol_step instantiation per bin
// (32 blocks per CU for the step kernels, as for k_shade: many short blocks balance the end of a launch better; cfg 5 -2 %)
                compact(COMPACT_CLASS, q_in, n, va.state, 3, 3, &dctr->q_class[0], s->queue_c0.p, s->queue_c1.p, s->queue_c2.p);
                {
                    int *qc[3] = {s->queue_c0.p, s->queue_c1.p, s->queue_c2.p};
                    // the scene's material and light tables go to LDS when they are small (as for k_shade)
                    const int vmats = s->cs.materials.size() <= 12 ? (int)s->cs.materials.size() : 0, vlights = (nL > 0 && nL <= 16) ? nL : 0;
                    const size_t vlds = (size_t)vmats * sizeof(DMaterial) + (size_t)vlights * sizeof(DLight);
#define GX_VS1(LMV, LTV, ST) hipLaunchKernelGGL((k_vol_step<LMV, LTV, ST>), dim3(grid_for(n, 32)), dim3(kBlock), vlds, stream, sc, mt, r, pa, va, (const int *)qc[ST], (const unsigned int *)&dctr->q_class[ST], vmats, vlights)
#define GX_VS(LMV, LTV) do { GX_VS1(LMV, LTV, VS_MAIN); GX_VS1(LMV, LTV, VS_SHADOW); GX_VS1(LMV, LTV, VS_MIS); } while (0)
#define GX_VST1(LTV, ST) hipLaunchKernelGGL((k_vol_step<LM_ALL, LTV, ST, true>), dim3(grid_for(n, 32)), dim3(kBlock), vlds, stream, sc, mt, r, pa, va, (const int *)qc[ST], (const unsigned int *)&dctr->q_class[ST], vmats, vlights)
#define GX_VST(LTV) do { GX_VST1(LTV, VS_MAIN); GX_VST1(LTV, VS_SHADOW); GX_VST1(LTV, VS_MIS); } while (0)
                    if (textured) { if (area_only) GX_VST(LT_AREA); else GX_VST(LT_ALL); }
                    else if (area_only) { if (class_mask <= 1) GX_VS(LM_DIFFUSE, LT_AREA); else if (class_mask <= 3) GX_VS(LM_GLOSSY, LT_AREA); else GX_VS(LM_ALL, LT_AREA); }
                    else { if (class_mask <= 1) GX_VS(LM_DIFFUSE, LT_ALL); else if (class_mask <= 3) GX_VS(LM_GLOSSY, LT_ALL); else GX_VS(LM_ALL, LT_ALL); }
#undef GX_VST
#undef GX_VST1
#undef GX_VS
#undef GX_VS1
                    launches += 2;
                }
                ++launches;
                compact(COMPACT_FLAGS, q_in, n, s->pflags.p, 4, 2, &dctr->q_next, q_cur, s->queue_nee.p, nullptr);
                if (timing) timer.end(stream);
                HIP_TRY(hipMemcpyAsync(s->h_counters, dctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                n = (int)s->h_counters->q_next;
                n_media = (int)s->h_counters->q_nee;
                q_media = s->queue_nee.p;
                q_in = q_cur;
                std::swap(q_cur, q_other);
                if (vol_pack && n >= (1 << 16) && 2ll * n <= span) {
                    // the survivors fill at most half of the span they are spread over: move them to the front of the other set
                    const VolPackSet from = pack_set(in_alt), to = pack_set(!in_alt);
                    hipLaunchKernelGGL(k_vol_pack, dim3(grid_for(n)), dim3(kBlock), 0, stream, q_in, n, from, to, s->vol_newslot.p);
                    if (n_media > 0) hipLaunchKernelGGL(k_vol_remap, dim3(grid_for(n_media)), dim3(kBlock), 0, stream, s->queue_nee.p, n_media, (const int *)s->vol_newslot.p);
                    launches += 2;
                    in_alt = !in_alt;
                    bind_set(to);
                    q_in = nullptr;   // the queue is the identity again
                    span = n;
                }
                if (++guard > (1 << 20)) { set_error("path loop did not terminate"); return GNXR_ERR_INVALID; }
            }
            if (in_alt) bind_set(pack_set(false));   // the next pass's k_raygen writes the primary set again
        }
        {
            PathArrays pr = pa;
            if (volpath) pr.L = s->vol_Lout.p;   // VolPath: results sit at the paths' original slots (packing moves the working state)
            hipLaunchKernelGGL(k_resolve, dim3(grid_for(r.npix)), dim3(kBlock), 0, stream, pr, s->accum.p, r.npix, kk);
        }
        ++launches;
        ++passes;
        if (timing) { HIP_TRY(hipStreamSynchronize(stream)); timer.collect(); }
    }
    hipLaunchKernelGGL(k_finish, dim3(grid_for(r.npix)), dim3(kBlock), 0, stream, r, (const float4 *)s->accum.p, (float4 *)d_rgba_out);
    ++launches;
    HIP_TRY(hipEventRecord(ev1, stream));
    HIP_TRY(hipMemcpyAsync(s->h_counters, dctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    if (timing) timer.collect();
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        if (!whitted && !volpath) {   // the device-driven loop counts on the device; only the camera rays are known to the host
            rays_closest = new_paths + s->h_counters->rays_continue + s->h_counters->rays_mis;
            rays_any = s->h_counters->rays_shadow;
            rays_mis = s->h_counters->rays_mis;
        }
        if (volpath) {   // a segment k_vol_media left at its step cap was traced and handed over once more: the same ray, counted once
            rays_closest -= s->h_counters->media_cont;
            media_segments -= s->h_counters->media_cont;
        }
        stats->rays_closest = rays_closest + (whitted ? s->h_counters->whitted_mis : 0);
        stats->rays_any = whitted ? s->h_counters->whitted_shadow : rays_any;
        stats->camera_samples = (uint64_t)r.npix * nsamples;
        stats->nodes_visited = s->h_counters->nodes;
        stats->tris_tested = s->h_counters->tris;
        stats->seconds_render = ms * 1e-3;
        stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        stats->kernel_launches = launches;
        stats->passes = passes;
        stats->passes_in_flight = (uint32_t)in_flight;
        stats->loop_iterations = loop_iterations;
        {   // what this render keeps resident per path slot: the float4 / uint2 / int state arrays, the queues and the per-path bytes
            const unsigned long long per_slot = (2ull * kRecGroups + 2) * sizeof(float4) + 8ull * sizeof(int) + 2 + sizeof(unsigned int);   // five record groups + mis_Y + L, hit + seven queues, pflags + pclass, nee_vis
            stats->state_bytes = (unsigned long long)cap * per_slot + (volpath ? (unsigned long long)cap * (6ull * sizeof(float4) + sizeof(int4) + 1) : 0ull);
        }
        stats->seconds_closest = timer.seconds[0]; stats->seconds_nee = timer.seconds[1]; stats->seconds_shade = timer.seconds[2];
        stats->seconds_trace = timer.seconds[0] + timer.seconds[1];
        stats->launches_closest = timer.launches[0]; stats->launches_nee = timer.launches[1];
        stats->rays_closest_nee = rays_mis;
        stats->media_segments = media_segments;
        stats->media_steps = s->h_counters->media_steps;
        stats->leaf_retests = s->h_counters->retests;
        stats->nodes_from_memory = s->h_counters->nodes_global;
    }
    return GNXR_OK;
}

// Several devices behind one handle (gnxr_init_devices): the rows of this render are dealt round-robin over the devices, each
// device renders its rows concurrently (one host thread and one stream per device, nothing exchanged during rendering) into a
// full-size plane of its own, and the rows are then copied into the caller's image on the primary device (peer copies over xGMI,
// one strided 2D copy per device).  A single device takes the direct path.
static int render_sharded(gnxr_scene *s, const gnxr_render_params *pin, void *d_rgba_out, void *hip_stream, gnxr_stats *stats) {
    if (!s || !pin || !d_rgba_out) { set_error("null argument"); return GNXR_ERR_INVALID; }
    const int nd = 1 + (int)s->replicas.size();
    if (nd == 1) return render_one(s, pin, d_rgba_out, hip_stream, stats);
    gnxr_render_params base = *pin;
    if (base.shard_count <= 0) base.shard_count = 1;
    if (base.shard_rows <= 0) base.shard_rows = 1;
    if (base.shard_rows != 1) { set_error("multi-device rendering deals single rows: shard_rows must be 1"); return GNXR_ERR_UNSUPPORTED; }
    if (base.width <= 0 || base.height <= 0 || base.shard_index < 0 || base.shard_index >= base.shard_count) { set_error("invalid render parameters"); return GNXR_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lock(s->render_mutex);
    const size_t npx = (size_t)base.width * base.height;
    std::vector<int> rcs(nd, GNXR_OK);
    std::vector<std::string> errs(nd);
    std::vector<gnxr_stats> sts(nd);
    std::vector<int> staged(nd, 0);   // rows a shard left in its pinned staging buffer (no peer access between its device and the primary)
    hipStream_t caller = (hipStream_t)hip_stream;
    int rc = s->bind();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(caller));   // the image must be safe to write from the other devices' streams
    const bool reserve_only = g_reserve_only;   // (thread-local: handed to the worker threads)
    auto worker = [&](int i) {
        g_reserve_only = reserve_only;
        gnxr_scene *r = i == 0 ? s : s->replicas[i - 1].get();
        gnxr_render_params p = base;   // rows y == shard_index (mod shard_count) of the caller, every nd-th of them
        p.shard_index = base.shard_index + base.shard_count * i;
        p.shard_count = base.shard_count * nd;
        int rc_ = r->bind();
        void *dst = d_rgba_out;
        if (rc_ == GNXR_OK && i > 0) { rc_ = r->shard_out.alloc(npx); dst = r->shard_out.p; }
        if (rc_ == GNXR_OK) rc_ = render_one(r, &p, dst, i == 0 ? hip_stream : nullptr, &sts[i]);
        if (rc_ == GNXR_OK && i > 0 && !reserve_only) {
            // rows p.shard_index, + p.shard_count, ...
            const int first = p.shard_index, step = p.shard_count;
            const int rows = first < base.height ? (base.height - first + step - 1) / step : 0;
            const size_t rowb = (size_t)base.width * sizeof(float4);
            const bool peer = (size_t)i < g_peer_ok.size() ? g_peer_ok[i] != 0 : r->device == s->device;
            if (rows > 0 && peer) {
                // one strided copy into the primary's image (peer access was enabled both ways at init: the runtime routes it over the link)
                hipError_t e = hipMemcpy2DAsync((char *)d_rgba_out + (size_t)first * rowb, rowb * step, (const char *)dst + (size_t)first * rowb, rowb * step, rowb, rows,
                                                hipMemcpyDeviceToDevice, nullptr);
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
                if (e != hipSuccess) { set_error("peer copy from device %d failed: %s", r->device, hipGetErrorString(e)); rc_ = hip_status(e); }
            } else if (rows > 0) {
                // no peer access for this pair: the shard's rows go to a pinned host buffer here (packed), and the primary uploads them after the join
                if (r->h_stage_bytes < rowb * rows) {
                    if (r->h_stage) (void)hipHostFree(r->h_stage);
                    r->h_stage = nullptr; r->h_stage_bytes = 0;
                    if (hipHostMalloc(&r->h_stage, rowb * rows) != hipSuccess) { set_error("hipHostMalloc of the %zu-byte staging buffer for device %d failed", rowb * rows, r->device); rc_ = GNXR_ERR_OOM; }
                    else r->h_stage_bytes = rowb * rows;
                }
                if (rc_ == GNXR_OK) {
                    hipError_t e = hipMemcpy2D(r->h_stage, rowb, (const char *)dst + (size_t)first * rowb, rowb * step, rowb, rows, hipMemcpyDeviceToHost);
                    if (e != hipSuccess) { set_error("download of device %d's rows failed: %s", r->device, hipGetErrorString(e)); rc_ = hip_status(e); }
                    else staged[i] = rows;
                }
            }
        }
        rcs[i] = rc_;
        if (rc_ != GNXR_OK) errs[i] = get_error();
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nd; ++i) pool.emplace_back(worker, i);
    worker(0);
    for (auto &t : pool) t.join();
    (void)s->bind();
    for (int i = 1; i < nd; ++i) {   // host-staged shards: upload their rows into the image on the primary device
        if (staged[i] <= 0 || rcs[i] != GNXR_OK) continue;
        gnxr_scene *r = s->replicas[i - 1].get();
        const int first = base.shard_index + base.shard_count * i, step = base.shard_count * nd;
        const size_t rowb = (size_t)base.width * sizeof(float4);
        hipError_t e = hipMemcpy2D((char *)d_rgba_out + (size_t)first * rowb, rowb * step, r->h_stage, rowb, rowb, staged[i], hipMemcpyHostToDevice);
        if (e != hipSuccess) { rcs[i] = hip_status(e); errs[i] = std::string("upload of the staged rows failed: ") + hipGetErrorString(e); }
    }
    for (int i = 0; i < nd; ++i) if (rcs[i] != GNXR_OK) { set_error("device %d: %s", i == 0 ? s->device : s->replicas[i - 1]->device, errs[i].c_str()); return rcs[i]; }
    if (stats) {
        *stats = sts[0];
        for (int i = 1; i < nd; ++i) {
            const gnxr_stats &t = sts[i];
            stats->rays_closest += t.rays_closest; stats->rays_any += t.rays_any; stats->camera_samples += t.camera_samples;
            stats->nodes_visited += t.nodes_visited; stats->tris_tested += t.tris_tested; stats->kernel_launches += t.kernel_launches;
            stats->rays_closest_nee += t.rays_closest_nee; stats->media_segments += t.media_segments; stats->media_steps += t.media_steps; stats->leaf_retests += t.leaf_retests;
            stats->nodes_from_memory += t.nodes_from_memory;
            stats->seconds_render = std::max(stats->seconds_render, t.seconds_render); stats->seconds_total = std::max(stats->seconds_total, t.seconds_total);
            stats->passes = std::max(stats->passes, t.passes);
        }
    }
    return GNXR_OK;
}

int gnxr_render_device(gnxr_scene *s, const gnxr_render_params *pin, void *d_rgba_out, void *hip_stream, gnxr_stats *stats) {
    return render_sharded(s, pin, d_rgba_out, hip_stream, stats);
}

// Allocates the path state a render with these parameters needs (on every device of the handle) without rendering: a caller that times
// its renders -- or a viewer that must not stall in its first frame -- pays for the allocation up front.  State only ever grows.
int gnxr_render_reserve(gnxr_scene *s, const gnxr_render_params *pin) {
    if (!s || !pin) { set_error("null argument"); return GNXR_ERR_INVALID; }
    g_reserve_only = true;
    int dummy = 0;   // never written: render_one returns before anything touches the output
    const int rc = render_sharded(s, pin, &dummy, nullptr, nullptr);
    g_reserve_only = false;
    return rc;
}

int gnxr_render(gnxr_scene *s, const gnxr_render_params *p, float *rgba_out, gnxr_stats *stats) {
    if (!s || !p || !rgba_out) { set_error("null argument"); return GNXR_ERR_INVALID; }
    if (p->width <= 0 || p->height <= 0) { set_error("invalid image size"); return GNXR_ERR_INVALID; }
    size_t npx = (size_t)p->width * p->height;
    std::lock_guard<std::recursive_mutex> lock(s->render_mutex);   // s->out is shared by every gnxr_render on this handle
    if (int brc = s->bind()) return brc;
    int rc = s->out.alloc(npx);
    if (rc) return rc;
    HIP_TRY(hipMemset(s->out.p, 0, npx * sizeof(float4)));
    rc = render_sharded(s, p, s->out.p, nullptr, stats);
    if (rc) return rc;
    // copy back only the rows this shard owns
    int sc = p->shard_count > 0 ? p->shard_count : 1, sr = p->shard_rows > 0 ? p->shard_rows : 1;
    if (sc == 1) {
        HIP_TRY(hipMemcpy(rgba_out, s->out.p, npx * sizeof(float4), hipMemcpyDeviceToHost));
    } else {
        for (int y = 0; y < p->height; ++y)
            if ((y / sr) % sc == p->shard_index)
                HIP_TRY(hipMemcpy(rgba_out + (size_t)y * p->width * 4, s->out.p + (size_t)y * p->width, (size_t)p->width * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return GNXR_OK;
}

int gnxr_scene_bvh(const gnxr_scene *s, float *bounds6, int32_t *meta3, int32_t *ordered, int64_t node_capacity, int64_t *n_nodes) {
    if (!s || !n_nodes) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    const CompiledScene &cs = s->cs;
    *n_nodes = (int64_t)cs.nodes.size();
    if (!bounds6 || !meta3 || !ordered || node_capacity < *n_nodes) return GNXR_OK;
    for (size_t i = 0; i < cs.nodes.size(); ++i) {
        const DNode &n = cs.nodes[i];
        bounds6[6 * i + 0] = n.lo[0]; bounds6[6 * i + 1] = n.lo[1]; bounds6[6 * i + 2] = n.lo[2];
        bounds6[6 * i + 3] = n.hi0; bounds6[6 * i + 4] = n.hi1; bounds6[6 * i + 5] = n.hi2;
        const int nPrims = (int)(n.meta & 0xffffu);
        meta3[3 * i + 0] = n.offset; meta3[3 * i + 1] = nPrims; meta3[3 * i + 2] = nPrims ? 0 : (int)(n.meta >> 16);
    }
    for (size_t i = 0; i < cs.tris.size(); ++i) ordered[i] = cs.tris[i].prim;
    return GNXR_OK;
}

int gnxr_trace_closest(gnxr_scene *s, const gnxr_ray *rays, int64_t n, gnxr_hit *hits) {
    if (!s || !rays || !hits || n < 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    if (n == 0) return GNXR_OK;
    if (int brc = s->bind()) return brc;
    DevBuf<gnxr_ray> dr;
    DevBuf<gnxr_hit> dh;
    int rc;
    if ((rc = dr.upload(rays, (size_t)n)) || (rc = dh.alloc((size_t)n))) return rc;
    DScene sc = s->device_scene(1, 1);
    if (s->stack_size > 32) hipLaunchKernelGGL((k_trace_closest_api<64>), dim3(grid_for(n, 2)), dim3(kBlock), 0, 0, sc, (const gnxr_ray *)dr.p, (long long)n, dh.p);
    else hipLaunchKernelGGL((k_trace_closest_api<32>), dim3(grid_for(n, 5)), dim3(kBlock), 0, 0, sc, (const gnxr_ray *)dr.p, (long long)n, dh.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits, dh.p, (size_t)n * sizeof(gnxr_hit), hipMemcpyDeviceToHost));
    return GNXR_OK;
}
int gnxr_trace_any(gnxr_scene *s, const gnxr_ray *rays, int64_t n, uint8_t *occluded) {
    if (!s || !rays || !occluded || n < 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    if (n == 0) return GNXR_OK;
    if (int brc = s->bind()) return brc;
    DevBuf<gnxr_ray> dr;
    DevBuf<unsigned char> dob;
    int rc;
    if ((rc = dr.upload(rays, (size_t)n)) || (rc = dob.alloc((size_t)n))) return rc;
    DScene sc = s->device_scene(1, 1);
    if (s->stack_size > 32) hipLaunchKernelGGL((k_trace_any_api<64>), dim3(grid_for(n, 2)), dim3(kBlock), 0, 0, sc, (const gnxr_ray *)dr.p, (long long)n, dob.p);
    else hipLaunchKernelGGL((k_trace_any_api<32>), dim3(grid_for(n, 5)), dim3(kBlock), 0, 0, sc, (const gnxr_ray *)dr.p, (long long)n, dob.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(occluded, dob.p, (size_t)n, hipMemcpyDeviceToHost));
    return GNXR_OK;
}

// sampler tables without a scene (probes)
static int probe_tables(CompiledScene *cs, DevBuf<uint16_t> *perms, DevBuf<int32_t> *primes, DevBuf<int32_t> *sums, DevBuf<uint32_t> *magic, DSamplerTables *st,
                        int W, int H) {
    // a one-triangle scene is the cheapest way to reuse the table builder
    float v[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0};
    int32_t idx[3] = {0, 1, 2}, mat[1] = {-1}, lt[1] = {-1};
    gnxr_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = GNXR_ABI_VERSION; d.n_vertices = 3; d.n_triangles = 1; d.vertices = v; d.indices = idx; d.tri_material = mat; d.tri_light = lt;
    if (!compile_scene(&d, cs)) return GNXR_ERR_INVALID;
    int rc;
    if ((rc = perms->upload(cs->perms)) || (rc = primes->upload(cs->primes)) || (rc = sums->upload(cs->prime_sums)) || (rc = magic->upload(cs->prime_magic))) return rc;
    st->perms = perms->p; st->primes = primes->p; st->prime_sums = sums->p; st->prime_magic = magic->p;
    st->h = make_halton(W, H);
    return GNXR_OK;
}

int gnxr_sample_halton(int32_t width, int32_t height, const int32_t *px, const int32_t *py, const int64_t *sidx, const int32_t *dim, int64_t n, float *out) {
    if (!px || !py || !sidx || !dim || !out || n < 0 || width <= 0 || height <= 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return GNXR_OK;
    CompiledScene cs;
    DevBuf<uint16_t> perms; DevBuf<int32_t> primes, sums; DevBuf<uint32_t> magic;
    DSamplerTables st;
    if ((rc = probe_tables(&cs, &perms, &primes, &sums, &magic, &st, width, height))) return rc;
    DevBuf<int32_t> dpx, dpy, ddim; DevBuf<long long> ds; DevBuf<float> dout;
    if ((rc = dpx.upload(px, n)) || (rc = dpy.upload(py, n)) || (rc = ddim.upload(dim, n)) || (rc = ds.upload((const long long *)sidx, n)) || (rc = dout.alloc(n))) return rc;
    {   // the probe takes the render path's 32-bit accumulator wherever its indices allow it
        unsigned long long smax = 0;
        for (int64_t i = 0; i < n; ++i) smax = std::max<unsigned long long>(smax, (unsigned long long)std::max<int64_t>(0, sidx[i]));
        const unsigned long long bound = (unsigned long long)st.h.stride * (smax + 2);
        st.h.base32_max = bound >= (1ull << 32) ? 0 : (int32_t)std::min<unsigned long long>(0x7fffffffull, 0xffffffffull / bound);
    }
    hipLaunchKernelGGL(k_halton_probe, dim3(grid_for(n)), dim3(kBlock), 0, 0, st, (const int *)dpx.p, (const int *)dpy.p, (const long long *)ds.p, (const int *)ddim.p, (long long)n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return GNXR_OK;
}

int gnxr_camera_rays(const gnxr_camera *cam, int32_t width, int32_t height, const int32_t *px, const int32_t *py, const int64_t *sidx, int64_t n, float *o_out,
                     float *d_out) {
    if (!cam || !px || !py || !sidx || !o_out || !d_out || n < 0 || width <= 0 || height <= 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return GNXR_OK;
    CompiledScene cs;
    DevBuf<uint16_t> perms; DevBuf<int32_t> primes, sums; DevBuf<uint32_t> magic;
    DSamplerTables st;
    if ((rc = probe_tables(&cs, &perms, &primes, &sums, &magic, &st, width, height))) return rc;
    DCamera dc = make_camera(*cam, width, height, -1);
    DevBuf<int32_t> dpx, dpy; DevBuf<long long> ds; DevBuf<float> dob, dd;
    if ((rc = dpx.upload(px, n)) || (rc = dpy.upload(py, n)) || (rc = ds.upload((const long long *)sidx, n)) || (rc = dob.alloc(3 * n)) || (rc = dd.alloc(3 * n))) return rc;
    hipLaunchKernelGGL(k_camera_probe, dim3(grid_for(n)), dim3(kBlock), 0, 0, st, dc, (const int *)dpx.p, (const int *)dpy.p, (const long long *)ds.p, (long long)n, dob.p, dd.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(o_out, dob.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d_out, dd.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return GNXR_OK;
}

int gnxr_light_grid_table(gnxr_scene *s, int32_t strategy, int32_t on_host, float *out, int64_t capacity, int64_t *n_floats) {
    if (!s || !n_floats) { set_error("null argument"); return GNXR_ERR_INVALID; }
    std::lock_guard<std::recursive_mutex> lock(s->render_mutex);
    if (int brc = s->bind()) return brc;
    int rc = s->ensure_grid(strategy, on_host != 0);
    if (rc) return rc;
    const int64_t n = (int64_t)s->grid.nvox[0] * s->grid.nvox[1] * s->grid.nvox[2] * s->grid.stride;
    *n_floats = n;
    if (out && capacity >= n) HIP_TRY(hipMemcpy(out, s->grid_table.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return GNXR_OK;
}

int gnxr_eval_libm(int32_t fn, const float *x, const float *x2, int64_t n, float *out) {
    if (!x || !out || n < 0 || fn < 0 || fn > 8 || (fn >= 7 && !x2)) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return GNXR_OK;
    DevBuf<float> dx, dx2, dout;
    if ((rc = dx.upload(x, (size_t)n)) || (rc = dout.alloc((size_t)n))) return rc;
    if (x2 && (rc = dx2.upload(x2, (size_t)n))) return rc;
    hipLaunchKernelGGL(k_libm_probe, dim3(grid_for(n)), dim3(kBlock), 0, 0, (int)fn, (const float *)dx.p, (const float *)(x2 ? dx2.p : nullptr), (long long)n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return GNXR_OK;
}

int gnxr_eval_libm_f64(int32_t fn, const float *x, int64_t n, double *out) {
    if (!x || !out || n < 0 || fn < 0 || fn > 3) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    if (n == 0) return GNXR_OK;
    DevBuf<float> dx;
    DevBuf<double> dout;
    if ((rc = dx.upload(x, (size_t)n)) || (rc = dout.alloc((size_t)n))) return rc;
    hipLaunchKernelGGL(k_libm_probe_f64, dim3(grid_for(n)), dim3(kBlock), 0, 0, (int)fn, (const float *)dx.p, (long long)n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return GNXR_OK;
}

int gnxr_framebuffer_update(float *running_mean_rgba, const float *frame_rgba, int32_t width, int32_t height, int32_t frame_count, uint8_t *rgba8_out) {
    if (!running_mean_rgba || !frame_rgba || !rgba8_out || width <= 0 || height <= 0 || frame_count <= 0) { set_error("bad argument"); return GNXR_ERR_INVALID; }
    int rc = ensure_device();
    if (rc) return rc;
    size_t nv = (size_t)width * height * 4;
    DevBuf<float> dm, df;
    DevBuf<unsigned char> du;
    if ((rc = dm.upload(running_mean_rgba, nv)) || (rc = df.upload(frame_rgba, nv)) || (rc = du.alloc(nv))) return rc;
    hipLaunchKernelGGL(k_framebuffer_update, dim3(grid_for((long long)nv)), dim3(kBlock), 0, 0, dm.p, (const float *)df.p, (long long)nv, frame_count, du.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(running_mean_rgba, dm.p, nv * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rgba8_out, du.p, nv, hipMemcpyDeviceToHost));
    return GNXR_OK;
}

}  // extern "C"
