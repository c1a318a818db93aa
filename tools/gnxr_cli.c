/* gnxr_cli -- the caller side of the boundary in plain C: what ui/RenderThread.cpp:46-187 does in the reference, against the C ABI
 * of libgnxr.so (include/gnxr.h) instead of the pbr:: classes.
 *
 *   scene      materials of RenderThread.cpp:79-103, AddModel (a .3d mesh, ModelList.cpp:47-69; optional), AddCornell
 *              (:71-118), AddAreaLight (:120-147), AddSpotLight / AddDistLight (:149-161; --spot / --dist), AddSkyLight (:163-170;
 *              --sky), camera of RenderThread.cpp:60-68
 *   integrator WhittedIntegrator(5, ...) as RenderThread.cpp:163 instantiates it, or --integrator path|volpath|direct
 *   loop       `while (renderFlag) { integrator->Render(...); emit PaintBuffer(getUCbuffer()) }` (:168-186) for --frames
 *              iterations: every Render() result is folded into the running mean and tone-mapped to RGBA8 as
 *              FrameBuffer::update_f_u_c does (ui/FrameBuffer.h:127-149), and the last RGBA8 plane is written as a PNG
 *              (FrameBuffer::saveToFile, ui/FrameBuffer.cpp:6-9)
 *
 * There is no CPU fallback: without a HIP device gnxr_init fails and the program exits with status 3. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gnxr.h"

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [--width W] [--height H] [--spp N] [--frames F] [--depth D] [--integrator whitted|path|volpath|direct]\n"
            "          [--model mesh.3d] [--model-material matte|plastic|metal|glass] [--sky] [--spot] [--dist] [--device I] --out image.png\n"
            "renders the reference's default scene (ui/RenderThread.cpp) through libgnxr.so\n",
            argv0);
}
#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ < 0) {                                                                 \
            fprintf(stderr, "gnxr_cli: %s failed (%d): %s\n", #call, rc_, gnxr_last_error()); \
            return rc_ == GNXR_ERR_NO_DEVICE ? 3 : 1;                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv) {
    int width = 500, height = 500, spp = 32, frames = 1, depth = 5, device = 0, sky = 0, spot = 0, dist = 0;   /* WIDTH / HEIGHT, HaltonSampler(32), Whitted(5) */
    const char *integrator = "whitted", *model = NULL, *model_material = "matte", *out = NULL;
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
#define ARG(name) (!strcmp(a, name) && i + 1 < argc)
        if (ARG("--width")) width = atoi(argv[++i]);
        else if (ARG("--height")) height = atoi(argv[++i]);
        else if (ARG("--spp")) spp = atoi(argv[++i]);
        else if (ARG("--frames")) frames = atoi(argv[++i]);
        else if (ARG("--depth")) depth = atoi(argv[++i]);
        else if (ARG("--device")) device = atoi(argv[++i]);
        else if (ARG("--integrator")) integrator = argv[++i];
        else if (ARG("--model")) model = argv[++i];
        else if (ARG("--model-material")) model_material = argv[++i];
        else if (ARG("--out")) out = argv[++i];
        else if (!strcmp(a, "--sky")) sky = 1;
        else if (!strcmp(a, "--spot")) spot = 1;   /* AddSpotLight, commented out at RenderThread.cpp:138 */
        else if (!strcmp(a, "--dist")) dist = 1;   /* AddDistLight, RenderThread.cpp:141 */
        else { usage(argv[0]); return !strcmp(a, "--help") || !strcmp(a, "-h") ? 0 : 2; }
#undef ARG
    }
    if (!out || width <= 0 || height <= 0 || spp <= 0 || frames <= 0) { usage(argv[0]); return 2; }
    gnxr_render_params p;
    memset(&p, 0, sizeof(p));
    p.width = width; p.height = height; p.spp = spp; p.spp_begin = 0; p.spp_end = spp; p.max_depth = depth;
    p.rr_threshold = 1.f; p.light_strategy = GNXR_LIGHTS_SPATIAL; p.shard_count = 1; p.shard_rows = 1;
    if (!strcmp(integrator, "whitted")) p.integrator = GNXR_INTEGRATOR_WHITTED;
    else if (!strcmp(integrator, "path")) p.integrator = GNXR_INTEGRATOR_PATH;
    else if (!strcmp(integrator, "volpath")) p.integrator = GNXR_INTEGRATOR_VOLPATH;
    else if (!strcmp(integrator, "direct")) { p.integrator = GNXR_INTEGRATOR_DIRECT; p.direct_strategy = GNXR_DIRECT_SAMPLE_ALL; }
    else { usage(argv[0]); return 2; }

    CHECK(gnxr_init(device));

    /* ---- scene authoring, RenderThread.cpp:70-151 */
    gnxr_builder *b = NULL;
    CHECK(gnxr_builder_create(&b));
    const float white[3] = {0.91f, 0.91f, 0.91f}, dragon[3] = {0.2f, 0.8f, 0.2f}, red[3] = {0.9f, 0.1f, 0.17f}, blue[3] = {0.14f, 0.21f, 0.87f};
    int m_white = gnxr_builder_matte(b, white, 60.f), m_red = gnxr_builder_matte(b, red, 60.f), m_blue = gnxr_builder_matte(b, blue, 60.f);
    int m_dragon;
    if (!strcmp(model_material, "plastic")) m_dragon = gnxr_builder_purple_plastic(b);
    else if (!strcmp(model_material, "metal")) m_dragon = gnxr_builder_yellow_metal(b);
    else if (!strcmp(model_material, "glass")) m_dragon = gnxr_builder_white_glass(b);
    else m_dragon = gnxr_builder_matte(b, dragon, 60.f);
    CHECK(m_white); CHECK(m_red); CHECK(m_blue); CHECK(m_dragon);
    if (model) CHECK(gnxr_builder_add_model_3d(b, model, m_dragon));
    CHECK(gnxr_builder_add_cornell(b, m_red, m_blue, m_white));
    CHECK(gnxr_builder_add_area_light(b, m_dragon));   /* the reference hands the light quad the dragon material (:133) */
    if (spot) CHECK(gnxr_builder_add_spot_light(b));
    if (dist) CHECK(gnxr_builder_add_dist_light(b));
    if (sky) CHECK(gnxr_builder_add_sky_light(b));
    gnxr_scene_desc desc;
    CHECK(gnxr_builder_desc(b, &desc));
    gnxr_scene *scene = NULL;
    CHECK(gnxr_scene_create(&desc, &scene));

    /* ---- the render loop, RenderThread.cpp:168-186 + FrameBuffer::update_f_u_c */
    const size_t npx = (size_t)width * height;
    float *frame = (float *)malloc(npx * 4 * sizeof(float)), *mean = (float *)calloc(npx * 4, sizeof(float));
    uint8_t *rgba8 = (uint8_t *)malloc(npx * 4);
    if (!frame || !mean || !rgba8) { fprintf(stderr, "gnxr_cli: out of memory\n"); return 1; }
    for (int f = 1; f <= frames; ++f) {
        gnxr_stats st;
        CHECK(gnxr_render(scene, &p, frame, &st));
        CHECK(gnxr_framebuffer_update(mean, frame, width, height, f, rgba8));
        fprintf(stderr, "frame %d: %.3f s, %.1f Mrays/s (%llu closest-hit + %llu any-hit rays)\n", f, st.seconds_render,
                (double)(st.rays_closest + st.rays_any) / st.seconds_render / 1e6, (unsigned long long)st.rays_closest, (unsigned long long)st.rays_any);
    }
    CHECK(gnxr_framebuffer_save_png(out, rgba8, width, height));
    free(frame); free(mean); free(rgba8);
    gnxr_scene_destroy(scene);
    gnxr_builder_destroy(b);
    gnxr_shutdown();
    return 0;
}
