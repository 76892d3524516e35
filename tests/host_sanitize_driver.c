/* TEST INFRASTRUCTURE (tests/test_host_sanitizers.py): drives the product's C host (csrc/sgm_host.c) through every entry point
 * family with the stub device layer (tests/stub_device.c: "device" memory is host memory, copies are real memcpy, kernels do
 * nothing) under AddressSanitizer / UBSan -- buffer sizes, staging copies, table uploads, tile hand-over offsets and the
 * lifetime of everything the host allocates.  Results are not checked (there are none). */
#include "../include/sgm_mi355x.h"
#include "../include/sgm_tiles.h"

#include <pthread.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "host_sanitize_driver: %s failed (line %d)\n", #x, __LINE__); return 1; } } while (0)

static SGMOption options(int d, int dmin)
{
    SGMOption o;
    memset(&o, 0, sizeof o);
    o.num_paths = 8; o.min_disparity = (uint16_t)dmin; o.max_disparity = (uint16_t)(dmin + d);
    o.is_check_lr = true; o.lrcheck_thres = 1.0f; o.is_check_unique = true; o.uniqueness_ratio = 0.99;
    o.is_remove_speckles = true; o.min_speckle_area = 20; o.p1 = 10; o.p2_init = 150;
    return o;
}

/* the multi-GPU host (csrc/sgm_tiles.c) with ranks as threads: the stub device's toy compute touches every plane cell, hand-over
 * buffer and map the pipeline allocates, so the sanitizer sees the offsets of the real data flow */
void stub_toy_compute(int on);
typedef struct { int rank, world, batch, lead; sgm_tiles_local* group; const uint8_t* img; float* ring; int ring_frames; int ok; } rank_arg;
static void* rank_main(void* p)
{
    rank_arg* a = (rank_arg*)p;
    const int W = 12, H = 13;
    SGMOption o = options(16, 0);
    sgm_tiles_transport tr;
    memset(&tr, 0, sizeof tr);
    if (a->world > 1 && !sgm_tiles_local_transport(a->group, a->rank, &tr)) return NULL;
    sgm_tiles* t = sgm_tiles_create(0, a->rank, a->world, (uint16_t)W, (uint16_t)H, &o, a->batch, a->lead, 1, 2, a->world > 1 ? &tr : NULL);
    if (!t) return NULL;
    sgm_tiles_result_ring(t, a->ring, a->ring_frames);
    int ok = 1;
    for (int k = 0; ok && k < 2 * a->world + 3; ++k) ok = sgm_tiles_submit(t, a->img, a->img, NULL);
    ok = ok && sgm_tiles_finish(t);
    /* a second stream on the same pipeline */
    for (int k = 0; ok && k < 2; ++k) ok = sgm_tiles_submit(t, a->img, a->img, NULL);
    ok = ok && sgm_tiles_finish(t);
    sgm_tiles_destroy(t);
    if (tr.destroy) tr.destroy(tr.ctx);
    a->ok = ok;
    return NULL;
}
static int tiles_pipeline(const uint8_t* img)
{
    stub_toy_compute(1);
    for (int world = 1; world <= 4; ++world)
        for (int batch = 1; batch <= 2; ++batch) {
            sgm_tiles_local* g = world > 1 ? sgm_tiles_local_group(world, 0) : NULL;
            rank_arg a[4];
            pthread_t th[4];
            const int ring_frames = 3;
            for (int r = 0; r < world; ++r) {
                a[r] = (rank_arg){r, world, batch, batch == 1 ? 2 : 0, g, img, NULL, ring_frames, 0};
                a[r].ring = (float*)calloc((size_t)ring_frames * batch * 12 * 13, sizeof(float));
                CHECK(a[r].ring && pthread_create(&th[r], NULL, rank_main, &a[r]) == 0);
            }
            for (int r = 0; r < world; ++r) {
                pthread_join(th[r], NULL);
                free(a[r].ring);
                CHECK(a[r].ok);
            }
            sgm_tiles_local_destroy(g);
        }
    stub_toy_compute(0);
    return 0;
}

int main(void)
{
    const int W = 48, H = 20;
    uint8_t* img = (uint8_t*)calloc((size_t)4 * 6 * W * H, 1);
    float* out = (float*)calloc((size_t)4 * W * H, sizeof(float));
    uint8_t* buf = (uint8_t*)calloc((size_t)4 * 3 * W * 64, 1);
    void* stage = malloc((size_t)W * H * 64 * 2);
    CHECK(img && out && buf && stage);

    /* the reference's global entry points */
    SGMOption o = options(16, 0);
    CHECK(SGM_Initialize((uint16_t)W, (uint16_t)H, &o));
    CHECK(SGM_Match(img, img, out));
    CHECK(SGM_Match(img, img, out));                                   /* no Reset: Q14 accumulation path */
    CHECK(SGM_Reset((uint16_t)W, (uint16_t)H, &o));
    CHECK(sgm_compute(img, img, (uint16_t)W, (uint16_t)H, &o, out));
    CHECK(!SGM_Initialize(0, (uint16_t)H, &o));
    SGM_KeepStages(1);
    CHECK(SGM_Reset((uint16_t)W, (uint16_t)H, &o) && SGM_Match(img, img, out));
    for (int which = 0; which <= 8; ++which) (void)SGM_ReadStage(which, stage, (size_t)W * H * 64 * 2);
    for (int which = 10; which < 18; ++which) (void)SGM_ReadStage(which, stage, (size_t)W * H * 64 * 2);
    SGM_Shutdown();

    /* instances: shapes that change, batches, the asynchronous host-pointer entry, a test-platform frame */
    sgm_instance* s = sgm_create(0);
    CHECK(s);
    SGMOption big = options(40, 3), wide = options(300, 0);
    CHECK(sgm_initialize(s, (uint16_t)W, (uint16_t)H, &o) && sgm_match(s, img, img, out));
    CHECK(sgm_reset(s, 31, 17, &big) && sgm_match(s, img, img, out));          /* smaller frame, padded range */
    CHECK(sgm_reset(s, (uint16_t)W, (uint16_t)H, &wide) && sgm_match(s, img, img, out));   /* D > 256: separate kernels, S allocated */
    CHECK(sgm_set_overlap_post(s, 1));
    CHECK(sgm_reset(s, (uint16_t)W, (uint16_t)H, &o) && sgm_match_async(s, img, img, out) && sgm_match_async(s, img, img, out) && sgm_match_wait(s));
    CHECK(sgm_match_planes(s, img, 1000.f, 100.f, 0.f, out));
    CHECK(sgm_match_planes_async(s, img, 1000.f, 100.f, 0.f, out) && sgm_match_wait(s));
    void* pinned = sgm_host_alloc(s, 4096);
    CHECK(pinned);
    sgm_host_free(s, pinned);
    sgm_destroy(s);

    s = sgm_create(0);
    CHECK(s && sgm_set_batch(s, 3) && sgm_initialize(s, (uint16_t)W, (uint16_t)H, &o));
    CHECK(sgm_match(s, img, img, out) && sgm_match_planes(s, img, 1000.f, 100.f, 0.f, out));
    sgm_select_frame(s, 2);
    (void)sgm_read_stage(s, 8, stage, (size_t)W * H * 4);
    sgm_destroy(s);

    /* extensions */
    s = sgm_create(0);
    CHECK(s && sgm_set_census_window(s, 9, 7) && !sgm_set_census_window(s, 9, 9));
    sgm_set_reference_view(s, 1);
    sgm_set_honor_num_paths(s, 1);
    SGMOption four = options(24, 0);
    four.num_paths = 4;
    CHECK(sgm_initialize(s, (uint16_t)W, (uint16_t)H, &four) && sgm_match(s, img, img, out));
    sgm_destroy(s);

    /* row tiles: every position of a tile in the frame, single frames and batches, W < H */
    const int tiles[4][2] = {{0, 7}, {7, 13}, {13, 20}, {5, 6}};
    for (int b = 1; b <= 2; ++b)
        for (int t = 0; t < 4; ++t)
            for (int tall = 0; tall < 2; ++tall) {
                const int w = tall ? 12 : W, h = H;
                s = sgm_create(0);
                CHECK(s && sgm_set_batch(s, b) && sgm_set_rows(s, tiles[t][0], tiles[t][1]) && sgm_initialize(s, (uint16_t)w, (uint16_t)h, &o));
                CHECK(sgm_tile_boundary_bytes(s) <= (size_t)4 * 3 * W * 64);
                CHECK(sgm_tile_begin(s, img, img));
                for (int fwd = 1; fwd >= 0; --fwd) {
                    const bool first = fwd ? tiles[t][0] == 0 : tiles[t][1] == h, last = fwd ? tiles[t][1] == h : tiles[t][0] == 0;
                    CHECK(sgm_tile_import_boundary(s, fwd, buf) == !first);
                    CHECK(sgm_tile_sweep(s, fwd));
                    if (!last) CHECK(sgm_tile_export_boundary(s, fwd, buf));
                }
                CHECK(sgm_tile_finish(s, out) && sgm_tile_post(s, out) && sgm_synchronize(s));
                CHECK(!sgm_match(s, img, img, out));                           /* whole-frame call refused in tile mode */
                CHECK(sgm_set_rows(s, 0, 0) && sgm_reset(s, (uint16_t)w, (uint16_t)h, &o) && sgm_match(s, img, img, out));
                sgm_destroy(s);
            }

    CHECK(tiles_pipeline(img) == 0);

    free(img); free(out); free(buf); free(stage);
    printf("host_sanitize_driver ok\n");
    return 0;
}
