/*
 * tiles_rank.c -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_tiles_processes.py): ONE rank of the C tile pipeline
 * (include/sgm_tiles.h) as a process of its own -- plain C, no Python, its own HIP context and its own copy of the library --
 * connected to its peers by tests/sock_transport.c.  The ranks share the box's one GPU (RCCL would refuse that), everything
 * else is what a multi-GPU run does: every rank submits the same frames, the owner of a frame ends up with its final map.
 *
 *   tiles_rank DIR RANK WORLD W H D BATCH LEAD STEPS SEED
 *
 * Frames are the synthetic pairs of SURVEY.md 8(d): step k holds frames seed + k * BATCH .. + BATCH - 1.  The rank writes the
 * maps of the steps it owns to DIR/step<k>.f32 ([BATCH][H][W] float32); the test compares them with the oracle.
 */
#define _POSIX_C_SOURCE 200809L
#include "../include/sgm_tiles.h"
#include "../soc_project_stereo_matching_amd/csrc/sgm_device.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

bool sock_transport_create(const char* dir, int rank, int world, int device, sgm_tiles_transport* out);

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "tiles_rank %d: %s failed (line %d)\n", rank, #x, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc != 11) { fprintf(stderr, "usage: tiles_rank DIR RANK WORLD W H D BATCH LEAD STEPS SEED\n"); return 2; }
    const char* dir = argv[1];
    const int rank = atoi(argv[2]), world = atoi(argv[3]), W = atoi(argv[4]), H = atoi(argv[5]), D = atoi(argv[6]);
    const int B = atoi(argv[7]), lead = atoi(argv[8]), steps = atoi(argv[9]);
    const unsigned seed = (unsigned)strtoul(argv[10], NULL, 0);
    SGMOption opt;
    memset(&opt, 0, sizeof opt);                       /* main.c:48-65 with max_disparity = D */
    opt.num_paths = 8; opt.min_disparity = 0; opt.max_disparity = (uint16_t)D;
    opt.is_check_lr = true; opt.lrcheck_thres = 1.0f; opt.is_check_unique = true; opt.uniqueness_ratio = 0.99;
    opt.is_remove_speckles = true; opt.min_speckle_area = 50; opt.p1 = 10; opt.p2_init = 150;
    const size_t px = (size_t)W * H;

    sgm_tiles_transport tr;
    memset(&tr, 0, sizeof tr);
    if (world > 1) CHECK(sock_transport_create(dir, rank, world, 0, &tr));
    sgm_tiles* t = sgm_tiles_create(0, rank, world, (uint16_t)W, (uint16_t)H, &opt, B, lead, 1, 4, world > 1 ? &tr : NULL);
    CHECK(t != NULL);

    /* the whole stream resident on the device (every rank holds the whole images) and a ring for every step this rank owns */
    void* stream = NULL;
    CHECK(sgmd_stream_create(0, &stream) == 0);
    uint8_t* host = (uint8_t*)malloc(px);
    uint8_t* hostr = (uint8_t*)malloc(px);
    void** dl = (void**)calloc((size_t)steps, sizeof *dl);
    void** dr = (void**)calloc((size_t)steps, sizeof *dr);
    CHECK(host && hostr && dl && dr);
    for (int k = 0; k < steps; ++k) {
        CHECK(sgmd_alloc(0, &dl[k], px * B) == 0 && sgmd_alloc(0, &dr[k], px * B) == 0);
        for (int j = 0; j < B; ++j) {
            SGM_SynthPair(W, H, D, seed + (unsigned)(k * B + j), host, hostr);
            CHECK(sgmd_h2d_async(0, stream, (char*)dl[k] + px * j, host, px) == 0 && sgmd_h2d_async(0, stream, (char*)dr[k] + px * j, hostr, px) == 0);
            CHECK(sgmd_stream_sync(0, stream) == 0);
        }
    }
    const int ring_frames = (steps + world - 1) / world;
    void* ring = NULL;
    CHECK(sgmd_alloc(0, &ring, (size_t)ring_frames * B * px * sizeof(float)) == 0);
    sgm_tiles_result_ring(t, (float*)ring, ring_frames);

    for (int k = 0; k < steps; ++k) CHECK(sgm_tiles_submit(t, (const uint8_t*)dl[k], (const uint8_t*)dr[k], NULL));
    CHECK(sgm_tiles_finish(t));

    float* out = (float*)malloc(B * px * sizeof(float));
    CHECK(out != NULL);
    for (int k = rank; k < steps; k += world) {
        CHECK(sgmd_d2h_async(0, stream, out, (char*)ring + (size_t)((k / world) % ring_frames) * B * px * sizeof(float), B * px * sizeof(float)) == 0);
        CHECK(sgmd_stream_sync(0, stream) == 0);
        char path[400];
        snprintf(path, sizeof path, "%s/step%d.f32", dir, k);
        FILE* f = fopen(path, "wb");
        CHECK(f && fwrite(out, sizeof(float), B * px, f) == B * px);
        fclose(f);
    }
    sgm_tiles_destroy(t);
    if (tr.destroy) tr.destroy(tr.ctx);
    for (int k = 0; k < steps; ++k) { sgmd_free(0, dl[k]); sgmd_free(0, dr[k]); }
    sgmd_free(0, ring);
    sgmd_stream_destroy(0, stream);
    free(host); free(hostr); free(dl); free(dr); free(out);
    printf("tiles_rank %d of %d: %d steps ok\n", rank, world, steps);
    return 0;
}
