/*
 * tiles_rank.c -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_tiles_processes.py): ONE rank of the C tile pipeline
 * (include/sgm_tiles.h) as a process of its own -- plain C, no Python, its own HIP context and its own copy of the library --
 * connected to its peers by tests/sock_transport.c.  The ranks share the box's one GPU (RCCL would refuse that), everything
 * else is what a multi-GPU run does: every rank submits the same frames, the owner of a frame ends up with its final map.
 *
 *   tiles_rank DIR RANK WORLD W H D BATCH LEAD STEPS SEED
 *
 * SGM_TILES_TRANSPORT=rccl (a node with >= WORLD GPUs): rank r runs on GPU r and the ranks are connected by the library's RCCL
 * transport instead (sgm_tiles_rccl_*; rank 0 leaves the 128-byte id in DIR/rccl_id) -- the run that has not happened yet on
 * this pool's one-GPU boxes.
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
#include <time.h>

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
    const char* kind = getenv("SGM_TILES_TRANSPORT");
    const bool rccl = kind && !strcmp(kind, "rccl");
    const int dev = rccl ? rank : 0;                   /* RCCL: one GPU per rank; sockets: the ranks share GPU 0 */
    if (world > 1 && rccl) {
        char id[SGM_TILES_ID_BYTES], path[400], tmp[420];
        snprintf(path, sizeof path, "%s/rccl_id", dir);
        if (rank == 0) {
            CHECK(sgm_tiles_rccl_unique_id(id));
            snprintf(tmp, sizeof tmp, "%s.tmp", path);
            FILE* f = fopen(tmp, "wb");
            CHECK(f && fwrite(id, 1, sizeof id, f) == sizeof id);
            fclose(f);
            CHECK(rename(tmp, path) == 0);
        } else {
            FILE* f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(path, "rb")); ++tries) nanosleep(&(struct timespec){0, 100000000}, NULL);
            CHECK(f && fread(id, 1, sizeof id, f) == sizeof id);
            fclose(f);
        }
        CHECK(sgm_tiles_rccl_transport(id, rank, world, dev, &tr));
    } else if (world > 1)
        CHECK(sock_transport_create(dir, rank, world, 0, &tr));
    sgm_tiles* t = sgm_tiles_create(dev, rank, world, (uint16_t)W, (uint16_t)H, &opt, B, lead, 1, 4, world > 1 ? &tr : NULL);
    CHECK(t != NULL);

    /* the whole stream resident on the device (every rank holds the whole images) and a ring for every step this rank owns */
    void* stream = NULL;
    CHECK(sgmd_stream_create(dev, &stream) == 0);
    uint8_t* host = (uint8_t*)malloc(px);
    uint8_t* hostr = (uint8_t*)malloc(px);
    void** dl = (void**)calloc((size_t)steps, sizeof *dl);
    void** dr = (void**)calloc((size_t)steps, sizeof *dr);
    CHECK(host && hostr && dl && dr);
    for (int k = 0; k < steps; ++k) {
        CHECK(sgmd_alloc(dev, &dl[k], px * B) == 0 && sgmd_alloc(dev, &dr[k], px * B) == 0);
        for (int j = 0; j < B; ++j) {
            SGM_SynthPair(W, H, D, seed + (unsigned)(k * B + j), host, hostr);
            CHECK(sgmd_h2d_async(dev, stream, (char*)dl[k] + px * j, host, px) == 0 && sgmd_h2d_async(dev, stream, (char*)dr[k] + px * j, hostr, px) == 0);
            CHECK(sgmd_stream_sync(dev, stream) == 0);
        }
    }
    const int ring_frames = (steps + world - 1) / world;
    void* ring = NULL;
    CHECK(sgmd_alloc(dev, &ring, (size_t)ring_frames * B * px * sizeof(float)) == 0);
    sgm_tiles_result_ring(t, (float*)ring, ring_frames);

    for (int k = 0; k < steps; ++k) CHECK(sgm_tiles_submit(t, (const uint8_t*)dl[k], (const uint8_t*)dr[k], NULL));
    CHECK(sgm_tiles_finish(t));

    float* out = (float*)malloc(B * px * sizeof(float));
    CHECK(out != NULL);
    for (int k = rank; k < steps; k += world) {
        CHECK(sgmd_d2h_async(dev, stream, out, (char*)ring + (size_t)((k / world) % ring_frames) * B * px * sizeof(float), B * px * sizeof(float)) == 0);
        CHECK(sgmd_stream_sync(dev, stream) == 0);
        char path[400];
        snprintf(path, sizeof path, "%s/step%d.f32", dir, k);
        FILE* f = fopen(path, "wb");
        CHECK(f && fwrite(out, sizeof(float), B * px, f) == B * px);
        fclose(f);
    }
    sgm_tiles_destroy(t);
    if (tr.destroy) tr.destroy(tr.ctx);
    for (int k = 0; k < steps; ++k) { sgmd_free(dev, dl[k]); sgmd_free(dev, dr[k]); }
    sgmd_free(dev, ring);
    sgmd_stream_destroy(dev, stream);
    free(host); free(hostr); free(dl); free(dr); free(out);
    printf("tiles_rank %d of %d: %d steps ok\n", rank, world, steps);
    return 0;
}
