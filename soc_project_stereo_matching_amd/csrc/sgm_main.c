/* sgm_main.c -- command-line driver with the flow of the reference's desktop driver
 * (SemiGlobalMatching/SemiGlobalMatching/main.c:16-126): load a stereo pair as 8-bit grey, fill SGMOption with
 * the same defaults (main.c:48-65), SGM_Initialize + SGM_Match, normalise the valid disparities to 8 bit
 * exactly as main.c:92-117 and write the map.  Unlike the reference (paths hard-coded, argv ignored) the files
 * and options come from the command line.
 *
 *   sgm_main LEFT RIGHT OUT.png|OUT.pgm [--min-disparity N] [--max-disparity N] [--p1 N] [--p2 N]
 *            [--no-lr] [--lr-thres F] [--no-unique] [--unique-ratio F] [--no-speckle] [--speckle-area N]
 *            [--raw OUT.f32] [--repeat N] [--device N]
 *            [--paths 4|8] [--census WxH] [--right-reference]      extensions of the boundary: 4-path
 *            aggregation (the reference stores num_paths and ignores it), census windows other than 5x5, the right image
 *            as reference view; the defaults are the reference's behaviour
 *   sgm_main --convert IN OUT.png        (image I/O only, no GPU: used by the CPU tests)
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/sgm_mi355x.h"
#include "sgm_image_io.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int ends_with(const char* s, const char* suf)
{
    const size_t n = strlen(s), m = strlen(suf);
    return n >= m && !strcmp(s + n - m, suf);
}

int main(int argc, char** argv)
{
    if (argc == 4 && !strcmp(argv[1], "--convert")) {
        int w, h;
        uint8_t* g = sgm_load_gray(argv[2], &w, &h);
        if (!g) return 1;
        const int rc = ends_with(argv[3], ".pgm") ? sgm_write_pgm(argv[3], g, w, h) : sgm_write_png_gray(argv[3], g, w, h);
        free(g);
        return rc ? 1 : 0;
    }
    if (argc < 4) {
        fprintf(stderr, "usage: %s LEFT RIGHT OUT.png [options]   (see the header of sgm_main.c)\n", argv[0]);
        return 2;
    }
    SGMOption opt;
    memset(&opt, 0, sizeof opt);                       /* main.c:48-65 */
    opt.num_paths = 8;
    opt.min_disparity = 0;
    opt.max_disparity = 64;
    opt.is_check_lr = true;
    opt.lrcheck_thres = 1.0f;
    opt.is_check_unique = true;
    opt.uniqueness_ratio = 0.99;
    opt.is_remove_speckles = true;
    opt.min_speckle_area = 50;
    opt.p1 = 10;
    opt.p2_init = 150;
    const char* raw_path = NULL;
    int repeat = 1, device = -1, census_w = 0, census_h = 0, right_ref = 0;
    for (int i = 4; i < argc; ++i) {
        const char* a = argv[i];
        const char* v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--no-lr")) opt.is_check_lr = false;
        else if (!strcmp(a, "--no-unique")) opt.is_check_unique = false;
        else if (!strcmp(a, "--no-speckle")) opt.is_remove_speckles = false;
        else if (v && !strcmp(a, "--min-disparity")) { opt.min_disparity = (uint16_t)atoi(v); ++i; }
        else if (v && !strcmp(a, "--max-disparity")) { opt.max_disparity = (uint16_t)atoi(v); ++i; }
        else if (v && !strcmp(a, "--p1")) { opt.p1 = (int16_t)atoi(v); ++i; }
        else if (v && !strcmp(a, "--p2")) { opt.p2_init = (int16_t)atoi(v); ++i; }
        else if (v && !strcmp(a, "--lr-thres")) { opt.lrcheck_thres = (float)atof(v); ++i; }
        else if (v && !strcmp(a, "--unique-ratio")) { opt.uniqueness_ratio = (float)atof(v); ++i; }
        else if (v && !strcmp(a, "--speckle-area")) { opt.min_speckle_area = (uint16_t)atoi(v); ++i; }
        else if (v && !strcmp(a, "--raw")) { raw_path = v; ++i; }
        else if (v && !strcmp(a, "--repeat")) { repeat = atoi(v); ++i; }
        else if (v && !strcmp(a, "--device")) { device = atoi(v); ++i; }
        else if (v && !strcmp(a, "--paths")) { opt.num_paths = (uint8_t)atoi(v); SGM_SetHonorNumPaths(1); ++i; }
        else if (v && !strcmp(a, "--census")) {
            if (sscanf(v, "%dx%d", &census_w, &census_h) != 2) { fprintf(stderr, "--census wants WxH, e.g. 7x7\n"); return 2; }
            ++i;
        }
        else if (!strcmp(a, "--right-reference")) right_ref = 1;
        else { fprintf(stderr, "unknown option %s\n", a); return 2; }
    }

    int w1, h1, w2, h2;
    uint8_t* left = sgm_load_gray(argv[1], &w1, &h1);
    uint8_t* right = sgm_load_gray(argv[2], &w2, &h2);
    if (!left || !right) { printf("Failed to load images\n"); return -1; }
    if (w1 != w2 || h1 != h2) { printf("Images must have same dimensions\n"); return -1; }
    if (w1 > 65535 || h1 > 65535) { printf("Image too large\n"); return -1; }
    printf("w = %d, h = %d, d = [%d,%d]\n", w1, h1, opt.min_disparity, opt.max_disparity);

    if (device >= 0) SGM_SetDevice(device);
    if (census_w && !SGM_SetCensusWindow(census_w, census_h)) { printf("unsupported census window %dx%d\n", census_w, census_h); return -2; }
    if (right_ref) SGM_SetReferenceView(1);
    if (!SGM_Initialize((uint16_t)w1, (uint16_t)h1, &opt)) { printf("SGM initialization failed\n"); return -2; }
    float* disp = (float*)malloc(sizeof(float) * (size_t)w1 * h1);
    double best = 1e30;
    for (int r = 0; r < repeat; ++r) {
        const double t0 = now_ms();
        if (r > 0 && !SGM_Reset((uint16_t)w1, (uint16_t)h1, &opt)) { printf("SGM reset failed\n"); return -2; }
        if (!SGM_Match(left, right, disp)) { printf("SGM matching failed\n"); return -2; }
        const double t = now_ms() - t0;
        if (t < best) best = t;
    }
    printf("SGM_Match (host images in, host disparity out): %.3f ms\n", best);

    /* main.c:92-117 */
    const size_t px = (size_t)w1 * h1;
    float lo = (float)w1, hi = -(float)w1;
    size_t valid = 0;
    for (size_t i = 0; i < px; ++i)
        if (disp[i] != INFINITY) {
            if (disp[i] < lo) lo = disp[i];
            if (disp[i] > hi) hi = disp[i];
            ++valid;
        }
    const float range = (hi - lo) != 0.0f ? (hi - lo) : 1.0f;
    uint8_t* u8 = (uint8_t*)malloc(px);
    for (size_t i = 0; i < px; ++i) {
        if (disp[i] == INFINITY) { u8[i] = 0; continue; }
        float v = (disp[i] - lo) / range * 255.0f;
        if (v < 0) v = 0;
        if (v > 255) v = 255;
        u8[i] = (unsigned char)v;
    }
    printf("valid %zu of %zu, disparity range [%g, %g]\n", valid, px, lo, hi);
    int rc = ends_with(argv[3], ".pgm") ? sgm_write_pgm(argv[3], u8, w1, h1) : sgm_write_png_gray(argv[3], u8, w1, h1);
    if (raw_path) {
        FILE* f = fopen(raw_path, "wb");
        if (!f || fwrite(disp, sizeof(float), px, f) != px) rc = -1;
        if (f) fclose(f);
    }
    SGM_Shutdown();
    free(u8); free(disp); free(left); free(right);
    return rc ? 1 : 0;
}
