/* sgm_stream.c -- a C caller with a stream of frames: what the library's host-pointer boundary delivers without any Python in
 * the process.  The call site this stands for is the reference's intended one, a C loop around the matcher
 * (ZedBoard/Vitis/lwip_tcp_perf_client/src/stereo_matching.c:34-40; the contract of SemiGlobalMatching.c:77-78,122: borrowed host
 * images in, host floats out).
 *
 *   sgm_stream [--width W] [--height H] [--disparities D] [--batch B] [--instances N] [--seconds S] [--frames F] [--seed X]
 *              [--pageable] [--blocking] [--numa-node K]
 *
 *   default      N instances, one host thread each, batches of B frames through sgm_reset + sgm_match_async + sgm_match_wait on
 *                page-locked buffers (sgm_host_alloc) for S seconds: the pipelined throughput path (bench.py's headline, in C)
 *   --blocking   one thread, one frame per call through sgm_compute (SGM_Reset + SGM_Match) on malloc'd buffers: the reference
 *                contract as it stands
 *   --pageable   malloc'd caller buffers instead of page-locked ones (staged by the library)
 *   --numa-node K  run (and allocate) on the CPUs of NUMA node K -- the node the GPU hangs off (/sys/bus/pci/devices/<bdf>/numa_node):
 *                the host threads spin in stream synchronisation and feed 11 GB/s over PCIe; bench.py pins itself the same way
 *
 * Frames are the synthetic pairs of SURVEY.md 8(d) (SGM_SynthPair, seed + frame index), F distinct ones cycled.  Prints one JSON
 * line: frames, seconds, fps, Mdisp/s and an FNV-1a hash of the disparity map of frame 0 (tests/test_gpu_stream_c.py compares
 * it with the oracle's map hashed the same way).
 */
#define _GNU_SOURCE
#include "../../include/sgm_mi355x.h"

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

/* restrict the process to the CPUs of a NUMA node (best effort: false if sysfs does not list them) */
static int pin_to_node(int node)
{
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f || !fgets(buf, sizeof buf, f)) { if (f) fclose(f); return 0; }
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(NULL, ",\n")) {
        int a = 0, b = 0;
        const int n = sscanf(tok, "%d-%d", &a, &b);
        if (n < 1) continue;
        if (n == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, &set);
    }
    return sched_setaffinity(0, sizeof set, &set) == 0;
}

static unsigned long long fnv1a(const void* p, size_t n)
{
    const unsigned char* b = (const unsigned char*)p;
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

typedef struct {
    int k, n_inst, W, H, B, n_batches, pageable;
    const SGMOption* opt;
    uint8_t **L, **R;            /* [n_batches] batches of B frames, shared, read-only */
    volatile double stop_at;     /* set by main right after the start barrier */
    long batches_done;
    unsigned long long hash0;    /* hash of frame 0 of batch 0 as this instance computed it (0: never) */
    int failed;
    pthread_barrier_t* start;
} worker;

static void* worker_main(void* p)
{
    worker* w = (worker*)p;
    const size_t px = (size_t)w->W * w->H;
    sgm_instance* s = sgm_create(0);
    float* out = NULL;
    w->failed = 1;
    if (s && sgm_set_batch(s, w->B) && sgm_set_overlap_post(s, 1) && sgm_initialize(s, (uint16_t)w->W, (uint16_t)w->H, w->opt)) {
        out = w->pageable ? (float*)malloc(w->B * px * sizeof(float)) : (float*)sgm_host_alloc(s, w->B * px * sizeof(float));
        w->failed = out == NULL;
    }
    /* one untimed batch: first-use allocations */
    if (!w->failed) w->failed = !(sgm_match_async(s, w->L[0], w->R[0], out) && sgm_match_wait(s));
    pthread_barrier_wait(w->start);
    for (long b = w->k; !w->failed && now_s() < w->stop_at; b += w->n_inst) {
        const int i = (int)(b % w->n_batches);
        if (!(sgm_reset(s, (uint16_t)w->W, (uint16_t)w->H, w->opt) && sgm_match_async(s, w->L[i], w->R[i], out) && sgm_match_wait(s))) {
            w->failed = 1;
            break;
        }
        if (i == 0) w->hash0 = fnv1a(out, px * sizeof(float));
        ++w->batches_done;
    }
    if (s) {
        if (out && !w->pageable) sgm_host_free(s, out);
        else free(out);
        sgm_destroy(s);
    }
    return NULL;
}

int main(int argc, char** argv)
{
    int W = 1242, H = 375, D = 128, B = 8, N = 4, F = 32, pageable = 0, blocking = 0, node = -1;
    double seconds = 2.0;
    unsigned seed = 0x5EED0002u;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "--pageable")) pageable = 1;
        else if (!strcmp(a, "--blocking")) blocking = 1;
        else if (v && !strcmp(a, "--width")) W = atoi(argv[++i]);
        else if (v && !strcmp(a, "--height")) H = atoi(argv[++i]);
        else if (v && !strcmp(a, "--disparities")) D = atoi(argv[++i]);
        else if (v && !strcmp(a, "--batch")) B = atoi(argv[++i]);
        else if (v && !strcmp(a, "--instances")) N = atoi(argv[++i]);
        else if (v && !strcmp(a, "--frames")) F = atoi(argv[++i]);
        else if (v && !strcmp(a, "--seconds")) seconds = atof(argv[++i]);
        else if (v && !strcmp(a, "--numa-node")) node = atoi(argv[++i]);
        else if (v && !strcmp(a, "--seed")) seed = (unsigned)strtoul(argv[++i], NULL, 0);
        else { fprintf(stderr, "sgm_stream: unknown argument %s (see the header of sgm_stream.c)\n", a); return 2; }
    }
    if (W < 1 || H < 1 || D < 1 || B < 1 || N < 1 || N > 16 || F < 1) return 2;
    if (node >= 0 && !pin_to_node(node)) fprintf(stderr, "sgm_stream: could not pin to NUMA node %d (continuing unpinned)\n", node);
    SGMOption opt;
    memset(&opt, 0, sizeof opt);                       /* main.c:48-65 with max_disparity = D */
    opt.num_paths = 8; opt.min_disparity = 0; opt.max_disparity = (uint16_t)D;
    opt.is_check_lr = true; opt.lrcheck_thres = 1.0f; opt.is_check_unique = true; opt.uniqueness_ratio = 0.99;
    opt.is_remove_speckles = true; opt.min_speckle_area = 50; opt.p1 = 10; opt.p2_init = 150;
    const size_t px = (size_t)W * H;

    if (blocking) {
        uint8_t* l = (uint8_t*)malloc(px * F);
        uint8_t* r = (uint8_t*)malloc(px * F);
        float* out = (float*)malloc(px * sizeof(float));
        if (!l || !r || !out) return 1;
        for (int f = 0; f < F; ++f) SGM_SynthPair(W, H, D, seed + (unsigned)f, l + px * f, r + px * f);
        unsigned long long hash0 = 0;
        for (int f = 0; f < 2; ++f)                    /* untimed: first-use allocations */
            if (!sgm_compute(l, r, (uint16_t)W, (uint16_t)H, &opt, out)) { fprintf(stderr, "sgm_stream: sgm_compute failed\n"); return 1; }
        hash0 = fnv1a(out, px * sizeof(float));
        long n = 0;
        const double t0 = now_s();
        while (now_s() - t0 < seconds) {
            const int f = (int)(n % F);
            if (!sgm_compute(l + px * f, r + px * f, (uint16_t)W, (uint16_t)H, &opt, out)) return 1;
            ++n;
        }
        const double el = now_s() - t0;
        printf("{\"mode\": \"blocking sgm_compute per frame, malloc'd buffers, one thread\", \"width\": %d, \"height\": %d, \"disparity_range\": %d, "
               "\"frames\": %ld, \"seconds\": %.4f, \"fps\": %.2f, \"ms_per_frame\": %.4f, \"mdisp_per_s\": %.1f, \"hash_frame0\": \"%016llx\"}\n",
               W, H, D, n, el, n / el, el / n * 1e3, (double)px * D * 8 * n / el / 1e6, hash0);
        SGM_Shutdown();
        free(l); free(r); free(out);
        return 0;
    }

    /* the shared input batches: page-locked (one throw-away instance owns the allocation) or malloc'd */
    const int n_batches = (F + B - 1) / B;
    sgm_instance* owner = sgm_create(0);
    if (!owner) { fprintf(stderr, "sgm_stream: no usable GPU\n"); return 1; }
    uint8_t** L = (uint8_t**)calloc((size_t)n_batches, sizeof *L);
    uint8_t** R = (uint8_t**)calloc((size_t)n_batches, sizeof *R);
    for (int i = 0; i < n_batches; ++i) {
        L[i] = pageable ? (uint8_t*)malloc(px * B) : (uint8_t*)sgm_host_alloc(owner, px * B);
        R[i] = pageable ? (uint8_t*)malloc(px * B) : (uint8_t*)sgm_host_alloc(owner, px * B);
        if (!L[i] || !R[i]) { fprintf(stderr, "sgm_stream: out of host memory\n"); return 1; }
        for (int j = 0; j < B; ++j) SGM_SynthPair(W, H, D, seed + (unsigned)((i * B + j) % F), L[i] + px * j, R[i] + px * j);
    }
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)N + 1);
    worker w[16];
    pthread_t th[16];
    memset(w, 0, sizeof w);
    for (int k = 0; k < N; ++k) {
        w[k] = (worker){k, N, W, H, B, n_batches, pageable, &opt, L, R, 1e300, 0, 0, 0, &start};
        if (pthread_create(&th[k], NULL, worker_main, &w[k]) != 0) return 1;
    }
    /* every worker is set up (its warm-up batch included) when the barrier opens */
    pthread_barrier_wait(&start);
    const double t1 = now_s();
    for (int k = 0; k < N; ++k) w[k].stop_at = t1 + seconds;
    long batches = 0;
    int failed = 0;
    unsigned long long hash0 = 0;
    for (int k = 0; k < N; ++k) {
        pthread_join(th[k], NULL);
        batches += w[k].batches_done;
        failed |= w[k].failed;
        if (w[k].hash0) {
            if (hash0 && hash0 != w[k].hash0) failed = 1;      /* every instance must compute the same map for the same frame */
            hash0 = w[k].hash0;
        }
    }
    const double el = now_s() - t1;
    for (int i = 0; i < n_batches; ++i) {
        if (pageable) { free(L[i]); free(R[i]); }
        else { sgm_host_free(owner, L[i]); sgm_host_free(owner, R[i]); }
    }
    sgm_destroy(owner);
    free(L); free(R);
    const long frames = batches * B;
    printf("{\"mode\": \"%d instances x batches of %d, sgm_reset + sgm_match_async + sgm_match_wait, %s buffers, one thread each\", "
           "\"width\": %d, \"height\": %d, \"disparity_range\": %d, \"frames\": %ld, \"seconds\": %.4f, \"fps\": %.2f, \"ms_per_frame\": %.4f, "
           "\"mdisp_per_s\": %.1f, \"hash_frame0\": \"%016llx\", \"failed\": %s}\n",
           N, B, pageable ? "malloc'd" : "page-locked", W, H, D, frames, el, frames / el, frames ? el / frames * 1e3 : 0.0,
           (double)px * D * 8 * frames / el / 1e6, hash0, failed ? "true" : "false");
    return failed ? 1 : 0;
}
