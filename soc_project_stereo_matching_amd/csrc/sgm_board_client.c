/* sgm_board_client.c -- a host with an MI355X standing in for the ZedBoard against the reference's test platform
 * (HostScript_Server/server.py).  It speaks the board side of the wire protocol and runs the SGM library where the
 * firmware's start_stereo_matching() stub is meant to (ZedBoard/.../src/stereo_matching.c:34-40).
 *
 * Wire protocol (server.py:105-131,148-177; tcp_perf_client.c:73-139,154-189), all little-endian:
 *   board -> server  1 byte request: 0 close, 1 image+calibration, 2 image, 3 result follows
 *   server -> board  header <BiHH> (type, seq, width, height); type 0 = close;
 *                    type 1: + 80 bytes = 20 float32 (cam0 3x3, cam1 3x3, doffs, baseline; stereo_calibration.py:177-195);
 *                    then 6 planes of height rows x width bytes: left B, G, R, right B, G, R
 *   board -> server  byte 3, <iHH> (seq, width, height), height rows of width float32 = depth in mm
 * Grey conversion is the firmware's (76 r + 150 g + 29 b) >> 8 (stereo_matching.c:18-25).
 * Depth follows the platform's own client simulator (client.py:40-45): NaN for invalid disparities,
 * float32(fx * baseline) / (disparity + doffs) otherwise.
 *
 *   sgm_board_client HOST PORT [--max-frames N] [--max-disparity D] [--min-disparity D] [--placeholder-gray]
 * --placeholder-gray reproduces what the firmware does today (main.c:227-233: the "depth" it returns is the grey
 * value of the left image) and needs no GPU; the CPU tests use it to check the framing.
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/sgm_mi355x.h"

#include <arpa/inet.h>
#include <math.h>
#include <netdb.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

static int recv_all(int fd, void* buf, size_t n)
{
    uint8_t* p = (uint8_t*)buf;
    while (n) {
        const ssize_t r = recv(fd, p, n, 0);
        if (r <= 0) return -1;
        p += r; n -= (size_t)r;
    }
    return 0;
}
static int send_all(int fd, const void* buf, size_t n)
{
    const uint8_t* p = (const uint8_t*)buf;
    while (n) {
        const ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) return -1;
        p += r; n -= (size_t)r;
    }
    return 0;
}
static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}
static float le_f32(const uint8_t* p)
{
    uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s HOST PORT [--max-frames N] [--max-disparity D] [--min-disparity D] [--placeholder-gray]\n", argv[0]); return 2; }
    int max_frames = 1 << 30, placeholder = 0;
    SGMOption opt;
    memset(&opt, 0, sizeof opt);                           /* the reference driver's defaults, main.c:48-65 */
    opt.num_paths = 8; opt.min_disparity = 0; opt.max_disparity = 128;
    opt.is_check_lr = true; opt.lrcheck_thres = 1.0f;
    opt.is_check_unique = true; opt.uniqueness_ratio = 0.99;
    opt.is_remove_speckles = true; opt.min_speckle_area = 50;
    opt.p1 = 10; opt.p2_init = 150;
    for (int i = 3; i < argc; ++i) {
        const char* v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(argv[i], "--placeholder-gray")) placeholder = 1;
        else if (v && !strcmp(argv[i], "--max-frames")) { max_frames = atoi(v); ++i; }
        else if (v && !strcmp(argv[i], "--max-disparity")) { opt.max_disparity = (uint16_t)atoi(v); ++i; }
        else if (v && !strcmp(argv[i], "--min-disparity")) { opt.min_disparity = (uint16_t)atoi(v); ++i; }
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }

    struct addrinfo hints, *res = NULL;
    memset(&hints, 0, sizeof hints);
    hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(argv[1], argv[2], &hints, &res) != 0 || !res) { fprintf(stderr, "cannot resolve %s:%s\n", argv[1], argv[2]); return 1; }
    const int fd = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
    if (fd < 0 || connect(fd, res->ai_addr, res->ai_addrlen) != 0) { fprintf(stderr, "cannot connect to %s:%s\n", argv[1], argv[2]); return 1; }
    freeaddrinfo(res);

    float fx = 0.f, doffs = 0.f, baseline = 0.f;
    int have_calib = 0, frames = 0, cur_w = 0, cur_h = 0, rc = 0;
    uint8_t* planes = NULL;
    float* depth = NULL;
    double t_sgm = 0.0;
    /* grey conversion, matching and disparity -> depth all run on the GPU (sgm_match_planes); the planes are received
     * straight into page-locked memory the DMA engines read */
    sgm_instance* sgm = placeholder ? NULL : sgm_create(0);
    if (!placeholder && !sgm) { fprintf(stderr, "no usable GPU\n"); close(fd); return 1; }
    const double t_start = now_s();

    while (frames < max_frames) {
        const uint8_t req = have_calib ? 2 : 1;            /* calibration with the first frame, then images only */
        if (send_all(fd, &req, 1)) { rc = 1; break; }
        uint8_t hdr[9];
        if (recv_all(fd, hdr, 1)) break;                   /* server closed */
        if (hdr[0] == 0) break;                            /* type 0: no more test data */
        if (recv_all(fd, hdr + 1, 8)) { rc = 1; break; }
        const int w = hdr[5] | (hdr[6] << 8), h = hdr[7] | (hdr[8] << 8);
        if (hdr[0] != 1 && hdr[0] != 2) { fprintf(stderr, "unexpected message type %d\n", hdr[0]); rc = 1; break; }
        if (hdr[0] == 1) {
            uint8_t cal[80];
            if (recv_all(fd, cal, 80)) { rc = 1; break; }
            fx = le_f32(cal);                              /* cam0[0][0] */
            doffs = le_f32(cal + 72);
            baseline = le_f32(cal + 76);
            have_calib = 1;
        }
        const size_t px = (size_t)w * h;
        if (w != cur_w || h != cur_h) {
            if (sgm) { sgm_host_free(sgm, planes); sgm_host_free(sgm, depth); } else { free(planes); free(depth); }
            planes = (uint8_t*)(sgm ? sgm_host_alloc(sgm, 6 * px) : malloc(6 * px));
            depth = (float*)(sgm ? sgm_host_alloc(sgm, px * sizeof(float)) : malloc(px * sizeof(float)));
            cur_w = w; cur_h = h;
            if (!planes || !depth) { rc = 1; break; }
        }
        if (recv_all(fd, planes, 6 * px)) { rc = 1; break; }       /* left B, G, R, right B, G, R */
        if (placeholder) {
            /* what the firmware sends today: the left grey image as floats (stereo_matching.c:13-33) */
            for (size_t i = 0; i < px; ++i)
                depth[i] = (float)(uint8_t)((76u * planes[2 * px + i] + 150u * planes[px + i] + 29u * planes[i]) >> 8);
        } else {
            const double t0 = now_s();
            if (!sgm_reset(sgm, (uint16_t)w, (uint16_t)h, &opt) || !sgm_match_planes(sgm, planes, fx, baseline, doffs, depth)) {
                fprintf(stderr, "SGM failed\n"); rc = 1; break;
            }
            t_sgm += now_s() - t0;
        }
        uint8_t out[9] = {3, hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6], hdr[7], hdr[8]};   /* type 3, the frame's seq, w, h */
        if (send_all(fd, out, 9) || send_all(fd, depth, px * sizeof(float))) { rc = 1; break; }
        ++frames;
    }
    const uint8_t bye = 0;
    send_all(fd, &bye, 1);
    close(fd);
    const double dt = now_s() - t_start;
    printf("frames %d in %.3f s (%.2f fps incl. network), SGM %.3f ms/frame\n", frames, dt, frames / (dt > 0 ? dt : 1), frames ? 1e3 * t_sgm / frames : 0.0);
    if (sgm) { sgm_host_free(sgm, planes); sgm_host_free(sgm, depth); sgm_destroy(sgm); } else { free(planes); free(depth); }
    return rc;
}
