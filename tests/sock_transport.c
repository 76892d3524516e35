/*
 * sock_transport.c -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_tiles_processes.py, tests/tiles_rank.c).
 *
 * An sgm_tiles_transport (include/sgm_tiles.h) between PROCESSES of one host over Unix-domain sockets, staged through host
 * memory.  RCCL refuses two ranks on one GPU, so on a one-GPU box the C tile pipeline can meet real peer processes only through
 * something like this: a rehearsal of the multi-process path (every rank its own process, its own HIP context, its own copy
 * of the library), not a transport anybody should ship -- the product transport is RCCL over xGMI (sgm_tiles_rccl_*).
 *
 * Grouped semantics as the pipeline needs them: operations between group_start and group_end progress together (non-blocking
 * sockets driven by poll(), so neighbouring ranks that list their sends first cannot deadlock on full socket buffers); sends
 * and receives between a pair of ranks match in the order they were listed.  group_end returns when every receive of the
 * group has landed in device memory.
 */
#define _GNU_SOURCE
#include "../include/sgm_tiles.h"
#include "../soc_project_stereo_matching_amd/csrc/sgm_device.h"

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#define MAX_RANKS 16
#define MAX_OPS 96

typedef struct {
    int is_send, peer;
    void* d_buf;
    size_t bytes, done;          /* progress in the framed stream: 8-byte length header + payload */
    unsigned char* host;         /* header + payload */
} sock_op;

typedef struct {
    int rank, world, device;
    int fd[MAX_RANKS];
    int listen_fd;
    char path[300];
    sock_op ops[MAX_OPS];
    int n_ops;
    void* stream;                /* the stream the group's operations were queued for */
} sock_ctx;

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static int full_io(int fd, void* buf, size_t n, int writing)
{
    unsigned char* p = (unsigned char*)buf;
    while (n) {
        const ssize_t k = writing ? write(fd, p, n) : read(fd, p, n);
        if (k < 0 && (errno == EINTR || errno == EAGAIN)) { usleep(100); continue; }
        if (k <= 0) return -1;
        p += k; n -= (size_t)k;
    }
    return 0;
}

static int sock_group_start(void* c) { ((sock_ctx*)c)->n_ops = 0; ((sock_ctx*)c)->stream = NULL; return 0; }

static int sock_queue(void* c, int is_send, void* d_buf, size_t bytes, int peer, void* stream)
{
    sock_ctx* x = (sock_ctx*)c;
    if (x->n_ops >= MAX_OPS || peer < 0 || peer >= x->world || peer == x->rank) return -1;
    sock_op* o = &x->ops[x->n_ops++];
    memset(o, 0, sizeof *o);
    o->is_send = is_send; o->peer = peer; o->d_buf = d_buf; o->bytes = bytes;
    x->stream = stream;
    return 0;
}
static int sock_send(void* c, const void* b, size_t n, int peer, void* st) { return sock_queue(c, 1, (void*)b, n, peer, st); }
static int sock_recv(void* c, void* b, size_t n, int peer, void* st) { return sock_queue(c, 0, b, n, peer, st); }

static int sock_group_end(void* c)
{
    sock_ctx* x = (sock_ctx*)c;
    int rc = 0;
    if (x->n_ops == 0) return 0;
    /* what the sends read has been queued on the stream before this call */
    if (x->stream && sgmd_stream_sync(x->device, x->stream) != 0) return -1;
    for (int i = 0; i < x->n_ops && rc == 0; ++i) {
        sock_op* o = &x->ops[i];
        o->host = (unsigned char*)malloc(8 + o->bytes);
        if (!o->host) { rc = -1; break; }
        if (o->is_send) {
            const unsigned long long n = o->bytes;
            memcpy(o->host, &n, 8);
            if (sgmd_d2h_async(x->device, x->stream, o->host + 8, o->d_buf, o->bytes) != 0) rc = -1;
        }
    }
    if (rc == 0 && x->stream && sgmd_stream_sync(x->device, x->stream) != 0) rc = -1;
    /* progress: per peer and direction the FIRST unfinished operation is the active one (order kept) */
    const double deadline = now_s() + 120.0;
    while (rc == 0) {
        struct pollfd pf[2 * MAX_RANKS];
        sock_op* act[2 * MAX_RANKS];
        int n = 0;
        for (int dir = 0; dir < 2; ++dir)
            for (int p = 0; p < x->world; ++p) {
                sock_op* first = NULL;
                for (int i = 0; i < x->n_ops && !first; ++i)
                    if (x->ops[i].peer == p && x->ops[i].is_send == dir && x->ops[i].done < 8 + x->ops[i].bytes) first = &x->ops[i];
                if (!first) continue;
                pf[n].fd = x->fd[p]; pf[n].events = dir ? POLLOUT : POLLIN; pf[n].revents = 0;
                act[n++] = first;
            }
        if (n == 0) break;
        if (now_s() > deadline) { fprintf(stderr, "sock transport: rank %d waited 120 s for its peers\n", x->rank); rc = -1; break; }
        if (poll(pf, (nfds_t)n, 1000) < 0 && errno != EINTR) { rc = -1; break; }
        for (int k = 0; k < n && rc == 0; ++k) {
            if (!(pf[k].revents & (POLLIN | POLLOUT | POLLHUP | POLLERR))) continue;
            sock_op* o = act[k];
            const size_t total = 8 + o->bytes;
            const ssize_t got = o->is_send ? write(pf[k].fd, o->host + o->done, total - o->done)
                                           : read(pf[k].fd, o->host + o->done, total - o->done);
            if (got < 0 && (errno == EAGAIN || errno == EINTR)) continue;
            if (got <= 0) { fprintf(stderr, "sock transport: rank %d lost rank %d\n", x->rank, o->peer); rc = -1; break; }
            o->done += (size_t)got;
            if (!o->is_send && o->done >= 8) {
                unsigned long long announced;
                memcpy(&announced, o->host, 8);
                if (announced != o->bytes) {
                    fprintf(stderr, "sock transport: rank %d expected %zu bytes from rank %d, it sends %llu\n", x->rank, o->bytes, o->peer, announced);
                    rc = -1;
                }
            }
        }
    }
    for (int i = 0; i < x->n_ops && rc == 0; ++i) {
        sock_op* o = &x->ops[i];
        if (!o->is_send && sgmd_h2d_async(x->device, x->stream, o->d_buf, o->host + 8, o->bytes) != 0) rc = -1;
    }
    if (x->stream && sgmd_stream_sync(x->device, x->stream) != 0) rc = -1;       /* the staging buffers are freed below */
    for (int i = 0; i < x->n_ops; ++i) { free(x->ops[i].host); x->ops[i].host = NULL; }
    x->n_ops = 0;
    return rc;
}

static void sock_destroy(void* c)
{
    sock_ctx* x = (sock_ctx*)c;
    if (!x) return;
    for (int p = 0; p < x->world; ++p)
        if (x->fd[p] >= 0) close(x->fd[p]);
    if (x->listen_fd >= 0) close(x->listen_fd);
    unlink(x->path);
    free(x);
}

/* every rank listens on DIR/r<rank>.sock, connects to every HIGHER rank (retrying until that rank is up) and accepts the
 * connections of the lower ones; returns when the full mesh stands.  false after 60 s without it. */
bool sock_transport_create(const char* dir, int rank, int world, int device, sgm_tiles_transport* out)
{
    if (!dir || !out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return false;
    sock_ctx* x = (sock_ctx*)calloc(1, sizeof *x);
    if (!x) return false;
    x->rank = rank; x->world = world; x->device = device; x->listen_fd = -1;
    for (int p = 0; p < MAX_RANKS; ++p) x->fd[p] = -1;
    snprintf(x->path, sizeof x->path, "%s/r%d.sock", dir, rank);
    struct sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    snprintf(a.sun_path, sizeof a.sun_path, "%s", x->path);
    unlink(x->path);
    x->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (x->listen_fd < 0 || bind(x->listen_fd, (struct sockaddr*)&a, sizeof a) != 0 || listen(x->listen_fd, MAX_RANKS) != 0) {
        perror("sock transport: listen");
        sock_destroy(x);
        return false;
    }
    const double deadline = now_s() + 60.0;
    for (int p = rank + 1; p < world; ++p) {
        struct sockaddr_un b;
        memset(&b, 0, sizeof b);
        b.sun_family = AF_UNIX;
        snprintf(b.sun_path, sizeof b.sun_path, "%s/r%d.sock", dir, p);
        for (;;) {
            const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, (struct sockaddr*)&b, sizeof b) == 0) {
                const int me = rank;
                if (full_io(fd, (void*)&me, sizeof me, 1) != 0) { close(fd); sock_destroy(x); return false; }
                x->fd[p] = fd;
                break;
            }
            if (fd >= 0) close(fd);
            if (now_s() > deadline) { fprintf(stderr, "sock transport: rank %d cannot reach rank %d\n", rank, p); sock_destroy(x); return false; }
            usleep(20000);
        }
    }
    for (int k = 0; k < rank; ++k) {
        struct pollfd pf = {x->listen_fd, POLLIN, 0};
        if (poll(&pf, 1, (int)((deadline - now_s()) * 1000)) <= 0) { fprintf(stderr, "sock transport: rank %d: a lower rank never connected\n", rank); sock_destroy(x); return false; }
        const int fd = accept(x->listen_fd, NULL, NULL);
        int who = -1;
        if (fd < 0 || full_io(fd, &who, sizeof who, 0) != 0 || who < 0 || who >= rank || x->fd[who] >= 0) { if (fd >= 0) close(fd); sock_destroy(x); return false; }
        x->fd[who] = fd;
    }
    for (int p = 0; p < world; ++p)
        if (x->fd[p] >= 0) fcntl(x->fd[p], F_SETFL, fcntl(x->fd[p], F_GETFL, 0) | O_NONBLOCK);
    out->ctx = x;
    out->group_start = sock_group_start; out->group_end = sock_group_end;
    out->send = sock_send; out->recv = sock_recv; out->destroy = sock_destroy;
    return true;
}
