/*
 * stub_rccl.c -- TEST INFRASTRUCTURE ONLY (tests/test_tiles_c.py).
 *
 * A stand-in for librccl with the eight entry points csrc/sgm_tiles.c binds at run time (SGM_RCCL_LIBRARY points at the
 * library built from this file), for ranks that are THREADS of one test process and "device" memory that is host memory
 * (tests/stub_device.c).  Semantics kept from RCCL: operations between ncclGroupStart and ncclGroupEnd are issued together;
 * sends and receives between a pair of ranks match in the order they were listed; a group never deadlocks on the order in
 * which neighbouring ranks list their operations.  To make sure the caller relies on nothing else, ncclGroupEnd works
 * through its peers in DESCENDING rank order and posts every send before it waits for any receive -- not the order the
 * operations were listed in.  stub_rccl_stats() tells the test how many messages and bytes went through.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAX_RANKS 16
#define MAX_OPS 256

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct msg { void* data; size_t bytes; struct msg* next; } msg;
typedef struct { msg *head, *tail; } mailbox;

typedef struct world {
    char id[128];
    int nranks, joined, destroyed;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    mailbox box[MAX_RANKS][MAX_RANKS];       /* [src][dst] */
    struct world* next;
} world;

typedef struct { world* w; int rank; } comm;
typedef struct { int is_send; void* buf; size_t bytes; int peer; comm* c; } op;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static world* g_worlds;
static unsigned g_next_id = 1;
static unsigned long long g_messages, g_bytes;
static __thread op t_ops[MAX_OPS];
static __thread int t_nops, t_depth;

void stub_rccl_stats(unsigned long long* messages, unsigned long long* bytes)
{
    pthread_mutex_lock(&g_mu);
    *messages = g_messages; *bytes = g_bytes;
    pthread_mutex_unlock(&g_mu);
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "success" : "stub rccl error"; }

int ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    pthread_mutex_lock(&g_mu);
    snprintf(id->internal, sizeof id->internal, "stub-rccl-%u", g_next_id++);
    pthread_mutex_unlock(&g_mu);
    return 0;
}

int ncclCommInitRank(void** out, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;
    pthread_mutex_lock(&g_mu);
    world* w = g_worlds;
    while (w && memcmp(w->id, id.internal, sizeof w->id) != 0) w = w->next;
    if (!w) {
        w = (world*)calloc(1, sizeof *w);
        memcpy(w->id, id.internal, sizeof w->id);
        w->nranks = nranks;
        pthread_mutex_init(&w->mu, NULL);
        pthread_cond_init(&w->cv, NULL);
        w->next = g_worlds;
        g_worlds = w;
    }
    pthread_mutex_unlock(&g_mu);
    if (w->nranks != nranks) return 4;
    /* like the real call: returns when every rank of the communicator has joined */
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += 60;
    pthread_mutex_lock(&w->mu);
    ++w->joined;
    pthread_cond_broadcast(&w->cv);
    int rc = 0;
    while (w->joined < nranks && rc == 0) rc = pthread_cond_timedwait(&w->cv, &w->mu, &until);
    pthread_mutex_unlock(&w->mu);
    if (rc != 0) return 6;
    comm* c = (comm*)calloc(1, sizeof *c);
    c->w = w; c->rank = rank;
    *out = c;
    return 0;
}

int ncclCommDestroy(void* cv)
{
    comm* c = (comm*)cv;
    if (!c) return 4;
    /* the world itself stays (a few hundred bytes per test): another rank may still be inside its last group */
    free(c);
    return 0;
}

static int run_ops(void)
{
    int rc = 0;
    /* every send first, peers in descending order; the order of the operations of one pair of ranks is kept */
    for (int peer = MAX_RANKS - 1; peer >= 0; --peer)
        for (int i = 0; i < t_nops; ++i) {
            op* o = &t_ops[i];
            if (!o->is_send || o->peer != peer) continue;
            world* w = o->c->w;
            msg* m = (msg*)calloc(1, sizeof *m);
            m->data = malloc(o->bytes ? o->bytes : 1);
            memcpy(m->data, o->buf, o->bytes);
            m->bytes = o->bytes;
            pthread_mutex_lock(&w->mu);
            mailbox* b = &w->box[o->c->rank][peer];
            if (b->tail) b->tail->next = m; else b->head = m;
            b->tail = m;
            pthread_cond_broadcast(&w->cv);
            pthread_mutex_unlock(&w->mu);
            pthread_mutex_lock(&g_mu);
            ++g_messages; g_bytes += o->bytes;
            pthread_mutex_unlock(&g_mu);
        }
    for (int peer = MAX_RANKS - 1; peer >= 0 && rc == 0; --peer)
        for (int i = 0; i < t_nops && rc == 0; ++i) {
            op* o = &t_ops[i];
            if (o->is_send || o->peer != peer) continue;
            world* w = o->c->w;
            struct timespec until;
            clock_gettime(CLOCK_REALTIME, &until);
            until.tv_sec += 60;
            pthread_mutex_lock(&w->mu);
            mailbox* b = &w->box[peer][o->c->rank];
            int waited = 0;
            while (!b->head && waited == 0) waited = pthread_cond_timedwait(&w->cv, &w->mu, &until);
            msg* m = b->head;
            if (m) { b->head = m->next; if (!b->head) b->tail = NULL; }
            pthread_mutex_unlock(&w->mu);
            if (!m) { fprintf(stderr, "stub rccl: rank %d waited 60 s for rank %d\n", o->c->rank, peer); rc = 6; break; }
            if (m->bytes != o->bytes) { fprintf(stderr, "stub rccl: rank %d expected %zu bytes from rank %d, got %zu\n", o->c->rank, o->bytes, peer, m->bytes); rc = 5; }
            else memcpy(o->buf, m->data, o->bytes);
            free(m->data);
            free(m);
        }
    t_nops = 0;
    return rc;
}

int ncclGroupStart(void) { ++t_depth; return 0; }
int ncclGroupEnd(void)
{
    if (t_depth <= 0) return 4;
    if (--t_depth > 0) return 0;
    return run_ops();
}

static int queue_op(int is_send, void* buf, size_t count, int dtype, int peer, void* c)
{
    if (dtype != 1 /* ncclUint8 */ || !c || peer < 0 || peer >= ((comm*)c)->w->nranks || t_nops >= MAX_OPS) return 4;
    t_ops[t_nops++] = (op){is_send, buf, count, peer, (comm*)c};
    return t_depth > 0 ? 0 : run_ops();
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* c, void* stream) { (void)stream; return queue_op(1, (void*)buf, count, dtype, peer, c); }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* c, void* stream) { (void)stream; return queue_op(0, buf, count, dtype, peer, c); }
