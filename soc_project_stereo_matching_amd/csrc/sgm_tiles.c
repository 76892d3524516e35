/*
 * sgm_tiles.c -- the row-tile pipeline of one rank on its GPU and the transports between ranks (include/sgm_tiles.h,
 * layers 2 and 3).  Plain C on the launcher interface of sgm_device.h and the instance API of sgm_mi355x.h; RCCL is bound at
 * run time.  The step order is sgm_tile_step's (sgm_tile_sched.c); this file is the engine behind it:
 *
 *   slot            = an sgm_instance restricted to the rank's rows (its planes hold 1/N of a frame), its HIP stream, four
 *                     hand-over buffers (forward / backward x incoming / outgoing) and a [batch][H][W] disparity map
 *   exchange        = the communication stream waits for what the touched slots have queued (one event each), runs ONE
 *                     transport group, and the touched slots' streams wait for it -- the host never blocks
 *   run-ahead       = bounded by `throttle` steps (an event per step)
 *
 * The reference has nothing of this (its matcher is one frame on one core, SemiGlobalMatching.c:68-125); the call site this
 * serves is a C caller with a stream of frames, ZedBoard/Vitis/lwip_tcp_perf_client/src/stereo_matching.c:34-40.
 */
#include "../../include/sgm_tiles.h"
#include "sgm_device.h"

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define TFAIL(...)                                       \
    do {                                                 \
        fprintf(stderr, "sgm_mi355x (tiles): " __VA_ARGS__); \
        fputc('\n', stderr);                             \
        return false;                                    \
    } while (0)

/* ================================================================================================ the pipeline of one rank */

typedef struct {
    sgm_instance* inst;
    void* stream;                 /* = sgm_stream(inst) */
    void* d_map;                  /* [batch][H][W] float32 */
    void* d_pack;                 /* batches on several ranks: the row gather's message buffer, [rank k: [batch][rows of k][W]] at batch * r0(k) * W */
    void* bnd[2][2];              /* [forward][incoming] hand-over buffers */
    void* ev_done;                /* re-recorded behind the slot's last queued work */
} tile_slot;

struct sgm_tiles {
    int device, rank, world, W, H, batch, lead, throttle, nslots, honor;
    SGMOption opt;
    int r0, r1;
    size_t bnd_bytes;
    tile_slot* slots;
    void* comm_stream;
    void* ev_comm;
    void** step_done;             /* throttle events, one per step in flight */
    sgm_tiles_transport tr;
    long frames;                  /* frames submitted to the current stream = index of the next step */
    const uint8_t *cur_left, *cur_right;
    void* cur_ready;
    sgm_tiles_result_fn on_result;
    void* on_result_user;
    float* d_ring;
    int ring_frames;
    bool ready;                   /* instances initialised (first submit) */
};

static int eng_begin(void* u, int slot, long frame)
{
    (void)frame;
    sgm_tiles* t = (sgm_tiles*)u;
    tile_slot* s = &t->slots[slot];
    if (t->cur_ready && sgmd_stream_wait_event(t->device, s->stream, t->cur_ready) != 0) return -2;
    /* per frame; allocates nothing after the first */
    if (!sgm_reset(s->inst, (uint16_t)t->W, (uint16_t)t->H, &t->opt)) return -3;
    return sgm_tile_begin(s->inst, t->cur_left, t->cur_right) ? 0 : -4;
}
static int eng_import(void* u, int slot, int forward)
{
    sgm_tiles* t = (sgm_tiles*)u;
    return sgm_tile_import_boundary(t->slots[slot].inst, forward, t->slots[slot].bnd[forward ? 1 : 0][1]) ? 0 : -5;
}
static int eng_sweep(void* u, int slot, int forward)
{
    sgm_tiles* t = (sgm_tiles*)u;
    return sgm_tile_sweep(t->slots[slot].inst, forward) ? 0 : -6;
}
static int eng_export(void* u, int slot, int forward)
{
    sgm_tiles* t = (sgm_tiles*)u;
    return sgm_tile_export_boundary(t->slots[slot].inst, forward, t->slots[slot].bnd[forward ? 1 : 0][0]) ? 0 : -7;
}
static int eng_finish(void* u, int slot)
{
    sgm_tiles* t = (sgm_tiles*)u;
    return sgm_tile_finish(t->slots[slot].inst, (float*)t->slots[slot].d_map) ? 0 : -8;
}
static int eng_post(void* u, int slot, long frame)
{
    sgm_tiles* t = (sgm_tiles*)u;
    tile_slot* s = &t->slots[slot];
    if (!sgm_tile_post(s->inst, (float*)s->d_map)) return -9;
    const size_t map_bytes = (size_t)t->batch * t->W * t->H * sizeof(float);
    if (t->d_ring && t->ring_frames > 0) {
        char* dst = (char*)t->d_ring + (size_t)((frame / t->world) % t->ring_frames) * map_bytes;
        if (sgmd_d2d_async(t->device, s->stream, dst, s->d_map, map_bytes) != 0) return -10;
    }
    if (t->on_result) {
        if (sgmd_event_record(t->device, s->ev_done, s->stream) != 0) return -11;
        t->on_result(t->on_result_user, frame, (const float*)s->d_map, s->ev_done);
    }
    return 0;
}
static int eng_exchange(void* u, const sgm_tile_xop* ops, int n_ops, const int* slots, int n_slots)
{
    sgm_tiles* t = (sgm_tiles*)u;
    const int dev = t->device;
    /* the communication stream waits for everything queued on the slots whose buffers the operations read or overwrite */
    for (int i = 0; i < n_slots; ++i) {
        tile_slot* s = &t->slots[slots[i]];
        if (sgmd_event_record(dev, s->ev_done, s->stream) != 0 || sgmd_stream_wait_event(dev, t->comm_stream, s->ev_done) != 0) return -12;
    }
    if (t->tr.group_start(t->tr.ctx) != 0) return -13;
    int rc = 0, unpack[64], n_unpack = 0;
    for (int i = 0; i < n_ops && rc == 0; ++i) {
        const sgm_tile_xop* o = &ops[i];
        tile_slot* s = &t->slots[o->slot];
        if (o->buf == SGM_XBUF_BOUNDARY) {
            void* b = s->bnd[o->forward ? 1 : 0][o->incoming ? 1 : 0];
            rc = o->kind == SGM_XOP_SEND ? t->tr.send(t->tr.ctx, b, t->bnd_bytes, o->peer, t->comm_stream)
                                         : t->tr.recv(t->tr.ctx, b, t->bnd_bytes, o->peer, t->comm_stream);
        } else {
            /* rows [row_begin, row_end) of every map of the slot.  One frame per step: they are contiguous in the map.  A batch: the
             * same rows of its B maps travel as ONE message per peer -- packed into / unpacked from the slot's d_pack by one strided
             * device copy on the communication stream (per step world - 1 messages on the owner instead of (world - 1) * B; the
             * exchange is latency-bound) */
            const size_t rows = (size_t)(o->row_end - o->row_begin), row_bytes = (size_t)t->W * sizeof(float);
            char* const in_map = (char*)s->d_map + (size_t)o->row_begin * row_bytes;
            if (t->batch == 1) {
                rc = o->kind == SGM_XOP_SEND ? t->tr.send(t->tr.ctx, in_map, rows * row_bytes, o->peer, t->comm_stream)
                                             : t->tr.recv(t->tr.ctx, in_map, rows * row_bytes, o->peer, t->comm_stream);
            } else {
                char* const packed = (char*)s->d_pack + (size_t)t->batch * (size_t)o->row_begin * row_bytes;
                if (o->kind == SGM_XOP_SEND) {
                    rc = sgmd_d2d_2d_async(dev, t->comm_stream, packed, rows * row_bytes, in_map, (size_t)t->H * row_bytes, rows * row_bytes,
                                           (size_t)t->batch) != 0
                             ? -1 : t->tr.send(t->tr.ctx, packed, (size_t)t->batch * rows * row_bytes, o->peer, t->comm_stream);
                } else {
                    rc = t->tr.recv(t->tr.ctx, packed, (size_t)t->batch * rows * row_bytes, o->peer, t->comm_stream);
                    if (n_unpack < 64) unpack[n_unpack++] = i;
                    else rc = -1;
                }
            }
        }
    }
    if (t->tr.group_end(t->tr.ctx) != 0 || rc != 0) return -14;
    for (int k = 0; k < n_unpack; ++k) {                             /* behind the group: the received rows into their maps */
        const sgm_tile_xop* o = &ops[unpack[k]];
        tile_slot* s = &t->slots[o->slot];
        const size_t rows = (size_t)(o->row_end - o->row_begin), row_bytes = (size_t)t->W * sizeof(float);
        if (sgmd_d2d_2d_async(dev, t->comm_stream, (char*)s->d_map + (size_t)o->row_begin * row_bytes, (size_t)t->H * row_bytes,
                              (char*)s->d_pack + (size_t)t->batch * (size_t)o->row_begin * row_bytes, rows * row_bytes, rows * row_bytes,
                              (size_t)t->batch) != 0) return -17;
    }
    /* ... and afterwards those slots' streams wait for the exchange */
    if (sgmd_event_record(dev, t->ev_comm, t->comm_stream) != 0) return -15;
    for (int i = 0; i < n_slots; ++i)
        if (sgmd_stream_wait_event(dev, t->slots[slots[i]].stream, t->ev_comm) != 0) return -16;
    return 0;
}

static const sgm_tile_engine* engine_of(sgm_tiles* t, sgm_tile_engine* e)
{
    e->user = t;
    e->begin = eng_begin; e->import_boundary = eng_import; e->sweep = eng_sweep; e->export_boundary = eng_export;
    e->exchange = eng_exchange; e->finish = eng_finish; e->post = eng_post;
    return e;
}

void sgm_tiles_destroy(sgm_tiles* t)
{
    if (!t) return;
    if (t->slots) {
        for (int i = 0; i < t->nslots; ++i) {
            tile_slot* s = &t->slots[i];
            if (s->inst) sgm_synchronize(s->inst);
        }
        if (t->comm_stream) sgmd_stream_sync(t->device, t->comm_stream);
        for (int i = 0; i < t->nslots; ++i) {
            tile_slot* s = &t->slots[i];
            sgmd_event_destroy(t->device, s->ev_done);
            sgmd_free(t->device, s->d_map);
            sgmd_free(t->device, s->d_pack);
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) sgmd_free(t->device, s->bnd[a][b]);
            sgm_destroy(s->inst);
        }
        free(t->slots);
    }
    if (t->step_done) {
        for (int i = 0; i < t->throttle; ++i) sgmd_event_destroy(t->device, t->step_done[i]);
        free(t->step_done);
    }
    sgmd_event_destroy(t->device, t->ev_comm);
    if (t->comm_stream) sgmd_stream_destroy(t->device, t->comm_stream);
    free(t);
}

sgm_tiles* sgm_tiles_create(int device, int rank, int world, uint16_t width, uint16_t height, const SGMOption* option, int batch,
                            int lead, int spare, int throttle, const sgm_tiles_transport* transport)
{
    if (!option || world < 1 || rank < 0 || rank >= world || batch < 1 || lead < 0 || spare < 1 || throttle < 0) return NULL;
    if (world > 1 && (!transport || !transport->send || !transport->recv || !transport->group_start || !transport->group_end)) return NULL;
    sgm_tiles* t = (sgm_tiles*)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->device = device; t->rank = rank; t->world = world; t->W = width; t->H = height; t->batch = batch;
    t->lead = lead; t->throttle = throttle; t->opt = *option;
    if (transport) t->tr = *transport;
    if (!sgm_tile_rows(height, world, rank, &t->r0, &t->r1)) {
        fprintf(stderr, "sgm_mi355x (tiles): cannot cut %d rows into %d tiles\n", height, world);
        free(t);
        return NULL;
    }
    t->nslots = sgm_tile_slots_needed(world, lead) + spare - 1;
    {
        const int D = (uint16_t)(option->max_disparity - option->min_disparity);
        const size_t per_slot = sgm_tile_slot_bytes(t->r0, t->r1, width, height, option, batch);
        size_t free_b = 0, total_b = 0;
        if (sgmd_mem_info(device, &free_b, &total_b) == 0 && per_slot * (size_t)t->nslots > free_b) {
            const size_t fit = free_b / (per_slot / (size_t)batch) / (size_t)t->nslots;
            fprintf(stderr, "sgm_mi355x (tiles): %d slots x batch %d of %dx%d D=%d need about %.1f GB on device %d, %.1f GB are free: "
                            "use a batch of at most %zu, a smaller lead, or more ranks\n", t->nslots, batch, width, height, D,
                    (double)per_slot * t->nslots / 1e9, device, (double)free_b / 1e9, fit);
            free(t);
            return NULL;
        }
    }
    t->slots = (tile_slot*)calloc((size_t)t->nslots, sizeof *t->slots);
    bool ok = t->slots != NULL;
    const size_t map_bytes = (size_t)batch * width * height * sizeof(float);
    /* a slot costs batch x (8 planes of the tile's rows + 2 hand-over rows + ~60 B per pixel of the whole frame): say so before
     * the allocator does, with what would fit */
    for (int i = 0; ok && i < t->nslots; ++i) {
        tile_slot* s = &t->slots[i];
        s->inst = sgm_create(device);
        ok = s->inst && sgm_set_batch(s->inst, batch) && sgm_set_rows(s->inst, t->r0, t->r1);
        if (ok) s->stream = sgm_stream(s->inst);
        ok = ok && sgmd_event_create(device, &s->ev_done) == 0 && sgmd_alloc(device, &s->d_map, map_bytes) == 0;
        if (ok && world > 1 && batch > 1) ok = sgmd_alloc(device, &s->d_pack, map_bytes) == 0;
    }
    ok = ok && sgmd_stream_create(device, &t->comm_stream) == 0 && sgmd_event_create(device, &t->ev_comm) == 0;
    if (ok && throttle > 0) {
        t->step_done = (void**)calloc((size_t)throttle, sizeof(void*));
        ok = t->step_done != NULL;
        for (int i = 0; ok && i < throttle; ++i) ok = sgmd_event_create(device, &t->step_done[i]) == 0;
    }
    if (!ok) {
        fprintf(stderr, "sgm_mi355x (tiles): setting up %d slots of %dx%d (batch %d) on device %d failed\n", t->nslots, width, height, batch, device);
        sgm_tiles_destroy(t);
        return NULL;
    }
    return t;
}

void sgm_tiles_set_honor_num_paths(sgm_tiles* t, int honor) { if (t) t->honor = honor; }
void sgm_tiles_on_result(sgm_tiles* t, sgm_tiles_result_fn fn, void* user) { if (t) { t->on_result = fn; t->on_result_user = user; } }
void sgm_tiles_result_ring(sgm_tiles* t, float* d_ring, int ring_frames) { if (t) { t->d_ring = d_ring; t->ring_frames = ring_frames; } }
void sgm_tiles_info(const sgm_tiles* t, int* r0, int* r1, int* slots)
{
    if (!t) return;
    if (r0) *r0 = t->r0;
    if (r1) *r1 = t->r1;
    if (slots) *slots = t->nslots;
}

/* the instances are initialised at the first submit (options such as the 4-path mode may be set until then); the hand-over
 * buffers need the instance's answer for their size */
static bool ensure_ready(sgm_tiles* t)
{
    if (t->ready) return true;
    for (int i = 0; i < t->nslots; ++i) {
        tile_slot* s = &t->slots[i];
        sgm_set_honor_num_paths(s->inst, t->honor);
        if (!sgm_reset(s->inst, (uint16_t)t->W, (uint16_t)t->H, &t->opt)) TFAIL("sgm_reset of slot %d failed", i);
        t->bnd_bytes = sgm_tile_boundary_bytes(s->inst);
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                if (!s->bnd[a][b] && sgmd_alloc(t->device, &s->bnd[a][b], t->bnd_bytes) != 0) TFAIL("hand-over buffers: out of device memory");
    }
    t->ready = true;
    return true;
}

static bool run_step(sgm_tiles* t, long step, long frames_known)
{
    if (t->throttle > 0) {
        void* ev = t->step_done[step % t->throttle];
        if (step >= t->throttle && sgmd_event_sync(t->device, ev) != 0) TFAIL("waiting for step %ld failed", step - t->throttle);
    }
    sgm_tile_engine e;
    const int rc = sgm_tile_step(engine_of(t, &e), t->rank, t->world, t->H, t->nslots, t->lead, step, frames_known);
    if (rc != 0) TFAIL("step %ld of rank %d failed (%d)", step, t->rank, rc);
    if (t->throttle > 0) {
        /* behind the work of the frame begun in this step if there is one, else behind the exchange */
        void* st = step < frames_known ? t->slots[step % t->nslots].stream : t->comm_stream;
        if (sgmd_event_record(t->device, t->step_done[step % t->throttle], st) != 0) TFAIL("recording step %ld failed", step);
    }
    return true;
}

bool sgm_tiles_submit(sgm_tiles* t, const uint8_t* d_left, const uint8_t* d_right, void* ready_event)
{
    if (!t || !d_left || !d_right) return false;
    if (!ensure_ready(t)) return false;
    t->cur_left = d_left; t->cur_right = d_right; t->cur_ready = ready_event;
    const bool ok = run_step(t, t->frames, t->frames + 1);
    t->cur_left = t->cur_right = NULL; t->cur_ready = NULL;
    if (ok) ++t->frames;
    return ok;
}

bool sgm_tiles_finish(sgm_tiles* t)
{
    if (!t) return false;
    bool ok = true;
    if (t->frames > 0) {
        const long total = sgm_tile_steps_total(t->frames, t->world, t->lead);
        for (long step = t->frames; ok && step < total; ++step) ok = run_step(t, step, t->frames);
    }
    for (int i = 0; i < t->nslots; ++i)
        if (t->slots[i].inst && !sgm_synchronize(t->slots[i].inst)) ok = false;
    if (sgmd_stream_sync(t->device, t->comm_stream) != 0) ok = false;
    t->frames = 0;
    return ok;
}

/* ================================================================================================ RCCL transport */

#include "sgm_rccl_abi.h"
static rccl_api g_rccl;
static pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;

static bool g_rccl_forced;                                            /* SGM_RCCL_LIBRARY names the library: bind to nothing else */
static void* rccl_sym(const char* name)
{
    void* p = g_rccl_forced ? NULL : dlsym(RTLD_DEFAULT, name);       /* the copy the process already uses, if it is visible */
    if (!p && g_rccl.lib) p = dlsym(g_rccl.lib, name);
    return p;
}
static bool rccl_bind(void)
{
    pthread_mutex_lock(&g_rccl_mu);
    bool ok = g_rccl.Send != NULL;
    if (!ok) {
        const char* forced = getenv("SGM_RCCL_LIBRARY");
        if (forced && *forced) {
            g_rccl_forced = true;
            if (!g_rccl.lib) g_rccl.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else if (!dlsym(RTLD_DEFAULT, "ncclSend")) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (size_t i = 0; i < sizeof names / sizeof names[0] && !g_rccl.lib; ++i)
                g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);                                  /* already loaded? */
            for (size_t i = 0; i < sizeof names / sizeof names[0] && !g_rccl.lib; ++i)
                g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        }
        *(void**)&g_rccl.GetUniqueId = rccl_sym("ncclGetUniqueId");
        *(void**)&g_rccl.CommInitRank = rccl_sym("ncclCommInitRank");
        *(void**)&g_rccl.CommDestroy = rccl_sym("ncclCommDestroy");
        *(void**)&g_rccl.GroupStart = rccl_sym("ncclGroupStart");
        *(void**)&g_rccl.GroupEnd = rccl_sym("ncclGroupEnd");
        *(void**)&g_rccl.Recv = rccl_sym("ncclRecv");
        *(void**)&g_rccl.GetErrorString = rccl_sym("ncclGetErrorString");
        *(void**)&g_rccl.Send = rccl_sym("ncclSend");
        ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.Recv && g_rccl.Send;
        if (!ok) {
            const char* why = dlerror();
            fprintf(stderr, "sgm_mi355x (tiles): RCCL is not available (%s); set SGM_RCCL_LIBRARY to librccl.so\n", why ? why : "symbols missing");
            g_rccl.Send = NULL;
        }
    }
    pthread_mutex_unlock(&g_rccl_mu);
    return ok;
}
static int rccl_check(int rc, const char* what)
{
    if (rc != 0) fprintf(stderr, "sgm_mi355x (tiles): %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return rc;
}

typedef struct { void* comm; int device; } rccl_ctx;
static int rccl_group_start(void* c) { (void)c; return rccl_check(g_rccl.GroupStart(), "ncclGroupStart"); }
static int rccl_group_end(void* c) { (void)c; return rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd"); }
static int rccl_send(void* c, const void* b, size_t n, int peer, void* st)
{
    return rccl_check(g_rccl.Send(b, n, RCCL_UINT8, peer, ((rccl_ctx*)c)->comm, st), "ncclSend");
}
static int rccl_recv(void* c, void* b, size_t n, int peer, void* st)
{
    return rccl_check(g_rccl.Recv(b, n, RCCL_UINT8, peer, ((rccl_ctx*)c)->comm, st), "ncclRecv");
}
static void rccl_destroy(void* c)
{
    rccl_ctx* x = (rccl_ctx*)c;
    if (x && x->comm) g_rccl.CommDestroy(x->comm);
    free(x);
}

bool sgm_tiles_rccl_unique_id(void* id_out)
{
    if (!id_out || !rccl_bind()) return false;
    rccl_uid id;
    memset(&id, 0, sizeof id);
    if (rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId") != 0) return false;
    memcpy(id_out, &id, sizeof id);
    return true;
}

bool sgm_tiles_rccl_transport(const void* id, int rank, int world, int device, sgm_tiles_transport* out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world || !rccl_bind()) return false;
    if (sgmd_device_is_gfx950(device) != 1) TFAIL("device %d is not a usable gfx950 GPU", device);
    rccl_ctx* x = (rccl_ctx*)calloc(1, sizeof *x);
    if (!x) return false;
    x->device = device;
    rccl_uid uid;
    memcpy(&uid, id, sizeof uid);
    if (sgmd_set_device(device) != 0) { free(x); return false; }    /* the communicator binds to the calling thread's current device */
    if (rccl_check(g_rccl.CommInitRank(&x->comm, world, uid, rank), "ncclCommInitRank") != 0) { free(x); return false; }
    out->ctx = x;
    out->group_start = rccl_group_start; out->group_end = rccl_group_end;
    out->send = rccl_send; out->recv = rccl_recv; out->destroy = rccl_destroy;
    return true;
}

/* ================================================================================================ local transport */
/* Ranks = threads of one process on one GPU.  A message = a staging buffer + two events (`sent`: the sender's copy into the
 * stage is done; `taken`: the receiver's copy out of it is done).  Messages are recycled per sender. */

typedef struct local_msg {
    void* stage;
    size_t cap, bytes;
    void *ev_sent, *ev_taken;
    bool used_before;
    struct local_msg* next;
} local_msg;

typedef struct { local_msg *head, *tail; } local_queue;

struct sgm_tiles_local {
    int world, device;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    local_queue* q;               /* [src * world + dst]: in flight */
    local_msg** pool;             /* [src]: free list */
};

typedef struct { void* buf; size_t bytes; int peer; void* stream; } local_recv;
typedef struct {
    sgm_tiles_local* g;
    int rank;
    local_recv pending[80];
    int n_pending;
} local_ctx;

sgm_tiles_local* sgm_tiles_local_group(int world, int device)
{
    if (world < 1) return NULL;
    sgm_tiles_local* g = (sgm_tiles_local*)calloc(1, sizeof *g);
    if (!g) return NULL;
    g->world = world; g->device = device;
    pthread_mutex_init(&g->mu, NULL);
    pthread_cond_init(&g->cv, NULL);
    g->q = (local_queue*)calloc((size_t)world * world, sizeof *g->q);
    g->pool = (local_msg**)calloc((size_t)world, sizeof *g->pool);
    if (!g->q || !g->pool) { free(g->q); free(g->pool); free(g); return NULL; }
    return g;
}

static void local_free_msg(sgm_tiles_local* g, local_msg* m)
{
    sgmd_free(g->device, m->stage);
    sgmd_event_destroy(g->device, m->ev_sent);
    sgmd_event_destroy(g->device, m->ev_taken);
    free(m);
}

void sgm_tiles_local_destroy(sgm_tiles_local* g)
{
    if (!g) return;
    for (int i = 0; i < g->world * g->world; ++i)
        for (local_msg* m = g->q[i].head; m;) { local_msg* n = m->next; local_free_msg(g, m); m = n; }
    for (int i = 0; i < g->world; ++i)
        for (local_msg* m = g->pool[i]; m;) { local_msg* n = m->next; local_free_msg(g, m); m = n; }
    pthread_mutex_destroy(&g->mu);
    pthread_cond_destroy(&g->cv);
    free(g->q); free(g->pool); free(g);
}

static int local_group_start(void* c) { ((local_ctx*)c)->n_pending = 0; return 0; }

static int local_send(void* c, const void* buf, size_t bytes, int peer, void* stream)
{
    local_ctx* x = (local_ctx*)c;
    sgm_tiles_local* g = x->g;
    if (peer < 0 || peer >= g->world) return -1;
    pthread_mutex_lock(&g->mu);
    local_msg *m = NULL, **pp = &g->pool[x->rank];
    for (; *pp; pp = &(*pp)->next)
        if ((*pp)->cap >= bytes) { m = *pp; *pp = m->next; break; }
    pthread_mutex_unlock(&g->mu);
    if (!m) {
        m = (local_msg*)calloc(1, sizeof *m);
        if (!m) return -1;
        if (sgmd_alloc(g->device, &m->stage, bytes) != 0 || sgmd_event_create(g->device, &m->ev_sent) != 0 ||
            sgmd_event_create(g->device, &m->ev_taken) != 0) { local_free_msg(g, m); return -1; }
        m->cap = bytes;
    }
    /* the previous receiver must have copied the stage out before it is overwritten */
    if ((m->used_before && sgmd_stream_wait_event(g->device, stream, m->ev_taken) != 0) ||
        sgmd_d2d_async(g->device, stream, m->stage, buf, bytes) != 0 || sgmd_event_record(g->device, m->ev_sent, stream) != 0) {
        pthread_mutex_lock(&g->mu);                                   /* back to the pool: the group frees it */
        m->next = g->pool[x->rank];
        g->pool[x->rank] = m;
        pthread_mutex_unlock(&g->mu);
        return -1;
    }
    m->bytes = bytes; m->next = NULL; m->used_before = true;
    pthread_mutex_lock(&g->mu);
    local_queue* q = &g->q[x->rank * g->world + peer];
    if (q->tail) q->tail->next = m; else q->head = m;
    q->tail = m;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
    return 0;
}

static int local_recv_post(void* c, void* buf, size_t bytes, int peer, void* stream)
{
    local_ctx* x = (local_ctx*)c;
    if (peer < 0 || peer >= x->g->world) return -1;
    if (x->n_pending >= (int)(sizeof x->pending / sizeof x->pending[0])) {   /* sgm_tile_step lists at most 2 + (world - 1 <= 64) receives */
        fprintf(stderr, "sgm_mi355x (tiles): more than %d receives in one group of the local transport\n", (int)(sizeof x->pending / sizeof x->pending[0]));
        return -1;
    }
    x->pending[x->n_pending++] = (local_recv){buf, bytes, peer, stream};
    return 0;
}

/* sends of the group are queued by now; take the receives in the order they were listed */
static int local_group_end(void* c)
{
    local_ctx* x = (local_ctx*)c;
    sgm_tiles_local* g = x->g;
    for (int i = 0; i < x->n_pending; ++i) {
        const local_recv* r = &x->pending[i];
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_sec += 120;                                          /* a lost peer must not hang the caller for ever */
        pthread_mutex_lock(&g->mu);
        local_queue* q = &g->q[r->peer * g->world + x->rank];
        int waited = 0;
        while (!q->head && waited == 0) waited = pthread_cond_timedwait(&g->cv, &g->mu, &until);
        local_msg* m = q->head;
        if (m) { q->head = m->next; if (!q->head) q->tail = NULL; }
        pthread_mutex_unlock(&g->mu);
        if (!m) { fprintf(stderr, "sgm_mi355x (tiles): rank %d waited 120 s for a message of rank %d\n", x->rank, r->peer); return -1; }
        int rc = m->bytes == r->bytes ? 0 : -1;
        if (rc != 0) fprintf(stderr, "sgm_mi355x (tiles): rank %d expected %zu bytes from rank %d, got %zu\n", x->rank, r->bytes, r->peer, m->bytes);
        if (rc == 0) rc = sgmd_stream_wait_event(g->device, r->stream, m->ev_sent);
        if (rc == 0) rc = sgmd_d2d_async(g->device, r->stream, r->buf, m->stage, r->bytes);
        if (rc == 0) rc = sgmd_event_record(g->device, m->ev_taken, r->stream);
        if (rc != 0) m->used_before = false;                          /* no `taken` event behind this use: the next sender must not wait for one */
        pthread_mutex_lock(&g->mu);
        m->next = g->pool[r->peer];
        g->pool[r->peer] = m;
        pthread_mutex_unlock(&g->mu);
        if (rc != 0) return -1;
    }
    x->n_pending = 0;
    return 0;
}

static void local_view_destroy(void* c) { free(c); }

bool sgm_tiles_local_transport(sgm_tiles_local* group, int rank, sgm_tiles_transport* out)
{
    if (!group || !out || rank < 0 || rank >= group->world) return false;
    local_ctx* x = (local_ctx*)calloc(1, sizeof *x);
    if (!x) return false;
    x->g = group; x->rank = rank;
    out->ctx = x;
    out->group_start = local_group_start; out->group_end = local_group_end;
    out->send = local_send; out->recv = local_recv_post; out->destroy = local_view_destroy;
    return true;
}
