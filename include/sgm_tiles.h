/*
 * sgm_tiles.h -- ONE frame over the GPUs of a node in row tiles, frames in flight: the C host of the multi-GPU path of
 * libsgm_mi355x.so (plain C on HIP and RCCL; no Python, no PyTorch on the path).
 *
 * The reference has no multi-GPU code at all ("no NCCL/MPI call site"); its intended call site for the
 * matcher is C -- ZedBoard/Vitis/lwip_tcp_perf_client/src/stereo_matching.c:34-40, start_stereo_matching() -- and this is
 * what such a caller uses to spread a frame over more than one MI355X.  What is split, and why it is a pipeline and not a
 * halo stencil, is described with the sgm_tile_* entry points in sgm_mi355x.h (row tiles) and in DESIGN.md section 6:
 * rank r computes rows [r0, r1) of every frame; the vertical and diagonal paths (SemiGlobalMatching.c:229-372, six of the
 * eight directions of .c:213-220) cross the tile borders, so each of the two vertical sweeps hands one image row of path
 * costs per direction from rank to rank, and with several frames in flight the ranks work as a systolic pipeline.
 *
 * Three layers, each usable on its own:
 *
 *   1. the STEP SCHEDULE (sgm_tile_step): which operation a rank performs in which step, as calls on an engine vtable.  No
 *      device code.  tiling.py's TilePipeline drives it with Python engines (its CPU tests over gloo run this very code).
 *   2. the DEVICE ENGINE + PIPELINE (sgm_tiles_*): slots = sgm_instances restricted to the rank's rows, one HIP stream each,
 *      a communication stream, HIP events in between, hand-over and row-gather buffers; sgm_tiles_submit(frame) queues one
 *      step and returns, nothing blocks the host except the bound on its run-ahead.
 *   3. the TRANSPORT (sgm_tiles_transport): grouped point-to-point sends and receives on a stream.  The product transport is
 *      RCCL over xGMI (sgm_tiles_rccl_*: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, bound at run time from the
 *      librccl the process already has or can load); sgm_tiles_local_* connects N pipelines that are threads of one process on
 *      one GPU (device copies ordered by events) for rehearsals and tests.
 */
#ifndef SGM_TILES_H
#define SGM_TILES_H

#include "sgm_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ 1. the step schedule */

enum { SGM_XOP_SEND = 0, SGM_XOP_RECV = 1 };
enum { SGM_XBUF_BOUNDARY = 0, SGM_XBUF_ROWS = 1 };

/* one operation of a step's grouped exchange */
typedef struct {
    int kind;                 /* SGM_XOP_SEND / SGM_XOP_RECV */
    int buf;                  /* SGM_XBUF_BOUNDARY: a slot's hand-over buffer;  SGM_XBUF_ROWS: rows of a slot's disparity map(s) */
    int slot;
    int forward;              /* boundary: the sweep it belongs to (1 = top to bottom) */
    int incoming;             /* boundary: the slot's incoming (1, filled by a receive) or outgoing (0, filled by an export) buffer */
    int row_begin, row_end;   /* rows: [row_begin, row_end) of every map of the slot */
    int peer;                 /* the other rank */
} sgm_tile_xop;

/* what a rank's engine offers; every function returns 0 on success (anything else ends the step with that value) */
typedef struct {
    void* user;
    int (*begin)(void* user, int slot, long frame);                    /* census, horizontal paths of the tile, anomalous lines */
    int (*import_boundary)(void* user, int slot, int forward);         /* incoming hand-over buffer -> the planes */
    int (*sweep)(void* user, int slot, int forward);                   /* the three directions of one vertical sense on the tile */
    int (*export_boundary)(void* user, int slot, int forward);         /* last row of the sweep -> outgoing hand-over buffer */
    int (*exchange)(void* user, const sgm_tile_xop* ops, int n_ops, const int* slots, int n_slots);   /* ONE group; slots = those touched */
    int (*finish)(void* user, int slot);                               /* cost sum, both WTAs, LR check on the tile's rows */
    int (*post)(void* user, int slot, long frame);                     /* owner rank: speckle + median on the gathered map */
} sgm_tile_engine;

/* rows [*r0, *r1) of rank `rank`: contiguous, balanced (the first height % world ranks get one row more).  false if height < world */
bool sgm_tile_rows(int height, int world, int rank, int* r0, int* r1);
/* a frame occupies its slot from tile_begin (step f) to the post pass (step f + lead + world + 1; one rank: f + lead) */
int  sgm_tile_slots_needed(int world, int lead);
/* Device memory one slot of a rank that owns rows [row_begin, row_end) takes: per frame of its batch the 8 direction planes of the tile's rows + one hand-over row
 * either side (1 B per cell of the padded disparity range), ~64 B per pixel of the WHOLE frame (maps, census, labels, median
 * scratch, the row-gather buffer) and 4 hand-over buffers.  A rank holds sgm_tile_slots_needed(world, lead) [+ spare - 1] of them:
 * sgm_tiles_create and tiling.DeviceSlotEngine refuse a configuration that does not fit (DESIGN.md section 6 has the table).
 * 0 for an empty or out-of-frame row range. */
size_t sgm_tile_slot_bytes(int row_begin, int row_end, uint16_t width, uint16_t height, const SGMOption* option, int batch);
/* steps a stream of n_frames takes until its last result is queued */
long sgm_tile_steps_total(long n_frames, int world, int lead);
/* Step `step` (0, 1, 2, ...) of rank `rank`, `frames_known` = how many frames exist so far (a frame may be begun in the step
 * with its own index; once the stream has ended, the total).  With s = step - lead, in this order:
 *     begin            of frame step
 *     forward sweep    of frame s - rank               (import first unless rank 0, export afterwards unless rank N-1)
 *     backward sweep   of frame s - (N-1-rank)         (import first unless rank N-1, export afterwards unless rank 0)
 *     ONE exchange:    forward hand-over to rank+1, receive of the next one from rank-1, backward hand-over to rank-1, receive
 *                      from rank+1, and the rows of frame s - N (finished on every rank in an earlier step) to its owner
 *                      (s - N) mod N -- queued BEFORE this step's finish: the next step's sweeps wait for the exchange
 *     finish           of frame s - max(rank, N-1-rank)
 *     post             of frame s - N - 1 on its owner rank (one rank: of frame s, nothing to gather)
 * Every rank issues the same step sequence, so each send meets its receive in the same exchange.  Returns 0, or the first
 * non-zero value an engine function returned. */
int  sgm_tile_step(const sgm_tile_engine* e, int rank, int world, int height, int slots, int lead, long step, long frames_known);

/* ------------------------------------------------------------------------------------------ 3. transports */

/* Grouped point-to-point operations on a HIP stream (hipStream_t as void*); every function returns 0 on success.  Operations
 * between group_start and group_end progress together (an RCCL group): the sends and receives of neighbouring ranks
 * cannot deadlock on each other's order. */
typedef struct {
    void* ctx;
    int  (*group_start)(void* ctx);
    int  (*send)(void* ctx, const void* d_buf, size_t bytes, int peer, void* stream);
    int  (*recv)(void* ctx, void* d_buf, size_t bytes, int peer, void* stream);
    int  (*group_end)(void* ctx);
    void (*destroy)(void* ctx);
} sgm_tiles_transport;

/* RCCL: rank 0 makes an id (SGM_TILES_ID_BYTES bytes), the caller moves it to every rank by whatever means it has (a file, a
 * socket, MPI, torch.distributed), every rank then joins.  `device` must be the rank's GPU.  The functions bind ncclGetUniqueId,
 * ncclCommInitRank, ncclGroupStart/End, ncclSend, ncclRecv, ncclCommDestroy from librccl at run time (dlopen: the copy the
 * process already uses if there is one); false + a message on stderr if that fails.  A world of ONE rank sends to itself
 * (what a one-GPU box can exercise of this transport). */
#define SGM_TILES_ID_BYTES 128
bool sgm_tiles_rccl_unique_id(void* id_out);
bool sgm_tiles_rccl_transport(const void* id, int rank, int world, int device, sgm_tiles_transport* out);

/* Threads of one process as ranks (one GPU): a send is a device copy into a staging buffer queued to the peer with a HIP event;
 * the receive waits for the event on the receiver's stream and copies into its buffer.  group = sgm_tiles_local_group(world);
 * every thread takes its own view.  Sends of a group are queued before its receives are waited for, like a grouped RCCL
 * exchange.  The group outlives its views; destroy it after the last pipeline that used it. */
typedef struct sgm_tiles_local sgm_tiles_local;
sgm_tiles_local* sgm_tiles_local_group(int world, int device);
bool sgm_tiles_local_transport(sgm_tiles_local* group, int rank, sgm_tiles_transport* out);
void sgm_tiles_local_destroy(sgm_tiles_local* group);

/* ------------------------------------------------------------------------------------------ 2. the pipeline of one rank */

typedef struct sgm_tiles sgm_tiles;

/* result of frame `frame` on its owner rank: d_map = [batch][H][W] float32 in the slot (valid once `hip_event` -- a hipEvent_t,
 * recorded behind the post pass -- has completed, and until the slot is reused `spare` steps later); called from
 * sgm_tiles_submit / sgm_tiles_finish on the calling thread */
typedef void (*sgm_tiles_result_fn)(void* user, long frame, const float* d_map, void* hip_event);

/* width x height frames (`batch` of them per step: every call covers the same tile of `batch` frames, images and maps
 * [batch][H][W]), options as for SGM_Initialize; lead = steps tile_begin is queued ahead of the frame's first sweep (2 is a good
 * value); spare >= 1 = slots beyond the schedule's need (how long a result stays readable);
 * throttle = how many steps the host may run ahead of the GPU (0: unbounded).  The transport is used, not owned. */
sgm_tiles* sgm_tiles_create(int device, int rank, int world, uint16_t width, uint16_t height, const SGMOption* option, int batch,
                            int lead, int spare, int throttle, const sgm_tiles_transport* transport);
void       sgm_tiles_destroy(sgm_tiles* t);
void       sgm_tiles_set_honor_num_paths(sgm_tiles* t, int honor);          /* before the first submit */
void       sgm_tiles_on_result(sgm_tiles* t, sgm_tiles_result_fn fn, void* user);
/* Alternatively the library keeps the results: d_ring = [ring_frames][batch][H][W] floats of device memory on the owner; the
 * map of owned frame f is copied (device to device, behind its post pass) to entry (f / world) % ring_frames. */
void       sgm_tiles_result_ring(sgm_tiles* t, float* d_ring, int ring_frames);
/* Next frame of the stream (every rank calls this with the same frames in the same order; the images are the WHOLE frames,
 * device pointers, complete on entry -- or complete once `ready_event`, a hipEvent_t, has -- and untouched until the frame's
 * tile_finish, i.e. lead + world steps later).  Queues one step of the pipeline and returns. */
bool       sgm_tiles_submit(sgm_tiles* t, const uint8_t* d_left, const uint8_t* d_right, void* ready_event);
/* End of the stream: queues the remaining steps (world + 1 + lead of them) and waits for everything.  The pipeline can then
 * take a new stream. */
bool       sgm_tiles_finish(sgm_tiles* t);
/* rows [r0, r1) of this rank; slots it holds */
void       sgm_tiles_info(const sgm_tiles* t, int* r0, int* r1, int* slots);

#ifdef __cplusplus
}
#endif
#endif /* SGM_TILES_H */
