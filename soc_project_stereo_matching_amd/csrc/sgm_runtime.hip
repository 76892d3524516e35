#include "sgm_common.hpp"

// device, stream, memory and event-timer helpers behind sgm_device.h

extern "C" {

int sgmd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sgmd_device_is_gfx950(int ordinal)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return -1;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int sgmd_stream_create(int ord, void** stream)
{
    HIP_TRY(hipSetDevice(ord));
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return 0;
}
int sgmd_device_cus(int ord, int* cus_per_xcd, int* xcds)
{
    int cus = 0, x = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ord));
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, ord) != hipSuccess || x <= 0) {
        (void)hipGetLastError();
        x = 1;
    }
    if (cus <= 0 || cus % x != 0) return -1;
    *xcds = x;
    *cus_per_xcd = cus / x;
    return 0;
}
// CU mask bit i = CU (i / xcds) of XCD (i % xcds): the driver deals the bits of a queue's mask round-robin over the XCDs, so
// "count CUs of every XCD" is one contiguous run of bits.  Every XCD keeps its own L2 and its own path to HBM in the game.
int sgmd_stream_create_cus(int ord, void** stream, int first_per_xcd, int count_per_xcd)
{
    if (count_per_xcd <= 0) return sgmd_stream_create(ord, stream);
    int per = 0, xcds = 0;
    if (sgmd_device_cus(ord, &per, &xcds) != 0) return -1;
    if (first_per_xcd < 0 || first_per_xcd + count_per_xcd > per) {
        fprintf(stderr, "sgm_mi355x: CUs [%d, %d) of an XCD do not exist (%d per XCD)\n", first_per_xcd, first_per_xcd + count_per_xcd, per);
        return -1;
    }
    HIP_TRY(hipSetDevice(ord));
    uint32_t mask[32] = {0};
    const int total = per * xcds;
    if (total > 32 * 32) return -1;
    for (int b = first_per_xcd * xcds; b < (first_per_xcd + count_per_xcd) * xcds; ++b) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask));
    *stream = (void*)s;
    return 0;
}
// priority: 0 normal, < 0 higher, > 0 lower (clamped to the device's range): the dispatcher serves a higher-priority queue's
// waiting workgroups first whenever execution resources free up
int sgmd_stream_create_prio(int ord, void** stream, int priority)
{
    HIP_TRY(hipSetDevice(ord));
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));        // numerically: least >= greatest
    if (priority > least) priority = least;
    if (priority < greatest) priority = greatest;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
    *stream = (void*)s;
    return 0;
}
int sgmd_stream_destroy(int ord, void* stream)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int sgmd_stream_sync(int ord, void* stream)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int sgmd_event_create(int ord, void** event)
{
    HIP_TRY(hipSetDevice(ord));
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void*)e;
    return 0;
}
void sgmd_event_destroy(int ord, void* event)
{
    if (!event) return;
    (void)hipSetDevice(ord);
    (void)hipEventDestroy((hipEvent_t)event);
}
int sgmd_event_record(int ord, void* event, void* stream)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return 0;
}
int sgmd_stream_wait_event(int ord, void* stream, void* event)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}
int sgmd_event_sync(int ord, void* event)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return 0;
}
int sgmd_set_device(int ord)
{
    HIP_TRY(hipSetDevice(ord));
    return 0;
}
int sgmd_mem_info(int ord, size_t* free_bytes, size_t* total_bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}
int sgmd_alloc(int ord, void** dptr, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}
int sgmd_free(int ord, void* dptr)
{
    if (!dptr) return 0;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipFree(dptr));
    return 0;
}
int sgmd_alloc_pinned(int ord, void** hptr, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}
int sgmd_free_pinned(int ord, void* hptr)
{
    if (!hptr) return 0;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipHostFree(hptr));
    return 0;
}
int sgmd_host_is_pinned(int ord, const void* hptr, size_t bytes)
{
    if (!hptr || hipSetDevice(ord) != hipSuccess) return 0;
    const char* ends[2] = {(const char*)hptr, (const char*)hptr + (bytes ? bytes - 1 : 0)};
    for (int i = 0; i < 2; ++i) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ends[i]) != hipSuccess) {
            (void)hipGetLastError();                     // an ordinary malloc'd pointer: not an error of ours
            return 0;
        }
        if (at.type != hipMemoryTypeHost) return 0;
    }
    return 1;
}
int sgmd_h2d_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
int sgmd_d2h_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
struct PlaneRows { int dirs[8]; };
__global__ __launch_bounds__(256) void sgm_plane_rows_copy_k(uint8_t* planes, size_t plane_bytes, size_t row_offset, unsigned row_vecs,
                                                             PlaneRows pr, int ndirs, uint8_t* buf, int to_buf)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= row_vecs) return;
    const int k = blockIdx.y, f = blockIdx.z;
    uint4* cell = reinterpret_cast<uint4*>(planes + ((size_t)f * 8 + (size_t)pr.dirs[k]) * plane_bytes + row_offset) + i;
    uint4* slot = reinterpret_cast<uint4*>(buf + ((size_t)f * ndirs + k) * (size_t)row_vecs * 16) + i;
    if (to_buf) *slot = *cell;
    else *cell = *slot;
}

int sgmd_plane_rows_copy(int ord, void* stream, void* planes, size_t plane_bytes, size_t row_offset, size_t row_bytes, const int* dirs,
                         int ndirs, int frames, void* buf, int to_buf)
{
    HIP_TRY(hipSetDevice(ord));
    if (ndirs <= 0 || ndirs > 8 || frames <= 0 || row_bytes % 16 != 0 || plane_bytes % 16 != 0) return -1;   /* Dp is a multiple of 16 */
    PlaneRows pr;
    for (int k = 0; k < 8; ++k) pr.dirs[k] = k < ndirs ? dirs[k] : 0;
    const unsigned vecs = (unsigned)(row_bytes / 16);
    hipLaunchKernelGGL(sgm_plane_rows_copy_k, dim3((vecs + 255) / 256, ndirs, frames), dim3(256), 0, (hipStream_t)stream, (uint8_t*)planes,
                       plane_bytes, row_offset, vecs, pr, ndirs, (uint8_t*)buf, to_buf);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sgmd_d2d_async(int ord, void* stream, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int sgmd_d2d_2d_async(int ord, void* stream, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int sgmd_memset_async(int ord, void* stream, void* dst, int value, size_t bytes)
{
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return 0;
}

struct sgmd_timer { int n; hipEvent_t* ev; };

int sgmd_timer_create(int ord, void** timer, int max_marks)
{
    HIP_TRY(hipSetDevice(ord));
    sgmd_timer* t = new sgmd_timer;
    t->n = max_marks;
    t->ev = new hipEvent_t[max_marks];
    for (int i = 0; i < max_marks; ++i) HIP_TRY(hipEventCreate(&t->ev[i]));
    *timer = t;
    return 0;
}
void sgmd_timer_destroy(int ord, void* timer)
{
    if (!timer) return;
    (void)hipSetDevice(ord);
    sgmd_timer* t = (sgmd_timer*)timer;
    for (int i = 0; i < t->n; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev;
    delete t;
}
int sgmd_timer_mark(int ord, void* timer, void* stream, int index)
{
    sgmd_timer* t = (sgmd_timer*)timer;
    if (!t || index < 0 || index >= t->n) return -1;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventRecord(t->ev[index], (hipStream_t)stream));
    return 0;
}
int sgmd_timer_elapsed(int ord, void* timer, int from, int to, float* ms)
{
    sgmd_timer* t = (sgmd_timer*)timer;
    if (!t) return -1;
    HIP_TRY(hipSetDevice(ord));
    HIP_TRY(hipEventElapsedTime(ms, t->ev[from], t->ev[to]));
    return 0;
}

}  // extern "C"
