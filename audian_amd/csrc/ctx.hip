// Context, error reporting, memory and event helpers of libhip_dsp.
#include "common.h"
#include <map>
#include <unordered_map>
#include <vector>

static thread_local char g_err[512] = "";

// Stream-ordered cache of freed device blocks.  hipMalloc/hipFree synchronise the device, which
// stalls an interactive loop that needs a temporary per redraw (a screen-resolution image, a
// min/max trace); blocks up to `max_block` bytes go back into a size-ordered free list instead
// and are handed out again for requests they fit without wasting more than a quarter.  A freed
// block carries an event recorded on the stream it was freed on; reuse on the same stream needs
// nothing (whatever still uses the block is ahead of its next user in that stream), reuse after
// hipdsp_ctx_set_stream makes the new stream wait for that event first.
struct hd_block {
    void *ptr;
    hipStream_t stream;        // the context's stream when the block was freed
    hipEvent_t freed;          // recorded on that stream at the free (NULL: freed during capture)
};

struct hd_pool {
    std::multimap<size_t, hd_block> cached;          // size -> block
    std::unordered_map<void *, size_t> live;         // handed out by hipdsp_malloc
    std::vector<hipEvent_t> spare_events;
    size_t cached_bytes = 0;
    size_t limit = (size_t)1 << 30;                  // bytes kept at most ("pool_limit_mb")
    size_t max_block = (size_t)256 << 20;            // larger blocks are never cached
    unsigned long long hits = 0, misses = 0;
};

static void pool_trim(hd_pool *p)
{
    for (auto &kv : p->cached) {
        (void)hipFree(kv.second.ptr);
        if (kv.second.freed) p->spare_events.push_back(kv.second.freed);
    }
    p->cached.clear();
    p->cached_bytes = 0;
}

static void pool_destroy_events(hd_pool *p)
{
    for (hipEvent_t e : p->spare_events) (void)hipEventDestroy(e);
    p->spare_events.clear();
}

void hipdsp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int hipdsp_version(void) { return HIPDSP_VERSION; }

const char *hipdsp_last_error(void) { return g_err; }

int hipdsp_device_count(int *count)
{
    HD_REQUIRE(count != nullptr, "count is NULL");
    HD_CHECK_HIP(hipGetDeviceCount(count));
    return HIPDSP_OK;
}

int hipdsp_ctx_create(int device, void *stream, hipdsp_ctx **out)
{
    HD_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    HD_CHECK_HIP(hipGetDeviceCount(&n));
    HD_REQUIRE(device >= 0 && device < n, "device %d out of range (%d devices)", device, n);
    HD_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    HD_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        hipdsp_set_error("device %d is %s; libhip_dsp is built for gfx950 (MI355X) only",
                         device, prop.gcnArchName);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    hipdsp_ctx *ctx = new hipdsp_ctx();
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    ctx->max_segments = 0;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    ctx->mid_event = nullptr;
    for (int i = 0; i < 20; i++) ctx->fft_tables[i] = nullptr;
    ctx->force_generic_fft = 0;
    ctx->spec_fpw = ctx->spec_kernel = 0;
    ctx->chain_debug = 0;
    ctx->sos_no_pin = 0;
    ctx->sos_trace = nullptr;
    ctx->sos_fair = 0;
    ctx->chain_reserve_cus = 0;
    ctx->chain_split_frames = 0;
    ctx->sos_waves_per_cu = 0;
    ctx->sos_waves_min = 0;
    ctx->sos_trace_rows = 0;
    ctx->sos_single_wave_wg = 0;
    ctx->sos_debug = 0;
    ctx->chain_pairs = 0;
    ctx->spec_no_half = 0;
    ctx->pool = new hd_pool();
    ctx->sos_prefetch = 1;
    for (int i = 0; i < 20; i++) ctx->fft_tables2[i] = nullptr;
    ctx->fault_host = nullptr;
    ctx->fault_dev = nullptr;
    ctx->graphs_alive = 0;
    ctx->seg_flags = nullptr;
    ctx->seg_flags_cap = 0;
    {
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped);
        if (e == hipSuccess) {
            memset(h, 0, 64);
            e = hipHostGetDevicePointer(&d, h, 0);
        }
        if (e != hipSuccess) {
            if (h) (void)hipHostFree(h);
            delete ctx->pool;
            delete ctx;
            HD_CHECK_HIP(e);
        }
        ctx->fault_host = (volatile int *)h;
        ctx->fault_dev = (int *)d;
    }
    {
        void *f = nullptr;
        const hipError_t e = hipMalloc(&f, (size_t)1 << 20);
        if (e != hipSuccess) {
            (void)hipHostFree((void *)ctx->fault_host);
            delete ctx->pool;
            delete ctx;
            HD_CHECK_HIP(e);
        }
        ctx->seg_flags = (unsigned char *)f;
        ctx->seg_flags_cap = (size_t)1 << 20;
    }
    *out = ctx;
    return HIPDSP_OK;
}

int hipdsp_ctx_destroy(hipdsp_ctx *ctx)
{
    if (!ctx) return HIPDSP_OK;
    if (ctx->pool) {
        pool_trim(ctx->pool);
        pool_destroy_events(ctx->pool);
        delete ctx->pool;
    }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->seg_flags) (void)hipFree(ctx->seg_flags);
    if (ctx->fault_host) (void)hipHostFree((void *)ctx->fault_host);
    for (int i = 0; i < 20; i++)
        if (ctx->fft_tables[i]) (void)hipFree(ctx->fft_tables[i]);
    for (int i = 0; i < 20; i++)
        if (ctx->fft_tables2[i]) (void)hipFree(ctx->fft_tables2[i]);
    delete ctx;
    return HIPDSP_OK;
}

int hipdsp_ctx_set_stream(hipdsp_ctx *ctx, void *stream)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return HIPDSP_OK;
}

int hipdsp_ctx_synchronize(hipdsp_ctx *ctx)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return hd_device_fault(ctx);
}

int hipdsp_ctx_set_max_segments(hipdsp_ctx *ctx, int max_segments)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(max_segments >= 0, "max_segments must be >= 0");
    ctx->max_segments = max_segments;
    return HIPDSP_OK;
}

int hipdsp_ctx_set_option(hipdsp_ctx *ctx, const char *name, long long value)
{
    HD_REQUIRE(ctx != nullptr && name != nullptr, "NULL argument");
    if (strcmp(name, "max_segments") == 0) return hipdsp_ctx_set_max_segments(ctx, (int)value);
    if (strcmp(name, "force_generic_fft") == 0) { ctx->force_generic_fft = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "spec_fpw") == 0) { ctx->spec_fpw = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "sos_waves_per_cu") == 0) { ctx->sos_waves_per_cu = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "sos_waves_min") == 0) { ctx->sos_waves_min = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "chain_pairs") == 0) {
        HD_REQUIRE(value >= 0 && value <= 8, "chain_pairs %lld not in [0, 8]", value);
        ctx->chain_pairs = (int)value;
        return HIPDSP_OK;
    }
    if (strcmp(name, "sos_debug") == 0) { ctx->sos_debug = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "sos_single_wave_wg") == 0) { ctx->sos_single_wave_wg = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "sos_prefetch") == 0) { ctx->sos_prefetch = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "pool_limit_mb") == 0) {
        HD_REQUIRE(value >= 0, "pool_limit_mb must be >= 0");
        ctx->pool->limit = (size_t)value << 20;
        // a limit raised beyond the default also lifts the largest block that is cached (256 MiB by default: the
        // big temporaries of a batch-sized hipdsp_envelope_multi cost a second of hipMalloc + hipFree per call otherwise)
        ctx->pool->max_block = ctx->pool->limit > ((size_t)256 << 20) ? ctx->pool->limit : ((size_t)256 << 20);
        if (ctx->pool->cached_bytes > ctx->pool->limit) {
            HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            pool_trim(ctx->pool);
        }
        return HIPDSP_OK;
    }
    if (strcmp(name, "spec_no_half") == 0) { ctx->spec_no_half = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "spec_kernel") == 0) { ctx->spec_kernel = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "spec_debug") == 0) { ctx->spec_debug = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "chain_debug") == 0) { ctx->chain_debug = (int)value; return HIPDSP_OK; }
    if (strcmp(name, "sos_no_pin") == 0) { ctx->sos_no_pin = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "sos_split") == 0) {
#ifdef HIPDSP_WITH_ENVSPLIT
        ctx->sos_split = value != 0;
        return HIPDSP_OK;
#else
        if (value == 0) return HIPDSP_OK;
        hipdsp_set_error("\"sos_split\": this library was built without envsplit.hip (make SPLIT=1)");
        return HIPDSP_ERR_UNSUPPORTED;
#endif
    }
    if (strcmp(name, "sos_fair") == 0) { ctx->sos_fair = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "sos_trace_rows") == 0) { ctx->sos_trace_rows = value; return HIPDSP_OK; }
    if (strcmp(name, "sos_trace") == 0) {
        HD_REQUIRE(value == 0 || ctx->sos_trace_rows > 0, "set \"sos_trace_rows\" (capacity of the buffer in rows of 9 int64) first");
        ctx->sos_trace = reinterpret_cast<long long *>((uintptr_t)value);
        return HIPDSP_OK;
    }
    if (strcmp(name, "chain_split_frames") == 0) { ctx->chain_split_frames = value != 0; return HIPDSP_OK; }
    if (strcmp(name, "chain_reserve_cus") == 0) {
        HD_REQUIRE(value >= 0 && value < ctx->n_cus, "chain_reserve_cus %lld not in [0, %d)", value, ctx->n_cus);
        ctx->chain_reserve_cus = (int)value;
        return HIPDSP_OK;
    }
    if (strcmp(name, "n_cus") == 0) {                 // tests: pretend a smaller chip (planning only)
        HD_REQUIRE(value >= 1 && value <= 1024, "n_cus out of range");
        HD_REQUIRE(value > ctx->chain_reserve_cus, "n_cus %lld would leave no CU beside the %d of \"chain_reserve_cus\"", value,
                   ctx->chain_reserve_cus);
        ctx->n_cus = (int)value;
        return HIPDSP_OK;
    }
    hipdsp_set_error("unknown option '%s'", name);
    return HIPDSP_ERR_INVALID;
}

int hipdsp_ctx_set_mid_event(hipdsp_ctx *ctx, void *event)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->mid_event = (hipEvent_t)event;
    return HIPDSP_OK;
}

int hipdsp_ctx_reserve(hipdsp_ctx *ctx, size_t bytes)
{
    void *p;
    return hipdsp_scratch(ctx, bytes, &p);
}

int hipdsp_malloc(hipdsp_ctx *ctx, size_t bytes, void **dptr)
{
    HD_REQUIRE(ctx != nullptr && dptr != nullptr, "NULL argument");
    *dptr = nullptr;
    if (bytes == 0) return HIPDSP_OK;
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hd_pool *p = ctx->pool;
    const size_t want = (bytes + 511) & ~(size_t)511;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &capturing);
    auto it = p->cached.lower_bound(want);
    // A block freed on ANOTHER stream needs an order between the two: an event wait -- which must not be issued into
    // a stream capture (the event was recorded outside it) and does not exist for a block that was freed DURING a
    // capture (no event could be recorded then).  Such candidates are passed over.
    while (it != p->cached.end() && it->first <= want + want / 4 + 4096 && it->second.stream != ctx->stream &&
           (it->second.freed == nullptr || capturing != hipStreamCaptureStatusNone))
        ++it;
    if (it != p->cached.end() && it->first <= want + want / 4 + 4096) {
        const hd_block b = it->second;
        if (b.freed) {
            // freed on another stream than the one that will use it now: order the two
            if (b.stream != ctx->stream) HD_CHECK_HIP(hipStreamWaitEvent(ctx->stream, b.freed, 0));
            p->spare_events.push_back(b.freed);
        }
        *dptr = b.ptr;
        p->live[*dptr] = it->first;
        p->cached_bytes -= it->first;
        p->cached.erase(it);
        p->hits++;
        return HIPDSP_OK;
    }
    hipError_t e = hipMalloc(dptr, want);
    if (e == hipErrorOutOfMemory && !p->cached.empty()) {
        (void)hipGetLastError();
        pool_trim(p);                                  // give the cache back and try once more
        e = hipMalloc(dptr, want);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        hipdsp_set_error("hipMalloc(%zu bytes) out of memory", bytes);
        return HIPDSP_ERR_NOMEM;
    }
    HD_CHECK_HIP(e);
    p->live[*dptr] = want;
    p->misses++;
    return HIPDSP_OK;
}

// Up to `tries` blocks are allocated side by side, a hipMemsetAsync over each is timed, the fastest stays.  Where an
// allocation lies in HBM decides how fast a WRITE stream into it runs (profiles/r03_placement_probe.log: three loose
// classes, whatever the size, the alignment or the offset inside the block; profiles/r05be_placement_shop.log: memset 2.28 /
// 2.35 / 2.40 ms over 14.7 GB <-> the envelope's backward sweep 5.50 / 5.68 / 6.15 ms into the same block, correlation
// 0.94 over six blocks) -- the whole run-to-run spread of that sweep in the bench lines of rounds 1-4.
int hipdsp_malloc_probed(hipdsp_ctx *ctx, size_t bytes, int tries, void **dptr)
{
    HD_REQUIRE(ctx != nullptr && dptr != nullptr, "NULL argument");
    *dptr = nullptr;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &capturing);
    if (tries > 8) tries = 8;
    // small blocks (the differences are a few per cent of a stream over the block) and captures: a plain allocation
    if (tries <= 1 || bytes < ((size_t)64 << 20) || capturing != hipStreamCaptureStatusNone) return hipdsp_malloc(ctx, bytes, dptr);
    void *cand[8];
    float ms[8];
    int n = 0;
    for (; n < tries; n++) {
        const int rc = hipdsp_malloc(ctx, bytes, &cand[n]);
        if (rc != HIPDSP_OK) {
            if (n == 0) return rc;
            (void)hipGetLastError();
            break;                                     // out of memory on the way: choose among what there is
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    int best = 0;
    if (e == hipSuccess) {
        for (int i = 0; i < n && e == hipSuccess; i++) {
            e = hipMemsetAsync(cand[i], 0, bytes, ctx->stream);                       // (first touch)
            if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
            for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipMemsetAsync(cand[i], 0, bytes, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            ms[i] = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms[i], e0, e1);
            if (e == hipSuccess && ms[i] < ms[best]) best = i;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (int i = 0; i < n; i++)
        if (i != best) (void)hipdsp_free(ctx, cand[i]);
    if (e != hipSuccess) {
        (void)hipdsp_free(ctx, cand[best]);
        HD_CHECK_HIP(e);
    }
    *dptr = cand[best];
    return HIPDSP_OK;
}

int hipdsp_free(hipdsp_ctx *ctx, void *dptr)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!dptr) return HIPDSP_OK;
    hd_pool *p = ctx->pool;
    auto it = p->live.find(dptr);
    if (it == p->live.end()) {                         // not one of ours (or of another context)
        HD_CHECK_HIP(hipFree(dptr));
        return HIPDSP_OK;
    }
    const size_t size = it->second;
    p->live.erase(it);
    if (size <= p->max_block && p->cached_bytes + size <= p->limit) {
        hd_block b{dptr, ctx->stream, nullptr};
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
        if (st == hipStreamCaptureStatusNone) {
            hipError_t e = hipSuccess;
            if (!p->spare_events.empty()) {
                b.freed = p->spare_events.back();
                p->spare_events.pop_back();
            } else {
                e = hipEventCreateWithFlags(&b.freed, hipEventDisableTiming);
                if (e != hipSuccess) b.freed = nullptr;
            }
            if (e == hipSuccess) e = hipEventRecord(b.freed, ctx->stream);
            if (e != hipSuccess) {
                // no event to order a later owner behind this stream: the block goes back to the driver instead
                // of being leaked or cached without an order
                (void)hipGetLastError();
                if (b.freed) p->spare_events.push_back(b.freed);
                HD_CHECK_HIP(hipFree(dptr));
                return HIPDSP_OK;
            }
        }
        p->cached.emplace(size, b);
        p->cached_bytes += size;
        return HIPDSP_OK;
    }
    HD_CHECK_HIP(hipFree(dptr));
    return HIPDSP_OK;
}

int hipdsp_pool_stats(hipdsp_ctx *ctx, size_t *cached_bytes, uint64_t *hits, uint64_t *misses)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (cached_bytes) *cached_bytes = ctx->pool->cached_bytes;
    if (hits) *hits = ctx->pool->hits;
    if (misses) *misses = ctx->pool->misses;
    return HIPDSP_OK;
}

int hipdsp_pool_trim(hipdsp_ctx *ctx)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    pool_trim(ctx->pool);
    // ... and the scratch, which only ever grows (hipdsp_envelope_multi leaves two slabs of the trace's size there:
    // tens of GB at BASELINE configs[2]) -- unless a captured graph still points into it
    if (ctx->scratch && ctx->graphs_alive == 0) {
        HD_CHECK_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        ctx->sweep_frames = -1;
    }
    return HIPDSP_OK;
}

int hipdsp_memset(hipdsp_ctx *ctx, void *dptr, int value, size_t bytes)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (bytes) HD_CHECK_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return HIPDSP_OK;
}

int hipdsp_memcpy_h2d(hipdsp_ctx *ctx, void *dst, const void *host_src, size_t bytes)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (bytes) {
        HD_CHECK_HIP(hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // host buffer may be pageable
    }
    return HIPDSP_OK;
}

int hipdsp_memcpy_d2h(hipdsp_ctx *ctx, void *host_dst, const void *src, size_t bytes)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (bytes) {
        HD_CHECK_HIP(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        return hd_device_fault(ctx);      // what was copied may come from a kernel that gave up
    }
    return HIPDSP_OK;
}

int hipdsp_memcpy_d2d(hipdsp_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (bytes) HD_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return HIPDSP_OK;
}

int hipdsp_memcpy2d_d2d(hipdsp_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch,
                        size_t width, size_t height)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (width == 0 || height == 0) return HIPDSP_OK;
    HD_REQUIRE(dst != nullptr && src != nullptr, "NULL data pointer");
    HD_REQUIRE(dst_pitch >= width && src_pitch >= width, "pitch smaller than width");
    HD_CHECK_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, height, hipMemcpyDeviceToDevice,
                                  ctx->stream));
    return HIPDSP_OK;
}

int hipdsp_stream_create(hipdsp_ctx *ctx, void **stream)
{
    HD_REQUIRE(ctx != nullptr && stream != nullptr, "NULL argument");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t s;
    HD_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return HIPDSP_OK;
}

int hipdsp_stream_destroy(hipdsp_ctx *ctx, void *stream)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (stream) {
        if (ctx->stream == (hipStream_t)stream) ctx->stream = nullptr;
        HD_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    }
    return HIPDSP_OK;
}

int hipdsp_graph_begin(hipdsp_ctx *ctx)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    HD_REQUIRE(ctx->stream != nullptr, "stream capture needs a non-default stream "
               "(hipdsp_stream_create + hipdsp_ctx_set_stream)");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    HD_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return HIPDSP_OK;
}

int hipdsp_graph_end(hipdsp_ctx *ctx, hipdsp_graph **out)
{
    HD_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    *out = nullptr;
    hipGraph_t g = nullptr;
    HD_CHECK_HIP(hipStreamEndCapture(ctx->stream, &g));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        HD_CHECK_HIP(e);
    }
    hipdsp_graph *h = new hipdsp_graph();
    h->graph = g;
    h->exec = exec;
    ctx->graphs_alive++;
    *out = h;
    return HIPDSP_OK;
}

int hipdsp_graph_launch(hipdsp_ctx *ctx, hipdsp_graph *graph)
{
    HD_REQUIRE(ctx != nullptr && graph != nullptr, "NULL argument");
    HD_CHECK_HIP(hipGraphLaunch(graph->exec, ctx->stream));
    return HIPDSP_OK;
}

int hipdsp_graph_destroy(hipdsp_ctx *ctx, hipdsp_graph *graph)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (graph) {
        (void)hipGraphExecDestroy(graph->exec);
        (void)hipGraphDestroy(graph->graph);
        delete graph;
        if (ctx->graphs_alive > 0) ctx->graphs_alive--;
    }
    return HIPDSP_OK;
}

int hipdsp_event_create(hipdsp_ctx *ctx, void **event)
{
    HD_REQUIRE(ctx != nullptr && event != nullptr, "NULL argument");
    hipEvent_t ev;
    HD_CHECK_HIP(hipEventCreate(&ev));
    *event = (void *)ev;
    return HIPDSP_OK;
}

int hipdsp_event_destroy(hipdsp_ctx *ctx, void *event)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (event) HD_CHECK_HIP(hipEventDestroy((hipEvent_t)event));
    return HIPDSP_OK;
}

int hipdsp_event_record(hipdsp_ctx *ctx, void *event)
{
    HD_REQUIRE(ctx != nullptr && event != nullptr, "NULL argument");
    HD_CHECK_HIP(hipEventRecord((hipEvent_t)event, ctx->stream));
    return HIPDSP_OK;
}

int hipdsp_event_wait(hipdsp_ctx *ctx, void *event)
{
    HD_REQUIRE(ctx != nullptr && event != nullptr, "NULL argument");
    HD_CHECK_HIP(hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return HIPDSP_OK;
}

int hipdsp_event_elapsed_ms(hipdsp_ctx *ctx, void *start, void *stop, float *ms)
{
    HD_REQUIRE(ctx != nullptr && start && stop && ms, "NULL argument");
    HD_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    HD_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return hd_device_fault(ctx);
}

}  // extern "C"

int hd_device_fault(hipdsp_ctx *ctx)
{
    if (!ctx || !ctx->fault_host) return HIPDSP_OK;
    const int code = ctx->fault_host[0];
    if (code == 0) return HIPDSP_OK;
    const int a = ctx->fault_host[1], b = ctx->fault_host[2], c = ctx->fault_host[3];
    ctx->fault_host[0] = 0;                       // reported once
    if (code == HD_FAULT_CHAIN_HANDOVER)
        hipdsp_set_error("chain_fwd_kernel: a wave of workgroup %d (pair %d) gave up waiting for its partner's "
                         "LDS hand-over in iteration %d; the outputs of that launch are invalid", a, b, c);
    else
        hipdsp_set_error("a kernel reported device fault %d (%d, %d, %d)", code, a, b, c);
    return HIPDSP_ERR_HIP;
}

int hd_seg_flags(hipdsp_ctx *ctx, size_t units, unsigned char **out)
{
    if (units > ctx->seg_flags_cap) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
        if (st != hipStreamCaptureStatusNone || ctx->graphs_alive > 0) {
            hipdsp_set_error("a sweep of %zu (channel, segment) units during a stream capture or next to captured graphs of "
                             "this context: run one of that size once before capturing", units);
            return HIPDSP_ERR_INVALID;
        }
        HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->seg_flags) HD_CHECK_HIP(hipFree(ctx->seg_flags));
        ctx->seg_flags = nullptr;
        ctx->seg_flags_cap = 0;
        void *f = nullptr;
        const hipError_t e = hipMalloc(&f, units);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            hipdsp_set_error("hipMalloc(%zu bytes) for the segment flags: out of memory", units);
            return HIPDSP_ERR_NOMEM;
        }
        HD_CHECK_HIP(e);
        ctx->seg_flags = (unsigned char *)f;
        ctx->seg_flags_cap = units;
    }
    *out = ctx->seg_flags;
    return HIPDSP_OK;
}

int hd_scratch_parked(hipdsp_ctx *ctx, size_t bytes, void **out)
{
    HD_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    if (ctx->sweep_frames < 0 || ctx->scratch == nullptr || bytes > ctx->scratch_bytes) {
        hipdsp_set_error("no forward sweep's tile states are parked in this context's scratch (another call has used the "
                         "scratch since, or none has run): run the forward sweep again");
        return HIPDSP_ERR_INVALID;
    }
    *out = ctx->scratch;
    return HIPDSP_OK;
}

int hipdsp_scratch(hipdsp_ctx *ctx, size_t bytes, void **out)
{
    HD_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
    // Whoever asks for the scratch is about to write it: tile states a forward sweep parked there are gone (ADVICE
    // round 4).  The forward sweeps note their grid again right after this call (hd_note_sweep); the backward sweeps,
    // which READ the parked states, take the pointer through hd_scratch_parked() instead.
    ctx->sweep_frames = -1;
    if (bytes > ctx->scratch_bytes) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->stream) (void)hipStreamIsCapturing(ctx->stream, &st);
        if (st != hipStreamCaptureStatusNone) {
            hipdsp_set_error("scratch of %zu bytes needed during stream capture; call "
                             "hipdsp_ctx_reserve() before capturing", bytes);
            return HIPDSP_ERR_INVALID;
        }
        if (ctx->scratch && ctx->graphs_alive > 0) {
            // a captured launch holds the old scratch pointer (envelope checkpoints, FFT work area)
            hipdsp_set_error("scratch would have to grow from %zu to %zu bytes while %d captured graph(s) of this "
                             "context still point into it; hipdsp_ctx_reserve() the largest size before capturing",
                             ctx->scratch_bytes, bytes, ctx->graphs_alive);
            return HIPDSP_ERR_INVALID;
        }
        HD_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) HD_CHECK_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        hipError_t e = hipMalloc(&ctx->scratch, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            hipdsp_set_error("scratch hipMalloc(%zu bytes) out of memory", bytes);
            return HIPDSP_ERR_NOMEM;
        }
        HD_CHECK_HIP(e);
        ctx->scratch_bytes = bytes;
    }
    *out = ctx->scratch;
    return HIPDSP_OK;
}
