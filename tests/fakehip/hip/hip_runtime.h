/* A fake HIP runtime for the CPU sanitizer build of libhip_dsp's HOST-ONLY translation units (ctx.hip: context,
 * stream-ordered block cache, scratch, options; sos_plan.hip: plan mathematics, segment planner) -- test
 * infrastructure, tests/test_shim_sanitizers.py.  "Device" memory is host memory, streams and events are small
 * heap objects, everything completes at once; stream capture is a flag per stream so that the block cache's
 * capture rules can be exercised.  Nothing of the product is built against this header. */
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <cstdint>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
struct fake_stream { int capturing; };
struct fake_event { int recorded; fake_stream *on; };
typedef fake_stream *hipStream_t;
typedef fake_event *hipEvent_t;
typedef struct fake_graph { int n; } *hipGraph_t;
typedef struct fake_graph_exec { int n; } *hipGraphExec_t;
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum hipStreamCaptureMode { hipStreamCaptureModeThreadLocal = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocMapped = 4, hipHostMallocDefault = 0 };

extern size_t fake_hip_limit;          /* bytes "device memory" may hold (0 = unlimited): out-of-memory path */
extern size_t fake_hip_in_use;
extern int fake_hip_fail_event_create; /* next hipEventCreate* calls fail this many times */
extern long fake_hip_waits;            /* hipStreamWaitEvent calls seen */

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : (e == hipErrorOutOfMemory ? "out of memory" : "error"); }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)calloc(1, sizeof(fake_stream)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st) { *st = (s && s->capturing) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { if (!s) return hipErrorInvalidValue; s->capturing = 1; return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) { s->capturing = 0; *g = (hipGraph_t)calloc(1, sizeof(fake_graph)); return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, void *, void *, size_t) { *e = (hipGraphExec_t)calloc(1, sizeof(fake_graph_exec)); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned);
static inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { if (s && s->capturing) return hipErrorInvalidValue; e->recorded = 1; e->on = s; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
/* waiting, inside a capture, for an event that was recorded outside it invalidates the capture on real HIP */
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { fake_hip_waits++; if (!e || !e->recorded) return hipErrorInvalidValue; if (s && s->capturing) return hipErrorInvalidValue; return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
