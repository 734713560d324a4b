/*
 * hip_dsp.h -- C ABI of libhip_dsp.so: MI355X (gfx950) kernels for audian's
 * BufferedData DSP hot path.
 *
 * The reference (bendalab/audian) is pure Python and has no FFI of its own; each
 * entry point below names the reference call (file:line under /root/reference)
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions across the boundary.
 *   - Every call returns an int status (HIPDSP_OK == 0); hipdsp_last_error()
 *     returns a thread-local message for the last failing call.
 *   - All data pointers are DEVICE pointers unless the name says "host"; buffers
 *     are caller-owned.  Work is enqueued on the context's HIP stream and is
 *     asynchronous; hipdsp_ctx_synchronize() waits for it.
 *   - Device-native layout is planar float32: a trace is (channels, frames) with
 *     a row pitch in elements; a spectrogram is (channels, frames', F) compact.
 *     The reference's layouts are time-major float64 (T, C) / (T', C, F)
 *     (src/audian/buffereddata.py:46-48,70); the pack/unpack entry points convert
 *     at the edge.
 *   - IIR coefficients and state are float64 (mandatory); HBM I/O is
 *     float32.
 */
#ifndef HIP_DSP_H
#define HIP_DSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPDSP_VERSION 102          /* 0.1.2 */

#define HIPDSP_OK               0
#define HIPDSP_ERR_INVALID      1   /* bad argument */
#define HIPDSP_ERR_HIP          2   /* HIP runtime failure (message has detail) */
#define HIPDSP_ERR_UNSUPPORTED  3   /* valid in the reference, not implemented here */
#define HIPDSP_ERR_TOO_SHORT    4   /* sosfiltfilt: frames <= padlen (scipy: ValueError) */
#define HIPDSP_ERR_NOMEM        5

#define HIPDSP_MAX_SECTIONS     4   /* second-order sections per plan (cascade longer ones) */

typedef struct hipdsp_ctx hipdsp_ctx;
typedef struct hipdsp_sosplan hipdsp_sosplan;

/* ---- library / context ------------------------------------------------- */

int hipdsp_version(void);
const char *hipdsp_last_error(void);
int hipdsp_device_count(int *count);

/* `stream` is a hipStream_t (NULL = the legacy default stream). */
int hipdsp_ctx_create(int device, void *stream, hipdsp_ctx **out);
int hipdsp_ctx_destroy(hipdsp_ctx *ctx);
int hipdsp_ctx_set_stream(hipdsp_ctx *ctx, void *stream);
int hipdsp_ctx_synchronize(hipdsp_ctx *ctx);
/* Tuning knob: upper bound on time segments per channel of the block-parallel
 * IIR (0 = automatic).  Results do not depend on it beyond fp64 rounding. */
int hipdsp_ctx_set_max_segments(hipdsp_ctx *ctx, int max_segments);
/* Named tuning/testing options (results do not depend on them beyond rounding):
 *   "max_segments"       as hipdsp_ctx_set_max_segments
 *   "sos_waves_per_cu"   most resident waves per CU the IIR segment planner uses (16; see hipdsp_sos_segments_host);
 *                        "sos_waves_min" >= that: exactly that many (experiments); "chain_pairs" (8): most pairs of
 *                        waves per CU of the fused sweeps
 *   "pool_limit_mb"      bytes (MiB) hipdsp_free may keep cached for hipdsp_malloc (1024); blocks of up to 256 MiB are
 *                        cached, with a larger limit blocks of up to the limit
 *   "sos_prefetch"       0: envelope sweeps without the register prefetch of the next tile (1)
 *   "chain_split_frames" non-zero: hipdsp_chain_forward (2048/1024, no db_out) leaves the odd frames to hipdsp_chain_backward
 *   "chain_reserve_cus"  CUs hipdsp_chain_forward plans no workgroup for (0): its 1024-thread workgroups want a
 *                        whole CU each, so a kernel that stays resident next to it (RCCL's all-gather in the
 *                        multi-GPU step) needs CUs of its own or a second round of workgroups forms
 *   "sos_no_pin"         non-zero: plan tables fetched by just-in-time scalar loads (A/B, tools/pin_ab.py)
 *   "sos_trace"          diagnostics: device address (0 = off) of 9 int64 per wave (= channel x segment) of the
 *                        envelope's backward sweep -- start and end of the wave in 100 MHz ticks, its HW_ID, 6 clock
 *                        sums (tools/sweep_trace.py); "sos_trace_rows" (set it first) is the buffer's capacity in such
 *                        rows, waves beyond it do not report
 *   "sos_fair"           0: the single-wave sweeps without rotating issue priorities (A/B)
 *   "force_generic_fft"  non-zero: every nfft takes the generic radix-2 / four-step kernels
 *   "spec_kernel"        cross-check paths (results equal within the parity bar): 0 = default per size, 2 = the kernel a
 *                        size's default replaced (two-stage FFT, workgroup per frame, four-step path through HBM), 3 = the
 *                        other of a size's two candidates (nfft 256, 512: one frame per lane group instead of the stream
 *                        through LDS; 1024: the stream for every hop; 4096: one wave per frame); tools/spec_kernel_ab.py
 *   "spec_fpw"           consecutive frames per wave (0 = automatic)
 *   "spec_no_half"       non-zero: do not reuse the overlapped half frame at 50 % overlap
 *   "chain_debug"        measurements only (results become wrong): 1 = the FFT waves of
 *                        hipdsp_chain_forward only copy their tiles, 2 = its IIR waves skip the cascades;
 *                        4 (results unchanged) = workgroup barriers instead of pairwise LDS flags;
 *                        8 = one FFT wave withholds one hand-over (test of the fault report below);
 *                        16 = clock counters of one wave into the first 16 bytes of the PSD;
 *                        32 = diagnostic build: db_out receives 16 clock sums per wave (tools/chain_stamps.py);
 *                        64 = no issue priorities at all; 128 (results unchanged) = FFT waves above IIR waves
 *                        only, without the progress words that keep the waves of a SIMD in step
 * Device-side faults: the waits between the waves of hipdsp_chain_forward's kernel are bounded; a wave
 * whose wait runs out writes a fault word owned by the context and ends the launch early.
 * hipdsp_ctx_synchronize, hipdsp_memcpy_d2h, hipdsp_event_elapsed_ms and the next
 * hipdsp_chain_forward / hipdsp_sosfilt_envelope on the context then return HIPDSP_ERR_HIP (once,
 * with a message): no path returns HIPDSP_OK for a launch that gave up. */
int hipdsp_ctx_set_option(hipdsp_ctx *ctx, const char *name, long long value);
/* Pre-size the internal scratch (envelope state checkpoints: 16 * n_sections bytes per
 * 2048-sample tile and channel, for ceil((frames + padlen) / 2048) + 1 tiles; four-step FFT work area) so that later calls do not
 * allocate; required before stream capture into a hipGraph.  While a graph captured on the context
 * is alive the scratch cannot grow (the graph holds its address): a call that would need more returns
 * HIPDSP_ERR_INVALID -- reserve the largest size before capturing. */
int hipdsp_ctx_reserve(hipdsp_ctx *ctx, size_t bytes);

/* ---- streams and hipGraph capture (interactive recompute, BASELINE configs[4]) ---- */

/* A non-blocking HIP stream owned by the library (stream capture cannot run on the
 * legacy default stream).  Pass it to hipdsp_ctx_set_stream. */
int hipdsp_stream_create(hipdsp_ctx *ctx, void **stream);
int hipdsp_stream_destroy(hipdsp_ctx *ctx, void *stream);

/* Capture everything enqueued on the context's stream between _begin and _end into an
 * executable graph; hipdsp_graph_launch replays it on the context's stream.  Run each
 * call once before capturing (FFT tables, scratch: hipdsp_ctx_reserve) -- nothing may
 * allocate during capture.  Filter cut-offs change between replays through
 * hipdsp_sosplan_set_host + a captured hipdsp_sosplan_upload (DataBrowser.update_filter
 * -> BufferedFilter.update -> recompute_all, databrowser.py:1264-1288).  The time segmentation and the
 * number of warm-up tiles are those of the plans AT CAPTURE TIME (the state hand-over between segments
 * is recomputed on the device from the plan it finds): capture with the slowest-decaying filters of the
 * sweep -- lowest high-pass and lowest envelope cut-off -- or a replay with a longer memory than the
 * captured warm-up starts its segments with a history that has not fully decayed. */
typedef struct hipdsp_graph hipdsp_graph;
int hipdsp_graph_begin(hipdsp_ctx *ctx);
int hipdsp_graph_end(hipdsp_ctx *ctx, hipdsp_graph **out);
int hipdsp_graph_launch(hipdsp_ctx *ctx, hipdsp_graph *graph);
int hipdsp_graph_destroy(hipdsp_ctx *ctx, hipdsp_graph *graph);

/* ---- device memory helpers (so a non-torch host can keep stages resident) */

/* hipdsp_free keeps blocks of up to 256 MiB (or "pool_limit_mb", if that is larger) in a per-context cache (at most "pool_limit_mb",
 * default 1024; 0 turns it off) and hipdsp_malloc hands them out again, because hipMalloc /
 * hipFree synchronise the device and an interactive redraw needs temporaries.  The cache is
 * stream-ordered: free a block through a context whose stream is behind all work on it (order
 * other contexts' streams with hipdsp_event_record / hipdsp_event_wait first).  A cached block
 * remembers the stream it was freed on: handed out again after hipdsp_ctx_set_stream, the new
 * stream first waits for an event recorded at the free. */
int hipdsp_malloc(hipdsp_ctx *ctx, size_t bytes, void **dptr);
int hipdsp_free(hipdsp_ctx *ctx, void *dptr);
/* A block for a trace that kernels WRITE (the buffers BufferedData.allocate_buffer creates, src/audian/buffereddata.py:69-70,
 * 112-114): up to `tries` (<= 8) blocks are allocated side by side, a memset over each is timed on the context's stream
 * (which is synchronised), the fastest stays, the others go back through hipdsp_free.  Which physical pages a block got
 * moves a write stream into it by up to 12 % on this part (the envelope's backward sweep: 5.5 ... 6.2 ms into blocks of
 * one process, and the memset predicts it); nothing in user space chooses them, but it can choose among them.  Blocks
 * under 64 MiB, tries <= 1 and calls inside a stream capture are plain hipdsp_malloc calls.  The block is zeroed. */
int hipdsp_malloc_probed(hipdsp_ctx *ctx, size_t bytes, int tries, void **dptr);
/* Cache statistics (any pointer may be NULL) / give every cached block back to the driver -- and the context's scratch,
 * which only grows otherwise (hipdsp_envelope_multi parks two slabs of the trace's size there), unless a captured graph
 * of the context is alive (it holds the scratch's address). */
int hipdsp_pool_stats(hipdsp_ctx *ctx, size_t *cached_bytes, uint64_t *hits, uint64_t *misses);
int hipdsp_pool_trim(hipdsp_ctx *ctx);
int hipdsp_memset(hipdsp_ctx *ctx, void *dptr, int value, size_t bytes);
int hipdsp_memcpy_h2d(hipdsp_ctx *ctx, void *dst, const void *host_src, size_t bytes);
int hipdsp_memcpy_d2h(hipdsp_ctx *ctx, void *host_dst, const void *src, size_t bytes);
int hipdsp_memcpy_d2d(hipdsp_ctx *ctx, void *dst, const void *src, size_t bytes);
/* `height` rows of `width` bytes between pitched device blocks (ring-buffer recycling
 * of the device mirror, buffereddata.py:87). */
int hipdsp_memcpy2d_d2d(hipdsp_ctx *ctx, void *dst, size_t dst_pitch, const void *src,
                        size_t src_pitch, size_t width, size_t height);

/* HIP events on the context's stream (bench.py times kernels with these). */
int hipdsp_event_create(hipdsp_ctx *ctx, void **event);
int hipdsp_event_destroy(hipdsp_ctx *ctx, void *event);
int hipdsp_event_record(hipdsp_ctx *ctx, void *event);
/* Work queued on the context's stream after this call waits for `event` (recorded on any stream
 * of the device): orders two contexts that run on streams of their own. */
int hipdsp_event_wait(hipdsp_ctx *ctx, void *event);
int hipdsp_event_elapsed_ms(hipdsp_ctx *ctx, void *start, void *stop, float *ms);

/* Profiling hook: when set (non-NULL), hipdsp_envelope records this event between
 * its forward and backward kernels, so a bench can time the two separately. */
int hipdsp_ctx_set_mid_event(hipdsp_ctx *ctx, void *event);

/* ---- layout conversion at the edge -------------------------------------- */

/* (T, C) interleaved float64 / float32 -> planar (C, dst_pitch) float32.
 * Replaces the implicit layout of BufferedArray buffers (buffereddata.py:70). */
int hipdsp_pack_f64(hipdsp_ctx *ctx, const double *src_tc, float *dst, int64_t dst_pitch,
                    int64_t frames, int64_t channels);
int hipdsp_pack_f32(hipdsp_ctx *ctx, const float *src_tc, float *dst, int64_t dst_pitch,
                    int64_t frames, int64_t channels);
/* planar (C, src_pitch) float32 -> (T, C) interleaved float64. */
int hipdsp_unpack_f64(hipdsp_ctx *ctx, const float *src, int64_t src_pitch, double *dst_tc,
                      int64_t frames, int64_t channels);
/* (C, T', F) float32 -> (T', C, F) float64: the reference's
 * Sxx.transpose((1, 2, 0)) (bufferedspectrogram.py:58).  src_pitch = elements
 * between consecutive channels of src (>= frames*nfreq). */
int hipdsp_unpack_spectrum_f64(hipdsp_ctx *ctx, const float *src, int64_t src_pitch,
                               double *dst_tcf, int64_t frames, int64_t channels,
                               int64_t nfreq);

/* ---- IIR filter plans ---------------------------------------------------- */

/* A plan holds, in device memory, everything the block-parallel biquad cascade
 * needs for one SOS table: coefficients, block transition matrices, steady-state
 * initial conditions (scipy sosfilt_zi) and the warm-up length.  Updating a plan
 * (hipdsp_sosplan_set) is an async copy on the stream, so a captured hipGraph
 * replays under new cut-offs. */
int hipdsp_sosplan_create(hipdsp_ctx *ctx, hipdsp_sosplan **out);
int hipdsp_sosplan_destroy(hipdsp_ctx *ctx, hipdsp_sosplan *plan);
/* host_sos: (n_sections, 6) float64 rows [b0 b1 b2 a0 a1 a2], a0 == 1, exactly
 * what scipy.signal.butter(..., output='sos') returns
 * (bufferedfilter.py:44-52, bufferedenvelope.py:47-52). */
int hipdsp_sosplan_set(hipdsp_ctx *ctx, hipdsp_sosplan *plan, const double *host_sos,
                       int n_sections);
/* The two halves of hipdsp_sosplan_set, for hipGraph use: _set_host computes the
 * plan into pinned host memory (no stream work); _upload enqueues the copy to the
 * device block and may be captured, so each replay picks up the latest _set_host. */
int hipdsp_sosplan_set_host(hipdsp_ctx *ctx, hipdsp_sosplan *plan, const double *host_sos,
                            int n_sections);
int hipdsp_sosplan_upload(hipdsp_ctx *ctx, hipdsp_sosplan *plan);
/* The host half of a plan without any device: warm-up length (smallest multiple of the
 * 2048-sample tile with ||A^warm||_inf < 2^-60; 2^50 tiles when the filter does not decay),
 * scipy's sosfiltfilt pad length and sosfilt_zi (2*n_sections values); any output may be NULL. */
int hipdsp_sos_plan_host(const double *host_sos, int n_sections, int64_t *warmup, int *edge,
                         double *zi);
/* How the block-parallel IIR cuts `frames` samples of `channels` channels into time segments
 * (one wave per channel and segment) on `n_cus` compute units that hold up to `waves_max` such
 * waves each ("sos_waves_per_cu", 16; the fused sweeps: 8 pairs), when a segment has to re-read
 * `warmup` samples before its range.  The count minimises
 *     rounds x (segment + warm-up) x cost of a tile step at w waves per CU
 * (rounds of n_cus x waves_max units when there are more).  per_simd selects the sweep's measured cost table:
 * 4 (the envelope's backward sweep, four SIMDs per CU): memory-bound from two waves per SIMD on, so a tile step
 * costs ceil(w / 4) / 2, and 0.65 at one wave per SIMD (profiles/r03_occupancy_sweep.log): 8 waves per CU are
 * preferred to 16, short jobs get 4.  0 (the fused sweeps): waves_max x (1 + 0.25 (1 - w / waves_max)) -- the
 * CU is filled whenever the job allows.  -1 (the band-pass alone, no prefetch): 7 + 0.5625 w, and -2 (band-pass +
 * envelope states, prefetching): 1.4 + 0.9125 w (profiles/r03_sos_waves.log) -- both fill the CU for long jobs.
 * segment_frames is a multiple of the 2048-sample tile.
 * Host only (tests, capacity planning). */
int hipdsp_sos_segments_host(int64_t n_cus, int waves_max, int per_simd, int max_segments, int64_t frames,
                             int64_t channels, int64_t warmup, int64_t *segment_frames,
                             int *n_segments);
/* Introspection (tests): warm-up length in samples, sosfiltfilt pad length. */
int hipdsp_sosplan_info(hipdsp_ctx *ctx, hipdsp_sosplan *plan, int64_t *warmup, int *edge);

/* ---- the hot path --------------------------------------------------------- */

/* Non-finite samples (NaN, +-Inf) in x, every entry point below: as in the reference, i.e. as scipy does -- the
 * filtered trace of that channel is NaN from the sample on TO THE END of the call's slab (sosfilt's state stays
 * NaN), every spectrogram frame that reaches that far is NaN in every bin (dB: NaN), the envelope of that channel
 * is NaN everywhere (sosfiltfilt's backward pass starts from the NaN end); other channels are not affected.  The
 * time segments a sweep is cut into do not show (csrc/sos_device.h: FloodArgs). */

/* Input and output of a sweep must not overlap (HIPDSP_ERR_INVALID): a sweep is cut into time segments that run
 * concurrently and re-read their warm-up from the input -- the reference never filters in place either
 * (dest is a view of the trace's own ring buffer).  hipdsp_envelope_multi copies its input first and may. */

/* BufferedFilter.process (bufferedfilter.py:31-36):
 *   y[c, :] = sosfilt(sos, x[c, :])[skip:]      zero initial state, per channel.
 * x: (channels, x_pitch) with `frames` valid samples; y: (channels, y_pitch) with
 * frames - skip valid samples.  plan == NULL copies x[skip:] (the sos-is-None
 * pass-through branch, bufferedfilter.py:32-33). */
int hipdsp_sosfilt(hipdsp_ctx *ctx, const hipdsp_sosplan *plan, const float *x,
                   int64_t x_pitch, float *y, int64_t y_pitch, int64_t channels,
                   int64_t frames, int64_t skip);

/* BufferedEnvelope.process (bufferedenvelope.py:34-41):
 *   y = sosfiltfilt(sos, gain*|x|, axis=0)[skip:]; if clamp: y[y < 0] = 0
 * with scipy's default odd padding of 3*ntaps samples, sosfilt_zi-scaled initial
 * conditions, forward then backward pass.  `rectify` != 0 applies gain*|x|
 * (gain = pi/2 in the reference), rectify == 0 filters x itself (plain
 * sosfiltfilt, e.g. the playback chain databrowser.py:1718-1729).
 * Returns HIPDSP_ERR_TOO_SHORT when frames <= padlen (scipy raises ValueError).
 * plan == NULL writes zeros (bufferedenvelope.py:35-36). */
int hipdsp_envelope(hipdsp_ctx *ctx, const hipdsp_sosplan *plan, const float *x,
                    int64_t x_pitch, float *y, int64_t y_pitch, int64_t channels,
                    int64_t frames, int64_t skip, int rectify, double gain, int clamp);

/* Batch form of the two calls above for a whole slab (the envelope is taken of the SAME
 * frames the filter produces, skip = 0):
 *   yf  = sosfilt(fplan, x)                          (BufferedFilter.process)
 *   env = sosfiltfilt(eplan, gain*|yf|) [clamped]    (BufferedEnvelope.process on yf)
 * One sweep over x writes yf and, instead of the forward output of sosfiltfilt, only the
 * envelope cascade's state at every 2048-sample tile border (context scratch); the backward
 * sweep re-reads yf, recomputes the forward output tile by tile from those states and filters
 * it backwards: 8 + 8 bytes per sample instead of 8 + 16.  Results equal the two separate calls
 * up to float64 rounding of the IIR state.
 * phase: 0 = both sweeps; 1 = forward sweep only (yf complete, states parked in the context
 * scratch); 2 = backward sweep only (env from yf and those states) -- so that other work on yf
 * (the spectrogram) can be enqueued in between; no call that uses the scratch of THIS context
 * (envelope, envelope_multi, nfft > 32768, mean_spectrum_db) may come between phase 1 and phase 2.
 * env_first: the envelope is taken of yf[env_first:] -- env rows hold frames - env_first samples, env[i] belongs
 * to sample env_first + i of yf -- which is what BufferedEnvelope's buffer is after a scroll (its second of
 * pre-roll trimmed by BufferedData.align_buffer, buffereddata.py:75-88; sosfiltfilt then pads and starts at that
 * sample).  Phase 2 must be given the env_first of the forward sweep whose tile states it consumes (phase 1 or
 * hipdsp_chain_forward; HIPDSP_ERR_INVALID otherwise).  HIPDSP_ERR_TOO_SHORT when frames - env_first <= padlen. */
int hipdsp_sosfilt_envelope(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan,
                            const hipdsp_sosplan *eplan, const float *x, int64_t x_pitch,
                            float *yf, int64_t yf_pitch, float *env, int64_t env_pitch,
                            int64_t channels, int64_t frames, int rectify, double gain, int clamp,
                            int phase, int64_t env_first);

/* BufferedEnvelope.process for cascades LONGER than HIPDSP_MAX_SECTIONS (the reference accepts any
 * filter_order: bufferedenvelope.py:13-16,44-55; a band-pass envelope of order >= 5 or a low-pass of
 * order >= 9 has more than four sections): the SOS table is split over `n_plans` plans (in cascade order,
 * each <= HIPDSP_MAX_SECTIONS sections) and scipy's sosfiltfilt (scipy/signal/_signaltools.py:4807-4828) is
 * run step by step -- odd extension by padlen of the WHOLE cascade, forward pass from zi * ext[0], time
 * reversal, forward pass from zi * y[-1], reversal and trim -- where every plan starts from its own
 * sosfilt_zi scaled by the DC gain of the sections in front of it, exactly as sosfilt_zi of the whole table
 * would give.  The hand-over between plans is float32.  Same arguments and errors as hipdsp_envelope
 * (HIPDSP_ERR_TOO_SHORT when frames <= padlen); 56 instead of 16 bytes per sample, two temporaries of
 * (channels, frames + 2 padlen) floats in the context's scratch (which grows to hold them once and keeps its size
 * until hipdsp_pool_trim(): like every call that uses the scratch, not between hipdsp_sosfilt_envelope's phases 1
 * and 2 -- a backward sweep behind it returns HIPDSP_ERR_INVALID, the tile states are gone). */
int hipdsp_envelope_multi(hipdsp_ctx *ctx, const hipdsp_sosplan *const *plans, int n_plans, const float *x,
                          int64_t x_pitch, float *y, int64_t y_pitch, int64_t channels, int64_t frames,
                          int64_t skip, int rectify, double gain, int clamp);

/* The forward half of the batch chain in ONE pass over x: BufferedFilter.process
 * (bufferedfilter.py:31-36) writes yf, the envelope's forward sweep parks its tile states in the
 * context scratch exactly like hipdsp_sosfilt_envelope(..., phase = 1), and
 * BufferedSpectrogram.process (bufferedspectrogram.py:45-59) of yf goes to psd -- the spectrogram
 * takes the filtered tiles from on-chip memory instead of reading yf back (12 instead of 16 bytes
 * per sample).  Follow with hipdsp_sosfilt_envelope(..., phase = 2) for the envelope.  Results
 * equal hipdsp_sosfilt_envelope(phase 1) + hipdsp_spectrogram up to float32 rounding of the
 * frames that straddle an internal segment border.
 * Covers the window lengths whose frames are register windows of the sweep's 2048-sample tiles -- nfft / hop
 * 2048/1024, 2048/512, 1024/512, 1024/256, 512/256 and 256/128 (the reference's overlap selector offers 50 % and 75 %,
 * databrowser.py:516-540; BASELINE configs[1] is 1024/256; 256/128 is BufferedSpectrogram's default,
 * bufferedspectrogram.py:14-16) -- band-pass plans of up to four and envelope
 * plans of up to two decaying sections, and frames >= 8192; anything else returns HIPDSP_ERR_UNSUPPORTED
 * (use the separate calls).  psd layout, the zero tail and the optional db_out (decibel(psd), fused epilogue:
 * what SpecItem.update_plot shows, specitem.py:36) as in hipdsp_spectrogram, for every window of the list.
 * eplan == NULL: no envelope behind the filter (the reference's default trace set is filter + spectrogram,
 * src/audian/plugins.py:11-13) -- band-pass and spectrogram only, nothing is parked in the scratch.
 * spec_frames: the spectrogram is handed only the first spec_frames samples of yf (0 = all `frames`): through
 * BufferedData.load_buffer (buffereddata.py:91-109) BufferedSpectrogram.process sees its own frames times hop
 * plus ONE sample of the filtered buffer, so its last frame(s) are zero although the filter has the samples;
 * this is what lets one launch serve BufferedFilter.recompute_all() (buffereddata.py:149-153).
 * spec_first, env_first: where the derived traces start inside the filtered buffer once the user has scrolled
 * (DataBrowser.set_times -> Data.update_times -> align_buffer, buffereddata.py:75-88, data.py:225-236): the filtered
 * buffer then starts at an arbitrary sample of the recording, the spectrogram's frame 0 at the next multiple of hop
 * -- sample spec_first = ceil(offset / hop) hop - offset of yf, frame k = yf[spec_first + k hop : + nfft], and
 * spec_frames counts from there -- and the envelope is taken of yf[env_first:] only (its one second of pre-roll is
 * trimmed: sosfiltfilt's odd extension and zi * ext[0] sit at sample env_first; the tile states parked for
 * hipdsp_sosfilt_envelope(..., phase = 2, env_first) belong to that envelope).  Any 0 <= spec_first, env_first <=
 * frames; the sweep shifts its tile grid (by less than one tile of zeros in front of the trace) so that frames stay
 * register windows of its tiles, and the tile the envelope starts in holds the odd extension in front of sample
 * env_first and the extension's first value in front of that, for which zi * value is the cascade's steady state.
 * The sweep counts tiles and frames in 32 bits: frames_out < 2^31 - 65536 (HIPDSP_ERR_INVALID beyond; a spectrogram of
 * that many frames is terabytes).  The spectrogram's frame means come from the band-pass's own float64 arithmetic (the
 * reference's float64 mean, detrend='constant'), not from a float32 sum. */
int hipdsp_chain_forward(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan,
                         const hipdsp_sosplan *eplan, const float *x, int64_t x_pitch, float *yf,
                         int64_t yf_pitch, int64_t channels, int64_t frames, int rectify,
                         double gain, int nfft, int hop, double fs, float *psd, float *db_out,
                         int64_t frames_out, int64_t psd_pitch, int64_t spec_frames, int64_t spec_first,
                         int64_t env_first);

/* Frame split of the batch chain (nfft 2048 / hop 1024): with the context option "chain_split_frames" set,
 * hipdsp_chain_forward writes only the EVEN frames 2t of psd (frame 2t is tile t of its sweep) and this call,
 * which replaces hipdsp_sosfilt_envelope(..., phase = 2) behind it, writes the envelope
 * (BufferedEnvelope.process, bufferedenvelope.py:34-41: the backward half of sosfiltfilt from the tile states
 * the forward sweep parked in the context scratch) AND the ODD frames 2t+1 (second half of tile t, first half
 * of tile t+1) of BufferedSpectrogram.process (bufferedspectrogram.py:45-59) -- both launches then carry one
 * FFT per tile instead of two in the forward sweep (which is bound by VALU issue) and none in the backward
 * sweep (which is not), and move 10 bytes per sample each.  Same psd / frames_out / psd_pitch as the
 * forward call; together the two calls write every frame below n_valid, the forward call the zero tail.
 * Covers envelope plans of one or two decaying sections; HIPDSP_ERR_UNSUPPORTED otherwise.  The forward call behind
 * it must have walked the unshifted grid (spec_first = env_first = 0, the same channels / frames / sections) and
 * nothing may have used the context's scratch in between: HIPDSP_ERR_INVALID otherwise. */
int hipdsp_chain_backward(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, const float *yf, int64_t yf_pitch,
                          float *env, int64_t env_pitch, int64_t channels, int64_t frames, int rectify,
                          double gain, int clamp, int nfft, int hop, double fs, float *psd,
                          int64_t frames_out, int64_t psd_pitch);

/* The time segmentation hipdsp_chain_forward uses for `channels` x `frames` with these plans
 * (one IIR wave per channel and segment): segment s covers frames [s * segment_frames,
 * (s + 1) * segment_frames).  For tests and integrators that want to look at the seams; the
 * results do not depend on it beyond float32 rounding of the frames that straddle a border. */
int hipdsp_chain_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *fplan, const hipdsp_sosplan *eplan,
                      int64_t channels, int64_t frames, int64_t *segment_frames, int *n_segments);

/* The same for hipdsp_chain_backward, whose segments are counted from the END of the trace: segment s covers
 * frames [first_border - s * segment_frames, first_border - (s - 1) * segment_frames), s = 0 the last one. */
int hipdsp_chain_backward_plan(hipdsp_ctx *ctx, const hipdsp_sosplan *eplan, int64_t channels, int64_t frames,
                               int64_t *first_border, int64_t *segment_frames, int *n_segments);

/* BufferedSpectrogram.process (bufferedspectrogram.py:45-59) ==
 * scipy.signal.spectrogram(x, fs, 'hann', nperseg=nfft, noverlap=nfft-hop,
 * detrend='constant', scaling='density', mode='psd'):
 *   out[c, k, :] = one-sided PSD of x[c, k*hop : k*hop + nfft] for k < n_valid,
 *   zeros for n_valid <= k < frames_out,
 * where n_valid = (nsource - (nfft - hop)) / hop and
 * nsource = min((frames_out - 1)*hop + nfft, frames)  (0 valid frames when
 * nsource < nfft).  out is (channels, frames_out, nfft/2 + 1) float32 with out_pitch
 * elements between consecutive channels (0 = compact, frames_out*(nfft/2 + 1)).
 * If db_out != NULL it additionally receives decibel(out) (fused epilogue,
 * specitem.py:36) in the same layout.  nfft: any power of two in [8, 524288] (the reference's
 * nfft selector, databrowser.py:516; 65536 in the registers of one workgroup, 131072 of two, 262144 and 524288 as tasks of one or two passes of such a workgroup)
 * and, for the values the reference's clamp to len(source)//2 can produce, any other size up
 * to 131072 (direct DFT, O(nfft^2), meant for the rare short recording).
 * nfft 262144 and 524288 (windows of 2.7 and 5.5 s at 96 kHz) are covered, not streamed: NOT roofline kernels --
 * every task re-reads the frame and the first pass parks its points in `out` and reads them back (4.8 x / 8.1 x the
 * algorithmic bytes, 0.6 / 0.4 TB/s).  For these two sizes `out` is therefore also a WORK AREA while the call runs:
 * ordinary device memory, nobody else reading or writing it until the call has completed on the context's stream.
 * nfft 131072 has the two workgroups of a frame own interleaved bins (one radix-2 step in front of the transform):
 * they share every 32-byte sector of the output, 1.1 x the algorithmic bytes for the PSD, 1.4 x with db_out. */
int hipdsp_spectrogram(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                       int64_t frames, int nfft, int hop, double fs, float *out,
                       float *db_out, int64_t frames_out, int64_t out_pitch);

/* thunderlab.powerspectrum.decibel (specitem.py:28,36; spectrogramplot.py:159;
 * bufferedspectrogram.py:116-117): out = 10*log10(p/ref_power), -inf where
 * p <= min_power. */
int hipdsp_decibel(hipdsp_ctx *ctx, const float *p, float *out, int64_t n, double ref_power,
                   double min_power);
/* SpecItem.update_plot (specitem.py:36): decibel(buffer[:, ch, :].T) -- one
 * channel's (frames, nfreq) slab to a (nfreq, frames) dB image. */
int hipdsp_decibel_image(hipdsp_ctx *ctx, const float *spec_tf, float *image_ft,
                         int64_t frames, int64_t nfreq, double ref_power, double min_power);
/* The same image at screen resolution: column c = decibel(max over the frames
 * [start + c*step, min(start + (c+1)*step, stop)) ), i.e. np.maximum.reduceat over the
 * segments arange(0, stop - start, step) -- the min/max screen decimation TraceItem.update_plot
 * applies to traces (traceitem.py:42-61), applied to the spectrogram image (the reference's
 * README TODO "Implement downsampling of spectrograms", README.md:96).  image_fc is
 * (nfreq, ceil((stop - start) / step)); NaN propagates like np.maximum. */
int hipdsp_decibel_image_decimate(hipdsp_ctx *ctx, const float *spec_tf, float *image_fc,
                                  int64_t frames, int64_t nfreq, int64_t start, int64_t stop,
                                  int64_t step, double ref_power, double min_power);

/* ---- next rows --------------------------------------------------- */

/* Screen-resolution decimation of traces on the device (TraceItem.update_plot,
 * traceitem.py:55-61; the same reduction fills the overview cache, compresseddata.py:48-52):
 *   segments = arange(0, stop - start, step)
 *   out[c, 0::2] = np.minimum.reduceat(x[c, start:stop], segments)
 *   out[c, 1::2] = np.maximum.reduceat(x[c, start:stop], segments)
 * for every channel; out is (channels, out_pitch) float32 with 2*ceil((stop-start)/step)
 * valid values per row.  Only the few thousand plot points then cross PCIe (or xGMI). */
int hipdsp_minmax_decimate(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels,
                           int64_t start, int64_t stop, int64_t step, float *out,
                           int64_t out_pitch);

/* Playback chain (DataBrowser.play_region, databrowser.py:1711-1729): out[k] = mean over the
 * listed channels of x[c, start + k], k < n, optionally times the heterodyne carrier
 * sin(2 pi k * heterodyne_cycles_per_sample) (0 = none).  The low-pass that follows is
 * hipdsp_envelope(rectify = 0, clamp = 0), i.e. plain sosfiltfilt, and `[::nstep]` is
 * hipdsp_stride_copy (out[i] = x[i*step], ceil(n/step) values). */
int hipdsp_channel_mean(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, const int *host_channels,
                        int count, int64_t start, int64_t n, double heterodyne_cycles_per_sample,
                        float *out);
int hipdsp_stride_copy(hipdsp_ctx *ctx, const float *x, int64_t n, int64_t step, float *out);

/* Maximum of n non-negative floats (PSD values) into out[0]; with the strided gather of
 * hipdsp_memcpy2d_d2d it serves BufferedSpectrogram.estimate_noiselevels
 * (bufferedspectrogram.py:109-126: max dB = decibel(max power), P95 of the top 1/16 band). */
int hipdsp_max_nonneg(hipdsp_ctx *ctx, const float *x, int64_t n, float *out);

/* out2[0], out2[1] = the order statistics of rank `rank` and `rank + 1` (zero based, ascending; the second
 * clamped to the last) of the rows x cols non-negative floats x[i * row_stride + j] -- what
 * np.percentile(..., 95) interpolates between in BufferedSpectrogram.estimate_noiselevels
 * (bufferedspectrogram.py:115-117: the top F/16 bins of one channel's (frames, F) slab; decibel is
 * monotonic, so the percentile of the dB values is the interpolation of the dB of these two).  Exact
 * (radix select on the float bits), one workgroup, nothing but two floats leaves the device. */
int hipdsp_band_order_stats(hipdsp_ctx *ctx, const float *x, int64_t rows, int64_t cols, int64_t row_stride,
                            int64_t rank, float *out2);

/* audioio's unwrap() of clipped recordings, which the reference arms on its raw loader for every buffer
 * it loads (Data.open -> self.data.set_unwrap(unwrap, unwrap_clip, False, unit), src/audian/data.py:180;
 * CLI -u / -U, src/audian/audian.py:1485-1512, default threshold 1.5): a step between successive samples
 * beyond `thresh` is a wrap-around of a signal that left [-ampl_max, ampl_max); from there on 2 * ampl_max
 * is subtracted (step up) or added (step down), cumulatively along time, per channel, starting from
 * zero at the first frame of the slab.  Then `clips`: clip to +-ampl_max; else `down_scale`: halve.
 * Planar float32 in and out; x and y must not overlap (a chunk reads the last sample of the chunk
 * before it while that one is being written).  audioio's source is neither in the reference tree nor
 * in this image: restated from its documentation and the reference's call sites, parity unpinned.
 * Uses the context scratch (4 bytes per 16384 frames and channel). */
int hipdsp_unwrap(hipdsp_ctx *ctx, const float *x, int64_t x_pitch, int64_t channels, int64_t frames,
                  double thresh, double ampl_max, int clips, int down_scale, float *y, int64_t y_pitch);

/* PCM ingest: interleaved little-endian signed PCM (frames, channels) of 2, 3 or 4 bytes per
 * sample -> planar float32 times `scale` (1/2^(bits-1) reproduces the [-1, 1) floats that
 * audioio / thunderlab's DataLoader give audian, data.py:172).  Uploading the file's own
 * integers instead of float64 cuts the PCIe volume of the raw slab 2.7-4x. */
int hipdsp_pcm_unpack(hipdsp_ctx *ctx, const void *pcm_tc, int sample_bytes, int64_t frames,
                      int64_t channels, double scale, float *dst, int64_t dst_pitch);

/* Power spectrum of the visible window (SpectrogramPlot.update_plot,
 * spectrogramplot.py:158-160):
 *   power = np.mean(spec[i0:i1, :], axis=0); power = decibel(power); power[power < floor] = floor
 * on one channel's (frames, nfreq) slab; out gets nfreq float32 values (floor_db = -200 in
 * the reference).  Uses the context scratch (shared with the envelope). */
int hipdsp_mean_spectrum_db(hipdsp_ctx *ctx, const float *spec_tf, int64_t nfreq, int64_t i0,
                            int64_t i1, double ref_power, double min_power, double floor_db,
                            float *out);

/* ---- multi-GPU exchange ---------------------------------------- */

/* One process per GPU, channels sharded in contiguous blocks of the planar layout, so
 * the merged spectrogram tile is ONE all-gather of contiguous per-rank chunks (RCCL over
 * xGMI).  Rank 0 obtains a 128-byte id (hipdsp_comm_unique_id) and hands it to the other
 * ranks by whatever channel the host has (file, socket, torch store); every rank then
 * creates its communicator.  RCCL is loaded on first use. */
#define HIPDSP_UNIQUE_ID_BYTES 128
typedef struct hipdsp_comm hipdsp_comm;
int hipdsp_comm_unique_id(void *id_out);
int hipdsp_comm_create(hipdsp_ctx *ctx, const void *unique_id, int rank, int nranks,
                       hipdsp_comm **out);
int hipdsp_comm_destroy(hipdsp_ctx *ctx, hipdsp_comm *comm);
/* recv[r*count : (r+1)*count] = rank r's send[0:count], on the context's stream. */
int hipdsp_allgather_f32(hipdsp_ctx *ctx, hipdsp_comm *comm, const float *send, float *recv,
                         int64_t count_per_rank);

/* ---- measurement aid ("also report a measured device-copy ceiling") ---- */

/* dst[0:bytes] = src[0:bytes] by the copy pattern that reaches this part's streaming ceiling: one float4 per
 * thread, 256-thread blocks, no loop (MI355X_MICROARCH.md: 6.29 TB/s read + write; grid-stride copies and
 * hipMemcpy D2D stay at 4.7-5.1).  bytes must be a multiple of 16; the buffers must not overlap.  bench.py
 * times it as `roofline.device_copy_GBps`. */
int hipdsp_copy_probe(hipdsp_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- synthetic input (bench / tests) --------------------------- */

/* x[c, t] = 0.5*u(seed, c, t) + 0.5*sin(2*pi*1000*(1 + (c0 + c)/c_total)*t/rate),
 * u uniform in [-1, 1) from a counter-based hash; generated on device. */
int hipdsp_synth(hipdsp_ctx *ctx, float *x, int64_t x_pitch, int64_t channels,
                 int64_t frames, double rate, uint64_t seed, int64_t c0, int64_t c_total);

#ifdef __cplusplus
}
#endif
#endif /* HIP_DSP_H */
