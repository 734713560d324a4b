"""The HOST-ONLY translation units of libhip_dsp under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build;
SURVEY 5 asks for it, GPU sanitizers are not available on the pool): ctx.hip (context, options, scratch, the
stream-ordered block cache behind hipdsp_malloc / hipdsp_free) and sos_plan.hip (plan mathematics, segment planner)
are compiled with g++ against a fake HIP runtime (tests/fakehip: host memory, streams and events as heap objects)
and driven through the C ABI by a small program; any out-of-bounds access, leak or UB aborts it."""

import os
import subprocess

from conftest import ROOT

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hip/hip_runtime.h"
#include "hip_dsp.h"
size_t fake_hip_limit = 0, fake_hip_in_use = 0;
int fake_hip_fail_event_create = 0;
long fake_hip_waits = 0;
static const size_t HEAD = 32;
hipError_t hipMalloc(void **p, size_t n)
{
    if (fake_hip_limit && fake_hip_in_use + n > fake_hip_limit) { *p = nullptr; return hipErrorOutOfMemory; }
    char *b = (char *)malloc(n + HEAD);
    if (!b) { *p = nullptr; return hipErrorOutOfMemory; }
    memcpy(b, &n, sizeof(n));
    fake_hip_in_use += n;
    *p = b + HEAD;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    char *b = (char *)p - HEAD;
    size_t n; memcpy(&n, b, sizeof(n));
    fake_hip_in_use -= n;
    free(b);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    if (fake_hip_fail_event_create > 0) { fake_hip_fail_event_create--; *e = nullptr; return hipErrorInvalidValue; }
    *e = (hipEvent_t)calloc(1, sizeof(fake_event));
    return hipSuccess;
}
#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { printf("line %d: rc %d: %s\n", __LINE__, rc_, hipdsp_last_error()); return 1; } } while (0)
#define EXPECT(c) do { if (!(c)) { printf("line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main()
{
    // ---- plan mathematics and the segment planner over many inputs
    unsigned long long seed = 88172645463325252ULL;
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
    for (int it = 0; it < 400; it++) {
        int S = 1 + (int)(rnd() % 4);
        double sos[24];
        for (int s = 0; s < S; s++) {
            double r = 0.05 + 0.9499 * (double)(rnd() % 10000) / 10000.0, th = 3.14159 * (double)(rnd() % 10000) / 10000.0;
            sos[6*s] = 0.3; sos[6*s+1] = (rnd() % 3 == 0) ? 0.0 : 0.6; sos[6*s+2] = (rnd() % 4 == 0) ? 0.0 : 0.3;
            sos[6*s+3] = 1.0; sos[6*s+4] = -2.0 * r * __builtin_cos(th); sos[6*s+5] = (rnd() % 5 == 0) ? 0.0 : r * r;
        }
        int64_t warm = 0; int edge = 0; double zi[8];
        CHECK(hipdsp_sos_plan_host(sos, S, &warm, &edge, zi));
        EXPECT(warm >= 2048 && warm % 2048 == 0 && edge >= 3 && edge <= 27);
        int64_t seg = 0; int n = 0;
        int64_t frames = 1 + (int64_t)(rnd() % 60000000), channels = 1 + (int64_t)(rnd() % 300);
        int per = (rnd() & 1) ? 4 : 0;
        CHECK(hipdsp_sos_segments_host(1 + (int64_t)(rnd() % 300), 1 + (int)(rnd() % 16), per, (int)(rnd() % 3 == 0 ? rnd() % 9 : 0),
                                       frames, channels, warm, &seg, &n));
        EXPECT(seg % 2048 == 0 && n >= 1 && (int64_t)n * seg >= frames && (int64_t)(n - 1) * seg < frames);
    }
    {   // error paths
        double bad[6] = {1, 0, 0, 2, 0, 0};
        EXPECT(hipdsp_sos_plan_host(bad, 1, nullptr, nullptr, nullptr) == HIPDSP_ERR_INVALID);
        EXPECT(hipdsp_sos_plan_host(bad, 5, nullptr, nullptr, nullptr) == HIPDSP_ERR_UNSUPPORTED);
        int64_t seg; int n;
        EXPECT(hipdsp_sos_segments_host(0, 16, 4, 0, 10, 1, 0, &seg, &n) == HIPDSP_ERR_INVALID);
    }
    // ---- context, options, scratch, block cache
    hipdsp_ctx *ctx = nullptr;
    CHECK(hipdsp_ctx_create(0, nullptr, &ctx));
    EXPECT(hipdsp_ctx_set_option(ctx, "no_such_option", 1) == HIPDSP_ERR_INVALID);
    CHECK(hipdsp_ctx_set_option(ctx, "chain_reserve_cus", 8));
    EXPECT(hipdsp_ctx_set_option(ctx, "n_cus", 4) == HIPDSP_ERR_INVALID);    // fewer CUs than reserved: refused (ADVICE round 2)
    CHECK(hipdsp_ctx_set_option(ctx, "n_cus", 9));
    CHECK(hipdsp_ctx_set_option(ctx, "n_cus", 256));
    EXPECT(hipdsp_ctx_set_option(ctx, "sos_trace", 1234) == HIPDSP_ERR_INVALID);     // capacity first
    CHECK(hipdsp_ctx_reserve(ctx, 1 << 20));
    CHECK(hipdsp_ctx_reserve(ctx, 1 << 10));                       // never shrinks
    void *sa = nullptr, *sb = nullptr;
    CHECK(hipdsp_stream_create(ctx, &sa));
    CHECK(hipdsp_stream_create(ctx, &sb));
    std::vector<void *> blocks;
    for (int round = 0; round < 50; round++) {
        CHECK(hipdsp_ctx_set_stream(ctx, (round & 1) ? sa : sb));
        for (int k = 0; k < 20; k++) {
            void *p = nullptr;
            size_t n = 1 + (size_t)(rnd() % (3 << 20));
            CHECK(hipdsp_malloc(ctx, n, &p));
            memset(p, 0x5a, n);                                    // the whole block is ours
            blocks.push_back(p);
        }
        while (blocks.size() > 7) {
            size_t i = (size_t)(rnd() % blocks.size());
            CHECK(hipdsp_free(ctx, blocks[i]));
            blocks.erase(blocks.begin() + (long)i);
        }
    }
    size_t cached = 0; uint64_t hits = 0, misses = 0;
    CHECK(hipdsp_pool_stats(ctx, &cached, &hits, &misses));
    EXPECT(hits > 0 && misses > 0);
    // a block freed when no event can be made goes back to the driver (not leaked, not cached without an order)
    {
        CHECK(hipdsp_pool_trim(ctx));
        void *p = nullptr;
        CHECK(hipdsp_malloc(ctx, 4096, &p));
        // use up the spare events: none are left after a trim? force creation to fail for the next free
        size_t before = fake_hip_in_use;
        fake_hip_fail_event_create = 1000;
        CHECK(hipdsp_free(ctx, p));
        fake_hip_fail_event_create = 0;
        CHECK(hipdsp_pool_stats(ctx, &cached, nullptr, nullptr));
        EXPECT(cached == 0 ? fake_hip_in_use == before - 4096 : true);
    }
    // a block freed DURING a capture has no event: another stream must not get it ...
    {
        CHECK(hipdsp_pool_trim(ctx));
        CHECK(hipdsp_ctx_set_stream(ctx, sa));
        void *p = nullptr, *q = nullptr;
        CHECK(hipdsp_malloc(ctx, 65536, &p));
        CHECK(hipdsp_graph_begin(ctx));
        CHECK(hipdsp_free(ctx, p));                                 // cached, freed == NULL
        hipdsp_graph *g = nullptr;
        CHECK(hipdsp_graph_end(ctx, &g));
        CHECK(hipdsp_ctx_set_stream(ctx, sb));
        CHECK(hipdsp_malloc(ctx, 65536, &q));
        EXPECT(q != p);                                             // passed over: a fresh block
        CHECK(hipdsp_free(ctx, q));
        CHECK(hipdsp_graph_destroy(ctx, g));
        // ... and a capturing stream takes no block that would need an event wait
        CHECK(hipdsp_ctx_set_stream(ctx, sa));
        long waits = fake_hip_waits;
        CHECK(hipdsp_graph_begin(ctx));
        void *r = nullptr;
        CHECK(hipdsp_malloc(ctx, 65536, &r));                       // q was freed on sb with an event: not for a capture on sa
        EXPECT(fake_hip_waits == waits);
        CHECK(hipdsp_graph_end(ctx, &g));
        CHECK(hipdsp_graph_destroy(ctx, g));
        CHECK(hipdsp_free(ctx, r));
    }
    // out of device memory: the cache is given back and the allocation retried
    {
        void *p = nullptr, *q = nullptr;
        CHECK(hipdsp_malloc(ctx, 8 << 20, &p));
        CHECK(hipdsp_free(ctx, p));                                 // cached
        fake_hip_limit = fake_hip_in_use + (4 << 20);
        CHECK(hipdsp_malloc(ctx, 10 << 20, &q));                    // does not fit next to the cached block: trim + retry
        EXPECT(hipdsp_malloc(ctx, (size_t)64 << 20, &p) == HIPDSP_ERR_NOMEM);
        fake_hip_limit = 0;
        CHECK(hipdsp_free(ctx, q));
    }
    for (void *p : blocks) CHECK(hipdsp_free(ctx, p));
    CHECK(hipdsp_ctx_set_stream(ctx, nullptr));
    CHECK(hipdsp_stream_destroy(ctx, sa));
    CHECK(hipdsp_stream_destroy(ctx, sb));
    CHECK(hipdsp_ctx_destroy(ctx));
    EXPECT(fake_hip_in_use == 0);                                   // nothing of the "device" left behind
    printf("ok\n");
    return 0;
}
'''


def test_host_only_units_are_clean_under_asan_ubsan(tmp_path):
    src = tmp_path/'driver.cpp'
    src.write_text(DRIVER)
    exe = tmp_path/'driver'
    csrc = os.path.join(ROOT, 'audian_amd', 'csrc')
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-I', os.path.join(ROOT, 'tests', 'fakehip'), '-I', os.path.join(ROOT, 'include'), '-I', csrc,
                           '-x', 'c++', str(src), os.path.join(csrc, 'ctx.hip'), os.path.join(csrc, 'sos_plan.hip'),
                           '-o', str(exe)])
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), (r.stdout[-2000:], r.stderr[-3000:])
