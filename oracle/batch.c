/*
 * batch.c -- TEST INFRASTRUCTURE ONLY: multi-threaded batch driver used for
 *   (a) bench.py's cpu_baseline leg (the CPU codec timed on the GPU box's host cores), and
 *   (b) fast oracle sweeps in tests.
 * The codec functions are passed in as pointers, so the same driver times either the restatement
 * in lz4_oracle.c ("port") or the reference's own C from oracle/_ref/libref_lz4.so ("reference").
 * Timing method mirrors src/LZ4.Tests.Helpers/TimedMethod.cs:66-69 (input bytes / elapsed) with
 * data pre-generated in memory and one worker per thread.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef int (*enc_fn)(const char* src, char* dst, int isize, int max_out);
typedef int (*dec_fn)(const char* src, char* dst, int osize);
typedef int (*decu_fn)(const char* src, char* dst, int isize, int max_out);

typedef struct {
    int op;                       /* 0 encode-like (enc_fn), 1 decode known size, 2 decode unknown size */
    void* fn;
    const uint8_t* src; int64_t src_stride; const int32_t* src_len;
    uint8_t* dst; int64_t dst_stride; const int32_t* dst_cap; int32_t* result;
    int64_t begin, end;
} job_t;

static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    for (int64_t i = j->begin; i < j->end; i++) {
        const char* s = (const char*)(j->src + i * j->src_stride);
        char* d = (char*)(j->dst + i * j->dst_stride);
        int r;
        if (j->op == 0)      r = ((enc_fn)j->fn)(s, d, j->src_len[i], j->dst_cap[i]);
        else if (j->op == 1) r = ((dec_fn)j->fn)(s, d, j->dst_cap[i]);
        else                 r = ((decu_fn)j->fn)(s, d, j->src_len[i], j->dst_cap[i]);
        j->result[i] = r;
    }
    return 0;
}

/* Runs fn over blocks [0,n) on `threads` pthreads; returns elapsed wall seconds. */
double lz4o_batch_run(int op, void* fn, const uint8_t* src, int64_t src_stride, const int32_t* src_len,
                      uint8_t* dst, int64_t dst_stride, const int32_t* dst_cap, int32_t* result,
                      int64_t n, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        job_t j = { op, fn, src, src_stride, src_len, dst, dst_stride, dst_cap, result,
                    n * t / threads, n * (t + 1) / threads };
        jobs[t] = j;
        pthread_create(&tid[t], 0, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- full-corpus encoder check: generate block -> compress -> checksum of the compressed bytes, nothing kept -------
 * For block index first + i*step (i in [0, n)) of distribution `dist`: the block is regenerated from its seed (CPU twin
 * of the device generator), compressed with `fn`, and only (compressed length, checksum of the compressed bytes) are
 * stored.  The GPU side computes the same two numbers over its own compressed rows (lz4hip_checksum_device), so every
 * block of a 2^20-block batch is compared with the CPU codec without holding the corpus in host memory: the reference's
 * own bar is identity over the whole corpus (src/LZ4.Tests/ConformanceTests.cs:121-133).
 * Blocks are handed out in chunks through an atomic cursor; work stops when `budget_seconds` have passed (0 = no limit).
 * Returns the number of blocks done: these are the blocks [0, returned) of the sequence. */
#include "synth.h"
#include <stdatomic.h>

typedef struct {
    enc_fn fn; int dist; uint64_t seed, first, step; int len, cap;
    int32_t* out_len; uint64_t* out_sum; int64_t n;
    _Atomic int64_t* cursor; double deadline; _Atomic int* stop;
} vjob_t;

static double now_s(void)
{
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void* vworker(void* arg)
{
    vjob_t* j = (vjob_t*)arg;
    uint8_t* raw = (uint8_t*)malloc((size_t)j->len + 64);
    uint8_t* comp = (uint8_t*)malloc((size_t)j->cap + 64);
    for (;;) {
        if (atomic_load(j->stop)) break;
        const int64_t i = atomic_fetch_add(j->cursor, 1);
        if (i >= j->n) break;
        lz4s_fill_block(j->dist, j->seed, j->first + (uint64_t)i * j->step, raw, j->len);
        const int r = j->fn((const char*)raw, (char*)comp, j->len, j->cap);
        j->out_len[i] = r;
        j->out_sum[i] = r > 0 ? lz4s_checksum(comp, r) : 0;
        if (j->deadline > 0 && (i & 15) == 0 && now_s() > j->deadline) atomic_store(j->stop, 1);
    }
    free(raw); free(comp);
    return 0;
}

int64_t lz4o_verify_stream(void* fn, int dist, uint64_t seed, uint64_t first, uint64_t step, int64_t n, int len, int cap,
                           int32_t* out_len, uint64_t* out_sum, int threads, double budget_seconds)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    _Atomic int64_t cursor = 0;
    _Atomic int stop = 0;
    vjob_t j = { (enc_fn)fn, dist, seed, first, step, len, cap, out_len, out_sum, n, &cursor,
                 budget_seconds > 0 ? now_s() + budget_seconds : 0.0, &stop };
    for (int64_t i = 0; i < n; i++) out_len[i] = -1;              /* -1 = not done */
    for (int t = 0; t < threads; t++) pthread_create(&tid[t], 0, vworker, &j);
    for (int t = 0; t < threads; t++) pthread_join(tid[t], 0);
    free(tid);
    /* blocks are claimed in order, so the done ones form a prefix except for the last `threads` claims */
    int64_t done = 0;
    while (done < n && out_len[done] >= 0) done++;
    return done;
}
