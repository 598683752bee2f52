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
