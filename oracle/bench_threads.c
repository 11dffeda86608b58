/*
 * oracle/bench_threads.c - the CPU baseline under T host threads (TEST INFRASTRUCTURE: bench.py's cpu_baseline leg only).
 *
 * The reference is single-threaded (src/kzg_proof.rs:251-277 and :399-444 are plain loops), so what an operator of it gets from
 * a many-core host is T INDEPENDENT calls side by side - the revm precompile's verify_kzg_proof from T threads, or T beacon-node
 * style verify_blob_kzg_proof_batch calls.  This file runs exactly that with plain pthreads (rounds 1-4 drove the oracle from
 * Python threads over ctypes) and reports per-thread rates, so that the figure can be explained: every thread runs the same
 * calls on the same inputs, and with the per-thread scratch of kzg.c nothing is shared between them but read-only tables.
 */
#include "kzg_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int kind;
    size_t per_call, n_calls, first, stride;
    const uint8_t *blobs, *c, *z, *y, *p;
    const oracle_settings *s;
    double seconds;
    volatile int *go;
    /* out */
    uint64_t calls, bad;
    double busy_s;
} worker_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    while (!*w->go) {
    }
    const double t0 = now_s();
    size_t i = w->first % w->n_calls;
    do {
        int ok = 0, rc;
        if (w->kind == 0)
            rc = oracle_verify_kzg_proof(&ok, w->c + 48 * i, w->z + 32 * i, w->y + 32 * i, w->p + 48 * i, w->s);
        else {
            const size_t f = i * w->per_call;
            rc = oracle_verify_blob_kzg_proof_batch(&ok, w->blobs + (size_t)131072 * f, w->c + 48 * f, w->p + 48 * f, w->per_call, w->s, 1, 0);
        }
        if (rc != ORACLE_OK || !ok) w->bad++;
        w->calls++;
        i = (i + w->stride) % w->n_calls;
    } while (now_s() - t0 < w->seconds);
    w->busy_s = now_s() - t0;
    return NULL;
}

/* T threads, each calling for `seconds` (every call is finished, so a thread's busy time may exceed it):
 *   kind 0: oracle_verify_kzg_proof on tuple i of n_items (all valid)
 *   kind 1: oracle_verify_blob_kzg_proof_batch (single-threaded) of per_call blobs, call i of n_items / per_call (all valid)
 * out: [0] calls in all, [1] wall seconds, [2] calls that did not return Ok(true), [3] sum over threads of calls / busy time
 * (= the aggregate rate), [4] the slowest thread's rate, [5] the fastest thread's rate (calls/s). */
int oracle_bench_threads(double out[6], int kind, size_t threads, double seconds, const uint8_t *blobs, const uint8_t *c, const uint8_t *z,
                         const uint8_t *y, const uint8_t *p, size_t n_items, size_t per_call, const oracle_settings *s) {
    if (!out || !threads || !n_items || !c || !p || !s || (kind == 0 && (!z || !y)) || (kind == 1 && (!blobs || !per_call || n_items < per_call)))
        return ORACLE_BADARGS;
    if (kind == 0) per_call = 1;
    worker_t *w = (worker_t *)calloc(threads, sizeof *w);
    pthread_t *th = (pthread_t *)calloc(threads, sizeof *th);
    volatile int go = 0;
    size_t started = 0;
    for (size_t t = 0; t < threads; t++) {
        w[t] = (worker_t){kind, per_call, n_items / per_call, t, threads, blobs, c, z, y, p, s, seconds, &go, 0, 0, 0.0};
        if (pthread_create(&th[t], NULL, worker, &w[t]) != 0) break;
        started++;
    }
    const double t0 = now_s();
    go = 1;
    for (size_t t = 0; t < started; t++) pthread_join(th[t], NULL);
    const double wall = now_s() - t0;
    double calls = 0, bad = 0, rate = 0, lo = 1e300, hi = 0;
    for (size_t t = 0; t < started; t++) {
        const double r = w[t].busy_s > 0 ? (double)w[t].calls / w[t].busy_s : 0.0;
        calls += (double)w[t].calls;
        bad += (double)w[t].bad;
        rate += r;
        if (r < lo) lo = r;
        if (r > hi) hi = r;
    }
    out[0] = calls; out[1] = wall; out[2] = bad; out[3] = rate; out[4] = started ? lo : 0.0; out[5] = hi;
    free(w);
    free(th);
    return started == threads ? ORACLE_OK : ORACLE_ERROR;
}
