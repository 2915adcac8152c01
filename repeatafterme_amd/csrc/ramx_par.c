/*
 * ramx_par.c -- host-side helper of the report / output writers (reference report.c:161-502, ram_extend.c:606-779, which
 * print one core after the other with a printf per field): the cores are split into contiguous chunks, every chunk is
 * formatted by its own thread into memory streams, and the streams are written out in chunk order -- the bytes that reach
 * stdout and the files are the same, the formatting no longer is one serial loop over 100,000 cores.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ramx_internal.h"

struct par_job
{
  int lo, hi, n_outs;
  FILE *ms[RAMX_PAR_MAX_OUTS];
  char *buf[RAMX_PAR_MAX_OUTS];
  size_t len[RAMX_PAR_MAX_OUTS];
  ramx_chunk_fn fn;
  void *user;
};

static void *par_worker(void *arg)
{
  struct par_job *j = (struct par_job *)arg;
  j->fn(j->lo, j->hi, j->ms, j->user);
  for (int k = 0; k < j->n_outs; k++) if (j->ms[k]) fflush(j->ms[k]);
  return NULL;
}

int ramx_host_threads(long items, long min_items_per_thread)
{
  const char *env = getenv("RAMX_HOST_THREADS");
  long t = 1;
  if (env) t = atol(env);
  else
  {
    t = sysconf(_SC_NPROCESSORS_ONLN);
    if (t > 16) t = 16;
    if (min_items_per_thread > 0 && items / min_items_per_thread < t) t = items / min_items_per_thread;
  }
  return t < 1 ? 1 : (int)t;
}

void ramx_parallel_chunks(int n, int n_outs, FILE **real_outs, ramx_chunk_fn fn, void *user)
{
  if (n_outs > RAMX_PAR_MAX_OUTS) n_outs = RAMX_PAR_MAX_OUTS;
  const int T = ramx_host_threads(n, 2048);
  if (T <= 1 || n <= 0)
  {
    fn(0, n, real_outs, user);        /* straight into the real streams */
    return;
  }
  struct par_job *jobs = (struct par_job *)calloc((size_t)T, sizeof(*jobs));
  pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(*tid));
  for (int t = 0; t < T; t++)
  {
    jobs[t].lo = (int)((long long)n * t / T); jobs[t].hi = (int)((long long)n * (t + 1) / T);
    jobs[t].n_outs = n_outs; jobs[t].fn = fn; jobs[t].user = user;
    for (int k = 0; k < n_outs; k++)
      jobs[t].ms[k] = real_outs[k] ? open_memstream(&jobs[t].buf[k], &jobs[t].len[k]) : NULL;
  }
  for (int t = 1; t < T; t++)
    if (pthread_create(&tid[t], NULL, par_worker, &jobs[t]) != 0) { par_worker(&jobs[t]); tid[t] = 0; }
  par_worker(&jobs[0]);
  for (int t = 1; t < T; t++) if (tid[t]) pthread_join(tid[t], NULL);
  for (int t = 0; t < T; t++)
    for (int k = 0; k < n_outs; k++)
      if (jobs[t].ms[k])
      {
        fclose(jobs[t].ms[k]);
        if (jobs[t].len[k]) fwrite(jobs[t].buf[k], 1, jobs[t].len[k], real_outs[k]);
        free(jobs[t].buf[k]);
      }
  free(jobs); free(tid);
}

/* fn(lo, hi, user) over [0, n) on the host's cores (at most 16; one thread per min_items items; RAMX_HOST_THREADS overrides) */
struct for_job { int lo, hi; ramx_range_fn fn; void *user; };
static void *for_worker(void *arg)
{
  struct for_job *j = (struct for_job *)arg;
  j->fn(j->lo, j->hi, j->user);
  return NULL;
}
void ramx_parallel_for(int n, long min_items, ramx_range_fn fn, void *user)
{
  const int T = ramx_host_threads(n, min_items);
  if (T <= 1 || n <= 0) { if (n > 0) fn(0, n, user); return; }
  struct for_job *jobs = (struct for_job *)calloc((size_t)T, sizeof(*jobs));
  pthread_t *tid = (pthread_t *)calloc((size_t)T, sizeof(*tid));
  for (int t = 0; t < T; t++)
  {
    jobs[t].lo = (int)((long long)n * t / T); jobs[t].hi = (int)((long long)n * (t + 1) / T);
    jobs[t].fn = fn; jobs[t].user = user;
  }
  for (int t = 1; t < T; t++)
    if (pthread_create(&tid[t], NULL, for_worker, &jobs[t]) != 0) { for_worker(&jobs[t]); tid[t] = 0; }
  for_worker(&jobs[0]);
  for (int t = 1; t < T; t++) if (tid[t]) pthread_join(tid[t], NULL);
  free(jobs); free(tid);
}
