/* The host streamer alone (csrc/bamstream.c): open, drain, close timed apart.  gcc -O2 -Iinclude tools/bench_bamstream.c -Lbs_call_amd/lib -lbscall_amd
 * usage: bench_bamstream in.bam threads [slab_MB [n_slabs]] */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <bscall_amd.h>
static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const double t0 = now();
  bsc_bamstream *s;
  if (bsc_bamstream_open(argv[1], atoi(argv[2]), argc > 3 ? (uint64_t)atoi(argv[3]) << 20 : 0, argc > 4 ? atoi(argv[4]) : 0, &s)) {
    fprintf(stderr, "%s\n", bsc_last_error());
    return 1;
  }
  const double t1 = now();
  bsc_bam_slab sl;
  unsigned long long nb = 0, nr = 0;
  int r;
  while ((r = bsc_bamstream_next(s, &sl)) == 1) {
    nb += sl.n_bytes;
    nr += sl.n_recs;
    bsc_bamstream_release(s, &sl);
  }
  const double t2 = now();
  const int th = bsc_bamstream_threads(s);
  bsc_bamstream_close(s);
  const double t3 = now();
  if (r < 0) fprintf(stderr, "%s\n", bsc_last_error());
  printf("{\"threads\": %d, \"open_s\": %.3f, \"drain_s\": %.3f, \"close_s\": %.3f, \"inflated_GB_per_s\": %.2f, \"records\": %llu}\n", th, t1 - t0, t2 - t1, t3 - t2,
         (double)nb / (t2 - t1) / 1e9, nr);
  return r < 0;
}
