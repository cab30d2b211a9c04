/*
 * shim_bench.c -- datagrams through the reference's entry point
 * netflow_process() / sflow_process() (netflow.h:157, sflow.h:17) of the GPU
 * shim, with a stand-in for the exit call monit_object_process_nf()
 * (monit-objects.h:288) that reads the record it is handed and counts: what one
 * capture thread of a program that links the shim gets through.
 *
 * Usage: shim_bench <workload dir> <passes> <overlap 0|1> [capture threads [max_datagrams [max_bytes]]]
 * The directory (written by tools/shimbench/run.py) holds
 *     batch.bin   u64 n, u64 offsets[n + 1], u32 src[n], u8 kind[n], datagram bytes
 *     mos.txt     one object per line: parent index, tab, filter text
 *     iplists/    one file per list
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include "xenoeye_gpu_shim.h"

#define MAX_THREADS 16
/* per capture thread (the consumers' state is per thread_id, monit-objects.c:923-1033) */
static struct { uint64_t calls, sum; char pad[48]; } g_cnt[MAX_THREADS];

static int
count_nf(struct xe_data *globl, struct monit_object *mo, size_t thread_id, uint64_t time_ns, struct flow_info *flow)
{
	/* a consumer reads its key and value fields: here the first 64 bytes of the record */
	const uint64_t *p = (const uint64_t *)flow;
	(void)globl; (void)time_ns;
	g_cnt[thread_id].sum += p[0] ^ p[3] ^ p[7] ^ (uint64_t)(uintptr_t)mo;
	g_cnt[thread_id].calls++;
	return 1;
}

static double
now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

static uint64_t g_n, *g_offs;
static uint32_t *g_src;
static uint8_t *g_kind, *g_data;
static int g_passes;
static pthread_barrier_t g_bar;
static double g_t0[MAX_THREADS], g_t1[MAX_THREADS];
static uint64_t g_st0[MAX_THREADS][5], g_st1[MAX_THREADS][5];

/* one capture thread: the loop of scapture.c:25-61 with the datagrams of the file in place of recvfrom() */
static void *
capture_main(void *arg)
{
	const size_t tid = (size_t)(uintptr_t)arg;
	struct flow_packet_info *fpi = calloc(1, sizeof(*fpi));

	for (int pass = -1; pass < g_passes; pass++) {
		if (pass == 0) {
			xenoeye_gpu_shim_flush(NULL, tid);
			xenoeye_gpu_shim_stats(tid, g_st0[tid]);
			g_cnt[tid].calls = 0;
			pthread_barrier_wait(&g_bar);
			g_t0[tid] = now();
		}
		for (uint64_t i = 0; i < g_n; i++) {
			const int len = (int)(g_offs[i + 1] - g_offs[i]);
			fpi->src_addr_ipv4 = g_src[i];
			memcpy(fpi->rawpacket, g_data + g_offs[i], (size_t)len);
			if (g_kind[i]) sflow_process(NULL, tid, fpi, len);
			else netflow_process(NULL, tid, fpi, len);
		}
	}
	xenoeye_gpu_shim_flush(NULL, tid);
	g_t1[tid] = now();
	xenoeye_gpu_shim_stats(tid, g_st1[tid]);
	free(fpi);
	return NULL;
}

static xe_gpu *
make_handle(const char *dir, size_t *nmo_out)
{
	char path[4096], line[65536];
	xe_gpu *h = NULL;
	size_t nmo = 0;
	FILE *f;
	int rc;

	if ((rc = xe_gpu_create(&h, 0)) != XE_OK) DIE("xe_gpu_create: %s", xe_gpu_strerror(rc));
	snprintf(path, sizeof(path), "%s/iplists", dir);
	if ((rc = xe_gpu_load_iplists(h, path)) < 0) DIE("iplists: %s", xe_gpu_last_error(h));
	snprintf(path, sizeof(path), "%s/mos.txt", dir);
	if (!(f = fopen(path, "r"))) DIE("%s: cannot open", path);
	while (fgets(line, sizeof(line), f)) {
		char *tab = strchr(line, '\t');
		if (!tab) continue;
		*tab++ = 0;
		tab[strcspn(tab, "\n")] = 0;
		if ((rc = xe_gpu_add_mo(h, "mo", atoi(line), tab)) < 0) DIE("object %zu: %s", nmo, xe_gpu_last_error(h));
		nmo++;
	}
	fclose(f);
	if ((rc = xe_gpu_commit(h)) != XE_OK) DIE("commit: %s", xe_gpu_last_error(h));
	*nmo_out = nmo;
	return h;
}

int
main(int argc, char **argv)
{
	char path[4096];
	xe_gpu *hs[MAX_THREADS];
	pthread_t th[MAX_THREADS];
	struct xe_shim_config cfg;
	struct monit_object **mos = NULL;
	size_t nmo = 0, nthreads = 1;
	FILE *f;
	int rc, overlap;
	static struct flow_packet_info fpi;

	if (argc < 4) DIE("usage: %s <dir> <passes> <overlap> [threads [max_datagrams [max_bytes]]]", argv[0]);
	g_passes = atoi(argv[2]);
	overlap = atoi(argv[3]);
	if (argc > 4) nthreads = (size_t)atol(argv[4]);
	if ((nthreads < 1) || (nthreads > MAX_THREADS)) DIE("1 to %d threads", MAX_THREADS);
	/* a handle per capture thread: its own template store, its own streams */
	for (size_t t = 0; t < nthreads; t++) hs[t] = make_handle(argv[1], &nmo);
	mos = calloc(nmo ? nmo : 1, sizeof(*mos));
	for (size_t i = 0; i < nmo; i++) mos[i] = (struct monit_object *)(uintptr_t)(i + 1);

	snprintf(path, sizeof(path), "%s/batch.bin", argv[1]);
	if (!(f = fopen(path, "rb"))) DIE("%s: cannot open", path);
	if (fread(&g_n, 8, 1, f) != 1) DIE("batch.bin: short");
	g_offs = malloc(8 * (g_n + 1));
	g_src = malloc(4 * g_n + 4);
	g_kind = malloc(g_n + 1);
	if ((fread(g_offs, 8, g_n + 1, f) != g_n + 1) || (fread(g_src, 4, g_n, f) != g_n) || (fread(g_kind, 1, g_n, f) != g_n)) DIE("batch.bin: short");
	g_data = malloc(g_offs[g_n] + 1);
	if (fread(g_data, 1, g_offs[g_n], f) != g_offs[g_n]) DIE("batch.bin: short");
	fclose(f);

	memset(&cfg, 0, sizeof(cfg));
	cfg.nthreads = nthreads;
	cfg.gpu = hs;
	cfg.mo = mos;
	cfg.nmo = nmo;
	cfg.process_nf = count_nf;
	cfg.overlap = overlap;
	if (argc > 5) cfg.max_datagrams = (size_t)atol(argv[5]);
	if (argc > 6) cfg.max_bytes = (size_t)atol(argv[6]);
	if ((rc = xenoeye_gpu_shim_init(&cfg)) != XE_OK) DIE("shim init: %s", xe_gpu_strerror(rc));
	netflow_process_init();

	/* what filling the caller's structure costs alone (recvfrom() does it in the reference) */
	double t0 = now();
	for (uint64_t i = 0; i < g_n; i++) {
		const size_t len = (size_t)(g_offs[i + 1] - g_offs[i]);
		fpi.src_addr_ipv4 = g_src[i];
		memcpy(fpi.rawpacket, g_data + g_offs[i], len);
		__asm__ volatile("" :: "r"(&fpi) : "memory");
	}
	const double fill_s = now() - t0;

	pthread_barrier_init(&g_bar, NULL, (unsigned)nthreads);
	for (size_t t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, capture_main, (void *)(uintptr_t)t);
	for (size_t t = 0; t < nthreads; t++) pthread_join(th[t], NULL);

	double first = g_t0[0], last = g_t1[0];
	uint64_t dg = 0, recs = 0, calls = 0, batches = 0, failed = 0, sum = 0, ns[4] = {0, 0, 0, 0};
	for (size_t t = 0; t < nthreads; t++) {
		uint64_t x[4];
		if (g_t0[t] < first) first = g_t0[t];
		if (g_t1[t] > last) last = g_t1[t];
		dg += g_st1[t][0] - g_st0[t][0];
		batches += g_st1[t][1] - g_st0[t][1];
		recs += g_st1[t][2] - g_st0[t][2];
		failed += g_st1[t][4] - g_st0[t][4];
		calls += g_cnt[t].calls;
		sum ^= g_cnt[t].sum;
		xenoeye_gpu_shim_times(t, x);
		for (int k = 0; k < 4; k++) ns[k] += x[k];
		if (g_st1[t][4]) fprintf(stderr, "thread %zu: %s\n", t, xenoeye_gpu_shim_error(t));
	}
	const double s = last - first;
	printf("{\"capture_threads\": %zu, \"datagrams\": %llu, \"records\": %llu, \"exit_calls\": %llu, \"batches\": %llu, \"failed_batches\": %llu, "
		"\"seconds\": %.4f, \"datagrams_per_s\": %.0f, \"records_per_s\": %.0f, \"exit_calls_per_s\": %.0f, "
		"\"fill_only_seconds_per_pass\": %.4f, \"overlap\": %d, \"checksum\": %llu, "
		"\"thread_seconds_by_part_with_warmup_pass\": {\"prepare_and_start\": %.4f, \"wait_and_copy_back\": %.4f, \"read_outs\": %.4f, "
		"\"records_and_exit_calls\": %.4f}}\n",
		nthreads, (unsigned long long)dg, (unsigned long long)recs, (unsigned long long)calls,
		(unsigned long long)batches, (unsigned long long)failed, s, dg / s, recs / s, calls / s, fill_s, overlap,
		(unsigned long long)sum, ns[0] * 1e-9, ns[1] * 1e-9, ns[2] * 1e-9, ns[3] * 1e-9);
	xenoeye_gpu_shim_shutdown();
	for (size_t t = 0; t < nthreads; t++) xe_gpu_destroy(hs[t]);
	return 0;
}
