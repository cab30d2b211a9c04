/*
 * shim_bench.c -- datagrams through the reference's entry point
 * netflow_process() / sflow_process() (netflow.h:157, sflow.h:17) of the GPU
 * shim, with a stand-in for the exit call monit_object_process_nf()
 * (monit-objects.h:288) that reads the record it is handed and counts: what one
 * capture thread of a program that links the shim gets through.
 *
 * Usage: shim_bench <workload dir> <passes> <overlap 0|1> [max_datagrams]
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
#include "xenoeye_gpu_shim.h"

static uint64_t g_calls, g_sum;

static int
count_nf(struct xe_data *globl, struct monit_object *mo, size_t thread_id, uint64_t time_ns, struct flow_info *flow)
{
	/* a consumer reads its key and value fields: here the first 64 bytes of the record */
	const uint64_t *p = (const uint64_t *)flow;
	(void)globl; (void)thread_id; (void)time_ns;
	g_sum += p[0] ^ p[3] ^ p[7] ^ (uint64_t)(uintptr_t)mo;
	g_calls++;
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

int
main(int argc, char **argv)
{
	char path[4096], line[65536];
	xe_gpu *h = NULL;
	struct xe_shim_config cfg;
	struct monit_object **mos = NULL;
	size_t nmo = 0;
	FILE *f;
	uint64_t n, *offs;
	uint32_t *src;
	uint8_t *kind, *data;
	int rc, passes, overlap;
	static struct flow_packet_info fpi;

	if (argc < 4) DIE("usage: %s <dir> <passes> <overlap> [max_datagrams]", argv[0]);
	passes = atoi(argv[2]);
	overlap = atoi(argv[3]);
	if ((rc = xe_gpu_create(&h, 0)) != XE_OK) DIE("xe_gpu_create: %s", xe_gpu_strerror(rc));
	snprintf(path, sizeof(path), "%s/iplists", argv[1]);
	if ((rc = xe_gpu_load_iplists(h, path)) < 0) DIE("iplists: %s", xe_gpu_last_error(h));
	snprintf(path, sizeof(path), "%s/mos.txt", argv[1]);
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
	mos = calloc(nmo ? nmo : 1, sizeof(*mos));
	for (size_t i = 0; i < nmo; i++) mos[i] = (struct monit_object *)(uintptr_t)(i + 1);

	snprintf(path, sizeof(path), "%s/batch.bin", argv[1]);
	if (!(f = fopen(path, "rb"))) DIE("%s: cannot open", path);
	if (fread(&n, 8, 1, f) != 1) DIE("batch.bin: short");
	offs = malloc(8 * (n + 1));
	src = malloc(4 * n + 4);
	kind = malloc(n + 1);
	if ((fread(offs, 8, n + 1, f) != n + 1) || (fread(src, 4, n, f) != n) || (fread(kind, 1, n, f) != n)) DIE("batch.bin: short");
	data = malloc(offs[n] + 1);
	if (fread(data, 1, offs[n], f) != offs[n]) DIE("batch.bin: short");
	fclose(f);

	memset(&cfg, 0, sizeof(cfg));
	cfg.nthreads = 1;
	cfg.gpu = &h;
	cfg.mo = mos;
	cfg.nmo = nmo;
	cfg.process_nf = count_nf;
	cfg.overlap = overlap;
	if (argc > 4) cfg.max_datagrams = (size_t)atol(argv[4]);
	if (argc > 5) cfg.max_bytes = (size_t)atol(argv[5]);
	if ((rc = xenoeye_gpu_shim_init(&cfg)) != XE_OK) DIE("shim init: %s", xe_gpu_strerror(rc));
	netflow_process_init();

	/* what filling the caller's structure costs alone (recvfrom() does it in the reference) */
	double t0 = now();
	for (uint64_t i = 0; i < n; i++) {
		const size_t len = (size_t)(offs[i + 1] - offs[i]);
		fpi.src_addr_ipv4 = src[i];
		memcpy(fpi.rawpacket, data + offs[i], len);
		__asm__ volatile("" :: "r"(&fpi) : "memory");
	}
	const double fill_s = now() - t0;

	uint64_t st0[5], st1[5];
	for (int pass = -1; pass < passes; pass++) {
		if (pass == 0) {
			xenoeye_gpu_shim_flush(NULL, 0);
			xenoeye_gpu_shim_stats(0, st0);
			g_calls = 0;
			t0 = now();
		}
		for (uint64_t i = 0; i < n; i++) {
			const int len = (int)(offs[i + 1] - offs[i]);
			fpi.src_addr_ipv4 = src[i];
			memcpy(fpi.rawpacket, data + offs[i], (size_t)len);
			if (kind[i]) sflow_process(NULL, 0, &fpi, len);
			else netflow_process(NULL, 0, &fpi, len);
		}
	}
	xenoeye_gpu_shim_flush(NULL, 0);
	const double s = now() - t0;
	xenoeye_gpu_shim_stats(0, st1);
	uint64_t ns[4];
	xenoeye_gpu_shim_times(0, ns);
	printf("{\"datagrams\": %llu, \"records\": %llu, \"exit_calls\": %llu, \"batches\": %llu, \"failed_batches\": %llu, "
		"\"seconds\": %.4f, \"datagrams_per_s\": %.0f, \"records_per_s\": %.0f, \"exit_calls_per_s\": %.0f, "
		"\"fill_only_seconds_per_pass\": %.4f, \"overlap\": %d, \"checksum\": %llu, "
		"\"seconds_by_part_with_warmup_pass\": {\"prepare_and_start\": %.4f, \"wait_and_copy_back\": %.4f, \"read_outs\": %.4f, "
		"\"records_and_exit_calls\": %.4f}}\n",
		(unsigned long long)(st1[0] - st0[0]), (unsigned long long)(st1[2] - st0[2]), (unsigned long long)g_calls,
		(unsigned long long)(st1[1] - st0[1]), (unsigned long long)(st1[4] - st0[4]), s,
		(st1[0] - st0[0]) / s, (st1[2] - st0[2]) / s, g_calls / s, fill_s, overlap, (unsigned long long)g_sum,
		ns[0] * 1e-9, ns[1] * 1e-9, ns[2] * 1e-9, ns[3] * 1e-9);
	if (st1[4]) fprintf(stderr, "error: %s\n", xenoeye_gpu_shim_error(0));
	xenoeye_gpu_shim_shutdown();
	xe_gpu_destroy(h);
	return 0;
}
