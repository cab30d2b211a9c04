/* Test driver (host code only): walks the same synthetic IPFIX batch with one
 * thread and with several, compares the plans, and is meant to be built with
 * -fsanitize=thread so that data races in the sharded pre-pass are reported.
 * Exporters re-send and change their templates in mid-stream. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "xe_internal.h"

static uint8_t *buf;
static size_t used, cap;
static uint64_t *offs;
static uint32_t *src;
static size_t n;

static void
put(const uint8_t *p, size_t len, uint32_t ip)
{
	if (used + len > cap) { cap = (cap ? cap * 2 : 1 << 20) + len; buf = realloc(buf, cap); }
	memcpy(buf + used, p, len);
	offs = realloc(offs, (n + 2) * sizeof(*offs));
	src = realloc(src, (n + 1) * sizeof(*src));
	offs[n] = used;
	used += len;
	offs[n + 1] = used;
	src[n] = ip;
	n++;
}

static void be16(uint8_t *p, unsigned v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void be32(uint8_t *p, unsigned v) { be16(p, v >> 16); be16(p + 2, v & 0xFFFF); }

static void
tmpl_pkt(uint32_t ip, unsigned domain, int wide)
{
	/* template 256: src addr, dst addr, ports, proto (+ octets when wide) */
	static const unsigned ids[] = { 8, 12, 7, 11, 4, 1 }, lens[] = { 4, 4, 2, 2, 1, 8 };
	unsigned nf = wide ? 6 : 5, i;
	uint8_t p[16 + 4 + 4 + 6 * 4];
	size_t len = 16 + 4 + 4 + nf * 4;
	memset(p, 0, sizeof(p));
	be16(p, 10); be16(p + 2, (unsigned)len); be32(p + 12, domain);
	be16(p + 16, 2); be16(p + 18, (unsigned)(len - 16));
	be16(p + 20, 256); be16(p + 22, nf);
	for (i = 0; i < nf; i++) { be16(p + 24 + 4 * i, ids[i]); be16(p + 26 + 4 * i, lens[i]); }
	put(p, len, ip);
}

static void
data_pkt(uint32_t ip, unsigned domain, int wide, unsigned nrec, unsigned seed)
{
	unsigned rl = wide ? 21 : 13, i;
	uint8_t p[16 + 4 + 64 * 21];
	size_t len = 16 + 4 + (size_t)nrec * rl;
	be16(p, 10); be16(p + 2, (unsigned)len); be32(p + 4, 0); be32(p + 8, seed); be32(p + 12, domain);
	be16(p + 16, 256); be16(p + 18, (unsigned)(len - 16));
	for (i = 0; i < nrec * rl; i++) p[20 + i] = (uint8_t)(seed * 2654435761u >> (i & 15)) ^ (uint8_t)i;
	put(p, len, ip);
}

static int
same(const struct xe_plan *a, const struct xe_plan *b)
{
	size_t i;
	if ((a->nrec != b->nrec) || (a->sets.n != b->sets.n) || (a->exporters.n != b->exporters.n)) return 0;
	if (a->skipped_datagrams != b->skipped_datagrams) return 0;
	if (memcmp(a->exporters.p, b->exporters.p, a->exporters.n * sizeof(struct xe_exporter))) return 0;
	for (i = 0; i < a->sets.n; i++) {
		struct xe_set x = ((struct xe_set *)a->sets.p)[i], y = ((struct xe_set *)b->sets.p)[i];
		x.tmpl = y.tmpl = 0;   /* slots may be numbered in another order */
		if (memcmp(&x, &y, sizeof(x))) return 0;
	}
	return 1;
}

int
main(void)
{
	struct xe_host h1, hN;
	struct xe_plan p1, pN;
	unsigned e, round, wide[48];
	int T, bad = 0;

	memset(wide, 0, sizeof(wide));
	for (round = 0; round < 160; round++) {
		for (e = 0; e < 48; e++) {
			uint32_t ip = 0x0100000au + (e << 24) + ((e * 7u) << 16);
			if ((round % 40) == (e % 40)) {
				if (round >= 40) wide[e] = !wide[e];       /* the template changes */
				tmpl_pkt(ip, e + 1, (int)wide[e]);
			}
			data_pkt(ip, e + 1, (int)wide[e], 1 + (round + e) % 60, round * 48 + e);
		}
	}
	xe_host_init(&h1);
	h1.prepass_threads = 1;
	if (xe_prepass(&h1, buf, offs, src, NULL, n, &p1) != XE_OK) { printf("single-thread walk failed\n"); return 1; }
	for (T = 2; T <= 12; T += 5) {
		int rep;
		xe_host_init(&hN);
		hN.prepass_threads = T;
		for (rep = 0; rep < 2; rep++) {
			if (xe_prepass(&hN, buf, offs, src, NULL, n, &pN) != XE_OK) { printf("walk with %d threads failed\n", T); return 1; }
			/* second walk of the same batch: templates known from the start, so
			 * only the first one must equal a fresh single-thread walk */
			if ((rep == 0) && !same(&p1, &pN)) { printf("plan with %d threads differs\n", T); bad = 1; }
			xe_plan_free(&pN);
		}
		xe_host_free(&hN);
	}
	printf("%zu datagrams, %zu records, %zu runs: %s\n", n, p1.nrec, p1.sets.n, bad ? "MISMATCH" : "same plans");
	xe_plan_free(&p1);
	xe_host_free(&h1);
	free(buf); free(offs); free(src);
	return bad;
}
