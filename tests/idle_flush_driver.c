/*
 * idle_flush_driver.c -- test program: the shim's idle flush under a capture
 * loop of the reference's shape, without a GPU.
 *
 * The loop below has the shape of scapture_thread() (scapture.c:25-61): a
 * blocking recvfrom() on a UDP socket, `continue` on an error, the entry call
 * on every datagram, and NO idle branch.  The flow-path library is replaced by
 * stand-ins in this file (every datagram is one record that object 0
 * receives), so only the shim's batching and its wake-up path are under test:
 * datagrams arrive, the exporter falls silent, and the exit calls must still
 * come - on the capture thread - once the batch is due.
 *
 * Prints one line: "calls=<n> on_capture_thread=<0|1> wakeups=<n> before_due=<n>".
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <netinet/in.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "xenoeye_gpu_shim.h"

/* ---- stand-ins for the library calls the shim makes */
struct xe_gpu { int unused; };
struct xe_batch { size_t n; };

const char *xe_gpu_strerror(int code) { (void)code; return "stub"; }
const char *xe_gpu_last_error(const xe_gpu *h) { (void)h; return ""; }
int xe_gpu_mo_count(const xe_gpu *h) { (void)h; return 1; }
int xe_gpu_mo_parent(const xe_gpu *h, int mo) { (void)h; (void)mo; return -1; }
int xe_gpu_mo_host_only(const xe_gpu *h, int mo) { (void)h; (void)mo; return 0; }
void *xe_gpu_host_alloc(xe_gpu *h, size_t bytes) { (void)h; return malloc(bytes); }
void xe_gpu_host_free(xe_gpu *h, void *p) { (void)h; free(p); }

int
xe_gpu_batch_prepare(xe_gpu *h, const uint8_t *data, const uint64_t *offsets, const uint32_t *src_ip4,
	const uint8_t *kinds, size_t n, xe_batch **out)
{
	(void)h; (void)data; (void)offsets; (void)src_ip4; (void)kinds;
	*out = calloc(1, sizeof(**out));
	if (!*out) return XE_ENOMEM;
	(*out)->n = n;
	return XE_OK;
}

void xe_gpu_batch_free(xe_batch *b) { free(b); }
int xe_gpu_batch_run(xe_batch *b, void *stream) { (void)b; (void)stream; return XE_OK; }
int xe_gpu_batch_done(xe_batch *b) { (void)b; return 1; }
int xe_gpu_set_overlap(xe_gpu *h, int on) { (void)h; (void)on; return XE_OK; }
size_t xe_gpu_batch_records(const xe_batch *b) { return b->n; }

int
xe_gpu_batch_masks(xe_batch *b, uint64_t *masks, size_t nwords_total)
{
	size_t i;
	for (i = 0; (i < b->n) && (i < nwords_total); i++) masks[i] = 1;
	return XE_OK;
}

int xe_gpu_batch_dropped(xe_batch *b, uint8_t *dropped, size_t n) { (void)b; memset(dropped, 0, n); return XE_OK; }

int
xe_gpu_batch_record_datagram(xe_batch *b, uint32_t *dgram, size_t n)
{
	size_t i;
	for (i = 0; (i < b->n) && (i < n); i++) dgram[i] = (uint32_t)i;
	return XE_OK;
}

struct xe_flowcur { int unused; };
static struct xe_flowcur the_cursor;

int
xe_gpu_batch_flow_cursor(xe_batch *b, void *flow, xe_flowcur **out)
{
	(void)b;
	memset(flow, 0, XE_FLOW_INFO_SIZE);
	*out = &the_cursor;
	return XE_OK;
}

int xe_gpu_flow_cursor_seek(xe_flowcur *cur, size_t r) { (void)cur; (void)r; return XE_OK; }
void xe_gpu_flow_cursor_free(xe_flowcur *cur) { (void)cur; }

/* ---- the consumer and the capture loop */
static pthread_t capture_tid;
static int calls, calls_elsewhere;
#define NCALLS() __atomic_load_n(&calls, __ATOMIC_SEQ_CST)
static int sockfd;

static int
process_nf(struct xe_data *globl, struct monit_object *mo, size_t thread_id, uint64_t time_ns, struct flow_info *flow)
{
	(void)globl; (void)mo; (void)thread_id; (void)time_ns; (void)flow;
	if (!pthread_equal(pthread_self(), capture_tid)) __atomic_add_fetch(&calls_elsewhere, 1, __ATOMIC_SEQ_CST);
	__atomic_add_fetch(&calls, 1, __ATOMIC_SEQ_CST);
	return 1;
}

static void *
capture_loop(void *arg)
{
	socklen_t clientlen = sizeof(struct sockaddr);
	struct flow_packet_info *pkt = malloc(sizeof(*pkt));

	(void)arg;
	for (;;) {
		ssize_t len = recvfrom(sockfd, pkt->rawpacket, MAX_NF_PACKET_SIZE, 0, &pkt->src_addr, &clientlen);
		if (len < 0) continue;
		if (pkt->src_addr.sa_family == AF_INET) {
			pkt->src_addr_ipv4 = *((uint32_t *)&((struct sockaddr_in *)&pkt->src_addr)->sin_addr);
		} else {
			pkt->src_addr_ipv4 = 0;
		}
		netflow_process(NULL, 0, pkt, (int)len);
	}
	return NULL;
}

static void
sleep_ms(long ms)
{
	struct timespec ts = { ms / 1000, (ms % 1000) * 1000000l };
	nanosleep(&ts, NULL);
}

int
main(void)
{
	struct sockaddr_in addr;
	socklen_t alen = sizeof(addr);
	struct xe_shim_config cfg;
	struct xe_gpu handle;
	xe_gpu *handles[1] = { &handle };
	struct monit_object *mos[1] = { (struct monit_object *)&handle };
	uint16_t port;
	uint8_t v9[24] = { 0, 9, 0, 0 };
	int out, i, before_due;

	sockfd = socket(AF_INET, SOCK_DGRAM, 0);
	memset(&addr, 0, sizeof(addr));
	addr.sin_family = AF_INET;
	addr.sin_addr.s_addr = htonl(INADDR_ANY);      /* scapture.c:143 */
	addr.sin_port = 0;
	if ((sockfd < 0) || bind(sockfd, (struct sockaddr *)&addr, sizeof(addr)) || getsockname(sockfd, (struct sockaddr *)&addr, &alen)) {
		printf("no socket\n");
		return 2;
	}
	port = ntohs(addr.sin_port);

	memset(&cfg, 0, sizeof(cfg));
	cfg.nthreads = 1;
	cfg.gpu = handles;
	cfg.mo = mos;
	cfg.nmo = 1;
	cfg.process_nf = process_nf;
	cfg.max_wait_ns = 50000000ull;                 /* 50 ms */
	cfg.wake_port = &port;
	if (xenoeye_gpu_shim_init(&cfg) != XE_OK) { printf("init failed\n"); return 2; }
	pthread_create(&capture_tid, NULL, capture_loop, NULL);

	/* three datagrams, then silence */
	out = socket(AF_INET, SOCK_DGRAM, 0);
	addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
	for (i = 0; i < 3; i++) sendto(out, v9, sizeof(v9), 0, (struct sockaddr *)&addr, sizeof(addr));
	sleep_ms(10);
	before_due = NCALLS();                           /* nothing is due yet */
	for (i = 0; (i < 200) && (NCALLS() < 3); i++) sleep_ms(10);
	sleep_ms(120);                                 /* no second delivery of the same batch */
	printf("calls=%d on_capture_thread=%d wakeups=%llu before_due=%d\n", NCALLS(),
		__atomic_load_n(&calls_elsewhere, __ATOMIC_SEQ_CST) ? 0 : 1,
		(unsigned long long)xenoeye_gpu_shim_wakeups(), before_due);
	/* the capture thread blocks in recvfrom() for good, as the reference's does: leave without it */
	fflush(stdout);
	_exit(0);
}
