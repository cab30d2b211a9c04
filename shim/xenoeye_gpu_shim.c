/*
 * xenoeye_gpu_shim.c -- netflow_process() / sflow_process() of the reference
 * (netflow.h:156-158, sflow.h:17-19) as a batching front of the GPU flow path,
 * and the replay of its results into monit_object_process_nf()
 * (monit-objects.h:288-289).  See include/xenoeye_gpu_shim.h for the contract.
 *
 * Plain C over the C ABI of include/xe_gpu.h; nothing here decodes a record or
 * evaluates a filter: records come back from the library as match masks and
 * re-created struct flow_info bytes, and the only filters evaluated on the
 * host are those of objects the library reports as host-only, through the
 * caller's own filter_match() (filter.h:331).
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "xenoeye_gpu_shim.h"

/* resolved by the program the shim is linked into, unless the configuration
 * names other functions */
extern int monit_object_process_nf(struct xe_data *globl, struct monit_object *mo, size_t thread_id,
	uint64_t time_ns, struct flow_info *flow) __attribute__((weak));
extern int filter_match(struct filter_expr *expr, struct flow_info *flow) __attribute__((weak));


struct shim_thread
{
	xe_gpu *h;
	pthread_mutex_t *lock;       /* not NULL: the handle is shared with another thread */
	/* the batch being collected */
	uint8_t *data;               /* datagrams back to back, page-locked */
	size_t used, cap;
	uint64_t *off;               /* n + 1 offsets */
	uint32_t *src;
	uint8_t *kind;
	uint64_t *time_ns;           /* arrival time of every datagram */
	size_t n;
	/* overlap: the batch that runs on the GPU while the next one is collected
	 * (its datagram buffer and arrival times stay with it until it is delivered) */
	xe_batch *pend;
	uint8_t *pend_data;
	uint64_t *pend_time_ns;
	unsigned since_query;
	uint64_t first_mono;         /* monotonic clock at the arrival of the batch's first datagram */
	/* when the batch being collected is due (0: nothing collected); written by
	 * the capture thread, read by the wake-up thread */
	uint64_t due_at;
	uint64_t woken_for;          /* wake-up thread only: the due_at it has sent a datagram for */
	/* results of the batch being delivered */
	uint64_t *masks;
	uint8_t *status;
	uint32_t *dgram;
	size_t rec_cap, mask_cap;
	uint64_t ns[4];              /* time spent: see xenoeye_gpu_shim_times() */
	uint8_t *flows;              /* the one struct flow_info the exit calls are handed: XE_FLOW_INFO_SIZE bytes */
	uint64_t stats[5];
	char err[320];
};

static struct
{
	int ready;
	struct xe_shim_config cfg;
	struct shim_thread *th;
	pthread_mutex_t *locks;      /* one per distinct handle that several threads use */
	size_t nlocks;
	struct monit_object **mo;
	struct filter_expr **expr;
	uint8_t *host_only;          /* per object */
	uint32_t *subtree_end;       /* per object: index behind its last descendant */
	size_t nmo, mask_words;
	xe_shim_process_nf_fn process_nf;
	xe_shim_filter_match_fn filter_match;
	/* idle flush: a thread that sends a wake-up datagram to the capture socket
	 * of a thread whose batch is due */
	uint16_t *wake_port;
	pthread_t waker;
	int waker_on, wake_fd;
	pthread_mutex_t wake_mu;
	pthread_cond_t wake_cv;
	int wake_stop;
	uint64_t wake_sent;
} S;

/* the wake-up datagram: too short and of no version the path takes, so a
 * reference build that received it would log and leave it (netflow.c:804-819);
 * the shim recognises it and batches nothing */
static const uint8_t wake_magic[16] = { 0, 0, 'x', 'e', '-', 'g', 'p', 'u', '-', 'w', 'a', 'k', 'e', 0, 0, 1 };

static uint64_t
mono_ns(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC_COARSE, &ts);
	return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* for the time accounting per batch (the coarse clock above is read per datagram) */
static uint64_t
fine_ns(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* The reference's capture loop blocks in recvfrom() and has no idle branch
 * (scapture.c:25-61), so a thread whose exporters fall silent would keep its
 * last partial batch.  This thread only wakes the owner: it sends one small
 * datagram to the owner's own capture socket over the loopback; recvfrom()
 * returns, the loop calls netflow_process() / sflow_process() as for any
 * datagram, and the shim - on the capture thread, where the exit calls belong -
 * finds the batch due and runs it. */
static void *
waker_main(void *arg)
{
	(void)arg;
	pthread_mutex_lock(&S.wake_mu);
	while (!S.wake_stop) {
		struct timespec ts;
		uint64_t step = S.cfg.max_wait_ns / 2, now;
		size_t t;

		if (step < 1000000ull) step = 1000000ull;
		clock_gettime(CLOCK_REALTIME, &ts);
		ts.tv_nsec += (long)(step % 1000000000ull);
		ts.tv_sec += (time_t)(step / 1000000000ull) + ts.tv_nsec / 1000000000l;
		ts.tv_nsec %= 1000000000l;
		pthread_cond_timedwait(&S.wake_cv, &S.wake_mu, &ts);
		if (S.wake_stop) break;
		now = mono_ns();
		for (t = 0; t < S.cfg.nthreads; t++) {
			struct shim_thread *T = &S.th[t];
			const uint64_t due = __atomic_load_n(&T->due_at, __ATOMIC_ACQUIRE);
			struct sockaddr_in to;

			if (!S.wake_port[t] || !due || (now < due) || (T->woken_for == due)) continue;
			memset(&to, 0, sizeof(to));
			to.sin_family = AF_INET;
			to.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
			to.sin_port = htons(S.wake_port[t]);
			if (sendto(S.wake_fd, wake_magic, sizeof(wake_magic), MSG_DONTWAIT, (struct sockaddr *)&to, sizeof(to))
				== (ssize_t)sizeof(wake_magic)) {
				T->woken_for = due;
				__atomic_add_fetch(&S.wake_sent, 1, __ATOMIC_RELAXED);
			}
		}
	}
	pthread_mutex_unlock(&S.wake_mu);
	return NULL;
}

static void
waker_stop(void)
{
	if (S.waker_on) {
		pthread_mutex_lock(&S.wake_mu);
		S.wake_stop = 1;
		pthread_cond_signal(&S.wake_cv);
		pthread_mutex_unlock(&S.wake_mu);
		pthread_join(S.waker, NULL);
		pthread_mutex_destroy(&S.wake_mu);
		pthread_cond_destroy(&S.wake_cv);
		close(S.wake_fd);
		S.waker_on = 0;
	}
	free(S.wake_port);
	S.wake_port = NULL;
}

void
xenoeye_gpu_shim_shutdown(void)
{
	size_t t;

	waker_stop();
	if (S.th) {
		for (t = 0; t < S.cfg.nthreads; t++) {
			struct shim_thread *T = &S.th[t];
			if (T->pend) xe_gpu_batch_free(T->pend);
			if (S.cfg.overlap && T->h) (void)xe_gpu_set_overlap(T->h, 0);
			if (T->data) xe_gpu_host_free(T->h, T->data);
			if (T->pend_data) xe_gpu_host_free(T->h, T->pend_data);
			free(T->off); free(T->src); free(T->kind); free(T->time_ns); free(T->pend_time_ns);
			free(T->masks); free(T->status); free(T->dgram); free(T->flows);
		}
	}
	free(S.th);
	for (t = 0; t < S.nlocks; t++) pthread_mutex_destroy(&S.locks[t]);
	free(S.locks);
	free(S.mo); free(S.expr); free(S.host_only); free(S.subtree_end);
	memset(&S, 0, sizeof(S));
}

int
xenoeye_gpu_shim_init(const struct xe_shim_config *cfg)
{
	size_t t, m, u;
	int rc = XE_ENOMEM;

	if (!cfg || !cfg->nthreads || !cfg->gpu || (cfg->nmo && !cfg->mo)) return XE_EINVAL;
	if (S.ready) xenoeye_gpu_shim_shutdown();
	S.cfg = *cfg;
	if (!S.cfg.max_datagrams) S.cfg.max_datagrams = 16384;
	if (!S.cfg.max_bytes) S.cfg.max_bytes = 32u << 20;
	if (S.cfg.max_bytes < MAX_NF_PACKET_SIZE) S.cfg.max_bytes = MAX_NF_PACKET_SIZE;
	if (!S.cfg.max_wait_ns) S.cfg.max_wait_ns = 20000000ull;
	S.process_nf = cfg->process_nf ? cfg->process_nf : monit_object_process_nf;
	S.filter_match = cfg->filter_match ? cfg->filter_match : filter_match;
	if (!S.process_nf) { rc = XE_EINVAL; goto fail; }
	S.nmo = cfg->nmo;
	S.mask_words = S.nmo ? (S.nmo + 63) / 64 : 1;

	S.mo = calloc(S.nmo ? S.nmo : 1, sizeof(*S.mo));
	S.expr = calloc(S.nmo ? S.nmo : 1, sizeof(*S.expr));
	S.host_only = calloc(S.nmo ? S.nmo : 1, 1);
	S.subtree_end = calloc(S.nmo ? S.nmo : 1, sizeof(uint32_t));
	S.th = calloc(cfg->nthreads, sizeof(*S.th));
	S.locks = calloc(cfg->nthreads, sizeof(*S.locks));
	if (!S.mo || !S.expr || !S.host_only || !S.subtree_end || !S.th || !S.locks) goto fail;

	for (t = 0; t < cfg->nthreads; t++) {
		if (!cfg->gpu[t] || ((size_t)xe_gpu_mo_count(cfg->gpu[t]) != S.nmo)) { rc = XE_EINVAL; goto fail; }
		/* every handle must see the objects as the first one does: the same tree,
		 * the same objects left to filter_match() (a handle without the port names
		 * or databases of another would hand out candidate bits as matches) */
		for (m = 0; (t > 0) && (m < S.nmo); m++) {
			if ((xe_gpu_mo_parent(cfg->gpu[t], (int)m) != xe_gpu_mo_parent(cfg->gpu[0], (int)m))
				|| (xe_gpu_mo_host_only(cfg->gpu[t], (int)m) != xe_gpu_mo_host_only(cfg->gpu[0], (int)m))) {
				rc = XE_EINVAL;
				goto fail;
			}
		}
	}
	for (m = 0; m < S.nmo; m++) {
		const int parent = xe_gpu_mo_parent(cfg->gpu[0], (int)m);
		S.mo[m] = cfg->mo[m];
		S.expr[m] = cfg->expr ? cfg->expr[m] : NULL;
		S.host_only[m] = (uint8_t)(xe_gpu_mo_host_only(cfg->gpu[0], (int)m) == 1);
		/* bits low to high must be the reference's pre-order: the parent of an
		 * object is the object before it or one of that object's ancestors */
		if (parent >= (int)m) { rc = XE_EINVAL; goto fail; }
		if (m && (parent != (int)m - 1)) {
			int a = xe_gpu_mo_parent(cfg->gpu[0], (int)m - 1);
			while ((a >= 0) && (a != parent)) a = xe_gpu_mo_parent(cfg->gpu[0], a);
			if (a != parent) { rc = XE_EINVAL; goto fail; }
		}
		if (S.host_only[m] && (!S.expr[m] || !S.filter_match)) { rc = XE_EINVAL; goto fail; }
	}
	for (m = 0; m < S.nmo; m++) {
		for (u = m + 1; (u < S.nmo) && (xe_gpu_mo_parent(cfg->gpu[0], (int)u) >= (int)m); u++) {}
		S.subtree_end[m] = (uint32_t)u;
	}

	for (t = 0; t < cfg->nthreads; t++) {
		struct shim_thread *T = &S.th[t];
		const size_t nd = S.cfg.max_datagrams;
		T->h = cfg->gpu[t];
		for (u = 0; u < t; u++) {
			if (S.th[u].h != T->h) continue;
			if (!S.th[u].lock) {
				pthread_mutex_init(&S.locks[S.nlocks], NULL);
				S.th[u].lock = &S.locks[S.nlocks++];
			}
			T->lock = S.th[u].lock;
			break;
		}
		T->cap = S.cfg.max_bytes;
		T->data = xe_gpu_host_alloc(T->h, T->cap + 64);
		T->off = calloc(nd + 1, sizeof(uint64_t));
		T->src = calloc(nd, sizeof(uint32_t));
		T->kind = calloc(nd, 1);
		T->time_ns = calloc(nd, sizeof(uint64_t));
		T->flows = aligned_alloc(64, (XE_FLOW_INFO_SIZE + 63) & ~(size_t)63);
		if (!T->data || !T->off || !T->src || !T->kind || !T->time_ns || !T->flows) goto fail;
		if (S.cfg.overlap) {
			T->pend_data = xe_gpu_host_alloc(T->h, T->cap + 64);
			T->pend_time_ns = calloc(nd, sizeof(uint64_t));
			if (!T->pend_data || !T->pend_time_ns) goto fail;
			if ((rc = xe_gpu_set_overlap(T->h, 1)) != XE_OK) goto fail;
			rc = XE_ENOMEM;
		}
	}
	if (cfg->wake_port) {
		int any = 0;
		S.wake_port = calloc(cfg->nthreads, sizeof(uint16_t));
		if (!S.wake_port) goto fail;
		for (t = 0; t < cfg->nthreads; t++) any |= (S.wake_port[t] = cfg->wake_port[t]) != 0;
		S.cfg.wake_port = S.wake_port;
		if (any) {
			S.wake_fd = socket(AF_INET, SOCK_DGRAM, 0);
			if (S.wake_fd < 0) { rc = XE_EINVAL; goto fail; }
			pthread_mutex_init(&S.wake_mu, NULL);
			pthread_cond_init(&S.wake_cv, NULL);
			S.wake_stop = 0;
			if (pthread_create(&S.waker, NULL, waker_main, NULL)) {
				pthread_mutex_destroy(&S.wake_mu);
				pthread_cond_destroy(&S.wake_cv);
				close(S.wake_fd);
				rc = XE_ENOMEM;
				goto fail;
			}
			S.waker_on = 1;
		}
	}
	S.ready = 1;
	return XE_OK;
fail:
	xenoeye_gpu_shim_shutdown();
	return rc;
}

/* the exit calls of one batch, on the calling thread: records in index order
 * (= arrival order: a batch holds datagrams of one kind), objects low to high
 * (= the reference's pre-order, netflow.c:213-244) */
static int
deliver(struct xe_data *globl, size_t thread_id, struct shim_thread *T, xe_batch *b, size_t nrec, const uint64_t *time_ns)
{
	const size_t w = S.mask_words;
	xe_flowcur *cur = NULL;
	uint64_t t0, t1;
	size_t r;
	int rc;

	if (nrec > T->rec_cap) {
		const size_t cap = nrec + nrec / 4 + 1024;
		free(T->masks); free(T->status); free(T->dgram);
		T->masks = malloc(cap * w * sizeof(uint64_t));
		T->status = malloc(cap);
		T->dgram = malloc(cap * sizeof(uint32_t));
		T->rec_cap = (T->masks && T->status && T->dgram) ? cap : 0;
		if (!T->rec_cap) return XE_ENOMEM;
	}
	/* all outputs of the batch in one copy; the reads below come out of it */
	t0 = fine_ns();
	if ((rc = xe_gpu_batch_flow_cursor(b, T->flows, &cur)) != XE_OK) return rc;   /* waits for the kernels */
	t1 = fine_ns();
	T->ns[1] += t1 - t0;
	if ((rc = xe_gpu_batch_masks(b, T->masks, nrec * w)) != XE_OK) goto out;
	if ((rc = xe_gpu_batch_dropped(b, T->status, nrec)) != XE_OK) goto out;
	if ((rc = xe_gpu_batch_record_datagram(b, T->dgram, nrec)) != XE_OK) goto out;
	t0 = fine_ns();
	T->ns[2] += t0 - t1;

	for (r = 0; r < nrec; r++) {
		struct flow_info *flow = (struct flow_info *)T->flows;
		const uint64_t *mk = &T->masks[r * w];
		uint64_t any = 0;
		size_t k, m, skip_to;

		if (T->status[r]) continue;    /* dropped by device rules / sample never classified */
		for (k = 0; k < w; k++) any |= mk[k];
		if (!any) continue;            /* no object receives it */
		/* the one struct flow_info becomes record r, as netflow.c:273-275 keeps one per record */
		if ((rc = xe_gpu_flow_cursor_seek(cur, r)) != XE_OK) goto out;
		skip_to = 0;
		for (k = 0; k < w; k++) {
			uint64_t x = mk[k];
			while (x) {
				m = 64 * k + (size_t)__builtin_ctzll(x);
				x &= x - 1;
				if ((m < skip_to) || (m >= S.nmo)) continue;
				if (S.host_only[m] && !S.filter_match(S.expr[m], flow)) {
					/* the object does not receive the record, and the objects
					 * below it are not even looked at (netflow.c:223-243) */
					skip_to = S.subtree_end[m];
					continue;
				}
				S.process_nf(globl, S.mo[m], thread_id, time_ns[T->dgram[r]], flow);
				T->stats[3]++;
			}
		}
	}
out:
	T->ns[3] += fine_ns() - t0;
	xe_gpu_flow_cursor_free(cur);
	return rc;
}

static void
batch_lost(struct shim_thread *T, size_t n, int rc)
{
	T->stats[4]++;
	snprintf(T->err, sizeof(T->err), "batch of %zu datagrams lost: %s: %.200s", n, xe_gpu_strerror(rc),
		xe_gpu_last_error(T->h));
}

/* overlap: the exit calls of the batch that was left running (caller holds the handle's lock) */
static int
deliver_pending(struct xe_data *globl, size_t thread_id, struct shim_thread *T)
{
	int rc;

	if (!T->pend) return XE_OK;
	rc = deliver(globl, thread_id, T, T->pend, xe_gpu_batch_records(T->pend), T->pend_time_ns);
	if (rc != XE_OK) batch_lost(T, 0, rc);
	xe_gpu_batch_free(T->pend);
	T->pend = NULL;
	return rc;
}

/* when the waker should look at this thread: a parked batch in a millisecond
 * (every call moves the time on, so the waker asks again while the batch is
 * still running), else when the batch being collected is due */
static void
set_due(struct shim_thread *T, uint64_t now)
{
	const uint64_t due = T->pend ? now + 1000000ull : (T->n ? T->first_mono + S.cfg.max_wait_ns : 0);
	__atomic_store_n(&T->due_at, due, __ATOMIC_RELEASE);
}

/* runs what the thread has collected.  Without overlap the exit calls follow
 * at once.  With it, the batch is left running while the thread collects the
 * next one (`park`), and what was left running before is delivered first, so
 * the order of the exit calls stays the order of arrival. */
static int
run_batch(struct xe_data *globl, size_t thread_id, struct shim_thread *T, int park)
{
	xe_batch *b = NULL;
	size_t nrec;
	int rc = XE_OK;

	if (T->lock) pthread_mutex_lock(T->lock);
	if (!T->n) {
		rc = deliver_pending(globl, thread_id, T);
		if (T->lock) pthread_mutex_unlock(T->lock);
		set_due(T, mono_ns());
		return rc;
	}
	{
		const uint64_t t0 = fine_ns();
		rc = xe_gpu_batch_prepare(T->h, T->data, T->off, T->src, T->kind, T->n, &b);
		if (rc == XE_OK) rc = xe_gpu_batch_run(b, NULL);
		T->ns[0] += fine_ns() - t0;
	}
	/* (the batch before this one: its exit calls while this one's bytes travel) */
	(void)deliver_pending(globl, thread_id, T);
	if (rc == XE_OK) {
		nrec = xe_gpu_batch_records(b);
		T->stats[2] += nrec;
		if (S.cfg.overlap && park) {
			uint8_t *d = T->data;
			uint64_t *tn = T->time_ns;
			T->pend = b;
			b = NULL;
			T->data = T->pend_data; T->pend_data = d;
			T->time_ns = T->pend_time_ns; T->pend_time_ns = tn;
			T->since_query = 0;
		} else {
			rc = deliver(globl, thread_id, T, b, nrec, T->time_ns);
		}
	}
	if (rc != XE_OK) batch_lost(T, T->n, rc);
	if (b) xe_gpu_batch_free(b);
	if (T->lock) pthread_mutex_unlock(T->lock);
	T->stats[1]++;
	T->n = 0;
	T->used = 0;
	set_due(T, mono_ns());
	return rc;
}

void
xenoeye_gpu_shim_times(size_t thread_id, uint64_t out[4])
{
	memset(out, 0, 4 * sizeof(out[0]));
	if (S.ready && (thread_id < S.cfg.nthreads)) memcpy(out, S.th[thread_id].ns, sizeof(S.th[thread_id].ns));
}

int
xenoeye_gpu_shim_flush(struct xe_data *globl, size_t thread_id)
{
	if (!S.ready || (thread_id >= S.cfg.nthreads)) return XE_EINVAL;
	return run_batch(globl, thread_id, &S.th[thread_id], 0);
}

/* what an idle moment is for: a due batch runs (and is delivered at once: no
 * next batch is in sight), a parked batch that has finished is delivered */
static int
attend(struct xe_data *globl, size_t thread_id, struct shim_thread *T, uint64_t now)
{
	if (T->n && ((now - T->first_mono) >= S.cfg.max_wait_ns)) return run_batch(globl, thread_id, T, 0);
	if (T->pend && !T->n) return run_batch(globl, thread_id, T, 0);
	if (T->pend && (xe_gpu_batch_done(T->pend) != 0)) {
		int rc;
		if (T->lock) pthread_mutex_lock(T->lock);
		rc = deliver_pending(globl, thread_id, T);
		if (T->lock) pthread_mutex_unlock(T->lock);
		set_due(T, now);
		return rc;
	}
	if (T->pend) set_due(T, now);
	return XE_OK;
}

int
xenoeye_gpu_shim_poll(struct xe_data *globl, size_t thread_id)
{
	struct shim_thread *T;

	if (!S.ready || (thread_id >= S.cfg.nthreads)) return XE_EINVAL;
	T = &S.th[thread_id];
	return attend(globl, thread_id, T, mono_ns());
}

uint64_t
xenoeye_gpu_shim_wakeups(void)
{
	return S.waker_on ? __atomic_load_n(&S.wake_sent, __ATOMIC_RELAXED) : 0;
}

void
xenoeye_gpu_shim_stats(size_t thread_id, uint64_t out[5])
{
	memset(out, 0, 5 * sizeof(uint64_t));
	if (S.ready && (thread_id < S.cfg.nthreads)) memcpy(out, S.th[thread_id].stats, 5 * sizeof(uint64_t));
}

const char *
xenoeye_gpu_shim_error(size_t thread_id)
{
	return (S.ready && (thread_id < S.cfg.nthreads)) ? S.th[thread_id].err : "";
}

/* one datagram into the thread's batch; the batch runs first when it is full,
 * when the datagram is of the other kind (a batch holds one kind, so that
 * record order is arrival order) or when its oldest datagram is due */
static int
take(struct xe_data *globl, size_t thread_id, struct flow_packet_info *fpi, int len, int kind)
{
	struct shim_thread *T;
	struct timespec tmsp;
	uint64_t now;

	if (!S.ready || (thread_id >= S.cfg.nthreads) || !fpi) return 0;
	/* the reference stamps the datagram first (netflow.c:798-802,
	 * sflow-impl.h:316-320: seconds * 1e9 + nanoseconds in double) */
	if (clock_gettime(CLOCK_REALTIME_COARSE, &tmsp) < 0) return 0;
	fpi->time_ns = tmsp.tv_sec * 1e9 + tmsp.tv_nsec;
	if ((len < 0) || (len > MAX_NF_PACKET_SIZE)) return 0;
	if ((len == (int)sizeof(wake_magic)) && !memcmp(fpi->rawpacket, wake_magic, sizeof(wake_magic))) {
		/* the wake-up thread's datagram: nothing to batch, only the due check */
		T = &S.th[thread_id];
		(void)attend(globl, thread_id, T, mono_ns());
		return 0;
	}
	if (kind == XE_KIND_NETFLOW) {
		/* netflow.c:804-819: other versions are logged and left */
		const unsigned version = (len >= 2) ? (((unsigned)fpi->rawpacket[0] << 8) | fpi->rawpacket[1]) : 0;
		if ((version != 5) && (version != 9) && (version != 10)) return 0;
	}
	T = &S.th[thread_id];
	now = mono_ns();
	if (T->n && ((T->n >= S.cfg.max_datagrams) || (T->used + (size_t)len > T->cap) || (T->kind[0] != kind))) {
		/* full (or the other kind follows): more is on its way, the batch may run beside it */
		(void)run_batch(globl, thread_id, T, 1);
	} else if (T->n && ((now - T->first_mono) >= S.cfg.max_wait_ns)) {
		(void)run_batch(globl, thread_id, T, 0);
	} else if (T->pend && (++T->since_query >= 64)) {
		/* a parked batch is delivered as soon as it has finished: looked at every 64 datagrams */
		T->since_query = 0;
		if (xe_gpu_batch_done(T->pend) != 0) {
			if (T->lock) pthread_mutex_lock(T->lock);
			(void)deliver_pending(globl, thread_id, T);
			if (T->lock) pthread_mutex_unlock(T->lock);
			set_due(T, now);
		}
	}
	if (!T->n) {
		T->off[0] = 0;
		T->first_mono = now;
	}
	memcpy(T->data + T->used, fpi->rawpacket, (size_t)len);
	T->used += (size_t)len;
	T->src[T->n] = fpi->src_addr_ipv4;
	T->kind[T->n] = (uint8_t)kind;
	T->time_ns[T->n] = fpi->time_ns;
	T->n++;
	T->off[T->n] = T->used;
	T->stats[0]++;
	if (T->n == 1) set_due(T, now);
	return 1;
}

/* netflow.c:824-837 fills the reference's table of per-field parse functions;
 * the field table of this path lives in the library */
void
netflow_process_init(void)
{
}

int
netflow_process(struct xe_data *data, size_t thread_id, struct flow_packet_info *npi, int len)
{
	return take(data, thread_id, npi, len, XE_KIND_NETFLOW);
}

int
sflow_process(struct xe_data *global, size_t thread_id, struct flow_packet_info *fpi, int len)
{
	return take(global, thread_id, fpi, len, XE_KIND_SFLOW);
}
