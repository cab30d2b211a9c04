/*
 * xenoeye_gpu_shim.h -- the reference's own entry points of the per-record
 * flow path over the batch ABI of xe_gpu.h: a drop-in for netflow.c / sflow.c.
 *
 * "reference" is vmxdev/xenoeye; file:line are relative to its tree.
 *
 * Entry (what calls the path; same symbols, same signatures):
 *     void netflow_process_init(void);                          netflow.h:156
 *     int  netflow_process(struct xe_data *, size_t thread_id,
 *              struct flow_packet_info *, int len);             netflow.h:157-158
 *     int  sflow_process(struct xe_data *, size_t thread_id,
 *              struct flow_packet_info *, int len);             sflow.h:17-19
 * Exit (what the path calls, once per record and monitoring object that
 * receives it, objects in the reference's pre-order, records in arrival
 * order per thread, each with the arrival time of ITS datagram):
 *     int  monit_object_process_nf(struct xe_data *, struct monit_object *,
 *              size_t thread_id, uint64_t time_ns,
 *              struct flow_info *);                             monit-objects.h:288-289
 *
 * The reference decodes and classifies every datagram inside the call.  The
 * shim copies the datagram (the caller's struct flow_packet_info is stack
 * memory that is only valid during the call, scapture.c:25-61), stamps it with
 * the time the reference would have stamped (netflow.c:798-802,
 * sflow-impl.h:316-320) and returns; when a thread's batch is full or old
 * enough, the next call on that thread runs the batch on the GPU and makes the
 * exit calls on that same thread, so the consumers' lock-free per-thread state
 * (monit-objects.c:923-1033) keeps working unchanged.
 *
 * struct xe_data, struct monit_object and struct filter_expr stay opaque here:
 * the shim never looks inside them.
 */
#ifndef xenoeye_gpu_shim_h_included
#define xenoeye_gpu_shim_h_included

#include <stddef.h>
#include <stdint.h>
#include <sys/socket.h>
#include "xe_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

struct xe_data;         /* xenoeye.h:54-135 */
struct monit_object;    /* monit-objects.h:244-279 */
struct filter_expr;     /* filter.h:256-275 */

/* flow-info.h:10-33: 2312 bytes on x86-64; the shim fills them with
 * xe_gpu_flow_cursor_seek() (the bytes xe_gpu_batch_flow_info() makes) and hands them over untyped */
struct flow_info;

#ifndef flow_info_h_included
/* flow-info.h:35-46, member for member (a translation unit that includes the
 * reference's flow-info.h first uses that definition) */
#define MAX_NF_PACKET_SIZE (64 * 1024)
struct flow_packet_info
{
	struct sockaddr src_addr;
	uint32_t src_addr_ipv4;

	uint32_t source_id;
	uint32_t epoch;
	uint64_t time_ns; /* nanoseconds */
	uint8_t rawpacket[MAX_NF_PACKET_SIZE];

	int sampling_rate;
};
#endif

/* monit_object_process_nf(), monit-objects.h:288-289 */
typedef int (*xe_shim_process_nf_fn)(struct xe_data *globl, struct monit_object *mo, size_t thread_id,
	uint64_t time_ns, struct flow_info *flow);
/* filter_match(), filter.h:331 */
typedef int (*xe_shim_filter_match_fn)(struct filter_expr *expr, struct flow_info *flow);

struct xe_shim_config
{
	/* capture threads (xenoeye.c:412: nthreads = nnfcap + nsfcap); thread_id
	 * of every entry call is below it */
	size_t nthreads;
	/* the committed handle every thread prepares its batches on: one per
	 * thread (templates are then per capture thread, which is where the
	 * reference's operator pins an exporter, STEP-BY-STEP.md:138-156), or the
	 * same handle several times (the shim then serialises the threads' use of
	 * it).  All handles carry the same objects in the same order. */
	xe_gpu **gpu;
	/* object of every mask bit: the reference's objects in the pre-order of
	 * its tree walk (netflow.c:213-244), i.e. bit m = the m-th object
	 * xe_gpu_load_mos() / xe_gpu_add_mo() added */
	struct monit_object **mo;
	size_t nmo;
	/* filter expression (mo->expr) of every object; only read for objects
	 * with xe_gpu_mo_host_only(): may be NULL when there are none */
	struct filter_expr **expr;
	/* NULL: the symbols monit_object_process_nf / filter_match of the program
	 * the shim is linked into */
	xe_shim_process_nf_fn process_nf;
	xe_shim_filter_match_fn filter_match;
	/* a thread's batch runs when it holds this many datagrams (0: 16384) or
	 * bytes (0: 32 MiB), or when its oldest datagram has waited this long
	 * (0: 20 ms; checked when the thread's next datagram arrives and by
	 * xenoeye_gpu_shim_poll()) */
	size_t max_datagrams, max_bytes;
	uint64_t max_wait_ns;
	/* idle flush without a change to the capture loop.  scapture_thread() blocks
	 * in recvfrom() and has no idle branch (scapture.c:25-37), so a thread whose
	 * exporters fall silent would keep its last partial batch.  NULL: the caller
	 * polls (below).  Else per thread the UDP port of its capture socket
	 * (cap->port, scapture.c:143-144; 0 for threads without a socket, e.g. pcap
	 * captures): when a thread's batch is due, a helper thread of the shim sends
	 * one 16-byte datagram to 127.0.0.1:port; recvfrom() returns, the loop makes
	 * its usual entry call, and the shim runs the due batch ON THE CAPTURE THREAD
	 * and drops the datagram.  The helper thread touches no batch and makes no
	 * exit call. */
	const uint16_t *wake_port;
	/* not 0: a full batch is left running on the GPU while the thread collects
	 * the next one (xe_gpu_set_overlap(): two page-locked datagram buffers per
	 * thread, the bytes of batch k + 1 travel while batch k is decoded); its
	 * exit calls are made - on the capture thread, in arrival order - at the
	 * first entry call that finds it finished, before those of any later batch,
	 * or by the idle flush */
	int overlap;
};

/* after the reference has loaded its objects (monit_objects_init(),
 * xenoeye.c) and the handles are committed; before the capture threads start.
 * Returns XE_OK or a negative XE_E* code. */
int xenoeye_gpu_shim_init(const struct xe_shim_config *cfg);
/* runs what thread `thread_id` has collected, on the calling thread (which
 * must be the one that makes the thread's entry calls): for the idle branch of
 * a capture loop that has one (a socket with SO_RCVTIMEO: call poll where
 * recvfrom() returns -1 / EAGAIN, scapture.c:33-37) and before shutdown.  poll
 * only runs a batch that is due.  With wake_port set no such branch is needed. */
int xenoeye_gpu_shim_flush(struct xe_data *globl, size_t thread_id);
int xenoeye_gpu_shim_poll(struct xe_data *globl, size_t thread_id);
/* wake-up datagrams sent so far (0 without wake_port) */
uint64_t xenoeye_gpu_shim_wakeups(void);
/* after the capture threads have stopped and flushed */
void xenoeye_gpu_shim_shutdown(void);
/* datagrams taken, batches run, records decoded, exit calls made, batches that
 * failed (their datagrams are lost, the reason is in xenoeye_gpu_shim_error) */
void xenoeye_gpu_shim_stats(size_t thread_id, uint64_t out[5]);
const char *xenoeye_gpu_shim_error(size_t thread_id);
/* nanoseconds the thread has spent so far: preparing and starting batches
 * (header walk, copy to the device), waiting for a batch and copying its
 * results back, reading masks / status / datagram numbers out of that copy,
 * and in the loop over the records (struct flow_info + the exit calls) */
void xenoeye_gpu_shim_times(size_t thread_id, uint64_t out[4]);

/* the reference's entry points */
void netflow_process_init(void);
int netflow_process(struct xe_data *data, size_t thread_id, struct flow_packet_info *npi, int len);
int sflow_process(struct xe_data *global, size_t thread_id, struct flow_packet_info *fpi, int len);

#ifdef __cplusplus
}
#endif
#endif
