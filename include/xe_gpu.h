/*
 * xe_gpu.h -- C ABI of the MI355X flow path: export datagrams in, decoded
 * structure-of-arrays flow fields + per-record monitoring-object match bitmask
 * + per-object counters out.
 *
 * Plain C, plain pointers and sizes.  Every function returns 0 on success or
 * a negative XE_E* code; xe_gpu_strerror() names it and xe_gpu_last_error()
 * returns the detail text of the last failure on that handle.  A handle is
 * single-producer: calls on one handle must not overlap.
 *
 * "reference" below is vmxdev/xenoeye; file:line are relative to its tree.
 */
#ifndef xe_gpu_h_included
#define xe_gpu_h_included

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xe_gpu xe_gpu;        /* configuration + device state */
typedef struct xe_batch xe_batch;    /* one prepared batch, resident in HBM */
typedef struct xe_upload xe_upload;  /* datagram bytes sent ahead of their batch */

enum
{
	XE_OK = 0,
	XE_EINVAL = -1,       /* bad argument */
	XE_ENOMEM = -2,       /* host allocation failed */
	XE_EHIP = -3,         /* a HIP call failed (text in xe_gpu_last_error) */
	XE_EPARSE = -4,       /* filter / list / config text did not parse */
	XE_EUNSUPPORTED = -5, /* valid in the reference, not evaluable on the device */
	XE_ESTATE = -6,       /* call out of order (e.g. batch before commit) */
	XE_EIO = -7           /* file or directory could not be read */
};

#define XE_FLOW_INFO_SIZE 2312   /* sizeof(struct flow_info), flow-info.h:10-33 */
#define XE_KIND_NETFLOW 0        /* Netflow v5 / v9 / IPFIX datagram */
#define XE_KIND_SFLOW 1          /* sFlow v5 datagram */

const char *xe_gpu_strerror(int code);
const char *xe_gpu_last_error(const xe_gpu *h);

/* ---- life cycle ---- */
/* device: HIP device ordinal.  Fails with XE_EHIP when no GPU is usable:
 * there is no CPU fallback. */
int xe_gpu_create(xe_gpu **out, int device);
/* every batch prepared on the handle must have been freed before */
void xe_gpu_destroy(xe_gpu *h);

/* ---- configuration (load-time, host) ---- */
/* replaces iplists_load() (iplist.c:299-322): every non-dot file of `dir`
 * becomes a list named after the file.  Must precede the filters that name
 * the lists, as in the reference (xenoeye.c:489 before monit_objects_init). */
int xe_gpu_load_iplists(xe_gpu *h, const char *dir);
/* the same loader (iplist.c:218-297) on a memory image of one list file */
int xe_gpu_add_iplist(xe_gpu *h, const char *name, const char *text, size_t len);

/* replaces the filter half of monit_object_info_parse() (monit-objects.c:47-105,
 * parse_filter() filter-parser.c:225): adds one monitoring object.  parent = -1
 * for a top-level object, else the index of an object added earlier; a child is
 * evaluated only for records its parent matched (netflow.c:213-244).  Returns
 * the object's index (= its bit in the match bitmask) or a negative code. */
int xe_gpu_add_mo(xe_gpu *h, const char *name, int parent, const char *filter);
/* Objects the device cannot decide.  A filter that compares strings, uses
 * mfreq() (state between records, filter.c:712-751), the geo / AS functions
 * (the caller's databases, filter.c:753-834), or portstr() / ppstr() on a
 * handle without a port-name table is still added: the object keeps its index
 * and its place in the tree, xe_gpu_mo_host_only() answers 1 for it, and on
 * the device it takes every record its parent took.  Its bit in the match
 * mask therefore means "candidate": the caller runs its own filter_match()
 * (filter.h:331) for that object on the re-created flow_info and, when it
 * says no, drops the bit together with the bits of the objects below it
 * (shim/xenoeye_gpu_shim.c does exactly that).  Nothing inside this library
 * evaluates such a filter on the CPU.  For a host-only object and every
 * object below one (xe_gpu_mo_host_gated() answers 1) the counters of
 * xe_gpu_batch_counters() count candidates, and fixed-window tables are
 * refused (XE_EUNSUPPORTED).  Filters of devices.conf marks get no such
 * treatment: a mark the device cannot evaluate fails xe_gpu_add_device(). */
int xe_gpu_mo_host_only(const xe_gpu *h, int mo);
const char *xe_gpu_mo_host_only_reason(const xe_gpu *h, int mo);   /* "" when it is not */
int xe_gpu_mo_host_gated(const xe_gpu *h, int mo);
/* Service names of the 65536 ports, what port_to_str() / ports_pair_to_str()
 * (utils.c:28-77) put in front of "(port)": names[port] = the name, NULL or ""
 * for a port without one (the reference's table is a static array of its
 * utils.c; a binding derives the names by calling port_to_str() for every
 * port and cutting the text at the "(").  With the table set, portstr() and ppstr() filters are
 * lowered at commit to sets / conditions over the port values and evaluated
 * on the device; must be called before the objects that use them are added.
 * NULL removes the table. */
int xe_gpu_set_port_names(xe_gpu *h, const char *const *names);
/* replaces monit_objects_init() (monit-objects.c:470, directory walk :252-458):
 * every directory below mo_dir holding an mo.conf is an object, nested
 * directories are children.  Only the "filter" key is read. */
int xe_gpu_load_mos(xe_gpu *h, const char *mo_dir);
int xe_gpu_mo_count(const xe_gpu *h);
const char *xe_gpu_mo_name(const xe_gpu *h, int mo);
int xe_gpu_mo_parent(const xe_gpu *h, int mo);
/* the compiled postfix program in the text form of filter_dump()
 * (filter.c:1226-1256); returns the length written or a negative code */
int xe_gpu_mo_dump(const xe_gpu *h, int mo, char *out, size_t outlen);

/* parse_filter() + filter_dump() without touching a handle's objects; the
 * handle supplies the IP-list names (may be NULL: no lists) */
int xe_filter_check(const xe_gpu *h, const char *filter, char *dump, size_t dumplen,
	char *err, size_t errlen);

/* replaces one entry of devices.conf (devices.c:60-139).  use_ip / use_id say
 * which of ip (network byte order) and id (host order, as written in the file)
 * are set; marks are filter texts. */
int xe_gpu_add_device(xe_gpu *h, int use_ip, uint32_t ip_net_order, int use_id, uint32_t id,
	int sampling_rate, int skip_unmarked, const char *const *marks, int nmarks);
/* replaces devices_load() (devices.c:142-196) */
int xe_gpu_load_devices(xe_gpu *h, const char *path);

/* flattens programs and tries and uploads them; required before the first
 * batch and after any configuration change */
int xe_gpu_commit(xe_gpu *h);

/* ---- fixed-window tables ----
 * The per-record part of monit_object_process_nf() (monit-objects.h:288,
 * monit-objects.c:923-1023) for one "fwm" entry of an object's mo.conf: per
 * distinct key (the bytes of the key fields as struct flow_info holds them,
 * one after the other) the wrapping u64 sums of value * scale * sampling rate
 * of the aggregable fields.  `fields` are the strings of the entry's "fields"
 * array (`[src|dst] name [asc|desc]`, `octets`, `bits`, `packets` ...,
 * filter-parser.c:396-484) or the function forms div(), div_r(), div_l(),
 * min(), tfstr(), portstr(), ppstr() (monit_object_key_add_fld(),
 * monit-objects.c:878-921; the last two need xe_gpu_set_port_names() first);
 * mfreq(), the geo / AS functions and string fields answer XE_EUNSUPPORTED.  The
 * table lives on the device as long as the handle; every xe_gpu_batch_run()
 * adds the batch's records to it.  Returns the table's index.  max_keys = the
 * distinct keys to make room for (0: 65536); records whose key finds no room
 * are counted in `dropped` (the reference drops them on TKVDB_ENOMEM). */
int xe_gpu_add_fwm(xe_gpu *h, int mo, const char *const *fields, int nfields, size_t max_keys);
/* xe_gpu_load_mos() that also registers the "fwm" entries of every mo.conf
 * (fwm_config(), monit-objects-fwm.c:139-205: "name" and "fields") as tables;
 * entries with fields this library does not aggregate (mfreq, geo / AS, strings)
 * are left to the reference's own consumer and counted in *skipped.  Returns
 * the number of tables added. */
int xe_gpu_load_mos_with_tables(xe_gpu *h, const char *mo_dir, size_t max_keys, int *skipped);
int xe_gpu_fwm_count(const xe_gpu *h);
int xe_gpu_fwm_object(const xe_gpu *h, int fwm);          /* index of the object it belongs to */
const char *xe_gpu_fwm_name(const xe_gpu *h, int fwm);    /* "name" of the mo.conf entry, "" if added by hand */
int xe_gpu_fwm_info(const xe_gpu *h, int fwm, size_t *key_bytes, int *nvals, size_t *slots);
/* rows in no particular order (the reference sorts when it exports a window,
 * monit-objects-fwm.c:497-691): keys[row][key_bytes], vals[row][nvals].
 * Returns the number of rows the table holds (copies at most max_rows). */
long xe_gpu_fwm_read(xe_gpu *h, int fwm, uint8_t *keys, uint64_t *vals, size_t max_rows, uint64_t *dropped);
/* "limit" (rows per export, 0 = all) and "time" (seconds per window) of the
 * mo.conf entry the table came from (fwm_config(), monit-objects-fwm.c:179-190);
 * both 0 for tables added by hand */
int xe_gpu_fwm_window(const xe_gpu *h, int fwm, int *limit, int *seconds);
/* the rows as fwm_sort_and_dump() + fwm_dump() write them
 * (monit-objects-fwm.c:496-589, 405-449): sorted by the fields in the order
 * of the entry (sums as big-endian numbers, `desc` fields inverted, bytewise
 * ascending), cut after `limit` rows (0 = all, < 0 = the entry's own limit);
 * others[nvals] receives the sums over the rows behind the cut ("others" row
 * of the export), *hit_limit whether the reference would write that row.
 * Returns the number of rows before the cut; at most max_rows are stored. */
long xe_gpu_fwm_export(xe_gpu *h, int fwm, int limit, uint8_t *keys, uint64_t *vals, size_t max_rows,
	uint64_t *others, int *hit_limit);
/* end of a window: empties the table */
int xe_gpu_fwm_clear(xe_gpu *h, int fwm);

/* Databases of the filter functions asn(), asd() and continent() ... long()
 * (geoip.c:33-88, filter.c:753-834): `image` = the bytes of the reference's own
 * geo4.db / geo6.db (kind 0) or as4.db / as6.db (kind 1) file, family 4 or 6
 * (an array of struct btrie_node_geo / btrie_node_as, geoip.h:43-55; the
 * library keeps a copy; 0 bytes drops the database).  Set them before the
 * objects and devices: a filter with such a function is then evaluated on the
 * device - at xe_gpu_commit the database becomes an IP list per function term,
 * with a leaf wherever the database has one whose payload meets the term's
 * operands, so the device's list walk answers what the function's own walk
 * answers (an absent field is looked up as zeros, as the reference reads it).
 * Without its database an object with such a function is host-only
 * (xe_gpu_mo_host_only).  XE_EPARSE: the image is not a tree of whole nodes.
 * The reference re-reads the files on a signal (geoip_reload(),
 * geoip.c:131-178): once objects exist a loaded database may be REPLACED by a
 * new image (not added or dropped: which objects the device decides was
 * settled when they were added; XE_ESTATE otherwise), and the next
 * xe_gpu_commit() rebuilds the lists made from it.  The lookups are checked
 * against the reference's own walk (ip-btrie.h:88-115) over synthetic
 * databases: the reference tree holds no database file. */
int xe_gpu_set_geodb(xe_gpu *h, int kind, int family, const void *image, size_t bytes);
/* the four files of `dir` that exist (geoip.c:146-149); returns how many were loaded */
int xe_gpu_load_geodb(xe_gpu *h, const char *dir);
/* geoip_reload() (geoip.c:131-178, on SIGHUP xenoeye.c:576): re-reads the files
 * of `dir` into a configured handle and commits.  Call it between batches, on
 * the thread that owns the handle; batches prepared before it must have been
 * freed.  Returns how many files were loaded, or a negative code (the handle
 * then still needs a successful xe_gpu_commit()). */
int xe_gpu_reload_geodb(xe_gpu *h, const char *dir);

/* IP lists on the device (iplist_match4/6, iplist.c:160-216, flattened): the
 * table of a group of 8 lists starts with a root indexed by the first 16
 * address bits (256 KiB, stays in the L2: the default) or the first 24 (64 MiB
 * per group and address family: every IPv4 prefix up to /24 is answered by
 * one load, which then comes from memory instead of the L2 - worth trying
 * only for lists of a million prefixes and more).  bits: 16 or 24 (0 = 16).
 * Takes effect at the next xe_gpu_commit(); results do not depend on it. */
int xe_gpu_set_list_root_bits(xe_gpu *h, int bits);

/* Host threads xe_gpu_batch_prepare() walks datagram headers with (default 8,
 * or the environment's XE_PREPASS_THREADS; 1 = the calling thread only).
 * Datagrams of one exporter stay on one thread, in arrival order, the way the
 * reference pins exporters to capture threads (xenoeye.c:584-625); the result
 * does not depend on the number of threads. */
int xe_gpu_set_prepass_threads(xe_gpu *h, int nthreads);

/* Where datagram, set and sample headers are walked (netflow.c:319-365,
 * :628-678, :727-786; sflow-impl.h:68-405).  mode bit 0: sFlow datagrams on
 * the device; bit 1: Netflow datagrams of the usual shape (one data set that
 * ends with the datagram, template known before the batch, complete
 * fixed-length records, exporter without mark rules) on the device, the others
 * - every datagram that carries a template among them - on the host, which
 * keeps the template store.  Default 3 (the environment's XE_DEVICE_WALK
 * overrides it); 0 = the host walks everything.  Results do not depend on it. */
int xe_gpu_set_device_walk(xe_gpu *h, int mode);
/* what the device walked of a prepared batch: Netflow datagrams and their
 * records, sFlow records (any pointer may be NULL) */
int xe_gpu_batch_device_walked(const xe_batch *b, size_t *netflow_datagrams, size_t *netflow_records, size_t *sflow_records);

/* Page-locked host memory for the datagram buffer handed to
 * xe_gpu_batch_prepare() and for result buffers: copies to and from such
 * memory run at the full speed of the link (the capture threads of the
 * reference fill a 64 KiB stack buffer per datagram, scapture.c:25-61; a
 * batching caller appends the datagrams to a buffer from here instead).
 * Ordinary memory works too, only slower.  NULL on failure. */
void *xe_gpu_host_alloc(xe_gpu *h, size_t bytes);
void xe_gpu_host_free(xe_gpu *h, void *p);

/* ---- data path ---- */
/* replaces the per-datagram entry points netflow_process() (netflow.h:157,
 * netflow.c:788) and sflow_process() (sflow.h:17, sflow-impl.h:293) for a
 * whole batch of datagrams in arrival order:
 *   data      all datagrams back to back
 *   offsets   n+1 byte offsets into data (datagram i = [offsets[i], offsets[i+1]))
 *   src_ip4   per datagram: exporter IPv4 as flow_packet_info.src_addr_ipv4
 *             (flow-info.h:38), i.e. the 4 address bytes in memory order
 *   kinds     per datagram XE_KIND_*, or NULL for all-Netflow
 * The host pre-pass walks set headers in order (templates arrive in-band),
 * then everything the kernels need is copied to HBM.  The batch keeps no
 * pointer into the caller's arrays. */
int xe_gpu_batch_prepare(xe_gpu *h, const uint8_t *data, const uint64_t *offsets,
	const uint32_t *src_ip4, const uint8_t *kinds, size_t n, xe_batch **out);
int xe_gpu_batch_prepare_uploaded(xe_gpu *h, xe_upload *up, const uint8_t *data, const uint64_t *offsets,
	const uint32_t *src_ip4, const uint8_t *kinds, size_t n, xe_batch **out);
void xe_gpu_batch_free(xe_batch *b);

/* Overlap of consecutive batches on ONE handle (off by default).  With it on,
 * xe_gpu_batch_prepare() does not wait for the datagram bytes to reach the
 * device: they travel on a stream of their own, xe_gpu_batch_run() orders the
 * kernels behind them, and the caller fetches the results of batch k while the
 * bytes of batch k + 1 are on the link (both directions of the link carry data
 * at once).  The caller keeps `data` unchanged until xe_gpu_batch_input_done()
 * has returned (or the batch has run and been waited for), i.e. it fills a
 * second buffer meanwhile.  The header walk then reads Netflow headers in
 * place from a page-locked `data` (xe_gpu_host_alloc); from ordinary memory,
 * and for sFlow, prepare still waits for the copy.  No counterpart in the
 * reference, which handles one datagram at a time (scapture.c:25-61). */
int xe_gpu_set_overlap(xe_gpu *h, int on);
/* The same one step further: the bytes of a batch are sent to the device before
 * the batch is prepared.  xe_gpu_upload_begin() only starts the copy (behind
 * the uploads started before it) and returns; xe_gpu_batch_prepare_uploaded()
 * is xe_gpu_batch_prepare() for those bytes (`data` and the per-datagram arrays
 * as there; the upload belongs to the batch afterwards, also when the call
 * fails).  The header walk then reads the device copy, so nothing but uploads
 * and results crosses the link: a caller that starts the upload of batch k + 1
 * before it prepares batch k keeps the link busy with one thread (three input
 * buffers in rotation; `data` of an upload stays unchanged until
 * xe_gpu_batch_input_done() of its batch).  xe_gpu_upload_free(): an upload
 * that will not be prepared.  No counterpart in the reference. */
int xe_gpu_upload_begin(xe_gpu *h, const uint8_t *data, size_t bytes, xe_upload **out);
void xe_gpu_upload_free(xe_upload *u);
int xe_gpu_batch_input_done(xe_batch *b);
/* 1 when the batch's last run has finished, 0 while it has not (never blocks) */
int xe_gpu_batch_done(xe_batch *b);

/* decode + classify every record of the batch on the device.  stream is a
 * hipStream_t (NULL = the null stream); the call only enqueues work.  May be
 * repeated; each run overwrites the batch's outputs and re-zeroes its counters. */
int xe_gpu_batch_run(xe_batch *b, void *stream);
/* blocks until the batch's last run has finished */
int xe_gpu_batch_sync(xe_batch *b);

/* instrumentation (no counterpart in the reference): with timing enabled every
 * run records HIP events on its stream around each kernel launch;
 * xe_gpu_batch_kernel_ms() then gives the last run's duration of the
 * template-grouped Netflow kernel, the general Netflow kernel and the sFlow
 * kernel in milliseconds (-1 for a kernel the batch does not need) */
int xe_gpu_batch_timing(xe_batch *b, int enable);
int xe_gpu_batch_kernel_ms(xe_batch *b, float ms[3]);
int xe_gpu_batch_fwm_ms(xe_batch *b, float *ms);   /* the fixed-window kernel, -1 without tables */
/* one more run of the batch that counts the loads of the IP-list walks
 * (iplist_match4/6, iplist.c:160-216, on the flattened tables): out[0] = loads
 * from the root tables (cache-resident), out[1] = loads of nodes and chain
 * records behind them (what a walk waits for); blocks until the run is done */
int xe_gpu_batch_count_gathers(xe_batch *b, void *stream, uint64_t out[2]);
/* the header walk on the device: with walk timing enabled on the handle every
 * xe_gpu_batch_prepare() records HIP events around the walk's launches (the
 * device side of netflow.c:319-365, :628-678, sflow-impl.h:293-405: shape,
 * counts, prefix sums, run and tile tables), xe_gpu_batch_walk_ms() gives
 * their summed duration in milliseconds, host waits in between left out
 * (-1: the batch was walked by the host or timing was off) */
int xe_gpu_set_walk_timing(xe_gpu *h, int enable);
int xe_gpu_batch_walk_ms(xe_batch *b, float *ms);

/* data sets left out because their template has more elements than this
 * library's device form of a template holds (254 within the first 64 KiB of a
 * record; the reference has no such limit): 0 in any sane stream */
size_t xe_gpu_batch_unsupported_sets(const xe_batch *b);

/* datagrams left out as a whole because they stand for more than 2^20 records:
 * a Netflow v9 data set is iterated header.count times whatever it holds
 * (netflow.c:272), so a datagram of empty sets for a known template id can
 * claim tens of millions of empty records.  The reference would walk them
 * all; no well-formed datagram comes near the bound.  0 in any sane stream. */
size_t xe_gpu_batch_oversized_datagrams(const xe_batch *b);

/* sizes */
size_t xe_gpu_batch_records(const xe_batch *b);      /* records reaching classification */
/* of those, the Netflow records left to the per-lane kernel (cut-short or
 * variable-length records, templates with repeated or string elements ...):
 * the others go through the kernel that decodes a tile of one template at a time */
size_t xe_gpu_batch_per_lane_records(const xe_batch *b);
size_t xe_gpu_batch_mask_words(const xe_batch *b);   /* 64-bit words per record */
size_t xe_gpu_batch_payload_bytes(const xe_batch *b);/* datagram bytes in the batch */
/* bytes the kernels must move at minimum for this batch: datagram bytes +
 * stored widths of every decoded field + 1 presence byte + mask words */
size_t xe_gpu_batch_algorithmic_bytes(const xe_batch *b);

/* results (device -> caller memory; each implies xe_gpu_batch_sync) */
/* masks[record][words]: bit m of a record = object m received it, i.e. what
 * monit_object_process_nf() (monit-objects.h:288) would have been called with */
int xe_gpu_batch_masks(xe_batch *b, uint64_t *masks, size_t nwords_total);
/* counters[object][3] = {records, sum in_bytes*sampling_rate, sum
 * in_pkts*sampling_rate}, wrapping u64 (monit-objects.c:995-1007, scale 1) */
int xe_gpu_batch_counters(xe_batch *b, uint64_t *counters, size_t n_u64);
/* status[record]: 0 = classified; 1 = device_rules_check() (devices.c:286-314)
 * returned 0; 2 = sFlow sample whose sampled header did not reach the IP layer
 * and is therefore not classified at all (sflow.c:164-168) */
int xe_gpu_batch_dropped(xe_batch *b, uint8_t *dropped, size_t n);
/* dgram[record] = index (in the arrays given to xe_gpu_batch_prepare) of the
 * datagram the record was decoded from: what a consumer needs to hand every
 * record the arrival time of ITS datagram, as the reference does
 * (fpi->time_ns, netflow.c:802 and :227-228, sflow.c:114-115).  Needs no
 * device work; n >= xe_gpu_batch_records(). */
int xe_gpu_batch_record_datagram(xe_batch *b, uint32_t *dgram, size_t n);
/* re-creates the reference's struct flow_info (flow-info.h:10-33) of
 * records [first, first+count) from the structure-of-arrays output:
 * out[count][XE_FLOW_INFO_SIZE]; payload_ptr is NULL */
int xe_gpu_batch_flow_info(xe_batch *b, size_t first, size_t count, void *out);
/* The same, record after record into ONE buffer of XE_FLOW_INFO_SIZE bytes: for
 * a consumer that looks at a record and is done with it, as the reference's
 * exit call is (monit_object_process_nf(), monit-objects.c:923-1033, gets a
 * pointer to the one struct flow_info netflow.c:273-275 keeps on its stack).
 * xe_gpu_batch_flow_cursor() waits for the batch, brings ALL its outputs to the
 * host in one copy (page-locked memory the handle keeps and reuses) and clears
 * `flow`; xe_gpu_flow_cursor_seek() makes `flow` record r - r must not be below
 * the record of the call before - by taking back the fields the last record
 * set and setting r's, so records nobody wants cost nothing and no record costs
 * a 2312-byte clear.  The caller must not write to `flow` in between.  While a
 * cursor is open the batch's other read-outs (masks, dropped, columns ...) are
 * served from the same host copy.  One open cursor per handle:
 * XE_ESTATE otherwise. */
typedef struct xe_flowcur xe_flowcur;
int xe_gpu_batch_flow_cursor(xe_batch *b, void *flow, xe_flowcur **out);
int xe_gpu_flow_cursor_seek(xe_flowcur *cur, size_t r);
void xe_gpu_flow_cursor_free(xe_flowcur *cur);

/* one decoded column as the device wrote it: width bytes per record (the
 * field's maximum size; 6-byte MACs are stored in 8).  field = index in the
 * field table (xe_field_*).  Returns XE_EINVAL when no template of the batch
 * carries the field. */
int xe_gpu_batch_column(xe_batch *b, int field, void *out, size_t outlen, size_t *width);
/* device pointers of the resident outputs, for consumers that stay on the GPU */
int xe_gpu_batch_device_ptrs(xe_batch *b, const void **masks, const void **counters,
	const void **nparsed);

/* ---- capture-file replay (host only, needs no GPU) ----
 * replaces pcap_packet() (pcapture.c:111-180) for frames read from a classic
 * pcap file (micro- or nanosecond stamps, either byte order, Ethernet link
 * type): 14-byte Ethernet header, IPv4 header of IHL*4 bytes and UDP header
 * are stripped by the reference's rules, the IPv4 source address becomes the
 * exporter address.  The reference selects frames with a BPF expression
 * (pcapture.c:291-305); here udp_port > 0 keeps datagrams to that UDP
 * destination port, 0 keeps every UDP datagram.
 * On failure xe_pcap_open() still returns a reader in *out (when memory
 * allowed) whose xe_pcap_error() gives the reason; close it. */
typedef struct xe_pcap xe_pcap;
int xe_pcap_open(const char *path, xe_pcap **out);
void xe_pcap_close(xe_pcap *p);
const char *xe_pcap_error(const xe_pcap *p);
/* appends datagrams in file order, in the form xe_gpu_batch_prepare() takes,
 * until max_datagrams are in, the next one would not fit into cap bytes (it
 * is kept for the next call) or the file ends.  offsets has room for
 * max_datagrams + 1 entries.  Returns the number of datagrams (0 = end of
 * file) or a negative XE_E* code. */
long xe_pcap_read(xe_pcap *p, int udp_port, uint8_t *data, size_t cap, uint64_t *offsets,
	uint32_t *src_ip4, size_t max_datagrams);
/* frames read; datagrams handed out; frames ignored by the reference's rules
 * (short, not UDP, other port); frames dropped because their lengths
 * contradict the captured bytes (undefined behaviour in the reference) */
void xe_pcap_stats(const xe_pcap *p, uint64_t *frames, uint64_t *taken, uint64_t *ignored,
	uint64_t *damaged);

/* ---- field table (netflow.def:2-64 + the virtual fields of flow-info.h:19-30) ---- */
int xe_field_count(void);
const char *xe_field_name(int field);
int xe_field_id(int field);          /* element id, -1 for dev_ip6 / dev_id / dev_mark */
int xe_field_size_min(int field);
int xe_field_size_max(int field);
size_t xe_field_offset(int field);   /* offsetof(struct flow_info, <field>) */

/* ---- flattened IP-list probe (tests): deepest-node match of iplist_match4/6
 * (iplist.c:160-216) evaluated on the DEVICE for n addresses */
int xe_gpu_iplist_match(xe_gpu *h, const char *list, int family, const uint8_t *addrs,
	size_t n, uint8_t *out);

/* xe_gpu_iplist_match() for callers that cannot allocate device memory (tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Addresses and answers are staged in a static device array of
 * the library instead of freshly allocated buffers (every GPU the library is
 * used on holds it until the process ends: 17 * XE_GPU_NOALLOC_MAX_ADDRS
 * bytes), so the call makes no hipMalloc / hipFree.  It copies from and to
 * the caller's host memory on the null stream and waits for the result: it is
 * not meant for (and cannot run under) stream capture.  n above
 * XE_GPU_NOALLOC_MAX_ADDRS is refused with XE_EINVAL; on a device index of 64
 * or more the call answers XE_ESTATE.  All calls on one GPU share that array:
 * calls that overlap on one GPU (from any handle or thread) race; serialise
 * them. */
#define XE_GPU_NOALLOC_MAX_ADDRS 65536
int xe_gpu_iplist_match_noalloc(xe_gpu *h, const char *list, int family, const uint8_t *addrs,
	size_t n, uint8_t *out);

/* library build info: "gfx950" when the code object was built for MI355X */
const char *xe_gpu_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif
