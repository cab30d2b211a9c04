/*
 * xe_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of xenoeye's per-record flow path
 * (export datagram -> struct flow_info -> per-monitoring-object filter match).
 * It exists only so that tests, the smoke check and the benchmark's CPU
 * baseline have something to compare the HIP path against.  Nothing in
 * xenoeye_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference checkout).
 *
 * Parity status:
 *   - IP-list load/match, MAC/tcp-flags helpers and the sFlow/raw-header walk
 *     are pinned against the reference's own translation units compiled
 *     unmodified into oracle/_ref (see oracle/Makefile).
 *   - netflow.c and filter*.c cannot be compiled here without stand-ins for
 *     pcap.h / tkvdb.h / aajson.h (absent from the image and the checkout), so
 *     the Netflow v5/v9/IPFIX decode and the filter compiler/VM restatements
 *     are pinned only by the known-answer cases recorded from the reference
 *     (tests/golden/known_answers.json); beyond those: parity unpinned.
 */
#ifndef xe_oracle_h_included
#define xe_oracle_h_included

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XO_FLOW_INFO_SIZE 2312
#define XO_MAX_MO 4096

/* wipe all state: ip lists, devices, templates, monitoring objects, log */
void xo_reset(void);

/* ---- field table ---- */
int xo_nfields(void);                 /* 66 = 63 table fields + dev_ip6, dev_id, dev_mark */
const char *xo_field_name(int idx);
int xo_field_id(int idx);             /* netflow element id, -1 for the 3 virtual ones */
int xo_field_type(int idx);           /* 0 addr, 1 int, 2 mac, 3 string */
int xo_field_min(int idx);
int xo_field_max(int idx);
size_t xo_field_off(int idx);         /* offsetof(value bytes) in struct flow_info */
size_t xo_field_off_size(int idx);
size_t xo_field_off_has(int idx);
size_t xo_off_sampling_rate(void);
size_t xo_off_payload_ptr(void);

/* ---- ip lists (iplist.c) ---- */
int xo_iplist_load_dir(const char *dir);
int xo_iplist_add_text(const char *name, const char *text, size_t len);
int xo_iplist_match4(const char *name, uint32_t addr_net_order);
int xo_iplist_match6(const char *name, const uint8_t addr[16]);
/* raw bitwise trie of a list, for node-count statistics */
size_t xo_iplist_nodes(const char *name, int family);

/* ---- filters (filter-lexer.c, filter-parser.c, filter.c) ---- */
/* returns handle >= 0, or -1 and fills err */
int xo_filter_parse(const char *text, char *err, size_t errlen);
int xo_filter_dump(int handle, char *out, size_t outlen);
int xo_filter_match(int handle, const void *flow_info);
void xo_filter_free_all(void);

/* ---- devices (devices.c) ---- */
int xo_device_add(int use_ip, uint32_t ip_net_order, int use_id, uint32_t id_host_order,
	int sampling_rate, int skip_unmarked, const char *const *marks, int nmarks,
	char *err, size_t errlen);

/* ---- monitoring objects ---- */
/* parent = -1 for a top-level object; children must be added after their
 * parent; returned index is the bit position in the match bitmask */
int xo_mo_add(int parent, const char *filter, char *err, size_t errlen);
int xo_nmo(void);
/* filter_match() of object `mo`'s expression alone (no parent gating) on a
 * record given as the 2312 bytes of struct flow_info */
int xo_mo_match(int mo, const void *flow_info);

/* ---- datagram entry points (netflow.c:788, sflow-impl.h:293) ---- */
int xo_netflow_process(const uint8_t *pkt, int len, uint32_t src_ip4_net_order);
int xo_sflow_process(const uint8_t *pkt, int len, uint32_t src_ip4_net_order);

/* ---- record log ---- */
/* keep_flows: store the full struct flow_info of each record (2312 B each) */
void xo_log_config(int keep_flows, size_t max_records);
size_t xo_log_nrecords(void);
/* per record: ceil(nmo/64) words of match bits (bit m set = object m received
 * the record), dropped flag (device rules, devices.c:304) */
const uint64_t *xo_log_mask(size_t rec);
int xo_log_dropped(size_t rec);
const void *xo_log_flow(size_t rec);          /* NULL unless keep_flows */
size_t xo_log_mask_words(void);
/* counters[mo][3] = {records, sum in_bytes*rate, sum in_pkts*rate} (wrapping u64) */
const uint64_t *xo_counters(void);
void xo_log_clear(void);

/* helpers restated from utils.c */
int xo_mac_addr_read(const char *s, uint8_t mac[6]);
/* fixed-window tables of an object (monit-objects.c:923-1023): fields as the
 * "fields" array of an mo.conf "fwm" entry gives them */
int xo_field_parse(const char *text, int out[9], char *err, size_t errlen);
int xo_fwm_add(int mo, const char *const *field_texts, int n, char *err, size_t errlen);
size_t xo_fwm_rows(int fwm);
size_t xo_fwm_keysize(int fwm);
size_t xo_fwm_nvals(int fwm);
void xo_fwm_dump(int fwm, uint8_t *keys, uint64_t *vals);
/* rows in the reference's export order, cut after `limit` rows (0 = all);
 * others[nvals] = sums of the rows behind the cut (monit-objects-fwm.c:405-449, 496-589) */
long xo_fwm_export(int fwm, int limit, uint8_t *keys, uint64_t *vals, uint64_t *others, int *hit_limit);
/* xdiv() of filter.h:396-418 (div / div_r / div_l) */
int xo_xdiv(uint64_t dividend, uint64_t divisor, int is_log, int k);
/* service names of the 65536 ports for portstr() / ppstr() (NULL or "" = the
 * port has none; NULL table = no port has one); port_to_str() and
 * ports_pair_to_str() of utils.c:28-77 over those names */
void xo_set_port_names(const char *const *names);
void xo_port_to_str(char *res, uint16_t port);
void xo_ports_pair_to_str(char *res, uint16_t port1, uint16_t port2);
/* geo / AS database image as the reference maps it from geo4.db / geo6.db
 * (kind 0) or as4.db / as6.db (kind 1); family 4 or 6; 0 bytes = none */
int xo_geodb_set(int kind, int family, const void *img, size_t bytes);
int xo_geodb_lookup(int kind, int family, const uint8_t *addr, uint8_t *payload);
/* text of a TCP flags byte, as tcp_flags_to_str() (utils.c:22-26) gives it */
void xo_tcp_flags_str(uint8_t tf, char out[40]);

#ifdef __cplusplus
}
#endif
#endif
