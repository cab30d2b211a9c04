/*
 * xe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see xe_oracle.h).
 *
 * Scalar restatement of the reference's per-record flow path.  The structure
 * deliberately follows the reference (array-of-structs flow record zeroed per
 * record, 12-byte bitwise-trie nodes walked bit by bit, postfix stack machine)
 * so that each function can be read next to the file:line it cites.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stddef.h>
#include <ctype.h>
#include <math.h>
#include <endian.h>
#include <dirent.h>
#include <limits.h>
#include <arpa/inet.h>

#include "xe_oracle.h"

/* ------------------------------------------------------------------ */
/* field table: facts of netflow.def:2-64 (name, element id, type,     */
/* minimum and maximum encoded size)                                   */
/* ------------------------------------------------------------------ */
enum { T_ADDR = 0, T_INT = 1, T_MAC = 2, T_STR = 3 };

#define XO_FIELDS(X) \
	X(in_bytes, 1, T_INT, 4, 8) \
	X(in_pkts, 2, T_INT, 4, 8) \
	X(protocol, 4, T_INT, 1, 1) \
	X(src_tos, 5, T_INT, 1, 1) \
	X(dst_tos, 55, T_INT, 1, 1) \
	X(tcp_flags, 6, T_INT, 1, 2) \
	X(l4_src_port, 7, T_INT, 2, 2) \
	X(ip4_src_addr, 8, T_ADDR, 4, 4) \
	X(src_mask, 9, T_INT, 1, 1) \
	X(input_snmp, 10, T_INT, 2, 4) \
	X(l4_dst_port, 11, T_INT, 2, 2) \
	X(ip4_dst_addr, 12, T_ADDR, 4, 4) \
	X(dst_mask, 13, T_INT, 1, 1) \
	X(output_snmp, 14, T_INT, 2, 4) \
	X(ip4_next_hop, 15, T_ADDR, 4, 4) \
	X(src_as, 16, T_INT, 2, 4) \
	X(dst_as, 17, T_INT, 2, 4) \
	X(bgp_next_hop, 18, T_ADDR, 4, 4) \
	X(last_switched, 21, T_INT, 4, 4) \
	X(first_switched, 22, T_INT, 4, 4) \
	X(ip6_src_addr, 27, T_ADDR, 16, 16) \
	X(ip6_dst_addr, 28, T_ADDR, 16, 16) \
	X(icmp_type, 32, T_INT, 2, 2) \
	X(sampler_id, 48, T_INT, 1, 4) \
	X(min_ttl, 52, T_INT, 1, 1) \
	X(max_ttl, 53, T_INT, 1, 1) \
	X(frag_id, 54, T_INT, 2, 4) \
	X(src_mac, 56, T_MAC, 6, 6) \
	X(dst_mac, 57, T_MAC, 6, 6) \
	X(src_vlan, 58, T_INT, 2, 2) \
	X(dst_vlan, 59, T_INT, 2, 2) \
	X(ip_protocol_version, 60, T_INT, 1, 1) \
	X(direction, 61, T_INT, 1, 1) \
	X(if_name, 82, T_STR, 1, 16) \
	X(fwd_status, 89, T_INT, 1, 1) \
	X(flow_start_ms, 152, T_INT, 8, 8) \
	X(flow_end_ms, 153, T_INT, 8, 8) \
	X(flow_end_reason, 136, T_INT, 1, 1) \
	X(ip_ttl, 192, T_INT, 1, 1) \
	X(ioctets, 231, T_INT, 8, 8) \
	X(roctets, 232, T_INT, 2, 8) \
	X(ipackets, 298, T_INT, 8, 8) \
	X(rpackets, 239, T_INT, 8, 8) \
	X(ingrs_vrf, 234, T_INT, 4, 4) \
	X(egrs_vrf, 235, T_INT, 4, 4) \
	X(src_dot1q_vlan, 243, T_INT, 2, 2) \
	X(src_dot1q_cvlan, 245, T_INT, 2, 2) \
	X(dst_dot1q_vlan, 254, T_INT, 2, 2) \
	X(dst_dot1q_cvlan, 255, T_INT, 2, 2) \
	X(obs_time_ms, 323, T_INT, 8, 8) \
	X(vas_session_id, 2000, T_INT, 8, 8) \
	X(vas_http_host, 2001, T_STR, 0, 256) \
	X(dev_ip, 65500, T_ADDR, 4, 4) \
	X(dns_name, 65510, T_STR, 1, 256) \
	X(dns_ips, 65511, T_STR, 1, 512) \
	X(sni, 65512, T_STR, 1, 256) \
	X(class0, 65520, T_STR, 1, 32) \
	X(class1, 65521, T_STR, 1, 32) \
	X(class2, 65522, T_STR, 1, 32) \
	X(class3, 65523, T_STR, 1, 32) \
	X(class4, 65524, T_STR, 1, 32) \
	X(pad1, 65530, T_INT, 1, 1) \
	X(pad2, 65531, T_INT, 2, 2)

/* flow-info.h:10-33 */
struct xo_flow
{
#define X(NAME, ID, TYPE, MN, MX) \
	uint8_t NAME[MX]; \
	int NAME##_size; \
	int has_##NAME;
	XO_FIELDS(X)
#undef X
	void *payload_ptr;

	uint8_t dev_ip6[16];
	int dev_ip6_size;
	int has_dev_ip6;

	uint8_t dev_id[4];
	int dev_id_size;
	int has_dev_id;

	uint8_t dev_mark[4];
	int dev_mark_size;
	int has_dev_mark;

	uint32_t sampling_rate;
};

_Static_assert(sizeof(struct xo_flow) == XO_FLOW_INFO_SIZE,
	"flow record layout must match the reference (2312 bytes on x86-64)");

enum
{
#define X(NAME, ID, TYPE, MN, MX) F_##NAME,
	XO_FIELDS(X)
#undef X
	F_dev_ip6,
	F_dev_id,
	F_dev_mark,
	F_COUNT
};

struct fdesc
{
	const char *name;
	int id, type, mn, mx;
	size_t off, off_size, off_has;
};

static const struct fdesc fields[F_COUNT] = {
#define X(NAME, ID, TYPE, MN, MX) \
	{ #NAME, ID, TYPE, MN, MX, offsetof(struct xo_flow, NAME), \
	  offsetof(struct xo_flow, NAME##_size), offsetof(struct xo_flow, has_##NAME) },
	XO_FIELDS(X)
#undef X
	{ "dev_ip6", -1, T_ADDR, 16, 16, offsetof(struct xo_flow, dev_ip6),
	  offsetof(struct xo_flow, dev_ip6_size), offsetof(struct xo_flow, has_dev_ip6) },
	{ "dev_id", -1, T_INT, 4, 4, offsetof(struct xo_flow, dev_id),
	  offsetof(struct xo_flow, dev_id_size), offsetof(struct xo_flow, has_dev_id) },
	{ "dev_mark", -1, T_INT, 4, 4, offsetof(struct xo_flow, dev_mark),
	  offsetof(struct xo_flow, dev_mark_size), offsetof(struct xo_flow, has_dev_mark) },
};

int xo_nfields(void) { return F_COUNT; }
const char *xo_field_name(int i) { return fields[i].name; }
int xo_field_id(int i) { return fields[i].id; }
int xo_field_type(int i) { return fields[i].type; }
int xo_field_min(int i) { return fields[i].mn; }
int xo_field_max(int i) { return fields[i].mx; }
size_t xo_field_off(int i) { return fields[i].off; }
size_t xo_field_off_size(int i) { return fields[i].off_size; }
size_t xo_field_off_has(int i) { return fields[i].off_has; }
size_t xo_off_sampling_rate(void) { return offsetof(struct xo_flow, sampling_rate); }
size_t xo_off_payload_ptr(void) { return offsetof(struct xo_flow, payload_ptr); }

/* element id -> field index, the table of netflow.c:56,824-837 */
static int16_t id2field[65536];
static int id2field_ready = 0;

static void
id2field_init(void)
{
	int i;
	if (id2field_ready) {
		return;
	}
	for (i = 0; i < 65536; i++) {
		id2field[i] = -1;
	}
	for (i = 0; i < F_dev_ip6; i++) {
		id2field[fields[i].id] = (int16_t)i;
	}
	id2field_ready = 1;
}

#define FBYTES(FLOW, IDX) ((uint8_t *)(FLOW) + fields[IDX].off)
#define FSIZE(FLOW, IDX) (*(int *)((uint8_t *)(FLOW) + fields[IDX].off_size))
#define FHAS(FLOW, IDX) (*(int *)((uint8_t *)(FLOW) + fields[IDX].off_has))

/* ------------------------------------------------------------------ */
/* the receive buffer: struct flow_packet_info (flow-info.h:35-46).    */
/* The reference re-uses one 64 KiB stack buffer, so bytes past `len`  */
/* are whatever the previous datagram left; the oracle defines them as */
/* zero and reads past the 64 KiB as zero too.                         */
/* ------------------------------------------------------------------ */
#define RAWSZ 65536

struct pktinfo
{
	uint32_t src_addr_ipv4;  /* network order */
	uint32_t source_id;      /* raw bytes of the header word */
	uint32_t epoch;
	int sampling_rate;
	uint8_t raw[RAWSZ + 64];
	int len;
};

static void
rd(const struct pktinfo *p, long off, void *dst, long n)
{
	long i;
	uint8_t *d = dst;
	for (i = 0; i < n; i++) {
		long o = off + i;
		d[i] = ((o >= 0) && (o < RAWSZ)) ? p->raw[o] : 0;
	}
}

static uint16_t
rd16(const struct pktinfo *p, long off)
{
	uint8_t b[2];
	rd(p, off, b, 2);
	return (uint16_t)((b[0] << 8) | b[1]);
}

/* ------------------------------------------------------------------ */
/* IP lists: iplist.c                                                  */
/* ------------------------------------------------------------------ */
struct trie_node  /* iplist.c:29-33 */
{
	uint32_t next[2];
	int is_leaf;
};

struct iplist  /* iplist.c:35-41 */
{
	char name[PATH_MAX];
	struct trie_node *nodes4, *nodes6;
	size_t n4, n6;
};

static struct iplist *iplists = NULL;
static size_t n_iplists = 0;

/* iplist.c:52-158 (same walk for both families; addr = bytes as stored) */
/* A mask longer than the address makes the reference walk past the address
 * bytes (iplist.c:75) and put the leaf flag on a node deeper than any lookup
 * goes (lookups stop after 32 / 128 bits, iplist.c:171,200): what lookups can
 * see is the full-length path WITHOUT a leaf.  maxbits restates that. */
static int
trie_add(struct trie_node **nodes, size_t *n, const uint8_t *addr, int mask, int maxbits)
{
	int i;
	uint32_t node = 0;

	if (!*nodes) {
		*nodes = calloc(1, sizeof(struct trie_node));
		if (!*nodes) {
			return 0;
		}
		*n = 1;
	}
	for (i = 0; (i < mask) && (i < maxbits); i++) {
		int bit = !!(addr[i / 8] & (1 << (7 - (i % 8))));
		uint32_t next = (*nodes)[node].next[bit];
		struct trie_node *tmp;

		if (next) {
			node = next;
			continue;
		}
		/* grow geometrically; the reference grows by one (iplist.c:85) */
		if ((*n & (*n - 1)) == 0) {
			tmp = realloc(*nodes, (*n * 2) * sizeof(struct trie_node));
			if (!tmp) {
				return 0;
			}
			*nodes = tmp;
		}
		memset(&(*nodes)[*n], 0, sizeof(struct trie_node));
		(*nodes)[node].next[bit] = (uint32_t)*n;
		node = (uint32_t)*n;
		(*n)++;
	}
	if (mask <= maxbits) {
		(*nodes)[node].is_leaf = 1;
	}
	return 1;
}

/* iplist.c:160-216: deepest node reached, NOT longest-prefix match */
static int
trie_match(const struct trie_node *nodes, const uint8_t *addr, int nbits)
{
	int i;
	uint32_t node = 0;

	if (!nodes) {
		return 0;
	}
	for (i = 0; i < nbits; i++) {
		int bit = !!(addr[i / 8] & (1 << (7 - (i % 8))));
		uint32_t next = nodes[node].next[bit];
		if (!next) {
			break;
		}
		node = next;
	}
	return nodes[node].is_leaf ? 1 : 0;
}

static char *
str_trim(char *str)  /* utils.h:38-63 */
{
	char *end;
	while (isspace((unsigned char)*str)) {
		str++;
	}
	if (*str == 0) {
		return str;
	}
	end = str + strlen(str) - 1;
	while (end > str && isspace((unsigned char)*end)) {
		end--;
	}
	end[1] = '\0';
	return str;
}

/* iplist.c:218-297, reading from an already opened stream */
static int
iplist_load_stream(FILE *f, const char *listname)
{
	struct iplist *tmp, *l;

	tmp = realloc(iplists, (n_iplists + 1) * sizeof(struct iplist));
	if (!tmp) {
		return 0;
	}
	iplists = tmp;
	l = &iplists[n_iplists];
	memset(l, 0, sizeof(struct iplist));
	strncpy(l->name, listname, sizeof(l->name) - 1);

	for (;;) {
		char line[1024];
		char *str_addr, *mask_sym;
		uint8_t a4[4], a6[16];
		int mask = 32;   /* also the default for IPv6 lines (iplist.c:248) */

		if (!fgets(line, sizeof(line) - 1, f)) {
			break;
		}
		/* a last line without '\n' sets EOF and is dropped (iplist.c:254) */
		if (feof(f)) {
			break;
		}
		line[strcspn(line, "\n#")] = 0;
		str_addr = str_trim(line);
		if (strlen(str_addr) == 0) {
			continue;
		}
		mask_sym = strchr(str_addr, '/');
		if (mask_sym) {
			char *endptr;
			*mask_sym = '\0';
			mask_sym++;
			mask = (int)strtol(mask_sym, &endptr, 10);
			if (*endptr != '\0') {
				continue;
			}
		}
		if (inet_pton(AF_INET, str_addr, a4)) {
			trie_add(&l->nodes4, &l->n4, a4, mask, 32);
		} else if (inet_pton(AF_INET6, str_addr, a6)) {
			trie_add(&l->nodes6, &l->n6, a6, mask, 128);
		}
	}
	n_iplists++;
	return 1;
}

int
xo_iplist_add_text(const char *name, const char *text, size_t len)
{
	FILE *f;
	int rc;

	if (len == 0) {
		/* fmemopen refuses empty buffers; an empty file is an empty list */
		f = fopen("/dev/null", "r");
	} else {
		f = fmemopen((void *)text, len, "r");
	}
	if (!f) {
		return 0;
	}
	rc = iplist_load_stream(f, name);
	fclose(f);
	return rc;
}

/* iplist.c:299-322 */
int
xo_iplist_load_dir(const char *dirname)
{
	DIR *d = opendir(dirname);
	struct dirent *dir;

	if (!d) {
		return 0;
	}
	while ((dir = readdir(d)) != NULL) {
		char path[PATH_MAX];
		FILE *f;
		if (dir->d_name[0] == '.') {
			continue;
		}
		snprintf(path, sizeof(path), "%s/%s", dirname, dir->d_name);
		f = fopen(path, "r");
		if (!f) {
			continue;
		}
		iplist_load_stream(f, dir->d_name);
		fclose(f);
	}
	closedir(d);
	return 1;
}

/* iplist.c:324-338: first list with that exact name */
static struct iplist *
iplist_by_name(const char *name)
{
	size_t i;
	for (i = 0; i < n_iplists; i++) {
		if (strcmp(name, iplists[i].name) == 0) {
			return &iplists[i];
		}
	}
	return NULL;
}

int
xo_iplist_match4(const char *name, uint32_t addr)
{
	struct iplist *l = iplist_by_name(name);
	if (!l) {
		return -1;
	}
	return trie_match(l->nodes4, (const uint8_t *)&addr, 32);
}

int
xo_iplist_match6(const char *name, const uint8_t addr[16])
{
	struct iplist *l = iplist_by_name(name);
	if (!l) {
		return -1;
	}
	return trie_match(l->nodes6, addr, 128);
}

size_t
xo_iplist_nodes(const char *name, int family)
{
	struct iplist *l = iplist_by_name(name);
	if (!l) {
		return 0;
	}
	return (family == 6) ? l->n6 : l->n4;
}

/* ------------------------------------------------------------------ */
/* filter language: tokens (filter.h:44-101, filter-lexer.c)           */
/* ------------------------------------------------------------------ */
#define TOKEN_MAX 512

/* facts of filter.def:2-37: keyword, operand type, src field, dst field */
enum { BT_ADDR4, BT_ADDR6, BT_RANGE, BT_MAC, BT_STRING };

struct ffield
{
	const char *kw;
	int type;
	int src, dst;
};

static const struct ffield ffields[] = {
	{ "host", BT_ADDR4, F_ip4_src_addr, F_ip4_dst_addr },
	{ "net", BT_ADDR4, F_ip4_src_addr, F_ip4_dst_addr },
	{ "host6", BT_ADDR6, F_ip6_src_addr, F_ip6_dst_addr },
	{ "net6", BT_ADDR6, F_ip6_src_addr, F_ip6_dst_addr },
	{ "port", BT_RANGE, F_l4_src_port, F_l4_dst_port },
	{ "proto", BT_RANGE, F_protocol, F_protocol },
	{ "tos", BT_RANGE, F_src_tos, F_dst_tos },
	{ "tcp-flags", BT_RANGE, F_tcp_flags, F_tcp_flags },
	{ "ifidx", BT_RANGE, F_input_snmp, F_output_snmp },
	{ "as", BT_RANGE, F_src_as, F_dst_as },
	{ "min-ttl", BT_RANGE, F_min_ttl, F_min_ttl },
	{ "max-ttl", BT_RANGE, F_max_ttl, F_max_ttl },
	{ "frag-id", BT_RANGE, F_frag_id, F_frag_id },
	{ "mac", BT_MAC, F_src_mac, F_dst_mac },
	{ "vlan", BT_RANGE, F_src_vlan, F_dst_vlan },
	{ "endreason", BT_RANGE, F_flow_end_reason, F_flow_end_reason },
	{ "d1qvlan", BT_RANGE, F_src_dot1q_vlan, F_dst_dot1q_vlan },
	{ "d1qcvlan", BT_RANGE, F_src_dot1q_cvlan, F_dst_dot1q_cvlan },
	{ "bgp-nh", BT_ADDR4, F_bgp_next_hop, F_bgp_next_hop },
	{ "dir", BT_RANGE, F_direction, F_direction },
	{ "fwdst", BT_RANGE, F_fwd_status, F_fwd_status },
	{ "smplr", BT_RANGE, F_sampler_id, F_sampler_id },
	{ "vrf", BT_RANGE, F_ingrs_vrf, F_egrs_vrf },
	{ "vashost", BT_STRING, F_vas_http_host, F_vas_http_host },
	{ "dev-ip", BT_ADDR4, F_dev_ip, F_dev_ip },
	{ "dev-ip6", BT_ADDR4, F_dev_ip6, F_dev_ip6 },  /* typed ADDR4: filter.def:27 */
	{ "dev-id", BT_RANGE, F_dev_id, F_dev_id },
	{ "dev-mark", BT_RANGE, F_dev_mark, F_dev_mark },
	{ "class0", BT_STRING, F_class0, F_class0 },
	{ "class1", BT_STRING, F_class1, F_class1 },
	{ "class2", BT_STRING, F_class2, F_class2 },
	{ "class3", BT_STRING, F_class3, F_class3 },
	{ "class4", BT_STRING, F_class4, F_class4 },
	{ "dns-name", BT_STRING, F_dns_name, F_dns_name },
	{ "dns-ips", BT_STRING, F_dns_ips, F_dns_ips },
	{ "sni", BT_STRING, F_sni, F_sni },
};
#define N_FFIELDS ((int)(sizeof(ffields) / sizeof(ffields[0])))

/* facts of filter-ag.def:2-8 */
struct agfield
{
	const char *kw;
	int field;
	int scale;
};

static const struct agfield agfields[] = {
	{ "octets", F_in_bytes, 1 }, { "bits", F_in_bytes, 8 },
	{ "packets", F_in_pkts, 1 }, { "ioctets", F_ioctets, 1 },
	{ "roctets", F_roctets, 1 }, { "ipackets", F_ipackets, 1 },
	{ "rpackets", F_rpackets, 1 },
};
#define N_AGFIELDS ((int)(sizeof(agfields) / sizeof(agfields[0])))

enum
{
	TK_ID, TK_INT_RANGE, TK_MAC, TK_STRING, TK_LPAREN, TK_RPAREN,
	TK_OR, TK_AND, TK_NOT, TK_SRC, TK_DST, TK_ASC, TK_DESC,
	TK_DIV, TK_DIV_R, TK_DIV_L, TK_MIN, TK_MFREQ, TK_GEOIP, TK_ASN, TK_ASD,
	TK_TFSTR, TK_PORTSTR, TK_PPSTR, TK_COMMA,
	TK_FIELD0 = 100,      /* + index into ffields */
	TK_AGFIELD0 = 200     /* + index into agfields */
};

struct token
{
	int id;
	int str_len;
	/* the reference keeps these in one union (filter.h:95-100); the
	 * oracle keeps them apart, no parser path reads a stale member */
	char str[TOKEN_MAX];
	int low, high;
	uint8_t mac[6];
};

struct finput  /* filter.h:277-287 */
{
	const char *s;
	int end;
	int line, col;
	int error;
	char errmsg[1024];
	struct token tok;
};

static void
mkerror(struct finput *f, const char *msg)  /* filter.c:40-45 */
{
	f->error = 1;
	snprintf(f->errmsg, sizeof(f->errmsg), "Line %d, col %d: %s", f->line, f->col, msg);
}

/* utils.c:79-123 */
static int
hexval(char c)
{
	if ((c >= 'a') && (c <= 'f')) return c - 'a' + 10;
	if ((c >= 'A') && (c <= 'F')) return c - 'A' + 10;
	if ((c >= '0') && (c <= '9')) return c - '0';
	return 0;
}

int
xo_mac_addr_read(const char *s, uint8_t mac[6])
{
	int i;
	uint8_t m[6];

	if (strlen(s) != 17) {
		return 0;
	}
	for (i = 0; i < 6; i++) {
		char delim = s[i * 3 + 2];
		if ((delim != ':') && (delim != '\0')) {
			return 0;
		}
		m[i] = (uint8_t)((hexval(s[i * 3 + 0]) << 4) | hexval(s[i * 3 + 1]));
	}
	memcpy(mac, m, 6);
	return 1;
}

/* filter-lexer.c:43-120.  Returns with q->end or q->error set, or at a
 * non-blank character. */
static void
lex_whitespace(struct finput *q)
{
	for (;;) {
		if (*q->s == '\0') { q->end = 1; return; }
		if ((*q->s == ' ') || (*q->s == '\t')) {
			q->col++;
		} else if ((*q->s == '\n') || (*q->s == '\r')) {
			q->line++;
			q->col = 1;
		} else if (*q->s == '/') {
			q->s++;
			if (*q->s == '\0') {
				q->end = 1; q->error = 1;
				strcpy(q->errmsg, "unexpected end of input after '/'");
				return;
			}
			q->col++;
			if (*q->s == '*') {
				/* filter-lexer.c:43-68 */
				for (;;) {
					q->s++;
					if (*q->s == '\0') goto unexp;
					q->col++;
					if (*q->s == '*') {
						q->s++;
						if (*q->s == '\0') goto unexp;
						q->col++;
						if (*q->s == '/') {
							break;
						}
					} else if (*q->s == '\n') {
						q->line++;
						q->col = 1;
					}
				}
			} else if (*q->s == '/') {
				/* filter-lexer.c:70-86 */
				for (;;) {
					q->s++;
					if (*q->s == '\0') { q->end = 1; return; }
					q->col++;
					if (*q->s == '\n') {
						q->line++;
						q->col = 1;
						break;
					}
				}
			} else {
				q->col--;
				q->s--;
				break;
			}
		} else {
			break;
		}
		q->s++;
	}
	return;
unexp:
	q->end = 1; q->error = 1;
	strcpy(q->errmsg, "unexpected end of input inside the comment");
}

static int
id_sym(int c)  /* filter-lexer.c:122-140 */
{
	return !(c == '\0' || c == ' ' || c == '\t' || c == ',' || c == '\r'
		|| c == '\n' || c == '(' || c == ')');
}

static const char *const geoip_kw[] = {
	"CONTINENT", "COUNTRY_CODE", "COUNTRY", "STATE", "CITY", "ZIP", "LAT", "LONG"
};

/* which geo function the last TK_GEOIP token was (the reference has one token
 * id per function; the oracle folds them into one and remembers the index) */
static int last_geo_kw;

/* filter-lexer.c:142-209; the order of the comparisons is the reference's */
static int
keyword(const char *s, int *id)
{
	int i;
	if (!strcasecmp(s, "src")) { *id = TK_SRC; return 1; }
	if (!strcasecmp(s, "dst")) { *id = TK_DST; return 1; }
	for (i = 0; i < N_FFIELDS; i++) {
		if (!strcasecmp(s, ffields[i].kw)) { *id = TK_FIELD0 + i; return 1; }
	}
	if (!strcasecmp(s, "or")) { *id = TK_OR; return 1; }
	if (!strcasecmp(s, "and")) { *id = TK_AND; return 1; }
	if (!strcasecmp(s, "not")) { *id = TK_NOT; return 1; }
	if (!strcasecmp(s, "asc")) { *id = TK_ASC; return 1; }
	if (!strcasecmp(s, "desc")) { *id = TK_DESC; return 1; }
	if (!strcasecmp(s, "div")) { *id = TK_DIV; return 1; }
	if (!strcasecmp(s, "div_r")) { *id = TK_DIV_R; return 1; }
	if (!strcasecmp(s, "div_l")) { *id = TK_DIV_L; return 1; }
	if (!strcasecmp(s, "min")) { *id = TK_MIN; return 1; }
	if (!strcasecmp(s, "mfreq")) { *id = TK_MFREQ; return 1; }
	if (!strcasecmp(s, "tfstr")) { *id = TK_TFSTR; return 1; }
	if (!strcasecmp(s, "portstr")) { *id = TK_PORTSTR; return 1; }
	if (!strcasecmp(s, "ppstr")) { *id = TK_PPSTR; return 1; }
	for (i = 0; i < 8; i++) {
		if (!strcasecmp(s, geoip_kw[i])) { *id = TK_GEOIP; last_geo_kw = i; return 1; }
	}
	if (!strcasecmp(s, "asn")) { *id = TK_ASN; return 1; }
	if (!strcasecmp(s, "asd")) { *id = TK_ASD; return 1; }
	for (i = 0; i < N_AGFIELDS; i++) {
		if (!strcasecmp(s, agfields[i].kw)) { *id = TK_AGFIELD0 + i; return 1; }
	}
	return 0;
}

/* filter-lexer.c:211-296 */
static void
read_token(struct finput *q)
{
	struct token *t = &q->tok;

	t->str_len = 0;
	lex_whitespace(q);
	if (q->end || q->error) {
		return;
	}
	if ((*q->s == '(') || (*q->s == ')') || (*q->s == ',')) {
		t->id = (*q->s == '(') ? TK_LPAREN : (*q->s == ')') ? TK_RPAREN : TK_COMMA;
		t->str[0] = *q->s;
		t->str[1] = '\0';
		t->str_len = 1;
		q->s++;
		q->col++;
		return;
	}
	if (*q->s == '\'') {
		q->s++;
		q->col++;
		do {
			/* the reference runs off the end of an unterminated
			 * string; the oracle reports an error instead */
			if ((*q->s == '\0') || (t->str_len >= TOKEN_MAX - 1)) {
				q->end = 1; q->error = 1;
				strcpy(q->errmsg, "unterminated string");
				return;
			}
			t->str[t->str_len++] = *q->s;
			q->s++;
			q->col++;
		} while (*q->s != '\'');
		q->s++;
		q->col++;
		t->str[t->str_len] = '\0';
		t->id = TK_STRING;
		return;
	}
	do {
		if (t->str_len >= TOKEN_MAX - 1) {
			q->end = 1; q->error = 1;
			strcpy(q->errmsg, "token too long");
			return;
		}
		t->str[t->str_len++] = *q->s;
		q->s++;
		q->col++;
	} while (id_sym(*q->s));
	t->str[t->str_len] = '\0';

	if (xo_mac_addr_read(t->str, t->mac)) {
		t->id = TK_MAC;
		return;
	}
	if (!keyword(t->str, &t->id)) {
		char *endptr;
		long res = strtol(t->str, &endptr, 0);
		if (*endptr == '\0') {
			t->id = TK_INT_RANGE;
			t->low = t->high = (int)res;
		} else if (*endptr == '-') {
			char *endptr2;
			long res2 = strtol(endptr + 1, &endptr2, 0);
			if (*endptr2 == '\0') {
				t->id = TK_INT_RANGE;
				t->low = (int)res;
				t->high = (int)res2;
			} else {
				t->id = TK_ID;
			}
		} else {
			t->id = TK_ID;
		}
	}
}

/* ------------------------------------------------------------------ */
/* filter expressions (filter.h:103-275)                               */
/* ------------------------------------------------------------------ */
enum { DIR_NONE = 0, DIR_SRC = 1, DIR_DST = 2, DIR_BOTH = 3 };
enum { OP_BASIC, OP_NOT, OP_AND, OP_OR };
enum { FN_NONE, FN_DIV, FN_DIV_R, FN_DIV_L, FN_MIN, FN_TFSTR, FN_PORTSTR, FN_PPSTR, FN_MFREQ, FN_GEO, FN_ASN, FN_ASD };

struct bdata  /* filter.h:109-119 */
{
	int is_list;
	int low, high;
	int ip_version, mask_len;
	uint8_t addr[16], mask[16];   /* network byte order */
	struct iplist *list;
	size_t list_idx;              /* iplists may be realloc'd: keep the index */
	char *str;
	uint8_t mac[6];
};

struct basic  /* filter.h:234-254 */
{
	int type;
	int ffield;       /* index into ffields, or -1 for functions */
	int direction;
	size_t n;
	struct bdata *data;
	int func;
	/* div: filter.h:153-164, min: filter.h:166-174 */
	unsigned int a_off, a_size, b_off, b_size;
	int is_log, k;
	uint64_t *freqmap;   /* mfreq: filter.h:176-186 */
	int geo_field;       /* geo functions: enum GEOIP_FIELD, geoip.h:23-28 */
};

struct fop { int op; struct basic *arg; };
struct fexpr { size_t n; struct fop *ops; };

static int
add_basic(struct fexpr *e, int type, int ffield, int dir)  /* filter.c:47-83 */
{
	struct basic *fb = calloc(1, sizeof(struct basic));
	struct fop *tmp;
	if (!fb) return 0;
	fb->type = type;
	fb->ffield = ffield;
	fb->direction = dir;
	tmp = realloc(e->ops, sizeof(struct fop) * (e->n + 1));
	if (!tmp) { free(fb); return 0; }
	e->ops = tmp;
	e->ops[e->n].op = OP_BASIC;
	e->ops[e->n].arg = fb;
	e->n++;
	return 1;
}

static int
add_op(struct fexpr *e, int op)  /* filter.c:234-253 */
{
	struct fop *tmp = realloc(e->ops, sizeof(struct fop) * (e->n + 1));
	if (!tmp) return 0;
	e->ops = tmp;
	e->ops[e->n].op = op;
	e->ops[e->n].arg = NULL;
	e->n++;
	return 1;
}

/* filter.c:86-156 */
static int
id_to_addr(struct finput *f, const char *host, struct bdata *d)
{
	char tmp[TOKEN_MAX];
	char *mask_sym;
	uint8_t a[16];
	int rc, i, nbits;

	strcpy(tmp, host);
	d->mask_len = -1;
	mask_sym = strchr(tmp, '/');
	if (mask_sym) {
		char *endptr;
		*mask_sym = '\0';
		mask_sym++;
		d->mask_len = (int)strtol(mask_sym, &endptr, 10);
		if (*endptr != '\0') {
			mkerror(f, "Incorrect network mask");
			return 0;
		}
	}
	rc = inet_pton((d->ip_version == 4) ? AF_INET : AF_INET6, tmp, a);
	if (rc != 1) {
		mkerror(f, "Can't parse IP address");
		return 0;
	}
	nbits = (d->ip_version == 4) ? 32 : 128;
	if (d->mask_len < 0) {
		d->mask_len = nbits;
	}
	memset(d->mask, 0, 16);
	memset(d->addr, 0, 16);
	/* bit i (from the most significant) set for i < mask_len; the
	 * reference shifts 1UL << (31 - i), undefined past the width, the
	 * oracle stops at the width */
	for (i = 0; (i < d->mask_len) && (i < nbits); i++) {
		d->mask[i / 8] |= (uint8_t)(1 << (7 - (i % 8)));
	}
	for (i = 0; i < nbits / 8; i++) {
		d->addr[i] = a[i] & d->mask[i];
	}
	return 1;
}

/* filter.c:158-232 */
static int
add_to_basic(struct finput *f, struct fexpr *e, struct token *tok, int type)
{
	struct fop *fo;
	struct basic *fb;
	struct bdata *tmp, *d;

	if (e->n < 1) return 0;
	fo = &e->ops[e->n - 1];
	if (fo->op != OP_BASIC) return 0;
	fb = fo->arg;
	tmp = realloc(fb->data, sizeof(struct bdata) * (fb->n + 1));
	if (!tmp) return 0;
	fb->data = tmp;
	d = &fb->data[fb->n];
	/* the reference leaves a freshly realloc'd slot uninitialised; an ID
	 * under a RANGE/MAC/STRING rule therefore compares against garbage.
	 * The oracle zeroes the slot. */
	memset(d, 0, sizeof(*d));

	if (tok->id == TK_ID) {
		if ((type == BT_ADDR4) || (type == BT_ADDR6)) {
			struct iplist *l = iplist_by_name(tok->str);
			if (l) {
				d->list = l;
				d->list_idx = (size_t)(l - iplists);
				d->is_list = 1;
			} else {
				d->ip_version = (type == BT_ADDR4) ? 4 : 6;
				if (!id_to_addr(f, tok->str, d)) {
					return 0;
				}
			}
		}
	} else if (tok->id == TK_INT_RANGE) {
		d->low = tok->low;
		d->high = tok->high;
	} else if (tok->id == TK_MAC) {
		memcpy(d->mac, tok->mac, 6);
	} else if (tok->id == TK_STRING) {
		d->str = strdup(tok->str);
	} else {
		return 0;
	}
	fb->n++;
	return 1;
}

/* ------------------------------------------------------------------ */
/* parser (filter-parser.c:23-255, filter-parser-funcs.c)              */
/* ------------------------------------------------------------------ */
#define READ_TOKEN_CHECK(I) do { read_token(I); if ((I)->error) return 0; } while (0)

static int expression(struct finput *q, struct fexpr *e);

static int
accept_(struct finput *i, int token)  /* filter-parser.c:23-33 */
{
	if (i->tok.id != token) {
		return 0;
	}
	READ_TOKEN_CHECK(i);
	return 1;
}

static int
is_value(int id)
{
	return (id == TK_ID) || (id == TK_INT_RANGE) || (id == TK_MAC) || (id == TK_STRING);
}

/* filter-parser.c:46-86.  After `or`, a token that is not a value is
 * skipped (line 79) before the rest is parsed as a new expression. */
static int
id(struct finput *f, struct fexpr *e, int type)
{
	if (!is_value(f->tok.id)) {
		mkerror(f, "Expected ID, INT, INT_RANGE, MAC or STRING");
		return 0;
	}
	add_to_basic(f, e, &f->tok, type);
	READ_TOKEN_CHECK(f);
	for (;;) {
		if (!accept_(f, TK_OR)) {
			break;
		}
		if (is_value(f->tok.id)) {
			add_to_basic(f, e, &f->tok, type);
			READ_TOKEN_CHECK(f);
			continue;
		}
		READ_TOKEN_CHECK(f);
		expression(f, e);
		add_op(e, OP_OR);
	}
	return 1;
}

static int
rule_without_direction(struct finput *q, struct fexpr *e, int dir)  /* :88-104 */
{
	int i;
	for (i = 0; i < N_FFIELDS; i++) {
		if (accept_(q, TK_FIELD0 + i)) {
			add_basic(e, ffields[i].type, i, dir);
			return id(q, e, ffields[i].type);
		}
	}
	return 0;
}

/* filter-parser-funcs.c:39-53 */
static int
aggr_field_off_size(struct finput *q, unsigned int *off, unsigned int *size)
{
	int i;
	for (i = 0; i < N_AGFIELDS; i++) {
		if (accept_(q, TK_AGFIELD0 + i)) {
			*off = (unsigned int)fields[agfields[i].field].off;
			*size = (unsigned int)fields[agfields[i].field].mx;
			return 1;
		}
	}
	return 0;
}

/* filter-parser-funcs.c:55-79, :194-213 */
static int
nonaggr_field(struct finput *in, unsigned int *off, unsigned int *size)
{
	int dir = DIR_BOTH, i;

	if (accept_(in, TK_SRC)) {
		dir = DIR_SRC;
	} else if (accept_(in, TK_DST)) {
		dir = DIR_DST;
	}
	for (i = 0; i < N_FFIELDS; i++) {
		if (accept_(in, TK_FIELD0 + i)) {
			int fi;
			if (dir == DIR_SRC) {
				fi = ffields[i].src;
			} else if (dir == DIR_DST) {
				fi = ffields[i].dst;
			} else {
				if (ffields[i].src != ffields[i].dst) {
					mkerror(in, "This field requires direction");
					return 0;
				}
				fi = ffields[i].src;
			}
			/* the three virtual fields are not in netflow.def, so
			 * nf_field_to_off() fails for them (funcs.c:25-37) */
			if (fi >= F_dev_ip6) {
				return 0;
			}
			*off = (unsigned int)fields[fi].off;
			*size = (unsigned int)fields[fi].mx;
			return 1;
		}
	}
	return 0;
}

/* filter-parser-funcs.c:83-192 */
static int
function_div(struct finput *in, struct fexpr *e)
{
	struct basic tmp, *fb;
	int tok, func;

	memset(&tmp, 0, sizeof(tmp));
	tmp.k = 1;
	if (accept_(in, TK_DIV)) {
		tok = TK_DIV; func = FN_DIV;
	} else if (accept_(in, TK_DIV_L)) {
		tok = TK_DIV_L; func = FN_DIV_L; tmp.is_log = 1;
	} else if (accept_(in, TK_DIV_R)) {
		tok = TK_DIV_R; func = FN_DIV_R;
	} else {
		return 0;
	}
	if (!accept_(in, TK_LPAREN)) { mkerror(in, "Expected '(' after 'div'"); return 0; }
	if (!aggr_field_off_size(in, &tmp.a_off, &tmp.a_size)) {
		mkerror(in, "Expected aggregable field name"); return 0;
	}
	if (!accept_(in, TK_COMMA)) { mkerror(in, "Expected ',' after field name"); return 0; }
	if (!aggr_field_off_size(in, &tmp.b_off, &tmp.b_size)) {
		mkerror(in, "Expected aggregable field name after comma"); return 0;
	}
	if (tok != TK_DIV) {
		int k;
		if (!accept_(in, TK_COMMA)) {
			mkerror(in, "Expected ',' after second field name"); return 0;
		}
		k = in->tok.low;
		if (!accept_(in, TK_INT_RANGE)) { mkerror(in, "Expected INT"); return 0; }
		tmp.k = k;
	}
	if (!accept_(in, TK_RPAREN)) { mkerror(in, "Expected ')'"); return 0; }

	if (!add_basic(e, BT_RANGE, -1, DIR_NONE)) return 0;
	fb = e->ops[e->n - 1].arg;
	fb->func = func;
	fb->a_off = tmp.a_off; fb->a_size = tmp.a_size;
	fb->b_off = tmp.b_off; fb->b_size = tmp.b_size;
	fb->is_log = tmp.is_log; fb->k = tmp.k;
	return id(in, e, BT_RANGE);
}

/* filter-parser-funcs.c:216-282 */
static int
function_min(struct finput *in, struct fexpr *e)
{
	struct basic tmp, *fb;

	memset(&tmp, 0, sizeof(tmp));
	if (!accept_(in, TK_MIN)) return 0;
	if (!accept_(in, TK_LPAREN)) { mkerror(in, "Expected '(' after 'min'"); return 0; }
	if (!nonaggr_field(in, &tmp.a_off, &tmp.a_size)) {
		mkerror(in, "Incorrect field name"); return 0;
	}
	if (!accept_(in, TK_COMMA)) { mkerror(in, "Expected ',' after field name"); return 0; }
	if (!nonaggr_field(in, &tmp.b_off, &tmp.b_size)) {
		mkerror(in, "Incorrect field name after comma"); return 0;
	}
	if (!accept_(in, TK_RPAREN)) { mkerror(in, "Expected ')'"); return 0; }
	if (!add_basic(e, BT_RANGE, -1, DIR_NONE)) return 0;
	fb = e->ops[e->n - 1].arg;
	fb->func = FN_MIN;
	fb->a_off = tmp.a_off; fb->a_size = tmp.a_size;
	fb->b_off = tmp.b_off; fb->b_size = tmp.b_size;
	return id(in, e, BT_RANGE);
}

/* filter-parser-funcs.c:526-580 */
static int
function_tfstr(struct finput *in, struct fexpr *e)
{
	struct basic tmp, *fb;

	memset(&tmp, 0, sizeof(tmp));
	if (!accept_(in, TK_TFSTR)) return 0;
	if (!accept_(in, TK_LPAREN)) { mkerror(in, "Expected '(' after 'tfstr'"); return 0; }
	if (!nonaggr_field(in, &tmp.a_off, &tmp.a_size)) {
		mkerror(in, "Incorrect field name"); return 0;
	}
	if (!accept_(in, TK_RPAREN)) { mkerror(in, "Expected ')'"); return 0; }
	if (!add_basic(e, BT_STRING, -1, DIR_NONE)) return 0;
	fb = e->ops[e->n - 1].arg;
	fb->func = FN_TFSTR;
	fb->a_off = tmp.a_off; fb->a_size = tmp.a_size;
	return id(in, e, BT_STRING);
}

/* mfreq (filter-parser-funcs.c:284-365), the geo functions (:368-444), asn /
 * asd (:447-524), portstr (:582-637), ppstr (:639-705): one or two field
 * arguments, then a value list */
static int
function_other(struct finput *in, struct fexpr *e)
{
	struct basic tmp, *fb;
	int tok = in->tok.id, func, two = 0, type = BT_STRING;
	const char *what;
	char msg[64];

	memset(&tmp, 0, sizeof(tmp));
	switch (tok) {
		case TK_MFREQ: func = FN_MFREQ; two = 1; type = BT_RANGE; what = "'mfreq'"; break;
		case TK_PORTSTR: func = FN_PORTSTR; what = "'portstr'"; break;
		case TK_PPSTR: func = FN_PPSTR; two = 1; what = "'ppstr'"; break;
		case TK_GEOIP: func = FN_GEO; tmp.geo_field = last_geo_kw; what = "GeoIP function"; break;
		case TK_ASN: func = FN_ASN; type = BT_RANGE; what = "AS function"; break;
		case TK_ASD: func = FN_ASD; what = "AS function"; break;
		default: return 0;
	}
	if (!accept_(in, tok)) return 0;
	if (!accept_(in, TK_LPAREN)) {
		snprintf(msg, sizeof(msg), "Expected '(' after %s", what);
		mkerror(in, msg);
		return 0;
	}
	if (!nonaggr_field(in, &tmp.a_off, &tmp.a_size)) { mkerror(in, "Incorrect field name"); return 0; }
	if (two) {
		if (!accept_(in, TK_COMMA)) { mkerror(in, "Expected ',' after field name"); return 0; }
		if (!nonaggr_field(in, &tmp.b_off, &tmp.b_size)) { mkerror(in, "Incorrect field name after comma"); return 0; }
	}
	if (!accept_(in, TK_RPAREN)) { mkerror(in, "Expected ')'"); return 0; }
	if (!add_basic(e, type, -1, DIR_NONE)) return 0;
	fb = e->ops[e->n - 1].arg;
	fb->func = func;
	fb->a_off = tmp.a_off; fb->a_size = tmp.a_size;
	fb->b_off = tmp.b_off; fb->b_size = tmp.b_size;
	fb->geo_field = tmp.geo_field;
	if (func == FN_MFREQ) {
		fb->freqmap = calloc(65536, sizeof(uint64_t));
		if (!fb->freqmap) return 0;
	}
	return id(in, e, type);
}

/* filter-parser.c:107-156 */
static int
rule(struct finput *q, struct fexpr *e)
{
	if (function_div(q, e)) return 1;
	if (function_min(q, e)) return 1;
	if (function_tfstr(q, e)) return 1;
	if (function_other(q, e)) return 1;
	if (rule_without_direction(q, e, DIR_BOTH)) return 1;
	if (accept_(q, TK_SRC)) {
		return rule_without_direction(q, e, DIR_SRC);
	} else if (accept_(q, TK_DST)) {
		return rule_without_direction(q, e, DIR_DST);
	}
	return 0;
}

static int
factor(struct finput *f, struct fexpr *e)  /* filter-parser.c:158-189 */
{
	int inverse = 0;

	if (accept_(f, TK_NOT)) {
		inverse = 1;
	}
	if (rule(f, e)) {
		/* ok */
	} else if (accept_(f, TK_LPAREN)) {
		if (f->end) {
			/* '(' was the last token: it stays current, and the reference
			 * recurses here until its stack runs out.  Reported instead. */
			mkerror(f, "Unexpected end of input after '('");
			return 0;
		}
		if (!expression(f, e)) return 0;
		if (!accept_(f, TK_RPAREN)) {
			mkerror(f, "Expected ')' after expression");
			return 0;
		}
	} else {
		mkerror(f, "Syntax error");
		return 0;
	}
	if (inverse) {
		if (!add_op(e, OP_NOT)) return 0;
	}
	return 1;
}

static int
term(struct finput *f, struct fexpr *e)  /* filter-parser.c:191-206 */
{
	if (!factor(f, e)) return 0;
	while (accept_(f, TK_AND)) {
		if (!factor(f, e)) return 0;
		add_op(e, OP_AND);
	}
	return 1;
}

static int
expression(struct finput *f, struct fexpr *e)  /* filter-parser.c:208-223 */
{
	if (!term(f, e)) return 0;
	while (accept_(f, TK_OR)) {
		if (!term(f, e)) return 0;
		add_op(e, OP_OR);
	}
	return 1;
}

static struct fexpr *
parse_filter(struct finput *f)  /* filter-parser.c:225-255 */
{
	struct fexpr *e = calloc(1, sizeof(struct fexpr));
	if (!e) return NULL;
	read_token(f);
	if (f->error) return e;
	if (f->end) return e;   /* empty filter */
	if (!expression(f, e)) return e;
	if (!f->end) {
		char err[1024];
		snprintf(err, sizeof(err), "Unexpected token '%.400s' after expression", f->tok.str);
		mkerror(f, err);
	}
	return e;
}

static void
fexpr_free(struct fexpr *e)
{
	size_t i, j;
	if (!e) return;
	for (i = 0; i < e->n; i++) {
		struct basic *fb = e->ops[i].arg;
		if (fb) {
			for (j = 0; j < fb->n; j++) free(fb->data[j].str);
			free(fb->data);
			free(fb->freqmap);
			free(fb);
		}
	}
	free(e->ops);
	free(e);
}

/* ------------------------------------------------------------------ */
/* filter evaluation (filter.c:255-1012)                               */
/* ------------------------------------------------------------------ */
static int
match_addr(struct basic *fb, struct xo_flow *flow, int nbytes)  /* :255-466 */
{
	const struct ffield *ff = &ffields[fb->ffield];
	const uint8_t *a1 = NULL, *a2 = NULL;
	size_t i;

	if (fb->direction == DIR_SRC) {
		if (!FHAS(flow, ff->src)) return 0;
		a1 = FBYTES(flow, ff->src);
	} else if (fb->direction == DIR_DST) {
		if (!FHAS(flow, ff->dst)) return 0;
		a1 = FBYTES(flow, ff->dst);
	} else if (fb->direction == DIR_BOTH) {
		if (FHAS(flow, ff->src)) a1 = FBYTES(flow, ff->src);
		if (FHAS(flow, ff->dst)) a2 = FBYTES(flow, ff->dst);
	} else {
		return 0;
	}
	if (!a1 && !a2) return 0;

	for (i = 0; i < fb->n; i++) {
		struct bdata *d = &fb->data[i];
		const uint8_t *cand[2] = { a1, a2 };
		int c;
		for (c = 0; c < 2; c++) {
			const uint8_t *a = cand[c];
			int k, eq = 1;
			if (!a) continue;
			if (d->is_list) {
				struct iplist *l = &iplists[d->list_idx];
				if (trie_match((nbytes == 4) ? l->nodes4 : l->nodes6, a, nbytes * 8)) {
					return 1;
				}
				continue;
			}
			for (k = 0; k < nbytes; k++) {
				if ((a[k] & d->mask[k]) != d->addr[k]) { eq = 0; break; }
			}
			if (eq) return 1;
		}
	}
	return 0;
}

/* filter.c:468-550: the first SIZEMAX bytes of the field, big-endian, into
 * an int; no presence check */
static int
range_val(struct xo_flow *flow, int fi)
{
	const uint8_t *p = FBYTES(flow, fi);
	switch (fields[fi].mx) {
		case 1: return p[0];
		case 2: return (p[0] << 8) | p[1];
		case 4: return (int)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16)
			| ((uint32_t)p[2] << 8) | p[3]);
		case 8: return 0;  /* be64toh of a 32-bit load keeps only zero low bits (:487) */
		default: return 0; /* uninitialised in the reference; no such RANGE field exists */
	}
}

static int
in_ranges(struct basic *fb, int64_t v)
{
	size_t i;
	for (i = 0; i < fb->n; i++) {
		if ((v >= fb->data[i].low) && (v <= fb->data[i].high)) return 1;
	}
	return 0;
}

static int
match_range(struct basic *fb, struct xo_flow *flow)
{
	const struct ffield *ff = &ffields[fb->ffield];
	if (fb->direction == DIR_SRC) return in_ranges(fb, range_val(flow, ff->src));
	if (fb->direction == DIR_DST) return in_ranges(fb, range_val(flow, ff->dst));
	if (fb->direction == DIR_BOTH) {
		size_t i;
		int r1 = range_val(flow, ff->src), r2 = range_val(flow, ff->dst);
		for (i = 0; i < fb->n; i++) {
			if ((r1 >= fb->data[i].low) && (r1 <= fb->data[i].high)) return 1;
			if ((r2 >= fb->data[i].low) && (r2 <= fb->data[i].high)) return 1;
		}
	}
	return 0;
}

static int
match_mac(struct basic *fb, struct xo_flow *flow)  /* filter.c:552-606 */
{
	const struct ffield *ff = &ffields[fb->ffield];
	size_t i;
	for (i = 0; i < fb->n; i++) {
		if ((fb->direction & DIR_SRC) || (fb->direction == DIR_DST)) {
			int f1 = (fb->direction == DIR_DST) ? ff->dst : ff->src;
			if (memcmp(fb->data[i].mac, FBYTES(flow, f1), 6) == 0) return 1;
		}
		if (fb->direction == DIR_BOTH) {
			if (memcmp(fb->data[i].mac, FBYTES(flow, ff->dst), 6) == 0) return 1;
		}
	}
	return 0;
}

static uint64_t
get_nf_val(const uint8_t *p, unsigned int size)  /* filter.h:370-394 */
{
	uint64_t v = 0;
	unsigned int i;
	if ((size != 8) && (size != 4) && (size != 2) && (size != 1)) return 0;
	for (i = 0; i < size; i++) v = (v << 8) | p[i];
	return v;
}

static int
xdiv(uint64_t dividend, uint64_t divisor, int is_log, int k)  /* filter.h:396-418 */
{
	int quotient;
	if (divisor == 0) return 0;
	quotient = (int)(dividend / divisor);
	if (is_log) {
		int lg = (int)(log(quotient) / log(k));
		quotient = (int)pow(k, lg);
	} else if (k > 1) {
		quotient /= k;
		quotient *= k;
	}
	return quotient;
}

int
xo_xdiv(uint64_t dividend, uint64_t divisor, int is_log, int k)
{
	return xdiv(dividend, divisor, is_log, k);
}

/* the text of a TCP flags byte: the names of the set bits, most significant
 * first, joined with '+' (table utils-data.inc:4-261, reached through
 * tcp_flags_to_str(), utils.c:22-26) */
void
xo_tcp_flags_str(uint8_t tf, char out[40])
{
	static const char *const names[8] = { "CWR", "ECE", "URG", "ACK", "PSH", "RST", "SYN", "FIN" };
	int b;

	out[0] = '\0';
	for (b = 0; b < 8; b++) {
		if (!(tf & (0x80 >> b))) continue;
		if (out[0]) strcat(out, "+");
		strcat(out, names[b]);
	}
}

static int
match_tfstr(struct basic *fb, struct xo_flow *flow)  /* filter.c:836-858 */
{
	const uint8_t *p = (const uint8_t *)flow + fb->a_off;
	uint8_t tf = (fb->a_size == 2) ? p[1] : p[0];
	char flg[40];
	size_t i;

	xo_tcp_flags_str(tf, flg);
	for (i = 0; i < fb->n; i++) {
		/* an operand that is not a quoted string has no text in the
		 * reference (uninitialised pointer): it never matches here */
		if (fb->data[i].str && (strcasecmp(fb->data[i].str, flg) == 0)) return 1;
	}
	return 0;
}

/* filter.c:608-656.  With a direction the reference compares the operand
 * with the field of that direction and then once more with a pointer and a
 * length it never set (:645): that second comparison has no defined result
 * and counts as "no match" here.  Without a direction only the source field
 * is compared (:649).  strncmp over the field's recorded size: an absent
 * field (size 0) equals every operand. */
static int
match_string(struct basic *fb, struct xo_flow *flow)
{
	const struct ffield *ff;
	int fi;
	size_t i;

	if (fb->ffield < 0) return 0;
	ff = &ffields[fb->ffield];
	if (fb->direction == DIR_SRC) fi = ff->src;
	else if (fb->direction == DIR_DST) fi = ff->dst;
	else if (fb->direction == DIR_BOTH) fi = ff->src;
	else return 0;
	for (i = 0; i < fb->n; i++) {
		/* an operand that is not a quoted string has no text in the reference */
		if (!fb->data[i].str) continue;
		if (strncmp(fb->data[i].str, (const char *)FBYTES(flow, fi), (size_t)FSIZE(flow, fi)) == 0) return 1;
	}
	return 0;
}

/* service names of the ports: the reference's table is generated data of its
 * build (utils-data.inc:263-), so the oracle takes the names from its caller */
static char *port_names[65536];

void
xo_set_port_names(const char *const *names)
{
	int i;
	for (i = 0; i < 65536; i++) {
		free(port_names[i]);
		port_names[i] = (names && names[i] && names[i][0]) ? strdup(names[i]) : NULL;
	}
}

void
xo_port_to_str(char *res, uint16_t port)  /* utils.c:28-37 */
{
	const char *s = port_names[port] ? port_names[port] : "";
	if (s[0]) {
		sprintf(res, "%s (%d)", s, port);
		return;
	}
	sprintf(res, "%d", port);
}

void
xo_ports_pair_to_str(char *res, uint16_t port1, uint16_t port2)  /* utils.c:39-77 */
{
	const char *s1 = port_names[port1] ? port_names[port1] : "";
	const char *s2 = port_names[port2] ? port_names[port2] : "";

	if (s1[0] && s2[0]) { sprintf(res, "%s(%d) -> %s(%d)", s1, port1, s2, port2); return; }
	if (port1 == port2) { sprintf(res, "%d -> %d", port1, port2); return; }
	if (port1 < port2) {
		if (!s1[0] && !s2[0]) { sprintf(res, "%d ->", port1); return; }
		else if (!s1[0] && s2[0]) { sprintf(res, "%d -> %s(%d)", port1, s2, port2); return; }
		else if (s1[0] && !s2[0]) { sprintf(res, "%s(%d) ->", s1, port1); return; }
	} else {
		if (!s1[0] && !s2[0]) { sprintf(res, "-> %d", port2); return; }
		else if (!s1[0] && s2[0]) { sprintf(res, "-> %s(%d)", s2, port2); }
		else if (s1[0] && !s2[0]) { sprintf(res, "%s(%d) -> %d", s1, port1, port2); return; }
	}
}

static int
match_portstr(struct basic *fb, struct xo_flow *flow)  /* filter.c:860-880 */
{
	uint16_t port = (uint16_t)get_nf_val((uint8_t *)flow + fb->a_off, fb->a_size);
	char p[300];
	size_t i;

	xo_port_to_str(p, port);
	for (i = 0; i < fb->n; i++) {
		if (fb->data[i].str && (strcasecmp(fb->data[i].str, p) == 0)) return 1;
	}
	return 0;
}

static int
match_ppstr(struct basic *fb, struct xo_flow *flow)  /* filter.c:882-904 */
{
	uint16_t port1 = (uint16_t)get_nf_val((uint8_t *)flow + fb->a_off, fb->a_size);
	uint16_t port2 = (uint16_t)get_nf_val((uint8_t *)flow + fb->b_off, fb->b_size);
	char pp[600];
	size_t i;

	xo_ports_pair_to_str(pp, port1, port2);
	for (i = 0; i < fb->n; i++) {
		if (fb->data[i].str && (strcasecmp(fb->data[i].str, pp) == 0)) return 1;
	}
	return 0;
}

/* filter.c:711-751, with its update of the map: arg1's count goes up twice,
 * arg2's never (:736-739) */
static int
match_mfreq(struct basic *fb, struct xo_flow *flow)
{
	uint16_t arg1 = (uint16_t)get_nf_val((uint8_t *)flow + fb->a_off, fb->a_size);
	uint16_t arg2 = (uint16_t)get_nf_val((uint8_t *)flow + fb->b_off, fb->b_size);
	uint64_t freq1 = fb->freqmap[arg1], freq2 = fb->freqmap[arg2];
	int64_t res;

	if (freq1 != freq2) res = (freq1 > freq2) ? arg1 : arg2;
	else res = (arg1 < arg2) ? arg1 : arg2;
	fb->freqmap[arg1] += 2;
	return in_ranges(fb, res);
}

/* ---- geo / AS databases (geoip.h:13-55, geoip.c:33-88, ip-btrie.h:88-115) ----
 * A database is the array of trie nodes the reference maps from geo4.db /
 * geo6.db / as4.db / as6.db: {uint32_t next[2]; int is_leaf; payload}, payload =
 * struct geoip_info (8 char arrays of 3, 3, 35, 64, 51, 15, 20, 23 bytes) or
 * struct as_info {uint32_t asn; char asd[200]}.  The lookup walks the address
 * bits from the most significant one while a child exists and answers with
 * the payload of the node it stops at if that node is a leaf. */
#define XO_GEO_NODE 228   /* 12 + 214, padded to the alignment of its ints */
#define XO_AS_NODE 216    /* 12 + 4 + 200 */
static const int geo_off[8] = { 0, 3, 6, 41, 105, 156, 171, 191 };
static struct { uint8_t *img; size_t nnodes; } geodb[2][2];   /* [0 geo, 1 as][0 IPv4, 1 IPv6] */

int
xo_geodb_set(int kind, int family, const void *img, size_t bytes)
{
	const size_t ns = kind ? XO_AS_NODE : XO_GEO_NODE;
	const int fi = (family == 6) ? 1 : 0;

	if ((kind < 0) || (kind > 1) || (bytes % ns)) return 0;
	free(geodb[kind][fi].img);
	geodb[kind][fi].img = NULL;
	geodb[kind][fi].nnodes = 0;
	if (!bytes) return 1;
	geodb[kind][fi].img = malloc(bytes);
	if (!geodb[kind][fi].img) return 0;
	memcpy(geodb[kind][fi].img, img, bytes);
	geodb[kind][fi].nnodes = bytes / ns;
	return 1;
}

/* IP_BTRIE_LOOKUP: the payload of the deepest node on the address's path, NULL
 * when there is no database or that node is not a leaf */
static const uint8_t *
geodb_lookup(int kind, const uint8_t *addr, int nbits)
{
	const int fi = (nbits == 128) ? 1 : 0;
	const size_t ns = kind ? XO_AS_NODE : XO_GEO_NODE;
	const uint8_t *db = geodb[kind][fi].img;
	uint32_t node = 0, next;
	int i, leaf;

	if (!db) return NULL;
	for (i = 0; i < nbits; i++) {
		const int bit = !!(addr[i / 8] & (1 << (7 - (i % 8))));
		memcpy(&next, db + (size_t)node * ns + 4 * bit, 4);
		if (!next || (next >= geodb[kind][fi].nnodes)) break;
		node = next;
	}
	memcpy(&leaf, db + (size_t)node * ns + 8, 4);
	return leaf ? db + (size_t)node * ns + 12 : NULL;
}

/* the lookup alone, for tests: copies the payload of the node the walk ends on
 * (214 bytes of struct geoip_info / 204 bytes of struct as_info) and returns 1,
 * or returns 0 as geoip_lookup4/6 / as_lookup4/6 do (geoip.c:33-88) */
int
xo_geodb_lookup(int kind, int family, const uint8_t *addr, uint8_t *payload)
{
	const uint8_t *pay;
	if ((kind < 0) || (kind > 1)) return 0;
	pay = geodb_lookup(kind, addr, (family == 6) ? 128 : 32);
	if (!pay) return 0;
	memcpy(payload, pay, kind ? 204 : 214);
	return 1;
}

/* filter_function_geoip() / filter_function_as(), filter.c:753-834: the field's
 * bytes are read whether the field is present or not; only fields of 4 and 16
 * bytes have a defined meaning */
static int
match_geo_as(struct basic *fb, struct xo_flow *flow)
{
	const uint8_t *addr = (const uint8_t *)flow + fb->a_off;
	const uint8_t *pay;
	size_t i;

	if ((fb->a_size != 4) && (fb->a_size != 16)) return 0;
	pay = geodb_lookup(fb->func != FN_GEO, addr, (fb->a_size == 4) ? 32 : 128);
	if (!pay) return 0;
	if (fb->func == FN_ASN) {
		uint32_t asn;
		memcpy(&asn, pay, 4);
		return in_ranges(fb, (int32_t)asn);
	}
	{
		const char *res = (fb->func == FN_ASD) ? (const char *)pay + 4 : (const char *)pay + geo_off[fb->geo_field];
		for (i = 0; i < fb->n; i++) {
			if (fb->data[i].str && (strcmp(fb->data[i].str, res) == 0)) return 1;
		}
	}
	return 0;
}

static int
basic_match(struct basic *fb, struct xo_flow *flow)  /* filter.c:907-958 */
{
	if (fb->func == FN_TFSTR) return match_tfstr(fb, flow);
	if (fb->func == FN_PORTSTR) return match_portstr(fb, flow);
	if (fb->func == FN_PPSTR) return match_ppstr(fb, flow);
	if (fb->func == FN_MFREQ) return match_mfreq(fb, flow);
	/* geo / AS functions: geoip_lookup4/6 and as_lookup4/6 answer 0 while no
	 * database is loaded (geoip.c:33-88) */
	if ((fb->func == FN_GEO) || (fb->func == FN_ASN) || (fb->func == FN_ASD)) return match_geo_as(fb, flow);
	if (fb->func == FN_MIN) {  /* filter.c:685-709 */
		uint64_t a = get_nf_val((uint8_t *)flow + fb->a_off, fb->a_size);
		uint64_t b = get_nf_val((uint8_t *)flow + fb->b_off, fb->b_size);
		return in_ranges(fb, (int64_t)((a < b) ? a : b));
	}
	if (fb->func != FN_NONE) {  /* filter.c:659-683 */
		uint64_t a = get_nf_val((uint8_t *)flow + fb->a_off, fb->a_size);
		uint64_t b = get_nf_val((uint8_t *)flow + fb->b_off, fb->b_size);
		return in_ranges(fb, xdiv(a, b, fb->is_log, fb->k));
	}
	switch (fb->type) {
		case BT_ADDR4: return match_addr(fb, flow, 4);
		case BT_ADDR6: return match_addr(fb, flow, 16);
		case BT_RANGE: return match_range(fb, flow);
		case BT_MAC: return match_mac(fb, flow);
		case BT_STRING: return match_string(fb, flow);
		default: return 0;
	}
}

static int
filter_match(struct fexpr *expr, struct xo_flow *flow)  /* filter.c:960-1012 */
{
	size_t i, sp = 0;
	int stack[256];

	if (expr->n == 0) return 1;
	if (expr->n > 256) return 0;
	for (i = 0; i < expr->n; i++) {
		struct fop *op = &expr->ops[i];
		switch (op->op) {
			case OP_BASIC:
				stack[sp++] = basic_match(op->arg, flow);
				break;
			case OP_NOT:
				if (sp < 1) return 0;
				stack[sp - 1] = !stack[sp - 1];
				break;
			case OP_AND:
				if (sp < 2) return 0;
				stack[sp - 2] &= stack[sp - 1];
				sp--;
				break;
			case OP_OR:
				if (sp < 2) return 0;
				stack[sp - 2] |= stack[sp - 1];
				sp--;
				break;
			default:
				return 0;
		}
	}
	return (sp != 1) ? 0 : stack[0];
}

/* filter.c:1069-1256 */
static void
filter_dump(struct fexpr *e, FILE *f)
{
	size_t i, j;
	int k;
	for (i = 0; i < e->n; i++) {
		struct basic *fb = e->ops[i].arg;
		switch (e->ops[i].op) {
			case OP_NOT: fprintf(f, "NOT\n"); continue;
			case OP_AND: fprintf(f, "AND\n"); continue;
			case OP_OR: fprintf(f, "OR\n"); continue;
			default: break;
		}
		if (fb->direction == DIR_SRC) fprintf(f, "SRC ");
		else if (fb->direction == DIR_DST) fprintf(f, "DST ");
		else if (fb->direction == DIR_BOTH) fprintf(f, "SRC OR DST ");

		if (fb->func == FN_DIV) {
			fprintf(f, "DIV ([offset %d]/[offset %d])", (int)fb->a_off, (int)fb->b_off);
		} else if (fb->func == FN_DIV_R) {
			fprintf(f, "DIV_R ([offset %d]/[offset %d]/k %d)", (int)fb->a_off, (int)fb->b_off, fb->k);
		} else if (fb->func == FN_DIV_L) {
			fprintf(f, "DIV_L ([offset %d]/[offset %d]/k %d)", (int)fb->a_off, (int)fb->b_off, fb->k);
		} else if (fb->func == FN_MIN) {
			fprintf(f, "MIN ([offset %d]/[offset %d])", (int)fb->a_off, (int)fb->b_off);
		} else if (fb->func == FN_MFREQ) {
			fprintf(f, "MFREQ ([offset %d]/[offset %d])", (int)fb->a_off, (int)fb->b_off);
		} else if (fb->func == FN_GEO) {
			fprintf(f, "%s ([offset %d]/[size %d])", geoip_kw[fb->geo_field], (int)fb->a_off, (int)fb->a_size);
		} else if ((fb->func == FN_ASN) || (fb->func == FN_ASD)) {
			fprintf(f, "AS ([offset %d]/[size %d])", (int)fb->a_off, (int)fb->a_size);
		} else if ((fb->func == FN_TFSTR) || (fb->func == FN_PORTSTR) || (fb->func == FN_PPSTR)) {
			/* no case for them in filter.c:1113-1172: the default branch prints
			 * the enum value (filter.h:130-151: 36 fields, then the functions) */
			fprintf(f, "<Unknown name %d> ", 1 + N_FFIELDS + 5 + 8 + 2 + (fb->func - FN_TFSTR));
		} else {
			/* fprintf(f, #STR" ") stringifies the quoted keyword (filter.c:1117) */
			fprintf(f, "\"%s\" ", ffields[fb->ffield].kw);
		}
		for (j = 0; j < fb->n; j++) {
			struct bdata *d = &fb->data[j];
			if ((fb->type == BT_ADDR4) || (fb->type == BT_ADDR6)) {
				int nb = (fb->type == BT_ADDR4) ? 4 : 16;
				if (d->is_list) {
					fprintf(f, " LIST '%s'", iplists[d->list_idx].name);
					continue;
				}
				fprintf(f, " ");
				for (k = 0; k < nb; k++) {
					fprintf(f, "%u", d->addr[k]);
					if (k != nb - 1) fprintf(f, (nb == 4) ? "." : ":");
				}
				fprintf(f, "/%d", d->mask_len);
			} else if (fb->type == BT_RANGE) {
				fprintf(f, " %d-%d", d->low, d->high);
			} else if (fb->type == BT_MAC) {
				fprintf(f, " %02x:%02x:%02x:%02x:%02x:%02x", d->mac[0], d->mac[1],
					d->mac[2], d->mac[3], d->mac[4], d->mac[5]);
			} else if (fb->type == BT_STRING) {
				fprintf(f, " %s", d->str ? d->str : "(null)");
			}
		}
		fprintf(f, "\n");
	}
}

/* handles for the ctypes-facing API */
static struct fexpr **fhandles = NULL;
static size_t n_fhandles = 0;

static int
fhandle_new(struct fexpr *e)
{
	struct fexpr **tmp = realloc(fhandles, sizeof(*tmp) * (n_fhandles + 1));
	if (!tmp) return -1;
	fhandles = tmp;
	fhandles[n_fhandles] = e;
	return (int)n_fhandles++;
}

static struct fexpr *
compile(const char *text, char *err, size_t errlen)
{
	struct finput in;
	struct fexpr *e;

	memset(&in, 0, sizeof(in));
	in.s = text;
	e = parse_filter(&in);
	if (!e) {
		if (err && errlen) snprintf(err, errlen, "allocation failed");
		return NULL;
	}
	if (in.error) {
		if (err && errlen) snprintf(err, errlen, "%s", in.errmsg);
		fexpr_free(e);
		return NULL;
	}
	return e;
}

int
xo_filter_parse(const char *text, char *err, size_t errlen)
{
	struct fexpr *e = compile(text, err, errlen);
	if (!e) return -1;
	return fhandle_new(e);
}

int
xo_filter_dump(int h, char *out, size_t outlen)
{
	FILE *f;
	if ((h < 0) || ((size_t)h >= n_fhandles) || !outlen) return -1;
	memset(out, 0, outlen);
	f = fmemopen(out, outlen - 1, "w");
	if (!f) return -1;
	filter_dump(fhandles[h], f);
	fclose(f);
	return (int)strlen(out);
}

int
xo_filter_match(int h, const void *flow)
{
	if ((h < 0) || ((size_t)h >= n_fhandles)) return -1;
	return filter_match(fhandles[h], (struct xo_flow *)flow);
}

void
xo_filter_free_all(void)
{
	size_t i;
	for (i = 0; i < n_fhandles; i++) fexpr_free(fhandles[i]);
	free(fhandles);
	fhandles = NULL;
	n_fhandles = 0;
}

/* ------------------------------------------------------------------ */
/* devices (devices.h:9-25, devices.c:199-314)                         */
/* ------------------------------------------------------------------ */
struct device
{
	int use_ip;
	uint32_t ip;       /* network order; ip_ver is always 4 (devices.c:86) */
	int use_id;
	uint32_t id;       /* htonl(atoi()) (devices.c:100) */
	size_t n_exprs;
	struct fexpr **exprs;
	int skip_unmarked;
	int sampling_rate;
};

static struct device *devices = NULL;
static size_t n_devices = 0;

int
xo_device_add(int use_ip, uint32_t ip, int use_id, uint32_t id_host, int sampling_rate,
	int skip_unmarked, const char *const *marks, int nmarks, char *err, size_t errlen)
{
	struct device *tmp = realloc(devices, sizeof(*tmp) * (n_devices + 1));
	struct device *d;
	int i;

	if (!tmp) return 0;
	devices = tmp;
	d = &devices[n_devices];
	memset(d, 0, sizeof(*d));
	d->use_ip = use_ip;
	d->ip = ip;
	d->use_id = use_id;
	d->id = htonl(id_host);
	d->sampling_rate = sampling_rate;
	d->skip_unmarked = skip_unmarked;
	for (i = 0; i < nmarks; i++) {
		struct fexpr *e = compile(marks[i], err, errlen);
		struct fexpr **t2;
		if (!e) return 0;
		t2 = realloc(d->exprs, sizeof(*t2) * (d->n_exprs + 1));
		if (!t2) return 0;
		d->exprs = t2;
		d->exprs[d->n_exprs++] = e;
	}
	n_devices++;
	return 1;
}

/* the lookup shared by devices.c:205-232 and :246-270.  Note `} if` at
 * :224/:263 is not `else if`: an entry with both ip and id whose first test
 * failed is tested again on the id alone. */
static struct device *
device_find(uint32_t ip, uint32_t id)
{
	size_t i;
	for (i = 0; i < n_devices; i++) {
		struct device *db = &devices[i];
		if (db->use_ip && db->use_id) {
			if ((ip == db->ip) && (id == db->id)) return db;
		} else if (db->use_ip) {
			if (ip == db->ip) return db;
		}
		if (db->use_id) {
			if (id == db->id) return db;
		}
	}
	return NULL;
}

static void
sampling_rate_init(struct pktinfo *fpi)  /* netflow.c:126-144 */
{
	struct device *db = device_find(fpi->src_addr_ipv4, fpi->source_id);
	fpi->sampling_rate = db ? db->sampling_rate : 1;
}

/* devices.c:237-314 */
static int
device_rules_check(struct xo_flow *flow, struct pktinfo *fpi)
{
	struct device *db = device_find(fpi->src_addr_ipv4, fpi->source_id);
	uint32_t mark = 0, mark_be;
	size_t i;

	if (!db) return 1;
	for (i = 0; i < db->n_exprs; i++) {
		if (filter_match(db->exprs[i], flow)) mark++;
	}
	if (db->skip_unmarked && (mark == 0)) return 0;
	mark_be = htobe32(mark);
	memcpy(flow->dev_mark, &mark_be, 4);
	flow->dev_mark_size = 4;
	flow->has_dev_mark = 1;
	return 1;
}

/* ------------------------------------------------------------------ */
/* monitoring objects and the record log                               */
/* ------------------------------------------------------------------ */
struct mo
{
	struct fexpr *expr;
	int parent;
	int *kids;
	size_t n_kids;
};

static struct mo *mos = NULL;
static size_t n_mos = 0;
static int *top_mos = NULL;
static size_t n_top = 0;

int
xo_mo_add(int parent, const char *filter, char *err, size_t errlen)
{
	struct fexpr *e;
	struct mo *tmp;
	int idx, **list;
	size_t *cnt;
	int *l2;

	if ((parent >= (int)n_mos) || (n_mos >= XO_MAX_MO)) return -1;
	e = compile(filter, err, errlen);
	if (!e) return -1;
	tmp = realloc(mos, sizeof(*tmp) * (n_mos + 1));
	if (!tmp) return -1;
	mos = tmp;
	idx = (int)n_mos++;
	memset(&mos[idx], 0, sizeof(struct mo));
	mos[idx].expr = e;
	mos[idx].parent = parent;
	if (parent < 0) { list = &top_mos; cnt = &n_top; }
	else { list = &mos[parent].kids; cnt = &mos[parent].n_kids; }
	l2 = realloc(*list, sizeof(int) * (*cnt + 1));
	if (!l2) return -1;
	*list = l2;
	(*list)[(*cnt)++] = idx;
	return idx;
}

int xo_nmo(void) { return (int)n_mos; }

static int log_keep_flows = 0;
static size_t log_max = 0, log_n = 0, log_cap = 0;
static uint64_t *log_masks = NULL;
static uint8_t *log_dropped = NULL;
static uint8_t *log_flows = NULL;
static uint64_t *counters = NULL;
static size_t counters_n = 0;

size_t xo_log_mask_words(void) { return (n_mos + 63) / 64 ? (n_mos + 63) / 64 : 1; }

void
xo_log_config(int keep_flows, size_t max_records)
{
	log_keep_flows = keep_flows;
	log_max = max_records;
}

void
xo_log_clear(void)
{
	free(log_masks); free(log_dropped); free(log_flows); free(counters);
	log_masks = NULL; log_dropped = NULL; log_flows = NULL; counters = NULL;
	log_n = log_cap = counters_n = 0;
}

size_t xo_log_nrecords(void) { return log_n; }
const uint64_t *xo_log_mask(size_t r) { return &log_masks[r * xo_log_mask_words()]; }
int xo_log_dropped(size_t r) { return log_dropped[r]; }
const void *xo_log_flow(size_t r) { return log_flows ? log_flows + r * XO_FLOW_INFO_SIZE : NULL; }

const uint64_t *
xo_counters(void)
{
	if (counters_n != n_mos) {
		free(counters);
		counters = calloc(n_mos ? n_mos : 1, 3 * sizeof(uint64_t));
		counters_n = n_mos;
	}
	return counters;
}

/* ------------------------------------------------------------------ */
/* fixed-window tables (monit-objects.c:923-1023, filter-parser.c:348-484) */
/* ------------------------------------------------------------------ */
struct ofield   /* struct field, monit-objects.h; without mfreq and the database functions */
{
	int aggr, descending;
	size_t off, size;
	int scale;
	/* function forms (field.func_data): which, arguments, div's is_log and k */
	int func;
	unsigned int a_off, a_size, b_off, b_size;
	int is_log, k;
};

struct ofwm
{
	int mo;
	struct ofield naggr[16], aggr[8];
	size_t n_naggr, n_aggr, keysize;
	struct ofield order[24];     /* all fields in the order given (mo_fwm.fieldset) */
	size_t n_order;
	/* keyed sums: open addressing over `cap` rows of keysize bytes + n_aggr u64 */
	uint8_t *keys;
	uint64_t *vals;
	uint8_t *used;
	size_t cap, n;
};

static struct ofwm *fwms = NULL;
static size_t n_fwms = 0;

/* parse_field(), filter-parser.c:396-484, without mfreq and the database functions */
static int
parse_field(const char *text, struct ofield *fld, char *err, size_t errlen)
{
	struct finput in;
	int i, dir = DIR_BOTH, found = 0;

	memset(&in, 0, sizeof(in));
	memset(fld, 0, sizeof(*fld));
	in.s = text;
	read_token(&in);
	if (in.error) { snprintf(err, errlen, "%s", in.errmsg); return 0; }
	if (in.end) { snprintf(err, errlen, "Empty field not allowed"); return 0; }
	switch (in.tok.id) {
		case TK_MFREQ: case TK_GEOIP: case TK_ASN: case TK_ASD:
			snprintf(err, errlen, "mfreq, geo and AS function fields are not restated by the oracle");
			return 0;
		default:
			break;
	}
	/* function_div_parse(), function_min_parse(), function_tfstr_parse(),
	 * function_portstr_parse(), function_ppstr_parse() (filter-parser-funcs.c:83-282,
	 * :526-705) as parse_field() tries them (filter-parser.c:413-442) */
	if ((in.tok.id == TK_DIV) || (in.tok.id == TK_DIV_R) || (in.tok.id == TK_DIV_L)) {
		const int tok = in.tok.id;
		(void)accept_(&in, tok);
		fld->func = (tok == TK_DIV) ? FN_DIV : (tok == TK_DIV_R) ? FN_DIV_R : FN_DIV_L;
		fld->is_log = tok == TK_DIV_L;
		fld->k = 1;
		fld->size = sizeof(uint64_t);
		if (!accept_(&in, TK_LPAREN)) { snprintf(err, errlen, "Expected '(' after 'div'"); return 0; }
		if (!aggr_field_off_size(&in, &fld->a_off, &fld->a_size)) { snprintf(err, errlen, "Expected aggregable field name"); return 0; }
		if (!accept_(&in, TK_COMMA)) { snprintf(err, errlen, "Expected ',' after field name"); return 0; }
		if (!aggr_field_off_size(&in, &fld->b_off, &fld->b_size)) { snprintf(err, errlen, "Expected aggregable field name after comma"); return 0; }
		if (tok != TK_DIV) {
			int k;
			if (!accept_(&in, TK_COMMA)) { snprintf(err, errlen, "Expected ',' after second field name"); return 0; }
			k = in.tok.low;
			if (!accept_(&in, TK_INT_RANGE)) { snprintf(err, errlen, "Expected INT"); return 0; }
			fld->k = k;
		}
		if (!accept_(&in, TK_RPAREN)) { snprintf(err, errlen, "Expected ')'"); return 0; }
		found = 2;
	} else if ((in.tok.id == TK_MIN) || (in.tok.id == TK_TFSTR) || (in.tok.id == TK_PORTSTR) || (in.tok.id == TK_PPSTR)) {
		const int tok = in.tok.id;
		const int two = (tok == TK_MIN) || (tok == TK_PPSTR);
		(void)accept_(&in, tok);
		fld->func = (tok == TK_MIN) ? FN_MIN : (tok == TK_TFSTR) ? FN_TFSTR : (tok == TK_PORTSTR) ? FN_PORTSTR : FN_PPSTR;
		/* utils.h:13-15 */
		fld->size = (tok == TK_MIN) ? sizeof(uint64_t) : (tok == TK_TFSTR) ? 32 : (tok == TK_PORTSTR) ? 40 : 90;
		if (!accept_(&in, TK_LPAREN)) { snprintf(err, errlen, "Expected '(' after function name"); return 0; }
		if (!nonaggr_field(&in, &fld->a_off, &fld->a_size)) { snprintf(err, errlen, "Incorrect field name"); return 0; }
		if (two) {
			if (!accept_(&in, TK_COMMA)) { snprintf(err, errlen, "Expected ',' after field name"); return 0; }
			if (!nonaggr_field(&in, &fld->b_off, &fld->b_size)) { snprintf(err, errlen, "Incorrect field name after comma"); return 0; }
		}
		if (!accept_(&in, TK_RPAREN)) { snprintf(err, errlen, "Expected ')'"); return 0; }
		found = 2;
	}
	if (found == 2) fld->off = fld->a_off;
	if (found != 2)
	for (i = 0; i < N_AGFIELDS; i++) {
		if (accept_(&in, TK_AGFIELD0 + i)) {
			fld->aggr = 1;
			fld->size = 8;
			fld->off = fields[agfields[i].field].off;
			fld->scale = agfields[i].scale;
			found = 1;
			break;
		}
	}
	if (!found) {
		if (accept_(&in, TK_SRC)) dir = DIR_SRC;
		else if (accept_(&in, TK_DST)) dir = DIR_DST;
		for (i = 0; i < N_FFIELDS; i++) {
			if (accept_(&in, TK_FIELD0 + i)) {
				int fi;
				if (dir == DIR_BOTH) { /* (field_without_direction(), filter-parser.c:348-387) */
					if (ffields[i].src != ffields[i].dst) {
						snprintf(err, errlen, "This field requires direction");
						return 0;
					}
					fi = ffields[i].src;
				} else {
					fi = (dir == DIR_SRC) ? ffields[i].src : ffields[i].dst;
				}
				/* field_map_flow_info() knows the netflow.def fields only */
				if (fi >= F_dev_ip6) { snprintf(err, errlen, "Unknown field"); return 0; }
				fld->off = fields[fi].off;
				fld->size = fields[fi].mx;
				found = 1;
				break;
			}
		}
		if (!found) { snprintf(err, errlen, "Unknown field '%s'", in.tok.str); return 0; }
	}
	if (in.error) { snprintf(err, errlen, "%s", in.errmsg); return 0; }
	if (accept_(&in, TK_DESC)) fld->descending = 1;
	else (void)accept_(&in, TK_ASC);
	if (in.error) { snprintf(err, errlen, "%s", in.errmsg); return 0; }
	if (!in.end) { snprintf(err, errlen, "Extra symbols after field name"); return 0; }
	return 1;
}

/* describes a field text: returns 1 and {aggr, offset, size, scale, descending, function, offset of
 * its second argument, k, sizes of the arguments as a * 256 + b} or 0 + err */
int
xo_field_parse(const char *text, int out[9], char *err, size_t errlen)
{
	static const int fn_code[] = { 0, 1, 2, 3, 4, 5, 6, 7 };   /* FN_DIV .. FN_PPSTR as the product numbers them */
	struct ofield f;
	if (!parse_field(text, &f, err, errlen)) return 0;
	out[0] = f.aggr; out[1] = (int)f.off; out[2] = (int)f.size; out[3] = f.scale; out[4] = f.descending;
	out[5] = f.func ? fn_code[f.func] : 0;
	out[6] = f.func ? (int)f.b_off : 0;
	out[7] = f.func ? f.k : 0;
	out[8] = f.func ? (int)(f.a_size * 256 + f.b_size) : 0;
	return 1;
}

/* one "fwm" entry of an object (fwm_fields_init(), monit-objects-fwm.c:28-100):
 * the fields in the order given, split into key fields and aggregable ones */
int
xo_fwm_add(int mo, const char *const *field_texts, int n, char *err, size_t errlen)
{
	struct ofwm w, *tmp;
	int i;

	memset(&w, 0, sizeof(w));
	w.mo = mo;
	for (i = 0; i < n; i++) {
		struct ofield f;
		if (!parse_field(field_texts[i], &f, err, errlen)) return -1;
		if (w.n_order < 24) w.order[w.n_order++] = f;
		if (f.aggr) {
			if (w.n_aggr == 8) { snprintf(err, errlen, "too many fields"); return -1; }
			w.aggr[w.n_aggr++] = f;
		} else {
			if (w.n_naggr == 16) { snprintf(err, errlen, "too many fields"); return -1; }
			w.naggr[w.n_naggr++] = f;
			w.keysize += f.size;
		}
	}
	tmp = realloc(fwms, (n_fwms + 1) * sizeof(*tmp));
	if (!tmp) return -1;
	fwms = tmp;
	fwms[n_fwms] = w;
	return (int)n_fwms++;
}

static void
fwm_grow(struct ofwm *w)
{
	size_t ncap = w->cap ? w->cap * 2 : 1024, i, ks = w->keysize ? w->keysize : 1, na = w->n_aggr ? w->n_aggr : 1;
	uint8_t *nk = calloc(ncap, ks), *nu = calloc(ncap, 1);
	uint64_t *nv = calloc(ncap * na, sizeof(uint64_t));
	for (i = 0; i < w->cap; i++) {
		size_t h = 1469598103934665603ull, j, k;
		if (!w->used[i]) continue;
		for (k = 0; k < w->keysize; k++) h = (h ^ w->keys[i * ks + k]) * 1099511628211ull;
		for (j = h & (ncap - 1); nu[j]; j = (j + 1) & (ncap - 1)) {}
		nu[j] = 1;
		memcpy(nk + j * ks, w->keys + i * ks, ks);
		memcpy(nv + j * na, w->vals + i * na, na * sizeof(uint64_t));
	}
	free(w->keys); free(w->vals); free(w->used);
	w->keys = nk; w->vals = nv; w->used = nu; w->cap = ncap;
}

/* monit_object_key_add_fld(), monit-objects.c:878-921, with the functions it
 * calls (:679-717, :827-876) */
static void
key_add_fld(const struct ofield *fld, uint8_t *key, struct xo_flow *flow)
{
	const uint8_t *fl = (const uint8_t *)flow;
	uint64_t a, b, res;
	int i;

	switch (fld->func) {
		case FN_DIV: case FN_DIV_R: case FN_DIV_L:
			a = get_nf_val(fl + fld->a_off, fld->a_size);
			b = get_nf_val(fl + fld->b_off, fld->b_size);
			/* htobe64(int): the quotient is sign-extended first */
			res = b ? (uint64_t)(int64_t)xdiv(a, b, fld->is_log, fld->k) : 0;
			for (i = 0; i < 8; i++) key[i] = (uint8_t)(res >> (56 - 8 * i));
			break;
		case FN_MIN:
			a = get_nf_val(fl + fld->a_off, fld->a_size);
			b = get_nf_val(fl + fld->b_off, fld->b_size);
			res = (a < b) ? a : b;
			for (i = 0; i < 8; i++) key[i] = (uint8_t)(res >> (56 - 8 * i));
			break;
		case FN_TFSTR: {
			char txt[40];
			memset(key, 0, 32);
			xo_tcp_flags_str((fld->a_size == 2) ? fl[fld->a_off + 1] : (uint8_t)get_nf_val(fl + fld->a_off, 1), txt);
			strcpy((char *)key, txt);
			break;
		}
		case FN_PORTSTR: {
			char txt[300];
			memset(key, 0, 40);
			xo_port_to_str(txt, (uint16_t)get_nf_val(fl + fld->a_off, fld->a_size));
			memcpy(key, txt, (strlen(txt) < 39) ? strlen(txt) : 39);
			break;
		}
		case FN_PPSTR: {
			char txt[600];
			memset(key, 0, 90);
			txt[0] = '\0';
			xo_ports_pair_to_str(txt, (uint16_t)get_nf_val(fl + fld->a_off, fld->a_size),
				(uint16_t)get_nf_val(fl + fld->b_off, fld->b_size));
			memcpy(key, txt, (strlen(txt) < 89) ? strlen(txt) : 89);
			break;
		}
		default:
			memcpy(key, fl + fld->off, fld->size);
			break;
	}
}

/* the fixed-window part of monit_object_process_nf(), monit-objects.c:927-1023 */
static void
fwm_update(struct ofwm *w, struct xo_flow *flow)
{
	uint8_t key[16 * 64];
	size_t pos = 0, f, j, h = 1469598103934665603ull, ks = w->keysize ? w->keysize : 1, na = w->n_aggr ? w->n_aggr : 1;

	for (f = 0; f < w->n_naggr; f++) {
		key_add_fld(&w->naggr[f], key + pos, flow);
		pos += w->naggr[f].size;
	}
	if ((w->n + 1) * 2 > w->cap) fwm_grow(w);
	for (j = 0; j < w->keysize; j++) h = (h ^ key[j]) * 1099511628211ull;
	for (j = h & (w->cap - 1); w->used[j]; j = (j + 1) & (w->cap - 1)) {
		if (!memcmp(w->keys + j * ks, key, w->keysize)) break;
	}
	if (!w->used[j]) {
		w->used[j] = 1;
		memcpy(w->keys + j * ks, key, w->keysize);
		w->n++;
	}
	for (f = 0; f < w->n_aggr; f++) {
		uint64_t val = get_nf_val((uint8_t *)flow + w->aggr[f].off, 8);
		w->vals[j * na + f] += val * (uint64_t)w->aggr[f].scale * flow->sampling_rate;
	}
}

size_t xo_fwm_rows(int fwm) { return fwms[fwm].n; }
size_t xo_fwm_keysize(int fwm) { return fwms[fwm].keysize; }
size_t xo_fwm_nvals(int fwm) { return fwms[fwm].n_aggr; }

/* rows of a table in no particular order: keys[row][keysize], vals[row][nvals] */
void
xo_fwm_dump(int fwm, uint8_t *keys, uint64_t *vals)
{
	struct ofwm *w = &fwms[fwm];
	size_t i, r = 0, ks = w->keysize ? w->keysize : 1, na = w->n_aggr ? w->n_aggr : 1;
	for (i = 0; i < w->cap; i++) {
		if (!w->used[i]) continue;
		memcpy(keys + r * w->keysize, w->keys + i * ks, w->keysize);
		memcpy(vals + r * w->n_aggr, w->vals + i * na, w->n_aggr * sizeof(uint64_t));
		r++;
	}
}

/* The export of a window.  fwm_sort_and_dump() (monit-objects-fwm.c:496-589)
 * re-keys every row: the fields in fieldset order, a sum as htobe64(value), a
 * key field as its bytes, both bitwise inverted when the field is `desc`; the
 * rows go into an ordered tkvdb transaction, whose cursor walks keys in
 * ascending byte order.  fwm_dump() (:405-449) writes rows until `limit` is
 * reached (hit_limit is raised then, also when no row is left) and adds up
 * the sums of the remaining rows as the "others" row. */
static const uint8_t *export_sk;
static size_t export_sklen;

static int
export_cmp(const void *a, const void *b)
{
	return memcmp(export_sk + *(const size_t *)a * export_sklen, export_sk + *(const size_t *)b * export_sklen,
		export_sklen);
}

long
xo_fwm_export(int fwm, int limit, uint8_t *keys, uint64_t *vals, uint64_t *others, int *hit_limit)
{
	struct ofwm *w = &fwms[fwm];
	size_t n = w->n, r, i, sklen = 0, nout = 0;
	uint8_t *rk, *sk;
	uint64_t *rv;
	size_t *idx;

	*hit_limit = 0;
	for (i = 0; i < w->n_aggr; i++) others[i] = 0;
	if (!n) return 0;
	for (i = 0; i < w->n_order; i++) sklen += w->order[i].size;
	rk = malloc(n * (w->keysize ? w->keysize : 1));
	rv = malloc(n * (w->n_aggr ? w->n_aggr : 1) * sizeof(uint64_t));
	sk = malloc(n * (sklen ? sklen : 1));
	idx = malloc(n * sizeof(size_t));
	xo_fwm_dump(fwm, rk, rv);
	for (r = 0; r < n; r++) {
		uint8_t *kptr = sk + r * sklen;
		const uint8_t *naggr = rk + r * w->keysize;
		const uint64_t *aggr = rv + r * w->n_aggr;
		for (i = 0; i < w->n_order; i++) {
			const struct ofield *fld = &w->order[i];
			if (fld->aggr) {
				uint64_t v = htobe64(*aggr);
				if (fld->descending) v = ~v;
				memcpy(kptr, &v, sizeof(uint64_t));
				kptr += sizeof(uint64_t);
				aggr++;
			} else {
				size_t j;
				for (j = 0; j < fld->size; j++) {
					*kptr++ = fld->descending ? (uint8_t)~*naggr : *naggr;
					naggr++;
				}
			}
		}
		idx[r] = r;
	}
	export_sk = sk;
	export_sklen = sklen;
	qsort(idx, n, sizeof(size_t), export_cmp);
	for (r = 0; r < n; r++) {
		const size_t s = idx[r];
		if (!*hit_limit) {
			memcpy(keys + nout * w->keysize, rk + s * w->keysize, w->keysize);
			memcpy(vals + nout * w->n_aggr, rv + s * w->n_aggr, w->n_aggr * sizeof(uint64_t));
			nout++;
			if (limit && ((int)nout >= limit)) *hit_limit = 1;
		} else {
			for (i = 0; i < w->n_aggr; i++) others[i] += rv[s * w->n_aggr + i];
		}
	}
	free(rk); free(rv); free(sk); free(idx);
	return (long)nout;
}

static void
fwms_free(void)
{
	size_t i;
	for (i = 0; i < n_fwms; i++) { free(fwms[i].keys); free(fwms[i].vals); free(fwms[i].used); }
	free(fwms);
	fwms = NULL;
	n_fwms = 0;
}

/* what monit_object_process_nf() sees (monit-objects.h:288); the oracle's
 * consumer adds {1, in_bytes*rate, in_pkts*rate} like the aggregable-field
 * arithmetic of monit-objects.c:995-1007 with scale 1 */
static void
deliver(int mo_idx, struct xo_flow *flow, uint64_t *mask)
{
	uint64_t *c;
	xo_counters();
	c = &counters[mo_idx * 3];
	mask[mo_idx / 64] |= 1ULL << (mo_idx % 64);
	c[0] += 1;
	c[1] += get_nf_val(flow->in_bytes, 8) * flow->sampling_rate;
	c[2] += get_nf_val(flow->in_pkts, 8) * flow->sampling_rate;
	{
		size_t i;
		for (i = 0; i < n_fwms; i++) if (fwms[i].mo == mo_idx) fwm_update(&fwms[i], flow);
	}
}

/* netflow.c:213-244 / sflow.c:84-129: pre-order, children only if the
 * parent matched */
static void
process_mo_rec(struct xo_flow *flow, const int *list, size_t n, uint64_t *mask)
{
	size_t i;
	for (i = 0; i < n; i++) {
		struct mo *m = &mos[list[i]];
		if (!filter_match(m->expr, flow)) continue;
		deliver(list[i], flow, mask);
		if (m->n_kids) process_mo_rec(flow, m->kids, m->n_kids, mask);
	}
}

static uint64_t *
log_record(struct xo_flow *flow, int dropped)
{
	size_t w = xo_log_mask_words();
	static uint64_t scratch[XO_MAX_MO / 64];
	uint64_t *m;

	if (log_max && (log_n >= log_max)) {
		memset(scratch, 0, sizeof(scratch));
		return scratch;
	}
	if (log_n == log_cap) {
		size_t ncap = log_cap ? log_cap * 2 : 1024;
		log_masks = realloc(log_masks, ncap * w * sizeof(uint64_t));
		log_dropped = realloc(log_dropped, ncap);
		if (log_keep_flows) log_flows = realloc(log_flows, ncap * XO_FLOW_INFO_SIZE);
		log_cap = ncap;
	}
	m = &log_masks[log_n * w];
	memset(m, 0, w * sizeof(uint64_t));
	log_dropped[log_n] = (uint8_t)dropped;
	if (log_keep_flows) {
		struct xo_flow *dst = (struct xo_flow *)(log_flows + log_n * XO_FLOW_INFO_SIZE);
		memcpy(dst, flow, XO_FLOW_INFO_SIZE);
		dst->payload_ptr = NULL;
	}
	log_n++;
	return m;
}

/* the tail every decoder shares: netflow.c:292-308, sflow.c:131-157 */
static void
record_done(struct xo_flow *flow, struct pktinfo *fpi)
{
	uint64_t *mask;
	if (!device_rules_check(flow, fpi)) {
		log_record(flow, 1);
		return;
	}
	mask = log_record(flow, 0);
	process_mo_rec(flow, top_mos, n_top, mask);
}

/* ------------------------------------------------------------------ */
/* templates: netflow-templates.c:140-251 reduced to its in-memory     */
/* meaning: the last template received for (version, template id as on */
/* the wire, exporter IPv4, raw source id) wins                        */
/* ------------------------------------------------------------------ */
struct tmpl
{
	uint8_t nf_version;
	uint16_t template_id;   /* as on the wire */
	uint32_t source_ip, source_id;
	uint8_t *data;
	size_t size;
};

static struct tmpl *tmpls = NULL;
static size_t n_tmpls = 0;

static struct tmpl *
template_find(uint8_t ver, uint16_t id, const struct pktinfo *fpi)
{
	size_t i;
	for (i = 0; i < n_tmpls; i++) {
		struct tmpl *t = &tmpls[i];
		if ((t->nf_version == ver) && (t->template_id == id)
			&& (t->source_ip == fpi->src_addr_ipv4) && (t->source_id == fpi->source_id)) {
			return t;
		}
	}
	return NULL;
}

static int
template_put(uint8_t ver, uint16_t id, const struct pktinfo *fpi, const uint8_t *data, size_t size)
{
	struct tmpl *t = template_find(ver, id, fpi);
	if (!t) {
		struct tmpl *tmp = realloc(tmpls, sizeof(*tmp) * (n_tmpls + 1));
		if (!tmp) return 0;
		tmpls = tmp;
		t = &tmpls[n_tmpls++];
		memset(t, 0, sizeof(*t));
		t->nf_version = ver;
		t->template_id = id;
		t->source_ip = fpi->src_addr_ipv4;
		t->source_id = fpi->source_id;
	}
	free(t->data);
	t->data = malloc(size ? size : 1);
	if (!t->data) return 0;
	memcpy(t->data, data, size);
	t->size = size;
	return 1;
}

/* ------------------------------------------------------------------ */
/* Netflow v9 / IPFIX / v5 decode: netflow.c                           */
/* ------------------------------------------------------------------ */

/* netflow.c:79-109: one field.  Non-strings are right-aligned in the
 * SIZEMAX-byte slot, strings left-aligned; a bad length leaves the field
 * absent; an unknown element id is ignored. */
static void
flow_parse(struct xo_flow *flow, int ftype, int flength, const struct pktinfo *p, long fptr)
{
	int fi = id2field[ftype & 0xffff];
	const struct fdesc *fd;

	if (fi < 0) return;
	fd = &fields[fi];
	if ((flength < fd->mn) || (flength > fd->mx)) return;
	if (fd->type == T_STR) {
		rd(p, fptr, FBYTES(flow, fi), flength);
	} else {
		rd(p, fptr, FBYTES(flow, fi) + (fd->mx - flength), flength);
	}
	FHAS(flow, fi) = 1;
	FSIZE(flow, fi) = flength;
}

static void
virtual_fields_init(struct xo_flow *flow, const struct pktinfo *fpi)  /* netflow.c:112-124 */
{
	memcpy(flow->dev_ip, &fpi->src_addr_ipv4, 4);
	flow->dev_ip_size = 4;
	flow->has_dev_ip = 1;
	memcpy(flow->dev_id, &fpi->source_id, 4);
	flow->dev_id_size = 4;
	flow->has_dev_id = 1;
	flow->sampling_rate = (uint32_t)fpi->sampling_rate;
}

/* netflow.c:147-186.  Only the first template of the flowset is read and
 * the cursor advances by that template's size, not by the flowset length. */
static int
parse_v9_template(struct pktinfo *fpi, long *ptr, int length)
{
	uint8_t idb[2];
	uint16_t template_id, field_count;
	int template_size;
	uint8_t *tbytes;

	rd(fpi, *ptr, idb, 2);
	memcpy(&template_id, idb, 2);
	field_count = rd16(fpi, *ptr + 2);
	template_size = 4 + field_count * 4;
	if (template_size > length) {
		return 0;
	}
	tbytes = malloc((size_t)template_size);
	if (!tbytes) return 0;
	rd(fpi, *ptr, tbytes, template_size);
	*ptr += template_size;
	template_put(9, template_id, fpi, tbytes, (size_t)template_size);
	free(tbytes);
	return 1;
}

/* netflow.c:246-317 */
static int
parse_v9_flowset(struct pktinfo *fpi, long *ptr, uint16_t flowset_id, int length, int count)
{
	struct tmpl *t = template_find(9, flowset_id, fpi);
	int field_count, cnt, i;
	long fptr;

	if (!t) return 0;
	field_count = (t->data[2] << 8) | t->data[3];
	fptr = *ptr;
	for (cnt = 0; cnt < count; cnt++) {
		struct xo_flow flow;

		memset(&flow, 0, sizeof(flow));
		for (i = 0; i < field_count; i++) {
			const uint8_t *tl = &t->data[4 + 4 * i];
			int ftype = (tl[0] << 8) | tl[1];
			int flength = (tl[2] << 8) | tl[3];

			/* `length` counts the 4-byte set header the cursor has
			 * already passed, so 4 bytes past the set are admitted */
			if ((fptr + flength - *ptr) > length) break;
			flow_parse(&flow, ftype, flength, fpi, fptr);
			fptr += flength;
		}
		virtual_fields_init(&flow, fpi);
		record_done(&flow, fpi);
	}
	*ptr += length;   /* overshoots the next set by 4 bytes (netflow.c:315) */
	return 1;
}

/* netflow.c:319-365 */
static int
parse_v9(struct pktinfo *fpi, int len)
{
	long ptr;
	int count;

	rd(fpi, 16, &fpi->source_id, 4);   /* kept un-swapped (netflow.c:328) */
	rd(fpi, 8, &fpi->epoch, 4);
	sampling_rate_init(fpi);
	count = rd16(fpi, 2);              /* datagram-wide record count */

	ptr = 20;
	while (ptr < len) {
		uint8_t idb[2];
		uint16_t flowset_id;
		int flowset_id_host, length;

		rd(fpi, ptr, idb, 2);
		memcpy(&flowset_id, idb, 2);
		flowset_id_host = (idb[0] << 8) | idb[1];
		length = rd16(fpi, ptr + 2);
		ptr += 4;

		if (flowset_id_host == 0) {
			if (!parse_v9_template(fpi, &ptr, length)) return 0;
		} else if (flowset_id_host == 1) {
			break;   /* options template: rest of the datagram dropped */
		} else {
			if (!parse_v9_flowset(fpi, &ptr, flowset_id, length, count)) break;
		}
	}
	return 1;
}

/* netflow.c:367-383 */
static void
ipfix_adjust_flength(int *flength, const struct pktinfo *p, long fptr, int *adj)
{
	*adj = 0;
	if (*flength == 65535) {
		uint8_t b[3];
		rd(p, fptr, b, 3);
		if (b[0] != 255) {
			*flength = b[0];
			*adj = 1;
		} else {
			*flength = (b[1] << 8) | b[2];
			*adj = 3;
		}
	}
}

/* netflow.c:389-485: every element normalised to {id without the enterprise
 * bit, length, enterprise number or 0}; stored = 4-byte header + 8 B each */
static int
parse_ipfix_template(struct pktinfo *fpi, long ptr, int length)
{
	uint8_t idb[2];
	uint16_t template_id, field_count;
	uint8_t *t;
	size_t tsize;
	long p, pstart = ptr;
	unsigned int i;
	int avail = length - 4;

	if (4 > length) return 0;
	rd(fpi, ptr, idb, 2);
	memcpy(&template_id, idb, 2);
	field_count = rd16(fpi, ptr + 2);
	if (field_count < 1) return 0;

	tsize = 4 + 8 * (size_t)field_count;
	t = calloc(1, tsize);
	if (!t) return 0;
	rd(fpi, ptr, t, 4);
	p = ptr + 4;
	for (i = 0; i < field_count; i++) {
		uint8_t e[8];
		rd(fpi, p, e, 8);
		if (e[0] & 0x80) {
			if ((p - pstart + 8) > avail) { free(t); return 0; }
			t[4 + 8 * i + 0] = e[0] & 0x7f;
			t[4 + 8 * i + 1] = e[1];
			memcpy(&t[4 + 8 * i + 2], &e[2], 6);
			p += 8;
		} else {
			if ((p - pstart + 4) > avail) { free(t); return 0; }
			memcpy(&t[4 + 8 * i], e, 4);
			p += 4;
		}
	}
	template_put(10, template_id, fpi, t, tsize);
	free(t);
	return 1;
}

/* netflow.c:547-625 */
static int
parse_ipfix_flowset(struct pktinfo *fpi, long ptr, uint16_t flowset_id, int length)
{
	struct tmpl *t = template_find(10, flowset_id, fpi);
	int field_count, i, stop = 0;
	long fptr;

	if (!t) return 0;
	field_count = (t->data[2] << 8) | t->data[3];
	fptr = ptr;
	while (!stop) {
		struct xo_flow flow;

		if ((length - (fptr - ptr)) < field_count) break;
		memset(&flow, 0, sizeof(flow));
		for (i = 0; i < field_count; i++) {
			const uint8_t *el = &t->data[4 + 8 * i];
			int ftype = (el[0] << 8) | el[1];
			int flength = (el[2] << 8) | el[3];
			int adj;

			ipfix_adjust_flength(&flength, fpi, fptr, &adj);
			flow_parse(&flow, ftype, flength, fpi, fptr + adj);
			fptr += flength + adj;
			if ((fptr - ptr) >= length) {
				stop = 1;
				break;
			}
		}
		virtual_fields_init(&flow, fpi);
		record_done(&flow, fpi);
	}
	return 1;
}

/* netflow.c:628-678 */
static int
parse_ipfix(struct pktinfo *fpi, int len)
{
	long ptr = 16;

	rd(fpi, 12, &fpi->source_id, 4);
	rd(fpi, 4, &fpi->epoch, 4);
	sampling_rate_init(fpi);

	while (ptr < len) {
		uint8_t idb[2];
		uint16_t flowset_id;
		int flowset_id_host, length;

		rd(fpi, ptr, idb, 2);
		memcpy(&flowset_id, idb, 2);
		flowset_id_host = (idb[0] << 8) | idb[1];
		length = rd16(fpi, ptr + 2);
		ptr += 4;

		if (flowset_id_host == 2) {
			if (!parse_ipfix_template(fpi, ptr, length)) break;
		} else if (flowset_id_host == 3) {
			/* options template: skipped */
		} else if (flowset_id_host > 255) {
			if (!parse_ipfix_flowset(fpi, ptr, flowset_id, length)) break;
		}
		/* a set length of 0 moves the reference's cursor back onto the
		 * same set header forever (netflow.c:675); the oracle stops */
		if (length == 0) break;
		ptr += length - 4;
	}
	return 1;
}

/* netflow.h:28-76, netflow.c:727-786: fixed 24-byte header + 48-byte records */
static int
parse_v5(struct pktinfo *fpi, int length)
{
	static const struct { int field, off, size; } v5[] = {
		{ F_ip4_src_addr, 0, 4 }, { F_ip4_dst_addr, 4, 4 }, { F_ip4_next_hop, 8, 4 },
		{ F_input_snmp, 12, 2 }, { F_output_snmp, 14, 2 }, { F_in_pkts, 16, 4 },
		{ F_in_bytes, 20, 4 }, { F_first_switched, 24, 4 }, { F_last_switched, 28, 4 },
		{ F_l4_src_port, 32, 2 }, { F_l4_dst_port, 34, 2 }, /* pad1 at 36 */
		{ F_tcp_flags, 37, 1 }, { F_protocol, 38, 1 }, { F_src_tos, 39, 1 },
		{ F_src_as, 40, 2 }, { F_dst_as, 42, 2 }, { F_src_mask, 44, 1 },
		{ F_dst_mask, 45, 1 }, /* pad2 at 46 */
	};
	int nflows = rd16(fpi, 2);
	int i;
	size_t k;

	if ((24 + 48 * nflows) != length) return 0;
	fpi->source_id = fpi->raw[21];           /* engine_id (netflow.c:742) */
	rd(fpi, 8, &fpi->epoch, 4);
	sampling_rate_init(fpi);

	for (i = 0; i < nflows; i++) {
		struct xo_flow flow;
		long base = 24 + 48L * i;

		memset(&flow, 0, sizeof(flow));
		for (k = 0; k < sizeof(v5) / sizeof(v5[0]); k++) {
			int fi = v5[k].field;
			rd(fpi, base + v5[k].off, FBYTES(&flow, fi) + (fields[fi].mx - v5[k].size), v5[k].size);
			FHAS(&flow, fi) = 1;
			FSIZE(&flow, fi) = v5[k].size;
		}
		virtual_fields_init(&flow, fpi);
		record_done(&flow, fpi);
	}
	return 1;
}

static struct pktinfo the_pkt;

static void
pkt_load(const uint8_t *pkt, int len, uint32_t src_ip4)
{
	memset(&the_pkt, 0, sizeof(the_pkt));
	if (len > RAWSZ) len = RAWSZ;
	if (len < 0) len = 0;
	memcpy(the_pkt.raw, pkt, (size_t)len);
	the_pkt.len = len;
	the_pkt.src_addr_ipv4 = src_ip4;
}

/* netflow.c:788-822 */
int
xo_netflow_process(const uint8_t *pkt, int len, uint32_t src_ip4)
{
	int version;

	id2field_init();
	pkt_load(pkt, len, src_ip4);
	version = rd16(&the_pkt, 0);
	switch (version) {
		case 5: return parse_v5(&the_pkt, the_pkt.len);
		case 9: return parse_v9(&the_pkt, the_pkt.len);
		case 10: return parse_ipfix(&the_pkt, the_pkt.len);
		default: return 0;
	}
}

/* ------------------------------------------------------------------ */
/* sFlow v5: sflow-impl.h, sflow.c, rawparse.h                         */
/* ------------------------------------------------------------------ */
enum { RP_NO_ETHER, RP_NO_VLAN, RP_NO_IP, RP_NO_IP_PROTO, RP_OK };
enum { RP_TYPE_ETHER, RP_TYPE_IPv4, RP_TYPE_IPv6 };

/* sflow.c:26-31: left-aligned copy, whatever the field's width */
static void
copy_to_flow(struct xo_flow *flow, int fi, const struct pktinfo *p, long off, int n)
{
	rd(p, off, FBYTES(flow, fi), n);
	FHAS(flow, fi) = 1;
	FSIZE(flow, fi) = n;
}

static void
on_ip(struct xo_flow *flow, const struct pktinfo *p, long iph)  /* sflow.c:47-54 */
{
	copy_to_flow(flow, F_ip4_src_addr, p, iph + 12, 4);
	copy_to_flow(flow, F_ip4_dst_addr, p, iph + 16, 4);
	copy_to_flow(flow, F_src_tos, p, iph + 1, 1);
	copy_to_flow(flow, F_dst_tos, p, iph + 1, 1);
	copy_to_flow(flow, F_min_ttl, p, iph + 8, 1);
	copy_to_flow(flow, F_max_ttl, p, iph + 8, 1);
	copy_to_flow(flow, F_protocol, p, iph + 9, 1);
}

/* sflow.c:57-62: the destination is copied from the SOURCE address, and the
 * protocol is the low byte of the parser's local next-header variable */
static void
on_ip6(struct xo_flow *flow, const struct pktinfo *p, long ip6h, uint64_t nexthdr)
{
	uint8_t nh = (uint8_t)nexthdr;
	copy_to_flow(flow, F_ip6_src_addr, p, ip6h + 8, 16);
	copy_to_flow(flow, F_ip6_dst_addr, p, ip6h + 8, 16);
	copy_to_flow(flow, F_min_ttl, p, ip6h + 7, 1);
	copy_to_flow(flow, F_max_ttl, p, ip6h + 7, 1);
	flow->protocol[0] = nh;
	flow->has_protocol = 1;
	flow->protocol_size = 1;
}

static uint8_t
rd8(const struct pktinfo *p, long off)
{
	uint8_t b;
	rd(p, off, &b, 1);
	return b;
}

/* rawparse.h:89-281 with the hooks of sflow.c:35-77.  Every bounds test is
 * `ptr + header >= end`: a header ending exactly at `end` fails. */
static int
rawpacket_parse(const struct pktinfo *p, long ptr, long end, int t, struct xo_flow *flow)
{
	uint16_t h_proto;  /* host order here; the comparisons are on constants */
	uint64_t nexthdr;
	long iph, ip6h;

	if ((ptr + 14) >= end) return RP_NO_ETHER;
	if (t != RP_TYPE_ETHER) {
		h_proto = (t == RP_TYPE_IPv4) ? 0x0800 : 0x86dd;
		goto ip;
	}
	/* ON_ETH, sflow.c:35-37 */
	copy_to_flow(flow, F_dst_mac, p, ptr + 0, 6);
	copy_to_flow(flow, F_src_mac, p, ptr + 6, 6);
	h_proto = rd16(p, ptr + 12);
	ptr += 14;

	if ((h_proto == 0x8100) || (h_proto == 0x88a8)) {
		if ((ptr + 4) >= end) return RP_NO_VLAN;
		/* ON_VLAN1: the whole TCI, priority bits included */
		copy_to_flow(flow, F_src_vlan, p, ptr, 2);
		copy_to_flow(flow, F_dst_vlan, p, ptr, 2);
		h_proto = rd16(p, ptr + 2);
		ptr += 4;
	}
	if ((h_proto == 0x8100) || (h_proto == 0x88a8)) {
		if ((ptr + 4) >= end) return RP_NO_VLAN;
		copy_to_flow(flow, F_src_dot1q_cvlan, p, ptr, 2);
		copy_to_flow(flow, F_dst_dot1q_cvlan, p, ptr, 2);
		h_proto = rd16(p, ptr + 2);
		ptr += 4;
	}
ip:
	if (h_proto == 0x0800) {
		uint8_t proto;

		if ((ptr + 20) >= end) return RP_NO_IP;
		iph = ptr;
		if ((rd16(p, iph + 6) & 0x3fff) != 0) {
			on_ip(flow, p, iph);
			return RP_NO_IP_PROTO;
		}
		ptr += (rd8(p, iph) & 0x0f) * 4;
		if (rd8(p, iph + 9) == 4) {
			if ((ptr + 20) >= end) return RP_NO_IP;
			iph = ptr;
			ptr += (rd8(p, iph) & 0x0f) * 4;
		}
		on_ip(flow, p, iph);
		proto = rd8(p, iph + 9);
		if (proto == 6) goto tcp;
		if (proto == 17) goto udp;
		if (proto == 1) goto icmp;
		return RP_NO_IP_PROTO;
	} else if (h_proto == 0x86dd) {
		if ((ptr + 40) >= end) return RP_NO_IP;
		ip6h = ptr;
		ptr += 40;
		nexthdr = rd8(p, ip6h + 6);
		if (nexthdr == 4) {
			if ((ptr + 20) >= end) return RP_NO_IP;
			iph = ptr;
			nexthdr = rd8(p, iph + 9);
			ptr += (rd8(p, iph) & 0x0f) * 4;
			on_ip(flow, p, iph);
		} else if (nexthdr == 41) {
			if ((ptr + 40) >= end) return RP_NO_IP;
			ip6h = ptr;
			ptr += 40;
			nexthdr = rd8(p, ip6h + 6);
			on_ip6(flow, p, ip6h, nexthdr);
		} else {
			on_ip6(flow, p, ip6h, nexthdr);
		}
		if (nexthdr == 6) goto tcp;
		if (nexthdr == 17) goto udp;
		if ((nexthdr == 1) || (nexthdr == 58)) goto icmp;
		return RP_NO_IP_PROTO;
	}
	return RP_NO_IP;

tcp:
	if ((ptr + 20) >= end) return RP_NO_IP_PROTO;
	copy_to_flow(flow, F_l4_src_port, p, ptr, 2);
	copy_to_flow(flow, F_l4_dst_port, p, ptr + 2, 2);
	copy_to_flow(flow, F_tcp_flags, p, ptr + 13, 1);
	return RP_OK;
udp:
	if ((ptr + 8) >= end) return RP_NO_IP_PROTO;
	copy_to_flow(flow, F_l4_src_port, p, ptr, 2);
	copy_to_flow(flow, F_l4_dst_port, p, ptr + 2, 2);
	return RP_OK;
icmp:
	if ((ptr + 8) >= end) return RP_NO_IP_PROTO;
	copy_to_flow(flow, F_icmp_type, p, ptr, 1);
	return RP_OK;
}

/* sflow-impl.h:51-65: reads fail when they would pass `end` */
#define SF_READ(DST, N) do { \
	if ((*p + (N)) > end) return 0; \
	rd(pk, *p, (DST), (N)); \
	*p += (((N) % 4) == 0) ? (N) : (N) + (4 - ((N) % 4)); \
} while (0)
#define SF_READ32(V) do { uint8_t b_[4]; SF_READ(b_, 4); \
	(V) = ((uint32_t)b_[0] << 24) | ((uint32_t)b_[1] << 16) | ((uint32_t)b_[2] << 8) | b_[3]; } while (0)

/* sflow.c:159-171 */
static int
sf5_eth(struct pktinfo *pk, struct xo_flow *flow, long p, int t, uint32_t header_len)
{
	long end = p + (long)header_len;   /* not clamped to the datagram */
	if (rawpacket_parse(pk, p, end, t, flow) < RP_NO_IP) {
		return 1;   /* only NO_ETHER / NO_VLAN are skipped */
	}
	record_done(flow, pk);
	return 1;
}

/* sflow-impl.h:68-272 */
static int
sf5_flow(struct pktinfo *pk, struct xo_flow *flow, long *p, long end, int exp)
{
	uint32_t v, nel, j, ifidx, ifidx_be;

	SF_READ32(v);   /* sample length */
	SF_READ32(v);   /* sequence */
	SF_READ32(v);   /* source id (class) */
	if (exp) SF_READ32(v);
	SF_READ32(v);
	flow->sampling_rate = v;
	SF_READ32(v);   /* pool */
	SF_READ32(v);   /* drops */
	if (exp) SF_READ32(v);   /* input format */
	SF_READ32(ifidx);
	if (!exp) ifidx &= 0x3fffffff;
	ifidx_be = htobe32(ifidx);
	memcpy(flow->input_snmp, &ifidx_be, 4);
	flow->has_input_snmp = 1;
	flow->input_snmp_size = 4;
	if (exp) SF_READ32(v);   /* output format */
	SF_READ32(ifidx);
	if (!exp) ifidx &= 0x3fffffff;
	ifidx_be = htobe32(ifidx);
	memcpy(flow->output_snmp, &ifidx_be, 4);
	flow->has_output_snmp = 1;
	flow->output_snmp_size = 4;

	SF_READ32(nel);
	for (j = 0; j < nel; j++) {
		uint32_t tag, ell;

		SF_READ32(tag);
		SF_READ32(ell);
		if (tag == 1) {
			uint32_t header_proto, stripped, header_len;
			int t;

			SF_READ32(header_proto);
			SF_READ(&flow->in_bytes[4], 4);
			flow->has_in_bytes = 1;
			flow->in_bytes_size = 8;
			flow->in_pkts[7] = 1;
			flow->has_in_pkts = 1;
			flow->in_pkts_size = 8;
			SF_READ32(stripped);
			(void)stripped;
			SF_READ32(header_len);
			if (header_proto == 1) t = RP_TYPE_ETHER;
			else if (header_proto == 11) t = RP_TYPE_IPv4;
			else if (header_proto == 12) t = RP_TYPE_IPv6;
			else return 0;
			if (!sf5_eth(pk, flow, *p, t, header_len)) return 0;
			*p += ((header_len % 4) == 0) ? header_len : header_len + (4 - (header_len % 4));
		} else if (tag == 1001) {
			SF_READ32(v); SF_READ32(v); SF_READ32(v); SF_READ32(v);
		} else if (tag == 3) {
			uint8_t w[4], dport[4];

			SF_READ(&flow->in_bytes[4], 4);
			flow->has_in_bytes = 1;
			flow->in_bytes_size = 8;
			flow->in_pkts[7] = 1;
			flow->has_in_pkts = 1;
			flow->in_pkts_size = 8;
			/* sflow-impl.h:236-252 */
			SF_READ(w, 4); flow->protocol[0] = w[3]; flow->has_protocol = 1; flow->protocol_size = 1;
			SF_READ(w, 4); memcpy(flow->ip4_src_addr, w, 4); flow->has_ip4_src_addr = 1; flow->ip4_src_addr_size = 4;
			SF_READ(w, 4); memcpy(flow->ip4_dst_addr, w, 4); flow->has_ip4_dst_addr = 1; flow->ip4_dst_addr_size = 4;
			SF_READ(w, 4); memcpy(flow->l4_src_port, &w[2], 2); flow->has_l4_src_port = 1; flow->l4_src_port_size = 2;
			SF_READ(dport, 4); memcpy(flow->l4_dst_port, &dport[2], 2); flow->has_l4_dst_port = 1; flow->l4_dst_port_size = 2;
			SF_READ(w, 4); flow->tcp_flags[0] = w[3]; flow->has_tcp_flags = 1; flow->tcp_flags_size = 1;
			SF_READ(w, 4); flow->src_tos[0] = w[3]; flow->has_src_tos = 1; flow->src_tos_size = 1;
			if (flow->protocol[0] == 1) {
				flow->icmp_type[0] = dport[1];
				flow->has_icmp_type = 1;
				flow->icmp_type_size = 1;
			}
			record_done(flow, pk);   /* sf5_parsed, sflow.c:131-157 */
		} else {
			*p += ell;   /* not re-aligned (sflow-impl.h:268) */
		}
	}
	return 1;
}

/* sflow-impl.h:274-405 */
int
xo_sflow_process(const uint8_t *pkt, int len, uint32_t src_ip4)
{
	struct pktinfo *pk = &the_pkt;
	long pos = 0, end, *p = &pos;
	uint32_t v, nsmpl, i;
	uint8_t dev_ip4[4] = {0}, dev_ip6[16] = {0}, dev_id[4];
	int dev_ip_ver = 4;
	struct xo_flow flow;

	id2field_init();
	pkt_load(pkt, len, src_ip4);
	end = pk->len;

	SF_READ32(v);
	if (v != 5) return 0;
	SF_READ32(v);
	if (v == 1) {
		SF_READ(dev_ip4, 4);
	} else if (v == 2) {
		dev_ip_ver = 6;
		SF_READ(dev_ip6, 16);
	} else {
		return 0;
	}
	SF_READ(dev_id, 4);
	SF_READ32(v);   /* sequence */
	SF_READ32(v);   /* uptime */
	SF_READ32(nsmpl);

	for (i = 0; i < nsmpl; i++) {
		SF_READ32(v);
		if ((v == 1) || (v == 3)) {
			/* sflow_reset, sflow-impl.h:274-291 */
			memset(&flow, 0, sizeof(flow));
			memcpy(flow.dev_id, dev_id, 4);
			flow.has_dev_id = 1;
			flow.dev_id_size = 4;
			if (dev_ip_ver == 4) {
				memcpy(flow.dev_ip, dev_ip4, 4);
				flow.has_dev_ip = 1;
				flow.dev_ip_size = 4;
			} else {
				memcpy(flow.dev_ip6, dev_ip6, 16);
				flow.has_dev_ip6 = 1;
				flow.dev_ip6_size = 16;
			}
			if (!sf5_flow(pk, &flow, p, end, v == 3)) return 0;
		} else {
			uint32_t l;
			SF_READ32(l);
			*p += l;
		}
	}
	return 1;
}

/* ------------------------------------------------------------------ */
void
xo_reset(void)
{
	size_t i, j;

	xo_log_clear();
	for (i = 0; i < 2; i++) for (j = 0; j < 2; j++) (void)xo_geodb_set((int)i, j ? 6 : 4, NULL, 0);
	for (i = 0; i < n_mos; i++) {
		fexpr_free(mos[i].expr);
		free(mos[i].kids);
	}
	free(mos); mos = NULL; n_mos = 0;
	free(top_mos); top_mos = NULL; n_top = 0;
	fwms_free();
	for (i = 0; i < n_devices; i++) {
		for (j = 0; j < devices[i].n_exprs; j++) fexpr_free(devices[i].exprs[j]);
		free(devices[i].exprs);
	}
	free(devices); devices = NULL; n_devices = 0;
	xo_filter_free_all();
	for (i = 0; i < n_iplists; i++) {
		free(iplists[i].nodes4);
		free(iplists[i].nodes6);
	}
	free(iplists); iplists = NULL; n_iplists = 0;
	for (i = 0; i < n_tmpls; i++) free(tmpls[i].data);
	free(tmpls); tmpls = NULL; n_tmpls = 0;
	log_keep_flows = 0;
	log_max = 0;
	xo_set_port_names(NULL);
}

/* filter_match() of object `mo`'s own expression on a record given as struct
 * flow_info bytes (what the reference's process_mo_*_rec() loops call per
 * object, netflow.c:223) */
int
xo_mo_match(int mo, const void *flow)
{
	if ((mo < 0) || ((size_t)mo >= n_mos)) return 0;
	return filter_match(mos[mo].expr, (struct xo_flow *)flow);
}
