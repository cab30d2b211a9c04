/*
 * xe_internal.h -- structures shared by the host modules and the HIP kernels.
 * Everything in the "device formats" section is laid out by hand (fixed-width
 * members, no pointers) because the same bytes are read by the GPU.
 */
#ifndef xe_internal_h_included
#define xe_internal_h_included

#include <stddef.h>
#include <stdint.h>
#include "../../include/xe_gpu.h"

#include <pthread.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ fields */
enum { XE_T_ADDR = 0, XE_T_INT = 1, XE_T_MAC = 2, XE_T_STR = 3 };

#define XE_NFIELDS 66          /* 63 table fields + dev_ip6, dev_id, dev_mark */
#define XE_NTABLE_FIELDS 63
#define XE_F_NONE 0xFF

/* lookup bits (list memberships) of a record: configurations with up to 8
 * (list group, family, field) lookups use one 64-bit word, up to 32 use four */
#define XE_LK_WORDS 4

struct xe_field
{
	const char *name;
	int32_t id;          /* element id on the wire, -1 for virtual fields */
	uint8_t type;
	uint16_t size_min, size_max;
	uint16_t off, off_size, off_has;   /* offsets in struct flow_info */
	uint16_t col_width;  /* bytes per record in the structure-of-arrays output */
};

extern const struct xe_field xe_fields[XE_NFIELDS];
int xe_field_by_id(int id);                /* -1 when unknown */
int xe_field_by_name(const char *name);    /* -1 when unknown */
#define XE_OFF_PAYLOAD_PTR 2248
#define XE_OFF_SAMPLING_RATE 2304

/* indices the code refers to by name (kept in step with xe_fields[] by a
 * start-up self check) */
enum
{
	XF_in_bytes = 0, XF_in_pkts = 1, XF_protocol = 2, XF_src_tos = 3, XF_dst_tos = 4,
	XF_tcp_flags = 5, XF_l4_src_port = 6, XF_ip4_src_addr = 7, XF_src_mask = 8,
	XF_input_snmp = 9, XF_l4_dst_port = 10, XF_ip4_dst_addr = 11, XF_dst_mask = 12,
	XF_output_snmp = 13, XF_ip4_next_hop = 14, XF_src_as = 15, XF_dst_as = 16,
	XF_bgp_next_hop = 17, XF_last_switched = 18, XF_first_switched = 19,
	XF_ip6_src_addr = 20, XF_ip6_dst_addr = 21, XF_icmp_type = 22, XF_sampler_id = 23,
	XF_min_ttl = 24, XF_max_ttl = 25, XF_frag_id = 26, XF_src_mac = 27, XF_dst_mac = 28,
	XF_src_vlan = 29, XF_dst_vlan = 30, XF_ip_protocol_version = 31, XF_direction = 32,
	XF_if_name = 33, XF_fwd_status = 34, XF_flow_start_ms = 35, XF_flow_end_ms = 36,
	XF_flow_end_reason = 37, XF_ip_ttl = 38, XF_ioctets = 39, XF_roctets = 40,
	XF_ipackets = 41, XF_rpackets = 42, XF_ingrs_vrf = 43, XF_egrs_vrf = 44,
	XF_src_dot1q_vlan = 45, XF_src_dot1q_cvlan = 46, XF_dst_dot1q_vlan = 47,
	XF_dst_dot1q_cvlan = 48, XF_obs_time_ms = 49, XF_vas_session_id = 50,
	XF_vas_http_host = 51, XF_dev_ip = 52, XF_dns_name = 53, XF_dns_ips = 54, XF_sni = 55,
	XF_class0 = 56, XF_class1 = 57, XF_class2 = 58, XF_class3 = 59, XF_class4 = 60,
	XF_pad1 = 61, XF_pad2 = 62, XF_dev_ip6 = 63, XF_dev_id = 64, XF_dev_mark = 65
};

/* ------------------------------------------------------------------ device formats */

/* one field of an exporter template */
struct xe_tent
{
	uint16_t len;        /* encoded length; 0xFFFF = IPFIX variable length */
	uint8_t field;       /* destination field, XE_F_NONE = unknown element (skipped) */
	uint8_t flags;       /* XE_TE_* */
	uint16_t off;        /* offset inside a fixed-length record */
	uint8_t next_same;   /* next entry with the same destination, 0xFF = none */
	uint8_t pad;
};
#define XE_TE_VALID 1    /* length inside [size_min, size_max] */
#define XE_TE_STR 2      /* string: left-aligned copy */
#define XE_TE_REPEAT 4   /* an earlier entry of this template has the same destination */

#define XE_MAX_RECORDS_PER_DATAGRAM (1u << 20)
#define XE_TMPL_MAXF 254
#define XE_TMPL_UNSUPPORTED 0xFFFFFFFFu   /* template slot: too many elements for the device form */
#define XE_MODE_V9 0     /* field admitted while pos + len <= avail (netflow.c:284) */
#define XE_MODE_IPFIX 1  /* field parsed, then stop when pos >= avail (netflow.c:596) */

struct xe_tmpl
{
	uint32_t ent_base;   /* first entry in the entries array */
	uint16_t nfields;
	uint16_t stride;     /* record length when fixed */
	uint8_t mode;
	uint8_t varlen;      /* has a variable-length element */
	uint8_t repeats;     /* some destination occurs more than once */
	uint8_t pad;
	uint32_t has_lo_hi[3]; /* unused on device; presence mask of a fully parsed record */
	uint8_t fmap[72];    /* field -> first entry index, 0xFF = not in template */
};

/* a run of records that share template, exporter and spacing */
struct xe_set
{
	uint64_t data_off;   /* first record, byte offset in the batch buffer */
	uint32_t first_rec;  /* index of its first record in the batch */
	uint32_t nrec;
	int32_t avail;       /* byte budget measured from data_off (the reference's
	                        `length` minus what was consumed before) */
	uint32_t stride;     /* bytes between consecutive records */
	uint32_t tmpl;
	int32_t dgram_left;  /* bytes from data_off to the datagram's end: reads
	                        past it return zero */
	uint32_t exporter;
	uint32_t flags;      /* XE_SET_* */
};
#define XE_SET_CHECKED 1 /* a read may pass the datagram end: bounds-check bytes */
#define XE_SET_PARTIAL 2 /* some record of the run does not get all template fields */

struct xe_exporter
{
	uint8_t dev_ip[4];    /* flow_packet_info.src_addr_ipv4 bytes */
	uint8_t dev_id[4];    /* raw source id / observation domain bytes */
	uint32_t sampling_rate;
	int32_t device;       /* index of the devices.conf entry, -1 none */
};

/* one sFlow flow sample element that yields a record */
struct xe_sfrec
{
	uint64_t hdr_off;     /* sampled header (tag 1) or IPv4 struct (tag 3) in the batch */
	uint32_t header_len;  /* tag 1: declared header length */
	int32_t dgram_left;   /* bytes from hdr_off to the datagram end */
	uint32_t frame_len_be;/* 4 bytes as on the wire: in_bytes[4..8] */
	uint32_t sampling_rate;
	uint32_t in_if_be, out_if_be; /* input_snmp / output_snmp bytes as stored */
	uint8_t kind;         /* 1 raw header, 3 IPv4 struct */
	uint8_t rp_type;      /* 0 ethernet, 1 IPv4, 2 IPv6 */
	uint8_t agent_ver;    /* 4 or 6 */
	uint8_t chained;      /* 1 = the previous record's fields carry over (same sample) */
	uint8_t dev_id[4];
	uint8_t agent[16];
	uint32_t exporter;
	uint32_t pad;
};

/* flat filter programs */
enum { XE_OP_BASIC = 0, XE_OP_NOT = 1, XE_OP_AND = 2, XE_OP_OR = 3 };
enum { XE_BT_ADDR4 = 0, XE_BT_ADDR6 = 1, XE_BT_RANGE = 2, XE_BT_MAC = 3, XE_BT_STRING = 4 };
enum { XE_FN_NONE = 0, XE_FN_DIV = 1, XE_FN_DIV_R = 2, XE_FN_DIV_L = 3, XE_FN_MIN = 4, XE_FN_TFSTR = 5,
	/* portstr / ppstr are lowered to sets over the port values when the handle
	 * has a port-name table; the rest is parsed (for the dump and so that the
	 * object keeps its place) and left to the caller's own filter_match() */
	XE_FN_PORTSTR = 6, XE_FN_PPSTR = 7, XE_FN_MFREQ = 8, XE_FN_GEO = 9, XE_FN_ASN = 10, XE_FN_ASD = 11 };
enum { XE_DIR_NONE = 0, XE_DIR_SRC = 1, XE_DIR_DST = 2, XE_DIR_BOTH = 3 };

struct xe_term
{
	uint8_t type, dir, fsrc, fdst;   /* fsrc/fdst: field indices */
	uint8_t func, fa, fb, pad;       /* functions: operand fields.  pad: on the host the keyword index (for the
	                                    dump); on the device XE_TERM_ZERO_ABSENT or 0 */
	int32_t k;
	uint32_t nopnd;
	uint32_t opnd_base;              /* index into the operand array */
};

#define XE_TERM_ZERO_ABSENT 1    /* address term made from a geo / AS function: an absent field reads as zeros */
#define XE_OPND_WORDS 10
struct xe_opnd
{
	uint32_t is_list;
	uint32_t w[XE_OPND_WORDS - 1];
	/* range: w[0]=low w[1]=high (int32)   addr4: w[0]=addr w[1]=mask (bytes as stored)
	 * list:  w[0]=list index              addr6: w[0..3]=addr w[4..7]=mask
	 * mac:   w[0..1] = 6 bytes
	 * tfstr: w[0..8] = the operand's text, NUL-terminated (cut at 35 characters;
	 *        the longest TCP flags text has 31) */
};

#define XE_PROG_CONST 1      /* result is `cval` whatever the record */
struct xe_prog
{
	uint32_t op_base, nops;
	int32_t parent;
	uint16_t flags;
	uint16_t cval;
};

/* Flattened IP lists.  Lists are merged in groups of XE_LIST_GROUP (registry
 * order); per group and address family: a root over the first 16 address bits
 * (65536 entries, 256 KB: it stays in the L2) or, on request, over the first
 * 24 (2^24 entries, 64 MB: every IPv4 prefix up to /24 is answered by ONE
 * load, but that load no longer hits the L2 - measured on MI355X: 8 lists of a
 * million prefixes 3 % faster, lists of 10 000 prefixes 25-80 % slower), then
 * nodes of 2^XE_NODE_BITS entries, one per value of the next XE_NODE_BITS
 * address bits.  8 bits per node measured faster on MI355X than 4
 * (tables a quarter of the size, but one more dependent load per walk).
 * An entry (root and node alike), by its top two bits:
 *   10  the walk ends here, bits 0..7 are the answers of the group's lists
 *       (bit l = list l);
 *   00  index of the node for the next address bits;
 *   01  index of a chain record;
 *   11  a short chain kept in the entry itself (no further load):
 *       bit 29 clear: answers of the other lists (bits 0-7) | l << 8 |
 *                     len << 11 (1..13) | path << 15 (len bits, right-aligned)
 *                     | s << 28
 *       bit 29 set:   path (bits 0-15) | l << 16 | s << 19 | len << 20 (1..16),
 *                     the other lists answer 0
 *       (s: see the chain record)
 * A visit is one 4-byte load.
 * Chain record (XE_CHAIN_WORDS words): only list `l` can still go deeper, along
 * one path of `len` bits that ends in a prefix:
 *   word 0      answers of the other lists (bits 0-7) | l << 8 | len << 11 | s << 18
 *   words 2-5   the path, left-aligned in 128 bits (hi word pair first)
 * the list's answer is 1 iff the next `len` address bits equal the path, or
 * s is set (the path starts at a prefix of the list) and the first of them
 * differs from the path's. */
#define XE_TRIE_TERM 0x80000000u
#define XE_TRIE_CHAIN 0x40000000u
#define XE_TRIE_INLINE 0xC0000000u
#define XE_TRIE_INLINE_LONG 0x20000000u
#define XE_CHAIN_WORDS 8
#define XE_LIST_GROUP 8
#ifndef XE_NODE_BITS
#define XE_NODE_BITS 8
#endif
#define XE_NODE_WORDS (1u << XE_NODE_BITS)
/* one distinct (list, address family, field) membership test; the fast kernel
 * resolves all of them for a tile in batches before any term is evaluated */
struct xe_lookup
{
	uint32_t list;
	uint8_t family;      /* 4 or 6 */
	uint8_t field;
	uint8_t wide4;       /* IPv4 test over the first 4 bytes of a 16-byte field */
	uint8_t pad;
};

/* A probe = all membership tests of one field against one group of lists: one
 * walk; its answers occupy lookup bits [8 * probe, 8 * probe + 8). */
struct xe_probe
{
	uint32_t root_off;   /* word offset of the group's root table (this family), 0 = empty */
	uint8_t family;      /* 4 or 6 */
	uint8_t field;
	uint8_t wide4;       /* bit 0: IPv4 test over the first 4 bytes of a 16-byte field;
	                        bit 1: an absent field is looked up as the zero address (lists made of geo / AS functions) */
	uint8_t used;        /* bit l: list (8 * group + l) is tested on this field */
	uint32_t group;
	uint32_t nodes_off;  /* word offset of node 0 */
	uint32_t chains_off; /* word offset of chain record 0 */
	uint32_t root_bits;  /* address bits the root table is indexed with: 16 or 24 */
	uint32_t pad[2];
};

/* ---- fast kernel tables (built per batch: they hold output addresses) ---- */

/* one run of complete records, with what the kernel needs of its exporter */
struct xe_frun
{
	uint32_t data_off;   /* first record, byte offset in the batch buffer (< 4 GiB) */
	uint32_t first_rec;  /* batch index of its first record */
	uint32_t start;      /* position of its first record in the template-grouped order */
	uint32_t nrec;
	uint32_t stride;
	uint32_t rate;       /* sampling rate */
	uint32_t dev_ip;     /* exporter address / source id as big-endian numbers */
	uint32_t dev_id;
};

/* one tile of the fast kernel: 64 * R consecutive positions of one template
 * group.  Tiles are listed in the order of the batch index of their first
 * record, whatever their template, so that neighbours in the list write
 * neighbouring parts of the output columns. */
struct xe_ftile
{
	uint32_t first_run;  /* first run that may hold its records */
	uint32_t tmpl;       /* template slot */
	uint32_t base;       /* first position in the template-grouped order */
	uint32_t pad;
};

/* decode micro-op: one known, valid template field.
 * w0: bits 0-15 offset in the record, 16-23 wire length, 24-31 kind,
 *     32-39 operand slot (0xFF none), 40-47 index of the slot's upper half
 * w1: device address of the field's column */
struct xe_fop { uint64_t w0, w1; };
#define XE_FOP_LGW 7       /* kind bits 0-2: log2(column width) */
#define XE_FOP_COPY 8      /* wire length == column width: the bytes are stored as they are */
#define XE_FOP_BYTES 16    /* in_bytes: value feeds the octet counter */
#define XE_FOP_PKTS 32     /* in_pkts */
#define XE_FOP_MAC 64      /* 6-byte value kept in an 8-byte column */

/* per template.  The micro-ops are sorted by kind: plain copies of 1, 2, 4, 8
 * and 16 bytes (wire length == column width), then everything else. */
struct xe_fplan
{
	uint32_t op_base;
	uint32_t cnt[6];     /* copies of 1, 2, 4, 8, 16 bytes; general ops */
	uint32_t nparsed;    /* fields a complete record parses (all of the template's) */
	uint32_t nchunk;     /* 16-byte pieces that cover one record */
	uint32_t pad;
	uint64_t probes[XE_LK_WORDS]; /* byte i = i-th probe whose field the template carries, 0xFF = end */
};

/* per template and field (XE_NFIELDS words per template): where the field's
 * bytes are in the record, for filter operands.
 * bits 0-15 offset, 16-20 wire length, 21-25 maximum size of the field,
 * 30 virtual (dev_ip / dev_id: value comes from the exporter), 31 present */
#define XE_FFLD_VIRTUAL 0x40000000u
#define XE_FFLD_PRESENT 0x80000000u

/* fast-kernel view of a distinct term */
struct xe_fterm
{
	uint64_t kind;       /* bit 0: every operand is an IP list -> result = lookup bits & mask;
	                        bit 1: some program needs the term's bit */
	uint64_t mask[XE_LK_WORDS];
};

/* Fast-kernel view of the object programs.
 *
 * Objects without a parent whose program is a single term are resolved through
 * a table: "atom" a (0-63: lookup bit a; 64-127: term a - 64) -> the set of
 * such objects that match when the atom is true; a record's match bits are the
 * OR of the rows of its true atoms (xe_kparams.motab[word][XE_MOTAB_ROWS(lw)]).
 * All other objects are listed, in index order, as xe_fprog entries. */
#define XE_MOTAB_ROWS(LW) (64 * (LW) + 64)   /* lookup atoms of LW words, then the first 64 terms */
struct xe_fprog
{
	uint64_t mask;       /* XE_FPROG_MASK: lookup bits that make it match */
	uint32_t kind;
	uint32_t term;       /* XE_FPROG_TERM: the one term */
	uint32_t op_base, nops;   /* XE_FPROG_GENERAL: the postfix program */
	int32_t parent;      /* object that must have matched too, -1 none */
	uint32_t cval;       /* XE_FPROG_CONST: the result */
	uint32_t mo;         /* the object's index */
	uint32_t pad[3];
};
enum { XE_FPROG_GENERAL = 0, XE_FPROG_CONST = 1, XE_FPROG_MASK = 2, XE_FPROG_TERM = 3,
	XE_FPROG_GATE = 4 };   /* GATE: matched elsewhere, only the parent condition is left */

/* Objects whose program is a conjunction of clauses (each an OR of possibly
 * negated atoms) are matched all at once: an object fails when one of the
 * atoms it needs is false or one it forbids is true.
 * Atoms: 0-255 lookup bits, 256-511 terms, 512-575 derived (multi-literal clauses).
 * One xe_cjrow per (atom, mask word) with objects that care about the atom. */
#define XE_ATOM_TERM0 (64 * XE_LK_WORDS)
#define XE_ATOM_DERIVED0 (XE_ATOM_TERM0 + 256)
#define XE_MAX_DERIVED 64
struct xe_cjrow
{
	uint32_t atom, word;
	uint64_t need;       /* objects of this word that fail when the atom is false */
	uint64_t forbid;     /* objects that fail when it is true */
};
/* derived atom: true iff some positive literal is true or some negative one
 * false; literals are lookup bits and terms 0-63 */
struct xe_cjderived { uint64_t pos_lk[XE_LK_WORDS], neg_lk[XE_LK_WORDS], pos_tb, neg_tb; };

/* Range terms grouped by the field they read.  Per group: `n` sorted
 * breakpoints over the field's value (int32, as the reference compares it)
 * and n + 1 table entries, one per interval between breakpoints: the set of
 * terms (4 x u64 = term bits 0..255) with a range that covers the interval.
 * Offsets count 32-bit words of the group blob. */
#define XE_MAX_RGROUPS 16
struct xe_rgroup { uint32_t field, n, bp_off, tab_off; };

/* Need / forbid rows folded into tables indexed by 8 atoms at once: per group
 * (one byte of the lookup bits, term bits or derived bits that some row
 * reads) 256 entries of `mask_words` u64: the objects that fail when the byte
 * has that value.  Byte k: 0..31 lookup bits, 32..63 term bits, 64..71 derived. */
#define XE_CJ_MAX_GROUPS (8 * XE_LK_WORDS + 32 + 8)


struct xe_list_dev
{
	uint32_t root4, root6;   /* word offsets of the list's GROUP tables, 0 = family empty */
	uint32_t nodes4, nodes6; /* word offsets of node 0 */
	uint32_t bit;            /* the list's bit in the group's answer bytes */
	uint32_t chains4, chains6; /* word offsets of chain record 0 */
	uint32_t rbits;          /* address bits the roots are indexed with: IPv4 table in bits 0-7, IPv6 table in bits 8-15 */
};

/* ------------------------------------------------------------------ host side */
struct xe_blob { void *p; size_t n, cap; };  /* growable array helper */
int xe_blob_push(struct xe_blob *b, const void *item, size_t item_size);

/* ip lists */
struct xe_bnode { uint32_t next[2]; int32_t is_leaf; };
struct xe_iplist
{
	char name[256];
	struct xe_bnode *n4, *n6;
	size_t c4, c6, cap4, cap6;
};
struct xe_iplists { struct xe_iplist *l; size_t n; };

int xe_iplist_load_text(struct xe_iplists *reg, const char *name, const char *text, size_t len);
int xe_iplist_load_dir(struct xe_iplists *reg, const char *dir);
int xe_iplist_find(const struct xe_iplists *reg, const char *name);  /* first match, -1 none */
void xe_iplists_free(struct xe_iplists *reg);
/* appends the flattened form of every list to `words` (uint32 array) and
 * fills dev[] (reg->n entries) */
/* root_bits: 16 or 24 (0 = 16) for every table */
int xe_iplists_flatten(const struct xe_iplists *reg, struct xe_blob *words, struct xe_list_dev *dev, int root_bits);

/* filters */
struct xe_filter     /* one compiled expression, host form */
{
	uint32_t *ops;  size_t nops;      /* op | term_index << 2 */
	struct xe_term *terms; size_t nterms;
	struct xe_opnd *opnds; size_t nopnds;
	/* host-only operand annotations for the dump: mask length / list index /
	 * offset of a quoted string's text in strs (-1: the operand has no text) */
	int32_t *opnd_note;
	char *strs; size_t nstrs;         /* texts of the quoted-string operands, NUL-terminated */
	int device_ok;                    /* 0: uses something the device cannot evaluate */
	int needs_port_names;             /* has portstr() / ppstr() terms: evaluable with a port-name table only */
	int needs_geodb;                  /* has geo / AS function terms: bit (2 * kind + (IPv6 ? 1 : 0)) per database they read */
	char why[128];
};
int xe_filter_compile(const struct xe_iplists *reg, const char *text, struct xe_filter *out,
	char *err, size_t errlen);
void xe_filter_free(struct xe_filter *f);
int xe_filter_dump_text(const struct xe_iplists *reg, const struct xe_filter *f, char *out, size_t outlen);
/* a field of a fixed-window table (struct field, monit-objects.h) */
/* func != XE_FN_NONE: a function form; field / fb are its arguments, k its third
 * (div_r, div_l), size the bytes it takes in the reference's key */
struct xe_agfield { uint8_t aggr, field, descending, scale; uint32_t size; uint8_t func, fb; int32_t k; };
int xe_field_parse(const char *text, struct xe_agfield *out, char *err, size_t errlen);
/* row i of the keyword tables (filter.def, filter-ag.def); 0 past the end */
int xe_filter_keyword(int i, const char **kw, int *type, int *src, int *dst);
int xe_filter_aggr_keyword(int i, const char **kw, int *field, int *scale);
/* static analysis of the stack discipline (filter.c:960-1012): returns 1 and
 * sets *cval when the result does not depend on the record */
int xe_filter_constant(const struct xe_filter *f, int *cval);
/* div_l(): the int32 quotient line cut into pieces [lo, hi] on which
 * (int)pow(k, (int)(log(q) / log(k))) (filter.h:407-409) has one value, found
 * with this host's libm; returns the number of pieces (<= cap) or -1 */
int xe_div_l_pieces(int k, int32_t *lo, int32_t *hi, int32_t *val, int cap);

struct xe_host;
/* geo / AS databases (xe_geodb.c) */
/* takes a copy of the image after checking that it is a tree of nodes inside
 * the image; bytes == 0 drops the database */
int xe_geodb_set(struct xe_host *hs, int kind, int family, const void *img, size_t bytes, char *err, size_t errlen);
/* A function term over a database as an IP list: the list's trie has the
 * shape of the database's, and a node is a leaf iff the database's node is one
 * and its payload meets the term (asn inside one of the n ranges lo[i]..hi[i];
 * asd / the geo field `geo_field` equal to one of the n strings).  Walking that
 * list the way iplist_match4/6 does answers exactly what
 * filter_function_as() / filter_function_geoip() answer (filter.c:753-834).
 * Appends the list to the registry (named so that no filter text can refer to
 * it) and returns its index, or a negative error code. */
int xe_geodb_list(struct xe_host *hs, int func, int geo_field, int family,
	const int32_t *lo, const int32_t *hi, const char *const *strs, size_t n);

/* devices.conf entries */
struct xe_device
{
	int use_ip, use_id;
	uint32_t ip, id;   /* ip: bytes as stored; id: htonl(value) */
	int sampling_rate, skip_unmarked;
	struct xe_filter *marks; size_t nmarks;
};

struct xe_mo
{
	char name[256];
	int parent;
	/* the filter uses something only the caller's own filter_match() evaluates
	 * (strings, mfreq, geo / AS functions ...): on the device the object matches
	 * whenever its parent does, the caller decides the rest */
	int host_only;
	char why[128];
	struct xe_filter filter;
};

/* exporter templates kept across batches */
struct xe_tkey { uint8_t ver; uint16_t tid; uint32_t src_ip, source_id; };
struct xe_tslot
{
	struct xe_tkey key;
	uint8_t *raw; size_t raw_len;     /* bytes as stored by the reference */
	uint32_t dev_slot;                /* index into the device template table */
};

/* geo / AS database: the reference's own file image (geoip.h:43-55), an array
 * of binary-trie nodes {uint32_t next[2]; int is_leaf; payload} */
struct xe_geodb { uint8_t *img; size_t nnodes; };
#define XE_GEO_NODE_BYTES 228     /* 12 + struct geoip_info (214), padded to 4 */
#define XE_AS_NODE_BYTES 216      /* 12 + struct as_info (4 + 200) */

struct xe_host
{
	struct xe_iplists lists;
	/* the last nlists_made lists of the registry were made at commit from the
	 * geo / AS databases: nlists_fn of them one per function term, the others
	 * empty lists that put the first of those at a group boundary */
	size_t nlists_made, nlists_fn;
	struct xe_geodb geodb[2][2];   /* [0 geo, 1 AS][0 IPv4, 1 IPv6] */
	struct xe_mo *mos; size_t nmos;
	struct xe_device *devs; size_t ndevs;
	/* template store */
	struct xe_tslot *tslots; size_t ntslots, tslots_cap;
	uint32_t *thash; size_t thash_cap;        /* open addressing, slot index + 1 */
	struct xe_blob dev_tmpls;   /* struct xe_tmpl */
	struct xe_blob dev_tents;   /* struct xe_tent */
	uint32_t v5_slot_plus1;     /* device template of the fixed v5 record, 0 = not built */
	/* the pre-pass may walk a batch with several threads (one exporter always
	 * on one thread): the template store is then guarded by tlock */
	pthread_rwlock_t tlock;
	int mt;                     /* 1 while several threads are walking */
	int prepass_threads;        /* 0 = default */
	/* memory kept between batches (fresh memory costs a page fault per 4 KiB,
	 * which dominates the pre-pass of large sFlow batches): the walking threads'
	 * state with its grown arrays, and a few big buffers of freed plans */
	void *walkers; int nwalkers;
	pthread_mutex_t pool_mu;
	void *pool_p[8]; size_t pool_cap[8];
	/* allocator of those big buffers (NULL: malloc / free); the GPU library
	 * supplies page-locked memory so that the plan travels at link speed */
	void *(*big_alloc)(size_t bytes);
	void (*big_free)(void *p);
	/* counts every change of the template store (what the device's copy of
	 * it is compared with) */
	uint64_t tmpl_epoch;
	/* log of the changes since xe_host_undo_begin() */
	int undo_on;
	struct xe_blob undo;        /* struct xe_tundo */
	char err[512];
};
struct xe_tundo { uint32_t slot; uint32_t dev_slot; uint8_t *raw; size_t raw_len; };
void xe_host_undo_begin(struct xe_host *hs);
size_t xe_host_undo_count(const struct xe_host *hs);
uint32_t xe_host_undo_src_ip(const struct xe_host *hs, size_t i);   /* exporter address of change i */
void xe_host_undo_rollback(struct xe_host *hs);
void xe_host_undo_end(struct xe_host *hs);

void xe_host_init(struct xe_host *hs);
void xe_host_free(struct xe_host *hs);

/* result of the host pre-pass over one batch */
struct xe_plan
{
	struct xe_blob sets;       /* struct xe_set */
	struct xe_blob exporters;  /* struct xe_exporter */
	struct xe_blob sfrecs;     /* struct xe_sfrec */
	size_t nrec;               /* netflow records (sets) */
	size_t nsf;                /* sflow records */
	uint64_t field_mask[2];    /* fields some template of the batch can set */
	size_t soa_bytes_per_rec_sum; /* sum over records of stored widths (algorithmic bytes) */
	size_t skipped_datagrams;  /* unknown version / aborted / oversized */
	size_t oversized_datagrams;/* left out: more than XE_MAX_RECORDS_PER_DATAGRAM records */
	int too_many;              /* the batch would hold more records than 32 bits index */
	size_t unsupported_sets;   /* data sets of templates with more than XE_TMPL_MAXF reachable elements */
	/* exporter (address, source id) -> index: open addressing, index + 1 */
	uint32_t *ehash; size_t ehash_cap;
	/* per device template slot, filled on first use in this batch: fields it
	 * sets and the column bytes a complete record of it fills */
	struct xe_blob tinfo;      /* struct xe_tinfo */
	struct xe_host *pool;      /* not NULL: sets / sfrecs go back to this host's buffer pool */
	/* per datagram i of the batch (ndg + 1 entries each): how many Netflow
	 * records / sFlow records the datagrams before it produced, i.e. datagram i
	 * owns Netflow records [dg_first_nf[i], dg_first_nf[i + 1]) and sFlow records
	 * [dg_first_sf[i], dg_first_sf[i + 1]) */
	uint32_t *dg_first_nf, *dg_first_sf;
	size_t ndg;
};
struct xe_tinfo { uint64_t mask[2]; uint32_t width; uint32_t known; };

/* internal flag on a datagram's kind: the device walks it, the host walk passes it by */
#define XE_KIND_ON_DEVICE 0x80
int xe_plan_note_template(struct xe_plan *plan, const struct xe_host *hs, uint32_t tmpl, size_t nrec);
int xe_prepass(struct xe_host *hs, const uint8_t *data, const uint64_t *offs, const uint32_t *src,
	const uint8_t *kinds, size_t n, struct xe_plan *plan);
void xe_plan_free(struct xe_plan *p);

/* small JSON-with-comments reader used by the config loaders */
/* add_fwm (may be NULL): called for every "fwm" entry of an object right after the object */
int xe_conf_load_mos(struct xe_host *hs, const char *mo_dir,
	int (*add)(void *user, const char *name, int parent, const char *filter),
	int (*add_fwm)(void *user, int mo, const char *name, const char *const *fields, int nfields, int limit, int time),
	void *user);
int xe_conf_load_devices(struct xe_host *hs, const char *path,
	int (*add)(void *user, int use_ip, uint32_t ip, int use_id, uint32_t id, int rate,
		int skip_unmarked, const char *const *marks, int nmarks), void *user);

#ifdef __cplusplus
}
#endif
#endif
