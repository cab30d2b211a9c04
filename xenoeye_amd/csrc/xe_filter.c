/*
 * xe_filter.c -- monitoring-object filter text -> flat postfix program.
 *
 * Accepts the reference's filter language (filter-lexer.c:211-296,
 * filter-parser.c:46-255, filter-parser-funcs.c) and produces the same
 * postfix sequence its parser builds in struct filter_expr, but as three flat
 * arrays (ops, terms, operands) the GPU can index.  Behaviours that differ
 * from what a textbook parser would do are kept on purpose, because the
 * match result must equal the reference's:
 *   - inside a value list, `or` followed by something that is not a value
 *     drops that following token and parses the rest as a new expression
 *     (filter-parser.c:62-83), so `src port 0 or dst port 0` becomes
 *     SRC port 0; BOTH port 0; OR;
 *   - `dev-ip6` is an IPv4-typed term over the first 4 bytes of the 16-byte
 *     field (filter.def:27);
 *   - any 17-character token with ':' at every third position is a MAC
 *     address, non-hex digits read as 0 (utils.c:97-123).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <arpa/inet.h>
#include <math.h>
#include <limits.h>
#include "xe_internal.h"

#define TOK_MAX 512

enum tok
{
	T_ID, T_RANGE, T_MAC, T_STR, T_LP, T_RP, T_COMMA,
	T_OR, T_AND, T_NOT, T_SRC, T_DST, T_ASC, T_DESC,
	T_DIV, T_DIV_R, T_DIV_L, T_MIN, T_TFSTR, T_PORTSTR, T_PPSTR, T_MFREQ, T_ASN, T_ASD,
	T_GEO,       /* val = index into kgeo */
	T_FIELD,     /* val = index into kfields */
	T_AGGR       /* val = index into kaggr */
};

struct kfield { const char *kw; uint8_t type, src, dst; };

/* filter.def:2-37 */
static const struct kfield kfields[] = {
	{ "host", XE_BT_ADDR4, XF_ip4_src_addr, XF_ip4_dst_addr },
	{ "net", XE_BT_ADDR4, XF_ip4_src_addr, XF_ip4_dst_addr },
	{ "host6", XE_BT_ADDR6, XF_ip6_src_addr, XF_ip6_dst_addr },
	{ "net6", XE_BT_ADDR6, XF_ip6_src_addr, XF_ip6_dst_addr },
	{ "port", XE_BT_RANGE, XF_l4_src_port, XF_l4_dst_port },
	{ "proto", XE_BT_RANGE, XF_protocol, XF_protocol },
	{ "tos", XE_BT_RANGE, XF_src_tos, XF_dst_tos },
	{ "tcp-flags", XE_BT_RANGE, XF_tcp_flags, XF_tcp_flags },
	{ "ifidx", XE_BT_RANGE, XF_input_snmp, XF_output_snmp },
	{ "as", XE_BT_RANGE, XF_src_as, XF_dst_as },
	{ "min-ttl", XE_BT_RANGE, XF_min_ttl, XF_min_ttl },
	{ "max-ttl", XE_BT_RANGE, XF_max_ttl, XF_max_ttl },
	{ "frag-id", XE_BT_RANGE, XF_frag_id, XF_frag_id },
	{ "mac", XE_BT_MAC, XF_src_mac, XF_dst_mac },
	{ "vlan", XE_BT_RANGE, XF_src_vlan, XF_dst_vlan },
	{ "endreason", XE_BT_RANGE, XF_flow_end_reason, XF_flow_end_reason },
	{ "d1qvlan", XE_BT_RANGE, XF_src_dot1q_vlan, XF_dst_dot1q_vlan },
	{ "d1qcvlan", XE_BT_RANGE, XF_src_dot1q_cvlan, XF_dst_dot1q_cvlan },
	{ "bgp-nh", XE_BT_ADDR4, XF_bgp_next_hop, XF_bgp_next_hop },
	{ "dir", XE_BT_RANGE, XF_direction, XF_direction },
	{ "fwdst", XE_BT_RANGE, XF_fwd_status, XF_fwd_status },
	{ "smplr", XE_BT_RANGE, XF_sampler_id, XF_sampler_id },
	{ "vrf", XE_BT_RANGE, XF_ingrs_vrf, XF_egrs_vrf },
	{ "vashost", XE_BT_STRING, XF_vas_http_host, XF_vas_http_host },
	{ "dev-ip", XE_BT_ADDR4, XF_dev_ip, XF_dev_ip },
	{ "dev-ip6", XE_BT_ADDR4, XF_dev_ip6, XF_dev_ip6 },
	{ "dev-id", XE_BT_RANGE, XF_dev_id, XF_dev_id },
	{ "dev-mark", XE_BT_RANGE, XF_dev_mark, XF_dev_mark },
	{ "class0", XE_BT_STRING, XF_class0, XF_class0 },
	{ "class1", XE_BT_STRING, XF_class1, XF_class1 },
	{ "class2", XE_BT_STRING, XF_class2, XF_class2 },
	{ "class3", XE_BT_STRING, XF_class3, XF_class3 },
	{ "class4", XE_BT_STRING, XF_class4, XF_class4 },
	{ "dns-name", XE_BT_STRING, XF_dns_name, XF_dns_name },
	{ "dns-ips", XE_BT_STRING, XF_dns_ips, XF_dns_ips },
	{ "sni", XE_BT_STRING, XF_sni, XF_sni },
};
#define NKFIELDS ((int)(sizeof(kfields) / sizeof(kfields[0])))

/* filter-ag.def:2-8 */
static const struct { const char *kw; uint8_t field; uint8_t scale; } kaggr[] = {
	{ "octets", XF_in_bytes, 1 }, { "bits", XF_in_bytes, 8 }, { "packets", XF_in_pkts, 1 },
	{ "ioctets", XF_ioctets, 1 }, { "roctets", XF_roctets, 1 }, { "ipackets", XF_ipackets, 1 },
	{ "rpackets", XF_rpackets, 1 },
};
#define NKAGGR ((int)(sizeof(kaggr) / sizeof(kaggr[0])))

/* the two tables above row by row, for checks against filter.def / filter-ag.def */
int
xe_filter_keyword(int i, const char **kw, int *type, int *src, int *dst)
{
	if ((i < 0) || (i >= NKFIELDS)) return 0;
	*kw = kfields[i].kw;
	*type = kfields[i].type;
	*src = kfields[i].src;
	*dst = kfields[i].dst;
	return 1;
}

int
xe_filter_aggr_keyword(int i, const char **kw, int *field, int *scale)
{
	if ((i < 0) || (i >= NKAGGR)) return 0;
	*kw = kaggr[i].kw;
	*field = kaggr[i].field;
	*scale = kaggr[i].scale;
	return 1;
}

static const struct { const char *kw; enum tok t; } kwords[] = {
	{ "or", T_OR }, { "and", T_AND }, { "not", T_NOT }, { "asc", T_ASC }, { "desc", T_DESC },
	{ "div", T_DIV }, { "div_r", T_DIV_R }, { "div_l", T_DIV_L }, { "min", T_MIN },
	{ "mfreq", T_MFREQ }, { "tfstr", T_TFSTR }, { "portstr", T_PORTSTR }, { "ppstr", T_PPSTR },
	{ "asn", T_ASN }, { "asd", T_ASD },
};

/* geoip.h:13-21, in the order of enum GEOIP_FIELD */
static const char *const kgeo[] = { "continent", "country_code", "country", "state", "city", "zip", "lat", "long" };
#define NKGEO ((int)(sizeof(kgeo) / sizeof(kgeo[0])))

struct lexer
{
	const char *s;
	int at_end, failed;
	int line, col;
	char msg[600];
	enum tok t;
	int val;
	char text[TOK_MAX];
	int lo, hi;
	uint8_t mac[6];
};

struct pctx
{
	struct lexer lx;
	const struct xe_iplists *reg;
	struct xe_blob ops, terms, opnds, notes, strs;
	int device_ok;
	int needs_port_names;
	int needs_geodb;
	char why[128];
};

static void
fail(struct lexer *lx, const char *m)
{
	lx->failed = 1;
	snprintf(lx->msg, sizeof(lx->msg), "Line %d, col %d: %s", lx->line, lx->col, m);
}

static int
read_mac(const char *s, uint8_t out[6])
{
	int i;
	uint8_t m[6];

	if (strlen(s) != 17) return 0;
	for (i = 0; i < 6; i++) {
		int hv[2], k;
		char sep = s[3 * i + 2];
		if (sep && (sep != ':')) return 0;
		for (k = 0; k < 2; k++) {
			char c = s[3 * i + k];
			hv[k] = ((c >= '0') && (c <= '9')) ? c - '0'
				: ((c >= 'a') && (c <= 'f')) ? c - 'a' + 10
				: ((c >= 'A') && (c <= 'F')) ? c - 'A' + 10 : 0;
		}
		m[i] = (uint8_t)(hv[0] * 16 + hv[1]);
	}
	memcpy(out, m, 6);
	return 1;
}

/* blanks, "//" and "/ * * /" comments: filter-lexer.c:43-120, including its
 * comment scanner that tests the character after a '*' only once */
static void
skip_blank(struct lexer *lx)
{
	const char *s = lx->s;

	for (;; s++) {
		char c = *s;
		if (!c) { lx->at_end = 1; break; }
		if ((c == ' ') || (c == '\t')) { lx->col++; continue; }
		if ((c == '\n') || (c == '\r')) { lx->line++; lx->col = 1; continue; }
		if (c != '/') break;
		if (!s[1]) {
			lx->at_end = lx->failed = 1;
			strcpy(lx->msg, "unexpected end of input after '/'");
			break;
		}
		if (s[1] == '/') {
			s++;
			lx->col++;
			do {
				s++;
				if (!*s) { lx->at_end = 1; lx->s = s; return; }
				lx->col++;
			} while (*s != '\n');
			lx->line++;
			lx->col = 1;
			continue;
		}
		if (s[1] == '*') {
			int closed = 0;
			s++;
			lx->col++;
			while (!closed) {
				s++;
				if (!*s) goto open_comment;
				lx->col++;
				if (*s == '*') {
					s++;
					if (!*s) goto open_comment;
					lx->col++;
					closed = (*s == '/');
				} else if (*s == '\n') {
					lx->line++;
					lx->col = 1;
				}
			}
			continue;
		}
		break;   /* a lone '/' starts a token */
	}
	lx->s = s;
	return;
open_comment:
	lx->at_end = lx->failed = 1;
	strcpy(lx->msg, "unexpected end of input inside the comment");
	lx->s = s;
}

static int
ends_token(char c)
{
	return !c || (c == ' ') || (c == '\t') || (c == ',') || (c == '\r') || (c == '\n')
		|| (c == '(') || (c == ')');
}

static void
classify_word(struct lexer *lx)
{
	const char *w = lx->text;
	char *e1, *e2;
	long a, b;
	size_t i;

	if (read_mac(w, lx->mac)) { lx->t = T_MAC; return; }
	/* keyword order of filter-lexer.c:142-209 */
	if (!strcasecmp(w, "src")) { lx->t = T_SRC; return; }
	if (!strcasecmp(w, "dst")) { lx->t = T_DST; return; }
	for (i = 0; i < (size_t)NKFIELDS; i++) {
		if (!strcasecmp(w, kfields[i].kw)) { lx->t = T_FIELD; lx->val = (int)i; return; }
	}
	for (i = 0; i < sizeof(kwords) / sizeof(kwords[0]); i++) {
		if (!strcasecmp(w, kwords[i].kw)) { lx->t = kwords[i].t; return; }
	}
	for (i = 0; i < (size_t)NKGEO; i++) {
		if (!strcasecmp(w, kgeo[i])) { lx->t = T_GEO; lx->val = (int)i; return; }
	}
	for (i = 0; i < (size_t)NKAGGR; i++) {
		if (!strcasecmp(w, kaggr[i].kw)) { lx->t = T_AGGR; lx->val = (int)i; return; }
	}
	/* number, a-b range, else identifier (filter-lexer.c:268-293) */
	a = strtol(w, &e1, 0);
	if (!*e1) {
		lx->t = T_RANGE;
		lx->lo = lx->hi = (int)a;
		return;
	}
	if (*e1 == '-') {
		b = strtol(e1 + 1, &e2, 0);
		if (!*e2) {
			lx->t = T_RANGE;
			lx->lo = (int)a;
			lx->hi = (int)b;
			return;
		}
	}
	lx->t = T_ID;
}

static void
next_token(struct lexer *lx)
{
	int n = 0;

	skip_blank(lx);
	if (lx->at_end || lx->failed) return;   /* the previous token stays current */

	if ((*lx->s == '(') || (*lx->s == ')') || (*lx->s == ',')) {
		lx->t = (*lx->s == '(') ? T_LP : (*lx->s == ')') ? T_RP : T_COMMA;
		lx->text[0] = *lx->s;
		lx->text[1] = '\0';
		lx->s++;
		lx->col++;
		return;
	}
	if (*lx->s == '\'') {
		/* at least one character is taken before the closing quote is
		 * looked for (filter-lexer.c:229-237) */
		lx->s++;
		lx->col++;
		do {
			if (!*lx->s || (n >= TOK_MAX - 1)) {
				lx->at_end = lx->failed = 1;
				strcpy(lx->msg, "unterminated string");
				return;
			}
			lx->text[n++] = *lx->s++;
			lx->col++;
		} while (*lx->s != '\'');
		lx->s++;
		lx->col++;
		lx->text[n] = '\0';
		lx->t = T_STR;
		return;
	}
	do {
		if (n >= TOK_MAX - 1) {
			lx->at_end = lx->failed = 1;
			strcpy(lx->msg, "token too long");
			return;
		}
		lx->text[n++] = *lx->s++;
		lx->col++;
	} while (!ends_token(*lx->s));
	lx->text[n] = '\0';
	classify_word(lx);
}

/* token matches -> consume it; a lexer error while fetching the next token
 * makes the match fail too (filter-parser.c:23-33) */
static int
take(struct pctx *p, enum tok t)
{
	if (p->lx.t != t) return 0;
	next_token(&p->lx);
	return !p->lx.failed;
}

static int
take_field(struct pctx *p, int *which)
{
	if (p->lx.t != T_FIELD) return 0;
	*which = p->lx.val;
	next_token(&p->lx);
	return !p->lx.failed;
}

static void
not_on_device(struct pctx *p, const char *why)
{
	if (p->device_ok) {
		p->device_ok = 0;
		snprintf(p->why, sizeof(p->why), "%s", why);
	}
}

static int
emit_op(struct pctx *p, uint32_t op, uint32_t term)
{
	uint32_t w = op | (term << 2);
	return xe_blob_push(&p->ops, &w, 4);
}

static int
emit_term(struct pctx *p, const struct xe_term *t)
{
	uint32_t idx = (uint32_t)p->terms.n;
	struct xe_term tt = *t;
	tt.nopnd = 0;
	tt.opnd_base = (uint32_t)p->opnds.n;
	if (!xe_blob_push(&p->terms, &tt, sizeof(tt))) return 0;
	return emit_op(p, XE_OP_BASIC, idx);
}

/* address text -> masked address + mask, bytes in network order (filter.c:86-156) */
static int
parse_addr(struct pctx *p, const char *txt, int v6, struct xe_opnd *o, int32_t *masklen)
{
	char tmp[TOK_MAX];
	char *slash;
	uint8_t a[16], m[16];
	int nb = v6 ? 16 : 4, i, ml = -1;

	snprintf(tmp, sizeof(tmp), "%s", txt);
	slash = strchr(tmp, '/');
	if (slash) {
		char *e;
		*slash++ = '\0';
		ml = (int)strtol(slash, &e, 10);
		if (*e) { fail(&p->lx, "Incorrect network mask"); return 0; }
	}
	if (inet_pton(v6 ? AF_INET6 : AF_INET, tmp, a) != 1) {
		fail(&p->lx, "Can't parse IP address");
		return 0;
	}
	if (ml < 0) ml = nb * 8;
	*masklen = ml;
	memset(m, 0, sizeof(m));
	for (i = 0; (i < ml) && (i < nb * 8); i++) m[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
	for (i = 0; i < nb; i++) a[i] &= m[i];
	memset(o, 0, sizeof(*o));
	if (v6) {
		memcpy(&o->w[0], a, 16);
		memcpy(&o->w[4], m, 16);
	} else {
		memcpy(&o->w[0], a, 4);
		memcpy(&o->w[1], m, 4);
	}
	return 1;
}

/* append the current token as an operand of the last term (filter.c:158-232) */
static void
add_operand(struct pctx *p, int type)
{
	struct xe_opnd o;
	int32_t note = 0;
	struct xe_term *t;

	if (!p->terms.n) return;
	memset(&o, 0, sizeof(o));
	switch (p->lx.t) {
		case T_ID:
			if (type == XE_BT_STRING) note = -1;
			if ((type == XE_BT_ADDR4) || (type == XE_BT_ADDR6)) {
				int li = p->reg ? xe_iplist_find(p->reg, p->lx.text) : -1;
				if (li >= 0) {
					o.is_list = 1;
					o.w[0] = (uint32_t)li;
					note = li;
				} else if (!parse_addr(p, p->lx.text, type == XE_BT_ADDR6, &o, &note)) {
					return;   /* error is set; the operand is not counted */
				}
			} else {
				/* a bare word under a range / mac / string rule: the
				 * reference stores an operand it never initialises */
				not_on_device(p, "identifier used where a number, MAC or string is expected");
			}
			break;
		case T_RANGE:
			if (type != XE_BT_RANGE) not_on_device(p, "number used where an address, MAC or string is expected");
			if (type == XE_BT_STRING) note = -1;
			o.w[0] = (uint32_t)p->lx.lo;
			o.w[1] = (uint32_t)p->lx.hi;
			break;
		case T_MAC:
			if (type != XE_BT_MAC) not_on_device(p, "MAC address used where another operand type is expected");
			if (type == XE_BT_STRING) note = -1;
			memcpy(&o.w[0], p->lx.mac, 6);
			break;
		case T_STR: {
			const uint8_t fn = ((struct xe_term *)p->terms.p)[p->terms.n - 1].func;
			size_t k, len = strlen(p->lx.text) + 1;
			/* the text is kept for the dump (and for the lowering of tfstr /
			 * portstr / ppstr at commit) */
			note = (int32_t)p->strs.n;
			for (k = 0; k < len; k++) if (!xe_blob_push(&p->strs, &p->lx.text[k], 1)) return;
			if (fn == XE_FN_TFSTR) {
				snprintf((char *)&o.w[0], sizeof(o.w), "%s", p->lx.text);
			} else if ((fn == XE_FN_PORTSTR) || (fn == XE_FN_PPSTR) || (fn == XE_FN_ASD) || (fn == XE_FN_GEO)) {
				/* compared on the host when the term is lowered at commit */
				if (type != XE_BT_STRING) not_on_device(p, "string operand");
			} else {
				not_on_device(p, "string operand");
			}
			break;
		}
		default:
			return;
	}
	if (!xe_blob_push(&p->opnds, &o, sizeof(o))) return;
	xe_blob_push(&p->notes, &note, sizeof(note));
	t = &((struct xe_term *)p->terms.p)[p->terms.n - 1];
	t->nopnd++;
}

static int is_value(enum tok t) { return (t == T_ID) || (t == T_RANGE) || (t == T_MAC) || (t == T_STR); }

static int parse_expr(struct pctx *p);

/* value (`or` value)* with the token-dropping `or` (filter-parser.c:46-86) */
static int
value_list(struct pctx *p, int type)
{
	if (!is_value(p->lx.t)) {
		fail(&p->lx, "Expected ID, INT, INT_RANGE, MAC or STRING");
		return 0;
	}
	add_operand(p, type);
	next_token(&p->lx);
	if (p->lx.failed) return 0;

	while (take(p, T_OR)) {
		if (is_value(p->lx.t)) {
			add_operand(p, type);
			next_token(&p->lx);
			if (p->lx.failed) return 0;
			continue;
		}
		next_token(&p->lx);          /* the token after `or` is dropped */
		if (p->lx.failed) return 0;
		parse_expr(p);               /* its result is not checked */
		emit_op(p, XE_OP_OR, 0);
	}
	return 1;
}

static int
plain_rule(struct pctx *p, int dir)
{
	int k;
	struct xe_term t;

	if (!take_field(p, &k)) return 0;
	memset(&t, 0, sizeof(t));
	t.type = kfields[k].type;
	t.dir = (uint8_t)dir;
	t.fsrc = kfields[k].src;
	t.fdst = kfields[k].dst;
	t.pad = (uint8_t)k;              /* keyword index, for the dump */
	emit_term(p, &t);
	if (t.type == XE_BT_STRING) not_on_device(p, "string field");
	return value_list(p, t.type);
}

static int
aggr_arg(struct pctx *p, uint8_t *field)
{
	if (p->lx.t != T_AGGR) return 0;
	*field = kaggr[p->lx.val].field;
	next_token(&p->lx);
	return !p->lx.failed;
}

/* [src|dst] field as a function argument (filter-parser-funcs.c:55-79,:194-213) */
static int
field_arg(struct pctx *p, uint8_t *field)
{
	int dir = XE_DIR_BOTH, k;

	if (take(p, T_SRC)) dir = XE_DIR_SRC;
	else if (take(p, T_DST)) dir = XE_DIR_DST;
	if (!take_field(p, &k)) return 0;
	if (dir == XE_DIR_SRC) {
		*field = kfields[k].src;
	} else if (dir == XE_DIR_DST) {
		*field = kfields[k].dst;
	} else {
		if (kfields[k].src != kfields[k].dst) {
			fail(&p->lx, "This field requires direction");
			return 0;
		}
		*field = kfields[k].src;
	}
	return *field < XE_NTABLE_FIELDS;   /* virtual fields have no table entry */
}

/* tfstr(field) 'FLAGS' ... (filter-parser-funcs.c:526-580): the byte that
 * the reference reads (filter.c:842-847) against the text of each operand */
static int
tfstr_rule(struct pctx *p)
{
	struct xe_term t;

	if (p->lx.t != T_TFSTR) return 0;
	if (!take(p, T_TFSTR)) return 0;
	memset(&t, 0, sizeof(t));
	t.type = XE_BT_STRING;
	t.dir = XE_DIR_NONE;
	t.func = XE_FN_TFSTR;
	if (!take(p, T_LP)) { fail(&p->lx, "Expected '(' after 'tfstr'"); return 0; }
	if (!field_arg(p, &t.fa)) { fail(&p->lx, "Incorrect field name"); return 0; }
	if (!take(p, T_RP)) { fail(&p->lx, "Expected ')'"); return 0; }
	if (xe_fields[t.fa].type == XE_T_STR) not_on_device(p, "tfstr() over a string field");
	emit_term(p, &t);
	return value_list(p, XE_BT_STRING);
}

/* div / div_r / div_l / min (filter-parser-funcs.c:83-282) */
static int
func_rule(struct pctx *p)
{
	struct xe_term t;
	enum tok which = p->lx.t;

	if ((which != T_DIV) && (which != T_DIV_R) && (which != T_DIV_L) && (which != T_MIN)) return 0;
	if (!take(p, which)) return 0;
	memset(&t, 0, sizeof(t));
	t.type = XE_BT_RANGE;
	t.dir = XE_DIR_NONE;
	t.k = 1;
	if (!take(p, T_LP)) { fail(&p->lx, "Expected '(' after function name"); return 0; }
	if (which == T_MIN) {
		t.func = XE_FN_MIN;
		if (!field_arg(p, &t.fa)) { fail(&p->lx, "Incorrect field name"); return 0; }
		if (!take(p, T_COMMA)) { fail(&p->lx, "Expected ',' after field name"); return 0; }
		if (!field_arg(p, &t.fb)) { fail(&p->lx, "Incorrect field name after comma"); return 0; }
	} else {
		t.func = (which == T_DIV) ? XE_FN_DIV : (which == T_DIV_R) ? XE_FN_DIV_R : XE_FN_DIV_L;
		if (!aggr_arg(p, &t.fa)) { fail(&p->lx, "Expected aggregable field name"); return 0; }
		if (!take(p, T_COMMA)) { fail(&p->lx, "Expected ',' after field name"); return 0; }
		if (!aggr_arg(p, &t.fb)) { fail(&p->lx, "Expected aggregable field name after comma"); return 0; }
		if (which != T_DIV) {
			int k;
			if (!take(p, T_COMMA)) { fail(&p->lx, "Expected ',' after second field name"); return 0; }
			k = p->lx.lo;
			if (!take(p, T_RANGE)) { fail(&p->lx, "Expected INT"); return 0; }
			t.k = k;
		}
	}
	if (!take(p, T_RP)) { fail(&p->lx, "Expected ')'"); return 0; }
	emit_term(p, &t);
	return value_list(p, XE_BT_RANGE);
}

/* mfreq(f1, f2), portstr(f), ppstr(f1, f2), the geo functions, asn(f), asd(f)
 * (filter-parser-funcs.c:285-524, :582-705).  portstr / ppstr are pure
 * functions of 16-bit values and are lowered at commit when the handle has a
 * port-name table; the others keep state or read the geo databases and stay
 * with the caller's own filter_match(). */
static int
other_func_rule(struct pctx *p)
{
	struct xe_term t;
	const enum tok which = p->lx.t;
	const char *name;
	int two_args = 0, type = XE_BT_STRING;

	memset(&t, 0, sizeof(t));
	t.dir = XE_DIR_NONE;
	switch (which) {
		case T_MFREQ: t.func = XE_FN_MFREQ; name = "'mfreq'"; two_args = 1; type = XE_BT_RANGE; break;
		case T_PORTSTR: t.func = XE_FN_PORTSTR; name = "'portstr'"; break;
		case T_PPSTR: t.func = XE_FN_PPSTR; name = "'ppstr'"; two_args = 1; break;
		case T_GEO: t.func = XE_FN_GEO; t.k = p->lx.val; name = "GeoIP function"; break;
		case T_ASN: t.func = XE_FN_ASN; name = "AS function"; type = XE_BT_RANGE; break;
		case T_ASD: t.func = XE_FN_ASD; name = "AS function"; break;
		default: return 0;
	}
	t.type = (uint8_t)type;
	if (!take(p, which)) return 0;
	if (!take(p, T_LP)) {
		char m[64];
		snprintf(m, sizeof(m), "Expected '(' after %s", name);
		fail(&p->lx, m);
		return 0;
	}
	if (!field_arg(p, &t.fa)) { fail(&p->lx, "Incorrect field name"); return 0; }
	if (two_args) {
		if (!take(p, T_COMMA)) { fail(&p->lx, "Expected ',' after field name"); return 0; }
		if (!field_arg(p, &t.fb)) { fail(&p->lx, "Incorrect field name after comma"); return 0; }
	}
	if (!take(p, T_RP)) { fail(&p->lx, "Expected ')'"); return 0; }
	if ((which == T_PORTSTR) || (which == T_PPSTR)) {
		p->needs_port_names = 1;
		/* the reference reads get_nf_val() of the field: 1, 2, 4 or 8 bytes */
		if ((xe_fields[t.fa].type == XE_T_STR) || (two_args && (xe_fields[t.fb].type == XE_T_STR))) {
			not_on_device(p, "portstr() / ppstr() over a string field");
		}
	} else if (which == T_MFREQ) {
		not_on_device(p, "mfreq() keeps state between records");
	} else if ((xe_fields[t.fa].type != XE_T_STR) && ((xe_fields[t.fa].size_max == 4) || (xe_fields[t.fa].size_max == 16))) {
		/* a lookup of the field's 4 or 16 bytes in the geo / AS database:
		 * lowered at commit to a list made from the database */
		p->needs_geodb |= 1 << (2 * (which != T_GEO) + (xe_fields[t.fa].size_max == 16));
	} else {
		/* the reference reads a result it never set (filter.c:769-783, :800-816) */
		not_on_device(p, "geo / AS function over a field that is not 4 or 16 bytes long");
	}
	emit_term(p, &t);
	return value_list(p, type);
}

static int
parse_rule(struct pctx *p)
{
	if (func_rule(p)) return 1;
	if (p->lx.failed) return 0;
	if (tfstr_rule(p)) return 1;
	if (p->lx.failed) return 0;
	if (other_func_rule(p)) return 1;
	if (p->lx.failed) return 0;
	if (plain_rule(p, XE_DIR_BOTH)) return 1;
	if (take(p, T_SRC)) return plain_rule(p, XE_DIR_SRC);
	if (take(p, T_DST)) return plain_rule(p, XE_DIR_DST);
	return 0;
}

static int
parse_factor(struct pctx *p)
{
	int neg = take(p, T_NOT);

	if (parse_rule(p)) {
		/* done */
	} else if (take(p, T_LP)) {
		if (p->lx.at_end) {
			/* '(' was the last token and stays current: the reference
			 * recurses without bound here; this is an error instead */
			fail(&p->lx, "Unexpected end of input after '('");
			return 0;
		}
		if (!parse_expr(p)) return 0;
		if (!take(p, T_RP)) {
			fail(&p->lx, "Expected ')' after expression");
			return 0;
		}
	} else {
		fail(&p->lx, "Syntax error");
		return 0;
	}
	if (neg) emit_op(p, XE_OP_NOT, 0);
	return 1;
}

static int
parse_term(struct pctx *p)
{
	if (!parse_factor(p)) return 0;
	while (take(p, T_AND)) {
		if (!parse_factor(p)) return 0;
		emit_op(p, XE_OP_AND, 0);
	}
	return 1;
}

static int
parse_expr(struct pctx *p)
{
	if (!parse_term(p)) return 0;
	while (take(p, T_OR)) {
		if (!parse_term(p)) return 0;
		emit_op(p, XE_OP_OR, 0);
	}
	return 1;
}

/* One field of a fixed-window table, `[src|dst] name [asc|desc]` or an
 * aggregable name, or a function form (parse_field(), filter-parser.c:396-484).
 * mfreq, the geo functions, asn and asd are recognised and reported as
 * XE_EUNSUPPORTED. */
int
xe_field_parse(const char *text, struct xe_agfield *out, char *err, size_t errlen)
{
	struct pctx p;
	int k, dir = XE_DIR_BOTH;

	memset(&p, 0, sizeof(p));
	memset(out, 0, sizeof(*out));
	p.lx.s = text;
	p.lx.t = T_ID;
	next_token(&p.lx);
	if (p.lx.failed) goto bad;
	if (p.lx.at_end) { fail(&p.lx, "Empty field not allowed"); goto bad; }
	switch (p.lx.t) {
		case T_MFREQ: case T_ASN: case T_ASD: case T_GEO:
			/* mfreq() keeps state from record to record, the others put database
			 * payload into the key */
			if (err && errlen) snprintf(err, errlen, "mfreq, geo and AS function fields are not supported by this library");
			return XE_EUNSUPPORTED;
		default:
			break;
	}
	if ((p.lx.t == T_DIV) || (p.lx.t == T_DIV_R) || (p.lx.t == T_DIV_L) || (p.lx.t == T_MIN) || (p.lx.t == T_TFSTR)
		|| (p.lx.t == T_PORTSTR) || (p.lx.t == T_PPSTR)) {
		/* the function forms (function_div_parse() ... function_ppstr_parse(),
		 * filter-parser-funcs.c:83-282, :526-705): the same argument lists as in a
		 * filter, no values behind them */
		const enum tok which = p.lx.t;
		const int two = (which != T_TFSTR) && (which != T_PORTSTR);
		const int aggr_args = (which == T_DIV) || (which == T_DIV_R) || (which == T_DIV_L);
		(void)take(&p, which);
		out->k = aggr_args ? 1 : 0;
		out->func = (which == T_DIV) ? XE_FN_DIV : (which == T_DIV_R) ? XE_FN_DIV_R : (which == T_DIV_L) ? XE_FN_DIV_L
			: (which == T_MIN) ? XE_FN_MIN : (which == T_TFSTR) ? XE_FN_TFSTR : (which == T_PORTSTR) ? XE_FN_PORTSTR : XE_FN_PPSTR;
		out->size = aggr_args || (which == T_MIN) ? 8 : (which == T_TFSTR) ? 32 : (which == T_PORTSTR) ? 40 : 90;
		if (!take(&p, T_LP)) { fail(&p.lx, "Expected '(' after function name"); goto bad; }
		if (!(aggr_args ? aggr_arg(&p, &out->field) : field_arg(&p, &out->field))) {
			fail(&p.lx, aggr_args ? "Expected aggregable field name" : "Incorrect field name");
			goto bad;
		}
		if (two) {
			if (!take(&p, T_COMMA)) { fail(&p.lx, "Expected ',' after field name"); goto bad; }
			if (!(aggr_args ? aggr_arg(&p, &out->fb) : field_arg(&p, &out->fb))) {
				fail(&p.lx, aggr_args ? "Expected aggregable field name after comma" : "Incorrect field name after comma");
				goto bad;
			}
		}
		if ((which == T_DIV_R) || (which == T_DIV_L)) {
			int k;
			if (!take(&p, T_COMMA)) { fail(&p.lx, "Expected ',' after second field name"); goto bad; }
			k = p.lx.lo;
			if (!take(&p, T_RANGE)) { fail(&p.lx, "Expected INT"); goto bad; }
			out->k = k;
		}
		if (!take(&p, T_RP)) { fail(&p.lx, "Expected ')'"); goto bad; }
		if (p.lx.failed) goto bad;
		if (!p.lx.at_end && take(&p, T_DESC)) out->descending = 1;
		else if (!p.lx.at_end) (void)take(&p, T_ASC);
		if (p.lx.failed) goto bad;
		if (!p.lx.at_end) { fail(&p.lx, "Extra symbols after field name"); goto bad; }
		if ((xe_fields[out->field].type == XE_T_STR) || (two && (xe_fields[out->fb].type == XE_T_STR))) {
			if (err && errlen) snprintf(err, errlen, "function fields over string fields are not supported as table keys by this library");
			return XE_EUNSUPPORTED;
		}
		return XE_OK;
	}
	if (p.lx.t == T_AGGR) {
		out->aggr = 1;
		out->field = kaggr[p.lx.val].field;
		out->scale = kaggr[p.lx.val].scale;
		out->size = 8;
		next_token(&p.lx);
	} else {
		if (take(&p, T_SRC)) dir = XE_DIR_SRC;
		else if (take(&p, T_DST)) dir = XE_DIR_DST;
		if (p.lx.failed) goto bad;
		if (p.lx.t != T_FIELD) {
			char m[600];
			snprintf(m, sizeof(m), "Unknown field '%.400s'", p.lx.text);
			fail(&p.lx, m);
			goto bad;
		}
		k = p.lx.val;
		if ((dir == XE_DIR_BOTH) && (kfields[k].src != kfields[k].dst)) {
			fail(&p.lx, "This field requires direction");
			goto bad;
		}
		out->field = (dir == XE_DIR_DST) ? kfields[k].dst : kfields[k].src;
		/* the three virtual fields have no entry in the field table the
		 * reference maps names through (filter-parser.c:256-271) */
		if (out->field >= XE_NTABLE_FIELDS) { fail(&p.lx, "Unknown field"); goto bad; }
		out->size = xe_fields[out->field].size_max;
		next_token(&p.lx);
	}
	if (p.lx.failed) goto bad;
	if (!p.lx.at_end && take(&p, T_DESC)) out->descending = 1;
	else if (!p.lx.at_end) (void)take(&p, T_ASC);
	if (p.lx.failed) goto bad;
	if (!p.lx.at_end) { fail(&p.lx, "Extra symbols after field name"); goto bad; }
	if (!out->aggr && (xe_fields[out->field].type == XE_T_STR)) {
		if (err && errlen) snprintf(err, errlen, "string fields are not supported as table keys by this library");
		return XE_EUNSUPPORTED;
	}
	return XE_OK;
bad:
	if (err && errlen) snprintf(err, errlen, "%s", p.lx.msg);
	return XE_EPARSE;
}

void
xe_filter_free(struct xe_filter *f)
{
	free(f->ops);
	free(f->terms);
	free(f->opnds);
	free(f->opnd_note);
	free(f->strs);
	memset(f, 0, sizeof(*f));
}

int
xe_filter_compile(const struct xe_iplists *reg, const char *text, struct xe_filter *out,
	char *err, size_t errlen)
{
	struct pctx p;

	memset(&p, 0, sizeof(p));
	memset(out, 0, sizeof(*out));
	p.lx.s = text;
	p.lx.t = T_ID;
	p.reg = reg;
	p.device_ok = 1;

	next_token(&p.lx);
	if (!p.lx.failed && !p.lx.at_end) {
		if (parse_expr(&p) && !p.lx.at_end) {
			char m[600];
			snprintf(m, sizeof(m), "Unexpected token '%.400s' after expression", p.lx.text);
			fail(&p.lx, m);
		}
	}
	out->ops = p.ops.p;       out->nops = p.ops.n;
	out->terms = p.terms.p;   out->nterms = p.terms.n;
	out->opnds = p.opnds.p;   out->nopnds = p.opnds.n;
	out->opnd_note = p.notes.p;
	out->strs = p.strs.p;     out->nstrs = p.strs.n;
	out->device_ok = p.device_ok;
	out->needs_port_names = p.needs_port_names;
	out->needs_geodb = p.needs_geodb;
	memcpy(out->why, p.why, sizeof(out->why));
	if (p.lx.failed) {
		xe_filter_free(out);
		if (err && errlen) snprintf(err, errlen, "%s", p.lx.msg);
		return XE_EPARSE;
	}
	return XE_OK;
}

/* The reference evaluates with an explicit stack and answers 0 on underflow
 * or when the final depth is not 1, and 1 for an empty program
 * (filter.c:960-1012).  Depth is a property of the program alone, so it is
 * resolved here once instead of per record. */
int
xe_filter_constant(const struct xe_filter *f, int *cval)
{
	size_t i, sp = 0;

	if (f->nops == 0) { *cval = 1; return 1; }
	for (i = 0; i < f->nops; i++) {
		switch (f->ops[i] & 3) {
			case XE_OP_BASIC: sp++; break;
			case XE_OP_NOT: if (sp < 1) { *cval = 0; return 1; } break;
			default: if (sp < 2) { *cval = 0; return 1; } sp--; break;
		}
	}
	if (sp != 1) { *cval = 0; return 1; }
	return 0;
}

/* text form of filter_dump() (filter.c:1069-1256) */
int
xe_filter_dump_text(const struct xe_iplists *reg, const struct xe_filter *f, char *out, size_t outlen)
{
	size_t i, j, pos = 0;
	int k;

#define PUT(...) do { int n_ = snprintf(out + pos, (pos < outlen) ? outlen - pos : 0, __VA_ARGS__); \
	if (n_ < 0) return XE_EINVAL; pos += (size_t)n_; } while (0)

	if (outlen) out[0] = '\0';
	for (i = 0; i < f->nops; i++) {
		uint32_t op = f->ops[i] & 3;
		const struct xe_term *t;

		if (op == XE_OP_NOT) { PUT("NOT\n"); continue; }
		if (op == XE_OP_AND) { PUT("AND\n"); continue; }
		if (op == XE_OP_OR) { PUT("OR\n"); continue; }
		t = &f->terms[f->ops[i] >> 2];
		if (t->dir == XE_DIR_SRC) PUT("SRC ");
		else if (t->dir == XE_DIR_DST) PUT("DST ");
		else if (t->dir == XE_DIR_BOTH) PUT("SRC OR DST ");
		switch (t->func) {
			case XE_FN_DIV:
				PUT("DIV ([offset %d]/[offset %d])", xe_fields[t->fa].off, xe_fields[t->fb].off);
				break;
			case XE_FN_DIV_R:
				PUT("DIV_R ([offset %d]/[offset %d]/k %d)", xe_fields[t->fa].off, xe_fields[t->fb].off, t->k);
				break;
			case XE_FN_DIV_L:
				PUT("DIV_L ([offset %d]/[offset %d]/k %d)", xe_fields[t->fa].off, xe_fields[t->fb].off, t->k);
				break;
			case XE_FN_MIN:
				PUT("MIN ([offset %d]/[offset %d])", xe_fields[t->fa].off, xe_fields[t->fb].off);
				break;
			case XE_FN_MFREQ:
				PUT("MFREQ ([offset %d]/[offset %d])", xe_fields[t->fa].off, xe_fields[t->fb].off);
				break;
			case XE_FN_GEO: {
				/* the macro prints the enumerator's name (filter.c:1154-1160) */
				char up[16];
				size_t c;
				for (c = 0; kgeo[t->k][c] && (c < sizeof(up) - 1); c++) {
					up[c] = (char)(((kgeo[t->k][c] >= 'a') && (kgeo[t->k][c] <= 'z')) ? kgeo[t->k][c] - 32 : kgeo[t->k][c]);
				}
				up[c] = '\0';
				PUT("%s ([offset %d]/[size %d])", up, xe_fields[t->fa].off, xe_fields[t->fa].size_max);
				break;
			}
			case XE_FN_ASN:
			case XE_FN_ASD:
				PUT("AS ([offset %d]/[size %d])", xe_fields[t->fa].off, xe_fields[t->fa].size_max);
				break;
			case XE_FN_TFSTR:
			case XE_FN_PORTSTR:
			case XE_FN_PPSTR:
				/* filter.c:1113-1172 has no case for them: the default branch
				 * prints the enum value (filter.h:130-151) */
				PUT("<Unknown name %d> ", 1 + NKFIELDS + 5 + 8 + 2 + (int)(t->func - XE_FN_TFSTR));
				break;
			default:
				PUT("\"%s\" ", kfields[t->pad].kw);
				break;
		}
		for (j = 0; j < t->nopnd; j++) {
			const struct xe_opnd *o = &f->opnds[t->opnd_base + j];
			int32_t note = f->opnd_note[t->opnd_base + j];
			const uint8_t *b = (const uint8_t *)&o->w[0];

			if ((t->type == XE_BT_ADDR4) || (t->type == XE_BT_ADDR6)) {
				int nb = (t->type == XE_BT_ADDR4) ? 4 : 16;
				if (o->is_list) {
					PUT(" LIST '%s'", (reg && ((size_t)note < reg->n)) ? reg->l[note].name : "?");
					continue;
				}
				PUT(" ");
				for (k = 0; k < nb; k++) PUT("%u%s", b[k], (k == nb - 1) ? "" : (nb == 4) ? "." : ":");
				PUT("/%d", note);
			} else if (t->type == XE_BT_RANGE) {
				PUT(" %d-%d", (int32_t)o->w[0], (int32_t)o->w[1]);
			} else if (t->type == XE_BT_MAC) {
				PUT(" %02x:%02x:%02x:%02x:%02x:%02x", b[0], b[1], b[2], b[3], b[4], b[5]);
			} else if (t->type == XE_BT_STRING) {
				/* an operand that is not a quoted string has no text in the
				 * reference (it prints an uninitialised pointer) */
				PUT(" %s", ((note >= 0) && ((size_t)note < f->nstrs)) ? f->strs + note : "(null)");
			}
		}
		PUT("\n");
	}
#undef PUT
	return (int)pos;
}

/* ------------------------------------------------------------------ div_l */
/* xdiv()'s is_log branch, filter.h:407-409, as the reference's compiler
 * evaluates it on this host: double log / pow, conversions by truncation (a
 * NaN or infinite ratio converts to INT_MIN on x86-64).  `volatile` keeps the
 * compiler from folding the conversions at build time. */
static int
div_l_value(int quotient, int k)
{
	volatile double ratio = log((double)quotient) / log((double)k);
	volatile int lg = (int)ratio;
	volatile double p = pow((double)k, (double)lg);
	return (int)p;
}

struct dl_out { int32_t *lo, *hi, *val; int n, cap, bad; };

static void
dl_piece(struct dl_out *o, int32_t lo, int32_t hi, int32_t val)
{
	if (o->n && (o->val[o->n - 1] == val) && (o->hi[o->n - 1] == lo - 1)) { o->hi[o->n - 1] = hi; return; }
	if (o->n >= o->cap) { o->bad = 1; return; }
	o->lo[o->n] = lo; o->hi[o->n] = hi; o->val[o->n] = val;
	o->n++;
}

/* the value is a non-decreasing step function of a positive quotient (log,
 * truncation and pow are monotone): equal ends = one piece, else cut in half */
static void
dl_split(struct dl_out *o, int k, int32_t lo, int32_t hi, int vlo, int vhi, int depth)
{
	int32_t mid;
	if (o->bad) return;
	if (vlo == vhi) { dl_piece(o, lo, hi, vlo); return; }
	if (hi == lo + 1) { dl_piece(o, lo, lo, vlo); dl_piece(o, hi, hi, vhi); return; }
	if (depth > 40) { o->bad = 1; return; }
	mid = lo + (hi - lo) / 2;
	{
		const int vm = div_l_value(mid, k), vm1 = div_l_value(mid + 1, k);
		dl_split(o, k, lo, mid, vlo, vm, depth + 1);
		dl_split(o, k, mid + 1, hi, vm1, vhi, depth + 1);
	}
}

int
xe_div_l_pieces(int k, int32_t *lo, int32_t *hi, int32_t *val, int cap)
{
	struct dl_out o = { lo, hi, val, 0, cap, 0 };
	int i;

	/* a negative quotient has no logarithm: every one of them gives the value of -1 */
	dl_piece(&o, INT_MIN, -1, div_l_value(-1, k));
	dl_piece(&o, 0, 0, div_l_value(0, k));
	dl_split(&o, k, 1, INT_MAX, div_l_value(1, k), div_l_value(INT_MAX, k), 0);
	if (o.bad) return -1;
	/* the pieces were found by bisection: look at both ends of each again */
	for (i = 0; i < o.n; i++) {
		if ((div_l_value(lo[i], k) != val[i]) || (div_l_value(hi[i], k) != val[i])) return -1;
	}
	return o.n;
}
