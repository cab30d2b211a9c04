/*
 * xe_kernels.hip -- gfx950 kernels of the flow path.
 *
 * Two kernels decode Netflow v5/v9/IPFIX records (one thread per record) from
 * the export packets into the structure-of-arrays columns, evaluate every
 * monitoring-object program against the record and store the match bitmask
 * and per-object counters:
 *   xe_nf_fast_kernel     regular runs, one template per wave, wave-uniform
 *                         interpretation (see its header comment);
 *   xe_nf_general_kernel  everything else, interpreted per lane.
 *
 * Field values are handled as big-endian integers: the reference keeps a
 * field as SIZEMAX bytes with the wire bytes right-aligned (netflow.c:91), so
 * "the SIZEMAX bytes read as one big-endian number" equals "the wire bytes
 * read as one big-endian number"; filters compare exactly that number
 * (filter.c:275, :484-488) and the IP-list walk consumes its bits from the
 * most significant end (iplist.c:171-184).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include "xe_kernels.h"
#include "xe_sfwalk.h"

#define XE_BLOCK 256

struct val128 { uint64_t hi, lo; };

/* ------------------------------------------------------------------ packet bytes */
struct pkt_src
{
	const uint8_t *p;   /* record start */
	int32_t left;       /* bytes to the datagram end, may be <= 0 */
	bool checked;
};

__device__ __forceinline__ uint32_t
pkt_byte(const pkt_src &s, int32_t off)
{
	if (s.checked && (off >= s.left)) return 0;
	return s.p[off];
}

/* len wire bytes at off as a big-endian number (len <= 16) */
__device__ __forceinline__ val128
pkt_be(const pkt_src &s, int32_t off, uint32_t len)
{
	val128 v = {0, 0};
	if (!s.checked && len) {
		/* every byte of the field is inside the datagram: one or two 8-byte
		 * loads (they may run past the field, never past the batch buffer,
		 * which ends in 64 spare bytes) instead of a load per byte */
		uint64_t a, b;
		__builtin_memcpy(&a, s.p + off, 8);
		a = __builtin_bswap64(a);
		if (len <= 8) {
			v.lo = a >> (64 - 8 * len);
		} else {
			__builtin_memcpy(&b, s.p + off + 8, 8);
			b = __builtin_bswap64(b);
			/* the len bytes a:b as one number, right-aligned in hi:lo */
			const uint32_t sh = 8 * (16 - len);          /* 0..56 */
			v.lo = sh ? ((b >> sh) | (a << (64 - sh))) : b;
			v.hi = sh ? (a >> sh) : a;
		}
		return v;
	}
	for (uint32_t i = 0; i < len; i++) {
		v.hi = (v.hi << 8) | (v.lo >> 56);
		v.lo = (v.lo << 8) | pkt_byte(s, off + (int32_t)i);
	}
	return v;
}

/* ------------------------------------------------------------------ flattened IP list */
/* the XE_NODE_BITS address bits that start `depth` bits into the left-aligned
 * 128-bit address (depth is a multiple of XE_NODE_BITS, which divides 16) */
__device__ __forceinline__ uint32_t
addr_byte(uint64_t hi, uint64_t lo, uint32_t depth)
{
	return (depth < 64) ? (uint32_t)((hi << depth) >> (64 - XE_NODE_BITS))
		: (uint32_t)((lo << (depth - 64)) >> (64 - XE_NODE_BITS));
}

/* chain record: the answers of the group's lists when the rest of the walk is
 * a single path of one list (see XE_CHAIN_WORDS); w[0..5] = its first six
 * words, depth = address bits consumed so far (16..120) */
__device__ __forceinline__ uint32_t
chain_answer(const uint32_t w[6], uint64_t hi, uint64_t lo, uint32_t depth)
{
	const uint32_t meta = w[0], len = (meta >> 11) & 127, l = (meta >> 8) & 7;
	const uint64_t ph = (uint64_t)w[2] | ((uint64_t)w[3] << 32), pl = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
	uint64_t sh, sl;
	if (depth >= 64) { sh = lo << (depth - 64); sl = 0; }
	else { sh = (hi << depth) | (lo >> (64 - depth)); sl = lo << depth; }
	const uint64_t dh = sh ^ ph, dl = sl ^ pl;
	bool m = (len <= 64) ? ((dh >> (64 - len)) == 0) : ((dh == 0) && ((dl >> (128 - len)) == 0));
	/* the path starts at a prefix: leaving it with the first bit stays there */
	m |= ((meta >> 18) & 1) && (dh >> 63);
	return XE_TRIE_TERM | (meta & 0xFF) | ((uint32_t)m << l);
}

/* the 16 address bits that start `depth` bits (16..120, a multiple of 8) into
 * the left-aligned 128-bit address; zeros beyond its end */
__device__ __forceinline__ uint32_t
addr_bits16(uint64_t hi, uint64_t lo, uint32_t depth)
{
	if (depth >= 128) return 0;
	if (depth >= 64) return (uint32_t)((lo << (depth - 64)) >> 48);
	return (uint32_t)(((hi << depth) | (lo >> (64 - depth))) >> 48);
}

/* an entry that carries a short chain in itself -> "ends here" with the
 * answers; any other entry unchanged.  `next16` = addr_bits16() at the depth
 * the entry stands for */
__device__ __forceinline__ uint32_t
entry_resolve(uint32_t e, uint32_t next16)
{
	if ((e & XE_TRIE_INLINE) != XE_TRIE_INLINE) return e;
	uint32_t base, l, len, path, sl;
	if (e & XE_TRIE_INLINE_LONG) {
		base = 0; l = (e >> 16) & 7; len = (e >> 20) & 31; path = e & 0xFFFF; sl = (e >> 19) & 1;
	} else {
		base = e & 0xFF; l = (e >> 8) & 7; len = (e >> 11) & 15; path = (e >> 15) & 0x1FFF; sl = (e >> 28) & 1;
	}
	/* the next len address bits against the path: all equal, or (sl) the
	 * first one different */
	const uint32_t d = (next16 >> (16 - len)) ^ path;
	const uint32_t m = ((d == 0) || (sl && ((d >> (len - 1)) & 1))) ? 1u : 0u;
	return XE_TRIE_TERM | base | (m << l);
}

/* answers of all lists of a group for one address (left-aligned in hi:lo,
 * `bits` wide): bit l = list l.  iplist_match4/6, iplist.c:160-216 */
__device__ __forceinline__ uint32_t
group_match(const uint32_t *T, uint32_t root, uint32_t nodes, uint32_t chains, uint64_t hi, uint64_t lo, uint32_t bits,
	uint32_t rbits)
{
	if (!root) return 0;
	/* the root is indexed with the first rbits (16 or 24) address bits */
	const uint32_t top = (uint32_t)(hi >> (64 - rbits));
	uint32_t e = entry_resolve(T[root + top], addr_bits16(hi, lo, rbits));
	for (uint32_t depth = rbits; !(e & XE_TRIE_TERM) && (depth < bits); depth += XE_NODE_BITS) {
		if (e & XE_TRIE_CHAIN) {
			const uint32_t *c = T + chains + (size_t)(e & ~XE_TRIE_CHAIN) * XE_CHAIN_WORDS;
			const uint32_t w[6] = { c[0], c[1], c[2], c[3], c[4], c[5] };
			e = chain_answer(w, hi, lo, depth);
		} else {
			e = entry_resolve(T[nodes + (size_t)e * XE_NODE_WORDS + addr_byte(hi, lo, depth)],
				addr_bits16(hi, lo, depth + XE_NODE_BITS));
		}
	}
	return (e & XE_TRIE_TERM) ? (e & 0xFF) : 0;
}

/* the last group walk of a lane: objects that test the lists of one group on
 * the same address (the usual `net listK` families) share one walk */
struct walk_cache
{
	uint32_t root;      /* 0 = empty */
	uint64_t hi, lo;
	uint32_t ans;
};

__device__ __forceinline__ uint32_t
group_match_cached(walk_cache &wc, const uint32_t *T, uint32_t root, uint32_t nodes, uint32_t chains,
	uint64_t hi, uint64_t lo, uint32_t bits, uint32_t rbits)
{
	if (!root) return 0;
	if ((wc.root != root) || (wc.hi != hi) || (wc.lo != lo)) {
		wc.ans = group_match(T, root, nodes, chains, hi, lo, bits, rbits);
		wc.root = root;
		wc.hi = hi;
		wc.lo = lo;
	}
	return wc.ans;
}

__device__ __forceinline__ uint32_t
trie_match4(const uint32_t *T, const xe_list_dev &L, uint32_t a)
{
	return (group_match(T, L.root4, L.nodes4, L.chains4, (uint64_t)a << 32, 0, 32, L.rbits & 0xFF) >> L.bit) & 1;
}

__device__ __forceinline__ uint32_t
trie_match6(const uint32_t *T, const xe_list_dev &L, uint64_t hi, uint64_t lo)
{
	return (group_match(T, L.root6, L.nodes6, L.chains6, hi, lo, 128, (L.rbits >> 8) & 0xFF) >> L.bit) & 1;
}

/* ------------------------------------------------------------------ field sources */
/* what a filter term asks of a record: presence + value of a field */
struct fval { bool has; val128 v; };

__device__ __forceinline__ uint64_t
low_mask(uint32_t bytes)
{
	return (bytes >= 8) ? ~0ull : ((1ull << (8 * bytes)) - 1ull);
}

/* a Netflow/IPFIX record in the export packet */
struct nf_rec
{
	pkt_src src;
	const xe_tmpl *t;
	const xe_tent *e;
	const xe_exporter *ex;
	uint32_t nparsed;
	uint32_t mark;        /* XE_MARK_* | count */

	/* offset and length of entry j; variable-length templates are re-walked */
	__device__ void locate(uint32_t j, int32_t *off, uint32_t *len) const
	{
		if (!t->varlen) {
			*off = e[j].off;
			*len = e[j].len;
			return;
		}
		int32_t pos = 0;
		for (uint32_t i = 0;; i++) {
			uint32_t l = e[i].len;
			int32_t adj = 0;
			if ((t->mode == XE_MODE_IPFIX) && (l == 0xFFFF)) {
				uint32_t b0 = pkt_byte(src, pos);
				if (b0 != 255) { l = b0; adj = 1; }
				else { l = (pkt_byte(src, pos + 1) << 8) | pkt_byte(src, pos + 2); adj = 3; }
			}
			if (i == j) { *off = pos + adj; *len = l; return; }
			pos += (int32_t)l + adj;
		}
	}

	/* is entry j's length acceptable for its field (netflow.c:83) */
	__device__ bool valid(uint32_t j, uint32_t len, const uint16_t *fmin, const uint16_t *fmax) const
	{
		if (e[j].len != 0xFFFF) return (e[j].flags & XE_TE_VALID) != 0;
		return (len >= fmin[e[j].field]) && (len <= fmax[e[j].field]);
	}
};

/* field size limits, filled once per launch from the host table */
__constant__ uint16_t c_fmin[XE_NFIELDS];
__constant__ uint16_t c_fmax[XE_NFIELDS];
__constant__ uint16_t c_colw[XE_NFIELDS];
__constant__ uint8_t c_fstr[XE_NFIELDS];

/* value of a non-string field after all parsed occurrences were applied in
 * template order (a later, shorter occurrence overlays only its own bytes:
 * netflow.c:91 copies into the tail of the slot without clearing it) */
__device__ __forceinline__ fval
nf_get(const nf_rec &r, uint32_t f, uint32_t *size_out)
{
	fval out;
	out.has = false;
	out.v.hi = out.v.lo = 0;
	if (size_out) *size_out = 0;
	if (f == XF_dev_ip) {
		/* virtual_fields_init() runs after the template fields (netflow.c:293) */
		const uint8_t *b = r.ex->dev_ip;
		out.has = true;
		out.v.lo = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
		if (size_out) *size_out = 4;
		return out;
	}
	if (f == XF_dev_id) {
		const uint8_t *b = r.ex->dev_id;
		out.has = true;
		out.v.lo = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
		if (size_out) *size_out = 4;
		return out;
	}
	if (f == XF_dev_mark) {
		out.has = (r.mark & XE_MARK_HAS) != 0;
		out.v.lo = r.mark & 0x3FFFFFFFu;
		if (size_out) *size_out = 4;
		return out;
	}
	if (f >= XE_NTABLE_FIELDS) return out;
	uint32_t j = r.t->fmap[f];
	while (j != 0xFF) {
		if (j < r.nparsed) {
			int32_t off;
			uint32_t len;
			r.locate(j, &off, &len);
			if (r.valid(j, len, c_fmin, c_fmax)) {
				val128 nv = pkt_be(r.src, off, len > 16 ? 16 : len);
				if (len >= 16) {
					out.v = nv;
				} else if (len > 8) {
					uint64_t m = low_mask(len - 8);
					out.v.hi = (out.v.hi & ~m) | (nv.hi & m);
					out.v.lo = nv.lo;
				} else {
					uint64_t m = low_mask(len);
					out.v.lo = (out.v.lo & ~m) | (nv.lo & m);
				}
				out.has = true;
				if (size_out) *size_out = len;
			}
		}
		j = r.e[j].next_same;
	}
	return out;
}

/* ------------------------------------------------------------------ filter VM */
__device__ __forceinline__ bool
in_ranges(const xe_dopnd *o, uint32_t n, int64_t v)
{
	for (uint32_t i = 0; i < n; i++) {
		if ((v >= o[i].low) && (v <= o[i].high)) return true;
	}
	return false;
}

/* get_nf_val(): sizes 1, 2, 4, 8 read big-endian, anything else is 0 (filter.h:370-394) */
__device__ __forceinline__ uint64_t
nf_val(const fval &f, uint32_t sizemax)
{
	return ((sizemax == 1) || (sizemax == 2) || (sizemax == 4) || (sizemax == 8)) ? f.v.lo : 0;
}

/* tfstr(): the byte the reference reads from the field's storage (byte 1 of
 * a 2-byte field, else byte 0; filter.c:842-847) out of the right-aligned
 * value; absent fields read as 0 */
__device__ __forceinline__ uint32_t
tfstr_byte(uint64_t hi, uint64_t lo, uint32_t fmax)
{
	if (fmax == 2) return (uint32_t)lo & 0xFF;
	if (fmax > 8) return (uint32_t)(hi >> ((fmax - 9) * 8)) & 0xFF;
	return (uint32_t)(lo >> ((fmax - 1) * 8)) & 0xFF;
}

/* portstr() / ppstr() (filter.c:861-904) on the port values.  portstr: every
 * operand is the set of ports whose text (port_to_str) equals the operand's,
 * 65536 bits at word `low` of the list tables.  ppstr: every operand is a
 * condition on the pair (see ppstr_lower() on the host): kind 1 both ports
 * given, kind 2 port1 given and port2 any larger port without a service name,
 * kind 3 the mirror image. */
__device__ __forceinline__ bool
port_text_match(const xe_kparams &kp, uint32_t func, const xe_dopnd *o, uint32_t n, uint32_t p1, uint32_t p2)
{
	bool res = false;
	if (func == XE_FN_PORTSTR) {
		for (uint32_t i = 0; i < n; i++) res |= ((kp.trie[(uint32_t)o[i].low + (p1 >> 5)] >> (p1 & 31)) & 1) != 0;
		return res;
	}
	const uint32_t *named = kp.trie + kp.port_named_off;
	const bool n1 = ((named[p1 >> 5] >> (p1 & 31)) & 1) != 0, n2 = ((named[p2 >> 5] >> (p2 & 31)) & 1) != 0;
	for (uint32_t i = 0; i < n; i++) {
		const uint32_t kind = o[i].list, X = (uint32_t)o[i].low, Y = (uint32_t)o[i].high;
		res |= (kind == 1) && (p1 == X) && (p2 == Y);
		res |= (kind == 2) && (p1 == X) && (p2 > p1) && !n2;
		res |= (kind == 3) && (p2 == Y) && (p1 > p2) && !n1;
	}
	return res;
}

template <typename REC, fval (*GET)(const REC &, uint32_t, uint32_t *)>
__device__ __forceinline__ bool
term_match(const xe_kparams &kp, const REC &rec, const xe_term &t, walk_cache &wc)
{
	const xe_dopnd *o = kp.opnds + t.opnd_base;

	if (t.func == XE_FN_TFSTR) {
		const fval v = GET(rec, t.fa, nullptr);
		const uint32_t by = tfstr_byte(v.v.hi, v.v.lo, c_fmax[t.fa]);
		bool res = false;
		for (uint32_t i = 0; i < t.nopnd; i++) {
			const uint64_t w = (by < 64) ? o[i].net_hi : (by < 128) ? o[i].net_lo : (by < 192) ? o[i].mask_hi : o[i].mask_lo;
			res |= ((w >> (by & 63)) & 1) != 0;
		}
		return res;
	}
	if ((t.func == XE_FN_PORTSTR) || (t.func == XE_FN_PPSTR)) {
		const uint32_t p1 = (uint32_t)nf_val(GET(rec, t.fa, nullptr), c_fmax[t.fa]) & 0xFFFF;
		const uint32_t p2 = (t.func == XE_FN_PPSTR) ? ((uint32_t)nf_val(GET(rec, t.fb, nullptr), c_fmax[t.fb]) & 0xFFFF) : 0;
		return port_text_match(kp, t.func, o, t.nopnd, p1, p2);
	}
	if (t.func != XE_FN_NONE) {
		uint64_t a = nf_val(GET(rec, t.fa, nullptr), c_fmax[t.fa]);
		uint64_t b = nf_val(GET(rec, t.fb, nullptr), c_fmax[t.fb]);
		if (t.func == XE_FN_MIN) {
			/* filter.c:685-709 */
			return in_ranges(o, t.nopnd, (int64_t)((a < b) ? a : b));
		}
		/* xdiv(), filter.h:396-418: the quotient is truncated to int first.
		 * div_l: the operands are ranges of the quotient (the logarithm was
		 * applied to their bounds at commit), k = the answer for a zero divisor */
		int32_t q = 0;
		if (b != 0) {
			q = (int32_t)(a / b);
			if ((t.func == XE_FN_DIV_R) && (t.k > 1)) q = (q / t.k) * t.k;
		} else if (t.func == XE_FN_DIV_L) {
			return t.k != 0;
		}
		return in_ranges(o, t.nopnd, q);
	}

	switch (t.type) {
		case XE_BT_ADDR4:
		case XE_BT_ADDR6: {
			/* filter.c:255-466: SRC/DST need the field present; BOTH tries
			 * whichever side is present */
			fval c[2];
			int nc = 0;
			if (t.dir == XE_DIR_SRC) {
				c[0] = GET(rec, t.fsrc, nullptr);
				if (!c[0].has) {
					if (t.pad != XE_TERM_ZERO_ABSENT) return false;
					c[0].v.hi = c[0].v.lo = 0;
				}
				nc = 1;
			}
			else if (t.dir == XE_DIR_DST) { c[0] = GET(rec, t.fdst, nullptr); if (!c[0].has) return false; nc = 1; }
			else if (t.dir == XE_DIR_BOTH) {
				fval s = GET(rec, t.fsrc, nullptr), d = GET(rec, t.fdst, nullptr);
				if (s.has) c[nc++] = s;
				if (d.has) c[nc++] = d;
			}
			if (!nc) return false;
			/* an IPv4-typed term over a 16-byte field looks at its first 4 bytes */
			bool wide4 = (t.type == XE_BT_ADDR4) && (c_fmax[t.fsrc] == 16);
			for (uint32_t i = 0; i < t.nopnd; i++) {
				for (int k = 0; k < nc; k++) {
					if (t.type == XE_BT_ADDR4) {
						uint32_t a = wide4 ? (uint32_t)(c[k].v.hi >> 32) : (uint32_t)c[k].v.lo;
						if (o[i].is_list) {
							const xe_list_dev &L = kp.lists[o[i].list];
							if ((group_match_cached(wc, kp.trie, L.root4, L.nodes4, L.chains4, (uint64_t)a << 32, 0, 32, L.rbits & 0xFF) >> L.bit) & 1) return true;
						} else if ((a & (uint32_t)o[i].mask_lo) == (uint32_t)o[i].net_lo) {
							return true;
						}
					} else {
						if (o[i].is_list) {
							const xe_list_dev &L = kp.lists[o[i].list];
							if ((group_match_cached(wc, kp.trie, L.root6, L.nodes6, L.chains6, c[k].v.hi, c[k].v.lo, 128, (L.rbits >> 8) & 0xFF) >> L.bit) & 1) return true;
						} else if (((c[k].v.hi & o[i].mask_hi) == o[i].net_hi)
							&& ((c[k].v.lo & o[i].mask_lo) == o[i].net_lo)) {
							return true;
						}
					}
				}
			}
			return false;
		}
		case XE_BT_RANGE: {
			/* filter.c:468-550: no presence test; 1/2/4-byte fields only */
			if (t.dir == XE_DIR_NONE) return false;
			if (t.dir & XE_DIR_SRC) {
				uint32_t sm = c_fmax[t.fsrc];
				int32_t r1 = (sm <= 4) ? (int32_t)(uint32_t)GET(rec, t.fsrc, nullptr).v.lo : 0;
				if (in_ranges(o, t.nopnd, r1)) return true;
			}
			if (t.dir & XE_DIR_DST) {
				uint32_t sm = c_fmax[t.fdst];
				int32_t r2 = (sm <= 4) ? (int32_t)(uint32_t)GET(rec, t.fdst, nullptr).v.lo : 0;
				if (in_ranges(o, t.nopnd, r2)) return true;
			}
			return false;
		}
		case XE_BT_MAC: {
			/* filter.c:552-606: 6-byte compare, no presence test */
			if (t.dir == XE_DIR_NONE) return false;
			uint64_t m1 = (t.dir & XE_DIR_SRC) ? GET(rec, t.fsrc, nullptr).v.lo : 0;
			uint64_t m2 = (t.dir & XE_DIR_DST) ? GET(rec, t.fdst, nullptr).v.lo : 0;
			for (uint32_t i = 0; i < t.nopnd; i++) {
				if ((t.dir & XE_DIR_SRC) && (o[i].net_lo == m1)) return true;
				if ((t.dir & XE_DIR_DST) && (o[i].net_lo == m2)) return true;
			}
			return false;
		}
		default:
			return false;
	}
}

/* postfix program over a bit stack: bit 0 = top (filter.c:960-1012; depth
 * errors are resolved at compile time into constant programs) */
template <typename REC, fval (*GET)(const REC &, uint32_t, uint32_t *)>
__device__ __forceinline__ bool
prog_match(const xe_kparams &kp, const REC &rec, const xe_prog &pr, walk_cache &wc)
{
	if (pr.flags & XE_PROG_CONST) return pr.cval != 0;
	uint64_t st = 0;
	for (uint32_t i = 0; i < pr.nops; i++) {
		uint32_t w = kp.ops[pr.op_base + i];
		switch (w & 3) {
			case XE_OP_BASIC:
				st = (st << 1) | (uint64_t)term_match<REC, GET>(kp, rec, kp.terms[w >> 2], wc);
				break;
			case XE_OP_NOT:
				st ^= 1;
				break;
			case XE_OP_AND: {
				uint64_t top = st & 1;
				st >>= 1;
				st &= (top | ~1ull);
				break;
			}
			default: {
				uint64_t top = st & 1;
				st >>= 1;
				st |= top;
				break;
			}
		}
	}
	return (st & 1) != 0;
}

__device__ __forceinline__ unsigned long long
wave_sum(unsigned long long v)
{
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

/* column stores of the per-lane kernels: XE_NT_LANE = with the streaming hint
 * (measured on MI355X: per-lane Netflow kernel 0.634 -> 0.612 ms, sFlow kernel
 * 0.408 -> 0.403 ms) */
#ifndef XE_NT_LANE
#define XE_NT_LANE 1
#endif
#if XE_NT_LANE
#define XE_LANE_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define XE_LANE_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif

/* store a non-string field value into its column in flow_info byte order */
__device__ __forceinline__ void
store_col(const xe_kparams &kp, uint32_t f, uint32_t ridx, const val128 &v)
{
	uint32_t w = c_colw[f], sm = c_fmax[f];
	uint8_t *dst = kp.out_base + kp.col_off[f] + (size_t)ridx * w;
	switch (w) {
		case 1: XE_LANE_STORE(dst, (uint8_t)v.lo); break;
		case 2: XE_LANE_STORE((uint16_t *)dst, __builtin_bswap16((uint16_t)v.lo)); break;
		case 4: XE_LANE_STORE((uint32_t *)dst, __builtin_bswap32((uint32_t)v.lo)); break;
		case 8: XE_LANE_STORE((uint64_t *)dst, __builtin_bswap64(v.lo << (8 * (8 - sm)))); break;
		default:
			XE_LANE_STORE((uint64_t *)dst, __builtin_bswap64(v.hi));
			XE_LANE_STORE((uint64_t *)dst + 1, __builtin_bswap64(v.lo));
			break;
	}
}

/* ------------------------------------------------------------------ fast kernel */
/*
 * One wave = one tile of 64 * R complete records that share a template (lane l
 * owns records l, 64 + l, ... of the tile), so the walk over the template's
 * fields, over the filter terms and over the object programs is wave-uniform:
 * scalar loads and scalar branches, paid once per 64 * R records.  Every
 * record of a tile carries all fields of its template (runs with cut-short
 * records go to the general kernel), so field presence is wave-uniform too.
 *
 * Phases per tile:
 *   runs     the runs overlapping the tile are walked with scalar loads; each
 *            lane keeps packet offset, record index and exporter data of its
 *            records;
 *   stage    every lane copies its records into LDS with 16-byte loads that
 *            are all in flight together (layout [dword][r][lane]: conflict
 *            free, aligned); everything later reads record bytes from LDS;
 *   decode   per known template field one 16-byte micro-op: bytes go from LDS
 *            to the field's column, consecutive lanes = consecutive records;
 *   classify IP-list walks (R at a time per probe), terms, object programs,
 *            match masks; counters are staged in LDS per block.
 */
#define XE_CONST __attribute__((address_space(4)))
#define XE_GLOBAL __attribute__((address_space(1)))
#define XE_LDS __attribute__((address_space(3)))
/* the block's counters and atom-table copy: named as LDS so that their
 * accesses are LDS instructions wherever the compiler loses track of the
 * pointer's origin (a generic pointer turns the adds into flat atomics) */
typedef XE_LDS unsigned long long lds_u64;
typedef const XE_CONST uint32_t *cu32;
typedef const XE_CONST uint64_t *cu64;
typedef const XE_CONST uint8_t *cu8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct term_s { uint32_t type, dir, fsrc, fdst, func, fa, fb, nopnd; int32_t k; const xe_dopnd *o; };
__device__ __forceinline__ term_s
ld_term(const xe_kparams &kp, uint32_t t)
{
	cu32 q = (cu32)(kp.terms + t);
	uint32_t w0 = q[0], w1 = q[1];
	term_s r;
	r.type = w0 & 0xFF; r.dir = (w0 >> 8) & 0xFF; r.fsrc = (w0 >> 16) & 0xFF; r.fdst = w0 >> 24;
	r.func = w1 & 0xFF; r.fa = (w1 >> 8) & 0xFF; r.fb = (w1 >> 16) & 0xFF;
	r.k = (int32_t)q[2];
	r.nopnd = q[3];
	r.o = kp.opnds + q[4];
	return r;
}

__device__ __forceinline__ bool
in_ranges_u(const xe_dopnd *o, uint32_t n, int64_t v)
{
	bool r = false;
	for (uint32_t i = 0; i < n; i++) {
		cu32 q = (cu32)(o + i);
		int32_t lo = (int32_t)q[2], hi = (int32_t)q[3];
		r |= (v >= lo) && (v <= hi);
	}
	return r;
}

/* per wave and tile */
template <int R>
struct fast_tile
{
	const uint32_t *stg;       /* LDS: record dword d of record r of this lane at stg[(d * R + r) * 64] */
	cu32 fld;                  /* per field of the template: XE_FFLD_* word */
	uint32_t dev_ip[R], dev_id[R];
	uint32_t mark[R];          /* the device mark once the mark filters have run (0 before) */
	bool live[R], markhas[R];
};

/* `len` wire bytes at record offset `off` (both wave-uniform) of the lane's
 * record r, as little-endian words a[0..3] (bytes beyond len are not cleared) */
template <int R>
__device__ __forceinline__ void
stage_read(const uint32_t *stg, uint32_t r, uint32_t off, uint32_t len, uint32_t a[4])
{
	const uint32_t d0 = off >> 2, sh = off & 3;
	const uint32_t nd = (sh + len + 3) >> 2;
	const uint32_t *p = stg + (d0 * R + r) * 64;
	uint32_t w0 = p[0], w1 = 0, w2 = 0, w3 = 0, w4 = 0;

	if (nd > 1) w1 = p[R * 64];
	if (nd > 2) {
		w2 = p[2 * R * 64];
		if (nd > 3) w3 = p[3 * R * 64];
		if (nd > 4) w4 = p[4 * R * 64];
	}
	if (sh) {
		a[0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
		a[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
		a[2] = __builtin_amdgcn_alignbyte(w3, w2, sh);
		a[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
	} else {
		a[0] = w0; a[1] = w1; a[2] = w2; a[3] = w3;
	}
}

/* little-endian wire words -> the value as a big-endian number */
__device__ __forceinline__ void
wire_number(const uint32_t a[4], uint32_t len, uint64_t &hi, uint64_t &lo)
{
	hi = 0;
	if (len <= 4) {
		lo = __builtin_bswap32(a[0]) >> (8 * (4 - len));
	} else if (len <= 8) {
		lo = __builtin_bswap64((uint64_t)a[0] | ((uint64_t)a[1] << 32)) >> (8 * (8 - len));
	} else {
		const uint64_t A = __builtin_bswap64((uint64_t)a[0] | ((uint64_t)a[1] << 32));
		const uint64_t B = __builtin_bswap64((uint64_t)a[2] | ((uint64_t)a[3] << 32));
		const uint32_t sh = 8 * (16 - len);
		lo = sh ? ((B >> sh) | (A << (64 - sh))) : B;
		hi = A >> sh;
	}
}

/* presence and value of a field of the lane's record r, as the filter terms
 * see it (flow_info bytes read as a big-endian number) */
template <int R>
__device__ __forceinline__ void
fld_get(const fast_tile<R> &T, uint32_t field, uint32_t r, bool &has, uint32_t &fmax, uint64_t &hi, uint64_t &lo)
{
	if (field == XF_dev_mark) {
		/* device_rules_check(), devices.c:308-311: 4 bytes, present once the rules have run */
		has = T.markhas[r];
		fmax = 4;
		hi = 0;
		lo = T.mark[r];
		return;
	}
	const uint32_t fe = T.fld[field];
	has = (fe >> 31) != 0;
	fmax = (fe >> 21) & 31;
	hi = lo = 0;
	if (!has) return;
	if (fe & XE_FFLD_VIRTUAL) {
		lo = (field == XF_dev_ip) ? T.dev_ip[r] : T.dev_id[r];
		return;
	}
	uint32_t a[4];
	const uint32_t len = (fe >> 16) & 31;
	stage_read<R>(T.stg, r, fe & 0xFFFF, len, a);
	wire_number(a, len, hi, lo);
}

/* field access of the classification code: from the staged record of a fast
 * tile, or through the field getter of the per-lane kernels */
template <int R>
__device__ __forceinline__ void
tile_get(const fast_tile<R> &T, uint32_t field, uint32_t r, bool &has, uint32_t &fmax, uint64_t &hi, uint64_t &lo)
{
	fld_get<R>(T, field, r, has, fmax, hi, lo);
}

template <typename REC, fval (*GET)(const REC &, uint32_t, uint32_t *)>
struct lane_tile
{
	const REC *rec;
	bool live[1];
};

template <typename REC, fval (*GET)(const REC &, uint32_t, uint32_t *)>
__device__ __forceinline__ void
tile_get(const lane_tile<REC, GET> &T, uint32_t field, uint32_t r, bool &has, uint32_t &fmax, uint64_t &hi, uint64_t &lo)
{
	(void)r;
	/* the record of a lane that is not live may be unset: never read it */
	fval v;
	v.has = false;
	v.v.hi = v.v.lo = 0;
	if (T.live[0]) v = GET(*T.rec, field, nullptr);
	has = v.has;
	fmax = c_fmax[field];
	hi = v.has ? v.v.hi : 0;
	lo = v.has ? v.v.lo : 0;
}

/* diagnostic phase timing (XE_ABLATE bit 6): wave-clock cycles per phase, summed
 * over the waves' tiles into xe_fast_phase_cycles[] */
#ifdef XE_DIAG
__device__ unsigned long long xe_fast_phase_cycles[8];
#endif
#ifdef XE_DIAG
/* diagnostic build only (tools/build_variant.sh -DXE_DIAG): phases of the kernels can be
 * switched off through XE_ABLATE (wrong results, timing only) and wave cycles per phase
 * collected; the product library is compiled without any of it */
#define XE_ABL(BIT) (kp.pad2 & (BIT))
#define XE_PHASE_DECL uint64_t tprev = __builtin_readcyclecounter()
#define XE_PHASE(N) do { if XE_ABL(64) { const uint64_t now_ = __builtin_readcyclecounter(); \
	if (lane == 0) atomicAdd(&xe_fast_phase_cycles[N], (unsigned long long)(now_ - tprev)); tprev = now_; } } while (0)
#else
#define XE_ABL(BIT) (0)
#define XE_PHASE_DECL do { } while (0)
#define XE_PHASE(N) do { (void)lane; } while (0)
#endif

/*
 * All IP-list membership tests of a tile.  Per probe (one field against one
 * group of up to 8 lists) ONE walk of the group's merged tables per record: a
 * 4-byte root gather, then per further address byte one 4-byte node visit
 * while several prefixes are still possible, and one chain record (24 bytes)
 * once a single prefix is left.  The walks are dependent gathers, so several
 * are kept in flight: PC probes x R records advance level by level together,
 * all loads of a level issued before the first result is used.  `plist` names
 * the probes whose field the template carries (byte i = probe index, 0xFF =
 * end).  Bit (8 * probe + l) of lk[r] = list l of the probe's group matched.
 */
/* XE_NT_GATHER: the gathers from the list tables with the streaming hint.
 * Measured on MI355X: C2 0.2314 -> 0.3453 ms, C1 0.0856 -> 0.1198 ms, C4 0.374 ->
 * 0.520 ms - the walks live on the upper table levels staying in the caches,
 * so the hint stays off; what streams through (packets in, columns out) is
 * what should keep out of their way. */
#ifndef XE_NT_GATHER
#define XE_NT_GATHER 0
#endif
#if XE_NT_GATHER
#define XE_GATHER_LOAD(PTR) __builtin_nontemporal_load(PTR)
#else
#define XE_GATHER_LOAD(PTR) (*(PTR))
#endif
template <int R, int LW, typename TT, int PC = 2>
__device__ void
fast_lookups(const xe_kparams &kp, const TT &T, const uint64_t (&plists)[LW], uint64_t (&lk)[R][LW])
{
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#pragma unroll
	for (int r = 0; r < R; r++) {
#pragma unroll
		for (int w = 0; w < LW; w++) lk[r][w] = 0;
	}

	/* LW words of 8 probe indices each, filled from the front */
#pragma unroll 1
	for (int pw = 0; pw < LW; pw++) {
	uint64_t plist = plists[0];
#pragma unroll
	for (int k = 1; k < LW; k++) plist = (pw == k) ? plists[k] : plist;
	while ((plist & 0xFF) != 0xFF) {
		uint32_t nodes[PC], chains[PC], used[PC], pidx[PC], bits[PC], rskip[PC];
		bool act[PC];
		uint32_t e[PC][R], roots[PC];
		uint64_t hi[PC][R], lo[PC][R];
		bool need[PC][R];
		uint32_t maxbits = 0;

#pragma unroll
		for (int c = 0; c < PC; c++) {
			const uint32_t pr = (uint32_t)(plist >> (8 * c)) & 0xFF;
			/* the list ends with 0xFF bytes: once inactive, always inactive */
			act[c] = (pr != 0xFF);
			cu32 q = (cu32)(kp.probes + (act[c] ? pr : 0));
			const uint32_t root = q[0], w1 = q[1];
			nodes[c] = q[3];
			chains[c] = q[4];
			/* address bits the root takes beyond 16 (0 or 8): this probe's walk is
			 * that much deeper at every step of the loop below */
			rskip[c] = (q[5] == 24) ? 8u : 0u;
			used[c] = w1 >> 24;
			pidx[c] = pr;
			bits[c] = act[c] ? (((w1 & 0xFF) == 4) ? 32u : 128u) : 0u;
			if (bits[c] > maxbits) maxbits = bits[c];
#pragma unroll
			for (int r = 0; r < R; r++) {
				bool has;
				uint32_t fmax;
				e[c][r] = XE_TRIE_TERM;
				hi[c][r] = lo[c][r] = 0;
				need[c][r] = false;
				if (!act[c]) continue;
				tile_get(T, (w1 >> 8) & 0xFF, r, has, fmax, hi[c][r], lo[c][r]);
				/* lists made of a geo / AS database: an absent field is looked up as zeros */
				if (!has) { hi[c][r] = lo[c][r] = 0; has = ((w1 >> 17) & 1) != 0; }
				if (bits[c] == 32) {
					/* IPv4: the field's 4 bytes, or the first 4 of a 16-byte field */
					const uint32_t a4 = ((w1 >> 16) & 1) ? (uint32_t)(hi[c][r] >> 32) : (uint32_t)lo[c][r];
					hi[c][r] = (uint64_t)a4 << 32;
					lo[c][r] = 0;
				}
				need[c][r] = act[c] && T.live[r] && has && root;
			}
			roots[c] = root;
		}
		if (kp.gathers) {
			uint32_t nroot = 0;
#pragma unroll
			for (int c = 0; c < PC; c++) {
#pragma unroll
				for (int r = 0; r < R; r++) nroot += (uint32_t)__popcll(__ballot(need[c][r]));
			}
			if (((threadIdx.x & 63) == 0) && nroot) atomicAdd(kp.gathers, (unsigned long long)nroot);
		}
#pragma unroll
		for (int c = 0; c < PC; c++) {
#pragma unroll
			for (int r = 0; r < R; r++) {
				if (!need[c][r]) continue;
				e[c][r] = XE_GATHER_LOAD(&((const XE_CONST uint32_t *)(kp.trie + roots[c]))[(uint32_t)(hi[c][r] >> (48 - rskip[c]))]);
			}
		}
#pragma unroll
		for (int c = 0; c < PC; c++) {
#pragma unroll
			for (int r = 0; r < R; r++) {
				e[c][r] = entry_resolve(e[c][r], addr_bits16(hi[c][r], lo[c][r], 16 + rskip[c]));
			}
		}
		for (uint32_t depth = 16; depth < maxbits; depth += XE_NODE_BITS) {
			bool walk[PC][R], any = false;
			uint32_t nx[PC][R];
			u32x4 ca[PC][R];
			u32x2 cb[PC][R];
#pragma unroll
			for (int c = 0; c < PC; c++) {
#pragma unroll
				for (int r = 0; r < R; r++) {
					walk[c][r] = act[c] && (depth + rskip[c] < bits[c]) && !(e[c][r] & XE_TRIE_TERM);
					any |= walk[c][r];
				}
			}
			if (!__ballot(any)) break;
			if (kp.gathers) {
				uint32_t nw = 0;
#pragma unroll
				for (int c = 0; c < PC; c++) {
#pragma unroll
					for (int r = 0; r < R; r++) nw += (uint32_t)__popcll(__ballot(walk[c][r]));
				}
				if ((threadIdx.x & 63) == 0) atomicAdd(kp.gathers + 1, (unsigned long long)nw);
			}
			/* only lanes that still walk touch memory */
#pragma unroll
			for (int c = 0; c < PC; c++) {
				if (!act[c] || (depth + rskip[c] >= bits[c])) continue;
#pragma unroll
				for (int r = 0; r < R; r++) {
					nx[c][r] = e[c][r];
					ca[c][r] = (u32x4){0, 0, 0, 0};
					cb[c][r] = (u32x2){0, 0};
					if (walk[c][r]) {
						if (e[c][r] & XE_TRIE_CHAIN) {
							const XE_CONST uint32_t *cp = (const XE_CONST uint32_t *)(kp.trie + chains[c])
								+ (size_t)((e[c][r] & ~XE_TRIE_CHAIN) & (XE_ABL(128) ? 63u : ~0u)) * XE_CHAIN_WORDS;
							ca[c][r] = XE_GATHER_LOAD((const XE_CONST u32x4 *)cp);
							cb[c][r] = XE_GATHER_LOAD((const XE_CONST u32x2 *)(cp + 4));
						} else {
							nx[c][r] = XE_GATHER_LOAD(&((const XE_CONST uint32_t *)(kp.trie + nodes[c]))[(size_t)(e[c][r] & (XE_ABL(128) ? 15u : ~0u)) * XE_NODE_WORDS
								+ addr_byte(hi[c][r], lo[c][r], depth + rskip[c])]);
						}
					}
				}
			}
#pragma unroll
			for (int c = 0; c < PC; c++) {
				if (!act[c] || (depth + rskip[c] >= bits[c])) continue;
#pragma unroll
				for (int r = 0; r < R; r++) {
					if (walk[c][r] && (e[c][r] & XE_TRIE_CHAIN)) {
						const uint32_t w[6] = { ca[c][r].x, ca[c][r].y, ca[c][r].z, ca[c][r].w, cb[c][r].x, cb[c][r].y };
						e[c][r] = chain_answer(w, hi[c][r], lo[c][r], depth + rskip[c]);
					} else if (walk[c][r]) {
						e[c][r] = entry_resolve(nx[c][r], addr_bits16(hi[c][r], lo[c][r], depth + rskip[c] + XE_NODE_BITS));
					}
				}
			}
		}
#pragma unroll
		for (int c = 0; c < PC; c++) {
			if (!act[c]) continue;
#pragma unroll
			for (int r = 0; r < R; r++) {
				const uint64_t bits8 = (uint64_t)(((e[c][r] & XE_TRIE_TERM) ? e[c][r] : 0u) & used[c] & 0xFFu)
					<< ((pidx[c] & 7) * XE_LIST_GROUP);
				if (LW == 1) {
					lk[r][0] |= bits8;
				} else {
#pragma unroll
					for (int w = 0; w < LW; w++) lk[r][w] |= ((pidx[c] >> 3) == (uint32_t)w) ? bits8 : 0;
				}
			}
		}
		plist = (PC >= 8) ? ~0ull : ((plist >> (8 * PC)) | (~0ull << (64 - 8 * PC)));
	}
	}
}

/* one term for record r of all lanes; same semantics as term_match(), with the
 * list memberships taken from fast_lookups() */
/* lookup bit `id` (8 * probe + list) of a record */
template <int LW>
__device__ __forceinline__ bool
lk_bit(const uint64_t (&lk)[LW], uint32_t id)
{
	uint64_t w = lk[0];
#pragma unroll
	for (int k = 1; k < LW; k++) w = ((id >> 6) == (uint32_t)k) ? lk[k] : w;
	return ((w >> (id & 63)) & 1) != 0;
}

template <int R, int LW, typename TT>
__device__ bool
fast_term(const xe_kparams &kp, const TT &T, const term_s &t, uint32_t r, const uint64_t (&lkbits)[LW])
{
	const xe_dopnd *o = t.o;
	bool has;
	uint32_t fmax;
	uint64_t hi, lo;

	if (t.func == XE_FN_TFSTR) {
		tile_get(T, t.fa, r, has, fmax, hi, lo);
		const uint32_t by = tfstr_byte(hi, lo, fmax);
		bool res = false;
		for (uint32_t i = 0; i < t.nopnd; i++) {
			cu64 w = (cu64)(o + i);
			const uint64_t set = (by < 64) ? w[2] : (by < 128) ? w[3] : (by < 192) ? w[4] : w[5];
			res |= ((set >> (by & 63)) & 1) != 0;
		}
		return res;
	}
	if (t.func != XE_FN_NONE) {
		tile_get(T, t.fa, r, has, fmax, hi, lo);
		uint64_t a = ((fmax == 1) || (fmax == 2) || (fmax == 4) || (fmax == 8)) ? lo : 0;
		if (t.func == XE_FN_PORTSTR) return port_text_match(kp, t.func, o, t.nopnd, (uint32_t)a & 0xFFFF, 0);
		tile_get(T, t.fb, r, has, fmax, hi, lo);
		uint64_t b = ((fmax == 1) || (fmax == 2) || (fmax == 4) || (fmax == 8)) ? lo : 0;
		if (t.func == XE_FN_PPSTR) return port_text_match(kp, t.func, o, t.nopnd, (uint32_t)a & 0xFFFF, (uint32_t)b & 0xFFFF);
		if (t.func == XE_FN_MIN) return in_ranges_u(o, t.nopnd, (int64_t)((a < b) ? a : b));
		int32_t q = 0;
		if (b != 0) {
			q = (int32_t)(a / b);
			if ((t.func == XE_FN_DIV_R) && (t.k > 1)) q = (q / t.k) * t.k;
		} else if (t.func == XE_FN_DIV_L) {
			return t.k != 0;
		}
		return in_ranges_u(o, t.nopnd, q);
	}
	if (t.dir == XE_DIR_NONE) return false;

	bool res = false;
	for (uint32_t side = XE_DIR_SRC; side <= XE_DIR_DST; side <<= 1) {
		if (!(t.dir & side)) continue;
		uint32_t f = (side == XE_DIR_SRC) ? t.fsrc : t.fdst;
		tile_get(T, f, r, has, fmax, hi, lo);
		if (((t.type == XE_BT_ADDR4) || (t.type == XE_BT_ADDR6)) && !has) continue;
		switch (t.type) {
			case XE_BT_ADDR4: {
				uint32_t a = (fmax == 16) ? (uint32_t)(hi >> 32) : (uint32_t)lo;
				for (uint32_t i = 0; i < t.nopnd; i++) {
					cu32 q = (cu32)(o + i);
					cu64 w = (cu64)(o + i);
					bool m;
					if (q[0]) m = lk_bit<LW>(lkbits, q[(side == XE_DIR_SRC) ? 2 : 3]);
					else m = (a & (uint32_t)w[5]) == (uint32_t)w[3];
					res |= m;
				}
				break;
			}
			case XE_BT_ADDR6:
				for (uint32_t i = 0; i < t.nopnd; i++) {
					cu32 q = (cu32)(o + i);
					cu64 w = (cu64)(o + i);
					bool m;
					if (q[0]) m = lk_bit<LW>(lkbits, q[(side == XE_DIR_SRC) ? 2 : 3]);
					else m = ((hi & w[4]) == w[2]) && ((lo & w[5]) == w[3]);
					res |= m;
				}
				break;
			case XE_BT_RANGE: {
				/* absent fields compare as 0; only 1/2/4-byte fields carry a value */
				int32_t v = (fmax <= 4) ? (int32_t)(uint32_t)lo : 0;
				res |= in_ranges_u(o, t.nopnd, v);
				break;
			}
			case XE_BT_MAC:
				for (uint32_t i = 0; i < t.nopnd; i++) res |= (((cu64)(o + i))[3] == lo);
				break;
			default:
				break;
		}
	}
	return res;
}

/* Plain copies of 2^LG wire bytes into a column (wire length == column
 * width): n micro-ops starting at fops[j0].  The next micro-op is fetched
 * while the current one is applied. */
/* XE_NT_STORES / XE_NT_LOADS: column stores / packet loads with the streaming
 * hint, so that what is touched once does not push the list tables out of the
 * caches.  Measured on MI355X: stores with the hint C2 0.2353 -> 0.2303 ms, C1
 * and C4 unchanged; loads with the hint C2 0.2715 ms, C1 0.0977 ms (the lanes
 * of a wave read neighbouring pieces of the same lines: the hint gives those
 * lines up too early); the hint on the last 16-byte piece of a record only,
 * when its lines have served the neighbouring lanes: C2 0.2013 against 0.2020
 * ms, C1 0.0892 against 0.0857 ms - nothing */
#ifndef XE_NT_STORES
#define XE_NT_STORES 1
#endif
#ifndef XE_NT_LOADS
#define XE_NT_LOADS 0
#endif
/* XE_NT_MASKS: the match masks (when a record has one word of them) and the
 * per-record field counts with the streaming hint as well.  Measured on MI355X:
 * C2 0.2312 -> 0.1981 ms, C1 0.0864 -> 0.0849 ms; with the hint on the 4-word
 * masks of C4 too, C4 0.3725 -> 0.4350 ms */
#ifndef XE_NT_MASKS
#define XE_NT_MASKS 1
#endif
/* an 8-byte store with the streaming hint that stays one: the compiler folds
 * "hinted store on one branch, plain store of the same value on the other"
 * into a single plain store */
__device__ __forceinline__ void
nt_store_u64(XE_GLOBAL uint64_t *p, uint64_t v)
{
	asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}
#if XE_NT_MASKS
#define XE_MASK_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define XE_MASK_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
#if XE_NT_STORES
#define XE_COL_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define XE_COL_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif

template <int R, int LG>
__device__ __forceinline__ void
copy_ops(const xe_kparams &kp, const uint32_t *stg, uint32_t j0, uint32_t n,
	const uint32_t (&ridx)[R], const bool (&live)[R])
{
	typedef uint64_t u64x8 __attribute__((ext_vector_type(8)));
	if (!n) return;
	/* byte offset of the record's element in a column of this width: 32 bits
	 * (the host keeps batches of the fast kernel below 2^28 records), so that
	 * a store is "uniform column base + per-lane offset" */
	uint32_t boff[R];
#pragma unroll
	for (int r = 0; r < R; r++) boff[r] = ridx[r] << LG;
	/* four micro-ops per scalar load (the host pads the list by four entries) */
	for (uint32_t j = 0; j < n; j += 4) {
		const u64x8 o = *(const XE_CONST u64x8 *)(kp.fops + j0 + j);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			if (j + k >= n) break;
			const uint32_t off = (uint32_t)o[2 * k] & 0xFFFF, sh = off & 3;
			const uint32_t *p = stg + (off >> 2) * (R * 64);
			XE_GLOBAL uint8_t *col = (XE_GLOBAL uint8_t *)o[2 * k + 1];
#pragma unroll
			for (int r = 0; r < R; r++) {
				const uint32_t *q = p + r * 64;
				XE_GLOBAL uint8_t *dst = col + boff[r];
				if (LG == 0) {
					const uint32_t w0 = q[0];
					if (live[r]) XE_COL_STORE(dst, (uint8_t)__builtin_amdgcn_alignbyte(w0, w0, sh));
				} else if (LG <= 2) {
					const uint32_t w0 = q[0], w1 = q[R * 64];
					const uint32_t v = __builtin_amdgcn_alignbyte(w1, w0, sh);
					if (live[r]) {
						if (LG == 1) XE_COL_STORE((XE_GLOBAL uint16_t *)dst, (uint16_t)v);
						else XE_COL_STORE((XE_GLOBAL uint32_t *)dst, v);
					}
				} else if (LG == 3) {
					const uint32_t w0 = q[0], w1 = q[R * 64], w2 = q[2 * R * 64];
					const uint32_t v0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
					const uint32_t v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
					if (live[r]) XE_COL_STORE((XE_GLOBAL uint64_t *)dst, (uint64_t)v0 | ((uint64_t)v1 << 32));
				} else {
					const uint32_t w0 = q[0], w1 = q[R * 64], w2 = q[2 * R * 64], w3 = q[3 * R * 64], w4 = q[4 * R * 64];
					u32x4 v;
					v.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
					v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
					v.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
					v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
					if (live[r]) XE_COL_STORE((XE_GLOBAL u32x4 *)dst, v);
				}
			}
		}
	}
}

/* Micro-ops whose wire length differs from the column width (and 6-byte MAC
 * values kept in 8-byte columns): mask, shift, store. */
template <int R>
__device__ void
general_ops(const xe_kparams &kp, const uint32_t *stg, uint32_t j0, uint32_t n,
	const uint32_t (&ridx)[R], const bool (&live)[R])
{
	for (uint32_t j = 0; j < n; j++) {
		cu64 xq = (cu64)(kp.fops + j0 + j);
		const uint64_t w0 = xq[0];
		const uint32_t off = (uint32_t)w0 & 0xFFFF, len = (uint32_t)(w0 >> 16) & 0xFF;
		const uint32_t kind = (uint32_t)(w0 >> 24) & 0xFF;
		const uint32_t lg = kind & XE_FOP_LGW;
		XE_GLOBAL uint8_t *col = (XE_GLOBAL uint8_t *)xq[1];
		const uint32_t pad = (kind & (XE_FOP_COPY | XE_FOP_MAC)) ? 0 : (8 * ((1u << lg) - len));

#pragma unroll
		for (int r = 0; r < R; r++) {
			uint32_t a[4];
			stage_read<R>(stg, r, off, len, a);
			if (!live[r]) continue;
			/* column bytes = the field as the reference stores it: the wire
			 * bytes right-aligned in the field's maximum size */
			if (lg <= 2) {
				uint32_t v = a[0];
				if (!(kind & XE_FOP_COPY)) v = (v & (0xFFFFFFFFu >> (32 - 8 * len))) << pad;
				if (lg == 0) XE_COL_STORE(&col[ridx[r]], (uint8_t)v);
				else if (lg == 1) XE_COL_STORE(&((XE_GLOBAL uint16_t *)col)[ridx[r]], (uint16_t)v);
				else XE_COL_STORE(&((XE_GLOBAL uint32_t *)col)[ridx[r]], v);
			} else if (lg == 3) {
				uint64_t v = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
				if (!(kind & XE_FOP_COPY)) v = (v & (~0ull >> (64 - 8 * len))) << pad;
				XE_COL_STORE(&((XE_GLOBAL uint64_t *)col)[ridx[r]], v);
			} else {
				uint64_t v0 = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
				uint64_t v1 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
				if (!(kind & XE_FOP_COPY)) {
					/* len < 16: shift the 128-bit little-endian value up by pad bits */
					if (len <= 8) { v1 = 0; v0 &= ~0ull >> (64 - 8 * len); }
					else v1 &= ~0ull >> (128 - 8 * len);
					if (pad >= 64) { v1 = v0 << (pad - 64); v0 = 0; }
					else if (pad) { v1 = (v1 << pad) | (v0 >> (64 - pad)); v0 <<= pad; }
				}
				XE_COL_STORE(&((XE_GLOBAL uint64_t *)col)[2 * (size_t)ridx[r]], v0);
				XE_COL_STORE(&((XE_GLOBAL uint64_t *)col)[2 * (size_t)ridx[r] + 1], v1);
			}
		}
	}
}

/* one listed object program for record r of all lanes */
template <int R>
__device__ __forceinline__ bool
prog_eval(const xe_kparams &kp, const uint32_t (&FP)[12], uint64_t lkbits, const uint64_t (&tb)[4])
{
	const uint32_t fkind = FP[2];
	if (fkind == XE_FPROG_MASK) return (lkbits & ((uint64_t)FP[0] | ((uint64_t)FP[1] << 32))) != 0;
	if (fkind == XE_FPROG_TERM) {
		const uint32_t t = FP[3], wsel = t >> 6;
		const uint64_t word = (wsel == 0) ? tb[0] : (wsel == 1) ? tb[1] : (wsel == 2) ? tb[2] : tb[3];
		return ((word >> (t & 63)) & 1) != 0;
	}
	if (fkind == XE_FPROG_CONST) return FP[7] != 0;
	uint64_t st = 0;
	for (uint32_t k = 0; k < FP[5]; k++) {
		const uint32_t op = ((cu32)kp.ops)[FP[4] + k], code = op & 3;
		if (code == XE_OP_BASIC) {
			const uint32_t id = op >> 2, wsel = id >> 6;
			const uint64_t word = (wsel == 0) ? tb[0] : (wsel == 1) ? tb[1] : (wsel == 2) ? tb[2] : tb[3];
			st = (st << 1) | ((word >> (id & 63)) & 1);
		} else if (code == XE_OP_NOT) {
			st ^= 1;
		} else {
			const uint64_t top = st & 1;
			st >>= 1;
			if (code == XE_OP_AND) st &= (top | ~1ull);
			else st |= top;
		}
	}
	return (st & 1) != 0;
}

/* per lane: where its records of a tile are and what their exporter says */
template <int R>
struct tile_lanes
{
	uint32_t voff[R], ridx[R], rate[R], dev_ip[R], dev_id[R];
	uint32_t dev[R];          /* devices.conf entry + 1 when it has mark filters or drops unmarked records, else 0 */
	bool live[R], hasdev[R];  /* hasdev: the exporter has a devices.conf entry */
	uint32_t tmpl, nchunk;    /* wave-uniform */
};

/* which run, which record: the runs that overlap the tile are walked with
 * scalar loads (the next run's descriptor is fetched while the current one is
 * applied); each lane keeps the run it falls into */
template <int R>
__device__ __forceinline__ void
tile_setup(const xe_kparams &kp, uint32_t tile, uint32_t lane, bool want_dev, tile_lanes<R> &S)
{
	typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
	cu32 TQ = (cu32)(kp.ftiles + tile);
	const uint32_t tbase = TQ[2], s_end = kp.nfruns - 1;
	uint32_t s = TQ[0];
	u32x8 RN = *(const XE_CONST u32x8 *)(kp.fruns + s);

	S.tmpl = TQ[1];
	S.nchunk = ((cu32)(kp.fplans + S.tmpl))[8];
#pragma unroll
	for (int r = 0; r < R; r++) {
		S.live[r] = false;
		S.hasdev[r] = false;
		S.voff[r] = S.ridx[r] = S.rate[r] = 0;
		S.dev_ip[r] = S.dev_id[r] = S.dev[r] = 0;
	}
	for (;;) {
		const uint32_t start = RN[2], nrec = RN[3];
		if (start >= tbase + 64 * R) break;
		const u32x8 NX = *(const XE_CONST u32x8 *)(kp.fruns + ((s < s_end) ? s + 1 : s_end));
#pragma unroll
		for (int r = 0; r < R; r++) {
			const uint32_t sub0 = tbase + 64 * r;
			if ((start + nrec <= sub0) || (start >= sub0 + 64)) continue;
			const uint32_t g = sub0 + lane, i = g - start;
			const bool in = (g >= start) && (i < nrec);
			S.voff[r] = in ? (RN[0] + i * (RN[4] & 0xFFFFu)) : S.voff[r];
			S.hasdev[r] = in ? ((RN[4] >> 31) != 0) : S.hasdev[r];
			S.dev[r] = in ? ((RN[4] >> 16) & 0x7FFFu) : S.dev[r];
			S.ridx[r] = in ? (RN[1] + i) : S.ridx[r];
			S.rate[r] = in ? RN[5] : S.rate[r];
			if (want_dev) {
				S.dev_ip[r] = in ? RN[6] : S.dev_ip[r];
				S.dev_id[r] = in ? RN[7] : S.dev_id[r];
			}
			S.live[r] |= in;
		}
		if (s >= s_end) break;
		s++;
		RN = NX;
	}
}

/* start the loads of the first (up to) four 16-byte pieces of every record */
template <int R>
__device__ __forceinline__ void
tile_issue(const xe_kparams &kp, const tile_lanes<R> &S, u32x4 (&q)[4][R])
{
	const XE_GLOBAL uint8_t *pkt = (const XE_GLOBAL uint8_t *)kp.pkt;
#pragma unroll
	for (int i = 0; i < 4; i++) {
#pragma unroll
		for (int r = 0; r < R; r++) {
			q[i][r] = (u32x4){0, 0, 0, 0};
			if (S.live[r] && ((uint32_t)i < S.nchunk) && !XE_ABL(32)) {
#if XE_NT_LOADS
				typedef u32x4 u32x4_any __attribute__((aligned(1)));
				q[i][r] = __builtin_nontemporal_load((const XE_GLOBAL u32x4_any *)(pkt + S.voff[r] + 16 * i));
#else
				__builtin_memcpy(&q[i][r], (const uint8_t *)pkt + S.voff[r] + 16 * i, 16);
#endif
			}
		}
	}
}

/* every IP-list membership (lookup bits), then every term some program needs
 * (term bits), once per record */
#ifndef XE_RG_STEP
#define XE_RG_STEP 2     /* range groups / fail-table groups advanced together (the host pads to 4) */
#endif
#ifndef XE_CJ_STEP
#define XE_CJ_STEP 2
#endif
#ifndef XE_RG_SCALAR
#define XE_RG_SCALAR 24  /* range groups of up to this many breakpoints are searched with scalar loads */
#endif

/* Range terms by field: one search of the field's value among the group's
 * breakpoints answers every range term that reads the field (a table entry
 * per interval holds the terms' bits).  Four groups advance together so their
 * dependent loads overlap. */
template <int R, typename TT, int GC>
__device__ __forceinline__ void
range_groups(const xe_kparams &kp, const TT &T, uint64_t (&tb)[R][4], const lds_u64 *lmt)
{
	typedef uint32_t u32x4r __attribute__((ext_vector_type(4)));

	for (uint32_t g0 = 0; g0 < kp.nrgroups; g0 += GC) {
		uint32_t n[GC], bpo[GC], tbo[GC], pos[GC][R], maxn = 1;
		int32_t v[GC][R];
#pragma unroll
		for (int c = 0; c < GC; c++) {
			cu32 q = (cu32)(kp.rgroups + g0 + c);
			const uint32_t field = q[0];
			/* a group of the padding: nothing to search, entry 0 of a table that
			 * is added again */
			const bool used = g0 + c < kp.nrgroups_used;
			n[c] = used ? q[1] : 0u;
			bpo[c] = q[2];
			tbo[c] = q[3];
			if (n[c] > maxn) maxn = n[c];
#pragma unroll
			for (int r = 0; r < R; r++) {
				bool has;
				uint32_t fmax;
				uint64_t hi, lo;
				v[c][r] = 0;
				pos[c][r] = 0;
				if (!used) continue;
				tile_get(T, field, r, has, fmax, hi, lo);
				/* absent fields compare as 0; only 1/2/4-byte fields carry a value */
				v[c][r] = (fmax <= 4) ? (int32_t)(uint32_t)lo : 0;
			}
		}
		/* pos = number of breakpoints <= v.  A few breakpoints: they are the
		 * same for every lane, so they come through the scalar cache and are
		 * counted with compares, no dependent gathers; many: binary search */
		if (maxn <= XE_RG_SCALAR) {
			for (uint32_t i = 0; i < maxn; i++) {
#pragma unroll
				for (int c = 0; c < GC; c++) {
					const bool in = i < n[c];
					const int32_t bi = ((const XE_CONST int32_t *)kp.rgdata)[bpo[c] + (in ? i : 0u)];
#pragma unroll
					for (int r = 0; r < R; r++) pos[c][r] += (in && (bi <= v[c][r])) ? 1u : 0u;
				}
			}
		} else
		for (uint32_t step = 1u << (31 - __builtin_clz(maxn)); step; step >>= 1) {
			int32_t b[GC][R];
#pragma unroll
			for (int c = 0; c < GC; c++) {
#pragma unroll
				for (int r = 0; r < R; r++) {
					const uint32_t cand = pos[c][r] + step;
					/* (a group of the padding has n = 0: it reads word 0 and keeps pos = 0) */
					const uint32_t top = (cand < n[c]) ? cand : n[c];
					const uint32_t idx = top ? top - 1 : 0u;
					b[c][r] = ((const XE_CONST int32_t *)kp.rgdata)[bpo[c] + idx];
				}
			}
#pragma unroll
			for (int c = 0; c < GC; c++) {
#pragma unroll
				for (int r = 0; r < R; r++) {
					const uint32_t cand = pos[c][r] + step;
					if ((cand <= n[c]) && (b[c][r] <= v[c][r])) pos[c][r] = cand;
				}
			}
		}
		u32x4r ea[GC][R], eb[GC][R];
		/* small tables ride behind the block's copy of the atom table in LDS:
		 * the entry is there in a hundred cycles, without a gather that waits
		 * its turn behind every vector memory operation before it */
		const bool in_lds = (kp.rg_lds != 0) && (kp.motab_lds != 0);
#pragma unroll
		for (int c = 0; c < GC; c++) {
#pragma unroll
			for (int r = 0; r < R; r++) {
				ea[c][r] = eb[c][r] = (u32x4r){0, 0, 0, 0};
				if (g0 + c >= kp.nrgroups_used) continue;
				if (in_lds) {
					const lds_u64 *e = lmt + kp.rg_lds + ((tbo[c] + (size_t)pos[c][r] * 8) >> 1);
					const uint64_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
					ea[c][r] = (u32x4r){ (uint32_t)e0, (uint32_t)(e0 >> 32), (uint32_t)e1, (uint32_t)(e1 >> 32) };
					eb[c][r] = (u32x4r){ (uint32_t)e2, (uint32_t)(e2 >> 32), (uint32_t)e3, (uint32_t)(e3 >> 32) };
				} else {
					const XE_CONST u32x4r *e = (const XE_CONST u32x4r *)(kp.rgdata + tbo[c] + (size_t)pos[c][r] * 8);
					ea[c][r] = e[0];
					eb[c][r] = e[1];
				}
			}
		}
#pragma unroll
		for (int c = 0; c < GC; c++) {
#pragma unroll
			for (int r = 0; r < R; r++) {
				if (!T.live[r]) continue;
				tb[r][0] |= (uint64_t)ea[c][r].x | ((uint64_t)ea[c][r].y << 32);
				tb[r][1] |= (uint64_t)ea[c][r].z | ((uint64_t)ea[c][r].w << 32);
				tb[r][2] |= (uint64_t)eb[c][r].x | ((uint64_t)eb[c][r].y << 32);
				tb[r][3] |= (uint64_t)eb[c][r].z | ((uint64_t)eb[c][r].w << 32);
			}
		}
	}
}

/* LEAN = the configuration has no range groups, conjunctive or listed objects
 * (objects with a single condition and no parent only): those parts are
 * compiled out, which keeps the common simple case small */
template <int R, bool LEAN, int LW, typename TT, int RGC = XE_RG_STEP, int PC = 2>
__device__ __forceinline__ void
classify_atoms(const xe_kparams &kp, const TT &T, const uint64_t (&plist)[LW], uint64_t (&lk)[R][LW], uint64_t (&tb)[R][4],
	const lds_u64 *lmt)
{
		if XE_ABL(1) {
#pragma unroll
			for (int r = 0; r < R; r++) {
#pragma unroll
				for (int w = 0; w < LW; w++) lk[r][w] = 0;
			}
		} else {
			XE_PHASE_DECL;
			const uint32_t lane = threadIdx.x & 63;
			fast_lookups<R, LW, TT, PC>(kp, T, plist, lk);
			XE_PHASE(6);
		}
#pragma unroll
		for (int r = 0; r < R; r++) tb[r][0] = tb[r][1] = tb[r][2] = tb[r][3] = 0;
		if (!LEAN && !XE_ABL(2)) {
			XE_PHASE_DECL;
			const uint32_t lane = threadIdx.x & 63;
			range_groups<R, TT, RGC>(kp, T, tb, lmt);
			XE_PHASE(7);
		}
		for (uint32_t ti = 0; (ti < kp.ntneed) && !XE_ABL(2); ti++) {
			const uint32_t t = ((cu32)kp.tneed)[ti];
			cu64 fq = (cu64)(kp.fterms + t);
			const uint64_t fkind = fq[0];
			const term_s tm = ld_term(kp, t);
#pragma unroll
			for (int r = 0; r < R; r++) {
				bool res;
				/* all operands IP lists: a mask over the lookup bits */
				if (fkind & 1) {
					uint64_t any = 0;
#pragma unroll
					for (int w = 0; w < LW; w++) any |= lk[r][w] & fq[1 + w];
					res = any != 0;
				} else {
					res = fast_term<R, LW, TT>(kp, T, tm, r, lk[r]);
				}
				const uint64_t bit = (uint64_t)(T.live[r] && res) << (t & 63);
				switch (t >> 6) {
					case 0: tb[r][0] |= bit; break;
					case 1: tb[r][1] |= bit; break;
					case 2: tb[r][2] |= bit; break;
					default: tb[r][3] |= bit; break;
				}
			}
		}

}

/* from lookup and term bits to the match masks and counters of a record:
 * the atom table, the conjunctive objects (need / forbid rows), the listed
 * objects; lcnt / lmt = the block's LDS counters and atom-table copy */
template <int R, bool LEAN, int LW>
__device__ __forceinline__ void
classify_objects(const xe_kparams &kp, const bool (&live)[R], const uint64_t (&lk)[R][LW], const uint64_t (&tb)[R][4],
	const uint32_t (&ridx)[R], const unsigned long long (&bxr)[R], const unsigned long long (&pxr)[R],
	lds_u64 *lcnt, const lds_u64 *lmt)
{
		/* derived atoms: the multi-literal clauses of the conjunctive objects */
		uint64_t dbits[R];
#pragma unroll
		for (int r = 0; r < R; r++) dbits[r] = 0;
		for (uint32_t j = 0; !LEAN && (j < kp.ncjderived) && !XE_ABL(8); j++) {
			/* xe_cjderived: pos_lk[4], neg_lk[4], pos_tb, neg_tb */
			cu64 dq = (cu64)(kp.cjderived + j);
			const uint64_t pt = dq[8], nt = dq[9];
#pragma unroll
			for (int r = 0; r < R; r++) {
				uint64_t any = (tb[r][0] & pt) | (~tb[r][0] & nt);
#pragma unroll
				for (int w = 0; w < LW; w++) any |= (lk[r][w] & dq[w]) | (~lk[r][w] & dq[4 + w]);
				dbits[r] |= (uint64_t)(any != 0) << j;
			}
		}

		/* the conjunctive objects: an object fails when one of its needed atoms
		 * is false or a forbidden one true; tables indexed by 8 atoms at once
		 * give the failing objects.  A table row holds all mask words side by
		 * side: up to four of them are fetched per visit, for all words at once */
		uint64_t failw[R][4];
		const bool cj_rows = !LEAN && kp.ncjgroups && (kp.mask_words <= 4) && !XE_ABL(8);
#pragma unroll
		for (int r = 0; r < R; r++) failw[r][0] = failw[r][1] = failw[r][2] = failw[r][3] = 0;
		if (cj_rows) {
			typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
			const uint32_t mw = kp.mask_words;
			for (uint32_t g0 = 0; g0 < kp.ncjgroups; g0 += XE_CJ_STEP) {
				u64x2 ea[XE_CJ_STEP][R], eb[XE_CJ_STEP][R];
#pragma unroll
				for (int c = 0; c < XE_CJ_STEP; c++) {
					cu32 gq = (cu32)(kp.cjgb + 2 * (g0 + c));
					const uint32_t k = gq[0], ti = gq[1];
#pragma unroll
					for (int r = 0; r < R; r++) {
						/* byte k: 0-31 lookup bits, 32-63 term bits, 64-71 derived */
						uint64_t src = lk[r][0];
#pragma unroll
						for (int q = 1; q < LW; q++) src = ((k >> 3) == (uint32_t)q) ? lk[r][q] : src;
						if (k >= 32) src = tb[r][0];
						if (k >= 40) src = tb[r][1];
						if (k >= 48) src = tb[r][2];
						if (k >= 56) src = tb[r][3];
						if (k >= 64) src = dbits[r];
						const uint32_t by = (uint32_t)(src >> ((k & 7) * 8)) & 0xFF;
						const XE_CONST uint64_t *row = (const XE_CONST uint64_t *)kp.cjtab + ((size_t)ti * 256 + by) * mw;
						ea[c][r] = (u64x2){0, 0};
						eb[c][r] = (u64x2){0, 0};
						if (mw == 4) {
							ea[c][r] = *(const XE_CONST u64x2 *)row;
							eb[c][r] = *(const XE_CONST u64x2 *)(row + 2);
						} else if (mw == 3) {
							ea[c][r] = *(const XE_CONST u64x2 *)row;
							eb[c][r].x = row[2];
						} else if (mw == 2) {
							ea[c][r] = *(const XE_CONST u64x2 *)row;
						} else {
							ea[c][r].x = row[0];
						}
					}
				}
#pragma unroll
				for (int c = 0; c < XE_CJ_STEP; c++) {
#pragma unroll
					for (int r = 0; r < R; r++) {
						failw[r][0] |= ea[c][r].x;
						failw[r][1] |= ea[c][r].y;
						failw[r][2] |= eb[c][r].x;
						failw[r][3] |= eb[c][r].y;
					}
				}
			}
		}

		/* object programs: the atom table first, then the conjunctive objects
		 * (need / forbid rows), then the listed objects */
		uint32_t bi = 0;
		for (uint32_t w = 0; (w < kp.mask_words) && !XE_ABL(8); w++) {
			uint64_t bits[R];
			const uint32_t m0 = w * 64, m1 = (m0 + 64 < kp.nmo) ? m0 + 64 : kp.nmo;
			/* atoms whose row for this word is not empty */
			/* table of this mask word: 64 * LW lookup rows, 64 term rows; behind
			 * all tables per mask word LW + 1 words: the rows that are not empty */
			const uint32_t mrows = 64 * LW + 64;
			const size_t any_at = (size_t)kp.mask_words * mrows + (size_t)(LW + 1) * w;
			uint64_t any_lk[LW];
#pragma unroll
			for (int k = 0; k < LW; k++) any_lk[k] = kp.motab_lds ? lmt[any_at + k] : ((cu64)kp.motab)[any_at + k];
			const uint64_t any_tb = kp.motab_lds ? lmt[any_at + LW] : ((cu64)kp.motab)[any_at + LW];
#pragma unroll
			for (int r = 0; r < R; r++) {
				uint64_t acc = 0;
				if (live[r]) {
					uint64_t x;
#pragma unroll
					for (int k = 0; k < LW; k++) {
						x = lk[r][k] & any_lk[k];
						while (x) {
							const uint32_t a = 64 * k + (uint32_t)__builtin_ctzll(x);
							x &= x - 1;
							acc |= kp.motab_lds ? lmt[w * mrows + a] : kp.motab[w * mrows + a];
						}
					}
					x = tb[r][0] & any_tb;
					while (x) {
						const uint32_t a = 64 * LW + (uint32_t)__builtin_ctzll(x);
						x &= x - 1;
						acc |= kp.motab_lds ? lmt[w * mrows + a] : kp.motab[w * mrows + a];
					}
				}
				bits[r] = acc;
			}
			if (!LEAN && kp.ncjgroups) {
				/* the conjunctive objects: an object fails when one of its needed
				 * atoms is false or a forbidden one true; tables indexed by 8
				 * atoms at once give the failing objects, 4 groups per step */
				uint64_t fail[R];
#pragma unroll
				for (int r = 0; r < R; r++) {
					fail[r] = (w == 0) ? failw[r][0] : (w == 1) ? failw[r][1] : (w == 2) ? failw[r][2] : failw[r][3];
				}
				for (uint32_t g0 = 0; !cj_rows && (g0 < kp.ncjgroups); g0 += XE_CJ_STEP) {
					uint64_t e[XE_CJ_STEP][R];
#pragma unroll
					for (int c = 0; c < XE_CJ_STEP; c++) {
						cu32 gq = (cu32)(kp.cjgb + 2 * (g0 + c));
						const uint32_t k = gq[0], ti = gq[1];
#pragma unroll
						for (int r = 0; r < R; r++) {
							/* byte k: 0-31 lookup bits, 32-63 term bits, 64-71 derived */
							uint64_t src = lk[r][0];
#pragma unroll
							for (int q = 1; q < LW; q++) src = ((k >> 3) == (uint32_t)q) ? lk[r][q] : src;
							if (k >= 32) src = tb[r][0];
							if (k >= 40) src = tb[r][1];
							if (k >= 48) src = tb[r][2];
							if (k >= 56) src = tb[r][3];
							if (k >= 64) src = dbits[r];
							const uint32_t by = (uint32_t)(src >> ((k & 7) * 8)) & 0xFF;
							e[c][r] = ((const XE_CONST uint64_t *)kp.cjtab)[((size_t)ti * 256 + by) * kp.mask_words + w];
						}
					}
#pragma unroll
					for (int c = 0; c < XE_CJ_STEP; c++) {
#pragma unroll
						for (int r = 0; r < R; r++) fail[r] |= e[c][r];
					}
				}
				const uint64_t cand = ((cu64)kp.cjcand)[w];
#pragma unroll
				for (int r = 0; r < R; r++) {
					if (live[r]) bits[r] |= cand & ~fail[r];
				}
			}
			while (!LEAN && (bi < kp.nfprogs)) {
				typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
				const XE_CONST u32x4v *fq = (const XE_CONST u32x4v *)(kp.fprogs + bi);
				const u32x4v f0 = fq[0], f1 = fq[1], f2 = fq[2];
				const uint32_t FP[12] = { f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w };
				const uint32_t m = FP[8];
				const int32_t parent = (int32_t)FP[6];
				if (m >= m1) break;
				bi++;
#pragma unroll
				for (int r = 0; r < R; r++) {
					bool h;
					if (FP[2] == XE_FPROG_GATE) h = ((bits[r] >> (m - m0)) & 1) != 0;   /* matched above */
					else h = prog_eval<R>(kp, FP, lk[r][0], tb[r]) && live[r];
					if (parent >= 0) {
						const uint32_t par = (uint32_t)parent;
						bool pok;
						if (par >= m0) pok = ((bits[r] >> (par - m0)) & 1) != 0;
						else pok = live[r] && (((kp.masks[(size_t)ridx[r] * kp.mask_words + (par >> 6)] >> (par & 63)) & 1) != 0);
						h = h && pok;
					}
					bits[r] = (bits[r] & ~(1ull << (m - m0))) | ((uint64_t)h << (m - m0));
				}
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
				if (!live[r]) continue;
				/* one mask word per record: consecutive records fill whole lines, nothing
				 * reads them back here, so they stream out past the list tables in the
				 * caches; with several words a record's words are stored one pass after
				 * the other (and parents in earlier words are read back): plain stores */
				XE_GLOBAL uint64_t *mp = &((XE_GLOBAL uint64_t *)kp.masks)[(size_t)ridx[r] * kp.mask_words + w];
				if (XE_NT_MASKS && (kp.mask_words == 1)) nt_store_u64(mp, bits[r]);
				else *mp = bits[r];
				/* counters: every lane adds for each object it matched */
				uint64_t x = XE_ABL(4) ? 0 : bits[r];
				while (x) {
					const uint32_t m = m0 + (uint32_t)__builtin_ctzll(x);
					x &= x - 1;
					if (kp.lds_counters) {
						/* the block keeps lds_counters copies of the counters (a power of
						 * two), a lane adds to the copy of its number: lanes that matched
						 * the same object take their turns at one address otherwise */
						lds_u64 *mine = lcnt + (size_t)(threadIdx.x & (kp.lds_counters - 1)) * kp.nmo * 3;
						__hip_atomic_fetch_add(mine + 3 * m + 0, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						__hip_atomic_fetch_add(mine + 3 * m + 1, bxr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						__hip_atomic_fetch_add(mine + 3 * m + 2, pxr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					} else {
						atomicAdd(kp.counters + 3 * m + 0, 1ull);
						atomicAdd(kp.counters + 3 * m + 1, bxr[r]);
						atomicAdd(kp.counters + 3 * m + 2, pxr[r]);
					}
				}
			}
		}
}

/* measured on MI355X: the lean kernel at 6 waves per SIMD is 3-8 % faster than at the compiler's own choice */
#ifndef XE_FAST_WAVES
#define XE_FAST_WAVES 6
#endif
#ifndef XE_PREFETCH_LATE
#define XE_PREFETCH_LATE 1
#endif
template <int R, bool LEAN, int LW>
__global__ void __launch_bounds__(XE_BLOCK)
#if XE_FAST_WAVES
__attribute__((amdgpu_waves_per_eu((R == 1 && LW == 1) ? XE_FAST_WAVES : 1)))
#endif
xe_nf_fast_kernel(const xe_kparams kp)
{
	XE_PHASE_DECL;
	extern __shared__ uint32_t lds32[];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* staging: kp.fast_dwords dwords per record, [wave][dword][r][lane]; then
	 * the block's counters and its copy of the atom table */
	uint32_t *stg = lds32 + (size_t)wave * kp.fast_dwords * R * 64 + lane;
	lds_u64 *lcnt = (lds_u64 *)(lds32 + (size_t)(XE_BLOCK / 64) * kp.fast_dwords * R * 64);
	lds_u64 *lmt = lcnt + (size_t)kp.lds_counters * kp.nmo * 3;
	const bool want_dev = kp.fast_want_dev != 0;
	const XE_GLOBAL uint8_t *pkt = (const XE_GLOBAL uint8_t *)kp.pkt;

	if (kp.lds_counters) {
		for (uint32_t i = threadIdx.x; i < kp.nmo * 3 * kp.lds_counters; i += XE_BLOCK) lcnt[i] = 0;
	}
	if (kp.motab_lds) {
		for (uint32_t i = threadIdx.x; i < kp.motab_words; i += XE_BLOCK) lmt[i] = kp.motab[i];
	}
	__syncthreads();

	/* The tile list is cut into 8 contiguous parts, one per XCD (blocks are
	 * dealt to the XCDs round-robin), so that tiles which write neighbouring
	 * parts of the same output lines run behind the same L2 and their partial
	 * lines merge there.  Only speed depends on the dealing order: every
	 * tile belongs to exactly one (block, wave, iteration). */
	const uint32_t part = blockIdx.x & 7, lblk = blockIdx.x >> 3;
	const uint32_t nblk = (gridDim.x + 7 - part) >> 3;
	const uint32_t per = (kp.ntiles + 7) >> 3;
	const uint32_t p0 = part * per, p1 = (p0 + per < kp.ntiles) ? p0 + per : kp.ntiles;
	const uint32_t wave0 = __builtin_amdgcn_readfirstlane(p0 + lblk * (XE_BLOCK / 64) + wave);
	/* Tiles are pipelined: while a tile is classified and written out, the
	 * packet loads of the wave's next tile are already in flight (they sit in
	 * registers until the staging area is free again). */
	const uint32_t tstride = nblk * (XE_BLOCK / 64);
	tile_lanes<R> C, N;
	u32x4 q[4][R];
	if (wave0 < p1) {
		tile_setup<R>(kp, wave0, lane, want_dev, C);
		tile_issue<R>(kp, C, q);
		N = C;
	}
	for (uint32_t tile = wave0; tile < p1; tile += tstride) {
		const uint32_t tmpl = C.tmpl, nchunk = C.nchunk;
		cu32 PL = (cu32)(kp.fplans + tmpl);
		const uint32_t op_base = PL[0], nparsed = PL[7];
		/* xe_fplan: the probe lists are its 64-bit words 5..8 */
		uint64_t plist[LW];
#pragma unroll
		for (int k = 0; k < LW; k++) plist[k] = ((cu64)PL)[5 + k];
		fast_tile<R> T;
		uint32_t ridx[R], rate[R];

		T.stg = stg;
		T.fld = (cu32)(kp.fflds + (size_t)tmpl * XE_NFIELDS);
#pragma unroll
		for (int r = 0; r < R; r++) {
			T.live[r] = C.live[r];
			T.dev_ip[r] = C.dev_ip[r];
			T.dev_id[r] = C.dev_id[r];
			T.mark[r] = 0;
			T.markhas[r] = false;
			ridx[r] = C.ridx[r];
			rate[r] = C.rate[r];
		}

		/* stage the records in LDS: the pieces loaded ahead, then (records
		 * longer than 64 bytes) the rest */
#pragma unroll
		for (int i = 0; i < 4; i++) {
			if ((uint32_t)i >= nchunk) break;
#pragma unroll
			for (int r = 0; r < R; r++) {
				uint32_t *d = stg + ((4 * i) * R + r) * 64;
				d[0] = q[i][r].x;
				d[R * 64] = q[i][r].y;
				d[2 * R * 64] = q[i][r].z;
				d[3 * R * 64] = q[i][r].w;
			}
		}
		for (uint32_t k = 4; k < nchunk; k++) {
#pragma unroll
			for (int r = 0; r < R; r++) {
				u32x4 x = {0, 0, 0, 0};
				if (C.live[r] && !XE_ABL(32)) __builtin_memcpy(&x, (const uint8_t *)pkt + C.voff[r] + 16 * k, 16);
				uint32_t *d = stg + ((4 * k) * R + r) * 64;
				d[0] = x.x;
				d[R * 64] = x.y;
				d[2 * R * 64] = x.z;
				d[3 * R * 64] = x.w;
			}
		}
		XE_PHASE(0);
#if !XE_PREFETCH_LATE
		if (tile + tstride < p1) {
			tile_setup<R>(kp, tile + tstride, lane, want_dev, N);
			tile_issue<R>(kp, N, q);
		}
#endif

		XE_PHASE(1);
		unsigned long long bxr[R], pxr[R];
#pragma unroll
		for (int r = 0; r < R; r++) {
			bool has;
			uint32_t fmax;
			uint64_t hi, b, p;
			/* what the counters add (monit-objects.c:995-1007, scale 1) */
			fld_get<R>(T, XF_in_bytes, r, has, fmax, hi, b);
			fld_get<R>(T, XF_in_pkts, r, has, fmax, hi, p);
			bxr[r] = b * rate[r];
			pxr[r] = p * rate[r];
		}

		/* every IP-list membership, then every term some program needs, once */
		uint64_t lk[R][LW], tb[R][4];
		classify_atoms<R, LEAN, LW>(kp, T, plist, lk, tb, lmt);
		XE_PHASE(2);
#if XE_PREFETCH_LATE
		/* the next tile's packet loads go out behind the list walks: vector
		 * loads return in order, so a walk issued behind them would wait for
		 * the packet bytes to come in from HBM first */
		if (tile + tstride < p1) {
			tile_setup<R>(kp, tile + tstride, lane, want_dev, N);
			tile_issue<R>(kp, N, q);
		}
#endif
		XE_PHASE(3);
		/* decode: the template's micro-ops, sorted by kind.  The stores come
		 * after the list walks so that no gather waits behind them (the vector
		 * memory counter retires in order).  Issued behind the first root
		 * gathers instead, to fill that wait: C2 0.2135 against 0.2019 ms, C1
		 * 0.0890 against 0.0856 ms - the later gathers then wait for them. */
#pragma unroll
		for (int r = 0; r < R; r++) {
			if (T.live[r]) {
				XE_MASK_STORE(&((XE_GLOBAL uint8_t *)kp.nparsed)[ridx[r]], (uint8_t)nparsed);
				if (kp.rates) ((XE_GLOBAL uint32_t *)kp.rates)[ridx[r]] = rate[r];
				if (kp.devips) ((XE_GLOBAL uint32_t *)kp.devips)[ridx[r]] = __builtin_bswap32(C.dev_ip[r]);
			}
		}
		if (!XE_ABL(16)) {
			uint32_t j = op_base;
			copy_ops<R, 0>(kp, stg, j, PL[1], ridx, T.live); j += PL[1];
			copy_ops<R, 1>(kp, stg, j, PL[2], ridx, T.live); j += PL[2];
			copy_ops<R, 2>(kp, stg, j, PL[3], ridx, T.live); j += PL[3];
			copy_ops<R, 3>(kp, stg, j, PL[4], ridx, T.live); j += PL[4];
			copy_ops<R, 4>(kp, stg, j, PL[5], ridx, T.live); j += PL[5];
			if (PL[6]) general_ops<R>(kp, stg, j, PL[6], ridx, T.live);
		}
		XE_PHASE(4);
		/* device_rules_check() (devices.c:286-314): the mark filters of the
		 * record's devices.conf entry are counted, a record that stays unmarked
		 * where the entry says so is dropped, and the objects then see the mark.
		 * The filters are wave-uniform programs over the term and lookup bits:
		 * one pass per distinct entry among the tile's lanes. */
		bool clive[R];
#pragma unroll
		for (int r = 0; r < R; r++) {
			uint32_t cnt = 0;
			bool drop = false;
			unsigned long long pend = LEAN ? 0ull : __ballot(T.live[r] && (C.dev[r] != 0));
			while (pend) {
				const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)C.dev[r], (int)__builtin_ctzll(pend));
				const bool mine = T.live[r] && (C.dev[r] == d1);
				cu32 DR = (cu32)(kp.devrules + (d1 - 1));
				const uint32_t first = DR[0] - kp.nmo, np = DR[1], skip = DR[2];
				uint32_t c = 0;
				for (uint32_t i = 0; i < np; i++) {
					typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
					const XE_CONST u32x4v *fq = (const XE_CONST u32x4v *)(kp.mark_fprogs + first + i);
					const u32x4v f0 = fq[0], f1 = fq[1], f2 = fq[2];
					const uint32_t FP[12] = { f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w };
					c += prog_eval<R>(kp, FP, lk[r][0], tb[r]) ? 1u : 0u;
				}
				if (mine) {
					cnt = c;
					drop = skip && (c == 0);
				}
				pend &= ~__ballot(mine);
			}
			T.mark[r] = cnt;
			T.markhas[r] = C.hasdev[r] && !drop;
			clive[r] = T.live[r] && !drop;
			if (T.live[r] && kp.marks) {
				XE_MASK_STORE(&((XE_GLOBAL uint32_t *)kp.marks)[ridx[r]], drop ? (uint32_t)XE_MARK_DROPPED
					: C.hasdev[r] ? (uint32_t)(XE_MARK_HAS | (cnt & 0x3FFFFFFFu)) : 0u);
			}
		}
		/* the terms that read the mark, now that it is known */
		for (uint32_t ti = 0; !LEAN && (ti < kp.ntmark); ti++) {
			const uint32_t t = ((cu32)kp.tmark)[ti];
			const term_s tm = ld_term(kp, t);
#pragma unroll
			for (int r = 0; r < R; r++) {
				const uint64_t bit = (uint64_t)(clive[r] && fast_term<R, LW, fast_tile<R>>(kp, T, tm, r, lk[r])) << (t & 63);
				const uint64_t clr = ~(1ull << (t & 63));
				switch (t >> 6) {
					case 0: tb[r][0] = (tb[r][0] & clr) | bit; break;
					case 1: tb[r][1] = (tb[r][1] & clr) | bit; break;
					case 2: tb[r][2] = (tb[r][2] & clr) | bit; break;
					default: tb[r][3] = (tb[r][3] & clr) | bit; break;
				}
			}
		}
		classify_objects<R, LEAN, LW>(kp, clive, lk, tb, ridx, bxr, pxr, lcnt, lmt);
		if (!LEAN) {
			/* a dropped record still owns its mask words */
#pragma unroll
			for (int r = 0; r < R; r++) {
				if (T.live[r] && !clive[r]) {
					for (uint32_t w = 0; w < kp.mask_words; w++) ((XE_GLOBAL uint64_t *)kp.masks)[(size_t)ridx[r] * kp.mask_words + w] = 0;
				}
			}
		}
		XE_PHASE(5);
		C = N;
	}

	if (kp.lds_counters) {
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < kp.nmo * 3; i += XE_BLOCK) {
			unsigned long long v = 0;
			for (uint32_t c = 0; c < kp.lds_counters; c++) v += lcnt[(size_t)c * kp.nmo * 3 + i];
			if (v) atomicAdd(kp.counters + i, v);
		}
	}
}

/* ------------------------------------------------------------------ general kernel */
/* device rules (devices.c:237-314), then the object walk (netflow.c:213-244)
 * through the classification code shared with the fast kernel: object m
 * receives the record iff its program matches and its parent received it.
 * plist = all probes (kp.all_probes): presence is per lane here. */
/* list walks a lane of the per-lane kernels keeps in flight (measured on MI355X:
 * 4 instead of 2 leaves the sFlow kernel where it is, 0.4075 against 0.4071 ms on
 * 3.2 M samples, and costs the per-lane Netflow kernel 3 %, 0.651 against 0.635 ms) */
#ifndef XE_LANE_PC
#define XE_LANE_PC 2
#endif
template <typename REC, fval (*GET)(const REC &, uint32_t, uint32_t *)>
__device__ __forceinline__ void
classify(const xe_kparams &kp, REC &rec, int32_t device, uint32_t ridx, bool active,
	unsigned long long bytes_x_rate, unsigned long long pkts_x_rate, lds_u64 *lcnt, const lds_u64 *lmt)
{
	bool dropped = false;
	walk_cache wc;
	wc.root = 0;
	wc.hi = wc.lo = 0;
	wc.ans = 0;

	if (active && kp.marks) {
		uint32_t mk = 0;
		if (device >= 0) {
			const xe_devrule dr = kp.devrules[device];
			uint32_t cnt = 0;
			for (uint32_t i = 0; i < dr.nprogs; i++) {
				cnt += prog_match<REC, GET>(kp, rec, kp.progs[dr.first_prog + i], wc) ? 1u : 0u;
			}
			if (dr.skip_unmarked && (cnt == 0)) {
				dropped = true;
				mk = XE_MARK_DROPPED;
			} else {
				mk = XE_MARK_HAS | (cnt & 0x3FFFFFFFu);
			}
		}
		rec.mark = mk;
		kp.marks[ridx] = mk;
	}

	if (kp.per_object) {
		/* configurations beyond the limits of the shared classification (more
		 * distinct terms or lookups than its bit sets hold): every object's
		 * program is run on its own, parents always have a smaller index */
		for (uint32_t w = 0; w < kp.mask_words; w++) {
			uint64_t bits = 0;
			uint32_t m0 = w * 64, m1 = (m0 + 64 < kp.nmo) ? m0 + 64 : kp.nmo;
			for (uint32_t m = m0; m < m1; m++) {
				bool hit = false;
				if (active && !dropped) {
					const xe_prog pr = kp.progs[m];
					bool parent_ok = true;
					if (pr.parent >= 0) {
						/* a parent in an earlier word was stored already */
						parent_ok = ((uint32_t)pr.parent >= m0)
							? ((bits >> ((uint32_t)pr.parent - m0)) & 1) != 0
							: ((kp.masks[(size_t)ridx * kp.mask_words + ((uint32_t)pr.parent >> 6)]
								>> ((uint32_t)pr.parent & 63)) & 1) != 0;
					}
					hit = parent_ok && prog_match<REC, GET>(kp, rec, pr, wc);
				}
				if (hit) bits |= 1ull << (m - m0);
				/* counters: wave-wide sums, one atomic per wave and counter */
				unsigned long long ball = __ballot(hit);
				if (ball) {
					unsigned long long sb = wave_sum(hit ? bytes_x_rate : 0ull);
					unsigned long long sp = wave_sum(hit ? pkts_x_rate : 0ull);
					if ((threadIdx.x & 63) == 0) {
						atomicAdd(&kp.counters[3 * m + 0], (unsigned long long)__popcll(ball));
						atomicAdd(&kp.counters[3 * m + 1], sb);
						atomicAdd(&kp.counters[3 * m + 2], sp);
					}
				}
			}
			if (active) kp.masks[(size_t)ridx * kp.mask_words + w] = bits;
		}
		return;
	}

	lane_tile<REC, GET> T;
	uint64_t tb[1][4];
	const uint32_t ri[1] = { ridx };
	const unsigned long long bx[1] = { bytes_x_rate }, px[1] = { pkts_x_rate };
	T.rec = &rec;
	T.live[0] = active && !dropped;
	if (kp.lw == 1) {
		uint64_t lk[1][1];
		const uint64_t pl[1] = { kp.all_probes[0] };
		XE_PHASE_DECL;
		const uint32_t lane = threadIdx.x & 63;
		/* per-lane kernels have registers to spare: four range groups at a time */
		classify_atoms<1, false, 1, lane_tile<REC, GET>, 4, XE_LANE_PC>(kp, T, pl, lk, tb, lmt);
		XE_PHASE(2);
		classify_objects<1, false, 1>(kp, T.live, lk, tb, ri, bx, px, lcnt, lmt);
		XE_PHASE(3);
	} else {
		uint64_t lk[1][XE_LK_WORDS];
		uint64_t pl[XE_LK_WORDS];
		for (int k = 0; k < XE_LK_WORDS; k++) pl[k] = kp.all_probes[k];
		classify_atoms<1, false, XE_LK_WORDS>(kp, T, pl, lk, tb, lmt);
		classify_objects<1, false, XE_LK_WORDS>(kp, T.live, lk, tb, ri, bx, px, lcnt, lmt);
	}
	/* a dropped record still owns its mask words */
	if (active && dropped) {
		for (uint32_t w = 0; w < kp.mask_words; w++) kp.masks[(size_t)ridx * kp.mask_words + w] = 0;
	}
}

/* the block's LDS counters and atom-table copy of the per-lane kernels */
__device__ __forceinline__ void
lane_kernel_lds(const xe_kparams &kp, uint64_t *lds, lds_u64 *&lcnt, lds_u64 *&lmt)
{
	lcnt = (lds_u64 *)lds;
	lmt = lcnt + (size_t)kp.lds_counters * kp.nmo * 3;
	if (kp.lds_counters) {
		for (uint32_t i = threadIdx.x; i < kp.nmo * 3 * kp.lds_counters; i += XE_BLOCK) lcnt[i] = 0;
	}
	if (kp.motab_lds) {
		for (uint32_t i = threadIdx.x; i < kp.motab_words; i += XE_BLOCK) lmt[i] = kp.motab[i];
	}
	__syncthreads();
}

__device__ __forceinline__ void
lane_kernel_flush(const xe_kparams &kp, const lds_u64 *lcnt)
{
	if (!kp.lds_counters) return;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < kp.nmo * 3; i += XE_BLOCK) {
		unsigned long long v = 0;
		for (uint32_t c = 0; c < kp.lds_counters; c++) v += lcnt[(size_t)c * kp.nmo * 3 + i];
		if (v) atomicAdd(kp.counters + i, v);
	}
}

/* General kernel: any run the fast kernel does not take (reads that may pass
 * the datagram end, variable-length / repeated / string elements, exporters
 * with a devices.conf entry).  One thread per record, everything interpreted
 * per lane. */
#ifndef XE_GEN_WAVES
#define XE_GEN_WAVES 5     /* measured on MI355X (C2 through this kernel): 4 waves/SIMD 0.70 ms, 5 waves 0.63 ms */
#endif
__global__ void __launch_bounds__(XE_BLOCK)
#if XE_GEN_WAVES
__attribute__((amdgpu_waves_per_eu(XE_GEN_WAVES, XE_GEN_WAVES)))
#endif
xe_nf_general_kernel(const xe_kparams kp)
{
	extern __shared__ uint64_t lane_lds[];
	lds_u64 *lcnt;
	lds_u64 *lmt;
	lane_kernel_lds(kp, lane_lds, lcnt, lmt);
	uint32_t g = blockIdx.x * XE_BLOCK + threadIdx.x;
	bool active = g < kp.nslowrec;
	uint32_t ridx = 0;
	nf_rec rec;
	unsigned long long bxr = 0, pxr = 0;
	int32_t device = -1;

	rec.mark = 0;
	rec.nparsed = 0;
	/* the run whose work range holds the wave's first thread: one search per
	 * wave with scalar loads (its lanes are consecutive records, which lie in
	 * that run or a few runs on) */
	uint32_t wlo = 0;
	{
		const uint32_t g0 = __builtin_amdgcn_readfirstlane(g - (threadIdx.x & 63));
		const XE_CONST uint32_t *sw = (const XE_CONST uint32_t *)kp.swstart;
		uint32_t hi = kp.nssets ? kp.nssets - 1 : 0;
		if (g0 < kp.nslowrec) {
			while (wlo < hi) {
				const uint32_t mid = (wlo + hi + 1) >> 1;
				if (sw[mid] <= g0) wlo = mid; else hi = mid - 1;
			}
		}
	}
	if (active) {
		uint32_t lo = wlo;
		while ((lo + 1 < kp.nssets) && (kp.swstart[lo + 1] <= g)) lo++;
		const xe_set s = kp.ssets[lo];
		uint32_t i = g - kp.swstart[lo];
		ridx = s.first_rec + i;
		int64_t roff = (int64_t)i * s.stride;
		int32_t avail = s.avail - (int32_t)roff;

		rec.src.p = kp.pkt + s.data_off + roff;
		rec.src.left = s.dgram_left - (int32_t)roff;
		rec.src.checked = (s.flags & XE_SET_CHECKED) != 0;
		rec.t = kp.tmpls + s.tmpl;
		rec.e = kp.tents + rec.t->ent_base;
		rec.ex = kp.exps + s.exporter;
		device = rec.ex->device;

		/* how many template fields the reference parses for this record */
		uint32_t nf = rec.t->nfields, np = 0;
		int32_t pos = 0;
		if (!(s.flags & XE_SET_PARTIAL)) {
			/* a run of complete records: all template fields */
			np = nf;
		} else if (rec.t->mode == XE_MODE_V9) {
			for (; np < nf; np++) {
				if ((pos + (int32_t)rec.e[np].len) > avail) break;
				pos += rec.e[np].len;
			}
		} else {
			for (; np < nf;) {
				int32_t off;
				uint32_t len;
				if (rec.t->varlen) {
					rec.locate(np, &off, &len);
					pos = off + (int32_t)len;
				} else {
					pos += rec.e[np].len;
				}
				np++;
				if (pos >= avail) break;
			}
		}
		rec.nparsed = np;
		XE_LANE_STORE(&kp.nparsed[ridx], (uint8_t)np);

		/* decode into the columns */
		for (uint32_t j = 0; j < np; j++) {
			uint32_t f = rec.e[j].field;
			if (f == XE_F_NONE) continue;
			bool special = rec.t->varlen || (rec.e[j].flags & XE_TE_REPEAT)
				|| (rec.e[j].next_same != 0xFF) || c_fstr[f];
			if (!special) {
				bool ok = (rec.e[j].flags & XE_TE_VALID) != 0;
				if (ok) store_col(kp, f, ridx, pkt_be(rec.src, rec.e[j].off, rec.e[j].len));
				if (kp.vsize_off[f] != ~0ull) {
					((uint16_t *)(kp.out_base + kp.vsize_off[f]))[ridx] = ok ? rec.e[j].len : (uint16_t)0xFFFF;
				}
				continue;
			}
			int32_t off;
			uint32_t len;
			rec.locate(j, &off, &len);
			bool ok = rec.valid(j, len, c_fmin, c_fmax);
			if (c_fstr[f]) {
				/* left-aligned copy; the first occurrence also clears the
				 * rest of the slot (the record starts zeroed, netflow.c:275) */
				uint32_t w = c_colw[f];
				uint8_t *dst = kp.out_base + kp.col_off[f] + (size_t)ridx * w;
				bool first = !(rec.e[j].flags & XE_TE_REPEAT);
				if (first) {
					for (uint32_t b = 0; b < w; b++) dst[b] = (ok && (b < len)) ? (uint8_t)pkt_byte(rec.src, off + (int32_t)b) : 0;
				} else if (ok) {
					for (uint32_t b = 0; b < len; b++) dst[b] = (uint8_t)pkt_byte(rec.src, off + (int32_t)b);
				}
				if (kp.vsize_off[f] != ~0ull) {
					uint16_t *vs = (uint16_t *)(kp.out_base + kp.vsize_off[f]) + ridx;
					if (ok) *vs = (uint16_t)len;
					else if (first) *vs = 0xFFFF;
				}
			} else {
				uint32_t sz = 0;
				fval fv = nf_get(rec, f, &sz);
				if (fv.has) store_col(kp, f, ridx, fv.v);
				if (kp.vsize_off[f] != ~0ull) {
					((uint16_t *)(kp.out_base + kp.vsize_off[f]))[ridx] = fv.has ? (uint16_t)sz : 0xFFFF;
				}
			}
		}

		/* what the counters add: get_nf_val(8 bytes) * sampling_rate, wrapping */
		uint64_t rate = rec.ex->sampling_rate;
		if (kp.rates) kp.rates[ridx] = (uint32_t)rate;
		if (kp.devips) {
			const uint8_t *d = rec.ex->dev_ip;
			kp.devips[ridx] = (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24);
		}
		bxr = nf_get(rec, XF_in_bytes, nullptr).v.lo * rate;
		pxr = nf_get(rec, XF_in_pkts, nullptr).v.lo * rate;
	}
	classify<nf_rec, nf_get>(kp, rec, device, ridx, active, bxr, pxr, lcnt, lmt);
	lane_kernel_flush(kp, lcnt);
}

/* ------------------------------------------------------------------ sFlow v5 */
/*
 * One thread per flow-sample element that yields a record (the host pre-pass
 * located them).  The thread walks the sampled header the way rawparse.h does
 * (Ethernet, up to two VLAN tags, IPv4 / IPv6 with the IPIP and 6in6 cases,
 * TCP / UDP / ICMP) and fills the flow fields with the hooks of sflow.c:35-77,
 * quirks included: every copy is left-aligned in its field, the IPv6
 * destination is copied from the source address, VLAN fields keep the whole
 * TCI, and only parser states below NO_IP are skipped.
 *
 * The flow fields live in registers: field i = v[i], its bytes in memory
 * order packed little-endian (byte k of the field at bits 8k); the two
 * 16-byte fields keep bytes 8-15 in v6hi[].  Every field index below is a
 * compile-time constant, so nothing is indexed dynamically.
 */
#define SF_NFIELDS 23
/* bit order of the fields the sFlow decoder can set (same order as the host
 * uses when it re-creates struct flow_info) */
enum { SFI_in_bytes, SFI_in_pkts, SFI_protocol, SFI_src_tos, SFI_dst_tos, SFI_tcp_flags, SFI_l4_src_port,
	SFI_ip4_src, SFI_input_snmp, SFI_l4_dst_port, SFI_ip4_dst, SFI_output_snmp, SFI_ip6_src, SFI_ip6_dst,
	SFI_icmp_type, SFI_min_ttl, SFI_max_ttl, SFI_src_mac, SFI_dst_mac, SFI_src_vlan, SFI_dst_vlan,
	SFI_src_cvlan, SFI_dst_cvlan };

__host__ __device__ constexpr int
sf_field_of(int i)
{
	constexpr int t[SF_NFIELDS] = {
		XF_in_bytes, XF_in_pkts, XF_protocol, XF_src_tos, XF_dst_tos, XF_tcp_flags, XF_l4_src_port,
		XF_ip4_src_addr, XF_input_snmp, XF_l4_dst_port, XF_ip4_dst_addr, XF_output_snmp, XF_ip6_src_addr,
		XF_ip6_dst_addr, XF_icmp_type, XF_min_ttl, XF_max_ttl, XF_src_mac, XF_dst_mac, XF_src_vlan,
		XF_dst_vlan, XF_src_dot1q_cvlan, XF_dst_dot1q_cvlan };
	return t[i];
}

/* maximum size of each of those fields in struct flow_info */
__host__ __device__ constexpr int
sf_size_of(int i)
{
	constexpr int t[SF_NFIELDS] = { 8, 8, 1, 1, 1, 2, 2, 4, 4, 2, 4, 4, 16, 16, 2, 1, 1, 6, 6, 2, 2, 2, 2 };
	return t[i];
}

/* For classification the decoded fields move from registers into the thread's
 * LDS column (the staging area is free by then): the filter code asks for
 * fields by number, which a table and an LDS read answer in a few
 * instructions, where a choice among 23 registers costs a compare per field
 * at every place a field is read.  Field i occupies sf_words_of(i) words from
 * word sf_slot_of(i) on, bytes in memory order. */
__host__ __device__ constexpr int
sf_words_of(int i)
{
	return (sf_size_of(i) <= 4) ? 1 : (sf_size_of(i) <= 8) ? 2 : 4;
}

/* fields the decoder only ever sets together with another one, to the same
 * bytes (sflow.c:47-62 and the VLAN hooks), share that one's words */
__host__ __device__ constexpr int
sf_alias_of(int i)
{
	return (i == SFI_ip6_dst) ? SFI_ip6_src : (i == SFI_max_ttl) ? SFI_min_ttl
		: (i == SFI_dst_vlan) ? SFI_src_vlan : (i == SFI_dst_cvlan) ? SFI_src_cvlan : i;
}

__host__ __device__ constexpr int
sf_slot_of(int i)
{
	int o = 0;
	if (i < SF_NFIELDS) i = sf_alias_of(i);
	for (int j = 0; j < i; j++) o += (sf_alias_of(j) == j) ? sf_words_of(j) : 0;
	return o;
}

/* per flow field: (index among the sFlow fields << 8) | first word, 0xFFFF = the
 * sFlow decoder never sets it */
__constant__ uint16_t c_sf_slot[XE_NFIELDS];

/* where the fields sflow_reset() sets per datagram come from (sflow-impl.h:274-291):
 * the agent address (4 or 16 bytes) and the sub-agent id, bytes as on the wire */
struct sf_meta
{
	const uint8_t *agent, *dev_id;
	uint32_t agent_ver;       /* 4 or 6 */
};

struct sf_rec
{
	uint64_t v[SF_NFIELDS];   /* field bytes 0-7 */
	uint64_t v6hi[2];         /* bytes 8-15 of ip6_src / ip6_dst */
	uint32_t has;
	uint32_t mark;
	sf_meta q;
	/* the sampled header / IPv4 structure: its first SF_STAGE_BYTES bytes staged
	 * in LDS (dword d of this thread at stg[d * XE_BLOCK]), the rest, if a
	 * header ever reaches that far, read from the datagram at p */
	const uint32_t *stg;
	const uint8_t *p;
	int32_t left;             /* bytes from p to the end of the datagram */
	int32_t lim;              /* reads of whole words stay below min(left, staged bytes) */
};

/* 112 bytes: Ethernet, two tags, two IPv6 headers and a TCP header up to its
 * flags; at 28 KiB a block, five blocks (20 waves) share a CU's LDS */
#define SF_STAGE_DWORDS 28
#define SF_STAGE_BYTES (4 * SF_STAGE_DWORDS)

/* byte `off` of the element; zero beyond the end of the datagram */
__device__ __forceinline__ uint32_t
sf_byte(const sf_rec &r, int32_t off)
{
	if (off >= r.left) return 0;
	if (off < SF_STAGE_BYTES) return (r.stg[(off >> 2) * XE_BLOCK] >> ((off & 3) * 8)) & 0xFF;
	return r.p[off];
}

/* the rare way: n bytes at `off` one by one, zero beyond the datagram's end
 * (kept out of line: dozens of call sites) */
__device__ __attribute__((noinline)) uint64_t
sf_bytes_slow(const uint32_t *stg, const uint8_t *p, int32_t left, int32_t off, int n)
{
	uint64_t x = 0;
	for (int i = 0; i < n; i++) {
		const int32_t o = off + i;
		uint32_t b = 0;
		if (o < left) b = (o < SF_STAGE_BYTES) ? ((stg[(o >> 2) * XE_BLOCK] >> ((o & 3) * 8)) & 0xFF) : p[o];
		x |= (uint64_t)b << (8 * i);
	}
	return x;
}

/* four staged bytes at `off` (off + 4 <= r.lim), packed little-endian: two
 * aligned LDS words and a byte shift */
__device__ __forceinline__ uint32_t
sf_word(const sf_rec &r, int32_t off)
{
	const uint32_t *w = r.stg + (off >> 2) * XE_BLOCK;
	return __builtin_amdgcn_alignbyte(w[XE_BLOCK], w[0], (uint32_t)off & 3);
}

/* N bytes of the sampled header at `off`, packed little-endian */
template <int N>
__device__ __forceinline__ uint64_t
sf_bytes(const sf_rec &r, int32_t off)
{
	uint64_t x = 0;
	/* the usual case: all of it staged and inside the datagram (lim leaves room
	 * for the second word of the last read) */
	if ((off >= 0) && (off + ((N + 3) & ~3) <= r.lim)) {
		if (N <= 4) {
			x = sf_word(r, off);
			if (N < 4) x &= (1u << (8 * (N & 3))) - 1u;
		} else {
			x = sf_word(r, off);
#pragma unroll
			for (int i = 4; i < N; i += 4) {
				uint64_t w = sf_word(r, off + i);
				if (N - i < 4) w &= (1u << (8 * ((N - i) & 3))) - 1u;
				x |= w << (8 * i);
			}
		}
		return x;
	}
	return sf_bytes_slow(r.stg, r.p, r.left, off, N);
}

/* COPY_TO_FLOW of sflow.c:26-33: N bytes to the START of the field */
template <int FI, int N>
__device__ __forceinline__ void
sf_copy(sf_rec &r, int32_t off)
{
	if (N <= 8) {
		const uint64_t m = (N == 8) ? ~0ull : ((1ull << (8 * (N & 7))) - 1ull);
		r.v[FI] = (r.v[FI] & ~m) | sf_bytes<N>(r, off);
	} else {
		r.v[FI] = sf_bytes<8>(r, off);
		r.v6hi[FI - SFI_ip6_src] = sf_bytes<8>(r, off + 8);
	}
	r.has |= 1u << FI;
}

template <int FI>
__device__ __forceinline__ void
sf_set_byte0(sf_rec &r, uint32_t b)
{
	r.v[FI] = (r.v[FI] & ~0xFFull) | (b & 0xFF);
	r.has |= 1u << FI;
}

__device__ __forceinline__ void
sf_on_ip(sf_rec &r, int32_t iph)   /* sflow.c:47-54 */
{
	sf_copy<SFI_ip4_src, 4>(r, iph + 12);
	sf_copy<SFI_ip4_dst, 4>(r, iph + 16);
	sf_copy<SFI_src_tos, 1>(r, iph + 1);
	sf_copy<SFI_dst_tos, 1>(r, iph + 1);
	sf_copy<SFI_min_ttl, 1>(r, iph + 8);
	sf_copy<SFI_max_ttl, 1>(r, iph + 8);
	sf_copy<SFI_protocol, 1>(r, iph + 9);
}

__device__ __forceinline__ void
sf_on_ip6(sf_rec &r, int32_t ip6h, uint32_t nexthdr)   /* sflow.c:57-62 */
{
	sf_copy<SFI_ip6_src, 16>(r, ip6h + 8);
	sf_copy<SFI_ip6_dst, 16>(r, ip6h + 8);
	sf_copy<SFI_min_ttl, 1>(r, ip6h + 7);
	sf_copy<SFI_max_ttl, 1>(r, ip6h + 7);
	sf_set_byte0<SFI_protocol>(r, nexthdr);
}

__device__ __forceinline__ uint32_t
sf_be16(const sf_rec &r, int32_t off)
{
	const uint32_t x = (uint32_t)sf_bytes<2>(r, off);
	return ((x & 0xFF) << 8) | (x >> 8);
}

/* rawpacket_parse(), rawparse.h:89-281; returns the parser state 0..4 */
__device__ __forceinline__ int
sf_rawparse(sf_rec &r, int32_t end, int type)
{
	int32_t ptr = 0;
	uint32_t h_proto;

	if ((ptr + 14) >= end) return 0;
	if (type == 0) {
		sf_copy<SFI_dst_mac, 6>(r, 0);
		sf_copy<SFI_src_mac, 6>(r, 6);
		h_proto = sf_be16(r, 12);
		ptr = 14;
		if ((h_proto == 0x8100) || (h_proto == 0x88a8)) {
			if ((ptr + 4) >= end) return 1;
			sf_copy<SFI_src_vlan, 2>(r, ptr);
			sf_copy<SFI_dst_vlan, 2>(r, ptr);
			h_proto = sf_be16(r, ptr + 2);
			ptr += 4;
		}
		if ((h_proto == 0x8100) || (h_proto == 0x88a8)) {
			if ((ptr + 4) >= end) return 1;
			sf_copy<SFI_src_cvlan, 2>(r, ptr);
			sf_copy<SFI_dst_cvlan, 2>(r, ptr);
			h_proto = sf_be16(r, ptr + 2);
			ptr += 4;
		}
	} else {
		h_proto = (type == 1) ? 0x0800 : 0x86dd;
	}

	uint32_t l4;
	if (h_proto == 0x0800) {
		if ((ptr + 20) >= end) return 2;
		int32_t iph = ptr;
		if ((sf_be16(r, iph + 6) & 0x3fff) != 0) {
			sf_on_ip(r, iph);
			return 3;
		}
		ptr += (int32_t)(sf_byte(r, iph) & 0x0f) * 4;
		if (sf_byte(r, iph + 9) == 4) {
			if ((ptr + 20) >= end) return 2;
			iph = ptr;
			ptr += (int32_t)(sf_byte(r, iph) & 0x0f) * 4;
		}
		sf_on_ip(r, iph);
		l4 = sf_byte(r, iph + 9);
		if ((l4 != 6) && (l4 != 17) && (l4 != 1)) return 3;
	} else if (h_proto == 0x86dd) {
		if ((ptr + 40) >= end) return 2;
		int32_t ip6h = ptr;
		ptr += 40;
		l4 = sf_byte(r, ip6h + 6);
		if (l4 == 4) {
			if ((ptr + 20) >= end) return 2;
			int32_t iph = ptr;
			l4 = sf_byte(r, iph + 9);
			ptr += (int32_t)(sf_byte(r, iph) & 0x0f) * 4;
			sf_on_ip(r, iph);
		} else if (l4 == 41) {
			if ((ptr + 40) >= end) return 2;
			ip6h = ptr;
			ptr += 40;
			l4 = sf_byte(r, ip6h + 6);
			sf_on_ip6(r, ip6h, l4);
		} else {
			sf_on_ip6(r, ip6h, l4);
		}
		if (l4 == 58) l4 = 1;
		if ((l4 != 6) && (l4 != 17) && (l4 != 1)) return 3;
	} else {
		return 2;
	}

	if (l4 == 6) {
		if ((ptr + 20) >= end) return 3;
		sf_copy<SFI_l4_src_port, 2>(r, ptr);
		sf_copy<SFI_l4_dst_port, 2>(r, ptr + 2);
		sf_copy<SFI_tcp_flags, 1>(r, ptr + 13);
	} else if (l4 == 17) {
		if ((ptr + 8) >= end) return 3;
		sf_copy<SFI_l4_src_port, 2>(r, ptr);
		sf_copy<SFI_l4_dst_port, 2>(r, ptr + 2);
	} else {
		if ((ptr + 8) >= end) return 3;
		sf_copy<SFI_icmp_type, 1>(r, ptr);
	}
	return 4;
}

/* what the decoder needs of an element's descriptor (struct xe_sfrec) */
struct sf_elem
{
	uint64_t hdr_off;
	uint32_t header_len, frame_len_be, kind, rp_type;
	int32_t left;
};

/* the descriptor arrives as four 16-byte loads, all in flight together */
struct sf_desc { u32x4 a, b, c, d; };

__device__ __forceinline__ sf_desc
sf_desc_load(const xe_sfrec *q)
{
	const u32x4 *qq = (const u32x4 *)q;
	sf_desc x;
	x.a = qq[0]; x.b = qq[1]; x.c = qq[2]; x.d = qq[3];
	return x;
}

__device__ __forceinline__ sf_elem
sf_elem_of(const sf_desc &x)
{
	sf_elem e;
	e.hdr_off = (uint64_t)x.a.x | ((uint64_t)x.a.y << 32);
	e.header_len = x.a.z;
	e.left = (int32_t)x.a.w;
	e.frame_len_be = x.b.x;
	e.kind = x.c.x & 0xFF;
	e.rp_type = (x.c.x >> 8) & 0xFF;
	return e;
}

/* apply one element to the flow image; returns the parser state (4 for the
 * IPv4 structure element, which is always delivered) */
/* wide loads of an element's first SF_STAGE_BYTES bytes, all issued before the
 * first one is used; a piece that starts inside the datagram may run past its
 * end (the batch buffer is padded), those bytes are never used; a piece that
 * starts beyond it is not read (the load goes to the element's first bytes
 * instead and its result is dropped) */
struct sf_staged { u32x4 v[SF_STAGE_DWORDS / 4]; };

__device__ __forceinline__ void
sf_stage_load(sf_staged &S, const uint8_t *p, int32_t left)
{
#pragma unroll
	for (int i = 0; i < SF_STAGE_DWORDS / 4; i++) {
		const bool in = 16 * i < left;
		S.v[i] = *(const XE_GLOBAL u32x4 *)(p + (in ? 16 * i : 0));
	}
}

/* ... into this thread's LDS column */
__device__ __forceinline__ void
sf_stage_put(sf_rec &r, const sf_staged &S)
{
	uint32_t *d = const_cast<uint32_t *>(r.stg);
#pragma unroll
	for (int i = 0; i < SF_STAGE_DWORDS / 4; i++) {
		const bool in = 16 * i < r.left;
		d[(4 * i + 0) * XE_BLOCK] = in ? S.v[i].x : 0u;
		d[(4 * i + 1) * XE_BLOCK] = in ? S.v[i].y : 0u;
		d[(4 * i + 2) * XE_BLOCK] = in ? S.v[i].z : 0u;
		d[(4 * i + 3) * XE_BLOCK] = in ? S.v[i].w : 0u;
	}
}

/* PRE: the element's bytes were asked for ahead (pre) */
template <bool PRE>
__device__ __forceinline__ int
sf_apply(const xe_kparams &kp, sf_rec &r, const sf_elem &q, const sf_staged *pre)
{
	r.p = kp.pkt + q.hdr_off;
	r.left = q.left;
	/* a word read looks one dword ahead of the one it starts in */
	r.lim = ((r.left < SF_STAGE_BYTES) ? r.left : SF_STAGE_BYTES) - 4;
	if (PRE) {
		sf_stage_put(r, *pre);
	} else {
		sf_staged S;
		sf_stage_load(S, r.p, r.left);
		sf_stage_put(r, S);
	}
	/* sflow-impl.h:164-171 / :224-234: in_bytes[4..8] = frame length as on
	 * the wire, in_pkts[7] = 1 */
	r.v[SFI_in_bytes] = (r.v[SFI_in_bytes] & 0xFFFFFFFFull) | ((uint64_t)q.frame_len_be << 32);
	r.v[SFI_in_pkts] = (r.v[SFI_in_pkts] & 0x00FFFFFFFFFFFFFFull) | (1ull << 56);
	r.has |= (1u << SFI_in_bytes) | (1u << SFI_in_pkts);
	/* the declared length is unsigned (sflow.c:132-134: `end = p + header_len`):
	 * from 2^31 on no bounds test of the parse can fail, as with the largest int */
	if (q.kind == 1) return sf_rawparse(r, (q.header_len > 0x7FFFFFFFu) ? 0x7FFFFFFF : (int32_t)q.header_len, (int)q.rp_type);

	/* IPv4 structure, sflow-impl.h:236-260: seven 4-byte words */
	sf_set_byte0<SFI_protocol>(r, sf_byte(r, 3));
	sf_copy<SFI_ip4_src, 4>(r, 4);
	sf_copy<SFI_ip4_dst, 4>(r, 8);
	sf_copy<SFI_l4_src_port, 2>(r, 14);
	sf_copy<SFI_l4_dst_port, 2>(r, 18);
	sf_copy<SFI_tcp_flags, 1>(r, 23);
	sf_copy<SFI_src_tos, 1>(r, 27);
	if ((r.v[SFI_protocol] & 0xFF) == 1) sf_set_byte0<SFI_icmp_type>(r, sf_byte(r, 17));
	return 4;
}

/* field FI as the filter terms see it: its maximum-size bytes read as one
 * big-endian number */
template <int FI>
__device__ __forceinline__ void
sf_number(const sf_rec &r, fval &out)
{
	constexpr int n = sf_size_of(FI);
	out.has = ((r.has >> FI) & 1) != 0;
	if (n <= 8) {
		out.v.hi = 0;
		out.v.lo = __builtin_bswap64(r.v[FI]) >> (64 - 8 * n);
	} else {
		out.v.hi = __builtin_bswap64(r.v[FI]);
		out.v.lo = __builtin_bswap64(r.v6hi[FI - SFI_ip6_src]);
	}
}

template <int FI>
__device__ __forceinline__ bool
sf_number_if(const sf_rec &r, uint32_t f, fval &out)
{
	if (f != (uint32_t)sf_field_of(FI)) return false;
	sf_number<FI>(r, out);
	return true;
}

template <int... FI>
__device__ __forceinline__ void
sf_number_any(const sf_rec &r, uint32_t f, fval &out, std::integer_sequence<int, FI...>)
{
	(void)(sf_number_if<FI>(r, f, out) || ...);
}

__device__ __forceinline__ fval
sf_get(const sf_rec &r, uint32_t f, uint32_t *size_out)
{
	fval out;
	out.has = false;
	out.v.hi = out.v.lo = 0;
	(void)size_out;
	if (f == XF_dev_id) {
		const uint8_t *b = r.q.dev_id;
		out.has = true;
		out.v.lo = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
		return out;
	}
	if ((f == XF_dev_ip) || (f == XF_dev_ip6)) {
		bool v4 = r.q.agent_ver == 4;
		if ((f == XF_dev_ip) != v4) return out;
		out.has = true;
		for (int i = 0; i < (v4 ? 4 : 16); i++) {
			out.v.hi = (out.v.hi << 8) | (out.v.lo >> 56);
			out.v.lo = (out.v.lo << 8) | r.q.agent[i];
		}
		return out;
	}
	if (f == XF_dev_mark) {
		out.has = (r.mark & XE_MARK_HAS) != 0;
		out.v.lo = r.mark & 0x3FFFFFFFu;
		return out;
	}
	sf_number_any(r, f, out, std::make_integer_sequence<int, SF_NFIELDS>());
	return out;
}

/* the record as the classification code reads it: the field image in LDS */
struct sf_img
{
	const uint32_t *img;      /* word w of this thread at img[w * XE_BLOCK] */
	uint32_t has;
	uint32_t mark;
	sf_meta q;
};

template <int FI>
__device__ __forceinline__ void
sf_dump(const sf_rec &r, uint32_t *img)
{
	constexpr int slot = sf_slot_of(FI), nw = sf_words_of(FI);
	if (sf_alias_of(FI) != FI) return;
	img[slot * XE_BLOCK] = (uint32_t)r.v[FI];
	if (nw >= 2) img[(slot + 1) * XE_BLOCK] = (uint32_t)(r.v[FI] >> 32);
	if (nw == 4) {
		img[(slot + 2) * XE_BLOCK] = (uint32_t)r.v6hi[(FI >= SFI_ip6_src) ? FI - SFI_ip6_src : 0];
		img[(slot + 3) * XE_BLOCK] = (uint32_t)(r.v6hi[(FI >= SFI_ip6_src) ? FI - SFI_ip6_src : 0] >> 32);
	}
}

template <int... FI>
__device__ __forceinline__ void
sf_dump_all(const sf_rec &r, uint32_t *img, std::integer_sequence<int, FI...>)
{
	(sf_dump<FI>(r, img), ...);
}

__device__ __forceinline__ fval
sf_img_get(const sf_img &r, uint32_t f, uint32_t *size_out)
{
	fval out;
	out.has = false;
	out.v.hi = out.v.lo = 0;
	(void)size_out;
	if (f == XF_dev_id) {
		const uint8_t *b = r.q.dev_id;
		out.has = true;
		out.v.lo = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
		return out;
	}
	if ((f == XF_dev_ip) || (f == XF_dev_ip6)) {
		bool v4 = r.q.agent_ver == 4;
		if ((f == XF_dev_ip) != v4) return out;
		out.has = true;
		for (int i = 0; i < (v4 ? 4 : 16); i++) {
			out.v.hi = (out.v.hi << 8) | (out.v.lo >> 56);
			out.v.lo = (out.v.lo << 8) | r.q.agent[i];
		}
		return out;
	}
	if (f == XF_dev_mark) {
		out.has = (r.mark & XE_MARK_HAS) != 0;
		out.v.lo = r.mark & 0x3FFFFFFFu;
		return out;
	}
	const uint32_t s = c_sf_slot[f];
	if (s == 0xFFFF) return out;
	const uint32_t n = c_fmax[f];
	const uint32_t *w = r.img + (s & 0xFF) * XE_BLOCK;
	out.has = ((r.has >> (s >> 8)) & 1) != 0;
	/* the field's bytes read as one big-endian number */
	if (n <= 4) {
		out.v.lo = __builtin_bswap32(w[0]) >> (32 - 8 * n);
	} else if (n <= 8) {
		out.v.lo = __builtin_bswap64((uint64_t)w[0] | ((uint64_t)w[XE_BLOCK] << 32)) >> (64 - 8 * n);
	} else {
		out.v.hi = __builtin_bswap64((uint64_t)w[0] | ((uint64_t)w[XE_BLOCK] << 32));
		out.v.lo = __builtin_bswap64((uint64_t)w[2 * XE_BLOCK] | ((uint64_t)w[3 * XE_BLOCK] << 32));
	}
	return out;
}

/* field FI from the thread's field image to its column, if the record has it */
template <int FI>
__device__ __forceinline__ void
sf_store(const xe_kparams &kp, const uint32_t *img, uint32_t has, uint32_t ridx)
{
	constexpr int f = sf_field_of(FI), n = sf_size_of(FI), slot = sf_slot_of(FI);
	constexpr int w = (n <= 1) ? 1 : (n <= 2) ? 2 : (n <= 4) ? 4 : (n <= 8) ? 8 : 16;
	if (!((has >> FI) & 1)) return;
	uint8_t *dst = kp.out_base + kp.col_off[f] + (size_t)ridx * w;
	const uint32_t w0 = img[slot * XE_BLOCK];
	if (w == 1) XE_LANE_STORE(dst, (uint8_t)w0);
	else if (w == 2) XE_LANE_STORE((uint16_t *)dst, (uint16_t)w0);
	else if (w == 4) XE_LANE_STORE((uint32_t *)dst, w0);
	else if (w == 8) XE_LANE_STORE((uint64_t *)dst, (uint64_t)w0 | ((uint64_t)img[(slot + 1) * XE_BLOCK] << 32));
	else {
		XE_LANE_STORE((uint64_t *)dst, (uint64_t)w0 | ((uint64_t)img[(slot + 1) * XE_BLOCK] << 32));
		XE_LANE_STORE((uint64_t *)dst + 1, (uint64_t)img[(slot + 2) * XE_BLOCK] | ((uint64_t)img[(slot + 3) * XE_BLOCK] << 32));
	}
}

template <int... FI>
__device__ __forceinline__ void
sf_store_all(const xe_kparams &kp, const uint32_t *img, uint32_t has, uint32_t ridx, std::integer_sequence<int, FI...>)
{
	(sf_store<FI>(kp, img, has, ridx), ...);
}

/* the slow way to an element: the word-by-word walk, kept out of line (datagrams
 * of unusual shape only) */
__device__ __attribute__((noinline)) bool
sf_next_slow(xe_sfiter &c, xe_sfel &d)
{
	return xsf_next<false>(c, d);
}

__device__ __forceinline__ sf_elem
sf_elem_of(const xe_sfel &e, uint64_t dgram_off, uint32_t dgram_len)
{
	sf_elem q;
	q.hdr_off = dgram_off + e.hdr_pos;
	q.header_len = e.header_len;
	q.frame_len_be = e.frame_len_be;
	q.kind = e.kind;
	q.rp_type = e.rp_type;
	q.left = (int32_t)(dgram_len - e.hdr_pos);
	return q;
}

#ifndef XE_SF_WAVES
#define XE_SF_WAVES 4      /* measured on MI355X (3.2 M samples): 4 waves 0.430 ms; 5 waves spill and take 0.526 ms */
#endif
/* SELFWALK: the datagrams were walked on the device (kp.sfdg, kp.sfbase,
 * kp.sf_rec2dg): a lane finds its sample through the datagram's entry and reads
 * the sample and element headers itself; else every record has a descriptor
 * the host walk wrote (kp.sfrecs) */
template <bool SELFWALK>
__global__ void __launch_bounds__(XE_BLOCK)
#if XE_SF_WAVES
__attribute__((amdgpu_waves_per_eu(XE_SF_WAVES, XE_SF_WAVES)))
#endif
xe_sf_kernel(const xe_kparams kp)
{
	extern __shared__ uint64_t lane_lds[];
	lds_u64 *lcnt;
	lds_u64 *lmt;
	lane_kernel_lds(kp, lane_lds, lcnt, lmt);
	XE_PHASE_DECL;
	const uint32_t lane = threadIdx.x & 63;
	uint32_t g = blockIdx.x * XE_BLOCK + threadIdx.x;
	bool active = g < kp.nsf;
	uint32_t ridx = kp.nrec + g;
	sf_rec rec;
	unsigned long long bxr = 0, pxr = 0;
	int32_t device = -1;
	bool deliver = false;
	uint32_t rate_out = 0, a4_out = 0;
	uint8_t state_out = 0;

#pragma unroll
	for (int i = 0; i < SF_NFIELDS; i++) rec.v[i] = 0;
	rec.v6hi[0] = rec.v6hi[1] = 0;
	rec.has = 0;
	rec.mark = 0;
	rec.q.agent = rec.q.dev_id = kp.pkt;
	rec.q.agent_ver = 4;
	/* the staging columns follow the counters and the atom-table copy */
	rec.stg = (const uint32_t *)(lane_lds + (size_t)kp.lds_counters * kp.nmo * 3
		+ (kp.motab_lds ? (size_t)kp.motab_words : 0)) + threadIdx.x;
	rec.p = kp.pkt;
	rec.left = 0;
	rec.lim = 0;
	if (active) {
		uint32_t rate = 0, in_if_be = 0, out_if_be = 0;
		int state = 4;
		if (SELFWALK) {
			const u32x4 rd = *(const u32x4 *)(kp.sfrd + g);
			const uint32_t dgi = rd.w;
			uint64_t dgoff = 0;
			uint32_t dglen = 0;
			xe_sfel el;
			xe_sfiter it;
			bool replay = false, found = true;

			rec.q.agent_ver = (rd.y & XE_SFRD_AGENT6) ? 6 : 4;
			device = (int32_t)(rd.y >> 16) - 1;
			if ((rd.y & XE_SFRD_WALK) || kp.fast_want_dev || kp.devips || kp.ndevrules) {
				/* the datagram's entry: for the walk below, and when something reads
				 * what sflow_reset() sets per datagram (sflow-impl.h:274-291) */
				const u32x4 ga = *(const u32x4 *)(kp.sfdg + dgi);
				dgoff = (uint64_t)ga.x | ((uint64_t)ga.y << 32);
				dglen = ga.z & XE_SFDG_LEN;
				rec.q.agent = kp.pkt + dgoff + XE_SF_AGENT_POS;
				rec.q.dev_id = rec.q.agent + ((rd.y & XE_SFRD_AGENT6) ? 16 : 4);
			}
			if (!(rd.y & XE_SFRD_WALK)) {
				/* the usual shape: the record is the first element of its sample, a
				 * sampled header; the 64 bytes before that header hold the element header
				 * (6 words) and, in front of it, the sample header: 10 words of a compact
				 * sample (type word included), the last 10 of 13 of an expanded one */
				const uint64_t hoff = (uint64_t)rd.x | ((uint64_t)(rd.y & 0xFFu) << 32);
				const bool expanded = (rd.y & XE_SFRD_EXPANDED) != 0;
				const uint8_t *hp = kp.pkt + hoff;
				u32x4 u[4];
				sf_staged pre;
#pragma unroll
				for (int i = 0; i < 4; i++) u[i] = *(const XE_GLOBAL u32x4 *)(hp - 64 + 16 * i);
				/* the sampled header's bytes are asked for in the same breath: where it
				 * lies does not depend on what the window holds */
				sf_stage_load(pre, hp, (int32_t)rd.z);
				/* the element is described relative to the sampled header itself */
				dgoff = hoff;
				dglen = (uint32_t)rd.z;
				el.hdr_pos = 0;
				el.sample_pos = 0;
				el.kind = 1;
				el.chained = 0;
				el.first = 1;
				el.rate = __builtin_bswap32(expanded ? u[0].z : u[1].x);
				el.in_if_be = expanded ? u[1].z : (u[1].w & __builtin_bswap32(0x3fffffffu));
				el.out_if_be = expanded ? u[2].x : (u[2].x & __builtin_bswap32(0x3fffffffu));
				const uint32_t hproto = __builtin_bswap32(u[3].x);
				el.rp_type = (hproto == 11) ? 1 : (hproto == 12) ? 2 : 0;
				el.frame_len_be = u[3].y;
				el.header_len = __builtin_bswap32(u[3].w);
				rate = el.rate;
				in_if_be = el.in_if_be;
				out_if_be = el.out_if_be;
				state = sf_apply<true>(kp, rec, sf_elem_of(el, dgoff, dglen), &pre);
				found = false;      /* (done: nothing left for the walk below) */
			} else {
				/* any other shape: walk the datagram to the record; when earlier
				 * elements of its sample yielded records too, the flow carries over
				 * (sflow-impl.h:145-270 never resets it between elements): replay the
				 * sample from its start */
				const uint32_t k = g - kp.sfbase[dgi];
				uint32_t av, idx = 0;
				found = xsf_begin(it, kp.pkt + dgoff, (int64_t)dglen, av);
				while (found && (found = sf_next_slow(it, el)) && (idx < k)) idx++;
				if (found && el.chained) {
					replay = true;
					it.pos = el.sample_pos;
					it.ns_left = 1;
					it.nel_left = 0;
					it.bad = false;
					it.chained = false;
				}
			}
			if (found) {
				rate = el.rate;
				in_if_be = el.in_if_be;
				out_if_be = el.out_if_be;
				for (;;) {
					xe_sfel e = el;
					bool last = true;
					if (replay) {
						if (!sf_next_slow(it, e)) break;
						last = e.hdr_pos >= el.hdr_pos;
					}
					state = sf_apply<false>(kp, rec, sf_elem_of(e, dgoff, dglen), nullptr);
					if (last) break;
				}
			} else if (rd.y & XE_SFRD_WALK) {
				state = 0;   /* (the count pass and this walk disagree: cannot happen; the record stays empty) */
			}
		} else {
			const xe_sfrec *q = kp.sfrecs + g;
			const sf_desc own = sf_desc_load(q);
			rec.q.agent = q->agent;
			rec.q.dev_id = q->dev_id;
			rec.q.agent_ver = (own.c.x >> 16) & 0xFF;
			/* elements of one sample share the flow: replay the earlier ones */
			uint32_t first = g;
			if ((own.c.x >> 24) & 0xFF) {
				first = g - 1;
				while ((first > 0) && kp.sfrecs[first].chained) first--;
			}
			rate = own.b.y;
			in_if_be = own.b.z;
			out_if_be = own.b.w;
			device = kp.exps[own.d.z].device;
			for (uint32_t k = first; k <= g; k++) {
				sf_desc x = own;
				if (k != g) x = sf_desc_load(kp.sfrecs + k);
				state = sf_apply<false>(kp, rec, sf_elem_of(x), nullptr);
			}
		}
		/* sf5_flow(), sflow-impl.h:106-143: interface indices, bytes as stored */
		rec.v[SFI_input_snmp] = in_if_be;
		rec.v[SFI_output_snmp] = out_if_be;
		rec.has |= (1u << SFI_input_snmp) | (1u << SFI_output_snmp);
		deliver = state >= 2;   /* sflow.c:164-168 */
		XE_PHASE(0);
		rate_out = rate;
		state_out = (uint8_t)(state | (deliver ? 0 : XE_SF_SKIPPED));
		if (kp.devips && (rec.q.agent_ver == 4)) {
			/* dev_ip = the agent's address when it is an IPv4 one (sflow-impl.h:330-345):
			 * its four bytes in memory order */
			__builtin_memcpy(&a4_out, rec.q.agent, 4);
		}
		bxr = sf_get(rec, XF_in_bytes, nullptr).v.lo * (uint64_t)rate;
		pxr = sf_get(rec, XF_in_pkts, nullptr).v.lo * (uint64_t)rate;
	}
	/* the fields move to the thread's LDS column for the filter code */
	static_assert(sf_slot_of(SF_NFIELDS) <= SF_STAGE_DWORDS, "the field image must fit the staging column");
	sf_img im;
	im.img = rec.stg;
	im.has = rec.has;
	im.mark = 0;
	im.q = rec.q;
	sf_dump_all(rec, const_cast<uint32_t *>(rec.stg), std::make_integer_sequence<int, SF_NFIELDS>());
	XE_PHASE(1);
	classify<sf_img, sf_img_get>(kp, im, device, ridx, active && deliver, bxr, pxr, lcnt, lmt);
	/* every store last: vector memory operations retire in order, and a list
	 * walk issued behind a store would wait for it */
	if (active) {
		XE_LANE_STORE(&kp.nparsed[ridx], state_out);
		XE_LANE_STORE(&kp.sf_has[g], rec.has);
		if (kp.sf_rate) XE_LANE_STORE(&kp.sf_rate[g], rate_out);
		if (kp.rates) kp.rates[ridx] = rate_out;
		if (kp.devips) kp.devips[ridx] = a4_out;
		if (!deliver && kp.marks) kp.marks[ridx] = 0;
		sf_store_all(kp, rec.stg, rec.has, ridx, std::make_integer_sequence<int, SF_NFIELDS>());
	}
	if (active && !deliver) {
		for (uint32_t w = 0; w < kp.mask_words; w++) kp.masks[(size_t)ridx * kp.mask_words + w] = 0;
	}
	lane_kernel_flush(kp, lcnt);
}

/* ------------------------------------------------------------------ fixed-window tables */
/*
 * After classification: for every table whose object received a record, the
 * key is put together from the decoded columns (the bytes struct flow_info
 * holds, monit-objects.c:917-920), looked up or inserted, and value * scale *
 * sampling rate is added to its sums (monit-objects.c:986-1008; wrapping u64,
 * so the order of records and the split over batches and GPUs do not matter).
 *
 * Traffic is full of repeated keys (a few protocols, a few ports, a few busy
 * hosts), and atomic adds to one row of the table in HBM run one after the
 * other.  So sums are collected on the way: equal keys of a wave first (the
 * lanes of one hash value elect a speaker), then in a small table in LDS that
 * the block keeps over all its records and writes out once at the end; only
 * keys that find their LDS entry taken by another key go to HBM one by one.
 */
#define XE_FWM_CACHE 256         /* entries of the block's table in LDS */
#ifndef XE_FWM_CHUNKS
#define XE_FWM_CHUNKS 16         /* 256-record chunks a block works through */
#endif

/* the row of one key in the table in HBM, claimed when the key is new;
 * XE_FWM_NOROW when the table is full.
 *
 * Key words are claimed with compare-and-swap and read with agent-scope
 * atomic loads.  Such loads are answered at the point where all CUs of the
 * device agree, never by a cache on the way that may hold the line as it was
 * before the word was written (a plain load may be: the L1 of a CU is not kept
 * coherent).  No word ever changes once it is set, so a probe needs no order
 * between its loads and waits for nobody.
 *
 * Keys in one array, the sums in another: with the sums inside the slot (one
 * line per record instead of two) the adds to a line hold up the probes of
 * the same line (measured on MI355X, 2.1 M records into a filled table of
 * 177 000 keys: 0.542 ms against 0.279 ms). */
#define XE_FWM_NOROW 0xFFFFFFFFu
template <int MKW>
__device__ __forceinline__ uint32_t
fwm_global_find(const xe_fwm_dev &F, bool want, const uint64_t (&kw)[MKW], uint64_t h, uint32_t nrecords)
{
	uint32_t slot = (uint32_t)h & F.cap_mask, tried = 0;
	bool done = !want;
	if (F.kwords == 1) {
		if (!want) return XE_FWM_NOROW;
		/* A key of one word is its own publication: the word holds key ^
		 * XE_FWM_KEY_XOR, 0 while the slot is free, and is claimed with one
		 * compare-and-swap that hands back whatever key got there first - no
		 * state to wait for, nothing to order.  (The state word is still set, for
		 * the host that reads the table after the kernel.)  The one key that
		 * would be stored as 0 has a row of its own behind the table. */
		const unsigned long long x = kw[0] ^ XE_FWM_KEY_XOR;
		if (x == 0) {
			slot = F.cap_mask + 1;
			__hip_atomic_store((uint32_t *)(F.slots + (size_t)slot * F.swords), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			for (;;) {
				unsigned long long *S = F.slots + (size_t)slot * F.swords;
				unsigned long long cur = __hip_atomic_load(&S[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (cur == 0) {
					cur = atomicCAS(&S[1], 0ull, x);
					if (cur == 0) {
						__hip_atomic_store((uint32_t *)S, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						cur = x;
					}
				}
				if (cur == x) break;
				slot = (slot + 1) & F.cap_mask;
				if (++tried > F.cap_mask) {
					/* full: the records behind this key are lost, and counted */
					atomicAdd(F.dropped, (unsigned long long)nrecords);
					return XE_FWM_NOROW;
				}
			}
		}
		return slot;
	}
	/* A key of several words lies in its slot seven bytes to a word under a
	 * marker byte, so that no word of a key is 0, and every word is claimed by
	 * itself, first come first served: a word that reads 0 is set with a
	 * compare-and-swap, and whatever it holds from then on it holds for the rest
	 * of the window.  A slot whose words all equal the key's is the key's row; a
	 * word that differs sends the key to the next slot - for good, so all
	 * threads with this key pass the slot the same way, and the key gets one row.
	 * (Two keys that share their first words may both write words of one slot;
	 * the slot ends up as the key of whoever set the first word they differ in.)
	 * Nobody waits for anybody, nothing is ordered: one round of loads per probe
	 * step where a state word ahead of the key cost two and a release per insert. */
	uint32_t row = XE_FWM_NOROW;
	if constexpr (MKW > 1) {
		constexpr int MK7 = (MKW * 8 + 6) / 7;
		uint64_t e[MK7];
#pragma unroll
		for (int j = 0; j < MK7; j++) {
			const int bit = 56 * j, w = bit >> 6, sh = bit & 63;
			uint64_t x = (w < MKW) ? kw[w] >> sh : 0;
			if ((sh > 8) && (w + 1 < MKW)) x |= kw[w + 1] << (64 - sh);
			e[j] = (x & 0x00FFFFFFFFFFFFFFull) | XE_FWM_WORD_MARK;
		}
		while (!done) {
			unsigned long long *S = F.slots + (size_t)slot * F.swords;
			unsigned long long cur[MK7];
#pragma unroll
			for (int j = 0; j < MK7; j++) {
				cur[j] = ((uint32_t)j < F.kw7) ? __hip_atomic_load(&S[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e[j];
			}
			bool eq = true;
#pragma unroll
			for (int j = 0; j < MK7; j++) {
				if (!eq || ((uint32_t)j >= F.kw7)) continue;
				unsigned long long c = cur[j];
				if (c == 0) {
					c = atomicCAS(&S[j], 0ull, (unsigned long long)e[j]);
					if (c == 0) c = e[j];
				}
				eq = c == e[j];
			}
			if (eq) {
				row = slot;
				done = true;
			} else {
				slot = (slot + 1) & F.cap_mask;
				if (++tried > F.cap_mask) {
					/* full: the records behind this key are lost, and counted */
					atomicAdd(F.dropped, (unsigned long long)nrecords);
					done = true;
				}
			}
		}
	}
	return row;
}

/* the sums of the wave's lanes into their rows (row = XE_FWM_NOROW: nothing to
 * add).  All lanes of the wave call this together.  The adds of one
 * instruction that fall into one 32-byte piece of memory travel as one
 * request, and requests are what the adds cost (measured on MI355X, 4.2 M
 * records with two sums each into 190 000 rows: 0.376 ms when every lane adds
 * its own two sums one after the other, 0.190 ms when the sums of a row leave
 * from neighbouring lanes of one instruction - the time of one sum per
 * record).  So the lanes take the sums in turns: with two sums per row
 * instruction g of 2 covers the rows of lanes 32 g ... 32 g + 31, lane l adding
 * sum l & 1 of the row of lane 32 g + (l >> 1); with three or four sums four
 * instructions cover 16 rows each. */
template <int MNV>
__device__ __forceinline__ void
fwm_wave_add(const xe_fwm_dev &F, uint32_t row, const unsigned long long (&val)[MNV])
{
	static_assert(MNV <= 4, "the lanes share out at most four sums");
	const uint32_t lane = threadIdx.x & 63;
	if (F.nval == 1) {
		if (row != XE_FWM_NOROW) atomicAdd(&F.vals[row], val[0]);
		return;
	}
	const uint32_t shift = (F.nval == 2) ? 1 : 2, per = 1u << shift;
	const uint32_t j = lane & (per - 1);
	const unsigned long long any = __ballot(row != XE_FWM_NOROW);
	for (uint32_t g = 0; g < per; g++) {
		const uint32_t first = g * (64 >> shift);
		const unsigned long long part = (any >> first) & ((shift == 1) ? 0xFFFFFFFFull : 0xFFFFull);
		if (!part) continue;
		const int src = (int)(first + (lane >> shift));
		const uint32_t r = __shfl(row, src, 64);
		unsigned long long v = 0;
#pragma unroll
		for (int q = 0; q < MNV; q++) {
			const unsigned long long x = __shfl(val[q], src, 64);
			if ((uint32_t)q == j) v = x;
		}
		if ((r != XE_FWM_NOROW) && (j < F.nval)) atomicAdd(&F.vals[(size_t)r * F.nval + j], v);
	}
}

/* LDS of a block, sized at launch by the longest key (kwmax words) and the
 * most sums (nvmax) among the tables: a block of the usual short keys takes a
 * quarter of what the limits would, and three times as many blocks fit a CU */
__host__ __device__ inline size_t
fwm_lds_bytes(uint32_t kwmax, uint32_t nvmax)
{
	const size_t waves = XE_BLOCK / 64;
	return waves * 128 * 4                         /* s_lane */
		+ waves * 64 * 4                           /* s_cnt */
		+ XE_FWM_CACHE * 4 * 2                     /* c_state, c_cnt */
		+ XE_FWM_CACHE * 8                         /* c_hash */
		+ waves * 64 * (size_t)nvmax * 8           /* s_sum */
		+ waves * (size_t)kwmax * 64 * 8           /* s_key */
		+ XE_FWM_CACHE * (size_t)kwmax * 8         /* c_key */
		+ XE_FWM_CACHE * (size_t)nvmax * 8;        /* c_sum */
}

/* a function form as a key field (monit-objects.c:679-717, :827-876), in the
 * form the table keeps it: div / div_r / div_l and min as the bytes of
 * htobe64(result), tfstr as the flags byte, portstr as the port, ppstr as
 * port1 | port2 << 16; a, b = the arguments' values */
__device__ __forceinline__ uint64_t
fwm_func_key(const xe_fwm_dev &F, int k, uint64_t a, uint64_t b)
{
	const uint32_t fn = F.kfunc[k];
	if (fn == XE_FN_MIN) return __builtin_bswap64((a < b) ? a : b);
	if (fn == XE_FN_TFSTR) {
		/* byte 1 of a 2-byte field, else byte 0 of the storage (monit-objects.c:836-843) */
		const uint32_t colw = c_colw[F.kfield[k]];
		return (colw == 2) ? (a & 0xFF) : ((a >> (8 * (colw - 1))) & 0xFF);
	}
	if (fn == XE_FN_PORTSTR) return a & 0xFFFF;
	if (fn == XE_FN_PPSTR) return (a & 0xFFFF) | ((b & 0xFFFF) << 16);
	/* xdiv(), filter.h:396-418: the quotient is an int from the start */
	int32_t q = 0;
	if (b != 0) {
		q = (int32_t)(a / b);
		if (fn == XE_FN_DIV_L) {
			const int32_t *pc = F.dl + F.dl_off[k];
			int32_t v = 0;
			for (uint32_t i = 0; i < F.dl_n[k]; i++) {
				if ((q >= pc[3 * i]) && (q <= pc[3 * i + 1])) v = pc[3 * i + 2];
			}
			q = v;
		} else if (F.kk[k] > 1) {
			q = (q / F.kk[k]) * F.kk[k];
		}
	}
	return __builtin_bswap64((uint64_t)(int64_t)q);
}

/* the column elements the key and the sums of one record need, all loads in
 * flight together (a 16-byte field takes two); a field no template of the
 * batch carries reads as zeros */
template <int MKF, int MNV>
struct fwm_cols
{
	uint64_t part0[MKF], part1[MKF], vraw[MNV];
	uint32_t rate;
};

template <int MKF, int MNV>
__device__ __forceinline__ void
fwm_load_cols(const xe_kparams &kp, const xe_fwm_dev &F, uint32_t r, bool hit, fwm_cols<MKF, MNV> &C)
{
#pragma unroll
	for (int k = 0; k < MKF; k++) {
		C.part0[k] = C.part1[k] = 0;
		if (!hit || ((uint32_t)k >= F.nkey)) continue;
		const uint32_t fld = F.kfield[k], colw = c_colw[fld];
		const uint64_t co = kp.col_off[fld];
		if (fld == XF_dev_ip) {
			C.part0[k] = kp.devips[r];                      /* comes from the exporter */
		} else if (co != ~0ull) {
			const uint8_t *p = kp.out_base + co + (size_t)r * colw;
			if (colw == 1) C.part0[k] = *p;
			else if (colw == 2) C.part0[k] = *(const uint16_t *)p;
			else if (colw == 4) C.part0[k] = *(const uint32_t *)p;
			else if (colw == 8) C.part0[k] = *(const uint64_t *)p;
			else { C.part0[k] = ((const uint64_t *)p)[0]; C.part1[k] = ((const uint64_t *)p)[1]; }
		}
		if (F.kfunc[k] != XE_FN_NONE) {
			/* a function form: both parts hold the arguments as get_nf_val()
			 * (filter.h:370-394) reads them, fields of 1, 2, 4 or 8 bytes as
			 * big-endian numbers (the host refuses longer ones) */
			const uint32_t f2 = F.kfield2[k], colw2 = c_colw[f2];
			const uint64_t co2 = kp.col_off[f2];
			uint64_t b = 0;
			const bool two = (F.kfunc[k] != XE_FN_TFSTR) && (F.kfunc[k] != XE_FN_PORTSTR);
			if (two && (co2 != ~0ull)) {
				const uint8_t *p = kp.out_base + co2 + (size_t)r * colw2;
				if (colw2 == 1) b = *p;
				else if (colw2 == 2) b = *(const uint16_t *)p;
				else if (colw2 == 4) b = *(const uint32_t *)p;
				else b = *(const uint64_t *)p;
			}
			C.part0[k] = __builtin_bswap64(C.part0[k]) >> (64 - 8 * colw);
			C.part1[k] = two ? __builtin_bswap64(b) >> (64 - 8 * colw2) : 0;
		}
	}
#pragma unroll
	for (int j = 0; j < MNV; j++) {
		C.vraw[j] = 0;
		if (!hit || ((uint32_t)j >= F.nval)) continue;
		const uint64_t co = kp.col_off[F.vfield[j]];
		if (co != ~0ull) C.vraw[j] = ((const uint64_t *)(kp.out_base + co))[r];
	}
	C.rate = hit ? kp.rates[r] : 0u;
}

/* MKW, MKF, MNV: the most key words, key fields and sums the kernel is built
 * for (registers are what limits the waves in flight, and the arrays of a
 * record's key and sums are most of them: 189 registers and two waves per SIMD
 * for the limits of the interface, four and more for the usual short keys) */
template <int MKW, int MKF, int MNV>
__global__ void __launch_bounds__(XE_BLOCK)
xe_fwm_kernel(const xe_kparams kp, const xe_fwm_dev *tables, uint32_t ntables, uint32_t kwmax, uint32_t nvmax, uint32_t nchunks)
{
	extern __shared__ uint64_t fwm_lds[];
	constexpr uint32_t WAVES = XE_BLOCK / 64;
	/* 8-byte arrays first, then the 4-byte ones */
	/* the block's table: state 0 free, 1 being written, 2 ready */
	uint64_t *c_hash = fwm_lds;                                   /* [XE_FWM_CACHE] */
	uint64_t *c_key_ = c_hash + XE_FWM_CACHE;                     /* [XE_FWM_CACHE][kwmax] */
	unsigned long long *c_sum_ = (unsigned long long *)(c_key_ + (size_t)XE_FWM_CACHE * kwmax);   /* [XE_FWM_CACHE][nvmax] */
	/* per wave: what the speaker of a hash value collects */
	unsigned long long *s_sum_ = c_sum_ + (size_t)XE_FWM_CACHE * nvmax;      /* [wave][lane][nvmax] */
	/* where a lane puts its key together, byte by byte: [wave][word][lane] */
	uint64_t *s_key_ = (uint64_t *)(s_sum_ + (size_t)WAVES * 64 * nvmax);    /* [wave][kwmax][64] */
	uint32_t *c_state = (uint32_t *)(s_key_ + (size_t)WAVES * kwmax * 64);   /* [XE_FWM_CACHE] */
	uint32_t *c_cnt = c_state + XE_FWM_CACHE;
	uint32_t *s_cnt_ = c_cnt + XE_FWM_CACHE;                      /* [wave][64] */
	uint32_t *s_lane_ = s_cnt_ + WAVES * 64;                      /* [wave][128]: which lane speaks for a hash value */
#define c_key(CI, I) c_key_[(size_t)(CI) * kwmax + (I)]
#define c_sum(CI, J) c_sum_[(size_t)(CI) * nvmax + (J)]
#define s_sum(W, L, J) s_sum_[((size_t)(W) * 64 + (L)) * nvmax + (J)]
#define s_key(W, I, L) s_key_[((size_t)(W) * kwmax + (I)) * 64 + (L)]
#define s_cnt(W, L) s_cnt_[(W) * 64 + (L)]
#define s_lane(W, H) s_lane_[(W) * 128 + (H)]
	const uint32_t ntot = kp.nrec + kp.nsf;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t r0 = blockIdx.x * (XE_BLOCK * nchunks);
	XE_PHASE_DECL;

	for (uint32_t f = 0; f < ntables; f++) {
		const xe_fwm_dev F = tables[f];

		for (uint32_t i = threadIdx.x; i < XE_FWM_CACHE; i += XE_BLOCK) {
			c_state[i] = 0;
			c_cnt[i] = 0;
#pragma unroll
			for (int j = 0; j < MNV; j++) if ((uint32_t)j < nvmax) c_sum(i, j) = 0;
		}
		__syncthreads();

		/* Two chunks ahead the mask word of a record is fetched, one chunk ahead the
		 * column elements its key and sums need: what a chunk waits for was asked
		 * for while the chunk before it went through the tables */
		fwm_cols<MKF, MNV> cnext;
		uint64_t m1, m2;
		bool hit_next;
		{
			const uint32_t ra = r0 + threadIdx.x, rb = ra + XE_BLOCK;
			const uint64_t m0 = (ra < ntot) ? kp.masks[(size_t)ra * kp.mask_words + (F.mo >> 6)] : 0;
			m1 = (rb < ntot) ? kp.masks[(size_t)rb * kp.mask_words + (F.mo >> 6)] : 0;
			hit_next = ((m0 >> (F.mo & 63)) & 1) != 0;
			fwm_load_cols(kp, F, ra, hit_next, cnext);
		}
		for (uint32_t chunk = 0; chunk < nchunks; chunk++) {
			const uint32_t r = r0 + chunk * XE_BLOCK + threadIdx.x;
			const bool hit = hit_next;
			const fwm_cols<MKF, MNV> C = cnext;
			m2 = ((chunk + 2 < nchunks) && (r + 2 * XE_BLOCK < ntot))
				? kp.masks[(size_t)(r + 2 * XE_BLOCK) * kp.mask_words + (F.mo >> 6)] : 0;
			hit_next = (chunk + 1 < nchunks) && (((m1 >> (F.mo & 63)) & 1) != 0);
			fwm_load_cols(kp, F, r + XE_BLOCK, hit_next, cnext);
			m1 = m2;
			XE_PHASE(0);
			if (!__ballot(hit)) continue;

			/* the key: field bytes one after the other, packed into words */
			uint64_t kw[MKW];
			unsigned long long val[MNV];
#pragma unroll
			for (int i = 0; i < MKW; i++) kw[i] = 0;
#pragma unroll
			for (int j = 0; j < MNV; j++) val[j] = 0;
			uint64_t h = 0xcbf29ce484222325ull;
			if (hit) {
				uint32_t pos = 0;
				const uint64_t (&part0)[MKF] = C.part0;
				const uint64_t (&part1)[MKF] = C.part1;
				const uint64_t (&vraw)[MNV] = C.vraw;
				const uint64_t rate = C.rate;
				for (uint32_t i = 0; i < F.kwords; i++) s_key(wave, i, lane) = 0;
#pragma unroll
				for (int k = 0; k < MKF; k++) {
					if ((uint32_t)k >= F.nkey) continue;
					const uint32_t size = F.ksize[k];
					uint64_t part[2] = { part0[k], part1[k] };
					if (F.kfunc[k] != XE_FN_NONE) part[0] = fwm_func_key(F, k, part0[k], part1[k]);
					/* its first `size` bytes go behind what the key holds so far */
#pragma unroll
					for (int h2 = 0; h2 < 2; h2++) {
						if (size <= 8u * h2) break;
						const uint32_t nb = (size - 8u * h2 < 8u) ? size - 8u * h2 : 8u;
						const uint64_t x = (nb < 8) ? (part[h2] & ((1ull << (8 * nb)) - 1)) : part[h2];
						const uint32_t w = pos >> 3, sh = (pos & 7) * 8;
						s_key(wave, w, lane) |= x << sh;
						if (sh && (sh + 8 * nb > 64)) s_key(wave, w + 1, lane) |= x >> (64 - sh);
						pos += nb;
					}
				}
#pragma unroll
				for (int i = 0; i < MKW; i++) {
					if ((uint32_t)i < F.kwords) {
						kw[i] = s_key(wave, i, lane);
						h = (h ^ kw[i]) * 0x100000001b3ull;
					}
				}
				/* final mix: keys that differ in a few low bits must spread over
				 * all the index bits used below */
				h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
				h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
				h ^= h >> 33;
#pragma unroll
				for (int j = 0; j < MNV; j++) {
					if ((uint32_t)j >= F.nval) continue;
					val[j] = __builtin_bswap64(vraw[j]) * F.vscale[j] * rate;
				}
			}

			XE_PHASE(1);
			/* equal keys of the wave: one speaker per hash value; lanes whose key
			 * is the speaker's hand it their sums, the others go on themselves */
			const uint32_t hs = (uint32_t)(h >> 7) & 127;
			if (hit) {
				s_lane(wave, hs) = lane;
#pragma unroll
				for (int j = 0; j < MNV; j++) if ((uint32_t)j < nvmax) s_sum(wave, lane, j) = 0;
				s_cnt(wave, lane) = 0;
			}
			__builtin_amdgcn_wave_barrier();
			const uint32_t spk = hit ? s_lane(wave, hs) : lane;
			bool same = hit;
#pragma unroll
			for (int i = 0; i < MKW; i++) {
				if ((uint32_t)i >= F.kwords) break;
				const uint64_t other = __shfl(kw[i], (int)spk, 64);
				same = same && (other == kw[i]);
			}
			if (same) {
#pragma unroll
				for (int j = 0; j < MNV; j++) {
					if ((uint32_t)j < F.nval) atomicAdd(&s_sum(wave, spk, j), val[j]);
				}
				atomicAdd(&s_cnt(wave, spk), 1u);
			}
			__builtin_amdgcn_wave_barrier();
			const bool go = hit && (!same || (spk == lane));
			uint32_t nspoken = 1;       /* records this lane speaks for */
			if (hit && same && (spk == lane)) {
				nspoken = s_cnt(wave, lane);
#pragma unroll
				for (int j = 0; j < MNV; j++) if ((uint32_t)j < nvmax) val[j] = s_sum(wave, lane, j);
			}
			__builtin_amdgcn_wave_barrier();
			XE_PHASE(2);
			if (!__ballot(go)) continue;

			/* the block's table: one entry per hash value; taken by another key
			 * (or still being written by another wave): straight to HBM */
			const uint32_t ci = (uint32_t)(h >> 14) & (XE_FWM_CACHE - 1);
			uint32_t cs = go ? __hip_atomic_load(&c_state[ci], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 1u;
			if (cs == 0) {
				const uint32_t old = atomicCAS(&c_state[ci], 0u, 1u);
				if (old == 0) {
					c_hash[ci] = h;
#pragma unroll
					for (int i = 0; i < MKW; i++) {
						if ((uint32_t)i < F.kwords) c_key(ci, i) = kw[i];
					}
					__hip_atomic_store(&c_state[ci], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					cs = 2;
				} else {
					cs = old;
				}
			}
			bool cached = false;
			if (cs == 2) {
				bool eq = true;
#pragma unroll
				for (int i = 0; i < MKW; i++) {
					if ((uint32_t)i < F.kwords) eq = eq && (c_key(ci, i) == kw[i]);
				}
				if (eq) {
#pragma unroll
					for (int j = 0; j < MNV; j++) {
						if ((uint32_t)j < F.nval) atomicAdd(&c_sum(ci, j), val[j]);
					}
					atomicAdd(&c_cnt[ci], nspoken);
					cached = true;
				}
			}
			XE_PHASE(3);
			fwm_wave_add(F, fwm_global_find(F, go && !cached, kw, h, nspoken), val);
			XE_PHASE(4);
		}

		/* what the block collected goes to the table once */
		__syncthreads();
		static_assert(XE_FWM_CACHE % XE_BLOCK == 0, "whole waves go through the block's table");
		for (uint32_t i = threadIdx.x; i < XE_FWM_CACHE; i += XE_BLOCK) {
			const bool full = c_state[i] == 2;
			uint64_t kw[MKW];
			unsigned long long val[MNV];
#pragma unroll
			for (int k = 0; k < MKW; k++) kw[k] = (full && ((uint32_t)k < F.kwords)) ? c_key(i, k) : 0;
#pragma unroll
			for (int j = 0; j < MNV; j++) val[j] = (full && ((uint32_t)j < nvmax)) ? c_sum(i, j) : 0ull;
			fwm_wave_add(F, fwm_global_find(F, full, kw, full ? c_hash[i] : 0, c_cnt[i]), val);
		}
		__syncthreads();
		XE_PHASE(5);
	}
}
#undef c_key
#undef c_sum
#undef s_sum
#undef s_key
#undef s_cnt
#undef s_lane

int
xe_launch_fwm(const struct xe_kparams *kp, const struct xe_fwm_dev *tables, uint32_t ntables, uint32_t kwmax, uint32_t nkmax, uint32_t nvmax,
	void *stream, void **events)
{
	hipStream_t st = (hipStream_t)stream;
	hipEvent_t *ev = (hipEvent_t *)events;
	const uint32_t ntot = kp->nrec + kp->nsf;
	/* chunks per block: long runs keep the block's table useful, but the launch
	 * should hold a few blocks for every CU */
	uint32_t nchunks = XE_FWM_CHUNKS;
	while ((nchunks > 4) && ((ntot + XE_BLOCK * nchunks - 1) / (XE_BLOCK * nchunks) < 2048)) nchunks >>= 1;

	if (!ntables || !ntot) return XE_OK;
	if (!kwmax || (kwmax > XE_FWM_MAX_KWORDS) || !nvmax || (nvmax > XE_FWM_MAX_VALS) || (nkmax > XE_FWM_MAX_KEY_FIELDS)) return XE_EINVAL;
#ifdef XE_DIAG
	struct xe_kparams kcopy;
	if (getenv("XE_ABLATE")) {      /* diagnostic switches, as in xe_launch_batch() */
		kcopy = *kp;
		kcopy.pad2 = (uint32_t)atoi(getenv("XE_ABLATE"));
		kp = &kcopy;
	}
	if (getenv("XE_FWM_CHUNKS_RT")) nchunks = (uint32_t)atoi(getenv("XE_FWM_CHUNKS_RT"));
#endif
	if (ev) (void)hipEventRecord(ev[0], st);
	{
		const uint32_t per_block = XE_BLOCK * nchunks;
		const dim3 grid((ntot + per_block - 1) / per_block), block(XE_BLOCK);
		const size_t lds = fwm_lds_bytes(kwmax, nvmax);
		if ((kwmax <= 1) && (nkmax <= 4) && (nvmax <= 2))
			hipLaunchKernelGGL((xe_fwm_kernel<1, 4, 2>), grid, block, lds, st, *kp, tables, ntables, kwmax, nvmax, nchunks);
		else if ((kwmax <= 2) && (nkmax <= 4) && (nvmax <= 2))
			hipLaunchKernelGGL((xe_fwm_kernel<2, 4, 2>), grid, block, lds, st, *kp, tables, ntables, kwmax, nvmax, nchunks);
		else
			hipLaunchKernelGGL((xe_fwm_kernel<XE_FWM_MAX_KWORDS, XE_FWM_MAX_KEY_FIELDS, XE_FWM_MAX_VALS>), grid, block, lds, st,
				*kp, tables, ntables, kwmax, nvmax, nchunks);
	}
	if (ev) (void)hipEventRecord(ev[1], st);
#ifdef XE_DIAG
	if (kp->pad2 & 64) {
		/* diagnostic: print and reset the per-phase wave cycles */
		unsigned long long pc[8], z[8] = {0};
		const double nw = (double)((ntot + XE_BLOCK * nchunks - 1) / (XE_BLOCK * nchunks)) * (XE_BLOCK / 64);
		(void)hipStreamSynchronize(st);
		(void)hipMemcpyFromSymbol(pc, HIP_SYMBOL(xe_fast_phase_cycles), sizeof(pc));
		(void)hipMemcpyToSymbol(HIP_SYMBOL(xe_fast_phase_cycles), z, sizeof(z));
		fprintf(stderr, "table kernel phase cycles per wave: mask %.0f key %.0f election %.0f block table %.0f global %.0f flush %.0f\n",
			(double)pc[0] / nw, (double)pc[1] / nw, (double)pc[2] / nw, (double)pc[3] / nw, (double)pc[4] / nw, (double)pc[5] / nw);
	}
#endif
	return (hipGetLastError() == hipSuccess) ? XE_OK : XE_EHIP;
}

/* ------------------------------------------------------------------ list probe */
__global__ void
xe_trie_probe_kernel(const uint32_t *trie, xe_list_dev L, int family, const uint8_t *addrs, uint32_t n, uint8_t *out)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (family == 4) {
		const uint8_t *a = addrs + 4 * (size_t)i;
		uint32_t v = ((uint32_t)a[0] << 24) | ((uint32_t)a[1] << 16) | ((uint32_t)a[2] << 8) | a[3];
		out[i] = (uint8_t)trie_match4(trie, L, v);
	} else {
		const uint8_t *a = addrs + 16 * (size_t)i;
		uint64_t hi = 0, lo = 0;
		for (int k = 0; k < 8; k++) { hi = (hi << 8) | a[k]; lo = (lo << 8) | a[8 + k]; }
		out[i] = (uint8_t)trie_match6(trie, L, hi, lo);
	}
}

/* ------------------------------------------------------------------ launch */
static int g_tables_ready = 0;

/* Staging area of xe_launch_trie_probe_static(): addresses in, answers out.
 * It is part of the code object, so every GPU the library runs on holds its
 * own copy from the first use of the library on it until the process ends.
 * In host code the name below is only a handle for hipGetSymbolAddress(); the
 * memory is reached through the per-device address kept in g_scratch_ptr[]. */
#define XE_SCRATCH_ADDR_BYTES ((size_t)XE_GPU_NOALLOC_MAX_ADDRS * 16)
#define XE_SCRATCH_BYTES (XE_SCRATCH_ADDR_BYTES + (size_t)XE_GPU_NOALLOC_MAX_ADDRS)
static __device__ __align__(256) uint8_t scratch[XE_SCRATCH_BYTES];
#define XE_MAX_DEVICES 64
static uint8_t *g_scratch_ptr[XE_MAX_DEVICES];

static hipError_t
upload_tables(void)
{
	uint16_t mn[XE_NFIELDS], mx[XE_NFIELDS], cw[XE_NFIELDS];
	uint8_t st[XE_NFIELDS];
	hipError_t e;

	for (int i = 0; i < XE_NFIELDS; i++) {
		mn[i] = xe_fields[i].size_min;
		mx[i] = xe_fields[i].size_max;
		cw[i] = xe_fields[i].col_width;
		st[i] = xe_fields[i].type == XE_T_STR;
	}
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_fmin), mn, sizeof(mn))) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_fmax), mx, sizeof(mx))) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_colw), cw, sizeof(cw))) != hipSuccess) return e;
	if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_fstr), st, sizeof(st))) != hipSuccess) return e;
	{
		uint16_t sl[XE_NFIELDS];
		for (int f = 0; f < XE_NFIELDS; f++) sl[f] = 0xFFFF;
		for (int i = 0; i < SF_NFIELDS; i++) sl[sf_field_of(i)] = (uint16_t)((i << 8) | sf_slot_of(i));
		if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_sf_slot), sl, sizeof(sl))) != hipSuccess) return e;
	}
	return hipSuccess;
}

/* clears a batch's output area with streaming stores: what a kernel writes
 * once and nothing reads back soon should not push the list tables (or the
 * datagrams that have just arrived) out of the caches, which the runtime's own
 * fill does */
__global__ void __launch_bounds__(256)
xe_zero_fill_kernel(u32x4 *p, size_t n16)
{
	const u32x4 z = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		__builtin_nontemporal_store(z, p + i);
	}
}

extern "C" int
xe_launch_zero_fill(void *p, size_t bytes, void *stream)
{
	/* (p is 256-byte aligned, bytes a multiple of 256: the output area's layout) */
	const size_t n16 = bytes / 16;
	if (!n16) return XE_OK;
	size_t blocks = (n16 + 256 * 8 - 1) / (256 * 8);
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(xe_zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (u32x4 *)p, n16);
	return (hipGetLastError() == hipSuccess) ? XE_OK : XE_EHIP;
}

extern "C" int
xe_launch_init_tables(void)
{
	/* constant tables live per device; the caller invokes this once per
	 * handle with its device current */
	hipError_t e = upload_tables();
	int dev = -1;
	if (e != hipSuccess) return XE_EHIP;
	/* this device's copy of the static staging area (a device beyond the table
	 * only goes without the entry point that uses it) */
	if ((hipGetDevice(&dev) != hipSuccess) || (dev < 0)) return XE_EHIP;
	if (dev < XE_MAX_DEVICES) {
		static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;   /* handles may be made on several threads */
		int rc = XE_OK;
		pthread_mutex_lock(&mu);
		if (!g_scratch_ptr[dev]) {
			uint8_t *p = NULL;
			if ((hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)) != hipSuccess) || !p) rc = XE_EHIP;
			else g_scratch_ptr[dev] = p;
		}
		pthread_mutex_unlock(&mu);
		if (rc != XE_OK) return rc;
	}
	g_tables_ready = 1;
	return XE_OK;
}

extern "C" int
xe_launch_batch(const struct xe_kparams *kp, void *stream, void **events)
{
	hipStream_t st = (hipStream_t)stream;
	hipEvent_t *ev = (hipEvent_t *)events;   /* NULL or 3 start/stop pairs */
	bool fast_started = false;

	if (!g_tables_ready) return XE_ESTATE;
#ifdef XE_DIAG
	/* XE_ABLATE (diagnostic build only): bit 0 skip list lookups, 1 terms, 2 counters,
	 * 3 programs, 4 column stores, 5 packet loads.  Results are wrong with
	 * any bit set. */
	struct xe_kparams kcopy;
	if (getenv("XE_ABLATE")) {
		kcopy = *kp;
		kcopy.pad2 = (uint32_t)atoi(getenv("XE_ABLATE"));
		kp = &kcopy;
	}
#endif
	if (kp->nmo) {
		if (hipMemsetAsync(kp->counters, 0, (size_t)kp->nmo * 3 * sizeof(unsigned long long), st) != hipSuccess) {
			return XE_EHIP;
		}
	}
	/* the runs the host walk listed, then the runs the device walk listed: the
	 * same kernel over either table pair */
	for (int part = 0; part < 2; part++) {
		struct xe_kparams kpart;
		const struct xe_kparams *kq = kp;
		if (part == 1) {
			if (!kp->ntiles2) break;
			kpart = *kp;
			kpart.fruns = kp->fruns2;
			kpart.ftiles = kp->ftiles2;
			kpart.nfruns = kp->nfruns2;
			kpart.ntiles = kp->ntiles2;
			kq = &kpart;
		}
		if (!kq->ntiles) continue;
		/* grid-stride over tiles: enough blocks to fill 256 CUs several times
		 * over, few enough that the per-block counter flush stays cheap */
		uint32_t per_block = XE_BLOCK / 64;
		uint32_t blocks = (kq->ntiles + per_block - 1) / per_block;
		size_t shmem = (size_t)per_block * kq->fast_dwords * kq->fast_r * 64 * 4
			+ (size_t)kq->lds_counters * kq->nmo * 24
			+ (kq->motab_lds ? (size_t)kq->motab_words * 8 : 0);
		if (blocks > 4096) blocks = 4096;
#ifdef XE_DIAG
		if (getenv("XE_FAST_BLOCKS")) {          /* diagnostic: grid size of the fast kernel */
			uint32_t want = (uint32_t)atoi(getenv("XE_FAST_BLOCKS"));
			if ((want >= 8) && (want < blocks)) blocks = want;
		}
#endif
		/* the kernel deals the tile list to 8 groups of blocks (blockIdx % 8):
		 * every group needs at least one block */
		if (blocks < 8) blocks = 8;
		if (ev && !fast_started) (void)hipEventRecord(ev[0], st);
		fast_started = true;
		const bool lean = !kq->nrgroups && !kq->ncjgroups && !kq->ncjderived && !kq->nfprogs && (kq->lw == 1)
			&& !kq->nruledevs && !kq->ntmark;
		if (kq->lw != 1) hipLaunchKernelGGL((xe_nf_fast_kernel<1, false, XE_LK_WORDS>), dim3(blocks), dim3(XE_BLOCK), shmem, st, *kq);
		else if (kq->fast_r == 4) hipLaunchKernelGGL((xe_nf_fast_kernel<4, false, 1>), dim3(blocks), dim3(XE_BLOCK), shmem, st, *kq);
		else if (kq->fast_r == 2) hipLaunchKernelGGL((xe_nf_fast_kernel<2, false, 1>), dim3(blocks), dim3(XE_BLOCK), shmem, st, *kq);
		else if (lean) hipLaunchKernelGGL((xe_nf_fast_kernel<1, true, 1>), dim3(blocks), dim3(XE_BLOCK), shmem, st, *kq);
		else hipLaunchKernelGGL((xe_nf_fast_kernel<1, false, 1>), dim3(blocks), dim3(XE_BLOCK), shmem, st, *kq);
		if (hipGetLastError() != hipSuccess) return XE_EHIP;
#ifdef XE_DIAG
		if (kq->pad2 & 64) {
			/* diagnostic: print and reset the per-phase wave cycles */
			unsigned long long pc[8], z[8] = {0};
			(void)hipStreamSynchronize(st);
			(void)hipMemcpyFromSymbol(pc, HIP_SYMBOL(xe_fast_phase_cycles), sizeof(pc));
			(void)hipMemcpyToSymbol(HIP_SYMBOL(xe_fast_phase_cycles), z, sizeof(z));
			fprintf(stderr, "phase cycles per tile: runs %.0f stage %.0f lookups %.0f terms %.0f decode %.0f programs %.0f\n",
				(double)pc[0] / kq->ntiles, (double)pc[1] / kq->ntiles, (double)pc[2] / kq->ntiles,
				(double)pc[3] / kq->ntiles, (double)pc[4] / kq->ntiles, (double)pc[5] / kq->ntiles);
		}
#endif
	}
	if (ev && fast_started) (void)hipEventRecord(ev[1], st);
	const size_t lane_lds = (size_t)kp->lds_counters * kp->nmo * 24
		+ (kp->motab_lds ? (size_t)kp->motab_words * 8 : 0) + 16;
	if (kp->nslowrec) {
		uint32_t blocks = (kp->nslowrec + XE_BLOCK - 1) / XE_BLOCK;
		if (ev) (void)hipEventRecord(ev[2], st);
		hipLaunchKernelGGL(xe_nf_general_kernel, dim3(blocks), dim3(XE_BLOCK), lane_lds, st, *kp);
		if (ev) (void)hipEventRecord(ev[3], st);
		if (hipGetLastError() != hipSuccess) return XE_EHIP;
	}
	if (kp->nsf) {
		uint32_t blocks = (kp->nsf + XE_BLOCK - 1) / XE_BLOCK;
		size_t sf_lds = lane_lds + (size_t)XE_BLOCK * SF_STAGE_BYTES;
		xe_kparams ksf;
		if (sf_lds > 64 * 1024) {
			/* many objects: their counters go to memory directly, the staging
			 * columns keep the LDS */
			ksf = *kp;
			sf_lds -= (size_t)kp->lds_counters * kp->nmo * 24;
			ksf.lds_counters = 0;
			kp = &ksf;
		}
		if (ev) (void)hipEventRecord(ev[4], st);
		if (kp->sfdg) hipLaunchKernelGGL(xe_sf_kernel<true>, dim3(blocks), dim3(XE_BLOCK), sf_lds, st, *kp);
		else hipLaunchKernelGGL(xe_sf_kernel<false>, dim3(blocks), dim3(XE_BLOCK), sf_lds, st, *kp);
		if (ev) (void)hipEventRecord(ev[5], st);
		if (hipGetLastError() != hipSuccess) return XE_EHIP;
#ifdef XE_DIAG
		if (kp->pad2 & 64) {
			/* diagnostic: print and reset the per-phase wave cycles */
			unsigned long long pc[8], z[8] = {0};
			const double nw = (double)((kp->nsf + 63) / 64);
			(void)hipStreamSynchronize(st);
			(void)hipMemcpyFromSymbol(pc, HIP_SYMBOL(xe_fast_phase_cycles), sizeof(pc));
			(void)hipMemcpyToSymbol(HIP_SYMBOL(xe_fast_phase_cycles), z, sizeof(z));
			fprintf(stderr, "sflow phase cycles per wave: load+parse %.0f stores+image %.0f atoms %.0f (lookups %.0f range groups %.0f) objects %.0f\n",
				(double)pc[0] / nw, (double)pc[1] / nw, (double)pc[2] / nw, (double)pc[6] / nw, (double)pc[7] / nw, (double)pc[3] / nw);
		}
#endif
	}
	return XE_OK;
}

extern "C" int
xe_launch_trie_probe(const uint32_t *trie, struct xe_list_dev list, int family,
	const uint8_t *addrs, uint32_t n, uint8_t *out, void *stream)
{
	if (!n) return XE_OK;
	hipLaunchKernelGGL(xe_trie_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
		trie, list, family, addrs, n, out);
	return (hipGetLastError() == hipSuccess) ? XE_OK : XE_EHIP;
}

/* The same probe with nothing allocated: addresses and answers are staged in
 * this device's static area (g_scratch_ptr[device], never the symbol's name).
 * `device` must be the current device.  The call returns when `out` is filled. */
extern "C" int
xe_launch_trie_probe_static(int device, const uint32_t *trie, struct xe_list_dev list, int family,
	const uint8_t *addrs, uint32_t n, uint8_t *out, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	const size_t ab = (family == 4) ? 4 : 16;
	uint8_t *ptr;

	if ((device < 0) || (device >= XE_MAX_DEVICES) || !(ptr = g_scratch_ptr[device])) return XE_ESTATE;
	if (n > XE_GPU_NOALLOC_MAX_ADDRS) return XE_EINVAL;
	if (!n) return XE_OK;
	if (hipMemcpyAsync(ptr, addrs, n * ab, hipMemcpyHostToDevice, st) != hipSuccess) return XE_EHIP;
	hipLaunchKernelGGL(xe_trie_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
		trie, list, family, (const uint8_t *)ptr, n, ptr + XE_SCRATCH_ADDR_BYTES);
	if (hipGetLastError() != hipSuccess) return XE_EHIP;
	if (hipMemcpyAsync(out, ptr + XE_SCRATCH_ADDR_BYTES, n, hipMemcpyDeviceToHost, st) != hipSuccess) return XE_EHIP;
	if (hipStreamSynchronize(st) != hipSuccess) return XE_EHIP;
	return XE_OK;
}
