/*
 * xe_sfwalk.h -- the sFlow v5 datagram walk as device code shared by the
 * header-walk kernels (xe_walk.hip) and the decode kernel (xe_kernels.hip).
 *
 * sflow_process() and sf5_flow() (sflow-impl.h:293-405, :68-272) read a
 * datagram word by word through READ32_H (sflow-impl.h:51-65: a read that would
 * pass the end of the datagram ends the walk).  Here the same cursor arithmetic
 * is an iterator that hands out, one by one, the elements that yield a flow
 * record (tag 1: sampled header, tag 3: IPv4 structure).  Where a flow sample
 * starts with at least XE_SF_WINDOW bytes of datagram left, its header and the
 * header of its first element are fetched with five 16-byte loads issued
 * together and read from registers: every word the reference would read is
 * inside those bytes, so the bounds rule cannot fire and the result is the
 * same; 64 strided cursors reading 4 bytes at a time are what made the first
 * version of this walk slow.
 */
#ifndef xe_sfwalk_h_included
#define xe_sfwalk_h_included

#include <hip/hip_runtime.h>
#include <stdint.h>

#define XE_SF_WINDOW 80       /* type word + expanded sample header (12 words) + element header (6 words), rounded to 16 */

typedef uint32_t xsf_u32x4 __attribute__((ext_vector_type(4)));

/* one element that yields a record */
struct xe_sfel
{
	uint32_t hdr_pos;        /* sampled header / IPv4 structure: byte offset from the datagram start */
	uint32_t header_len;     /* tag 1: declared header length; tag 3: 28 */
	uint32_t frame_len_be;   /* 4 bytes as on the wire: in_bytes[4..8] */
	uint32_t rate;
	uint32_t in_if_be, out_if_be;   /* input_snmp / output_snmp bytes as stored */
	uint32_t sample_pos;     /* byte offset of the sample's type word */
	uint8_t kind;            /* 1 sampled header, 3 IPv4 structure */
	uint8_t rp_type;         /* 0 ethernet, 1 IPv4, 2 IPv6 */
	uint8_t chained;         /* an earlier element of the same sample yielded a record too: the flow carries over */
	uint8_t first;           /* read through the window as the first element of its sample */
};

struct xe_sfiter
{
	const uint8_t *p;        /* datagram start */
	int64_t pos, len;
	uint32_t ns_left;        /* samples the datagram header still announces */
	uint32_t nel_left;       /* elements the current flow sample still announces */
	uint32_t rate, in_if_be, out_if_be, sample_pos;
	bool bad, chained;
};

static __device__ __forceinline__ uint32_t
xsf_ld32(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

/* READ_SF_BYTES / READ32_H, sflow-impl.h:51-65 */
static __device__ __forceinline__ uint32_t
xsf_raw32(xe_sfiter &c)
{
	if (c.bad || ((c.pos + 4) > c.len)) { c.bad = true; return 0; }
	const uint32_t v = xsf_ld32(c.p + c.pos);
	c.pos += 4;
	return v;
}

static __device__ __forceinline__ uint32_t
xsf_u32(xe_sfiter &c)
{
	return __builtin_bswap32(xsf_raw32(c));
}

/* the datagram header, sflow-impl.h:293-370.  Returns false when the datagram
 * yields nothing; else agent_ver is 4 or 6, agent_pos / devid_pos are the byte
 * offsets of the agent address and the sub-agent id, and the iterator stands
 * at the first sample */
static __device__ __forceinline__ bool
xsf_begin(xe_sfiter &c, const uint8_t *p, int64_t len, uint32_t &agent_ver)
{
	c.p = p;
	c.pos = 0;
	c.len = (len > 65536) ? 65536 : len;
	c.bad = false;
	c.chained = false;
	c.ns_left = c.nel_left = 0;
	c.rate = c.in_if_be = c.out_if_be = c.sample_pos = 0;
	agent_ver = 0;
	if (xsf_u32(c) != 5) return false;
	const uint32_t v = xsf_u32(c);
	if (c.bad) return false;
	if (v == 1) {
		agent_ver = 4;
		if ((c.pos + 4) > c.len) return false;
		c.pos += 4;
	} else if (v == 2) {
		agent_ver = 6;
		if ((c.pos + 16) > c.len) return false;
		c.pos += 16;
	} else {
		return false;
	}
	(void)xsf_raw32(c);      /* sub-agent id: dev_id */
	(void)xsf_u32(c);        /* sequence */
	(void)xsf_u32(c);        /* uptime */
	c.ns_left = xsf_u32(c);
	if (c.bad) return false;
	return true;
}
#define XE_SF_AGENT_POS 8     /* the agent address starts here; the sub-agent id follows it */

static __device__ __forceinline__ int64_t
xsf_padded(uint32_t hl)
{
	return (hl % 4) ? (int64_t)hl + (4 - (hl % 4)) : (int64_t)hl;
}

/* a flow sample read through the window: the words at `sp` (its type word) */
struct xe_sfwin { uint32_t w[XE_SF_WINDOW / 4]; };

static __device__ __forceinline__ void
xsf_window_load(xe_sfwin &W, const uint8_t *at)
{
	xsf_u32x4 q[XE_SF_WINDOW / 16];
#pragma unroll
	for (int i = 0; i < XE_SF_WINDOW / 16; i++) __builtin_memcpy(&q[i], at + 16 * i, 16);
#pragma unroll
	for (int i = 0; i < XE_SF_WINDOW / 16; i++) {
		W.w[4 * i + 0] = q[i].x; W.w[4 * i + 1] = q[i].y; W.w[4 * i + 2] = q[i].z; W.w[4 * i + 3] = q[i].w;
	}
}

/* sample header words of a window whose type word says flow sample:
 * sf5_flow(), sflow-impl.h:76-143 */
static __device__ __forceinline__ void
xsf_window_sample(const xe_sfwin &W, bool expanded, uint32_t &rate, uint32_t &in_if_be, uint32_t &out_if_be, uint32_t &nel)
{
	if (expanded) {
		rate = __builtin_bswap32(W.w[5]);
		in_if_be = W.w[9];
		out_if_be = W.w[11];
		nel = __builtin_bswap32(W.w[12]);
	} else {
		rate = __builtin_bswap32(W.w[4]);
		in_if_be = W.w[7] & __builtin_bswap32(0x3fffffffu);
		out_if_be = W.w[8] & __builtin_bswap32(0x3fffffffu);
		nel = __builtin_bswap32(W.w[9]);
	}
}

/* first element of the window's sample when it is a sampled header (tag 1);
 * returns false (and sets bad) on a header protocol the reference refuses */
static __device__ __forceinline__ bool
xsf_window_first(const xe_sfwin &W, bool expanded, uint32_t sp, xe_sfel &d, bool &is_raw, bool &bad)
{
	const int e0 = expanded ? 13 : 10;
	const uint32_t tag = __builtin_bswap32(W.w[e0]);
	is_raw = tag == 1;
	if (!is_raw) return false;
	const uint32_t hp = __builtin_bswap32(W.w[e0 + 2]);
	d.frame_len_be = W.w[e0 + 3];
	d.header_len = __builtin_bswap32(W.w[e0 + 5]);
	if (hp == 1) d.rp_type = 0;
	else if (hp == 11) d.rp_type = 1;
	else if (hp == 12) d.rp_type = 2;
	else { bad = true; return false; }
	d.kind = 1;
	d.chained = 0;
	d.first = 1;
	d.sample_pos = sp;
	d.hdr_pos = sp + (expanded ? 76u : 64u);
	return true;
}

/* the next element that yields a record; false at the end of the walk */
template <bool WINDOW>
static __device__ __forceinline__ bool
xsf_next(xe_sfiter &c, xe_sfel &d)
{
	for (;;) {
		if (c.bad) return false;
		if (c.nel_left == 0) {
			/* sflow_process(), sflow-impl.h:372-402: the next sample */
			if (c.ns_left == 0) return false;
			c.ns_left--;
			if (WINDOW && ((c.len - c.pos) >= XE_SF_WINDOW)) {
				xe_sfwin W;
				xsf_window_load(W, c.p + c.pos);
				const uint32_t type = __builtin_bswap32(W.w[0]);
				if ((type == 1) || (type == 3)) {
					const bool expanded = type == 3;
					uint32_t nel;
					bool is_raw;
					xsf_window_sample(W, expanded, c.rate, c.in_if_be, c.out_if_be, nel);
					c.sample_pos = (uint32_t)c.pos;
					c.chained = false;
					if (nel && xsf_window_first(W, expanded, (uint32_t)c.pos, d, is_raw, c.bad)) {
						d.rate = c.rate;
						d.in_if_be = c.in_if_be;
						d.out_if_be = c.out_if_be;
						c.pos = (int64_t)d.hdr_pos + xsf_padded(d.header_len);
						c.chained = true;
						c.nel_left = nel - 1;
						return true;
					}
					if (c.bad) return false;
					/* no element, or a first element of another kind: the elements go
					 * through the word-by-word code below */
					c.pos += expanded ? 52 : 40;
					c.nel_left = nel;
				} else {
					c.pos += 8 + (int64_t)__builtin_bswap32(W.w[1]);
				}
				continue;
			}
			const uint32_t v = xsf_u32(c);
			if (c.bad) return false;
			if ((v == 1) || (v == 3)) {
				/* sf5_flow(), sflow-impl.h:76-143 */
				const bool expanded = v == 3;
				uint32_t x;
				c.sample_pos = (uint32_t)(c.pos - 4);
				xsf_u32(c);                       /* length */
				xsf_u32(c);                       /* sequence */
				xsf_u32(c);                       /* source id (or class) */
				if (expanded) xsf_u32(c);         /* source index */
				c.rate = xsf_u32(c);
				xsf_u32(c);                       /* pool */
				xsf_u32(c);                       /* drops */
				if (expanded) xsf_u32(c);         /* input format */
				x = xsf_u32(c);
				if (!expanded) x &= 0x3fffffff;
				c.in_if_be = __builtin_bswap32(x);
				if (expanded) xsf_u32(c);         /* output format */
				x = xsf_u32(c);
				if (!expanded) x &= 0x3fffffff;
				c.out_if_be = __builtin_bswap32(x);
				x = xsf_u32(c);
				if (c.bad) return false;
				c.nel_left = x;
				c.chained = false;
			} else {
				const uint32_t l = xsf_u32(c);
				if (c.bad) return false;
				c.pos += l;
			}
			continue;
		}
		/* sf5_flow(), sflow-impl.h:145-270: the next element of the sample */
		c.nel_left--;
		const uint32_t tag = xsf_u32(c), ell = xsf_u32(c);
		if (c.bad) return false;
		if (tag == 1) {
			const uint32_t hp = xsf_u32(c);
			d.frame_len_be = xsf_raw32(c);
			xsf_u32(c);                           /* stripped */
			d.header_len = xsf_u32(c);
			if (c.bad) return false;
			if (hp == 1) d.rp_type = 0;
			else if (hp == 11) d.rp_type = 1;
			else if (hp == 12) d.rp_type = 2;
			else { c.bad = true; return false; }
			d.kind = 1;
			d.hdr_pos = (uint32_t)c.pos;
			c.pos += xsf_padded(d.header_len);
		} else if (tag == 1001) {
			xsf_u32(c); xsf_u32(c); xsf_u32(c); xsf_u32(c);
			continue;
		} else if (tag == 3) {
			d.frame_len_be = xsf_raw32(c);
			if (c.bad) return false;
			/* the 7 words after the length are read one by one and a short
			 * datagram aborts before the record is delivered */
			if ((c.pos + 28) > c.len) { c.bad = true; return false; }
			d.kind = 3;
			d.rp_type = 0;
			d.header_len = 28;
			d.hdr_pos = (uint32_t)c.pos;
			c.pos += 28;
		} else {
			c.pos += ell;
			continue;
		}
		d.rate = c.rate;
		d.in_if_be = c.in_if_be;
		d.out_if_be = c.out_if_be;
		d.sample_pos = c.sample_pos;
		d.chained = c.chained ? 1 : 0;
		d.first = 0;
		c.chained = true;
		return true;
	}
}

#endif
