/*
 * xe_pcap.c -- replay of capture files into datagram batches.
 *
 * The reference can take its export datagrams from a libpcap handle instead
 * of a socket: pcap_packet() (pcapture.c:111-180) strips a 14-byte Ethernet
 * header, an IPv4 header of IHL*4 bytes and the 8-byte UDP header from every
 * frame, takes the IPv4 source address as the exporter's address and hands
 * the rest to netflow_process() / sflow_process().  This file does the same
 * for frames read from a classic pcap file and appends the datagrams to the
 * arrays xe_gpu_batch_prepare() takes.  Host code only; no libpcap.
 *
 * Rules taken from pcap_packet():
 *   - frames with fewer than 12 + 20 + 8 + 24 captured bytes are ignored (:131)
 *   - IHL * 4 < 20 ignores the frame (:139-143); the IP version and the
 *     Ethernet type are not looked at (the capture filter is expected to
 *     have done that)
 *   - only protocol 17 (:146)
 *   - datagram length = IP total length - (IHL * 4 + 8), lowered to the UDP
 *     header's length field when that is smaller (:163-166: the field counts
 *     the UDP header too, the comparison is taken as it stands)
 * Where the reference's behaviour is undefined this reader is strict: a
 * length that is not positive, exceeds 64 KiB (the size of
 * flow_packet_info.rawpacket, flow-info.h:43) or reaches beyond the captured
 * bytes ignores the frame and counts it as damaged.
 *
 * The reference selects frames with a BPF expression (pcapture.c:291-305);
 * here the only selection is an optional UDP destination port.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>

#include "xe_internal.h"

#define PCAP_MAGIC_US   0xa1b2c3d4u
#define PCAP_MAGIC_NS   0xa1b23c4du
#define PCAPNG_SHB      0x0a0d0d0au
#define LINKTYPE_EN10MB 1u
#define FRAME_MAX       (256u * 1024u)

#define ETHER_LEN 14u
#define UDP_LEN   8u
#define MIN_CAPLEN (12u + 20u + 8u + 24u)
#define DATAGRAM_MAX (64u * 1024u)

struct xe_pcap
{
	FILE *f;
	int swapped;
	uint32_t snaplen;
	uint8_t *frame;             /* the frame read last */
	/* a datagram that did not fit into the caller's arrays any more */
	int pending;
	uint32_t pend_off, pend_len, pend_src;
	int eof;
	uint64_t frames, taken, ignored, damaged;
	char err[256];
};

static uint32_t
rd32(const struct xe_pcap *p, const uint8_t *b)
{
	uint32_t v;

	memcpy(&v, b, 4);
	return p->swapped ? __builtin_bswap32(v) : v;
}

int
xe_pcap_open(const char *path, xe_pcap **out)
{
	uint8_t hdr[24];
	uint32_t magic, linktype;
	struct xe_pcap *p;

	if (!path || !out) return XE_EINVAL;
	*out = NULL;
	p = calloc(1, sizeof(*p));
	if (!p) return XE_ENOMEM;
	*out = p;   /* kept on failure so that the caller can read the reason */

	p->f = fopen(path, "rb");
	if (!p->f) {
		snprintf(p->err, sizeof(p->err), "%s: %s", path, strerror(errno));
		return XE_EIO;
	}
	if (fread(hdr, 1, sizeof(hdr), p->f) != sizeof(hdr)) {
		snprintf(p->err, sizeof(p->err), "%s: shorter than a pcap file header", path);
		return XE_EPARSE;
	}
	memcpy(&magic, hdr, 4);
	if (magic == PCAP_MAGIC_US || magic == PCAP_MAGIC_NS) {
		p->swapped = 0;
	} else if (__builtin_bswap32(magic) == PCAP_MAGIC_US || __builtin_bswap32(magic) == PCAP_MAGIC_NS) {
		p->swapped = 1;
	} else if (magic == PCAPNG_SHB) {
		snprintf(p->err, sizeof(p->err), "%s: pcapng is not read here, convert it to classic pcap", path);
		return XE_EUNSUPPORTED;
	} else {
		snprintf(p->err, sizeof(p->err), "%s: not a pcap file (magic %08x)", path, magic);
		return XE_EPARSE;
	}
	p->snaplen = rd32(p, hdr + 16);
	linktype = rd32(p, hdr + 20) & 0x0FFFFFFFu;
	if (linktype != LINKTYPE_EN10MB) {
		/* pcap_packet() assumes Ethernet framing whatever the handle delivers */
		snprintf(p->err, sizeof(p->err), "%s: link type %u, only Ethernet (1) frames can be replayed",
			path, linktype);
		return XE_EUNSUPPORTED;
	}
	p->frame = malloc(FRAME_MAX);
	if (!p->frame) return XE_ENOMEM;
	return XE_OK;
}

void
xe_pcap_close(xe_pcap *p)
{
	if (!p) return;
	if (p->f) fclose(p->f);
	free(p->frame);
	free(p);
}

const char *
xe_pcap_error(const xe_pcap *p)
{
	return p ? p->err : "no reader";
}

void
xe_pcap_stats(const xe_pcap *p, uint64_t *frames, uint64_t *taken, uint64_t *ignored, uint64_t *damaged)
{
	if (frames) *frames = p ? p->frames : 0;
	if (taken) *taken = p ? p->taken : 0;
	if (ignored) *ignored = p ? p->ignored : 0;
	if (damaged) *damaged = p ? p->damaged : 0;
}

/* the next frame that carries a datagram: 1 = found (pend_* set), 0 = end
 * of file, < 0 = error */
static int
next_datagram(struct xe_pcap *p, int udp_port)
{
	for (;;) {
		uint8_t rh[16];
		uint32_t caplen, size_ip, ip_len, ulen;
		int32_t size_payload;
		const uint8_t *ip, *udp;
		size_t got;

		got = fread(rh, 1, sizeof(rh), p->f);
		if (got == 0) {
			p->eof = 1;
			return 0;
		}
		if (got != sizeof(rh)) {
			/* a capture cut off in the middle of a record: what came before stands */
			p->eof = 1;
			p->damaged++;
			return 0;
		}
		caplen = rd32(p, rh + 8);
		if (caplen > FRAME_MAX) {
			snprintf(p->err, sizeof(p->err), "frame %llu: captured length %u, the file is damaged",
				(unsigned long long)p->frames, caplen);
			return XE_EPARSE;
		}
		if (fread(p->frame, 1, caplen, p->f) != caplen) {
			p->eof = 1;
			p->damaged++;
			return 0;
		}
		p->frames++;

		if (caplen < MIN_CAPLEN) { p->ignored++; continue; }
		ip = p->frame + ETHER_LEN;
		size_ip = (ip[0] & 0x0Fu) * 4u;
		if (size_ip < 20) { p->ignored++; continue; }
		if (ip[9] != 17) { p->ignored++; continue; }
		if (ETHER_LEN + size_ip + UDP_LEN > caplen) { p->damaged++; continue; }
		udp = ip + size_ip;
		if (udp_port > 0 && (((uint32_t)udp[2] << 8) | udp[3]) != (uint32_t)udp_port) {
			p->ignored++;
			continue;
		}
		ip_len = ((uint32_t)ip[2] << 8) | ip[3];
		ulen = ((uint32_t)udp[4] << 8) | udp[5];
		size_payload = (int32_t)ip_len - (int32_t)(size_ip + UDP_LEN);
		if (size_payload > (int32_t)ulen) size_payload = (int32_t)ulen;
		if (size_payload <= 0 || (uint32_t)size_payload > DATAGRAM_MAX
			|| ETHER_LEN + size_ip + UDP_LEN + (uint32_t)size_payload > caplen) {
			p->damaged++;
			continue;
		}
		p->pend_off = ETHER_LEN + size_ip + UDP_LEN;
		p->pend_len = (uint32_t)size_payload;
		memcpy(&p->pend_src, ip + 12, 4);
		return 1;
	}
}

long
xe_pcap_read(xe_pcap *p, int udp_port, uint8_t *data, size_t cap, uint64_t *offsets,
	uint32_t *src_ip4, size_t max_datagrams)
{
	size_t n = 0, used = 0;

	if (!p || !p->f || !p->frame || !data || !offsets || !src_ip4) return XE_EINVAL;
	offsets[0] = 0;
	while (n < max_datagrams) {
		if (!p->pending) {
			int rc;

			if (p->eof) break;
			rc = next_datagram(p, udp_port);
			if (rc < 0) return rc;
			if (rc == 0) break;
			p->pending = 1;
		}
		if (p->pend_len > cap - used) {
			if (n == 0) {
				snprintf(p->err, sizeof(p->err), "a datagram of %u bytes does not fit into %zu",
					p->pend_len, cap);
				return XE_EINVAL;
			}
			break;   /* stays pending for the next call */
		}
		memcpy(data + used, p->frame + p->pend_off, p->pend_len);
		used += p->pend_len;
		src_ip4[n] = p->pend_src;
		offsets[++n] = used;
		p->pending = 0;
		p->taken++;
	}
	return (long)n;
}
