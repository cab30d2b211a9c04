/*
 * ref_sflow_harness.c -- TEST INFRASTRUCTURE.
 *
 * Builds the reference's own sFlow datagram walk (sflow-impl.h) and raw
 * header parser (rawparse.h), unmodified and from where they lie in the
 * reference checkout, into oracle/_ref/libref_sflow.so.  Both files are
 * hook-macro templates meant to be instantiated by an including file (the
 * reference does so twice: sflow.c and xesflow.c).  sflow.c itself cannot be
 * compiled here (it pulls in pcap.h / tkvdb.h / aajson.h through filter.h),
 * so the hooks below restate sflow.c:26-77 and the two glue functions
 * sflow.c:131-171, minus device rules, debug printing and the monitoring
 * object walk: every record that would reach process_mo_sflow_rec() is
 * captured instead.
 *
 * Only compiled when /root/reference exists (see Makefile).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <endian.h>
#include <arpa/inet.h>

#include "utils.h"
#include "flow-info.h"
#include "sflow.h"
#include <netinet/if_ether.h>
#include <netinet/ip.h>
#include <netinet/ip6.h>

#undef LOG
#define LOG(...)

/*
 * Hook bodies: which bytes of each parsed header land in which flow field,
 * as sflow.c:26-77 defines them.  Every copy is left-aligned in the field
 * (so a 1-byte value sits in byte 0 of a 2-byte field), the IPv6 destination
 * is filled from the SOURCE address, and the IPv6 protocol is the low byte of
 * the parser's local 64-bit next-header variable.
 */
#define PUT(FLOW, FIELD, SRC, LEN) \
	(memcpy((FLOW)->FIELD, (SRC), (LEN)), (FLOW)->has_##FIELD = 1, (FLOW)->FIELD##_size = (LEN))

#define USER_TYPE struct flow_info *

static inline void
hk_eth(struct flow_info *fl, struct ethhdr *e)
{
	PUT(fl, dst_mac, e->h_dest, 6);
	PUT(fl, src_mac, e->h_source, 6);
}

static inline void
hk_ip4(struct flow_info *fl, struct iphdr *ip)
{
	PUT(fl, ip4_src_addr, &ip->saddr, 4);
	PUT(fl, ip4_dst_addr, &ip->daddr, 4);
	PUT(fl, src_tos, &ip->tos, 1);
	PUT(fl, dst_tos, &ip->tos, 1);
	PUT(fl, min_ttl, &ip->ttl, 1);
	PUT(fl, max_ttl, &ip->ttl, 1);
	PUT(fl, protocol, &ip->protocol, 1);
}

static inline void
hk_ip6(struct flow_info *fl, struct ip6_hdr *ip6, uint64_t *next_header)
{
	PUT(fl, ip6_src_addr, &ip6->ip6_src, 16);
	PUT(fl, ip6_dst_addr, &ip6->ip6_src, 16);
	PUT(fl, min_ttl, &ip6->ip6_ctlun.ip6_un1.ip6_un1_hlim, 1);
	PUT(fl, max_ttl, &ip6->ip6_ctlun.ip6_un1.ip6_un1_hlim, 1);
	PUT(fl, protocol, next_header, 1);
}

#define ON_ETH(D, V) hk_eth(D, V)
#define ON_VLAN1(D, V) do { PUT(D, src_vlan, &V->h_vlan_TCI, 2); PUT(D, dst_vlan, &V->h_vlan_TCI, 2); } while (0)
#define ON_VLAN2(D, V) do { PUT(D, src_dot1q_cvlan, &V->h_vlan_TCI, 2); PUT(D, dst_dot1q_cvlan, &V->h_vlan_TCI, 2); } while (0)
#define ON_IP(D, V) hk_ip4(D, V)
#define ON_IP6(D, V) hk_ip6(D, V, &nexthdr)
#define ON_TCP(D, V) do { PUT(D, l4_src_port, &V->th_sport, 2); PUT(D, l4_dst_port, &V->th_dport, 2); PUT(D, tcp_flags, &V->th_flags, 1); } while (0)
#define ON_UDP(D, V) do { PUT(D, l4_src_port, &V->uh_sport, 2); PUT(D, l4_dst_port, &V->uh_dport, 2); } while (0)
#define ON_ICMP(D, V) PUT(D, icmp_type, &V->type, 1)
#define ON_PAYLOAD(D, V) ((D)->payload_ptr = (V))

#include "rawparse.h"

static struct flow_info *captured = NULL;
static size_t n_captured = 0, cap_captured = 0;

static inline int
sf5_parsed(struct sfdata *s, uint8_t *p, uint32_t header_len)
{
	(void)p;
	(void)header_len;
	if (n_captured == cap_captured) {
		cap_captured = cap_captured ? cap_captured * 2 : 64;
		captured = realloc(captured, cap_captured * sizeof(struct flow_info));
	}
	captured[n_captured] = *s->flow;
	captured[n_captured].payload_ptr = NULL;
	n_captured++;
	return 1;
}

/* sflow.c:159-171 */
static inline int
sf5_eth(struct sfdata *s, uint8_t *p, enum RP_TYPE t, uint32_t header_len)
{
	uint8_t *end = p + header_len;

	if (rawpacket_parse(p, end, t, s->flow) < RP_PARSER_STATE_NO_IP) {
		return 1;
	}
	return sf5_parsed(s, p, header_len);
}

#include "sflow-impl.h"

static struct flow_packet_info ref_fpi;

int
ref_sflow_process(const uint8_t *pkt, int len)
{
	n_captured = 0;
	memset(&ref_fpi, 0, sizeof(ref_fpi));
	if (len > MAX_NF_PACKET_SIZE) {
		len = MAX_NF_PACKET_SIZE;
	}
	memcpy(ref_fpi.rawpacket, pkt, (size_t)len);
	return sflow_process(NULL, 0, &ref_fpi, len);
}

size_t ref_sflow_nrec(void) { return n_captured; }
const void *ref_sflow_rec(size_t i) { return &captured[i]; }
size_t ref_flow_info_size(void) { return sizeof(struct flow_info); }
