/*
 * xe_fields.c -- the flow field table: element ids, size limits and the byte
 * layout of the reference's struct flow_info (netflow.def:2-64,
 * flow-info.h:10-33), plus the column width each field gets in the
 * structure-of-arrays output.
 */
#include <string.h>
#include "xe_internal.h"

/* name, element id, type, minimum size, maximum size */
#define XE_FIELD_LIST(F) \
	F(in_bytes, 1, XE_T_INT, 4, 8)              F(in_pkts, 2, XE_T_INT, 4, 8) \
	F(protocol, 4, XE_T_INT, 1, 1)              F(src_tos, 5, XE_T_INT, 1, 1) \
	F(dst_tos, 55, XE_T_INT, 1, 1)              F(tcp_flags, 6, XE_T_INT, 1, 2) \
	F(l4_src_port, 7, XE_T_INT, 2, 2)           F(ip4_src_addr, 8, XE_T_ADDR, 4, 4) \
	F(src_mask, 9, XE_T_INT, 1, 1)              F(input_snmp, 10, XE_T_INT, 2, 4) \
	F(l4_dst_port, 11, XE_T_INT, 2, 2)          F(ip4_dst_addr, 12, XE_T_ADDR, 4, 4) \
	F(dst_mask, 13, XE_T_INT, 1, 1)             F(output_snmp, 14, XE_T_INT, 2, 4) \
	F(ip4_next_hop, 15, XE_T_ADDR, 4, 4)        F(src_as, 16, XE_T_INT, 2, 4) \
	F(dst_as, 17, XE_T_INT, 2, 4)               F(bgp_next_hop, 18, XE_T_ADDR, 4, 4) \
	F(last_switched, 21, XE_T_INT, 4, 4)        F(first_switched, 22, XE_T_INT, 4, 4) \
	F(ip6_src_addr, 27, XE_T_ADDR, 16, 16)      F(ip6_dst_addr, 28, XE_T_ADDR, 16, 16) \
	F(icmp_type, 32, XE_T_INT, 2, 2)            F(sampler_id, 48, XE_T_INT, 1, 4) \
	F(min_ttl, 52, XE_T_INT, 1, 1)              F(max_ttl, 53, XE_T_INT, 1, 1) \
	F(frag_id, 54, XE_T_INT, 2, 4)              F(src_mac, 56, XE_T_MAC, 6, 6) \
	F(dst_mac, 57, XE_T_MAC, 6, 6)              F(src_vlan, 58, XE_T_INT, 2, 2) \
	F(dst_vlan, 59, XE_T_INT, 2, 2)             F(ip_protocol_version, 60, XE_T_INT, 1, 1) \
	F(direction, 61, XE_T_INT, 1, 1)            F(if_name, 82, XE_T_STR, 1, 16) \
	F(fwd_status, 89, XE_T_INT, 1, 1)           F(flow_start_ms, 152, XE_T_INT, 8, 8) \
	F(flow_end_ms, 153, XE_T_INT, 8, 8)         F(flow_end_reason, 136, XE_T_INT, 1, 1) \
	F(ip_ttl, 192, XE_T_INT, 1, 1)              F(ioctets, 231, XE_T_INT, 8, 8) \
	F(roctets, 232, XE_T_INT, 2, 8)             F(ipackets, 298, XE_T_INT, 8, 8) \
	F(rpackets, 239, XE_T_INT, 8, 8)            F(ingrs_vrf, 234, XE_T_INT, 4, 4) \
	F(egrs_vrf, 235, XE_T_INT, 4, 4)            F(src_dot1q_vlan, 243, XE_T_INT, 2, 2) \
	F(src_dot1q_cvlan, 245, XE_T_INT, 2, 2)     F(dst_dot1q_vlan, 254, XE_T_INT, 2, 2) \
	F(dst_dot1q_cvlan, 255, XE_T_INT, 2, 2)     F(obs_time_ms, 323, XE_T_INT, 8, 8) \
	F(vas_session_id, 2000, XE_T_INT, 8, 8)     F(vas_http_host, 2001, XE_T_STR, 0, 256) \
	F(dev_ip, 65500, XE_T_ADDR, 4, 4)           F(dns_name, 65510, XE_T_STR, 1, 256) \
	F(dns_ips, 65511, XE_T_STR, 1, 512)         F(sni, 65512, XE_T_STR, 1, 256) \
	F(class0, 65520, XE_T_STR, 1, 32)           F(class1, 65521, XE_T_STR, 1, 32) \
	F(class2, 65522, XE_T_STR, 1, 32)           F(class3, 65523, XE_T_STR, 1, 32) \
	F(class4, 65524, XE_T_STR, 1, 32)           F(pad1, 65530, XE_T_INT, 1, 1) \
	F(pad2, 65531, XE_T_INT, 2, 2)

/* mirror of the reference layout, used only for offsetof() */
struct layout
{
#define F(NAME, ID, TYPE, MN, MX) uint8_t NAME[MX]; int NAME##_size; int has_##NAME;
	XE_FIELD_LIST(F)
#undef F
	void *payload_ptr;
	uint8_t dev_ip6[16]; int dev_ip6_size; int has_dev_ip6;
	uint8_t dev_id[4]; int dev_id_size; int has_dev_id;
	uint8_t dev_mark[4]; int dev_mark_size; int has_dev_mark;
	uint32_t sampling_rate;
};

_Static_assert(sizeof(struct layout) == XE_FLOW_INFO_SIZE, "struct flow_info is 2312 bytes");
_Static_assert(offsetof(struct layout, payload_ptr) == XE_OFF_PAYLOAD_PTR, "payload_ptr offset");
_Static_assert(offsetof(struct layout, sampling_rate) == XE_OFF_SAMPLING_RATE, "sampling_rate offset");

/* column width: power of two that holds the maximum size (strings keep theirs) */
#define COLW(MX) ((MX) <= 1 ? 1 : (MX) <= 2 ? 2 : (MX) <= 4 ? 4 : (MX) <= 8 ? 8 : (MX) <= 16 ? 16 : (MX))

#define ENTRY(NAME, ID, TYPE, MN, MX) \
	{ #NAME, ID, TYPE, MN, MX, offsetof(struct layout, NAME), \
	  offsetof(struct layout, NAME##_size), offsetof(struct layout, has_##NAME), COLW(MX) },

const struct xe_field xe_fields[XE_NFIELDS] = {
	XE_FIELD_LIST(ENTRY)
	ENTRY(dev_ip6, -1, XE_T_ADDR, 16, 16)
	ENTRY(dev_id, -1, XE_T_INT, 4, 4)
	ENTRY(dev_mark, -1, XE_T_INT, 4, 4)
};

_Static_assert(sizeof(xe_fields) / sizeof(xe_fields[0]) == XE_NFIELDS, "field count");

int
xe_field_by_id(int id)
{
	static int16_t map[65536];
	static int ready = 0;
	int i;

	if (!ready) {
		for (i = 0; i < 65536; i++) map[i] = -1;
		for (i = 0; i < XE_NTABLE_FIELDS; i++) map[xe_fields[i].id] = (int16_t)i;
		__atomic_store_n(&ready, 1, __ATOMIC_RELEASE);
	}
	if ((id < 0) || (id > 65535)) return -1;
	return map[id];
}

int
xe_field_by_name(const char *name)
{
	int i;
	for (i = 0; i < XE_NFIELDS; i++) {
		if (strcmp(xe_fields[i].name, name) == 0) return i;
	}
	return -1;
}

int xe_field_count(void) { return XE_NFIELDS; }
const char *xe_field_name(int f) { return ((f >= 0) && (f < XE_NFIELDS)) ? xe_fields[f].name : NULL; }
int xe_field_id(int f) { return ((f >= 0) && (f < XE_NFIELDS)) ? xe_fields[f].id : -1; }
int xe_field_size_min(int f) { return ((f >= 0) && (f < XE_NFIELDS)) ? xe_fields[f].size_min : -1; }
int xe_field_size_max(int f) { return ((f >= 0) && (f < XE_NFIELDS)) ? xe_fields[f].size_max : -1; }
size_t xe_field_offset(int f) { return ((f >= 0) && (f < XE_NFIELDS)) ? xe_fields[f].off : 0; }
