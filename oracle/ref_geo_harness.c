/*
 * ref_geo_harness.c -- TEST INFRASTRUCTURE.
 *
 * Builds the reference's own geo / AS database structures (geoip.h) and its
 * bitwise-trie macros (ip-btrie.h), unmodified and from where they lie in the
 * reference checkout, into oracle/_ref/libref_geo.so.  geoip.c itself cannot
 * be compiled here (it includes xenoeye.h, which needs pcap.h / tkvdb.h /
 * aajson.h), so the four lookup functions below instantiate IP_BTRIE_LOOKUP
 * the way geoip.c:33-88 does - over database images the caller hands in
 * instead of the files geoip.c maps - and two more instantiate IP_BTRIE_ADD
 * the way the database builder does (xemkgeodb.c), so that test images are
 * made by the reference's own insertion code as well.
 *
 * Only compiled when /root/reference exists (see Makefile).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>

#include "geoip.h"
#include "ip-btrie.h"

#undef LOG
#define LOG(...)

size_t ref_geo_node_size(int kind) { return kind ? sizeof(struct btrie_node_as) : sizeof(struct btrie_node_geo); }
size_t ref_geo_payload_offset(int kind)
{
	return kind ? offsetof(struct btrie_node_as, a) : offsetof(struct btrie_node_geo, g);
}
size_t ref_geo_field_offset(int field)
{
	struct geoip_info g;
	return (size_t)(geoip_get_field(&g, (enum GEOIP_FIELD)field) - (char *)&g);
}
int ref_geo_field_size(int field) { return geoip_get_field_size((enum GEOIP_FIELD)field); }

/* ---- lookups over a caller's image: geoip.c:33-88 ---- */
int
ref_geo_lookup4(const void *img, uint32_t addr, struct geoip_info *out)
{
	uint8_t *addr_ptr = (uint8_t *)&addr;
	struct btrie_node_geo *geo4 = (struct btrie_node_geo *)img;

	IP_BTRIE_LOOKUP(geo4, 4 * 8);
	*out = geo4[node].g;
	return 1;
}

int
ref_geo_lookup6(const void *img, const uint8_t addr[16], struct geoip_info *out)
{
	uint8_t *addr_ptr = (uint8_t *)addr;
	struct btrie_node_geo *geo6 = (struct btrie_node_geo *)img;

	IP_BTRIE_LOOKUP(geo6, 16 * 8);
	*out = geo6[node].g;
	return 1;
}

int
ref_as_lookup4(const void *img, uint32_t addr, struct as_info *out)
{
	uint8_t *addr_ptr = (uint8_t *)&addr;
	struct btrie_node_as *as4 = (struct btrie_node_as *)img;

	IP_BTRIE_LOOKUP(as4, 4 * 8);
	*out = as4[node].a;
	return 1;
}

int
ref_as_lookup6(const void *img, const uint8_t addr[16], struct as_info *out)
{
	uint8_t *addr_ptr = (uint8_t *)addr;
	struct btrie_node_as *as6 = (struct btrie_node_as *)img;

	IP_BTRIE_LOOKUP(as6, 16 * 8);
	*out = as6[node].a;
	return 1;
}

/* ---- database images made with the reference's insertion macro ---- */
struct ref_geo_db { struct btrie_node_geo *db; size_t size; };
struct ref_as_db { struct btrie_node_as *db; size_t size; };

static int
geo_add(struct ref_geo_db *d, const uint8_t *addr_ptr, int mask, const struct geoip_info *g)
{
	IP_BTRIE_ADD(d->db, d->size, btrie_node_geo);
	d->db[node].g = *g;
	return 1;
}

static int
as_add(struct ref_as_db *d, const uint8_t *addr_ptr, int mask, const struct as_info *a)
{
	IP_BTRIE_ADD(d->db, d->size, btrie_node_as);
	d->db[node].a = *a;
	return 1;
}

void *ref_geo_db_new(int kind) { return calloc(1, kind ? sizeof(struct ref_as_db) : sizeof(struct ref_geo_db)); }

int
ref_geo_db_add(void *db, int kind, const uint8_t *addr, int mask, const void *payload)
{
	if (kind) return as_add((struct ref_as_db *)db, addr, mask, (const struct as_info *)payload);
	return geo_add((struct ref_geo_db *)db, addr, mask, (const struct geoip_info *)payload);
}

/* the image (an array of nodes, what the .db files hold) and its size in bytes */
const void *
ref_geo_db_image(void *db, int kind, size_t *bytes)
{
	if (kind) { *bytes = ((struct ref_as_db *)db)->size * sizeof(struct btrie_node_as); return ((struct ref_as_db *)db)->db; }
	*bytes = ((struct ref_geo_db *)db)->size * sizeof(struct btrie_node_geo);
	return ((struct ref_geo_db *)db)->db;
}

void
ref_geo_db_free(void *db, int kind)
{
	if (!db) return;
	if (kind) free(((struct ref_as_db *)db)->db); else free(((struct ref_geo_db *)db)->db);
	free(db);
}
