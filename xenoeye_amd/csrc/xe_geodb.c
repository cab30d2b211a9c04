/*
 * xe_geodb.c -- geo / AS databases of the filter functions asn(), asd() and
 * continent() ... long() (geoip.c:33-88, filter.c:753-834).
 *
 * The reference maps four files (geo4.db, geo6.db, as4.db, as6.db), each an
 * array of binary-trie nodes with a payload (geoip.h:30-55), and a function
 * term walks the trie with the bytes of its field (ip-btrie.h:88-115): from
 * the most significant bit on while a child exists, and the node it stops at
 * counts only if it is a leaf.  That is the walk of iplist_match4/6 over a
 * trie of the same shape, so a term is turned into an IP list here: same
 * nodes, and "leaf" where the database's node is a leaf whose payload meets
 * the term's operands.  The device then answers the term with the list
 * machinery it has; no database payload ever goes to the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "xe_internal.h"

static const int geo_off[8] = { 0, 3, 6, 41, 105, 156, 171, 191 };   /* continent ... long, geoip.h:13-21 */

static size_t
node_bytes(int kind)
{
	return kind ? XE_AS_NODE_BYTES : XE_GEO_NODE_BYTES;
}

int
xe_geodb_set(struct xe_host *hs, int kind, int family, const void *img, size_t bytes, char *err, size_t errlen)
{
	const size_t ns = node_bytes(kind);
	struct xe_geodb *db;
	size_t n, i;
	uint8_t *seen;

	if ((kind < 0) || (kind > 1) || ((family != 4) && (family != 6))) return XE_EINVAL;
	db = &hs->geodb[kind][family == 6];
	if (!bytes) {
		free(db->img);
		db->img = NULL;
		db->nnodes = 0;
		return XE_OK;
	}
	if (!img || (bytes % ns)) {
		snprintf(err, errlen, "database image: %zu bytes is not a whole number of %zu-byte nodes", bytes, ns);
		return XE_EPARSE;
	}
	n = bytes / ns;
	if (n > 0xFFFFFFF0u) return XE_EUNSUPPORTED;
	/* every child inside the image and behind its parent (the builder appends
	 * children, ip-btrie.h:36-49), no node with two parents: a tree, so that
	 * what is made of it below is finite */
	seen = calloc(n, 1);
	if (!seen) return XE_ENOMEM;
	for (i = 0; i < n; i++) {
		int b;
		for (b = 0; b < 2; b++) {
			uint32_t next;
			memcpy(&next, (const uint8_t *)img + i * ns + 4 * (size_t)b, 4);
			if (!next) continue;
			if ((next >= n) || (next <= i) || seen[next]) {
				free(seen);
				snprintf(err, errlen, "database image: node %zu has a child (%u) that is outside the image, in front of it or shared", i, next);
				return XE_EPARSE;
			}
			seen[next] = 1;
		}
	}
	free(seen);
	{
		uint8_t *copy = malloc(bytes);
		if (!copy) return XE_ENOMEM;
		memcpy(copy, img, bytes);
		free(db->img);
		db->img = copy;
		db->nnodes = n;
	}
	return XE_OK;
}

int
xe_geodb_list(struct xe_host *hs, int func, int geo_field, int family,
	const int32_t *lo, const int32_t *hi, const char *const *strs, size_t n)
{
	const int kind = (func != XE_FN_GEO);
	const struct xe_geodb *db = &hs->geodb[kind][family == 6];
	const size_t ns = node_bytes(kind);
	struct xe_iplists *reg = &hs->lists;
	struct xe_iplist *nl, *L;
	struct xe_bnode *nodes;
	size_t i, k;

	if (!db->img || !db->nnodes) return XE_ESTATE;
	if ((func == XE_FN_GEO) && ((geo_field < 0) || (geo_field > 7))) return XE_EINVAL;
	nodes = calloc(db->nnodes, sizeof(*nodes));
	if (!nodes) return XE_ENOMEM;
	for (i = 0; i < db->nnodes; i++) {
		const uint8_t *nd = db->img + i * ns, *pay = nd + 12;
		int leaf, hit = 0;

		memcpy(nodes[i].next, nd, 8);
		memcpy(&leaf, nd + 8, 4);
		if (leaf) {
			if (func == XE_FN_ASN) {
				uint32_t asn;
				memcpy(&asn, pay, 4);
				for (k = 0; k < n; k++) if (((int32_t)asn >= lo[k]) && ((int32_t)asn <= hi[k])) hit = 1;
			} else {
				const char *res = (func == XE_FN_ASD) ? (const char *)pay + 4 : (const char *)pay + geo_off[geo_field];
				/* the payload's text ends at its first NUL or with its array
				 * (the reference's strcmp would run on into the next field) */
				const size_t cap = (func == XE_FN_ASD) ? 200 : (size_t)((geo_field < 7 ? geo_off[geo_field + 1] : 214) - geo_off[geo_field]);
				const size_t rl = strnlen(res, cap);
				for (k = 0; k < n; k++) if (strs[k] && (strlen(strs[k]) == rl) && !memcmp(strs[k], res, rl)) hit = 1;
			}
		}
		nodes[i].is_leaf = hit;
	}
	nl = realloc(reg->l, (reg->n + 1) * sizeof(*nl));
	if (!nl) { free(nodes); return XE_ENOMEM; }
	reg->l = nl;
	L = &reg->l[reg->n];
	memset(L, 0, sizeof(*L));
	/* a name no filter text can produce (the lexer never yields control characters) */
	snprintf(L->name, sizeof(L->name), "\001fn%d.%d.%d.%zu", func, geo_field, family, reg->n);
	if (family == 4) { L->n4 = nodes; L->c4 = L->cap4 = db->nnodes; }
	else { L->n6 = nodes; L->c6 = L->cap6 = db->nnodes; }
	return (int)reg->n++;
}
