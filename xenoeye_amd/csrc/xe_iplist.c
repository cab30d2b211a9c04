/*
 * xe_iplist.c -- IP-list files -> bitwise trie -> flattened device arrays.
 *
 * Behaviour follows the reference's iplist.c: the text loader of :218-297
 * (including: '#' starts a comment anywhere, the default prefix length is 32
 * for IPv6 lines too, a bad length skips the line, a last line without '\n'
 * is dropped) and the lookup of :160-216, which returns the leaf flag of the
 * DEEPEST trie node the address reaches -- not a longest-prefix match.
 *
 * The device form replaces the one-bit-per-step pointer chase with a
 * 65536-entry root table over the first 16 address bits followed by
 * 8-bit-stride nodes of one 128-byte line each (child map / result map /
 * popcount-indexed child base per 32 chunk values), so a lookup is one root
 * load plus at most (bits-16)/8 node visits of one 16-byte load each.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <dirent.h>
#include <arpa/inet.h>
#include "xe_internal.h"

int
xe_blob_push(struct xe_blob *b, const void *item, size_t sz)
{
	if ((b->n + 1) * sz > b->cap) {
		size_t ncap = b->cap ? b->cap * 2 : 4096;
		void *np;
		while (ncap < (b->n + 1) * sz) ncap *= 2;
		np = realloc(b->p, ncap);
		if (!np) return 0;
		b->p = np;
		b->cap = ncap;
	}
	memcpy((uint8_t *)b->p + b->n * sz, item, sz);
	b->n++;
	return 1;
}

/* insert a prefix: iplist.c:52-158 */
static int
bt_add(struct xe_bnode **nodes, size_t *cnt, size_t *cap, const uint8_t *addr, int plen, int maxbits)
{
	uint32_t cur = 0;
	int i;

	if (!*nodes) {
		*cap = 64;
		*nodes = calloc(*cap, sizeof(struct xe_bnode));
		if (!*nodes) return 0;
		*cnt = 1;
	}
	for (i = 0; (i < plen) && (i < maxbits); i++) {
		int b = (addr[i >> 3] >> (7 - (i & 7))) & 1;
		uint32_t nx = (*nodes)[cur].next[b];
		if (!nx) {
			if (*cnt == *cap) {
				struct xe_bnode *t = realloc(*nodes, *cap * 2 * sizeof(struct xe_bnode));
				if (!t) return 0;
				*nodes = t;
				*cap *= 2;
			}
			nx = (uint32_t)(*cnt)++;
			memset(&(*nodes)[nx], 0, sizeof(struct xe_bnode));
			(*nodes)[cur].next[b] = nx;
		}
		cur = nx;
	}
	/* a length beyond the address width puts the reference's leaf flag below
	 * the depth any lookup reaches: the path exists, the leaf is invisible */
	if (plen <= maxbits) (*nodes)[cur].is_leaf = 1;
	return 1;
}

static char *
trim(char *s)
{
	char *e;
	while (isspace((unsigned char)*s)) s++;
	if (!*s) return s;
	e = s + strlen(s) - 1;
	while ((e > s) && isspace((unsigned char)*e)) e--;
	e[1] = '\0';
	return s;
}

static int
load_stream(struct xe_iplists *reg, FILE *f, const char *name)
{
	struct xe_iplist *t = realloc(reg->l, (reg->n + 1) * sizeof(struct xe_iplist));
	struct xe_iplist *l;
	char line[1024];

	if (!t) return XE_ENOMEM;
	reg->l = t;
	l = &reg->l[reg->n];
	memset(l, 0, sizeof(*l));
	snprintf(l->name, sizeof(l->name), "%s", name);

	/* fgets with size-1 and the feof() test right after it are the
	 * reference's (iplist.c:250-256): they decide which lines count */
	while (fgets(line, sizeof(line) - 1, f)) {
		char *txt, *slash;
		uint8_t a[16];
		int plen = 32;

		if (feof(f)) break;
		line[strcspn(line, "\n#")] = '\0';
		txt = trim(line);
		if (!*txt) continue;
		slash = strchr(txt, '/');
		if (slash) {
			char *end;
			*slash++ = '\0';
			plen = (int)strtol(slash, &end, 10);
			if (*end) continue;
		}
		if (inet_pton(AF_INET, txt, a) == 1) {
			if (!bt_add(&l->n4, &l->c4, &l->cap4, a, plen, 32)) return XE_ENOMEM;
		} else if (inet_pton(AF_INET6, txt, a) == 1) {
			if (!bt_add(&l->n6, &l->c6, &l->cap6, a, plen, 128)) return XE_ENOMEM;
		}
	}
	reg->n++;
	return XE_OK;
}

int
xe_iplist_load_text(struct xe_iplists *reg, const char *name, const char *text, size_t len)
{
	FILE *f = len ? fmemopen((void *)text, len, "r") : fopen("/dev/null", "r");
	int rc;

	if (!f) return XE_EIO;
	rc = load_stream(reg, f, name);
	fclose(f);
	return rc;
}

/* iplist.c:299-322: readdir order, dot files skipped, list name = file name */
int
xe_iplist_load_dir(struct xe_iplists *reg, const char *dir)
{
	DIR *d = opendir(dir);
	struct dirent *de;

	if (!d) return XE_EIO;
	while ((de = readdir(d)) != NULL) {
		char path[4096];
		FILE *f;
		int rc;

		if (de->d_name[0] == '.') continue;
		snprintf(path, sizeof(path), "%s/%s", dir, de->d_name);
		f = fopen(path, "r");
		if (!f) continue;
		rc = load_stream(reg, f, de->d_name);
		fclose(f);
		if (rc != XE_OK) {
			closedir(d);
			return rc;
		}
	}
	closedir(d);
	return XE_OK;
}

int
xe_iplist_find(const struct xe_iplists *reg, const char *name)
{
	size_t i;
	for (i = 0; i < reg->n; i++) {
		if (strcmp(reg->l[i].name, name) == 0) return (int)i;
	}
	return -1;
}

void
xe_iplists_free(struct xe_iplists *reg)
{
	size_t i;
	for (i = 0; i < reg->n; i++) {
		free(reg->l[i].n4);
		free(reg->l[i].n6);
	}
	free(reg->l);
	reg->l = NULL;
	reg->n = 0;
}

/* ------------------------------------------------------------------ flatten */

/*
 * Expand `depth` more bits below binary node `bn`.  For every value of those
 * bits the bit-by-bit walk either stops early at some node (the entry is then
 * terminal with that node's leaf flag) or arrives at a node `depth` levels
 * down (terminal if that node has no children, else a child entry).
 * cb(ctx, first_index, count, kind, value): kind 0 = terminal(value = leaf
 * flag) for `count` consecutive chunk values, kind 1 = child (value = binary
 * node id, count = 1).
 */
typedef void (*expand_cb)(void *ctx, uint32_t first, uint32_t count, int kind, uint32_t value);

static void
expand(const struct xe_bnode *bt, uint32_t bn, int depth, uint32_t prefix, expand_cb cb, void *ctx)
{
	int b;

	if (depth == 0) {
		if (bt[bn].next[0] || bt[bn].next[1]) {
			cb(ctx, prefix, 1, 1, bn);
		} else {
			cb(ctx, prefix, 1, 0, (uint32_t)(bt[bn].is_leaf != 0));
		}
		return;
	}
	for (b = 0; b < 2; b++) {
		uint32_t nx = bt[bn].next[b];
		uint32_t base = ((prefix << 1) | (uint32_t)b) << (depth - 1);
		if (!nx) {
			/* prefix here counts bits already fixed; the block of
			 * values below shares the stop node bn */
			cb(ctx, base, 1u << (depth - 1), 0, (uint32_t)(bt[bn].is_leaf != 0));
		} else {
			expand(bt, nx, depth - 1, (prefix << 1) | (uint32_t)b, cb, ctx);
		}
	}
}

struct pending { uint32_t bnode; uint32_t dnode; };

struct fl_ctx
{
	uint32_t *entries;        /* root: 65536 entries; node: unused */
	uint32_t *node;           /* current device node words */
	struct xe_blob *queue;    /* struct pending */
	uint32_t *next_dnode;
	int is_root;
	uint32_t child_bnodes[256];
	uint32_t child_chunk[256];
	uint32_t nchild;
};

static void
root_cb(void *vctx, uint32_t first, uint32_t count, int kind, uint32_t value)
{
	struct fl_ctx *c = vctx;
	uint32_t i;

	if (kind == 0) {
		for (i = 0; i < count; i++) c->entries[first + i] = XE_TRIE_TERM | value;
	} else {
		struct pending p;
		p.bnode = value;
		p.dnode = (*c->next_dnode)++;
		c->entries[first] = p.dnode;
		xe_blob_push(c->queue, &p, sizeof(p));
	}
}

static void
node_cb(void *vctx, uint32_t first, uint32_t count, int kind, uint32_t value)
{
	struct fl_ctx *c = vctx;
	uint32_t i;

	if (kind == 0) {
		if (value) {
			for (i = first; i < first + count; i++) c->node[4 * (i >> 5) + 1] |= 1u << (i & 31);
		}
	} else {
		c->node[4 * (first >> 5)] |= 1u << (first & 31);
		c->child_bnodes[c->nchild] = value;
		c->child_chunk[c->nchild] = first;
		c->nchild++;
	}
}

/* one family of one list; returns 0 on allocation failure */
static int
flatten_family(const struct xe_bnode *bt, struct xe_blob *words, uint32_t *root_off, uint32_t *nodes_off)
{
	struct xe_blob queue = {0};
	struct fl_ctx c;
	uint32_t next_dnode = 0;
	uint32_t *root;
	size_t qi, i;
	size_t root_pos, nodes_pos;
	uint32_t zero = 0;

	*root_off = 0;
	*nodes_off = 0;
	if (!bt) return 1;

	/* word 0 of the array is reserved so that offset 0 means "empty" */
	if (words->n == 0) {
		if (!xe_blob_push(words, &zero, 4)) return 0;
	}
	root_pos = words->n;
	for (i = 0; i < 65536; i++) {
		if (!xe_blob_push(words, &zero, 4)) return 0;
	}
	memset(&c, 0, sizeof(c));
	c.queue = &queue;
	c.next_dnode = &next_dnode;
	c.is_root = 1;
	root = (uint32_t *)words->p + root_pos;
	c.entries = root;
	expand(bt, 0, 16, 0, root_cb, &c);

	/* breadth-first so that the children of a node get consecutive ids;
	 * node k of the family lives at nodes_pos + k * XE_NODE_WORDS, and ids
	 * are handed out in queue order, so the queue index IS the node id */
	while (words->n % XE_NODE_WORDS) {
		if (!xe_blob_push(words, &zero, 4)) { free(queue.p); return 0; }
	}
	nodes_pos = words->n;
	for (qi = 0; qi < queue.n; qi++) {
		struct pending p = ((struct pending *)queue.p)[qi];
		uint32_t nw[XE_NODE_WORDS];
		uint32_t k, cum = 0;

		memset(nw, 0, sizeof(nw));
		c.node = nw;
		c.nchild = 0;
		expand(bt, p.bnode, 8, 0, node_cb, &c);
		for (k = 0; k < 8; k++) {
			nw[4 * k + 2] = next_dnode + cum;   /* first child of this group */
			cum += (uint32_t)__builtin_popcount(nw[4 * k]);
		}
		/* children in increasing chunk order = rank order of the bitmap */
		for (k = 0; k < c.nchild; k++) {
			struct pending ch;
			ch.bnode = c.child_bnodes[k];
			ch.dnode = next_dnode++;
			if (!xe_blob_push(&queue, &ch, sizeof(ch))) { free(queue.p); return 0; }
		}
		if (p.dnode != qi) { free(queue.p); return 0; }
		for (k = 0; k < XE_NODE_WORDS; k++) {
			if (!xe_blob_push(words, &nw[k], 4)) { free(queue.p); return 0; }
		}
	}
	free(queue.p);
	if (words->n > 0xFFFFFFF0u) return 0;
	*root_off = (uint32_t)root_pos;
	*nodes_off = (uint32_t)nodes_pos;
	return 1;
}

int
xe_iplists_flatten(const struct xe_iplists *reg, struct xe_blob *words, struct xe_list_dev *dev)
{
	size_t i;
	for (i = 0; i < reg->n; i++) {
		if (!flatten_family(reg->l[i].n4, words, &dev[i].root4, &dev[i].nodes4)) return XE_ENOMEM;
		if (!flatten_family(reg->l[i].n6, words, &dev[i].root6, &dev[i].nodes6)) return XE_ENOMEM;
	}
	return XE_OK;
}
