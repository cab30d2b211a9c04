/*
 * xe_iplist.c -- IP-list files -> bitwise trie -> flattened device arrays.
 *
 * Behaviour follows the reference's iplist.c: the text loader of :218-297
 * (including: '#' starts a comment anywhere, the default prefix length is 32
 * for IPv6 lines too, a bad length skips the line, a last line without '\n'
 * is dropped) and the lookup of :160-216, which returns the leaf flag of the
 * DEEPEST trie node the address reaches -- not a longest-prefix match.
 *
 * The device form replaces the one-bit-per-step pointer chase of up to 8
 * lists at once with a 65536-entry root table over the first 16 address bits
 * followed by 8-bit-stride nodes of 256 directly indexed entries (terminal
 * with one answer bit per list, or the index of the next node), so a lookup
 * is one 4-byte load per level, at most 1 + (bits-16)/8 of them, whatever the
 * number of lists; tails along which only one prefix of one list is left
 * are collapsed into one comparison (chain records, or the entry itself when
 * the tail is short).
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
 * Lists are flattened in groups of XE_LIST_GROUP (registry order): the tries
 * of a group are walked together, so that ONE device walk per address answers
 * all lists of the group.  Per list the answer must stay the reference's: the
 * leaf flag of the deepest node of THAT list's trie on the address's path.
 *
 * gstate = where each list of the group stands after some address bits:
 * still walking (at binary node cur[l]) or stopped with a final answer.
 */
struct gstate
{
	uint32_t cur[XE_LIST_GROUP];
	uint8_t walking;     /* bit l: list l has not stopped yet */
	uint8_t res;         /* bit l: answer of a stopped list */
};

/* cb(ctx, first, count, res, child): `count` consecutive chunk values starting
 * at `first` end with answers `res`; child != NULL (count == 1): some list can
 * go deeper, its state is *child and the final answers come from below */
typedef int (*gexpand_cb)(void *ctx, uint32_t first, uint32_t count, uint8_t res, const struct gstate *child);

static int
gexpand(const struct xe_bnode *const *bt, const struct gstate *st, int depth, uint32_t prefix,
	gexpand_cb cb, void *ctx)
{
	int b, l;

	if (depth == 0) {
		struct gstate ch = *st;
		ch.walking = 0;
		for (l = 0; l < XE_LIST_GROUP; l++) {
			const struct xe_bnode *n;
			if (!((st->walking >> l) & 1)) continue;
			n = &bt[l][st->cur[l]];
			if (n->next[0] || n->next[1]) {
				ch.walking |= (uint8_t)(1u << l);
			} else if (n->is_leaf) {
				ch.res |= (uint8_t)(1u << l);
			}
		}
		return cb(ctx, prefix, 1, ch.res, ch.walking ? &ch : NULL);
	}
	for (b = 0; b < 2; b++) {
		struct gstate ns = *st;
		uint32_t np = (prefix << 1) | (uint32_t)b;
		for (l = 0; l < XE_LIST_GROUP; l++) {
			const struct xe_bnode *n;
			if (!((st->walking >> l) & 1)) continue;
			n = &bt[l][st->cur[l]];
			if (n->next[b]) {
				ns.cur[l] = n->next[b];
			} else {
				/* the list's walk stops at n */
				ns.walking &= (uint8_t)~(1u << l);
				if (n->is_leaf) ns.res |= (uint8_t)(1u << l);
			}
		}
		if (!ns.walking) {
			if (!cb(ctx, np << (depth - 1), 1u << (depth - 1), ns.res, NULL)) return 0;
		} else if (!gexpand(bt, &ns, depth - 1, np, cb, ctx)) {
			return 0;
		}
	}
	return 1;
}

struct gpending { struct gstate st; };

struct gfl_ctx
{
	const struct xe_bnode *const *bt;
	uint32_t *root;           /* 2^root_bits entries while the root is expanded */
	uint32_t *node;           /* XE_NODE_WORDS words while a node is expanded */
	struct xe_blob *queue;    /* struct gpending; queue index = device node index */
	struct xe_blob *chains;   /* XE_CHAIN_WORDS words per record */
};

/*
 * Where the walk can only go on in ONE list, along a single path that ends in
 * a prefix with nothing below and no prefix on the way, the rest of the walk
 * is one comparison: the address follows the path to its end (answer 1 for
 * that list: the deepest node is the prefix's) or leaves it at an inner node
 * (answer 0); when the node the tail starts at is itself a prefix, an address
 * that leaves the path with its first bit stays at that prefix (answer 1).
 * Such a tail becomes a chain record instead of a node per address byte.  Returns 1 and fills rec[] when `st` is such a state, 0 when
 * it is not, and 2 when the path ends without a prefix (answer 0 always).
 */
static int
chain_of(const struct xe_bnode *const *bt, const struct gstate *st, uint32_t rec[XE_CHAIN_WORDS])
{
	int l = -1, k;
	uint32_t n, len = 0, start_leaf = 0;
	uint64_t pat[2] = { 0, 0 };

	for (k = 0; k < XE_LIST_GROUP; k++) {
		if ((st->walking >> k) & 1) {
			if (l >= 0) return 0;
			l = k;
		}
	}
	if (l < 0) return 0;
	n = st->cur[l];
	for (;;) {
		const struct xe_bnode *b = &bt[l][n];
		int c0 = b->next[0] != 0, c1 = b->next[1] != 0;
		if (!c0 && !c1) break;
		if ((c0 && c1) || (b->is_leaf && len) || (len >= 112)) return 0;
		if (b->is_leaf) start_leaf = 1;
		if (c1) pat[len >> 6] |= 1ull << (63 - (len & 63));
		n = c1 ? b->next[1] : b->next[0];
		len++;
	}
	memset(rec, 0, XE_CHAIN_WORDS * sizeof(uint32_t));
	if (!bt[l][n].is_leaf) {
		/* (a path that ends without a prefix comes from a mask beyond the
		 * address width; with a prefix at its start it is left to a node) */
		if (start_leaf) return 0;
		rec[0] = st->res & (uint8_t)~(1u << l);
		return 2;
	}
	rec[0] = (uint32_t)(st->res & (uint8_t)~(1u << l)) | ((uint32_t)l << 8) | (len << 11) | (start_leaf << 18);
	rec[2] = (uint32_t)pat[0];
	rec[3] = (uint32_t)(pat[0] >> 32);
	rec[4] = (uint32_t)pat[1];
	rec[5] = (uint32_t)(pat[1] >> 32);
	return 1;
}

/* entry for a state some list can leave deeper: chain record, plain answer or
 * a node of its own */
static int
child_entry(struct gfl_ctx *c, const struct gstate *child, uint32_t *entry)
{
	uint32_t rec[XE_CHAIN_WORDS];
	int kind = chain_of(c->bt, child, rec);
	size_t k;

	if (kind == 2) {
		*entry = XE_TRIE_TERM | (rec[0] & 0xFF);
		return 1;
	}
	if (kind == 1) {
		const uint32_t len = (rec[0] >> 11) & 127, l = (rec[0] >> 8) & 7, base = rec[0] & 0xFF;
		const uint32_t sl = (rec[0] >> 18) & 1;
		size_t idx;

		/* a path of up to 16 bits fits into the entry */
		if (len && (len <= 13)) {
			const uint32_t path = rec[3] >> (32 - len);
			*entry = XE_TRIE_INLINE | base | (l << 8) | (len << 11) | (path << 15) | (sl << 28);
			return 1;
		}
		if (len && (len <= 16) && !base) {
			const uint32_t path = rec[3] >> (32 - len);
			*entry = XE_TRIE_INLINE | XE_TRIE_INLINE_LONG | (len << 20) | (sl << 19) | (l << 16) | path;
			return 1;
		}
		idx = c->chains->n / XE_CHAIN_WORDS;
		if (idx >= 0x3FFFFFF0u) return 0;
		for (k = 0; k < XE_CHAIN_WORDS; k++) {
			if (!xe_blob_push(c->chains, &rec[k], 4)) return 0;
		}
		*entry = XE_TRIE_CHAIN | (uint32_t)idx;
		return 1;
	}
	if (c->queue->n >= 0x3FFFFFF0u) return 0;
	*entry = (uint32_t)c->queue->n;
	return xe_blob_push(c->queue, child, sizeof(struct gpending));
}

static int
groot_cb(void *vctx, uint32_t first, uint32_t count, uint8_t res, const struct gstate *child)
{
	struct gfl_ctx *c = vctx;
	uint32_t i;

	if (!child) {
		for (i = 0; i < count; i++) c->root[first + i] = XE_TRIE_TERM | res;
		return 1;
	}
	return child_entry(c, child, &c->root[first]);
}

static int
gnode_cb(void *vctx, uint32_t first, uint32_t count, uint8_t res, const struct gstate *child)
{
	struct gfl_ctx *c = vctx;
	uint32_t i;

	if (!child) {
		for (i = first; i < first + count; i++) c->node[i] = XE_TRIE_TERM | res;
		return 1;
	}
	return child_entry(c, child, &c->node[first]);
}

/* n zero words at the end of b */
static int
blob_zeros(struct xe_blob *b, size_t n)
{
	const size_t want = (b->n + n) * 4;
	if (want > b->cap) {
		size_t nc = b->cap ? b->cap : 4096;
		void *np;
		while (nc < want) nc *= 2;
		np = realloc(b->p, nc);
		if (!np) return 0;
		b->p = np;
		b->cap = nc;
	}
	memset((uint8_t *)b->p + b->n * 4, 0, n * 4);
	b->n += n;
	return 1;
}

/* one address family of one group of lists; bt[l] == NULL: list l has no
 * entries of this family (answers 0).  want_bits: the root's width, 24 or
 * (anything else) 16.  Returns 0 on allocation failure. */
static int
flatten_group(const struct xe_bnode *const *bt, struct xe_blob *words, int want_bits,
	uint32_t *root_off, uint32_t *nodes_off, uint32_t *chains_off, uint32_t *root_bits)
{
	struct xe_blob queue = {0}, chains = {0};
	struct gfl_ctx c;
	struct gstate st0;
	size_t qi, i, root_pos, nodes_pos, chains_pos;
	uint32_t zero = 0;
	int l, any = 0, ok = 0, rb = (want_bits == 24) ? 24 : 16;

	*root_off = 0;
	*nodes_off = 0;
	*chains_off = 0;
	*root_bits = 16;
	memset(&st0, 0, sizeof(st0));
	for (l = 0; l < XE_LIST_GROUP; l++) {
		if (bt[l]) { st0.walking |= (uint8_t)(1u << l); any = 1; }
	}
	if (!any) return 1;

	/* word 0 of the array is reserved so that offset 0 means "empty" */
	if (words->n == 0) {
		if (!xe_blob_push(words, &zero, 4)) return 0;
	}
	root_pos = words->n;
	c.bt = bt;
	c.queue = &queue;
	c.chains = &chains;
	c.node = NULL;
	if (!blob_zeros(words, (size_t)1 << rb)) goto out;
	/* the root lives in `words`, which does not grow while it is expanded */
	c.root = (uint32_t *)words->p + root_pos;
	if (!gexpand(bt, &st0, rb, 0, groot_cb, &c)) goto out;
	while (words->n % 16) {
		if (!xe_blob_push(words, &zero, 4)) goto out;
	}
	nodes_pos = words->n;
	for (qi = 0; qi < queue.n; qi++) {
		struct gstate st = ((struct gpending *)queue.p)[qi].st;
		uint32_t nw[XE_NODE_WORDS];
		uint32_t k;

		c.node = nw;
		if (!gexpand(bt, &st, XE_NODE_BITS, 0, gnode_cb, &c)) goto out;
		for (k = 0; k < XE_NODE_WORDS; k++) {
			if (!xe_blob_push(words, &nw[k], 4)) goto out;
		}
	}
	while (words->n % XE_CHAIN_WORDS) {
		if (!xe_blob_push(words, &zero, 4)) goto out;
	}
	chains_pos = words->n;
	for (i = 0; i < chains.n; i++) {
		if (!xe_blob_push(words, (uint32_t *)chains.p + i, 4)) goto out;
	}
	/* one spare record so that a wide load of the last one stays inside */
	for (i = 0; i < XE_CHAIN_WORDS; i++) {
		if (!xe_blob_push(words, &zero, 4)) goto out;
	}
	if (words->n > 0xFFFFFFF0u) goto out;
	*root_off = (uint32_t)root_pos;
	*nodes_off = (uint32_t)nodes_pos;
	*chains_off = (uint32_t)chains_pos;
	*root_bits = (uint32_t)rb;
	ok = 1;
out:
	free(queue.p);
	free(chains.p);
	return ok;
}

int
xe_iplists_flatten(const struct xe_iplists *reg, struct xe_blob *words, struct xe_list_dev *dev, int root_bits)
{
	size_t g, l;

	for (g = 0; g * XE_LIST_GROUP < reg->n; g++) {
		const struct xe_bnode *b4[XE_LIST_GROUP], *b6[XE_LIST_GROUP];
		uint32_t r4, n4, c4, r6, n6, c6, b4bits, b6bits;

		for (l = 0; l < XE_LIST_GROUP; l++) {
			size_t li = g * XE_LIST_GROUP + l;
			b4[l] = (li < reg->n) ? reg->l[li].n4 : NULL;
			b6[l] = (li < reg->n) ? reg->l[li].n6 : NULL;
		}
		if (!flatten_group(b4, words, root_bits, &r4, &n4, &c4, &b4bits)) return XE_ENOMEM;
		if (!flatten_group(b6, words, root_bits, &r6, &n6, &c6, &b6bits)) return XE_ENOMEM;
		for (l = 0; (l < XE_LIST_GROUP) && (g * XE_LIST_GROUP + l < reg->n); l++) {
			struct xe_list_dev *d = &dev[g * XE_LIST_GROUP + l];
			memset(d, 0, sizeof(*d));
			d->root4 = r4; d->nodes4 = n4; d->chains4 = c4;
			d->root6 = r6; d->nodes6 = n6; d->chains6 = c6;
			d->bit = (uint32_t)l;
			d->rbits = b4bits | (b6bits << 8);
		}
	}
	return XE_OK;
}
