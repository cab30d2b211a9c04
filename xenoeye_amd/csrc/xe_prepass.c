/*
 * xe_prepass.c -- host walk over a batch of export datagrams.
 *
 * Templates arrive in-band and the reference's set loops carry a cursor from
 * one set to the next, so the order-dependent part of the path is done here,
 * sequentially per datagram, reading only headers: Netflow v9 / IPFIX / v5
 * datagram headers, set headers and template sets (netflow.c:319-365,
 * :628-678, :727-746), sFlow datagram / sample / element headers
 * (sflow-impl.h:68-405).  The output is a list of runs ("sets") of records
 * that share template, exporter and spacing; record bytes themselves are only
 * touched on the GPU (the one exception: IPFIX records with variable-length
 * elements, whose length prefixes must be read to find the next record).
 *
 * The set arithmetic reproduces the reference exactly, including:
 *   - the set length passed down includes the 4-byte set header although the
 *     cursor already passed it (netflow.c:342-358, :651-666);
 *   - a v9 data set is iterated header.count times, whatever it holds
 *     (netflow.c:272), with the field cursor carried from record to record;
 *   - after a v9 data set the cursor advances by the full length, i.e. 4
 *     bytes too far (netflow.c:315); after a v9 template set only by the size
 *     of the first template (netflow.c:173);
 *   - bytes past the end of a datagram read as zero (the reference re-uses a
 *     64 KiB buffer; zero is what a fresh buffer holds).
 */
#include <stdlib.h>
#include <stdio.h>
#include <time.h>
#include <string.h>
#include <arpa/inet.h>
#include "xe_internal.h"

/* ------------------------------------------------------------------ helpers */
struct dgram
{
	const uint8_t *p;     /* first byte */
	long len;
	uint64_t base;        /* offset of p in the batch buffer */
	uint32_t src_ip;      /* bytes as stored */
	uint32_t source_id;   /* raw */
	uint32_t exporter;    /* index in plan->exporters */
};

static inline uint8_t
g8(const struct dgram *d, long off)
{
	return ((off >= 0) && (off < d->len)) ? d->p[off] : 0;
}

static inline uint32_t
g16(const struct dgram *d, long off)
{
	return ((uint32_t)g8(d, off) << 8) | g8(d, off + 1);
}

static inline uint32_t
g32raw(const struct dgram *d, long off)   /* 4 bytes in memory order */
{
	uint8_t b[4];
	uint32_t v;
	b[0] = g8(d, off); b[1] = g8(d, off + 1); b[2] = g8(d, off + 2); b[3] = g8(d, off + 3);
	memcpy(&v, b, 4);
	return v;
}

static void
gcopy(const struct dgram *d, long off, uint8_t *dst, long n)
{
	long i;
	for (i = 0; i < n; i++) dst[i] = g8(d, off + i);
}

void
xe_host_init(struct xe_host *hs)
{
	pthread_rwlockattr_t a;

	memset(hs, 0, sizeof(*hs));
	pthread_rwlockattr_init(&a);
	/* a waiting writer (a new template) goes before readers that come back */
	pthread_rwlockattr_setkind_np(&a, PTHREAD_RWLOCK_PREFER_WRITER_NONRECURSIVE_NP);
	pthread_rwlock_init(&hs->tlock, &a);
	pthread_rwlockattr_destroy(&a);
	pthread_mutex_init(&hs->pool_mu, NULL);
}

/* a buffer of at least `bytes` from the host's pool of kept buffers */
static void *
pool_get(struct xe_host *hs, size_t bytes, size_t *cap)
{
	int i, best = -1;
	void *p = NULL;

	if (!bytes) bytes = 1;
	pthread_mutex_lock(&hs->pool_mu);
	for (i = 0; i < 8; i++) {
		if (hs->pool_p[i] && (hs->pool_cap[i] >= bytes) && ((best < 0) || (hs->pool_cap[i] < hs->pool_cap[best]))) best = i;
	}
	if ((best >= 0) && (hs->pool_cap[best] <= 4 * bytes + (1u << 20))) {
		p = hs->pool_p[best];
		*cap = hs->pool_cap[best];
		hs->pool_p[best] = NULL;
	}
	pthread_mutex_unlock(&hs->pool_mu);
	if (p) return p;
	*cap = bytes + bytes / 4;
	return hs->big_alloc ? hs->big_alloc(*cap) : malloc(*cap);
}

static void
pool_release(struct xe_host *hs, void *p)
{
	if (!p) return;
	if (hs->big_free) hs->big_free(p);
	else free(p);
}

static void
pool_put(struct xe_host *hs, void *p, size_t cap)
{
	int i, slot = -1;

	if (!p) return;
	pthread_mutex_lock(&hs->pool_mu);
	for (i = 0; i < 8; i++) {
		if (!hs->pool_p[i]) { slot = i; break; }
	}
	if (slot < 0) {
		/* full: the smallest kept buffer makes room if this one is bigger */
		int small = 0;
		for (i = 1; i < 8; i++) if (hs->pool_cap[i] < hs->pool_cap[small]) small = i;
		if (hs->pool_cap[small] < cap) {
			void *old = hs->pool_p[small];
			hs->pool_p[small] = p;
			hs->pool_cap[small] = cap;
			p = old;
		}
	} else {
		hs->pool_p[slot] = p;
		hs->pool_cap[slot] = cap;
		p = NULL;
	}
	pthread_mutex_unlock(&hs->pool_mu);
	pool_release(hs, p);
}

static void walkers_free(struct xe_host *hs);

/* Template store access while several threads walk datagrams: every walking
 * thread holds the read side all the time (it lets go for a moment every few
 * hundred datagrams), and trades it for the write side while it stores a new
 * or changed template.  Nothing to do when one thread walks alone. */
static inline void
tl_write(struct xe_host *hs)
{
	if (!hs->mt) return;
	pthread_rwlock_unlock(&hs->tlock);
	pthread_rwlock_wrlock(&hs->tlock);
}

static inline void
tl_write_done(struct xe_host *hs)
{
	if (!hs->mt) return;
	pthread_rwlock_unlock(&hs->tlock);
	pthread_rwlock_rdlock(&hs->tlock);
}

void
xe_host_free(struct xe_host *hs)
{
	size_t i, j;
	for (i = 0; i < hs->nmos; i++) xe_filter_free(&hs->mos[i].filter);
	free(hs->mos);
	for (i = 0; i < hs->ndevs; i++) {
		for (j = 0; j < hs->devs[i].nmarks; j++) xe_filter_free(&hs->devs[i].marks[j]);
		free(hs->devs[i].marks);
	}
	free(hs->devs);
	xe_iplists_free(&hs->lists);
	for (i = 0; i < hs->ntslots; i++) free(hs->tslots[i].raw);
	free(hs->tslots);
	free(hs->thash);
	xe_host_undo_end(hs);
	free(hs->undo.p);
	for (i = 0; i < 4; i++) free(hs->geodb[i >> 1][i & 1].img);
	free(hs->dev_tmpls.p);
	free(hs->dev_tents.p);
	walkers_free(hs);
	for (i = 0; i < 8; i++) pool_release(hs, hs->pool_p[i]);
	pthread_mutex_destroy(&hs->pool_mu);
	pthread_rwlock_destroy(&hs->tlock);
	memset(hs, 0, sizeof(*hs));
}

void
xe_plan_free(struct xe_plan *p)
{
	if (p->pool) {
		pool_put(p->pool, p->sets.p, p->sets.cap);
		pool_put(p->pool, p->sfrecs.p, p->sfrecs.cap);
	} else {
		free(p->sets.p);
		free(p->sfrecs.p);
	}
	free(p->exporters.p);
	free(p->ehash);
	free(p->tinfo.p);
	free(p->dg_first_nf);
	memset(p, 0, sizeof(*p));
}

/* ------------------------------------------------------------------ devices.conf lookup */
/* devices.c:205-232 / :246-270; the second test is `if`, not `else if` */
static int
device_find(const struct xe_host *hs, uint32_t ip, uint32_t id)
{
	size_t i;
	for (i = 0; i < hs->ndevs; i++) {
		const struct xe_device *d = &hs->devs[i];
		if (d->use_ip && d->use_id) {
			if ((ip == d->ip) && (id == d->id)) return (int)i;
		} else if (d->use_ip) {
			if (ip == d->ip) return (int)i;
		}
		if (d->use_id) {
			if (id == d->id) return (int)i;
		}
	}
	return -1;
}

/* exporter = (source address, source id) pair seen in this batch; carries the
 * per-packet values of netflow.c:112-144 */
static int
exporter_get(struct xe_host *hs, struct xe_plan *plan, uint32_t ip, uint32_t id, uint32_t *out)
{
	struct xe_exporter *ex;
	struct xe_exporter ne;
	size_t i, m;
	uint64_t hv = ((uint64_t)ip << 32 | id) * 0x9E3779B97F4A7C15ull;
	int dev;

	if ((plan->exporters.n + 1) * 2 > plan->ehash_cap) {
		size_t ncap = plan->ehash_cap ? plan->ehash_cap * 2 : 256, j;
		uint32_t *nh = calloc(ncap, sizeof(uint32_t));
		if (!nh) return 0;
		ex = plan->exporters.p;
		for (j = 0; j < plan->exporters.n; j++) {
			uint32_t eip, eid;
			memcpy(&eip, ex[j].dev_ip, 4);
			memcpy(&eid, ex[j].dev_id, 4);
			for (i = ((((uint64_t)eip << 32 | eid) * 0x9E3779B97F4A7C15ull) >> 32) & (ncap - 1); nh[i]; i = (i + 1) & (ncap - 1)) {}
			nh[i] = (uint32_t)j + 1;
		}
		free(plan->ehash);
		plan->ehash = nh;
		plan->ehash_cap = ncap;
	}
	ex = plan->exporters.p;
	m = plan->ehash_cap - 1;
	for (i = (hv >> 32) & m; plan->ehash[i]; i = (i + 1) & m) {
		const struct xe_exporter *e = &ex[plan->ehash[i] - 1];
		if (!memcmp(e->dev_ip, &ip, 4) && !memcmp(e->dev_id, &id, 4)) {
			*out = plan->ehash[i] - 1;
			return 1;
		}
	}
	memset(&ne, 0, sizeof(ne));
	memcpy(ne.dev_ip, &ip, 4);
	memcpy(ne.dev_id, &id, 4);
	dev = device_find(hs, ip, id);
	ne.device = dev;
	ne.sampling_rate = (dev >= 0) ? (uint32_t)hs->devs[dev].sampling_rate : 1;
	*out = (uint32_t)plan->exporters.n;
	plan->ehash[i] = (uint32_t)plan->exporters.n + 1;
	return xe_blob_push(&plan->exporters, &ne, sizeof(ne));
}

/* ------------------------------------------------------------------ template store */
static uint64_t
tkey_hash(const struct xe_tkey *k)
{
	uint64_t h = 0xcbf29ce484222325ull;
	h = (h ^ k->ver) * 0x100000001b3ull;
	h = (h ^ k->tid) * 0x100000001b3ull;
	h = (h ^ k->src_ip) * 0x100000001b3ull;
	h = (h ^ k->source_id) * 0x100000001b3ull;
	return h ^ (h >> 29);
}

static int
tkey_eq(const struct xe_tkey *a, const struct xe_tkey *b)
{
	return (a->ver == b->ver) && (a->tid == b->tid) && (a->src_ip == b->src_ip)
		&& (a->source_id == b->source_id);
}

/* netflow_template_find() (netflow-templates.c:140-181) reduced to "latest
 * template for (version, template id as on the wire, exporter, source id)" */
static struct xe_tslot *
tmpl_find_any(struct xe_host *hs, const struct xe_tkey *k)
{
	size_t m, i;
	if (!hs->thash_cap) return NULL;
	m = hs->thash_cap - 1;
	for (i = tkey_hash(k) & m;; i = (i + 1) & m) {
		uint32_t v = hs->thash[i];
		if (!v) return NULL;
		if (tkey_eq(&hs->tslots[v - 1].key, k)) return &hs->tslots[v - 1];
	}
}

/* (a slot without bytes is one whose first template was taken back by
 * xe_host_undo_rollback(): it keeps its place in the hash, nothing else) */
static struct xe_tslot *
tmpl_find(struct xe_host *hs, const struct xe_tkey *k)
{
	struct xe_tslot *s = tmpl_find_any(hs, k);
	return (s && s->raw) ? s : NULL;
}

static int
thash_insert(struct xe_host *hs, uint32_t slot)
{
	size_t m, i;
	if ((hs->ntslots + 1) * 2 > hs->thash_cap) {
		size_t ncap = hs->thash_cap ? hs->thash_cap * 2 : 1024, j;
		uint32_t *nh = calloc(ncap, sizeof(uint32_t));
		if (!nh) return 0;
		for (j = 0; j < hs->ntslots; j++) {
			if (j == slot) continue;
			for (i = tkey_hash(&hs->tslots[j].key) & (ncap - 1); nh[i]; i = (i + 1) & (ncap - 1)) {}
			nh[i] = (uint32_t)j + 1;
		}
		free(hs->thash);
		hs->thash = nh;
		hs->thash_cap = ncap;
	}
	m = hs->thash_cap - 1;
	for (i = tkey_hash(&hs->tslots[slot].key) & m; hs->thash[i]; i = (i + 1) & m) {}
	hs->thash[i] = slot + 1;
	return 1;
}

/* build (or re-use) the device form of a template.  pairs: n x {type, len}. */
static int
dev_template(struct xe_host *hs, int mode, const uint16_t *types, const uint16_t *lens, size_t n,
	uint32_t *slot_out)
{
	struct xe_tmpl t;
	struct xe_tent *ents;
	struct xe_tmpl *existing = hs->dev_tmpls.p;
	size_t i, j, keep = n;
	uint64_t pos = 0;
	int varlen = 0;

	/* entries that start past 65535 can never be reached: every set length
	 * is a 16-bit number and both record loops stop at it */
	for (i = 0; i < n; i++) {
		if (pos > 0xFFFF) { keep = i; break; }
		if ((mode == XE_MODE_IPFIX) && (lens[i] == 0xFFFF)) varlen = 1;
		else pos += lens[i];
	}
	if (keep > XE_TMPL_MAXF) return XE_EUNSUPPORTED;

	ents = calloc(keep ? keep : 1, sizeof(*ents));
	if (!ents) return XE_ENOMEM;
	memset(&t, 0, sizeof(t));
	memset(t.fmap, 0xFF, sizeof(t.fmap));
	t.mode = (uint8_t)mode;
	t.nfields = (uint16_t)keep;
	t.varlen = (uint8_t)varlen;
	pos = 0;
	for (i = 0; i < keep; i++) {
		int f = xe_field_by_id(types[i]);
		struct xe_tent *e = &ents[i];

		e->len = lens[i];
		e->off = (uint16_t)((pos > 0xFFFF) ? 0xFFFF : pos);
		e->field = (f < 0) ? XE_F_NONE : (uint8_t)f;
		e->next_same = 0xFF;
		if (f >= 0) {
			int l = lens[i];
			/* a variable-length element is validated per record */
			if ((mode == XE_MODE_IPFIX) && (lens[i] == 0xFFFF)) e->flags |= XE_TE_VALID;
			else if ((l >= xe_fields[f].size_min) && (l <= xe_fields[f].size_max)) e->flags |= XE_TE_VALID;
			if (xe_fields[f].type == XE_T_STR) e->flags |= XE_TE_STR;
			if (t.fmap[f] == 0xFF) {
				t.fmap[f] = (uint8_t)i;
			} else {
				for (j = t.fmap[f]; ents[j].next_same != 0xFF; j = ents[j].next_same) {}
				ents[j].next_same = (uint8_t)i;
				e->flags |= XE_TE_REPEAT;
				t.repeats = 1;
			}
		}
		if (!((mode == XE_MODE_IPFIX) && (lens[i] == 0xFFFF))) pos += lens[i];
	}
	t.stride = (uint16_t)((pos > 0xFFFF) ? 0xFFFF : pos);
	if (pos > 0xFFFF) t.varlen = 2;   /* records longer than any set: walked one by one */

	/* identical templates (typically the same router model behind many
	 * exporters) share one device slot */
	for (i = 0; i < hs->dev_tmpls.n; i++) {
		const struct xe_tmpl *o = &existing[i];
		if ((o->mode == t.mode) && (o->nfields == t.nfields) && (o->stride == t.stride)
			&& (o->varlen == t.varlen)
			&& !memcmp((struct xe_tent *)hs->dev_tents.p + o->ent_base, ents, keep * sizeof(*ents))) {
			free(ents);
			*slot_out = (uint32_t)i;
			return XE_OK;
		}
	}
	t.ent_base = (uint32_t)hs->dev_tents.n;
	for (i = 0; i < keep; i++) {
		if (!xe_blob_push(&hs->dev_tents, &ents[i], sizeof(*ents))) { free(ents); return XE_ENOMEM; }
	}
	free(ents);
	*slot_out = (uint32_t)hs->dev_tmpls.n;
	if (!xe_blob_push(&hs->dev_tmpls, &t, sizeof(t))) return XE_ENOMEM;
	return XE_OK;
}

/* netflow_template_add() (netflow-templates.c:183-251) without persistence:
 * raw = the bytes the reference stores */
static int
tmpl_put(struct xe_host *hs, const struct xe_tkey *k, const uint8_t *raw, size_t raw_len)
{
	struct xe_tslot *s = tmpl_find_any(hs, k);
	uint16_t *types, *lens;
	size_t n, i, step, first;
	int rc, mode;
	uint32_t slot;

	if (s && s->raw && (s->raw_len == raw_len) && !memcmp(s->raw, raw, raw_len)) return XE_OK;

	n = ((size_t)raw[2] << 8) | raw[3];
	if (k->ver == 9) { mode = XE_MODE_V9; step = 4; first = 4; }
	else { mode = XE_MODE_IPFIX; step = 8; first = 4; }
	types = malloc((n ? n : 1) * 2 * sizeof(uint16_t));
	if (!types) return XE_ENOMEM;
	lens = types + (n ? n : 1);
	for (i = 0; i < n; i++) {
		const uint8_t *e = raw + first + step * i;
		types[i] = (uint16_t)((e[0] << 8) | e[1]);
		lens[i] = (uint16_t)((e[2] << 8) | e[3]);
	}
	rc = dev_template(hs, mode, types, lens, n, &slot);
	free(types);
	/* more elements than a device template holds: the template is remembered
	 * as such and its data sets are left out (and counted) */
	if (rc == XE_EUNSUPPORTED) { slot = XE_TMPL_UNSUPPORTED; rc = XE_OK; }
	if (rc != XE_OK) return rc;

	if (!s) {
		if (hs->ntslots == hs->tslots_cap) {
			size_t nc = hs->tslots_cap ? hs->tslots_cap * 2 : 256;
			struct xe_tslot *nt = realloc(hs->tslots, nc * sizeof(*nt));
			if (!nt) return XE_ENOMEM;
			hs->tslots = nt;
			hs->tslots_cap = nc;
		}
		s = &hs->tslots[hs->ntslots];
		memset(s, 0, sizeof(*s));
		s->key = *k;
		if (!thash_insert(hs, (uint32_t)hs->ntslots)) return XE_ENOMEM;
		hs->ntslots++;
	}
	if (hs->undo_on) {
		/* what the slot held goes to the log instead of being freed */
		struct xe_tundo u;
		u.slot = (uint32_t)(s - hs->tslots);
		u.raw = s->raw;
		u.raw_len = s->raw_len;
		u.dev_slot = s->dev_slot;
		if (!xe_blob_push(&hs->undo, &u, sizeof(u))) return XE_ENOMEM;
		s->raw = NULL;
	}
	free(s->raw);
	s->raw = malloc(raw_len ? raw_len : 1);
	if (!s->raw) return XE_ENOMEM;
	memcpy(s->raw, raw, raw_len);
	s->raw_len = raw_len;
	s->dev_slot = slot;
	hs->tmpl_epoch++;
	return XE_OK;
}

/* Template changes of one batch can be taken back: between undo_begin and
 * undo_end every change is logged with what it replaced. */
void
xe_host_undo_begin(struct xe_host *hs)
{
	hs->undo.n = 0;
	hs->undo_on = 1;
}

size_t
xe_host_undo_count(const struct xe_host *hs)
{
	return hs->undo.n;
}

uint32_t
xe_host_undo_src_ip(const struct xe_host *hs, size_t i)
{
	return hs->tslots[((const struct xe_tundo *)hs->undo.p)[i].slot].key.src_ip;
}

/* the store as it was at undo_begin (device templates made meanwhile stay in
 * their table, unused) */
void
xe_host_undo_rollback(struct xe_host *hs)
{
	struct xe_tundo *u = hs->undo.p;
	size_t i;

	for (i = hs->undo.n; i-- > 0;) {
		struct xe_tslot *s = &hs->tslots[u[i].slot];
		free(s->raw);
		s->raw = u[i].raw;
		s->raw_len = u[i].raw_len;
		s->dev_slot = u[i].dev_slot;
	}
	if (hs->undo.n) hs->tmpl_epoch++;
	hs->undo.n = 0;
}

void
xe_host_undo_end(struct xe_host *hs)
{
	struct xe_tundo *u = hs->undo.p;
	size_t i;

	for (i = 0; i < hs->undo.n; i++) free(u[i].raw);
	hs->undo.n = 0;
	hs->undo_on = 0;
}

/* tmpl_put() for the walkers: a template that is already stored unchanged
 * (the usual periodic re-send) only takes the read side of the lock */
static int
tmpl_store(struct xe_host *hs, const struct xe_tkey *k, const uint8_t *raw, size_t raw_len)
{
	struct xe_tslot *s = tmpl_find(hs, k);
	int rc;

	if (s && (s->raw_len == raw_len) && !memcmp(s->raw, raw, raw_len)) return XE_OK;
	tl_write(hs);
	rc = tmpl_put(hs, k, raw, raw_len);
	tl_write_done(hs);
	return rc;
}

/* ------------------------------------------------------------------ set emission */
/* `nrec` more records of device template `tmpl` in the batch: the fields it
 * sets and the column bytes a complete record fills, computed once per
 * template and batch */
int
xe_plan_note_template(struct xe_plan *plan, const struct xe_host *hs, uint32_t tmpl, size_t nrec)
{
	const struct xe_tmpl *t = (const struct xe_tmpl *)hs->dev_tmpls.p + tmpl;
	const struct xe_tent *e = (const struct xe_tent *)hs->dev_tents.p + t->ent_base;
	struct xe_tinfo *ti;
	size_t i, w = 0;

	while (plan->tinfo.n <= tmpl) {
		struct xe_tinfo z;
		memset(&z, 0, sizeof(z));
		if (!xe_blob_push(&plan->tinfo, &z, sizeof(z))) return 0;
	}
	ti = (struct xe_tinfo *)plan->tinfo.p + tmpl;
	if (!ti->known) {
		for (i = 0; i < t->nfields; i++) {
			if ((e[i].field != XE_F_NONE) && (e[i].flags & XE_TE_VALID)) {
				int f = e[i].field;
				ti->mask[f >> 6] |= 1ull << (f & 63);
				if (!(e[i].flags & XE_TE_REPEAT)) w += xe_fields[f].col_width;
			}
		}
		ti->width = (uint32_t)w;
		ti->known = 1;
		plan->field_mask[0] |= ti->mask[0];
		plan->field_mask[1] |= ti->mask[1];
	}
	plan->soa_bytes_per_rec_sum += (size_t)ti->width * nrec;
	return 1;
}

static int
emit_set(struct xe_plan *plan, const struct xe_host *hs, const struct dgram *d, long rel_off,
	uint32_t nrec, long avail, uint32_t stride, uint32_t tmpl, long extent, uint32_t flags)
{
	struct xe_set s;
	const struct xe_tmpl *t = (const struct xe_tmpl *)hs->dev_tmpls.p + tmpl;

	if (!nrec) return 1;
	if (plan->nrec + nrec > 0xFFFFFFF0u) { plan->too_many = 1; return 0; }
	memset(&s, 0, sizeof(s));
	s.data_off = d->base + (uint64_t)rel_off;
	s.first_rec = (uint32_t)plan->nrec;
	s.nrec = nrec;
	s.avail = (int32_t)avail;
	s.stride = stride;
	s.tmpl = tmpl;
	s.dgram_left = (int32_t)(d->len - rel_off);
	s.exporter = d->exporter;
	s.flags = flags;
	/* extent = the furthest byte (relative to rel_off) any record of the run
	 * may look at; past the datagram end the kernels must substitute zeros */
	if ((extent < 0) || ((rel_off + extent) > d->len) || (rel_off < 0) || t->varlen) s.flags |= XE_SET_CHECKED;
	plan->nrec += nrec;
	if (!xe_plan_note_template(plan, hs, tmpl, nrec)) return 0;
	return xe_blob_push(&plan->sets, &s, sizeof(s));
}

/* ------------------------------------------------------------------ netflow v9 */
/* parse_netflow_v9_flowset(), netflow.c:246-317 */
static int
v9_data_set(struct xe_host *hs, struct xe_plan *plan, struct dgram *d, long *ptr, uint16_t fsid_raw,
	long length, long count)
{
	struct xe_tkey k = { 9, fsid_raw, d->src_ip, d->source_id };
	struct xe_tslot *ts = tmpl_find(hs, &k);
	const struct xe_tmpl *t;
	const struct xe_tent *e;
	long S, nfull, done, start;

	if (!ts) return 0;
	if (ts->dev_slot == XE_TMPL_UNSUPPORTED) { plan->unsupported_sets++; return 0; }
	t = (const struct xe_tmpl *)hs->dev_tmpls.p + ts->dev_slot;
	e = (const struct xe_tent *)hs->dev_tents.p + t->ent_base;
	S = (t->varlen == 2) ? 0x10000 : t->stride;

	/* records that fit completely: (k+1)*S <= length */
	nfull = (S > 0) ? ((length >= 0) ? (long)((uint32_t)length / (uint32_t)S) : length / S) : count;
	if (nfull > count) nfull = count;
	if (S == 0) {
		/* every field is empty: `count` records, all at the set start */
		if (!emit_set(plan, hs, d, *ptr, (uint32_t)count, length, 0, ts->dev_slot, 0, XE_SET_PARTIAL)) return -1;
		*ptr += length;
		return 1;
	}
	if (!emit_set(plan, hs, d, *ptr, (uint32_t)nfull, length, (uint32_t)S, ts->dev_slot, nfull * S, 0)) return -1;
	done = nfull;
	start = nfull * S;
	/* the remaining iterations each start where the previous one stopped
	 * (the field cursor is not reset, netflow.c:271-291) */
	while (done < count) {
		long pos = start;
		size_t i;
		for (i = 0; i < t->nfields; i++) {
			if ((pos + e[i].len) > length) break;
			pos += e[i].len;
		}
		if (pos == start) {
			/* nothing consumed: all later iterations are identical */
			if (!emit_set(plan, hs, d, *ptr + start, (uint32_t)(count - done), length - start, 0,
				ts->dev_slot, 0, XE_SET_PARTIAL)) return -1;
			break;
		}
		if (!emit_set(plan, hs, d, *ptr + start, 1, length - start, 0, ts->dev_slot, pos - start,
			XE_SET_PARTIAL)) return -1;
		start = pos;
		done++;
	}
	*ptr += length;
	return 1;
}

/* parse_netflow_v9(), netflow.c:319-365 */
static int
walk_v9(struct xe_host *hs, struct xe_plan *plan, struct dgram *d)
{
	long ptr = 20;
	long count = g16(d, 2);

	d->source_id = g32raw(d, 16);
	if (!exporter_get(hs, plan, d->src_ip, d->source_id, &d->exporter)) return XE_ENOMEM;

	while (ptr < d->len) {
		uint8_t idb[2] = { g8(d, ptr), g8(d, ptr + 1) };
		uint16_t fsid_raw;
		long fsid = (idb[0] << 8) | idb[1];
		long length = g16(d, ptr + 2);

		memcpy(&fsid_raw, idb, 2);
		ptr += 4;
		if (fsid == 0) {
			/* parse_netflow_v9_template(), netflow.c:147-186: one
			 * template, cursor += its size */
			long fc = g16(d, ptr + 2);
			long tsize = 4 + fc * 4;
			uint8_t *raw;
			struct xe_tkey k = { 9, 0, d->src_ip, d->source_id };
			int rc;

			if (tsize > length) { plan->skipped_datagrams++; return XE_OK; }
			raw = malloc((size_t)tsize);
			if (!raw) return XE_ENOMEM;
			gcopy(d, ptr, raw, tsize);
			memcpy(&k.tid, raw, 2);
			rc = tmpl_store(hs, &k, raw, (size_t)tsize);
			free(raw);
			if (rc != XE_OK) return rc;
			ptr += tsize;
		} else if (fsid == 1) {
			break;
		} else {
			int rc = v9_data_set(hs, plan, d, &ptr, fsid_raw, length, count);
			if (rc < 0) return XE_ENOMEM;
			if (rc == 0) break;
		}
	}
	return XE_OK;
}

/* ------------------------------------------------------------------ IPFIX */
/* parse_ipfix_template() + ipfix_template_convert(), netflow.c:389-485 */
static int
ipfix_template(struct xe_host *hs, struct dgram *d, long ptr, long length, int *ok)
{
	long fc, avail = length - 4, p, i;
	uint8_t *raw;
	struct xe_tkey k = { 10, 0, d->src_ip, d->source_id };
	int rc;

	*ok = 0;
	if (4 > length) return XE_OK;
	fc = g16(d, ptr + 2);
	if (fc < 1) return XE_OK;
	raw = calloc(1, 4 + 8 * (size_t)fc);
	if (!raw) return XE_ENOMEM;
	gcopy(d, ptr, raw, 4);
	p = ptr + 4;
	for (i = 0; i < fc; i++) {
		uint8_t e[8];
		gcopy(d, p, e, 8);
		if (e[0] & 0x80) {
			if ((p - ptr + 8) > avail) { free(raw); return XE_OK; }
			e[0] &= 0x7f;
			memcpy(raw + 4 + 8 * i, e, 8);
			p += 8;
		} else {
			if ((p - ptr + 4) > avail) { free(raw); return XE_OK; }
			memcpy(raw + 4 + 8 * i, e, 4);
			p += 4;
		}
	}
	memcpy(&k.tid, raw, 2);
	rc = tmpl_store(hs, &k, raw, 4 + 8 * (size_t)fc);
	free(raw);
	if (rc == XE_OK) *ok = 1;
	return rc;
}

/* parse_ipfix_flowset(), netflow.c:547-625 */
static int
ipfix_data_set(struct xe_host *hs, struct xe_plan *plan, struct dgram *d, long ptr, uint16_t fsid_raw,
	long length)
{
	struct xe_tkey k = { 10, fsid_raw, d->src_ip, d->source_id };
	struct xe_tslot *ts = tmpl_find(hs, &k);
	const struct xe_tmpl *t;
	const struct xe_tent *e;
	long fc, S;

	if (!ts) return 0;
	if (ts->dev_slot == XE_TMPL_UNSUPPORTED) { plan->unsupported_sets++; return 0; }
	t = (const struct xe_tmpl *)hs->dev_tmpls.p + ts->dev_slot;
	e = (const struct xe_tent *)hs->dev_tents.p + t->ent_base;
	fc = ((long)ts->raw[2] << 8) | ts->raw[3];   /* the loop test uses the full field count */
	S = t->stride;

	if (!t->varlen) {
		long nrec;
		if (S == 0) {
			/* all fields empty: the reference never leaves its loop */
			plan->skipped_datagrams++;
			return 1;
		}
		/* record k exists iff length - k*S >= field_count; a record that
		 * reaches `length` ends the loop but is itself delivered */
		if (length < fc) return 1;
		/* (all operands are 16-bit quantities and not negative here: 32-bit
		 * unsigned division is several times cheaper than the 64-bit signed one) */
		nrec = (long)((uint32_t)(length - fc) / (uint32_t)S) + 1;
		/* record k gets every field iff its last field starts before
		 * `length` (netflow.c:596-600); the rest of the run is cut short */
		{
			long off_last = t->nfields ? e[t->nfields - 1].off : 0;
			long nfull = (length > off_last) ? (long)((uint32_t)(length - off_last - 1) / (uint32_t)S) + 1 : 0;
			if (nfull > nrec) nfull = nrec;
			if (!emit_set(plan, hs, d, ptr, (uint32_t)nfull, length, (uint32_t)S, ts->dev_slot, nfull * S, 0)) return -1;
			if (!emit_set(plan, hs, d, ptr + nfull * S, (uint32_t)(nrec - nfull), length - nfull * S, (uint32_t)S,
				ts->dev_slot, (nrec - nfull) * S, XE_SET_PARTIAL)) return -1;
		}
		return 1;
	}
	/* variable-length elements (netflow.c:367-383): find each record's start */
	{
		long pos = 0;
		int stop = 0;
		while (!stop) {
			long start = pos;
			size_t i;
			if ((length - pos) < fc) break;
			for (i = 0; i < t->nfields; i++) {
				long fl = e[i].len, adj = 0;
				if (fl == 0xFFFF) {
					uint8_t b0 = g8(d, ptr + pos);
					if (b0 != 255) { fl = b0; adj = 1; }
					else { fl = g16(d, ptr + pos + 1); adj = 3; }
				}
				pos += fl + adj;
				if (pos >= length) { stop = 1; break; }
			}
			if (!emit_set(plan, hs, d, ptr + start, 1, length - start, 0, ts->dev_slot, -1, XE_SET_PARTIAL)) return -1;
			/* a template cut at 65535 bytes never gets here without stop */
			if (!stop && (pos == start)) { plan->skipped_datagrams++; break; }
		}
	}
	return 1;
}

/* parse_ipfix(), netflow.c:628-678 */
static int
walk_ipfix(struct xe_host *hs, struct xe_plan *plan, struct dgram *d)
{
	long ptr = 16;

	d->source_id = g32raw(d, 12);
	if (!exporter_get(hs, plan, d->src_ip, d->source_id, &d->exporter)) return XE_ENOMEM;

	while (ptr < d->len) {
		uint8_t idb[2] = { g8(d, ptr), g8(d, ptr + 1) };
		uint16_t fsid_raw;
		long fsid = (idb[0] << 8) | idb[1];
		long length = g16(d, ptr + 2);

		memcpy(&fsid_raw, idb, 2);
		ptr += 4;
		if (fsid == 2) {
			int ok, rc = ipfix_template(hs, d, ptr, length, &ok);
			if (rc != XE_OK) return rc;
			if (!ok) break;
		} else if (fsid == 3) {
			/* options template: skipped */
		} else if (fsid > 255) {
			int rc = ipfix_data_set(hs, plan, d, ptr, fsid_raw, length);
			if (rc < 0) return XE_ENOMEM;
			if (rc == 0) break;
		}
		/* length 0 would park the reference on this set header forever */
		if (length == 0) { plan->skipped_datagrams++; break; }
		ptr += length - 4;
	}
	return XE_OK;
}

/* ------------------------------------------------------------------ netflow v5 */
/* netflow.h:42-62: the fixed 48-byte record as a template; the two pads are
 * unknown elements */
static int
v5_template(struct xe_host *hs, uint32_t *slot)
{
	static const uint16_t types[] = { 8, 12, 15, 10, 14, 2, 1, 22, 21, 7, 11, 0xFFFE, 6, 4, 5, 16, 17, 9, 13, 0xFFFD };
	static const uint16_t lens[] = { 4, 4, 4, 2, 2, 4, 4, 4, 4, 2, 2, 1, 1, 1, 1, 2, 2, 1, 1, 2 };
	return dev_template(hs, XE_MODE_V9, types, lens, sizeof(lens) / sizeof(lens[0]), slot);
}

/* parse_netflow_v5(), netflow.c:727-786 */
static int
walk_v5(struct xe_host *hs, struct xe_plan *plan, struct dgram *d)
{
	long n = g16(d, 2);
	uint32_t slot;
	int rc = XE_OK;

	if ((24 + 48 * n) != d->len) { plan->skipped_datagrams++; return XE_OK; }
	d->source_id = g8(d, 21);   /* engine_id widened to a host-order word */
	if (!exporter_get(hs, plan, d->src_ip, d->source_id, &d->exporter)) return XE_ENOMEM;
	slot = hs->v5_slot_plus1;
	if (!slot) {
		tl_write(hs);
		if (!hs->v5_slot_plus1) {
			rc = v5_template(hs, &slot);
			if (rc == XE_OK) { hs->v5_slot_plus1 = slot + 1; hs->tmpl_epoch++; }
		}
		slot = hs->v5_slot_plus1;
		tl_write_done(hs);
		if (rc != XE_OK) return rc;
	}
	slot--;
	return emit_set(plan, hs, d, 24, (uint32_t)n, 48 * n, 48, slot, 48 * n, 0) ? XE_OK : XE_ENOMEM;
}

/* ------------------------------------------------------------------ sFlow v5 */
struct sfcur { const struct dgram *d; long p; int bad; };

/* READ_SF_BYTES / READ32_H, sflow-impl.h:51-65 */
static uint32_t
sf_u32(struct sfcur *c)
{
	uint32_t v;
	if (c->bad || ((c->p + 4) > c->d->len)) { c->bad = 1; return 0; }
	/* the cursor only moves forward from 0, so the four bytes are inside */
	memcpy(&v, c->d->p + c->p, 4);
	c->p += 4;
	return __builtin_bswap32(v);
}

static uint32_t
sf_raw32(struct sfcur *c)
{
	uint32_t v;
	if (c->bad || ((c->p + 4) > c->d->len)) { c->bad = 1; return 0; }
	memcpy(&v, c->d->p + c->p, 4);
	c->p += 4;
	return v;
}

/* sf5_flow(), sflow-impl.h:68-272 */
static int
sf_flow_sample(struct xe_plan *plan, struct sfcur *c, int expanded, const struct xe_sfrec *proto)
{
	struct xe_sfrec r = *proto;
	uint32_t nel, j, v;
	int chained = 0;

	v = sf_u32(c);                   /* length */
	/* only the first bytes of every sample are read, a cache line or two out
	 * of several: ask for those of the samples to come, taking them to be as
	 * long as this one (a wrong guess costs nothing) */
	if (!c->bad && (v < 65536)) {
		const uint8_t *nx = c->d->p + c->p + 3 * ((long)v + 8);
		__builtin_prefetch(nx);
		__builtin_prefetch(nx + 64);
	}
	sf_u32(c);                       /* sequence */
	sf_u32(c);                       /* source id (or class) */
	if (expanded) sf_u32(c);         /* source index */
	r.sampling_rate = sf_u32(c);
	sf_u32(c);                       /* pool */
	sf_u32(c);                       /* drops */
	if (expanded) sf_u32(c);         /* input format */
	v = sf_u32(c);
	if (!expanded) v &= 0x3fffffff;
	v = htonl(v);
	r.in_if_be = v;
	if (expanded) sf_u32(c);         /* output format */
	v = sf_u32(c);
	if (!expanded) v &= 0x3fffffff;
	v = htonl(v);
	r.out_if_be = v;
	nel = sf_u32(c);
	if (c->bad) return XE_OK;

	for (j = 0; j < nel; j++) {
		uint32_t tag = sf_u32(c), ell = sf_u32(c);
		if (c->bad) return XE_OK;
		if (tag == 1) {
			uint32_t hp = sf_u32(c), hl;
			r.frame_len_be = sf_raw32(c);
			sf_u32(c);               /* stripped */
			hl = sf_u32(c);
			if (c->bad) return XE_OK;
			if (hp == 1) r.rp_type = 0;
			else if (hp == 11) r.rp_type = 1;
			else if (hp == 12) r.rp_type = 2;
			else { c->bad = 1; return XE_OK; }
			r.kind = 1;
			r.header_len = hl;
			r.hdr_off = c->d->base + (uint64_t)c->p;
			r.dgram_left = (int32_t)(c->d->len - c->p);
			r.chained = (uint8_t)chained;
			if (!xe_blob_push(&plan->sfrecs, &r, sizeof(r))) return XE_ENOMEM;
			plan->nsf++;
			chained = 1;
			c->p += (hl % 4) ? (long)hl + (4 - (hl % 4)) : (long)hl;
		} else if (tag == 1001) {
			sf_u32(c); sf_u32(c); sf_u32(c); sf_u32(c);
		} else if (tag == 3) {
			r.frame_len_be = sf_raw32(c);
			if (c->bad) return XE_OK;
			/* the 7 words after the length are read one by one and a
			 * short datagram aborts before the record is delivered */
			if ((c->p + 28) > c->d->len) { c->bad = 1; return XE_OK; }
			r.kind = 3;
			r.header_len = 28;
			r.hdr_off = c->d->base + (uint64_t)c->p;
			r.dgram_left = (int32_t)(c->d->len - c->p);
			r.chained = (uint8_t)chained;
			if (!xe_blob_push(&plan->sfrecs, &r, sizeof(r))) return XE_ENOMEM;
			plan->nsf++;
			chained = 1;
			c->p += 28;
		} else {
			c->p += ell;
		}
	}
	return XE_OK;
}

/* sflow_process(), sflow-impl.h:293-405 */
static int
walk_sflow(struct xe_host *hs, struct xe_plan *plan, struct dgram *d)
{
	struct sfcur c = { d, 0, 0 };
	struct xe_sfrec proto;
	uint32_t v, n, i;

	memset(&proto, 0, sizeof(proto));
	if (sf_u32(&c) != 5) return XE_OK;
	v = sf_u32(&c);
	if (c.bad) return XE_OK;
	if (v == 1) {
		proto.agent_ver = 4;
		if ((c.p + 4) > d->len) return XE_OK;
		gcopy(d, c.p, proto.agent, 4);
		c.p += 4;
	} else if (v == 2) {
		proto.agent_ver = 6;
		if ((c.p + 16) > d->len) return XE_OK;
		gcopy(d, c.p, proto.agent, 16);
		c.p += 16;
	} else {
		return XE_OK;
	}
	v = sf_raw32(&c);
	memcpy(proto.dev_id, &v, 4);
	sf_u32(&c);
	sf_u32(&c);
	n = sf_u32(&c);
	if (c.bad) return XE_OK;
	/* device rules use the UDP source and the packet's source_id field,
	 * which sflow_process() never sets: 0 here */
	if (!exporter_get(hs, plan, d->src_ip, 0, &d->exporter)) return XE_ENOMEM;
	proto.exporter = d->exporter;

	for (i = 0; i < n; i++) {
		v = sf_u32(&c);
		if (c.bad) break;
		if ((v == 1) || (v == 3)) {
			int rc = sf_flow_sample(plan, &c, v == 3, &proto);
			if (rc != XE_OK) return rc;
			if (c.bad) break;
		} else {
			uint32_t l = sf_u32(&c);
			if (c.bad) break;
			c.p += l;
		}
	}
	return XE_OK;
}

/* ------------------------------------------------------------------ entry */
/* one datagram: netflow_process(), netflow.c:804-819 / sflow_process() */
static int
walk_one(struct xe_host *hs, struct xe_plan *plan, const uint8_t *data, const uint64_t *offs,
	const uint32_t *src, const uint8_t *kinds, size_t i)
{
	struct dgram d;

	/* walked on the device: nothing to do here */
	if (kinds && (kinds[i] & XE_KIND_ON_DEVICE)) return XE_OK;
	memset(&d, 0, sizeof(d));
	d.p = data + offs[i];
	d.len = (long)(offs[i + 1] - offs[i]);
	if (d.len > 65536) d.len = 65536;
	d.base = offs[i];
	d.src_ip = src[i];
	if (kinds && (kinds[i] == XE_KIND_SFLOW)) return walk_sflow(hs, plan, &d);
	switch (g16(&d, 0)) {
		case 5: return walk_v5(hs, plan, &d);
		case 9: return walk_v9(hs, plan, &d);
		case 10: return walk_ipfix(hs, plan, &d);
		default: plan->skipped_datagrams++; break;
	}
	return XE_OK;
}

/* One datagram, with a bound on what it may expand to.  A Netflow v9 data set is
 * iterated header.count times whatever it holds (netflow.c:272), and a set of
 * length 0 moves the cursor by its 4 header bytes only, so a 1.4 KB datagram
 * of empty sets for a known template id stands for tens of millions of empty
 * records: the reference would grind through them, here each one would be a
 * row of every output column.  A datagram that yields more than
 * XE_MAX_RECORDS_PER_DATAGRAM records (no well-formed one comes near: 64 KiB
 * of one-byte records are 65536) is left out as a whole and counted. */
static int
walk_one_bounded(struct xe_host *hs, struct xe_plan *plan, const uint8_t *data, const uint64_t *offs,
	const uint32_t *src, const uint8_t *kinds, size_t i)
{
	const size_t s0 = plan->sets.n, r0 = plan->nrec, w0 = plan->soa_bytes_per_rec_sum;
	int rc = walk_one(hs, plan, data, offs, src, kinds, i);

	if ((rc == XE_OK) && (plan->nrec - r0 > XE_MAX_RECORDS_PER_DATAGRAM)) {
		plan->sets.n = s0;
		plan->nrec = r0;
		plan->soa_bytes_per_rec_sum = w0;
		plan->skipped_datagrams++;
		plan->oversized_datagrams++;
	}
	/* more records than a batch can index is not a memory problem */
	if ((rc == XE_ENOMEM) && plan->too_many) rc = XE_EUNSUPPORTED;
	return rc;
}

/*
 * Several threads: all state of the walk is keyed by exporter (templates,
 * sampling rate), so thread t takes the datagrams of the exporter addresses
 * that hash to t, in arrival order, into a plan of its own.  Per datagram the
 * owner notes how many sets, records and sFlow elements it produced; prefix
 * sums over the datagrams then give every set its place and every record its
 * index exactly as the single-threaded walk assigns them, and the threads copy
 * their pieces there.
 */
struct pp_ent          /* one datagram of a thread */
{
	uint32_t nsets, nrec, nsf, pad;        /* what its walk produced */
	uint64_t base_sets, base_rec, base_sf; /* where that goes in the merged plan */
};

struct pp_shared
{
	struct xe_host *hs;
	const uint8_t *data;
	const uint64_t *offs;
	const uint32_t *src;
	const uint8_t *kinds;
	size_t n;
	int T;
	uint8_t *own;                              /* thread of every datagram */
	struct xe_set *sets;                       /* final arrays */
	struct xe_sfrec *sfrecs;
	struct pp_thread *th;
};

struct pp_thread
{
	struct pp_shared *sh;
	int t, rc;
	pthread_t tid;
	struct xe_plan plan;
	struct xe_blob ents;        /* struct pp_ent, one per datagram of this thread, in order */
	struct xe_blob efirst;      /* size_t: datagram that introduced each local exporter */
	uint32_t *eremap;           /* local exporter index -> index in the merged table */
} __attribute__((aligned(128)));   /* no cache line shared between two threads' state */

static inline int
pp_owner(uint32_t ip, int T)
{
	/* the high half of a 64-bit product mixes every address bit */
	return (int)((uint32_t)(((uint64_t)ip * 0x9E3779B97F4A7C15ull) >> 40) % (uint32_t)T);
}

struct pp_efirst { size_t first; int t; uint32_t local; };

static int
pp_efirst_cmp(const void *a, const void *b)
{
	const struct pp_efirst *x = a, *y = b;
	return (x->first > y->first) - (x->first < y->first);
}

/* between the two phases, on one thread: places of sets and records, the
 * merged exporter table (in order of first appearance), the final arrays */
static int
pp_merge(struct pp_shared *sh, struct xe_plan *out)
{
	size_t i, ne = 0, k = 0, ns = 0, nr = 0, nf = 0;
	size_t cur[64];
	struct pp_efirst *ef;
	int t;

	memset(cur, 0, sizeof(cur));
	for (i = 0; i < sh->n; i++) {
		struct pp_ent *e;
		t = sh->own[i];
		e = (struct pp_ent *)sh->th[t].ents.p + cur[t]++;
		out->dg_first_nf[i] = (uint32_t)nr;
		out->dg_first_sf[i] = (uint32_t)nf;
		e->base_sets = ns; ns += e->nsets;
		e->base_rec = nr;  nr += e->nrec;
		e->base_sf = nf;   nf += e->nsf;
	}
	out->dg_first_nf[sh->n] = (uint32_t)nr;
	out->dg_first_sf[sh->n] = (uint32_t)nf;
	if (nr > 0xFFFFFFF0u) return XE_EUNSUPPORTED;
	for (t = 0; t < sh->T; t++) ne += sh->th[t].plan.exporters.n;
	ef = malloc((ne ? ne : 1) * sizeof(*ef));
	if (!ef) return XE_ENOMEM;
	for (t = 0; t < sh->T; t++) {
		for (i = 0; i < sh->th[t].plan.exporters.n; i++) {
			ef[k].first = ((size_t *)sh->th[t].efirst.p)[i];
			ef[k].t = t;
			ef[k].local = (uint32_t)i;
			k++;
		}
		sh->th[t].eremap = malloc((sh->th[t].plan.exporters.n ? sh->th[t].plan.exporters.n : 1) * sizeof(uint32_t));
		if (!sh->th[t].eremap) { free(ef); return XE_ENOMEM; }
	}
	qsort(ef, ne, sizeof(*ef), pp_efirst_cmp);
	/* an exporter met by several threads (sFlow datagrams are dealt out by
	 * position, not by address) is one entry, at its first appearance */
	{
		size_t cap = 64, m;
		uint32_t *hm;
		while (cap < 2 * ne + 2) cap <<= 1;
		m = cap - 1;
		hm = calloc(cap, sizeof(uint32_t));
		if (!hm) { free(ef); return XE_ENOMEM; }
		for (k = 0; k < ne; k++) {
			const struct xe_exporter *e = (const struct xe_exporter *)sh->th[ef[k].t].plan.exporters.p + ef[k].local;
			uint64_t key;
			size_t j;
			memcpy(&key, e->dev_ip, 8);   /* address and source id lie side by side */
			for (j = ((key * 0x9E3779B97F4A7C15ull) >> 32) & m; hm[j]; j = (j + 1) & m) {
				if (!memcmp(((const struct xe_exporter *)out->exporters.p)[hm[j] - 1].dev_ip, e->dev_ip, 8)) break;
			}
			if (!hm[j]) {
				hm[j] = (uint32_t)out->exporters.n + 1;
				if (!xe_blob_push(&out->exporters, e, sizeof(*e))) { free(hm); free(ef); return XE_ENOMEM; }
			}
			sh->th[ef[k].t].eremap[ef[k].local] = hm[j] - 1;
		}
		free(hm);
	}
	free(ef);
	out->pool = sh->hs;
	out->sets.p = pool_get(sh->hs, ns * sizeof(struct xe_set), &out->sets.cap);
	out->sfrecs.p = pool_get(sh->hs, nf * sizeof(struct xe_sfrec), &out->sfrecs.cap);
	if (!out->sets.p || !out->sfrecs.p) return XE_ENOMEM;
	out->sets.n = ns;
	out->sfrecs.n = nf;
	out->nrec = nr;
	out->nsf = nf;
	for (t = 0; t < sh->T; t++) {
		const struct xe_plan *p = &sh->th[t].plan;
		const struct xe_tinfo *ti = p->tinfo.p;
		out->field_mask[0] |= p->field_mask[0];
		out->field_mask[1] |= p->field_mask[1];
		out->soa_bytes_per_rec_sum += p->soa_bytes_per_rec_sum;
		out->skipped_datagrams += p->skipped_datagrams;
		out->oversized_datagrams += p->oversized_datagrams;
		out->unsupported_sets += p->unsupported_sets;
		for (i = 0; i < p->tinfo.n; i++) {
			while (out->tinfo.n <= i) {
				struct xe_tinfo z;
				memset(&z, 0, sizeof(z));
				if (!xe_blob_push(&out->tinfo, &z, sizeof(z))) return XE_ENOMEM;
			}
			if (ti[i].known) ((struct xe_tinfo *)out->tinfo.p)[i] = ti[i];
		}
	}
	return XE_OK;
}

/* phase 1: walk this thread's datagrams */
static void *
pp_walk(void *arg)
{
	struct pp_thread *me = arg;
	struct pp_shared *sh = me->sh;
	struct xe_plan *plan = &me->plan;
	size_t i, since = 0, la = 0;
	unsigned ahead = 0;

	pthread_rwlock_rdlock(&sh->hs->tlock);
	for (i = 0; (i < sh->n) && (me->rc == XE_OK); i++) {
		struct pp_ent e;
		size_t s0, r0, f0, e0;
		if (sh->own[i] != me->t) continue;
		/* only headers are read, a cache line per datagram, far apart: a cursor
		 * runs ahead and asks for the first line of this thread's next eight */
		if (la <= i) { la = i + 1; ahead = 0; }
		else if (ahead) ahead--;
		while ((ahead < 8) && (la < sh->n)) {
			if ((sh->own[la] == me->t) && !(sh->kinds && (sh->kinds[la] & XE_KIND_ON_DEVICE))) {
				__builtin_prefetch(sh->data + sh->offs[la]);
				ahead++;
			}
			la++;
		}
		if (++since == 512) {
			/* a thread that waits to store a template gets its turn */
			pthread_rwlock_unlock(&sh->hs->tlock);
			pthread_rwlock_rdlock(&sh->hs->tlock);
			since = 0;
		}
		s0 = plan->sets.n; r0 = plan->nrec; f0 = plan->nsf; e0 = plan->exporters.n;
		me->rc = walk_one_bounded(sh->hs, plan, sh->data, sh->offs, sh->src, sh->kinds, i);
		memset(&e, 0, sizeof(e));
		e.nsets = (uint32_t)(plan->sets.n - s0);
		e.nrec = (uint32_t)(plan->nrec - r0);
		e.nsf = (uint32_t)(plan->nsf - f0);
		if (!xe_blob_push(&me->ents, &e, sizeof(e))) me->rc = XE_ENOMEM;
		for (; e0 < plan->exporters.n; e0++) {
			if (!xe_blob_push(&me->efirst, &i, sizeof(i))) me->rc = XE_ENOMEM;
		}
	}
	pthread_rwlock_unlock(&sh->hs->tlock);
	return NULL;
}

/* phase 2: this thread's pieces to their places */
static void *
pp_place(void *arg)
{
	struct pp_thread *me = arg;
	struct pp_shared *sh = me->sh;
	const struct xe_set *ls = me->plan.sets.p;
	const struct xe_sfrec *lf = me->plan.sfrecs.p;
	const struct pp_ent *en = me->ents.p;
	size_t cs = 0, cr = 0, cf = 0, i;

	for (i = 0; i < me->ents.n; i++) {
		const struct pp_ent *e = &en[i];
		uint32_t k;
		for (k = 0; k < e->nsets; k++) {
			struct xe_set s = ls[cs + k];
			s.first_rec = (uint32_t)(e->base_rec + (s.first_rec - cr));
			s.exporter = me->eremap[s.exporter];
			sh->sets[e->base_sets + k] = s;
		}
		for (k = 0; k < e->nsf; k++) {
			struct xe_sfrec r = lf[cf + k];
			r.exporter = me->eremap[r.exporter];
			sh->sfrecs[e->base_sf + k] = r;
		}
		cs += e->nsets;
		cr += e->nrec;
		cf += e->nsf;
	}
	return NULL;
}

/* fn on every thread's share; the caller is thread 0, and takes over the share
 * of any thread that could not be started */
static void
pp_all(struct pp_shared *sh, void *(*fn)(void *))
{
	int t;
	char started[64];

	memset(started, 0, sizeof(started));
	for (t = 1; t < sh->T; t++) {
		if (pthread_create(&sh->th[t].tid, NULL, fn, &sh->th[t]) == 0) started[t] = 1;
	}
	fn(&sh->th[0]);
	for (t = 1; t < sh->T; t++) {
		if (started[t]) pthread_join(sh->th[t].tid, NULL);
		else fn(&sh->th[t]);
	}
}

static void
walkers_free(struct xe_host *hs)
{
	struct pp_thread *th = hs->walkers;
	int t;

	for (t = 0; t < hs->nwalkers; t++) {
		xe_plan_free(&th[t].plan);
		free(th[t].ents.p);
		free(th[t].efirst.p);
		free(th[t].eremap);
	}
	free(th);
	hs->walkers = NULL;
	hs->nwalkers = 0;
}

/* a walker's state for the next batch: empty, arrays kept at their size */
static void
walker_reset(struct pp_thread *w)
{
	struct xe_plan *p = &w->plan;

	p->sets.n = p->exporters.n = p->sfrecs.n = p->tinfo.n = 0;
	p->nrec = p->nsf = 0;
	p->field_mask[0] = p->field_mask[1] = 0;
	p->soa_bytes_per_rec_sum = 0;
	p->skipped_datagrams = 0;
	p->oversized_datagrams = 0;
	p->too_many = 0;
	p->unsupported_sets = 0;
	if (p->ehash) memset(p->ehash, 0, p->ehash_cap * sizeof(uint32_t));
	w->ents.n = w->efirst.n = 0;
	free(w->eremap);
	w->eremap = NULL;
	w->rc = XE_OK;
}

static int
prepass_threads(struct xe_host *hs, const uint8_t *data, const uint64_t *offs, const uint32_t *src,
	const uint8_t *kinds, size_t n, struct xe_plan *plan, int T)
{
	struct pp_shared sh;
	struct pp_thread *th;
	int t, rc = XE_OK;

	memset(&sh, 0, sizeof(sh));
	if (hs->nwalkers != T) {
		walkers_free(hs);
		if (posix_memalign((void **)&th, 128, (size_t)T * sizeof(*th)) != 0) return XE_ENOMEM;
		memset(th, 0, (size_t)T * sizeof(*th));
		hs->walkers = th;
		hs->nwalkers = T;
	}
	th = hs->walkers;
	for (t = 0; t < T; t++) walker_reset(&th[t]);
	sh.hs = hs; sh.data = data; sh.offs = offs; sh.src = src; sh.kinds = kinds; sh.n = n; sh.T = T; sh.th = th;
	for (t = 0; t < T; t++) { th[t].sh = &sh; th[t].t = t; }
	sh.own = malloc(n ? n : 1);
	if (!sh.own) return XE_ENOMEM;
	{
		size_t i;
		/* Netflow: an exporter's datagrams stay on one thread, in order (its
		 * templates are state).  sFlow datagrams carry no state from one to
		 * the next: runs of 32 are dealt out in turn, whatever the agents */
		for (i = 0; i < n; i++) {
			if (kinds && (kinds[i] == XE_KIND_SFLOW)) sh.own[i] = (uint8_t)((i >> 5) % (size_t)T);
			else sh.own[i] = (uint8_t)pp_owner(src[i], T);
		}
	}

	struct timespec t0_, t1_, t2_, t3_;
	clock_gettime(CLOCK_MONOTONIC, &t0_);
	hs->mt = 1;
	pp_all(&sh, pp_walk);
	hs->mt = 0;
	clock_gettime(CLOCK_MONOTONIC, &t1_);
	for (t = 0; t < T; t++) if ((rc == XE_OK) && (th[t].rc != XE_OK)) rc = th[t].rc;
	if (rc == XE_OK) rc = pp_merge(&sh, plan);
	clock_gettime(CLOCK_MONOTONIC, &t2_);
	if (rc == XE_OK) {
		sh.sets = plan->sets.p;
		sh.sfrecs = plan->sfrecs.p;
		pp_all(&sh, pp_place);
	}
	clock_gettime(CLOCK_MONOTONIC, &t3_);
	if (getenv("XE_DEBUG")) fprintf(stderr, "prepass T=%d walk %.2f merge %.2f place %.2f ms\n", T,
		((t1_.tv_sec - t0_.tv_sec) * 1e3 + (t1_.tv_nsec - t0_.tv_nsec) * 1e-6),
		((t2_.tv_sec - t1_.tv_sec) * 1e3 + (t2_.tv_nsec - t1_.tv_nsec) * 1e-6),
		((t3_.tv_sec - t2_.tv_sec) * 1e3 + (t3_.tv_nsec - t2_.tv_nsec) * 1e-6));
	free(sh.own);
	if (rc != XE_OK) xe_plan_free(plan);
	return rc;
}

int
xe_prepass(struct xe_host *hs, const uint8_t *data, const uint64_t *offs, const uint32_t *src,
	const uint8_t *kinds, size_t n, struct xe_plan *plan)
{
	size_t i;
	int T = hs->prepass_threads;

	memset(plan, 0, sizeof(*plan));
	/* both per-datagram arrays in one allocation */
	plan->dg_first_nf = malloc(2 * (n + 1) * sizeof(uint32_t));
	if (!plan->dg_first_nf) return XE_ENOMEM;
	plan->dg_first_sf = plan->dg_first_nf + (n + 1);
	plan->ndg = n;
	if (T <= 0) {
		const char *e = getenv("XE_PREPASS_THREADS");
		T = e ? atoi(e) : 8;
	}
	if (T > 64) T = 64;
	/* small batches are not worth the thread start (the environment variable
	 * lets tests send small batches through the threaded walk) */
	{
		const char *e = getenv("XE_PREPASS_MIN_DATAGRAMS");
		const size_t least = e ? (size_t)atol(e) : 4096;
		/* what the device has walked itself is only passed by: a batch that leaves
		 * the host a few datagrams does not pay for starting threads either */
		size_t left = n;
		if (kinds) {
			left = 0;
			for (i = 0; i < n; i++) left += !(kinds[i] & XE_KIND_ON_DEVICE);
		}
		if ((T > 1) && (n >= least) && (left >= least)) {
			int rc = prepass_threads(hs, data, offs, src, kinds, n, plan, T);
			if (rc != XE_OK) xe_plan_free(plan);   /* also after an early return */
			return rc;
		}
	}
	for (i = 0; i < n; i++) {
		int rc;

		/* only headers are read, one cache line per datagram, far apart: ask
		 * for the lines of the datagrams to come */
		if ((i + 16 < n) && !(kinds && (kinds[i + 16] & XE_KIND_ON_DEVICE))) __builtin_prefetch(data + offs[i + 16]);
		plan->dg_first_nf[i] = (uint32_t)plan->nrec;
		plan->dg_first_sf[i] = (uint32_t)plan->nsf;
		rc = walk_one_bounded(hs, plan, data, offs, src, kinds, i);
		if (rc != XE_OK) {
			xe_plan_free(plan);
			return rc;
		}
	}
	plan->dg_first_nf[n] = (uint32_t)plan->nrec;
	plan->dg_first_sf[n] = (uint32_t)plan->nsf;
	return XE_OK;
}
