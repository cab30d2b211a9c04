/*
 * xe_hosttest.c -- entry points that let the host-side logic (filter
 * compiler, IP-list loader and flattener, datagram pre-pass) be exercised
 * without a GPU.  They are not part of the public ABI (include/xe_gpu.h does
 * not declare them) and compute no flow results: decode and classification
 * exist only in the HIP kernels.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "xe_internal.h"

struct xe_host *
xe_ht_create(void)
{
	struct xe_host *hs = calloc(1, sizeof(*hs));
	if (hs) xe_host_init(hs);
	return hs;
}

void
xe_ht_free(struct xe_host *hs)
{
	if (!hs) return;
	xe_host_free(hs);
	free(hs);
}

int
xe_ht_add_iplist(struct xe_host *hs, const char *name, const char *text, size_t len)
{
	return xe_iplist_load_text(&hs->lists, name, text, len);
}

int
xe_ht_load_iplists(struct xe_host *hs, const char *dir)
{
	return xe_iplist_load_dir(&hs->lists, dir);
}

/* XE_OK, XE_EPARSE (err filled) or XE_EUNSUPPORTED (parses, dump filled, not
 * evaluable on the device: reason in err) */
int
xe_ht_filter_dump(struct xe_host *hs, const char *text, char *out, size_t outlen, char *err, size_t errlen)
{
	struct xe_filter f;
	int rc = xe_filter_compile(hs ? &hs->lists : NULL, text, &f, err, errlen);
	if (rc != XE_OK) return rc;
	xe_filter_dump_text(hs ? &hs->lists : NULL, &f, out, outlen);
	if (!f.device_ok) {
		rc = XE_EUNSUPPORTED;
		if (err && errlen) snprintf(err, errlen, "%s", f.why);
	} else if (f.needs_port_names) {
		/* no handle, no port-name table */
		rc = XE_EUNSUPPORTED;
		if (err && errlen) snprintf(err, errlen, "portstr() / ppstr() without a port-name table");
	} else if (f.needs_geodb) {
		/* no handle, no geo / AS database */
		rc = XE_EUNSUPPORTED;
		if (err && errlen) snprintf(err, errlen, "geo / AS function without its database");
	}
	xe_filter_free(&f);
	return rc;
}

/* flattened lists: *words is malloc'd (caller frees with xe_ht_release),
 * dev[] gets one entry per list in load order */
int
xe_ht_flatten(struct xe_host *hs, uint32_t **words, size_t *nwords, struct xe_list_dev *dev, size_t ndev, int root_bits)
{
	struct xe_blob b = {0};
	int rc;
	if (ndev < hs->lists.n) return XE_EINVAL;
	rc = xe_iplists_flatten(&hs->lists, &b, dev, root_bits);
	if (rc != XE_OK) { free(b.p); return rc; }
	*words = b.p;
	*nwords = b.n;
	return XE_OK;
}

void xe_ht_release(void *p) { free(p); }
size_t xe_ht_nlists(struct xe_host *hs) { return hs->lists.n; }

int
xe_ht_add_device(struct xe_host *hs, int use_ip, uint32_t ip, int use_id, uint32_t id_net, int rate, int skip)
{
	struct xe_device d, *nd;
	memset(&d, 0, sizeof(d));
	d.use_ip = use_ip; d.ip = ip; d.use_id = use_id; d.id = id_net;
	d.sampling_rate = rate; d.skip_unmarked = skip;
	nd = realloc(hs->devs, (hs->ndevs + 1) * sizeof(*nd));
	if (!nd) return XE_ENOMEM;
	hs->devs = nd;
	hs->devs[hs->ndevs++] = d;
	return XE_OK;
}

/* run the datagram pre-pass; the plan stays inside the returned object */
struct xe_plan *
xe_ht_prepass(struct xe_host *hs, const uint8_t *data, const uint64_t *offs, const uint32_t *src,
	const uint8_t *kinds, size_t n, int *rc_out)
{
	struct xe_plan *p = calloc(1, sizeof(*p));
	int rc;
	if (!p) { if (rc_out) *rc_out = XE_ENOMEM; return NULL; }
	rc = xe_prepass(hs, data, offs, src, kinds, n, p);
	if (rc_out) *rc_out = rc;
	if (rc != XE_OK) { free(p); return NULL; }
	return p;
}

void xe_ht_plan_free(struct xe_plan *p) { if (p) { xe_plan_free(p); free(p); } }
void xe_ht_set_prepass_threads(struct xe_host *hs, int n) { hs->prepass_threads = n; }

/* a fixed-window field text: {aggr, offset in struct flow_info, size, scale, descending} */
int
xe_ht_field_parse(const char *text, int out[9], char *err, size_t errlen)
{
	struct xe_agfield f;
	int rc = xe_field_parse(text, &f, err, errlen);
	if (rc != XE_OK) return rc;
	out[0] = f.aggr; out[1] = xe_fields[f.field].off; out[2] = (int)f.size; out[3] = f.scale; out[4] = f.descending;
	/* function forms: which, the second argument, k (the first argument is out[1]) */
	const int two = f.func && (f.func != XE_FN_TFSTR) && (f.func != XE_FN_PORTSTR);
	out[5] = f.func; out[6] = two ? (int)xe_fields[f.fb].off : 0; out[7] = f.func ? f.k : 0;
	out[8] = f.func ? (int)xe_fields[f.field].size_max * 256 + (two ? (int)xe_fields[f.fb].size_max : 0) : 0;
	return XE_OK;
}

/* stride of the flattened list nodes, for the tests' own walk */
int xe_ht_node_bits(void) { return XE_NODE_BITS; }

size_t xe_ht_plan_nrec(const struct xe_plan *p) { return p->nrec; }
size_t xe_ht_plan_nsf(const struct xe_plan *p) { return p->nsf; }
size_t xe_ht_plan_nsets(const struct xe_plan *p) { return p->sets.n; }
size_t xe_ht_plan_nexporters(const struct xe_plan *p) { return p->exporters.n; }
size_t xe_ht_plan_skipped(const struct xe_plan *p) { return p->skipped_datagrams; }
size_t xe_ht_plan_oversized(const struct xe_plan *p) { return p->oversized_datagrams; }
size_t xe_ht_plan_ndatagrams(const struct xe_plan *p) { return p->ndg; }
const uint32_t *xe_ht_plan_dg_first_nf(const struct xe_plan *p) { return p->dg_first_nf; }
const uint32_t *xe_ht_plan_dg_first_sf(const struct xe_plan *p) { return p->dg_first_sf; }
const struct xe_set *xe_ht_plan_sets(const struct xe_plan *p) { return p->sets.p; }
const struct xe_exporter *xe_ht_plan_exporters(const struct xe_plan *p) { return p->exporters.p; }
size_t xe_ht_ntemplates(const struct xe_host *hs) { return hs->dev_tmpls.n; }
const struct xe_tmpl *xe_ht_templates(const struct xe_host *hs) { return hs->dev_tmpls.p; }
const struct xe_tent *xe_ht_template_entries(const struct xe_host *hs) { return hs->dev_tents.p; }

/* the config file readers */
struct collect { char *buf; size_t cap, len; int count; };

static int
collect_mo(void *user, const char *name, int parent, const char *filter)
{
	struct collect *c = user;
	int n = snprintf(c->buf + c->len, (c->len < c->cap) ? c->cap - c->len : 0, "%s\t%d\t%s\n", name, parent, filter);
	if (n > 0) c->len += (size_t)n;
	return c->count++;
}

/* lists "name<TAB>parent<TAB>filter" lines of the objects xe_gpu_load_mos() would add */
int
xe_ht_list_mos(const char *mo_dir, char *out, size_t outlen)
{
	struct xe_host hs;
	struct collect c = { out, outlen, 0, 0 };
	int rc;
	xe_host_init(&hs);
	if (outlen) out[0] = '\0';
	rc = xe_conf_load_mos(&hs, mo_dir, collect_mo, NULL, &c);
	xe_host_free(&hs);
	return rc;
}

static int
collect_dev(void *user, int use_ip, uint32_t ip, int use_id, uint32_t id, int rate, int skip,
	const char *const *marks, int nmarks)
{
	struct collect *c = user;
	int i, n = snprintf(c->buf + c->len, (c->len < c->cap) ? c->cap - c->len : 0, "%d\t%u\t%d\t%u\t%d\t%d",
		use_ip, ip, use_id, id, rate, skip);
	if (n > 0) c->len += (size_t)n;
	for (i = 0; i < nmarks; i++) {
		n = snprintf(c->buf + c->len, (c->len < c->cap) ? c->cap - c->len : 0, "\t%s", marks[i]);
		if (n > 0) c->len += (size_t)n;
	}
	n = snprintf(c->buf + c->len, (c->len < c->cap) ? c->cap - c->len : 0, "\n");
	if (n > 0) c->len += (size_t)n;
	return XE_OK;
}

int
xe_ht_list_devices(const char *path, char *out, size_t outlen)
{
	struct xe_host hs;
	struct collect c = { out, outlen, 0, 0 };
	int rc;
	xe_host_init(&hs);
	if (outlen) out[0] = '\0';
	rc = xe_conf_load_devices(&hs, path, collect_dev, &c);
	xe_host_free(&hs);
	return rc;
}

/* geo / AS databases and the lists made of them (xe_geodb.c) */
int
xe_ht_geodb_set(struct xe_host *hs, int kind, int family, const void *img, size_t bytes)
{
	char err[256];
	return xe_geodb_set(hs, kind, family, img, bytes, err, sizeof(err));
}

/* ranges: n pairs lo, hi (asn); strs: n texts (asd / geo field); returns the list's index */
int
xe_ht_geodb_list(struct xe_host *hs, int func, int geo_field, int family, const int32_t *lo, const int32_t *hi,
	const char *const *strs, size_t n)
{
	return xe_geodb_list(hs, func, geo_field, family, lo, hi, strs, n);
}

/* the log of template changes (xe_prepass.c) */
void xe_ht_undo_begin(struct xe_host *hs) { xe_host_undo_begin(hs); }
size_t xe_ht_undo_count(struct xe_host *hs) { return xe_host_undo_count(hs); }
void xe_ht_undo_rollback(struct xe_host *hs) { xe_host_undo_rollback(hs); }
void xe_ht_undo_end(struct xe_host *hs) { xe_host_undo_end(hs); }
