/*
 * xe_conf.c -- readers for the reference's configuration files, as far as the
 * flow path needs them: the "filter" key of every mo.conf below mo-dir
 * (monit-objects.c:47-105, directory walk :252-458) and devices.conf
 * (devices.c:60-139).  Both are JSON with C and C++ style comments
 * (CONFIG.md:13-17); unknown keys are ignored.  The reference parses them
 * with the aajson library, which is not part of its tree; this is a small
 * reader of the same syntax that reports every scalar with its key path.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dirent.h>
#include <sys/stat.h>
#include <arpa/inet.h>
#include "xe_internal.h"

#define PATH_DEPTH 16

enum { JV_STRING, JV_NUMBER, JV_TRUE, JV_FALSE, JV_NULL };

struct jpath_item { int is_index; long index; char key[128]; };

struct jparser
{
	const char *s;
	int failed;
	struct jpath_item path[PATH_DEPTH];
	int depth;
	int (*cb)(void *ctx, const struct jpath_item *path, int depth, int type, const char *val);
	void *ctx;
};

static void
jskip(struct jparser *p)
{
	for (;;) {
		while ((*p->s == ' ') || (*p->s == '\t') || (*p->s == '\n') || (*p->s == '\r')) p->s++;
		if ((p->s[0] == '/') && (p->s[1] == '/')) {
			while (*p->s && (*p->s != '\n')) p->s++;
		} else if ((p->s[0] == '/') && (p->s[1] == '*')) {
			p->s += 2;
			while (*p->s && !((p->s[0] == '*') && (p->s[1] == '/'))) p->s++;
			if (*p->s) p->s += 2; else p->failed = 1;
		} else {
			return;
		}
	}
}

static int
jstring(struct jparser *p, char *out, size_t outlen)
{
	size_t n = 0;

	if (*p->s != '"') { p->failed = 1; return 0; }
	p->s++;
	while (*p->s && (*p->s != '"')) {
		char c = *p->s++;
		if (c == '\\') {
			c = *p->s++;
			switch (c) {
				case 'n': c = '\n'; break;
				case 't': c = '\t'; break;
				case 'r': c = '\r'; break;
				case 'b': c = '\b'; break;
				case 'f': c = '\f'; break;
				case '\0': p->failed = 1; return 0;
				default: break;   /* \" \\ \/ and anything else: the character itself */
			}
		}
		if (n + 1 < outlen) out[n++] = c;
	}
	if (*p->s != '"') { p->failed = 1; return 0; }
	p->s++;
	out[n] = '\0';
	return 1;
}

static void jvalue(struct jparser *p);

static void
jscalar(struct jparser *p, int type, const char *val)
{
	if (p->cb && !p->failed) {
		if (!p->cb(p->ctx, p->path, p->depth, type, val)) p->failed = 1;
	}
}

static void
jvalue(struct jparser *p)
{
	char buf[4096];

	jskip(p);
	if (p->failed) return;
	if (*p->s == '{') {
		p->s++;
		jskip(p);
		if (*p->s == '}') { p->s++; return; }
		for (;;) {
			if (p->depth >= PATH_DEPTH) { p->failed = 1; return; }
			jskip(p);
			if (!jstring(p, p->path[p->depth].key, sizeof(p->path[p->depth].key))) return;
			p->path[p->depth].is_index = 0;
			jskip(p);
			if (*p->s != ':') { p->failed = 1; return; }
			p->s++;
			p->depth++;
			jvalue(p);
			p->depth--;
			if (p->failed) return;
			jskip(p);
			if (*p->s == ',') { p->s++; continue; }
			if (*p->s == '}') { p->s++; return; }
			p->failed = 1;
			return;
		}
	}
	if (*p->s == '[') {
		long idx = 0;
		p->s++;
		jskip(p);
		if (*p->s == ']') { p->s++; return; }
		for (;;) {
			if (p->depth >= PATH_DEPTH) { p->failed = 1; return; }
			p->path[p->depth].is_index = 1;
			p->path[p->depth].index = idx++;
			p->path[p->depth].key[0] = '\0';
			p->depth++;
			jvalue(p);
			p->depth--;
			if (p->failed) return;
			jskip(p);
			if (*p->s == ',') { p->s++; continue; }
			if (*p->s == ']') { p->s++; return; }
			p->failed = 1;
			return;
		}
	}
	if (*p->s == '"') {
		if (jstring(p, buf, sizeof(buf))) jscalar(p, JV_STRING, buf);
		return;
	}
	if (!strncmp(p->s, "true", 4)) { p->s += 4; jscalar(p, JV_TRUE, "true"); return; }
	if (!strncmp(p->s, "false", 5)) { p->s += 5; jscalar(p, JV_FALSE, "false"); return; }
	if (!strncmp(p->s, "null", 4)) { p->s += 4; jscalar(p, JV_NULL, "null"); return; }
	{
		size_t n = 0;
		while (*p->s && strchr("+-0123456789.eE", *p->s) && (n + 1 < sizeof(buf))) buf[n++] = *p->s++;
		buf[n] = '\0';
		if (!n) { p->failed = 1; return; }
		jscalar(p, JV_NUMBER, buf);
	}
}

static char *
read_file(const char *path)
{
	FILE *f = fopen(path, "rb");
	long len;
	char *buf;

	if (!f) return NULL;
	fseek(f, 0, SEEK_END);
	len = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf = malloc((size_t)len + 1);
	if (buf && (fread(buf, 1, (size_t)len, f) != (size_t)len)) { free(buf); buf = NULL; }
	if (buf) buf[len] = '\0';
	fclose(f);
	return buf;
}

/* ------------------------------------------------------------------ mo.conf */
#define MO_MAX_FWM 16
#define MO_MAX_FWM_FIELDS 24
struct mo_fwm_conf { char name[256]; char fields[MO_MAX_FWM_FIELDS][128]; int nfields, limit, time; };
struct mo_ctx { char filter[4096]; int seen, bad; struct mo_fwm_conf *fwm; int nfwm; };

static int
mo_cb(void *vctx, const struct jpath_item *path, int depth, int type, const char *val)
{
	struct mo_ctx *c = vctx;
	/* only a top-level "filter" counts (monit-objects.c:55-56) */
	if ((depth == 1) && !path[0].is_index && !strcmp(path[0].key, "filter") && (type == JV_STRING)) {
		snprintf(c->filter, sizeof(c->filter), "%s", val);
		c->seen = 1;
	}
	/* "fwm": [ { "name": ..., "fields": [ ... ] } ] (fwm_config(), monit-objects-fwm.c:139-205) */
	if (c->fwm && (depth >= 3) && !path[0].is_index && !strcmp(path[0].key, "fwm") && path[1].is_index
		&& (path[1].index >= 0) && (path[1].index < MO_MAX_FWM) && !path[2].is_index) {
		struct mo_fwm_conf *w = &c->fwm[path[1].index];
		if (path[1].index >= c->nfwm) c->nfwm = (int)path[1].index + 1;
		if ((type == JV_STRING) && (depth == 3) && !strcmp(path[2].key, "name")) snprintf(w->name, sizeof(w->name), "%s", val);
		if ((type == JV_STRING) && (depth == 4) && !strcmp(path[2].key, "fields") && path[3].is_index
			&& (w->nfields < MO_MAX_FWM_FIELDS)) {
			snprintf(w->fields[w->nfields++], sizeof(w->fields[0]), "%s", val);
		}
		/* atoi() of the value's text whatever its JSON type; a negative
		 * number fails the object's configuration (:179-190) */
		if ((depth == 3) && val && !strcmp(path[2].key, "time")) { w->time = atoi(val); if (w->time < 0) c->bad = 1; }
		if ((depth == 3) && val && !strcmp(path[2].key, "limit")) { w->limit = atoi(val); if (w->limit < 0) c->bad = 1; }
	}
	return 1;
}

static int
load_mo_dir(struct xe_host *hs, const char *root, const char *suffix, int parent,
	int (*add)(void *, const char *, int, const char *),
	int (*add_fwm)(void *, int, const char *, const char *const *, int, int, int), void *user)
{
	char dirname[4096];
	DIR *d;
	struct dirent *de;

	snprintf(dirname, sizeof(dirname), "%s/%s/", root, suffix);
	d = opendir(dirname);
	if (!d) return (parent < 0) ? XE_EIO : XE_OK;
	while ((de = readdir(d)) != NULL) {
		char path[8192], name[1024], inner[1024];
		struct stat st;
		char *json;
		struct jparser p;
		struct mo_ctx ctx;
		int idx;
		size_t i;

		if (de->d_name[0] == '.') continue;
		snprintf(path, sizeof(path), "%s/%s", dirname, de->d_name);
		if ((stat(path, &st) != 0) || !S_ISDIR(st.st_mode)) continue;
		snprintf(path, sizeof(path), "%s/%s/mo.conf", dirname, de->d_name);
		if (*suffix) snprintf(inner, sizeof(inner), "%s/%s", suffix, de->d_name);
		else snprintf(inner, sizeof(inner), "%s", de->d_name);

		json = read_file(path);
		if (!json) continue;   /* "Monitoring object not added": its sub-directories are not visited */
		memset(&p, 0, sizeof(p));
		memset(&ctx, 0, sizeof(ctx));
		if (add_fwm) {
			ctx.fwm = calloc(MO_MAX_FWM, sizeof(*ctx.fwm));
			if (!ctx.fwm) { free(json); closedir(d); return XE_ENOMEM; }
		}
		p.s = json;
		p.cb = mo_cb;
		p.ctx = &ctx;
		jvalue(&p);
		free(json);
		if (p.failed || ctx.bad) { free(ctx.fwm); continue; }

		/* '/' in the name becomes '_' (monit-objects.c:162-167) */
		snprintf(name, sizeof(name), "%s", inner);
		for (i = 0; name[i]; i++) if (name[i] == '/') name[i] = '_';
		/* an object without a filter key has an empty program: matches all */
		idx = add(user, name, parent, ctx.seen ? ctx.filter : "");
		if (idx < 0) {
			closedir(d);
			free(ctx.fwm);
			snprintf(hs->err, sizeof(hs->err), "monitoring object '%s' rejected", name);
			return idx;
		}
		for (int k = 0; add_fwm && (k < ctx.nfwm); k++) {
			const char *fl[MO_MAX_FWM_FIELDS];
			int rc;
			for (int q = 0; q < ctx.fwm[k].nfields; q++) fl[q] = ctx.fwm[k].fields[q];
			if (!ctx.fwm[k].nfields) continue;
			rc = add_fwm(user, idx, ctx.fwm[k].name, fl, ctx.fwm[k].nfields, ctx.fwm[k].limit, ctx.fwm[k].time);
			if (rc < 0) { closedir(d); free(ctx.fwm); return rc; }
		}
		free(ctx.fwm);
		idx = load_mo_dir(hs, root, inner, idx, add, add_fwm, user);
		if (idx < 0) { closedir(d); return idx; }
	}
	closedir(d);
	return XE_OK;
}

int
xe_conf_load_mos(struct xe_host *hs, const char *mo_dir,
	int (*add)(void *user, const char *name, int parent, const char *filter),
	int (*add_fwm)(void *user, int mo, const char *name, const char *const *fields, int nfields, int limit, int time),
	void *user)
{
	return load_mo_dir(hs, mo_dir, "", -1, add, add_fwm, user);
}

/* ------------------------------------------------------------------ devices.conf */
#define MAX_MARKS 64
struct dev_acc
{
	int use_ip, use_id, rate, skip;
	uint32_t ip, id;
	char *marks[MAX_MARKS];
	int nmarks;
	int any;
};

struct dev_ctx
{
	long cur;
	struct dev_acc acc;
	int (*add)(void *, int, uint32_t, int, uint32_t, int, int, const char *const *, int);
	void *user;
	int rc;
};

static int
dev_flush(struct dev_ctx *c)
{
	int i, rc = XE_OK;
	if (c->acc.any) {
		rc = c->add(c->user, c->acc.use_ip, c->acc.ip, c->acc.use_id, c->acc.id, c->acc.rate,
			c->acc.skip, (const char *const *)c->acc.marks, c->acc.nmarks);
	}
	for (i = 0; i < c->acc.nmarks; i++) free(c->acc.marks[i]);
	memset(&c->acc, 0, sizeof(c->acc));
	return rc;
}

/* devices.c:60-139: an array of objects; keys ip, id, sampling-rate, mark
 * (string or array of strings), skip-unmarked */
static int
dev_cb(void *vctx, const struct jpath_item *path, int depth, int type, const char *val)
{
	struct dev_ctx *c = vctx;
	const char *key;

	if ((depth < 2) || !path[0].is_index || path[1].is_index) return 1;
	if (path[0].index != c->cur) {
		c->rc = dev_flush(c);
		if (c->rc != XE_OK) return 0;
		c->cur = path[0].index;
	}
	c->acc.any = 1;
	key = path[1].key;
	if (!strcmp(key, "ip")) {
		uint8_t a[4];
		if (inet_pton(AF_INET, val, a) != 1) { c->rc = XE_EPARSE; return 0; }
		c->acc.use_ip = 1;
		memcpy(&c->acc.ip, a, 4);
	} else if (!strcmp(key, "id")) {
		c->acc.use_id = 1;
		c->acc.id = (uint32_t)atoi(val);
	} else if (!strcmp(key, "sampling-rate")) {
		c->acc.rate = atoi(val);
	} else if (!strcmp(key, "mark")) {
		if (c->acc.nmarks >= MAX_MARKS) { c->rc = XE_EUNSUPPORTED; return 0; }
		c->acc.marks[c->acc.nmarks++] = strdup(val);
	} else if (!strcmp(key, "skip-unmarked")) {
		if (type == JV_TRUE) c->acc.skip = 1;
	}
	return 1;
}

int
xe_conf_load_devices(struct xe_host *hs, const char *path,
	int (*add)(void *user, int use_ip, uint32_t ip, int use_id, uint32_t id, int rate,
		int skip_unmarked, const char *const *marks, int nmarks), void *user)
{
	char *json = read_file(path);
	struct jparser p;
	struct dev_ctx c;

	if (!json) {
		snprintf(hs->err, sizeof(hs->err), "cannot read '%s'", path);
		return XE_EIO;
	}
	memset(&p, 0, sizeof(p));
	memset(&c, 0, sizeof(c));
	c.cur = -1;
	c.add = add;
	c.user = user;
	c.rc = XE_OK;
	p.s = json;
	p.cb = dev_cb;
	p.ctx = &c;
	jvalue(&p);
	free(json);
	if (c.rc == XE_OK) c.rc = dev_flush(&c);
	else dev_flush(&c);
	if ((c.rc == XE_OK) && p.failed) c.rc = XE_EPARSE;
	return c.rc;
}
