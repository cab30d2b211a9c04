/*
 * xe_api.cpp -- the C ABI of include/xe_gpu.h: handle life cycle, flattening
 * of the configuration into device arrays, batch upload / launch / download
 * and the re-creation of struct flow_info from the column output.
 *
 * There is no CPU evaluation path in this file or behind it: without a HIP
 * device xe_gpu_create() fails and nothing else can be called.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <arpa/inet.h>
#include "xe_kernels.h"

extern "C" int xe_launch_init_tables(void);

#define XE_DPOOL 64   /* device buffers a handle keeps for reuse (a batch owns up to 15) */

struct xe_gpu
{
	int device;
	struct xe_host hs;
	int committed;
	void *d_progs, *d_ops, *d_terms, *d_opnds, *d_trie, *d_lists, *d_devrules, *d_lookups, *d_fterms, *d_fprogs, *d_motab, *d_tneed, *d_cjgb, *d_cjtab, *d_cjderived, *d_cjcand, *d_rgroups, *d_rgdata, *d_mark_fprogs, *d_tmark;
	uint32_t nmarkprogs, ntmark;   /* mark filters in their fast-kernel form; terms that read the device mark */
	uint32_t *dev_canon;       /* per devices.conf entry: the first entry with the same mark filters and skip-unmarked */
	uint32_t nmo, mask_words, nprogs_total, nlookups;
	struct xe_list_dev *h_lists;
	struct xe_probe *h_probes;   /* host copy of the probe table (nlookups entries) */
	void *d_tmpls, *d_tents;
	size_t up_tmpls, up_tents;
	uint32_t nterms;           /* distinct terms of the object programs */
	uint32_t nfprogs;          /* objects the fast kernel evaluates one by one */
	uint32_t ntneed;           /* terms the fast kernel has to evaluate */
	uint32_t ncjgroups, ncjderived, nrgroups;
	/* fixed-window tables: they live as long as the handle, batches add to them */
	hipStream_t copy_stream;   /* datagrams travel on it while the host walks their headers */
	hipStream_t walk_stream;   /* the device's header walk over a page-locked datagram buffer runs beside the copy */
	/* overlap of consecutive batches on one handle: the datagram bytes travel on a stream of their own
	 * that prepare does not wait for */
	hipStream_t bulk_stream;
	int overlap;
	/* device buffers of freed batches, kept for the next ones: hipMalloc and
	 * hipFree of hundreds of MB cost about a millisecond each, and hipFree
	 * waits for the whole device, which serialises handles that share a GPU */
	struct { void *p; size_t cap; int busy; } dpool[XE_DPOOL];
	void *stage;               /* page-locked scratch the run list is built in (copies at link speed) */
	uint8_t *mirror;           /* page-locked copy of all outputs of the batch a flow cursor reads ... */
	size_t mirror_bytes;
	struct xe_batch *mirror_owner;   /* ... that batch, NULL while no cursor is open */
	size_t stage_cap;
	struct fwm_host *fwms; size_t nfwm;
	void *d_fwmdesc;
	int fwm_devip;             /* some table is keyed by dev-ip: the kernels note it per record */
	uint32_t lw;               /* 64-bit words of lookup bits per record: 1 or XE_LK_WORDS */
	uint32_t nplain_terms;     /* object terms [0, nplain_terms) have operands other than IP lists */
	size_t motab_words;        /* u64 words of the atom table and its non-empty-row masks (and the range tables behind them, from rg_lds on) */
	uint64_t *h_motab;         /* host copy while commit runs */
	uint32_t rg_lds;
	int fast_ok, want_dev;     /* want_dev: some object program reads dev_ip / dev_id */
	int reads_mark;            /* some object program reads dev_mark */
	/* service names of the 65536 ports (NULL = none given; an entry is NULL or
	 * "" for a port without a name): what portstr() / ppstr() compare against */
	char **port_names;
	uint32_t port_named_off;   /* word offset in the list tables of the "port has a name" set, 0 = none */
	/* header walk on the device (xe_walk.hip): bit 0 sFlow, bit 1 Netflow */
	int device_walk;
	int walk_timing;           /* record HIP events around the walk's launches of every prepare */
	int list_root_bits;        /* width of the list tables' roots: 0 = chosen per table, 16, 24 */
	/* read from the environment once, when the handle is made */
	int dbg;                   /* XE_DEBUG: timers of prepare on stderr */
	int diag_force_general, diag_fast_r, diag_counter_copies;   /* diagnostic builds (-DXE_DIAG) only, else 0 */
	void *d_devkeys;           /* struct xe_devkey per devices.conf entry */
	void *d_devhash;           /* struct xe_devhent: the look-up table of the device walk (NULL: few entries) */
	uint32_t devhash_mask;
	void *d_thash, *d_twalk;   /* the template store and the device templates as the walk reads them */
	void *d_traw;              /* the stored templates' bytes, for the walk's "same template again?" test */
	uint32_t thash_mask;
	size_t twalk_n;
	uint64_t walk_epoch;       /* hs.tmpl_epoch the two tables were built from (+ 1) */
	void *wstage;              /* small page-locked buffer for counts coming back */
	/* page-locked scratch of one prepare call (per-datagram arrays the walk
	 * kernels read in place, small tables on their way to the device): handed
	 * out from the front, everything is free again when the next call begins.
	 * Copies out of ordinary memory block the calling thread inside the runtime,
	 * and with it the other handles' threads */
	uint8_t *pin;
	size_t pin_cap, pin_used;
	char err[512];
};
#define XE_WSTAGE_BYTES (64 * 1024)

struct fwm_host
{
	struct xe_fwm_dev dev;     /* with the device pointers */
	size_t slots, max_keys;
	uint32_t keybytes;         /* of a row as the reference keeps it */
	uint32_t devkeybytes;      /* of a row in the device table (function fields in their short forms) */
	uint8_t kfunc[XE_FWM_MAX_KEY_FIELDS];
	uint16_t krefsize[XE_FWM_MAX_KEY_FIELDS];
	int texts;                 /* some key field is a text made at read-out */
	char name[256];            /* the entry's "name" in mo.conf, when it came from there */
	int limit, time;           /* "limit" (rows exported, 0 = all) and "time" (window seconds) of the entry */
	/* the fields in the order of the entry: what the export sorts by */
	int nfields;
	struct { uint8_t aggr, descending; uint16_t size; } order[XE_FWM_MAX_KEY_FIELDS + XE_FWM_MAX_VALS];
};

#define XE_WALK_EVENTS 16
/* what the header walk on the device leaves for one batch */
struct walk_dev
{
	int mode;                  /* bit 0: the sFlow datagrams were walked on the device, bit 1: Netflow ones of the usual shape */
	hipStream_t st;            /* what the walk is enqueued on */
	const uint8_t *hdr_src;    /* where the Netflow walk reads headers: the batch on the device, or the caller's page-locked buffer */
	size_t n;                  /* datagrams */
	void *arena, *tiles;       /* device allocations the pointers below are carved from */
	uint64_t *d_offs;
	uint32_t *d_src;
	uint8_t *d_kinds;
	uint32_t *d_sfcnt, *d_sfbase;
	struct xe_sfdg *d_sfdg;    /* per datagram: what the sFlow decode kernel needs of it */
	struct xe_sfagent *d_sfag; /* per datagram: agent address and sub-agent id, for the host */
	void *d_rec2dg;            /* struct xe_sfrd per sFlow record (an allocation of its own: sized after the count) */
	unsigned long long *d_total;   /* [0] sFlow records, [1] Netflow records */
	void *d_temp; size_t temp_bytes;
	struct xe_dgrun *d_dgruns;
	uint8_t *d_walked;
	uint32_t *d_hostfirst, *d_recbase;
	int sums_valid;            /* the shape kernel's per-template sums hold (no datagram was left to the host) */
	struct xe_wgroup *d_groups;
	struct xe_frun *d_fruns2;
	struct xe_ftile *d_ftiles2;
	uint32_t nfruns2, ntiles2, maxchunk2;
	uint32_t nsets;   /* the most runs a datagram of this batch has */
	size_t nrec2;              /* records in those runs */
	/* host side */
	uint8_t *kinds2;           /* the kinds with XE_KIND_ON_DEVICE on what the host walk passes by */
	uint8_t *h_walked;         /* per datagram: its Netflow records were listed on the device */
	uint32_t *h_src;           /* exporter addresses (the caller's array is gone after prepare) */
	/* instrumentation: event pairs (start, stop) around every stretch of walk
	 * launches that is enqueued without the host waiting in between */
	hipEvent_t wev[XE_WALK_EVENTS];
	int nwev;
	/* fetched when somebody asks for them */
	uint32_t *h_recbase, *h_sfbase;
	struct xe_dgrun *h_dgruns;
};

struct xe_batch
{
	struct xe_gpu *h;
	struct xe_plan plan;
	size_t ntot, payload_bytes;
	void *d_pkt, *d_sets, *d_exps, *d_sfrecs, *d_out;
	void *d_fruns, *d_ftiles, *d_ssets, *d_swstart, *d_fplans, *d_fops, *d_fflds;
	size_t out_bytes, masks_off, nparsed_off, marks_off, counters_off, sfhas_off, sfrate_off, rates_off, devips_off;
	struct xe_kparams kp;
	hipStream_t last_stream;
	int ran;
	hipEvent_t ev_bulk;        /* overlap mode: the datagram bytes have arrived */
	hipEvent_t ev_ran;         /* overlap mode: this batch's last run has finished (the stream may carry the next batch already) */
	int sent_ahead;            /* its datagram bytes were sent with xe_gpu_upload_begin() */
	int timing;              /* record events around each kernel launch */
	hipEvent_t ev[6];
	hipEvent_t ev_fwm[2];      /* around the fixed-window kernel */
	struct walk_dev w;
};

#define ALIGN_UP(x, a) (((x) + (a) - 1) / (a) * (a))

static int
hip_fail(struct xe_gpu *h, hipError_t e, const char *what)
{
	if (h) snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
	return XE_EHIP;
}

#define HIPCHK(H, CALL) do { hipError_t e_ = (CALL); if (e_ != hipSuccess) return hip_fail((H), e_, #CALL); } while (0)

/* instrumentation: marks the start or the end of a stretch of walk launches */
static void
walk_mark(struct xe_gpu *h, struct walk_dev *w, hipStream_t st)
{
	if (!h->walk_timing || (w->nwev >= XE_WALK_EVENTS)) return;
	if (hipEventCreate(&w->wev[w->nwev]) != hipSuccess) { (void)hipGetLastError(); return; }
	if (hipEventRecord(w->wev[w->nwev], st) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(w->wev[w->nwev]); return; }
	w->nwev++;
}

extern "C" const char *
xe_gpu_strerror(int code)
{
	switch (code) {
		case XE_OK: return "ok";
		case XE_EINVAL: return "invalid argument";
		case XE_ENOMEM: return "out of host memory";
		case XE_EHIP: return "HIP runtime error";
		case XE_EPARSE: return "parse error";
		case XE_EUNSUPPORTED: return "not evaluable on the device";
		case XE_ESTATE: return "call out of order";
		case XE_EIO: return "cannot read file or directory";
		default: return "unknown error";
	}
}

extern "C" const char *
xe_gpu_last_error(const xe_gpu *h)
{
	return h ? h->err : "";
}

extern "C" const char *
xe_gpu_build_arch(void)
{
	return XE_BUILD_ARCH;
}

static void *
pinned_alloc(size_t bytes)
{
	void *p = NULL;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		return NULL;
	}
	return p;
}

static void
pinned_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

extern "C" int
xe_gpu_create(xe_gpu **out, int device)
{
	int ndev = 0;
	xe_gpu *h;

	if (!out) return XE_EINVAL;
	*out = NULL;
	if ((hipGetDeviceCount(&ndev) != hipSuccess) || (device < 0) || (device >= ndev)) return XE_EHIP;
	if (hipSetDevice(device) != hipSuccess) return XE_EHIP;
	h = (xe_gpu *)calloc(1, sizeof(*h));
	if (!h) return XE_ENOMEM;
	h->device = device;
	xe_host_init(&h->hs);
	h->hs.big_alloc = pinned_alloc;
	h->hs.big_free = pinned_free;
	if (xe_launch_init_tables() != XE_OK) {
		free(h);
		return XE_EHIP;
	}
	if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) {
		(void)hipGetLastError();
		xe_host_free(&h->hs);
		free(h);
		return XE_EHIP;
	}
	h->wstage = pinned_alloc(XE_WSTAGE_BYTES);
	if (h->wstage && (hipStreamCreateWithFlags(&h->walk_stream, hipStreamNonBlocking) != hipSuccess)) {
		(void)hipGetLastError();
		pinned_free(h->wstage);
		h->wstage = NULL;
	}
	if (!h->wstage) {
		(void)hipStreamDestroy(h->copy_stream);
		xe_host_free(&h->hs);
		free(h);
		return XE_EHIP;
	}
	h->device_walk = 3;
	if (getenv("XE_DEVICE_WALK")) h->device_walk = atoi(getenv("XE_DEVICE_WALK")) & 3;
	h->dbg = getenv("XE_DEBUG") != NULL;
#ifdef XE_DIAG
	h->diag_force_general = getenv("XE_FORCE_GENERAL") != NULL;
	if (getenv("XE_FAST_R")) h->diag_fast_r = atoi(getenv("XE_FAST_R"));
	if (getenv("XE_COUNTER_COPIES")) h->diag_counter_copies = atoi(getenv("XE_COUNTER_COPIES"));
#endif
	*out = h;
	return XE_OK;
}

static void
free_config_dev(xe_gpu *h)
{
	void **p[] = { &h->d_progs, &h->d_ops, &h->d_terms, &h->d_opnds, &h->d_trie, &h->d_lists, &h->d_devrules,
		&h->d_lookups, &h->d_fterms, &h->d_fprogs, &h->d_motab, &h->d_tneed,
		&h->d_cjgb, &h->d_cjtab, &h->d_cjderived, &h->d_cjcand, &h->d_rgroups, &h->d_rgdata, &h->d_mark_fprogs, &h->d_tmark };
	for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); i++) {
		if (*p[i]) (void)hipFree(*p[i]);
		*p[i] = NULL;
	}
	free(h->h_lists);
	h->h_lists = NULL;
	free(h->h_probes);
	h->h_probes = NULL;
	free(h->dev_canon);
	h->dev_canon = NULL;
}

extern "C" void
xe_gpu_destroy(xe_gpu *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	free_config_dev(h);
	if (h->d_tmpls) (void)hipFree(h->d_tmpls);
	if (h->d_tents) (void)hipFree(h->d_tents);
	for (size_t i = 0; i < h->nfwm; i++) {
		if (h->fwms[i].dev.slots) (void)hipFree(h->fwms[i].dev.slots);
		if (h->fwms[i].dev.vals) (void)hipFree(h->fwms[i].dev.vals);
		if (h->fwms[i].dev.dropped) (void)hipFree(h->fwms[i].dev.dropped);
		if (h->fwms[i].dev.dl) (void)hipFree((void *)h->fwms[i].dev.dl);
	}
	free(h->fwms);
	if (h->d_fwmdesc) (void)hipFree(h->d_fwmdesc);
	if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
	if (h->bulk_stream) (void)hipStreamDestroy(h->bulk_stream);
	if (h->walk_stream) (void)hipStreamDestroy(h->walk_stream);
	for (int i = 0; i < XE_DPOOL; i++) if (h->dpool[i].p) (void)hipFree(h->dpool[i].p);
	if (h->stage) (void)hipHostFree(h->stage);
	if (h->mirror) (void)hipHostFree(h->mirror);
	free(h->h_motab);
	pinned_free(h->wstage);
	pinned_free(h->pin);
	if (h->d_devkeys) (void)hipFree(h->d_devkeys);
	if (h->d_devhash) (void)hipFree(h->d_devhash);
	if (h->d_thash) (void)hipFree(h->d_thash);
	if (h->d_twalk) (void)hipFree(h->d_twalk);
	if (h->d_traw) (void)hipFree(h->d_traw);
	if (h->port_names) {
		for (int i = 0; i < 65536; i++) free(h->port_names[i]);
		free(h->port_names);
	}
	xe_host_free(&h->hs);
	free(h);
}

/* ------------------------------------------------------------------ configuration */
/* the lists at the end of the registry that the last commit made from the
 * geo / AS databases (and the empty ones it put in front of them) */
static void
drop_made_lists(xe_gpu *h)
{
	struct xe_iplists *reg = &h->hs.lists;
	while (h->hs.nlists_made && reg->n) {
		reg->n--;
		free(reg->l[reg->n].n4);
		free(reg->l[reg->n].n6);
		h->hs.nlists_made--;
	}
	h->hs.nlists_made = 0;
	h->hs.nlists_fn = 0;
}

extern "C" int
xe_gpu_load_iplists(xe_gpu *h, const char *dir)
{
	int rc;
	if (!h || !dir) return XE_EINVAL;
	drop_made_lists(h);
	rc = xe_iplist_load_dir(&h->hs.lists, dir);
	if (rc != XE_OK) snprintf(h->err, sizeof(h->err), "cannot load IP lists from '%s'", dir);
	h->committed = 0;
	return rc;
}

extern "C" int
xe_gpu_add_iplist(xe_gpu *h, const char *name, const char *text, size_t len)
{
	if (!h || !name || (!text && len)) return XE_EINVAL;
	h->committed = 0;
	drop_made_lists(h);
	return xe_iplist_load_text(&h->hs.lists, name, text, len);
}

/* deepest stack the bit-stack evaluator would need */
static size_t
prog_depth(const struct xe_filter *f)
{
	size_t sp = 0, mx = 0;
	for (size_t i = 0; i < f->nops; i++) {
		uint32_t op = f->ops[i] & 3;
		if (op == XE_OP_BASIC) { sp++; if (sp > mx) mx = sp; }
		else if ((op == XE_OP_AND) || (op == XE_OP_OR)) { if (sp >= 2) sp--; }
	}
	return mx;
}

/* host_only == NULL: the filter must be evaluable on the device (device mark
 * filters).  Else *host_only says whether the object is left to the caller's
 * own filter_match() (why[128] then holds the reason). */
static int
compile_checked(xe_gpu *h, const char *text, struct xe_filter *f, int *host_only, char *why)
{
	char perr[600];
	const char *reason = NULL;
	int rc = xe_filter_compile(&h->hs.lists, text ? text : "", f, perr, sizeof(perr));

	if (rc != XE_OK) {
		snprintf(h->err, sizeof(h->err), "filter '%.200s': %.250s", text ? text : "", perr);
		return rc;
	}
	if (!f->device_ok) reason = f->why;
	else if (f->needs_port_names && !h->port_names) reason = "portstr() / ppstr() without a port-name table (xe_gpu_set_port_names)";
	else if (f->needs_geodb) {
		for (int k = 0; k < 4; k++) {
			if (((f->needs_geodb >> k) & 1) && !h->hs.geodb[k >> 1][k & 1].img) {
				reason = "geo / AS function without its database (xe_gpu_set_geodb)";
			}
		}
	}
	else if (prog_depth(f) > 64) reason = "expression nests deeper than 64";
	if (host_only) *host_only = reason != NULL;
	if (reason) {
		if (host_only) {
			snprintf(why, 128, "%s", reason);
			return XE_OK;
		}
		snprintf(h->err, sizeof(h->err), "filter '%.200s': %s", text ? text : "", reason);
		xe_filter_free(f);
		return XE_EUNSUPPORTED;
	}
	return XE_OK;
}

extern "C" int
xe_gpu_add_mo(xe_gpu *h, const char *name, int parent, const char *filter)
{
	struct xe_mo *nm;
	struct xe_filter f;
	int rc;

	int host_only = 0;
	char why[128] = "";

	if (!h || (parent < -1) || (parent >= (int)h->hs.nmos)) return XE_EINVAL;
	rc = compile_checked(h, filter, &f, &host_only, why);
	if (rc != XE_OK) return rc;
	nm = (struct xe_mo *)realloc(h->hs.mos, (h->hs.nmos + 1) * sizeof(*nm));
	if (!nm) { xe_filter_free(&f); return XE_ENOMEM; }
	h->hs.mos = nm;
	nm = &h->hs.mos[h->hs.nmos];
	memset(nm, 0, sizeof(*nm));
	snprintf(nm->name, sizeof(nm->name), "%s", name ? name : "");
	nm->parent = parent;
	nm->filter = f;
	nm->host_only = host_only;
	memcpy(nm->why, why, sizeof(nm->why));
	h->committed = 0;
	return (int)h->hs.nmos++;
}

extern "C" int
xe_gpu_mo_host_only(const xe_gpu *h, int mo)
{
	if (!h || (mo < 0) || ((size_t)mo >= h->hs.nmos)) return XE_EINVAL;
	return h->hs.mos[mo].host_only;
}

extern "C" const char *
xe_gpu_mo_host_only_reason(const xe_gpu *h, int mo)
{
	if (!h || (mo < 0) || ((size_t)mo >= h->hs.nmos) || !h->hs.mos[mo].host_only) return "";
	return h->hs.mos[mo].why;
}

/* 1 when object `mo` or one of its ancestors is host-only: its match bit and
 * counters are those of a candidate, not of a final match */
static int
mo_host_gated(const xe_gpu *h, int mo)
{
	for (; mo >= 0; mo = h->hs.mos[mo].parent) if (h->hs.mos[mo].host_only) return 1;
	return 0;
}

extern "C" int
xe_gpu_mo_host_gated(const xe_gpu *h, int mo)
{
	if (!h || (mo < 0) || ((size_t)mo >= h->hs.nmos)) return XE_EINVAL;
	return mo_host_gated(h, mo);
}

extern "C" int
xe_gpu_set_port_names(xe_gpu *h, const char *const *names)
{
	if (!h) return XE_EINVAL;
	if (h->port_names) {
		for (int i = 0; i < 65536; i++) free(h->port_names[i]);
		free(h->port_names);
		h->port_names = NULL;
	}
	h->committed = 0;
	if (!names) return XE_OK;
	h->port_names = (char **)calloc(65536, sizeof(char *));
	if (!h->port_names) return XE_ENOMEM;
	for (int i = 0; i < 65536; i++) {
		if (!names[i] || !names[i][0]) continue;
		h->port_names[i] = strdup(names[i]);
		if (!h->port_names[i]) return XE_ENOMEM;
	}
	return XE_OK;
}

extern "C" int xe_gpu_mo_count(const xe_gpu *h) { return h ? (int)h->hs.nmos : XE_EINVAL; }

extern "C" int
xe_gpu_set_geodb(xe_gpu *h, int kind, int family, const void *image, size_t bytes)
{
	int rc;
	if (!h) return XE_EINVAL;
	if ((h->hs.nmos || h->hs.ndevs) && !((kind >= 0) && (kind < 2) && ((family == 4) || (family == 6))
		&& h->hs.geodb[kind][family == 6].img && image && bytes)) {
		/* which objects the device decides was settled when they were added: a
		 * database may be replaced by a newer image afterwards (the reference's
		 * reload, geoip.c:131-178), not added or dropped */
		snprintf(h->err, sizeof(h->err), "the geo / AS databases are set before the objects and devices are added "
			"(afterwards a loaded database can only be replaced by a new image)");
		return XE_ESTATE;
	}
	rc = xe_geodb_set(&h->hs, kind, family, image, bytes, h->err, sizeof(h->err));
	if (rc == XE_OK) h->committed = 0;
	return rc;
}

extern "C" int
xe_gpu_load_geodb(xe_gpu *h, const char *dir)
{
	static const char *const names[4] = { "geo4", "geo6", "as4", "as6" };   /* geoip.c:146-149 */
	int loaded = 0;

	if (!h || !dir) return XE_EINVAL;
	for (int k = 0; k < 4; k++) {
		char path[4200];
		FILE *f;
		long len;
		void *buf;
		int rc;
		snprintf(path, sizeof(path), "%.4000s/%s.db", dir, names[k]);
		f = fopen(path, "rb");
		if (!f) continue;             /* the reference runs without a database it cannot open */
		if ((fseek(f, 0, SEEK_END) != 0) || ((len = ftell(f)) < 0) || (fseek(f, 0, SEEK_SET) != 0)) { fclose(f); return XE_EIO; }
		buf = malloc(len ? (size_t)len : 1);
		if (!buf) { fclose(f); return XE_ENOMEM; }
		if (fread(buf, 1, (size_t)len, f) != (size_t)len) { free(buf); fclose(f); return XE_EIO; }
		fclose(f);
		rc = xe_gpu_set_geodb(h, k >> 1, (k & 1) ? 6 : 4, buf, (size_t)len);
		free(buf);
		if (rc != XE_OK) return rc;
		loaded++;
	}
	return loaded;
}

extern "C" int
xe_gpu_reload_geodb(xe_gpu *h, const char *dir)
{
	int n, rc;
	if (!h || !dir) return XE_EINVAL;
	n = xe_gpu_load_geodb(h, dir);
	if (n < 0) return n;
	rc = xe_gpu_commit(h);
	return (rc == XE_OK) ? n : rc;
}

extern "C" const char *
xe_gpu_mo_name(const xe_gpu *h, int mo)
{
	return (h && (mo >= 0) && ((size_t)mo < h->hs.nmos)) ? h->hs.mos[mo].name : NULL;
}

extern "C" int
xe_gpu_mo_parent(const xe_gpu *h, int mo)
{
	return (h && (mo >= 0) && ((size_t)mo < h->hs.nmos)) ? h->hs.mos[mo].parent : XE_EINVAL;
}

extern "C" int
xe_gpu_mo_dump(const xe_gpu *h, int mo, char *out, size_t outlen)
{
	if (!h || (mo < 0) || ((size_t)mo >= h->hs.nmos) || !out) return XE_EINVAL;
	return xe_filter_dump_text(&h->hs.lists, &h->hs.mos[mo].filter, out, outlen);
}

extern "C" int
xe_filter_check(const xe_gpu *h, const char *filter, char *dump, size_t dumplen, char *err, size_t errlen)
{
	struct xe_filter f;
	int rc = xe_filter_compile(h ? &h->hs.lists : NULL, filter ? filter : "", &f, err, errlen);
	if (rc != XE_OK) return rc;
	if (dump && dumplen) xe_filter_dump_text(h ? &h->hs.lists : NULL, &f, dump, dumplen);
	rc = f.device_ok ? XE_OK : XE_EUNSUPPORTED;
	if ((rc != XE_OK) && err && errlen) snprintf(err, errlen, "%s", f.why);
	xe_filter_free(&f);
	return rc;
}

extern "C" int
xe_gpu_add_device(xe_gpu *h, int use_ip, uint32_t ip, int use_id, uint32_t id, int sampling_rate,
	int skip_unmarked, const char *const *marks, int nmarks)
{
	struct xe_device d, *nd;

	if (!h || (nmarks < 0) || (nmarks && !marks)) return XE_EINVAL;
	memset(&d, 0, sizeof(d));
	d.use_ip = use_ip;
	d.ip = ip;
	d.use_id = use_id;
	d.id = htonl(id);   /* devices.c:100 */
	d.sampling_rate = sampling_rate;
	d.skip_unmarked = skip_unmarked;
	if (nmarks) {
		d.marks = (struct xe_filter *)calloc((size_t)nmarks, sizeof(struct xe_filter));
		if (!d.marks) return XE_ENOMEM;
		for (int i = 0; i < nmarks; i++) {
			int rc = compile_checked(h, marks[i], &d.marks[i], NULL, NULL);
			if (rc != XE_OK) {
				for (int j = 0; j < i; j++) xe_filter_free(&d.marks[j]);
				free(d.marks);
				return rc;
			}
			d.nmarks++;
		}
	}
	nd = (struct xe_device *)realloc(h->hs.devs, (h->hs.ndevs + 1) * sizeof(*nd));
	if (!nd) return XE_ENOMEM;
	h->hs.devs = nd;
	h->hs.devs[h->hs.ndevs++] = d;
	h->committed = 0;
	return XE_OK;
}

static int
add_mo_cb(void *user, const char *name, int parent, const char *filter)
{
	return xe_gpu_add_mo((xe_gpu *)user, name, parent, filter);
}

static int
add_dev_cb(void *user, int use_ip, uint32_t ip, int use_id, uint32_t id, int rate, int skip,
	const char *const *marks, int nmarks)
{
	return xe_gpu_add_device((xe_gpu *)user, use_ip, ip, use_id, id, rate, skip, marks, nmarks);
}

extern "C" int
xe_gpu_load_mos(xe_gpu *h, const char *mo_dir)
{
	if (!h || !mo_dir) return XE_EINVAL;
	return xe_conf_load_mos(&h->hs, mo_dir, add_mo_cb, NULL, h);
}

struct fwm_load_ctx { xe_gpu *h; size_t max_keys; int added, skipped; };

static int
add_mo_cb2(void *user, const char *name, int parent, const char *filter)
{
	return add_mo_cb(((struct fwm_load_ctx *)user)->h, name, parent, filter);
}

static int
add_fwm_cb(void *user, int mo, const char *name, const char *const *fields, int nfields, int limit, int time)
{
	struct fwm_load_ctx *c = (struct fwm_load_ctx *)user;
	int rc = xe_gpu_add_fwm(c->h, mo, fields, nfields, c->max_keys);
	if (rc == XE_EUNSUPPORTED) { c->skipped++; return 0; }   /* stays on the reference's CPU path */
	if (rc < 0) return rc;
	snprintf(c->h->fwms[rc].name, sizeof(c->h->fwms[rc].name), "%s", name);
	c->h->fwms[rc].limit = limit;
	c->h->fwms[rc].time = time;
	c->added++;
	return 0;
}

extern "C" int
xe_gpu_load_mos_with_tables(xe_gpu *h, const char *mo_dir, size_t max_keys, int *skipped)
{
	struct fwm_load_ctx c;
	int rc;
	if (!h || !mo_dir) return XE_EINVAL;
	c.h = h; c.max_keys = max_keys; c.added = 0; c.skipped = 0;
	rc = xe_conf_load_mos(&h->hs, mo_dir, add_mo_cb2, add_fwm_cb, &c);
	if (skipped) *skipped = c.skipped;
	return (rc < 0) ? rc : c.added;
}

extern "C" int
xe_gpu_fwm_count(const xe_gpu *h)
{
	return h ? (int)h->nfwm : 0;
}

extern "C" int
xe_gpu_fwm_object(const xe_gpu *h, int fwm)
{
	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	return (int)h->fwms[fwm].dev.mo;
}

extern "C" const char *
xe_gpu_fwm_name(const xe_gpu *h, int fwm)
{
	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return "";
	return h->fwms[fwm].name;
}

extern "C" int
xe_gpu_load_devices(xe_gpu *h, const char *path)
{
	if (!h || !path) return XE_EINVAL;
	return xe_conf_load_devices(&h->hs, path, add_dev_cb, h);
}

static uint64_t
be_bytes(const uint8_t *b, int n)
{
	uint64_t v = 0;
	for (int i = 0; i < n; i++) v = (v << 8) | b[i];
	return v;
}

static int
upload(xe_gpu *h, void **dst, const void *src, size_t bytes)
{
	*dst = NULL;
	if (!bytes) bytes = 16;
	HIPCHK(h, hipMalloc(dst, ALIGN_UP(bytes, 256)));
	if (src) HIPCHK(h, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
	return XE_OK;
}

/* a device buffer for a batch: one that an earlier batch of the handle gave
 * back if it is large enough and not more than twice too large, else a new one */
static int
dev_get(xe_gpu *h, void **dst, size_t bytes)
{
	int best = -1, empty = -1, victim = -1;

	*dst = NULL;
	bytes = ALIGN_UP(bytes ? bytes : 16, 256);
	for (int i = 0; i < XE_DPOOL; i++) {
		if (!h->dpool[i].p) { if (empty < 0) empty = i; continue; }
		if (h->dpool[i].busy) continue;
		if ((h->dpool[i].cap >= bytes) && (h->dpool[i].cap <= 2 * bytes + 65536)
			&& ((best < 0) || (h->dpool[i].cap < h->dpool[best].cap))) best = i;
		if ((victim < 0) || (h->dpool[i].cap < h->dpool[victim].cap)) victim = i;
	}
	if (best >= 0) {
		h->dpool[best].busy = 1;
		*dst = h->dpool[best].p;
		return XE_OK;
	}
	if ((empty < 0) && (victim >= 0)) {
		(void)hipFree(h->dpool[victim].p);
		h->dpool[victim].p = NULL;
		empty = victim;
	}
	if (hipMalloc(dst, bytes) != hipSuccess) {
		/* out of device memory: what the handle keeps for reuse goes first */
		(void)hipGetLastError();
		for (int i = 0; i < XE_DPOOL; i++) {
			if (h->dpool[i].p && !h->dpool[i].busy) {
				(void)hipFree(h->dpool[i].p);
				h->dpool[i].p = NULL;
				if (empty < 0) empty = i;
			}
		}
		HIPCHK(h, hipMalloc(dst, bytes));
	}
	if (empty >= 0) {
		h->dpool[empty].p = *dst;
		h->dpool[empty].cap = bytes;
		h->dpool[empty].busy = 1;
	}
	return XE_OK;
}

static void
dev_put(xe_gpu *h, void *p)
{
	if (!p) return;
	for (int i = 0; i < XE_DPOOL; i++) {
		if (h->dpool[i].p == p) { h->dpool[i].busy = 0; return; }
	}
	(void)hipFree(p);   /* the table was full when it was made */
}

static int
upload_pooled(xe_gpu *h, void **dst, const void *src, size_t bytes)
{
	int rc = dev_get(h, dst, bytes);
	if (rc != XE_OK) return rc;
	/* on the handle's copy stream (prepare waits for it before it returns); the
	 * source may be reused as soon as the call returns: pageable memory is
	 * staged by the runtime before it does */
	if (src && bytes) HIPCHK(h, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, h->copy_stream));
	return XE_OK;
}

/* texts of the portstr() operands of a configuration, each once */
struct strpool { struct xe_blob chars, offs; };

static int32_t
strpool_intern(struct strpool *sp, const char *str)
{
	const uint32_t *off = (const uint32_t *)sp->offs.p;
	uint32_t at = (uint32_t)sp->chars.n;

	for (size_t i = 0; i < sp->offs.n; i++) {
		if (!strcmp((const char *)sp->chars.p + off[i], str)) return (int32_t)i;
	}
	for (size_t k = 0; k <= strlen(str); k++) if (!xe_blob_push(&sp->chars, &str[k], 1)) return -1;
	if (!xe_blob_push(&sp->offs, &at, 4)) return -1;
	return (int32_t)(sp->offs.n - 1);
}

/* what the interning of terms needs besides the arrays it fills */
struct intern_ctx
{
	xe_gpu *h;
	struct xe_blob *terms, *opnds, *lookups;
	struct strpool *strs;
};

static const char *
port_name(const xe_gpu *h, uint32_t port)
{
	const char *n = h->port_names ? h->port_names[port & 0xFFFF] : NULL;
	return n ? n : "";
}

/* port_to_str(), utils.c:28-37 */
static void
port_text(const xe_gpu *h, uint32_t port, char out[96])
{
	const char *n = port_name(h, port);
	if (n[0]) snprintf(out, 96, "%.60s (%d)", n, (int)port);
	else snprintf(out, 96, "%d", (int)port);
}

/* ports_pair_to_str(), utils.c:39-77 */
static void
port_pair_text(const xe_gpu *h, uint32_t p1, uint32_t p2, char out[192])
{
	const char *s1 = port_name(h, p1), *s2 = port_name(h, p2);

	out[0] = '\0';
	if (s1[0] && s2[0]) { snprintf(out, 192, "%.60s(%d) -> %.60s(%d)", s1, (int)p1, s2, (int)p2); return; }
	if (p1 == p2) { snprintf(out, 192, "%d -> %d", (int)p1, (int)p2); return; }
	if (p1 < p2) {
		if (!s1[0] && !s2[0]) snprintf(out, 192, "%d ->", (int)p1);
		else if (!s1[0]) snprintf(out, 192, "%d -> %.60s(%d)", (int)p1, s2, (int)p2);
		else snprintf(out, 192, "%.60s(%d) ->", s1, (int)p1);
	} else {
		if (!s1[0] && !s2[0]) snprintf(out, 192, "-> %d", (int)p2);
		else if (!s1[0]) snprintf(out, 192, "-> %.60s(%d)", s2, (int)p2);
		else snprintf(out, 192, "%.60s(%d) -> %d", s1, (int)p1, (int)p2);
	}
}

/* ppstr() 'text' as conditions on the two port values.  The text of a pair
 * (ports_pair_to_str) names both ports only when both have a service name, or
 * the named / smaller one otherwise, so an operand is
 *   kind 1: port1 == X and port2 == Y
 *   kind 2: port1 == X, port2 > X and port2 has no name    ("X ->", "name(X) ->")
 *   kind 3: port2 == Y, port1 > Y and port1 has no name    ("-> Y", "-> name(Y)")
 * or several of them.  Every candidate is checked by formatting it and
 * comparing like the reference (strcasecmp, filter.c:894-901). */
static int
ppstr_lower(const xe_gpu *h, const char *text, struct xe_blob *out)
{
	char buf[192];
	struct xe_dopnd d;
	/* an unnamed port above p exists? (kinds 2 and 3 need one to ever match; kept anyway) */
	memset(&d, 0, sizeof(d));
#define PP_PUSH(K, X, Y) do { d.list = (K); d.low = (int32_t)(X); d.high = (int32_t)(Y); \
	if (!xe_blob_push(out, &d, sizeof(d))) return XE_ENOMEM; } while (0)
	for (uint32_t p1 = 0; p1 < 65536; p1++) {
		const int named1 = port_name(h, p1)[0] != 0;
		size_t plen;
		/* kind 2: the text of (p1, an unnamed port above it) does not depend on that port */
		snprintf(buf, sizeof(buf), named1 ? "%.60s(%d) ->" : "%.0s%d ->", port_name(h, p1), (int)p1);
		if (!strcasecmp(text, buf)) PP_PUSH(2, p1, 0);
		/* kind 3 with Y = p1 */
		snprintf(buf, sizeof(buf), named1 ? "-> %.60s(%d)" : "-> %.0s%d", port_name(h, p1), (int)p1);
		if (!strcasecmp(text, buf)) PP_PUSH(3, 0, p1);
		/* exact pairs: every text of a pair that names p1 starts with p1's part and " -> " */
		snprintf(buf, sizeof(buf), named1 ? "%.60s(%d) -> " : "%.0s%d -> ", port_name(h, p1), (int)p1);
		plen = strlen(buf);
		if (strncasecmp(text, buf, plen) != 0) continue;
		for (uint32_t p2 = 0; p2 < 65536; p2++) {
			const int named2 = port_name(h, p2)[0] != 0;
			/* pairs whose text carries both ports */
			if (!((named1 && named2) || (p1 == p2) || (!named1 && named2 && (p1 < p2)) || (named1 && !named2 && (p1 > p2)))) continue;
			port_pair_text(h, p1, p2, buf);
			if (!strcasecmp(text, buf)) PP_PUSH(1, p1, p2);
		}
	}
#undef PP_PUSH
	return XE_OK;
}

/* device form of one operand */
/* the text of a TCP flags byte: the names of the set bits, most significant
 * first, joined with '+' (the reference's table utils-data.inc:4-261) */
static void
tcp_flags_text(unsigned tf, char out[40])
{
	static const char *const names[8] = { "CWR", "ECE", "URG", "ACK", "PSH", "RST", "SYN", "FIN" };

	out[0] = '\0';
	for (int b = 0; b < 8; b++) {
		if (!(tf & (0x80u >> b))) continue;
		if (out[0]) strcat(out, "+");
		strcat(out, names[b]);
	}
}

static struct xe_dopnd
make_dopnd(const struct xe_term *t, const struct xe_opnd *o)
{
	const uint8_t *b = (const uint8_t *)&o->w[0];
	struct xe_dopnd d;

	memset(&d, 0, sizeof(d));
	d.is_list = o->is_list;
	if (t->func == XE_FN_TFSTR) {
		/* strcasecmp() against the text of the flags byte (filter.c:849-855)
		 * is a set of byte values: 256 bits in net_hi, net_lo, mask_hi, mask_lo */
		uint64_t set[4] = { 0, 0, 0, 0 };
		for (unsigned v = 0; v < 256; v++) {
			char txt[40];
			tcp_flags_text(v, txt);
			if (strcasecmp((const char *)b, txt) == 0) set[v >> 6] |= 1ull << (v & 63);
		}
		d.net_hi = set[0]; d.net_lo = set[1]; d.mask_hi = set[2]; d.mask_lo = set[3];
		return d;
	}
	switch (t->type) {
		case XE_BT_ADDR4:
			if (o->is_list) d.list = o->w[0];
			else { d.net_lo = be_bytes(b, 4); d.mask_lo = be_bytes(b + 4, 4); }
			break;
		case XE_BT_ADDR6:
			if (o->is_list) d.list = o->w[0];
			else {
				d.net_hi = be_bytes(b, 8); d.net_lo = be_bytes(b + 8, 8);
				d.mask_hi = be_bytes(b + 16, 8); d.mask_lo = be_bytes(b + 24, 8);
			}
			break;
		case XE_BT_RANGE:
			d.low = (int32_t)o->w[0];
			d.high = (int32_t)o->w[1];
			break;
		case XE_BT_MAC:
			d.net_lo = be_bytes(b, 6);
			break;
		default:
			break;
	}
	return d;
}

/* index of an identical term already in the table, or a new one: objects
 * that share a condition (the same list on the same field, the same port
 * range ...) have it evaluated once per record */
/* lookup bit of the (list, family, field) membership test; the probe for
 * (group of the list, family, field) is added on first use */
static int32_t
intern_lookup(struct xe_blob *probes, uint32_t list, int family, uint8_t field, int zero_absent)
{
	struct xe_probe *have = (struct xe_probe *)probes->p;
	struct xe_probe pr;
	uint32_t group = list / XE_LIST_GROUP, l = list % XE_LIST_GROUP;

	for (size_t i = 0; i < probes->n; i++) {
		if ((have[i].group == group) && (have[i].family == family) && (have[i].field == field)) {
			have[i].used |= (uint8_t)(1u << l);
			return (int32_t)(i * XE_LIST_GROUP + l);
		}
	}
	memset(&pr, 0, sizeof(pr));
	pr.group = group;
	pr.family = (uint8_t)family;
	pr.field = field;
	pr.wide4 = (uint8_t)((((family == 4) && (xe_fields[field].size_max == 16)) ? 1 : 0) | (zero_absent ? 2 : 0));
	pr.used = (uint8_t)(1u << l);
	if (!xe_blob_push(probes, &pr, sizeof(pr))) return -1;
	return (int32_t)((probes->n - 1) * XE_LIST_GROUP + l);
}

static int
intern_term(struct intern_ctx *cx, const struct xe_filter *f, size_t ti, uint32_t *id)
{
	struct xe_blob *terms = cx->terms, *opnds = cx->opnds;
	struct xe_term t = f->terms[ti];
	struct xe_blob dl = {0};          /* the term's device operands */
	const struct xe_dopnd *ops;
	const struct xe_term *have;
	const struct xe_dopnd *hop;
	int rc = XE_OK;

	t.pad = 0;
	if ((t.func == XE_FN_GEO) || (t.func == XE_FN_ASN) || (t.func == XE_FN_ASD)) {
		/* geo / AS function: the database becomes an IP list whose leaves are
		 * the nodes that meet the operands (xe_geodb.c); the term becomes
		 * "the field is in that list", with an absent field read as zeros
		 * (the reference reads the field's storage without asking, filter.c:763-783) */
		const uint32_t n = f->terms[ti].nopnd;
		const int fam = (xe_fields[t.fa].size_max == 16) ? 6 : 4;
		int32_t *lo = (int32_t *)calloc(n ? n : 1, sizeof(int32_t)), *hi = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
		const char **strs = (const char **)calloc(n ? n : 1, sizeof(char *));
		struct xe_dopnd d;
		int li = XE_ENOMEM;

		if (lo && hi && strs) {
			for (uint32_t j = 0; j < n; j++) {
				const size_t oi = f->terms[ti].opnd_base + j;
				const int32_t note = f->opnd_note[oi];
				lo[j] = (int32_t)f->opnds[oi].w[0];
				hi[j] = (int32_t)f->opnds[oi].w[1];
				strs[j] = ((t.type == XE_BT_STRING) && (note >= 0) && ((size_t)note < f->nstrs)) ? f->strs + note : NULL;
			}
			/* lists made from the databases start at a group boundary: their
			 * probes carry the "absent reads as zeros" rule for the whole group */
			while ((cx->h->hs.lists.n % XE_LIST_GROUP) && (cx->h->hs.nlists_fn == 0)) {
				char nm[32];
				snprintf(nm, sizeof(nm), "\001pad%zu", cx->h->hs.lists.n);
				if (xe_iplist_load_text(&cx->h->hs.lists, nm, "", 0) != XE_OK) break;
				cx->h->hs.nlists_made++;
			}
			if (!(cx->h->hs.lists.n % XE_LIST_GROUP) || cx->h->hs.nlists_fn) {
				li = xe_geodb_list(&cx->h->hs, t.func, t.k, fam, lo, hi, strs, n);
				if (li >= 0) { cx->h->hs.nlists_fn++; cx->h->hs.nlists_made++; }
			}
		}
		free(lo); free(hi); free(strs);
		if (li < 0) return li;
		memset(&d, 0, sizeof(d));
		d.is_list = 1;
		d.list = (uint32_t)li;
		t.type = (uint8_t)((fam == 4) ? XE_BT_ADDR4 : XE_BT_ADDR6);
		t.dir = XE_DIR_SRC;
		t.fsrc = t.fa;
		t.fdst = 0;
		t.func = XE_FN_NONE;
		t.fa = t.fb = 0;
		t.k = 0;
		t.pad = XE_TERM_ZERO_ABSENT;
		d.low = intern_lookup(cx->lookups, d.list, fam, t.fsrc, 1);
		d.high = -1;
		if (!xe_blob_push(&dl, &d, sizeof(d))) rc = XE_ENOMEM;
	}
	for (uint32_t j = 0; (t.pad != XE_TERM_ZERO_ABSENT) && (j < f->terms[ti].nopnd) && (rc == XE_OK); j++) {
		const size_t oi = f->terms[ti].opnd_base + j;
		const int32_t note = f->opnd_note[oi];
		const char *text = ((t.type == XE_BT_STRING) && (note >= 0) && ((size_t)note < f->nstrs)) ? f->strs + note : NULL;
		struct xe_dopnd d = make_dopnd(&t, &f->opnds[oi]);

		if (t.func == XE_FN_PPSTR) {
			/* conditions on the two port values instead of the text */
			if (text) rc = ppstr_lower(cx->h, text, &dl);
			continue;
		}
		if (t.func == XE_FN_PORTSTR) {
			/* the set of ports whose text equals the operand is laid out behind
			 * the list tables at the end of the commit; here: which text */
			int32_t sid = text ? strpool_intern(cx->strs, text) : -1;
			if (!text) continue;
			if (sid < 0) { rc = XE_ENOMEM; break; }
			d.list = (uint32_t)sid;
		}
		if (d.is_list && (t.func == XE_FN_NONE) && ((t.type == XE_BT_ADDR4) || (t.type == XE_BT_ADDR6))) {
			int fam = (t.type == XE_BT_ADDR4) ? 4 : 6;
			d.low = (t.dir & XE_DIR_SRC) ? intern_lookup(cx->lookups, d.list, fam, t.fsrc, 0) : -1;
			d.high = (t.dir & XE_DIR_DST) ? intern_lookup(cx->lookups, d.list, fam, t.fdst, 0) : -1;
		}
		if (!xe_blob_push(&dl, &d, sizeof(d))) rc = XE_ENOMEM;
	}
	if ((rc == XE_OK) && (t.func == XE_FN_DIV_L)) {
		/* div_l: the quotient goes through pow(k, (int)(log(q) / log(k)))
		 * (filter.h:396-418), a step function of the int quotient.  Its steps
		 * are found with this host's libm, and the operand ranges over the
		 * result become ranges over the quotient itself.  A zero divisor gives
		 * 0 without going through the logarithm (filter.h:401-403): whether 0
		 * is inside the operand ranges is kept in `k`. */
		int32_t lo[80], hi[80], val[80];
		const int np = xe_div_l_pieces(t.k, lo, hi, val, 80);
		struct xe_blob nl = {0};
		const struct xe_dopnd *o = (const struct xe_dopnd *)dl.p;
		int zero_hit = 0;
		for (size_t j = 0; j < dl.n; j++) if ((0 >= o[j].low) && (0 <= o[j].high)) zero_hit = 1;
		for (int i = 0; (i < np) && (rc == XE_OK); i++) {
			int hit = 0;
			for (size_t j = 0; j < dl.n; j++) if ((val[i] >= o[j].low) && (val[i] <= o[j].high)) hit = 1;
			if (hit) {
				struct xe_dopnd d;
				memset(&d, 0, sizeof(d));
				d.low = lo[i];
				d.high = hi[i];
				if (!xe_blob_push(&nl, &d, sizeof(d))) rc = XE_ENOMEM;
			}
		}
		if (np < 0) rc = XE_EUNSUPPORTED;
		free(dl.p);
		dl = nl;
		t.k = zero_hit;
	}
	if (rc != XE_OK) goto out;
	t.nopnd = (uint32_t)dl.n;
	ops = (const struct xe_dopnd *)dl.p;
	have = (const struct xe_term *)terms->p;
	hop = (const struct xe_dopnd *)opnds->p;
	for (size_t k = 0; k < terms->n; k++) {
		const struct xe_term *o = &have[k];
		if ((o->type == t.type) && (o->dir == t.dir) && (o->fsrc == t.fsrc) && (o->fdst == t.fdst)
			&& (o->func == t.func) && (o->fa == t.fa) && (o->fb == t.fb) && (o->k == t.k) && (o->pad == t.pad)
			&& (o->nopnd == t.nopnd)
			&& (!t.nopnd || !memcmp(hop + o->opnd_base, ops, t.nopnd * sizeof(*ops)))) {
			*id = (uint32_t)k;
			goto out;
		}
	}
	t.opnd_base = (uint32_t)opnds->n;
	for (uint32_t j = 0; j < t.nopnd; j++) {
		if (!xe_blob_push(opnds, &ops[j], sizeof(*ops))) { rc = XE_ENOMEM; goto out; }
	}
	*id = (uint32_t)terms->n;
	if (!xe_blob_push(terms, &t, sizeof(t))) rc = XE_ENOMEM;
out:
	free(dl.p);
	return rc;
}

/* append one compiled filter to the flat arrays */
static int
append_prog(struct xe_blob *progs, struct xe_blob *ops, struct intern_ctx *cx, const struct xe_filter *f, int parent,
	int host_only)
{
	struct xe_prog pr;
	int cval = 0;

	memset(&pr, 0, sizeof(pr));
	pr.parent = parent;
	pr.op_base = (uint32_t)ops->n;
	if (host_only) {
		/* decided by the caller's own filter_match(): on the device the object
		 * takes every record its parent took */
		pr.flags = XE_PROG_CONST;
		pr.cval = 1;
		return xe_blob_push(progs, &pr, sizeof(pr)) ? XE_OK : XE_ENOMEM;
	}
	if (xe_filter_constant(f, &cval)) {
		pr.flags = XE_PROG_CONST;
		pr.cval = (uint16_t)cval;
		return xe_blob_push(progs, &pr, sizeof(pr)) ? XE_OK : XE_ENOMEM;
	}
	pr.nops = (uint32_t)f->nops;
	for (size_t i = 0; i < f->nops; i++) {
		uint32_t w = f->ops[i];
		if ((w & 3) == XE_OP_BASIC) {
			uint32_t id;
			int rc = intern_term(cx, f, w >> 2, &id);
			if (rc != XE_OK) return rc;
			w = XE_OP_BASIC | (id << 2);
		}
		if (!xe_blob_push(ops, &w, 4)) return XE_ENOMEM;
	}
	return xe_blob_push(progs, &pr, sizeof(pr)) ? XE_OK : XE_ENOMEM;
}

/* ------------------------------------------------------------------ programs as clause sets */
/* A postfix program as a conjunction of clauses, each an OR of literals
 * (atom, negated?).  Small limits: what does not fit stays a general program. */
#define CNF_MAXC 8
#define CNF_MAXL 8
struct cnf_lit { uint16_t atom; uint8_t neg; };
struct cnf_clause { struct cnf_lit l[CNF_MAXL]; int n; };
struct cnf { struct cnf_clause c[CNF_MAXC]; int n; int ok; };

static int
clause_add(struct cnf_clause *c, struct cnf_lit l, int *tautology)
{
	for (int i = 0; i < c->n; i++) {
		if (c->l[i].atom == l.atom) {
			if (c->l[i].neg != l.neg) *tautology = 1;
			return 1;
		}
	}
	if (c->n >= CNF_MAXL) return 0;
	c->l[c->n++] = l;
	return 1;
}

static void
cnf_and(struct cnf *a, const struct cnf *b)
{
	if (!a->ok || !b->ok || (a->n + b->n > CNF_MAXC)) { a->ok = 0; return; }
	for (int i = 0; i < b->n; i++) a->c[a->n++] = b->c[i];
}

static void
cnf_or(struct cnf *a, const struct cnf *b)
{
	struct cnf r;
	memset(&r, 0, sizeof(r));
	r.ok = a->ok && b->ok;
	for (int i = 0; r.ok && (i < a->n); i++) {
		for (int j = 0; r.ok && (j < b->n); j++) {
			struct cnf_clause c = a->c[i];
			int taut = 0;
			for (int k = 0; k < b->c[j].n; k++) if (!clause_add(&c, b->c[j].l[k], &taut)) r.ok = 0;
			if (taut) continue;
			if (r.n >= CNF_MAXC) { r.ok = 0; break; }
			r.c[r.n++] = c;
		}
	}
	*a = r;
}

/* NOT(c1 & c2 & ...) = OR over clauses of (AND of the negated literals): one
 * clause per choice of one literal from every clause */
static void
cnf_not(struct cnf *a)
{
	struct cnf r;
	int idx[CNF_MAXC] = {0};
	memset(&r, 0, sizeof(r));
	r.ok = a->ok && (a->n > 0) && (a->n <= CNF_MAXL);
	while (r.ok) {
		struct cnf_clause c;
		int taut = 0, k;
		c.n = 0;
		for (k = 0; k < a->n; k++) {
			struct cnf_lit l = a->c[k].l[idx[k]];
			l.neg = !l.neg;
			if (!clause_add(&c, l, &taut)) r.ok = 0;
		}
		if (!taut) {
			if (r.n >= CNF_MAXC) { r.ok = 0; break; }
			r.c[r.n++] = c;
		}
		for (k = 0; k < a->n; k++) {
			if (++idx[k] < a->c[k].n) break;
			idx[k] = 0;
		}
		if (k == a->n) break;
	}
	*a = r;
}

/* the program of object `pg` as clauses over atoms; 0 = not expressible */
static int
prog_to_cnf(const struct xe_prog *pg, const uint32_t *op, const struct xe_fterm *ft, struct cnf *out)
{
	struct cnf *st = (struct cnf *)calloc(64, sizeof(struct cnf));
	int sp = 0, ok = 1;
	if (!st) return 0;
	for (uint32_t k = 0; ok && (k < pg->nops); k++) {
		uint32_t w = op[pg->op_base + k], code = w & 3;
		if (code == XE_OP_BASIC) {
			uint32_t t = w >> 2;
			struct cnf *c;
			if (sp >= 64) { ok = 0; break; }
			c = &st[sp++];
			memset(c, 0, sizeof(*c));
			c->ok = 1;
			c->n = 1;
			if (ft[t].kind & 1) {
				/* all operands are lists: an OR of lookup bits */
				for (int b = 0; b < 64 * XE_LK_WORDS; b++) {
					if (!((ft[t].mask[b >> 6] >> (b & 63)) & 1)) continue;
					if (c->c[0].n >= CNF_MAXL) { c->ok = 0; break; }
					c->c[0].l[c->c[0].n].atom = (uint16_t)b;
					c->c[0].l[c->c[0].n].neg = 0;
					c->c[0].n++;
				}
				if (c->c[0].n == 0) c->ok = 0;   /* no lookup bit: never true; leave to the interpreter */
			} else {
				if (t >= 256) { c->ok = 0; }
				c->c[0].l[0].atom = (uint16_t)(XE_ATOM_TERM0 + t);
				c->c[0].l[0].neg = 0;
				c->c[0].n = 1;
			}
		} else if (code == XE_OP_NOT) {
			if (sp < 1) { ok = 0; break; }
			cnf_not(&st[sp - 1]);
		} else {
			if (sp < 2) { ok = 0; break; }
			if (code == XE_OP_AND) cnf_and(&st[sp - 2], &st[sp - 1]);
			else cnf_or(&st[sp - 2], &st[sp - 1]);
			sp--;
		}
		if (sp && !st[sp - 1].ok) ok = 0;
	}
	if (ok && (sp == 1) && st[0].ok && (st[0].n > 0)) *out = st[0];
	else ok = 0;
	free(st);
	return ok;
}

/* which per-exporter virtual fields some object program reads:
 * bit 0 exporter address / source id, bit 1 the device mark */
static int
reads_device_fields(const struct xe_term *terms, size_t n)
{
	int r = 0;
	for (size_t i = 0; i < n; i++) {
		const struct xe_term *t = &terms[i];
		uint8_t fl[2];
		int k = 0;
		if (t->func != XE_FN_NONE) { fl[k++] = t->fa; fl[k++] = t->fb; }
		else {
			if (t->dir & XE_DIR_SRC) fl[k++] = t->fsrc;
			if (t->dir & XE_DIR_DST) fl[k++] = t->fdst;
		}
		for (int j = 0; j < k; j++) {
			if ((fl[j] == XF_dev_ip) || (fl[j] == XF_dev_id)) r |= 1;
			if (fl[j] == XF_dev_mark) r |= 2;
		}
	}
	return r;
}

/* ------------------------------------------------------------------ fixed-window tables */
extern "C" int
xe_gpu_add_fwm(xe_gpu *h, int mo, const char *const *fields, int nfields, size_t max_keys)
{
	struct fwm_host w, *tmp;
	uint32_t pos = 0, refpos = 0;
	int32_t dlp[3 * 80 * 2];       /* pieces of the table's div_l() fields */
	size_t ndl = 0;

	if (!h || !fields || (nfields < 1) || (mo < 0) || ((size_t)mo >= h->hs.nmos)) return XE_EINVAL;
	if (mo_host_gated(h, mo)) {
		snprintf(h->err, sizeof(h->err), "object %d (or an object above it) is matched by the caller's own "
			"filter_match(): its records are not known on the device", mo);
		return XE_EUNSUPPORTED;
	}
	memset(&w, 0, sizeof(w));
	w.dev.mo = (uint32_t)mo;
	for (int i = 0; i < nfields; i++) {
		struct xe_agfield f;
		char err[400];
		int rc = xe_field_parse(fields[i], &f, err, sizeof(err));
		if (rc != XE_OK) {
			snprintf(h->err, sizeof(h->err), "field '%.100s': %s", fields[i], err);
			return rc;
		}
		if (w.nfields >= (int)(sizeof(w.order) / sizeof(w.order[0]))) {
			snprintf(h->err, sizeof(h->err), "more than %d fields in one table", (int)(sizeof(w.order) / sizeof(w.order[0])));
			return XE_EUNSUPPORTED;
		}
		w.order[w.nfields].aggr = f.aggr;
		w.order[w.nfields].descending = f.descending;
		w.order[w.nfields].size = (uint16_t)(f.aggr ? 8 : f.size);
		w.nfields++;
		if (f.aggr) {
			if (w.dev.nval == XE_FWM_MAX_VALS) {
				snprintf(h->err, sizeof(h->err), "more than %d aggregable fields", XE_FWM_MAX_VALS);
				return XE_EUNSUPPORTED;
			}
			w.dev.vfield[w.dev.nval] = f.field;
			w.dev.vscale[w.dev.nval] = f.scale;
			w.dev.nval++;
		} else {
			/* bytes of the field in the device table */
			uint32_t dsize = f.size;
			if (f.func != XE_FN_NONE) {
				const int two = (f.func != XE_FN_TFSTR) && (f.func != XE_FN_PORTSTR);
				dsize = (f.func == XE_FN_TFSTR) ? 1 : (f.func == XE_FN_PORTSTR) ? 2 : (f.func == XE_FN_PPSTR) ? 4 : 8;
				const uint32_t sa = xe_fields[f.field].size_max, sb = two ? xe_fields[f.fb].size_max : 1;
				if (((sa != 1) && (sa != 2) && (sa != 4) && (sa != 8)) || ((sb != 1) && (sb != 2) && (sb != 4) && (sb != 8))) {
					/* (get_nf_val() reads such a field as 0; none of the port, flag or
					 * counter fields is one) */
					snprintf(h->err, sizeof(h->err), "field '%.100s': function over a field that is not 1, 2, 4 or 8 bytes long", fields[i]);
					return XE_EUNSUPPORTED;
				}
				if (((f.func == XE_FN_PORTSTR) || (f.func == XE_FN_PPSTR)) && !h->port_names) {
					snprintf(h->err, sizeof(h->err), "field '%.100s': portstr() / ppstr() need the port names (xe_gpu_set_port_names)", fields[i]);
					return XE_EUNSUPPORTED;
				}
				if (f.func == XE_FN_DIV_L) {
					int32_t lo[80], hi[80], val[80];
					const int np = xe_div_l_pieces(f.k, lo, hi, val, 80);
					if ((np < 0) || (ndl + 3 * (size_t)np > sizeof(dlp) / sizeof(dlp[0]))) {
						snprintf(h->err, sizeof(h->err), "field '%.100s': div_l() has too many steps", fields[i]);
						return XE_EUNSUPPORTED;
					}
					w.dev.dl_off[w.dev.nkey] = (uint16_t)ndl;
					w.dev.dl_n[w.dev.nkey] = (uint16_t)np;
					for (int q = 0; q < np; q++) { dlp[ndl++] = lo[q]; dlp[ndl++] = hi[q]; dlp[ndl++] = val[q]; }
				}
				if (f.func >= XE_FN_TFSTR) w.texts = 1;
			} else if ((f.size > 16) || (f.size > xe_fields[f.field].col_width)) {
				snprintf(h->err, sizeof(h->err), "field '%.100s': key fields of more than 16 bytes are not aggregated", fields[i]);
				return XE_EUNSUPPORTED;
			}
			if ((w.dev.nkey == XE_FWM_MAX_KEY_FIELDS) || (pos + dsize > 8 * XE_FWM_MAX_KWORDS)) {
				snprintf(h->err, sizeof(h->err), "key of more than %d fields or %d bytes",
					XE_FWM_MAX_KEY_FIELDS, 8 * XE_FWM_MAX_KWORDS);
				return XE_EUNSUPPORTED;
			}
			if (f.field == XF_dev_ip) h->fwm_devip = 1;
			w.dev.kfield[w.dev.nkey] = f.field;
			w.dev.ksize[w.dev.nkey] = (uint8_t)dsize;
			w.dev.kfunc[w.dev.nkey] = w.kfunc[w.dev.nkey] = f.func;
			w.dev.kfield2[w.dev.nkey] = f.fb;
			w.dev.kk[w.dev.nkey] = f.k;
			w.krefsize[w.dev.nkey] = (uint16_t)f.size;
			w.dev.nkey++;
			pos += dsize;
			refpos += f.size;
		}
	}
	w.keybytes = refpos;
	w.devkeybytes = pos;
	w.dev.kwords = (pos + 7) / 8;
	if (!w.dev.kwords) w.dev.kwords = 1;     /* no key field: one row for everything */
	w.max_keys = max_keys ? max_keys : 65536;
	w.slots = 1024;
	while (w.slots < 2 * w.max_keys) w.slots <<= 1;
	if (w.slots > 0x40000000u) return XE_EINVAL;
	w.dev.cap_mask = (uint32_t)(w.slots - 1);

	HIPCHK(h, hipSetDevice(h->device));
	/* a one-word key: state + key; longer ones: seven bytes to a word, no state
	 * (see struct xe_fwm_dev); rounded up to a whole number of 16 bytes */
	w.dev.kw7 = (w.dev.kwords > 1) ? (uint32_t)((pos + 6) / 7) : 0;
	w.dev.swords = (w.dev.kwords > 1) ? ((w.dev.kw7 + 1) & ~1u) : 2;
	tmp = (struct fwm_host *)realloc(h->fwms, (h->nfwm + 1) * sizeof(*tmp));
	if (!tmp) return XE_ENOMEM;
	h->fwms = tmp;
	{
		/* (one slot more than the table has: the row of the one key that the
		 * one-word tables cannot keep among the others, see fwm_global_add()) */
		hipError_t e = hipMalloc((void **)&w.dev.slots, (w.slots + 1) * w.dev.swords * 8);
		if (e == hipSuccess) e = hipMalloc((void **)&w.dev.vals, (w.slots + 1) * (w.dev.nval ? w.dev.nval : 1) * 8);
		if (e == hipSuccess) e = hipMalloc((void **)&w.dev.dropped, 8);
		if ((e == hipSuccess) && ndl) {
			e = hipMalloc((void **)&w.dev.dl, ndl * 4);
			if (e == hipSuccess) e = hipMemcpy((void *)w.dev.dl, dlp, ndl * 4, hipMemcpyHostToDevice);
		}
		if (e == hipSuccess) e = hipMemset(w.dev.slots, 0, (w.slots + 1) * w.dev.swords * 8);
		if (e == hipSuccess) e = hipMemset(w.dev.vals, 0, (w.slots + 1) * (w.dev.nval ? w.dev.nval : 1) * 8);
		if (e == hipSuccess) e = hipMemset(w.dev.dropped, 0, 8);
		if (e != hipSuccess) {
			if (w.dev.slots) (void)hipFree(w.dev.slots);
			if (w.dev.vals) (void)hipFree(w.dev.vals);
			if (w.dev.dropped) (void)hipFree(w.dev.dropped);
			if (w.dev.dl) (void)hipFree((void *)w.dev.dl);
			return hip_fail(h, e, "device memory of a fixed-window table");
		}
	}
	h->fwms[h->nfwm] = w;
	h->committed = 0;
	return (int)h->nfwm++;
}

extern "C" int
xe_gpu_fwm_info(const xe_gpu *h, int fwm, size_t *key_bytes, int *nvals, size_t *slots)
{
	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	if (key_bytes) *key_bytes = h->fwms[fwm].keybytes;
	if (nvals) *nvals = (int)h->fwms[fwm].dev.nval;
	if (slots) *slots = h->fwms[fwm].slots;
	return XE_OK;
}

static long fwm_rows(xe_gpu *h, struct fwm_host *w, uint8_t **keys, uint64_t **vals);

extern "C" long
xe_gpu_fwm_read(xe_gpu *h, int fwm, uint8_t *keys, uint64_t *vals, size_t max_rows, uint64_t *dropped)
{
	struct fwm_host *w;
	uint8_t *rk = NULL;
	uint64_t *rv = NULL;
	long n;

	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	w = &h->fwms[fwm];
	n = fwm_rows(h, w, &rk, &rv);
	if ((n >= 0) && dropped && (hipMemcpy(dropped, w->dev.dropped, 8, hipMemcpyDeviceToHost) != hipSuccess)) {
		n = hip_fail(h, hipGetLastError(), "hipMemcpy(fixed-window table)");
	}
	if (n > 0) {
		const size_t m = ((size_t)n < max_rows) ? (size_t)n : max_rows;
		if (keys) memcpy(keys, rk, m * w->keybytes);
		if (vals) memcpy(vals, rv, m * (size_t)w->dev.nval * 8);
	}
	free(rk);
	free(rv);
	return n;
}

extern "C" int
xe_gpu_fwm_window(const xe_gpu *h, int fwm, int *limit, int *seconds)
{
	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	if (limit) *limit = h->fwms[fwm].limit;
	if (seconds) *seconds = h->fwms[fwm].time;
	return XE_OK;
}

/* rows in the order fwm_sort_and_dump() (monit-objects-fwm.c:496-589) gives
 * them: per row a sort key = the fields in the order of the entry, sums as
 * big-endian u64, key fields as stored, `desc` fields bitwise inverted;
 * rows ascending by the bytes of that key. */
struct fwm_sortrow { const uint8_t *sk; uint32_t row; };

/* a row's key as the reference keeps it (monit_object_key_add_fld(),
 * monit-objects.c:878-921) from the row of the device table: plain fields, div
 * and min are the same bytes; tfstr, portstr and ppstr are zero-padded texts
 * of TCP_FLAGS_STR_MAX_SIZE, TCP_UDP_PORT_STR_MAX_SIZE, TCP_UDP_PP_STR_MAX_SIZE
 * bytes (utils.h:13-15) made of the flags byte / the ports the table keeps */
static void
fwm_ref_key(const xe_gpu *h, const struct fwm_host *w, const uint8_t *dk, uint8_t *out)
{
	for (uint32_t k = 0; k < w->dev.nkey; k++) {
		const uint32_t ds = w->dev.ksize[k], rs = w->krefsize[k];
		char txt[200];
		txt[0] = '\0';
		switch (w->kfunc[k]) {
			case XE_FN_TFSTR:
				tcp_flags_text(dk[0], txt);
				break;
			case XE_FN_PORTSTR:
				port_text(h, (uint32_t)dk[0] | ((uint32_t)dk[1] << 8), txt);
				break;
			case XE_FN_PPSTR:
				port_pair_text(h, (uint32_t)dk[0] | ((uint32_t)dk[1] << 8), (uint32_t)dk[2] | ((uint32_t)dk[3] << 8), txt);
				break;
			default:
				memcpy(out, dk, rs);
				dk += ds;
				out += rs;
				continue;
		}
		memset(out, 0, rs);
		memcpy(out, txt, (strlen(txt) < rs - 1) ? strlen(txt) : rs - 1);
		dk += ds;
		out += rs;
	}
}

/* rows whose texts came out equal are one row in the reference (two pairs of
 * ports can print the same ppstr text): sums added, first row kept */
struct fwm_mrow { const uint8_t *key; uint32_t row; };
static int
fwm_mrow_cmp(const void *a, const void *b, void *len)
{
	const struct fwm_mrow *x = (const struct fwm_mrow *)a, *y = (const struct fwm_mrow *)b;
	const int c = memcmp(x->key, y->key, *(const size_t *)len);
	return c ? c : (x->row > y->row) - (x->row < y->row);
}

static size_t
fwm_merge_rows(const struct fwm_host *w, uint8_t *keys, uint64_t *vals, size_t n)
{
	const size_t kb = w->keybytes, nv = w->dev.nval;
	struct fwm_mrow *idx = (struct fwm_mrow *)malloc(n * sizeof(*idx));
	uint8_t *dead = (uint8_t *)calloc(n, 1);
	size_t kblen = kb, out = 0;
	if (!idx || !dead) { free(idx); free(dead); return n; }
	for (size_t r = 0; r < n; r++) { idx[r].key = keys + r * kb; idx[r].row = (uint32_t)r; }
	qsort_r(idx, n, sizeof(*idx), fwm_mrow_cmp, &kblen);
	for (size_t r = 1, first = 0; r < n; r++) {
		if (memcmp(idx[r].key, idx[first].key, kb)) { first = r; continue; }
		for (size_t j = 0; j < nv; j++) vals[(size_t)idx[first].row * nv + j] += vals[(size_t)idx[r].row * nv + j];
		dead[idx[r].row] = 1;
	}
	for (size_t r = 0; r < n; r++) {
		if (dead[r]) continue;
		if (out != r) {
			memmove(keys + out * kb, keys + r * kb, kb);
			memmove(vals + out * nv, vals + r * nv, nv * 8);
		}
		out++;
	}
	free(idx);
	free(dead);
	return out;
}

/* all rows of a table in freshly allocated arrays (the caller frees them):
 * keys[row][keybytes], vals[row][nval]; returns the number of rows or a
 * negative code */
static long
fwm_rows(xe_gpu *h, struct fwm_host *w, uint8_t **keys, uint64_t **vals)
{
	uint64_t *sl = NULL, *vl = NULL;
	const size_t nv = w->dev.nval ? w->dev.nval : 1;
	size_t n = 0, k = 0;
	long rc = XE_ENOMEM;

	*keys = NULL;
	*vals = NULL;
	HIPCHK(h, hipSetDevice(h->device));
	HIPCHK(h, hipDeviceSynchronize());
	sl = (uint64_t *)malloc((w->slots + 1) * w->dev.swords * 8);
	vl = (uint64_t *)malloc((w->slots + 1) * nv * 8);
	if (!sl || !vl) goto out;
	if ((hipMemcpy(sl, w->dev.slots, (w->slots + 1) * w->dev.swords * 8, hipMemcpyDeviceToHost) != hipSuccess)
		|| (hipMemcpy(vl, w->dev.vals, (w->slots + 1) * nv * 8, hipMemcpyDeviceToHost) != hipSuccess)) {
		rc = hip_fail(h, hipGetLastError(), "hipMemcpy(fixed-window table)");
		goto out;
	}
	for (size_t i = 0; i < w->slots + 1; i++) {
		const uint64_t s0 = sl[i * w->dev.swords];
		if (w->dev.kw7 ? (s0 != 0) : ((uint32_t)s0 == 2)) n++;
	}
	*keys = (uint8_t *)malloc((n ? n : 1) * (size_t)(w->keybytes ? w->keybytes : 1));
	*vals = (uint64_t *)malloc((n ? n : 1) * nv * 8);
	if (!*keys || !*vals) goto out;
	for (size_t i = 0; i < w->slots + 1; i++) {
		uint64_t *S = sl + i * w->dev.swords;
		if (w->dev.kw7) {
			/* seven key bytes in every word, lowest first */
			uint8_t kb[XE_FWM_MAX_KWORDS * 8 + 8];
			if (S[0] == 0) continue;
			for (uint32_t j = 0; j < w->dev.kw7; j++) memcpy(kb + 7 * j, &S[j], 7);
			fwm_ref_key(h, w, kb, *keys + k * w->keybytes);
		} else {
			if ((uint32_t)S[0] != 2) continue;
			S[1] ^= XE_FWM_KEY_XOR;
			fwm_ref_key(h, w, (const uint8_t *)(S + 1), *keys + k * w->keybytes);
		}
		memcpy(*vals + k * w->dev.nval, vl + i * nv, (size_t)w->dev.nval * 8);
		k++;
	}
	if (w->texts && (n > 1)) n = fwm_merge_rows(w, *keys, *vals, n);
	rc = (long)n;
out:
	free(sl);
	free(vl);
	return rc;
}
static int
fwm_sort_cmp(const void *a, const void *b, void *len)
{
	return memcmp(((const struct fwm_sortrow *)a)->sk, ((const struct fwm_sortrow *)b)->sk, *(const size_t *)len);
}

extern "C" long
xe_gpu_fwm_export(xe_gpu *h, int fwm, int limit, uint8_t *keys, uint64_t *vals, size_t max_rows,
	uint64_t *others, int *hit_limit)
{
	struct fwm_host *w;
	uint8_t *rk = NULL, *sk = NULL;
	uint64_t *rv = NULL;
	struct fwm_sortrow *idx = NULL;
	size_t sklen = 0, nout;
	long n, rc = XE_ENOMEM;

	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	w = &h->fwms[fwm];
	if (limit < 0) limit = w->limit;
	if (hit_limit) *hit_limit = 0;
	for (uint32_t j = 0; others && (j < w->dev.nval); j++) others[j] = 0;
	/* the table travels once: the rows are taken out of one copy of it */
	n = fwm_rows(h, w, &rk, &rv);
	if (n <= 0) { free(rk); free(rv); return n; }
	for (int i = 0; i < w->nfields; i++) sklen += w->order[i].size;
	sk = (uint8_t *)malloc((size_t)n * (sklen ? sklen : 1));
	idx = (struct fwm_sortrow *)malloc((size_t)n * sizeof(*idx));
	if (!sk || !idx) goto out;
	for (long r = 0; r < n; r++) {
		uint8_t *d = sk + (size_t)r * sklen;
		const uint8_t *kp = rk + (size_t)r * w->keybytes;
		const uint64_t *vp = rv + (size_t)r * w->dev.nval;
		for (int i = 0; i < w->nfields; i++) {
			if (w->order[i].aggr) {
				uint64_t v = __builtin_bswap64(*vp++);
				if (w->order[i].descending) v = ~v;
				memcpy(d, &v, 8);
				d += 8;
			} else {
				for (uint32_t j = 0; j < w->order[i].size; j++) {
					*d++ = w->order[i].descending ? (uint8_t)~*kp : *kp;
					kp++;
				}
			}
		}
		idx[r].sk = sk + (size_t)r * sklen;
		idx[r].row = (uint32_t)r;
	}
	qsort_r(idx, (size_t)n, sizeof(*idx), fwm_sort_cmp, &sklen);

	/* the first `limit` rows; the sums of all later rows go to `others`
	 * (fwm_dump(), monit-objects-fwm.c:405-449: the flag is raised when the
	 * limit is reached, even with nothing behind it) */
	nout = (size_t)n;
	if (limit && ((size_t)limit <= nout)) {
		nout = (size_t)limit;
		if (hit_limit) *hit_limit = 1;
	}
	for (size_t r = 0; r < (size_t)n; r++) {
		const uint32_t s = idx[r].row;
		if (r < nout) {
			if (r < max_rows) {
				if (keys) memcpy(keys + r * w->keybytes, rk + (size_t)s * w->keybytes, w->keybytes);
				if (vals) memcpy(vals + r * w->dev.nval, rv + (size_t)s * w->dev.nval, (size_t)w->dev.nval * 8);
			}
		} else if (others) {
			for (uint32_t j = 0; j < w->dev.nval; j++) others[j] += rv[(size_t)s * w->dev.nval + j];
		}
	}
	rc = (long)nout;
out:
	free(rk); free(rv); free(sk); free(idx);
	return rc;
}

extern "C" int
xe_gpu_fwm_clear(xe_gpu *h, int fwm)
{
	struct fwm_host *w;
	if (!h || (fwm < 0) || ((size_t)fwm >= h->nfwm)) return XE_EINVAL;
	w = &h->fwms[fwm];
	HIPCHK(h, hipSetDevice(h->device));
	HIPCHK(h, hipDeviceSynchronize());
	HIPCHK(h, hipMemset(w->dev.slots, 0, (w->slots + 1) * w->dev.swords * 8));
	HIPCHK(h, hipMemset(w->dev.vals, 0, (w->slots + 1) * (w->dev.nval ? w->dev.nval : 1) * 8));
	HIPCHK(h, hipMemset(w->dev.dropped, 0, 8));
	return XE_OK;
}

extern "C" size_t
xe_gpu_batch_unsupported_sets(const xe_batch *b)
{
	return b ? b->plan.unsupported_sets : 0;
}

extern "C" size_t
xe_gpu_batch_oversized_datagrams(const xe_batch *b)
{
	return b ? b->plan.oversized_datagrams : 0;
}

extern "C" void *
xe_gpu_host_alloc(xe_gpu *h, size_t bytes)
{
	void *p = NULL;
	if (!h || !bytes) return NULL;
	if (hipSetDevice(h->device) != hipSuccess) return NULL;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		snprintf(h->err, sizeof(h->err), "hipHostMalloc(%zu) failed", bytes);
		return NULL;
	}
	return p;
}

extern "C" void
xe_gpu_host_free(xe_gpu *h, void *p)
{
	(void)h;
	if (p) (void)hipHostFree(p);
}

extern "C" int
xe_gpu_set_prepass_threads(xe_gpu *h, int nthreads)
{
	if (!h || (nthreads < 0) || (nthreads > 64)) return XE_EINVAL;
	h->hs.prepass_threads = nthreads;
	return XE_OK;
}

extern "C" int
xe_gpu_commit(xe_gpu *h)
{
	struct xe_blob progs = {0}, ops = {0}, terms = {0}, opnds = {0}, trie = {0}, rules = {0}, lookups = {0};
	struct strpool strs = { {0}, {0} };
	struct intern_ctx cx = { h, &terms, &opnds, &lookups, &strs };

	/* the lists an earlier commit made from the geo / AS databases go: this one makes its own */
	drop_made_lists(h);
	int rc = XE_OK;

	if (!h) return XE_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	free_config_dev(h);

	for (size_t m = 0; (m < h->hs.nmos) && (rc == XE_OK); m++) {
		rc = append_prog(&progs, &ops, &cx, &h->hs.mos[m].filter, h->hs.mos[m].parent, h->hs.mos[m].host_only);
	}
	/* the mark filters of devices.conf follow the objects' programs; their terms share the term table */
	for (size_t d = 0; (d < h->hs.ndevs) && (rc == XE_OK); d++) {
		struct xe_devrule r;
		memset(&r, 0, sizeof(r));
		r.first_prog = (uint32_t)progs.n;
		r.nprogs = (uint32_t)h->hs.devs[d].nmarks;
		r.skip_unmarked = (uint32_t)h->hs.devs[d].skip_unmarked;
		for (size_t k = 0; (k < h->hs.devs[d].nmarks) && (rc == XE_OK); k++) {
			rc = append_prog(&progs, &ops, &cx, &h->hs.devs[d].marks[k], -1, 0);
		}
		if ((rc == XE_OK) && !xe_blob_push(&rules, &r, sizeof(r))) rc = XE_ENOMEM;
	}
	if (rc == XE_OK) {
		/* the fast kernel evaluates the terms of the object programs and of the
		 * mark filters alike */
		h->nterms = (uint32_t)terms.n;
		h->fast_ok = 1;
		{
			int rd = reads_device_fields((const struct xe_term *)terms.p, terms.n);
			h->want_dev = rd & 1;
			h->reads_mark = (rd & 2) != 0;
		}
		{
			/* Terms whose operands are all IP lists are answered from the lookup
			 * bits and need no term bit of their own unless a listed program
			 * reads them: they are numbered after the others, so that only
			 * the others count against the 256 term bits of a record. */
			struct xe_term *tt = (struct xe_term *)terms.p;
			const struct xe_dopnd *oo = (const struct xe_dopnd *)opnds.p;
			uint32_t *op = (uint32_t *)ops.p;
			size_t n = terms.n, k = 0;
			uint32_t *newid = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
			struct xe_term *copy = (struct xe_term *)malloc((n ? n : 1) * sizeof(*copy));
			if (!newid || !copy) {
				rc = XE_ENOMEM;
			} else {
				for (int pass = 0; pass < 2; pass++) {
					for (size_t t = 0; t < n; t++) {
						int all_lists = (tt[t].func == XE_FN_NONE) && (tt[t].nopnd > 0)
							&& ((tt[t].type == XE_BT_ADDR4) || (tt[t].type == XE_BT_ADDR6));
						for (uint32_t j = 0; all_lists && (j < tt[t].nopnd); j++) {
							if (!oo[tt[t].opnd_base + j].is_list) all_lists = 0;
						}
						if (all_lists == pass) newid[t] = (uint32_t)k++;
					}
					if (pass == 0) h->nplain_terms = (uint32_t)k;
				}
				memcpy(copy, tt, n * sizeof(*copy));
				for (size_t t = 0; t < n; t++) tt[newid[t]] = copy[t];
				for (size_t i = 0; i < ops.n; i++) {
					if ((op[i] & 3) == XE_OP_BASIC) op[i] = XE_OP_BASIC | (newid[op[i] >> 2] << 2);
				}
			}
			free(newid);
			free(copy);
		}
		if (h->nplain_terms > XE_MAX_FAST_TERMS) h->fast_ok = 0;
		h->nlookups = (uint32_t)lookups.n;
		if (h->nlookups > XE_MAX_PROBES) h->fast_ok = 0;
		h->lw = (h->nlookups <= 8) ? 1u : (uint32_t)XE_LK_WORDS;
	}
	if (rc == XE_OK) {
		h->h_lists = (struct xe_list_dev *)calloc(h->hs.lists.n ? h->hs.lists.n : 1, sizeof(struct xe_list_dev));
		if (!h->h_lists) rc = XE_ENOMEM;
		else rc = xe_iplists_flatten(&h->hs.lists, &trie, h->h_lists, h->list_root_bits);
	}
	h->port_named_off = 0;
	if ((rc == XE_OK) && h->port_names) {
		/* behind the list tables: which ports have a service name (65536 bits),
		 * then per portstr() text the set of ports whose text equals it
		 * (port_to_str + strcasecmp, filter.c:861-880) */
		const uint32_t zero = 0;
		int uses_ports = 0;
		const struct xe_term *tt = (const struct xe_term *)terms.p;
		for (size_t t = 0; t < terms.n; t++) if ((tt[t].func == XE_FN_PORTSTR) || (tt[t].func == XE_FN_PPSTR)) uses_ports = 1;
		if (uses_ports) {
			uint32_t *set = (uint32_t *)calloc(2048, 4), *setoff = (uint32_t *)calloc(strs.offs.n + 1, 4);
			if (!set || !setoff) rc = XE_ENOMEM;
			if ((rc == XE_OK) && (trie.n == 0) && !xe_blob_push(&trie, &zero, 4)) rc = XE_ENOMEM;
			while ((rc == XE_OK) && (trie.n % 4)) if (!xe_blob_push(&trie, &zero, 4)) rc = XE_ENOMEM;
			if (rc == XE_OK) {
				for (uint32_t p = 0; p < 65536; p++) if (port_name(h, p)[0]) set[p >> 5] |= 1u << (p & 31);
				h->port_named_off = (uint32_t)trie.n;
				for (int k = 0; (k < 2048) && (rc == XE_OK); k++) if (!xe_blob_push(&trie, &set[k], 4)) rc = XE_ENOMEM;
			}
			for (size_t i = 0; (i < strs.offs.n) && (rc == XE_OK); i++) {
				const char *text = (const char *)strs.chars.p + ((const uint32_t *)strs.offs.p)[i];
				memset(set, 0, 2048 * 4);
				for (uint32_t p = 0; p < 65536; p++) {
					char buf[96];
					port_text(h, p, buf);
					if (!strcasecmp(text, buf)) set[p >> 5] |= 1u << (p & 31);
				}
				setoff[i] = (uint32_t)trie.n;
				for (int k = 0; (k < 2048) && (rc == XE_OK); k++) if (!xe_blob_push(&trie, &set[k], 4)) rc = XE_ENOMEM;
			}
			if (rc == XE_OK) {
				struct xe_dopnd *oo = (struct xe_dopnd *)opnds.p;
				for (size_t t = 0; t < terms.n; t++) {
					if (tt[t].func != XE_FN_PORTSTR) continue;
					for (uint32_t j = 0; j < tt[t].nopnd; j++) oo[tt[t].opnd_base + j].low = (int32_t)setoff[oo[tt[t].opnd_base + j].list];
				}
			}
			free(set);
			free(setoff);
		}
	}
	if ((rc == XE_OK) && (trie.n > 0x7FFFFFF0u) && h->port_named_off) rc = XE_EUNSUPPORTED;
	if ((rc == XE_OK) && (trie.n > 0xFFFFFFF0u)) rc = XE_EUNSUPPORTED;
	if (rc == XE_OK) {
		/* a probe walks the merged tables of its group of lists */
		struct xe_probe *pp = (struct xe_probe *)lookups.p;
		for (size_t i = 0; i < lookups.n; i++) {
			const struct xe_list_dev *ld = &h->h_lists[(size_t)pp[i].group * XE_LIST_GROUP];
			pp[i].root_off = (pp[i].family == 4) ? ld->root4 : ld->root6;
			pp[i].nodes_off = (pp[i].family == 4) ? ld->nodes4 : ld->nodes6;
			pp[i].chains_off = (pp[i].family == 4) ? ld->chains4 : ld->chains6;
			pp[i].root_bits = (pp[i].family == 4) ? (ld->rbits & 0xFF) : ((ld->rbits >> 8) & 0xFF);
		}
	}
	if (rc == XE_OK) {
		/* fast-kernel view of the object programs and their terms */
		const struct xe_term *tt = (const struct xe_term *)terms.p;
		const struct xe_dopnd *oo = (const struct xe_dopnd *)opnds.p;
		const struct xe_prog *pg = (const struct xe_prog *)progs.p;
		const uint32_t *op = (const uint32_t *)ops.p;
		struct xe_fterm *ft = (struct xe_fterm *)calloc(h->nterms ? h->nterms : 1, sizeof(*ft));
		struct xe_fprog *fp = (struct xe_fprog *)calloc(h->hs.nmos ? h->hs.nmos : 1, sizeof(*fp));
		if (!ft || !fp) rc = XE_ENOMEM;
		for (uint32_t t = 0; (rc == XE_OK) && (t < h->nterms); t++) {
			int all_lists = (tt[t].func == XE_FN_NONE) && (tt[t].nopnd > 0)
				&& ((tt[t].type == XE_BT_ADDR4) || (tt[t].type == XE_BT_ADDR6));
			uint64_t m[XE_LK_WORDS];
			memset(m, 0, sizeof(m));
			for (uint32_t j = 0; all_lists && (j < tt[t].nopnd); j++) {
				const struct xe_dopnd *o = &oo[tt[t].opnd_base + j];
				if (!o->is_list) { all_lists = 0; break; }
				if ((o->low >= 0) && (o->low < 64 * XE_LK_WORDS)) m[o->low >> 6] |= 1ull << (o->low & 63);
				if ((o->high >= 0) && (o->high < 64 * XE_LK_WORDS)) m[o->high >> 6] |= 1ull << (o->high & 63);
			}
			ft[t].kind = all_lists ? 1 : 0;
			for (int k = 0; k < XE_LK_WORDS; k++) ft[t].mask[k] = all_lists ? m[k] : 0;
		}
		{
			/* objects resolved through the atom table, the rest listed */
			size_t words = h->hs.nmos ? (h->hs.nmos + 63) / 64 : 1, nb = 0;
			/* + per word: which lookup atoms / term atoms have a row that is not empty */
			const size_t mrows = XE_MOTAB_ROWS(h->lw);
			uint64_t *mt = (uint64_t *)calloc(words * (mrows + h->lw + 1), sizeof(uint64_t));
			uint64_t *cand = (uint64_t *)calloc(words, sizeof(uint64_t));
			struct xe_cjderived *der = (struct xe_cjderived *)calloc(XE_MAX_DERIVED, sizeof(struct xe_cjderived));
			struct xe_cjrow *rows = NULL;
			size_t nrows = 0, rows_cap = 0, nder = 0;
			struct cj_obj { uint16_t atom[CNF_MAXC]; uint8_t neg[CNF_MAXC]; int n; };
			if (!mt || !cand || !der) rc = XE_ENOMEM;
			for (size_t m = 0; (rc == XE_OK) && (m < h->hs.nmos); m++) {
				struct xe_fprog q;
				memset(&q, 0, sizeof(q));
				q.op_base = pg[m].op_base;
				q.nops = pg[m].nops;
				q.parent = pg[m].parent;
				q.cval = pg[m].cval;
				q.mo = (uint32_t)m;
				if (pg[m].flags & XE_PROG_CONST) {
					q.kind = XE_FPROG_CONST;
				} else if ((pg[m].nops == 1) && ((op[pg[m].op_base] & 3) == XE_OP_BASIC)) {
					uint32_t t = op[pg[m].op_base] >> 2;
					if ((ft[t].kind & 1) && (h->lw == 1)) {
						q.kind = XE_FPROG_MASK;
						q.mask = ft[t].mask[0];
					} else if ((ft[t].kind & 1) && (pg[m].parent < 0)) {
						/* lookup bits in several words: straight to the atom table */
						for (int bit = 0; bit < 64 * XE_LK_WORDS; bit++) {
							if ((ft[t].mask[bit >> 6] >> (bit & 63)) & 1) mt[(m / 64) * mrows + bit] |= 1ull << (m % 64);
						}
						continue;
					} else {
						q.kind = XE_FPROG_TERM;
						q.term = t;
					}
				} else {
					q.kind = XE_FPROG_GENERAL;
				}
				if ((q.parent < 0) && (q.kind == XE_FPROG_MASK)) {
					for (int bit = 0; bit < 64; bit++) {
						if ((q.mask >> bit) & 1) mt[(m / 64) * mrows + bit] |= 1ull << (m % 64);
					}
					continue;
				}
				if ((q.parent < 0) && (q.kind == XE_FPROG_TERM) && (q.term < 64)) {
					mt[(m / 64) * mrows + 64 * h->lw + q.term] |= 1ull << (m % 64);
					ft[q.term].kind |= 2;
					continue;
				}
				/* a conjunction of clauses: matched through the need / forbid rows */
				if (q.kind != XE_FPROG_CONST) {
					struct cnf f;
					struct cj_obj co;
					int fits = prog_to_cnf(&pg[m], op, ft, &f);
					memset(&co, 0, sizeof(co));
					for (int ci = 0; fits && (ci < f.n); ci++) {
						const struct cnf_clause *c = &f.c[ci];
						uint32_t atom;
						int neg = 0;
						if (c->n == 1) {
							atom = c->l[0].atom;
							neg = c->l[0].neg;
						} else {
							struct xe_cjderived d;
							size_t j;
							memset(&d, 0, sizeof(d));
							for (int li = 0; li < c->n; li++) {
								uint32_t a = c->l[li].atom;
								if (a < XE_ATOM_TERM0) {
									if (c->l[li].neg) d.neg_lk[a >> 6] |= 1ull << (a & 63);
									else d.pos_lk[a >> 6] |= 1ull << (a & 63);
								} else if (a < XE_ATOM_TERM0 + 64) {
									if (c->l[li].neg) d.neg_tb |= 1ull << (a - XE_ATOM_TERM0);
									else d.pos_tb |= 1ull << (a - XE_ATOM_TERM0);
								} else {
									fits = 0;
								}
							}
							if (!fits) break;
							for (j = 0; j < nder; j++) if (!memcmp(&der[j], &d, sizeof(d))) break;
							if (j == nder) {
								if (nder >= XE_MAX_DERIVED) { fits = 0; break; }
								der[nder++] = d;
							}
							atom = XE_ATOM_DERIVED0 + (uint32_t)j;
						}
						co.atom[co.n] = (uint16_t)atom;
						co.neg[co.n] = (uint8_t)neg;
						co.n++;
					}
					if (fits) {
						for (int ci = 0; ci < co.n; ci++) {
							/* row of (atom, word): find or add */
							size_t j;
							uint32_t a = co.atom[ci];
							for (j = 0; j < nrows; j++) if ((rows[j].atom == a) && (rows[j].word == m / 64)) break;
							if (j == nrows) {
								if (nrows == rows_cap) {
									struct xe_cjrow *nr = (struct xe_cjrow *)realloc(rows, (rows_cap ? rows_cap * 2 : 64) * sizeof(*rows));
									if (!nr) { rc = XE_ENOMEM; break; }
									rows = nr;
									rows_cap = rows_cap ? rows_cap * 2 : 64;
								}
								memset(&rows[nrows], 0, sizeof(rows[0]));
								rows[nrows].atom = a;
								rows[nrows].word = (uint32_t)(m / 64);
								nrows++;
							}
							if (co.neg[ci]) rows[j].forbid |= 1ull << (m % 64);
							else rows[j].need |= 1ull << (m % 64);
							/* terms the rows read must be evaluated */
							if ((a >= XE_ATOM_TERM0) && (a < XE_ATOM_DERIVED0)) ft[a - XE_ATOM_TERM0].kind |= 2;
							if (a >= XE_ATOM_DERIVED0) {
								const struct xe_cjderived *d = &der[a - XE_ATOM_DERIVED0];
								uint64_t tbm = d->pos_tb | d->neg_tb;
								for (int b = 0; b < 64; b++) if ((tbm >> b) & 1) ft[b].kind |= 2;
							}
						}
						if (rc != XE_OK) break;
						cand[m / 64] |= 1ull << (m % 64);
						if (q.parent >= 0) {
							q.kind = XE_FPROG_GATE;
							fp[nb++] = q;
						}
						continue;
					}
				}
				/* evaluated one by one: mark the terms it needs */
				if (q.kind == XE_FPROG_TERM) ft[q.term].kind |= 2;
				if (q.kind == XE_FPROG_GENERAL) {
					for (uint32_t k = 0; k < pg[m].nops; k++) {
						uint32_t w = op[pg[m].op_base + k];
						if ((w & 3) == XE_OP_BASIC) ft[w >> 2].kind |= 2;
					}
				}
				fp[nb++] = q;
			}
			h->nfprogs = (uint32_t)nb;
			/* the mark filters of devices.conf (device_get_mark(), devices.c:276-281) in
			 * the form the fast kernel interprets; entry i belongs to program nmos + i */
			h->nmarkprogs = 0;
			if ((rc == XE_OK) && (progs.n > h->hs.nmos)) {
				const size_t nm = progs.n - h->hs.nmos;
				struct xe_fprog *mf = (struct xe_fprog *)calloc(nm, sizeof(*mf));
				if (!mf) rc = XE_ENOMEM;
				for (size_t i = 0; (rc == XE_OK) && (i < nm); i++) {
					const struct xe_prog *g = &pg[h->hs.nmos + i];
					struct xe_fprog q;
					memset(&q, 0, sizeof(q));
					q.op_base = g->op_base;
					q.nops = g->nops;
					q.parent = -1;
					q.cval = g->cval;
					if (g->flags & XE_PROG_CONST) {
						q.kind = XE_FPROG_CONST;
					} else if ((g->nops == 1) && ((op[g->op_base] & 3) == XE_OP_BASIC)) {
						const uint32_t t = op[g->op_base] >> 2;
						if ((ft[t].kind & 1) && (h->lw == 1)) {
							q.kind = XE_FPROG_MASK;
							q.mask = ft[t].mask[0];
						} else {
							q.kind = XE_FPROG_TERM;
							q.term = t;
							ft[t].kind |= 2;
						}
					} else {
						q.kind = XE_FPROG_GENERAL;
						for (uint32_t k = 0; k < g->nops; k++) {
							const uint32_t w = op[g->op_base + k];
							if ((w & 3) == XE_OP_BASIC) ft[w >> 2].kind |= 2;
						}
					}
					mf[i] = q;
				}
				if (rc == XE_OK) rc = upload(h, &h->d_mark_fprogs, mf, nm * sizeof(*mf));
				if (rc == XE_OK) h->nmarkprogs = (uint32_t)nm;
				free(mf);
			}
			/* rows in mask-word order (stable: a simple insertion sort, few rows) */
			for (size_t i = 1; i < nrows; i++) {
				struct xe_cjrow x = rows[i];
				size_t j = i;
				while (j && (rows[j - 1].word > x.word)) { rows[j] = rows[j - 1]; j--; }
				rows[j] = x;
			}
			h->ncjderived = (uint32_t)nder;
			{
				/* fold the rows into tables indexed by 8 atoms at once */
				uint8_t gof[XE_CJ_MAX_GROUPS];
				uint32_t gb[XE_CJ_MAX_GROUPS + 4], ng = 0;
				uint64_t *tab = NULL;
				memset(gof, 0xFF, sizeof(gof));
				for (size_t j = 0; j < nrows; j++) {
					uint32_t a = rows[j].atom;
					uint32_t k = (a < XE_ATOM_TERM0) ? (a >> 3) : (a < XE_ATOM_DERIVED0) ? 32 + ((a - XE_ATOM_TERM0) >> 3)
						: 64 + ((a - XE_ATOM_DERIVED0) >> 3);
					gof[k] = 0;
				}
				for (uint32_t k = 0; k < XE_CJ_MAX_GROUPS; k++) if (gof[k] == 0) { gof[k] = (uint8_t)ng; gb[ng++] = k; }
				if (ng) {
					tab = (uint64_t *)calloc((size_t)ng * 256 * words, sizeof(uint64_t));
					if (!tab) rc = XE_ENOMEM;
				}
				for (size_t j = 0; (rc == XE_OK) && (j < nrows); j++) {
					uint32_t a = rows[j].atom;
					uint32_t k = (a < XE_ATOM_TERM0) ? (a >> 3) : (a < XE_ATOM_DERIVED0) ? 32 + ((a - XE_ATOM_TERM0) >> 3)
						: 64 + ((a - XE_ATOM_DERIVED0) >> 3);
					uint64_t *t0 = tab + (size_t)gof[k] * 256 * words + rows[j].word;
					for (uint32_t v = 0; v < 256; v++) {
						t0[(size_t)v * words] |= ((v >> (a & 7)) & 1) ? rows[j].forbid : rows[j].need;
					}
				}
				/* the kernel takes 4 groups per step: repeating a group is harmless (OR) */
				h->ncjgroups = ng;
				while (ng & 3) { gb[ng] = gb[0]; ng++; }
				if ((rc == XE_OK) && h->ncjgroups) {
					/* a repeated group must also find its table: copies share group 0's
					 * index, so the list holds (byte, table index) pairs */
					uint32_t pairs[2 * (XE_CJ_MAX_GROUPS + 4)];
					for (uint32_t g = 0; g < ng; g++) {
						pairs[2 * g] = gb[g];
						pairs[2 * g + 1] = (g < h->ncjgroups) ? g : 0;
					}
					rc = upload(h, &h->d_cjgb, pairs, (size_t)ng * 8);
					if (rc == XE_OK) rc = upload(h, &h->d_cjtab, tab, (size_t)h->ncjgroups * 256 * words * sizeof(uint64_t));
				}
				free(tab);
			}
			if (h->dbg) {
				fprintf(stderr, "xe_gpu_commit: %zu objects, %u terms, %u lookups, %zu listed programs, "
					"%zu need/forbid rows in %u byte tables, %zu derived atoms\n", h->hs.nmos, h->nterms, h->nlookups, nb, nrows, h->ncjgroups, nder);
			}
			for (size_t w = 0; mt && (w < words); w++) {
				for (size_t a = 0; a < mrows; a++) {
					if (mt[w * mrows + a]) mt[words * mrows + (h->lw + 1) * w + (a >> 6)] |= 1ull << (a & 63);
				}
			}
			h->motab_words = words * (mrows + h->lw + 1);
			h->rg_lds = 0;
			if (rc == XE_OK) rc = upload(h, &h->d_motab, mt, h->motab_words * sizeof(uint64_t));
			if (rc == XE_OK) rc = upload(h, &h->d_fprogs, fp, nb * sizeof(*fp));
			if (rc == XE_OK) rc = upload(h, &h->d_cjderived, der, nder * sizeof(*der));
			if (rc == XE_OK) rc = upload(h, &h->d_cjcand, cand, words * sizeof(uint64_t));
			free(h->h_motab);
			h->h_motab = mt;      /* (the range groups' tables may join it below) */
			free(cand);
			free(der);
			free(rows);
		}
		h->nrgroups = 0;
		if (rc == XE_OK) {
			/* needed range terms, grouped by the field they read */
			uint8_t chosen[XE_NFIELDS];
			struct xe_rgroup rg[XE_MAX_RGROUPS + 4];
			struct xe_blob rgd = {0};
			uint32_t nrg = 0, nchosen = 0;
			memset(chosen, 0, sizeof(chosen));
#define XE_RTERM(T) ((ft[T].kind & 2) && (tt[T].type == XE_BT_RANGE) && (tt[T].func == XE_FN_NONE) && (tt[T].dir != XE_DIR_NONE))
			for (uint32_t t = 0; t < h->nterms; t++) {
				if (!XE_RTERM(t)) continue;
				for (int side = 0; side < 2; side++) {
					uint32_t f = side ? tt[t].fdst : tt[t].fsrc;
					if (!(tt[t].dir & (side ? XE_DIR_DST : XE_DIR_SRC)) || chosen[f]) continue;
					if (nchosen < XE_MAX_RGROUPS) { chosen[f] = 1; nchosen++; }
				}
			}
			for (uint32_t f = 0; (rc == XE_OK) && (f < XE_NFIELDS); f++) {
				int64_t *bp;
				size_t nb = 0, cap = 0;
				if (!chosen[f]) continue;
				/* a term is taken out of the one-by-one list only when all the
				 * fields it reads have a group */
				for (uint32_t t = 0; t < h->nterms; t++) {
					if (!XE_RTERM(t)) continue;
					if ((tt[t].dir & XE_DIR_SRC) && !chosen[tt[t].fsrc]) continue;
					if ((tt[t].dir & XE_DIR_DST) && !chosen[tt[t].fdst]) continue;
					if (!(((tt[t].dir & XE_DIR_SRC) && (tt[t].fsrc == f)) || ((tt[t].dir & XE_DIR_DST) && (tt[t].fdst == f)))) continue;
					cap += 2 * (size_t)tt[t].nopnd;
				}
				bp = (int64_t *)malloc((cap ? cap : 1) * sizeof(int64_t));
				if (!bp) { rc = XE_ENOMEM; break; }
				for (uint32_t t = 0; t < h->nterms; t++) {
					if (!XE_RTERM(t)) continue;
					if ((tt[t].dir & XE_DIR_SRC) && !chosen[tt[t].fsrc]) continue;
					if ((tt[t].dir & XE_DIR_DST) && !chosen[tt[t].fdst]) continue;
					if (!(((tt[t].dir & XE_DIR_SRC) && (tt[t].fsrc == f)) || ((tt[t].dir & XE_DIR_DST) && (tt[t].fdst == f)))) continue;
					for (uint32_t j = 0; j < tt[t].nopnd; j++) {
						const struct xe_dopnd *o = &oo[tt[t].opnd_base + j];
						if (o->low > o->high) continue;
						bp[nb++] = o->low;
						if (o->high < INT32_MAX) bp[nb++] = (int64_t)o->high + 1;
					}
				}
				if (nb) {
					/* sort, unique */
					for (size_t i = 1; i < nb; i++) {
						int64_t x = bp[i];
						size_t j = i;
						while (j && (bp[j - 1] > x)) { bp[j] = bp[j - 1]; j--; }
						bp[j] = x;
					}
					size_t u = 1;
					for (size_t i = 1; i < nb; i++) if (bp[i] != bp[u - 1]) bp[u++] = bp[i];
					nb = u;
					struct xe_rgroup g;
					g.field = f;
					g.n = (uint32_t)nb;
					g.bp_off = (uint32_t)rgd.n;
					for (size_t i = 0; i < nb; i++) {
						int32_t v = (int32_t)bp[i];
						if (!xe_blob_push(&rgd, &v, 4)) rc = XE_ENOMEM;
					}
					while ((rc == XE_OK) && (rgd.n & 3)) { uint32_t z = 0; if (!xe_blob_push(&rgd, &z, 4)) rc = XE_ENOMEM; }
					g.tab_off = (uint32_t)rgd.n;
					for (size_t i = 0; (rc == XE_OK) && (i <= nb); i++) {
						/* a value inside interval i: below the first breakpoint, or the
						 * breakpoint that opens it */
						int64_t rep = i ? bp[i - 1] : bp[0] - 1;
						uint64_t e[4] = { 0, 0, 0, 0 };
						for (uint32_t t = 0; t < h->nterms; t++) {
							if (!XE_RTERM(t)) continue;
							if ((tt[t].dir & XE_DIR_SRC) && !chosen[tt[t].fsrc]) continue;
							if ((tt[t].dir & XE_DIR_DST) && !chosen[tt[t].fdst]) continue;
							if (!(((tt[t].dir & XE_DIR_SRC) && (tt[t].fsrc == f)) || ((tt[t].dir & XE_DIR_DST) && (tt[t].fdst == f)))) continue;
							for (uint32_t j = 0; j < tt[t].nopnd; j++) {
								const struct xe_dopnd *o = &oo[tt[t].opnd_base + j];
								if ((rep >= o->low) && (rep <= o->high)) { e[(t >> 6) & 3] |= 1ull << (t & 63); break; }
							}
						}
						for (int k = 0; (k < 4) && (rc == XE_OK); k++) {
							uint32_t lo32 = (uint32_t)e[k], hi32 = (uint32_t)(e[k] >> 32);
							if (!xe_blob_push(&rgd, &lo32, 4) || !xe_blob_push(&rgd, &hi32, 4)) rc = XE_ENOMEM;
						}
					}
					if (rc == XE_OK) rg[nrg++] = g;
				}
				free(bp);
			}
			/* grouped terms leave the one-by-one list (also those whose ranges are all empty) */
			for (uint32_t t = 0; (rc == XE_OK) && (t < h->nterms); t++) {
				if (!XE_RTERM(t)) continue;
				if ((tt[t].dir & XE_DIR_SRC) && !chosen[tt[t].fsrc]) continue;
				if ((tt[t].dir & XE_DIR_DST) && !chosen[tt[t].fdst]) continue;
				ft[t].kind = (ft[t].kind & ~2u) | 4u;
			}
#undef XE_RTERM
			h->nrgroups = nrg;
			while (nrg & 3) { rg[nrg] = rg[0]; nrg++; }   /* taken 4 at a time; a repeat is harmless (OR) */
			if ((rc == XE_OK) && h->nrgroups) {
				rc = upload(h, &h->d_rgroups, rg, (size_t)nrg * sizeof(rg[0]));
				if (rc == XE_OK) rc = upload(h, &h->d_rgdata, rgd.p, rgd.n * 4);
				/* small tables also go behind the atom table, which the kernels copy
				 * to LDS while all of it stays below the size tested everywhere
				 * (motab_words): an entry then comes from LDS instead of a gather */
				const size_t rgw = (rgd.n + 1) / 2;
				if ((rc == XE_OK) && h->h_motab && h->motab_words && !h->rg_lds
					&& ((h->motab_words + rgw) * 8 <= 8192 + 64)) {
					uint64_t *both = (uint64_t *)calloc(h->motab_words + rgw, 8);
					if (!both) {
						rc = XE_ENOMEM;
					} else {
						memcpy(both, h->h_motab, h->motab_words * 8);
						memcpy(both + h->motab_words, rgd.p, rgd.n * 4);
						if (h->d_motab) (void)hipFree(h->d_motab);
						rc = upload(h, &h->d_motab, both, (h->motab_words + rgw) * 8);
						if (rc == XE_OK) {
							h->rg_lds = (uint32_t)h->motab_words;
							h->motab_words += rgw;
						}
						free(both);
					}
				}
			}
			if (h->dbg) fprintf(stderr, "xe_gpu_commit: %u range groups, %zu table bytes\n", h->nrgroups, rgd.n * 4);
			free(rgd.p);
		}
		if (rc == XE_OK) rc = upload(h, &h->d_fterms, ft, (size_t)h->nterms * sizeof(*ft));
		if (rc == XE_OK) {
			uint32_t *need = (uint32_t *)calloc(h->nterms ? h->nterms : 1, sizeof(uint32_t));
			uint32_t n = 0;
			if (!need) rc = XE_ENOMEM;
			for (uint32_t t = 0; (rc == XE_OK) && (t < h->nterms); t++) {
				if (!(ft[t].kind & 2)) continue;
				need[n++] = t;
				/* a list-only term beyond the term bits that a listed program reads */
				if (t >= XE_MAX_FAST_TERMS) h->fast_ok = 0;
			}
			h->ntneed = n;
			if (h->dbg) fprintf(stderr, "xe_gpu_commit: %u terms evaluated per record\n", n);
			if (rc == XE_OK) rc = upload(h, &h->d_tneed, need, (size_t)n * sizeof(uint32_t));
			/* terms that read the device mark: the fast kernel evaluates them again
			 * once the mark filters have run (the filters themselves see no mark) */
			n = 0;
			for (uint32_t t = 0; (rc == XE_OK) && (t < h->nterms); t++) {
				if (!(ft[t].kind & 6)) continue;
				if (reads_device_fields(&tt[t], 1) & 2) need[n++] = t;
			}
			h->ntmark = n;
			if ((rc == XE_OK) && n) rc = upload(h, &h->d_tmark, need, (size_t)n * sizeof(uint32_t));
			free(need);
		}
		free(ft);
		free(fp);
	}
	if ((rc == XE_OK) && h->nfwm) {
		struct xe_fwm_dev *dv = (struct xe_fwm_dev *)malloc(h->nfwm * sizeof(*dv));
		if (!dv) rc = XE_ENOMEM;
		for (size_t i = 0; dv && (i < h->nfwm); i++) dv[i] = h->fwms[i].dev;
		if (rc == XE_OK) rc = upload(h, &h->d_fwmdesc, dv, h->nfwm * sizeof(*dv));
		free(dv);
	}
	if (rc == XE_OK) rc = upload(h, &h->d_progs, progs.p, progs.n * sizeof(struct xe_prog));
	if (rc == XE_OK) rc = upload(h, &h->d_ops, ops.p, ops.n * 4);
	if (rc == XE_OK) rc = upload(h, &h->d_terms, terms.p, terms.n * sizeof(struct xe_term));
	if (rc == XE_OK) rc = upload(h, &h->d_opnds, opnds.p, opnds.n * sizeof(struct xe_dopnd));
	if (rc == XE_OK) rc = upload(h, &h->d_trie, trie.p, trie.n * 4);
	if (rc == XE_OK) rc = upload(h, &h->d_lists, h->h_lists, h->hs.lists.n * sizeof(struct xe_list_dev));
	if (rc == XE_OK) rc = upload(h, &h->d_devrules, rules.p, rules.n * sizeof(struct xe_devrule));
	if (rc == XE_OK) {
		/* entries whose filters compile to the same programs are one rule set to the
		 * fast kernel, which makes one pass per distinct set among a tile's lanes */
		const struct xe_devrule *rl = (const struct xe_devrule *)rules.p;
		const struct xe_prog *pg = (const struct xe_prog *)progs.p;
		const uint32_t *op = (const uint32_t *)ops.p;
		h->dev_canon = (uint32_t *)calloc(h->hs.ndevs ? h->hs.ndevs : 1, sizeof(uint32_t));
		if (!h->dev_canon) rc = XE_ENOMEM;
		for (size_t d = 0; (rc == XE_OK) && (d < h->hs.ndevs); d++) {
			h->dev_canon[d] = (uint32_t)d;
			for (size_t e = 0; e < d; e++) {
				int same = (h->dev_canon[e] == e) && (rl[e].nprogs == rl[d].nprogs) && (rl[e].skip_unmarked == rl[d].skip_unmarked);
				for (uint32_t k = 0; same && (k < rl[d].nprogs); k++) {
					const struct xe_prog *a = &pg[rl[e].first_prog + k], *b2 = &pg[rl[d].first_prog + k];
					same = (a->nops == b2->nops) && (a->flags == b2->flags) && (a->cval == b2->cval)
						&& !memcmp(op + a->op_base, op + b2->op_base, (size_t)a->nops * 4);
				}
				if (same) { h->dev_canon[d] = (uint32_t)e; break; }
			}
		}
	}
	if (rc == XE_OK) rc = upload(h, &h->d_lookups, lookups.p, lookups.n * sizeof(struct xe_probe));
	if (rc == XE_OK) {
		h->h_probes = (struct xe_probe *)calloc(lookups.n ? lookups.n : 1, sizeof(struct xe_probe));
		if (!h->h_probes) rc = XE_ENOMEM;
		else if (lookups.n) memcpy(h->h_probes, lookups.p, lookups.n * sizeof(struct xe_probe));
	}
	if (rc == XE_OK) {
		/* devices.conf as the header walk on the device looks exporters up */
		struct xe_devkey *dk = (struct xe_devkey *)calloc(h->hs.ndevs ? h->hs.ndevs : 1, sizeof(*dk));
		if (!dk) rc = XE_ENOMEM;
		for (size_t i = 0; dk && (i < h->hs.ndevs); i++) {
			const struct xe_device *d = &h->hs.devs[i];
			dk[i].use_ip = (uint32_t)d->use_ip;
			dk[i].use_id = (uint32_t)d->use_id;
			dk[i].ip = d->ip;
			dk[i].id = d->id;
			dk[i].rate = (uint32_t)d->sampling_rate;
			/* (a run names its entry in 15 bits) */
			dk[i].plain = (i < 0x7FFEu) ? 1u : 0u;
			dk[i].rules = (d->nmarks || d->skip_unmarked) ? h->dev_canon[i] + 1u : 0u;
		}
		if (h->d_devkeys) (void)hipFree(h->d_devkeys);
		h->d_devkeys = NULL;
		if (rc == XE_OK) rc = upload(h, &h->d_devkeys, dk, h->hs.ndevs * sizeof(*dk));
		if (h->d_devhash) (void)hipFree(h->d_devhash);
		h->d_devhash = NULL;
		h->devhash_mask = 0;
		if ((rc == XE_OK) && (h->hs.ndevs >= XE_DEVHASH_MIN)) {
			/* device_find() (devices.c:205-232) as three look-ups: per kind of test
			 * and key the FIRST entry that passes it */
			size_t cap = 16;
			struct xe_devhent *ht;
			while (cap < 4 * 2 * h->hs.ndevs) cap <<= 1;
			ht = (struct xe_devhent *)calloc(cap, sizeof(*ht));
			if (!ht) rc = XE_ENOMEM;
			for (size_t i = 0; ht && (i < h->hs.ndevs); i++) {
				for (uint32_t kind = 1; kind <= 3; kind++) {
					uint32_t kip, kid;
					size_t j;
					if ((kind == 1) && !(dk[i].use_ip && !dk[i].use_id)) continue;
					if ((kind == 2) && !(dk[i].use_ip && dk[i].use_id)) continue;
					if ((kind == 3) && !dk[i].use_id) continue;
					kip = (kind == 3) ? 0u : dk[i].ip;
					kid = (kind == 1) ? 0u : dk[i].id;
					for (j = xe_devh_hash(kind, kip, kid) & (cap - 1); ht[j].kind; j = (j + 1) & (cap - 1)) {
						if ((ht[j].kind == kind) && (ht[j].ip == kip) && (ht[j].id == kid)) break;   /* an earlier entry answers */
					}
					if (!ht[j].kind) { ht[j].kind = kind; ht[j].ip = kip; ht[j].id = kid; ht[j].dev = (uint32_t)i; }
				}
			}
			if (rc == XE_OK) rc = upload(h, &h->d_devhash, ht, cap * sizeof(*ht));
			if (rc == XE_OK) h->devhash_mask = (uint32_t)(cap - 1);
			free(ht);
		}
		free(dk);
		h->walk_epoch = 0;   /* which templates the fast kernel takes depends on the configuration */
	}
	if (rc == XE_OK) {
		h->nmo = (uint32_t)h->hs.nmos;
		h->mask_words = h->nmo ? (h->nmo + 63) / 64 : 1;
		h->nprogs_total = (uint32_t)progs.n;
		h->committed = 1;
	} else {
		free_config_dev(h);
	}
	free(progs.p); free(ops.p); free(terms.p); free(opnds.p); free(trie.p); free(rules.p); free(lookups.p);
	free(strs.chars.p); free(strs.offs.p);
	return rc;
}

/* ------------------------------------------------------------------ batches */
static int
sync_templates(xe_gpu *h)
{
	size_t nt = h->hs.dev_tmpls.n, ne = h->hs.dev_tents.n;
	int rc;

	if (h->d_tmpls && h->d_tents && (nt == h->up_tmpls) && (ne == h->up_tents)) return XE_OK;
	if (h->d_tmpls) (void)hipFree(h->d_tmpls);
	if (h->d_tents) (void)hipFree(h->d_tents);
	h->d_tmpls = h->d_tents = NULL;
	rc = upload(h, &h->d_tmpls, h->hs.dev_tmpls.p, nt * sizeof(struct xe_tmpl));
	if (rc == XE_OK) rc = upload(h, &h->d_tents, h->hs.dev_tents.p, ne * sizeof(struct xe_tent));
	if (rc != XE_OK) {
		/* nothing half-uploaded stays behind: the next batch starts over */
		if (h->d_tmpls) (void)hipFree(h->d_tmpls);
		if (h->d_tents) (void)hipFree(h->d_tents);
		h->d_tmpls = h->d_tents = NULL;
		h->up_tmpls = h->up_tents = 0;
		return rc;
	}
	h->up_tmpls = nt;
	h->up_tents = ne;
	return XE_OK;
}

/* From this many records on a batch takes the per-lane kernel: the fast kernel
 * addresses a column element with a 32-bit byte offset (16-byte columns: below
 * 2^28 records).  (XE_FAST_MAX_RECORDS in the environment lets a test reach
 * that path with a small batch.) */
static uint64_t
fast_record_limit(const xe_gpu *h)
{
	const char *e = getenv("XE_FAST_MAX_RECORDS");   /* once per prepared batch, never per launch */
	const uint64_t v = e ? strtoull(e, NULL, 0) : 0;
	(void)h;
	return (v && (v < (1ull << 28))) ? v : (1ull << 28);
}

/* page-locked scratch of the current prepare call */
static void
pin_begin(xe_gpu *h, size_t want)
{
	h->pin_used = 0;
	if (want <= h->pin_cap) return;
	pinned_free(h->pin);
	h->pin_cap = 0;
	want = want * 5 / 4 + 4096;
	h->pin = (uint8_t *)pinned_alloc(want);
	if (h->pin) h->pin_cap = want;
}

/* NULL when the scratch is used up (the caller then takes ordinary memory) */
static void *
pin_get(xe_gpu *h, size_t bytes)
{
	const size_t at = ALIGN_UP(h->pin_used, 256);
	if (!h->pin || (at + bytes > h->pin_cap)) return NULL;
	h->pin_used = at + bytes;
	return h->pin + at;
}

/* ------------------------------------------------------------------ header walk on the device */
static int tmpl_fast(const xe_gpu *h, const struct xe_tmpl *t, const struct xe_tent *te);

/* the template store and the device templates in the form xe_walk.hip reads,
 * uploaded again whenever the store has changed */
static int
walk_tables_sync(xe_gpu *h)
{
	const struct xe_host *hs = &h->hs;
	const struct xe_tmpl *tm = (const struct xe_tmpl *)hs->dev_tmpls.p;
	const struct xe_tent *te = (const struct xe_tent *)hs->dev_tents.p;
	size_t nt = hs->dev_tmpls.n, cap = 16, live = 0;
	struct xe_twalk *tw;
	struct xe_thent *th;
	uint8_t *rawb;
	size_t rawn = 0, rawat = 0;
	int rc;

	if (h->walk_epoch == hs->tmpl_epoch + 1) return XE_OK;
	for (size_t i = 0; i < hs->ntslots; i++) if (hs->tslots[i].raw && (hs->tslots[i].dev_slot != XE_TMPL_UNSUPPORTED)) live++;
	while (cap < 2 * live + 2) cap <<= 1;
	for (size_t i = 0; i < hs->ntslots; i++) if (hs->tslots[i].raw) rawn += hs->tslots[i].raw_len;
	tw = (struct xe_twalk *)calloc(nt ? nt : 1, sizeof(*tw));
	th = (struct xe_thent *)calloc(cap, sizeof(*th));
	rawb = (uint8_t *)malloc(rawn + 16);
	if (!tw || !th || !rawb || (rawn > 0xFFFF0000ull)) { free(tw); free(th); free(rawb); return XE_ENOMEM; }
	for (size_t t = 0; t < nt; t++) {
		tw[t].stride = tm[t].stride;
		tw[t].nfields = tm[t].nfields;
		tw[t].off_last = tm[t].nfields ? te[tm[t].ent_base + tm[t].nfields - 1].off : 0;
		tw[t].fast = (uint8_t)tmpl_fast(h, &tm[t], te);
		tw[t].mode = tm[t].mode;
	}
	for (size_t i = 0; i < hs->ntslots; i++) {
		const struct xe_tslot *ts = &hs->tslots[i];
		struct xe_thent e;
		size_t j;
		if (!ts->raw || (ts->dev_slot == XE_TMPL_UNSUPPORTED) || (ts->raw_len < 4)) continue;
		memset(&e, 0, sizeof(e));
		e.ver_tid = ((uint32_t)ts->key.ver << 16) | ts->key.tid;
		e.src_ip = ts->key.src_ip;
		e.source_id = ts->key.source_id;
		e.slot = ts->dev_slot;
		e.fc = ((uint32_t)ts->raw[2] << 8) | ts->raw[3];
		e.raw_off = (uint32_t)rawat;
		e.raw_len = (uint32_t)ts->raw_len;
		memcpy(rawb + rawat, ts->raw, ts->raw_len);
		rawat += ts->raw_len;
		for (j = xe_thent_hash(e.ver_tid, e.src_ip, e.source_id) & (cap - 1); th[j].ver_tid; j = (j + 1) & (cap - 1)) {}
		th[j] = e;
	}
	if (h->d_thash) (void)hipFree(h->d_thash);
	if (h->d_twalk) (void)hipFree(h->d_twalk);
	if (h->d_traw) (void)hipFree(h->d_traw);
	h->d_thash = h->d_twalk = h->d_traw = NULL;
	rc = upload(h, &h->d_thash, th, cap * sizeof(*th));
	if (rc == XE_OK) rc = upload(h, &h->d_twalk, tw, nt * sizeof(*tw));
	if (rc == XE_OK) rc = upload(h, &h->d_traw, rawb, rawat + 16);
	free(tw);
	free(th);
	free(rawb);
	if (rc != XE_OK) {
		if (h->d_thash) (void)hipFree(h->d_thash);
		if (h->d_twalk) (void)hipFree(h->d_twalk);
		if (h->d_traw) (void)hipFree(h->d_traw);
		h->d_thash = h->d_twalk = h->d_traw = NULL;
		h->walk_epoch = 0;
		return rc;
	}
	h->thash_mask = (uint32_t)(cap - 1);
	h->twalk_n = nt;
	h->walk_epoch = hs->tmpl_epoch + 1;
	return XE_OK;
}

static void
walk_free(xe_gpu *h, struct walk_dev *w)
{
	dev_put(h, w->arena);
	dev_put(h, w->tiles);
	dev_put(h, w->d_rec2dg);
	for (int i = 0; i < w->nwev; i++) (void)hipEventDestroy(w->wev[i]);
	free(w->kinds2);
	free(w->h_walked);
	free(w->h_src);
	free(w->h_recbase);
	free(w->h_sfbase);
	free(w->h_dgruns);
	memset(w, 0, sizeof(*w));
}

static void
walk_input(const xe_gpu *h, const xe_batch *b, struct xe_walk_in *in)
{
	const struct walk_dev *w = &b->w;
	memset(in, 0, sizeof(*in));
	in->pkt = (const uint8_t *)b->d_pkt;
	in->offs = w->d_offs;
	in->src = w->d_src;
	in->kinds = w->d_kinds;
	in->n = (uint32_t)w->n;
	in->ndevs = (uint32_t)h->hs.ndevs;
	in->devs = (const struct xe_devkey *)h->d_devkeys;
	in->devhash = (const struct xe_devhent *)h->d_devhash;
	in->devhash_mask = h->devhash_mask;
	in->thash = (const struct xe_thent *)h->d_thash;
	in->thash_mask = h->thash_mask;
	in->v5_slot_plus1 = h->hs.v5_slot_plus1;
	in->twalk = (const struct xe_twalk *)h->d_twalk;
	in->traw = (const uint8_t *)h->d_traw;
	in->ntmpl = (uint32_t)h->twalk_n;
}

/* First half, before the host walk: the per-datagram arrays go to the device,
 * the sFlow datagrams are counted, the Netflow datagrams sorted into those the
 * device lists itself and those the host must walk (w->kinds2 tells the host
 * walk which to pass by).  Everything is enqueued on the copy stream behind
 * the datagram bytes. */
static int
walk_begin(xe_gpu *h, xe_batch *b, const uint8_t *data, const uint64_t *offsets, const uint32_t *src, const uint8_t *kinds,
	size_t n, int mode)
{
	struct walk_dev *w = &b->w;
	hipStream_t st = h->copy_stream;
	struct xe_walk_in in;
	size_t off = 0;
	uint8_t *base;
	int rc;

	w->mode = mode;
	w->n = n;
	w->hdr_src = (const uint8_t *)b->d_pkt;
	if ((mode == 2) && !b->sent_ahead) {
		/* Netflow only, from page-locked memory the GPU can read in place: the
		 * walk needs 24 bytes of every datagram, which it fetches over the link
		 * itself while the copy of the whole buffer is under way */
		hipPointerAttribute_t at;
		memset(&at, 0, sizeof(at));
		if ((hipPointerGetAttributes(&at, data) == hipSuccess) && (at.type == hipMemoryTypeHost) && at.devicePointer) {
			w->hdr_src = (const uint8_t *)at.devicePointer;
			st = h->walk_stream;
		} else {
			(void)hipGetLastError();
		}
	}
	w->st = st;
	if (b->ev_bulk && ((mode & 1) || (w->hdr_src == (const uint8_t *)b->d_pkt))) {
		/* the walk reads the device copy of the datagrams */
		HIPCHK(h, hipStreamWaitEvent(st, b->ev_bulk, 0));
	}
	if (mode & 2) {
		rc = walk_tables_sync(h);
		if (rc != XE_OK) return rc;
		if (h->twalk_n >= (1u << 16)) return XE_EUNSUPPORTED;   /* (prepare keeps such a store off this path) */
	}
	const size_t nr = (mode & 2) ? n * XE_WALK_SETS : n;   /* runs: up to XE_WALK_SETS per datagram */
	const size_t nt = (mode & 2) ? h->twalk_n : 0;         /* device templates as the walk tables hold them */
	w->temp_bytes = xe_walk_temp_bytes(n + 1);
#define CARVE(BYTES) (off = ALIGN_UP(off, 256), off += (BYTES), off - (BYTES))
	const size_t o_offs = CARVE(8 * (n + 1)), o_src = CARVE(4 * n), o_kinds = CARVE(n), o_sfcnt = CARVE(4 * (n + 2)),
		o_sfbase = CARVE(4 * (n + 2)), o_total = CARVE(XE_WZERO_BYTES(nt)), o_temp = CARVE((mode & 1) ? w->temp_bytes : 0),
		o_sfdg = CARVE((mode & 1) ? sizeof(struct xe_sfdg) * (n + 1) : 0),
		o_sfag = CARVE((mode & 1) ? sizeof(struct xe_sfagent) * (n + 1) : 0);
	size_t o_dgruns = 0, o_walked = 0, o_hostfirst = 0, o_recbase = 0;
	if (mode & 2) {
		o_dgruns = CARVE(sizeof(struct xe_dgrun) * nr); o_walked = CARVE(n); o_hostfirst = CARVE(4 * (n + 2));
		o_recbase = CARVE(4 * (n + 2));
	}
#undef CARVE
	rc = dev_get(h, &w->arena, off + 256);
	if (rc != XE_OK) return rc;
	base = (uint8_t *)w->arena;
	w->d_offs = (uint64_t *)(base + o_offs);
	w->d_src = (uint32_t *)(base + o_src);
	w->d_kinds = kinds ? base + o_kinds : NULL;
	w->d_sfcnt = (uint32_t *)(base + o_sfcnt);
	w->d_sfbase = (uint32_t *)(base + o_sfbase);
	w->d_sfdg = (struct xe_sfdg *)(base + o_sfdg);
	w->d_sfag = (struct xe_sfagent *)(base + o_sfag);
	w->d_total = (unsigned long long *)(base + o_total);
	w->d_temp = base + o_temp;
	if (mode & 2) {
		w->d_dgruns = (struct xe_dgrun *)(base + o_dgruns);
		w->d_walked = base + o_walked;
		w->d_hostfirst = (uint32_t *)(base + o_hostfirst);
		w->d_recbase = (uint32_t *)(base + o_recbase);
	}
	/* (cleared ahead of the copies below, which it then runs beside) */
	HIPCHK(h, hipMemsetAsync(w->d_total, 0, (mode & 2) ? XE_WZERO_BYTES(nt) : 32, st));
	{
		/* the per-datagram arrays go through page-locked scratch (a copy out of
		 * ordinary memory blocks the calling thread inside the runtime, and with it
		 * the other handles' threads) into the arena: the walk kernels, and later
		 * the decode kernels, read them from device memory */
		uint64_t *po = (uint64_t *)pin_get(h, 8 * (n + 1));
		uint32_t *ps = (uint32_t *)pin_get(h, 4 * n + 4);
		uint8_t *pk = kinds ? (uint8_t *)pin_get(h, n + 1) : NULL;
		const int pinned = po && ps && (pk || !kinds);
		if (pinned) {
			memcpy(po, offsets, 8 * (n + 1));
			memcpy(ps, src, 4 * n);
			if (kinds) memcpy(pk, kinds, n);
		}
		HIPCHK(h, hipMemcpyAsync(w->d_offs, pinned ? po : offsets, 8 * (n + 1), hipMemcpyHostToDevice, st));
		HIPCHK(h, hipMemcpyAsync(w->d_src, pinned ? ps : src, 4 * n, hipMemcpyHostToDevice, st));
		if (kinds) HIPCHK(h, hipMemcpyAsync(w->d_kinds, pinned ? pk : kinds, n, hipMemcpyHostToDevice, st));
	}
	walk_mark(h, w, st);
	walk_input(h, b, &in);
	w->kinds2 = (uint8_t *)malloc(n ? n : 1);
	w->h_src = (uint32_t *)malloc((n ? n : 1) * 4);
	if (!w->kinds2 || !w->h_src) return XE_ENOMEM;
	memcpy(w->h_src, src, 4 * n);
	if (mode & 2) {
		in.pkt = w->hdr_src;
		rc = xe_walk_nf_shape(&in, w->d_dgruns, w->d_walked, w->d_total, st);
		if (rc != XE_OK) return hip_fail(h, hipGetLastError(), "header walk (Netflow)");
		w->h_walked = (uint8_t *)malloc(n ? n : 1);
		if (!w->h_walked) return XE_ENOMEM;
		HIPCHK(h, hipMemcpyAsync((uint8_t *)h->wstage + 32, w->d_total + 4, XE_WZERO_BYTES(nt) - 32, hipMemcpyDeviceToHost, st));
	}
	if (mode & 1) {
		in.pkt = (const uint8_t *)b->d_pkt;
		rc = xe_walk_sf_count(&in, w->d_sfcnt, w->d_sfdg, w->d_sfag, w->d_total, st);
		if (rc == XE_OK) rc = xe_walk_scan(w->d_temp, w->temp_bytes, w->d_sfcnt, w->d_sfbase, n, st);
		if (rc != XE_OK) return hip_fail(h, hipGetLastError(), "header walk (sFlow)");
		HIPCHK(h, hipMemcpyAsync(h->wstage, w->d_total, 8, hipMemcpyDeviceToHost, st));
	}
	walk_mark(h, w, st);
	/* the host walk needs to know which Netflow datagrams are left to it; with only sFlow
	 * on the device it starts at once and the device counts beside it */
	if (mode & 2) {
		HIPCHK(h, hipStreamSynchronize(st));
		{
			/* most runs of a datagram, datagrams left to the host: over the copies */
			uint32_t kmax = 0, left = 0;
			for (size_t c = 0; c < XE_WCOPIES; c++) {
				const volatile uint32_t *q = (const volatile uint32_t *)((uint8_t *)h->wstage + 32 + c * XE_WCOPY_BYTES(nt) + 8);
				if (q[0] > kmax) kmax = q[0];
				left += q[1];
			}
			w->nsets = kmax;
			w->sums_valid = left == 0;
		}
		if (w->nsets < 1) w->nsets = 1;
		if (w->nsets > XE_WALK_SETS) w->nsets = XE_WALK_SETS;
		if (w->sums_valid) {
			/* every Netflow datagram is of the usual shape */
			for (size_t i = 0; i < n; i++) w->h_walked[i] = !(kinds && (kinds[i] == XE_KIND_SFLOW));
		} else {
			uint8_t *pin_walked = (uint8_t *)pin_get(h, n + 1);
			HIPCHK(h, hipMemcpyAsync(pin_walked ? pin_walked : w->h_walked, w->d_walked, n, hipMemcpyDeviceToHost, st));
			HIPCHK(h, hipStreamSynchronize(st));
			if (pin_walked) memcpy(w->h_walked, pin_walked, n);
		}
	}
	for (size_t i = 0; i < n; i++) {
		uint8_t k = kinds ? kinds[i] : (uint8_t)XE_KIND_NETFLOW;
		if (k == XE_KIND_SFLOW) { if (mode & 1) k |= XE_KIND_ON_DEVICE; }
		else if ((mode & 2) && w->h_walked[i]) k |= XE_KIND_ON_DEVICE;
		w->kinds2[i] = k;
	}
	return XE_OK;
}

/* The host walk over what the device left it, with the template changes of
 * the batch logged: Netflow datagrams the device listed were decided with the
 * templates known when the batch began, so an exporter whose templates change
 * inside the batch is walked by the host from its first datagram on (the store
 * is put back to where it was and the walk repeated). */
static int
u32_cmp(const void *a, const void *b)
{
	const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
	return (x > y) - (x < y);
}

static int
walk_host_part(xe_gpu *h, xe_batch *b, const uint8_t *data, const uint64_t *offsets, const uint32_t *src, size_t n)
{
	struct walk_dev *w = &b->w;
	int rc;

	if (!(w->mode & 2)) return xe_prepass(&h->hs, data, offsets, src, w->kinds2, n, &b->plan);
	rc = xe_prepass(&h->hs, data, offsets, src, w->kinds2, n, &b->plan);
	if ((rc == XE_OK) && xe_host_undo_count(&h->hs)) {
		const size_t nu = xe_host_undo_count(&h->hs);
		uint32_t *ips = (uint32_t *)malloc(nu * 4);
		size_t nips = 0, changed = 0;
		if (!ips) { xe_plan_free(&b->plan); return XE_ENOMEM; }
		/* the exporters whose templates changed, sorted: every datagram is then
		 * looked up in a few steps however many exporters changed at once */
		for (size_t i = 0; i < nu; i++) ips[i] = xe_host_undo_src_ip(&h->hs, i);
		qsort(ips, nu, 4, u32_cmp);
		for (size_t i = 0; i < nu; i++) if (!i || (ips[i] != ips[i - 1])) ips[nips++] = ips[i];
		for (size_t i = 0; i < n; i++) {
			if (!w->h_walked[i]) continue;
			if (bsearch(&src[i], ips, nips, 4, u32_cmp)) {
				w->h_walked[i] = 0;
				w->kinds2[i] &= (uint8_t)~XE_KIND_ON_DEVICE;
				changed++;
			}
		}
		free(ips);
		if (changed) {
			hipError_t e;
			xe_host_undo_rollback(&h->hs);
			xe_plan_free(&b->plan);
			rc = xe_prepass(&h->hs, data, offsets, src, w->kinds2, n, &b->plan);
			e = hipMemcpyAsync(w->d_walked, w->h_walked, n, hipMemcpyHostToDevice, w->st);
			if ((rc == XE_OK) && (e != hipSuccess)) { xe_plan_free(&b->plan); rc = hip_fail(h, e, "hipMemcpyAsync(walked)"); }
		}
	}
	return rc;
}

#define XE_WALK_RETRY_ON_HOST 1   /* walk_finish(): the batch needs the host walk for everything */

/* Second half, after the host walk: record indices in arrival order over both
 * walks, the sFlow element descriptors, the run and tile tables of the
 * Netflow datagrams the device listed. */
static int
walk_finish(xe_gpu *h, xe_batch *b)
{
	struct walk_dev *w = &b->w;
	struct xe_plan *plan = &b->plan;
	hipStream_t st = w->st;
	const size_t n = w->n, nt = h->twalk_n;
	struct xe_walk_in in;
	int rc;

	const int dbg = h->dbg;
	double tk[6] = {0, 0, 0, 0, 0, 0};
#define WTICK(I) do { if (dbg) { struct timespec ts_; clock_gettime(CLOCK_MONOTONIC, &ts_); tk[I] = ts_.tv_sec * 1e3 + ts_.tv_nsec * 1e-6; } } while (0)

	walk_input(h, b, &in);
	WTICK(0);
	if (w->mode & 1) {
		unsigned long long nsf;
		HIPCHK(h, hipStreamSynchronize(st));
		nsf = *(volatile unsigned long long *)h->wstage;
		if (nsf + plan->nrec > 0xFFFFFFF0ull) return XE_EUNSUPPORTED;
		rc = dev_get(h, &w->d_rec2dg, ((size_t)nsf + 1) * sizeof(struct xe_sfrd));
		if (rc != XE_OK) return rc;
		walk_mark(h, w, st);
		rc = xe_walk_sf_place((uint32_t)n, w->d_sfbase, w->d_sfdg, (struct xe_sfrd *)w->d_rec2dg, st);
		walk_mark(h, w, st);
		if (rc != XE_OK) return hip_fail(h, hipGetLastError(), "header walk (sFlow records)");
		plan->nsf = (size_t)nsf;
	}
	if (w->mode & 2) {
		/* the words the device added up, as they lie in its arena (XE_WZERO_BYTES):
		 * cnt[0, nt + 2) records, cnt[nt + 2, ...) runs per template */
		uint8_t *back = (uint8_t *)h->wstage;
		uint32_t *cnt = (uint32_t *)(back + 32 + 16), *cntruns = cnt + (nt + 2);
		struct xe_wgroup *gr;
		uint8_t *tmap;
		const struct xe_tmpl *tm = (const struct xe_tmpl *)h->hs.dev_tmpls.p;
		uint64_t total, dev_total = 0, work = 0, nruns = 0, ntiles = 0;
		size_t nactive = 0, npass, off = 0;
		const uint32_t *d_hostfirst = plan->nrec ? w->d_hostfirst : NULL;
		uint8_t *fbase;

		if (XE_WZERO_BYTES(nt) > XE_WSTAGE_BYTES) return XE_WALK_RETRY_ON_HOST;
		if (plan->nrec) HIPCHK(h, hipMemcpyAsync(w->d_hostfirst, plan->dg_first_nf, 4 * (n + 1), hipMemcpyHostToDevice, st));
		if (!w->sums_valid || xe_host_undo_count(&h->hs)) {
			/* the host walked some datagrams (and may have taken exporters back from
			 * the device): the sums once more, over what is still listed here */
			walk_mark(h, w, st);
			HIPCHK(h, hipMemsetAsync((uint8_t *)w->d_total + 32, 0, XE_WZERO_BYTES(nt) - 32, st));
			rc = xe_walk_nf_totals(&in, w->d_dgruns, w->nsets, w->d_walked, d_hostfirst, w->d_total, st);
			if (rc != XE_OK) return hip_fail(h, hipGetLastError(), "header walk (totals)");
			HIPCHK(h, hipMemcpyAsync(back + 32, (uint8_t *)w->d_total + 32, XE_WZERO_BYTES(nt) - 32, hipMemcpyDeviceToHost, st));
			walk_mark(h, w, st);
			WTICK(1);
			HIPCHK(h, hipStreamSynchronize(st));
			WTICK(2);
		}
		/* the copies add up in the first */
		for (size_t c = 1; c < XE_WCOPIES; c++) {
			const uint8_t *src = back + 32 + c * XE_WCOPY_BYTES(nt);
			uint64_t a, bsum;
			memcpy(&a, back + 32, 8);
			memcpy(&bsum, src, 8);
			a += bsum;
			memcpy(back + 32, &a, 8);
			for (size_t t = 0; t < 2 * (nt + 2); t++) cnt[t] += ((const uint32_t *)(src + 16))[t];
		}
		memcpy(&total, back + 32, 8);
		if (total + plan->nsf > 0xFFFFFFF0ull) return XE_EUNSUPPORTED;
		/* from 2^28 records on the fast kernel is out (32-bit column offsets) */
		if (total + plan->nsf >= fast_record_limit(h)) return XE_WALK_RETRY_ON_HOST;
		for (size_t t = 0; t < nt; t++) if (cntruns[t]) nactive++;
		/* the prefix sums of XE_WTUPLE templates are scanned together; a batch with
		 * more distinct templates than a few passes cover goes to the host walk */
		npass = (nactive + XE_WTUPLE - 1) / XE_WTUPLE;
		if (npass > 8) return XE_WALK_RETRY_ON_HOST;
		if (!npass) npass = 1;
		/* group table and template maps, side by side: one copy */
		const size_t gr_bytes = ALIGN_UP((nt + 2) * sizeof(*gr), 256), up_bytes = gr_bytes + npass * (nt + 1);
		gr = (struct xe_wgroup *)pin_get(h, up_bytes);
		if (!gr) return XE_WALK_RETRY_ON_HOST;   /* (the scratch is sized for them in xe_gpu_batch_prepare) */
		tmap = (uint8_t *)gr + gr_bytes;
		memset(tmap, 0xFF, npass * (nt + 1));
		w->maxchunk2 = 1;
		nactive = 0;
		for (size_t t = 0; t < nt; t++) {
			gr[t].pos0 = (uint32_t)work;
			gr[t].run0 = (uint32_t)nruns;
			gr[t].tile0 = (uint32_t)ntiles;
			gr[t].pad = 0;
			if (!cntruns[t]) continue;
			tmap[(nactive / XE_WTUPLE) * (nt + 1) + t] = (uint8_t)(nactive % XE_WTUPLE);
			nactive++;
			dev_total += cnt[t];
			work += ((uint64_t)cnt[t] + 63) / 64 * 64;
			nruns += cntruns[t];
			ntiles = work / 64;
			if (((uint32_t)tm[t].stride + 15) / 16 > w->maxchunk2) w->maxchunk2 = ((uint32_t)tm[t].stride + 15) / 16;
			if (!xe_plan_note_template(plan, &h->hs, (uint32_t)t, cnt[t])) return XE_ENOMEM;
		}
		gr[nt].pos0 = (uint32_t)work;
		gr[nt].run0 = (uint32_t)nruns;
		gr[nt].tile0 = (uint32_t)ntiles;
		gr[nt].pad = 0;
		if (work > 0xFFFFFF00ull) return XE_WALK_RETRY_ON_HOST;
		w->nfruns2 = (uint32_t)nruns;
		w->ntiles2 = (uint32_t)ntiles;
		{
			/* the two tables the fast kernel reads, the group table with the maps,
			 * and the scratch of the scans */
			const size_t wb = xe_walk_nf_work_bytes(n, w->nsets);
#define CARVE(BYTES) (off = ALIGN_UP(off, 256), off += (BYTES), off - (BYTES))
			const size_t o_fruns = CARVE((nruns + 1) * sizeof(struct xe_frun)), o_tiles = CARVE((ntiles + 1) * sizeof(struct xe_ftile)),
				o_up = CARVE(up_bytes), o_work = CARVE(wb);
#undef CARVE
			rc = dev_get(h, &w->tiles, off + 256);
			if (rc != XE_OK) return rc;
			fbase = (uint8_t *)w->tiles;
			w->d_fruns2 = (struct xe_frun *)(fbase + o_fruns);
			w->d_ftiles2 = (struct xe_ftile *)(fbase + o_tiles);
			w->d_groups = (struct xe_wgroup *)(fbase + o_up);
			walk_mark(h, w, st);
			HIPCHK(h, hipMemcpyAsync(fbase + o_up, gr, up_bytes, hipMemcpyHostToDevice, st));
			rc = xe_walk_nf_tables(&in, w->d_dgruns, w->nsets, w->d_walked, d_hostfirst, fbase + o_up + gr_bytes, (uint32_t)npass,
				(uint32_t)nactive, w->d_groups, fbase + o_work, w->d_fruns2, w->nfruns2, w->d_ftiles2, w->ntiles2, w->d_recbase, st);
			walk_mark(h, w, st);
			if (rc != XE_OK) return hip_fail(h, hipGetLastError(), "header walk (runs and tiles)");
		}
		/* the host walk numbered its records among themselves: move them to
		 * their places in arrival order over all datagrams */
		if (plan->sets.n) {
			struct xe_set *sets = (struct xe_set *)plan->sets.p;
			size_t k = 0;
			w->h_recbase = (uint32_t *)malloc(4 * (n + 1));
			if (!w->h_recbase) return XE_ENOMEM;
			HIPCHK(h, hipMemcpyAsync(w->h_recbase, w->d_recbase, 4 * (n + 1), hipMemcpyDeviceToHost, st));
			HIPCHK(h, hipStreamSynchronize(st));
			for (size_t i = 0; (i < n) && (k < plan->sets.n); i++) {
				while ((k < plan->sets.n) && (sets[k].first_rec < plan->dg_first_nf[i + 1])) {
					sets[k].first_rec = w->h_recbase[i] + (sets[k].first_rec - plan->dg_first_nf[i]);
					k++;
				}
			}
		}
		w->nrec2 = (size_t)dev_total;
		plan->nrec = (size_t)total;
		WTICK(3);
		if (dbg) fprintf(stderr, "device walk: totals enqueued %.2f, wait %.2f, scans / runs / tiles enqueued %.2f ms\n",
			tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2]);
	}
#undef WTICK
	return XE_OK;
}

extern "C" void
xe_gpu_batch_free(xe_batch *b)
{
	if (!b) return;
	(void)hipSetDevice(b->h->device);
	if (b->h->mirror_owner == b) b->h->mirror_owner = NULL;   /* (a cursor left open: its copy is no longer read) */
	if (b->ran && b->ev_ran) (void)hipEventSynchronize(b->ev_ran);
	else if (b->ran) (void)hipStreamSynchronize(b->last_stream);
	if (b->ev_ran) (void)hipEventDestroy(b->ev_ran);
	if (b->h && b->h->copy_stream) (void)hipStreamSynchronize(b->h->copy_stream);
	if (b->h && b->h->walk_stream && b->w.mode) (void)hipStreamSynchronize(b->h->walk_stream);
	if (b->ev_bulk) {
		/* (this batch's bytes only: the stream may already carry the next batch) */
		(void)hipEventSynchronize(b->ev_bulk);
		(void)hipEventDestroy(b->ev_bulk);
	}
	dev_put(b->h, b->d_pkt);
	dev_put(b->h, b->d_sets);
	dev_put(b->h, b->d_exps);
	dev_put(b->h, b->d_sfrecs);
	dev_put(b->h, b->d_out);
	for (int i = 0; i < 6; i++) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
	{
		void *extra[] = { b->d_fruns, b->d_ftiles, b->d_ssets, b->d_swstart,
			b->d_fplans, b->d_fops, b->d_fflds };
		for (size_t i = 0; i < sizeof(extra) / sizeof(extra[0]); i++) dev_put(b->h, extra[i]);
	}
	walk_free(b->h, &b->w);
	xe_plan_free(&b->plan);
	free(b);
}

/* Split the runs between the two kernels and lay out the fast kernel's tiles.
 * A run is "regular" when its template is fixed-length without repeated or
 * string elements, none of its reads can pass the datagram end and its
 * exporter has no devices.conf entry.  Regular runs are grouped by template
 * (stable, so arrival order is kept inside a group) and each group is padded
 * to whole 64-record tiles: a wave then works on one template only. */
/* can the fast kernel decode records of this template (fixed length, no
 * repeated, string or over-long element, staging fits the LDS)? */
/* copies of the per-object counters a block of the kernels keeps in LDS: as
 * many as 3 KB hold, 16 at most (0: more objects than LDS counters are kept for) */
static uint32_t
lds_counter_copies(const xe_gpu *h)
{
	uint32_t copies = 1;
	if (h->hs.nmos > XE_MAX_LDS_COUNTER_MOS) return 0;
	while ((copies < 16) && ((size_t)h->hs.nmos * 24 * copies * 2 <= 3072)) copies *= 2;
	if (h->diag_counter_copies) {       /* diagnostic builds only */
		const int want = h->diag_counter_copies;
		if ((want == 1) || (want == 2) || (want == 4) || (want == 8) || (want == 16)) copies = (uint32_t)want;
	}
	return copies;
}

static int
tmpl_fast(const xe_gpu *h, const struct xe_tmpl *t, const struct xe_tent *te)
{
	/* a block stages 4 tiles of 64 records in LDS, beside the counters (all
	 * their copies) and the atom-table copy: long records only while all of it fits */
	const size_t cnt_bytes = (size_t)lds_counter_copies(h) * h->hs.nmos * 24;
	const size_t motab_bytes = (h->motab_words * 8 <= 8192 + 64) ? h->motab_words * 8 : 0;
	const size_t stage_bytes = (size_t)4 * 64 * 4 * (4 * (((size_t)t->stride + 15) / 16) + 1);
	int ok = h->fast_ok && !t->varlen && !t->repeats
		&& (t->stride > 0) && (t->stride <= XE_FAST_MAX_STRIDE)
		&& (stage_bytes + cnt_bytes + motab_bytes <= XE_FAST_LDS_BUDGET);
	for (uint32_t j = 0; ok && (j < t->nfields); j++) {
		const struct xe_tent *e = &te[t->ent_base + j];
		if ((e->field != XE_F_NONE) && (e->flags & XE_TE_STR)) ok = 0;
		if (((e->len > 16) || (e->len == 0)) && (e->flags & XE_TE_VALID) && (e->field != XE_F_NONE)) ok = 0;
	}
	return ok;
}

static int
ftile_cmp(const void *a, const void *b)
{
	const struct xe_ftile *x = (const struct xe_ftile *)a, *y = (const struct xe_ftile *)b;
	return (x->pad > y->pad) - (x->pad < y->pad);
}

static int
build_work_lists(xe_gpu *h, xe_batch *b, size_t total_bytes)
{
	const struct xe_set *sets = (const struct xe_set *)b->plan.sets.p;
	const struct xe_tmpl *tm = (const struct xe_tmpl *)h->hs.dev_tmpls.p;
	const struct xe_tent *te = (const struct xe_tent *)h->hs.dev_tents.p;
	const struct xe_exporter *ex = (const struct xe_exporter *)b->plan.exporters.p;
	size_t ns = b->plan.sets.n, nt = h->hs.dev_tmpls.n, i;
	uint8_t *tfast = NULL, *is_fast = NULL;
	uint64_t *gcount = NULL, *gpos = NULL;
	struct xe_frun *fr = NULL;
	struct xe_set *ss = NULL;
	uint32_t *sw = NULL, *rtmpl = NULL;
	struct xe_ftile *ft = NULL;
	size_t nfast = 0, nslow = 0, ntiles = 0, work = 0, slowrec = 0;
	/* the fast kernel addresses packet bytes with 32-bit offsets */
	const int small = (total_bytes + 64) < 0xFFFF0000ull;
	/* the fast kernel addresses a column element with a 32-bit byte offset
	 * (16-byte columns: below 2^28 records); larger batches take the general
	 * kernel.  The environment variable is a diagnostic: time the general kernel */
	const int force_general = h->diag_force_general || (b->ntot >= fast_record_limit(h));
	uint32_t maxchunk = 1, R, tilesz;
	size_t cnt_bytes = (size_t)lds_counter_copies(h) * h->hs.nmos * 24;
	int rc = XE_ENOMEM;

	tfast = (uint8_t *)calloc(nt + 1, 1);
	is_fast = (uint8_t *)calloc(ns + 1, 1);
	gcount = (uint64_t *)calloc(nt + 1, sizeof(uint64_t));
	gpos = (uint64_t *)calloc(nt + 1, sizeof(uint64_t));
	if (!tfast || !is_fast || !gcount || !gpos) goto out;
	for (i = 0; i < nt; i++) tfast[i] = (uint8_t)(small && tmpl_fast(h, &tm[i], te));
	for (i = 0; i < ns; i++) {
		/* an exporter with a devices.conf entry that only sets the sampling rate
		 * (no mark filters, nothing dropped) needs no per-record rule check */
		const int dev = ex[sets[i].exporter].device;
		/* (a run names its devices.conf entry in 15 bits) */
		const int plain_dev = (dev < 0) || (dev < 0x7FFE);
		int ok = !force_general && tfast[sets[i].tmpl] && !(sets[i].flags & (XE_SET_CHECKED | XE_SET_PARTIAL))
			&& plain_dev;
		is_fast[i] = (uint8_t)ok;
		if (ok) {
			uint32_t nc = ((uint32_t)tm[sets[i].tmpl].stride + 15) / 16;
			if (nc > maxchunk) maxchunk = nc;
			nfast++;
			gcount[sets[i].tmpl]++;
		} else { nslow++; slowrec += sets[i].nrec; }
	}
	/* records per lane: one measured fastest on MI355X (more records per lane
	 * amortise the scalar work but lengthen every dependent chain); 2 and 4
	 * stay available while they fit the LDS budget (4 waves x R x 64 records
	 * x 16 * maxchunk bytes) */
	R = 1;
	if (h->diag_fast_r) {
		int want = h->diag_fast_r;   /* diagnostic override */
		if ((want == 1) || (want == 2) || (want == 4)) R = (uint32_t)want;
		while ((R > 1) && ((size_t)4 * R * 64 * 4 * (4 * maxchunk + 1) + cnt_bytes + 8192 > XE_FAST_LDS_BUDGET)) R >>= 1;
	}
	tilesz = 64 * R;
	if ((nfast + 1) * sizeof(*fr) > h->stage_cap) {
		const size_t want = (nfast + 1) * sizeof(*fr) * 5 / 4;
		if (h->stage) (void)hipHostFree(h->stage);
		h->stage = NULL;
		h->stage_cap = 0;
		if (hipHostMalloc(&h->stage, want, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			h->stage = NULL;
			rc = XE_ENOMEM;
			goto out;
		}
		h->stage_cap = want;
	}
	fr = (struct xe_frun *)h->stage;
	ss = (struct xe_set *)malloc((nslow + 1) * sizeof(*ss));
	sw = (uint32_t *)malloc((nslow + 1) * sizeof(uint32_t));
	if (!fr || !ss || !sw) goto out;
	/* group positions */
	{
		uint64_t acc = 0;
		for (i = 0; i < nt; i++) { gpos[i] = acc; acc += gcount[i]; }
	}
	{
		size_t k = 0;
		uint64_t w = 0;
		for (i = 0; i < ns; i++) {
			if (is_fast[i]) {
				struct xe_frun *r = &fr[gpos[sets[i].tmpl]++];
				const struct xe_exporter *e = &ex[sets[i].exporter];
				uint32_t ipw, idw;
				memcpy(&ipw, e->dev_ip, 4);
				memcpy(&idw, e->dev_id, 4);
				r->data_off = (uint32_t)sets[i].data_off;
				r->first_rec = sets[i].first_rec;
				r->start = sets[i].tmpl;   /* template for now, position below */
				r->nrec = sets[i].nrec;
				/* bit 31: the exporter has a devices.conf entry (mark present, value 0) */
				/* bit 31: the exporter has a devices.conf entry; bits 16-30: that entry + 1
				 * when it has mark filters to run or drops what they leave unmarked */
				r->stride = sets[i].stride | ((e->device >= 0) ? 0x80000000u : 0u);
				if ((e->device >= 0) && (h->hs.devs[e->device].nmarks || h->hs.devs[e->device].skip_unmarked)) {
					r->stride |= (h->dev_canon[e->device] + 1u) << 16;
				}
				r->rate = e->sampling_rate;
				r->dev_ip = ntohl(ipw);
				r->dev_id = ntohl(idw);
			} else { ss[k] = sets[i]; sw[k] = (uint32_t)w; w += sets[i].nrec; k++; }
		}
	}
	/* positions with per-group padding (r->start holds the template so far),
	 * then the tile tables */
	rtmpl = (uint32_t *)malloc((nfast + 1) * sizeof(uint32_t));
	if (!rtmpl) goto out;
	for (i = 0; i < nfast; i++) {
		rtmpl[i] = fr[i].start;
		if (i && (rtmpl[i] != rtmpl[i - 1])) work = (work + tilesz - 1) / tilesz * tilesz;
		if (work + fr[i].nrec > 0xFFFFFF00u) { rc = XE_EUNSUPPORTED; goto out; }
		fr[i].start = (uint32_t)work;
		work += fr[i].nrec;
	}
	ntiles = (work + tilesz - 1) / tilesz;
	ft = (struct xe_ftile *)calloc(ntiles + 1, sizeof(*ft));
	if (!ft) goto out;
	{
		size_t sidx = 0;
		for (size_t t = 0; t < ntiles; t++) {
			uint64_t g0 = (uint64_t)t * tilesz;
			while ((sidx + 1 < nfast) && (fr[sidx + 1].start <= g0)) sidx++;
			ft[t].first_run = (uint32_t)sidx;
			ft[t].tmpl = rtmpl[sidx];
			ft[t].base = (uint32_t)g0;
			/* sort key: batch index of the tile's first record */
			ft[t].pad = fr[sidx].first_rec + (uint32_t)(g0 - fr[sidx].start);
		}
		/* every template's tiles are in arrival order already: merge the
		 * groups by that key (a few templates: pick the smallest head; many:
		 * sort) */
		{
			size_t gstart[17], gend[17], ng = 0, t;
			for (t = 0; t < ntiles; t++) {
				if (!t || (ft[t].tmpl != ft[t - 1].tmpl)) {
					if (ng == 16) { ng = 17; break; }
					if (ng) gend[ng - 1] = t;
					gstart[ng++] = t;
				}
			}
			if (ng && (ng <= 16)) gend[ng - 1] = ntiles;
			if (ng > 16) {
				qsort(ft, ntiles, sizeof(*ft), ftile_cmp);
			} else if (ng > 1) {
				struct xe_ftile *m = (struct xe_ftile *)malloc((ntiles + 1) * sizeof(*m));
				if (!m) goto out;
				for (t = 0; t < ntiles; t++) {
					size_t best = 17, g;
					for (g = 0; g < ng; g++) {
						if (gstart[g] == gend[g]) continue;
						if ((best == 17) || (ft[gstart[g]].pad < ft[gstart[best]].pad)) best = g;
					}
					m[t] = ft[gstart[best]++];
				}
				free(ft);
				ft = m;
			}
		}
	}
	rc = upload_pooled(h, &b->d_fruns, fr, nfast * sizeof(*fr));
	if (rc == XE_OK) rc = upload_pooled(h, &b->d_ftiles, ft, ntiles * sizeof(*ft));
	if (rc == XE_OK) rc = upload_pooled(h, &b->d_ssets, ss, nslow * sizeof(*ss));
	if (rc == XE_OK) rc = upload_pooled(h, &b->d_swstart, sw, nslow * 4);
	if (rc == XE_OK) {
		b->kp.fruns = (const struct xe_frun *)b->d_fruns;
		b->kp.ftiles = (const struct xe_ftile *)b->d_ftiles;
		b->kp.nfruns = (uint32_t)nfast;
		b->kp.ntiles = (uint32_t)ntiles;
		b->kp.ssets = (const struct xe_set *)b->d_ssets;
		b->kp.swstart = (const uint32_t *)b->d_swstart;
		b->kp.nssets = (uint32_t)nslow;
		b->kp.nslowrec = (uint32_t)slowrec;
		b->kp.fast_r = R;
		b->kp.fast_dwords = 4 * maxchunk + 1;   /* + 1: field reads look one dword ahead */
		b->kp.fast_want_dev = (uint32_t)(h->want_dev || h->fwm_devip);
		b->kp.nterms = h->nterms;
		b->kp.fterms = (const struct xe_fterm *)h->d_fterms;
		b->kp.tneed = (const uint32_t *)h->d_tneed;
		b->kp.ntneed = h->ntneed;
		b->kp.fprogs = (const struct xe_fprog *)h->d_fprogs;
		b->kp.nfprogs = h->nfprogs;
		b->kp.cjgb = (const uint32_t *)h->d_cjgb;
		b->kp.cjtab = (const uint64_t *)h->d_cjtab;
		b->kp.ncjgroups = (h->ncjgroups + 3) & ~3u;
		b->kp.rgroups = (const struct xe_rgroup *)h->d_rgroups;
		b->kp.rgdata = (const uint32_t *)h->d_rgdata;
		b->kp.nrgroups = (h->nrgroups + 3) & ~3u;
		b->kp.nrgroups_used = h->nrgroups;
		b->kp.cjderived = (const struct xe_cjderived *)h->d_cjderived;
		b->kp.cjcand = (const uint64_t *)h->d_cjcand;
		b->kp.ncjderived = h->ncjderived;
		{
			uint32_t k = 0;
			for (int w = 0; w < XE_LK_WORDS; w++) b->kp.all_probes[w] = ~0ull;
			for (uint32_t pr = 0; (pr < h->nlookups) && (k < XE_MAX_PROBES); pr++) {
				if (!h->h_probes[pr].root_off) continue;
				b->kp.all_probes[k >> 3] = (b->kp.all_probes[k >> 3] & ~(0xFFull << (8 * (k & 7))))
					| ((uint64_t)pr << (8 * (k & 7)));
				k++;
			}
			b->kp.per_object = h->fast_ok ? 0u : 1u;
			b->kp.lw = h->lw;
		}
		b->kp.motab = (const uint64_t *)h->d_motab;
		b->kp.motab_words = (uint32_t)h->motab_words;
		b->kp.rg_lds = h->rg_lds;
		b->kp.motab_lds = (h->motab_words * 8 <= 8192 + 64) ? 1u : 0u;
		b->kp.probes = (const struct xe_probe *)h->d_lookups;
		b->kp.nprobes = h->nlookups;
		b->kp.lds_counters = lds_counter_copies(h);
	}
out:
	free(tfast); free(is_fast); free(gcount); free(gpos);
	free(ss); free(sw); free(ft); free(rtmpl);
	return rc;
}

/* datagram bytes on their way to the device ahead of the batch they belong to */
struct xe_upload
{
	xe_gpu *h;
	void *d_pkt;
	hipEvent_t ev;
	size_t bytes;
};

static int
prepare_impl(xe_gpu *h, const uint8_t *data, const uint64_t *offsets, const uint32_t *src_ip4,
	const uint8_t *kinds, size_t n, xe_batch **out, int walk_mode, struct xe_upload *up);

extern "C" int
xe_gpu_batch_prepare(xe_gpu *h, const uint8_t *data, const uint64_t *offsets, const uint32_t *src_ip4,
	const uint8_t *kinds, size_t n, xe_batch **out)
{
	int rc;
	if (!h) return XE_EINVAL;
	rc = prepare_impl(h, data, offsets, src_ip4, kinds, n, out, h->device_walk, NULL);
	/* a batch the device walk cannot take (2^28 records and more) */
	if (rc == XE_WALK_RETRY_ON_HOST) rc = prepare_impl(h, data, offsets, src_ip4, kinds, n, out, 0, NULL);
	return rc;
}

extern "C" int
xe_gpu_upload_begin(xe_gpu *h, const uint8_t *data, size_t bytes, xe_upload **out)
{
	struct xe_upload *u;
	hipError_t e = hipSuccess;
	int rc;

	if (!h || !out || (bytes && !data)) return XE_EINVAL;
	*out = NULL;
	HIPCHK(h, hipSetDevice(h->device));
	if (!h->bulk_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->bulk_stream, hipStreamNonBlocking));
	u = (struct xe_upload *)calloc(1, sizeof(*u));
	if (!u) return XE_ENOMEM;
	u->h = h;
	u->bytes = bytes;
	rc = dev_get(h, &u->d_pkt, bytes + 64);
	if (rc != XE_OK) { free(u); return rc; }
	if (bytes) e = hipMemcpyAsync(u->d_pkt, data, bytes, hipMemcpyHostToDevice, h->bulk_stream);
	if (e == hipSuccess) e = hipMemsetAsync((uint8_t *)u->d_pkt + bytes, 0, 64, h->bulk_stream);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&u->ev, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventRecord(u->ev, h->bulk_stream);
	if (e != hipSuccess) {
		(void)hipStreamSynchronize(h->bulk_stream);
		if (u->ev) (void)hipEventDestroy(u->ev);
		dev_put(h, u->d_pkt);
		free(u);
		return hip_fail(h, e, "hipMemcpyAsync(packets)");
	}
	*out = u;
	return XE_OK;
}

extern "C" void
xe_gpu_upload_free(xe_upload *u)
{
	if (!u) return;
	(void)hipSetDevice(u->h->device);
	if (u->ev) { (void)hipEventSynchronize(u->ev); (void)hipEventDestroy(u->ev); }
	dev_put(u->h, u->d_pkt);
	free(u);
}

extern "C" int
xe_gpu_batch_prepare_uploaded(xe_gpu *h, xe_upload *up, const uint8_t *data, const uint64_t *offsets, const uint32_t *src_ip4,
	const uint8_t *kinds, size_t n, xe_batch **out)
{
	int rc;
	if (!h || !up || (up->h != h)) return XE_EINVAL;
	if (up->bytes != (n ? (size_t)offsets[n] : 0)) {
		snprintf(h->err, sizeof(h->err), "the upload holds %zu bytes, the batch's datagrams %zu", up->bytes, n ? (size_t)offsets[n] : 0);
		xe_gpu_upload_free(up);
		return XE_EINVAL;
	}
	/* (the upload belongs to the batch from here on, whatever becomes of it) */
	rc = prepare_impl(h, data, offsets, src_ip4, kinds, n, out, h->device_walk, up);
	if (rc == XE_WALK_RETRY_ON_HOST) rc = prepare_impl(h, data, offsets, src_ip4, kinds, n, out, 0, NULL);
	return rc;
}

extern "C" int
xe_gpu_batch_device_walked(const xe_batch *b, size_t *netflow_datagrams, size_t *netflow_records, size_t *sflow_records)
{
	size_t nd = 0;
	if (!b) return XE_EINVAL;
	if (b->w.mode & 2) for (size_t i = 0; i < b->w.n; i++) nd += b->w.h_walked[i] ? 1 : 0;
	if (netflow_datagrams) *netflow_datagrams = nd;
	if (netflow_records) *netflow_records = b->w.nrec2;
	if (sflow_records) *sflow_records = (b->w.mode & 1) ? b->plan.nsf : 0;
	return XE_OK;
}

extern "C" int
xe_gpu_set_device_walk(xe_gpu *h, int mode)
{
	if (!h || (mode < 0) || (mode > 3)) return XE_EINVAL;
	h->device_walk = mode;
	return XE_OK;
}

extern "C" int
xe_gpu_set_list_root_bits(xe_gpu *h, int bits)
{
	if (!h || ((bits != 0) && (bits != 16) && (bits != 24))) return XE_EINVAL;
	h->list_root_bits = bits;
	h->committed = 0;
	return XE_OK;
}

extern "C" int
xe_gpu_set_walk_timing(xe_gpu *h, int enable)
{
	if (!h) return XE_EINVAL;
	h->walk_timing = enable ? 1 : 0;
	return XE_OK;
}

extern "C" int
xe_gpu_batch_walk_ms(xe_batch *b, float *ms)
{
	float sum = 0.0f;
	if (!b || !ms) return XE_EINVAL;
	*ms = -1.0f;
	if (!b->w.nwev) return XE_OK;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	for (int i = 0; i + 1 < b->w.nwev; i += 2) {
		float t = 0.0f;
		HIPCHK(b->h, hipEventSynchronize(b->w.wev[i + 1]));
		HIPCHK(b->h, hipEventElapsedTime(&t, b->w.wev[i], b->w.wev[i + 1]));
		sum += t;
	}
	*ms = sum;
	return XE_OK;
}

static int
prepare_impl(xe_gpu *h, const uint8_t *data, const uint64_t *offsets, const uint32_t *src_ip4,
	const uint8_t *kinds, size_t n, xe_batch **out, int walk_mode, struct xe_upload *up)
{
	xe_batch *b;
	int rc;
	size_t total, off, f;
	const int dbg = h ? h->dbg : 0;
	double tm_[8];
	int ntm_ = 0;
#define XE_TICK() do { if (dbg && (ntm_ < 8)) { struct timespec ts_; clock_gettime(CLOCK_MONOTONIC, &ts_); \
	tm_[ntm_++] = ts_.tv_sec * 1e3 + ts_.tv_nsec * 1e-6; } } while (0)

	if (!h || !out || (n && (!data || !offsets || !src_ip4))) { xe_gpu_upload_free(up); return XE_EINVAL; }
	*out = NULL;
	if (!h->committed) {
		snprintf(h->err, sizeof(h->err), "xe_gpu_commit() must follow configuration changes");
		xe_gpu_upload_free(up);
		return XE_ESTATE;
	}
	if (hipSetDevice(h->device) != hipSuccess) { xe_gpu_upload_free(up); return hip_fail(h, hipGetLastError(), "hipSetDevice"); }
	b = (xe_batch *)calloc(1, sizeof(*b));
	if (!b) { xe_gpu_upload_free(up); return XE_ENOMEM; }
	b->h = h;
	pin_begin(h, 14 * (n + 8) + (h->hs.dev_tmpls.n + 8) * (sizeof(struct xe_fplan) + XE_NFIELDS * 4 + sizeof(struct xe_wgroup) + 8 + 256)
		+ (h->hs.dev_tents.n + 16) * sizeof(struct xe_fop) + 16 * 256);
	/* The datagram bytes start moving to the device before the header walk:
	 * neither needs the other (from page-locked memory the copy runs beside
	 * the walk; from ordinary memory the call returns when it is done).
	 * 64 zero bytes follow them so that an unchecked wide load near the end
	 * stays inside the allocation. */
	total = n ? (size_t)offsets[n] : 0;
	b->payload_bytes = total;
	if (up) {
		/* the bytes were sent ahead (xe_gpu_upload_begin): the batch takes the
		 * buffer and the event over; the walk reads the device copy, behind the
		 * event, and the link carries nothing but the next upload meanwhile */
		b->d_pkt = up->d_pkt;
		b->ev_bulk = up->ev;
		b->sent_ahead = 1;
		free(up);
		up = NULL;
	} else {
		rc = upload_pooled(h, &b->d_pkt, NULL, total + 64);
		if (rc != XE_OK) { xe_gpu_batch_free(b); return rc; }
	}
	if (!b->sent_ahead) {
		hipError_t e = hipSuccess;
		const hipStream_t bulk = h->overlap ? h->bulk_stream : h->copy_stream;
		if (total) e = hipMemcpyAsync(b->d_pkt, data, total, hipMemcpyHostToDevice, bulk);
		if (e == hipSuccess) e = hipMemsetAsync((uint8_t *)b->d_pkt + total, 0, 64, bulk);
		if ((e == hipSuccess) && h->overlap) {
			/* whoever reads the bytes on the device waits for this event; the call itself does not */
			e = hipEventCreateWithFlags(&b->ev_bulk, hipEventDisableTiming);
			if (e == hipSuccess) e = hipEventRecord(b->ev_bulk, bulk);
		}
		if (e != hipSuccess) {
			if (h->overlap) (void)hipStreamSynchronize(h->bulk_stream);
			(void)hipStreamSynchronize(h->copy_stream);
			xe_gpu_batch_free(b);
			return hip_fail(h, e, "hipMemcpyAsync(packets)");
		}
	}
	XE_TICK();
	/* which headers the device walks itself: all sFlow datagrams; Netflow ones
	 * while the fast kernel is in use at all (its 32-bit packet offsets, no
	 * diagnostic override of its shape) */
	{
		int has_sf = 0, has_nf = 0, ordered = 1;
		for (size_t i = 0; i < n; i++) {
			if (kinds && (kinds[i] == XE_KIND_SFLOW)) has_sf = 1; else has_nf = 1;
			if (offsets[i + 1] < offsets[i]) ordered = 0;
		}
		if (!ordered) {
			/* the kernels trust the offsets: a datagram may not end before it starts */
			snprintf(h->err, sizeof(h->err), "datagram offsets must not decrease");
			(void)hipStreamSynchronize(h->copy_stream);
			xe_gpu_batch_free(b);
			return XE_EINVAL;
		}
		if (!has_sf || (h->hs.ndevs >= 0xFFFFu)) walk_mode &= ~1;   /* (a record's entry names its devices.conf entry in 16 bits) */
		if (!has_nf || !h->fast_ok || ((total + 64) >= 0xFFFF0000ull) || h->diag_force_general || h->diag_fast_r) walk_mode &= ~2;
		/* (the per-template counts come back through a small page-locked buffer) */
		if ((XE_WZERO_BYTES(h->hs.dev_tmpls.n) > XE_WSTAGE_BYTES) || (h->hs.dev_tmpls.n >= 0xFFF0u)) walk_mode &= ~2;
		if (n >= 0x7FFFFFF0ull) walk_mode = 0;
		/* (run indices, XE_WALK_SETS per datagram, are sorted as 32-bit values) */
		if (n >= (0xFFFFFFF0ull / XE_WALK_SETS)) walk_mode &= ~2;
	}
	if (walk_mode) {
		rc = walk_begin(h, b, data, offsets, src_ip4, kinds, n, walk_mode);
		if (rc != XE_OK) { (void)hipStreamSynchronize(h->copy_stream); xe_gpu_batch_free(b); return rc; }
		if (walk_mode & 2) xe_host_undo_begin(&h->hs);
		rc = walk_host_part(h, b, data, offsets, src_ip4, n);
	} else {
		rc = xe_prepass(&h->hs, data, offsets, src_ip4, kinds, n, &b->plan);
	}
	XE_TICK();
	if (rc != XE_OK) {
		snprintf(h->err, sizeof(h->err), "pre-pass failed: %s%s", xe_gpu_strerror(rc),
			(rc == XE_EUNSUPPORTED) ? " (the batch would hold more than 2^32 records: prepare fewer datagrams at a time)" : "");
		if (walk_mode & 2) xe_host_undo_end(&h->hs);
		(void)hipStreamSynchronize(h->copy_stream);
		if (b->w.mode && b->w.st) (void)hipStreamSynchronize(b->w.st);
		if (b->ev_bulk) { (void)hipEventSynchronize(b->ev_bulk); (void)hipEventDestroy(b->ev_bulk); }
		dev_put(h, b->d_pkt);
		walk_free(h, &b->w);
		free(b);
		return rc;
	}
	rc = sync_templates(h);
	if ((rc == XE_OK) && walk_mode) {
		rc = walk_finish(h, b);
		/* the whole batch goes to the host walk: with the template store as it
		 * was when the batch came in */
		if ((rc == XE_WALK_RETRY_ON_HOST) && (walk_mode & 2)) xe_host_undo_rollback(&h->hs);
		if (rc == XE_EUNSUPPORTED) snprintf(h->err, sizeof(h->err),
			"the batch would hold more than 2^32 records: prepare fewer datagrams at a time");
	}
	if (walk_mode & 2) xe_host_undo_end(&h->hs);
	if (rc != XE_OK) { (void)hipStreamSynchronize(h->copy_stream); xe_gpu_batch_free(b); return rc; }
	b->ntot = b->plan.nrec + b->plan.nsf;
	XE_TICK();
	/* (the sets themselves stay on the host: the kernels read the run lists
	 * build_work_lists() makes of them) */
	rc = upload_pooled(h, &b->d_exps, b->plan.exporters.p, b->plan.exporters.n * sizeof(struct xe_exporter));
	if ((rc == XE_OK) && !(b->w.mode & 1)) rc = upload_pooled(h, &b->d_sfrecs, b->plan.sfrecs.p, b->plan.sfrecs.n * sizeof(struct xe_sfrec));
	if (rc != XE_OK) { xe_gpu_batch_free(b); return rc; }

	memset(&b->kp, 0, sizeof(b->kp));
	XE_TICK();
	rc = build_work_lists(h, b, total);
	if (rc != XE_OK) { xe_gpu_batch_free(b); return rc; }
	if (b->w.ntiles2) {
		b->kp.fruns2 = b->w.d_fruns2;
		b->kp.ftiles2 = b->w.d_ftiles2;
		b->kp.nfruns2 = b->w.nfruns2;
		b->kp.ntiles2 = b->w.ntiles2;
		if (4 * b->w.maxchunk2 + 1 > b->kp.fast_dwords) b->kp.fast_dwords = 4 * b->w.maxchunk2 + 1;
	}
	XE_TICK();

	/* output layout */
	off = 0;
	b->masks_off = off;    off = ALIGN_UP(off + b->ntot * h->mask_words * 8, 256);
	b->nparsed_off = off;  off = ALIGN_UP(off + b->ntot, 256);
	b->marks_off = off;    if (h->hs.ndevs) off = ALIGN_UP(off + b->ntot * 4, 256);
	b->counters_off = off; off = ALIGN_UP(off + (size_t)(h->nmo ? h->nmo : 1) * 24, 256);
	b->sfhas_off = off;    off = ALIGN_UP(off + b->plan.nsf * 4, 256);
	b->sfrate_off = off;   if (b->w.mode & 1) off = ALIGN_UP(off + b->plan.nsf * 4, 256);
	b->rates_off = off;    if (h->nfwm) off = ALIGN_UP(off + b->ntot * 4, 256);
	b->devips_off = off;   if (h->fwm_devip) off = ALIGN_UP(off + b->ntot * 4, 256);
	for (f = 0; f < XE_NFIELDS; f++) {
		b->kp.col_off[f] = ~0ull;
		b->kp.vsize_off[f] = ~0ull;
	}
	if (b->plan.nsf) {
		/* fields the sFlow decoder can set (sflow.c:35-77, sflow-impl.h:97-260) */
		static const int sf_fields[] = { XF_in_bytes, XF_in_pkts, XF_protocol, XF_src_tos, XF_dst_tos,
			XF_tcp_flags, XF_l4_src_port, XF_ip4_src_addr, XF_input_snmp, XF_l4_dst_port,
			XF_ip4_dst_addr, XF_output_snmp, XF_ip6_src_addr, XF_ip6_dst_addr, XF_icmp_type,
			XF_min_ttl, XF_max_ttl, XF_src_mac, XF_dst_mac, XF_src_vlan, XF_dst_vlan,
			XF_src_dot1q_cvlan, XF_dst_dot1q_cvlan };
		for (size_t i = 0; i < sizeof(sf_fields) / sizeof(sf_fields[0]); i++) {
			b->plan.field_mask[sf_fields[i] >> 6] |= 1ull << (sf_fields[i] & 63);
		}
	}
	for (f = 0; f < XE_NTABLE_FIELDS; f++) {
		if (!((b->plan.field_mask[f >> 6] >> (f & 63)) & 1)) continue;
		b->kp.col_off[f] = off;
		off = ALIGN_UP(off + b->ntot * xe_fields[f].col_width, 256);
	}
	/* per-record sizes: strings, and anything some template declares variable-length */
	{
		const struct xe_tmpl *tm = (const struct xe_tmpl *)h->hs.dev_tmpls.p;
		const struct xe_tent *te = (const struct xe_tent *)h->hs.dev_tents.p;
		uint64_t need[2] = {0, 0};
		for (size_t t = 0; t < h->hs.dev_tmpls.n; t++) {
			for (uint32_t j = 0; j < tm[t].nfields; j++) {
				const struct xe_tent *e = &te[tm[t].ent_base + j];
				if ((e->field != XE_F_NONE) && ((e->len == 0xFFFF) || (e->flags & XE_TE_STR))) {
					need[e->field >> 6] |= 1ull << (e->field & 63);
				}
			}
		}
		for (f = 0; f < XE_NTABLE_FIELDS; f++) {
			if (b->kp.col_off[f] == ~0ull) continue;
			if (!((need[f >> 6] >> (f & 63)) & 1)) continue;
			b->kp.vsize_off[f] = off;
			off = ALIGN_UP(off + b->ntot * 2, 256);
		}
	}
	b->out_bytes = off ? off : 256;
	{
		hipError_t e;
		rc = dev_get(h, &b->d_out, b->out_bytes);
		if (rc != XE_OK) { xe_gpu_batch_free(b); return rc; }
		/* records the kernels do not reach (none by construction) read as empty.
		 * On the handle's copy stream, which prepare waits for before it
		 * returns: nothing is left to the ordering rules of the null stream,
		 * and other handles' streams are not held up */
		/* (with streaming stores: measured on MI355X, the runtime's fill of a C2 batch's
		 * 245 MB leaves the decode kernel that follows 0.281 ms, this one 0.249 ms -
		 * the list tables and the datagrams stay in the caches) */
		e = (xe_launch_zero_fill(b->d_out, b->out_bytes, h->copy_stream) == XE_OK) ? hipSuccess : hipErrorLaunchFailure;
		if (e != hipSuccess) { (void)hipStreamSynchronize(h->copy_stream); xe_gpu_batch_free(b); return hip_fail(h, e, "clearing the outputs"); }
	}

	XE_TICK();
	/* decode plans of the fast kernel: per template the micro-ops of its
	 * known, valid fields with this batch's column addresses, and the
	 * location of every field for the filter terms */
	{
		const struct xe_tmpl *tm = (const struct xe_tmpl *)h->hs.dev_tmpls.p;
		const struct xe_tent *te = (const struct xe_tent *)h->hs.dev_tents.p;
		size_t nt = h->hs.dev_tmpls.n, ne = h->hs.dev_tents.n, nops = 0;
		const size_t pl_bytes = (nt ? nt : 1) * sizeof(struct xe_fplan), fo_bytes = (ne + 4) * sizeof(struct xe_fop),
			ff_bytes = (nt ? nt : 1) * XE_NFIELDS * sizeof(uint32_t);
		/* in page-locked scratch when there is room: the three uploads then do not block */
		const size_t pin_mark = h->pin_used;
		struct xe_fplan *pl = (struct xe_fplan *)pin_get(h, pl_bytes);
		struct xe_fop *fo = (struct xe_fop *)pin_get(h, fo_bytes);   /* + 4: the kernel loads four at a time */
		uint32_t *ff = (uint32_t *)pin_get(h, ff_bytes);
		const int tables_pinned = pl && fo && ff;
		if (!tables_pinned) {
			h->pin_used = pin_mark;
			pl = (struct xe_fplan *)malloc(pl_bytes);
			fo = (struct xe_fop *)malloc(fo_bytes);
			ff = (uint32_t *)malloc(ff_bytes);
		}
		if (!pl || !fo || !ff) { if (!tables_pinned) { free(pl); free(fo); free(ff); } xe_gpu_batch_free(b); return XE_ENOMEM; }
		memset(pl, 0, pl_bytes);
		memset(fo, 0, fo_bytes);
		memset(ff, 0, ff_bytes);
		for (size_t t = 0; t < nt; t++) {
			uint32_t *fld = ff + t * XE_NFIELDS;
			pl[t].op_base = (uint32_t)nops;
			pl[t].nparsed = tm[t].nfields;
			pl[t].nchunk = ((uint32_t)tm[t].stride + 15) / 16;
			for (int f = 0; f < XE_NFIELDS; f++) {
				uint32_t mx = xe_fields[f].size_max;
				fld[f] = ((mx > 31) ? 31u : mx) << 21;
			}
			fld[XF_dev_ip] |= XE_FFLD_PRESENT | XE_FFLD_VIRTUAL;
			fld[XF_dev_id] |= XE_FFLD_PRESENT | XE_FFLD_VIRTUAL;
			for (int pass = 0; pass < 6; pass++) {
				for (uint32_t j = 0; j < tm[t].nfields; j++) {
					const struct xe_tent *e = &te[tm[t].ent_base + j];
					int f = e->field;
					uint32_t cw, lg = 0, kind, cls;
					if ((f == XE_F_NONE) || !(e->flags & XE_TE_VALID) || (e->len > 16) || (e->len == 0)
						|| (b->kp.col_off[f] == ~0ull)) continue;
					cw = xe_fields[f].col_width;
					while ((1u << lg) < cw) lg++;
					kind = lg;
					if (e->len == cw) kind |= XE_FOP_COPY;
					if (xe_fields[f].size_max == 6) kind |= XE_FOP_MAC;
					cls = (kind & XE_FOP_COPY) ? lg : 5;
					if ((int)cls != pass) continue;
					fld[f] = (fld[f] & (31u << 21)) | XE_FFLD_PRESENT | ((uint32_t)e->len << 16) | e->off;
					fo[nops].w0 = (uint64_t)e->off | ((uint64_t)e->len << 16) | ((uint64_t)kind << 24);
					fo[nops].w1 = (uint64_t)(uintptr_t)((uint8_t *)b->d_out + b->kp.col_off[f]);
					nops++;
					pl[t].cnt[pass]++;
				}
			}
			/* probes that can match at all: the field is there, the tables are not empty */
			{
				uint32_t k = 0;
				for (int w = 0; w < XE_LK_WORDS; w++) pl[t].probes[w] = ~0ull;
				for (uint32_t pr = 0; (pr < h->nlookups) && (k < XE_MAX_PROBES); pr++) {
					const struct xe_probe *q = &h->h_probes[pr];
					if (q->root_off && ((fld[q->field] & XE_FFLD_PRESENT) || (q->wide4 & 2))) {
						pl[t].probes[k >> 3] = (pl[t].probes[k >> 3] & ~(0xFFull << (8 * (k & 7))))
							| ((uint64_t)pr << (8 * (k & 7)));
						k++;
					}
				}
			}
		}
		rc = upload_pooled(h, &b->d_fplans, pl, nt * sizeof(*pl));
		if (rc == XE_OK) rc = upload_pooled(h, &b->d_fops, fo, (nops + 4) * sizeof(*fo));
		if (rc == XE_OK) rc = upload_pooled(h, &b->d_fflds, ff, nt * XE_NFIELDS * sizeof(uint32_t));
		if (!tables_pinned) {
			free(pl);
			free(fo);
			free(ff);
		}
		if (rc != XE_OK) { xe_gpu_batch_free(b); return rc; }
		b->kp.fplans = (const struct xe_fplan *)b->d_fplans;
		b->kp.fops = (const struct xe_fop *)b->d_fops;
		b->kp.fflds = (const uint32_t *)b->d_fflds;
	}

	b->kp.progs = (const struct xe_prog *)h->d_progs;
	b->kp.ops = (const uint32_t *)h->d_ops;
	b->kp.terms = (const struct xe_term *)h->d_terms;
	b->kp.opnds = (const struct xe_dopnd *)h->d_opnds;
	b->kp.trie = (const uint32_t *)h->d_trie;
	b->kp.port_named_off = h->port_named_off;
	b->kp.lists = (const struct xe_list_dev *)h->d_lists;
	b->kp.devrules = (const struct xe_devrule *)h->d_devrules;
	b->kp.nmo = h->nmo;
	b->kp.mask_words = h->mask_words;
	b->kp.ndevrules = (uint32_t)h->hs.ndevs;
	b->kp.mark_fprogs = (const struct xe_fprog *)h->d_mark_fprogs;
	b->kp.nmarkprogs = h->nmarkprogs;
	b->kp.tmark = (const uint32_t *)h->d_tmark;
	b->kp.ntmark = h->ntmark;
	b->kp.nruledevs = 0;
	for (size_t d = 0; d < h->hs.ndevs; d++) b->kp.nruledevs += (h->hs.devs[d].nmarks || h->hs.devs[d].skip_unmarked) ? 1u : 0u;
	b->kp.tmpls = (const struct xe_tmpl *)h->d_tmpls;
	b->kp.tents = (const struct xe_tent *)h->d_tents;
	b->kp.pkt = (const uint8_t *)b->d_pkt;
	b->kp.sets = (const struct xe_set *)b->d_sets;
	b->kp.exps = (const struct xe_exporter *)b->d_exps;
	b->kp.sfrecs = (const struct xe_sfrec *)b->d_sfrecs;
	b->kp.nsets = (uint32_t)b->plan.sets.n;
	b->kp.nrec = (uint32_t)b->plan.nrec;
	b->kp.nsf = (uint32_t)b->plan.nsf;
	b->kp.out_base = (uint8_t *)b->d_out;
	b->kp.masks = (uint64_t *)((uint8_t *)b->d_out + b->masks_off);
	b->kp.nparsed = (uint8_t *)b->d_out + b->nparsed_off;
	b->kp.marks = h->hs.ndevs ? (uint32_t *)((uint8_t *)b->d_out + b->marks_off) : NULL;
	b->kp.counters = (unsigned long long *)((uint8_t *)b->d_out + b->counters_off);
	b->kp.sf_has = (uint32_t *)((uint8_t *)b->d_out + b->sfhas_off);
	if (b->w.mode & 1) {
		/* the device walked the sFlow datagrams: no element descriptors, the kernel
		 * finds every record's sample through its datagram's entry */
		b->kp.sfdg = b->w.d_sfdg;
		b->kp.sfbase = b->w.d_sfbase;
		b->kp.sfrd = (const struct xe_sfrd *)b->w.d_rec2dg;
		b->kp.sf_rate = (uint32_t *)((uint8_t *)b->d_out + b->sfrate_off);
	}
	b->kp.rates = h->nfwm ? (uint32_t *)((uint8_t *)b->d_out + b->rates_off) : NULL;
	b->kp.devips = h->fwm_devip ? (uint32_t *)((uint8_t *)b->d_out + b->devips_off) : NULL;
	XE_TICK();
	{
		/* the caller's buffer is free again, and the kernels may run on any stream */
		hipError_t e = hipStreamSynchronize(h->copy_stream);
		if ((e == hipSuccess) && b->w.mode && (b->w.st != h->copy_stream)) e = hipStreamSynchronize(b->w.st);
		if (e != hipSuccess) { xe_gpu_batch_free(b); return hip_fail(h, e, "hipStreamSynchronize(packets)"); }
	}
	XE_TICK();
	if (dbg && (ntm_ == 8)) {
		fprintf(stderr, "prepare: pre-pass %.2f, plan tables %.2f, work lists %.2f, outputs %.2f, decode plans %.2f, "
			"wait for the packets %.2f, all %.2f ms\n", tm_[1] - tm_[0], tm_[3] - tm_[2],
			tm_[4] - tm_[3], tm_[5] - tm_[4], tm_[6] - tm_[5], tm_[7] - tm_[6], tm_[7] - tm_[0]);
	}
#undef XE_TICK
	*out = b;
	return XE_OK;
}

extern "C" int
xe_gpu_batch_run(xe_batch *b, void *stream)
{
	int rc;
	if (!b) return XE_EINVAL;
	/* the calling thread may have another device current */
	HIPCHK(b->h, hipSetDevice(b->h->device));
	/* templates learnt by later batches may have moved the device tables */
	b->kp.tmpls = (const struct xe_tmpl *)b->h->d_tmpls;
	b->kp.tents = (const struct xe_tent *)b->h->d_tents;
	if (b->ev_bulk) HIPCHK(b->h, hipStreamWaitEvent((hipStream_t)stream, b->ev_bulk, 0));
	rc = xe_launch_batch(&b->kp, stream, b->timing ? (void **)b->ev : NULL);
	if ((rc == XE_OK) && b->h->nfwm) {
		uint32_t kwmax = 1, nvmax = 1, nkmax = 0;
		for (size_t i = 0; i < b->h->nfwm; i++) {
			if (b->h->fwms[i].dev.kwords > kwmax) kwmax = b->h->fwms[i].dev.kwords;
			if (b->h->fwms[i].dev.nkey > nkmax) nkmax = b->h->fwms[i].dev.nkey;
			if (b->h->fwms[i].dev.nval > nvmax) nvmax = b->h->fwms[i].dev.nval;
		}
		rc = xe_launch_fwm(&b->kp, (const struct xe_fwm_dev *)b->h->d_fwmdesc, (uint32_t)b->h->nfwm, kwmax, nkmax, nvmax, stream,
			b->timing ? (void **)b->ev_fwm : NULL);
	}
	if (rc != XE_OK) {
		snprintf(b->h->err, sizeof(b->h->err), "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
		return rc;
	}
	b->last_stream = (hipStream_t)stream;
	b->ran = 1;
	if (b->ev_bulk) {
		if (!b->ev_ran) HIPCHK(b->h, hipEventCreateWithFlags(&b->ev_ran, hipEventDisableTiming));
		HIPCHK(b->h, hipEventRecord(b->ev_ran, (hipStream_t)stream));
	}
	return XE_OK;
}

extern "C" int
xe_gpu_batch_count_gathers(xe_batch *b, void *stream, uint64_t out[2])
{
	void *d = NULL;
	int rc;
	if (!b || !out) return XE_EINVAL;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	rc = dev_get(b->h, &d, 256);
	if (rc != XE_OK) return rc;
	if (hipMemsetAsync(d, 0, 16, (hipStream_t)stream) != hipSuccess) { dev_put(b->h, d); return hip_fail(b->h, hipGetLastError(), "hipMemsetAsync"); }
	b->kp.gathers = (unsigned long long *)d;
	rc = xe_gpu_batch_run(b, stream);
	b->kp.gathers = NULL;
	if ((rc == XE_OK) && (hipStreamSynchronize((hipStream_t)stream) != hipSuccess)) rc = hip_fail(b->h, hipGetLastError(), "hipStreamSynchronize");
	if ((rc == XE_OK) && (hipMemcpy(out, d, 16, hipMemcpyDeviceToHost) != hipSuccess)) rc = hip_fail(b->h, hipGetLastError(), "hipMemcpy");
	dev_put(b->h, d);
	return rc;
}

extern "C" int
xe_gpu_set_overlap(xe_gpu *h, int on)
{
	if (!h) return XE_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	if (on && !h->bulk_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->bulk_stream, hipStreamNonBlocking));
	h->overlap = on ? 1 : 0;
	return XE_OK;
}

extern "C" int
xe_gpu_batch_input_done(xe_batch *b)
{
	if (!b) return XE_EINVAL;
	if (!b->ev_bulk) return XE_OK;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	HIPCHK(b->h, hipEventSynchronize(b->ev_bulk));
	return XE_OK;
}

extern "C" int
xe_gpu_batch_done(xe_batch *b)
{
	hipError_t e;
	if (!b) return XE_EINVAL;
	if (!b->ran) return XE_ESTATE;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	e = b->ev_ran ? hipEventQuery(b->ev_ran) : hipStreamQuery(b->last_stream);
	if (e == hipSuccess) return 1;
	if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
	return hip_fail(b->h, e, "hipEventQuery");
}

extern "C" int
xe_gpu_batch_sync(xe_batch *b)
{
	if (!b) return XE_EINVAL;
	if (!b->ran) return XE_ESTATE;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	if (b->ev_ran) HIPCHK(b->h, hipEventSynchronize(b->ev_ran));
	else HIPCHK(b->h, hipStreamSynchronize(b->last_stream));
	return XE_OK;
}

extern "C" int
xe_gpu_batch_timing(xe_batch *b, int enable)
{
	if (!b) return XE_EINVAL;
	HIPCHK(b->h, hipSetDevice(b->h->device));
	if (enable && !b->ev[0]) {
		for (int i = 0; i < 6; i++) HIPCHK(b->h, hipEventCreate(&b->ev[i]));
		for (int i = 0; i < 2; i++) HIPCHK(b->h, hipEventCreate(&b->ev_fwm[i]));
	}
	b->timing = enable ? 1 : 0;
	return XE_OK;
}

extern "C" int
xe_gpu_batch_fwm_ms(xe_batch *b, float *ms)
{
	if (!b || !ms) return XE_EINVAL;
	if (!b->timing || !b->ran || !b->h->nfwm || !(b->plan.nrec + b->plan.nsf)) { *ms = -1.0f; return XE_OK; }
	HIPCHK(b->h, hipStreamSynchronize(b->last_stream));
	HIPCHK(b->h, hipEventElapsedTime(ms, b->ev_fwm[0], b->ev_fwm[1]));
	return XE_OK;
}

extern "C" int
xe_gpu_batch_kernel_ms(xe_batch *b, float ms[3])
{
	int rc;
	const uint32_t launched[3] = { b ? (b->kp.ntiles | b->kp.ntiles2) : 0, b ? b->kp.nslowrec : 0, b ? b->kp.nsf : 0 };

	if (!b || !ms) return XE_EINVAL;
	if (!b->timing) return XE_ESTATE;
	rc = xe_gpu_batch_sync(b);
	if (rc != XE_OK) return rc;
	for (int k = 0; k < 3; k++) {
		ms[k] = -1.0f;
		if (launched[k]) HIPCHK(b->h, hipEventElapsedTime(&ms[k], b->ev[2 * k], b->ev[2 * k + 1]));
	}
	return XE_OK;
}

extern "C" size_t xe_gpu_batch_records(const xe_batch *b) { return b ? b->ntot : 0; }
extern "C" size_t xe_gpu_batch_per_lane_records(const xe_batch *b) { return b ? b->kp.nslowrec : 0; }
extern "C" size_t xe_gpu_batch_mask_words(const xe_batch *b) { return b ? b->h->mask_words : 0; }
extern "C" size_t xe_gpu_batch_payload_bytes(const xe_batch *b) { return b ? b->payload_bytes : 0; }

extern "C" size_t
xe_gpu_batch_algorithmic_bytes(const xe_batch *b)
{
	if (!b) return 0;
	return b->payload_bytes + b->plan.soa_bytes_per_rec_sum + b->ntot * (1 + 8 * (size_t)b->h->mask_words);
}

static int
fetch(xe_batch *b, void *dst, size_t off, size_t bytes)
{
	int rc = xe_gpu_batch_sync(b);
	if (rc != XE_OK) return rc;
	if (!bytes) return XE_OK;
	if (b->h->mirror_owner == b) {
		/* a cursor holds a copy of all the batch's outputs on the host */
		memcpy(dst, b->h->mirror + off, bytes);
		return XE_OK;
	}
	/* on the handle's own stream: the default stream would order this copy
	 * against every other handle's work on the device */
	HIPCHK(b->h, hipMemcpyAsync(dst, (uint8_t *)b->d_out + off, bytes, hipMemcpyDeviceToHost, b->h->copy_stream));
	HIPCHK(b->h, hipStreamSynchronize(b->h->copy_stream));
	return XE_OK;
}

extern "C" int
xe_gpu_batch_masks(xe_batch *b, uint64_t *masks, size_t nwords_total)
{
	if (!b || (nwords_total < b->ntot * b->h->mask_words)) return XE_EINVAL;
	return fetch(b, masks, b->masks_off, b->ntot * b->h->mask_words * 8);
}

extern "C" int
xe_gpu_batch_counters(xe_batch *b, uint64_t *counters, size_t n_u64)
{
	if (!b || (n_u64 < (size_t)b->h->nmo * 3)) return XE_EINVAL;
	return fetch(b, counters, b->counters_off, (size_t)b->h->nmo * 24);
}

extern "C" int
xe_gpu_batch_dropped(xe_batch *b, uint8_t *dropped, size_t n)
{
	int rc;

	if (!b || (n < b->ntot)) return XE_EINVAL;
	memset(dropped, 0, b->ntot);
	rc = xe_gpu_batch_sync(b);
	if (rc != XE_OK) return rc;
	if (b->kp.marks) {
		uint32_t *mk = (uint32_t *)malloc(b->ntot * 4 + 4);
		if (!mk) return XE_ENOMEM;
		rc = fetch(b, mk, b->marks_off, b->ntot * 4);
		if (rc == XE_OK) {
			for (size_t i = 0; i < b->ntot; i++) dropped[i] = (mk[i] & XE_MARK_DROPPED) ? 1 : 0;
		}
		free(mk);
		if (rc != XE_OK) return rc;
	}
	if (b->plan.nsf) {
		/* sFlow samples whose header parse ended before the IP layer never
		 * reach classification (sflow.c:164-168) */
		uint8_t *st = (uint8_t *)malloc(b->plan.nsf + 1);
		if (!st) return XE_ENOMEM;
		rc = fetch(b, st, b->nparsed_off + b->plan.nrec, b->plan.nsf);
		if (rc == XE_OK) {
			for (size_t i = 0; i < b->plan.nsf; i++) if (st[i] & XE_SF_SKIPPED) dropped[b->plan.nrec + i] = 2;
		}
		free(st);
	}
	return rc;
}

/* per-datagram results of the header walk on the device, fetched when a caller
 * asks for something that needs them: what: 1 = first Netflow record of every
 * datagram, 2 = first sFlow record, 4 = the runs */
static int
walk_fetch(xe_batch *b, int what)
{
	struct walk_dev *w = &b->w;
	xe_gpu *h = b->h;
	const size_t n = w->n;

	HIPCHK(h, hipSetDevice(h->device));
	if ((what & 1) && (w->mode & 2) && !w->h_recbase) {
		w->h_recbase = (uint32_t *)malloc(4 * (n + 1));
		if (!w->h_recbase) return XE_ENOMEM;
		HIPCHK(h, hipMemcpy(w->h_recbase, w->d_recbase, 4 * (n + 1), hipMemcpyDeviceToHost));
	}
	if ((what & 2) && (w->mode & 1) && !w->h_sfbase) {
		w->h_sfbase = (uint32_t *)malloc(4 * (n + 1));
		if (!w->h_sfbase) return XE_ENOMEM;
		HIPCHK(h, hipMemcpy(w->h_sfbase, w->d_sfbase, 4 * (n + 1), hipMemcpyDeviceToHost));
	}
	if ((what & 4) && (w->mode & 2) && !w->h_dgruns) {
		/* (run k of datagram i at [k * n + i]) */
		w->h_dgruns = (struct xe_dgrun *)malloc(sizeof(struct xe_dgrun) * (n ? n : 1) * w->nsets);
		if (!w->h_dgruns) return XE_ENOMEM;
		HIPCHK(h, hipMemcpy(w->h_dgruns, w->d_dgruns, sizeof(struct xe_dgrun) * n * w->nsets, hipMemcpyDeviceToHost));
	}
	return XE_OK;
}

extern "C" int
xe_gpu_batch_record_datagram(xe_batch *b, uint32_t *dgram, size_t n)
{
	const struct xe_plan *p;
	const uint32_t *nf_first, *sf_first;
	int rc;

	if (!b || (n < b->ntot) || (b->ntot && !dgram)) return XE_EINVAL;
	p = &b->plan;
	rc = walk_fetch(b, 3);
	if (rc != XE_OK) return rc;
	nf_first = (b->w.mode & 2) ? b->w.h_recbase : p->dg_first_nf;
	sf_first = (b->w.mode & 1) ? b->w.h_sfbase : p->dg_first_sf;
	for (size_t i = 0; i < p->ndg; i++) {
		for (uint32_t r = nf_first[i]; r < nf_first[i + 1]; r++) dgram[r] = (uint32_t)i;
		for (uint32_t r = sf_first[i]; r < sf_first[i + 1]; r++) dgram[p->nrec + r] = (uint32_t)i;
	}
	return XE_OK;
}

extern "C" int
xe_gpu_batch_column(xe_batch *b, int field, void *out, size_t outlen, size_t *width)
{
	size_t w;
	if (!b || (field < 0) || (field >= XE_NTABLE_FIELDS) || (b->kp.col_off[field] == ~0ull)) return XE_EINVAL;
	w = xe_fields[field].col_width;
	if (width) *width = w;
	if (outlen < b->ntot * w) return XE_EINVAL;
	return fetch(b, out, b->kp.col_off[field], b->ntot * w);
}

extern "C" int
xe_gpu_batch_device_ptrs(xe_batch *b, const void **masks, const void **counters, const void **nparsed)
{
	if (!b) return XE_EINVAL;
	if (masks) *masks = b->kp.masks;
	if (counters) *counters = b->kp.counters;
	if (nparsed) *nparsed = b->kp.nparsed;
	return XE_OK;
}

/* ------------------------------------------------------------------ flow_info */
struct slice
{
	uint8_t *nparsed;
	uint32_t *marks;
	uint32_t *sfhas;
	uint8_t *col[XE_NFIELDS];
	uint16_t *vsize[XE_NFIELDS];
};

static void
slice_free(struct slice *s)
{
	free(s->nparsed);
	free(s->marks);
	free(s->sfhas);
	for (int f = 0; f < XE_NFIELDS; f++) { free(s->col[f]); free(s->vsize[f]); }
}

/* fields a record set, for a caller that wants to take them back (NULL: nobody does) */
struct flow_dirty
{
	uint16_t n;
	uint8_t f[XE_NFIELDS + 8];
};

/* n bytes, n mostly one of a few small sizes: copies the compiler can inline */
static inline void
small_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
	switch (n) {
		case 1: dst[0] = src[0]; break;
		case 2: memcpy(dst, src, 2); break;
		case 4: memcpy(dst, src, 4); break;
		case 8: memcpy(dst, src, 8); break;
		case 16: memcpy(dst, src, 16); break;
		default: memcpy(dst, src, n);
	}
}

static inline void
small_clear(uint8_t *dst, size_t n)
{
	switch (n) {
		case 1: dst[0] = 0; break;
		case 2: memset(dst, 0, 2); break;
		case 4: memset(dst, 0, 4); break;
		case 8: memset(dst, 0, 8); break;
		case 16: memset(dst, 0, 16); break;
		default: memset(dst, 0, n);
	}
}

static inline void
put_field(struct flow_dirty *dirty, uint8_t *fi, int f, const uint8_t *bytes, int size)
{
	const struct xe_field *fd = &xe_fields[f];
	int one = 1;
	small_copy(fi + fd->off, bytes, fd->size_max);
	memcpy(fi + fd->off_size, &size, 4);
	memcpy(fi + fd->off_has, &one, 4);
	if (dirty && (dirty->n < sizeof(dirty->f))) dirty->f[dirty->n++] = (uint8_t)f;
}

/* sizes the sFlow decoder reports for its fields (sflow.c:35-77, sflow-impl.h) */
static int
sf_field_size(int f)
{
	switch (f) {
		case XF_in_bytes: case XF_in_pkts: return 8;
		case XF_input_snmp: case XF_output_snmp: case XF_ip4_src_addr: case XF_ip4_dst_addr: return 4;
		case XF_ip6_src_addr: case XF_ip6_dst_addr: return 16;
		case XF_src_mac: case XF_dst_mac: return 6;
		case XF_l4_src_port: case XF_l4_dst_port: case XF_src_vlan: case XF_dst_vlan:
		case XF_src_dot1q_cvlan: case XF_dst_dot1q_cvlan: return 2;
		default: return 1;
	}
}

/* sFlow records [first, first + count) of a batch whose datagrams the device
 * walked: descriptors with the per-datagram fields (agent address, sub-agent
 * id) and the sampling rate filled in, for the re-creation of struct flow_info */
static int
sf_meta_fetch(xe_batch *b, size_t first, size_t count, struct xe_sfrec **out)
{
	xe_gpu *h = b->h;
	struct xe_sfrec *sl = (struct xe_sfrec *)calloc(count + 1, sizeof(*sl));
	struct xe_sfrd *r2d = (struct xe_sfrd *)malloc((count + 1) * sizeof(*r2d));
	uint32_t *rate = (uint32_t *)malloc((count + 1) * 4);
	struct xe_sfagent *ag = NULL;
	int rc = XE_ENOMEM;

	*out = NULL;
	if (!sl || !r2d || !rate) goto out;
	rc = xe_gpu_batch_sync(b);
	if (rc != XE_OK) goto out;
	rc = XE_EHIP;
	if (hipMemcpy(r2d, (const struct xe_sfrd *)b->w.d_rec2dg + first, count * sizeof(*r2d), hipMemcpyDeviceToHost) != hipSuccess) goto hipfail;
	if (hipMemcpy(rate, (const uint8_t *)b->d_out + b->sfrate_off + first * 4, count * 4, hipMemcpyDeviceToHost) != hipSuccess) goto hipfail;
	{
		const uint32_t d0 = r2d[0].dgram, d1 = r2d[count - 1].dgram;
		const size_t nd = (size_t)d1 - d0 + 1;
		/* the datagrams in between: what the count pass noted of their headers */
		rc = XE_ENOMEM;
		ag = (struct xe_sfagent *)malloc(nd * sizeof(*ag));
		if (!ag) goto out;
		rc = XE_EHIP;
		if (hipMemcpy(ag, b->w.d_sfag + d0, nd * sizeof(*ag), hipMemcpyDeviceToHost) != hipSuccess) goto hipfail;
		for (size_t k = 0; k < count; k++) {
			const struct xe_sfagent *g = &ag[r2d[k].dgram - d0];
			const int v6 = g->ver == 6;
			sl[k].agent_ver = v6 ? 6 : 4;
			memcpy(sl[k].agent, g->agent, v6 ? 16 : 4);
			memcpy(sl[k].dev_id, g->dev_id, 4);
			sl[k].sampling_rate = rate[k];
		}
	}
	rc = XE_OK;
	*out = sl;
	sl = NULL;
	goto out;
hipfail:
	rc = hip_fail(h, hipGetLastError(), "hipMemcpy(sFlow datagram headers)");
out:
	free(sl); free(r2d); free(rate); free(ag);
	return rc;
}

/* what the re-creation of records needs of a batch: the output columns of the
 * records [base, ...) (copies of a range, or pointers into a copy of all the
 * outputs), and where the walk over sets and datagrams stands */
struct flow_ctx
{
	struct slice s;
	size_t base;             /* record of element 0 of nparsed / marks / col / vsize */
	size_t sfhas_origin;     /* record of element 0 of sfhas */
	struct xe_sfrec *sf_slice;   /* element descriptors that exist on the device only */
	size_t sf_slice_first;
	size_t si, di;           /* set and datagram of the last record built (records are asked for in rising order) */
	struct flow_dirty *dirty;    /* NULL, or where the fields a record sets are noted */
};

/* struct flow_info of record r into fi (zeroed, or holding nothing but zeros where this record sets nothing) */
static void
flow_build(xe_batch *b, struct flow_ctx *c, size_t r, uint8_t *fi)
{
	xe_gpu *h = b->h;
	const struct xe_set *sets = (const struct xe_set *)b->plan.sets.p;
	const struct xe_tmpl *tm = (const struct xe_tmpl *)h->hs.dev_tmpls.p;
	const struct xe_tent *te = (const struct xe_tent *)h->hs.dev_tents.p;
	const struct xe_exporter *ex = (const struct xe_exporter *)b->plan.exporters.p;
	const struct xe_sfrec *sf = (const struct xe_sfrec *)b->plan.sfrecs.p;
	const size_t k = r - c->base;
	const struct xe_exporter *e;
	uint32_t rate;

	if (r < b->plan.nrec) {
		const struct xe_tmpl *t;
		const struct xe_tent *en;
		uint32_t np = c->s.nparsed[k];
		struct xe_exporter from_dev;
		int on_dev = 0;

		if (b->w.mode & 2) {
			while (r >= b->w.h_recbase[c->di + 1]) c->di++;
			on_dev = b->w.h_walked[c->di];
		}
		if (on_dev) {
			/* a datagram the device listed: template, source id and sampling
			 * rate are in its run, the exporter address came with the datagram */
			const struct xe_dgrun *g = &b->w.h_dgruns[c->di];
			size_t within = r - b->w.h_recbase[c->di];
			for (uint32_t q = 0; (q + 1 < b->w.nsets) && (within >= g->nrec); q++) {
				within -= g->nrec;
				g += b->w.n;
			}
			const uint32_t idb = htonl(g->dev_id);
			t = &tm[g->tmpl & 0xFFFFu];
			memcpy(from_dev.dev_ip, &b->w.h_src[c->di], 4);
			memcpy(from_dev.dev_id, &idb, 4);
			from_dev.sampling_rate = g->rate & 0x7FFFFFFFu;
			from_dev.device = -1;
			e = &from_dev;
		} else {
			while (r >= (size_t)sets[c->si].first_rec + sets[c->si].nrec) c->si++;
			t = &tm[sets[c->si].tmpl];
			e = &ex[sets[c->si].exporter];
		}
		en = &te[t->ent_base];
		if (np > t->nfields) np = t->nfields;   /* never trust a count beyond the template */
		for (uint32_t j = 0; j < np; j++) {
			int f = en[j].field;
			if ((f == XE_F_NONE) || !c->s.col[f]) continue;
			if (c->s.vsize[f]) {
				uint16_t vs = c->s.vsize[f][k];
				if (vs == 0xFFFF) continue;
				put_field(c->dirty, fi, f, c->s.col[f] + k * xe_fields[f].col_width, vs);
			} else {
				if (!(en[j].flags & XE_TE_VALID)) continue;
				put_field(c->dirty, fi, f, c->s.col[f] + k * xe_fields[f].col_width, en[j].len);
			}
		}
		/* virtual_fields_init(), netflow.c:112-124 */
		put_field(c->dirty, fi, XF_dev_ip, e->dev_ip, 4);
		put_field(c->dirty, fi, XF_dev_id, e->dev_id, 4);
		rate = e->sampling_rate;
	} else {
		const struct xe_sfrec *q = c->sf_slice ? &c->sf_slice[r - b->plan.nrec - c->sf_slice_first] : &sf[r - b->plan.nrec];
		uint32_t hb = c->s.sfhas[r - c->sfhas_origin];
		static const int order[] = { XF_in_bytes, XF_in_pkts, XF_protocol, XF_src_tos, XF_dst_tos,
			XF_tcp_flags, XF_l4_src_port, XF_ip4_src_addr, XF_input_snmp, XF_l4_dst_port,
			XF_ip4_dst_addr, XF_output_snmp, XF_ip6_src_addr, XF_ip6_dst_addr, XF_icmp_type,
			XF_min_ttl, XF_max_ttl, XF_src_mac, XF_dst_mac, XF_src_vlan, XF_dst_vlan,
			XF_src_dot1q_cvlan, XF_dst_dot1q_cvlan };
		for (size_t i = 0; i < sizeof(order) / sizeof(order[0]); i++) {
			int f = order[i];
			if (!((hb >> i) & 1)) continue;
			put_field(c->dirty, fi, f, c->s.col[f] + k * xe_fields[f].col_width, sf_field_size(f));
		}
		/* sflow_reset(), sflow-impl.h:274-291 */
		put_field(c->dirty, fi, XF_dev_id, q->dev_id, 4);
		if (q->agent_ver == 4) put_field(c->dirty, fi, XF_dev_ip, q->agent, 4);
		else put_field(c->dirty, fi, XF_dev_ip6, q->agent, 16);
		rate = q->sampling_rate;
	}
	if (c->s.marks && (c->s.marks[k] & XE_MARK_HAS)) {
		uint32_t mk = htonl(c->s.marks[k] & 0x3FFFFFFFu);
		put_field(c->dirty, fi, XF_dev_mark, (const uint8_t *)&mk, 4);
	}
	memcpy(fi + XE_OFF_SAMPLING_RATE, &rate, 4);
}

extern "C" int
xe_gpu_batch_flow_info(xe_batch *b, size_t first, size_t count, void *out)
{
	struct slice s;
	struct flow_ctx c;
	const struct xe_set *sets;
	struct xe_sfrec *sf_slice = NULL;   /* element descriptors that exist on the device only */
	size_t sf_slice_first = 0;
	int rc = XE_OK;
	size_t si = 0, di = 0;

	if (!b || !out || (first + count > b->ntot)) return XE_EINVAL;
	if (!count) return XE_OK;
	memset(&s, 0, sizeof(s));
	memset(out, 0, count * XE_FLOW_INFO_SIZE);

#define GRAB(DST, TYPE, OFF, ELT) do { \
	(DST) = (TYPE *)malloc(count * (ELT) + 8); \
	if (!(DST)) { rc = XE_ENOMEM; goto done; } \
	rc = fetch(b, (DST), (OFF) + first * (ELT), count * (ELT)); \
	if (rc != XE_OK) goto done; } while (0)

	GRAB(s.nparsed, uint8_t, b->nparsed_off, 1);
	if (b->kp.marks) GRAB(s.marks, uint32_t, b->marks_off, 4);
	for (int f = 0; f < XE_NTABLE_FIELDS; f++) {
		if (b->kp.col_off[f] != ~0ull) GRAB(s.col[f], uint8_t, b->kp.col_off[f], xe_fields[f].col_width);
		if (b->kp.vsize_off[f] != ~0ull) GRAB(s.vsize[f], uint16_t, b->kp.vsize_off[f], 2);
	}
	if (b->plan.nsf && (first + count > b->plan.nrec)) {
		size_t sf_first = (first > b->plan.nrec) ? first - b->plan.nrec : 0;
		size_t sf_count = first + count - b->plan.nrec - sf_first;
		s.sfhas = (uint32_t *)calloc(count + 2, 4);
		if (!s.sfhas) { rc = XE_ENOMEM; goto done; }
		rc = fetch(b, s.sfhas + (b->plan.nrec + sf_first - first), b->sfhas_off + sf_first * 4, sf_count * 4);
		if (rc != XE_OK) goto done;
		if (b->w.mode & 1) {
			/* the device walked these datagrams: what sflow_reset() sets per datagram
			 * (sflow-impl.h:274-291) is read back from the datagram headers in the batch
			 * buffer, the sampling rate from the column the kernel wrote */
			rc = sf_meta_fetch(b, sf_first, sf_count, &sf_slice);
			if (rc != XE_OK) goto done;
			sf_slice_first = sf_first;
		}
	}
	if ((b->w.mode & 2) && (first < b->plan.nrec)) {
		rc = walk_fetch(b, 5);
		if (rc != XE_OK) goto done;
		/* the datagram of record `first`: the last one that starts at or before it */
		{
			size_t lo = 0, hi = b->w.n;
			while (lo + 1 < hi) {
				size_t mid = (lo + hi) / 2;
				if (b->w.h_recbase[mid] <= first) lo = mid; else hi = mid;
			}
			di = lo;
		}
	}

	sets = (const struct xe_set *)b->plan.sets.p;
	if ((first < b->plan.nrec) && b->plan.sets.n) {
		size_t lo = 0, hi = b->plan.sets.n - 1;
		while (lo < hi) {
			size_t mid = (lo + hi + 1) / 2;
			if (sets[mid].first_rec <= first) lo = mid; else hi = mid - 1;
		}
		si = lo;
	}
	c.s = s;
	c.base = first;
	c.sfhas_origin = first;
	c.sf_slice = sf_slice;
	c.sf_slice_first = sf_slice_first;
	c.si = si;
	c.di = di;
	c.dirty = NULL;
	for (size_t k = 0; k < count; k++) flow_build(b, &c, first + k, (uint8_t *)out + k * XE_FLOW_INFO_SIZE);
done:
	slice_free(&s);
	free(sf_slice);
	return rc;
#undef GRAB
}

/* The records of a batch one after the other into one caller buffer: all
 * outputs of the batch come to the host in one copy (into page-locked memory
 * the handle keeps), and a record takes back what the record before it set
 * instead of clearing the 2312 bytes again. */
struct xe_flowcur
{
	xe_batch *b;
	uint8_t *flow;
	struct flow_ctx c;
	struct flow_dirty dirty;
	size_t last;
	int started;
};

extern "C" int
xe_gpu_batch_flow_cursor(xe_batch *b, void *flow, xe_flowcur **out)
{
	xe_gpu *h;
	struct xe_flowcur *cur;
	int rc;

	if (!b || !flow || !out) return XE_EINVAL;
	*out = NULL;
	h = b->h;
	if (h->mirror_owner) {
		snprintf(h->err, sizeof(h->err), "xe_gpu_batch_flow_cursor: the handle's other cursor is still open");
		return XE_ESTATE;
	}
	if ((rc = xe_gpu_batch_sync(b)) != XE_OK) return rc;
	HIPCHK(h, hipSetDevice(h->device));
	if (h->mirror_bytes < b->out_bytes) {
		if (h->mirror) (void)hipHostFree(h->mirror);
		h->mirror = NULL;
		h->mirror_bytes = 0;
		const size_t want = b->out_bytes + b->out_bytes / 4 + 4096;
		if (hipHostMalloc((void **)&h->mirror, want, hipHostMallocDefault) != hipSuccess) {
			(void)hipGetLastError();
			snprintf(h->err, sizeof(h->err), "hipHostMalloc(%zu) failed", want);
			return XE_ENOMEM;
		}
		h->mirror_bytes = want;
	}
	if (b->out_bytes) {
		HIPCHK(h, hipMemcpyAsync(h->mirror, b->d_out, b->out_bytes, hipMemcpyDeviceToHost, h->copy_stream));
		HIPCHK(h, hipStreamSynchronize(h->copy_stream));
	}
	cur = (struct xe_flowcur *)calloc(1, sizeof(*cur));
	if (!cur) return XE_ENOMEM;
	cur->b = b;
	cur->flow = (uint8_t *)flow;
	cur->c.s.nparsed = h->mirror + b->nparsed_off;
	if (b->kp.marks) cur->c.s.marks = (uint32_t *)(h->mirror + b->marks_off);
	for (int f = 0; f < XE_NTABLE_FIELDS; f++) {
		if (b->kp.col_off[f] != ~0ull) cur->c.s.col[f] = h->mirror + b->kp.col_off[f];
		if (b->kp.vsize_off[f] != ~0ull) cur->c.s.vsize[f] = (uint16_t *)(h->mirror + b->kp.vsize_off[f]);
	}
	h->mirror_owner = b;      /* from here on the batch's results are read from the copy */
	if (b->plan.nsf) {
		cur->c.s.sfhas = (uint32_t *)(h->mirror + b->sfhas_off);
		cur->c.sfhas_origin = b->plan.nrec;
		if (b->w.mode & 1) {
			rc = sf_meta_fetch(b, 0, b->plan.nsf, &cur->c.sf_slice);
			if (rc != XE_OK) { xe_gpu_flow_cursor_free(cur); return rc; }
		}
	}
	if ((b->w.mode & 2) && b->plan.nrec) {
		rc = walk_fetch(b, 5);
		if (rc != XE_OK) { xe_gpu_flow_cursor_free(cur); return rc; }
	}
	memset(flow, 0, XE_FLOW_INFO_SIZE);
	*out = cur;
	return XE_OK;
}

extern "C" int
xe_gpu_flow_cursor_seek(xe_flowcur *cur, size_t r)
{
	if (!cur || (r >= cur->b->ntot) || (cur->started && (r < cur->last))) return XE_EINVAL;
	if (cur->dirty.n >= sizeof(cur->dirty.f)) {
		memset(cur->flow, 0, XE_FLOW_INFO_SIZE);      /* more fields than the list holds */
	} else {
		for (uint32_t i = 0; i < cur->dirty.n; i++) {
			const struct xe_field *fd = &xe_fields[cur->dirty.f[i]];
			small_clear(cur->flow + fd->off, fd->size_max);
			memset(cur->flow + fd->off_size, 0, 4);
			memset(cur->flow + fd->off_has, 0, 4);
		}
	}
	cur->dirty.n = 0;
	cur->c.dirty = &cur->dirty;
	flow_build(cur->b, &cur->c, r, cur->flow);
	cur->last = r;
	cur->started = 1;
	return XE_OK;
}

extern "C" void
xe_gpu_flow_cursor_free(xe_flowcur *cur)
{
	if (!cur) return;
	if (cur->b->h->mirror_owner == cur->b) cur->b->h->mirror_owner = NULL;
	free(cur->c.sf_slice);
	free(cur);
}

/* ------------------------------------------------------------------ list probe */
extern "C" int
xe_gpu_iplist_match(xe_gpu *h, const char *list, int family, const uint8_t *addrs, size_t n, uint8_t *out)
{
	int li, rc;
	void *d_a = NULL, *d_o = NULL;
	size_t ab = (family == 4) ? 4 : 16;

	if (!h || !list || ((family != 4) && (family != 6)) || (n && (!addrs || !out)) || (n > 0xFFFFFFF0u)) return XE_EINVAL;
	if (!h->committed) return XE_ESTATE;
	li = xe_iplist_find(&h->hs.lists, list);
	if (li < 0) return XE_EINVAL;
	if (!n) return XE_OK;
	HIPCHK(h, hipSetDevice(h->device));
	rc = upload(h, &d_a, addrs, n * ab);
	if (rc == XE_OK) rc = upload(h, &d_o, NULL, n);
	if (rc == XE_OK) rc = xe_launch_trie_probe((const uint32_t *)h->d_trie, h->h_lists[li], family,
		(const uint8_t *)d_a, (uint32_t)n, (uint8_t *)d_o, NULL);
	if (rc == XE_OK) {
		hipError_t e = hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost);
		if (e != hipSuccess) rc = hip_fail(h, e, "hipMemcpy(probe)");
	}
	if (d_a) (void)hipFree(d_a);
	if (d_o) (void)hipFree(d_o);
	return rc;
}

/* the same probe without any allocation (static device staging area) */
extern "C" int
xe_gpu_iplist_match_noalloc(xe_gpu *h, const char *list, int family, const uint8_t *addrs, size_t n, uint8_t *out)
{
	int li, rc;

	if (!h || !list || ((family != 4) && (family != 6)) || (n && (!addrs || !out))) return XE_EINVAL;
	if (n > XE_GPU_NOALLOC_MAX_ADDRS) {
		snprintf(h->err, sizeof(h->err), "xe_gpu_iplist_match_noalloc: %zu addresses, the static staging area holds %d",
			n, XE_GPU_NOALLOC_MAX_ADDRS);
		return XE_EINVAL;
	}
	if (!h->committed) return XE_ESTATE;
	li = xe_iplist_find(&h->hs.lists, list);
	if (li < 0) return XE_EINVAL;
	if (!n) return XE_OK;
	HIPCHK(h, hipSetDevice(h->device));
	rc = xe_launch_trie_probe_static(h->device, (const uint32_t *)h->d_trie, h->h_lists[li], family, addrs,
		(uint32_t)n, out, NULL);
	if (rc == XE_EHIP) snprintf(h->err, sizeof(h->err), "xe_gpu_iplist_match_noalloc: %s", hipGetErrorString(hipGetLastError()));
	return rc;
}
