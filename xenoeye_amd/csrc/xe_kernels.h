/*
 * xe_kernels.h -- launch interface between the C ABI glue and the HIP kernels.
 */
#ifndef xe_kernels_h_included
#define xe_kernels_h_included

#include "xe_internal.h"

/* per devices.conf entry: its mark programs live after the object programs */
struct xe_devrule
{
	uint32_t first_prog, nprogs;
	uint32_t skip_unmarked;
	uint32_t pad;
};

/* device operand, numeric form: addresses and MACs as big-endian integers so
 * that "bit i from the most significant" is plain shifting */
struct xe_dopnd
{
	uint32_t is_list;
	uint32_t list;
	int32_t low, high;           /* list operands: lookup ids of the src / dst side, -1 none */
	uint64_t net_hi, net_lo;     /* addr4 / mac use net_lo only */
	uint64_t mask_hi, mask_lo;
};

#define XE_MARK_DROPPED 0x80000000u
#define XE_MARK_HAS 0x40000000u

/* sFlow per-record status written by the kernel */
#define XE_SF_SKIPPED 0x80   /* parser state below NO_IP: record not delivered (sflow.c:164-168) */

struct xe_kparams
{
	/* configuration */
	const struct xe_prog *progs;
	const uint32_t *ops;
	const struct xe_term *terms;
	const struct xe_dopnd *opnds;
	const uint32_t *trie;
	const struct xe_list_dev *lists;
	const struct xe_devrule *devrules;
	uint32_t nmo, mask_words, ndevrules, pad0;
	/* templates */
	const struct xe_tmpl *tmpls;
	const struct xe_tent *tents;
	/* batch input */
	const uint8_t *pkt;
	const struct xe_set *sets;
	const struct xe_exporter *exps;
	const struct xe_sfrec *sfrecs;
	uint32_t nsets, nrec, nsf;
	uint32_t nrgroups_used;      /* range groups before the padding to a multiple of 4 */
	/* batch output; record index space = [0,nrec) netflow then [nrec,nrec+nsf) sflow */
	uint64_t *masks;
	uint8_t *nparsed;            /* netflow: fields parsed; sflow: parser state | XE_SF_SKIPPED */
	uint32_t *marks;             /* NULL when no devices.conf entry exists */
	unsigned long long *counters;
	uint8_t *out_base;           /* columns: out_base + col_off[f] + rec * width */
	uint64_t col_off[XE_NFIELDS];   /* ~0 = column absent */
	uint64_t vsize_off[XE_NFIELDS]; /* u16 per record, ~0 = absent */
	uint32_t *sf_has;            /* sflow: presence bits of the fields its parser can set */

	/* fast path: runs of complete records (fixed-length template without
	 * repeated or string elements, every read inside the datagram, exporter
	 * without a devices.conf entry), grouped by template so that a wave
	 * decodes one template; 64 records per tile */
	const struct xe_frun *fruns;
	const struct xe_ftile *ftiles;
	const struct xe_fplan *fplans;  /* per template slot */
	const struct xe_fop *fops;
	const struct xe_fprog *fprogs;  /* objects not resolved through motab, in index order */
	const uint64_t *motab;       /* [mask word][XE_MOTAB_ROWS] */
	uint32_t nfprogs, motab_lds; /* motab_lds: the table is small enough to be copied to LDS */
	const uint32_t *cjgb;        /* byte index of every fail-table group (count padded to a multiple of 4) */
	const uint64_t *cjtab;       /* [group][byte value][mask word] */
	const struct xe_cjderived *cjderived;
	const struct xe_rgroup *rgroups;     /* range-term groups (count padded to a multiple of 4) */
	const uint32_t *rgdata;      /* their breakpoints and tables */
	uint32_t rg_lds, pad_rg;     /* not 0: rgdata also lies behind the atom table (motab), from this 64-bit word on, and goes to LDS with it */
	uint32_t nrgroups, ncjgroups;
	const uint64_t *cjcand;      /* per mask word: the objects matched through cjrows */
	uint32_t pad_cj, ncjderived;
	uint64_t all_probes[XE_LK_WORDS]; /* byte i = i-th probe with non-empty tables, 0xFF = end */
	uint32_t lw;                 /* 64-bit words of lookup bits per record: 1, or XE_LK_WORDS */
	uint32_t motab_words;        /* u64 words of the atom table with its non-empty-row masks */
	uint32_t *rates;             /* per record: sampling rate (only when fixed-window tables exist) */
	uint32_t *devips;            /* per record: dev_ip bytes (only when a table is keyed by it) */
	uint32_t per_object, pad_po;   /* per_object: run every object's program on its own (limits exceeded) */
	const uint32_t *fflds;       /* [template slot][XE_NFIELDS] XE_FFLD_* words */
	uint32_t nfruns, ntiles;
	uint32_t fast_r;             /* records per lane and tile: 1, 2 or 4 */
	uint32_t fast_dwords;        /* LDS dwords staged per record (largest fast template) */
	uint32_t fast_want_dev, pad3;/* some program reads dev_ip / dev_id */
	/* everything else goes through the general kernel */
	const struct xe_set *ssets;
	const uint32_t *swstart;
	uint32_t nssets, nslowrec;
	uint32_t nterms;             /* distinct terms */
	const struct xe_fterm *fterms;  /* parallel to terms[0, nterms) */
	const uint32_t *tneed;       /* the terms some program needs evaluated */
	uint32_t ntneed, pad4;
	const struct xe_probe *probes;
	uint32_t nprobes, pad2;
	uint32_t lds_counters;       /* copies of the counters a block keeps in LDS (a power of two; 0: none, adds go to memory) */
	uint32_t port_named_off;     /* word offset in `trie` of the 65536-bit set of ports with a service name, 0 = none */
	/* a second run / tile table pair for the fast kernel: what the header walk
	 * on the device listed (the first pair is the host walk's) */
	const struct xe_frun *fruns2;
	const struct xe_ftile *ftiles2;
	uint32_t nfruns2, ntiles2;
	/* sFlow records whose datagrams the device walked (sfrecs is NULL then): per
	 * datagram its entry and the index of its first record, per record its datagram */
	const struct xe_sfdg *sfdg;
	const uint32_t *sfbase;
	const struct xe_sfrd *sfrd;
	uint32_t *sf_rate;           /* per sFlow record: the sampling rate of its sample */
	/* the mark filters of devices.conf for the fast kernel: entry i is program nmo + i */
	const struct xe_fprog *mark_fprogs;
	const uint32_t *tmark;       /* terms that read the device mark */
	uint32_t nmarkprogs, ntmark;
	uint32_t nruledevs, pad5;    /* devices.conf entries with mark filters or skip-unmarked */
	/* instrumentation (NULL = off): [0] += root-table loads, [1] += node and chain-record loads of the list walks */
	unsigned long long *gathers;
};

#define XE_MAX_FAST_TERMS 256
#define XE_FAST_MAX_STRIDE 256       /* longest record the fast kernel stages in LDS */
#define XE_FAST_LDS_BUDGET (64 * 1024) /* LDS bytes per 256-thread block (2 blocks per CU) */
#define XE_MAX_PROBES (8 * XE_LK_WORDS)
#define XE_MAX_LDS_COUNTER_MOS 1024

#ifdef __cplusplus
extern "C" {
#endif
/* zeroes `bytes` (a multiple of 16) at p (16-byte aligned) with streaming stores */
int xe_launch_zero_fill(void *p, size_t bytes, void *stream);
/* enqueue decode+classify of the whole batch on `stream` (hipStream_t) */
/* events: NULL, or 6 hipEvent_t = start/stop pairs recorded on `stream` around
 * the fast, general and sFlow kernel launches (only around those that run) */
int xe_launch_batch(const struct xe_kparams *kp, void *stream, void **events);

/* Fixed-window table of one object on the device (the per-record part of
 * monit_object_process_nf(), monit-objects.c:927-1023): an open-addressing
 * hash table of `cap_mask + 1` slots of `swords` 64-bit words.  A key is the
 * key fields' bytes as struct flow_info holds them, one after the other, zero
 * padded to `kwords` whole words.  A key of one word: slots[i] = a state word
 * (2 = taken; for the host) and the key ^ XE_FWM_KEY_XOR.  Longer keys:
 * slots[i] = `kw7` words of seven key bytes each under XE_FWM_WORD_MARK, 0 =
 * not claimed yet.  Either way a word is written once in a window and then
 * only read.  vals[i][nval] = the sums (wrapping: value * scale * sampling
 * rate), kept apart so that the atomic adds to a popular row do not fight with
 * the readers of its key. */
#define XE_FWM_MAX_KEY_FIELDS 8
#define XE_FWM_MAX_VALS 4
#define XE_FWM_MAX_KWORDS 8
/* tables whose key is one word keep it as key ^ XE_FWM_KEY_XOR, 0 = free slot */
#define XE_FWM_KEY_XOR 0x9E3779B97F4A7C15ull
/* longer keys: seven key bytes to a word, this byte on top (so no word of a key is 0) */
#define XE_FWM_WORD_MARK 0x8000000000000000ull
struct xe_fwm_dev
{
	uint32_t mo, nkey, nval, kwords;
	uint32_t cap_mask, swords;
	uint32_t kw7, pad;                /* keys of several words: words of the seven-bytes-a-word form */
	unsigned long long *slots;
	unsigned long long *vals;
	unsigned long long *dropped;      /* records whose key found no free slot */
	uint8_t kfield[XE_FWM_MAX_KEY_FIELDS], ksize[XE_FWM_MAX_KEY_FIELDS];
	uint8_t vfield[XE_FWM_MAX_VALS], vscale[XE_FWM_MAX_VALS];
	/* function forms as key fields (monit_object_key_add_fld(), monit-objects.c:878-921):
	 * kfunc = XE_FN_* (0: the field's bytes), kfield / kfield2 the arguments, kk the
	 * third argument of div_r / div_l.  The table keeps div and min as the 8 bytes
	 * the reference keeps (big-endian), tfstr as the flags byte, portstr as the
	 * port (2 bytes) and ppstr as both ports (4 bytes): ksize is that size, and
	 * the texts are made of them when the table is read.  div_l: dl + dl_off[k]
	 * holds dl_n[k] pieces {lo, hi, value} of the int quotient line. */
	uint8_t kfunc[XE_FWM_MAX_KEY_FIELDS], kfield2[XE_FWM_MAX_KEY_FIELDS];
	int32_t kk[XE_FWM_MAX_KEY_FIELDS];
	uint16_t dl_off[XE_FWM_MAX_KEY_FIELDS], dl_n[XE_FWM_MAX_KEY_FIELDS];
	const int32_t *dl;
};
/* kwmax / nkmax / nvmax: the longest key (words), the most key fields and the most sums among the tables */
int xe_launch_fwm(const struct xe_kparams *kp, const struct xe_fwm_dev *tables, uint32_t ntables, uint32_t kwmax, uint32_t nkmax, uint32_t nvmax,
	void *stream, void **events);
/* n addresses (4 or 16 bytes each) against one flattened list */
int xe_launch_trie_probe(const uint32_t *trie, struct xe_list_dev list, int family,
	const uint8_t *addrs, uint32_t n, uint8_t *out, void *stream);
/* the same without any allocation: staged in a static device array (at most
 * XE_GPU_NOALLOC_MAX_ADDRS addresses); addrs / out are host memory */
int xe_launch_trie_probe_static(int device, const uint32_t *trie, struct xe_list_dev list, int family,
	const uint8_t *addrs, uint32_t n, uint8_t *out, void *stream);
#ifdef __cplusplus
}
#endif

/* ------------------------------------------------------------------ header walk on the device
 * The walk over datagram, set and sample headers (netflow.c:319-365, :628-678,
 * :727-786; sflow-impl.h:68-405) for datagrams that carry no template: one
 * thread per datagram.  sFlow datagrams keep no state from one to the next, so
 * all of them are walked here.  Of the Netflow datagrams those of the usual
 * shape are (v5; v9 with one data set that ends with the datagram; IPFIX with
 * up to XE_WALK_SETS data sets; templates the handle already knows - a
 * template set that repeats one byte for byte included -, fixed-length
 * records, all of them complete); anything else - template sets that define or
 * change a template, several v9 sets, cut-short records, unknown templates -
 * is left to the host walk, which keeps the template store. */

/* devices.conf entry as device_find() tests it (devices.c:205-232) */
struct xe_devkey
{
	uint32_t use_ip, use_id, ip, id;   /* ip: bytes as stored; id: htonl(value) */
	uint32_t rate;
	uint32_t plain;                    /* the device walk may list this exporter's runs */
	uint32_t rules;                    /* mark filters to run, or unmarked records to drop: 1 + the entry runs name for them (the first with the same filters) */
	uint32_t pad;
};

/* one slot of the devices.conf look-up table: open addressing, kind 0 = free */
struct xe_devhent
{
	uint32_t kind, ip, id;
	uint32_t dev;                      /* first entry of that kind with that key */
};
#define XE_DEVHASH_MIN 9               /* devices.conf entries from which the table is built */
#ifdef __cplusplus
static inline __host__ __device__
#else
static inline
#endif
uint32_t
xe_devh_hash(uint32_t kind, uint32_t ip, uint32_t id)
{
	uint32_t h = kind * 0x9E3779B1u ^ ip * 0x85EBCA6Bu ^ id * 0xC2B2AE35u;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
	return h;
}

/* template store entry for the device: open addressing over `cap` entries
 * (a power of two), ver == 0 marks a free entry */
struct xe_thent
{
	uint32_t ver_tid;     /* version << 16 | template id in the byte order of the wire */
	uint32_t src_ip, source_id;
	uint32_t slot;        /* device template */
	uint32_t fc;          /* field count the template set announced */
	uint32_t raw_off, raw_len;   /* the template as the reference stores it: bytes in xe_walk_in.traw */
	uint32_t pad;
};

#ifdef __HIPCC__
#define XE_HD __host__ __device__
#else
#define XE_HD
#endif
static inline XE_HD uint32_t
xe_thent_hash(uint32_t ver_tid, uint32_t src_ip, uint32_t source_id)
{
	/* every bit of the three words reaches the low bits the table is indexed with
	 * (exporter addresses differ in their last byte, which is the top byte of the
	 * word) */
	uint32_t h = (ver_tid * 0x9E3779B1u) ^ (src_ip * 0x85EBCA77u) ^ (source_id * 0xC2B2AE3Du);
	h ^= (src_ip >> 16) ^ (src_ip >> 24);
	h ^= h >> 16;
	h *= 0x7FEB352Du;
	h ^= h >> 15;
	h *= 0x846CA68Bu;
	h ^= h >> 16;
	return h;
}

/* what the Netflow walk needs of a device template */
struct xe_twalk
{
	uint16_t stride, nfields, off_last;
	uint8_t fast;         /* the fast kernel decodes it */
	uint8_t mode;
};

/* a data set of a datagram the device walked: one run of complete records
 * (v5, v9: one per datagram; IPFIX: up to XE_WALK_SETS) */
#define XE_WALK_SETS 4
struct xe_dgrun
{
	uint32_t nrec;        /* 0: no such run */
	uint32_t tmpl;        /* device template (16 bits) | bytes from the datagram start to the first record << 16 */
	uint32_t dev_id;      /* source id / observation domain as a big-endian number */
	uint32_t rate;        /* sampling rate | has a devices.conf entry << 31 */
};

/* Device words the walk adds up into, cleared once per batch: 32 bytes ([0] u64
 * sFlow records of the batch), then XE_WCOPIES copies (a block adds to copy
 * blockIdx % XE_WCOPIES: adds to one address from all over the chip queue up)
 * of { u64 records of all datagrams; u32 most runs of a datagram; u32 Netflow
 * datagrams left to the host; u32 records[nt + 2]; u32 runs[nt + 2] } per
 * device template */
#define XE_WCOPIES 8
#define XE_WCOPY_BYTES(NT) (16 + 8 * ((size_t)(NT) + 2))
#define XE_WZERO_BYTES(NT) (32 + XE_WCOPIES * XE_WCOPY_BYTES(NT))

/* what the sFlow walk leaves per datagram */
#define XE_SFDG_POS 8
#define XE_SFDG_WALK 0x80000000u     /* len: not of the usual shape, the decode kernel walks the datagram itself */
#define XE_SFDG_AGENT6 0x40000000u   /* len: the agent address is an IPv6 one */
#define XE_SFDG_LEN 0x0001FFFFu
struct xe_sfdg
{
	uint64_t off;                 /* the datagram in the batch buffer */
	uint32_t len;                 /* min(its length, 65536) | XE_SFDG_*; 0 = yields nothing */
	uint32_t dev;                 /* bits 0-15: devices.conf entry + 1, 0 = none; bit 16 + k: record k comes from an
	                                 expanded flow sample (usual shape) */
	uint16_t pos[XE_SFDG_POS];    /* usual shape: byte offset of the sampled header of record k */
};
/* ... and per record (16 bytes, written once the prefix sums over the datagrams' counts are known) */
#define XE_SFRD_EXPANDED 0x100u
#define XE_SFRD_WALK 0x200u
#define XE_SFRD_AGENT6 0x400u
struct xe_sfrd
{
	uint32_t hdr_lo;              /* usual shape: the sampled header in the batch buffer, bits 0-31 */
	uint32_t hi;                  /* bits 0-7: bits 32-39 of that offset; XE_SFRD_*; bits 16-31: devices.conf entry + 1 */
	int32_t left;                 /* usual shape: bytes from the sampled header to the end of the datagram */
	uint32_t dgram;               /* the record's datagram */
};

struct xe_walk_in
{
	const uint8_t *pkt;
	const uint64_t *offs;          /* n + 1 */
	const uint32_t *src;
	const uint8_t *kinds;          /* NULL: all Netflow */
	uint32_t n;
	uint32_t ndevs;
	const struct xe_devkey *devs;
	const struct xe_thent *thash;  /* NULL: no template known */
	uint32_t thash_mask;
	uint32_t v5_slot_plus1;        /* device template of the v5 record, 0 = not built yet */
	const struct xe_twalk *twalk;  /* per device template */
	uint32_t ntmpl;
	uint32_t devhash_mask;         /* entries of devhash - 1 (a power of two) */
	const uint8_t *traw;           /* the stored templates' bytes (xe_thent.raw_off) */
	/* device_find() without the scan over all entries: the first entry that
	 * matches an exporter is the smallest of three answers - entries that test
	 * the address alone (kind 1, key = address), address and id (kind 2), the id
	 * in their second test (kind 3: every entry with use_id, devices.c:224-230) -
	 * each the first entry of its kind with that key.  NULL: few entries, scanned */
	const struct xe_devhent *devhash;
};

#ifdef __cplusplus
extern "C" {
#endif
/* bytes of scratch the scans and sorts below need for up to n items */
size_t xe_walk_temp_bytes(size_t n);
/* out[0..n] = exclusive prefix sums of in[0..n) (out[n] = the total); in must have n + 1 entries, in[n] is overwritten with 0 */
int xe_walk_scan(void *temp, size_t temp_bytes, uint32_t *in, uint32_t *out, size_t n, void *stream);

/* sFlow: cnt[i] = records datagram i yields (0 for other kinds), dgs[i] = what the decode kernel
 * needs of the datagram; *total += the counts */
/* what sflow_reset() sets per datagram (sflow-impl.h:274-291), bytes as on the wire: for the
 * host, which re-creates struct flow_info without reading the datagrams again */
struct xe_sfagent
{
	uint8_t agent[16];            /* 4 or 16 bytes of it */
	uint8_t dev_id[4];            /* the sub-agent id */
	uint32_t ver;                 /* 4 or 6; 0 = the datagram yields nothing */
};
int xe_walk_sf_count(const struct xe_walk_in *in, uint32_t *cnt, struct xe_sfdg *dgs, struct xe_sfagent *agents,
	unsigned long long *total, void *stream);
/* sFlow: recs[r] for the records r in [base[i], base[i + 1]) of datagram i < n */
int xe_walk_sf_place(uint32_t n, const uint32_t *base, const struct xe_sfdg *dgs, struct xe_sfrd *recs, void *stream);

/* Netflow: per datagram, the runs it yields when its shape is the usual one
 * (runs[k * n + i], k < XE_WALK_SETS, nrec > 0 for those it has; regular[i] = 1),
 * else regular[i] = 0.  sums = the XE_WZERO_BYTES(in->ntmpl) words above: most runs of a datagram (atomic max), Netflow
 * datagrams that are not regular, and - valid when there are none of those - per template the records and runs listed */
int xe_walk_nf_shape(const struct xe_walk_in *in, struct xe_dgrun *runs, uint8_t *regular, void *sums, void *stream);
/* the same sums (the copies of `sums` must have been cleared) after the host walk has had its say: over the datagrams
 * with walked[i] != 0, plus hostfirst[i + 1] - hostfirst[i] records for each of the others (hostfirst NULL: none) */
int xe_walk_nf_totals(const struct xe_walk_in *in, const struct xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	void *sums, void *stream);
/* per template group t: first position, first run (ntmpl + 1 entries each) */
struct xe_wgroup { uint32_t pos0, run0, tile0, pad; };
/* Run slots are numbered in arrival order, slot j = datagram * nsets + set.  From the runs to the fast kernel's
 * tables: tmaps[p * (ntmpl + 1) + template] = the template's place among the XE_WTUPLE templates of pass p (0xFF = another
 * pass), nactive = templates with runs in this batch; out[groups[t].run0 + ...] the runs of template t in arrival order,
 * tiles[] the tile list in the order of the batch index of each tile's first record, recbase[0..n] the batch index of
 * every datagram's first record (hostfirst: what the host walk made of the datagrams it walked, NULL = it walked none;
 * nruns / ntiles: entries the two tables hold);
 * work = xe_walk_nf_work_bytes() bytes of device scratch */
#define XE_WTUPLE 8
size_t xe_walk_nf_work_bytes(size_t n, uint32_t nsets);
int xe_walk_nf_tables(const struct xe_walk_in *in, const struct xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	const uint8_t *tmaps, uint32_t npass, uint32_t nactive, const struct xe_wgroup *groups, void *work,
	struct xe_frun *out, uint32_t nruns, struct xe_ftile *tiles, uint32_t ntiles, uint32_t *recbase, void *stream);
#ifdef __cplusplus
}
#endif
#endif
