/*
 * xe_walk.hip -- the header walk on the device (gfx950).
 *
 * One thread per datagram follows the reference's cursor arithmetic over
 * datagram, set and sample headers and leaves what the decode kernels work
 * from: sFlow element descriptors (struct xe_sfrec) and, for Netflow datagrams
 * of the usual shape, runs of complete records grouped by template with their
 * tile list.  Prefix sums give every record the index the walk in arrival
 * order gives it.  Datagrams that touch the template store (or are unusual in
 * any other way) are not decided here: they are flagged and the host walks
 * them.
 *
 * The prefix sums and the two stable sorts are rocPRIM's device-wide
 * primitives; everything that knows about flow export formats is below.
 */
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include "xe_kernels.h"
#include "xe_sfwalk.h"

#define XW_BLOCK 256
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ uint32_t
ld32(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}

static __device__ __forceinline__ uint32_t
be16(const uint8_t *p)
{
	return ((uint32_t)p[0] << 8) | p[1];
}

/* device_find(), devices.c:205-232 / :246-270: the second test is `if`, not `else if` */
static __device__ int
dev_find(const xe_walk_in &in, uint32_t ip, uint32_t id)
{
	for (uint32_t i = 0; i < in.ndevs; i++) {
		const xe_devkey d = in.devs[i];
		if (d.use_ip && d.use_id) {
			if ((ip == d.ip) && (id == d.id)) return (int)i;
		} else if (d.use_ip) {
			if (ip == d.ip) return (int)i;
		}
		if (d.use_id) {
			if (id == d.id) return (int)i;
		}
	}
	return -1;
}

/* ------------------------------------------------------------------ sFlow v5 */
/* sFlow datagrams carry no state from one to the next.  One pass, one thread
 * per datagram, walks sample and element headers (xe_sfwalk.h) and leaves per
 * datagram: the number of records it yields and a 32-byte entry with where it
 * lies, its devices.conf entry and - for datagrams of the usual shape, up to
 * XE_SFDG_POS records, each from the first element of its sample - where each
 * record's sample starts.  After the prefix sum over the counts a second,
 * trivial pass notes the datagram of every record.  The decode kernel reads a
 * record's sample header itself (one window load); only for datagrams of any
 * other shape it walks the datagram again up to its record. */
__global__ void __launch_bounds__(XW_BLOCK)
xw_sf_count_kernel(const xe_walk_in in, uint32_t *cnt, xe_sfdg *dgs, unsigned long long *total)
{
	const uint32_t i = blockIdx.x * XW_BLOCK + threadIdx.x;
	uint32_t n = 0;

	if (i < in.n) {
		xe_sfdg g;
		g.off = 0;
		g.len = 0;
		g.dev = 0;
#pragma unroll
		for (int k = 0; k < XE_SFDG_POS; k++) g.pos[k] = 0;
		if (in.kinds && (in.kinds[i] == XE_KIND_SFLOW)) {
			const uint64_t off = in.offs[i];
			const int64_t len = (int64_t)(in.offs[i + 1] - off);
			xe_sfiter it;
			xe_sfel d;
			uint32_t agent_ver;
			bool usual = true;
			if (xsf_begin(it, in.pkt + off, len, agent_ver)) {
				while (xsf_next<true>(it, d)) {
					if (!d.first) usual = false;
					if (n < XE_SFDG_POS) {
						/* (the index is a loop counter, not a constant: spelled out so
						 * that the array stays in registers) */
#pragma unroll
						for (int k = 0; k < XE_SFDG_POS; k++) if (n == (uint32_t)k) g.pos[k] = (uint16_t)d.hdr_pos;
						if (d.hdr_pos - d.sample_pos == 76) g.dev |= 0x10000u << n;
					}
					n++;
				}
				if (n > XE_SFDG_POS) usual = false;
				g.off = off;
				g.len = (uint32_t)it.len | (usual ? 0u : XE_SFDG_WALK) | ((agent_ver == 6) ? XE_SFDG_AGENT6 : 0u);
				/* device rules use the UDP source and the packet's source_id field, which
				 * sflow_process() never sets: 0 */
				g.dev |= (uint32_t)(dev_find(in, in.src[i], 0) + 1) & 0xFFFFu;
			}
		}
		cnt[i] = n;
		dgs[i] = g;
	}
	/* the batch total in 64 bits: the 32-bit prefix sums could wrap unnoticed */
	unsigned long long s = n;
	for (int d = 32; d; d >>= 1) s += __shfl_down(s, d, 64);
	if (((threadIdx.x & 63) == 0) && s) atomicAdd(total, s);
}

/* the 16-byte entry of every record, at its place in arrival order */
__global__ void __launch_bounds__(XW_BLOCK)
xw_sf_place_kernel(uint32_t n, const uint32_t *base, const xe_sfdg *dgs, xe_sfrd *recs)
{
	const uint32_t i = blockIdx.x * XW_BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint32_t b0 = base[i], b1 = base[i + 1];
	if (b0 == b1) return;
	const u32x4 *gq = (const u32x4 *)(dgs + i);
	const u32x4 ga = gq[0], gb = gq[1];
	const uint64_t off = (uint64_t)ga.x | ((uint64_t)ga.y << 32);
	const uint32_t len = ga.z & XE_SFDG_LEN;
	const uint32_t common = ((ga.w & 0xFFFFu) << 16) | ((ga.z & XE_SFDG_AGENT6) ? XE_SFRD_AGENT6 : 0u);
	if (ga.z & XE_SFDG_WALK) {
		const u32x4 r = { 0u, common | XE_SFRD_WALK, 0u, i };
		for (uint32_t q = b0; q < b1; q++) ((u32x4 *)recs)[q] = r;
		return;
	}
	for (uint32_t k = 0; k < b1 - b0; k++) {
		const uint32_t pw = (k < 2) ? gb.x : (k < 4) ? gb.y : (k < 6) ? gb.z : gb.w;
		const uint32_t hp = (pw >> (16 * (k & 1))) & 0xFFFFu;
		const uint64_t h = off + hp;
		const u32x4 r = { (uint32_t)h, (uint32_t)((h >> 32) & 0xFFu) | common | (((ga.w >> (16 + k)) & 1u) ? XE_SFRD_EXPANDED : 0u),
			len - hp, i };
		((u32x4 *)recs)[b0 + k] = r;
	}
}

/* ------------------------------------------------------------------ Netflow v5 / v9 / IPFIX */
static __device__ __forceinline__ bool
tmpl_lookup(const xe_walk_in &in, uint32_t ver_tid, uint32_t src_ip, uint32_t source_id, xe_thent &e)
{
	if (!in.thash) return false;
	for (uint32_t j = xe_thent_hash(ver_tid, src_ip, source_id) & in.thash_mask;; j = (j + 1) & in.thash_mask) {
		e = in.thash[j];
		if (!e.ver_tid) return false;
		if ((e.ver_tid == ver_tid) && (e.src_ip == src_ip) && (e.source_id == source_id)) return true;
	}
}

/* devices.conf: sampling rate and whether the exporter's records need no rule check */
static __device__ __forceinline__ bool
exporter_plain(const xe_walk_in &in, uint32_t ip, uint32_t id, uint32_t &rate)
{
	const int dev = dev_find(in, ip, id);
	if (dev < 0) { rate = 1; return true; }
	const xe_devkey d = in.devs[dev];
	rate = d.rate | 0x80000000u;
	return d.plain && !(d.rate & 0x80000000u);
}

/* A datagram of the usual shape yields up to XE_WALK_SETS runs, one per data
 * set: runs[k * n + i] is run k of datagram i (nrec 0: none). */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_shape_kernel(const xe_walk_in in, xe_dgrun *runs, uint8_t *regular, uint32_t *kmax)
{
	const uint32_t i = blockIdx.x * XW_BLOCK + threadIdx.x;
	const size_t n = in.n;
	uint32_t k = 0;
	bool ok = false;

	if ((i < in.n) && !(in.kinds && (in.kinds[i] == XE_KIND_SFLOW))) {
		const uint64_t off = in.offs[i];
		int64_t len64 = (int64_t)(in.offs[i + 1] - off);
		const uint32_t len = (len64 > 65536) ? 65536u : (len64 < 0) ? 0u : (uint32_t)len64;
		const uint8_t *p = in.pkt + off;
		const uint32_t src = in.src[i];
		xe_dgrun r = { 0, 0, 0, 0 };
		if (len >= 24) {
			const uint32_t ver = be16(p);
			if (ver == 5) {
				/* parse_netflow_v5(), netflow.c:727-786 */
				const uint32_t nr = be16(p + 2), sid = p[21];
				if (nr && ((24 + 48 * nr) == len) && in.v5_slot_plus1 && in.twalk[in.v5_slot_plus1 - 1].fast
					&& exporter_plain(in, src, sid, r.rate)) {
					r.nrec = nr;
					r.tmpl = (in.v5_slot_plus1 - 1) | (24u << 16);
					r.dev_id = __builtin_bswap32(sid);
					runs[i] = r;
					k = 1;
					ok = true;
				}
			} else if (ver == 9) {
				/* parse_netflow_v9() + parse_netflow_v9_flowset(), netflow.c:319-365, :246-317:
				 * one data set that ends with the datagram, header.count complete records in it */
				const uint32_t count = be16(p + 2), sid = ld32(p + 16), fsid = be16(p + 20), length = be16(p + 22);
				xe_thent e;
				if ((fsid >= 2) && ((20 + length) == len) && count
					&& tmpl_lookup(in, (9u << 16) | p[20] | ((uint32_t)p[21] << 8), src, sid, e)
					&& (e.slot < in.ntmpl)) {
					const xe_twalk tw = in.twalk[e.slot];
					if (tw.fast && (count * tw.stride <= length) && ((24 + count * tw.stride) <= len)
						&& exporter_plain(in, src, sid, r.rate)) {
						r.nrec = count;
						r.tmpl = e.slot | (24u << 16);
						r.dev_id = __builtin_bswap32(sid);
						runs[i] = r;
						k = 1;
						ok = true;
					}
				}
			} else if (ver == 10) {
				/* parse_ipfix() + parse_ipfix_flowset(), netflow.c:628-678, :547-625: data
				 * sets of known templates with complete records, options template sets
				 * (skipped there too) and set ids that mean nothing between them; a
				 * template set, a set length below 4 or a header cut by the end of the
				 * datagram leave the datagram to the host */
				const uint32_t sid = ld32(p + 12);
				uint32_t ptr = 16;
				bool bad = false;
				while (ptr < len) {
					if ((ptr + 4) > len) { bad = true; break; }
					const uint32_t fsid = be16(p + ptr), length = be16(p + ptr + 2);
					if ((length < 4) || (fsid == 2)) { bad = true; break; }
					if (fsid > 255) {
						xe_thent e;
						bad = true;
						if ((k < XE_WALK_SETS)
							&& tmpl_lookup(in, (10u << 16) | p[ptr] | ((uint32_t)p[ptr + 1] << 8), src, sid, e)
							&& (e.slot < in.ntmpl)) {
							const xe_twalk tw = in.twalk[e.slot];
							if (tw.fast && (e.fc == tw.nfields) && (length >= e.fc)) {
								/* record j exists iff length - j * stride >= field count, and gets
								 * every field iff its last field starts before `length` */
								const uint32_t nrec = (length - e.fc) / tw.stride + 1;
								const uint32_t nfull = (length > tw.off_last) ? (length - tw.off_last - 1) / tw.stride + 1 : 0;
								if ((nfull >= nrec) && ((ptr + 4 + nrec * tw.stride) <= len)) {
									r.nrec = nrec;
									r.tmpl = e.slot | ((ptr + 4) << 16);
									r.dev_id = __builtin_bswap32(sid);
									runs[k * n + i] = r;
									k++;
									bad = false;
								}
							}
						}
						if (bad) break;
					}
					ptr += length;
				}
				ok = !bad && k && exporter_plain(in, src, sid, r.rate);
				if (ok) {
					for (uint32_t j = 0; j < k; j++) runs[j * n + i].rate = r.rate;
				}
			}
		}
	}
	if (i < in.n) {
		const xe_dgrun none = { 0, 0, 0, 0 };
		if (!ok) k = 0;
		for (uint32_t j = k; j < XE_WALK_SETS; j++) runs[j * n + i] = none;
		regular[i] = ok ? 1 : 0;
	}
	/* the most runs any datagram has: the later steps look at that many only */
	{
		uint32_t m = k;
		for (int d = 32; d; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = (o > m) ? o : m; }
		if (((threadIdx.x & 63) == 0) && m) atomicMax(kmax, m);
	}
}

/* Per datagram: its record count (from its runs, or what the host walk made of
 * it); per run: the sort entry at [i * nsets + k] (template, index of the run) */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_counts_kernel(const xe_walk_in in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	uint32_t *cnt, uint32_t *key, uint32_t *val, uint32_t *gtotal, uint32_t *gruns, unsigned long long *total)
{
	const uint32_t i = blockIdx.x * XW_BLOCK + threadIdx.x;
	const size_t n = in.n;
	const uint32_t lane = threadIdx.x & 63;
	const bool w = (i < in.n) && (walked[i] != 0);
	uint32_t c = 0;

	for (uint32_t k = 0; k < nsets; k++) {
		uint32_t mine = 0xFFFFFFFFu, nrec = 0;
		if (i < in.n) {
			const xe_dgrun r = runs[k * n + i];
			const bool use = w && r.nrec;
			key[(size_t)i * nsets + k] = use ? (r.tmpl & 0xFFFFu) : in.ntmpl;
			val[(size_t)i * nsets + k] = (uint32_t)(k * n + i);
			if (use) { mine = r.tmpl & 0xFFFFu; nrec = r.nrec; c += nrec; }
		}
		/* per-template sums: a batch has a handful of templates, so the lanes of a
		 * wave that share one add up among themselves and one of them does the
		 * atomic add (155 000 adds to two addresses took over a millisecond) */
		unsigned long long todo = __ballot(mine != 0xFFFFFFFFu);
		while (todo) {
			const int leader = __ffsll((long long)todo) - 1;
			const uint32_t t = __shfl(mine, leader, 64);
			const bool same = mine == t;
			const unsigned long long grp = __ballot(same);
			uint32_t sum = same ? nrec : 0;
			for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d, 64);
			if (lane == (uint32_t)leader) {
				atomicAdd(&gtotal[t], sum);
				atomicAdd(&gruns[t], (uint32_t)__popcll(grp));
			}
			todo &= ~grp;
		}
	}
	if (i < in.n) {
		if (!w) c = hostfirst[i + 1] - hostfirst[i];
		cnt[i] = c;
	}
	/* the batch total in 64 bits: the 32-bit prefix sums could wrap unnoticed */
	unsigned long long s = c;
	for (int d = 32; d; d >>= 1) s += __shfl_down(s, d, 64);
	if ((lane == 0) && s) atomicAdd(total, s);
}

__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_gather_kernel(const xe_dgrun *runs, const uint32_t *ord, uint32_t nruns, uint32_t *snrec)
{
	const uint32_t j = blockIdx.x * XW_BLOCK + threadIdx.x;
	if (j < nruns) snrec[j] = runs[ord[j]].nrec;
}

__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_runs_kernel(const xe_walk_in in, const xe_dgrun *runs, const uint32_t *ord, const uint32_t *keys,
	const uint32_t *sscan, const uint32_t *recbase, const xe_wgroup *groups, uint32_t nruns, xe_frun *out)
{
	const uint32_t j = blockIdx.x * XW_BLOCK + threadIdx.x;
	if (j >= nruns) return;
	const size_t n = in.n;
	const uint32_t v = ord[j], t = keys[j];
	const uint32_t i = (uint32_t)(v % n), k = (uint32_t)(v / n);
	const xe_dgrun r = runs[v];
	const xe_wgroup g = groups[t];
	xe_frun f;
	uint32_t before = 0;
	/* records of the datagram's earlier sets come first */
	for (uint32_t q = 0; q < k; q++) before += runs[q * n + i].nrec;
	f.data_off = (uint32_t)in.offs[i] + (r.tmpl >> 16);
	f.first_rec = recbase[i] + before;
	f.start = g.pos0 + (sscan[j] - sscan[g.run0]);
	f.nrec = r.nrec;
	f.stride = in.twalk[t].stride | (r.rate & 0x80000000u);
	f.rate = r.rate & 0x7FFFFFFFu;
	f.dev_ip = __builtin_bswap32(in.src[i]);
	f.dev_id = r.dev_id;
	out[j] = f;
}

/* tile k: 64 positions of one template group; listed with the batch index of
 * its first record as the sort key */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_tiles_kernel(const xe_frun *fruns, const xe_wgroup *groups, uint32_t ntmpl, uint32_t ntiles,
	xe_ftile *tiles, uint32_t *tkey, uint32_t *tval)
{
	const uint32_t k = blockIdx.x * XW_BLOCK + threadIdx.x;
	if (k >= ntiles) return;
	/* the group: last t with tile0[t] <= k among groups that have tiles */
	uint32_t lo = 0, hi = ntmpl;
	while (lo + 1 < hi) {
		const uint32_t mid = (lo + hi) / 2;
		if (groups[mid].tile0 <= k) lo = mid; else hi = mid;
	}
	/* empty groups share their tile0 with the next one: the search above ends
	 * on the last of them, which is the one that owns the tile */
	const uint32_t t = lo;
	const uint32_t base = groups[t].pos0 + 64 * (k - groups[t].tile0);
	/* the run that holds position `base`: last run of the group with start <= base */
	uint32_t a = groups[t].run0, b = groups[t + 1].run0;
	while (a + 1 < b) {
		const uint32_t mid = (a + b) / 2;
		if (fruns[mid].start <= base) a = mid; else b = mid;
	}
	xe_ftile ft;
	ft.first_run = a;
	ft.tmpl = t;
	ft.base = base;
	ft.pad = fruns[a].first_rec + (base - fruns[a].start);
	tiles[k] = ft;
	tkey[k] = ft.pad;
	tval[k] = k;
}

__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_tiles_place_kernel(const xe_ftile *tiles, const uint32_t *order, uint32_t ntiles, xe_ftile *out)
{
	const uint32_t k = blockIdx.x * XW_BLOCK + threadIdx.x;
	if (k < ntiles) out[k] = tiles[order[k]];
}

/* ------------------------------------------------------------------ launches */
static inline uint32_t
xw_blocks(size_t n)
{
	return (uint32_t)((n + XW_BLOCK - 1) / XW_BLOCK);
}

static inline int
xw_done(void)
{
	return (hipGetLastError() == hipSuccess) ? XE_OK : XE_EHIP;
}

extern "C" size_t
xe_walk_temp_bytes(size_t n)
{
	size_t a = 0, b = 0;
	uint32_t *z = nullptr;
	if (!n) n = 1;
	(void)rocprim::exclusive_scan(nullptr, a, z, z, 0u, n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0);
	(void)rocprim::radix_sort_pairs(nullptr, b, z, z, z, z, n, 0, 32, (hipStream_t)0);
	return ((a > b) ? a : b) + 256;
}

extern "C" int
xe_walk_scan(void *temp, size_t temp_bytes, uint32_t *in, uint32_t *out, size_t n, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	if (hipMemsetAsync(in + n, 0, 4, st) != hipSuccess) return XE_EHIP;
	if (rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n + 1, rocprim::plus<uint32_t>(), st) != hipSuccess) return XE_EHIP;
	return XE_OK;
}

extern "C" int
xe_walk_sort(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
	size_t n, unsigned bits, void *stream)
{
	if (!n) return XE_OK;
	if (bits < 1) bits = 1;
	if (rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, bits, (hipStream_t)stream) != hipSuccess) return XE_EHIP;
	return XE_OK;
}

extern "C" int
xe_walk_sf_count(const struct xe_walk_in *in, uint32_t *cnt, struct xe_sfdg *dgs, unsigned long long *total, void *stream)
{
	if (!in->n) return XE_OK;
	hipLaunchKernelGGL(xw_sf_count_kernel, dim3(xw_blocks(in->n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, cnt, dgs, total);
	return xw_done();
}

extern "C" int
xe_walk_sf_place(uint32_t n, const uint32_t *base, const struct xe_sfdg *dgs, struct xe_sfrd *recs, void *stream)
{
	if (!n) return XE_OK;
	hipLaunchKernelGGL(xw_sf_place_kernel, dim3(xw_blocks(n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, n, base, dgs, recs);
	return xw_done();
}

extern "C" int
xe_walk_nf_shape(const struct xe_walk_in *in, struct xe_dgrun *runs, uint8_t *regular, uint32_t *kmax, void *stream)
{
	if (!in->n) return XE_OK;
	hipLaunchKernelGGL(xw_nf_shape_kernel, dim3(xw_blocks(in->n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, runs, regular, kmax);
	return xw_done();
}

extern "C" int
xe_walk_nf_counts(const struct xe_walk_in *in, const struct xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	uint32_t *cnt, uint32_t *key, uint32_t *val, uint32_t *gtotal, uint32_t *gruns, unsigned long long *total, void *stream)
{
	if (!in->n) return XE_OK;
	hipLaunchKernelGGL(xw_nf_counts_kernel, dim3(xw_blocks(in->n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, runs, nsets, walked,
		hostfirst, cnt, key, val, gtotal, gruns, total);
	return xw_done();
}

extern "C" int
xe_walk_nf_gather(const struct xe_dgrun *runs, const uint32_t *ord, uint32_t nruns, uint32_t *snrec, void *stream)
{
	if (!nruns) return XE_OK;
	hipLaunchKernelGGL(xw_nf_gather_kernel, dim3(xw_blocks(nruns)), dim3(XW_BLOCK), 0, (hipStream_t)stream, runs, ord, nruns, snrec);
	return xw_done();
}

extern "C" int
xe_walk_nf_runs(const struct xe_walk_in *in, const struct xe_dgrun *runs, const uint32_t *ord, const uint32_t *keys,
	const uint32_t *sscan, const uint32_t *recbase, const struct xe_wgroup *groups, uint32_t nruns, struct xe_frun *out, void *stream)
{
	if (!nruns) return XE_OK;
	hipLaunchKernelGGL(xw_nf_runs_kernel, dim3(xw_blocks(nruns)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, runs, ord, keys,
		sscan, recbase, groups, nruns, out);
	return xw_done();
}

extern "C" int
xe_walk_nf_tiles(const struct xe_frun *fruns, const struct xe_wgroup *groups, uint32_t ntmpl, uint32_t ntiles,
	struct xe_ftile *tiles, uint32_t *tkey, uint32_t *tval, void *stream)
{
	if (!ntiles) return XE_OK;
	hipLaunchKernelGGL(xw_nf_tiles_kernel, dim3(xw_blocks(ntiles)), dim3(XW_BLOCK), 0, (hipStream_t)stream, fruns, groups, ntmpl,
		ntiles, tiles, tkey, tval);
	return xw_done();
}

extern "C" int
xe_walk_nf_tiles_place(const struct xe_ftile *tiles, const uint32_t *order, uint32_t ntiles, struct xe_ftile *out, void *stream)
{
	if (!ntiles) return XE_OK;
	hipLaunchKernelGGL(xw_nf_tiles_place_kernel, dim3(xw_blocks(ntiles)), dim3(XW_BLOCK), 0, (hipStream_t)stream, tiles, order,
		ntiles, out);
	return xw_done();
}
