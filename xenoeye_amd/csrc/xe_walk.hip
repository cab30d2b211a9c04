/*
 * xe_walk.hip -- the header walk on the device (gfx950).
 *
 * One thread per datagram follows the reference's cursor arithmetic over
 * datagram, set and sample headers and leaves what the decode kernels work
 * from: per sFlow datagram its record count and where its first samples start
 * (struct xe_sfdg), per sFlow record a 16-byte entry (struct xe_sfrd); for
 * Netflow datagrams of the usual shape, runs of complete records grouped by
 * template with their tile list.  Prefix sums give every record the index the
 * walk in arrival order gives it.  Datagrams that change the template store
 * (or are unusual in any other way) are not decided here: they are flagged and
 * the host walks them.
 *
 * The Netflow tables need no sort: per template, prefix sums over the
 * datagrams' runs and tiles (in chunks of XW_CHUNK slots) give every run and
 * tile its place.  Only the sFlow record counts go through rocPRIM's scan.
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
	if (in.devhash) {
		/* the first matching entry = the smallest of the first entries that match
		 * by address alone, by address and id, and by id in the second test */
		uint32_t best = 0xFFFFFFFFu;
#pragma unroll
		for (uint32_t kind = 1; kind <= 3; kind++) {
			const uint32_t kip = (kind == 3) ? 0u : ip, kid = (kind == 1) ? 0u : id;
			for (uint32_t j = xe_devh_hash(kind, kip, kid) & in.devhash_mask;; j = (j + 1) & in.devhash_mask) {
				const xe_devhent e = in.devhash[j];
				if (!e.kind) break;
				if ((e.kind == kind) && (e.ip == kip) && (e.id == kid)) { best = (e.dev < best) ? e.dev : best; break; }
			}
		}
		return (best == 0xFFFFFFFFu) ? -1 : (int)best;
	}
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
xw_sf_count_kernel(const xe_walk_in in, uint32_t *cnt, xe_sfdg *dgs, xe_sfagent *agents, unsigned long long *total)
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
		{
			/* agent address and sub-agent id (the five words behind version and address type) */
			u32x4 a = { 0u, 0u, 0u, 0u };
			uint32_t id = 0, ver = 0;
			if (g.len) {
				const uint8_t *p = in.pkt + g.off + XE_SF_AGENT_POS;
				ver = (g.len & XE_SFDG_AGENT6) ? 6u : 4u;
				a.x = ld32(p);
				if (ver == 6) { a.y = ld32(p + 4); a.z = ld32(p + 8); a.w = ld32(p + 12); }
				id = ld32(p + ((ver == 6) ? 16 : 4));
			}
			uint32_t *qw = (uint32_t *)(agents + i);
			qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w; qw[4] = id; qw[5] = ver;
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

/* is the template at `t` (tsize bytes: id, field count, (type, length) pairs -
 * what the reference stores of a v9 template, netflow.c:160-172) the one held? */
static __device__ bool
tmpl_same(const xe_walk_in &in, const xe_thent &e, const uint8_t *t, uint32_t tsize)
{
	if (!in.traw || (e.raw_len != tsize)) return false;
	const uint8_t *h = in.traw + e.raw_off;
	for (uint32_t j = 0; j < tsize; j++) if (h[j] != t[j]) return false;
	return true;
}

/* the same for the first template of an IPFIX template set at p + at (`avail`
 * bytes of set behind the set header): the reference stores the template header
 * and 8 bytes per element - id with the enterprise bit cleared, length,
 * enterprise number or zeros (ipfix_template_convert(), netflow.c:389-433) - and
 * gives up on the datagram when an element passes the end of the set */
static __device__ bool
tmpl_same_ipfix(const xe_walk_in &in, const xe_thent &e, const uint8_t *p, uint32_t at, uint32_t avail, uint32_t len)
{
	if (!in.traw || (e.raw_len < 4)) return false;
	const uint8_t *h = in.traw + e.raw_off;
	const uint32_t fc = be16(p + at + 2);
	if ((fc < 1) || (e.raw_len != 4 + 8 * fc)) return false;
	for (uint32_t j = 0; j < 4; j++) if (h[j] != p[at + j]) return false;
	uint32_t q = at + 4;
	for (uint32_t i = 0; i < fc; i++) {
		const uint8_t *s = h + 4 + 8 * i;
		if ((q + 4) > len) return false;
		const bool ent = (p[q] & 0x80) != 0;
		const uint32_t sz = ent ? 8 : 4;
		if (((q - at + sz) > avail) || ((q + sz) > len)) return false;
		if (((p[q] & 0x7F) != s[0]) || (p[q + 1] != s[1]) || (p[q + 2] != s[2]) || (p[q + 3] != s[3])) return false;
		for (uint32_t j = 4; j < 8; j++) if (s[j] != (ent ? p[q + j] : 0)) return false;
		q += sz;
	}
	return true;
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

/* Per template: records and runs, added up per block in LDS first (a batch has
 * a handful of templates, and a hundred thousand atomic adds to two addresses
 * take longer than everything else here), then into one of the copies */
#define XW_TCACHE 32
struct wtotals { uint32_t t[XW_TCACHE], rec[XW_TCACHE], run[XW_TCACHE]; unsigned long long all; uint32_t kmax, left; };

static __device__ __forceinline__ uint8_t *
wtotals_copy(void *sums, uint32_t nt)
{
	return (uint8_t *)sums + 32 + (size_t)(blockIdx.x % XE_WCOPIES) * XE_WCOPY_BYTES(nt);
}

static __device__ __forceinline__ void
wtotals_init(wtotals &T)
{
	if (threadIdx.x < XW_TCACHE) { T.t[threadIdx.x] = 0xFFFFFFFFu; T.rec[threadIdx.x] = T.run[threadIdx.x] = 0; }
	if (threadIdx.x == 0) { T.all = 0; T.kmax = T.left = 0; }
	__syncthreads();
}

/* all lanes of the wave call this together; mine = template or 0xFFFFFFFF */
static __device__ __forceinline__ void
wtotals_add(wtotals &T, uint32_t mine, uint32_t nrec, void *sums, uint32_t nt)
{
	const uint32_t lane = threadIdx.x & 63;
	unsigned long long todo = __ballot(mine != 0xFFFFFFFFu);
	while (todo) {
		const int leader = __ffsll((long long)todo) - 1;
		const uint32_t t = __shfl(mine, leader, 64);
		const bool same = mine == t;
		const unsigned long long grp = __ballot(same);
		uint32_t sum = same ? nrec : 0;
		for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d, 64);
		if (lane == (uint32_t)leader) {
			/* the block's entry of this template, or memory when the block has met more than it holds */
			uint32_t e = t % XW_TCACHE, tries = 0;
			for (; tries < XW_TCACHE; tries++, e = (e + 1) % XW_TCACHE) {
				const uint32_t was = atomicCAS(&T.t[e], 0xFFFFFFFFu, t);
				if ((was == 0xFFFFFFFFu) || (was == t)) break;
			}
			if (tries < XW_TCACHE) {
				atomicAdd(&T.rec[e], sum);
				atomicAdd(&T.run[e], (uint32_t)__popcll(grp));
			} else {
				uint32_t *g = (uint32_t *)(wtotals_copy(sums, nt) + 16);
				atomicAdd(&g[t], sum);
				atomicAdd(&g[nt + 2 + t], (uint32_t)__popcll(grp));
			}
		}
		todo &= ~grp;
	}
}

/* all threads of the block; `all` = this thread's records of any template */
static __device__ __forceinline__ void
wtotals_flush(wtotals &T, unsigned long long all, void *sums, uint32_t nt)
{
	uint8_t *c = wtotals_copy(sums, nt);
	uint32_t *g = (uint32_t *)(c + 16);
	for (int d = 32; d; d >>= 1) all += __shfl_down(all, d, 64);
	if (((threadIdx.x & 63) == 0) && all) atomicAdd(&T.all, all);
	__syncthreads();
	if ((threadIdx.x < XW_TCACHE) && (T.t[threadIdx.x] != 0xFFFFFFFFu)) {
		atomicAdd(&g[T.t[threadIdx.x]], T.rec[threadIdx.x]);
		atomicAdd(&g[nt + 2 + T.t[threadIdx.x]], T.run[threadIdx.x]);
	}
	if (threadIdx.x == 0) {
		if (T.all) atomicAdd((unsigned long long *)c, T.all);
		if (T.kmax) atomicMax((uint32_t *)(c + 8), T.kmax);
		if (T.left) atomicAdd((uint32_t *)(c + 12), T.left);
	}
}

/* A datagram of the usual shape yields up to XE_WALK_SETS runs, one per data
 * set: runs[k * n + i] is run k of datagram i (nrec 0: none). */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_shape_kernel(const xe_walk_in in, xe_dgrun *runs, uint8_t *regular, void *sums)
{
	__shared__ wtotals T;
	/* the first 32 bytes of every thread's datagram (dword d of thread t at
	 * hw[d][t]): datagram header and first set header, fetched with two wide loads
	 * instead of a dozen dependent narrow ones */
	__shared__ uint32_t hw[8][XW_BLOCK];
	const uint32_t i = blockIdx.x * XW_BLOCK + threadIdx.x;
	const size_t n = in.n;
	uint32_t k = 0;
	uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
	bool ok = false;

	static_assert(XE_WALK_SETS == 4, "the shape kernel notes four runs per datagram");
	wtotals_init(T);

	if ((i < in.n) && !(in.kinds && (in.kinds[i] == XE_KIND_SFLOW))) {
		const uint64_t off = in.offs[i];
		int64_t len64 = (int64_t)(in.offs[i + 1] - off);
		const uint32_t len = (len64 > 65536) ? 65536u : (len64 < 0) ? 0u : (uint32_t)len64;
		const uint8_t *p = in.pkt + off;
		const uint32_t src = in.src[i];
		xe_dgrun r = { 0, 0, 0, 0 };
		const uint32_t have = (len >= 32) ? 32u : 0u;
		/* (what the thread lists, kept for the sums at the end) */
#define XW_NOTE(K, R) do { if ((K) == 0) { n0 = (R).nrec; t0 = (R).tmpl & 0xFFFFu; } else if ((K) == 1) { n1 = (R).nrec; t1 = (R).tmpl & 0xFFFFu; } \
	else if ((K) == 2) { n2 = (R).nrec; t2 = (R).tmpl & 0xFFFFu; } else { n3 = (R).nrec; t3 = (R).tmpl & 0xFFFFu; } } while (0)
		if (have) {
			u32x4 a, b;
			__builtin_memcpy(&a, p, 16);
			__builtin_memcpy(&b, p + 16, 16);
			hw[0][threadIdx.x] = a.x; hw[1][threadIdx.x] = a.y; hw[2][threadIdx.x] = a.z; hw[3][threadIdx.x] = a.w;
			hw[4][threadIdx.x] = b.x; hw[5][threadIdx.x] = b.y; hw[6][threadIdx.x] = b.z; hw[7][threadIdx.x] = b.w;
		}
		/* byte x of the datagram (callers keep x below len) */
		auto B = [&](uint32_t x) -> uint32_t {
			return (x < have) ? ((hw[x >> 2][threadIdx.x] >> (8 * (x & 3))) & 0xFFu) : (uint32_t)p[x];
		};
		auto H16 = [&](uint32_t x) -> uint32_t { return (B(x) << 8) | B(x + 1); };
		auto W32 = [&](uint32_t x) -> uint32_t {     /* four bytes in memory order, as ld32() */
			if (!(x & 3) && (x + 4 <= have)) return hw[x >> 2][threadIdx.x];
			return B(x) | (B(x + 1) << 8) | (B(x + 2) << 16) | (B(x + 3) << 24);
		};
		if (len >= 24) {
			const uint32_t ver = H16(0);
			if (ver == 5) {
				/* parse_netflow_v5(), netflow.c:727-786 */
				const uint32_t nr = H16(2), sid = B(21);
				if (nr && ((24 + 48 * nr) == len) && in.v5_slot_plus1 && in.twalk[in.v5_slot_plus1 - 1].fast
					&& exporter_plain(in, src, sid, r.rate)) {
					r.nrec = nr;
					r.tmpl = (in.v5_slot_plus1 - 1) | (24u << 16);
					r.dev_id = __builtin_bswap32(sid);
					runs[i] = r;
					XW_NOTE(0, r);
					k = 1;
					ok = true;
				}
			} else if (ver == 9) {
				/* parse_netflow_v9() + parse_netflow_v9_flowset(), netflow.c:319-365, :246-317:
				 * template flowsets that repeat, byte for byte, the template the handle holds
				 * (parse_netflow_v9_template(), netflow.c:147-186: the first template of the
				 * flowset is compared, the cursor moves on by its size), then at most one data
				 * set, which ends with the datagram and holds header.count complete records.
				 * Whatever else (new or changed templates, flowset headers or templates cut by
				 * the end of the datagram, a second data set - which the reference misreads)
				 * is the host's */
				const uint32_t count = H16(2), sid = W32(16);
				uint32_t ptr = 20;
				bool bad = false;
				while (ptr < len) {
					if ((ptr + 4) > len) { bad = true; break; }
					const uint32_t fsid = H16(ptr), length = H16(ptr + 2);
					if (fsid == 0) {
						xe_thent e;
						if ((ptr + 8) > len) { bad = true; break; }
						const uint32_t tsize = 4 + 4 * H16(ptr + 6);
						if ((tsize > length) || ((ptr + 4 + tsize) > len)
							|| !tmpl_lookup(in, (9u << 16) | B(ptr + 4) | (B(ptr + 5) << 8), src, sid, e)
							|| !tmpl_same(in, e, p + ptr + 4, tsize)) { bad = true; break; }
						ptr += 4 + tsize;
						continue;
					}
					if (fsid == 1) break;
					{
						xe_thent e;
						bad = true;
						if (((ptr + length) == len) && count
							&& tmpl_lookup(in, (9u << 16) | B(ptr) | (B(ptr + 1) << 8), src, sid, e)
							&& (e.slot < in.ntmpl)) {
							const xe_twalk tw = in.twalk[e.slot];
							if (tw.fast && (count * tw.stride <= length) && ((ptr + 4 + count * tw.stride) <= len)
								&& exporter_plain(in, src, sid, r.rate)) {
								r.nrec = count;
								r.tmpl = e.slot | ((ptr + 4) << 16);
								r.dev_id = __builtin_bswap32(sid);
								runs[i] = r;
								XW_NOTE(0, r);
								k = 1;
								bad = false;
							}
						}
						break;
					}
				}
				ok = !bad;
			} else if (ver == 10) {
				/* parse_ipfix() + parse_ipfix_flowset(), netflow.c:628-678, :547-625: data
				 * sets of known templates with complete records, template sets whose first
				 * template (the one parse_ipfix_template() looks at, netflow.c:435-485) is
				 * the one the handle holds, options template sets (skipped there too) and set
				 * ids that mean nothing between them; a new or changed template, a set
				 * length below 4 or a header cut by the end of the datagram leave the
				 * datagram to the host */
				const uint32_t sid = W32(12);
				uint32_t ptr = 16;
				bool bad = false;
				while (ptr < len) {
					if ((ptr + 4) > len) { bad = true; break; }
					const uint32_t fsid = H16(ptr), length = H16(ptr + 2);
					if (length < 4) { bad = true; break; }
					if (fsid == 2) {
						xe_thent e;
						if (((ptr + 8) > len) || (length < 8)
							|| !tmpl_lookup(in, (10u << 16) | B(ptr + 4) | (B(ptr + 5) << 8), src, sid, e)
							|| !tmpl_same_ipfix(in, e, p, ptr + 4, length - 4, len)) { bad = true; break; }
					} else if (fsid > 255) {
						xe_thent e;
						bad = true;
						if ((k < XE_WALK_SETS)
							&& tmpl_lookup(in, (10u << 16) | B(ptr) | (B(ptr + 1) << 8), src, sid, e)
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
									XW_NOTE(k, r);
									k++;
									bad = false;
								}
							}
						}
						if (bad) break;
					}
					ptr += length;
				}
				ok = !bad && (!k || exporter_plain(in, src, sid, r.rate));
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
	/* the most runs any datagram has: the later steps look at that many only;
	 * how many Netflow datagrams are left to the host (none, on a collector
	 * whose exporters' templates are known: the per-datagram flags then need not
	 * travel); and the per-template sums of what was listed, which hold when
	 * nothing is left to the host */
	{
		uint32_t m = k;
		unsigned long long all = 0;
		const unsigned long long left = __ballot((i < in.n) && !ok && !(in.kinds && (in.kinds[i] == XE_KIND_SFLOW)));
		for (int d = 32; d; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = (o > m) ? o : m; }
		if ((threadIdx.x & 63) == 0) {
			if (m) atomicMax(&T.kmax, m);
			if (left) atomicAdd(&T.left, (uint32_t)__popcll(left));
		}
		for (uint32_t q = 0; q < m; q++) {
			uint32_t mine = 0xFFFFFFFFu, nrec = 0;
			if (q < k) {
				mine = (q == 0) ? t0 : (q == 1) ? t1 : (q == 2) ? t2 : t3;
				nrec = (q == 0) ? n0 : (q == 1) ? n1 : (q == 2) ? n2 : n3;
				all += nrec;
			}
			wtotals_add(T, mine, nrec, sums, in.ntmpl);
		}
		wtotals_flush(T, all, sums, in.ntmpl);
	}
}

/* ---- from the datagrams' runs to the fast kernel's run and tile tables ----
 *
 * The fast kernel wants the runs grouped by template (a wave decodes 64 records
 * of ONE template), in arrival order inside a group, and the tiles of all groups
 * listed in the order of the batch index of their first record.  With the run
 * slots in arrival order (slot j = datagram * nsets + set) that is:
 *   position of a run inside its group = records of the same template in the
 *       slots before it  -> a prefix sum per template.  Batches carry a handful
 *       of distinct templates, so the sums of up to XE_WTUPLE templates are
 *       scanned together (one pass per XE_WTUPLE templates a batch uses);
 *   batch index of a run's first record = records of any template in the slots
 *       before it, and place of a tile in the tile list = tiles that start in the
 *       slots before it (a tile starts wherever a group position is a multiple
 *       of 64): one more scan, over both numbers packed in 64 bits.
 * No sort is involved.  Both scans have the same three steps - sums of chunks of
 * XW_CHUNK slots, prefix sums over the chunks (one small block), prefix sums
 * inside every chunk - and the step that finishes one scan prepares the next:
 * five short kernels from runs to tables. */
#define XW_ITEMS 4
#define XW_CHUNK (XW_BLOCK * XW_ITEMS)

/* what run slot j holds: its template's place in this pass's tuple (0xFF = no
 * run, or a template of another pass) and its record count */
struct wslot { uint32_t a, nrec, tmpl; bool walked, run; };

static __device__ __forceinline__ wslot
slot_of(const xe_walk_in &in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint8_t *tmap, size_t j, xe_dgrun &r)
{
	wslot s = { 0xFFu, 0, 0, false, false };
	const size_t n = in.n;
	r = (xe_dgrun){ 0, 0, 0, 0 };
	if (j >= n * nsets) return s;
	const size_t i = j / nsets;
	const uint32_t k = (uint32_t)(j - i * nsets);
	s.walked = walked[i] != 0;
	if (!s.walked) return s;
	r = runs[k * n + i];
	if (!r.nrec) return s;
	s.run = true;
	s.nrec = r.nrec;
	s.tmpl = r.tmpl & 0xFFFFu;
	s.a = tmap[s.tmpl];
	return s;
}

/* exclusive prefix sum of v over the block's threads; *total = the block's sum */
static __device__ __forceinline__ unsigned long long
block_scan64(unsigned long long v, unsigned long long *wsum, unsigned long long *total)
{
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	unsigned long long x = v;
	for (int d = 1; d < 64; d <<= 1) {
		const unsigned long long y = __shfl_up(x, d, 64);
		if (lane >= (uint32_t)d) x += y;
	}
	__syncthreads();                      /* (wsum may still be read from the scan before) */
	if (lane == 63) wsum[wave] = x;
	__syncthreads();
	unsigned long long before = 0, all = 0;
	for (uint32_t w = 0; w < XW_BLOCK / 64; w++) {
		const unsigned long long t = wsum[w];
		if (w < wave) before += t;
		all += t;
	}
	*total = all;
	return before + x - v;
}

/* the sums of the shape kernel once more, after the host walk: over the
 * datagrams still listed here, plus what the host made of the others */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_totals_kernel(const xe_walk_in in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	void *sums)
{
	__shared__ wtotals T;
	const size_t n = in.n;
	unsigned long long all = 0;

	wtotals_init(T);
	for (size_t i0 = (size_t)blockIdx.x * XW_BLOCK; i0 < n; i0 += (size_t)gridDim.x * XW_BLOCK) {
		const size_t i = i0 + threadIdx.x;
		const bool w = (i < n) && (walked[i] != 0);
		if ((i < n) && !w && hostfirst) all += hostfirst[i + 1] - hostfirst[i];
		for (uint32_t k = 0; k < nsets; k++) {
			uint32_t mine = 0xFFFFFFFFu, nrec = 0;
			if (w) {
				const xe_dgrun r = runs[k * n + i];
				if (r.nrec) { mine = r.tmpl & 0xFFFFu; nrec = r.nrec; all += nrec; }
			}
			wtotals_add(T, mine, nrec, sums, in.ntmpl);
		}
	}
	wtotals_flush(T, all, sums, in.ntmpl);
}

/* step 1 of the tuple scan: per chunk and template of the pass, runs << 32 | records */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_tuple_sums_kernel(const xe_walk_in in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint8_t *tmap,
	unsigned long long *csum)
{
	__shared__ unsigned long long acc[XE_WTUPLE];
	const uint32_t lane = threadIdx.x & 63;
	const size_t j0 = (size_t)blockIdx.x * XW_CHUNK + (size_t)threadIdx.x * XW_ITEMS;

	if (threadIdx.x < XE_WTUPLE) acc[threadIdx.x] = 0;
	__syncthreads();
	for (int q = 0; q < XW_ITEMS; q++) {
		xe_dgrun r;
		const wslot s = slot_of(in, runs, nsets, walked, tmap, j0 + q, r);
		unsigned long long todo = __ballot(s.a != 0xFFu);
		while (todo) {
			const int leader = __ffsll((long long)todo) - 1;
			const uint32_t a = __shfl(s.a, leader, 64);
			const bool same = s.a == a;
			const unsigned long long grp = __ballot(same);
			uint32_t sum = same ? s.nrec : 0;
			for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d, 64);
			if (lane == (uint32_t)leader) atomicAdd(&acc[a], ((unsigned long long)__popcll(grp) << 32) | sum);
			todo &= ~grp;
		}
	}
	__syncthreads();
	if (threadIdx.x < XE_WTUPLE) csum[(size_t)blockIdx.x * XE_WTUPLE + threadIdx.x] = acc[threadIdx.x];
}

/* step 2 of either scan: exclusive prefix sums over the chunks, column by
 * column (one wave per column of `width` columns; the halves of a value are
 * summed apart: neither carries into the other) */
__global__ void __launch_bounds__(64 * XE_WTUPLE)
xw_chunk_scan_kernel(const unsigned long long *csum, unsigned long long *cpre, uint32_t nchunks, uint32_t width)
{
	const uint32_t lane = threadIdx.x & 63, col = threadIdx.x >> 6;
	unsigned long long carry = 0;
	if (col >= width) return;
	for (uint32_t c0 = 0; c0 < nchunks; c0 += 64) {
		const uint32_t c = c0 + lane;
		const unsigned long long v = (c < nchunks) ? csum[(size_t)c * width + col] : 0;
		unsigned long long x = v;
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long y = __shfl_up(x, d, 64);
			if (lane >= (uint32_t)d) x += y;
		}
		if (c < nchunks) cpre[(size_t)c * width + col] = carry + x - v;
		carry += __shfl(x, 63, 64);
	}
}

/* step 2 done by the consumer itself (few chunks): the sum of column `col` of
 * the chunks before this block's; all threads of the block call it */
static __device__ __forceinline__ unsigned long long
chunk_prefix(const unsigned long long *csum, uint32_t width, uint32_t col, unsigned long long *wsum)
{
	unsigned long long t = 0, all;
	for (uint32_t c = threadIdx.x; c < blockIdx.x; c += XW_BLOCK) t += csum[(size_t)c * width + col];
	(void)block_scan64(t, wsum, &all);
	return all;
}

/* step 3 of the tuple scan: every run whose template is in the pass gets its
 * two prefix sums (pos[j] = {records, runs of its template before it}) and
 * notes what the second scan adds up: the tiles that start inside it (low
 * word) and its records (high word).  Slots without a run note theirs in the
 * first pass: nothing, or - first slot of a datagram the host walked - the
 * records the host walk made of it.  With one pass the sums of the chunks for
 * the second scan are taken here as well (vsum). */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_tuple_apply_kernel(const xe_walk_in in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	const uint8_t *tmap, const unsigned long long *cpre, int inline_prefix, uint32_t nact, int first_pass, uint2 *pos,
	unsigned long long *val, unsigned long long *vsum)
{
	__shared__ unsigned long long wsum[XW_BLOCK / 64];
	const size_t n = in.n, nslots = n * nsets;
	const size_t j0 = (size_t)blockIdx.x * XW_CHUNK + (size_t)threadIdx.x * XW_ITEMS;
	wslot s[XW_ITEMS];
	unsigned long long myval = 0;
	uint32_t px[XW_ITEMS], py[XW_ITEMS];

	for (int q = 0; q < XW_ITEMS; q++) {
		xe_dgrun r;
		s[q] = slot_of(in, runs, nsets, walked, tmap, j0 + q, r);
		px[q] = py[q] = 0;
	}
	for (uint32_t a = 0; a < nact; a++) {
		unsigned long long x[XW_ITEMS], t = 0, all;
		for (int q = 0; q < XW_ITEMS; q++) {
			x[q] = (s[q].a == a) ? ((1ull << 32) | s[q].nrec) : 0;
			t += x[q];
		}
		/* (cpre: the chunks' sums themselves when this block adds up those before it) */
		unsigned long long before = inline_prefix ? chunk_prefix(cpre, XE_WTUPLE, a, wsum) : cpre[(size_t)blockIdx.x * XE_WTUPLE + a];
		before += block_scan64(t, wsum, &all);
		for (int q = 0; q < XW_ITEMS; q++) {
			if (s[q].a == a) { px[q] = (uint32_t)before; py[q] = (uint32_t)(before >> 32); }
			before += x[q];
		}
	}
	for (int q = 0; q < XW_ITEMS; q++) {
		const size_t j = j0 + q;
		if (j >= nslots) continue;
		if (s[q].a != 0xFFu) {
			const unsigned long long v = (unsigned long long)((px[q] + s[q].nrec + 63) / 64 - (px[q] + 63) / 64) | ((unsigned long long)s[q].nrec << 32);
			pos[j] = make_uint2(px[q], py[q]);
			val[j] = v;
			myval += v;
		} else if (first_pass && !s[q].run) {
			const size_t i = j / nsets;
			const unsigned long long v = (!s[q].walked && hostfirst && (j == i * nsets)) ? ((unsigned long long)(hostfirst[i + 1] - hostfirst[i]) << 32) : 0ull;
			val[j] = v;
			myval += v;
		}
	}
	if (vsum) {
		unsigned long long all;
		(void)block_scan64(myval, wsum, &all);
		if (threadIdx.x == 0) vsum[blockIdx.x] = all;
	}
}

/* step 1 of the second scan when the tuple scan took several passes */
__global__ void __launch_bounds__(XW_BLOCK)
xw_val_sums_kernel(const unsigned long long *val, size_t nslots, unsigned long long *vsum)
{
	__shared__ unsigned long long wsum[XW_BLOCK / 64];
	const size_t j0 = (size_t)blockIdx.x * XW_CHUNK + (size_t)threadIdx.x * XW_ITEMS;
	unsigned long long t = 0, all;
	for (int q = 0; q < XW_ITEMS; q++) if (j0 + q < nslots) t += val[j0 + q];
	(void)block_scan64(t, wsum, &all);
	if (threadIdx.x == 0) vsum[blockIdx.x] = all;
}

/* step 3 of the second scan, and what it was for: the run table (grouped by
 * template, arrival order inside a group), the tile list (in the order of the
 * batch index of each tile's first record) and recbase[i] = batch index of
 * datagram i's first record */
__global__ void __launch_bounds__(XW_BLOCK)
xw_nf_runs_kernel(const xe_walk_in in, const xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint2 *pos,
	const unsigned long long *val, const unsigned long long *vpre, int inline_prefix, const xe_wgroup *groups, xe_frun *out,
	uint32_t nruns, xe_ftile *tiles, uint32_t ntiles, uint32_t *recbase)
{
	__shared__ unsigned long long wsum[XW_BLOCK / 64];
	const size_t n = in.n, nslots = n * nsets;
	const size_t j0 = (size_t)blockIdx.x * XW_CHUNK + (size_t)threadIdx.x * XW_ITEMS;
	unsigned long long v[XW_ITEMS], t = 0, all;

	for (int q = 0; q < XW_ITEMS; q++) {
		v[q] = (j0 + q < nslots) ? val[j0 + q] : 0;
		t += v[q];
	}
	unsigned long long x = inline_prefix ? chunk_prefix(vpre, 1, 0, wsum) : vpre[blockIdx.x];
	x += block_scan64(t, wsum, &all);
	for (int q = 0; q < XW_ITEMS; q++, x += v[q - 1]) {
		const size_t j = j0 + q;
		if (j > nslots) break;
		const uint32_t first_rec = (uint32_t)(x >> 32);
		if (j == nslots) { recbase[n] = first_rec; break; }
		const size_t i = j / nsets;
		const uint32_t k = (uint32_t)(j - i * nsets);
		if (k == 0) recbase[i] = first_rec;
		if (!walked[i]) continue;
		const xe_dgrun r = runs[k * n + i];
		if (!r.nrec) continue;
		const uint32_t tm = r.tmpl & 0xFFFFu;
		const xe_wgroup g = groups[tm];
		const uint2 p = pos[j];
		xe_frun f;
		f.data_off = (uint32_t)in.offs[i] + (r.tmpl >> 16);
		f.first_rec = first_rec;
		f.start = g.pos0 + p.x;
		f.nrec = r.nrec;
		f.stride = in.twalk[tm].stride | (r.rate & 0x80000000u);
		if (r.rate & 0x80000000u) {
			/* an entry with mark filters (or one that drops unmarked records) is named
			 * in the run: the decode kernel runs its filters */
			const int d = dev_find(in, in.src[i], __builtin_bswap32(r.dev_id));
			if (d >= 0) f.stride |= (in.devs[d].rules & 0x7FFFu) << 16;
		}
		f.rate = r.rate & 0x7FFFFFFFu;
		f.dev_ip = __builtin_bswap32(in.src[i]);
		f.dev_id = r.dev_id;
		/* (the tables were sized from the sums of another pass over the same runs: a
		 * place beyond them cannot come up, and must not be written if it did) */
		if (g.run0 + p.y >= nruns) continue;
		out[g.run0 + p.y] = f;
		/* the tiles that start inside this run */
		uint32_t at = (uint32_t)x;
		for (uint32_t tq = (p.x + 63) / 64; ((uint64_t)tq * 64 < (uint64_t)p.x + r.nrec) && (at < ntiles); tq++, at++) {
			xe_ftile ft;
			ft.first_run = g.run0 + p.y;
			ft.tmpl = tm;
			ft.base = g.pos0 + 64 * tq;
			ft.pad = first_rec + (64 * tq - p.x);
			tiles[at] = ft;
		}
	}
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
	size_t a = 0;
	uint32_t *z = nullptr;
	if (!n) n = 1;
	(void)rocprim::exclusive_scan(nullptr, a, z, z, 0u, n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0);
	return a + 256;
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
xe_walk_sf_count(const struct xe_walk_in *in, uint32_t *cnt, struct xe_sfdg *dgs, struct xe_sfagent *agents,
	unsigned long long *total, void *stream)
{
	if (!in->n) return XE_OK;
	hipLaunchKernelGGL(xw_sf_count_kernel, dim3(xw_blocks(in->n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, cnt, dgs, agents, total);
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
xe_walk_nf_shape(const struct xe_walk_in *in, struct xe_dgrun *runs, uint8_t *regular, void *sums, void *stream)
{
	if (!in->n) return XE_OK;
	hipLaunchKernelGGL(xw_nf_shape_kernel, dim3(xw_blocks(in->n)), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, runs, regular, sums);
	return xw_done();
}

extern "C" int
xe_walk_nf_totals(const struct xe_walk_in *in, const struct xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	void *sums, void *stream)
{
	uint32_t blocks = xw_blocks(in->n);
	if (!in->n) return XE_OK;
	if (blocks > 256) blocks = 256;
	hipLaunchKernelGGL(xw_nf_totals_kernel, dim3(blocks), dim3(XW_BLOCK), 0, (hipStream_t)stream, *in, runs, nsets, walked, hostfirst, sums);
	return xw_done();
}

extern "C" size_t
xe_walk_nf_chunks(size_t n, uint32_t nsets)
{
	return (n * nsets + 1 + XW_CHUNK - 1) / XW_CHUNK;
}

extern "C" int
xe_walk_nf_tables(const struct xe_walk_in *in, const struct xe_dgrun *runs, uint32_t nsets, const uint8_t *walked, const uint32_t *hostfirst,
	const uint8_t *tmaps, uint32_t npass, uint32_t nactive, const struct xe_wgroup *groups, void *work,
	struct xe_frun *out, uint32_t nruns, struct xe_ftile *tiles, uint32_t ntiles, uint32_t *recbase, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	const size_t nslots = (size_t)in->n * nsets;
	const uint32_t nchunks = (uint32_t)xe_walk_nf_chunks(in->n, nsets);
	uint8_t *wk = (uint8_t *)work;
	/* the layout xe_walk_nf_work_bytes() sizes */
	unsigned long long *csum = (unsigned long long *)wk; wk += (size_t)nchunks * XE_WTUPLE * 8;
	unsigned long long *cpre = (unsigned long long *)wk; wk += (size_t)nchunks * XE_WTUPLE * 8;
	unsigned long long *vsum = (unsigned long long *)wk; wk += (size_t)nchunks * 8;
	unsigned long long *vpre = (unsigned long long *)wk; wk += (size_t)nchunks * 8;
	uint2 *pos = (uint2 *)wk; wk += (nslots + 1) * 8;
	unsigned long long *val = (unsigned long long *)wk;

	if (!in->n) return XE_OK;
	/* few chunks: the kernels that need the sums of the chunks before theirs add them up
	 * themselves (a launch costs more than reading a few hundred numbers per block) */
	const int inl = nchunks <= 2048;
	for (uint32_t p = 0; p < npass; p++) {
		const uint8_t *tmap = tmaps + (size_t)p * (in->ntmpl + 1);
		const uint32_t nact = (nactive > p * XE_WTUPLE) ? ((nactive - p * XE_WTUPLE > XE_WTUPLE) ? XE_WTUPLE : nactive - p * XE_WTUPLE) : 0;
		hipLaunchKernelGGL(xw_nf_tuple_sums_kernel, dim3(nchunks), dim3(XW_BLOCK), 0, st, *in, runs, nsets, walked, tmap, csum);
		if (!inl) hipLaunchKernelGGL(xw_chunk_scan_kernel, dim3(1), dim3(64 * XE_WTUPLE), 0, st, csum, cpre, nchunks, (uint32_t)XE_WTUPLE);
		hipLaunchKernelGGL(xw_nf_tuple_apply_kernel, dim3(nchunks), dim3(XW_BLOCK), 0, st, *in, runs, nsets, walked, hostfirst, tmap,
			inl ? csum : cpre, inl, nact, (int)(p == 0), pos, val, (npass == 1) ? vsum : (unsigned long long *)nullptr);
	}
	if (npass != 1) hipLaunchKernelGGL(xw_val_sums_kernel, dim3(nchunks), dim3(XW_BLOCK), 0, st, val, nslots, vsum);
	if (!inl) hipLaunchKernelGGL(xw_chunk_scan_kernel, dim3(1), dim3(64 * XE_WTUPLE), 0, st, vsum, vpre, nchunks, 1u);
	hipLaunchKernelGGL(xw_nf_runs_kernel, dim3(nchunks), dim3(XW_BLOCK), 0, st, *in, runs, nsets, walked, pos, val, inl ? vsum : vpre, inl,
		groups, out, nruns, tiles, ntiles, recbase);
	return xw_done();
}

extern "C" size_t
xe_walk_nf_work_bytes(size_t n, uint32_t nsets)
{
	const size_t nchunks = xe_walk_nf_chunks(n, nsets);
	return nchunks * XE_WTUPLE * 8 * 2 + nchunks * 8 * 2 + (n * nsets + 1) * 8 * 2 + 256;
}
