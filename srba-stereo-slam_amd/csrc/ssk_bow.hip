// ssk_bow.hip -- bag-of-words place recognition on the device: the vocabulary-tree transform and the L1 database
// query behind CBoWManager (reference src/CBoWManager.h:59-109; DBoW2 1.0.2 BriefVocabulary / BriefDatabase,
// loaded with weightingType TF_IDF and scoringType L1_NORM from demo/voc.yml.gz).  Contract W1-W5: oracle/oracle_bow.c.
//
// Data layout in HBM
//   vocabulary   nodes renumbered breadth-first so that the children of a node are contiguous and in file order:
//                ndesc uint4[2n] (32-B BRIEF centres, 1.2 MB for the reference's 36 940 nodes: L2 resident),
//                kids int2[n] (first child, child count), leaf_word i32[n], word_weight f64[n_words]
//   keyframe     bow_word i32[B][N] ascending word ids, bow_val f64[B][N] L1-normalised values, bow_n i32[B]
//   database     forward lists stored ENTRY-MINOR: words i32[stride][cap], vals f64[stride][cap], ent_n i32[cap]: the query
//                kernel gives every entry a lane, and the c-th words of 64 consecutive entries are one coalesced row; an
//                entry is appended by a small scatter kernel, nothing is re-laid out as the map grows
// Kernels
//   bow_words_kernel   8 lanes per descriptor, one tree level per step: each lane takes one child (256 coalesced bytes per
//                      8-child node), popcount of 8 dwords, (distance, child) min over the 8 lanes (first child wins ties)
//   bow_vector_kernel  one workgroup per keyframe: bitonic sort of the word ids in LDS, run lengths, weight of a run by
//                      repeated addition (the f64 sequence of W3), serial L1 norm in ascending word order, division
//   bow_score_kernel   one wave per (query, 64 entries), one LANE per entry: the lane walks its entry's ascending words; a bitmap
//                      of the query's words in LDS rejects most, a per-dword prefix count turns a hit into the query index
//                      (rank) without a search; the lane adds its terms in word order, which is the f64 sum of W5
//   bow_topk_kernel    one workgroup per query: max_results rounds of a lexicographic (score, entry id) arg-min
// Everything is integer work except the two serial f64 sums, which reproduce the reference order bit for bit.
#include "ssk_frontend.h"
#include <climits>
#include <cstdlib>
#include <cmath>
#include <vector>

struct ssk_bow_vocab {
	ssk_ctx* ctx = nullptr;
	int k = 0, L = 0, n_nodes = 0, n_words = 0;
	uint4* ndesc = nullptr;
	int2* kids = nullptr;
	int32_t* leaf_word = nullptr;
	double* word_weight = nullptr;
};

struct ssk_bow_db {
	ssk_ctx* ctx = nullptr;
	const ssk_bow_vocab* voc = nullptr;
	int cap = 0, stride = 0, n_entries = 0;
	int32_t* words = nullptr;
	double* vals = nullptr;
	int32_t* ent_n = nullptr;
	double* scores = nullptr;     // [q_cap][cap] raw L1 sums of the last query call (+inf: no common word)
	int q_cap = 0;
};

namespace {

constexpr int BOW_SORT_MAX = SSK_MAX_FEATS;     // 2048 word ids per keyframe
constexpr int BOW_BITMAP_WORDS_MAX = 1 << 20;   // LDS budget of the query bitmap + its prefix counts: 128 KB (vocabularies up to 512 Ki words)

__device__ __forceinline__ int popc256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
	return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
	       __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// W2.  desc rows of keyframe kf start at row (kf * kf_row_stride); n_rows[kf * n_stride] of them are valid.
__global__ __launch_bounds__(256)
void bow_words_kernel(const uint4* __restrict__ ndesc, const int2* __restrict__ kids, const int32_t* __restrict__ leaf_word,
                      const double* __restrict__ word_weight, const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_rows,
                      int n_stride, size_t kf_row_stride, int kf_begin, int n_kf, int max_feats, int32_t* __restrict__ out_word)
{
	const long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	const int sub = threadIdx.x & 7;
	if (g >= (long long)n_kf * max_feats) return;
	const int kf = kf_begin + (int)(g / max_feats), i = (int)(g % max_feats);
	int32_t* o = out_word + (size_t)kf * max_feats + i;
	if (i >= n_rows[(size_t)kf * n_stride]) { if (sub == 0) *o = INT_MAX; return; }
	const uint4* f = reinterpret_cast<const uint4*>(desc + 32 * ((size_t)kf * kf_row_stride + i));
	const uint4 f0 = f[0], f1 = f[1];
	int node = 0;
	while (true) {
		const int2 kd = kids[node];
		if (kd.y == 0) break;
		int key = INT_MAX;
		for (int c = sub; c < kd.y; c += 8) {
			const uint4* d = ndesc + 2 * (size_t)(kd.x + c);
			const int dist = popc256(f0, f1, d[0], d[1]);
			key = min(key, (dist << 20) | c);                     // children in file order: the lowest c wins ties
		}
		key = min(key, __shfl_xor(key, 1, 8));
		key = min(key, __shfl_xor(key, 2, 8));
		key = min(key, __shfl_xor(key, 4, 8));
		node = kd.x + (key & 0xFFFFF);
	}
	if (sub == 0) {
		int w = node == 0 ? -1 : leaf_word[node];
		if (w >= 0 && !(word_weight[w] > 0.0)) w = -1;            // "if (w > 0) v.addWeight(id, w)"
		*o = w < 0 ? INT_MAX : w;
	}
}

// W3.  One workgroup (256 threads) per keyframe.
__global__ __launch_bounds__(256)
void bow_vector_kernel(const double* __restrict__ word_weight, const int32_t* __restrict__ n_rows, int n_stride, int kf_begin,
                       int max_feats, int32_t* __restrict__ bow_word, double* __restrict__ bow_val, int32_t* __restrict__ bow_n)
{
	__shared__ int s_key[BOW_SORT_MAX];
	__shared__ int s_ow[BOW_SORT_MAX];
	__shared__ double s_ov[BOW_SORT_MAX];
	__shared__ int s_scan[256];
	__shared__ double s_norm;
	const int kf = kf_begin + blockIdx.x, tid = threadIdx.x;
	const int n = min(n_rows[(size_t)kf * n_stride], max_feats);
	int P = 256;
	while (P < n) P <<= 1;
	int32_t* gw = bow_word + (size_t)kf * max_feats;
	double* gv = bow_val + (size_t)kf * max_feats;
	for (int i = tid; i < P; i += 256) s_key[i] = i < n ? gw[i] : INT_MAX;
	__syncthreads();
	for (int k = 2; k <= P; k <<= 1)
		for (int j = k >> 1; j > 0; j >>= 1) {
			for (int i = tid; i < P; i += 256) {
				const int x = i ^ j;
				if (x > i) {
					const int a = s_key[i], b = s_key[x];
					if (((i & k) == 0) == (a > b)) { s_key[i] = b; s_key[x] = a; }
				}
			}
			__syncthreads();
		}
	// run starts: thread t owns the P/256 consecutive slots [t*per, (t+1)*per)
	const int per = P >> 8, i0 = tid * per;
	int cnt = 0;
	for (int i = i0; i < i0 + per; ++i) cnt += s_key[i] != INT_MAX && (i == 0 || s_key[i] != s_key[i - 1]);
	s_scan[tid] = cnt;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		const int v = tid >= d ? s_scan[tid - d] : 0;
		__syncthreads();
		s_scan[tid] += v;
		__syncthreads();
	}
	int pos = s_scan[tid] - cnt;
	const int nw = s_scan[255];
	for (int i = i0; i < i0 + per; ++i) {
		const int w = s_key[i];
		if (w == INT_MAX || (i > 0 && w == s_key[i - 1])) continue;
		const double wt = word_weight[w];
		double acc = wt;
		for (int j = i + 1; j < P && s_key[j] == w; ++j) acc += wt;
		s_ow[pos] = w; s_ov[pos] = acc; ++pos;
	}
	__syncthreads();
	if (tid == 0) {                                                    // the serial sum of W3, eight LDS reads in flight
		double norm = 0;
		int i = 0;
		for (; i + 8 <= nw; i += 8) {
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) v[u] = s_ov[i + u];
#pragma unroll
			for (int u = 0; u < 8; ++u) norm += fabs(v[u]);
		}
		for (; i < nw; ++i) norm += fabs(s_ov[i]);
		s_norm = norm;
	}
	__syncthreads();
	const double norm = s_norm;
	for (int i = tid; i < nw; i += 256) { gw[i] = s_ow[i]; gv[i] = norm > 0 ? s_ov[i] / norm : s_ov[i]; }
	if (tid == 0) bow_n[kf] = nw;
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
	return __hiloint2double(hi, lo);
}

// W5, raw sums.  grid (n_queries, ceil(n_entries / 64)), 64 threads: lane = entry.  Query q = keyframe q_begin + blockIdx.x of the
// bow_* arrays (stride q_stride).  Dynamic LDS: s_qv f64[BOW_SORT_MAX] | s_qw i32[BOW_SORT_MAX] | bitmap u32[bm_dwords] | prefix
// i32[bm_dwords]; bm_dwords == 0 (vocabulary too large for a bitmap): binary search of s_qw instead.
__global__ __launch_bounds__(64)
void bow_score_kernel(const int32_t* __restrict__ q_word, const double* __restrict__ q_val, const int32_t* __restrict__ q_n, int q_stride,
                      int q_begin, const int32_t* __restrict__ e_word, const double* __restrict__ e_val, const int32_t* __restrict__ e_n,
                      int e_stride, int e_cap, int n_entries, const int32_t* __restrict__ max_ids, int bm_dwords, double* __restrict__ scores,
                      int score_stride)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
	double* s_qv = reinterpret_cast<double*>(s_raw);
	int* s_qw = reinterpret_cast<int*>(s_raw + 8 * BOW_SORT_MAX);
	unsigned* s_bm = reinterpret_cast<unsigned*>(s_raw + 12 * BOW_SORT_MAX);
	int* s_pre = reinterpret_cast<int*>(s_raw + 12 * BOW_SORT_MAX) + bm_dwords;
	const int q = blockIdx.x, kf = q_begin + q, lane = threadIdx.x;
	const int nq = min(q_n[kf], q_stride);
	for (int i = lane; i < bm_dwords; i += 64) s_bm[i] = 0;
	for (int i = lane; i < nq; i += 64) { s_qw[i] = q_word[(size_t)kf * q_stride + i]; s_qv[i] = q_val[(size_t)kf * q_stride + i]; }
	__syncthreads();
	if (bm_dwords > 0) {
		for (int i = lane; i < nq; i += 64) atomicOr(&s_bm[s_qw[i] >> 5], 1u << (s_qw[i] & 31));
		__syncthreads();
		// s_pre[d] = number of query words below dword d: every lane counts a contiguous slice, one wave scan joins them
		const int per = (bm_dwords + 63) / 64, d0 = lane * per, d1 = min(bm_dwords, d0 + per);
		int cnt = 0;
		for (int d = d0; d < d1; ++d) cnt += __popc(s_bm[d]);
		int incl = cnt;
		for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
		int run = incl - cnt;
		for (int d = d0; d < d1; ++d) { s_pre[d] = run; run += __popc(s_bm[d]); }
		__syncthreads();
	}
	const int max_id = max_ids ? max_ids[q] : -1;
	const int e_end = max_id < 0 ? n_entries : min(n_entries, max_id);
	const int e = blockIdx.y * 64 + lane;
	const int ne = e < e_end ? min(e_n[e], e_stride) : 0;
	int nmax = ne;
	for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
	double s = 0;
	int hits = 0;
	// One wave per SIMD is all the parallelism a (query, 64 entries) pair offers, so the loop is written for latency: U rows per
	// pass, the next pass's words already in flight, every LDS read of a pass issued back to back (no branch between them: a
	// miss reads slot 0 and contributes +0.0, which leaves the sum as it is), then the values of the hits, then the U terms are
	// added in row order.
	constexpr int U = 32;
	const int32_t* ew = e_word + e;
	const double* ev = e_val + e;
	int wn[U];
#pragma unroll
	for (int u = 0; u < U; ++u) wn[u] = u < ne ? ew[(size_t)u * e_cap] : -1;
	for (int c0 = 0; c0 < nmax; c0 += U) {
		int w[U];
#pragma unroll
		for (int u = 0; u < U; ++u) w[u] = wn[u];
		if (c0 + U < nmax) {
#pragma unroll
			for (int u = 0; u < U; ++u) wn[u] = c0 + U + u < ne ? ew[(size_t)(c0 + U + u) * e_cap] : -1;
		}
		int qi[U];
		if (bm_dwords > 0) {
			unsigned dw[U];
			int pre[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const int d = max(w[u], 0) >> 5; dw[u] = s_bm[d]; pre[u] = s_pre[d]; }
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const unsigned bit = 1u << (w[u] & 31);
				qi[u] = (w[u] >= 0 && (dw[u] & bit)) ? pre[u] + __popc(dw[u] & (bit - 1)) : -1;
			}
		} else {
#pragma unroll 1
			for (int u = 0; u < U; ++u) {
				qi[u] = -1;
				if (w[u] < 0) continue;
				int lo = 0, hi = nq;
				while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_qw[mid] < w[u]) lo = mid + 1; else hi = mid; }
				if (lo < nq && s_qw[lo] == w[u]) qi[u] = lo;
			}
		}
		double dv[U], qv[U];
#pragma unroll
		for (int u = 0; u < U; ++u) dv[u] = qi[u] >= 0 ? ev[(size_t)(c0 + u) * e_cap] : 0.0;
#pragma unroll
		for (int u = 0; u < U; ++u) qv[u] = s_qv[max(qi[u], 0)];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const double term = fabs(qv[u] - dv[u]) - fabs(qv[u]) - fabs(dv[u]);
			s += qi[u] >= 0 ? term : 0.0;                              // ascending word order within the lane's entry
			hits += qi[u] >= 0;
		}
	}
	if (e < n_entries) scores[(size_t)q * score_stride + e] = hits > 0 ? s : INFINITY;
}

// entry e of the entry-minor database <- a bag-of-words vector (n from the device when d_n != NULL)
__global__ __launch_bounds__(256)
void bow_db_insert_kernel(const int32_t* __restrict__ src_word, const double* __restrict__ src_val, const int32_t* __restrict__ d_n, int n_host,
                          int e, int e_cap, int e_stride, int32_t* __restrict__ e_word, double* __restrict__ e_val, int32_t* __restrict__ e_n)
{
	const int n = min(d_n ? *d_n : n_host, e_stride);
	for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
		e_word[(size_t)c * e_cap + e] = src_word[c];
		e_val[(size_t)c * e_cap + e] = src_val[c];
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) e_n[e] = n;
}

// results of query q: up to max_results entries by (raw sum ascending, entry id ascending); score = -s/2
__global__ __launch_bounds__(256)
void bow_topk_kernel(const double* __restrict__ scores, int score_stride, int n_entries, int max_results, int r_stride, int q_begin,
                     int32_t* __restrict__ r_ids, double* __restrict__ r_scores, int32_t* __restrict__ r_n)
{
	__shared__ double s_s[256];
	__shared__ int s_i[256];
	const int q = blockIdx.x, kf = q_begin + q, tid = threadIdx.x;
	const double* sc = scores + (size_t)q * score_stride;
	double last_s = -INFINITY;
	int last_i = -1, found = 0;
	for (int r = 0; r < max_results; ++r) {
		double bs = INFINITY;
		int bi = INT_MAX;
		for (int e = tid; e < n_entries; e += 256) {
			const double s = sc[e];
			if (s == INFINITY) continue;
			if (s < last_s || (s == last_s && e <= last_i)) continue;          // already reported
			if (s < bs || (s == bs && e < bi)) { bs = s; bi = e; }
		}
		s_s[tid] = bs; s_i[tid] = bi;
		__syncthreads();
		for (int d = 128; d > 0; d >>= 1) {
			if (tid < d) {
				const double os = s_s[tid + d];
				const int oi = s_i[tid + d];
				if (os < s_s[tid] || (os == s_s[tid] && oi < s_i[tid])) { s_s[tid] = os; s_i[tid] = oi; }
			}
			__syncthreads();
		}
		bs = s_s[0]; bi = s_i[0];
		__syncthreads();
		if (bi == INT_MAX) break;
		if (tid == 0) { r_ids[(size_t)kf * r_stride + r] = bi; r_scores[(size_t)kf * r_stride + r] = -bs / 2.0; }
		last_s = bs; last_i = bi; ++found;
	}
	if (tid == 0) r_n[kf] = found;
}

int bow_alloc_batch(ssk_ctx* ctx, ssk_kfbatch* b)
{
	if (b->bow_word) return SSK_OK;
	const size_t B = (size_t)b->n_kf, N = (size_t)b->max_feats;
	SSK_HIP(ctx, hipMalloc(&b->bow_word, B * N * 4));
	SSK_HIP(ctx, hipMalloc(&b->bow_val, B * N * 8));
	SSK_HIP(ctx, hipMalloc(&b->bow_n, B * 4));
	SSK_HIP(ctx, hipMalloc(&b->q_ids, B * SSK_BOW_MAX_RESULTS * 4));
	SSK_HIP(ctx, hipMalloc(&b->q_scores, B * SSK_BOW_MAX_RESULTS * 8));
	SSK_HIP(ctx, hipMalloc(&b->q_n, B * 4));
	SSK_HIP(ctx, hipMemsetAsync(b->bow_n, 0, B * 4, ctx->stream));
	SSK_HIP(ctx, hipMemsetAsync(b->q_n, 0, B * 4, ctx->stream));
	return SSK_OK;
}

int launch_transform(ssk_ctx* ctx, const ssk_bow_vocab* v, const uint8_t* d_desc, const int32_t* d_n_rows, int n_stride,
                     size_t kf_row_stride, int kf_begin, int n_kf, int max_feats, int32_t* bow_word, double* bow_val, int32_t* bow_n)
{
	ssk_stage_scope prof(ctx, SSK_STAGE_BOW);
	const long long threads = (long long)n_kf * max_feats * 8;
	hipLaunchKernelGGL(bow_words_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, v->ndesc, v->kids, v->leaf_word,
	                   v->word_weight, d_desc, d_n_rows, n_stride, kf_row_stride, kf_begin, n_kf, max_feats, bow_word);
	SSK_LAUNCH_CHECK(ctx, "bow_words_kernel");
	hipLaunchKernelGGL(bow_vector_kernel, dim3(n_kf), dim3(256), 0, ctx->stream, v->word_weight, d_n_rows, n_stride, kf_begin, max_feats,
	                   bow_word, bow_val, bow_n);
	SSK_LAUNCH_CHECK(ctx, "bow_vector_kernel");
	return SSK_OK;
}

// query keyframes [q_begin, q_begin + nq) of the (word, val, n) arrays against db; d_max_ids: device, per query, or NULL
int launch_query(ssk_ctx* ctx, ssk_bow_db* db, const int32_t* q_word, const double* q_val, const int32_t* q_n, int q_stride, int q_begin,
                 int nq, int max_results, const int32_t* d_max_ids, int r_stride, int32_t* r_ids, double* r_scores, int32_t* r_n)
{
	if (nq > db->q_cap) {
		if (db->scores) { SSK_HIP(ctx, hipStreamSynchronize(ctx->stream)); SSK_HIP(ctx, hipFree(db->scores)); db->scores = nullptr; }
		SSK_HIP(ctx, hipMalloc(&db->scores, (size_t)nq * db->cap * 8));
		db->q_cap = nq;
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_BOW);
	const int ne = db->n_entries;
	if (ne > 0) {
		// vocabularies beyond 512 Ki words (bitmap + prefix counts > 128 KB of LDS) binary-search the query instead;
		// SSK_BOW_NO_BITMAP=1 forces that path (tests)
		static const bool no_bitmap = std::getenv("SSK_BOW_NO_BITMAP") != nullptr;
		const int bm_dwords = (!no_bitmap && db->voc->n_words <= BOW_BITMAP_WORDS_MAX / 2) ? (db->voc->n_words + 31) / 32 : 0;
		const size_t lds = 12 * (size_t)BOW_SORT_MAX + 8 * (size_t)bm_dwords;
		static ssk_dev_once attr_once;      // (the attribute is per device)
		if (attr_once.need(ctx->device)) {
			SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(bow_score_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
			                                 12 * BOW_SORT_MAX + BOW_BITMAP_WORDS_MAX / 8));
			attr_once.done(ctx->device);
		}
		hipLaunchKernelGGL(bow_score_kernel, dim3(nq, (ne + 63) / 64), dim3(64), lds, ctx->stream, q_word, q_val, q_n, q_stride, q_begin, db->words,
		                   db->vals, db->ent_n, db->stride, db->cap, ne, d_max_ids, bm_dwords, db->scores, db->cap);
		SSK_LAUNCH_CHECK(ctx, "bow_score_kernel");
	}
	hipLaunchKernelGGL(bow_topk_kernel, dim3(nq), dim3(256), 0, ctx->stream, db->scores, db->cap, ne, max_results, r_stride, q_begin, r_ids,
	                   r_scores, r_n);
	SSK_LAUNCH_CHECK(ctx, "bow_topk_kernel");
	return SSK_OK;
}

}  // namespace

void ssk_kfbatch_bow_free(ssk_kfbatch* b)
{
	void* ptrs[] = {b->bow_word, b->bow_val, b->bow_n, b->q_ids, b->q_scores, b->q_n};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	b->bow_word = nullptr; b->bow_val = nullptr; b->bow_n = nullptr; b->q_ids = nullptr; b->q_scores = nullptr; b->q_n = nullptr;
}

extern "C" {

int ssk_bow_vocab_create(ssk_ctx* ctx, int k, int L, int weighting, int scoring, int n_rec, const int32_t* node_id,
                         const int32_t* parent_id, const double* weight, const uint8_t* desc, int n_words, const int32_t* word_node,
                         ssk_bow_vocab** out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!out || n_rec < 0 || n_words < 0 || (n_rec > 0 && (!node_id || !parent_id || !weight || !desc)) || (n_words > 0 && !word_node))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: bad argument");
	if (k < 1 || L < 1) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: k = %d, L = %d", k, L);
	if (weighting != 0 || scoring != 0)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: only weightingType 0 (TF_IDF) with scoringType 0 (L1_NORM) is "
		                     "implemented (the reference's vocabulary); got %d / %d", weighting, scoring);
	int max_id = 0;
	for (int i = 0; i < n_rec; ++i) {
		if (node_id[i] <= 0 || parent_id[i] < 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: record %d has nodeId %d, parentId %d", i, node_id[i], parent_id[i]);
		if (node_id[i] > max_id) max_id = node_id[i];
	}
	const int n = max_id + 1;
	std::vector<int32_t> first((size_t)n + 2, 0), child((size_t)n_rec + 1), rec_of((size_t)n, -1);
	for (int i = 0; i < n_rec; ++i) {
		if (parent_id[i] >= n) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: record %d has parentId %d outside the tree", i, parent_id[i]);
		if (rec_of[node_id[i]] >= 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: nodeId %d appears twice", node_id[i]);
		rec_of[node_id[i]] = i;
		first[parent_id[i] + 2]++;
	}
	for (int i = 0; i < n; ++i) first[i + 2] += first[i + 1];
	for (int i = 0; i < n_rec; ++i) child[first[parent_id[i] + 1]++] = node_id[i];    // file order inside a parent (W1)
	// breadth-first renumbering: children of a node contiguous, in file order
	std::vector<int32_t> order; order.reserve(n);                                    // new index -> node id
	std::vector<int32_t> new_of((size_t)n, -1);
	std::vector<int2> kids;
	order.push_back(0); new_of[0] = 0;
	for (size_t h = 0; h < order.size(); ++h) {
		const int nd = order[h];
		const int c0 = first[nd], c1 = first[nd + 1];
		kids.push_back(make_int2((int)order.size(), c1 - c0));
		for (int c = c0; c < c1; ++c) {
			if (new_of[child[c]] >= 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: the nodes do not form a tree (node %d)", child[c]);
			new_of[child[c]] = (int)order.size();
			order.push_back(child[c]);
		}
	}
	const int nn = (int)order.size();
	for (int i = 0; i < nn; ++i) if (kids[i].y >= (1 << 20)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: a node has %d children", kids[i].y);
	std::vector<uint8_t> nd((size_t)nn * 32, 0);
	std::vector<int32_t> leaf_word((size_t)nn, -1);
	std::vector<double> word_weight((size_t)n_words + 1, 0.0);
	for (int i = 1; i < nn; ++i) std::memcpy(nd.data() + 32 * (size_t)i, desc + 32 * (size_t)rec_of[order[i]], 32);
	for (int w = 0; w < n_words; ++w) {
		const int id = word_node[w];
		if (id <= 0 || id >= n || new_of[id] < 0 || rec_of[id] < 0)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_vocab_create: word %d names node %d, which is not a node of the tree", w, id);
		leaf_word[new_of[id]] = w;
		word_weight[w] = weight[rec_of[id]];
	}
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	ssk_bow_vocab* v = new ssk_bow_vocab();
	v->ctx = ctx; v->k = k; v->L = L; v->n_nodes = nn; v->n_words = n_words;
	hipError_t e = hipMalloc(&v->ndesc, (size_t)nn * 32);
	if (e == hipSuccess) e = hipMalloc(&v->kids, (size_t)nn * sizeof(int2));
	if (e == hipSuccess) e = hipMalloc(&v->leaf_word, (size_t)nn * 4);
	if (e == hipSuccess) e = hipMalloc(&v->word_weight, ((size_t)n_words + 1) * 8);
	if (e == hipSuccess) e = hipMemcpy(v->ndesc, nd.data(), (size_t)nn * 32, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(v->kids, kids.data(), (size_t)nn * sizeof(int2), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(v->leaf_word, leaf_word.data(), (size_t)nn * 4, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(v->word_weight, word_weight.data(), ((size_t)n_words + 1) * 8, hipMemcpyHostToDevice);
	if (e != hipSuccess) { ssk_bow_vocab_destroy(v); return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_bow_vocab_create: %s", hipGetErrorString(e)); }
	*out = v;
	return SSK_OK;
}

void ssk_bow_vocab_destroy(ssk_bow_vocab* v)
{
	if (!v) return;
	if (v->ctx) { (void)hipSetDevice(v->ctx->device); (void)hipStreamSynchronize(v->ctx->stream); }
	void* ptrs[] = {v->ndesc, v->kids, v->leaf_word, v->word_weight};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	delete v;
}

int ssk_bow_vocab_info(const ssk_bow_vocab* v, int* k, int* L, int* n_nodes, int* n_words)
{
	if (!v) return SSK_ERR_BAD_ARG;
	if (k) *k = v->k;
	if (L) *L = v->L;
	if (n_nodes) *n_nodes = v->n_nodes;
	if (n_words) *n_words = v->n_words;
	return SSK_OK;
}

int ssk_bow_transform(ssk_ctx* ctx, const ssk_bow_vocab* v, const uint8_t* desc, int n, int32_t* words, double* values, int* n_out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!v || n < 0 || (n > 0 && !desc) || !n_out || (n > 0 && (!words || !values)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_transform: bad argument");
	if (n > BOW_SORT_MAX) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_transform: %d descriptors, limit %d", n, BOW_SORT_MAX);
	*n_out = 0;
	if (n == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad((size_t)n * 32) + ssk_pad((size_t)n * 4) + ssk_pad((size_t)n * 8) + 4 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	uint8_t* d_desc = ssk_arena_take<uint8_t>(ctx->dev, (size_t)n * 32);
	int32_t* d_w = ssk_arena_take<int32_t>(ctx->dev, (size_t)n);
	double* d_v = ssk_arena_take<double>(ctx->dev, (size_t)n);
	int32_t* d_n = ssk_arena_take<int32_t>(ctx->dev, 2);
	const int32_t hn[2] = {n, 0};
	SSK_HIP(ctx, hipMemcpyAsync(d_desc, desc, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_n, hn, 8, hipMemcpyHostToDevice, ctx->stream));
	if ((rc = launch_transform(ctx, v, d_desc, d_n, 0, 0, 0, 1, n, d_w, d_v, d_n + 1))) return rc;
	int32_t nw = 0;
	SSK_HIP(ctx, hipMemcpyAsync(&nw, d_n + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	SSK_HIP(ctx, hipMemcpy(words, d_w, (size_t)nw * 4, hipMemcpyDeviceToHost));
	SSK_HIP(ctx, hipMemcpy(values, d_v, (size_t)nw * 8, hipMemcpyDeviceToHost));
	*n_out = nw;
	return SSK_OK;
}

int ssk_bow_transform_batch(ssk_ctx* ctx, const ssk_bow_vocab* v, ssk_kfbatch* b, int kf_begin, int kf_end)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!v || !b || kf_begin < 0 || kf_end > b->n_kf || kf_begin > kf_end) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_transform_batch: bad argument");
	if (kf_begin == kf_end) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	if ((rc = bow_alloc_batch(ctx, b))) return rc;
	// the left image of keyframe kf is image 2*kf: descriptor rows 2*kf*N ..., count nkp[2*kf] (all rows of
	// m_descriptors_left, BOW.h:73,86)
	return launch_transform(ctx, v, b->desc, b->nkp, 2, 2 * (size_t)b->max_feats, kf_begin, kf_end - kf_begin, b->max_feats, b->bow_word, b->bow_val, b->bow_n);
}

int ssk_kfbatch_bow_download(ssk_ctx* ctx, ssk_kfbatch* b, int kf, int32_t* words, double* values, int cap, int* n)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || kf < 0 || kf >= b->n_kf || !n || cap < 0) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_bow_download: bad argument");
	if (!b->bow_word) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_bow_download: no transform has run on this batch");
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	int32_t nw = 0;
	SSK_HIP(ctx, hipMemcpy(&nw, b->bow_n + kf, 4, hipMemcpyDeviceToHost));
	*n = nw;
	if (nw > cap) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_kfbatch_bow_download: %d words, capacity %d", nw, cap);
	if (nw > 0) {
		SSK_HIP(ctx, hipMemcpy(words, b->bow_word + (size_t)kf * b->max_feats, (size_t)nw * 4, hipMemcpyDeviceToHost));
		SSK_HIP(ctx, hipMemcpy(values, b->bow_val + (size_t)kf * b->max_feats, (size_t)nw * 8, hipMemcpyDeviceToHost));
	}
	return SSK_OK;
}

int ssk_bow_db_create(ssk_ctx* ctx, const ssk_bow_vocab* v, int capacity_entries, int max_words_per_entry, ssk_bow_db** out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!v || !out || capacity_entries < 1 || max_words_per_entry < 1 || max_words_per_entry > BOW_SORT_MAX)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_create: bad argument (entries >= 1, 1 <= words per entry <= %d)", BOW_SORT_MAX);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	ssk_bow_db* d = new ssk_bow_db();
	d->ctx = ctx; d->voc = v; d->cap = capacity_entries; d->stride = max_words_per_entry;
	hipError_t e = hipMalloc(&d->words, (size_t)d->cap * d->stride * 4);
	if (e == hipSuccess) e = hipMalloc(&d->vals, (size_t)d->cap * d->stride * 8);
	if (e == hipSuccess) e = hipMalloc(&d->ent_n, (size_t)d->cap * 4);
	if (e == hipSuccess) e = hipMemsetAsync(d->ent_n, 0, (size_t)d->cap * 4, ctx->stream);
	if (e != hipSuccess) { ssk_bow_db_destroy(d); return ssk_set_error(ctx, SSK_ERR_HIP, "ssk_bow_db_create: %s", hipGetErrorString(e)); }
	*out = d;
	return SSK_OK;
}

void ssk_bow_db_destroy(ssk_bow_db* d)
{
	if (!d) return;
	if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); }
	void* ptrs[] = {d->words, d->vals, d->ent_n, d->scores};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	delete d;
}

int ssk_bow_db_size(const ssk_bow_db* d) { return d ? d->n_entries : 0; }

int ssk_bow_db_clear(ssk_ctx* ctx, ssk_bow_db* d)
{
	if (!ctx || !d) return SSK_ERR_BAD_ARG;
	d->n_entries = 0;
	return SSK_OK;
}

int ssk_bow_db_add(ssk_ctx* ctx, ssk_bow_db* d, const int32_t* words, const double* values, int n, int* entry_id)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!d || n < 0 || (n > 0 && (!words || !values))) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_add: bad argument");
	if (n > d->stride) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_db_add: %d words, the database stores %d per entry", n, d->stride);
	if (d->n_entries >= d->cap) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_db_add: database full (%d entries)", d->cap);
	for (int i = 0; i < n; ++i)
		if (words[i] < 0 || words[i] >= d->voc->n_words || (i > 0 && words[i] <= words[i - 1]))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_add: word ids must be ascending and inside the vocabulary (index %d)", i);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const int e = d->n_entries;
	int rc;
	const size_t n1 = (size_t)(n > 0 ? n : 1);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad(n1 * 4) + ssk_pad(n1 * 8) + 2 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	int32_t* d_w = ssk_arena_take<int32_t>(ctx->dev, n1);
	double* d_v = ssk_arena_take<double>(ctx->dev, n1);
	if (n > 0) {
		SSK_HIP(ctx, hipMemcpyAsync(d_w, words, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
		SSK_HIP(ctx, hipMemcpyAsync(d_v, values, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
	}
	hipLaunchKernelGGL(bow_db_insert_kernel, dim3(1), dim3(256), 0, ctx->stream, d_w, d_v, (const int32_t*)nullptr, n, e, d->cap, d->stride, d->words, d->vals, d->ent_n);
	SSK_LAUNCH_CHECK(ctx, "bow_db_insert_kernel");
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the sources are the caller's (pageable) buffers
	d->n_entries = e + 1;
	if (entry_id) *entry_id = e;
	return SSK_OK;
}

int ssk_bow_db_add_from_batch(ssk_ctx* ctx, ssk_bow_db* d, ssk_kfbatch* b, int kf, int* entry_id)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!d || !b || kf < 0 || kf >= b->n_kf) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_add_from_batch: bad argument");
	if (!b->bow_word) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_add_from_batch: no transform has run on this batch");
	if (b->max_feats > d->stride) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_db_add_from_batch: batch holds %d words per keyframe, the database %d", b->max_feats, d->stride);
	if (d->n_entries >= d->cap) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_db_add_from_batch: database full (%d entries)", d->cap);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const int e = d->n_entries;
	const size_t N = (size_t)b->max_feats;
	hipLaunchKernelGGL(bow_db_insert_kernel, dim3(1), dim3(256), 0, ctx->stream, b->bow_word + kf * N, b->bow_val + kf * N, b->bow_n + kf, 0, e, d->cap,
	                   d->stride, d->words, d->vals, d->ent_n);
	SSK_LAUNCH_CHECK(ctx, "bow_db_insert_kernel");
	d->n_entries = e + 1;
	if (entry_id) *entry_id = e;
	return SSK_OK;
}

int ssk_bow_db_query(ssk_ctx* ctx, ssk_bow_db* d, const int32_t* words, const double* values, int n, int max_results, int max_id,
                     int32_t* ids, double* scores, int* n_results)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!d || n < 0 || (n > 0 && (!words || !values)) || !ids || !scores || !n_results) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query: bad argument");
	if (max_results < 1 || max_results > SSK_BOW_MAX_RESULTS)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query: max_results must be in [1, %d] (the reference asks for 4, EST.cpp:426)", SSK_BOW_MAX_RESULTS);
	if (n > BOW_SORT_MAX) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_bow_db_query: %d words, limit %d", n, BOW_SORT_MAX);
	for (int i = 0; i < n; ++i)
		if (words[i] < 0 || words[i] >= d->voc->n_words || (i > 0 && words[i] <= words[i - 1]))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query: word ids must be ascending and inside the vocabulary (index %d)", i);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int rc;
	const size_t n1 = (size_t)(n > 0 ? n : 1);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad(n1 * 4) + ssk_pad(n1 * 8) + ssk_pad(SSK_BOW_MAX_RESULTS * 12) + 6 * 256))) return rc;
	ssk_arena_reset(ctx->dev);
	int32_t* d_w = ssk_arena_take<int32_t>(ctx->dev, n1);
	double* d_v = ssk_arena_take<double>(ctx->dev, n1);
	int32_t* d_n = ssk_arena_take<int32_t>(ctx->dev, 3);           // [n words, max_id, n results]
	int32_t* d_ids = ssk_arena_take<int32_t>(ctx->dev, SSK_BOW_MAX_RESULTS);
	double* d_sc = ssk_arena_take<double>(ctx->dev, SSK_BOW_MAX_RESULTS);
	const int32_t hn[3] = {n, max_id, 0};
	if (n > 0) {
		SSK_HIP(ctx, hipMemcpyAsync(d_w, words, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
		SSK_HIP(ctx, hipMemcpyAsync(d_v, values, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
	}
	SSK_HIP(ctx, hipMemcpyAsync(d_n, hn, 12, hipMemcpyHostToDevice, ctx->stream));
	if ((rc = launch_query(ctx, d, d_w, d_v, d_n, BOW_SORT_MAX, 0, 1, max_results, d_n + 1, SSK_BOW_MAX_RESULTS, d_ids, d_sc, d_n + 2))) return rc;
	int32_t nr = 0;
	SSK_HIP(ctx, hipMemcpyAsync(&nr, d_n + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (nr > 0) {
		SSK_HIP(ctx, hipMemcpy(ids, d_ids, (size_t)nr * 4, hipMemcpyDeviceToHost));
		SSK_HIP(ctx, hipMemcpy(scores, d_sc, (size_t)nr * 8, hipMemcpyDeviceToHost));
	}
	*n_results = nr;
	return SSK_OK;
}

int ssk_bow_db_query_batch(ssk_ctx* ctx, ssk_bow_db* d, ssk_kfbatch* b, int kf_begin, int kf_end, int max_results, const int32_t* max_ids)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!d || !b || kf_begin < 0 || kf_end > b->n_kf || kf_begin > kf_end) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query_batch: bad argument");
	if (!b->bow_word) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query_batch: no transform has run on this batch");
	if (max_results < 1 || max_results > SSK_BOW_MAX_RESULTS)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_bow_db_query_batch: max_results must be in [1, %d]", SSK_BOW_MAX_RESULTS);
	const int nq = kf_end - kf_begin;
	if (nq == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	int32_t* d_max = nullptr;
	if (max_ids) {
		int rc;
		if ((rc = ssk_arena_reserve(ctx, ctx->dev, ssk_pad((size_t)nq * 4) + 256))) return rc;
		ssk_arena_reset(ctx->dev);
		d_max = ssk_arena_take<int32_t>(ctx->dev, (size_t)nq);
		SSK_HIP(ctx, hipMemcpyAsync(d_max, max_ids, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
	}
	return launch_query(ctx, d, b->bow_word, b->bow_val, b->bow_n, b->max_feats, kf_begin, nq, max_results, d_max, SSK_BOW_MAX_RESULTS,
	                    b->q_ids, b->q_scores, b->q_n);
}

int ssk_kfbatch_bow_results(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, int32_t* ids, double* scores, int32_t* n_results)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!b || kf_begin < 0 || kf_end > b->n_kf || kf_begin > kf_end || !ids || !scores || !n_results)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_bow_results: bad argument");
	if (!b->q_ids) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_kfbatch_bow_results: no query has run on this batch");
	const size_t nq = (size_t)(kf_end - kf_begin);
	if (nq == 0) return SSK_OK;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	SSK_HIP(ctx, hipMemcpy(ids, b->q_ids + (size_t)kf_begin * SSK_BOW_MAX_RESULTS, nq * SSK_BOW_MAX_RESULTS * 4, hipMemcpyDeviceToHost));
	SSK_HIP(ctx, hipMemcpy(scores, b->q_scores + (size_t)kf_begin * SSK_BOW_MAX_RESULTS, nq * SSK_BOW_MAX_RESULTS * 8, hipMemcpyDeviceToHost));
	SSK_HIP(ctx, hipMemcpy(n_results, b->q_n + kf_begin, nq * 4, hipMemcpyDeviceToHost));
	return SSK_OK;
}

}  // extern "C"
