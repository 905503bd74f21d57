// ssk_select.hip -- keypoint selection for one image per workgroup: retain the best 2N FAST candidates,
// Harris response, optional window suppression, retain the best N in the total order
// (response desc, y asc, x asc).  Contract C5 of DESIGN.md "Front-end contract"; replaces the
// retainBest / HarrisResponses / retainBest steps of cv::ORB (reached from EST.cpp:112,267 through
// stereo-vo) plus stereo-vo's keypoint non-max suppression (KITTI.ini:12-13).
//
// Mapping (gfx950): one 512-thread workgroup per image (a batch of B keyframes gives 2B workgroups).
//   1. exact top-K of the packed 32-bit candidate keys by 4-pass MSD radix select (8 bits per pass,
//      LDS histogram with ds atomics; keys are unique, so "key >= K-th key" selects exactly K).
//   2. one candidate per thread: Harris response from the 9x9 pixel window (27 aligned dword loads, integer sums,
//      then the IEEE float sequence of the contract).
//   3. window suppression: bitonic sort by position, then every candidate tests only the two runs of the sorted
//      list that lie within r rows of it; suppressed keys set to 0.
//   4. bitonic sort (descending) of <= 4096 64-bit keys  orderable(response)<<32 | position  in LDS; the first
//      min(N, survivors) are written as keypoints (x, y, response).
// Data touched: the image's candidate list (4 B each, L2/MALL resident, written by the scan kernel) and
// 81 px per retained candidate; latency/VALU bound, not a bandwidth kernel.
#include "ssk_frontend.h"

namespace {

#ifndef SSK_SEL_NT
#define SSK_SEL_NT 512                  // (per 2 048 images: 0.285 ms at 1 024 threads, 0.234 at 512, 0.281 at 256)
#endif
constexpr int NT = SSK_SEL_NT;
constexpr int MAXK = 2 * SSK_MAX_FEATS;   // 4096

__device__ __forceinline__ uint32_t float_key(float f)
{
	const uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
	return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// The 9x9 pixel window comes in as 27 aligned dwords (3 per row, all loads in flight at once) instead of ~400
// scattered byte loads -- the byte version spent the kernel in the texture addresser; rows are realigned with
// v_alignbyte and the 49 gradient pairs are pure register arithmetic (integers: same sums, bit for bit).
__device__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x, int y, float k)
{
	const int xs = x - 4;
	const uint32_t sh = (uint32_t)(xs & 3);
	const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>(img + (size_t)(y - 4) * pitch + (xs & ~3));
	const int pw = pitch >> 2;
	uint32_t d[9][3];
#pragma unroll
	for (int r = 0; r < 9; ++r)
#pragma unroll
		for (int c = 0; c < 3; ++c) d[r][c] = base[r * pw + c];
	int px[9][9];
#pragma unroll
	for (int r = 0; r < 9; ++r) {
		const uint32_t w0 = __builtin_amdgcn_alignbyte(d[r][1], d[r][0], sh), w1 = __builtin_amdgcn_alignbyte(d[r][2], d[r][1], sh);
		const uint32_t w2 = d[r][2] >> (8 * sh);
#pragma unroll
		for (int c = 0; c < 4; ++c) { px[r][c] = (int)((w0 >> (8 * c)) & 255u); px[r][4 + c] = (int)((w1 >> (8 * c)) & 255u); }
		px[r][8] = (int)(w2 & 255u);
	}
	int a = 0, b = 0, c = 0;
#pragma unroll
	for (int cy = 1; cy <= 7; ++cy)
#pragma unroll
		for (int cx = 1; cx <= 7; ++cx) {
			const int Ix = (px[cy][cx + 1] - px[cy][cx - 1]) * 2 + (px[cy - 1][cx + 1] - px[cy - 1][cx - 1]) + (px[cy + 1][cx + 1] - px[cy + 1][cx - 1]);
			const int Iy = (px[cy + 1][cx] - px[cy - 1][cx]) * 2 + (px[cy + 1][cx - 1] - px[cy - 1][cx - 1]) + (px[cy + 1][cx + 1] - px[cy - 1][cx + 1]);
			a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
		}
	const float scale = 1.0f / (4.0f * 7.0f * 255.0f);
	// scale_sq_sq = scale * scale * scale * scale, left to right as cv::ORB's HarrisResponses writes it (not (s*s)*(s*s): 1 ulp apart)
	const float s4 = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
	const float fa = (float)a, fb = (float)b, fc = (float)c;
	const float t1 = __fmul_rn(fa, fb);
	const float t2 = __fmul_rn(fc, fc);
	const float t3 = __fadd_rn(fa, fb);
	const float t4 = __fmul_rn(k, t3);
	const float t5 = __fmul_rn(t4, t3);
	return __fmul_rn(__fsub_rn(__fsub_rn(t1, t2), t5), s4);
}

// in-place bitonic sort of P (power of two) 64-bit keys in LDS, descending, by the whole workgroup.  Thread t owns
// the pair (lo, lo + stride): for stride <= 64 the 64 pairs of a wave live in one 128-key span that no other wave
// touches, so those steps only need the wave's own LDS ordering; a workgroup barrier separates steps only when
// one of the two has stride > 64 (6 of the 55 steps at P = 1024).
// P == 2 NT (1 024 keys, the usual case: K = 2 x 500 candidates): thread t keeps elements 2 t and 2 t + 1 in REGISTERS through the whole network.
// Stride 1 is inside the thread, strides 2 .. 64 meet the partner lane of the same wavefront by a lane-xor shuffle (39 of the 55 steps: no LDS
// round trip, no fence), strides 128 .. 512 (6 steps) go through LDS between two workgroup barriers.  (The LDS form below spent ~520 cycles a step
// on a read - compare - write - fence chain: 28.5 k cycles per sort, two sorts = half the kernel.)
__device__ __forceinline__ void cmpx(unsigned long long& mine, unsigned long long other, bool keep_max)
{
	const bool take = keep_max ? (other > mine) : (other < mine);
	if (take) mine = other;
}
__device__ void bitonic_desc_regs(unsigned long long* s_key, int tid)
{
	static_assert(NT >= 128, "two wavefronts or more");
	constexpr int P = 2 * NT;
	const int e0 = 2 * tid;
	unsigned long long a = s_key[e0], b = s_key[e0 + 1];
	for (int size = 2; size <= P; size <<= 1) {
		const bool desc = (e0 & size) == 0;                      // (both elements: size >= 2)
		for (int stride = size >> 1; stride >= 128; stride >>= 1) {
			// partner thread in another wavefront: through LDS
			__syncthreads();
			s_key[e0] = a; s_key[e0 + 1] = b;
			__syncthreads();
			const bool lower = (e0 & stride) == 0;                // this thread's elements are the lower ones of their pairs
			const unsigned long long oa = s_key[e0 ^ stride], ob = s_key[(e0 ^ stride) + 1];
			cmpx(a, oa, lower == desc); cmpx(b, ob, lower == desc);
		}
		for (int stride = min(size >> 1, 64); stride >= 2; stride >>= 1) {
			const bool lower = (e0 & stride) == 0;
			const unsigned long long oa = __shfl_xor(a, stride >> 1, 64), ob = __shfl_xor(b, stride >> 1, 64);
			cmpx(a, oa, lower == desc); cmpx(b, ob, lower == desc);
		}
		// stride 1: the thread's own pair (descending: the larger one first)
		if ((a < b) == desc) { const unsigned long long t = a; a = b; b = t; }
	}
	__syncthreads();
	s_key[e0] = a; s_key[e0 + 1] = b;
	__syncthreads();
}

__device__ void bitonic_desc(unsigned long long* s_key, int P, int tid)
{
	if (P == 2 * NT) { bitonic_desc_regs(s_key, tid); return; }   // (uniform)
	for (int size = 2; size <= P; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = tid; t < P / 2; t += NT) {
				const int lo = 2 * t - (t & (stride - 1));
				const int hi = lo + stride;
				const bool desc = (lo & size) == 0;
				const unsigned long long a = s_key[lo], b = s_key[hi];
				if ((a < b) == desc) { s_key[lo] = b; s_key[hi] = a; }
			}
			const int next = stride > 1 ? stride >> 1 : size;      // stride of the step that follows (size = first stride of 2*size)
			if (stride > 64 || next > 64 || P / 2 > NT) __syncthreads();
			else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
		}
	}
	__syncthreads();
}

__global__ __launch_bounds__(NT)
void select_kernel(const uint8_t* __restrict__ images, int h, int pitch, int img_begin,
                   const uint32_t* __restrict__ cand, int cand_cap, const int32_t* __restrict__ ncand,
                   int n_feats, int nms_radius, float harris_k, int max_feats,
                   short2* __restrict__ kp_xy, float* __restrict__ kp_resp, int32_t* __restrict__ nkp)
{
	__shared__ unsigned long long s_key[MAXK];
	__shared__ int s_hist[256];
	__shared__ uint32_t s_prefix;
	__shared__ int s_krem, s_n;

	const int img = img_begin + blockIdx.x;
	const int tid = threadIdx.x;
#ifdef SSK_SELECT_STAMPS
	__shared__ unsigned long long s_t[8];
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
#define SSTAMP(i) do { __syncthreads(); if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_t[i] = t_ - t_last; t_last = t_; } } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif
	const uint32_t* __restrict__ keys = cand + (size_t)img * cand_cap;
	const int n = min(ncand[img], cand_cap);
	const int K = min(2 * n_feats, n);
	const uint8_t* __restrict__ src = images + (size_t)img * h * pitch;

	// ---- 1. K-th largest key (exact, keys unique).  A thread's first KREG keys stay in registers for all passes
	//      (candidate lists are a few thousand keys: one trip to L2 instead of five) ----
	constexpr int KREG = 8;
	uint32_t kreg[KREG];
#pragma unroll
	for (int u = 0; u < KREG; ++u) kreg[u] = tid + u * NT < n ? keys[tid + u * NT] : 0u;   // valid keys are never 0 (score >= 1 in the top byte)
	uint32_t kth = 0;
	if (n > K) {
		if (tid == 0) { s_prefix = 0; s_krem = K; }
		for (int pass = 0; pass < 4; ++pass) {
			const int shift = 24 - 8 * pass;
			if (tid < 256) s_hist[tid] = 0;
			__syncthreads();
			const uint32_t prefix = s_prefix;
#pragma unroll
			for (int u = 0; u < KREG; ++u) {
				const uint32_t k = kreg[u];
				if (k != 0u && (pass == 0 || (k >> (shift + 8)) == prefix)) atomicAdd(&s_hist[(k >> shift) & 255], 1);
			}
			for (int i = tid + KREG * NT; i < n; i += NT) {
				const uint32_t k = keys[i];
				if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
			}
			__syncthreads();
			// digit of the K-th key: the bin d with  sum(bins > d) < rem <= sum(bins >= d)  (d = 0 if none above): one wave,
			// 4 bins per lane, suffix sums by shuffles (a serial walk over 256 LDS bins cost more than the histogram)
			if (tid < 64) {
				const int rem = s_krem;
				const int h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
				const int tot = h0 + h1 + h2 + h3;
				int suf = tot;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_down(suf, o, 64); if (tid + o < 64) suf += v; }
				const int a3 = suf - tot, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1, s0 = a0 + h0;   // a_k = sum of the bins above bin 4*tid+k
				int d = -1, above = 0;
				if (a3 < rem && a2 >= rem) { d = 4 * tid + 3; above = a3; }
				else if (a2 < rem && a1 >= rem) { d = 4 * tid + 2; above = a2; }
				else if (a1 < rem && a0 >= rem) { d = 4 * tid + 1; above = a1; }
				else if (a0 < rem && (s0 >= rem || tid == 0)) { d = 4 * tid; above = a0; }
				if (d >= 0) { s_prefix = (prefix << 8) | (uint32_t)d; s_krem = rem - above; }
			}
			__syncthreads();
		}
		kth = s_prefix;
	}
	SSTAMP(0);
	if (tid == 0) s_n = 0;
	__syncthreads();
#pragma unroll
	for (int u = 0; u < KREG; ++u) {
		const uint32_t k = kreg[u];
		if (k != 0u && k >= kth) {
			const int slot = atomicAdd(&s_n, 1);
			if (slot < MAXK) s_key[slot] = k;    // exactly K entries; guard only against corrupt input
		}
	}
	for (int i = tid + KREG * NT; i < n; i += NT) {
		const uint32_t k = keys[i];
		if (k >= kth) {
			const int slot = atomicAdd(&s_n, 1);
			if (slot < MAXK) s_key[slot] = k;
		}
	}
	__syncthreads();

	SSTAMP(1);
	// ---- 2. Harris response; 64-bit key  position(24) << 32 | orderable(response) ----
	int P = 1;
	while (P < K) P <<= 1;
	for (int i = tid; i < P; i += NT) {
		if (i < K) {
			const uint32_t k = (uint32_t)s_key[i];
			const int x = 4095 - (int)(k & 0xFFF), y = 4095 - (int)((k >> 12) & 0xFFF);
			const float r = harris_response(src, pitch, x, y, harris_k);
			s_key[i] = ((unsigned long long)(k & 0xFFFFFFu) << 32) | (unsigned long long)float_key(r);
		} else {
			s_key[i] = 0ull;
		}
	}
	__syncthreads();
	SSTAMP(2);

	// ---- 3. window suppression (strictly greater (response, position) key within |dx|,|dy| <= r wins): the keys are
	//      first sorted by position (row-major), so the candidates within r rows of a candidate are two short runs
	//      next to it -- ~K * (2r+1) / rows tests per candidate instead of K ----
	if (nms_radius > 0) {
		bitonic_desc(s_key, P, tid);
		unsigned long long mine[MAXK / NT];
		bool kill[MAXK / NT];
#pragma unroll
		for (int s = 0; s < MAXK / NT; ++s) {
			const int i = tid + s * NT;
			kill[s] = false;
			mine[s] = 0ull;
			if (i < K) {
				const unsigned long long me = s_key[i];
				const int my = (int)(me >> 44) & 0xFFF, mx = (int)(me >> 32) & 0xFFF;
				const unsigned long long mk = (me << 32) | (me >> 32);
				mine[s] = mk;
				for (int j = i - 1; j >= 0 && !kill[s]; --j) {          // smaller rows (larger 4095 - y)
					const unsigned long long o = s_key[j];
					if (((int)(o >> 44) & 0xFFF) - my > nms_radius) break;
					kill[s] = abs(((int)(o >> 32) & 0xFFF) - mx) <= nms_radius && ((o << 32) | (o >> 32)) > mk;
				}
				for (int j = i + 1; j < K && !kill[s]; ++j) {
					const unsigned long long o = s_key[j];
					if (my - ((int)(o >> 44) & 0xFFF) > nms_radius) break;
					kill[s] = abs(((int)(o >> 32) & 0xFFF) - mx) <= nms_radius && ((o << 32) | (o >> 32)) > mk;
				}
			}
		}
		__syncthreads();
#pragma unroll
		for (int s = 0; s < MAXK / NT; ++s) {
			const int i = tid + s * NT;
			if (i < K) s_key[i] = kill[s] ? 0ull : mine[s];
		}
		__syncthreads();
	} else {
		for (int i = tid; i < K; i += NT) { const unsigned long long me = s_key[i]; s_key[i] = (me << 32) | (me >> 32); }
		__syncthreads();
	}
	SSTAMP(3);

	// ---- 4. sort by (response, position) descending; x,y travel inside the key ----
	bitonic_desc(s_key, P, tid);
	SSTAMP(4);
	// ---- output ----
	const int cap = min(n_feats, max_feats);
	short2* oxy = kp_xy + (size_t)img * max_feats;
	float* oresp = kp_resp + (size_t)img * max_feats;
	int cnt = 0;
	for (int i = tid; i < min(cap, P); i += NT) {
		const unsigned long long k = s_key[i];
		if (k != 0ull) {
			const uint32_t lo = (uint32_t)(k & 0xFFFFFFu);
			oxy[i] = make_short2((short)(4095 - (int)(lo & 0xFFF)), (short)(4095 - (int)((lo >> 12) & 0xFFF)));
			oresp[i] = key_float((uint32_t)(k >> 32));
			++cnt;
		}
	}
	// survivors are a prefix of the sorted array: count them
	__syncthreads();
	if (tid == 0) s_n = 0;
	__syncthreads();
	if (cnt) atomicAdd(&s_n, cnt);
	__syncthreads();
	if (tid == 0) nkp[img] = s_n;
#ifdef SSK_SELECT_STAMPS
	SSTAMP(5);
	if (tid == 0 && blockIdx.x == 0) printf("select stamps (cycles): n %d K %d | radix %llu compact %llu harris %llu nms %llu sort %llu out %llu\n", n, K, s_t[0], s_t[1], s_t[2], s_t[3], s_t[4], s_t[5]);
#endif
}

}  // namespace

int ssk_launch_select(ssk_ctx* ctx, ssk_kfbatch* b, int img_begin, int img_end, const ssk_frontend_params& p)
{
	const int nimg = img_end - img_begin;
	if (nimg <= 0) return SSK_OK;
	ssk_stage_scope prof(ctx, SSK_STAGE_SELECT);
	hipLaunchKernelGGL(select_kernel, dim3(nimg), dim3(NT), 0, ctx->stream, b->img, b->h, b->pitch, img_begin,
	                   b->cand, b->cand_cap, b->ncand, p.n_feats, p.nms_radius, p.harris_k, b->max_feats,
	                   b->kp_xy, b->kp_resp, b->nkp);
	SSK_LAUNCH_CHECK(ctx, "select_kernel");
	return SSK_OK;
}
