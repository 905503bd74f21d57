// ssk_select.hip -- keypoint selection for one image per workgroup: retain the best 2N FAST candidates,
// Harris response, optional window suppression, retain the best N in the total order
// (response desc, y asc, x asc).  Contract C5 of DESIGN.md "Front-end contract"; replaces the
// retainBest / HarrisResponses / retainBest steps of cv::ORB (reached from EST.cpp:112,267 through
// stereo-vo) plus stereo-vo's keypoint non-max suppression (KITTI.ini:12-13).
//
// Mapping (gfx950): one 1024-thread workgroup per image (a batch of B keyframes gives 2B workgroups).
//   1. exact top-K of the packed 32-bit candidate keys by 4-pass MSD radix select (8 bits per pass,
//      LDS histogram with ds atomics; keys are unique, so "key >= K-th key" selects exactly K).
//   2. one candidate per thread: Harris response from the 9x9 pixel window (integer sums, then the IEEE
//      float sequence of the contract), 64-bit sort key  orderable(response)<<32 | low 24 key bits.
//   3. window suppression: O(K^2) compare in LDS, suppressed keys set to 0.
//   4. bitonic sort (descending) of <= 4096 64-bit keys in LDS; the first min(N, survivors) are written
//      as keypoints (x, y, response).
// Data touched: the image's candidate list (4 B each, L2/MALL resident, written by the scan kernel) and
// 81 px per retained candidate; latency/VALU bound, not a bandwidth kernel.
#include "ssk_frontend.h"

namespace {

constexpr int NT = 1024;
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

__device__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x, int y, float k)
{
	int a = 0, b = 0, c = 0;
	// 9x9 window rows; gradients by the 3x3 Sobel-like stencil of the contract
	for (int dy = -3; dy <= 3; ++dy) {
		const uint8_t* r0 = img + (size_t)(y + dy - 1) * pitch + x;
		const uint8_t* r1 = r0 + pitch;
		const uint8_t* r2 = r1 + pitch;
#pragma unroll
		for (int dx = -3; dx <= 3; ++dx) {
			const int p00 = r0[dx - 1], p01 = r0[dx], p02 = r0[dx + 1];
			const int p10 = r1[dx - 1], p12 = r1[dx + 1];
			const int p20 = r2[dx - 1], p21 = r2[dx], p22 = r2[dx + 1];
			const int Ix = (p12 - p10) * 2 + (p02 - p00) + (p22 - p20);
			const int Iy = (p21 - p01) * 2 + (p20 - p00) + (p22 - p02);
			a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
		}
	}
	const float scale = 1.0f / (4.0f * 7.0f * 255.0f);
	const float s2 = __fmul_rn(scale, scale);
	const float s4 = __fmul_rn(s2, s2);
	const float fa = (float)a, fb = (float)b, fc = (float)c;
	const float t1 = __fmul_rn(fa, fb);
	const float t2 = __fmul_rn(fc, fc);
	const float t3 = __fadd_rn(fa, fb);
	const float t4 = __fmul_rn(k, t3);
	const float t5 = __fmul_rn(t4, t3);
	return __fmul_rn(__fsub_rn(__fsub_rn(t1, t2), t5), s4);
}

__global__ __launch_bounds__(NT)
void select_kernel(const uint8_t* __restrict__ images, int h, int pitch, int img_begin,
                   const uint32_t* __restrict__ cand, int cand_cap, const int32_t* __restrict__ ncand,
                   int n_feats, int nms_radius, float harris_k, int max_feats,
                   short2* __restrict__ kp_xy, float* __restrict__ kp_resp, int32_t* __restrict__ nkp)
{
	__shared__ unsigned long long s_key[MAXK];
	__shared__ short2 s_xy[MAXK];
	__shared__ int s_hist[256];
	__shared__ uint32_t s_prefix;
	__shared__ int s_krem, s_n;

	const int img = img_begin + blockIdx.x;
	const int tid = threadIdx.x;
	const uint32_t* __restrict__ keys = cand + (size_t)img * cand_cap;
	const int n = min(ncand[img], cand_cap);
	const int K = min(2 * n_feats, n);
	const uint8_t* __restrict__ src = images + (size_t)img * h * pitch;

	// ---- 1. K-th largest key (exact, keys unique) ----
	uint32_t kth = 0;
	if (n > K) {
		if (tid == 0) { s_prefix = 0; s_krem = K; }
		for (int pass = 0; pass < 4; ++pass) {
			const int shift = 24 - 8 * pass;
			if (tid < 256) s_hist[tid] = 0;
			__syncthreads();
			const uint32_t prefix = s_prefix;
			for (int i = tid; i < n; i += NT) {
				const uint32_t k = keys[i];
				if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
			}
			__syncthreads();
			if (tid == 0) {
				int rem = s_krem, d = 255;
				for (; d > 0; --d) { if (s_hist[d] >= rem) break; rem -= s_hist[d]; }
				s_prefix = (prefix << 8) | (uint32_t)d;
				s_krem = rem;
			}
			__syncthreads();
		}
		kth = s_prefix;
	}
	if (tid == 0) s_n = 0;
	__syncthreads();
	for (int i = tid; i < n; i += NT) {
		const uint32_t k = keys[i];
		if (k >= kth) {
			const int slot = atomicAdd(&s_n, 1);
			if (slot < MAXK) s_key[slot] = k;    // exactly K entries; guard only against corrupt input
		}
	}
	__syncthreads();

	// ---- 2. Harris response -> 64-bit key ----
	int P = 1;
	while (P < K) P <<= 1;
	for (int i = tid; i < P; i += NT) {
		if (i < K) {
			const uint32_t k = (uint32_t)s_key[i];
			const int x = 4095 - (int)(k & 0xFFF), y = 4095 - (int)((k >> 12) & 0xFFF);
			const float r = harris_response(src, pitch, x, y, harris_k);
			s_xy[i] = make_short2((short)x, (short)y);
			s_key[i] = ((unsigned long long)float_key(r) << 32) | (unsigned long long)(k & 0xFFFFFFu);
		} else {
			s_key[i] = 0ull;
			s_xy[i] = make_short2(-30000, -30000);
		}
	}
	__syncthreads();

	// ---- 3. window suppression (strictly greater key within |dx|,|dy| <= r wins) ----
	if (nms_radius > 0) {
		bool kill[MAXK / NT];
#pragma unroll
		for (int s = 0; s < MAXK / NT; ++s) {
			const int i = tid + s * NT;
			kill[s] = false;
			if (i < K) {
				const short2 me = s_xy[i];
				const unsigned long long mk = s_key[i];
				for (int j = 0; j < K; ++j) {
					const short2 o = s_xy[j];
					const int dx = abs((int)o.x - (int)me.x), dy = abs((int)o.y - (int)me.y);
					if (dx <= nms_radius && dy <= nms_radius && s_key[j] > mk) { kill[s] = true; break; }
				}
			}
		}
		__syncthreads();
#pragma unroll
		for (int s = 0; s < MAXK / NT; ++s) {
			const int i = tid + s * NT;
			if (i < K && kill[s]) s_key[i] = 0ull;
		}
		__syncthreads();
	}

	// ---- 4. bitonic sort, descending, P elements (x,y travel inside the key) ----
	for (int size = 2; size <= P; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = tid; t < P / 2; t += NT) {
				const int lo = 2 * t - (t & (stride - 1));
				const int hi = lo + stride;
				const bool desc = (lo & size) == 0;
				const unsigned long long a = s_key[lo], b = s_key[hi];
				if ((a < b) == desc) { s_key[lo] = b; s_key[hi] = a; }
			}
			__syncthreads();
		}
	}

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
