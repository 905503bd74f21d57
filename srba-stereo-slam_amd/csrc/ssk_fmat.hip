// ssk_fmat.hip -- fundamental-matrix RANSAC outlier filter of the data association: replaces
// m_detect_outliers_with_F (EST.cpp:2015-2055) including cv::findFundamentalMat(..., FM_RANSAC, ...) (call :2044).
// Contract F1-F6 of DESIGN.md "Fundamental-matrix contract" (oracle/oracle_fmat.c); the arithmetic below is the
// oracle's operation sequence in IEEE double without contraction (this file keeps the build's -ffp-contract=off),
// so models, inlier sets and verdicts are bit-identical to it.
//
// Mapping (gfx950): one workgroup of SSK_FM_NT threads (64: one wavefront) per keyframe pair.  The tracked pairs are compacted in order into LDS.
// RANSAC is evaluated HR = 8 hypotheses at a time: thread t handles hypothesis t % HR and the part t / HR of the points (32
// parts) -- the threads of a hypothesis repeat its 7-point solver (the counter-based generator makes hypotheses
// independent: Gauss-Jordan null space in registers with static indexing, cubic by bisection, up to 3 models) and
// count the models' inliers over their part (integer LDS adds).  One thread then walks the hypotheses in index order
// with OpenCV's adaptive stop, so the parallel evaluation yields the sequential result whatever HR is.  Typical keyframe
// pairs (90 % inliers) stop after ~5 hypotheses: one round; measured 0.19 / 0.157 / 0.148 / 0.171 ms per 256 pairs at
// HR = 64 / 16 / 8 / 4.
#include "ssk_frontend.h"
#include <vector>

namespace {

#ifndef SSK_FM_NT
#define SSK_FM_NT 64                 // round 5: one wavefront per keyframe pair (a hypothesis x an eighth of the points per lane): 0.227 ms per 1 024 pairs against
#endif                             // 0.247 at 256 threads, 0.232 at 128 (same box, tools/stage_ab.sh) -- a latency kernel at 176 VGPRs is sized by the workgroups a CU holds
constexpr int NT = SSK_FM_NT;
#ifndef SSK_FM_HR
#define SSK_FM_HR 8
#endif
constexpr int HR = SSK_FM_HR;             // hypotheses per round (a power of two <= 64)
constexpr int MAXM = SSK_MAX_FEATS;

__device__ __forceinline__ uint32_t pcg_next(uint64_t& st)
{
	const uint64_t old = st;
	st = old * 6364136223846793005ULL + 1442695040888963407ULL;
	const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	const uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

__device__ __forceinline__ void fm_sample(uint64_t seed, int h, int n, int (&idx)[7])
{
	uint64_t st = seed + (uint64_t)h * 0x9E3779B97F4A7C15ULL;
	(void)pcg_next(st);
#pragma unroll
	for (int k = 0; k < 7; ++k) {
		for (;;) {
			const int v = (int)(pcg_next(st) % (uint32_t)n);
			bool dup = false;
#pragma unroll
			for (int j = 0; j < 7; ++j) dup |= j < k && idx[j] == v;
			if (!dup) { idx[k] = v; break; }
		}
	}
}

__device__ __forceinline__ double det3(const double* M)
{
	return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

__device__ __forceinline__ double cubic_eval(double a, double b, double c, double x) { return ((x + a) * x + b) * x + c; }

__device__ __forceinline__ double bisect(double a, double b, double c, double lo, double hi)
{
	double flo = cubic_eval(a, b, c, lo);
	for (int it = 0; it < 80; ++it) {
		const double mid = 0.5 * (lo + hi);
		const double fm = cubic_eval(a, b, c, mid);
		if ((flo <= 0.0) == (fm <= 0.0)) { lo = mid; flo = fm; } else hi = mid;
	}
	return 0.5 * (lo + hi);
}

__device__ __forceinline__ int cubic_roots(double c3, double c2, double c1, double c0, double (&roots)[3])
{
	const double m = fmax(fmax(fabs(c3), fabs(c2)), fmax(fabs(c1), fabs(c0)));
	if (!(m > 0.0)) return 0;
	if (fabs(c3) <= 1e-12 * m) {
		if (fabs(c2) <= 1e-12 * m) {
			if (fabs(c1) <= 1e-12 * m) return 0;
			roots[0] = -c0 / c1;
			return 1;
		}
		const double disc = c1 * c1 - 4.0 * c2 * c0;
		if (disc < 0.0) return 0;
		const double sq = sqrt(disc);
		const double r0 = (-c1 - sq) / (2.0 * c2), r1 = (-c1 + sq) / (2.0 * c2);
		roots[0] = fmin(r0, r1); roots[1] = fmax(r0, r1);
		return 2;
	}
	const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
	const double R = 1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)));
	const double disc = a * a - 3.0 * b;
	if (!(disc > 0.0)) { roots[0] = bisect(a, b, c, -R, R); return 1; }
	const double sq = sqrt(disc);
	const double x1 = (-a - sq) / 3.0, x2 = (-a + sq) / 3.0;
	const double f1 = cubic_eval(a, b, c, x1), f2 = cubic_eval(a, b, c, x2);
	// x1 is the local maximum, x2 the local minimum (f1 >= f2): a root left of x1 iff f1 >= 0, right of x2 iff f2 <= 0, between them iff both.  The
	// roots in ascending order with CONSTANT indices (round 5: `roots[n++]` put the array -- and, through the call, the caller's three models -- into
	// scratch: 240 B per lane)
	const bool left = f1 >= 0.0, right = f2 <= 0.0;
	double rl = 0.0, rm = 0.0, rr = 0.0;
	if (left) rl = bisect(a, b, c, -R, x1);
	if (left && right) rm = bisect(a, b, c, x1, x2);
	if (right) rr = bisect(a, b, c, x2, R);
	roots[0] = left ? rl : rr; roots[1] = rm; roots[2] = rr;
	// (neither: f1 < 0 < f2 -- a tiny positive disc with x1 ~ x2, rounding puts the maximum below and the minimum above zero: the oracle's form returns no root)
	return left && right ? 3 : ((left || right) ? 1 : 0);
}

// F3: the 7-point solver (static register indexing: every loop bound is a constant, the pivot row is applied by selects)
__device__ __forceinline__ int seven_point(const double2* m1, const double2* m2, const int (&idx)[7], const ssk_fm_params& P, double (&F)[3][9])
{
	double A[7][9];
#pragma unroll
	for (int d = 0; d < 3; ++d)
#pragma unroll
		for (int i = 0; i < 9; ++i) F[d][i] = 0.0;
#pragma unroll
	for (int r = 0; r < 7; ++r) {
		const double2 a = m1[idx[r]], b = m2[idx[r]];
		const double x1 = (a.x - P.norm_cx) * P.norm_scale, y1 = (a.y - P.norm_cy) * P.norm_scale;
		const double x2 = (b.x - P.norm_cx) * P.norm_scale, y2 = (b.y - P.norm_cy) * P.norm_scale;
		A[r][0] = x2 * x1; A[r][1] = x2 * y1; A[r][2] = x2; A[r][3] = y2 * x1; A[r][4] = y2 * y1; A[r][5] = y2;
		A[r][6] = x1; A[r][7] = y1; A[r][8] = 1.0;
	}
	bool ok = true;
#pragma unroll
	for (int c = 0; c < 7; ++c) {
		int p = c;
		double best = fabs(A[c][c]);
#pragma unroll
		for (int r = c + 1; r < 7; ++r) { const double v = fabs(A[r][c]); if (v > best) { best = v; p = r; } }
		ok = ok && best >= 1e-12;
#pragma unroll
		for (int r = c + 1; r < 7; ++r)
			if (p == r) {
#pragma unroll
				for (int j = 0; j < 9; ++j) { const double t = A[c][j]; A[c][j] = A[r][j]; A[r][j] = t; }
			}
		const double inv = 1.0 / A[c][c];
#pragma unroll
		for (int j = 0; j < 9; ++j) A[c][j] = A[c][j] * inv;
#pragma unroll
		for (int r = 0; r < 7; ++r) {
			if (r == c) continue;
			const double f = A[r][c];
#pragma unroll
			for (int j = 0; j < 9; ++j) A[r][j] = A[r][j] - f * A[c][j];
		}
	}
	if (!ok) return 0;                                   // (the oracle stops at the first small pivot: same verdict, its later values unused)
	double N1[9], N2[9], Mp[9], Mm[9];
#pragma unroll
	for (int j = 0; j < 7; ++j) { N1[j] = -A[j][7]; N2[j] = -A[j][8]; }
	N1[7] = 1.0; N1[8] = 0.0; N2[7] = 0.0; N2[8] = 1.0;
#pragma unroll
	for (int i = 0; i < 9; ++i) { Mp[i] = N1[i] + N2[i]; Mm[i] = N2[i] - N1[i]; }
	const double c0 = det3(N2), c3 = det3(N1), p1 = det3(Mp), pm = det3(Mm);
	const double c2 = 0.5 * (p1 + pm) - c0, c1 = 0.5 * (p1 - pm) - c3;
	double roots[3];
	const int nr = cubic_roots(c3, c2, c1, c0, roots);
	const double s = P.norm_scale, tx = -P.norm_scale * P.norm_cx, ty = -P.norm_scale * P.norm_cy;
	int nm = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		if (k >= nr) break;
		double G[9], GT[9], Fp[9];
#pragma unroll
		for (int i = 0; i < 9; ++i) G[i] = roots[k] * N1[i] + N2[i];
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			GT[3 * r] = G[3 * r] * s; GT[3 * r + 1] = G[3 * r + 1] * s;
			GT[3 * r + 2] = G[3 * r] * tx + G[3 * r + 1] * ty + G[3 * r + 2];
		}
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			Fp[c] = s * GT[c]; Fp[3 + c] = s * GT[3 + c];
			Fp[6 + c] = tx * GT[c] + ty * GT[3 + c] + GT[6 + c];
		}
		double nrm = 0;
#pragma unroll
		for (int i = 0; i < 9; ++i) nrm += Fp[i] * Fp[i];
		if (!(nrm > 0.0)) continue;
		const double inv = 1.0 / sqrt(nrm);
		// models are compacted in root order; nm is at most k, so the destination index is one of three constants
		// (selects, not a guarded store: the compiler turns `if (d == nm) F[d][i] = ...` back into F[nm][i] -- a dynamic index, i.e. the models in scratch)
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const bool here = d == nm;
#pragma unroll
			for (int i = 0; i < 9; ++i) F[d][i] = here ? Fp[i] * inv : F[d][i];
		}
		++nm;
	}
	return nm;
}

__device__ __forceinline__ bool fm_inlier(const double* F, const double2 p1, const double2 p2, double thr2)
{
	double a = F[0] * p1.x + F[1] * p1.y + F[2], b = F[3] * p1.x + F[4] * p1.y + F[5], c = F[6] * p1.x + F[7] * p1.y + F[8];
	const double n2 = a * a + b * b, d2 = p2.x * a + p2.y * b + c;
	a = F[0] * p2.x + F[3] * p2.y + F[6]; b = F[1] * p2.x + F[4] * p2.y + F[7]; c = F[2] * p2.x + F[5] * p2.y + F[8];
	const double n1 = a * a + b * b, d1 = p1.x * a + p1.y * b + c;
	return d2 * d2 <= thr2 * n2 && d1 * d1 <= thr2 * n1;
}

__device__ int fm_update_iters(double p, int good, int n, int max_iters)
{
	const double w = (double)good / (double)n;
	const double w2 = w * w, w4 = w2 * w2, base = 1.0 - w4 * w2 * w, target = 1.0 - p;
	double q = 1.0;
	int k = 0;
	while (q > target && k < max_iters) { q = q * base; ++k; }
	return k;
}

#ifndef SSK_FM_WAVES
#define SSK_FM_WAVES 2
#endif
__global__ __launch_bounds__(NT, SSK_FM_WAVES)
void fm_kernel(const int32_t* __restrict__ train_idx, int32_t* __restrict__ status, const int32_t* __restrict__ out_beg,
               const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end, const int32_t* __restrict__ t_beg,
               const float2* __restrict__ this_xy, const float2* __restrict__ other_xy, ssk_fm_params P, int cap,
               ssk_fm_result* __restrict__ results, int32_t* __restrict__ tracked_out, const unsigned long long* __restrict__ seeds)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // [m1 double2 x cap | m2 double2 x cap | idx_of int x cap]
	double2* s_m1 = reinterpret_cast<double2*>(s_dyn);
	double2* s_m2 = s_m1 + cap;
	int* s_idx = reinterpret_cast<int*>(s_m2 + cap);
	__shared__ double s_F[HR * 3 * 9];
	__shared__ double s_Fbest[9];
	__shared__ int s_cnt[HR * 3];
	__shared__ int s_nm[HR];
	__shared__ int s_scan[NT];
	__shared__ int s_niters, s_best, s_h, s_besth;

	const int pair = blockIdx.x, tid = threadIdx.x;
	const int q0 = q_beg[pair], n_this = min(q_end[pair] - q0, cap), t0 = t_beg[pair];
	const int o0 = out_beg ? out_beg[pair] : q0;

	// ordered compaction of the tracked pairs  [literal EST.cpp:2022-2040]
	const int per = (cap + NT - 1) / NT, kb = tid * per;
	int cnt = 0;
	for (int e = 0; e < per; ++e) { const int k = kb + e; cnt += k < n_this && status[o0 + k] == SSK_DA_TRACKED; }
	s_scan[tid] = cnt;
	__syncthreads();
	for (int o = 1; o < NT; o <<= 1) {
		const int v = tid >= o ? s_scan[tid - o] : 0;
		__syncthreads();
		s_scan[tid] += v;
		__syncthreads();
	}
	const int n = s_scan[NT - 1];
	int pos = s_scan[tid] - cnt;
	for (int e = 0; e < per; ++e) {
		const int k = kb + e;
		if (k < n_this && status[o0 + k] == SSK_DA_TRACKED) {
			const float2 a = this_xy[q0 + k], b = other_xy[t0 + train_idx[o0 + k]];
			s_m2[pos] = make_double2((double)a.x, (double)a.y);
			s_m1[pos] = make_double2((double)b.x, (double)b.y);
			s_idx[pos] = k;
			++pos;
		}
	}
	if (tid == 0) { s_niters = P.max_iters; s_best = 0; s_h = 0; s_besth = -1; }
	__syncthreads();

	const bool run = n >= P.min_points && n >= 7;                        // F1 (uniform)
	const double thr2 = P.threshold * P.threshold;
	const uint64_t seed = seeds ? (uint64_t)seeds[pair] : P.seed + (uint64_t)pair;      // (per-pair seeds: the un-batched caller seeds by keyframe ids)
	if (run) {
		const int hl = tid & (HR - 1), quarter = tid / HR;
		for (int r = 0; HR * r < s_niters; ++r) {                        // (s_niters only changes between the barriers below)
			if (tid < HR * 3) s_cnt[tid] = 0;
			__syncthreads();
			const int h = HR * r + hl;
			int idx[7];
			double F[3][9];
			fm_sample(seed, h, n, idx);
			const int nm = seven_point(s_m1, s_m2, idx, P, F);
			int good[3] = {0, 0, 0};
			for (int i = quarter; i < n; i += NT / HR) {
				const double2 p1 = s_m1[i], p2 = s_m2[i];
#pragma unroll
				for (int k = 0; k < 3; ++k) good[k] += (k < nm && fm_inlier(F[k], p1, p2, thr2)) ? 1 : 0;
			}
#pragma unroll
			for (int k = 0; k < 3; ++k) if (k < nm && good[k]) atomicAdd(&s_cnt[3 * hl + k], good[k]);
			if (quarter == 0) {
				s_nm[hl] = nm;
#pragma unroll
				for (int k = 0; k < 3; ++k)
#pragma unroll
					for (int i = 0; i < 9; ++i) s_F[(3 * hl + k) * 9 + i] = F[k][i];
			}
			__syncthreads();
			if (tid == 0) {                                               // F5: the sequential walk over this round's hypotheses
				int hh = s_h, niters = s_niters, best = s_best;
				for (; hh < niters && hh < HR * (r + 1); ++hh) {
					const int l = hh - HR * r;
					for (int k = 0; k < s_nm[l]; ++k) {
						const int g = s_cnt[3 * l + k];
						if (g > (best > 6 ? best : 6)) {
							best = g; s_besth = hh;
							for (int i = 0; i < 9; ++i) s_Fbest[i] = s_F[(3 * l + k) * 9 + i];
							niters = fm_update_iters(P.confidence, g, n, niters);
						}
					}
				}
				s_h = hh; s_niters = niters; s_best = best;
			}
			__syncthreads();
			if (s_h < HR * (r + 1)) break;                                // stopped inside this round (uniform)
		}
	}
	const bool valid = run && s_best > 0;
	int rej = 0;
	if (valid)
		for (int i = tid; i < n; i += NT)
			if (!fm_inlier(s_Fbest, s_m1[i], s_m2[i], thr2)) { status[o0 + s_idx[i]] = SSK_DA_REJ_FUND_MATRIX; ++rej; }   // :2047-2054
	__syncthreads();
	s_scan[tid] = rej;
	__syncthreads();
	for (int o = NT / 2; o > 0; o >>= 1) { if (tid < o) s_scan[tid] += s_scan[tid + o]; __syncthreads(); }
	if (tid == 0) {
		const int n_rej = s_scan[0];
		if (tracked_out) tracked_out[pair] = n - n_rej;
		if (results) {
			ssk_fm_result& R = results[pair];
			for (int i = 0; i < 9; ++i) R.F[i] = valid ? s_Fbest[i] : 0.0;
			R.valid = valid; R.n_pairs = n; R.n_inliers = valid ? s_best : 0; R.n_rejected = n_rej;
			R.hypotheses = run ? s_h : 0; R.best_hypothesis = valid ? s_besth : -1;
		}
	}
}

}  // namespace

int ssk_launch_fm(ssk_ctx* ctx, const int32_t* d_train_idx, int32_t* d_status, const int32_t* d_out_beg, const int32_t* d_q_beg,
                  const int32_t* d_q_end, const int32_t* d_t_beg, const float2* d_this_xy, const float2* d_other_xy,
                  int n_pairs, int cap, const ssk_fm_params& prm, ssk_fm_result* d_results, int32_t* d_tracked, const unsigned long long* d_seeds)
{
	if (n_pairs <= 0) return SSK_OK;
	if (cap < 1) cap = 1;
	if (cap > MAXM) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "fundamental-matrix filter: %d matches per keyframe (> %d)", cap, MAXM);
	cap = (cap + 15) & ~15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * (MAXM + 16)));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_FUND_MATRIX);
	hipLaunchKernelGGL(fm_kernel, dim3(n_pairs), dim3(NT), 36 * (size_t)cap, ctx->stream, d_train_idx, d_status, d_out_beg, d_q_beg, d_q_end, d_t_beg,
	                   d_this_xy, d_other_xy, prm, cap, d_results, d_tracked, d_seeds);
	SSK_LAUNCH_CHECK(ctx, "fm_kernel");
	return SSK_OK;
}

extern "C" {

int ssk_fm_filter(ssk_ctx* ctx, const float* this_xy, int n_this, const float* other_xy, int n_other,
                  const int32_t* train_idx, int32_t* status, const ssk_fm_params* params, ssk_fm_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_this < 0 || n_other < 0 || !params || !out || (n_this > 0 && (!this_xy || !train_idx || !status)) || (n_other > 0 && !other_xy))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_filter: bad argument");
	if (n_this > MAXM) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_fm_filter: %d matches (> %d)", n_this, MAXM);
	if (!(params->threshold > 0) || !(params->norm_scale > 0) || params->max_iters < 1)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_filter: threshold, norm_scale and max_iters must be positive");
	for (int k = 0; k < n_this; ++k)
		if (status[k] == SSK_DA_TRACKED && (train_idx[k] < 0 || train_idx[k] >= n_other))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_fm_filter: tracked match %d points outside the other keyframe", k);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const size_t nt = (size_t)(n_this > 0 ? n_this : 1), no = (size_t)(n_other > 0 ? n_other : 1);
	int rc;
	const size_t total = ssk_pad(8 * nt) + ssk_pad(8 * no) + 2 * ssk_pad(4 * nt) + ssk_pad(sizeof(ssk_fm_result)) + ssk_pad(16);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, total + 2 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, total + 2 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	// one transfer in (coordinates, train indices, ranges, statuses), one out (statuses, result)
	ssk_xfer_pack pk; pk.begin(ctx, total);
	int32_t *h_st, *h_rng;
	ssk_fm_result* h_out;
	float2* d_tx = pk.put(reinterpret_cast<const float2*>(this_xy), (size_t)n_this);
	pk.off = ssk_pad(8 * nt);
	float2* d_ox = pk.put(reinterpret_cast<const float2*>(other_xy), (size_t)n_other);
	pk.off = ssk_pad(8 * nt) + ssk_pad(8 * no);
	int32_t* d_ti = pk.put(train_idx, (size_t)n_this);
	pk.off = ssk_pad(8 * nt) + ssk_pad(8 * no) + ssk_pad(4 * nt);
	int32_t* d_rng = pk.slot<int32_t>(4, &h_rng);
	h_rng[0] = 0; h_rng[1] = n_this; h_rng[2] = 0; h_rng[3] = 0;
	const size_t st_begin = pk.off;
	int32_t* d_st = pk.put(status, (size_t)n_this);
	pk.off = st_begin + ssk_pad(4 * nt);
	h_st = reinterpret_cast<int32_t*>(pk.h + st_begin);
	const size_t in_end = pk.off;
	ssk_fm_result* d_out = pk.slot<ssk_fm_result>(1, &h_out);
	SSK_HIP(ctx, pk.upload(ctx->stream, 0, in_end));
	if ((rc = ssk_launch_fm(ctx, d_ti, d_st, d_rng + 3, d_rng, d_rng + 1, d_rng + 2, d_tx, d_ox, 1, n_this, *params, d_out, nullptr))) return rc;
	SSK_HIP(ctx, pk.download(ctx->stream, st_begin, pk.off));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int k = 0; k < n_this; ++k) status[k] = h_st[k];
	*out = *h_out;
	return SSK_OK;
}

int ssk_da_fundamental_filter(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                              const ssk_dmatch* da_matches, int32_t* status, const ssk_fm_params* params, ssk_fm_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!this_kf || !other_kf || !params || !out || (this_kf->n_matches > 0 && (!da_matches || !status)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_fundamental_filter: bad argument");
	auto gather = [&](const ssk_kf_view* kf, std::vector<float>& xy) -> bool {      // EST.cpp:2022-2040: left keypoints by match
		xy.resize(2 * (size_t)kf->n_matches);
		for (int k = 0; k < kf->n_matches; ++k) {
			const int l = kf->matches[k].queryIdx;
			if (l < 0 || l >= kf->n_left) return false;
			xy[2 * (size_t)k] = kf->kp_left[l].x; xy[2 * (size_t)k + 1] = kf->kp_left[l].y;
		}
		return true;
	};
	std::vector<float> tx, ox;
	if (!gather(this_kf, tx) || !gather(other_kf, ox)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_fundamental_filter: stereo match index outside the keypoint list");
	std::vector<int32_t> ti((size_t)this_kf->n_matches);
	for (int k = 0; k < this_kf->n_matches; ++k) ti[k] = da_matches[k].trainIdx;
	return ssk_fm_filter(ctx, tx.data(), this_kf->n_matches, ox.data(), other_kf->n_matches, ti.data(), status, params, out);
}

}  // extern "C"
