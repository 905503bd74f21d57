// ssk_cip.hip -- change-in-pose outlier filter of the data association: replaces
// m_detect_outliers_with_change_in_pose (EST.cpp:2113-2177) including stereo-vo's getChangeInPose
// (call site EST.cpp:2139-2146).  Contract P1-P6 of DESIGN.md "Change-in-pose contract" (oracle/oracle_cip.c).
//
// Mapping (gfx950): one 256-thread workgroup per keyframe pair, persistent over both Gauss-Newton stages.
//   * the tracked pairs are staged once in LDS: 3-D point of the other keyframe's stereo match (projectMatchTo3D,
//     UTIL.h:558-574, FP64) and this keyframe's 4 pixel coordinates; a byte per match says whether it takes part;
//   * one pass per iteration: every thread linearises its matches (k = tid, tid+256, ...) at the current pose and
//     accumulates the 21 + 6 + 1 numbers of (H upper triangle, g, robust cost); fixed shuffle tree inside a wave,
//     the four wave partials added in wave order: deterministic;
//   * the 6x6 Cholesky solve and the SE(3) update run redundantly in every thread (no broadcast needed).
// The cost after a step comes out of the next linearisation pass, so an iteration is one pass over the pairs.
// Bound: latency (a few hundred pairs, <= 60 serial iterations); tens of KB per pair, everything else on chip.
#include "ssk_frontend.h"
#include <vector>

// parity with the oracle is a relative tolerance here (FP64 sums in another order), not bit-exactness
#pragma clang fp contract(fast)

namespace {

#ifndef SSK_CIP_NT
#define SSK_CIP_NT 128                  // (256: 0.46 ms per 1 024 pairs, 128 and 64: 0.21 -- the kernel is latency per Gauss-Newton pass at 256 VGPRs: the smaller the workgroup, the more pairs a CU holds)
#endif
constexpr int NT = SSK_CIP_NT, NW = NT / 64;
constexpr int MAXM = SSK_MAX_FEATS;      // matches per keyframe

struct cip_pose { double R[9], t[3]; };

__device__ __forceinline__ void so3_exp(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
	const double th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j)
			R[3 * i + j] = a * K[3 * i + j] + b * (K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j]) + (i == j ? 1.0 : 0.0);
}

// residual r = z - h(R P + t), its squared norm; optionally the 4x6 Jacobian J = D [I | -[q]x]  (rows: q x D_a)
template <bool WITH_J>
__device__ __forceinline__ double pair_eval(const ssk_cip_params& c, const cip_pose& T, const double* P, const float4 z, double* r, double* J)
{
	double q[3];
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T.R[3 * i] * P[0] + T.R[3 * i + 1] * P[1] + T.R[3 * i + 2] * P[2] + T.t[i];
	const double iz = 1.0 / q[2], iz2 = iz * iz;
	r[0] = (double)z.x - (c.cx_l + c.fx_l * q[0] * iz);
	r[1] = (double)z.y - (c.cy_l + c.fy_l * q[1] * iz);
	r[2] = (double)z.z - (c.cx_r + c.fx_r * (q[0] - c.baseline) * iz);
	r[3] = (double)z.w - (c.cy_r + c.fy_r * q[1] * iz);
	if (WITH_J) {
		const double D[12] = {c.fx_l * iz, 0, -c.fx_l * q[0] * iz2,
		                      0, c.fy_l * iz, -c.fy_l * q[1] * iz2,
		                      c.fx_r * iz, 0, -c.fx_r * (q[0] - c.baseline) * iz2,
		                      0, c.fy_r * iz, -c.fy_r * q[1] * iz2};
#pragma unroll
		for (int a = 0; a < 4; ++a) {
			const double* d = D + 3 * a;
			J[6 * a] = d[0]; J[6 * a + 1] = d[1]; J[6 * a + 2] = d[2];
			J[6 * a + 3] = q[1] * d[2] - q[2] * d[1];
			J[6 * a + 4] = q[2] * d[0] - q[0] * d[2];
			J[6 * a + 5] = q[0] * d[1] - q[1] * d[0];
		}
	}
	return r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
}

// workgroup sum of K values per thread: shuffle tree in the wave, wave partials in wave order; every thread gets the sums
template <int K>
__device__ __forceinline__ void block_sum_k(double (&v)[K], double* s_red)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int i = 0; i < K; ++i)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
	__syncthreads();
	if (lane == 0) {
#pragma unroll
		for (int i = 0; i < K; ++i) s_red[wave * K + i] = v[i];
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < K; ++i) {
		double s = s_red[i];
#pragma unroll
		for (int w = 1; w < NW; ++w) s += s_red[w * K + i];
		v[i] = s;
	}
}

// H (lower triangle, row-major packed 21), g (6), cost at pose T over the matches flagged in s_act
__device__ __forceinline__ void build_system(const ssk_cip_params& c, const cip_pose& T, int n, const double* s_P, const float4* s_z,
                                             const uint8_t* s_act, double* s_red, double (&acc)[28])
{
#pragma unroll
	for (int i = 0; i < 28; ++i) acc[i] = 0.0;
	const double ik2 = 1.0 / (c.kernel_param * c.kernel_param);
	for (int k = threadIdx.x; k < n; k += NT) {
		if (!s_act[k]) continue;
		double r[4], J[24];
		const double s2 = pair_eval<true>(c, T, s_P + 3 * k, s_z[k], r, J);
		const double root = sqrt(1.0 + s2 * ik2);
		const double om = c.use_robust_kernel ? 1.0 / root : 1.0;
		acc[27] += c.use_robust_kernel ? 2.0 * c.kernel_param * c.kernel_param * (root - 1.0) : s2;
#pragma unroll
		for (int a = 0; a < 6; ++a) {
			acc[21 + a] += om * (J[a] * r[0] + J[6 + a] * r[1] + J[12 + a] * r[2] + J[18 + a] * r[3]);
#pragma unroll
			for (int b = 0; b <= a; ++b)
				acc[a * (a + 1) / 2 + b] += om * (J[a] * J[b] + J[6 + a] * J[6 + b] + J[12 + a] * J[12 + b] + J[18 + a] * J[18 + b]);
		}
	}
	block_sum_k<28>(acc, s_red);
}

// Cholesky LL^T solve of the packed 6x6 system; d <- H^-1 g; false on a non positive pivot
__device__ __forceinline__ bool solve6(const double (&acc)[28], double (&d)[6])
{
	double L[21];
#pragma unroll
	for (int i = 0; i < 21; ++i) L[i] = acc[i];
	bool ok = true;
#pragma unroll
	for (int j = 0; j < 6; ++j) {
		double p = L[j * (j + 1) / 2 + j];
#pragma unroll
		for (int k = 0; k < j; ++k) p -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
		ok = ok && p > 0.0;
		const double l = sqrt(p), il = 1.0 / l;
		L[j * (j + 1) / 2 + j] = l;
#pragma unroll
		for (int i = j + 1; i < 6; ++i) {
			double s = L[i * (i + 1) / 2 + j];
#pragma unroll
			for (int k = 0; k < j; ++k) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
			L[i * (i + 1) / 2 + j] = s * il;
		}
	}
#pragma unroll
	for (int i = 0; i < 6; ++i) {
		double s = acc[21 + i];
#pragma unroll
		for (int k = 0; k < i; ++k) s -= L[i * (i + 1) / 2 + k] * d[k];
		d[i] = s / L[i * (i + 1) / 2 + i];
	}
#pragma unroll
	for (int i = 5; i >= 0; --i) {
		double s = d[i];
#pragma unroll
		for (int k = i + 1; k < 6; ++k) s -= L[k * (k + 1) / 2 + i] * d[k];
		d[i] = s / L[i * (i + 1) / 2 + i];
	}
	return ok;
}

// T <- exp(d) (+) T, d = (t, w)   (L3)
__device__ __forceinline__ void pose_update(cip_pose& T, const double (&d)[6])
{
	double Rd[9], Rn[9], tn[3];
	so3_exp(&d[3], Rd);
#pragma unroll
	for (int i = 0; i < 3; ++i) {
#pragma unroll
		for (int j = 0; j < 3; ++j) Rn[3 * i + j] = Rd[3 * i] * T.R[j] + Rd[3 * i + 1] * T.R[3 + j] + Rd[3 * i + 2] * T.R[6 + j];
		tn[i] = Rd[3 * i] * T.t[0] + Rd[3 * i + 1] * T.t[1] + Rd[3 * i + 2] * T.t[2] + d[i];
	}
#pragma unroll
	for (int i = 0; i < 9; ++i) T.R[i] = Rn[i];
#pragma unroll
	for (int i = 0; i < 3; ++i) T.t[i] = tn[i];
}

// one stage (P4); returns the iterations done, -1 on a non positive definite system
__device__ int run_stage(const ssk_cip_params& c, cip_pose& T, int n, const double* s_P, const float4* s_z, const uint8_t* s_act,
                         double* s_red, int max_it)
{
	double acc[28];
	build_system(c, T, n, s_P, s_z, s_act, s_red, acc);
	double prev = acc[27];
	int incr = 0, it = 0;
	while (it < max_it) {
		double d[6];
		if (!solve6(acc, d)) return -1;
		pose_update(T, d);
		++it;
		const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
		if (nd < c.min_mod_out_vector) break;
		build_system(c, T, n, s_P, s_z, s_act, s_red, acc);         // cost at the new pose + the next system in one pass
		if (acc[27] > prev && ++incr > c.max_incr_cost) break;
		prev = acc[27];
	}
	return it;
}

#ifndef SSK_CIP_WAVES
#define SSK_CIP_WAVES 1
#endif
__global__ __launch_bounds__(NT, SSK_CIP_WAVES)
void cip_kernel(const int32_t* __restrict__ train_idx, int32_t* __restrict__ status, const int32_t* __restrict__ out_beg,
                const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end, const int32_t* __restrict__ t_beg,
                const float4* __restrict__ this_obs, const float4* __restrict__ other_obs, const double* __restrict__ init_poses,
                ssk_cip_params c, int cap, double* __restrict__ residuals, ssk_cip_result* __restrict__ results, int32_t* __restrict__ tracked_out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // [z float4 x cap | P double x 3 cap | act | trk]
	__shared__ double s_red[NW * 28];
	float4* s_z = reinterpret_cast<float4*>(s_dyn);
	double* s_P = reinterpret_cast<double*>(s_dyn + 16 * (size_t)cap);
	uint8_t* s_act = s_dyn + 40 * (size_t)cap;
	uint8_t* s_trk = s_act + cap;

	const int pair = blockIdx.x, tid = threadIdx.x;
	const int q0 = q_beg[pair], n = min(q_end[pair] - q0, cap), t0 = t_beg[pair];
	const int o0 = out_beg ? out_beg[pair] : q0;

	// P1: stage the tracked pairs  [literal EST.cpp:2120-2133]
	double cnt[1] = {0.0};
	for (int k = tid; k < n; k += NT) {
		const bool trk = status[o0 + k] == SSK_DA_TRACKED;
		s_trk[k] = trk; s_act[k] = trk;
		if (trk) {
			const float4 o = other_obs[t0 + train_idx[o0 + k]];
			const double ul = (double)o.x, vl = (double)o.y, ur = (double)o.z;
			const double b_d = c.baseline / (c.fx_l * (c.cx_r - ur) + c.fx_r * (ul - c.cx_l));      // UTIL.h:558-574
			s_P[3 * k] = b_d * c.fx_r * (ul - c.cx_l);
			s_P[3 * k + 1] = b_d * c.fx_r * (vl - c.cy_l);
			s_P[3 * k + 2] = b_d * c.fx_l * c.fx_r;
			s_z[k] = this_obs[q0 + k];
			cnt[0] += 1.0;
		}
	}
	block_sum_k<1>(cnt, s_red);                              // (also the barrier that publishes the staged pairs)
	const int n_pairs = (int)cnt[0];

	cip_pose T;
	if (init_poses) {
		const double* ip = init_poses + 12 * (size_t)pair;
#pragma unroll
		for (int i = 0; i < 3; ++i) { T.R[3 * i] = ip[4 * i]; T.R[3 * i + 1] = ip[4 * i + 1]; T.R[3 * i + 2] = ip[4 * i + 2]; T.t[i] = ip[4 * i + 3]; }
	} else {
#pragma unroll
		for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
		T.t[0] = T.t[1] = T.t[2] = 0.0;
	}

	bool valid = n_pairs >= c.bad_tracking_th;
	int it1 = 0, it2 = 0, n_in = 0;
	if (valid) { it1 = run_stage(c, T, n, s_P, s_z, s_act, s_red, c.initial_max_iters); if (it1 < 0) { valid = false; it1 = 0; } }
	if (valid) {                                             // P5: inlier gate between the stages
		double ci[1] = {0.0};
		for (int k = tid; k < n; k += NT) {
			if (!s_trk[k]) continue;
			double r[4];
			const bool in = pair_eval<false>(c, T, s_P + 3 * k, s_z[k], r, nullptr) <= c.residual_threshold;
			s_act[k] = in; ci[0] += in ? 1.0 : 0.0;
		}
		block_sum_k<1>(ci, s_red);
		n_in = (int)ci[0];
		if (n_in < c.bad_tracking_th) valid = false;
	}
	if (valid) { it2 = run_stage(c, T, n, s_P, s_z, s_act, s_red, c.max_iters); if (it2 < 0) { valid = false; it2 = 0; } }

	// P6 + [literal EST.cpp:2152-2176]: residual of every tracked pair at the final pose, the gate, the recount
	const double ik2 = 1.0 / (c.kernel_param * c.kernel_param);
	double fin[3] = {0.0, 0.0, 0.0};                         // rejected, still tracked, cost of the inliers
	for (int k = tid; k < n; k += NT) {
		double s2 = -1.0;
		if (s_trk[k]) {
			double r[4];
			s2 = valid ? pair_eval<false>(c, T, s_P + 3 * k, s_z[k], r, nullptr) : 0.0;
			if (!valid || s2 > c.residual_th) { status[o0 + k] = SSK_DA_REJ_CHANGE_POSE; fin[0] += 1.0; } else fin[1] += 1.0;
			if (valid && s_act[k]) fin[2] += c.use_robust_kernel ? 2.0 * c.kernel_param * c.kernel_param * (sqrt(1.0 + s2 * ik2) - 1.0) : s2;
		}
		if (residuals) residuals[o0 + k] = s2;
	}
	block_sum_k<3>(fin, s_red);
	if (tid == 0) {
		if (tracked_out) tracked_out[pair] = (int)fin[1];
		if (results) {
			ssk_cip_result& R = results[pair];
#pragma unroll
			for (int i = 0; i < 3; ++i) { R.pose[4 * i] = T.R[3 * i]; R.pose[4 * i + 1] = T.R[3 * i + 1]; R.pose[4 * i + 2] = T.R[3 * i + 2]; R.pose[4 * i + 3] = T.t[i]; }
			R.cost = valid ? fin[2] : 0.0;
			R.valid = valid; R.iters_initial = it1; R.iters_final = it2; R.n_pairs = n_pairs; R.n_inliers = n_in; R.n_rejected = (int)fin[0];
		}
	}
}

}  // namespace

int ssk_launch_cip(ssk_ctx* ctx, const int32_t* d_train_idx, int32_t* d_status, const int32_t* d_out_beg, const int32_t* d_q_beg,
                   const int32_t* d_q_end, const int32_t* d_t_beg, const float4* d_this_obs, const float4* d_other_obs,
                   const double* d_init_poses, int n_pairs, int cap, const ssk_cip_params& prm, double* d_residuals, ssk_cip_result* d_results,
                   int32_t* d_tracked)
{
	if (n_pairs <= 0) return SSK_OK;
	if (cap < 1) cap = 1;
	if (cap > MAXM) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "change-in-pose filter: %d matches per keyframe (> %d)", cap, MAXM);
	cap = (cap + 15) & ~15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 42 * (MAXM + 16)));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_CHANGE_IN_POSE);
	hipLaunchKernelGGL(cip_kernel, dim3(n_pairs), dim3(NT), 42 * (size_t)cap, ctx->stream, d_train_idx, d_status, d_out_beg, d_q_beg, d_q_end, d_t_beg,
	                   d_this_obs, d_other_obs, d_init_poses, prm, cap, d_residuals, d_results, d_tracked);
	SSK_LAUNCH_CHECK(ctx, "cip_kernel");
	return SSK_OK;
}

extern "C" {

int ssk_cip_filter(ssk_ctx* ctx, const float* this_obs, int n_this, const float* other_obs, int n_other,
                   const int32_t* train_idx, int32_t* status, const double init_pose[12],
                   const ssk_cip_params* params, double* residuals, ssk_cip_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_this < 0 || n_other < 0 || !params || !out || !init_pose ||
	    (n_this > 0 && (!this_obs || !train_idx || !status)) || (n_other > 0 && !other_obs))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_filter: bad argument");
	if (n_this > MAXM) return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_cip_filter: %d matches (> %d)", n_this, MAXM);
	if (params->use_robust_kernel && !(params->kernel_param > 0)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_filter: kernel_param must be > 0");
	for (int k = 0; k < n_this; ++k)
		if (status[k] == SSK_DA_TRACKED && (train_idx[k] < 0 || train_idx[k] >= n_other))
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_cip_filter: tracked match %d points outside the other keyframe", k);
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const size_t nt = (size_t)(n_this > 0 ? n_this : 1), no = (size_t)(n_other > 0 ? n_other : 1);
	int rc;
	const size_t total = ssk_pad(16 * nt) + ssk_pad(16 * no) + 2 * ssk_pad(4 * nt) + ssk_pad(8 * nt) + ssk_pad(96) + ssk_pad(sizeof(ssk_cip_result)) + ssk_pad(16);
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, total + 2 * 256))) return rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->pin, total + 2 * 256))) return rc;
	ssk_arena_reset(ctx->dev); ssk_arena_reset(ctx->pin);
	// one transfer in (observations, train indices, start pose, ranges, statuses), one out (statuses, residuals, result)
	ssk_xfer_pack pk; pk.begin(ctx, total);
	int32_t* h_rng;
	double* h_res;
	ssk_cip_result* h_out;
	float4* d_to = pk.put(reinterpret_cast<const float4*>(this_obs), (size_t)n_this);
	pk.off = ssk_pad(16 * nt);
	float4* d_oo = pk.put(reinterpret_cast<const float4*>(other_obs), (size_t)n_other);
	pk.off = ssk_pad(16 * nt) + ssk_pad(16 * no);
	int32_t* d_ti = pk.put(train_idx, (size_t)n_this);
	pk.off = ssk_pad(16 * nt) + ssk_pad(16 * no) + ssk_pad(4 * nt);
	double* d_T = pk.put(init_pose, 12);
	int32_t* d_rng = pk.slot<int32_t>(4, &h_rng);
	h_rng[0] = 0; h_rng[1] = n_this; h_rng[2] = 0; h_rng[3] = 0;      // q_beg, q_end, t_beg, out_beg
	const size_t st_begin = pk.off;
	int32_t* d_st = pk.put(status, (size_t)n_this);
	pk.off = st_begin + ssk_pad(4 * nt);
	const int32_t* h_st = reinterpret_cast<const int32_t*>(pk.h + st_begin);
	const size_t in_end = pk.off;
	double* d_res = pk.slot<double>(nt, &h_res);
	ssk_cip_result* d_out = pk.slot<ssk_cip_result>(1, &h_out);
	SSK_HIP(ctx, pk.upload(ctx->stream, 0, in_end));
	if ((rc = ssk_launch_cip(ctx, d_ti, d_st, d_rng + 3, d_rng, d_rng + 1, d_rng + 2, d_to, d_oo, d_T, 1, n_this, *params, d_res, d_out, nullptr))) return rc;
	SSK_HIP(ctx, pk.download(ctx->stream, st_begin, pk.off));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	int j = 0;
	for (int k = 0; k < n_this; ++k) {
		if (residuals && status[k] == SSK_DA_TRACKED) residuals[j++] = h_res[k];     // one per tracked pair, in match order (out_residual)
		status[k] = h_st[k];
	}
	*out = *h_out;
	return SSK_OK;
}

int ssk_da_change_in_pose(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                          const ssk_dmatch* da_matches, int32_t* status, const double init_pose[12],
                          const ssk_cip_params* params, double* residuals, ssk_cip_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (!this_kf || !other_kf || !params || !out || !init_pose || (this_kf->n_matches > 0 && (!da_matches || !status)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_change_in_pose: bad argument");
	// marshalling of EST.cpp:2139-2146: per stereo match the left/right keypoint pixels of both keyframes
	auto gather = [&](const ssk_kf_view* kf, std::vector<float>& obs) -> bool {
		obs.resize(4 * (size_t)kf->n_matches);
		for (int k = 0; k < kf->n_matches; ++k) {
			const int l = kf->matches[k].queryIdx, r = kf->matches[k].trainIdx;
			if (l < 0 || l >= kf->n_left || r < 0 || r >= kf->n_right) return false;
			obs[4 * (size_t)k] = kf->kp_left[l].x; obs[4 * (size_t)k + 1] = kf->kp_left[l].y;
			obs[4 * (size_t)k + 2] = kf->kp_right[r].x; obs[4 * (size_t)k + 3] = kf->kp_right[r].y;
		}
		return true;
	};
	std::vector<float> to, oo;
	if (!gather(this_kf, to) || !gather(other_kf, oo)) return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_da_change_in_pose: stereo match index outside the keypoint list");
	std::vector<int32_t> ti((size_t)this_kf->n_matches);
	for (int k = 0; k < this_kf->n_matches; ++k) ti[k] = da_matches[k].trainIdx;
	return ssk_cip_filter(ctx, to.data(), this_kf->n_matches, oo.data(), other_kf->n_matches, ti.data(), status, init_pose, params, residuals, out);
}

}  // extern "C"
