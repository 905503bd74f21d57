// ssk_lm.hip -- relative bundle adjustment on the GPU: Levenberg-Marquardt with Schur complement on the
// landmarks and dense Cholesky of the reduced pose system, for SRBA's problem types
// (kf2kf SE(3) edges, Euclidean3D landmarks, 4-D stereo observations, sensor pose on robot;
// reference SRBA.h:30-54, solver_LM_schur_dense_cholesky SRBA.h:34), entered in the reference through
// rba.define_new_keyframe (EST.cpp:166,782).  Contract L1-L5 of DESIGN.md "LM contract".
//
// Mapping (gfx950): ONE 512-thread workgroup per optimisation window, persistent over the whole LM loop
// (a keyframe's window is ~10 edges / ~1.5-2.5k observations / a few Mflop per iteration: launch- and
// latency-bound, so the design removes every launch from the loop, keeps the serial algebra in LDS and
// gives every thread long runs of independent loads).  A batch of windows = a grid of workgroups.
// Per linearisation / attempt, separated by workgroup barriers:
//   A   item = observation: SE(3) chain along the spanning-tree path (edge poses cached in LDS), stereo
//       projection, residual, IRLS weight, weighted Jacobian blocks 4x3 / 4x6 -> per-window scratch (L2);
//   B1  item = (landmark, optimised edge) incidence: W = sum Je^T Jf (6x3), g_e part = sum Je^T r, in registers;
//   B2  item = optimised landmark: H_ff (3x3), g_f;
//   D   item = (landmark, edge pair) block: P = sum Je_i^T Je_j (6x6) in 36 registers, 48 independent loads per
//       contributing observation, stored as a 36-double slab;
//   E   V = (H_ff + lambda I)^-1 per landmark, T = W V per incidence;
//   F1  item = block slab: C = P - T_i W_j^T;   F2: H_red block (e,e') = lambda I + contiguous sum of its slabs
//       (slabs are sorted by block pair, so the reduction is a coalesced streaming sum), g_red likewise;
//   G   dense Cholesky LL^T + both triangular solves by ONE wavefront on the LDS-resident matrix (no
//       workgroup barriers inside; falls back to a workgroup version on an L2-resident matrix when 6E > ~120);
//   H   back-substitution for the landmarks, trial state (Rodrigues update of the edges, into LDS);
//   I   total (robustified) error of the trial state, accept / reject, lambda schedule.
// Every sum runs in an order fixed by the symbolic structure or by the fixed shuffle tree (no floating-point
// atomics), so a solve is reproducible run to run.  All arithmetic FP64.  Bound: latency, not bandwidth.
#include "ssk_lm.h"

// LM parity is a 1e-6 relative tolerance (contract L5), not bit-exactness: let the compiler fuse multiply-adds here
// (the build disables contraction globally for the bit-exact front-end kernels)
#pragma clang fp contract(fast)

namespace {

#include "ssk_lm_device.h"

// phase A: linearise observation o at the current state (pose: window-relative current poses)
__device__ void linearize_obs(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* pose, int o)
{
	const int pb = D.obs_path_off[W.opo0 + o], n = D.obs_path_off[W.opo0 + o + 1] - pb;
	double q[3], t[3];
	{
		const double* f = D.lm + 3 * (size_t)(W.lm0 + D.obs_lm[W.obs0 + o]);
		q[0] = f[0]; q[1] = f[1]; q[2] = f[2];
	}
	// backward pass: q_i = X_i (+) q_{i+1}; path_q[entry i] keeps the point a Jacobian of entry i needs:
	// dir 0 -> q_i (after the step), dir 1 -> q_{i+1} (before it)
	for (int i = n - 1; i >= 0; --i) {
		const int c = D.path_edge[W.path0 + pb + i];
		const double* T = pose + 12 * (c >> 1);
		double* pq = D.path_q + 3 * (size_t)(W.path0 + pb + i);
		if ((c & 1) == 0) { se3_apply(T, q, t); pq[0] = t[0]; pq[1] = t[1]; pq[2] = t[2]; }
		else { pq[0] = q[0]; pq[1] = q[1]; pq[2] = q[2]; se3_apply_inv(T, q, t); }
		q[0] = t[0]; q[1] = t[1]; q[2] = t[2];
	}
	double pc[3], h[4], r[4];
	se3_apply_inv(P.S, q, pc);
	project(P, pc, h);
	const float* z = D.obs_px + 4 * (size_t)(W.obs0 + o);
	double s2 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { r[k] = (double)z[k] - h[k]; s2 += r[k] * r[k]; }
	const double w = P.robust ? sqrt(1.0 / sqrt(1.0 + s2 / (P.kernel * P.kernel))) : 1.0;
	const double iz = 1.0 / pc[2], iz2 = iz * iz;
	const double Dm[12] = {P.fx_l * iz, 0, -P.fx_l * pc[0] * iz2,
	                       0, P.fy_l * iz, -P.fy_l * pc[1] * iz2,
	                       P.fx_r * iz, 0, -P.fx_r * (pc[0] - P.baseline) * iz2,
	                       0, P.fy_r * iz, -P.fy_r * pc[1] * iz2};
	double M[12];     // D * R_S^T
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j)
			M[3 * i + j] = Dm[3 * i] * P.S[4 * j] + Dm[3 * i + 1] * P.S[4 * j + 1] + Dm[3 * i + 2] * P.S[4 * j + 2];
	double* orr = D.obs_r + 4 * (size_t)(W.obs0 + o);
#pragma unroll
	for (int k = 0; k < 4; ++k) orr[k] = w * r[k];
	// forward pass: RA = product of the step rotations so far
	double RA[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int i = 0; i < n; ++i) {
		const int c = D.path_edge[W.path0 + pb + i];
		const int e = c >> 1, dir = c & 1;
		const double* T = pose + 12 * e;
		double RN[9];
#pragma unroll
		for (int a = 0; a < 3; ++a)
#pragma unroll
			for (int b = 0; b < 3; ++b)
				RN[3 * a + b] = dir == 0 ? RA[3 * a] * T[b] + RA[3 * a + 1] * T[4 + b] + RA[3 * a + 2] * T[8 + b]
				                         : RA[3 * a] * T[4 * b] + RA[3 * a + 1] * T[4 * b + 1] + RA[3 * a + 2] * T[4 * b + 2];
		if (D.edge_slot[W.edge0 + e] >= 0) {
			const double* RR = dir == 0 ? RA : RN;
			const double sg = dir == 0 ? w : -w;
			const double* qq = D.path_q + 3 * (size_t)(W.path0 + pb + i);
			double* J = D.path_Je + 24 * (size_t)(W.path0 + pb + i);
#pragma unroll
			for (int a = 0; a < 4; ++a) {
				const double A0 = sg * (M[3 * a] * RR[0] + M[3 * a + 1] * RR[3] + M[3 * a + 2] * RR[6]);
				const double A1 = sg * (M[3 * a] * RR[1] + M[3 * a + 1] * RR[4] + M[3 * a + 2] * RR[7]);
				const double A2 = sg * (M[3 * a] * RR[2] + M[3 * a + 1] * RR[5] + M[3 * a + 2] * RR[8]);
				J[6 * a] = A0; J[6 * a + 1] = A1; J[6 * a + 2] = A2;
				J[6 * a + 3] = -A1 * qq[2] + A2 * qq[1];
				J[6 * a + 4] = A0 * qq[2] - A2 * qq[0];
				J[6 * a + 5] = -A0 * qq[1] + A1 * qq[0];
			}
		}
#pragma unroll
		for (int a = 0; a < 9; ++a) RA[a] = RN[a];
	}
	double* Jf = D.obs_Jf + 12 * (size_t)(W.obs0 + o);
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j)
			Jf[3 * i + j] = w * (M[3 * i] * RA[j] + M[3 * i + 1] * RA[3 + j] + M[3 * i + 2] * RA[6 + j]);
}

__device__ void sym3_inv(const double* V, double* I)
{
	const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
	const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
	const double det = a * c00 + b * c01 + c * c02;
	const double id = 1.0 / det;
	I[0] = c00 * id; I[1] = I[3] = c01 * id; I[2] = I[6] = c02 * id;
	I[4] = (a * f - c * c) * id; I[5] = I[7] = (b * c - a * e) * id; I[8] = (a * d - b * b) * id;
}

// Sum of a 36-vector over the 64 lanes of a wave in LANE ORDER (deterministic), through the wave's LDS scratch:
// two halves of 18 values, each lane writes its row (pitch 19 doubles), lanes 0..17 add a column each and
// store the result (coalesced).  ~36 ds_write + 64 ds_read per lane instead of 36 x 6 cross-lane double shuffles.
constexpr int RED_PITCH = 19;
constexpr int RED_WAVE = 64 * RED_PITCH;           // doubles of LDS scratch per wave
__device__ __forceinline__ void wave_sum36(const double (&acc)[36], double* red, int lane, double* __restrict__ dst)
{
#pragma unroll
	for (int h = 0; h < 2; ++h) {
#pragma unroll
		for (int i = 0; i < 18; ++i) red[lane * RED_PITCH + i] = acc[18 * h + i];
		WAVE_SYNC();
		if (lane < 18) {
			double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
			for (int r = 0; r < 64; r += 4) {
				s0 += red[r * RED_PITCH + lane]; s1 += red[(r + 1) * RED_PITCH + lane];
				s2 += red[(r + 2) * RED_PITCH + lane]; s3 += red[(r + 3) * RED_PITCH + lane];
			}
			dst[18 * h + lane] = (s0 + s1) + (s2 + s3);
		}
		WAVE_SYNC();
	}
}

__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_kernel(lm_dev D, lm_cam P, int lds_n6_cap)
{
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // Cholesky matrix n6 x (n6|1) when it fits; before that, per-wave reduction scratch
	__shared__ double s_pose[SSK_LM_LDS_EDGES * 12];
	__shared__ double s_pose_new[SSK_LM_LDS_EDGES * 12];
	__shared__ double s_b[SSK_LM_MAX_N6];
	__shared__ double s_inv[128];                                     // reciprocal pivots of the LDS solve (n6 <= 127)
	__shared__ double s_part[NW];
	__shared__ double s_lambda, s_nu, s_err, s_piv;
	__shared__ int s_flag, s_accept, s_stop, s_iters, s_needlin, s_accepted;

	const lm_win W = D.win[blockIdx.x];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int n6 = 6 * W.E;
	const bool lds_mat = n6 <= lds_n6_cap;
	const int np = lds_mat ? (n6 | 1) : n6;
	double* A = lds_mat ? s_dyn : (D.Hred + W.h0);
#ifdef SSK_LM_STAMPS
	__shared__ unsigned long long s_acc[16];
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
	if (tid < 16) s_acc[tid] = 0;
#define STAMP(i) do { __syncthreads(); if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_acc[i] += t_ - t_last; t_last = t_; } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
	const bool lds_pose = W.n_edges <= SSK_LM_LDS_EDGES;
	double* pose_cur = lds_pose ? s_pose : D.pose + 12 * (size_t)W.edge0;
	double* pose_new = lds_pose ? s_pose_new : D.pose_new + 12 * (size_t)W.edge0;

	for (int i = tid; i < 12 * W.n_edges; i += NT) {
		const double v = D.pose[12 * (size_t)W.edge0 + i];
		pose_cur[i] = v; pose_new[i] = v;
	}
	for (int i = tid; i < 3 * W.n_lms; i += NT) D.lm_new[3 * (size_t)W.lm0 + i] = D.lm[3 * (size_t)W.lm0 + i];
	__syncthreads();
	const double e0 = total_error(D, W, P, pose_cur, D.lm, s_part);
	if (tid == 0) {
		s_lambda = -1.0; s_nu = 2.0; s_err = e0; s_iters = 0; s_needlin = 1; s_accepted = 0; s_stop = 0;
		ssk_lm_result& R = D.res[blockIdx.x];
		R.err_init = e0; R.n_obs = W.n_obs; R.stop_reason = (W.E == 0 && W.L == 0) ? 3 : 0;
	}
	__syncthreads();
	if (W.E == 0 && W.L == 0) {
		if (tid == 0) { ssk_lm_result& R = D.res[blockIdx.x]; R.iters = 0; R.accepted = 0; R.err_final = e0; R.lambda_final = -1.0; }
		return;
	}

	while (true) {
		__syncthreads();
		if (s_iters >= P.max_iters) break;
		if (W.n_obs > 0 && sqrt(s_err / W.n_obs) < P.stop_rmse) { if (tid == 0) s_stop = 1; break; }
		STAMP(0);
		if (s_needlin) {
			// ---- A ----
			for (int o = tid; o < W.n_obs; o += NT) linearize_obs(D, W, P, pose_cur, o);
			__syncthreads();
			STAMP(1);
			// ---- B1: (landmark, edge) incidences ----
			for (int m = tid; m < W.n_lme; m += NT) {
				double Wm[18], ge[6];
#pragma unroll
				for (int c = 0; c < 18; ++c) Wm[c] = 0.0;
#pragma unroll
				for (int c = 0; c < 6; ++c) ge[c] = 0.0;
				const bool unk = D.lme_ls[W.lme0 + m] >= 0;
				const int k0 = D.lmep_off[W.lmeo0 + m], k1 = D.lmep_off[W.lmeo0 + m + 1];
				for (int k = k0; k < k1; ++k) {
					const int p = D.lmep_idx[W.lmep0 + k];
					const int o = D.path_obs[W.path0 + p];
					const double* Je = D.path_Je + 24 * (size_t)(W.path0 + p);
					const double* Jf = D.obs_Jf + 12 * (size_t)(W.obs0 + o);
					const double* r = D.obs_r + 4 * (size_t)(W.obs0 + o);
					double je[24], jf[12], rr[4];
#pragma unroll
					for (int c = 0; c < 24; ++c) je[c] = Je[c];
#pragma unroll
					for (int c = 0; c < 4; ++c) rr[c] = r[c];
#pragma unroll
					for (int a = 0; a < 6; ++a) ge[a] += je[a] * rr[0] + je[6 + a] * rr[1] + je[12 + a] * rr[2] + je[18 + a] * rr[3];
					if (unk) {
#pragma unroll
						for (int c = 0; c < 12; ++c) jf[c] = Jf[c];
#pragma unroll
						for (int a = 0; a < 6; ++a)
#pragma unroll
							for (int b = 0; b < 3; ++b)
								Wm[3 * a + b] += je[a] * jf[b] + je[6 + a] * jf[3 + b] + je[12 + a] * jf[6 + b] + je[18 + a] * jf[9 + b];
					}
				}
				double* Wo = D.Wb + 18 * (size_t)(W.lme0 + m);
#pragma unroll
				for (int c = 0; c < 18; ++c) Wo[c] = Wm[c];
				double* go = D.gpe + 6 * (size_t)(W.lme0 + m);
#pragma unroll
				for (int c = 0; c < 6; ++c) go[c] = ge[c];
			}
			STAMP(2);
			// ---- B2: optimised landmarks ----
			for (int l = tid; l < W.L; l += NT) {
				double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
				const int o0 = D.lmo_off[W.lso0 + l], o1 = D.lmo_off[W.lso0 + l + 1];
				for (int oi = o0; oi < o1; ++oi) {
					const int o = D.lmo_idx[W.lmo0 + oi];
					const double* Jf = D.obs_Jf + 12 * (size_t)(W.obs0 + o);
					const double* r = D.obs_r + 4 * (size_t)(W.obs0 + o);
					double jf[12], rr[4];
#pragma unroll
					for (int c = 0; c < 12; ++c) jf[c] = Jf[c];
#pragma unroll
					for (int c = 0; c < 4; ++c) rr[c] = r[c];
#pragma unroll
					for (int a = 0; a < 3; ++a) {
#pragma unroll
						for (int b = 0; b < 3; ++b) Hm[3 * a + b] += jf[a] * jf[b] + jf[3 + a] * jf[3 + b] + jf[6 + a] * jf[6 + b] + jf[9 + a] * jf[9 + b];
						g[a] += jf[a] * rr[0] + jf[3 + a] * rr[1] + jf[6 + a] * rr[2] + jf[9 + a] * rr[3];
					}
				}
#pragma unroll
				for (int c = 0; c < 9; ++c) D.Hff[9 * (size_t)(W.ls0 + l) + c] = Hm[c];
#pragma unroll
				for (int c = 0; c < 3; ++c) D.gf[3 * (size_t)(W.ls0 + l) + c] = g[c];
			}
			STAMP(3);
			// ---- D: H_pp blocks.  item = (landmark, edge pair) with co-observations: P = sum Je_i^T Je_j (6x6) in 36
			//      registers; items are sorted by block, so a wave sums a chunk of <= 64 items of ONE block with a fixed
			//      shuffle tree and lane 0 stores the 36-double partial; D2 adds a block's few partials in order ----
			for (int c = wave; c < W.n_pchunk; c += NW) {
				const int3 ch = D.pchunk[W.pc0 + c];          // (block pid, first item, count)
				double acc[36];
#pragma unroll
				for (int i = 0; i < 36; ++i) acc[i] = 0.0;
				if (lane < ch.z) {
					const int q = D.qp_q[W.qp0 + ch.y + lane];
					const int k0 = D.qo_off[W.qo0 + q], k1 = D.qo_off[W.qo0 + q + 1];
					for (int k = k0; k < k1; ++k) {
						const int2 pp = D.qo[W.qe0 + k];
						const double* J1 = D.path_Je + 24 * (size_t)(W.path0 + pp.x);
						const double* J2 = D.path_Je + 24 * (size_t)(W.path0 + pp.y);
						double j1[24], j2[24];
#pragma unroll
						for (int i = 0; i < 24; ++i) { j1[i] = J1[i]; j2[i] = J2[i]; }
#pragma unroll
						for (int a = 0; a < 6; ++a)
#pragma unroll
							for (int b = 0; b < 6; ++b)
								acc[6 * a + b] += j1[a] * j2[b] + j1[6 + a] * j2[6 + b] + j1[12 + a] * j2[12 + b] + j1[18 + a] * j2[18 + b];
					}
				}
				wave_sum36(acc, s_dyn + wave * RED_WAVE, lane, D.Ppart + 36 * (size_t)(W.pc0 + c));
			}
			__syncthreads();
			for (int it = tid; it < W.n_pairs * 36; it += NT) {
				const int pid = it / 36, ab = it - pid * 36;
				const int c0 = D.pc_off[W.pairo0 + pid], c1 = D.pc_off[W.pairo0 + pid + 1];
				double s = 0;
				for (int c = c0; c < c1; ++c) s += D.Ppart[36 * (size_t)(W.pc0 + c) + ab];
				D.Hpair[W.h0 + it] = s;
			}
			__syncthreads();
			STAMP(4);
			if (s_lambda < 0) {
				// lambda0 = tau * max diag(H)
				double mx = 0;
				for (int it = tid; it < n6; it += NT) {
					const int e = it / 6, a = it - 6 * e;
					const int pid = e * (e + 1) / 2 + e;
					mx = fmax(mx, D.Hpair[W.h0 + 36 * pid + 7 * a]);
				}
				for (int l = tid; l < W.L; l += NT)
					for (int a = 0; a < 3; ++a) mx = fmax(mx, D.Hff[9 * (size_t)(W.ls0 + l) + 4 * a]);
				mx = block_max(mx, s_part);
				if (tid == 0) s_lambda = P.tau * mx;
			}
			if (tid == 0) s_needlin = 0;
			__syncthreads();
		}
		STAMP(5);
		// ---- one attempt ----
		const double lambda = s_lambda;
		const int it_idx = s_iters;
		// ---- E ----
		for (int l = tid; l < W.L; l += NT) {
			double V[9], Vi[9];
#pragma unroll
			for (int c = 0; c < 9; ++c) V[c] = D.Hff[9 * (size_t)(W.ls0 + l) + c];
			V[0] += lambda; V[4] += lambda; V[8] += lambda;
			sym3_inv(V, Vi);
#pragma unroll
			for (int c = 0; c < 9; ++c) D.Vinv[9 * (size_t)(W.ls0 + l) + c] = Vi[c];
		}
		__syncthreads();
		// T = W V per incidence, and the incidence's reduced-gradient term gre = Je^T r - T g_f, stored edge-major
		// (position of the incidence in its edge's list) so the per-edge sums below stream contiguously
		for (int m = tid; m < W.n_lme; m += NT) {
			const int ls = D.lme_ls[W.lme0 + m];
			const double* ge = D.gpe + 6 * (size_t)(W.lme0 + m);
			double* gr = D.gre + 6 * (size_t)(W.el0 + D.lme_elpos[W.lme0 + m]);
			if (ls >= 0) {
				const double* Vi = D.Vinv + 9 * (size_t)(W.ls0 + ls);
				const double* Wm = D.Wb + 18 * (size_t)(W.lme0 + m);
				const double* gl = D.gf + 3 * (size_t)(W.ls0 + ls);
				double* Tm = D.Tb + 18 * (size_t)(W.lme0 + m);
				double v[9], w[18], g[3];
#pragma unroll
				for (int c = 0; c < 9; ++c) v[c] = Vi[c];
#pragma unroll
				for (int c = 0; c < 18; ++c) w[c] = Wm[c];
#pragma unroll
				for (int c = 0; c < 3; ++c) g[c] = gl[c];
#pragma unroll
				for (int a = 0; a < 6; ++a) {
					double t[3];
#pragma unroll
					for (int b = 0; b < 3; ++b) { t[b] = w[3 * a] * v[b] + w[3 * a + 1] * v[3 + b] + w[3 * a + 2] * v[6 + b]; Tm[3 * a + b] = t[b]; }
					gr[a] = ge[a] - (t[0] * g[0] + t[1] * g[1] + t[2] * g[2]);
				}
			} else {
#pragma unroll
				for (int a = 0; a < 6; ++a) gr[a] = ge[a];
			}
		}
		__syncthreads();
		STAMP(6);
		// ---- F: Schur term.  item = (optimised landmark, edge pair): T_i W_j^T (6x6) in registers; a wave sums a chunk
		//      of <= 64 items of ONE block with the fixed shuffle tree, lane 0 stores the partial ----
		for (int c = wave; c < W.n_schunk; c += NW) {
			const int3 ch = D.schunk[W.sc0 + c];              // (block pid, first q, count)
			double acc[36];
#pragma unroll
			for (int i = 0; i < 36; ++i) acc[i] = 0.0;
			if (lane < ch.z) {
				const int2 mm = D.q_mm[W.q0 + ch.y + lane];
				const double* T1 = D.Tb + 18 * (size_t)(W.lme0 + mm.x);
				const double* W2 = D.Wb + 18 * (size_t)(W.lme0 + mm.y);
				double t1[18], w2[18];
#pragma unroll
				for (int i = 0; i < 18; ++i) { t1[i] = T1[i]; w2[i] = W2[i]; }
#pragma unroll
				for (int a = 0; a < 6; ++a)
#pragma unroll
					for (int b = 0; b < 6; ++b)
						acc[6 * a + b] = t1[3 * a] * w2[3 * b] + t1[3 * a + 1] * w2[3 * b + 1] + t1[3 * a + 2] * w2[3 * b + 2];
			}
			wave_sum36(acc, s_dyn + wave * RED_WAVE, lane, D.Spart + 36 * (size_t)(W.sc0 + c));
		}
		__syncthreads();
		// ---- F2: H_red block (e,e') = H_pp + lambda I - its Schur partials in order; diagonal blocks are symmetrised
		//      from their upper triangle ((T W^T)(a,b) and (b,a) differ in the last bit: one writer) ----
		for (int it = tid; it < W.n_pairs * 36; it += NT) {
			const int pid = it / 36, ab = it - pid * 36, a = ab / 6, b = ab - a * 6;
			const int2 ee = D.pair_ee[W.pair0 + pid];
			if (ee.x == ee.y && a > b) continue;
			const int c0 = D.sc_off[W.pairo0 + pid], c1 = D.sc_off[W.pairo0 + pid + 1];
			double t = 0;
			for (int c = c0; c < c1; ++c) t += D.Spart[36 * (size_t)(W.sc0 + c) + ab];
			double sres = D.Hpair[W.h0 + it] - t;
			if (ee.x == ee.y && a == b) sres += lambda;
			A[(6 * ee.x + a) * np + 6 * ee.y + b] = sres;
			A[(6 * ee.y + b) * np + 6 * ee.x + a] = sres;
		}
		// g_red(e) = sum of its incidences' terms (contiguous, edge-major): 8 lanes per component, each a strided
		// partial, combined in lane order
		for (int it = tid; it < n6 * 8; it += NT) {
			const int comp = it >> 3, j = it & 7, e = comp / 6, a = comp - e * 6;
			const int k0 = D.el_off[W.uso0 + e], k1 = D.el_off[W.uso0 + e + 1];
			const double* src = D.gre + 6 * (size_t)W.el0 + a;
			double s0 = 0, s1 = 0;
			int k = k0 + j;
			for (; k + 8 < k1; k += 16) { s0 += src[6 * (size_t)k]; s1 += src[6 * (size_t)(k + 8)]; }
			if (k < k1) s0 += src[6 * (size_t)k];
			double v = s0 + s1;
			v += __shfl_down(v, 4, 8); v += __shfl_down(v, 2, 8); v += __shfl_down(v, 1, 8);
			if (j == 0) { if (lds_mat) A[n6 * np + comp] = v; else s_b[comp] = v; }
		}
		__syncthreads();
		STAMP(8);
		// ---- G ----
		bool ok = true;
		if (n6 > 0) {
			if (lds_mat) {
				ok = ldl_solve_lds(s_dyn, np, n6, s_b, s_inv);
			} else {
				ok = chol_solve_block(A, n6, s_b, &s_flag, &s_piv);
			}
		}
		STAMP(9);
		double dpn = 0, dfn = 0;
		if (ok) {
			for (int i = tid; i < n6; i += NT) { D.dp[6 * (size_t)W.us0 + i] = s_b[i]; dpn += s_b[i] * s_b[i]; }
			// ---- H: landmarks ----
			for (int l = tid; l < W.L; l += NT) {
				double rhs[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) rhs[c] = D.gf[3 * (size_t)(W.ls0 + l) + c];
				const int m0 = D.lsm_off[W.lso0 + l], m1 = D.lsm_off[W.lso0 + l + 1];
				for (int m = m0; m < m1; ++m) {
					const double* Wm = D.Wb + 18 * (size_t)(W.lme0 + m);
					const int e = D.lme_edge[W.lme0 + m];
#pragma unroll
					for (int a = 0; a < 6; ++a) {
						const double d = s_b[6 * e + a];
						rhs[0] -= Wm[3 * a] * d; rhs[1] -= Wm[3 * a + 1] * d; rhs[2] -= Wm[3 * a + 2] * d;
					}
				}
				const double* Vi = D.Vinv + 9 * (size_t)(W.ls0 + l);
				const int li = D.slot_lm[W.ls0 + l];
#pragma unroll
				for (int a = 0; a < 3; ++a) {
					const double d = Vi[3 * a] * rhs[0] + Vi[3 * a + 1] * rhs[1] + Vi[3 * a + 2] * rhs[2];
					D.df[3 * (size_t)(W.ls0 + l) + a] = d;
					dfn += d * d;
					D.lm_new[3 * (size_t)(W.lm0 + li) + a] = D.lm[3 * (size_t)(W.lm0 + li) + a] + d;
				}
			}
			// ---- H: edges, T_new = exp(eps) (+) T ----
			for (int e = tid; e < W.E; e += NT) {
				const int ei = D.slot_edge[W.us0 + e];
				const double* T = pose_cur + 12 * ei;
				double* O = pose_new + 12 * ei;
				double eps[6], R[9];
#pragma unroll
				for (int a = 0; a < 6; ++a) eps[a] = s_b[6 * e + a];
				so3_exp(eps + 3, R);
#pragma unroll
				for (int i = 0; i < 3; ++i) {
#pragma unroll
					for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
					O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + eps[i];
				}
			}
			if (D.trace_on)
				for (int i = tid; i < n6; i += NT) D.dp_trace[W.dp_trace0 + (size_t)it_idx * n6 + i] = s_b[i];
		}
		dpn = block_sum(dpn, s_part);
		dfn = block_sum(dfn, s_part);
		if (ok && D.trace_on)
			for (int i = tid; i < 3 * W.L; i += NT) D.df_trace[W.df_trace0 + (size_t)it_idx * 3 * W.L + i] = D.df[3 * (size_t)W.ls0 + i];
		STAMP(10);
		// ---- I ----
		double err_new = s_err;
		if (ok) err_new = total_error(D, W, P, pose_new, D.lm_new, s_part);
		__syncthreads();
		if (tid == 0) {
			const int accepted = ok && err_new < s_err;
			if (D.trace_on) {
				ssk_lm_trace_entry& t = D.trace[W.trace0 + it_idx];
				t.lambda = lambda; t.err_before = s_err; t.err_after = err_new; t.delta_p_norm = sqrt(dpn);
				t.delta_f_norm = sqrt(dfn); t.accepted = accepted; t.cholesky_ok = ok;
			}
			const int converged = ok && P.conv_rel > 0 && fabs(s_err - err_new) <= P.conv_rel * s_err;
			s_accept = accepted;
			s_iters = it_idx + 1;
			if (accepted) {
				s_err = err_new; s_lambda = lambda * 0.1; s_nu = 2.0; s_needlin = 1; s_accepted += 1;
				if (converged) s_stop = 4;
			} else if (converged) {
				s_stop = 4;
			} else {
				s_lambda = lambda * s_nu; s_nu *= 2.0;
				if (s_lambda > P.max_lambda) s_stop = 2;
			}
		}
		__syncthreads();
		if (s_accept) {
			for (int e = tid; e < W.E; e += NT) {
				const int ei = D.slot_edge[W.us0 + e];
				for (int c = 0; c < 12; ++c) pose_cur[12 * ei + c] = pose_new[12 * ei + c];
			}
			for (int l = tid; l < W.L; l += NT) {
				const int li = D.slot_lm[W.ls0 + l];
				for (int c = 0; c < 3; ++c) D.lm[3 * (size_t)(W.lm0 + li) + c] = D.lm_new[3 * (size_t)(W.lm0 + li) + c];
			}
		}
		__syncthreads();
		STAMP(11);
		if (s_stop == 2 || s_stop == 4) break;
	}
	__syncthreads();
	// publish the final edge poses (they lived in LDS during the solve)
	if (lds_pose)
		for (int e = tid; e < W.E; e += NT) {
			const int ei = D.slot_edge[W.us0 + e];
			for (int c = 0; c < 12; ++c) D.pose[12 * (size_t)(W.edge0 + ei) + c] = pose_cur[12 * ei + c];
		}
	if (tid == 0) {
		ssk_lm_result& R = D.res[blockIdx.x];
		R.iters = s_iters; R.accepted = s_accepted; R.stop_reason = s_stop; R.err_final = s_err; R.lambda_final = s_lambda;
#ifdef SSK_LM_STAMPS
		if (blockIdx.x == 0)
			printf("LM stamps (cycles, window 0, %d iters): top %llu A %llu B1 %llu B2 %llu D %llu lam %llu E %llu F1 %llu F2 %llu G %llu H %llu I+dec %llu\n", s_iters,
			       s_acc[0], s_acc[1], s_acc[2], s_acc[3], s_acc[4], s_acc[5], s_acc[6], s_acc[7], s_acc[8], s_acc[9], s_acc[10], s_acc[11]);
#endif
	}
}

__global__ void lm_reset_kernel(double* pose, const double* pose0, size_t n_pose, double* lm, const double* lm0, size_t n_lm)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t k = i; k < n_pose; k += stride) pose[k] = pose0[k];
	for (size_t k = i; k < n_lm; k += stride) lm[k] = lm0[k];
}

}  // namespace

int ssk_lm_launch(ssk_ctx* ctx, const lm_dev& D, const lm_cam& P, int n_windows, int max_n6)
{
	// Cholesky matrix in LDS when the largest reduced system of the batch fits (gfx950: 160 KiB of LDS per CU;
	// one window per workgroup, ~32 KiB of static LDS besides)
	int cap = 0;
	size_t dyn = 0;
	const size_t LDS_BUDGET = 120 * 1024;
	const size_t need = ((size_t)max_n6 + 1) * (size_t)(max_n6 | 1) * sizeof(double);   // + the right-hand-side border row
	const size_t red_bytes = (size_t)NW * RED_WAVE * sizeof(double);      // wave_sum36 scratch, aliased with the matrix
	dyn = red_bytes;
	if (need <= LDS_BUDGET && max_n6 <= 127) { cap = max_n6; dyn = need > red_bytes ? need : red_bytes; }
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	hipLaunchKernelGGL(lm_solve_kernel, dim3(n_windows), dim3(NT), dyn, ctx->stream, D, P, cap);
	SSK_LAUNCH_CHECK(ctx, "lm_solve_kernel");
	return SSK_OK;
}

int ssk_lm_launch_reset(ssk_ctx* ctx, double* pose, const double* pose0, size_t n_pose, double* lm, const double* lm0, size_t n_lm)
{
	hipLaunchKernelGGL(lm_reset_kernel, dim3(256), dim3(256), 0, ctx->stream, pose, pose0, n_pose, lm, lm0, n_lm);
	SSK_LAUNCH_CHECK(ctx, "lm_reset_kernel");
	return SSK_OK;
}
