// ssk_lm.hip -- relative bundle adjustment on the GPU: Levenberg-Marquardt with Schur complement on the
// landmarks and dense Cholesky of the reduced pose system, for SRBA's problem types
// (kf2kf SE(3) edges, Euclidean3D landmarks, 4-D stereo observations, sensor pose on robot;
// reference SRBA.h:30-54, solver_LM_schur_dense_cholesky SRBA.h:34), entered in the reference through
// rba.define_new_keyframe (EST.cpp:166,782).  Contract L1-L5 of DESIGN.md "LM contract".
//
// Mapping (gfx950): ONE 256-thread workgroup per optimisation window, persistent over the whole LM loop
// (a keyframe's window is ~10 edges / ~2.5k observations / ~5 Mflop per iteration: launch- and
// latency-bound, so the win is zero launches inside the loop and a batch of windows = a grid of workgroups,
// one per CU).  Per iteration, inside the workgroup, separated by workgroup barriers:
//   A  thread-per-observation: SE(3) chain along the spanning-tree path, stereo projection, residual,
//      IRLS weight, weighted Jacobian blocks 4x3 (landmark) and 4x6 (each optimised edge on the path)
//      -> per-window scratch in HBM/L2 (never atomics);
//   B  thread-per-landmark: H_ff (3x3), g_f, W = H_pf blocks (6x3 per touched edge), fixed summation order;
//   D  thread-per-entry of the upper block triangle of H_pp (from precomputed observation pair lists), g_p;
//   E  thread-per-landmark: V = (H_ff + lambda I)^-1, T = W V;
//   F  thread-per-entry: H_red = H_pp + lambda I - sum_l T W^T (precomputed landmark pair lists), g_red;
//   G  dense Cholesky LL^T of H_red (LDS when it fits, else L2-resident scratch) + triangular solves;
//   H  back-substitution for the landmarks, trial state (Rodrigues update of the edges);
//   I  total (robustified) error of the trial state, accept / reject, lambda schedule.
// Every sum runs in an order fixed by the symbolic structure (no floating-point atomics), so a solve is
// reproducible run to run.  All arithmetic FP64.  Bound: latency (dependent L2 accesses + barriers).
#include "ssk_lm.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void se3_apply(const double* __restrict__ T, const double* p, double* q)
{
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * p[0] + T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2] + T[4 * i + 3];
}
__device__ __forceinline__ void se3_apply_inv(const double* __restrict__ T, const double* p, double* q)
{
	const double d0 = p[0] - T[3], d1 = p[1] - T[7], d2 = p[2] - T[11];
#pragma unroll
	for (int i = 0; i < 3; ++i) q[i] = T[i] * d0 + T[4 + i] * d1 + T[8 + i] * d2;
}
__device__ __forceinline__ double huber_rho(double s2, double k) { return 2.0 * k * k * (sqrt(1.0 + s2 / (k * k)) - 1.0); }

__device__ __forceinline__ void project(const lm_cam& P, const double* pc, double* h)
{
	const double iz = 1.0 / pc[2];
	h[0] = P.cx_l + P.fx_l * pc[0] * iz;
	h[1] = P.cy_l + P.fy_l * pc[1] * iz;
	h[2] = P.cx_r + P.fx_r * (pc[0] - P.baseline) * iz;
	h[3] = P.cy_r + P.fy_r * pc[1] * iz;
}

// robustified squared error of observation o (window relative) for the given state arrays (batch wide)
__device__ double obs_error(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* __restrict__ pose,
                            const double* __restrict__ lm, int o)
{
	double q[3], t[3];
	const double* f = lm + 3 * (size_t)(W.lm0 + D.obs_lm[W.obs0 + o]);
	q[0] = f[0]; q[1] = f[1]; q[2] = f[2];
	const int pb = D.obs_path_off[W.opo0 + o], pe = D.obs_path_off[W.opo0 + o + 1];
	for (int i = pe - 1; i >= pb; --i) {
		const int c = D.path_edge[W.path0 + i];
		const double* T = pose + 12 * (size_t)(W.edge0 + (c >> 1));
		if ((c & 1) == 0) se3_apply(T, q, t); else se3_apply_inv(T, q, t);
		q[0] = t[0]; q[1] = t[1]; q[2] = t[2];
	}
	double pc[3], h[4];
	se3_apply_inv(P.S, q, pc);
	project(P, pc, h);
	const float* z = D.obs_px + 4 * (size_t)(W.obs0 + o);
	double s2 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { const double r = (double)z[k] - h[k]; s2 += r * r; }
	return P.robust ? huber_rho(s2, P.kernel) : s2;
}

// deterministic workgroup sum: thread t holds a partial over its strided items; fixed LDS tree
__device__ double block_sum(double v, double* s_red)
{
	const int tid = threadIdx.x;
	__syncthreads();
	s_red[tid] = v;
	__syncthreads();
	for (int o = NT / 2; o > 0; o >>= 1) {
		if (tid < o) s_red[tid] += s_red[tid + o];
		__syncthreads();
	}
	const double r = s_red[0];
	__syncthreads();
	return r;
}
__device__ double block_max(double v, double* s_red)
{
	const int tid = threadIdx.x;
	__syncthreads();
	s_red[tid] = v;
	__syncthreads();
	for (int o = NT / 2; o > 0; o >>= 1) {
		if (tid < o) s_red[tid] = fmax(s_red[tid], s_red[tid + o]);
		__syncthreads();
	}
	const double r = s_red[0];
	__syncthreads();
	return r;
}

__device__ double total_error(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* pose, const double* lm, double* s_red)
{
	double e = 0;
	for (int o = threadIdx.x; o < W.n_obs; o += NT) e += obs_error(D, W, P, pose, lm, o);
	return block_sum(e, s_red);
}

// phase A: linearise observation o at the current state
__device__ void linearize_obs(const lm_dev& D, const lm_win& W, const lm_cam& P, int o)
{
	const int pb = D.obs_path_off[W.opo0 + o], n = D.obs_path_off[W.opo0 + o + 1] - pb;
	const double* __restrict__ pose = D.pose;
	double q[3], t[3];
	{
		const double* f = D.lm + 3 * (size_t)(W.lm0 + D.obs_lm[W.obs0 + o]);
		q[0] = f[0]; q[1] = f[1]; q[2] = f[2];
	}
	// backward pass: q_i = X_i (+) q_{i+1}; path_q[entry i] keeps the point a Jacobian of entry i needs:
	// dir 0 -> q_i (after the step), dir 1 -> q_{i+1} (before it)
	for (int i = n - 1; i >= 0; --i) {
		const int c = D.path_edge[W.path0 + pb + i];
		const double* T = pose + 12 * (size_t)(W.edge0 + (c >> 1));
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
		const double* T = pose + 12 * (size_t)(W.edge0 + e);
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

__device__ void so3_exp(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
	const double th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double K2[9];
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
#pragma unroll
	for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
	R[0] += 1; R[4] += 1; R[8] += 1;
}

// dense Cholesky (lower, in place) + solve A x = b; A is n x n row-major (LDS or global), b in LDS.
// returns false (uniformly) on a non-positive pivot.
__device__ bool chol_solve(double* A, int n, double* b, int* s_flag, double* s_piv)
{
	const int tid = threadIdx.x;
	for (int j = 0; j < n; ++j) {
		if (tid == 0) {
			const double d = A[(size_t)j * n + j];
			*s_flag = d > 0.0;
			*s_piv = sqrt(d);
			if (d > 0.0) A[(size_t)j * n + j] = *s_piv;
		}
		__syncthreads();
		if (!*s_flag) { __syncthreads(); return false; }
		const double inv = 1.0 / *s_piv;
		const int m = n - j - 1;
		for (int i = j + 1 + tid; i < n; i += NT) A[(size_t)i * n + j] = A[(size_t)i * n + j] * inv;
		__syncthreads();
		for (int t = tid; t < m * m; t += NT) {
			const int i = j + 1 + t / m, k = j + 1 + t % m;
			if (k <= i) A[(size_t)i * n + k] -= A[(size_t)i * n + j] * A[(size_t)k * n + j];
		}
		__syncthreads();
	}
	// forward substitution L y = b (column oriented)
	for (int i = 0; i < n; ++i) {
		if (tid == 0) b[i] = b[i] / A[(size_t)i * n + i];
		__syncthreads();
		const double yi = b[i];
		for (int k = i + 1 + tid; k < n; k += NT) b[k] -= A[(size_t)k * n + i] * yi;
		__syncthreads();
	}
	// backward substitution L^T x = y
	for (int i = n - 1; i >= 0; --i) {
		if (tid == 0) b[i] = b[i] / A[(size_t)i * n + i];
		__syncthreads();
		const double xi = b[i];
		for (int k = tid; k < i; k += NT) b[k] -= A[(size_t)i * n + k] * xi;
		__syncthreads();
	}
	return true;
}

__global__ __launch_bounds__(NT)
void lm_solve_kernel(lm_dev D, lm_cam P, int lds_n6_cap)
{
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [lds_n6_cap^2] Cholesky workspace when it fits
	__shared__ double s_red[NT];
	__shared__ double s_b[SSK_LM_MAX_N6];
	__shared__ double s_lambda, s_nu, s_err, s_piv;
	__shared__ int s_flag, s_accept, s_stop, s_iters, s_needlin, s_accepted;

	const lm_win W = D.win[blockIdx.x];
	const int tid = threadIdx.x;
	const int n6 = 6 * W.E;
	double* A = (n6 <= lds_n6_cap) ? s_dyn : (D.Hred + W.h0);

	// trial state starts as a copy of the current one (fixed entries never change)
	for (int i = tid; i < 12 * W.n_edges; i += NT) D.pose_new[12 * (size_t)W.edge0 + i] = D.pose[12 * (size_t)W.edge0 + i];
	for (int i = tid; i < 3 * W.n_lms; i += NT) D.lm_new[3 * (size_t)W.lm0 + i] = D.lm[3 * (size_t)W.lm0 + i];
	const double e0 = total_error(D, W, P, D.pose, D.lm, s_red);
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
		if (s_needlin) {
			// A
			for (int o = tid; o < W.n_obs; o += NT) linearize_obs(D, W, P, o);
			__syncthreads();
			// B
			for (int l = tid; l < W.L; l += NT) {
				double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
				const int m0 = D.lme_off[W.lso0 + l], m1 = D.lme_off[W.lso0 + l + 1];
				for (int m = m0; m < m1; ++m)
					for (int c = 0; c < 18; ++c) D.Wb[18 * (size_t)(W.lme0 + m) + c] = 0.0;
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
						for (int b = 0; b < 3; ++b) {
							double s = 0;
#pragma unroll
							for (int k = 0; k < 4; ++k) s += jf[3 * k + a] * jf[3 * k + b];
							H[3 * a + b] += s;
						}
						double s = 0;
#pragma unroll
						for (int k = 0; k < 4; ++k) s += jf[3 * k + a] * rr[k];
						g[a] += s;
					}
					const int pb = D.obs_path_off[W.opo0 + o], pe = D.obs_path_off[W.opo0 + o + 1];
					for (int p = pb; p < pe; ++p) {
						const int m = D.path_lme[W.path0 + p];
						if (m < 0) continue;
						const double* Je = D.path_Je + 24 * (size_t)(W.path0 + p);
						double* Wm = D.Wb + 18 * (size_t)(W.lme0 + m);
						for (int a = 0; a < 6; ++a)
#pragma unroll
							for (int b = 0; b < 3; ++b) {
								double s = 0;
#pragma unroll
								for (int k = 0; k < 4; ++k) s += Je[6 * k + a] * jf[3 * k + b];
								Wm[3 * a + b] += s;
							}
					}
				}
#pragma unroll
				for (int c = 0; c < 9; ++c) D.Hff[9 * (size_t)(W.ls0 + l) + c] = H[c];
#pragma unroll
				for (int c = 0; c < 3; ++c) D.gf[3 * (size_t)(W.ls0 + l) + c] = g[c];
			}
			// D: H_pp upper block pairs and g_p
			for (int it = tid; it < W.n_pairs * 36; it += NT) {
				const int pid = it / 36, ab = it - pid * 36, a = ab / 6, b = ab - a * 6;
				const int2 ee = D.pair_ee[W.pair0 + pid];
				if (ee.x == ee.y && a > b) continue;      // diagonal block: one writer per symmetric entry pair
				const int k0 = D.ppo_off[W.pairo0 + pid], k1 = D.ppo_off[W.pairo0 + pid + 1];
				double s = 0;
				for (int k = k0; k < k1; ++k) {
					const int2 pp = D.ppo[W.ppo0 + k];
					const double* J1 = D.path_Je + 24 * (size_t)(W.path0 + pp.x);
					const double* J2 = D.path_Je + 24 * (size_t)(W.path0 + pp.y);
					s += J1[a] * J2[b] + J1[6 + a] * J2[6 + b] + J1[12 + a] * J2[12 + b] + J1[18 + a] * J2[18 + b];
				}
				D.Hpp[W.h0 + (size_t)(6 * ee.x + a) * n6 + 6 * ee.y + b] = s;
				D.Hpp[W.h0 + (size_t)(6 * ee.y + b) * n6 + 6 * ee.x + a] = s;
			}
			for (int it = tid; it < n6; it += NT) {
				const int e = it / 6, a = it - e * 6;
				const int k0 = D.pe_off[W.uso0 + e], k1 = D.pe_off[W.uso0 + e + 1];
				double s = 0;
				for (int k = k0; k < k1; ++k) {
					const int p = D.pe_idx[W.pe0 + k];
					const double* J = D.path_Je + 24 * (size_t)(W.path0 + p);
					const double* r = D.obs_r + 4 * (size_t)(W.obs0 + D.path_obs[W.path0 + p]);
					s += J[a] * r[0] + J[6 + a] * r[1] + J[12 + a] * r[2] + J[18 + a] * r[3];
				}
				D.gp[6 * (size_t)W.us0 + it] = s;
			}
			__syncthreads();
			if (s_lambda < 0) {
				double mx = 0;
				for (int i = tid; i < n6; i += NT) mx = fmax(mx, D.Hpp[W.h0 + (size_t)i * n6 + i]);
				for (int l = tid; l < W.L; l += NT)
					for (int a = 0; a < 3; ++a) mx = fmax(mx, D.Hff[9 * (size_t)(W.ls0 + l) + 4 * a]);
				mx = block_max(mx, s_red);
				if (tid == 0) s_lambda = P.tau * mx;
			}
			if (tid == 0) s_needlin = 0;
			__syncthreads();
		}
		// ---- one attempt ----
		const double lambda = s_lambda;
		const int it_idx = s_iters;
		// E
		for (int l = tid; l < W.L; l += NT) {
			double V[9], Vi[9];
#pragma unroll
			for (int c = 0; c < 9; ++c) V[c] = D.Hff[9 * (size_t)(W.ls0 + l) + c];
			V[0] += lambda; V[4] += lambda; V[8] += lambda;
			sym3_inv(V, Vi);
#pragma unroll
			for (int c = 0; c < 9; ++c) D.Vinv[9 * (size_t)(W.ls0 + l) + c] = Vi[c];
			const int m0 = D.lme_off[W.lso0 + l], m1 = D.lme_off[W.lso0 + l + 1];
			for (int m = m0; m < m1; ++m) {
				const double* Wm = D.Wb + 18 * (size_t)(W.lme0 + m);
				double* Tm = D.Tb + 18 * (size_t)(W.lme0 + m);
				for (int a = 0; a < 6; ++a)
#pragma unroll
					for (int b = 0; b < 3; ++b)
						Tm[3 * a + b] = Wm[3 * a] * Vi[b] + Wm[3 * a + 1] * Vi[3 + b] + Wm[3 * a + 2] * Vi[6 + b];
			}
		}
		__syncthreads();
		// F
		for (int it = tid; it < W.n_pairs * 36; it += NT) {
			const int pid = it / 36, ab = it - pid * 36, a = ab / 6, b = ab - a * 6;
			const int2 ee = D.pair_ee[W.pair0 + pid];
			if (ee.x == ee.y && a > b) continue;          // (T W^T)(a,b) and (b,a) differ in the last bit: one writer
			const int k0 = D.ppl_off[W.pairo0 + pid], k1 = D.ppl_off[W.pairo0 + pid + 1];
			double s = D.Hpp[W.h0 + (size_t)(6 * ee.x + a) * n6 + 6 * ee.y + b];
			for (int k = k0; k < k1; ++k) {
				const int2 mm = D.ppl[W.ppl0 + k];
				const double* T1 = D.Tb + 18 * (size_t)(W.lme0 + mm.x) + 3 * a;
				const double* W2 = D.Wb + 18 * (size_t)(W.lme0 + mm.y) + 3 * b;
				s -= T1[0] * W2[0] + T1[1] * W2[1] + T1[2] * W2[2];
			}
			if (ee.x == ee.y && a == b) s += lambda;
			A[(size_t)(6 * ee.x + a) * n6 + 6 * ee.y + b] = s;
			A[(size_t)(6 * ee.y + b) * n6 + 6 * ee.x + a] = s;
		}
		for (int it = tid; it < n6; it += NT) {
			const int e = it / 6, a = it - e * 6;
			const int k0 = D.el_off[W.uso0 + e], k1 = D.el_off[W.uso0 + e + 1];
			double s = D.gp[6 * (size_t)W.us0 + it];
			for (int k = k0; k < k1; ++k) {
				const int m = D.el_idx[W.el0 + k];
				const double* T = D.Tb + 18 * (size_t)(W.lme0 + m) + 3 * a;
				const double* g = D.gf + 3 * (size_t)(W.ls0 + D.lme_lm[W.lme0 + m]);
				s -= T[0] * g[0] + T[1] * g[1] + T[2] * g[2];
			}
			s_b[it] = s;
		}
		__syncthreads();
		// G
		const bool ok = n6 == 0 ? true : chol_solve(A, n6, s_b, &s_flag, &s_piv);
		double dpn = 0, dfn = 0;
		if (ok) {
			for (int i = tid; i < n6; i += NT) { D.dp[6 * (size_t)W.us0 + i] = s_b[i]; dpn += s_b[i] * s_b[i]; }
			// H: landmarks
			for (int l = tid; l < W.L; l += NT) {
				double rhs[3];
#pragma unroll
				for (int c = 0; c < 3; ++c) rhs[c] = D.gf[3 * (size_t)(W.ls0 + l) + c];
				const int m0 = D.lme_off[W.lso0 + l], m1 = D.lme_off[W.lso0 + l + 1];
				for (int m = m0; m < m1; ++m) {
					const double* Wm = D.Wb + 18 * (size_t)(W.lme0 + m);
					const int e = D.lme_edge[W.lme0 + m];
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
			// H: edges, T_new = exp(eps) (+) T
			for (int e = tid; e < W.E; e += NT) {
				const int ei = D.slot_edge[W.us0 + e];
				const double* T = D.pose + 12 * (size_t)(W.edge0 + ei);
				double* O = D.pose_new + 12 * (size_t)(W.edge0 + ei);
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
			if (D.trace_on) {
				for (int i = tid; i < n6; i += NT) D.dp_trace[W.dp_trace0 + (size_t)it_idx * n6 + i] = s_b[i];
			}
		}
		dpn = block_sum(dpn, s_red);
		dfn = block_sum(dfn, s_red);
		if (ok && D.trace_on)
			for (int i = tid; i < 3 * W.L; i += NT) D.df_trace[W.df_trace0 + (size_t)it_idx * 3 * W.L + i] = D.df[3 * (size_t)W.ls0 + i];
		// I
		double err_new = s_err;
		if (ok) err_new = total_error(D, W, P, D.pose_new, D.lm_new, s_red);
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
				for (int c = 0; c < 12; ++c) D.pose[12 * (size_t)(W.edge0 + ei) + c] = D.pose_new[12 * (size_t)(W.edge0 + ei) + c];
			}
			for (int l = tid; l < W.L; l += NT) {
				const int li = D.slot_lm[W.ls0 + l];
				for (int c = 0; c < 3; ++c) D.lm[3 * (size_t)(W.lm0 + li) + c] = D.lm_new[3 * (size_t)(W.lm0 + li) + c];
			}
		} else {
			// the trial state must again equal the current one for the next attempt's untouched entries
			for (int l = tid; l < W.L; l += NT) {
				const int li = D.slot_lm[W.ls0 + l];
				for (int c = 0; c < 3; ++c) D.lm_new[3 * (size_t)(W.lm0 + li) + c] = D.lm[3 * (size_t)(W.lm0 + li) + c];
			}
		}
		__syncthreads();
		if (s_stop == 2 || s_stop == 4) break;
	}
	__syncthreads();
	if (tid == 0) {
		ssk_lm_result& R = D.res[blockIdx.x];
		R.iters = s_iters; R.accepted = s_accepted; R.stop_reason = s_stop; R.err_final = s_err; R.lambda_final = s_lambda;
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
	// Cholesky workspace in LDS when the largest reduced system of the batch fits in 64 KiB
	// (gfx950 has 160 KiB of LDS per CU; one window per CU, so a workgroup may take most of it)
	int cap = 0;
	size_t dyn = 0;
	const size_t LDS_BUDGET = 140 * 1024;
	if ((size_t)max_n6 * max_n6 * sizeof(double) <= LDS_BUDGET) { cap = max_n6; dyn = (size_t)max_n6 * max_n6 * sizeof(double); }
	static bool attr_set = false;
	if (!attr_set) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
		attr_set = true;
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
