// ssk_lm_pose.hip -- the POSE-ONLY Levenberg-Marquardt solve: windows that optimise one or two k2k edges and NO landmark.  This is stage 1 of
// rba.define_new_keyframe (EST.cpp:782-784: optimize_edges on the keyframe's new edge(s) with every landmark held, before the local area is
// optimised) -- once per keyframe, on ~300 observations.  Same contract L1-L5 as every LM kernel (oracle/oracle_lm.c).
//
// Why a kernel of its own.  With the landmarks held there is no Schur complement: H = sum J^T J is 6 x 6 (12 x 12 for two edges), g = sum J^T r.
// The general node kernel (ssk_lm_node.hip) spends 21 us per LM attempt on such a window -- group phase, a pair phase whose single block keeps
// 8 lanes busy, a tile solver with barriers per 4 columns, table rebuilds -- of which nothing is arithmetic.  Here: one 128-thread workgroup per
// window, a thread per observation in turn (chain of the path's edge poses -> stereo projection -> the 4 x 6 Jacobian block per optimised path
// entry, registers only), the lower triangle of H and g summed by a fixed shuffle tree + the waves' partials in wave order (bit-reproducible),
// the 6E x 6E system factorised (L D L^T, positive pivots as the contract's Cholesky) by one thread in registers, the trial error by all threads.
// A rejected attempt keeps the linearisation (the state has not moved: the general kernels recompute the same numbers).
// The batch builder (ssk_capi_lm.hip) needs no node tables for these windows: the kernel reads the window's raw arrays.
#include "ssk_lm.h"

#pragma clang fp contract(fast)

namespace {

#ifndef SSK_PO_NT
#define SSK_PO_NT 128                   // (1 024 stage-1 windows: 0.68 ms at 256 threads, 0.39 at 128, 0.49 at 64; one window: the same 0.30 ms)
#endif
#define SSK_LM_NT SSK_PO_NT
#include "ssk_lm_device.h"

// one observation at the current state: weighted residual and the weighted 4 x 6 Jacobian block of each optimised edge on its path
// (slot 0 / 1; absent: zero).  The algebra of linearize_obs (ssk_lm.hip), kept in registers.
template <int E>
__device__ __forceinline__ void po_linearise_obs(const lm_dev& D, const lm_win& W, const lm_cam& P, const double* pose, int o, double (&r)[4], double (&J)[E][24], bool (&on)[E])
{
	const int pb = D.obs_path_off[W.opo0 + o], n = D.obs_path_off[W.opo0 + o + 1] - pb;
	const int* pe = D.path_edge + W.path0 + pb;
	double q[3], t[3];
	{
		const double* f = D.lm + 3 * (size_t)(W.lm0 + D.obs_lm[W.obs0 + o]);
		q[0] = f[0]; q[1] = f[1]; q[2] = f[2];
	}
	// backward pass: the point each optimised entry's Jacobian needs (dir 0: after the step, dir 1: before it)
	double qe[E][3];
	int ie[E];
#pragma unroll
	for (int s = 0; s < E; ++s) { on[s] = false; ie[s] = -1; qe[s][0] = qe[s][1] = qe[s][2] = 0.0; }
	for (int i = n - 1; i >= 0; --i) {
		const int c = pe[i];
		const double* T = pose + 12 * (c >> 1);
		const int slot = D.edge_slot[W.edge0 + (c >> 1)];
		if ((c & 1) == 0) se3_apply(T, q, t); else se3_apply_inv(T, q, t);
#pragma unroll
		for (int s = 0; s < E; ++s)
			if (slot == s) {
				on[s] = true; ie[s] = i;
				const double* src = (c & 1) == 0 ? t : q;
				qe[s][0] = src[0]; qe[s][1] = src[1]; qe[s][2] = src[2];
			}
		q[0] = t[0]; q[1] = t[1]; q[2] = t[2];
	}
	double pc[3], h[4];
	se3_apply_inv(P.S, q, pc);
	project(P, pc, h);
	const float* z = D.obs_px + 4 * (size_t)(W.obs0 + o);
	double s2 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { r[k] = (double)z[k] - h[k]; s2 += r[k] * r[k]; }
	const double w = P.robust ? sqrt(1.0 / sqrt(1.0 + s2 / (P.kernel * P.kernel))) : 1.0;
#pragma unroll
	for (int k = 0; k < 4; ++k) r[k] *= w;
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
#pragma unroll
	for (int s = 0; s < E; ++s)
#pragma unroll
		for (int k = 0; k < 24; ++k) J[s][k] = 0.0;
	// forward pass: RA = product of the step rotations so far
	double RA[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int i = 0; i < n; ++i) {
		const int c = pe[i];
		const int dir = c & 1;
		const double* T = pose + 12 * (c >> 1);
		double RN[9];
#pragma unroll
		for (int a = 0; a < 3; ++a)
#pragma unroll
			for (int b = 0; b < 3; ++b)
				RN[3 * a + b] = dir == 0 ? RA[3 * a] * T[b] + RA[3 * a + 1] * T[4 + b] + RA[3 * a + 2] * T[8 + b]
				                         : RA[3 * a] * T[4 * b] + RA[3 * a + 1] * T[4 * b + 1] + RA[3 * a + 2] * T[4 * b + 2];
#pragma unroll
		for (int s = 0; s < E; ++s)
			if (ie[s] == i) {
				const double* RR = dir == 0 ? RA : RN;
				const double sg = dir == 0 ? w : -w;
#pragma unroll
				for (int a = 0; a < 4; ++a) {
					const double A0 = sg * (M[3 * a] * RR[0] + M[3 * a + 1] * RR[3] + M[3 * a + 2] * RR[6]);
					const double A1 = sg * (M[3 * a] * RR[1] + M[3 * a + 1] * RR[4] + M[3 * a + 2] * RR[7]);
					const double A2 = sg * (M[3 * a] * RR[2] + M[3 * a + 1] * RR[5] + M[3 * a + 2] * RR[8]);
					J[s][6 * a] = A0; J[s][6 * a + 1] = A1; J[s][6 * a + 2] = A2;
					J[s][6 * a + 3] = -A1 * qe[s][2] + A2 * qe[s][1];
					J[s][6 * a + 4] = A0 * qe[s][2] - A2 * qe[s][0];
					J[s][6 * a + 5] = -A0 * qe[s][1] + A1 * qe[s][0];
				}
			}
#pragma unroll
		for (int a = 0; a < 9; ++a) RA[a] = RN[a];
	}
}

template <int E>
__device__ void po_solve(lm_dev D, const int* __restrict__ order, lm_cam P)
{
	constexpr int N6 = 6 * E, NH = N6 * (N6 + 1) / 2, NA = NH + N6;       // lower triangle of H (row-major: (i, j), j <= i, at i (i + 1) / 2 + j), then g
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];          // [pose_cur | pose_new]
	__shared__ double s_part[NW];
	__shared__ double s_red[NW][NA];
	__shared__ double s_H[NA];
	__shared__ double s_x[N6];
	__shared__ double s_lambda, s_nu, s_err;
	__shared__ int s_iters, s_accepted, s_stop, s_ok, s_accept, s_fresh;
	const int wi = order[blockIdx.x];
	const lm_win W = D.win[wi];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double* pose_cur = s_dyn;
	double* pose_new = s_dyn + 12 * W.n_edges;
	for (int i = tid; i < 12 * W.n_edges; i += NT) {
		const double v = D.pose[12 * (size_t)W.edge0 + i];
		pose_cur[i] = v; pose_new[i] = v;
	}
	__syncthreads();
	auto error_of = [&](const double* pose) {
		double e = 0.0;
		for (int o = tid; o < W.n_obs; o += NT) e += obs_error(D, W, P, pose, D.lm, o);
		return block_sum(e, s_part);
	};
	const double e0 = error_of(pose_cur);
	if (tid == 0) {
		s_lambda = -1.0; s_nu = 2.0; s_err = e0; s_iters = 0; s_accepted = 0; s_stop = 0; s_fresh = 0;
		ssk_lm_result& R = D.res[wi];
		R.err_init = e0; R.n_obs = W.n_obs; R.stop_reason = 0;
	}
	while (true) {
		__syncthreads();
		if (s_iters >= P.max_iters) break;
		if (W.n_obs > 0 && sqrt(s_err / W.n_obs) < P.stop_rmse) { if (tid == 0) s_stop = 1; break; }
		const double lambda = s_lambda;
		const bool pre = lambda < 0;                   // first pass of a solve: lambda_0 = tau max diag(H)
		const int it_idx = s_iters;
		if (!s_fresh) {
			// ---- linearisation at the current state (kept across rejected attempts) ----
			double acc[NA];
#pragma unroll
			for (int k = 0; k < NA; ++k) acc[k] = 0.0;
			for (int o = tid; o < W.n_obs; o += NT) {
				double r[4], J[E][24];
				bool on[E];
				po_linearise_obs<E>(D, W, P, pose_cur, o, r, J, on);
#pragma unroll
				for (int s = 0; s < E; ++s)
#pragma unroll
					for (int a = 0; a < 6; ++a) {
						const int i = 6 * s + a;
#pragma unroll
						for (int s2 = 0; s2 <= s; ++s2)
#pragma unroll
							for (int b = 0; b < 6; ++b) {
								const int j = 6 * s2 + b;
								if (j <= i) acc[i * (i + 1) / 2 + j] += J[s][a] * J[s2][b] + J[s][6 + a] * J[s2][6 + b] + J[s][12 + a] * J[s2][12 + b] + J[s][18 + a] * J[s2][18 + b];
							}
						acc[NH + i] += J[s][a] * r[0] + J[s][6 + a] * r[1] + J[s][12 + a] * r[2] + J[s][18 + a] * r[3];
					}
			}
			// fixed shuffle tree inside a wave, then the waves' partials in wave order
#pragma unroll
			for (int k = 0; k < NA; ++k) {
				double v = acc[k];
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
				if (lane == 0) s_red[wave][k] = v;
			}
			__syncthreads();
			for (int k = tid; k < NA; k += NT) {
				double v = 0.0;
#pragma unroll
				for (int w = 0; w < NW; ++w) v += s_red[w][k];
				s_H[k] = v;
			}
			__syncthreads();
			if (tid == 0) s_fresh = 1;
		}
		if (pre) {
			if (tid == 0) {
				double mx = 0.0;
#pragma unroll
				for (int i = 0; i < N6; ++i) mx = fmax(mx, s_H[i * (i + 1) / 2 + i]);
				s_lambda = P.tau * mx;
			}
			continue;
		}
		// ---- (H + lambda I) d = g by one thread: L D L^T in registers, positive pivots ----
		if (tid == 0) {
			double a[NH], x[N6];
#pragma unroll
			for (int k = 0; k < NH; ++k) a[k] = s_H[k];
#pragma unroll
			for (int i = 0; i < N6; ++i) { a[i * (i + 1) / 2 + i] += lambda; x[i] = s_H[NH + i]; }
			bool ok = true;
			double iv[N6];
#pragma unroll
			for (int j = 0; j < N6; ++j) {
				// U(i, j) = a(i, j) - sum_k U(i, k) U(j, k) / d_k is in place already (right-looking); column j: pivot, scale, update the rest
				const double d = a[j * (j + 1) / 2 + j];
				ok = ok && d > 0.0;
				iv[j] = 1.0 / d;
#pragma unroll
				for (int i = j + 1; i < N6; ++i) {
					const double f = a[i * (i + 1) / 2 + j] * iv[j];
#pragma unroll
					for (int k = j + 1; k <= i; ++k) a[i * (i + 1) / 2 + k] -= f * a[k * (k + 1) / 2 + j];
				}
				const double fx = x[j] * iv[j];
#pragma unroll
				for (int k = j + 1; k < N6; ++k) x[k] -= fx * a[k * (k + 1) / 2 + j];
			}
			// x = D^-1 z, then L'^T x
#pragma unroll
			for (int j = N6 - 1; j >= 0; --j) {
				double v = x[j];
#pragma unroll
				for (int i = j + 1; i < N6; ++i) v -= a[i * (i + 1) / 2 + j] * x[i];
				x[j] = v * iv[j];
			}
#pragma unroll
			for (int i = 0; i < N6; ++i) s_x[i] = x[i];
			s_ok = ok ? 1 : 0;
		}
		__syncthreads();
		const bool ok = s_ok != 0;
		double dpn = 0.0, err_new = s_err;
		if (ok) {
			if (tid < N6) {
				dpn = s_x[tid] * s_x[tid];
				if (D.trace_on) D.dp_trace[W.dp_trace0 + (size_t)it_idx * N6 + tid] = s_x[tid];
			}
			// edges: T_new = exp(eps) (+) T  (contract L3)
			if (tid < E) {
				const int ei = D.slot_edge[W.us0 + tid];
				const double* T = pose_cur + 12 * ei;
				double* O = pose_new + 12 * ei;
				double eps[6], R[9];
#pragma unroll
				for (int a = 0; a < 6; ++a) eps[a] = s_x[6 * tid + a];
				so3_exp(eps + 3, R);
#pragma unroll
				for (int i = 0; i < 3; ++i) {
#pragma unroll
					for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
					O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + eps[i];
				}
			}
			__syncthreads();
			err_new = error_of(pose_new);
		}
		dpn = block_sum(dpn, s_part);
		__syncthreads();
		if (tid == 0) {
			const int accepted = ok && err_new < s_err;
			if (D.trace_on) {
				ssk_lm_trace_entry& t = D.trace[W.trace0 + it_idx];
				t.lambda = lambda; t.err_before = s_err; t.err_after = err_new; t.delta_p_norm = sqrt(dpn);
				t.delta_f_norm = 0.0; t.accepted = accepted; t.cholesky_ok = ok;
			}
			const int converged = ok && P.conv_rel > 0 && fabs(s_err - err_new) <= P.conv_rel * s_err;
			s_accept = accepted;
			s_iters = it_idx + 1;
			if (accepted) {
				s_err = err_new; s_lambda = lambda * 0.1; s_nu = 2.0; s_accepted += 1; s_fresh = 0;
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
			if (tid < E) {
				const int ei = D.slot_edge[W.us0 + tid];
				for (int c = 0; c < 12; ++c) pose_cur[12 * ei + c] = pose_new[12 * ei + c];
			}
		} else if (ok && tid < E) {                       // (a rejected trial pose must not stay: the next attempt's trial starts from pose_cur anyway)
			const int ei = D.slot_edge[W.us0 + tid];
			for (int c = 0; c < 12; ++c) pose_new[12 * ei + c] = pose_cur[12 * ei + c];
		}
		__syncthreads();
		if (s_stop == 2 || s_stop == 4) break;
	}
	__syncthreads();
	if (tid < E) {
		const int ei = D.slot_edge[W.us0 + tid];
		for (int c = 0; c < 12; ++c) D.pose[12 * (size_t)(W.edge0 + ei) + c] = pose_cur[12 * ei + c];
	}
	if (tid == 0) {
		ssk_lm_result& R = D.res[wi];
		R.iters = s_iters; R.accepted = s_accepted; R.stop_reason = s_stop; R.err_final = s_err; R.lambda_final = s_lambda;
	}
}

// (an instance per E: one edge needs a third of the registers of two; both run over the batch's list, a workgroup of the other size leaves at once)
template <int E>
__global__ __launch_bounds__(NT)
void lm_pose_only_kernel(lm_dev D, const int* __restrict__ order, lm_cam P)
{
	if (D.win[order[blockIdx.x]].E != E) return;
	po_solve<E>(D, order, P);
}

}  // namespace

int ssk_lm_launch_pose_only(ssk_ctx* ctx, const lm_dev& D, const int* d_order, const lm_cam& P, int n_windows, int n_two_edges, size_t dyn_lds_bytes)
{
	if (n_windows <= 0) return SSK_OK;
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	if (n_two_edges < n_windows) hipLaunchKernelGGL(lm_pose_only_kernel<1>, dim3((unsigned)n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, d_order, P);
	if (n_two_edges > 0) hipLaunchKernelGGL(lm_pose_only_kernel<2>, dim3((unsigned)n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, d_order, P);
	SSK_LAUNCH_CHECK(ctx, "lm_pose_only_kernel");
	return SSK_OK;
}
