// ssk_lm_split.hip -- phase-split node form of the relative-BA LM solve: the per-tile kernels (linearise, score), the per-window decision
// kernel and the launch sequence.  Overview, data model: ssk_lm_split.h; algebra: ssk_lm_node.h; contract L1-L5: oracle/oracle_lm.c;
// reference: rba.define_new_keyframe EST.cpp:782-784 (optimise depth 5, KITTI.ini:104-106), solver_LM_schur_dense_cholesky SRBA.h:34.
#include "ssk_lm_split.h"

// LM parity is a 1e-6 relative tolerance (contract L5), not bit-exactness: let the compiler fuse multiply-adds here
#pragma clang fp contract(fast)

namespace {

#define SSK_LM_NT 256
#undef SSK_LM_WAVES
#define SSK_LM_WAVES 3
#include "ssk_lm_device.h"

constexpr int NDW = NW;                 // wavefronts per workgroup = groups per tile
constexpr int NQ = NT / 4;              // quads per workgroup
constexpr int REC = SSK_ND_REC, STG = SSK_ND_STAGE, LREC = SSK_ND_LREC, SCR = SSK_ND_SCR;
constexpr int R_Q = 9, R_P = 12, R_V = 21;
static_assert(NDW == SSK_SP_WAVES, "tile = SSK_SP_WAVES groups");

#include "ssk_lm_node_dev.h"

// ---- K0: per window: loop state, trial buffers = state, suffix table of the initial state ----
__global__ __launch_bounds__(NT)
void lm_split_init_kernel(lm_dev D, nd_dev N, sp_dev S)
{
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // [poses | trie records]
	const int wi = S.win_map[blockIdx.x];
	if (wi < 0) return;
	const nd_win W = N.win[wi];
	const int tid = threadIdx.x;
	double* pose = s_dyn;
	nd_i2* s_suf = reinterpret_cast<nd_i2*>(s_dyn + 12 * W.n_edges);
	for (int i = tid; i < 12 * W.n_edges; i += NT) {
		const double v = D.pose[12 * (size_t)W.edge0 + i];
		pose[i] = v; D.pose_new[12 * (size_t)W.edge0 + i] = v;
	}
	for (int i = tid; i < 3 * W.n_lms; i += NT) D.lm_new[3 * (size_t)W.lm0 + i] = D.lm[3 * (size_t)W.lm0 + i];
	for (int i = tid; i < W.n_suf; i += NT) s_suf[i] = N.suf[W.suf0 + i];
	if (tid == 0) {
		sp_ctl c{};
		c.lambda = -1.0; c.nu = 2.0; c.err = 0.0; c.dpn = 0.0;
		c.iters = 0; c.accepted = 0; c.stop = 0; c.cur = 0; c.active = 1; c.ok = 0; c.pre = 1;
		S.ctl[wi] = c;
	}
	__syncthreads();
	nd_build_table(W, s_suf, pose, N.tab + W.tab0);
}

// ---- K1: one tile (NDW landmark groups) of one window: group phase, pair phase -> partial slots ----
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_split_linearise_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P_arg)
{
	__shared__ lm_cam s_P;
	__shared__ double s_part[NW];
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // node records of the tile: wave w at s_dyn + 64 REC w
	const nd_i2 tm = S.tile_map[blockIdx.x];
	if (tm.x < 0) return;
	const int wi = tm.x;
	const sp_ctl c = S.ctl[wi];
	if (!c.active) return;
	if (threadIdx.x == 0) s_P = P_arg;
	const nd_win W = N.win[wi];
	const nd_split SW = S.sw[wi];
	const nd_tile T = N.tile[W.tile0 + tm.y];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, quad = tid >> 2, ql = tid & 3;
	const bool pre = c.pre != 0;
	const double lambda = c.lambda;
	const double* tab_cur = N.tab + W.tab0 + (c.cur ? 12 * (size_t)W.n_suf : 0);
	double* reg = s_dyn;
	double* stage = reg + 64 * REC * wave;
	// the first round's slot and item codes: in flight behind the group phase
	nd_i4 q0 = nd_i4{0, 0, 0, 0};
	if (quad < T.n_rounds) q0 = S.sqs[SW.sqs0 + T.qs0 + quad];
	__syncthreads();
	const lm_cam& P = s_P;
	double vmax = 0.0;
	if (wave < T.n_grp) {
		const nd_group G = N.group[W.grp0 + T.grp0 + wave];
		const nd_recs rc = nd_fetch_recs(N, W, G, lane);
		nd_linearise_group(D, N, W, P, G, rc, tab_cur, stage, lambda, pre, lane, vmax);
	}
	nd_items its = nd_items_fetch(N.item + W.item0 + q0.z, q0.y, ql, 4);
	__syncthreads();
	for (int k = quad; k < T.n_rounds; k += NQ) {
		const nd_i4 q = k == quad ? q0 : S.sqs[SW.sqs0 + T.qs0 + k];
		const int n = q.y;
		if (k != quad) its = nd_items_fetch(N.item + W.item0 + q.z, n, ql, 4);
		// the lambda_0 pass only needs the diagonal blocks (bit 30 of x)
		if (pre && !((q.x >> 30) & 1)) continue;
		double acc[36];
#pragma unroll
		for (int i = 0; i < 36; ++i) acc[i] = 0.0;
		nd_pair_items(reg, N.item + W.item0 + q.z, its, n, ql, 4, pre, acc);
		nd_quad_sum36(acc);
		if (ql == 0) {
			v2d_t* o = reinterpret_cast<v2d_t*>(S.Hpart + 36 * (size_t)(SW.part0 + q.w));
#pragma unroll
			for (int i = 0; i < 18; ++i) { v2d_t x; x.x = acc[2 * i]; x.y = acc[2 * i + 1]; o[i] = x; }
		}
	}
	if (!pre)
		for (int k = quad; k < T.n_gqs; k += NQ) {
			// reduced gradient of one edge: sum over the tile's nodes on it of [vh; q x vh]
			const nd_i2 q = N.gqs[W.gqs0 + T.gqs0 + k];
			const int n = (q.x >> 16) & 0xFFFF;
			const int* items = N.gitem + W.gitem0 + q.y;
			double ga[6] = {0, 0, 0, 0, 0, 0};
			for (int i = ql; i < n; i += 4) {
				const double* rcd = reg + REC * items[i];
				const double vh[3] = {rcd[R_V], rcd[R_V + 1], rcd[R_V + 2]}, qq[3] = {rcd[R_Q], rcd[R_Q + 1], rcd[R_Q + 2]};
				double x[3];
				nd_cross(qq, vh, x);
#pragma unroll
				for (int i2 = 0; i2 < 3; ++i2) { ga[i2] += vh[i2]; ga[3 + i2] += x[i2]; }
			}
#pragma unroll
			for (int i = 0; i < 6; ++i) ga[i] += dpp_f64<0xB1>(ga[i]);
#pragma unroll
			for (int i = 0; i < 6; ++i) ga[i] += dpp_f64<0x4E>(ga[i]);
			if (ql == 0) {
				double* o = S.gpart + 6 * (size_t)(SW.gpart0 + S.gslot[SW.gslot0 + T.gqs0 + k]);
#pragma unroll
				for (int i = 0; i < 6; ++i) o[i] = ga[i];
			}
		}
	if (pre) {
		const double mx = block_max(vmax, s_part);
		if (tid == 0) S.vpart[SW.gtile0 + tm.y] = mx;
	}
}

// ---- K3: trial state of one tile: landmark steps + robust error (init != 0: the error of the initial state, no step) ----
__global__ __launch_bounds__(NT, 4)
void lm_split_score_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P_arg, int init)
{
	__shared__ lm_cam s_P;
	__shared__ double s_part[NW];
	__shared__ double s_stage[NW * 64 * 3];
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // the edge step (6E)
	const nd_i2 tm = S.tile_map[blockIdx.x];
	if (tm.x < 0) return;
	const int wi = tm.x;
	const sp_ctl c = S.ctl[wi];
	if (!c.active) return;
	if (!init && !c.ok) return;
	if (threadIdx.x == 0) s_P = P_arg;
	const nd_win W = N.win[wi];
	const nd_split SW = S.sw[wi];
	const nd_tile T = N.tile[W.tile0 + tm.y];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double* s_b = s_dyn;
	if (!init)
		for (int i = tid; i < 6 * W.E; i += NT) s_b[i] = D.dp[6 * (size_t)W.us0 + i];
	__syncthreads();
	const lm_cam& P = s_P;
	const double* tab = N.tab + W.tab0 + ((init ? c.cur : (c.cur ^ 1)) ? 12 * (size_t)W.n_suf : 0);
	double e = 0.0, dfn = 0.0;
	if (wave < T.n_grp) {
		const nd_group G = N.group[W.grp0 + T.grp0 + wave];
		const nd_recs rc = nd_fetch_recs(N, W, G, lane);
		nd_score_group(D, N, W, P, G, rc, tab, s_stage + 64 * 3 * wave, s_b, !init, c.iters, lane, e, dfn);
	}
	e = block_sum(e, s_part);
	dfn = block_sum(dfn, s_part);
	if (tid == 0) { S.epart[SW.gtile0 + tm.y] = e; S.fpart[SW.gtile0 + tm.y] = dfn; }
}

// ---- K4: per window: the attempt's verdict (contract L5), trace, state copy, stop flags ----
__global__ __launch_bounds__(NT)
void lm_split_decide_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P, int init)
{
	__shared__ int s_copy;
	const int wi = S.win_map[blockIdx.x];
	if (wi < 0) return;
	const int tid = threadIdx.x;
	const sp_ctl c0 = S.ctl[wi];
	if (!c0.active) return;
	const nd_win W = N.win[wi];
	const nd_split SW = S.sw[wi];
	if (tid == 0) {
		sp_ctl c = c0;
		double e = 0.0, dfn = 0.0;
		// tile order: the same sum in every launch
		if (init || c.ok)
			for (int t = 0; t < W.n_tiles; ++t) { e += S.epart[SW.gtile0 + t]; dfn += S.fpart[SW.gtile0 + t]; }
		int copy = 0;
		ssk_lm_result& R = D.res[wi];
		if (init) {
			c.err = e;
			R.err_init = e; R.n_obs = W.n_obs; R.stop_reason = 0;
			if (W.E == 0 && W.L == 0) { c.stop = 3; c.active = 0; c.lambda = -1.0; }
		} else {
			const double err_new = c.ok ? e : c.err;
			const int accepted = c.ok && err_new < c.err;
			if (D.trace_on) {
				ssk_lm_trace_entry& t = D.trace[W.trace0 + c.iters];
				t.lambda = c.lambda; t.err_before = c.err; t.err_after = err_new; t.delta_p_norm = sqrt(c.dpn);
				t.delta_f_norm = sqrt(dfn); t.accepted = accepted; t.cholesky_ok = c.ok;
			}
			const int converged = c.ok && P.conv_rel > 0 && fabs(c.err - err_new) <= P.conv_rel * c.err;
			c.iters += 1;
			if (accepted) {
				c.err = err_new; c.lambda *= 0.1; c.nu = 2.0; c.accepted += 1; c.cur ^= 1; copy = 1;
				if (converged) c.stop = 4;
			} else if (converged) {
				c.stop = 4;
			} else {
				c.lambda *= c.nu; c.nu *= 2.0;
				if (c.lambda > P.max_lambda) c.stop = 2;
			}
			if (c.stop == 2 || c.stop == 4) c.active = 0;
		}
		// the loop's head for the next attempt
		if (c.active) {
			if (c.iters >= P.max_iters) c.active = 0;
			else if (W.n_obs > 0 && sqrt(c.err / W.n_obs) < P.stop_rmse) { c.stop = 1; c.active = 0; }
		}
		if (!c.active) { R.iters = c.iters; R.accepted = c.accepted; R.stop_reason = c.stop; R.err_final = c.err; R.lambda_final = c.lambda; }
		S.ctl[wi] = c;
		s_copy = copy;
	}
	__syncthreads();
	if (s_copy) {
		// accepted: the trial state becomes the state (the tables swapped by `cur`)
		for (int e = tid; e < W.E; e += NT) {
			const int ei = D.slot_edge[W.us0 + e];
#pragma unroll
			for (int k = 0; k < 12; ++k) D.pose[12 * (size_t)(W.edge0 + ei) + k] = D.pose_new[12 * (size_t)(W.edge0 + ei) + k];
		}
		for (int l = tid; l < W.L; l += NT) {
			const int li = D.slot_lm[W.ls0 + l];
#pragma unroll
			for (int k = 0; k < 3; ++k) D.lm[3 * (size_t)(W.lm0 + li) + k] = D.lm_new[3 * (size_t)(W.lm0 + li) + k];
		}
	}
}

}  // namespace

int ssk_lm_launch_split(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const sp_dev& S, const lm_cam& P, int n_windows, size_t solve_dyn_bytes, size_t init_dyn_bytes)
{
	(void)n_windows;
	static bool attr_set = false;
	if (!attr_set) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_linearise_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		attr_set = true;
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	const size_t lin_lds = sizeof(double) * (size_t)NDW * 64 * REC;
	const size_t small_lds = (init_dyn_bytes + 15) & ~(size_t)15;      // K0: the window's poses and trie records
	const dim3 gt((unsigned)S.n_tile_wgs), gw((unsigned)S.n_win_wgs), bt(NT);
	hipLaunchKernelGGL(lm_split_init_kernel, gw, bt, small_lds, ctx->stream, D, N, S);
	hipLaunchKernelGGL(lm_split_score_kernel, gt, bt, 0, ctx->stream, D, N, S, P, 1);
	hipLaunchKernelGGL(lm_split_decide_kernel, gw, bt, 0, ctx->stream, D, N, S, P, 1);
	SSK_LAUNCH_CHECK(ctx, "lm_split_init/score/decide");
	for (int pass = 0; pass <= P.max_iters; ++pass) {
		hipLaunchKernelGGL(lm_split_linearise_kernel, gt, bt, lin_lds, ctx->stream, D, N, S, P);
		const int rc = ssk_lm_launch_split_solve(ctx, D, N, S, P, solve_dyn_bytes);
		if (rc) return rc;
		if (pass == 0) continue;                           // every window's first pass is its lambda_0 pass: no attempt to score
		hipLaunchKernelGGL(lm_split_score_kernel, gt, bt, sizeof(double) * 6 * 128, ctx->stream, D, N, S, P, 0);
		hipLaunchKernelGGL(lm_split_decide_kernel, gw, bt, 0, ctx->stream, D, N, S, P, 0);
	}
	SSK_LAUNCH_CHECK(ctx, "lm_split kernels");
	return SSK_OK;
}
