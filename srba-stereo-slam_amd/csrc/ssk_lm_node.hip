// ssk_lm_node.hip -- relative-BA Levenberg-Marquardt solve in the NODE form (data model and algebra: ssk_lm_node.h; contract L1-L5 of
// oracle/oracle_lm.c; reference: rba.define_new_keyframe EST.cpp:782-784, solver_LM_schur_dense_cholesky SRBA.h:34).
// One persistent 256-thread workgroup per window, several workgroups per CU; wave-private landmark groups, block-major pair phase with
// register accumulators, dense LDL^T in LDS.  No float atomics: every sum has one owner and a fixed order.
#include "ssk_lm.h"
#include "ssk_lm_node.h"

// LM parity is a 1e-6 relative tolerance (contract L5), not bit-exactness: let the compiler fuse multiply-adds here
#pragma clang fp contract(fast)

namespace {

#define SSK_LM_NT 256
#undef SSK_LM_WAVES
#define SSK_LM_WAVES 2
#include "ssk_lm_device.h"

constexpr int NDW = NW;                 // wavefronts per workgroup = groups per tile
constexpr int NQ = NT / 4;              // quads per workgroup
constexpr int REC = SSK_ND_REC, STG = SSK_ND_STAGE, LREC = SSK_ND_LREC, SCR = SSK_ND_SCR;
// node record in LDS (doubles): Zh 0..8, q 9..11, Ph 12..20, vh 21..23
constexpr int R_Q = 9, R_P = 12, R_V = 21;

#include "ssk_lm_node_dev.h"
#include "ssk_lm_ldl_wave.h"

// BIG: the reduced system does not fit LDS: block accumulators (global_atomic_add_f64, still one owner per block and tile, the tiles' barriers
// order the adds: bit-reproducible) and the dense bordered system in HBM / L2, factorised by ldl_solve_hybrid with the node records' LDS
// region (and everything behind it) as its staging
// SMALL: every window of the launch has E <= 10 (the host checks): the only dense solver compiled in is the one-wavefront ldl_solve_mfma_wave (ND_BLOCKS6 = 1: ldl_solve_blocks6_wave) -- the other solvers'
// loop-invariant LDS addresses, hoisted in front of the LM loop, were what the kernel spilled
template <bool BIG, bool SMALL = false>
__device__ __forceinline__ void nd_body(const lm_dev& D0, const nd_dev& N, const lm_cam& P_arg, int dyn_doubles)
{
	lm_dev D = D0;                                     // (lm / lm_new are swapped per accepted step below: ND_LM_FLIP)
	__shared__ lm_cam s_P;
	if (threadIdx.x == 0) s_P = P_arg;
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [pose_cur | pose_new | delta | pivots | block accumulators | gradient | records / reduced system]
	__shared__ double s_part[NW];
	__shared__ double s_lambda, s_nu, s_err;
	__shared__ int s_accept, s_stop, s_iters, s_accepted, s_cur;
	__syncthreads();
	const lm_cam& P = s_P;
	const int wi = N.order[blockIdx.x];
	const nd_win W = N.win[wi];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, quad = tid >> 2, ql = tid & 3;
	const int n6 = 6 * W.E, n6p = (n6 + 1) & ~1, np = n6 | 1, n_pairs = W.E * (W.E + 1) / 2;
	double* pose_cur = s_dyn;
	double* pose_new = s_dyn + 12 * W.n_edges;
	double* s_b = s_dyn + 24 * W.n_edges;
	double* s_inv = s_b + n6p;
	double* Hb = BIG ? N.hb + W.hb0 : s_inv + n6p;     // 36 doubles per block pair (e <= e2): H(6e + a, 6e2 + b) at 36 pid + 6a + b
	double* g_acc = BIG ? s_inv + n6p : Hb + 36 * n_pairs;
	nd_i2* s_suf = reinterpret_cast<nd_i2*>(g_acc + n6p);    // the trie's (first step, rest) records: the table builds walk them
	// the window's tile and group descriptors (a few hundred bytes): every phase starts from them
	nd_tile* s_tile = reinterpret_cast<nd_tile*>(g_acc + n6p + ((W.n_suf + 1) & ~1));
	nd_group* groups = reinterpret_cast<nd_group*>(s_tile + W.n_tiles);
	const int n_groups = W.n_groups;
	double* reg = reinterpret_cast<double*>(groups + n_groups + (W.n_tiles & 1));      // node records of a tile (wave w: reg + 64 REC w), later the dense bordered system
	const int reg_doubles = dyn_doubles - (int)(reg - s_dyn);
	(void)reg_doubles;
	double* stage = reg + 64 * REC * wave;
	double* tabs[2] = {N.tab + W.tab0, N.tab + W.tab0 + 12 * (size_t)W.n_suf};
#ifdef SSK_LM_STAMPS
	// per-phase cycle stamps of window 0 (thread 0, after the phase's barrier): diagnostics build only
	__shared__ unsigned long long s_acc[16];
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
	if (tid < 16) s_acc[tid] = 0;
#define STAMP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_acc[i] += t_ - t_last; t_last = t_; } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

	for (int i = tid; i < 12 * W.n_edges; i += NT) {
		const double v = D.pose[12 * (size_t)W.edge0 + i];
		pose_cur[i] = v; pose_new[i] = v;
	}
	for (int i = tid; i < W.n_suf; i += NT) s_suf[i] = N.suf[W.suf0 + i];
	for (int i = tid; i < W.n_tiles; i += NT) s_tile[i] = N.tile[W.tile0 + i];
	for (int i = tid; i < n_groups; i += NT) groups[i] = N.group[W.grp0 + i];
	// (a landmark nobody observes has no root lane: its trial position is its position)
	for (int i = tid; i < 3 * W.n_lms; i += NT) D.lm_new[3 * (size_t)W.lm0 + i] = D.lm[3 * (size_t)W.lm0 + i];
	__syncthreads();
	nd_build_table(W, s_suf, pose_cur, tabs[0]);
	__syncthreads();
	double e0 = 0.0, dummy = 0.0;
	{
		nd_recs nx = nd_fetch_recs(N, W, groups[min(wave, max(n_groups - 1, 0))], lane);
		for (int g = wave; g < n_groups; g += NDW) {
			const nd_recs rc = nx;
			if (g + NDW < n_groups) nx = nd_fetch_recs(N, W, groups[g + NDW], lane);
			nd_score_group(D, N, W, P, groups[g], rc, tabs[0], stage, s_b, false, 0, lane, e0, dummy);
		}
	}
	e0 = block_sum(e0, s_part);
	if (tid == 0) {
		s_lambda = -1.0; s_nu = 2.0; s_err = e0; s_iters = 0; s_accepted = 0; s_stop = 0; s_cur = 0;
		ssk_lm_result& R = D.res[wi];
		R.err_init = e0; R.n_obs = W.n_obs; R.stop_reason = (W.E == 0 && W.L == 0) ? 3 : 0;
	}
	__syncthreads();
	if (W.E == 0 && W.L == 0) {
		if (tid == 0) { ssk_lm_result& R = D.res[wi]; R.iters = 0; R.accepted = 0; R.err_final = e0; R.lambda_final = -1.0; }
		return;
	}

	while (true) {
		__syncthreads();
		if (s_iters >= P.max_iters) break;
		if (W.n_obs > 0 && sqrt(s_err / W.n_obs) < P.stop_rmse) { if (tid == 0) s_stop = 1; break; }
		const double lambda = s_lambda;
		const bool pre = lambda < 0;                   // first pass of a solve: only diag(H_pp) and diag(V), for lambda_0
		const int it_idx = s_iters, cur = s_cur;
		const double* tab_cur = tabs[cur];
#if ND_LM_FLIP
		// (round 5) the landmarks double-buffered like the suffix tables: the trial positions of an accepted step ARE the next state (no copy pass, no
		// barrier behind it); landmarks the window holds are equal in both buffers from the start
		D.lm = cur ? D0.lm_new : D0.lm; D.lm_new = cur ? D0.lm : D0.lm_new;
#endif
		double vmax = 0.0;
		STAMP(0);
		if (BIG) {
			for (int i = tid; i < 36 * n_pairs; i += NT) __hip_atomic_store(Hb + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (int i = tid; i < n6p; i += NT) g_acc[i] = 0.0;
		} else
			for (int i = tid; i < 36 * n_pairs + n6p; i += NT) Hb[i] = 0.0;      // (block accumulators and gradient are contiguous)
		__syncthreads();
		STAMP(1);
		// ---- linearisation: tile by tile ----
		nd_recs nx = nd_fetch_recs(N, W, groups[min(wave, max(n_groups - 1, 0))], lane);
#if ND_PREFETCH
		nd_pre pf;
		nd_prefetch_group(D, W, groups[min(wave, max(n_groups - 1, 0))], nx, tab_cur, lane, pf);
#endif
		for (int t = 0; t < W.n_tiles; ++t) {
			const nd_tile T = s_tile[t];
			nd_i2 q0 = nd_i2{0xFFFF, 0};               // (round 0's slot in flight behind the group phase)
#if SSK_ND_LANE_RUNS
			if (T.n_rounds > 0) q0 = N.qs[W.qs0 + T.qs0 + tid];
#else
			if (T.n_rounds > 0) q0 = N.qs[W.qs0 + T.qs0 + quad];
#endif
			const nd_recs rc = nx;
			if (t + 1 < W.n_tiles) nx = nd_fetch_recs(N, W, groups[min(s_tile[t + 1].grp0 + wave, n_groups - 1)], lane);      // the next tile's records: in flight during this one
#if ND_PREFETCH
			if (wave < T.n_grp) nd_linearise_group<false, true>(D, N, W, P, groups[T.grp0 + wave], rc, tab_cur, stage, lambda, pre, lane, vmax, &pf);
#else
			if (wave < T.n_grp) nd_linearise_group(D, N, W, P, groups[T.grp0 + wave], rc, tab_cur, stage, lambda, pre, lane, vmax);
#endif
#if SSK_ND_LANE_RUNS
			nd_items its = nd_items_fetch(N.item + W.item0 + q0.y, (q0.x >> 16) & 0x7FFF, 0, 1);      // (in flight across the barrier)
			STAMP(2);
			__syncthreads();
			STAMP(3);
			for (int r = 0; r < T.n_rounds; ++r) {
				// a lane = a run of consecutive items of one block; its 36 sums go straight to the block's accumulator (ds_add_f64, no return)
				const nd_i2 q = r == 0 ? q0 : N.qs[W.qs0 + T.qs0 + r * NT + tid];
				const int pid = q.x & 0xFFFF, n = (q.x >> 16) & 0x7FFF;
				if (r > 0) its = nd_items_fetch(N.item + W.item0 + q.y, n, 0, 1);
				double acc[36];
#pragma unroll
				for (int c = 0; c < 36; ++c) acc[c] = 0.0;
				FINE0();
				nd_pair_items(reg, N.item + W.item0 + q.y, its, n, 0, 1, pre, acc);
				FINE(5);
				if (pid != 0xFFFF) {
					double* hb = Hb + 36 * pid;
#pragma unroll
					for (int c = 0; c < 36; ++c) (void)__hip_atomic_fetch_add(hb + c, acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
				FINE(6);
			}
#else
			nd_items its = nd_items_fetch(N.item + W.item0 + (q0.y & 0x0FFFFFFF), (q0.x >> 16) & 0x7FFF, ql, 4);      // (in flight across the barrier)
			STAMP(2);
			__syncthreads();
			STAMP(3);
			for (int r = 0; r < T.n_rounds; ++r) {
				const nd_i2 q = r == 0 ? q0 : N.qs[W.qs0 + T.qs0 + r * NQ + quad];
				const int pid = q.x & 0xFFFF, n = (q.x >> 16) & 0x7FFF;
				if (r > 0) its = nd_items_fetch(N.item + W.item0 + (q.y & 0x0FFFFFFF), n, ql, 4);
				double acc[36];
#pragma unroll
				for (int c = 0; c < 36; ++c) acc[c] = 0.0;
				FINE0();
				nd_pair_items(reg, N.item + W.item0 + (q.y & 0x0FFFFFFF), its, n, ql, 4, pre, acc);
				FINE(5);
				nd_quad_sum36(acc);
				// a block cut in two parts sits in the two quads of one aligned lane octet: the second quad's sums come over by a mirror step
				// (only in the wavefronts that hold such blocks: bit 31 of y is set in all their slots)
				if (__builtin_amdgcn_readfirstlane(q.y) < 0) {
					const double on = ((q.y >> 30) & 1) ? 1.0 : 0.0;
#pragma unroll
					for (int c = 0; c < 36; ++c) acc[c] += on * dpp_f64<0x141>(acc[c]);      // row_half_mirror: lane i <-> 7 - i of its octet
				}
				if (ql == 0 && pid != 0xFFFF && ((q.y >> 28) & 3) == 0) {
#if ND_HB_ATOMIC
					// one owner per block and tile: ds_add_f64 without a return is a plain read-modify-write in the LDS unit (no wait for a read)
					double* hb = Hb + 36 * pid;
					if (BIG) {
#pragma unroll
						for (int c = 0; c < 36; ++c) (void)unsafeAtomicAdd(hb + c, acc[c]);
					} else {
#pragma unroll
						for (int c = 0; c < 36; ++c) (void)__hip_atomic_fetch_add(hb + c, acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
#else
					v2d_t* hb = reinterpret_cast<v2d_t*>(Hb + 36 * pid);
					v2d_t old[18];
#pragma unroll
					for (int c = 0; c < 18; ++c) old[c] = hb[c];
#pragma unroll
					for (int c = 0; c < 18; ++c) { v2d_t x = old[c]; x.x += acc[2 * c]; x.y += acc[2 * c + 1]; hb[c] = x; }
#endif
				}
				FINE(6);
			}
#endif
			FINE0();
			if (!pre)
				for (int k = quad; k < T.n_gqs; k += NQ) {
					// reduced gradient of one edge: sum over the tile's nodes on it of [vh; q x vh]
					const nd_i2 q = N.gqs[W.gqs0 + T.gqs0 + k];
					const int e = q.x & 0xFFFF, n = (q.x >> 16) & 0xFFFF;
					const int* items = N.gitem + W.gitem0 + q.y;
					double ga[6] = {0, 0, 0, 0, 0, 0};
					int gi[ND_ITEMS_REG];
#pragma unroll
					for (int k2 = 0; k2 < ND_ITEMS_REG; ++k2) gi[k2] = items[min(ql + 4 * k2, n - 1)];
#pragma unroll
					for (int k2 = 0; k2 < ND_ITEMS_REG; ++k2)
						if (ql + 4 * k2 < n) {
							const double* rcd = reg + REC * gi[k2];
							const double vh[3] = {rcd[R_V], rcd[R_V + 1], rcd[R_V + 2]}, qq[3] = {rcd[R_Q], rcd[R_Q + 1], rcd[R_Q + 2]};
							double x[3];
							nd_cross(qq, vh, x);
#pragma unroll
							for (int c = 0; c < 3; ++c) { ga[c] += vh[c]; ga[3 + c] += x[c]; }
						}
					for (int i = ql + 4 * ND_ITEMS_REG; i < n; i += 4) {
						const double* rcd = reg + REC * items[i];
						const double vh[3] = {rcd[R_V], rcd[R_V + 1], rcd[R_V + 2]}, qq[3] = {rcd[R_Q], rcd[R_Q + 1], rcd[R_Q + 2]};
						double x[3];
						nd_cross(qq, vh, x);
#pragma unroll
						for (int c = 0; c < 3; ++c) { ga[c] += vh[c]; ga[3 + c] += x[c]; }
					}
#pragma unroll
					for (int c = 0; c < 6; ++c) ga[c] += dpp_f64<0xB1>(ga[c]);
#pragma unroll
					for (int c = 0; c < 6; ++c) ga[c] += dpp_f64<0x4E>(ga[c]);
					if (ql == 0) {
#pragma unroll
						for (int c = 0; c < 6; ++c) (void)__hip_atomic_fetch_add(g_acc + 6 * e + c, ga[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
			FINE(7);
			STAMP(4);
#if ND_PREFETCH
			// the next tile's group-phase operands (its records arrived during this tile): in flight across the barrier
			// (not a cycle earlier: hoisted into the pair phase their 60 registers are spilled)
			__builtin_amdgcn_sched_barrier(0);
			// (unconditional -- the last tile fetches its own group's again --: a guarded assignment keeps the OLD operands alive through the whole tile)
			nd_prefetch_group(D, W, groups[min(s_tile[min(t + 1, W.n_tiles - 1)].grp0 + wave, n_groups - 1)], nx, tab_cur, lane, pf);
#endif
			__syncthreads();
			STAMP(5);
		}
		if (pre) {
			// lambda0 = tau * max diag(H): the diagonal blocks of H_pp and the landmark blocks' diagonals
			double mx = vmax;
			for (int it = tid; it < n6; it += NT) {
				const int e = it / 6, a = it - 6 * e;
				const double* hd = Hb + 36 * (e * (e + 1) / 2 + e) + 7 * a;
				mx = fmax(mx, BIG ? __hip_atomic_load(hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *hd);
			}
			mx = block_max(mx, s_part);
			if (tid == 0) s_lambda = P.tau * mx;
			continue;
		}
		// ---- the reduced system: A = H_pp - S + lambda I, border row = reduced gradient; LDL^T ----
		bool ok = true;
		if (!BIG && ND_BLOCKS6 && SMALL) {
			// E <= 10 (the bench windows): one wavefront, no workgroup barrier inside the solve
			STAMP(6);
#if ND_BLOCKS6 == 2
			if (n6 > 0) ok = ldl_solve_mfma_wave(Hb, g_acc, lambda, n6, s_b, reg);          // one wavefront, the system in matrix-core accumulator tiles
#else
			if (n6 > 0) ok = ldl_solve_blocks6_wave(Hb, g_acc, lambda, W.E, s_b, reg);
#endif
		} else if (SMALL) {
			ok = false;                                    // (not reached: the host launches this variant for batches of E <= 10 only)
		} else if (!BIG && ND_TILES4 && 2 * ((n6 + 3) >> 2) * (((n6 + 3) >> 2) + 1) <= 4 * NT) {
			// register tiles straight from the block accumulators (no dense copy)
			STAMP(6);
			if (n6 > 0) ok = ldl_solve_tiles4_lds(Hb, g_acc, lambda, n6, s_b, reg);
		} else {
			double* Asys = BIG ? D.Hred + W.h0 : reg;
			for (int it = tid; it < 36 * n_pairs; it += NT) {
				const int pid = it / 36, ab = it - 36 * pid, a = ab / 6, b = ab - 6 * a;
				int e2 = (int)((sqrtf(8.0f * (float)pid + 1.0f) - 1.0f) * 0.5f);
				while ((e2 + 1) * (e2 + 2) / 2 <= pid) ++e2;
				while (e2 * (e2 + 1) / 2 > pid) --e2;
				const int e = pid - e2 * (e2 + 1) / 2;
				if (e == e2 && a > b) continue;
				const int row = 6 * e + a, col = 6 * e2 + b;
				double v = BIG ? __hip_atomic_load(Hb + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : Hb[it];
				if (row == col) v += lambda;
				Asys[(size_t)row * np + col] = v;
				Asys[(size_t)col * np + row] = v;
			}
			for (int i = tid; i < n6; i += NT) Asys[(size_t)n6 * np + i] = g_acc[i];
			__syncthreads();
			STAMP(6);
			if (n6 > 0) {
				if constexpr (BIG) ok = SSK_LDL_BIG(Asys, np, n6, s_b, s_inv, reg, reg_doubles);
				else ok = ldl_solve_lds(reg, np, n6, s_b, s_inv);
			}
		}
		STAMP(7);
		double dpn = 0, dfn = 0, err_new = s_err;
		if (ok) {
			for (int i = tid; i < n6; i += NT) {
				dpn += s_b[i] * s_b[i];
				if (D.trace_on) D.dp_trace[W.dp_trace0 + (size_t)it_idx * n6 + i] = s_b[i];
			}
			// edges: T_new = exp(eps) (+) T
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
			__syncthreads();
			nd_build_table(W, s_suf, pose_new, tabs[cur ^ 1]);
			__syncthreads();
			STAMP(8);
			double e = 0.0;
#if ND_PREFETCH
			{
				// records two groups ahead, operands one group ahead (indices clamped: the fetches past the last group repeat it and are not used)
				const int gl = max(n_groups - 1, 0);
				nd_recs r0 = nd_fetch_recs(N, W, groups[min(wave, gl)], lane);
				nd_recs r1 = nd_fetch_recs(N, W, groups[min(wave + NDW, gl)], lane);
				nd_spre d0;
				nd_prefetch_score(D, N, W, groups[min(wave, gl)], r0, tabs[cur ^ 1], lane, d0);
				for (int g = wave; g < n_groups; g += NDW) {
					const nd_recs rc = r0;
					const nd_spre dc = d0;
					r0 = r1;
					r1 = nd_fetch_recs(N, W, groups[min(g + 2 * NDW, gl)], lane);
					nd_prefetch_score(D, N, W, groups[min(g + NDW, gl)], r0, tabs[cur ^ 1], lane, d0);
					nd_score_group<true>(D, N, W, P, groups[g], rc, tabs[cur ^ 1], stage, s_b, true, it_idx, lane, e, dfn, &dc);
				}
			}
#else
			{
				nd_recs nx = nd_fetch_recs(N, W, groups[min(wave, max(n_groups - 1, 0))], lane);
				for (int g = wave; g < n_groups; g += NDW) {
					const nd_recs rc = nx;
					if (g + NDW < n_groups) nx = nd_fetch_recs(N, W, groups[g + NDW], lane);
					nd_score_group(D, N, W, P, groups[g], rc, tabs[cur ^ 1], stage, s_b, true, it_idx, lane, e, dfn);
				}
			}
#endif
			err_new = block_sum(e, s_part);
			STAMP(9);
		}
		dpn = block_sum(dpn, s_part);
		dfn = block_sum(dfn, s_part);
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
				s_err = err_new; s_lambda = lambda * 0.1; s_nu = 2.0; s_accepted += 1; s_cur = cur ^ 1;
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
#if !ND_LM_FLIP
			for (int l = tid; l < W.L; l += NT) {
				const int li = D.slot_lm[W.ls0 + l];
				for (int c = 0; c < 3; ++c) D.lm[3 * (size_t)(W.lm0 + li) + c] = D.lm_new[3 * (size_t)(W.lm0 + li) + c];
			}
#endif
		}
#if !ND_LM_FLIP
		__syncthreads();
#endif
		// (with the landmarks double-buffered the accepted state's copy is the edge poses in LDS alone: the barrier at the loop's head -- or the one
		// behind the loop -- orders it before its readers; s_stop was written in front of the barrier above)
		STAMP(10);
		if (s_stop == 2 || s_stop == 4) break;
	}
	__syncthreads();
#if ND_LM_FLIP
	if (s_cur) {                                       // the final landmarks sit in the second buffer: to the batch's state array
		for (int l = tid; l < W.L; l += NT) {
			const int li = D0.slot_lm[W.ls0 + l];
			for (int c = 0; c < 3; ++c) D0.lm[3 * (size_t)(W.lm0 + li) + c] = D0.lm_new[3 * (size_t)(W.lm0 + li) + c];
		}
	}
#endif
	// publish the final edge poses (they lived in LDS during the solve)
	for (int e = tid; e < W.E; e += NT) {
		const int ei = D.slot_edge[W.us0 + e];
		for (int c = 0; c < 12; ++c) D.pose[12 * (size_t)(W.edge0 + ei) + c] = pose_cur[12 * ei + c];
	}
	if (tid == 0) {
		ssk_lm_result& R = D.res[wi];
		R.iters = s_iters; R.accepted = s_accepted; R.stop_reason = s_stop; R.err_final = s_err; R.lambda_final = s_lambda;
#ifdef SSK_LM_STAMPS
		if (wi == 0)
			printf("node stamps (cycles, window 0, %d iters, %d tiles): loop top %llu zero %llu | groups: own %llu + barrier %llu, pairs: own %llu + barrier %llu | assemble %llu solve %llu "
			       "edges + table %llu score %llu decision + copy %llu\n", s_iters, W.n_tiles, s_acc[0], s_acc[1], s_acc[2], s_acc[3], s_acc[4], s_acc[5], s_acc[6], s_acc[7], s_acc[8], s_acc[9], s_acc[10]);
		if (wi == 0)
			printf("node fine (thread 0, cumulative over launches): group: loads + observation %llu, run sums %llu, roots %llu, nodes %llu, record stores %llu | pairs: items %llu, quad sum + adds %llu, gradient %llu\n",
			       g_nd_fine[0], g_nd_fine[1], g_nd_fine[2], g_nd_fine[3], g_nd_fine[4], g_nd_fine[5], g_nd_fine[6], g_nd_fine[7]);
#endif
	}
}

__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_node_kernel(lm_dev D, nd_dev N, lm_cam P, int dyn_doubles)
{
	nd_body<false>(D, N, P, dyn_doubles);
}

__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_solve_node_e10_kernel(lm_dev D, nd_dev N, lm_cam P, int dyn_doubles)
{
	nd_body<false, true>(D, N, P, dyn_doubles);
}

__global__ __launch_bounds__(NT, 1)
void lm_solve_node_big_kernel(lm_dev D, nd_dev N, lm_cam P, int dyn_doubles)
{
	nd_body<true>(D, N, P, dyn_doubles);
}

}  // namespace

int ssk_lm_launch_node(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const lm_cam& P, int n_windows, size_t dyn_lds_bytes, bool big, bool small_e)
{
	dyn_lds_bytes = (dyn_lds_bytes + 15) & ~(size_t)15;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_node_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_node_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_solve_node_e10_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	if (big) hipLaunchKernelGGL(lm_solve_node_big_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, N, P, (int)(dyn_lds_bytes / 8));
	else if (small_e && ND_BLOCKS6) hipLaunchKernelGGL(lm_solve_node_e10_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, N, P, (int)(dyn_lds_bytes / 8));
	else hipLaunchKernelGGL(lm_solve_node_kernel, dim3(n_windows), dim3(NT), dyn_lds_bytes, ctx->stream, D, N, P, (int)(dyn_lds_bytes / 8));
	SSK_LAUNCH_CHECK(ctx, big ? "lm_solve_node_big_kernel" : (small_e && ND_BLOCKS6) ? "lm_solve_node_e10_kernel" : "lm_solve_node_kernel");
	return SSK_OK;
}
