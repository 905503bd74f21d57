// ssk_lm_split.hip -- phase-split node form of the relative-BA LM solve: the per-tile kernels (linearise, score), the per-window decision
// kernel and the launch sequence.  Overview, data model: ssk_lm_split.h; algebra: ssk_lm_node.h; contract L1-L5: oracle/oracle_lm.c;
// reference: rba.define_new_keyframe EST.cpp:782-784 (optimise depth 5, KITTI.ini:104-106), solver_LM_schur_dense_cholesky SRBA.h:34.
#include "ssk_lm_split.h"
#include <algorithm>
#include <cstdlib>

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

#ifdef SSK_LM_STAMPS
// per-phase clock sums over all K1 workgroups of the non-lambda_0 passes (thread 0 of each): diagnostics build only (tools/split_stamps.py)
__device__ unsigned long long g_sp_t[16];
#define SP_T0() unsigned long long sp_t_ = __builtin_amdgcn_s_memtime()
#define SP_T(i) do { if (threadIdx.x == 0 && !pre) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_sp_t[i], t_ - sp_t_); sp_t_ = t_; } } while (0)
#else
#define SP_T0() do {} while (0)
#define SP_T(i) do {} while (0)
#endif

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
		atomicAdd(S.n_active, 1);
	}
	__syncthreads();
	nd_build_table(W, s_suf, pose, N.tab + W.tab0);
}

// ---- K1: one tile (NDW landmark groups) of one window: group phase, pair phase -> partial slots ----
// Pair phase: LANE slots (host: ssk_lm_node_build.h): a block has 1, 2 or 4 lanes, its items dealt out among them, so that every lane of the
// workgroup walks a run of about the same length (a lane per block made the first wavefront walk 30-item runs while the others idled; a quad per
// block paid a 216-instruction DPP sum for every one-item block); the lanes of a block are joined by one or two DPP steps, its first lane stores.
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_split_linearise_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P)
{
	__shared__ double s_part[NW];
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // node records of the tile: wave w at s_dyn + 64 REC w
	SP_T0();
	const sp_tile T = S.tile_map[blockIdx.x];
	if (T.wi < 0) return;
	const int wi = T.wi;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, quad = tid >> 2, ql = tid & 3;
	// one level of loads behind the tile record: loop state, window headers, the wavefront's group, the first lane slot, the quad's gradient slot
	const sp_ctl c = S.ctl[wi];
	const nd_win W = N.win[wi];
	const nd_split SW = S.sw[wi];
	const nd_group G = N.group[T.grp + min(wave, max(T.n_grp - 1, 0))];
	nd_i4 q0 = nd_i4{0, 0, 0, 0};
	if (tid < T.n_slots) q0 = S.sqs[T.sqs + tid];
	if (!c.active) return;
	const bool pre = c.pre != 0;
	SP_T(0);
	const double lambda = c.lambda;
	const double* tab_cur = N.tab + W.tab0 + (c.cur ? 12 * (size_t)W.n_suf : 0);
	double* reg = s_dyn;
	double* stage = reg + 64 * REC * wave;
	double vmax = 0.0;
	if (wave < T.n_grp) {
		const nd_recs rc = nd_fetch_recs(N, W, G, lane);
#ifdef SSK_LM_STAMPS
		if (tid == 0 && !pre) { (void)__builtin_amdgcn_readfirstlane(rc.lrec.x + rc.orec.x); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_sp_t[2], t_ - sp_t_); sp_t_ = t_; }
#endif
		nd_linearise_group<true>(D, N, W, P, G, rc, tab_cur, stage, lambda, pre, lane, vmax);
	}
	if (pre) {
		// (the lambda_0 pass) the wavefront's largest landmark-block diagonal: into LDS now, a register held to the kernel's end was spilled
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) vmax = fmax(vmax, __shfl_down(vmax, o, 64));
		if (lane == 0) s_part[wave] = vmax;
	}
	SP_T(3);
	// item codes of the first lane slot and of the quad's gradient slot: in flight across the barrier
	nd_i2 gq = nd_i2{0, 0};
	int gs = 0;
	if (quad < T.n_gqs) { gq = N.gqs[T.gqs + quad]; gs = S.gslot[T.gslot + quad]; }
	const int* items = N.item + T.item0;
	nd_items its = nd_items_fetch(items + q0.z, q0.y & 0xFFFF, (q0.y >> 20) & 3, 1 << ((q0.y >> 16) & 3));
	const int gn = (gq.x >> 16) & 0xFFFF;
	const int* gitems = N.gitem + T.gitem0 + gq.y;
	int gi[ND_ITEMS_REG];
#pragma unroll
	for (int k2 = 0; k2 < ND_ITEMS_REG; ++k2) gi[k2] = gitems[min(ql + 4 * k2, max(gn - 1, 0))];
	__syncthreads();
	SP_T(4);
	if (pre && tid == 0) {
		double mx = s_part[0];
#pragma unroll
		for (int w = 1; w < NW; ++w) mx = fmax(mx, s_part[w]);
		S.vpart[T.gtile] = mx;
	}
	if (!pre)
		for (int k = quad; k < T.n_gqs; k += NQ) {
			// reduced gradient of one edge: sum over the tile's nodes on it of [vh; q x vh]
			if (k != quad) {
				gq = N.gqs[T.gqs + k]; gs = S.gslot[T.gslot + k];
			}
			const int n = (gq.x >> 16) & 0xFFFF;
			const int* gis = N.gitem + T.gitem0 + gq.y;
			if (k != quad) {
#pragma unroll
				for (int k2 = 0; k2 < ND_ITEMS_REG; ++k2) gi[k2] = gis[min(ql + 4 * k2, n - 1)];
			}
			double ga[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int k2 = 0; k2 < ND_ITEMS_REG; ++k2)
				if (ql + 4 * k2 < n) {
					const double* rcd = reg + REC * gi[k2];
					const double vh[3] = {rcd[R_V], rcd[R_V + 1], rcd[R_V + 2]}, qq[3] = {rcd[R_Q], rcd[R_Q + 1], rcd[R_Q + 2]};
					double x[3];
					nd_cross(qq, vh, x);
#pragma unroll
					for (int i2 = 0; i2 < 3; ++i2) { ga[i2] += vh[i2]; ga[3 + i2] += x[i2]; }
				}
			for (int i = ql + 4 * ND_ITEMS_REG; i < n; i += 4) {
				const double* rcd = reg + REC * gis[i];
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
				double* o = S.gpart + 6 * (size_t)(SW.gpart0 + gs);
#pragma unroll
				for (int i = 0; i < 6; ++i) o[i] = ga[i];
			}
		}
	SP_T(5);
	for (int k = tid; k < T.n_slots; k += NT) {
		const nd_i4 q = k == tid ? q0 : S.sqs[T.sqs + k];
		const int n = q.y & 0xFFFF, kc = (q.y >> 16) & 3, rank = (q.y >> 20) & 3, step = 1 << kc;
		if (k != tid) its = nd_items_fetch(items + q.z, n, rank, step);
		// the lambda_0 pass only needs the diagonal blocks (bit 30 of x)
		if (pre && !((q.x >> 30) & 1)) continue;
		double acc[36];
#pragma unroll
		for (int i = 0; i < 36; ++i) acc[i] = 0.0;
		nd_pair_items(reg, items + q.z, its, n, rank, step, pre, acc);
		if ((__builtin_amdgcn_readfirstlane(q.x) >> 29) & 1) {
			// the lanes of a block: (l0 + l1) + (l2 + l3), a fixed order; lanes of one-lane blocks add 0
			const double m1 = kc >= 1 ? 1.0 : 0.0, m2 = kc == 2 ? 1.0 : 0.0;
#pragma unroll
			for (int i = 0; i < 36; ++i) acc[i] = fma(m1, dpp_f64<0xB1>(acc[i]), acc[i]);      // quad_perm [1,0,3,2]
#pragma unroll
			for (int i = 0; i < 36; ++i) acc[i] = fma(m2, dpp_f64<0x4E>(acc[i]), acc[i]);      // quad_perm [2,3,0,1]
		}
		if (rank == 0) {
			v2d_t* o = reinterpret_cast<v2d_t*>(S.Hpart + 36 * (size_t)(SW.part0 + q.w));
			// (every byte of the slot is written, also the pairs of a diagonal block that K2a does not read: leaving those six 16-byte pairs out made the
			// depth-5 batch 2.7 % SLOWER -- a line written in part is read from HBM when L2 evicts it, a line written whole is not)
#pragma unroll
			for (int i = 0; i < 18; ++i) { v2d_t x; x.x = acc[2 * i]; x.y = acc[2 * i + 1]; o[i] = x; }
#ifdef SP_K1_STORE2
			// diagnostics: every partial slot written twice (same address: the request count doubles, the HBM traffic does not)
			asm volatile("" ::: "memory");
#pragma unroll
			for (int i = 0; i < 18; ++i) { v2d_t x; x.x = acc[2 * i]; x.y = acc[2 * i + 1]; __builtin_nontemporal_store(x, o + i); }
#endif
		}
	}
	SP_T(6);
#ifdef SSK_LM_STAMPS
	if (tid == 0 && !pre) atomicAdd(&g_sp_t[15], 1ULL);
#endif
}

// sum of the 36-double (6-double) partial slots [a, b) at component c: eight loads in flight, adds in slot order
template <int PITCH>
__device__ __forceinline__ double sp_slot_sum(const double* __restrict__ p, int a, int b, int c)
{
	double s = 0.0;
	int k = a;
	for (; k + 8 <= b; k += 8) {
		double v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = p[(size_t)PITCH * (k + u) + c];
#pragma unroll
		for (int u = 0; u < 8; ++u) s += v[u];
	}
	if (k < b) {
		double v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = p[(size_t)PITCH * min(k + u, b - 1) + c];
#pragma unroll
		for (int u = 0; u < 8; ++u) s += k + u < b ? v[u] : 0.0;
	}
	return s;
}

// ---- K2a: the reduced system from its partial slots: workgroup (window, chunk of SP_RB blocks); a block's slots are contiguous and summed in slot
// (= tile) order; A = H_pp - S + lambda I, both triangles, border row = reduced gradient ----
constexpr int SP_RB = 7;                // blocks per workgroup: 7 x 36 = 252 of 256 threads
__global__ __launch_bounds__(NT)
void lm_split_reduce_kernel(lm_dev D, nd_dev N, sp_dev S)
{
	const int wi = S.win_map[blockIdx.x];
	if (wi < 0) return;
	const sp_ctl c = S.ctl[wi];
	if (!c.active || c.pre != 0) return;
	const nd_win W = N.win[wi];
	const int n_pairs = W.E * (W.E + 1) / 2;
	if ((int)blockIdx.y * SP_RB >= n_pairs && blockIdx.y != 0) return;
	const nd_split SW = S.sw[wi];
	const int tid = threadIdx.x;
	const int n6 = 6 * W.E, np = n6 | 1;
	double* A = D.Hred + W.h0;
	const int b = tid / 36, ab = tid - 36 * b, pid = (int)blockIdx.y * SP_RB + b;
	if (b < SP_RB && pid < n_pairs) {
		const int* poff = S.poff + SW.poff0;
		int e2 = (int)((sqrtf(8.0f * (float)pid + 1.0f) - 1.0f) * 0.5f);
		while ((e2 + 1) * (e2 + 2) / 2 <= pid) ++e2;
		while (e2 * (e2 + 1) / 2 > pid) --e2;
		const int e = pid - e2 * (e2 + 1) / 2, a = ab / 6, bb = ab - 6 * a;
		if (!(e == e2 && a > bb)) {                        // (below the diagonal of a diagonal block: the mirror of an element above; K1 does not even write all of them)
			double v = sp_slot_sum<36>(S.Hpart + 36 * (size_t)SW.part0, poff[pid], poff[pid + 1], ab);
			const int row = 6 * e + a, col = 6 * e2 + bb;
			if (row == col) v += c.lambda;
			A[(size_t)row * np + col] = v;
			// (the matrix-core solvers read the upper triangle -- rows along a tile's columns -- and the lower one only inside the diagonal 16 x 16
			// tiles: the mirror element elsewhere is 45 % of this kernel's writes, a column walk of 8-byte stores)
			if (!S.solve_mfma || (row >> 4) == (col >> 4)) A[(size_t)col * np + row] = v;
		}
	}
	if (blockIdx.y == 0) {
		const double* gp = S.gpart + 6 * (size_t)SW.gpart0;
		const int* goff = S.goff + SW.goff0;
		for (int i = tid; i < n6; i += NT) { const int e = i / 6; A[(size_t)n6 * np + i] = sp_slot_sum<6>(gp, goff[e], goff[e + 1], i - 6 * e); }
	}
}

// ---- K3: trial state of one tile: landmark steps + robust error (init != 0: the error of the initial state, no step) ----
#ifndef SP_K3_WAVES
#define SP_K3_WAVES 5                   // (34.74 ms per 1 024 depth-5 windows at 4 waves per SIMD, 34.50 at 5, 34.70 at 6, 40.0 at 8: spills)
#endif
__global__ __launch_bounds__(NT, SP_K3_WAVES)
void lm_split_score_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P_arg, int init)
{
	__shared__ lm_cam s_P;
	__shared__ double s_part[NW];
	__shared__ double s_stage[NW * 64 * 3];
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // the edge step (6E)
	const sp_tile T = S.tile_map[blockIdx.x];
	if (T.wi < 0) return;
	const int wi = T.wi;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const sp_ctl c = S.ctl[wi];
	const nd_win W = N.win[wi];
	const nd_group G = N.group[T.grp + min(wave, max(T.n_grp - 1, 0))];
	if (!c.active) return;
	if (!init && !c.ok) return;
	if (threadIdx.x == 0) s_P = P_arg;
	double* s_b = s_dyn;
	if (!init)
		for (int i = tid; i < 6 * W.E; i += NT) s_b[i] = D.dp[6 * (size_t)W.us0 + i];
	__syncthreads();
	const lm_cam& P = s_P;
	const double* tab = N.tab + W.tab0 + ((init ? c.cur : (c.cur ^ 1)) ? 12 * (size_t)W.n_suf : 0);
	double e = 0.0, dfn = 0.0;
	if (wave < T.n_grp) {
		const nd_recs rc = nd_fetch_recs(N, W, G, lane);
		nd_score_group(D, N, W, P, G, rc, tab, s_stage + 64 * 3 * wave, s_b, !init, c.iters, lane, e, dfn);
	}
	e = block_sum(e, s_part);
	dfn = block_sum(dfn, s_part);
	if (tid == 0) { S.epart[T.gtile] = e; S.fpart[T.gtile] = dfn; }
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
	// the tiles' partials: a thread per tile in turn, then the fixed shuffle tree of block_sum -- the same sum in every launch (one thread walking
	// a local area's hundred tiles by dependent loads was 26 us per pass of the un-batched path)
	__shared__ double s_part[NW];
	double e_all = 0.0, dfn_all = 0.0;
	if (init || c0.ok) {
		double e1 = 0.0, f1 = 0.0;
		for (int t = tid; t < W.n_tiles; t += NT) { e1 += S.epart[SW.gtile0 + t]; f1 += S.fpart[SW.gtile0 + t]; }
		e_all = block_sum(e1, s_part);
		dfn_all = block_sum(f1, s_part);
	}
	if (tid == 0) {
		sp_ctl c = c0;
		const double e = e_all, dfn = dfn_all;
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
		if (!c.active) { R.iters = c.iters; R.accepted = c.accepted; R.stop_reason = c.stop; R.err_final = c.err; R.lambda_final = c.lambda; atomicSub(S.n_active, 1); }
		S.ctl[wi] = c;
		s_copy = copy;
	}
	__syncthreads();
	if (s_copy) {
		// accepted: the trial state becomes the state (the tables swapped by `cur`)
		for (int e = tid; e < W.E; e += NT) {
			const int ei = D.slot_edge[W.us0 + e];
			double v[12];
#pragma unroll
			for (int k = 0; k < 12; ++k) v[k] = D.pose_new[12 * (size_t)(W.edge0 + ei) + k];
#pragma unroll
			for (int k = 0; k < 12; ++k) D.pose[12 * (size_t)(W.edge0 + ei) + k] = v[k];
		}
		// (four landmarks per thread at a time, every load before the first store: the compiler cannot tell that lm and lm_new do not overlap, and a
		//  local area's 4 000 landmarks copied by dependent load -> store pairs cost ~15 us per accepted step of the un-batched path)
		const int* __restrict__ slot_lm = D.slot_lm + W.ls0;
		const double* __restrict__ src = D.lm_new + 3 * (size_t)W.lm0;
		double* __restrict__ dst = D.lm + 3 * (size_t)W.lm0;
		for (int l0 = tid; l0 < W.L; l0 += 4 * NT) {
			int li[4];
			double v[4][3];
#pragma unroll
			for (int u = 0; u < 4; ++u) li[u] = slot_lm[min(l0 + u * NT, W.L - 1)];
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int k = 0; k < 3; ++k) v[u][k] = src[3 * (size_t)li[u] + k];
#pragma unroll
			for (int u = 0; u < 4; ++u)
				if (l0 + u * NT < W.L) {
#pragma unroll
					for (int k = 0; k < 3; ++k) dst[3 * (size_t)li[u] + k] = v[u][k];
				}
		}
	}
}

}  // namespace

#ifdef SSK_LM_STAMPS
extern "C" int ssk_lm_split_stamps(unsigned long long* out16, int reset)
{
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sp_t), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
	if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sp_t), z, sizeof z) != hipSuccess) return 1; }
	return 0;
}
#endif

int ssk_lm_launch_split(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const sp_dev& S, const lm_cam& P, int n_windows, size_t solve_dyn_bytes, size_t init_dyn_bytes)
{
	(void)n_windows;
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_linearise_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		attr_once.done(ctx->device);
	}
	ssk_stage_scope prof(ctx, SSK_STAGE_LM_SOLVE);
	const size_t lin_lds = sizeof(double) * (size_t)NDW * 64 * REC;
	const size_t small_lds = (init_dyn_bytes + 15) & ~(size_t)15;      // K0: the window's poses and trie records
	const dim3 gt((unsigned)S.n_tile_wgs), gw((unsigned)S.n_win_wgs), bt(NT);
	SSK_HIP(ctx, hipMemsetAsync(S.n_active, 0, sizeof(int), ctx->stream));
	hipLaunchKernelGGL(lm_split_init_kernel, gw, bt, small_lds, ctx->stream, D, N, S);
	hipLaunchKernelGGL(lm_split_score_kernel, gt, bt, 0, ctx->stream, D, N, S, P, 1);
	hipLaunchKernelGGL(lm_split_decide_kernel, gw, bt, 0, ctx->stream, D, N, S, P, 1);
	SSK_LAUNCH_CHECK(ctx, "lm_split_init/score/decide");
	// A small batch (the un-batched drop-in: one local area per keyframe) is latency bound: after the passes a solve usually takes the host asks
	// how many windows still run and stops enqueueing when none does (a window's workgroups leave at once in a pass it does not need, but five
	// launches per pass are not free).  A large batch is enqueued whole: ssk_lm_batch_run stays asynchronous there.
	const bool poll = n_windows <= 8 && !ctx->env.sp_no_poll;
	int* h_poll = nullptr;
	if (poll) { if (ssk_arena_reserve(ctx, ctx->pin, 256)) return SSK_ERR_HIP; h_poll = reinterpret_cast<int*>(ctx->pin.base); }
	int next_poll = 10;
	for (int pass = 0; pass <= P.max_iters; ++pass) {
		hipLaunchKernelGGL(lm_split_linearise_kernel, gt, bt, lin_lds, ctx->stream, D, N, S, P);
		if (pass > 0) hipLaunchKernelGGL(lm_split_reduce_kernel, dim3((unsigned)S.n_win_wgs, (unsigned)std::max(1, (S.max_pairs + SP_RB - 1) / SP_RB)), bt, 0, ctx->stream, D, N, S);
		const int rc = ssk_lm_launch_split_solve(ctx, D, N, S, P, solve_dyn_bytes);
		if (rc) return rc;
		if (pass == 0) continue;                           // every window's first pass is its lambda_0 pass: no attempt to score
		hipLaunchKernelGGL(lm_split_score_kernel, gt, bt, sizeof(double) * 6 * 128, ctx->stream, D, N, S, P, 0);      // (dynamic LDS: the edge step, 6E doubles)
		hipLaunchKernelGGL(lm_split_decide_kernel, gw, bt, 0, ctx->stream, D, N, S, P, 0);
		if (poll && pass >= next_poll && pass < P.max_iters) {
			SSK_HIP(ctx, hipMemcpyAsync(h_poll, S.n_active, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
			SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
			if (*h_poll <= 0) break;
			next_poll = pass + 2;
		}
	}
	SSK_LAUNCH_CHECK(ctx, "lm_split kernels");
	return SSK_OK;
}
