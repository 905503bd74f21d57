// ssk_lm_split_solve.hip -- K2 of the phase-split LM solve (ssk_lm_split.h): one 512-thread workgroup per window sums the partial slots of
// every 6x6 block in slot (= tile) order into the dense bordered system, factorises it (LDL^T, register tiles: ldl_solve_regtile of
// ssk_lm_device.h -- the solver in a kernel of its own), applies the edge step exp(eps) (+) T (contract L3 of oracle/oracle_lm.c) and builds
// the trial state's suffix table.  Reference: solver_LM_schur_dense_cholesky, SRBA.h:34; call EST.cpp:782-784.
#include "ssk_lm_split.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

#pragma clang fp contract(fast)

namespace {

#define SSK_LM_NT 512
#undef SSK_LM_WAVES
#define SSK_LM_WAVES 2
#include "ssk_lm_device.h"

constexpr int REC = SSK_ND_REC, STG = SSK_ND_STAGE, LREC = SSK_ND_LREC, SCR = SSK_ND_SCR;
constexpr int R_Q = 9, R_P = 12, R_V = 21;
#include "ssk_lm_node_dev.h"
#include "ssk_lm_ldl_mfma.h"
#ifndef SSK_K2_MFMA
#define SSK_K2_MFMA 1                    // 1: ldl_solve_mfma_tiles (the matrix in MFMA accumulator tiles) for systems up to 255 unknowns, 0: SSK_LDL_BIG always (same-box A/B: tools/split_variant.sh)
#endif

// sum of the 36-double (6-double) partial slots [a, b) at component c: eight loads in flight, adds in slot order
template <int PITCH>
__device__ __forceinline__ double slot_sum(const double* __restrict__ p, int a, int b, int c)
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

// MODE (the host's choice, sp_dev::solve_mfma: the solvers live in three instances of the kernel, each with its own register budget):
// 0 the register-tile solver; 1 matrix cores, windows up to LDLM_LEAD_NMAX unknowns (leading columns + the tile solver); 2 matrix cores, every window
// of the batch within the tile solver's 255 unknowns (the leading columns' code beside it cost the tile solver 12 us a pass in registers: depth-5 A/B)
template <int MODE>
__global__ __launch_bounds__(NT, SSK_LM_WAVES)
void lm_split_solve_kernel(lm_dev D, nd_dev N, sp_dev S, lm_cam P, int dyn_doubles)
{
	extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // [delta | pivots | poses | trie records | solver staging]
	__shared__ double s_part[NW];
	const int wi = S.win_map[blockIdx.x];
	if (wi < 0) return;
	const sp_ctl c = S.ctl[wi];
	if (!c.active) return;
	const nd_win W = N.win[wi];
	const nd_split SW = S.sw[wi];
	const int tid = threadIdx.x;
	const int n6 = 6 * W.E, n6p = (n6 + 1) & ~1, np = n6 | 1;
	const double* Hp = S.Hpart + 36 * (size_t)SW.part0;
	const int* poff = S.poff + SW.poff0;
	if (c.pre != 0) {
		// lambda_0 = tau * max diag(H): the diagonal blocks of H_pp and the landmark blocks' diagonals (contract L5)
		double mx = 0.0;
		for (int t = tid; t < W.n_tiles; t += NT) mx = fmax(mx, S.vpart[SW.gtile0 + t]);
		for (int it = tid; it < n6; it += NT) {
			const int e = it / 6, a = it - 6 * e, pid = e * (e + 1) / 2 + e;
			mx = fmax(mx, slot_sum<36>(Hp, poff[pid], poff[pid + 1], 7 * a));
		}
		mx = block_max(mx, s_part);
		if (tid == 0) { S.ctl[wi].lambda = P.tau * mx; S.ctl[wi].pre = 0; }
		return;
	}
	double* s_b = s_dyn;
	double* s_inv = s_b + n6p;
	double* lds = s_inv + n6p;                           // the solver's staging; AFTER the solve: the window's edge poses and trie records
	const int lds_doubles = dyn_doubles - (int)(lds - s_dyn);
	// (round 5: poses and trie records OVERLAY the staging -- the solve needs neither, the edge step and the table need no staging --: a local area of
	// 70 edges / 1 000 suffixes beside the matrix-core solver's 138 KB did not fit, and such windows fell back to the register-tile solver)
	double* pose = lds;                                  // the window's edge poses: current, then trial
	nd_i2* s_suf = reinterpret_cast<nd_i2*>(pose + 12 * W.n_edges);
	// (the reduced system was assembled by lm_split_reduce_kernel: A = H_pp - S + lambda I, both triangles, border row = reduced gradient)
	double* A = D.Hred + W.h0;
	bool ok = true;
	if (n6 > 0) {
		if constexpr (MODE == 1) ok = ldl_solve_mfma_lead(A, np, n6, s_b, s_inv, lds, lds_doubles);
		else if constexpr (MODE == 2) ok = ldl_solve_mfma_tiles(A, np, n6, s_b, lds);
		else ok = SSK_LDL_BIG(A, np, n6, s_b, s_inv, lds, lds_doubles);
	}
	__syncthreads();
	if (ok) {
		for (int i = tid; i < 12 * W.n_edges; i += NT) pose[i] = D.pose[12 * (size_t)W.edge0 + i];
		for (int i = tid; i < W.n_suf; i += NT) s_suf[i] = N.suf[W.suf0 + i];
	}
	__syncthreads();
	double dpn = 0.0;
	if (ok) {
		for (int i = tid; i < n6; i += NT) {
			dpn += s_b[i] * s_b[i];
			D.dp[6 * (size_t)W.us0 + i] = s_b[i];
			if (D.trace_on) D.dp_trace[W.dp_trace0 + (size_t)c.iters * n6 + i] = s_b[i];
		}
		// edges: T_new = exp(eps) (+) T
		for (int e = tid; e < W.E; e += NT) {
			const int ei = D.slot_edge[W.us0 + e];
			double* T = pose + 12 * ei;
			double eps[6], R[9], O[12];
#pragma unroll
			for (int a = 0; a < 6; ++a) eps[a] = s_b[6 * e + a];
			so3_exp_inl(eps + 3, R);
#pragma unroll
			for (int i = 0; i < 3; ++i) {
#pragma unroll
				for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
				O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + eps[i];
			}
#pragma unroll
			for (int k = 0; k < 12; ++k) { T[k] = O[k]; D.pose_new[12 * (size_t)(W.edge0 + ei) + k] = O[k]; }
		}
		__syncthreads();
		nd_build_table(W, s_suf, pose, N.tab + W.tab0 + (c.cur ? 0 : 12 * (size_t)W.n_suf));
	}
	dpn = block_sum(dpn, s_part);
	if (tid == 0) { S.ctl[wi].ok = ok ? 1 : 0; S.ctl[wi].dpn = dpn; }
}

}  // namespace

#ifdef SSK_LM_STAMPS
extern "C" int ssk_lm_k2_stamps(unsigned long long* out16, int reset)
{
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_k2_t), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
	if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_k2_t), z, sizeof z) != hipSuccess) return 1; }
	return 0;
}
#endif

int ssk_sp_solve_mfma(int E_max, bool force_regtile)
{
	if (!SSK_K2_MFMA || force_regtile || 6 * E_max > LDLM_LEAD_NMAX) return 0;
	return 6 * E_max <= LDLM_NMAX ? 2 : 1;
}

size_t ssk_sp_solve_lds_doubles(int n_edges, int n_suf, int E, bool mfma)
{
	const size_t n6p = ((size_t)6 * E + 1) & ~(size_t)1;
	// the solver's staging (ldl_solve_mfma_tiles: LDLM_DOUBLES = 17 680 doubles, ldl_solve_regtile: L_TOTAL = 8 080) + a little head room
	return 2 * n6p + std::max(12 * (size_t)n_edges + (((size_t)n_suf + 1) & ~(size_t)1), (size_t)(mfma ? LDLM_DOUBLES + 256 : 8192));      // (poses + trie records overlay the staging)
}

int ssk_lm_launch_split_solve(ssk_ctx* ctx, const lm_dev& D, const nd_dev& N, const sp_dev& S, const lm_cam& P, size_t dyn_bytes)
{
	static ssk_dev_once attr_once;      // (the attribute is per device)
	if (attr_once.need(ctx->device)) {
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_solve_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_solve_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		SSK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lm_split_solve_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
		attr_once.done(ctx->device);
	}
	dyn_bytes = (dyn_bytes + 15) & ~(size_t)15;
	if (S.solve_mfma == 2)
		hipLaunchKernelGGL(lm_split_solve_kernel<2>, dim3((unsigned)S.n_win_wgs), dim3(NT), dyn_bytes, ctx->stream, D, N, S, P, (int)(dyn_bytes / 8));
	else if (S.solve_mfma == 1)
		hipLaunchKernelGGL(lm_split_solve_kernel<1>, dim3((unsigned)S.n_win_wgs), dim3(NT), dyn_bytes, ctx->stream, D, N, S, P, (int)(dyn_bytes / 8));
	else
		hipLaunchKernelGGL(lm_split_solve_kernel<0>, dim3((unsigned)S.n_win_wgs), dim3(NT), dyn_bytes, ctx->stream, D, N, S, P, (int)(dyn_bytes / 8));
	return SSK_OK;
}
