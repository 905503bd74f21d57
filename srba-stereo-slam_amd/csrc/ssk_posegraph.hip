// ssk_posegraph.hip -- the global pose-graph optimisation the reference runs once after its frame loop:
//   rba.get_global_graphslam_problem(poseGraph); mrpt::graphslam::optimize_graph_spa_levmarq(poseGraph, ...)   (EST.cpp:945-957)
// Contract G1-G6: oracle/oracle_posegraph.c.  Levenberg-Marquardt on e = (t_E, log R_E), E = Z^-1 X_i^-1 X_j, unit
// information.  The step (H + lambda I) d = -g, H = sum_e P_e M_e P_e^T with P_e = E_j - E_i and M_e = J_e^T J_e:
//   * direct (default when at most SSK_PG_DIRECT_MAX_EDGES edges lie off the spanning tree -- a keyframe graph is a tree plus
//     a few loop closures): T = lambda I + the tree edges' terms is factorised by block elimination on the tree (no fill,
//     children before parents); the c other edges are a rank-6c term U U^T, U_e = P_e J_e^T, handled by Woodbury:
//     d = t - W y with t = T^-1 (-g), W = T^-1 U, (I + U^T W) y = U^T t.  The 6c + 1 right-hand sides go through the tree
//     side by side, one THREAD per column walking the elimination order by itself (no barrier inside a sweep);
//   * conjugate gradients preconditioned with the spanning-tree part of the system (the oracle's solver, G6), level by level
//     with a workgroup barrier per level: a chain-like graph of n nodes costs O(n) barriers per CG iteration and ~6c
//     iterations per attempt -- 13 s at KITTI-00 size where the direct path takes a fraction of a second.
//
// One persistent 256-thread workgroup runs the whole optimisation (a keyframe graph has a few thousand nodes and about as
// many edges; the state, ~1 KB per node, stays in L2): every phase is a loop over edges or over nodes, a node gathers its
// incident edges through a CSR list in ascending edge order -- the order in which the oracle accumulates -- so no atomics
// and no run-to-run variation; only the three dot products of a CG iteration are tree-reduced (f64, fixed tree).
#include "ssk_internal.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int PG_NT = 256, PG_NW = PG_NT / 64;

struct pg_dev {
	int n, m, fixed;
	double *X, *Xn, *e, *M, *ge, *D, *Lb, *g, *d, *r, *z, *p, *q;
	const int32_t *edge_ij, *inc_off, *inc;
	// spanning tree (breadth-first from the fixed node): free nodes by level, parents before children; per node its parent
	// (-1: none / the fixed node), the edge to it, and its children (descending breadth-first position: the oracle's order)
	int n_lvl;
	const int32_t *lvl_off, *order, *par, *par_e, *ch_off, *ch;
	double *S, *y;
	const double* Z;
	ssk_pg_result* res;
	// direct solver: eflag[e] = -1 for an edge inside T (tree edge, or an edge at the fixed node: diagonal only), else its index
	// among the n_nt edges off the tree; Dt = diagonal blocks of T (without lambda), Li = reciprocal Cholesky diagonals,
	// Jnt = the Jacobians of the edges off the tree, W = [6 n][n_col] right-hand sides / solutions (n_col = 6 n_nt + 1, the last
	// column is t), Sm = capacitance matrix [6 n_nt][6 n_nt], sv = its right-hand side / solution
	int direct, n_nt, n_col;
	const int32_t *eflag, *nt_edge;
	double *Dt, *Li, *Jnt, *W, *Sm, *sv;
};

__device__ __forceinline__ void se3_inv(const double* T, double* O)
{
#pragma unroll
	for (int i = 0; i < 3; ++i) {
#pragma unroll
		for (int j = 0; j < 3; ++j) O[4 * i + j] = T[4 * j + i];
		O[4 * i + 3] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
	}
}

__device__ __forceinline__ void se3_mul(const double* A, const double* B, double* O)
{
#pragma unroll
	for (int i = 0; i < 3; ++i) {
#pragma unroll
		for (int j = 0; j < 3; ++j) O[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
		O[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
	}
}

__device__ __forceinline__ void so3_log(const double* T, double* w)
{
	double c = 0.5 * (T[0] + T[5] + T[10] - 1.0);
	c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
	const double th = acos(c);
	const double k = th < 1e-6 ? 0.5 + th * th / 12.0 : th / (2.0 * sin(th));
	w[0] = k * (T[9] - T[6]); w[1] = k * (T[2] - T[8]); w[2] = k * (T[4] - T[1]);
}

__device__ void so3_exp3(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j)
			R[3 * i + j] = (i == j ? 1.0 : 0.0) + a * K[3 * i + j] + b * (K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j]);
}

// G2 (+ G4 when J != nullptr)
__device__ void edge_lin(const double* Xi, const double* Xj, const double* Z, double* e, double* J)
{
	double Zi[12], Xii[12], A[12], E[12];
	se3_inv(Z, Zi); se3_inv(Xi, Xii);
	se3_mul(Zi, Xii, A);
	se3_mul(A, Xj, E);
	e[0] = E[3]; e[1] = E[7]; e[2] = E[11];
	so3_log(E, e + 3);
	if (!J) return;
	const double* w = e + 3;
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
	const double c = th < 1e-4 ? 1.0 / 12.0 : 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
	const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double Jr[9];
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int b = 0; b < 3; ++b)
			Jr[3 * a + b] = (a == b ? 1.0 : 0.0) + 0.5 * W[3 * a + b] + c * (W[3 * a] * W[b] + W[3 * a + 1] * W[3 + b] + W[3 * a + 2] * W[6 + b]);
	const double TB[9] = {0, -Xj[11], Xj[7], Xj[11], 0, -Xj[3], -Xj[7], Xj[3], 0};
#pragma unroll
	for (int i = 0; i < 36; ++i) J[i] = 0;
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int b = 0; b < 3; ++b) {
			J[6 * a + b] = A[4 * a + b];
			J[6 * a + 3 + b] = -(A[4 * a] * TB[b] + A[4 * a + 1] * TB[3 + b] + A[4 * a + 2] * TB[6 + b]);
			J[6 * (a + 3) + 3 + b] = Jr[3 * a] * Xj[4 * b] + Jr[3 * a + 1] * Xj[4 * b + 1] + Jr[3 * a + 2] * Xj[4 * b + 2];
		}
}

__device__ double pg_block_sum(double v, double* s_part)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
	__syncthreads();
	double r = 0;
#pragma unroll
	for (int w = 0; w < PG_NW; ++w) r += s_part[w];
	return r;
}

__device__ double pg_block_max(double v, double* s_part)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
	__syncthreads();
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
	__syncthreads();
	double r = s_part[0];
#pragma unroll
	for (int w = 1; w < PG_NW; ++w) r = fmax(r, s_part[w]);
	return r;
}

__device__ double pg_cost(const pg_dev& G, const double* X, double* s_part)
{
	double c = 0;
	for (int k = threadIdx.x; k < G.m; k += PG_NT) {
		double e[6];
		edge_lin(X + 12 * (size_t)G.edge_ij[2 * k], X + 12 * (size_t)G.edge_ij[2 * k + 1], G.Z + 12 * (size_t)k, e, nullptr);
#pragma unroll
		for (int a = 0; a < 6; ++a) c += e[a] * e[a];
	}
	return pg_block_sum(c, s_part);
}

__device__ __forceinline__ void chol6_solve(const double* L, const double* b, double* x)
{
	double y[6];
#pragma unroll
	for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
#pragma unroll
	for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
}

__device__ __forceinline__ void mat6_vec(const double* __restrict__ A, const double* x, double* y)
{
#pragma unroll
	for (int a = 0; a < 6; ++a) {
		double s = 0;
#pragma unroll
		for (int b = 0; b < 6; ++b) s += A[6 * a + b] * x[b];
		y[a] = s;
	}
}

// x = (L L^T)^-1 b with the reciprocal diagonal in li (the direct solver's sweeps: no divisions)
__device__ __forceinline__ void chol6_solve_r(const double* __restrict__ L, const double* __restrict__ li, const double* b, double* x)
{
	double y[6];
#pragma unroll
	for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s * li[i]; }
#pragma unroll
	for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s * li[i]; }
}

// G6: S_k = D_k + lambda I - sum over children c of M_e S_c^-1 M_e, Cholesky factors into Lb; levels from the deepest up.
// Returns through *s_ok (cleared when a block is not positive definite).
__device__ void tree_factor(const pg_dev& G, const double* __restrict__ Dsrc, double lambda, int* s_ok)
{
	for (int lvl = G.n_lvl - 1; lvl >= 0; --lvl) {
		for (int idx = G.lvl_off[lvl] + threadIdx.x; idx < G.lvl_off[lvl + 1]; idx += PG_NT) {
			const int k = G.order[idx];
			double Sk[36];
#pragma unroll
			for (int i = 0; i < 36; ++i) Sk[i] = Dsrc[36 * (size_t)k + i];
#pragma unroll
			for (int a = 0; a < 6; ++a) Sk[7 * a] += lambda;
			for (int t = G.ch_off[k]; t < G.ch_off[k + 1]; ++t) {
				const int c = G.ch[t];
				const double* Me = G.M + 36 * (size_t)G.par_e[c];
				const double* Lc = G.Lb + 36 * (size_t)c;
				for (int col = 0; col < 6; ++col) {
					double mc[6], x[6], yv[6];
#pragma unroll
					for (int a = 0; a < 6; ++a) mc[a] = Me[6 * a + col];
					if (G.direct) chol6_solve_r(Lc, G.Li + 6 * (size_t)c, mc, x);      // (the CG path keeps the oracle's divisions)
					else chol6_solve(Lc, mc, x);
					mat6_vec(Me, x, yv);
#pragma unroll
					for (int a = 0; a < 6; ++a) Sk[6 * a + col] -= yv[a];
				}
			}
			bool ok = true;
			double li6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int j = 0; j < 6; ++j) {
				double dj = Sk[6 * j + j];
				for (int c = 0; c < j; ++c) dj -= Sk[6 * j + c] * Sk[6 * j + c];
				if (!(dj > 0.0)) { ok = false; dj = 1.0; }
				const double l = sqrt(dj);
				Sk[6 * j + j] = l;
				if (G.direct) {
					const double rl = 1.0 / l;
					li6[j] = rl;
					for (int i = j + 1; i < 6; ++i) {
						double sv = Sk[6 * i + j];
						for (int c = 0; c < j; ++c) sv -= Sk[6 * i + c] * Sk[6 * j + c];
						Sk[6 * i + j] = sv * rl;
					}
				} else {
					for (int i = j + 1; i < 6; ++i) {
						double sv = Sk[6 * i + j];
						for (int c = 0; c < j; ++c) sv -= Sk[6 * i + c] * Sk[6 * j + c];
						Sk[6 * i + j] = sv / l;
					}
				}
			}
			if (!ok) *s_ok = 0;
#pragma unroll
			for (int i = 0; i < 36; ++i) G.Lb[36 * (size_t)k + i] = Sk[i];
			if (G.direct) {
#pragma unroll
				for (int a = 0; a < 6; ++a) G.Li[6 * (size_t)k + a] = li6[a];
			}
		}
		__syncthreads();
	}
}

// tree_factor for the direct solver (Dt, reciprocal diagonals): EIGHT lanes per node, lane s < 6 owns column s of S_k, so the six
// triangular solves + products of a child run side by side instead of one after the other in a single thread (a chain-like tree has
// one or two nodes per level: the serial work per node is what a factorisation costs).  The block is then handed to every lane of
// the group by shuffles and factorised redundantly; lane 0 stores.  Same operations per entry as tree_factor.
__device__ void tree_factor_direct(const pg_dev& G, double lambda, int* s_ok)
{
	const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3, col = sub < 6 ? sub : 5;
	for (int lvl = G.n_lvl - 1; lvl >= 0; --lvl) {
		for (int idx = G.lvl_off[lvl] + grp; idx < G.lvl_off[lvl + 1]; idx += PG_NT / 8) {
			const int k = G.order[idx];
			double sc[6];
#pragma unroll
			for (int a = 0; a < 6; ++a) sc[a] = G.Dt[36 * (size_t)k + 6 * a + col] + (a == col ? lambda : 0.0);
			for (int t = G.ch_off[k]; t < G.ch_off[k + 1]; ++t) {
				const int c = G.ch[t];
				const double* Me = G.M + 36 * (size_t)G.par_e[c];
				double mc[6], x[6], yv[6];
#pragma unroll
				for (int a = 0; a < 6; ++a) mc[a] = Me[6 * a + col];
				chol6_solve_r(G.Lb + 36 * (size_t)c, G.Li + 6 * (size_t)c, mc, x);
				mat6_vec(Me, x, yv);
#pragma unroll
				for (int a = 0; a < 6; ++a) sc[a] -= yv[a];
			}
			double Sk[36];
#pragma unroll
			for (int a = 0; a < 6; ++a)
#pragma unroll
				for (int b = 0; b < 6; ++b) Sk[6 * a + b] = __shfl(sc[a], b, 8);
			bool ok = true;
			double li6[6];
#pragma unroll
			for (int j = 0; j < 6; ++j) {
				double dj = Sk[6 * j + j];
				for (int c = 0; c < j; ++c) dj -= Sk[6 * j + c] * Sk[6 * j + c];
				if (!(dj > 0.0)) { ok = false; dj = 1.0; }
				const double l = sqrt(dj), rl = 1.0 / l;
				Sk[6 * j + j] = l;
				li6[j] = rl;
				for (int i = j + 1; i < 6; ++i) {
					double sv = Sk[6 * i + j];
					for (int c = 0; c < j; ++c) sv -= Sk[6 * i + c] * Sk[6 * j + c];
					Sk[6 * i + j] = sv * rl;
				}
			}
			if (sub == 0) {
				if (!ok) *s_ok = 0;
#pragma unroll
				for (int i = 0; i < 36; ++i) G.Lb[36 * (size_t)k + i] = Sk[i];
#pragma unroll
				for (int a = 0; a < 6; ++a) G.Li[6 * (size_t)k + a] = li6[a];
			}
		}
		__syncthreads();
	}
}

// W <- T^-1 W for all columns at once: thread `col` owns one column and walks the elimination order by itself -- deepest
// nodes first on the way up (t_k = S_k^-1 y_k, y_parent += M_e t_k), parents first on the way down (z_k = t_k + S_k^-1 M_e z_p).
// No barrier: a column never reads another column.  Rows of the fixed node are never touched (they stay 0).
__device__ void tree_solve_columns(const pg_dev& G, int n_free)
{
	const int col = threadIdx.x;
	if (col >= G.n_col) return;
	const size_t nc = (size_t)G.n_col;
	double* __restrict__ Wc = G.W + col;
	for (int idx = n_free - 1; idx >= 0; --idx) {
		const int k = G.order[idx], p = G.par[k];
		double yk[6], tk[6];
#pragma unroll
		for (int a = 0; a < 6; ++a) yk[a] = Wc[(6 * (size_t)k + a) * nc];
		chol6_solve_r(G.Lb + 36 * (size_t)k, G.Li + 6 * (size_t)k, yk, tk);
#pragma unroll
		for (int a = 0; a < 6; ++a) Wc[(6 * (size_t)k + a) * nc] = tk[a];
		if (p >= 0) {
			double u[6];
			mat6_vec(G.M + 36 * (size_t)G.par_e[k], tk, u);
#pragma unroll
			for (int a = 0; a < 6; ++a) Wc[(6 * (size_t)p + a) * nc] += u[a];
		}
	}
	for (int idx = 0; idx < n_free; ++idx) {
		const int k = G.order[idx], p = G.par[k];
		if (p < 0) continue;
		double zp[6], u[6], v[6];
#pragma unroll
		for (int a = 0; a < 6; ++a) zp[a] = Wc[(6 * (size_t)p + a) * nc];
		mat6_vec(G.M + 36 * (size_t)G.par_e[k], zp, u);
		chol6_solve_r(G.Lb + 36 * (size_t)k, G.Li + 6 * (size_t)k, u, v);
#pragma unroll
		for (int a = 0; a < 6; ++a) Wc[(6 * (size_t)k + a) * nc] += v[a];
	}
}

// The direct step: G.d <- -(H + lambda I)^-1 g.  Needs G.M, G.Jnt, G.Dt, G.g of the current linearisation.
__device__ void direct_step(const pg_dev& G, double lambda, int* s_ok)
{
	const int tid = threadIdx.x;
	const int ns = 6 * G.n_nt, nc = G.n_col, n_free = G.lvl_off[G.n_lvl];
	tree_factor_direct(G, lambda, s_ok);
	if (!*s_ok) return;                                  // uniform: tree_factor ends with a barrier
	// right-hand sides: column 6 q + a = U_e[:, a] = +J_e[a, :] on node j, -J_e[a, :] on node i (edge e = nt_edge[q]); last = -g
	for (size_t k = tid; k < (size_t)6 * G.n * nc; k += PG_NT) G.W[k] = 0.0;
	__syncthreads();
	for (int k = tid; k < 6 * G.n; k += PG_NT) G.W[(size_t)k * nc + (nc - 1)] = (k / 6 == G.fixed) ? 0.0 : -G.g[k];
	for (int it = tid; it < 36 * G.n_nt; it += PG_NT) {
		const int q = it / 36, r = it - 36 * q, a = r / 6, b = r - 6 * a;
		const int ed = G.nt_edge[q], i = G.edge_ij[2 * ed], j = G.edge_ij[2 * ed + 1];
		const double v = G.Jnt[36 * (size_t)q + 6 * a + b];
		G.W[(6 * (size_t)j + b) * nc + 6 * q + a] = v;
		G.W[(6 * (size_t)i + b) * nc + 6 * q + a] = -v;
	}
	__syncthreads();
	tree_solve_columns(G, n_free);
	__syncthreads();
	if (ns > 0) {
		// capacitance matrix I + U^T W (symmetric positive definite) and its right-hand side U^T t
		for (int it = tid; it < ns * (ns + 1); it += PG_NT) {
			const int r = it / (ns + 1), cidx = it - r * (ns + 1);          // cidx == ns: the right-hand side (column of t)
			const int q = r / 6, a = r - 6 * q;
			const int ed = G.nt_edge[q], i = G.edge_ij[2 * ed], j = G.edge_ij[2 * ed + 1];
			const int wc = cidx == ns ? nc - 1 : cidx;
			double sacc = (cidx == r) ? 1.0 : 0.0;
#pragma unroll
			for (int b = 0; b < 6; ++b)
				sacc += G.Jnt[36 * (size_t)q + 6 * a + b] * (G.W[(6 * (size_t)j + b) * nc + wc] - G.W[(6 * (size_t)i + b) * nc + wc]);
			if (cidx == ns) G.sv[r] = sacc; else G.Sm[(size_t)r * ns + cidx] = sacc;
		}
		__syncthreads();
		// right-looking Cholesky of Sm in place (lower triangle), then the two triangular solves on sv
		for (int j = 0; j < ns; ++j) {
			const double djj = G.Sm[(size_t)j * ns + j];
			const double l = sqrt(djj > 0.0 ? djj : 1.0), li = 1.0 / l;
			if (!(djj > 0.0) && tid == 0) *s_ok = 0;
			__syncthreads();
			for (int i = j + tid; i < ns; i += PG_NT) G.Sm[(size_t)i * ns + j] = (i == j) ? l : G.Sm[(size_t)i * ns + j] * li;
			__syncthreads();
			const int rem = ns - j - 1;
			for (int it = tid; it < rem * rem; it += PG_NT) {
				const int a = it / rem, b = it - a * rem;
				if (b <= a) {
					const int i = j + 1 + a, k = j + 1 + b;
					G.Sm[(size_t)i * ns + k] -= G.Sm[(size_t)i * ns + j] * G.Sm[(size_t)k * ns + j];
				}
			}
			__syncthreads();
		}
		if (tid < 64) {
			// forward and backward substitution by one wave: lane-strided dot products, shuffle reduction
			for (int i = 0; i < ns; ++i) {
				double acc = 0;
				for (int k = tid; k < i; k += 64) acc += G.Sm[(size_t)i * ns + k] * G.sv[k];
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
				if (tid == 0) G.sv[i] = (G.sv[i] - acc) / G.Sm[(size_t)i * ns + i];
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			}
			for (int i = ns - 1; i >= 0; --i) {
				double acc = 0;
				for (int k = i + 1 + tid; k < ns; k += 64) acc += G.Sm[(size_t)k * ns + i] * G.sv[k];
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
				if (tid == 0) G.sv[i] = (G.sv[i] - acc) / G.Sm[(size_t)i * ns + i];
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			}
		}
		__syncthreads();
	}
	// d = t - W y
	for (int k = tid; k < 6 * G.n; k += PG_NT) {
		const double* __restrict__ wr = G.W + (size_t)k * nc;
		double acc = wr[nc - 1];
		for (int c = 0; c < ns; ++c) acc -= wr[c] * G.sv[c];
		G.d[k] = acc;
	}
	__syncthreads();
}

// z = M^-1 r on the free nodes (z of the fixed node must be 0 on entry and stays 0); y: work vector
__device__ void tree_apply(const pg_dev& G, const double* __restrict__ r, double* __restrict__ z)
{
	for (int lvl = G.n_lvl - 1; lvl >= 0; --lvl) {          // forward: y_k = r_k + sum_c M_e t_c, t_k = S_k^-1 y_k
		for (int idx = G.lvl_off[lvl] + threadIdx.x; idx < G.lvl_off[lvl + 1]; idx += PG_NT) {
			const int k = G.order[idx];
			double yk[6], tk[6];
#pragma unroll
			for (int a = 0; a < 6; ++a) yk[a] = r[6 * (size_t)k + a];
			for (int t = G.ch_off[k]; t < G.ch_off[k + 1]; ++t) {
				const int c = G.ch[t];
				double tc[6], u[6];
#pragma unroll
				for (int a = 0; a < 6; ++a) tc[a] = z[6 * (size_t)c + a];
				mat6_vec(G.M + 36 * (size_t)G.par_e[c], tc, u);
#pragma unroll
				for (int a = 0; a < 6; ++a) yk[a] += u[a];
			}
			chol6_solve(G.Lb + 36 * (size_t)k, yk, tk);
#pragma unroll
			for (int a = 0; a < 6; ++a) z[6 * (size_t)k + a] = tk[a];
		}
		__syncthreads();
	}
	for (int lvl = 1; lvl < G.n_lvl; ++lvl) {               // backward: z_k = t_k + S_k^-1 M_e z_p (level 0 has no free parent)
		for (int idx = G.lvl_off[lvl] + threadIdx.x; idx < G.lvl_off[lvl + 1]; idx += PG_NT) {
			const int k = G.order[idx], p = G.par[k];
			if (p < 0) continue;
			double zp[6], u[6], v[6];
#pragma unroll
			for (int a = 0; a < 6; ++a) zp[a] = z[6 * (size_t)p + a];
			mat6_vec(G.M + 36 * (size_t)G.par_e[k], zp, u);
			chol6_solve(G.Lb + 36 * (size_t)k, u, v);
#pragma unroll
			for (int a = 0; a < 6; ++a) z[6 * (size_t)k + a] += v[a];
		}
		__syncthreads();
	}
}

__global__ __launch_bounds__(PG_NT)
void pg_solve_kernel(pg_dev G, ssk_pg_params P)
{
	__shared__ double s_part[PG_NW];
	__shared__ double s_lambda, s_nu, s_cost;
	__shared__ int s_stop, s_iters, s_accepted, s_ok, s_cg, s_run;
	const int tid = threadIdx.x;
	const double cost0 = pg_cost(G, G.X, s_part);
	if (tid == 0) {
		s_lambda = -1.0; s_nu = 2.0; s_cost = cost0; s_iters = 0; s_accepted = 0; s_cg = 0;
		s_stop = (G.n <= 1 || G.m == 0) ? 3 : -1;
		G.res->cost_init = cost0;
	}
	__syncthreads();
	while (s_stop < 0) {
		__syncthreads();
		if (s_iters >= P.max_iters) { if (tid == 0) s_stop = 0; break; }
		if (!(s_cost > 1e-22 * G.m)) { if (tid == 0) s_stop = 1; break; }
		// ---- linearise: per edge e, M = J^T J, ge = J^T e ----
		for (int k = tid; k < G.m; k += PG_NT) {
			double e[6], J[36];
			edge_lin(G.X + 12 * (size_t)G.edge_ij[2 * k], G.X + 12 * (size_t)G.edge_ij[2 * k + 1], G.Z + 12 * (size_t)k, e, J);
			double* Mk = G.M + 36 * (size_t)k;
			if (G.direct && G.eflag[k] >= 0) {
#pragma unroll
				for (int i = 0; i < 36; ++i) G.Jnt[36 * (size_t)G.eflag[k] + i] = J[i];
			}
#pragma unroll
			for (int a = 0; a < 6; ++a) {
				double ga = 0;
#pragma unroll
				for (int c = 0; c < 6; ++c) ga += J[6 * c + a] * e[c];
				G.ge[6 * (size_t)k + a] = ga;
#pragma unroll
				for (int b = 0; b < 6; ++b) {
					double s = 0;
#pragma unroll
					for (int c = 0; c < 6; ++c) s += J[6 * c + a] * J[6 * c + b];
					Mk[6 * a + b] = s;
				}
			}
		}
		__syncthreads();
		// ---- per node: diagonal block and gradient from the incident edges, ascending edge order ----
		double mx = 0;
		for (int k = tid; k < G.n; k += PG_NT) {
			double Dk[36], gk[6];
#pragma unroll
			for (int i = 0; i < 36; ++i) Dk[i] = 0;
#pragma unroll
			for (int a = 0; a < 6; ++a) gk[a] = 0;
			if (G.direct) {
				// the diagonal of T: the incident edges that are inside T (same ascending order)
				double Tk[36];
#pragma unroll
				for (int i = 0; i < 36; ++i) Tk[i] = 0;
				for (int t = G.inc_off[k]; t < G.inc_off[k + 1]; ++t) {
					const int ed = G.inc[t] >> 1;
					if (G.eflag[ed] >= 0) continue;
					const double* Mk = G.M + 36 * (size_t)ed;
#pragma unroll
					for (int i = 0; i < 36; ++i) Tk[i] += Mk[i];
				}
#pragma unroll
				for (int i = 0; i < 36; ++i) G.Dt[36 * (size_t)k + i] = Tk[i];
			}
			for (int t = G.inc_off[k]; t < G.inc_off[k + 1]; ++t) {
				const int ed = G.inc[t] >> 1, role = G.inc[t] & 1;            // role 1: the node is j of the edge
				const double* Mk = G.M + 36 * (size_t)ed;
#pragma unroll
				for (int i = 0; i < 36; ++i) Dk[i] += Mk[i];
#pragma unroll
				for (int a = 0; a < 6; ++a) gk[a] = role ? gk[a] + G.ge[6 * (size_t)ed + a] : gk[a] - G.ge[6 * (size_t)ed + a];
			}
#pragma unroll
			for (int i = 0; i < 36; ++i) G.D[36 * (size_t)k + i] = Dk[i];
#pragma unroll
			for (int a = 0; a < 6; ++a) { G.g[6 * (size_t)k + a] = gk[a]; if (k != G.fixed) mx = fmax(mx, Dk[7 * a]); }
		}
		mx = pg_block_max(mx, s_part);
		if (tid == 0) { if (s_lambda < 0) s_lambda = P.tau * mx; s_ok = 1; }
		__syncthreads();
		const double lambda = s_lambda;
		int cg_it = 0;
		double rz = 0, rz0 = 0;
		if (G.direct) {
			// ---- direct: tree elimination + Woodbury correction for the edges off the tree ----
			direct_step(G, lambda, &s_ok);
			__syncthreads();
		} else {
			// ---- G6: tree preconditioner, then CG on (H + lambda I) d = -g over the free nodes ----
			tree_factor(G, G.D, lambda, &s_ok);
			for (int k = tid; k < 6 * G.n; k += PG_NT) { G.d[k] = 0; G.z[k] = 0; G.r[k] = (k / 6 == G.fixed) ? 0.0 : -G.g[k]; }
			__syncthreads();
			tree_apply(G, G.r, G.z);
			double rz_l = 0;
			for (int k = tid; k < 6 * G.n; k += PG_NT) { const double zv = G.z[k]; G.p[k] = zv; rz_l += G.r[k] * zv; }
			rz = pg_block_sum(rz_l, s_part);
			rz0 = rz;
		}
		const bool ok = s_ok != 0;
		if (ok) {
			while (!G.direct && cg_it < P.pcg_max_iters && rz > P.pcg_tol * P.pcg_tol * rz0 && rz > 0) {
				double pq_l = 0;
				for (int k = tid; k < G.n; k += PG_NT) {
					double q6[6] = {0, 0, 0, 0, 0, 0};
					if (k != G.fixed) {
						double pk[6];
#pragma unroll
						for (int a = 0; a < 6; ++a) pk[a] = G.p[6 * (size_t)k + a];
#pragma unroll
						for (int a = 0; a < 6; ++a) {
							double s = lambda * pk[a];
#pragma unroll
							for (int b = 0; b < 6; ++b) s += G.D[36 * (size_t)k + 6 * a + b] * pk[b];
							q6[a] = s;
						}
						for (int t = G.inc_off[k]; t < G.inc_off[k + 1]; ++t) {
							const int ed = G.inc[t] >> 1, role = G.inc[t] & 1;
							const int other = G.edge_ij[2 * ed + (role ? 0 : 1)];
							const double* Mk = G.M + 36 * (size_t)ed;
							double po[6];
#pragma unroll
							for (int a = 0; a < 6; ++a) po[a] = G.p[6 * (size_t)other + a];
#pragma unroll
							for (int a = 0; a < 6; ++a) {
								double s = 0;
#pragma unroll
								for (int b = 0; b < 6; ++b) s += Mk[6 * a + b] * po[b];
								q6[a] -= s;
							}
						}
#pragma unroll
						for (int a = 0; a < 6; ++a) pq_l += pk[a] * q6[a];
					}
#pragma unroll
					for (int a = 0; a < 6; ++a) G.q[6 * (size_t)k + a] = q6[a];
				}
				const double pq = pg_block_sum(pq_l, s_part);
				if (!(pq > 0)) break;
				const double alpha = rz / pq;
				for (int k = tid; k < 6 * G.n; k += PG_NT) { G.d[k] += alpha * G.p[k]; G.r[k] -= alpha * G.q[k]; }
				__syncthreads();
				tree_apply(G, G.r, G.z);
				double rzn_l = 0;
				for (int k = tid; k < 6 * G.n; k += PG_NT) rzn_l += G.r[k] * G.z[k];
				const double rz_new = pg_block_sum(rzn_l, s_part);
				const double beta = rz_new / rz;
				for (int k = tid; k < 6 * G.n; k += PG_NT) G.p[k] = G.z[k] + beta * G.p[k];
				rz = rz_new;
				++cg_it;
				__syncthreads();
			}
			// ---- trial state (G3) ----
			for (int k = tid; k < G.n; k += PG_NT) {
				double R[9];
				const double* dk = G.d + 6 * (size_t)k;
				const double* T = G.X + 12 * (size_t)k;
				double* O = G.Xn + 12 * (size_t)k;
				so3_exp3(dk + 3, R);
#pragma unroll
				for (int i = 0; i < 3; ++i) {
#pragma unroll
					for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
					O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + dk[i];
				}
			}
			__syncthreads();
		}
		const double cost_new = ok ? pg_cost(G, G.Xn, s_part) : s_cost;
		double pred_l = 0;                                  // d.(lambda d - g): the decrease the linear model predicts
		if (ok) for (int k = tid; k < 6 * G.n; k += PG_NT) pred_l += G.d[k] * (lambda * G.d[k] - G.g[k]);
		const double pred = pg_block_sum(pred_l, s_part);
		__syncthreads();
		if (tid == 0) {
			const double cost = s_cost;
			const int acc = ok && cost_new < cost;
			const int converged = ok && P.converged_rel_decrease > 0 && fabs(cost - cost_new) <= P.converged_rel_decrease * cost;
			s_iters += 1; s_cg += cg_it;
			s_run = acc;
			if (acc) {
				const double rho = pred > 0 ? (cost - cost_new) / pred : 1.0, c3 = 2.0 * rho - 1.0;
				const double f = 1.0 - c3 * c3 * c3;
				s_cost = cost_new; s_lambda = lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0); s_nu = 2.0; s_accepted += 1;
				if (converged) s_stop = 4;
			} else if (converged) {
				s_stop = 4;
			} else {
				s_lambda = lambda * s_nu; s_nu *= 2.0;
				if (s_lambda > P.max_lambda) s_stop = 2;
			}
		}
		__syncthreads();
		if (s_run)
			for (int k = tid; k < 12 * G.n; k += PG_NT) G.X[k] = G.Xn[k];
		__syncthreads();
	}
	__syncthreads();
	if (tid == 0) {
		G.res->iters = s_iters; G.res->accepted = s_accepted; G.res->stop_reason = s_stop; G.res->pcg_iters_total = s_cg;
		G.res->cost_final = s_cost; G.res->lambda_final = s_lambda;
	}
}

}  // namespace

extern "C" int ssk_posegraph_optimize(ssk_ctx* ctx, int n_nodes, double* node_poses, int n_edges, const int32_t* edge_ij,
                                      const double* edge_z, int fixed_node, const ssk_pg_params* params, ssk_pg_result* out)
{
	if (!ctx) return SSK_ERR_BAD_ARG;
	if (n_nodes < 0 || n_edges < 0 || !params || !out || (n_nodes > 0 && !node_poses) || (n_edges > 0 && (!edge_ij || !edge_z)))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_posegraph_optimize: bad argument");
	if (n_nodes > 0 && (fixed_node < 0 || fixed_node >= n_nodes))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_posegraph_optimize: fixed_node %d outside [0, %d)", fixed_node, n_nodes);
	if (params->max_iters < 0 || params->pcg_max_iters < 1 || !(params->tau > 0) || !(params->pcg_tol > 0) || params->solver < 0 || params->solver > 2)
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_posegraph_optimize: max_iters >= 0, pcg_max_iters >= 1, tau > 0, pcg_tol > 0, solver in 0..2 required");
	for (int k = 0; k < n_edges; ++k) {
		const int i = edge_ij[2 * k], j = edge_ij[2 * k + 1];
		if (i < 0 || i >= n_nodes || j < 0 || j >= n_nodes || i == j)
			return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_posegraph_optimize: edge %d joins nodes %d and %d of %d", k, i, j, n_nodes);
	}
	std::memset(out, 0, sizeof *out);
	if (n_nodes == 0) { out->stop_reason = 3; return SSK_OK; }
	// incidence lists, ascending edge id per node
	const size_t n = (size_t)n_nodes, m = (size_t)n_edges;
	std::vector<int32_t> off(n + 1, 0), inc(2 * m + 1);
	for (size_t k = 0; k < m; ++k) { off[(size_t)edge_ij[2 * k] + 1]++; off[(size_t)edge_ij[2 * k + 1] + 1]++; }
	for (size_t k = 0; k < n; ++k) off[k + 1] += off[k];
	{
		std::vector<int32_t> cur(off.begin(), off.end() - 1);
		for (size_t k = 0; k < m; ++k) {
			inc[(size_t)cur[(size_t)edge_ij[2 * k]]++] = (int32_t)(k << 1);
			inc[(size_t)cur[(size_t)edge_ij[2 * k + 1]]++] = (int32_t)(k << 1 | 1);
		}
	}
	// breadth-first spanning tree from the fixed node (nodes and incident edges in ascending order); unreached nodes start
	// their own trees.  Levels: depth in the forest of FREE nodes (children of the fixed node are level 0).
	std::vector<int32_t> order, lvl_off, par(n, -1), par_e(n, -1), depth(n, -1), ch_off(n + 1, 0), ch;
	{
		std::vector<char> seen(n, 0);
		std::vector<int32_t> queue;
		queue.reserve(n);
		for (int start = -1; start < n_nodes; ++start) {
			const int root = start < 0 ? fixed_node : start;
			if (seen[(size_t)root]) continue;
			size_t qh = queue.size();
			queue.push_back(root); seen[(size_t)root] = 1;
			if (root != fixed_node) { order.push_back(root); depth[(size_t)root] = 0; }
			while (qh < queue.size()) {
				const int u = queue[qh++];
				for (int t = off[(size_t)u]; t < off[(size_t)u + 1]; ++t) {
					const int ed = inc[(size_t)t] >> 1, v = edge_ij[2 * ed] == u ? edge_ij[2 * ed + 1] : edge_ij[2 * ed];
					if (seen[(size_t)v]) continue;
					seen[(size_t)v] = 1; queue.push_back(v);
					order.push_back(v);
					if (u != fixed_node) { par[(size_t)v] = u; par_e[(size_t)v] = ed; depth[(size_t)v] = depth[(size_t)u] + 1; }
					else depth[(size_t)v] = 0;
				}
			}
		}
		// several trees interleave depths in `order`: sort by depth, keeping the breadth-first order inside a level
		std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return depth[(size_t)a] < depth[(size_t)b]; });
		int n_lvl = 0;
		for (int32_t k : order) n_lvl = std::max(n_lvl, depth[(size_t)k] + 1);
		lvl_off.assign((size_t)n_lvl + 1, 0);
		for (int32_t k : order) lvl_off[(size_t)depth[(size_t)k] + 1]++;
		for (int l = 0; l < n_lvl; ++l) lvl_off[(size_t)l + 1] += lvl_off[(size_t)l];
		// children of a node in DESCENDING position of `order` (the oracle walks the order backwards)
		for (int32_t k : order) if (par[(size_t)k] >= 0) ch_off[(size_t)par[(size_t)k] + 1]++;
		for (size_t k = 0; k < n; ++k) ch_off[k + 1] += ch_off[k];
		ch.assign((size_t)ch_off[n] + 1, 0);
		std::vector<int32_t> cur(ch_off.begin(), ch_off.end() - 1);
		for (size_t t = order.size(); t-- > 0;) { const int32_t k = order[t]; if (par[(size_t)k] >= 0) ch[(size_t)cur[(size_t)par[(size_t)k]]++] = k; }
	}
	const int n_lvl = (int)lvl_off.size() - 1;
	// edges inside T (tree edges; edges at the fixed node only add to a diagonal block) and the edges off the tree
	std::vector<int32_t> eflag(m + 1, 0), nt_edge;
	{
		std::vector<char> in_t(m + 1, 0);
		for (size_t k = 0; k < n; ++k) if (par_e[k] >= 0) in_t[(size_t)par_e[k]] = 1;
		for (size_t k = 0; k < m; ++k) {
			if (edge_ij[2 * k] == fixed_node || edge_ij[2 * k + 1] == fixed_node) in_t[k] = 1;
			if (in_t[k]) eflag[k] = -1;
			else { eflag[k] = (int32_t)nt_edge.size(); nt_edge.push_back((int32_t)k); }
		}
	}
	const size_t c_nt = nt_edge.size();
	const bool direct = params->solver != 1 && c_nt <= (size_t)SSK_PG_DIRECT_MAX_EDGES;
	if (params->solver == 2 && !direct)
		return ssk_set_error(ctx, SSK_ERR_CAPACITY, "ssk_posegraph_optimize: %zu edges off the spanning tree, the direct solver takes %d", c_nt, SSK_PG_DIRECT_MAX_EDGES);
	static_assert(6 * SSK_PG_DIRECT_MAX_EDGES + 1 <= PG_NT, "one thread per right-hand side");
	const size_t ns = direct ? 6 * c_nt : 0, ncol = ns + 1;
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const size_t m1 = m ? m : 1;
	const size_t dbl = 12 * n * 2 + 6 * m1 + 36 * m1 + 6 * m1 + 36 * n * 3 + 6 * n * 7 + 12 * m1;
	const size_t dbl_direct = direct ? 36 * n + 6 * n + 36 * (c_nt + 1) + 6 * n * ncol + (ns + 1) * (ns + 1) + ns + 1 : 0;
	const size_t bytes = ssk_pad(dbl * 8) + ssk_pad(dbl_direct * 8) + ssk_pad((m + 1) * 4) + ssk_pad((c_nt + 1) * 4) + ssk_pad(2 * m1 * 4) + ssk_pad((n + 1) * 4) * 8 +
	                     ssk_pad((2 * m + 1) * 4) + ssk_pad(sizeof(ssk_pg_result)) + 48 * 256;
	int rc;
	if ((rc = ssk_arena_reserve(ctx, ctx->dev, bytes))) return rc;
	ssk_arena_reset(ctx->dev);
	pg_dev G;
	G.n = n_nodes; G.m = n_edges; G.fixed = fixed_node;
	G.X = ssk_arena_take<double>(ctx->dev, 12 * n); G.Xn = ssk_arena_take<double>(ctx->dev, 12 * n);
	G.e = ssk_arena_take<double>(ctx->dev, 6 * m1); G.M = ssk_arena_take<double>(ctx->dev, 36 * m1); G.ge = ssk_arena_take<double>(ctx->dev, 6 * m1);
	G.D = ssk_arena_take<double>(ctx->dev, 36 * n); G.Lb = ssk_arena_take<double>(ctx->dev, 36 * n);
	G.g = ssk_arena_take<double>(ctx->dev, 6 * n); G.d = ssk_arena_take<double>(ctx->dev, 6 * n); G.r = ssk_arena_take<double>(ctx->dev, 6 * n);
	G.z = ssk_arena_take<double>(ctx->dev, 6 * n); G.p = ssk_arena_take<double>(ctx->dev, 6 * n); G.q = ssk_arena_take<double>(ctx->dev, 6 * n);
	double* d_Z = ssk_arena_take<double>(ctx->dev, 12 * m1);
	int32_t* d_ij = ssk_arena_take<int32_t>(ctx->dev, 2 * m1);
	int32_t* d_off = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_inc = ssk_arena_take<int32_t>(ctx->dev, 2 * m + 1);
	G.res = ssk_arena_take<ssk_pg_result>(ctx->dev, 1);
	G.Z = d_Z; G.edge_ij = d_ij; G.inc_off = d_off; G.inc = d_inc;
	G.S = ssk_arena_take<double>(ctx->dev, 36 * n); G.y = ssk_arena_take<double>(ctx->dev, 6 * n);
	int32_t* d_lvl = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_order = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_par = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_pare = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_choff = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	int32_t* d_ch = ssk_arena_take<int32_t>(ctx->dev, n + 1);
	G.n_lvl = n_lvl; G.lvl_off = d_lvl; G.order = d_order; G.par = d_par; G.par_e = d_pare; G.ch_off = d_choff; G.ch = d_ch;
	G.direct = direct ? 1 : 0; G.n_nt = direct ? (int)c_nt : 0; G.n_col = (int)ncol;
	G.eflag = nullptr; G.nt_edge = nullptr; G.Dt = G.Li = G.Jnt = G.W = G.Sm = G.sv = nullptr;
	if (direct) {
		int32_t* d_eflag = ssk_arena_take<int32_t>(ctx->dev, m + 1);
		int32_t* d_nt = ssk_arena_take<int32_t>(ctx->dev, c_nt + 1);
		G.Dt = ssk_arena_take<double>(ctx->dev, 36 * n); G.Li = ssk_arena_take<double>(ctx->dev, 6 * n);
		G.Jnt = ssk_arena_take<double>(ctx->dev, 36 * (c_nt + 1));
		G.W = ssk_arena_take<double>(ctx->dev, 6 * n * ncol);
		G.Sm = ssk_arena_take<double>(ctx->dev, (ns + 1) * (ns + 1)); G.sv = ssk_arena_take<double>(ctx->dev, ns + 1);
		G.eflag = d_eflag; G.nt_edge = d_nt;
		SSK_HIP(ctx, hipMemcpyAsync(d_eflag, eflag.data(), (m + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
		if (c_nt) SSK_HIP(ctx, hipMemcpyAsync(d_nt, nt_edge.data(), c_nt * 4, hipMemcpyHostToDevice, ctx->stream));
	}
	SSK_HIP(ctx, hipMemcpyAsync(d_lvl, lvl_off.data(), lvl_off.size() * 4, hipMemcpyHostToDevice, ctx->stream));
	if (!order.empty()) SSK_HIP(ctx, hipMemcpyAsync(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_par, par.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_pare, par_e.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_choff, ch_off.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(d_ch, ch.data(), ch.size() * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(G.X, node_poses, 12 * n * 8, hipMemcpyHostToDevice, ctx->stream));
	if (m) {
		SSK_HIP(ctx, hipMemcpyAsync(d_Z, edge_z, 12 * m * 8, hipMemcpyHostToDevice, ctx->stream));
		SSK_HIP(ctx, hipMemcpyAsync(d_ij, edge_ij, 2 * m * 4, hipMemcpyHostToDevice, ctx->stream));
		SSK_HIP(ctx, hipMemcpyAsync(d_inc, inc.data(), 2 * m * 4, hipMemcpyHostToDevice, ctx->stream));
	}
	SSK_HIP(ctx, hipMemcpyAsync(d_off, off.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	SSK_HIP(ctx, hipMemsetAsync(G.res, 0, sizeof(ssk_pg_result), ctx->stream));
	{
		ssk_stage_scope prof(ctx, SSK_STAGE_POSE_GRAPH);
		hipLaunchKernelGGL(pg_solve_kernel, dim3(1), dim3(PG_NT), 0, ctx->stream, G, *params);
		SSK_LAUNCH_CHECK(ctx, "pg_solve_kernel");
	}
	SSK_HIP(ctx, hipMemcpyAsync(out, G.res, sizeof *out, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipMemcpyAsync(node_poses, G.X, 12 * n * 8, hipMemcpyDeviceToHost, ctx->stream));
	SSK_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return SSK_OK;
}
