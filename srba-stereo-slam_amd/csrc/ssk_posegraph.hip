// ssk_posegraph.hip -- the global pose-graph optimisation the reference runs once after its frame loop:
//   rba.get_global_graphslam_problem(poseGraph); mrpt::graphslam::optimize_graph_spa_levmarq(poseGraph, ...)   (EST.cpp:945-957)
// Contract G1-G6: oracle/oracle_posegraph.c.  Levenberg-Marquardt on e = (t_E, log R_E), E = Z^-1 X_i^-1 X_j, unit
// information; the step comes from conjugate gradients preconditioned with the spanning-tree part of the system (block LDL^T on a
// tree has no fill: children before parents, level by level).
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

// G6: S_k = D_k + lambda I - sum over children c of M_e S_c^-1 M_e, Cholesky factors into Lb; levels from the deepest up.
// Returns through *s_ok (cleared when a block is not positive definite).
__device__ void tree_factor(const pg_dev& G, double lambda, int* s_ok)
{
	for (int lvl = G.n_lvl - 1; lvl >= 0; --lvl) {
		for (int idx = G.lvl_off[lvl] + threadIdx.x; idx < G.lvl_off[lvl + 1]; idx += PG_NT) {
			const int k = G.order[idx];
			double Sk[36];
#pragma unroll
			for (int i = 0; i < 36; ++i) Sk[i] = G.D[36 * (size_t)k + i];
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
					chol6_solve(Lc, mc, x);
					mat6_vec(Me, x, yv);
#pragma unroll
					for (int a = 0; a < 6; ++a) Sk[6 * a + col] -= yv[a];
				}
			}
			bool ok = true;
#pragma unroll
			for (int j = 0; j < 6; ++j) {
				double dj = Sk[6 * j + j];
				for (int c = 0; c < j; ++c) dj -= Sk[6 * j + c] * Sk[6 * j + c];
				if (!(dj > 0.0)) { ok = false; dj = 1.0; }
				const double l = sqrt(dj);
				Sk[6 * j + j] = l;
				for (int i = j + 1; i < 6; ++i) {
					double sv = Sk[6 * i + j];
					for (int c = 0; c < j; ++c) sv -= Sk[6 * i + c] * Sk[6 * j + c];
					Sk[6 * i + j] = sv / l;
				}
			}
			if (!ok) *s_ok = 0;
#pragma unroll
			for (int i = 0; i < 36; ++i) G.Lb[36 * (size_t)k + i] = Sk[i];
		}
		__syncthreads();
	}
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
		// ---- G6: tree preconditioner, then CG on (H + lambda I) d = -g over the free nodes ----
		tree_factor(G, lambda, &s_ok);
		for (int k = tid; k < 6 * G.n; k += PG_NT) { G.d[k] = 0; G.z[k] = 0; G.r[k] = (k / 6 == G.fixed) ? 0.0 : -G.g[k]; }
		__syncthreads();
		tree_apply(G, G.r, G.z);
		double rz_l = 0;
		for (int k = tid; k < 6 * G.n; k += PG_NT) { const double zv = G.z[k]; G.p[k] = zv; rz_l += G.r[k] * zv; }
		double rz = pg_block_sum(rz_l, s_part);
		const double rz0 = rz;
		const bool ok = s_ok != 0;
		int cg_it = 0;
		if (ok) {
			while (cg_it < P.pcg_max_iters && rz > P.pcg_tol * P.pcg_tol * rz0 && rz > 0) {
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
	if (params->max_iters < 0 || params->pcg_max_iters < 1 || !(params->tau > 0) || !(params->pcg_tol > 0))
		return ssk_set_error(ctx, SSK_ERR_BAD_ARG, "ssk_posegraph_optimize: max_iters >= 0, pcg_max_iters >= 1, tau > 0, pcg_tol > 0 required");
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
	SSK_HIP(ctx, hipSetDevice(ctx->device));
	const size_t m1 = m ? m : 1;
	const size_t dbl = 12 * n * 2 + 6 * m1 + 36 * m1 + 6 * m1 + 36 * n * 3 + 6 * n * 7 + 12 * m1;
	const size_t bytes = ssk_pad(dbl * 8) + ssk_pad(2 * m1 * 4) + ssk_pad((n + 1) * 4) * 8 + ssk_pad((2 * m + 1) * 4) + ssk_pad(sizeof(ssk_pg_result)) + 32 * 256;
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
