/*
 * oracle_posegraph.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h): the global pose-graph
 * optimisation the reference runs once after the frame loop (EST.cpp:945-987):
 *     rba.get_global_graphslam_problem(poseGraph);  mrpt::graphslam::optimize_graph_spa_levmarq(poseGraph, out_info, NULL, extra_params);
 * Both live in absent libraries (MRPT/srba, MRPT >= 1.3: README.md:14-15).  Restated from the published method (Konolige et
 * al., "Efficient sparse pose adjustment for 2D mapping", IROS 2010, generalised to SE(3) as MRPT does: Levenberg-Marquardt on
 * the relative-pose errors with unit information) with the open choices fixed here ("Pose-graph contract"):
 *   G1  nodes X_k = pose of keyframe k in the frame of the root keyframe, row-major [R|t]; the root is held fixed.  An edge
 *       (i, j, Z) states X_i^-1 X_j = Z.  (The k2k edge {from, to, inv_pose} of the relative map gives i = to, j = from,
 *       Z = inv_pose; initial X_k = composition of the edges along the breadth-first tree from the root.)
 *   G2  error of an edge: E = Z^-1 X_i^-1 X_j = [R_E | t_E];  e = (t_E, log R_E) in R^6 (MRPT's SE_traits<3>::pseudo_ln: the
 *       translation as it is, the rotation as its rotation vector);  cost = sum of e.e over the edges.
 *   G3  increments: X_k <- exp(d_k) (+) X_k with d = (t, w), R <- exp(w) R, t <- exp(w) t + d_t  (the update of the LM contract, L3).
 *   G4  Jacobians to first order: with A = Z^-1 X_i^-1, B = X_j:  de/dd_j = J,  de/dd_i = -J,
 *           J = [ R_A   -R_A [t_B]x ;  0   Jr^-1(log R_E) R_B^T ],   Jr^-1(w) = I + [w]x / 2 + c [w]x^2,
 *           c = 1/th^2 - (1 + cos th) / (2 th sin th)   (1/12 below th = 1e-4).
 *   G5  Levenberg-Marquardt as MRPT's optimize_graph_spa_levmarq runs it (Madsen-Nielsen-Tingleff): H = sum of J^T J blocks,
 *       g = sum of +-J^T e;  (H + lambda I) d = -g;  lambda0 = tau * max diag H;  gain ratio rho = (cost - cost_new) /
 *       (d.(lambda d - g));  a step is kept iff it lowers the cost: lambda <- lambda * max(1/3, 1 - (2 rho - 1)^3), nu <- 2;
 *       otherwise lambda <- lambda * nu, nu <- 2 nu.  Stop: max_iters attempts, lambda > max_lambda, or an attempt changes the cost by
 *       no more than converged_rel_decrease * cost (either sign), or the cost is at round-off level (<= 1e-22 per edge:
 *       residuals of 1e-11 m / rad; a tree-shaped graph starts there).
 *   G6  the linear system is solved by conjugate gradients, from d = 0, until r.z <= pcg_tol^2 * r0.z0 or pcg_max_iters
 *       iterations; sums in node order.  Preconditioner: the system restricted to a spanning tree -- the full 6x6 diagonal
 *       blocks (+ lambda I) with the off-diagonal blocks of the TREE edges only, where the tree is the breadth-first tree
 *       from the fixed node (nodes and their incident edges visited in ascending order; a node the fixed one does not reach
 *       starts its own tree).  On a tree the block LDL^T factorisation has no fill: children before parents,
 *       S_k = D_k + lambda I - sum over children c of M_e S_c^-1 M_e, and two sweeps solve it exactly.  What conjugate
 *       gradients still has to resolve are the loop-closure edges (about six iterations per edge off the tree), not the
 *       length of the trajectory.  (MRPT factorises the whole sparse system; to the tolerance used here the step is the same.)
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void se3_inv(const double* T, double* O)
{
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) O[4 * i + j] = T[4 * j + i];
		O[4 * i + 3] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
	}
}

static void se3_mul(const double* A, const double* B, double* O)
{
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) O[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
		O[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
	}
}

/* rotation vector of R (row-major inside a 3x4) */
void ssko_so3_log(const double* T, double* w)
{
	const double tr = T[0] + T[5] + T[10];
	double c = 0.5 * (tr - 1.0);
	c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
	const double th = acos(c);
	const double v[3] = {T[9] - T[6], T[2] - T[8], T[4] - T[1]};      /* R - R^T = 2 sin(th) [axis]x */
	const double k = th < 1e-6 ? 0.5 + th * th / 12.0 : th / (2.0 * sin(th));
	for (int a = 0; a < 3; ++a) w[a] = k * v[a];
}

/* G2 + G4: error (6) and J (6x6 row-major) of one edge */
static void edge_lin(const double* Xi, const double* Xj, const double* Z, double* e, double* J)
{
	double Zi[12], Xii[12], A[12], E[12];
	se3_inv(Z, Zi); se3_inv(Xi, Xii);
	se3_mul(Zi, Xii, A);
	se3_mul(A, Xj, E);
	e[0] = E[3]; e[1] = E[7]; e[2] = E[11];
	ssko_so3_log(E, e + 3);
	if (!J) return;
	const double* w = e + 3;
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
	const double c = th < 1e-4 ? 1.0 / 12.0 : 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
	const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double Jr[9];
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b) {
			double w2 = 0;
			for (int k = 0; k < 3; ++k) w2 += W[3 * a + k] * W[3 * k + b];
			Jr[3 * a + b] = (a == b ? 1.0 : 0.0) + 0.5 * W[3 * a + b] + c * w2;
		}
	const double tB[3] = {Xj[3], Xj[7], Xj[11]};
	const double TB[9] = {0, -tB[2], tB[1], tB[2], 0, -tB[0], -tB[1], tB[0], 0};
	memset(J, 0, 36 * sizeof(double));
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b) {
			J[6 * a + b] = A[4 * a + b];                                   /* R_A */
			double s = 0, r = 0;
			for (int k = 0; k < 3; ++k) { s += A[4 * a + k] * TB[3 * k + b]; r += Jr[3 * a + k] * Xj[4 * b + k]; }
			J[6 * a + 3 + b] = -s;                                         /* -R_A [t_B]x */
			J[6 * (a + 3) + 3 + b] = r;                                    /* Jr^-1 R_B^T */
		}
}

double ssko_pg_cost(int n_edges, const int32_t* edge_ij, const double* edge_z, const double* X)
{
	double cost = 0, e[6];
	for (int k = 0; k < n_edges; ++k) {
		edge_lin(X + 12 * (size_t)edge_ij[2 * k], X + 12 * (size_t)edge_ij[2 * k + 1], edge_z + 12 * (size_t)k, e, NULL);
		for (int a = 0; a < 6; ++a) cost += e[a] * e[a];
	}
	return cost;
}

static int chol6(const double* Ain, double* L)
{
	memcpy(L, Ain, 36 * sizeof(double));
	for (int j = 0; j < 6; ++j) {
		double d = L[6 * j + j];
		for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
		if (!(d > 0.0)) return 0;
		const double l = sqrt(d);
		L[6 * j + j] = l;
		for (int i = j + 1; i < 6; ++i) {
			double s = L[6 * i + j];
			for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
			L[6 * i + j] = s / l;
		}
	}
	return 1;
}

static void chol6_solve(const double* L, const double* b, double* x)
{
	double y[6];
	for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
	for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
}

/* G6 tree preconditioner.  order: free nodes, parents before children; par[k] / par_e[k]: tree parent (-1: none or the fixed
 * node) and the edge to it.  factor: S-factors (Cholesky, 36 per node) into Ls; returns 0 when a block is not positive definite. */
typedef struct pg_tree { int nf; int32_t* order; int32_t* par; int32_t* par_e; } pg_tree;

static void tree_build(pg_tree* T, int n, int m, const int32_t* edge_ij, int fixed_node)
{
	T->order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
	T->par = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
	T->par_e = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
	int32_t* off = (int32_t*)calloc((size_t)n + 2, sizeof(int32_t));
	int32_t* inc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * m + 1));
	for (int k = 0; k < m; ++k) { off[edge_ij[2 * k] + 2]++; off[edge_ij[2 * k + 1] + 2]++; }
	for (int k = 0; k < n; ++k) off[k + 2] += off[k + 1];
	for (int k = 0; k < m; ++k) { inc[off[edge_ij[2 * k] + 1]++] = k; inc[off[edge_ij[2 * k + 1] + 1]++] = k; }   /* ascending edge id per node */
	unsigned char* seen = (unsigned char*)calloc((size_t)n + 1, 1);
	int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
	T->nf = 0;
	for (int k = 0; k < n; ++k) { T->par[k] = -1; T->par_e[k] = -1; }
	for (int start = -1; start < n; ++start) {              /* the fixed node first, then every node it did not reach */
		const int root = start < 0 ? fixed_node : start;
		if (root < 0 || root >= n || seen[root]) continue;
		int qh = 0, qt = 0;
		queue[qt++] = root; seen[root] = 1;
		if (root != fixed_node) T->order[T->nf++] = root;
		while (qh < qt) {
			const int u = queue[qh++];
			for (int t = off[u]; t < off[u + 1]; ++t) {
				const int ed = inc[t], v = edge_ij[2 * ed] == u ? edge_ij[2 * ed + 1] : edge_ij[2 * ed];
				if (seen[v]) continue;
				seen[v] = 1; queue[qt++] = v;
				T->order[T->nf++] = v;
				if (u != fixed_node) { T->par[v] = u; T->par_e[v] = ed; }
			}
		}
	}
	free(off); free(inc); free(seen); free(queue);
}

static void tree_free(pg_tree* T) { free(T->order); free(T->par); free(T->par_e); }

static void mat6_mul_vec(const double* A, const double* x, double* y)
{
	for (int a = 0; a < 6; ++a) { double s = 0; for (int b = 0; b < 6; ++b) s += A[6 * a + b] * x[b]; y[a] = s; }
}

static int tree_factor(const pg_tree* T, const double* D, const double* M, double lambda, double* S, double* Ls)
{
	for (int t = 0; t < T->nf; ++t) {
		const int k = T->order[t];
		memcpy(S + 36 * (size_t)k, D + 36 * (size_t)k, 36 * sizeof(double));
		for (int a = 0; a < 6; ++a) S[36 * (size_t)k + 7 * a] += lambda;
	}
	for (int t = T->nf - 1; t >= 0; --t) {                  /* children before parents */
		const int c = T->order[t];
		if (!chol6(S + 36 * (size_t)c, Ls + 36 * (size_t)c)) return 0;
		const int p = T->par[c];
		if (p < 0) continue;
		const double* Me = M + 36 * (size_t)T->par_e[c];
		for (int col = 0; col < 6; ++col) {                  /* S_p -= M_e S_c^-1 M_e, column by column */
			double m_col[6], x[6], y[6];
			for (int a = 0; a < 6; ++a) m_col[a] = Me[6 * a + col];
			chol6_solve(Ls + 36 * (size_t)c, m_col, x);
			mat6_mul_vec(Me, x, y);
			for (int a = 0; a < 6; ++a) S[36 * (size_t)p + 6 * a + col] -= y[a];
		}
	}
	return 1;
}

/* z = M^-1 r on the free nodes (z of the fixed node stays 0); y: work vector */
static void tree_apply(const pg_tree* T, const double* M, const double* Ls, const double* r, double* y, double* z)
{
	for (int t = 0; t < T->nf; ++t) { const int k = T->order[t]; memcpy(y + 6 * (size_t)k, r + 6 * (size_t)k, 6 * sizeof(double)); }
	for (int t = T->nf - 1; t >= 0; --t) {                  /* forward: t_c = S_c^-1 y_c, y_p += M_e t_c */
		const int c = T->order[t];
		double tc[6], u[6];
		chol6_solve(Ls + 36 * (size_t)c, y + 6 * (size_t)c, tc);
		memcpy(z + 6 * (size_t)c, tc, sizeof tc);
		const int p = T->par[c];
		if (p < 0) continue;
		mat6_mul_vec(M + 36 * (size_t)T->par_e[c], tc, u);
		for (int a = 0; a < 6; ++a) y[6 * (size_t)p + a] += u[a];
	}
	for (int t = 0; t < T->nf; ++t) {                       /* backward: z_k = t_k + S_k^-1 M_e z_p */
		const int k = T->order[t], p = T->par[k];
		if (p < 0) continue;
		double u[6], v[6];
		mat6_mul_vec(M + 36 * (size_t)T->par_e[k], z + 6 * (size_t)p, u);
		chol6_solve(Ls + 36 * (size_t)k, u, v);
		for (int a = 0; a < 6; ++a) z[6 * (size_t)k + a] += v[a];
	}
}

int ssko_pg_optimize(int n_nodes, double* X, int n_edges, const int32_t* edge_ij, const double* edge_z, int fixed_node,
                     const ssk_pg_params* P, ssk_pg_result* out)
{
	memset(out, 0, sizeof *out);
	const size_t n = (size_t)n_nodes, m = (size_t)n_edges;
	double* e = (double*)malloc(sizeof(double) * 6 * (m + 1));
	double* M = (double*)malloc(sizeof(double) * 36 * (m + 1));     /* J^T J per edge */
	double* D = (double*)malloc(sizeof(double) * 36 * (n + 1));     /* diagonal blocks */
	double* Lb = (double*)malloc(sizeof(double) * 36 * (n + 1));
	double* g = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* d = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* r = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* z = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* p = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* q = (double*)malloc(sizeof(double) * 6 * (n + 1));
	double* Xn = (double*)malloc(sizeof(double) * 12 * (n + 1));
	double* S = (double*)malloc(sizeof(double) * 36 * (n + 1));
	double* yv = (double*)malloc(sizeof(double) * 6 * (n + 1));
	pg_tree T;
	tree_build(&T, n_nodes, n_edges, edge_ij, fixed_node);
	double cost = ssko_pg_cost(n_edges, edge_ij, edge_z, X);
	out->cost_init = cost;
	double lambda = -1.0, nu = 2.0;
	int stop = 0, iters = 0, accepted = 0;
	if (n_nodes <= 1 || n_edges == 0) stop = 3;
	while (!stop) {
		if (iters >= P->max_iters) { stop = 0; break; }
		if (!(cost > 1e-22 * n_edges)) { stop = 1; break; }
		/* linearise (G2, G4) */
		memset(D, 0, sizeof(double) * 36 * n); memset(g, 0, sizeof(double) * 6 * n);
		for (size_t k = 0; k < m; ++k) {
			const int i = edge_ij[2 * k], j = edge_ij[2 * k + 1];
			double J[36];
			edge_lin(X + 12 * (size_t)i, X + 12 * (size_t)j, edge_z + 12 * k, e + 6 * k, J);
			double* Mk = M + 36 * k;
			for (int a = 0; a < 6; ++a) {
				double ga = 0;
				for (int c = 0; c < 6; ++c) ga += J[6 * c + a] * e[6 * k + c];
				g[6 * (size_t)j + a] += ga; g[6 * (size_t)i + a] -= ga;
				for (int b = 0; b < 6; ++b) {
					double s = 0;
					for (int c = 0; c < 6; ++c) s += J[6 * c + a] * J[6 * c + b];
					Mk[6 * a + b] = s;
					D[36 * (size_t)i + 6 * a + b] += s; D[36 * (size_t)j + 6 * a + b] += s;
				}
			}
		}
		if (lambda < 0) {
			double mx = 0;
			for (size_t k = 0; k < n; ++k) if ((int)k != fixed_node) for (int a = 0; a < 6; ++a) mx = fmax(mx, D[36 * k + 7 * a]);
			lambda = P->tau * mx;
		}
		/* G6: tree-preconditioned CG on (H + lambda I) d = -g, fixed node excluded */
		int ok = tree_factor(&T, D, M, lambda, S, Lb);
		double dn2 = 0;
		int cg_it = 0;
		if (ok) {
			double rz = 0;
			for (size_t k = 0; k < n; ++k) {
				for (int a = 0; a < 6; ++a) { d[6 * k + a] = 0; z[6 * k + a] = 0; r[6 * k + a] = (int)k == fixed_node ? 0.0 : -g[6 * k + a]; }
			}
			tree_apply(&T, M, Lb, r, yv, z);
			for (size_t k = 0; k < n; ++k)
				for (int a = 0; a < 6; ++a) { p[6 * k + a] = z[6 * k + a]; rz += r[6 * k + a] * z[6 * k + a]; }
			const double rz0 = rz;
			while (cg_it < P->pcg_max_iters && rz > P->pcg_tol * P->pcg_tol * rz0 && rz > 0) {
				/* q = (H + lambda I) p */
				for (size_t k = 0; k < n; ++k)
					for (int a = 0; a < 6; ++a) {
						double s = lambda * p[6 * k + a];
						for (int b = 0; b < 6; ++b) s += D[36 * k + 6 * a + b] * p[6 * k + b];
						q[6 * k + a] = s;
					}
				for (size_t k = 0; k < m; ++k) {
					const size_t i = (size_t)edge_ij[2 * k], j = (size_t)edge_ij[2 * k + 1];
					const double* Mk = M + 36 * k;
					for (int a = 0; a < 6; ++a) {
						double si = 0, sj = 0;
						for (int b = 0; b < 6; ++b) { si += Mk[6 * a + b] * p[6 * j + b]; sj += Mk[6 * a + b] * p[6 * i + b]; }
						q[6 * i + a] -= si; q[6 * j + a] -= sj;
					}
				}
				if (fixed_node >= 0) for (int a = 0; a < 6; ++a) q[6 * (size_t)fixed_node + a] = 0;
				double pq = 0;
				for (size_t k = 0; k < 6 * n; ++k) pq += p[k] * q[k];
				if (!(pq > 0)) break;
				const double alpha = rz / pq;
				double rz_new = 0;
				for (size_t k = 0; k < n; ++k)
					for (int a = 0; a < 6; ++a) { d[6 * k + a] += alpha * p[6 * k + a]; r[6 * k + a] -= alpha * q[6 * k + a]; }
				tree_apply(&T, M, Lb, r, yv, z);
				for (size_t k = 0; k < n; ++k)
					for (int a = 0; a < 6; ++a) rz_new += r[6 * k + a] * z[6 * k + a];
				const double beta = rz_new / rz;
				for (size_t k = 0; k < 6 * n; ++k) p[k] = z[k] + beta * p[k];
				rz = rz_new;
				++cg_it;
			}
			for (size_t k = 0; k < n; ++k) {
				ssko_se3_left_update(d + 6 * k, X + 12 * k, Xn + 12 * k);
				for (int a = 0; a < 6; ++a) dn2 += d[6 * k + a] * d[6 * k + a];
			}
		}
		out->pcg_iters_total += cg_it;
		const double cost_new = ok ? ssko_pg_cost(n_edges, edge_ij, edge_z, Xn) : cost;
		double pred = 0;                                     /* d.(lambda d - g): the decrease the linear model predicts */
		if (ok) for (size_t k = 0; k < 6 * n; ++k) pred += d[k] * (lambda * d[k] - g[k]);
		const int acc = ok && cost_new < cost;
		const int converged = ok && P->converged_rel_decrease > 0 && fabs(cost - cost_new) <= P->converged_rel_decrease * cost;
		++iters;
		if (acc) {
			memcpy(X, Xn, sizeof(double) * 12 * n);
			const double rho = pred > 0 ? (cost - cost_new) / pred : 1.0, c3 = 2.0 * rho - 1.0;
			const double f = 1.0 - c3 * c3 * c3;
			cost = cost_new; lambda *= f > 1.0 / 3.0 ? f : 1.0 / 3.0; nu = 2.0; ++accepted;
			if (converged) stop = 4;
		} else if (converged) {
			stop = 4;
		} else {
			lambda *= nu; nu *= 2.0;
			if (lambda > P->max_lambda) stop = 2;
		}
		(void)dn2;
	}
	out->iters = iters; out->accepted = accepted; out->stop_reason = stop; out->cost_final = cost; out->lambda_final = lambda;
	free(e); free(M); free(D); free(Lb); free(g); free(d); free(r); free(z); free(p); free(q); free(Xn); free(S); free(yv);
	tree_free(&T);
	return 0;
}
