/*
 * oracle_lm.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h):
 * relative bundle adjustment by Levenberg-Marquardt with Schur complement and dense Cholesky.
 *
 * In the reference this is SRBA (MRPT/srba, header-only, un-vendored, unpinned: README.md:15) instantiated
 * at SRBA.h:30-54 and entered through rba.define_new_keyframe (EST.cpp:166,782).  SRBA's sources are not in
 * the image; this file restates the PUBLISHED algorithm (Blanco, Gonzalez-Jimenez, Fernandez-Madrigal,
 * "Sparser Relative Bundle Adjustment (SRBA): constant-time maintenance and local optimization of arbitrarily
 * large maps", ICRA 2013; MRPT/srba programming guide) and fixes the open choices ("LM contract"):
 *   L1  unknowns: inv_pose of each optimised k2k edge (SE(3)), relative position of each optimised landmark.
 *   L2  prediction of an observation: p = (+)-chain of the path's edge poses (observer side first) applied to
 *       the landmark's relative position, then the inverse sensor pose, then the stereo pinhole model
 *       h = (cx+fx*x/z, cy+fy*y/z, cxr+fxr*(x-b)/z, cyr+fyr*y/z); residual r = z - h.
 *   L3  edge increment: inv_pose <- exp(eps) (+) inv_pose, eps = (t, w): translation taken as is, rotation by
 *       Rodrigues (the "pseudo-exponential" of SE(3)); landmark increment additive.
 *   L4  robust kernel pseudo-Huber: rho(s) = 2k^2(sqrt(1+s^2/k^2)-1) on the residual norm s, total error =
 *       sum rho; iteratively re-weighted: residual and Jacobian rows of an observation are scaled by
 *       sqrt(omega), omega = rho'(s)/(2s) = 1/sqrt(1+s^2/k^2), so J^T r = -1/2 grad(sum rho) exactly.
 *       Observation noise: identical sigma for all (EST.cpp:1157) => a uniform scale that cancels.
 *   L5  damping: lambda0 = tau*max diag(H); each attempt solves (H+lambda I) d = g through the Schur
 *       complement on the landmarks and a dense Cholesky (LL^T) of the reduced pose system;
 *       a non positive pivot or a larger error rejects: lambda *= nu, nu *= 2; success: lambda *= 0.1, nu = 2,
 *       relinearise.  Every attempt counts as an iteration.  Stops: max_iters, rmse threshold, max_lambda,
 *       and "converged": an attempt that changes the error by <= converged_rel_decrease * err (either sign;
 *       the step is kept iff it lowered the error) -- without it the last iterations only chase rounding.
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- SE(3) helpers on row-major 3x4 [R|t] ---- */
static void se3_apply(const double* T, const double* p, double* q)
{
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * p[0] + T[4 * i + 1] * p[1] + T[4 * i + 2] * p[2] + T[4 * i + 3];
}
static void se3_apply_inv(const double* T, const double* p, double* q)
{
	const double d0 = p[0] - T[3], d1 = p[1] - T[7], d2 = p[2] - T[11];
	for (int i = 0; i < 3; ++i) q[i] = T[i] * d0 + T[4 + i] * d1 + T[8 + i] * d2;
}
static void rot_mul(const double* A, int sa, const double* B, int sb, int transB, double* C)   /* C(3x3) = A*B or A*B^T */
{
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < 3; ++k) s += A[sa * i + k] * (transB ? B[sb * j + k] : B[sb * k + j]);
			C[3 * i + j] = s;
		}
}
static void so3_exp(const double* w, double* R)
{
	const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
	const double th = sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
	else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
	const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	double K2[9];
	rot_mul(K, 3, K, 3, 0, K2);
	for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
	R[0] += 1; R[4] += 1; R[8] += 1;
}
/* T <- exp(eps) (+) T  (L3) */
void ssko_se3_left_update(const double* eps, const double* T, double* out)
{
	double R[9];
	so3_exp(eps + 3, R);
	for (int i = 0; i < 3; ++i) {
		for (int j = 0; j < 3; ++j) out[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
		out[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + eps[i];
	}
}

static double huber_rho(double s, double k) { return 2.0 * k * k * (sqrt(1.0 + (s * s) / (k * k)) - 1.0); }

/* prediction + residual of one observation; returns 0 if the point is not in front of the camera plane
 * (z <= 0 still evaluates: the model has no visibility test, as SRBA's). Also returns intermediates. */
typedef struct obs_lin {
	double r[4];          /* weighted residual */
	double w;             /* robust weight */
	double Jf[12];        /* weighted 4x3 */
	double Je[8][24];     /* weighted 4x6 per path position (only unknown edges valid) */
	int n_path;
} obs_lin;

static void project(const ssk_lm_params* P, const double* pc, double* h)
{
	const double iz = 1.0 / pc[2];
	h[0] = P->cx_l + P->fx_l * pc[0] * iz;
	h[1] = P->cy_l + P->fy_l * pc[1] * iz;
	h[2] = P->cx_r + P->fx_r * (pc[0] - P->baseline) * iz;
	h[3] = P->cy_r + P->fy_r * pc[1] * iz;
}

/* residual only: returns robustified squared error contribution */
static double obs_error(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms, int o)
{
	double q[3], t[3];
	memcpy(q, lms + 3 * W->obs_lm[o], sizeof q);
	for (int i = W->obs_path_off[o + 1] - 1; i >= W->obs_path_off[o]; --i) {
		const int e = W->path_edge[i] >> 1, dir = W->path_edge[i] & 1;
		if (dir == 0) se3_apply(poses + 12 * e, q, t); else se3_apply_inv(poses + 12 * e, q, t);
		memcpy(q, t, sizeof q);
	}
	double pc[3], h[4];
	se3_apply_inv(P->sensor_pose, q, pc);
	project(P, pc, h);
	double s2 = 0;
	for (int k = 0; k < 4; ++k) { const double r = (double)W->obs_px[4 * o + k] - h[k]; s2 += r * r; }
	if (!P->use_robust_kernel) return s2;
	return huber_rho(sqrt(s2), P->kernel_param);
}

double ssko_lm_total_error(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms)
{
	double e = 0;
	for (int o = 0; o < W->n_obs; ++o) e += obs_error(W, P, poses, lms, o);
	return e;
}

static void linearize_obs(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms, int o, obs_lin* L)
{
	const int p0 = W->obs_path_off[o], n = W->obs_path_off[o + 1] - p0;
	double q[9][3];               /* q[i]: point after applying path steps i..n-1 (q[n] = landmark) */
	memcpy(q[n], lms + 3 * W->obs_lm[o], sizeof q[n]);
	for (int i = n - 1; i >= 0; --i) {
		const int e = W->path_edge[p0 + i] >> 1, dir = W->path_edge[p0 + i] & 1;
		if (dir == 0) se3_apply(poses + 12 * e, q[i + 1], q[i]); else se3_apply_inv(poses + 12 * e, q[i + 1], q[i]);
	}
	double RA[9][9];              /* RA[i] = product of the step rotations 0..i-1 (RA[0] = I) */
	memset(RA[0], 0, sizeof RA[0]); RA[0][0] = RA[0][4] = RA[0][8] = 1;
	for (int i = 0; i < n; ++i) {
		const int e = W->path_edge[p0 + i] >> 1, dir = W->path_edge[p0 + i] & 1;
		rot_mul(RA[i], 3, poses + 12 * e, 4, dir, RA[i + 1]);   /* dir 1: times R^T */
	}
	double pc[3], h[4];
	se3_apply_inv(P->sensor_pose, q[0], pc);
	project(P, pc, h);
	double r[4], s2 = 0;
	for (int k = 0; k < 4; ++k) { r[k] = (double)W->obs_px[4 * o + k] - h[k]; s2 += r[k] * r[k]; }
	double w = 1.0;
	if (P->use_robust_kernel) {
		/* IRLS weight omega = rho'(s)/(2s) = 1/sqrt(1+s^2/k^2); rows are scaled by sqrt(omega) so that
		 * J^T r is exactly -1/2 of the gradient of sum rho */
		w = sqrt(1.0 / sqrt(1.0 + s2 / (P->kernel_param * P->kernel_param)));
	}
	/* dh/dpc (4x3) */
	const double iz = 1.0 / pc[2], iz2 = iz * iz;
	const double D[12] = {P->fx_l * iz, 0, -P->fx_l * pc[0] * iz2,
	                      0, P->fy_l * iz, -P->fy_l * pc[1] * iz2,
	                      P->fx_r * iz, 0, -P->fx_r * (pc[0] - P->baseline) * iz2,
	                      0, P->fy_r * iz, -P->fy_r * pc[1] * iz2};
	/* M = D * R_S^T (4x3) */
	double M[12];
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 3; ++j)
			M[3 * i + j] = D[3 * i] * P->sensor_pose[4 * j] + D[3 * i + 1] * P->sensor_pose[4 * j + 1] + D[3 * i + 2] * P->sensor_pose[4 * j + 2];
	L->w = w; L->n_path = n;
	for (int k = 0; k < 4; ++k) L->r[k] = w * r[k];
	/* Jf = M * RA[n] */
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 3; ++j)
			L->Jf[3 * i + j] = w * (M[3 * i] * RA[n][j] + M[3 * i + 1] * RA[n][3 + j] + M[3 * i + 2] * RA[n][6 + j]);
	for (int i = 0; i < n; ++i) {
		const int e = W->path_edge[p0 + i] >> 1, dir = W->path_edge[p0 + i] & 1;
		if (W->edge_unknown_slot[e] < 0) continue;
		/* A = M * RA[i] (dir 0) or -M * RA[i+1] (dir 1: RA[i+1] = RA[i] R^T) ; G = [I | -[qq]x] */
		const double* RR = dir == 0 ? RA[i] : RA[i + 1];
		const double* qq = dir == 0 ? q[i] : q[i + 1];
		const double sg = dir == 0 ? w : -w;
		double A[12];
		for (int a = 0; a < 4; ++a)
			for (int b = 0; b < 3; ++b)
				A[3 * a + b] = sg * (M[3 * a] * RR[b] + M[3 * a + 1] * RR[3 + b] + M[3 * a + 2] * RR[6 + b]);
		for (int a = 0; a < 4; ++a) {
			double* J = L->Je[i] + 6 * a;
			J[0] = A[3 * a]; J[1] = A[3 * a + 1]; J[2] = A[3 * a + 2];
			/* A * (-[qq]x): -[q]x = [[0,qz,-qy],[-qz,0,qx],[qy,-qx,0]] */
			J[3] = -A[3 * a + 1] * qq[2] + A[3 * a + 2] * qq[1];
			J[4] = A[3 * a] * qq[2] - A[3 * a + 2] * qq[0];
			J[5] = -A[3 * a] * qq[1] + A[3 * a + 1] * qq[0];
		}
	}
}

static int chol_solve(double* A, int n, double* b)   /* in-place LL^T of the lower triangle, then solve; 0 = not PD */
{
	for (int j = 0; j < n; ++j) {
		double d = A[j * n + j];
		for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
		if (!(d > 0)) return 0;
		d = sqrt(d);
		A[j * n + j] = d;
		for (int i = j + 1; i < n; ++i) {
			double s = A[i * n + j];
			for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
			A[i * n + j] = s / d;
		}
	}
	for (int i = 0; i < n; ++i) {
		double s = b[i];
		for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k];
		b[i] = s / A[i * n + i];
	}
	for (int i = n - 1; i >= 0; --i) {
		double s = b[i];
		for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k];
		b[i] = s / A[i * n + i];
	}
	return 1;
}

static void sym3_inv(const double* V, double* I)   /* inverse of a symmetric 3x3 (row-major full) */
{
	const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
	const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
	const double det = a * c00 + b * c01 + c * c02;
	const double id = 1.0 / det;
	I[0] = c00 * id; I[1] = I[3] = c01 * id; I[2] = I[6] = c02 * id;
	I[4] = (a * f - c * c) * id; I[5] = I[7] = (b * c - a * e) * id; I[8] = (a * d - b * b) * id;
}

/* The LM loop of one window.  poses [n_edges_total*12] and lms [n_lms_total*3] are updated in place.
 * trace (may be NULL, capacity max_iters); dp_trace [max_iters*6E], df_trace [max_iters*3L] (may be NULL). */
int ssko_lm_solve(const ssk_lm_window* W, const ssk_lm_params* P, double* poses, double* lms,
                  ssk_lm_result* res, ssk_lm_trace_entry* trace, double* dp_trace, double* df_trace)
{
	int E = 0, Lu = 0;
	for (int e = 0; e < W->n_edges_total; ++e) if (W->edge_unknown_slot[e] >= 0) ++E;
	for (int l = 0; l < W->n_lms_total; ++l) if (W->lm_unknown_slot[l] >= 0) ++Lu;
	const int n = 6 * E;
	int* e_of_slot = (int*)malloc(sizeof(int) * (size_t)(E + 1));
	int* l_of_slot = (int*)malloc(sizeof(int) * (size_t)(Lu + 1));
	for (int e = 0; e < W->n_edges_total; ++e) if (W->edge_unknown_slot[e] >= 0) e_of_slot[W->edge_unknown_slot[e]] = e;
	for (int l = 0; l < W->n_lms_total; ++l) if (W->lm_unknown_slot[l] >= 0) l_of_slot[W->lm_unknown_slot[l]] = l;
	double* Hpp = (double*)calloc((size_t)n * n + 1, sizeof(double));
	double* gp = (double*)calloc((size_t)n + 1, sizeof(double));
	double* Hff = (double*)calloc((size_t)Lu * 9 + 1, sizeof(double));
	double* gf = (double*)calloc((size_t)Lu * 3 + 1, sizeof(double));
	double* Wpf = (double*)calloc((size_t)n * 3 * Lu + 1, sizeof(double));   /* dense (6E) x (3L): oracle sizes only */
	double* Hred = (double*)malloc(sizeof(double) * ((size_t)n * n + 1));
	double* dp = (double*)malloc(sizeof(double) * (size_t)(n + 1));
	double* df = (double*)malloc(sizeof(double) * (size_t)(3 * Lu + 1));
	double* Vinv = (double*)malloc(sizeof(double) * (size_t)(9 * Lu + 1));
	double* nposes = (double*)malloc(sizeof(double) * 12 * (size_t)W->n_edges_total);
	double* nlms = (double*)malloc(sizeof(double) * 3 * (size_t)(W->n_lms_total + 1));
	obs_lin* L = (obs_lin*)malloc(sizeof(obs_lin));

	double err = ssko_lm_total_error(W, P, poses, lms);
	res->err_init = err; res->n_obs = W->n_obs; res->iters = 0; res->accepted = 0; res->stop_reason = 0;
	double lambda = -1, nu = 2;
	int need_lin = 1;
	if (n == 0 && Lu == 0) { res->stop_reason = 3; goto done; }
	while (res->iters < P->max_iters) {
		if (W->n_obs > 0 && sqrt(err / W->n_obs) < P->max_error_per_obs_to_stop) { res->stop_reason = 1; break; }
		if (need_lin) {
			memset(Hpp, 0, sizeof(double) * (size_t)n * n); memset(gp, 0, sizeof(double) * (size_t)n);
			memset(Hff, 0, sizeof(double) * 9 * (size_t)Lu); memset(gf, 0, sizeof(double) * 3 * (size_t)Lu);
			memset(Wpf, 0, sizeof(double) * (size_t)n * 3 * Lu);
			for (int o = 0; o < W->n_obs; ++o) {
				linearize_obs(W, P, poses, lms, o, L);
				const int p0 = W->obs_path_off[o];
				const int ls = W->lm_unknown_slot[W->obs_lm[o]];
				if (ls >= 0) {
					for (int a = 0; a < 3; ++a) {
						for (int b = 0; b < 3; ++b) {
							double s = 0;
							for (int k = 0; k < 4; ++k) s += L->Jf[3 * k + a] * L->Jf[3 * k + b];
							Hff[9 * ls + 3 * a + b] += s;
						}
						double s = 0;
						for (int k = 0; k < 4; ++k) s += L->Jf[3 * k + a] * L->r[k];
						gf[3 * ls + a] += s;
					}
				}
				for (int i = 0; i < L->n_path; ++i) {
					const int ei = W->edge_unknown_slot[W->path_edge[p0 + i] >> 1];
					if (ei < 0) continue;
					for (int a = 0; a < 6; ++a) {
						double s = 0;
						for (int k = 0; k < 4; ++k) s += L->Je[i][6 * k + a] * L->r[k];
						gp[6 * ei + a] += s;
						if (ls >= 0)
							for (int b = 0; b < 3; ++b) {
								double t = 0;
								for (int k = 0; k < 4; ++k) t += L->Je[i][6 * k + a] * L->Jf[3 * k + b];
								Wpf[(size_t)(6 * ei + a) * 3 * Lu + 3 * ls + b] += t;
							}
					}
					for (int j = 0; j < L->n_path; ++j) {
						const int ej = W->edge_unknown_slot[W->path_edge[p0 + j] >> 1];
						if (ej < 0) continue;
						for (int a = 0; a < 6; ++a)
							for (int b = 0; b < 6; ++b) {
								double s = 0;
								for (int k = 0; k < 4; ++k) s += L->Je[i][6 * k + a] * L->Je[j][6 * k + b];
								Hpp[(size_t)(6 * ei + a) * n + 6 * ej + b] += s;
							}
					}
				}
			}
			if (lambda < 0) {
				double mx = 0;
				for (int i = 0; i < n; ++i) if (Hpp[(size_t)i * n + i] > mx) mx = Hpp[(size_t)i * n + i];
				for (int l = 0; l < Lu; ++l) for (int a = 0; a < 3; ++a) if (Hff[9 * l + 4 * a] > mx) mx = Hff[9 * l + 4 * a];
				lambda = P->tau * mx;
			}
			need_lin = 0;
		}
		/* ---- one attempt ---- */
		const int it = res->iters++;
		for (int l = 0; l < Lu; ++l) {
			double V[9];
			memcpy(V, Hff + 9 * l, sizeof V);
			V[0] += lambda; V[4] += lambda; V[8] += lambda;
			sym3_inv(V, Vinv + 9 * l);
		}
		memcpy(Hred, Hpp, sizeof(double) * (size_t)n * n);
		memcpy(dp, gp, sizeof(double) * (size_t)n);
		for (int i = 0; i < n; ++i) Hred[(size_t)i * n + i] += lambda;
		for (int l = 0; l < Lu; ++l) {
			/* T = W_l V^-1  (n x 3), only rows with a nonzero W_l matter */
			for (int i = 0; i < n; ++i) {
				const double* wi = Wpf + (size_t)i * 3 * Lu + 3 * l;
				if (wi[0] == 0 && wi[1] == 0 && wi[2] == 0) continue;
				double t[3];
				for (int b = 0; b < 3; ++b) t[b] = wi[0] * Vinv[9 * l + b] + wi[1] * Vinv[9 * l + 3 + b] + wi[2] * Vinv[9 * l + 6 + b];
				dp[i] -= t[0] * gf[3 * l] + t[1] * gf[3 * l + 1] + t[2] * gf[3 * l + 2];
				for (int j = 0; j < n; ++j) {
					const double* wj = Wpf + (size_t)j * 3 * Lu + 3 * l;
					Hred[(size_t)i * n + j] -= t[0] * wj[0] + t[1] * wj[1] + t[2] * wj[2];
				}
			}
		}
		int ok = n == 0 ? 1 : chol_solve(Hred, n, dp);
		double dpn = 0, dfn = 0, err_new = err;
		int accepted = 0;
		if (ok) {
			for (int l = 0; l < Lu; ++l) {
				double rhs[3] = {gf[3 * l], gf[3 * l + 1], gf[3 * l + 2]};
				for (int i = 0; i < n; ++i) {
					const double* wi = Wpf + (size_t)i * 3 * Lu + 3 * l;
					rhs[0] -= wi[0] * dp[i]; rhs[1] -= wi[1] * dp[i]; rhs[2] -= wi[2] * dp[i];
				}
				for (int a = 0; a < 3; ++a)
					df[3 * l + a] = Vinv[9 * l + 3 * a] * rhs[0] + Vinv[9 * l + 3 * a + 1] * rhs[1] + Vinv[9 * l + 3 * a + 2] * rhs[2];
			}
			memcpy(nposes, poses, sizeof(double) * 12 * (size_t)W->n_edges_total);
			memcpy(nlms, lms, sizeof(double) * 3 * (size_t)W->n_lms_total);
			for (int s = 0; s < E; ++s) ssko_se3_left_update(dp + 6 * s, poses + 12 * e_of_slot[s], nposes + 12 * e_of_slot[s]);
			for (int s = 0; s < Lu; ++s) for (int a = 0; a < 3; ++a) nlms[3 * l_of_slot[s] + a] += df[3 * s + a];
			for (int i = 0; i < n; ++i) dpn += dp[i] * dp[i];
			for (int i = 0; i < 3 * Lu; ++i) dfn += df[i] * df[i];
			err_new = ssko_lm_total_error(W, P, nposes, nlms);
			if (err_new < err) accepted = 1;
		}
		if (trace) {
			trace[it].lambda = lambda; trace[it].err_before = err; trace[it].err_after = err_new;
			trace[it].delta_p_norm = sqrt(dpn); trace[it].delta_f_norm = sqrt(dfn);
			trace[it].accepted = accepted; trace[it].cholesky_ok = ok;
			if (dp_trace && ok) memcpy(dp_trace + (size_t)it * n, dp, sizeof(double) * (size_t)n);
			if (df_trace && ok) memcpy(df_trace + (size_t)it * 3 * Lu, df, sizeof(double) * 3 * (size_t)Lu);
		}
		const int converged = ok && P->converged_rel_decrease > 0 && fabs(err - err_new) <= P->converged_rel_decrease * err;
		if (accepted) {
			memcpy(poses, nposes, sizeof(double) * 12 * (size_t)W->n_edges_total);
			memcpy(lms, nlms, sizeof(double) * 3 * (size_t)W->n_lms_total);
			err = err_new; lambda *= 0.1; nu = 2; need_lin = 1; res->accepted++;
			if (converged) { res->stop_reason = 4; break; }
		} else if (converged) {
			res->stop_reason = 4; break;
		} else {
			lambda *= nu; nu *= 2;
			if (lambda > P->max_lambda) { res->stop_reason = 2; break; }
		}
	}
done:
	res->err_final = err; res->lambda_final = lambda;
	free(e_of_slot); free(l_of_slot); free(Hpp); free(gp); free(Hff); free(gf); free(Wpf); free(Hred); free(dp); free(df);
	free(Vinv); free(nposes); free(nlms); free(L);
	return 0;
}

/* a10/a11 (EST.cpp:142-161,667-704): observation pixels + initial landmark position
 * camera_pose_on_robot (+) projectMatchTo3D(xl,yl,xr). */
void ssko_build_observations(const ssk_keypoint* kl, const ssk_keypoint* kr, const ssk_dmatch* m, int n,
                             const ssk_lm_params* P, float* obs_px, double* init_pos)
{
	ssko_stereo_cam cam = {P->fx_l, P->fy_l, P->cx_l, P->cy_l, P->fx_r, P->fy_r, P->cx_r, P->cy_r, P->baseline};
	for (int i = 0; i < n; ++i) {
		const ssk_keypoint* a = &kl[m[i].queryIdx];
		const ssk_keypoint* b = &kr[m[i].trainIdx];
		obs_px[4 * i] = a->x; obs_px[4 * i + 1] = a->y; obs_px[4 * i + 2] = b->x; obs_px[4 * i + 3] = b->y;
		double p[3];
		ssko_project_match_to_3d((double)a->x, (double)a->y, (double)b->x, &cam, p);
		se3_apply(P->sensor_pose, p, init_pos + 3 * i);
	}
}

/* Gradient g = J^T r of the (weighted) problem at the given state, for finite-difference checks in tests:
 * gp [6E] ordered by unknown slot, gf [3L]. */
void ssko_lm_gradient(const ssk_lm_window* W, const ssk_lm_params* P, const double* poses, const double* lms,
                      double* gp, double* gf)
{
	int E = 0, Lu = 0;
	for (int e = 0; e < W->n_edges_total; ++e) if (W->edge_unknown_slot[e] >= 0) ++E;
	for (int l = 0; l < W->n_lms_total; ++l) if (W->lm_unknown_slot[l] >= 0) ++Lu;
	memset(gp, 0, sizeof(double) * 6 * (size_t)E);
	memset(gf, 0, sizeof(double) * 3 * (size_t)Lu);
	obs_lin* L = (obs_lin*)malloc(sizeof(obs_lin));
	for (int o = 0; o < W->n_obs; ++o) {
		linearize_obs(W, P, poses, lms, o, L);
		const int p0 = W->obs_path_off[o];
		const int ls = W->lm_unknown_slot[W->obs_lm[o]];
		if (ls >= 0)
			for (int a = 0; a < 3; ++a)
				for (int k = 0; k < 4; ++k) gf[3 * ls + a] += L->Jf[3 * k + a] * L->r[k];
		for (int i = 0; i < L->n_path; ++i) {
			const int ei = W->edge_unknown_slot[W->path_edge[p0 + i] >> 1];
			if (ei < 0) continue;
			for (int a = 0; a < 6; ++a)
				for (int k = 0; k < 4; ++k) gp[6 * ei + a] += L->Je[i][6 * k + a] * L->r[k];
		}
	}
	free(L);
}
