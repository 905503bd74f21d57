/*
 * oracle_cip.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h): the change-in-pose
 * outlier filter of the data association, m_detect_outliers_with_change_in_pose (EST.cpp:2113-2177).
 *
 * In-tree parts restated literally: which pairs enter (status == sTRACKED, pair = (trainIdx, k),
 * EST.cpp:2120-2133), what happens when the estimate is invalid (every tracked match becomes
 * sREJ_CHANGE_POSE, :2152-2162) and the residual gate (:2164-2176).  The estimator itself is
 * rso::CStereoOdometryEstimator::getChangeInPose of stereo-vo (github famoreno/stereo-vo, unpinned,
 * README.md:16), absent from the image; it is restated from its published description (stereo visual odometry
 * by robust least squares on the stereo reprojection error, two stages with an inlier gate in between;
 * parameter names and values [LEAST_SQUARES] KITTI.ini:40-50) with the open choices fixed here
 * ("Change-in-pose contract"):
 *   P1  3-D point of a pair: projectMatchTo3D (UTIL.h:558-574) of the OTHER keyframe's stereo match, in that
 *       keyframe's left-camera frame; measurement z = (xl, yl, xr, yr) of THIS keyframe's stereo match.
 *   P2  unknown: T = [R|t], the other frame seen from this one; prediction h(R P + t) with the stereo pinhole
 *       model of the LM contract (L2); residual r = z - h, squared norm s2 = r.r (pixel^2) is "the residual"
 *       the gates compare.
 *   P3  Gauss-Newton, left-multiplicative update T <- exp(d) (+) T with d = (t, w) as in L3; pseudo-Huber
 *       kernel by IRLS weights omega = 1/sqrt(1 + s2/k^2) as in L4 (cost = sum 2k^2(sqrt(1+s2/k^2)-1), or
 *       sum s2 without kernel); normal equations H d = g, H = sum omega J^T J, g = sum omega J^T r,
 *       J = dh/dd = D [I | -[q]x], solved by Cholesky LL^T; a non positive pivot makes the estimate invalid.
 *   P4  a stage: up to max iterations; each applies the step, counts, stops when |d| < min_mod_out_vector, and
 *       compares the new cost with the previous one: a larger cost counts, more than max_incr_cost of them stop
 *       the stage.
 *   P5  stage 1 on all tracked pairs (initial_max_iters), then inliers = pairs with s2 <= residual_threshold,
 *       stage 2 on the inliers (max_iters).  Fewer than bad_tracking_th pairs before either stage: invalid.
 *   P6  output residual of EVERY tracked pair = s2 at the final pose; s2 > residual_th => sREJ_CHANGE_POSE.
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct pair_t { double P[3]; double z[4]; } pair_t;

/* residual, squared norm and (optionally) the 4x6 Jacobian J = D [I | -[q]x] of one pair at T */
static double pair_eval(const ssk_cip_params* c, const double* T, const pair_t* p, double* r, double* J)
{
	double q[3];
	for (int i = 0; i < 3; ++i) q[i] = T[4 * i] * p->P[0] + T[4 * i + 1] * p->P[1] + T[4 * i + 2] * p->P[2] + T[4 * i + 3];
	const double iz = 1.0 / q[2], iz2 = iz * iz;
	const double h[4] = {c->cx_l + c->fx_l * q[0] * iz, c->cy_l + c->fy_l * q[1] * iz,
	                     c->cx_r + c->fx_r * (q[0] - c->baseline) * iz, c->cy_r + c->fy_r * q[1] * iz};
	double s2 = 0;
	for (int k = 0; k < 4; ++k) { r[k] = p->z[k] - h[k]; s2 += r[k] * r[k]; }
	if (J) {
		const double D[12] = {c->fx_l * iz, 0, -c->fx_l * q[0] * iz2,
		                      0, c->fy_l * iz, -c->fy_l * q[1] * iz2,
		                      c->fx_r * iz, 0, -c->fx_r * (q[0] - c->baseline) * iz2,
		                      0, c->fy_r * iz, -c->fy_r * q[1] * iz2};
		for (int a = 0; a < 4; ++a) {
			const double* d = D + 3 * a;
			J[6 * a] = d[0]; J[6 * a + 1] = d[1]; J[6 * a + 2] = d[2];
			J[6 * a + 3] = q[1] * d[2] - q[2] * d[1];       /* (q x D_a): the rotation part, as in the LM contract */
			J[6 * a + 4] = q[2] * d[0] - q[0] * d[2];
			J[6 * a + 5] = q[0] * d[1] - q[1] * d[0];
		}
	}
	return s2;
}

static double rho(const ssk_cip_params* c, double s2)
{
	if (!c->use_robust_kernel) return s2;
	const double k2 = c->kernel_param * c->kernel_param;
	return 2.0 * k2 * (sqrt(1.0 + s2 / k2) - 1.0);
}

static double total_cost(const ssk_cip_params* c, const double* T, const pair_t* pr, const unsigned char* act, int n)
{
	double s = 0, r[4];
	for (int i = 0; i < n; ++i) if (act[i]) s += rho(c, pair_eval(c, T, pr + i, r, 0));
	return s;
}

/* Cholesky solve of the 6x6 system; returns 0 on a non positive pivot */
static int solve6(double* H, double* g)
{
	for (int j = 0; j < 6; ++j) {
		double d = H[6 * j + j];
		for (int k = 0; k < j; ++k) d -= H[6 * j + k] * H[6 * j + k];
		if (!(d > 0.0)) return 0;
		const double l = sqrt(d);
		H[6 * j + j] = l;
		for (int i = j + 1; i < 6; ++i) {
			double s = H[6 * i + j];
			for (int k = 0; k < j; ++k) s -= H[6 * i + k] * H[6 * j + k];
			H[6 * i + j] = s / l;
		}
	}
	for (int i = 0; i < 6; ++i) { double s = g[i]; for (int k = 0; k < i; ++k) s -= H[6 * i + k] * g[k]; g[i] = s / H[6 * i + i]; }
	for (int i = 5; i >= 0; --i) { double s = g[i]; for (int k = i + 1; k < 6; ++k) s -= H[6 * k + i] * g[k]; g[i] = s / H[6 * i + i]; }
	return 1;
}

/* one stage (P4); returns iterations done, -1 if the system was not positive definite */
static int run_stage(const ssk_cip_params* c, double* T, const pair_t* pr, const unsigned char* act, int n, int max_it)
{
	double prev = total_cost(c, T, pr, act, n);
	int incr = 0, it = 0;
	while (it < max_it) {
		double H[36], g[6];
		memset(H, 0, sizeof H); memset(g, 0, sizeof g);
		for (int i = 0; i < n; ++i) {
			if (!act[i]) continue;
			double r[4], J[24];
			const double s2 = pair_eval(c, T, pr + i, r, J);
			const double om = c->use_robust_kernel ? 1.0 / sqrt(1.0 + s2 / (c->kernel_param * c->kernel_param)) : 1.0;
			for (int a = 0; a < 6; ++a) {
				double ga = 0;
				for (int k = 0; k < 4; ++k) ga += J[6 * k + a] * r[k];
				g[a] += om * ga;
				for (int b = 0; b <= a; ++b) {
					double hab = 0;
					for (int k = 0; k < 4; ++k) hab += J[6 * k + a] * J[6 * k + b];
					H[6 * a + b] += om * hab;
				}
			}
		}
		for (int a = 0; a < 6; ++a) for (int b = a + 1; b < 6; ++b) H[6 * a + b] = H[6 * b + a];
		if (!solve6(H, g)) return -1;
		double Tn[12];
		ssko_se3_left_update(g, T, Tn);                    /* d = (t, w) */
		memcpy(T, Tn, sizeof Tn);
		++it;
		double nd = 0;
		for (int a = 0; a < 6; ++a) nd += g[a] * g[a];
		if (sqrt(nd) < c->min_mod_out_vector) break;
		const double cost = total_cost(c, T, pr, act, n);
		if (cost > prev && ++incr > c->max_incr_cost) break;
		prev = cost;
	}
	return it;
}

int ssko_change_in_pose(const float* this_obs, int n_this, const float* other_obs, int n_other,
                        const int32_t* train_idx, int32_t* status, const double init_pose[12],
                        const ssk_cip_params* c, double* residuals, ssk_cip_result* out)
{
	(void)n_other;
	memset(out, 0, sizeof *out);
	pair_t* pr = (pair_t*)malloc(sizeof(pair_t) * (size_t)(n_this > 0 ? n_this : 1));
	int* idx = (int*)malloc(sizeof(int) * (size_t)(n_this > 0 ? n_this : 1));
	unsigned char* act = (unsigned char*)calloc((size_t)(n_this > 0 ? n_this : 1), 1);      /* (zeroed: entries beyond the n tracked pairs are never read, gcc cannot see that) */
	const ssko_stereo_cam cam = {c->fx_l, c->fy_l, c->cx_l, c->cy_l, c->fx_r, c->fy_r, c->cx_r, c->cy_r, c->baseline};
	int n = 0;
	for (int k = 0; k < n_this; ++k) {                     /* [literal EST.cpp:2120-2133] */
		if (status[k] != SSK_DA_TRACKED) continue;
		const float* o = other_obs + 4 * (size_t)train_idx[k];
		ssko_project_match_to_3d((double)o[0], (double)o[1], (double)o[2], &cam, pr[n].P);   /* P1 */
		for (int a = 0; a < 4; ++a) pr[n].z[a] = (double)this_obs[4 * (size_t)k + a];
		idx[n] = k; act[n] = 1; ++n;
	}
	out->n_pairs = n;
	double T[12];
	memcpy(T, init_pose, sizeof T);
	int valid = n >= c->bad_tracking_th;
	if (valid) {
		const int it1 = run_stage(c, T, pr, act, n, c->initial_max_iters);
		if (it1 < 0) valid = 0; else out->iters_initial = it1;
	}
	int n_in = 0;
	if (valid) {
		double r[4];
		for (int i = 0; i < n; ++i) { act[i] = pair_eval(c, T, pr + i, r, 0) <= c->residual_threshold; n_in += act[i]; }   /* P5 */
		if (n_in < c->bad_tracking_th) valid = 0;
	}
	if (valid) {
		const int it2 = run_stage(c, T, pr, act, n, c->max_iters);
		if (it2 < 0) valid = 0; else out->iters_final = it2;
	}
	out->valid = valid;
	out->n_inliers = n_in;
	memcpy(out->pose, T, sizeof T);
	if (valid) out->cost = total_cost(c, T, pr, act, n);
	for (int i = 0; i < n; ++i) {
		double r[4];
		const double s2 = valid ? pair_eval(c, T, pr + i, r, 0) : 0.0;
		if (residuals) residuals[i] = s2;
		if (!valid || s2 > c->residual_th) { status[idx[i]] = SSK_DA_REJ_CHANGE_POSE; out->n_rejected++; }   /* [literal :2152-2176] */
	}
	free(pr); free(idx); free(act);
	return valid;
}
