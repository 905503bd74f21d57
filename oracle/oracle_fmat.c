/*
 * oracle_fmat.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h): the fundamental-matrix outlier
 * filter of the data association, m_detect_outliers_with_F (EST.cpp:2015-2055).
 *
 * In-tree parts restated literally: which points enter (left keypoints of the tracked pairs, OTHER keyframe first,
 * EST.cpp:2022-2040), the call  cv::findFundamentalMat(p_other_left, p_this_left, FM_RANSAC, max_y_diff_epipolar,
 * ransac_fit_prob, mask)  (:2044) and the verdict (mask 0 => sREJ_FUND_MATRIX, :2047-2054).  findFundamentalMat itself
 * is OpenCV 2.4 (absent).  Its RANSAC draws samples from OpenCV's global RNG, so bit parity with the reference is
 * impossible by construction ("only statistical parity possible"); what is restated is its
 * published procedure (fundam.cpp / modelest.cpp of OpenCV 2.4: 7-point minimal solver inside RANSAC with an adaptive
 * iteration count, symmetric epipolar distance) with every open choice fixed ("Fundamental-matrix contract"):
 *   F1  at least min_points (15: below that OpenCV 2.4 switches to LMedS and the reference never calls, EST.cpp:1590)
 *       pairs, else the filter is invalid and changes nothing.
 *   F2  hypothesis h = 0, 1, ... draws 7 distinct pair indices from its OWN generator (PCG32 XSH-RR seeded with
 *       seed + h * 0x9E3779B97F4A7C15; index = next() % n, a repeated index is redrawn) -- counter based, so hypotheses
 *       are independent of each other and can be evaluated in any order or in parallel.
 *   F3  7-point solver on coordinates normalised by fixed constants (x - cx) * s, (y - cy) * s: rows
 *       [x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1]; null space by Gauss-Jordan on columns 0..6 with row pivoting
 *       (a pivot below 1e-12 discards the sample); F(l) = l N1 + N2; det F(l) = 0 is a cubic whose coefficients come from
 *       det at l = 0, 1, -1 and det N1; its real roots are bracketed by the critical points and found by 80 bisections
 *       each (arithmetic only: reproducible on any IEEE machine); every root gives a model, ascending root order;
 *       models are denormalised and scaled to unit Frobenius norm.
 *   F4  a pair is an inlier of F iff both squared point-line distances d(m2, F m1)^2, d(m1, F^T m2)^2 are <= threshold^2
 *       (OpenCV: max of the two; tested here with the line norms multiplied over, no division).
 *   F5  sequential selection as OpenCV's loop: niters = max_iters; for h < niters, for each model of h: if its inlier
 *       count exceeds max(best, 6) it becomes the best and niters = the smallest k with (1 - w^7)^k <= 1 - confidence,
 *       w = inliers / n, capped by the previous niters (cvRANSACUpdateNumIters up to its rounding, computed without libm).
 *   F6  verdicts from the best model's inlier set; no model at all => invalid, nothing changes.
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* PCG32 XSH-RR, one stream per hypothesis */
static uint32_t pcg_next(uint64_t* st)
{
	const uint64_t old = *st;
	*st = old * 6364136223846793005ULL + 1442695040888963407ULL;
	const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	const uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

void ssko_fm_sample(uint64_t seed, int h, int n, int idx[7])
{
	uint64_t st = seed + (uint64_t)h * 0x9E3779B97F4A7C15ULL;
	(void)pcg_next(&st);
	for (int k = 0; k < 7; ++k) {
		for (;;) {
			const int v = (int)(pcg_next(&st) % (uint32_t)n);
			int dup = 0;
			for (int j = 0; j < k; ++j) dup |= idx[j] == v;
			if (!dup) { idx[k] = v; break; }
		}
	}
}

static double det3(const double* M)
{
	return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static double cubic_eval(double a, double b, double c, double x) { return ((x + a) * x + b) * x + c; }

/* root of the monic cubic in [lo, hi] given a sign change: 80 bisections */
static double bisect(double a, double b, double c, double lo, double hi)
{
	double flo = cubic_eval(a, b, c, lo);
	for (int it = 0; it < 80; ++it) {
		const double mid = 0.5 * (lo + hi);
		const double fm = cubic_eval(a, b, c, mid);
		if ((flo <= 0.0) == (fm <= 0.0)) { lo = mid; flo = fm; } else hi = mid;
	}
	return 0.5 * (lo + hi);
}

/* real roots of c3 x^3 + c2 x^2 + c1 x + c0, ascending; returns their number (0..3) */
int ssko_fm_cubic_roots(double c3, double c2, double c1, double c0, double* roots)
{
	const double m = fmax(fmax(fabs(c3), fabs(c2)), fmax(fabs(c1), fabs(c0)));
	if (!(m > 0.0)) return 0;
	if (fabs(c3) <= 1e-12 * m) {                       /* degenerate to a quadratic / linear equation */
		if (fabs(c2) <= 1e-12 * m) {
			if (fabs(c1) <= 1e-12 * m) return 0;
			roots[0] = -c0 / c1;
			return 1;
		}
		const double disc = c1 * c1 - 4.0 * c2 * c0;
		if (disc < 0.0) return 0;
		const double sq = sqrt(disc);
		const double r0 = (-c1 - sq) / (2.0 * c2), r1 = (-c1 + sq) / (2.0 * c2);
		roots[0] = fmin(r0, r1); roots[1] = fmax(r0, r1);
		return 2;
	}
	const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
	const double R = 1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)));
	const double disc = a * a - 3.0 * b;
	if (!(disc > 0.0)) { roots[0] = bisect(a, b, c, -R, R); return 1; }
	const double sq = sqrt(disc);
	const double x1 = (-a - sq) / 3.0, x2 = (-a + sq) / 3.0;      /* local max, local min */
	const double f1 = cubic_eval(a, b, c, x1), f2 = cubic_eval(a, b, c, x2);
	int n = 0;
	if (f1 >= 0.0) roots[n++] = bisect(a, b, c, -R, x1);           /* f(-R) < 0 */
	if (f1 >= 0.0 && f2 <= 0.0) roots[n++] = bisect(a, b, c, x1, x2) ;
	if (f2 <= 0.0) roots[n++] = bisect(a, b, c, x2, R);            /* f(R) > 0 */
	return n;
}

/* F3: up to 3 fundamental matrices (row-major, pixel coordinates, unit Frobenius norm) from 7 pairs */
int ssko_fm_seven_point(const double* m1, const double* m2, const int idx[7], const ssk_fm_params* P, double F[3][9])
{
	double A[7][9];
	for (int r = 0; r < 7; ++r) {
		const double x1 = (m1[2 * idx[r]] - P->norm_cx) * P->norm_scale, y1 = (m1[2 * idx[r] + 1] - P->norm_cy) * P->norm_scale;
		const double x2 = (m2[2 * idx[r]] - P->norm_cx) * P->norm_scale, y2 = (m2[2 * idx[r] + 1] - P->norm_cy) * P->norm_scale;
		A[r][0] = x2 * x1; A[r][1] = x2 * y1; A[r][2] = x2; A[r][3] = y2 * x1; A[r][4] = y2 * y1; A[r][5] = y2;
		A[r][6] = x1; A[r][7] = y1; A[r][8] = 1.0;
	}
	for (int c = 0; c < 7; ++c) {                      /* Gauss-Jordan, row pivoting */
		int p = c;
		double best = fabs(A[c][c]);
		for (int r = c + 1; r < 7; ++r) if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
		if (!(best >= 1e-12)) return 0;
		if (p != c) for (int j = 0; j < 9; ++j) { const double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
		const double inv = 1.0 / A[c][c];
		for (int j = 0; j < 9; ++j) A[c][j] = A[c][j] * inv;
		for (int r = 0; r < 7; ++r) {
			if (r == c) continue;
			const double f = A[r][c];
			for (int j = 0; j < 9; ++j) A[r][j] = A[r][j] - f * A[c][j];
		}
	}
	double N1[9], N2[9];
	for (int j = 0; j < 7; ++j) { N1[j] = -A[j][7]; N2[j] = -A[j][8]; }
	N1[7] = 1.0; N1[8] = 0.0; N2[7] = 0.0; N2[8] = 1.0;
	double Mp[9], Mm[9];
	for (int i = 0; i < 9; ++i) { Mp[i] = N1[i] + N2[i]; Mm[i] = N2[i] - N1[i]; }
	const double c0 = det3(N2), c3 = det3(N1), p1 = det3(Mp), pm = det3(Mm);
	const double c2 = 0.5 * (p1 + pm) - c0, c1 = 0.5 * (p1 - pm) - c3;
	double roots[3];
	const int nr = ssko_fm_cubic_roots(c3, c2, c1, c0, roots);
	const double s = P->norm_scale, tx = -P->norm_scale * P->norm_cx, ty = -P->norm_scale * P->norm_cy;
	int nm = 0;
	for (int k = 0; k < nr; ++k) {
		double G[9];
		for (int i = 0; i < 9; ++i) G[i] = roots[k] * N1[i] + N2[i];
		/* F = T^T G T with T = [s 0 tx; 0 s ty; 0 0 1]   (both point sets share the normalisation) */
		double GT[9];                                   /* G T */
		for (int r = 0; r < 3; ++r) {
			GT[3 * r] = G[3 * r] * s; GT[3 * r + 1] = G[3 * r + 1] * s;
			GT[3 * r + 2] = G[3 * r] * tx + G[3 * r + 1] * ty + G[3 * r + 2];
		}
		double Fp[9];
		for (int c = 0; c < 3; ++c) {
			Fp[c] = s * GT[c]; Fp[3 + c] = s * GT[3 + c];
			Fp[6 + c] = tx * GT[c] + ty * GT[3 + c] + GT[6 + c];
		}
		double nrm = 0;
		for (int i = 0; i < 9; ++i) nrm += Fp[i] * Fp[i];
		if (!(nrm > 0.0)) continue;
		const double inv = 1.0 / sqrt(nrm);
		for (int i = 0; i < 9; ++i) F[nm][i] = Fp[i] * inv;
		++nm;
	}
	return nm;
}

/* F4: inlier test without divisions: d(m2, F m1)^2 <= t^2 and d(m1, F^T m2)^2 <= t^2 with the line norms multiplied over */
int ssko_fm_inlier(const double* F, const double* p1, const double* p2, double thr2)
{
	double a = F[0] * p1[0] + F[1] * p1[1] + F[2], b = F[3] * p1[0] + F[4] * p1[1] + F[5], c = F[6] * p1[0] + F[7] * p1[1] + F[8];
	const double n2 = a * a + b * b, d2 = p2[0] * a + p2[1] * b + c;
	a = F[0] * p2[0] + F[3] * p2[1] + F[6]; b = F[1] * p2[0] + F[4] * p2[1] + F[7]; c = F[2] * p2[0] + F[5] * p2[1] + F[8];
	const double n1 = a * a + b * b, d1 = p1[0] * a + p1[1] * b + c;
	return d2 * d2 <= thr2 * n2 && d1 * d1 <= thr2 * n1;
}

/* F5: iterations needed so that a sample of 7 inliers has been drawn with probability >= p when the inlier ratio is
 * good / n: the smallest k with (1 - w^7)^k <= 1 - p, by repeated multiplication (no libm: the same on any IEEE machine;
 * OpenCV's cvRANSACUpdateNumIters rounds log(1-p)/log(1-w^7) to nearest instead), capped by max_iters. */
int ssko_fm_update_iters(double p, int good, int n, int max_iters)
{
	const double w = (double)good / (double)n;
	const double w2 = w * w, w4 = w2 * w2, base = 1.0 - w4 * w2 * w, target = 1.0 - p;
	double q = 1.0;
	int k = 0;
	while (q > target && k < max_iters) { q = q * base; ++k; }
	return k;
}

int ssko_fundamental_filter(const float* this_xy, int n_this, const float* other_xy, int n_other, const int32_t* train_idx,
                            int32_t* status, const ssk_fm_params* P, ssk_fm_result* out)
{
	(void)n_other;
	memset(out, 0, sizeof *out);
	out->best_hypothesis = -1;
	double* m1 = (double*)malloc(sizeof(double) * 2 * (size_t)(n_this > 0 ? n_this : 1));
	double* m2 = (double*)malloc(sizeof(double) * 2 * (size_t)(n_this > 0 ? n_this : 1));
	int* idx_of = (int*)malloc(sizeof(int) * (size_t)(n_this > 0 ? n_this : 1));
	int n = 0;
	for (int k = 0; k < n_this; ++k) {                  /* [literal EST.cpp:2022-2040] */
		if (status[k] != SSK_DA_TRACKED) continue;
		m2[2 * n] = (double)this_xy[2 * (size_t)k]; m2[2 * n + 1] = (double)this_xy[2 * (size_t)k + 1];
		m1[2 * n] = (double)other_xy[2 * (size_t)train_idx[k]]; m1[2 * n + 1] = (double)other_xy[2 * (size_t)train_idx[k] + 1];
		idx_of[n++] = k;
	}
	out->n_pairs = n;
	if (n >= P->min_points && n >= 7) {
		const double thr2 = P->threshold * P->threshold;
		int niters = P->max_iters, best = 0;
		double Fbest[9] = {0};
		int h = 0;
		for (; h < niters; ++h) {                       /* F5 */
			int idx[7];
			double F[3][9];
			ssko_fm_sample(P->seed, h, n, idx);
			const int nm = ssko_fm_seven_point(m1, m2, idx, P, F);
			for (int k = 0; k < nm; ++k) {
				int good = 0;
				for (int i = 0; i < n; ++i) good += ssko_fm_inlier(F[k], m1 + 2 * i, m2 + 2 * i, thr2);
				if (good > (best > 6 ? best : 6)) {
					best = good; memcpy(Fbest, F[k], sizeof Fbest); out->best_hypothesis = h;
					niters = ssko_fm_update_iters(P->confidence, good, n, niters);
				}
			}
		}
		out->hypotheses = h;
		if (best > 0) {
			out->valid = 1; out->n_inliers = best;
			memcpy(out->F, Fbest, sizeof Fbest);
			for (int i = 0; i < n; ++i)
				if (!ssko_fm_inlier(Fbest, m1 + 2 * i, m2 + 2 * i, thr2)) { status[idx_of[i]] = SSK_DA_REJ_FUND_MATRIX; out->n_rejected++; }   /* :2047-2054 */
		}
	}
	free(m1); free(m2); free(idx_of);
	return out->valid;
}
