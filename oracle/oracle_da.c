/*
 * oracle_da.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h):
 * keyframe<->keyframe data association of the reference, EST.cpp = src/CSRBAStereoSLAMEstimator.cpp.
 */
#include "ssk_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* 256-bit Hamming distance of two 32-byte rows (rows need not be aligned). */
static int hamming256(const uint8_t* a, const uint8_t* b)
{
	int d = 0;
	for (int w = 0; w < 4; ++w) {
		uint64_t x, y;
		memcpy(&x, a + 8 * w, 8);
		memcpy(&y, b + 8 * w, 8);
		d += __builtin_popcountll(x ^ y);
	}
	return d;
}

/* OpenCV 2.4 BFMatcher::knnMatchImpl with K=1, crossCheck=false -> batchDistance(..., K=0 path):
 * per query row the minimum distance and its first index (strict '<' while scanning train rows in
 * order); DMatch(queryIdx, trainIdx, imgIdx=0, (float)dist); empty query or train -> no matches.
 * Reference call: EST.cpp:1447-1452. */
int ssko_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt, ssk_dmatch* out)
{
	if (nq <= 0 || nt <= 0) return 0;
	for (int q = 0; q < nq; ++q) {
		int best = INT_MAX, bi = -1;
		for (int t = 0; t < nt; ++t) {
			const int d = hamming256(query + 32 * q, train + 32 * t);
			if (d < best) { best = d; bi = t; }
		}
		out[q].queryIdx = q;
		out[q].trainIdx = bi;
		out[q].imgIdx = 0;
		out[q].distance = (float)best;
	}
	return nq;
}

/* EST.cpp:1883-1946.  Line-by-line:
 *   :1909-1913  tx,ty,ox float; oy = other.pt.y + offset  (float + size_t -> float)
 *   :1915       double ang = 180/M_PI*atanf((oy-ty)/(ox-tx))+180.   (float quotient, float atanf, double rest)
 *   :1916       bin = floor(ang/BIN_WIDTH)  (BIN_WIDTH int 10)
 *   :1923-1931  mode = first bin with hist > best_so_far (strict)
 *   :1934-1936  keep mode-1, mode, mode+1 with wrap over 36 bins
 *   :1940-1945  everything histogrammed elsewhere -> sREJ_SLOPE
 * Only matches whose status is sTRACKED take part (:1903).  A NaN quotient (both keypoints on the same
 * pixel AND offset==0) indexes hist[] out of range in the reference (undefined behaviour); the oracle
 * defines that case as sREJ_SLOPE without histogramming it. */
void ssko_da_filter_direction(const ssk_dmatch* matches, int n_this, const float* this_xy,
                              const float* other_xy, int offset, int32_t* status)
{
	enum { N_BINS = 36, BIN_WIDTH = 10 };
	int hist[N_BINS];
	memset(hist, 0, sizeof hist);
	int* bin_of = (int*)malloc(sizeof(int) * (size_t)(n_this > 0 ? n_this : 1));
	for (int k = 0; k < n_this; ++k) {
		bin_of[k] = -1;
		if (status[k] != SSK_DA_TRACKED) continue;
		const float tx = this_xy[2 * k], ty = this_xy[2 * k + 1];
		const int j = matches[k].trainIdx;
		const float ox = other_xy[2 * j];
		const float oy = other_xy[2 * j + 1] + (float)(size_t)offset;
		const float r = (oy - ty) / (ox - tx);
		const double ang = 180 / M_PI * atanf(r) + 180.;
		if (ang != ang) { bin_of[k] = -2; continue; }
		const int bin = (int)floor(ang / BIN_WIDTH);
		bin_of[k] = bin;
		hist[bin]++;
	}
	int mode = 0, best_so_far = 0;
	for (int k = 0; k < N_BINS; ++k)
		if (hist[k] > best_so_far) { best_so_far = hist[k]; mode = k; }
	const int ibin_0 = mode == 0 ? N_BINS - 1 : mode - 1;
	const int ibin_1 = mode;
	const int ibin_2 = mode == N_BINS - 1 ? 0 : mode + 1;
	for (int k = 0; k < n_this; ++k) {
		const int b = bin_of[k];
		if (b == -1) continue;
		if (b == -2 || !(b == ibin_0 || b == ibin_1 || b == ibin_2)) status[k] = SSK_DA_REJ_SLOPE;
	}
	free(bin_of);
}

/* EST.cpp:1951-2010, including quirk Q1 (:2002-2007): with the two middle lines commented out the
 * statement reads   other_matched[trainIdx].second = ++stage1_counter;   so .second holds a running
 * counter (converted to the pair's float), initially FLT_MAX (:1961). */
void ssko_da_filter_orb_distance(const ssk_dmatch* matches, int n_this, int n_other,
                                 double max_orb_distance_da, int32_t* status)
{
	int* om_first = (int*)malloc(sizeof(int) * (size_t)(n_other > 0 ? n_other : 1));
	float* om_second = (float*)malloc(sizeof(float) * (size_t)(n_other > 0 ? n_other : 1));
	for (int j = 0; j < n_other; ++j) { om_first[j] = -1; om_second[j] = FLT_MAX; }
	size_t stage1_counter = 0;
	for (int k = 0; k < n_this; ++k) {
		const ssk_dmatch* it = &matches[k];
		const int q = it->queryIdx, t = it->trainIdx;
		if (status[q] != SSK_DA_TRACKED) continue;                                  /* :1969 */
		if ((double)it->distance > max_orb_distance_da) { status[q] = SSK_DA_REJ_ORB; continue; }  /* :1980 */
		if (it->distance > om_second[t]) { status[q] = SSK_DA_NON_TRACKED; continue; }             /* :1988 */
		if (om_first[t] != -1) { status[om_first[t]] = SSK_DA_NON_TRACKED; --stage1_counter; }     /* :1995-1999 */
		om_first[t] = q;                                                                           /* :2002 */
		om_second[t] = (float)(++stage1_counter);                                                  /* :2003+2007 */
	}
	free(om_first);
	free(om_second);
}

/* EST.cpp:1413-1515 (+ :1647-1669).  Q3: descriptors and coordinates are re-indexed by MATCH. */
int ssko_data_association(const ssk_kf_view* a, const ssk_kf_view* b, const ssk_da_params* p,
                          ssk_dmatch* out_matches, int32_t* out_status)
{
	const int na = a->n_matches, nb = b->n_matches;
	uint8_t* curL = (uint8_t*)malloc((size_t)(na > 0 ? na : 1) * 32);
	uint8_t* preL = (uint8_t*)malloc((size_t)(nb > 0 ? nb : 1) * 32);
	float* axy = (float*)malloc(sizeof(float) * 2 * (size_t)(na > 0 ? na : 1));
	float* bxy = (float*)malloc(sizeof(float) * 2 * (size_t)(nb > 0 ? nb : 1));
	for (int k = 0; k < na; ++k) {                                   /* :1354-1358 */
		const int i = a->matches[k].queryIdx;
		memcpy(curL + 32 * k, a->desc_left + 32 * i, 32);
		axy[2 * k] = a->kp_left[i].x; axy[2 * k + 1] = a->kp_left[i].y;   /* :1905,1909-1910 */
	}
	for (int k = 0; k < nb; ++k) {                                   /* :1435-1439 */
		const int i = b->matches[k].queryIdx;
		memcpy(preL + 32 * k, b->desc_left + 32 * i, 32);
		bxy[2 * k] = b->kp_left[i].x; bxy[2 * k + 1] = b->kp_left[i].y;   /* :1906,1912-1913 */
	}
	for (int k = 0; k < na; ++k) out_status[k] = SSK_DA_TRACKED;     /* TVectorDAMatchInfo ctor */
	const int nm = ssko_hamming_match(curL, na, preL, nb, out_matches);   /* :1452 */
	int tracked = 0;
	if (nm == na && na > 0) {
		if (p->filter_by_direction)                                   /* :1482-1488 */
			ssko_da_filter_direction(out_matches, na, axy, bxy, p->rows_offset, out_status);
		if (p->filter_by_orb_distance)                                /* :1500-1506 */
			ssko_da_filter_orb_distance(out_matches, na, nb, (double)p->max_orb_distance_da, out_status);
		for (int k = 0; k < na; ++k) tracked += (out_status[k] == SSK_DA_TRACKED);   /* :1664-1668 */
	} else {
		/* other KF without matches: BFMatcher returns nothing; the reference would index an empty
		 * vector.  Oracle: nothing is tracked. */
		for (int k = 0; k < na; ++k) {
			out_matches[k].queryIdx = k; out_matches[k].trainIdx = -1; out_matches[k].imgIdx = 0;
			out_matches[k].distance = FLT_MAX; out_status[k] = SSK_DA_NON_TRACKED;
		}
	}
	free(curL); free(preL); free(axy); free(bxy);
	return tracked;
}

/* EST.cpp:572-617. 'foundIds' is a std::set in the reference; a sorted insert keeps it O(n log n)-free
 * and simple here (n <= a few hundred). */
int ssko_propagate_ids(int n_matches, int n_sim, const int32_t* tracking_first,
                       const uint64_t* const* other_ids, uint64_t* last_match_id,
                       uint64_t* out_ids, int32_t* out_tracked_per_kf)
{
	uint64_t* found = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_matches > 0 ? n_matches : 1));
	int n_found = 0, n_tracked = 0;
	for (int s = 0; s < n_sim; ++s) out_tracked_per_kf[s] = 0;
	for (int m = 0; m < n_matches; ++m) {
		int tracked = 0;
		for (int s = 0; !tracked && s < n_sim; ++s) {
			const int32_t oi = tracking_first[(size_t)s * n_matches + m];
			if (oi == -1) continue;                                   /* :582 */
			const uint64_t oid = other_ids[s][oi];                    /* :585 */
			int dup = 0;
			for (int f = 0; f < n_found; ++f) if (found[f] == oid) { dup = 1; break; }
			if (dup) break;                                           /* :587-592: leaves the k loop */
			found[n_found++] = oid;                                   /* :593 */
			out_ids[m] = oid;                                         /* :595 */
			out_tracked_per_kf[s]++;
			n_tracked++;
			tracked = 1;
		}
		if (!tracked) out_ids[m] = ++(*last_match_id);                /* :607-609 */
	}
	free(found);
	return n_tracked;
}
