/*
 * oracle_policy.c -- CPU oracle (test infrastructure; see ssk_oracle.h): the keyframe-insertion / loop-closure policy of the
 * reference's main loop, restated literally from in-tree code [literal file:line]: no third-party arithmetic is involved.
 * Open choice fixed here: DATrackedSorter is used with std::sort (EST.cpp:458), which leaves the order of keyframes with
 * EQUAL tracked counts unspecified; they keep their list order here (and in the product).
 */
#include "ssk_oracle.h"
#include <math.h>
#include <stdlib.h>

/* [literal UTIL.cpp:153-161] */
double ssko_update_translation_threshold(double x, double th)
{
	double newTh = 0.02 + (0.25 / th) * x;
	newTh = newTh < 0.02 ? 0.02 : newTh;
	newTh = newTh > 0.3 ? 0.3 : newTh;
	return newTh;
}

/* [literal UTIL.cpp:163-171] */
double ssko_update_rotation_threshold(double x, double th)
{
	double newTh = 15 + 13 / th * (x - th);
	newTh = newTh < 2 ? 2 : newTh;
	newTh = newTh > 15 ? 15 : newTh;
	return newTh;
}

/* [literal EST.cpp:366-392]; out[4] = voForceNewKf, voForceCheckTracking, voForceCheckDistance, voForceCheck */
void ssko_vo_check(double inc_translation_kf, double inc_rotation_kf, double incr_translation, double incr_rotation,
                   int64_t tracked_feats_from_last_kf, double max_translation_limit, double max_rotation_limit,
                   double max_translation, double max_rotation, int vo_id_tracking_th, int32_t* out)
{
	const double deg2rad = 3.14159265358979323846 / 180.0;
	out[0] = inc_translation_kf > max_translation_limit || inc_rotation_kf > max_rotation_limit * deg2rad;
	out[1] = vo_id_tracking_th == 0 ? 0 : (int)tracked_feats_from_last_kf < vo_id_tracking_th;
	out[2] = incr_translation > max_translation || incr_rotation > max_rotation * deg2rad;
	out[3] = out[1] || out[2];
}

/* [literal EST.cpp:1737-1826] topo[i]: topological distance from the current local-map centre to q_ids[i], -1 = not in its
 * spanning tree (numeric_limits max in the reference).  similar has capacity n_q + 1.  Returns potential loop closure. */
int ssko_get_similar_kfs(int64_t new_kf_id, int n_q, const int64_t* q_ids, const double* q_scores, const int64_t* topo,
                         int64_t submap_size, int64_t lc_distance, int64_t* similar, int* n_similar, int64_t* lc_id)
{
	int n = 0, found = 0;
	similar[n++] = new_kf_id - 1;
	*lc_id = -1;
	if (n_q <= 1) { *n_similar = n; return 0; }
	const int64_t from_id_base = submap_size * ((new_kf_id - 1) / submap_size);
	const double add_similar_kf_th = 0.8 * q_scores[0];
	for (int i = 0; i < n_q; ++i) {
		const int64_t to_id = q_ids[i];
		if (to_id == new_kf_id - 1) continue;
		if (q_scores[i] > add_similar_kf_th) { similar[n++] = to_id; continue; }
		const int far = from_id_base == to_id ? 0 : (topo[i] < 0 || topo[i] > lc_distance);
		if (*lc_id == -1 && far && q_scores[i] > 0.05) { *lc_id = to_id; found = 1; similar[n++] = to_id; }
	}
	*n_similar = n;
	return found;
}

/* [literal EST.cpp:437-548] da_kf_idx / da_tracked: the data-association list in its own order.  sorted has capacity n_da.
 * flags_out[3] = insertNewKf, confirmedLoopClosure, thresholdsUpdated; th_inout[2] = max_translation, max_rotation. */
void ssko_kf_decision(int insert_in, int n_da, const int64_t* da_kf_idx, const int64_t* da_tracked, int potential_lc, int64_t lc_id,
                      int vo_force_new_kf, int64_t updated_matches_th, int64_t up_matches_th_plus, double* th_inout,
                      int64_t* sorted, int64_t* highest_out, int32_t* flags_out)
{
	int insert = insert_in, confirmed = 0, updated = 0;
	for (int i = 0; i < n_da; ++i) sorted[i] = i;
	for (int i = 1; i < n_da; ++i) {                          /* stable insertion sort, descending tracked count */
		const int64_t v = sorted[i];
		int j = i - 1;
		while (j >= 0 && da_tracked[sorted[j]] < da_tracked[v]) { sorted[j + 1] = sorted[j]; --j; }
		sorted[j + 1] = v;
	}
	*highest_out = 0;
	flags_out[0] = insert; flags_out[1] = 0; flags_out[2] = 0;
	if (n_da == 0) return;
	const int64_t highest = da_tracked[sorted[0]];
	*highest_out = highest;
	if (vo_force_new_kf) insert = 1;
	if (potential_lc) {
		int lc_idx = 0;
		for (int i = 0; !confirmed && i < n_da; ++i) {
			confirmed = da_kf_idx[i] == lc_id && (double)da_tracked[i] > 0.5 * (double)highest;
			if (confirmed) lc_idx = i;
		}
		if (confirmed) {
			insert = 1;
			int at = 0;
			while (sorted[at] != lc_idx) ++at;
			for (int i = at; i > 0; --i) sorted[i] = sorted[i - 1];
			sorted[0] = lc_idx;
		}
	} else if (highest < updated_matches_th) {
		insert = 1;
	} else {
		const int64_t olimit = updated_matches_th + up_matches_th_plus;
		if (highest <= olimit) {
			th_inout[0] = ssko_update_translation_threshold((double)(highest - updated_matches_th), (double)up_matches_th_plus);
			th_inout[1] = ssko_update_rotation_threshold((double)highest, (double)olimit);
			updated = 1;
		}
	}
	flags_out[0] = insert; flags_out[1] = confirmed; flags_out[2] = updated;
}
