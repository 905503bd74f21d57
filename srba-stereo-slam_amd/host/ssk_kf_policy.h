// ssk_kf_policy.h -- the keyframe-insertion and loop-closure policy of the reference's main loop, as pure functions on
// plain values (no device, no engine state) so that the estimator and the tests call the same code:
//   voCheck           EST.cpp:366-392   is this frame a candidate keyframe / far enough to force one
//   getSimilarKfs     EST.cpp:1737-1826 which earlier keyframes to associate with, potential loop closure
//   decideInsertion   EST.cpp:437-548   order by tracked features, confirm the loop closure, insert or adapt the thresholds
//   update*Threshold  UTIL.cpp:153-171
// CPU restatement for the tests: oracle/oracle_policy.c.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <utility>
#include <vector>

namespace ssk {
namespace policy {

static const size_t INVALID_KF_ID = (size_t)-1;                       // COMMON.h
static const size_t TOPO_FAR = std::numeric_limits<size_t>::max();     // "not in the spanning tree" (EST.cpp:1797)

inline double updateTranslationThreshold(const double x, const double th)      // UTIL.cpp:153-161
{
	double newTh = 0.02 + (0.25 / th) * x;
	newTh = newTh < 0.02 ? 0.02 : newTh;
	newTh = newTh > 0.3 ? 0.3 : newTh;
	return newTh;
}

inline double updateRotationThreshold(const double x, const double th)         // UTIL.cpp:163-171
{
	double newTh = 15 + 13 / th * (x - th);
	newTh = newTh < 2 ? 2 : newTh;
	newTh = newTh > 15 ? 15 : newTh;
	return newTh;
}

struct TVOCheck { bool voForceNewKf, voForceCheckTracking, voForceCheckDistance, voForceCheck; };

// EST.cpp:366-392.  Translations in metres, incremental rotations in radians (norm of the rotation vector), limits in degrees.
inline TVOCheck voCheck(double incTranslationKf, double incRotationKf, double incr_translation, double incr_rotation,
                        size_t tracked_feats_from_last_KF, double max_translation_limit, double max_rotation_limit,
                        double max_translation, double max_rotation, int vo_id_tracking_th)
{
	const double deg2rad = 3.14159265358979323846 / 180.0;
	TVOCheck r;
	r.voForceNewKf = incTranslationKf > max_translation_limit || incRotationKf > max_rotation_limit * deg2rad;          // :369
	r.voForceCheckTracking = vo_id_tracking_th == 0 ? false : int(tracked_feats_from_last_KF) < vo_id_tracking_th;      // :375-378
	r.voForceCheckDistance = incr_translation > max_translation || incr_rotation > max_rotation * deg2rad;              // :383-385
	r.voForceCheck = r.voForceCheckTracking || r.voForceCheckDistance;                                                  // :387
	return r;
}

struct TQuery { size_t Id; double Score; };

// EST.cpp:1737-1826 (the selection; the estimator adds the initial poses, :1828-1862).  topo(toId) = topological distance from
// the current local-map centre fromIdBase = submap_size * ((newKfId-1) / submap_size) to toId in its depth-limited spanning
// tree, TOPO_FAR when toId is not in it.  Returns "potential loop closure".
template <class TopoFn>
inline bool getSimilarKfs(size_t newKfId, const std::vector<TQuery>& dbQueryResults, size_t submap_size, size_t lc_distance, TopoFn topo,
                          std::vector<size_t>& similar_kfs, size_t& lc_id)
{
	const size_t qSize = dbQueryResults.size();
	similar_kfs.clear();
	similar_kfs.reserve(qSize + 1);
	similar_kfs.push_back(newKfId - 1);                                  // :1758 always the previous keyframe
	lc_id = INVALID_KF_ID;
	if (qSize <= 1) return false;                                        // :1760-1768 (an empty list throws in the reference, :1743)
	bool found_potential_loop_closure = false;
	const size_t fromIdBase = submap_size * ((newKfId - 1) / submap_size);   // :1779
	const double add_similar_kf_th = 0.8 * dbQueryResults[0].Score;      // :1786
	for (size_t i = 0; i < qSize; ++i) {
		const size_t toId = dbQueryResults[i].Id;
		if (toId == newKfId - 1) continue;                               // :1791
		if (dbQueryResults[i].Score > add_similar_kf_th) { similar_kfs.push_back(toId); continue; }   // :1794-1798
		const size_t topoDistance = fromIdBase == toId ? 0 : topo(toId);  // :1801-1819
		if (lc_id == INVALID_KF_ID && topoDistance > lc_distance && dbQueryResults[i].Score > 0.05) {   // :1822
			lc_id = toId;                                                // only the first one found
			found_potential_loop_closure = true;
			similar_kfs.push_back(toId);
		}
	}
	return found_potential_loop_closure;
}

struct TDaSummary { size_t kf_idx, tracked_matches; };

struct TInsertDecision {
	bool insertNewKf = false, confirmedLoopClosure = false, thresholdsUpdated = false;
	std::vector<size_t> sortedIndices;              // into the data-association list; the confirmed old keyframe first
	size_t highestNumberTrackedFeats = 0;
	double max_translation = 0, max_rotation = 0;   // the (possibly adapted) check thresholds
};

// EST.cpp:437-548.  insertNewKf_in: what the caller already decided (best query score < 0.05 forces a keyframe, :440-441).
inline TInsertDecision decideInsertion(bool insertNewKf_in, const std::vector<TDaSummary>& daInfo, bool potentialLoopClosure, size_t lc_id,
                                       bool voForceNewKf, size_t updated_matches_th, size_t up_matches_th_plus, double max_translation,
                                       double max_rotation)
{
	TInsertDecision d;
	d.insertNewKf = insertNewKf_in;
	d.max_translation = max_translation; d.max_rotation = max_rotation;
	const size_t n = daInfo.size();
	d.sortedIndices.resize(n);
	for (size_t i = 0; i < n; ++i) d.sortedIndices[i] = i;
	// DATrackedSorter (KF.h:118-126): descending tracked matches; std::sort leaves equal counts open, kept in list order here
	std::stable_sort(d.sortedIndices.begin(), d.sortedIndices.end(), [&](size_t a, size_t b) { return daInfo[a].tracked_matches > daInfo[b].tracked_matches; });
	if (n == 0) return d;
	d.highestNumberTrackedFeats = daInfo[d.sortedIndices[0]].tracked_matches;       // :472
	if (voForceNewKf) d.insertNewKf = true;                                         // :475-479
	if (potentialLoopClosure) {
		size_t loopClosureIdx = 0;
		for (size_t i = 0; !d.confirmedLoopClosure && i < n; ++i) {                 // :486-493
			d.confirmedLoopClosure = daInfo[i].kf_idx == lc_id && daInfo[i].tracked_matches > 0.5 * d.highestNumberTrackedFeats;
			if (d.confirmedLoopClosure) loopClosureIdx = i;
		}
		if (d.confirmedLoopClosure) {                                               // :495-513: priority to the old keyframe
			d.insertNewKf = true;
			for (size_t i = 0; i < d.sortedIndices.size(); ++i)
				if (d.sortedIndices[i] == loopClosureIdx) { d.sortedIndices.erase(d.sortedIndices.begin() + i); break; }
			d.sortedIndices.insert(d.sortedIndices.begin(), loopClosureIdx);
		}
	} else if (d.highestNumberTrackedFeats < updated_matches_th) {                  // :523-527
		d.insertNewKf = true;
	} else {
		const size_t olimit = updated_matches_th + up_matches_th_plus;              // :531
		if (d.highestNumberTrackedFeats <= olimit) {                                // :532-540
			d.max_translation = updateTranslationThreshold(double(d.highestNumberTrackedFeats - updated_matches_th), double(up_matches_th_plus));
			d.max_rotation = updateRotationThreshold(double(d.highestNumberTrackedFeats), double(olimit));
			d.thresholdsUpdated = true;
		}
	}
	return d;
}

// EST.cpp:625-631: similar keyframes that really contributed (prefix of the sorted list with >= half the best count)
inline size_t realNumberSimilarKfs(const std::vector<int>& vectorNumberTrackedFeats, const std::vector<size_t>& sortedIndices, size_t highestNumberTrackedFeats)
{
	size_t r = 0;
	for (size_t m = 0; m < sortedIndices.size(); ++m) {
		if (vectorNumberTrackedFeats[sortedIndices[m]] >= 0.5 * highestNumberTrackedFeats) ++r; else break;
	}
	return r;
}

}  // namespace policy
}  // namespace ssk
