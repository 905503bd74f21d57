// CSRBAStereoSLAMEstimator.cpp -- see the header.  EST.cpp = reference src/CSRBAStereoSLAMEstimator.cpp.
#include "CSRBAStereoSLAMEstimator.h"
#include "ssk_kf_io.h"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <algorithm>
#include <cstdio>
#include <iostream>
#include <set>

namespace ssk {

CSRBAStereoSLAMEstimator::CSRBAStereoSLAMEstimator(Context& ctx, const TSRBAStereoSLAMOptions& opts, const TGeneralOptions& gopts)
	: srba_options(opts), general_options(gopts), rba(ctx, opts), m_bow_manager(ctx),
	  m_max_translation(2 * opts.max_translation), m_max_rotation(2 * opts.max_rotation),                    // EST.cpp:1163-1164
	  m_max_translation_limit(opts.max_translation), m_max_rotation_limit(opts.max_rotation),
	  m_ctx(ctx), m_fast_th(opts.detect_fast_th), m_orb_th(opts.orb_max_distance) {}

// EST.cpp:263-315: the do { processNewImagePair } while (repeat) loop with the adaptive FAST / ORB thresholds
void CSRBAStereoSLAMEstimator::m_frontend(const StereoImages& stImgs, CStereoSLAMKF& kf)
{
	bool repeat;
	do {
		repeat = false;
		kf.create(m_ctx, stImgs, srba_options, m_fast_th, m_orb_th);
		const int n_matches = (int)kf.m_matches.size();
		if (!srba_options.orb_adaptive_fast_th) break;
		if (n_matches < srba_options.adaptive_th_min_matches) {                          // :273
			if (m_fast_th > FAST_TH_MIN) { m_fast_th = std::max(FAST_TH_MIN, m_fast_th - 10); repeat = true; }          // :275-280
			else if (m_orb_th < ORB_TH_MAX) { m_orb_th = std::min(ORB_TH_MAX, m_orb_th + 10); repeat = true; }          // :281-287
			else if (n_matches >= 8) { /* limits reached but a minimal set of matches exists: continue */ }             // :288-291
			else throw std::runtime_error("The number of found matches is less than the minimum. Aborting");           // :292-295
		} else if (n_matches < srba_options.adaptive_th_min_matches * 1.2) {              // :297-306
			if (m_fast_th > FAST_TH_MIN) m_fast_th = std::max(FAST_TH_MIN, m_fast_th - 5);
			else if (m_orb_th < ORB_TH_MAX) m_orb_th = std::min(ORB_TH_MAX, m_orb_th + 5);
		} else {                                                                         // :307-313
			m_fast_th = std::min(srba_options.detect_fast_th, m_fast_th + 5);
			m_orb_th = srba_options.orb_max_distance;
		}
	} while (repeat);
}

// EST.cpp:1413-1515 + 1647-1669 on the GPU: gather matched-only descriptors (Q3), Hamming K=1, direction filter,
// ORB-distance + uniqueness filter, tracking_info assembly
void CSRBAStereoSLAMEstimator::m_internal_data_association(const CStereoSLAMKF& this_kf, const CStereoSLAMKF& other_kf, t_kf_da_info& out_da,
                                                           const Pose3D* other_seen_from_this, ssk_cip_result* out_cip)
{
	if (out_cip) std::memset(out_cip, 0, sizeof *out_cip);
	const size_t n = this_kf.m_matches.size();
	out_da.kf_idx = other_kf.m_kf_ID;                                                    // :1427
	out_da.tracked_matches = 0;
	out_da.tracking_info.assign(n, std::make_pair(INVALID_IDX, 0.0));                    // :1429
	if (n == 0) return;
	std::vector<DMatch> matches(n);
	std::vector<int32_t> status(n);
	const ssk_kf_view a = this_kf.view(), b = other_kf.view();
	const ssk_da_params p = srba_options.da();
	// the association, then stage 2 (EST.cpp:1587-1640): fundamental-matrix filter (m_detect_outliers_with_F, :2015-2055; the RANSAC
	// samples come from a counter-based generator seeded by the keyframe pair, so a run is reproducible), then the change-in-pose
	// filter (m_detect_outliers_with_change_in_pose, :2113-2177), started from the caller's estimate of the other camera seen
	// from this one (similar_kfs_poses, :1828-1862) or, without one, from the identity: keyframe-to-keyframe motion is small
	// against the scene depth, and the estimator converges from there.  One call: the stages run back to back on the device
	// (one upload, one synchronisation), the reference's gates between them are applied here on the counts it returns.
	const bool fm_on = srba_options.da_filter_by_fund_matrix, cip_on = srba_options.da_filter_by_pose_change;
	const ssk_fm_params fp = srba_options.fm(0x5EED0000ull + ((uint64_t)this_kf.m_kf_ID << 20) + other_kf.m_kf_ID);
	const ssk_cip_params cp = srba_options.cip();
	const Pose3D identity;
	const double* init = (other_seen_from_this ? *other_seen_from_this : identity).m;
	ssk_da_staged_out so;
	m_ctx.check(ssk_data_association_staged(m_ctx.get(), &a, &b, &p, fm_on ? &fp : nullptr, cip_on ? &cp : nullptr, init,
	                                        matches.data(), status.data(), &so), "m_internal_data_association");
	if (fm_on && so.tracked_da < 15) return;                                             // :1590-1594  "not enough input data": invalid
	if (cip_on && so.tracked_fm < 15) return;                                            // :1617-1621  (filter2_inliers, :1610-1613)
	if (cip_on) {
		if (out_cip) *out_cip = so.cip;
	} else if (out_cip) {
		// the odometry stand-in asks for the pose estimate whatever da_stage2_method says (stereo-vo estimates its incremental
		// pose independently of the SLAM-side filters): the same estimator on a COPY of the statuses, so that it filters nothing
		std::vector<int32_t> st2(status);
		m_ctx.check(ssk_da_change_in_pose(m_ctx.get(), &a, &b, matches.data(), st2.data(), init, &cp, nullptr, out_cip), "m_vo_process: incremental pose");
	}
	for (size_t k = 0; k < n; ++k)                                                       // :1657-1669
		if (status[k] == SSK_DA_TRACKED) {
			out_da.tracking_info[k] = std::make_pair((size_t)matches[k].trainIdx, (double)matches[k].distance);
			out_da.tracked_matches++;
		}
}

// The candidate against every similar keyframe (EST.cpp:1341-1411): the reference loops over m_internal_data_association; here
// the S pairs go to the device together (ssk_data_association_staged_multi: one upload, one launch per stage, one synchronisation)
// and the per-pair logic of m_internal_data_association -- gates and tracking_info assembly -- runs on what comes back.
void CSRBAStereoSLAMEstimator::m_data_association_pairs(const CStereoSLAMKF& kf, const std::vector<size_t>& others, const std::vector<Pose3D>* poses,
                                                        TVectorKfsDaInfo& out_da)
{
	const size_t S = others.size(), n = kf.m_matches.size();
	out_da.assign(S, t_kf_da_info());                                                    // :1346-1347
	for (size_t k = 0; k < S; ++k) {
		out_da[k].kf_idx = m_keyframes[others[k]].m_kf_ID;                               // :1427
		out_da[k].tracked_matches = 0;
		out_da[k].tracking_info.assign(n, std::make_pair(INVALID_IDX, 0.0));             // :1429
	}
	if (S == 0 || n == 0) return;
	const bool fm_on = srba_options.da_filter_by_fund_matrix, cip_on = srba_options.da_filter_by_pose_change;
	const ssk_kf_view a = kf.view();
	std::vector<ssk_kf_view> views(S);
	std::vector<uint64_t> seeds(S);
	std::vector<double> init(12 * S);
	const Pose3D identity;
	for (size_t k = 0; k < S; ++k) {
		const CStereoSLAMKF& o = m_keyframes[others[k]];
		views[k] = o.view();
		seeds[k] = 0x5EED0000ull + ((uint64_t)kf.m_kf_ID << 20) + o.m_kf_ID;
		std::memcpy(&init[12 * k], (poses && k < poses->size() ? (*poses)[k] : identity).m, 12 * sizeof(double));
	}
	const ssk_da_params p = srba_options.da();
	const ssk_fm_params fp = srba_options.fm(0);
	const ssk_cip_params cp = srba_options.cip();
	std::vector<DMatch> matches(S * n);
	std::vector<int32_t> status(S * n);
	std::vector<ssk_da_staged_out> so(S);
	m_ctx.check(ssk_data_association_staged_multi(m_ctx.get(), &a, views.data(), (int)S, &p, fm_on ? &fp : nullptr, seeds.data(), cip_on ? &cp : nullptr,
	                                              init.data(), matches.data(), status.data(), so.data()), "m_data_association");
	for (size_t k = 0; k < S; ++k) {
		if (fm_on && so[k].tracked_da < 15) continue;                                    // :1590-1594  "not enough input data": invalid
		if (cip_on && so[k].tracked_fm < 15) continue;                                   // :1617-1621
		for (size_t m = 0; m < n; ++m)                                                   // :1657-1669
			if (status[k * n + m] == SSK_DA_TRACKED) {
				out_da[k].tracking_info[m] = std::make_pair((size_t)matches[k * n + m].trainIdx, (double)matches[k * n + m].distance);
				out_da[k].tracked_matches++;
			}
	}
}

void CSRBAStereoSLAMEstimator::m_data_association(const CStereoSLAMKF& kf, const std::vector<size_t>& similar_kfs, TVectorKfsDaInfo& out_da)
{
	m_data_association_pairs(kf, similar_kfs, nullptr, out_da);                         // :1365-1398
}

void CSRBAStereoSLAMEstimator::m_data_association(const CStereoSLAMKF& kf, const TLoopClosureInfo& info, TVectorKfsDaInfo& out_da)
{
	m_data_association_pairs(kf, info.similar_kfs, &info.similar_kfs_poses, out_da);    // :1365-1398
}

// EST.cpp:142-161 / 667-704: observation list; the pixel + initial-position arithmetic (projectMatchTo3D and the
// camera pose composition) runs on the GPU (ssk_build_observations)
void CSRBAStereoSLAMEstimator::m_build_observations(const CStereoSLAMKF& kf, new_kf_observations_t& listObs)
{
	const size_t n = kf.m_matches.size();
	listObs.assign(n, new_kf_observation_t());
	if (n == 0) return;
	std::vector<float> px(4 * n);
	std::vector<double> pos(3 * n);
	const ssk_lm_params cam = srba_options.lm(true);
	m_ctx.check(ssk_build_observations(m_ctx.get(), kf.m_keypoints_left.data(), kf.m_keypoints_right.data(), kf.m_matches.data(), (int)n,
	                                   &cam, px.data(), pos.data()), "m_build_observations");
	for (size_t m = 0; m < n; ++m) {
		new_kf_observation_t& o = listObs[m];
		o.is_fixed = false;                                                              // :143
		o.is_unknown_with_init_val = true;                                               // :144
		o.obs.feat_id = kf.m_matches_ID[m];                                              // :153
		o.obs.obs_data.left_px.x = px[4 * m]; o.obs.obs_data.left_px.y = px[4 * m + 1];  // :154
		o.obs.obs_data.right_px.x = px[4 * m + 2]; o.obs.obs_data.right_px.y = px[4 * m + 3];   // :155
		o.setRelPos(&pos[3 * m]);                                                        // :158
	}
}

TNewKeyFrameInfo CSRBAStereoSLAMEstimator::processKeyframe(const StereoImages& stImgs, const std::vector<size_t>& similar_in, const Pose3D& incr_pose)
{
	m_keyframes.push_back(CStereoSLAMKF());                                              // :84 / :413
	CStereoSLAMKF& new_kf = m_keyframes.back();
	new_kf.setKFID(m_last_kf_ID++);
	static const bool timing = std::getenv("SSK_FACADE_TIMING") != nullptr;   // per-stage wall time of a keyframe on stderr
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const auto t0 = now();
	m_frontend(stImgs, new_kf);
	const auto t1 = now();
	TNewKeyFrameInfo newKFInfo;
	new_kf_observations_t listObs;
	const size_t nMatches = new_kf.m_matches.size();
	if (m_keyframes.size() == 1) {
		// ---- first keyframe, EST.cpp:82-182 ----
		new_kf.generateMatchesIDs(0);                                                    // IDs as the VO engine hands them over (KF.h:75-80)
		m_last_match_ID = nMatches ? new_kf.m_matches_ID.back() : 0;                     // :115
		if (general_options.debug) new_kf.saveInfoToFiles(general_options);              // :130-131
		m_build_observations(new_kf, listObs);                                           // :134-161
		rba.define_new_keyframe(listObs, newKFInfo, false);                              // :166-168: not optimised the first time
		return newKFInfo;
	}
	// ---- data association with the similar keyframes, EST.cpp:447-450 ----
	std::vector<size_t> similar = similar_in;
	if (similar.empty()) similar.push_back(m_keyframes.size() - 2);                      // the previous keyframe (:1758)
	TVectorKfsDaInfo daInfo;
	m_data_association(new_kf, similar, daInfo);
	const auto t2 = now();
	// order by tracked matches, descending (DATrackedSorter, KF.h:117-126; EST.cpp:455-470)
	std::vector<size_t> sortedIndices(daInfo.size());
	for (size_t i = 0; i < sortedIndices.size(); ++i) sortedIndices[i] = i;
	std::stable_sort(sortedIndices.begin(), sortedIndices.end(), [&](size_t a, size_t b) { return daInfo[a].tracked_matches > daInfo[b].tracked_matches; });
	m_insert_keyframe(new_kf, daInfo, sortedIndices, incr_pose, newKFInfo);
	if (timing) std::fprintf(stderr, "[facade timing] kf %zu: front-end %.2f ms, data association %.2f ms, insertion (ids, observations, define_new_keyframe) %.2f ms\n",
	                         new_kf.m_kf_ID, ms(t0, t1), ms(t1, t2), ms(t2, now()));
	return newKFInfo;
}

// EST.cpp:560-850 for an accepted candidate: ids from the similar keyframes in `sortedIndices` order, observation list, SRBA
void CSRBAStereoSLAMEstimator::m_insert_keyframe(CStereoSLAMKF& new_kf, const TVectorKfsDaInfo& daInfo, const std::vector<size_t>& sortedIndices,
                                                 const Pose3D& incr_pose, TNewKeyFrameInfo& newKFInfo)
{
	const size_t nMatches = new_kf.m_matches.size();
	new_kf_observations_t listObs;
	// ---- set ids for the current keyframe matches, EST.cpp:572-617 ----
	new_kf.m_matches_ID.assign(nMatches, 0);
	std::set<size_t> foundIds;
	for (size_t m = 0; m < nMatches; ++m) {
		bool tracked = false;
		for (size_t k = 0; !tracked && k < daInfo.size(); ++k) {
			const t_kf_da_info& da_info = daInfo[sortedIndices[k]];
			if (da_info.tracking_info[m].first != INVALID_IDX) {                         // :582
				const size_t other_match_idx = da_info.tracking_info[m].first;
				const size_t other_match_id = m_keyframes[da_info.kf_idx].m_matches_ID[other_match_idx];   // :585
				if (foundIds.count(other_match_id)) break;                               // :587-592
				foundIds.insert(other_match_id);                                         // :593
				new_kf.m_matches_ID[m] = other_match_id;                                 // :595
				tracked = true;
			}
		}
		if (!tracked) new_kf.m_matches_ID[m] = ++m_last_match_ID;                        // :607-611
	}
	if (general_options.debug) new_kf.saveInfoToFiles(general_options);                  // :735
	// ---- observation list + SRBA, EST.cpp:667-704, 770-784 ----
	m_build_observations(new_kf, listObs);
	if (general_options.debug) {                                                         // :751-763 da_info_%04d.txt = the exact SRBA input
		char name[1024];
		std::snprintf(name, sizeof name, "%s/da_info_%04d.txt", general_options.out_dir.c_str(), (int)new_kf.m_kf_ID);
		if (FILE* f = std::fopen(name, "wt")) {
			for (const auto& o : listObs)
				std::fprintf(f, "%llu %.2f %.2f %.2f %.2f\n", (unsigned long long)o.obs.feat_id, o.obs.obs_data.left_px.x, o.obs.obs_data.left_px.y,
				             o.obs.obs_data.right_px.x, o.obs.obs_data.right_px.y);
			std::fclose(f);
		}
	}
	rba.setInitialKFPose(srba_options.use_initial_pose ? incr_pose : Pose3D());          // :772-778
	rba.define_new_keyframe(listObs, newKFInfo, true);                                   // :782-784
	if (general_options.verbose_level >= 1)                                              // :846-850
		std::cout << "   Created KF #" << newKFInfo.kf_id << " | # kf-to-kf edges created: " << newKFInfo.created_edge_ids.size() << std::endl
		          << "   Optimization error: " << newKFInfo.optimize_results.total_sqr_error_init << " -> " << newKFInfo.optimize_results.total_sqr_error_final << std::endl;
}

// ---- stand-in for rso::CStereoOdometryEstimator (stereo-vo, absent: README.md:16) --------------------------------------
// What the loop reads of it: the current frame's features (getDataFromVOEngine, KF.h:68), m_odom_result.{valid, outPose,
// tracked_feats_from_last_frame, tracked_feats_from_last_KF} (EST.cpp:317-378) and resetIds() (EST.cpp:403,622).  Built from
// this library's stages: front-end with the adaptive thresholds on the new pair, Hamming association with the previous
// frame + the data-association filters, and the change-in-pose estimate as the incremental pose.  A match keeps its track
// id while it is re-found from frame to frame; "from last KF" counts the ids that existed at the last resetIds().
TOdometryResult CSRBAStereoSLAMEstimator::m_vo_process(const StereoImages& stImgs)
{
	TOdometryResult r;
	if (m_vo_has_prev) { std::swap(m_vo_prev, m_vo_cur); m_vo_ids_prev.swap(m_vo_ids_cur); }
	m_vo_cur = CStereoSLAMKF();
	m_frontend(stImgs, m_vo_cur);                                                        // EST.cpp:263-315
	const size_t n = m_vo_cur.m_matches.size();
	m_vo_ids_cur.assign(n, 0);
	if (!m_vo_has_prev) {
		for (size_t k = 0; k < n; ++k) m_vo_ids_cur[k] = m_vo_next_id++;
		m_vo_kf_max_id = m_vo_next_id;                                                    // ids below this existed at the last reset
		m_vo_has_prev = true;
		r.valid = true;                                                                   // first pair: nothing to compare with
		return r;
	}
	t_kf_da_info da;
	ssk_cip_result cip;
	m_internal_data_association(m_vo_cur, m_vo_prev, da, nullptr, &cip);
	for (size_t k = 0; k < n; ++k) {
		if (da.tracking_info[k].first != INVALID_IDX) {
			m_vo_ids_cur[k] = m_vo_ids_prev[da.tracking_info[k].first];
			r.tracked_feats_from_last_frame++;
			if (m_vo_ids_cur[k] < m_vo_kf_max_id) r.tracked_feats_from_last_KF++;
		} else {
			m_vo_ids_cur[k] = m_vo_next_id++;
		}
	}
	r.valid = cip.valid != 0;
	if (r.valid) {
		Pose3D prev_seen_from_cur;
		std::memcpy(prev_seen_from_cur.m, cip.pose, sizeof cip.pose);
		r.outPose = prev_seen_from_cur.inverse();                                         // this frame seen from the previous one (:324)
	}
	return r;
}

void CSRBAStereoSLAMEstimator::m_vo_reset_ids()
{
	// "this will define previous frame as a virtual KF for the visual odometry system so that it can start a new record of
	// tracked matches from last KF" (EST.cpp:400-403)
	for (size_t k = 0; k < m_vo_ids_cur.size(); ++k) m_vo_ids_cur[k] = k;
	m_vo_next_id = m_vo_kf_max_id = m_vo_ids_cur.size();
}

// EST.cpp:1737-1878
bool CSRBAStereoSLAMEstimator::m_get_similar_kfs(const size_t newKfId, const QueryResults& dbQueryResults, TLoopClosureInfo& out)
{
	if (dbQueryResults.empty())                                                          // :1743-1744
		throw std::runtime_error("Parameter 'dbQueryResults' is empty. This method should not be called here.");
	if (dbQueryResults[0].Score < srba_options.query_score_th && general_options.verbose_level >= 1)          // :1748-1751
		std::cout << "[Warning] Best result in 'dbQueryResults' is below a minimum threshold. Lost camera?" << std::endl;
	std::vector<policy::TQuery> q(dbQueryResults.size());
	for (size_t i = 0; i < q.size(); ++i) { q[i].Id = dbQueryResults[i].Id; q[i].Score = dbQueryResults[i].Score; }
	const size_t SUBMAP_SIZE = rba.parameters.submap_size;                               // :1778
	const size_t fromIdBase = SUBMAP_SIZE * ((newKfId - 1) / SUBMAP_SIZE);               // :1779
	const bool found = policy::getSimilarKfs(newKfId, q, SUBMAP_SIZE, srba_options.lc_distance,
	                                         [&](size_t toId) { return rba.topo_distance(fromIdBase, toId, rba.parameters.max_tree_depth); },
	                                         out.similar_kfs, out.lc_id);
	// :1828-1862 initial camera-to-camera poses of the similar keyframes seen from the candidate
	out.similar_kfs_poses.assign(out.similar_kfs.size(), Pose3D());
	if (srba_options.da_filter_by_pose_change) {
		const Pose3D incr_inv = m_incr_pose_from_last_kf.inverse();
		const Pose3D& cam = srba_options.camera_pose_on_robot_rvt;
		const Pose3D cam_inv = cam.inverse();
		for (size_t k = 0; k < out.similar_kfs.size(); ++k) {
			if (out.similar_kfs[k] == newKfId - 1) { out.similar_kfs_poses[k] = incr_inv; continue; }                // :1846
			Pose3D rel;                                                                   // similar keyframe seen from the previous one
			if (rba.get_kf_relative_pose(newKfId - 1, out.similar_kfs[k], rba.parameters.max_tree_depth, rel))          // :1850-1858
				out.similar_kfs_poses[k] = incr_inv + (cam_inv + rel + cam);
		}
	}
	return found;
}

// One iteration of the reference's while loop, EST.cpp:44-937
TFrameReport CSRBAStereoSLAMEstimator::performStereoSLAM(const StereoImages& stImgs)
{
	TFrameReport rep;
	rep.frame = m_frame_count++;
	if (!m_bow_manager.hasVocabulary()) throw std::runtime_error("performStereoSLAM: no vocabulary loaded (m_bow_manager.loadVocabulary)");
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const auto t_0 = now();
	rep.odometry = m_vo_process(stImgs);
	timing.vo_ms += ms(t_0, now()); timing.frames++;
	// ---- first iteration, EST.cpp:82-182 ----
	if (m_keyframes.empty()) {
		m_keyframes.push_back(m_vo_cur);                                                 // new_kf.getDataFromVOEngine (:112-114)
		CStereoSLAMKF& new_kf = m_keyframes.back();
		new_kf.setKFID(m_last_kf_ID++);
		new_kf.m_matches_ID.clear();
		new_kf.generateMatchesIDs(0);
		const size_t nMatches = new_kf.m_matches.size();
		m_last_match_ID = nMatches ? new_kf.m_matches_ID.back() : 0;                     // :115
		m_bow_manager.insertIntoDB(new_kf);                                              // :118
		if (general_options.debug) new_kf.saveInfoToFiles(general_options);
		new_kf_observations_t listObs;
		m_build_observations(new_kf, listObs);
		rba.define_new_keyframe(listObs, rep.kf_info, false);                            // :166-168
		m_last_kf_pose = new_kf.m_camera_pose;                                           // :182
		m_vo_reset_ids();
		rep.checked_for_kf = rep.inserted_kf = true;
		return rep;
	}
	if (!rep.odometry.valid) return rep;                                                 // :318-323 "Not a valid result! Skipping this frame."
	const Pose3D& incr_pose = rep.odometry.outPose;                                      // :327-330
	m_current_pose = m_current_pose + incr_pose;
	m_incr_pose_from_last_kf = m_incr_pose_from_last_kf + incr_pose;
	m_incr_pose_from_last_check = m_incr_pose_from_last_check + incr_pose;
	const policy::TVOCheck vo = policy::voCheck(m_incr_pose_from_last_kf.translationNorm(), m_incr_pose_from_last_kf.rotationAngle(),
	                                            m_incr_pose_from_last_check.translationNorm(), m_incr_pose_from_last_check.rotationAngle(),
	                                            rep.odometry.tracked_feats_from_last_KF, m_max_translation_limit, m_max_rotation_limit,
	                                            m_max_translation, m_max_rotation, srba_options.vo_id_tracking_th);        // :366-392
	rep.vo_force_new_kf = vo.voForceNewKf; rep.vo_force_check_tracking = vo.voForceCheckTracking; rep.vo_force_check_distance = vo.voForceCheckDistance;
	if (!(vo.voForceNewKf || vo.voForceCheck)) return rep;                               // :394
	rep.checked_for_kf = true;
	if (vo.voForceCheckTracking) m_vo_reset_ids();                                       // :398-404
	m_incr_pose_from_last_check = Pose3D();                                              // :407
	m_keyframes.push_back(m_vo_cur);                                                     // :413-415 a temporary keyframe
	CStereoSLAMKF* new_kf = &m_keyframes.back();
	new_kf->setKFID(m_last_kf_ID);                                                       // :418
	new_kf->m_matches_ID.clear();
	const auto t_q = now();
	m_bow_manager.queryDB(*new_kf, rep.query, 4);                                        // :424-427
	timing.query_ms += ms(t_q, now());
	const bool potentialLoopClosure = m_get_similar_kfs(new_kf->m_kf_ID, rep.query, rep.similar);   // :434-436
	rep.potential_loop_closure = potentialLoopClosure;
	const bool insert_in = rep.query[0].Score < 0.05;                                    // :439-441 "result is too low, force the creation of a new KF"
	TVectorKfsDaInfo daInfo;
	const auto t_d = now();
	m_data_association(*new_kf, rep.similar, daInfo);                                    // :446-451
	timing.da_ms += ms(t_d, now());
	std::vector<policy::TDaSummary> das(daInfo.size());
	for (size_t i = 0; i < das.size(); ++i) { das[i].kf_idx = daInfo[i].kf_idx; das[i].tracked_matches = daInfo[i].tracked_matches; }
	const policy::TInsertDecision dec = policy::decideInsertion(insert_in, das, potentialLoopClosure, rep.similar.lc_id, vo.voForceNewKf,
	                                                            srba_options.updated_matches_th, srba_options.up_matches_th_plus,
	                                                            m_max_translation, m_max_rotation);                        // :455-548
	rep.highest_tracked = dec.highestNumberTrackedFeats;
	rep.da = daInfo; rep.da_sorted = dec.sortedIndices;
	rep.confirmed_loop_closure = dec.confirmedLoopClosure;
	m_max_translation = dec.max_translation; m_max_rotation = dec.max_rotation;
	if (dec.confirmedLoopClosure) { rba.loopClosureDetected(); rba.setLoopClosureOldID(rep.similar.lc_id); }           // :497-498
	else rba.loopClosureDetected(false);                                                 // :518,546
	if (!dec.insertNewKf) {                                                              // :558-562 undo the insertion
		m_keyframes.resize(m_keyframes.size() - 1);
		return rep;
	}
	const auto t_i = now();
	m_insert_keyframe(*new_kf, daInfo, dec.sortedIndices, m_incr_pose_from_last_kf, rep.kf_info);                     // :564-850
	timing.insert_ms += ms(t_i, now()); timing.kf_frames++;
	m_vo_reset_ids();                                                                    // :620-622
	m_last_kf_ID++;                                                                      // :657
	m_max_translation = srba_options.max_translation;                                    // :660-663 as written: the CHECK thresholds restart from
	m_max_rotation = srba_options.max_rotation;                                          // the limits (the constructor starts them at twice that)
	Pose3D pose;                                                                         // :909-915 pose of the new keyframe from keyframe 0
	if (rba.get_kf_relative_pose(0, new_kf->m_kf_ID, rba.parameters.max_tree_depth, pose)) new_kf->m_camera_pose = m_last_kf_pose = pose;
	m_bow_manager.insertIntoDB(*new_kf);                                                 // :919
	m_current_pose = m_last_kf_pose + srba_options.camera_pose_on_robot_rvt;             // :922
	m_incr_pose_from_last_kf = Pose3D();                                                 // :923
	rep.inserted_kf = true;
	return rep;
}

// EST.cpp:945-987
std::vector<Pose3D> CSRBAStereoSLAMEstimator::optimizeGlobalPoseGraph(ssk_pg_result* info, const std::string& path)
{
	std::vector<Pose3D> nodes, edge_z;
	std::vector<int32_t> edge_ij;
	std::vector<bool> reachable;
	rba.get_global_graphslam_problem(0, nodes, edge_ij, edge_z, reachable);
	ssk_pg_result res;
	std::memset(&res, 0, sizeof res);
	if (!nodes.empty()) {
		ssk_pg_params prm;
		std::memset(&prm, 0, sizeof prm);                              // solver 0: direct (spanning tree + Woodbury) when it applies
		prm.max_iters = 100; prm.tau = 1e-3;                         // mrpt::graphslam defaults (max_iterations, tau)
		prm.pcg_max_iters = 1000;                                      // CG fallback, tree-preconditioned: ~6 iterations per edge off the spanning tree
		prm.max_lambda = 1e20; prm.converged_rel_decrease = 1e-9; prm.pcg_tol = 1e-10;
		static_assert(sizeof(Pose3D) == 12 * sizeof(double), "Pose3D is 12 doubles");
		m_ctx.check(ssk_posegraph_optimize(m_ctx.get(), (int)nodes.size(), nodes[0].m, (int)edge_z.size(), edge_ij.data(),
		                                   edge_z.empty() ? nullptr : edge_z[0].m, 0, &prm, &res), "optimize_graph_spa_levmarq");
	}
	if (info) *info = res;
	if (!path.empty()) {
		if (FILE* f = std::fopen(path.c_str(), "wt")) {                // EST.cpp:976-984
			for (size_t kf = 0; kf < nodes.size(); ++kf) {
				const double* m = nodes[kf].m;
				const double yaw = std::atan2(m[4], m[0]), pitch = std::atan2(-m[8], std::sqrt(m[0] * m[0] + m[4] * m[4])), roll = std::atan2(m[9], m[10]);
				std::fprintf(f, "%d %.3f %.3f %.3f %.3f %.3f %.3f\n", (int)kf, m[3], m[7], m[11], yaw, pitch, roll);
			}
			std::fclose(f);
		}
	}
	return nodes;
}

bool CSRBAStereoSLAMEstimator::saveKeyframes(const std::string& path) const
{
	std::ofstream f(path, std::ios::binary);
	if (!f) return false;
	kfio::put<uint64_t>(f, (uint64_t)m_keyframes.size());                                // EST.cpp:2678-2679
	for (const CStereoSLAMKF& kf : m_keyframes)
		if (!kfio::dump_keyframe(f, kf)) return false;
	return (bool)f;
}

bool CSRBAStereoSLAMEstimator::loadKeyframes(const std::string& path)
{
	std::ifstream f(path, std::ios::binary);
	uint64_t n = 0;
	if (!f || !kfio::get(f, n) || n > (1u << 20)) return false;
	t_vector_kf kfs((size_t)n);
	for (CStereoSLAMKF& kf : kfs)
		if (!kfio::load_keyframe(f, kf)) return false;
	m_keyframes.swap(kfs);
	return true;
}

}  // namespace ssk
