// CSRBAStereoSLAMEstimator.h -- the per-keyframe hot path of the reference's estimator
// (src/CSRBAStereoSLAMEstimator.h:33-204, .cpp) behind the same names: srba_options, rba, m_keyframes,
// m_data_association, m_internal_data_association, the landmark-ID propagation and the observation
// marshalling, with the numerical work on the MI355X through the C-ABI.
//
// Two ways in:
//   processKeyframe()     the caller has decided that this stereo pair is a keyframe and which earlier keyframes to
//                         associate with (default: the previous one, as EST.cpp:1758 always includes)
//   performStereoSLAM()   one iteration of the reference's frame loop (EST.cpp:44-937): odometry between consecutive
//                         frames, keyframe check, bag-of-words query (CBoWManager), similar keyframes and potential loop
//                         closure (m_get_similar_kfs), data association, loop-closure confirmation and the insertion
//                         decision (host/ssk_kf_policy.h), then the same insertion path as processKeyframe()
// NOT here: the frame grabber, GUI, .ini parsing, application-state save/restore beyond the keyframes.
// stereo-vo's CStereoOdometryEstimator is absent from the image (un-vendored, unpinned): the frame-to-frame odometry the
// loop reads (valid, outPose, tracked_feats_from_last_frame / _KF, resetIds) is a build-side stand-in made of this
// library's own stages (front-end, Hamming association + filters, change-in-pose estimate) -- see m_vo_process.
#pragma once
#include "CBoWManager.h"
#include "CStereoSLAMKF.h"
#include "srba_stereo_slam.h"
#include "ssk_kf_policy.h"

namespace ssk {

struct t_kf_da_info {                                      // UTIL.h:55-66
	size_t kf_idx = (size_t)-1;
	size_t tracked_matches = 0;
	std::vector<std::pair<size_t, double>> tracking_info;  // (other match idx or INVALID_IDX, distance)
};
typedef std::vector<t_kf_da_info> TVectorKfsDaInfo;
static const size_t INVALID_IDX = (size_t)-1;              // COMMON.h:88 (-1)

struct TLoopClosureInfo {                                  // UTIL.h:68-76
	std::vector<size_t> similar_kfs;
	std::vector<Pose3D> similar_kfs_poses;                 // the similar keyframe's camera seen from the candidate's (initial value)
	size_t lc_id = policy::INVALID_KF_ID;
};

struct TOdometryResult {                                   // the fields of rso::...::TStereoOdometryResult the loop reads
	bool valid = false;
	Pose3D outPose;                                        // this frame's camera seen from the previous frame's
	size_t tracked_feats_from_last_frame = 0, tracked_feats_from_last_KF = 0;
};

struct TFrameReport {                                      // what one loop iteration decided (the reference prints it)
	size_t frame = 0;
	TOdometryResult odometry;
	bool checked_for_kf = false, inserted_kf = false, potential_loop_closure = false, confirmed_loop_closure = false;
	bool vo_force_new_kf = false, vo_force_check_tracking = false, vo_force_check_distance = false;
	QueryResults query;
	TLoopClosureInfo similar;
	size_t highest_tracked = 0;
	TNewKeyFrameInfo kf_info;
	TVectorKfsDaInfo da;                                   // data association with rep.similar.similar_kfs, in that order (EST.cpp:446-451)
	std::vector<size_t> da_sorted;                         // the order the ids were taken from them (DATrackedSorter, EST.cpp:455-470)
};

class CSRBAStereoSLAMEstimator {
public:
	CSRBAStereoSLAMEstimator(Context& ctx, const TSRBAStereoSLAMOptions& opts, const TGeneralOptions& gopts = TGeneralOptions());

	// cumulative host-clock milliseconds per part of performStereoSLAM (the sections the reference instruments with its
	// profiler: performDA EST.cpp:446-451, define_kf :781-787, plus the front-end / odometry and the place-recognition query)
	struct TTiming { double vo_ms = 0, query_ms = 0, da_ms = 0, insert_ms = 0; size_t frames = 0, kf_frames = 0; } timing;
	TSRBAStereoSLAMOptions srba_options;                   // EST.h:48
	TGeneralOptions general_options;                       // the reference's process-wide global (MAIN.cpp:26)
	mySRBA rba;                                            // EST.h:54
	t_vector_kf m_keyframes;                               // EST.h

	// One stereo pair that the caller has decided is a keyframe: front-end with the adaptive FAST / ORB thresholds
	// (EST.cpp:263-315), first-keyframe path (EST.cpp:82-182) or data association against `similar_kfs`
	// (indices into m_keyframes; empty = the previous keyframe), ID propagation (EST.cpp:572-617), observation list
	// (EST.cpp:667-704) and rba.define_new_keyframe (EST.cpp:782-784).  incr_pose: pose of this keyframe seen from the
	// previous one (m_incr_pose_from_last_kf, used when use_initial_pose).
	// Throws std::runtime_error when fewer than 8 stereo matches remain at the threshold limits (EST.cpp:288-295).
	TNewKeyFrameInfo processKeyframe(const StereoImages& stImgs, const std::vector<size_t>& similar_kfs = std::vector<size_t>(),
	                                 const Pose3D& incr_pose = Pose3D());

	// One iteration of the frame loop, EST.cpp:44-937 (first frame: :82-182).  Needs a vocabulary in m_bow_manager.
	TFrameReport performStereoSLAM(const StereoImages& stImgs);

	// After the loop, EST.cpp:945-987: global pose graph of the relative map, optimised on the device (ssk_posegraph_optimize);
	// returns the keyframe poses in the frame of keyframe 0 and, when `path` is not empty, writes them as the reference's
	// out_kf_poses.txt ("kf x y z yaw pitch roll", %.3f, EST.cpp:976-984).
	std::vector<Pose3D> optimizeGlobalPoseGraph(ssk_pg_result* info = nullptr, const std::string& path = std::string());

	CBoWManager m_bow_manager;                             // EST.h (private there)
	Pose3D m_current_pose, m_last_kf_pose, m_incr_pose_from_last_kf, m_incr_pose_from_last_check;   // EST.h:150-156
	double m_max_translation, m_max_rotation, m_max_translation_limit, m_max_rotation_limit;        // EST.cpp:1163-1164

	// EST.cpp:1737-1878
	bool m_get_similar_kfs(const size_t newKfId, const QueryResults& dbQueryResults, TLoopClosureInfo& out);
	// EST.cpp:1341-1411 with the initial poses of the similar keyframes
	void m_data_association(const CStereoSLAMKF& kf, const TLoopClosureInfo& similar_kfs_info, TVectorKfsDaInfo& out_da);

	// EST.cpp:1341-1411 / 1413-1727: Hamming + direction + ORB-distance filters, then the fundamental-matrix filter
	// (EST.cpp:2015-2055) and the change-in-pose filter (EST.cpp:2113-2177)
	void m_data_association_pairs(const CStereoSLAMKF& kf, const std::vector<size_t>& others, const std::vector<Pose3D>* poses, TVectorKfsDaInfo& out_da);
	void m_data_association(const CStereoSLAMKF& kf, const std::vector<size_t>& similar_kfs, TVectorKfsDaInfo& out_da);
	void m_internal_data_association(const CStereoSLAMKF& this_kf, const CStereoSLAMKF& other_kf, t_kf_da_info& out_da,
	                                 const Pose3D* other_seen_from_this = nullptr, ssk_cip_result* out_cip = nullptr);

	// the keyframe section of the reference's state file (EST.cpp:2678-2700 / 2311-2346): number of keyframes, then per
	// keyframe id, camera pose, left features, right features, matches -- byte layout of ssk_kf_io.h
	bool saveKeyframes(const std::string& path) const;
	bool loadKeyframes(const std::string& path);

	int currentFASTThreshold() const { return m_fast_th; }
	int currentORBThreshold() const { return m_orb_th; }
	size_t lastMatchID() const { return m_last_match_ID; }

private:
	Context& m_ctx;
	size_t m_last_kf_ID = 0, m_last_match_ID = 0;
	int m_fast_th, m_orb_th;
	static constexpr int FAST_TH_MIN = 5, ORB_TH_MAX = 100;   // stereo-vo's isFASTThMin / isORBThMax limits (not in the tree)
	void m_frontend(const StereoImages& stImgs, CStereoSLAMKF& kf);
	// EST.cpp:560-935 once a candidate is accepted: ids, observation list, define_new_keyframe
	void m_insert_keyframe(CStereoSLAMKF& new_kf, const TVectorKfsDaInfo& daInfo, const std::vector<size_t>& sortedIndices,
	                       const Pose3D& incr_pose, TNewKeyFrameInfo& newKFInfo);
	// stand-in for m_voEngine (stereo-vo): current / previous frame, per-match track ids
	TOdometryResult m_vo_process(const StereoImages& stImgs);
	void m_vo_reset_ids();
	CStereoSLAMKF m_vo_cur, m_vo_prev;
	std::vector<size_t> m_vo_ids_cur, m_vo_ids_prev;
	size_t m_vo_next_id = 0, m_vo_kf_max_id = 0, m_frame_count = 0;
	bool m_vo_has_prev = false;
	void m_build_observations(const CStereoSLAMKF& kf, new_kf_observations_t& listObs);
};

}  // namespace ssk
