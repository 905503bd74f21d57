// CSRBAStereoSLAMEstimator.h -- the per-keyframe hot path of the reference's estimator
// (src/CSRBAStereoSLAMEstimator.h:33-204, .cpp) behind the same names: srba_options, rba, m_keyframes,
// m_data_association, m_internal_data_association, the landmark-ID propagation and the observation
// marshalling, with the numerical work on the MI355X through the C-ABI.
//
// NOT here (out of scope for this path): the frame grabber, GUI, BoW place recognition
// (DBoW2), keyframe-insertion policy, loop-closure confirmation, final pose-graph optimisation.  The caller
// decides which frames are keyframes and which earlier keyframes to associate with (default: the previous
// one, as EST.cpp:1758 always includes).
#pragma once
#include "CStereoSLAMKF.h"
#include "srba_stereo_slam.h"

namespace ssk {

struct t_kf_da_info {                                      // UTIL.h:55-66
	size_t kf_idx = (size_t)-1;
	size_t tracked_matches = 0;
	std::vector<std::pair<size_t, double>> tracking_info;  // (other match idx or INVALID_IDX, distance)
};
typedef std::vector<t_kf_da_info> TVectorKfsDaInfo;
static const size_t INVALID_IDX = (size_t)-1;              // COMMON.h:88 (-1)

class CSRBAStereoSLAMEstimator {
public:
	CSRBAStereoSLAMEstimator(Context& ctx, const TSRBAStereoSLAMOptions& opts, const TGeneralOptions& gopts = TGeneralOptions());

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

	// EST.cpp:1341-1411 / 1413-1727: Hamming + direction + ORB-distance filters, then the fundamental-matrix filter
	// (EST.cpp:2015-2055) and the change-in-pose filter (EST.cpp:2113-2177)
	void m_data_association(const CStereoSLAMKF& kf, const std::vector<size_t>& similar_kfs, TVectorKfsDaInfo& out_da);
	void m_internal_data_association(const CStereoSLAMKF& this_kf, const CStereoSLAMKF& other_kf, t_kf_da_info& out_da);

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
	void m_build_observations(const CStereoSLAMKF& kf, new_kf_observations_t& listObs);
};

}  // namespace ssk
