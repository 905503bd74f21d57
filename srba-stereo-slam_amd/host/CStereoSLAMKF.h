// CStereoSLAMKF.h -- keyframe container of the reference (src/CStereoSLAMKF.h:56-106) with the same member
// names, plus the DEFINITION of create() that the reference declares (KF.h:82-86) and never defines: the
// stereo front-end runs on the MI355X through the C-ABI (ssk_frontend_run).
#pragma once
#include "srba_stereo_slam_options.h"

namespace ssk {

class CStereoSLAMKF;
typedef std::vector<CStereoSLAMKF> t_vector_kf;    // KF.h:51

class CStereoSLAMKF {
public:
	CStereoSLAMKF() : m_kf_ID(0) {}                                                   // KF.h:60

	inline void setKFID(const size_t ID) { m_kf_ID = ID; }                            // KF.h:63-65

	// KF.h:75-80
	inline void generateMatchesIDs(const size_t _starting_idx)
	{
		size_t starting_idx = _starting_idx;
		for (size_t m = 0; m < m_matches.size(); ++m) m_matches_ID.push_back(starting_idx++);
	}

	// KF.h:82-86: "detect ORB features in left and right image, match the features and create a new KF with
	// the pairings".  fast_threshold / orb_threshold: the current adaptive values (EST.cpp:271-315 adapts them
	// around this call); < 0 = the options' own detect_fast_th / orb_max_distance.
	void create(Context& ctx, const StereoImages& stImgs, const TSRBAStereoSLAMOptions& options = TSRBAStereoSLAMOptions(),
	            int fast_threshold = -1, int orb_threshold = -1);

	void dumpToConsole() const;                                                       // KF.cpp:30-55
	// KF.cpp:60-110: info_kf%04d.txt ("kf_id match_id xl yl xr yr dist") and info_feats_kf%04d.txt
	void saveInfoToFiles(const TGeneralOptions& general_options, const std::string& str_modif = std::string()) const;

	ssk_kf_view view() const;                                                         // for the data-association calls

	// DATA MEMBERS (KF.h:99-104)
	TKeyPointList m_keypoints_left, m_keypoints_right;
	DescMat m_descriptors_left, m_descriptors_right;
	TDMatchList m_matches;
	std::vector<size_t> m_matches_ID;
	Pose3D m_camera_pose;
	size_t m_kf_ID;
};

}  // namespace ssk
