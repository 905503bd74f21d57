// CStereoSLAMKF.cpp -- see CStereoSLAMKF.h.  Dump formats follow src/CStereoSLAMKF.cpp:30-110 (file names use
// '/' where the reference hard-codes a Windows '\\').
#include "CStereoSLAMKF.h"
#include <cstdio>
#include <iostream>

namespace ssk {

void CStereoSLAMKF::create(Context& ctx, const StereoImages& stImgs, const TSRBAStereoSLAMOptions& options,
                           int fast_threshold, int orb_threshold)
{
	const ssk_frontend_params prm = options.frontend(fast_threshold < 0 ? options.detect_fast_th : fast_threshold,
	                                                  orb_threshold < 0 ? options.orb_max_distance : orb_threshold);
	const int cap = (int)options.n_feats;
	m_keypoints_left.assign(cap, KeyPoint());
	m_keypoints_right.assign(cap, KeyPoint());
	m_descriptors_left.create(cap);
	m_descriptors_right.create(cap);
	m_matches.assign(cap, DMatch());
	ssk_kf_out out;
	out.capacity = cap;
	out.kp_left = m_keypoints_left.data(); out.kp_right = m_keypoints_right.data();
	out.desc_left = m_descriptors_left.data.data(); out.desc_right = m_descriptors_right.data.data();
	out.matches = m_matches.data();
	out.n_left = out.n_right = out.n_matches = 0;
	ctx.check(ssk_frontend_run(ctx.get(), stImgs.left, stImgs.right, stImgs.width, stImgs.height, stImgs.stride, &prm, &out),
	          "CStereoSLAMKF::create");
	m_keypoints_left.resize(out.n_left);
	m_keypoints_right.resize(out.n_right);
	m_descriptors_left.rows = out.n_left; m_descriptors_left.data.resize((size_t)out.n_left * 32);
	m_descriptors_right.rows = out.n_right; m_descriptors_right.data.resize((size_t)out.n_right * 32);
	m_matches.resize(out.n_matches);
	m_matches_ID.clear();
}

ssk_kf_view CStereoSLAMKF::view() const
{
	ssk_kf_view v;
	v.kp_left = m_keypoints_left.data(); v.n_left = (int32_t)m_keypoints_left.size();
	v.kp_right = m_keypoints_right.data(); v.n_right = (int32_t)m_keypoints_right.size();
	v.desc_left = m_descriptors_left.data.data(); v.desc_right = m_descriptors_right.data.data();
	v.matches = m_matches.data(); v.n_matches = (int32_t)m_matches.size();
	return v;
}

void CStereoSLAMKF::dumpToConsole() const
{
	std::cout << "KEYFRAME [" << m_kf_ID << "]" << std::endl
	          << "---------------------------------------------" << std::endl
	          << "   :: Camera pose = (" << m_camera_pose.x() << "," << m_camera_pose.y() << "," << m_camera_pose.z() << ")" << std::endl
	          << "   :: Matches [" << m_matches.size() << " out of " << m_keypoints_left.size() << "/" << m_keypoints_right.size()
	          << "]: ID: left_kp_x,left_kp_y --> right_kp_x,right_kp_y" << std::endl
	          << "   -------------------------------------" << std::endl;
	for (size_t k = 0; k < m_matches.size(); ++k) {
		const size_t id1 = m_matches[k].queryIdx, id2 = m_matches[k].trainIdx;
		std::cout << "   " << (k < m_matches_ID.size() ? m_matches_ID[k] : 0) << ": " << m_keypoints_left[id1].x << "," << m_keypoints_left[id1].y
		          << " --> " << m_keypoints_right[id2].x << "," << m_keypoints_right[id2].y << std::endl;
	}
	std::cout << "++++++++++++++++++++++++++++++++++++++++++++++" << std::endl;
}

void CStereoSLAMKF::saveInfoToFiles(const TGeneralOptions& general_options, const std::string& str_modif) const
{
	char name[1024];
	const std::string pre = general_options.out_dir + "/" + (str_modif.empty() ? std::string() : str_modif + "_");
	std::snprintf(name, sizeof name, "%sinfo_kf%04d.txt", pre.c_str(), (int)m_kf_ID);
	FILE* f = std::fopen(name, "wt");
	if (!f) throw std::runtime_error(std::string("Output file ") + name + " could not be opened");            // KF.cpp:75
	std::fprintf(f, "%% [KF_ID] [MATCH_ID] [LEFT_PT{x y}] [RIGHT_PT{x y}] [MATCH_DISTANCE]\n");
	for (size_t m = 0; m < m_matches.size(); ++m) {
		const KeyPoint& l = m_keypoints_left[m_matches[m].queryIdx];
		const KeyPoint& r = m_keypoints_right[m_matches[m].trainIdx];
		std::fprintf(f, "%d %d %.2f %.2f %.2f %.2f %.2f\n", (int)m_kf_ID, m < m_matches_ID.size() ? (int)m_matches_ID[m] : 0,
		             l.x, l.y, r.x, r.y, m_matches[m].distance);
	}
	std::fclose(f);
	std::snprintf(name, sizeof name, "%sinfo_feats_kf%04d.txt", pre.c_str(), (int)m_kf_ID);
	f = std::fopen(name, "wt");
	if (!f) throw std::runtime_error(std::string("Output file ") + name + " could not be opened");
	std::fprintf(f, "%d %d\n", (int)m_keypoints_left.size(), (int)m_keypoints_right.size());
	for (const KeyPoint& k : m_keypoints_left) std::fprintf(f, "%.2f %.2f\n", k.x, k.y);
	for (const KeyPoint& k : m_keypoints_right) std::fprintf(f, "%.2f %.2f\n", k.x, k.y);
	std::fclose(f);
}

}  // namespace ssk
