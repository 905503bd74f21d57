// ssk_kf_io.h -- the reference's on-disk keyframe record, byte for byte:
//   m_dump_keypoints_to_stream / m_load_keypoints_from_stream   (EST.cpp:2517-2560 / 2440-2484)
//   m_dump_matches_to_stream   / m_load_matches_from_stream     (EST.cpp:2591-2616 / 2486-2512)
//   the per-keyframe section of the state file: kf id, camera pose as (coords, rotvec), left features, right
//   features, matches   (save EST.cpp:2682-2700, load :2321-2346; DUMP/LOAD_ROTVEC_*_STREAM common.h:58-72)
// so that keyframes written by this build can be read by the reference (and vice versa) and third parties with the
// real dependencies can diff.  Host-only C++, no GPU work; size_t is the 8-byte LP64 size_t of the reference build.
// cv::Mat::type() of the descriptors is CV_8U = 0.
#pragma once
#include "CStereoSLAMKF.h"
#include <cmath>
#include <istream>
#include <ostream>

namespace ssk {
namespace kfio {

template <typename T> inline void put(std::ostream& s, const T& v) { s.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> inline bool get(std::istream& s, T& v) { s.read(reinterpret_cast<char*>(&v), sizeof(T)); return (bool)s; }

// EST.cpp:2517-2560
inline bool dump_keypoints(std::ostream& s, const TKeyPointList& kp, const DescMat& desc)
{
	if (!s) return false;
	put<uint64_t>(s, (uint64_t)kp.size());
	for (size_t f = 0; f < kp.size(); ++f) {
		put(s, kp[f].x); put(s, kp[f].y); put(s, kp[f].response); put(s, kp[f].size); put(s, kp[f].angle);
		put(s, kp[f].octave); put(s, kp[f].class_id);
	}
	// descriptors.rows / .cols / .type() of the cv::Mat (EST.cpp:2470-2474): an empty Mat has cols = 0
	put<int32_t>(s, desc.rows); put<int32_t>(s, desc.rows > 0 ? 32 : 0); put<int32_t>(s, 0 /* CV_8U */);
	if (!desc.data.empty()) s.write(reinterpret_cast<const char*>(desc.data.data()), (std::streamsize)desc.data.size());
	return (bool)s;
}

// EST.cpp:2440-2484
inline bool load_keypoints(std::istream& s, TKeyPointList& kp, DescMat& desc)
{
	uint64_t n = 0;
	if (!get(s, n) || n > (1u << 24)) return false;
	kp.resize((size_t)n);
	for (size_t f = 0; f < kp.size(); ++f)
		if (!(get(s, kp[f].x) && get(s, kp[f].y) && get(s, kp[f].response) && get(s, kp[f].size) && get(s, kp[f].angle) &&
		      get(s, kp[f].octave) && get(s, kp[f].class_id))) return false;
	int32_t rows = 0, cols = 0, type = 0;
	// a side with no keypoints was written from an empty cv::Mat: rows = cols = 0 (and whatever type() an empty Mat reports)
	if (!(get(s, rows) && get(s, cols) && get(s, type)) || rows < 0) return false;
	if (rows > 0 ? (cols != 32 || type != 0) : (cols != 0 && cols != 32)) return false;
	desc.create(rows);
	if (rows) s.read(reinterpret_cast<char*>(desc.data.data()), (std::streamsize)desc.data.size());
	return (bool)s;
}

// EST.cpp:2591-2616  (order on disk: id, queryIdx, trainIdx, distance, imgIdx)
inline bool dump_matches(std::ostream& s, const TDMatchList& m, const std::vector<size_t>& ids)
{
	if (!s || ids.size() != m.size()) return false;
	put<uint64_t>(s, (uint64_t)m.size());
	for (size_t k = 0; k < m.size(); ++k) {
		put<uint64_t>(s, (uint64_t)ids[k]);
		put(s, m[k].queryIdx); put(s, m[k].trainIdx); put(s, m[k].distance); put(s, m[k].imgIdx);
	}
	return (bool)s;
}

// EST.cpp:2486-2512
inline bool load_matches(std::istream& s, TDMatchList& m, std::vector<size_t>& ids)
{
	uint64_t n = 0;
	if (!get(s, n) || n > (1u << 24)) return false;
	m.resize((size_t)n); ids.resize((size_t)n);
	for (size_t k = 0; k < m.size(); ++k) {
		uint64_t id = 0;
		if (!(get(s, id) && get(s, m[k].queryIdx) && get(s, m[k].trainIdx) && get(s, m[k].distance) && get(s, m[k].imgIdx))) return false;
		ids[k] = (size_t)id;
	}
	return true;
}

// rotation matrix <-> rotation vector (mrpt::poses::CPose3DRotVec keeps m_coords and m_rotvec)
inline void rotvec_from_pose(const Pose3D& p, double coords[3], double rv[3])
{
	coords[0] = p.m[3]; coords[1] = p.m[7]; coords[2] = p.m[11];
	const double tr = p.m[0] + p.m[5] + p.m[10];
	const double c = std::fmin(1.0, std::fmax(-1.0, 0.5 * (tr - 1.0)));
	const double th = std::acos(c);
	const double ax[3] = {p.m[9] - p.m[6], p.m[2] - p.m[8], p.m[4] - p.m[1]};      // 2 sin(th) * axis
	if (th < 1e-9) { rv[0] = 0.5 * ax[0]; rv[1] = 0.5 * ax[1]; rv[2] = 0.5 * ax[2]; return; }
	if (M_PI - th < 1e-6) {                                                        // near pi: axis from the diagonal
		double a[3] = {std::sqrt(std::fmax(0.0, 0.5 * (p.m[0] + 1.0))), std::sqrt(std::fmax(0.0, 0.5 * (p.m[5] + 1.0))), std::sqrt(std::fmax(0.0, 0.5 * (p.m[10] + 1.0)))};
		if (ax[0] < 0) a[0] = -a[0];
		if (ax[1] < 0) a[1] = -a[1];
		if (ax[2] < 0) a[2] = -a[2];
		for (int i = 0; i < 3; ++i) rv[i] = th * a[i];
		return;
	}
	const double k = th / (2.0 * std::sin(th));
	for (int i = 0; i < 3; ++i) rv[i] = k * ax[i];
}

inline Pose3D pose_from_rotvec(const double coords[3], const double rv[3])
{
	const double th2 = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2], th = std::sqrt(th2);
	double a, b;
	if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; } else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; }
	const double K[9] = {0, -rv[2], rv[1], rv[2], 0, -rv[0], -rv[1], rv[0], 0};
	double R[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double k2 = 0;
			for (int q = 0; q < 3; ++q) k2 += K[3 * i + q] * K[3 * q + j];
			R[3 * i + j] = a * K[3 * i + j] + b * k2 + (i == j ? 1.0 : 0.0);
		}
	return Pose3D::fromRt(R, coords);
}

// the per-keyframe section of the state file (EST.cpp:2682-2700)
inline bool dump_keyframe(std::ostream& s, const CStereoSLAMKF& kf)
{
	put<uint64_t>(s, (uint64_t)kf.m_kf_ID);
	double coords[3], rv[3];
	rotvec_from_pose(kf.m_camera_pose, coords, rv);
	for (int i = 0; i < 3; ++i) put(s, coords[i]);                                 // DUMP_ROTVEC_TO_STREAM: coords, then rotvec
	for (int i = 0; i < 3; ++i) put(s, rv[i]);
	return dump_keypoints(s, kf.m_keypoints_left, kf.m_descriptors_left) && dump_keypoints(s, kf.m_keypoints_right, kf.m_descriptors_right) &&
	       dump_matches(s, kf.m_matches, kf.m_matches_ID);
}

// EST.cpp:2321-2346
inline bool load_keyframe(std::istream& s, CStereoSLAMKF& kf)
{
	uint64_t id = 0;
	double coords[3], rv[3];
	if (!get(s, id)) return false;
	for (int i = 0; i < 3; ++i) if (!get(s, coords[i])) return false;
	for (int i = 0; i < 3; ++i) if (!get(s, rv[i])) return false;
	kf.setKFID((size_t)id);
	kf.m_camera_pose = pose_from_rotvec(coords, rv);
	return load_keypoints(s, kf.m_keypoints_left, kf.m_descriptors_left) && load_keypoints(s, kf.m_keypoints_right, kf.m_descriptors_right) &&
	       load_matches(s, kf.m_matches, kf.m_matches_ID);
}

}  // namespace kfio
}  // namespace ssk
