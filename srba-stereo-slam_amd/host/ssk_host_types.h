// ssk_host_types.h -- the plain value types the C++ facade uses where the reference uses OpenCV / MRPT types.
// They are the facade's own (this build links neither OpenCV nor MRPT); layouts follow the reference's:
//   cv::KeyPoint / cv::DMatch            -> ssk_keypoint / ssk_dmatch (include/ssk_capi.h; UTIL.h:47-48)
//   cv::Mat descriptors (N x 32, CV_8U)  -> ssk::DescMat
//   CObservationStereoImagesPtr          -> ssk::StereoImages (two 8-bit gray images)
//   mrpt::poses::CPose3D                 -> ssk::Pose3D (row-major 3x4 [R|t])
//   mrpt::utils::TStereoCamera           -> ssk::TStereoCamera
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ssk_capi.h"

namespace ssk {

typedef ssk_keypoint KeyPoint;
typedef ssk_dmatch DMatch;
typedef std::vector<KeyPoint> TKeyPointList;   // UTIL.h:47
typedef std::vector<DMatch> TDMatchList;       // UTIL.h:48

struct DescMat {                               // cv::Mat(N, 32, CV_8U)
	int rows = 0;
	std::vector<uint8_t> data;
	void create(int n) { rows = n; data.assign((size_t)n * 32, 0); }
	uint8_t* row(int r) { return data.data() + (size_t)r * 32; }
	const uint8_t* row(int r) const { return data.data() + (size_t)r * 32; }
};

struct StereoImages {                          // what the hot path reads of CObservationStereoImages
	const uint8_t* left = nullptr;
	const uint8_t* right = nullptr;
	int width = 0, height = 0, stride = 0;
};

struct Pose3D {                                // SE(3): p_parent = R p + t
	double m[12];
	Pose3D() { identity(); }
	void identity() { std::memset(m, 0, sizeof m); m[0] = m[5] = m[10] = 1.0; }
	static Pose3D fromRt(const double R[9], const double t[3])
	{
		Pose3D p;
		for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) p.m[4 * i + j] = R[3 * i + j]; p.m[4 * i + 3] = t[i]; }
		return p;
	}
	Pose3D operator+(const Pose3D& b) const     // composition a (+) b
	{
		Pose3D r;
		for (int i = 0; i < 3; ++i) {
			for (int j = 0; j < 3; ++j) r.m[4 * i + j] = m[4 * i] * b.m[j] + m[4 * i + 1] * b.m[4 + j] + m[4 * i + 2] * b.m[8 + j];
			r.m[4 * i + 3] = m[4 * i] * b.m[3] + m[4 * i + 1] * b.m[7] + m[4 * i + 2] * b.m[11] + m[4 * i + 3];
		}
		return r;
	}
	Pose3D inverse() const
	{
		Pose3D r;
		for (int i = 0; i < 3; ++i) {
			for (int j = 0; j < 3; ++j) r.m[4 * i + j] = m[4 * j + i];
			r.m[4 * i + 3] = -(m[i] * m[3] + m[4 + i] * m[7] + m[8 + i] * m[11]);
		}
		return r;
	}
	void composePoint(const double p[3], double out[3]) const
	{
		for (int i = 0; i < 3; ++i) out[i] = m[4 * i] * p[0] + m[4 * i + 1] * p[1] + m[4 * i + 2] * p[2] + m[4 * i + 3];
	}
	double translationNorm() const { return std::sqrt(m[3] * m[3] + m[7] * m[7] + m[11] * m[11]); }    // CPose3DRotVec::m_coords.norm()
	double rotationAngle() const                                                                          // CPose3DRotVec::m_rotvec.norm()
	{
		double c = 0.5 * (m[0] + m[5] + m[10] - 1.0);
		c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
		return std::acos(c);
	}
	double x() const { return m[3]; }
	double y() const { return m[7]; }
	double z() const { return m[11]; }
};

struct TStereoCamera {                         // the fields the hot path reads (UTIL.h:366, KITTI.ini:138-155)
	double fx_l = 707.0912, fy_l = 707.0912, cx_l = 601.8873, cy_l = 183.1104;
	double fx_r = 707.0912, fy_r = 707.0912, cx_r = 601.8873, cy_r = 183.1104;
	double baseline = 0.54;
	int ncols = 1226, nrows = 370;
};

// RAII context + error mapping: a non-zero C-ABI status becomes std::runtime_error, the way the reference
// reports errors through MRPT's THROW_EXCEPTION (e.g. EST.cpp:288-295).
class Context {
public:
	explicit Context(int device = 0, void* hip_stream = nullptr)
	{
		const int rc = ssk_create(&h_, device, hip_stream);
		if (rc != SSK_OK) throw std::runtime_error(std::string("ssk_create: ") + ssk_last_error(nullptr));
	}
	// a view of a context somebody else owns (the C-ABI of the facade, include/ssk_rba.h)
	explicit Context(ssk_ctx* borrowed) : h_(borrowed), owns_(false)
	{
		if (!borrowed) throw std::runtime_error("Context: NULL ssk_ctx");
	}
	~Context() { if (owns_) ssk_destroy(h_); }
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
	ssk_ctx* get() const { return h_; }
	void check(int rc, const char* what) const
	{
		if (rc != SSK_OK) throw std::runtime_error(std::string(what) + ": " + ssk_last_error(h_));
	}
private:
	ssk_ctx* h_ = nullptr;
	bool owns_ = true;
};

}  // namespace ssk
