// facade_demo.cpp -- drives the C++ facade the way the reference's frame loop drives its estimator for the
// frames it has chosen as keyframes (EST.cpp:44-937 reduced to the hot path): read raw 8-bit stereo pairs,
// processKeyframe() each, print one line per keyframe.  Used by tests/test_facade_gpu.py (configs[0] plumbing
// on synthetic KITTI-shaped pairs) and as the example for INTEGRATION.md.
//   usage: facade_demo <dir> <n_pairs> <width> <height> [debug_out_dir]
//   files: <dir>/left_%04d.raw, <dir>/right_%04d.raw   (width*height bytes each)
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "CSRBAStereoSLAMEstimator.h"

static bool read_raw(const std::string& path, std::vector<uint8_t>& buf, size_t n)
{
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) return false;
	buf.resize(n);
	const size_t got = std::fread(buf.data(), 1, n, f);
	std::fclose(f);
	return got == n;
}

int main(int argc, char** argv)
{
	if (argc < 5) { std::fprintf(stderr, "usage: %s <dir> <n_pairs> <width> <height> [debug_out_dir]\n", argv[0]); return 2; }
	const std::string dir = argv[1];
	const int n = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]);
	try {
		ssk::Context ctx(0);
		ssk::TSRBAStereoSLAMOptions opts = ssk::TSRBAStereoSLAMOptions::kitti();
		opts.stereo_camera.ncols = w; opts.stereo_camera.nrows = h;
		ssk::TGeneralOptions g;
		if (argc > 5) { g.debug = true; g.out_dir = argv[5]; }
		ssk::CSRBAStereoSLAMEstimator est(ctx, opts, g);
		std::vector<uint8_t> left, right;
		for (int k = 0; k < n; ++k) {
			char a[64], b[64];
			std::snprintf(a, sizeof a, "/left_%04d.raw", k);
			std::snprintf(b, sizeof b, "/right_%04d.raw", k);
			if (!read_raw(dir + a, left, (size_t)w * h) || !read_raw(dir + b, right, (size_t)w * h)) {
				std::fprintf(stderr, "cannot read pair %d\n", k);
				return 3;
			}
			ssk::StereoImages im;
			im.left = left.data(); im.right = right.data(); im.width = w; im.height = h; im.stride = w;
			const ssk::TNewKeyFrameInfo info = est.processKeyframe(im);
			const ssk::CStereoSLAMKF& kf = est.m_keyframes.back();
			ssk::Pose3D rel;
			const bool have = est.rba.get_kf_relative_pose(0, info.kf_id, 1000, rel);   // pose of this KF seen from KF 0
			size_t shared = 0;
			if (k > 0) {
				const auto& prev = est.m_keyframes[est.m_keyframes.size() - 2].m_matches_ID;
				for (size_t id : kf.m_matches_ID) shared += std::count(prev.begin(), prev.end(), id) ? 1 : 0;
			}
			std::printf("KF %zu nl %zu nr %zu matches %zu shared_ids %zu edges %zu stg1_rmse %.6f err %.6f -> %.6f obs %zu iters %zu pose %d %.6f %.6f %.6f fast_th %d\n",
			            info.kf_id, kf.m_keypoints_left.size(), kf.m_keypoints_right.size(), kf.m_matches.size(), shared,
			            info.created_edge_ids.size(), info.optimize_results_stg1.obs_rmse, info.optimize_results.total_sqr_error_init,
			            info.optimize_results.total_sqr_error_final, info.optimize_results.num_observations, info.optimize_results.num_iters,
			            (int)have, rel.x(), rel.y(), rel.z(), est.currentFASTThreshold());
		}
		std::printf("landmarks %zu edges %zu last_match_id %zu\n", est.rba.get_rba_state().unknown_lms.size(),
		            est.rba.get_rba_state().k2k_edges.size(), est.lastMatchID());
		if (g.debug) {                                             // keyframe section of the reference's state file
			const std::string f = g.out_dir + "/keyframes.kfrec";
			ssk::CSRBAStereoSLAMEstimator back(ctx, opts, g);
			if (!est.saveKeyframes(f) || !back.loadKeyframes(f) || back.m_keyframes.size() != est.m_keyframes.size()) {
				std::fprintf(stderr, "keyframe record round trip failed\n");
				return 4;
			}
			std::printf("saved %zu keyframes to %s\n", est.m_keyframes.size(), f.c_str());
		}
	} catch (const std::exception& e) {     // MAIN.cpp:66-75
		std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
		return 1;
	}
	return 0;
}
