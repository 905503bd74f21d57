// kf_io_tool -- CPU-only driver of ssk_kf_io.h for tests/test_kf_io.py:
//   kf_io_tool roundtrip <in.kfrec> <out.kfrec>    load a keyframe record and dump it again (must be byte identical)
//   kf_io_tool make <out.kfrec> <n_left> <n_right> <n_matches> <seed>   synthesise a keyframe and dump it
#include "ssk_kf_io.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>

using namespace ssk;

int main(int argc, char** argv)
{
	if (argc >= 4 && std::string(argv[1]) == "roundtrip") {
		std::ifstream in(argv[2], std::ios::binary);
		CStereoSLAMKF kf;
		if (!in || !kfio::load_keyframe(in, kf)) { std::fprintf(stderr, "load failed\n"); return 2; }
		std::ofstream out(argv[3], std::ios::binary);
		if (!kfio::dump_keyframe(out, kf)) { std::fprintf(stderr, "dump failed\n"); return 3; }
		std::printf("kf %zu: %zu + %zu keypoints, %zu matches\n", kf.m_kf_ID, kf.m_keypoints_left.size(), kf.m_keypoints_right.size(), kf.m_matches.size());
		return 0;
	}
	if (argc >= 7 && std::string(argv[1]) == "make") {
		const int nl = std::atoi(argv[3]), nr = std::atoi(argv[4]), nm = std::atoi(argv[5]);
		unsigned s = (unsigned)std::atoi(argv[6]);
		auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return s >> 8; };
		CStereoSLAMKF kf;
		kf.setKFID(rnd() % 1000);
		const double coords[3] = {1.5, -0.25, 12.0}, rv[3] = {0.02, -0.4, 0.1};
		kf.m_camera_pose = kfio::pose_from_rotvec(coords, rv);
		auto fill = [&](TKeyPointList& kp, DescMat& d, int n) {
			kp.resize((size_t)n); d.create(n);
			for (int i = 0; i < n; ++i) {
				kp[i].x = (float)(rnd() % 1241); kp[i].y = (float)(rnd() % 376); kp[i].size = 31.f; kp[i].angle = (float)(rnd() % 36000) / 100.f;
				kp[i].response = (float)(rnd() % 100000) * 1e-7f; kp[i].octave = 0; kp[i].class_id = -1;
				for (int b = 0; b < 32; ++b) d.row(i)[b] = (uint8_t)rnd();
			}
		};
		fill(kf.m_keypoints_left, kf.m_descriptors_left, nl);
		fill(kf.m_keypoints_right, kf.m_descriptors_right, nr);
		kf.m_matches.resize((size_t)nm);
		for (int i = 0; i < nm; ++i) { kf.m_matches[i].queryIdx = (int)(rnd() % (nl ? nl : 1)); kf.m_matches[i].trainIdx = (int)(rnd() % (nr ? nr : 1)); kf.m_matches[i].imgIdx = 0; kf.m_matches[i].distance = (float)(rnd() % 61); }
		kf.generateMatchesIDs(1000);
		std::ofstream out(argv[2], std::ios::binary);
		return kfio::dump_keyframe(out, kf) ? 0 : 3;
	}
	std::fprintf(stderr, "usage: kf_io_tool roundtrip <in> <out> | make <out> <nl> <nr> <nm> <seed>\n");
	return 1;
}
