/*
 * oracle_orb.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h):
 * stereo front-end = ORB detect/describe on both images + row-band L<->R Hamming match.
 *
 * In the reference this work is done by stereo-vo (rso::CStereoOdometryEstimator::processNewImagePair,
 * called at EST.cpp:112,267) through OpenCV 2.4's cv::ORB; neither library is in the tree or the image.
 * What follows restates the PUBLISHED algorithms (FAST: Rosten & Drummond 2006; ORB: Rublee et al. 2011;
 * OpenCV 2.4 modules/features2d/src/{fast.cpp,orb.cpp} as the public description of cv::ORB's steps) and
 * fixes every choice they leave open.  The fixed choices ("front-end contract", also in DESIGN.md):
 *   C1  one pyramid level (reference forces nlevels = 1: EST.cpp:1141, UTIL.h:288, KITTI.ini:5).
 *   C2  FAST-9/16, score = (largest t at which the pixel is still a corner) = s-1 with
 *       s = max over the 16 arcs of 9 contiguous ring pixels of min(|diff| of one sign); corner iff s > th.
 *   C3  3x3 non-max suppression on the score map, strict '>' against all 8 neighbours (non-corners = 0).
 *   C4  border: only 31 <= x < W-31, 31 <= y < H-31 (cv::ORB edgeThreshold 31, runByImageBorder).
 *   C5  keep the best 2*n_feats by FAST score, then Harris (7x7 block, Sobel-like 3x3 gradients, k),
 *       optional (2r+1)^2 window suppression on the Harris response, then keep the best n_feats.
 *       EVERY ordering is the total order (response desc, y asc, x asc) and truncation is exact
 *       (OpenCV's nth_element order / keep-all-ties is implementation defined; the total order is not).
 *   C6  orientation: intensity centroid over the radius-15 disc (u_max table), integer moments;
 *       KeyPoint.angle = the degree polynomial of OpenCV 2.4 fastAtan2 in IEEE float, no contraction.
 *   C7  steering (default, mode 1): as cv::ORB -- the fastAtan2 degree angle times (float)(CV_PI/180.f), then (float)cos / (float)sin of
 *       that float, evaluated by the shared include/ssk_sincos.h (equal to the host libm's double cos / sin rounded to float for every
 *       float angle in [0, 360]: tools/sincos_check.c).  Mode 0 (rounds 1-2): (cos,sin) = (m10,m01)/sqrt(m10^2+m01^2) in double
 *       -- the exact direction of the centroid -- not cos/sin of the polynomial angle, so the result
 *       needs only correctly-rounded IEEE sqrt/div (bit-reproducible on CPU and GPU).
 *   C8  7x7 sigma=2 Gaussian blur in 8.8 fixed point, taps {18,34,49,55,49,34,18}, result
 *       min(255,(sum + 2^15) >> 16); samples never reach the image border (C4), so no border rule.
 *   C9  BRIEF-256: bit i of byte j is test 8j+i: blur(p0) < blur(p1), sample offsets
 *       (rint(x*a - y*b), rint(x*b + y*a)) in float, round-half-even.  Pattern is an input table.
 *   C10 L<->R: for left i the right j minimising Hamming distance among |yL-yR| <= max_y_diff and
 *       min_disparity <= xL-xR <= max_disparity, lowest j on ties, accepted iff distance <= orb_max_distance;
 *       matches are emitted in left-keypoint order; 1-to-1 is not enforced (KITTI.ini:18).
 */
#include "ssk_oracle.h"
#include "../include/ssk_sincos.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EDGE 31
#define HALF_PATCH 15

static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
static const int UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
static const int GAUSS7[7] = {18, 34, 49, 55, 49, 34, 18};

/* C2: s for one pixel (0 <= result <= 255). */
static int fast_s(const uint8_t* p, int stride)
{
	int d[25];
	const int c = p[0];
	for (int i = 0; i < 16; ++i) d[i] = (int)p[RING_DY[i] * stride + RING_DX[i]] - c;
	for (int i = 16; i < 25; ++i) d[i] = d[i - 16];
	int best = 0;
	for (int i = 0; i < 16; ++i) {
		int mn = d[i], mx = d[i];
		for (int k = 1; k < 9; ++k) {
			if (d[i + k] < mn) mn = d[i + k];
			if (d[i + k] > mx) mx = d[i + k];
		}
		if (mn > best) best = mn;        /* all 9 brighter by at least mn */
		if (-mx > best) best = -mx;      /* all 9 darker by at least -mx */
	}
	return best;
}

/* FAST score map over the whole image (0 where not a corner or within 3 px of the border). */
void ssko_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score)
{
	memset(score, 0, (size_t)w * h);
	for (int y = 3; y < h - 3; ++y) {
		const uint8_t* row = img + (size_t)y * stride;
		for (int x = 3; x < w - 3; ++x) {
			const uint8_t* p = row + x;
			/* necessary condition (any 9-arc contains two adjacent compass points): cheap reject */
			const int c = p[0], hi = c + threshold, lo = c - threshold;
			const int a0 = p[3 * stride], a4 = p[3], a8 = p[-3 * stride], a12 = p[-3];
			const int nb = (a0 > hi) + (a4 > hi) + (a8 > hi) + (a12 > hi);
			const int nd = (a0 < lo) + (a4 < lo) + (a8 < lo) + (a12 < lo);
			if (nb < 2 && nd < 2) continue;
			const int s = fast_s(p, stride);
			if (s > threshold) score[(size_t)y * w + x] = (uint8_t)(s - 1);
		}
	}
}

static int cmp_u32_desc(const void* a, const void* b)
{
	const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	return x < y ? 1 : (x > y ? -1 : 0);
}
static int cmp_u64_desc(const void* a, const void* b)
{
	const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
	return x < y ? 1 : (x > y ? -1 : 0);
}

/* C3+C4: candidates as packed keys  score<<24 | (4095-y)<<12 | (4095-x), sorted descending
 * (= score desc, y asc, x asc).  Returns the count; writes at most cap keys. */
int ssko_fast_candidates(const uint8_t* img, int w, int h, int stride, int threshold, uint32_t* keys, int cap)
{
	if (w > 4096 || h > 4096 || w < 2 * EDGE + 1 || h < 2 * EDGE + 1 || threshold < 1 || threshold > 254) return 0;
	uint8_t* sc = (uint8_t*)malloc((size_t)w * h);
	ssko_fast_score_map(img, w, h, stride, threshold, sc);
	int n = 0;
	uint32_t* all = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)((w / 2 + 1) * (h / 2 + 1)));
	for (int y = EDGE; y < h - EDGE; ++y)
		for (int x = EDGE; x < w - EDGE; ++x) {
			const uint8_t* s = sc + (size_t)y * w + x;
			const int v = s[0];
			if (v == 0) continue;                       /* not a corner: corners score s-1 >= threshold >= 1 */
			if (v > s[-1] && v > s[1] && v > s[-w - 1] && v > s[-w] && v > s[-w + 1] && v > s[w - 1] && v > s[w] && v > s[w + 1])
				all[n++] = ((uint32_t)v << 24) | ((uint32_t)(4095 - y) << 12) | (uint32_t)(4095 - x);
		}
	qsort(all, (size_t)n, sizeof(uint32_t), cmp_u32_desc);
	memcpy(keys, all, sizeof(uint32_t) * (size_t)(n < cap ? n : cap));
	free(all);
	free(sc);
	return n;
}

/* C5: Harris response, [published: OpenCV 2.4 orb.cpp HarrisResponses], float ops in this exact order. */
float ssko_harris_response(const uint8_t* img, int stride, int x, int y, float harris_k)
{
	int a = 0, b = 0, c = 0;
	for (int dy = -3; dy <= 3; ++dy)
		for (int dx = -3; dx <= 3; ++dx) {
			const uint8_t* p = img + (size_t)(y + dy) * stride + (x + dx);
			const int Ix = (p[1] - p[-1]) * 2 + (p[-stride + 1] - p[-stride - 1]) + (p[stride + 1] - p[stride - 1]);
			const int Iy = (p[stride] - p[-stride]) * 2 + (p[stride - 1] - p[-stride - 1]) + (p[stride + 1] - p[-stride + 1]);
			a += Ix * Ix;
			b += Iy * Iy;
			c += Ix * Iy;
		}
	const float scale = 1.0f / (4.0f * 7.0f * 255.0f);
	const float s3 = (scale * scale) * scale;          /* cv::ORB HarrisResponses: scale * scale * scale * scale, left to right */
	const float s4 = s3 * scale;
	const float fa = (float)a, fb = (float)b, fc = (float)c;
	const float t1 = fa * fb;
	const float t2 = fc * fc;
	const float t3 = fa + fb;
	const float t4 = harris_k * t3;
	const float t5 = t4 * t3;
	return ((t1 - t2) - t5) * s4;
}

/* monotone float -> uint32 (larger float = larger key); -0 and +0 map to neighbouring keys, NaN unused. */
static uint32_t float_key(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* C6: degrees in [0,360), [published: OpenCV 2.4 fastAtan2 polynomial], IEEE float, no contraction. */
float ssko_atan2_deg(float y, float x)
{
	const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
	const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
	const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
	const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
	const float eps = (float)2.2204460492503131e-16;
	const float ax = fabsf(x), ay = fabsf(y);
	float a, c, c2;
	if (ax >= ay) {
		c = ay / (ax + eps);
		c2 = c * c;
		a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	} else {
		c = ax / (ay + eps);
		c2 = c * c;
		a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
	}
	if (x < 0) a = 180.f - a;
	if (y < 0) a = 360.f - a;
	return a;
}

/* C8: blurred pixel at (x,y). */
static int blur_at(const uint8_t* img, int stride, int x, int y)
{
	int sum = 0;
	for (int dy = -3; dy <= 3; ++dy) {
		const uint8_t* p = img + (size_t)(y + dy) * stride + x;
		int r = 0;
		for (int dx = -3; dx <= 3; ++dx) r += GAUSS7[dx + 3] * p[dx];
		sum += GAUSS7[dy + 3] * r;
	}
	const int v = (sum + (1 << 15)) >> 16;
	return v > 255 ? 255 : v;
}

static int g_orb_steering = 1;          /* C7: 1 = cv::ORB's (float)cos / (float)sin of the fastAtan2 angle, 0 = exact centroid direction */
void ssko_set_orb_steering(int mode) { g_orb_steering = mode ? 1 : 0; }

/* C6/C7/C9 for one keypoint: angle (deg), descriptor (32 B). */
void ssko_orient_describe(const uint8_t* img, int stride, int x, int y, const int8_t* pattern,
                          float* angle_deg, uint8_t* desc)
{
	long m01 = 0, m10 = 0;
	for (int v = -HALF_PATCH; v <= HALF_PATCH; ++v) {
		const int um = UMAX[v < 0 ? -v : v];
		const uint8_t* p = img + (size_t)(y + v) * stride + x;
		int rs = 0;
		for (int u = -um; u <= um; ++u) { const int I = p[u]; m10 += (long)u * I; rs += I; }
		m01 += (long)v * rs;
	}
	*angle_deg = ssko_atan2_deg((float)m01, (float)m10);
	float ca = 1.f, sa = 0.f;
	if (g_orb_steering) ssk_orb_sincos_deg(*angle_deg, &ca, &sa);
	else if (m01 != 0 || m10 != 0) {
		const double dx = (double)m10, dy = (double)m01;
		const double r = sqrt(dx * dx + dy * dy);
		ca = (float)(dx / r);
		sa = (float)(dy / r);
	}
	for (int j = 0; j < 32; ++j) {
		unsigned val = 0;
		for (int i = 0; i < 8; ++i) {
			const int8_t* t = pattern + 4 * (8 * j + i);
			const float x0 = (float)t[0], y0 = (float)t[1], x1 = (float)t[2], y1 = (float)t[3];
			const int ix0 = (int)rintf(x0 * ca - y0 * sa), iy0 = (int)rintf(x0 * sa + y0 * ca);
			const int ix1 = (int)rintf(x1 * ca - y1 * sa), iy1 = (int)rintf(x1 * sa + y1 * ca);
			const int v0 = blur_at(img, stride, x + ix0, y + iy0);
			const int v1 = blur_at(img, stride, x + ix1, y + iy1);
			val |= (unsigned)(v0 < v1) << i;
		}
		desc[j] = (uint8_t)val;
	}
}

/* Whole detector for one image (C1-C9).  Returns the number of keypoints (<= n_feats <= cap). */
int ssko_orb_detect(const uint8_t* img, int w, int h, int stride, const ssk_frontend_params* prm,
                    const int8_t* pattern, ssk_keypoint* kp, uint8_t* desc, int cap)
{
	const int N = prm->n_feats;
	if (N <= 0 || N > cap || prm->n_levels != 1) return -1;
	const int K2 = 2 * N;
	const int maxc = (w / 2 + 1) * (h / 2 + 1);
	uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)maxc);
	int nc = ssko_fast_candidates(img, w, h, stride, prm->fast_threshold, keys, maxc);
	if (nc > K2) nc = K2;                                   /* retainBest(2N) on the FAST score */
	uint64_t* hk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(nc > 0 ? nc : 1));
	int* cx = (int*)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
	int* cy = (int*)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
	for (int i = 0; i < nc; ++i) {
		const int x = 4095 - (int)(keys[i] & 0xFFF), y = 4095 - (int)((keys[i] >> 12) & 0xFFF);
		cx[i] = x; cy[i] = y;
		const float r = ssko_harris_response(img, stride, x, y, prm->harris_k);
		hk[i] = ((uint64_t)float_key(r) << 32) | (uint64_t)(keys[i] & 0xFFFFFFu);
	}
	const int rad = prm->nms_radius;
	int ns = 0;
	uint64_t* sk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(nc > 0 ? nc : 1));
	for (int i = 0; i < nc; ++i) {
		int keep = 1;
		if (rad > 0)
			for (int j = 0; j < nc && keep; ++j)
				if (j != i && abs(cx[j] - cx[i]) <= rad && abs(cy[j] - cy[i]) <= rad && hk[j] > hk[i]) keep = 0;
		if (keep) sk[ns++] = hk[i];
	}
	qsort(sk, (size_t)ns, sizeof(uint64_t), cmp_u64_desc);
	if (ns > N) ns = N;                                     /* retainBest(N) on the Harris response */
	for (int i = 0; i < ns; ++i) {
		const uint32_t lo = (uint32_t)(sk[i] & 0xFFFFFFu);
		const int x = 4095 - (int)(lo & 0xFFF), y = 4095 - (int)((lo >> 12) & 0xFFF);
		kp[i].x = (float)x; kp[i].y = (float)y; kp[i].size = 31.f;
		kp[i].response = ssko_harris_response(img, stride, x, y, prm->harris_k);
		kp[i].octave = 0; kp[i].class_id = -1;
		ssko_orient_describe(img, stride, x, y, pattern, &kp[i].angle, desc + 32 * (size_t)i);
	}
	free(keys); free(hk); free(cx); free(cy); free(sk);
	return ns;
}

static int hamming256o(const uint8_t* a, const uint8_t* b)
{
	int d = 0;
	for (int w = 0; w < 4; ++w) {
		uint64_t x, y;
		memcpy(&x, a + 8 * w, 8);
		memcpy(&y, b + 8 * w, 8);
		d += __builtin_popcountll(x ^ y);
	}
	return d;
}

/* C10. returns the number of matches written (<= nl). */
int ssko_stereo_match(const ssk_keypoint* kl, int nl, const uint8_t* dl, const ssk_keypoint* kr, int nr,
                      const uint8_t* dr, const ssk_frontend_params* prm, ssk_dmatch* out)
{
	int n = 0;
	for (int i = 0; i < nl; ++i) {
		int best = INT_MAX, bj = -1;
		for (int j = 0; j < nr; ++j) {
			if (fabsf(kl[i].y - kr[j].y) > prm->max_y_diff) continue;
			const float disp = kl[i].x - kr[j].x;
			if (disp < prm->min_disparity || disp > prm->max_disparity) continue;
			const int d = hamming256o(dl + 32 * (size_t)i, dr + 32 * (size_t)j);
			if (d < best) { best = d; bj = j; }
		}
		if (bj >= 0 && best <= prm->orb_max_distance) {
			out[n].queryIdx = i; out[n].trainIdx = bj; out[n].imgIdx = 0; out[n].distance = (float)best;
			++n;
		}
	}
	return n;
}

/* The whole front-end for one stereo pair (what processNewImagePair + getValues hand to the KF). */
int ssko_frontend(const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                  const ssk_frontend_params* prm, const int8_t* pattern, ssk_kf_out* out)
{
	const int nl = ssko_orb_detect(left, w, h, stride, prm, pattern, out->kp_left, out->desc_left, out->capacity);
	const int nr = ssko_orb_detect(right, w, h, stride, prm, pattern, out->kp_right, out->desc_right, out->capacity);
	if (nl < 0 || nr < 0) return -1;
	out->n_left = nl; out->n_right = nr;
	out->n_matches = ssko_stereo_match(out->kp_left, nl, out->desc_left, out->kp_right, nr, out->desc_right, prm, out->matches);
	return 0;
}
