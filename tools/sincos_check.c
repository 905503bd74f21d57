/* sincos_check.c -- ssk_orb_sincos_deg (include/ssk_sincos.h) against the host libm for EVERY float angle in [0, 360] (1.1e9 values):
 * counts the angles whose (float)cos((double)rad) or (float)sin((double)rad) differs, rad = angle * (float)(CV_PI/180.f).
 * build: gcc -O2 -ffp-contract=off -fopenmp tools/sincos_check.c -o /tmp/sincos_check -lm ; run: /tmp/sincos_check [stride] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/ssk_sincos.h"

int main(int argc, char** argv)
{
	const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
	const float top = 360.0f;
	uint32_t last;
	memcpy(&last, &top, 4);
	long long n = 0, bad_c = 0, bad_s = 0, bad_f = 0;
	float worst = 0;
#pragma omp parallel for reduction(+ : n, bad_c, bad_s, bad_f) schedule(static, 1 << 20)
	for (uint32_t b = 0; b <= last; b += stride) {
		float a;
		memcpy(&a, &b, 4);
		float c, s;
		ssk_orb_sincos_deg(a, &c, &s);
		const float rad = a * 0.01745329238474369f;
		const float cl = (float)cos((double)rad), sl = (float)sin((double)rad);
		++n;
		if (c != cl) ++bad_c;
		if (s != sl) ++bad_s;
		if (c != cosf(rad) || s != sinf(rad)) ++bad_f;
	}
	(void)worst;
	printf("%lld angles: cos differs from (float)cos((double)x) at %lld, sin at %lld; from cosf / sinf at %lld\n", n, bad_c, bad_s, bad_f);
	return bad_c + bad_s > 0 ? 1 : 0;
}
