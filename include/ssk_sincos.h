/* ssk_sincos.h -- (cos, sin) of an ORB keypoint angle exactly as cv::ORB's computeDescriptors forms them (OpenCV 2.4 orb.cpp; reached
 * from the reference at EST.cpp:1135-1142 through stereo-vo):
 *     float angle = kpt.angle;  angle *= (float)(CV_PI/180.f);  float a = (float)cos(angle), b = (float)sin(angle);
 * i.e. the degree angle of fastAtan2 times the FLOAT constant, then the cosine / sine of that float, rounded to float.
 * The libm behind cos / sin differs from host to host and does not exist on the GPU, so both sides of the parity test -- the HIP kernel
 * (csrc/ssk_describe.hip) and the CPU oracle (oracle/oracle_orb.c) -- evaluate THIS function: double precision, a fixed operation order
 * (compile without contraction, as the whole front-end is), Cody-Waite reduction by pi/2 in three exact steps and the fdlibm kernel
 * polynomials (relative error < 2^-52 before the rounding to float).  tools/sincos_check.c compares it with the host libm's
 * (float)cos((double)x) / (float)sin((double)x) for EVERY float angle in [0, 360]: see DESIGN.md 2 for the count. */
#ifndef SSK_SINCOS_H
#define SSK_SINCOS_H

#if defined(__HIPCC__)
#define SSK_SINCOS_FN __host__ __device__ static inline
#else
#define SSK_SINCOS_FN static inline
#endif

SSK_SINCOS_FN void ssk_orb_sincos_deg(float angle_deg, float* c_out, float* s_out)
{
	const float rad_f = angle_deg * 0.01745329238474369f;       /* (float)(CV_PI/180.f) */
	const double x = (double)rad_f;                              /* 0 <= x <= 6.2832 */
	const int k = (int)(x * 6.36619772367581382433e-01 + 0.5);   /* nearest multiple of pi/2 */
	const double fk = (double)k;
	/* pi/2 = p1 + p2 + p3 (33 + 33 + 53 bits, fdlibm's rem_pio2 constants): k * p1 and k * p2 are exact for k <= 4 */
	double r = x - fk * 1.57079632673412561417e+00;
	r = r - fk * 6.07710050630396597660e-11;
	r = r - fk * 2.02226624879595063154e-21;
	const double z = r * r;
	/* fdlibm __kernel_sin / __kernel_cos on |r| <= pi/4 */
	const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
	const double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
	const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
	const double cs = 1.0 - (0.5 * z - z * pc);
	/* quadrant k: (c, s) = (cs, sn), (-sn, cs), (-cs, -sn), (sn, -cs).  Rounded to float first, selected and signed after: (float)(-v) = -(float)v
	 * exactly, and the GPU selects two floats instead of two doubles */
	{
		const float fs = (float)sn, fc = (float)cs;
		const float c = (k & 1) ? fs : fc, s = (k & 1) ? fc : fs;
		*c_out = ((k + 1) & 2) ? -c : c;
		*s_out = (k & 2) ? -s : s;
	}
}

#endif /* SSK_SINCOS_H */
