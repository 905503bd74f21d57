/*
 * oracle_geom.c -- CPU oracle (test infrastructure; parity unpinned, see ssk_oracle.h):
 * landmark initialisation by stereo triangulation.
 */
#include "ssk_oracle.h"

/* [literal UTIL.h:558-574] projectMatchTo3D.  b_d = baseline/(fl*(cur-ur)+fr*(ul-cul));
 * P = (b_d*fr*(ul-cul), b_d*fr*(vl-cvl), b_d*fl*fr), all in double, operation order as written. */
void ssko_project_match_to_3d(double ul, double vl, double ur, const ssko_stereo_cam* cam, double out_xyz[3])
{
	const double cul = cam->cx_l, cvl = cam->cy_l, fl = cam->fx_l;
	const double cur = cam->cx_r, fr = cam->fx_r, baseline = cam->baseline;
	const double b_d = baseline / (fl * (cur - ur) + fr * (ul - cul));
	out_xyz[0] = b_d * fr * (ul - cul);
	out_xyz[1] = b_d * fr * (vl - cvl);
	out_xyz[2] = b_d * fl * fr;
}
