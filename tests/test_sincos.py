"""include/ssk_sincos.h -- the cosine / sine of an ORB angle as cv::ORB forms them, shared by the HIP kernel and the CPU oracle -- against the
host libm: tools/sincos_check.c walks the float angles of [0, 360] (every 61st one here: 18.6 M angles, a second; the full walk of all
1 135 869 953 floats was run in the build container: no difference from (float)cos((double)x) / (float)sin((double)x) at any of them)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sincos_matches_libm_on_a_sample_of_every_binade(tmp_path):
    exe = str(tmp_path / "sincos_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tools", "sincos_check.c"), "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "61"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "cos differs from (float)cos((double)x) at 0, sin at 0" in r.stdout, r.stdout
