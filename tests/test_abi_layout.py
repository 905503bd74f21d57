"""The ctypes mirror (srba-stereo-slam_amd/capi.py, tests and bench only) against the C headers: every structure that crosses
the C-ABI must have the size the C compiler gives it, and the numpy record dtypes the itemsize of their C struct.  gcc compiles
a few lines that print sizeof for each struct of include/*.h; no GPU, no library call."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PAIRS = {                                   # C struct -> ctypes class / numpy dtype in capi.py
    "ssk_kf_view": "KfView", "ssk_da_params": "DaParams", "ssk_cip_params": "CipParams", "ssk_cip_result": "CipResult",
    "ssk_fm_params": "FmParams", "ssk_fm_result": "FmResult", "ssk_frontend_params": "FrontendParams", "ssk_kf_out": "KfOut",
    "ssk_pg_params": "PgParams", "ssk_pg_result": "PgResult", "ssk_rba_options": "RbaOptions", "ssk_rba_kf_info": "RbaKfInfo",
    "ssk_slam_frame_report": "SlamFrameReport", "ssk_slam_timing": "SlamTiming", "ssk_lm_window": "LmWindow", "ssk_lm_params": "LmParams",
    "ssk_lm_result": "LmResult", "ssk_lm_trace_entry": "LmTraceEntry",
    "ssk_keypoint": "KEYPOINT_DTYPE", "ssk_dmatch": "DMATCH_DTYPE", "ssk_rba_obs": "RBA_OBS_DTYPE",
}


def test_ctypes_structures_match_the_headers(pkg, tmp_path):
    from ssk_amd import capi
    src = tmp_path / "sizes.c"
    lines = ['#include <stdio.h>', '#include "ssk_capi.h"', '#include "ssk_rba.h"', '#include "ssk_slam.h"', "int main(void) {"]
    lines += ['    printf("%s %%zu\\n", sizeof(%s));' % (c, c) for c in PAIRS]
    lines += ["    return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    c_size = {k: int(v) for k, v in (ln.split() for ln in out.strip().splitlines())}
    assert set(c_size) == set(PAIRS)
    for cname, pyname in PAIRS.items():
        obj = getattr(capi, pyname)
        size = obj.itemsize if hasattr(obj, "itemsize") else C.sizeof(obj)
        assert size == c_size[cname], (cname, pyname, size, c_size[cname])
