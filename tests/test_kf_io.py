"""CPU: the on-disk keyframe record of the reference (EST.cpp:2440-2616, 2682-2700) as written and
read by the facade (host/ssk_kf_io.h).  The bytes are parsed here independently, straight from the FORMAT comments of
the reference, so that a third party with the real dependencies can exchange keyframes with this build."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "srba-stereo-slam_amd", "host", "kf_io_tool")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MATCH = np.dtype([("id", "<u8"), ("queryIdx", "<i4"), ("trainIdx", "<i4"), ("distance", "<f4"), ("imgIdx", "<i4")])


def parse(buf):
    off = 0

    def take(dt, n=1):
        nonlocal off
        a = np.frombuffer(buf, dt, n, off)
        off += a.nbytes
        return a
    rec = {"kf_id": int(take("<u8")[0]), "coords": take("<f8", 3).copy(), "rotvec": take("<f8", 3).copy()}
    for eye in ("left", "right"):
        n = int(take("<u8")[0])
        rec["kp_" + eye] = take(KP, n).copy()
        rows, cols, typ = take("<i4", 3)
        assert rows == n and typ == 0 and cols == (32 if n else 0)               # cv::Mat(N, 32, CV_8U); an empty Mat has cols = 0
        rec["desc_" + eye] = take("u1", rows * cols).reshape(rows, cols).copy()
    n = int(take("<u8")[0])
    rec["matches"] = take(MATCH, n).copy()
    assert off == len(buf)
    return rec


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        import __graft_entry__ as ge
        ge.build()
    assert os.path.exists(TOOL)
    return TOOL


@pytest.mark.parametrize("nl,nr,nm", [(500, 497, 361), (1, 1, 1), (0, 0, 0)])
def test_record_layout_and_roundtrip(tool, tmp_path, nl, nr, nm):
    a, b = str(tmp_path / "a.kfrec"), str(tmp_path / "b.kfrec")
    subprocess.run([tool, "make", a, str(nl), str(nr), str(nm), "7"], check=True)
    ra = parse(open(a, "rb").read())
    assert len(ra["kp_left"]) == nl and len(ra["kp_right"]) == nr and len(ra["matches"]) == nm
    assert os.path.getsize(a) == 8 + 48 + 2 * 8 + (nl + nr) * (28 + 32) + 2 * 12 + 8 + nm * 24
    if nl == 0:
        # a side without keypoints is an empty cv::Mat on the reference's side: rows = cols = 0 on disk (EST.cpp:2534-2537)
        assert np.array_equal(np.frombuffer(open(a, "rb").read(), "<i4", 3, 8 + 48 + 8), [0, 0, 0])
    if nm:
        assert np.array_equal(ra["matches"]["id"], 1000 + np.arange(nm))          # generateMatchesIDs (KF.h:75-80)
        assert ra["matches"]["queryIdx"].max() < nl and ra["matches"]["trainIdx"].max() < nr
    assert np.allclose(ra["coords"], [1.5, -0.25, 12.0]) and np.allclose(ra["rotvec"], [0.02, -0.4, 0.1], atol=1e-12)
    out = subprocess.run([tool, "roundtrip", a, b], check=True, capture_output=True, text=True).stdout
    assert "%d + %d keypoints, %d matches" % (nl, nr, nm) in out
    rb = parse(open(b, "rb").read())
    for k in ra:                                                                  # features and matches byte for byte; the pose goes
        if k in ("coords", "rotvec"):                                             # through a rotation matrix in the facade (1e-12)
            assert np.allclose(ra[k], rb[k], rtol=0, atol=1e-12)
        else:
            assert np.array_equal(ra[k], rb[k]), k


def test_truncated_record_is_rejected(tool, tmp_path):
    a, b = str(tmp_path / "a.kfrec"), str(tmp_path / "b.kfrec")
    subprocess.run([tool, "make", a, "20", "20", "10", "3"], check=True)
    data = open(a, "rb").read()
    open(a, "wb").write(data[:len(data) // 2])
    assert subprocess.run([tool, "roundtrip", a, b], capture_output=True).returncode != 0


def test_reference_written_empty_side_loads(tool, tmp_path):
    """A record as the reference writes it for a keyframe whose right side has no keypoints (descriptors = empty cv::Mat:
    rows 0, cols 0) must load, and a record that says cols = 32 with rows = 0 (what this build wrote before) too."""
    a, b = str(tmp_path / "a.kfrec"), str(tmp_path / "b.kfrec")
    subprocess.run([tool, "make", a, "5", "0", "0", "3"], check=True)
    data = bytearray(open(a, "rb").read())
    off = 8 + 48 + 8 + 5 * 28 + 12 + 5 * 32 + 8          # right side: count, then rows / cols / type
    assert list(np.frombuffer(bytes(data), "<i4", 3, off)) == [0, 0, 0]
    assert subprocess.run([tool, "roundtrip", a, b], capture_output=True).returncode == 0
    data[off + 4:off + 8] = np.int32(32).tobytes()
    open(a, "wb").write(bytes(data))
    assert subprocess.run([tool, "roundtrip", a, b], capture_output=True).returncode == 0
    assert list(np.frombuffer(open(b, "rb").read(), "<i4", 3, off)) == [0, 0, 0]
    data[off + 4:off + 8] = np.int32(7).tobytes()          # any other width is not a descriptor matrix
    open(a, "wb").write(bytes(data))
    assert subprocess.run([tool, "roundtrip", a, b], capture_output=True).returncode != 0
