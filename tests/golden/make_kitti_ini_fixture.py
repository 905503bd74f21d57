#!/usr/bin/env python3
"""Generates tests/golden/kitti_ini.json from the reference's own parameter file, demo/config_imgdir_kitti_srba.ini.

Run in the BUILD container only (the reference tree does not exist on the GPU box); the JSON travels.  The file is data:
every `key = value` of every section, parsed with the conventions of MRPT's CConfigFile (`;` starts a comment line,
`//` ends a value), values typed as bool / int / float / list / string.  tests/test_kitti_ini.py then checks every preset
of this build that claims "the KITTI parameters" against it, field by field.
"""
import json
import os
import re
import sys

REF = "/root/reference/demo/config_imgdir_kitti_srba.ini"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti_ini.json")


def typed(v):
    v = v.strip()
    if v.lower() in ("true", "false"):
        return v.lower() == "true"
    if v.startswith("[") and v.endswith("]"):
        items = [x for x in re.split(r"[\s,]+", v[1:-1].strip()) if x]
        return [typed(x) for x in items]
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v)
    except ValueError:
        return v


def parse(path):
    out, sec = {}, None
    with open(path, "r", encoding="latin-1") as f:
        for ln, raw in enumerate(f, 1):
            line = raw.strip()
            if not line or line.startswith(";"):
                continue
            m = re.match(r"^\[(.+)\]$", line)
            if m:
                sec = m.group(1)
                out.setdefault(sec, {})
                continue
            if "=" not in line or sec is None:
                continue
            k, v = line.split("=", 1)
            v = v.split("//", 1)[0]
            out[sec][k.strip()] = {"value": typed(v), "line": ln}
    return out


def main():
    if not os.path.exists(REF):
        sys.exit("reference tree not present: run this in the build container")
    d = parse(REF)
    with open(OUT, "w") as f:
        json.dump({"source": "demo/config_imgdir_kitti_srba.ini", "sections": d}, f, indent=1, sort_keys=True)
    print("wrote %s: %d sections, %d keys" % (OUT, len(d), sum(len(s) for s in d.values())))


if __name__ == "__main__":
    main()
