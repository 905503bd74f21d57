import sys, os, subprocess, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, synth
frames, strips = synth.synth_translation_sequence(3, 4)
d = tempfile.mkdtemp()
for k, (l, r) in enumerate(frames):
    l.tofile(os.path.join(d, "left_%04d.raw" % k)); r.tofile(os.path.join(d, "right_%04d.raw" % k))
p = subprocess.run(["srba-stereo-slam_amd/host/facade_demo", d, "4", str(synth.W), str(synth.H), d], capture_output=True, text=True)
print(p.stdout); print(p.stderr[-500:])
print("strips", strips)
