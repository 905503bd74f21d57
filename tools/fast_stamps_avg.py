import sys, re
rows = []
for l in sys.stdin:
    m = re.search(r"load (\d+) A (\d+) \(q1 (\d+)\) B1 (\d+) \(q2 of wave 0: (\d+)\) B (\d+) C\+out (\d+) \((\d+)\)", l)
    if m: rows.append([int(x) for x in m.groups()])
import numpy as np
a = np.array(rows)
print("n", len(a)); print("mean load %.0f A %.0f q1 %.0f B %.0f nw0 %.0f C+out %.0f on %.1f" % (a[:,0].mean(), a[:,1].mean(), a[:,2].mean(), a[:,3].mean(), a[:,4].mean(), (a[:,5]+a[:,6]).mean(), a[:,7].mean()))
for lo, hi in ((0,1),(1,50),(50,200),(200,400),(400,5000)):
    s = a[(a[:,2]>=lo)&(a[:,2]<hi)]
    if len(s): print("q1 in [%d,%d): n %d load %.0f A %.0f B %.0f C+out %.0f" % (lo,hi,len(s), s[:,0].mean(), s[:,1].mean(), s[:,3].mean(), (s[:,5]+s[:,6]).mean()))
