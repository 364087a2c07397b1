"""Aggregate a tools/step_census2.py listing by (entry point, kernel size / stride)."""
import re, collections, sys
def agg(f):
    cat = collections.defaultdict(lambda:[0,0.0,0.0])
    for l in open(f):
        m = re.match(r'\s*([\d.]+) ms\s+[\d.]+ %\s+(\d+) x\s+([\d.]+)\s+(\S+)\s+(.*)', l)
        if not m: continue
        t, n, per, name, rest = float(m.group(1)), int(m.group(2)), float(m.group(3)), m.group(4), m.group(5)
        k = re.search(r'k(\d)x(\d) s(\d)', rest)
        key = name
        if k and name in ('conv_igemm','conv_wgrad','conv3x3_wino','conv_igemm_multi'):
            key = f"{name} k{k.group(1)}x{k.group(2)} s{k.group(3)}"
        tf = re.search(r'^\s*(\d+) TF', rest)
        cat[key][0]+=n; cat[key][1]+=t
        if tf: cat[key][2]+= float(tf.group(1))*t
    tot=sum(v[1] for v in cat.values())
    print(f, "%.1f ms" % tot)
    for k,(n,t,fl) in sorted(cat.items(), key=lambda kv:-kv[1][1])[:32]: print(f"{t:8.2f} ms {100*t/tot:5.1f}% {n:4d} {k:34s} {fl/t if fl else 0:6.0f} TF")
for f in sys.argv[1:]: agg(f)
