"""median / mean of the reduce-kernel times in an ab.sh log"""
import re, statistics as st, sys
a, b, cur = [], [], None
for l in open(sys.argv[1]):
    if l.startswith("A:"): cur = a
    elif l.startswith("B:"): cur = b
    m = re.search(r"= \['([0-9.]+)'.*'([0-9.]+)'\] \(min", l)
    if m and cur is not None: cur.append((float(m.group(1)), float(m.group(2))))
for name, v in (("A", a), ("B", b)):
    print(name, "reduce median %.3f mean %.3f | total median %.3f" % (st.median(x[0] for x in v), st.mean(x[0] for x in v), st.median(x[1] for x in v)), [x[0] for x in v])
