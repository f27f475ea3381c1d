"""Per-kernel totals of ONE training step from a rocprofv3 --kernel-trace results .db (rocpd sqlite).
usage: python tools/db_summary.py <results.db> [step_index]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [s for n, s, e in rows if "stem_stats_kernel" in n or "stem_im2col" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
step = [(n, s, e) for n, s, e in rows if a <= s < b]
print("step %d: %.3f ms, %d kernels" % (k, (b - a) / 1e6, len(step)))
agg, cnt = collections.Counter(), collections.Counter()
for n, s, e in step:
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("frhip::", "").replace("_ZN5frhip", "")
    n = re.sub(r"^\d+", "", n)[:60]
    agg[n] += (e - s) / 1e6
    cnt[n] += 1
for n, t in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print("  %-62s %4d %8.3f ms  avg %7.1f us" % (n, cnt[n], t, t / cnt[n] * 1e3))
print("  sum of kernel durations %.2f ms" % sum(agg.values()))
