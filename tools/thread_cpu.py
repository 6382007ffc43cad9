"""Diagnostic (not part of the product): runs a command, and for a window of its execution samples the CPU time of every thread of
the process tree (/proc/<pid>/task/*/stat), so that the host side of a pipeline can be attributed: which threads burn the cgroup's
CPU quota.  usage: python tools/thread_cpu.py START_S WINDOW_S -- cmd ..."""
import os
import subprocess
import sys
import time

start, window = float(sys.argv[1]), float(sys.argv[2])
cmd = sys.argv[sys.argv.index("--") + 1:]
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)


def tree(pid):
    out = [pid]
    try:
        for t in os.listdir("/proc/%d/task" % pid):
            try:
                kids = open("/proc/%d/task/%s/children" % (pid, t)).read().split()
            except Exception:
                kids = []
            for k in kids:
                out += tree(int(k))
    except Exception:
        pass
    return out


def snap():
    res = {}
    tck = os.sysconf("SC_CLK_TCK")
    for pid in set(tree(p.pid)):
        try:
            for t in os.listdir("/proc/%d/task" % pid):
                try:
                    st = open("/proc/%d/task/%s/stat" % (pid, t)).read()
                    name = st[st.index("(") + 1:st.rindex(")")]
                    f = st[st.rindex(")") + 2:].split()
                    res[(pid, int(t))] = (name, (int(f[11]) + int(f[12])) / tck)
                except Exception:
                    pass
        except Exception:
            pass
    return res


time.sleep(start)
a = snap()
time.sleep(window)
b = snap()
rows = sorted(((b[k][1] - a[k][1]) / window, b[k][0], k) for k in b if k in a)
tot = sum(r[0] for r in rows)
print("threads: %d, total %.2f cores over a %.0f s window" % (len(rows), tot, window))
byname = {}
for c, n, k in rows:
    byname.setdefault(n, [0, 0.0])
    byname[n][0] += 1
    byname[n][1] += c
for n, (cnt, c) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-20s x%-3d %.2f cores" % (n, cnt, c))
print("  busiest threads: " + ", ".join("%s %.2f" % (n, c) for c, n, k in rows[-12:][::-1]))
out = p.communicate()[0]
sys.stdout.write(out.decode()[-300:])
