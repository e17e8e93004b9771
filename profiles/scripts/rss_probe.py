import os, sys, threading, time, subprocess
import psutil
p = subprocess.Popen([sys.executable, "tests/fuzz_gpu.py", "--iters", "6000", "--seed", "20261102", "--modes", "0,1,2,4", "--counts"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
ps = psutil.Process(p.pid)
t0 = time.time()
while p.poll() is None:
    try:
        print("t=%.0f rss=%.0f MB" % (time.time() - t0, ps.memory_info().rss / 1e6), flush=True)
    except Exception as e:
        break
    time.sleep(10)
out = p.communicate()[0]
print(out[-400:])
print("rc", p.returncode, "host mem", psutil.virtual_memory())
