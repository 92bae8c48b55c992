"""Cost of the host's small dense solves on this machine, with the BLAS pool as found and limited to one thread."""
import time, os
import numpy as np
from threadpoolctl import threadpool_info, threadpool_limits
print("cpus", os.cpu_count(), [(d["internal_api"], d["num_threads"]) for d in threadpool_info()])
a = np.random.rand(9, 9); a = a @ a.T
b = np.random.rand(50, 9, 9); b = b @ b.transpose(0, 2, 1)
c = np.random.rand(50, 6, 6) + 6 * np.eye(6); r = np.random.rand(50, 6, 1)


def t(f, n=2000):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


def table(tag):
    print(tag, "eigh 9x9 %.1f us | svd 9x9 %.1f | eigh 50x9x9 %.1f | solve 50x6x6 %.1f | lstsq 6x6 %.1f | matmul 9x9 %.2f" % (
        t(lambda: np.linalg.eigh(a)), t(lambda: np.linalg.svd(a)), t(lambda: np.linalg.eigh(b), 300), t(lambda: np.linalg.solve(c, r), 300),
        t(lambda: np.linalg.lstsq(c[0], r[0], rcond=None)), t(lambda: a @ a)))


table("pool as found:")
t0 = time.perf_counter()
with threadpool_limits(limits=1, user_api="blas"):
    t1 = time.perf_counter()
    table("one thread:   ")
    t2 = time.perf_counter()
t3 = time.perf_counter()
print("threadpool_limits enter %.1f us, exit %.1f us" % ((t1 - t0) * 1e6, (t3 - t2) * 1e6))
