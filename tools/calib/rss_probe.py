"""What the process's resident set does when HIP memory is made: hipMalloc, hipMemset on it, hipHostMalloc, and how long each takes
(and the process's exit afterwards: run with `time`).  usage (on the GPU box): python tools/calib/rss_probe.py"""
import ctypes, os, time
hip = ctypes.CDLL("libamdhip64.so")


def rss():
	with open("/proc/self/status") as f:
		d = dict(l.split(":", 1) for l in f if l.startswith(("RssAnon", "RssFile", "RssShmem")))
	return ", ".join("%s %s" % (k, v.strip()) for k, v in d.items())


def step(what, fn):
	t = time.perf_counter(); rc = fn(); dt = time.perf_counter() - t
	print("%-44s rc %d  %.1f ms   %s" % (what, rc, 1e3 * dt, rss()), flush=True)


print("start: " + rss())
step("hipInit", lambda: hip.hipInit(0))
step("hipSetDevice(0)", lambda: hip.hipSetDevice(0))
p = ctypes.c_void_p()
step("hipMalloc 16 B (first)", lambda: hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(16)))
big = ctypes.c_void_p()
step("hipMalloc 1 GB", lambda: hip.hipMalloc(ctypes.byref(big), ctypes.c_size_t(1 << 30)))
if os.environ.get("PVC3_PROBE_SMALL_FIRST"):
	small = (ctypes.c_char * 1024)()
	step("hipMemset 16 B (the first kernel)", lambda: hip.hipMemset(p, 0, ctypes.c_size_t(16)))
	step("hipMemcpy H2D 1 KB from ordinary memory", lambda: hip.hipMemcpy(big, small, ctypes.c_size_t(1024), 1))
step("hipMemset 1 GB", lambda: hip.hipMemset(big, 0, ctypes.c_size_t(1 << 30)))
step("hipDeviceSynchronize", lambda: hip.hipDeviceSynchronize())
big2 = ctypes.c_void_p()
step("hipMalloc 1 GB again", lambda: hip.hipMalloc(ctypes.byref(big2), ctypes.c_size_t(1 << 30)))
h = ctypes.c_void_p()
step("hipHostMalloc 256 MB", lambda: hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(256 << 20), 0))
h2 = ctypes.c_void_p()
step("hipHostMalloc 256 MB again", lambda: hip.hipHostMalloc(ctypes.byref(h2), ctypes.c_size_t(256 << 20), 0))
src = (ctypes.c_char * (64 << 20))()
step("hipMemcpy H2D 64 MB from ordinary memory", lambda: hip.hipMemcpy(big, src, ctypes.c_size_t(64 << 20), 1))
step("the same again", lambda: hip.hipMemcpy(big, src, ctypes.c_size_t(64 << 20), 1))
step("hipFree 1 GB", lambda: hip.hipFree(big2))
step("hipHostFree 256 MB", lambda: hip.hipHostFree(h2))
if os.environ.get("PVC3_PROBE_FREE"):
	step("hipFree 1 GB", lambda: hip.hipFree(big))
	step("hipHostFree 256 MB", lambda: hip.hipHostFree(h))
print("leaving at %.3f" % time.time(), flush=True)
os._exit(0)
