"""What does the box's file system charge for the CLI's output?  Writes N GiB with write(2) in 64 MiB pieces and times the
writes and the close(), on the scratch directory and on /dev/shm (tmpfs).  usage: python tools/io_probe.py [GiB]"""
import os, sys, time
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
buf = bytes(64 << 20)
for d in (os.environ.get("TMPDIR", "/tmp"), "/dev/shm"):
    p = os.path.join(d, "sbwt_io_probe.bin")
    try:
        t0 = time.perf_counter()
        fd = os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
        for _ in range(n * 16):
            os.write(fd, buf)
        t1 = time.perf_counter()
        os.close(fd)
        t2 = time.perf_counter()
        print("%-10s write %d GiB: %.2f s (%.1f GB/s), close: %.2f s" % (d, n, t1 - t0, n * 1.0737 / (t1 - t0), t2 - t1), flush=True)
        t0 = time.perf_counter(); os.remove(p); print("%-10s unlink: %.2f s" % (d, time.perf_counter() - t0), flush=True)
    except Exception as ex:
        print(d, "failed:", ex)
