"""`sbwt search` on 10 M reads with the output on the scratch directory and on /dev/shm: where do the seconds after the last
batch go?  (SBWT_CLI_TIMING stage marks.)"""
import os, subprocess, sys, time, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbwt_amd import synth
n = int(os.environ.get("NREADS", 10_000_000)); L = 150
SBWT = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
d = tempfile.mkdtemp(prefix="sbwt_fs_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    genomes = synth.coli3_like(5_000_000)
    with open(d + "/g.fna", "wb") as f:
        for i, g in enumerate(genomes):
            f.write(b">g%d\n" % i + g.tobytes() + b"\n")
    subprocess.run([SBWT, "build", "-i", d + "/g.fna", "-o", d + "/i.sbwt", "-k", "30", "-t", "16"], check=True, capture_output=True)
    with open(d + "/r.fastq", "wb") as f:
        for lo in range(0, n, 1_000_000):
            m = min(1_000_000, n - lo)
            bases, _ = synth.sample_reads(genomes, m, L, 0.01, 42 + lo)
            rec = np.empty((m, 7 + 2 * L), dtype=np.uint8)
            rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8); rec[:, 3:3 + L] = bases.reshape(m, L)
            rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8); rec[:, 6 + L:6 + 2 * L] = ord("I"); rec[:, 6 + 2 * L] = 10
            rec.tofile(f)
    for out in (d + "/out.txt", "/dev/shm/sbwt_fs_out.txt", d + "/out.txt"):
        t0 = time.perf_counter()
        p = subprocess.run([SBWT, "search", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "-o", out], capture_output=True,
                           env=dict(os.environ, SBWT_CLI_TIMING="1"))
        dt = time.perf_counter() - t0
        marks = [l for l in p.stderr.decode().splitlines() if l.startswith("timing")]
        print(out, "wall %.2f s" % dt, "|", " | ".join(marks[-7:]), flush=True)
        if os.path.exists(out) and out.startswith("/dev/shm"): os.remove(out)
finally:
    shutil.rmtree(d, ignore_errors=True)
