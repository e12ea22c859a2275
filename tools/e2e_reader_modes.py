import os, subprocess, sys, time, tempfile, torch, numpy as np
ROOT = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, ROOT)
import bench as B
from sbwt_amd import synth
SBWT = os.path.join(ROOT, "sbwt_amd", "bin", "sbwt")
genomes = synth.coli3_like(5_000_000)
d = tempfile.mkdtemp(dir="/tmp")
with open(d + "/g.fna", "wb") as f:
    for i, g in enumerate(genomes): f.write(b">g%d\n" % i + g.tobytes() + b"\n")
subprocess.run([SBWT, "build", "-i", d + "/g.fna", "-o", d + "/i.sbwt", "-k", "30", "-t", "16"], check=True, capture_output=True)
n = 10_000_000
db = B.gpu_reads(genomes, n, 42, torch.device("cuda", 0)).view(n, 150).cpu().numpy()
rec = np.empty((n, 307), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8); rec[:, 3:153] = db; rec[:, 153:156] = np.frombuffer(b"\n+\n", dtype=np.uint8); rec[:, 156:306] = ord("I"); rec[:, 306] = ord("\n")
rec.tofile(d + "/r.fastq")
nk = n * 121
for rep in range(2):
  for env in ({"SBWT_CLI_SERIAL_READER": "1"}, {"SBWT_CLI_PARSER_THREADS": "1"}, {"SBWT_CLI_PARSER_THREADS": "2"}, {"SBWT_CLI_PARSER_THREADS": "3"}):
    if os.path.exists(d + "/out.txt"): os.remove(d + "/out.txt")
    t = time.time()
    p = subprocess.run([SBWT, "search", "-i", d + "/i.sbwt", "-q", d + "/r.fastq", "-o", d + "/out.txt"], capture_output=True, env=dict(os.environ, SBWT_CLI_TIMING="1", **env))
    dt = time.time() - t
    marks = [l[8:] for l in p.stderr.decode().splitlines() if l.startswith("timing:") and ("parse" in l or "loop left" in l or "index ready" in l)]
    print(env, "wall %.2f s -> %.2f G k-mers/s" % (dt, nk / dt / 1e9), marks, flush=True)
