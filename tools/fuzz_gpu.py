"""GPU fuzz: random indexes (k, genome shapes with repeats / related strains / tiny alphabets) and random reads
(substitutions, N, lower case, ragged lengths, long reads); every route (5 fused, 4 two-pass path kernel, 1 blocks) must
equal the reference-order kernel (variant 0) bit for bit, and a sample must equal the oracle.
Usage: python tools/fuzz_gpu.py [seconds]   (SEED=n);   tests/test_gpu_fuzz.py runs fuzz() under the driver's -m gpu suite"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sbwt_amd import capi, hostlib, synth
from oracle import OracleIndex



class FuzzMismatch(AssertionError):
    pass


def dev_i32(idx, bases, off, k, streaming):
    """The device-pointer call with int32 results (sbwtgpu_*_dev_i32), back as int64."""
    import torch
    dev = torch.device("cuda:0")
    ooff = capi.out_offsets(off, k)
    if len(bases) == 0 or int(ooff[-1]) == 0:
        return np.zeros(0, dtype=np.int64)
    d_b, d_ro, d_oo = torch.from_numpy(np.ascontiguousarray(bases)).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ooff).to(dev)
    wsb = capi.search_workspace_bytes(d_b.numel())
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    d32 = torch.full((int(ooff[-1]),), -99, dtype=torch.int32, device=dev)
    idx.streaming_search_dev_i32(d_b.data_ptr(), d_b.numel(), d_ro.data_ptr(), len(off) - 1, d32.data_ptr(), d_oo.data_ptr(),
                                 d_ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream, streaming)
    torch.cuda.synchronize()
    return d32.cpu().numpy().astype(np.int64)


def reset_tuning():
    capi.set_tuning("search_variant", -1)
    capi.set_tuning("sort_reads", -1)
    capi.set_tuning("debug", 0)
    capi.set_tuning("fused_pieces", -1)
    capi.set_tuning("fused_sort", 3632)
    capi.set_tuning("fused_table", 1)
    capi.set_tuning("path_lookahead", 8); capi.set_tuning("path_safe", 2); capi.set_tuning("image_level", 0)
    capi.set_tuning("path_stitch", 1); capi.set_tuning("path_stitch_min", 1)


def fuzz(budget, seed, max_cases=None):
    """Runs random cases for `budget` seconds (or max_cases); returns the number of cases.  Raises FuzzMismatch with a
    description of the first difference."""
    try:
        return _fuzz(budget, seed, max_cases)
    finally:
        reset_tuning()


def _fuzz(budget, seed, max_cases):
    capi.set_tuning("poison_results", 1)
    t_end = time.time() + budget
    case = 0
    rng = np.random.default_rng(seed)
    while time.time() < t_end and (max_cases is None or case < max_cases):
        case += 1
        k = int(rng.choice([4, 7, 12, 16, 21, 30, 31, 32, 33, 40, 63]))
        shape = int(rng.integers(0, 5))
        glen = int(rng.integers(2_000, 120_000))
        g0 = synth.random_genome(glen, int(rng.integers(1, 1 << 30)))
        if shape == 0:
            genomes = [g0]
        elif shape == 1:
            genomes = [g0, synth.mutate(g0, float(rng.choice([0.001, 0.01, 0.05])), int(rng.integers(1, 1 << 30)))]
        elif shape == 2:      # tandem repeats and a low-complexity stretch
            unit = g0[: int(rng.integers(3, 40))]
            genomes = [np.concatenate([g0[:500], np.tile(unit, 60), g0[500:1500], np.frombuffer(b"AC" * 200, dtype=np.uint8), g0[1500:]])]
        elif shape == 3:      # several short sequences (many dummy nodes)
            genomes = [g0[i:i + int(rng.integers(k, 4 * k + 10))].copy() for i in range(0, min(glen, 20_000), 997)]
        else:                 # star of related genomes
            genomes = [g0] + [synth.mutate(g0, 0.02, int(rng.integers(1, 1 << 30))) for _ in range(4)]
        genomes = [g for g in genomes if len(g) >= k + 2]
        ssup = bool(rng.integers(0, 2))
        rc = bool(rng.integers(0, 2))
        bits = hostlib.build_bits([g.tobytes() for g in genomes], k, rc, ssup, n_threads=4)
        # knobs that must not change results: how the paths are chosen, which safe rule, the image level
        capi.set_tuning("path_lookahead", int(rng.choice([0, 1, 8])))
        capi.set_tuning("path_safe", int(rng.choice([0, 1, 2, 2])))
        capi.set_tuning("image_level", int(rng.choice([0, 0, 0, 1, 2])))
        capi.set_tuning("path_stitch", int(rng.choice([1, 1, 1, 0])))
        capi.set_tuning("path_stitch_min", int(rng.choice([1, 1, 4, 16])))
        idx = capi.Index.create(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup if ssup else None,
                                bits.n_nodes, k, bits.n_kmers, int(rng.choice([0, 0, 2, min(k, 8)])))
        # reads
        nr = int(rng.integers(200, 4000))
        if rng.integers(0, 2):
            L = int(rng.integers(max(k - 2, 1), int(rng.choice([4 * k + 60, 480, 1500]))))      # (up to and past three fused pieces)
            long_enough = [g for g in genomes if len(g) >= L]
            if not long_enough:
                continue
            bases, off = synth.sample_reads(long_enough, nr, L, float(rng.choice([0, 0.005, 0.02, 0.1])), int(rng.integers(1, 1 << 30)))
        else:                 # ragged lengths
            cat = np.concatenate(genomes)
            lens = np.minimum(rng.integers(0, int(rng.choice([3 * k + 40, 330, 460, 2000])), size=nr), len(cat))
            st = (rng.random(nr) * (len(cat) - lens + 1)).astype(np.int64)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            bases = np.empty(int(off[-1]), dtype=np.uint8)
            for r in range(nr):
                bases[off[r]:off[r + 1]] = cat[st[r]:st[r] + lens[r]]
            flip = rng.random(len(bases)) < 0.01
            bases[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(flip.sum()))]
        if rng.integers(0, 4) == 0:      # a few long reads (cut into pieces on the device), some with lower-case stretches
            cat = np.concatenate(genomes)
            extra = []
            for _ in range(int(rng.integers(1, 12))):
                ln = int(min(rng.integers(260, 6000), len(cat)))
                s0 = int(rng.integers(0, len(cat) - ln + 1))
                rd = cat[s0:s0 + ln].copy()
                for _ in range(int(rng.integers(0, 4))):
                    a0 = int(rng.integers(0, ln)); a1 = min(ln, a0 + int(rng.integers(1, 400)))
                    rd[a0:a1] = np.frombuffer(rd[a0:a1].tobytes().lower(), dtype=np.uint8)
                extra.append(rd)
            pos = int(rng.integers(0, len(off)))                      # ... somewhere among the others
            lens = np.diff(off)
            parts = [bases[off[r]:off[r + 1]] for r in range(len(lens))]
            parts[pos:pos] = extra
            bases = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
            off = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.int64)
            nr = len(parts)
        if len(bases) > 100:
            bases = synth.inject(bases, int(rng.integers(0, 30)), ord("N"), int(rng.integers(1, 1 << 30)))
            bases = synth.inject(bases, int(rng.integers(0, 30)), int(rng.choice(list(b"acgtn"))), int(rng.integers(1, 1 << 30)))
        res = {}
        for v in (0, 1, 4, 5):
            capi.set_tuning("search_variant", v)
            capi.set_tuning("sort_reads", int(rng.integers(0, 2)) if v == 4 else -1)
            # the fused kernel's alignments (anchors, seeds, resumed compares): off / as shipped / for every k
            # (+128, at random: the general instantiation instead of the one for batches of one read length)
            # (+ workgroups << 8, at random: the launcher's choice by batch size, one workgroup, a few, many)
            capi.set_tuning("debug", (int(rng.choice([0, 0, 32, 64])) | int(rng.choice([0, 128])) |
                                      (int(rng.choice([0, 0, 1, 3, 40, 1280])) << 8)) if v == 5 else 0)
            capi.set_tuning("fused_pieces", int(rng.choice([-1, 1, 2, 3])) if v == 5 else -1)
            # the fused kernel with its lanes sorted by state (k <= 31; + 4096: always, whatever the workspace's hint says): off / no
            # waiting, no shipping / waves wait for 48 busy lanes, two finished reads in eight written by the searchers / ... by the
            # followers' load / every one shipped / as shipped: by the hint the call before left in the workspace
            capi.set_tuning("fused_sort", int(rng.choice([0, 4097, 4656, 7728, 7728, 6192, 6448, 3632])) if v == 5 else 0)
            # the ticket table for batches with many long reads (round 6): on, as shipped / off (the general kernel takes them)
            capi.set_tuning("fused_table", int(rng.choice([1, 1, 1, 0])))
            a = idx.streaming_search(bases, off)[0] if ssup else None
            b = idx.search(bases, off)[0]
            res[(v, -1)] = (a, b)
            if bits.n_nodes < (1 << 31) and int(rng.integers(0, 4)) == 0:      # the same route writing int32 results
                res[(v, 32)] = (dev_i32(idx, bases, off, k, True) if ssup else None, dev_i32(idx, bases, off, k, False))
        ref = res[(0, -1)]

        def explain(got, want):
            """where two result vectors differ: the read, its text, both results around the first difference"""
            d = int(np.flatnonzero(got != want)[0])
            oo = np.concatenate([[0], np.cumsum(np.maximum(np.diff(off) - k + 1, 0))])
            r = int(np.searchsorted(oo, d, side="right") - 1)
            print(" first difference at result", d, "= k-mer", d - int(oo[r]), "of read", r, "length", int(off[r + 1] - off[r]),
                  "differences in all:", int((got != want).sum()))
            print(" read:", bases[off[r]:off[r + 1]].tobytes().decode("latin1"))
            print(" got :", got[oo[r]:oo[r + 1]].tolist())
            print(" want:", want[oo[r]:oo[r + 1]].tolist())
            print(" index: n_nodes", bits.n_nodes, "device precalc", idx.device_precalc_k, "paths", idx.n_paths, "branching", idx.n_branch)

        for key, (a, b) in res.items():
            if ssup and not np.array_equal(a, ref[0]):
                explain(a, ref[0]); raise FuzzMismatch("MISMATCH streaming %s seed %d case %d k %d shape %d ssup %s rc %s" % (key, seed, case, k, shape, ssup, rc))
            if not np.array_equal(b, ref[1]):
                explain(b, ref[1]); raise FuzzMismatch("MISMATCH search %s seed %d case %d k %d shape %d ssup %s rc %s" % (key, seed, case, k, shape, ssup, rc))
        if case % 10 == 1 and bits.n_nodes < 400_000:     # the oracle on a sample
            orc = OracleIndex.from_bits(bits.cols[0], bits.cols[1], bits.cols[2], bits.cols[3], bits.ssup if ssup else None,
                                        bits.n_nodes, k, bits.n_kmers, 0)
            for r in range(min(nr, 40)):
                s = bases[off[r]:off[r + 1]].tobytes()
                want = orc.streaming_search(s) if ssup else orc.search_all(s)
                oo = np.concatenate([[0], np.cumsum(np.maximum(np.diff(off) - k + 1, 0))])
                got = (ref[0] if ssup else ref[1])[oo[r]:oo[r + 1]]
                if not np.array_equal(got, want):
                    raise FuzzMismatch("MISMATCH vs oracle: seed %d case %d read %d" % (seed, case, r))
    return case


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    try:
        n = fuzz(budget, int(os.environ.get("SEED", 1)))
    except FuzzMismatch as ex:
        print(ex)
        sys.exit(1)
    print("fuzz ok:", n, "cases")
