"""bench.py's self-policing pieces (CPU): a PMC traffic figure is attached only to a line of the kernel, image level, route,
workload and sources it was measured on (level-1/2 lines once carried the fused kernel's bytes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_dominant_kernel_names():
    assert bench.dominant_kernel(5, 0) == "k_search_fused"
    assert bench.dominant_kernel(4, 0) == "k_search_cert<PATH,SEG>"
    assert bench.dominant_kernel(5, 1) == "k_search_cert" and bench.dominant_kernel(1, 2) == "k_search_cert"
    assert bench.dominant_kernel(0, 0) == "k_search"


def test_traffic_is_attached_only_to_the_line_it_belongs_to(tmp_path, monkeypatch):
    sha = bench.kernel_source_sha16()
    ent = {"kernel": "k_search_fused", "image_level": 0, "search_variant": 5, "reads_per_gpu": 10_000_000,
           "kernel_source_sha16": sha, "hbm_bytes_per_launch": 3.0e10, "read_requests_128B": 1.5e8}
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    (root / "sbwt_amd").symlink_to(os.path.join(ROOT, "sbwt_amd"))
    json.dump({bench.traffic_key(2, 0, 5): ent, "config2": dict(ent)}, open(root / "profiles" / "traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(root))
    assert bench.load_traffic(2, 10_000_000, 0, 5, "k_search_fused") == ent
    assert bench.load_traffic(2, 10_000_000, 1, 1, "k_search_cert") is None          # another image level / route: no entry
    assert bench.load_traffic(2, 10_000_000, 0, 5, "k_search_cert") is None          # another kernel
    assert bench.load_traffic(2, 4_000_000, 0, 5, "k_search_fused") is None          # another workload
    assert bench.load_traffic(3, 10_000_000, 0, 5, "k_search_fused") is None         # another config
    monkeypatch.setenv("SBWTGPU_PATH_SAFE", "0")                                     # another build of the image
    assert bench.load_traffic(2, 10_000_000, 0, 5, "k_search_fused") is None


def test_committed_traffic_entries_name_their_kernels():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, ent in tj.items():
        cfg, lvl, var = key.split("_")
        assert ent["image_level"] == int(lvl[len("level"):]) and ent["search_variant"] == int(var[len("variant"):])
        assert ent["kernel"] == bench.dominant_kernel(ent["search_variant"], ent["image_level"]), key


def test_gpus_flag_fails_fast_without_devices():
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "SBWT_BENCH_FORCE_DEVICE")}, timeout=300)
    assert p.returncode == 2 and "only" in p.stderr and "HIP device" in p.stderr


def test_request_model_prices_reads_and_writes_against_the_measured_ceilings():
    # config 2 as profiled in round 4: 154.2 M read lines + 169.6 M 64-byte write requests per launch, kernel 5.149 ms
    rm = bench.request_model(154.194e6, 169.645e6, 5.149)
    assert abs(rm["floor_ms"] - (154.194e6 / 55.7e9 + 169.645e6 * 64 / 6.46e12) * 1e3) < 1e-9
    assert 0.80 < rm["frac_of_deliverable"] < 0.95
    assert bench.request_model(1.0, 1.0, 0.0)["frac_of_deliverable"] is None
