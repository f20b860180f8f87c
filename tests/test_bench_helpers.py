"""bench.py's bookkeeping that needs no GPU: the algorithmic-traffic table, the PMC file it reads `roofline.traffic`
from (the kernels it sums must be the ones the committed profile lists -- a renamed kernel must not silently turn the figure into
None), and the shape of the committed bench line."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_traffic_table():
    import bench
    assert bench.ALGO["mixnet"] == 55172 * 8 * 8 + 4 * 64 * 8          # 3.53 MB per input byte
    assert 8.5e6 < bench.lstm_algo_bytes(205) < 9.2e6                    # 8.86 MB per byte at enwik8's alphabet
    assert bench.lstm_algo_bytes(145) < bench.lstm_algo_bytes(205)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_pmc_file_lists_every_kernel_the_bench_sums():
    import bench
    with open(bench.PMC_FILE) as f:
        z = json.load(f)
    assert z["_meta"]["stream_bytes_processed"] > 100000
    for stage, kernels in bench.PMC_KERNELS.items():
        for k in kernels:
            assert any(name.startswith(k) for name in z), (stage, k, "not in the committed PMC profile: roofline.traffic would be null")
        v = bench.pmc_traffic_per_byte(stage)
        assert v is not None and v > 0, stage
    # the mixing network moves less than its algorithmic bytes (rows whose selector repeats stay in registers), more than a tenth of them
    assert 0.1 * bench.ALGO["mixnet"] < bench.pmc_traffic_per_byte("mixnet") < bench.ALGO["mixnet"]


def test_committed_bench_line_has_the_contract_fields():
    import glob
    with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_1m.json")))[-1]) as f:   # the newest round's
        d = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0
    assert d["verified"]["identical_to_reference_file"] is True
    assert d["value"] > 20 * c["value"]
