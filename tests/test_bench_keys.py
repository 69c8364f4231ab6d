"""bench.py's mapping from the library's per-launch profile labels to the kernel|grid keys of the rocprofv3 PMC summary
(tools/pmc_summary.py): the `traffic` figure of the bench line is looked up through it, so a label the launchers change
must keep resolving to the kernel instantiation and grid that really ran."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("label,key", [
    ("M1536 N512 K6144 TN d192x128 s8 e0", "gemm_tn_direct_kernel<3, 2>|g65536"),
    ("M1536 N512 K6144 TN d192x128 s4 e0 x2", "gemm_tn_direct_kernel<3, 2>|g65536"),          # two products, half the split
    ("M1536 N1024 K6144 TN d192x128 s2 e0 x2", "gemm_tn_direct_kernel<3, 2>|g65536"),
    ("M6144 N1536 K1024 NT d192x192 s1 e0", "gemm_kc_direct_kernel<6, 6, false>|g65536"),
    ("M6144 N3072 K1024 bf3p9 t192x192 s1 e0", "gemm_bf3_kernel<2, 4, 6, 3, 9>|g262144"),     # bf16-piece products: 512 threads per workgroup
    ("M6144 N1024 K3072 bf3p9 t192x128 s1 e4", "gemm_bf3_kernel<4, 2, 3, 4, 9>|g131072"),
    ("M1536 N512 K6144 bf3p9 t192x128 s4 e0 x2", "gemm_bf3_kernel<4, 2, 3, 4, 9>|g131072"),
    ("M1536 N1024 K6144 bf3p6 t192x128 s2 e0 x2", "gemm_bf3_kernel<4, 2, 3, 4, 6>|g131072"),
    ("M6144 N1024 K1536 NN d192x128 s1 e4", "gemm_kc_direct_kernel<6, 4, true>|g65536"),
    ("M256 N1024 K2048 NT k32x32 s1 e1", "gemm_ks_kernel<2, 2, false, false>|g65536"),
    ("M48 N1536 K10 NT t64x64 s1 e0", "gemm_kernel<1, 1, false, false>|g6144"),
    ("gru_chain_fwd ms4 np2 T24 B256 H512", "gru_chain_fwd_kernel<4, 8, 1>|g65536"),
    ("gru_chain_fwd ms4x2 np2 T24 B256 H512", "gru_chain_fwd_kernel<4, 8, 2>|g65536"),         # two launches per CU
    ("gru_chain_bwd ms8 np4 T6 B256 H512", "gru_chain_bwd_kernel<8, 24, false>|g65536"),
    ("gru_chain_bwd ms4e np2 T24 B256 H512", "gru_chain_bwd_kernel<4, 24, true>|g65536"),      # ... the build that writes dgi row pieces
    ("gru_chain_bwd ms4 np2 T24 B256 H512", "gru_chain_bwd_kernel<4, 24, false>|g65536"),
    ("gru_chain_fwd v2w4 p9 np2 T24 B256 H512", "gru_chain2_fwd_kernel<4, 16, 9, false>|g65536"),      # second generation
    ("gru_chain_fwd v2w4e p9 np2 T24 B256 H512", "gru_chain2_fwd_kernel<4, 16, 9, true>|g65536"),     # ... the build that writes piece outputs
    ("gru_chain_bwd v2w4e p9 np2 T24 B256 H512", "gru_chain2_bwd_kernel<4, 48, 9, true>|g65536"),
    ("gru_chain_bwd v2w4 p6 np2 T24 B256 H512", "gru_chain2_bwd_kernel<4, 48, 6, false>|g65536"),
    ("gru_chain_fwd v2w4 p9 np2 T6 B128 H256", "gru_chain2_fwd_kernel<4, 8, 9, false>|g32768"),
    ("gru_chain_bwd ms2 np2 T6 B128 H512", "gru_chain_bwd_kernel<2, 24, false>|g65536"),
    ("adam", "adam_kernel|"),
])
def test_profile_label_to_pmc_key(bench, label, key):
    assert bench.pmc_key(label) == key


def test_committed_pmc_file_has_the_dominant_kernel(bench):
    """The traffic figure the bench line quotes for its dominant kernel comes from this committed file."""
    import json
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))["kernels"]
    label = "M1536 N512 K6144 TN d192x128 s4 e0 x2"
    hit = pmc[bench.pmc_key(label) + "#" + label]
    assert 100.0 < hit["hbm_mbytes_per_launch"] < 200.0      # 107 MB algorithmic (two 53.5 MB products)
    assert pmc["adam_kernel|g1048576"]["hbm_mbytes_per_launch"] == pytest.approx(494.7, rel=0.01)   # the calibration point


def test_piece_products_of_labels(bench):
    """Kernels on the bf16 pipe are priced against that pipe's roof divided by their piece products per f32 product."""
    assert bench.piece_products("M6144 N3072 K1024 bf3p9 t192x192 s1 e0") == 9
    assert bench.piece_products("M1536 N1024 K6144 bf3p6 t192x128 s2 e0 x2") == 6
    assert bench.piece_products("gru_chain_bwd v2w4 p9 np2 T24 B256 H512") == 9
    assert bench.piece_products("gru_chain_bwd v2w4e p9 np2 T24 B256 H512") == 9
    assert bench.piece_products("gru_chain_fwd ms4 np2 T24 B256 H512") == 0
    assert bench.piece_products("M1536 N512 K6144 TN d192x128 s4 e0 x2") == 0


def test_stale_traffic_file_is_refused(bench):
    """roofline.traffic comes from a committed PMC summary: when the library launches a label (with a PMC key) that the file's
    passes never saw, no figure of the file is attached and the missing labels are reported; with every label listed the
    per-shape figure is attached; files without a label list (rounds 2-3) keep working."""
    lab_a, lab_b = "M6144 N3072 K1024 bf3p9 t192x192 s1 e0", "gru_chain_bwd ms4 np2 T24 B256 H512"
    doc = {"labels": [lab_a], "label_hash": bench.label_hash([lab_a]),
           "kernels": {bench.pmc_key(lab_a): {"hbm_mbytes_per_launch": 325.6}, bench.pmc_key(lab_b): {"hbm_mbytes_per_launch": 474.0}}}
    table = [{"kernel": lab_a}, {"kernel": lab_b}, {"kernel": "dropout mask"}]
    assert bench.attach_traffic(table, doc) == [lab_b]
    assert all("traffic_mbytes_per_launch" not in r for r in table)
    doc["labels"].append(lab_b)
    assert bench.attach_traffic(table, doc) == []
    assert table[0]["traffic_mbytes_per_launch"] == 325.6 and table[1]["traffic_mbytes_per_launch"] == 474.0
    table2 = [{"kernel": lab_b}]
    assert bench.attach_traffic(table2, {"kernels": doc["kernels"]}) == [] and table2[0]["traffic_mbytes_per_launch"] == 474.0
    assert bench.piece_products("gru_step_bf3 p9 np2 B2048 H512 sv") == 9


# ------------------------------------------------------------------------------------------------ the ONE stdout line
def _canned():
    """A full result as main() assembles it: round 5's 20 KB line, the one the driver could not parse (BENCH_r05.json)."""
    import json
    return json.load(open(os.path.join(ROOT, "profiles", "r05_zz_bench_driver_command.json")))


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def test_stdout_line_is_small_and_complete(bench):
    import json
    full = _canned()
    assert len(json.dumps(full)) > 16384                     # the canned result really is the one that was dropped
    text = bench.compact_line(full)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 4096
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert line["config"]["workload"].startswith("MeasureVAE training") and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][k] == full["roofline"][k], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-3)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert line["cpu_baseline"][k] == full["cpu_baseline"][k], k
    assert line["parity_checked"] is True and line["chain_timeouts"] == 0
    for k in ("latent_ms", "latent_ar_ms", "arnn_tf_ms", "arnn_fr_ms", "decode_b1_ms", "decode_b16_ms", "vae4096_measures_per_s"):
        assert isinstance(line["extras"][k], float), k
    assert all(not isinstance(v, (dict, list)) for v in line["extras"].values())      # one scalar per extra, no tables


def test_stdout_line_stays_under_the_limit_whatever_the_extras_hold(bench):
    """The limit is hard: a result with absurdly long fields still prints a parseable line with every contract key."""
    import json
    full = _canned()
    full["config"]["workload"] = "w" * 6000
    full["cpu_baseline"]["sample"] = "s" * 6000
    full["first_steps_ms"] = [3.5] * 32
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["value"] == full["value"]


def test_multi_gpu_line_and_launcher(bench):
    """--gpus 8: the launcher's command and environment, and rank 0's line with the data-parallel fields, under the limit."""
    import json
    import sys
    cmd, env = bench.rank_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], environ={"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    assert bench.rank_command(2, [], environ={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})[1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    full = _canned()
    full.update(n_gpus=8, value=8 * full["value"], cpu_baseline=None,
                per_rank_units_per_s=[71305.1] * 8, allreduce_ms_per_step=0.4321, allreduce_mbytes=70.7,
                dp={"ranges": [{"kind": "bucket", "mbytes": 10.0}] * 7, "ms_per_step_without_exchange": 3.6,
                    "exposed_exchange_ms_per_step": 0.2, "chain_timeouts_per_rank": [0] * 8, "skipped_steps_per_rank": [0] * 8,
                    "side_streams_in_rotation": 1})
    full["config"].update(global_batch=2048, parallelism="dp8")
    full["extras"] = {"latent_rnn_train_dp": {"sequences_per_s": 100000.0, "measures_per_s": 1.6e6, "ms_per_step": 10.2,
                                              "allreduce_ms": 1.0, "arena_mbytes": 159.6, "workload": "x" * 300}}
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak"
    assert len(line["per_rank_units_per_s"]) == 8 and line["allreduce_ms_per_step"] == 0.4321 and line["allreduce_mbytes"] == 70.7
    assert line["dp"]["exposed_exchange_ms_per_step"] == 0.2 and line["extras"]["latent_dp_ms"] == 10.2
    assert line["cpu_baseline"] is None and line["vs_baseline"] is None
