"""bench.py's final stdout line must stay short enough for the driver to parse (r05: 21 KB, `parsed: null`) and carry the contract
keys with `roofline` and `cpu_baseline` (CPU test on canned numbers: no GPU needed for the line builder)."""
import json

import bench


def canned(extra_legs=12):
    big = {"median_ms": 0.123456789, "p10_ms": 0.12, "p90_ms": 0.13, "n": 50, "note": "x" * 400}
    extra = {f"leg{i}": {"forward_kernel": big, "roofline_forward": {"frac": 0.5}, "blob": "y" * 1500} for i in range(extra_legs)}
    for key in ("K2_train", "K4_3_18_euler", "NSDE_3_18_srk_K4_shape", "NSDE_3_18_milstein_K4_shape"):
        extra[key] = {"forward_kernel": big, "roofline_forward": {"frac": 0.39}, "forward_backward": big, "roofline_training": {"frac": 0.22},
                      "forward_backward_graph_replay": big, "roofline_bytes": {"traffic": 983.5e6}}
    for key in ("K3_strong", "K3_shard_512"):
        extra[key] = {"kernel_ms": 1.2345678, "roofline": {"frac": 0.857}, "ms_per_solve": 1.25}
    for key in ("K5_strong_train", "K5_shard_128_train"):
        extra[key] = {"ms_per_step": 1.286, "roofline": {"frac": 0.085}}
    for key in ("K1_tutorial_lsde", "tutorial_field"):
        extra[key] = {"fused": big, "fused_forward_backward": big, "optimizer_step_graph_replayed": big}
    out = {
        "metric": "SDE solver steps/sec (batch x steps / s), forward solve", "value": 4.97e8, "unit": "row-steps/s", "n_gpus": 1,
        "steps": 50, "warmup": 10, "ms_per_step": 0.2061, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "K2: Neural LNSDE (io=4,no=17) NL=2 B=1024/GPU H=128 C=21 L=101 natural-spline coeffs 30% NaN, 100 Euler "
                               "steps dt=1, ts=[0,100], in-kernel Philox dW", "rows_per_gpu": 1024, "solver_steps": 100, "global_rows": 1024,
                   "parallelism": "row-shard x1, no collective in the solver", "kernel": "auto", "prepare": "every call " + "z" * 80},
        "timing": {"solve_call": big, "solve_call_with_prepare": big, "solve_kernel": big, "note": "n" * 900},
        "roofline": {"bound": "mfma", "achieved": 93.54321, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.59468, "traffic": 38070272,
                     "traffic_source": "s" * 600, "launched_path": "lean", "profiled_kernel": "k" * 60, "kernel_ms": 0.18581,
                     "flop_per_rowstep": 169728, "bytes_per_rowstep": 346, "executed_flop_per_rowstep": 106496, "executed_frac": 0.3731,
                     "hbm_frac": 0.0238, "hbm_achieved_GBs": 190.6, "note": "n" * 700},
        "extra": extra,
        "cpu_baseline": {"value": 643999.7, "unit": "row-steps/s", "cores": 16, "threads": 16, "usable_cores": 96, "value_1thread": 241000.2,
                         "kind": "port", "sample": "5000+800 Euler steps of K2 (B=1024) in 9.0+3.0 s, oracle/torch_loop.py, torch 2.10.0+rocm7.0 CPU f32",
                         "note": "n" * 300},
        "speedup_vs_cpu": 771.9, "full_record": "gpurun_out/bench_full.json",
    }
    out["summary"] = bench.summary_of(out, extra)
    return out


def test_bench_line_is_short_strict_json_with_the_contract_keys():
    out = canned()
    assert len(json.dumps(out)) > 20000                     # (the full record is what r05 printed)
    text = bench.compact_line(out)
    assert len(text) < 4096 and '\n' not in text
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))      # strict: no NaN / Infinity
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in line, k
    assert line["config"]["workload"].startswith("K2") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] == 38070272
    assert {"kernel_ms", "executed_frac", "unit", "peak"} <= set(r)
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == c["threads"] == 16 and c["value_1thread"] > 0 and "sample" in c
    assert set(line["summary"]) >= {"K2_forward", "K2_train", "K4_3_18_euler", "K3_strong", "K3_shard_512", "K5_strong_train", "K5_shard_128_train"}
    assert all(isinstance(v, (int, float)) for leg in line["summary"].values() for v in leg.values())      # numbers only, no prose


def test_bench_line_without_extra_legs_or_cpu_baseline():
    out = canned()
    for k in ("extra", "summary", "cpu_baseline", "speedup_vs_cpu"):
        out.pop(k)
    line = json.loads(bench.compact_line(out))
    assert "roofline" in line and "cpu_baseline" not in line and "summary" not in line
