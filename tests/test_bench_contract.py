"""CPU checks of bench.py's pieces that need no GPU: the argument contract and the CPU baseline leg."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_arguments_and_defaults(monkeypatch):
    bench = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (4, 7, 3)


def test_cpu_baseline_leg_reports_the_contract_fields():
    bench = load_bench()
    r = bench.cpu_baseline(24, 12.0, sample_sweeps=2, threads=1)
    assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "Mcells/s"
    assert r["value"] > 0 and "sweeps" in r["sample"]
    r = bench.cpu_baseline(24, 12.0, sample_sweeps=2, threads=2)
    assert r["cores"] == 2 and "2 threads" in r["sample"]


def test_reference_order_cpu_baseline_is_one_thread_lexicographic():
    bench = load_bench()
    r = bench.cpu_baseline_reference_order(24, 12.0, sample_sweeps=2)
    assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "Mcells/s" and r["value"] > 0
    assert "lexicographic" in r["sample"] and "semi_lagrange" in r["sample"]


def test_step_byte_model():
    bench = load_bench()
    # fp32: 33 + 45 + 25 per sweep; fp64 doubles everything but the flag bytes
    assert bench.step_bytes_per_cell(2, 1.0) == 33 + 45 + 50
    assert bench.step_bytes_per_cell(1, 2.0) == 65 + 89 + 49
