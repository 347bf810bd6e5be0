"""The bench.py output contract, checked on the latest committed set of bench lines (profiles/rNNx_*_bench.json, written by
`python bench.py ...` on an MI355X through tools/collect_profiles.sh): one JSON line with the keys the driver reads, BASELINE.json's metric, a roofline block for
the dominant kernel and, on the single-GPU run, the CPU baseline leg."""
import glob
import sys
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = max(os.path.basename(p).split("_")[0] for p in glob.glob(os.path.join(ROOT, "profiles", "r*_codec_bench.json")))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{TAG}_*_bench.json")))
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": (int, float),
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def _last_json(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"{path}: bench.py must print exactly one JSON line"
    return json.loads(lines[0])


def test_bench_lines_exist():
    names = {os.path.basename(p) for p in LINES}
    assert {f"{TAG}_{w}_bench.json" for w in ("codec", "lm", "gpt", "e2e1")} <= names
    if TAG >= "r05":        # round 5: the batched lines and the fp8 line have their own standalone lines, traces and counter summaries
        assert {f"{TAG}_{w}_bench.json" for w in ("lm32", "e2e32", "gpt_fp8")} <= names
        for w in ("lm32", "e2e32", "gpt_fp8"):
            assert os.path.exists(os.path.join(ROOT, "profiles", f"{TAG}_{w}_kernel_stats.csv")), w


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = _last_json(path)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), (k, d.get(k))
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"].get("frames_per_step_per_gpu", d["value"] * d["ms_per_step"] / 1e3) / d["ms_per_step"] * 1e3) \
        <= 0.02 * d["value"] or "frames_per_step_per_gpu" not in d["config"]
    r = d.get("roofline")
    if r is not None:
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
        assert "traffic" in r and (r["traffic"] is None or r["traffic"]["bytes_per_launch"] > 0)
    c = d.get("cpu_baseline")
    if c is not None:
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]


def test_default_workload_line_has_roofline_and_cpu_baseline():
    d = _last_json(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json"))
    # the dominant kernel is the three-plane bf16 GEMM: fp32 results, priced in algorithmic (fp32) flops against the bf16 dense peak / 6
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["peak"] == 419.4 and d["dtype"].startswith("f32")
    assert "gemm_win_b3" in d["roofline"]["kernel"] and d["roofline"]["x_f32_mfma_peak"] > 1.0
    assert d["cpu_baseline"]["kind"] == "port" and "configs[1]" in d["config"]["workload"]
    assert d["code_exact_match_vs_cpu_oracle"] == 1.0 and d["wav_rel_err_vs_cpu_oracle"] < 1e-3      # the parity sample rides on the default line
    assert d["timing"]["samples"] >= 50 and d["timing"]["median_ms"] <= d["timing"]["p95_ms"]
    # the trace / counter cross-checks quote the streamed kernel of the same round, and the matrix pipe's busy fraction at the
    # clock the power budget allowed sits next to the nominal-peak fraction
    r = d["roofline"]
    assert r["rocprof"]["source"].startswith(f"profiles/{TAG}_") and abs(r["rocprof"]["frac"] - r["frac"]) < 0.05
    assert r["traffic"]["source"].startswith(f"profiles/{TAG}_") and r["traffic"]["bytes_per_launch"] > 0
    m = r["mfma_pipe"]
    assert 0.3 < m["busy_frac"] <= 1.0 and 1.5 < m["shader_clock_ghz"] < 2.6 and m["peak_at_that_clock_tflops"] <= 425


def test_default_line_carries_the_north_star_sub_benchmarks():
    """the driver-run line times the batch-1 LM step and the batch-1 end-to-end frame too, each with its own
    roofline and CPU baseline, median + p95 over >= 50 frames."""
    d = _last_json(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json"))
    for key, what in (("lm_b1", "configs[2]"), ("e2e_b1", "configs[3]")):
        sub = d[key]
        assert what in sub["config"]["workload"] and sub["unit"] == "frames/s" and sub["value"] > 0
        assert sub["timing"]["samples"] >= 50 and sub["steps"] >= 50 and sub["ms_per_step"] > 0
        r = sub["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
        c = sub["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["e2e_b1"]["x_realtime_per_stream"] >= 10.0            # the north-star target: >= 10x real time end to end at batch 1


@pytest.mark.skipif(TAG < "r03", reason="sub-objects added in round 3")
def test_default_line_carries_every_baseline_config():
    """the driver-run line also times configs[2] at a full 3000-slot ring, configs[3] at its per-GPU size (32
    streams: LM alone and end to end) and configs[4] (GPT Qwen-0.5B shape, 32 streams) in bf16 hi+lo and with the fp8 block GEMMs --
    each with its own roofline block and CPU baseline leg; the codec CPU leg also carries the single-thread figure."""
    d = _last_json(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json"))
    for key, what, batch in (("lm_ctx3000", "configs[2]", 1), ("lm_b32", "configs[2]", 32), ("e2e_b32", "configs[3]", 32),
                             ("gpt_b32", "configs[4]", 32), ("gpt_b32_fp8", "configs[4]", 32)):
        sub = d[key]
        assert what in sub["config"]["workload"] and sub["unit"] == "frames/s" and sub["value"] > 0 and sub["ms_per_step"] > 0
        assert sub["config"].get("batch_per_gpu", sub["config"].get("streams_per_gpu")) == batch
        assert sub["timing"]["samples"] >= 50 and sub["steps"] >= 50
        c = sub["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
        r = sub.get("roofline")
        if key != "e2e_b32":
            assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert d["lm_ctx3000"]["config"]["context_frames"] >= 3000
    assert "fp8" in d["gpt_b32_fp8"]["config"]["gemm_precision"] and "fp8" not in d["gpt_b32"]["config"]["gemm_precision"]
    one = d["cpu_baseline"]["single_thread"]
    assert one["cores"] == 1 and one["value"] > 0 and one["unit"] == "frames/s" and one["sample"]


@pytest.mark.skipif(TAG < "r05", reason="trace-backed fractions for every line: round 5")
def test_every_line_of_the_summary_has_a_trace_backed_fraction():
    """all eight benchmark lines carry a non-null roofline fraction below 1 that follows from a kernel trace of THAT
    workload committed under profiles/ (the event-pair figures of the 8-20 us launches, which exceed the graph-replayed step, moved
    to `event_pairs` and are flagged)."""
    d = _last_json(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json"))
    s = d["summary"]
    for name in ("codec_b64", "lm_b1", "e2e_b1", "lm_ctx3000", "lm_b32", "e2e_b32", "gpt_b32", "gpt_b32_fp8"):
        assert isinstance(s[f"{name}_frac"], float) and 0 < s[f"{name}_frac"] < 1, (name, s[f"{name}_frac"])
    assert d["roofline"]["rocprof"]["source"] == f"profiles/{TAG}_codec_kernel_stats.csv"
    want = {"lm_b1": "lm", "e2e_b1": "e2e1", "lm_ctx3000": "lm", "lm_b32": "lm32", "e2e_b32": "e2e32", "gpt_b32": "gpt", "gpt_b32_fp8": "gpt_fp8"}
    for name, tag in want.items():
        r = d[name]["roofline"]
        src = f"profiles/{TAG}_{tag}_kernel_stats.csv"
        assert os.path.exists(os.path.join(ROOT, src)), src
        assert r["rocprof"]["source"] == src and src in r["frac_source"], (name, r.get("frac_source"))
        assert abs(r["frac"] - r["rocprof"]["frac"]) < 1e-9 and r["kernel_ms_per_step"] < d[name]["ms_per_step"], name
        ev = r.get("event_pairs")
        if ev is not None and ev.get("kernel_ms_per_step") and ev["kernel_ms_per_step"] > d[name]["ms_per_step"]:
            assert ev.get("exceeds_step") is True       # the inflated figure is labelled, and is not the headline


# ---- host logic of bench.py itself (no GPU needed)

def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_spawn_command_is_the_contract_launch_line():
    b = _bench_module()
    cmd = b.spawn_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_gpus_flag_without_torchrun_spawns_the_ranks(monkeypatch):
    """`python bench.py --gpus 8` from a plain shell must not die on WORLD_SIZE: it re-launches itself."""
    b = _bench_module()
    calls = []
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert "--nproc-per-node=2" in cmd and cmd[-4:] == ["--gpus", "2", "--steps", "3"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_timing_summary_is_median_and_p95():
    b = _bench_module()
    t = b._timing([float(i) for i in range(1, 101)])
    assert t["samples"] == 100 and t["median_ms"] == 50.5 and t["p95_ms"] == 95.0 and t["min_ms"] == 1.0


def test_summary_object_is_compact_and_complete():
    """the driver keeps a 2 000-character tail and scalar keys only -- the line ends with one flat `summary` object that
    carries every configuration's time, x-real-time and roofline fraction."""
    b = _bench_module()
    d = _last_json(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json"))
    names = ("lm_b1", "e2e_b1", "lm_ctx3000", "lm_b32", "e2e_b32", "gpt_b32", "gpt_b32_fp8")
    s = b.make_summary(d, {n: d[n] for n in names})
    assert all(isinstance(v, (int, float, type(None))) for v in s.values())
    for n in names:
        assert s[f"{n}_ms"] == d[n]["ms_per_step"] and s[f"{n}_xrt"] == d[n]["x_realtime_per_stream"]
    assert s["codec_b64_ms"] == d["ms_per_step"] and s["codec_b64_frac"] == d["roofline"]["frac"]
    assert len(json.dumps(s)) < 1400


@pytest.mark.skipif(TAG < "r04", reason="summary added in round 4")
def test_default_line_ends_with_the_summary():
    with open(os.path.join(ROOT, "profiles", f"{TAG}_codec_bench.json")) as f:
        line = [ln for ln in f.read().splitlines() if ln.startswith("{")][0]
    d = json.loads(line)
    assert list(d)[-1] == "summary" and line.rstrip().endswith("}}")
    assert '"summary"' in line[-2000:]
    assert d["summary"]["e2e_b1_xrt"] >= 10.0 and d["summary"]["codec_b64_code_match"] is not None


def _walk(obj, path=""):
    if isinstance(obj, dict):
        yield path, obj
        for k, v in obj.items():
            yield from _walk(v, f"{path}.{k}" if path else k)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            yield from _walk(v, f"{path}[{i}]")


@pytest.mark.skipif(TAG < "r06", reason="build ids next to the kernel traces exist since round 6")
@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_no_committed_line_quotes_a_stale_trace(path):
    """Every trace-backed figure of a committed line names the build of the library it was taken with, that build is the one the line
    itself ran (`loaded_build_id`), and no roofline block is marked stale (bench.py then quotes the whole-frame figure instead)."""
    d = _last_json(path)
    for where, blk in _walk(d):
        assert blk.get("stale") is not True, f"{os.path.basename(path)}: {where} is stale"
        if "loaded_build_id" in blk:
            assert blk["build_id"] == blk["loaded_build_id"], (where, blk["build_id"], blk["loaded_build_id"])


@pytest.mark.skipif(TAG < "r06", reason="build ids next to the kernel traces exist since round 6")
def test_committed_traces_belong_to_the_library_in_the_tree():
    """profiles/rNN_*_kernel_stats.meta.json (tools/profile_meta.py) vs rst_build_id of the library built from THIS tree: a kernel change
    after the collection run shows up here, not as a silently stale fraction on the driver's line."""
    from rstnet_amd import _lib
    mine = _lib.build_id()
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{TAG}_*_kernel_stats.meta.json")))
    assert metas, "no build ids next to the committed kernel traces"
    for m in metas:
        with open(m) as f:
            assert json.load(f)["build_id"] == mine, (os.path.basename(m), mine)
