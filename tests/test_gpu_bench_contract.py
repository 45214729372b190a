"""-m gpu: bench.py keeps its contract with the driver -- the LAST stdout line is one JSON object carrying metric / value /
unit / n_gpus / steps / warmup / ms_per_step / roofline / cpu_baseline, for the default workload and for the other
BASELINE configurations' modes.  Runs the real script in a subprocess with short settings."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_LINES = {}


def _run(*flags, **extra_env):
    """bench.py with these flags -> its JSON line (an identical invocation is run once per session: several tests read one line)"""
    key = (flags, tuple(sorted(extra_env.items())))
    if key not in _LINES:
        _LINES[key] = _run_bench(*flags, **extra_env)
    return _LINES[key]


def _default_line():
    # the line as the driver takes it, with everything in it (other engine settings, the other BASELINE configs, a short CPU leg)
    return _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1")


def _run_bench(*flags, **extra_env):
    env = dict(os.environ, XM_BENCH_PREWARM_S="0.05", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=env,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    return json.loads(last)


def test_default_line_as_the_driver_calls_it():
    d = _default_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["value"] > 1000 and d["unit"] == "Mevents/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel"] == "k_scatter"
    assert 0.05 < r["frac"] < 1.0 and r["avg_launch_us"]["k_scatter"] > 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["parity"]["depth_bit_exact"] and d["parity"]["bgr_equal"]
    fps = d["config"]["frames_per_step"]  # a step = one group of frames through one xm_process_batch call
    assert fps == 32 and d["config"]["events_per_step"] == 32_000_000 and r["frames_per_launch"] == fps
    assert r["algorithmic_bytes_per_launch"] == 24.0 * 1e6 * fps and d["parity"]["last_frame_of_the_group_depth_bit_exact"]
    assert d["config"]["k1_paths_frames"]["cols"] > 0  # the groups took the column-tile K1
    assert abs(d["value"] - 1e6 * fps * 20 / (d["ms_per_step"] * 20 * 1e-3) / 1e6) / d["value"] < 0.01  # value == events / time
    # (the PCIe-inclusive figure is the box's link as much as the code: correctness and an order of magnitude are asserted here,
    #  the >= 1 Gevent/s verdict is reported in the line)
    assert d["host_path"]["depth_equals_oracle"] and d["host_path"]["Mevents_per_s_pinned_pipelined"] > 300, d["host_path"]
    assert d["ingest_path"]["first_frame_depth_equals_oracle"] and d["ingest_path"]["same_frames_as_host_trigger_finder"], d["ingest_path"]
    e3 = d["ingest_path"]["from_evt3_words"]  # the same stream as EVT 3.0 words, decoded on the device in front of the ingest
    assert e3["quarter_period_chunks"]["same_frames_as_from_eventcd_records"] and e3["period_chunks"]["same_frames_as_host_trigger_finder"]
    assert e3["period_chunks"]["Mevents_per_s_end_to_end"] > 500 and e3["period_chunks"]["bytes_per_event_over_pcie"] < 8


def test_the_default_line_carries_the_other_engine_settings():
    d = _default_line()
    om = d["other_modes"]
    for k in ("eventcd_records", "one_frame_per_call", "one_frame_per_call_eager", "forced_general", "camera_view"):
        assert om[k]["value"] > 1000 and om[k]["unit"] == "Mevents/s", k
    # the reference's own input layout (Metavision's 16-byte EventCD records) through xm_process_batch_aos
    assert om["eventcd_records"]["depth_equals_oracle"] and om["eventcd_records"]["k1_paths"]["cols"] > 0


def test_one_frame_per_call_mode_still_prints_the_line():
    # without adaptive batching every frame is three launches of its own (compact key frame for lone C-1M frames)
    d = _run("--batch", "0", "--no-adaptive", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-other-modes", "--no-host-path")
    assert d["config"]["frames_per_step"] == 1 and d["roofline"]["frames_per_launch"] == 1 and d["value"] > 1000
    assert d["parity"]["depth_bit_exact"] and d["config"]["k1_paths_frames"]["key32"] > 0
    # with it (the default of --batch 0) frames that arrive while the GPU is busy go out as groups: the column tiles
    d = _run("--batch", "0", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-other-modes", "--no-host-path")
    assert d["config"]["frames_per_step"] == 1 and d["value"] > 1000 and "ADAPTIVE" in d["config"]["launch"]
    assert d["parity"]["depth_bit_exact"] and d["config"]["k1_paths_frames"]["cols"] > 0


@pytest.mark.parametrize("flags,workload", [(("--graph", "--steps", "60"), "C-60x1M"), (("--sharded", "--steps", "5"), "C-10M"),
                                            (("--esl", "--steps", "20", "--no-host-path"), "C-ESL")])
def test_other_configurations_print_one_json_line(flags, workload):
    d = _run(*flags, "--no-cpu-baseline")
    assert workload in d["config"]["workload"] and d["value"] > 100 and d["ms_per_step"] > 0
    if "--graph" in flags:
        assert d["latency_us"]["batch_of_60_frames"]["p99"] >= d["latency_us"]["batch_of_60_frames"]["p50"] > 0
        assert all(v["depth_bit_exact"] for v in d["parity"].values())
    if "--sharded" in flags:
        assert d["scaling"] == "strong" and d["config"]["host_synchronisations_per_frame"] == 0 and d["parity"]["depth_bit_exact"]
    if "--esl" in flags:
        assert d["config"]["k1_geometry"]["mode"] == "own" and d["config"]["k1_paths_frames"]["cols"] > 0  # the owner-tile K1
        assert d["parity"]["group_last_frame_depth_bit_exact"] and d["other_modes"]["one_frame_per_call"]["value"] > 100
    _check_roofline(d["roofline"])


def test_a_columns_merge_that_fails_parity_falls_back_to_the_packed_keys_on_every_rank():
    # (the verdict is broadcast from rank 0: no rank leaves a collective on its own)
    d = _run("--sharded", "--steps", "5", "--no-cpu-baseline", XM_BENCH_TEST_FAIL_COLUMNS="1")
    assert d["config"]["merge"] == "all_reduce" and d["config"]["fell_back"]["from"] == "columns" and d["parity"]["depth_bit_exact"]
    d = _run("--sharded", "--steps", "5", "--no-cpu-baseline")
    assert d["config"]["merge"] == "columns" and d["config"]["fell_back"] is None and d["parity"]["no_piece_objected"]
    assert d["config"]["collectives_issued_by"].startswith("the library") and d["parity"]["library_communicator_frames_equal_oracle"]
    d = _run("--sharded", "--steps", "5", "--no-cpu-baseline", "--comm", "torch", "--lanes", "1")  # rounds 1-3's form
    assert d["config"]["collectives_issued_by"].startswith("torch.distributed") and d["config"]["frames_in_flight"] == 1


def _check_roofline(r):
    """Every mode's line carries the roofline object: the dominant kernel with the three fractions side by side (SURVEY 8(d)
    algorithmic bytes, counter bytes when a PMC summary of the workload is committed, the event stream's 14 B/event), and K1 / K2
    each with their own figures."""
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_counter_bytes", "event_stream_read_roofline_frac",
              "kernels", "avg_launch_us", "whole_frame", "timing"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["kernel"] in ("k_scatter", "k_frame")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.0 < r["frac"] < 1.5
    assert 0.0 < r["event_stream_read_roofline_frac"] < 1.0
    ks, kf = r["kernels"]["k_scatter"], r["kernels"]["k_frame"]
    for k in ("avg_launch_us", "us_per_frame", "algorithmic_bytes_per_launch", "frac_algorithmic", "hbm_bytes_per_launch_counters",
              "frac_counter_bytes"):
        assert k in ks and k in kf, k
    assert ks["frac_event_stream_read"] > 0 and kf["frac_own_minimal_bytes"] > 0
    assert ks["avg_launch_us"] > 0.5 and kf["avg_launch_us"] > 0.5
    if r["traffic"] is not None:
        assert r["frac_counter_bytes"] > 0


def test_default_and_single_frame_lines_carry_the_three_fractions():
    _check_roofline(_default_line()["roofline"])
    d = _run("--batch", "0", "--no-adaptive", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-other-modes", "--no-host-path")
    _check_roofline(d["roofline"])



def test_multi_gpu_line_carries_the_sharded_frame_beside_the_replicas():
    """With N > 1 ranks the default line reports frame-level replicas (no collective) and, beside it, one C-10M frame sharded by
    event index over the same ranks with the collective time listed separately.  One GPU here: the leg is forced through a
    one-rank process group (RCCL kernels run, nothing crosses xGMI)."""
    d = _run("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-host-path", "--no-other-configs", "--no-pmc",
             XM_BENCH_FORCE_DIST="1", XM_BENCH_FORCE_SHARDED_LEG="1")
    sh = d["other_modes"]["one_frame_sharded_over_the_ranks"]
    assert sh["scaling"] == "strong" and sh["value"] > 1000 and sh["parity"]["depth_bit_exact"]
    assert sh["collective_ms"]["key_frame_merge"] > 0 and sh["kernels_us"]["k_scatter"] > 0
    # the leg runs on the library's own communicators (one per lane), the torch.distributed form timed beside it
    assert sh["merge"] == "columns" and sh["frames_in_flight"] == 2 and sh["collectives_issued_by"].startswith("the library")
    assert sh["parity"]["library_communicator_frames_equal_oracle"] and sh["Mevents_per_s_via_torch_distributed"] > 1000


def test_the_default_line_carries_the_other_baseline_configs():
    """the one line the driver records proves every BASELINE config: compact legs for the ESL-like stand-in (configs 0 / 2, with the
    camera-like stream through the device ingest and through the processor), the 60-frame graph (config 4) and the sharded C-10M
    frame (config 3), each with value, ms_per_step, roofline fractions and parity"""
    d = _default_line()
    assert d["n_gpus"] == 1 and d["rccl_ranks_seen"] is None
    oc = d["other_configs"]
    for name in ("esl", "graph60", "sharded_c10m"):
        leg = oc[name]
        assert "error" not in leg, leg
        assert leg["value"] > 100 and leg["ms_per_step"] > 0 and leg["parity_ok"] is True, (name, leg)
        assert 0.0 < leg["roofline"]["frac"] < 1.5 and "frac_counter_bytes" in leg["roofline"], name
    ip = oc["esl"]["ingest_path"]
    assert ip["same_frames_as_host_trigger_finder"] and ip["first_frame_equals_oracle"] and ip["Mevents_per_s_end_to_end"] > 200
    assert ip["host_us_per_push"] < 20
    assert oc["esl"]["full_replay_through_processor_device_ingest"]["same_frames_as_host_path"]
    dp = oc["esl"]["full_replay_through_processor_default_params"]  # (the reference's call pattern, this build's default RuntimeParams)
    assert dp["same_frames_as_host_path"] and dp["every_pass_the_same_frames"] and dp["Mevents_per_s_end_to_end"] > 200, dp
    assert oc["esl"]["in_a_process_without_torch"]["ingest_path"]["same_frames_as_host_trigger_finder"]
    assert oc["graph60"]["latency_us"]["batch_of_60_frames"]["p50"] > 0 and oc["sharded_c10m"]["collective_ms"]["key_frame_merge"] > 0


def test_esl_leg_of_the_default_line_carries_the_stream_legs():
    """the camera-like stream through the ingest, through the processor and as EVT 3.0 words, in bench.py's process and in a child
    that never imports torch -- read off the ESL-like leg of the line the driver records (`bench.py --esl` prints the same legs in
    full: `stream_legs`)"""
    e = _default_line()["other_configs"]["esl"]
    ip = e["ingest_path"]
    assert ip["same_frames_as_host_trigger_finder"] and ip["first_frame_equals_oracle"] and ip["frames_cut"] > 20
    for k in ("ingest_path_depth_and_bgr", "ingest_path_fresh_arrays"):
        assert e[k]["same_frames_as_host_trigger_finder"] and e[k]["first_frame_equals_oracle"], k
    assert e["full_replay_through_processor_host_trigger_finder"]["frames_shown"] == ip["frames_cut"]
    assert e["full_replay_through_processor_device_ingest"]["same_frames_as_host_path"]
    assert e["full_replay_through_processor_default_params"]["same_frames_as_host_path"]
    assert e["from_evt3_words_period_chunks"]["overflow"] == 0 and e["from_evt3_words_period_chunks"]["frames_cut"] > 20
    # the same legs in a child process that never imports torch (the reference's situation): same frames, checked the same way
    ch = e["in_a_process_without_torch"]
    assert "error" not in ch, ch
    assert ch["ingest_path"]["same_frames_as_host_trigger_finder"] and ch["ingest_path"]["first_frame_equals_oracle"]
    assert ch["ingest_path"]["frames_cut"] == ip["frames_cut"] and ch["full_replay_through_processor_device_ingest"]["same_frames_as_host_path"]
