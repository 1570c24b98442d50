"""bench.py's secondary-leg guard: the headline line must survive a failing or hanging leg."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_env_bound_counts_serial_envs_per_worker():
    import bench
    assert bench.env_bound(32, 4, 100.0, 16) == 40000.0          # 4 workers x 8 serial envs x 100 us
    assert bench.env_bound(32, 32, 100.0, 16) == 160000.0        # 32 workers on 16 CPUs: two rounds
    assert bench.env_bound(32, 32, 100.0, 64) == 320000.0
    assert bench.env_bound(32, 0, 100.0, 16) == 32 / (32 * 100e-6)   # in-process env: one lane
    assert bench.env_bound(32, 4, 0.0, 16) is None


def test_failing_leg_is_recorded_and_later_legs_still_run(capsys):
    import bench
    out = {"value": 1.0}
    os.environ["FSRL_BENCH_FAIL_LEG"] = "grouped"
    try:
        legs = bench.Legs(out, 0, 30.0)
    finally:
        del os.environ["FSRL_BENCH_FAIL_LEG"]
    assert legs.run("a", lambda: 3, 5.0) == 3
    assert legs.run("grouped", lambda: 4, 5.0) is None
    assert legs.run("b", lambda: 1 / 0, 5.0) is None
    assert legs.run("c", lambda: {"x": 1}, 5.0) == {"x": 1}
    legs.emit(); legs.emit()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["a"] == 3 and d["c"] == {"x": 1}
    assert "test hook" in d["grouped"]["error"] and "ZeroDivisionError" in d["b"]["error"]


def test_hanging_leg_trips_the_watchdog_and_the_headline_is_printed_once():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'value': 2.5}\n"
            "legs = bench.Legs(out, 0, 60.0)\n"
            "legs.run('first', lambda: 'ok', 5.0)\n"
            "legs.run('stuck', lambda: time.sleep(3600), 1.0)\n"
            "legs.run('never', lambda: 'no', 5.0)\n"
            "legs.emit()\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 2.5 and d["first"] == "ok" and "watchdog" in d["stuck"]["error"] and "never" not in d


def test_nonzero_rank_watchdog_exits_quietly():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "legs = bench.Legs(None, 1, 60.0)\n"
            "legs.run('stuck', lambda: time.sleep(3600), 1.0)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_configs_summary_is_the_last_key_and_compact(capsys):
    """the record keeps only the tail of the line: every BASELINE configuration's figures sit in ONE last key of <= 600 characters
    (built here from a bench line this repo recorded on an MI355X: profiles/r04_bench.json)"""
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    rec.pop("configs", None)
    legs = bench.Legs(rec, 0, 30.0)
    legs.emit()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][0]
    d = json.loads(line)
    assert list(d)[-1] == "configs"
    c = d["configs"]
    assert len(json.dumps(c)) <= 600, len(json.dumps(c))
    assert {"c1_ppo", "c2_cpo", "c1_trpo", "c3_sac", "kl_on"} <= set(c)
    assert abs(c["c2_cpo"]["ms"] - rec["cpo_c2"]["hip_ms_per_update"]) < 0.01 and c["c2_cpo"]["frac"] > 0 and c["c2_cpo"]["cpu_updates_s"] > 0
    assert abs(c["c3_sac"]["us"] - rec["sac_c3"]["ms_per_update"] * 1e3) < 0.06 and c["c3_sac"]["cpu_updates_s"] > 0
    assert c["c1_trpo"]["traffic_gb"] > 1 and c["c1_ppo"]["frac"] > 0
    # r6: both halves of the headline metric in the kept tail: env-steps/s of the training loop beside updates/s
    assert c["e2e"]["env_steps_s"] == round(rec["end_to_end"]["env_steps_per_s"]) and c["e2e"]["updates_s"] == round(rec["value"], 1)
    assert 0 < c["e2e"]["w32_b100_frac_of_bound"] <= 1.0


def test_configs_summary_carries_the_job_figures_at_n_gt_1():
    import bench
    out = {"value": 900.0, "ms_per_step": 8.9, "roofline": {"frac": 0.1, "step_us": 26.0},
           "end_to_end_job": {"env_steps_per_s": 2.5e6, "policy_updates_per_s": 400.0, "ranks": 8, "ranks_ok": 8},
           "torch_rocm_baseline": {"value": 1.23456}}
    c = bench.configs_summary(out)
    assert c["job"] == {"env_steps_s": 2500000.0, "updates_s": 900.0, "ranks_ok": 8}
    assert c["c1_ppo"]["torch_gpu_updates_s"] == 1.23 and "e2e" not in c
    assert len(json.dumps(c)) <= 600


def test_headline_workload_string_survives_the_records_cut():
    import bench, inspect
    src = inspect.getsource(bench.main)
    i = src.index('"config": {"workload": ')
    txt = "".join(__import__("re").findall(r'"([^"]*)"', src[i + 22:src.index('"seeds_per_gpu"', i)]))
    assert len(txt) < 120 and txt.startswith("configs[1] PPO-Lag update: repeat 4, clip 0.5"), (len(txt), txt)
