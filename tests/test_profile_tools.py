"""The measurement tools' own rules (no GPU): a per-kernel mean of a profile becomes a roofline entry only if the profiled
launches were the workload's (tools/valu_roofline.py: check_against_bench), and tools/pmc_summary.py reports n / min / max
of the dispatch durations so that a mixture shows."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_round_4s_c5_profile_is_refused_and_its_1080p_profile_is_not():
    """profiles/r04_pmc_c5_stream.json: 77 dispatches of streams_split_kernel, 72 of them 64-spp ramp launches -- mean 5.6 ms against
    a 28.8-ms step.  The rule refuses it; the 1080p / 64-spp profile, whose ramp launches ARE the workload, passes."""
    vr = _load("valu_roofline")
    c5 = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_c5_stream.json")))
    name, rec = next((k, v) for k, v in c5.items() if "streams_split_kernel" in k)
    bench = dict(c5["_bench"], kernel_ms_under_rocprof=c5["_bench"]["ms_per_step_under_rocprof"])
    with pytest.raises(vr.NotTheWorkload) as e:
        vr.check_against_bench(name, rec, bench)
    assert "not a profile of this workload" in str(e.value)
    glass = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_glass_stream.json")))
    name, rec = next((k, v) for k, v in glass.items() if "streams_split_kernel" in k)
    bench = dict(glass["_bench"], kernel_ms_under_rocprof=glass["_bench"]["ms_per_step_under_rocprof"])
    vr.check_against_bench(name, rec, bench)                       # 7.535 ms per call against 7.72 ms per step: within 10 %
    with pytest.raises(vr.NotTheWorkload):
        vr.check_against_bench(name, rec, glass["_bench"])         # no kernel time of the bench line to hold it against: refused too
    with pytest.raises(vr.NotTheWorkload) as e:
        vr.check_against_bench(name, dict(rec, one_launch_size=False, min_us=3950.0, max_us=28480.0), bench)
    assert "mixes launch sizes" in str(e.value)


def test_pmc_summary_reports_n_min_max_and_the_bench_lines_kernel_time(tmp_path):
    """A synthetic trace: one kernel launched 3 x 1 ms (ramp at another size) and 2 x 5 ms -> one_launch_size false; a second kernel of
    one size -> true; _bench carries kernel_ms, steps, the ramp and the binary's build id."""
    stats = tmp_path / "stats" / "host"
    stats.mkdir(parents=True)
    rows, t = [], 1000
    for d in (1_000_000, 1_000_000, 1_000_000, 5_000_000, 5_000_000):
        rows.append({"Kernel_Name": "void ptmi::(anonymous namespace)::streams_split_kernel<true, true>(ptmi::RenderArgs, ptmi::ItemArgs)", "Start_Timestamp": t, "End_Timestamp": t + d})
        t += d + 10
    for d in (200_000, 210_000):
        rows.append({"Kernel_Name": "ptmi::streams_slot_seeds_kernel(Planes)", "Start_Timestamp": t, "End_Timestamp": t + d})
        t += d + 10
    with open(stats / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, ["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        w.writerows(rows)
    pmc = tmp_path / "pmc_1" / "host"
    pmc.mkdir(parents=True)
    with open(pmc / "1_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, ["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow({"Kernel_Name": r["Kernel_Name"], "Counter_Name": "SQ_INSTS_VALU", "Counter_Value": 100.0})
    json.dump({"config": {"workload": "synthetic"}, "ms_per_step": 5.1, "steps": 2, "warmup": 0, "roofline": {"kernel_ms": 5.0},
               "ramp": {"spp_per_launch": 64, "launches": 3}, "binary_build_id": "0123456789abcdef"}, open(tmp_path / "bench.json", "w"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout)
    split = out["streams_split_kernel<true, true>"]
    assert (split["calls"], split["min_us"], split["median_us"], split["max_us"], split["one_launch_size"]) == (5, 1000.0, 1000.0, 5000.0, False)
    assert split["avg_us"] == 2600.0 and split["valu_wave_instr_per_call"] == 100
    assert out["streams_slot_seeds_kernel(Planes)"]["one_launch_size"] is True
    assert out["_bench"] == {"workload": "synthetic", "ms_per_step_under_rocprof": 5.1, "kernel_ms_under_rocprof": 5.0, "steps": 2, "warmup": 0,
                             "ramp": {"spp_per_launch": 64, "launches": 3}, "binary_build_id": "0123456789abcdef"}
    vr = _load("valu_roofline")
    with pytest.raises(vr.NotTheWorkload):
        vr.check_against_bench("split", split, out["_bench"])


def test_the_documents_and_the_committed_profiles_name_one_binary():
    """DESIGN.md section 7 says which binary its tables were measured on; the committed round-6 profiles (the VALU accountings bench.py reads, the PMC
    summaries, the part bounds, the traffic terms, the C0 call timings) and the fuzz record must carry that same build id -- a table of one binary
    next to a profile of another is the kind of evidence round 4's verdict took apart.  Since round 6 the id is a hash over the COMPILED code
    (_build.code_id): comment and documentation edits after the profile round leave every record valid."""
    import re
    # the measured tables of DESIGN.md are GENERATED from the committed profiles (tools/design_tables.py): they must be what the profiles say
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), "r06", "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(design.encode()) <= 27 * 1024, "DESIGN.md is what a maintainer reads: ~26 KB, generated tables included (narrative goes to HISTORY.md)"
    ids = set(re.findall(r"`ptmi_build_id\(\)` = `([0-9a-f]{16})`", design))
    assert len(ids) == 1, ids
    (build_id,) = ids

    def load(name):
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    assert load("r06_valu_roofline.json")["build_id"] == build_id
    streams = load("r06_valu_roofline_streams.json")
    assert set(streams) == {"streams", "s16_stream", "glass_tree", "glass_stream", "c5_tree", "c5_stream"}
    for key, entry in streams.items():
        assert "refused" not in entry, (key, entry.get("refused"))                            # every profile is its workload's
        assert entry["build_id"] == build_id, key
        assert abs(entry["kernel_us_in_profile"] - entry["bench_kernel_us_under_rocprof"]) <= 0.1 * entry["bench_kernel_us_under_rocprof"], key
    for w in ("c2", "streams", "s16_stream", "glass_tree", "glass_stream", "c5_tree", "c5_stream"):
        summary = load("r06_pmc_%s.json" % w)
        assert summary["_bench"]["binary_build_id"] == build_id, w
        assert summary["_bench"]["ramp"]["spp_per_launch"] == (512 if w.startswith("c5") else 64), w      # the ramp launched the workload's own sample count
        assert all(v["one_launch_size"] for k, v in summary.items() if k != "_bench" and not k.startswith("__amd_")), w      # (the runtime's own copy kernels move buffers of many sizes)
    bench = load("r06_bench.json")
    assert bench["binary_build_id"] == build_id == bench["code_id_now"]
    assert load("r06_c5_part.json")["binary_build_id"] == build_id and load("r06_c4_part.json")["binary_build_id"] == build_id
    assert all(v["build_id"].split("+")[0] == build_id or "+" in v["build_id"] for v in load("r06_traffic_terms.json")["variants"].values())
    assert load("r06_traffic_terms.json")["variants"]["product"]["build_id"] == build_id
    assert load("r06_c0_calls.json")["build_id"] == build_id
    assert ("build id %s" % build_id) in open(os.path.join(ROOT, "profiles", "r06_fuzz_campaign.txt")).read()
