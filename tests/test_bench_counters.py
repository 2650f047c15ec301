"""bench.py quotes the committed rocprofv3 counter summaries (profiles/rNN_pmc_traffic.json, rNN_issue.json) only while they
describe the kernels the run launched: same kernel sources, same launches per step. A renamed or changed kernel makes the line say
so instead of quoting stale traffic."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def write(tmp, name, sig):
    os.makedirs(os.path.join(tmp, "profiles"), exist_ok=True)
    with open(os.path.join(tmp, "profiles", name), "w") as f:
        json.dump({"_per_operator_call": {"change_dir_light_hbm_bytes": 123.0, "raymarch_hbm_bytes": 45.0}, "_signature": sig}, f)


def test_counter_summaries_are_quoted_only_for_the_kernels_they_were_collected_from(tmp_path, monkeypatch):
    tmp = str(tmp_path)
    monkeypatch.setattr(bench, "ROOT", tmp)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "aaaa")
    per_step = {"k_light_sweep": 2.1, "k_light_occlusion": 1.0, "k_raymarch_lit": 1.0}
    # nothing committed
    data, why = bench.committed_counters("pmc_traffic", per_step)
    assert data is None and "no profiles" in why
    # an older round's file without a signature is not quoted
    write(tmp, "r03_pmc_traffic.json", None)
    data, why = bench.committed_counters("pmc_traffic", per_step)
    assert data is None and "no launch signature" in why
    # the newest round wins; matching sources and launches: quoted
    write(tmp, "r04_pmc_traffic.json", {"kernel_source_hash": "aaaa", "launches_per_step": {"k_light_sweep": 2.0, "k_light_occlusion": 1.0, "k_raymarch_lit": 1.0}})
    data, why = bench.committed_counters("pmc_traffic", per_step)
    assert data is not None and why.endswith("r04_pmc_traffic.json")
    # other kernel sources: stale
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "bbbb")
    data, why = bench.committed_counters("pmc_traffic", per_step)
    assert data is None and "stale" in why
    # same sources, but the run launched something else per step (another path took the passes)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "aaaa")
    data, why = bench.committed_counters("pmc_traffic", dict(per_step, k_light_occlusion=2.0))
    assert data is None and "k_light_occlusion" in why


def test_kernel_source_hash_follows_the_sources():
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
